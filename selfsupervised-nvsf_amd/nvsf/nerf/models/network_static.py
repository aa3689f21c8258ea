"""Static hash-grid field (BASELINE config 2: "hash-grid L=16 F=2, 2x64 MLP").

The static sub-graph of the reference's NeRFNetwork (network_dynamic.py:12-192): one 3-D multiresolution
hash grid per modality (hash_field.py:107-119) feeding the shared density MLP (`sigma_net`,
network_dynamic.py:125-135), and the same direction encoders + heads (`view_encoder_lidar` Frequency ->
`intensity_net` / `raydrop_net`, `view_encoder_camera` SH4 -> `color_net`, :108-114,138-189), with the same
`density` / `color` / `get_params` signatures and result conventions (LiDAR channel order = [raydrop,
intensity], :317).  The space-time parts (K-planes, time-sliced 2-D grids, flow field) live in
network_dynamic.py.

Two execution paths, numerically equivalent (tests/test_render_static_gpu.py):
  * operator path (`density` / `color`): stand-alone HIP operators with autograd -- used for training;
  * `fused_uniform_render`: three fused kernels per ray batch -- used whenever no gradient is recorded.
"""

import numpy as np
import torch

import tinycudann as tcnn
from nvsf import field_ops as ops
from nvsf import testing
from nvsf.nerf import activation
from nvsf.nerf.activation import trunc_exp
from nvsf.nerf.models.renderer_dynamic import NeRFRenderer


class NeRFNetworkStatic(NeRFRenderer):
    def __init__(self, base_resolution=16, max_resolution=2048, n_levels_hash=16, n_features_per_level_hash=2,
                 log2_hashmap_size=19, num_layers_sigma=2, hidden_dim_sigma=64, geo_feat_dim=15, num_layers_lidar=3,
                 hidden_dim_lidar=64, num_layers_color=3, hidden_dim_color=64, out_color_dim=3, out_lidar_color_dim=2,
                 num_frames=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.out_color_dim, self.out_lidar_color_dim = out_color_dim, out_lidar_color_dim
        self.num_frames = num_frames
        per_level_scale = float(np.exp2(np.log2(max_resolution / base_resolution) / (n_levels_hash - 1)))
        grid_cfg = {"otype": "HashGrid", "n_levels": n_levels_hash, "n_features_per_level": n_features_per_level_hash,
                    "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
                    "per_level_scale": per_level_scale}
        self.hash_encoder_lidar = tcnn.Encoding(n_input_dims=3, encoding_config=grid_cfg, seed=11)
        self.hash_encoder_camera = tcnn.Encoding(n_input_dims=3, encoding_config=grid_cfg, seed=12)

        def mlp(n_in, n_out, hidden, layers, seed):
            return tcnn.Network(n_input_dims=n_in, n_output_dims=n_out, seed=seed,
                                network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                                "n_neurons": hidden, "n_hidden_layers": layers - 1})

        self.sigma_net = mlp(self.hash_encoder_lidar.n_output_dims, 1 + geo_feat_dim, hidden_dim_sigma, num_layers_sigma, 21)
        self.view_encoder_lidar = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Frequency", "degree": 12})
        self.intensity_net = mlp(self.view_encoder_lidar.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar, 22)
        self.raydrop_net = mlp(self.view_encoder_lidar.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar, 23)
        self.view_encoder_camera = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "SphericalHarmonics", "degree": 4})
        self.color_net = mlp(self.view_encoder_camera.n_output_dims + geo_feat_dim, out_color_dim, hidden_dim_color,
                             num_layers_color, 24)

    # ---- operator path (signatures of network_dynamic.py:213, :290) ---------------------------------
    def density(self, x, t=None, cal_lidar_color=False, **kwargs):
        x = (x + self.bound) / (2 * self.bound)
        enc = self.hash_encoder_lidar if cal_lidar_color else self.hash_encoder_camera
        net = self.sigma_net
        if (x.is_cuda and x.dim() == 2 and enc.spec.D == 3 and net.spec.n_hidden <= 2 and net.spec.hidden == 64
                and net.spec.out_cols == 16 and testing.get("density_fn") == "fused"):
            # one autograd node for encode -> MLP -> trunc_exp / slice (ops.DensityFn): same forward kernels, leaner backward
            sigma, geo = ops.DensityFn.apply(x, enc.params, enc.table_f16(), enc.spec, net.params, net.weights_f16(), net.spec,
                                             activation._LO, activation._HI, ops.rows_hint(self), ops.train_context(self))
            return {"sigma": sigma, "geo_feat": geo}
        h = net(enc(x))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def density_from_rays(self, rays_o, rays_d, nears, fars, T, noise, cal_lidar_color, **kwargs):
        """Training forward from the rays (NeRFRenderer.run asks for it when gradients are recorded): sampler + unit-cube
        normalisation + hash grid + density MLP + trunc_exp as ONE autograd node and one launch (ops.DensityRaysFn; two launches in
        the level-sliced form, chosen as for the no-grad render).  Returns None where the fused kernels are not built (the caller
        then runs the operator chain through `density`)."""
        enc = self.hash_encoder_lidar if cal_lidar_color else self.hash_encoder_camera
        net = self.sigma_net
        if not (self.fused_train_forward and rays_o.is_cuda and ops.render_uniform_eligible(enc.spec, rays_o.shape[0] * T) and net.spec.n_hidden == 1
                and net.spec.hidden == 64 and net.spec.in_cols == 32 and net.spec.out_cols == 16):
            return None
        ray_length = float(self.lidar_max_depth - self.min_near_lidar) if cal_lidar_color else 2.0 * float(self.bound)
        sliced = ops.prefer_sliced(enc.spec, rays_o.shape[0], T, ray_length, float(self.bound))
        z_vals, sigma, geo, geo16 = ops.DensityRaysFn.apply(rays_o, rays_d, nears, fars, int(T), self._aabb_host, float(self.bound), noise,
                                                            enc.params, enc.table_f16(), enc.spec, net.params, net.weights_f16(), net.spec,
                                                            activation._LO, activation._HI, bool(sliced), ops.train_context(self))
        return {"z_vals": z_vals, "sigma": sigma, "geo_feat": geo, "geo16": geo16}

    fused_train_forward = True  # False (tests): NeRFRenderer.run takes the operator chain (uniform_samples -> density)
    fused_train_render = True   # False (tests): training forward as DensityRaysFn + compositor + heads nodes instead of ONE node

    def render_from_rays_train(self, rays_o, rays_d, nears, fars, T, noise, cal_lidar_color, bg_host, **kwargs):
        """The whole training forward of a ray batch as ONE autograd node (ops.RenderRaysFn: the evaluation render's kernels in their
        TRAIN form; the chain's backward kernels).  Returns (z_vals, weights, weights_sum, depth, image) or None where the fused
        kernels are not built (NeRFRenderer.run then composes DensityRaysFn / the operator chain)."""
        enc = self.hash_encoder_lidar if cal_lidar_color else self.hash_encoder_camera
        net = self.sigma_net
        if cal_lidar_color:
            head_a, head_b, venc = self.raydrop_net, self.intensity_net, self.view_encoder_lidar
        else:
            head_a, head_b, venc = self.color_net, None, self.view_encoder_camera
        hs = head_a.spec
        n_enc = venc.n_output_dims
        if not (self.fused_train_forward and self.fused_train_render and rays_o.is_cuda and ops.render_uniform_eligible(enc.spec, rays_o.shape[0] * T)
                and (T % 32 == 0 or not ops.sliced_only(enc.spec)) and net.spec.n_hidden == 1 and net.spec.hidden == 64 and net.spec.in_cols == 32 and net.spec.out_cols == 16
                and hs.n_hidden == 2 and hs.n_in == n_enc + 15 and n_enc % 8 == 0 and T % 16 == 0
                and (head_b is None or (head_b.spec.n_in == hs.n_in and head_b.spec.n_out == hs.n_out and head_b.spec.n_hidden == 2))):
            return None
        ray_length = float(self.lidar_max_depth - self.min_near_lidar) if cal_lidar_color else 2.0 * float(self.bound)
        sliced = ops.prefer_sliced(enc.spec, rays_o.shape[0], T, ray_length, float(self.bound)) and T % 32 == 0
        with torch.no_grad():  # per-ray direction encoding: the prefix rows the heads' backward reads
            d01 = (rays_d + 1) / 2
            enc_ray = ops.freq_encode(d01, venc.n_frequencies) if venc.otype == "Frequency" else ops.sh4_encode(d01)
        return ops.RenderRaysFn.apply(rays_o, rays_d, nears, fars, int(T), self._aabb_host, float(self.bound), noise, enc.params, enc.table_f16(),
                                      enc.spec, net.params, net.weights_f16(), net.spec, head_a.params, head_a.weights_f16(),
                                      None if head_b is None else head_b.params, None if head_b is None else head_b.weights_f16(), hs, enc_ray,
                                      n_enc, bool(cal_lidar_color), self._k_scale(), None if cal_lidar_color else bg_host, ops.W_THRESH,
                                      activation._LO, activation._HI, bool(sliced), ops.train_context(self))

    def color(self, x, d, cal_lidar_color=False, mask=None, geo_feat=None, **kwargs):
        ray_dirs = kwargs.get("ray_dirs")  # [N, 3] from NeRFRenderer.run: d is these rows, each repeated for its ray's samples
        if d is None:  # the renderer's fused training forward does not expand the directions: done here only if a path needs them
            d_rows = lambda: ray_dirs.view(-1, 1, 3).expand(ray_dirs.shape[0], geo_feat.shape[0] // ray_dirs.shape[0], 3).reshape(-1, 3)
        else:
            d_rows = lambda: d
        like = geo_feat if x is None else x  # dtype / device of the result
        dense_mask = None
        if mask is not None and ops.mask_was_dense(self, cal_lidar_color, mask):
            # the previous batch of this modality had >= 25 % of its samples above the weight threshold: evaluate all samples and
            # zero the rest without reading the count back first (values and gradients do not depend on this choice, only the
            # launch queue does: the read is a host sync that drains it)
            dense_mask, mask = mask, None
        if mask is not None:
            n_active = int(ops.count_true(mask)) if mask.is_cuda else int(mask.sum())  # one host sync (the reference's `mask.any()` costs the same)
            ops.note_mask_count(self, cal_lidar_color, n_active, mask.numel())
            if n_active == 0:
                return torch.zeros(mask.shape[0], self.out_dim, dtype=like.dtype, device=like.device)
            if 4 * n_active >= mask.numel():
                # most samples are active: evaluate all of them and zero the rest -- same values and gradients as the
                # gather / scatter form (masked rows are constants), without three index kernels each way
                dense_mask, mask = mask, None
            else:
                rgbs = torch.zeros(mask.shape[0], self.out_dim, dtype=like.dtype, device=like.device)
                d, geo_feat = d_rows()[mask], geo_feat[mask]
        # [direction encoding | geo_feat] assembled once in an aligned fp16 buffer shared by the heads (ops.HeadsFn); same
        # values as the reference's torch.cat + tcnn calls (network_dynamic.py:310-325), LiDAR order [raydrop, intensity]
        if mask is None and ray_dirs is not None and geo_feat.shape[0] % ray_dirs.shape[0] == 0:
            logits = ops.heads(self, None, geo_feat, cal_lidar_color, ray_dirs01=(ray_dirs + 1) / 2, geo16=kwargs.get("geo16"))
        else:
            d = (d_rows() if mask is None else d)
            d = (d + 1) / 2  # the direction encoders expect [0, 1]
            logits = ops.heads(self, d, geo_feat, cal_lidar_color)
        if dense_mask is not None and logits.is_cuda:
            return ops.MaskedSigmoidFn.apply(logits, dense_mask).to(like.dtype)  # sigmoid and mask in one launch
        h = torch.sigmoid(logits)
        if dense_mask is not None:
            return (h * dense_mask.unsqueeze(-1)).to(like.dtype)
        if mask is None:
            return h
        rgbs[mask] = h.to(rgbs.dtype)
        return rgbs

    # ---- fused path ------------------------------------------------------------------------------------
    def fused_uniform_render(self, rays_o, rays_d, nears, fars, T, aabb, noise, cal_lidar_color, bg_host, **kwargs):
        enc = self.hash_encoder_lidar if cal_lidar_color else self.hash_encoder_camera
        aabb_host = self._aabb_host  # host copy of the (constant) box: no device->host read on the hot path
        # host-side estimate of far - near: exact for LiDAR (constant range), the box side for camera rays (AABB exit)
        ray_length = float(self.lidar_max_depth - self.min_near_lidar) if cal_lidar_color else 2.0 * float(self.bound)
        sliced = ops.prefer_sliced(enc.spec, rays_o.shape[0], T, ray_length, float(self.bound), coherent=bool(kwargs.get("rays_in_image_order", False)))
        if ops.render_uniform_eligible(enc.spec, rays_o.shape[0] * T):
            # one launch per batch (plus the encode pass of the level-sliced path): sigma / geo never reach HBM
            if cal_lidar_color:
                head_a, head_b = self.raydrop_net.weights_f16(), self.intensity_net.weights_f16()
            else:
                head_a, head_b = self.color_net.weights_f16(), None
            return ops.render_uniform(rays_o, rays_d, nears, fars, T, aabb_host, float(self.bound), enc.table_f16(), enc.spec,
                                      self.sigma_net.weights_f16(), cal_lidar_color, head_a, head_b, self._k_scale(),
                                      None if cal_lidar_color else bg_host, noise, sliced=sliced)
        z_vals, sigmas, geo = ops.density_uniform(rays_o, rays_d, nears, fars, T, aabb_host, float(self.bound), enc.table_f16(),
                                                  enc.spec, self.sigma_net.weights_f16(), noise, sliced=sliced)
        weights, weights_sum, depth = ops.CompositeWeightsFn.apply(sigmas, z_vals, nears, fars, self._k_scale())
        if cal_lidar_color:
            image = ops.heads_uniform(weights, geo, rays_d, weights_sum, True, self.raydrop_net.weights_f16(),
                                      self.intensity_net.weights_f16(), None)
        else:
            image = ops.heads_uniform(weights, geo, rays_d, weights_sum, False, self.color_net.weights_f16(), None, bg_host)
        return z_vals, weights, weights_sum, depth, image

    def fused_occupancy_render(self, rays_o, rays_d, nears, fars, cal_lidar_color, dt_gamma, max_steps, T_thresh, bg_host):
        """Evaluation-mode occupancy render in one launch (NeRFRenderer.run_cuda hands over here when it can)."""
        enc = self.hash_encoder_lidar if cal_lidar_color else self.hash_encoder_camera
        if cal_lidar_color:
            head_a, head_b = self.raydrop_net.weights_f16(), self.intensity_net.weights_f16()
        else:
            head_a, head_b = self.color_net.weights_f16(), None
        return ops.render_occupancy(rays_o, rays_d, nears, fars, self.density_bitfield, float(self.bound), dt_gamma, max_steps,
                                    self.cascade, self.grid_size, enc.table_f16(), enc.spec, self.sigma_net.weights_f16(),
                                    cal_lidar_color, head_a, head_b, float(self.density_scale), T_thresh, bg_host)

    def get_params(self, lr):
        """Optimiser groups with the reference's per-group learning rates (network_dynamic.py:335-357)."""
        return [
            {"params": self.hash_encoder_lidar.parameters(), "lr": lr},
            {"params": self.hash_encoder_camera.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr},
            {"params": self.intensity_net.parameters(), "lr": 0.1 * lr},
            {"params": self.raydrop_net.parameters(), "lr": 0.1 * lr},
            {"params": self.color_net.parameters(), "lr": lr},
        ]
