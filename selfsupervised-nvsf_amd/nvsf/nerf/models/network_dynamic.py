"""The full space-time field (`NeRFNetwork`) for MI355X.

Constructor arguments, sub-module names (hence state_dict keys), `density` / `color` / `flow` / `get_params`
signatures and results follow /root/reference/nvsf/nerf/models/network_dynamic.py:12-357:

  density(x, t):  x -> [0,1]^3;  per modality: static 3-D hash grid + time-sliced 2-D hash grids (HashGrid4D),
                  K-planes (Planes4D); the flow field warps x to the previous / next frame where the dynamic
                  encoders are evaluated again (hash grids without gradient, planes with gradient, :242-271);
                  dynamic features = 0.5 current + 0.25 (previous + next); the 120 features feed the density MLP
                  (120 -> 64 -> 16); sigma = trunc_exp(h0), geo_feat = h[1:].
  color(x, d, mask, geo_feat): direction encoding + heads on the masked samples, sigmoid, scatter (:290-332);
                  LiDAR channel order = [raydrop, intensity] (:317).

Every encoder / MLP evaluation is a HIP kernel (hash grid, K-planes, fused MLP, Frequency / SH); the glue between
them is the same tensor algebra as the reference.  Differences, all host-side: `t` is read back once per call
(the reference syncs >= 10 times per `density`); the modules the reference constructs but never uses
(`planes_encoder`, `hash_encoder`, `unet`: network_dynamic.py:47-65,192, excluded from the optimiser at :337-338)
are not instantiated, so DistributedDataParallel needs no find_unused_parameters.
"""

import numpy as np
import torch

import tinycudann as tcnn
from nvsf import field_ops as ops
from nvsf import testing
from nvsf.nerf.activation import trunc_exp
from nvsf.nerf.models.flow_field import FlowField
from nvsf.nerf.models.hash_field import HashGrid4D, _host_time
from nvsf.nerf.models.planes_field import Planes4D
from nvsf.nerf.models.renderer_dynamic import NeRFRenderer


class NeRFNetwork(NeRFRenderer):
    def __init__(self, min_resolution=32, base_resolution=512, max_resolution=32768, time_resolution=25, n_levels_plane=4,
                 n_features_per_level_plane=8, n_levels_hash=8, n_features_per_level_hash=4, log2_hashmap_size=19,
                 num_layers_flow=3, hidden_dim_flow=64, num_layers_sigma=2, hidden_dim_sigma=64, geo_feat_dim=15,
                 num_layers_lidar=3, hidden_dim_lidar=64, num_layers_color=3, hidden_dim_color=64, out_color_dim=3,
                 out_lidar_color_dim=2, num_frames=51, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.out_color_dim, self.out_lidar_color_dim = out_color_dim, out_lidar_color_dim
        self.num_frames = num_frames
        self.n_features_per_level_hash = n_features_per_level_hash

        def planes():
            return Planes4D(grid_dimensions=2, input_dim=4, output_dim=n_features_per_level_plane,
                            resolution=[min_resolution] * 3 + [time_resolution], multiscale_res=[2 ** n for n in range(n_levels_plane)],
                            concat_ms_feat=True, decompose=True)

        def hashes():
            return HashGrid4D(base_resolution=base_resolution, max_resolution=max_resolution, time_resolution=time_resolution,
                              n_levels=n_levels_hash, n_features_per_level=n_features_per_level_hash, log2_hashmap_size=log2_hashmap_size)

        def mlp(n_in, n_out, hidden, layers):
            return tcnn.Network(n_input_dims=n_in, n_output_dims=n_out,
                                network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                                "n_neurons": hidden, "n_hidden_layers": layers - 1})

        self.planes_encoder_lidar, self.hash_encoder_lidar = planes(), hashes()
        self.planes_encoder_camera, self.hash_encoder_camera = planes(), hashes()
        self.view_encoder_lidar = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Frequency", "degree": 12})
        self.flow_net = FlowField(input_dim=4, num_layers=num_layers_flow, hidden_dim=hidden_dim_flow, use_grid=True)
        self.sigma_net = mlp(self.planes_encoder_lidar.n_output_dims + self.hash_encoder_lidar.n_output_dims, 1 + geo_feat_dim,
                             hidden_dim_sigma, num_layers_sigma)
        self.intensity_net = mlp(self.view_encoder_lidar.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar)
        self.raydrop_net = mlp(self.view_encoder_lidar.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar)
        self.view_encoder_camera = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "SphericalHarmonics", "degree": 4})
        self.color_net = mlp(self.view_encoder_camera.n_output_dims + geo_feat_dim, self.out_color_dim, hidden_dim_color, num_layers_color)

    def forward(self, x, d):
        pass

    def _unit_cube(self, x):
        return (x + self.bound) / (2 * self.bound)

    def flow(self, x, t):
        x = self._unit_cube(x)
        if t.shape[0] == 1:
            t = t.repeat(x.shape[0], 1)
        f = self.flow_net(torch.cat([x, t], dim=-1))
        return {"flow_forward": f[:, :3], "flow_backward": f[:, 3:]}

    def _dynamic_features(self, x, t, cal_lidar_color, fp16=None):
        """x in [0,1]^3 -> (plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2), network_dynamic.py:220-271.
        `fp16`: the reference's `--fp16` option as it reaches render through `**vars(self.opt)` (trainer.py:200): the flow MLP's
        Linear layers then compute in fp16 with fp32 accumulation, as under the Trainer's autocast (flow_field.FlowField.forward)."""
        t_host = _host_time(t)  # the one device->host read of this call
        frame_idx = int(np.float32(t_host) * np.float32(self.num_frames - 1))
        hash_enc = self.hash_encoder_lidar if cal_lidar_color else self.hash_encoder_camera
        planes_enc = self.planes_encoder_lidar if cal_lidar_color else self.planes_encoder_camera

        if not torch.is_grad_enabled() and t.shape[0] == 1 and testing.get("dynamic_fused"):
            return self._dynamic_features_fused(x, t, t_host, frame_idx, hash_enc, planes_enc, fp16)
        t_col = t.repeat(x.shape[0], 1) if t.shape[0] == 1 else t
        xt = torch.cat([x, t_col], dim=-1)
        flow = self.flow_net(xt, t_host, fp16=fp16)
        # the K-planes of the query -- static + dynamic at (x, t), dynamic at the two flow-warped neighbour positions -- as ONE autograd node
        # (ops.PlanesMultiFn: one forward launch, one texel-scatter launch, one launch for the flow gradients) where it is built
        planes_node = (torch.is_grad_enabled() and testing.get("planes_train") == "fused" and x.is_cuda and t.shape[0] == 1
                       and len(planes_enc.multiscale_res) == 4 and not x.requires_grad)
        nb_time = [float(np.float32(f / self.num_frames)) if 0 <= f <= self.num_frames - 1 else None for f in (frame_idx + 1, frame_idx - 1)]
        if planes_node:
            planes_enc.wait_pending_update()
            # where the fused density tail consumes them, the three dynamic evaluations leave the node already blended (0.5 d + 0.25 (d1 +
            # d2), network_dynamic.py:273): the tail's own blend of (v, v, v) is v exactly, its gradient comes back as one slice
            planes_blend = self._tail_fused_in_training()
            plane_s, plane_d, pm_1, pm_2 = ops.PlanesMultiFn.apply(x, flow, planes_enc.planes_cl, planes_enc._res_host, float(np.float32(t_host)),
                                                                   nb_time[0], nb_time[1], ops.train_context(planes_enc), planes_blend)
            if planes_blend:
                pm_1 = pm_2 = plane_d
        else:
            plane_s, plane_d = planes_enc(xt)
        fused3 = hash_enc.training_fused3(x, t, t_host, flow, frame_idx, self.num_frames)
        if fused3 is not None:
            # level-major [8, M, 4] where the fused density tail (DensityTailFn) will take it: `density` turns it back into rows otherwise
            hash_s = hash_enc.forward_static(x, level_major=torch.is_grad_enabled() and self._tail_fused_in_training())
            hash_d, hash_1f, hash_2f = fused3
        else:
            hash_s, hash_d = hash_enc(x, t, t_host)

        def neighbour(offset, frame):
            """dynamic features at the flow-warped position in an adjacent frame (:242-271)"""
            tn = torch.tensor(frame / self.num_frames)
            xn = None if (fused3 is not None and planes_node) else x + offset  # (both fused nodes read x + flow inside their kernels)
            if fused3 is not None:
                hn = hash_1f if frame > frame_idx else hash_2f
            else:
                with torch.no_grad():
                    hn = hash_enc.forward_dynamic(xn, tn, float(np.float32(frame / self.num_frames)))
            if planes_node:
                return hn, (pm_1 if frame > frame_idx else pm_2)
            # the reference builds this column on the host and copies it (t1.repeat(N, 1).to(device), :250): a pageable
            # multi-megabyte H2D copy that also drains the stream; the same fp32 value is written on the device instead
            t_coln = torch.full((xn.shape[0], 1), float(tn), dtype=torch.float32, device=xn.device)
            pn = planes_enc.forward_dynamic(torch.cat([xn, t_coln], dim=-1))
            return hn, pn

        hash_1 = hash_2 = hash_d
        plane_1 = plane_2 = plane_d
        if frame_idx < self.num_frames - 1:
            hash_1, plane_1 = neighbour(flow[:, :3], frame_idx + 1)
        if frame_idx > 0:
            hash_2, plane_2 = neighbour(flow[:, 3:], frame_idx - 1)
        return plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2

    def _dynamic_features_fused(self, x, t, t_host, frame_idx, hash_enc, planes_enc, fp16=None):
        """The no-autograd form of _dynamic_features: same values, fewer launches and temporaries -- the neighbour hash
        evaluations read `x + flow` inside the kernel instead of from an [M, 3] temporary, the neighbour time columns are
        written once.  (A single K-planes launch for all three positions that keeps the base evaluation's texel quads in
        registers was measured and rejected: 0.90 ms against 3 x 0.21 ms -- the quads cost the occupancy that hides the
        gather latency, and the neighbours' re-gathers hit L1 anyway.)"""
        F = self.num_frames
        xt = torch.cat([x, t.float().expand(x.shape[0], 1)], dim=-1)
        flow = self.flow_net(xt, t_host, fp16=fp16)
        hash_s = hash_enc.forward_static(x, level_major=True)  # [8, M, 4] where available: _density_tail_fused reads either layout
        nb = []
        for col, frame in ((0, frame_idx + 1), (3, frame_idx - 1)):
            # 0-dim CPU tensor as in the reference (:244, :260): the fp16 regime of HashGridT
            nb.append((torch.tensor(frame / F), float(np.float32(frame / F)), col) if 0 <= frame <= F - 1 else None)
        # ONE launch for the three space-time evaluations: a neighbour re-uses the base gathers wherever its cell coincides
        hash_d, hash_1, hash_2 = hash_enc.forward_dynamic3(x, t, t_host, flow, nb)
        # ONE launch for the K-planes: static + dynamic at (x, t) + dynamic at the flow-warped positions of the neighbour
        # frames (x + flow read inside the kernel; the reference builds [M,4] copies, :250-252, :265-267)
        # ... and the blend 0.5 d + 0.25 (d1 + d2) of :273 is formed inside the kernel: the neighbour features never reach memory.
        # A missing neighbour (first / last frame) is the base evaluation itself, as in the reference (plane_feat_1 = plane_feat_d).
        base = (1, None, 0, float(np.float32(t_host)))
        evals = [(0, None, 0, float(np.float32(t_host))), base]
        for n_ in nb:
            evals.append(base if n_ is None else (1, flow, n_[2], float(n_[0])))
        # fp16 rows: the density kernel rounds its inputs to fp16 and (fp16)(0.5 v + 0.25 (v + v)) == (fp16)v, so the producer rounds
        plane_s, plane_d = planes_enc.forward_multi(x, evals, blend=True, out_f16=True)
        # plane_d is already blended: 0.5 v + 0.25 (v + v) == v exactly, so the density kernel's own blend leaves it unchanged
        plane_1 = plane_2 = plane_d
        return (plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_d if hash_1 is None else hash_1, hash_d if hash_2 is None else hash_2)

    def _tail_fused_in_training(self):
        return testing.get("density_tail_train") == "fused" and self.sigma_net.spec.in_cols == 128 and self.sigma_net.spec.n_hidden == 1

    def density(self, x, t=None, cal_lidar_color=False, **kwargs):
        plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2 = self._dynamic_features(self._unit_cube(x), t, cal_lidar_color,
                                                                                                     fp16=kwargs.get("fp16"))
        tail_train = (torch.is_grad_enabled() and self._tail_fused_in_training() and hash_s.dtype == torch.float16 and plane_s.dtype == torch.float32
                      and hash_d.dtype == torch.float32 and not hash_1.requires_grad and not hash_2.requires_grad)
        if torch.is_grad_enabled() and hash_s.dim() == 3 and not tail_train:  # level-major features without the node that reads them: rows
            hash_s = hash_s.permute(1, 0, 2).reshape(hash_s.shape[1], -1)
        if not torch.is_grad_enabled():
            # fused tail (csrc/density_dynamic.hip): neighbour blend + concatenation + density MLP in one kernel
            h = self._density_tail_fused(plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2)
        elif tail_train:
            # neighbour blend + concatenation + density MLP as ONE forward launch that also leaves the rounded network input
            # for the fused MLP backward; no [M,120] fp32 concatenation, no blend temporaries, in either direction
            sigma, geo = DensityTailFn.apply(self, plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2, self.sigma_net.params)
            return {"sigma": sigma, "geo_feat": geo}
        else:
            plane_d = 0.5 * plane_d + 0.25 * (plane_1 + plane_2)
            hash_d = 0.5 * hash_d + 0.25 * (hash_1 + hash_2)
            h = self.sigma_net(torch.cat([plane_s, plane_d, hash_s, hash_d], dim=-1))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def fused_uniform_render(self, rays_o, rays_d, nears, fars, T, aabb, noise, cal_lidar_color, bg_host, time=None, **kwargs):
        """No-autograd render of a uniform ray batch: sampler kernel -> encoders -> fused density tail (sigma + fp16
        geometry rows) -> weights kernel -> fused heads kernel; masks, per-sample colours and the [M,120] feature matrix are
        never materialised (renderer_dynamic.NeRFRenderer.run dispatches here)."""
        N = rays_o.shape[0]
        z_vals, xyzs = ops.uniform_samples(rays_o, rays_d, nears, fars, T, aabb, noise)
        feats = self._dynamic_features(self._unit_cube(xyzs.view(-1, 3)), time, cal_lidar_color, fp16=kwargs.get("fp16"))
        sigmas, geo = self._density_tail_fused(*feats, sigma_geo=True)
        weights, weights_sum, depth = ops.CompositeWeightsFn.apply(sigmas.view(N, T), z_vals, nears, fars, self._k_scale())
        if cal_lidar_color:
            image = ops.heads_uniform(weights, geo, rays_d, weights_sum, True, self.raydrop_net.weights_f16(), self.intensity_net.weights_f16(), None)
        else:
            image = ops.heads_uniform(weights, geo, rays_d, weights_sum, False, self.color_net.weights_f16(), None, bg_host)
        return z_vals, weights, weights_sum, depth, image

    def _density_tail_fused(self, plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2, sigma_geo=False, keep_input=False):
        from nvsf import _hip
        if self.sigma_net.spec.in_cols != 128 or self.sigma_net.spec.n_hidden != 1:
            raise NotImplementedError("fused density tail: 120 features, one hidden layer")
        M, dev = plane_s.shape[0], plane_s.device
        c = lambda a: a.contiguous()
        hash_d = hash_d.float()
        entry = "nvsf_density_dynamic_fwd"
        planes = [_hip.ptr(c(plane_s)), _hip.ptr(c(plane_d)), _hip.ptr(c(plane_1)), _hip.ptr(c(plane_2))]
        if plane_s.dtype == torch.float16:  # rows of forward_multi(..., out_f16=True): plane_d is the blend already
            if plane_d.dtype != torch.float16 or plane_1 is not plane_d or plane_2 is not plane_d:
                raise ValueError("fp16 plane rows come blended (plane_1 = plane_2 = plane_d)")
            entry, planes = "nvsf_density_dynamic_f16planes_fwd", planes[:2]
        if hash_s.dim() == 3:  # level-major static hash features (hash_field.forward_static(level_major=True))
            if tuple(hash_s.shape) != (8, M, 4) or hash_s.dtype != torch.float16:
                raise ValueError("level-major static hash features come as fp16 [8, M, 4]")
            entry = "nvsf_density_dynamic_lm_fwd" if entry == "nvsf_density_dynamic_f16planes_fwd" else "nvsf_density_dynamic_lm32_fwd"
        args = planes + [_hip.ptr(c(hash_s)), _hip.ptr(c(hash_d)),
                         _hip.ptr(c(hash_1)), 1 if hash_1.dtype == torch.float16 else 0, _hip.ptr(c(hash_2)), 1 if hash_2.dtype == torch.float16 else 0,
                         M, _hip.ptr(self.sigma_net.weights_f16())]
        if sigma_geo:
            sigmas = torch.empty(M, dtype=torch.float32, device=dev)
            geo = torch.empty(M, 16, dtype=torch.float16, device=dev)
            _hip.call(entry, *args, None, _hip.ptr(sigmas), _hip.ptr(geo), None)
            return sigmas, geo
        h = torch.empty(M, 16, dtype=torch.float32, device=dev)
        x16 = torch.empty(M, 128, dtype=torch.float16, device=dev) if keep_input else None
        _hip.call(entry, *args, _hip.ptr(h), None, None, _hip.ptr(x16))
        return (h, x16) if keep_input else h

    def color(self, x, d, cal_lidar_color=False, mask=None, geo_feat=None, **kwargs):
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
                return torch.zeros(mask.shape[0], self.out_dim, dtype=x.dtype, device=x.device)
            if 4 * n_active >= mask.numel():
                # most samples are active: evaluate all of them and zero the rest -- same values and gradients as the
                # gather / scatter form (masked rows are constants), without three index kernels each way
                dense_mask, mask = mask, None
            else:
                rgbs = torch.zeros(mask.shape[0], self.out_dim, dtype=x.dtype, device=x.device)
                d, geo_feat = d[mask], geo_feat[mask]
        # [direction encoding | geo_feat] assembled once in an aligned fp16 buffer shared by the heads (ops.HeadsFn); same
        # values as the reference's torch.cat + tcnn calls (network_dynamic.py:310-325), LiDAR order [raydrop, intensity]
        ray_dirs = kwargs.get("ray_dirs")  # [N, 3] from NeRFRenderer.run: d is these rows, each repeated for its ray's samples
        if mask is None and ray_dirs is not None and d.shape[0] % ray_dirs.shape[0] == 0:
            logits = ops.heads(self, None, geo_feat, cal_lidar_color, ray_dirs01=(ray_dirs + 1) / 2)
        else:
            d = (d + 1) / 2  # the direction encoders expect [0, 1]
            logits = ops.heads(self, d, geo_feat, cal_lidar_color)
        if dense_mask is not None and logits.is_cuda:
            return ops.MaskedSigmoidFn.apply(logits, dense_mask).to(x.dtype)  # sigmoid and mask in one launch
        h = torch.sigmoid(logits)
        if dense_mask is not None:
            return (h * dense_mask.unsqueeze(-1)).to(x.dtype)
        if mask is None:
            return h
        rgbs[mask] = h.to(rgbs.dtype)
        return rgbs

    def get_params(self, lr):
        """Optimiser groups and learning-rate ratios of network_dynamic.py:335-357."""
        groups = [(self.planes_encoder_lidar, 1.0), (self.hash_encoder_lidar, 1.0), (self.planes_encoder_camera, 1.0),
                  (self.hash_encoder_camera, 1.0), (self.view_encoder_lidar, 1.0), (self.view_encoder_camera, 1.0),
                  (self.flow_net, 0.1), (self.sigma_net, 1.0), (self.intensity_net, 0.1), (self.raydrop_net, 0.1), (self.color_net, 1.0)]
        params = [{"params": m.parameters(), "lr": r * lr} for m, r in groups]
        if self.bg_radius > 0:
            params.append({"params": self.encoder_bg.parameters(), "lr": lr})
            params.append({"params": self.bg_net.parameters(), "lr": lr})
        return params


class DensityTailFn(torch.autograd.Function):
    """h = sigma_net([plane_s | 0.5 plane_d + 0.25 (plane_1 + plane_2) | hash_s | 0.5 hash_d + 0.25 (hash_1 + hash_2)])
    (network_dynamic.py:273-287) with autograd: forward = csrc/density_dynamic.hip (also writes the rounded input),
    backward = nvsf_mlp_bwd on that input; the input gradient is handed back slice by slice with the blend factors."""

    @staticmethod
    def forward(ctx, net, plane_s, plane_d, plane_1, plane_2, hash_s, hash_d, hash_1, hash_2, params):
        h, x16 = net._density_tail_fused(plane_s.detach(), plane_d.detach(), plane_1.detach(), plane_2.detach(), hash_s.detach(),
                                         hash_d.detach(), hash_1.detach(), hash_2.detach(), keep_input=True)
        sigma = torch.exp(h[:, 0])  # trunc_exp's forward (activation.py:6-20); its backward is folded into the pass below
        ctx.save_for_backward(x16, net.sigma_net.weights_f16(), sigma)
        ctx.spec = net.sigma_net.spec
        ctx.hash_s_dtype = hash_s.dtype
        ctx.hash_s_lm = hash_s.dim() == 3
        # (plane_d, plane_1, plane_2) one tensor = the blend formed by its producer (ops.PlanesMultiFn(blend=True)): 0.5 v + 0.25 (v + v) = v,
        # and the gradient of that one tensor is the whole slice g[:, 32:64], unscaled (nvsf_density_tail_grad_split(plane_half_scale = 1))
        ctx.planes_blended = plane_1 is plane_d and plane_2 is plane_d
        return sigma, h[:, 1:net.sigma_net.spec.n_out]

    @staticmethod
    def _hash_s_grad(ctx, grad_x):
        g = grad_x[:, 64:96].to(ctx.hash_s_dtype)
        return g.reshape(-1, 8, 4).permute(1, 0, 2).contiguous() if ctx.hash_s_lm else g

    @staticmethod
    def backward(ctx, g_sigma, g_geo):
        from nvsf import _hip
        from nvsf.nerf import activation
        x16, w16, sigma = ctx.saved_tensors
        spec = ctx.spec
        need = ctx.needs_input_grad
        need_x = any(need[1:9])
        # logit gradient [g_sigma * clamp(sigma) | g_geo] in one pass (ops.DensityFn does the same for the static field)
        M = x16.shape[0]
        if g_sigma is not None:
            g_sigma = g_sigma.float().contiguous()
        if g_geo is not None and (g_geo.dtype != torch.float32 or g_geo.stride(1) != 1):
            g_geo = g_geo.float().contiguous()
        # the logit gradient formed inside the MLP backward where the pieces have the layout it reads (ops.mlp_backward(density_grad=...),
        # as the static field's node does), else by a pass of its own
        parts = None
        if testing.get("density_grad") == "composed" and x16.is_cuda:
            parts = ops.density_logit_gradient_parts(g_sigma, sigma, g_geo, None, spec.n_out - 1, (activation._LO, activation._HI))
        if parts is not None:
            grad_x, gw = ops.mlp_backward(x16[:, :spec.n_in], w16, spec, None, need_grad_x=need_x, density_grad=parts)
        else:
            grad_h = torch.empty(M, 16, dtype=torch.float32, device=x16.device)
            _hip.call("nvsf_sigma_geo_bwd", None if g_sigma is None else _hip.ptr(g_sigma), _hip.ptr(sigma),
                      None if g_geo is None else _hip.ptr_rows(g_geo), 0 if g_geo is None else g_geo.stride(0), spec.n_out - 1, M,
                      _hip.ptr(grad_h), 16, activation._LO, activation._HI)
            grad_h = grad_h[:, :spec.n_out]
            grad_x, gw = ops.mlp_backward(x16[:, :spec.n_in], w16, spec, grad_h, need_grad_x=need_x)
        out = [None] * 10
        if need_x and grad_x.is_cuda and grad_x.stride(1) == 1 and grad_x.stride(0) % 4 == 0 and grad_x.shape[1] >= 120:
            # the blend factors and the dtype of hash_s applied in ONE pass over the rows (nvsf_density_tail_grad_split)
            dev = grad_x.device
            f32 = dict(dtype=torch.float32, device=dev)
            blended = ctx.planes_blended and need[2]   # one tensor for (plane_d, plane_1, plane_2): its gradient is the whole slice, unscaled
            g_half = torch.empty(M, 32, **f32) if need[2] else None
            g_quarter = torch.empty(M, 32, **f32) if ((need[3] or need[4]) and not blended) else None
            g_hs = None
            if need[5] and ctx.hash_s_dtype in (torch.float16, torch.float32):  # in hash_s's own layout: rows [M, 32] or level-major [8, M, 4]
                g_hs = torch.empty((8, M, 4) if ctx.hash_s_lm else (M, 32), dtype=ctx.hash_s_dtype, device=dev)
            # hash_d's gradient column-major ([M, 24] with strides (1, M)): the table-gradient kernel of the space-time grids has every
            # workgroup read ONE column of it (hash_field.HashDynFn._backward passes such a tensor on as it is)
            g_hd = torch.empty(24, M, **f32).t() if need[6] else None
            g_ps = torch.empty(M, 32, **f32) if need[1] else None  # rows of its own: the planes' backward reads contiguous rows
            if g_half is not None or g_quarter is not None or g_hs is not None or g_hd is not None or g_ps is not None:
                _hip.call("nvsf_density_tail_grad_split", _hip.ptr(grad_x), grad_x.stride(0), M, _hip.ptr(g_half), _hip.ptr(g_quarter), _hip.ptr(g_hs),
                          1 if ctx.hash_s_dtype == torch.float16 else 0, 1 if ctx.hash_s_lm else 0, None if g_hd is None else g_hd.data_ptr(), 1,
                          _hip.ptr(g_ps), 1.0 if blended else 0.5)
            out[1], out[2], out[6] = g_ps, g_half, g_hd
            if need[3] and not blended:
                out[3] = g_quarter
            if need[4] and not blended:
                out[4] = g_quarter
            if need[5]:
                out[5] = g_hs if g_hs is not None else DensityTailFn._hash_s_grad(ctx, grad_x)
        elif need_x:
            g_pd = grad_x[:, 32:64]
            quarter = 0.25 * g_pd if (need[3] or need[4]) else None
            if need[1]:
                out[1] = grad_x[:, 0:32]
            if need[2]:
                out[2] = 0.5 * g_pd
            if need[3]:
                out[3] = quarter
            if need[4]:
                out[4] = quarter
            if need[5]:
                out[5] = DensityTailFn._hash_s_grad(ctx, grad_x)
            if need[6]:
                out[6] = 0.5 * grad_x[:, 96:120]
        if need[9]:
            out[9] = gw
        return tuple(out)
