"""Volume renderer (`NeRFRenderer`) for MI355X.

Same public surface as /root/reference/nvsf/nerf/models/renderer_dynamic.py:67-326 -- constructor
arguments, `run(rays_o, rays_d, time, cal_lidar_color, num_steps, upsample_steps, bg_color, perturb)`,
`render(..., staged, max_ray_batch)` and the result dictionaries -- with the per-ray arithmetic executed
by HIP kernels instead of a chain of torch elementwise / cumprod ops:

    near/far            nvsf_near_far_from_aabb            (camera; LiDAR uses the two range constants)
    z_vals, xyzs        nvsf_uniform_samples               (:155-169)
    sigma, geo_feat     self.density(...)                  (network)
    weights, ws, depth  nvsf_composite_uniform_weights_*   (:181-194, 216-221)
    rgbs                self.color(..., mask = w > 1e-4)   (network)
    image               nvsf_composite_uniform_image_*     (:224, 236-237)

A network may additionally provide `fused_uniform_render(...)` (see network_static.py): when gradients are
not being recorded the whole chain then runs as three fused kernels and the [N,T,3] positions, the masks
and the per-sample colours are never materialised.
"""
import math

import torch
import torch.nn as nn

from nvsf import field_ops as ops
from nvsf.nerf.raymarching import raymarching


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, density_scale=1, min_near=0.01, min_near_lidar=0.01, lidar_max_depth=0.81,
                 density_thresh=0.01, bg_radius=-1, active_sensor=False):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.min_near_lidar = min_near_lidar
        self.lidar_max_depth = lidar_max_depth
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        self.active_sensor = active_sensor
        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self._aabb_host = [float(v) for v in aabb.tolist()]
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())

    # -- to be provided by the field network -----------------------------------------------------
    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x, t=None, cal_lidar_color=False, **kwargs):
        raise NotImplementedError()

    def color(self, x, d, cal_lidar_color=False, mask=None, **kwargs):
        raise NotImplementedError()

    # ---------------------------------------------------------------------------------------------
    def _k_scale(self):
        # alpha = 1 - exp(-delta * density_scale * sigma), doubled in the exponent for an active sensor (:185-189)
        return float(self.density_scale) * (2.0 if self.active_sensor else 1.0)

    def _near_far(self, rays_o, rays_d, cal_lidar_color, aabb):
        N = rays_o.shape[0]
        if cal_lidar_color:
            nears = torch.full((N,), float(self.min_near_lidar), dtype=torch.float32, device=rays_o.device)
            fars = torch.full((N,), float(self.lidar_max_depth), dtype=torch.float32, device=rays_o.device)
            return nears, fars
        return raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)

    def run(self, rays_o, rays_d, time, cal_lidar_color=False, num_steps=768, upsample_steps=128, bg_color=None,
            perturb=False, **kwargs):
        """rays_o, rays_d: [B, N, 3] (B == 1).  Returns the reference's dictionary: depth / image /
        weights_sum (with a `_lidar` suffix for LiDAR rays) plus `weights` and `z_vals` [N, T]."""
        self.out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, T = rays_o.shape[0], int(num_steps)
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = self._near_far(rays_o, rays_d, cal_lidar_color, aabb)
        noise = torch.rand(N, T, dtype=torch.float32, device=rays_o.device) if perturb else None

        per_ray_bg = None
        bg_host = None
        if not cal_lidar_color:
            if self.bg_radius > 0:
                sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)
                per_ray_bg = self.background(sph, rays_d)
            elif bg_color is None:
                bg_host = [1.0] * self.out_dim
            elif torch.is_tensor(bg_color) and bg_color.numel() > self.out_dim:
                per_ray_bg = bg_color.reshape(-1, self.out_dim).to(rays_o.device, torch.float32)
            elif torch.is_tensor(bg_color):
                bg_host = [float(v) for v in bg_color.reshape(-1).expand(self.out_dim).tolist()] if bg_color.numel() == 1 \
                    else [float(v) for v in bg_color.reshape(-1).tolist()]
            else:
                bg_host = [float(bg_color)] * self.out_dim

        fused = getattr(self, "fused_uniform_render", None)
        if fused is not None and not torch.is_grad_enabled():
            z_vals, weights, weights_sum, depth, image = fused(rays_o, rays_d, nears, fars, T, aabb, noise, cal_lidar_color,
                                                               bg_host, time=time, **kwargs)
        else:
            z_vals, xyzs = ops.uniform_samples(rays_o, rays_d, nears, fars, T, aabb, noise)
            density_outputs = self.density(xyzs.view(-1, 3), time, cal_lidar_color, **kwargs)
            sigma = density_outputs["sigma"].view(N, T)
            weights, weights_sum, depth = ops.CompositeWeightsFn.apply(sigma, z_vals, nears, fars, self._k_scale())
            dirs = rays_d.view(-1, 1, 3).expand(N, T, 3)
            mask = weights > ops.W_THRESH
            extra = {k: v.view(N * T, -1) for k, v in density_outputs.items() if k != "sigma"}
            rgbs = self.color(xyzs.view(-1, 3), dirs.reshape(-1, 3), cal_lidar_color=cal_lidar_color, mask=mask.reshape(-1),
                              **extra)
            bg_dev = torch.tensor(bg_host, dtype=torch.float32, device=rays_o.device) if bg_host is not None else None
            image = ops.CompositeImageFn.apply(weights, rgbs.view(N, T, self.out_dim), weights_sum, bg_dev)
        if per_ray_bg is not None:
            image = image + (1 - weights_sum).unsqueeze(-1) * per_ray_bg

        image = image.view(*prefix, self.out_dim)
        depth = depth.view(*prefix)
        suffix = "_lidar" if cal_lidar_color else ""
        return {"depth" + suffix: depth, "image" + suffix: image, "weights_sum" + suffix: weights_sum, "weights": weights,
                "z_vals": z_vals}

    def render(self, rays_o, rays_d, time, cal_lidar_color=False, staged=False, max_ray_batch=4096, **kwargs):
        """`staged`: evaluate in chunks of max_ray_batch rays and keep only depth / image (:286-316)."""
        B, N = rays_o.shape[:2]
        if not staged:
            return self.run(rays_o, rays_d, time, cal_lidar_color=cal_lidar_color, **kwargs)
        out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        keys = ("depth_lidar", "image_lidar") if cal_lidar_color else ("depth", "image")
        depth = torch.empty((B, N), device=rays_o.device)
        image = torch.empty((B, N, out_dim), device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                part = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], time[b:b + 1],
                                cal_lidar_color=cal_lidar_color, **kwargs)
                depth[b:b + 1, head:tail] = part[keys[0]]
                image[b:b + 1, head:tail] = part[keys[1]]
        return {keys[0]: depth, keys[1]: image}
