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
        self.cuda_ray = False  # set by enable_occupancy_grid(): render() then marches the occupancy grid (run_cuda)

    # -- occupancy-grid acceleration (BASELINE config 3) ------------------------------------------------
    # The reference ships the kernels of this mode (raymarching.cu: march_rays_train, composite_rays_train,
    # march_rays, composite_rays, packbits, morton3D) and describes the intended call sequence in docstrings
    # (raymarching.py:192-212, 296-306, 389-409, 480-493) but contains no density grid and no caller (SURVEY
    # finding 1).  The driver below supplies that missing piece, following the call sequence of those docstrings.
    def enable_occupancy_grid(self):
        C, H = self.cascade, self.grid_size
        self.register_buffer("density_grid", torch.zeros(C, H ** 3))
        self.register_buffer("density_bitfield", torch.zeros(C * H ** 3 // 8, dtype=torch.uint8))
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))  # ring of (samples, rays) per step
        self.mean_density, self.iter_density, self.mean_count, self.local_step = 0.0, 0, 0, 0
        self.cuda_ray = True
        return self

    def set_density_grid(self, grid, thresh=None):
        """Installs a density grid [C, H^3] (Morton order per cascade) and rebuilds the bit field."""
        self.density_grid.copy_(grid.to(self.density_grid.device, torch.float32))
        self.mean_density = float(self.density_grid.clamp(min=0).mean())
        t = min(self.mean_density, self.density_thresh) if thresh is None else thresh
        raymarching.packbits(self.density_grid, t, self.density_bitfield)

    @torch.no_grad()
    def update_extra_state(self, time, cal_lidar_color=False, decay=0.95, S=128):
        """Refreshes the density grid from the field (exponential moving maximum) and re-packs the bit field.
        The first 16 calls visit every cell; later calls visit a random quarter of the cells plus a sample of the
        currently occupied ones."""
        C, H, dev = self.cascade, self.grid_size, self.density_grid.device
        tmp = -torch.ones_like(self.density_grid)

        def visit(coords, cas_list):
            idx = raymarching.morton3D(coords).long()
            unit = 2 * coords.float() / (H - 1) - 1
            for cas in cas_list:
                b = min(2 ** cas, self.bound)
                half = b / H
                xyz = unit * (b - half) + (torch.rand_like(unit) * 2 - 1) * half
                sig = self.density(xyz, time, cal_lidar_color)["sigma"].reshape(-1).float() * self.density_scale
                tmp[cas, idx] = sig

        if self.iter_density < 16:
            ax = torch.arange(H, dtype=torch.int32, device=dev)
            for xs in ax.split(S):
                for ys in ax.split(S):
                    for zs in ax.split(S):
                        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                        visit(torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1), range(C))
        else:
            n = H ** 3 // 4
            for cas in range(C):
                rnd = torch.randint(0, H, (n, 3), device=dev, dtype=torch.int32)
                occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                if occ.numel() > 0:
                    pick = occ[torch.randint(0, occ.numel(), (n,), device=dev)]
                    rnd = torch.cat([rnd, raymarching.morton3D_invert(pick.int())], 0)
                visit(rnd, [cas])
        valid = (self.density_grid >= 0) & (tmp >= 0)
        self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp[valid])
        self.mean_density = float(self.density_grid.clamp(min=0).mean())
        self.iter_density += 1
        raymarching.packbits(self.density_grid, min(self.mean_density, self.density_thresh), self.density_bitfield)
        n_steps = min(16, self.local_step)
        if n_steps > 0:
            self.mean_count = int(self.step_counter[:n_steps, 0].sum().item() / n_steps)
        self.local_step = 0

    def _packed_field(self, xyzs, dirs, time, cal_lidar_color):
        """sigma [M] and 3-channel colour [M,3] for packed samples (LiDAR: raydrop, intensity, 0)."""
        self.out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        if xyzs.shape[0] == 0:
            return xyzs.new_zeros(0), xyzs.new_zeros(0, 3)
        d = self.density(xyzs, time, cal_lidar_color)
        rgbs = self.color(xyzs, dirs, cal_lidar_color=cal_lidar_color, mask=None, geo_feat=d["geo_feat"]).float()
        if rgbs.shape[1] < 3:
            rgbs = torch.cat([rgbs, rgbs.new_zeros(rgbs.shape[0], 3 - rgbs.shape[1])], -1)
        return d["sigma"].float() * self.density_scale, rgbs

    def run_cuda(self, rays_o, rays_d, time, cal_lidar_color=False, dt_gamma=0, bg_color=None, perturb=False,
                 force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        """Occupancy-grid render: training mode packs all samples of the batch (march_rays_train ->
        field -> composite_rays_train, differentiable); evaluation mode advances the surviving rays a few steps
        at a time (march_rays -> field -> composite_rays) until all have terminated."""
        out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = self._near_far(rays_o, rays_d, cal_lidar_color, aabb)
        C, H = self.cascade, self.grid_size
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, C, H, nears, fars,
                                                                    counter, self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self._packed_field(xyzs, dirs, time, cal_lidar_color)
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
        elif (not perturb and not torch.is_grad_enabled() and hasattr(self, "fused_occupancy_render")
              and (bg_color is None or not torch.is_tensor(bg_color) or bg_color.numel() <= 3) and kwargs.get("fused", True)):
            # one launch for the whole survivor loop (static hash fields): nvsf_render_occupancy_fwd
            if cal_lidar_color:
                bg_host = None
            elif bg_color is None:
                bg_host = [1.0, 1.0, 1.0]
            elif torch.is_tensor(bg_color):
                bg_host = [float(v) for v in bg_color.reshape(-1).expand(3).tolist()] if bg_color.numel() == 1 \
                    else [float(v) for v in bg_color.reshape(-1).tolist()]
            else:
                bg_host = [float(bg_color)] * 3
            weights_sum, depth, image = self.fused_occupancy_render(rays_o, rays_d, nears, fars, cal_lidar_color, dt_gamma, max_steps,
                                                                    T_thresh, bg_host)
            suffix = "_lidar" if cal_lidar_color else ""
            return {"depth" + suffix: depth.view(*prefix), "image" + suffix: image.view(*prefix, out_dim), "weights_sum" + suffix: weights_sum}
        else:
            weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
            depth = torch.zeros(N, dtype=torch.float32, device=dev)
            image = torch.zeros(N, 3, dtype=torch.float32, device=dev)
            rays_alive = torch.arange(N, dtype=torch.int32, device=dev)
            rays_t = nears.clone()
            step = 0
            while step < max_steps:
                n_alive = rays_alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                            self.density_bitfield, C, H, nears, fars, 128, perturb if step == 0 else False,
                                                            dt_gamma, max_steps)
                sigmas, rgbs = self._packed_field(xyzs, dirs, time, cal_lidar_color)
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
                rays_alive = rays_alive[rays_alive >= 0]
                step += n_step
        image = image[:, :out_dim]
        if not cal_lidar_color:
            bg = 1 if bg_color is None else bg_color
            image = image + (1 - weights_sum).unsqueeze(-1) * bg
        suffix = "_lidar" if cal_lidar_color else ""
        return {"depth" + suffix: depth.view(*prefix), "image" + suffix: image.view(*prefix, out_dim), "weights_sum" + suffix: weights_sum}

    def _rows_hint(self):
        """The model's RowHint (field_ops), shared with every sub-module the first time it is asked for: the hash-grid autograd nodes
        of the encoders read it at forward time (samples per ray of the rows they are given)."""
        hint = self.__dict__.get("_row_hint")
        if hint is None:
            hint = ops.RowHint()
            for m in self.modules():
                m.__dict__["_row_hint"] = hint
        return hint

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
            # constants (renderer_dynamic.py:140-146): filled once per (N, device, values) and handed out read-only
            key = (N, rays_o.device, float(self.min_near_lidar), float(self.lidar_max_depth))
            if getattr(self, "_lidar_range_key", None) != key:
                self._lidar_range = (torch.full((N,), key[2], dtype=torch.float32, device=rays_o.device),
                                     torch.full((N,), key[3], dtype=torch.float32, device=rays_o.device))
                self._lidar_range_key = key
            return self._lidar_range
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
        rendered = None
        if fused is not None and not torch.is_grad_enabled():
            z_vals, weights, weights_sum, depth, image = fused(rays_o, rays_d, nears, fars, T, aabb, noise, cal_lidar_color,
                                                               bg_host, time=time, **kwargs)
        else:
            whole = getattr(self, "render_from_rays_train", None)
            rendered = whole(rays_o, rays_d, nears, fars, T, noise, cal_lidar_color, bg_host, **kwargs) if whole is not None else None
        if rendered is not None:
            # training forward of a field that renders a ray batch as ONE autograd node (network_static: ops.RenderRaysFn)
            z_vals, weights, weights_sum, depth, image = rendered
        elif fused is None or torch.is_grad_enabled():
            density_rays = getattr(self, "density_from_rays", None)
            density_outputs = density_rays(rays_o, rays_d, nears, fars, T, noise, cal_lidar_color, **kwargs) if density_rays is not None else None
            if density_outputs is not None:
                # training forward of a field that samples, encodes and evaluates its density MLP in one launch (network_static):
                # the [N, T, 3] positions and the expanded directions are never materialised; `color` gets them on request
                z_vals = density_outputs.pop("z_vals")
                xyz_arg = dirs_arg = None
            else:
                z_vals, xyzs = ops.uniform_samples(rays_o, rays_d, nears, fars, T, aabb, noise)
                with self._rows_hint().rows(T):  # rows are T consecutive samples per ray: lets the table scatters pick their form
                    density_outputs = self.density(xyzs.view(-1, 3), time, cal_lidar_color, **kwargs)
                xyz_arg, dirs_arg = xyzs.view(-1, 3), rays_d.view(-1, 1, 3).expand(N, T, 3).reshape(-1, 3)
            sigma = density_outputs["sigma"].view(N, T)
            weights, weights_sum, depth = ops.CompositeWeightsFn.apply(sigma, z_vals, nears, fars, self._k_scale())
            mask = weights > ops.W_THRESH
            extra = {k: v.view(N * T, -1) for k, v in density_outputs.items() if k != "sigma"}
            # ray_dirs: the per-ray rows `dirs` repeats -- lets the heads encode each direction once (ops.heads)
            rgbs = self.color(xyz_arg, dirs_arg, cal_lidar_color=cal_lidar_color, mask=mask.reshape(-1), ray_dirs=rays_d, **extra)
            bg_dev = ops.device_constant(bg_host, rays_o.device) if bg_host is not None else None
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
        _run = self.run_cuda if self.cuda_ray else self.run
        if not staged:
            return _run(rays_o, rays_d, time, cal_lidar_color=cal_lidar_color, **kwargs)
        out_dim = self.out_lidar_color_dim if cal_lidar_color else self.out_color_dim
        keys = ("depth_lidar", "image_lidar") if cal_lidar_color else ("depth", "image")
        depth = torch.empty((B, N), device=rays_o.device)
        image = torch.empty((B, N, out_dim), device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                # rays_in_image_order: a staged call walks a whole frame, chunk = consecutive pixels (trainer.py:1511-1524); a field
                # may choose its kernel formulation by that (network_static.fused_uniform_render)
                part = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], time[b:b + 1],
                                cal_lidar_color=cal_lidar_color, **dict(kwargs, rays_in_image_order=True))
                depth[b:b + 1, head:tail] = part[keys[0]]
                image[b:b + 1, head:tail] = part[keys[1]]
        return {keys[0]: depth, keys[1]: image}
