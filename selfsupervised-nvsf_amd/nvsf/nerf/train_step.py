"""Thin multimodal training step + the two quality metrics of the headline benchmark.

Only what sits directly around the hot path, restated from the reference's `Trainer.train_step`
(nvsf/nerf/trainer.py:153-656) with the default CLI settings of nvsf/scripts/main_nvsf.py:60-97:
  * LiDAR: ground truth masked by the ray-drop channel (trainer.py:186-189), predictions masked the same way (:203-206),
    per-ray L1 range (alpha_d 1) + MSE ray-drop against label-smoothed targets (`--smooth_factor`, default 0.0; alpha_r 0.01)
    + MSE intensity (alpha_i 0.1), all `reduction="none"` and SUMMED over the rays (main_nvsf.py:205-221, trainer.py:540-543);
    chamfer distance between the predicted and the true point cloud `rays_d * depth / scale`, `(d1 + d2).mean() * 0.5`
    (trainer.py:229-233; it IS part of the total at :542), on csrc/chamfer.hip; optional scene-flow loss (`--flow_loss`,
    trainer.py:236-267: chamfer between the flow-warped point cloud of the frame and its neighbours' clouds + mean |flow|) and
    URF line-of-sight loss (trainer.py:276-294);
  * camera: MSE RGB summed over rays and channels (alpha_rgb 1; trainer.py:491-503);
  * NaN -> 0, Inf -> 1e5 on the scalar (trainer.py:545-546);
  * optimiser: Adam(betas 0.9/0.99, eps 1e-15) on `model.get_params(lr)` and the 0.1^(iter/iters) decay
    (main_nvsf.py:350-362), under the GradScaler of the reference's fp16 run (trainer.py:119, 1332-1334);
  * EMA of the weights, decay 0.95, updated once per epoch (trainer.py:112-114, 1420-1421; nvsf/nerf/ema.py);
  * PSNR (nvsf/lib/error_matrices.py:48-57) and depth RMSE in metres (error_matrices.py:263-285).
Across GPUs the step is frame-sharded: every rank renders its own frame, then ONE bucketed gradient all-reduce
(nvsf/frame_shard.py).  Checkpoints use the reference's dict (utils.py:622-747).
Shipped-config extras (configs/kitti360_1908.txt:13-15 `grad_loss`, `use_error_map`): the structural regularisation on LiDAR patches
in its `grad_loss` form (trainer.py:296-470: first differences (or Sobel gradients) of the rendered / true range inside pH x pW patches, masked by the true
ray-drop channel and the flatness of the true frame; `LidarGradLossFn`, csrc/losses.hip), the alternation between random pixels and
patches (trainer.py:1035-1062: `set_epoch`) and the error maps the patch sampler draws from (trainer.py:552-630: `update_error_maps`).
The regulariser's criteria (`--depth_grad_loss` l1 / mse / huber / smoothl1 / cos) and `--sobel_grad` are built; its three smoothness switches
fail in the reference itself (see __init__).  Logging and UNet refinement are out of scope.
"""
import math
import warnings

import numpy as np
import torch

from nvsf import frame_shard


def urf_line_of_sight_loss(weights, z_vals, gt_depth, eps):
    """Line-of-sight loss of Urban Radiance Fields as written at trainer.py:276-294."""
    gt = gt_depth.reshape(z_vals.shape[0], 1)
    n_valid = (gt > 0.0).sum()
    empty = (z_vals < gt - eps) | (z_vals > gt + eps)
    loss_empty = ((empty * weights) ** 2).sum() / n_valid
    near = (z_vals > gt - eps) & (z_vals < gt + eps)
    distance = near * (z_vals - gt)
    sigma = eps / 3.0
    distr = 1.0 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(distance ** 2 / (2 * sigma ** 2)))
    distr = distr / distr.max()
    distr = distr * near
    loss_near = ((near * weights - distr) ** 2).sum() / n_valid
    return 0.1 * loss_empty + 0.1 * loss_near


class LidarLossFn(torch.autograd.Function):
    """The LiDAR terms of Trainer.train_step (trainer.py:187-219) as one HIP launch forward and one backward (csrc/losses.hip): at
    4096 rays each of the ~60 elementwise / reduction launches the expression costs under autograd is a launch latency.
    Returns (loss_depth, loss_raydrop, loss_intensity, pred_depth [1,N], pred_points [1,N,3], gt_points [1,N,3]); the point clouds
    are the inputs of the chamfer term (None when `rays_d` is None)."""

    @staticmethod
    def forward(ctx, image_lidar, depth_lidar, gt_rd, gt_i, gt_d, rays_d, alpha_d, alpha_r, alpha_i, smooth, scale):
        from nvsf import _hip
        N, dev = depth_lidar.numel(), depth_lidar.device
        c = lambda t: t.detach().float().contiguous()
        img, dep, rd, gi, gd = c(image_lidar), c(depth_lidar), c(gt_rd), c(gt_i), c(gt_d)
        dirs = c(rays_d) if rays_d is not None else None
        l = [torch.empty((), dtype=torch.float32, device=dev) for _ in range(3)]
        pred_depth = torch.empty(1, N, dtype=torch.float32, device=dev)
        pp = torch.empty(1, N, 3, dtype=torch.float32, device=dev) if dirs is not None else None
        gp = torch.empty(1, N, 3, dtype=torch.float32, device=dev) if dirs is not None else None
        _hip.call("nvsf_lidar_losses_fwd", _hip.ptr(img), _hip.ptr(dep), _hip.ptr(rd), _hip.ptr(gi), _hip.ptr(gd), _hip.ptr(dirs), N,
                  float(alpha_d), float(alpha_r), float(alpha_i), float(smooth), float(scale), _hip.ptr(l[0]), _hip.ptr(l[1]), _hip.ptr(l[2]),
                  _hip.ptr(pred_depth), _hip.ptr(pp), _hip.ptr(gp))
        ctx.save_for_backward(img, dep, rd, gi, gd, dirs)
        ctx.consts = (float(alpha_d), float(alpha_r), float(alpha_i), float(smooth), float(scale), image_lidar.shape, depth_lidar.shape)
        if gp is not None:
            ctx.mark_non_differentiable(gp)
        return l[0], l[1], l[2], pred_depth, pp, gp

    @staticmethod
    def backward(ctx, g_d, g_r, g_i, g_pd, g_pp, _g_gp):
        from nvsf import _hip
        img, dep, rd, gi, gd, dirs = ctx.saved_tensors
        a_d, a_r, a_i, smooth, scale, shape_img, shape_dep = ctx.consts
        N = dep.numel()
        c = lambda t: None if t is None else t.float().contiguous()
        g_d, g_r, g_i, g_pd, g_pp = c(g_d), c(g_r), c(g_i), c(g_pd), c(g_pp)
        grad_img, grad_dep = torch.empty_like(img), torch.empty_like(dep)
        _hip.call("nvsf_lidar_losses_bwd", _hip.ptr(img), _hip.ptr(dep), _hip.ptr(rd), _hip.ptr(gi), _hip.ptr(gd), _hip.ptr(dirs), N,
                  a_d, a_r, a_i, smooth, scale, _hip.ptr(g_d), _hip.ptr(g_r), _hip.ptr(g_i), _hip.ptr(g_pd), _hip.ptr(g_pp),
                  _hip.ptr(grad_img), _hip.ptr(grad_dep))
        return (grad_img.view(shape_img), grad_dep.view(shape_dep)) + (None,) * 9


class MseSumFn(torch.autograd.Function):
    """sum(alpha (a - b)^2) -- the camera term (trainer.py:491-503, summed at :540-543) -- as one launch each way."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        from nvsf import _hip
        a32, b32 = a.detach().float().contiguous(), b.detach().float().contiguous()
        out = torch.empty((), dtype=torch.float32, device=a.device)
        _hip.call("nvsf_mse_sum_fwd", _hip.ptr(a32), _hip.ptr(b32), a32.numel(), float(alpha), _hip.ptr(out))
        ctx.save_for_backward(a32, b32)
        ctx.alpha, ctx.shape = float(alpha), a.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from nvsf import _hip
        a32, b32 = ctx.saved_tensors
        grad = torch.empty_like(a32)
        _hip.call("nvsf_mse_sum_bwd", _hip.ptr(a32), _hip.ptr(b32), a32.numel(), ctx.alpha, _hip.ptr(g.float().contiguous()), _hip.ptr(grad))
        return grad.view(ctx.shape), None, None


class LidarGradLossFn(torch.autograd.Function):
    """Structural regularisation of trainer.py:296-470 in its `grad_loss` form as one launch each way (nvsf_lidar_grad_loss_fwd / _bwd):
    pred_depth, gt_depth [1, N] the masked ranges (scene units), gt_raydrop [1, N], pano_inds int64 [1, N] pixel indices of the batch
    (N / (pH pW) patches in patch order), pano_frame [H, W, 3] (or [1, H, W, 3]) the frame's ground truth (channel 2 = range x scale).
    criterion: `--depth_grad_loss` (main_nvsf.py:86, 204-221); sobel: `--sobel_grad` (:107)."""
    CRITERIA = {"l1": 0, "mse": 1, "huber": 2, "smoothl1": 3, "cos": 4}

    @staticmethod
    def forward(ctx, pred_depth, gt_depth, gt_raydrop, pano_inds, pano_frame, patch, scale, criterion, alpha, sobel=False):
        from nvsf import _hip
        if criterion not in LidarGradLossFn.CRITERIA:
            raise ValueError(f"depth_grad_loss={criterion!r}: one of {sorted(LidarGradLossFn.CRITERIA)} (main_nvsf.py:86)")
        c = lambda t: t.detach().float().contiguous()
        pd, gd, rd = c(pred_depth), c(gt_depth), c(gt_raydrop)
        inds = pano_inds.detach().long().contiguous()
        frame = pano_frame.detach().float()
        frame = frame[0] if frame.dim() == 4 else frame
        frame = frame.contiguous()
        H, W, C = frame.shape
        if C < 3:  # the reference reads data['pano_frame'][0, ..., 2] (trainer.py:296-470): the range channel is channel 2
            raise ValueError(f"LidarGradLossFn: pano_frame needs >= 3 channels [raydrop, intensity, range], got {C}")
        pH, pW = int(patch[0]), int(patch[1])
        N = pd.numel()
        kind = LidarGradLossFn.CRITERIA[criterion]
        param = 0.2 * float(scale) if criterion == "huber" else (0.1 if criterion == "smoothl1" else 0.0)  # main_nvsf.py:207-208
        out = torch.empty((), dtype=torch.float32, device=pd.device)
        stats = torch.empty(N // (pH * pW), 6, dtype=torch.float32, device=pd.device) if criterion == "cos" else None
        args = (_hip.ptr(pd), _hip.ptr(gd), _hip.ptr(rd), _hip.ptr(inds), frame.data_ptr() + 4 * 2, C, N, pH, pW, H, W,
                float(scale), kind, float(param), float(alpha), 1 if sobel else 0, _hip.ptr(stats))
        _hip.call("nvsf_lidar_grad_loss_fwd", *args, _hip.ptr(out))
        ctx.save_for_backward(pd, gd, rd, inds, frame, *(() if stats is None else (stats,)))
        ctx.args, ctx.shape = args, pred_depth.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from nvsf import _hip
        pd = ctx.saved_tensors[0]
        grad = torch.empty_like(pd)
        _hip.call("nvsf_lidar_grad_loss_bwd", *ctx.args, _hip.ptr(g.float().contiguous()), _hip.ptr(grad))
        return (grad.view(ctx.shape),) + (None,) * 9


class RenderTrainStep:
    def __init__(self, model, lr=1e-2, iters=30000, num_steps=768, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, alpha_rgb=1.0,
                 smooth_factor=0.0, use_urf_loss=False, bucket_bytes=64 << 20, fp16=True, scale=1.0, chamfer_loss=True,
                 flow_loss=False, pc_list=None, ema_decay=0.95, split_backward=True, ray_chunks=1, grad_loss=False, depth_grad_loss="l1",
                 alpha_grad=0.1, patch_size_lidar=1, change_patch_size_lidar=(2, 8), change_patch_size_epoch=2, use_error_map=False,
                 sobel_grad=False, grad_norm_smooth=False, spatial_smooth=False, tv_loss=False):
        """Defaults = the reference's CLI defaults (main_nvsf.py:60-97).  `scale`: the scene scale the chamfer loss divides by
        (opt.scale); `pc_list`: {frame index: [P, 3] tensor} world-frame point clouds for the scene-flow loss
        (Trainer.process_pointcloud, trainer.py:1848-1912, builds them from the range images); `ema_decay=None` disables EMA.
        `split_backward`: the LiDAR terms and the camera term of the loss are sums over disjoint ray sets, so their gradients add:
        the step runs camera forward + backward, then LiDAR forward + backward, and the camera table scatter (the longest kernel
        of the step, on its side stream) overlaps the whole LiDAR pass instead of the tail of one joint backward.  Same
        gradients (test_train_step_gpu.py); the reference's `nan_to_num` of the total is applied per modality, which differs
        only when a loss is not finite (then that modality contributes no gradient; GradScaler skips such a step anyway)."""
        self.model = model
        # loss scaling of the reference's mixed-precision run (trainer.py:119, 1332-1334: GradScaler(enabled=fp16) -> scale(loss)
        # .backward() -> step -> update; `-L` / `--fp16` in main_nvsf.py:17,43,159).  The encoders hand fp16 features to the
        # MLPs, so the gradients that travel back between them are fp16 tensors.
        # LossScaler = GradScaler's rule, state_dict and kernels; it lets the step decide about an overflow before the last table
        # scatter has finished (nvsf/nerf/loss_scaler.py)
        from nvsf.nerf.loss_scaler import LossScaler
        self.scaler = LossScaler(enabled=bool(fp16) and torch.cuda.is_available())
        if fp16 and hasattr(model, "flow_net"):
            model.flow_net.flow_mlp_mode = "fused"  # the flow MLP as autocast runs it: fp16 MFMA kernels (flow_field.FlowMlpFn)
        # Adam as one HIP pass per parameter tensor with the scaler's overflow flag consumed on the device (nvsf/nerf/adam.py).
        # torch's own fused=True variant was tried first: under this eps = 1e-15 / GradScaler setting it trains measurably worse
        # (tests/test_train_step_gpu.py: loss 0.289 -> 0.258 after 120 steps against 0.227).
        on_gpu = torch.cuda.is_available() and next(model.parameters()).is_cuda
        if on_gpu:
            from nvsf.nerf.adam import FusedAdam
            self.opt = FusedAdam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        else:  # host-side logic tests only (checkpoint schema, gloo): the product path above needs the HIP device
            self.opt = torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda it: 0.1 ** min(it / iters, 1))
        self.iters, self.num_steps = iters, num_steps
        self.alpha_d, self.alpha_r, self.alpha_i, self.alpha_rgb = alpha_d, alpha_r, alpha_i, alpha_rgb
        self.smooth, self.use_urf, self.bucket_bytes = smooth_factor, use_urf_loss, bucket_bytes
        self.scale, self.use_chamfer, self.use_flow, self.pc_list = float(scale), bool(chamfer_loss), bool(flow_loss), pc_list or {}
        # structural regularisation on LiDAR patches + the error maps of the patch sampler (main_nvsf.py:79-111 defaults; the shipped
        # config switches grad_loss and use_error_map on, configs/kitti360_1908.txt:13-14)
        self.grad_loss, self.depth_grad_loss, self.alpha_grad = bool(grad_loss), str(depth_grad_loss), float(alpha_grad)
        self.sobel_grad = bool(sobel_grad)
        if grad_norm_smooth or spatial_smooth or tv_loss:
            # trainer.py:337-350 add a [num_patch, 1, pH, pW] tensor to the loss, so that `loss` is no scalar any more and
            # scaler.scale(loss).backward() (trainer.py:1332) raises: the three switches cannot be trained with in the reference
            raise NotImplementedError("grad_norm_smooth / spatial_smooth / tv_loss make the reference's loss a non-scalar tensor "
                                      "(trainer.py:337-350, 543-545) that its own backward call rejects; there is no behaviour to match")
        self.patch_size_lidar = patch_size_lidar
        self.change_patch_size_lidar, self.change_patch_size_epoch = tuple(change_patch_size_lidar), int(change_patch_size_epoch)
        self.use_error_map, self.pixel_sampler = bool(use_error_map), "random"
        self.error_maps = None   # the FrameSet whose error_map / error_map_rgb the step keeps up to date (attach_error_maps)
        self._for_error_map = {}
        self.ema = None
        if ema_decay is not None and on_gpu:
            from nvsf.nerf.ema import ExponentialMovingAverage
            self.ema = ExponentialMovingAverage(model.parameters(), decay=ema_decay)
            self.ema.before_access = self.sync  # evaluate_frames(ema=step.ema) right after step(): store / copy_to / restore wait
        self._cham = None
        self.global_step = 0
        self.failed_to_load = []  # components of the last load_checkpoint whose state could not be restored
        self.scatter_overlap = True  # table scatters on a side stream beside the rest of backward (field_ops.DensityFn)
        self.split_backward = bool(split_backward)
        self.ray_chunks = max(1, int(ray_chunks))
        # more than one rank: gradients live in flat buckets that are all-reduced while backward still runs (frame_shard.GradBuckets)
        # what the table-scatter nodes of the model need to know about this step (gradient sink, side-stream overlap, scatters still
        # expected): an object of the step, attached to the model -- not process-wide state
        from nvsf import field_ops
        self.train_ctx = field_ops.TrainContext()
        for mod in model.modules():
            mod.__dict__["_train_ctx"] = self.train_ctx
        self.buckets = None
        if frame_shard.world()[1] > 1:
            self.buckets = frame_shard.GradBuckets([p for g in self.opt.param_groups for p in g["params"]], bucket_bytes, train_ctx=self.train_ctx)
        # One process: the optimiser pass of the table whose scatter ran LAST is issued on that scatter's stream, behind it, and the
        # main stream goes on with the next step (whose first pass -- the camera's -- does not read that table): `defer_last_table`.
        self.defer_last_table = on_gpu and self.buckets is None
        self._sink = None
        self._pending = None  # (event behind the deferred optimiser pass, its parameters)
        self._late_carry = None  # overflow found ONLY in a deferred table's gradient: reaches the scaler one step later (step())
        if self.defer_last_table:
            # a direct model.state_dict() / optimiser state_dict right after step() must see the deferred pass finished as well
            # (the fp16-cache and K-planes readers wait by themselves; fp32 readers outside this package go through these hooks)
            import weakref
            me = weakref.ref(self)
            settle = lambda *a, **k: (me() is not None and me().sync()) and None
            model.register_state_dict_pre_hook(settle)
            self.opt.register_state_dict_pre_hook(lambda *a, **k: settle())
        # parameter -> the Planes4D module that owns it (its kernels read the fp32 parameter: planes_field.Planes4D.wait_pending_update)
        self._plane_owner = {mod.planes_cl: mod for mod in model.modules() if isinstance(getattr(mod, "planes_cl", None), torch.nn.Parameter)}
        self._cache_of = {}   # parameter -> the fp16 cache of the module that owns it (tinycudann.Encoding / Network)
        for mod in model.modules():
            if hasattr(mod, "_cache") and isinstance(getattr(mod, "params", None), torch.nn.Parameter):
                self._cache_of[mod.params] = mod._cache
        if on_gpu:  # the optimiser pass writes the fp16 copies the forward kernels read (no cast pass per parameter and step)
            self.opt.half_caches = {p: c for p, c in self._cache_of.items() if p.numel() > 0}

    def _render(self, rays_o, rays_d, time, **kw):
        """model.render on `ray_chunks` slices of the batch, results concatenated.  Each slice is its own autograd sub-graph, and
        backward walks them one after the other (last slice first), so the table scatter of a slice (side stream) runs beside the
        backward of the next slice, and the activations of one slice at a time are alive (peak memory 1.9 -> 1.4 GiB at 4
        slices).  Measured on one MI355X it does NOT shorten the step (10.6 ms at 1 slice, 10.9 at 2, 12.7 at 4): the scatter
        saturates the memory-side atomic path -- read-modify-write of a 64-B segment per 8-16 useful bytes, about 5 TB/s of
        HBM traffic -- and whatever runs beside it is slowed by what it gains.  Default 1; a memory knob, not a speed knob.
        Per-ray results do not depend on the slicing; only the order in which the fp32 table gradients are added changes."""
        N = rays_o.shape[1]
        k = min(self.ray_chunks, max(1, N // 256))
        if k <= 1:
            return self.model.render(rays_o, rays_d, time, **kw)
        cuts = [N * i // k for i in range(k + 1)]
        outs = [self.model.render(rays_o[:, a:b], rays_d[:, a:b], time, **kw) for a, b in zip(cuts[:-1], cuts[1:])]
        merged = {}
        for key, v in outs[0].items():
            if not torch.is_tensor(v):
                merged[key] = v
                continue
            dim = 1 if (v.dim() >= 2 and v.shape[0] == 1 and v.shape[1] == cuts[1] - cuts[0]) else 0
            merged[key] = torch.cat([o[key] for o in outs], dim=dim)
        return merged

    def _chamfer(self, a, b):
        if self._cham is None:
            from nvsf.nerf.chamfer3D.dist_chamfer_3D import chamfer_3DDist
            self._cham = chamfer_3DDist()
        return self._cham(a, b)

    def flow_loss(self, time):
        """Scene-flow loss of trainer.py:236-267: the frame's point cloud moved by the predicted forward / backward flow against
        the next / previous frame's cloud (sum of squared NN distances both ways, x 0.5) + the mean absolute flow."""
        frame_idx = int(float(time.reshape(-1)[0]) * (self.model.num_frames - 1))
        pc = self.pc_list.get(frame_idx)
        if pc is None:
            return None
        pc = pc.float().contiguous()
        pred = self.model.flow(pc, time)
        total = None
        for key, other in (("flow_forward", frame_idx + 1), ("flow_backward", frame_idx - 1)):
            tgt = self.pc_list.get(other)
            if tgt is None:
                continue
            d1, d2, _, _ = self._chamfer((pc + pred[key]).unsqueeze(0), tgt.float().contiguous().unsqueeze(0))
            term = (d1.sum() + d2.sum()) * 0.5 + pred[key].abs().mean()
            total = term if total is None else total + term
        return total

    def losses(self, batch):
        """batch: rays_o_lidar, rays_d_lidar [1,N,3] with either images_lidar [1,N,3] = (raydrop, intensity, range) as the
        reference's loader delivers them or gt_raydrop / gt_intensity / gt_depth [1,N]; rays_o, rays_d [1,N,3] with gt_rgb
        (or images) [1,N,3]; time [1,1].  Returns (total, parts)."""
        m, out = self.model, {}
        if "rays_o_lidar" in batch:
            if "images_lidar" in batch:
                gt_rd, gt_i, gt_d = (batch["images_lidar"][:, :, k] for k in range(3))
            else:
                gt_rd, gt_i, gt_d = batch["gt_raydrop"], batch["gt_intensity"], batch["gt_depth"]
            r = self._render(batch["rays_o_lidar"], batch["rays_d_lidar"], batch["time"], cal_lidar_color=True, perturb=True,
                             num_steps=self.num_steps)
            if r["depth_lidar"].is_cuda:  # one HIP launch each way for the three sums, the masked range and the chamfer point clouds
                d = batch["rays_d_lidar"] if self.use_chamfer else None
                out["depth"], out["raydrop"], out["intensity"], pred_depth, pred_pts, gt_pts = LidarLossFn.apply(
                    r["image_lidar"], r["depth_lidar"], gt_rd, gt_i, gt_d, d, self.alpha_d, self.alpha_r, self.alpha_i, self.smooth, self.scale)
                if self.use_chamfer:
                    d1, d2, _, _ = self._chamfer(pred_pts, gt_pts)
                    out["chamfer"] = (d1 + d2).mean() * 0.5
                pH, pW = self._patch_dims()
                if pH > 1 and self.grad_loss:
                    if "rays_pano_inds" not in batch or "pano_frame" not in batch:
                        raise ValueError("grad_loss on LiDAR patches needs batch['rays_pano_inds'] and batch['pano_frame'] (FrameSet.train_batch)")
                    out["sr"] = LidarGradLossFn.apply(pred_depth, gt_d * gt_rd, gt_rd, batch["rays_pano_inds"], batch["pano_frame"], (pH, pW),
                                                      self.scale, self.depth_grad_loss, self.alpha_grad, self.sobel_grad)
                if self.error_maps is not None:
                    self._for_error_map["lidar"] = (r["image_lidar"].detach(), r["depth_lidar"].detach(), gt_rd, gt_i, gt_d)
            else:  # host-side logic tests: the same terms as torch expressions
                if self.grad_loss and self._patch_dims()[0] > 1:
                    raise NotImplementedError("grad_loss (the structural regulariser on LiDAR patches) exists as HIP kernels only: the host branch would drop the 'sr' term silently")
                gt_int, gt_depth = gt_i * gt_rd, gt_d * gt_rd  # trainer.py:187-189
                pred_rd = r["image_lidar"][:, :, 0]
                pred_int = r["image_lidar"][:, :, 1] * gt_rd
                pred_depth = r["depth_lidar"] * gt_rd
                out["depth"] = (self.alpha_d * (pred_depth - gt_depth).abs()).sum()
                out["raydrop"] = (self.alpha_r * (pred_rd - gt_rd.clamp(self.smooth, 1 - self.smooth)) ** 2).sum()
                out["intensity"] = (self.alpha_i * (pred_int - gt_int) ** 2).sum()
                if self.use_chamfer:
                    d = batch["rays_d_lidar"]
                    d1, d2, _, _ = self._chamfer(d * pred_depth.unsqueeze(-1) / self.scale, d * gt_depth.unsqueeze(-1) / self.scale)
                    out["chamfer"] = (d1 + d2).mean() * 0.5
            if self.use_flow:
                fl = self.flow_loss(batch["time"])
                if fl is not None:
                    out["flow"] = fl
            if self.use_urf:
                eps = 0.02 * 0.1 ** min(self.global_step / self.iters, 1)
                out["los"] = urf_line_of_sight_loss(r["weights"], r["z_vals"], gt_d * gt_rd, eps)
        if "rays_o" in batch:
            gt_rgb = batch["gt_rgb"] if "gt_rgb" in batch else batch["images"][..., :3]
            r = self._render(batch["rays_o"], batch["rays_d"], batch["time"], perturb=True, num_steps=self.num_steps, bg_color=1)
            out["rgb"] = MseSumFn.apply(r["image"], gt_rgb, self.alpha_rgb) if r["image"].is_cuda else (self.alpha_rgb * (r["image"] - gt_rgb) ** 2).sum()
            if self.error_maps is not None and r["image"].is_cuda:
                self._for_error_map["camera"] = (r["image"].detach(), gt_rgb)
        total = sum(out.values())
        total = torch.nan_to_num(total, nan=0.0, posinf=1e5, neginf=1e5)  # trainer.py:545-546 (|inf| -> 1e5)
        return total, out

    def _patch_dims(self):
        p = self.patch_size_lidar
        if isinstance(p, int):
            return p, p
        return (p[0], p[0]) if len(p) == 1 else (p[0], p[1])

    def set_epoch(self, epoch, frames=None):
        """The per-epoch switch of trainer.py:1035-1062: with change_patch_size_lidar[0] > 1, every change_patch_size_epoch-th epoch
        samples LiDAR PATCHES (and regularises on them; the sampler draws from the error map when use_error_map), the others random
        pixels.  `frames`: the FrameSet whose sampler follows (default: the one given to attach_error_maps)."""
        frames = frames if frames is not None else self.error_maps
        if self.change_patch_size_lidar[0] > 1:
            patch = epoch % self.change_patch_size_epoch == 0
            self.pixel_sampler = "patch" if patch else "random"
            self.patch_size_lidar = tuple(self.change_patch_size_lidar) if patch else 1
            if frames is not None:
                frames.patch_size_lidar = self.patch_size_lidar
        if frames is not None:
            frames.use_error_map = self.use_error_map and self.pixel_sampler != "random"
        return self.pixel_sampler

    def attach_error_maps(self, frames):
        """`frames`: a FrameSet with enable_error_maps() called -- after every step the per-ray losses of the batch are written into
        its error_map / error_map_rgb (update_error_maps), as Trainer.train_step does whatever the sampler (trainer.py:552-630)."""
        if getattr(frames, "error_map", None) is None:
            frames.enable_error_maps()
        self.error_maps = frames

    def update_error_maps(self, batch):
        """trainer.py:552-630 on the device: per-ray loss -> min / max -> normalised to [1, 1000] -> EMA into the frame's coarse map
        at the rays' pixels (five small launches per modality, no host read)."""
        from nvsf import _hip
        frames, stash = self.error_maps, self._for_error_map
        self._for_error_map = {}
        if frames is None or "index" not in batch:
            return
        idx = int(batch["index"][0])
        for key, inds_key, emap, H, W in (("lidar", "rays_pano_inds", frames.error_map, frames.H_lidar, frames.W_lidar),
                                          ("camera", "rays_rgb_inds", frames.error_map_rgb, frames.H, frames.W)):
            if key not in stash or inds_key not in batch or emap is None:
                continue
            inds = batch[inds_key].detach().long().contiguous().view(-1)
            N, dev = inds.numel(), inds.device
            ray_loss = torch.empty(N, dtype=torch.float32, device=dev)
            stats = frames.error_map_stats(dev)
            if key == "lidar":
                img, dep, rd, gi, gd = (t.float().contiguous() for t in stash[key])
                _hip.call("nvsf_lidar_ray_losses", _hip.ptr(img), _hip.ptr(dep), _hip.ptr(rd), _hip.ptr(gi), _hip.ptr(gd), N, self.alpha_d, self.alpha_r,
                          self.alpha_i, self.smooth, _hip.ptr(ray_loss), _hip.ptr(stats))
            else:
                img, gt = (t.float().contiguous() for t in stash[key])
                _hip.call("nvsf_mse_rows", _hip.ptr(img), _hip.ptr(gt), N, img.shape[-1], self.alpha_rgb, _hip.ptr(ray_loss), _hip.ptr(stats))
            eH, eW = emap.shape[-2:]
            _hip.call("nvsf_error_map_update", _hip.ptr(ray_loss), _hip.ptr(inds), N, int(W), emap[idx].data_ptr(), int(eH), int(eW), float(eH / H),
                      float(eW / W), _hip.ptr(stats), _hip.ptr(frames.error_map_owner(dev, eH * eW)))

    CAMERA_KEYS = ("rays_o", "rays_d", "gt_rgb", "images", "rays_rgb_inds", "H", "W")

    def _backward(self, loss):
        overlap = loss.is_cuda and self.scatter_overlap
        local_sink = False
        tctx = self.train_ctx
        if overlap:
            from nvsf import field_ops
            tctx.overlap = True
            if tctx.sink is None:  # one process: the side-stream scatters accumulate straight into p.grad
                if self._sink is None:
                    self._sink = field_ops.LocalGradSink()
                tctx.sink, local_sink = self._sink, True
        try:
            self.scaler.scale(loss).backward()
        finally:
            if overlap:
                tctx.overlap = False
                if local_sink:
                    tctx.sink = None
        return overlap

    def sync(self):
        """The calling stream waits for the optimiser pass `step` left on the scatter stream (the last table's).  Readers inside this
        package that go through the encoders' fp16 copies wait by themselves (tinycudann._HalfCache.pending); call this before
        touching the fp32 parameters or the optimiser state directly (end_epoch / checkpoint_state / load_checkpoint do)."""
        if self._late_carry is not None:
            # a late-only overflow that has not reached the scaler yet is settled here (callers: end of epoch, checkpoint, load -- the
            # host is waiting anyway), so that it is neither lost with a checkpoint nor applied to a freshly loaded scaler
            ev_c, carry = self._late_carry
            self._late_carry = None
            ev_c.synchronize()
            if float(carry) != 0.0:
                self.scaler.update(carry)
        if self._pending is not None:
            ev, params = self._pending
            torch.cuda.current_stream().wait_event(ev)
            for p in params:
                cache = self._cache_of.get(p)
                if cache is not None and cache.pending is ev:
                    cache.pending = None
                owner = self._plane_owner.get(p)
                if owner is not None and owner.__dict__.get("_pending_update") is ev:
                    owner.__dict__["_pending_update"] = None
            self._pending = None

    def step(self, batch):
        defer = self.defer_last_table and self.scatter_overlap
        loss, parts, n_coll, late = self._run(batch, defer)
        if getattr(self.model, "cuda_ray", False):
            from nvsf.nerf.raymarching import raymarching
            if raymarching.march_status_pending():  # occupancy-grid training without a counter read-back: the marcher's failure flag
                raymarching.check_march_status(wait=True)  # is looked at before the optimiser, not one call later
        if not late:
            found = self.scaler.step(self.opt)
        else:
            # every gradient but the late tables' is final on this stream: the step's overflow decision is taken from those (for a
            # table fed by an fp32 input gradient that is sufficient: loss_scaler.py), their update here, the late tables' update
            # behind their scatter on the side stream.  A table fed through an fp16 hand-over (the static hash of the space-time
            # field: the density tail's gradient-split kernel casts a finite fp32 dX > 65504 to inf) is NOT covered by that argument,
            # so the late gradients are inspected as well -- on the side stream, behind their scatter: `found_late` = early OR late
            # is what the late tables' Adam pass skips on (no inf / nan can reach a table, its EMA shadow or its fp16 copy), and an
            # overflow seen ONLY there reaches the scaler's update one step later (`_late_carry`: the scale is halved after the next
            # step instead of this one; waiting for it here would put the scatter back on the critical path).
            from nvsf import field_ops
            late_set = {p for p, _ in late}
            early = [p for g in self.opt.param_groups for p in g["params"] if p.grad is not None and p not in late_set]
            if self.scaler.is_enabled():
                found = self.scaler.found_inf([p.grad for p in early])
                scale = self.scaler.scale_tensor()
                self.opt.grad_scale, self.opt.found_inf = scale, found
            else:
                found = None
            carry = None
            try:
                self.opt.step(params=early)
                main = torch.cuda.current_stream()
                side = field_ops.side_stream(late[0][0].device)
                side.wait_stream(main)  # found / scale exist; (the scatters themselves are already in front of us on `side`)
                with torch.cuda.stream(side):
                    for p in late_set:
                        p.grad.record_stream(side)  # zero_grad of the next step drops the tensor while this pass may still read it
                    if found is not None:
                        found.record_stream(side)
                        scale.record_stream(side)
                        late_only = self.scaler.found_inf([p.grad for p in late_set])
                        self.opt.found_inf = torch.maximum(found, late_only)
                        carry = torch.clamp_(late_only - found, min=0.0)  # 1 only when the early gradients were clean
                    self.opt.step(params=list(late_set))
                    ev = torch.cuda.Event()
                    ev.record(side)
            finally:
                if found is not None:
                    del self.opt.grad_scale, self.opt.found_inf
            self._pending = (ev, list(late_set))
            for p in late_set:
                cache = self._cache_of.get(p)
                if cache is not None:
                    cache.pending = ev
                owner = self._plane_owner.get(p)
                if owner is not None:  # K-planes: read as fp32 by their kernels, the module makes its first reader wait
                    owner.__dict__["_pending_update"] = ev
        if found is not None:
            prev, self._late_carry = self._late_carry, (None if not late or carry is None else (ev, carry))
            if prev is not None:  # a late-only overflow of the PREVIOUS step (its event is long past: this step's passes read that table)
                main = torch.cuda.current_stream()
                main.wait_event(prev[0])
                prev[1].record_stream(main)  # allocated on the side stream, consumed here
                found = torch.maximum(found, prev[1])
            if self._late_carry is not None:
                # this step's late check is still pending: its `found` may yet turn out to have been an overflow, so the step must not be
                # the one that completes a growth interval (the scale would double now and halve one step later)
                self.scaler.hold_growth()
            self.scaler.update(found)
        if self.error_maps is not None:
            self.update_error_maps(batch)
        self.sched.step()
        self.global_step += 1
        return loss, parts, n_coll

    def forward_backward(self, batch):
        """Everything of a step up to (not including) the optimiser: both renders, the losses, backward through the HIP operators,
        the gradient all-reduce when there is more than one rank; on return every parameter's `.grad` (still multiplied by the
        loss scale) is final on the current stream.  -> (loss, parts, number of collectives).  `step` = this + Adam (with the
        wait for the last table scatter moved behind the optimiser pass of everything else)."""
        return self._run(batch, False)[:3]

    def _run(self, batch, defer):
        if self.model.__dict__.get("_train_ctx") is not self.train_ctx:  # another step object was built on this model since
            for mod in self.model.modules():
                mod.__dict__["_train_ctx"] = self.train_ctx
        self.train_ctx.begin_step()  # a table is final after the LAST scatter it receives in this step (ray_chunks > 1: several)
        self._sink = None
        try:
            return self._forward_backward(batch, defer)
        finally:
            self.train_ctx.end_step()
            self._sink = None

    def _forward_backward(self, batch, defer=False):
        self.model.train()
        if self.buckets is not None:
            self.buckets.begin_step()
        else:
            self.opt.zero_grad(set_to_none=True)
        if self.split_backward and "rays_o_lidar" in batch and "rays_o" in batch:
            lidar = {k: v for k, v in batch.items() if k not in self.CAMERA_KEYS}
            camera = {k: v for k, v in batch.items() if k in self.CAMERA_KEYS or k == "time"}
            loss, parts, overlap = None, {}, False
            # Camera pass first: its table scatter is the longer one (2.85 against 2.07 ms at 4096 + 4096 rays x 768: camera samples
            # sit in cells of their own at the five finest levels) and, issued first, it runs on the side stream beside the whole
            # LiDAR pass (forward + backward, 2.7 ms); only the shorter LiDAR scatter is left exposed at the end of the step.
            for first, sub in ((True, camera), (False, lidar)):
                if self.buckets is not None:  # parameters both passes reach (the sigma MLP) are final only after the second
                    self.buckets.hold(first)
                part_loss, part = self.losses(sub)
                n_before = len(self._sink.side_scatters) if self._sink is not None else 0
                overlap = self._backward(part_loss) or overlap
                self._last_pass_tables = {p for p, _ in self._sink.side_scatters[n_before:]} if self._sink is not None else set()
                loss = part_loss.detach() if loss is None else loss + part_loss.detach()
                parts.update(part)
            if self.buckets is not None:
                self.buckets.release_held()
        else:
            loss, parts = self.losses(batch)
            overlap = self._backward(loss)
            self._last_pass_tables = set()  # one joint backward: nothing is left to overlap a deferred pass with -- all tables early
        # the all-reduce is linear: it runs on the scaled gradients (an inf / nan on one rank reaches every rank, so all of
        # them skip the step together); scaler.step unscales, checks and steps.  Buckets go out during backward (hooks); finish()
        # closes the rest, waits and averages.  The table scatters run on a side stream: their consumers wait here.
        late = []
        if self.buckets is not None:
            n_coll = self.buckets.finish()
        else:
            n_coll = 0
            if overlap:
                from nvsf import field_ops
                scatters = self._sink.side_scatters if self._sink is not None else []
                if defer and scatters:
                    # wait only for the tables scattered BEFORE the last backward pass (long finished); the last pass' tables stay late
                    last_pass = self._last_pass_tables
                    main = torch.cuda.current_stream()
                    for p, ev in scatters:
                        if p in last_pass:
                            late.append((p, ev))
                        else:
                            main.wait_event(ev)
                else:
                    field_ops.sync_side_streams()
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, n_coll, late

    def end_epoch(self):
        """What the reference does after the step loop of an epoch (trainer.py:1420-1421): one EMA update."""
        self.sync()
        if self.ema is not None:
            self.ema.update()

    def checkpoint_state(self, epoch=0, stats=None, full=True, reference_layout=False):
        """The dict Trainer.save_checkpoint writes with torch.save (nvsf/nerf/utils.py:622-648): epoch, global_step, stats,
        model and -- for a `full` checkpoint -- optimizer, lr_scheduler, scaler, ema.  `model` always speaks the reference's
        schema (names and shapes).  `optimizer` and `ema` are lists by parameter position: written in this package's layout (one
        entry per `planes_cl`) unless `reference_layout`, which expands the optimiser entry to one state per plane so that a
        reference Trainer resumes from it (checkpoint_compat); the `ema` entry cannot be written for the reference -- torch_ema
        wants shadows for the three unused modules this model does not have -- and the reference then starts a fresh average,
        as it does for any checkpoint whose ema fails to load (utils.py:728-747).  load_checkpoint reads both layouts."""
        self.sync()
        state = {"epoch": epoch, "global_step": self.global_step, "stats": stats if stats is not None else {}}
        if full:
            state["optimizer"] = self.opt.state_dict()
            if reference_layout:
                from nvsf.nerf import checkpoint_compat as compat
                state["optimizer"] = compat.optimizer_state_in_reference_layout(state["optimizer"], self.model, self.opt)
            state["lr_scheduler"] = self.sched.state_dict()
            state["scaler"] = self.scaler.state_dict()
            if self.ema is not None:
                state["ema"] = self.ema.state_dict()
        state["model"] = self.model.state_dict()
        return state

    def save_checkpoint(self, path, epoch=0, stats=None, full=True, reference_layout=False):
        torch.save(self.checkpoint_state(epoch, stats, full, reference_layout), path)

    def load_checkpoint(self, checkpoint, model_only=False):
        """Trainer.load_checkpoint (nvsf/nerf/utils.py:682-747): a bare state_dict loads strictly; a checkpoint dict loads its
        model with strict=False (the reference's files carry three unused sub-modules), then global_step / optimizer /
        lr_scheduler / scaler / ema where present.  checkpoint: path or the loaded dict.  Returns (missing_keys, unexpected_keys,
        epoch); `self.failed_to_load` lists the components whose state could not be restored (each also raises a warning, as
        the reference logs "[WARN] Failed to load ..." and trains on, utils.py:728-747)."""
        self.sync()
        if not isinstance(checkpoint, dict):
            checkpoint = torch.load(checkpoint, map_location=next(self.model.parameters()).device)
        if "model" not in checkpoint:
            self.model.load_state_dict(checkpoint)
            return [], [], None
        missing, unexpected = self.model.load_state_dict(checkpoint["model"], strict=False)
        self.failed_to_load = []

        from nvsf.nerf import checkpoint_compat as compat

        def translate(key, state):
            """`ema` / `optimizer` entries are ordered by parameter POSITION: a reference checkpoint counts three unused modules and
            24 tensors per Planes4D where this model has one (checkpoint_compat)."""
            if key == "ema":
                state = dict(state, shadow_params=compat.shadows_for_model(state["shadow_params"], self.model))
                if state.get("collected_params") is not None:
                    state["collected_params"] = compat.shadows_for_model(state["collected_params"], self.model)
            elif key == "optimizer":
                state = compat.optimizer_state_for_model(state, self.model, self.opt)
            return state

        def restore(key, obj):
            if key in checkpoint and obj is not None:
                try:
                    obj.load_state_dict(translate(key, checkpoint[key]))
                except Exception as e:  # noqa: BLE001 -- reference behaviour: warn and continue
                    self.failed_to_load.append(key)
                    warnings.warn(f"[WARN] Failed to load {key} state from the checkpoint ({type(e).__name__}: {e}); "
                                  f"continuing with a fresh {key}")
                    if key == "ema":
                        # "fresh" = the average starts again FROM THE WEIGHTS JUST LOADED: the shadows still hold the clones taken
                        # at construction (random init), which evaluate_frames(ema=...) would otherwise copy over the checkpoint
                        obj.reset_to_parameters()
        restore("ema", self.ema)  # before the model_only return, as the reference does (utils.py:715-719): evaluation runs under the EMA weights
        if model_only:
            return list(missing), list(unexpected), checkpoint.get("epoch")
        if "global_step" in checkpoint:
            self.global_step = checkpoint["global_step"]
        for key, obj in (("optimizer", self.opt), ("lr_scheduler", self.sched), ("scaler", self.scaler)):
            restore(key, obj)
        return list(missing), list(unexpected), checkpoint.get("epoch")


def psnr(pred, truth):
    """-10 log10(mean((p - t)^2) + 1e-8), images in [0, 1]."""
    p, t = (np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64) for a in (pred, truth))
    return float(-10 * np.log10(np.mean((p - t) ** 2) + 1e-8))


def depth_rmse(pred, truth, scale, min_depth=1e-6, max_depth=80.0):
    """RMSE in metres: both ranges divided by the scene scale and clamped to [1e-6, 80] m."""
    p, t = (np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64) / scale for a in (pred, truth))
    p, t = np.clip(p, min_depth, max_depth), np.clip(t, min_depth, max_depth)
    return float(np.sqrt(np.mean((t - p) ** 2)))


def fscore(dist1, dist2, threshold=0.001):
    """F-score of two point clouds from their SQUARED nearest-neighbour distances [B, n] / [B, m] (nvsf/lib/error_matrices.py:12-26):
    2 p r / (p + r) with p, r the fractions below `threshold`, 0 where both are 0.  Returns fscore, precision, recall ([B])."""
    p1 = (dist1 < threshold).float().mean(dim=1)
    p2 = (dist2 < threshold).float().mean(dim=1)
    f = 2 * p1 * p2 / (p1 + p2)
    return torch.nan_to_num(f, nan=0.0), p1, p2


def pano_to_lidar(pano, intrinsics, intrinsics_hoz=(180.0, 360.0)):
    """Range image [H, W] -> points [n, 3] in the LiDAR frame, on the device of `pano` (nvsf/lib/convert.py:221-291): pixel (row j,
    column i) looks along azimuth beta = -(i - W / 2) / W * fov_hoz and elevation alpha = fov_up - j / H * fov (degrees; `intrinsics` =
    (fov_up, fov), `intrinsics_hoz` = (fov_hoz_up, fov_hoz)).  Pixels of range exactly 0 (dropped rays, ~30 % of a KITTI-360 frame) give
    no point -- the reference filters them with `np.where(pano != 0.0)` (convert.py:262-266) before the chamfer distance sees the
    cloud -- so n <= H * W, in row-major pixel order (one device->host read for the count; this is evaluation code)."""
    H, W = pano.shape
    fov_up, fov = (float(v) for v in intrinsics)
    fov_hoz = float(intrinsics_hoz[1])
    i = torch.arange(W, dtype=torch.float32, device=pano.device)[None, :]
    j = torch.arange(H, dtype=torch.float32, device=pano.device)[:, None]
    beta = -(i - W / 2) / W * fov_hoz / 180 * np.pi
    alpha = (fov_up - j / H * fov) / 180 * np.pi
    dirs = torch.stack([torch.cos(alpha) * torch.cos(beta), torch.cos(alpha) * torch.sin(beta), torch.sin(alpha).expand(H, W)], -1)
    pano = pano.float()
    return (dirs * pano[..., None])[pano != 0.0]


class PointsMeter:
    """Chamfer distance and F-score of whole rendered range images against the measured ones -- the reference's PointsMeter
    (nvsf/lib/error_matrices.py:299-356: ranges divided by the scene scale, pano_to_lidar with the sensor's intrinsics,
    CD = mean d1 + mean d2 over squared distances, F-score at 0.05) with the clouds built on the device and the nearest neighbours
    from the HIP chamfer kernel (csrc/chamfer.hip), which also serves the training loss."""

    def __init__(self, scale, intrinsics, intrinsics_hoz=(180.0, 360.0), threshold=0.05):
        self.scale, self.intrinsics, self.intrinsics_hoz, self.threshold = float(scale), intrinsics, intrinsics_hoz, threshold
        self.clear()

    def clear(self):
        self.V, self.N = [], 0

    def update(self, preds, truths):
        """preds, truths: [1, H, W] (or [H, W]) range images in scene units."""
        from nvsf.nerf.chamfer3D.dist_chamfer_3D import chamfer_3DDist
        p, t = (a.reshape(a.shape[-2], a.shape[-1]).float() / self.scale for a in (preds, truths))
        with torch.no_grad():
            cp, ct = pano_to_lidar(p, self.intrinsics, self.intrinsics_hoz), pano_to_lidar(t, self.intrinsics, self.intrinsics_hoz)
            if cp.shape[0] == 0 or ct.shape[0] == 0:
                # a cloud without points (e.g. every predicted pixel gated off by the ray-drop mask): the reference's means over
                # an empty distance array are NaN and its F-score 0 / 0 -> 0 (error_matrices.py:12-26, 322-335)
                cd, f = float("nan"), 0.0
            else:
                d1, d2, _, _ = chamfer_3DDist()(cp[None], ct[None])
                cd = d1.mean() + d2.mean()
                f = fscore(d1, d2, self.threshold)[0][0]
        self.V.append([float(cd), float(f)])
        self.N += 1

    def measure(self):
        assert self.N == len(self.V), "prediction and gt should should be equal"
        return np.array(self.V).mean(0)

    def report(self):
        cd, f = self.measure()
        return f"Points_error(CD, F-score) = {[round(float(cd), 3), round(float(f), 3)]}"


def eval_step(model, data, num_steps, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, alpha_rgb=1.0, raydrop_thres=0.5, max_ray_batch=4096,
              split_rays=True, **render_kwargs):
    """Whole-frame evaluation of one frame, the reference's Trainer.eval_step (nvsf/nerf/trainer.py:658-815) without its optional
    U-Net ray-drop refinement: `data` = FrameSet(..., training=False).collate([i]) -- every pixel of the range image and of the camera
    image.  Both modalities go through the staged render, a frame's rays split over the ranks (frame_shard.render_sharded); the
    predicted ray-drop mask (`pred_raydrop > raydrop_thres`, :726) gates predicted intensity and range, the ground truth is gated by
    its own mask (:695-696); loss = the reference's mean-reduced L1 range + MSE ray-drop + MSE intensity + MSE RGB (:733-737, :795-796;
    criteria as main_nvsf.py:205-221).  Returns a dict of [B, H, W(, C)] predictions / ground truths and `loss`."""
    from nvsf import frame_shard
    if split_rays:
        render = lambda *a, **k: frame_shard.render_sharded(model, *a, **k)
    else:  # this rank renders the whole frame by itself (evaluate_frames(shard="frames"))
        render = lambda o, d, t, **k: model.render(o, d, t, staged=True, **k)
    out = {}
    loss = torch.zeros((), device=data["rays_o_lidar"].device)
    with torch.no_grad():
        gl = data["images_lidar"]  # [B, H, W, 3] = raydrop, intensity, range
        B, Hl, Wl, _ = gl.shape
        gt_raydrop = gl[..., 0]
        gt_intensity, gt_depth = gl[..., 1] * gt_raydrop, gl[..., 2] * gt_raydrop
        o = render(data["rays_o_lidar"], data["rays_d_lidar"], data["time"], cal_lidar_color=True, num_steps=num_steps,
                   max_ray_batch=max_ray_batch, **render_kwargs)
        img = o["image_lidar"].reshape(B, Hl, Wl, 2)
        pred_raydrop, pred_intensity, pred_depth = img[..., 0], img[..., 1], o["depth_lidar"].reshape(B, Hl, Wl)
        mask = (pred_raydrop > raydrop_thres).to(pred_depth.dtype)
        pred_intensity, pred_depth = pred_intensity * mask, pred_depth * mask
        loss = loss + alpha_d * (pred_depth - gt_depth).abs().mean() + alpha_r * ((pred_raydrop - gt_raydrop) ** 2).mean() \
            + alpha_i * ((pred_intensity - gt_intensity) ** 2).mean()
        out.update(pred_raydrop=pred_raydrop, pred_intensity=pred_intensity, pred_depth=pred_depth, gt_raydrop=gt_raydrop,
                   gt_intensity=gt_intensity, gt_depth=gt_depth)
        gi = data["images"]  # [B, H, W, 3 or 4]
        _, H, W, C = gi.shape
        gt_rgb = gi[..., :3] * gi[..., 3:] + (1 - gi[..., 3:]) if C == 4 else gi  # fixed white background (:774-781)
        c = render(data["rays_o"], data["rays_d"], data["time"], num_steps=num_steps, max_ray_batch=max_ray_batch, bg_color=1, **render_kwargs)
        pred_rgb = c["image"].reshape(B, H, W, 3)
        loss = loss + alpha_rgb * ((pred_rgb - gt_rgb) ** 2).mean()
        out.update(pred_rgb=pred_rgb, pred_rgb_depth=c["depth"].reshape(B, H, W), gt_rgb=gt_rgb, loss=loss)
    return out


def evaluate_frames(model, frames, num_steps, indices=None, ema=None, shard="rays", **eval_kwargs):
    """The metric half of the reference's evaluate_one_epoch (trainer.py:1458-1560) over a FrameSet opened with training=False:
    per frame eval_step, then the two quality metrics of the headline benchmark -- PSNR of the image (error_matrices.py:48-57), range
    RMSE in metres (:263-285) -- and chamfer distance / F-score of the range image's point cloud (PointsMeter, :299-356, on
    csrc/chamfer.hip); means over the frames.  `ema`: the step's ExponentialMovingAverage (RenderTrainStep.ema) -- the reference
    evaluates under `ema.store(); ema.copy_to()` and `restore()`s afterwards (trainer.py:1475-1477, 1843-1844), so metrics are
    those of the averaged weights once EMA is on (its default).
    Across ranks, `shard`:
      "rays"   every frame's rays are split over the ranks and the renders all-gathered (frame_shard.render_sharded): every rank
               computes the same statistics from the same full frames, no statistics collective is needed;
      "frames" the reference's scheme (trainer.py:1495-1524): rank r evaluates frames r, r + W, ... on its own and the per-rank SUMS
               of loss and metrics go through ONE all-reduce (frame_shard.allreduce_sums = the `dist.all_reduce(loss)` of
               trainer.py:1508, widened to the metrics); no per-pixel data crosses xGMI.
    Every rank returns the same numbers.  The reference's other meters (ray-drop accuracy / F1, intensity MAE, SSIM, LPIPS: SURVEY 2
    #18) are outside this package's scope; nvsf/nerf/meters_extra.py restates two of them for users who want them beside these."""
    from nvsf import frame_shard
    if shard not in ("rays", "frames"):
        raise ValueError("shard: 'rays' or 'frames'")
    rank, ws = frame_shard.world()
    was_training = model.training
    model.eval()
    if ema is not None:
        ema.store()
        ema.copy_to()
    try:
        points = PointsMeter(frames.scale, frames.intrinsics_lidar, frames.intrinsics_hoz_lidar)
        ps, rm, ls = [], [], []
        todo = list(range(len(frames)) if indices is None else indices)
        if shard == "frames" and ws > 1:
            todo = todo[rank::ws]
        for i in todo:
            e = eval_step(model, frames.collate([int(i)]), num_steps, split_rays=(shard == "rays"), **eval_kwargs)
            ps.append(psnr(e["pred_rgb"], e["gt_rgb"]))
            rm.append(depth_rmse(e["pred_depth"], e["gt_depth"], frames.scale))
            points.update(e["pred_depth"], e["gt_depth"])
            ls.append(float(e["loss"]))
    finally:
        if ema is not None:
            ema.restore()
        model.train(was_training)
    cdf = np.array(points.V, dtype=np.float64).reshape(-1, 2).sum(0)
    sums = [float(np.sum(ls)), float(np.sum(ps)), float(np.sum(rm)), float(cdf[0]), float(cdf[1]), float(len(ps))]
    if shard == "frames":
        sums = frame_shard.allreduce_sums(sums, device=next(model.parameters()).device)
    n = max(sums[5], 1.0)
    return {"loss": sums[0] / n, "psnr": sums[1] / n, "depth_rmse_m": sums[2] / n, "chamfer_distance": sums[3] / n, "f_score": sums[4] / n,
            "frames": int(sums[5])}
