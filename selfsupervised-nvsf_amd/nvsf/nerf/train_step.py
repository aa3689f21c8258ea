"""Thin multimodal training step + the two quality metrics of the headline benchmark.

Only what sits directly around the hot path, restated from the reference:
  * losses of Trainer.train_step (nvsf/nerf/trainer.py:193-219, 276-294, 491-503 with the default CLI weights of
    nvsf/scripts/main_nvsf.py:84-97): L1 range, MSE ray-drop against label-smoothed targets (smooth 0.2, weight
    0.01), MSE intensity on returned rays (weight 0.1), MSE RGB, optional URF line-of-sight loss on
    (weights, z_vals);
  * optimiser: Adam(betas 0.9/0.99, eps 1e-15) on `model.get_params(lr)` and the 0.1^(iter/iters) decay
    (main_nvsf.py:350-362), under the GradScaler of the reference's fp16 run (trainer.py:119, 1332-1334);
  * PSNR (nvsf/lib/error_matrices.py:48-57) and depth RMSE in metres (error_matrices.py:263-285).
Across GPUs the step is frame-sharded: every rank renders its own frame, then ONE bucketed gradient all-reduce
(nvsf/frame_shard.py).  Checkpoints use the reference's dict (utils.py:622-747).  The reference's Trainer (logging, EMA, UNet
refinement, error maps, chamfer and flow losses) is out of scope.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

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


class RenderTrainStep:
    def __init__(self, model, lr=1e-2, iters=30000, num_steps=768, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, alpha_rgb=1.0,
                 smooth_factor=0.2, use_urf_loss=False, bucket_bytes=64 << 20, fp16=True):
        self.model = model
        # loss scaling of the reference's mixed-precision run (trainer.py:119, 1332-1334: GradScaler(enabled=fp16) -> scale(loss)
        # .backward() -> step -> update; `-L` / `--fp16` in main_nvsf.py:17,43,159).  The encoders hand fp16 features to the
        # MLPs, so the gradients that travel back between them are fp16 tensors: unscaled, those of a mean-over-rays loss sit
        # below the fp16 subnormal range and the hash tables receive zeros.
        self.scaler = torch.amp.GradScaler("cuda", enabled=bool(fp16) and torch.cuda.is_available())
        if fp16 and hasattr(model, "flow_net"):
            model.flow_net.flow_mlp_mode = "fused"  # the flow MLP as autocast runs it: fp16 MFMA kernels (flow_field.FlowMlpFn)
        # Adam as one HIP pass per parameter tensor with the scaler's overflow flag consumed on the device (nvsf/nerf/adam.py).
        # torch's own fused=True variant was tried first: under this eps = 1e-15 / GradScaler setting it trains measurably worse
        # (tests/test_train_step_gpu.py: loss 0.289 -> 0.258 after 120 steps against 0.227).  NVSF_ADAM=torch: the default
        # multi-tensor implementation.
        import os
        on_gpu = torch.cuda.is_available() and next(model.parameters()).is_cuda
        if on_gpu and os.environ.get("NVSF_ADAM", "hip") == "hip":
            from nvsf.nerf.adam import FusedAdam
            self.opt = FusedAdam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        else:
            self.opt = torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda it: 0.1 ** min(it / iters, 1))
        self.iters, self.num_steps = iters, num_steps
        self.alpha_d, self.alpha_r, self.alpha_i, self.alpha_rgb = alpha_d, alpha_r, alpha_i, alpha_rgb
        self.smooth, self.use_urf, self.bucket_bytes = smooth_factor, use_urf_loss, bucket_bytes
        self.global_step = 0

    def losses(self, batch):
        """batch: rays_o_lidar, rays_d_lidar [1,N,3], gt_depth, gt_intensity, gt_raydrop [1,N]; rays_o, rays_d [1,N,3],
        gt_rgb [1,N,3]; time [1,1]."""
        m, out = self.model, {}
        total = 0.0
        if "rays_o_lidar" in batch:
            r = m.render(batch["rays_o_lidar"], batch["rays_d_lidar"], batch["time"], cal_lidar_color=True, perturb=True,
                         num_steps=self.num_steps)
            gt_rd = batch["gt_raydrop"]
            pred_rd = r["image_lidar"][:, :, 0]
            pred_int = r["image_lidar"][:, :, 1] * gt_rd
            pred_depth = r["depth_lidar"] * gt_rd
            out["depth"] = self.alpha_d * F.l1_loss(pred_depth, batch["gt_depth"])
            out["raydrop"] = self.alpha_r * F.mse_loss(pred_rd, gt_rd.clamp(self.smooth, 1 - self.smooth))
            out["intensity"] = self.alpha_i * F.mse_loss(pred_int, batch["gt_intensity"])
            if self.use_urf:
                eps = 0.02 * 0.1 ** min(self.global_step / self.iters, 1)
                out["los"] = urf_line_of_sight_loss(r["weights"], r["z_vals"], batch["gt_depth"], eps)
        if "rays_o" in batch:
            r = m.render(batch["rays_o"], batch["rays_d"], batch["time"], perturb=True, num_steps=self.num_steps, bg_color=1)
            out["rgb"] = self.alpha_rgb * F.mse_loss(r["image"], batch["gt_rgb"])
        for v in out.values():
            total = total + v
        return total, out

    def step(self, batch):
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        loss, parts = self.losses(batch)
        self.scaler.scale(loss).backward()
        # the all-reduce is linear: it runs on the scaled gradients (an inf / nan on one rank reaches every rank, so all of
        # them skip the step together); scaler.step unscales, checks and steps
        n_coll = frame_shard.allreduce_gradients([p for g in self.opt.param_groups for p in g["params"]], self.bucket_bytes)
        self.scaler.step(self.opt)
        self.scaler.update()
        self.sched.step()
        self.global_step += 1
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, n_coll

    def checkpoint_state(self, epoch=0, stats=None, full=True):
        """The dict Trainer.save_checkpoint writes with torch.save (nvsf/nerf/utils.py:622-648): epoch, global_step, stats,
        model and -- for a `full` checkpoint -- optimizer, lr_scheduler, scaler (no EMA here)."""
        state = {"epoch": epoch, "global_step": self.global_step, "stats": stats if stats is not None else {}}
        if full:
            state["optimizer"] = self.opt.state_dict()
            state["lr_scheduler"] = self.sched.state_dict()
            state["scaler"] = self.scaler.state_dict()
        state["model"] = self.model.state_dict()
        return state

    def save_checkpoint(self, path, epoch=0, stats=None, full=True):
        torch.save(self.checkpoint_state(epoch, stats, full), path)

    def load_checkpoint(self, checkpoint, model_only=False):
        """Trainer.load_checkpoint (nvsf/nerf/utils.py:682-747): a bare state_dict loads strictly; a checkpoint dict loads its
        model with strict=False (the reference's files carry three unused sub-modules), then global_step / optimizer /
        lr_scheduler / scaler where present.  checkpoint: path or the loaded dict.  Returns (missing_keys, unexpected_keys,
        epoch)."""
        if not isinstance(checkpoint, dict):
            checkpoint = torch.load(checkpoint, map_location=next(self.model.parameters()).device)
        if "model" not in checkpoint:
            self.model.load_state_dict(checkpoint)
            return [], [], None
        missing, unexpected = self.model.load_state_dict(checkpoint["model"], strict=False)
        if model_only:
            return list(missing), list(unexpected), checkpoint.get("epoch")
        if "global_step" in checkpoint:
            self.global_step = checkpoint["global_step"]
        # the reference swallows a failure to restore these three (optimiser groups of another model layout) and trains on
        for key, obj in (("optimizer", self.opt), ("lr_scheduler", self.sched), ("scaler", self.scaler)):
            if key in checkpoint:
                try:
                    obj.load_state_dict(checkpoint[key])
                except Exception:  # noqa: BLE001 -- reference behaviour: warn and continue
                    pass
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
