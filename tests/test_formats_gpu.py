"""FrameSet.collate / train_batch on a synthetic dataset written in the reference's on-disk formats (row f4): the rays are
those of dataset_utils at the sampled indices, the ground truth is gathered at the same indices, and the batch drives one
training step."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_frameset_collate_and_train_step(tmp_path):
    from test_formats_cpu import make_dataset
    from nvsf.nerf.dataset import formats as F, dataset_utils as DU
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    dev = torch.device("cuda:0")
    seq, frames, images, pcs, K = make_dataset(str(tmp_path), n_frames=3, H=24, W=32, Hl=16, Wl=64)
    scale = 0.0108
    fs = F.FrameSet(str(tmp_path), seq, "train", scale, num_rays=128, num_rays_lidar=96, device=dev)
    assert len(fs) == 3
    torch.manual_seed(0)
    c = fs.collate([1])
    assert c["rays_o"].shape == (1, 128, 3) and c["rays_o_lidar"].shape == (1, 96, 3) and float(c["time"]) == pytest.approx(1 / 63)
    ref = DU.get_rays(fs.poses[1:2], fs.intrinsics, 24, 32, inds=c["rays_rgb_inds"])
    assert torch.equal(ref["rays_d"], c["rays_d"]) and torch.equal(ref["rays_o"], c["rays_o"])
    refl = DU.get_lidar_rays(fs.poses_lidar[1:2], fs.intrinsics_lidar, fs.intrinsics_hoz_lidar, 16, 64, inds=c["rays_pano_inds"])
    assert torch.equal(refl["rays_d"], c["rays_d_lidar"])
    img = torch.from_numpy(images[1].astype(np.float32) / 255).to(dev).reshape(-1, 3)
    assert torch.equal(c["images"][0], img[c["rays_rgb_inds"][0]])
    gt = torch.from_numpy(F.range_image_ground_truth(pcs[1], scale).astype(np.float32)).to(dev).reshape(-1, 3)
    assert torch.equal(c["images_lidar"][0], gt[c["rays_pano_inds"][0]])
    # evaluation mode: whole frames
    fe = F.FrameSet(str(tmp_path), seq, "train", scale, device=dev, training=False)
    ce = fe.collate([0])
    assert ce["rays_o"].shape == (1, 24 * 32, 3) and ce["images_lidar"].shape == (1, 16, 64, 3)
    # the batch drives a training step
    m = NeRFNetworkStatic(bound=2.0, min_near=0.01, min_near_lidar=0.01, lidar_max_depth=0.9).to(dev)
    step = RenderTrainStep(m, num_steps=32)
    loss, parts, _ = step.step(fs.train_batch([2]))
    assert torch.isfinite(loss) and set(parts) == {"depth", "raydrop", "intensity", "chamfer", "rgb"}


def test_eval_step_and_evaluate_frames(tmp_path):
    """Whole-frame evaluation (train_step.eval_step / evaluate_frames = Trainer.eval_step, trainer.py:658-815, and the metric half
    of evaluate_one_epoch, :1458-1560): shapes, the ray-drop gate on predictions and ground truth, the loss recomputed from the
    returned images, staged chunks smaller than a frame, and the frame means of PSNR / range RMSE / CD / F-score against the
    stand-alone meters."""
    from test_formats_cpu import make_dataset
    from nvsf.nerf.dataset import formats as F
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import eval_step, evaluate_frames, psnr, depth_rmse, PointsMeter
    dev = torch.device("cuda:0")
    seq, frames, images, pcs, K = make_dataset(str(tmp_path), n_frames=2, H=24, W=32, Hl=16, Wl=64)
    scale = 0.0108
    fe = F.FrameSet(str(tmp_path), seq, "train", scale, device=dev, training=False)
    torch.manual_seed(1)
    m = NeRFNetworkStatic(bound=2.0, min_near=0.01, min_near_lidar=0.01, lidar_max_depth=0.9).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1 and p.numel() > 10000:
                p.normal_(0, 0.3)  # tables away from zero so that the heads' outputs vary over the frame
    data = fe.collate([1])
    # a ray-drop threshold inside the range this (untrained) field predicts, so that the gate keeps some pixels and drops others
    thres = float(eval_step(m, data, 48)["pred_raydrop"].median())
    e = eval_step(m, data, 48, max_ray_batch=300, raydrop_thres=thres)  # 768 camera / 1024 LiDAR rays: several chunks, the last one ragged
    assert e["pred_rgb"].shape == (1, 24, 32, 3) and e["pred_depth"].shape == (1, 16, 64) and e["pred_raydrop"].shape == (1, 16, 64)
    gate = e["pred_raydrop"] > thres
    assert gate.any() and (~gate).any()
    assert float(e["pred_depth"][~gate].abs().max() if (~gate).any() else 0.0) == 0.0
    assert float(e["pred_intensity"][~gate].abs().max() if (~gate).any() else 0.0) == 0.0
    gl = data["images_lidar"]
    assert torch.equal(e["gt_depth"], gl[..., 2] * gl[..., 0]) and torch.equal(e["gt_intensity"], gl[..., 1] * gl[..., 0])
    want = (e["pred_depth"] - e["gt_depth"]).abs().mean() + 0.01 * ((e["pred_raydrop"] - e["gt_raydrop"]) ** 2).mean() \
        + 0.1 * ((e["pred_intensity"] - e["gt_intensity"]) ** 2).mean() + ((e["pred_rgb"] - e["gt_rgb"]) ** 2).mean()
    assert float(e["loss"]) == pytest.approx(float(want), rel=1e-6)
    whole = eval_step(m, data, 48, max_ray_batch=1 << 20, raydrop_thres=thres)  # one chunk: the same frame bit for bit
    for k in ("pred_rgb", "pred_depth", "pred_raydrop", "pred_intensity"):
        assert torch.equal(e[k], whole[k]), k
    res = evaluate_frames(m, fe, 48, raydrop_thres=thres)
    res_frames = evaluate_frames(m, fe, 48, raydrop_thres=thres, shard="frames")  # the reference's per-rank-frames + all-reduced sums scheme
    assert res_frames.keys() == res.keys() and all(res_frames[k] == pytest.approx(res[k], rel=1e-6, nan_ok=True) for k in res)
    assert res["frames"] == 2 and all(np.isfinite(v) for v in res.values())
    assert set(res) == {"loss", "psnr", "depth_rmse_m", "chamfer_distance", "f_score", "frames"}
    pm = PointsMeter(scale, fe.intrinsics_lidar, fe.intrinsics_hoz_lidar)
    ps, rm = [], []
    for i in range(2):
        ei = eval_step(m, fe.collate([i]), 48, raydrop_thres=thres)
        ps.append(psnr(ei["pred_rgb"], ei["gt_rgb"]))
        rm.append(depth_rmse(ei["pred_depth"], ei["gt_depth"], scale))
        pm.update(ei["pred_depth"], ei["gt_depth"])
    assert res["psnr"] == pytest.approx(np.mean(ps)) and res["depth_rmse_m"] == pytest.approx(np.mean(rm))
    assert res["chamfer_distance"] == pytest.approx(pm.measure()[0]) and res["f_score"] == pytest.approx(pm.measure()[1])
    assert m.training  # evaluate_frames restores the mode it found
    # under an EMA (the reference's evaluate_one_epoch: ema.store(); ema.copy_to() ... ema.restore(), trainer.py:1475-1477, 1843-1844):
    # the metrics are those of the averaged weights, and the model gets its own weights back
    from nvsf.nerf.ema import ExponentialMovingAverage
    ema = ExponentialMovingAverage(m.parameters(), decay=0.95)
    before = [p.detach().clone() for p in m.parameters()]
    # a frame whose every predicted pixel is gated off has no predicted cloud: CD = NaN, F-score 0, as the reference's means over an
    # empty array give (convert.py:262-266 + error_matrices.py:322-335) -- not a rejected kernel launch
    pm0 = PointsMeter(scale, fe.intrinsics_lidar, fe.intrinsics_hoz_lidar)
    pm0.update(torch.zeros_like(ei["pred_depth"]), ei["gt_depth"])
    assert np.isnan(pm0.measure()[0]) and pm0.measure()[1] == 0.0
    assert evaluate_frames(m, fe, 48, ema=ema, raydrop_thres=thres)["psnr"] == pytest.approx(res["psnr"])  # shadows == weights so far
    with torch.no_grad():
        for sh in ema.shadow_params:
            if sh.numel():
                sh.mul_(0.5)
    shifted = evaluate_frames(m, fe, 48, ema=ema, raydrop_thres=thres)
    assert shifted["psnr"] != pytest.approx(res["psnr"]) and shifted["loss"] != pytest.approx(res["loss"])
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))
    assert evaluate_frames(m, fe, 48, raydrop_thres=thres)["psnr"] == pytest.approx(res["psnr"])  # fp16 copies of tables / weights follow the restore
