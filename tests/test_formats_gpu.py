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
