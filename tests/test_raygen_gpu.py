"""GPU parity of device ray generation (row f1) against fixtures produced by the reference's own
dataset_utils.get_lidar_rays / get_rays on CPU (tests/golden/golden_rays.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lidar_and_camera_rays_match_reference(dev):
    from nvsf.nerf.dataset import dataset_utils as du
    g = np.load(os.path.join(GOLD, "rays.npz"))
    t = lambda a: torch.from_numpy(a).to(dev)
    pose = t(g["lidar_pose"])[None]
    r = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, inds=t(g["lidar_inds"])[None])
    np.testing.assert_allclose(r["rays_d"][0].cpu().numpy(), g["lidar_rays_d"], atol=2e-6, rtol=0)
    assert np.array_equal(r["rays_o"][0].cpu().numpy(), g["lidar_rays_o"])
    full = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, N=-1)
    assert full["rays_d"].shape == (1, int(g["lidar_full_n"]), 3) and full["inds"].shape == (1, 66 * 1030)
    np.testing.assert_allclose(full["rays_d"][0].cpu().numpy()[g["lidar_full_sel"]], g["lidar_full_rays_d"], atol=2e-6, rtol=0)
    rp = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, inds=t(g["lidar_patch_inds"])[None])
    np.testing.assert_allclose(rp["rays_d"][0].cpu().numpy(), g["lidar_patch_rays_d"], atol=2e-6, rtol=0)
    rc = du.get_rays(t(g["cam_pose"])[None], torch.from_numpy(g["cam_K"]), 376, 1408, inds=t(g["cam_inds"])[None])
    np.testing.assert_allclose(rc["rays_d"][0].cpu().numpy(), g["cam_rays_d"], atol=2e-6, rtol=0)
    assert np.array_equal(rc["rays_o"][0].cpu().numpy(), g["cam_rays_o"])
    np.testing.assert_allclose(np.linalg.norm(rc["rays_d"][0].cpu().numpy(), axis=1), 1.0, atol=1e-6)


def test_index_sampling_modes(dev):
    from nvsf.nerf.dataset import dataset_utils as du
    torch.manual_seed(0)
    pose = torch.eye(4, device=dev)[None]
    r = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, N=2048)
    assert r["inds"].shape == (1, 2048) and int(r["inds"].max()) < 66 * 1030 and r["rays_d"].shape == (1, 2048, 3)
    r = du.get_lidar_rays(pose, [2.0, 26.9], [180.0, 360.0], 66, 1030, N=1024, patch_size=[2, 8])
    inds = r["inds"][0].view(-1, 16)
    rows, cols = inds // 1030, inds % 1030
    assert (rows[:, 8:] == rows[:, :8] + 1).all() and (cols[:, :8] == cols[:, :1] + torch.arange(8, device=dev)).all()
    err = torch.ones(1, 16, 64, device=dev)
    err[0, :8] = 0  # error map concentrated on the lower half of the frame
    r = du.get_rays(pose, torch.tensor([[500.0, 0, 700], [0, 500, 190], [0, 0, 1]]), 376, 1408, N=512, error_map=err, use_error_map=True)
    assert ((r["inds"][0] // 1408) >= 376 // 2).all()
