"""GPU: the two pieces of the reference Trainer that its shipped configuration switches on beside the default losses
(configs/kitti360_1908.txt:13-14 `grad_loss`, `use_error_map`; VERDICT r4 "missing" 3): the structural regularisation on LiDAR patches
(nvsf_lidar_grad_loss_fwd / _bwd, trainer.py:296-470) and the error-map write-back of the pixel sampler (nvsf_error_map_update,
trainer.py:552-630), against the torch restatement oracle/torch_losses.py, and through RenderTrainStep / FrameSet over two epochs
(random pixels, then 2 x 8 patches drawn from the error map: trainer.py:1035-1062)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch_losses as TL  # noqa: E402

pytestmark = pytest.mark.gpu


def _patch_batch(rng, H, W, pH, pW, n_patch, edge=True):
    iy = rng.integers(0, H - pH, n_patch)
    ix = rng.integers(0, W - pW, n_patch)
    if edge:  # patches that touch the last row / column of the frame (the padded differences of the masks)
        iy[0], ix[0] = H - pH, W - pW
        iy[1], ix[1] = 0, W - pW
        iy[2], ix[2] = H - pH, 0
    r, c = np.meshgrid(np.arange(pH), np.arange(pW), indexing="ij")
    h = iy[:, None, None] + r[None]
    w = ix[:, None, None] + c[None]
    return (h * W + w).reshape(-1).astype(np.int64)


@pytest.mark.parametrize("sobel", [False, True])  # first differences (the default) / --sobel_grad
@pytest.mark.parametrize("criterion", ["l1", "mse", "huber", "smoothl1", "cos"])
@pytest.mark.parametrize("patch", [(2, 8), (4, 4), (8, 2)])
def test_structural_grad_loss_matches_the_restatement(dev, patch, criterion, sobel):
    from nvsf.nerf.train_step import LidarGradLossFn
    rng = np.random.default_rng(3)
    H, W, scale = 66, 1030, 0.010851959895748291
    pH, pW = patch
    n_patch = 4096 // (pH * pW)
    # a range image with flat stretches (mask on) and steps / noise (mask off), some dropped pixels
    rows = np.linspace(8.0, 40.0, W)[None, :] + 4.0 * np.sin(np.arange(H) / 7.0)[:, None]
    rows = rows + (rng.random((H, W)) < 0.15) * rng.normal(0, 3.0, (H, W)) + 6.0 * (np.arange(W)[None, :] // 97 % 2)
    drop = (rng.random((H, W)) > 0.2).astype(np.float32)
    frame = np.stack([drop, rng.random((H, W)), rows * scale * drop], -1).astype(np.float32)   # [raydrop, intensity, range x scale]
    inds = _patch_batch(rng, H, W, pH, pW, n_patch)
    gt_rd = torch.from_numpy(frame.reshape(-1, 3)[inds, 0])
    gt_d = torch.from_numpy(frame.reshape(-1, 3)[inds, 2])
    pred = (gt_d + torch.from_numpy(rng.normal(0, 0.05 * scale * 10, inds.shape).astype(np.float32))) * gt_rd
    pred_ref = pred.clone().requires_grad_()
    ref = TL.structural_grad_loss(pred_ref, gt_d * gt_rd, gt_rd, torch.from_numpy(inds), torch.from_numpy(frame[..., 2]), patch, scale, criterion, 0.1,
                                  sobel=sobel)
    (ref * 3.0).backward()
    pred_dev = pred.to(dev).view(1, -1).requires_grad_()
    got = LidarGradLossFn.apply(pred_dev, (gt_d * gt_rd).to(dev).view(1, -1), gt_rd.to(dev).view(1, -1), torch.from_numpy(inds).to(dev).view(1, -1),
                                torch.from_numpy(frame).to(dev)[None], patch, scale, criterion, 0.1, sobel)
    (got * 3.0).backward()
    assert float(ref) > 0
    assert abs(float(got) - float(ref)) <= 2e-5 * abs(float(ref))          # one deterministic tree sum against torch's
    g_ref, g_got = pred_ref.grad, pred_dev.grad.cpu().view(-1)
    assert float(g_ref.abs().max()) > 0
    # the cosine criterion divides by the patch norms: its gradient carries their rounding
    assert float((g_got - g_ref).abs().max()) <= (1e-4 if criterion == "cos" else 1e-5) * float(g_ref.abs().max())
    masked = (g_ref == 0)
    assert 0.01 < float(masked.float().mean()) < 0.95                      # masks bite ...
    if not sobel and criterion != "cos":                                   # ... identically (nine Sobel taps / the two cosine terms of a pixel
        assert bool((g_got[masked] == 0).all())                            #     cancel only up to rounding)


def test_cosine_criterion_on_patches_without_any_valid_gradient(dev):
    """A patch whose pixels are all dropped (or all on a range step) has u = v = 0: CosineSimilarity's eps clamp makes its cosine 0, the
    patch adds alpha pH pW 2 to the loss and nothing to the gradient (trainer.py:442-452 with main_nvsf.py:211)."""
    from nvsf.nerf.train_step import LidarGradLossFn
    H, W, scale, patch = 16, 64, 0.01, (2, 8)
    frame = torch.zeros(H, W, 3)
    frame[..., 2] = torch.linspace(5.0, 9.0, W)[None, :] * scale
    frame[..., 0] = 1.0
    inds = torch.cat([torch.arange(8), W + torch.arange(8), 20 + torch.arange(8), W + 20 + torch.arange(8)]).long()
    gt_rd = torch.cat([torch.zeros(16), torch.ones(16)])       # first patch dropped altogether
    gt_d = frame.reshape(-1, 3)[inds, 2] * gt_rd
    pred = ((gt_d + 0.003 * torch.randn(32, generator=torch.Generator().manual_seed(0))) * gt_rd)
    pred_ref = pred.clone().requires_grad_()
    ref = TL.structural_grad_loss(pred_ref, gt_d, gt_rd, inds, frame[..., 2], patch, scale, "cos", 0.1)
    ref.backward()
    pred_dev = pred.to(dev).view(1, -1).requires_grad_()
    got = LidarGradLossFn.apply(pred_dev, gt_d.to(dev).view(1, -1), gt_rd.to(dev).view(1, -1), inds.to(dev).view(1, -1), frame.to(dev), patch, scale,
                                "cos", 0.1)
    got.backward()
    assert float(ref) >= 0.1 * 16 * 2 and abs(float(got) - float(ref)) <= 1e-5 * float(ref)
    g = pred_dev.grad.cpu().view(-1)
    assert not g[:16].any() and g[16:].any()
    assert float((g - pred_ref.grad).abs().max()) <= 1e-4 * float(pred_ref.grad.abs().max())


def test_structural_grad_loss_rejects_what_is_not_built(dev):
    """An unknown criterion is an error; the three smoothness switches of the same block make the reference's loss a non-scalar tensor
    that its own backward call rejects (trainer.py:337-350, 1332), so there is nothing to match and the step says so."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import LidarGradLossFn, RenderTrainStep
    z = torch.zeros(1, 16, device=dev)
    with pytest.raises(ValueError):
        LidarGradLossFn.apply(z, z, z, torch.zeros(1, 16, dtype=torch.long, device=dev), torch.zeros(8, 8, 3, device=dev), (2, 8), 1.0, "l3", 0.1)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=12).to(dev)
    for switch in ("grad_norm_smooth", "spatial_smooth", "tv_loss"):
        with pytest.raises(NotImplementedError):
            RenderTrainStep(m, num_steps=16, grad_loss=True, **{switch: True})


@pytest.mark.parametrize("modality", ["lidar", "camera"])
def test_error_map_update_matches_the_restatement(dev, modality):
    from nvsf import _hip
    rng = np.random.default_rng(5)
    H, W = (66, 1030) if modality == "lidar" else (376, 1408)
    eH, eW = (H // 2, W // 2) if modality == "lidar" else (H // 4, W // 4)
    N = 4096
    inds = rng.integers(0, H * W, N).astype(np.int64)
    inds[100:140] = inds[100]  # many rays in one cell: the last one stays
    emap = (1.0 + rng.random((eH, eW)) * 50).astype(np.float32)
    if modality == "lidar":
        img, dep = rng.random((N, 2)).astype(np.float32), rng.random(N).astype(np.float32)
        rd, gi, gd = (rng.random(N) > 0.3).astype(np.float32), rng.random(N).astype(np.float32), rng.random(N).astype(np.float32)
        pd, gdm, gim, pim = dep * rd, gd * rd, gi * rd, img[:, 1] * rd
        loss = 1.0 * np.abs(pd - gdm) + 0.01 * (img[:, 0] - rd) ** 2 + 0.1 * (pim - gim) ** 2   # trainer.py:213-216 with the CLI's alphas
    else:
        a, b = rng.random((N, 3)).astype(np.float32), rng.random((N, 3)).astype(np.float32)
        loss = ((a - b) ** 2).sum(-1)
    ref = TL.error_map_update(torch.from_numpy(emap.copy()), torch.from_numpy(loss.astype(np.float32)), torch.from_numpy(inds), H, W).numpy()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    ray_loss = torch.empty(N, device=dev)
    stats = torch.tensor([0x7f800000, 0], dtype=torch.int64).to(torch.int32).to(dev)
    if modality == "lidar":
        ti, td, trd, tgi, tgd = t(img), t(dep), t(rd), t(gi), t(gd)
        _hip.call("nvsf_lidar_ray_losses", _hip.ptr(ti), _hip.ptr(td), _hip.ptr(trd), _hip.ptr(tgi), _hip.ptr(tgd), N, 1.0, 0.01, 0.1, 0.0,
                  _hip.ptr(ray_loss), _hip.ptr(stats))
    else:
        ta, tb = t(a), t(b)
        _hip.call("nvsf_mse_rows", _hip.ptr(ta), _hip.ptr(tb), N, 3, 1.0, _hip.ptr(ray_loss), _hip.ptr(stats))
    np.testing.assert_allclose(ray_loss.cpu().numpy(), loss, rtol=2e-6, atol=1e-7)
    lo, hi = stats.cpu().numpy().view(np.float32)
    assert lo == ray_loss.min().item() and hi == ray_loss.max().item()
    owner = torch.zeros(eH * eW, dtype=torch.int32, device=dev)
    dmap, tinds = t(emap), t(inds)
    _hip.call("nvsf_error_map_update", _hip.ptr(ray_loss), _hip.ptr(tinds), N, W, _hip.ptr(dmap), eH, eW, float(eH / H), float(eW / W), _hip.ptr(stats),
              _hip.ptr(owner))
    got = dmap.cpu().numpy()
    assert not owner.any()                                   # scratch handed back clean
    touched = got != emap
    assert 0.3 * N < touched.sum() <= N
    np.testing.assert_allclose(got, ref, rtol=3e-5, atol=1e-4)  # normalisation in fp32 both ways; same winner per cell


def test_patch_epochs_regularise_and_feed_the_error_map(dev, tmp_path):
    """Two epochs of the shipped configuration's loop on a synthetic dataset: epoch 1 samples random pixels (no `sr` term, sampler not
    drawing from the error map), epoch 2 samples 2 x 8 LiDAR patches from the error map and adds the structural term; every step
    writes its per-ray losses back into the frame's maps (LiDAR and camera)."""
    from test_formats_cpu import make_dataset
    from nvsf.nerf.dataset import formats as F
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    seq, frames, images, pcs, K = make_dataset(str(tmp_path), n_frames=2, H=24, W=32, Hl=16, Wl=64)
    fs = F.FrameSet(str(tmp_path), seq, "train", 0.0108, num_rays=40, num_rays_lidar=96, device=dev)  # 40 <= the 6 x 8 cells of the camera map (dataset_utils.py:613)
    m = NeRFNetworkStatic(bound=2.0, min_near=0.01, min_near_lidar=0.01, lidar_max_depth=0.9).to(dev)
    step = RenderTrainStep(m, num_steps=32, scale=0.0108, grad_loss=True, use_error_map=True)
    step.attach_error_maps(fs)
    assert tuple(fs.error_map.shape) == (2, 8, 32) and tuple(fs.error_map_rgb.shape) == (2, 6, 8) and bool((fs.error_map == 1).all())
    torch.manual_seed(0)
    assert step.set_epoch(1) == "random" and fs.patch_size_lidar == 1 and fs.use_error_map is False
    batch = fs.train_batch([0])
    loss, parts, _ = step.step(batch)
    assert "sr" not in parts and bool(torch.isfinite(loss))
    em, em_rgb = fs.error_map.clone(), fs.error_map_rgb.clone()
    assert bool((em[1] == 1).all()) and bool((em_rgb[1] == 1).all())                      # only the trained frame's maps move
    assert 10 < int((em[0] != 1).sum()) <= 96 and 5 < int((em_rgb[0] != 1).sum()) <= 40
    assert float(em[0].max()) <= 0.1 + 0.9 * 1000.0 + 1e-3 and float(em[0].min()) >= 1.0 - 1e-6
    # the cells that moved are the cells of the sampled pixels
    h, w = batch["rays_pano_inds"][0] // 64, batch["rays_pano_inds"][0] % 64
    cells = torch.zeros(8, 32, dtype=torch.bool, device=dev)
    cells[(h * 0.5).long(), (w * 0.5).long()] = True
    assert bool(((em[0] != 1) <= cells).all())
    assert step.set_epoch(2) == "patch" and fs.patch_size_lidar == (2, 8) and fs.use_error_map is True
    batch = fs.train_batch([0])
    inds = batch["rays_pano_inds"][0].view(-1, 2, 8)                                        # 96 rays = 6 patches of 2 x 8
    assert bool((inds[:, 0, 1:] - inds[:, 0, :-1] == 1).all()) and bool((inds[:, 1] - inds[:, 0] == 64).all())
    loss, parts, _ = step.step(batch)
    assert "sr" in parts and float(parts["sr"]) >= 0 and bool(torch.isfinite(loss))
    assert not torch.equal(fs.error_map[0], em[0])
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    step.sync()
