"""GPU parity of the occupancy-grid render (BASELINE config 3: march_rays_train / composite_rays_train in training
mode, the march_rays / composite_rays survivor loop in evaluation mode) of the static hash field against the CPU
oracle composition (tests/oracle_lib.render_occupancy_*), plus the density-grid maintenance driver."""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# the grid of BASELINE config 2 (16 levels x 2 features) and the reference-default shape (8 levels x 4; here with dense coarse levels)
GRIDS = {"L16F2": {}, "L8F4": dict(n_levels_hash=8, n_features_per_level_hash=4, base_resolution=16, max_resolution=1024)}


@pytest.fixture(scope="module", params=list(GRIDS))
def setup(dev, request):
    from nvsf import synthetic as S, field_ops as ops
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(0)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=15,
                          **GRIDS[request.param])
    assert ops.occupancy_fused_eligible(m.hash_encoder_camera.spec)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
            enc.params.copy_(torch.randn(enc.params.shape, generator=g) * 0.1)
        m.sigma_net.params.mul_(2.0)
    m = m.to(dev).enable_occupancy_grid().to(dev)
    rng = np.random.default_rng(0)
    grid = S.boxes_density_grid(rng, cascades=m.cascade, H=m.grid_size, n_boxes=48)
    m.set_density_grid(_t(grid, dev), thresh=0.5)
    bits = O.packbits(grid, 0.5)
    assert np.array_equal(m.density_bitfield.cpu().numpy(), bits)
    return m, bits, S


def _field(m, lidar):
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    f16 = lambda net: net.params.detach().cpu().numpy().astype(np.float16)
    return (enc.params.detach().cpu().numpy().astype(np.float16), enc.spec, f16(m.sigma_net),
            f16(m.raydrop_net) if lidar else f16(m.color_net), f16(m.intensity_net) if lidar else None)


def _near_far(m, S, o, d, lidar):
    if lidar:
        return np.full(len(o), m.min_near_lidar, np.float32), np.full(len(o), m.lidar_max_depth, np.float32)
    return O.near_far_from_aabb(o, d, np.array([-S.BOUND] * 3 + [S.BOUND] * 3, np.float32), m.min_near)


def _assert_close_up_to_one_terminal_sample(out, ref, sfx, pick, T_thresh, t_max, min_exact):
    """1e-4 abs on every ray, except that a ray whose transmittance lands within float noise of T_thresh may take ONE more
    sample on one side only.  That sample is met with transmittance T ~ T_thresh (the compositor stops after the first sample
    whose incoming T is below the threshold, raymarching.cu:1015-1040), so its weight alpha * T is at most T_thresh: the exempt
    rays are bounded by that one weight -- weights_sum and every image channel (colour in [0, 1], background <= 1) by T_thresh,
    depth by T_thresh * t_max -- on top of the 1e-4."""
    ws = out["weights_sum" + sfx].cpu().numpy()[pick]
    dp = out["depth" + sfx][0].cpu().numpy()[pick]
    im = out["image" + sfx][0].cpu().numpy()[pick]
    e_ws, e_dp, e_im = np.abs(ws - ref["weights_sum"]), np.abs(dp - ref["depth"]), np.abs(im - ref["image"]).max(-1)
    exact = (e_ws <= 1e-4) & (e_dp <= 1e-4) & (e_im <= 1e-4)
    assert exact.mean() >= min_exact, exact.mean()
    w1 = 1.05 * T_thresh  # one terminal sample's weight (5 % for the noise in T itself)
    assert (e_ws <= 1e-4 + w1).all(), float(e_ws.max())
    assert (e_im <= 1e-4 + w1).all(), float(e_im.max())
    assert (e_dp <= 1e-4 + w1 * t_max).all(), float(e_dp.max())


@pytest.mark.parametrize("lidar", [True, False])
def test_training_mode_matches_oracle_and_backpropagates(dev, setup, lidar):
    m, bits, S = setup
    m.train()
    rng = np.random.default_rng(5)
    N, max_steps = 300, 256
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    nears, fars = _near_far(m, S, o, d, lidar)
    ref = O.render_occupancy_train(o, d, nears, fars, bits, float(S.BOUND), m.cascade, m.grid_size, max_steps, 0.0, np.zeros(N, np.float32),
                                   _field(m, lidar), lidar)
    out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, force_all_rays=True,
                   max_steps=max_steps, perturb=False)
    sfx = "_lidar" if lidar else ""
    assert int(m.step_counter[(m.local_step - 1) % 16, 0]) == ref["n_samples"] > 0
    np.testing.assert_allclose(out["weights_sum" + sfx].detach().cpu().numpy(), ref["weights_sum"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].detach().cpu().numpy(), ref["depth"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["image" + sfx][0].detach().cpu().numpy(), ref["image"], atol=1e-4, rtol=0)
    (out["image" + sfx].sum() + out["weights_sum" + sfx].sum()).backward()
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    assert enc.params.grad is not None and torch.isfinite(enc.params.grad).all() and enc.params.grad.abs().sum() > 0
    m.zero_grad(set_to_none=True)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("lidar", [True, False])
def test_evaluation_loop_matches_oracle(dev, setup, lidar, fused):
    """fused=False: the host loop over surviving rays (march_rays -> field -> composite_rays);
    fused=True: the one-launch kernel nvsf_render_occupancy_fwd.  Both against the oracle's loop."""
    m, bits, S = setup
    m.eval()
    rng = np.random.default_rng(6)
    N, max_steps = 257, 128
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    nears, fars = _near_far(m, S, o, d, lidar)
    ref = O.render_occupancy_infer(o, d, nears, fars, bits, float(S.BOUND), m.cascade, m.grid_size, max_steps, 0.0, _field(m, lidar), lidar,
                                   T_thresh=1e-2)
    with torch.no_grad():
        out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, max_steps=max_steps,
                       T_thresh=1e-2, fused=fused)
    sfx = "_lidar" if lidar else ""
    _assert_close_up_to_one_terminal_sample(out, ref, sfx, slice(None), 1e-2, float(fars[fars < 1e30].max()), min_exact=0.99)


@pytest.mark.parametrize("lidar", [True, False])
def test_fused_occupancy_equals_host_loop(dev, setup, lidar):
    """Larger batch, default T_thresh, coloured background: one-launch kernel vs the host loop of the same library.
    The two differ only in where a march restarts (the loop restarts from the compositor's accumulated t), i.e. by
    float noise; a ray may flip one borderline sample, hence the 99.5 % criterion."""
    m, bits, S = setup
    m.eval()
    rng = np.random.default_rng(8)
    N, max_steps = 2000, 512
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    args = (_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev))
    bg = None if lidar else torch.tensor([0.1, 0.5, 0.9], device=dev)
    from nvsf import field_ops as ops
    launches, real = [], ops.render_occupancy
    ops.render_occupancy = lambda *a_, **k_: (launches.append(1), real(*a_, **k_))[1]
    try:
        with torch.no_grad():
            a = m.render(*args, cal_lidar_color=lidar, max_steps=max_steps, bg_color=bg, fused=True)
            b = m.render(*args, cal_lidar_color=lidar, max_steps=max_steps, bg_color=bg, fused=False)
    finally:
        ops.render_occupancy = real
    assert len(launches) == 1  # the one-launch kernel ran for `fused=True` (either grid shape), the survivor loop for the other
    sfx = "_lidar" if lidar else ""
    assert float(b["weights_sum" + sfx].max()) > 0.05  # the batch does hit occupied space
    for k in ("weights_sum" + sfx, "depth" + sfx, "image" + sfx):
        x, y = a[k].reshape(N, -1), b[k].reshape(N, -1)
        ok = ((x - y).abs() <= 1e-4).all(-1).float().mean()
        assert float(ok) > 0.995, (k, float(ok))


def test_fused_occupancy_step_cap_and_empty(dev, setup):
    """max_steps caps the samples of a ray; rays that never meet an occupied cell return the background."""
    m, bits, S = setup
    m.eval()
    rng = np.random.default_rng(9)
    o, d = S.camera_rays(64, rng)
    args = (_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev))
    with torch.no_grad():
        few = m.render(*args, max_steps=4, fused=True)
        many = m.render(*args, max_steps=1024, fused=True)
    assert float(few["weights_sum"].max()) < float(many["weights_sum"].max())
    saved = m.density_bitfield.clone()
    m.density_bitfield.zero_()
    with torch.no_grad():
        empty = m.render(*args, max_steps=256, fused=True)
    m.density_bitfield.copy_(saved)
    assert float(empty["weights_sum"].abs().max()) == 0.0 and float(empty["depth"].abs().max()) == 0.0
    assert torch.equal(empty["image"], torch.ones_like(empty["image"]))


def test_density_grid_maintenance(dev, setup):
    m, bits, S = setup
    import copy
    mm = copy.deepcopy(m)
    mm.density_grid.zero_()
    mm.iter_density, mm.local_step = 0, 0
    t = torch.tensor([[0.5]], device=dev)
    mm.update_extra_state(t, cal_lidar_color=True)
    g1 = mm.density_grid.clone()
    assert mm.iter_density == 1 and (g1 > 0).all()  # every cell visited; sigma = exp(.) > 0
    # values are field densities at jittered cell centres: spot-check against the field at the exact centres (same order of magnitude)
    idx = torch.tensor([[10, 20, 30], [64, 64, 64], [127, 0, 5]], dtype=torch.int32, device=dev)
    from nvsf.nerf.raymarching import raymarching
    mo = raymarching.morton3D(idx).long()
    centres = (2 * idx.float() / (mm.grid_size - 1) - 1) * (1 - 1 / mm.grid_size)
    with torch.no_grad():
        sig = mm.density(centres, t, True)["sigma"]
    ratio = g1[0, mo] / sig
    assert (ratio > 0.2).all() and (ratio < 5).all()
    mm.update_extra_state(t, cal_lidar_color=True)
    assert (mm.density_grid >= g1 * 0.95 - 1e-6).all()  # exponential moving maximum with decay 0.95
    thr = min(mm.mean_density, mm.density_thresh)
    assert np.array_equal(mm.density_bitfield.cpu().numpy(), O.packbits(mm.density_grid.cpu().numpy(), thr))
    mm.iter_density = 16
    mm.update_extra_state(t, cal_lidar_color=True)  # partial update path
    assert mm.iter_density == 17 and torch.isfinite(mm.density_grid).all()


@pytest.mark.parametrize("lidar", [True, False])
def test_config3_full_size_matches_oracle(dev, lidar):
    """BASELINE config 3 as stated: the config-2 field (L16 F2, log2_hashmap_size 19), occupancy grid, at most 1024 samples per
    ray, early termination at T < 1e-4 -- the whole 4096-ray batch through the one-launch kernel, 48 of its rays against the
    oracle's survivor loop (march_rays -> field -> composite_rays)."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(0)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH)
    assert m.hash_encoder_lidar.spec.log2_hashmap_size == 19 and m.hash_encoder_lidar.spec.L == 16
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
            enc.params.copy_(torch.randn(enc.params.shape, generator=g) * 0.1)
        m.sigma_net.params.mul_(2.0)
    m = m.to(dev).enable_occupancy_grid().to(dev).eval()
    rng = np.random.default_rng(0)
    grid = S.boxes_density_grid(rng, cascades=m.cascade, H=m.grid_size, n_boxes=64)
    m.set_density_grid(_t(grid, dev), thresh=0.5)
    bits = O.packbits(grid, 0.5)
    N, K, max_steps = 4096, 48, 1024
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    with torch.no_grad():
        out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, max_steps=max_steps,
                       T_thresh=1e-4, fused=True)
    pick = np.sort(rng.choice(N, K, replace=False))
    nears, fars = _near_far(m, S, o[pick], d[pick], lidar)
    ref = O.render_occupancy_infer(o[pick], d[pick], nears, fars, bits, float(S.BOUND), m.cascade, m.grid_size, max_steps, 0.0, _field(m, lidar),
                                   lidar, T_thresh=1e-4)
    sfx = "_lidar" if lidar else ""
    assert float(ref["weights_sum"].max()) > 0.05
    _assert_close_up_to_one_terminal_sample(out, ref, sfx, pick, 1e-4, float(fars[fars < 1e30].max()), min_exact=0.97)
