"""The level-sliced formulation of the fused density operator (nvsf_field_density_uniform_sliced_fwd: levels
partitioned over the XCDs + a streaming MLP pass) must reproduce the single-launch kernel bit for bit -- same
per-lane arithmetic, same MFMA operand order -- and therefore inherits its parity with the oracle
(test_render_static_gpu.py).  Also covers the host-side choice between the two and the argument checks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_render_static_gpu import _model, _oracle, _t


def _batch(m, dev, lidar, N, rng):
    from nvsf import synthetic as S
    from nvsf.nerf.raymarching import raymarching
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    o, d = _t(o, dev), _t(d, dev)
    if lidar:
        nears = torch.full((N,), float(m.min_near_lidar), device=dev)
        fars = torch.full((N,), float(m.lidar_max_depth), device=dev)
    else:
        nears, fars = raymarching.near_far_from_aabb(o, d, m.aabb_infer, m.min_near)
    return o, d, nears, fars


@pytest.mark.parametrize("lidar", [True, False])
@pytest.mark.parametrize("N,T,noise", [(64, 128, False), (37, 100, True), (1, 16, False), (300, 768, True), (5, 7, False)])
def test_sliced_equals_fused_bitwise(dev, lidar, N, T, noise):
    """T % 64 == 0 (wave-uniform rays) and ragged T, with and without perturbation, tiny and multi-block sizes."""
    from nvsf import field_ops as ops
    m = _model(dev, 0.1)
    rng = np.random.default_rng(11)
    o, d, nears, fars = _batch(m, dev, lidar, N, rng)
    nz = torch.rand(N, T, device=dev) if noise else None
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    args = (o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16(), nz)
    a = ops.density_uniform(*args, sliced=False)
    b = ops.density_uniform(*args, sliced=True)
    for x, y, name in zip(a, b, ("z_vals", "sigmas", "geo")):
        assert torch.equal(x, y), name


@pytest.mark.parametrize("N,T", [(400, 768), (700, 381)])  # T % 32 == 0 (scalar ray cursor) and ragged
def test_balanced_slice_plan_writes_the_same_planes(dev, N, T, variants):
    """Above 8192 units of 32 samples the encode pass moves the tail of the heavy slices (pairs of levels) to the workgroups of
    the light ones (slice_plan, fused_field.hip).  Who encodes a unit must not change what is written: the planes, z and the
    sigma MLP's outputs equal those of the home plan (every XCD its own slice) and of the single-launch kernel bit for bit."""
    from nvsf import field_ops as ops
    m = _model(dev, 0.1)
    rng = np.random.default_rng(23)
    o, d, nears, fars = _batch(m, dev, False, N, rng)
    enc = m.hash_encoder_camera
    args = (o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16())
    assert (N * T + 31) // 32 >= 8192
    outs = {}
    for plan in ("model", "home"):
        variants.set(slice_plan={"model": "balanced", "home": "home"}[plan])
        bufs = (torch.zeros(N, T, device=dev), torch.zeros(N, T, device=dev), torch.zeros(N, T, 16, dtype=torch.float16, device=dev),
                torch.zeros(16, N * T, dtype=torch.int32, device=dev))
        ops.density_uniform(*args, sliced=True, _buffers=bufs)
        outs[plan] = bufs
    variants.clear("slice_plan")
    for a, b in zip(outs["model"], outs["home"]):
        assert torch.equal(a, b)
    ref = ops.density_uniform(*args, sliced=False)
    for a, b in zip(outs["model"][:3], ref):
        assert torch.equal(a, b)


def test_sliced_render_matches_oracle(dev, variants):
    """Whole camera render through the sliced path (forced) against the CPU oracle composition, 1e-4."""
    from nvsf import synthetic as S
    variants.set(density_sliced=True)
    m = _model(dev, 0.1)
    rng = np.random.default_rng(13)
    N, T = 160, 128
    o, d = S.camera_rays(N, rng)
    ref = _oracle(m, o, d, False, T)
    with torch.no_grad():
        out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=False, num_steps=T)
    assert np.array_equal(out["z_vals"].cpu().numpy(), ref["z_vals"])
    np.testing.assert_allclose(out["weights"].cpu().numpy(), ref["weights"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["depth"][0].cpu().numpy(), ref["depth"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["image"][0].cpu().numpy(), ref["image"], atol=1e-4, rtol=0)


def test_sliced_passes_split(dev):
    """passes = 1 then passes = 2 on shared buffers == passes = 3."""
    from nvsf import field_ops as ops
    m = _model(dev, 0.1)
    rng = np.random.default_rng(17)
    N, T = 48, 192
    o, d, nears, fars = _batch(m, dev, False, N, rng)
    enc = m.hash_encoder_camera
    args = (o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16())
    ref = ops.density_uniform(*args, sliced=True)
    bufs = (torch.zeros(N, T, device=dev), torch.zeros(N, T, device=dev), torch.zeros(N, T, 16, dtype=torch.float16, device=dev),
            torch.zeros(16, N * T, dtype=torch.int32, device=dev))
    ops.density_uniform(*args, sliced=True, _passes=1, _buffers=bufs)
    assert torch.equal(bufs[0], ref[0]) and float(bufs[1].abs().sum()) == 0.0  # encode pass: z written, sigma untouched
    ops.density_uniform(*args, sliced=True, _passes=2, _buffers=bufs)
    assert torch.equal(bufs[1], ref[1]) and torch.equal(bufs[2], ref[2])


def test_prefer_sliced_choice_and_rejections(dev):
    from nvsf import field_ops as ops, _hip, synthetic as S
    m = _model(dev)
    spec = m.hash_encoder_camera.spec
    lidar_len = float(m.lidar_max_depth - m.min_near_lidar)
    assert ops.prefer_sliced(spec, 4096, 768, 2.0 * S.BOUND, S.BOUND) is True       # camera batch of config 2
    assert ops.prefer_sliced(spec, 4096, 768, lidar_len, S.BOUND) is False           # LiDAR batch: < 1 finest cell per step
    assert ops.prefer_sliced(spec, 64, 64, 2.0 * S.BOUND, S.BOUND) is False          # too small to pay for two launches
    f4 = ops.GridSpec(3, 8, 4, 19, 16, 1.5)
    assert ops.prefer_sliced(f4, 64, 64, lidar_len, S.BOUND) is True                 # L8 F4: fused kernels in the sliced form only (tests/test_l8f4_fused_gpu.py)
    small = ops.GridSpec(3, 4, 8, 19, 16, 1.5)
    assert ops.prefer_sliced(small, 4096, 768, 2.0 * S.BOUND, S.BOUND) is False      # shape the sliced kernels are not built for
    # the C entry point rejects such a shape instead of silently running something else
    o, d, nears, fars = _batch(m, dev, False, 8, np.random.default_rng(1))
    table = torch.zeros(small.n_params, dtype=torch.float16, device=dev)
    with pytest.raises(_hip.NvsfHipError):
        ops.density_uniform(o, d, nears, fars, 16, m._aabb_host, float(m.bound), table, small, m.sigma_net.weights_f16(), sliced=True)


@pytest.mark.parametrize("lidar", [True, False])
@pytest.mark.parametrize("N,T,noise", [(64, 128, False), (37, 100, True), (3, 16, False), (200, 768, True), (5, 7, False), (9, 48, False)])
def test_two_tile_tail_equals_one_tile_tail_and_gather_render(dev, lidar, N, T, noise, variants):
    """The sliced render's tail kernel takes two MFMA tiles per iteration (k_render_tail2: every weight fragment read from LDS
    feeds two MFMAs).  Per-tile arithmetic is that of the one-tile kernel, so all five outputs are bit-identical -- also for
    ragged T (last pair half empty, T < 16) -- and equal to the one-launch gather render up to its own fp32 scan order."""
    from nvsf import field_ops as ops
    m = _model(dev, 0.1)
    rng = np.random.default_rng(23)
    o, d, nears, fars = _batch(m, dev, lidar, N, rng)
    nz = torch.rand(N, T, device=dev) if noise else None
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    heads = (m.raydrop_net.weights_f16(), m.intensity_net.weights_f16()) if lidar else (m.color_net.weights_f16(), None)
    args = (o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16(), lidar, heads[0], heads[1],
            m._k_scale(), None if lidar else [1.0, 0.5, 0.25], nz)
    two = ops.render_uniform(*args, sliced=True)
    variants.set(render_tail="one")
    one = ops.render_uniform(*args, sliced=True)
    variants.clear("render_tail")
    for x, y, name in zip(two, one, ("z_vals", "weights", "weights_sum", "depth", "image")):
        assert torch.equal(x, y), name
    gather = ops.render_uniform(*args, sliced=False)
    assert torch.equal(two[0], gather[0]) and torch.equal(two[1], gather[1])
    for x, y in zip(two[2:], gather[2:]):
        assert float((x - y).abs().max()) <= 2e-6


@pytest.mark.parametrize("lidar", [True, False])
@pytest.mark.parametrize("N,T,noise", [(64, 128, False), (37, 100, True), (200, 768, True)])
def test_level_kinds_compiled_in_equal_the_run_time_form(dev, lidar, N, T, noise, variants):
    """On the grid of BASELINE config 2 (levels 0-4 dense, 5-15 hashed) the gathering kernels run the instance of density_encode that
    knows each level group's kind at compile time (only one index form per group, no selects; fused_field.hip kFirstHashedC2); on any
    other grid the general instance reads first_hashed at run time.  Same arithmetic: the evaluation render, the training forward with
    everything it keeps for the backward, and the occupancy render are bit-identical between the two instances."""
    from nvsf import field_ops as ops
    from nvsf import synthetic as S
    m = _model(dev, 0.1)
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    spec = enc.spec
    assert spec.L == 16 and spec.F == 2 and next(l for l in range(16) if spec.res[l] ** 3 > spec.offsets[l + 1] - spec.offsets[l]) == 5
    rng = np.random.default_rng(29)
    o, d, nears, fars = _batch(m, dev, lidar, N, rng)
    nz = torch.rand(N, T, device=dev) if noise else None
    heads = (m.raydrop_net.weights_f16(), m.intensity_net.weights_f16()) if lidar else (m.color_net.weights_f16(), None)
    args = (o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), spec, m.sigma_net.weights_f16(), lidar, heads[0], heads[1],
            m._k_scale(), None if lidar else [1.0, 0.5, 0.25], nz)
    targs = (o, d, nears, fars, T, m._aabb_host, float(m.bound), nz, enc.table_f16(), spec, m.sigma_net.weights_f16(), lidar, heads[0], heads[1],
             m._k_scale(), None if lidar else [1.0, 0.5, 0.25], ops.W_THRESH, False)
    res = {}
    for form in ("compiled", "runtime"):
        variants.set(level_kinds=form)
        res[form] = tuple(ops.render_uniform(*args, sliced=False)) + tuple(ops.render_uniform_train_forward(*targs))
    variants.clear("level_kinds")
    assert len(res["compiled"]) == 15
    for k, (a, b) in enumerate(zip(res["compiled"], res["runtime"])):
        assert torch.equal(a, b), k
    assert float(res["compiled"][2].max()) > 0.05


@pytest.mark.parametrize("lidar", [True, False])
def test_level_kinds_compiled_in_equal_the_run_time_form_in_the_occupancy_render(dev, lidar, variants):
    from nvsf import synthetic as S
    m = _model(dev, 0.1).enable_occupancy_grid().to(dev).eval()
    rng = np.random.default_rng(31)
    grid = S.boxes_density_grid(rng, cascades=m.cascade, H=m.grid_size, n_boxes=64)
    m.set_density_grid(torch.from_numpy(grid).to(dev), thresh=0.5)
    o, d = (S.lidar_rays if lidar else S.camera_rays)(300, rng)
    o, d = torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None]
    res = {}
    for form in ("compiled", "runtime"):
        variants.set(level_kinds=form)
        with torch.no_grad():
            out = m.render(o, d, torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, max_steps=512, T_thresh=1e-4, fused=True)
        res[form] = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    variants.clear("level_kinds")
    assert set(res["compiled"]) == set(res["runtime"]) and len(res["compiled"]) >= 3
    for k, a in res["compiled"].items():
        assert torch.equal(a, res["runtime"][k]), k
    sfx = "_lidar" if lidar else ""
    assert float(res["compiled"]["weights_sum" + sfx].max()) > 0.05
