"""GPU parity: the nine `nvsf.nerf.raymarching.raymarching` operators (HIP, through the C ABI) against the CPU
oracle (oracle/raymarching_oracle.c) on the same seeded inputs.

Bars: bit-exact for integer / index / byte outputs and for the marcher (sample counts, offsets, positions,
step sizes: every decision is fp32 with identical rounding); 1e-5 abs for the compositors (wave-parallel
product scan + fast exp vs sequential expf; north_star tolerance is 1e-4).
"""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def rm(dev):
    from nvsf.nerf.raymarching import raymarching
    return raymarching


@pytest.fixture(scope="module")
def scene():
    from nvsf import synthetic as S
    rng = np.random.default_rng(0)
    grid = S.boxes_density_grid(rng, cascades=2, H=128, n_boxes=64)
    bits = O.packbits(grid, 0.5)
    return dict(grid=grid, bits=bits)


def _rays(n, seed, kind="cam"):
    from nvsf import synthetic as S
    rng = np.random.default_rng(seed)
    return (S.camera_rays if kind == "cam" else S.lidar_rays)(n, rng)


@pytest.mark.parametrize("n", [1, 63, 4096])
def test_near_far_from_aabb(rm, dev, n):
    o, d = _rays(n, 1)
    o = o * 8.0  # push some origins outside the box so that misses occur
    d[: n // 8, 0] = 0.0  # axis-parallel rays: 1/0 = inf must behave as in the reference
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nr, fr = O.near_far_from_aabb(o, d, aabb, 0.05)
    ng, fg = rm.near_far_from_aabb(_t(o, dev), _t(d, dev), _t(aabb, dev), 0.05)
    assert np.array_equal(nr, ng.cpu().numpy(), equal_nan=True)
    assert np.array_equal(fr, fg.cpu().numpy(), equal_nan=True)


def test_near_far_default_min_near_and_cpu_inputs(rm, dev):
    o, d = _rays(100, 2)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=torch.float32, device=dev)
    n1, f1 = rm.near_far_from_aabb(torch.from_numpy(o), torch.from_numpy(d), aabb)  # CPU rays are moved to the device
    nr, fr = O.near_far_from_aabb(o, d, aabb.cpu().numpy(), 0.2)
    assert n1.is_cuda and np.array_equal(nr, n1.cpu().numpy()) and np.array_equal(fr, f1.cpu().numpy())


def test_sph_from_ray(rm, dev):
    o, d = _rays(2048, 3)
    ref = O.sph_from_ray(o, d, 3.0)
    got = rm.sph_from_ray(_t(o, dev), _t(d, dev), 3.0).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-6, rtol=0)  # atan2f / sqrtf: device libm vs glibc


def test_morton_roundtrip_and_parity(rm, dev):
    rng = np.random.default_rng(4)
    c = rng.integers(0, 1024, size=(100000, 3)).astype(np.int32)
    ref = O.morton3D(c)
    got = rm.morton3D(_t(c, dev))
    assert np.array_equal(ref, got.cpu().numpy())
    back = rm.morton3D_invert(got)
    assert np.array_equal(back.cpu().numpy(), c)
    assert np.array_equal(O.morton3D_invert(ref), c)


def test_morton_full_128_cube_is_a_permutation(rm, dev):
    idx = torch.arange(128, device=dev, dtype=torch.int32)
    c = torch.stack(torch.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)
    m = rm.morton3D(c).long()
    assert torch.equal(torch.sort(m).values, torch.arange(128 ** 3, device=dev))


def test_packbits(rm, dev, scene):
    grid = scene["grid"]
    got = rm.packbits(_t(grid, dev), 0.5).cpu().numpy()
    assert np.array_equal(got, scene["bits"])
    assert np.array_equal(got, np.packbits(grid.reshape(-1) > 0.5, bitorder="little"))
    rng = np.random.default_rng(5)
    g2 = rng.standard_normal((2, 128 ** 3)).astype(np.float32)
    assert np.array_equal(rm.packbits(_t(g2, dev), 0.01).cpu().numpy(), O.packbits(g2, 0.01))


@pytest.mark.parametrize("march", ["wave", "serial", "thread"])  # wave-per-ray ChainWalker kernels (default: whole batches decided at once; serial: member by member) / one thread per ray
@pytest.mark.parametrize("n,max_steps,dt_gamma,perturb_seed", [(4096, 1024, 0.0, None), (1000, 256, 1.0 / 128, 7), (1, 64, 0.0, None)])
def test_march_rays_train(rm, dev, scene, n, max_steps, dt_gamma, perturb_seed, march, variants):
    variants.set(march=march)
    o, d = _rays(n, 6, "lidar" if n == 1000 else "cam")
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    noises = np.zeros(n, np.float32) if perturb_seed is None else np.random.default_rng(perturb_seed).random(n).astype(np.float32)
    M = n * max_steps
    xr, dr, lr, rr, cr = O.march_rays_train(o, d, scene["bits"], 2.0, dt_gamma, max_steps, 2, 128, M, nears, fars, noises)
    # drive the C ABI directly so that the same noises are used
    from nvsf import _hip
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf, tz = T(o), T(d), T(scene["bits"]), T(nears), T(fars), T(noises)
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
    _hip.call("nvsf_march_rays_train_passes", _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), 2.0, float(dt_gamma), max_steps, n, 2, 128, M,
              _hip.ptr(tn), _hip.ptr(tf), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
    assert np.array_equal(counter.cpu().numpy(), cr)
    assert np.array_equal(rays.cpu().numpy(), rr)
    m = int(cr[0])
    assert m > 0 or n == 1
    assert np.array_equal(xyzs.cpu().numpy()[:m], xr[:m])
    assert np.array_equal(dirs.cpu().numpy()[:m], dr[:m])
    assert np.array_equal(deltas.cpu().numpy()[:m], lr[:m])
    assert not xyzs[m:].any()


def _march_ws(dev, to, td, tb, tn, tf, tz, n, max_steps, dt_gamma, M, counter0=(0, 0), spin_limit=0):
    from nvsf import _hip
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(n, 3, dtype=torch.int32, device=dev)
    counter = torch.tensor(list(counter0), dtype=torch.int32, device=dev)
    nb = _hip.march_ws_bytes(n)
    ws = torch.full((nb // 8,), -1, dtype=torch.int64, device=dev)  # garbage on entry: the entry point clears it
    _hip.call("nvsf_march_rays_train_ws", _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), 2.0, float(dt_gamma), max_steps, n, 2, 128, M,
              _hip.ptr(tn), _hip.ptr(tf), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz),
              _hip.ptr(ws), nb, spin_limit)
    return counter.cpu(), rays.cpu(), xyzs.cpu(), dirs.cpu(), deltas.cpu()


@pytest.mark.parametrize("n,max_steps,dt_gamma,perturb_seed", [(4096, 1024, 0.0, None), (1000, 256, 1.0 / 128, 7), (1, 64, 0.0, None), (7, 33, 0.0, 3)])
def test_march_rays_train_ws_matches_oracle(rm, dev, scene, n, max_steps, dt_gamma, perturb_seed):
    """nvsf_march_rays_train_ws (one launch: count once, ranges from the scanner wave, replay of the recorded sample
    masks) against the oracle: counter, rays (ray-index order), positions, directions, step sizes bit for bit."""
    o, d = _rays(n, 6, "lidar" if n == 1000 else "cam")
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    noises = np.zeros(n, np.float32) if perturb_seed is None else np.random.default_rng(perturb_seed).random(n).astype(np.float32)
    M = n * max_steps
    xr, dr, lr, rr, cr = O.march_rays_train(o, d, scene["bits"], 2.0, dt_gamma, max_steps, 2, 128, M, nears, fars, noises)
    T = lambda a: _t(a, dev)
    counter, rays, xyzs, dirs, deltas = _march_ws(dev, T(o), T(d), T(scene["bits"]), T(nears), T(fars), T(noises), n, max_steps, dt_gamma, M)
    assert np.array_equal(counter.numpy(), cr) and np.array_equal(rays.numpy(), rr)
    m = int(cr[0])
    assert np.array_equal(xyzs.numpy()[:m], xr[:m]) and np.array_equal(dirs.numpy()[:m], dr[:m]) and np.array_equal(deltas.numpy()[:m], lr[:m])
    assert not xyzs[m:].any()


@pytest.mark.parametrize("n", [8192, 32768, 20])
@pytest.mark.parametrize("queue", ["queue0", "queue3", "queue7"])
def test_march_rays_train_ws_with_skewed_ticket_queues(rm, dev, scene, variants, n, queue):
    """The eight ticket queues do NOT advance in step here: the workgroups of one queue start ~0.5 ms late (test variant
    `march_skew`), so the others exhaust their own queues and take tickets of the lagging one that lie BELOW tickets they still
    have pending -- the case in which an unguarded steal makes a worker wait for a prefix that needs the very ticket it holds
    (the launch then only ends at the spin limit, with counter[1] < 0).  The early-own / late-any draw (raymarching.hip,
    march_draw_own / march_draw_any) must finish normally, in order, with the three-launch form's outputs bit for bit; n = 20 (five tickets, five workers on queues 1-5) leaves
    queue 0 -- ticket 0 -- without a workgroup of its own."""
    from nvsf import _hip
    max_steps = 256
    o, d = _rays(n, 21, "cam")
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf, tz = T(o), T(d), T(scene["bits"]), T(nears), T(fars), torch.zeros(n, device=dev)
    M = n * max_steps
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
    _hip.call("nvsf_march_rays_train_passes", _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), 2.0, 0.0, max_steps, n, 2, 128, M,
              _hip.ptr(tn), _hip.ptr(tf), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
    variants.set(march_skew=queue)
    # a spin limit far below the default (2^22 polls = seconds): a stalled launch fails this test quickly instead of hanging it
    c2, r2, x2, d2, l2 = _march_ws(dev, to, td, tb, tn, tf, tz, n, max_steps, 0.0, M, spin_limit=1 << 17)
    variants.clear("march_skew")
    assert int(c2[1]) == n, "a bounded wait expired: the ticket order stalled"
    assert torch.equal(c2, counter.cpu()) and torch.equal(r2, rays.cpu())
    assert torch.equal(x2, xyzs.cpu()) and torch.equal(d2, dirs.cpu()) and torch.equal(l2, deltas.cpu())


@pytest.mark.parametrize("kind", ["random10", "random50", "dense", "empty", "scene", "stripes"])
@pytest.mark.parametrize("dt_gamma,max_steps", [(0.0, 1024), (0.0, 300), (1.0 / 256, 512), (0.0, 4096)])
def test_march_rays_train_ws_equals_three_launch_form(rm, dev, scene, kind, dt_gamma, max_steps):
    """Same outputs as nvsf_march_rays_train on grids that stress the skip logic, with a pre-loaded counter, with a sample
    buffer too small for the batch (rays beyond M are recorded but write nothing), and -- "stripes" at 4096 steps: one sample
    every other batch -- with more sample-bearing batches per ray than the kernel keeps records for (the re-march path)."""
    from nvsf import _hip
    n = 3000 if max_steps == 4096 else 6000
    rng = np.random.default_rng(11)
    if kind == "scene":
        bits = scene["bits"]
    elif kind == "stripes":  # thin occupied slabs: many batches that each hold a few samples
        g = np.zeros((2, 128 ** 3), bool)
        idx = np.arange(128 ** 3)
        g[:, (idx % 7) == 0] = True
        bits = np.packbits(g.reshape(-1), bitorder="little")
    else:
        p = {"random10": 0.1, "random50": 0.5, "dense": 1.0, "empty": 0.0}[kind]
        bits = np.packbits(rng.random(2 * 128 ** 3) < p, bitorder="little")
    o, d = _rays(n, 12, "cam")
    o[: n // 2] *= 3.0
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    noises = rng.random(n).astype(np.float32)
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf, tz = T(o), T(d), T(bits), T(nears), T(fars), T(noises)
    for M, c0 in ((n * max_steps, (0, 0)), (n * max_steps // 8, (40, 3))):
        xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
        rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.tensor(list(c0), dtype=torch.int32, device=dev)
        _hip.call("nvsf_march_rays_train_passes", _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), 2.0, float(dt_gamma), max_steps, n, 2, 128, M,
                  _hip.ptr(tn), _hip.ptr(tf), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
        ref = (counter.cpu(), rays.cpu(), xyzs.cpu(), dirs.cpu(), deltas.cpu())
        got = _march_ws(dev, to, td, tb, tn, tf, tz, n, max_steps, dt_gamma, M, c0)
        assert (int(ref[0][0]) > c0[0]) == (kind != "empty")
        for a, b, name in zip(ref, got, ("counter", "rays", "xyzs", "dirs", "deltas")):
            assert torch.equal(a, b), (name, kind, M)


@pytest.mark.parametrize("C,H,bound", [(3, 64, 4.0), (2, 256, 2.0), (1, 128, 1.0), (4, 128, 8.0)])
@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 128])
def test_march_rays_train_other_grid_shapes(rm, dev, C, H, bound, dt_gamma):
    """Cascade counts and resolutions other than the reference's default 2 x 128^3: C H^3 <= 2^24 takes the integer cell index
    (march_device.h: every value an integer that fp32 holds exactly), 2 x 256^3 the reference's fp32 expression; the level clamp
    on the integer exponent has to agree with the float clamp for every C.  Both entry points against the oracle, bit for bit."""
    from nvsf import _hip
    n, max_steps = 700, 512
    rng = np.random.default_rng(C * 1000 + H)
    bits = np.packbits(rng.random(C * H ** 3) < 0.3, bitorder="little")
    o, d = _rays(n, 31, "cam")
    o *= np.float32(bound / 2.0)  # the synthetic rays are laid out for bound 2
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    noises = rng.random(n).astype(np.float32)
    M = n * max_steps
    xr, dr, lr, rr, cr = O.march_rays_train(o, d, bits, bound, dt_gamma, max_steps, C, H, M, nears, fars, noises)
    m = int(cr[0])
    assert m > 10000
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf, tz = T(o), T(d), T(bits), T(nears), T(fars), T(noises)
    # the three-launch form, the reference-shaped entry (one launch on a scratch block of the stream-ordered pool) and the one-launch
    # form on a caller-owned scratch block
    for entry in ("nvsf_march_rays_train_passes", "nvsf_march_rays_train", "nvsf_march_rays_train_ws"):
        xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
        rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
        extra = ()
        if entry.endswith("_ws"):
            nb = _hip.march_ws_bytes(n)
            ws = torch.empty(nb // 8, dtype=torch.int64, device=dev)
            extra = (_hip.ptr(ws), nb, 0)
        _hip.call(entry, _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), float(bound), float(dt_gamma), max_steps, n, C, H, M, _hip.ptr(tn), _hip.ptr(tf),
                  _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz), *extra)
        assert np.array_equal(counter.cpu().numpy(), cr) and np.array_equal(rays.cpu().numpy(), rr), entry
        assert np.array_equal(xyzs.cpu().numpy()[:m], xr[:m]) and np.array_equal(deltas.cpu().numpy()[:m], lr[:m]), entry


@pytest.mark.parametrize("p,dt_gamma", [(1.0, 0.0), (0.1, 0.0), (0.5, 1.0 / 256)])
def test_march_rays_train_ws_more_tickets_than_workgroups(rm, dev, p, dt_gamma):
    """20 000 rays = 5000 tickets of four rays for at most 2047 worker workgroups: every workgroup takes several tickets (both LDS
    record buffers in turn, stores one ticket behind the count) and the scanner wave walks more sums than one poll covers.
    Bit for bit the three-launch form."""
    from nvsf import _hip
    n, max_steps = 20000, 256
    rng = np.random.default_rng(5)
    bits = np.packbits(rng.random(2 * 128 ** 3) < p, bitorder="little")
    o, d = _rays(n, 21, "cam")
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    noises = rng.random(n).astype(np.float32)
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf, tz = T(o), T(d), T(bits), T(nears), T(fars), T(noises)
    M = n * max_steps
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
    _hip.call("nvsf_march_rays_train_passes", _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), 2.0, float(dt_gamma), max_steps, n, 2, 128, M,
              _hip.ptr(tn), _hip.ptr(tf), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
    ref = (counter.cpu(), rays.cpu(), xyzs.cpu(), dirs.cpu(), deltas.cpu())
    assert int(ref[0][0]) > 100000 and int(ref[0][1]) == n
    for _ in range(2):  # twice: the workspace is cleared by the call itself
        got = _march_ws(dev, to, td, tb, tn, tf, tz, n, max_steps, dt_gamma, M)
        for a, b, name in zip(ref, got, ("counter", "rays", "xyzs", "dirs", "deltas")):
            assert torch.equal(a, b), name


@pytest.mark.parametrize("kind", ["random10", "random50", "dense", "empty", "scene"])
@pytest.mark.parametrize("dt_gamma,max_steps", [(0.0, 1024), (0.0, 300), (1.0 / 256, 512)])
def test_march_rays_train_wave_forms_equal_the_thread_form(rm, dev, scene, kind, dt_gamma, max_steps, variants):
    """The one-thread-per-ray kernels are pinned to the oracle above; here the wave kernels (closed-form chain,
    batch-parallel visit decision, and the serial walk) must reproduce them bit for bit on grids that stress the skip
    logic: per-cell random occupancy (a jump every few members), fully occupied, empty, and the blocky test scene."""
    from nvsf import _hip
    n = 6000
    rng = np.random.default_rng(11)
    if kind == "scene":
        bits = scene["bits"]
    else:
        p = {"random10": 0.1, "random50": 0.5, "dense": 1.0, "empty": 0.0}[kind]
        bits = np.packbits(rng.random(2 * 128 ** 3) < p, bitorder="little")
    o, d = _rays(n, 12, "cam")
    o[: n // 2] *= 3.0  # half of the origins far from the centre / outside the inner cascade
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    noises = rng.random(n).astype(np.float32)
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf, tz = T(o), T(d), T(bits), T(nears), T(fars), T(noises)
    M = n * max_steps
    out = {}
    for march in ("thread", "wave", "serial"):
        variants.set(march=march)
        xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
        rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
        _hip.call("nvsf_march_rays_train_passes", _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), 2.0, float(dt_gamma), max_steps, n, 2, 128, M,
                  _hip.ptr(tn), _hip.ptr(tf), _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
        out[march] = (counter.cpu(), rays.cpu(), xyzs.cpu(), dirs.cpu(), deltas.cpu())
    total = int(out["thread"][0][0])
    assert (total > 0) == (kind != "empty")
    for march in ("wave", "serial"):
        for a, b in zip(out["thread"], out[march]):
            assert torch.equal(a, b), (march, kind)


def test_march_rays_train_wrapper_semantics(rm, dev, scene):
    """Python-level rules of raymarching.py:225-284: aligned slicing, mean_count overflow drops rays."""
    n = 512
    o, d = _rays(n, 8)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=torch.float32, device=dev)
    to, td, tb = _t(o, dev), _t(d, dev), _t(scene["bits"], dev)
    nears, fars = rm.near_far_from_aabb(to, td, aabb, 0.02)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 2.0, tb, 2, 128, nears, fars, counter, -1, False, 128, True, 0, 256)
    total = int(counter[0])
    assert int(counter[1]) == n and xyzs.shape[0] % 128 == 0 and xyzs.shape[0] >= total and xyzs.shape[0] - total <= 128
    assert int(rays[:, 2].sum()) == total
    assert torch.equal(rays[:, 1].long(), torch.cumsum(rays[:, 2].long(), 0) - rays[:, 2].long())  # ray-order prefix sums
    # under-estimated mean_count: rays past the budget are recorded but emit nothing, composite zeroes them
    small = max(total // 2, 1)
    counter.zero_()
    x2, d2, l2, r2 = rm.march_rays_train(to, td, 2.0, tb, 2, 128, nears, fars, counter, small, False, 128, False, 0, 256)
    M2 = x2.shape[0]
    assert M2 == small + (128 - small % 128)
    over = (r2[:, 1] + r2[:, 2]) > M2
    assert over.any()
    sig = torch.rand(M2, device=dev); rgb = torch.rand(M2, 3, device=dev)
    ws, dp, img = rm.composite_rays_train(sig, rgb, l2, r2)
    assert not ws[over].any() and not img[over].any() and not dp[over].any()


def _packed_inputs(scene, n=2048, max_steps=512, seed=9):
    o, d = _rays(n, seed)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    M = n * max_steps
    xr, dr, lr, rr, cr = O.march_rays_train(o, d, scene["bits"], 2.0, 0.0, max_steps, 2, 128, M, nears, fars, np.zeros(n, np.float32))
    m = int(cr[0])
    rng = np.random.default_rng(seed + 1)
    sig = (rng.random(m).astype(np.float32) * 60.0)
    sig[rng.random(m) < 0.3] = 0.0
    rgb = rng.random((m, 3)).astype(np.float32)
    return sig, rgb, lr[:m].copy(), rr


@pytest.mark.parametrize("T_thresh", [1e-4, 0.0, 0.5])
def test_composite_rays_train_forward(rm, dev, scene, T_thresh):
    sig, rgb, dl, rays = _packed_inputs(scene)
    wr, dr, ir = O.composite_rays_train_forward(sig, rgb, dl, rays, T_thresh)
    wg, dg, ig = rm.composite_rays_train(_t(sig, dev), _t(rgb, dev), _t(dl, dev), _t(rays, dev), T_thresh)
    np.testing.assert_allclose(wg.cpu().numpy(), wr, atol=1e-5, rtol=0)
    np.testing.assert_allclose(dg.cpu().numpy(), dr, atol=1e-5, rtol=0)
    np.testing.assert_allclose(ig.cpu().numpy(), ir, atol=1e-5, rtol=0)


def test_composite_rays_train_backward(rm, dev, scene):
    sig, rgb, dl, rays = _packed_inputs(scene, n=1024, seed=11)
    T_thresh = 1e-4
    ws, dp, img = O.composite_rays_train_forward(sig, rgb, dl, rays, T_thresh)
    rng = np.random.default_rng(12)
    g_ws, g_img = rng.standard_normal(ws.shape).astype(np.float32), rng.standard_normal(img.shape).astype(np.float32)
    gs_r, gc_r = O.composite_rays_train_backward(g_ws, g_img, sig, rgb, dl, rays, ws, img, T_thresh)
    ts, tc = _t(sig, dev).requires_grad_(), _t(rgb, dev).requires_grad_()
    wg, dg, ig = rm.composite_rays_train(ts, tc, _t(dl, dev), _t(rays, dev), T_thresh)
    (wg * _t(g_ws, dev)).sum().add((ig * _t(g_img, dev)).sum()).add(dg.sum() * 3.0).backward()  # depth grad must be ignored
    np.testing.assert_allclose(ts.grad.cpu().numpy(), gs_r, atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), gc_r, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("n_step", [4, 8, 3, 1])  # 8: the record-store march kernel; <= 8: eight lanes per ray in composite_rays
def test_inference_march_and_composite_loop(rm, dev, scene, n_step):
    """The n_alive loop of the reference's docstrings (raymarching.py:389-409, 480-493), against the oracle."""
    n, max_steps = 1500, 256
    o, d = _rays(n, 13)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    rng = np.random.default_rng(14)
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf = T(o), T(d), T(scene["bits"]), T(nears), T(fars)
    alive_r = np.arange(n, dtype=np.int32); t_r = nears.copy()
    ws_r, dp_r, img_r = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    alive_g, t_g = T(alive_r), T(t_r)
    ws_g, dp_g, img_g = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, 3, device=dev)
    for it in range(6):
        n_alive = len(alive_r)
        if n_alive == 0:
            break
        xr, dr, lr = O.march_rays(n_alive, n_step, alive_r, t_r, o, d, 2.0, 0.0, max_steps, 2, 128, scene["bits"], nears, fars,
                                  np.zeros(n_alive, np.float32))
        xg, dg, lg = rm.march_rays(n_alive, n_step, alive_g, t_g, to, td, 2.0, tb, 2, 128, tn, tf, -1, False, 0, max_steps)
        assert np.array_equal(xg.cpu().numpy(), xr) and np.array_equal(lg.cpu().numpy(), lr) and np.array_equal(dg.cpu().numpy(), dr)
        sig = rng.random(n_alive * n_step).astype(np.float32) * 80.0
        rgb = rng.random((n_alive * n_step, 3)).astype(np.float32)
        alive_r, t_r, ws_r, dp_r, img_r = O.composite_rays(n_alive, n_step, 1e-2, alive_r, t_r, sig, rgb, lr, ws_r, dp_r, img_r)
        rm.composite_rays(n_alive, n_step, alive_g, t_g, T(sig), T(rgb), lg, ws_g, dp_g, img_g, 1e-2)
        assert np.array_equal(alive_g.cpu().numpy(), alive_r)
        np.testing.assert_allclose(ws_g.cpu().numpy(), ws_r, atol=1e-5, rtol=0)
        np.testing.assert_allclose(img_g.cpu().numpy(), img_r, atol=1e-5, rtol=0)
        np.testing.assert_allclose(dp_g.cpu().numpy(), dp_r, atol=1e-5, rtol=0)
        np.testing.assert_allclose(t_g.cpu().numpy(), t_r, atol=0, rtol=0)
        keep = alive_r >= 0
        alive_r = alive_r[keep]
        alive_g = alive_g[alive_g >= 0].contiguous()


def test_empty_inputs(rm, dev):
    z3 = torch.zeros(0, 3, device=dev)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=dev)
    n, f = rm.near_far_from_aabb(z3, z3, aabb, 0.1)
    assert n.shape == (0,) and f.shape == (0,)
    assert rm.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=dev)).shape == (0,)


def test_march_rays_train_expired_wait_is_reported(rm, dev, scene):
    """nvsf_march_rays_train_ws with a spin limit of one poll (test-only argument): the scanner finds no published sum at its first
    look, gives up, and every worker gives up on its prefix -- the launch terminates and marks counter[1] negative (include/nvsf_hip.h).
    The operator must not hand such a call's samples on: when it reads the counter back (force_all_rays / no mean_count) it repeats the
    call through the three-launch entry point (an exception if the flag was already set by the caller's counter); when it does not
    (mean_count > 0) the next call or check_march_status() raises, and the un-read path must not hand garbage ranges to the
    compositor (`rays` rows stay (0, 0, 0))."""
    from nvsf import _hip
    n, max_steps = 4096, 256
    o, d = _rays(n, 11, "cam")
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    T = lambda a: _t(a, dev)
    to, td, tb, tn, tf = T(o), T(d), T(scene["bits"]), T(nears), T(fars)
    counter, rays, *_ = _march_ws(dev, to, td, tb, tn, tf, torch.zeros(n, device=dev), n, max_steps, 0.0, n * max_steps, spin_limit=1)
    assert int(counter[1]) < 0
    # the flag is sticky: a good launch on the same counter leaves it
    ctr = torch.tensor([0, -2 ** 31], dtype=torch.int32, device=dev)
    with pytest.raises(_hip.NvsfHipError, match="expired"):
        rm.march_rays_train(to, td, 2.0, tb, 2, 128, tn, tf, ctr, -1, False, 128, True, 0, max_steps)
    # read-back path: the call is repeated through the reference-shaped three-launch entry point (no inter-workgroup waits) with a
    # warning, and returns what that entry point returns
    with pytest.warns(UserWarning, match="expired"):
        got = rm.march_rays_train(to, td, 2.0, tb, 2, 128, tn, tf, None, -1, False, 128, True, 0, max_steps, "ws", 1)
    want = rm.march_rays_train(to, td, 2.0, tb, 2, 128, tn, tf, None, -1, False, 128, True, 0, max_steps, "ref")
    assert want[0].shape[0] > 0
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # no read-back (mean_count > 0): reported by the deferred check; the ranges handed on are empty, not garbage
    xyzs, dirs, deltas, rays = rm.march_rays_train(to, td, 2.0, tb, 2, 128, tn, tf, None, 4096, False, 128, False, 0, max_steps, "ws", 1)
    assert not rays.any()
    with pytest.raises(_hip.NvsfHipError, match="expired"):
        rm.check_march_status(wait=True)
    rm.check_march_status(wait=True)  # reported once
    # and a normal call afterwards is fine
    out = rm.march_rays_train(to, td, 2.0, tb, 2, 128, tn, tf, None, -1, False, 128, True, 0, max_steps)
    assert out[0].shape[0] > 0
    rm.check_march_status(wait=True)


def test_reference_shaped_entry_takes_its_scratch_from_the_stream_ordered_pool(rm, dev):
    """nvsf_march_rays_train has the reference's argument list (raymarching.h:27-44: no scratch) and runs the one-launch kernel on a block
    it takes from the device's stream-ordered pool around the launch: repeated calls, calls of different sizes back to back and calls on
    two streams give what nvsf_march_rays_train_passes gives, bit for bit; the wrapper's `_entry="c"` goes through it."""
    from nvsf import _hip
    rng = np.random.default_rng(41)
    C, H, bound, max_steps = 2, 128, 2.0, 256
    grid = (rng.random((C, H ** 3)) < 0.3).astype(np.float32)
    bits = O.packbits(grid, 0.5)
    tb = _t(bits, dev)
    side = torch.cuda.Stream(device=dev)

    def run(entry, n, stream=None):
        r = np.random.default_rng(n)
        o = ((r.random((n, 3)) * 2 - 1) * 0.5).astype(np.float32)
        d = r.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        nears, fars = O.near_far_from_aabb(o, d, np.array([-bound] * 3 + [bound] * 3, np.float32), 0.02)
        M = n * max_steps
        to, td, tn, tf = _t(o, dev), _t(d, dev), _t(nears, dev), _t(fars, dev)
        tz = torch.zeros(n, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
            xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
            rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
            _hip.call(entry, _hip.ptr(to), _hip.ptr(td), _hip.ptr(tb), bound, 0.0, max_steps, n, C, H, M, _hip.ptr(tn), _hip.ptr(tf),
                      _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
        torch.cuda.synchronize()
        m = int(counter[0])
        return counter.cpu(), rays.cpu(), xyzs[:m].cpu(), dirs[:m].cpu(), deltas[:m].cpu()

    for n, stream in ((700, None), (4000, None), (700, None), (33, side), (4000, side), (1, None)):
        ref = run("nvsf_march_rays_train_passes", n)
        got = run("nvsf_march_rays_train", n, stream)
        assert int(ref[0][0]) > 0 and int(got[0][1]) == n
        for a, b in zip(ref, got):
            assert torch.equal(a, b), (n, stream is not None)
    # through the drop-in wrapper
    n = 500
    r = np.random.default_rng(5)
    o = _t(((r.random((n, 3)) * 2 - 1) * 0.5).astype(np.float32), dev)
    d = torch.nn.functional.normalize(_t(r.normal(size=(n, 3)).astype(np.float32), dev), dim=-1)
    nears, fars = rm.near_far_from_aabb(o, d, torch.tensor([-bound] * 3 + [bound] * 3, device=dev), 0.02)
    outs = {e: rm.march_rays_train(o, d, bound, tb, C, H, nears, fars, None, -1, False, -1, True, 0, max_steps, e) for e in ("ws", "c", "ref")}
    for e in ("c", "ref"):
        for a, b in zip(outs["ws"], outs[e]):
            assert torch.equal(a, b), e
    assert outs["ws"][0].shape[0] > 1000


def test_reference_shaped_entry_does_not_grow_the_pool_it_borrows_from(rm, dev):
    """VERDICT r5 item 8 / SURVEY 8b "Ownership": nvsf_march_rays_train is the one entry point that borrows memory the caller did not
    pass in (1 KB + 16 B per four rays from the device's default stream-ordered pool).  1000 calls interleaved with torch allocations,
    frees, stream and device synchronisations (each of which returns a freed block to the OS under the pool's default threshold of
    0): the pool's reserved bytes do not grow after the first call, nothing stays handed out, and the threshold the library set is
    the documented 16 MiB (or whatever larger value the application had)."""
    import ctypes
    from nvsf import _hip
    lib = _hip.load()

    def stats():
        v = [ctypes.c_uint64(0) for _ in range(3)]
        assert lib.nvsf_scratch_pool_stats(*[ctypes.byref(x) for x in v]) == 0
        return tuple(int(x.value) for x in v)
    rng = np.random.default_rng(3)
    C, H, bound, max_steps, n = 1, 64, 1.0, 64, 2048
    tb = _t(O.packbits((rng.random((C, H ** 3)) < 0.2).astype(np.float32), 0.5), dev)
    o = _t(((rng.random((n, 3)) * 2 - 1) * 0.5).astype(np.float32), dev)
    d = torch.nn.functional.normalize(_t(rng.normal(size=(n, 3)).astype(np.float32), dev), dim=-1)
    nears, fars = rm.near_far_from_aabb(o, d, torch.tensor([-bound] * 3 + [bound] * 3, device=dev), 0.02)
    M = n * max_steps
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(n, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
    tz = torch.zeros(n, device=dev)

    def call():
        counter.zero_()
        _hip.call("nvsf_march_rays_train", _hip.ptr(o), _hip.ptr(d), _hip.ptr(tb), bound, 0.0, max_steps, n, C, H, M, _hip.ptr(nears), _hip.ptr(fars),
                  _hip.ptr(xyzs), _hip.ptr(dirs), _hip.ptr(deltas), _hip.ptr(rays), _hip.ptr(counter), _hip.ptr(tz))
    call()
    torch.cuda.synchronize()
    first = int(counter[0])
    reserved0, used0, threshold = stats()
    assert first > 0 and threshold >= 16 << 20 and used0 == 0 and 0 < reserved0 <= threshold
    keep = []
    for i in range(1000):
        call()
        if i % 7 == 0:
            keep.append(torch.empty(int(rng.integers(1, 1 << 20)), device=dev))   # the caller's allocator at work in between
        if i % 50 == 0:
            keep.clear()
            torch.cuda.synchronize()   # where a pool with threshold 0 gives its blocks back
        elif i % 13 == 0:
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    reserved1, used1, _ = stats()
    assert int(counter[0]) == first
    assert used1 == 0 and reserved1 <= reserved0, (reserved0, reserved1)
