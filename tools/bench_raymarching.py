"""Roofline of the raymarching-extension kernels at sizes where the launch is not
latency-bound.  At the BASELINE shapes (4096 rays) every one of these kernels moves < 1 MB and finishes inside the
launch latency, so a GB/s figure there says nothing about the kernel; here each kernel gets >= 100 MB of algorithmic
traffic.  Bytes per unit are SURVEY.md 8(d)'s figures.  Entry points are called through the C ABI with outputs
pre-allocated (the wrappers' zero fills are not part of the kernels).

    python tools/bench_raymarching.py          -> one JSON line with a row per kernel
    bench.py imports raymarching_rooflines() for its `raymarching` leg.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def _time_ms(fn, iters=10):
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def _row(kernel, ms, nbytes, per_unit, units):
    gbs = nbytes / (ms * 1e-3) / 1e9
    if os.environ.get("NVSF_BENCH_VERBOSE"):
        print(f"  {kernel}: {ms:.4f} ms, {gbs:.0f} GB/s", file=sys.stderr, flush=True)
    return {"kernel": kernel, "ms": ms, "bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "frac": gbs / HBM_PEAK_GBS, "per_unit": per_unit, "units": units}


def raymarching_rooflines(dev, n_rays_march=32768, occupied=1.0, seed=0):
    from nvsf import _hip, synthetic as S
    from nvsf.nerf.raymarching import raymarching as rm
    P = _hip.ptr
    g = torch.Generator(device=dev).manual_seed(seed)
    rows = []
    bound, C, H, max_steps = float(S.BOUND), 2, 128, 1024
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=dev)

    # a1 / a2: 2^22 rays
    N = 1 << 22
    o = (torch.rand(N, 3, device=dev, generator=g) - 0.5) * 0.6
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_near_far_from_aabb", P(o), P(d), P(aabb), N, float(S.MIN_NEAR), P(nears), P(fars)))
    rows.append(_row("near_far_from_aabb", ms, 32 * N, "32 B/ray", N))
    sph = torch.empty(N, 2, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_sph_from_ray", P(o), P(d), 4.0, N, P(sph)))
    rows.append(_row("sph_from_ray", ms, 32 * N, "32 B/ray", N))
    del o, d, nears, fars, sph

    # a3: 2^24 coordinate triples
    N = 1 << 24
    coords = torch.randint(0, 1024, (N, 3), dtype=torch.int32, device=dev, generator=g)
    idx = torch.empty(N, dtype=torch.int32, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_morton3D", P(coords), N, P(idx)))
    rows.append(_row("morton3D", ms, 16 * N, "16 B/element", N))
    ms = _time_ms(lambda: _hip.call("nvsf_morton3D_invert", P(idx), N, P(coords)))
    rows.append(_row("morton3D_invert", ms, 16 * N, "16 B/element", N))
    del coords, idx

    # a4: 16 cascades x 128^3 cells (the reference's grid is 2 x 128^3 = 16 MB: launch-bound)
    N = 16 * H ** 3
    grid = torch.rand(N, device=dev, generator=g)
    bits = torch.empty(N // 8, dtype=torch.uint8, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_packbits", P(grid), N // 8, 0.5, P(bits)))  # N of the ABI = output bytes
    rows.append(_row("packbits", ms, 33 * (N // 8), "33 B/output byte", N // 8))
    del grid, bits

    # a5-a7: camera-shaped rays through a grid with the given occupied fraction
    N = n_rays_march
    rng = np.random.default_rng(seed)
    co, cd = S.camera_rays(N, rng)
    o, d = torch.from_numpy(co).to(dev), torch.from_numpy(cd).to(dev)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, float(S.MIN_NEAR))
    dens = (torch.rand(C * H ** 3, device=dev, generator=g) < occupied).float()
    bitfield = rm.packbits(dens.view(C, -1), 0.5)
    M = N * max_steps
    xyzs, dirs = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev)
    deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    noises = torch.zeros(N, device=dev)

    def march():
        counter.zero_()
        _hip.call("nvsf_march_rays_train", P(o), P(d), P(bitfield), bound, 0.0, max_steps, N, C, H, M, P(nears), P(fars),
                  P(xyzs), P(dirs), P(deltas), P(rays), P(counter), P(noises))
    ms = _time_ms(march, 5)
    m = int(counter[0].item())
    rows.append(_row(f"march_rays_train[{occupied:.0%} occupied, reference-shaped entry: one launch, scratch from the stream-ordered pool]", ms,
                     48 * N + 32 * m, "48 B/ray + 32 B/sample", m))

    def march_passes():
        counter.zero_()
        _hip.call("nvsf_march_rays_train_passes", P(o), P(d), P(bitfield), bound, 0.0, max_steps, N, C, H, M, P(nears), P(fars),
                  P(xyzs), P(dirs), P(deltas), P(rays), P(counter), P(noises))
    ms = _time_ms(march_passes, 5)
    assert int(counter[0].item()) == m
    rows.append(_row(f"march_rays_train_passes[{occupied:.0%} occupied: count / scan / write launches]", ms, 48 * N + 32 * m,
                     "48 B/ray + 32 B/sample", m))
    ws_bytes = _hip.march_ws_bytes(N)
    workspace = torch.empty(ws_bytes // 8, dtype=torch.int64, device=dev)

    def march_ws():  # what raymarching.march_rays_train launches: one kernel (+ the memset of the 64 KB scan workspace)
        counter.zero_()
        _hip.call("nvsf_march_rays_train_ws", P(o), P(d), P(bitfield), bound, 0.0, max_steps, N, C, H, M, P(nears), P(fars),
                  P(xyzs), P(dirs), P(deltas), P(rays), P(counter), P(noises), P(workspace), ws_bytes, 0)
    ms = _time_ms(march_ws, 5)
    assert int(counter[0].item()) == m
    rows.append(_row(f"march_rays_train_ws[{occupied:.0%} occupied]", ms, 48 * N + 32 * m, "48 B/ray + 32 B/sample", m))

    sigmas = torch.rand(m, device=dev, generator=g) * 0.05  # T stays above T_thresh: every sample is read
    rgbs = torch.rand(m, 3, device=dev, generator=g)
    dl = deltas[:m].contiguous()
    ws, depth, image = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_composite_rays_train_forward", P(sigmas), P(rgbs), P(dl), P(rays), m, N, 1e-4,
                                    P(ws), P(depth), P(image)), 5)
    rows.append(_row("composite_rays_train_forward", ms, 24 * m + 32 * N, "24 B/sample + 32 B/ray", m))
    gws, gim = torch.rand(N, device=dev, generator=g), torch.rand(N, 3, device=dev, generator=g)
    gs, gr = torch.zeros(m, device=dev), torch.zeros(m, 3, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_composite_rays_train_backward", P(gws), P(gim), P(sigmas), P(rgbs), P(dl), P(rays),
                                    P(ws), P(image), m, N, 1e-4, P(gs), P(gr)), 5)
    rows.append(_row("composite_rays_train_backward", ms, 40 * m + 48 * N, "24 B/sample read + 16 B/sample written + 48 B/ray", m))
    del xyzs, dirs, deltas, sigmas, rgbs, dl, gs, gr

    # a8 / a9: one survivor round, 2^20 alive rays x 8 samples
    N = 1 << 20
    n_step = 8
    co, cd = S.camera_rays(N, rng)
    o, d = torch.from_numpy(co).to(dev), torch.from_numpy(cd).to(dev)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, float(S.MIN_NEAR))
    alive = torch.arange(N, dtype=torch.int32, device=dev)
    rays_t = nears.clone()
    M = N * n_step
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    noises = torch.zeros(N, device=dev)
    ms = _time_ms(lambda: _hip.call("nvsf_march_rays", N, n_step, P(alive), P(rays_t), P(o), P(d), bound, 0.0, max_steps, C, H,
                                    P(bitfield), P(nears), P(fars), P(xyzs), P(dirs), P(deltas), P(noises)), 5)
    rows.append(_row("march_rays[n_step 8]", ms, 40 * N + 32 * M, "40 B/ray + 32 B/sample slot", M))
    sigmas, rgbs = torch.rand(M, device=dev, generator=g) * 0.05, torch.rand(M, 3, device=dev, generator=g)
    ws, depth, image = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
    alive2, t2 = alive.clone(), rays_t.clone()

    def reset():  # T = 1 - weights_sum stays above T_thresh and no ray is marked dead (-1): every launch reads every slot
        ws.zero_()
        alive2.copy_(alive)
        t2.copy_(rays_t)

    def comp():
        reset()
        _hip.call("nvsf_composite_rays", N, n_step, 1e-2, P(alive2), P(t2), P(sigmas), P(rgbs), P(deltas), P(ws), P(depth), P(image))
    ms = max(_time_ms(comp, 5) - _time_ms(reset, 5), 1e-4)
    rows.append(_row("composite_rays[n_step 8]", ms, 24 * M + 48 * N, "24 B/sample + 48 B/ray", M))
    return rows


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    out = {"full": raymarching_rooflines(dev, occupied=1.0), "sparse": [r for r in raymarching_rooflines(dev, occupied=0.1) if "train" in r["kernel"]]}
    for k, rows in out.items():
        for r in rows:
            print(f"{k:7s} {r['kernel']:42s} {r['ms']:9.4f} ms  {r['achieved']:8.1f} GB/s  frac {r['frac']:.3f}  units {r['units']}", file=sys.stderr)
    print(json.dumps(out))
