"""Times nvsf_march_rays_train_ws alone (32 768 camera rays x 1024 steps; dense and 10 % per-cell random grid), N_RAYS / MAX_STEPS / OCC (comma list of occupied fractions) / SORT_RAYS from the environment:  python tools/bench_march_only.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np
import torch


def main():
    from nvsf import _hip, synthetic as S
    from nvsf.nerf.raymarching import raymarching as rm
    dev = torch.device("cuda:0")
    P = _hip.ptr
    abs_ = sys.argv[1:] or ["0"]
    bound, C, H, max_steps, N = float(S.BOUND), 2, 128, int(os.environ.get("MAX_STEPS", 1024)), int(os.environ.get("N_RAYS", 32768))
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=dev)
    rng = np.random.default_rng(0)
    co, cd = S.camera_rays(N, rng)
    o, d = torch.from_numpy(co).to(dev), torch.from_numpy(cd).to(dev)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, float(S.MIN_NEAR))
    if os.environ.get("SORT_RAYS"):  # experiment: tickets of four rays of similar length (what does the per-ticket barrier cost?)
        order = torch.argsort(fars - nears)
        o, d, nears, fars = o[order].contiguous(), d[order].contiguous(), nears[order].contiguous(), fars[order].contiguous()
    M = N * max_steps
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    noises = torch.zeros(N, device=dev)
    ws_bytes = _hip.march_ws_bytes(N)
    workspace = torch.empty(ws_bytes // 8, dtype=torch.int64, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    ref = {}
    for occupied in [float(v) for v in os.environ.get("OCC", "1.0,0.1").split(",")]:
        dens = (torch.rand(C * H ** 3, device=dev, generator=g) < occupied).float()
        bitfield = rm.packbits(dens.view(C, -1), 0.5)

        def run():
            counter.zero_()
            _hip.call("nvsf_march_rays_train_ws", P(o), P(d), P(bitfield), bound, 0.0, max_steps, N, C, H, M, P(nears), P(fars),
                      P(xyzs), P(dirs), P(deltas), P(rays), P(counter), P(noises), P(workspace), ws_bytes, 0)
        for ab in abs_:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            times = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    run()
                b.record()
                b.synchronize()
                times.append(a.elapsed_time(b) / 10)
            m = int(counter[0])
            sig = (m, float(xyzs[:m].double().sum()), float(dirs[:m].double().sum()), float(deltas[:m].double().sum()), int(rays.long().sum()))
            ref.setdefault(occupied, sig)
            ok = sig == ref[occupied]
            nbytes = 48 * N + 32 * m
            best = min(times)
            print(f"occupied {occupied:4.0%}  AB={ab:>2s}  {best:.4f} ms (median {sorted(times)[2]:.4f})  {nbytes / best / 1e6:7.0f} GB/s  frac {nbytes / best / 1e6 / 8000:.3f}  "
                  f"samples {m}  same-as-first {ok}", flush=True)


if __name__ == "__main__":
    main()
