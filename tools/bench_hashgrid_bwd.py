"""Table-gradient scatter of the config-2 grid (L16 F2 T2^19) and the reference-default static grid (L8 F4 T2^19) on the bench's
LiDAR / camera sample batches: nvsf_hashgrid_bwd against nvsf_hashgrid_bwd_binned for every choice of the first per-row level
(FF=..) and, with MERGE=a,b,.., of the first run-merged level; BATCH= / GRID= restrict the sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S, field_ops as ops
dev = torch.device("cuda:0")
N, T = 4096, 768
rng = np.random.default_rng(0)


def samples(o, d):
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    from nvsf.nerf.raymarching import raymarching
    aabb = torch.tensor([-S.BOUND] * 3 + [S.BOUND] * 3, dtype=torch.float32, device=dev)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, S.MIN_NEAR)
    z = nears[:, None] + (fars - nears)[:, None] * torch.linspace(0, 1, T, device=dev)[None]
    x = o[:, None] + d[:, None] * z[..., None]
    return ((x + S.BOUND) / (2 * S.BOUND)).clamp(0, 1).reshape(-1, 3).contiguous()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


grids = {"C2 L16 F2": ops.GridSpec(3, 16, 2, 19, 16, float(np.exp2(np.log2(2048 / 16) / 15))),
         "RD L8 F4": ops.GridSpec(3, 8, 4, 19, 512, float(np.exp2(np.log2(32768 / 512) / 7)))}
ONLY_B, ONLY_G, ONLY_FF = os.environ.get("BATCH"), os.environ.get("GRID"), os.environ.get("FF")
for tag, (o, d) in {"lidar": S.lidar_rays(N, rng), "camera": S.camera_rays(N, rng)}.items():
    if ONLY_B and tag != ONLY_B:
        continue
    x = samples(o, d)
    for gname, spec in grids.items():
        if ONLY_G and not gname.startswith(ONLY_G):
            continue
        g = torch.randn(x.shape[0], spec.L * spec.F, device=dev) * 1e-3
        ref = ops.hashgrid_backward(x, (0, 1, 2), spec, g)
        base = timed(lambda: ops.hashgrid_backward(x, (0, 1, 2), spec, g, grad_table=ref))
        print(f"{tag:7s} {gname:10s} corners {base:.3f} ms  (rule: fine_from = {ops.fine_levels_from(spec, T)})", flush=True)
        ref = ops.hashgrid_backward(x, (0, 1, 2), spec, g)
        for ff in range(spec.L - 1, -1, -1):
            if int(spec.res[ff]) ** 3 <= int(spec.offsets[ff + 1] - spec.offsets[ff]):
                break
            if ONLY_FF and ff != int(ONLY_FF):
                continue
            merges = [ff] if not os.environ.get("MERGE") else sorted({int(v) for v in os.environ["MERGE"].split(",") if int(v) <= ff})
            g_in = g.view(-1, spec.L, spec.F).permute(1, 0, 2).contiguous() if os.environ.get("LM", "1") == "1" else g  # level-major, as the MLP backward hands it over
            for mf in merges:
                out = ops.hashgrid_backward(x, (0, 1, 2), spec, g_in, fine_from=ff, merge_from=mf)
                err = float((out - ref).abs().max() / ref.abs().max())
                ms = timed(lambda: ops.hashgrid_backward(x, (0, 1, 2), spec, g_in, grad_table=out, fine_from=ff, merge_from=mf))
                print(f"    fine_from {ff:2d} merge_from {mf:2d}: {ms:.3f} ms   rel err {err:.2e}", flush=True)
