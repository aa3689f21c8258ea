"""Experiment: what the look-ups of each class of levels cost inside the headline's render kernels.

The level scale of a set of levels is set to zero, so that every sample falls into cell 0 of those levels: the same instructions are issued
(index arithmetic, eight gathers, blend), but all lanes of a gather read ONE table entry -- the cheapest look-up there is, an upper bound of
what re-using a cell's corners across consecutive samples could save on those levels.  Outputs are of course different: timing only.

    python tools/exp_level_cost.py            # LiDAR batch and camera batch, 4096 rays x 768 samples each
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S, _hip
from nvsf.nerf.models.network_static import NeRFNetworkStatic

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES).to(dev).eval()
rng = np.random.default_rng(0)
N, T = 4096, 768
lo, ld = S.lidar_rays(N, rng); co, cd = S.camera_rays(N, rng)
rays = {"lidar": (torch.from_numpy(lo).to(dev)[None], torch.from_numpy(ld).to(dev)[None]),
        "camera": (torch.from_numpy(co).to(dev)[None], torch.from_numpy(cd).to(dev)[None])}
tm = torch.tensor([[0.5]], device=dev)


def timed(kind, iters=60):
    o, d = rays[kind]
    with torch.no_grad():
        for _ in range(5):
            m.render(o, d, tm, cal_lidar_color=kind == "lidar", num_steps=T)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(iters):
            m.render(o, d, tm, cal_lidar_color=kind == "lidar", num_steps=T)
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


if "--forms" in sys.argv:  # each batch through both forms of the render: one-launch gather / level-sliced encode + streaming tail
    from nvsf import testing
    for form in (False, True):
        with testing.variant(density_sliced=form):
            print("sliced" if form else "gather", "lidar %.4f ms  camera %.4f ms" % (timed("lidar", 200), timed("camera", 200)))
    sys.exit(0)
if "--shipped-only" in sys.argv:  # A/B of two builds of the library (tools/ab_run.sh): the two renders as shipped, nothing else
    print("lidar %.4f ms  camera %.4f ms" % (timed("lidar", 300), timed("camera", 300)))
    sys.exit(0)

for kind in ("lidar", "camera"):
    enc = m.hash_encoder_lidar if kind == "lidar" else m.hash_encoder_camera
    spec = enc.spec
    real = list(spec.scales)
    L = spec.L
    first_hashed = next((l for l in range(L) if spec.res[l] ** 3 > spec.offsets[l + 1] - spec.offsets[l]), L)
    print(f"{kind}: res {spec.res}, first hashed level {first_hashed}")
    cases = [("as shipped", []), ("levels 0-5 in one cell", range(0, 6)), ("levels 0-8 in one cell", range(0, 9)),
             ("levels 9-15 in one cell", range(9, 16)), ("levels 12-15 in one cell", range(12, 16)), ("all levels in one cell", range(L))]
    for name, flat in cases:
        s = list(real)
        for l in flat:
            s[l] = 0.0
        spec.h_scales = _hip.host_f32(s)
        print(f"  {name:28s} {timed(kind):.4f} ms")
    spec.h_scales = _hip.host_f32(real)
