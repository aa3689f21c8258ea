"""The XCD slice balance of the camera encode pass away from the benchmark's shape: config-2 field, 4096 camera rays, for
T in {256, 512, 768, 1024} samples per ray and a scene bound of 2 and 4 (the rays then cross half as much of the unit cube): one
no-grad camera render (encode pass + tail) with the balanced plan against the home plan (every XCD group encodes its own slice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S, testing
from nvsf.nerf.models.network_static import NeRFNetworkStatic
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
co, cd = S.camera_rays(4096, rng)
co, cd = torch.from_numpy(co).to(dev)[None], torch.from_numpy(cd).to(dev)[None]
tm = torch.tensor([[0.5]], device=dev)


def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for bound in (S.BOUND, 2 * S.BOUND):
    torch.manual_seed(0)
    m = NeRFNetworkStatic(bound=bound, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES).to(dev).eval()
    for T in (256, 512, 768, 1024):
        def render():
            with torch.no_grad():
                m.render(co, cd, tm, cal_lidar_color=False, num_steps=T)
        bal = timed(render)
        with testing.variant(slice_plan="home"):
            home = timed(render)
        print(f"bound {bound}  T {T:4d}: balanced {bal:.3f} ms   home plan {home:.3f} ms   ({(home / bal - 1) * 100:+.1f} %)", flush=True)
