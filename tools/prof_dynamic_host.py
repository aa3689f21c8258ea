import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S
from nvsf.nerf.models.network_dynamic import NeRFNetwork
dev = torch.device("cuda:0")
m = NeRFNetwork(time_resolution=8, num_frames=S.NUM_FRAMES, bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev).eval()
rng = np.random.default_rng(0)
lo, ld = S.lidar_rays(2048, rng)
tl = [torch.from_numpy(a).to(dev)[None] for a in (lo, ld)]
tm = torch.tensor([[0.5]], device=dev)
def step():
    with torch.no_grad():
        m.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=768)
    torch.cuda.synchronize()
for _ in range(2): step()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
