"""Experiment: render the LiDAR and the camera batch of a step on two HIP streams vs one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S
from nvsf.nerf.models.network_static import NeRFNetworkStatic
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev).eval()
rng = np.random.default_rng(1000)
lo, ld = S.lidar_rays(4096, rng); co, cd = S.camera_rays(4096, rng)
tl = [torch.from_numpy(a).to(dev)[None] for a in (lo, ld)]; tc = [torch.from_numpy(a).to(dev)[None] for a in (co, cd)]
tm = torch.tensor([[0.5]], device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    with torch.no_grad():
        m.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=768)
        m.render(tc[0], tc[1], tm, cal_lidar_color=False, num_steps=768)
def two():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(sa):
            m.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=768)
        with torch.cuda.stream(sb):
            m.render(tc[0], tc[1], tm, cal_lidar_color=False, num_steps=768)
    cur.wait_stream(sa); cur.wait_stream(sb)
for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name}: {dt*1e3:.3f} ms/step  {8192/dt/1e6:.2f} M rays/s")
