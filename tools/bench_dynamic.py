"""Timing of the full space-time field (reference default configuration 'RD': 2048 LiDAR + 2048 camera rays x 768
samples, hash L8 F4 T2^19 512->32768, time_resolution 8, K-planes 4 scales, flow field) -- forward render, no_grad, in the
reference's shipped --fp16 regime (render(fp16=True): the flow MLP on the fused fp16 MFMA kernel); FP32=1 for the fp32 flow MLP."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S
from nvsf.nerf.models.network_dynamic import NeRFNetwork
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NeRFNetwork(time_resolution=8, num_frames=S.NUM_FRAMES, bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev).eval()
print("params (M):", sum(p.numel() for p in m.parameters()) / 1e6)
if os.environ.get("FLOW_SCALE"):  # scene flow of this magnitude (unit cube units) instead of the ~0 of a fresh initialisation
    with torch.no_grad():
        for p in m.flow_net.grid_enc.parameters():
            p.uniform_(-0.5, 0.5)
        m.flow_net.mlp[-1].weight.normal_(0, float(os.environ["FLOW_SCALE"]) * 0.6)
rng = np.random.default_rng(0)
N, T = int(os.environ.get("N", 2048)), int(os.environ.get("T", 768))
lo, ld = S.lidar_rays(N, rng); co, cd = S.camera_rays(N, rng)
tl = [torch.from_numpy(a).to(dev)[None] for a in (lo, ld)]; tc = [torch.from_numpy(a).to(dev)[None] for a in (co, cd)]
tm = torch.tensor([[0.5]], device=dev)
FP16 = os.environ.get("FP32", "0") != "1"
def step():
    with torch.no_grad():
        m.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=T, fp16=FP16)
        m.render(tc[0], tc[1], tm, cal_lidar_color=False, num_steps=T, fp16=FP16)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 5
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
with torch.no_grad():
    xs = torch.rand(100000, 3, device=dev)
    fl = m.flow_net(torch.cat([xs, torch.full((100000, 1), 0.5, device=dev)], -1), 0.5, fp16=FP16)
print(f"mean |flow| = {float(fl.abs().mean()):.2e} (finest / coarsest space-time cell: {1/32768:.1e} / {1/512:.1e})")
print(f"RD forward: {dt*1e3:.2f} ms/step, {2*N/dt:.0f} rays/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
