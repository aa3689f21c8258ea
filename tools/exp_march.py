import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import _hip, synthetic as S
from nvsf.nerf.raymarching import raymarching as rm
dev = torch.device("cuda:0"); P = _hip.ptr
bound, C, H, max_steps = float(S.BOUND), 2, 128, 1024
aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=dev)
N = 32768
rng = np.random.default_rng(0)
co, cd = S.camera_rays(N, rng)
o, d = torch.from_numpy(co).to(dev), torch.from_numpy(cd).to(dev)
nears, fars = rm.near_far_from_aabb(o, d, aabb, float(S.MIN_NEAR))
for occ in (1.0, 0.1):
    g = torch.Generator(device=dev).manual_seed(0)
    dens = (torch.rand(C * H ** 3, device=dev, generator=g) < occ).float()
    bitfield = rm.packbits(dens.view(C, -1), 0.5)
    M = N * max_steps
    xyzs, dirs = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev)
    deltas = torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    noises = torch.zeros(N, device=dev)
    ws_bytes = _hip.march_ws_bytes(N)
    workspace = torch.empty(ws_bytes // 8, dtype=torch.int64, device=dev)
    def march_ws():
        counter.zero_()
        _hip.call("nvsf_march_rays_train_ws", P(o), P(d), P(bitfield), bound, 0.0, max_steps, N, C, H, M, P(nears), P(fars),
                  P(xyzs), P(dirs), P(deltas), P(rays), P(counter), P(noises), P(workspace), ws_bytes)
    def march_ref():
        counter.zero_()
        _hip.call("nvsf_march_rays_train", P(o), P(d), P(bitfield), bound, 0.0, max_steps, N, C, H, M, P(nears), P(fars),
                  P(xyzs), P(dirs), P(deltas), P(rays), P(counter), P(noises))
    march_ref(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): march_ref()
    b.record(); b.synchronize()
    print(f"occ {occ} three-launch reference entry: {a.elapsed_time(b)/10:.4f} ms", flush=True)
    for dbg in (0, 4, 0, 4):
        os.environ["NVSF_MARCH_DEBUG"] = str(dbg)
        march_ws(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): march_ws()
        b.record(); b.synchronize()
        print(f"occ {occ} dbg {dbg}: {a.elapsed_time(b)/10:.4f} ms", flush=True)
