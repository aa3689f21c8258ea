"""Stand-alone timing of the K-planes texel scatter of a density query (nvsf_planes_multi_bwd: static + three time-plane evaluations sharing
one gradient slice, as PlanesMultiFn(blend=True).backward issues it) at the config-5 batch, production form against the round-5 form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from nvsf import synthetic as S, testing
from nvsf.nerf.models.planes_field import Planes4D
from planes_calls import multi_bwd_call
dev = torch.device("cuda:0")
N, T = int(os.environ.get("N", 4096)), 768
enc = Planes4D(resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8]).to(dev)
with torch.no_grad():
    enc.planes_cl.add_(torch.randn_like(enc.planes_cl) * 0.1)
rng = np.random.default_rng(0)
res = {}
for kind in ("lidar", "camera"):
    o, d = (S.lidar_rays if kind == "lidar" else S.camera_rays)(N, rng)
    z = torch.linspace(float(S.MIN_NEAR), float(S.LIDAR_MAX_DEPTH) if kind == "lidar" else 3.0, T, device=dev)
    x = torch.from_numpy(o).to(dev)[:, None, :] + torch.from_numpy(d).to(dev)[:, None, :] * z[None, :, None]
    x = ((x + S.BOUND) / (2 * S.BOUND)).clamp(0, 1).reshape(-1, 3).contiguous()
    M = x.shape[0]
    flow = (1e-4 * torch.sin(40.0 * torch.cat([x, x.flip(-1)], -1))).contiguous()
    g = torch.randn(M, 120, device=dev) * 1e-3
    times = [0.5, 0.5, 0.5 + 1 / 64, 0.5 - 1 / 64]
    gp = torch.zeros_like(enc.planes_cl)
    for variant in ("runs", "global"):
        with testing.variant(planes_bwd=variant):
            for _ in range(2):
                multi_bwd_call(enc, x, flow, g, times, dev, grad=gp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K = 5
            e0.record()
            for _ in range(K):
                multi_bwd_call(enc, x, flow, g, times, dev, grad=gp)
            e1.record(); torch.cuda.synchronize()
            res[(kind, variant)] = e0.elapsed_time(e1) / K
            print(f"{kind:6s} batch, M = {M}: planes_bwd={variant:6s} {res[(kind, variant)]:.3f} ms per call (static + 3 time-plane evaluations)")
