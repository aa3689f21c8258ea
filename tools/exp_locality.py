"""Experiment: how much of the density kernel's time is cache locality? (table size sweep, camera + lidar)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from nvsf import field_ops as ops, synthetic as S
from nvsf.nerf.models.network_static import NeRFNetworkStatic
from nvsf.nerf.raymarching import raymarching
dev = torch.device("cuda:0")
T = 768
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it
for log2T in (19, 17, 15, 12):
    torch.manual_seed(0)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=log2T).to(dev).eval()
    rng = np.random.default_rng(1000)
    for name, fn in (("lidar", S.lidar_rays), ("camera", S.camera_rays)):
        o, d = fn(4096, rng); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        if name == "lidar":
            nears = torch.full((4096,), float(m.min_near_lidar), device=dev); fars = torch.full((4096,), float(m.lidar_max_depth), device=dev)
        else:
            nears, fars = raymarching.near_far_from_aabb(o, d, m.aabb_infer, m.min_near)
        enc = m.hash_encoder_lidar if name == "lidar" else m.hash_encoder_camera
        t = timeit(lambda: ops.density_uniform(o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16()))
        print(f"log2T={log2T} table={enc.spec.n_params*2/1e6:.1f}MB {name}: {t:.4f} ms")
