"""A/B timing of density-kernel variants in ONE process (interleaved rounds), config-2 shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from nvsf import field_ops as ops, synthetic as S
from nvsf.nerf.models.network_static import NeRFNetworkStatic
from nvsf.nerf.raymarching import raymarching
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev).eval()
rng = np.random.default_rng(1000)
T = 768
batches = {}
for name, fn in (("lidar", S.lidar_rays), ("camera", S.camera_rays)):
    o, d = fn(4096, rng); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    if name == "lidar":
        nears = torch.full((4096,), float(m.min_near_lidar), device=dev); fars = torch.full((4096,), float(m.lidar_max_depth), device=dev)
    else:
        nears, fars = raymarching.near_far_from_aabb(o, d, m.aabb_infer, m.min_near)
    batches[name] = (o, d, nears, fars)
variants = sys.argv[1:] or ["1", "2"]
SLICED = [False]
def run(name):
    o, d, nears, fars = batches[name]
    enc = m.hash_encoder_lidar if name == "lidar" else m.hash_encoder_camera
    return ops.density_uniform(o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, m.sigma_net.weights_f16(), sliced=SLICED[0])
res = {}
outs = {}
for rnd in range(5):
    for v in variants:
        SLICED[0] = v == "S"
        os.environ["NVSF_DENSITY_KERNEL"] = "0" if v == "S" else v.split(":")[0]
        if ":" in v: os.environ["NVSF_DENSITY_SEG_TILES"] = v.split(":")[1]
        for name in batches:
            out = run(name); torch.cuda.synchronize()
            outs[(v, name)] = out
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): run(name)
            e.record(); e.synchronize()
            res.setdefault((v, name), []).append(s.elapsed_time(e) / 10)
for k, v in sorted(res.items()):
    print(k, "median %.4f ms  min %.4f" % (np.median(v), np.min(v)))
if len(variants) > 1:
    for name in batches:
        a, b = outs[(variants[0], name)], outs[(variants[1], name)]
        print(name, "z equal", torch.equal(a[0], b[0]), "sigma max rel", float(((a[1] - b[1]).abs() / b[1]).max()), "sigma equal", torch.equal(a[1], b[1]), "geo equal frac", float((a[2] == b[2]).float().mean()))
