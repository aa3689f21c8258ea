"""The static field with the reference-default hash grid (8 levels x 4 features, 512 -> 32768, T = 2^19): evaluation render of
4096 LiDAR + 4096 camera rays x 768 samples through the fused path (k_encode_sliced_f4 + k_render_tail2) against the unfused
composition (density_uniform's generic kernel -> compositor -> heads kernel), per modality."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S, field_ops as ops
from nvsf.nerf.models.network_static import NeRFNetworkStatic
from nvsf.nerf.raymarching import raymarching as rm
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES,
                      n_levels_hash=8, n_features_per_level_hash=4, base_resolution=512, max_resolution=32768, log2_hashmap_size=19).to(dev).eval()
N, T = int(os.environ.get("N", 4096)), int(os.environ.get("T", 768))
rng = np.random.default_rng(0)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
total = 0.0
for lidar in (True, False):
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    if lidar:
        nears, fars = torch.full((N,), float(m.min_near_lidar), device=dev), torch.full((N,), float(m.lidar_max_depth), device=dev)
    else:
        nears, fars = rm.near_far_from_aabb(o, d, m.aabb_infer, m.min_near)
    enc, net = (m.hash_encoder_lidar if lidar else m.hash_encoder_camera), m.sigma_net
    ha, hb = (m.raydrop_net.weights_f16(), m.intensity_net.weights_f16()) if lidar else (m.color_net.weights_f16(), None)
    bg = None if lidar else np.ones(3, np.float32)
    args = (o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, net.weights_f16(), lidar, ha, hb, m._k_scale(), bg, None)
    with torch.no_grad():
        bufs = ops.render_uniform(*args, sliced=True, _stage="encode")
        t_enc = timed(lambda: ops.render_uniform(*args, sliced=True, _stage="encode", _buffers=bufs))
        t_tail = timed(lambda: ops.render_uniform(*args, sliced=True, _stage="tail", _buffers=bufs))
        t_all = timed(lambda: ops.render_uniform(*args, sliced=True))
        def unfused():
            z, sg, geo = ops.density_uniform(o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, net.weights_f16(), None, sliced=False)
            w, ws, dp = ops.CompositeWeightsFn.apply(sg, z, nears, fars, m._k_scale())
            return ops.heads_uniform(w, geo, d, ws, lidar, ha, hb, bg)
        t_un = timed(unfused, reps=5)
    M = N * T
    print(f"{'lidar' if lidar else 'camera'}: fused render {t_all:.3f} ms (encode {t_enc:.3f} = {580.0 * M / t_enc / 1e6:.0f} GB/s of 580 B/sample, tail {t_tail:.3f}); "
          f"unfused kernels (generic gather + compositor + heads) {t_un:.3f} ms", flush=True)
    total += t_all
print(f"L8 F4 static field, {N}+{N} rays x {T}: {total:.3f} ms per step = {2 * N / total / 1e3:.2f} M rays/s")
