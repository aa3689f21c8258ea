"""Timing of the multimodal training step (config 4 shape on one GPU): 4096 LiDAR + 4096 camera rays x 768 samples."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import numpy as np, torch
from nvsf import synthetic as S
from nvsf.nerf.models.network_static import NeRFNetworkStatic
from nvsf.nerf.train_step import RenderTrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
GRID = {}
if os.environ.get('GRID') == 'L8F4':  # the reference-default hash grid (main_nvsf.py:45-52) on the static field
    GRID = dict(n_levels_hash=8, n_features_per_level_hash=4, base_resolution=512, max_resolution=32768, log2_hashmap_size=19)
m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES, **GRID).to(dev)
if os.environ.get('NODE') == '0':     # training forward as DensityRaysFn + compositor + heads nodes instead of ONE node
    m.fused_train_render = False
if os.environ.get('FUSED') == '0':    # the operator chain
    m.fused_train_forward = m.fused_train_render = False
rng = np.random.default_rng(0)
N, T = int(os.environ.get("N", 4096)), int(os.environ.get("T", 768))
lo, ld = S.lidar_rays(N, rng); co, cd = S.camera_rays(N, rng)
g = torch.Generator(device="cpu").manual_seed(3)
batch = {"rays_o_lidar": torch.from_numpy(lo).to(dev)[None], "rays_d_lidar": torch.from_numpy(ld).to(dev)[None],
         "rays_o": torch.from_numpy(co).to(dev)[None], "rays_d": torch.from_numpy(cd).to(dev)[None], "time": torch.tensor([[0.5]], device=dev),
         "gt_depth": torch.rand(1, N, generator=g).to(dev) * 0.5, "gt_raydrop": (torch.rand(1, N, generator=g) > 0.3).float().to(dev),
         "gt_intensity": torch.rand(1, N, generator=g).to(dev), "gt_rgb": torch.rand(1, N, 3, generator=g).to(dev)}
step = RenderTrainStep(m, num_steps=T, scale=S.SCALE, ray_chunks=int(os.environ.get('CHUNKS', 1)), split_backward=os.environ.get('SPLIT', '1') == '1')
step.scatter_overlap = os.environ.get('OVERLAP', '1') == '1'
from nvsf import field_ops as _ops
if os.environ.get('HALF') == '0':  # the fp16 copies of tables / weights by a cast pass per step instead of by the optimiser pass
    step.opt.half_caches = {}
if os.environ.get('LM') == '0':
    _ops.LEVEL_MAJOR_GRADIENT = False
if os.environ.get('MERGE') == '0':  # run sums off: atomics for every level below the per-row ones
    _ops.merge_levels_from = lambda spec, fine, rows: fine
if os.environ.get('BINS') == '0':   # no bins at all
    _ops._bin_from = lambda spec, M, rows: None
if os.environ.get('DG'):  # composed / matrix: the density network's logit gradient inside the MLP backward or by nvsf_sigma_geo_bwd
    from nvsf import testing as _testing
    _cg = _testing.variant(density_grad=os.environ['DG']); _cg.__enter__()
if os.environ.get('MLP_BWD'):  # staged / wave: force one kernel of nvsf_mlp_bwd
    from nvsf import testing as _testing
    _cm = _testing.variant(mlp_bwd=os.environ['MLP_BWD']); _cm.__enter__()
if os.environ.get('SIDE_PRIO'):  # experiment: the scatter stream at another queue priority (lower number = higher priority)
    lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (None, None)
    print('priority range', lo_p, hi_p)
    _ops._SIDE_STREAMS[(dev.type, dev.index)] = torch.cuda.Stream(device=dev, priority=int(os.environ['SIDE_PRIO']))
if os.environ.get('SIDE_CUS'):  # experiment: the scatter stream restricted to a fraction of the CUs (hipExtStreamCreateWithCUMask)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    frac = float(os.environ['SIDE_CUS'])
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    bits = [0] * words
    pattern = os.environ.get('SIDE_PATTERN', 'stride')
    chosen = 0
    for i in range(n_cu):
        take = (i % 100) < frac * 100 if pattern == 'stride' else i < frac * n_cu
        if take:
            bits[i // 32] |= 1 << (i % 32); chosen += 1
    arr = (ctypes.c_uint32 * words)(*bits)
    stream = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(words), arr)
    print('cu mask stream rc', rc, 'CUs', chosen, 'of', n_cu)
    _ops._SIDE_STREAMS[(dev.type, dev.index)] = torch.cuda.ExternalStream(stream.value, device=dev)
for _ in range(2): step.step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(os.environ.get("K", 5))
for _ in range(K): step.step(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"train step: {dt*1e3:.2f} ms/step, {2*N/dt:.0f} rays/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
