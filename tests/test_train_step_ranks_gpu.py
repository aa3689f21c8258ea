"""GPU: table gradients that arrive in several scatters per step (RenderTrainStep(ray_chunks=2): each slice of the ray batch is its
own autograd sub-graph, each with its own DensityFn.backward) -- on one process and on two ranks that share cuda:0 over gloo (the
N > 1 control flow without a second GPU: GradBuckets, hold / release, the side-stream scatter into bucket views).  The scatters of
one table accumulate into ONE buffer on the side stream; a table's bucket may only be all-reduced after its LAST scatter."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _LossScaler():
    from nvsf.nerf.loss_scaler import LossScaler  # the step's loss scaler (GradScaler's rule and state_dict; nvsf/nerf/loss_scaler.py)
    return LossScaler

KW = dict(log2_hashmap_size=15)


def _model_and_batch(dev, batch_seed, n=512, T=64, dynamic=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "selfsupervised-nvsf_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from test_train_step_gpu import _batch
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **KW)
    torch.manual_seed(4)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=n, T=T, seed=batch_seed)

    def student():
        torch.manual_seed(9)
        if dynamic:  # the space-time field (K-planes, time-sliced grids, flow field): every kind of gradient sink user
            from nvsf.nerf.models.network_dynamic import NeRFNetwork
            return NeRFNetwork(time_resolution=4, num_frames=16, bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR,
                               lidar_max_depth=S.LIDAR_MAX_DEPTH, min_resolution=16, base_resolution=32, max_resolution=512,
                               log2_hashmap_size=13).to(dev)
        return NeRFNetworkStatic(**kw).to(dev)
    return student, batch, S


def _grads_of_one_step(student, batch, S, T=64, **step_kw):
    from nvsf.nerf.train_step import RenderTrainStep
    m = student()
    step = RenderTrainStep(m, num_steps=T, scale=S.SCALE, ema_decay=None, **step_kw)
    step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)
    torch.manual_seed(10)  # the jitter of perturb=True: drawn per render call, [N, T] at a time ...
    # ... so that slices of a batch see the jitter of the whole batch, the draws are replayed from one [N, T] table
    N = batch["rays_o"].shape[1]
    table = {True: torch.rand(N, T, device=batch["rays_o"].device), False: torch.rand(N, T, device=batch["rays_o"].device)}
    state = {"lidar": None, "row": 0}
    real_rand, real_render = torch.rand, m.render

    def render(o, d, t, cal_lidar_color=False, **kw):
        lidar = bool(cal_lidar_color)
        if state["lidar"] != lidar:
            state["lidar"], state["row"] = lidar, 0
        r0 = state["row"]
        state["row"] += o.shape[1]
        torch.rand = lambda *a, **k: table[lidar][r0:r0 + o.shape[1]]
        try:
            return real_render(o, d, t, cal_lidar_color=cal_lidar_color, **kw)
        finally:
            torch.rand = real_rand
    m.render = render
    loss, parts, n_coll = step.step(batch)
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.numel() and p.grad is not None}
    inv = 1.0 / float(step.scaler.get_scale())
    return {n: v * inv for n, v in g.items()}, float(loss), n_coll


def _assert_same(a, b, tol=2e-5, n_params=6):
    assert set(a) == set(b) and len(a) == n_params
    for n in a:
        scale = float(a[n].abs().max())
        assert scale > 0 and float((a[n] - b[n]).abs().max()) <= tol * scale, n


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("split", [True, False])
def test_ray_chunks_give_the_same_gradients_on_one_process(dev, overlap, split):
    student, batch, S = _model_and_batch(dev, 2)
    ref, loss_ref, _ = _grads_of_one_step(student, batch, S, ray_chunks=1, split_backward=split)
    from nvsf.nerf import train_step
    real = train_step.RenderTrainStep.__init__

    def init(self, *a, **k):
        real(self, *a, **k)
        self.scatter_overlap = overlap
    train_step.RenderTrainStep.__init__ = init
    try:
        got, loss, n_coll = _grads_of_one_step(student, batch, S, ray_chunks=2, split_backward=split)
    finally:
        train_step.RenderTrainStep.__init__ = real
    assert n_coll == 0 and abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    _assert_same(ref, got)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_worker(rank, world_size, port, q, dynamic=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    try:
        # single-process gradients of BOTH ranks' frames first (no process group yet: plain step, one scatter per table)
        per_rank = []
        for r in range(world_size):
            student, batch, S = _model_and_batch(dev, 20 + r, n=256, dynamic=dynamic)
            per_rank.append(_grads_of_one_step(student, batch, S, ray_chunks=1)[0])
        want = {n: sum(g[n] for g in per_rank) / world_size for n in per_rank[0]}
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        student, batch, S = _model_and_batch(dev, 20 + rank, n=256, dynamic=dynamic)
        got, _, n_coll = _grads_of_one_step(student, batch, S, ray_chunks=1 if dynamic else 2, bucket_bytes=1 << 20)
        def rel(n):
            scale, diff = float(want[n].abs().max()), float((got[n] - want[n]).abs().max())
            return diff / scale if scale > 0 else (0.0 if diff == 0 else float("inf"))
        worst = max(rel(n) for n in want)
        # every parameter lives in a bucket view on a multi-rank step: what no pass touched (time slices away from t) stays zero there
        untouched_zero = all(not bool(got[n].any()) for n in got if n not in want)
        q.put((rank, set(want) <= set(got) and untouched_zero and (dynamic or set(got) == set(want)), worst, n_coll))
    except Exception as e:  # noqa: BLE001 -- reported to the parent
        import traceback
        q.put((rank, False, traceback.format_exc() + repr(e), -1))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_ray_chunks_on_two_ranks_sharing_the_gpu(dev):
    """Two ranks (gloo, both on cuda:0), each with its own frame, ray_chunks = 2: the averaged gradients equal the mean of the two
    single-process gradients -- which fails if a table's bucket is all-reduced before the last slice has been scattered into it."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same_keys, worst, n_coll in results:
        assert same_keys, worst
        assert worst <= 2e-5, (rank, worst)
        assert n_coll >= 3
    assert results[0][3] == results[1][3]



def test_space_time_model_on_two_ranks_sharing_the_gpu(dev):
    """The same protocol with the space-time field: K-planes texel gradients, time-sliced grids, the flow grid and the static hash all
    reach their bucket views through the gradient sink (some from the side stream) -- the averaged gradients of two ranks equal the mean
    of the two single-process gradients on every parameter that has one."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same_keys, worst, n_coll in results:
        assert same_keys, worst
        assert worst <= 1e-4, (rank, worst)
        assert n_coll >= 3
    assert results[0][3] == results[1][3]
