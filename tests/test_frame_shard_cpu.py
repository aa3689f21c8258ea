"""World-size-2 tests of the frame-sharding logic on CPU (gloo): frame partition, ray chunking + all-gather, bucketed
gradient all-reduce (the one collective of the training path).  Rendezvous on 127.0.0.1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "selfsupervised-nvsf_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from nvsf import frame_shard as FS
    try:
        # --- gradients: ragged parameter sizes, several buckets, a parameter that is unused on one rank
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 1000, 3, 50000, 129)]
        params.append(torch.nn.Parameter(torch.zeros(4, 5), requires_grad=False))
        for i, p in enumerate(params[:5]):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
        if rank == 1:
            params[2].grad = None  # e.g. the camera tables in a LiDAR-only step of this rank
        n_coll = FS.allreduce_gradients(params, bucket_bytes=4096)
        expect = [(1 + 2) * (i + 1) / 2.0 for i in range(5)]
        expect[2] = (1 * 3 + 0) / 2.0
        ok_grad = all(torch.allclose(p.grad, torch.full_like(p, e)) for p, e in zip(params[:5], expect)) and params[5].grad is None
        # --- the overlapped form used by the training step: hooks launch a bucket's all-reduce during backward, gradients live
        # in flat bucket views, a parameter NO rank produced a gradient for ends with grad None (same training with
        # any number of ranks), and the results equal the one-shot path
        torch.manual_seed(1)
        w = [torch.nn.Parameter(torch.randn(n)) for n in (300, 5, 70000, 11)]
        gb = FS.GradBuckets(w, bucket_bytes=2048)
        x = torch.arange(1.0, 4.0) * (rank + 1)
        gb.begin_step()
        loss = (w[0][:3] * x).sum() + (w[2][:3] * x * 2).sum() + ((w[3][:3] * x * 3).sum() if rank == 0 else 0.0)  # w[1] unused everywhere
        loss.backward()
        n_over = gb.finish()
        exp0 = torch.zeros(300); exp0[:3] = torch.arange(1.0, 4.0) * 1.5
        exp3 = torch.zeros(11); exp3[:3] = torch.arange(1.0, 4.0) * 1.5  # only rank 0 contributes: (3 x + 0) / 2
        ok_grad = ok_grad and torch.allclose(w[0].grad, exp0) and torch.allclose(w[2].grad[:3], exp0[:3] * 2) and w[1].grad is None \
            and torch.allclose(w[3].grad, exp3) and w[0].grad.data_ptr() == gb.views[w[0]].data_ptr() and n_over >= 3
        # --- a step made of two backward passes (LiDAR pass, camera pass): a parameter both passes reach must be reduced after the
        # second one, and what only the first pass touched is released afterwards
        gb.begin_step()
        gb.hold(True)
        ((w[0][:3] * x).sum() + (w[2][:3] * x).sum()).backward()
        gb.hold(False)
        ((w[2][:3] * x).sum() + ((w[3][:3] * x * 3).sum() if rank == 0 else 0.0)).backward()
        gb.release_held()
        gb.finish()
        ok_grad = ok_grad and torch.allclose(w[0].grad, exp0) and torch.allclose(w[2].grad[:3], exp0[:3] * 2) and w[1].grad is None \
            and torch.allclose(w[3].grad, exp3)
        gb.close()
        # --- rays: ragged chunks, gather restores the original order
        N = 4099
        full = torch.arange(N * 3, dtype=torch.float32).view(N, 3)
        b, e = FS.ray_chunk(N, rank, world_size)
        gathered = FS.gather_ray_outputs(full[b:e] * 2.0, N)
        ok_rays = torch.equal(gathered, full * 2.0)
        # --- evaluation render of one frame split over the ranks: every rank gets the whole frame back, in ray order
        class Stub:
            calls = []

            def render(self, o, d, time, cal_lidar_color=False, staged=False, max_ray_batch=4096, **kw):
                assert staged and o.shape[0] == 1 and o.shape == d.shape
                Stub.calls.append(o.shape[1])
                k = ("depth_lidar", "image_lidar") if cal_lidar_color else ("depth", "image")
                return {k[0]: o[..., 0] + d[..., 1], k[1]: torch.stack([o[..., 0], d[..., 2]], -1) * float(time)}
        o = torch.arange(N * 3, dtype=torch.float32).view(1, N, 3)
        d = -o
        out = FS.render_sharded(Stub(), o, d, torch.tensor([[0.5]]), cal_lidar_color=True)
        ok_rays = ok_rays and Stub.calls == [e - b] and torch.equal(out["depth_lidar"], o[..., 0] + d[..., 1]) \
            and torch.equal(out["image_lidar"], torch.stack([o[..., 0], d[..., 2]], -1) * 0.5)
        # --- evaluation statistics: per-rank sums through ONE all-reduce (the reference's dist.all_reduce(loss), trainer.py:1508)
        sums = FS.allreduce_sums([1.5 * (rank + 1), 10.0 + rank, 3.0])
        ok_rays = ok_rays and sums == [4.5, 21.0, 6.0]
        # --- frames
        mine = FS.frames_for_rank(61, epoch=3, rank=rank, world_size=world_size, seed=5)
        objs = [None] * world_size
        dist.all_gather_object(objs, mine)
        q.put((rank, ok_grad, n_coll, ok_rays, objs))
    finally:
        dist.destroy_process_group()


def test_frame_sharding_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_grad, n_coll, ok_rays, objs in results:
        assert ok_grad and ok_rays
        assert n_coll >= 3  # 4 KiB buckets force several collectives; both ranks issue the same number
        a, b = objs
        assert len(a) == len(b) == 31                      # padded so both ranks run the same number of steps
        assert set(a) | set(b) == set(range(61))           # every frame is rendered ...
        assert len(set(a) & set(b)) <= 1                   # ... once (one wrap-around duplicate pads the odd epoch)
    assert results[0][2] == results[1][2]


def test_single_process_paths_are_no_ops():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "selfsupervised-nvsf_amd"))
    from nvsf import frame_shard as FS
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    assert FS.allreduce_gradients([p]) == 0 and torch.equal(p.grad, torch.ones(3))
    assert FS.ray_chunk(10, 0, 1) == (0, 10)
    assert [FS.ray_chunk(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sorted(FS.frames_for_rank(9, 0, 0, 1)) == list(range(9))
    assert FS.frames_for_rank(9, 1, 0, 2, drop_last=True) + FS.frames_for_rank(9, 1, 1, 2, drop_last=True) != []
    x = torch.randn(5, 2)
    assert FS.gather_ray_outputs(x, 5) is x
    assert FS.allreduce_sums([1.25, 2]) == [1.25, 2.0]
