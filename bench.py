#!/usr/bin/env python3
"""Benchmark of the NVSF render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one of the ranks.  Without one (`python bench.py --gpus N`)
this process only starts the N ranks itself -- as children of `torch.distributed.run`, BEFORE anything here touches the GPU --
forwards rank 0's JSON line and exits with the launcher's status.

Workload (BASELINE.json configs[1], "C2"): per step and per GPU one KITTI-360-shaped frame =
4096 LiDAR rays + 4096 camera rays, 768 uniform samples per ray, static hash grid L=16 F=2 T=2^19
(base 16 -> 2048), sigma MLP 32->64->16, LiDAR heads 2 x (87->64->64->1), colour head 31->64->64->3,
freshly initialised parameters (tcnn init: every sample passes the w > 1e-4 colour mask, i.e. no work is
skipped), synthetic rays resident in HBM.  A step renders both ray batches (forward render, fp16 field /
fp32 compositing).  Frames are independent: with N GPUs each rank renders its own frame, no data-path
collective (weak scaling); value = total rays / max-over-ranks time.

stdout of rank 0 is ONE compact JSON line (<= FINAL_LINE_MAX_BYTES; `compact_line`): the contract keys, `roofline`, `cpu_baseline`
and one-number summaries of the legs.  Everything else the legs measured -- per-kernel rows, per-parameter errors, notes -- goes to
gpurun_out/bench_detail.json (`write_detail`), never to stdout (round 4's 30 KB line was not parsed by the driver).  Keys of the detail:
  roofline      the dominant launch of the step (the hash-grid encode pass of the camera batch), timed live with HIP
                events on the launch stream; achieved = algorithmic B/sample (512 gathered + what the
                launch writes) x samples / duration; traffic = 2*FETCH_SIZE + WRITE_SIZE of the committed PMC passes
  kernels       the same figures for every launch of the step
  occupancy / dynamic / train   secondary legs (BASELINE configs 3, 5, 4), never `value`
  raymarching   the raymarching-extension kernels (rows a1-a9) against the HBM roofline at non-latency-bound sizes
  field_ops     the stand-alone field operators (rows a12-a17) against their rooflines on the config-2 sample batches
  cpu_baseline  the CPU oracle (oracle/, scalar C port) on a bounded sample of the same workload: one core, and the rays
                split over a thread pool on all host cores (`cores` = threads used; `one_core` beside it); `torch_cpu` = the
                vectorised PyTorch-CPU restatement (oracle/torch_cpu_path.py), all threads and one thread
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ.setdefault("TORCH_CPP_LOG_LEVEL", "ERROR")  # the train leg's kernel trace (torch.profiler) is chatty on stderr
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore", message=".*Profiler clears events.*")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-ms", type=float, default=500.0, help="untimed device spin-up before the W warm-up steps: the same step repeated for "
                    "this long, so that the clocks have ramped (they take ~50 steps = 40 ms after an idle gap) whatever W is; 0 = none")
    ap.add_argument("--num-rays", type=int, default=4096)
    ap.add_argument("--num-rays-lidar", type=int, default=4096)
    ap.add_argument("--num-steps", type=int, default=768)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the multi-core cpu_baseline leg (0 = the host's cores, at most 16; 1 = skip it)")
    ap.add_argument("--cpu-rays", type=int, default=320, help="rays per modality in the cpu_baseline sample (~10 s of CPU work; 0 = skip)")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the occupancy-grid (config 3) and dynamic-field (config 5) legs")
    ap.add_argument("--train-steps", type=int, default=40, help="extra leg: timed training steps reported under `train` (0 = skip)")
    return ap.parse_args()


def launch_command(gpus, argv, port):
    """The launcher invocation `python bench.py --gpus N` turns into when no launcher started it."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """--gpus N > 1 without WORLD_SIZE: start the N ranks as fresh child processes (this parent has not initialised HIP and
    never does), pass their stderr through, print rank 0's JSON line and return the launcher's exit status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's (and torch's)
    # cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it; it is pinned here so
    # that ranks started from a scrubbed environment still get it (an explicit setting of the caller wins).
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(launch_command(args.gpus, sys.argv[1:], port), stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in proc.stdout.splitlines():
        if l not in lines[-1:]:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def event_time_ms(fn, iters):
    """Average duration of fn() over `iters` back-to-back launches, HIP events on the current stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def kernel_breakdown(model, batches, T, iters):
    """Times each launch of the step IN the step's own launch sequence (LiDAR render, camera encode pass, camera tail, repeated
    `iters` times with HIP events between the launches on the launch stream) and prices it against its roofline.  A launch timed
    by repeating it alone back to back reads 5-15 % longer than it runs inside the step (what rocprofv3's kernel trace of the step
    loop shows): consecutive launches of one kernel contend for the same table lines in the same phase."""
    from nvsf import field_ops as ops
    stages = []  # (label, callable, row builder)
    for name, (o, d, lidar) in batches.items():
        N = o.shape[0]
        M = N * T
        enc = model.hash_encoder_lidar if lidar else model.hash_encoder_camera
        if lidar:
            nears = torch.full((N,), float(model.min_near_lidar), device=o.device)
            fars = torch.full((N,), float(model.lidar_max_depth), device=o.device)
        else:
            from nvsf.nerf.raymarching import raymarching
            nears, fars = raymarching.near_far_from_aabb(o, d, model.aabb_infer, model.min_near)
        ray_length = float(model.lidar_max_depth - model.min_near_lidar) if lidar else 2.0 * float(model.bound)
        sliced = ops.prefer_sliced(enc.spec, N, T, ray_length, float(model.bound))  # the choice model.render makes
        dargs = (o, d, nears, fars, T, model._aabb_host, float(model.bound), enc.table_f16(), enc.spec, model.sigma_net.weights_f16())
        if lidar:
            head_a, head_b, head_flops, bg = model.raydrop_net.weights_f16(), model.intensity_net.weights_f16(), 2 * 22528, None
        else:
            head_a, head_b, head_flops, bg = model.color_net.weights_f16(), None, 14336, [1.0, 1.0, 1.0]
        sigma_flops = 2 * (32 * 64 + 64 * 16)
        rargs = dargs + (lidar, head_a, head_b, model._k_scale(), bg)
        z, w, ws, dp, img = ops.render_uniform(*rargs, sliced=sliced)
        active = float((w > ops.W_THRESH).float().mean())
        flops = sigma_flops + head_flops * active
        if sliced:
            bufs = ops.render_uniform(*rargs, sliced=True, _stage="encode")
            stages.append((lambda rargs=rargs, bufs=bufs: ops.render_uniform(*rargs, sliced=True, _stage="encode", _buffers=bufs),
                           lambda t, name=name, M=M: dict(kernel=f"density_encode_sliced[{name}]", ms=t, bound="hbm", unit="GB/s", achieved=580.0 * M / t / 1e6,
                                                          peak=HBM_PEAK_GBS, per_unit="580 B/sample (512 gathered + 64 features + 4 z written)", units=M)))
            stages.append((lambda rargs=rargs, bufs=bufs: ops.render_uniform(*rargs, sliced=True, _stage="tail", _buffers=bufs),
                           lambda t, name=name, M=M, flops=flops, active=active, head_flops=head_flops: dict(
                               kernel=f"render_uniform_tail[{name}]", ms=t, bound="mfma", unit="TFLOP/s", achieved=flops * M / t / 1e9, peak=MFMA_PEAK_TFLOPS, units=M,
                               per_unit=f"{sigma_flops} + {head_flops} x active fraction {active:.3f} FLOP/sample (sigma MLP, compositing, heads); "
                                        "76 B/sample of HBM traffic (64 features + 4 z read, 4 weights written)")))
        else:
            stages.append((lambda rargs=rargs: ops.render_uniform(*rargs, sliced=False),
                           lambda t, name=name, M=M, flops=flops: dict(
                               kernel=f"render_uniform[{name}]", ms=t, bound="hbm", unit="GB/s", achieved=520.0 * M / t / 1e6, peak=HBM_PEAK_GBS, units=M,
                               per_unit="520 B/sample (512 gathered + 4 z + 4 weights written); gather, sigma MLP, compositing and heads "
                                        f"in one launch ({flops:.0f} FLOP/sample = {flops * M / t / 1e9:.0f} TFLOP/s)")))
    n = len(stages)
    for _ in range(10):
        for fn, _ in stages:
            fn()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] for _ in range(iters)]
    for it in range(iters):
        evs[it][0].record()
        for k, (fn, _) in enumerate(stages):
            fn()
            evs[it][k + 1].record()
    torch.cuda.synchronize()
    rows = []
    for k, (_, row) in enumerate(stages):
        t = float(np.median([evs[it][k].elapsed_time(evs[it][k + 1]) for it in range(iters)]))
        rows.append(row(t))
    for r in rows:
        r["frac"] = r["achieved"] / r["peak"]
    return rows


def _running_render_digest(nvsf_build):
    """Digest of the render kernels' sources the MAPPED library was built from (nvsf_build_digest(1)); without a HIP device
    (the CPU contract test) nothing is mapped for compute and the digest of the sources on disk stands in."""
    try:
        from nvsf import _hip
        return _hip.build_digest(render_only=True)
    except Exception:
        return nvsf_build.csrc_digest()


def pmc_traffic(kernel_label):
    """HBM-side bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json, produced by tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command; counters cannot be collected from inside the timed process).  gfx950 correction: FETCH_SIZE
    counts 64 B per 128-B request, i.e. half the bytes (verified on k_weights_fwd: 12.4 MB reported for 25.2 MB of
    coalesced reads), so bytes = 2 * FETCH_SIZE + WRITE_SIZE.  The profile records the digest of the kernel sources it was
    taken on (build.csrc_digest); a profile of other sources is reported as stale, not used."""
    import glob
    import build as nvsf_build
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return {"traffic": None}
    prof = json.load(open(files[-1]))
    running = _running_render_digest(nvsf_build)
    if prof.get("csrc_digest") != running:
        # counters of another build of the kernels say nothing about this one: no figure rather than a stale one
        return {"traffic": None, "traffic_stale": f"{os.path.relpath(files[-1], ROOT)} was taken on kernel sources {prof.get('csrc_digest', '(unrecorded)')}, "
                                                  f"this build is {running}"}
    ks = prof["kernels"]
    which = "camera" if "camera" in kernel_label else "lidar"
    src = os.path.relpath(files[-1], ROOT) + " (2*FETCH_SIZE + WRITE_SIZE, bytes per launch)"
    base = kernel_label.split("[")[0]
    names = {"density_encode_sliced": ["encode_sliced"], "render_uniform": [f"render_uniform<{which}>"],
             "render_uniform_tail": [f"render_uniform_tail<{which}>"], "density_uniform": ["density_uniform_v2", "density_uniform"],
             "density_from_features": ["density_from_features"], "heads_uniform": [f"heads_uniform<{which}>"],
             "composite_weights": ["k_weights_fwd"]}.get(base, [])
    for name in names:
        k = ks.get(name)
        if k:
            f, w = k.get(f"fetch_kb_{which}", k["fetch_kb_mean"]), k.get(f"write_kb_{which}", k["write_kb_mean"])
            return {"traffic": (2.0 * f + w) * 1024.0, "traffic_source": src}
    return {"traffic": None}


def cpu_baseline(model, T, rays, n_rays, threads=1):
    """Times the scalar CPU oracle (an unvectorised C restatement, oracle/*.c) on the first `n_rays` LiDAR and the first `n_rays`
    camera rays of the timed batches (`rays` = {True: (o, d), False: (o, d)} numpy).  threads > 1: the rays of each batch are
    split into `threads` contiguous chunks rendered by a thread pool (rays are independent; the C oracle is called through
    ctypes, which releases the GIL; the numpy glue between the calls holds it).  Returns (figures, oracle outputs per modality):
    the outputs are what `outputs_match_oracle` compares the GPU's render of the same rays with."""
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    from nvsf import synthetic as S
    f16 = lambda net: net.params.detach().cpu().numpy().astype(np.float16)
    lin = torch.linspace(0.0, 1.0, T).numpy()
    aabb = np.array([-S.BOUND] * 3 + [S.BOUND] * 3, np.float32)
    threads = max(1, min(int(threads), n_rays))
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    outputs = {}
    t0 = time.perf_counter()
    for lidar in (True, False):
        o, d = (a[:n_rays] for a in rays[lidar])
        enc = model.hash_encoder_lidar if lidar else model.hash_encoder_camera
        table = enc.params.detach().cpu().numpy().astype(np.float16)
        if lidar:
            nears, fars = np.full(n_rays, model.min_near_lidar, np.float32), np.full(n_rays, model.lidar_max_depth, np.float32)
        else:
            nears, fars = O.near_far_from_aabb(o, d, aabb, model.min_near)
        w_sigma = f16(model.sigma_net)
        w_a = f16(model.raydrop_net) if lidar else f16(model.color_net)
        w_b = f16(model.intensity_net) if lidar else None

        def chunk(lo_hi):
            a, b = lo_hi
            r = O.render_static(o[a:b], d[a:b], nears[a:b], fars[a:b], lin, None, float(S.BOUND), table, enc.spec, w_sigma, lidar,
                                w_a, w_b, np.ones(3, np.float32), k_scale=model._k_scale())
            return r["image"], r["depth"], r["weights_sum"]
        edges = np.linspace(0, n_rays, threads + 1).astype(int)
        spans = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]
        parts = [chunk(spans[0])] if pool is None else list(pool.map(chunk, spans))
        outputs[lidar] = tuple(np.concatenate([p[i] for p in parts], 0) for i in range(3))
    dt = time.perf_counter() - t0
    if pool is not None:
        pool.shutdown()
    return ({"value": 2 * n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
             "sample": f"the first {n_rays} LiDAR + {n_rays} camera rays of the timed batches x {T} samples, same field; scalar (unvectorised) C "
                       f"restatement of the path (oracle/*.c) + numpy glue, {threads} thread(s), {dt:.1f} s wall"}, outputs)


def torch_cpu_baseline(model, T, rays, n_rays):
    """The vectorised PyTorch-CPU restatement of the path (oracle/torch_cpu_path.py: the reference's own tensor algebra around torch
    formulations of the tiny-cuda-nn operators -- what BASELINE.md section 3 planned as the CPU figure) on the first `n_rays` rays per
    modality of the timed batches: all host threads, then one thread on an eighth of the sample (~20 s of CPU work in all)."""
    import importlib.util
    from nvsf import synthetic as S
    spec_ = importlib.util.spec_from_file_location("torch_cpu_path", os.path.join(ROOT, "oracle", "torch_cpu_path.py"))
    TP = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(TP)
    import oracle_lib as O
    f16 = lambda net: net.params.detach().cpu().to(torch.float16)
    aabb = np.array([-S.BOUND] * 3 + [S.BOUND] * 3, np.float32)
    prev = torch.get_num_threads()

    def run(n, threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        with torch.no_grad():
            for lidar in (True, False):
                o, d = (torch.from_numpy(a[:n]) for a in rays[lidar])
                enc = model.hash_encoder_lidar if lidar else model.hash_encoder_camera
                if lidar:
                    nears, fars = torch.full((n,), float(model.min_near_lidar)), torch.full((n,), float(model.lidar_max_depth))
                else:
                    nears, fars = (torch.from_numpy(v) for v in O.near_far_from_aabb(rays[lidar][0][:n], rays[lidar][1][:n], aabb, model.min_near))
                TP.render_static(o, d, nears, fars, T, float(S.BOUND), f16(enc), enc.spec, f16(model.sigma_net), model.sigma_net.spec, lidar,
                                 f16(model.raydrop_net if lidar else model.color_net), f16(model.intensity_net) if lidar else None,
                                 (model.raydrop_net if lidar else model.color_net).spec, k_scale=model._k_scale())
        return time.perf_counter() - t0
    try:
        cores = min(os.cpu_count() or 1, 16)
        run(min(64, n_rays), cores)  # warm-up (thread pool, allocator)
        dt_all = run(n_rays, cores)
        n1 = max(32, n_rays // 8)
        dt_one = run(n1, 1)
    finally:
        torch.set_num_threads(prev)
    return {"value": 2 * n_rays / dt_all, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"the first {n_rays} LiDAR + {n_rays} camera rays of the timed batches x {T} samples, vectorised PyTorch-CPU restatement "
                      f"(oracle/torch_cpu_path.py), {cores} threads, {dt_all:.1f} s wall",
            "one_core": {"value": 2 * n1 / dt_one, "sample": f"{n1} + {n1} rays, 1 thread, {dt_one:.1f} s wall"}}


def outputs_match_oracle(out, oracle_outputs, tol=1e-4):
    """The render the timed loop produced (its last step) against the CPU oracle on the same rays: max |error| of the composited
    image / depth / weights_sum per modality (north_star: 1e-4 abs)."""
    res = {"tolerance": tol, "ok": True}
    for lidar, (img, dep, ws) in oracle_outputs.items():
        sfx = "_lidar" if lidar else ""
        n = img.shape[0]
        errs = {"image" + sfx: float(np.abs(out[0 if lidar else 1]["image" + sfx][0, :n].cpu().numpy() - img).max()),
                "depth" + sfx: float(np.abs(out[0 if lidar else 1]["depth" + sfx][0, :n].cpu().numpy() - dep).max()),
                "weights_sum" + sfx: float(np.abs(out[0 if lidar else 1]["weights_sum" + sfx][:n].cpu().numpy() - ws).max())}
        res.update({"max_abs_err_" + k: v for k, v in errs.items()})
        res["checked_rays" + sfx] = n
        res["ok"] = bool(res["ok"] and all(np.isfinite(v) and v <= tol for v in errs.values()))
    return res


def train_kernel_rows(step, batch, M, n_steps=3):
    """Per-kernel rows of the training step as it runs (side-stream scatters beside the MLP backward included): kernel trace of
    `n_steps` steps through torch.profiler (kineto / roctracer), average duration per launch, launches per step, and for the kernels
    whose algorithmic work is known a roofline fraction -- the fused MLP kernels against the dense fp16 MFMA peak (FLOP counted from
    the template arguments <IN_STEPS, N_HIDDEN>: in_cols = 32 IN_STEPS, 64-wide hidden layers, 16 outputs; backward = 3 x forward:
    recomputed forward + data + weight gradients), the streaming kernels against HBM.  Not run under rocprofv3 (two tracers in one
    process): the committed profiles/*_train_kernel_stats.csv + *_pmc_train.json are that view, with counters."""
    import re
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"skipped": "running under rocprofv3"}
    try:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(n_steps):
                step.step(batch)
            torch.cuda.synchronize()
        per = {}
        for ev in prof.events():
            if ev.device_type is not None and "cuda" in str(ev.device_type).lower() and ev.device_time > 0:
                d = per.setdefault(ev.name, [0, 0.0])
                d[0] += 1
                d[1] += ev.device_time  # us
    except Exception as e:  # noqa: BLE001 -- a secondary figure: never fail the bench line for it
        return {"skipped": f"{type(e).__name__}: {e}"}
    if not per:
        return {"skipped": "the profiler returned no device events"}

    def mlp_flops(name):
        m = re.search(r"k_mlp_(fwd|bwd)(?:_wave)?(?:<|ILi)(\d+)(?:, |ELi)(\d+)", name)
        if not m:
            return None
        in_cols, n_hidden = 32 * int(m.group(2)), int(m.group(3))
        f = 2 * (64 * in_cols + 64 * 64 * (n_hidden - 1) + 16 * 64)
        return f * (3 if m.group(1) == "bwd" else 1)

    streams = {"k_render_uniform": (520 + 128, "520 + 128 B/sample (gathers, z, weights; positions, feature rows, geometry rows, sigma, colours kept for the "
                                               "backward); sigma MLP + compositing + both heads in the same launch (51 200 FLOP/sample)"),
               "k_density_uniform_v2": (588 + 140, "588 + 140 B/sample (gathers, features, training outputs)"),
               "k_encode_sliced_pairs": (580, "580 B/sample"), "k_density_from_features": (76 + 140, "216 B/sample"),
               "k_sigma_geo_bwd": (132, "132 B/sample"), "k_weights_fwd": (12, "12 B/sample"), "k_weights_bwd": (24, "24 B/sample"),
               "k_image_fwd": (16, "16 B/sample"), "k_image_bwd": (28, "28 B/sample"), "k_masked_sigmoid": (16, "16 B/sample"), "k_sigmoid_bwd": (24, "24 B/sample")}
    total_us = sum(v[1] for v in per.values())
    ours, glue_launches, glue_us = [], 0, 0.0
    for name, (calls, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        short = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", name).split("(")[0][:72]
        if not re.search(r"(?<![A-Za-z])\d*k_[a-z]", name) or name.startswith("void at::"):
            glue_launches += calls
            glue_us += us
            continue
        ms = us / calls / 1e3
        row = {"kernel": short, "launches_per_step": calls / n_steps, "ms": ms, "share_of_kernel_time": us / total_us}
        fl = mlp_flops(name)
        if fl is None and "k_render_tail2" in name:  # camera batch: sigma MLP + colour head on the feature planes of the encode pass
            fl = 6144 + 14336
        if fl is not None:
            tf = fl * M / (ms * 1e-3) / 1e12
            row.update({"bound": "mfma", "unit": "TFLOP/s", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "frac": tf / MFMA_PEAK_TFLOPS, "per_unit": f"{fl} FLOP/sample"})
        else:
            for key, (b, note) in streams.items():
                if key in name:
                    gbs = b * M / (ms * 1e-3) / 1e9
                    row.update({"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS, "per_unit": note})
                    break
        ours.append(row)
    return {"steps_traced": n_steps, "kernel_ms_per_step": total_us / n_steps / 1e3, "launches_per_step": sum(v[0] for v in per.values()) / n_steps,
            "glue": {"what": "torch / rocclr launches (fills, copies, elementwise, random numbers)", "launches_per_step": glue_launches / n_steps,
                     "ms_per_step": glue_us / n_steps / 1e3},
            "rows": ours[:24], "units": M,
            "note": "durations are device times inside the step (the table scatters run on a side stream beside the MLP backward, so the sum "
                    "exceeds ms_per_step); MFMA fractions from algorithmic FLOP / duration -- the counter view is profiles/*_pmc_train.json"}


def train_grads_match(model, batch, T, scale, reference=None, what=None, init_scale=128.0):
    """The check of tests/test_config4_full_size_gpu.py / test_config5_train_full_size_gpu.py on the bench's own batch: one forward +
    backward of the production plan (binned scatter, level-major gradient, side stream) against the reference formulations
    (`reference`: nvsf.testing variants; default: every level of every table through nvsf_hashgrid_bwd), same jitter; max |difference|
    per parameter relative to that parameter's largest gradient entry."""
    from nvsf import testing
    from nvsf.nerf.train_step import RenderTrainStep
    from nvsf.nerf.loss_scaler import LossScaler
    reference = reference or {"table_scatter": "atomic"}
    step = RenderTrainStep(model, num_steps=T, scale=scale, ema_decay=None)
    step.scaler = LossScaler(init_scale=float(init_scale))  # a loss scale at which the fp16 gradients of this model are finite
    out = {}
    for name, variants in (("production", {}), ("reference", reference)):
        torch.manual_seed(17)
        with testing.variant(**variants):
            step.forward_backward(batch)
        torch.cuda.synchronize()
        out[name] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    errs = {}
    for n, a in out["reference"].items():
        b = out["production"].get(n)
        scale_n = float(a.abs().max())
        errs[n] = float("inf") if (b is None or scale_n == 0.0 or not bool(torch.isfinite(b).all())) else float((a - b).abs().max()) / scale_n
    tol = 5e-5
    del step, out
    return {"ok": bool(errs) and all(v <= tol for v in errs.values()), "tolerance": tol, "max_rel_err": errs, "parameters": len(errs),
            "reference": reference, "loss_scale": float(init_scale),
            "what": what or "production table scatter (bins + level-major hand-over) against the atomic variant, every parameter, one full-size step"}


def train_leg(model, tl, tc, tm, T, steps, dev, dist):
    """Secondary figure (not `value`): full multimodal training steps (BASELINE config 4 shape) -- both renders with
    gradient, losses, backward through the HIP operators, one bucketed RCCL gradient all-reduce when N > 1, Adam."""
    from nvsf.nerf.train_step import RenderTrainStep
    from nvsf.synthetic import SCALE as S_SCALE
    n_l, n_c = tl[0].shape[1], tc[0].shape[1]
    g = torch.Generator(device="cpu").manual_seed(3)
    batch = {"rays_o_lidar": tl[0], "rays_d_lidar": tl[1], "rays_o": tc[0], "rays_d": tc[1], "time": tm,
             "gt_depth": torch.rand(1, n_l, generator=g).to(dev) * 0.5, "gt_raydrop": (torch.rand(1, n_l, generator=g) > 0.3).float().to(dev),
             "gt_intensity": torch.rand(1, n_l, generator=g).to(dev), "gt_rgb": torch.rand(1, n_c, 3, generator=g).to(dev)}
    step = RenderTrainStep(model, num_steps=T, scale=S_SCALE)
    n_coll = 0
    t_spin, n_spin = time.perf_counter(), 0
    while n_spin < 5 or (time.perf_counter() - t_spin) < 0.3:  # the loss scale settles (overflowing first steps are skipped), optimiser
        step.step(batch)                                       # state and allocator pools exist, the clocks are up again after the
        n_spin += 1                                            # host-side legs before this one (as --spinup-ms for the headline)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, n_coll = step.step(batch)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    world = dist.get_world_size() if dist is not None else 1
    per_rank_ms = [dt / steps * 1e3]
    allreduce = None
    if dist is not None:
        every = [None] * world
        dist.all_gather_object(every, dt)
        per_rank_ms = [float(v) / steps * 1e3 for v in every]
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if step.buckets is not None:  # the LAST step's bucket all-reduces, device time on the communication stream (this rank)
            rows = step.buckets.allreduce_ms()
            allreduce = {"payload_MB": step.buckets.payload_bytes / 1e6, "buckets": len(step.buckets.flat),
                         "per_bucket": [{"bucket": b, "MB": mb, "ms": ms} for b, mb, ms in rows], "sum_ms": sum(r[2] for r in rows),
                         "ring_estimate_ms": 2.0 * (world - 1) / world * step.buckets.payload_bytes / 153e9 * 1e3,
                         "ring_estimate": "2 (W - 1) / W x payload / 153 GB/s (one xGMI link per ring direction; DESIGN.md section 7)"}
    kernels = train_kernel_rows(step, batch, n_l * T) if (dist is None or dist.get_rank() == 0) else None
    grads_match = train_grads_match(model, batch, T, S_SCALE) if world == 1 else None
    model.eval()
    return {"metric": "trained rays/sec (LiDAR+cam, fwd+bwd+Adam)", "value": (n_l + n_c) * world * steps / dt, "ms_per_step": dt / steps * 1e3,
            "kernels": kernels, "grads_match": grads_match,
            "steps": steps, "untimed_steps": n_spin, "allreduce_collectives_per_step": n_coll, "per_rank_ms_per_step": per_rank_ms, "allreduce": allreduce,
            "losses": "the reference's Trainer.train_step defaults: per-ray L1 range + MSE ray-drop + MSE intensity summed over rays, chamfer distance of the predicted point cloud, summed MSE RGB",
            "path": "training forward of a ray batch as ONE autograd node (the evaluation render's kernels in TRAIN form: one launch for LiDAR, "
                    "level-sliced encode + streaming tail for camera); backward: image / sigmoid / heads (wave-independent fused MLP backward) / "
                    "compositor / density MLP kernels, table scatter on a side stream (levels 8-15: run sums through bins summed in LDS; levels 0-7: "
                    "corner-parallel run-merging atomics); loss scaling with the overflow decision taken before the last scatter has finished, that "
                    "table's Adam pass behind its scatter (nvsf/nerf/loss_scaler.py)"}


def eval_leg(model, dev, T, frames, dist):
    """Secondary figure: the reference's evaluation workload -- whole frames, LiDAR 66 x 1030 = 67 980 rays and
    camera 376 x 1408 = 529 408 rays, generated on the device (csrc/raygen.hip) and rendered with the staged chunk loop
    (max_ray_batch 4096).  Across ranks a frame's rays are split into contiguous chunks and the outputs all-gathered
    (frame_shard.render_sharded, what trainer.py:1511-1524 intended): fixed work per frame, i.e. STRONG scaling."""
    from nvsf import frame_shard, synthetic as S
    from nvsf.nerf.dataset import dataset_utils as DU
    pose = torch.eye(4, device=dev)[None]
    K = torch.tensor([[S.CAM_K[0], 0.0, S.CAM_K[2]], [0.0, S.CAM_K[1], S.CAM_K[3]], [0.0, 0.0, 1.0]])
    lid = DU.get_lidar_rays(pose, S.LIDAR_FOV[:2], (0.0, S.LIDAR_FOV[2]), S.LIDAR_HW[0], S.LIDAR_HW[1])
    cam = DU.get_rays(pose, K, S.CAM_HW[0], S.CAM_HW[1])
    tm = torch.tensor([[0.5]], device=dev)

    def frame():
        with torch.no_grad():
            a = frame_shard.render_sharded(model, lid["rays_o"], lid["rays_d"], tm, cal_lidar_color=True, num_steps=T)
            b = frame_shard.render_sharded(model, cam["rays_o"], cam["rays_d"], tm, cal_lidar_color=False, num_steps=T)
        return a, b
    out = frame()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = frame()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    world = dist.get_world_size() if dist is not None else 1
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n = lid["rays_o"].shape[1] + cam["rays_o"].shape[1]
    ok = bool(out[0]["image_lidar"].shape[1] == lid["rays_o"].shape[1] and out[1]["image"].shape[1] == cam["rays_o"].shape[1]
              and torch.isfinite(out[0]["depth_lidar"]).all() and torch.isfinite(out[1]["image"]).all())
    return {"metric": "evaluation frames/sec (LiDAR 66x1030 + camera 376x1408 rays per frame, staged)", "value": frames / dt,
            "rays_per_s": n * frames / dt, "ms_per_frame": dt / frames * 1e3, "frames": frames, "rays_per_frame": n, "scaling": "strong",
            "ranks": world, "collective": "all_gather of depth / image chunks" if world > 1 else "none", "outputs_complete_and_finite": ok}


def occupancy_outputs_match_oracle(m, grid, lidar_rays, camera_rays, tensors, n_check, T_thresh=1e-4, max_steps=1024):
    """Config 3's parity field: the one-launch occupancy render of the TIMED batches against the CPU oracle's
    survivor loop (oracle/: march_rays -> field -> composite_rays, raymarching.cu:808-1053) on the first `n_check` rays of each
    batch.  Rule of tests/test_occupancy_gpu.py: 1e-4 abs on every output, except that a ray whose transmittance lands within float
    noise of T_thresh may take ONE more sample on one side -- bounded by that sample's weight (<= 1.05 T_thresh on weights_sum and
    image, x t_max on depth); at least 97 % of the rays must meet the plain 1e-4."""
    import oracle_lib as O
    from nvsf import synthetic as S
    tl, tc, tm = tensors
    bits = O.packbits(grid, 0.5)
    f16 = lambda net: net.params.detach().cpu().numpy().astype(np.float16)
    res = {"tolerance": 1e-4, "terminal_sample_allowance": 1.05 * T_thresh, "ok": True}
    with torch.no_grad():
        outs = {True: m.render(tl[0], tl[1], tm, cal_lidar_color=True, max_steps=max_steps, T_thresh=T_thresh, fused=True),
                False: m.render(tc[0], tc[1], tm, cal_lidar_color=False, max_steps=max_steps, T_thresh=T_thresh, fused=True)}
    for lidar, (o, d) in ((True, lidar_rays), (False, camera_rays)):
        o, d = o[:n_check], d[:n_check]
        enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
        field = (enc.params.detach().cpu().numpy().astype(np.float16), enc.spec, f16(m.sigma_net),
                 f16(m.raydrop_net) if lidar else f16(m.color_net), f16(m.intensity_net) if lidar else None)
        if lidar:
            nears, fars = np.full(len(o), m.min_near_lidar, np.float32), np.full(len(o), m.lidar_max_depth, np.float32)
        else:
            nears, fars = O.near_far_from_aabb(o, d, np.array([-S.BOUND] * 3 + [S.BOUND] * 3, np.float32), m.min_near)
        ref = O.render_occupancy_infer(o, d, nears, fars, bits, float(S.BOUND), m.cascade, m.grid_size, max_steps, 0.0, field, lidar, T_thresh=T_thresh)
        sfx = "_lidar" if lidar else ""
        out = outs[lidar]
        e_ws = np.abs(out["weights_sum" + sfx].cpu().numpy()[:n_check] - ref["weights_sum"])
        e_dp = np.abs(out["depth" + sfx][0].cpu().numpy()[:n_check] - ref["depth"])
        e_im = np.abs(out["image" + sfx][0].cpu().numpy()[:n_check] - ref["image"]).max(-1)
        exact = (e_ws <= 1e-4) & (e_dp <= 1e-4) & (e_im <= 1e-4)
        t_max = float(fars[fars < 1e30].max()) if (fars < 1e30).any() else 0.0
        w1 = 1.05 * T_thresh
        within = bool((e_ws <= 1e-4 + w1).all() and (e_im <= 1e-4 + w1).all() and (e_dp <= 1e-4 + w1 * max(t_max, 1.0)).all())
        res["checked_rays" + sfx] = int(len(o))
        res["exact_fraction" + sfx] = float(exact.mean())
        res["max_abs_err" + sfx] = float(max(e_ws[exact].max(initial=0.0), e_dp[exact].max(initial=0.0), e_im[exact].max(initial=0.0)))
        res["max_abs_err_terminal_rays" + sfx] = float(max(e_ws.max(initial=0.0), e_im.max(initial=0.0)))
        res["largest_weights_sum" + sfx] = float(ref["weights_sum"].max())
        res["ok"] = bool(res["ok"] and within and exact.mean() >= 0.97)
    return res


def occupancy_leg(model_cls, dev, n_rays, steps, n_check=128):
    """Secondary figure (BASELINE config 3): the config-2 field and ray batches through the occupancy-grid renderer,
    procedural occupancy grid (union of 64 random boxes, ~10 % occupied), max 1024 samples per ray.
    eval = the one-launch fused kernel; eval_host_loop = the reference's protocol (march_rays -> field -> composite_rays
    per survivor round, one device->host sync each); train = march_rays_train -> field -> composite_rays_train forward."""
    from nvsf import synthetic as S
    torch.manual_seed(0)
    m = model_cls(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES)
    m = m.to(dev).enable_occupancy_grid().to(dev)
    rng = np.random.default_rng(0)
    grid = S.boxes_density_grid(rng, cascades=m.cascade, H=m.grid_size, n_boxes=64)
    m.set_density_grid(torch.from_numpy(grid).to(dev), thresh=0.5)
    lo, ld = S.lidar_rays(n_rays, rng)
    co, cd = S.camera_rays(n_rays, rng)
    tl = [torch.from_numpy(a).to(dev)[None] for a in (lo, ld)]
    tc = [torch.from_numpy(a).to(dev)[None] for a in (co, cd)]
    tm = torch.tensor([[0.5]], device=dev)

    def timed(fused, k):
        def step():
            with torch.no_grad():
                m.render(tl[0], tl[1], tm, cal_lidar_color=True, max_steps=1024, fused=fused)
                m.render(tc[0], tc[1], tm, cal_lidar_color=False, max_steps=1024, fused=fused)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    out = {"metric": "rendered rays/sec (LiDAR+cam), occupancy grid + early termination", "max_steps": 1024,
           "occupied_fraction": [float((g > 0.5).mean()) for g in grid]}
    m.eval()
    dt = timed(True, steps)
    out["eval"] = {"value": 2 * n_rays / dt, "ms_per_step": dt * 1e3, "path": "nvsf_render_occupancy_fwd (one launch per batch)"}
    out["outputs_match_oracle"] = occupancy_outputs_match_oracle(m, grid, (lo, ld), (co, cd), (tl, tc, tm), n_check)
    dt = timed(False, max(2, steps // 4))
    out["eval_host_loop"] = {"value": 2 * n_rays / dt, "ms_per_step": dt * 1e3, "path": "march_rays -> field -> composite_rays survivor loop"}
    m.train()
    dt = timed(True, steps)
    out["train_forward"] = {"value": 2 * n_rays / dt, "ms_per_step": dt * 1e3, "path": "march_rays_train -> field -> composite_rays_train",
                            "samples_camera_batch": int(m.step_counter[(m.local_step - 1) % 16][0])}
    return out


def reference_default_grid_leg(model_cls, dev, n_rays, T, steps):
    """Secondary figure: the static field with the reference's DEFAULT hash grid (8 levels x 4 features, 512 -> 32768, T = 2^19,
    main_nvsf.py:45-52) through the same fused render (level-sliced encode pass, one level per XCD, + streaming tail) and the same
    training step as the headline's L16 F2 field.  Never `value`."""
    from nvsf import synthetic as S
    from nvsf.nerf.train_step import RenderTrainStep
    torch.manual_seed(0)
    m = model_cls(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES,
                  n_levels_hash=8, n_features_per_level_hash=4, base_resolution=512, max_resolution=32768, log2_hashmap_size=19).to(dev).eval()
    rng = np.random.default_rng(0)
    lo, ld = S.lidar_rays(n_rays, rng)
    co, cd = S.camera_rays(n_rays, rng)
    tl = [torch.from_numpy(a).to(dev)[None] for a in (lo, ld)]
    tc = [torch.from_numpy(a).to(dev)[None] for a in (co, cd)]
    tm = torch.tensor([[0.5]], device=dev)

    def render():
        with torch.no_grad():
            m.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=T)
            m.render(tc[0], tc[1], tm, cal_lidar_color=False, num_steps=T)
    for _ in range(3):
        render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"metric": "rendered rays/sec (LiDAR+cam), static field with the reference-default grid L8 F4", "value": 2 * n_rays / dt,
           "ms_per_step": dt * 1e3, "path": "k_encode_sliced_f4 + k_render_tail2 per ray batch"}
    g = torch.Generator(device="cpu").manual_seed(3)
    batch = {"rays_o_lidar": tl[0], "rays_d_lidar": tl[1], "rays_o": tc[0], "rays_d": tc[1], "time": tm,
             "gt_depth": torch.rand(1, n_rays, generator=g).to(dev) * 0.5, "gt_raydrop": (torch.rand(1, n_rays, generator=g) > 0.3).float().to(dev),
             "gt_intensity": torch.rand(1, n_rays, generator=g).to(dev), "gt_rgb": torch.rand(1, n_rays, 3, generator=g).to(dev)}
    m.train()
    grads_match = train_grads_match(m, batch, T, S.SCALE, what="binned plan (0, 8) of the L8 F4 grid (fp32 level-major gradient) against fp32 "
                                                                "atomics for every level, every parameter, one full-size step")
    step = RenderTrainStep(m, num_steps=T, scale=S.SCALE)
    for _ in range(3):
        step.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.step(batch)
    step.sync()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out["train"] = {"value": 2 * n_rays / dt, "ms_per_step": dt * 1e3, "path": "RenderRaysFn (one node per ray batch) + binned table scatter + FusedAdam",
                    "grads_match": grads_match}
    return out


def raymarching_leg(dev):
    """Secondary figure: every kernel of the raymarching extension against the HBM roofline at a
    size where the launch is not latency-bound (tools/bench_raymarching.py; at 4096 rays each of them moves < 1 MB).
    march_rays_train is timed on a fully occupied grid (every chain member is a sample: the write-bound case) and on a
    10 % per-cell random grid (a skip every few members: the worst case for the marcher)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_raymarching import raymarching_rooflines
    rows = raymarching_rooflines(dev, occupied=1.0)
    rows += [r for r in raymarching_rooflines(dev, occupied=0.1) if r["kernel"].startswith("march_rays_train")]
    torch.cuda.empty_cache()
    return {"note": "algorithmic bytes (SURVEY 8d) / HIP-event time, HBM peak 8000 GB/s", "kernels": rows}


def field_ops_leg(dev, n_rays, T):
    """Secondary figure: the stand-alone field operators (SURVEY 8a rows a12-a17 -- the tiny-cuda-nn surface, K-planes,
    the space-time grids) against their rooflines on the config-2 sample batches (tools/bench_field_ops.py).  These are
    the launches of the training path and of the dynamic model; the fused render kernels above replace them for the
    static no-grad render."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_field_ops import field_op_rooflines
    return {"note": "algorithmic bytes / flops per sample (SURVEY 8d) / HIP-event time; peaks 8000 GB/s, 2500 TFLOP/s",
            "kernels": field_op_rooflines(dev, n_rays, T)}


def dynamic_fixture_check(m, dev, n_rays, T):
    """Max |image / depth / weights_sum| difference of the fp16-regime render against the reference-generated fixture (24 rays per
    modality inside full-size batches), parameters by the fixture's name-derived seeds; the model's own parameters are restored."""
    path = os.path.join(ROOT, "tests", "golden", "network_dynamic_rd.npz")
    if not os.path.exists(path) or T != 768:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import golden_dynamic as GD
    from nvsf import synthetic as S
    g = np.load(path)
    saved = {k: v.detach().clone() for k, v in m.state_dict().items()}
    try:
        GD.init_by_name(m)
        last = [l for l in m.flow_net.mlp if isinstance(l, torch.nn.Linear)][-1]
        w0 = last.weight.detach().clone()
        worst = {}
        for flow, _ in GD.RD_FLOWS:
            with torch.no_grad():
                last.weight.copy_(w0 * float(g[f"flow_gain_{flow}"]))
            err = 0.0
            for tag, tv in GD.RD_TIMES:
                for lidar in (True, False):
                    o24, d24 = GD.rd_rays(tag, lidar, S)
                    rng = np.random.default_rng(99)
                    o, d = (S.lidar_rays if lidar else S.camera_rays)(n_rays - GD.RD_N, rng)
                    o, d = np.concatenate([o24, o]), np.concatenate([d24, d])
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):  # the Trainer's autocast region (trainer.py:1332)
                        r = m.render(torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None],
                                     torch.tensor([[tv]], dtype=torch.float32, device=dev), cal_lidar_color=lidar, num_steps=T)
                    sfx, key = ("_lidar" if lidar else ""), f"{flow}/{tag}/{'lidar' if lidar else 'cam'}"
                    for k in ("image", "depth", "weights_sum"):
                        ref = g[f"{key}/{k}"]
                        got = r[k + sfx].reshape(n_rays, -1)[:GD.RD_N].cpu().numpy().reshape(ref.shape)
                        err = max(err, float(np.abs(got - ref).max()))
            worst[flow] = err
        return {"max_abs_err": worst, "tolerance": 1e-4, "ok": bool(max(worst.values()) <= 1e-4), "rays_checked_per_render": GD.RD_N,
                "renders": 12, "regime": "fp16 flow MLP (fused MFMA kernel)"}
    finally:
        m.load_state_dict(saved)


def dynamic_leg(dev, n_rays, T, steps):
    """Secondary figure (BASELINE config 5: "dynamic 4D field, fp16 MFMA path"): the reference-default space-time field
    (K-planes + static / dynamic hash grids + flow field, 93.6 M parameters, time_resolution 8), forward render of n_rays
    LiDAR + n_rays camera rays, rendered as the reference's shipped configuration (`fp16 = True`) does: inside an autocast region
    (trainer.py:1332-1334, 1487: `torch.cuda.amp.autocast(enabled=self.fp16)`), where the flow MLP's nn.Linear layers compute in fp16
    -- here the fused fp16 MFMA kernel; the fp32 form (no autocast) is reported beside it.  `outputs_match_fixture`:
    the same model with the fixture's parameters renders batches of the same size whose first 24 rays are the rays of
    tests/golden/network_dynamic_rd.npz -- renders of the REFERENCE's NeRFNetwork at this size -- still and moving scene."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    torch.manual_seed(0)
    m = NeRFNetwork(time_resolution=8, num_frames=S.NUM_FRAMES, bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR,
                    lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev).eval()
    rng = np.random.default_rng(0)
    lo, ld = S.lidar_rays(n_rays, rng)
    co, cd = S.camera_rays(n_rays, rng)
    tl = [torch.from_numpy(a).to(dev)[None] for a in (lo, ld)]
    tc = [torch.from_numpy(a).to(dev)[None] for a in (co, cd)]
    tm = torch.tensor([[0.5]], device=dev)

    def step(fp16):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
            m.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=T)
            m.render(tc[0], tc[1], tm, cal_lidar_color=False, num_steps=T)

    def timed(fp16):
        """fp16 = the reference's `--fp16` option (set by the shipped config, configs/kitti360_1908.txt), i.e. the Trainer's autocast
        region around render: the flow MLP on the fused fp16 MFMA kernel.  False: its Linear layers in fp32 (rocBLAS)."""
        for _ in range(2):
            step(fp16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(fp16)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    dt = timed(True)
    dt32 = timed(False)
    out = {"metric": "rendered rays/sec (LiDAR+cam), dynamic 4-D field", "value": 2 * n_rays / dt, "ms_per_step": dt * 1e3,
           "flow_mlp": "fp16 MFMA (fused kernel)", "fp32_flow_mlp": {"value": 2 * n_rays / dt32, "ms_per_step": dt32 * 1e3},
           "scene_flow": "fresh initialisation: |flow| ~ 1e-8 of the unit cube, i.e. a static scene -- the neighbour-frame evaluations re-use "
                         "the base evaluation's gathers at every level (k_hash_dynamic3)",
           "num_rays": n_rays, "num_rays_lidar": n_rays, "num_steps": T, "parameters_M": sum(p.numel() for p in m.parameters()) / 1e6}
    # the other end: every sample moving by ~8e-4 of the unit cube (~0.3 m per frame at the KITTI-360 scale), 25 finest cells:
    # the neighbours gather their own entries at almost every level
    saved = {k: v.clone() for k, v in m.flow_net.state_dict().items()}
    with torch.no_grad():
        for p in m.flow_net.grid_enc.parameters():
            p.uniform_(-0.5, 0.5)
        m.flow_net.mlp[-1].weight.normal_(0, 6e-3)
    dtm = timed(True)
    out["moving_scene"] = {"value": 2 * n_rays / dtm, "ms_per_step": dtm * 1e3, "mean_abs_flow": 8e-4}
    m.flow_net.load_state_dict(saved)
    out["outputs_match_fixture"] = dynamic_fixture_check(m, dev, n_rays, T)
    # the multimodal training step on the same model (what main_nvsf.py trains): fwd + bwd + Adam, loss-scaled
    from nvsf.nerf.train_step import RenderTrainStep
    g = torch.Generator(device="cpu").manual_seed(3)
    batch = {"rays_o_lidar": tl[0], "rays_d_lidar": tl[1], "rays_o": tc[0], "rays_d": tc[1], "time": tm,
             "gt_depth": torch.rand(1, n_rays, generator=g).to(dev) * 0.5, "gt_raydrop": (torch.rand(1, n_rays, generator=g) > 0.3).float().to(dev),
             "gt_intensity": torch.rand(1, n_rays, generator=g).to(dev), "gt_rgb": torch.rand(1, n_rays, 3, generator=g).to(dev)}
    m.train()
    trainer = RenderTrainStep(m, num_steps=T, scale=S.SCALE)
    for i in range(24):  # the loss scaler backs off from 2^16 until the fp16 gradients of this model fit; then optimiser state and pools are in place
        before = trainer.scaler.get_scale()
        trainer.step(batch)
        if i >= 3 and trainer.scaler.get_scale() >= before:
            break
    trainer.sync()
    grads_match = train_grads_match(m, batch, T, S.SCALE, reference={"table_scatter": "atomic", "hash4d_bwd": "runs"}, init_scale=trainer.scaler.get_scale(),
                                    what="production plan of the space-time model (static hash: fp16 level-major gradient through the bins, plan "
                                         "(0, 8); flow grid: 2-feature view through the bins; space-time grids: LDS image fed column-major) against "
                                         "fp32 atomics for every table, every parameter, one full-size step at the loss scale the step's scaler "
                                         "settled at (K-planes: same kernel both ways)")
    for _ in range(2):
        trainer.step(batch)
    torch.cuda.synchronize()
    # sustained steps: the step leaves its last table's optimiser pass on the side stream for the next step to overlap, so a handful of
    # steps from an idle device reads ~1 ms per step higher than the loop the trainer runs (3 steps: 30.0 ms, 10+: 28.9 on the same box)
    n_train = max(10, steps)
    t0 = time.perf_counter()
    for _ in range(n_train):
        trainer.step(batch)
    torch.cuda.synchronize()
    dtt = (time.perf_counter() - t0) / n_train
    out["train"] = {"metric": "trained rays/sec (LiDAR+cam, fwd+bwd+Adam), dynamic 4-D field", "value": 2 * n_rays / dtt, "ms_per_step": dtt * 1e3,
                    "steps": n_train, "grads_match": grads_match}
    out["peak_mem_GiB"] = torch.cuda.max_memory_allocated() / 2 ** 30
    del trainer, m
    torch.cuda.empty_cache()
    return out


FINAL_LINE_MAX_BYTES = 6144  # the 29.7 KB line of round 4 was not parsed by the driver (18.2 KB in round 3 was)


def _r(v, digits=5):
    """Floats of the final line to `digits` significant digits (enough for every figure quoted; keeps the line short)."""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _r(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, digits) for x in v]
    return v


def _extremes(rows, bound):
    """Worst and best row of a leg against one roofline: {"n", "worst": [kernel, frac], "best": [kernel, frac]}."""
    rows = [r for r in rows if r.get("bound") == bound and r.get("frac")]
    if not rows:
        return None
    lo, hi = min(rows, key=lambda r: r["frac"]), max(rows, key=lambda r: r["frac"])
    return {"n": len(rows), "worst": [lo["kernel"][:48], lo["frac"]], "best": [hi["kernel"][:48], hi["frac"]]}


def write_detail(line):
    """Everything the legs measured (per-kernel rows, per-parameter errors, notes) goes to gpurun_out/bench_detail.json -- next to
    the driver's n1.out, merged back by gpurun -- NOT to stdout: the stdout of this program is the one compact line."""
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "bench_detail.json")
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:  # a read-only checkout: the figures of the final line do not depend on the file
        print(f"bench.py: detail not written ({e})", file=sys.stderr)
        return None


def _get(d, *path, default=None):
    """d[path[0]][path[1]]... or `default` when a key is missing / a level is not a dict (a leg that was skipped or failed)."""
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_line(line, detail_path=None):
    """The ONE stdout line: the contract keys in full, `roofline` and `cpu_baseline` in full (long sample descriptions shortened),
    and one-number summaries of the secondary legs.  The per-kernel rows, per-parameter gradient errors and notes are in the
    detail file.  Tolerant of legs that were skipped (N > 1, --no-extra-legs, a run under rocprofv3): a missing figure is left out,
    never an exception between the timed loop and the line.  Size is bounded (FINAL_LINE_MAX_BYTES)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "outputs_finite", "spinup_ms", "per_rank_ms_per_step", "ranks_seen", "build", "invalid", "kernel_ms_sum")
    out = {k: line[k] for k in keep if k in line}
    out["config"] = {k: v for k, v in line["config"].items() if k != "pass"}
    if isinstance(line.get("roofline"), dict):
        out["roofline"] = dict(line["roofline"])
    if isinstance(line.get("kernels"), list):
        out["kernels"] = [[r.get("kernel"), r.get("ms"), r.get("bound"), r.get("frac")] for r in line["kernels"]]
    c = line.get("cpu_baseline")
    if isinstance(c, dict):
        cb = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
        cb["sample"] = str(c.get("sample", ""))[:200]
        if _get(c, "one_core", "value") is not None:
            cb["one_core"] = c["one_core"]["value"]
        if isinstance(c.get("torch_cpu"), dict):
            cb["torch_cpu"] = {"value": c["torch_cpu"].get("value"), "cores": c["torch_cpu"].get("cores"),
                               "one_core": _get(c, "torch_cpu", "one_core", "value")}
        out["cpu_baseline"] = cb
    o = line.get("outputs_match_oracle")
    if isinstance(o, dict):
        errs = [v for k, v in o.items() if k.startswith("max_abs_err")]
        out["outputs_match_oracle"] = {"ok": o.get("ok"), "tolerance": o.get("tolerance"), "max_abs_err": max(errs) if errs else None,
                                       "checked_rays": [o.get("checked_rays_lidar"), o.get("checked_rays")]}
    if isinstance(line.get("fresh_batches"), dict):
        out["fresh_batches"] = {k: line["fresh_batches"].get(k) for k in ("ms_per_step", "value")}
    o = line.get("occupancy")
    if isinstance(o, dict):
        out["occupancy"] = {"value": _get(o, "eval", "value"), "ms_per_step": _get(o, "eval", "ms_per_step"),
                            "train_forward_ms": _get(o, "train_forward", "ms_per_step"), "occupied_fraction": o.get("occupied_fraction")}
        om = o.get("outputs_match_oracle")
        if isinstance(om, dict):
            out["occupancy"]["outputs_match_oracle"] = {"ok": om.get("ok"), "tolerance": om.get("tolerance"),
                                                         "max_abs_err": max(om.get("max_abs_err_lidar") or 0.0, om.get("max_abs_err") or 0.0),
                                                         "exact_fraction": [om.get("exact_fraction_lidar"), om.get("exact_fraction")],
                                                         "checked_rays": [om.get("checked_rays_lidar"), om.get("checked_rays")]}

    def grads(g):
        if not isinstance(g, dict):
            return None
        errs = g.get("max_rel_err")
        return {"ok": g.get("ok"), "tolerance": g.get("tolerance"), "max_rel_err": max(errs.values()) if isinstance(errs, dict) and errs else None}
    d = line.get("dynamic")
    if isinstance(d, dict):
        out["dynamic"] = {"value": d.get("value"), "ms_per_step": d.get("ms_per_step"), "fp32_flow_mlp_ms": _get(d, "fp32_flow_mlp", "ms_per_step"),
                          "moving_scene_ms": _get(d, "moving_scene", "ms_per_step"),
                          "outputs_match_fixture": {k: _get(d, "outputs_match_fixture", k) for k in ("ok", "tolerance", "max_abs_err")},
                          "train": {k: _get(d, "train", k) for k in ("value", "ms_per_step")}}
        if grads(_get(d, "train", "grads_match")):
            out["dynamic"]["train"]["grads_match"] = grads(d["train"]["grads_match"])
    r = line.get("reference_default_grid")
    if isinstance(r, dict):
        out["reference_default_grid"] = {"value": r.get("value"), "ms_per_step": r.get("ms_per_step"), "train_ms_per_step": _get(r, "train", "ms_per_step")}
        if grads(_get(r, "train", "grads_match")):
            out["reference_default_grid"]["train_grads_match"] = grads(r["train"]["grads_match"])
    if isinstance(line.get("eval"), dict):
        out["eval"] = {k: line["eval"][k] for k in ("value", "rays_per_s", "ms_per_frame", "scaling", "ranks") if k in line["eval"]}
        out["eval"]["unit"] = "frames/s"
    t = line.get("train")
    if isinstance(t, dict):
        tr = {"value": t.get("value"), "ms_per_step": t.get("ms_per_step"), "steps": t.get("steps"), "per_rank_ms_per_step": t.get("per_rank_ms_per_step")}
        if grads(t.get("grads_match")):
            tr["grads_match"] = grads(t["grads_match"])
        k = t.get("kernels")
        if isinstance(k, dict) and "launches_per_step" in k:
            tr["launches_per_step"] = k["launches_per_step"]
            tr["glue_launches_per_step"] = _get(k, "glue", "launches_per_step")
            tr["kernel_ms_per_step"] = k.get("kernel_ms_per_step")
            mf = [r for r in k.get("rows", []) if r.get("bound") == "mfma"]
            tr["mlp_kernels_mfma_frac_algorithmic"] = [[r["kernel"].split("(")[0][:28], r.get("ms"), r.get("frac")] for r in mf[:6]]
        a = t.get("allreduce")
        if isinstance(a, dict):
            tr["allreduce"] = {k: a[k] for k in ("payload_MB", "buckets", "sum_ms", "ring_estimate_ms") if k in a}
        if t.get("allreduce_collectives_per_step") is not None:
            tr["allreduce_collectives_per_step"] = t["allreduce_collectives_per_step"]
        out["train"] = tr
    for leg in ("raymarching", "field_ops"):
        rows = _get(line, leg, "kernels")
        if isinstance(rows, list):
            out[leg] = {"hbm": _extremes(rows, "hbm")}
            if _extremes(rows, "mfma"):
                out[leg]["mfma"] = _extremes(rows, "mfma")
            if leg == "field_ops":  # the stand-alone MLP rows carry both columns
                pipes = [[r["kernel"][:40], r["mfma_frac"]] for r in rows if r.get("mfma_frac")][:10]
                if pipes:
                    out[leg]["mlp_rows_mfma_frac"] = pipes
    if detail_path:
        out["detail"] = detail_path
    out = _r(out)
    n = len(json.dumps(out))
    if n > FINAL_LINE_MAX_BYTES:  # never print a line the driver may not parse: shed the optional summaries, keep the contract
        for k in ("field_ops", "raymarching", "kernels", "reference_default_grid", "occupancy", "eval", "fresh_batches", "dynamic", "train"):
            out.pop(k, None)
            if len(json.dumps(out)) <= FINAL_LINE_MAX_BYTES:
                break
        out["shed"] = f"summaries dropped to stay under {FINAL_LINE_MAX_BYTES} bytes (line was {n})"
    return out


def rank_agreement(dist, nvsf_build):
    """After rank 0's build + barrier: every rank loads libnvsf_hip.so and reports (nvsf_version(), digest of ALL kernel sources,
    sha1 of the shared object it mapped); a rank that disagrees with rank 0 ends the run.  Also what each rank believes the world
    to be (`ranks_seen`: MIN / MAX of get_world_size() and the number of ranks that answered)."""
    import hashlib
    from nvsf import _hip
    # csrc_digest: of the sources the mapped library was BUILT from (embedded by build.py), not of the sources on disk
    mine = {"version": _hip.version(), "csrc_digest": _hip.build_digest(), "sources_on_disk": nvsf_build.csrc_digest_all(),
            "lib_sha1": hashlib.sha1(open(_hip.LIB_PATH, "rb").read()).hexdigest()[:16]}
    if dist is None:
        return dict(mine, ranks_agree=True, ranks_seen={"min": 1, "max": 1, "answered": 1})
    ws = dist.get_world_size()
    views = [None] * ws
    dist.all_gather_object(views, dict(mine, world=ws, rank=dist.get_rank()))
    bad = [v for v in views if any(v[k] != views[0][k] for k in ("version", "csrc_digest", "lib_sha1"))]
    if bad:
        raise SystemExit(f"bench.py: ranks disagree about the kernel library: rank 0 {views[0]}, others {bad}")
    worlds = [v["world"] for v in views]
    return dict(mine, ranks_agree=True, ranks_seen={"min": min(worlds), "max": max(worlds), "answered": len({v["rank"] for v in views})})


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the NVSF hot path has no CPU fallback")
    # NVSF_BENCH_SAME_DEVICE=1 is a control-flow check for 1-GPU boxes: every rank uses cuda:0 and the (timing /
    # gradient) collectives run over gloo.  The numbers of such a run are meaningless; the default is one GPU per rank.
    same_device = os.environ.get("NVSF_BENCH_SAME_DEVICE", "0") == "1"
    dev_index = 0 if same_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL

    import build as nvsf_build
    if rank == 0:
        nvsf_build.build(verbose=False)
    if dist is not None:
        dist.barrier()
    build_info = rank_agreement(dist, nvsf_build)  # every rank loads the library rank 0 built: same version, same sources
    if build_info["csrc_digest"] != build_info["sources_on_disk"]:
        raise SystemExit(f"bench.py: the mapped library was built from sources {build_info['csrc_digest']}, the sources on disk are "
                         f"{build_info['sources_on_disk']} (build.build() should have relinked it)")
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic

    torch.manual_seed(0)
    model = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                              num_frames=S.NUM_FRAMES).to(dev).eval()
    rng = np.random.default_rng(1000 + rank)  # every rank renders its own frame
    lo, ld = S.lidar_rays(args.num_rays_lidar, rng)
    co, cd = S.camera_rays(args.num_rays, rng)
    tl = (torch.from_numpy(lo).to(dev)[None], torch.from_numpy(ld).to(dev)[None])
    tc = (torch.from_numpy(co).to(dev)[None], torch.from_numpy(cd).to(dev)[None])
    tm = torch.tensor([[0.5]], device=dev)
    T = args.num_steps

    def step():
        with torch.no_grad():
            a = model.render(tl[0], tl[1], tm, cal_lidar_color=True, num_steps=T)
            b = model.render(tc[0], tc[1], tm, cal_lidar_color=False, num_steps=T)
        return a, b

    # Warm-up = the timed loop's own pattern (the previous step's outputs stay referenced while the next step runs, so the caching
    # allocator has both sets of output buffers before the clock starts: a hipMalloc inside the timed region costs tens of ms),
    # and Python's cyclic collector is parked for the timed region (a full collection over torch's object graph is a host stall
    # of the same order; nothing in the loop creates reference cycles).
    import gc
    out = None
    gc.collect()
    gc.disable()  # before the spin-up: a collection between warm-up and timing would idle the device for tens of ms
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:  # device spin-up (clock ramp), not part of W or K
        for _ in range(50):  # back to back: an idle gap after every step would keep the device in its low state
            out = step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    trace = os.environ.get("NVSF_BENCH_TRACE") == "1"
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if trace else None
    if trace:
        evs[0].record()
    t0 = time.perf_counter()
    host = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        out = step()
        if trace:
            host.append(time.perf_counter() - h0)
            evs[i + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if trace:
        g = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        print("per-step GPU ms:", " ".join(f"{v:.2f}" for v in g), file=sys.stderr)
        print("per-step host enqueue ms:", " ".join(f"{v * 1e3:.2f}" for v in host), file=sys.stderr)
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        every = [None] * world
        dist.all_gather_object(every, elapsed)
        per_rank_ms = [float(v) / args.steps * 1e3 for v in every]
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(out[0]["image_lidar"]).all() and torch.isfinite(out[1]["image"]).all())

    # Second figure (not `value`): a FRESH ray batch every step with the sampler's jitter on (perturb = True), as an epoch of the
    # reference's loader delivers them -- the timed loop above re-renders one batch, which is cache-warm in the coarse levels.
    fresh = None
    if not args.no_extra_legs or os.environ.get("NVSF_BENCH_FRESH") == "1":
        n_b = 8
        batches = []
        for b in range(n_b):
            rng_b = np.random.default_rng(5000 + 97 * rank + b)
            lo_b, ld_b = S.lidar_rays(args.num_rays_lidar, rng_b)
            co_b, cd_b = S.camera_rays(args.num_rays, rng_b)
            batches.append(tuple(torch.from_numpy(a).to(dev)[None] for a in (lo_b, ld_b, co_b, cd_b)))

        def fresh_step(i):
            lo_t, ld_t, co_t, cd_t = batches[i % n_b]
            with torch.no_grad():
                a = model.render(lo_t, ld_t, tm, cal_lidar_color=True, num_steps=T, perturb=True)
                b = model.render(co_t, cd_t, tm, cal_lidar_color=False, num_steps=T, perturb=True)
            return a, b
        keep = None
        for i in range(2 * n_b):
            keep = fresh_step(i)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for i in range(args.steps):
            keep = fresh_step(i)
        torch.cuda.synchronize()
        tf = time.perf_counter() - tf0
        fresh = {"ms_per_step": tf / args.steps * 1e3, "value": (args.num_rays + args.num_rays_lidar) * args.steps / tf, "unit": "rays/s (this rank)",
                 "batches": n_b, "perturb": True,
                 "note": "a different pre-generated ray batch (another camera pose / LiDAR origin) per step, sampler jitter on: what an epoch delivers"}
        del keep, batches

    if rank == 0:
        rays_per_step = (args.num_rays + args.num_rays_lidar) * world
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "rendered rays/sec (LiDAR+cam)", "value": rays_per_step * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "C2: KITTI-360 seq-1908-shaped frame per GPU, uniform sampling, static hash field",
                       "num_rays": args.num_rays, "num_rays_lidar": args.num_rays_lidar, "num_steps": T,
                       "hash_grid": "L16 F2 T2^19 base16 max2048", "sigma_mlp": "32-64-16", "heads": "lidar 2x(87-64-64-1), rgb 31-64-64-3",
                       "pass": "forward render (no_grad): one wave-per-ray launch per batch (+ the XCD-sliced encode pass for the camera batch)", "parallelism": f"frame-sharded x{world}, no collective"},
            "outputs_finite": finite, "spinup_ms": args.spinup_ms,
            "per_rank_ms_per_step": per_rank_ms, "ranks_seen": build_info["ranks_seen"],
            "build": {k: build_info[k] for k in ("version", "csrc_digest", "sources_on_disk", "lib_sha1", "ranks_agree")},
        }
        if fresh is not None:
            line["fresh_batches"] = fresh
        if same_device:
            line["invalid"] = "NVSF_BENCH_SAME_DEVICE=1: all ranks shared cuda:0 (control-flow check only)"
        if not args.no_kernel_breakdown:
            rows = kernel_breakdown(model, {"lidar": (tl[0][0], tl[1][0], True), "camera": (tc[0][0], tc[1][0], False)}, T,
                                    min(max(5, args.steps), 40))
            line["kernels"] = rows
            pick = max(rows, key=lambda r: r["ms"])  # the dominant launch of the step
            line["roofline"] = {"kernel": pick["kernel"], "bound": pick["bound"], "achieved": pick["achieved"], "peak": pick["peak"],
                                "unit": pick["unit"], "frac": pick["frac"], "traffic": None, "avg_launch_ms": pick["ms"],
                                "algorithmic": pick["per_unit"], "units_per_launch": pick["units"]}
            line["roofline"].update(pmc_traffic(pick["kernel"]))
            line["kernel_ms_sum"] = sum(r["ms"] for r in rows)
        if args.cpu_rays > 0 and world == 1:
            rays = {True: (lo, ld), False: (co, cd)}
            n1 = min(args.cpu_rays, args.num_rays, args.num_rays_lidar)
            base, ref_out = cpu_baseline(model, T, rays, n1)  # one core: the scalar port as it is
            threads = min(args.cpu_threads if args.cpu_threads > 0 else (os.cpu_count() or 1), 16)  # a 1-GPU box's CPU share
            if threads > 1:  # and on the host's cores: the same port, rays split over a thread pool
                multi, ref_out = cpu_baseline(model, T, rays, min(args.cpu_rays * min(threads, 8), args.num_rays, args.num_rays_lidar), threads=threads)
                base, single = multi, base
                base["one_core"] = {"value": single["value"], "sample": single["sample"]}
            base["torch_cpu"] = torch_cpu_baseline(model, T, rays, min(2 * args.cpu_rays, args.num_rays, args.num_rays_lidar))
            line["cpu_baseline"] = base
            # the cpu_baseline leg's renders double as the checker of the timed outputs (the oracle is never on the timed path)
            line["outputs_match_oracle"] = outputs_match_oracle(out, ref_out)
        if not args.no_extra_legs and world == 1:  # secondary figures for BASELINE configs 3 and 5 (never `value`)
            line["occupancy"] = occupancy_leg(NeRFNetworkStatic, dev, args.num_rays, 10)
            line["dynamic"] = dynamic_leg(dev, args.num_rays, T, 3)
            line["reference_default_grid"] = reference_default_grid_leg(NeRFNetworkStatic, dev, args.num_rays, T, 10)
            line["raymarching"] = raymarching_leg(dev)
            line["field_ops"] = field_ops_leg(dev, args.num_rays, T)
    if not args.no_extra_legs:
        ev = eval_leg(model, dev, T, 3, dist)
        if rank == 0:
            line["eval"] = ev
    if args.train_steps > 0:
        tr = train_leg(model, tl, tc, tm, T, args.train_steps, dev, dist)
        if rank == 0:
            line["train"] = tr
    if rank == 0:
        detail_path = write_detail(line)
        try:
            final = compact_line(line, detail_path)
        except Exception as e:  # the figures are measured: a summary that trips over an unexpected leg shape must not lose the line
            keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline")
            final = _r({k: line[k] for k in keys if k in line})
            final["summary_error"], final["detail"] = f"{type(e).__name__}: {e}"[:200], detail_path
        print(json.dumps(final), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
