"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output) into
profiles/<tag>_pmc_traffic.json: mean KB per launch for each kernel, split by launch order where a kernel is
launched with alternating workloads (LiDAR, camera).

    python tools/pmc_summary.py <fetch.csv> <write.csv> <out.json> [--alternating kernel_key ...]

Kernels named after --alternating (default: k_weights_fwd) are launched once per batch kind in LiDAR, camera order;
the others are launched for one batch kind only in bench.py's step (fused density: LiDAR; sliced density: camera)."""
import csv, json, os, sys, collections

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "selfsupervised-nvsf_amd"))
import build as nvsf_build


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


KEYS = {"k_render_uniformILb1ELb0": "render_uniform<lidar>", "k_render_uniformILb0ELb0": "render_uniform<camera>",
        "k_render_tail2ILb1": "render_uniform_tail<lidar>", "k_render_tail2ILb0": "render_uniform_tail<camera>",
        "k_render_uniformILb1ELb1": "render_uniform_tail1<lidar>", "k_render_uniformILb0ELb1": "render_uniform_tail1<camera>",
        "k_density_uniform_v2": "density_uniform_v2", "k_density_uniformILi": "density_uniform", "k_encode_sliced": "encode_sliced",
        "k_density_from_features": "density_from_features", "heads_uniformILb1": "heads_uniform<lidar>",
        "heads_uniformILb0": "heads_uniform<camera>", "k_weights_fwd": "k_weights_fwd", "k_near_far": "k_near_far"}


def short(name):
    for key, label in KEYS.items():
        if key in name:
            return label
    return None


def mfma_summary(counter_csv, kernel_stats_csv, out_path):
    """python tools/pmc_summary.py --mfma <pmc counter_collection.csv> <kernel-trace kernel_stats.csv> <out.json>: means per launch of
    the SQ counters of the render kernels; mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel-trace duration x 2.4 GHz x 1024 SIMDs),
    mfma_busy_over_sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs per CU-level busy count ... reported raw too)."""
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(counter_csv)):
        if short(r["Kernel_Name"]):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(kernel_stats_csv))}
    out = {"csrc_digest": nvsf_build.csrc_digest(),
           "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 (own pass); "
                   "means per launch. SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 = FLOP issued; SQ_VALU_MFMA_BUSY_CYCLES / MFMA instructions = 16 cycles per "
                   "v_mfma_f32_16x16x32_f16; mfma_pipe_util = busy cycles / (kernel-trace duration x 2.4 GHz x 1024 SIMDs); tflops = FLOP / duration.",
           "kernels": {}}
    for name, cs in per.items():
        e = {k: sum(v) / len(v) for k, v in cs.items()}
        e["label"] = short(name)
        d = next((v for k, v in dur.items() if k[:60] == name[:60]), None)
        if d and "SQ_INSTS_VALU_MFMA_MOPS_F16" in e:
            e["flop"] = e["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512.0
            e["avg_duration_ns"] = d
            e["tflops"] = e["flop"] / d / 1e3
            e["mfma_pipe_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (d * 2.4 * 1024)
        out["kernels"][name[:90]] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for k, e in out["kernels"].items():
        print(e["label"], {x: round(e[x], 4) for x in ("tflops", "mfma_pipe_util") if x in e})


def train_summary(mfma_csv, atomics_csv, kernel_stats_csv, out_path):
    """python tools/pmc_summary.py --train <SQ pass csv> <TCC atomics pass csv> <train kernel_stats.csv> <out.json>: the kernels of the
    training step (tools/bench_train.py) -- MFMA pipe utilisation of the MLP / fused field kernels and 64-byte atomic requests of the
    table-gradient kernels, means per launch; durations from the kernel trace of the same script."""
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(kernel_stats_csv))}
    want = ("k_mlp_fwd", "k_mlp_bwd", "k_density_uniform_v2", "k_density_from_features", "k_encode_sliced", "k_hashgrid_bwd")
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in (mfma_csv, atomics_csv):
        for r in csv.DictReader(open(path)):
            if any(k in r["Kernel_Name"] for k in want):
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"csrc_digest_all": nvsf_build.csrc_digest_all(),
           "note": "tools/bench_train.py (config 4 step, 4096 + 4096 rays x 768), two separate --pmc passes: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                   "SQ_INSTS_VALU_MFMA_MOPS_F16 and TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum; means per launch; mfma_pipe_util = MFMA busy cycles / "
                   "(kernel-trace duration x 2.4 GHz x 1024 SIMDs); atomic_TBps = atomic requests x 64 B / duration", "kernels": {}}
    for name, cs in per.items():
        e = {k: sum(v) / len(v) for k, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        d = next((v for k, v in dur.items() if k[:60] == name[:60]), None)
        if d:
            e["avg_duration_ns"] = d
            if "SQ_INSTS_VALU_MFMA_MOPS_F16" in e and e["SQ_INSTS_VALU_MFMA_MOPS_F16"] > 0:
                e["tflops"] = e["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512.0 / d / 1e3
                e["mfma_pipe_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (d * 2.4 * 1024)
            if "TCC_EA0_ATOMIC_sum" in e:
                e["atomic_TBps"] = e["TCC_EA0_ATOMIC_sum"] * 64.0 / d / 1e3
        out["kernels"][name[:100]] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for k, e in out["kernels"].items():
        print(k[:70], {x: round(e[x], 4) for x in ("tflops", "mfma_pipe_util", "TCC_EA0_ATOMIC_sum", "atomic_TBps") if x in e})


args = sys.argv[1:]
if args and args[0] == "--mfma":
    mfma_summary(*args[1:4])
    sys.exit(0)
if args and args[0] == "--train":
    train_summary(*args[1:5])
    sys.exit(0)
alternating = ["k_weights_fwd"]
if "--alternating" in args:
    i = args.index("--alternating")
    alternating = args[i + 1:]
    args = args[:i]
fetch, write, out_path = args
f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
out = {"csrc_digest": nvsf_build.csrc_digest(), "units": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, 4-byte gathers uncalibrated)", "kernels": {}}
for name in f:
    s = short(name)
    if not s:
        continue
    fv, wv = f[name], w.get(name, [])
    entry = {"launches": len(fv), "fetch_kb_mean": sum(fv) / len(fv), "write_kb_mean": (sum(wv) / len(wv)) if wv else None}
    if s in alternating and len(fv) % 2 == 0:
        entry["fetch_kb_lidar"] = sum(fv[0::2]) / (len(fv) // 2)
        entry["fetch_kb_camera"] = sum(fv[1::2]) / (len(fv) // 2)
        if wv:
            entry["write_kb_lidar"] = sum(wv[0::2]) / (len(wv) // 2)
            entry["write_kb_camera"] = sum(wv[1::2]) / (len(wv) // 2)
    out["kernels"][s] = entry
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
