"""Copies the summaries of a `tools/run_prof.sh <tag>` run (gpurun_out/<tag>/) into profiles/<tag>_*:
kernel stats of every leg, the PMC passes trimmed to the render kernels' rows of the last dispatches, the two PMC summaries
(tools/pmc_summary.py), the roofline JSONs of the tool legs and the bench lines (one JSON line per file).

    python tools/collect_prof.py <tag>"""
import csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def one(pat):
    f = glob.glob(os.path.join(src, pat), recursive=True)
    assert f, pat
    return max(f, key=os.path.getmtime)  # a tag run twice leaves both runs' files behind: the newest counts


for leg, name in (("kt", "bench"), ("kt_train", "train"), ("kt_occ", "occupancy"), ("kt_dyn", "dynamic"), ("kt_dyn_train", "dynamic_train")):
    shutil.copy(one(f"{leg}/**/*_kernel_stats.csv"), f"{dst}/{tag}_{name}_kernel_stats.csv")


def trim(srcf, out, last=48):
    rows = list(csv.DictReader(open(srcf)))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_render", "k_encode", "k_near_far", "k_weights"))]
    ids = sorted({int(r["Dispatch_Id"]) for r in keep})[-last:]
    keep = sorted((r for r in keep if int(r["Dispatch_Id"]) in ids), key=lambda r: int(r["Dispatch_Id"]))
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)


trim(one("pf/**/*counter_collection.csv"), f"{dst}/{tag}_pmc_fetch_size.csv")
trim(one("pw/**/*counter_collection.csv"), f"{dst}/{tag}_pmc_write_size.csv")
trim(one("pm/**/*counter_collection.csv"), f"{dst}/{tag}_pmc_mfma.csv")
shutil.copy(f"{src}/raymarching.json", f"{dst}/{tag}_raymarching_rooflines.json")
shutil.copy(f"{src}/field_ops.json", f"{dst}/{tag}_field_ops_rooflines.json")
# the stdout of bench.py is ONE compact line; the per-kernel rows of the legs are in the detail file written beside it
for log, name, detail in (("bench1.log", "bench_line", "bench1_detail.json"), ("bench2.log", "bench2_same_device_line", "bench2_detail.json")):
    line = [x for x in open(f"{src}/{log}") if x.startswith("{")][-1]
    open(f"{dst}/{tag}_{name}.json", "w").write(json.dumps(json.loads(line)) + "\n")
    if os.path.exists(f"{src}/{detail}"):
        shutil.copy(f"{src}/{detail}", f"{dst}/{tag}_{name.replace('_line', '_detail')}.json")
py = sys.executable
subprocess.check_call([py, os.path.join(ROOT, "tools", "pmc_summary.py"), f"{dst}/{tag}_pmc_fetch_size.csv", f"{dst}/{tag}_pmc_write_size.csv",
                       f"{dst}/{tag}_pmc_traffic.json"])
subprocess.check_call([py, os.path.join(ROOT, "tools", "pmc_summary.py"), "--mfma", f"{dst}/{tag}_pmc_mfma.csv", f"{dst}/{tag}_bench_kernel_stats.csv",
                       f"{dst}/{tag}_pmc_mfma.json"])
if glob.glob(os.path.join(src, "pm_train/**/*counter_collection.csv"), recursive=True):
    subprocess.check_call([py, os.path.join(ROOT, "tools", "pmc_summary.py"), "--train", one("pm_train/**/*counter_collection.csv"),
                           one("pa_train/**/*counter_collection.csv"), f"{dst}/{tag}_train_kernel_stats.csv", f"{dst}/{tag}_pmc_train.json"])
d = json.loads(open(f"{dst}/{tag}_bench_line.json").read())
print(tag, d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
print({k: {a: b for a, b in v.items() if a in ("value", "ms_per_step")} for k, v in d.items() if isinstance(v, dict) and k in ("train", "dynamic", "eval")},
      "dynamic train", d["dynamic"]["train"]["ms_per_step"], "moving", d["dynamic"]["moving_scene_ms"], "occupancy", d["occupancy"]["value"])
print(d["kernels"], d.get("outputs_match_oracle", {}).get("ok"))
