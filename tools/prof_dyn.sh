export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dyn -- python3 $R/tools/bench_dynamic.py > $R/gpurun_out/dyn.log 2>&1
grep "RD forward\|params" $R/gpurun_out/dyn.log
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/dyn/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (7 steps):", tot / 7e6)
for r in rows[:22]:
    print("%-70s calls %5s  total %8.2f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
