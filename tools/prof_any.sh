# usage: bash tools/prof_any.sh <tag> <python script> [args...]   -> per-kernel stats of one run (rocprofv3 kernel trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/"$@" > $R/gpurun_out/$TAG.log 2>&1
grep -v "^W2026\|^E2026\|^I2026" $R/gpurun_out/$TAG.log | tail -6
TAG=$TAG python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/" + os.environ["TAG"] + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-72s calls %5s  avg %9.1f us  min %9.1f  max %9.1f  %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
