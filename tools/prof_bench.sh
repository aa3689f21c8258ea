export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0 > $R/gpurun_out/pb.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pb/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-80s calls %4s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
