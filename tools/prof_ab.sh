export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab -- python3 $R/tools/ab_density.py "$@" > $R/gpurun_out/ab.log 2>&1
tail -8 $R/gpurun_out/ab.log
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ab/*/*_kernel_trace.csv")[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    per[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    if "k_" in k:
        h = len(v) // 2
        print(k, len(v), "first-half(lidar/camera interleaved) mean us: even %.1f odd %.1f" % (sum(v[0::2]) / len(v[0::2]), sum(v[1::2]) / max(1, len(v[1::2]))))
PY
