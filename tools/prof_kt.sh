# kernel trace of one tool: bash tools/prof_kt.sh <tag> <script.py> [env assignments...]   (outputs under gpurun_out/<tag>/)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; S=$2; shift 2
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/$S > $O/run.log 2>&1
echo "rc=$?" >> $O/run.log
F=$(find $O -name '*_kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:26]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:8.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
tail -4 $O/run.log
