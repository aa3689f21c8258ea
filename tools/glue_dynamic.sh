# torch / runtime glue launches of a training step, per step:  bash tools/glue_dynamic.sh [bench_train_dynamic.py | bench_train.py] [steps traced]
S=${1:-bench_train_dynamic.py}
export GLUE_STEPS=${2:-5}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/glue_dyn
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/$S > $O/log.txt 2>&1
tail -1 $O/log.txt
python3 - $(find $O -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
import os
steps = int(os.environ.get('GLUE_STEPS', 5))
tot = sum(float(r['TotalDurationNs']) for r in rows)
glue = [r for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name']]
print(f"kernel time {tot/1e6/steps:.2f} ms/step, launches {sum(int(r['Calls']) for r in rows)/steps:.0f}/step; glue {sum(int(r['Calls']) for r in glue)/steps:.0f} launches, {sum(float(r['TotalDurationNs']) for r in glue)/1e6/steps:.2f} ms/step")
for r in sorted(glue, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    n = r['Name'].replace('void at::native::', '').replace('(anonymous namespace)::', '').replace('at::native::', '')
    print(f"  {int(r['Calls'])/steps:5.1f}/step {float(r['TotalDurationNs'])/1e6/steps:6.3f} ms  {n[:150]}")
PY
