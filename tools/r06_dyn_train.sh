# config-5 training step: same-box A/B of the K-planes scatter forms + kernel trace:  bash tools/r06_dyn_train.sh <N rays> <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-4096}; T=${2:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r06_dyn_$T
rm -rf $O; mkdir -p $O
for i in 1 2; do
  for pb in lds global; do
    echo "== planes_bwd=$pb"; PB=$pb N=$N K=8 timeout -k 10 200 python tools/bench_train_dynamic.py 2>&1 | grep "ms/step"
  done
done > $O/ab.log 2>&1
cat $O/ab.log
cd /tmp
N=$N K=5 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $GRAFT_REPO_ROOT/tools/bench_train_dynamic.py > $O/kt.log 2>&1
tail -1 $O/kt.log
python3 - $(find $O/kt -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"kernel time total {tot/1e6:.1f} ms over the traced run")
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 - $(find $O/kt -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'time_lds' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('time_lds launches (us):', [round(x) for x in d[-16:]], 'LDS', [r.get('LDS_Block_Size') for r in rows[-4:]], 'grid', [r.get('Grid_Size_X') for r in rows[-4:]])
PY
rm -rf $O/kt
