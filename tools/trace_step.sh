# kernel timeline of one step of a tools/bench_*.py script:  bash tools/trace_step.sh <script.py> <marker kernel substring> <markers per step>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_step
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/$1 > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$2" "$3" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
per = int(sys.argv[3])
a, b = marks[-2 * per], marks[-per]          # the second to last step
t0 = int(rows[a]["Start_Timestamp"])
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3
print(f"kernels in step: {b - a}  span {span:.1f} us  sum of kernel durations {busy:.1f} us")
prev_end = t0
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    gap = s - (prev_end - t0) / 1e3
    if e - s > 25 or gap > 25:
        print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} us  gap {gap:7.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:80]}")
    prev_end = max(prev_end, int(r["End_Timestamp"]))
PY
