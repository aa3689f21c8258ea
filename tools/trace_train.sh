# kernel timeline of the training step: rocprofv3 --kernel-trace (timestamps) of tools/bench_train.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_train
mkdir -p $O
cd /tmp
K=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/bench_train.py > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last k_adam_prepare-ish back to the previous adam
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "k_adam_update" in n]
# steps end with a group of adam updates; find boundaries (gap > 20 kernels between adam groups)
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 5]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
print("kernels in step:", b - a, "span ms:", (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e6)
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e - s > 15:
        print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:90]}")
PY
