# SQ instruction mix of the one-launch marcher: bash tools/prof_march.sh   (two --pmc passes over tools/bench_march_only.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_march
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/bench_march_only.py ${NVSF_AB:-0} > $O/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/p2 -- python3 $R/tools/bench_march_only.py ${NVSF_AB:-0} > $O/p2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/tools/bench_march_only.py ${NVSF_AB:-0} > $O/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "march_train_onepass" in k:
            per[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        # launches alternate: first half dense grid, second half 10 % grid
        for c, v in cs.items():
            h = len(v) // 2
            print(k, c, "dense %.2f M" % (sum(v[:h]) / h / 1e6), "sparse %.2f M" % (sum(v[h:]) / (len(v) - h) / 1e6), len(v))
PY
