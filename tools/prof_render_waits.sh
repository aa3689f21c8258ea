# Where the waves of the headline's render kernels spend their cycles (SQ counters, own passes):  bash tools/prof_render_waits.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_render_waits
rm -rf $O; mkdir -p $O
cd /tmp
B="python3 $R/bench.py --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0 --steps 3 --warmup 1 --spinup-ms 0"
timeout -k 10 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_render" in k or "k_encode" in k:
            per[k[20:62]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()}, "(millions per launch)")
PY
tail -2 $O/p3.log
