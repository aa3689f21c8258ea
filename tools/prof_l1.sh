# L1 / TA / L2 counters of the render kernels (diagnosis of the gather passes): bash tools/prof_l1.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-l1}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
B="python3 $R/bench.py --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0 --steps 3 --warmup 1 --spinup-ms 0"
timeout -k 10 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d $O/p5 -- $B > $O/p5.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_render" in k or "k_encode" in k:
            per[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
