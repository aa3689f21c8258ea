# SQ instruction mix of the fused MLP backward kernels as the step launches them (tools/bench_heads.py):  bash tools/prof_heads.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_heads
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/bench_heads.py > $O/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/p2 -- python3 $R/tools/bench_heads.py > $O/p2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/tools/bench_heads.py > $O/p3.log 2>&1
cat $O/p1.log | grep -v amdgpu.ids | tail -8
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_mlp_bwd" in k:
            per[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            print(k, c, "%.3f M per launch" % (sum(v) / len(v) / 1e6), len(v))
PY
