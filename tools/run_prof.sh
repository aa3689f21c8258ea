# Round profile set: bash tools/run_prof.sh <tag>   (run on the GPU box through gpurun; outputs under gpurun_out/<tag>/)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-d}
O=$R/gpurun_out/$T
mkdir -p $O
# 1. two-rank control-flow check on one device (numbers meaningless, tagged invalid)
NVSF_BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --train-steps 2 --no-extra-legs --cpu-rays 0 > $O/bench2.log 2>&1
rc2=$?   # of the bench run itself (captured before the copy below replaces $?)
cp $R/gpurun_out/bench_detail.json $O/bench2_detail.json
echo "rc2=$rc2" >> $O/bench2.log
# 2. default bench line
timeout 900 python bench.py > $O/bench1.log 2>&1
cp $R/gpurun_out/bench_detail.json $O/bench1_detail.json   # (the profiled runs below overwrite gpurun_out/bench_detail.json)
cd /tmp
B="python3 $R/bench.py --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0"
# 3. kernel trace of the timed render loop; 4./5. PMC passes (separate runs, counters only)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 20 --warmup 5 > $O/kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- $B --steps 3 --warmup 1 > $O/pf.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- $B --steps 3 --warmup 1 > $O/pw.log 2>&1
# 5b. MFMA pipe occupancy of the render kernels (SQ counters only, their own pass)
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pm -- python3 $R/bench.py --cpu-rays 0 --no-extra-legs --train-steps 0 --steps 5 --warmup 1 > $O/pm.log 2>&1
# 6. kernel traces of the secondary legs
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/tools/bench_train.py > $O/kt_train.log 2>&1
# 6b. counters of the training step's kernels: MFMA pipe (SQ) and 64-byte atomic requests (TCC), separate passes
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pm_train -- python3 $R/tools/bench_train.py > $O/pm_train.log 2>&1
timeout 900 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/pa_train -- python3 $R/tools/bench_train.py > $O/pa_train.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_occ -- python3 $R/tools/bench_occupancy.py > $O/kt_occ.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dyn -- python3 $R/tools/bench_dynamic.py > $O/kt_dyn.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dyn_train -- python3 $R/tools/bench_train_dynamic.py > $O/kt_dyn_train.log 2>&1
# 7. roofline tools of the extension kernels and the stand-alone operators
timeout 600 python3 $R/tools/bench_raymarching.py > $O/raymarching.json 2> $O/raymarching.log
timeout 600 python3 $R/tools/bench_field_ops.py > $O/field_ops.json 2> $O/field_ops.log
find $O -name '*_kernel_stats.csv' -o -name '*_counter_collection.csv'
tail -c 300 $O/bench2.log
