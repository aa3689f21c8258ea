set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c
# 1. two-rank control-flow check on one device
NVSF_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 --train-steps 2 > $R/gpurun_out/c/bench2.log 2>&1
echo "rc2=$?" >> $R/gpurun_out/c/bench2.log
# 2. default bench line
timeout 900 python bench.py > $R/gpurun_out/c/bench1.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c/kt -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0 > $R/gpurun_out/c/kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c/pf -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0 > $R/gpurun_out/c/pf.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c/pw -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-kernel-breakdown --no-extra-legs --train-steps 0 > $R/gpurun_out/c/pw.log 2>&1
find $R/gpurun_out/c -name '*.csv' | head -30
tail -3 $R/gpurun_out/c/bench2.log
