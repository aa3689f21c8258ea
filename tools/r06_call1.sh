# round 6, call 1: new kernels' tests, the statistical full-size tests with their printed spreads, the full suite, config-4 step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_dynamic_gpu.py -x -q -m gpu -k "planes or time_plane" > $O/r06_c1_planes.log 2>&1; echo "planes rc=$?"
timeout -k 10 600 python -m pytest tests/test_config5_train_full_size_gpu.py -x -q -m gpu -s > $O/r06_c1_full5.log 2>&1; echo "full5 rc=$?"
timeout -k 10 300 python -m pytest tests/test_mlp_bwd_gpu.py tests/test_train_step_gpu.py -x -q -m gpu > $O/r06_c1_mlp.log 2>&1; echo "mlp rc=$?"
for i in 1 2; do
  for dg in composed matrix; do
    echo "== DG=$dg"; DG=$dg K=40 timeout -k 10 200 python tools/bench_train.py 2>&1 | tail -2
  done
done > $O/r06_c1_ab_train.log 2>&1
tail -12 $O/r06_c1_ab_train.log
