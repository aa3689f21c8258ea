# the whole -m gpu suite as the driver runs it (+ durations), then the default bench line:  bash tools/r06_suite.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-a}
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r06_suite_$T.log 2>&1; rc=$?
echo "suite rc=$rc"; tail -4 gpurun_out/r06_suite_$T.log
if [ $rc -eq 0 ] && [ "$2" != "nobench" ]; then
  timeout -k 10 600 python bench.py > gpurun_out/r06_bench_$T.log 2> gpurun_out/r06_bench_$T.err; echo "bench rc=$?"
  cp gpurun_out/bench_detail.json gpurun_out/r06_bench_${T}_detail.json
  tail -c 4200 gpurun_out/r06_bench_$T.log
fi
