cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in side main; do echo "== H4D=$v"; H4D=$v N=${1:-4096} K=10 timeout -k 10 200 python tools/bench_train_dynamic.py 2>&1 | grep "ms/step"; done; done > gpurun_out/r06_ab_h4d.log 2>&1
cat gpurun_out/r06_ab_h4d.log
