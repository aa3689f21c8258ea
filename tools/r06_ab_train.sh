# same-box A/B of the density network's logit gradient: composed in the MLP backward's operand fetch / nvsf_sigma_geo_bwd matrix
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for dg in composed matrix; do
    echo "== DG=$dg"; DG=$dg K=40 timeout -k 10 200 python tools/bench_train.py 2>&1 | grep -v "^Traceback\|^  File\|TypeError\|^Exception ignored" | tail -3
  done
done > gpurun_out/r06_c2_ab_train.log 2>&1
tail -30 gpurun_out/r06_c2_ab_train.log
