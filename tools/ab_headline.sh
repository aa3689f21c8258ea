# same-box A/B of two builds of the library on the headline loop:  bash tools/ab_headline.sh <a.so> <b.so> [rounds]
R=$GRAFT_REPO_ROOT
L=$R/selfsupervised-nvsf_amd/lib/libnvsf_hip.so
for i in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp $R/$v $L
    python $R/bench.py --no-extra-legs --train-steps 0 --cpu-rays 0 --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), [round(r[1],4) for r in d['kernels'][:3]])"
  done
done
