cd $GRAFT_REPO_ROOT
bash tools/ab_headline.sh lib_exp/prod.so lib_exp/tcnn.so 3 > gpurun_out/r06_ab_tcnn.log 2>&1
cp lib_exp/tcnn.so selfsupervised-nvsf_amd/lib/libnvsf_hip.so
python bench.py --no-extra-legs --train-steps 0 --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tcnn blend vs fp32-spec oracle:', d.get('outputs_match_oracle'))" >> gpurun_out/r06_ab_tcnn.log 2>&1
cat gpurun_out/r06_ab_tcnn.log
