# quick check of the headline path: bit-identity / oracle tests of the render kernels, then the bench line (no extra legs)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_density_sliced_gpu.py tests/test_render_static_gpu.py tests/test_occupancy_gpu.py tests/test_reference_fixtures_gpu.py -x -q -m gpu > gpurun_out/t1.log 2>&1
rc=$?
tail -3 gpurun_out/t1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-extra-legs --train-steps 0 --cpu-rays 0 > gpurun_out/b1.log 2>&1 || { tail -5 gpurun_out/b1.log; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/b1.log") if x.startswith("{")][-1]
d=json.loads(l)
print(d["value"], d["ms_per_step"], [(k[0], round(k[1],4)) for k in d["kernels"]])
PY
