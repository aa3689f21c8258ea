python -m pytest tests/test_render_static_gpu.py tests/test_density_sliced_gpu.py -x -q 2>&1 | tail -8
for v in split fused; do echo "== $v"; NVSF_RENDER_UNIFORM=$v python bench.py --no-extra-legs --cpu-rays 0 --train-steps 0 --no-kernel-breakdown 2>&1 | tail -1 | cut -c1-200; done
