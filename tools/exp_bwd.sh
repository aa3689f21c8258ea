python -m pytest tests/test_mlp_bwd_gpu.py tests/test_field_gpu.py tests/test_render_static_gpu.py tests/test_train_step_gpu.py tests/test_dynamic_gpu.py tests/test_occupancy_gpu.py -x -q 2>&1 | tail -12
K=8 python tools/bench_train.py 2>&1 | grep "train step"
