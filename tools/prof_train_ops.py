"""torch.profiler view of one training step (which aten ops own the time outside the HIP kernels): `python tools/prof_train_ops.py [dynamic]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = "bench_train_dynamic.py" if len(sys.argv) > 1 and sys.argv[1] == "dynamic" else "bench_train.py"
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", which)).read().split("for _ in range(2): step.step(batch)")[0]
exec(src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT)))
from torch.profiler import profile, ProfilerActivity
for _ in range(3): step.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step.step(batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=int(os.environ.get("ROWS", 45)), max_name_column_width=50, max_shapes_column_width=70))
