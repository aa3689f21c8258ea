"""Host issue time of a training step against its device time (a host that cannot run ahead pays every launch-bound stretch in full),
and a cProfile of ten steps: `python tools/host_train.py [dynamic]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = "bench_train_dynamic.py" if len(sys.argv) > 1 and sys.argv[1] == "dynamic" else "bench_train.py"
sys.argv = ["x"]
src = open(os.path.join(ROOT, "tools", which)).read().split("for _ in range(2): step.step(batch)")[0]
exec(src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT)))
for _ in range(5): step.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1 - t0) / 20 * 1e3:.2f} ms; total per step {(t2 - t0) / 20 * 1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step.step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
