import os, sys, time
sys.path.insert(0, "selfsupervised-nvsf_amd"); sys.argv=["x"]
os.environ["K"]="1"
exec(open("tools/bench_train.py").read().split("for _ in range(2): step.step(batch)")[0])
for _ in range(5): step.step(batch)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): step.step(batch)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"host issue time per step {(t1-t0)/20*1e3:.2f} ms; total per step {(t2-t0)/20*1e3:.2f} ms")
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(10): step.step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
