import torch
dev = torch.device("cuda:0")
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
for mb in (256, 710, 2048):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    ms = t(lambda: x.zero_())
    print(f"zero_ {mb} MB: {ms:.4f} ms, {mb*1.048576/ms:.0f} GB/s", flush=True)
    ms = t(lambda: x.fill_(1.5))
    print(f"fill_ {mb} MB: {ms:.4f} ms, {mb*1.048576/ms:.0f} GB/s", flush=True)
    ms = t(lambda: y.copy_(x))
    print(f"copy_ {mb} MB: {ms:.4f} ms, {2*mb*1.048576/ms:.0f} GB/s (r+w)", flush=True)
