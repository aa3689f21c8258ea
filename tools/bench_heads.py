"""The per-sample heads' forward as the training step launches it: 87-64-64-1 (LiDAR) and 31-64-64-3 (colour) on shared-prefix rows,
only the logits stored; 4096 rays x 768 samples.  MFMA utilisation from the algorithmic FLOP."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import torch
from nvsf import field_ops as ops
dev = torch.device("cuda:0")
N, T = 4096, 768
M = N * T
g = torch.Generator(device="cpu").manual_seed(0)
x16 = torch.randn(M, 16, generator=g).to(dev).half()
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, n_in, n_out, n_pre in (("lidar head 87-64-64-1", 87, 1, 72), ("colour head 31-64-64-3", 31, 3, 16)):
    spec = ops.MlpSpec(n_in, n_out, 64, 2)
    enc = torch.randn(N, n_pre, generator=g).to(dev).half()
    w = (torch.randn(spec.n_params, generator=g) * 0.1).to(dev).half()
    out = torch.empty(M, 4 if n_out == 3 else n_out, device=dev)
    ms = timed(lambda: ops.mlp_forward(x16, w, spec, out=out, prefix=(enc, T, n_pre), n_store=n_out))
    flop = 2.0 * M * sum(a * b for a, b in spec.shapes)
    print(f"{name}: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s = {flop / ms / 1e9 / 2500:.3f} of the dense fp16 MFMA peak", flush=True)

# ---- the backward as the training step launches it: shared-prefix rows, only the 15 geometry columns of dL/dx requested (16-byte rows),
# the second LiDAR head accumulating into the first one's gradient; and the density MLP handing its input gradient over level-major
from nvsf import testing
for name, n_in, n_out, n_pre in (("lidar head 87-64-64-1", 87, 1, 72), ("colour head 31-64-64-3", 31, 3, 16)):
    spec = ops.MlpSpec(n_in, n_out, 64, 2)
    enc = torch.randn(N, n_pre, generator=g).to(dev).half()
    w = (torch.randn(spec.n_params, generator=g) * 0.1).to(dev).half()
    go = (torch.randn(M, n_out, generator=g) * 0.01).to(dev)
    gx = torch.empty(M, 16, device=dev)[:, :15]
    flop = 3 * 2.0 * M * sum(a * b for a, b in spec.shapes)
    for kernel in ("wave", "staged"):
        with testing.variant(mlp_bwd=kernel):
            ms = timed(lambda: ops.mlp_backward(x16, w, spec, go, grad_x=gx, gx_col0=n_pre, prefix=(enc, T, n_pre)), reps=10)
        print(f"bwd {name} [{kernel}]: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s = {flop / ms / 1e9 / 2500:.3f} of the dense fp16 MFMA peak (3 x forward FLOP)", flush=True)
        if kernel == "wave" and n_out == 1:  # the second LiDAR head of the step ADDS its geometry gradient to the first one's
            gx.zero_()
            with testing.variant(mlp_bwd=kernel):
                ms = timed(lambda: ops.mlp_backward(x16, w, spec, go, grad_x=gx, gx_col0=n_pre, prefix=(enc, T, n_pre), accumulate=True), reps=10)
            print(f"bwd {name} [{kernel}, accumulating dL/dx]: {ms:.4f} ms", flush=True)
spec = ops.MlpSpec(32, 16, 64, 1)
w = (torch.randn(spec.n_params, generator=g) * 0.1).to(dev).half()
feat = torch.randn(M, 32, generator=g).to(dev).half()
go = (torch.randn(M, 16, generator=g) * 0.01).to(dev)
flop = 3 * 2.0 * M * sum(a * b for a, b in spec.shapes)
for kernel in ("wave", "staged"):
    with testing.variant(mlp_bwd=kernel):
        ms = timed(lambda: ops.mlp_backward(feat, w, spec, go, grad_x_blocks=2), reps=10)
    print(f"bwd sigma 32-64-16, level-major dL/dx [{kernel}]: {ms:.4f} ms  {flop / ms / 1e9:.0f} TFLOP/s = {flop / ms / 1e9 / 2500:.3f}; "
          f"{(64 + 64 + 128) * M / ms / 1e6:.0f} GB/s of rows", flush=True)
