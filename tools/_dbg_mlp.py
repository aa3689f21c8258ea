import sys
sys.path.insert(0, "selfsupervised-nvsf_amd")
import numpy as np, torch
from nvsf import field_ops as ops, testing
dev = torch.device("cuda:0")
n_in, n_out, n_hidden, M = 32, 16, 1, 64
spec = ops.MlpSpec(n_in, n_out, 64, n_hidden)
rng = np.random.default_rng(1)
mats = []
for a, b in spec.shapes:
    W = rng.integers(-2, 3, size=(a, b)).astype(np.float32); W[rng.random((a, b)) < 0.8] = 0; mats.append(W)
x = rng.integers(-2, 3, size=(M, n_in)).astype(np.float32)
g_out = rng.integers(-2, 3, size=(M, n_out)).astype(np.float32)
w16 = torch.from_numpy(np.concatenate([m.reshape(-1) for m in mats]).astype(np.float16)).to(dev)
res = {}
for k in ("wave", "staged"):
    with testing.variant(mlp_bwd=k):
        gx, gw = ops.mlp_backward(torch.from_numpy(x).to(dev), w16, spec, torch.from_numpy(g_out).to(dev), grad_scale=1.0)
    res[k] = (gx.cpu().numpy(), gw.cpu().numpy())
print("gx equal", np.array_equal(res["wave"][0], res["staged"][0]))
a, b = res["wave"][1], res["staged"][1]
d0a, d0b = a[:2048].reshape(64, 32), b[:2048].reshape(64, 32)
doa, dob = a[2048:].reshape(16, 64), b[2048:].reshape(16, 64)
print("dw0 equal", np.array_equal(d0a, d0b), "dwo equal", np.array_equal(doa, dob))
np.set_printoptions(linewidth=250, precision=3, suppress=True)
print("dwo wave\n", doa[:4, :20]); print("dwo ref\n", dob[:4, :20])
print("dw0 wave\n", d0a[:4, :20]); print("dw0 ref\n", d0b[:4, :20])
