"""CPU stand-in for the `tinycudann` module surface, used ONLY to run the reference's Python model files on CPU
when generating golden fixtures (tests/golden/make_golden.py) and in CPU tests.  Forward passes call the CPU
oracle (oracle/field_oracle.c), i.e. the written specification of these operators; nothing here is shipped.

Both module kinds are `torch.autograd.Function`s, so the reference's own `NeRFNetwork.render(...)` + loss can be
back-propagated on CPU (golden_dynamic.gen_network_grads):
  * HashGrid: d L / d table = oracle_hashgrid_bwd (the transpose of the forward's interpolation, fp32 sums); no gradient
    to the input positions (nothing on the reference's path needs it: the hash encoders see ray samples, and the
    flow-warped neighbour evaluations run under no_grad, network_dynamic.py:244-262);
  * FullyFusedMLP: the exact gradient of the specified forward with the fp16 roundings treated as identities --
    activations recomputed by the oracle (fp16 hidden activations), the chain rule evaluated in fp64 with tcnn's padding
    (input columns n_in..in_cols are constant ones, output rows n_out..out_cols receive zero gradient); gradients to the
    input leave in the input's dtype (fp16 for encoder outputs, as tcnn returns them).

Parameters are created from `param_init` (seeded numpy) in construction order, so a test can rebuild exactly the
same parameter values for the HIP modules without storing multi-megabyte tables in the fixtures.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "selfsupervised-nvsf_amd"))

import oracle_lib as O  # noqa: E402
import param_init  # noqa: E402
from nvsf.field_ops import GridSpec, MlpSpec  # host-side level / layout tables only (pure Python)  # noqa: E402

_counter = [0]


def reset_seed_counter(start=0):
    _counter[0] = start


def _next_seed():
    _counter[0] += 1
    return _counter[0]


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, seed=None, dtype=None):
        super().__init__()
        self.n_input_dims = int(n_input_dims)
        cfg = dict(encoding_config)
        self.otype = cfg.get("otype", "HashGrid")
        self.seed = _next_seed()
        if self.otype == "HashGrid":
            self.spec = GridSpec(self.n_input_dims, cfg["n_levels"], cfg["n_features_per_level"], cfg["log2_hashmap_size"],
                                 cfg["base_resolution"], cfg["per_level_scale"])
            self.n_output_dims = self.spec.n_output_dims
            self.params = nn.Parameter(torch.from_numpy(param_init.grid_params(self.spec.n_params, self.seed)))
        elif self.otype == "Frequency":
            self.n_frequencies = int(cfg.get("n_frequencies", 12))
            self.n_output_dims = self.n_input_dims * 2 * self.n_frequencies
            self.params = nn.Parameter(torch.zeros(0))
        elif self.otype == "SphericalHarmonics":
            self.n_output_dims = 16
            self.params = nn.Parameter(torch.zeros(0))
        else:
            raise NotImplementedError(self.otype)

    def forward(self, x):
        if self.otype == "HashGrid":
            return _HashGridFn.apply(x, self.params, self)
        xn = np.ascontiguousarray(x.detach().float().numpy().reshape(-1, self.n_input_dims))
        out = O.freq_encode(xn, self.n_frequencies) if self.otype == "Frequency" else O.sh4_encode(xn)
        return torch.from_numpy(out)  # fp16, like tcnn


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, enc):
        xn = np.ascontiguousarray(x.detach().float().numpy().reshape(-1, enc.n_input_dims))
        table = params.detach().numpy().astype(np.float16)
        ctx.xn, ctx.enc = xn, enc
        return torch.from_numpy(O.hashgrid_fwd(xn, tuple(range(enc.n_input_dims)), table, enc.spec))  # fp16, like tcnn

    @staticmethod
    def backward(ctx, grad_out):
        enc = ctx.enc
        g = np.ascontiguousarray(grad_out.detach().float().numpy())
        return None, torch.from_numpy(O.hashgrid_bwd(ctx.xn, tuple(range(enc.n_input_dims)), enc.spec, g)), None


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=None):
        super().__init__()
        cfg = dict(network_config)
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.spec = MlpSpec(self.n_input_dims, self.n_output_dims, int(cfg["n_neurons"]), int(cfg["n_hidden_layers"]))
        self.seed = _next_seed()
        self.params = nn.Parameter(torch.from_numpy(param_init.mlp_params(self.spec.shapes, self.seed)))

    def forward(self, x):
        return _MlpFn.apply(x, self.params, self)


class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, net):
        xn = x.detach()
        xn = xn.numpy() if xn.dtype == torch.float16 else xn.float().numpy()
        xn = np.ascontiguousarray(xn.reshape(-1, net.n_input_dims))
        w = params.detach().numpy().astype(np.float16)
        sp = net.spec
        out, hid = O.mlp_fwd(xn, w, sp.n_in, sp.in_cols, sp.n_hidden, sp.hidden, sp.out_cols, want_hidden=True)
        ctx.xn, ctx.w, ctx.hid, ctx.net, ctx.x_dtype = xn, w, hid, net, x.dtype
        return torch.from_numpy(out[:, :net.n_output_dims].copy())  # fp32 logits

    @staticmethod
    def backward(ctx, grad_out):
        sp, M = ctx.net.spec, ctx.xn.shape[0]
        mats = [m.double().numpy() for m in sp.split(torch.from_numpy(ctx.w))]  # W0 [hidden, in_cols], ..., W_out [out_cols, hidden]
        a0 = np.ones((M, sp.in_cols), np.float64)
        a0[:, :sp.n_in] = ctx.xn.astype(np.float16).astype(np.float64)  # the network sees its input rounded to fp16
        acts = [a0] + [ctx.hid[:, l, :].astype(np.float64) for l in range(sp.n_hidden)]
        g = np.zeros((M, sp.out_cols), np.float64)
        g[:, :sp.n_out] = grad_out.detach().double().numpy()
        grads = []
        for li in range(len(mats) - 1, -1, -1):
            grads.append(g.T @ acts[li])
            g = g @ mats[li]
            if li > 0:
                g = g * (acts[li] > 0)
        grad_params = torch.from_numpy(np.concatenate([t.reshape(-1) for t in reversed(grads)]).astype(np.float32))
        grad_x = torch.from_numpy(g[:, :sp.n_in].astype(np.float32)).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        return grad_x, grad_params, None
