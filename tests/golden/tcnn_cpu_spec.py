"""CPU stand-in for the `tinycudann` module surface, used ONLY to run the reference's Python model files on CPU
when generating golden fixtures (tests/golden/make_golden.py) and in CPU tests.  Forward passes call the CPU
oracle (oracle/field_oracle.c), i.e. the written specification of these operators; nothing here is shipped.

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
        xn = np.ascontiguousarray(x.detach().float().numpy().reshape(-1, self.n_input_dims))
        if self.otype == "HashGrid":
            table = self.params.detach().numpy().astype(np.float16)
            out = O.hashgrid_fwd(xn, tuple(range(self.n_input_dims)), table, self.spec)
        elif self.otype == "Frequency":
            out = O.freq_encode(xn, self.n_frequencies)
        else:
            out = O.sh4_encode(xn)
        return torch.from_numpy(out)  # fp16, like tcnn


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=None):
        super().__init__()
        cfg = dict(network_config)
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.spec = MlpSpec(self.n_input_dims, self.n_output_dims, int(cfg["n_neurons"]), int(cfg["n_hidden_layers"]))
        self.seed = _next_seed()
        self.params = nn.Parameter(torch.from_numpy(param_init.mlp_params(self.spec.shapes, self.seed)))

    def forward(self, x):
        xn = x.detach()
        xn = xn.numpy() if xn.dtype == torch.float16 else xn.float().numpy()
        xn = np.ascontiguousarray(xn.reshape(-1, self.n_input_dims))
        w = self.params.detach().numpy().astype(np.float16)
        out = O.mlp_fwd(xn, w, self.spec.n_in, self.spec.in_cols, self.spec.n_hidden, self.spec.hidden, self.spec.out_cols)
        return torch.from_numpy(out[:, :self.n_output_dims].copy())  # fp32 logits
