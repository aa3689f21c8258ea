"""`tinycudann`-compatible module surface for MI355X.

The reference builds its encoders and MLPs as `tcnn.Encoding(n_input_dims, encoding_config)` and
`tcnn.Network(n_input_dims, n_output_dims, network_config)` (hash_field.py:47-57,109-119;
flow_field.py:70-80; network_dynamic.py:108-114,125-135,138-161,165-170,180-189).  This package offers
the same constructors, `.n_input_dims`, `.n_output_dims`, a flat fp32 `.params` nn.Parameter and
`forward(x)` (encodings return fp16 features, networks return their fp32 logits -- see DESIGN.md 4.3), implemented on the HIP kernels of libnvsf_hip.so -- so the reference's model
files construct and run against it unchanged.  Numerics follow DESIGN.md section 4 (the published
tiny-cuda-nn algorithm; that library is an unpinned third-party dependency of the reference).

Supported otypes: encodings HashGrid / Grid (hash, linear interpolation), Frequency, SphericalHarmonics
(degree 4); networks FullyFusedMLP / CutlassMLP with ReLU hidden activation, no output activation,
64 neurons.  Anything else raises (no silent substitution).
"""
import math

import torch
import torch.nn as nn

from nvsf import field_ops as _ops

__version__ = "nvsf-hip"


class _HalfCache:
    """fp16 copy of an fp32 master parameter, refreshed when the parameter is modified in place."""

    def __init__(self):
        self._key, self._half = None, None
        self.pending = None  # event behind an optimiser update of the parameter that was issued on another stream (RenderTrainStep)

    def get(self, p):
        if self.pending is not None:  # the reader's stream waits for that update before it looks at the parameter
            torch.cuda.current_stream(p.device).wait_event(self.pending)
            self.pending = None
        key = (p.data_ptr(), p._version, p.device)
        if key != self._key:
            self._half = p.detach().to(torch.float16).contiguous()
            self._key = key
        return self._half

    def writable(self, p):
        """The fp16 buffer for a kernel that is about to update `p` and write the copy itself (nvsf_adam_update): allocated (and
        filled) if it does not exist or `p` moved; the caller calls `mark_fresh(p)` after bumping the parameter's version."""
        # ... or is STALE (`p` changed since the copy was made -- load_state_dict, ema.restore -- with no forward read since): the
        # update pass leaves the copy alone when the step is skipped on an overflow, and mark_fresh() would then bless old values
        if self._half is None or self._half.numel() != p.numel() or self._key != (p.data_ptr(), p._version, p.device):
            self.get(p)
        return self._half

    def mark_fresh(self, p):
        self._key = (p.data_ptr(), p._version, p.device)


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        super().__init__()
        self.n_input_dims = int(n_input_dims)
        self.encoding_config = dict(encoding_config)
        self.otype = self.encoding_config.get("otype", "HashGrid")
        self.dtype = torch.float16
        self._cache = _HalfCache()
        if self.otype in ("HashGrid", "Grid"):
            cfg = self.encoding_config
            if cfg.get("type", "Hash") != "Hash" or cfg.get("interpolation", "Linear") != "Linear":
                raise NotImplementedError("only hashed, linearly interpolated grids are implemented")
            self.spec = _ops.GridSpec(self.n_input_dims, cfg.get("n_levels", 16), cfg.get("n_features_per_level", 2),
                                      cfg.get("log2_hashmap_size", 19), cfg.get("base_resolution", 16),
                                      cfg.get("per_level_scale", 2.0))
            self.n_output_dims = self.spec.n_output_dims
            g = torch.Generator().manual_seed(int(seed))
            init = (torch.rand(self.spec.n_params, generator=g) * 2.0 - 1.0) * 1e-4  # tcnn: U(-1e-4, 1e-4)
            self.params = nn.Parameter(init)
            self._cols = tuple(range(self.n_input_dims))
        elif self.otype == "Frequency":
            self.n_frequencies = int(self.encoding_config.get("n_frequencies", 12))  # "degree" is not a tcnn key
            self.n_output_dims = self.n_input_dims * self.n_frequencies * 2
            self.params = nn.Parameter(torch.zeros(0))
        elif self.otype == "SphericalHarmonics":
            if int(self.encoding_config.get("degree", 4)) != 4 or self.n_input_dims != 3:
                raise NotImplementedError("SphericalHarmonics: degree 4 on 3-D inputs only")
            self.n_output_dims = 16
            self.params = nn.Parameter(torch.zeros(0))
        else:
            raise NotImplementedError(f"encoding otype {self.otype!r}")

    def table_f16(self):
        return self._cache.get(self.params)

    def forward(self, x, level_major=False):
        """level_major (not part of tiny-cuda-nn's interface): fp16 [n_levels, M, n_features_per_level] where ops.level_major_eligible,
        with autograd -- the static hash of the space-time field hands its features to the fused density tail that way."""
        x = x.reshape(-1, self.n_input_dims) if x.dim() != 2 else x
        if self.otype in ("HashGrid", "Grid"):
            lm = bool(level_major) and x.is_cuda and self._cols == (0, 1, 2) and _ops.level_major_eligible(self.spec)
            return _ops.HashGridFn.apply(x, self.params, self.table_f16(), self.spec, self._cols, _ops.rows_hint(self), _ops.train_context(self), lm)
        if self.otype == "Frequency":
            return _ops.freq_encode(x, self.n_frequencies)
        return _ops.sh4_encode(x)

    def forward_level_major(self, x):
        """HashGrid, no autograd: the features of forward(x) stored level by level, fp16 [n_levels, M, n_features_per_level], or None
        where that form is not built (ops.level_major_eligible).  Not part of tiny-cuda-nn's interface: the fused density tail of the
        space-time field reads it (network_dynamic._density_tail_fused)."""
        if self.otype not in ("HashGrid", "Grid") or torch.is_grad_enabled() or not x.is_cuda or x.dim() != 2 or not _ops.level_major_eligible(self.spec):
            return None
        return _ops.hashgrid_forward_level_major(x, self.table_f16(), self.spec)

    def encode_columns(self, x, cols):
        """HashGrid only: encode the columns `cols` of a wider coordinate matrix in place (no gather copy) --
        e.g. the (x, z) pair of an [N, 3] position tensor for a 2-D time-slice grid."""
        assert self.otype in ("HashGrid", "Grid") and len(cols) == self.n_input_dims
        return _ops.HashGridFn.apply(x, self.params, self.table_f16(), self.spec, tuple(cols), _ops.rows_hint(self), _ops.train_context(self), False)

    def extra_repr(self):
        return f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, {self.encoding_config}"


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        cfg = dict(network_config)
        self.network_config = cfg
        if cfg.get("otype", "FullyFusedMLP") not in ("FullyFusedMLP", "CutlassMLP"):
            raise NotImplementedError(f"network otype {cfg.get('otype')!r}")
        if cfg.get("activation", "ReLU") != "ReLU" or cfg.get("output_activation", "None") != "None":
            raise NotImplementedError("only ReLU hidden / linear output MLPs are implemented")
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.spec = _ops.MlpSpec(self.n_input_dims, self.n_output_dims, hidden=int(cfg.get("n_neurons", 64)),
                                 n_hidden=int(cfg.get("n_hidden_layers", 1)))
        if self.spec.hidden != 64 or self.spec.out_cols != 16 or not (1 <= self.spec.n_hidden <= 3) or self.spec.in_cols > 128:
            raise NotImplementedError("MLP shape outside the built kernels (64 neurons, <=16 outputs, 1-3 hidden layers, <=128 inputs)")
        self.dtype = torch.float16
        g = torch.Generator().manual_seed(int(seed))
        chunks = []
        for fan_out, fan_in in self.spec.shapes:  # tcnn: xavier-uniform on the padded shapes
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            chunks.append((torch.rand(fan_out * fan_in, generator=g) * 2.0 - 1.0) * bound)
        self.params = nn.Parameter(torch.cat(chunks))
        self._cache = _HalfCache()

    def weights_f16(self):
        return self._cache.get(self.params)

    def forward(self, x):
        x = x.reshape(-1, self.n_input_dims) if x.dim() != 2 else x
        return _ops.MlpFn.apply(x, self.params, self.weights_f16(), self.spec)

    def extra_repr(self):
        return f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, {self.network_config}"


class NetworkWithInputEncoding(nn.Module):
    """tcnn.NetworkWithInputEncoding: encoding followed by a network (not used by the reference; provided
    because it is part of the same constructor family)."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        super().__init__()
        self.encoding = Encoding(n_input_dims, encoding_config, seed=seed)
        self.network = Network(self.encoding.n_output_dims, n_output_dims, network_config, seed=seed + 1)
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)

    def forward(self, x):
        return self.network(self.encoding(x))
