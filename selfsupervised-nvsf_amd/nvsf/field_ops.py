"""Host-side (PyTorch) entry points of the per-sample field operators and the uniform compositor.

Thin, allocation-only wrappers over the C ABI (include/nvsf_hip.h sections 2-4) plus the autograd glue
the trainer needs.  No arithmetic of the hot path happens in Python: every function here ends in a HIP
kernel launch on the current stream, forward and backward.
"""

import numpy as np
import torch
from torch.autograd import Function

from nvsf import _hip
from nvsf import testing as _testing

W_THRESH = 1e-4  # colour is only evaluated where the compositing weight exceeds this (renderer_dynamic.py:202)


# ------------------------------------------------------------------------------------------------
# multiresolution hash grid
# ------------------------------------------------------------------------------------------------
class GridSpec:
    """Level table of a multiresolution hash grid (Instant-NGP / tcnn "HashGrid" conventions).

    scale_l = exp2(l * log2(per_level_scale)) * base_resolution - 1 ; res_l = ceil(scale_l) + 1 ;
    rows_l = min(round_up(res_l^D, 8), 2^log2_hashmap_size).  Evaluated once, on the host, in fp32.
    """

    def __init__(self, n_dims, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale):
        self.D, self.L, self.F = int(n_dims), int(n_levels), int(n_features)
        self.log2_hashmap_size = int(log2_hashmap_size)
        self.base_resolution = int(base_resolution)
        self.per_level_scale = float(per_level_scale)
        log2_pls = np.float32(np.log2(np.float32(per_level_scale)))
        scales, res, offsets = [], [], [0]
        for l in range(self.L):
            s = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
            r = int(np.ceil(s)) + 1
            rows = min(r ** self.D, 2 ** 31)
            rows = (rows + 7) // 8 * 8
            rows = min(rows, 1 << self.log2_hashmap_size)
            scales.append(float(s))
            res.append(r)
            offsets.append(offsets[-1] + rows)
        self.scales, self.res, self.offsets = scales, res, offsets
        self.n_rows = offsets[-1]
        self.n_params = self.n_rows * self.F
        self.n_output_dims = self.L * self.F
        self.h_scales, self.h_res, self.h_offsets = _hip.host_f32(scales), _hip.host_u32(res), _hip.host_u32(offsets)


def level_major_eligible(spec):
    """Grids nvsf_hashgrid_fwd_level_major is built for: the static hash of the space-time field (3-D, 8 levels x 4 features, every
    level hashed into a power-of-two table)."""
    return (spec.D == 3 and spec.L == 8 and spec.F == 4 and spec.n_params * 2 < 2 ** 31
            and all(int(r) ** 3 > (int(b) - int(a)) and ((int(b) - int(a)) & (int(b) - int(a) - 1)) == 0
                    for r, a, b in zip(spec.res, spec.offsets[:-1], spec.offsets[1:])))


def hashgrid_forward_level_major(x, table_f16, spec):
    """x fp32 [M, >= 3] (columns 0..2) -> fp16 [L, M, F]: the features of hashgrid_forward stored level by level (the one-level-per-XCD
    kernel then writes whole lines; density tail: nvsf_density_dynamic_lm_fwd).  No autograd."""
    x = x.contiguous()
    M = x.shape[0]
    out = torch.empty(spec.L, M, spec.F, dtype=torch.float16, device=x.device)
    _hip.call("nvsf_hashgrid_fwd_level_major", _hip.ptr(x), M, x.shape[1], _hip.ptr(table_f16), spec.L, spec.F, spec.h_scales, spec.h_res,
              spec.h_offsets, _hip.ptr(out))
    return out


def hashgrid_forward(x, cols, table_f16, spec, out=None):
    """x fp32 [M, x_stride]; cols = the D columns of x to encode; table fp16 [n_params] -> fp16 [M, L*F]."""
    x = x.contiguous()
    M, x_stride = x.shape[0], x.shape[1]
    if out is None:
        out = torch.empty(M, spec.n_output_dims, dtype=torch.float16, device=x.device)
    _hip.call("nvsf_hashgrid_fwd", _hip.ptr(x), M, x_stride, _hip.host_u32(cols), spec.D, _hip.ptr(table_f16), spec.L, spec.F,
              spec.h_scales, spec.h_res, spec.h_offsets, _hip.ptr(out), out.stride(0))
    return out


class RowHint:
    """`T`: samples per ray of the rows a renderer is handing to its field right now ([M, ...] rows = T consecutive samples per ray),
    None otherwise.  One object per MODEL (NeRFRenderer creates it and gives the same object to every sub-module, renderer_dynamic.py):
    the autograd nodes of the hash grids read it in `forward`, on the caller's thread, and choose the form of their table scatter by
    it (_bin_from); values and gradients do not depend on it.  Not process-wide: two models in one process do not see each other's."""

    def __init__(self):
        self.T = None

    def rows(self, T):
        return _RowScope(self, T)


class _RowScope:
    def __init__(self, hint, T):
        self.hint, self.T = hint, int(T)

    def __enter__(self):
        self.prev, self.hint.T = self.hint.T, self.T

    def __exit__(self, *exc):
        self.hint.T = self.prev


def rows_hint(module):
    """The samples-per-ray hint of the model `module` belongs to (None outside a renderer's field evaluation)."""
    hint = module.__dict__.get("_row_hint") if module is not None else None
    return None if hint is None else hint.T


def _bin_from(spec, M, rows_per_ray):
    """(merge_from, fine_from) for a batch of M ray-ordered rows, or None: the form of the table scatter (hashgrid_backward)."""
    if not (rows_per_ray and M >= (1 << 18) and M % rows_per_ray == 0) or _testing.get("table_scatter") == "atomic":
        return None
    fine = fine_levels_from(spec, rows_per_ray)
    # Every binned level goes in as RUN SUMS (fine_from = L): measured on the config-2 batches (4096 rays x 768, jittered) the plan
    # (8, 16) takes 1.33 / 1.54 ms (LiDAR / camera) against 1.65 / 1.63 ms for per-row contributions from level 11 on, and on the
    # reference-default grid (0, 8) takes 2.03 / 2.26 ms against 2.62 / 2.52 ms -- a LiDAR step is 0.57 of a finest cell, so even the
    # finest level merges rows there, and where no two rows share a cell the run form costs no more than the per-row form.
    # `fine_levels_from` still decides WHETHER the grid has levels worth binning (hashed, equally sized, cells of a few steps at most).
    return None if fine >= spec.L else (merge_levels_from(spec, fine, rows_per_ray), spec.L)


_BIN_ROWS_MAX = 1 << 23
LEVEL_MAJOR_GRADIENT = True  # the density MLP hands dL/d(features) to the binned scatter level by level ([L, M, F])


def merge_levels_from(spec, fine_from, rows_per_ray):
    """First level whose RUN SUMS go through the bins: the levels whose cells are 1.4 ... 4 ray steps long (res >= T / 4), down to a
    whole group of the run-merging atomic kernel (64 / (8 F) levels per wave).  Measured on the config-2 batches (tools/
    bench_hashgrid_bwd.py, MERGE=...): 8 is the minimum (1.80 ms against 1.88 with the atomics for levels 8-10, 1.94 with no atomics at
    all -- the bin pass has a fixed cost per level and tile that the coarse levels' few contributions do not repay)."""
    first = fine_from
    for l in range(fine_from - 1, -1, -1):
        if int(spec.res[l]) < 0.25 * rows_per_ray:
            break
        first = l
    per_wave = max(1, 64 // (8 * spec.F))
    return min(fine_from, (first + per_wave - 1) // per_wave * per_wave if first % per_wave else first)


def fine_levels_from(spec, rows_per_ray):
    """First level of the trailing run of hashed, equally sized levels whose cells are shorter than ~1.4 steps of a ray that
    crosses the unit cube in `rows_per_ray` samples (no two consecutive samples share a cell there: one contribution per row and
    vertex); spec.L when there is none or the grid has no binned form."""
    if rows_per_ray is None or spec.D != 3 or spec.F not in (2, 4):
        return spec.L
    rows = [int(spec.offsets[l + 1] - spec.offsets[l]) for l in range(spec.L)]
    first = spec.L
    for l in range(spec.L - 1, -1, -1):
        hashed = int(spec.res[l]) ** 3 > rows[l]
        if not (hashed and rows[l] == rows[-1] and int(spec.res[l]) >= 0.7 * rows_per_ray):
            break
        first = l
    if first < spec.L and _hip.hashgrid_bwd_ws_bytes(1 << 20, spec, first, first) == 0:
        return spec.L
    return first


def hashgrid_backward(x, cols, spec, grad_out, grad_table=None, fine_from=None, merge_from=None, ws_pool=None):
    """Scatter-adds d L / d table (fp32 [n_params]) from grad_out ([M, L*F], fp16 or fp32).  `fine_from` < L sends the levels from
    there on through the binned scatter, one contribution per (row, vertex); `merge_from` <= fine_from the levels in between as sums
    over runs of consecutive rows in one cell (nvsf_hashgrid_bwd_binned; fine_levels_from / merge_levels_from).  `fine_from` may be a
    (merge_from, fine_from) pair."""
    if isinstance(fine_from, tuple):
        merge_from, fine_from = fine_from
    x = x.contiguous()
    grad_out = grad_out.contiguous()
    level_major = grad_out.dim() == 3  # [L, M, F] (mlp_backward(grad_x_blocks=F)): only the binned entry point reads it
    if grad_table is None:
        grad_table = torch.zeros(spec.n_params, dtype=torch.float32, device=x.device)
    M = x.shape[0]
    need = 0
    if fine_from is not None and (fine_from < spec.L or (merge_from is not None and merge_from < spec.L)):
        merge_from = fine_from if merge_from is None else min(merge_from, fine_from)
        if M > _BIN_ROWS_MAX and not level_major:  # the fixed-point image of the bins takes 2^26 addends per row: longer batches go in
            for i in range(0, M, _BIN_ROWS_MAX):    # pieces (sums accumulate)
                hashgrid_backward(x[i:i + _BIN_ROWS_MAX], cols, spec, grad_out[i:i + _BIN_ROWS_MAX], grad_table, fine_from, merge_from, ws_pool)
            return grad_table
        # 0: the grid has no binned form: every level through the atomics (first without the run-sum levels: a level of more than
        # 2^20 rows cannot be binned)
        need = _hip.hashgrid_bwd_ws_bytes(M, spec, merge_from, fine_from)
        if need == 0 and merge_from < fine_from < spec.L:
            merge_from = fine_from
            need = _hip.hashgrid_bwd_ws_bytes(M, spec, merge_from, fine_from)
    if need:
        # scratch of the binned scatter, one per stream (scatters issued from two streams may overlap): the caller's pool (a training
        # step shares one between its tables: TrainContext.ws_pool), else kept on the grid's spec (= per encoder)
        pool = ws_pool if ws_pool is not None else spec.__dict__.setdefault("_bin_ws", {})
        key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
        ws = pool.get(key)
        if ws is None or ws.numel() < need:
            ws = pool[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
        _hip.call("nvsf_hashgrid_bwd_binned", _hip.ptr(x), M, x.shape[1], _hip.host_u32(cols), spec.D, spec.L, spec.F, spec.h_scales,
                  spec.h_res, spec.h_offsets, _hip.ptr(grad_out), 1 if grad_out.dtype == torch.float16 else 0,
                  spec.F if level_major else grad_out.stride(0), M * spec.F if level_major else 0,
                  _hip.ptr(grad_table), merge_from, fine_from, _hip.ptr(ws), ws.numel())
        return grad_table
    if level_major:
        raise _hip.NvsfHipError("a level-major gradient needs the binned scatter (nvsf_hashgrid_bwd_binned)")
    _hip.call("nvsf_hashgrid_bwd", _hip.ptr(x), M, x.shape[1], _hip.host_u32(cols), spec.D, spec.L, spec.F, spec.h_scales,
              spec.h_res, spec.h_offsets, _hip.ptr(grad_out), 1 if grad_out.dtype == torch.float16 else 0, grad_out.stride(0),
              _hip.ptr(grad_table))
    return grad_table


class HashGridFn(Function):
    """fp16 features = encode(x; table).  Gradient flows to the (fp32 master) table only."""

    @staticmethod
    def forward(ctx, x, params, table_f16, spec, cols, rows_per_ray=None, train_ctx=None, level_major=False):
        x = x.float().contiguous()
        # level_major: fp16 [L, M, F] instead of rows [M, L F] (hashgrid_forward_level_major; the caller has checked level_major_eligible):
        # the gradient then comes back in the same layout, which is the one the binned scatter reads level by level
        out = hashgrid_forward_level_major(x, table_f16, spec) if level_major else hashgrid_forward(x, cols, table_f16, spec)
        ctx.save_for_backward(x)
        ctx.spec, ctx.cols, ctx.rows_per_ray = spec, cols, rows_per_ray
        ctx.train_ctx, ctx.table_param = train_ctx, params
        if train_ctx is not None and isinstance(params, torch.nn.Parameter) and params.requires_grad and torch.is_grad_enabled():
            train_ctx.expect(params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None, None, None
        fine = _bin_from(ctx.spec, x.shape[0], ctx.rows_per_ray)
        if grad_out.dim() == 3 and (fine is None or _hip.hashgrid_bwd_ws_bytes(x.shape[0], ctx.spec, fine[0], fine[1]) <= 0 or x.shape[0] > _BIN_ROWS_MAX):
            grad_out = grad_out.permute(1, 0, 2).reshape(x.shape[0], -1)  # only the binned entry point reads level-major gradients
        done = scatter_beside_backward(ctx.train_ctx, ctx.table_param, (x, grad_out), lambda view, pool: hashgrid_backward(
            x, ctx.cols, ctx.spec, grad_out, grad_table=view.view(-1), fine_from=fine, ws_pool=pool))
        grad_params = None if done else hashgrid_backward(x, ctx.cols, ctx.spec, grad_out, fine_from=fine)
        return None, grad_params, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# direction encodings
# ------------------------------------------------------------------------------------------------
def freq_encode(x, n_freq=12, out=None):
    """`out`: optional fp16 buffer with >= 2 * n_dims * n_freq columns (the encoding goes to its leading columns)."""
    x = x.float().contiguous()
    M, n_dims = x.shape
    if out is None:
        out = torch.empty(M, 2 * n_dims * n_freq, dtype=torch.float16, device=x.device)
    _hip.call("nvsf_freq_encode", _hip.ptr(x), M, n_dims, n_freq, _hip.ptr_rows(out), out.stride(0))
    return out


def sh4_encode(d01, out=None):
    d01 = d01.float().contiguous()
    M = d01.shape[0]
    if out is None:
        out = torch.empty(M, 16, dtype=torch.float16, device=d01.device)
    _hip.call("nvsf_sh4_encode", _hip.ptr(d01), M, _hip.ptr_rows(out), out.stride(0))
    return out


# ------------------------------------------------------------------------------------------------
# fused MLP
# ------------------------------------------------------------------------------------------------
class MlpSpec:
    """Shapes + fp16 parameter layout of a bias-free ReLU MLP (tcnn "FullyFusedMLP" conventions):
    W0 [hidden][in_cols] ++ (n_hidden-1) x [hidden][hidden] ++ W_out [out_cols][hidden], row-major;
    in_cols = round_up(n_in, 16) (padding inputs read as 1), out_cols = round_up(n_out, 16)."""

    def __init__(self, n_in, n_out, hidden=64, n_hidden=1):
        self.n_in, self.n_out, self.hidden, self.n_hidden = int(n_in), int(n_out), int(hidden), int(n_hidden)
        self.in_cols = (self.n_in + 15) // 16 * 16
        self.out_cols = (self.n_out + 15) // 16 * 16
        self.shapes = [(self.hidden, self.in_cols)] + [(self.hidden, self.hidden)] * (self.n_hidden - 1) + [(self.out_cols, self.hidden)]
        self.n_params = sum(a * b for a, b in self.shapes)

    def split(self, flat):
        mats, o = [], 0
        for a, b in self.shapes:
            mats.append(flat[o:o + a * b].view(a, b))
            o += a * b
        return mats


def _rows(x):
    """x as the kernels want it: fp16 / fp32, unit column stride; a row-strided view (columns of a wider, aligned
    buffer) is passed as it is -- with 16-byte aligned rows the kernels load 8 halves per instruction."""
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    if x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1) or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    return x


def mlp_forward(x, weights_f16, spec, out=None, prefix=None, n_store=None):
    """x [M, n_in] (fp32 or fp16) -> fp32 [M, out_cols] (fp16 operands, fp32 accumulate, output not rounded).
    `out`: optional fp32 destination, 16-byte aligned rows of >= out_cols columns (e.g. a column block of a wider buffer).
    `n_store` in 1..4 (with `out` [M, >= n_store], any 4-byte alignment, unit column stride): only the leading n_store outputs are
    stored -- a head with one or three outputs writes 4 ... 16 bytes per row instead of 64.
    `prefix` = (rows fp16 [G, >= n_cols], rows_per_prefix, n_cols): the first n_cols input columns of row r are
    prefix_rows[r // rows_per_prefix] and x holds the remaining n_in - n_cols columns only (nvsf_mlp_fwd_prefix)."""
    x = _rows(x)
    M = x.shape[0]
    if out is None:
        out = torch.empty(M, spec.out_cols if n_store is None else n_store, dtype=torch.float32, device=x.device)
    cols = spec.out_cols if n_store is None else int(n_store)
    if prefix is not None:
        rows, per, n_cols = prefix
        _hip.call("nvsf_mlp_fwd_prefix", _hip.ptr_rows(rows), rows.stride(0), int(per), int(n_cols), _hip.ptr_rows(x), M, spec.n_in, x.stride(0),
                  _hip.ptr(weights_f16), spec.in_cols, spec.hidden, spec.n_hidden, cols, _hip.ptr_rows(out), out.stride(0))
        return out
    _hip.call("nvsf_mlp_fwd", _hip.ptr_rows(x), 1 if x.dtype == torch.float16 else 0, M, spec.n_in, x.stride(0), _hip.ptr(weights_f16),
              spec.in_cols, spec.hidden, spec.n_hidden, cols, _hip.ptr_rows(out), out.stride(0))
    return out


MLP_GRAD_SCALE = 128.0  # fp16 gradients inside nvsf_mlp_bwd are multiplied by this (tcnn's default loss_scale)


def density_logit_gradient_parts(g_sigma, sigma, g_geo, g_geo_b, n_geo, clamp):
    """(grad_sigma, sigma, geo_a, geo_b, n_geo, lo, hi) for mlp_backward(density_grad=...) when the pieces have the layout
    nvsf_mlp_bwd_density reads (fp32, geometry-gradient rows of >= 16 floats, 16-byte aligned), else None."""
    def rows_ok(t):
        return (t is not None and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= 16
                and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.shape[1] >= n_geo)
    if not (1 <= n_geo <= 15 and rows_ok(g_geo) and (g_geo_b is None or (rows_ok(g_geo_b) and g_geo_b.stride(0) == g_geo.stride(0)))):
        return None
    if sigma is None or sigma.dtype != torch.float32 or not sigma.is_contiguous():
        return None
    if g_sigma is not None and (g_sigma.dtype != torch.float32 or not g_sigma.is_contiguous()):
        return None
    return (g_sigma, sigma, g_geo, g_geo_b, int(n_geo), float(clamp[0]), float(clamp[1]))


def mlp_backward(x, weights_f16, spec, grad_out, need_grad_x=True, grad_scale=MLP_GRAD_SCALE, grad_x=None, gx_col0=0,
                 accumulate=False, prefix=None, grad_x_blocks=0, density_grad=None):
    """One fused kernel: (x, weights, dL/dout [M, n_out] fp32) -> dL/dx fp32 [M, n_in] (or None), dL/dW fp32 [n_params].
    With `grad_x` given (fp32, unit column stride) the input gradient of columns gx_col0.. is written (or added, with
    `accumulate`) there: grad_x[:, j] = dL/dx[:, gx_col0 + j].
    `density_grad` (density_logit_gradient_parts; `grad_out` is then None): dL/dout is formed inside the kernel from
    (grad_sigma, sigma, geometry-gradient rows of one or two heads) -- nvsf_mlp_bwd_density."""
    x = _rows(x)
    M = x.shape[0]
    grad_w = torch.zeros(spec.n_params, dtype=torch.float32, device=x.device)
    mode = (1 if accumulate else 0) | (int(grad_x_blocks) << 8)
    if density_grad is not None:
        g_sigma, sigma, geo_a, geo_b, n_geo, lo, hi = density_grad
        assert prefix is None and spec.n_out == 1 + n_geo
        if grad_x_blocks:
            assert need_grad_x and gx_col0 == 0 and spec.n_in % grad_x_blocks == 0
            if grad_x is None:
                grad_x = torch.empty(spec.n_in // grad_x_blocks, M, grad_x_blocks, dtype=torch.float32, device=x.device)
            gx_ptr, gx_stride = _hip.ptr(grad_x), grad_x_blocks
        else:
            if grad_x is None and need_grad_x:
                n_gx = spec.n_in - gx_col0
                grad_x = torch.empty(M, (n_gx + 3) // 4 * 4, dtype=torch.float32, device=x.device)[:, :n_gx]
            gx_ptr, gx_stride = (None, 0) if grad_x is None else (_hip.ptr_rows(grad_x), grad_x.stride(0))
        _hip.call("nvsf_mlp_bwd_density", _hip.ptr_rows(x), 1 if x.dtype == torch.float16 else 0, M, spec.n_in, x.stride(0), _hip.ptr(weights_f16),
                  spec.in_cols, spec.hidden, spec.n_hidden, spec.out_cols, None if g_sigma is None else _hip.ptr(g_sigma), _hip.ptr(sigma),
                  _hip.ptr_rows(geo_a), None if geo_b is None else _hip.ptr_rows(geo_b), geo_a.stride(0), n_geo, lo, hi, float(grad_scale),
                  gx_ptr, gx_stride, _hip.ptr(grad_w), int(gx_col0), mode)
        return grad_x, grad_w
    grad_out = grad_out.float()
    if grad_out.dim() != 2 or (grad_out.shape[1] > 1 and grad_out.stride(1) != 1):
        grad_out = grad_out.contiguous()
    if grad_x_blocks:  # dL/dx as column blocks [n_in / B, M, B] (the gradient of a hash grid's features, level by level)
        assert need_grad_x and gx_col0 == 0 and spec.n_in % grad_x_blocks == 0 and prefix is None
        if grad_x is None:
            grad_x = torch.empty(spec.n_in // grad_x_blocks, M, grad_x_blocks, dtype=torch.float32, device=x.device)
        _hip.call("nvsf_mlp_bwd", _hip.ptr_rows(x), 1 if x.dtype == torch.float16 else 0, M, spec.n_in, x.stride(0), _hip.ptr(weights_f16),
                  spec.in_cols, spec.hidden, spec.n_hidden, spec.out_cols, _hip.ptr_rows(grad_out), grad_out.shape[1], grad_out.stride(0),
                  float(grad_scale), _hip.ptr(grad_x), grad_x_blocks, _hip.ptr(grad_w), 0, mode)
        return grad_x, grad_w
    if grad_x is None and need_grad_x:
        n_gx = spec.n_in - gx_col0  # rows padded to four floats: the kernel then stores 16 bytes per lane
        grad_x = torch.empty(M, (n_gx + 3) // 4 * 4, dtype=torch.float32, device=x.device)[:, :n_gx]
    if prefix is not None:  # rows with a shared prefix (see mlp_forward)
        rows, per, n_cols = prefix
        _hip.call("nvsf_mlp_bwd_prefix", _hip.ptr_rows(rows), rows.stride(0), int(per), int(n_cols), _hip.ptr_rows(x), M, spec.n_in, x.stride(0),
                  _hip.ptr(weights_f16), spec.in_cols, spec.hidden, spec.n_hidden, spec.out_cols, _hip.ptr_rows(grad_out), grad_out.shape[1],
                  grad_out.stride(0), float(grad_scale), None if grad_x is None else _hip.ptr_rows(grad_x), 0 if grad_x is None else grad_x.stride(0),
                  _hip.ptr(grad_w), int(gx_col0), 1 if accumulate else 0)
        return grad_x, grad_w
    _hip.call("nvsf_mlp_bwd", _hip.ptr_rows(x), 1 if x.dtype == torch.float16 else 0, M, spec.n_in, x.stride(0), _hip.ptr(weights_f16),
              spec.in_cols, spec.hidden, spec.n_hidden, spec.out_cols, _hip.ptr_rows(grad_out), grad_out.shape[1], grad_out.stride(0),
              float(grad_scale), None if grad_x is None else _hip.ptr_rows(grad_x), 0 if grad_x is None else grad_x.stride(0),
              _hip.ptr(grad_w), int(gx_col0), 1 if accumulate else 0)
    return grad_x, grad_w


class MaskedSigmoidFn(Function):
    """out = mask ? sigmoid(logits) : 0, one launch on the strided logits view (torch: a strided sigmoid + a broadcast multiply);
    the backward is sigmoid's own (g * out * (1 - out) vanishes on masked rows by itself)."""

    @staticmethod
    def forward(ctx, logits, mask):
        if logits.dtype != torch.float32 or logits.dim() != 2:
            logits = logits.float().reshape(logits.shape[0], -1)
        M, C = logits.shape
        out = torch.empty(M, C, dtype=torch.float32, device=logits.device)
        mk = None if mask is None else mask.reshape(-1).to(torch.uint8 if mask.dtype != torch.bool else torch.bool).contiguous()
        _hip.call("nvsf_masked_sigmoid", logits.data_ptr(), logits.stride(0), logits.stride(1) if C > 1 else 1,
                  None if mk is None else mk.data_ptr(), M, C, _hip.ptr(out))
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = g.float().contiguous()
        grad = torch.empty_like(out)
        _hip.call("nvsf_sigmoid_bwd", _hip.ptr(g), _hip.ptr(out), out.numel(), _hip.ptr(grad))
        return grad, None


# ---- dense / sparse choice of the per-sample heads without a host sync per call ------------------------------------------------
# `color` evaluates the heads on all samples (and zeroes the masked ones) when at least a quarter of them pass the weight
# threshold, otherwise on the gathered subset.  The count that decides this lives on the device.  While a modality stays in the
# dense regime its count is copied to pinned host memory asynchronously and looked at one call later; only the sparse regime
# (whose boolean indexing synchronises anyway) reads it immediately.
def mask_was_dense(model, lidar, mask):
    st = model.__dict__.setdefault("_mask_regime", {}).setdefault(bool(lidar), {"dense": False, "pending": None})
    if st["pending"] is not None:
        host, event, numel = st["pending"]
        if event.query():  # the copy has landed: adopt the regime it shows
            st["dense"] = 4 * int(host[0]) >= numel
            st["pending"] = None
    if not st["dense"] or not mask.is_cuda:
        return False
    if st["pending"] is None:  # keep observing: this batch's count, read by a later call
        host = st.get("host")
        if host is None:
            host = st["host"] = torch.empty(1, dtype=torch.int64).pin_memory()  # allocated once per (model, modality)
        host.copy_(count_true(mask), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        st["pending"] = (host, event, mask.numel())
    return True


def count_true(mask):
    """Population of a bool / uint8 mask as a one-element int64 device tensor, one HIP launch (`mask.sum()` is an element-wise int64
    reduction in torch: 0.21 ms for the 3.1 M samples of a batch)."""
    mk = mask.reshape(-1)
    if not mk.is_contiguous() or mk.data_ptr() % 16:
        mk = mk.clone()
    out = torch.empty(1, dtype=torch.int64, device=mask.device)
    _hip.call("nvsf_count_nonzero_u8", mk.data_ptr(), mk.numel(), _hip.ptr(out))
    return out


def note_mask_count(model, lidar, n_active, numel):
    st = model.__dict__.setdefault("_mask_regime", {}).setdefault(bool(lidar), {"dense": False, "pending": None})
    st["dense"] = 4 * n_active >= numel
    st["pending"] = None


def heads(model, d01, geo_feat, cal_lidar_color, ray_dirs01=None, geo16=None):
    """Logits of the per-sample heads of `model` (a NeRFNetwork / NeRFNetworkStatic): [M, 2] = [raydrop, intensity] for
    LiDAR samples, [M, 3] colour logits otherwise.  d01: directions mapped to [0, 1]; geo_feat: [M, geo_feat_dim].
    ray_dirs01 [N, 3] (with M = N * T, sample rows ordered ray by ray): the directions per RAY instead of d01 -- the
    encoding is then evaluated N times and broadcast to the samples (same values: every sample of a ray has its direction).
    geo16: fp16 [M, 16] rows (geometry features, 1.0) that already hold fp16(geo_feat) -- the fused training forward writes them
    (DensityRaysFn) -- and serve as the MLP kernels' per-sample input as they are."""
    if cal_lidar_color:
        net_a, net_b, enc = model.raydrop_net, model.intensity_net, model.view_encoder_lidar
    else:
        net_a, net_b, enc = model.color_net, None, model.view_encoder_camera
    spec = net_a.spec
    M = geo_feat.shape[0]
    n_enc_cols = enc.n_output_dims
    # per-ray direction rows read by the MLP kernels themselves (shared-prefix rows): no [M, in_cols] input matrix at all
    prefix_mode = (ray_dirs01 is not None and n_enc_cols % 8 == 0 and M % ray_dirs01.shape[0] == 0 and (M // ray_dirs01.shape[0]) % 16 == 0
                   and spec.n_in - n_enc_cols <= 16 and spec.n_hidden <= 2 and _testing.get("heads_input") == "prefix")  # tests: "rows" =
    # the assembled [M, in_cols] input
    buf = None if prefix_mode else torch.empty(M, spec.in_cols, dtype=torch.float16, device=geo_feat.device)
    with torch.no_grad():
        src = d01 if ray_dirs01 is None else ray_dirs01
        dst = buf if ray_dirs01 is None else torch.empty(src.shape[0], (n_enc_cols + 7) // 8 * 8, dtype=torch.float16, device=geo_feat.device)
        if enc.otype == "Frequency":
            freq_encode(src, enc.n_frequencies, out=dst)
        else:
            sh4_encode(src, out=dst)
    enc_ray = None
    if ray_dirs01 is not None:
        assert M % src.shape[0] == 0
        if n_enc_cols % 8 == 0:
            enc_ray = dst  # HeadsFn writes whole rows (encoding | geometry | ones) in one pass
        else:
            _hip.call("nvsf_repeat_rows_f16", _hip.ptr(dst), src.shape[0], n_enc_cols, dst.stride(0), M // src.shape[0], _hip.ptr(buf),
                      buf.stride(0))
    x16 = geo16 if (prefix_mode and geo16 is not None and geo_feat.shape[1] == 15 and tuple(geo16.shape) == (M, 16)) else None
    if net_b is None:
        return HeadsFn.apply(buf, enc.n_output_dims, geo_feat, net_a.params, net_a.weights_f16(), spec, None, None, enc_ray, x16)
    return HeadsFn.apply(buf, enc.n_output_dims, geo_feat, net_a.params, net_a.weights_f16(), spec, net_b.params, net_b.weights_f16(), enc_ray, x16)


class HeadsFn(Function):
    """h = [MLP_a(u) | MLP_b(u)] for the per-sample heads (network_dynamic.py:297-330: colour, or ray-drop + intensity),
    u = [direction encoding | geometry features | tcnn's ones padding].

    `buf` is a caller-allocated fp16 [M, in_cols] buffer whose first n_enc columns already hold the direction encoding
    (no parameters, no gradient); the geometry features are cast into the following columns here.  Rows are 16-byte
    aligned, so both MLP kernels read 8 halves per load (the reference's torch.cat gives 87- / 31-column rows: element
    loads), the heads share one input, and the backward produces only the gradient the graph needs -- the n_geo geometry
    columns, summed over the heads inside the kernel -- instead of two [M, n_in] fp32 matrices that are sliced and added
    afterwards."""

    @staticmethod
    def forward(ctx, buf, n_enc, geo, params_a, w16_a, spec, params_b=None, w16_b=None, enc_ray=None, x16_ready=None):
        n_geo = geo.shape[1]
        if buf is None:  # shared-prefix rows: x = [geometry | ones] fp16 [M, 16], the encoding stays one row per ray
            assert enc_ray is not None and n_enc + n_geo == spec.n_in
            N, M = enc_ray.shape[0], geo.shape[0]
            if x16_ready is not None:
                x16 = x16_ready
            else:
                g = geo if (geo.dtype in (torch.float16, torch.float32) and geo.stride(1) == 1) else geo.float().contiguous()
                x16 = torch.empty(M, 16, dtype=torch.float16, device=geo.device)
                _hip.call("nvsf_heads_input_f16", _hip.ptr(enc_ray), N, 0, enc_ray.stride(0), M // N, _hip.ptr_rows(g),
                          1 if g.dtype == torch.float16 else 0, n_geo, g.stride(0), _hip.ptr(x16), 16, 16)
            prefix = (enc_ray, M // N, n_enc)
            ctx.save_for_backward(x16, w16_a, w16_b, enc_ray)
            ctx.spec, ctx.n_enc, ctx.n_geo, ctx.geo_dtype, ctx.prefix_rows = spec, n_enc, n_geo, geo.dtype, M // N
            return HeadsFn._logits(x16, w16_a, w16_b, spec, prefix)
        assert n_enc + n_geo == spec.n_in and buf.shape[1] >= spec.n_in and buf.dtype == torch.float16
        ctx.prefix_rows = None
        if enc_ray is not None:  # per-ray encoding [N, >= n_enc] fp16: rows assembled whole (nvsf_heads_input_f16)
            g = geo if (geo.dtype in (torch.float16, torch.float32) and geo.stride(1) == 1) else geo.float().contiguous()
            N = enc_ray.shape[0]
            _hip.call("nvsf_heads_input_f16", _hip.ptr(enc_ray), N, n_enc, enc_ray.stride(0), buf.shape[0] // N, _hip.ptr_rows(g),
                      1 if g.dtype == torch.float16 else 0, n_geo, g.stride(0), _hip.ptr(buf), spec.in_cols, buf.stride(0))
        else:
            cast_cols_f16(geo, buf[:, n_enc:n_enc + n_geo])
        u = buf[:, :spec.n_in]
        ctx.save_for_backward(buf, w16_a, w16_b)
        ctx.spec, ctx.n_enc, ctx.n_geo, ctx.geo_dtype = spec, n_enc, n_geo, geo.dtype
        return HeadsFn._logits(u, w16_a, w16_b, spec, None)

    @staticmethod
    def _logits(u, w16_a, w16_b, spec, prefix):
        """[M, n_out] (one head) or [M, 2 n_out] = [head a | head b]: each head stores only its n_out <= 4 logits, the two heads
        interleaved in one buffer (no 16-column fp32 blocks of which one column is used, no torch.cat pass over the samples)."""
        M, n = u.shape[0], spec.n_out
        if n > 4:
            raise _hip.NvsfHipError("per-sample heads with more than 4 outputs are not supported")
        if w16_b is None:
            width = 4 if n == 3 else n  # colour: 16-byte rows, the view drops the padding column
            return mlp_forward(u, w16_a, spec, out=torch.empty(M, width, dtype=torch.float32, device=u.device), prefix=prefix, n_store=n)[:, :n]
        both = torch.empty(M, 2 * n, dtype=torch.float32, device=u.device)
        mlp_forward(u, w16_a, spec, out=both[:, :n], prefix=prefix, n_store=n)
        mlp_forward(u, w16_b, spec, out=both[:, n:], prefix=prefix, n_store=n)
        return both

    @staticmethod
    def backward(ctx, grad_h):
        prefix = None
        if ctx.prefix_rows is not None:
            buf, w16_a, w16_b, enc_ray = ctx.saved_tensors
            prefix = (enc_ray, ctx.prefix_rows, ctx.n_enc)
            u = buf
        else:
            buf, w16_a, w16_b = ctx.saved_tensors
            u = buf[:, :ctx.spec.n_in]
        spec, n_enc, n_geo = ctx.spec, ctx.n_enc, ctx.n_geo
        grad_h = grad_h.float().contiguous()
        M = buf.shape[0]
        need_geo = ctx.needs_input_grad[2]
        grad_geo = torch.empty(M, (n_geo + 3) // 4 * 4, dtype=torch.float32, device=buf.device)[:, :n_geo] if need_geo else None
        _, gw_a = mlp_backward(u, w16_a, spec, grad_h[:, :spec.n_out], need_grad_x=need_geo, grad_x=grad_geo, gx_col0=n_enc, prefix=prefix)
        gw_b = None
        if w16_b is not None:
            _, gw_b = mlp_backward(u, w16_b, spec, grad_h[:, spec.n_out:2 * spec.n_out], need_grad_x=need_geo, grad_x=grad_geo,
                                   gx_col0=n_enc, accumulate=True, prefix=prefix)
        if grad_geo is not None and grad_geo.dtype != ctx.geo_dtype:
            grad_geo = grad_geo.to(ctx.geo_dtype)
        return (None, None, grad_geo, gw_a if ctx.needs_input_grad[3] else None, None, None,
                gw_b if (w16_b is not None and ctx.needs_input_grad[6]) else None, None, None, None)


class MlpFn(Function):
    """out = MLP(x; params).  Forward: one fused MFMA kernel.  Backward: one fused MFMA kernel (nvsf_mlp_bwd:
    activations recomputed from the saved input, data and weight gradients in the same launch)."""

    @staticmethod
    def forward(ctx, x, params, weights_f16, spec):
        out = mlp_forward(x, weights_f16, spec)
        ctx.save_for_backward(x, weights_f16)
        ctx.spec = spec
        return out[:, :spec.n_out]

    @staticmethod
    def backward(ctx, grad_out):
        x, weights_f16 = ctx.saved_tensors
        spec = ctx.spec
        if not (spec.n_hidden <= 2 and spec.hidden == 64 and spec.out_cols == 16):
            raise _hip.NvsfHipError("nvsf_mlp_bwd is built for 64-wide MLPs with one or two hidden layers and <= 16 outputs "
                                    "(every network of the reference's model); there is no fallback")
        grad_x, grad_w = mlp_backward(x, weights_f16, spec, grad_out, need_grad_x=ctx.needs_input_grad[0])
        if grad_x is not None and grad_x.dtype != x.dtype:
            grad_x = grad_x.to(x.dtype)
        return grad_x, (grad_w if ctx.needs_input_grad[1] else None), None, None


def cast_cols_f16(src, dst):
    """dst[:, j] = fp16(src[:, j]) for row-strided 2-D views (src fp32 / fp16, dst fp16): one streaming kernel instead of
    torch's strided element-wise copy."""
    if src.dtype not in (torch.float16, torch.float32) or src.dim() != 2 or (src.shape[1] > 1 and src.stride(1) != 1):
        src = src.float().contiguous()
    M, n = src.shape
    _hip.call("nvsf_cast_cols_f16", _hip.ptr_rows(src), 1 if src.dtype == torch.float16 else 0, M, n, src.stride(0), _hip.ptr_rows(dst),
              dst.stride(0))
    return dst


# ---- table-gradient scatter beside the rest of backward -------------------------------------------------------------------------
# The table scatter (coarse levels at the memory-side float-atomic rate, fine levels as two streaming passes: ~2 ms per config-2 batch)
# with the vector and matrix pipes idle; the MLP backward kernels of the OTHER modality's branch are MFMA / LDS work.  DensityFn
# therefore issues the scatter on a side stream: the main stream goes on with the next branch of backward and only the consumers
# of the table gradient (gradient all-reduce, overflow check, optimiser) wait for it (`sync_side_streams`).
# Only a caller that synchronises afterwards turns this on (`TrainContext.overlap` around backward, then `sync_side_streams()`:
# nvsf.nerf.train_step.RenderTrainStep.step); everywhere else the scatter stays on the calling stream.
_SIDE_STREAMS = {}  # (device) -> the stream table scatters are issued on beside backward (a device resource, shared by design)


class LocalGradSink:
    """Single-process destination of the side-stream table scatters: the parameter's own `.grad`, zero-filled on the CALLING
    (main) stream when the first scatter of a step asks for it.  Every scatter of that table then accumulates into this one
    buffer on the side stream and the autograd node returns no tensor for the table: nothing the autograd engine could clone,
    add or read on the main stream while the side stream still writes (a table that receives more than one DensityFn.backward
    per pass -- RenderTrainStep(ray_chunks > 1) -- would otherwise be summed by the engine without any wait on the side stream)."""

    def __init__(self):
        self.side_scatters = []  # (parameter, event recorded on the side stream behind its LAST scatter of the step), in issue order

    def view_for(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32)
        return p.grad

    def mark_ready(self, p):
        """Called on the stream that ran the last scatter of table `p` in this step, right behind it."""
        stream = torch.cuda.current_stream(p.device)
        if p.is_cuda and stream != torch.cuda.default_stream(p.device):
            ev = torch.cuda.Event()
            ev.record(stream)
            self.side_scatters.append((p, ev))


class TrainContext:
    """What the table-scatter nodes (DensityFn / DensityRaysFn) need to know about the training step they run in -- owned by the step
    object (RenderTrainStep.train_ctx), attached to its model (`model._train_ctx`), read by the nodes at `forward` and carried to
    `backward` on the autograd ctx.  Nothing here is process-wide: two steps / two models in one process keep separate state.
      sink     where table gradients are scattered: frame_shard.GradBuckets (multi-rank) or a LocalGradSink; None: the node returns
               the gradient tensor to autograd
      overlap  issue the scatters on the side stream beside the rest of backward; only a caller that waits for them afterwards
               turns this on (RenderTrainStep, around backward)
      left     scatters a table still has to receive in the running step (counted at forward): a sink may only be told that a
               table's gradient is final -- and send its bucket to the all-reduce -- after the LAST of them."""

    def __init__(self):
        self.sink, self.overlap, self.left = None, False, None
        self.ws_pool = {}  # (device, stream) -> scratch of the binned table scatters of this step's tables

    def begin_step(self):
        self.left = {}

    def end_step(self):
        self.left = None

    def expect(self, p):
        if self.left is not None and p is not None:
            self.left[p] = self.left.get(p, 0) + 1

    def done(self, p):
        """True when this was the last outstanding scatter of table `p` in the step (always, outside a counted step)."""
        if self.left is None or p not in self.left:
            return True
        self.left[p] -= 1
        return self.left[p] <= 0


def train_context(module):
    return module.__dict__.get("_train_ctx") if module is not None else None


def scatter_beside_backward(tctx, param, tensors, scatter):
    """Runs `scatter(view, scratch pool)` -- which ADDS a table gradient into `view` -- on the side stream of the training step `tctx`
    belongs to, into the step's gradient sink (the parameter's .grad or its bucket view); the autograd node then returns no tensor
    for the table.  Returns False (nothing done) outside such a step: the caller scatters on its own stream and returns the
    gradient to autograd.  `tensors`: what the scatter reads (kept alive for the side stream)."""
    if tctx is None or not tctx.overlap or tctx.sink is None or not isinstance(param, torch.nn.Parameter) or not param.is_cuda:
        if tctx is not None and isinstance(param, torch.nn.Parameter):
            tctx.done(param)
        return False
    last = tctx.done(param)
    view = tctx.sink.view_for(param)  # obtained (and, the first time, zero-filled) on the main stream
    if view is None:
        raise _hip.NvsfHipError("the gradient sink has no buffer for this table")
    main, side = torch.cuda.current_stream(param.device), side_stream(param.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for t in tensors:
            t.record_stream(side)
        scatter(view, tctx.ws_pool)
        if last:
            tctx.sink.mark_ready(param)
    return True


def scatter_beside_backward_multi(tctx, params, tensors, scatter):
    """scatter_beside_backward for a node whose ONE scatter feeds several parameters (the time slices of the space-time grids: one
    launch sums G[row] per coordinate pair, every slice's gradient is a multiple of it): `scatter(views, scratch pool)` adds into
    `views[i]` = the sink's buffer of `params[i]` (distinct parameters), on the side stream.  False: nothing done (see above)."""
    ok = (tctx is not None and tctx.overlap and tctx.sink is not None
          and all(isinstance(p, torch.nn.Parameter) and p.is_cuda and p.dtype == torch.float32 for p in params))
    if not ok:
        if tctx is not None:
            for p in params:
                if isinstance(p, torch.nn.Parameter):
                    tctx.done(p)
        return False
    last = [tctx.done(p) for p in params]
    views = [tctx.sink.view_for(p) for p in params]  # obtained (and, the first time, zero-filled) on the main stream
    if any(v is None for v in views):
        raise _hip.NvsfHipError("the gradient sink has no buffer for one of these tables")
    main, side = torch.cuda.current_stream(params[0].device), side_stream(params[0].device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for t in tensors:
            t.record_stream(side)
        scatter(views, tctx.ws_pool)
        for p, is_last in zip(params, last):
            if is_last:
                tctx.sink.mark_ready(p)
    return True


def side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def sync_side_streams():
    """The calling stream waits for every table scatter issued so far."""
    for s in _SIDE_STREAMS.values():
        torch.cuda.current_stream(s.device).wait_stream(s)


class DensityFn(Function):
    """(sigma, geo_feat) = split(MLP(encode(x01)))  --  `NeRFNetwork.density` of a static hash field as ONE autograd node
    (network_dynamic.py:213-287 without the space-time terms: hash grid -> sigma_net -> trunc_exp / slice).

    Same forward kernels as the operator chain (HashGridFn -> MlpFn -> trunc_exp); what changes is the backward: the
    gradient of the logits is assembled by one streaming kernel (nvsf_sigma_geo_bwd) instead of autograd's zero-fill /
    strided-copy / add chain for the two slices, and the MLP's input gradient goes to the table scatter in fp32 as it
    leaves nvsf_mlp_bwd (the operator chain rounds it to fp16 in between, a pass over [M, 32])."""

    @staticmethod
    def forward(ctx, x01, table_params, table_f16, grid_spec, mlp_params, mlp_w16, mlp_spec, sigma_lo, sigma_hi, rows_per_ray=None, train_ctx=None):
        x01 = x01.float().contiguous()
        feat = hashgrid_forward(x01, (0, 1, 2), table_f16, grid_spec)
        h = mlp_forward(feat, mlp_w16, mlp_spec)
        sigma = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
        _hip.call("nvsf_exp_col", _hip.ptr(h), h.stride(0), 0, h.shape[0], _hip.ptr(sigma))
        ctx.save_for_backward(x01, feat, sigma, mlp_w16)
        ctx.grid_spec, ctx.mlp_spec, ctx.clamp = grid_spec, mlp_spec, (float(sigma_lo), float(sigma_hi))
        ctx.table_param = table_params
        ctx.rows_per_ray, ctx.train_ctx = rows_per_ray, train_ctx
        ctx.need_table, ctx.need_w = ctx.needs_input_grad[1], ctx.needs_input_grad[4]
        if train_ctx is not None and table_params is not None and table_params.requires_grad and torch.is_grad_enabled():
            train_ctx.expect(table_params)
        return sigma, h[:, 1:mlp_spec.n_out]

    @staticmethod
    def backward(ctx, g_sigma, g_geo):
        return (None,) + _density_backward(ctx, g_sigma, g_geo) + (None, None)


def _density_backward(ctx, g_sigma, g_geo):
    """Backward shared by DensityFn and DensityRaysFn: -> (grad_table, None, None, grad_mlp_weights, None, None, None, None)."""
    x01, feat, sigma, mlp_w16 = ctx.saved_tensors
    spec, M = ctx.mlp_spec, x01.shape[0]
    if g_sigma is not None:
        g_sigma = g_sigma.float().contiguous()
    if g_geo is not None and (g_geo.dtype != torch.float32 or g_geo.stride(1) != 1):
        g_geo = g_geo.float().contiguous()
    # the logit gradient [g_sigma clamp(sigma) | g_geo (+ g_geo_b)] is formed inside the MLP backward where the pieces have the layout it
    # reads (nvsf_mlp_bwd_density); otherwise (and as the test reference, testing.variant(density_grad="matrix")) by a pass of its own
    g_geo_b = getattr(ctx, "g_geo_b", None)
    parts = None
    if _testing.get("density_grad") == "composed" and x01.is_cuda:
        parts = density_logit_gradient_parts(g_sigma, sigma.view(-1), g_geo, g_geo_b, spec.n_out - 1, ctx.clamp)
    grad_h = None
    if parts is None:
        if g_geo_b is not None:
            g_geo = g_geo + g_geo_b
        grad_h = torch.empty(M, 16, dtype=torch.float32, device=x01.device)
        _hip.call("nvsf_sigma_geo_bwd", None if g_sigma is None else _hip.ptr(g_sigma), _hip.ptr(sigma),
                  None if g_geo is None else _hip.ptr_rows(g_geo), 0 if g_geo is None else g_geo.stride(0), spec.n_out - 1, M,
                  _hip.ptr(grad_h), 16, ctx.clamp[0], ctx.clamp[1])
    need_table, need_w = ctx.need_table, ctx.need_w
    # levels whose cells are shorter than a few ray steps go through the binned scatter (rows are ray-ordered with a known ray length);
    # the MLP then hands its input gradient over level by level ([L, M, F]): every pass of the scatter reads one contiguous column
    fine = _bin_from(ctx.grid_spec, M, getattr(ctx, "rows_per_ray", None)) if need_table else None
    blocks = ctx.grid_spec.F if (LEVEL_MAJOR_GRADIENT and fine is not None and M <= _BIN_ROWS_MAX and spec.n_in == ctx.grid_spec.L * ctx.grid_spec.F
                                 and feat.dtype == torch.float16 and feat.stride(0) % 8 == 0
                                 and _hip.hashgrid_bwd_ws_bytes(M, ctx.grid_spec, fine[0], fine[1]) > 0) else 0
    grad_feat, grad_w = mlp_backward(feat, mlp_w16, spec, None if grad_h is None else grad_h[:, :spec.n_out], need_grad_x=need_table,
                                     grad_x_blocks=blocks, density_grad=parts)
    grad_table = None
    if need_table:
        tctx = getattr(ctx, "train_ctx", None)
        last = tctx.done(ctx.table_param) if tctx is not None else True
        sink = tctx.sink if tctx is not None else None
        pool = tctx.ws_pool if tctx is not None else None
        if not (tctx is not None and tctx.overlap and x01.is_cuda):
            view = sink.view_for(ctx.table_param) if sink is not None else None
            if view is not None:
                hashgrid_backward(x01, (0, 1, 2), ctx.grid_spec, grad_feat, grad_table=view.view(-1), fine_from=fine, ws_pool=pool)
                if last:
                    sink.mark_ready(ctx.table_param)
            else:
                grad_table = hashgrid_backward(x01, (0, 1, 2), ctx.grid_spec, grad_feat, fine_from=fine, ws_pool=pool)
        else:
            # Side stream.  The destination is ONE buffer per table and step, obtained (and, the first time, zero-filled) on the
            # main stream BEFORE the side stream is made to wait for it: a bucket view (multi-rank) or the parameter's .grad
            # (LocalGradSink).  The node returns no tensor for the table, so the autograd engine never touches the buffer.
            if sink is None:
                raise _hip.NvsfHipError("TrainContext.overlap needs a gradient sink (RenderTrainStep sets TrainContext.sink)")
            view = sink.view_for(ctx.table_param)
            if view is None:
                raise _hip.NvsfHipError("the gradient sink has no buffer for this table")
            main, side = torch.cuda.current_stream(x01.device), side_stream(x01.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                x01.record_stream(side)
                grad_feat.record_stream(side)
                hashgrid_backward(x01, (0, 1, 2), ctx.grid_spec, grad_feat, grad_table=view.view(-1), fine_from=fine, ws_pool=pool)
                if last:  # the table's gradient is final: its bucket may go out (event recorded on the side stream)
                    sink.mark_ready(ctx.table_param)
    return grad_table, None, None, (grad_w if need_w else None), None, None, None, None


def density_uniform_train_forward(rays_o, rays_d, nears, fars, T, aabb_host, bound, noise, table_f16, grid_spec, mlp_w16, sliced):
    """nvsf_field_density_uniform_train_fwd: rays -> z_vals [N, T], sigma [N T], geo16 [N T, 16] fp16 and what the backward reads:
    normalised positions x01 [N T, 3], feature rows [N T, 32] fp16, MLP outputs h32 [N T, 16] fp32.  `sliced`: the level-sliced
    (XCD-aware) encode pass + streaming MLP pass instead of the one-launch gather form; same values bit for bit."""
    N, dev = rays_o.shape[0], rays_o.device
    M = N * T
    z_vals = torch.empty(N, T, dtype=torch.float32, device=dev)
    sigma = torch.empty(M, dtype=torch.float32, device=dev)
    geo16 = torch.empty(M, 16, dtype=torch.float16, device=dev)
    x01 = torch.empty(M, 3, dtype=torch.float32, device=dev)
    feat = torch.empty(M, 32, dtype=torch.float16, device=dev)
    h32 = torch.empty(M, 16, dtype=torch.float32, device=dev)
    planes = torch.empty(16, M, dtype=torch.int32, device=dev) if sliced else None  # 8 planes of 8 bytes per sample
    _hip.call("nvsf_field_density_uniform_train_fwd", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars),
              _hip.ptr(linspace01(T, dev)), _hip.ptr(noise), _hip.host_f32(aabb_host), float(bound), N, T, _hip.ptr(table_f16),
              grid_spec.L, grid_spec.F, grid_spec.h_scales, grid_spec.h_res, grid_spec.h_offsets, _hip.ptr(mlp_w16), _hip.ptr(z_vals),
              _hip.ptr(sigma), _hip.ptr(geo16), _hip.ptr(x01), _hip.ptr(feat), _hip.ptr(h32), _hip.ptr(planes))
    return z_vals, sigma, geo16, x01, feat, h32


class DensityRaysFn(Function):
    """The training forward of a static hash field from the RAYS: z_vals [N, T], sigma [N T], geo_feat [N T, 15] (fp32) and --
    not differentiable -- geo16 [N T, 16] fp16 = (fp16(geo_feat), 1.0), the per-sample input rows of the heads.

    One launch (nvsf_field_density_uniform_train_fwd; two in its level-sliced form) where the operator chain runs the sampler, the
    unit-cube normalisation (two torch passes over [M, 3]), the hash-grid encoder, the density MLP and the exponential as seven:
    sample positions are formed in registers, the encoded features go from the gathers into the MFMA operand, and what the
    backward needs (positions, feature rows, outputs) is written once.  Same values as the chain (same kernels' arithmetic:
    tests/test_train_step_gpu.py); the backward is DensityFn's."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, nears, fars, T, aabb_host, bound, noise, table_params, table_f16, grid_spec, mlp_params, mlp_w16,
                mlp_spec, sigma_lo, sigma_hi, sliced, train_ctx=None):
        z_vals, sigma, geo16, x01, feat, h32 = density_uniform_train_forward(rays_o, rays_d, nears, fars, T, aabb_host, bound, noise, table_f16,
                                                                             grid_spec, mlp_w16, sliced)
        ctx.save_for_backward(x01, feat, sigma, mlp_w16)
        ctx.grid_spec, ctx.mlp_spec, ctx.clamp = grid_spec, mlp_spec, (float(sigma_lo), float(sigma_hi))
        ctx.table_param = table_params
        ctx.need_table, ctx.need_w = ctx.needs_input_grad[8], ctx.needs_input_grad[11]
        ctx.rows_per_ray, ctx.train_ctx = T, train_ctx
        if train_ctx is not None and table_params is not None and table_params.requires_grad and torch.is_grad_enabled():
            train_ctx.expect(table_params)
        ctx.mark_non_differentiable(z_vals, geo16)
        return z_vals, sigma, h32[:, 1:mlp_spec.n_out], geo16

    @staticmethod
    def backward(ctx, _g_z, g_sigma, g_geo, _g_geo16):
        g = _density_backward(ctx, g_sigma, g_geo)  # (grad_table, None, None, grad_w, ...)
        return (None,) * 8 + (g[0], None, None, g[3], None, None, None, None, None, None)


def render_uniform_train_forward(rays_o, rays_d, nears, fars, T, aabb_host, bound, noise, table_f16, grid_spec, sigma_w16, lidar, head_a_w16,
                                 head_b_w16, k_scale, bg_host, w_thresh, sliced):
    """nvsf_render_uniform_train_fwd: rays -> z_vals, weights [N, T], weights_sum, depth [N], image [N, C] and what the backward of the
    render reads: x01 [M, 3], feature rows [M, 32] fp16, geo16 [M, 16] fp16, sigma [M], masked per-sample colours [M, C].  `sliced`: the
    level-sliced encode pass + streaming tail instead of the one-launch gather form; same values bit for bit."""
    N, dev = rays_o.shape[0], rays_o.device
    M, C = N * T, (2 if lidar else 3)
    f32 = dict(dtype=torch.float32, device=dev)
    z_vals, weights = torch.empty(N, T, **f32), torch.empty(N, T, **f32)
    ws, depth, image = torch.empty(N, **f32), torch.empty(N, **f32), torch.empty(N, C, **f32)
    x01, sigma, rgbs = torch.empty(M, 3, **f32), torch.empty(M, **f32), torch.empty(M, C, **f32)
    feat = torch.empty(M, 32, dtype=torch.float16, device=dev)
    geo16 = torch.empty(M, 16, dtype=torch.float16, device=dev)
    planes = torch.empty(16, M, dtype=torch.int32, device=dev) if sliced else None  # 8 planes of 8 bytes per sample
    _hip.call("nvsf_render_uniform_train_fwd", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(linspace01(T, dev)),
              _hip.ptr(noise), _hip.host_f32(aabb_host), float(bound), N, T, _hip.ptr(table_f16), grid_spec.L, grid_spec.F, grid_spec.h_scales,
              grid_spec.h_res, grid_spec.h_offsets, _hip.ptr(sigma_w16), 1 if lidar else 0, _hip.ptr(head_a_w16), _hip.ptr(head_b_w16),
              float(k_scale), float(w_thresh), _hip.host_f32(bg_host) if bg_host is not None else None, _hip.ptr(planes), _hip.ptr(z_vals),
              _hip.ptr(weights), _hip.ptr(ws), _hip.ptr(depth), _hip.ptr(image), _hip.ptr(x01), _hip.ptr(feat), _hip.ptr(geo16), _hip.ptr(sigma),
              _hip.ptr(rgbs))
    return z_vals, weights, ws, depth, image, x01, feat, geo16, sigma, rgbs


class RenderRaysFn(Function):
    """The TRAINING forward of a whole uniform render of a static hash field from the rays, as ONE autograd node:
    (rays, tables, MLP weights) -> z_vals [N, T], weights [N, T], weights_sum [N], depth [N], image [N, C]
    (renderer_dynamic.py:155-237 with network_dynamic.py:213-330 inside).

    Forward = the evaluation render's kernel(s) in their TRAIN form (nvsf_render_uniform_train_fwd: one launch, a wave per ray;
    level-sliced encode pass + streaming tail for camera batches), which also keep what the backward reads: positions, feature
    rows, sigma, geometry rows, masked per-sample colours -- 128 B per sample.  The operator chain it replaces (DensityRaysFn ->
    CompositeWeightsFn -> mask / count -> HeadsFn x 1-2 -> MaskedSigmoidFn -> CompositeImageFn) runs the same arithmetic as seven
    to nine launches that write and re-read the [M, 16] network outputs, the per-sample logits and the mask: 1.0 against 0.4 ms of
    device time for a LiDAR batch of 4096 x 768.  Backward = the chain's backward kernels in the chain's order: image -> sigmoid ->
    heads (fused data + weight gradients on shared-prefix rows) -> compositor -> density MLP -> table scatter (_density_backward:
    side stream, gradient sink, level-major hand-over -- unchanged)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, nears, fars, T, aabb_host, bound, noise, table_params, table_f16, grid_spec, sigma_params, sigma_w16,
                sigma_spec, head_a_params, head_a_w16, head_b_params, head_b_w16, head_spec, enc_ray, n_enc, lidar, k_scale, bg_host, w_thresh,
                sigma_lo, sigma_hi, sliced, train_ctx):
        N, C = rays_o.shape[0], (2 if lidar else 3)
        z_vals, weights, ws, depth, image, x01, feat, geo16, sigma, rgbs = render_uniform_train_forward(
            rays_o, rays_d, nears, fars, T, aabb_host, bound, noise, table_f16, grid_spec, sigma_w16, lidar, head_a_w16, head_b_w16, k_scale, bg_host,
            w_thresh, sliced)
        ctx.save_for_backward(x01, feat, sigma, sigma_w16, geo16, rgbs, weights, z_vals, nears, fars, head_a_w16, head_b_w16, enc_ray)
        ctx.dims = (N, T, C, int(n_enc), bool(lidar), float(k_scale), None if bg_host is None or lidar else tuple(float(v) for v in bg_host))
        ctx.grid_spec, ctx.mlp_spec, ctx.head_spec, ctx.clamp = grid_spec, sigma_spec, head_spec, (float(sigma_lo), float(sigma_hi))
        ctx.table_param, ctx.rows_per_ray, ctx.train_ctx = table_params, T, train_ctx
        ctx.need_table, ctx.need_w = ctx.needs_input_grad[8], ctx.needs_input_grad[11]
        ctx.need_heads = (ctx.needs_input_grad[14], ctx.needs_input_grad[16])
        if train_ctx is not None and table_params is not None and table_params.requires_grad and torch.is_grad_enabled():
            train_ctx.expect(table_params)
        ctx.mark_non_differentiable(z_vals)
        return z_vals, weights, ws, depth, image

    @staticmethod
    def backward(ctx, _g_z, g_weights, g_ws, g_depth, g_image):
        import types
        x01, feat, sigma, sigma_w16, geo16, rgbs, weights, z_vals, nears, fars, w16_a, w16_b, enc_ray = ctx.saved_tensors
        N, T, C, n_enc, lidar, k_scale, bg = ctx.dims
        M, dev = N * T, x01.device
        f32 = dict(dtype=torch.float32, device=dev)
        c = lambda t: None if t is None else t.float().contiguous()
        g_weights, g_ws, g_depth, g_image = c(g_weights), c(g_ws), c(g_depth), c(g_image)
        # ---- image = sum_i w rgb (+ (1 - ws) bg)
        g_w = g_ws_img = None
        g_rgb = torch.zeros(M, C, **f32) if g_image is None else torch.empty(M, C, **f32)
        if g_image is not None:
            g_w = torch.empty(N, T, **f32)
            g_ws_img = torch.empty(N, **f32) if bg is not None else None
            _hip.call("nvsf_composite_uniform_image_bwd", _hip.ptr(weights), _hip.ptr(rgbs), _hip.ptr(g_image), N, T, C,
                      _hip.ptr(device_constant(bg, dev)) if bg is not None else None, _hip.ptr(g_w), _hip.ptr(g_rgb), _hip.ptr(g_ws_img))
        # ---- rgb = [w > thresh] sigmoid(logits): g rgb (1 - rgb) vanishes on the masked samples by itself
        g_logits = torch.empty(M, C, **f32)
        _hip.call("nvsf_sigmoid_bwd", _hip.ptr(g_rgb), _hip.ptr(rgbs), M * C, _hip.ptr(g_logits))
        # ---- heads on shared-prefix rows (the encoded direction once per ray): data + weight gradients in one launch per head
        hs = ctx.head_spec
        n_geo = hs.n_in - n_enc
        grad_geo = torch.empty(M, (n_geo + 3) // 4 * 4, **f32)[:, :n_geo]
        prefix = (enc_ray, T, n_enc)
        _, gw_a = mlp_backward(geo16, w16_a, hs, g_logits[:, :hs.n_out], need_grad_x=True, grad_x=grad_geo, gx_col0=n_enc, prefix=prefix)
        gw_b = grad_geo_b = None
        if lidar:
            # the second head's geometry gradient goes to rows of its own and the density MLP's backward sums the two while it fetches
            # its operands (nvsf_mlp_bwd_density): adding into the first head's rows is a read-modify-write whose load the head
            # kernel -- one wave per SIMD -- cannot hide (0.45 against 0.35 ms alone)
            composed = _testing.get("density_grad") == "composed" and n_geo == ctx.mlp_spec.n_out - 1
            if composed:
                grad_geo_b = torch.empty(M, (n_geo + 3) // 4 * 4, **f32)[:, :n_geo]
            _, gw_b = mlp_backward(geo16, w16_b, hs, g_logits[:, hs.n_out:2 * hs.n_out], need_grad_x=True,
                                   grad_x=grad_geo_b if composed else grad_geo, gx_col0=n_enc, accumulate=not composed, prefix=prefix)
        # ---- weights / weights_sum / depth from sigma
        if g_w is not None and g_weights is not None:
            g_w = g_w + g_weights
        elif g_w is None:
            g_w = g_weights
        if g_ws_img is not None:
            g_ws = g_ws_img if g_ws is None else g_ws + g_ws_img
        g_sigma = torch.empty(N, T, **f32)
        _hip.call("nvsf_composite_uniform_weights_bwd", _hip.ptr(sigma), _hip.ptr(z_vals), _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(g_w), _hip.ptr(g_ws),
                  _hip.ptr(g_depth), N, T, k_scale, _hip.ptr(g_sigma))
        # ---- density MLP + table scatter: DensityFn's backward on what this node saved
        dctx = types.SimpleNamespace(saved_tensors=(x01, feat, sigma, sigma_w16), mlp_spec=ctx.mlp_spec, clamp=ctx.clamp, grid_spec=ctx.grid_spec,
                                     table_param=ctx.table_param, rows_per_ray=ctx.rows_per_ray, train_ctx=ctx.train_ctx, need_table=ctx.need_table,
                                     need_w=ctx.need_w, g_geo_b=grad_geo_b)
        d = _density_backward(dctx, g_sigma.view(-1), grad_geo)  # (grad_table, None, None, grad_w_sigma, ...)
        out = [None] * 29
        out[8], out[11] = d[0], d[3]
        if ctx.need_heads[0]:
            out[14] = gw_a
        if lidar and ctx.need_heads[1]:
            out[16] = gw_b
        return tuple(out)


# ------------------------------------------------------------------------------------------------
# K-planes (planes_field.py)
# ------------------------------------------------------------------------------------------------
PLANE_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))  # itertools.combinations(range(4), 2)


class PlanesFn(Function):
    """(xt [M,4], planes_cl: every plane channel-last in one fp32 buffer = Planes4D's parameter) -> static [M, S*8] and / or dynamic
    [M, S*8] features.  The gradient of `planes_cl` is the buffer the backward kernel scatters into, as it is."""

    @staticmethod
    def forward(ctx, xt, planes_cl, res_host, want, train_ctx=None):
        xt = xt.float().contiguous()
        M, S = xt.shape[0], len(res_host) // 4
        dev = xt.device
        cl = planes_cl.detach()
        if cl.dtype != torch.float32 or not cl.is_contiguous():
            cl = cl.float().contiguous()
        out_s = torch.empty(M, S * 8, dtype=torch.float32, device=dev) if want & 1 else None
        out_d = torch.empty(M, S * 8, dtype=torch.float32, device=dev) if want & 2 else None
        _hip.call("nvsf_planes_fwd", _hip.ptr(xt), M, _hip.ptr(cl), S, 8, _hip.host_u32(res_host), int(want), _hip.ptr(out_s),
                  _hip.ptr(out_d))
        ctx.save_for_backward(xt, cl)
        ctx.res_host, ctx.want = res_host, want
        ctx.train_ctx, ctx.planes_param = train_ctx, planes_cl
        if train_ctx is not None and isinstance(planes_cl, torch.nn.Parameter) and planes_cl.requires_grad and torch.is_grad_enabled():
            train_ctx.expect(planes_cl)
        outs = tuple(o for o in (out_s, out_d) if o is not None)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        xt, planes_cl = ctx.saved_tensors
        want, S = ctx.want, len(ctx.res_host) // 4
        gi = iter(grads)
        g_s = next(gi).float().contiguous() if want & 1 else None
        g_d = next(gi).float().contiguous() if want & 2 else None
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # Inside a training step the texel gradients are ADDED straight into the step's gradient sink (the parameter's .grad or its
        # bucket view), as the table scatters are: a space-time model evaluates its planes three times per ray batch, and autograd
        # would zero-fill a 67 MB buffer per evaluation and add the three up again.
        tctx, param = ctx.train_ctx, ctx.planes_param
        res = _hip.host_u32(ctx.res_host)

        def launch(g_planes, g_xt):  # the coordinate gradients and the texel scatter are separate kernels: either pointer may be NULL
            _hip.call("nvsf_planes_bwd", _hip.ptr(xt), xt.shape[0], _hip.ptr(planes_cl), S, 8, res, int(want), _hip.ptr(g_s), _hip.ptr(g_d),
                      _hip.ptr(g_planes), _hip.ptr(g_xt))
        g_xt = torch.empty_like(xt) if need_x else None
        sink_ok = (need_p and tctx is not None and tctx.sink is not None and isinstance(param, torch.nn.Parameter) and param.is_cuda
                   and param.dtype == torch.float32)
        if sink_ok and tctx.overlap:
            # the coordinate gradients (what the flow field's backward waits for) on this stream, the texel scatter -- the longest
            # kernel of a space-time step -- on the step's side stream beside the rest of backward, as the table scatters
            if need_x:
                launch(None, g_xt)
            tensors = tuple(t for t in (xt, planes_cl, g_s, g_d) if t is not None)
            if scatter_beside_backward(tctx, param, tensors, lambda view, pool: launch(view, None)):
                return g_xt, None, None, None, None
            raise _hip.NvsfHipError("PlanesFn: the step's gradient sink refused the planes parameter")
        if sink_ok:
            last = tctx.done(param)
            view = tctx.sink.view_for(param)
            launch(view, g_xt)
            if last:
                tctx.sink.mark_ready(param)
            return g_xt, None, None, None, None
        if tctx is not None and isinstance(param, torch.nn.Parameter):
            tctx.done(param)
        g_planes = torch.zeros_like(planes_cl) if need_p else None
        if need_p or need_x:
            launch(g_planes, g_xt)
        return g_xt, g_planes, None, None, None


def planes_multi_forward(x, evals, planes_cl, res_host, blend=False, out_f16=False):
    """Several evaluations of ONE position set in one launch (nvsf_planes_multi_fwd): `evals` = [(group, offsets | None, offset_col, time)]
    with group 0 = static / 1 = dynamic planes; returns the list of [M, n_scales * 8] feature matrices (fp32; blend: [static, blended
    dynamic], fp16 rows with out_f16).  No autograd here: Planes4D.forward_multi (no-grad render) and PlanesMultiFn (training) call it."""
    import ctypes
    x = x.float()
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.contiguous()
    M, n = x.shape[0], len(evals)
    S = len(res_host) // 4
    width = S * 8
    if out_f16 and not blend:
        raise ValueError("planes_multi_forward: out_f16 needs blend=True")
    outs = [torch.empty(M, width, dtype=torch.float16 if out_f16 else torch.float32, device=x.device) for _ in (evals[:2] if blend else evals)]
    offs = []
    for _, o, _, _ in evals:
        if o is not None and (o.dtype != torch.float32 or o.dim() != 2 or o.stride(1) != 1):
            o = o.float().contiguous()
        offs.append(o)
    _hip.call("nvsf_planes_multi_fwd", _hip.ptr_rows(x), x.stride(0), M, _hip.ptr(planes_cl), S, 8, _hip.host_u32(res_host), n,
              _hip.host_i32([e[0] for e in evals]), (ctypes.c_void_p * n)(*[None if o is None else o.data_ptr() for o in offs]),
              _hip.host_u32([0 if o is None else o.stride(0) for o in offs]), _hip.host_u32([e[2] for e in evals]),
              _hip.host_f32([e[3] for e in evals]), (ctypes.c_void_p * n)(*([t.data_ptr() for t in outs] + [None] * (n - len(outs)))),
              (2 if out_f16 else 1) if blend else 0)
    return outs


class PlanesMultiFn(Function):
    """What one density query of the space-time field evaluates on its K-planes, as ONE autograd node (network_dynamic.py:220-271):
    (x [M, 3], flow [M, 6] | None, planes_cl) -> (static at x, dynamic at (x, t), dynamic at (x + flow[:, :3], t1), dynamic at
    (x + flow[:, 3:], t2)); a neighbour whose time is None is absent (its output is an empty tensor).
    `blend`: the second output is the neighbour blend 0.5 d + 0.25 (d1 + d2) of network_dynamic.py:273 itself (formed inside the forward
    kernel, an absent neighbour being the base evaluation as in the reference) and the other two outputs are empty: half the feature
    matrices written, ONE gradient read by the three dynamic evaluations (x 0.5, 0.25, 0.25).
    Forward = nvsf_planes_multi_fwd (one launch, bit-identical to one PlanesFn per evaluation); backward = nvsf_planes_multi_bwd: ONE
    texel-scatter launch for all evaluations (on the training step's side stream, into its gradient sink, like PlanesFn's) and ONE
    launch for the gradients of the two flow offsets; gradients that arrive as column slices of a wider matrix (the density tail's input
    gradient) are read in place.  Replaces three PlanesFn nodes per ray batch: three forward, three scatter and two coordinate-gradient
    launches, the [M, 4] position copies of the neighbours and their time columns."""

    @staticmethod
    def forward(ctx, x, flow, planes_cl, res_host, t0, t1, t2, train_ctx, blend=False):
        x = x.float()
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.contiguous()
        cl = planes_cl.detach()
        if cl.dtype != torch.float32 or not cl.is_contiguous():
            cl = cl.float().contiguous()
        fl = None
        if flow is not None:
            fl = flow.detach().float()
            if fl.dim() != 2 or fl.stride(1) != 1:
                fl = fl.contiguous()
        base = (1, None, 0, float(t0))
        evals = [(0, None, 0, float(t0)), base]
        slots = [0, 1]
        for slot, (tn, col) in enumerate(((t1, 0), (t2, 3)), start=2):
            if tn is not None:
                if fl is None:
                    raise ValueError("PlanesMultiFn: a neighbour evaluation needs the flow offsets")
                evals.append((1, fl, col, float(tn)))
                slots.append(slot)
            elif blend:  # the reference's plane_feat_1 = plane_feat_d for a frame without that neighbour
                evals.append(base)
                slots.append(slot)
        outs = planes_multi_forward(x, evals, cl, res_host, blend=bool(blend))
        ctx.save_for_backward(x, cl) if fl is None else ctx.save_for_backward(x, cl, fl)
        ctx.evals_meta = [(e[0], e[2], e[3], e[1] is not None) for e in evals]
        ctx.slots, ctx.res_host, ctx.blend = slots, res_host, bool(blend)
        ctx.train_ctx, ctx.planes_param = train_ctx, planes_cl
        if train_ctx is not None and isinstance(planes_cl, torch.nn.Parameter) and planes_cl.requires_grad and torch.is_grad_enabled():
            train_ctx.expect(planes_cl)
        full = [x.new_zeros(0)] * 4
        for slot, o in zip(slots, outs):  # (blend: two outputs)
            full[slot] = o
        return tuple(full)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        saved = ctx.saved_tensors
        x, cl = saved[0], saved[1]
        fl = saved[2] if len(saved) > 2 else None
        M, S = x.shape[0], len(ctx.res_host) // 4
        need_flow, need_p = ctx.needs_input_grad[1] and fl is not None, ctx.needs_input_grad[2]
        n = len(ctx.evals_meta)

        def rows(t):  # fp32 rows with unit column stride are read where they are (a slice of the density tail's input gradient)
            if t is None:
                return None
            if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
                t = t.float().contiguous()
            return t
        if ctx.blend:
            g_s, g_b = rows(grads[0]), rows(grads[1])
            g = [g_s] + [g_b] * (n - 1)
            scales = [1.0, 0.5] + [0.25] * (n - 2)
        else:
            g = [rows(grads[slot]) for slot in ctx.slots]
            scales = [1.0] * n
        res = _hip.host_u32(ctx.res_host)
        groups = _hip.host_i32([m[0] for m in ctx.evals_meta])
        has_off = [m[3] for m in ctx.evals_meta]
        offs = (ctypes.c_void_p * n)(*[fl.data_ptr() if has_off[i] else None for i in range(n)])
        off_stride = _hip.host_u32([fl.stride(0) if has_off[i] else 0 for i in range(n)])
        off_col = _hip.host_u32([m[1] for m in ctx.evals_meta])
        times = _hip.host_f32([m[2] for m in ctx.evals_meta])
        g_ptrs = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in g])
        g_strides = _hip.host_u32([S * 8 if t is None else t.stride(0) for t in g])
        g_scales = _hip.host_f32(scales)

        def launch(g_planes, g_flow):
            if g_flow is not None:
                go = (ctypes.c_void_p * n)(*[g_flow.data_ptr() if has_off[i] else None for i in range(n)])
                gs = _hip.host_u32([g_flow.stride(0) if has_off[i] else 0 for i in range(n)])
                gc = _hip.host_u32([ctx.evals_meta[i][1] if has_off[i] else 0 for i in range(n)])
            else:
                go = gs = gc = None
            _hip.call("nvsf_planes_multi_bwd", _hip.ptr_rows(x), x.stride(0), M, _hip.ptr(cl), S, 8, res, n, groups, offs, off_stride, off_col, times,
                      g_ptrs, g_strides, g_scales, _hip.ptr(g_planes), go, gs, gc)
        g_flow = None
        if need_flow:
            # columns of an absent neighbour, or of one whose features received no gradient, stay zero
            complete = sum(has_off) == 2 and all(g[i] is not None for i in range(n) if has_off[i])
            g_flow = (torch.empty if complete else torch.zeros)(M, 6, dtype=torch.float32, device=x.device)
        tctx, param = ctx.train_ctx, ctx.planes_param
        sink_ok = (need_p and tctx is not None and tctx.sink is not None and isinstance(param, torch.nn.Parameter) and param.is_cuda
                   and param.dtype == torch.float32)
        none = (None,) * 6
        if sink_ok and tctx.overlap:
            if g_flow is not None:
                launch(None, g_flow)  # what the flow field's backward waits for: on this stream
            tensors = tuple(t for t in (x, cl, fl, *g) if t is not None)
            if scatter_beside_backward(tctx, param, tensors, lambda view, pool: launch(view, None)):
                return (None, g_flow, None) + none
            raise _hip.NvsfHipError("PlanesMultiFn: the step's gradient sink refused the planes parameter")
        if sink_ok:
            last = tctx.done(param)
            launch(tctx.sink.view_for(param), g_flow)
            if last:
                tctx.sink.mark_ready(param)
            return (None, g_flow, None) + none
        if tctx is not None and isinstance(param, torch.nn.Parameter):
            tctx.done(param)
        g_planes = torch.zeros_like(cl) if need_p else None
        if g_planes is not None or g_flow is not None:
            launch(g_planes, g_flow)
        return (None, g_flow, g_planes) + none


# ------------------------------------------------------------------------------------------------
# uniform sampler / compositor (renderer_dynamic.py:155-237)
# ------------------------------------------------------------------------------------------------
_LIN_CACHE = {}
_CONST_CACHE = {}


def device_constant(values, device):
    """Small fp32 constant (background colour, Lagrange weights ...) as a device tensor, uploaded ONCE per distinct value:
    `torch.tensor(list, device=...)` is a synchronous copy from pageable memory -- it blocks the host until the stream has
    drained, so one such call per render keeps the host from ever running ahead of the device."""
    key = (tuple(float(v) for v in values), str(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 4096:
            _CONST_CACHE.clear()
        t = _CONST_CACHE[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
    return t


def linspace01(T, device):
    """torch.linspace(0, 1, T) on `device`, cached (renderer_dynamic.py:155)."""
    key = (int(T), str(device))
    if key not in _LIN_CACHE:
        _LIN_CACHE[key] = torch.linspace(0.0, 1.0, int(T), device=device)
    return _LIN_CACHE[key]


def uniform_samples(rays_o, rays_d, nears, fars, T, aabb, noise=None, want_xyz=True):
    """z_vals [N,T] (+ clipped xyzs [N,T,3])."""
    N = rays_o.shape[0]
    dev = rays_o.device
    z_vals = torch.empty(N, T, dtype=torch.float32, device=dev)
    xyzs = torch.empty(N, T, 3, dtype=torch.float32, device=dev) if want_xyz else None
    _hip.call("nvsf_uniform_samples", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(linspace01(T, dev)),
              _hip.ptr(noise), _hip.ptr(aabb), N, T, _hip.ptr(z_vals), _hip.ptr(xyzs))
    return z_vals, xyzs


class CompositeWeightsFn(Function):
    """(sigmas [N,T], z_vals, nears, fars) -> weights [N,T], weights_sum [N], depth [N]."""

    @staticmethod
    def forward(ctx, sigmas, z_vals, nears, fars, k_scale):
        sigmas = sigmas.float().contiguous()
        N, T = sigmas.shape
        weights = torch.empty_like(sigmas)
        weights_sum = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        depth = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        _hip.call("nvsf_composite_uniform_weights_fwd", _hip.ptr(sigmas), _hip.ptr(z_vals), _hip.ptr(nears), _hip.ptr(fars), N, T,
                  float(k_scale), _hip.ptr(weights), _hip.ptr(weights_sum), _hip.ptr(depth))
        ctx.save_for_backward(sigmas, z_vals, nears, fars)
        ctx.k_scale = float(k_scale)
        return weights, weights_sum, depth

    @staticmethod
    def backward(ctx, g_weights, g_ws, g_depth):
        sigmas, z_vals, nears, fars = ctx.saved_tensors
        N, T = sigmas.shape
        grad = torch.empty_like(sigmas)
        gw = g_weights.float().contiguous() if g_weights is not None else None
        gs = g_ws.float().contiguous() if g_ws is not None else None
        gd = g_depth.float().contiguous() if g_depth is not None else None
        _hip.call("nvsf_composite_uniform_weights_bwd", _hip.ptr(sigmas), _hip.ptr(z_vals), _hip.ptr(nears), _hip.ptr(fars),
                  _hip.ptr(gw), _hip.ptr(gs), _hip.ptr(gd), N, T, ctx.k_scale, _hip.ptr(grad))
        return grad, None, None, None, None


class CompositeImageFn(Function):
    """image[n] = sum_i w[n,i] rgb[n,i,:] (+ (1 - ws[n]) * bg)."""

    @staticmethod
    def forward(ctx, weights, rgbs, weights_sum, bg):
        weights, rgbs = weights.contiguous(), rgbs.float().contiguous()
        N, T, C = rgbs.shape
        image = torch.empty(N, C, dtype=torch.float32, device=rgbs.device)
        _hip.call("nvsf_composite_uniform_image_fwd", _hip.ptr(weights), _hip.ptr(rgbs), _hip.ptr(weights_sum), N, T, C, _hip.ptr(bg),
                  _hip.ptr(image))
        ctx.save_for_backward(weights, rgbs, bg if bg is not None else torch.empty(0, device=rgbs.device))
        ctx.has_bg = bg is not None
        return image

    @staticmethod
    def backward(ctx, g_image):
        weights, rgbs, bg = ctx.saved_tensors
        N, T, C = rgbs.shape
        g_image = g_image.float().contiguous()
        g_w = torch.empty_like(weights)
        g_rgb = torch.empty_like(rgbs)
        g_ws = torch.empty(N, dtype=torch.float32, device=rgbs.device) if ctx.has_bg else None
        _hip.call("nvsf_composite_uniform_image_bwd", _hip.ptr(weights), _hip.ptr(rgbs), _hip.ptr(g_image), N, T, C,
                  _hip.ptr(bg) if ctx.has_bg else None, _hip.ptr(g_w), _hip.ptr(g_rgb), _hip.ptr(g_ws))
        return g_w, g_rgb, g_ws, None


# ------------------------------------------------------------------------------------------------
# fused uniform-render kernels (static hash field)
# ------------------------------------------------------------------------------------------------
L2_BYTES_PER_XCD = 4 << 20


def prefer_sliced(spec, N, T, ray_length, bound, coherent=False):
    """Host-side choice between the fused density kernel and its level-sliced formulation (same results).

    The sliced form wins when the table is far larger than one XCD's L2 AND consecutive samples of a ray are more
    than about one finest-level cell apart (no reuse of fine-level cache lines along the ray): measured on config 2,
    camera rays (2.7 cells) 0.53 -> 0.46 ms, LiDAR rays (0.6 cells) 0.24 -> 0.35 ms (whole render with the two-lanes-per-sample
    encode pass: LiDAR 0.27 + 0.20 ms tail against 0.36 ms for the one-launch gather form).  `ray_length` is the caller's
    host-side estimate of far - near.  `coherent`: the batch is a run of consecutive pixels of a frame (staged evaluation) --
    neighbouring rays then ask for neighbouring cells and the one-launch gather form finds them in L1 / L2 (measured on whole
    frames: 51.5 against 62.8 ms per frame, 19.4 against 15.9 frames/s).  Tests force either form (testing.variant(density_sliced=...))."""
    if sliced_only(spec):  # L8 F4: the fused kernels exist in the level-sliced form only (one level per XCD)
        return N * T < 2 ** 28
    if not (spec.L == 16 and spec.F == 2 and spec.D == 3 and N * T < 2 ** 32):
        return False
    forced = _testing.get("density_sliced")
    if forced is not None:
        return bool(forced)
    if coherent:
        return False
    table_bytes = spec.n_params * 2
    cells_per_step = (ray_length / T) * max(spec.res) / (2.0 * bound)
    return table_bytes >= 2 * L2_BYTES_PER_XCD and cells_per_step >= 1.0 and N * T >= (1 << 18)


def density_uniform(rays_o, rays_d, nears, fars, T, aabb_host, bound, table_f16, spec, sigma_weights_f16, noise=None, sliced=False,
                    _passes=3, _buffers=None):
    """ray batch -> z_vals [N,T] fp32, sigmas [N,T] fp32, geo [N,T,16] fp16 (h1..h15, 1.0).
    sliced=True: the two-launch formulation with the levels partitioned over the XCDs (same results bit for bit)."""
    N = rays_o.shape[0]
    dev = rays_o.device
    if _buffers is not None:  # (z_vals, sigmas, geo, feat) of an earlier call: per-pass timing in bench.py
        z_vals, sigmas, geo, feat = _buffers
    else:
        z_vals = torch.empty(N, T, dtype=torch.float32, device=dev)
        sigmas = torch.empty(N, T, dtype=torch.float32, device=dev)
        geo = torch.empty(N, T, 16, dtype=torch.float16, device=dev)
        feat = torch.empty(16, N * T, dtype=torch.int32, device=dev) if sliced else None  # 8 planes of 8 bytes per sample
    args = (_hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars),
            _hip.ptr(linspace01(T, dev)), _hip.ptr(noise), _hip.host_f32(aabb_host), float(bound), N, T, _hip.ptr(table_f16),
            spec.L, spec.F, spec.h_scales, spec.h_res, spec.h_offsets, _hip.ptr(sigma_weights_f16), _hip.ptr(z_vals),
            _hip.ptr(sigmas), _hip.ptr(geo))
    if sliced:
        _hip.call("nvsf_field_density_uniform_sliced_fwd", *args, _hip.ptr(feat), int(_passes))
    else:
        _hip.call("nvsf_field_density_uniform_fwd", *args)
    if _buffers is not None:
        return z_vals, sigmas, geo, feat
    return z_vals, sigmas, geo


def heads_uniform(weights, geo, rays_d, weights_sum, lidar, head_a_f16, head_b_f16, bg_host=None, w_thresh=W_THRESH):
    """(weights, geo, dirs) -> image [N, 2 (LiDAR: raydrop, intensity) | 3 (camera rgb)]."""
    N, T = weights.shape
    C = 2 if lidar else 3
    image = torch.empty(N, C, dtype=torch.float32, device=weights.device)
    _hip.call("nvsf_field_heads_uniform_fwd", _hip.ptr(weights), _hip.ptr(geo), _hip.ptr(rays_d), _hip.ptr(weights_sum),
              1 if lidar else 0, _hip.ptr(head_a_f16), _hip.ptr(head_b_f16), N, T, float(w_thresh),
              _hip.host_f32(bg_host) if bg_host is not None else None, _hip.ptr(image))
    return image


def render_uniform(rays_o, rays_d, nears, fars, T, aabb_host, bound, table_f16, spec, sigma_weights_f16, lidar, head_a_f16, head_b_f16,
                   k_scale, bg_host=None, noise=None, sliced=False, w_thresh=W_THRESH, _stage=None, _buffers=None):
    """Whole uniform render of a ray batch (evaluation): -> z_vals, weights [N,T], weights_sum, depth [N], image [N, 2 | 3].
    One launch (a wave per ray); with sliced=True the encode pass of the level-sliced path runs first and the render
    kernel reads its feature planes instead of gathering.  `_stage` ("encode" | "tail") with `_buffers` = the tuple
    returned by an earlier call lets bench.py time the two launches of the sliced form separately."""
    N = rays_o.shape[0]
    dev = rays_o.device
    lin = linspace01(T, dev)
    if _buffers is not None:
        z_vals, weights, ws, depth, image, feat = _buffers
    else:
        z_vals = torch.empty(N, T, dtype=torch.float32, device=dev)
        weights = torch.empty(N, T, dtype=torch.float32, device=dev)
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 2 if lidar else 3, dtype=torch.float32, device=dev)
        feat = torch.empty(16, N * T, dtype=torch.int32, device=dev) if sliced else None  # 8 planes of 8 bytes per sample
    if sliced and _stage != "tail":
        _hip.call("nvsf_field_density_uniform_sliced_fwd", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(lin),
                  _hip.ptr(noise), _hip.host_f32(aabb_host), float(bound), N, T, _hip.ptr(table_f16), spec.L, spec.F, spec.h_scales,
                  spec.h_res, spec.h_offsets, _hip.ptr(sigma_weights_f16), _hip.ptr(z_vals), _hip.ptr(weights), _hip.ptr(weights),
                  _hip.ptr(feat), 1)  # passes = 1: encode only (the sigma / geo arguments are not touched)
    if _stage == "encode":
        return z_vals, weights, ws, depth, image, feat
    _hip.call("nvsf_render_uniform_fwd", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(lin), _hip.ptr(noise),
              _hip.host_f32(aabb_host), float(bound), N, T, _hip.ptr(table_f16), spec.L, spec.F, spec.h_scales, spec.h_res, spec.h_offsets,
              _hip.ptr(sigma_weights_f16), 1 if lidar else 0, _hip.ptr(head_a_f16), _hip.ptr(head_b_f16), float(k_scale), float(w_thresh),
              _hip.host_f32(bg_host) if bg_host is not None else None, _hip.ptr(feat), _hip.ptr(z_vals), _hip.ptr(weights), _hip.ptr(ws),
              _hip.ptr(depth), _hip.ptr(image))
    if _stage is not None:
        return z_vals, weights, ws, depth, image, feat
    return z_vals, weights, ws, depth, image


def sliced_only(spec):
    """The reference-default grid shape (8 levels x 4 features, main_nvsf.py:45-52): its encode pass (k_encode_sliced_f4) writes the
    same feature planes as the L16 F2 pass, so the streaming render / density tails serve both; there is no one-launch gather form."""
    return spec.D == 3 and spec.F == 4 and spec.L == 8


def render_uniform_eligible(spec, n_samples=0):
    """Grid shapes the fused render / training-forward kernels are built for (32 encoded features either way)."""
    return spec.D == 3 and ((spec.F == 2 and spec.L == 16) or (sliced_only(spec) and n_samples < 2 ** 28))


def occupancy_fused_eligible(spec):
    """Static hash fields nvsf_render_occupancy_fwd is built for: 32 encoded features as 16 levels x 2 (BASELINE config 2) or
    8 levels x 4 (the reference-default grid shape)."""
    return spec.D == 3 and spec.F in (2, 4) and spec.L * spec.F == 32


def render_occupancy(rays_o, rays_d, nears, fars, bitfield, bound, dt_gamma, max_steps, C, H, table_f16, spec, sigma_weights_f16, lidar,
                     head_a_f16, head_b_f16, density_scale, T_thresh, bg_host=None):
    """One-launch evaluation-mode occupancy render -> weights_sum [N], depth [N], image [N, 2 | 3]."""
    N = rays_o.shape[0]
    dev = rays_o.device
    ws = torch.empty(N, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    image = torch.empty(N, 2 if lidar else 3, dtype=torch.float32, device=dev)
    _hip.call("nvsf_render_occupancy_fwd", _hip.ptr(rays_o), _hip.ptr(rays_d), _hip.ptr(nears), _hip.ptr(fars), _hip.ptr(bitfield),
              float(bound), float(dt_gamma), int(max_steps), int(C), int(H), N, _hip.ptr(table_f16), spec.L, spec.F, spec.h_scales,
              spec.h_res, spec.h_offsets, _hip.ptr(sigma_weights_f16), 1 if lidar else 0, _hip.ptr(head_a_f16), _hip.ptr(head_b_f16),
              float(density_scale), float(T_thresh), _hip.host_f32(bg_host) if bg_host is not None else None, _hip.ptr(ws),
              _hip.ptr(depth), _hip.ptr(image))
    return ws, depth, image
