"""Space-time hash encoders (`HashGridT`, `HashGrid4D`) for MI355X.

Module tree, constructor arguments, `n_output_dims` and results follow
/root/reference/nvsf/nerf/models/hash_field.py:29-173 (state_dict keys `hash_static.params`,
`hash_dynamic.<plane>.hash_t.<slice>.params`): one 3-D multiresolution grid for the static scene plus, for each
coordinate pair (xy, xz, yz), `time_resolution` 2-D grids that are blended linearly between the two time slices
around t and then reduced over groups of 4 features with cubic Lagrange weights at nodes 0, 1/3, 2/3, 1.
Every grid evaluation is the HIP hash-grid kernel (csrc/hashgrid.hip) reading the coordinate columns of x in
place (no x[:, [0, 2]] copies).

Arithmetic order and dtypes are kept exactly as in the reference (its results depend on them): with `t` a
dimensioned tensor the blend and the Lagrange reduction run in fp32; with a 0-dim `t` (the flow-warped
neighbour frames, network_dynamic.py:244,260) they run in fp16 -- PyTorch type promotion does the same here
because the same expressions are applied to the same operand types.  The reference indexes its ModuleList with
device tensors and branches on `idx1 == idx2` (one device->host sync per plane per call); here the slice indices
come from one host copy of t per call.
"""
import math

import numpy as np
import torch
from torch import nn

import tinycudann as tcnn
from nvsf import testing
from nvsf import field_ops as _ops


_HOST_TIME_CACHE = {}  # id(tensor) -> (weakref, version, value): the last few time tensors read back


def _host_time(t):
    """The value of the frame time on the host.  A device tensor costs a device->host read that drains the stream; a staged
    render (renderer_dynamic.py:286-316: a frame in ~130 ray chunks) or a benchmark loop passes the SAME tensor object again
    and again, so the value is remembered per tensor object for as long as that object is alive and unmodified (same
    `_version`).  Another tensor -- even at the same address -- is read again."""
    if not torch.is_tensor(t):
        return float(t)
    if not t.is_cuda:
        return float(t.detach().reshape(-1)[0])
    import weakref
    hit = _HOST_TIME_CACHE.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return hit[2]
    value = float(t.detach().reshape(-1)[0])
    if len(_HOST_TIME_CACHE) >= 16:
        for k in [k for k, v in _HOST_TIME_CACHE.items() if v[0]() is None] or list(_HOST_TIME_CACHE)[:8]:
            _HOST_TIME_CACHE.pop(k, None)
    _HOST_TIME_CACHE[id(t)] = (weakref.ref(t), t._version, value)
    return value


def lagrange_weights_host(t_host, num_basis, reciprocal_division):
    """The values `lagrange_weights` produces, evaluated on the host in fp32.  PyTorch divides a device tensor by a
    Python scalar as a multiplication by the fp32 reciprocal and a CPU tensor by a true division; both are offered."""
    t32 = np.float32(t_host)
    nodes = [i / (num_basis - 1) for i in range(num_basis)]
    out = []
    for j in range(num_basis):
        w = None
        for m in range(num_basis):
            if m == j:
                continue
            num = np.float32(t32 - np.float32(nodes[m]))
            den = np.float32(nodes[j] - nodes[m])
            f = np.float32(num * (np.float32(1.0) / den)) if reciprocal_division else np.float32(num / den)
            w = f if w is None else np.float32(w * f)
        out.append(float(w))
    return out


def lagrange_weights(t, num_basis):
    """w_j(t) = prod_{m != j} (t - T_m) / (T_j - T_m), T_m = m / (num_basis - 1); factors multiplied in order of m."""
    nodes = [i / (num_basis - 1) for i in range(num_basis)]
    weights = []
    for j in range(num_basis):
        w = None
        for m in range(num_basis):
            if m == j:
                continue
            f = (t - nodes[m]) / (nodes[j] - nodes[m])
            w = f if w is None else w * f
        weights.append(w)
    return weights


def lagrange_reduce(feat, t, n_levels, n_features, num_basis):
    """[N, L*F] -> [N, L*F/num_basis]: the F features of every level are split into `num_basis` chunks which are
    combined with the Lagrange weights (hash_field.py:65-74, flow_field.py:105-114)."""
    chunks = torch.chunk(feat.view(-1, n_levels, n_features), num_basis, dim=-1)
    w = lagrange_weights(t, num_basis)
    out = w[0] * chunks[0]
    for i in range(1, num_basis):
        out = out + w[i] * chunks[i]
    return out.view(-1, n_levels * n_features // num_basis)


class HashGridT(nn.Module):
    def __init__(self, time_resolution=25, base_resolution=512, max_resolution=32768, n_levels=8, n_features_per_level=4,
                 log2_hashmap_size=14, num_basis=4, cols=(0, 1)):
        super().__init__()
        self.time_resolution = time_resolution
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        cfg = {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features_per_level,
               "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution, "per_level_scale": per_level_scale}
        self.hash_t = nn.ModuleList([tcnn.Encoding(n_input_dims=2, encoding_config=cfg) for _ in range(time_resolution)])
        self.n_levels, self.n_features_per_level, self.num_basis = n_levels, n_features_per_level, num_basis
        self.n_output_dims = n_levels * n_features_per_level // num_basis
        self.cols = tuple(cols)

    def _slice(self, k, x):
        return self.hash_t[k].encode_columns(x, self.cols)

    def forward(self, x, t, t_host=None):
        """x: [N, >=2] (the two columns `self.cols` are encoded), t in [0, 1]."""
        t_host = _host_time(t) if t_host is None else t_host
        idx_host = np.float32(t_host) * np.float32(self.time_resolution - 1)
        k1, k2 = int(math.floor(idx_host)), int(math.ceil(idx_host))
        if k1 == k2:
            feat = self._slice(k1, x)
        else:
            idx = t * (self.time_resolution - 1)
            feat = (k2 - idx) * self._slice(k1, x) + (idx - k1) * self._slice(k2, x)
        return lagrange_reduce(feat, t, self.n_levels, self.n_features_per_level, self.num_basis)


class HashGrid4D(nn.Module):
    def __init__(self, base_resolution=512, max_resolution=32768, time_resolution=8, n_levels=8, n_features_per_level=4,
                 log2_hashmap_size=19, hash_size_dynamic=[15, 13, 13], decompose=True, reduction="concat"):
        super().__init__()
        if reduction not in ("concat", "prod", "sum", "mean"):
            raise ValueError("Invalid reduction")  # hash_field.py:26-27
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        self.hash_static = tcnn.Encoding(n_input_dims=3, encoding_config={
            "otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution, "per_level_scale": per_level_scale})
        pairs = ((0, 1), (0, 2), (1, 2))  # xyt, xzt, yzt
        self.hash_dynamic = nn.ModuleList([
            HashGridT(time_resolution=time_resolution, base_resolution=base_resolution, max_resolution=max_resolution, n_levels=n_levels,
                      n_features_per_level=n_features_per_level, log2_hashmap_size=hash_size_dynamic[i], cols=pairs[i])
            for i in range(3)])
        self.decompose, self.reduction = decompose, reduction
        self.n_output_dims = self.hash_static.n_output_dims + self.hash_dynamic[0].n_output_dims * (3 if reduction == "concat" else 1)

    def _reduce(self, feat):
        """The reference's `reduction_func` (hash_field.py:16-27) on the three pair features, which the kernels deliver side by side
        as [N, 3 * n]: 'concat' is that buffer; 'sum' / 'mean' / 'prod' fold the three chunks in the reference's order and dtype
        (Python's sum starts from 0 + xy, math.prod from 1 * xy: both exact)."""
        if self.reduction == "concat":
            return feat
        n = self.hash_dynamic[0].n_output_dims
        xy, xz, yz = feat[:, :n], feat[:, n:2 * n], feat[:, 2 * n:3 * n]
        if self.reduction == "prod":
            return (xy * xz) * yz
        total = (xy + xz) + yz
        return total / 3 if self.reduction == "mean" else total

    def forward_static(self, x, level_major=False):
        """Static features [N, 32]; level_major=True (no autograd): fp16 [8, N, 4] where the encoder offers it -- the layout its
        one-level-per-XCD kernel writes as whole lines and the fused density tail reads (network_dynamic._density_tail_fused)."""
        if level_major:
            if torch.is_grad_enabled():
                return self.hash_static(x, level_major=True)  # rows where the level-major form is not built
            out = self.hash_static.forward_level_major(x)
            if out is not None:
                return out
        return self.hash_static(x)

    def forward_dynamic(self, x, t, t_host=None, offset=None, offset_col=0):
        """Dynamic features [N, 24] at positions x (+ offset[:, offset_col:offset_col+3]).  Without autograd the three
        planes run as ONE fused kernel (csrc/hashgrid4d.hip); with autograd recording, the per-slice operator path."""
        t_host = _host_time(t) if t_host is None else t_host
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if offset is not None:
                x = x + offset[:, offset_col:offset_col + 3]
            first = self.hash_dynamic[0]
            fp32_regime = torch.is_tensor(t) and t.dim() > 0
            if (fp32_regime and not x.requires_grad and first.n_levels == 8 and first.n_features_per_level == 4 and first.num_basis == 4
                    and testing.get("hash4d_train") == "fused"):
                # fused forward + fused table-gradient kernel (csrc/hashgrid4d.hip) instead of six per-slice encoder calls,
                # their blends / Lagrange reductions and six backward launches
                idx = np.float32(t_host) * np.float32(first.time_resolution - 1)
                k1, k2 = int(math.floor(idx)), int(math.ceil(idx))
                params = [pl.hash_t[k1].params for pl in self.hash_dynamic] + [pl.hash_t[k2].params for pl in self.hash_dynamic]
                return self._reduce(HashDynFn.apply(self, x, t, t_host, k1, k2, *params))
            return self._reduce(torch.cat([plane(x, t, t_host) for plane in self.hash_dynamic], dim=-1))
        return self._reduce(self._forward_dynamic_fused(x, t, t_host, offset, offset_col))

    def _forward_dynamic_fused(self, x, t, t_host, offset, offset_col):
        from nvsf import _hip
        import ctypes
        first = self.hash_dynamic[0]
        R, L = first.time_resolution, first.n_levels
        if L != 8 or first.n_features_per_level != 4 or first.num_basis != 4:
            raise NotImplementedError("fused HashGridT kernel: 8 levels x 4 features, 4 Lagrange nodes")
        idx = np.float32(t_host) * np.float32(R - 1)
        k1, k2 = int(math.floor(idx)), int(math.ceil(idx))
        fp16_regime = torch.is_tensor(t) and t.dim() == 0  # PyTorch promotion: a 0-dim t keeps the arithmetic in fp16
        on_device = torch.is_tensor(t) and t.is_cuda
        lag = lagrange_weights_host(t_host, 4, reciprocal_division=on_device)
        h_time = _hip.host_f32([float(np.float32(k2) - idx), float(idx - np.float32(k1))] + lag)
        tables = [pl.hash_t[k1].table_f16() for pl in self.hash_dynamic] + [pl.hash_t[k2].table_f16() for pl in self.hash_dynamic]
        h_tables = (ctypes.c_void_p * 6)(*[tb.data_ptr() for tb in tables])
        specs = [pl.hash_t[0].spec for pl in self.hash_dynamic]
        h_scales = _hip.host_f32([v for s in specs for v in s.scales])
        h_res = _hip.host_u32([v for s in specs for v in s.res])
        h_off = _hip.host_u32([v for s in specs for v in s.offsets])
        x = x.float().contiguous()
        M = x.shape[0]
        out = torch.empty(M, 24, dtype=torch.float16 if fp16_regime else torch.float32, device=x.device)
        off = None
        if offset is not None:  # rows of a wider buffer (e.g. the padded output of the fused flow MLP) are read in place
            off = offset.float()
            if off.dim() != 2 or off.stride(1) != 1:
                off = off.contiguous()
        _hip.call("nvsf_hashgrid4d_dynamic_fwd", _hip.ptr(x), x.shape[1], None if off is None else _hip.ptr_rows(off), off.stride(0) if off is not None else 0,
                  int(offset_col), M, h_tables, h_scales, h_res, h_off, h_time, 1 if k1 == k2 else 0, 1 if fp16_regime else 0, _hip.ptr(out))
        return out

    def training_fused3(self, x, t, t_host, flow, frame_idx, num_frames):
        """With autograd recording: (hash_d, hash_1, hash_2) of one density query from one forward launch + the fused
        table-gradient kernel (HashDyn3Fn), or None when the fused training path does not apply."""
        first = self.hash_dynamic[0]
        if not (torch.is_grad_enabled() and torch.is_tensor(t) and t.dim() > 0 and not x.requires_grad and first.n_levels == 8
                and first.n_features_per_level == 4 and first.num_basis == 4 and testing.get("hash4d_train") == "fused"):
            return None
        idx = np.float32(t_host) * np.float32(first.time_resolution - 1)
        k1, k2 = int(math.floor(idx)), int(math.ceil(idx))
        nb = []
        for col, frame in ((0, frame_idx + 1), (3, frame_idx - 1)):
            nb.append((torch.tensor(frame / num_frames), float(np.float32(frame / num_frames)), col) if 0 <= frame <= num_frames - 1 else None)
        if nb[0] is None and nb[1] is None:
            return None
        params = [pl.hash_t[k1].params for pl in self.hash_dynamic] + [pl.hash_t[k2].params for pl in self.hash_dynamic]
        return tuple(o if o is None else self._reduce(o) for o in HashDyn3Fn.apply(self, x, t, t_host, k1, k2, flow, nb[0], nb[1], *params))

    def forward_dynamic3(self, x, t, t_host, offsets, neighbours):
        """No-autograd: the dynamic features of one density query in one launch (csrc/hashgrid4d.hip, k_hash_dynamic3):
        at (x, t) [fp32 regime] and, for each entry (t_n 0-dim tensor, its host value, first offset column) | None of
        `neighbours`, at (x + offsets[:, col:col+3], t_n) [fp16 regime].  Returns (hash_d, hash_1 | None, hash_2 | None),
        bit-identical to the separate forward_dynamic calls."""
        from nvsf import _hip
        import ctypes
        first = self.hash_dynamic[0]
        R = first.time_resolution
        if first.n_levels != 8 or first.n_features_per_level != 4 or first.num_basis != 4:
            raise NotImplementedError("fused HashGridT kernel: 8 levels x 4 features, 4 Lagrange nodes")
        assert neighbours[0] is None or neighbours[0][2] == 0
        assert neighbours[1] is None or neighbours[1][2] == 3
        evals = [(t, t_host)] + [None if n is None else (n[0], n[1]) for n in neighbours]
        tables, h_time, flags, base = [], [], [], None
        for e, ev in enumerate(evals):
            if ev is None:
                tables += [0] * 6
                h_time += [0.0] * 6
                flags += [0, 0, 0]
                continue
            te, te_host = ev
            idx = np.float32(te_host) * np.float32(R - 1)
            k1, k2 = int(math.floor(idx)), int(math.ceil(idx))
            on_device = torch.is_tensor(te) and te.is_cuda
            lag = lagrange_weights_host(te_host, 4, reciprocal_division=on_device)
            tables += [pl.hash_t[k1].table_f16().data_ptr() for pl in self.hash_dynamic] + [pl.hash_t[k2].table_f16().data_ptr() for pl in self.hash_dynamic]
            h_time += [float(np.float32(k2) - idx), float(idx - np.float32(k1))] + lag
            if e == 0:
                base = (k1, k2)
            flags += [1, 1 if k1 == k2 else 0, 1 if (k1, k2) == base else 0]
        specs = [pl.hash_t[0].spec for pl in self.hash_dynamic]
        x = x.float().contiguous()
        M, dev = x.shape[0], x.device
        off = offsets.float()
        if off.dim() != 2 or off.stride(1) != 1:
            off = off.contiguous()
        out0 = torch.empty(M, 24, dtype=torch.float32, device=dev)
        out1 = torch.empty(M, 24, dtype=torch.float16, device=dev) if neighbours[0] is not None else None
        out2 = torch.empty(M, 24, dtype=torch.float16, device=dev) if neighbours[1] is not None else None
        _hip.call("nvsf_hashgrid4d_dynamic3_fwd", _hip.ptr(x), x.shape[1], _hip.ptr_rows(off), off.stride(0), M, (ctypes.c_void_p * 18)(*tables),
                  _hip.host_f32([v for s in specs for v in s.scales]), _hip.host_u32([v for s in specs for v in s.res]),
                  _hip.host_u32([v for s in specs for v in s.offsets]), _hip.host_f32(h_time), _hip.host_i32(flags), _hip.ptr(out0),
                  _hip.ptr(out1), _hip.ptr(out2))
        return tuple(o if o is None else self._reduce(o) for o in (out0, out1, out2))

    def forward(self, x, t, t_host=None):
        static, dynamic = self.forward_static(x), self.forward_dynamic(x, t, t_host)
        return [static, dynamic] if self.decompose else torch.cat([static, dynamic], dim=-1)


class HashDynFn(torch.autograd.Function):
    """HashGrid4D.forward_dynamic at the current frame (fp32 regime) with autograd: the fused forward kernel and the fused
    table-gradient kernel.  `params` = the slice parameters (floor slice of the three pairs, then ceil slice) so that the
    gradients reach exactly the tensors the per-slice path would have touched."""

    @staticmethod
    def forward(ctx, enc, x, t, t_host, k1, k2, *params):
        x = x.float().contiguous()
        out = enc._forward_dynamic_fused(x, t, t_host, None, 0)
        ctx.save_for_backward(x)
        ctx.enc, ctx.t, ctx.t_host, ctx.k1, ctx.k2 = enc, t, t_host, k1, k2
        HashDynFn._expect(ctx, enc, params)
        return out

    @staticmethod
    def _expect(ctx, enc, params):
        """The slice parameters this node will scatter into, announced to the training step (ops.TrainContext.expect) so that the
        step's gradient sink knows when each of them has received its last scatter."""
        ctx.params = params
        ctx.train_ctx = _ops.train_context(enc)
        if ctx.train_ctx is not None and torch.is_grad_enabled():
            seen = []
            for p in params:
                if isinstance(p, torch.nn.Parameter) and p.requires_grad and not any(p is q for q in seen):
                    seen.append(p)
                    ctx.train_ctx.expect(p)

    @staticmethod
    def backward(ctx, grad_out):
        return HashDynFn._backward(ctx, grad_out)

    @staticmethod
    def _backward(ctx, grad_out):
        from nvsf import _hip
        import ctypes
        (x,) = ctx.saved_tensors
        enc, t_host, k1, k2 = ctx.enc, ctx.t_host, ctx.k1, ctx.k2
        first = enc.hash_dynamic[0]
        idx = np.float32(t_host) * np.float32(first.time_resolution - 1)
        lag = lagrange_weights_host(t_host, 4, reciprocal_division=torch.is_tensor(ctx.t) and ctx.t.is_cuda)
        h_time = _hip.host_f32([float(np.float32(k2) - idx), float(idx - np.float32(k1))] + lag)
        specs = [pl.hash_t[0].spec for pl in enc.hash_dynamic]
        h_scales = _hip.host_f32([v for s in specs for v in s.scales])
        h_res = _hip.host_u32([v for s in specs for v in s.res])
        h_off = _hip.host_u32([v for s in specs for v in s.offsets])
        same = k1 == k2
        # The two slices of a pair see the same cells and corner weights; their gradients differ by the scalar blend factors
        # only (dL/dtable_lo = blend_lo G, dL/dtable_hi = blend_hi G with G = sum g lag_i w_c).  G is scattered ONCE -- half the
        # atomics, which are what bounds this pass -- and scaled into the two gradients afterwards (tables of ~1 M floats).
        # a column-major gradient ([M, 24] with strides (1, M): DensityTailFn.backward hands it over that way) goes to the kernel as it is
        M = x.shape[0]
        col_major = (grad_out.dtype == torch.float32 and grad_out.dim() == 2 and grad_out.shape[1] == 24 and grad_out.stride() == (1, M)
                     and M >= (1 << 16) and testing.get("hash4d_train") == "fused")
        g_out = grad_out if col_major else grad_out.float().contiguous()
        lag_t = _ops.device_constant(lag, x.device)
        b_lo, b_hi = float(np.float32(k2) - idx), float(idx - np.float32(k1))

        # ... and the four features of an entry are the four Lagrange chunks: dL/dtable[row][i] = lag_i * blend * G[row] with ONE
        # scalar sum per entry (nvsf_hashgrid4d_dynamic_bwd_scalar), expanded afterwards
        def scatter_sums(g_rows_or_cols, is_col_major):
            sums = [torch.zeros(s.n_rows, dtype=torch.float32, device=x.device) for s in specs]
            sum_ptrs = (ctypes.c_void_p * 3)(*[g.data_ptr() for g in sums])
            if is_col_major:
                try:
                    _hip.call("nvsf_hashgrid4d_dynamic_bwd_scalar_t", _hip.ptr(x), x.shape[1], M, h_scales, h_res, h_off, g_rows_or_cols.data_ptr(), sum_ptrs)
                    return sums
                except _hip.NvsfHipError:  # the launch would not take the LDS kernel: rows for the run-merging one
                    g_rows_or_cols = g_rows_or_cols.contiguous()
            _hip.call("nvsf_hashgrid4d_dynamic_bwd_scalar", _hip.ptr(x), x.shape[1], M, h_scales, h_res, h_off, _hip.ptr(g_rows_or_cols), sum_ptrs)
            return sums

        # Inside a training step the scatter and its expansion run on the step's side stream, straight into the gradient sink (round 6: the
        # step's main stream is its critical path -- 30.4 ms busy without a gap at 4096 + 4096 rays -- and these 2.5 ms per pass were on
        # it, while the side stream idled for 10 ms per step); nothing on the main stream reads these gradients before the optimiser.
        params, tctx = getattr(ctx, "params", None), getattr(ctx, "train_ctx", None)
        settled = False
        if params is not None and tctx is not None and len(params) == 6 and testing.get("hash4d_scatter") == "side":
            plan = [(params[i], i, 1.0) for i in range(3)] if same else [(params[i], i % 3, b_lo if i < 3 else b_hi) for i in range(6)]
            if len({id(p) for p, _, _ in plan}) == len(plan):
                def scatter(views, pool):
                    sums = scatter_sums(g_out, col_major)
                    for view, (_, pair, factor) in zip(views, plan):
                        view.view(-1, 4).addcmul_(sums[pair].view(-1, 1), lag_t.view(1, 4), value=factor)
                if _ops.scatter_beside_backward_multi(tctx, [p for p, _, _ in plan], (x, g_out, lag_t), scatter):
                    return (None,) * 6 + (None,) * len(params)
                settled = True  # (a refusal has settled them)
        if params is not None and tctx is not None and not settled:  # the gradients go back through autograd: the announced scatters are settled here
            seen = []
            for p in params:
                if isinstance(p, torch.nn.Parameter) and p.requires_grad and not any(p is q for q in seen):
                    seen.append(p)
                    tctx.done(p)
        sums = scatter_sums(g_out, col_major)
        acc = [(g.view(-1, 1) * lag_t.view(1, 4)).reshape(-1) for g in sums]
        if same:  # the same parameter tensors were passed twice: the whole gradient goes to the first occurrence
            grads = acc + [None, None, None]
        else:
            grads = [g * b_lo for g in acc] + [g * b_hi for g in acc]
        return (None, None, None, None, None, None, *grads)


class HashDyn3Fn(torch.autograd.Function):
    """The three space-time evaluations of one density query with autograd on the first: (hash_d, hash_1, hash_2) from ONE
    forward launch (k_hash_dynamic3); the neighbour outputs carry no gradient, as in the reference (network_dynamic.py:244-262
    evaluates them under no_grad), the table gradients of hash_d come from the fused backward kernel."""

    @staticmethod
    def forward(ctx, enc, x, t, t_host, k1, k2, offsets, nb1, nb2, *params):
        x = x.float().contiguous()
        out0, out1, out2 = enc.forward_dynamic3(x, t, t_host, offsets.detach(), [nb1, nb2])
        ctx.save_for_backward(x)
        ctx.enc, ctx.t, ctx.t_host, ctx.k1, ctx.k2 = enc, t, t_host, k1, k2
        HashDynFn._expect(ctx, enc, params)
        outs = [out0, out1 if out1 is not None else out0.new_zeros(0), out2 if out2 is not None else out0.new_zeros(0)]
        ctx.mark_non_differentiable(outs[1], outs[2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, grad_out, _g1, _g2):
        g = HashDynFn._backward(ctx, grad_out)  # (None x 6, *grads)
        return (None,) * 9 + tuple(g[6:])
