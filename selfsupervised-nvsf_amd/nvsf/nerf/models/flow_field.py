"""Scene-flow field (`FlowField`) for MI355X: 3-D hash grid (L16 F8, HIP kernel) -> cubic Lagrange reduction over
time -> bias-free MLP 32 -> 64 -> 64 -> 6 (forward and backward flow).  Same constructor, parameter names
(`grid_enc.params`, `mlp.<i>.weight`) and arithmetic as /root/reference/nvsf/nerf/models/flow_field.py:41-133.
The three small dense layers are `nn.Linear` modules as in the reference (same state_dict keys; last layer
initialised N(0, 1e-3), :103).  Which arithmetic they run in follows the caller's precision regime, as in the reference:
fp32 (torch) by default; fp16 operands with fp32 accumulation on the fused MFMA MLP kernels -- what `torch.cuda.amp.autocast`
makes of a Linear layer -- when the call runs inside an autocast region (the reference's Trainer wraps every step in
`autocast(enabled=opt.fp16)`, trainer.py:1318, 1491), when `fp16=True` reaches `render` (the reference passes `**vars(self.opt)`,
trainer.py:200, and its shipped config sets `fp16`: configs/kitti360_1908.txt), or when the module's `flow_mlp_mode` is "fused"
(RenderTrainStep(fp16=True))."""
import numpy as np
import torch
import torch.nn as nn

import tinycudann as tcnn
from nvsf import field_ops as ops
from nvsf.nerf.models.hash_field import lagrange_reduce


class FreqEmbed(nn.Module):
    """sin / cos embedding of every input column (flow_field.py:17-38): output columns [sin(f_0 pi x) | sin(f_1 pi x) | ... |
    cos(f_0 pi x) | ...], each block as wide as x; f = linspace(1, n + 1, n) or 2^(0 .. n-1).  An option of FlowField that no
    configuration of the reference switches on: plain tensor algebra here as there."""

    def __init__(self, num_freqs, linspace=True):
        super().__init__()
        self.freqs = torch.linspace(1, num_freqs + 1, steps=num_freqs) if linspace else 2 ** torch.linspace(0, num_freqs - 1, steps=num_freqs)

    def forward(self, x):
        arg = torch.cat([f * x * torch.pi for f in self.freqs], -1)  # (freq * x) * pi with freq a 0-dim CPU tensor, as in the reference
        return torch.cat([torch.sin(arg), torch.cos(arg)], -1)


class FlowField(nn.Module):
    def __init__(self, input_dim=4, num_layers=3, hidden_dim=64, use_freq=False, num_freqs=6, use_grid=True, num_basis=4,
                 n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=8192, log2_hashmap_size=18):
        super().__init__()
        if not (use_freq or use_grid):
            raise ValueError("FlowField needs at least one of use_freq / use_grid")
        if use_freq and use_grid:
            # the reference's own forward fails for this pair of options: interpT views the grid features with the SUMMED input
            # width (flow_field.py:121) and the concatenation at :130 raises -- there is no behaviour to reproduce
            raise NotImplementedError("FlowField(use_freq=True, use_grid=True) does not run in the reference (flow_field.py:121, 130)")
        self.use_freq, self.use_grid = use_freq, use_grid
        self.input_dim = 0
        if use_freq:
            self.freq_enc = FreqEmbed(num_freqs=num_freqs)
            self.input_dim += input_dim * num_freqs * 2
        self.n_levels, self.n_features_per_level, self.num_basis = n_levels, n_features_per_level, num_basis
        if use_grid:
            per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
            self.grid_enc = tcnn.Encoding(n_input_dims=3, encoding_config={
                "otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features_per_level,
                "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution, "per_level_scale": per_level_scale})
            self.input_dim += self.grid_enc.n_output_dims // num_basis
        layers = []
        for l in range(num_layers):
            last = l == num_layers - 1
            layers.append(nn.Linear(self.input_dim if l == 0 else hidden_dim, 6 if last else hidden_dim, bias=False))
            if not last:
                layers.append(nn.ReLU())
        self.mlp = nn.Sequential(*layers)
        torch.nn.init.normal_(self.mlp[-1].weight.data, 0, 0.001)

    def forward(self, xt, t_host=None, fp16=None):
        """xt: [N, 4] = (x, y, z, t) in [0, 1]; all rows share one t (the reference reads xt[0, 3], :125).
        Without autograd the grid lookup and the Lagrange reduction are one fused kernel (csrc/hashgrid4d.hip);
        `t_host` (the value of t, if the caller already has it on the host) avoids a device->host read.
        `fp16`: True / False selects the MLP's arithmetic for this call, None follows the regime (module docstring)."""
        t = xt[0, 3]
        fused_mlp = self.mlp_mode(fp16) == "fused" and self._fused_mlp_ok()
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        h = [self.freq_enc(xt.float())] if self.use_freq else []
        if self.use_grid:
            if training:
                if self.grid_train_mode == "fused" and self.n_features_per_level == 8 and self.num_basis == 4 and not xt.requires_grad:
                    # grid lookup + Lagrange reduction as the fused forward kernel, the table gradient straight from dL/d(reduced)
                    t_h = float(t) if t_host is None else t_host
                    h.append(FlowGridFn.apply(self, xt.float().contiguous(), t_h, self.grid_enc.params))
                else:
                    feat = self.grid_enc.encode_columns(xt, (0, 1, 2)).float()
                    h.append(lagrange_reduce(feat, t, self.n_levels, self.n_features_per_level, self.num_basis))
            else:
                if self.n_features_per_level != 8 or self.num_basis != 4:
                    raise NotImplementedError("fused flow grid kernel: 8 features per level, 4 Lagrange nodes")
                t_host = float(t) if t_host is None else t_host
                h.append(self._grid_lagrange(xt.float().contiguous(), t_host))
        red = h[0] if len(h) == 1 else torch.cat(h, dim=-1)
        if training:
            if fused_mlp:
                # the mixed-precision training run: the Linear layers on the fused MFMA forward / backward kernels (what
                # autocast makes of them in the reference's Trainer), instead of three fp32 GEMMs + two ReLU launches forward
                # and six GEMMs backward whose weight gradients reduce over millions of rows
                lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
                return FlowMlpFn.apply(red, lin[0].weight, lin[1].weight, lin[2].weight)
            return self.mlp(red)
        if fused_mlp:
            # the fp16 regime: the three bias-free layers as ONE fused MFMA kernel -- fp16 operands, fp32 accumulation, i.e. what the
            # reference's Linear layers compute under the Trainer's autocast (6x faster than three fp32 GEMM + two ReLU launches;
            # fp16-accurate: 4e-5 abs on flows of 1e-2 against the fp32 form).  Columns 6..15 are padding.
            return ops.mlp_forward(red.contiguous(), self._mlp_weights_f16(), self._mlp_spec)[:, :6]
        return self.mlp(red)

    def _grid_lagrange(self, xt, t_host):
        """[M, >=3] positions -> fp32 [M, 2 L]: 3-D grid lookup + cubic Lagrange reduction over the 4 feature chunks (one kernel)."""
        from nvsf import _hip
        from nvsf.nerf.models.hash_field import lagrange_weights_host
        spec = self.grid_enc.spec
        M = xt.shape[0]
        red = torch.empty(M, 2 * spec.L, dtype=torch.float32, device=xt.device)
        _hip.call("nvsf_hashgrid3d_lagrange_fwd", _hip.ptr(xt), xt.shape[1], M, _hip.ptr(self.grid_enc.table_f16()), spec.L, spec.F,
                  spec.h_scales, spec.h_res, spec.h_offsets, _hip.host_f32(lagrange_weights_host(t_host, 4, xt.is_cuda)), _hip.ptr(red))
        return red

    flow_mlp_mode = "auto"     # "auto": follow the caller's regime; "fused" / "torch": pinned (RenderTrainStep(fp16=True) pins "fused")
    grid_train_mode = "fused"  # "fused": FlowGridFn; "chain" (tests): encoder + lagrange_reduce under plain autograd

    def mlp_mode(self, fp16=None):
        """"torch" (fp32 Linear layers) or "fused" (fp16 MFMA kernels = the reference's Linear layers under autocast).
        In the reference precision is decided by the Trainer's CUDA autocast region alone (trainer.py:1332, 1487); `fp16` reaches
        `render` through **vars(opt) and is ignored there.  Here: "auto" follows the CUDA autocast state (not the CPU one), which is
        the reference's behaviour; an explicit `fp16=` argument of render / forward is an EXTENSION of this package that pins the
        regime without an autocast region (tests, tools); the fixture-pinned default outside autocast is the fp32 form."""
        if fp16 is not None:
            return "fused" if fp16 else "torch"
        if self.flow_mlp_mode != "auto":
            return self.flow_mlp_mode
        try:
            on = torch.is_autocast_enabled("cuda")
        except TypeError:  # older torch: one device-agnostic flag
            on = torch.is_autocast_enabled()
        return "fused" if on else "torch"

    def _fused_mlp_ok(self):
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        return (len(lin) == 3 and lin[0].in_features % 16 == 0 and lin[0].in_features <= 128 and lin[0].out_features == 64
                and lin[1].out_features == 64 and lin[2].out_features <= 16)

    def _mlp_weights_f16(self):
        """fp16 copy of the Linear weights in the fused kernel's layout (W0 [64, in] ++ W1 [64, 64] ++ W2 zero-padded to
        [16, 64], row-major = nn.Linear's [out, in]); rebuilt when a weight changes."""
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        key = tuple((l.weight.data_ptr(), l.weight._version) for l in lin)
        if getattr(self, "_mlp_key", None) != key:
            w2 = torch.zeros(16, 64, dtype=torch.float32, device=lin[2].weight.device)
            w2[:lin[2].out_features] = lin[2].weight.detach().float()
            self._mlp_w16 = torch.cat([lin[0].weight.detach().float().reshape(-1), lin[1].weight.detach().float().reshape(-1),
                                       w2.reshape(-1)]).to(torch.float16).contiguous()
            self._mlp_spec = ops.MlpSpec(lin[0].in_features, lin[2].out_features, hidden=64, n_hidden=2)
            self._mlp_key = key
        return self._mlp_w16


def _pack_flow_weights(w0, w1, w2):
    """nn.Linear weights [out, in] -> the fused kernels' fp16 layout W0 [64, in] ++ W1 [64, 64] ++ W2 zero-padded to [16, 64]."""
    w2p = torch.zeros(16, 64, dtype=torch.float32, device=w2.device)
    w2p[:w2.shape[0]] = w2.detach().float()
    return torch.cat([w0.detach().float().reshape(-1), w1.detach().float().reshape(-1), w2p.reshape(-1)]).to(torch.float16).contiguous()


class FlowMlpFn(torch.autograd.Function):
    """flow = W2 relu(W1 relu(W0 x)) on nvsf_mlp_fwd / nvsf_mlp_bwd; gradients are returned in the Linear layers' shapes."""

    @staticmethod
    def forward(ctx, x, w0, w1, w2):
        spec = ops.MlpSpec(w0.shape[1], w2.shape[0], hidden=64, n_hidden=2)
        w16 = _pack_flow_weights(w0, w1, w2)
        out = ops.mlp_forward(x, w16, spec)
        ctx.save_for_backward(x, w16)
        ctx.spec, ctx.shapes = spec, (w0.shape, w1.shape, w2.shape)
        return out[:, :spec.n_out]

    @staticmethod
    def backward(ctx, grad_out):
        x, w16 = ctx.saved_tensors
        spec, (s0, s1, s2) = ctx.spec, ctx.shapes
        grad_x, gw = ops.mlp_backward(x, w16, spec, grad_out, need_grad_x=ctx.needs_input_grad[0])
        g0, g1, g2 = spec.split(gw)
        return grad_x, g0.view(s0), g1.view(s1), g2[:s2[0]].contiguous().view(s2)


class FlowGridFn(torch.autograd.Function):
    """reduced = Lagrange_t(grid(x)) for the flow field with autograd on the grid table: forward = the fused kernel of the
    no-grad path; backward expands dL/d(reduced) [M, 2L] to dL/d(features) [M, 8L] (feature 2i+e of a level gets w_i times
    the gradient of reduced column e) and scatters it with the hash-grid backward kernel."""

    @staticmethod
    def forward(ctx, field, xt, t_host, params):
        ctx.save_for_backward(xt)
        ctx.field, ctx.t_host, ctx.rows_per_ray = field, t_host, ops.rows_hint(field)
        ctx.train_ctx, ctx.table_param = ops.train_context(field), params
        if ctx.train_ctx is not None and params.requires_grad and torch.is_grad_enabled():
            ctx.train_ctx.expect(params)
        return field._grid_lagrange(xt, t_host)

    @staticmethod
    def backward(ctx, grad_red):
        from nvsf.nerf.models.hash_field import lagrange_weights_host
        (xt,) = ctx.saved_tensors
        field = ctx.field
        spec = field.grid_enc.spec
        w = ops.device_constant(lagrange_weights_host(ctx.t_host, 4, xt.is_cuda), xt.device)
        # feature 2i+e of a level receives w_i * dL/d(reduced column e): the four chunks of an entry get the same scattered sum
        # up to the scalar w_i.  Scatter G[row][e] = sum g_e w_corner ONCE on a 2-feature view of the grid (a quarter of the
        # atomics, which bound this pass) and expand to the 8 features afterwards.
        spec2 = field.__dict__.get("_spec2")
        if spec2 is None or spec2.n_rows != spec.n_rows:
            import copy
            spec2 = copy.copy(spec)
            spec2.F, spec2.n_params, spec2.n_output_dims = 2, spec.n_rows * 2, spec.L * 2
            field.__dict__["_spec2"] = spec2
        g = grad_red.float().contiguous()
        fine = ops._bin_from(spec2, xt.shape[0], ctx.rows_per_ray)

        def scatter(view, pool):  # on the step's side stream: scatter the 2-feature sums, expand them into the table's gradient
            G = ops.hashgrid_backward(xt, (0, 1, 2), spec2, g, fine_from=fine, ws_pool=pool)
            view.view(-1, 4, 2).addcmul_(G.view(-1, 1, 2), w.view(1, 4, 1))
        if ops.scatter_beside_backward(ctx.train_ctx, ctx.table_param, (xt, g, w), scatter):
            return None, None, None, None
        G = ops.hashgrid_backward(xt, (0, 1, 2), spec2, g, fine_from=fine)
        grad_table = (G.view(-1, 1, 2) * w.view(1, 4, 1)).reshape(-1)
        return None, None, None, grad_table
