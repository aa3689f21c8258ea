"""K-planes space-time encoder (`Planes4D`) for MI355X.

Same constructor, checkpoint schema (`planes.<scale>.<pair>` of shape [1, C, res_b, res_a]; time planes initialised
to 1, spatial planes U(0.1, 0.5)), `n_output_dims` and `forward` / `forward_static` / `forward_dynamic` results as
/root/reference/nvsf/nerf/models/planes_field.py:142-238, so reference checkpoints load unchanged.  The 24
`F.grid_sample` launches + products + concatenations of the reference are one HIP kernel (csrc/planes.hip) that
reads a channel-last copy of the planes; the backward kernel returns gradients for the planes and for the
coordinates (the flow field is trained through the latter, network_dynamic.py:250-271).
"""
import itertools

import torch
import torch.nn as nn

from nvsf import field_ops as ops


class Planes4D(nn.Module):
    """Parameter storage.  The reference keeps 6 x n_scales parameters `planes.<scale>.<pair>` of shape [1, C, res_b, res_a]
    (planes_field.py:31-52, 172-190).  Here they are ONE parameter, `planes_cl`: every plane channel-last ([H][W][C], one texel =
    32 contiguous bytes = what the kernels gather), one after the other, fp32.  The kernels read it as it is (no per-step copy of
    24 permuted planes), the backward kernel's gradient buffer IS its gradient (no per-plane permute + accumulate launches), and
    the optimiser / EMA / gradient all-reduce see one tensor instead of 24 (the training step of the space-time model made ~200
    launches per step around the per-plane parameters).  state_dict()/load_state_dict() speak the reference's schema: the
    `planes.<scale>.<pair>` keys with [1, C, H, W] tensors (hooks below), so the `model` entry of a reference checkpoint loads
    unchanged and ours loads there; the position-ordered `optimizer` / `ema` entries are translated by nvsf/nerf/checkpoint_compat.py; `plane(scale, pair)` / `plane_grad(scale, pair)` give [1, C, H, W] views for inspection."""

    def __init__(self, grid_dimensions=2, input_dim=4, output_dim=8, resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8],
                 concat_ms_feat=True, decompose=True, reduction="prod"):
        super().__init__()
        if grid_dimensions != 2 or input_dim != 4 or output_dim != 8 or not concat_ms_feat or reduction != "prod":
            raise NotImplementedError("Planes4D HIP kernel: 2-D planes of a 4-D input, 8 features, concatenated scales, product reduction")
        self.config = {"grid_dimensions": grid_dimensions, "input_dim": input_dim, "output_dim": output_dim, "resolution": resolution}
        self.multiscale_res, self.concat_ms_feat, self.decompose, self.reduction = multiscale_res, concat_ms_feat, decompose, reduction
        pairs = list(itertools.combinations(range(input_dim), grid_dimensions))
        assert tuple(pairs) == ops.PLANE_PAIRS
        res_host, self._layout, pieces, off = [], [], [], 0
        for si, mult in enumerate(multiscale_res):
            reso = [r * mult for r in resolution[:3]] + list(resolution[3:])  # multi-resolution on the spatial axes only
            res_host += reso
            for pi, (a, b) in enumerate(pairs):
                p = torch.empty(1, output_dim, reso[b], reso[a])  # the reference's tensor, initialised as the reference does (same RNG stream)
                if b == 3:
                    nn.init.ones_(p)
                else:
                    nn.init.uniform_(p, a=0.1, b=0.5)
                pieces.append(p[0].permute(1, 2, 0).reshape(-1))
                self._layout.append((si, pi, off, output_dim, reso[b], reso[a]))
                off += p.numel()
        self.planes_cl = nn.Parameter(torch.cat(pieces).contiguous())
        self._res_host = tuple(res_host)
        self.n_output_dims = output_dim * len(multiscale_res) * 2
        self._register_state_dict_hook(Planes4D._export_reference_keys)
        self._register_load_state_dict_pre_hook(self._import_reference_keys)

    # ---- the reference's per-plane view of the one parameter ---------------------------------------------------------------
    def _view(self, flat, si, pi):
        _, _, off, C, H, W = self._layout[si * len(ops.PLANE_PAIRS) + pi]
        return flat[off:off + C * H * W].view(H, W, C).permute(2, 0, 1).unsqueeze(0)

    def plane(self, si, pi):
        """[1, C, res_b, res_a] view of plane (scale si, pair pi) -- the tensor the reference calls planes.<si>.<pi>."""
        return self._view(self.planes_cl.detach(), si, pi)

    def plane_grad(self, si, pi):
        return None if self.planes_cl.grad is None else self._view(self.planes_cl.grad, si, pi)

    def reference_named_parameters(self, prefix=""):
        """(name, value view, gradient view or None) under the reference's names."""
        for si, pi, *_ in self._layout:
            yield f"{prefix}planes.{si}.{pi}", self.plane(si, pi), self.plane_grad(si, pi)

    @staticmethod
    def _export_reference_keys(module, state, prefix, local_metadata):
        flat = state.pop(prefix + "planes_cl")
        for si, pi, *_ in module._layout:
            state[f"{prefix}planes.{si}.{pi}"] = module._view(flat, si, pi).contiguous()
        return state

    def _import_reference_keys(self, state, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        keys = [f"{prefix}planes.{si}.{pi}" for si, pi, *_ in self._layout]
        if prefix + "planes_cl" in state or not any(k in state for k in keys):
            return
        flat = self.planes_cl.detach().clone()
        for (si, pi, off, C, H, W), k in zip(self._layout, keys):
            if k in state:
                t = state.pop(k)
                if tuple(t.shape) != (1, C, H, W):
                    error_msgs.append(f"size mismatch for {k}: {tuple(t.shape)} in the checkpoint, {(1, C, H, W)} in the model")
                    continue
                flat[off:off + C * H * W] = t.to(flat.device, flat.dtype)[0].permute(1, 2, 0).reshape(-1)
            elif strict:
                missing_keys.append(k)
        state[prefix + "planes_cl"] = flat

    def wait_pending_update(self):
        """A training step may leave the optimiser pass of this parameter on its table-scatter stream (RenderTrainStep: the planes
        scattered in the last backward pass are updated behind their scatter, like the hash tables).  The first reader of the
        parameter makes ITS stream wait for that pass here (the hash tables do the same through their fp16 cache)."""
        ev = self.__dict__.get("_pending_update")
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.__dict__["_pending_update"] = None

    def _channel_last(self):
        """The buffer the kernels read: the parameter itself."""
        self.wait_pending_update()
        p = self.planes_cl
        return p if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous()

    def _encode(self, xt, want):
        xt = xt.reshape(-1, 4)
        self.wait_pending_update()
        return ops.PlanesFn.apply(xt, self.planes_cl, self._res_host, want, ops.train_context(self))

    @torch.no_grad()
    def forward_multi(self, x, evals, blend=False, out_f16=False):
        """Several evaluations of one position set in ONE launch, without autograd (nvsf_planes_multi_fwd): `evals` is a list of
        (group, offsets, offset_col, time) with group 0 = static / 1 = dynamic planes, offsets = None or an fp32 [M, >= col + 3]
        tensor added to x[:, :3] (the scene flow towards a neighbour frame), time = the frame time as a Python float.  Returns the
        list of fp32 [M, n_output_dims / 2] feature matrices.  Same values as forward_static / forward_dynamic on
        cat([x + offsets[:, col:col+3], time]).  blend=True: `evals` = (static, dynamic, dynamic at neighbour 1, dynamic at
        neighbour 2) and the result is [static, 0.5 d + 0.25 (d1 + d2)] (the blend of network_dynamic.py:273, formed in the kernel:
        the neighbour features never reach memory).  out_f16 (with blend): the two results as fp16 rows, rounded as the density
        kernel rounds its inputs (nvsf_density_dynamic_f16planes_fwd reads them)."""
        return ops.planes_multi_forward(x, evals, self._channel_last(), self._res_host, blend=blend, out_f16=out_f16)

    def forward_static(self, input):
        return self._encode(input, 1)

    def forward_dynamic(self, input):
        return self._encode(input, 2)

    def forward(self, input):
        s, d = self._encode(input, 3)
        return [s, d] if self.decompose else torch.cat([s, d], dim=-1)
