"""K-planes space-time encoder (`Planes4D`) for MI355X.

Same constructor, parameter tree (`planes.<scale>.<pair>` of shape [1, C, res_b, res_a]; time planes initialised
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
    def __init__(self, grid_dimensions=2, input_dim=4, output_dim=8, resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8],
                 concat_ms_feat=True, decompose=True, reduction="prod"):
        super().__init__()
        if grid_dimensions != 2 or input_dim != 4 or output_dim != 8 or not concat_ms_feat or reduction != "prod":
            raise NotImplementedError("Planes4D HIP kernel: 2-D planes of a 4-D input, 8 features, concatenated scales, product reduction")
        self.config = {"grid_dimensions": grid_dimensions, "input_dim": input_dim, "output_dim": output_dim, "resolution": resolution}
        self.multiscale_res, self.concat_ms_feat, self.decompose, self.reduction = multiscale_res, concat_ms_feat, decompose, reduction
        pairs = list(itertools.combinations(range(input_dim), grid_dimensions))
        assert tuple(pairs) == ops.PLANE_PAIRS
        self.planes = nn.ModuleList()
        res_host = []
        for mult in multiscale_res:
            reso = [r * mult for r in resolution[:3]] + list(resolution[3:])  # multi-resolution on the spatial axes only
            res_host += reso
            group = nn.ParameterList()
            for a, b in pairs:
                p = nn.Parameter(torch.empty(1, output_dim, reso[b], reso[a]))
                if b == 3:
                    nn.init.ones_(p)
                else:
                    nn.init.uniform_(p, a=0.1, b=0.5)
                group.append(p)
            self.planes.append(group)
        self._res_host = tuple(res_host)
        self.n_output_dims = output_dim * len(multiscale_res) * 2
        self._cl_key, self._cl = None, None

    def _flat_params(self):
        return [p for group in self.planes for p in group]

    def _channel_last(self):
        """One buffer with every plane as [H][W][C]; rebuilt only when a parameter changed."""
        params = self._flat_params()
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._cl_key:
            with torch.no_grad():
                self._cl = torch.cat([p.detach()[0].permute(1, 2, 0).reshape(-1) for p in params]).float().contiguous()
            self._cl_key = key
        return self._cl

    def _encode(self, xt, want):
        xt = xt.reshape(-1, 4)
        return ops.PlanesFn.apply(xt, self._channel_last(), self._res_host, want, *self._flat_params())

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
        import ctypes
        from nvsf import _hip
        x = x.float()
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.contiguous()
        M, n = x.shape[0], len(evals)
        width = self.n_output_dims // 2
        if out_f16 and not blend:
            raise ValueError("forward_multi: out_f16 needs blend=True")
        outs = [torch.empty(M, width, dtype=torch.float16 if out_f16 else torch.float32, device=x.device) for _ in (evals[:2] if blend else evals)]
        offs = []
        for _, o, _, _ in evals:
            if o is not None and (o.dtype != torch.float32 or o.dim() != 2 or o.stride(1) != 1):
                o = o.float().contiguous()
            offs.append(o)
        _hip.call("nvsf_planes_multi_fwd", _hip.ptr_rows(x), x.stride(0), M, _hip.ptr(self._channel_last()), len(self.multiscale_res), 8,
                  _hip.host_u32(self._res_host), n, _hip.host_i32([e[0] for e in evals]),
                  (ctypes.c_void_p * n)(*[None if o is None else o.data_ptr() for o in offs]),
                  _hip.host_u32([0 if o is None else o.stride(0) for o in offs]), _hip.host_u32([e[2] for e in evals]),
                  _hip.host_f32([e[3] for e in evals]), (ctypes.c_void_p * n)(*([t.data_ptr() for t in outs] + [None] * (n - len(outs)))),
                  (2 if out_f16 else 1) if blend else 0)
        return outs

    def forward_static(self, input):
        return self._encode(input, 1)

    def forward_dynamic(self, input):
        return self._encode(input, 2)

    def forward(self, input):
        s, d = self._encode(input, 3)
        return [s, d] if self.decompose else torch.cat([s, d], dim=-1)
