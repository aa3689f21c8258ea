"""TEST INFRASTRUCTURE / CPU baseline -- never imported by the product (selfsupervised-nvsf_amd/).

A vectorised PyTorch-CPU restatement of the static render path: what the reference's `NeRFRenderer.run` would execute on a CPU if
its tiny-cuda-nn modules were PyTorch modules -- the reference's own tensor algebra for sampling and compositing
(nvsf/nerf/models/renderer_dynamic.py:155-237: linspace samples, clip to the box, alpha = 1 - exp(-delta sigma), cumprod
transmittance, weights > 1e-4 mask, weighted sums) around torch formulations of the hash-grid encoder, the direction encodings and
the bias-free ReLU MLPs as DESIGN.md section 4 specifies them (fp16 tables and weights, fp16-rounded inputs and hidden
activations, fp32 accumulation; torch's matmul decides the summation order, so results agree with oracle/*.c to the fp16 level,
not bit for bit: tests/test_oracle_cpu.py pins it at 1e-4 on the rendered images).

bench.py times it beside the scalar C port (`cpu_baseline.torch_cpu`): all host threads and one thread.  BASELINE.md section 3
planned exactly this figure; the scalar port remains the checker of the GPU outputs."""
import math

import numpy as np
import torch

_P1, _P2 = 2654435761, 805459861


def hashgrid(x01, table_f16, spec):
    """x01 [M,3] fp32 in [0,1] -> [M, L*F] features (fp16-rounded, returned as fp32)."""
    M = x01.shape[0]
    F = spec.F
    table = table_f16.view(-1, F)
    out = torch.empty(M, spec.L * F, dtype=torch.float32)
    for l in range(spec.L):
        scale, res = np.float32(spec.scales[l]), spec.res[l]
        off, rows = spec.offsets[l], spec.offsets[l + 1] - spec.offsets[l]
        pos = x01 * float(scale) + 0.5
        cell = torch.floor(pos)
        frac = pos - cell
        c = cell.to(torch.int64)
        dense = res ** 3 <= rows
        acc = torch.zeros(M, F, dtype=torch.float32)
        for k in range(8):
            b = [(k >> d) & 1 for d in range(3)]
            cc = [c[:, d] + b[d] for d in range(3)]
            if dense:
                idx = (cc[0] + cc[1] * res + cc[2] * res * res) % rows
            else:
                idx = ((cc[0] & 0xFFFFFFFF) ^ ((cc[1] * _P1) & 0xFFFFFFFF) ^ ((cc[2] * _P2) & 0xFFFFFFFF)) % rows
            w = (frac[:, 0] if b[0] else 1 - frac[:, 0]) * (frac[:, 1] if b[1] else 1 - frac[:, 1]) * (frac[:, 2] if b[2] else 1 - frac[:, 2])
            acc += w[:, None] * table[off + idx].float()
        out[:, l * F:(l + 1) * F] = acc.half().float()
    return out


def mlp(x, weights_f16, spec):
    """x [M, n_in] (rounded to fp16, padded with ones to in_cols) -> fp32 [M, out_cols]; hidden activations fp16-rounded."""
    M = x.shape[0]
    a = torch.ones(M, spec.in_cols, dtype=torch.float32)
    a[:, :spec.n_in] = x.half().float()
    mats = spec.split(weights_f16.float())
    for W in mats[:-1]:
        a = torch.relu(a @ W.t()).half().float()
    return a @ mats[-1].t()


def freq_encode(d01, n_freq=12):
    k = torch.arange(n_freq, dtype=torch.float64)
    a = d01.double()[:, :, None] * (2.0 ** k)[None, None, :] * math.pi
    return torch.stack([torch.sin(a), torch.cos(a)], -1).reshape(d01.shape[0], -1).float().half().float()


def sh4_encode(d01):
    x, y, z = (d01[:, i] * 2.0 - 1.0 for i in range(3))
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = [torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
         1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
         0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
         0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
         1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    return torch.stack(o, -1).half().float()


def render_static(rays_o, rays_d, nears, fars, T, bound, table_f16, grid_spec, w_sigma, sigma_spec, lidar, w_a, w_b, head_spec,
                  bg=1.0, k_scale=1.0, w_thresh=1e-4):
    """rays [N,3], nears / fars [N] (torch fp32, CPU) -> image [N, 2 | 3], depth [N], weights_sum [N]
    (renderer_dynamic.py:155-237 with the field of network_dynamic.py:213-332 restricted to its static hash branch)."""
    N = rays_o.shape[0]
    z = nears[:, None] + (fars - nears)[:, None] * torch.linspace(0.0, 1.0, T)[None, :]
    sample_dist = (fars - nears) / T
    xyz = (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).clamp(-bound, bound)
    x01 = ((xyz.reshape(-1, 3) + bound) / (2 * bound))
    h = mlp(hashgrid(x01, table_f16, grid_spec), w_sigma, sigma_spec)
    sigma = torch.exp(h[:, 0]).view(N, T)
    geo = h[:, 1:16]
    deltas = torch.cat([z[:, 1:] - z[:, :-1], sample_dist[:, None]], -1)
    alphas = 1 - torch.exp(-deltas * k_scale * sigma)
    weights = alphas * torch.cumprod(torch.cat([torch.ones(N, 1), 1 - alphas + 1e-15], -1), -1)[:, :-1]
    mask = (weights > w_thresh).reshape(-1)
    d01 = (rays_d + 1) / 2
    enc = (freq_encode(d01) if lidar else sh4_encode(d01)).repeat_interleave(T, 0)
    C = 2 if lidar else 3
    rgbs = torch.zeros(N * T, C)
    if bool(mask.any()):
        inp = torch.cat([enc[mask], geo[mask]], -1)
        if lidar:
            logits = torch.cat([mlp(inp, w_a, head_spec)[:, :1], mlp(inp, w_b, head_spec)[:, :1]], -1)
        else:
            logits = mlp(inp, w_a, head_spec)[:, :3]
        rgbs[mask] = torch.sigmoid(logits)
    ws = weights.sum(-1)
    depth = (weights * z).sum(-1)
    image = (weights[:, :, None] * rgbs.view(N, T, C)).sum(1)
    if not lidar:
        image = image + (1 - ws)[:, None] * bg
    return image, depth, ws
