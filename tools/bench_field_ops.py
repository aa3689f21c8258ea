"""Roofline of the stand-alone field operators (the tiny-cuda-nn surface and K-planes, plus
the space-time kernels), the ones the training path and the dynamic model launch one by one.  Samples are the
config-2 batches (4096 rays x 768 uniform samples along LiDAR / camera rays: the spatial coherence the gathers see in
the renderer).  Bytes / flops per sample are SURVEY.md 8(d)'s algorithmic figures; HIP-event time on the launch stream.

    python tools/bench_field_ops.py      -> one JSON line, a row per operator
    bench.py imports field_op_rooflines() for its `field_ops` leg.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = 2500.0


def _time_ms(fn, iters=10):
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


L1_PEAK_GBS = 64.0 * 256 * 2.4  # vector L1 -> registers: 64 B/clk/CU x 256 CUs x 2.4 GHz = 39.3 TB/s (MI355X_MICROARCH.md)


def _l1(kernel, ms, per_sample_bytes, M, per_unit):
    """A gather whose table stays in L2 / Infinity Cache moves more bytes than HBM could deliver: priced against the rate of
    the path that does bound it, the L1 -> register data path (a fraction above 1 of the HBM peak carries no information)."""
    gbs = per_sample_bytes * M / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "ms": ms, "bound": "l1", "unit": "GB/s", "achieved": gbs, "peak": L1_PEAK_GBS, "frac": gbs / L1_PEAK_GBS,
            "per_unit": per_unit, "units": M}


def _hbm(kernel, ms, per_sample_bytes, M, per_unit):
    gbs = per_sample_bytes * M / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "ms": ms, "bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
            "per_unit": per_unit, "units": M}


def _mlp(kernel, ms, flop_per_sample, bytes_per_sample, M, per_unit):
    """A stand-alone MLP launch streams its rows through HBM: below the ridge (2500 TFLOP/s / 8 TB/s = 312 FLOP/B) the
    bytes bound it, not the MFMA pipe; both figures are reported, `bound` names the binding one."""
    tf = flop_per_sample * M / (ms * 1e-3) / 1e12
    gbs = bytes_per_sample * M / (ms * 1e-3) / 1e9
    hbm_bound = flop_per_sample / bytes_per_sample < MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    row = {"kernel": kernel, "ms": ms, "per_unit": per_unit, "units": M, "tflops": tf, "mfma_frac": tf / MFMA_PEAK_TFLOPS, "gbs": gbs,
           "hbm_frac": gbs / HBM_PEAK_GBS}
    if hbm_bound:
        row.update({"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS})
    else:
        row.update({"bound": "mfma", "unit": "TFLOP/s", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "frac": tf / MFMA_PEAK_TFLOPS})
    return row


def _bin_contributions(x01, spec, plan, T):
    """Number of (row, values) records the binned scatter writes for ray-ordered rows x01 [M, 3]: 8 per row on the levels from
    fine_from on; on the run-sum levels 8 per RUN = up to 8 consecutive rows of a wave (64 consecutive rows) that share a cell."""
    merge_from, fine_from = plan
    M = x01.shape[0]
    total = 8 * M * (spec.L - fine_from)
    idx = torch.arange(M, device=x01.device)
    for l in range(merge_from, fine_from):
        cell = torch.floor(x01 * spec.scales[l] + 0.5).to(torch.int64)
        key = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
        new = torch.ones(M, dtype=torch.bool, device=x01.device)
        new[1:] = key[1:] != key[:-1]
        new |= (idx % 64) == 0
        starts = torch.nonzero(new).squeeze(1)
        lengths = torch.diff(starts, append=torch.tensor([M], device=x01.device))
        total += 8 * int(((lengths + 7) // 8).sum())
    return int(total)


def field_op_rooflines(dev, n_rays=4096, T=768, seed=0):
    import tinycudann as tcnn
    from nvsf import field_ops as ops, synthetic as S
    from nvsf.nerf.models.hash_field import HashGrid4D
    from nvsf.nerf.models.planes_field import Planes4D
    from nvsf.nerf.models.flow_field import FlowField
    rows = []
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    aabb = torch.tensor([-S.BOUND] * 3 + [S.BOUND] * 3, dtype=torch.float32, device=dev)

    def samples(lidar):
        o, d = (S.lidar_rays if lidar else S.camera_rays)(n_rays, rng)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        if lidar:
            nears = torch.full((n_rays,), float(S.MIN_NEAR), device=dev)
            fars = torch.full((n_rays,), float(S.LIDAR_MAX_DEPTH), device=dev)
        else:
            from nvsf.nerf.raymarching import raymarching
            nears, fars = raymarching.near_far_from_aabb(o, d, aabb, float(S.MIN_NEAR))
        _, xyz = ops.uniform_samples(o, d, nears, fars, T, aabb, None)
        x01 = ((xyz.view(-1, 3) + S.BOUND) / (2 * S.BOUND)).contiguous()
        dirs = d[:, None, :].expand(n_rays, T, 3).reshape(-1, 3).contiguous()
        return x01, dirs

    xl, dl = samples(True)
    xc, dc = samples(False)
    M = xl.shape[0]

    # ---- a12: 3-D hash grids (config 2: L16 F2; reference default: L8 F4), forward and table gradient
    for name, cfg in (("C2 L16 F2", dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                                         per_level_scale=float(np.exp2(np.log2(2048 / 16) / 15)))),
                      ("RD L8 F4", dict(n_levels=8, n_features_per_level=4, log2_hashmap_size=19, base_resolution=512,
                                        per_level_scale=float(np.exp2(np.log2(32768 / 512) / 7))))):
        enc = tcnn.Encoding(3, dict(otype="HashGrid", **cfg)).to(dev)
        table = enc.table_f16()
        for tag, x in (("lidar", xl), ("camera", xc)):
            out = torch.empty(M, enc.n_output_dims, dtype=torch.float16, device=dev)
            ms = _time_ms(lambda: ops.hashgrid_forward(x, (0, 1, 2), table, enc.spec, out=out))
            rows.append(_hbm(f"hashgrid_fwd[{name}, {tag}]", ms, 512 + 12 + 64, M, "588 B/sample (8 corners x 32 features x 2 B gathered + 12 + 64)"))
            g = torch.randn(M, enc.n_output_dims, device=dev).half()
            gt = torch.zeros(enc.spec.n_params, dtype=torch.float32, device=dev)
            # (1) what a training step LAUNCHES at this size: the plan of field_ops._bin_from (run sums + per-row bins for the fine levels,
            # run-merging atomics below) fed with the density MLP's level-major fp32 gradient [L, M, F] (mlp_backward(grad_x_blocks=F))
            plan = ops._bin_from(enc.spec, M, T)
            g_lm = g.float().view(M, enc.spec.L, enc.spec.F).permute(1, 0, 2).contiguous()
            if plan is not None:
                ms = _time_ms(lambda: ops.hashgrid_backward(x, (0, 1, 2), enc.spec, g_lm, grad_table=gt, fine_from=plan), 5)
                row = _hbm(f"hashgrid_bwd[{name}, {tag}]", ms, 8 * 32 * 4 + 12 + 128, M,
                           "production plan (merge_from, fine_from) = %s, level-major fp32 gradient; 1164 B/sample (8 corners x 32 features x 4 B added "
                           "+ 12 + 128 read)" % (plan,))
                contrib = _bin_contributions(x, enc.spec, plan, T)
                rec = 2 + 4 * enc.spec.F
                row.update({"plan": list(plan), "bin_contributions": contrib, "contribution_bytes": 2 * rec * contrib,
                            "contribution_GBs": 2 * rec * contrib / (ms * 1e-3) / 1e9,
                            "contribution_note": "%d-byte (row, values) records of the binned levels, written once and read once; the time also covers "
                                                 "the atomics of the levels below merge_from" % rec})
                rows.append(row)
                ms = _time_ms(lambda: ops.hashgrid_backward(x, (0, 1, 2), enc.spec, g, grad_table=gt, fine_from=plan), 5)
                rows.append(_hbm(f"hashgrid_bwd[{name}, {tag}, row-major fp16 gradient]", ms, 8 * 32 * 4 + 12 + 64, M,
                                 "same plan fed with [M, L F] fp16 rows (HashGridFn outside DensityFn)"))
            # (2) the fallback every level takes below 2^18 rows / without ray order: run-merging fp32 atomics
            ms = _time_ms(lambda: ops.hashgrid_backward(x, (0, 1, 2), enc.spec, g, grad_table=gt), 5)
            rows.append(_hbm(f"hashgrid_bwd[{name}, {tag}, atomics only]", ms, 8 * 32 * 4 + 12 + 64, M,
                             "NOT what a full-size step launches: every level through nvsf_hashgrid_bwd (the form small batches take); 1100 B/sample "
                             "(8 corners x 32 features x 4 B of fp32 atomics + 12 + 64 read); ceiling = atomic rate ~1.3 TB/s"))
        del enc, table

    # ---- a16: fused MLPs at the shapes of the model, aligned fp16 rows
    def mlp_rows(name, n_in, n_out, n_hidden):
        spec = ops.MlpSpec(n_in, n_out, 64, n_hidden)
        w16 = ((torch.rand(spec.n_params, device=dev) * 2 - 1) * 0.2).half()
        x = torch.randn(M, spec.in_cols, device=dev).half()[:, :n_in]
        flops = 2 * sum(a * b for a, b in spec.shapes)
        ms = _time_ms(lambda: ops.mlp_forward(x, w16, spec))
        rows.append(_mlp(f"mlp_fwd[{name}]", ms, flops, 2 * spec.in_cols + 64, M,
                         f"{flops} FLOP/sample; {2 * spec.in_cols + 64} B/sample (fp16 rows in, fp32 logits out)"))
        if n_hidden <= 2:
            g = torch.randn(M, n_out, device=dev) * 0.01
            gx = torch.empty(M, (n_in + 3) // 4 * 4, device=dev)[:, :n_in]  # 16-byte aligned rows, as ops.mlp_backward allocates them
            ms = _time_ms(lambda: ops.mlp_backward(x, w16, spec, g, grad_x=gx), 5)
            nbytes = 2 * spec.in_cols + 4 * n_out + 4 * n_in
            rows.append(_mlp(f"mlp_bwd[{name}]", ms, 3 * flops, nbytes, M,
                             f"{3 * flops} FLOP/sample (recomputed forward + data + weight gradients); {nbytes} B/sample (rows in, dL/dout in, dL/dx out)"))
    mlp_rows("sigma C2 32-64-16", 32, 16, 1)
    mlp_rows("sigma RD 120-64-16", 120, 16, 1)
    mlp_rows("lidar head 87-64-64-1", 87, 1, 2)
    mlp_rows("colour 31-64-64-3", 31, 3, 2)

    # ---- a17: direction encodings
    d01 = (dl + 1) / 2
    out = torch.empty(M, 72, dtype=torch.float16, device=dev)
    ms = _time_ms(lambda: ops.freq_encode(d01, 12, out=out))
    rows.append(_hbm("frequency_encode[3 -> 72]", ms, 12 + 144, M, "156 B/sample (12 read, 72 x 2 B written)"))
    out = torch.empty(M, 16, dtype=torch.float16, device=dev)
    ms = _time_ms(lambda: ops.sh4_encode(d01, out=out))
    rows.append(_hbm("sh4_encode[3 -> 16]", ms, 12 + 32, M, "44 B/sample"))
    del d01, out

    # ---- a13 / a14 / a15: space-time encoders of the reference-default model
    with torch.no_grad():
        h4 = HashGrid4D(base_resolution=512, max_resolution=32768, time_resolution=8, n_levels=8, n_features_per_level=4, log2_hashmap_size=19).to(dev)
        t = torch.tensor([[0.5]], device=dev)
        for tag, x in (("lidar", xl), ("camera", xc)):
            ms = _time_ms(lambda: h4.forward_dynamic(x, t, 0.5), 5)
            rows.append(_hbm(f"hashgrid4d_dynamic_fwd[{tag}]", ms, 3 * 2 * 4 * 8 * 8 + 12 + 96, M,
                             "1644 B/sample (3 pairs x 2 slices x 4 corners x 8 levels x 8 B gathered + 12 + 24 x 4 written)"))
        pl = Planes4D(grid_dimensions=2, input_dim=4, output_dim=8, resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8],
                      concat_ms_feat=True, decompose=True).to(dev)
        for tag, x in (("lidar", xl), ("camera", xc)):
            xt = torch.cat([x, torch.full((M, 1), 0.5, device=dev)], dim=-1)
            ms = _time_ms(lambda: pl(xt), 5)
            rows.append(_l1(f"planes_fwd[{tag}]", ms, 3072 + 16 + 256, M, "3344 B/sample (6 planes x 4 scales x 4 texels x 32 B gathered + 16 + 2 x 128 "
                            "written); the 8.7 MB of planes are served from L2 / Infinity Cache, so the kernel is priced against the L1 -> register data "
                            "path (64 B/clk/CU), not against HBM"))
        fl = FlowField().to(dev)
        for tag, x in (("lidar", xl), ("camera", xc)):
            xt = torch.cat([x, torch.full((M, 1), 0.5, device=dev)], dim=-1)
            ms = _time_ms(lambda: fl(xt, 0.5), 5)
            rows.append(_hbm(f"flow_field[{tag}]", ms, 8 * 16 * 16 + 16 + 24, M,
                             "2088 B/sample (8 corners x 16 levels x 16 B + 16 + 24) + 13 056 FLOP/sample of fp32 MLP; grid kernel + 3 GEMMs + 2 ReLU launches"))
    torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    rows = field_op_rooflines(torch.device("cuda:0"))
    for r in rows:
        print(f"{r['kernel']:44s} {r['ms']:8.4f} ms  {r['achieved']:9.1f} {r['unit']:8s} frac {r['frac']:.3f} ({r['bound']})", file=sys.stderr)
    print(json.dumps(rows))
