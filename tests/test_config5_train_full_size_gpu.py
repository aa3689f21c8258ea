"""GPU: the TRAINING half of BASELINE config 5 (and of the static field with the reference-default grid) at the size it is timed.

`bench.py`'s `dynamic.train` / `reference_default_grid.train` legs run one `RenderTrainStep` step of the reference-default models at
4096 LiDAR + 4096 camera rays x 768 samples.  Only at that size (M = 3.1 M ray-ordered rows >= 2^18, field_ops._bin_from) does the
production plan of these models switch on:

  * static hash of the space-time field, L8 F4 T2^19: EVERY level through the bins as run sums -- plan (0, 8) -- fed by the fp16
    LEVEL-MAJOR gradient `[8, M, 4]` the density tail hands over (nvsf_density_tail_grad_split);
  * flow grid, L16 F8 T2^18, scattered as its 2-feature view (flow_field.FlowGridFn) with the plan `_bin_from` gives that view;
  * the three space-time grids through the LDS kernel fed by the COLUMN-MAJOR `[24][M]` gradient (nvsf_hashgrid4d_dynamic_bwd_scalar_t);
  * K-planes texel scatter and the table scatters on the step's side stream, straight into `.grad`.

Every `-m gpu` training test of the space-time model elsewhere stays below the switch ("What's weak" 2).  Here, as
tests/test_config4_full_size_gpu.py does for config 4 (reference: nvsf/nerf/models/network_dynamic.py:213-287, trainer.py:153-219, 491-503):

  (a) each binned table gradient == the CPU oracle's scatter summed in fp64 (oracle_hashgrid_bwd_f64) of exactly the positions and
      feature gradients the step handed to its scatter, <= 2e-5 of each level's largest entry;
  (b) production plan == the reference formulations (`table_scatter="atomic"`, `hash4d_bwd="runs"`, `planes_bwd="atomic"`) on EVERY
      parameter, `planes_cl` included;
  (c) scatters beside backward (side stream) == everything on one stream, at this size, judged against each form's own run-to-run spread;
  (d) the K-planes node's texel scatter (static + three time-plane evaluations) against a deterministic fp64 sum of the same addends.
The same (a) + (b) for the static model with the L8 F4 grid (RenderRaysFn, level-major fp32 hand-over, plan (0, 8)).
"""
import os
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import golden_dynamic as GD  # noqa: E402

pytestmark = pytest.mark.gpu
N_RAYS, T = 4096, 768
M = N_RAYS * T


def _batch(S, dev):
    rng = np.random.default_rng(0)
    lo, ld = S.lidar_rays(N_RAYS, rng)
    co, cd = S.camera_rays(N_RAYS, rng)
    g = torch.Generator(device="cpu").manual_seed(3)
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    return {"rays_o_lidar": t(lo), "rays_d_lidar": t(ld), "rays_o": t(co), "rays_d": t(cd), "time": torch.tensor([[0.5]], device=dev),
            "gt_depth": torch.rand(1, N_RAYS, generator=g).to(dev) * 0.5, "gt_raydrop": (torch.rand(1, N_RAYS, generator=g) > 0.3).float().to(dev),
            "gt_intensity": torch.rand(1, N_RAYS, generator=g).to(dev), "gt_rgb": torch.rand(1, N_RAYS, 3, generator=g).to(dev)}


def _make_dynamic(dev):
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    m = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.RD)
    GD.init_by_name(m)
    assert 93e6 < sum(p.numel() for p in m.parameters()) < 94e6
    return m.to(dev)


@pytest.fixture(scope="module")
def dynamic(dev):
    """The reference's NeRFNetwork with its defaults (93.6 M parameters), parameters from name-derived seeds as in tests/test_config5_gpu.py
    (tables N(0, 0.1)-sized: densities, weights and therefore gradients are not uniformly tiny)."""
    from nvsf import synthetic as S
    return S, _make_dynamic(dev), _batch(S, dev)


@pytest.fixture(scope="module")
def static_rd(dev):
    """The static field with the reference-default grid (main_nvsf.py:45-52: 8 levels x 4 features, 512 -> 32768, T 2^19)."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(0)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES,
                          n_levels_hash=8, n_features_per_level_hash=4, base_resolution=512, max_resolution=32768, log2_hashmap_size=19)
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
            enc.params.copy_(torch.randn(enc.params.shape, generator=g) * 0.1)
    return S, m.to(dev), _batch(S, dev)


def _new_step(S, m):
    from nvsf.nerf.train_step import RenderTrainStep
    from nvsf.nerf.loss_scaler import LossScaler
    step = RenderTrainStep(m, num_steps=T, scale=S.SCALE, ema_decay=None)
    step.scaler = LossScaler(init_scale=128.0)  # where the benchmark's scaler settles: ONE step with finite fp16 gradients
    return step


def _grads(step, m, batch, seed=11):
    torch.manual_seed(seed)  # sampler jitter (perturb=True) from torch.rand: same seed, same samples
    loss, parts, _ = step.forward_backward(batch)
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def _recording(monkeypatch):
    """Records every call of field_ops.hashgrid_backward (inputs cloned on the stream the scatter is issued on) and what it returned."""
    from nvsf import field_ops as ops
    seen, real = [], ops.hashgrid_backward

    def recording(x, cols, spec, grad_out, grad_table=None, fine_from=None, merge_from=None, **kw):
        rec = {"x": x.detach().clone(), "g": grad_out.detach().clone(), "plan": fine_from, "spec": spec, "cols": tuple(cols),
               "given_table": grad_table, "stream": torch.cuda.current_stream().cuda_stream}
        out = real(x, cols, spec, grad_out, grad_table=grad_table, fine_from=fine_from, merge_from=merge_from, **kw)
        rec["out"] = out.detach().clone()
        seen.append(rec)
        return out
    monkeypatch.setattr(ops, "hashgrid_backward", recording)
    return seen


def _check_against_f64(rec, got, what):
    spec = rec["spec"]
    g = rec["g"]
    if g.dim() == 3:  # level-major [L, M, F] -> rows
        g = g.permute(1, 0, 2).reshape(g.shape[1], spec.L * spec.F)
    ref = O.hashgrid_bwd_f64(rec["x"].cpu().numpy(), rec["cols"], spec, g.float().contiguous().cpu().numpy())
    got = got.double().cpu().numpy().reshape(-1)
    assert got.shape == ref.shape and np.isfinite(got).all(), what
    worst = 0.0
    for l in range(spec.L):
        a, b = spec.offsets[l] * spec.F, spec.offsets[l + 1] * spec.F
        scale = float(np.abs(ref[a:b]).max())
        assert scale > 0.0, (what, l)
        err = float(np.abs(got[a:b] - ref[a:b]).max()) / scale
        worst = max(worst, err)
        # the bins' fixed-point image truncates at 2^-40 of a level's largest addend, fp32 atomics round per addend: both far below
        assert err <= 2e-5, (what, l, err)
    print(f"{what}: worst level error {worst:.2e} of the level's largest entry (plan {rec['plan']}, gradient {tuple(rec['g'].shape)} {rec['g'].dtype})")


def test_space_time_step_runs_the_production_plans_and_matches_the_oracle_scatter(dev, dynamic, monkeypatch):
    from nvsf import field_ops as ops
    S, m, batch = dynamic
    step = _new_step(S, m)
    seen = _recording(monkeypatch)
    loss, grads = _grads(step, m, batch)
    assert np.isfinite(loss)
    main = torch.cuda.current_stream().cuda_stream
    names = {id(p): n for n, p in m.named_parameters()}
    static_calls = [r for r in seen if r["spec"].F == 4 and r["spec"].L == 8]
    flow_calls = [r for r in seen if r["spec"].F == 2 and r["spec"].L == 16]
    assert len(static_calls) == 2 and len(flow_calls) == 2 and len(seen) == 4  # camera pass, then LiDAR pass (split_backward)
    for rec, enc in zip(static_calls, (m.hash_encoder_camera.hash_static, m.hash_encoder_lidar.hash_static)):
        spec = enc.spec
        # what the benchmark times: fp16 level-major hand-over, every level as run sums through the bins, on the side stream, into .grad
        assert tuple(rec["g"].shape) == (8, M, 4) and rec["g"].dtype == torch.float16
        assert rec["plan"] == (0, 8) == ops._bin_from(spec, M, T)
        assert rec["stream"] != main
        assert rec["given_table"].data_ptr() == enc.params.grad.data_ptr()
        _check_against_f64(rec, grads[names[id(enc.params)]], f"static hash ({names[id(enc.params)]})")
    # flow grid: both passes scatter G[row][e] on the 2-feature view of the L16 F8 T2^18 grid and expand it by the Lagrange weights
    spec8 = m.flow_net.grid_enc.spec
    assert (spec8.L, spec8.F, spec8.log2_hashmap_size) == (16, 8, 18)
    total = torch.zeros(spec8.n_rows, 4, 2, dtype=torch.float64, device=dev)
    from nvsf.nerf.models.hash_field import lagrange_weights_host
    w = torch.tensor(np.asarray(lagrange_weights_host(0.5, 4, True), np.float64), device=dev)
    for i, rec in enumerate(flow_calls):
        assert tuple(rec["g"].shape) == (M, 32) and rec["g"].dtype == torch.float32
        assert rec["plan"] == ops._bin_from(rec["spec"], M, T) and rec["plan"] is not None  # binned at this size
        assert rec["stream"] != main
        _check_against_f64(rec, rec["out"], f"flow grid, pass {i} (2-feature view)")
        total += rec["out"].double().view(-1, 1, 2) * w.view(1, 4, 1)
    got = grads["flow_net.grid_enc.params"].double().view(-1, 4, 2)
    scale = float(total.abs().max())
    assert scale > 0 and float((got - total).abs().max()) <= 1e-5 * scale  # the expansion + the sum of the two passes


def test_space_time_production_plan_equals_the_reference_formulations_on_every_parameter(dev, dynamic, variants):
    S, m, batch = dynamic
    step = _new_step(S, m)
    loss_p, prod = _grads(step, m, batch)
    variants.set(table_scatter="atomic")    # static hash + flow grid: every level through nvsf_hashgrid_bwd (fp32 atomics)
    variants.set(hash4d_bwd="runs")         # space-time grids: run-merging atomics instead of the LDS image
    variants.set(planes_bwd="atomic")       # K-planes: per-sample atomics instead of run sums
    loss_r, ref = _grads(step, m, batch)
    for k in ("table_scatter", "hash4d_bwd", "planes_bwd"):
        variants.clear(k)
    assert loss_p == loss_r  # same forward, bit for bit
    # (time slices of the space-time grids other than the two around t = 0.5 receive no gradient: same set both ways)
    untouched = {n for n, p in m.named_parameters() if p.numel() > 0 and p.requires_grad} - set(prod)
    assert set(prod) == set(ref) and all(".hash_t." in n for n in untouched) and len(untouched) == 2 * 3 * 6
    assert any("planes_cl" in n for n in prod) and any("hash_dynamic" in n for n in prod) and "flow_net.grid_enc.params" in prod
    worst = {}
    for name in sorted(prod):
        a, b = ref[name].double(), prod[name].double()
        assert torch.isfinite(b).all(), name
        scale = float(a.abs().max())
        assert scale > 0.0, name
        err = float((a - b).abs().max()) / scale
        worst[name] = err
        # same operands everywhere; fp32 atomics in another order / fixed-point bins / LDS integer sums against fp32 atomics.
        # planes_cl: the REFERENCE formulation (one fp32 atomic per sample, texel and channel) is the inaccurate side at this size --
        # a texel of a time plane receives ~10^5..10^6 same-sign addends one by one -- so it only bounds the production kernel
        # loosely here; test_planes_texel_gradient_at_full_size_against_fp64 below holds both against a deterministic fp64 sum
        assert err <= (1e-2 if name.endswith("planes_cl") else 5e-5), (name, err)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print("production plan vs reference formulations, largest relative differences:", ", ".join(f"{n} {e:.1e}" for n, e in top))


def _fp64_plane_grad(enc, xt, g_out, si, pi, pairs):
    """dL/d(plane si, pi) of Planes4D in fp64 WITHOUT atomics (the reference's formulas, planes_field.py:54-141: bilinear,
    align_corners, border clamp, product over the three planes of a group): per-sample corner contributions sorted by texel,
    segment sums as differences of a cumulative sum."""
    pts = xt.double()
    comb = pairs[pi]
    group = [pj for pj, c in enumerate(pairs) if (3 in c) == (3 in comb)]
    vals = {}
    for pj in group:
        _, _, off, C, H, W = enc._layout[si * 6 + pj]
        plane = enc.planes_cl.detach()[off:off + C * H * W].double().view(H, W, C)
        u, v = pts[:, pairs[pj][0]].clamp(0, 1) * (W - 1), pts[:, pairs[pj][1]].clamp(0, 1) * (H - 1)
        u0, v0 = u.floor().clamp(max=max(W - 2, 0)), v.floor().clamp(max=max(H - 2, 0))
        fu, fv = u - u0, v - v0
        u0, v0 = u0.long(), v0.long()
        val = (plane[v0, u0] * ((1 - fu) * (1 - fv))[:, None] + plane[v0, u0 + 1] * (fu * (1 - fv))[:, None]
               + plane[v0 + 1, u0] * ((1 - fu) * fv)[:, None] + plane[v0 + 1, u0 + 1] * (fu * fv)[:, None])
        vals[pj] = (val, u0, v0, fu, fv, H, W, C)
    others = [pj for pj in group if pj != pi]
    gval = g_out.double()[:, 8 * si: 8 * si + 8] * vals[others[0]][0] * vals[others[1]][0]
    _, u0, v0, fu, fv, H, W, C = vals[pi]
    keys = torch.cat([v0 * W + u0, v0 * W + u0 + 1, (v0 + 1) * W + u0, (v0 + 1) * W + u0 + 1])
    wts = torch.cat([(1 - fu) * (1 - fv), fu * (1 - fv), (1 - fu) * fv, fu * fv])
    contrib = gval.repeat(4, 1) * wts[:, None]
    order = torch.argsort(keys)
    keys = keys[order]
    cs = torch.cumsum(contrib[order].t().contiguous(), 1)  # [C, 4 M], scanned along the contiguous axis (the strided scan takes 3 s per plane)
    last = torch.ones_like(keys, dtype=torch.bool)
    last[:-1] = keys[1:] != keys[:-1]
    ends = cs[:, last]
    seg = ends.clone()
    seg[:, 1:] -= ends[:, :-1]
    out = torch.zeros(H * W, C, dtype=torch.float64, device=xt.device)
    out[keys[last]] = seg.t()
    return out.view(-1)


def test_planes_texel_gradient_at_full_size_against_fp64(dev, dynamic, variants):
    """The K-planes texel scatter of the step (k_planes_bwd_runs: run sums, one atomic per texel-quad change) at the timed size, on
    the LiDAR batch's own sample positions and a SMOOTH, same-sign feature gradient (what a loss gradient looks like: the worst
    case for one-by-one fp32 accumulation), against the deterministic fp64 sum.  Both formulations are reported; the production
    kernel is held to 2e-5 of each plane's largest entry."""
    import itertools
    S, m, batch = dynamic
    enc = m.planes_encoder_lidar
    pairs = list(itertools.combinations(range(4), 2))
    z = torch.linspace(float(S.MIN_NEAR), float(S.LIDAR_MAX_DEPTH), T, device=dev)
    x = batch["rays_o_lidar"][0][:, None, :] + batch["rays_d_lidar"][0][:, None, :] * z[None, :, None]
    x = ((x + S.BOUND) / (2 * S.BOUND)).clamp(0, 1).reshape(-1, 3)
    xt0 = torch.cat([x, torch.full((x.shape[0], 1), 0.5, device=dev)], -1).contiguous()
    gen = torch.Generator().manual_seed(1)
    base = (0.5 + x[:, :1]) * 1e-3                                                        # smooth in space, one sign
    ws = (base * (1.0 + 0.05 * torch.randn(x.shape[0], 32, generator=gen).to(dev))).contiguous()
    wd = (base * (1.0 + 0.05 * torch.randn(x.shape[0], 32, generator=gen).to(dev))).contiguous()
    got = {}
    for variant in ("runs", "atomic"):
        variants.set(planes_bwd=variant)
        enc.planes_cl.grad = None
        xt = xt0.clone().requires_grad_()
        s_, d_ = enc(xt)
        ((s_ * ws).sum() + (d_ * wd).sum()).backward()
        torch.cuda.synchronize()
        got[variant] = enc.planes_cl.grad.detach().clone()
    variants.clear("planes_bwd")
    enc.planes_cl.grad = None
    worst = {"runs": 0.0, "atomic": 0.0}
    for si in range(len(enc.multiscale_res)):
        for pi in range(6):
            _, _, off, C, H, W = enc._layout[si * 6 + pi]
            ref = _fp64_plane_grad(enc, xt0, wd if 3 in pairs[pi] else ws, si, pi, pairs)
            scale = float(ref.abs().max())
            assert scale > 0
            for variant in worst:
                worst[variant] = max(worst[variant], float((got[variant][off:off + C * H * W].double() - ref).abs().max()) / scale)
    print(f"K-planes texel gradient vs deterministic fp64 at M = {xt0.shape[0]}: run sums {worst['runs']:.2e}, per-sample atomics {worst['atomic']:.2e} "
          "of a plane's largest entry")
    assert worst["runs"] <= 2e-5
    assert worst["atomic"] <= 2e-2  # the reference formulation's own accuracy at this size (informational bound)


def test_planes_multi_texel_gradient_at_full_size_against_fp64(dev, dynamic, variants):
    """The texel scatter of the STEP's K-planes node at the timed size: nvsf_planes_multi_bwd with the static evaluation and the three
    time-plane evaluations of a density query (x, x + flow to the next frame, x + flow to the previous one) sharing one gradient slice
    x 0.5 / 0.25 / 0.25, on the LiDAR batch's sample positions and a smooth same-sign gradient, against the deterministic fp64 sum of the
    same addends.  Production (time planes through the fp64 LDS images) is held to 2e-5 of each plane's largest
    entry on every plane and to 1e-5 on the time planes (measured 2e-6: ~10^3 slice sums per texel leave the images as fp32 atomics); the round-5 form (run sums into memory-side fp32 atomics, planes_bwd="global")
    is reported beside it: a time-plane texel receives 10^5 - 10^6 run sums there and loses them to fp32 rounding in arrival order."""
    import itertools
    from planes_calls import multi_bwd_call
    S, m, batch = dynamic
    enc = m.planes_encoder_lidar
    pairs = list(itertools.combinations(range(4), 2))
    z = torch.linspace(float(S.MIN_NEAR), float(S.LIDAR_MAX_DEPTH), T, device=dev)
    x = batch["rays_o_lidar"][0][:, None, :] + batch["rays_d_lidar"][0][:, None, :] * z[None, :, None]
    x = ((x + S.BOUND) / (2 * S.BOUND)).clamp(0, 1).reshape(-1, 3).contiguous()
    Mx = x.shape[0]
    flow = (2e-3 * torch.sin(40.0 * torch.cat([x, x.flip(-1)], -1))).contiguous()  # smooth, a fraction of a texel to several texels at the finest scale
    gen = torch.Generator().manual_seed(1)
    g = ((0.5 + x[:, :1]) * 1e-3 * (1.0 + 0.05 * torch.randn(Mx, 120, generator=gen).to(dev))).contiguous()
    times = [float(np.float32(v)) for v in (0.5, 0.5, 0.5 + 1 / 64, 0.5 - 1 / 64)]
    got = {}
    for variant in ("runs", "global"):
        variants.set(planes_bwd=variant)
        got[variant] = multi_bwd_call(enc, x, flow, g, times, dev)
    variants.clear("planes_bwd")
    col = lambda v: torch.full((Mx, 1), v, dtype=torch.float32, device=dev)
    positions = [torch.cat([x, col(times[1])], -1), torch.cat([x + flow[:, 0:3], col(times[2])], -1), torch.cat([x + flow[:, 3:6], col(times[3])], -1)]
    worst = {"runs": [0.0, 0.0], "global": [0.0, 0.0]}  # [spatial planes, time planes]
    for si in range(len(enc.multiscale_res)):
        for pi in range(6):
            _, _, off, C, H, W = enc._layout[si * 6 + pi]
            if 3 in pairs[pi]:
                ref = sum(w * _fp64_plane_grad(enc, xt, g[:, 32:64], si, pi, pairs) for w, xt in zip((0.5, 0.25, 0.25), positions))
            else:
                ref = _fp64_plane_grad(enc, positions[0], g[:, 0:32], si, pi, pairs)
            scale = float(ref.abs().max())
            assert scale > 0
            for variant in worst:
                err = float((got[variant][off:off + C * H * W].double() - ref).abs().max()) / scale
                worst[variant][int(3 in pairs[pi])] = max(worst[variant][int(3 in pairs[pi])], err)
    print(f"K-planes multi-evaluation texel gradient vs fp64 at M = {Mx}: production spatial {worst['runs'][0]:.2e} / time {worst['runs'][1]:.2e}; "
          f"global atomics only: spatial {worst['global'][0]:.2e} / time {worst['global'][1]:.2e} of a plane's largest entry")
    assert worst["runs"][0] <= 2e-5 and worst["runs"][1] <= 1e-5
    assert worst["global"][0] <= 2e-5 and worst["global"][1] <= 2e-3  # (informational: the form the LDS image replaced)


def test_space_time_step_planes_gradient_against_fp64_of_its_own_addends(dev, dynamic, variants, monkeypatch):
    """the K-planes texel gradient the STEP produces (not a synthetic gradient) against a deterministic fp64 sum of
    the step's own addends.  The operands of the LiDAR pass' PlanesMultiFn.backward -- positions, flow offsets, the two gradient
    slices of the density tail's input gradient, evaluation times -- are recorded as the step hands them over; the reference is
    _fp64_plane_grad over the static evaluation and the three time-plane evaluations x 0.5 / 0.25 / 0.25 (network_dynamic.py:273).
    Production (time planes through the fp64 LDS images) is held to 2e-5 of each plane's largest entry (1e-5 on the time
    planes); the round-5 form (planes_bwd="global": run sums into memory-side fp32 atomics) is measured beside it -- it is the side
    whose time-plane texels lose 10^5 - 10^6 addends' low bits in arrival order (1e-4 between two runs of one step in GPUTEST_r05)."""
    import itertools
    from nvsf import field_ops as ops
    S, m, batch = dynamic
    step = _new_step(S, m)
    enc = m.planes_encoder_lidar
    seen, real = [], ops.PlanesMultiFn.backward

    def recording(ctx, *grads):
        if ctx.planes_param is enc.planes_cl:
            saved = ctx.saved_tensors
            seen.append({"x": saved[0].detach().clone(), "fl": saved[2].detach().clone() if len(saved) > 2 else None, "meta": list(ctx.evals_meta),
                         "blend": ctx.blend, "g": [g.detach().float().clone() for g in grads[:2]]})
        return real(ctx, *grads)
    monkeypatch.setattr(ops.PlanesMultiFn, "backward", staticmethod(recording))
    got = {}
    for variant in ("runs", "global"):
        variants.set(planes_bwd=variant)
        got[variant] = _grads(step, m, batch)[1]["planes_encoder_lidar.planes_cl"]
    variants.clear("planes_bwd")
    assert len(seen) == 2 and all(r["blend"] and [mt[0] for mt in r["meta"]] == [0, 1, 1, 1] for r in seen)
    # (the two runs' operands agree only up to an fp16 rounding step here and there: the chamfer term's backward adds its point gradients
    # with atomics, as the reference's does (chamfer3D.cu:133-157), and a last-bit difference of dL/d(range) flips roundings of the fp16
    # hand-overs on its way back through the MLPs -- measured 2.7e-4 of the largest entry -- so each run is held against the fp64 sum
    # of the operands IT recorded)
    assert torch.equal(seen[0]["x"], seen[1]["x"]) and float((seen[0]["g"][1] - seen[1]["g"][1]).abs().max()) <= 2e-3 * float(seen[0]["g"][1].abs().max())
    pairs = list(itertools.combinations(range(4), 2))
    Mx = seen[0]["x"].shape[0]
    assert Mx == M
    col = lambda v: torch.full((Mx, 1), v, dtype=torch.float32, device=dev)
    worst = {"runs": [0.0, 0.0], "global": [0.0, 0.0]}
    for variant, rec in zip(("runs", "global"), seen):
        positions = []
        for grp, off_col, t_e, has_off in rec["meta"]:
            xe = rec["x"][:, :3] + rec["fl"][:, off_col:off_col + 3] if has_off else rec["x"][:, :3]  # fp32 add, as the kernels form x + flow
            positions.append(torch.cat([xe, col(t_e)], -1))
        g_s, g_d = rec["g"]
        for si in range(len(enc.multiscale_res)):
            for pi in range(6):
                _, _, off, C, H, W = enc._layout[si * 6 + pi]
                if 3 in pairs[pi]:
                    ref = sum(w * _fp64_plane_grad(enc, xt, g_d, si, pi, pairs) for w, xt in zip((0.5, 0.25, 0.25), positions[1:]))
                else:
                    ref = _fp64_plane_grad(enc, positions[0], g_s, si, pi, pairs)
                scale = float(ref.abs().max())
                assert scale > 0
                err = float((got[variant][off:off + C * H * W].double() - ref).abs().max()) / scale
                worst[variant][int(3 in pairs[pi])] = max(worst[variant][int(3 in pairs[pi])], err)
    print(f"K-planes texel gradient of the STEP vs fp64 of its own addends (LiDAR pass, M = {Mx}): production spatial {worst['runs'][0]:.2e} / time "
          f"{worst['runs'][1]:.2e}; global atomics only: spatial {worst['global'][0]:.2e} / time {worst['global'][1]:.2e} of a plane's largest entry")
    assert worst["runs"][0] <= 2e-5 and worst["runs"][1] <= 1e-5
    assert worst["global"][0] <= 2e-5 and worst["global"][1] <= 1e-2  # (informational: the form the LDS image replaced)


def _gap(a, b):
    """{parameter: max |a - b| / max |a|}"""
    out = {}
    for name in a:
        x, y = a[name].double(), b[name].double()
        scale = float(x.abs().max())
        assert scale > 0.0, name
        out[name] = float((x - y).abs().max()) / scale
    return out


def test_space_time_scatters_beside_backward_equal_one_stream(dev, dynamic):
    """Scatters on the side stream == everything on one stream, decided against the run-to-run spread of each configuration itself
    (ADVICE r5): the two forms launch the same kernels on the same operands, so they may differ only by what two runs
    of ONE form differ by -- the arrival order of fp32 atomics.  Each form runs twice; a parameter's cross-form gap must stay within
    8 x the larger within-form gap (or a floor of 1e-5 / 2e-6 for parameters whose sums happen to repeat bit for bit), and every gap within an absolute
    bound derived from the addend counts: an entry is an fp32 sum of n addends in arrival order, each addition rounding the running sum
    by <= eps / 2 = 6e-8 of it -- n eps / 2 worst case, ~sqrt(n) eps typical, of the sum of magnitudes, which cancellation makes a
    multiple of the entry itself.  Tables and weights: n up to ~10^4 bin / workgroup sums => 1e-4 (measured: 1.5e-5 on the static
    hash, 7e-6 on the space-time grids, r06_c1).  K-planes: ~130 slice sums of the LDS image per time texel, <= 10^3 run sums per
    spatial texel => 2e-5 (measured <= 5e-6).  A race (a gradient buffer rewritten on the main stream while the side stream still reads
    it) shows as a cross-form gap far outside the within-form spread; round 5's single-sample 1e-4 bound on `planes_cl` could not tell."""
    S, m, batch = dynamic
    step = _new_step(S, m)
    runs = {}
    for overlap in (True, False, True, False):
        step.scatter_overlap = overlap
        runs.setdefault(overlap, []).append(_grads(step, m, batch)[1])
    step.scatter_overlap = True
    side, one = runs[True], runs[False]
    assert set(side[0]) == set(one[0])
    within = {n: max(v, _gap(one[0], one[1])[n]) for n, v in _gap(side[0], side[1]).items()}
    cross = {n: max(_gap(one[i], side[j])[n] for i in range(2) for j in range(2)) for n in within}
    top = sorted(cross, key=lambda n: -cross[n])[:4]
    print("side stream vs one stream, largest gaps (cross-form / within-form):", ", ".join(f"{n} {cross[n]:.1e} / {within[n]:.1e}" for n in top))
    for name in sorted(cross):
        bound = 2e-5 if name.endswith("planes_cl") else 1e-4
        assert within[name] <= bound, (name, within[name])
        # (a lost or doubled addend -- what a race produces -- is 1e-2 ... 1 of an entry; the factor only has to cover that a maximum over four
        # cross pairs exceeds a maximum over two within pairs, the floor that a form may repeat itself bit for bit on a quiet parameter)
        assert cross[name] <= max(8.0 * within[name], 2e-6 if name.endswith("planes_cl") else 1e-5), (name, cross[name], within[name])
        assert cross[name] <= bound, (name, cross[name])
    print("planes_cl gaps (cross / within):", ", ".join(f"{n} {cross[n]:.1e} / {within[n]:.1e}" for n in sorted(cross) if n.endswith("planes_cl")))


def test_space_time_full_step_updates_every_parameter_and_stays_finite(dev, dynamic):
    """Whole steps (forward, backward, loss scaling, Adam with the last pass' parameters updated behind their scatter) at the timed size."""
    S, _, batch = dynamic
    m = _make_dynamic(dev)  # a model of its own: the steps below move the parameters
    step = _new_step(S, m)
    before = {n: p.detach().clone() for n, p in m.named_parameters() if p.numel()}
    losses = []
    for i in range(3):
        torch.manual_seed(20 + i)
        losses.append(float(step.step(batch)[0]))
    step.sync()
    torch.cuda.synchronize()
    assert all(np.isfinite(losses))
    moved = 0
    for n, p in m.named_parameters():
        if p.numel() and p.requires_grad:
            assert bool(torch.isfinite(p).all()), n
            if p.grad is None:  # the time slices of the space-time grids away from t = 0.5 are not touched by this batch
                assert ".hash_t." in n and torch.equal(p.detach(), before[n]), n
            else:
                assert not torch.equal(p.detach(), before[n]), n
                moved += 1
    assert moved >= 20
    del m, step
    torch.cuda.empty_cache()


def test_static_rd_step_runs_the_production_plan_and_matches_the_oracle_scatter(dev, static_rd, monkeypatch):
    from nvsf import field_ops as ops
    S, m, batch = static_rd
    step = _new_step(S, m)
    seen = _recording(monkeypatch)
    loss, grads = _grads(step, m, batch)
    assert np.isfinite(loss) and len(seen) == 2
    main = torch.cuda.current_stream().cuda_stream
    names = {id(p): n for n, p in m.named_parameters()}
    for rec, enc in zip(seen, (m.hash_encoder_camera, m.hash_encoder_lidar)):
        spec = enc.spec
        assert (spec.L, spec.F) == (8, 4)
        assert tuple(rec["g"].shape) == (8, M, 4) and rec["g"].dtype == torch.float32  # level-major fp32 from the density MLP's backward
        assert rec["plan"] == (0, 8) == ops._bin_from(spec, M, T)
        assert rec["stream"] != main and rec["given_table"].data_ptr() == enc.params.grad.data_ptr()
        _check_against_f64(rec, grads[names[id(enc.params)]], f"static L8 F4 ({names[id(enc.params)]})")


def test_static_rd_binned_plan_equals_the_atomic_variant_on_every_parameter(dev, static_rd, variants):
    S, m, batch = static_rd
    step = _new_step(S, m)
    loss_b, binned = _grads(step, m, batch)
    variants.set(table_scatter="atomic")
    loss_a, atomic = _grads(step, m, batch)
    variants.clear("table_scatter")
    assert loss_a == loss_b
    assert set(binned) == set(atomic) == {n for n, p in m.named_parameters() if p.numel() > 0}
    for name in binned:
        a, b = atomic[name].double(), binned[name].double()
        assert torch.isfinite(b).all(), name
        scale = float(a.abs().max())
        assert scale > 0.0, name
        assert float((a - b).abs().max()) / scale <= 5e-5, name
