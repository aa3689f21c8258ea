"""GPU parity of the space-time rows (a13 HashGridT / HashGrid4D, a14 Planes4D, a15 FlowField, a18 NeRFNetwork) against
fixtures produced by running the REFERENCE's own model files on CPU (tests/golden/make_golden.py + golden_dynamic.py;
`tinycudann` inside the reference = the CPU specification of those operators).  Parameters are rebuilt from
name-derived seeds (golden_dynamic.init_by_name), not stored.

Tolerances: K-planes fp32 1e-5 (same formulas, different summation granularity inside grid_sample); hash features
that end in fp16 arithmetic 2e-3 relative (1 fp16 ulp); network outputs 1e-4 abs on composited images / depth.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import golden_dynamic as GD  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(HERE, "golden")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_planes4d_forward_backward(dev):
    from nvsf.nerf.models.planes_field import Planes4D
    g = np.load(os.path.join(GOLD, "planes4d.npz"))
    enc = Planes4D(resolution=[8, 8, 8, 5], multiscale_res=[1, 2, 4, 8])
    GD.init_by_name(enc)
    enc = enc.to(dev)
    assert enc.n_output_dims == 64 and tuple(enc.plane(1, 2).shape) == (1, 8, 5, 16)
    assert [n for n, _ in enc.named_parameters()] == ["planes_cl"] and len(enc.state_dict()) == 24  # one parameter, the reference's schema
    xt = _t(g["xt"], dev).requires_grad_()
    s, d = enc(xt)
    np.testing.assert_allclose(s.detach().cpu().numpy(), g["static"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(d.detach().cpu().numpy(), g["dynamic"], atol=1e-5, rtol=1e-5)
    assert torch.equal(enc.forward_static(xt.detach()), s.detach()) and torch.equal(enc.forward_dynamic(xt.detach()), d.detach())
    ((s * _t(g["grad_static"], dev)).sum() + (d * _t(g["grad_dynamic"], dev)).sum()).backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["grad_xt"], atol=2e-3, rtol=2e-3)
    for (si, pi), key in (((0, 0), "grad_plane_0_0"), ((3, 5), "grad_plane_3_5"), ((2, 3), "grad_plane_2_3")):
        np.testing.assert_allclose(enc.plane_grad(si, pi).cpu().numpy(), g[key], atol=2e-4, rtol=2e-4)
    # the kernels read the parameter itself: an in-place update is seen by the next call
    with torch.no_grad():
        enc.plane(0, 0).mul_(2.0)
    s2 = enc.forward_static(xt.detach()).detach()
    np.testing.assert_allclose(s2[:, :8].cpu().numpy(), 2 * g["static"][:, :8], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("dynamic_only", [False, True])
def test_planes_backward_run_merging_equals_per_sample_atomics(dev, dynamic_only, variants):
    """The run-merging plane-gradient kernel (default) against the one-atomic-per-(sample, texel, channel) kernel -- itself
    pinned by the reference's autograd above -- on ray-ordered rows (long runs inside one texel quad at the coarse scales,
    a constant time coordinate as in a training step), both `want` forms, M not a multiple of the chunk length."""
    from nvsf.nerf.models.planes_field import Planes4D
    rng = np.random.default_rng(3)
    n_rays, T = 29, 211
    o = rng.random((n_rays, 1, 3)) * 0.4 + 0.3
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = np.clip(o + d * np.linspace(0, 0.3, T).reshape(1, T, 1), 0, 1).reshape(-1, 3)
    xt_np = np.concatenate([x, np.full((x.shape[0], 1), 0.37)], -1).astype(np.float32)
    grads = {}
    for variant in ("runs", "atomic"):
        if variant == "atomic":
            variants.set(planes_bwd="atomic")
        torch.manual_seed(0)
        enc = Planes4D(resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8]).to(dev)
        xt = _t(xt_np, dev).requires_grad_()
        g = torch.Generator().manual_seed(1)
        if dynamic_only:
            out = enc.forward_dynamic(xt)
            w = torch.randn(out.shape, generator=g).to(dev)
            w[::7] = 0.0
            (out * w).sum().backward()
        else:
            s_, d_ = enc(xt)
            ws, wd = torch.randn(s_.shape, generator=g).to(dev), torch.randn(d_.shape, generator=g).to(dev)
            ws[::5] = 0.0
            ((s_ * ws).sum() + (d_ * wd).sum()).backward()
        grads[variant] = [xt.grad.clone()] + [enc.plane_grad(si, pi).clone() for si, pi, *_ in enc._layout]
    assert torch.equal(grads["runs"][0], grads["atomic"][0])  # coordinate gradients come from the same kernel
    n_checked = 0
    for a, b in zip(grads["runs"][1:], grads["atomic"][1:]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()) + 1e-7)
        n_checked += 1
    assert n_checked == 24  # with want = dynamic the static planes receive zero gradients (one flat parameter)


def test_hashgrid4d_and_flow(dev):
    from nvsf.nerf.models.hash_field import HashGrid4D
    from nvsf.nerf.models.flow_field import FlowField
    g = np.load(os.path.join(GOLD, "hash4d_flow.npz"))
    enc = HashGrid4D(base_resolution=16, max_resolution=256, time_resolution=4, n_levels=8, n_features_per_level=4, log2_hashmap_size=12)
    GD.init_by_name(enc)
    enc = enc.to(dev)
    assert enc.n_output_dims == 56 and "hash_dynamic.1.hash_t.3.params" in enc.state_dict()
    x = _t(g["x"], dev)
    torch.set_grad_enabled(False)
    for i in range(5):
        tv = float(g[f"t{i}"])
        s, d = enc(x, torch.tensor([[tv]], dtype=torch.float32, device=dev))
        assert np.array_equal(s.cpu().numpy().view(np.uint16), g[f"static{i}"].view(np.uint16))  # 3-D grid: bit-exact fp16
        assert d.dtype == torch.float32
        np.testing.assert_allclose(d.cpu().numpy(), g[f"dyn_t11_{i}"], atol=1e-6, rtol=1e-5)
        d0 = enc.forward_dynamic(x, torch.tensor(tv))
        assert d0.dtype == torch.float16  # 0-dim t keeps the blend in fp16, as in the reference
        ref0 = g[f"dyn_t0_{i}"].astype(np.float32)
        np.testing.assert_allclose(d0.float().cpu().numpy(), ref0, atol=2e-3 * np.abs(ref0).max(), rtol=0)
    flow = FlowField(n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=8192, log2_hashmap_size=14)
    GD.init_by_name(flow)
    flow = flow.to(dev)
    out = flow(_t(g["flow_xt"], dev))
    torch.set_grad_enabled(True)
    np.testing.assert_allclose(out.cpu().numpy(), g["flow"], atol=1e-5, rtol=1e-4)
    # opt-in fp16 MFMA form of the flow MLP (what autocast computes in the reference's Trainer): fp16-level agreement
    with torch.no_grad():
        out16 = flow(_t(g["flow_xt"], dev), fp16=True)  # the reference's --fp16 regime, per call
        with torch.autocast("cuda", dtype=torch.float16):  # ... or through the Trainer's autocast region (trainer.py:1318, 1491)
            out_ac = flow(_t(g["flow_xt"], dev))
        out_off = flow(_t(g["flow_xt"], dev), fp16=False)
    assert out16.shape == out.shape and torch.equal(out16, out_ac) and torch.equal(out_off, out)
    assert not torch.equal(out16, out)
    np.testing.assert_allclose(out16.cpu().numpy(), g["flow"], atol=1e-4, rtol=1e-2)


def test_constructor_options_outside_the_reference_configuration(dev):
    """Options of the reference's encoders that no configuration of the reference switches on: HashGrid4D's 'prod' / 'sum' / 'mean'
    reductions of the three pair features (hash_field.py:16-27, 157-158; both dtype regimes of the time blend, and with autograd
    recording) and FlowField's frequency embedding instead of the grid (flow_field.py:17-38, 63-66), against outputs of the
    reference's own modules; use_freq together with use_grid fails in the reference itself and is rejected."""
    from nvsf.nerf.models.hash_field import HashGrid4D
    from nvsf.nerf.models.flow_field import FlowField
    g = np.load(os.path.join(GOLD, "hash4d_flow.npz"))
    x = _t(g["x"], dev)
    for red in ("prod", "sum", "mean"):
        enc = HashGrid4D(base_resolution=16, max_resolution=256, time_resolution=4, n_levels=8, n_features_per_level=4,
                         log2_hashmap_size=12, reduction=red)
        GD.init_by_name(enc)
        enc = enc.to(dev)
        assert enc.n_output_dims == 40
        ref = g[f"dyn_t11_{red}"]
        tol = dict(atol=1e-6 + 1e-5 * float(np.abs(ref).max()), rtol=1e-5)
        with torch.no_grad():
            s, d = enc(x, torch.tensor([[0.77]], dtype=torch.float32, device=dev))
            d0 = enc.forward_dynamic(x, torch.tensor(0.77))
        assert d.dtype == torch.float32 and d.shape == (600, 8) and d0.dtype == torch.float16
        np.testing.assert_allclose(d.cpu().numpy(), ref, **tol)
        ref0 = g[f"dyn_t0_{red}"].astype(np.float32)
        np.testing.assert_allclose(d0.float().cpu().numpy(), ref0, atol=4e-3 * np.abs(ref0).max(), rtol=0)
        d_train = enc.forward_dynamic(x, torch.tensor([[0.77]], dtype=torch.float32, device=dev))  # autograd recording: HashDynFn + fold
        np.testing.assert_allclose(d_train.detach().cpu().numpy(), ref, **tol)
        d_train.sum().backward()
        assert any(p.grad is not None and bool(p.grad.abs().sum() > 0) for p in enc.hash_dynamic.parameters())
    with pytest.raises(ValueError):
        HashGrid4D(reduction="max")
    flow = FlowField(n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=8192, log2_hashmap_size=14, use_freq=True,
                     use_grid=False)
    GD.init_by_name(flow)
    flow = flow.to(dev)
    assert flow.input_dim == 48 and not hasattr(flow, "grid_enc")
    xt = _t(g["flow_xt"], dev)
    with torch.no_grad():
        out, out16 = flow(xt), flow(xt, fp16=True)
    np.testing.assert_allclose(out.cpu().numpy(), g["flow_freq_only"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out16.cpu().numpy(), g["flow_freq_only"], atol=5e-3, rtol=2e-2)  # fp16 operands on flows of ~0.8
    flow(xt).sum().backward()  # autograd recording: plain Linear layers
    assert flow.mlp[0].weight.grad is not None
    with pytest.raises(NotImplementedError):
        FlowField(use_freq=True, use_grid=True)


@pytest.mark.parametrize("t_val", [0.37, 0.0, 1.0])  # between two slices / exactly on the first / on the last slice
def test_hashgrid4d_fused_training_path_equals_the_per_slice_path(dev, t_val, variants):
    """HashDynFn (fused forward + fused table-gradient kernel) against the per-slice operator path (six tcnn.Encoding calls,
    blend, Lagrange reduction through autograd): same features, same gradients on the same slice parameters, none elsewhere."""
    from nvsf.nerf.models.hash_field import HashGrid4D
    rng = np.random.default_rng(5)
    n_rays, T = 23, 150
    o = rng.random((n_rays, 1, 3)) * 0.4 + 0.3
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = _t(np.clip(o + d * np.linspace(0, 0.3, T).reshape(1, T, 1), 0, 1).reshape(-1, 3).astype(np.float32), dev)
    t = torch.tensor([[t_val]], dtype=torch.float32, device=dev)
    res = {}
    for mode in ("fused", "ops"):
        variants.set(hash4d_train={"fused": "fused", "ops": "slices"}[mode])
        torch.manual_seed(0)
        enc = HashGrid4D(base_resolution=16, max_resolution=512, time_resolution=4, n_levels=8, n_features_per_level=4, log2_hashmap_size=12,
                         hash_size_dynamic=[11, 10, 10]).to(dev)
        with torch.no_grad():
            for p in enc.parameters():
                p.uniform_(-0.5, 0.5)
        out = enc.forward_dynamic(x, t)
        g = torch.Generator().manual_seed(1)
        w = torch.randn(out.shape, generator=g).to(dev)
        w[::6] = 0.0
        (out.float() * w).sum().backward()
        res[mode] = (out.detach().float().clone(), {n: (None if p.grad is None else p.grad.clone()) for n, p in enc.named_parameters()})
    torch.testing.assert_close(res["fused"][0], res["ops"][0], rtol=1e-5, atol=1e-6)
    n_grad = 0
    for name, gr in res["ops"][1].items():
        gf = res["fused"][1][name]
        assert (gr is None or not gr.any()) == (gf is None or not gf.any()), name
        if gr is not None and gr.any():
            torch.testing.assert_close(gf, gr, rtol=2e-3, atol=2e-4 * float(gr.abs().max()))
            n_grad += 1
    assert n_grad == (3 if t_val in (0.0, 1.0) else 6)


@pytest.mark.parametrize("flow_scale,t_val", [(1e-3, 0.5), (0.05, 0.5), (0.05, 0.0), (0.05, 1.0)])
def test_fused_dynamic_features_equal_the_separate_launches(dev, flow_scale, t_val, variants):
    """The no-grad density query with in-kernel `x + flow` for the neighbour hash grids and in-place neighbour coordinates,
    against the launch-by-launch form -- bit for bit, with small and with large flows, first / last frame included."""
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    torch.manual_seed(4)
    m = NeRFNetwork(time_resolution=4, num_frames=16, bound=2.0, min_resolution=16, base_resolution=32, max_resolution=512,
                    log2_hashmap_size=13).to(dev).eval()
    with torch.no_grad():
        m.flow_net.mlp[-1].weight.normal_(0, flow_scale * 30.0)
        for p in m.flow_net.grid_enc.parameters():
            p.uniform_(-0.5, 0.5)
    x = (torch.rand(20000, 3, device=dev) * 2 - 1) * 1.9
    t = torch.tensor([[t_val]], device=dev)
    outs = {}
    for mode in ("1", "0"):
        variants.set(dynamic_fused=mode == "1")
        with torch.no_grad():
            feats = m._dynamic_features(m._unit_cube(x), t, True)
            dens = m.density(x, t, cal_lidar_color=True)
        # the fused form takes the static hash features level-major, fp16 [8, M, 4] (nvsf_hashgrid_fwd_level_major): rows for the comparison
        feats = [v.permute(1, 0, 2).reshape(v.shape[1], -1) if v.dim() == 3 else v for v in feats]
        if mode == "1":
            with torch.no_grad():
                assert m.hash_encoder_lidar.forward_static(m._unit_cube(x), level_major=True).shape == (8, x.shape[0], 4)
        f = [v.float() for v in feats]
        # the fused form blends the K-planes neighbours inside its kernel (plane_d == plane_1 == plane_2 == the blend) and hands the
        # plane features over as fp16 rows: compare the blend network_dynamic.py:273 forms, rounded to fp16 as the density MLP
        # takes it either way
        outs[mode] = [f[0].half(), (0.5 * f[1] + 0.25 * (f[2] + f[3])).half()] + f[4:] + [dens["sigma"], dens["geo_feat"].float()]
    with torch.no_grad():
        assert float(m.flow_net(torch.cat([m._unit_cube(x), t.expand(x.shape[0], 1)], -1)).abs().mean()) > 0.1 * flow_scale
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("logit_grad", ["composed", "matrix"])  # formed inside the MLP backward (production) / by nvsf_sigma_geo_bwd
@pytest.mark.parametrize("t_val", [0.4, 0.0])  # interior frame (two neighbours) / first frame (one neighbour aliases the current frame)
def test_density_tail_training_path_equals_the_operator_path(dev, t_val, logit_grad, variants):
    """DensityTailFn (blend + concatenation + density MLP in one forward launch, fused MLP backward, gradient handed back per
    input with the blend factors) against the operator path (torch blends, torch.cat, tcnn.Network autograd): density outputs
    and every parameter gradient of a density query."""
    variants.set(density_grad=logit_grad)
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    x = (torch.rand(6000, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev) * 1.9
    t = torch.tensor([[t_val]], device=dev)
    res = {}
    for mode in ("fused", "ops"):
        variants.set(density_tail_train={"fused": "fused", "ops": "chain"}[mode])
        torch.manual_seed(4)
        m = NeRFNetwork(time_resolution=4, num_frames=16, bound=2.0, min_resolution=16, base_resolution=32, max_resolution=512,
                        log2_hashmap_size=13).to(dev)
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if "hash" in n_ and p.numel() > 1000:
                    p.uniform_(-0.5, 0.5)
        d = m.density(x, t, cal_lidar_color=True)
        g = torch.Generator().manual_seed(2)
        ws, wg = torch.randn(d["sigma"].shape, generator=g).to(dev), torch.randn(d["geo_feat"].shape, generator=g).to(dev)
        ((d["sigma"] * ws).sum() + (d["geo_feat"].float() * wg).sum()).backward()
        res[mode] = (d["sigma"].detach().clone(), d["geo_feat"].detach().float().clone(),
                     {n_: p.grad.clone() for n_, p in m.named_parameters() if p.grad is not None})
    torch.testing.assert_close(res["fused"][0], res["ops"][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res["fused"][1], res["ops"][1], rtol=1e-5, atol=1e-6)
    assert set(res["fused"][2]) == set(res["ops"][2])
    n = 0
    for name, gr in res["ops"][2].items():
        gf = res["fused"][2][name]
        scale = float(gr.abs().max())
        if scale == 0.0:
            assert not gf.any(), name
            continue
        # the hash features' gradient passes through fp16 in the operator path and stays fp32 in the fused one
        assert float((gf - gr).abs().max()) <= 2e-2 * scale and float((gf - gr).abs().mean()) <= 2e-3 * scale, name
        n += 1
    assert n >= 6  # each K-planes field is one flat parameter


def test_flow_grid_training_path_equals_the_operator_path(dev):
    """FlowGridFn (fused grid + Lagrange forward kernel, table gradient straight from dL/d(reduced)) against encoder ->
    .float() -> lagrange_reduce through autograd: flows and the gradient of the grid table and of the Linear layers."""
    from nvsf.nerf.models.flow_field import FlowField
    xt = torch.cat([torch.rand(7000, 3, generator=torch.Generator().manual_seed(3)), torch.full((7000, 1), 0.43)], -1).to(dev)
    res = {}
    for mode in ("fused", "ops"):
        torch.manual_seed(6)
        flow = FlowField(n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=1024, log2_hashmap_size=13).to(dev)
        flow.grid_train_mode = mode  # test-only switch: "ops" = encoder -> .float() -> lagrange_reduce under plain autograd
        with torch.no_grad():
            flow.grid_enc.params.uniform_(-0.5, 0.5)
            flow.mlp[-1].weight.normal_(0, 0.2)
        out = flow(xt)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).to(dev) * 64.0  # what a loss scaler would supply
        (out * w).sum().backward()
        res[mode] = (out.detach().clone(), {n: p.grad.clone() for n, p in flow.named_parameters()})
    torch.testing.assert_close(res["fused"][0], res["ops"][0], rtol=1e-5, atol=1e-7)
    for name, gr in res["ops"][1].items():
        gf = res["fused"][1][name]
        scale = float(gr.abs().max())
        # the operator path rounds dL/d(features) to fp16 on its way into the encoder, the fused path keeps fp32
        assert float((gf - gr).abs().max()) <= 5e-3 * scale, name


def test_flow_mlp_fused_training_path_against_fp32_autograd(dev):
    """flow_field.FlowMlpFn (the Linear layers on the fused MFMA forward / backward kernels, used by the loss-scaled training
    step) against torch's fp32 Linear stack: values and all gradients at fp16 accuracy."""
    from nvsf.nerf.models.flow_field import FlowField, FlowMlpFn
    import torch.nn as nn
    torch.manual_seed(2)
    flow = FlowField(n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=512, log2_hashmap_size=12).to(dev)
    with torch.no_grad():
        flow.mlp[-1].weight.normal_(0, 0.3)  # the reference's 1e-3 init makes every gradient tiny: use a visible scale
    lin = [m for m in flow.mlp if isinstance(m, nn.Linear)]
    x = (torch.randn(5000, 32, device=dev) * 0.5).requires_grad_()
    g = torch.randn(5000, 6, device=dev)
    out = FlowMlpFn.apply(x, lin[0].weight, lin[1].weight, lin[2].weight)
    out.backward(g)
    got = [out.detach().clone(), x.grad.clone()] + [l.weight.grad.clone() for l in lin]
    x.grad = None
    for l in lin:
        l.weight.grad = None
    ref_out = flow.mlp(x)
    ref_out.backward(g)
    ref = [ref_out.detach(), x.grad] + [l.weight.grad for l in lin]
    # fp16 activations: ~1e-3 relative per element; the ReLU gate of a unit near zero may flip, which moves single entries
    # of the input gradient by a whole weight -- hence the fraction / mean criteria (as in tests/test_mlp_bwd_gpu.py)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        scale = float(b.abs().max())
        err = (a - b).abs()
        assert float((err <= 2e-2 * scale).float().mean()) > 0.995 and float(err.mean()) < 5e-3 * scale, (float(err.max()), scale)


@pytest.fixture(scope="module")
def net(dev):
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    from nvsf import synthetic as S
    m = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL).eval()
    GD.init_by_name(m)
    return m.to(dev)


def test_network_density_and_flow(dev, net):
    g = np.load(os.path.join(GOLD, "network_dynamic.npz"))
    pts = _t(g["pts"], dev)
    with torch.no_grad():
        for tag, tv in (("mid", 0.5), ("first", 0.0), ("last", 1.0)):
            t = torch.tensor([[tv]], dtype=torch.float32, device=dev)
            for lidar in (1, 0):
                out = net.density(pts, t, bool(lidar))
                sr, gr = g[f"density_{tag}_{lidar}_sigma"], g[f"density_{tag}_{lidar}_geo"]
                np.testing.assert_allclose(out["sigma"].cpu().numpy(), sr, rtol=3e-3, atol=1e-6, err_msg=f"{tag} {lidar}")
                np.testing.assert_allclose(out["geo_feat"].cpu().numpy(), gr, atol=3e-3 * np.abs(gr).max(), rtol=0)
        fl = net.flow(pts, torch.tensor([[0.5]], device=dev))
    np.testing.assert_allclose(fl["flow_forward"].cpu().numpy(), g["flow_forward"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(fl["flow_backward"].cpu().numpy(), g["flow_backward"], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("lidar", [True, False])
def test_network_render_matches_reference(dev, net, lidar):
    g = np.load(os.path.join(GOLD, "network_dynamic.npz"))
    k = "lidar" if lidar else "cam"
    sfx = "_lidar" if lidar else ""
    o, d = _t(g[f"{k}_rays_o"], dev)[None], _t(g[f"{k}_rays_d"], dev)[None]
    with torch.no_grad():
        out = net.render(o, d, torch.tensor([[0.375]], device=dev), cal_lidar_color=lidar, num_steps=g[f"{k}_z_vals"].shape[1])
    assert np.array_equal(out["z_vals"].cpu().numpy(), g[f"{k}_z_vals"])
    np.testing.assert_allclose(out["weights"].cpu().numpy(), g[f"{k}_weights"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["weights_sum" + sfx].cpu().numpy(), g[f"{k}_weights_sum"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].cpu().numpy(), g[f"{k}_depth"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["image" + sfx][0].cpu().numpy(), g[f"{k}_image"], atol=1e-4, rtol=0)


def test_network_trains_end_to_end(dev, net):
    """One optimisation step through every encoder (gradients reach the hash tables, the planes, the flow MLP)."""
    import copy
    from nvsf import synthetic as S
    m = copy.deepcopy(net).train()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    rng = np.random.default_rng(3)
    o, d = S.lidar_rays(64, rng)
    out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=True, num_steps=32, perturb=True)
    loss = (out["depth_lidar"] - 0.3).abs().mean() + (out["image_lidar"] - 0.5).pow(2).mean()
    loss.backward()
    for name in ("hash_encoder_lidar.hash_static.params", "planes_encoder_lidar.planes.0.0", "planes_encoder_lidar.planes.0.2",
                 "flow_net.mlp.0.weight", "sigma_net.params", "raydrop_net.params"):
        _, grad = GD.reference_named(m)[name]
        assert grad is not None and torch.isfinite(grad).all() and grad.abs().sum() > 0, name
    assert dict(m.named_parameters())["hash_encoder_camera.hash_static.params"].grad is None  # untouched modality
    before = m.sigma_net.params.detach().clone()
    opt.step()
    assert not torch.equal(before, m.sigma_net.params.detach())


def test_training_step_scatters_beside_backward_and_defers_the_last_pass(dev, net):
    """RenderTrainStep on the space-time model: the texel scatter of the K-planes and the table scatters run on the step's side stream
    into the gradient sink, the parameters scattered in the LAST backward pass (the LiDAR encoders) are updated behind their scatter
    and their first reader waits for that (Planes4D.wait_pending_update / the encoders' fp16 caches).  Same gradients as with every
    kernel on one stream (atomic order only), and steps that follow each other stay finite and keep decreasing the loss."""
    import copy
    from nvsf import synthetic as S
    from nvsf.nerf.train_step import RenderTrainStep
    from nvsf.nerf.loss_scaler import LossScaler
    rng = np.random.default_rng(5)
    N = 256
    lo, ld = S.lidar_rays(N, rng)
    co, cd = S.camera_rays(N, rng)
    g = torch.Generator().manual_seed(3)
    batch = {"rays_o_lidar": _t(lo, dev)[None], "rays_d_lidar": _t(ld, dev)[None], "rays_o": _t(co, dev)[None], "rays_d": _t(cd, dev)[None],
             "time": torch.tensor([[0.4]], device=dev), "gt_depth": (torch.rand(1, N, generator=g) * 0.5).to(dev),
             "gt_raydrop": (torch.rand(1, N, generator=g) > 0.3).float().to(dev), "gt_intensity": torch.rand(1, N, generator=g).to(dev),
             "gt_rgb": torch.rand(1, N, 3, generator=g).to(dev)}
    grads, losses = {}, {}
    for overlap in (False, True):
        m = copy.deepcopy(net).train()
        step = RenderTrainStep(m, num_steps=64, scale=S.SCALE, ema_decay=None)
        step.scatter_overlap = overlap
        step.scaler = LossScaler(init_scale=64.0, growth_interval=10 ** 6)
        torch.manual_seed(10)
        loss0, _, _ = step.step(batch)
        grads[overlap] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.numel() and p.grad is not None}
        planes = m.planes_encoder_lidar
        if overlap:
            assert step._pending is not None and any(p is planes.planes_cl for p in step._pending[1])   # updated behind its scatter ...
            assert planes.__dict__.get("_pending_update") is step._pending[0]                             # ... its first reader will wait
            with torch.no_grad():
                m.eval()
                m.render(batch["rays_o_lidar"], batch["rays_d_lidar"], batch["time"], cal_lidar_color=True, num_steps=32)
                m.train()
            assert planes.__dict__.get("_pending_update") is None
        cur = [float(loss0)]
        for _ in range(4):
            torch.manual_seed(10)
            cur.append(float(step.step(batch)[0]))
        step.sync()
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
        losses[overlap] = cur
        assert cur[-1] < cur[0]
    assert set(grads[False]) == set(grads[True])
    for n, a in grads[False].items():
        b = grads[True][n]
        scale = float(a.abs().max())
        if scale > 0:
            assert float((a - b).abs().max()) <= 2e-5 * scale, n
    assert abs(losses[True][-1] - losses[False][-1]) <= 2e-2 * abs(losses[False][-1])


def test_overflow_only_in_a_deferred_table_gradient_never_reaches_the_table(dev, net):
    """with the last pass' parameters deferred, the step's overflow decision is taken from the EARLY gradients.  For the
    space-time model that is not sufficient by itself: the static hash of the field receives its feature gradient through an fp16
    hand-over (nvsf_density_tail_grad_split), where a finite fp32 value above 65504 becomes inf while the density MLP's own weight
    gradient stays finite.  Emulated here by writing an inf into the deferred table's gradient behind its scatter (on the scatter's
    stream, every early gradient clean): the deferred Adam pass must skip (table, fp16 copy and EMA shadow untouched, no nan
    anywhere), everything early may move (its gradients were clean), and the scaler halves its scale one step later -- or at the next
    sync() (end of epoch / checkpoint), whichever comes first."""
    import copy
    from nvsf import field_ops
    from nvsf import synthetic as S
    from nvsf.nerf.train_step import RenderTrainStep
    from nvsf.nerf.loss_scaler import LossScaler
    rng = np.random.default_rng(7)
    N = 128
    lo, ld = S.lidar_rays(N, rng)
    co, cd = S.camera_rays(N, rng)
    g = torch.Generator().manual_seed(4)
    batch = {"rays_o_lidar": _t(lo, dev)[None], "rays_d_lidar": _t(ld, dev)[None], "rays_o": _t(co, dev)[None], "rays_d": _t(cd, dev)[None],
             "time": torch.tensor([[0.4]], device=dev), "gt_depth": (torch.rand(1, N, generator=g) * 0.5).to(dev),
             "gt_raydrop": (torch.rand(1, N, generator=g) > 0.3).float().to(dev), "gt_intensity": torch.rand(1, N, generator=g).to(dev),
             "gt_rgb": torch.rand(1, N, 3, generator=g).to(dev)}
    m = copy.deepcopy(net).train()
    step = RenderTrainStep(m, num_steps=32, scale=S.SCALE, ema_decay=0.95)
    step.scaler = LossScaler(init_scale=1.0, growth_interval=10 ** 6)  # small: no genuine fp16 overflow in these steps
    step.ema.attach(step.opt)  # the every-step form: the shadow update rides in the (deferred) Adam pass
    table = m.hash_encoder_lidar.hash_static.params
    real_run, poisoned = step._run, []

    def run_and_poison(b, defer):
        out = real_run(b, defer)
        late = out[3]
        assert any(p is table for p, _ in late)  # the static hash of the LiDAR field is scattered in the last pass
        if not poisoned:
            with torch.cuda.stream(field_ops.side_stream(table.device)):
                table.grad.view(-1)[12345] = float("inf")
            poisoned.append(True)
        return out
    step._run = run_and_poison
    torch.manual_seed(1)
    step.ema.before_step()
    step.step(batch)  # the poison is in this first step
    assert step.scaler.get_scale() == 1.0       # the early decision was "clean" ...
    step.sync()                                  # ... and sync() settles a late-only overflow that has not reached the scaler yet
    assert step.scaler.get_scale() == 0.5 and step._late_carry is None
    torch.cuda.synchronize()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    names = dict(m.named_parameters())
    # first step: poisoned -> every deferred parameter skipped, nothing non-finite anywhere
    ref = {n: p.detach() for n, p in net.named_parameters()}  # the untrained weights the step started from
    t0 = ref["hash_encoder_lidar.hash_static.params"]
    assert torch.equal(table.detach(), t0)
    assert torch.equal(m.hash_encoder_lidar.hash_static._cache.get(table), t0.half())
    shadow = dict(zip([id(p) for p in step.ema._params], step.ema.shadow_params))[id(table)]
    assert torch.equal(shadow, t0)
    assert not torch.equal(m.sigma_net.params.detach(), ref["sigma_net.params"])  # early parameters were clean and moved
    for n, p in names.items():
        assert bool(torch.isfinite(p).all()), n
    torch.manual_seed(2)
    step.ema.before_step()
    step.step(batch)
    step.sync()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)  # this step itself was clean
    assert step.scaler.get_scale() == 0.5        # settled once, not twice
    assert not torch.equal(table.detach(), before["hash_encoder_lidar.hash_static.params"])  # a clean step updates the table again
    for n, p in names.items():
        assert bool(torch.isfinite(p).all()), n
    # the same without a sync() in between (the training loop's case): the late-only overflow reaches the scaler with the NEXT step's update
    poisoned.clear()
    torch.manual_seed(3)
    step.ema.before_step()
    step.step(batch)                             # poisoned again
    assert step.scaler.get_scale() == 0.5 and step._late_carry is not None
    torch.manual_seed(4)
    step.ema.before_step()
    step.step(batch)
    assert step.scaler.get_scale() == 0.25       # one step later
    torch.manual_seed(5)
    step.ema.before_step()
    step.step(batch)
    step.sync()
    assert step.scaler.get_scale() == 0.25       # carried once, not twice
    for n, p in names.items():
        assert bool(torch.isfinite(p).all()), n


def test_fp32_readers_wait_for_the_deferred_update(dev, net):
    """evaluate_frames(ema=step.ema) / model.state_dict() / optimiser.state_dict() right after step() -- the deferred
    optimiser pass may still be writing the last table, its EMA shadow and Adam state on the side stream.  EMA store / copy_to /
    restore / state_dict and the two state_dict() entry points call RenderTrainStep.sync() themselves."""
    import copy
    from nvsf import synthetic as S
    from nvsf.nerf.train_step import RenderTrainStep
    from nvsf.nerf.loss_scaler import LossScaler
    rng = np.random.default_rng(9)
    N = 128
    lo, ld = S.lidar_rays(N, rng)
    g = torch.Generator().manual_seed(4)
    batch = {"rays_o_lidar": _t(lo, dev)[None], "rays_d_lidar": _t(ld, dev)[None],
             "time": torch.tensor([[0.4]], device=dev), "gt_depth": (torch.rand(1, N, generator=g) * 0.5).to(dev),
             "gt_raydrop": (torch.rand(1, N, generator=g) > 0.3).float().to(dev), "gt_intensity": torch.rand(1, N, generator=g).to(dev)}
    co, cd = S.camera_rays(N, rng)
    batch.update({"rays_o": _t(co, dev)[None], "rays_d": _t(cd, dev)[None], "gt_rgb": torch.rand(1, N, 3, generator=g).to(dev)})
    m = copy.deepcopy(net).train()
    step = RenderTrainStep(m, num_steps=32, scale=S.SCALE, ema_decay=0.95)
    step.scaler = LossScaler(init_scale=64.0, growth_interval=10 ** 6)
    readers = {"ema.store": step.ema.store, "ema.copy_to": step.ema.copy_to, "ema.restore": step.ema.restore, "ema.state_dict": step.ema.state_dict,
               "ema.update": step.ema.update, "model.state_dict": m.state_dict, "optimizer.state_dict": step.opt.state_dict}
    for name, reader in readers.items():
        if name == "ema.restore":
            step.ema.store()
        step.step(batch)
        assert step._pending is not None, name
        reader()
        assert step._pending is None, name  # the reader's stream has been made to wait for the deferred pass
    torch.cuda.synchronize()


def test_host_time_cache_follows_the_tensor_object(dev):
    from nvsf.nerf.models.hash_field import _host_time
    t = torch.tensor([[0.25]], device=dev)
    assert _host_time(t) == 0.25 and _host_time(t) == 0.25
    t.fill_(0.75)  # in-place change: the version moves, the value is read again
    assert _host_time(t) == 0.75
    u = torch.tensor([[0.5]], device=dev)
    assert _host_time(u) == 0.5 and _host_time(t) == 0.75
    del t
    v = torch.tensor([[0.125]], device=dev)  # may re-use the freed address / id: must not see the old value
    assert _host_time(v) == 0.125
    assert _host_time(0.3) == 0.3 and _host_time(torch.tensor(0.4)) == pytest.approx(0.4)


GRAD_KEYS = [f"{tag}_{mod}" for tag, _ in GD.GRAD_CASES for mod in ("lidar", "cam")]
# bars of test_training_graph_gradients_match_reference, relative to each tensor's largest entry (max, median over the tensors of
# the max) and to its L2 norm; SIG_ERR: norms / bucket sums / largest entries of the flow grid's 30 M-entry gradient
MAX_ERR, L2_ERR, MED_ERR, SIG_ERR = 2e-3, 1.5e-3, 6e-4, 5e-3  # measured on MI355X: worst 8.7e-4 / 7.0e-4 / 3.5e-4


@pytest.mark.parametrize("key", GRAD_KEYS)
def test_training_graph_gradients_match_reference(dev, net, key):
    """Parameter gradients of the HIP training graph (DensityTailFn, HashDyn3Fn / HashDynFn, FlowGridFn, PlanesFn, HeadsFn,
    the compositor backward kernels) against gradients of the REFERENCE's own NeRFNetwork.render + Trainer losses
    back-propagated on CPU (tests/golden/network_dynamic_grads.npz: golden_dynamic.gen_network_grads; the reference's code
    decides which neighbour terms carry gradient, network_dynamic.py:242-271).  Same rays, jitter, targets and loss
    (golden_dynamic.reference_losses = trainer.py:186-219, 491-503, 540-543).

    Tolerances.  Forward (image, depth) 1e-4 abs.  The same parameters must receive a gradient, with the same sparsity.
    Gradients.  The MLPs are ReLU networks with fp16 activations.  The geometry features that enter the per-sample heads agree
    between this implementation and the CPU specification to a few 1e-4 (measured: max 4e-4 -- an fp16 rounding of a hidden
    activation of the density MLP that falls the other way), which moves a head's pre-activations by up to ~1e-3; a hidden unit
    whose pre-activation is closer to zero than that is "on" on one side and "off" on the other, and that sample's whole head
    gradient changes by several per cent (measured on mid_cam: dL/dgeo differs by 7 % of its largest entry on one such sample,
    1.4 % in L2 over the batch, while dL/dsigma and dL/drgb agree to 2e-4; replacing nvsf_mlp_bwd by the exact fp64 chain rule
    leaves these figures unchanged: tools/diag_grad_camera.py, tools/diag_grad_inter.py).  So that the comparison cannot hide a
    real error behind that noise, the fixture stores for every case the rows with a head pre-activation within 2e-3 of zero
    (golden_dynamic.fragile_rows, evaluated on the reference side) and BOTH sides cut the heads of exactly those rows out of
    the autograd graph (golden_dynamic.detach_rows around `color`: values unchanged).  What remains -- about half of the rows
    through the heads, every row through sigma -- is free of coin flips and must agree to the bars below for LiDAR and camera
    cases alike."""
    import copy
    from nvsf import synthetic as S
    g = np.load(os.path.join(GOLD, "network_dynamic_grads.npz"))
    tag, mod = key.rsplit("_", 1)
    lidar = mod == "lidar"
    tv = dict(GD.GRAD_CASES)[tag]
    o, d, noise, gt = GD.grad_case_inputs(tag, lidar, S)
    m = copy.deepcopy(net).train()
    noise_dev = _t(noise, dev)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: noise_dev
    rows = torch.from_numpy(g[f"{key}/fragile_rows"]).to(dev)
    assert 0.2 < float(rows.float().mean()) < 0.7  # the comparison keeps a substantial share of the rows through the heads
    real_color = m.color
    m.color = lambda *a, **k: GD.detach_rows(real_color(*a, **k), rows)
    try:
        out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[tv]], dtype=torch.float32, device=dev), cal_lidar_color=lidar,
                       num_steps=GD.GRAD_T, perturb=True, staged=False)
    finally:
        torch.rand = real_rand
        m.color = real_color
    loss = GD.reference_losses(out, _t(gt, dev), lidar)
    sfx = "_lidar" if lidar else ""
    np.testing.assert_allclose(out["image" + sfx][0].detach().cpu().numpy(), g[f"{key}/image"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].detach().cpu().numpy(), g[f"{key}/depth"], atol=1e-4, rtol=0)
    assert abs(float(loss.detach()) - float(g[f"{key}/loss"])) <= 2e-3
    loss.backward()
    want = {k[len(key) + 6:] for k in g.files if k.startswith(key + "/grad/")}
    sigs = {k[len(key) + 9:].rsplit("/", 1)[0] for k in g.files if k.startswith(key + "/gradsig/")}
    params = GD.reference_named(m)
    got = {n for n, (p, gr) in params.items() if gr is not None and p.numel() and bool((gr != 0).any())}
    assert got == want | sigs, (sorted(got - want - sigs), sorted((want | sigs) - got))
    emax, el2 = {}, {}
    for name in sorted(want):
        ref = g[f"{key}/grad/{name}"].astype(np.float64)
        mine = params[name][1].detach().double().cpu().numpy().reshape(ref.shape)
        scale = float(np.abs(ref).max())
        emax[name] = float(np.abs(mine - ref).max()) / scale
        el2[name] = float(np.sqrt(((mine - ref) ** 2).sum() / (ref ** 2).sum()))
        # the sparsity pattern is part of the contract: no gradient where the reference has none (tables; the reference's
        # gradients pass through fp16 tensors, where contributions below 6e-8 vanish -- hence a small allowance, not zero)
        if ref.ndim == 1 and ref.size > 20000:
            assert float(np.abs(mine[ref == 0]).max(initial=0.0)) <= 1e-4 * scale, name
    med = float(np.median(list(emax.values())))
    print(key, "median max-err", med, "worst max-err", max(emax.items(), key=lambda kv: kv[1]), "worst L2", max(el2.items(), key=lambda kv: kv[1]))
    bad = {n: (emax[n], el2[n]) for n in emax if emax[n] > MAX_ERR or el2[n] > L2_ERR}
    assert not bad, bad
    assert med <= MED_ERR, med
    for name in sorted(sigs):  # the flow field's 30 M-entry grid: fingerprint (bucket sums, norms, largest entries)
        mine = params[name][1].detach().double().cpu().numpy().reshape(-1)
        sig = GD.grad_signature(mine)
        ref = {k: g[f"{key}/gradsig/{name}/{k}"] for k in ("buckets", "l1", "l2", "nnz", "top_idx", "top_val")}
        tol = SIG_ERR
        assert abs(float(sig["l2"]) - float(ref["l2"])) <= tol * float(ref["l2"]) and abs(float(sig["l1"]) - float(ref["l1"])) <= tol * float(ref["l1"])
        btol = SIG_ERR
        np.testing.assert_allclose(sig["buckets"], ref["buckets"], atol=btol * float(np.abs(ref["buckets"]).max()), rtol=0)
        np.testing.assert_allclose(mine[ref["top_idx"]], ref["top_val"], atol=btol * float(np.abs(ref["top_val"]).max()), rtol=0)
        # the reference's feature gradients are fp16 tensors: entries whose every contribution lies below 6e-8 are exactly zero
        # there and tiny here (fp32 transport) -- up to 3 % more non-zeros, never fewer
        assert -0.002 * int(ref["nnz"]) <= int(sig["nnz"]) - int(ref["nnz"]) <= 0.03 * int(ref["nnz"])


@pytest.mark.parametrize("M,kind", [(64 * 50, "rays"), (1000, "random"), (7, "random"), (64 * 33 + 5, "rays")])
def test_planes_forward_along_rays_is_bit_identical_and_multi_eval(dev, M, kind, variants):
    """k_planes_fwd_runs (items walk consecutive rows and re-gather only when a plane's texel cell changes) against the
    one-thread-per-(sample, scale) kernel it replaces (testing.variant(planes_fwd="sample"); itself pinned by the reference's fixtures):
    bit-identical static and dynamic features on ray-ordered and on random rows; and nvsf_planes_multi_fwd (static + dynamic +
    two flow-warped dynamic evaluations in one launch) equals the separate calls on explicitly built [M,4] inputs."""
    from nvsf.nerf.models.planes_field import Planes4D
    torch.manual_seed(3)
    enc = Planes4D(resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8]).to(dev)
    with torch.no_grad():
        for p in enc.parameters():
            p.copy_(torch.rand_like(p) + 0.1)
    rng = np.random.default_rng(M)
    if kind == "rays":
        T = 64
        n_rays = (M + T - 1) // T
        o = rng.random((n_rays, 1, 3)) * 0.6 + 0.2
        d = rng.standard_normal((n_rays, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
        x = np.clip(o + d * np.linspace(0, 0.5, T).reshape(1, T, 1), 0, 1).reshape(-1, 3)[:M]
    else:
        x = rng.random((M, 3))
        x[:3] = [[0, 0, 0], [1, 1, 1], [0.999999, 1e-7, 0.5]][:min(3, M)]
    x = _t(x.astype(np.float32), dev)
    tv = 0.37
    xt = torch.cat([x, torch.full((M, 1), tv, device=dev)], -1)
    with torch.no_grad():
        s1, d1 = enc(xt)
        d_only = enc.forward_dynamic(xt)
        variants.set(planes_fwd="sample")
        s0, d0 = enc(xt)
        variants.clear("planes_fwd")
        assert torch.equal(s1, s0) and torch.equal(d1, d0) and torch.equal(d_only, d0)
        flow = (torch.rand(M, 8, device=dev) - 0.5) * 0.02  # rows wider than the six flow components (padded MLP output)
        t1, t2 = 0.4, 0.34375
        outs = enc.forward_multi(x, [(0, None, 0, float(np.float32(tv))), (1, None, 0, float(np.float32(tv))), (1, flow, 0, t1), (1, flow, 3, t2)])
        assert torch.equal(outs[0], s0) and torch.equal(outs[1], d0)
        for o_, col, tn in ((outs[2], 0, t1), (outs[3], 3, t2)):
            xtn = torch.cat([x + flow[:, col:col + 3], torch.full((M, 1), tn, device=dev)], -1)
            assert torch.equal(o_, enc.forward_dynamic(xtn))
        # blend mode: [static, 0.5 d + 0.25 (d1 + d2)] formed in the kernel == the same expression on the separate outputs
        evals = [(0, None, 0, float(np.float32(tv))), (1, None, 0, float(np.float32(tv))), (1, flow, 0, t1), (1, flow, 3, t2)]
        bs, bd = enc.forward_multi(x, evals, blend=True)
        assert torch.equal(bs, s0) and torch.equal(bd, 0.5 * outs[1] + 0.25 * (outs[2] + outs[3]))
        assert torch.equal(0.5 * bd + 0.25 * (bd + bd), bd)  # what the density kernel's own blend makes of an already blended input
        hs, hd = enc.forward_multi(x, evals, blend=True, out_f16=True)  # the same two results as fp16 rows
        assert hs.dtype == torch.float16 and torch.equal(hs, bs.half()) and torch.equal(hd, bd.half())


def test_dynamic_hash_gradient_through_lds_equals_the_run_merging_kernel(dev, variants):
    """nvsf_hashgrid4d_dynamic_bwd_scalar: the LDS form (a workgroup accumulates one level of one pair -- 2^13 / 2^15 scalar sums --
    in LDS and adds it to the global sums once) against the run-merging global-atomic kernel (testing.variant(hash4d_bwd="runs"), itself pinned
    through HashDynFn against the per-slice autograd path above): same sums up to fp32 addition order, on ray-ordered rows, with
    zero gradients mixed in, accumulating into non-zero buffers."""
    import ctypes
    from nvsf import _hip
    from nvsf.nerf.models.hash_field import HashGrid4D
    enc = HashGrid4D(time_resolution=4).to(dev)  # reference defaults: 512 -> 32768, hash_size_dynamic [15, 13, 13]
    specs = [pl.hash_t[0].spec for pl in enc.hash_dynamic]
    assert [s.n_rows // 8 for s in specs] == [2 ** 15, 2 ** 13, 2 ** 13]
    rng = np.random.default_rng(8)
    n_rays, T = 180, 512
    o = rng.random((n_rays, 1, 3)) * 0.5 + 0.25
    d = rng.standard_normal((n_rays, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = _t(np.clip(o + d * np.linspace(0, 0.3, T).reshape(1, T, 1), 0, 1).reshape(-1, 3).astype(np.float32), dev)
    M = x.shape[0]
    assert M >= 1 << 16
    g = torch.randn(M, 24, device=dev)
    g[torch.rand(M, device=dev) < 0.2] = 0.0
    h_scales = _hip.host_f32([v for s in specs for v in s.scales])
    h_res = _hip.host_u32([v for s in specs for v in s.res])
    h_off = _hip.host_u32([v for s in specs for v in s.offsets])
    out = {}
    for variant in ("lds", "runs"):
        if variant == "runs":
            variants.set(hash4d_bwd="runs")
        sums = [torch.full((s.n_rows,), 0.5, dtype=torch.float32, device=dev) for s in specs]
        _hip.call("nvsf_hashgrid4d_dynamic_bwd_scalar", _hip.ptr(x), 3, M, h_scales, h_res, h_off, _hip.ptr(g),
                  (ctypes.c_void_p * 3)(*[t.data_ptr() for t in sums]))
        out[variant] = sums
    for a, b in zip(out["lds"], out["runs"]):
        scale = float((b - 0.5).abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale
        assert torch.equal(a == 0.5, b == 0.5)  # the same rows are touched


@pytest.mark.parametrize("poison", [float("inf"), float("-inf"), float("nan")])
def test_dynamic_hash_gradient_through_lds_keeps_non_finite_gradients(dev, poison):
    """An inf / NaN gradient row (an fp16 overflow under GradScaler) must leave the level's sums non-finite, as the memory-atomic
    kernels do, so that the scaler's found_inf check skips the step -- the fixed-point LDS image must not launder it into finite
    garbage or zeros; the levels without a poisoned entry stay finite."""
    import ctypes
    from nvsf import _hip
    from nvsf.nerf.models.hash_field import HashGrid4D
    enc = HashGrid4D(time_resolution=4).to(dev)
    specs = [pl.hash_t[0].spec for pl in enc.hash_dynamic]
    M = 1 << 16
    x = torch.rand(M, 3, device=dev)
    g = torch.randn(M, 24, device=dev)
    g[12345, 8 + 3] = poison   # pair 1, level 3
    g[:, 16 + 5] = poison      # pair 2, level 5: a whole column (its largest finite |g| is 0)
    h_scales = _hip.host_f32([v for s in specs for v in s.scales])
    h_res = _hip.host_u32([v for s in specs for v in s.res])
    h_off = _hip.host_u32([v for s in specs for v in s.offsets])
    sums = [torch.zeros(s.n_rows, dtype=torch.float32, device=dev) for s in specs]
    _hip.call("nvsf_hashgrid4d_dynamic_bwd_scalar", _hip.ptr(x), 3, M, h_scales, h_res, h_off, _hip.ptr(g),
              (ctypes.c_void_p * 3)(*[t.data_ptr() for t in sums]))
    lvl = lambda p, l: sums[p][specs[p].offsets[l]:specs[p].offsets[l + 1]]
    assert not torch.isfinite(lvl(1, 3)).all() and not torch.isfinite(lvl(2, 5)).all()
    assert torch.isfinite(sums[0]).all() and torch.isfinite(lvl(1, 2)).all() and torch.isfinite(lvl(2, 4)).all()


@pytest.mark.parametrize("M,neighbours", [(29 * 211, (True, True)), (5000, (True, False)), (300, (False, True)), (64 * 130 + 7, (False, False)),
                                          (331 * 211, (True, True)), (70001, (False, True))])  # >= 2^16 rows: time planes through the LDS image
def test_planes_multi_node_equals_one_node_per_evaluation(dev, M, neighbours):
    """ops.PlanesMultiFn (round 5: the K-planes of one density query as ONE autograd node -- nvsf_planes_multi_fwd / nvsf_planes_multi_bwd)
    against one PlanesFn per evaluation, which is pinned by the reference's autograd (test_planes4d_forward_backward): features bit for
    bit, the gradient of the flow offsets bit for bit (same arithmetic per row), texel gradients up to the order of the fp32 additions
    (the fused scatter merges the evaluations' addends of one texel quad); first / last frame (a neighbour absent) and no neighbour at all."""
    from nvsf import field_ops as ops
    from nvsf.nerf.models.planes_field import Planes4D
    rng = np.random.default_rng(M)
    n_rays = max(1, M // 211)
    o = rng.random((n_rays, 1, 3)) * 0.4 + 0.3
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x_np = np.clip(o + d * np.linspace(0, 0.3, 211).reshape(1, 211, 1), 0, 1).reshape(-1, 3)
    x_np = np.concatenate([x_np, rng.random((M - x_np.shape[0], 3))])[:M].astype(np.float32)
    t0, t1, t2 = 0.37, 0.37 + 1 / 64, 0.37 - 1 / 64
    torch.manual_seed(0)
    enc = Planes4D(resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8]).to(dev)
    with torch.no_grad():
        enc.planes_cl.add_(torch.randn_like(enc.planes_cl) * 0.1)   # time planes away from their all-ones initialisation
    x = _t(x_np, dev)
    flow0 = _t((rng.standard_normal((M, 6)) * 3e-3).astype(np.float32), dev)
    gen = torch.Generator().manual_seed(1)
    w = [torch.randn(M, 32, generator=gen).to(dev) for _ in range(4)]
    # ---- one PlanesFn per evaluation (the form of rounds 1-4)
    flow_a = flow0.clone().requires_grad_()
    enc.planes_cl.grad = None
    col = lambda v: torch.full((M, 1), v, dtype=torch.float32, device=dev)
    s_a, d_a = enc(torch.cat([x, col(t0)], -1))
    outs_a = [s_a, d_a]
    if neighbours[0]:
        outs_a.append(enc.forward_dynamic(torch.cat([x + flow_a[:, :3], col(t1)], -1)))
    if neighbours[1]:
        outs_a.append(enc.forward_dynamic(torch.cat([x + flow_a[:, 3:], col(t2)], -1)))
    sum((o_ * w_).sum() for o_, w_ in zip(outs_a, w)).backward()
    gp_a = enc.planes_cl.grad.detach().clone()
    gf_a = flow_a.grad.detach().clone() if flow_a.grad is not None else torch.zeros_like(flow0)
    # ---- one node
    flow_b = flow0.clone().requires_grad_()
    enc.planes_cl.grad = None
    full = ops.PlanesMultiFn.apply(x, flow_b, enc.planes_cl, enc._res_host, float(np.float32(t0)), float(np.float32(t1)) if neighbours[0] else None,
                                   float(np.float32(t2)) if neighbours[1] else None, None)
    outs_b = [full[0], full[1]] + ([full[2]] if neighbours[0] else []) + ([full[3]] if neighbours[1] else [])
    assert (full[2].numel() > 0) == neighbours[0] and (full[3].numel() > 0) == neighbours[1]
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a.detach(), b.detach())
    sum((o_ * w_).sum() for o_, w_ in zip(outs_b, w)).backward()
    gp_b = enc.planes_cl.grad.detach().clone()
    gf_b = flow_b.grad.detach().clone() if flow_b.grad is not None else torch.zeros_like(flow0)
    assert torch.equal(gf_a, gf_b)
    if any(neighbours):
        assert float(gf_a.abs().max()) > 0
    scale = float(gp_a.abs().max())
    assert scale > 0 and float((gp_a - gp_b).abs().max()) <= 2e-5 * scale
    # ---- the blended form (what the fused density tail consumes): second output = 0.5 d + 0.25 (d1 + d2), an absent neighbour = the base
    # evaluation (network_dynamic.py:240-273); its gradient arrives as a column slice of a wider matrix and is read in place
    flow_c = flow0.clone().requires_grad_()
    enc.planes_cl.grad = None
    d1 = outs_a[2] if neighbours[0] else d_a
    d2 = outs_a[-1] if neighbours[1] else d_a
    blend_ref = (0.5 * d_a + 0.25 * (d1 + d2)).detach()
    wide = torch.randn(M, 120, generator=gen).to(dev)   # "the density tail's input gradient": plane_s | plane_d slices
    full = ops.PlanesMultiFn.apply(x, flow_c, enc.planes_cl, enc._res_host, float(np.float32(t0)), float(np.float32(t1)) if neighbours[0] else None,
                                   float(np.float32(t2)) if neighbours[1] else None, None, True)
    assert full[2].numel() == 0 and full[3].numel() == 0
    assert torch.equal(full[0].detach(), s_a.detach()) and torch.equal(full[1].detach(), blend_ref)
    torch.autograd.backward([full[0], full[1]], [wide[:, 0:32], wide[:, 32:64]])
    gp_c, gf_c = enc.planes_cl.grad.detach().clone(), (flow_c.grad.detach().clone() if flow_c.grad is not None else torch.zeros_like(flow0))
    # reference: the same loss through one PlanesFn per evaluation and torch's blend
    flow_d = flow0.clone().requires_grad_()
    enc.planes_cl.grad = None
    s_d, d_d = enc(torch.cat([x, col(t0)], -1))
    e1 = enc.forward_dynamic(torch.cat([x + flow_d[:, :3], col(t1)], -1)) if neighbours[0] else d_d
    e2 = enc.forward_dynamic(torch.cat([x + flow_d[:, 3:], col(t2)], -1)) if neighbours[1] else d_d
    ((s_d * wide[:, 0:32]).sum() + ((0.5 * d_d + 0.25 * (e1 + e2)) * wide[:, 32:64]).sum()).backward()
    gp_d, gf_d = enc.planes_cl.grad.detach().clone(), (flow_d.grad.detach().clone() if flow_d.grad is not None else torch.zeros_like(flow0))
    assert torch.equal(gf_c, gf_d)   # (0.25 g) * ... in the kernel, 0.25 * g by autograd: the same fp32 products
    scale = float(gp_d.abs().max())
    assert scale > 0 and float((gp_c - gp_d).abs().max()) <= 2e-5 * scale


def _ray_rows(n_rays, T, rng, dev):
    o = rng.random((n_rays, 1, 3)) * 0.4 + 0.3
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return _t(np.clip(o + d * np.linspace(0, 0.35, T).reshape(1, T, 1), 0, 1).reshape(-1, 3).astype(np.float32), dev)


@pytest.mark.parametrize("t0", [0.37, 0.0, 1.0])
def test_time_plane_gradient_through_the_lds_image_equals_the_run_merging_kernel(dev, t0, variants):
    """k_planes_multi_bwd_time_lds (production for >= 2^16 rows: the time-plane evaluations of nvsf_planes_multi_bwd accumulate in 1-D fp64
    images in LDS, planes.hip) against k_planes_multi_bwd_runs (testing.variant(planes_bwd="global"): every evaluation as run sums
    into memory-side fp32 atomics, itself pinned against one PlanesFn per evaluation above and through that against the reference's
    autograd): same sums up to fp32 addition order, accumulating into a non-zero buffer, the same texels touched; first / last frame
    (t = 0, 1: the upper time row carries weight 0 / is the border row); and two runs of the production form agree to 2e-6 (the fp64 image
    does not care about arrival order at this precision, what is left is the order of the slice sums per texel), where the run-merging form is
    only good for ~1e-5 here."""
    from nvsf.nerf.models.planes_field import Planes4D
    from planes_calls import multi_bwd_call as _multi_bwd_call
    rng = np.random.default_rng(5)
    torch.manual_seed(0)
    enc = Planes4D(resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8]).to(dev)
    with torch.no_grad():
        enc.planes_cl.add_(torch.randn_like(enc.planes_cl) * 0.1)
    x = _ray_rows(400, 211, rng, dev)
    M = x.shape[0]
    assert M >= 1 << 16
    flow = _t((rng.standard_normal((M, 8)) * 3e-3).astype(np.float32), dev)
    gen = torch.Generator().manual_seed(2)
    g = torch.randn(M, 120, generator=gen).to(dev)
    g[torch.rand(M, generator=gen).to(dev) < 0.1] = 0.0
    times = [t0, t0, min(t0 + 1 / 64, 1.0), max(t0 - 1 / 64, 0.0)]
    base = torch.full_like(enc.planes_cl, 0.25)
    lds = _multi_bwd_call(enc, x, flow, g, times, dev, grad=base.clone())
    lds2 = _multi_bwd_call(enc, x, flow, g, times, dev, grad=base.clone())
    variants.set(planes_bwd="global")
    ref = _multi_bwd_call(enc, x, flow, g, times, dev, grad=base.clone())
    variants.clear("planes_bwd")
    assert torch.equal(lds == 0.25, ref == 0.25)  # the same texels are touched
    for si in range(4):
        for pi in range(6):
            _, _, off, C, H, W = enc._layout[si * 6 + pi]
            a, a2, b = (t[off:off + C * H * W] - 0.25 for t in (lds, lds2, ref))
            scale = float(b.abs().max())
            assert scale > 0
            assert float((a - b).abs().max()) <= 2e-5 * scale, (si, pi)
            if pi in (2, 4, 5):
                assert float((a - a2).abs().max()) <= 2e-6 * scale, (si, pi)


@pytest.mark.parametrize("poison", [float("inf"), float("nan")])
def test_time_plane_gradient_through_the_lds_image_keeps_non_finite_gradients(dev, poison):
    """A non-finite feature gradient (fp16 overflow under GradScaler) must leave the time planes of its scale non-finite, as the
    memory-atomic kernels do (it travels through the run sums and the fp64 image like any addend), so that found_inf skips the step;
    the other scales' time planes stay finite."""
    from nvsf.nerf.models.planes_field import Planes4D
    from planes_calls import multi_bwd_call as _multi_bwd_call
    rng = np.random.default_rng(6)
    torch.manual_seed(0)
    enc = Planes4D(resolution=[32, 32, 32, 8], multiscale_res=[1, 2, 4, 8]).to(dev)
    x = _ray_rows(320, 211, rng, dev)
    M = x.shape[0]
    flow = torch.zeros(M, 6, device=dev)
    g = torch.randn(M, 120, device=dev)
    g[12345, 32 + 8 * 2 + 3] = poison   # dynamic slice, scale 2, channel 3
    gp = _multi_bwd_call(enc, x, flow, g, [0.5, 0.5, 0.5 + 1 / 64, 0.5 - 1 / 64], dev)
    for si in range(4):
        for pi in (2, 4, 5):
            _, _, off, C, H, W = enc._layout[si * 6 + pi]
            assert bool(torch.isfinite(gp[off:off + C * H * W]).all()) == (si != 2), (si, pi)
