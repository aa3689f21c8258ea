"""nvsf_mlp_bwd (fused data + weight gradients of the width-64 MLPs) against two independent formulations:
  * exact small-integer arithmetic (every product and sum representable: a transposed / permuted MFMA fragment
    or a wrong LDS transpose cannot pass), evaluated in numpy float64;
  * fp32 torch autograd of the same network on random data (tolerances of the fp16 operand rounding).
Shapes: every (in_cols, n_hidden) the models instantiate (32/1, 32/2, 96/2, 128/1) plus ragged inputs and sizes."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _numpy_backward(x, mats, g_out, n_in):
    """float64 reference: padded-ones input, ReLU hidden layers, no bias."""
    M = x.shape[0]
    in_cols = mats[0].shape[1]
    a = np.ones((M, in_cols), np.float64)
    a[:, :n_in] = x
    acts = [a]
    for W in mats[:-1]:
        acts.append(np.maximum(acts[-1] @ W.T.astype(np.float64), 0.0))
    g = np.zeros((M, mats[-1].shape[0]), np.float64)
    g[:, :g_out.shape[1]] = g_out
    grads = []
    peak = max(np.abs(a_).max() for a_ in acts)  # largest fp16 operand the kernel has to carry
    for li in range(len(mats) - 1, -1, -1):
        peak = max(peak, np.abs(g).max())
        grads.append(g.T @ acts[li])
        g = g @ mats[li].astype(np.float64)
        if li > 0:
            g = g * (acts[li] > 0)
    return g[:, :n_in], np.concatenate([w.reshape(-1) for w in reversed(grads)]), peak


@pytest.mark.parametrize("kernel", ["wave", "staged"])
@pytest.mark.parametrize("n_in,n_out,n_hidden,M", [(32, 16, 1, 64), (31, 3, 2, 100), (87, 1, 2, 257), (120, 16, 1, 130), (16, 2, 2, 16), (87, 1, 2, 40001),
                                                   (64, 16, 2, 33), (128, 4, 2, 1000)])
def test_mlp_bwd_exact_integers(dev, n_in, n_out, n_hidden, M, kernel, variants):
    """kernel: "wave" = the production form (a wave owns two 16-sample tiles, dW operands transposed on the matrix core, all of dW
    in accumulation registers), "staged" = the LDS-staged kernel it replaced (kept as the test reference)."""
    from nvsf import field_ops as ops
    variants.set(mlp_bwd=kernel)
    spec = ops.MlpSpec(n_in, n_out, 64, n_hidden)
    rng = np.random.default_rng(n_in + M)
    # sparse small integers keep every intermediate exactly representable in fp16 operands / fp32 accumulators
    mats = []
    for a, b in spec.shapes:
        W = rng.integers(-2, 3, size=(a, b)).astype(np.float32)
        W[rng.random((a, b)) < 0.8] = 0
        mats.append(W)
    x = rng.integers(-2, 3, size=(M, n_in)).astype(np.float32)
    g_out = rng.integers(-2, 3, size=(M, n_out)).astype(np.float32)
    if M > 10000:  # several iterations per wave: keep the sums over the samples exactly representable
        g_out[rng.random(M) < 0.97] = 0
    w16 = np.concatenate([m.reshape(-1) for m in mats]).astype(np.float16)
    gx_ref, gw_ref, peak = _numpy_backward(x.astype(np.float64), mats, g_out.astype(np.float64), n_in)
    assert peak <= 2048 and np.abs(gw_ref).max() < 2 ** 24 and np.abs(gx_ref).max() < 2 ** 24  # fp16 operands / fp32 sums stay exact
    gx, gw = ops.mlp_backward(_t(x, dev), _t(w16, dev), spec, _t(g_out, dev), grad_scale=1.0)
    assert np.array_equal(gx.cpu().numpy(), gx_ref.astype(np.float32))
    assert np.array_equal(gw.cpu().numpy(), gw_ref.astype(np.float32))
    # fp16 input rows, no input gradient requested
    gx2, gw2 = ops.mlp_backward(_t(x.astype(np.float16), dev), _t(w16, dev), spec, _t(g_out, dev), need_grad_x=False, grad_scale=1.0)
    assert gx2 is None and np.array_equal(gw2.cpu().numpy(), gw_ref.astype(np.float32))


@pytest.mark.parametrize("n_in,n_out,n_hidden,M", [(32, 16, 1, 5000), (31, 3, 2, 4097), (87, 1, 2, 3000), (120, 16, 1, 2048)])
def test_mlp_bwd_random_vs_fp32_autograd(dev, n_in, n_out, n_hidden, M):
    from nvsf import field_ops as ops
    spec = ops.MlpSpec(n_in, n_out, 64, n_hidden)
    g = torch.Generator().manual_seed(n_in)
    w16 = torch.cat([((torch.rand(a * b, generator=g) * 2 - 1) * (6.0 / (a + b)) ** 0.5) for a, b in spec.shapes]).half().to(dev)
    x = torch.randn(M, n_in, generator=g).to(dev)
    g_out = (torch.randn(M, n_out, generator=g) * 0.01).to(dev)  # small gradients: exercises the fp16 scaling
    gx, gw = ops.mlp_backward(x, w16, spec, g_out)
    # fp32 autograd on the fp16-rounded operands
    mats = [m.float().clone().requires_grad_() for m in spec.split(w16)]
    xr = x.half().float().requires_grad_()
    a = torch.cat([xr, torch.ones(M, spec.in_cols - n_in, device=dev)], 1)
    for W in mats[:-1]:
        a = torch.relu(a @ W.t())
    y = (a @ mats[-1].t())[:, :n_out]
    (y * g_out).sum().backward()
    gw_ref = torch.cat([m.grad.reshape(-1) for m in mats])
    gx_ref = xr.grad
    # fp16 rounding of activations / gradients: relative error ~1e-3 per element, ReLU gates of near-zero units may flip
    sw, sx = float(gw_ref.abs().max()), float(gx_ref.abs().max())
    # measured: kernel vs an fp16 GEMM chain 2-4e-4 of the largest entry; either of them vs fp32 up to 3e-2 (fp16 activations)
    assert float((gw - gw_ref).abs().max()) < 5e-2 * sw
    assert float(((gw - gw_ref).abs() <= 1e-2 * sw).float().mean()) > 0.99
    assert float(((gx - gx_ref).abs() <= 2e-2 * sx).float().mean()) > 0.995
    assert float((gx - gx_ref).abs().mean()) < 2e-3 * sx


def test_mlp_bwd_accumulates_and_rejects(dev):
    from nvsf import field_ops as ops, _hip
    spec = ops.MlpSpec(32, 16, 64, 1)
    x = torch.randn(100, 32, device=dev)
    w16 = (torch.randn(spec.n_params, device=dev) * 0.1).half()
    g_out = torch.randn(100, 16, device=dev)
    _, gw = ops.mlp_backward(x, w16, spec, g_out)
    # the entry point ADDS into grad_weights
    buf = gw.clone()
    _hip.call("nvsf_mlp_bwd", _hip.ptr(x), 0, 100, 32, 32, _hip.ptr(w16), 32, 64, 1, 16, _hip.ptr(g_out), 16, 16, 128.0, None, 0,
              _hip.ptr(buf), 0, 0)
    torch.testing.assert_close(buf, 2 * gw, rtol=1e-5, atol=1e-6)
    with pytest.raises(_hip.NvsfHipError):  # three hidden layers: not built (the GEMM chain serves them)
        _hip.call("nvsf_mlp_bwd", _hip.ptr(x), 0, 100, 32, 32, _hip.ptr(w16), 32, 64, 3, 16, _hip.ptr(g_out), 16, 16, 128.0, None, 0,
                  _hip.ptr(buf), 0, 0)


@pytest.mark.parametrize("kernel", ["wave", "staged"])
@pytest.mark.parametrize("n_enc,n_geo,n_out,M", [(72, 15, 1, 300), (16, 15, 3, 257)])
def test_mlp_bwd_column_window_accumulate_and_aligned_rows(dev, n_enc, n_geo, n_out, M, kernel, variants):
    """The heads' form of the call: x = the first n_in columns of a wider 16-byte aligned fp16 buffer, only the gradient of
    the trailing n_geo columns requested, a second head accumulating into the same buffer.  Exact small-integer case."""
    from nvsf import field_ops as ops
    variants.set(mlp_bwd=kernel)
    n_in = n_enc + n_geo
    spec = ops.MlpSpec(n_in, n_out, 64, 2)
    rng = np.random.default_rng(n_in)
    heads = []
    for _ in range(2):
        mats = []
        for a, b in spec.shapes:
            W = rng.integers(-2, 3, size=(a, b)).astype(np.float32)
            W[rng.random((a, b)) < 0.8] = 0
            mats.append(W)
        heads.append(mats)
    x = rng.integers(-2, 3, size=(M, n_in)).astype(np.float32)
    g_out = rng.integers(-2, 3, size=(M, 2 * n_out)).astype(np.float32)
    buf = torch.full((M, spec.in_cols), 7.0, dtype=torch.float16, device=dev)  # padding columns hold garbage on purpose
    buf[:, :n_in] = _t(x, dev).half()
    grad_geo = torch.full((M, 16), -3.0, dtype=torch.float32, device=dev)[:, :n_geo]
    tg = _t(g_out, dev)
    gx_sum = np.zeros((M, n_in))
    for i, mats in enumerate(heads):
        w16 = _t(np.concatenate([m.reshape(-1) for m in mats]).astype(np.float16), dev)
        gx_ref, gw_ref, peak = _numpy_backward(x.astype(np.float64), mats, g_out[:, i * n_out:(i + 1) * n_out].astype(np.float64), n_in)
        assert peak <= 2048
        gx_sum += gx_ref
        got_x, gw = ops.mlp_backward(buf[:, :n_in], w16, spec, tg[:, i * n_out:(i + 1) * n_out], grad_scale=1.0, grad_x=grad_geo,
                                     gx_col0=n_enc, accumulate=(i == 1))
        assert got_x is grad_geo
        assert np.array_equal(gw.cpu().numpy(), gw_ref.astype(np.float32))
        # forward on the same strided rows agrees with the contiguous call
        y = ops.mlp_forward(buf[:, :n_in], w16, spec)
        y2 = ops.mlp_forward(_t(x, dev), w16, spec)
        assert torch.equal(y, y2)
    assert np.array_equal(grad_geo.cpu().numpy(), gx_sum[:, n_enc:].astype(np.float32))


@pytest.mark.parametrize("lidar", [True, False])
def test_heads_fn_equals_the_cat_formulation(dev, lidar):
    """ops.heads (aligned shared buffer, windowed + accumulated input gradient) against the reference's formulation
    torch.cat([encoder(d), geo_feat]) -> one tcnn.Network per head, values and gradients."""
    from nvsf import field_ops as ops
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(5)
    m = NeRFNetworkStatic(bound=2.0).to(dev)
    M = 3000
    d01 = torch.rand(M, 3, device=dev)
    geo = torch.randn(M, 15, device=dev, requires_grad=True)
    g_h = torch.randn(M, 2 if lidar else 3, device=dev) * 0.05
    nets = [m.raydrop_net, m.intensity_net] if lidar else [m.color_net]
    enc = m.view_encoder_lidar if lidar else m.view_encoder_camera
    h = ops.heads(m, d01, geo, lidar)
    h.backward(g_h)
    got = [h.detach().clone(), geo.grad.clone()] + [n.params.grad.clone() for n in nets]
    geo.grad = None
    for n in nets:
        n.params.grad = None
    logits = torch.cat([enc(d01), geo], dim=-1)
    h2 = torch.cat([n(logits) for n in nets], dim=-1)
    h2.backward(g_h)
    ref = [h2.detach(), geo.grad] + [n.params.grad for n in nets]
    assert torch.equal(got[0], ref[0])  # same fp16 operands, same MFMA order
    s = float(ref[1].abs().max())
    assert float((got[1] - ref[1]).abs().max()) <= 2e-3 * s  # two heads summed in fp32 inside the kernel vs by torch
    for a, b in zip(got[2:], ref[2:]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))


def test_heads_per_ray_direction_encoding(dev):
    """ops.heads with ray_dirs01 (encode each ray's direction once, nvsf_repeat_rows_f16 broadcasts it to the samples) gives
    the logits of the per-sample encoding bit for bit, for both head families."""
    import torch
    from nvsf import field_ops as ops, synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(5)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                          log2_hashmap_size=14).to(dev)
    N, T = 37, 19
    rays = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    d01_ray = (rays + 1) / 2
    d01 = d01_ray.view(N, 1, 3).expand(N, T, 3).reshape(-1, 3)
    geo = torch.randn(N * T, 15, device=dev)
    for lidar in (True, False):
        a = ops.heads(m, d01, geo, lidar)
        b = ops.heads(m, None, geo, lidar, ray_dirs01=d01_ray)
        assert a.shape == b.shape == (N * T, 2 if lidar else 3)
        assert torch.equal(a, b)


def test_glue_kernels_against_torch(dev):
    """nvsf_cast_cols_f16 / nvsf_repeat_rows_f16 / nvsf_sigma_geo_bwd on aligned and unaligned layouts against the torch
    expressions they replace (exact: copies, one fp16 rounding, one fp32 multiply)."""
    import torch
    from nvsf import _hip, field_ops as ops
    from nvsf.nerf import activation
    torch.manual_seed(11)
    M = 3001
    for n_cols, src_w, dtype in ((15, 16, torch.float32), (15, 15, torch.float32), (20, 24, torch.float16), (40, 40, torch.float32),
                                 (70, 72, torch.float32), (1, 1, torch.float32)):
        src = torch.randn(M, src_w, device=dev).to(dtype)[:, :n_cols]
        dst = torch.full((M, n_cols + 9), 7.0, dtype=torch.float16, device=dev)
        ops.cast_cols_f16(src, dst[:, 3:3 + n_cols])
        assert torch.equal(dst[:, 3:3 + n_cols], src.half()) and bool((dst[:, :3] == 7).all()) and bool((dst[:, 3 + n_cols:] == 7).all())
    for n_cols, src_w, dst_w, T in ((72, 72, 96, 19), (16, 16, 32, 7), (15, 16, 21, 5)):
        N = 53
        src = torch.randn(N, src_w, device=dev).half()
        dst = torch.full((N * T, dst_w), 3.0, dtype=torch.float16, device=dev)
        _hip.call("nvsf_repeat_rows_f16", _hip.ptr(src), N, n_cols, src_w, T, _hip.ptr(dst), dst_w)
        assert torch.equal(dst[:, :n_cols], src[:, :n_cols].repeat_interleave(T, dim=0)) and bool((dst[:, n_cols:] == 3).all())
    sigma = torch.exp(torch.randn(M, device=dev) * 8)
    sigma[:4] = torch.tensor([1e-9, 1e9, 0.0, 1.0], device=dev)
    gs = torch.randn(M, device=dev)
    for stride in (16, 15, 20):
        gg = torch.randn(M, stride, device=dev)
        for use_gs, use_gg in ((True, True), (True, False), (False, True)):
            out = torch.full((M, 16), 9.0, device=dev)
            _hip.call("nvsf_sigma_geo_bwd", _hip.ptr(gs) if use_gs else None, _hip.ptr(sigma), _hip.ptr(gg) if use_gg else None, stride, 15, M,
                      _hip.ptr(out), 16, activation._LO, activation._HI)
            ref = torch.zeros(M, 16, device=dev)
            if use_gs:
                ref[:, 0] = gs * sigma.clamp(activation._LO, activation._HI)
            if use_gg:
                ref[:, 1:] = gg[:, :15]
            assert torch.equal(out, ref), (stride, use_gs, use_gg)


def test_input_gradient_in_column_blocks(dev):
    """nvsf_mlp_bwd's column-block layout of dL/dx ([n_in / B][M][B], bits 8..15 of gx_accumulate): the same numbers as the row layout,
    transposed, for B = 2 and 4, written and accumulated; M not a multiple of the tile."""
    from nvsf import field_ops as ops
    spec = ops.MlpSpec(32, 16, 64, 1)
    g = torch.Generator().manual_seed(5)
    M = 3003
    x = torch.randn(M, 32, generator=g).to(dev).half()
    w = (torch.randn(spec.n_params, generator=g) * 0.2).to(dev).half()
    go = torch.randn(M, 16, generator=g).to(dev)
    ref, gw_ref = ops.mlp_backward(x, w, spec, go)
    for B in (2, 4):
        got, gw = ops.mlp_backward(x, w, spec, go, grad_x_blocks=B)
        assert tuple(got.shape) == (32 // B, M, B)
        assert torch.equal(got.permute(1, 0, 2).reshape(M, 32), ref)
        torch.testing.assert_close(gw, gw_ref, rtol=1e-4, atol=1e-5)
        twice, _ = ops.mlp_backward(x, w, spec, go, grad_x=got.clone(), grad_x_blocks=B, accumulate=True)
        torch.testing.assert_close(twice.permute(1, 0, 2).reshape(M, 32), 2 * ref, rtol=1e-6, atol=0)


@pytest.mark.parametrize("n_in,two_heads,blocks", [(32, False, 0), (32, True, 2), (120, True, 0), (32, True, 0), (32, False, 4)])
def test_density_logit_gradient_composed_in_the_kernel(dev, n_in, two_heads, blocks, variants):
    """nvsf_mlp_bwd_density (round 5): the density network's logit gradient [g_sigma clamp(sigma) | g_geo_a (+ g_geo_b)] formed while the
    operands are fetched == nvsf_sigma_geo_bwd (+ the sum of the two heads' rows) followed by nvsf_mlp_bwd: same dL/dx bit for bit (the
    same fp32 values enter the same kernel), dL/dW up to the order of its per-workgroup float atomics.  32-64-16 = the LDS-staged kernel
    (rows and level-major column blocks), 120-64-16 = the wave-independent kernel."""
    from nvsf import field_ops as ops, _hip
    from nvsf.nerf import activation
    M = 5000 + 37
    spec = ops.MlpSpec(n_in, 16, 64, 1)
    g = torch.Generator().manual_seed(n_in + int(two_heads))
    x = (torch.randn(M, n_in, generator=g)).to(dev).half()
    w = (torch.randn(spec.n_params, generator=g) * 0.1).to(dev).half()
    g_sigma = (torch.randn(M, generator=g) * 0.01).to(dev)
    sigma = torch.exp(torch.randn(M, generator=g) * 2).to(dev)
    sigma[:10], sigma[10:20] = 1e-9, 1e8                                   # beyond trunc_exp's clamp on either side
    g_sigma[10:20] = 1e-9                                                  # (the clamped product stays inside fp16 after the loss scale)
    geo_a = torch.full((M, 16), 123.0, device=dev)                         # column 15 is alignment padding: garbage on purpose
    geo_a[:, :15] = (torch.randn(M, 15, generator=g) * 0.01).to(dev)
    geo_b = None
    if two_heads:
        geo_b = torch.full((M, 16), -77.0, device=dev)
        geo_b[:, :15] = (torch.randn(M, 15, generator=g) * 0.01).to(dev)
    lo, hi = activation._LO, activation._HI
    # reference: the matrix form
    summed = geo_a[:, :15] if geo_b is None else (geo_a + geo_b)[:, :15]
    grad_h = torch.empty(M, 16, device=dev)
    _hip.call("nvsf_sigma_geo_bwd", _hip.ptr(g_sigma), _hip.ptr(sigma), _hip.ptr_rows(summed), summed.stride(0), 15, M, _hip.ptr(grad_h), 16, lo, hi)
    gx_ref, gw_ref = ops.mlp_backward(x, w, spec, grad_h, grad_x_blocks=blocks)
    parts = ops.density_logit_gradient_parts(g_sigma, sigma, geo_a[:, :15], None if geo_b is None else geo_b[:, :15], 15, (lo, hi))
    assert parts is not None
    gx, gw = ops.mlp_backward(x, w, spec, None, grad_x_blocks=blocks, density_grad=parts)
    assert torch.equal(gx, gx_ref)
    scale = float(gw_ref.abs().max())
    assert scale > 0 and float((gw - gw_ref).abs().max()) <= 2e-6 * scale
    # layouts the kernel does not read are refused by the helper (the caller then takes the matrix form)
    assert ops.density_logit_gradient_parts(g_sigma, sigma, geo_a[:, 1:16], None, 15, (lo, hi)) is None      # rows not 16-byte aligned
    assert ops.density_logit_gradient_parts(g_sigma, sigma, geo_a[:, :15].contiguous(), None, 15, (lo, hi)) is None  # 15-float rows
