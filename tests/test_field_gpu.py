"""GPU parity of the per-sample field operators (hash grid, Frequency, SH4, fused MLP) and of the uniform
sampler / compositor kernels against the CPU oracle (oracle/field_oracle.c).

Bars: hash grid forward, SH4, sampler: bit-exact (same fp32 operation order, fp16 RNE rounding);
Frequency: <= 1 fp16 ulp (sinpi/cospi vs fp64 sin/cos); MLP: fp16 outputs within 2 ulp-ish (MFMA summation
order is unspecified) -> atol 2e-3*|scale|; compositor: 1e-6 abs (parallel scan vs sequential product).
"""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def ops(dev):
    from nvsf import field_ops
    return field_ops


GRID_CASES = [
    # D, L, F, log2T, base, max   (what the reference / BASELINE instantiate)
    (3, 16, 2, 19, 16, 2048),     # BASELINE config 2
    (3, 8, 4, 19, 512, 32768),    # reference default static grid (hash_field.py:107-119)
    (2, 8, 4, 15, 512, 32768),    # reference time-slice grids (hash_field.py:47-57; 2^15 / 2^13)
    (2, 8, 4, 13, 512, 32768),
    (3, 16, 8, 18, 32, 8192),     # flow field grid (flow_field.py:68-84)
    (3, 4, 2, 10, 4, 32),         # tiny: every level dense or barely hashed, exercises the dense/hash switch
]


def _spec(ops, D, L, F, log2T, base, mx):
    pls = float(np.exp2(np.log2(mx / base) / (L - 1)))
    return ops.GridSpec(D, L, F, log2T, base, pls)


@pytest.mark.parametrize("case", GRID_CASES)
def test_hashgrid_forward_bit_exact(ops, dev, case):
    D, L, F, log2T, base, mx = case
    spec = _spec(ops, *case)
    rng = np.random.default_rng(D * 100 + L)
    M = 10007  # ragged: not a multiple of the 64-sample workgroup tile
    x = rng.random((M, 4)).astype(np.float32)
    x[:8] = np.array([[0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 1], [1, 0, 1, 0], [0.5, 0.5, 0.5, 0.5], [1e-7, 1 - 1e-7, 0.25, 0.75],
                      [0.999999, 0.000001, 0.5, 0.5], [0.333333, 0.666667, 0.1, 0.9]], np.float32)
    table = (rng.standard_normal(spec.n_params) * 0.5).astype(np.float16)
    cols = (0, 2, 1)[:D] if D == 3 else (1, 3)  # non-trivial column selection
    ref = O.hashgrid_fwd(x, cols, table, spec)
    got = ops.hashgrid_forward(_t(x, dev), cols, _t(table, dev), spec).cpu().numpy()
    assert got.dtype == np.float16
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))


def test_hashgrid_forward_one_level_per_xcd_is_bit_identical(ops, dev, variants):
    """Eight hashed F = 4 levels on a large batch take k_hashgrid_fwd_levels8 (one level per XCD, two lanes per sample).  Same fma
    chains as the generic kernel: equal bit for bit, also on a ragged batch, boundary samples and a strided input; a sample of
    the rows is checked against the oracle."""
    spec = _spec(ops, 3, 8, 4, 19, 512, 32768)
    rng = np.random.default_rng(7)
    M = (1 << 16) + 4099
    x = rng.random((M, 4)).astype(np.float32)
    x[:6, :3] = np.array([[0, 0, 0], [1, 1, 1], [0, 1, 0], [0.5, 0.5, 0.5], [1e-7, 1 - 1e-7, 0.25], [0.999999, 0.000001, 0.5]], np.float32)
    table = (rng.standard_normal(spec.n_params) * 0.5).astype(np.float16)
    xd, td = _t(x, dev), _t(table, dev)
    got = ops.hashgrid_forward(xd, (0, 1, 2), td, spec)
    variants.set(hashgrid_fwd="generic")
    ref = ops.hashgrid_forward(xd, (0, 1, 2), td, spec)
    variants.clear("hashgrid_fwd")
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    rows = np.concatenate([np.arange(64), rng.integers(0, M, 400), np.arange(M - 40, M)])
    exp = O.hashgrid_fwd(x[rows], (0, 1, 2), table, spec)
    assert np.array_equal(got.cpu().numpy()[rows].view(np.uint16), exp.view(np.uint16))


@pytest.mark.parametrize("M", [(1 << 16) + 4099, 77, 1])
def test_hashgrid_forward_level_major_holds_the_rows(ops, dev, M):
    """nvsf_hashgrid_fwd_level_major: fp16 [L, M, F] with out[l, m, f] == rows[m, l F + f] bit for bit (any batch size, strided
    input, boundary samples); grids it is not built for are refused (the caller keeps the row form)."""
    from nvsf import _hip
    spec = _spec(ops, 3, 8, 4, 19, 512, 32768)
    rng = np.random.default_rng(9)
    x = rng.random((M, 4)).astype(np.float32)
    x[:min(M, 4), :3] = np.array([[0, 0, 0], [1, 1, 1], [1e-7, 1 - 1e-7, 0.25], [0.999999, 0.000001, 0.5]], np.float32)[:min(M, 4)]
    table = (rng.standard_normal(spec.n_params) * 0.5).astype(np.float16)
    xd, td = _t(x, dev), _t(table, dev)
    assert ops.level_major_eligible(spec)
    lm = ops.hashgrid_forward_level_major(xd, td, spec)
    assert lm.shape == (8, M, 4) and lm.dtype == torch.float16
    rows = ops.hashgrid_forward(xd, (0, 1, 2), td, spec)
    assert torch.equal(lm.permute(1, 0, 2).reshape(M, 32).view(torch.int16), rows.view(torch.int16))
    exp = O.hashgrid_fwd(x[:64], (0, 1, 2), table, spec)
    assert np.array_equal(rows.cpu().numpy()[:64].view(np.uint16), exp.view(np.uint16))
    for other in (_spec(ops, 3, 16, 2, 19, 16, 2048), _spec(ops, 3, 8, 4, 19, 16, 2048)):  # another shape; dense coarse levels
        assert not ops.level_major_eligible(other)
        with pytest.raises(_hip.NvsfHipError):
            ops.hashgrid_forward_level_major(xd, torch.zeros(other.n_params, dtype=torch.float16, device=dev), other)


def test_hashgrid_level_table_matches_published_rules(ops):
    spec = _spec(ops, 3, 16, 2, 19, 16, 2048)
    assert spec.res[0] == 16 and spec.res[-1] == 2048
    rows = np.diff(spec.offsets)
    assert rows[0] == 4096 and rows.max() == 2 ** 19 and np.all(rows % 8 == 0)
    assert spec.n_params == 2 * spec.offsets[-1]
    dense = [r ** 3 <= n for r, n in zip(spec.res, rows)]
    assert dense[0] and not dense[-1]


def test_hashgrid_backward(ops, dev):
    spec = _spec(ops, 3, 8, 4, 14, 16, 256)
    rng = np.random.default_rng(21)
    M = 5000
    x = rng.random((M, 3)).astype(np.float32)
    go = rng.standard_normal((M, spec.L * spec.F)).astype(np.float32)
    ref = O.hashgrid_bwd(x, (0, 1, 2), spec, go)
    got32 = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go, dev)).cpu().numpy()
    np.testing.assert_allclose(got32, ref, atol=2e-4, rtol=1e-4)  # fp32 atomics: order-dependent rounding
    got16 = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go.astype(np.float16), dev)).cpu().numpy()
    ref16 = O.hashgrid_bwd(x, (0, 1, 2), spec, go.astype(np.float16).astype(np.float32))
    np.testing.assert_allclose(got16, ref16, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("D,L,F,log2_T,base,top", [(3, 16, 2, 14, 16, 512), (3, 8, 4, 12, 16, 256), (3, 16, 8, 12, 32, 2048), (2, 8, 4, 10, 16, 256)])
@pytest.mark.parametrize("variant", ["corners", "atomic"])
def test_hashgrid_backward_kernels_agree_with_the_oracle(ops, dev, D, L, F, log2_T, base, top, variant, variants):
    """The production form of the table gradient (corner-parallel run merging) and the plain one-thread-per-(row, level) kernel
    (fallback shape + test reference) on ray-ordered rows -- long runs inside one cell at the coarse levels, a new cell per
    row at the fine ones -- with zero-gradient rows and zero features mixed in, fp32 and fp16 gradients."""
    variants.set(hashgrid_bwd=variant)
    spec = _spec(ops, D, L, F, log2_T, base, top)
    rng = np.random.default_rng(D * 100 + L + F)
    n_rays, T = 37, 97  # M is not a multiple of the chunk length
    o = rng.random((n_rays, 1, 3)) * 0.5 + 0.1
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = np.linspace(0.0, 0.35, T).reshape(1, T, 1)
    x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)
    M = x.shape[0]
    go = rng.standard_normal((M, L * F)).astype(np.float32)
    go[rng.random(M) < 0.3] = 0.0                 # rows without gradient
    go[rng.random((M, L * F)) < 0.2] = 0.0         # single features without gradient
    cols = (0, 1, 2)[:D]
    for g_in in (go, go.astype(np.float16)):
        ref = O.hashgrid_bwd(x, cols, spec, g_in.astype(np.float32))
        got = ops.hashgrid_backward(_t(x, dev), cols, spec, _t(g_in, dev)).cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=5e-4, rtol=1e-4)  # fp32 atomics: order-dependent rounding


@pytest.mark.parametrize("form", ["atomic", "corners", "binned"])
def test_hashgrid_backward_positions_outside_the_unit_cube(ops, dev, form, variants):
    """tiny-cuda-nn wraps ANY cell index into the level's table (index modulo size), so a position outside [0, 1]^3 -- a flow-warped
    neighbour position, a caller that does not normalise -- scatters to some in-range row instead of faulting.  Every form of the table
    gradient does the same as the oracle for positions in [-0.4, 1.5]^3, dense levels (whose one-subtraction wrap only covers the unit
    cube) included."""
    spec = _spec(ops, 3, 16, 2, 14, 16, 512)
    rng = np.random.default_rng(77)
    n_rays, T = 33, 96
    o = rng.random((n_rays, 1, 3)) * 1.9 - 0.4
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = (o + d * np.linspace(0.0, 0.3, T).reshape(1, T, 1)).reshape(-1, 3).astype(np.float32)
    assert (x < 0).any() and (x > 1).any()
    go = rng.standard_normal((x.shape[0], spec.L * spec.F)).astype(np.float32)
    ref = O.hashgrid_bwd(x, (0, 1, 2), spec, go)
    if form == "binned":
        got = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go, dev), fine_from=(0, 16))
    else:
        variants.set(hashgrid_bwd=form)
        got = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go, dev))
    np.testing.assert_allclose(got.cpu().numpy(), ref, atol=5e-4, rtol=1e-4)
    fwd = ops.hashgrid_forward(_t(x, dev), (0, 1, 2), _t((rng.standard_normal(spec.n_params) * 0.5).astype(np.float16), dev), spec)
    assert bool(torch.isfinite(fwd.float()).all())


@pytest.mark.parametrize("L,F,log2_T,base,top,fine_from,merge_from", [
    (16, 2, 14, 16, 512, 12, 12), (16, 2, 14, 16, 512, 5, 5), (8, 4, 12, 16, 256, 4, 4), (8, 4, 13, 64, 1024, 0, 0),
    (16, 2, 14, 16, 512, 12, 0), (16, 2, 14, 16, 512, 9, 3), (8, 4, 12, 16, 256, 6, 0), (16, 2, 14, 16, 512, 16, 0)])
def test_hashgrid_backward_binned_levels_agree_with_the_oracle(ops, dev, L, F, log2_T, base, top, fine_from, merge_from):
    """nvsf_hashgrid_bwd_binned: the levels from `fine_from` on go through the bins one contribution per (row, vertex), the levels
    from `merge_from` on as sums over runs of consecutive rows in one cell (dense levels with a short last bin included), the others
    through the run-merging atomics; ray-ordered rows with zero rows / zero features, fp32 and fp16 gradients, M not a multiple of the
    tile; a batch in pieces; then a batch whose samples all sit in ONE cell, which overflows the bins (the direct-add path)."""
    spec = _spec(ops, 3, L, F, log2_T, base, top)
    fine_from = (merge_from, fine_from)
    rows = np.diff(spec.offsets)
    assert all(int(spec.res[l]) ** 3 > rows[l] for l in range(fine_from[1], L))
    rng = np.random.default_rng(L + F + sum(fine_from))
    n_rays, T = 41, 131
    o = rng.random((n_rays, 1, 3)) * 0.5 + 0.1
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = np.clip(o + d * np.linspace(0.0, 0.35, T).reshape(1, T, 1), 0.0, 1.0).reshape(-1, 3).astype(np.float32)
    M = x.shape[0]
    go = rng.standard_normal((M, L * F)).astype(np.float32)
    go[rng.random(M) < 0.3] = 0.0
    go[rng.random((M, L * F)) < 0.2] = 0.0
    for g_in in (go, go.astype(np.float16)):
        ref = O.hashgrid_bwd(x, (0, 1, 2), spec, g_in.astype(np.float32))
        got = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(g_in, dev), fine_from=fine_from).cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=5e-4, rtol=1e-4)
    # accumulates into the caller's table, twice
    acc = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go, dev), fine_from=fine_from)
    acc = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go, dev), grad_table=acc, fine_from=fine_from).cpu().numpy()
    np.testing.assert_allclose(acc, 2 * O.hashgrid_bwd(x, (0, 1, 2), spec, go), atol=1e-3, rtol=1e-4)
    # the same gradient handed over level by level ([L, M, F], what mlp_backward(grad_x_blocks=F) writes)
    go_lm = _t(np.ascontiguousarray(go.reshape(M, L, F).transpose(1, 0, 2)), dev)
    got = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, go_lm, fine_from=fine_from).cpu().numpy()
    np.testing.assert_allclose(got, O.hashgrid_bwd(x, (0, 1, 2), spec, go), atol=5e-4, rtol=1e-4)
    # a batch longer than the bins' fixed-point headroom allows goes in pieces
    old_max, ops._BIN_ROWS_MAX = ops._BIN_ROWS_MAX, 1777
    try:
        got = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, _t(go, dev), fine_from=fine_from).cpu().numpy()
    finally:
        ops._BIN_ROWS_MAX = old_max
    np.testing.assert_allclose(got, O.hashgrid_bwd(x, (0, 1, 2), spec, go), atol=5e-4, rtol=1e-4)
    # every sample in one cell: 8 rows per level receive everything, far beyond a bin's capacity
    M2 = 20001
    x2 = (np.array([[0.3141, 0.2718, 0.5772]]) + rng.random((M2, 3)) * 1e-6).astype(np.float32)
    go2 = (rng.standard_normal((M2, L * F)) * 0.01).astype(np.float32)
    ref = O.hashgrid_bwd(x2, (0, 1, 2), spec, go2)
    got = ops.hashgrid_backward(_t(x2, dev), (0, 1, 2), spec, _t(go2, dev), fine_from=fine_from).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-4 * float(np.abs(ref).max()), rtol=1e-4)


def test_hashgrid_autograd_module(dev):
    import tinycudann as tcnn
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12,
                            "base_resolution": 8, "per_level_scale": 2.0}).to(dev)
    with torch.no_grad():
        enc.params.normal_(0.0, 1.0)  # normal fp16 range: doubling commutes with rounding (not so for subnormal outputs)
    x = torch.rand(257, 3, device=dev)
    y = enc(x)
    assert y.dtype == torch.float16 and y.shape == (257, 8)
    (y.float() ** 2).sum().backward()
    assert enc.params.grad is not None and enc.params.grad.shape == enc.params.shape and enc.params.grad.abs().sum() > 0
    # linearity of the encoding in the table: encode(2*table) == 2*encode(table) exactly in fp16 (powers of two)
    with torch.no_grad():
        enc.params.mul_(2.0)
    y2 = enc(x).detach()
    normal = y.detach().abs() > 2.0 ** -13  # below that the fp16 result is subnormal and rounds on an absolute grid
    assert torch.equal(y2[normal], (y.detach() * 2)[normal])
    assert torch.allclose(y2.float(), y.detach().float() * 2, atol=2.0 ** -23)


def test_frequency_and_sh(ops, dev):
    rng = np.random.default_rng(22)
    d = rng.random((4099, 3)).astype(np.float32)
    d[:3] = [[0, 0, 0], [1, 1, 1], [0.5, 0.25, 0.75]]
    fr, fg = O.freq_encode(d, 12), ops.freq_encode(_t(d, dev), 12).cpu().numpy()
    assert fg.shape == (4099, 72)
    diff = np.abs(fg.astype(np.float32) - fr.astype(np.float32))
    assert diff.max() <= 2 ** -10 and (diff > 0).mean() < 0.02  # at most 1 fp16 ulp (values in [-1,1]), rarely
    sr, sg = O.sh4_encode(d), ops.sh4_encode(_t(d, dev)).cpu().numpy()
    assert np.array_equal(sg.view(np.uint16), sr.view(np.uint16))


MLP_CASES = [  # n_in, n_out, n_hidden, input dtype
    (32, 16, 1, np.float16),   # sigma net, BASELINE config 2
    (120, 16, 1, np.float32),  # sigma net, reference default (network_dynamic.py:125-135)
    (87, 1, 2, np.float32),    # intensity / raydrop nets
    (31, 3, 2, np.float16),    # colour net
    (48, 5, 3, np.float32),    # odd k-step count (in_cols 48 -> zero-filled second half of the last step)
]


@pytest.mark.parametrize("case", MLP_CASES)
def test_mlp_forward(ops, dev, case):
    n_in, n_out, n_hidden, dt = case
    spec = ops.MlpSpec(n_in, n_out, 64, n_hidden)
    rng = np.random.default_rng(n_in)
    M = 3001
    x = rng.standard_normal((M, n_in)).astype(dt)
    w = np.concatenate([(rng.uniform(-1, 1, a * b) * np.sqrt(6.0 / (a + b))).astype(np.float16) for a, b in spec.shapes])
    ref = O.mlp_fwd(x, w, n_in, spec.in_cols, n_hidden)
    got = ops.mlp_forward(_t(x, dev), _t(w, dev), spec).cpu().numpy()
    assert got.dtype == np.float32 and ref.dtype == np.float32
    scale = np.abs(ref).max()
    # fp32 logits: identical up to fp32 summation order, except where a hidden activation's fp16 rounding flipped
    # (MFMA accumulation order is unspecified; measured flip rate < 1e-3 per activation, effect <= 2^-11 |h w|)
    np.testing.assert_allclose(got, ref, atol=1e-3 * scale, rtol=0)
    assert (np.abs(got - ref) <= 2e-6 * scale).mean() > 0.97


def test_mlp_forward_stores_only_the_leading_outputs(ops, dev):
    """out_cols in 1..4 (mlp_forward(n_store=...)): the leading outputs of the same evaluation, bit for bit, into a narrow buffer --
    one column of an interleaved [M, 2] buffer (two heads side by side) and an unaligned three-column view; the neighbouring
    columns are not touched."""
    spec = ops.MlpSpec(87, 1, 64, 2)
    rng = np.random.default_rng(9)
    M = 2049
    x = _t(rng.standard_normal((M, 87)).astype(np.float16), dev)
    w = _t(np.concatenate([(rng.uniform(-1, 1, a * b) * np.sqrt(6.0 / (a + b))).astype(np.float16) for a, b in spec.shapes]), dev)
    full = ops.mlp_forward(x, w, spec)
    both = torch.full((M, 2), 7.0, device=dev)
    ops.mlp_forward(x, w, spec, out=both[:, 1:], n_store=1)
    assert torch.equal(both[:, 1], full[:, 0]) and bool((both[:, 0] == 7.0).all())
    for n in (2, 3, 4):
        buf = torch.full((M, 5), 7.0, device=dev)
        ops.mlp_forward(x, w, spec, out=buf[:, 1:], n_store=n)
        assert torch.equal(buf[:, 1:1 + n], full[:, :n]) and bool((buf[:, 0] == 7.0).all()) and bool((buf[:, 1 + n:] == 7.0).all())
    with pytest.raises(Exception):
        ops.mlp_forward(x, w, spec, out=torch.empty(M, 8, device=dev), n_store=5)


def test_mlp_operand_layout_with_integers(ops, dev):
    """Exact-integer check of the MFMA fragment maps (asymmetric weights: a transposed or permuted fragment
    cannot pass): one hidden layer, W0 = small integers, identity-like second layer."""
    spec = ops.MlpSpec(32, 16, 64, 1)
    rng = np.random.default_rng(5)
    W0 = rng.integers(-3, 4, size=(64, 32)).astype(np.float16)
    W1 = np.zeros((16, 64), np.float16)
    for o in range(16):
        W1[o, (7 * o + 3) % 64] = 1.0
        W1[o, (11 * o + 5) % 64] = -2.0
    x = rng.integers(-4, 5, size=(64, 32)).astype(np.float16)
    w = np.concatenate([W0.reshape(-1), W1.reshape(-1)])
    h = np.maximum(x.astype(np.float32) @ W0.astype(np.float32).T, 0)
    ref = h @ W1.astype(np.float32).T
    got = ops.mlp_forward(_t(x, dev), _t(w, dev), spec).cpu().numpy()
    assert np.array_equal(got, ref)
    assert np.array_equal(O.mlp_fwd(x, w, 32, 32, 1), ref)


def test_mlp_autograd_module(dev):
    import tinycudann as tcnn
    net = tcnn.Network(31, 3, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                               "n_hidden_layers": 2}).to(dev)
    x = torch.randn(513, 31, device=dev, requires_grad=True)
    y = net(x)
    assert y.shape == (513, 3) and y.dtype == torch.float32
    y.float().sum().backward()
    # compare with a plain fp32 torch evaluation of the same weights
    mats = [m.float() for m in net.spec.split(net.params.detach())]
    xr = x.detach().clone().requires_grad_()
    a = torch.cat([xr, torch.ones(513, 1, device=dev)], 1)
    for W in mats[:-1]:
        a = torch.relu(a @ W.t())
    yr = (a @ mats[-1].t())[:, :3]
    yr.sum().backward()
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), yr.detach().cpu().numpy(), atol=2e-2, rtol=2e-2)
    # fp16 GEMM chain vs fp32: ReLU gates of near-zero pre-activations may differ, so compare in aggregate
    gx, gxr = x.grad.cpu().numpy(), xr.grad.cpu().numpy()
    assert (np.abs(gx - gxr) <= 3e-2 + 5e-2 * np.abs(gxr)).mean() > 0.99
    assert np.abs(gx - gxr).mean() < 2e-3 * np.abs(gxr).mean() + 1e-3
    assert net.params.grad.shape == net.params.shape and torch.isfinite(net.params.grad).all()


@pytest.mark.parametrize("N,T,perturb", [(257, 64, False), (100, 768, True), (3, 1, False), (65, 100, True)])
def test_uniform_sampler_and_compositor(ops, dev, N, T, perturb):
    rng = np.random.default_rng(N + T)
    from nvsf import synthetic as S
    o, d = S.camera_rays(N, rng)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.02)
    lin = torch.linspace(0.0, 1.0, T).numpy()
    noise = rng.random((N, T)).astype(np.float32) if perturb else None
    zr, xr = O.uniform_samples(o, d, nears, fars, lin, noise, aabb)
    zg, xg = ops.uniform_samples(_t(o, dev), _t(d, dev), _t(nears, dev), _t(fars, dev), T, _t(aabb, dev),
                                 _t(noise, dev) if perturb else None)
    assert np.array_equal(zg.cpu().numpy(), zr) and np.array_equal(xg.cpu().numpy(), xr)
    sig = (rng.random((N, T)).astype(np.float32) * 40.0)
    sig[rng.random((N, T)) < 0.4] = 0
    for k in (1.0, 2.0):
        wr, wsr, dpr = O.composite_uniform_weights(sig, zr, nears, fars, k)
        wg, wsg, dpg = ops.CompositeWeightsFn.apply(_t(sig, dev), zg, _t(nears, dev), _t(fars, dev), k)
        np.testing.assert_allclose(wg.cpu().numpy(), wr, atol=1e-6, rtol=0)
        np.testing.assert_allclose(wsg.cpu().numpy(), wsr, atol=2e-6, rtol=0)
        np.testing.assert_allclose(dpg.cpu().numpy(), dpr, atol=2e-6, rtol=0)
    for C, bg in ((3, np.array([1, 0.5, 0.25], np.float32)), (2, None)):
        rgb = rng.random((N, T, C)).astype(np.float32)
        ir = O.composite_uniform_image(wr, rgb, wsr, bg)
        ig = ops.CompositeImageFn.apply(_t(wr, dev), _t(rgb, dev), _t(wsr, dev), _t(bg, dev) if bg is not None else None)
        np.testing.assert_allclose(ig.cpu().numpy(), ir, atol=2e-6, rtol=0)


def test_compositor_backward_matches_torch_autograd(ops, dev):
    """Gradient of the weights / image kernels against torch autograd through the reference's own formulas
    (renderer_dynamic.py:181-194, 216-224, 236-237) evaluated in fp64."""
    N, T = 97, 130
    g = torch.Generator().manual_seed(3)
    z = torch.sort(torch.rand(N, T, generator=g) * 0.8 + 0.05, dim=1).values
    nears, fars = z[:, 0].clone(), z[:, -1].clone() + 0.01
    sig = torch.rand(N, T, generator=g) * 30
    rgb = torch.rand(N, T, 3, generator=g)
    bg = torch.tensor([1.0, 0.3, 0.6])
    gw, gws, gdp, gim = torch.randn(N, T, generator=g), torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(N, 3, generator=g)

    def reference(sig, rgb):
        z64 = z.double()
        deltas = torch.cat([z64[:, 1:] - z64[:, :-1], ((fars - nears) / T).double()[:, None]], -1)
        alphas = 1 - torch.exp(-deltas * 1.0 * sig)
        shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-15], -1)
        w = alphas * torch.cumprod(shifted, -1)[:, :-1]
        ws, dp = w.sum(-1), (w * z64).sum(-1)
        img = (w.unsqueeze(-1) * rgb).sum(-2) + (1 - ws).unsqueeze(-1) * bg.double()
        return w, ws, dp, img

    s64, c64 = sig.double().requires_grad_(), rgb.double().requires_grad_()
    w, ws, dp, img = reference(s64, c64)
    ((w * gw).sum() + (ws * gws).sum() + (dp * gdp).sum() + (img * gim).sum()).backward()

    sd, cd = sig.to(dev).requires_grad_(), rgb.to(dev).requires_grad_()
    wg, wsg, dpg = ops.CompositeWeightsFn.apply(sd, z.to(dev), nears.to(dev), fars.to(dev), 1.0)
    imgg = ops.CompositeImageFn.apply(wg, cd, wsg, bg.to(dev))
    ((wg * gw.to(dev)).sum() + (wsg * gws.to(dev)).sum() + (dpg * gdp.to(dev)).sum() + (imgg * gim.to(dev)).sum()).backward()
    np.testing.assert_allclose(imgg.detach().cpu().numpy(), img.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(cd.grad.cpu().numpy(), c64.grad.numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(sd.grad.cpu().numpy(), s64.grad.numpy(), atol=2e-5, rtol=2e-3)


def test_hashgrid_backward_accumulates_and_ignores_zero_gradients(ops, dev):
    """nvsf_hashgrid_bwd ADDS into the caller's table (the multi-rank step scatters straight into its gradient bucket) and a
    zero gradient is a no-op."""
    spec = _spec(ops, 3, 16, 2, 15, 16, 1024)
    rng = np.random.default_rng(5)
    n_rays, T = 61, 128
    o = rng.random((n_rays, 1, 3)) * 0.4 + 0.2
    d = rng.standard_normal((n_rays, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = _t(np.clip(o + d * np.linspace(0, 0.3, T).reshape(1, T, 1), 0, 1).reshape(-1, 3).astype(np.float32), dev)
    go = torch.randn(x.shape[0], 32, device=dev)
    a = ops.hashgrid_backward(x, (0, 1, 2), spec, go)
    base = torch.full((spec.n_params,), 3.0, device=dev)
    acc = ops.hashgrid_backward(x, (0, 1, 2), spec, go, grad_table=base.clone())
    assert float(a.abs().max()) > 0 and torch.allclose(acc, a + 3.0, rtol=0, atol=1e-4) and torch.equal(acc[a == 0], base[a == 0])
    z = ops.hashgrid_backward(x, (0, 1, 2), spec, torch.zeros_like(go), grad_table=base.clone())
    assert torch.equal(z, base)


@pytest.mark.parametrize("L,F,log2_T,base,top,plan", [(16, 2, 14, 16, 512, (9, 12)), (8, 4, 12, 16, 256, (0, 4))])
@pytest.mark.parametrize("poison", [float("inf"), float("nan")])
@pytest.mark.parametrize("half", [False, True])
def test_hashgrid_backward_binned_keeps_non_finite_gradients(ops, dev, L, F, log2_T, base, top, plan, poison, half):
    """An inf / NaN entry of dL/d(features) -- an fp16 overflow under GradScaler -- reaches the table gradient through the binned
    scatter as it does through nvsf_hashgrid_bwd's memory atomics (GradScaler's found_inf check then skips the step): the bins'
    fixed-point image must not turn it into zeros or finite garbage.  Row-major and level-major gradients; a run-sum
    level, a per-row level and an atomic level are poisoned in turn; the other levels stay finite."""
    spec = _spec(ops, 3, L, F, log2_T, base, top)
    rng = np.random.default_rng(3)
    n_rays, T = 32, 128
    o = rng.random((n_rays, 1, 3)) * 0.5 + 0.1
    d = rng.standard_normal((n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = np.clip(o + d * np.linspace(0.0, 0.35, T).reshape(1, T, 1), 0.0, 1.0).reshape(-1, 3).astype(np.float32)
    M = x.shape[0]
    go = rng.standard_normal((M, L * F)).astype(np.float32)
    for level in sorted({0, plan[0], L - 1}):
        g = go.copy()
        g[1777, level * F + F - 1] = poison
        for level_major in (False, True):
            if level_major and half:
                continue
            gin = np.ascontiguousarray(g.reshape(M, L, F).transpose(1, 0, 2)) if level_major else g
            gin = _t(gin.astype(np.float16) if half else gin, dev)
            got = ops.hashgrid_backward(_t(x, dev), (0, 1, 2), spec, gin, fine_from=plan).cpu().numpy()
            for l in range(L):
                a, b = spec.offsets[l] * F, spec.offsets[l + 1] * F
                assert np.isfinite(got[a:b]).all() == (l != level), (level, l, level_major)
