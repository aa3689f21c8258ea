"""The fused render / training forward for the OTHER 32-feature grid shape: 8 levels x 4 features, the reference's default hash
grid (/root/reference/nvsf/scripts/main_nvsf.py:45-52: 512 -> 32768, T = 2^19).  Its encode pass (k_encode_sliced_f4: one level per
XCD group, two lanes per sample) writes the feature planes the L16 F2 pass writes, column pair by column pair, so the streaming
tails, their TRAIN forms and the backward are the code BASELINE config 2 runs.  Checked here: the planes against the stand-alone
encoder bit for bit (dense and hashed levels, ragged T, perturbation), the render against the CPU oracle composition
(tests/oracle_lib.render_static with the grid's own spec), the one-node training forward against the no-grad render bit for bit and
against the operator chain's gradients, and that a model of this shape takes the fused paths.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_render_static_gpu import _oracle, _t
from test_density_sliced_gpu import _batch

SHAPES = {
    "reference-default": dict(base_resolution=512, max_resolution=32768, log2_hashmap_size=19),  # every level hashed
    "dense-and-hashed": dict(base_resolution=8, max_resolution=1024, log2_hashmap_size=14),      # levels 0-2 dense, 3-7 hashed
}


def _model(dev, shape, table_std=0.1, seed=0):
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf import synthetic as S
    torch.manual_seed(seed)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, num_frames=S.NUM_FRAMES,
                          n_levels_hash=8, n_features_per_level_hash=4, **SHAPES[shape])
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
            enc.params.copy_(torch.randn(enc.params.shape, generator=g) * table_std)
        m.sigma_net.params.mul_(2.0)
    m = m.to(dev).eval()
    spec = m.hash_encoder_camera.spec
    assert (spec.L, spec.F) == (8, 4)
    return m


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("lidar", [True, False])
@pytest.mark.parametrize("N,T,noise", [(64, 128, False), (37, 100, True), (1, 16, False), (300, 768, True), (5, 7, False)])
def test_feature_planes_equal_the_encoder_rows(dev, shape, lidar, N, T, noise):
    """Sliced training forward of the density (encode pass + streaming MLP pass in TRAIN form): positions, z and the [M, 32]
    feature rows equal uniform_samples / hashgrid_forward bit for bit; sigma / geometry rows equal the stand-alone MLP kernel's to
    fp16 rounding of its order of additions."""
    from nvsf import field_ops as ops
    m = _model(dev, shape)
    rng = np.random.default_rng(11)
    o, d, nears, fars = _batch(m, dev, lidar, N, rng)
    nz = torch.rand(N, T, device=dev) if noise else None
    enc, net = (m.hash_encoder_lidar if lidar else m.hash_encoder_camera), m.sigma_net
    assert ops.sliced_only(enc.spec) and ops.prefer_sliced(enc.spec, N, T, 1.0, float(m.bound)) and ops.render_uniform_eligible(enc.spec, N * T)
    z, sigma, geo16, x01, feat, h32 = ops.density_uniform_train_forward(o, d, nears, fars, T, m._aabb_host, float(m.bound), nz, enc.table_f16(), enc.spec,
                                                                        net.weights_f16(), True)
    zz, xyz = ops.uniform_samples(o, d, nears, fars, T, m.aabb_train, nz)
    assert torch.equal(zz, z)
    assert torch.equal((xyz.view(-1, 3) + m.bound) / (2 * m.bound), x01)
    rows = ops.hashgrid_forward(x01, (0, 1, 2), enc.table_f16(), enc.spec)
    assert torch.equal(rows, feat)
    h = ops.mlp_forward(feat, net.weights_f16(), net.spec)
    assert float((h - h32).abs().max()) <= 2e-4 * float(h.abs().max())
    assert torch.equal(geo16[:, :15], h32[:, 1:].half()) and bool((geo16[:, 15] == 1).all())
    assert torch.allclose(sigma, torch.exp(h32[:, 0]), rtol=2e-6, atol=0)
    # the no-grad sliced density (same two passes without the TRAIN outputs)
    z2, s2, g2 = ops.density_uniform(o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, net.weights_f16(), nz, sliced=True)
    assert torch.equal(z2, z) and torch.equal(s2.view(-1), sigma) and torch.equal(g2.view(-1, 16), geo16)


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("lidar", [True, False])
def test_fused_render_matches_oracle(dev, shape, lidar):
    from nvsf import synthetic as S, field_ops as ops
    m = _model(dev, shape)
    rng = np.random.default_rng(3)
    N, T = 200, 96
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    ref = _oracle(m, o, d, lidar, T)
    calls = []
    real = ops.render_uniform
    ops.render_uniform = lambda *a, **k: (calls.append(k.get("sliced")), real(*a, **k))[1]
    try:
        with torch.no_grad():
            out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, num_steps=T)
    finally:
        ops.render_uniform = real
    assert calls == [True]  # the fused render ran, in its level-sliced form
    sfx = "_lidar" if lidar else ""
    assert np.array_equal(out["z_vals"].cpu().numpy(), ref["z_vals"])
    np.testing.assert_allclose(out["weights"].cpu().numpy(), ref["weights"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["weights_sum" + sfx].cpu().numpy(), ref["weights_sum"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].cpu().numpy(), ref["depth"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["image" + sfx][0].cpu().numpy(), ref["image"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("lidar", [True, False])
@pytest.mark.parametrize("N,T,noise", [(64, 128, True), (9, 96, False), (300, 768, False)])
def test_training_render_equals_the_evaluation_render(dev, lidar, N, T, noise):
    """nvsf_render_uniform_train_fwd for L8 F4: the five render outputs equal nvsf_render_uniform_fwd's bit for bit, what it keeps
    for the backward equals the density training forward's (positions, feature rows, sigma, geometry rows)."""
    from nvsf import field_ops as ops
    m = _model(dev, "reference-default")
    rng = np.random.default_rng(5)
    o, d, nears, fars = _batch(m, dev, lidar, N, rng)
    nz = torch.rand(N, T, device=dev) if noise else None
    enc, net = (m.hash_encoder_lidar if lidar else m.hash_encoder_camera), m.sigma_net
    if lidar:
        ha, hb = m.raydrop_net.weights_f16(), m.intensity_net.weights_f16()
    else:
        ha, hb = m.color_net.weights_f16(), None
    bg = None if lidar else np.array([1.0, 1.0, 1.0], np.float32)
    ev = ops.render_uniform(o, d, nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec, net.weights_f16(), lidar, ha, hb,
                            m._k_scale(), bg, nz, sliced=True)
    tr = ops.render_uniform_train_forward(o, d, nears, fars, T, m._aabb_host, float(m.bound), nz, enc.table_f16(), enc.spec, net.weights_f16(),
                                          lidar, ha, hb, m._k_scale(), bg, ops.W_THRESH, True)
    for a, b, name in zip(ev, tr[:5], ("z_vals", "weights", "weights_sum", "depth", "image")):
        assert torch.equal(a, b), name
    z, sigma, geo16, x01, feat, h32 = ops.density_uniform_train_forward(o, d, nears, fars, T, m._aabb_host, float(m.bound), nz, enc.table_f16(), enc.spec,
                                                                        net.weights_f16(), True)
    assert torch.equal(tr[5], x01) and torch.equal(tr[6], feat) and torch.equal(tr[7], geo16) and torch.equal(tr[8], sigma)


@pytest.mark.parametrize("lidar", [True, False])
def test_one_node_training_forward_against_the_operator_chain(dev, lidar):
    """A model of the reference-default grid shape takes ops.RenderRaysFn under autograd; outputs and gradients agree with the operator
    chain (uniform_samples -> encoder -> MLP -> compositor -> heads, every operator its own node) as for the config-2 shape."""
    from nvsf import field_ops as ops
    from nvsf import synthetic as S
    N, T = 160, 64
    rng = np.random.default_rng(7)
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    args = (_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev))
    sfx = "_lidar" if lidar else ""
    res = {}
    for mode in ("chain", "node"):
        m = _model(dev, "reference-default", table_std=0.3)
        m.fused_train_forward = m.fused_train_render = mode == "node"
        taken = []
        real = ops.RenderRaysFn.apply
        ops.RenderRaysFn.apply = lambda *a, **k: (taken.append(1), real(*a, **k))[1]
        try:
            with torch.enable_grad():
                out = m.render(*args, cal_lidar_color=lidar, num_steps=T)
                w = torch.linspace(0.5, 1.5, N, device=dev)
                loss = (out["image" + sfx][0] * w[:, None]).sum() + (out["depth" + sfx][0] * w).sum() + out["weights_sum" + sfx].sum()
                loss.backward()
        finally:
            ops.RenderRaysFn.apply = real
        assert len(taken) == (1 if mode == "node" else 0)
        torch.cuda.synchronize()
        res[mode] = ({k: out[k].detach().clone() for k in ("z_vals", "weights", "depth" + sfx, "image" + sfx, "weights_sum" + sfx)},
                     {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.numel() and p.grad is not None})
    for k, v in res["chain"][0].items():
        assert float((v - res["node"][0][k]).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max())), k
    assert set(res["chain"][1]) == set(res["node"][1]) and len(res["node"][1]) == (4 if lidar else 3)
    for n, a in res["chain"][1].items():
        b = res["node"][1][n]
        err = float((a - b).abs().max()) / float(a.abs().max())
        assert float(a.abs().max()) > 0 and err <= 1e-4, (n, err)
