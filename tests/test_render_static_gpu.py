"""GPU parity of the uniform-sampling render of the static hash field (BASELINE config 2 shape) against the CPU
oracle composition (tests/oracle_lib.render_static), for both execution paths of NeRFNetworkStatic:
the fused kernels (no_grad) and the operator path (autograd).  Tolerance: 1e-4 abs on composited depth /
image / weights (north_star), tighter where the arithmetic allows.
"""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(dev, table_std=None, seed=0):
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf import synthetic as S
    torch.manual_seed(seed)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                          num_frames=S.NUM_FRAMES)
    if table_std is not None:  # non-trivial density field (a freshly initialised table gives sigma == 1 everywhere)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
                enc.params.copy_(torch.randn(enc.params.shape, generator=g) * table_std)
            m.sigma_net.params.mul_(2.0)
    return m.to(dev).eval()


def _oracle(m, o, d, lidar, T, noise=None, bg=(1.0, 1.0, 1.0)):
    from nvsf import synthetic as S
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    table = enc.params.detach().cpu().numpy().astype(np.float16)
    f16 = lambda net: net.params.detach().cpu().numpy().astype(np.float16)
    N = o.shape[0]
    if lidar:
        nears, fars = np.full(N, m.min_near_lidar, np.float32), np.full(N, m.lidar_max_depth, np.float32)
    else:
        nears, fars = O.near_far_from_aabb(o, d, np.array([-S.BOUND] * 3 + [S.BOUND] * 3, np.float32), m.min_near)
    lin = torch.linspace(0.0, 1.0, T).numpy()
    return O.render_static(o, d, nears, fars, lin, noise, float(S.BOUND), table, enc.spec, f16(m.sigma_net), lidar,
                           f16(m.raydrop_net) if lidar else f16(m.color_net), f16(m.intensity_net) if lidar else None,
                           np.array(bg, np.float32), k_scale=m._k_scale())


@pytest.mark.parametrize("lidar", [True, False])
@pytest.mark.parametrize("table_std", [None, 0.1])
def test_fused_render_matches_oracle(dev, lidar, table_std):
    from nvsf import synthetic as S
    m = _model(dev, table_std)
    rng = np.random.default_rng(3)
    N, T = 200, 96
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    ref = _oracle(m, o, d, lidar, T)
    with torch.no_grad():
        out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, num_steps=T)
    sfx = "_lidar" if lidar else ""
    assert out["image" + sfx].shape == (1, N, 2 if lidar else 3) and out["depth" + sfx].shape == (1, N)
    assert np.array_equal(out["z_vals"].cpu().numpy(), ref["z_vals"])
    np.testing.assert_allclose(out["weights"].cpu().numpy(), ref["weights"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["weights_sum" + sfx].cpu().numpy(), ref["weights_sum"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].cpu().numpy(), ref["depth"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["image" + sfx][0].cpu().numpy(), ref["image"], atol=1e-4, rtol=0)


def test_fused_density_per_sample(dev):
    """sigma / geo of the fused density kernel, sample by sample (fp16 logits: identical except for rare
    1-ulp flips from the MFMA summation order)."""
    from nvsf import field_ops as ops, synthetic as S
    m = _model(dev, 0.1)
    rng = np.random.default_rng(5)
    N, T = 64, 80
    o, d = S.lidar_rays(N, rng)
    ref = _oracle(m, o, d, True, T)
    nears = torch.full((N,), float(m.min_near_lidar), device=dev)
    fars = torch.full((N,), float(m.lidar_max_depth), device=dev)
    enc = m.hash_encoder_lidar
    z, sig, geo = ops.density_uniform(_t(o, dev), _t(d, dev), nears, fars, T, m._aabb_host, float(m.bound), enc.table_f16(), enc.spec,
                                      m.sigma_net.weights_f16())
    assert np.array_equal(z.cpu().numpy(), ref["z_vals"])
    geo = geo.cpu().numpy()
    assert np.all(geo[..., 15] == 1.0)
    g_ref = ref["geo"].astype(np.float32)
    np.testing.assert_allclose(geo[..., :15].astype(np.float32), g_ref, atol=2e-3 * np.abs(g_ref).max(), rtol=0)
    assert (geo[..., :15] == ref["geo"]).mean() > 0.99  # fp16 features: identical except rare 1-ulp rounding flips
    s, sr = sig.cpu().numpy(), ref["sigmas"]
    np.testing.assert_allclose(s, sr, rtol=2e-3, atol=0)
    assert (np.abs(s / sr - 1) < 1e-5).mean() > 0.99  # fp32 logit + expf: ulp-level agreement for almost every sample


@pytest.mark.parametrize("lidar", [True, False])
def test_operator_path_equals_fused_path(dev, lidar):
    from nvsf import synthetic as S
    m = _model(dev, 0.1)
    rng = np.random.default_rng(7)
    N, T = 150, 64
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    args = (_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev))
    with torch.no_grad():
        fused = m.render(*args, cal_lidar_color=lidar, num_steps=T)
    with torch.enable_grad():
        oper = m.render(*args, cal_lidar_color=lidar, num_steps=T)
    sfx = "_lidar" if lidar else ""
    for k in ("z_vals", "weights", "depth" + sfx, "image" + sfx, "weights_sum" + sfx):
        np.testing.assert_allclose(oper[k].detach().cpu().numpy(), fused[k].cpu().numpy(), atol=5e-5, rtol=0, err_msg=k)
    loss = oper["image" + sfx].sum() + oper["depth" + sfx].sum()
    loss.backward()
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    assert enc.params.grad is not None and torch.isfinite(enc.params.grad).all() and enc.params.grad.abs().sum() > 0
    assert m.sigma_net.params.grad.abs().sum() > 0


def test_perturb_and_staged_and_bg(dev):
    from nvsf import synthetic as S
    m = _model(dev, 0.1)
    rng = np.random.default_rng(9)
    N, T = 300, 32
    o, d = S.camera_rays(N, rng)
    args = (_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev))
    with torch.no_grad():
        full = m.render(*args, num_steps=T, bg_color=torch.tensor([0.2, 0.4, 0.6], device=dev))
        staged = m.render(*args, staged=True, max_ray_batch=128, num_steps=T, bg_color=torch.tensor([0.2, 0.4, 0.6], device=dev))
        assert set(staged.keys()) == {"depth", "image"}
        assert torch.equal(staged["image"], full["image"]) and torch.equal(staged["depth"], full["depth"])
        ref = _oracle(m, o, d, False, T, bg=(0.2, 0.4, 0.6))
        np.testing.assert_allclose(full["image"][0].cpu().numpy(), ref["image"], atol=1e-4, rtol=0)
        torch.manual_seed(0)
        p = m.render(*args, num_steps=T, perturb=True)
        torch.manual_seed(0)
        noise = torch.rand(N, T, device=dev)
        ref_p = _oracle(m, o, d, False, T, noise=noise.cpu().numpy())
        assert np.array_equal(p["z_vals"].cpu().numpy(), ref_p["z_vals"])
        np.testing.assert_allclose(p["image"][0].cpu().numpy(), ref_p["image"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("lidar", [True, False])
def test_headline_shape_matches_oracle(dev, lidar):
    """BASELINE config 2 at its full size -- 4096 rays x 768 samples per modality, L16 F2 T2^19 tables with a non-trivial
    density field -- through the path `model.render` selects by itself at that size (N * T >= 2^18: the XCD-sliced encode pass
    + tail for the camera batch, the one-launch gather kernel for the LiDAR batch; exactly what bench.py times).  The oracle
    renders 64 of the 4096 rays (rays are independent); 1e-4 abs on depth / image, z_vals bit-exact."""
    from nvsf import field_ops as ops, synthetic as S
    m = _model(dev, 0.1)
    rng = np.random.default_rng(21)
    N, T, K = 4096, 768, 64
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    ray_length = float(m.lidar_max_depth - m.min_near_lidar) if lidar else 2.0 * float(m.bound)
    assert ops.prefer_sliced(enc.spec, N, T, ray_length, float(m.bound)) == (not lidar)  # the bench's path choice
    with torch.no_grad():
        out = m.render(_t(o, dev)[None], _t(d, dev)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, num_steps=T)
    pick = np.sort(rng.choice(N, K, replace=False))
    ref = _oracle(m, o[pick], d[pick], lidar, T)
    sfx = "_lidar" if lidar else ""
    assert np.array_equal(out["z_vals"].cpu().numpy()[pick], ref["z_vals"])
    assert float(ref["weights_sum"].max()) > 0.5 and float(ref["weights_sum"].min()) < 0.999  # a non-trivial field
    np.testing.assert_allclose(out["weights"].cpu().numpy()[pick], ref["weights"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["weights_sum" + sfx].cpu().numpy()[pick], ref["weights_sum"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].cpu().numpy()[pick], ref["depth"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["image" + sfx][0].cpu().numpy()[pick], ref["image"], atol=1e-4, rtol=0)
