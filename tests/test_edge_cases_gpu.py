"""Empty, single-element and ragged inputs through every layer of the boundary (C ABI -> Python operators -> model.render):
nothing may fault or hang, shapes follow the reference's conventions, and tiny batches still match the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_raymarching_operators_on_empty_inputs(dev):
    from nvsf.nerf.raymarching import raymarching as rm
    e3 = torch.zeros(0, 3, device=dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=torch.float32, device=dev)
    n, f = rm.near_far_from_aabb(e3, e3, aabb, 0.2)
    assert n.shape == (0,) and f.shape == (0,)
    assert rm.sph_from_ray(e3, e3, 3.0).shape == (0, 2)
    assert rm.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=dev)).shape == (0,)
    assert rm.morton3D_invert(torch.zeros(0, dtype=torch.int32, device=dev)).shape == (0, 3)
    bits = torch.zeros(2 * 128 ** 3 // 8, dtype=torch.uint8, device=dev)
    x, d, dl, rays = rm.march_rays_train(e3, e3, 2.0, bits, 2, 128, n, f, None, -1, False, 128, True, 0, 64)
    assert rays.shape == (0, 3) and x.shape[1] == 3 and d.shape == x.shape and dl.shape[1] == 2
    # a batch whose rays all miss the box / cross empty space: zero samples, composite gives zeros
    o = torch.tensor([[10.0, 10.0, 10.0], [0.0, 0.0, 0.0]], device=dev)
    dd = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], device=dev)
    n2, f2 = rm.near_far_from_aabb(o, dd, aabb, 0.2)
    x, d, dl, rays = rm.march_rays_train(o, dd, 2.0, bits, 2, 128, n2, f2, None, -1, False, 128, True, 0, 64)
    assert int(rays[:, 2].sum()) == 0
    ws, dp, img = rm.composite_rays_train(torch.zeros(x.shape[0], device=dev), torch.zeros(x.shape[0], 3, device=dev), dl, rays)
    assert not ws.any() and not dp.any() and not img.any()


def test_field_operators_on_empty_and_single_rows(dev):
    import tinycudann as tcnn
    from nvsf import field_ops as ops
    import oracle_lib as O
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
                            "per_level_scale": 1.3819}).to(dev)
    net = tcnn.Network(32, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                                "n_hidden_layers": 1}).to(dev)
    for M in (0, 1, 17):
        x = torch.rand(M, 3, device=dev)
        f = enc(x)
        h = net(f)
        assert f.shape == (M, 32) and f.dtype == torch.float16 and h.shape == (M, 16)
        if M:
            ref = O.hashgrid_fwd(x.cpu().numpy(), (0, 1, 2), enc.params.detach().cpu().numpy().astype(np.float16), enc.spec)
            assert np.array_equal(f.detach().cpu().numpy().view(np.uint16), ref.view(np.uint16))
            (h.sum()).backward()
            assert enc.params.grad is not None and torch.isfinite(enc.params.grad).all()
    assert ops.freq_encode(torch.rand(0, 3, device=dev)).shape == (0, 72)
    assert ops.sh4_encode(torch.rand(0, 3, device=dev)).shape == (0, 16)


@pytest.mark.parametrize("N,T", [(1, 1), (1, 5), (3, 17), (65, 64)])
@pytest.mark.parametrize("lidar", [True, False])
def test_render_tiny_and_ragged_batches_match_the_oracle(dev, N, T, lidar):
    """model.render on batches far below one workgroup / one MFMA tile (T not a multiple of 16, a single ray, a single
    sample) through the fused path, against the CPU oracle composition."""
    import oracle_lib as O
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(3)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH)
    with torch.no_grad():
        for e in (m.hash_encoder_lidar, m.hash_encoder_camera):
            e.params.normal_(0.0, 0.5)
    m = m.to(dev).eval()
    rng = np.random.default_rng(N * 100 + T)
    o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
    with torch.no_grad():
        out = m.render(torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None], torch.tensor([[0.5]], device=dev),
                       cal_lidar_color=lidar, num_steps=T)
    enc = m.hash_encoder_lidar if lidar else m.hash_encoder_camera
    if lidar:
        nears, fars = np.full(N, m.min_near_lidar, np.float32), np.full(N, m.lidar_max_depth, np.float32)
    else:
        nears, fars = O.near_far_from_aabb(o, d, np.array([-S.BOUND] * 3 + [S.BOUND] * 3, np.float32), m.min_near)
    f16 = lambda net: net.params.detach().cpu().numpy().astype(np.float16)
    ref = O.render_static(o, d, nears, fars, torch.linspace(0.0, 1.0, T).numpy(), None, float(S.BOUND),
                          enc.params.detach().cpu().numpy().astype(np.float16), enc.spec, f16(m.sigma_net), lidar,
                          f16(m.raydrop_net) if lidar else f16(m.color_net), f16(m.intensity_net) if lidar else None, np.ones(3, np.float32))
    sfx = "_lidar" if lidar else ""
    assert out["image" + sfx].shape == (1, N, 2 if lidar else 3) and out["weights"].shape == (N, T)
    np.testing.assert_allclose(out["image" + sfx][0].cpu().numpy(), ref["image"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["depth" + sfx][0].cpu().numpy(), ref["depth"], atol=1e-4, rtol=0)


def test_occupancy_render_with_an_empty_grid_and_single_ray(dev):
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev)
    m = m.enable_occupancy_grid().to(dev).eval()
    m.set_density_grid(torch.zeros(m.cascade, m.grid_size ** 3, device=dev), thresh=0.5)  # nothing occupied
    rng = np.random.default_rng(0)
    o, d = S.camera_rays(1, rng)
    with torch.no_grad():
        out = m.render(torch.from_numpy(o).to(dev)[None], torch.from_numpy(d).to(dev)[None], torch.tensor([[0.5]], device=dev), max_steps=64)
    assert torch.allclose(out["image"], torch.ones_like(out["image"])) and not out["depth"].any()  # background only
