"""GPU parity of the Chamfer-distance kernels against the CPU oracle (oracle/chamfer_oracle.c)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle():
    import oracle_lib as O
    O.build()
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_chamfer.so"))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("B,n,m", [(1, 4096, 4096), (2, 1000, 37), (3, 5, 3000), (1, 1, 1)])
def test_chamfer_forward_backward(dev, B, n, m):
    from nvsf.nerf.chamfer3D.dist_chamfer_3D import chamfer_3DDist
    lib = _oracle()
    rng = np.random.default_rng(B * 100 + n)
    a = rng.standard_normal((B, n, 3)).astype(np.float32)
    b = rng.standard_normal((B, m, 3)).astype(np.float32)
    if n > 10 and m > 10:
        b[:, 5] = b[:, 3]  # duplicate target: ties must resolve to the lower index
        a[:, 7] = b[:, 5]
    d1, d2 = np.empty((B, n), np.float32), np.empty((B, m), np.float32)
    i1, i2 = np.empty((B, n), np.int32), np.empty((B, m), np.int32)
    U = ctypes.c_uint32
    lib.oracle_chamfer_forward(_p(a), _p(b), U(B), U(n), U(m), _p(d1), _p(d2), _p(i1), _p(i2))
    ta, tb = torch.from_numpy(a).to(dev).requires_grad_(), torch.from_numpy(b).to(dev).requires_grad_()
    gd1, gd2, gi1, gi2 = chamfer_3DDist()(ta, tb)
    assert np.array_equal(gi1.cpu().numpy(), i1) and np.array_equal(gi2.cpu().numpy(), i2)
    assert np.array_equal(gd1.detach().cpu().numpy(), d1) and np.array_equal(gd2.detach().cpu().numpy(), d2)
    if n > 10 and m > 10:
        assert (i1[:, 7] == 3).all()
    g1, g2 = rng.standard_normal((B, n)).astype(np.float32), rng.standard_normal((B, m)).astype(np.float32)
    ga, gb = np.zeros_like(a), np.zeros_like(b)
    lib.oracle_chamfer_backward(_p(a), _p(b), U(B), U(n), U(m), _p(g1), _p(g2), _p(i1), _p(i2), _p(ga), _p(gb))
    ((gd1 * torch.from_numpy(g1).to(dev)).sum() + (gd2 * torch.from_numpy(g2).to(dev)).sum()).backward()
    # fp32 sums in the arrival order of the atomics: a point that is the nearest neighbour of ~600 others (n = 5, m = 3000) adds ~600
    # terms, so the last bits depend on the run (observed: 1.3e-5 relative on an entry of magnitude 129)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ga, atol=2e-5, rtol=5e-5)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), gb, atol=2e-5, rtol=5e-5)
    # one cloud without a gradient (the training loss: predicted cloud against the measured one, and the other way round): the other
    # cloud's gradient is the same, formed without the zero fill / the atomics of the cloud that needs none
    for first in (True, False):
        ta2 = torch.from_numpy(a).to(dev).requires_grad_(first)
        tb2 = torch.from_numpy(b).to(dev).requires_grad_(not first)
        e1, e2, _, _ = chamfer_3DDist()(ta2, tb2)
        ((e1 * torch.from_numpy(g1).to(dev)).sum() + (e2 * torch.from_numpy(g2).to(dev)).sum()).backward()
        if first:
            assert tb2.grad is None
            np.testing.assert_allclose(ta2.grad.cpu().numpy(), ga, atol=2e-5, rtol=5e-5)
        else:
            assert ta2.grad is None
            np.testing.assert_allclose(tb2.grad.cpu().numpy(), gb, atol=2e-5, rtol=5e-5)


def test_points_meter_cd_and_fscore(dev):
    """PointsMeter = the reference's CD / F-score meter (nvsf/lib/error_matrices.py:12-26, 299-356; pano_to_lidar:
    nvsf/lib/convert.py:221-291) on the HIP chamfer kernel: against a numpy restatement of those formulas (brute-force nearest
    neighbours in float64), incl. zero-range pixels and a second frame in the running mean."""
    from nvsf.nerf.train_step import PointsMeter, fscore, pano_to_lidar
    rng = np.random.default_rng(3)
    H, W, scale, K, Kh = 16, 96, 0.0125, (10.0, 40.0), (180.0, 360.0)

    def cloud(pano):
        i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
        beta = -(i - W / 2) / W * Kh[1] / 180 * np.pi
        alpha = (K[0] - j / H * K[1]) / 180 * np.pi
        dirs = np.stack([np.cos(alpha) * np.cos(beta), np.cos(alpha) * np.sin(beta), np.sin(alpha)], -1)
        return (dirs * pano.reshape(H, W, 1))[pano != 0.0]  # convert.py:262-266: zero-range pixels give no point

    meter = PointsMeter(scale, K, Kh)
    want = []
    for k in range(2):
        gt = (rng.uniform(2.0, 60.0, (H, W)) * scale).astype(np.float32)
        gt[rng.random((H, W)) < 0.3] = 0.0  # dropped rays on the measured side ...
        pred = (gt + rng.normal(0, 0.2 * scale, (H, W)) * (rng.random((H, W)) < 0.6)).astype(np.float32)
        pred[rng.random((H, W)) < 0.1] = 0.0  # ... and (other) pixels gated off on the predicted side: clouds of different sizes
        assert (pred == 0).sum() != (gt == 0).sum()
        a, b = cloud(pred / np.float32(scale)).astype(np.float64), cloud(gt / np.float32(scale)).astype(np.float64)
        np.testing.assert_allclose(pano_to_lidar(torch.from_numpy(pred / np.float32(scale)).to(dev), K, Kh).cpu().numpy(), a, atol=2e-5)
        d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        d1, d2 = d.min(1), d.min(0)
        p1, p2 = (d1 < 0.05).mean(), (d2 < 0.05).mean()
        want.append([d1.mean() + d2.mean(), 2 * p1 * p2 / (p1 + p2) if p1 + p2 > 0 else 0.0])
        meter.update(torch.from_numpy(pred).to(dev)[None], torch.from_numpy(gt).to(dev)[None])
    got = meter.measure()
    np.testing.assert_allclose(got, np.mean(want, 0), rtol=2e-4, atol=1e-6)
    assert 0.0 < got[1] < 1.0 and "Points_error" in meter.report()
    z = torch.ones(1, 5, device=dev)
    assert float(fscore(z, z, 0.5)[0]) == 0.0  # nothing below the threshold: 0 / 0 -> 0, as the reference sets it
