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
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ga, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), gb, atol=2e-5, rtol=1e-5)
