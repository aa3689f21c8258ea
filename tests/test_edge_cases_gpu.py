"""Empty, ragged and rejected inputs of the entry points added in round 2 (loss terms, activation glue, the one-launch marcher),
called through the C ABI as a host binding would: a zero-sized call is a no-op that returns NVSF_OK without touching its pointers,
a missing pointer or an impossible shape is refused with NVSF_ERR_INVALID_ARG before anything is launched, and sizes that are not
multiples of a wave / a workgroup give the same numbers as the torch restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _raises(name, *args):
    from nvsf import _hip
    with pytest.raises(_hip.NvsfHipError, match="rejected arguments"):
        _hip.call(name, *args)


def test_zero_sized_calls_are_noops(dev):
    from nvsf import _hip
    P = _hip.ptr
    out = torch.full((3,), 7.0, device=dev)
    # sums over zero rays are zero (written), everything else is not touched -- null data pointers are fine
    _hip.call("nvsf_lidar_losses_fwd", None, None, None, None, None, None, 0, 1.0, 0.01, 0.1, 0.0, 1.0, P(out[0:1]), P(out[1:2]), P(out[2:3]),
              None, None, None)
    assert out.tolist() == [0.0, 0.0, 0.0]
    _hip.call("nvsf_lidar_losses_bwd", None, None, None, None, None, None, 0, 1.0, 0.01, 0.1, 0.0, 1.0, None, None, None, None, None, None, None)
    one = torch.full((1,), 5.0, device=dev)
    _hip.call("nvsf_mse_sum_fwd", None, None, 0, 1.0, P(one))
    assert float(one) == 0.0
    _hip.call("nvsf_mse_sum_bwd", None, None, 0, 1.0, None, None)
    _hip.call("nvsf_sigmoid_bwd", None, None, 0, None)
    _hip.call("nvsf_exp_col", None, 16, 0, 0, None)
    _hip.call("nvsf_march_rays_train_ws", None, None, None, 2.0, 0.0, 1024, 0, 2, 128, 0, None, None, None, None, None, None, None, None, None, 0, 0)
    assert _hip.march_ws_bytes(0) == 1024 and _hip.march_ws_bytes(1) == 1040 and _hip.march_ws_bytes(5) == 1056  # 8 queue heads x 128 B + 16 B per ticket of four rays


def test_rejected_arguments(dev):
    from nvsf import _hip
    P = _hip.ptr
    x = torch.zeros(64, device=dev)
    _raises("nvsf_lidar_losses_fwd", P(x), P(x), P(x), P(x), P(x), None, 8, 1.0, 0.01, 0.1, 0.0, 1.0, None, P(x), P(x), P(x), None, None)  # no sum
    _raises("nvsf_lidar_losses_fwd", P(x), P(x), P(x), P(x), P(x), None, 8, 1.0, 0.01, 0.1, 0.0, 1.0, P(x), P(x), P(x), P(x), P(x), None)  # one cloud
    _raises("nvsf_lidar_losses_fwd", P(x), P(x), P(x), P(x), P(x), None, 8, 1.0, 0.01, 0.1, 0.0, 1.0, P(x), P(x), P(x), P(x), P(x), P(x))  # no rays_d
    _raises("nvsf_mse_sum_fwd", P(x), None, 8, 1.0, P(x))
    _raises("nvsf_mse_sum_bwd", P(x), P(x), 8, 1.0, None, P(x))
    _raises("nvsf_sigmoid_bwd", P(x), None, 8, P(x))
    _raises("nvsf_exp_col", P(x), 4, 4, 8, P(x))  # column outside the row
    ws = torch.zeros(1024, dtype=torch.int64, device=dev)
    rays = torch.zeros(8, 3, dtype=torch.int32, device=dev)
    ctr = torch.zeros(2, dtype=torch.int32, device=dev)
    bits = torch.zeros(2 * 128 ** 3 // 8, dtype=torch.uint8, device=dev)
    args = lambda C, H, wsb, wp: (P(x), P(x), P(bits), 2.0, 0.0, 64, 8, C, H, 512, P(x), P(x), P(x), P(x), P(x), P(rays), P(ctr), P(x), wp, wsb, 0)
    assert _hip.march_ws_bytes(8) == 1056
    _raises("nvsf_march_rays_train_ws", *args(2, 128, 1048, P(ws)))       # scratch smaller than nvsf_march_rays_train_ws_bytes(8)
    _raises("nvsf_march_rays_train_ws", *args(2, 128, 8000, P(ws) + 4))   # not 8-byte aligned
    _raises("nvsf_march_rays_train_ws", *args(9, 128, 8192, P(ws)))       # more cascades than the operator is defined for
    _raises("nvsf_march_rays_train_ws", *args(2, 2048, 8192, P(ws)))      # H beyond the spread table
    assert ctr.tolist() == [0, 0]


@pytest.mark.parametrize("n", [1, 63, 65, 1000, 4097])
def test_loss_and_activation_glue_on_ragged_sizes(dev, n):
    """Sizes around the wave / workgroup edges against the torch expressions the kernels replace."""
    from nvsf import _hip
    P = _hip.ptr
    g = torch.Generator(device=dev).manual_seed(n)
    a, b = torch.rand(n, 3, device=dev, generator=g), torch.rand(n, 3, device=dev, generator=g)
    loss = torch.empty(1, device=dev)
    _hip.call("nvsf_mse_sum_fwd", P(a), P(b), 3 * n, 0.5, P(loss))
    ref = (0.5 * (a.double() - b.double()) ** 2).sum()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, float(ref))
    gl, ga = torch.full((1,), 2.0, device=dev), torch.empty_like(a)
    _hip.call("nvsf_mse_sum_bwd", P(a), P(b), 3 * n, 0.5, P(gl), P(ga))
    assert torch.allclose(ga, 2.0 * 0.5 * 2.0 * (a - b), atol=1e-6)
    out, go = torch.rand(n, device=dev, generator=g), torch.randn(n, device=dev, generator=g)
    gi = torch.empty(n, device=dev)
    _hip.call("nvsf_sigmoid_bwd", P(go), P(out), n, P(gi))
    assert torch.equal(gi, (go * (1 - out)) * out)
    h = torch.randn(n, 16, device=dev, generator=g)
    e = torch.empty(n, device=dev)
    _hip.call("nvsf_exp_col", P(h), 16, 5, n, P(e))
    assert torch.allclose(e, torch.exp(h[:, 5]), rtol=2e-6, atol=0)
    # LiDAR sums + masked range, no chamfer clouds
    img, dep = torch.rand(n, 2, device=dev, generator=g), torch.rand(n, device=dev, generator=g)
    rd = (torch.rand(n, device=dev, generator=g) > 0.3).float()
    gi_, gd_ = torch.rand(n, device=dev, generator=g), torch.rand(n, device=dev, generator=g)
    sums, pd = torch.empty(3, device=dev), torch.empty(n, device=dev)
    _hip.call("nvsf_lidar_losses_fwd", P(img), P(dep), P(rd), P(gi_), P(gd_), None, n, 1.0, 0.01, 0.1, 0.2, 1.0, P(sums[0:1]), P(sums[1:2]), P(sums[2:3]),
              P(pd), None, None)
    want = [((dep * rd).double() - (gd_ * rd).double()).abs().sum(), (0.01 * (img[:, 0].double() - rd.clamp(0.2, 0.8).double()) ** 2).sum(),
            (0.1 * ((img[:, 1] * rd).double() - (gi_ * rd).double()) ** 2).sum()]
    for k in range(3):
        assert abs(float(sums[k]) - float(want[k])) <= 2e-5 * max(1.0, float(want[k])), k
    assert torch.equal(pd, dep * rd)


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 4097, 3145728 + 5])
def test_mask_population(dev, n):
    """nvsf_count_nonzero_u8 (the population of `weights > 1e-4`) against torch, on sizes around its 16-byte pieces, for bool masks
    and for byte masks whose non-zero values are not 1; an unaligned view goes through a copy in the wrapper."""
    from nvsf import field_ops as ops
    g = torch.Generator(device=dev).manual_seed(n + 1)
    mask = torch.rand(n, device=dev, generator=g) > 0.7
    assert int(ops.count_true(mask)) == int(mask.sum())
    if n:
        bytes_ = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g) * mask.to(torch.uint8)
        assert int(ops.count_true(bytes_)) == int((bytes_ != 0).sum())
    if n > 3:
        assert int(ops.count_true(mask[3:])) == int(mask[3:].sum())
        assert int(ops.count_true(mask.reshape(1, -1)[:, ::2])) == int(mask[::2].sum())
