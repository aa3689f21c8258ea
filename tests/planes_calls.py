"""Shared by the GPU tests of the K-planes texel scatter: the C-ABI call of nvsf_planes_multi_bwd in the shape the training step issues it."""
import torch


def multi_bwd_call(enc, x, flow, g_wide, times, dev, grad=None):
    """nvsf_planes_multi_bwd as PlanesMultiFn(blend=True).backward issues it: static + three time-plane evaluations (x, x + flow[:, :3],
    x + flow[:, 3:]) sharing ONE gradient slice scaled 0.5 / 0.25 / 0.25; returns the texel gradient."""
    import ctypes
    from nvsf import _hip
    M, n = x.shape[0], 4
    gp = torch.zeros_like(enc.planes_cl) if grad is None else grad
    g_s, g_d = g_wide[:, 0:32], g_wide[:, 32:64]
    offs = (ctypes.c_void_p * n)(None, None, flow.data_ptr(), flow.data_ptr())
    g_ptrs = (ctypes.c_void_p * n)(g_s.data_ptr(), g_d.data_ptr(), g_d.data_ptr(), g_d.data_ptr())
    _hip.call("nvsf_planes_multi_bwd", _hip.ptr_rows(x), x.stride(0), M, _hip.ptr(enc.planes_cl.detach()), 4, 8, _hip.host_u32(enc._res_host), n,
              _hip.host_i32([0, 1, 1, 1]), offs, _hip.host_u32([0, 0, flow.stride(0), flow.stride(0)]), _hip.host_u32([0, 0, 0, 3]),
              _hip.host_f32(times), g_ptrs, _hip.host_u32([g_wide.stride(0)] * n), _hip.host_f32([1.0, 0.5, 0.25, 0.25]), _hip.ptr(gp),
              None, None, None)
    torch.cuda.synchronize()
    return gp
