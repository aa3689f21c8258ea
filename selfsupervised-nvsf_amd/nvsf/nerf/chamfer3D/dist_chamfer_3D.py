"""`nvsf.nerf.chamfer3D.dist_chamfer_3D` for MI355X: `chamfer_3DDist()(xyz1, xyz2) -> dist1, dist2, idx1, idx2`
(squared nearest-neighbour distances in both directions + int32 indices, differentiable in both clouds), the
interface of /root/reference/nvsf/nerf/chamfer3D/dist_chamfer_3D.py:42-97, on the HIP kernels of csrc/chamfer.hip
instead of a JIT-compiled CUDA extension."""
import torch
from torch import nn
from torch.autograd import Function

from nvsf import _hip


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        B, n, dim = xyz1.size()
        assert dim == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        _, m, dim = xyz2.size()
        assert dim == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        xyz1, xyz2 = xyz1.float().contiguous(), xyz2.float().contiguous()
        dev = xyz1.device
        dist1 = torch.empty(B, n, device=dev)
        dist2 = torch.empty(B, m, device=dev)
        idx1 = torch.empty(B, n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(B, m, dtype=torch.int32, device=dev)
        ws = torch.empty(B * max(n, m), dtype=torch.int64, device=dev)
        _hip.call("nvsf_chamfer_forward", _hip.ptr(xyz1), _hip.ptr(xyz2), B, n, m, _hip.ptr(dist1), _hip.ptr(dist2), _hip.ptr(idx1),
                  _hip.ptr(idx2), _hip.ptr(ws))
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = graddist1.float().contiguous(), graddist2.float().contiguous()
        # the first cloud's gradient is WRITTEN by the first direction's kernel (no zero fill); the second cloud's is formed only when
        # it is asked for (the training loss compares against the measured cloud, which has no gradient)
        need1, need2 = ctx.needs_input_grad
        if not (need1 or need2):
            return None, None
        gradxyz1 = torch.empty_like(xyz1) if need1 else None
        gradxyz2 = torch.zeros_like(xyz2) if need2 else None
        _hip.call("nvsf_chamfer_backward", _hip.ptr(xyz1), _hip.ptr(xyz2), B, n, m, _hip.ptr(g1), _hip.ptr(g2), _hip.ptr(idx1), _hip.ptr(idx2),
                  _hip.ptr(gradxyz1), _hip.ptr(gradxyz2))
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    def forward(self, input1, input2):
        return chamfer_3DFunction.apply(input1.contiguous(), input2.contiguous())
