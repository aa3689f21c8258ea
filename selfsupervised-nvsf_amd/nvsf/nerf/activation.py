"""`trunc_exp` (ref: /root/reference/nvsf/nerf/activation.py:6-20): exp in fp32 forward; the backward
multiplies by exp(clamp(x, -15, 15)) so large logits cannot overflow the gradient."""
import torch
from torch.autograd import Function


class _TruncExp(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, -15.0, 15.0))


trunc_exp = _TruncExp.apply
