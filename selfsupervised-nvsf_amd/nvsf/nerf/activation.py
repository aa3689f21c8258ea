"""`trunc_exp`, the density activation (ref: /root/reference/nvsf/nerf/activation.py:6-20): sigma = exp(h) evaluated in
fp32; towards h the gradient is multiplied by exp(clamp(h, -15, 15)), so a large logit cannot overflow it.

exp is monotonic, hence exp(clamp(h, -15, 15)) == clamp(exp(h), exp(-15), exp(15)): the backward re-uses the saved OUTPUT
and needs no second exp kernel over the [M] logits (the fixture of the reference's own function pins values and gradients
bit for bit, tests/test_oracle_cpu.py::test_trunc_exp_golden)."""
import torch

_LO = float(torch.exp(torch.tensor(-15.0, dtype=torch.float32)))
_HI = float(torch.exp(torch.tensor(15.0, dtype=torch.float32)))


class TruncatedExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        sigma = torch.exp(logits.float())  # fp32 whatever the autocast state (the reference: custom_fwd(cast_inputs=float32))
        ctx.save_for_backward(sigma)
        ctx.in_dtype = logits.dtype
        return sigma

    @staticmethod
    def backward(ctx, grad_sigma):
        (sigma,) = ctx.saved_tensors
        return (grad_sigma * sigma.clamp(_LO, _HI)).to(ctx.in_dtype)


def trunc_exp(logits):
    return TruncatedExp.apply(logits)
