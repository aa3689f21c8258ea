"""Adam of the training step as one HIP pass per parameter tensor (csrc/adam.hip).

Update rule, hyper-parameter names and the `state_dict` layout ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}},
'param_groups': [...]}) are those of `torch.optim.Adam` as the reference constructs it (nvsf/scripts/main_nvsf.py:350-352:
betas (0.9, 0.99), eps 1e-15, no weight decay, no amsgrad), so optimiser states move between the two.  Under a GradScaler
the step takes the scaler's `grad_scale` / `found_inf` tensors (`_step_supports_amp_scaling`): gradients are unscaled inside
the update pass and an overflowing step is skipped ON THE DEVICE -- the default path reads the overflow flag back to the
host every step (`GradScaler._maybe_opt_step`: `found_inf.item()`), which stalls the launch queue of the next step.
"""
import torch

from nvsf import _hip


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0 or lr < 0.0:
            raise ValueError("invalid Adam hyper-parameters")
        # the remaining keys of torch.optim.Adam's groups, at the values this optimiser implements: a state_dict written here
        # loads into torch.optim.Adam and back
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
        betas0 = self.param_groups[0]["betas"]
        if any(tuple(g["betas"]) != tuple(betas0) for g in self.param_groups):
            raise ValueError("FusedAdam keeps one step counter: every group must use the same betas")
        self._dev_state = None  # device [4]: step, 1 - b1^step, sqrt(1 - b2^step), skip

    def _device_state(self, device):
        if self._dev_state is None or self._dev_state.device != device:
            old = 0.0 if self._dev_state is None else float(self._dev_state[0])
            self._dev_state = torch.tensor([old, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        return self._dev_state

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        first = next((p for g in self.param_groups for p in g["params"] if p.grad is not None), None)
        if first is None:
            return loss
        if not first.is_cuda:
            raise _hip.NvsfHipError("FusedAdam runs on a HIP device (there is no CPU fallback)")
        st = self._device_state(first.device)
        beta1, beta2 = self.param_groups[0]["betas"]
        _hip.call("nvsf_adam_prepare", _hip.ptr(st), None if found_inf is None else _hip.ptr(found_inf.float()), float(beta1), float(beta2))
        scale_ptr = None if grad_scale is None else _hip.ptr(grad_scale.float())
        for group in self.param_groups:
            lr, eps = float(group["lr"]), float(group["eps"])
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise _hip.NvsfHipError("FusedAdam: plain Adam only (no weight decay / amsgrad / maximize)")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise _hip.NvsfHipError("FusedAdam: fp32 contiguous parameters and gradients")
                state = self.state[p]
                if "exp_avg" not in state:
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _hip.call("nvsf_adam_update", p.data_ptr(), grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                          p.numel(), lr, float(beta1), float(beta2), eps, _hip.ptr(st), scale_ptr)
                # the kernel wrote through the raw pointer: tell autograd (and every cache keyed on `_version`, e.g. the fp16
                # copies of tables and weights the forward kernels read) that the parameter changed
                torch.autograd.graph.increment_version(p)
        return loss

    # ---- torch.optim.Adam-compatible state dicts ---------------------------------------------------------------------------
    def state_dict(self):
        step = 0.0 if self._dev_state is None else float(self._dev_state[0])  # one device read, at checkpoint time only
        out = super().state_dict()
        out["state"] = {k: dict(st, step=torch.tensor(step, dtype=torch.float32)) for k, st in out["state"].items()}
        return out

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(s.pop("step")) for s in self.state.values() if "step" in s]
        step = max(steps) if steps else 0.0
        dev = next((p.device for g in self.param_groups for p in g["params"] if p.is_cuda), None)
        self._dev_state = None if dev is None else torch.tensor([step, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        if dev is None and step:
            raise _hip.NvsfHipError("FusedAdam state needs a HIP device")
