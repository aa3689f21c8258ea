"""Exponential moving average of the model weights as the reference's Trainer keeps it.

The reference wraps `torch_ema.ExponentialMovingAverage(model.parameters(), decay=opt.ema_decay)` (nvsf/nerf/trainer.py:112-114,
`--ema_decay` default 0.95: nvsf/scripts/main_nvsf.py:78), calls `update()` ONCE PER EPOCH after the step loop (trainer.py:1420-1421),
evaluates under `store() / copy_to() ... restore()` (trainer.py:1475-1477, 1843-1844) and writes `ema.state_dict()` into the
checkpoint (nvsf/nerf/utils.py:622-648).  This class has the same methods, the same update rule

    decay_t = min(decay, (1 + num_updates) / (10 + num_updates));   shadow -= (1 - decay_t) * (shadow - param)

and the same state_dict keys (decay, num_updates, shadow_params, collected_params); the update is one HIP streaming pass per
parameter tensor (csrc/adam.hip: nvsf_ema_update).  `attach(optimizer)` is the every-step variant: the shadow update then rides
in the FusedAdam pass that has the parameter in registers anyway (no extra launch, 8 B/parameter of extra traffic).
"""
import torch

from nvsf import _hip


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = float(decay)
        self.num_updates = 0 if use_num_updates else None
        # torch_ema keeps a shadow for EVERY parameter it is given (frozen ones included): a reference checkpoint's `shadow_params`
        # has one tensor per model parameter, and load_state_dict below checks that count.  Frozen parameters never move, so only
        # their update is skipped.
        self._params = list(parameters)
        self.shadow_params = [p.detach().clone() for p in self._params]
        self.collected_params = None
        self._optimizer = None
        # called before anything here reads or writes the parameters / shadows: RenderTrainStep leaves the optimiser pass of the last
        # table on a side stream (with the shadow update folded in) and sets this to its sync()
        self.before_access = None

    def _settle(self):
        if self.before_access is not None:
            self.before_access()

    def _decay_now(self):
        """torch_ema: the count is incremented first, then decay = min(decay, (1 + n) / (10 + n))."""
        if self.num_updates is None:
            return self.decay
        self.num_updates += 1
        return min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))

    @torch.no_grad()
    def update(self):
        self._settle()
        one_minus_decay = 1.0 - self._decay_now()
        for s, p in zip(self.shadow_params, self._params):
            if not p.requires_grad or p.numel() == 0:
                continue
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _hip.NvsfHipError("EMA: fp32 contiguous parameters")
            _hip.call("nvsf_ema_update", _hip.ptr(s), _hip.ptr(p.detach()), p.numel(), float(one_minus_decay))

    # ---- every-step form, folded into the optimiser pass ----------------------------------------------------------------------
    def attach(self, optimizer):
        """The shadow update of every following `optimizer.step()` happens inside nvsf_adam_update.  Call `before_step()` ahead
        of each step (it advances num_updates / the warm-up decay exactly like `update()` would)."""
        self._optimizer = optimizer
        return self

    def before_step(self):
        if self._optimizer is None:
            raise _hip.NvsfHipError("EMA.before_step(): attach(optimizer) first")
        self._optimizer.ema = ({p: s for p, s in zip(self._params, self.shadow_params) if p.requires_grad}, 1.0 - self._decay_now())

    # ---- evaluation under the averaged weights (trainer.py:1475-1477, 1843-1844) --------------------------------------------
    @torch.no_grad()
    def copy_to(self):
        self._settle()
        for s, p in zip(self.shadow_params, self._params):
            p.copy_(s)  # in-place: bumps the version the fp16 weight / table caches are keyed on

    @torch.no_grad()
    def store(self):
        self._settle()
        self.collected_params = [p.detach().clone() for p in self._params]

    @torch.no_grad()
    def restore(self):
        if self.collected_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        self._settle()
        for c, p in zip(self.collected_params, self._params):
            p.copy_(c)
        self.collected_params = None

    @torch.no_grad()
    def reset_to_parameters(self):
        """Starts the average again from the current weights (a checkpoint whose `ema` entry could not be restored)."""
        self._settle()
        for s, p in zip(self.shadow_params, self._params):
            s.copy_(p.detach())
        if self.num_updates is not None:
            self.num_updates = 0
        self.collected_params = None

    def state_dict(self):
        self._settle()
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                "collected_params": self.collected_params}

    def load_state_dict(self, state):
        self._settle()
        self.decay = float(state["decay"])
        self.num_updates = state["num_updates"]
        if len(state["shadow_params"]) != len(self._params):
            raise ValueError("shadow_params: wrong number of tensors")
        with torch.no_grad():
            for s, new in zip(self.shadow_params, state["shadow_params"]):
                s.copy_(new.to(s.device, s.dtype))
        c = state.get("collected_params")
        self.collected_params = None if c is None else [t.to(p.device, p.dtype).clone() for t, p in zip(c, self._params)]
