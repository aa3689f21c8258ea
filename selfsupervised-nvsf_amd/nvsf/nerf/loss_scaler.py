"""Dynamic loss scaling of the reference's mixed-precision run, with the overflow decision available BEFORE every gradient is final.

The reference trains under `torch.cuda.amp.GradScaler(enabled=self.fp16)` (nvsf/nerf/trainer.py:119, 1332-1334: scale(loss).backward()
-> scaler.step(optimizer) -> scaler.update()).  GradScaler.step inspects EVERY gradient for inf / nan before it lets the optimiser
touch any parameter, so the step cannot begin until the last kernel of backward has finished -- here the table scatter of the
modality that went last, 1.3 ms on a side stream with nothing left to overlap it.

This class keeps GradScaler's rule (scale x 2 after `growth_interval` clean steps, x 0.5 and the step skipped on an overflow), its
constructor arguments and its `state_dict` keys (scale, growth_factor, backoff_factor, growth_interval, _growth_tracker: reference
checkpoints load, ours load there), and the same ATen kernels (`_amp_foreach_non_finite_check_and_unscale_`, `_amp_update_scale_`).
What changes is WHICH gradients are inspected: `found_inf(grads)` takes the list the caller passes.  RenderTrainStep takes the step's
decision from every gradient except the parameters still being scattered.  For a table fed by an fp32 input gradient (the static
field: DensityFn / RenderRaysFn) that is sufficient, not an approximation:

    a table gradient is  sum_s w_c(s) g[s]  with interpolation weights in [0, 1] and g = dL/d(features) = the input gradient of the
    density MLP, dX[s][k] = sum_o W0[o][k] dP0[s][o];  it is non-finite only if some dP0[s][o] is (finite fp16 operands cannot
    overflow an fp32 sum of 64 / 8 terms).  The same dP0[s][o] is an addend of the MLP's own weight gradient dW0[o][k] =
    sum_s dP0[s][o] x[s][k] for every k, and inf x anything is inf or nan -- so the density MLP's weight gradient, which IS
    inspected, is non-finite whenever a table gradient is (tests/test_train_step_gpu.py: an injected overflow skips every
    parameter, tables included).

A table that receives its feature gradient through an fp16 hand-over (the static hash of the space-time field: a finite fp32 dX above
65504 becomes inf in the cast) is not covered by that argument, so RenderTrainStep.step also inspects the deferred gradients -- on
the side stream behind their scatter: the deferred Adam pass skips on (early OR late), and an overflow found only there reaches
`update` one step later (tests/test_dynamic_gpu.py::test_overflow_only_in_a_deferred_table_gradient_never_reaches_the_table).
"""
import torch


class LossScaler:
    def __init__(self, device="cuda", init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._enabled = bool(enabled)
        self._device = torch.device(device)
        self._init_scale, self._growth_factor = float(init_scale), float(growth_factor)
        self._backoff_factor, self._growth_interval = float(backoff_factor), int(growth_interval)
        self._init_growth_tracker = 0
        self._scale = self._growth_tracker = self._one = None

    def is_enabled(self):
        return self._enabled

    def _lazy_init(self):
        if self._scale is None:
            self._scale = torch.full((), self._init_scale, dtype=torch.float32, device=self._device)
            self._growth_tracker = torch.full((), self._init_growth_tracker, dtype=torch.int32, device=self._device)

    def scale(self, loss):
        if not self._enabled:
            return loss
        self._lazy_init()
        return loss * self._scale.to(loss.device, non_blocking=True)

    def get_scale(self):
        """Host value (a device->host read: checkpoint / logging time only)."""
        if not self._enabled:
            return 1.0
        return self._init_scale if self._scale is None else float(self._scale)

    def scale_tensor(self):
        """A COPY of the current scale (device, fp32): what this step's optimiser pass unscales by -- `update` changes the
        scaler's own tensor in place, possibly while a deferred optimiser pass of this step is still queued on another stream."""
        self._lazy_init()
        return self._scale.clone()

    def found_inf(self, grads):
        """1.0 if any of `grads` holds an inf / nan, else 0.0 (device fp32 [1]); the gradients are left as they are (inv_scale 1)."""
        self._lazy_init()
        found = torch.zeros(1, dtype=torch.float32, device=self._device)
        grads = [g for g in grads if g is not None and g.numel()]
        if grads:
            if self._one is None:
                self._one = torch.ones((), dtype=torch.float32, device=self._device)
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found, self._one)
        return found

    def step(self, optimizer, found_inf=None, params=None):
        """GradScaler.step for the callers that have every gradient final: inspects them (unless `found_inf` is given) and runs the
        optimiser -- FusedAdam unscales inside its update pass and skips on the device; any other optimiser gets unscaled gradients
        and is skipped on the host.  Returns the found_inf tensor (None when disabled) for `update`."""
        if not self._enabled:
            optimizer.step()
            return None
        self._lazy_init()
        if found_inf is None:
            found_inf = self.found_inf([p.grad for g in optimizer.param_groups for p in g["params"]])
        if getattr(optimizer, "_step_supports_amp_scaling", False):
            optimizer.grad_scale, optimizer.found_inf = self.scale_tensor(), found_inf
            try:
                optimizer.step(params=params) if params is not None else optimizer.step()
            finally:
                del optimizer.grad_scale, optimizer.found_inf
        else:
            inv = 1.0 / self.get_scale()
            if float(found_inf) == 0.0:
                for g in optimizer.param_groups:
                    for p in g["params"]:
                        if p.grad is not None:
                            p.grad.mul_(inv)
                optimizer.step()
        return found_inf

    def hold_growth(self):
        """The next `update` cannot complete a growth interval (device-side clamp of the tracker, no read-back): for a step whose
        overflow decision is not final yet (RenderTrainStep's deferred table check)."""
        if self._enabled and self._growth_interval >= 2:
            self._lazy_init()
            self._growth_tracker.clamp_(max=self._growth_interval - 2)

    def update(self, found_inf):
        if not self._enabled:
            return
        self._lazy_init()
        torch._amp_update_scale_(self._scale, self._growth_tracker, found_inf, self._growth_factor, self._backoff_factor, self._growth_interval)

    # ---- torch.amp.GradScaler's checkpoint format (nvsf/nerf/utils.py:622-648 stores scaler.state_dict()) ---------------------------
    def state_dict(self):
        if not self._enabled:
            return {}
        return {"scale": self.get_scale(), "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval,
                "_growth_tracker": self._init_growth_tracker if self._growth_tracker is None else int(self._growth_tracker)}

    def load_state_dict(self, state):
        if not self._enabled:
            return
        if len(state) == 0:
            raise RuntimeError("The source state dict is empty, possibly because it was saved from a disabled instance of GradScaler.")
        self._init_scale = float(state["scale"])
        self._growth_factor, self._backoff_factor = float(state["growth_factor"]), float(state["backoff_factor"])
        self._growth_interval = int(state["growth_interval"])
        self._init_growth_tracker = int(state["_growth_tracker"])
        if self._scale is not None:
            self._scale.fill_(self._init_scale)
            self._growth_tracker.fill_(self._init_growth_tracker)
