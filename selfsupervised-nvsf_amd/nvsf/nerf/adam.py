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
    """Step counters: torch.optim.Adam keeps one `step` per parameter and advances it only when that parameter has a gradient
    (a LiDAR-only step leaves the camera tables' counters alone).  Here parameters that have so far been updated in exactly the
    same steps share one device-side counter row [step, 1 - b1^step, sqrt(1 - b2^step), skip]; when a step updates only part of
    a row's members those members fork into a row of their own (a 16-byte device copy).  All parameters always active -- the
    usual case -- is one row and one `nvsf_adam_prepare` launch per step; the bias corrections every parameter sees are those of
    torch.optim.Adam, and `state_dict()` / `load_state_dict()` carry the per-parameter steps losslessly."""
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
            raise ValueError("FusedAdam shares step-counter rows between groups: every group must use the same betas")
        self._rows = None      # device [R, 4]: step, 1 - b1^step, sqrt(1 - b2^step), skip
        self._row_of = {}      # parameter -> row index
        self._n_rows = 0
        self.ema = None        # optional (shadow tensors by parameter, one_minus_decay): EMA folded into the update pass
        self.half_caches = {}  # parameter -> the fp16 cache of the module that owns it (tinycudann._HalfCache): the update pass writes
                               # the fp16 copy the forward kernels read and marks the cache fresh (no cast pass per parameter and step)

    _MAX_ROWS = 64

    def _row_table(self, device):
        if self._rows is None or self._rows.device != device:
            old = self._rows
            self._rows = torch.zeros(self._MAX_ROWS, 4, dtype=torch.float32, device=device)
            if old is not None:
                self._rows.copy_(old)
        return self._rows

    def _assign_rows(self, active, device):
        """Row of every active parameter for this step; forks rows whose members are only partly active.  Host logic only."""
        rows = self._row_table(device)
        members = {}
        for p, r in self._row_of.items():
            members.setdefault(r, []).append(p)
        active_set = set(active)
        fresh = [p for p in active if p not in self._row_of]
        for r, ps in members.items():
            act = [p for p in ps if p in active_set]
            if act and len(act) < len(ps):  # fork: the active members continue on a copy of the row
                new = self._new_row()
                rows[new].copy_(rows[r])
                for p in act:
                    self._row_of[p] = new
        if fresh:
            # a parameter seen for the first time has step 0: it may share the row of others that are also at step 0 only if
            # they are all active now, which the fork rule above already guarantees for later steps -- give them a new row
            new = self._new_row()
            rows[new].zero_()
            for p in fresh:
                self._row_of[p] = new
        return sorted({self._row_of[p] for p in active})

    def _new_row(self):
        used = set(self._row_of.values())
        for r in range(self._MAX_ROWS):
            if r not in used:
                return r
        raise _hip.NvsfHipError("FusedAdam: more than 64 distinct update histories among the parameters")

    @torch.no_grad()
    def step(self, closure=None, params=None):
        """`params`: restrict the update to these parameters (a set / list); the others are left for a later call of the same
        step -- RenderTrainStep updates everything whose gradient is final, then, on the stream of the last table scatter, the table
        that scatter writes.  The two calls of one step see the same `grad_scale` / `found_inf`; step counts stay per parameter."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        only = None if params is None else set(params)
        active = [p for g in self.param_groups for p in g["params"] if p.grad is not None and (only is None or p in only)]
        if not active:
            return loss
        first = active[0]
        if not first.is_cuda:
            raise _hip.NvsfHipError("FusedAdam runs on a HIP device (there is no CPU fallback)")
        beta1, beta2 = self.param_groups[0]["betas"]
        rows = self._row_table(first.device)
        found = None if found_inf is None else _hip.ptr(found_inf.float())
        for r in self._assign_rows(active, first.device):
            _hip.call("nvsf_adam_prepare", rows[r].data_ptr(), found, float(beta1), float(beta2))
        scale_ptr = None if grad_scale is None else _hip.ptr(grad_scale.float())
        shadows, omd = (self.ema if self.ema is not None else ({}, 0.0))
        for group in self.param_groups:
            lr, eps = float(group["lr"]), float(group["eps"])
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise _hip.NvsfHipError("FusedAdam: plain Adam only (no weight decay / amsgrad / maximize)")
            for p in group["params"]:
                if p.grad is None or (only is not None and p not in only):
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise _hip.NvsfHipError("FusedAdam: fp32 contiguous parameters and gradients")
                state = self.state[p]
                if "exp_avg" not in state:
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                shadow = shadows.get(p)
                cache = self.half_caches.get(p)
                half = cache.writable(p) if cache is not None else None
                _hip.call("nvsf_adam_update", p.data_ptr(), grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                          p.numel(), lr, float(beta1), float(beta2), eps, rows[self._row_of[p]].data_ptr(), scale_ptr,
                          None if shadow is None else shadow.data_ptr(), float(omd), None if half is None else half.data_ptr())
                # the kernel wrote through the raw pointer: tell autograd (and every cache keyed on `_version`, e.g. the fp16
                # copies of tables and weights the forward kernels read) that the parameter changed
                torch.autograd.graph.increment_version(p)
                if half is not None:
                    cache.mark_fresh(p)  # ... except the cache this very pass has just brought up to date
        return loss

    # ---- torch.optim.Adam-compatible state dicts ---------------------------------------------------------------------------
    def state_dict(self):
        steps = None if self._rows is None else self._rows[:, 0].cpu()  # one device read, at checkpoint time only
        out = super().state_dict()
        index = {}
        i = 0
        for g in self.param_groups:  # torch's packing order: parameter ids count up through the groups
            for p in g["params"]:
                index[i] = p
                i += 1
        state = {}
        for k, st in out["state"].items():
            p = index[k]
            step = float(steps[self._row_of[p]]) if (steps is not None and p in self._row_of) else 0.0
            state[k] = dict(st, step=torch.tensor(step, dtype=torch.float32))
        out["state"] = state
        return out

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        by_step = {}
        for p, st in self.state.items():
            if "step" in st:
                by_step.setdefault(float(st.pop("step")), []).append(p)
        self._rows, self._row_of = None, {}
        if not by_step:
            return
        dev = next((p.device for g in self.param_groups for p in g["params"] if p.is_cuda), None)
        if dev is None:
            raise _hip.NvsfHipError("FusedAdam state needs a HIP device")
        if len(by_step) > self._MAX_ROWS:
            raise _hip.NvsfHipError("FusedAdam: more than 64 distinct step counts in the loaded state")
        table = torch.zeros(self._MAX_ROWS, 4, dtype=torch.float32)
        for r, (step, ps) in enumerate(sorted(by_step.items())):
            table[r, 0] = step
            for p in ps:
                self._row_of[p] = r
        self._rows = table.to(dev)
