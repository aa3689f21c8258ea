"""Test-only selection of reference formulations.

Every operator of this package has ONE production form.  Where a second formulation of the same operator is kept -- the one the
production form is pinned against in tests/ (the first, straightforward kernel; the unfused operator chain) -- it is selected
here, explicitly and for the duration of a `with` block:

    with nvsf.testing.variant(hashgrid_bwd="atomic"):
        ...                                   # nvsf_hashgrid_bwd launches the one-thread-per-(row, level) kernel

Nothing reads the process environment on the call path: the kernel launchers read an int of the library's variant table
(nvsf_test_variant, include/nvsf_hip.h), the Python operators a dict of this module.  Production code never calls this module.
"""
import contextlib

# kernel-side choices (libnvsf_hip.so: csrc/common.h NvsfVariantKey): name -> {value name -> int}
_NATIVE = {
    "march": {"wave": 0, "thread": 1, "serial": 2},
    "planes_fwd": {"runs": 0, "sample": 1},
    "planes_bwd": {"runs": 0, "atomic": 1, "global": 2},  # multi entry: 0 time planes through the LDS image, 1 / 2 every evaluation as run sums into global atomics
    "hashgrid_fwd": {"auto": 0, "generic": 1},
    "hashgrid_bwd": {"corners": 0, "atomic": 1},
    "hash4d_bwd": {"lds": 0, "runs": 1},
    "slice_plan": {"balanced": 0, "home": 1},
    "render_tail": {"two": 0, "one": 1},
    "mlp_bwd": {"auto": 0, "staged": 1, "wave": 2},
    "level_kinds": {"compiled": 0, "runtime": 1},
    "march_skew": {"off": 0, **{f"queue{q}": q + 1 for q in range(8)}},
}
# operator-side choices (Python): name -> allowed values, the first one is the production form
_PYTHON = {
    "heads_input": ("prefix", "rows"),          # field_ops.heads: per-ray prefix rows / assembled [M, in_cols] input rows
    "density_sliced": (None, False, True),      # field_ops.prefer_sliced: host-side choice / forced off / forced on
    "hash4d_train": ("fused", "slices"),        # hash_field.HashGrid4D training forward: fused kernels / per-slice encoder calls
    "dynamic_fused": (True, False),             # network_dynamic no-grad feature path: fused launches / operator calls
    "density_tail_train": ("fused", "chain"),   # network_dynamic.density with autograd: DensityTailFn / torch blend + cat + MLP
    "density_fn": ("fused", "chain"),           # network_static.density: DensityFn / encoder -> MLP -> trunc_exp modules
    "planes_train": ("fused", "separate"),       # network_dynamic training features: ONE K-planes node for a density query / one PlanesFn per evaluation
    "density_grad": ("composed", "matrix"),      # field_ops._density_backward: logit gradient formed inside the MLP backward / nvsf_sigma_geo_bwd pass
    "hash4d_scatter": ("side", "main"),          # hash_field.HashDynFn.backward inside a training step: on the step's side stream into the gradient sink / on the main stream through autograd
    "table_scatter": ("binned", "atomic"),      # field_ops._bin_from: binned fine levels where they pay / every level through nvsf_hashgrid_bwd
}
_state = {}


def get(name):
    """Current value of an operator-side choice (the production value unless a `variant` block is active)."""
    return _state.get(name, _PYTHON[name][0])


@contextlib.contextmanager
def variant(**choices):
    from nvsf import _hip
    lib = _hip.load()
    undo_native, undo_python = [], []
    try:
        for name, value in choices.items():
            if name in _NATIVE:
                code = _NATIVE[name][value]
                old = lib.nvsf_test_variant(name.encode(), code)
                if old < 0:
                    raise _hip.NvsfHipError(f"nvsf_test_variant({name!r}) rejected")
                undo_native.append((name, old))
            elif name in _PYTHON:
                if value not in _PYTHON[name]:
                    raise ValueError(f"{name}: one of {_PYTHON[name]}")
                undo_python.append((name, _state.get(name, _PYTHON[name][0])))
                _state[name] = value
            else:
                raise KeyError(name)
        yield
    finally:
        for name, old in reversed(undo_native):
            lib.nvsf_test_variant(name.encode(), old)
        for name, old in reversed(undo_python):
            _state[name] = old
