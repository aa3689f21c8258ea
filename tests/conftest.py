import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "selfsupervised-nvsf_amd")
for p in (PKG, ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


# Collection order of the `-m gpu` files ("What's weak" 2).  pytest collects files alphabetically, which put the full-size /
# multi-stream / statistical files (test_config4_*, test_config5_train_*) in front of the deterministic oracle-parity files; under the
# driver's `-x` one noisy self-comparison then hid every parity row behind it.  The order below is by what a test proves:
#   tier 0  deterministic parity against the oracle / the reference's fixtures, operator by operator (b, f1, g1);
#   tier 1  whole-path renders against the reference's own fixtures, reference-shaped drivers, formats (a10, a11, a13-a18, f3, f4);
#   tier 2  training steps at test sizes (f2, e);
#   tier 3  full-size steps: binned plans, side streams, fp32-atomic sums with statistical bounds (run last).
# Files not listed keep their alphabetical place inside tier 1.  CPU files are not touched (tier 1, alphabetical, as before).
_GPU_FILE_ORDER = (
    ("test_abi_c_gpu.py", "test_raymarching_gpu.py", "test_edge_cases_gpu.py", "test_field_gpu.py", "test_mlp_bwd_gpu.py",
     "test_reference_fixtures_gpu.py", "test_raygen_gpu.py", "test_occupancy_gpu.py"),
    ("test_render_static_gpu.py", "test_config5_gpu.py", "test_dynamic_gpu.py", "test_density_sliced_gpu.py", "test_l8f4_fused_gpu.py",
     "test_chamfer_gpu.py", "test_formats_gpu.py"),
    ("test_train_step_gpu.py", "test_shipped_config_losses_gpu.py", "test_train_step_ranks_gpu.py"),
    ("test_config4_full_size_gpu.py", "test_config5_train_full_size_gpu.py"),
)


def gpu_file_rank(basename):
    """(tier, position) of a test file in the order above; unlisted files: tier 1, after the listed ones, alphabetical."""
    for tier, files in enumerate(_GPU_FILE_ORDER):
        if basename in files:
            return (tier, files.index(basename), "")
    return (1, len(_GPU_FILE_ORDER[1]), basename)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=lambda it: gpu_file_rank(os.path.basename(str(it.fspath))))  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads libnvsf_hip.so."""
    sys.path.insert(0, PKG)
    import build as nvsf_build  # selfsupervised-nvsf_amd/build.py
    nvsf_build.build(verbose=False)
    from nvsf import _hip
    return _hip.load()


@pytest.fixture(scope="session")
def dev(hip_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a HIP device: there is no CPU fallback for the HIP path")
    return torch.device("cuda:0")


@pytest.fixture
def variants(hip_lib):
    """Test-only selection of reference formulations (nvsf.testing.variant) with set / clear calls; whatever is still selected when
    the test ends is put back."""
    import contextlib
    from nvsf import testing

    class Variants:
        def __init__(self):
            self._active = {}

        def set(self, **choices):
            for name, value in choices.items():
                self.clear(name)
                stack = contextlib.ExitStack()
                stack.enter_context(testing.variant(**{name: value}))
                self._active[name] = stack

        def clear(self, name):
            stack = self._active.pop(name, None)
            if stack is not None:
                stack.close()

    v = Variants()
    yield v
    for name in list(v._active):
        v.clear(name)
