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
