"""The C ABI used from plain C (no Python, no torch): tests/abi_c/abi_smoke.c is compiled against include/nvsf_hip.h, linked
with libnvsf_hip.so (+ the CPU oracle as checker) and run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_client_of_the_abi(tmp_path):
    cc = shutil.which("gcc") or "cc"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    lib_dir = os.path.join(ROOT, "selfsupervised-nvsf_amd", "lib")
    assert os.path.exists(os.path.join(lib_dir, "libnvsf_hip.so")), "build the library first (python __graft_entry__.py)"
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    exe = str(tmp_path / "abi_smoke")
    # a C compiler, the HIP runtime API header for hipMalloc / hipMemcpy, and the two shared objects: nothing else
    subprocess.run([cc, "-std=gnu11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_c", "abi_smoke.c"), "-o", exe, "-L", lib_dir, "-lnvsf_hip", "-L", os.path.join(ROOT, "oracle"),
                    "-loracle_raymarching", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([lib_dir, os.path.join(ROOT, "oracle"), os.path.join(rocm, "lib"),
                                                            os.environ.get("LD_LIBRARY_PATH", "")]))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke OK" in out.stdout
