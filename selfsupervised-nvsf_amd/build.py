"""Builds libnvsf_hip.so (every HIP kernel + the C ABI) for gfx950 with hipcc, in-tree.

    python selfsupervised-nvsf_amd/build.py [--report] [--force]

The library is a plain C-ABI shared object (no torch / pybind linkage); it cross-compiles on a machine
without a GPU.  `--report` prints per-kernel VGPR / spill / occupancy figures from
-Rpass-analysis=kernel-resource-usage.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnvsf_hip.so")
ARCH = "gfx950"

# -ffp-contract=off: every fp32 op is individually rounded (fma only where the source says fmaf), which is
# what makes the marcher / hash-grid index arithmetic agree bit for bit with the CPU oracle.
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has a unified register file), which removes the
# v_accvgpr_read per accumulator register that otherwise precedes every activation / epilogue (measured: -66 VALU
# per 28 MFMAs in the heads kernel).
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fvisibility=hidden", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form", f"--offload-arch={ARCH}"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, report=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not report and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + sources() + ["-o", LIB + ".tmp"]
    if report:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stderr[-8000:])
        raise RuntimeError(f"hipcc failed ({proc.returncode}) building {LIB}")
    os.replace(LIB + ".tmp", LIB)
    warn = [l for l in proc.stderr.splitlines() if "warning:" in l]
    if verbose and warn:
        print("\n".join(warn[:40]))
    if report:
        _print_report(proc.stderr)
    return LIB


def _print_report(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"),
                         ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    try:
        import subprocess as sp
        names = sp.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    except Exception:
        names = [r["name"] for r in rows]
    print(f"{'kernel':70s} vgpr agpr sgpr spill scratch occ lds")
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = n.split("(")[0]
        print(f"{n[:70]:70s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} {r.get('spill', 0):5d} "
              f"{r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('lds', 0)}")


if __name__ == "__main__":
    build(force="--force" in sys.argv, report="--report" in sys.argv)
    print(LIB)
