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


def csrc_digest(unit="fused_field.hip"):
    """sha1 over one translation unit of the kernels -- by default the one that holds the render kernels of the headline
    benchmark -- every csrc header, and the compile flags: identifies the build a profile was taken on (bench.py reports PMC
    traffic only from a profile whose digest matches the running sources)."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f == unit or f.endswith(".h"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def csrc_digest_all():
    """sha1 over EVERY translation unit, every header and the flags: identifies the library as a whole (bench.py checks that all
    ranks run the same build; profiles of the secondary legs are tagged with it)."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or f.endswith(".h"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


OBJDIR = os.path.join(LIBDIR, "obj")


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.abspath(__file__)]


def _obj(src):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _compile(args):
    hipcc, src, report = args
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", _obj(src)]
    if report:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    proc = subprocess.run(cmd, capture_output=True, text=True)
    return src, proc.returncode, proc.stderr


def build(force=False, report=False, verbose=True):
    """One translation unit per .hip file, compiled in parallel into lib/obj/*.o (only the ones older than their source or any
    csrc header), then linked into libnvsf_hip.so.  Kernels live in anonymous namespaces: no cross-file device linking."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    newest_header = max(os.path.getmtime(h) for h in _headers())
    srcs = sources()
    todo = [s for s in srcs if force or report or not os.path.exists(_obj(s))
            or os.path.getmtime(_obj(s)) < max(os.path.getmtime(s), newest_header)]
    stale_objs = [o for o in os.listdir(OBJDIR) if o.endswith(".o") and os.path.join(OBJDIR, o) not in {_obj(s) for s in srcs}]
    for o in stale_objs:
        os.remove(os.path.join(OBJDIR, o))
    if not todo and not stale_objs and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(_obj(s)) for s in srcs):
        return LIB
    errors, logs = [], []
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        for src, rc, err in pool.map(_compile, [(hipcc, s, report) for s in todo]):
            logs.append(err)
            if rc != 0:
                errors.append((src, err))
    if errors:
        for src, err in errors:
            sys.stderr.write(f"---- {src}\n{err[-6000:]}\n")
        raise RuntimeError(f"hipcc failed building {[os.path.basename(s) for s, _ in errors]}")
    link = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden"] + [_obj(s) for s in srcs] + ["-o", LIB + ".tmp"],
                          capture_output=True, text=True)
    if link.returncode != 0:
        sys.stderr.write(link.stderr[-8000:])
        raise RuntimeError(f"hipcc failed ({link.returncode}) linking {LIB}")
    os.replace(LIB + ".tmp", LIB)
    text = "\n".join(logs)
    warn = [l for l in text.splitlines() if "warning:" in l]
    if verbose and warn:
        print("\n".join(warn[:40]))
    if report:
        _print_report(text)
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
