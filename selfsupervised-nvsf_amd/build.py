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
DIGEST_MARK = b"NVSF_CSRC_DIGEST_ALL="  # followed by the 16 hex digits of csrc_digest_all() inside the shared object (csrc/version.hip)


def _obj(src):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _unit_digest(src, extra=""):
    """What an object file was compiled from: flags, its translation unit, every header (+ for version.hip the digests it embeds)."""
    import hashlib
    h = hashlib.sha1((" ".join(FLAGS) + extra).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(src, "rb").read())
    return h.hexdigest()


def embedded_digest(path=LIB):
    """The csrc_digest_all() of the sources a built libnvsf_hip.so was compiled from, read from the file's bytes (no dlopen); None
    when the file is missing or predates the marker.  `nvsf_build_digest()` returns the same string from the mapped library."""
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    i = blob.find(DIGEST_MARK)
    if i < 0:
        return None
    d = blob[i + len(DIGEST_MARK): i + len(DIGEST_MARK) + 16]
    return d.decode() if re.fullmatch(rb"[0-9a-f]{16}", d) else None


def _compile(args):
    hipcc, src, report, defines, digest = args
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + defines + ["-c", src, "-o", _obj(src)]
    if report:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode == 0:
        with open(_obj(src) + ".sha1", "w") as f:
            f.write(digest)
    return src, proc.returncode, proc.stderr


def build(force=False, report=False, verbose=True):
    """One translation unit per .hip file, compiled in parallel into lib/obj/*.o, then linked into libnvsf_hip.so.  Staleness is
    decided by CONTENT, not by mtime (a prebuilt library pushed to another box is newer than its sources whatever
    they say): an object is rebuilt when the sha1 of (flags, its source, every header) differs from the one stored beside it, and
    the library is relinked when the digest embedded in it (version.hip is compiled with -DNVSF_CSRC_DIGEST_ALL=csrc_digest_all())
    is not the digest of the sources on disk.  Kernels live in anonymous namespaces: no cross-file device linking."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sources()
    want_all, want_render = csrc_digest_all(), csrc_digest()
    if not force and not report and embedded_digest() == want_all:
        return LIB  # the library on disk was built from exactly these sources and flags (whether or not its objects came along)
    defines = {s: [] for s in srcs}
    digests = {}
    for s in srcs:
        extra = ""
        if os.path.basename(s) == "version.hip":
            defines[s] = [f'-DNVSF_CSRC_DIGEST_ALL="{want_all}"', f'-DNVSF_CSRC_DIGEST_RENDER="{want_render}"']
            extra = want_all + want_render
        digests[s] = _unit_digest(s, extra)

    def stored(s):
        try:
            return open(_obj(s) + ".sha1").read().strip()
        except OSError:
            return None
    todo = [s for s in srcs if force or report or not os.path.exists(_obj(s)) or stored(s) != digests[s]]
    keep = {_obj(s) for s in srcs} | {_obj(s) + ".sha1" for s in srcs}
    stale_objs = [o for o in os.listdir(OBJDIR) if os.path.join(OBJDIR, o) not in keep]
    for o in stale_objs:
        os.remove(os.path.join(OBJDIR, o))
    if not todo and not stale_objs and embedded_digest() == want_all:
        return LIB
    errors, logs = [], []
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        for src, rc, err in pool.map(_compile, [(hipcc, s, report, defines[s], digests[s]) for s in todo]):
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
    if embedded_digest(LIB + ".tmp") != want_all:
        raise RuntimeError("the linked library does not carry the digest of the sources it was built from")
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
