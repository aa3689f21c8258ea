"""Prints the entry-point index of INTEGRATION.md from include/nvsf_hip.h: every C-ABI entry point, the reference interface its
header comment cites, and the product modules that call it.  `python tools/abi_index.py` -> markdown table on stdout."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r"[A-Za-z_/0-9]+\.(?:py|cu|h|cpp):[0-9]+(?:-[0-9]+)?(?:,\s*[0-9]+(?:-[0-9]+)?)*")
BLOCK = re.compile(r"/\*(.*?)\*/\s*((?:(?:int|size_t|const char\*)\s+nvsf_[A-Za-z0-9_]+\s*\([^;]*\);\s*)+)", re.S)


def callers(name):
    found = []
    pkg = os.path.join(ROOT, "selfsupervised-nvsf_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "_hip.py":
                if name in open(os.path.join(base, f)).read():
                    found.append(os.path.relpath(os.path.join(base, f), pkg))
    return sorted(found)


def rows():
    text = open(os.path.join(ROOT, "include", "nvsf_hip.h")).read()
    last = []
    for m in BLOCK.finditer(text):
        comment = " ".join(l.strip().lstrip("*").strip() for l in m.group(1).strip().split("\n"))
        cites = CITE.findall(comment)
        if cites:
            last = cites
        for name in re.findall(r"(nvsf_[A-Za-z0-9_]+)\s*\(", m.group(2)):
            yield name, cites or last, callers(name)


if __name__ == "__main__":
    print("| entry point | reference interface it stands for | called from |")
    print("|---|---|---|")
    for name, cites, who in rows():
        special = {"nvsf_version": "library version string", "nvsf_build_digest": "none: digest of the sources the mapped library was built from (`build.py`, `bench.py`)", "nvsf_hashgrid_bwd_binned_ws_bytes": "workspace size of `nvsf_hashgrid_bwd_binned`", "nvsf_test_variant": "none: test-only choice of a reference formulation",
                   "nvsf_scratch_pool_stats": "none: diagnostics of the stream-ordered pool `nvsf_march_rays_train` (`raymarching.h:27-44`: no scratch argument) borrows from"}
        ref = special.get(name) or ("; ".join(f"`{c}`" for c in cites[:3]) or "—")
        print(f"| `{name}` | {ref}{'' if cites or name in special else ' (derivative of the entry above)'} | {', '.join(f'`{w}`' for w in who) or 'C clients only'} |")
