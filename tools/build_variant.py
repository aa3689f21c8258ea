"""Experiment helper: a copy of the library with ONE translation unit compiled with extra -D flags.

    python tools/build_variant.py fused_field.hip lib_exp/variant.so -DSOME_SWITCH=1   # a switch the translation unit reads under #ifdef for the experiment

The production library is built first (build.py); the variant re-uses its objects.  For tools/ab_run.sh / ab_headline.sh.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
import build as B

unit, out, defines = sys.argv[1], os.path.join(ROOT, sys.argv[2]), sys.argv[3:]
B.build()
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
srcs = B.sources()
src = [s for s in srcs if os.path.basename(s) == unit][0]
os.makedirs(os.path.dirname(out), exist_ok=True)
obj = out + ".o"
subprocess.run([hipcc] + [f for f in B.FLAGS if f != "-shared"] + defines + ["-c", src, "-o", obj], check=True)
# The variant keeps the production digest of ALL sources (build.py / bench.py would otherwise rebuild it over the experiment) but
# carries a RENDER digest of its own: bench.py's pmc_traffic then refuses the production build's counter profile for it
# instead of passing it off as this library's.
import hashlib
ver_src = [s for s in srcs if os.path.basename(s) == "version.hip"][0]
ver_obj = out + ".version.o"
render = hashlib.sha1((B.csrc_digest() + unit + " ".join(defines)).encode()).hexdigest()[:16]
subprocess.run([hipcc] + [f for f in B.FLAGS if f != "-shared"] + [f'-DNVSF_CSRC_DIGEST_ALL="{B.csrc_digest_all()}"', f'-DNVSF_CSRC_DIGEST_RENDER="{render}"',
                                                                  "-c", ver_src, "-o", ver_obj], check=True)
objs = [obj if s == src else (ver_obj if s == ver_src else B._obj(s)) for s in srcs]
subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-fvisibility=hidden"] + objs + ["-o", out], check=True)
os.remove(obj)
os.remove(ver_obj)
print(out, "render digest", render)
