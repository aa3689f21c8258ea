#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE'S OWN PYTHON from
/root/reference (read-only) and running it on CPU.  Run in the build container only:

    python tests/golden/make_golden.py [--only renderer|planes|hash4d|network]

The reference package cannot be imported wholesale (nvsf/__init__.py pulls open3d, cv2, tinycudann, ...),
so the individual model files are loaded by path under their real dotted names, with
  * `trimesh` stubbed (debug-only import of renderer_dynamic.py:2),
  * `nvsf.nerf.raymarching.raymarching.near_far_from_aabb` provided by the CPU oracle (the reference's
    implementation is a CUDA kernel),
  * `tinycudann` provided by tests/golden/tcnn_cpu_spec.py, the CPU specification of the tcnn operators
    (oracle/field_oracle.c) -- tiny-cuda-nn itself is an unpinned external dependency that is not installed.
Only inputs / outputs (numpy arrays) and seeds are written; no reference source text is stored.
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
sys.path.insert(0, HERE)

import oracle_lib as O  # noqa: E402
import tcnn_cpu_spec  # noqa: E402  (imports this repo's nvsf.field_ops BEFORE the name `nvsf` is stubbed below)
from nvsf import synthetic as S  # noqa: E402

sys.modules["make_golden_synth"] = S  # lets golden_dynamic.py reach this repo's ray generators after `nvsf` is stubbed

_OWN = {k: v for k, v in sys.modules.items() if k == "nvsf" or k.startswith("nvsf.")}  # this repo's package modules


def _stub_packages():
    for name in ("nvsf", "nvsf.nerf", "nvsf.nerf.models", "nvsf.nerf.raymarching"):
        if name not in sys.modules or getattr(sys.modules[name], "__golden_stub__", False) is False:
            m = types.ModuleType(name)
            m.__path__ = []
            m.__golden_stub__ = True
            sys.modules[name] = m
    sys.modules["trimesh"] = types.ModuleType("trimesh")
    rm = types.ModuleType("nvsf.nerf.raymarching.raymarching")

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        n, f = O.near_far_from_aabb(rays_o.detach().numpy(), rays_d.detach().numpy(), aabb.detach().numpy(), float(min_near))
        return torch.from_numpy(n), torch.from_numpy(f)

    rm.near_far_from_aabb = near_far_from_aabb
    sys.modules["nvsf.nerf.raymarching.raymarching"] = rm
    sys.modules["nvsf.nerf.raymarching"].raymarching = rm
    sys.modules["tinycudann"] = tcnn_cpu_spec


def _load(dotted, rel):
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_models():
    _stub_packages()
    act = _load("nvsf.nerf.activation", "nvsf/nerf/activation.py")
    rd = _load("nvsf.nerf.models.renderer_dynamic", "nvsf/nerf/models/renderer_dynamic.py")
    pf = _load("nvsf.nerf.models.planes_field", "nvsf/nerf/models/planes_field.py")
    hf = _load("nvsf.nerf.models.hash_field", "nvsf/nerf/models/hash_field.py")
    ff = _load("nvsf.nerf.models.flow_field", "nvsf/nerf/models/flow_field.py")
    _load("nvsf.nerf.models.unet", "nvsf/nerf/models/unet.py")
    nd = _load("nvsf.nerf.models.network_dynamic", "nvsf/nerf/models/network_dynamic.py")
    return dict(activation=act, renderer=rd, planes=pf, hash=hf, flow=ff, network=nd)


# ------------------------------------------------------------------------------------------------
def analytic_field(xyz, dirs, C):
    """Closed-form density / colour used to drive the reference compositor (any smooth function works)."""
    r2 = (xyz ** 2).sum(-1)
    sigma = 40.0 * torch.exp(-3.0 * r2) * (0.5 + 0.5 * torch.sin(7.0 * xyz[..., 0] + 3.0 * xyz[..., 1] * xyz[..., 2])) ** 2
    sigma = torch.where(r2 > 2.5, torch.zeros_like(sigma), sigma)
    freqs = torch.arange(1, C + 1, dtype=xyz.dtype)
    rgb = torch.sigmoid(torch.sin(xyz[..., :1] * 3.0 * freqs + dirs[..., 1:2] * 2.0) + xyz[..., 2:3] * freqs)
    return sigma, rgb


def gen_renderer(mods, out_dir):
    """a10/a11: NeRFRenderer.run / render (renderer_dynamic.py:109-326) with an analytic field."""
    rd = mods["renderer"]

    class Probe(rd.NeRFRenderer):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.out_color_dim, self.out_lidar_color_dim = 3, 2
            self.rec = {}

        def density(self, x, t=None, cal_lidar_color=False, **kw):
            self.rec["xyzs"] = x.detach().clone()
            sigma, _ = analytic_field(x, x, 1)
            self.rec["sigma"] = sigma.detach().clone()
            return {"sigma": sigma, "geo_feat": x[..., :1] * 0}

        def color(self, x, d, cal_lidar_color=False, mask=None, **kw):
            C = self.out_dim
            _, rgb = analytic_field(x, d, C)
            self.rec["rgb_full"] = rgb.detach().clone()
            self.rec["mask"] = mask.detach().clone()
            out = torch.zeros(mask.shape[0], C)
            out[mask] = rgb[mask]
            return out

    cases = {}
    rng = np.random.default_rng(42)
    specs = [
        ("lidar_plain", dict(lidar=True, active=False, dscale=1.0, perturb=False, N=64, T=64, bg=None)),
        ("lidar_active_perturb", dict(lidar=True, active=True, dscale=1.5, perturb=True, N=48, T=96, bg=None)),
        ("camera_bg1", dict(lidar=False, active=False, dscale=1.0, perturb=False, N=64, T=64, bg=None)),
        ("camera_bgvec_perturb", dict(lidar=False, active=False, dscale=2.0, perturb=True, N=40, T=130, bg=[0.1, 0.6, 0.3])),
        ("camera_c1_1024x64", dict(lidar=False, active=False, dscale=1.0, perturb=False, N=1024, T=64, bg=None)),  # BASELINE config 1 shape
    ]
    for name, c in specs:
        m = Probe(bound=S.BOUND, density_scale=c["dscale"], min_near=0.05, min_near_lidar=S.MIN_NEAR,
                  lidar_max_depth=1.5, active_sensor=c["active"]).eval()
        o, d = (S.lidar_rays if c["lidar"] else S.camera_rays)(c["N"], rng)
        o = o * (1.0 if c["lidar"] else 2.0)
        to, td = torch.from_numpy(o)[None], torch.from_numpy(d)[None]
        seed = int(rng.integers(1 << 30))
        torch.manual_seed(seed)
        bg = None if c["bg"] is None else torch.tensor(c["bg"])
        with torch.no_grad():
            out = m.render(to, td, torch.tensor([[0.5]]), cal_lidar_color=c["lidar"], num_steps=c["T"], perturb=c["perturb"], bg_color=bg)
        noise = None
        if c["perturb"]:
            torch.manual_seed(seed)
            noise = torch.rand(c["N"], c["T"]).numpy()
        sfx = "_lidar" if c["lidar"] else ""
        C = 2 if c["lidar"] else 3
        cases[name] = dict(
            rays_o=o, rays_d=d, lidar=c["lidar"], active=c["active"], density_scale=c["dscale"], T=c["T"],
            min_near=0.05, min_near_lidar=S.MIN_NEAR, lidar_max_depth=1.5, bound=S.BOUND,
            noise=noise if noise is not None else np.zeros(0, np.float32), bg=np.asarray(c["bg"] if c["bg"] else [], np.float32),
            xyzs=(m.rec["xyzs"].numpy().reshape(c["N"], c["T"], 3) if c["N"] <= 128 else np.zeros(0, np.float32)),
            sigma=m.rec["sigma"].numpy().reshape(c["N"], c["T"]),
            rgb_full=m.rec["rgb_full"].numpy().reshape(c["N"], c["T"], C), mask=m.rec["mask"].numpy().reshape(c["N"], c["T"]),
            z_vals=out["z_vals"].numpy(), weights=out["weights"].numpy(), weights_sum=out["weights_sum" + sfx].numpy(),
            depth=out["depth" + sfx][0].numpy(), image=out["image" + sfx][0].numpy())
        # staged render must equal the single-pass render (renderer_dynamic.py:286-316)
        torch.manual_seed(seed)
        if not c["perturb"]:
            with torch.no_grad():
                st = m.render(to, td, torch.tensor([[0.5]]), cal_lidar_color=c["lidar"], staged=True, max_ray_batch=17, num_steps=c["T"], bg_color=bg)
            assert torch.allclose(st["image" + sfx], out["image" + sfx], atol=1e-6) and torch.allclose(st["depth" + sfx], out["depth" + sfx], atol=1e-6)
    flat = {f"{k}/{kk}": np.asarray(vv) for k, v in cases.items() for kk, vv in v.items()}
    np.savez_compressed(os.path.join(out_dir, "renderer_uniform.npz"), **flat)
    # trunc_exp (activation.py:6-20): forward + backward on a few values
    x = torch.tensor([-20.0, -15.0, -1.0, 0.0, 0.5, 3.0, 15.0, 20.0], requires_grad=True)
    y = mods["activation"].trunc_exp(x)
    y.backward(torch.ones_like(y))
    np.savez(os.path.join(out_dir, "trunc_exp.npz"), x=x.detach().numpy(), y=y.detach().numpy(), grad=x.grad.numpy())
    print("renderer_uniform.npz:", {k: v["weights"].shape for k, v in cases.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    mods = load_reference_models()
    gens = {"renderer": gen_renderer}
    try:
        import golden_dynamic  # planes / hash4d / flow / full network fixtures (added with those kernels)
        gens.update(golden_dynamic.GENERATORS)
    except ImportError:
        pass
    try:
        import golden_rays  # device ray generation (row f1)
        gens.update(golden_rays.GENERATORS)
    except ImportError:
        pass
    for name, fn in gens.items():
        if args.only in (None, name):
            fn(mods, HERE)


if __name__ == "__main__":
    main()
