"""Fixture generators for the space-time rows (a13 HashGridT/HashGrid4D, a14 Planes4D, a15 FlowField,
a18 NeRFNetwork.density/color/render), run through tests/golden/make_golden.py.  The REFERENCE's model files are
executed on CPU; `tinycudann` inside them is tests/golden/tcnn_cpu_spec.py (the CPU specification).  Parameters are
never stored: `init_by_name` fills any module tree (the reference's or this repo's -- they have the same names)
from seeds derived from the parameter names, so tests rebuild identical values.
"""
import os
import zlib

import numpy as np
import torch

import param_init

SMALL = dict(min_resolution=8, base_resolution=16, max_resolution=256, time_resolution=3, n_levels_plane=4,
             n_features_per_level_plane=8, n_levels_hash=8, n_features_per_level_hash=4, log2_hashmap_size=12, num_frames=9, bound=2)


def _seed(name):
    return zlib.crc32(name.encode()) % 100000


def init_by_name(model):
    """Deterministic, structure-revealing parameter values keyed by parameter name."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.numel() == 0:
                continue
            key = name.split("unet")[0]
            if ".planes." in name or name.startswith("planes."):
                pair = int(name.split(".")[-1])
                v = param_init.plane_params(tuple(p.shape), _seed(name), time_plane=pair in (2, 4, 5))
            elif name.endswith("params") and p.dim() == 1:
                owner = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
                spec = owner.spec
                if hasattr(spec, "shapes"):
                    v = param_init.mlp_params(spec.shapes, _seed(name), gain=1.0)
                else:
                    v = param_init.grid_params(p.numel(), _seed(name), std=0.15)
            elif "flow_net.mlp" in name or name.startswith("mlp."):
                rng = np.random.default_rng(7000 + _seed(name))
                last = p.shape[0] == 6
                v = (rng.standard_normal(tuple(p.shape)) * (0.01 if last else 0.35)).astype(np.float32)
            else:
                continue
            p.copy_(torch.from_numpy(np.ascontiguousarray(v)).reshape(p.shape))
            del key


def gen_planes(mods, out_dir):
    pf = mods["planes"]
    enc = pf.Planes4D(resolution=[8, 8, 8, 5], multiscale_res=[1, 2, 4, 8])
    init_by_name(enc)
    rng = np.random.default_rng(11)
    xt = rng.random((777, 4)).astype(np.float32)
    xt[:6] = [[0, 0, 0, 0], [1, 1, 1, 1], [0.5, 0.5, 0.5, 0.5], [1, 0, 1, 0], [0.999999, 1e-7, 0.25, 0.75], [0.3, 0.6, 0.9, 1.0]]
    with torch.no_grad():
        s, d = enc(torch.from_numpy(xt))
        d_only = enc.forward_dynamic(torch.from_numpy(xt))
        s_only = enc.forward_static(torch.from_numpy(xt))
    assert torch.equal(d, d_only) and torch.equal(s, s_only)
    # gradients (torch autograd through F.grid_sample) for the backward kernel
    xg = torch.from_numpy(xt).clone().requires_grad_()
    s2, d2 = enc(xg)
    gs = torch.from_numpy(rng.standard_normal(tuple(s2.shape)).astype(np.float32))
    gd = torch.from_numpy(rng.standard_normal(tuple(d2.shape)).astype(np.float32))
    ((s2 * gs).sum() + (d2 * gd).sum()).backward()
    np.savez_compressed(os.path.join(out_dir, "planes4d.npz"), xt=xt, static=s.numpy(), dynamic=d.numpy(), grad_static=gs.numpy(),
                        grad_dynamic=gd.numpy(), grad_xt=xg.grad.numpy(),
                        grad_plane_0_0=enc.planes[0][0].grad.numpy(), grad_plane_3_5=enc.planes[3][5].grad.numpy(),
                        grad_plane_2_3=enc.planes[2][3].grad.numpy())
    print("planes4d.npz", s.shape, d.shape)


def gen_hash4d(mods, out_dir):
    hf, ff = mods["hash"], mods["flow"]
    enc = hf.HashGrid4D(base_resolution=16, max_resolution=256, time_resolution=4, n_levels=8, n_features_per_level=4, log2_hashmap_size=12)
    init_by_name(enc)
    rng = np.random.default_rng(12)
    x = rng.random((600, 3)).astype(np.float32)
    out = {"x": x}
    with torch.no_grad():
        for i, tv in enumerate([0.0, 0.2, 1.0 / 3.0, 0.77, 1.0]):
            s, d = enc(torch.from_numpy(x), torch.tensor([[tv]], dtype=torch.float32))       # dimensioned t: fp32 blend
            d0 = enc.forward_dynamic(torch.from_numpy(x), torch.tensor(tv))                    # 0-dim t: fp16 blend
            out[f"t{i}"] = np.float32(tv)
            out[f"static{i}"], out[f"dyn_t11_{i}"], out[f"dyn_t0_{i}"] = s.numpy(), d.numpy(), d0.numpy()
    flow = ff.FlowField(n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=8192, log2_hashmap_size=14)
    init_by_name(flow)
    xt = np.concatenate([x, np.full((600, 1), 0.4, np.float32)], 1)
    with torch.no_grad():
        out["flow_xt"] = xt
        out["flow"] = flow(torch.from_numpy(xt)).numpy()
    np.savez_compressed(os.path.join(out_dir, "hash4d_flow.npz"), **out)
    print("hash4d_flow.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def gen_network(mods, out_dir):
    import sys
    S = sys.modules["make_golden_synth"]
    nd = mods["network"]
    net = nd.NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **SMALL).eval()
    init_by_name(net)
    rng = np.random.default_rng(13)
    out = {}
    pts = rng.uniform(-1.9, 1.9, (300, 3)).astype(np.float32)
    with torch.no_grad():
        for tag, tv in (("mid", 0.5), ("first", 0.0), ("last", 1.0)):
            t = torch.tensor([[tv]], dtype=torch.float32)
            for lidar in (True, False):
                dres = net.density(torch.from_numpy(pts), t, lidar)
                out[f"density_{tag}_{int(lidar)}_sigma"] = dres["sigma"].numpy()
                out[f"density_{tag}_{int(lidar)}_geo"] = dres["geo_feat"].numpy()
        fl = net.flow(torch.from_numpy(pts), torch.tensor([[0.5]]))
        out["flow_forward"], out["flow_backward"] = fl["flow_forward"].numpy(), fl["flow_backward"].numpy()
        N, T = 40, 24
        for lidar in (True, False):
            o, d = (S.lidar_rays if lidar else S.camera_rays)(N, rng)
            res = net.render(torch.from_numpy(o)[None], torch.from_numpy(d)[None], torch.tensor([[0.375]]), cal_lidar_color=lidar, num_steps=T)
            sfx = "_lidar" if lidar else ""
            k = "lidar" if lidar else "cam"
            out[f"{k}_rays_o"], out[f"{k}_rays_d"] = o, d
            out[f"{k}_image"], out[f"{k}_depth"] = res["image" + sfx][0].numpy(), res["depth" + sfx][0].numpy()
            out[f"{k}_weights"], out[f"{k}_z_vals"], out[f"{k}_weights_sum"] = res["weights"].numpy(), res["z_vals"].numpy(), res["weights_sum" + sfx].numpy()
    out["pts"] = pts
    import json
    with open(os.path.join(out_dir, "network_state_dict_keys.json"), "w") as f:  # checkpoint schema of the reference model (row f4)
        json.dump({k: list(v.shape) for k, v in net.state_dict().items()}, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(out_dir, "network_dynamic.npz"), **out)
    print("network_dynamic.npz", len(out), "arrays")


GENERATORS = {"planes": gen_planes, "hash4d": gen_hash4d, "network": gen_network}
