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


def reference_named(model):
    """{reference parameter name: (value, gradient or None)} of a module tree of this repo OR of the reference: this repo's Planes4D
    keeps its 24 planes in one channel-last parameter (`planes_cl`) and presents them under the reference's `planes.<scale>.<pair>`
    names as [1, C, H, W] views (Planes4D.reference_named_parameters)."""
    out = {}
    flat_owners = {n: m for n, m in model.named_modules() if hasattr(m, "reference_named_parameters")}
    for name, p in model.named_parameters():
        owner = name.rsplit(".", 1)[0] if "." in name else ""
        if name.endswith("planes_cl") and owner in flat_owners:
            for rname, value, grad in flat_owners[owner].reference_named_parameters(prefix=(owner + "." if owner else "")):
                out[rname] = (value, grad)
        else:
            out[name] = (p, p.grad)
    return out


def init_by_name(model):
    """Deterministic, structure-revealing parameter values keyed by (reference) parameter name."""
    with torch.no_grad():
        for name, (p, _) in reference_named(model).items():
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
    # options no configuration of the reference switches on: the other reductions of the three pair features, the frequency embedding
    with torch.no_grad():
        for red in ("prod", "sum", "mean"):
            enc_r = hf.HashGrid4D(base_resolution=16, max_resolution=256, time_resolution=4, n_levels=8, n_features_per_level=4,
                                  log2_hashmap_size=12, reduction=red)
            init_by_name(enc_r)  # name-derived seeds: the same tables as `enc`
            s, d = enc_r(torch.from_numpy(x), torch.tensor([[0.77]], dtype=torch.float32))
            out[f"dyn_t11_{red}"], out[f"dyn_t0_{red}"] = d.numpy(), enc_r.forward_dynamic(torch.from_numpy(x), torch.tensor(0.77)).numpy()
        # (use_freq together with use_grid does not run in the reference: interpT views the grid features with the summed width,
        # flow_field.py:121, and torch.cat at :130 fails)
        for name, kw in (("freq_only", dict(use_freq=True, use_grid=False)),):
            fl = ff.FlowField(n_levels=16, n_features_per_level=8, base_resolution=32, max_resolution=8192, log2_hashmap_size=14, **kw)
            init_by_name(fl)
            out[f"flow_{name}"] = fl(torch.from_numpy(xt)).numpy()
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


GRAD_CASES = (("first", 0.0), ("mid", 0.5), ("last", 1.0))
GRAD_N, GRAD_T = 32, 24
BIG = 1 << 20       # tensors above this size (the flow field's 30 M-entry grid) are stored as a signature, see grad_signature
N_BUCKETS = 4093


def reference_losses(out, gt, lidar, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, alpha_rgb=1.0, smooth=0.0):
    """The loss terms of Trainer.train_step that reach the renderer, restated for the fixture generator AND the GPU test
    (trainer.py:186-219: masked L1 range + MSE ray-drop + MSE intensity, criteria with reduction="none" summed at :540-543;
    :491-503: MSE RGB).  `gt`: images_lidar [1,N,3] = (raydrop, intensity, range) or the RGB targets [1,N,3]."""
    if lidar:
        gt_rd = gt[:, :, 0]
        gt_int, gt_depth = gt[:, :, 1] * gt_rd, gt[:, :, 2] * gt_rd
        pred_rd = out["image_lidar"][:, :, 0]
        pred_int = out["image_lidar"][:, :, 1] * gt_rd
        pred_depth = out["depth_lidar"] * gt_rd
        loss = alpha_d * (pred_depth - gt_depth).abs() + alpha_r * (pred_rd - gt_rd.clamp(smooth, 1 - smooth)) ** 2 \
            + alpha_i * (pred_int - gt_int) ** 2
        return loss.sum()
    return (alpha_rgb * (out["image"] - gt) ** 2).sum()


def grad_case_inputs(tag, lidar, synth):
    """Rays, jitter and targets of one gradient case (deterministic; shared by the generator and the GPU test)."""
    rng = np.random.default_rng(zlib.crc32(f"{tag}/{int(lidar)}".encode()) % 100000)
    o, d = (synth.lidar_rays if lidar else synth.camera_rays)(GRAD_N, rng)
    noise = rng.random((GRAD_N, GRAD_T)).astype(np.float32)
    if lidar:
        gt = np.stack([(rng.random(GRAD_N) > 0.3).astype(np.float32), rng.random(GRAD_N).astype(np.float32),
                       (rng.random(GRAD_N) * 0.8).astype(np.float32)], -1)[None]
    else:
        gt = rng.random((1, GRAD_N, 3)).astype(np.float32)
    return o, d, noise, gt


def grad_signature(g):
    """Compact fingerprint of the gradient of a very large table (fp64 sums): 4093 bucket sums (entry index mod 4093), L1 and
    L2 norm, number of non-zeros, and the 256 largest entries with their indices."""
    g = np.asarray(g, np.float64).reshape(-1)
    buckets = np.bincount(np.arange(g.size) % N_BUCKETS, weights=g, minlength=N_BUCKETS)
    top = np.argsort(-np.abs(g))[:256]
    return {"buckets": buckets.astype(np.float32), "l1": np.float64(np.abs(g).sum()), "l2": np.float64(np.sqrt((g * g).sum())),
            "nnz": np.int64(np.count_nonzero(g)), "top_idx": top.astype(np.int64), "top_val": g[top].astype(np.float32)}


FRAGILE_TOL = 2e-3


def fragile_rows(x, weights_f16, spec, tol=FRAGILE_TOL):
    """Rows of a head's input whose gradient depends on a coin flip: a hidden unit with |pre-activation| <= tol.  The inputs of the
    per-sample heads (geometry features = outputs of the density MLP) agree between two implementations of the specified fp16
    arithmetic to a few 1e-4 (an fp16 rounding of a hidden activation of the density MLP that falls the other way), which moves a
    head's pre-activations by up to ~1e-3; a ReLU whose pre-activation is closer to zero than that is "on" on one side and "off"
    on the other, and the whole input gradient of that sample changes by several per cent of its largest entry.  `x` [M, n_in]
    (any float dtype; the network sees it rounded to fp16), `weights_f16` the flat fp16 parameters; pre-activations in float64
    from fp16-rounded operands and fp16-rounded hidden activations."""
    mats = [m.double().numpy() for m in spec.split(torch.from_numpy(np.asarray(weights_f16, np.float16)))]
    M = x.shape[0]
    a = np.ones((M, spec.in_cols), np.float64)
    a[:, :spec.n_in] = np.asarray(x, np.float32).astype(np.float16).astype(np.float64)
    flag = np.zeros(M, bool)
    for W in mats[:-1]:
        p = a @ W.T
        flag |= (np.abs(p) <= tol).any(1)
        a = np.maximum(p, 0.0).astype(np.float16).astype(np.float64)
    return flag


def detach_rows(rgbs, rows):
    """rgbs with the autograd path of the flagged rows cut (values unchanged): those samples' heads contribute no gradient."""
    return torch.where(rows.view(-1, 1), rgbs.detach(), rgbs)


def gen_network_grads(mods, out_dir):
    """Parameter gradients of the reference's own NeRFNetwork.render + Trainer losses (network_dynamic.py:213-332 under
    autograd: which neighbour terms carry gradient is decided by the reference's code, :242-271), LiDAR and camera batch at
    the first / a middle / the last frame, training mode with jitter (perturb=True: the jitter the reference draws with
    torch.rand is replayed from the fixture)."""
    import sys
    S = sys.modules["make_golden_synth"]
    nd = mods["network"]
    net = nd.NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **SMALL).train()
    init_by_name(net)
    out = {}
    real_rand = torch.rand
    for tag, tv in GRAD_CASES:
        for lidar in (True, False):
            o, d, noise, gt = grad_case_inputs(tag, lidar, S)
            for p in net.parameters():
                p.grad = None
            torch.rand = lambda *a, **k: torch.from_numpy(noise)  # renderer_dynamic.py:163-164 draws rand(z_vals.shape)
            # The per-sample heads of the fragile rows (see fragile_rows) are cut out of the autograd graph -- in the reference's
            # render here and, with the same stored row mask, in the GPU test: what is compared is then free of ReLU coin flips.
            heads = (net.raydrop_net, net.intensity_net) if lidar else (net.color_net,)
            seen, rows_box = [], {}
            real_fwd = {h: h.forward for h in heads}
            real_color = net.color

            def color(x, dd, cal_lidar_color=False, mask=None, **kw):
                for h in heads:  # record each head's input on its way in
                    h.forward = (lambda hh: lambda inp: (seen.append((hh, inp.detach().float().numpy().copy())), real_fwd[hh](inp))[1])(h)
                try:
                    rgbs = real_color(x, dd, cal_lidar_color=cal_lidar_color, mask=mask, **kw)
                finally:
                    for h in heads:
                        h.forward = real_fwd[h]
                assert bool(mask.all()) and len(seen) == len(heads)  # every sample passes the weight threshold in these cases
                rows = np.zeros(rgbs.shape[0], bool)
                for h, inp in seen:
                    rows |= fragile_rows(inp, h.params.detach().numpy().astype(np.float16), h.spec)
                rows_box["rows"] = rows
                return detach_rows(rgbs, torch.from_numpy(rows))
            net.color = color
            try:
                res = net.render(torch.from_numpy(o)[None], torch.from_numpy(d)[None], torch.tensor([[tv]], dtype=torch.float32),
                                 cal_lidar_color=lidar, num_steps=GRAD_T, perturb=True, staged=False)
            finally:
                torch.rand = real_rand
                net.color = real_color
            loss = reference_losses(res, torch.from_numpy(gt), lidar)
            loss.backward()
            key = f"{tag}_{'lidar' if lidar else 'cam'}"
            out[f"{key}/loss"] = np.float32(loss.item())
            out[f"{key}/fragile_rows"] = rows_box["rows"]
            sfx = "_lidar" if lidar else ""
            out[f"{key}/image"], out[f"{key}/depth"] = res["image" + sfx][0].detach().numpy(), res["depth" + sfx][0].detach().numpy()
            n_with = 0
            for name, p in net.named_parameters():
                if p.grad is None or p.numel() == 0:
                    continue
                g = p.grad.detach().float().numpy()
                if not np.any(g):
                    continue
                if g.size > BIG:
                    for sk, sv in grad_signature(g).items():
                        out[f"{key}/gradsig/{name}/{sk}"] = sv
                else:
                    out[f"{key}/grad/{name}"] = g
                n_with += 1
            print(key, "loss", float(loss), "tensors with gradient:", n_with, "fragile rows:", int(rows_box["rows"].sum()), "of", rows_box["rows"].size)
    np.savez_compressed(os.path.join(out_dir, "network_dynamic_grads.npz"), **out)
    print("network_dynamic_grads.npz", len(out), "arrays")


# ---- BASELINE config 5 at its stated size: the model main_nvsf.py trains -------------------------------------------------------
# network_dynamic.py:16-23 + main_nvsf.py:45-52 defaults (L8 F4 T2^19 all levels hashed, base 512 -> max 32768, planes 4 scales of 32,
# flow grid L16 F8 T2^18) with time_resolution 8 (main_nvsf.py:47) and num_frames 64 (configs/kitti360_1908.txt): 93.6 M parameters.
RD = dict(time_resolution=8, num_frames=64, bound=2)
RD_N, RD_T = 24, 768
RD_TIMES = (("first", 0.0), ("mid", 0.5), ("last", 1.0))
RD_FLOWS = (("still", 1e-8), ("moving", 8e-4))  # target mean |flow| (unit cube): ~static scene / every sample displaced by ~25 finest cells


def rd_rays(tag, lidar, synth):
    rng = np.random.default_rng(zlib.crc32(f"rd/{tag}/{int(lidar)}".encode()) % 100000)
    return (synth.lidar_rays if lidar else synth.camera_rays)(RD_N, rng)


def rd_set_flow_gain(net, gain):
    """Scales the last (bias-free, linear) layer of the flow MLP: the flow is linear in it."""
    last = [m for m in net.flow_net.mlp if isinstance(m, torch.nn.Linear)][-1]
    with torch.no_grad():
        last.weight.mul_(float(gain))


def gen_network_rd(mods, out_dir):
    """The reference's own NeRFNetwork at its default size on CPU (`tinycudann` := tcnn_cpu_spec): LiDAR and camera renders of
    RD_N rays x 768 samples at the first / a middle / the last frame, once with ~zero scene flow (the neighbour-frame evaluations
    fall into the base evaluation's cells: k_hash_dynamic3's re-use branch) and once with |flow| ~ 8e-4 (every level re-gathers).
    Stored: rays, flow gains and the renders -- parameters come from init_by_name."""
    import sys
    import time as _time
    S = sys.modules["make_golden_synth"]
    nd = mods["network"]
    t0 = _time.time()
    net = nd.NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **RD).eval()
    init_by_name(net)
    print("reference NeRFNetwork (defaults):", sum(p.numel() for p in net.parameters()) / 1e6, "M parameters,", round(_time.time() - t0, 1), "s")
    out = {}
    rng = np.random.default_rng(77)
    probe = torch.from_numpy(rng.uniform(-1.9, 1.9, (4096, 3)).astype(np.float32))
    with torch.no_grad():
        base = float(net.flow(probe, torch.tensor([[0.5]]))["flow_forward"].abs().mean())
    gain_now = 1.0
    for ftag, target in RD_FLOWS:
        gain = target / base
        rd_set_flow_gain(net, gain / gain_now)
        gain_now = gain
        out[f"flow_gain_{ftag}"] = np.float64(gain)
        with torch.no_grad():
            out[f"mean_abs_flow_{ftag}"] = np.float32(net.flow(probe, torch.tensor([[0.5]]))["flow_forward"].abs().mean())
        for tag, tv in RD_TIMES:
            for lidar in (True, False):
                o, d = rd_rays(tag, lidar, S)
                t1 = _time.time()
                with torch.no_grad():
                    res = net.render(torch.from_numpy(o)[None], torch.from_numpy(d)[None], torch.tensor([[tv]], dtype=torch.float32),
                                     cal_lidar_color=lidar, num_steps=RD_T)
                sfx = "_lidar" if lidar else ""
                key = f"{ftag}/{tag}/{'lidar' if lidar else 'cam'}"
                out[key + "/image"], out[key + "/depth"] = res["image" + sfx][0].numpy(), res["depth" + sfx][0].numpy()
                out[key + "/weights_sum"] = res["weights_sum" + sfx].numpy()
                if tag == "mid":
                    out[key + "/weights"] = res["weights"].numpy().astype(np.float16)  # for locating a difference, not for the bar
                print(key, "ws", float(res["weights_sum" + sfx].mean()), round(_time.time() - t1, 1), "s")
    np.savez_compressed(os.path.join(out_dir, "network_dynamic_rd.npz"), **out)
    print("network_dynamic_rd.npz", len(out), "arrays")


GENERATORS = {"planes": gen_planes, "hash4d": gen_hash4d, "network": gen_network, "network_grads": gen_network_grads,
              "network_rd": gen_network_rd}
