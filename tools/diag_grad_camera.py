"""Where do the per-cent level camera-case gradient differences of tests/test_dynamic_gpu.py::test_training_graph_gradients_match_reference
come from?  (a) nvsf_mlp_bwd alone against the exact chain rule of the specified forward on camera-head shaped data;
(b) the mid_cam fixture case with ops.mlp_backward replaced by an fp64 torch chain (test-only), per-tensor errors with and without."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle_lib as O  # noqa: E402
from nvsf import field_ops as ops  # noqa: E402

dev = torch.device("cuda:0")


def exact_chain(x, w16, spec, g_out):
    """tcnn_cpu_spec._MlpFn.backward: activations from the oracle forward (fp16 hidden), chain rule in fp64."""
    xn = x.detach().cpu().numpy()
    w = w16.detach().cpu().numpy().astype(np.float16)
    out, hid = O.mlp_fwd(xn, w, spec.n_in, spec.in_cols, spec.n_hidden, spec.hidden, spec.out_cols, want_hidden=True)
    M = xn.shape[0]
    mats = [m.double().numpy() for m in spec.split(torch.from_numpy(w))]
    a0 = np.ones((M, spec.in_cols), np.float64)
    a0[:, :spec.n_in] = xn.astype(np.float16).astype(np.float64)
    acts = [a0] + [hid[:, l, :].astype(np.float64) for l in range(spec.n_hidden)]
    g = np.zeros((M, spec.out_cols), np.float64)
    g[:, :spec.n_out] = g_out.detach().double().cpu().numpy()
    grads = []
    for li in range(len(mats) - 1, -1, -1):
        grads.append(g.T @ acts[li])
        g = g @ mats[li]
        if li > 0:
            g = g * (acts[li] > 0)
    return g[:, :spec.n_in], np.concatenate([t.reshape(-1) for t in reversed(grads)]), hid


def part_a():
    print("== (a) nvsf_mlp_bwd alone, colour-head shape 31-64-64-3")
    spec = ops.MlpSpec(31, 3, 64, 2)
    rng = np.random.default_rng(0)
    M = 768
    w = np.concatenate([(rng.standard_normal(a * b) * (1.0 / np.sqrt(b))).astype(np.float32) for a, b in spec.shapes])
    w16 = torch.from_numpy(w).half().to(dev)
    x = torch.from_numpy((rng.standard_normal((M, 31)) * 0.5).astype(np.float32)).to(dev)
    for label, mag in (("g_out ~ 1e-2", 1e-2), ("g_out log-uniform 1e-7..1e-2", None), ("g_out ~ 1e-5", 1e-5)):
        if mag is None:
            g = rng.standard_normal((M, 3)) * 10 ** rng.uniform(-7, -2, (M, 1))
        else:
            g = rng.standard_normal((M, 3)) * mag
        g_out = torch.from_numpy(g.astype(np.float32)).to(dev)
        gx, gw = ops.mlp_backward(x, w16, spec, g_out)
        gx_ref, gw_ref, hid = exact_chain(x, w16, spec, g_out)
        ex = np.abs(gx.cpu().numpy() - gx_ref)
        ew = np.abs(gw.cpu().numpy() - gw_ref)
        row_scale = np.abs(gx_ref).max(1) + 1e-30
        row_rel = ex.max(1) / row_scale
        print(f"  {label}: dX max err / max {ex.max() / np.abs(gx_ref).max():.2e}; rows with rel err > 1e-2: {(row_rel > 1e-2).sum()} of {M}, "
              f"> 1e-3: {(row_rel > 1e-3).sum()}; median row rel {np.median(row_rel):.2e}; dW max err / max {ew.max() / np.abs(gw_ref).max():.2e}")
        # hidden activations of the kernel's forward against the oracle's (flips?)
    out = ops.mlp_forward(x, w16, spec)
    ref = O.mlp_fwd(x.cpu().numpy(), w16.cpu().numpy(), spec.n_in, spec.in_cols, spec.n_hidden, spec.hidden, spec.out_cols)
    print("  forward max |diff| vs oracle:", float(np.abs(out.cpu().numpy() - ref).max()))


def part_b():
    print("== (b) mid_cam / mid_lidar fixture case: per-tensor max error / largest entry, kernel backward vs fp64-chain backward")
    import copy
    import golden_dynamic as GD
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_dynamic import NeRFNetwork
    g = np.load(os.path.join(ROOT, "tests", "golden", "network_dynamic_grads.npz"))
    net = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL).eval()
    GD.init_by_name(net)
    net = net.to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    real_bwd = ops.mlp_backward

    def exact_bwd(x, weights_f16, spec, grad_out, need_grad_x=True, grad_scale=128.0, grad_x=None, gx_col0=0, accumulate=False, prefix=None):
        if prefix is not None:
            rows, per, n_cols = prefix
            full = torch.cat([rows[:, :n_cols].repeat_interleave(per, 0), x[:, :spec.n_in - n_cols]], 1)
        else:
            full = x
        gx, gw, _ = exact_chain(full[:, :spec.n_in].float(), weights_f16, spec, grad_out)
        gw = torch.from_numpy(gw.astype(np.float32)).to(dev)
        if not need_grad_x and grad_x is None:
            return None, gw
        gxt = torch.from_numpy(gx[:, gx_col0:].astype(np.float32)).to(dev)
        if grad_x is not None:
            if accumulate:
                grad_x += gxt
            else:
                grad_x.copy_(gxt)
            return grad_x, gw
        return gxt, gw

    for key in ("mid_cam", "mid_lidar", "first_cam"):
        tag, mod = key.rsplit("_", 1)
        lidar = mod == "lidar"
        tv = dict(GD.GRAD_CASES)[tag]
        o, d, noise, gt = GD.grad_case_inputs(tag, lidar, S)
        res = {}
        for which in ("kernel", "exact"):
            ops.mlp_backward = real_bwd if which == "kernel" else exact_bwd
            m = copy.deepcopy(net).train()
            noise_dev = t(noise)
            real_rand = torch.rand
            torch.rand = lambda *a, **k: noise_dev
            try:
                out = m.render(t(o)[None], t(d)[None], torch.tensor([[tv]], dtype=torch.float32, device=dev), cal_lidar_color=lidar,
                               num_steps=GD.GRAD_T, perturb=True, staged=False)
            finally:
                torch.rand = real_rand
            loss = GD.reference_losses(out, t(gt), lidar)
            loss.backward()
            params = dict(m.named_parameters())
            emax = {}
            for k in g.files:
                if not k.startswith(key + "/grad/"):
                    continue
                name = k[len(key) + 6:]
                ref = g[k].astype(np.float64)
                mine = params[name].grad.detach().double().cpu().numpy().reshape(ref.shape)
                emax[name] = float(np.abs(mine - ref).max() / np.abs(ref).max())
            res[which] = emax
        ops.mlp_backward = real_bwd
        for which in ("kernel", "exact"):
            v = np.array(list(res[which].values()))
            worst = max(res[which].items(), key=lambda kv: kv[1])
            print(f"  {key} [{which:6s}] median {np.median(v):.2e}  worst {worst[1]:.2e} ({worst[0]})")
        for name in ("color_net.params", "sigma_net.params", "hash_encoder_camera.hash_static.params", "raydrop_net.params"):
            if name in res["kernel"]:
                print(f"     {name}: kernel {res['kernel'][name]:.2e}  exact {res['exact'][name]:.2e}")


if __name__ == "__main__":
    part_a()
    part_b()
