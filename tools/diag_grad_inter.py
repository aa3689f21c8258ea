"""Compares intermediates (sigma, geo_feat, rgbs and their gradients) of the mid_cam / mid_lidar gradient cases between the reference
(build_tmp/diag_midcam.npz, generated on CPU by importing the reference) and the HIP training graph."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selfsupervised-nvsf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_dynamic as GD  # noqa: E402
from nvsf import synthetic as S  # noqa: E402
from nvsf.nerf.models.network_dynamic import NeRFNetwork  # noqa: E402

dev = torch.device("cuda:0")
ref = np.load(os.path.join(ROOT, "build_tmp", "diag_midcam.npz"))
net = NeRFNetwork(min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, **GD.SMALL).train()
GD.init_by_name(net)
net = net.to(dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for lidar in (False, True):
    k = "lidar" if lidar else "cam"
    o, d, noise, gt = GD.grad_case_inputs("mid", lidar, S)
    rec = {}
    real_density, real_color = net.density, net.color

    def density(x, tt, l, **kw):
        r = real_density(x, tt, l, **kw)
        r["sigma"].retain_grad(); r["geo_feat"].retain_grad()
        rec["sigma"], rec["geo"] = r["sigma"], r["geo_feat"]
        return r

    def color(x, dd, cal_lidar_color=False, mask=None, **kw):
        r = real_color(x, dd, cal_lidar_color=cal_lidar_color, mask=mask, **kw)
        r.retain_grad()
        rec["rgb"] = r
        return r
    net.density, net.color = density, color
    for p in net.parameters():
        p.grad = None
    noise_dev = t(noise)
    real_rand = torch.rand
    torch.rand = lambda *a, **kk: noise_dev
    try:
        out = net.render(t(o)[None], t(d)[None], torch.tensor([[0.5]], device=dev), cal_lidar_color=lidar, num_steps=GD.GRAD_T, perturb=True, staged=False)
    finally:
        torch.rand = real_rand
    net.density, net.color = real_density, real_color
    loss = GD.reference_losses(out, t(gt), lidar)
    loss.backward()
    print(f"== {k}")
    for name, mine in (("sigma", rec["sigma"]), ("geo", rec["geo"]), ("rgb", rec["rgb"]), ("weights", out["weights"]),
                       ("g_sigma", rec["sigma"].grad), ("g_geo", rec["geo"].grad), ("g_rgb", rec["rgb"].grad)):
        r = ref[f"{k}/{name}"].astype(np.float64).reshape(-1)
        m = mine.detach().double().cpu().numpy().reshape(-1)
        e = np.abs(m - r)
        rel_each = e / (np.abs(r) + 1e-30)
        print(f"  {name:8s} max|ref| {np.abs(r).max():.3e}  max err {e.max():.3e} ({e.max() / np.abs(r).max():.2e} of max)  "
              f"median rel {np.median(rel_each):.2e}  L2 rel {np.sqrt((e ** 2).sum() / (r ** 2).sum()):.2e}  worst idx {int(e.argmax())}")
