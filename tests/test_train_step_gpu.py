"""GPU: the thin multimodal training step (losses -> backward through the HIP operators -> Adam) learns a synthetic
target, and the quality metrics agree with their reference formulas."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(S, teacher, dev, n=512, T=48, seed=0):
    rng = np.random.default_rng(seed)
    lo, ld = S.lidar_rays(n, rng)
    co, cd = S.camera_rays(n, rng)
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    b = {"rays_o_lidar": t(lo), "rays_d_lidar": t(ld), "rays_o": t(co), "rays_d": t(cd), "time": torch.tensor([[0.5]], device=dev)}
    with torch.no_grad():
        rl = teacher.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], cal_lidar_color=True, num_steps=T)
        rc = teacher.render(b["rays_o"], b["rays_d"], b["time"], num_steps=T)
    b["gt_depth"] = rl["depth_lidar"]
    b["gt_raydrop"] = (rl["image_lidar"][..., 0] > 0.45).float()
    b["gt_intensity"] = rl["image_lidar"][..., 1] * b["gt_raydrop"]
    b["gt_rgb"] = rc["image"]
    return b


def test_training_reduces_loss_and_metrics(dev):
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep, psnr, depth_rmse
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=14)
    torch.manual_seed(1)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.copy_(torch.randn(enc.params.shape, generator=g) * 0.2)
    teacher = teacher.to(dev).eval()
    student = NeRFNetworkStatic(**kw).to(dev)
    step = RenderTrainStep(student, lr=1e-2, iters=400, num_steps=48, use_urf_loss=True)
    batch = _batch(S, teacher, dev)
    losses = []
    for _ in range(120):
        loss, parts, n_coll = step.step(batch)
        losses.append(float(loss))
        assert n_coll == 0  # single process: no collective
    first, last = float(np.mean(losses[:5])), float(np.mean(losses[-5:]))
    assert np.isfinite(losses).all() and last < 0.8 * first, (first, last)
    assert set(parts) == {"depth", "raydrop", "intensity", "los", "rgb"}
    student.eval()
    with torch.no_grad():
        rc = student.render(batch["rays_o"], batch["rays_d"], batch["time"], num_steps=48)
        rl = student.render(batch["rays_o_lidar"], batch["rays_d_lidar"], batch["time"], cal_lidar_color=True, num_steps=48)
    p = psnr(rc["image"], batch["gt_rgb"])
    assert p > 12.0
    ref = -10 * np.log10(np.mean((rc["image"].cpu().numpy().astype(np.float64) - batch["gt_rgb"].cpu().numpy()) ** 2) + 1e-8)
    assert abs(p - ref) < 1e-9
    r = depth_rmse(rl["depth_lidar"], batch["gt_depth"], S.SCALE)
    assert 0.0 <= r < 80.0


def test_density_fn_equals_operator_chain(dev, monkeypatch):
    """ops.DensityFn (encode -> MLP -> trunc_exp / slice as one autograd node; logit gradient assembled by nvsf_sigma_geo_bwd,
    feature gradient handed to the table scatter in fp32) against the operator chain HashGridFn -> MlpFn -> trunc_exp."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(3)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                          log2_hashmap_size=15).to(dev)
    with torch.no_grad():
        m.hash_encoder_lidar.params.copy_(torch.randn_like(m.hash_encoder_lidar.params) * 0.3)
    M = 5000  # not a multiple of the kernels' tiles
    x = (torch.rand(M, 3, device=dev) * 2 - 1) * S.BOUND
    w_sigma, w_geo = torch.randn(M, device=dev), torch.randn(M, 15, device=dev)

    def run(mode):
        monkeypatch.setenv("NVSF_DENSITY_FN", mode)
        for p in m.parameters():
            p.grad = None
        out = m.density(x, None, cal_lidar_color=True)
        # loss scaled like a GradScaler step: the chain rounds the feature gradient to fp16 on its way to the table
        loss = 64.0 * ((out["sigma"] * w_sigma).sum() + (out["geo_feat"].float() * w_geo).sum())
        loss.backward()
        return (out["sigma"].detach().clone(), out["geo_feat"].detach().float().clone(), m.hash_encoder_lidar.params.grad.clone(),
                m.sigma_net.params.grad.clone())

    s1, g1, gt1, gw1 = run("fused")
    s0, g0, gt0, gw0 = run("chain")
    assert torch.equal(s1, s0) and torch.equal(g1, g0)  # same forward kernels
    assert float((gw1 - gw0).abs().max()) <= 1e-3 * float(gw0.abs().max())  # same kernel and operands: atomic order only
    # table gradient: fp32 feature gradients against the chain's fp16-rounded ones (2^-11 relative per contribution)
    assert float((gt1 - gt0).abs().max()) <= 2e-3 * float(gt0.abs().max())
    assert float(gt0.abs().max()) > 0 and int((gt1 != 0).sum()) >= int((gt0 != 0).sum())

    # sigma-only and geo-only graphs (the other gradient arrives as None)
    monkeypatch.setenv("NVSF_DENSITY_FN", "fused")
    for sel in ("sigma", "geo_feat"):
        for p in m.parameters():
            p.grad = None
        out = m.density(x, None, cal_lidar_color=True)
        (64.0 * out[sel].float().sum()).backward()
        assert torch.isfinite(m.hash_encoder_lidar.params.grad).all() and float(m.hash_encoder_lidar.params.grad.abs().max()) > 0


def test_fused_adam_matches_torch_adam(dev):
    """nvsf.nerf.adam.FusedAdam against torch.optim.Adam (betas 0.9 / 0.99, eps 1e-15, two groups with different lr): plain
    steps, steps under a GradScaler (device-side unscale), a skipped overflow step, and state_dict exchange both ways."""
    from nvsf.nerf.adam import FusedAdam
    torch.manual_seed(0)
    shapes = [(1000003,), (64, 32), (7,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]
    a, b = mk(), None
    torch.manual_seed(0)
    b = mk()
    groups = lambda ps: [{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 1e-3}]
    oa, ob = FusedAdam(groups(a), betas=(0.9, 0.99), eps=1e-15), torch.optim.Adam(groups(b), betas=(0.9, 0.99), eps=1e-15)

    def grads(seed, scale=1.0, poison=False):
        g = torch.Generator(device=dev).manual_seed(seed)
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, device=dev, generator=g) * 1e-3
            if poison and pa.numel() == 7:
                gr[3] = float("inf")
            pa.grad, pb.grad = (gr * scale).clone(), (gr * scale).clone()

    def close():
        for pa, pb in zip(a, b):
            # a step moves a parameter by ~lr = 1e-2: agreement to a few ulps of that (1e-2 * 2^-23 = 1.2e-9 each) over the steps
            assert torch.allclose(pa, pb, rtol=2e-6, atol=2e-7), float((pa - pb).abs().max())

    for s in range(5):
        grads(s)
        oa.step(); ob.step()
    close()
    # under GradScaler: scaled gradients, one overflowing step in the middle (skipped by both, scale halved by both)
    sa, sb = torch.amp.GradScaler("cuda", init_scale=1024.0), torch.amp.GradScaler("cuda", init_scale=1024.0)
    for s in range(5, 11):
        grads(s, scale=float(sa.get_scale()), poison=(s == 7))
        sa.scale(torch.zeros((), device=dev)); sb.scale(torch.zeros((), device=dev))  # what scale(loss) does to the scaler's state
        sa.step(oa); sb.step(ob)
        sa.update(); sb.update()
        assert sa.get_scale() == sb.get_scale()
    close()
    assert sa.get_scale() == 512.0 and float(oa._dev_state[0]) == 10.0  # 11 calls, one skipped
    # state dicts: same layout, loadable both ways
    da, db = oa.state_dict(), ob.state_dict()
    assert set(da["state"][0]) == set(db["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(da["state"][0]["step"]) == 10.0
    assert torch.allclose(da["state"][0]["exp_avg_sq"], db["state"][0]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    oa.load_state_dict(db); ob.load_state_dict(da)
    grads(99)
    oa.step(); ob.step()
    close()


def test_fused_adam_invalidates_fp16_weight_caches(dev):
    """The forward kernels read fp16 copies of tables / weights cached per parameter version: an optimiser that writes through
    raw pointers has to bump that version, or the next forward runs on stale weights."""
    import tinycudann as tcnn
    from nvsf.nerf.adam import FusedAdam
    net = tcnn.Network(32, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                                "n_hidden_layers": 1}).to(dev)
    opt = FusedAdam(net.parameters(), lr=1e-1, betas=(0.9, 0.99), eps=1e-15)
    before = net.weights_f16().clone()
    net.params.grad = torch.ones_like(net.params)
    v0 = net.params._version
    opt.step()
    assert net.params._version > v0
    after = net.weights_f16()
    assert not torch.equal(before, after) and torch.equal(after, net.params.detach().half())


def test_dense_regime_needs_no_host_sync_and_follows_the_mask(dev):
    """color(): the dense / sparse choice of the heads uses the previous batch's count while the modality stays dense (no host
    read per call), falls back to the synchronous read when the observed count drops, and gives the same values either way."""
    from nvsf import field_ops as ops, synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(2)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                          log2_hashmap_size=14).to(dev)
    m.out_dim = 2  # NeRFRenderer.run sets it per modality (renderer_dynamic.py:131)
    M = 4096
    x = torch.rand(M, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=-1)
    geo = torch.randn(M, 15, device=dev)
    dense = torch.rand(M, device=dev) < 0.8
    sparse = torch.rand(M, device=dev) < 0.05

    def ref(mask):  # gather / scatter form, no regime state involved
        out = torch.zeros(M, 2, device=dev)
        out[mask] = torch.sigmoid(ops.heads(m, (d[mask] + 1) / 2, geo[mask], True)).float()
        return out

    with torch.no_grad():
        a = m.color(x, d, cal_lidar_color=True, mask=dense, geo_feat=geo)       # first call: synchronous count, regime := dense
        assert m._mask_regime[True]["dense"] and torch.equal(a.float(), ref(dense))
        b = m.color(x, d, cal_lidar_color=True, mask=dense, geo_feat=geo)       # dense regime: no read, count copy in flight
        assert m._mask_regime[True]["pending"] is not None and torch.equal(b.float(), ref(dense))
        c = m.color(x, d, cal_lidar_color=True, mask=sparse, geo_feat=geo)      # still treated as dense: all samples, masked
        assert torch.equal(c.float(), ref(sparse))
        torch.cuda.synchronize()
        e = m.color(x, d, cal_lidar_color=True, mask=sparse, geo_feat=geo)      # the count of a sparse batch has landed by now
        f = m.color(x, d, cal_lidar_color=True, mask=sparse, geo_feat=geo)
        assert not m._mask_regime[True]["dense"] and torch.equal(e.float(), ref(sparse)) and torch.equal(f.float(), ref(sparse))
        z = m.color(x, d, cal_lidar_color=True, mask=torch.zeros(M, dtype=torch.bool, device=dev), geo_feat=geo)
        assert float(z.abs().max()) == 0.0
