"""GPU: the thin multimodal training step (losses -> backward through the HIP operators -> Adam) learns a synthetic
target, and the quality metrics agree with their reference formulas."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _LossScaler():
    from nvsf.nerf.loss_scaler import LossScaler  # the step's loss scaler (GradScaler's rule and state_dict; nvsf/nerf/loss_scaler.py)
    return LossScaler


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
    rd = rl["image_lidar"][..., 0]
    b["gt_raydrop"] = (rd > rd.median()).float()  # about half of the rays return (the losses mask range / intensity by this channel)
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
    step = RenderTrainStep(student, lr=1e-2, iters=400, num_steps=48, use_urf_loss=True, scale=S.SCALE)
    batch = _batch(S, teacher, dev)
    losses = []
    for _ in range(120):
        loss, parts, n_coll = step.step(batch)
        losses.append(float(loss))
        assert n_coll == 0  # single process: no collective
    first, last = float(np.mean(losses[:5])), float(np.mean(losses[-5:]))
    assert np.isfinite(losses).all() and last < 0.8 * first, (first, last)
    assert set(parts) == {"depth", "raydrop", "intensity", "chamfer", "los", "rgb"}
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


def test_density_fn_equals_operator_chain(dev, variants):
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
        variants.set(density_fn=mode)
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
    variants.set(density_fn="fused")
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
    assert sa.get_scale() == 512.0 and float(oa._rows[oa._row_of[a[0]], 0]) == 10.0  # 11 calls, one skipped
    assert len(set(oa._row_of.values())) == 1  # every parameter updated in the same steps: one counter row
    # state dicts: same layout, loadable both ways
    da, db = oa.state_dict(), ob.state_dict()
    assert set(da["state"][0]) == set(db["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(da["state"][0]["step"]) == 10.0
    assert torch.allclose(da["state"][0]["exp_avg_sq"], db["state"][0]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    oa.load_state_dict(db); ob.load_state_dict(da)
    grads(99)
    oa.step(); ob.step()
    close()


def test_fused_adam_keeps_a_step_count_per_parameter(dev):
    """torch.optim.Adam advances `step` only for parameters that received a gradient (LiDAR-only / camera-only
    steps leave the other modality's tables alone); FusedAdam must apply the same bias corrections and round-trip the steps."""
    from nvsf.nerf.adam import FusedAdam
    torch.manual_seed(1)
    mk = lambda: [torch.nn.Parameter(torch.randn(n, device=dev) * 0.1) for n in (4099, 257, 64)]
    a = mk()
    torch.manual_seed(1)
    b = mk()
    oa, ob = FusedAdam(a, lr=1e-2, betas=(0.9, 0.99), eps=1e-15), torch.optim.Adam(b, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    pattern = [(1, 1, 1), (1, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 0), (1, 1, 1), (0, 0, 1)]  # which parameters get a gradient
    for s, act in enumerate(pattern):
        g = torch.Generator(device=dev).manual_seed(100 + s)
        for pa, pb, on in zip(a, b, act):
            gr = torch.randn(pa.shape, device=dev, generator=g) * 1e-3
            pa.grad, pb.grad = (gr.clone(), gr.clone()) if on else (None, None)
        oa.step(); ob.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=2e-7), float((pa - pb).abs().max())
    da, db = oa.state_dict(), ob.state_dict()
    assert [float(da["state"][i]["step"]) for i in range(3)] == [float(db["state"][i]["step"]) for i in range(3)] == [5.0, 4.0, 6.0]
    # lossless round trip, and torch's state loads with its three different steps
    oa2 = FusedAdam(a, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    import copy
    oa2.load_state_dict(copy.deepcopy(db))  # Optimizer.load_state_dict aliases tensors that already sit on the right device
    assert [float(oa2.state_dict()["state"][i]["step"]) for i in range(3)] == [5.0, 4.0, 6.0]
    for pa, pb in zip(a, b):
        pa.grad = torch.full_like(pa, 1e-3)
        pb.grad = torch.full_like(pb, 1e-3)
    oa2.step(); ob.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=2e-7)


def test_ema_matches_torch_ema_rule_and_rides_in_the_adam_pass(dev):
    """nvsf.nerf.ema.ExponentialMovingAverage: update rule of torch_ema as the reference's Trainer uses it (trainer.py:112-114,
    1420-1421: decay 0.95 with the (1 + n) / (10 + n) warm-up), store / copy_to / restore, state_dict; and the every-step form
    folded into nvsf_adam_update."""
    from nvsf.nerf.adam import FusedAdam
    from nvsf.nerf.ema import ExponentialMovingAverage
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (100003, 64)]
    ema = ExponentialMovingAverage(ps, decay=0.95)
    ref = [p.detach().clone().double() for p in ps]
    for n in range(1, 15):
        with torch.no_grad():
            for p in ps:
                p.add_(torch.randn_like(p) * 0.1)
        ema.update()
        d = min(0.95, (1 + n) / (10 + n))
        ref = [r - (1 - d) * (r - p.detach().double()) for r, p in zip(ref, ps)]
    assert ema.num_updates == 14
    for s, r in zip(ema.shadow_params, ref):
        assert torch.allclose(s.double(), r, rtol=0, atol=2e-6)
    cur = [p.detach().clone() for p in ps]
    ema.store(); ema.copy_to()
    assert all(torch.equal(p.detach(), s) for p, s in zip(ps, ema.shadow_params))
    ema.restore()
    assert all(torch.equal(p.detach(), c) for p, c in zip(ps, cur))
    sd = ema.state_dict()
    assert set(sd) == {"decay", "num_updates", "shadow_params", "collected_params"} and sd["decay"] == 0.95
    # every-step form: shadow after the fused pass == separate update applied to the stepped parameter
    opt = FusedAdam(ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ema2 = ExponentialMovingAverage(ps, decay=0.9).attach(opt)
    before = [s.clone() for s in ema2.shadow_params]
    for p in ps:
        p.grad = torch.randn_like(p)
    ema2.before_step()
    opt.step()
    d = min(0.9, 2 / 11)
    for s, b0, p in zip(ema2.shadow_params, before, ps):
        assert torch.allclose(s, b0 - (b0 - p.detach()) * (1 - d), rtol=1e-6, atol=1e-7)


def test_losses_follow_the_reference_reductions(dev):
    """RenderTrainStep.losses against a plain-torch restatement of trainer.py:186-233, 491-503, 540-546: ground truth and
    predictions masked by the ray-drop channel, per-ray terms SUMMED (criteria are reduction="none", main_nvsf.py:205-221),
    chamfer term (d1 + d2).mean() / 2 on rays_d * depth / scale, smooth_factor default 0."""
    from nvsf.nerf.train_step import RenderTrainStep
    N = 700
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.rand(*s, generator=g).to(dev)
    fake = {"image_lidar": rnd(1, N, 2), "depth_lidar": rnd(1, N) * 0.8, "image": rnd(1, N, 3)}

    class Model(torch.nn.Module):
        num_frames = 8

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1, device=dev))

        def get_params(self, lr):
            return [{"params": [self.w], "lr": lr}]

        def render(self, o, d, t, cal_lidar_color=False, **kw):
            keys = ("image_lidar", "depth_lidar") if cal_lidar_color else ("image",)
            return {k: fake[k] * self.w for k in keys}

    d = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1).to(dev)
    images_lidar = torch.stack([(rnd(1, N) > 0.3).float(), rnd(1, N), rnd(1, N) * 0.8], -1)
    batch = {"rays_o_lidar": torch.zeros(1, N, 3, device=dev), "rays_d_lidar": d, "images_lidar": images_lidar, "rays_o": torch.zeros(1, N, 3, device=dev),
             "rays_d": d, "gt_rgb": rnd(1, N, 3), "time": torch.tensor([[0.5]], device=dev)}
    scale = 0.0108
    step = RenderTrainStep(Model(), scale=scale, ema_decay=None)
    assert step.smooth == 0.0
    total, parts = step.losses(batch)
    rd, it, dp = images_lidar[..., 0], images_lidar[..., 1] * images_lidar[..., 0], images_lidar[..., 2] * images_lidar[..., 0]
    pd = fake["depth_lidar"] * rd
    exp = {"depth": (pd - dp).abs().sum(), "raydrop": 0.01 * ((fake["image_lidar"][..., 0] - rd) ** 2).sum(),
           "intensity": 0.1 * ((fake["image_lidar"][..., 1] * rd - it) ** 2).sum(), "rgb": ((fake["image"] - batch["gt_rgb"]) ** 2).sum()}
    a, b = (d * pd[..., None] / scale)[0].double(), (d * dp[..., None] / scale)[0].double()
    dist = torch.cdist(a, b) ** 2
    exp["chamfer"] = (dist.min(1).values + dist.min(0).values).mean() * 0.5
    assert set(parts) == set(exp)
    for k in exp:
        assert abs(float(parts[k]) - float(exp[k])) <= 2e-5 * max(1.0, abs(float(exp[k]))), (k, float(parts[k]), float(exp[k]))
    assert abs(float(total) - sum(float(v) for v in exp.values())) <= 1e-4 * float(total)
    total.backward()  # the chamfer term back-propagates into the prediction
    assert torch.isfinite(step.model.w.grad).all() and float(step.model.w.grad.abs()) > 0
    # scene-flow loss (trainer.py:236-267)
    pcs = {3: rnd(300, 3), 4: rnd(280, 3), 2: rnd(310, 3)}

    class FlowModel(Model):
        def flow(self, x, t):
            return {"flow_forward": 0.01 * self.w * torch.ones_like(x), "flow_backward": -0.02 * self.w * torch.ones_like(x)}
    fs = RenderTrainStep(FlowModel(), scale=scale, ema_decay=None, flow_loss=True, pc_list=pcs)
    fl = fs.flow_loss(torch.tensor([[3.4 / 7]], device=dev))  # int(t * (F - 1)) = 3
    want = 0.0
    for off, tgt in ((0.01, pcs[4]), (-0.02, pcs[2])):
        dist = torch.cdist((pcs[3] + off).double(), tgt.double()) ** 2
        want += float((dist.min(1).values.sum() + dist.min(0).values.sum()) * 0.5) + abs(off)
    assert abs(float(fl) - want) <= 1e-5 * want


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


def test_side_stream_scatter_gives_the_same_step(dev):
    """RenderTrainStep issues the table scatters of DensityFn.backward on a side stream (they are atomic-bound; the MLP backward
    of the other modality runs beside them).  Same gradients as with everything on one stream (fp32 atomics: order-dependent
    in the last bits only), and the step's consumers see completed gradients."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=15)
    torch.manual_seed(4)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=700, T=64, seed=2)
    grads = {}
    for overlap in (False, True):
        torch.manual_seed(9)
        m = NeRFNetworkStatic(**kw).to(dev)
        step = RenderTrainStep(m, num_steps=64, scale=S.SCALE, ema_decay=None)
        step.scatter_overlap = overlap
        step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)  # no skipped steps, same scale both ways
        torch.manual_seed(10)  # the jitter of perturb=True
        step.step(batch)
        grads[overlap] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.numel() and p.grad is not None}
        step.step(batch)  # and the next step starts from completed state (no stale / half-written gradient buffers)
        torch.cuda.synchronize()
        assert all(torch.isfinite(p).all() for p in m.parameters())
    assert set(grads[False]) == set(grads[True]) and len(grads[True]) == 6
    for n in grads[False]:
        a, b = grads[False][n], grads[True][n]
        assert float(a.abs().max()) > 0
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), n  # same kernels and operands: fp32 atomic order only


def test_split_backward_gives_the_same_gradients(dev):
    """RenderTrainStep(split_backward=True) runs LiDAR forward + backward, then camera forward + backward (the LiDAR table scatter
    then overlaps the whole camera pass).  The loss terms are sums over disjoint ray sets: same loss, same gradients as one joint
    backward (the sigma MLP, which both passes reach, adds its two contributions in either order)."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=15)
    torch.manual_seed(4)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=700, T=64, seed=2)
    grads, losses = {}, {}
    noise = {True: torch.rand(700, 64, device=dev), False: torch.rand(700, 64, device=dev)}  # the jitter of each modality, whichever pass runs first
    for split in (False, True):
        torch.manual_seed(9)
        m = NeRFNetworkStatic(**kw).to(dev)
        step = RenderTrainStep(m, num_steps=64, scale=S.SCALE, ema_decay=None, split_backward=split)
        step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)
        real_render, real_rand = m.render, torch.rand

        def render(o, d, t, cal_lidar_color=False, _rr=real_render, **k):
            torch.rand = lambda *a, **kk: noise[bool(cal_lidar_color)]
            try:
                return _rr(o, d, t, cal_lidar_color=cal_lidar_color, **k)
            finally:
                torch.rand = real_rand
        m.render = render
        loss, parts, _ = step.step(batch)
        torch.cuda.synchronize()
        losses[split] = (float(loss), {k: float(v) for k, v in parts.items()})
        grads[split] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.numel() and p.grad is not None}
    assert set(losses[False][1]) == set(losses[True][1]) == {"depth", "raydrop", "intensity", "chamfer", "rgb"}
    assert abs(losses[False][0] - losses[True][0]) <= 1e-6 * abs(losses[False][0])
    assert set(grads[False]) == set(grads[True]) and len(grads[True]) == 6
    for n in grads[False]:
        a, b = grads[False][n], grads[True][n]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), n


def test_fused_loss_kernels_match_the_torch_expressions(dev):
    """LidarLossFn / MseSumFn (csrc/losses.hip: one launch each way) against the expressions of Trainer.train_step written with torch
    ops (trainer.py:187-219, 229-233, 491-503): values and the gradients that reach the rendered image / range, including the
    gradient that comes back through the chamfer point cloud and a range target equal to the prediction (|x| has gradient 0 at 0)."""
    from nvsf.nerf.train_step import LidarLossFn, MseSumFn
    g = torch.Generator(device="cpu").manual_seed(5)
    N = 1500
    rnd = lambda *s: torch.rand(*s, generator=g).to(dev)
    img, dep = rnd(1, N, 2).requires_grad_(), rnd(1, N).requires_grad_()
    gt_rd, gt_i, gt_d, dirs = (rnd(1, N) > 0.4).float(), rnd(1, N), rnd(1, N), torch.nn.functional.normalize(rnd(1, N, 3) - 0.5, dim=-1)
    with torch.no_grad():
        gt_d[0, :7] = dep[0, :7]  # zero residuals
    a_d, a_r, a_i, smooth, scale = 1.0, 0.01, 0.1, 0.1, 0.0125
    w = rnd(1, N, 3)  # stands for the chamfer gradient
    got = LidarLossFn.apply(img, dep, gt_rd, gt_i, gt_d, dirs, a_d, a_r, a_i, smooth, scale)
    total = 3.0 * got[0] + 5.0 * got[1] + 7.0 * got[2] + (got[4] * w).sum() + (got[3] * 0.25).sum()
    gi, gd = torch.autograd.grad(total, (img, dep))
    pd = dep * gt_rd
    ref = [(a_d * (pd - gt_d * gt_rd).abs()).sum(), (a_r * (img[:, :, 0] - gt_rd.clamp(smooth, 1 - smooth)) ** 2).sum(),
           (a_i * (img[:, :, 1] * gt_rd - gt_i * gt_rd) ** 2).sum()]
    pts, gpts = dirs * pd.unsqueeze(-1) / scale, dirs * (gt_d * gt_rd).unsqueeze(-1) / scale
    ri, rd_ = torch.autograd.grad(3.0 * ref[0] + 5.0 * ref[1] + 7.0 * ref[2] + (pts * w).sum() + (pd * 0.25).sum(), (img, dep))
    for a, b in zip(got[:3], ref):
        assert abs(float(a) - float(b)) <= 1e-5 * max(1.0, abs(float(b)))
    assert torch.equal(got[3], pd) and torch.allclose(got[4], pts, rtol=1e-6, atol=0) and torch.allclose(got[5], gpts, rtol=1e-6, atol=0)
    assert torch.allclose(gi, ri, rtol=1e-5, atol=1e-7) and torch.allclose(gd, rd_, rtol=1e-5, atol=1e-5 * float(rd_.abs().max()))
    a, b = rnd(1, N, 3).requires_grad_(), rnd(1, N, 3)
    l = MseSumFn.apply(a, b, 0.7)
    (ga,) = torch.autograd.grad(l * 2.0, a)
    lr = (0.7 * (a - b) ** 2).sum()
    (gr,) = torch.autograd.grad(lr * 2.0, a)
    assert abs(float(l) - float(lr)) <= 1e-5 * float(lr) and torch.allclose(ga, gr, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("lidar", [True, False])
def test_heads_on_shared_prefix_rows_equal_assembled_rows(dev, lidar, variants):
    """ops.heads with per-ray directions lets the MLP kernels read the ray's encoded direction from ONE row per ray
    (nvsf_mlp_fwd_prefix / nvsf_mlp_bwd_prefix) instead of from an assembled [M, in_cols] matrix (testing.variant(heads_input="rows")): the
    operands are the same fp16 values, so the logits are equal bit for bit; parameter and geometry gradients equal up to the
    order of the fp32 atomics of the weight gradients."""
    from nvsf import field_ops as ops, synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(3)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=12).to(dev)
    N, T = 37, 48
    g = torch.Generator(device="cpu").manual_seed(8)
    dirs01 = torch.rand(N, 3, generator=g).to(dev)
    geo0 = (torch.randn(N * T, 16, generator=g).to(dev))[:, 1:]  # the unaligned fp32 view the density MLP's output gives
    w = torch.randn(N * T, 2 if lidar else 3, generator=g).to(dev)
    nets = (m.raydrop_net, m.intensity_net) if lidar else (m.color_net,)
    res = {}
    for mode in ("prefix", "rows"):
        if mode == "rows":
            variants.set(heads_input="rows")
        geo = geo0.clone().requires_grad_()
        for n in nets:
            n.params.grad = None
        out = ops.heads(m, None, geo, lidar, ray_dirs01=dirs01)
        (out * w).sum().backward()
        res[mode] = (out.detach().clone(), geo.grad.clone(), [n.params.grad.clone() for n in nets])
    variants.clear("heads_input")
    assert torch.equal(res["prefix"][0], res["rows"][0])
    assert torch.allclose(res["prefix"][1], res["rows"][1], rtol=1e-5, atol=1e-6 * float(res["rows"][1].abs().max()))
    for a, b in zip(res["prefix"][2], res["rows"][2]):
        assert float(b.abs().max()) > 0 and float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


def test_step_without_the_chamfer_term_and_with_a_lidar_only_batch(dev):
    """Option branches of the step: chamfer_loss=False (the loss kernel then forms no point clouds), a LiDAR-only batch (no split, the
    camera parameters keep grad None and FusedAdam skips them) -- finite losses, the expected parts, parameters move."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=14)
    torch.manual_seed(6)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=300, T=32, seed=5)
    torch.manual_seed(7)
    m = NeRFNetworkStatic(**kw).to(dev)
    step = RenderTrainStep(m, num_steps=32, scale=S.SCALE, ema_decay=None, chamfer_loss=False)
    step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)
    loss, parts, _ = step.step(batch)
    assert set(parts) == {"depth", "raydrop", "intensity", "rgb"} and bool(torch.isfinite(loss))
    before = m.hash_encoder_camera.params.detach().clone()
    lidar_only = {k: v for k, v in batch.items() if k not in RenderTrainStep.CAMERA_KEYS}
    loss, parts, _ = step.step(lidar_only)
    torch.cuda.synchronize()
    assert set(parts) == {"depth", "raydrop", "intensity"} and bool(torch.isfinite(loss))
    assert m.hash_encoder_camera.params.grad is None and torch.equal(m.hash_encoder_camera.params, before)
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())


def test_fused_training_forward_equals_operator_chain(dev):
    """NeRFRenderer.run with gradients recorded takes the static field's one-launch training forward (ops.DensityRaysFn: sampler +
    unit-cube normalisation + hash grid + density MLP + exp in one kernel, its fp16 geometry rows handed to the heads as they are)
    instead of the operator chain (uniform_samples -> density -> DensityFn; `fused_train_forward = False`).  Same render, same loss,
    same gradients (the forward kernels share their arithmetic; table gradients differ by fp32 atomic order only), for the one-launch
    and for the level-sliced form of the kernel."""
    from nvsf import synthetic as S, field_ops as ops
    from nvsf.nerf import activation
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=15)
    torch.manual_seed(4)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=700, T=64, seed=2)
    noise = {True: torch.rand(700, 64, device=dev), False: torch.rand(700, 64, device=dev)}
    res = {}
    # False: operator chain; True: DensityRaysFn + compositor / heads nodes; "render": the whole forward as ONE node (ops.RenderRaysFn:
    # the evaluation render's kernels in their TRAIN form, the chain's backward kernels) -- what RenderTrainStep runs by default
    for fused in (False, True, "render"):
        torch.manual_seed(9)
        m = NeRFNetworkStatic(**kw).to(dev)
        with torch.no_grad():
            for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
                enc.params.normal_(0.0, 0.3)
        m.fused_train_forward = bool(fused)
        m.fused_train_render = fused == "render"
        step = RenderTrainStep(m, num_steps=64, scale=S.SCALE, ema_decay=None)
        step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)
        real_render, real_rand = m.render, torch.rand
        outs = {}

        def render(o, d, t, cal_lidar_color=False, _rr=real_render, **k):
            torch.rand = lambda *a, **kk: noise[bool(cal_lidar_color)]
            try:
                r = _rr(o, d, t, cal_lidar_color=cal_lidar_color, **k)
            finally:
                torch.rand = real_rand
            outs[bool(cal_lidar_color)] = {kk: v.detach().clone() for kk, v in r.items()}
            return r
        m.render = render
        loss, parts, _ = step.step(batch)
        torch.cuda.synchronize()
        res[fused] = (float(loss), outs, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.numel() and p.grad is not None})
    # the one-node render scans the transmittance 16 samples at a time and adds the direction-only part of the heads' first layer as
    # the MFMA's C operand (the evaluation kernels' arithmetic, pinned to the oracle at 1e-4): outputs agree with the chain to fp32
    # rounding, gradients to the rounding of the fp16 activations they pass through
    for mode, out_tol, grad_tol in ((True, 5e-6, 2e-5), ("render", 2e-5, 1e-4)):
        assert abs(res[mode][0] - res[False][0]) <= max(1e-6, out_tol) * abs(res[False][0]), mode
        for lidar in (True, False):
            for k, v in res[False][1][lidar].items():
                w = res[mode][1][lidar][k]
                assert v.shape == w.shape, k
                if k == "z_vals":
                    assert torch.equal(v, w)
                else:
                    assert float((v - w).abs().max()) <= out_tol * max(1.0, float(v.abs().max())), (mode, lidar, k)
        assert set(res[mode][2]) == set(res[False][2]) and len(res[mode][2]) == 6
        for n, a in res[False][2].items():
            b = res[mode][2][n]
            err = float((a - b).abs().max()) / float(a.abs().max())
            assert float(a.abs().max()) > 0 and err <= grad_tol, (mode, n, err)
            print(f"training forward {mode!r} vs chain: {n} max rel grad err {err:.2e}")
    # the level-sliced form of the training forward (what a camera batch of the full-size field takes): every output bit for bit
    # the one-launch form's
    torch.manual_seed(1)
    big = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH).to(dev)
    with torch.no_grad():
        big.hash_encoder_camera.params.normal_(0.0, 0.3)
    enc, net = big.hash_encoder_camera, big.sigma_net
    rng = np.random.default_rng(7)
    co, cd = S.camera_rays(300, rng)
    o, d = torch.from_numpy(co).to(dev), torch.from_numpy(cd).to(dev)
    from nvsf.nerf.raymarching import raymarching as rm
    nears, fars = rm.near_far_from_aabb(o, d, big.aabb_train, big.min_near)
    nz = torch.rand(300, 96, device=dev)
    got = {}
    for sliced in (False, True):
        z, sg, geo, g16 = ops.DensityRaysFn.apply(o, d, nears, fars, 96, big._aabb_host, float(big.bound), nz, enc.params, enc.table_f16(), enc.spec,
                                                  net.params, net.weights_f16(), net.spec, activation._LO, activation._HI, sliced)
        node = sg.grad_fn
        got[sliced] = (z, sg.detach(), geo.detach(), g16, [t.clone() for t in node.saved_tensors[:2]])
    for a, b in zip(got[False][:4], got[True][:4]):
        assert torch.equal(a, b)
    assert torch.equal(got[False][4][0], got[True][4][0]) and torch.equal(got[False][4][1], got[True][4][1])  # x01, feature rows
    # ... and they are what the stand-alone operators give: positions, encoder rows, MLP outputs
    zz, xyz = ops.uniform_samples(o, d, nears, fars, 96, big.aabb_train, nz)
    x01 = ((xyz.view(-1, 3) + big.bound) / (2 * big.bound))
    assert torch.equal(zz, got[False][0]) and torch.equal(x01, got[False][4][0])
    feat = ops.hashgrid_forward(x01, (0, 1, 2), enc.table_f16(), enc.spec)
    assert torch.equal(feat, got[False][4][1])
    h = ops.mlp_forward(feat, net.weights_f16(), net.spec)
    # (the fused kernels feed the 32 inputs of the first layer to the MFMA in their own column order -- the hardware's order of
    # additions inside a 32-wide k-step differs from the stand-alone kernel's: ~1e-4 relative, as between any two fp16 MLP kernels)
    assert float((h[:, 1:16] - got[False][2]).abs().max()) <= 2e-4 * float(h.abs().max())
    assert torch.equal(got[False][3][:, :15], got[False][2].half()) and bool((got[False][3][:, 15] == 1).all())


def test_overflowing_step_is_skipped_for_every_parameter_with_the_last_table_deferred(dev):
    """RenderTrainStep.step decides about an overflow from every gradient EXCEPT the table whose scatter is still running on the side
    stream, and issues that table's optimiser pass behind the scatter (loss_scaler.py: a non-finite table gradient implies a
    non-finite weight gradient of the density MLP, which is inspected).  With a loss scale far too large for fp16 the gradients
    overflow inside the MLP backward: NO parameter may move -- the deferred table included --, the scale halves, and the next steps
    (scale back in range) train normally.  Then: a normal step updates the deferred table exactly as a step without deferral does."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=14)
    torch.manual_seed(2)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=700, T=64)

    def make(defer):
        torch.manual_seed(6)
        m = NeRFNetworkStatic(**kw).to(dev)
        step = RenderTrainStep(m, num_steps=64, scale=S.SCALE, ema_decay=None)
        step.defer_last_table = defer
        return m, step

    m, step = make(True)
    assert step.defer_last_table
    step.scaler = _LossScaler()(init_scale=2.0 ** 60)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    step.step(batch)
    assert step._pending is not None and {id(p) for p in step._pending[1]} == {id(m.hash_encoder_lidar.params)}  # LiDAR goes last
    step.sync()
    torch.cuda.synchronize()
    assert not torch.isfinite(m.sigma_net.params.grad).all()  # the inspected gradient that stands in for the tables'
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n]), n  # skipped everywhere, the deferred table included
    assert step.scaler.get_scale() == 2.0 ** 59
    # same training with and without deferral (same seeds: same jitter), several steps
    results = []
    for defer in (True, False):
        m, step = make(defer)
        step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)
        for i in range(4):
            torch.manual_seed(100 + i)
            loss, _, _ = step.step(batch)
        step.sync()
        torch.cuda.synchronize()
        results.append(({n: p.detach().clone() for n, p in m.named_parameters()}, float(loss)))
    (pa, la), (pb, lb) = results
    # the table scatters and the MLP weight-gradient flushes add fp32 atomics in a run-dependent order; Adam divides by sqrt(v), so an
    # entry whose gradient is at the rounding level may move differently: equal up to that (same bar for two runs of ONE setting)
    assert abs(la - lb) <= 1e-5 * abs(la)
    for n in pa:
        d = (pa[n] - pb[n]).abs()
        if d.numel():
            assert float(d.mean()) <= 1e-6 and float((d <= 1e-4).float().mean()) >= 0.999, (n, float(d.mean()), float(d.max()))


def test_density_gradient_composed_equals_the_matrix_form_in_the_step(dev, variants):
    """RenderRaysFn.backward / _density_backward with the density network's logit gradient formed inside the MLP backward
    (nvsf_mlp_bwd_density; the second LiDAR head writes geometry-gradient rows of its own) against the sigma_geo pass + accumulating
    second head they replace: same gradients on every parameter (fp32 atomic order only)."""
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    from nvsf.nerf.train_step import RenderTrainStep
    kw = dict(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH, log2_hashmap_size=14)
    torch.manual_seed(2)
    teacher = NeRFNetworkStatic(**kw)
    with torch.no_grad():
        for enc in (teacher.hash_encoder_lidar, teacher.hash_encoder_camera):
            enc.params.normal_(0.0, 0.2)
    teacher = teacher.to(dev).eval()
    batch = _batch(S, teacher, dev, n=600, T=64)
    grads = {}
    for form in ("composed", "matrix"):
        variants.set(density_grad=form)
        torch.manual_seed(6)
        m = NeRFNetworkStatic(**kw).to(dev)
        step = RenderTrainStep(m, num_steps=64, scale=S.SCALE, ema_decay=None)
        step.scaler = _LossScaler()(init_scale=64.0, growth_interval=10 ** 6)
        torch.manual_seed(10)
        step.forward_backward(batch)
        torch.cuda.synchronize()
        grads[form] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    variants.clear("density_grad")
    assert set(grads["composed"]) == set(grads["matrix"]) and len(grads["matrix"]) == 6
    for n, a in grads["matrix"].items():
        b = grads["composed"][n]
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 1e-5 * scale, n


def test_loss_scaler_hold_growth_delays_the_doubling_by_one_update(dev):
    """a step whose deferred overflow check is still pending must not be the one that completes a growth interval
    (LossScaler.hold_growth: a device-side clamp of GradScaler's growth tracker before `update`)."""
    from nvsf.nerf.loss_scaler import LossScaler
    clean = torch.zeros(1, device=dev)
    plain, held = LossScaler(device=dev, init_scale=4.0, growth_interval=3), LossScaler(device=dev, init_scale=4.0, growth_interval=3)
    for sc in (plain, held):
        sc.update(clean)
        sc.update(clean)
    plain.update(clean)
    assert plain.get_scale() == 8.0             # third clean update in a row: doubled
    held.hold_growth()
    held.update(clean)
    assert held.get_scale() == 4.0              # held: this update cannot complete the interval ...
    held.update(clean)
    assert held.get_scale() == 8.0              # ... the next one does
    held.hold_growth()
    held.update(torch.ones(1, device=dev))
    assert held.get_scale() == 4.0              # an overflow halves as always
