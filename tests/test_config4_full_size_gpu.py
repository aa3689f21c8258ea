"""GPU: BASELINE config 4 at FULL size on the code path the benchmark times.

One multimodal training step of the default `RenderTrainStep` at 4096 LiDAR + 4096 camera rays x 768 samples with the 2^19-row tables
of config 2: at this size (M = 3.1 M ray-ordered rows >= 2^18, field_ops._bin_from) the production plan switches on -- binned table
scatter of the fine levels (run sums from `merge_from` on), level-major `[L, M, F]` hand-over of the
density MLP's input gradient, the 2.3-GB workspace, side-stream scatters into `.grad`, and the level-sliced (XCD-aware) training
forward for the camera batch.  The small-shape tests of tests/test_train_step_gpu.py stay below the switch and exercise the atomic
scatter; here the same quantities are checked where they are timed (reference: nvsf/nerf/trainer.py:153-219, 491-503).

  (a) the table gradient of each modality == the CPU oracle's scatter (oracle_hashgrid_bwd_f64: the sums in fp64) of exactly the
      positions and feature gradients the step handed to its scatter;
  (b) binned + level-major == the atomic variant (`table_scatter="atomic"`, every level through nvsf_hashgrid_bwd) on EVERY parameter;
  (c) the level-sliced training forward == the one-launch form, bit for bit, on every output the backward reads.
"""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _LossScaler():
    from nvsf.nerf.loss_scaler import LossScaler  # the step's loss scaler (GradScaler's rule and state_dict; nvsf/nerf/loss_scaler.py)
    return LossScaler

N_RAYS, T = 4096, 768


@pytest.fixture(scope="module")
def setup(dev):
    from nvsf import synthetic as S
    from nvsf.nerf.models.network_static import NeRFNetworkStatic
    torch.manual_seed(0)
    m = NeRFNetworkStatic(bound=S.BOUND, min_near=S.MIN_NEAR, min_near_lidar=S.MIN_NEAR, lidar_max_depth=S.LIDAR_MAX_DEPTH,
                          num_frames=S.NUM_FRAMES)
    with torch.no_grad():  # tables large enough that densities, weights and therefore gradients are not uniformly tiny
        g = torch.Generator().manual_seed(5)
        for enc in (m.hash_encoder_lidar, m.hash_encoder_camera):
            enc.params.copy_(torch.randn(enc.params.shape, generator=g) * 0.1)
    m = m.to(dev)
    rng = np.random.default_rng(0)
    lo, ld = S.lidar_rays(N_RAYS, rng)
    co, cd = S.camera_rays(N_RAYS, rng)
    g = torch.Generator(device="cpu").manual_seed(3)
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    batch = {"rays_o_lidar": t(lo), "rays_d_lidar": t(ld), "rays_o": t(co), "rays_d": t(cd), "time": torch.tensor([[0.5]], device=dev),
             "gt_depth": torch.rand(1, N_RAYS, generator=g).to(dev) * 0.5, "gt_raydrop": (torch.rand(1, N_RAYS, generator=g) > 0.3).float().to(dev),
             "gt_intensity": torch.rand(1, N_RAYS, generator=g).to(dev), "gt_rgb": torch.rand(1, N_RAYS, 3, generator=g).to(dev)}
    return S, m, batch


def _new_step(S, m):
    from nvsf.nerf.train_step import RenderTrainStep
    step = RenderTrainStep(m, num_steps=T, scale=S.SCALE)
    # GradScaler starts at 2^16 and halves on every overflowing step until the fp16 gradients fit; the comparison wants ONE step with
    # finite gradients, so it starts where the benchmark's scaler settles (2^7 = tcnn's default loss scale)
    step.scaler = _LossScaler()(init_scale=128.0)
    return step


def _grads(step, m, batch, seed=11):
    torch.manual_seed(seed)  # the sampler jitter (perturb=True) is drawn with torch.rand: same seed, same samples
    loss, parts, _ = step.forward_backward(batch)
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def test_full_size_step_runs_the_production_plan_and_matches_the_oracle_scatter(dev, setup, monkeypatch):
    from nvsf import field_ops as ops
    S, m, batch = setup
    step = _new_step(S, m)
    seen = []
    real = ops.hashgrid_backward

    def recording(x, cols, spec, grad_out, grad_table=None, fine_from=None, merge_from=None, **kw):
        # runs on the stream the scatter is issued on (the side stream): copies are ordered before it
        seen.append({"x": x.detach().clone(), "g": grad_out.detach().clone(), "fine_from": fine_from, "table": grad_table,
                     "stream": torch.cuda.current_stream().cuda_stream})
        return real(x, cols, spec, grad_out, grad_table=grad_table, fine_from=fine_from, merge_from=merge_from, **kw)

    monkeypatch.setattr(ops, "hashgrid_backward", recording)
    loss, grads = _grads(step, m, batch)
    assert np.isfinite(loss)
    assert len(seen) == 2  # camera first, then LiDAR (split_backward)
    main = torch.cuda.current_stream().cuda_stream
    for rec, enc in zip(seen, (m.hash_encoder_camera, m.hash_encoder_lidar)):
        spec = enc.spec
        # the plan the benchmark times: level-major gradient, run sums through the bins from level 8 on, atomics below, side stream
        assert rec["g"].dim() == 3 and tuple(rec["g"].shape) == (spec.L, N_RAYS * T, spec.F) and rec["g"].dtype == torch.float32
        assert rec["fine_from"] == (8, 16) == ops._bin_from(spec, N_RAYS * T, T)
        assert rec["stream"] != main
        assert rec["table"].data_ptr() == enc.params.grad.data_ptr()  # scattered straight into .grad
        x = rec["x"].cpu().numpy()
        g_rows = rec["g"].permute(1, 0, 2).reshape(N_RAYS * T, spec.L * spec.F).contiguous().cpu().numpy()
        ref = O.hashgrid_bwd_f64(x, (0, 1, 2), spec, g_rows)
        got = grads[[n for n, p in m.named_parameters() if p is enc.params][0]].double().cpu().numpy()
        assert np.isfinite(got).all()
        worst = 0.0
        for l in range(spec.L):
            a, b = spec.offsets[l] * spec.F, spec.offsets[l + 1] * spec.F
            scale = float(np.abs(ref[a:b]).max())
            assert scale > 0.0
            err = float(np.abs(got[a:b] - ref[a:b]).max()) / scale
            worst = max(worst, err)
            # fp32 atomics (levels 0-7) round per addend, the bins' fixed-point image (levels 8-15) truncates at 2^-40 of the level's
            # largest addend: both sit orders of magnitude below this bar
            assert err <= 2e-5, (l, err)
        print(f"table gradient vs fp64 oracle: worst level error {worst:.2e} of the level's largest entry")


def test_full_size_binned_plan_equals_the_atomic_variant_on_every_parameter(dev, setup, variants):
    S, m, batch = setup
    step = _new_step(S, m)
    loss_b, binned = _grads(step, m, batch)
    variants.set(table_scatter="atomic")
    loss_a, atomic = _grads(step, m, batch)
    variants.clear("table_scatter")
    assert loss_a == loss_b  # same forward, bit for bit
    assert set(binned) == set(atomic) == {n for n, p in m.named_parameters() if p.numel() > 0}  # (the direction encoders' .params are empty)
    for name in binned:
        a, b = atomic[name].double(), binned[name].double()
        assert torch.isfinite(b).all(), name
        scale = float(a.abs().max())
        assert scale > 0.0, name
        err = float((a - b).abs().max()) / scale
        # MLP weights: identical kernels on both sides except the layout dL/dx leaves in (same values) -> equal up to the order of
        # the fp32 atomics that sum the per-workgroup weight gradients; tables: fp32 atomics against fixed-point bins
        assert err <= 5e-5, (name, err)


def test_full_size_sliced_training_forward_is_bit_identical(dev, setup):
    from nvsf import field_ops as ops
    from nvsf.nerf.raymarching import raymarching
    S, m, batch = setup
    enc, net = m.hash_encoder_camera, m.sigma_net
    rays_o, rays_d = batch["rays_o"][0].contiguous(), batch["rays_d"][0].contiguous()
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, m.aabb_train, m.min_near)
    # the production choice at this shape is the sliced form for the camera batch, the one-launch form for LiDAR
    assert ops.prefer_sliced(enc.spec, N_RAYS, T, 2.0 * S.BOUND, float(S.BOUND)) is True
    assert ops.prefer_sliced(enc.spec, N_RAYS, T, float(S.LIDAR_MAX_DEPTH - S.MIN_NEAR), float(S.BOUND)) is False
    g = torch.Generator(device=dev).manual_seed(2)
    noise = torch.rand(N_RAYS, T, device=dev, generator=g)
    outs = [ops.density_uniform_train_forward(rays_o, rays_d, nears, fars, T, m._aabb_host, float(S.BOUND), noise, enc.table_f16(), enc.spec,
                                              net.weights_f16(), sliced) for sliced in (False, True)]
    torch.cuda.synchronize()
    for name, a, b in zip(("z_vals", "sigma", "geo16", "x01", "feat", "h32"), *outs):
        assert a.dtype == b.dtype and a.shape == b.shape
        ia = a.view(torch.int16 if a.dtype == torch.float16 else torch.int32)
        ib = b.view(torch.int16 if b.dtype == torch.float16 else torch.int32)
        assert torch.equal(ia, ib), name


@pytest.mark.parametrize("lidar", [False, True])
def test_full_size_training_render_sliced_equals_one_launch_and_the_evaluation_render(dev, setup, lidar):
    """ops.RenderRaysFn's forward (nvsf_render_uniform_train_fwd: what the default RenderTrainStep launches) at full size: the
    level-sliced form (camera batches) == the one-launch form bit for bit on every output and on everything kept for the backward,
    and image / depth / weights == the evaluation render's (nvsf_render_uniform_fwd), which is pinned to the oracle."""
    from nvsf import field_ops as ops
    from nvsf.nerf.raymarching import raymarching
    S, m, batch = setup
    enc, net = (m.hash_encoder_lidar, m.sigma_net) if lidar else (m.hash_encoder_camera, m.sigma_net)
    sfx = "_lidar" if lidar else ""
    rays_o, rays_d = batch["rays_o" + sfx][0].contiguous(), batch["rays_d" + sfx][0].contiguous()
    if lidar:
        nears = torch.full((N_RAYS,), float(m.min_near_lidar), device=dev)
        fars = torch.full((N_RAYS,), float(m.lidar_max_depth), device=dev)
        head_a, head_b, bg = m.raydrop_net.weights_f16(), m.intensity_net.weights_f16(), None
    else:
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, m.aabb_train, m.min_near)
        head_a, head_b, bg = m.color_net.weights_f16(), None, [1.0, 1.0, 1.0]
    g = torch.Generator(device=dev).manual_seed(4)
    noise = torch.rand(N_RAYS, T, device=dev, generator=g)
    outs = [ops.render_uniform_train_forward(rays_o, rays_d, nears, fars, T, m._aabb_host, float(S.BOUND), noise, enc.table_f16(), enc.spec,
                                             net.weights_f16(), lidar, head_a, head_b, m._k_scale(), bg, ops.W_THRESH, sliced) for sliced in (False, True)]
    torch.cuda.synchronize()
    names = ("z_vals", "weights", "weights_sum", "depth", "image", "x01", "feat", "geo16", "sigma", "rgbs")
    for name, a, b in zip(names, *outs):
        assert a.dtype == b.dtype and a.shape == b.shape
        bits = torch.int16 if a.dtype == torch.float16 else torch.int32
        assert torch.equal(a.view(bits), b.view(bits)), name
    ev = ops.render_uniform(rays_o, rays_d, nears, fars, T, m._aabb_host, float(S.BOUND), enc.table_f16(), enc.spec, net.weights_f16(), lidar, head_a,
                            head_b, m._k_scale(), bg, noise, sliced=not lidar)
    for name, a, b in zip(names[:5], outs[0][:5], ev):
        assert torch.equal(a, b), name
    assert float(outs[0][2].mean()) > 0.05 and bool((outs[0][9] > 0).any())  # not an empty render: some samples carry colour
