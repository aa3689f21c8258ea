"""ctypes access to the CPU oracle libraries (oracle/*.c) for tests, smoke() and bench.py's cpu_baseline.

TEST INFRASTRUCTURE: nothing under selfsupervised-nvsf_amd/ imports this module.
All arrays are numpy, C-contiguous; fp16 buffers are numpy float16.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_libs = {}


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, capture_output=True)


def _lib(name):
    if name not in _libs:
        path = os.path.join(ORACLE_DIR, f"liboracle_{name}.so")
        src = os.path.join(ORACLE_DIR, f"{name.replace('_fmad', '')}_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _libs[name] = ctypes.CDLL(path)
    return _libs[name]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


U, F = ctypes.c_uint32, ctypes.c_float


# ---- raymarching (a1-a9) ----------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    o, d, bb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    _lib("raymarching").oracle_near_far_from_aabb(_p(o), _p(d), _p(bb), U(N), F(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = o.shape[0]
    coords = np.empty((N, 2), np.float32)
    _lib("raymarching").oracle_sph_from_ray(_p(o), _p(d), F(radius), U(N), _p(coords))
    return coords


def morton3D(coords):
    c = _i32(coords)
    out = np.empty(c.shape[0], np.int32)
    _lib("raymarching").oracle_morton3D(_p(c), U(c.shape[0]), _p(out))
    return out


def morton3D_invert(indices):
    i = _i32(indices)
    out = np.empty((i.shape[0], 3), np.int32)
    _lib("raymarching").oracle_morton3D_invert(_p(i), U(i.shape[0]), _p(out))
    return out


def packbits(grid, thresh):
    g = _f32(grid).reshape(-1)
    N = g.size // 8
    out = np.empty(N, np.uint8)
    _lib("raymarching").oracle_packbits(_p(g), U(N), F(thresh), _p(out))
    return out


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, C, H, M, nears, fars, noises, counter=None, fmad=False):
    """fmad: the build of the same restatement whose multiply-adds are single-rounding fmaf (nvcc's default contraction)."""
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = o.shape[0]
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if counter is None else counter
    g = np.ascontiguousarray(grid, dtype=np.uint8)
    _lib("raymarching_fmad" if fmad else "raymarching").oracle_march_rays_train(_p(o), _p(d), _p(g), F(bound), F(dt_gamma), U(max_steps), U(N), U(C), U(H), U(M),
                                                _p(_f32(nears)), _p(_f32(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                                                _p(counter), _p(_f32(noises)))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh):
    s, c, dl, r = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = s.shape[0], r.shape[0]
    ws, dp, img = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    _lib("raymarching").oracle_composite_rays_train_forward(_p(s), _p(c), _p(dl), _p(r), U(M), U(N), F(T_thresh), _p(ws), _p(dp), _p(img))
    return ws, dp, img


def composite_rays_train_backward(g_ws, g_img, sigmas, rgbs, deltas, rays, ws, img, T_thresh):
    s, c, dl, r = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = s.shape[0], r.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    _lib("raymarching").oracle_composite_rays_train_backward(_p(_f32(g_ws)), _p(_f32(g_img)), _p(s), _p(c), _p(dl), _p(r), _p(_f32(ws)),
                                                             _p(_f32(img)), U(M), U(N), F(T_thresh), _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, noises, M=None):
    M = n_alive * n_step if M is None else M
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    g = np.ascontiguousarray(grid, dtype=np.uint8)
    _lib("raymarching").oracle_march_rays(U(n_alive), U(n_step), _p(_i32(rays_alive)), _p(_f32(rays_t)), _p(_f32(rays_o)), _p(_f32(rays_d)),
                                          F(bound), F(dt_gamma), U(max_steps), U(C), U(H), _p(g), _p(_f32(nears)), _p(_f32(fars)),
                                          _p(xyzs), _p(dirs), _p(deltas), _p(_f32(noises)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """In-place on copies; returns (rays_alive, rays_t, weights_sum, depth, image)."""
    ra, rt = _i32(rays_alive).copy(), _f32(rays_t).copy()
    ws, dp, img = _f32(weights_sum).copy(), _f32(depth).copy(), _f32(image).copy()
    _lib("raymarching").oracle_composite_rays(U(n_alive), U(n_step), F(T_thresh), _p(ra), _p(rt), _p(_f32(sigmas)), _p(_f32(rgbs)),
                                              _p(_f32(deltas)), _p(ws), _p(dp), _p(img))
    return ra, rt, ws, dp, img


# ---- field operators ----------------------------------------------------------------------------
def hashgrid_fwd(x, cols, table_f16, spec):
    """spec: object with D, L, F, scales, res, offsets (e.g. nvsf.field_ops.GridSpec or tests.specs.GridSpecNP)."""
    xx = _f32(x)
    M, xs = xx.shape
    out = np.empty((M, spec.L * spec.F), np.float16)
    cols_a = np.asarray(cols, np.uint32)
    t = np.ascontiguousarray(table_f16, dtype=np.float16)
    _lib("field").oracle_hashgrid_fwd(_p(xx), U(M), U(xs), _p(cols_a), U(spec.D), _p(t), U(spec.L), U(spec.F),
                                      _p(_f32(spec.scales)), _p(np.asarray(spec.res, np.uint32)), _p(np.asarray(spec.offsets, np.uint32)),
                                      _p(out), U(spec.L * spec.F))
    return out


def hashgrid_fwd_tcnn(x, cols, table_f16, spec):
    """tiny-cuda-nn's published accumulation (fp16 weight, one fp16-rounded fma per corner) instead of the specification's."""
    xx = _f32(x)
    M, xs = xx.shape
    out = np.empty((M, spec.L * spec.F), np.float16)
    _lib("field").oracle_hashgrid_fwd_tcnn(_p(xx), U(M), U(xs), _p(np.asarray(cols, np.uint32)), U(spec.D), _p(np.ascontiguousarray(table_f16)),
                                           U(spec.L), U(spec.F), _p(_f32(spec.scales)), _p(np.asarray(spec.res, np.uint32)),
                                           _p(np.asarray(spec.offsets, np.uint32)), _p(out), U(out.shape[1]))
    return out


def hashgrid_bwd(x, cols, spec, grad_out):
    xx, go = _f32(x), _f32(grad_out)
    M, xs = xx.shape
    gt = np.zeros(spec.offsets[-1] * spec.F, np.float32)
    _lib("field").oracle_hashgrid_bwd(_p(xx), U(M), U(xs), _p(np.asarray(cols, np.uint32)), U(spec.D), U(spec.L), U(spec.F),
                                      _p(_f32(spec.scales)), _p(np.asarray(spec.res, np.uint32)), _p(np.asarray(spec.offsets, np.uint32)),
                                      _p(go), U(go.shape[1]), _p(gt))
    return gt


def hashgrid_bwd_f64(x, cols, spec, grad_out):
    """The table gradient accumulated in fp64 (order-independent to ~1e-16): the bar for full-size batches."""
    xx, go = _f32(x), _f32(grad_out)
    M, xs = xx.shape
    gt = np.zeros(spec.offsets[-1] * spec.F, np.float64)
    _lib("field").oracle_hashgrid_bwd_f64(_p(xx), U(M), U(xs), _p(np.asarray(cols, np.uint32)), U(spec.D), U(spec.L), U(spec.F),
                                          _p(_f32(spec.scales)), _p(np.asarray(spec.res, np.uint32)), _p(np.asarray(spec.offsets, np.uint32)),
                                          _p(go), U(go.shape[1]), _p(gt))
    return gt


def freq_encode(x, n_freq=12):
    xx = _f32(x)
    M, nd = xx.shape
    out = np.empty((M, 2 * nd * n_freq), np.float16)
    _lib("field").oracle_freq_encode(_p(xx), U(M), U(nd), U(n_freq), _p(out), U(out.shape[1]))
    return out


def sh4_encode(d01):
    xx = _f32(d01)
    out = np.empty((xx.shape[0], 16), np.float16)
    _lib("field").oracle_sh4_encode(_p(xx), U(xx.shape[0]), _p(out), U(16))
    return out


def mlp_fwd(x, weights_f16, n_in, in_cols, n_hidden, hidden=64, out_cols=16, want_hidden=False):
    is_f16 = x.dtype == np.float16
    xx = np.ascontiguousarray(x)
    if not is_f16:
        xx = _f32(xx)
    M = xx.shape[0]
    out = np.empty((M, out_cols), np.float32)  # fp32 logits (the output layer is not rounded to fp16)
    hid = np.empty((M, n_hidden, hidden), np.float16) if want_hidden else None
    w = np.ascontiguousarray(weights_f16, dtype=np.float16)
    _lib("field").oracle_mlp_fwd(_p(xx), ctypes.c_int(1 if is_f16 else 0), U(M), U(n_in), U(xx.shape[1]), _p(w), U(in_cols), U(hidden),
                                 U(n_hidden), U(out_cols), _p(out), U(out_cols), _p(hid))
    return (out, hid) if want_hidden else out


def uniform_samples(rays_o, rays_d, nears, fars, lin, noise, aabb, want_xyz=True):
    o, d = _f32(rays_o), _f32(rays_d)
    N, T = o.shape[0], len(lin)
    z = np.empty((N, T), np.float32)
    xyz = np.empty((N, T, 3), np.float32) if want_xyz else None
    _lib("field").oracle_uniform_samples(_p(o), _p(d), _p(_f32(nears)), _p(_f32(fars)), _p(_f32(lin)),
                                         _p(_f32(noise)) if noise is not None else None, _p(_f32(aabb)), U(N), U(T), _p(z), _p(xyz))
    return z, xyz


def composite_uniform_weights(sigmas, z_vals, nears, fars, k_scale):
    s, z = _f32(sigmas), _f32(z_vals)
    N, T = s.shape
    w, ws, dp = np.empty((N, T), np.float32), np.empty(N, np.float32), np.empty(N, np.float32)
    _lib("field").oracle_composite_uniform_weights(_p(s), _p(z), _p(_f32(nears)), _p(_f32(fars)), U(N), U(T), F(k_scale), _p(w), _p(ws), _p(dp))
    return w, ws, dp


def composite_uniform_image(weights, rgbs, weights_sum, bg):
    w, c = _f32(weights), _f32(rgbs)
    N, T, C = c.shape
    img = np.empty((N, C), np.float32)
    _lib("field").oracle_composite_uniform_image(_p(w), _p(c), _p(_f32(weights_sum)), U(N), U(T), U(C),
                                                 _p(_f32(bg)) if bg is not None else None, _p(img))
    return img


def planes_fwd(xt, planes, res, want=3, C=8):
    """planes: list of 6*S arrays in the reference layout [1, C, H, W] (or [C, H, W]); res: [S][4].
    Returns (static, dynamic) fp32 [M, S*C] (None where not wanted)."""
    x = _f32(xt)
    M, S = x.shape[0], len(res)
    arrs = [np.ascontiguousarray(np.asarray(p, np.float32).reshape(C, *np.asarray(p).shape[-2:])) for p in planes]
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    r = np.ascontiguousarray(np.asarray(res, np.uint32).reshape(-1))
    out_s = np.empty((M, S * C), np.float32) if want & 1 else None
    out_d = np.empty((M, S * C), np.float32) if want & 2 else None
    _lib("field").oracle_planes_fwd(_p(x), U(M), ptrs, U(S), U(C), _p(r), ctypes.c_int(want), _p(out_s), _p(out_d))
    return out_s, out_d


# ---- composition: occupancy-grid render of the static field (BASELINE config 3) ---------------------
def _static_field_packed(xyz, dirs, bound, table_f16, spec, w_sigma, lidar, w_head_a, w_head_b):
    """sigma [M], rgb [M,3] for packed samples (all samples go through the heads; LiDAR = raydrop, intensity, 0)."""
    M = xyz.shape[0]
    if M == 0:
        return np.zeros(0, np.float32), np.zeros((0, 3), np.float32)
    x01 = ((_f32(xyz) + np.float32(bound)) * np.float32(1.0 / (2.0 * bound))).astype(np.float32)
    h = mlp_fwd(hashgrid_fwd(x01, (0, 1, 2), table_f16, spec), w_sigma, 32, 32, 1)
    sigma = np.exp(h[:, 0]).astype(np.float32)
    geo = h[:, 1:16].astype(np.float16)
    d01 = ((_f32(dirs) + 1.0) / 2.0).astype(np.float32)
    rgb = np.zeros((M, 3), np.float32)
    if lidar:
        logits = np.concatenate([freq_encode(d01), geo], 1)
        rgb[:, 0] = sigmoid_f32(mlp_fwd(logits, w_head_a, 87, 96, 2)[:, 0])
        rgb[:, 1] = sigmoid_f32(mlp_fwd(logits, w_head_b, 87, 96, 2)[:, 0])
    else:
        rgb[:] = sigmoid_f32(mlp_fwd(np.concatenate([sh4_encode(d01), geo], 1), w_head_a, 31, 32, 2)[:, :3])
    return sigma, rgb


def render_occupancy_train(rays_o, rays_d, nears, fars, bits, bound, C, H, max_steps, dt_gamma, noises, field, lidar, T_thresh=1e-4, bg=1.0):
    """march_rays_train -> field -> composite_rays_train (the training-mode sequence of raymarching.py:192-212, 296-306)."""
    N = rays_o.shape[0]
    xyz, dirs, deltas, rays, counter = march_rays_train(rays_o, rays_d, bits, bound, dt_gamma, max_steps, C, H, N * max_steps, nears, fars, noises)
    m = int(counter[0])
    sigma, rgb = _static_field_packed(xyz[:m], dirs[:m], bound, field[0], field[1], field[2], lidar, field[3], field[4])
    ws, dp, img = composite_rays_train_forward(sigma, rgb, deltas[:m], rays, T_thresh)
    if not lidar:
        img = img + (1.0 - ws)[:, None] * np.float32(bg)
    return dict(weights_sum=ws, depth=dp, image=img[:, :2] if lidar else img, n_samples=m, rays=rays)


def render_occupancy_infer(rays_o, rays_d, nears, fars, bits, bound, C, H, max_steps, dt_gamma, field, lidar, T_thresh=1e-4, bg=1.0):
    """Evaluation-mode loop: march_rays -> field -> composite_rays on the surviving rays (raymarching.py:389-409, 480-493)."""
    N = rays_o.shape[0]
    ws, dp, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive, rays_t = np.arange(N, dtype=np.int32), _f32(nears).copy()
    step = 0
    while step < max_steps:
        n_alive = len(alive)
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        M += 128 - M % 128
        xyz, dirs, deltas = march_rays(n_alive, n_step, alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, bits, nears, fars,
                                       np.zeros(n_alive, np.float32), M=M)
        sigma, rgb = _static_field_packed(xyz, dirs, bound, field[0], field[1], field[2], lidar, field[3], field[4])
        alive, rays_t, ws, dp, img = composite_rays(n_alive, n_step, T_thresh, alive, rays_t, sigma, rgb, deltas, ws, dp, img)
        alive = alive[alive >= 0]
        step += n_step
    if not lidar:
        img = img + (1.0 - ws)[:, None] * np.float32(bg)
    return dict(weights_sum=ws, depth=dp, image=img[:, :2] if lidar else img)


# ---- composition: the static-field uniform render (what NeRFNetworkStatic.render computes) --------
def sigmoid_f32(h):
    x = h.astype(np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x))).astype(np.float32)


def render_static(rays_o, rays_d, nears, fars, lin, noise, bound, table_f16, spec, w_sigma, lidar, w_head_a, w_head_b, bg,
                  k_scale=1.0, w_thresh=1e-4, tcnn_arith=False):
    """Returns dict(z_vals, sigmas, geo (fp16 [N,T,15]), weights, weights_sum, depth, image).
    tcnn_arith: the two places where the specification departs from tiny-cuda-nn's published arithmetic are
    switched to tiny-cuda-nn's: hash-grid corners accumulated in fp16 (hashgrid_fwd_tcnn) and every network output rounded to fp16
    (FullyFusedMLP returns __half) before trunc_exp / sigmoid / the heads read it -- with what follows in the reference's own graph:
    trunc_exp casts the half logit to fp32 (activation.py:9, under the Trainer's autocast), torch.sigmoid returns a half
    (network_dynamic.py:323-326).  NOT restated: FullyFusedMLP's half-precision WMMA accumulator fragments (hardware-defined)."""
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    z, xyz = uniform_samples(rays_o, rays_d, nears, fars, lin, noise, aabb)
    N, T = z.shape
    x01 = ((xyz.reshape(-1, 3) + np.float32(bound)) * np.float32(1.0 / (2.0 * bound))).astype(np.float32)
    feat = (hashgrid_fwd_tcnn if tcnn_arith else hashgrid_fwd)(x01, (0, 1, 2), table_f16, spec)
    h = mlp_fwd(feat, w_sigma, 32, 32, 1)
    if tcnn_arith:
        _net_out = lambda a: a.astype(np.float16).astype(np.float32)
        h = _net_out(h)
    else:
        _net_out = lambda a: a
    sigmas = np.exp(h[:, 0]).astype(np.float32).reshape(N, T)
    geo = h[:, 1:16].astype(np.float16)  # geometry features enter the heads as fp16 operands
    w, ws, dp = composite_uniform_weights(sigmas, z, nears, fars, k_scale)
    mask = (w > np.float32(w_thresh)).reshape(-1)
    d01 = ((_f32(rays_d) + 1.0) / 2.0).astype(np.float32)
    C = 2 if lidar else 3
    rgbs = np.zeros((N * T, C), np.float32)
    if mask.any():
        enc_ray = freq_encode(d01) if lidar else sh4_encode(d01)
        enc = np.repeat(enc_ray, T, axis=0)[mask]
        logits = np.concatenate([enc, geo[mask]], axis=1)
        if lidar:
            ra = mlp_fwd(logits, w_head_a, 87, 96, 2)[:, :1]
            it = mlp_fwd(logits, w_head_b, 87, 96, 2)[:, :1]
            hh = np.concatenate([ra, it], axis=1)
        else:
            hh = mlp_fwd(logits, w_head_a, 31, 32, 2)[:, :3]
        # the reference applies torch.sigmoid to the network's __half output (network_dynamic.py:323-326): a half result
        rgbs[mask] = _net_out(sigmoid_f32(_net_out(hh)))
    img = composite_uniform_image(w, rgbs.reshape(N, T, C), ws, None if lidar else bg)
    return dict(z_vals=z, sigmas=sigmas, geo=geo.reshape(N, T, 15), weights=w, weights_sum=ws, depth=dp, image=img,
                rgbs=rgbs.reshape(N, T, C))
