// Per-ray loss terms of the thin training step for gfx950.
//
// Trainer.train_step (nvsf/nerf/trainer.py:187-219, criteria with reduction="none" from main_nvsf.py:205-221, summed at
// trainer.py:540-543) forms the LiDAR terms from five [N] vectors with ~25 elementwise / reduction launches and autograd adds
// ~35 more on the way back; at N = 4096 rays every one of them is a launch latency.  Here the three sums, the masked range and
// the two point clouds the chamfer term compares are ONE single-workgroup launch (deterministic tree sums), and the gradients
// with respect to the rendered image / range are one elementwise launch:
//   gt_int = gt_intensity * gt_raydrop, gt_depth = gt_range * gt_raydrop            (trainer.py:187-189)
//   pred_rd = image[:, 0], pred_int = image[:, 1] * gt_raydrop, pred_depth = depth * gt_raydrop
//   L_depth = sum alpha_d |pred_depth - gt_depth|                                   (L1, trainer.py:193-197)
//   L_raydrop = sum alpha_r (pred_rd - clamp(gt_raydrop, s, 1 - s))^2               (trainer.py:209-215, s = smooth_factor)
//   L_intensity = sum alpha_i (pred_int - gt_int)^2                                 (trainer.py:199-207)
//   pred_pts = rays_d * pred_depth / scale, gt_pts = rays_d * gt_depth / scale      (trainer.py:229-233, inputs of chamfer_3DDist)
// and for the camera batch L_rgb = sum alpha_rgb (image - gt)^2 (trainer.py:491-503).
#include "common.h"

namespace {
constexpr int kBlock = 1024;

template <int K>
__device__ __forceinline__ void block_sums(float (&v)[K], float* __restrict__ out[K]) {
    __shared__ float part[K][kBlock / kWave];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) part[k][wave] = v[k];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = lane < kBlock / kWave ? part[k][lane] : 0.0f;
            s = wave_sum(s);
            if (lane == 0) *out[k] = s;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_lidar_losses_fwd(const float* __restrict__ image, const float* __restrict__ depth,
                                                             const float* __restrict__ gt_rd, const float* __restrict__ gt_i,
                                                             const float* __restrict__ gt_d, const float* __restrict__ rays_d, uint32_t N,
                                                             float alpha_d, float alpha_r, float alpha_i, float smooth, float scale,
                                                             float* __restrict__ l_depth, float* __restrict__ l_rd, float* __restrict__ l_int,
                                                             float* __restrict__ pred_depth, float* __restrict__ pred_pts,
                                                             float* __restrict__ gt_pts) {
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t n = threadIdx.x; n < N; n += kBlock) {
        const float m = gt_rd[n];
        const float gd = gt_d[n] * m, gi = gt_i[n] * m;
        const float pd = depth[n] * m, pr = image[2 * (size_t)n], pi = image[2 * (size_t)n + 1] * m;
        acc[0] += alpha_d * fabsf(pd - gd);
        const float tr = fminf(fmaxf(m, smooth), 1.0f - smooth);
        const float er = pr - tr, ei = pi - gi;
        acc[1] += alpha_r * (er * er);
        acc[2] += alpha_i * (ei * ei);
        pred_depth[n] = pd;
        if (pred_pts) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = rays_d[3 * (size_t)n + k];
                pred_pts[3 * (size_t)n + k] = d * pd / scale;
                gt_pts[3 * (size_t)n + k] = d * gd / scale;
            }
        }
    }
    float* out[3] = {l_depth, l_rd, l_int};
    block_sums<3>(acc, out);
}

__global__ __launch_bounds__(256) void k_lidar_losses_bwd(const float* __restrict__ image, const float* __restrict__ depth,
                                                          const float* __restrict__ gt_rd, const float* __restrict__ gt_i,
                                                          const float* __restrict__ gt_d, const float* __restrict__ rays_d, uint32_t N,
                                                          float alpha_d, float alpha_r, float alpha_i, float smooth, float scale,
                                                          const float* __restrict__ g_depth, const float* __restrict__ g_rd,
                                                          const float* __restrict__ g_int, const float* __restrict__ g_pred_depth,
                                                          const float* __restrict__ g_pred_pts, float* __restrict__ grad_image,
                                                          float* __restrict__ grad_depth) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    const float gD = g_depth ? *g_depth : 0.0f, gR = g_rd ? *g_rd : 0.0f, gI = g_int ? *g_int : 0.0f;
    const float m = gt_rd[n];
    const float gd = gt_d[n] * m, gi = gt_i[n] * m;
    const float pd = depth[n] * m, pr = image[2 * (size_t)n], pi = image[2 * (size_t)n + 1] * m;
    const float tr = fminf(fmaxf(m, smooth), 1.0f - smooth);
    // d|x|/dx = sign(x), 0 at 0 (torch.abs backward)
    const float e = pd - gd;
    const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
    float gpd = gD * alpha_d * sgn;  // gradient of the masked range
    if (g_pred_depth) gpd += g_pred_depth[n];
    if (g_pred_pts) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += g_pred_pts[3 * (size_t)n + k] / scale * rays_d[3 * (size_t)n + k];
        gpd += s;
    }
    grad_depth[n] = gpd * m;
    grad_image[2 * (size_t)n] = gR * alpha_r * 2.0f * (pr - tr);
    grad_image[2 * (size_t)n + 1] = gI * alpha_i * 2.0f * (pi - gi) * m;
}

__global__ __launch_bounds__(kBlock) void k_mse_sum_fwd(const float* __restrict__ a, const float* __restrict__ b, uint32_t n, float alpha,
                                                        float* __restrict__ out) {
    float acc[1] = {0.0f};
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
        const float e = a[i] - b[i];
        acc[0] += alpha * (e * e);
    }
    float* o[1] = {out};
    block_sums<1>(acc, o);
}

__global__ __launch_bounds__(256) void k_mse_sum_bwd(const float* __restrict__ a, const float* __restrict__ b, uint32_t n, float alpha,
                                                     const float* __restrict__ g, float* __restrict__ grad_a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) grad_a[i] = *g * alpha * 2.0f * (a[i] - b[i]);
}

// ---- structural regularisation on LiDAR patches: the `grad_loss` branch of Trainer.train_step (trainer.py:296-470) ----------------
// The batch is num_patch patches of pH x pW range-image pixels in row-major patch order (dataset_utils.py:407-503 draws them so):
// pixel j = (p pH + r) pW + c, inds[j] = h W + w.  With q = depth / scale (metres):
//   grad_x(j) = q(j) - q(j + 1)          for c < pW - 1,   grad_x(r, pW - 1) = grad_x(r, pW - 2)      (trainer.py:340-343, 381-384)
//   grad_y(j) = q(j) - q(j + pW)         for r < pH - 1,   grad_y(pH - 1, c) = grad_y(pH - 2, c)
// for the rendered and the true range alike.  The terms are masked by the true ray-drop channel and by the flatness of the TRUE
// range image around the pixel (trainer.py:386-431): first differences of the whole frame (same padding rule), their absolute second
// differences, mask = |second difference at (h, w)| < 0.05.  loss = alpha sum_j crit(grad_x m_x, gt_grad_x m_x) + the same in y
// (trainer.py:449-458: criterion with reduction "none", summed), crit in {L1, MSE, Huber(delta), SmoothL1(beta)} (main_nvsf.py:204-221);
// the cosine criterion and the Sobel gradients of the same block: sr_cos / sr_tap below.
struct PatchGeom {
    uint32_t pH, pW, H, W;
    float scale;
};

__device__ __forceinline__ float sr_crit(float e, int kind, float param) {
    const float a = fabsf(e);
    if (kind == 0) return a;                                                       // L1
    if (kind == 1) return e * e;                                                   // MSE
    if (kind == 2) return a <= param ? 0.5f * e * e : param * (a - 0.5f * param);  // Huber(delta)
    return a < param ? 0.5f * e * e / param : a - 0.5f * param;                    // SmoothL1(beta)
}
__device__ __forceinline__ float sr_crit_grad(float e, int kind, float param) {
    const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
    const float a = fabsf(e);
    if (kind == 0) return sgn;
    if (kind == 1) return 2.0f * e;
    if (kind == 2) return a <= param ? e : param * sgn;
    return a < param ? e / param : sgn;
}
// first difference of the frame's range channel along x at (h, w), in metres, with the reference's padding of the last column
__device__ __forceinline__ float pano_dx(const float* __restrict__ pano, uint32_t stride, uint32_t h, uint32_t w, const PatchGeom& g) {
    const uint32_t ww = w < g.W - 1 ? w : g.W - 2;
    const float* row = pano + (size_t)h * g.W * stride;
    return (row[(size_t)ww * stride] - row[(size_t)(ww + 1) * stride]) / g.scale;
}
__device__ __forceinline__ float pano_dy(const float* __restrict__ pano, uint32_t stride, uint32_t h, uint32_t w, const PatchGeom& g) {
    const uint32_t hh = h < g.H - 1 ? h : g.H - 2;
    return (pano[((size_t)hh * g.W + w) * stride] - pano[((size_t)(hh + 1) * g.W + w) * stride]) / g.scale;
}
// d grad_dir(j) / d q(k) for two pixels of ONE patch (0 where pixel k does not enter the gradient at pixel j):
//   manual differences: grad(j) = q(a) - q(b) with (a, b) = (j, j + step), in the last column / row the pair of the one before;
//   --sobel_grad: F.conv2d(patch, K, padding = 1) (trainer.py:316-328, 367-380): cross-correlation with
//                 Kx = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], Ky = Kx^T, zero padding at the PATCH border.
__device__ __forceinline__ float sr_tap(uint32_t rj, uint32_t cj, uint32_t rk, uint32_t ck, int dir, const PatchGeom& g, bool sobel) {
    const int dr = (int)rk - (int)rj, dc = (int)ck - (int)cj;
    if (sobel) {
        if (dr < -1 || dr > 1 || dc < -1 || dc > 1) return 0.0f;
        const int along = dir == 0 ? dc : dr, across = dir == 0 ? dr : dc;
        return (float)along * (across == 0 ? 2.0f : 1.0f);
    }
    if (dir == 0) {
        if (dr != 0) return 0.0f;
        const uint32_t a = cj < g.pW - 1 ? cj : cj - 1;
        return ck == a ? 1.0f : (ck == a + 1 ? -1.0f : 0.0f);
    }
    if (dc != 0) return 0.0f;
    const uint32_t a = rj < g.pH - 1 ? rj : rj - 1;
    return rk == a ? 1.0f : (rk == a + 1 ? -1.0f : 0.0f);
}
// gradient of the patch image q = v / scale at pixel j = (patch base + rj pW + cj)
__device__ __forceinline__ float sr_grad(const float* __restrict__ v, uint32_t base, uint32_t rj, uint32_t cj, int dir, const PatchGeom& g, bool sobel) {
    if (!sobel) {
        uint32_t a, b;
        if (dir == 0) { a = base + rj * g.pW + (cj < g.pW - 1 ? cj : cj - 1); b = a + 1; }
        else { a = base + (rj < g.pH - 1 ? rj : rj - 1) * g.pW + cj; b = a + g.pW; }
        return v[a] / g.scale - v[b] / g.scale;
    }
    float acc = 0.0f;  // row-major over the 3 x 3 window, zero taps skipped
    for (int dr = -1; dr <= 1; ++dr)
        for (int dc = -1; dc <= 1; ++dc) {
            const int r = (int)rj + dr, c = (int)cj + dc;
            if (r < 0 || c < 0 || r >= (int)g.pH || c >= (int)g.pW) continue;
            const float k = sr_tap(rj, cj, (uint32_t)r, (uint32_t)c, dir, g, true);
            if (k != 0.0f) acc += k * (v[base + (uint32_t)r * g.pW + (uint32_t)c] / g.scale);
        }
    return acc;
}
// mask of pixel j in direction dir: true ray-drop channel x flatness of the TRUE frame at the pixel (trainer.py:388-439)
__device__ __forceinline__ float sr_mask(const float* __restrict__ gt_rd, const long long* __restrict__ inds, const float* __restrict__ pano,
                                         uint32_t pano_stride, uint32_t j, int dir, const PatchGeom& g) {
    const uint32_t h = (uint32_t)(inds[j] / (long long)g.W), w = (uint32_t)(inds[j] % (long long)g.W);
    float second;
    if (dir == 0) {
        const uint32_t ww = w < g.W - 1 ? w : g.W - 2;  // padding of the second difference's last column
        second = fabsf(pano_dx(pano, pano_stride, h, ww, g)) - fabsf(pano_dx(pano, pano_stride, h, ww + 1, g));
    } else {
        const uint32_t hh = h < g.H - 1 ? h : g.H - 2;
        second = fabsf(pano_dy(pano, pano_stride, hh, w, g)) - fabsf(pano_dy(pano, pano_stride, hh + 1, w, g));
    }
    return gt_rd[j] * (fabsf(second) < 0.05f ? 1.0f : 0.0f);
}
struct SrTerm { float u, v, m; };  // masked gradient of the rendered range, of the true range, the mask
__device__ __forceinline__ SrTerm sr_term(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ gt_rd,
                                          const long long* __restrict__ inds, const float* __restrict__ pano, uint32_t pano_stride, uint32_t j,
                                          int dir, const PatchGeom& g, bool sobel) {
    const uint32_t area = g.pH * g.pW, base = j / area * area, rj = (j - base) / g.pW, cj = (j - base) % g.pW;
    SrTerm t;
    t.m = sr_mask(gt_rd, inds, pano, pano_stride, j, dir, g);
    t.u = sr_grad(pred, base, rj, cj, dir, g, sobel) * t.m;
    t.v = sr_grad(gt, base, rj, cj, dir, g, sobel) * t.m;
    return t;
}

// criterion 4, `--depth_grad_loss cos` (main_nvsf.py:211, trainer.py:442-452): per patch and direction cos = <u, v> / (max(|u|, eps)
// max(|v|, eps)) over the patch's pH pW masked gradients (torch.nn.CosineSimilarity(dim = 1, eps = 1e-8) on the flattened patch), and
// (1 - cos) expanded over the patch and summed: loss = alpha pH pW sum_patches (1 - cos_x) + (1 - cos_y).
// patch_stats [P, 6] = (<u,v>, <u,u>, <v,v>) for x, then for y -- written by the forward, read by the backward.
constexpr float kCosEps = 1e-8f;
__device__ __forceinline__ float sr_cos(const float* __restrict__ st) {
    return st[0] / (fmaxf(sqrtf(st[1]), kCosEps) * fmaxf(sqrtf(st[2]), kCosEps));
}

__global__ __launch_bounds__(kBlock) void k_lidar_grad_loss_fwd(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                const float* __restrict__ gt_rd, const long long* __restrict__ inds,
                                                                const float* __restrict__ pano, uint32_t pano_stride, uint32_t N, PatchGeom g,
                                                                int kind, float param, float alpha, int sobel, float* __restrict__ patch_stats,
                                                                float* __restrict__ loss) {
    float acc[1] = {0.0f};
    if (kind < 4) {
        for (uint32_t j = threadIdx.x; j < N; j += kBlock) {
            const SrTerm x = sr_term(pred, gt, gt_rd, inds, pano, pano_stride, j, 0, g, sobel != 0);
            const SrTerm y = sr_term(pred, gt, gt_rd, inds, pano, pano_stride, j, 1, g, sobel != 0);
            acc[0] += alpha * (sr_crit(x.u - x.v, kind, param) + sr_crit(y.u - y.v, kind, param));
        }
    } else {
        const uint32_t area = g.pH * g.pW;
        for (uint32_t p = threadIdx.x; p < N / area; p += kBlock) {
            float st[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t j = p * area; j < (p + 1) * area; ++j)
                for (int dir = 0; dir < 2; ++dir) {
                    const SrTerm t = sr_term(pred, gt, gt_rd, inds, pano, pano_stride, j, dir, g, sobel != 0);
                    st[3 * dir] += t.u * t.v;
                    st[3 * dir + 1] += t.u * t.u;
                    st[3 * dir + 2] += t.v * t.v;
                }
            for (int k = 0; k < 6; ++k) patch_stats[6 * (size_t)p + k] = st[k];
            acc[0] += alpha * (float)area * ((1.0f - sr_cos(st)) + (1.0f - sr_cos(st + 3)));
        }
    }
    float* o[1] = {loss};
    block_sums<1>(acc, o);
}

// d loss / d pred(k): pixel k enters the gradients of the pixels of its 3 x 3 neighbourhood inside its patch (sr_tap)
__global__ __launch_bounds__(256) void k_lidar_grad_loss_bwd(const float* __restrict__ pred, const float* __restrict__ gt,
                                                             const float* __restrict__ gt_rd, const long long* __restrict__ inds,
                                                             const float* __restrict__ pano, uint32_t pano_stride, uint32_t N, PatchGeom g,
                                                             int kind, float param, float alpha, int sobel, const float* __restrict__ patch_stats,
                                                             const float* __restrict__ g_loss, float* __restrict__ grad_pred) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= N) return;
    const uint32_t area = g.pH * g.pW, p = k / area, base = p * area, rk = (k - base) / g.pW, ck = (k - base) % g.pW;
    float sum = 0.0f;
    for (int dr = -1; dr <= 1; ++dr)
        for (int dc = -1; dc <= 1; ++dc) {
            const int r = (int)rk + dr, c = (int)ck + dc;
            if (r < 0 || c < 0 || r >= (int)g.pH || c >= (int)g.pW) continue;
            const uint32_t j = base + (uint32_t)r * g.pW + (uint32_t)c;
            for (int dir = 0; dir < 2; ++dir) {
                const float tap = sr_tap((uint32_t)r, (uint32_t)c, rk, ck, dir, g, sobel != 0);
                if (tap == 0.0f) continue;
                const SrTerm t = sr_term(pred, gt, gt_rd, inds, pano, pano_stride, j, dir, g, sobel != 0);
                float d;  // d loss / d u(j)
                if (kind < 4) {
                    d = alpha * sr_crit_grad(t.u - t.v, kind, param);
                } else {
                    const float* st = patch_stats + 6 * (size_t)p + 3 * dir;
                    const float nu_raw = sqrtf(st[1]), nu = fmaxf(nu_raw, kCosEps), nv = fmaxf(sqrtf(st[2]), kCosEps);
                    float dcos = t.v / (nu * nv);
                    if (nu_raw > kCosEps) dcos -= st[0] / (nu * nu * nv) * (t.u / nu_raw);
                    d = -alpha * (float)area * dcos;
                }
                sum += tap * d * t.m;
            }
        }
    grad_pred[k] = *g_loss * sum / g.scale;
}

// ---- error map of the pixel sampler (trainer.py:552-630) -----------------------------------------------------------------------
// per-ray LiDAR loss = alpha_d |.| + alpha_r (.)^2 + alpha_i (.)^2 (trainer.py:213-216, reduction "none") / per-ray camera loss =
// sum over channels of alpha_rgb (.)^2 (trainer.py:598); stats[0 .. 1] = (min, max) over the rays as ordered unsigned bit patterns
// (the losses are >= 0), to be initialised to (0x7f800000, 0) by the caller
__global__ __launch_bounds__(256) void k_lidar_ray_losses(const float* __restrict__ image, const float* __restrict__ depth, const float* __restrict__ gt_rd,
                                                          const float* __restrict__ gt_i, const float* __restrict__ gt_d, uint32_t N, float alpha_d,
                                                          float alpha_r, float alpha_i, float smooth, float* __restrict__ out, uint32_t* __restrict__ stats) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    const float m = gt_rd[n];
    const float gd = gt_d[n] * m, gi = gt_i[n] * m;
    const float pd = depth[n] * m, pr = image[2 * (size_t)n], pi = image[2 * (size_t)n + 1] * m;
    const float tr = fminf(fmaxf(m, smooth), 1.0f - smooth);
    const float er = pr - tr, ei = pi - gi;
    const float v = alpha_d * fabsf(pd - gd) + alpha_r * (er * er) + alpha_i * (ei * ei);
    out[n] = v;
    if (stats && v >= 0.0f) { atomicMin(stats, __float_as_uint(v)); atomicMax(stats + 1, __float_as_uint(v)); }
}
__global__ __launch_bounds__(256) void k_mse_rows(const float* __restrict__ a, const float* __restrict__ b, uint32_t N, uint32_t C, float alpha,
                                                  float* __restrict__ out, uint32_t* __restrict__ stats) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    float v = 0.0f;
    for (uint32_t k = 0; k < C; ++k) {
        const float e = a[(size_t)n * C + k] - b[(size_t)n * C + k];
        v += alpha * (e * e);
    }
    out[n] = v;
    if (stats && v >= 0.0f) { atomicMin(stats, __float_as_uint(v)); atomicMax(stats + 1, __float_as_uint(v)); }
}
// error = (loss - min) / (max - min + eps) * 999 + 1; cell = (floor(h eH / H), floor(w eW / W)); map[cell] = 0.1 map[cell] + 0.9 error.
// Several rays of a batch may fall into one cell: the reference's indexed assignment keeps an unspecified one of them (each computed
// from the OLD map value); here the ray with the LARGEST index wins (what a sequential execution of the assignment gives): pass 1
// leaves max(ray index + 1) per touched cell in `owner` (zero on entry), pass 2 lets that ray write and clears the slot again.
__global__ __launch_bounds__(256) void k_error_map_claim(const long long* __restrict__ inds, uint32_t N, uint32_t W, uint32_t eH, uint32_t eW,
                                                         float sh, float sw, uint32_t* __restrict__ owner) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    const uint32_t h = (uint32_t)(inds[n] / (long long)W), w = (uint32_t)(inds[n] % (long long)W);
    uint32_t ch = (uint32_t)((float)h * sh), cw = (uint32_t)((float)w * sw);
    ch = ch < eH ? ch : eH - 1;  // (a caller-side scale that rounds up: stay inside the map)
    cw = cw < eW ? cw : eW - 1;
    atomicMax(owner + (size_t)ch * eW + cw, n + 1u);
}
__global__ __launch_bounds__(256) void k_error_map_write(const float* __restrict__ loss, const long long* __restrict__ inds, uint32_t N,
                                                         uint32_t W, uint32_t eH, uint32_t eW, float sh, float sw, const uint32_t* __restrict__ stats,
                                                         uint32_t* __restrict__ owner, float* __restrict__ map) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    const uint32_t h = (uint32_t)(inds[n] / (long long)W), w = (uint32_t)(inds[n] % (long long)W);
    uint32_t ch = (uint32_t)((float)h * sh), cw = (uint32_t)((float)w * sw);
    ch = ch < eH ? ch : eH - 1;
    cw = cw < eW ? cw : eW - 1;
    const size_t cell = (size_t)ch * eW + cw;
    if (owner[cell] != n + 1u) return;
    const float lo = __uint_as_float(stats[0]), hi = __uint_as_float(stats[1]);
    float e = (loss[n] - lo) / (hi - lo + 1.1920928955078125e-07f);  // torch.finfo().eps
    e = e * (1000.0f - 1.0f) + 1.0f;
    map[cell] = 0.1f * map[cell] + 0.9f * e;
    owner[cell] = 0u;
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_lidar_losses_fwd(const float* image_lidar, const float* depth_lidar, const float* gt_raydrop, const float* gt_intensity,
                                   const float* gt_range, const float* rays_d, uint32_t N, float alpha_d, float alpha_r, float alpha_i,
                                   float smooth_factor, float scale, float* loss_depth, float* loss_raydrop, float* loss_intensity,
                                   float* pred_depth, float* pred_points, float* gt_points, hipStream_t stream) {
    REQUIRE(loss_depth && loss_raydrop && loss_intensity);
    if (N == 0) {
        if (hipMemsetAsync(loss_depth, 0, 4, stream) != hipSuccess || hipMemsetAsync(loss_raydrop, 0, 4, stream) != hipSuccess ||
            hipMemsetAsync(loss_intensity, 0, 4, stream) != hipSuccess)
            return (int)hipGetLastError();
        return NVSF_OK;
    }
    REQUIRE(image_lidar && depth_lidar && gt_raydrop && gt_intensity && gt_range && pred_depth);
    REQUIRE((pred_points == nullptr) == (gt_points == nullptr) && (!pred_points || (rays_d && scale != 0.0f)));
    hipLaunchKernelGGL(k_lidar_losses_fwd, dim3(1), dim3(kBlock), 0, stream, image_lidar, depth_lidar, gt_raydrop, gt_intensity, gt_range, rays_d, N,
                       alpha_d, alpha_r, alpha_i, smooth_factor, scale, loss_depth, loss_raydrop, loss_intensity, pred_depth, pred_points, gt_points);
    return nvsf_launch_status();
}

NVSF_API int nvsf_lidar_losses_bwd(const float* image_lidar, const float* depth_lidar, const float* gt_raydrop, const float* gt_intensity,
                                   const float* gt_range, const float* rays_d, uint32_t N, float alpha_d, float alpha_r, float alpha_i,
                                   float smooth_factor, float scale, const float* grad_loss_depth, const float* grad_loss_raydrop,
                                   const float* grad_loss_intensity, const float* grad_pred_depth, const float* grad_pred_points,
                                   float* grad_image_lidar, float* grad_depth_lidar, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(image_lidar && depth_lidar && gt_raydrop && gt_intensity && gt_range && grad_image_lidar && grad_depth_lidar);
    REQUIRE(!grad_pred_points || (rays_d && scale != 0.0f));
    hipLaunchKernelGGL(k_lidar_losses_bwd, dim3(cdiv(N, 256)), dim3(256), 0, stream, image_lidar, depth_lidar, gt_raydrop, gt_intensity, gt_range,
                       rays_d, N, alpha_d, alpha_r, alpha_i, smooth_factor, scale, grad_loss_depth, grad_loss_raydrop, grad_loss_intensity,
                       grad_pred_depth, grad_pred_points, grad_image_lidar, grad_depth_lidar);
    return nvsf_launch_status();
}

NVSF_API int nvsf_mse_sum_fwd(const float* a, const float* b, uint32_t n, float alpha, float* loss, hipStream_t stream) {
    REQUIRE(loss);
    if (n == 0) return hipMemsetAsync(loss, 0, 4, stream) == hipSuccess ? NVSF_OK : (int)hipGetLastError();
    REQUIRE(a && b);
    hipLaunchKernelGGL(k_mse_sum_fwd, dim3(1), dim3(kBlock), 0, stream, a, b, n, alpha, loss);
    return nvsf_launch_status();
}

NVSF_API int nvsf_mse_sum_bwd(const float* a, const float* b, uint32_t n, float alpha, const float* grad_loss, float* grad_a, hipStream_t stream) {
    if (n == 0) return NVSF_OK;
    REQUIRE(a && b && grad_loss && grad_a);
    hipLaunchKernelGGL(k_mse_sum_bwd, dim3(cdiv(n, 256)), dim3(256), 0, stream, a, b, n, alpha, grad_loss, grad_a);
    return nvsf_launch_status();
}

NVSF_API int nvsf_lidar_grad_loss_fwd(const float* pred_depth, const float* gt_depth, const float* gt_raydrop, const int64_t* pano_inds,
                                      const float* pano_range, uint32_t pano_stride, uint32_t N, uint32_t patch_h, uint32_t patch_w, uint32_t H,
                                      uint32_t W, float scale, int criterion, float criterion_param, float alpha, int sobel, float* patch_stats,
                                      float* loss, hipStream_t stream) {
    REQUIRE(loss);
    if (N == 0) return hipMemsetAsync(loss, 0, 4, stream) == hipSuccess ? NVSF_OK : (int)hipGetLastError();
    REQUIRE(pred_depth && gt_depth && gt_raydrop && pano_inds && pano_range && pano_stride >= 1);
    REQUIRE(patch_h >= 2 && patch_w >= 2 && N % (patch_h * patch_w) == 0 && H >= 2 && W >= 2 && scale != 0.0f);
    REQUIRE(criterion >= 0 && criterion <= 4 && (criterion < 2 || criterion == 4 || criterion_param > 0.0f) && (criterion != 4 || patch_stats));
    const PatchGeom g = {patch_h, patch_w, H, W, scale};
    hipLaunchKernelGGL(k_lidar_grad_loss_fwd, dim3(1), dim3(kBlock), 0, stream, pred_depth, gt_depth, gt_raydrop,
                       reinterpret_cast<const long long*>(pano_inds), pano_range, pano_stride, N, g, criterion, criterion_param, alpha, sobel,
                       patch_stats, loss);
    return nvsf_launch_status();
}

NVSF_API int nvsf_lidar_grad_loss_bwd(const float* pred_depth, const float* gt_depth, const float* gt_raydrop, const int64_t* pano_inds,
                                      const float* pano_range, uint32_t pano_stride, uint32_t N, uint32_t patch_h, uint32_t patch_w, uint32_t H,
                                      uint32_t W, float scale, int criterion, float criterion_param, float alpha, int sobel,
                                      const float* patch_stats, const float* grad_loss, float* grad_pred_depth, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(pred_depth && gt_depth && gt_raydrop && pano_inds && pano_range && pano_stride >= 1 && grad_loss && grad_pred_depth);
    REQUIRE(patch_h >= 2 && patch_w >= 2 && N % (patch_h * patch_w) == 0 && H >= 2 && W >= 2 && scale != 0.0f);
    REQUIRE(criterion >= 0 && criterion <= 4 && (criterion < 2 || criterion == 4 || criterion_param > 0.0f) && (criterion != 4 || patch_stats));
    const PatchGeom g = {patch_h, patch_w, H, W, scale};
    hipLaunchKernelGGL(k_lidar_grad_loss_bwd, dim3(cdiv(N, 256)), dim3(256), 0, stream, pred_depth, gt_depth, gt_raydrop,
                       reinterpret_cast<const long long*>(pano_inds), pano_range, pano_stride, N, g, criterion, criterion_param, alpha, sobel,
                       patch_stats, grad_loss, grad_pred_depth);
    return nvsf_launch_status();
}

NVSF_API int nvsf_lidar_ray_losses(const float* image_lidar, const float* depth_lidar, const float* gt_raydrop, const float* gt_intensity,
                                   const float* gt_range, uint32_t N, float alpha_d, float alpha_r, float alpha_i, float smooth_factor,
                                   float* ray_loss, uint32_t* min_max_bits, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(image_lidar && depth_lidar && gt_raydrop && gt_intensity && gt_range && ray_loss);
    hipLaunchKernelGGL(k_lidar_ray_losses, dim3(cdiv(N, 256)), dim3(256), 0, stream, image_lidar, depth_lidar, gt_raydrop, gt_intensity, gt_range, N,
                       alpha_d, alpha_r, alpha_i, smooth_factor, ray_loss, min_max_bits);
    return nvsf_launch_status();
}

NVSF_API int nvsf_mse_rows(const float* a, const float* b, uint32_t N, uint32_t C, float alpha, float* row_loss, uint32_t* min_max_bits,
                           hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(a && b && row_loss && C >= 1);
    hipLaunchKernelGGL(k_mse_rows, dim3(cdiv(N, 256)), dim3(256), 0, stream, a, b, N, C, alpha, row_loss, min_max_bits);
    return nvsf_launch_status();
}

NVSF_API int nvsf_error_map_update(const float* ray_loss, const int64_t* pixel_inds, uint32_t N, uint32_t W, float* error_map, uint32_t map_h,
                                   uint32_t map_w, float scale_h, float scale_w, const uint32_t* min_max_bits, uint32_t* owner, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(ray_loss && pixel_inds && error_map && min_max_bits && owner && W >= 1 && map_h >= 1 && map_w >= 1 && scale_h > 0.0f && scale_w > 0.0f);
    hipLaunchKernelGGL(k_error_map_claim, dim3(cdiv(N, 256)), dim3(256), 0, stream, reinterpret_cast<const long long*>(pixel_inds), N, W, map_h,
                       map_w, scale_h, scale_w, owner);
    hipLaunchKernelGGL(k_error_map_write, dim3(cdiv(N, 256)), dim3(256), 0, stream, ray_loss, reinterpret_cast<const long long*>(pixel_inds), N, W,
                       map_h, map_w, scale_h, scale_w, min_max_bits, owner, error_map);
    return nvsf_launch_status();
}
