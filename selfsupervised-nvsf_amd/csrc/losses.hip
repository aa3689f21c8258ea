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
