// Uniform sampler and alpha compositor of NeRFRenderer.run (the path the reference actually executes:
// /root/reference/nvsf/nerf/models/renderer_dynamic.py:155-237) as gfx950 kernels.
//
// The reference materialises ~10 [N,T] / [N,T,3] fp32 intermediates with separate elementwise / cumprod
// torch kernels.  Here one 64-lane wave owns one ray: every round the wave reads 64 consecutive samples
// (one fully coalesced 256-B request per array), the transmittance is a cross-lane product scan carried
// from round to round, and weights / weights_sum / depth come out of a single pass over sigma.
// HBM traffic: 8 B read + 4 B written per sample in the weights kernel, 4*(1+C) B read in the image kernel.
#include "common.h"
#include <math.h>

namespace {
constexpr int kBlock = 256;
constexpr int kRaysPerBlock = kBlock / kWave;

// z = near + (far - near) * lin[i]  [+ (noise - 0.5) * (far - near) / T];  xyz = clip(o + d z, aabb)
__global__ __launch_bounds__(kBlock) void k_uniform_samples(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            const float* __restrict__ nears, const float* __restrict__ fars,
                                                            const float* __restrict__ lin, const float* __restrict__ noise,
                                                            const float* __restrict__ aabb, uint32_t N, uint32_t T,
                                                            float* __restrict__ z_vals, float* __restrict__ xyzs) {
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (size_t)N * T) return;
    const uint32_t n = (uint32_t)(idx / T), i = (uint32_t)(idx - (size_t)n * T);
    const float near = nears[n], range = fars[n] - near;
    float z = near + range * lin[i];
    if (noise) z = z + (noise[idx] - 0.5f) * (range / (float)T);
    z_vals[idx] = z;
    if (xyzs) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float p = rays_o[3 * (size_t)n + k] + rays_d[3 * (size_t)n + k] * z;
            xyzs[3 * idx + k] = fminf(fmaxf(p, aabb[k]), aabb[3 + k]);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_weights_fwd(const float* __restrict__ sigmas, const float* __restrict__ z_vals,
                                                        const float* __restrict__ nears, const float* __restrict__ fars,
                                                        uint32_t N, uint32_t T, float k_scale, float* __restrict__ weights,
                                                        float* __restrict__ weights_sum, float* __restrict__ depth) {
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    const float* s = sigmas + (size_t)n * T;
    const float* z = z_vals + (size_t)n * T;
    float* w_out = weights + (size_t)n * T;
    const float sample_dist = (fars[n] - nears[n]) / (float)T;
    float carry = 1.0f, ws = 0.0f, dp = 0.0f;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        float alpha = 0.0f, zi = 0.0f;
        if (valid) {
            zi = z[i];
            const float delta = (i + 1 < T) ? z[i + 1] - zi : sample_dist;
            alpha = 1.0f - expf(-delta * k_scale * s[i]);
        }
        const float om = valid ? (1.0f - alpha + 1e-15f) : 1.0f;
        const float incl = wave_scan_mul(om);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = alpha * (carry * excl);
        if (valid) w_out[i] = w;
        ws += w;
        dp += w * zi;
        carry = carry * __shfl(incl, 63, 64);
    }
    ws = wave_sum(ws);
    dp = wave_sum(dp);
    if (lane == 0) { weights_sum[n] = ws; depth[n] = dp; }
}

// d L / d sigma_i = delta_i k [ (1-a_i) G_i T_i - (1-a_i)/(1-a_i+eps) * sum_{j>i} G_j w_j ],
// G_i = gw_i + g_ws + g_depth z_i  (autograd of :185-194, :216-221)
__global__ __launch_bounds__(kBlock) void k_weights_bwd(const float* __restrict__ sigmas, const float* __restrict__ z_vals,
                                                        const float* __restrict__ nears, const float* __restrict__ fars,
                                                        const float* __restrict__ grad_weights, const float* __restrict__ grad_ws,
                                                        const float* __restrict__ grad_depth, uint32_t N, uint32_t T,
                                                        float k_scale, float* __restrict__ grad_sigmas) {
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    const float* s = sigmas + (size_t)n * T;
    const float* z = z_vals + (size_t)n * T;
    const float* gw = grad_weights ? grad_weights + (size_t)n * T : nullptr;
    const float g_ws = grad_ws ? grad_ws[n] : 0.0f, g_dp = grad_depth ? grad_depth[n] : 0.0f;
    const float sample_dist = (fars[n] - nears[n]) / (float)T;
    // pass A: total = sum_i G_i w_i
    float carry = 1.0f, total = 0.0f;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        float alpha = 0.0f, G = 0.0f;
        if (valid) {
            const float zi = z[i];
            const float delta = (i + 1 < T) ? z[i + 1] - zi : sample_dist;
            alpha = 1.0f - expf(-delta * k_scale * s[i]);
            G = (gw ? gw[i] : 0.0f) + g_ws + g_dp * zi;
        }
        const float incl = wave_scan_mul(valid ? (1.0f - alpha + 1e-15f) : 1.0f);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        total += G * alpha * (carry * excl);
        carry = carry * __shfl(incl, 63, 64);
    }
    total = wave_sum(total);
    // pass B
    carry = 1.0f;
    float prefix = 0.0f;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < T;
        float alpha = 0.0f, G = 0.0f, delta = 0.0f;
        if (valid) {
            const float zi = z[i];
            delta = (i + 1 < T) ? z[i + 1] - zi : sample_dist;
            alpha = 1.0f - expf(-delta * k_scale * s[i]);
            G = (gw ? gw[i] : 0.0f) + g_ws + g_dp * zi;
        }
        const float om = 1.0f - alpha;
        const float incl = wave_scan_mul(valid ? (om + 1e-15f) : 1.0f);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float Ti = carry * excl;
        const float P = prefix + wave_scan_add(G * alpha * Ti);
        if (valid) grad_sigmas[(size_t)n * T + i] = delta * k_scale * (om * G * Ti - (om / (om + 1e-15f)) * (total - P));
        carry = carry * __shfl(incl, 63, 64);
        prefix = __shfl(P, 63, 64);
    }
}

template <int C>
__global__ __launch_bounds__(kBlock) void k_image_fwd(const float* __restrict__ weights, const float* __restrict__ rgbs,
                                                      const float* __restrict__ weights_sum, uint32_t N, uint32_t T,
                                                      const float* __restrict__ bg, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (uint32_t i = lane; i < T; i += 64) {
        const float w = weights[(size_t)n * T + i];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += w * rgbs[((size_t)n * T + i) * C + c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
        const float rest = bg ? 1.0f - weights_sum[n] : 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) image[(size_t)n * C + c] = bg ? acc[c] + rest * bg[c] : acc[c];
    }
}

// grad_weights_i = sum_c g_c rgb_ic ; grad_rgb_ic = g_c w_i ; grad_ws = -sum_c g_c bg_c
template <int C>
__global__ __launch_bounds__(kBlock) void k_image_bwd(const float* __restrict__ weights, const float* __restrict__ rgbs,
                                                      const float* __restrict__ grad_image, uint32_t N, uint32_t T,
                                                      const float* __restrict__ bg, float* __restrict__ grad_weights,
                                                      float* __restrict__ grad_rgbs, float* __restrict__ grad_ws) {
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (size_t)N * T) return;
    const uint32_t n = (uint32_t)(idx / T);
    const float w = weights[idx];
    float gw = 0.0f, gb = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g = grad_image[(size_t)n * C + c];
        gw += g * rgbs[idx * C + c];
        if (grad_rgbs) grad_rgbs[idx * C + c] = g * w;
        if (bg) gb += g * bg[c];
    }
    if (grad_weights) grad_weights[idx] = gw;
    if (grad_ws && idx == (size_t)n * T) grad_ws[n] = -gb;
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_uniform_samples(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                  const float* lin, const float* noise, const float* aabb, uint32_t N, uint32_t T,
                                  float* z_vals, float* xyzs, hipStream_t stream) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(nears && fars && lin && z_vals);
    REQUIRE(!xyzs || (rays_o && rays_d && aabb));
    const unsigned long long total = (unsigned long long)N * T;
    REQUIRE(total < (1ull << 40));
    hipLaunchKernelGGL(k_uniform_samples, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, rays_o, rays_d, nears, fars, lin,
                       noise, aabb, N, T, z_vals, xyzs);
    return nvsf_launch_status();
}

NVSF_API int nvsf_composite_uniform_weights_fwd(const float* sigmas, const float* z_vals, const float* nears, const float* fars,
                                                uint32_t N, uint32_t T, float k_scale, float* weights, float* weights_sum,
                                                float* depth, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(T > 0 && sigmas && z_vals && nears && fars && weights && weights_sum && depth);
    hipLaunchKernelGGL(k_weights_fwd, dim3(cdiv(N, kRaysPerBlock)), dim3(kBlock), 0, stream, sigmas, z_vals, nears, fars, N, T,
                       k_scale, weights, weights_sum, depth);
    return nvsf_launch_status();
}

NVSF_API int nvsf_composite_uniform_weights_bwd(const float* sigmas, const float* z_vals, const float* nears, const float* fars,
                                                const float* grad_weights, const float* grad_weights_sum,
                                                const float* grad_depth, uint32_t N, uint32_t T, float k_scale,
                                                float* grad_sigmas, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(T > 0 && sigmas && z_vals && nears && fars && grad_sigmas);
    hipLaunchKernelGGL(k_weights_bwd, dim3(cdiv(N, kRaysPerBlock)), dim3(kBlock), 0, stream, sigmas, z_vals, nears, fars,
                       grad_weights, grad_weights_sum, grad_depth, N, T, k_scale, grad_sigmas);
    return nvsf_launch_status();
}

NVSF_API int nvsf_composite_uniform_image_fwd(const float* weights, const float* rgbs, const float* weights_sum, uint32_t N,
                                              uint32_t T, uint32_t C, const float* bg_color, float* image, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(T > 0 && weights && rgbs && image && (!bg_color || weights_sum));
    const dim3 grid(cdiv(N, kRaysPerBlock)), block(kBlock);
    switch (C) {
        case 1: hipLaunchKernelGGL(k_image_fwd<1>, grid, block, 0, stream, weights, rgbs, weights_sum, N, T, bg_color, image); break;
        case 2: hipLaunchKernelGGL(k_image_fwd<2>, grid, block, 0, stream, weights, rgbs, weights_sum, N, T, bg_color, image); break;
        case 3: hipLaunchKernelGGL(k_image_fwd<3>, grid, block, 0, stream, weights, rgbs, weights_sum, N, T, bg_color, image); break;
        case 4: hipLaunchKernelGGL(k_image_fwd<4>, grid, block, 0, stream, weights, rgbs, weights_sum, N, T, bg_color, image); break;
        default: return NVSF_ERR_UNSUPPORTED;
    }
    return nvsf_launch_status();
}

NVSF_API int nvsf_composite_uniform_image_bwd(const float* weights, const float* rgbs, const float* grad_image, uint32_t N,
                                              uint32_t T, uint32_t C, const float* bg_color, float* grad_weights,
                                              float* grad_rgbs, float* grad_weights_sum, hipStream_t stream) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(weights && rgbs && grad_image);
    const dim3 grid(cdiv((unsigned long long)N * T, kBlock)), block(kBlock);
    switch (C) {
        case 1: hipLaunchKernelGGL(k_image_bwd<1>, grid, block, 0, stream, weights, rgbs, grad_image, N, T, bg_color, grad_weights, grad_rgbs, grad_weights_sum); break;
        case 2: hipLaunchKernelGGL(k_image_bwd<2>, grid, block, 0, stream, weights, rgbs, grad_image, N, T, bg_color, grad_weights, grad_rgbs, grad_weights_sum); break;
        case 3: hipLaunchKernelGGL(k_image_bwd<3>, grid, block, 0, stream, weights, rgbs, grad_image, N, T, bg_color, grad_weights, grad_rgbs, grad_weights_sum); break;
        case 4: hipLaunchKernelGGL(k_image_bwd<4>, grid, block, 0, stream, weights, rgbs, grad_image, N, T, bg_color, grad_weights, grad_rgbs, grad_weights_sum); break;
        default: return NVSF_ERR_UNSUPPORTED;
    }
    return nvsf_launch_status();
}
