// Two small streaming kernels around the density MLP of the training path (gfx950), replacing chains of strided torch
// elementwise launches:
//   nvsf_sigma_geo_bwd   backward of  sigma = trunc_exp(h[:, 0]), geo_feat = h[:, 1:]  (network_dynamic.py:284-287 with
//                        activation.py:6-20): grad_h[:, 0] = grad_sigma * clamp(sigma, lo, hi) (lo / hi: the caller's fp32 e^-15 / e^15), grad_h[:, 1 + j] =
//                        grad_geo[:, j] -- one pass instead of two zero-fills, two strided copies, an add, a clamp and a mul
//   nvsf_cast_cols_f16   dst[:, j] = fp16(src[:, j]) between row-strided 2-D views (geometry features into the aligned
//                        input buffer of the heads, ops.HeadsFn)
//   nvsf_repeat_rows_f16 dst[n * T + t][:] = src[n][:]: a per-ray fp16 row (direction encoding) broadcast to the T samples of
//                        the ray -- the encoders then run once per ray instead of once per sample
// All are HBM-stream bound: 132 B per row, 6 B and 2 B (written) per element respectively.
#include "common.h"

namespace {
constexpr int kBlock = 256;
typedef float float4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void k_sigma_geo_bwd(const float* __restrict__ grad_sigma, const float* __restrict__ sigma,
                                                          const float* __restrict__ grad_geo, uint32_t gg_stride, uint32_t n_geo, uint32_t M,
                                                          float* __restrict__ grad_h, uint32_t gh_stride, float lo, float hi) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    float v[16];
    v[0] = grad_sigma ? grad_sigma[m] * fminf(fmaxf(sigma[m], lo), hi) : 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 15; ++j) v[1 + j] = (grad_geo && j < n_geo) ? grad_geo[(size_t)m * gg_stride + j] : 0.0f;
    float4_t* out = reinterpret_cast<float4_t*>(grad_h + (size_t)m * gh_stride);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = float4_t{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

template <bool SRC_F16>
__global__ __launch_bounds__(kBlock) void k_cast_cols(const void* __restrict__ src, uint32_t M, uint32_t n_cols, uint32_t src_stride,
                                                      _Float16* __restrict__ dst, uint32_t dst_stride) {
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (size_t)M * n_cols) return;
    const uint32_t m = (uint32_t)(idx / n_cols), c = (uint32_t)(idx - (size_t)m * n_cols);
    _Float16 v;
    if constexpr (SRC_F16) v = reinterpret_cast<const _Float16*>(src)[(size_t)m * src_stride + c];
    else v = (_Float16) reinterpret_cast<const float*>(src)[(size_t)m * src_stride + c];
    dst[(size_t)m * dst_stride + c] = v;
}

// one thread per (destination row, group of VEC halves)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_repeat_rows(const _Float16* __restrict__ src, uint32_t n_cols, uint32_t src_stride, uint32_t T,
                                                        unsigned long long total, _Float16* __restrict__ dst, uint32_t dst_stride) {
    const unsigned long long idx = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= total) return;
    const uint32_t groups = n_cols / VEC;
    const unsigned long long row = idx / groups;
    const uint32_t grp = (uint32_t)(idx - row * groups), n = (uint32_t)(row / T);
    typedef _Float16 vec_t __attribute__((ext_vector_type(VEC)));
    *reinterpret_cast<vec_t*>(dst + row * dst_stride + (size_t)grp * VEC) =
        *reinterpret_cast<const vec_t*>(src + (size_t)n * src_stride + (size_t)grp * VEC);
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_sigma_geo_bwd(const float* grad_sigma, const float* sigma, const float* grad_geo, uint32_t gg_stride, uint32_t n_geo,
                                uint32_t M, float* grad_h, uint32_t gh_stride, float sigma_lo, float sigma_hi, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(grad_h && gh_stride >= 16 && gh_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_h) & 15u) == 0);
    REQUIRE(n_geo <= 15 && (!grad_geo || gg_stride >= n_geo) && (!grad_sigma || sigma));
    hipLaunchKernelGGL(k_sigma_geo_bwd, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, stream, grad_sigma, sigma, grad_geo, gg_stride, n_geo, M,
                       grad_h, gh_stride, sigma_lo, sigma_hi);
    return nvsf_launch_status();
}

NVSF_API int nvsf_cast_cols_f16(const void* src, int src_is_f16, uint32_t M, uint32_t n_cols, uint32_t src_stride, void* dst_f16,
                                uint32_t dst_stride, hipStream_t stream) {
    if (M == 0 || n_cols == 0) return NVSF_OK;
    REQUIRE(src && dst_f16 && src_stride >= n_cols && dst_stride >= n_cols);
    const unsigned long long total = (unsigned long long)M * n_cols;
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f16);
    if (src_is_f16) hipLaunchKernelGGL(k_cast_cols<true>, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, src, M, n_cols, src_stride, dst, dst_stride);
    else hipLaunchKernelGGL(k_cast_cols<false>, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, src, M, n_cols, src_stride, dst, dst_stride);
    return nvsf_launch_status();
}

NVSF_API int nvsf_repeat_rows_f16(const void* src_f16, uint32_t N, uint32_t n_cols, uint32_t src_stride, uint32_t T, void* dst_f16,
                                  uint32_t dst_stride, hipStream_t stream) {
    if (N == 0 || T == 0 || n_cols == 0) return NVSF_OK;
    REQUIRE(src_f16 && dst_f16 && src_stride >= n_cols && dst_stride >= n_cols);
    const _Float16* src = reinterpret_cast<const _Float16*>(src_f16);
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f16);
    const bool vec8 = n_cols % 8 == 0 && src_stride % 8 == 0 && dst_stride % 8 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    const unsigned long long rows = (unsigned long long)N * T;
    if (vec8) {
        const unsigned long long total = rows * (n_cols / 8);
        REQUIRE((total + kBlock - 1) / kBlock < (1ull << 31));
        hipLaunchKernelGGL(k_repeat_rows<8>, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, src, n_cols, src_stride, T, total, dst, dst_stride);
    } else {
        const unsigned long long total = rows * n_cols;
        REQUIRE((total + kBlock - 1) / kBlock < (1ull << 31));
        hipLaunchKernelGGL(k_repeat_rows<1>, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, src, n_cols, src_stride, T, total, dst, dst_stride);
    }
    return nvsf_launch_status();
}
