// Direction encodings for gfx950: tcnn "Frequency" (network_dynamic.py:108-114, 3 -> 72) and
// "SphericalHarmonics" degree 4 (network_dynamic.py:165-170, 3 -> 16).  fp32 in, fp16 out.
// Specification: DESIGN.md section 4.2.  One thread per (sample, output pair): both are tiny,
// launch-latency-bound operators; in the fused render path they are evaluated once per RAY instead.
#include "encodings_device.h"

namespace {
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_freq(const float* __restrict__ x, uint32_t M, uint32_t n_dims, uint32_t n_freq,
                                                 _Float16* __restrict__ out, uint32_t out_stride) {
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t per_row = n_dims * n_freq;
    if (idx >= (size_t)M * per_row) return;
    const uint32_t m = (uint32_t)(idx / per_row), r = (uint32_t)(idx - (size_t)m * per_row);
    const uint32_t i = r / n_freq, k = r - i * n_freq;
    float s, c;
    freq_pair(x[(size_t)m * n_dims + i], (int)k, s, c);
    h2_pair_t p;
    p[0] = (_Float16)s;
    p[1] = (_Float16)c;
    *reinterpret_cast<h2_pair_t*>(out + (size_t)m * out_stride + 2 * r) = p;
}

// The same values, one thread per 16-byte piece of a row (four consecutive frequencies of one input dimension = eight fp16): a wave
// stores 1 KB of consecutive bytes per instruction where the pair form stores 256 B, and the input coordinate is read once per
// four pairs.  n_freq a multiple of 4, rows 16-byte aligned.
__global__ __launch_bounds__(kBlock) void k_freq_x8(const float* __restrict__ x, uint32_t M, uint32_t n_dims, uint32_t n_freq,
                                                    _Float16* __restrict__ out, uint32_t out_stride) {
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t per_row = n_dims * n_freq / 4u;  // pieces per row
    if (idx >= (size_t)M * per_row) return;
    const uint32_t m = (uint32_t)(idx / per_row), piece = (uint32_t)(idx - (size_t)m * per_row);
    const uint32_t quads = n_freq / 4u, i = piece / quads, k0 = 4u * (piece - i * quads);
    const float xi = x[(size_t)m * n_dims + i];
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    h8v v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s, c;
        freq_pair(xi, (int)k0 + j, s, c);
        v[2 * j] = (_Float16)s;
        v[2 * j + 1] = (_Float16)c;
    }
    *reinterpret_cast<h8v*>(out + (size_t)m * out_stride + 8u * piece) = v;
}

__global__ __launch_bounds__(kBlock) void k_sh4(const float* __restrict__ d01, uint32_t M, _Float16* __restrict__ out,
                                                uint32_t out_stride) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    float o[16];
    sh4_basis(d01[3 * (size_t)m], d01[3 * (size_t)m + 1], d01[3 * (size_t)m + 2], o);
#pragma unroll
    for (int j = 0; j < 16; ++j) out[(size_t)m * out_stride + j] = (_Float16)o[j];
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_freq_encode(const float* x, uint32_t M, uint32_t n_dims, uint32_t n_freq, void* out_f16, uint32_t out_stride,
                              hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && out_f16 && n_dims >= 1 && n_freq >= 1 && n_freq <= 24);
    REQUIRE(out_stride >= 2 * n_dims * n_freq && out_stride % 2 == 0 && (reinterpret_cast<uintptr_t>(out_f16) & 3u) == 0);
    const unsigned long long total = (unsigned long long)M * n_dims * n_freq;
    if (n_freq % 4u == 0u && out_stride % 8u == 0u && (reinterpret_cast<uintptr_t>(out_f16) & 15u) == 0) {
        hipLaunchKernelGGL(k_freq_x8, dim3(cdiv(total / 4ull, kBlock)), dim3(kBlock), 0, stream, x, M, n_dims, n_freq,
                           reinterpret_cast<_Float16*>(out_f16), out_stride);
        return nvsf_launch_status();
    }
    hipLaunchKernelGGL(k_freq, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, x, M, n_dims, n_freq,
                       reinterpret_cast<_Float16*>(out_f16), out_stride);
    return nvsf_launch_status();
}

NVSF_API int nvsf_sh4_encode(const float* dirs01, uint32_t M, void* out_f16, uint32_t out_stride, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(dirs01 && out_f16 && out_stride >= 16);
    hipLaunchKernelGGL(k_sh4, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, stream, dirs01, M, reinterpret_cast<_Float16*>(out_f16),
                       out_stride);
    return nvsf_launch_status();
}
