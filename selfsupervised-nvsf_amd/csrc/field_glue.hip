// Two small streaming kernels around the density MLP of the training path (gfx950), replacing chains of strided torch
// elementwise launches:
//   nvsf_sigma_geo_bwd   backward of  sigma = trunc_exp(h[:, 0]), geo_feat = h[:, 1:]  (network_dynamic.py:284-287 with
//                        activation.py:6-20): grad_h[:, 0] = grad_sigma * clamp(sigma, lo, hi) (lo / hi: the caller's fp32 e^-15 / e^15), grad_h[:, 1 + j] =
//                        grad_geo[:, j] -- one pass instead of two zero-fills, two strided copies, an add, a clamp and a mul
//   nvsf_cast_cols_f16   dst[:, j] = fp16(src[:, j]) between row-strided 2-D views (geometry features into the aligned
//                        input buffer of the heads, ops.HeadsFn)
//   nvsf_heads_input_f16 whole input rows of the per-sample heads, [ray's direction encoding | geometry features | ones], in one
//                        pass of 16-byte stores (what the two kernels below do in two passes that each leave partial lines)
//   nvsf_masked_sigmoid  out = mask ? sigmoid(logits) : 0 on a strided logits view (network_dynamic.py:325-330 for the dense case)
//   nvsf_repeat_rows_f16 dst[n * T + t][:] = src[n][:]: a per-ray fp16 row (direction encoding) broadcast to the T samples of
//                        the ray -- the encoders then run once per ray instead of once per sample
// All are HBM-stream bound: 132 B per row, 6 B and 2 B (written) per element respectively.
#include "common.h"

namespace {
constexpr int kBlock = 256;
typedef float float4_t __attribute__((ext_vector_type(4)));

// general form: one thread per row
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

// 16-float, 16-byte aligned geometry-gradient rows (what ops.HeadsFn hands back): four lanes per row, one 16-byte load and
// one 16-byte store each; the shift by one column comes from the left neighbour's last element (wave shuffle).
__global__ __launch_bounds__(kBlock) void k_sigma_geo_bwd_rows16(const float* __restrict__ grad_sigma, const float* __restrict__ sigma,
                                                                 const float* __restrict__ grad_geo, uint32_t gg_stride, uint32_t n_geo,
                                                                 uint32_t M, float* __restrict__ grad_h, uint32_t gh_stride, float lo, float hi) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t m_raw = idx >> 2, q = idx & 3u;
    const uint32_t m = m_raw < M ? m_raw : M - 1;
    float4_t v = *reinterpret_cast<const float4_t*>(grad_geo + (size_t)m * gg_stride + 4 * q);
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e)
        if (4 * q + e >= n_geo) v[e] = 0.0f;  // alignment padding of the source row
    const float prev = __shfl_up(v[3], 1);
    float first = prev;
    if (q == 0) first = grad_sigma ? grad_sigma[m] * fminf(fmaxf(sigma[m], lo), hi) : 0.0f;
    if (m_raw < M) *reinterpret_cast<float4_t*>(grad_h + (size_t)m * gh_stride + 4 * q) = float4_t{first, v[0], v[1], v[2]};
}

// PITCH lanes per row (PITCH = power of two >= n_cols): a wave reads whole rows side by side
template <bool SRC_F16, int PITCH>
__global__ __launch_bounds__(kBlock) void k_cast_cols(const void* __restrict__ src, uint32_t M, uint32_t n_cols, uint32_t src_stride,
                                                      _Float16* __restrict__ dst, uint32_t dst_stride) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    uint32_t m, c;
    if constexpr (PITCH > 0) { m = idx / PITCH; c = idx % PITCH; }
    else { m = idx / n_cols; c = idx - m * n_cols; }
    if (m >= M || c >= n_cols) return;
    _Float16 v;
    if constexpr (SRC_F16) v = reinterpret_cast<const _Float16*>(src)[(size_t)m * src_stride + c];
    else v = (_Float16) reinterpret_cast<const float*>(src)[(size_t)m * src_stride + c];
    dst[(size_t)m * dst_stride + c] = v;
}

// a workgroup belongs to one source row n (blockIdx.x / blocks_per_row: scalar arithmetic); its threads cover
// (t, group of VEC halves) of that row's T copies
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_repeat_rows(const _Float16* __restrict__ src, uint32_t groups, uint32_t src_stride, uint32_t T,
                                                        uint32_t blocks_per_row, _Float16* __restrict__ dst, uint32_t dst_stride) {
    const uint32_t n = blockIdx.x / blocks_per_row, b = blockIdx.x - n * blocks_per_row;
    const uint32_t i = b * kBlock + threadIdx.x;
    const uint32_t t = i / groups, grp = i - t * groups;
    if (t >= T) return;
    typedef _Float16 vec_t __attribute__((ext_vector_type(VEC)));
    *reinterpret_cast<vec_t*>(dst + ((size_t)n * T + t) * dst_stride + (size_t)grp * VEC) =
        *reinterpret_cast<const vec_t*>(src + (size_t)n * src_stride + (size_t)grp * VEC);
}

// Whole input rows of the per-sample heads in one pass: [direction encoding of the sample's ray | geometry features | ones]
// as 16-byte stores (a row written in pieces by different launches reaches HBM as partial lines).  A workgroup belongs to one
// ray; a thread owns (sample t of the ray, group of 8 columns).
template <bool GEO_F16>
__global__ __launch_bounds__(kBlock) void k_heads_input(const _Float16* __restrict__ enc_ray, uint32_t n_enc, uint32_t enc_stride, uint32_t T,
                                                        uint32_t blocks_per_ray, const void* __restrict__ geo, uint32_t n_geo, uint32_t geo_stride,
                                                        int geo_vec, _Float16* __restrict__ dst, uint32_t in_cols, uint32_t dst_stride) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    const uint32_t n = blockIdx.x / blocks_per_ray, b = blockIdx.x - n * blocks_per_ray;
    const uint32_t groups = in_cols / 8;
    const uint32_t i = b * kBlock + threadIdx.x;
    const uint32_t t = i / groups, gi = i - t * groups;
    if (t >= T) return;
    const size_t m = (size_t)n * T + t;
    const uint32_t col0 = 8 * gi;
    half8_t v;
    if (col0 < n_enc) {  // n_enc is a multiple of 8: a group is encoding or it is not
        v = *reinterpret_cast<const half8_t*>(enc_ray + (size_t)n * enc_stride + col0);
    } else {
        const uint32_t j0 = col0 - n_enc;
        float e[8];
        if (j0 < n_geo) {
            if constexpr (GEO_F16) {
                const _Float16* g = reinterpret_cast<const _Float16*>(geo) + m * geo_stride + j0;
                if (geo_vec) {
                    const half8_t h = *reinterpret_cast<const half8_t*>(g);
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = (float)h[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = (float)g[j0 + k < n_geo ? k : 0];
                }
            } else {
                const float* g = reinterpret_cast<const float*>(geo) + m * geo_stride + j0;
                if (geo_vec == 1) {
                    const float4_t a = *reinterpret_cast<const float4_t*>(g), c = *reinterpret_cast<const float4_t*>(g + 4);
                    e[0] = a[0]; e[1] = a[1]; e[2] = a[2]; e[3] = a[3]; e[4] = c[0]; e[5] = c[1]; e[6] = c[2]; e[7] = c[3];
                } else if (geo_vec == 2) {  // rows start one float past a 16-byte boundary (columns 1.. of an aligned matrix:
                                            // geo_feat = h[:, 1:]): aligned loads from the boundary, elements shifted by one
                    const float* gb = g - 1;
                    const float4_t a = *reinterpret_cast<const float4_t*>(gb), c = *reinterpret_cast<const float4_t*>(gb + 4);
                    float4_t d = {0, 0, 0, 0};
                    if (j0 + 8 < n_geo) d = *reinterpret_cast<const float4_t*>(gb + 8);
                    e[0] = a[1]; e[1] = a[2]; e[2] = a[3]; e[3] = c[0]; e[4] = c[1]; e[5] = c[2]; e[6] = c[3]; e[7] = d[0];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = g[j0 + k < n_geo ? k : 0];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (j0 + k < n_geo) ? (_Float16)e[k] : (_Float16)1.0f;
    }
    *reinterpret_cast<half8_t*>(dst + m * dst_stride + col0) = v;
}

// out[m][c] = mask[m] ? sigmoid(logits[m][c]) : 0 for logits with arbitrary row / column strides (the heads' logits are a strided
// view of their 16-column output blocks); thread = (row, column).
__global__ __launch_bounds__(kBlock) void k_masked_sigmoid(const float* __restrict__ logits, uint32_t row_stride, uint32_t col_stride,
                                                           const unsigned char* __restrict__ mask, uint32_t M, uint32_t C,
                                                           float* __restrict__ out) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t m = idx / C, c = idx - m * C;
    if (m >= M) return;
    float v = 0.0f;
    if (!mask || mask[m]) v = 1.0f / (1.0f + expf(-logits[(size_t)m * row_stride + (size_t)c * col_stride]));
    out[idx] = v;
}

// grad_in = (grad_out * (1 - out)) * out: sigmoid's backward from its output (what aten::sigmoid_backward computes), one pass
__global__ __launch_bounds__(kBlock) void k_sigmoid_bwd(const float* __restrict__ grad_out, const float* __restrict__ out, uint32_t n,
                                                        float* __restrict__ grad_in) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) grad_in[i] = (grad_out[i] * (1.0f - out[i])) * out[i];
}

// out[m] = exp(h[m][col]) for row-strided h: trunc_exp's forward (activation.py:9-11) on the density logit column of the sigma MLP's output
__global__ __launch_bounds__(kBlock) void k_exp_col(const float* __restrict__ h, uint32_t row_stride, uint32_t col, uint32_t M, float* __restrict__ out) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m < M) out[m] = expf(h[(size_t)m * row_stride + col]);
}
// number of non-zero bytes of x[0..n): the population of a sample mask (one byte per sample, what `weights > 1e-4` produces).
// torch's sum over a bool tensor reduces int64 element by element: 0.21 ms for the 3.1 M samples of a batch, twice per training step.
// Input gradient of the space-time density tail, handed back per input with the blend factors of network_dynamic.py:273-287
// (DensityTailFn.backward): from dL/dx [M, 120] of the density MLP
//   g[:, 0:32] (plane_s, as rows of its own), 0.5 g[:, 32:64] (plane_d), 0.25 g[:, 32:64] (plane_1 = plane_2), g[:, 64:96] in the
//   dtype of hash_s, 0.5 g[:, 96:120] (hash_d)
// in ONE pass over the rows (five elementwise / strided-copy launches re-read the row once each).  Item = (row, piece of four columns).
__global__ __launch_bounds__(kBlock) void k_density_tail_grad_split(const float* __restrict__ gx, uint32_t gx_stride, uint32_t M, float* __restrict__ g_half,
                                                                    float* __restrict__ g_quarter, void* __restrict__ g_hash_s, int hash_s_f16,
                                                                    float* __restrict__ g_hash_d, float* __restrict__ g_plane_s, int hash_d_col_major,
                                                                    int hash_s_lm, float half_scale) {
    // A workgroup stages 64 rows (columns 0 .. 119) in LDS with whole-line reads and writes every output from there in the order that
    // output wants: rows as 16-byte pieces, the column-major hash_d gradient ([24][M]: its consumer k_hash_dynamic_bwd_lds reads one
    // column per workgroup) as 64 consecutive floats per column.
    constexpr int kRows = 64, kPitch = 121;  // odd pitch: a column of the tile is conflict-free
    __shared__ float s_t[kRows * kPitch];
    const uint32_t n_tiles = (M + kRows - 1) / kRows;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t m0 = tile * kRows, rows = min((uint32_t)kRows, M - m0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < rows * 30u; i += kBlock) {
            const uint32_t r = i / 30u, q = i - r * 30u;
            const float4 v = *reinterpret_cast<const float4*>(gx + (size_t)(m0 + r) * gx_stride + 4u * q);
            float* d = s_t + r * kPitch + 4u * q;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < rows * 8u; i += kBlock) {  // 32-column outputs, 16 bytes per item
            const uint32_t r = i >> 3, q = i & 7u;
            const size_t at = (size_t)(m0 + r) * 32 + 4u * q;
            if (g_plane_s) {
                const float* a = s_t + r * kPitch + 4u * q;
                *reinterpret_cast<float4*>(g_plane_s + at) = make_float4(a[0], a[1], a[2], a[3]);
            }
            const float* b = s_t + r * kPitch + 32 + 4u * q;
            if (g_half) *reinterpret_cast<float4*>(g_half + at) = make_float4(half_scale * b[0], half_scale * b[1], half_scale * b[2], half_scale * b[3]);
            if (g_quarter) *reinterpret_cast<float4*>(g_quarter + at) = make_float4(0.25f * b[0], 0.25f * b[1], 0.25f * b[2], 0.25f * b[3]);
            if (g_hash_s) {
                const float* c = s_t + r * kPitch + 64 + 4u * q;
                const size_t at = hash_s_lm ? ((size_t)q * M + m0 + r) * 4 : (size_t)(m0 + r) * 32 + 4u * q;  // [8][M][4]: piece q = level q
                if (hash_s_f16) {
                    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                    const h4v h = {(_Float16)c[0], (_Float16)c[1], (_Float16)c[2], (_Float16)c[3]};
                    *reinterpret_cast<h4v*>(reinterpret_cast<_Float16*>(g_hash_s) + at) = h;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(g_hash_s) + at) = make_float4(c[0], c[1], c[2], c[3]);
                }
            }
        }
        if (g_hash_d) {
            if (hash_d_col_major) {
                for (uint32_t i = threadIdx.x; i < 24u * kRows; i += kBlock) {
                    const uint32_t c = i >> 6, r = i & 63u;
                    if (r < rows) g_hash_d[(size_t)c * M + m0 + r] = 0.5f * s_t[r * kPitch + 96 + c];
                }
            } else {
                for (uint32_t i = threadIdx.x; i < rows * 6u; i += kBlock) {
                    const uint32_t r = i / 6u, q = i - r * 6u;
                    const float* d = s_t + r * kPitch + 96 + 4u * q;
                    *reinterpret_cast<float4*>(g_hash_d + (size_t)(m0 + r) * 24 + 4u * q) = make_float4(0.5f * d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[3]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_count_nonzero_u8(const uint8_t* __restrict__ x, unsigned long long n, unsigned long long* __restrict__ out) {
    const unsigned long long words = n / 16ull;  // 16-byte pieces (the pointer is 16-byte aligned: checked by the launcher)
    const uint4* x16 = reinterpret_cast<const uint4*>(x);
    uint32_t count = 0;
    auto nz = [](uint32_t w) {  // bytes of w that are not zero
        w |= w >> 4;
        w |= w >> 2;
        w |= w >> 1;
        return (uint32_t)__builtin_popcount(w & 0x01010101u);
    };
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * kBlock) {
        const uint4 v = x16[i];
        count += nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(n - words * 16ull)) count += x[words * 16ull + threadIdx.x] != 0;
    count = wave_sum(count);
    if (lane_id() == 0 && count) atomicAdd(out, (unsigned long long)count);
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_sigma_geo_bwd(const float* grad_sigma, const float* sigma, const float* grad_geo, uint32_t gg_stride, uint32_t n_geo,
                                uint32_t M, float* grad_h, uint32_t gh_stride, float sigma_lo, float sigma_hi, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(grad_h && gh_stride >= 16 && gh_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_h) & 15u) == 0);
    REQUIRE(n_geo <= 15 && (!grad_geo || gg_stride >= n_geo) && (!grad_sigma || sigma));
    const bool rows16 = grad_geo && gg_stride % 4 == 0 && gg_stride >= 16 && (reinterpret_cast<uintptr_t>(grad_geo) & 15u) == 0 && M < (1u << 30);
    if (rows16)
        hipLaunchKernelGGL(k_sigma_geo_bwd_rows16, dim3(cdiv(4ull * M, kBlock)), dim3(kBlock), 0, stream, grad_sigma, sigma, grad_geo, gg_stride,
                           n_geo, M, grad_h, gh_stride, sigma_lo, sigma_hi);
    else
        hipLaunchKernelGGL(k_sigma_geo_bwd, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, stream, grad_sigma, sigma, grad_geo, gg_stride, n_geo, M,
                           grad_h, gh_stride, sigma_lo, sigma_hi);
    return nvsf_launch_status();
}

NVSF_API int nvsf_cast_cols_f16(const void* src, int src_is_f16, uint32_t M, uint32_t n_cols, uint32_t src_stride, void* dst_f16,
                                uint32_t dst_stride, hipStream_t stream) {
    if (M == 0 || n_cols == 0) return NVSF_OK;
    REQUIRE(src && dst_f16 && src_stride >= n_cols && dst_stride >= n_cols);
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f16);
    const uint32_t pitch = n_cols <= 16 ? 16u : (n_cols <= 32 ? 32u : (n_cols <= 64 ? 64u : 0u));
    const unsigned long long total = (unsigned long long)M * (pitch ? pitch : n_cols);
    REQUIRE(total < (1ull << 32));
#define CAST(SF, P) hipLaunchKernelGGL((k_cast_cols<SF, P>), dim3(cdiv(total, kBlock)), dim3(kBlock), 0, stream, src, M, n_cols, src_stride, dst, dst_stride)
#define CAST_P(SF) do { if (pitch == 16u) CAST(SF, 16); else if (pitch == 32u) CAST(SF, 32); else if (pitch == 64u) CAST(SF, 64); else CAST(SF, 0); } while (0)
    if (src_is_f16) CAST_P(true); else CAST_P(false);
    return nvsf_launch_status();
}

NVSF_API int nvsf_repeat_rows_f16(const void* src_f16, uint32_t N, uint32_t n_cols, uint32_t src_stride, uint32_t T, void* dst_f16,
                                  uint32_t dst_stride, hipStream_t stream) {
    if (N == 0 || T == 0 || n_cols == 0) return NVSF_OK;
    REQUIRE(src_f16 && dst_f16 && src_stride >= n_cols && dst_stride >= n_cols);
    const _Float16* src = reinterpret_cast<const _Float16*>(src_f16);
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f16);
    const bool vec8 = n_cols % 8 == 0 && src_stride % 8 == 0 && dst_stride % 8 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    const uint32_t groups = vec8 ? n_cols / 8 : n_cols;
    const unsigned long long per_row = (unsigned long long)T * groups;
    REQUIRE(per_row < (1ull << 31));
    const uint32_t blocks_per_row = cdiv(per_row, kBlock);
    REQUIRE((unsigned long long)N * blocks_per_row < (1ull << 31));
    if (vec8) hipLaunchKernelGGL(k_repeat_rows<8>, dim3(N * blocks_per_row), dim3(kBlock), 0, stream, src, groups, src_stride, T, blocks_per_row, dst, dst_stride);
    else hipLaunchKernelGGL(k_repeat_rows<1>, dim3(N * blocks_per_row), dim3(kBlock), 0, stream, src, groups, src_stride, T, blocks_per_row, dst, dst_stride);
    return nvsf_launch_status();
}

NVSF_API int nvsf_heads_input_f16(const void* enc_ray_f16, uint32_t N, uint32_t n_enc, uint32_t enc_stride, uint32_t T, const void* geo,
                                  int geo_is_f16, uint32_t n_geo, uint32_t geo_stride, void* dst_f16, uint32_t in_cols, uint32_t dst_stride,
                                  hipStream_t stream) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(enc_ray_f16 && geo && dst_f16);
    REQUIRE(n_enc % 8 == 0 && enc_stride % 8 == 0 && enc_stride >= n_enc && (reinterpret_cast<uintptr_t>(enc_ray_f16) & 15u) == 0);
    REQUIRE(in_cols % 8 == 0 && n_enc + n_geo <= in_cols && dst_stride >= in_cols && dst_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(dst_f16) & 15u) == 0);
    REQUIRE(geo_stride >= n_geo);
    const size_t esz = geo_is_f16 ? 2 : 4;
    // 8 source columns per thread as 16-byte loads: aligned rows, wide enough for the last group of 8
    int geo_vec = (reinterpret_cast<uintptr_t>(geo) & 15u) == 0 && (geo_stride * esz) % 16 == 0 && (n_geo + 7u) / 8u * 8u <= geo_stride;
    // fp32 rows that begin one float past a 16-byte boundary, wide enough for every group's aligned 8-float window
    if (!geo_vec && !geo_is_f16 && (reinterpret_cast<uintptr_t>(geo) & 15u) == 4 && geo_stride % 4 == 0 && (n_geo + 1u + 7u) / 8u * 8u <= geo_stride)
        geo_vec = 2;
    const unsigned long long per_ray = (unsigned long long)T * (in_cols / 8);
    REQUIRE(per_ray < (1ull << 31));
    const uint32_t blocks_per_ray = cdiv(per_ray, kBlock);
    REQUIRE((unsigned long long)N * blocks_per_ray < (1ull << 31));
    const _Float16* enc = reinterpret_cast<const _Float16*>(enc_ray_f16);
    _Float16* dst = reinterpret_cast<_Float16*>(dst_f16);
    if (geo_is_f16)
        hipLaunchKernelGGL(k_heads_input<true>, dim3(N * blocks_per_ray), dim3(kBlock), 0, stream, enc, n_enc, enc_stride, T, blocks_per_ray, geo, n_geo,
                           geo_stride, geo_vec, dst, in_cols, dst_stride);
    else
        hipLaunchKernelGGL(k_heads_input<false>, dim3(N * blocks_per_ray), dim3(kBlock), 0, stream, enc, n_enc, enc_stride, T, blocks_per_ray, geo, n_geo,
                           geo_stride, geo_vec, dst, in_cols, dst_stride);
    return nvsf_launch_status();
}

NVSF_API int nvsf_masked_sigmoid(const float* logits, uint32_t row_stride, uint32_t col_stride, const void* mask_u8, uint32_t M, uint32_t C,
                                 float* out, hipStream_t stream) {
    if (M == 0 || C == 0) return NVSF_OK;
    REQUIRE(logits && out && (unsigned long long)M * C < (1ull << 32));
    hipLaunchKernelGGL(k_masked_sigmoid, dim3(cdiv((unsigned long long)M * C, kBlock)), dim3(kBlock), 0, stream, logits, row_stride, col_stride,
                       reinterpret_cast<const unsigned char*>(mask_u8), M, C, out);
    return nvsf_launch_status();
}

NVSF_API int nvsf_sigmoid_bwd(const float* grad_out, const float* out, uint32_t n, float* grad_in, hipStream_t stream) {
    if (n == 0) return NVSF_OK;
    REQUIRE(grad_out && out && grad_in);
    hipLaunchKernelGGL(k_sigmoid_bwd, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, grad_out, out, n, grad_in);
    return nvsf_launch_status();
}

NVSF_API int nvsf_count_nonzero_u8(const void* x, uint64_t n, int64_t* count, hipStream_t stream) {
    REQUIRE(count);
    if (hipMemsetAsync(count, 0, 8, stream) != hipSuccess) return (int)hipGetLastError();
    if (n == 0) return NVSF_OK;
    REQUIRE(x && (reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    const unsigned long long words = n / 16ull;
    const uint32_t blocks = (uint32_t)(words / kBlock < 1024ull ? words / kBlock + 1ull : 1024ull);
    hipLaunchKernelGGL(k_count_nonzero_u8, dim3(blocks), dim3(kBlock), 0, stream, static_cast<const uint8_t*>(x), (unsigned long long)n,
                       reinterpret_cast<unsigned long long*>(count));
    return nvsf_launch_status();
}

NVSF_API int nvsf_exp_col(const float* h, uint32_t row_stride, uint32_t col, uint32_t M, float* out, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(h && out && col < row_stride);
    hipLaunchKernelGGL(k_exp_col, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, stream, h, row_stride, col, M, out);
    return nvsf_launch_status();
}

NVSF_API int nvsf_density_tail_grad_split(const float* grad_x, uint32_t gx_stride, uint32_t M, float* g_plane_half, float* g_plane_quarter,
                                          void* g_hash_s, int hash_s_is_f16, int hash_s_level_major, float* g_hash_d_half, int hash_d_col_major,
                                          float* g_plane_s, float plane_half_scale, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(grad_x && gx_stride >= 120 && gx_stride % 4 == 0 && (g_plane_half || g_plane_quarter || g_hash_s || g_hash_d_half || g_plane_s));
    const void* ptrs[] = {grad_x, g_plane_half, g_plane_quarter, g_hash_d_half, g_plane_s};
    for (const void* q : ptrs) REQUIRE((reinterpret_cast<uintptr_t>(q) & 15u) == 0);
    REQUIRE((reinterpret_cast<uintptr_t>(g_hash_s) & (hash_s_is_f16 ? 7u : 15u)) == 0);
    const uint32_t want = (M + 63u) / 64u;
    hipLaunchKernelGGL(k_density_tail_grad_split, dim3(want < 4096u ? want : 4096u), dim3(kBlock), 0, stream, grad_x, gx_stride, M,
                       g_plane_half, g_plane_quarter, g_hash_s, hash_s_is_f16, g_hash_d_half, g_plane_s, hash_d_col_major, hash_s_level_major,
                       plane_half_scale);
    return nvsf_launch_status();
}
