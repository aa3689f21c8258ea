// Stand-alone multiresolution hash-grid encoding (forward + table gradient) for gfx950.
// Replaces the tcnn.Encoding("HashGrid") modules the reference builds at hash_field.py:47-57 (2-D, per
// time slice), hash_field.py:109-119 (3-D static) and flow_field.py:70-80 (3-D, L16 F8).
//
// Work mapping (forward): a 256-thread workgroup encodes 64 consecutive samples.  Lanes of a wave are
// 64 consecutive samples (consecutive samples along a ray touch neighbouring cells, so the 64 gathers of
// one wave instruction fall into few cache lines); the four waves split the levels (wave w takes levels
// w, w+4, ...).  Features are staged as fp16 in LDS in output order and leave the workgroup as 16-byte
// stores of whole rows, instead of 4..16-byte stores at a stride of L*F*2 bytes.
#include "hashgrid_device.h"

namespace {
constexpr int kBlock = 256;
constexpr int kSamplesPerBlock = 64;

template <int D, int F>
__global__ __launch_bounds__(kBlock) void k_hashgrid_fwd(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                         uint32_t c1, uint32_t c2, const _Float16* __restrict__ table, uint32_t L,
                                                         GridMeta meta, _Float16* __restrict__ out, uint32_t out_stride) {
    // staging tile [64 samples][L*F/2 + 1] dwords (half2 granularity; the odd row stride keeps the
    // per-lane b32 writes and the row-major b32 read-back free of bank conflicts)
    extern __shared__ uint32_t stage[];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t m = blockIdx.x * kSamplesPerBlock + lane;
    const uint32_t row_dw = L * F / 2, row_pitch = row_dw + 1;
    float xs[D];
    {
        const uint32_t mm = m < M ? m : M - 1;  // clamp: out-of-range lanes compute a valid sample and are dropped at the store
        const float* px = x + (size_t)mm * x_stride;
        xs[0] = px[c0];
        xs[1] = px[c1];
        if constexpr (D == 3) xs[2] = px[c2];
    }
    for (uint32_t l = wave; l < L; l += 4) {
        float acc[F];
        encode_level<D, F>(xs, table, meta.scale[l], meta.res[l], meta.offset[l], meta.offset[l + 1] - meta.offset[l], acc);
#pragma unroll
        for (int f = 0; f < F; f += 2) {
            h2_t p;
            p[0] = (_Float16)acc[f];
            p[1] = (_Float16)acc[f + 1];
            stage[lane * row_pitch + (l * F + f) / 2] = __builtin_bit_cast(uint32_t, p);
        }
    }
    __syncthreads();
    const uint32_t n_rows = min((uint32_t)kSamplesPerBlock, M - blockIdx.x * kSamplesPerBlock);
    uint32_t* out_dw = reinterpret_cast<uint32_t*>(out);
    for (uint32_t c = threadIdx.x; c < n_rows * row_dw; c += kBlock) {
        const uint32_t row = c / row_dw, col = c - row * row_dw;
        out_dw[(size_t)(blockIdx.x * kSamplesPerBlock + row) * (out_stride / 2) + col] = stage[row * row_pitch + col];
    }
}

// Table gradient: one thread per (sample, level); fp32 atomics into grad_table.
template <int D, int F, bool GRAD_F16>
__global__ __launch_bounds__(kBlock) void k_hashgrid_bwd(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                         uint32_t c1, uint32_t c2, uint32_t L, GridMeta meta,
                                                         const void* __restrict__ grad_out, uint32_t go_stride,
                                                         float* __restrict__ grad_table) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t l = blockIdx.y;
    if (m >= M) return;
    float xs[D];
    const float* px = x + (size_t)m * x_stride;
    xs[0] = px[c0];
    xs[1] = px[c1];
    if constexpr (D == 3) xs[2] = px[c2];
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if constexpr (GRAD_F16) g[f] = (float)reinterpret_cast<const _Float16*>(grad_out)[(size_t)m * go_stride + l * F + f];
        else g[f] = reinterpret_cast<const float*>(grad_out)[(size_t)m * go_stride + l * F + f];
    }
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; ++f) any |= (g[f] != 0.0f);
    if (!any) return;
    const float scale = meta.scale[l];
    const uint32_t res = meta.res[l], row0 = meta.offset[l], hsize = meta.offset[l + 1] - row0;
    float frac[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pos = fmaf(scale, xs[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        cell[d] = (uint32_t)(int32_t)fl;
    }
#pragma unroll
    for (int c = 0; c < (1 << D); ++c) {
        uint32_t cc[D];
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (c & (1 << d)) { w = w * frac[d]; cc[d] = cell[d] + 1u; }
            else { w = w * (1.0f - frac[d]); cc[d] = cell[d]; }
        }
        float* dst = grad_table + ((size_t)row0 + grid_row<D>(cc, res, hsize)) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(dst + f, w * g[f]);
    }
}

// Table gradient, run-merging formulation.  Float atomics execute at the memory side and their cost follows the
// number of distinct 64-B segments a wave-instruction touches (MI355X_MICROARCH.md, Global float atomics), while
// consecutive rows of x are neighbouring samples of a ray: at every level but the finest few they stay in the same
// cell for several rows.  A thread therefore owns (a chunk of `run` consecutive rows, one level), walks the rows in
// order, keeps the 2^D x F corner sums of the current cell in registers and issues its atomics only when the cell
// changes.  Lanes of a chunk are adjacent (lane = chunk * L + level), so the gradient row is read as one contiguous
// segment and the position is a broadcast.  Same sums as k_hashgrid_bwd up to the order of the fp32 additions.
template <int D, int F, bool GRAD_F16>
__global__ __launch_bounds__(kBlock) void k_hashgrid_bwd_runs(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                              uint32_t c1, uint32_t c2, uint32_t L, GridMeta meta,
                                                              const void* __restrict__ grad_out, uint32_t go_stride,
                                                              float* __restrict__ grad_table, uint32_t run) {
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t chunk = tid / L, l = tid - chunk * L;
    const unsigned long long first = (unsigned long long)chunk * run;
    if (first >= M) return;
    const uint32_t m0 = (uint32_t)first, m1 = (uint32_t)(first + run < M ? first + run : M);
    const float scale = meta.scale[l];
    const uint32_t res = meta.res[l], row0 = meta.offset[l], hsize = meta.offset[l + 1] - row0;
    constexpr int NC = 1 << D;
    float acc[NC][F];
    uint32_t cur[D];
    bool have = false;
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = 0u;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int f = 0; f < F; ++f) acc[c][f] = 0.0f;

    auto flush = [&]() {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint32_t cc[D];
#pragma unroll
            for (int d = 0; d < D; ++d) cc[d] = cur[d] + ((c >> d) & 1u);
            float* dst = grad_table + ((size_t)row0 + grid_row<D>(cc, res, hsize)) * F;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                atomicAdd(dst + f, acc[c][f]);
                acc[c][f] = 0.0f;
            }
        }
    };
    auto load_row = [&](uint32_t m, float (&xs)[D], float (&g)[F]) {
        const float* px = x + (size_t)m * x_stride;
        xs[0] = px[c0];
        xs[1] = px[c1];
        if constexpr (D == 3) xs[2] = px[c2];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if constexpr (GRAD_F16) g[f] = (float)reinterpret_cast<const _Float16*>(grad_out)[(size_t)m * go_stride + l * F + f];
            else g[f] = reinterpret_cast<const float*>(grad_out)[(size_t)m * go_stride + l * F + f];
        }
    };
    float xs_n[D], g_n[F];
    load_row(m0, xs_n, g_n);
    for (uint32_t m = m0; m < m1; ++m) {
        float xs[D], g[F];
#pragma unroll
        for (int d = 0; d < D; ++d) xs[d] = xs_n[d];
#pragma unroll
        for (int f = 0; f < F; ++f) g[f] = g_n[f];
        if (m + 1 < m1) load_row(m + 1, xs_n, g_n);  // next row in flight while this one is processed
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) any |= (g[f] != 0.0f);
        if (!any) continue;
        float frac[D];
        uint32_t cell[D];
        bool same = have;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float pos = fmaf(scale, xs[d], 0.5f);
            const float fl = floorf(pos);
            frac[d] = pos - fl;
            cell[d] = (uint32_t)(int32_t)fl;
            same = same && (cell[d] == cur[d]);
        }
        if (!same) {
            if (have) flush();
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = cell[d];
            have = true;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float w = 1.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) w = w * ((c & (1 << d)) ? frac[d] : (1.0f - frac[d]));
#pragma unroll
            for (int f = 0; f < F; ++f) acc[c][f] += w * g[f];
        }
    }
    if (have) flush();
}

// Table gradient, corner-parallel run merging.  The atomic adds are what bounds this pass once the gradients are dense
// (805 M fp32 adds per 3.1 M-sample batch at L16 F2), and MI355X executes float atomics at the memory side at a cost
// per wave-instruction and 64-B segment (MI355X_MICROARCH.md, Global float atomics): a lane that walks the 2^D x F
// floats of a cell one after the other -- k_hashgrid_bwd_runs -- makes 2^D x F instructions whose 64 lanes sit in 64
// unrelated rows, the slowest shape there is.  Here the 2^D x F floats of one (chunk of rows, level) item are spread over
// G = 2^D x F adjacent lanes (lane = corner x F + feature), 64 / G items (consecutive levels of one chunk) per wave:
// every lane keeps ONE running sum, the cell changes for all lanes of an item at once, and a flush is ONE atomic
// instruction whose lanes cover whole table entries (F contiguous floats; the two x-neighbours of a corner pair are
// adjacent entries on dense levels and for even cells on hashed ones).  Same sums as the other two kernels up to the
// order of the fp32 additions.
//
// FIXED: the sums go to a 64-bit fixed-point table instead (global_atomic_add_x2: 23.7 G segments/s against 21.1 G for
// global_atomic_add_f32, tools/exp_atomics.hip) -- integer addition is associative, so the table gradient no longer depends
// on the order in which the atomics land: two runs on the same inputs agree bit for bit.  The scale is a power of two chosen
// per call from max |grad| (k_absmax_bits) such that M adders of that magnitude cannot overflow: 2^shift with
// shift = 62 - ceil(log2 M) - exponent(max |grad|); a non-finite gradient leaves the table untouched and k_fixed_to_f32
// writes NaN everywhere (what the fp32 atomics would have spread; the GradScaler skips the step either way).
__device__ __forceinline__ int fixed_shift(uint32_t gmax_bits, uint32_t M) {
    int e;
    frexpf(__uint_as_float(gmax_bits), &e);  // max |grad| < 2^e
    const int log2m = 32 - __builtin_clz(M > 1u ? M - 1u : 1u);
    int shift = 62 - log2m - e;
    return shift > 100 ? 100 : shift;  // 2^shift stays a normal float
}
__device__ __forceinline__ bool bits_finite(uint32_t bits) { return bits < 0x7f800000u; }

template <int D, int F, bool GRAD_F16, bool FIXED>
__global__ __launch_bounds__(kBlock) void k_hashgrid_bwd_corners(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                                 uint32_t c1, uint32_t c2, uint32_t L, GridMeta meta,
                                                                 const void* __restrict__ grad_out, uint32_t go_stride,
                                                                 float* __restrict__ grad_table, uint32_t run,
                                                                 const uint32_t* __restrict__ gmax_bits) {
    float fx_scale = 1.0f;
    if constexpr (FIXED) {
        const uint32_t gb = gmax_bits[0];
        if (gb == 0u || !bits_finite(gb)) return;  // nothing to add / poisoned step (k_fixed_to_f32 reports it)
        fx_scale = ldexpf(1.0f, fixed_shift(gb, M));
    }
    auto flush = [&](float* dst, float acc) {
        if constexpr (FIXED) {
            // dst indexes the fp32 layout; the fixed-point table has the same element order at 8 bytes per element
            unsigned long long* d64 = reinterpret_cast<unsigned long long*>(grad_table) + (dst - grad_table);
            atomicAdd(d64, (unsigned long long)__float2ll_rn(acc * fx_scale));
        } else {
            atomicAdd(dst, acc);
        }
    };
    constexpr int G = (1 << D) * F;  // lanes per item
    constexpr int IPW = kWave / G;   // items per wave
    static_assert(G <= kWave && kWave % G == 0, "2^D x F must divide the wave");
    const int lane = lane_id();
    const int sub = lane / G, r = lane - sub * G, c = r / F, f = r - c * F;
    const unsigned long long item = ((unsigned long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * IPW + (unsigned)sub;
    const uint32_t chunk = (uint32_t)(item / L), l = (uint32_t)(item - (unsigned long long)chunk * L);
    const unsigned long long first = (unsigned long long)chunk * run;
    if (first >= M) return;
    const uint32_t m0 = (uint32_t)first, m1 = (uint32_t)(first + run < M ? first + run : M);
    const float scale = meta.scale[l];
    const uint32_t res = meta.res[l], row0 = meta.offset[l], hsize = meta.offset[l + 1] - row0;
    float acc = 0.0f;
    uint32_t cur[D];
    bool have = false;
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = 0u;
    float* dst = grad_table;
    auto load_row = [&](uint32_t m, float (&xs)[D], float& g) {
        const float* px = x + (size_t)m * x_stride;
        xs[0] = px[c0];
        xs[1] = px[c1];
        if constexpr (D == 3) xs[2] = px[c2];
        if constexpr (GRAD_F16) g = (float)reinterpret_cast<const _Float16*>(grad_out)[(size_t)m * go_stride + l * F + f];
        else g = reinterpret_cast<const float*>(grad_out)[(size_t)m * go_stride + l * F + f];
    };
    float xs_n[D], g_n;
    load_row(m0, xs_n, g_n);
    for (uint32_t m = m0; m < m1; ++m) {
        float xs[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xs[d] = xs_n[d];
        const float g = g_n;
        if (m + 1 < m1) load_row(m + 1, xs_n, g_n);  // next row in flight while this one is processed
        if (g == 0.0f) continue;                     // this lane's feature has nothing to add (its sum keeps its cell)
        float w = 1.0f;
        uint32_t cell[D];
        bool same = have;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float pos = fmaf(scale, xs[d], 0.5f);
            const float fl = floorf(pos);
            const float frac = pos - fl;
            cell[d] = (uint32_t)(int32_t)fl;
            same = same && (cell[d] == cur[d]);
            w = w * ((c & (1 << d)) ? frac : (1.0f - frac));
        }
        if (!same) {
            if (acc != 0.0f) flush(dst, acc);
            acc = 0.0f;
            have = true;
            uint32_t cc[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                cur[d] = cell[d];
                cc[d] = cell[d] + ((c >> d) & 1u);
            }
            dst = grad_table + ((size_t)row0 + grid_row<D>(cc, res, hsize)) * F + f;
        }
        acc += w * g;
    }
    if (acc != 0.0f) flush(dst, acc);
}

// max |grad| as float bits (non-negative floats order like unsigned integers; inf / NaN sort above every finite value)
template <bool GRAD_F16>
__global__ __launch_bounds__(kBlock) void k_absmax_bits(const void* __restrict__ grad_out, uint32_t M, uint32_t n_cols, uint32_t go_stride,
                                                        int flat16, uint32_t* __restrict__ gmax_bits) {
    const unsigned long long total = (unsigned long long)M * n_cols;
    const unsigned long long tid = (unsigned long long)blockIdx.x * kBlock + threadIdx.x, nthreads = (unsigned long long)gridDim.x * kBlock;
    uint32_t best = 0u;
    if (flat16) {  // dense rows, 16-byte aligned: the matrix as a flat array of 16-byte words
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const unsigned long long per = GRAD_F16 ? 8 : 4, words = total / per;
        const u4* p = reinterpret_cast<const u4*>(grad_out);
        for (unsigned long long i = tid; i < words; i += nthreads) {
            const u4 w = p[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (GRAD_F16) {  // two halves per dword: compare their magnitudes as fp32 bit patterns
                    const uint32_t lo = __float_as_uint((float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] & 0x7fffu)));
                    const uint32_t hi = __float_as_uint((float)__builtin_bit_cast(_Float16, (unsigned short)((w[k] >> 16) & 0x7fffu)));
                    best = lo > best ? lo : best;
                    best = hi > best ? hi : best;
                } else {
                    const uint32_t b = w[k] & 0x7fffffffu;
                    best = b > best ? b : best;
                }
            }
        }
        for (unsigned long long i = words * per + tid; i < total; i += nthreads) {  // tail
            float v;
            if constexpr (GRAD_F16) v = (float)reinterpret_cast<const _Float16*>(grad_out)[i];
            else v = reinterpret_cast<const float*>(grad_out)[i];
            const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
            best = b > best ? b : best;
        }
    } else {
        for (unsigned long long i = tid; i < total; i += nthreads) {
            const unsigned long long m = i / n_cols;
            const size_t at = (size_t)m * go_stride + (size_t)(i - m * n_cols);
            float v;
            if constexpr (GRAD_F16) v = (float)reinterpret_cast<const _Float16*>(grad_out)[at];
            else v = reinterpret_cast<const float*>(grad_out)[at];
            const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
            best = b > best ? b : best;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)best, o);
        best = other > best ? other : best;
    }
    if (lane_id() == 0 && best) atomicMax(gmax_bits, best);
}

__global__ __launch_bounds__(kBlock) void k_fixed_to_f32(const long long* __restrict__ acc, unsigned long long n, uint32_t M,
                                                         const uint32_t* __restrict__ gmax_bits, float* __restrict__ grad_table) {
    const uint32_t gb = gmax_bits[0];
    if (gb == 0u) return;
    const bool poisoned = !bits_finite(gb);
    const int shift = poisoned ? 0 : fixed_shift(gb, M);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * kBlock) {
        if (poisoned) { grad_table[i] = __uint_as_float(0x7fc00000u); continue; }
        const long long v = acc[i];
        if (v != 0) grad_table[i] += (float)ldexp((double)v, -shift);
    }
}

int fill_meta(GridMeta& meta, uint32_t L, const float* scales, const uint32_t* res, const uint32_t* offsets) {
    if (L == 0 || L > (uint32_t)kMaxLevels || !scales || !res || !offsets) return NVSF_ERR_INVALID_ARG;
    for (uint32_t l = 0; l < L; ++l) {
        meta.scale[l] = scales[l];
        meta.res[l] = res[l];
        meta.offset[l] = offsets[l];
        if (offsets[l + 1] <= offsets[l] || res[l] == 0) return NVSF_ERR_INVALID_ARG;
    }
    meta.offset[L] = offsets[L];
    return NVSF_OK;
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

#define DISPATCH_DF(D, F, CALL)                                         \
    do {                                                                \
        if (D == 2 && F == 2) { CALL(2, 2); }                           \
        else if (D == 2 && F == 4) { CALL(2, 4); }                      \
        else if (D == 2 && F == 8) { CALL(2, 8); }                      \
        else if (D == 3 && F == 2) { CALL(3, 2); }                      \
        else if (D == 3 && F == 4) { CALL(3, 4); }                      \
        else if (D == 3 && F == 8) { CALL(3, 8); }                      \
        else return NVSF_ERR_UNSUPPORTED;                               \
    } while (0)

NVSF_API int nvsf_hashgrid_fwd(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, const void* table_f16,
                               uint32_t L, uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                               void* out_f16, uint32_t out_stride, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && cols && table_f16 && out_f16 && (D == 2 || D == 3));
    REQUIRE(out_stride >= L * F && out_stride % 2 == 0);
    REQUIRE((reinterpret_cast<uintptr_t>(out_f16) & 3u) == 0 && (reinterpret_cast<uintptr_t>(table_f16) & 15u) == 0);
    for (uint32_t d = 0; d < D; ++d) REQUIRE(cols[d] < x_stride);
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    const uint32_t c0 = cols[0], c1 = cols[1], c2 = D == 3 ? cols[2] : 0;
    const size_t lds = (size_t)kSamplesPerBlock * (L * F / 2 + 1) * sizeof(uint32_t);
#define CALL(DD, FF)                                                                                                        \
    hipLaunchKernelGGL((k_hashgrid_fwd<DD, FF>), dim3(cdiv(M, kSamplesPerBlock)), dim3(kBlock), lds, stream, x, M, x_stride, c0, \
                       c1, c2, reinterpret_cast<const _Float16*>(table_f16), L, meta, reinterpret_cast<_Float16*>(out_f16),  \
                       out_stride)
    DISPATCH_DF(D, F, CALL);
#undef CALL
    return nvsf_launch_status();
}

NVSF_API int nvsf_hashgrid_bwd(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L, uint32_t F,
                               const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, const void* grad_out,
                               int grad_is_f16, uint32_t go_stride, float* grad_table_f32, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && cols && grad_out && grad_table_f32 && (D == 2 || D == 3) && go_stride >= L * F);
    for (uint32_t d = 0; d < D; ++d) REQUIRE(cols[d] < x_stride);
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    const uint32_t c0 = cols[0], c1 = cols[1], c2 = D == 3 ? cols[2] : 0;
    const char* variant = getenv("NVSF_HASHGRID_BWD");  // "atomic": one thread per (row, level); "runs": one thread per (chunk, level) (A/B timing, tests)
    if (!(variant && (variant[0] == 'a' || variant[0] == 'r')) && L % (kWave / ((1u << D) * F)) == 0) {
        const char* run_env = getenv("NVSF_HASHGRID_BWD_RUN");
        const uint32_t run = run_env ? (uint32_t)atoi(run_env) : (M >= (1u << 20) ? 128u : 32u);
        const uint32_t ipw = kWave / ((1u << D) * F);
        const unsigned long long waves = ((unsigned long long)cdiv(M, run) * L + ipw - 1) / ipw;
        const dim3 cgrid((uint32_t)((waves + kBlock / kWave - 1) / (kBlock / kWave)));
#define CALLC(DD, FF)                                                                                                                \
    do {                                                                                                                             \
        if (grad_is_f16)                                                                                                             \
            hipLaunchKernelGGL((k_hashgrid_bwd_corners<DD, FF, true, false>), cgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, grad_table_f32, run, (const uint32_t*)nullptr);                                                            \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_hashgrid_bwd_corners<DD, FF, false, false>), cgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, grad_table_f32, run, (const uint32_t*)nullptr);                                                            \
    } while (0)
        DISPATCH_DF(D, F, CALLC);
#undef CALLC
        return nvsf_launch_status();
    }
    if (!(variant && variant[0] == 'a')) {
        const char* run_env = getenv("NVSF_HASHGRID_BWD_RUN");
        const uint32_t run = run_env ? (uint32_t)atoi(run_env) : (M >= (1u << 20) ? 128u : 32u);
        const unsigned long long threads = (unsigned long long)cdiv(M, run) * L;
        const dim3 rgrid((uint32_t)((threads + kBlock - 1) / kBlock));
#define CALLR(DD, FF)                                                                                                                \
    do {                                                                                                                             \
        if (grad_is_f16)                                                                                                             \
            hipLaunchKernelGGL((k_hashgrid_bwd_runs<DD, FF, true>), rgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, grad_table_f32, run);                                                            \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_hashgrid_bwd_runs<DD, FF, false>), rgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, grad_table_f32, run);                                                            \
    } while (0)
        DISPATCH_DF(D, F, CALLR);
#undef CALLR
        return nvsf_launch_status();
    }
    const dim3 grid(cdiv(M, kBlock), L);
#define CALL(DD, FF)                                                                                                          \
    do {                                                                                                                         \
        if (grad_is_f16)                                                                                                         \
            hipLaunchKernelGGL((k_hashgrid_bwd<DD, FF, true>), grid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, grad_table_f32);                                                             \
        else                                                                                                                     \
            hipLaunchKernelGGL((k_hashgrid_bwd<DD, FF, false>), grid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, grad_table_f32);                                                             \
    } while (0)
    DISPATCH_DF(D, F, CALL);
#undef CALL
    return nvsf_launch_status();
}

NVSF_API int nvsf_hashgrid_bwd_fixed(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L, uint32_t F,
                                     const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, const void* grad_out,
                                     int grad_is_f16, uint32_t go_stride, void* acc_i64, uint32_t* gmax_bits, float* grad_table_f32,
                                     hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && cols && grad_out && grad_table_f32 && acc_i64 && gmax_bits && (D == 2 || D == 3) && go_stride >= L * F);
    REQUIRE((reinterpret_cast<uintptr_t>(acc_i64) & 7u) == 0);
    for (uint32_t d = 0; d < D; ++d) REQUIRE(cols[d] < x_stride);
    if (L % (kWave / ((1u << D) * F)) != 0) return NVSF_ERR_UNSUPPORTED;
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    const uint32_t c0 = cols[0], c1 = cols[1], c2 = D == 3 ? cols[2] : 0;
    const unsigned long long n_params = (unsigned long long)h_offsets[L] * F;
    if (hipMemsetAsync(acc_i64, 0, n_params * sizeof(long long), stream) != hipSuccess) return nvsf_launch_status();
    if (hipMemsetAsync(gmax_bits, 0, sizeof(uint32_t), stream) != hipSuccess) return nvsf_launch_status();
    const unsigned long long cells = (unsigned long long)M * L * F;
    const uint32_t mblocks = (uint32_t)(cells / kBlock / 8 + 1 < 4096ull ? cells / kBlock / 8 + 1 : 4096ull);
    const int flat16 = go_stride == L * F && (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0;
    if (grad_is_f16) hipLaunchKernelGGL(k_absmax_bits<true>, dim3(mblocks), dim3(kBlock), 0, stream, grad_out, M, L * F, go_stride, flat16, gmax_bits);
    else hipLaunchKernelGGL(k_absmax_bits<false>, dim3(mblocks), dim3(kBlock), 0, stream, grad_out, M, L * F, go_stride, flat16, gmax_bits);
    const char* run_env = getenv("NVSF_HASHGRID_BWD_RUN");
    const uint32_t run = run_env ? (uint32_t)atoi(run_env) : (M >= (1u << 20) ? 128u : 32u);
    const uint32_t ipw = kWave / ((1u << D) * F);
    const unsigned long long waves = ((unsigned long long)cdiv(M, run) * L + ipw - 1) / ipw;
    const dim3 cgrid((uint32_t)((waves + kBlock / kWave - 1) / (kBlock / kWave)));
    float* acc_as_f32 = reinterpret_cast<float*>(acc_i64);  // the kernel indexes elements; FIXED addresses them at 8 bytes each
#define CALLX(DD, FF)                                                                                                                   \
    do {                                                                                                                                \
        if (grad_is_f16)                                                                                                                \
            hipLaunchKernelGGL((k_hashgrid_bwd_corners<DD, FF, true, true>), cgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, acc_as_f32, run, (const uint32_t*)gmax_bits);                                       \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_hashgrid_bwd_corners<DD, FF, false, true>), cgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, acc_as_f32, run, (const uint32_t*)gmax_bits);                                       \
    } while (0)
    DISPATCH_DF(D, F, CALLX);
#undef CALLX
    const uint32_t cblocks = (uint32_t)(n_params / kBlock / 4 + 1 < 8192ull ? n_params / kBlock / 4 + 1 : 8192ull);
    hipLaunchKernelGGL(k_fixed_to_f32, dim3(cblocks), dim3(kBlock), 0, stream, reinterpret_cast<const long long*>(acc_i64), n_params, M,
                       (const uint32_t*)gmax_bits, grad_table_f32);
    return nvsf_launch_status();
}
