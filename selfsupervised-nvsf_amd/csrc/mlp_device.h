// MFMA building blocks for the small bias-free MLPs on the NVSF hot path (width 64, fp16 operands, fp32
// accumulate) on gfx950.  Specification: DESIGN.md section 4.3 (restates tcnn "FullyFusedMLP" as used at
// network_dynamic.py:125-135,138-161,180-189).
//
// Orientation: every layer is computed TRANSPOSED,  H^T[o][s] = sum_k W[o][k] X^T[k][s],  with the weight
// matrix as the MFMA A operand (M = 16 output units per tile) and 16 samples on the N dimension.
// With v_mfma_f32_16x16x32_f16 the result tile has the sample on the lane (col = lane & 15) and the output
// unit in the registers (row = 4*(lane>>4) + r), which is exactly the shape the NEXT layer needs for its B
// operand (B[k][n]: k in registers, n on the lane) -- up to a permutation of k inside each 32-wide k-step.
// That permutation is folded into the order in which the next layer's weight fragment is fetched, so the
// hidden activations never leave registers: no LDS round trip, no cross-lane traffic between layers.
//
//   k-step s, lane group g = lane>>4, element j = 0..7  <->  hidden unit  kappa = 16*(2s + (j>>2)) + 4g + (j&3)
#pragma once
#include "common.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef float float8_t __attribute__((ext_vector_type(8)));

constexpr int kHidden = 64;            // width of every hidden layer on this path
constexpr int kHidTiles = kHidden / 16;  // 16-row output tiles per hidden layer
constexpr int kHidSteps = kHidden / 32;  // 32-deep k-steps when a hidden layer is the input

__device__ __forceinline__ float4_t mfma16(half8_t a, half8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// A-fragment of W[16t .. 16t+15][natural k order], k-step s, for a first layer fed from memory/encoders.
// Columns >= in_cols (the tcnn-padded width, a multiple of 16) read as zero.
__device__ __forceinline__ half8_t load_w_natural(const _Float16* __restrict__ W, int in_cols, int t, int s, int lane) {
    const int o = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k0 + 8 <= in_cols) v = *reinterpret_cast<const half8_t*>(W + (size_t)o * in_cols + k0);
    return v;
}
// A-fragment of W[16t .. 16t+15][kappa order] for a layer fed by the previous layer's accumulators.
// `rot` rotates the 16 rows of the tile: fragment row r holds matrix row 16t + (r + rot) % 16.
__device__ __forceinline__ half8_t load_w_chained(const _Float16* __restrict__ W, int t, int s, int lane, int rot = 0) {
    const int o = 16 * t + (((lane & 15) + rot) & 15), g = lane >> 4;
    const half4_t lo = *reinterpret_cast<const half4_t*>(W + (size_t)o * kHidden + 32 * s + 4 * g);
    const half4_t hi = *reinterpret_cast<const half4_t*>(W + (size_t)o * kHidden + 32 * s + 16 + 4 * g);
    half8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
}

// ---- network input -> B fragments -------------------------------------------------------------------------------
// Lane (g, c) of k-step s holds columns 32s + 8g .. +7 of its sample's row, rounded to fp16; columns n_in .. in_cols-1
// read as 1.0 (tcnn pads the network input with ones), anything beyond as 0.  in_cols is n_in rounded up to 16, so
// padding can only occur in the LAST k-step.
//
// Fast form (rows 16-byte aligned and at least round_up(n_in, 8) columns wide -- every buffer of this package): one 16-byte
// load per k-step and lane, no condition anywhere; the last k-step loads from a per-lane column fixed before the loop
// (its own slice, or column 0 where the slice is padding only) and is patched with two per-lane bit masks,
// (v & keep) | pad.  `issue` only loads, `finish` patches: a kernel can request the next tile before it computes the
// current one without a wait in between.
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

struct XTail {
    uint4_t keep, pad;
    int koff;
    __device__ __forceinline__ void init(int k0, int n_in, int in_cols) {
        const int n_keep = n_in - k0 < 0 ? 0 : (n_in - k0 > 8 ? 8 : n_in - k0);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t kp = 0, pd = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * d + e, k = k0 + j;
                if (j < n_keep) kp |= 0xFFFFu << (16 * e);
                else if (k < in_cols) pd |= 0x3C00u << (16 * e);  // fp16 1.0
            }
            keep[d] = kp;
            pad[d] = pd;
        }
        koff = n_keep > 0 ? k0 : 0;
    }
    __device__ __forceinline__ half8_t apply(half8_t v) const {
        return __builtin_bit_cast(half8_t, (__builtin_bit_cast(uint4_t, v) & keep) | pad);
    }
};

inline __host__ __device__ bool x_rows_fast(uint32_t n_in, uint32_t x_stride, int vec_ok) {
    return vec_ok != 0 && (n_in + 7u) / 8u * 8u <= x_stride;
}

template <bool X_F16>
__device__ __forceinline__ half8_t load_x_vec(const void* __restrict__ x, size_t row, uint32_t x_stride, int k0) {
    if constexpr (X_F16) {
        return *reinterpret_cast<const half8_t*>(reinterpret_cast<const _Float16*>(x) + row * x_stride + k0);
    } else {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + row * x_stride + k0);
        const float4 a = p[0], b = p[1];
        const float8_t v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        return __builtin_convertvector(v, half8_t);
    }
}

// General form (any stride / alignment): element loads with the column clamped into the row, issued back to back, padding
// selected afterwards.
template <bool X_F16>
__device__ __forceinline__ half8_t load_x_frag(const void* __restrict__ x, size_t row, uint32_t x_stride, int k0, int n_in, int in_cols,
                                               bool vec_ok) {
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k0 < n_in) {
        if (vec_ok && k0 + 8 <= n_in) {
            v = load_x_vec<X_F16>(x, row, x_stride, k0);
        } else {
            const int last = n_in - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j < last ? k0 + j : last;
                if constexpr (X_F16) v[j] = reinterpret_cast<const _Float16*>(x)[row * x_stride + k];
                else v[j] = (_Float16) reinterpret_cast<const float*>(x)[row * x_stride + k];
            }
        }
    }
    if (k0 + 8 > n_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j >= n_in) v[j] = k0 + j < in_cols ? (_Float16)1.0f : (_Float16)0.0f;
    }
    return v;
}

// Rows with a shared prefix: the first `split` columns of the network input are the same for `rows_per_a` consecutive rows (the
// encoded direction of a ray for the T samples of that ray) and live once per group in `a`; `x` then holds the remaining columns
// only.  The heads read 144 of their 192 input bytes per sample from a row that stays in L1 / L2 instead of from a per-sample copy.
struct XPrefix {
    const _Float16* a;  // nullptr: plain rows
    uint32_t a_stride, rows_per_a, split;
    __device__ __forceinline__ const _Float16* row_of(uint32_t first_row_of_tile) const {  // wave-uniform: a 16-row tile lies in one group
        return a + (size_t)(first_row_of_tile / rows_per_a) * a_stride;
    }
};

template <int IN_STEPS, bool X_F16, bool FAST>
__device__ __forceinline__ void issue_x_row(half8_t (&xf)[IN_STEPS], const void* __restrict__ x, size_t row, uint32_t x_stride, int g, int n_in,
                                            int in_cols, bool vec_ok, const XTail& tail, const _Float16* __restrict__ prefix_row = nullptr,
                                            uint32_t split = 0) {
    if constexpr (FAST) {
        if constexpr (X_F16) {
            if (prefix_row) {  // wave-uniform
                const _Float16* xrow = reinterpret_cast<const _Float16*>(x) + row * x_stride;
                auto at = [&](int col) { return col < (int)split ? prefix_row + col : xrow + (col - (int)split); };
#pragma unroll
                for (int s = 0; s + 1 < IN_STEPS; ++s) xf[s] = *reinterpret_cast<const half8_t*>(at(32 * s + 8 * g));
                xf[IN_STEPS - 1] = *reinterpret_cast<const half8_t*>(at(tail.koff));
                return;
            }
        }
#pragma unroll
        for (int s = 0; s + 1 < IN_STEPS; ++s) xf[s] = load_x_vec<X_F16>(x, row, x_stride, 32 * s + 8 * g);
        xf[IN_STEPS - 1] = load_x_vec<X_F16>(x, row, x_stride, tail.koff);
    } else {
#pragma unroll
        for (int s = 0; s < IN_STEPS; ++s) xf[s] = load_x_frag<X_F16>(x, row, x_stride, 32 * s + 8 * g, n_in, in_cols, vec_ok);
    }
}

// ReLU + round to fp16 of two accumulator tiles (2s, 2s+1) -> B fragment of k-step s of the next layer.
// Order: round to fp16 first (v_cvt_pk_f16_f32, two values per instruction), then clamp the packed halves with
// an INTEGER max against 0 (v_pk_max_i16): an fp16 with the sign bit set is a negative int16.  Rounding is
// sign-preserving and monotonic, so this equals fp16(max(x, 0)) for every finite x, and it avoids both the
// per-value fp32 v_max and the canonicalising v_max hipcc puts in front of fmaxf on MFMA results.

__device__ __forceinline__ half8_t relu_pack(float4_t a, float4_t b) {
    // a VECTOR fp32 -> fp16 conversion selects gfx950's v_cvt_pk_f16_f32 (two values per instruction, round-to-nearest-even)
    const float8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const half8_t h = __builtin_convertvector(v, half8_t);
    const short8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(half8_t, __builtin_elementwise_max(__builtin_bit_cast(short8_t, h), zero));
}

// Hidden layer (64 -> 64) on register-resident activations.
struct HiddenLayerW {
    half8_t w[kHidTiles][kHidSteps];
    __device__ __forceinline__ void load(const _Float16* __restrict__ W, int lane) {
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) w[t][s] = load_w_chained(W, t, s, lane);
    }
    __device__ __forceinline__ void apply(const half8_t (&h)[kHidSteps], float4_t (&acc)[kHidTiles]) const {
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) c = mfma16(w[t][s], h[s], c);
            acc[t] = c;
        }
    }
};

// Output layer (64 -> 16 padded outputs)
struct OutLayerW {
    half8_t w[kHidSteps];
    __device__ __forceinline__ void load(const _Float16* __restrict__ W, int lane, int rot = 0) {
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s) w[s] = load_w_chained(W, 0, s, lane, rot);
    }
    __device__ __forceinline__ float4_t apply(const half8_t (&h)[kHidSteps]) const {
        float4_t c = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s) c = mfma16(w[s], h[s], c);
        return c;
    }
};

__device__ __forceinline__ void pack_hidden(const float4_t (&acc)[kHidTiles], half8_t (&h)[kHidSteps]) {
#pragma unroll
    for (int s = 0; s < kHidSteps; ++s) h[s] = relu_pack(acc[2 * s], acc[2 * s + 1]);
}
