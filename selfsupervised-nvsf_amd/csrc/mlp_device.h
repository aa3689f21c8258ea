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

// ReLU + round to fp16 of two accumulator tiles (2s, 2s+1) -> B fragment of k-step s of the next layer.
// Order: round to fp16 first (v_cvt_pk_f16_f32, two values per instruction), then clamp the packed halves with
// an INTEGER max against 0 (v_pk_max_i16): an fp16 with the sign bit set is a negative int16.  Rounding is
// sign-preserving and monotonic, so this equals fp16(max(x, 0)) for every finite x, and it avoids both the
// per-value fp32 v_max and the canonicalising v_max hipcc puts in front of fmaxf on MFMA results.
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef float float8_t __attribute__((ext_vector_type(8)));

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
