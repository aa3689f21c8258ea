// Device-side building blocks of the multiresolution hash-grid encoding (shared by hashgrid.hip and the
// fused field kernels).  Specification: DESIGN.md section 4.1 (restates Instant-NGP sec. 3 / the tcnn
// "HashGrid" encoding the reference instantiates at hash_field.py:47-57,109-119 and flow_field.py:70-80).
#pragma once
#include "common.h"
#include <hip/hip_fp16.h>

constexpr int kMaxLevels = 32;

// Per-level metadata, computed on the host once per encoder and passed by value (kernarg -> SGPRs).
struct GridMeta {
    float scale[kMaxLevels];        // exp2(l*log2(per_level_scale))*base_resolution - 1
    uint32_t res[kMaxLevels];       // ceil(scale)+1
    uint32_t offset[kMaxLevels + 1];  // first feature-vector row of each level; offset[L] = total rows
};

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

template <int F> struct FeatVec;
template <> struct FeatVec<1> { typedef _Float16 type; };
template <> struct FeatVec<2> { typedef h2_t type; };
template <> struct FeatVec<4> { typedef h4_t type; };
template <> struct FeatVec<8> { typedef h8_t type; };

template <int F>
__device__ __forceinline__ void load_feat(const _Float16* __restrict__ table, size_t row, float (&v)[F]) {
    if constexpr (F == 1) {
        v[0] = (float)table[row];
    } else {
        typedef typename FeatVec<F>::type vec_t;
        const vec_t t = *reinterpret_cast<const vec_t*>(table + row * F);
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = (float)t[f];
    }
}

template <int D>
__device__ __forceinline__ uint32_t grid_row(const uint32_t (&cell)[D], uint32_t res, uint32_t hsize) {
    // dense index while res^d still fits the level's table, spatial hash otherwise
    unsigned long long stride = 1;
    uint32_t index = 0;
    bool dense = true;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (stride <= hsize) {
            index += cell[d] * (uint32_t)stride;
            stride *= res;
        } else {
            dense = false;
        }
    }
    if (!dense || hsize < stride) {
        constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
        index = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) index ^= cell[d] * primes[d];
    }
    // table sizes of hashed levels are powers of two: avoid the integer division on the hot path
    return (hsize & (hsize - 1u)) == 0u ? (index & (hsize - 1u)) : (index % hsize);
}

// Encodes one sample at one level: acc[f] = sum_c w_c * table[row_c][f]  (fp32 fmaf chain, corner order c)
template <int D, int F>
__device__ __forceinline__ void encode_level(const float (&x)[D], const _Float16* __restrict__ table, float scale,
                                             uint32_t res, uint32_t row0, uint32_t hsize, float (&acc)[F]) {
    float frac[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pos = fmaf(scale, x[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        cell[d] = (uint32_t)(int32_t)fl;
    }
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.0f;
    // issue all 2^D gathers first, then blend (keeps 2^D loads in flight per lane)
    float v[1 << D][F];
    float w[1 << D];
#pragma unroll
    for (int c = 0; c < (1 << D); ++c) {
        uint32_t cc[D];
        float wc = 1.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (c & (1 << d)) { wc = wc * frac[d]; cc[d] = cell[d] + 1u; }
            else { wc = wc * (1.0f - frac[d]); cc[d] = cell[d]; }
        }
        w[c] = wc;
        load_feat<F>(table, (size_t)row0 + grid_row<D>(cc, res, hsize), v[c]);
    }
#pragma unroll
    for (int c = 0; c < (1 << D); ++c)
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf(w[c], v[c][f], acc[f]);
}
