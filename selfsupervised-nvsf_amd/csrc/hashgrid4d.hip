// Fused space-time hash encoder (the dynamic half of HashGrid4D) and the flow-field grid reduction for gfx950.
// Reference: /root/reference/nvsf/nerf/models/hash_field.py:29-88 (HashGridT), :148-159 (forward_dynamic) and
// flow_field.py:105-128 -- there: per coordinate pair two tcnn HashGrid launches (the time slices around t), a linear
// blend, a view/chunk and seven elementwise kernels for the cubic Lagrange reduction, three times per call, three
// calls per density evaluation (t, t+1, t-1): ~40 launches and ~30 [M,32] temporaries per call.
// Here: ONE launch per (positions, time): every (sample, pair, level) item gathers its 2 x 4 corners, blends the two
// slices and reduces its 4 features with the Lagrange weights; results are staged in LDS and leave as whole rows.
//
// The reference's arithmetic depends on the operand types PyTorch sees (hash_field.py docstring in this repo):
//   mode 0 ("tensor t"): blend and reduction in fp32, output fp32;
//   mode 1 ("0-dim t", the flow-warped neighbour frames): every product and every sum is rounded to fp16.
// Both are reproduced operation by operation (tests/test_dynamic_gpu.py against fixtures from the reference code).
#include <stdlib.h>
#include "hashgrid_device.h"

namespace {
constexpr int kBlock = 256;
constexpr int kSamplesPerBlock = 64;
constexpr int kPlaneLevels = 8;   // HashGridT: n_levels = 8, F = 4, num_basis = 4 (hash_field.py:35-38)
constexpr int kF = 4;

struct PlaneSet {
    const _Float16* table_lo[3];  // slice floor(idx) of the pairs (x,y), (x,z), (y,z)
    const _Float16* table_hi[3];  // slice ceil(idx)
    GridMeta meta[3];
    float blend_lo, blend_hi;     // (k2 - idx), (idx - k1)
    float lag[4];                 // Lagrange weights at t
    int same_slice;               // idx integral: no blend, the slice features are used as they are (fp16)
};

__device__ __forceinline__ float r16(float v) { return (float)(_Float16)v; }  // round-trip through fp16

template <int MODE, typename PS>
__device__ __forceinline__ float blend_reduce(const float (&f_lo)[kF], const float (&f_hi)[kF], const PS& ps) {
    float feat[kF];
#pragma unroll
    for (int i = 0; i < kF; ++i) {
        if (ps.same_slice) feat[i] = f_lo[i];
        else if (MODE == 0) feat[i] = ps.blend_lo * f_lo[i] + ps.blend_hi * f_hi[i];
        else feat[i] = r16(r16(ps.blend_lo * f_lo[i]) + r16(ps.blend_hi * f_hi[i]));
    }
    float acc;
    if (MODE == 0) {
        acc = ps.lag[0] * feat[0];
#pragma unroll
        for (int i = 1; i < kF; ++i) acc = acc + ps.lag[i] * feat[i];
    } else {
        acc = r16(ps.lag[0] * feat[0]);
#pragma unroll
        for (int i = 1; i < kF; ++i) acc = r16(acc + r16(ps.lag[i] * feat[i]));
    }
    return acc;
}

// x' = x[:, 0:3] (+ offset[:, off_col : off_col+3]); out [M, 24]: fp32 (MODE 0) or fp16 (MODE 1)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_hash_dynamic(const float* __restrict__ x, uint32_t x_stride, const float* __restrict__ offset,
                                                         uint32_t off_stride, uint32_t off_col, uint32_t M, PlaneSet ps,
                                                         void* __restrict__ out) {
    __shared__ float stage[kSamplesPerBlock][3 * kPlaneLevels + 1];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform by construction: level tables, pointers and loop control in SGPRs
    const uint32_t m = blockIdx.x * kSamplesPerBlock + lane;
    const uint32_t mm = m < M ? m : M - 1;
    float p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        p[d] = x[(size_t)mm * x_stride + d];
        if (offset) p[d] = p[d] + offset[(size_t)mm * off_stride + off_col + d];
    }
    for (int item = wave; item < 3 * kPlaneLevels; item += 4) {
        const int pl = item / kPlaneLevels, l = item - pl * kPlaneLevels;
        const float xy[2] = {p[pl == 2 ? 1 : 0], p[pl == 0 ? 1 : 2]};
        const GridMeta& g = ps.meta[pl];
        const uint32_t rows = g.offset[l + 1] - g.offset[l];
        float f_lo[kF], f_hi[kF];
        encode_level<2, kF>(xy, ps.table_lo[pl], g.scale[l], g.res[l], g.offset[l], rows, f_lo);
#pragma unroll
        for (int i = 0; i < kF; ++i) f_lo[i] = r16(f_lo[i]);  // the slice encoders return fp16
        if (!ps.same_slice) {
            encode_level<2, kF>(xy, ps.table_hi[pl], g.scale[l], g.res[l], g.offset[l], rows, f_hi);
#pragma unroll
            for (int i = 0; i < kF; ++i) f_hi[i] = r16(f_hi[i]);
        } else {
#pragma unroll
            for (int i = 0; i < kF; ++i) f_hi[i] = 0.0f;
        }
        stage[lane][item] = blend_reduce<MODE>(f_lo, f_hi, ps);
    }
    __syncthreads();
    const uint32_t n_rows = min((uint32_t)kSamplesPerBlock, M - blockIdx.x * kSamplesPerBlock);
    constexpr int kOut = 3 * kPlaneLevels;
    for (uint32_t c = threadIdx.x; c < n_rows * kOut; c += kBlock) {
        const uint32_t row = c / kOut, col = c - row * kOut;
        const size_t o = (size_t)(blockIdx.x * kSamplesPerBlock + row) * kOut + col;
        if (MODE == 0) reinterpret_cast<float*>(out)[o] = stage[row][col];
        else reinterpret_cast<_Float16*>(out)[o] = (_Float16)stage[row][col];
    }
}

// The three space-time evaluations of one density query (network_dynamic.py:220-271) in one launch: features at (x, t) in
// regime 0 and at the flow-warped positions (x + f1, t1), (x + f2, t2) of the neighbour frames in regime 1.  The warped
// positions are fractions of a cell away from x at the coarse levels (and everywhere the scene is static), and the
// neighbour frames usually fall between the same two time slices as t: whenever a neighbour's cell at a level equals the
// base cell and its slices are the base slices, the 2 x 4 table entries the base evaluation has just gathered (kept as
// raw 8-byte values, 16 registers) are re-used and only the weights differ; otherwise the neighbour gathers its own.
// Same gathers / same arithmetic per evaluation as three k_hash_dynamic launches: bit-identical outputs.
struct TimeSet {  // what differs between the evaluations: the two slices and the time weights
    const _Float16* table_lo[3];
    const _Float16* table_hi[3];
    float blend_lo, blend_hi;
    float lag[4];
    int same_slice;
};
struct PlaneSet3 {
    TimeSet ev[3];
    GridMeta meta[3];  // per pair (shared by the evaluations)
    int enabled[3];    // ev[0] always; ev[1], ev[2]: neighbour present
    int share[3];      // ev[e] reads the same slice tables as ev[0] (same k1, k2): its cells may re-use ev[0]'s gathers
};

typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gather_quad(const float (&xy)[2], const _Float16* __restrict__ table, float scale, uint32_t res, uint32_t row0,
                                            uint32_t hsize, uint32_t (&cell)[2], float (&frac)[2], u2_t (&raw)[4]) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float pos = fmaf(scale, xy[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        cell[d] = (uint32_t)(int32_t)fl;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t cc[2] = {cell[0] + (uint32_t)(c & 1), cell[1] + (uint32_t)((c >> 1) & 1)};
        raw[c] = *reinterpret_cast<const u2_t*>(table + ((size_t)row0 + grid_row<2>(cc, res, hsize)) * kF);
    }
}
// encode_level<2, 4>'s arithmetic on already gathered entries: acc[f] = fma(w_c, v_c[f], acc[f]) over the corners in order
__device__ __forceinline__ void blend_quad(const u2_t (&raw)[4], const float (&frac)[2], float (&acc)[kF]) {
#pragma unroll
    for (int f = 0; f < kF; ++f) acc[f] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float wc = 1.0f;
#pragma unroll
        for (int d = 0; d < 2; ++d) wc = wc * ((c & (1 << d)) ? frac[d] : (1.0f - frac[d]));
        const h4_t v = __builtin_bit_cast(h4_t, raw[c]);
#pragma unroll
        for (int f = 0; f < kF; ++f) acc[f] = fmaf(wc, (float)v[f], acc[f]);
    }
}

// Two lanes per sample (lane = 2 * sample + x-bit, 32 samples per pass, two passes per item): a lane gathers the two corners with
// its x-bit of a slice, so a gather instruction fetches BOTH x-neighbours of 32 samples -- the same 16-byte pair for even cells,
// the same 128-byte line otherwise -- and the kernel, which is bound by the L1 look-ups of its ~200 eight-byte gathers per sample,
// makes half of them.  The even lane blends features 0, 1 of all four corners, the odd lane features 2, 3 (the partner's halves
// arrive by a quad swap), each in the corner order of encode_level<2, 4>; the odd lane then hands its two rounded features per
// slice to the even lane, which runs blend_reduce as the one-lane form does: bit-identical outputs.
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) {  // value of lane ^ 1
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ float pair_swap_f(float v) { return __builtin_bit_cast(float, pair_swap(__builtin_bit_cast(uint32_t, v))); }

// this lane's two corners (x-bit xb, y = 0, 1) of the cell of xy
__device__ __forceinline__ void gather_pair(const float (&xy)[2], uint32_t xb, const _Float16* __restrict__ table, float scale, uint32_t res,
                                            uint32_t row0, uint32_t hsize, uint32_t (&cell)[2], float (&frac)[2], u2_t (&raw)[2]) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float pos = fmaf(scale, xy[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        cell[d] = (uint32_t)(int32_t)fl;
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const uint32_t cc[2] = {cell[0] + xb, cell[1] + (uint32_t)y};
        raw[y] = *reinterpret_cast<const u2_t*>(table + ((size_t)row0 + grid_row<2>(cc, res, hsize)) * kF);
    }
}
// features (2 xb, 2 xb + 1) of the cell, blended over the four corners in encode_level's order, rounded to fp16 as the slice encoders return them
__device__ __forceinline__ void blend_pair(const u2_t (&raw)[2], uint32_t xb, const float (&frac)[2], float (&f)[2]) {
    uint32_t mine[4];  // this lane's feature pair of corner c = x + 2 y
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const uint32_t own = xb ? raw[y][1] : raw[y][0], send = xb ? raw[y][0] : raw[y][1];
        const uint32_t recv = pair_swap(send);
        mine[2 * y] = xb ? recv : own;
        mine[2 * y + 1] = xb ? own : recv;
    }
    f[0] = f[1] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float wc = 1.0f;
#pragma unroll
        for (int d = 0; d < 2; ++d) wc = wc * ((c & (1 << d)) ? frac[d] : (1.0f - frac[d]));
        const h2_t v = __builtin_bit_cast(h2_t, mine[c]);
        f[0] = fmaf(wc, (float)v[0], f[0]);
        f[1] = fmaf(wc, (float)v[1], f[1]);
    }
    f[0] = r16(f[0]);
    f[1] = r16(f[1]);
}

__global__ __launch_bounds__(kBlock) void k_hash_dynamic3(const float* __restrict__ x, uint32_t x_stride, const float* __restrict__ off,
                                                          uint32_t off_stride, uint32_t M, PlaneSet3 ps, float* __restrict__ out0,
                                                          _Float16* __restrict__ out1, _Float16* __restrict__ out2) {
    __shared__ float stage[3][kSamplesPerBlock][3 * kPlaneLevels + 1];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform by construction: level tables, pointers and loop control in SGPRs
    const uint32_t xb = (uint32_t)(lane & 1);
    float p[2][3][3];  // [pass][evaluation][axis]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t m = blockIdx.x * kSamplesPerBlock + 32u * h + (uint32_t)(lane >> 1);
        const uint32_t mm = m < M ? m : M - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float xd = x[(size_t)mm * x_stride + d];
            p[h][0][d] = xd;
            p[h][1][d] = ps.enabled[1] ? xd + off[(size_t)mm * off_stride + d] : xd;
            p[h][2][d] = ps.enabled[2] ? xd + off[(size_t)mm * off_stride + 3 + d] : xd;
        }
    }
    for (int item = wave; item < 3 * kPlaneLevels; item += 4) {
        const int pl = item / kPlaneLevels, l = item - pl * kPlaneLevels;
        const int ia = pl == 2 ? 1 : 0, ib = pl == 0 ? 1 : 2;
        const GridMeta& g = ps.meta[pl];
        const float scale = g.scale[l];
        const uint32_t res = g.res[l], row0 = g.offset[l], rows = g.offset[l + 1] - g.offset[l];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 32 * h + (lane >> 1);
            // ---- base evaluation: gathers kept
            uint32_t cell0[2];
            float frac0[2];
            u2_t lo0[2], hi0[2];
            {
                const float xy[2] = {p[h][0][ia], p[h][0][ib]};
                gather_pair(xy, xb, ps.ev[0].table_lo[pl], scale, res, row0, rows, cell0, frac0, lo0);
                if (!ps.ev[0].same_slice) {
                    uint32_t c_[2];
                    float f_[2];
                    gather_pair(xy, xb, ps.ev[0].table_hi[pl], scale, res, row0, rows, c_, f_, hi0);
                } else {
                    hi0[0] = lo0[0];
                    hi0[1] = lo0[1];
                }
                float a[2], b[2] = {0.0f, 0.0f};
                blend_pair(lo0, xb, frac0, a);
                if (!ps.ev[0].same_slice) blend_pair(hi0, xb, frac0, b);
                // features 2, 3 come from the odd lane; the even lane reduces
                const float a2 = pair_swap_f(a[0]), a3 = pair_swap_f(a[1]), b2 = pair_swap_f(b[0]), b3 = pair_swap_f(b[1]);
                const float f_lo[kF] = {a[0], a[1], a2, a3}, f_hi[kF] = {b[0], b[1], b2, b3};
                if (xb == 0u) stage[0][row][item] = blend_reduce<0>(f_lo, f_hi, ps.ev[0]);
            }
            // ---- neighbour evaluations (regime 1)
#pragma unroll
            for (int e = 1; e < 3; ++e) {
                if (!ps.enabled[e]) continue;
                const TimeSet& pe = ps.ev[e];
                const float xy[2] = {p[h][e][ia], p[h][e][ib]};
                uint32_t cell[2];
                float frac[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const float pos = fmaf(scale, xy[d], 0.5f);
                    const float fl = floorf(pos);
                    frac[d] = pos - fl;
                    cell[d] = (uint32_t)(int32_t)fl;
                }
                u2_t lo[2], hi[2];
                const bool reuse = ps.share[e] && cell[0] == cell0[0] && cell[1] == cell0[1];  // the same for both lanes of a sample
                if (reuse) {
#pragma unroll
                    for (int y = 0; y < 2; ++y) { lo[y] = lo0[y]; hi[y] = hi0[y]; }
                } else {
                    uint32_t c_[2];
                    float f_[2];
                    gather_pair(xy, xb, pe.table_lo[pl], scale, res, row0, rows, c_, f_, lo);
                    if (!pe.same_slice) gather_pair(xy, xb, pe.table_hi[pl], scale, res, row0, rows, c_, f_, hi);
                    else { hi[0] = lo[0]; hi[1] = lo[1]; }
                }
                float a[2], b[2] = {0.0f, 0.0f};
                blend_pair(lo, xb, frac, a);
                if (!pe.same_slice) blend_pair(hi, xb, frac, b);
                const float a2 = pair_swap_f(a[0]), a3 = pair_swap_f(a[1]), b2 = pair_swap_f(b[0]), b3 = pair_swap_f(b[1]);
                const float f_lo[kF] = {a[0], a[1], a2, a3}, f_hi[kF] = {b[0], b[1], b2, b3};
                if (xb == 0u) stage[e][row][item] = blend_reduce<1>(f_lo, f_hi, pe);
            }
        }
    }
    __syncthreads();
    const uint32_t n_rows = min((uint32_t)kSamplesPerBlock, M - blockIdx.x * kSamplesPerBlock);
    constexpr int kOut = 3 * kPlaneLevels;
    for (uint32_t c = threadIdx.x; c < n_rows * kOut; c += kBlock) {
        const uint32_t row = c / kOut, col = c - row * kOut;
        const size_t o = (size_t)(blockIdx.x * kSamplesPerBlock + row) * kOut + col;
        out0[o] = stage[0][row][col];
        if (ps.enabled[1]) out1[o] = (_Float16)stage[1][row][col];
        if (ps.enabled[2]) out2[o] = (_Float16)stage[2][row][col];
    }
}

// Flow-field front end (flow_field.py:123-128): 3-D grid (L levels, F = 8) -> .float() -> Lagrange reduction over the
// two groups of 4 features of every level -> fp32 [M, 2 L].
__global__ __launch_bounds__(kBlock) void k_hash3d_lagrange(const float* __restrict__ x, uint32_t x_stride, uint32_t M,
                                                            const _Float16* __restrict__ table, uint32_t L, GridMeta meta, float w0, float w1,
                                                            float w2, float w3, float* __restrict__ out) {
    extern __shared__ float stage_dyn[];  // [64][2L + 1]
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform by construction: level tables, pointers and loop control in SGPRs
    const uint32_t m = blockIdx.x * kSamplesPerBlock + lane;
    const uint32_t mm = m < M ? m : M - 1;
    const uint32_t pitch = 2 * L + 1;
    const float p[3] = {x[(size_t)mm * x_stride], x[(size_t)mm * x_stride + 1], x[(size_t)mm * x_stride + 2]};
    const float w[4] = {w0, w1, w2, w3};
    for (uint32_t l = wave; l < L; l += 4) {
        float f[8];
        encode_level<3, 8>(p, table, meta.scale[l], meta.res[l], meta.offset[l], meta.offset[l + 1] - meta.offset[l], f);
        // x.view(-1, L, 8) chunked into 4 along the feature axis: chunk i = features (2i, 2i+1)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float acc = w[0] * r16(f[e]);
#pragma unroll
            for (int i = 1; i < 4; ++i) acc = acc + w[i] * r16(f[2 * i + e]);
            stage_dyn[lane * pitch + 2 * l + e] = acc;
        }
    }
    __syncthreads();
    const uint32_t n_rows = min((uint32_t)kSamplesPerBlock, M - blockIdx.x * kSamplesPerBlock);
    for (uint32_t c = threadIdx.x; c < n_rows * 2 * L; c += kBlock) {
        const uint32_t row = c / (2 * L), col = c - row * 2 * L;
        out[(size_t)(blockIdx.x * kSamplesPerBlock + row) * 2 * L + col] = stage_dyn[row * pitch + col];
    }
}

// Table gradients of the fused space-time encoder, regime 0 (fp32 blend / reduction: the current frame, the only call that
// records a gradient -- the neighbour frames are evaluated under no_grad, network_dynamic.py:242-271):
//   out[pair][level] = sum_i lag_i (blend_lo f_lo[i] + blend_hi f_hi[i]),  f_*[i] = sum_corners w_c table_*[row_c][i]
//   => d L / d table_lo[row_c][i] += g[pair][level] lag_i blend_lo w_c      (the fp16 rounding of f passes the gradient)
// One launch for the three pairs and both slices instead of six hash-grid backward launches fed by ~30 elementwise kernels
// that materialise per-slice gradient matrices.  Same scheme as k_hashgrid_bwd_corners (hashgrid.hip): an item = (chunk of
// rows, pair, level, slice), 16 lanes per item (lane = corner x 4 + feature), the sums of the current cell in registers,
// one atomic instruction per cell change covering whole 16-byte table entries.
struct PlaneGrads {
    float* g_lo[3];
    float* g_hi[3];
    GridMeta meta[3];
    float blend_lo, blend_hi;
    float lag[4];
    int same_slice;
};

__global__ __launch_bounds__(kBlock) void k_hash_dynamic_bwd(const float* __restrict__ x, uint32_t x_stride, uint32_t M,
                                                             const float* __restrict__ grad_out, PlaneGrads pg, uint32_t run) {
    const int lane = lane_id();
    const int sub = lane >> 4, r = lane & 15, c = r >> 2, f = r & 3;
    const uint32_t n_sl = pg.same_slice ? 1u : 2u;
    const uint32_t per_chunk = 3u * kPlaneLevels * n_sl;
    const unsigned long long item = ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 4ull + (unsigned)sub;
    const uint32_t chunk = (uint32_t)(item / per_chunk), rest = (uint32_t)(item - (unsigned long long)chunk * per_chunk);
    const uint32_t pl = rest / (kPlaneLevels * n_sl), l = (rest / n_sl) % kPlaneLevels, sl = rest % n_sl;
    const unsigned long long first = (unsigned long long)chunk * run;
    if (first >= M) return;
    const uint32_t m0 = (uint32_t)first, m1 = (uint32_t)(first + run < M ? first + run : M);
    const GridMeta& g = pg.meta[pl];
    const float scale = g.scale[l];
    const uint32_t res = g.res[l], row0 = g.offset[l], hsize = g.offset[l + 1] - row0;
    float* table = sl ? pg.g_hi[pl] : pg.g_lo[pl];
    const float factor = pg.lag[f] * (pg.same_slice ? 1.0f : (sl ? pg.blend_hi : pg.blend_lo));
    const uint32_t ca = pl == 2 ? 1u : 0u, cb = pl == 0 ? 1u : 2u;  // coordinate columns of the pair: (x,y), (x,z), (y,z)
    float acc = 0.0f;
    uint32_t cur0 = 0u, cur1 = 0u;
    bool have = false;
    float* dst = table;
    float xa_n = x[(size_t)m0 * x_stride + ca], xb_n = x[(size_t)m0 * x_stride + cb];
    float g_n = grad_out[(size_t)m0 * (3 * kPlaneLevels) + pl * kPlaneLevels + l];
    for (uint32_t m = m0; m < m1; ++m) {
        const float xa = xa_n, xb = xb_n, go = g_n;
        if (m + 1 < m1) {
            xa_n = x[(size_t)(m + 1) * x_stride + ca];
            xb_n = x[(size_t)(m + 1) * x_stride + cb];
            g_n = grad_out[(size_t)(m + 1) * (3 * kPlaneLevels) + pl * kPlaneLevels + l];
        }
        if (go == 0.0f) continue;
        const float pa = fmaf(scale, xa, 0.5f), pb = fmaf(scale, xb, 0.5f);
        const float fa = floorf(pa), fb = floorf(pb);
        const float ra = pa - fa, rb = pb - fb;
        const uint32_t ia = (uint32_t)(int32_t)fa, ib = (uint32_t)(int32_t)fb;
        const float w = ((c & 1) ? ra : (1.0f - ra)) * ((c & 2) ? rb : (1.0f - rb));
        if (!(have && ia == cur0 && ib == cur1)) {
            if (acc != 0.0f) atomicAdd(dst, acc);
            acc = 0.0f;
            have = true;
            cur0 = ia; cur1 = ib;
            const uint32_t cc[2] = {ia + (uint32_t)(c & 1), ib + (uint32_t)((c >> 1) & 1)};
            dst = table + ((size_t)row0 + grid_row<2>(cc, res, hsize)) * kF + f;
        }
        acc += w * (go * factor);
    }
    if (acc != 0.0f) atomicAdd(dst, acc);
}

// The scalar form of the same gradient: the four features of a table entry are the four Lagrange chunks, so their gradients
// are lag_i * blend * G[row] with ONE scattered sum G[row] = sum_samples g[pair][level] * w_corner per entry (and per pair,
// shared by the two slices).  Items = (chunk, pair, level) on 4 lanes (lane = corner), 16 items per wave: a sixteenth of the
// atomic floats and a quarter of the atomic instructions of k_hash_dynamic_bwd; the host expands G to the slice gradients.
struct PlaneSums {
    float* g[3];       // per pair: fp32 [rows]
    GridMeta meta[3];
};
__global__ __launch_bounds__(kBlock) void k_hash_dynamic_bwd_scalar(const float* __restrict__ x, uint32_t x_stride, uint32_t M,
                                                                    const float* __restrict__ grad_out, PlaneSums pg, uint32_t run) {
    const int lane = lane_id();
    const int sub = lane >> 2, c = lane & 3;
    constexpr uint32_t per_chunk = 3u * kPlaneLevels;
    const unsigned long long item = ((unsigned long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 16ull + (unsigned)sub;
    const uint32_t chunk = (uint32_t)(item / per_chunk), rest = (uint32_t)(item - (unsigned long long)chunk * per_chunk);
    const uint32_t pl = rest / kPlaneLevels, l = rest % kPlaneLevels;
    const unsigned long long first = (unsigned long long)chunk * run;
    if (first >= M) return;
    const uint32_t m0 = (uint32_t)first, m1 = (uint32_t)(first + run < M ? first + run : M);
    const GridMeta& g = pg.meta[pl];
    const float scale = g.scale[l];
    const uint32_t res = g.res[l], row0 = g.offset[l], hsize = g.offset[l + 1] - row0;
    float* table = pg.g[pl];
    const uint32_t ca = pl == 2 ? 1u : 0u, cb = pl == 0 ? 1u : 2u;
    float acc = 0.0f;
    uint32_t cur0 = 0u, cur1 = 0u;
    bool have = false;
    float* dst = table;
    float xa_n = x[(size_t)m0 * x_stride + ca], xb_n = x[(size_t)m0 * x_stride + cb];
    float g_n = grad_out[(size_t)m0 * (3 * kPlaneLevels) + pl * kPlaneLevels + l];
    for (uint32_t m = m0; m < m1; ++m) {
        const float xa = xa_n, xb = xb_n, go = g_n;
        if (m + 1 < m1) {
            xa_n = x[(size_t)(m + 1) * x_stride + ca];
            xb_n = x[(size_t)(m + 1) * x_stride + cb];
            g_n = grad_out[(size_t)(m + 1) * (3 * kPlaneLevels) + pl * kPlaneLevels + l];
        }
        if (go == 0.0f) continue;
        const float pa = fmaf(scale, xa, 0.5f), pb = fmaf(scale, xb, 0.5f);
        const float fa = floorf(pa), fb = floorf(pb);
        const float ra = pa - fa, rb = pb - fb;
        const uint32_t ia = (uint32_t)(int32_t)fa, ib = (uint32_t)(int32_t)fb;
        const float w = ((c & 1) ? ra : (1.0f - ra)) * ((c & 2) ? rb : (1.0f - rb));
        if (!(have && ia == cur0 && ib == cur1)) {
            if (acc != 0.0f) atomicAdd(dst, acc);
            acc = 0.0f;
            have = true;
            cur0 = ia; cur1 = ib;
            const uint32_t cc[2] = {ia + (uint32_t)(c & 1), ib + (uint32_t)((c >> 1) & 1)};
            dst = table + ((size_t)row0 + grid_row<2>(cc, res, hsize));
        }
        acc += w * go;
    }
    if (acc != 0.0f) atomicAdd(dst, acc);
}

// The same sums through LDS (the form nvsf_hashgrid4d_dynamic_bwd_scalar launches).  A time-sliced 2-D grid has 2^13 - 2^15
// rows per level (hash_size_dynamic = [15, 13, 13], hash_field.py:96): ONE level of ONE pair is a small table of scalar sums that
// fits the 160 KB of LDS of a CU.  So a workgroup owns (pair, level, row range, slice of the samples), accumulates its slice into an
// LDS image of its rows (no memory-side atomic per sample and corner: those cost one 64-byte segment each, 2.96 ms per 1.6 M
// samples in the run-merging kernel above) and adds the image to the global sums once, with contiguous atomics (the full-rate
// shape).  The image is fp64 (ds_add_f64): an addend w g is an fp32 number, its conversion is exact, and the fp64 sum of a slice's
// addends rounds at 2^-53 of the running sum -- the arrival order of the waves changes the image by ~1e-13 of an entry, far below the
// one fp32 rounding on the way out.  (Rounds 2-5 used 64-bit fixed point, scale 2^(36 - e) from the largest |g| of the slice and
// level: that needs a pass over the gradient column before the walk.  Round 6 measured the fp64 form: config-5 step 28.9-29.2 against
// 29.5-29.7 ms on the same box -- one read of the column instead of two, a one-instruction conversion instead of the float -> int64
// sequence; the LDS fp64 adder is no slower here than the integer one was.)  A non-finite gradient (fp16 overflow under GradScaler)
// travels through the sums as through the memory-side atomics: the rows it touches end up non-finite and found_inf skips the step
// (ADVICE r3).  2^15-row levels take two workgroups per slice, one per half of the rows (128 KB each); both walk the slice, each
// keeps the addends of its rows.
template <int SPLIT>
__global__ void k_hash_dynamic_bwd_lds(const float* __restrict__ x, uint32_t x_stride, uint32_t M, const float* __restrict__ grad_out,
                                       PlaneSums pg, uint32_t pl0, uint32_t chunk_len, uint32_t go_row, uint32_t go_col) {
    // grad_out[m * go_row + column * go_col]: rows [M, 24] (24, 1) or COLUMN-major (1, M) -- a workgroup reads ONE column of all its
    // samples: 4 bytes of every 96-byte row in the row form (every line of the matrix fetched by each of the 24 columns' workgroups:
    // 2.4 GB of line traffic per 1.57 M samples), contiguous in the column form
    extern __shared__ double s_sum[];
    const uint32_t part = blockIdx.y % SPLIT, yl = blockIdx.y / SPLIT;
    const uint32_t pl = pl0 + yl / kPlaneLevels, l = yl % kPlaneLevels;
    const GridMeta& g = pg.meta[pl];
    const float scale = g.scale[l];
    const uint32_t res = g.res[l], row0 = g.offset[l], hsize = g.offset[l + 1] - row0;
    const uint32_t rows_wg = (hsize + SPLIT - 1) / SPLIT, row_lo = part * rows_wg;
    const uint32_t n_rows = row_lo < hsize ? (hsize - row_lo < rows_wg ? hsize - row_lo : rows_wg) : 0u;
    const uint32_t ca = pl == 2 ? 1u : 0u, cb = pl == 0 ? 1u : 2u;
    const unsigned long long first = (unsigned long long)blockIdx.x * chunk_len;
    const uint32_t m0 = (uint32_t)(first < M ? first : M), m1 = (uint32_t)(first + chunk_len < M ? first + chunk_len : M);
    const size_t gcol = ((size_t)pl * kPlaneLevels + l) * go_col;
    if (n_rows == 0u || m0 >= m1) return;  // uniform
    for (uint32_t i = threadIdx.x; i < n_rows; i += blockDim.x) s_sum[i] = 0.0;
    __syncthreads();
    // four samples per thread and round: their (strided, line-per-sample) loads are in flight together
    constexpr int U = 4;
    for (uint32_t mb = m0 + threadIdx.x; mb < m1; mb += U * blockDim.x) {
        float go[U], xa[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t m = mb + (uint32_t)u * blockDim.x;
            const bool ok = m < m1;
            const size_t mm = ok ? m : m0;
            go[u] = ok ? grad_out[mm * go_row + gcol] : 0.0f;
            xa[u] = x[mm * x_stride + ca];
            xb[u] = x[mm * x_stride + cb];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (go[u] == 0.0f) continue;
            const float pa = fmaf(scale, xa[u], 0.5f), pb = fmaf(scale, xb[u], 0.5f);
            const float fa = floorf(pa), fb = floorf(pb);
            const float ra = pa - fa, rb = pb - fb;
            const uint32_t ia = (uint32_t)(int32_t)fa, ib = (uint32_t)(int32_t)fb;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float w = ((c & 1) ? ra : (1.0f - ra)) * ((c & 2) ? rb : (1.0f - rb));
                const uint32_t cc[2] = {ia + (uint32_t)(c & 1), ib + (uint32_t)((c >> 1) & 1)};
                const uint32_t row = grid_row<2>(cc, res, hsize) - row_lo;
                const float v = w * go[u];
                if (row < n_rows && v != 0.0f) atomicAdd(&s_sum[row], (double)v);
            }
        }
    }
    __syncthreads();
    float* table = pg.g[pl] + row0 + row_lo;
    for (uint32_t i = threadIdx.x; i < n_rows; i += blockDim.x) {
        const double v = s_sum[i];
        if (v != 0.0) atomicAdd(table + i, (float)v);
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

// tables: 6 device pointers (lo slice of pair 0,1,2 then hi slice of pair 0,1,2); h_scales / h_res: [3][8];
// h_offsets: [3][9]; h_time: {blend_lo, blend_hi, w0, w1, w2, w3}
NVSF_API int nvsf_hashgrid4d_dynamic_fwd(const float* x, uint32_t x_stride, const float* offset, uint32_t off_stride, uint32_t off_col,
                                         uint32_t M, const void* const* h_tables_f16, const float* h_scales, const uint32_t* h_res,
                                         const uint32_t* h_offsets, const float* h_time, int same_slice, int mode, void* out,
                                         hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && h_tables_f16 && h_time && out && x_stride >= 3 && (mode == 0 || mode == 1));
    REQUIRE(!offset || off_stride >= off_col + 3);
    PlaneSet ps;
    for (int p = 0; p < 3; ++p) {
        ps.table_lo[p] = reinterpret_cast<const _Float16*>(h_tables_f16[p]);
        ps.table_hi[p] = reinterpret_cast<const _Float16*>(h_tables_f16[3 + p]);
        REQUIRE(ps.table_lo[p] && (same_slice || ps.table_hi[p]));
        REQUIRE((reinterpret_cast<uintptr_t>(ps.table_lo[p]) & 7u) == 0 && (reinterpret_cast<uintptr_t>(ps.table_hi[p]) & 7u) == 0);
        const int st = fill_meta(ps.meta[p], kPlaneLevels, h_scales + p * kPlaneLevels, h_res + p * kPlaneLevels, h_offsets + p * (kPlaneLevels + 1));
        if (st != NVSF_OK) return st;
    }
    ps.blend_lo = h_time[0]; ps.blend_hi = h_time[1];
    for (int i = 0; i < 4; ++i) ps.lag[i] = h_time[2 + i];
    ps.same_slice = same_slice;
    const dim3 grid(cdiv(M, kSamplesPerBlock)), block(kBlock);
    if (mode == 0) hipLaunchKernelGGL(k_hash_dynamic<0>, grid, block, 0, stream, x, x_stride, offset, off_stride, off_col, M, ps, out);
    else hipLaunchKernelGGL(k_hash_dynamic<1>, grid, block, 0, stream, x, x_stride, offset, off_stride, off_col, M, ps, out);
    return nvsf_launch_status();
}

NVSF_API int nvsf_hashgrid3d_lagrange_fwd(const float* x, uint32_t x_stride, uint32_t M, const void* table_f16, uint32_t L, uint32_t F,
                                          const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, const float* h_weights4,
                                          float* out, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && table_f16 && h_weights4 && out && x_stride >= 3);
    if (F != 8) return NVSF_ERR_UNSUPPORTED;
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    const size_t lds = (size_t)kSamplesPerBlock * (2 * L + 1) * sizeof(float);
    hipLaunchKernelGGL(k_hash3d_lagrange, dim3(cdiv(M, kSamplesPerBlock)), dim3(kBlock), lds, stream, x, x_stride, M,
                       reinterpret_cast<const _Float16*>(table_f16), L, meta, h_weights4[0], h_weights4[1], h_weights4[2], h_weights4[3], out);
    return nvsf_launch_status();
}

// grad tables: 6 device pointers to fp32 buffers in the layout of the slice tables (lo slice of pair 0,1,2 then hi slice of pair 0,1,2;
// the hi pointers are ignored when same_slice); dL/dtable is ADDED to them.  grad_out fp32 [M, 24] (regime 0 only).
NVSF_API int nvsf_hashgrid4d_dynamic_bwd(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                         const uint32_t* h_offsets, const float* h_time, int same_slice, const float* grad_out,
                                         void* const* h_grad_tables_f32, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && h_time && grad_out && h_grad_tables_f32 && x_stride >= 3);
    PlaneGrads pg;
    for (int p = 0; p < 3; ++p) {
        pg.g_lo[p] = reinterpret_cast<float*>(h_grad_tables_f32[p]);
        pg.g_hi[p] = reinterpret_cast<float*>(h_grad_tables_f32[3 + p]);
        REQUIRE(pg.g_lo[p] && (same_slice || pg.g_hi[p]));
        const int st = fill_meta(pg.meta[p], kPlaneLevels, h_scales + p * kPlaneLevels, h_res + p * kPlaneLevels, h_offsets + p * (kPlaneLevels + 1));
        if (st != NVSF_OK) return st;
    }
    pg.blend_lo = h_time[0]; pg.blend_hi = h_time[1];
    for (int i = 0; i < 4; ++i) pg.lag[i] = h_time[2 + i];
    pg.same_slice = same_slice;
    const uint32_t run = M >= (1u << 20) ? 128u : 32u;
    const unsigned long long items = (unsigned long long)cdiv(M, run) * 3ull * kPlaneLevels * (same_slice ? 1u : 2u);
    const unsigned long long waves = (items + 3) / 4;
    hipLaunchKernelGGL(k_hash_dynamic_bwd, dim3((uint32_t)((waves + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, stream, x, x_stride, M,
                       grad_out, pg, run);
    return nvsf_launch_status();
}

// The three evaluations of one density query.  h_tables_f16: 18 device pointers = for evaluation e = 0, 1, 2: lo slice of pair 0,1,2 then
// hi slice of pair 0,1,2; h_time: 18 floats = per evaluation {blend_lo, blend_hi, w0, w1, w2, w3}; h_flags: 9 ints = per evaluation
// {enabled, same_slice, shares the slice tables of evaluation 0}.  out0 fp32 [M,24] (regime 0), out1 / out2 fp16 [M,24] (regime 1).
// offsets fp32 [M, off_stride >= 6]: columns 0..2 warp evaluation 1, columns 3..5 evaluation 2.
NVSF_API int nvsf_hashgrid4d_dynamic3_fwd(const float* x, uint32_t x_stride, const float* offsets, uint32_t off_stride, uint32_t M,
                                          const void* const* h_tables_f16, const float* h_scales, const uint32_t* h_res,
                                          const uint32_t* h_offsets, const float* h_time, const int* h_flags, float* out0, void* out1,
                                          void* out2, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && h_tables_f16 && h_time && h_flags && out0 && x_stride >= 3);
    PlaneSet3 ps;
    for (int p = 0; p < 3; ++p) {
        const int st = fill_meta(ps.meta[p], kPlaneLevels, h_scales + p * kPlaneLevels, h_res + p * kPlaneLevels, h_offsets + p * (kPlaneLevels + 1));
        if (st != NVSF_OK) return st;
    }
    for (int e = 0; e < 3; ++e) {
        ps.enabled[e] = e == 0 ? 1 : h_flags[3 * e];
        ps.share[e] = h_flags[3 * e + 2];
        TimeSet& pe = ps.ev[e];
        pe.same_slice = h_flags[3 * e + 1];
        for (int p = 0; p < 3; ++p) {
            pe.table_lo[p] = reinterpret_cast<const _Float16*>(h_tables_f16[6 * e + p]);
            pe.table_hi[p] = reinterpret_cast<const _Float16*>(h_tables_f16[6 * e + 3 + p]);
            if (ps.enabled[e]) {
                REQUIRE(pe.table_lo[p] && (pe.same_slice || pe.table_hi[p]));
                REQUIRE((reinterpret_cast<uintptr_t>(pe.table_lo[p]) & 7u) == 0 && (reinterpret_cast<uintptr_t>(pe.table_hi[p]) & 7u) == 0);
            }
        }
        pe.blend_lo = h_time[6 * e]; pe.blend_hi = h_time[6 * e + 1];
        for (int i = 0; i < 4; ++i) pe.lag[i] = h_time[6 * e + 2 + i];
        if (e > 0 && ps.enabled[e]) {
            REQUIRE(offsets && off_stride >= 6 && (e == 1 ? out1 : out2));
            if (ps.share[e]) REQUIRE(pe.same_slice == ps.ev[0].same_slice);
        }
    }
    hipLaunchKernelGGL(k_hash_dynamic3, dim3(cdiv(M, kSamplesPerBlock)), dim3(kBlock), 0, stream, x, x_stride, offsets, off_stride, M, ps, out0,
                       reinterpret_cast<_Float16*>(out1), reinterpret_cast<_Float16*>(out2));
    return nvsf_launch_status();
}

// Scalar table-gradient sums of the space-time encoder: h_sums_f32 = 3 device pointers (one per pair) to fp32 [rows] buffers; the sum
// G[row] = sum over samples of grad_out[pair][level] * w_corner is ADDED to them.  dL/dtable[slice][row][i] = lag_i * blend_slice * G[row].
static int hash4d_bwd_scalar_impl(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                  const uint32_t* h_offsets, const float* grad_out, int grad_col_major, void* const* h_sums_f32, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && grad_out && h_sums_f32 && x_stride >= 3);
    PlaneSums pg;
    for (int p = 0; p < 3; ++p) {
        pg.g[p] = reinterpret_cast<float*>(h_sums_f32[p]);
        REQUIRE(pg.g[p]);
        const int st = fill_meta(pg.meta[p], kPlaneLevels, h_scales + p * kPlaneLevels, h_res + p * kPlaneLevels, h_offsets + p * (kPlaneLevels + 1));
        if (st != NVSF_OK) return st;
    }
    // LDS form: one launch per group of pairs with the same level size (pair 0: 2^15 rows = two row ranges of 128 KB of 64-bit sums,
    // 1024-thread workgroups; pairs 1, 2: 2^13 rows = 64 KB, 512 threads).  Variant 1 (tests) selects the run-merging global-atomic
    // kernel (the reference form; also taken when a level does not fit LDS or the batch is too small to fill the chip with slices).
    uint32_t max_rows[3];
    bool fits = true;
    for (int p = 0; p < 3; ++p) {
        max_rows[p] = 0;
        for (uint32_t l = 0; l < (uint32_t)kPlaneLevels; ++l) {
            const uint32_t rows = pg.meta[p].offset[l + 1] - pg.meta[p].offset[l];
            max_rows[p] = rows > max_rows[p] ? rows : max_rows[p];
        }
        fits = fits && max_rows[p] * sizeof(double) <= 2u * 128u * 1024u;
    }
    const uint32_t go_row = grad_col_major ? 1u : (uint32_t)(3 * kPlaneLevels), go_col = grad_col_major ? M : 1u;
    if (!(fits && M >= (1u << 16) && nvsf_variant(kVarHash4dBwd) == 0) && grad_col_major) return NVSF_ERR_UNSUPPORTED;  // rows for the other kernel
    if (fits && M >= (1u << 16) && nvsf_variant(kVarHash4dBwd) == 0) {
        int p = 0;
        while (p < 3) {
            int q = p + 1;
            while (q < 3 && max_rows[q] == max_rows[p]) ++q;  // consecutive pairs with equal level size share a launch
            const uint32_t split = max_rows[p] * (uint32_t)sizeof(double) > 128u * 1024u ? 2u : 1u;
            const uint32_t lds = (max_rows[p] + split - 1) / split * (uint32_t)sizeof(double);
            const uint32_t threads = lds > 64u * 1024u ? 1024u : 512u;
            const uint32_t n_slices = lds > 64u * 1024u ? 32u : 64u;
            const uint32_t chunk_len = (M + n_slices - 1) / n_slices;
            const dim3 grid(n_slices, (uint32_t)(q - p) * kPlaneLevels * split);
            if (split == 2)
                hipLaunchKernelGGL(k_hash_dynamic_bwd_lds<2>, grid, dim3(threads), lds, stream, x, x_stride, M, grad_out, pg, (uint32_t)p, chunk_len, go_row, go_col);
            else
                hipLaunchKernelGGL(k_hash_dynamic_bwd_lds<1>, grid, dim3(threads), lds, stream, x, x_stride, M, grad_out, pg, (uint32_t)p, chunk_len, go_row, go_col);
            p = q;
        }
        return nvsf_launch_status();
    }
    const uint32_t run = M >= (1u << 20) ? 128u : 32u;
    const unsigned long long items = (unsigned long long)cdiv(M, run) * 3ull * kPlaneLevels;
    const unsigned long long waves = (items + 15) / 16;
    hipLaunchKernelGGL(k_hash_dynamic_bwd_scalar, dim3((uint32_t)((waves + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, stream, x, x_stride,
                       M, grad_out, pg, run);
    return nvsf_launch_status();
}

NVSF_API int nvsf_hashgrid4d_dynamic_bwd_scalar(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                                const uint32_t* h_offsets, const float* grad_out, void* const* h_sums_f32, hipStream_t stream) {
    return hash4d_bwd_scalar_impl(x, x_stride, M, h_scales, h_res, h_offsets, grad_out, 0, h_sums_f32, stream);
}

// The same with the gradient COLUMN-major, fp32 [24][M] (what nvsf_density_tail_grad_split writes): the LDS kernel's workgroups read one
// column each.  NVSF_ERR_UNSUPPORTED where the launch would take the run-merging kernel (small batches, levels beyond LDS, test variant):
// the caller then passes rows.
NVSF_API int nvsf_hashgrid4d_dynamic_bwd_scalar_t(const float* x, uint32_t x_stride, uint32_t M, const float* h_scales, const uint32_t* h_res,
                                                  const uint32_t* h_offsets, const float* grad_out_t, void* const* h_sums_f32, hipStream_t stream) {
    return hash4d_bwd_scalar_impl(x, x_stride, M, h_scales, h_res, h_offsets, grad_out_t, 1, h_sums_f32, stream);
}
