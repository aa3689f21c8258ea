// Fused per-sample field evaluation for the uniform-sampling render path on gfx950 (the kernels that carry
// BASELINE config 2: hash grid L*F = 32 features -> sigma MLP 32->64->16 -> heads).
//
//   nvsf_field_density_uniform_fwd : ray -> sample position -> hash-grid encode -> sigma MLP -> sigma, geo
//   nvsf_field_heads_uniform_fwd   : (weights, geo, ray dir) -> direction encoding -> colour / LiDAR heads
//                                    -> sigmoid -> image accumulation
//
// Reference path being replaced: renderer_dynamic.py:155-224 + network_dynamic.py:213-332 with
// tcnn.Encoding(HashGrid) / tcnn.Network(FullyFusedMLP) inside (a static-field configuration of it).
//
// Density kernel layout.  A wave evaluates 16 samples per MFMA tile.  Lane l owns sample (l & 15) and the
// feature slice 8*(l>>4) .. +7 of the 32-wide encoding, i.e. it gathers and blends (8/F) levels for that
// sample and thereby produces exactly the B-operand fragment of v_mfma_f32_16x16x32_f16 in registers
// (B[k = 8*(l>>4)+j][n = l&15]).  Encoded features never touch LDS or HBM; hidden activations stay in the
// accumulator layout (mlp_device.h).  The 16 lanes of a group are 16 consecutive samples of one ray, so
// their gathers into one level are spatially adjacent.
#include "hashgrid_device.h"
#include "mlp_device.h"
#include "encodings_device.h"
#include "march_device.h"
// k_render_occupancy_lds asks for 4 waves per SIMD; its LiDAR instantiation settles at 3 (LDS-bound anyway)
#pragma clang diagnostic ignored "-Wpass-failed"
#include <stdlib.h>

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

struct RayBatch {
    const float* rays_o;   // [N,3]
    const float* rays_d;   // [N,3]
    const float* nears;    // [N]
    const float* fars;     // [N]
    const float* lin;      // [T]   torch.linspace(0,1,T)
    const float* noise;    // [N,T] or null
    float lo[3], hi[3];    // aabb
    float inv_extent;      // 1 / (2*bound)
    float bound;
    uint32_t N, T;
};

template <int F>
__global__ __launch_bounds__(kBlock) void k_density_uniform(RayBatch rb, const _Float16* __restrict__ table, GridMeta meta,
                                                            const _Float16* __restrict__ w_sigma, float* __restrict__ z_vals,
                                                            float* __restrict__ sigmas, _Float16* __restrict__ geo) {
    constexpr int kLevelsPerGroup = 8 / F;
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;
    // sigma net: 32 -> 64 -> 16
    half8_t w0[kHidTiles];
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) w0[t] = load_w_natural(w_sigma, 32, t, 0, lane);
    // output rows rotated by one: accumulator row r holds network output (r+1)%16, i.e. rows 0..14 are the
    // geometry features h1..h15 and row 15 is the density logit h0 -> every lane stores an aligned half4.
    OutLayerW wout;
    wout.load(w_sigma + kHidden * 32, lane, 1);

    const unsigned long long total = (unsigned long long)rb.N * rb.T;
    const unsigned long long n_tiles = (total + 15) / 16;
    const unsigned long long wave_global = (unsigned long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const unsigned long long wave_count = (unsigned long long)gridDim.x * kWavesPerBlock;
    for (unsigned long long tile = wave_global; tile < n_tiles; tile += wave_count) {
        const unsigned long long s_raw = tile * 16 + sl;
        const bool in_range = s_raw < total;
        const unsigned long long s = in_range ? s_raw : total - 1;
        const uint32_t n = (uint32_t)(s / rb.T), i = (uint32_t)(s - (unsigned long long)n * rb.T);
        // sample position (renderer_dynamic.py:155-169) and normalisation to [0,1] (network_dynamic.py:217)
        const float near = rb.nears[n], range = rb.fars[n] - near;
        float z = near + range * rb.lin[i];
        if (rb.noise) z = z + (rb.noise[s] - 0.5f) * (range / (float)rb.T);
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float p = rb.rays_o[3 * (size_t)n + k] + rb.rays_d[3 * (size_t)n + k] * z;
            p = fminf(fmaxf(p, rb.lo[k]), rb.hi[k]);
            x[k] = (p + rb.bound) * rb.inv_extent;
        }
        // hash-grid encode: this lane's 8 features = levels g*kLevelsPerGroup .. +kLevelsPerGroup-1
        half8_t xf;
#pragma unroll
        for (int q = 0; q < kLevelsPerGroup; ++q) {
            const int l = g * kLevelsPerGroup + q;
            float acc[F];
            encode_level<3, F>(x, table, meta.scale[l], meta.res[l], meta.offset[l], meta.offset[l + 1] - meta.offset[l], acc);
#pragma unroll
            for (int f = 0; f < F; ++f) xf[q * F + f] = (_Float16)acc[f];
        }
        // sigma MLP
        float4_t acc1[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            const float4_t zero = {0, 0, 0, 0};
            acc1[t] = mfma16(w0[t], xf, zero);
        }
        half8_t h[kHidSteps];
        pack_hidden(acc1, h);
        const float4_t o = wout.apply(h);  // rotated rows 4g..4g+3, column = sample sl
        if (in_range) {
            half4_t ov;
            ov[0] = (_Float16)o[0]; ov[1] = (_Float16)o[1]; ov[2] = (_Float16)o[2]; ov[3] = (_Float16)o[3];
            if (g == 3) {
                z_vals[s] = z;
                sigmas[s] = expf(o[3]);  // trunc_exp forward (activation.py:9-11) on the fp32 density logit
                ov[3] = (_Float16)1.0f;          // geo row = (h1 .. h15, 1.0): 15 features + the ones-padding the heads expect
            }
            *reinterpret_cast<half4_t*>(geo + s * 16 + 4 * g) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Density kernel, second formulation (same arithmetic, same results bit for bit in the encoding):
//   * lane group g takes levels {g, 4+g, 8+g, ...} (interleaved) instead of a contiguous block, so that in
//     unrolled step q all four lane groups work on levels 4q..4q+3: "all dense" / "all hashed" becomes a
//     wave-uniform branch (levels are dense up to `first_hashed` and hashed from there on) and only one
//     index formula is evaluated per step.  The feature order seen by the MLP changes, which is undone by
//     permuting the columns of W0 when its fragment is fetched;
//   * per-lane level constants are read once from an LDS copy of the level table and stay in registers;
//   * gathers are buffer loads with 32-bit byte offsets (no 64-bit address arithmetic per corner);
//   * the corner hashes share their y/z partial terms; the dense index wraps by a conditional subtract
//     (index < 2*rows there) instead of an integer division.
template <int F>
struct LaneLevels {
    static constexpr int Q = 8 / F;
    float scale[Q];
    uint32_t res[Q], res2[Q], boff[Q], rows[Q];
};

// a0 += w * lo(raw), a1 += w * hi(raw) for a table entry of two fp16 features: v_fma_mix_f32 converts the fp16 operand inside
// the fused multiply-add (exact conversion, one rounding: the same result as fmaf(w, (float)h, a)), one VALU instruction per
// feature where hipcc emits two conversions + one packed fma per entry -- 3 issue slots against 2, and the gather kernels are
// bound by VALU issue.
__device__ __forceinline__ void fma_entry(float w, uint32_t raw, float& a0, float& a1) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(a0) : "v"(w), "v"(raw));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a1) : "v"(w), "v"(raw));
}

template <int F>
__device__ __forceinline__ uint32_t gather_raw(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off);
template <>
__device__ __forceinline__ uint32_t gather_raw<2>(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 0);
}
// F = 4: a table entry is 8 bytes (two dwords of two fp16 features), one 8-byte gather
typedef uint32_t entry4_t __attribute__((ext_vector_type(2)));
template <int F> struct EntryOf { typedef uint32_t type; };
template <> struct EntryOf<4> { typedef entry4_t type; };
template <>
__device__ __forceinline__ uint32_t gather_raw<4>(__amdgpu_buffer_rsrc_t, uint32_t) = delete;
__device__ __forceinline__ entry4_t gather_raw4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, 0);
}

// Per-wave constants of the v2 kernel: level table slice, permuted W0 fragment, rotated output fragment.
template <int F>
struct DensityCtx {
    static constexpr int Q = 8 / F;
    LaneLevels<F> lv;
    half8_t w0[kHidTiles];
    OutLayerW wout;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t first_hashed;
    int g;
    const uint4* lds_lv;  // LDS_LV form of density_encode: {scale, res, byte offset, rows} of level 4q + g at [4 q] (the per-ray
                          // render kernels re-read the 16 B per step instead of holding 20 VGPRs across their long tile loop)
};

// One 16-sample tile: hash-grid encode of x in [0,1]^3 (per lane) -> this lane's B fragment of the sigma MLP
// (levels {g, g+4, g+8, g+12} of sample lane & 15).
constexpr uint32_t kFirstHashedC2 = 5;  // L16 F2 T2^19 base 16 -> 2048: 81^3 > 2^19
// FH >= 0: the first hashed level as a compile-time constant (the launcher picks the instance of the grid at hand): the kind of every
// level group -- dense, hashed, mixed -- is then known per unrolled q, the other index form and its selects are not compiled in, and the
// four groups form one basic block.  FH < 0: the run-time value (cx.first_hashed), any grid.
template <int F, int QG, bool LDS_LV = false, int FH = -1>
__device__ __forceinline__ half8_t density_encode(const DensityCtx<F>& cx, const float (&x)[3]) {
    constexpr int Q = 8 / F;
    LaneLevels<F> lv_local;
    if constexpr (LDS_LV) {
        uint32_t again = 0;
        asm volatile("" : "+v"(again));  // re-read per call: hoisted out of the tile loop the constants would pin 20 VGPRs
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint4 v = cx.lds_lv[4 * q + again];
            lv_local.scale[q] = __builtin_bit_cast(float, v.x);
            lv_local.res[q] = v.y;
            lv_local.res2[q] = v.y * v.y;
            lv_local.boff[q] = v.z;
            lv_local.rows[q] = v.w;
        }
    }
    const LaneLevels<F>& lv = LDS_LV ? lv_local : cx.lv;
    const uint32_t first_hashed = FH >= 0 ? (uint32_t)FH : cx.first_hashed;
    const int g = cx.g;
    // phase 1: cell / fraction per level, issue all 8*Q gathers
    half8_t xf;
#pragma unroll
    for (int q0 = 0; q0 < Q; q0 += QG) {
    float frac[Q][3];
    typename EntryOf<F>::type raw[Q][8];
#pragma unroll
    for (int q = q0; q < q0 + QG; ++q) {
        uint32_t c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pos = fmaf(lv.scale[q], x[d], 0.5f);
            const float fl = floorf(pos);
            frac[q][d] = pos - fl;
            c[d] = (uint32_t)(int32_t)fl;
        }
        uint32_t idx[8];
        const bool all_dense = (uint32_t)(4 * q + 3) < first_hashed, all_hashed = (uint32_t)(4 * q) >= first_hashed;
        if (all_dense || !all_hashed) {  // dense rows: c0 + c1*res + c2*res^2, wrapped once
            const uint32_t b00 = c[0] + c[1] * lv.res[q] + c[2] * lv.res2[q];
            const uint32_t b10 = b00 + lv.res[q], b01 = b00 + lv.res2[q], b11 = b10 + lv.res2[q];
            const uint32_t base[4] = {b00, b10, b01, b11};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t v = base[k >> 1] + (uint32_t)(k & 1);
                v = v >= lv.rows[q] ? v - lv.rows[q] : v;
                idx[k] = v;
            }
        }
        if (all_hashed || !all_dense) {
            const uint32_t hy0 = c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
            const uint32_t hz0 = c[2] * 805459861u, hz1 = hz0 + 805459861u;
            const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
            const uint32_t mask = lv.rows[q] - 1u;
            const bool lane_hashed = (uint32_t)(4 * q + g) >= first_hashed;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t v = ((c[0] + (uint32_t)(k & 1)) ^ yz[k >> 1]) & mask;
                idx[k] = (all_hashed || lane_hashed) ? v : idx[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (F == 2) raw[q][k] = gather_raw<2>(cx.rsrc, lv.boff[q] + idx[k] * (uint32_t)(F * sizeof(_Float16)));
            else raw[q][k] = gather_raw4(cx.rsrc, lv.boff[q] + idx[k] * (uint32_t)(F * sizeof(_Float16)));
        }
    }
    // phase 2: trilinear blend (corner order and fma chain of the specification)
#pragma unroll
    for (int q = q0; q < q0 + QG; ++q) {
        const float fx = frac[q][0], fy = frac[q][1], fz = frac[q][2];
        const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy}, wz[2] = {1.0f - fz, fz};
        if constexpr (F == 2) {
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
                fma_entry(w, raw[q][k], a0, a1);
            }
            xf[q * F] = (_Float16)a0;
            xf[q * F + 1] = (_Float16)a1;
        } else {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
                fma_entry(w, raw[q][k][0], a0, a1);
                fma_entry(w, raw[q][k][1], a2, a3);
            }
            xf[q * F] = (_Float16)a0;
            xf[q * F + 1] = (_Float16)a1;
            xf[q * F + 2] = (_Float16)a2;
            xf[q * F + 3] = (_Float16)a3;
        }
    }
    }
    return xf;
}

// encode -> sigma MLP.  Returns the rotated output rows 4g..4g+3 of sample (lane & 15): rows 0..14 = h1..h15,
// row 15 = density logit h0.
template <int F, int QG>
__device__ __forceinline__ float4_t density_eval(const DensityCtx<F>& cx, const float (&x)[3]) {
    const half8_t xf = density_encode<F, QG>(cx, x);
    float4_t acc1[kHidTiles];
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) {
        const float4_t zero = {0, 0, 0, 0};
        acc1[t] = mfma16(cx.w0[t], xf, zero);
    }
    half8_t h[kHidSteps];
    pack_hidden(acc1, h);
    return cx.wout.apply(h);
}

// What the TRAINING forward keeps beside sigma / geo / z (ops.DensityRaysFn): the unit-cube position of every sample (the table
// scatter of the backward reads it), the 32 encoded features as rows in level order (the fused MLP backward recomputes the
// hidden activations from them) and the 16 network outputs in fp32 (the autograd graph carries the geometry features in fp32).
struct TrainOut {
    float* x01;       // [M, 3]
    _Float16* feat;   // [M, 32], column 2 l + f
    float* h32;       // [M, 16] = (h0 .. h15)
    uint32_t* tile;   // device side only: this wave's LDS transpose tile (kTrWords dwords)
};

// The four lane groups of a tile (lanes sl, sl + 16, sl + 32, sl + 48) hold the pieces of sample sl's rows interleaved; both
// helpers below move them between the lane groups first (ds_bpermute: the LDS pipe is idle in these kernels) so that every lane
// stores ONE aligned 16-byte piece and a wave instruction writes 16 whole rows -- written as 4-byte pieces the same rows cost the
// training forward 150 us per 3.1 M samples.  Every lane of the wave must call them (cross-lane reads), `ok` gates the store.
//
// lane group g holds rotated output rows 4g .. 4g+3 = network outputs (4g+1 .. 4g+4) mod 16: output 4g comes from group g - 1
__device__ __forceinline__ void store_h32(float* __restrict__ h32, unsigned long long s, int lane, int g, float o0, float o1, float o2, float o3, bool ok) {
    const int from = (lane + 48) & 63;  // same sample, previous lane group
    const float prev = __int_as_float(__builtin_amdgcn_ds_bpermute(from << 2, __float_as_int(o3)));
    if (ok) *reinterpret_cast<float4*>(h32 + s * 16 + 4 * g) = make_float4(prev, o0, o1, o2);
}

// lane group g holds the encoded features of levels g, 4+g, 8+g, 12+g as the dwords p.x .. p.w; row dword j = level j, and group g
// stores dwords 4g .. 4g+3: a 4 x 4 transpose over the lane groups, through a per-wave LDS tile (rows of 20 dwords: 16-byte
// aligned, at most 2-way bank conflicts)
constexpr int kTrPitch = 20;
constexpr int kTrWords = 16 * kTrPitch;  // per wave
__device__ __forceinline__ void store_feat_row(_Float16* __restrict__ feat, unsigned long long s, int lane, int g, const uint4& p, bool ok,
                                               uint32_t* tile) {
    uint32_t* row = tile + (lane & 15) * kTrPitch;
    row[g] = p.x; row[4 + g] = p.y; row[8 + g] = p.z; row[12 + g] = p.w;
    const uint4 mine = *reinterpret_cast<const uint4*>(row + 4 * g);  // same wave: LDS operations complete in order
    if (ok) *reinterpret_cast<uint4*>(feat + s * 32 + 8 * g) = mine;
}

// The same 4 x 4 transpose over the lane groups without LDS: gfx950's row / half-wave swaps.  v_permlane16_swap exchanges the odd
// rows (of 16 lanes) of its first operand with the even rows of its second, v_permlane32_swap the upper half-wave of the first
// with the lower half-wave of the second.  With (x, y) = levels (g, 4 + g) the row swap leaves the level PAIRS (0,1) (4,5) (2,3)
// (6,7) in lane groups 0..3 -- likewise (8,9) (12,13) (10,11) (14,15) from (z, w) -- and the half-wave swap of the two pairs hands
// lane group g the dwords 4g .. 4g + 3 of its sample's row: four cross-lane instructions, one 16-byte store.
__device__ __forceinline__ void store_feat_row_swap(_Float16* __restrict__ feat, unsigned long long s, int g, const uint4& p, bool ok) {
    const auto a = __builtin_amdgcn_permlane16_swap(p.x, p.y, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(p.z, p.w, false, false);
    const auto lo = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    if (ok) *reinterpret_cast<uint4*>(feat + s * 32 + 8 * g) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
}

// density_eval + store of sigma / geo / z.
template <int F, int QG, bool TRAIN = false>
__device__ __forceinline__ void density_tile(const DensityCtx<F>& cx, const float (&x)[3], float z, unsigned long long s, bool in_range,
                                             float* __restrict__ z_vals, float* __restrict__ sigmas, _Float16* __restrict__ geo,
                                             const TrainOut* tr = nullptr) {
    const int g = cx.g;
    float4_t o;
    if constexpr (TRAIN) {
        const half8_t xf = density_encode<F, QG>(cx, x);
        float4_t acc1[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            const float4_t zero = {0, 0, 0, 0};
            acc1[t] = mfma16(cx.w0[t], xf, zero);
        }
        half8_t h[kHidSteps];
        pack_hidden(acc1, h);
        o = cx.wout.apply(h);
        // fragment element q * F + f of lane group g = feature (4 q + g) * F + f: four 4-byte pieces of the sample's 64-byte row
        const int lane = lane_id();
        store_feat_row(tr->feat, s, lane, g, __builtin_bit_cast(uint4, xf), in_range, tr->tile);
        store_h32(tr->h32, s, lane, g, o[0], o[1], o[2], o[3], in_range);
        if (in_range && g == 0) { tr->x01[3 * s] = x[0]; tr->x01[3 * s + 1] = x[1]; tr->x01[3 * s + 2] = x[2]; }
    } else {
        o = density_eval<F, QG>(cx, x);
    }
    if (in_range) {
        half4_t ov;
        ov[0] = (_Float16)o[0]; ov[1] = (_Float16)o[1]; ov[2] = (_Float16)o[2]; ov[3] = (_Float16)o[3];
        if (g == 3) {
            z_vals[s] = z;
            sigmas[s] = expf(o[3]);
            ov[3] = (_Float16)1.0f;
        }
        *reinterpret_cast<half4_t*>(geo + s * 16 + 4 * g) = ov;
    }
}

// SEG = true (T % 16 == 0): the unit of work is a ray SEGMENT owned by a whole workgroup.  The ray index is
// block-uniform, so origin / direction / near / far are scalar loads held in SGPRs for the whole segment and the
// per-lane 64-bit sample -> (ray, step) division disappears.  SEG = false: generic tile loop (any T).
template <int F, bool SEG, int QG, bool TRAIN = false>
__global__ __launch_bounds__(kBlock) void k_density_uniform_v2(RayBatch rb, const _Float16* __restrict__ table, uint32_t table_bytes,
                                                               GridMeta meta, uint32_t first_hashed, uint32_t seg_tiles,
                                                               const _Float16* __restrict__ w_sigma, float* __restrict__ z_vals,
                                                               float* __restrict__ sigmas, _Float16* __restrict__ geo, TrainOut tr = TrainOut()) {
    static_assert(F == 2, "v2 is built for F = 2 (BASELINE config 2); other shapes use k_density_uniform");
    constexpr int Q = 8 / F;
    __shared__ float s_scale[kMaxLevels];
    __shared__ uint32_t s_res[kMaxLevels], s_off[kMaxLevels + 1];
    if (threadIdx.x < kMaxLevels) {
        s_scale[threadIdx.x] = meta.scale[threadIdx.x];
        s_res[threadIdx.x] = meta.res[threadIdx.x];
    }
    if (threadIdx.x <= kMaxLevels) s_off[threadIdx.x] = meta.offset[threadIdx.x];
    __syncthreads();
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;
    DensityCtx<F> cx;
    cx.g = g;
    cx.first_hashed = first_hashed;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int l = 4 * q + g;
        cx.lv.scale[q] = s_scale[l];
        cx.lv.res[q] = s_res[l];
        cx.lv.res2[q] = s_res[l] * s_res[l];
        cx.lv.boff[q] = s_off[l] * (uint32_t)(F * sizeof(_Float16));
        cx.lv.rows[q] = s_off[l + 1] - s_off[l];
    }
    // W0 fragment with permuted columns: fragment element j = q*F + f of lane group g  <-  feature (4q+g)*F + f
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) {
        const _Float16* row = w_sigma + (size_t)(16 * t + sl) * 32;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int f = 0; f < F; ++f) cx.w0[t][q * F + f] = row[(4 * q + g) * F + f];
    }
    cx.wout.load(w_sigma + kHidden * 32, lane, 1);
    cx.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(table), 0, (int)table_bytes, 0x00020000);
    __shared__ uint32_t s_tr[TRAIN ? kWavesPerBlock * kTrWords : 1];
    if constexpr (TRAIN) tr.tile = s_tr + (threadIdx.x >> 6) * kTrWords;

    const uint32_t wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t wave_count = gridDim.x * kWavesPerBlock;
    if constexpr (SEG) {
        // unit = 16 consecutive tiles (256 samples) of one ray, walked by the 4 waves of the workgroup in lock step
        // (wave w takes tiles 4j + w): the workgroup's gathers of one round are spatially adjacent, and a CU holds
        // only as many rays as it holds workgroups, so the rays' cache lines survive in L1 from round to round.
        const uint32_t tiles_per_ray = rb.T / 16;
        const uint32_t segs_per_ray = (tiles_per_ray + seg_tiles - 1) / seg_tiles;
        const uint32_t n_units = rb.N * segs_per_ray;
        const uint32_t wave_in_block = threadIdx.x >> 6;
        for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
            const uint32_t n = unit / segs_per_ray;
            const uint32_t seg = unit - n * segs_per_ray;
            const float near = rb.nears[n], range = rb.fars[n] - near;
            const float ox = rb.rays_o[3 * (size_t)n], oy = rb.rays_o[3 * (size_t)n + 1], oz = rb.rays_o[3 * (size_t)n + 2];
            const float dx = rb.rays_d[3 * (size_t)n], dy = rb.rays_d[3 * (size_t)n + 1], dz = rb.rays_d[3 * (size_t)n + 2];
            const float sample_dist = range / (float)rb.T;
            const uint32_t t_end = min(tiles_per_ray, (seg + 1) * seg_tiles);
            for (uint32_t tt = seg * seg_tiles + wave_in_block; tt < t_end; tt += kWavesPerBlock) {
                const uint32_t i = tt * 16 + sl;
                const unsigned long long s = (unsigned long long)n * rb.T + i;
                float z = near + range * rb.lin[i];
                if (rb.noise) z = z + (rb.noise[s] - 0.5f) * sample_dist;
                float x[3];
                x[0] = (fminf(fmaxf(ox + dx * z, rb.lo[0]), rb.hi[0]) + rb.bound) * rb.inv_extent;
                x[1] = (fminf(fmaxf(oy + dy * z, rb.lo[1]), rb.hi[1]) + rb.bound) * rb.inv_extent;
                x[2] = (fminf(fmaxf(oz + dz * z, rb.lo[2]), rb.hi[2]) + rb.bound) * rb.inv_extent;
                density_tile<F, QG, TRAIN>(cx, x, z, s, true, z_vals, sigmas, geo, &tr);
            }
        }
    } else {
        const unsigned long long total = (unsigned long long)rb.N * rb.T;
        const unsigned long long n_tiles = (total + 15) / 16;
        for (unsigned long long tile = wave_global; tile < n_tiles; tile += wave_count) {
            const unsigned long long s_raw = tile * 16 + sl;
            const bool in_range = s_raw < total;
            const unsigned long long s = in_range ? s_raw : total - 1;
            const uint32_t n = (uint32_t)(s / rb.T), i = (uint32_t)(s - (unsigned long long)n * rb.T);
            const float near = rb.nears[n], range = rb.fars[n] - near;
            float z = near + range * rb.lin[i];
            if (rb.noise) z = z + (rb.noise[s] - 0.5f) * (range / (float)rb.T);
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float p = rb.rays_o[3 * (size_t)n + k] + rb.rays_d[3 * (size_t)n + k] * z;
                p = fminf(fmaxf(p, rb.lo[k]), rb.hi[k]);
                x[k] = (p + rb.bound) * rb.inv_extent;
            }
            density_tile<F, QG, TRAIN>(cx, x, z, s, in_range, z_vals, sigmas, geo, &tr);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Density, level-sliced formulation (two launches; same arithmetic, bit-identical results).
//
// Why: with a 23 MB table every XCD's 4 MiB L2 sees all 16 levels; when consecutive samples of a ray are several
// finest-level cells apart (camera rays through the whole box) the fine levels have no reuse along the ray and most
// of their gathers miss L2 (measured: 3.8 GB of L2 fills per launch for 1.85 GB of algorithmic bytes).  Here the
// LEVELS are partitioned over the XCDs instead: workgroups are dealt round-robin over the 8 XCDs, so all blocks with
// equal (blockIdx.x & 7) share one L2; such a block group encodes only two levels (slice 2g: {g, g+12}, slice 2g+1:
// {g+4, g+8}; at most ~4 MB of table per L2) for ALL samples and writes the two encoded half2 per sample to
// its scratch plane [slice][M] of uint2 (64 B/sample over the 8 planes, written and read once, full lines).  A second
// streaming kernel turns 32 features -> sigma MLP -> sigma / geo; its lane group g reads planes 2g and 2g+1, i.e.
// levels {g, g+4, g+8, g+12} -- the assignment of the fused kernel, so the MFMA sums in the same order (bit-identical).  If the dispatcher placed blocks differently the result is the same,
// only slower.
template <int F>
struct SliceLevel {
    float scale;
    uint32_t res, res2, boff, rows;
    bool hashed;
};

template <int F>
__device__ __forceinline__ void slice_issue(const SliceLevel<F>& lv, __amdgpu_buffer_rsrc_t rsrc, const float (&x)[3], float (&frac)[3],
                                            uint32_t (&raw)[8]) {
    uint32_t c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = fmaf(lv.scale, x[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        c[d] = (uint32_t)(int32_t)fl;
    }
    uint32_t idx[8];
    if (!lv.hashed) {  // block-uniform
        const uint32_t b00 = c[0] + c[1] * lv.res + c[2] * lv.res2;
        const uint32_t b10 = b00 + lv.res, b01 = b00 + lv.res2, b11 = b10 + lv.res2;
        const uint32_t base[4] = {b00, b10, b01, b11};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t v = base[k >> 1] + (uint32_t)(k & 1);
            idx[k] = v >= lv.rows ? v - lv.rows : v;
        }
    } else {
        const uint32_t hy0 = c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = c[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        const uint32_t mask = lv.rows - 1u;
#pragma unroll
        for (int k = 0; k < 8; ++k) idx[k] = ((c[0] + (uint32_t)(k & 1)) ^ yz[k >> 1]) & mask;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = gather_raw<F>(rsrc, lv.boff + idx[k] * (uint32_t)(F * sizeof(_Float16)));
}

__device__ __forceinline__ uint32_t slice_blend(const float (&frac)[3], const uint32_t (&raw)[8]) {
    const float fx = frac[0], fy = frac[1], fz = frac[2];
    const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy}, wz[2] = {1.0f - fz, fz};
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
        fma_entry(w, raw[k], a0, a1);
    }
    h2_t o;
    o[0] = (_Float16)a0;
    o[1] = (_Float16)a1;
    return __builtin_bit_cast(uint32_t, o);
}

// pass A: grid = 8 * blocks_per_slice; TWO lanes per sample (lane = 2 * sample + x-bit), 32 samples per wave.  A lane gathers the four corners
// with its x-bit of both levels, so every gather instruction fetches BOTH x-neighbours of 32 samples: they are adjacent table
// entries on dense levels and for even cells on hashed ones, i.e. one L1 line look-up instead of two -- and the look-ups (one
// line per clock and CU) are what bounds this pass.  The partner's four values of "its" level arrive by a quad swap (DPP);
// the even lane blends the slice's first level, the odd lane the second, each with all eight corners in the specification's
// order: bit-identical features.
template <int F>
__device__ __forceinline__ void slice_issue_half(const SliceLevel<F>& lv, __amdgpu_buffer_rsrc_t rsrc, const float (&x)[3], uint32_t xb,
                                                 float (&frac)[3], uint32_t (&raw)[4]) {
    uint32_t c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = fmaf(lv.scale, x[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        c[d] = (uint32_t)(int32_t)fl;
    }
    uint32_t idx[4];
    if (!lv.hashed) {  // block-uniform
        const uint32_t b00 = c[0] + c[1] * lv.res + c[2] * lv.res2;
        const uint32_t base[4] = {b00, b00 + lv.res, b00 + lv.res2, b00 + lv.res + lv.res2};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t v = base[p] + xb;
            idx[p] = v >= lv.rows ? v - lv.rows : v;
        }
    } else {
        const uint32_t hy0 = c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = c[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        const uint32_t mask = lv.rows - 1u;
#pragma unroll
        for (int p = 0; p < 4; ++p) idx[p] = ((c[0] + xb) ^ yz[p]) & mask;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) raw[p] = gather_raw<F>(rsrc, lv.boff + idx[p] * (uint32_t)(F * sizeof(_Float16)));
}

__device__ __forceinline__ uint32_t quad_swap(uint32_t v) {  // value of the neighbouring lane (lane ^ 1); every lane has a source
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// Which workgroups encode which units of which slice.  Workgroup b belongs to group b % 8 (workgroups are dealt to the XCDs
// round-robin: a group = the workgroups of one XCD); a group works through up to kPlanItems items = (slice, range of units), its own
// slice first.  The default plan gives every group exactly its slice; slice_plan() moves the tail of the heavy slices to the
// groups of the light ones (the slices of a camera batch take 0.19 ... 0.37 ms when run alone, and the launch lasts as long as
// the slowest group).  Which workgroup encodes a unit does not change what is written.
constexpr int kPlanItems = 3;
struct SlicePlan {
    uint32_t n[8];
    uint32_t slice[8][kPlanItems], begin[8][kPlanItems], end[8][kPlanItems];
};

template <int F, bool UNIFORM_RAY>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_encode_sliced_pairs(RayBatch rb, const _Float16* __restrict__ table, uint32_t table_bytes,
                                                                GridMeta meta, uint32_t L, uint32_t first_hashed, uint32_t M,
                                                                float* __restrict__ z_vals, uint2* __restrict__ feat, SlicePlan plan,
                                                                float* __restrict__ x01 = nullptr) {
    static_assert(F == 2, "F = 2 only");
    const uint32_t group = blockIdx.x & 7u, sb = blockIdx.x >> 3, n_sb = gridDim.x >> 3;
    for (uint32_t item = 0; item < plan.n[group]; ++item) {
    const uint32_t slice = plan.slice[group][item], unit_begin = plan.begin[group][item], unit_end = plan.end[group][item];
    const uint32_t grp = slice >> 1;
    const uint32_t lvl[2] = {(slice & 1u) ? grp + 4u : grp, (slice & 1u) ? grp + 8u : grp + 12u};
    SliceLevel<F> lv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const uint32_t l = lvl[a];
        lv[a].scale = meta.scale[l];
        lv[a].res = meta.res[l];
        lv[a].res2 = meta.res[l] * meta.res[l];
        lv[a].boff = meta.offset[l] * (uint32_t)(F * sizeof(_Float16));
        lv[a].rows = meta.offset[l + 1] - meta.offset[l];
        lv[a].hashed = l >= first_hashed;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(table), 0, (int)table_bytes, 0x00020000);
    const int lane = lane_id();
    const uint32_t xb = (uint32_t)(lane & 1), half_lane = (uint32_t)(lane >> 1);
    const uint32_t n_units = unit_end;
    // wave-uniform bookkeeping in SGPRs: the pass is bound by VALU issue as much as by the L1 look-ups (rocprofv3: VALU active
    // 81 % of the SIMD cycles), so the unit / ray cursor must not cost vector instructions
    const uint32_t wave = __builtin_amdgcn_readfirstlane(unit_begin + sb * kWavesPerBlock + (threadIdx.x >> 6)), wave_count = n_sb * kWavesPerBlock;
    if (wave >= n_units) continue;
    const uint32_t T = rb.T;
    struct Unit {
        uint32_t s, n;
        bool in_range;
        float near, far, lin, noise, o[3], d[3];
    };
    // UNIFORM_RAY (T % 32 == 0): a unit lies inside one ray; (ray, first sample) advance by a fixed stride without a division
    uint32_t ray = 0, first = 0, d_ray = 0, d_first = 0;
    if constexpr (UNIFORM_RAY) {
        ray = (wave * 32u) / T;
        first = wave * 32u - ray * T;
        d_ray = (wave_count * 32u) / T;
        d_first = wave_count * 32u - d_ray * T;
    }
    auto fetch = [&](uint32_t unit, uint32_t n_u, uint32_t i_u) {
        Unit u;
        const uint32_t s_raw = unit * 32u + half_lane;
        uint32_t i;
        if constexpr (UNIFORM_RAY) {
            u.in_range = true;  // M % 32 == 0
            u.s = s_raw;
            u.n = n_u;
            i = i_u + half_lane;
        } else {
            u.in_range = s_raw < M;
            u.s = u.in_range ? s_raw : M - 1u;
            u.n = u.s / T;
            i = u.s - u.n * T;
        }
        u.near = rb.nears[u.n];
        u.far = rb.fars[u.n];
        u.lin = rb.lin[i];
        u.noise = rb.noise ? rb.noise[u.s] : 0.5f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u.o[k] = rb.rays_o[3 * (size_t)u.n + k];
            u.d[k] = rb.rays_d[3 * (size_t)u.n + k];
        }
        return u;
    };
    Unit cur = fetch(wave, ray, first);
    for (uint32_t unit = wave; unit < n_units; unit += wave_count) {
        const bool more = unit + wave_count < n_units;
        uint32_t ray_n = ray, first_n = first;
        if constexpr (UNIFORM_RAY) {
            if (more) {
                ray_n = ray + d_ray;
                first_n = first + d_first;
                if (first_n >= T) {
                    first_n -= T;
                    ray_n += 1u;
                }
            }
        }
        const Unit nxt = fetch(more ? unit + wave_count : unit, ray_n, first_n);
        ray = ray_n;
        first = first_n;
        const float range = cur.far - cur.near;
        float z = cur.near + range * cur.lin;
        if (rb.noise) z = z + (cur.noise - 0.5f) * (range / (float)T);
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float p = cur.o[k] + cur.d[k] * z;
            p = fminf(fmaxf(p, rb.lo[k]), rb.hi[k]);
            x[k] = (p + rb.bound) * rb.inv_extent;
        }
        float frac[2][3];
        uint32_t half_raw[2][4];
        slice_issue_half<F>(lv[0], rsrc, x, xb, frac[0], half_raw[0]);
        slice_issue_half<F>(lv[1], rsrc, x, xb, frac[1], half_raw[1]);
        // the even lane finishes level 0 of the slice, the odd lane level 1: each sends the partner the half it does not need
        uint32_t raw[8];
        float fr[3];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t own = xb ? half_raw[1][p] : half_raw[0][p];
            const uint32_t send = xb ? half_raw[0][p] : half_raw[1][p];
            const uint32_t recv = quad_swap(send);
            raw[2 * p] = xb ? recv : own;
            raw[2 * p + 1] = xb ? own : recv;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) fr[d] = xb ? frac[1][d] : frac[0][d];
        const uint32_t mine = slice_blend(fr, raw);
        const uint32_t other = quad_swap(mine);
        if (cur.in_range && xb == 0u) {
            feat[(size_t)slice * M + cur.s] = make_uint2(mine, other);
            if (slice == 0u) {
                z_vals[cur.s] = z;
                if (x01) { x01[3 * (size_t)cur.s] = x[0]; x01[3 * (size_t)cur.s + 1] = x[1]; x01[3 * (size_t)cur.s + 2] = x[2]; }
            }
        }
        cur = nxt;
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Encode pass for the OTHER 32-feature shape: L = 8 levels of F = 4 features (the reference's default grid, main_nvsf.py:45-52:
// 512 -> 32768, T = 2^19).  One LEVEL per workgroup group (= per XCD): with 4 MB per hashed level an XCD keeps gathering from the
// table that fits its L2 -- the scheme of k_hashgrid_fwd_levels8, here from the RAYS (positions formed in registers, z_vals /
// positions written by the group of level 0) and for dense or hashed levels.  Two lanes per sample (lane = 2 * sample + x-bit): a lane
// gathers the four 8-byte entries with its x-bit; the even lane blends features 0, 1 and the odd lane features 2, 3, each over
// all eight corners in the specification's order (the partner's half of every entry arrives by a quad swap): bit-identical to
// encode_level<3, 4>.
//
// Where the features go: the row of a sample is 32 fp16 = 16 column PAIRS; for F = 2 pair v is level v, for F = 4 pair v is
// features 2 (v & 1), 2 (v & 1) + 1 of level v >> 1.  The planes are laid out by pair exactly as k_encode_sliced_pairs lays them
// out by level (plane 2g = pairs {g, g + 12}, plane 2g + 1 = pairs {g + 4, g + 8}), so every consumer of the planes -- the
// streaming tails k_render_tail2 / k_render_uniform<*, true> / k_density_from_features, their permuted W0 fragments and the
// TRAIN forms' feature rows -- is the same code for both shapes.  A lane stores its own pair (4 bytes) into its plane.
__device__ __forceinline__ uint32_t pair_plane_dword(uint32_t v, uint32_t M, uint32_t s) {  // dword index of column pair v of sample s
    const uint32_t vg = v & 3u, vq = v >> 2;
    const uint32_t plane = 2u * vg + ((vq == 1u || vq == 2u) ? 1u : 0u), slot = vq >> 1;
    return (plane * M + s) * 2u + slot;  // < 2^32: M < 2^28 is required by the launcher
}

template <bool UNIFORM_RAY>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_encode_sliced_f4(
    RayBatch rb, const _Float16* __restrict__ table, uint32_t table_bytes, GridMeta meta, uint32_t first_hashed, uint32_t M,
    float* __restrict__ z_vals, uint32_t* __restrict__ feat_dw, SlicePlan plan, float* __restrict__ x01 = nullptr) {
    const uint32_t group = blockIdx.x & 7u, sb = blockIdx.x >> 3, n_sb = gridDim.x >> 3;
    for (uint32_t item = 0; item < plan.n[group]; ++item) {
    const uint32_t level = plan.slice[group][item], unit_begin = plan.begin[group][item], unit_end = plan.end[group][item];
    SliceLevel<4> lv;
    lv.scale = meta.scale[level];
    lv.res = meta.res[level];
    lv.res2 = meta.res[level] * meta.res[level];
    lv.boff = meta.offset[level] * 8u;
    lv.rows = meta.offset[level + 1] - meta.offset[level];
    lv.hashed = level >= first_hashed;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(table), 0, (int)table_bytes, 0x00020000);
    const int lane = lane_id();
    const uint32_t xb = (uint32_t)(lane & 1), half_lane = (uint32_t)(lane >> 1);
    const uint32_t pair = 2u * level + xb;  // this lane's column pair
    const uint32_t n_units = unit_end;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(unit_begin + sb * kWavesPerBlock + (threadIdx.x >> 6)), wave_count = n_sb * kWavesPerBlock;
    if (wave >= n_units) continue;
    const uint32_t T = rb.T;
    struct Unit {
        uint32_t s, n;
        bool in_range;
        float near, far, lin, noise, o[3], d[3];
    };
    uint32_t ray = 0, first = 0, d_ray = 0, d_first = 0;
    if constexpr (UNIFORM_RAY) {  // T % 32 == 0: a unit lies inside one ray; the cursor advances without a division
        ray = (wave * 32u) / T;
        first = wave * 32u - ray * T;
        d_ray = (wave_count * 32u) / T;
        d_first = wave_count * 32u - d_ray * T;
    }
    auto fetch = [&](uint32_t unit, uint32_t n_u, uint32_t i_u) {
        Unit u;
        const uint32_t s_raw = unit * 32u + half_lane;
        uint32_t i;
        if constexpr (UNIFORM_RAY) {
            u.in_range = true;
            u.s = s_raw;
            u.n = n_u;
            i = i_u + half_lane;
        } else {
            u.in_range = s_raw < M;
            u.s = u.in_range ? s_raw : M - 1u;
            u.n = u.s / T;
            i = u.s - u.n * T;
        }
        u.near = rb.nears[u.n];
        u.far = rb.fars[u.n];
        u.lin = rb.lin[i];
        u.noise = rb.noise ? rb.noise[u.s] : 0.5f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u.o[k] = rb.rays_o[3 * (size_t)u.n + k];
            u.d[k] = rb.rays_d[3 * (size_t)u.n + k];
        }
        return u;
    };
    Unit cur = fetch(wave, ray, first);
    for (uint32_t unit = wave; unit < n_units; unit += wave_count) {
        const bool more = unit + wave_count < n_units;
        uint32_t ray_n = ray, first_n = first;
        if constexpr (UNIFORM_RAY) {
            if (more) {
                ray_n = ray + d_ray;
                first_n = first + d_first;
                if (first_n >= T) {
                    first_n -= T;
                    ray_n += 1u;
                }
            }
        }
        const Unit nxt = fetch(more ? unit + wave_count : unit, ray_n, first_n);
        ray = ray_n;
        first = first_n;
        const float range = cur.far - cur.near;
        float z = cur.near + range * cur.lin;
        if (rb.noise) z = z + (cur.noise - 0.5f) * (range / (float)T);
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float p = cur.o[k] + cur.d[k] * z;
            p = fminf(fmaxf(p, rb.lo[k]), rb.hi[k]);
            x[k] = (p + rb.bound) * rb.inv_extent;
        }
        float frac[3];
        uint32_t c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pos = fmaf(lv.scale, x[d], 0.5f);
            const float fl = floorf(pos);
            frac[d] = pos - fl;
            c[d] = (uint32_t)(int32_t)fl;
        }
        uint32_t idx[4];
        if (!lv.hashed) {  // block-uniform
            const uint32_t b00 = c[0] + c[1] * lv.res + c[2] * lv.res2;
            const uint32_t base[4] = {b00, b00 + lv.res, b00 + lv.res2, b00 + lv.res + lv.res2};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t v = base[p] + xb;
                idx[p] = v >= lv.rows ? v - lv.rows : v;
            }
        } else {
            const uint32_t hy0 = c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
            const uint32_t hz0 = c[2] * 805459861u, hz1 = hz0 + 805459861u;
            const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
            const uint32_t mask = lv.rows - 1u;
#pragma unroll
            for (int p = 0; p < 4; ++p) idx[p] = ((c[0] + xb) ^ yz[p]) & mask;
        }
        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
        u2v raw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) raw[p] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lv.boff + (idx[p] << 3), 0, 0);
        // this lane's feature pair of all eight corners: its own entries' dword, and the partner's (the corners with the other x-bit)
        uint32_t mine[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t own = xb ? raw[p][1] : raw[p][0], send = xb ? raw[p][0] : raw[p][1];
            const uint32_t recv = quad_swap(send);
            mine[2 * p] = xb ? recv : own;
            mine[2 * p + 1] = xb ? own : recv;
        }
        const uint32_t packed = slice_blend(frac, mine);
        if (cur.in_range) {
            feat_dw[pair_plane_dword(pair, M, cur.s)] = packed;
            if (level == 0u && xb == 0u) {
                z_vals[cur.s] = z;
                if (x01) { x01[3 * (size_t)cur.s] = x[0]; x01[3 * (size_t)cur.s + 1] = x[1]; x01[3 * (size_t)cur.s + 2] = x[2]; }
            }
        }
        cur = nxt;
    }
    }
}

// Static balance of the encode pass (host).  Cost of a slice relative to the VALU work of streaming every sample through one XCD
// (= 1; 0.16 ms for 3.1 M samples).  With c = res_l / T = the number of cells a step crosses at level l on a ray that spans the
// box, the finer level of the slice costs fill(c) = 1 + 0.72 clamp((c - 0.3) / 0.72, 0, 1) + 0.25 max(0, c - 1): from c ~ 1 on
// every look-up misses L1 and the CU moves a 128-byte line per look-up (levels 9 ... 15 alone at T = 768: 1.07 1.21 1.47 1.72 1.77
// 1.85 2.14 measured, 1.08 1.23 1.43 1.72 1.82 1.95 2.14 by this formula); the coarser level adds 0.16, a hashed one 0.16 + 1.6 c
// (levels 5, 6, 7 as partners: 0.31 0.39 0.47 measured).  Slices above the mean give the tail of their units to the groups of the
// slices below it, at most kPlanItems - 1 foreign items per group; a foreign item is priced 5 % higher (its levels are not in that
// XCD's L2; 1.0 ... 1.3 measure within 1 %).  A model, not a measurement: it only decides who encodes what, never what is written.
static SlicePlan slice_plan(uint32_t n_units, uint32_t T, const uint32_t* h_res, uint32_t first_hashed, bool balance) {
    SlicePlan plan = {};
    for (uint32_t g = 0; g < 8; ++g) {
        plan.n[g] = 1;
        plan.slice[g][0] = g;
        plan.begin[g][0] = 0;
        plan.end[g][0] = n_units;
    }
    if (!balance || n_units < 8192u) return plan;
    double cost[8], mean = 0.0;
    for (uint32_t p = 0; p < 8; ++p) {
        const uint32_t grp = p >> 1, la = (p & 1u) ? grp + 4u : grp, lb = (p & 1u) ? grp + 8u : grp + 12u;
        const double c = (double)h_res[lb] / (double)T, ca = (double)h_res[la] / (double)T;
        const double ramp = c <= 0.3 ? 0.0 : (c >= 1.02 ? 1.0 : (c - 0.3) / 0.72);
        const double fill = 1.0 + 0.72 * ramp + 0.25 * (c > 1.0 ? c - 1.0 : 0.0);
        cost[p] = fill + (la >= first_hashed ? 0.16 + 1.6 * ca : 0.16);
        mean += cost[p] / 8.0;
    }
    const double penalty = 1.05;
    double spare[8];
    for (uint32_t p = 0; p < 8; ++p) spare[p] = mean - cost[p];  // > 0: capacity of group p, < 0: excess of slice p
    for (int round = 0; round < 8; ++round) {
        int donor = -1, taker = -1;
        for (int p = 0; p < 8; ++p) {
            if (spare[p] < -0.02 && (donor < 0 || spare[p] < spare[donor])) donor = p;
            if (spare[p] > 0.02 && plan.n[p] < (uint32_t)kPlanItems && (taker < 0 || spare[p] > spare[taker])) taker = p;
        }
        if (donor < 0 || taker < 0) break;
        const double moved = (-spare[donor] < spare[taker] / penalty) ? -spare[donor] : spare[taker] / penalty;  // in donor cost units
        uint32_t units = (uint32_t)((double)n_units * moved / cost[donor]);
        units &= ~63u;  // whole rounds of a workgroup's waves
        const uint32_t have = plan.end[donor][0] - plan.begin[donor][0];
        if (units == 0u || units + 4096u > have) break;
        const uint32_t k = plan.n[taker]++;
        plan.slice[taker][k] = (uint32_t)donor;
        plan.end[taker][k] = plan.end[donor][0];
        plan.end[donor][0] -= units;
        plan.begin[taker][k] = plan.end[donor][0];
        spare[donor] += moved;
        spare[taker] -= moved * penalty;
    }
    return plan;
}

// pass B: 32 encoded features per sample (scratch planes) -> sigma MLP -> sigma, geo.
template <int F, bool TRAIN = false>
__global__ __launch_bounds__(kBlock) void k_density_from_features(const uint2* __restrict__ feat, uint32_t M, uint32_t L,
                                                                  const _Float16* __restrict__ w_sigma, float* __restrict__ sigmas,
                                                                  _Float16* __restrict__ geo, TrainOut tr = TrainOut()) {
    static_assert(F == 2, "F = 2 only");
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;
    const uint32_t lv4[4] = {(uint32_t)g, (uint32_t)g + 4u, (uint32_t)g + 8u, (uint32_t)g + 12u};  // as in k_density_uniform_v2
    half8_t w0[kHidTiles];
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t) {
        const _Float16* row = w_sigma + (size_t)(16 * t + sl) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < F; ++f) w0[t][i * F + f] = row[lv4[i] * F + f];
    }
    OutLayerW wout;
    wout.load(w_sigma + kHidden * 32, lane, 1);
    __shared__ uint32_t s_tr[TRAIN ? kWavesPerBlock * kTrWords : 1];
    const uint2* plane0 = feat + (size_t)(2 * g) * M;
    const uint2* plane1 = feat + (size_t)(2 * g + 1) * M;
    const uint32_t n_tiles = (M + 15u) / 16u;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), wave_count = gridDim.x * kWavesPerBlock;
    constexpr int U = 4;  // tiles per iteration (independent load / MFMA chains)
    for (uint32_t tile0 = wave * U; tile0 < n_tiles; tile0 += wave_count * U) {
        uint2 w[U][2];
        uint32_t s[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t s_raw = (tile0 + u) * 16u + (uint32_t)sl;
            ok[u] = s_raw < M;
            s[u] = ok[u] ? s_raw : M - 1u;
            w[u][0] = plane0[s[u]];
            w[u][1] = plane1[s[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
            const u4_t packed = {w[u][0].x, w[u][1].x, w[u][1].y, w[u][0].y};
            const half8_t xf = __builtin_bit_cast(half8_t, packed);
            float4_t acc1[kHidTiles];
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) {
                const float4_t zero = {0, 0, 0, 0};
                acc1[t] = mfma16(w0[t], xf, zero);
            }
            half8_t h[kHidSteps];
            pack_hidden(acc1, h);
            const float4_t o = wout.apply(h);
            if constexpr (TRAIN) {  // feature rows in level order (fragment pieces = levels g, g+4, g+8, g+12) and the fp32 outputs
                store_feat_row(tr.feat, s[u], lane, g, make_uint4(packed.x, packed.y, packed.z, packed.w), ok[u], s_tr + (threadIdx.x >> 6) * kTrWords);
                store_h32(tr.h32, s[u], lane, g, o[0], o[1], o[2], o[3], ok[u]);
            }
            if (ok[u]) {
                half4_t ov;
                ov[0] = (_Float16)o[0]; ov[1] = (_Float16)o[1]; ov[2] = (_Float16)o[2]; ov[3] = (_Float16)o[3];
                if (g == 3) {
                    sigmas[s[u]] = expf(o[3]);
                    ov[3] = (_Float16)1.0f;
                }
                *reinterpret_cast<half4_t*>(geo + (size_t)s[u] * 16 + 4 * g) = ov;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Heads + image accumulation, one wave per ray.
//   LIDAR = false : colour net  [SH16(d) | geo15 | 1] (32)  -> 64 -> 64 -> 3 (padded 16)
//   LIDAR = true  : raydrop net and intensity net  [Freq72(d) | geo15 | 1 x 9] (96) -> 64 -> 64 -> 1;
//                   image channels = (raydrop, intensity)  (network_dynamic.py:317)
// rgb = sigmoid(fp32 logits); samples with weight <= w_thresh contribute 0 (renderer_dynamic.py:202,
// network_dynamic.py:297-307, 325-330); image = sum_i w_i rgb_i (+ (1 - ws) * bg for the camera, :236-237).
// One head.  Only the LAST 32-wide k-step of the first layer varies along a ray (it holds the geometry
// features); the leading k-steps hold the direction encoding, which is constant per ray, so their partial
// products are computed once per ray (`ray_const`) and re-used as the MFMA's C operand for every tile.  The
// accumulation chain (k-steps in order 0,1,2, fp32) is unchanged, hence the result is bit-identical to the
// unfactored evaluation while 8 of the 12 first-layer MFMAs of a LiDAR head disappear from the tile loop.
template <int IN_STEPS>
struct HeadW {
    half8_t w0_last[kHidTiles];      // first layer, last k-step
    float4_t ray_const[kHidTiles];   // first layer, k-steps 0..IN_STEPS-2 applied to the ray-constant features
    HiddenLayerW w1;
    OutLayerW w2;
    __device__ __forceinline__ void load(const _Float16* __restrict__ W, int lane, const half8_t (&xf_const)[IN_STEPS]) {
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < IN_STEPS - 1; ++s) c = mfma16(load_w_natural(W, 32 * IN_STEPS, t, s, lane), xf_const[s], c);
            ray_const[t] = c;
            w0_last[t] = load_w_natural(W, 32 * IN_STEPS, t, IN_STEPS - 1, lane);
        }
        w1.load(W + kHidden * 32 * IN_STEPS, lane);
        w2.load(W + kHidden * 32 * IN_STEPS + kHidden * kHidden, lane);
    }
    __device__ __forceinline__ float4_t apply(const half8_t& x_last) const {
        float4_t acc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) acc[t] = mfma16(w0_last[t], x_last, ray_const[t]);
        half8_t h[kHidSteps];
        pack_hidden(acc, h);
        w1.apply(h, acc);
        pack_hidden(acc, h);
        return w2.apply(h);
    }
};

// sigmoid with the hardware exp2 / reciprocal (each ~1 ulp): well inside the 1e-4 budget of the composited image
__device__ __forceinline__ float sigmoid_f32(float logit) { return __builtin_amdgcn_rcpf(1.0f + __expf(-logit)); }

// Two 16-sample tiles per iteration (independent MFMA chains for the scheduler to interleave); the weights and
// geometry rows of the next iteration are fetched before the current one is evaluated.
template <bool LIDAR>
__global__ __launch_bounds__(kBlock) void k_heads_uniform(const float* __restrict__ weights, const _Float16* __restrict__ geo,
                                                          const float* __restrict__ rays_d, const float* __restrict__ weights_sum,
                                                          const _Float16* __restrict__ w_a, const _Float16* __restrict__ w_b,
                                                          uint32_t N, uint32_t T, float w_thresh, float bg0, float bg1, float bg2,
                                                          int use_bg, float* __restrict__ image) {
    constexpr int IN_STEPS = LIDAR ? 3 : 1;
    constexpr int C = LIDAR ? 1 : 3;   // channels accumulated by this wave
    constexpr int CI = LIDAR ? 2 : 3;  // channels of the image
    const uint32_t n = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;

    // per-ray direction encoding -> the ray-constant part of the B fragments
    const float d0 = (rays_d[3 * (size_t)n] + 1.0f) / 2.0f, d1 = (rays_d[3 * (size_t)n + 1] + 1.0f) / 2.0f,
                d2 = (rays_d[3 * (size_t)n + 2] + 1.0f) / 2.0f;  // network_dynamic.py:310,319
    half8_t xf[IN_STEPS];
    if constexpr (!LIDAR) {
        float sh[16];
        sh4_basis(d0, d1, d2, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[0][j] = (_Float16)(g == 0 ? sh[j] : sh[8 + j]);  // groups 2,3 are overwritten per tile
    } else {
        // feature k = i*24 + 2*f + (0: sin, 1: cos), k < 72
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int k = 32 * s + 8 * g + j;
                float sn = 1.0f, cs = 1.0f;  // k >= 72: placeholder (geo / ones filled per tile)
                if (k < 72) {
                    const int i = k / 24, f = (k - 24 * i) >> 1;
                    freq_pair(i == 0 ? d0 : (i == 1 ? d1 : d2), f, sn, cs);
                }
                xf[s][j] = (_Float16)sn;
                xf[s][j + 1] = (_Float16)cs;
            }
    }
    // LiDAR: the two heads run as separate waves (blockIdx.y selects raydrop / intensity -> image channel 0 / 1):
    // half the weight registers per wave (twice the resident waves) and twice as many waves to fill the chip.
    const int head = LIDAR ? (int)blockIdx.y : 0;
    HeadW<IN_STEPS> net_a;
    net_a.load(head == 0 ? w_a : w_b, lane, xf);
    const half8_t x_base = xf[IN_STEPS - 1];  // ray-constant lanes of the varying k-step
    // lane groups that carry geometry features in the varying k-step, and which half of the 16-wide geo row they take
    const bool takes_geo = LIDAR ? (g == 1 || g == 2) : (g >= 2);
    const int geo_half = LIDAR ? (g == 2 ? 8 : 0) : (g == 3 ? 8 : 0);

    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    const float* w_row = weights + (size_t)n * T;
    const _Float16* geo_row = geo + (size_t)n * T * 16 + geo_half;

    auto fetch = [&](uint32_t base, float (&w)[2], half8_t (&gv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t i = base + 16 * u + sl;
            w[u] = i < T ? w_row[i] : 0.0f;
            gv[u] = x_base;
            if (takes_geo && i < T) gv[u] = *reinterpret_cast<const half8_t*>(geo_row + (size_t)i * 16);
        }
    };
    // software pipeline: the loads of the next iteration are in flight while the current one is evaluated
    // (a second stage was measured: no gain for the camera head, one wave per SIMD fewer for the LiDAR heads)
    float w_cur[2], w_nxt[2];
    half8_t g_cur[2], g_nxt[2];
    fetch(0, w_cur, g_cur);
    for (uint32_t base = 0; base < T; base += 32) {
        fetch(base + 32, w_nxt, g_nxt);  // out-of-range tiles read nothing and carry w = 0
        const bool on0 = w_cur[0] > w_thresh, on1 = w_cur[1] > w_thresh;
        if (__ballot(on0 || on1)) {
            // both tiles in one basic block: two independent MFMA chains for the scheduler to interleave
            const float4_t oa0 = net_a.apply(g_cur[0]);
            const float4_t oa1 = net_a.apply(g_cur[1]);
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (on0) acc[c] += w_cur[0] * sigmoid_f32(oa0[c]);
                    if (on1) acc[c] += w_cur[1] * sigmoid_f32(oa1[c]);
                }
            }
        }
        w_cur[0] = w_nxt[0]; w_cur[1] = w_nxt[1];
        g_cur[0] = g_nxt[0]; g_cur[1] = g_nxt[1];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = wave_sum(acc[c]);  // only lanes 0..15 hold non-zero partial sums
    if (lane == 0) {
        const float bg[3] = {bg0, bg1, bg2};
        const float rest = use_bg ? 1.0f - weights_sum[n] : 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) image[(size_t)n * CI + head + c] = use_bg ? acc[c] + rest * bg[c] : acc[c];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused occupancy-grid render, evaluation mode (BASELINE config 3).  The reference's protocol is a host loop over
// the surviving rays (march_rays -> field -> composite_rays, a device->host sync per iteration, raymarching.py:
// 389-409 / 480-493); here ONE launch renders the batch: a wave owns a ray, marches it through the occupancy
// bit field 16 samples at a time (all lanes execute the serial march redundantly and lane l keeps sample l & 15,
// so no cross-lane traffic is needed to form the MFMA tile), evaluates hash grid -> sigma MLP -> heads on the tile
// with the fragments of the uniform kernels, composites the 16 samples in order and stops at the first sample whose
// incoming transmittance is below T_thresh, at the far plane, or after max_steps samples.
struct OccRays {
    const float* rays_o;
    const float* rays_d;
    const float* nears;
    const float* fars;
    const uint8_t* grid;
    float bound, dt_gamma;
    uint32_t max_steps, C, H, N;
};

__device__ __forceinline__ float readlane_f32(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    h2_t v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}


// Every MLP weight fragment lives in LDS in MFMA-operand order ([fragment][lane] x 16 B, conflict-free ds_read_b128) instead of
// registers.  The kernel is latency-bound (a serial march per ray), so what counts is how many rays are resident: with the
// fragments in registers it needs 226 / 256 VGPRs (2 / 1 waves per SIMD), with LDS fragments ~128 (4 per SIMD, all 4096 rays of
// a batch in flight at once).
template <bool LIDAR>
struct OccFrags {
    static constexpr int IN_STEPS = LIDAR ? 3 : 1;
    static constexpr int kHeads = LIDAR ? 2 : 1;
    static constexpr int kSigma = 0;                           // 4 x W0 (permuted columns), 2 x W_out (rows rotated by one)
    static constexpr int kHead = 6;                            // per head: the LAST k-step of the first layer (4), 8 hidden, 2 output
    static constexpr int kPerHead = 4 + 8 + 2;
    static constexpr int kCount = kHead + kHeads * kPerHead;
    static constexpr int kPre = kHeads * 16;                   // float4 per wave: the per-ray part of the first layer (ray_head_constants)
};

// F: features per level of the grid whose encode fills the sigma net's B fragment IN THIS KERNEL (density_encode<F>: fragment
// element q F + f of lane group g = feature (4 q + g) F + f); the kernels that read feature planes use F = 2 (column pairs).
template <bool LIDAR, int F = 2>
__device__ __forceinline__ half8_t occ_fragment(int f, int lane, const _Float16* __restrict__ w_sigma, const _Float16* __restrict__ w_a,
                                                const _Float16* __restrict__ w_b) {
    using FR = OccFrags<LIDAR>;
    constexpr int IN_STEPS = FR::IN_STEPS;
    const int g = lane >> 4, sl = lane & 15;
    if (f < 4) {
        const _Float16* row = w_sigma + (size_t)(16 * f + sl) * 32;
        half8_t v;
        if constexpr (F == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[2 * q] = row[(4 * q + g) * 2];
                v[2 * q + 1] = row[(4 * q + g) * 2 + 1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8 / F; ++q)
#pragma unroll
                for (int e = 0; e < F; ++e) v[q * F + e] = row[(4 * q + g) * F + e];
        }
        return v;
    }
    if (f < 6) return load_w_chained(w_sigma + kHidden * 32, 0, f - 4, lane, 1);
    int r = f - FR::kHead;
    const _Float16* W = w_a;
    if (r >= FR::kPerHead) { r -= FR::kPerHead; W = w_b; }
    if (r < 4) return load_w_natural(W, 32 * IN_STEPS, r, IN_STEPS - 1, lane);
    r -= 4;
    if (r < 8) return load_w_chained(W + kHidden * 32 * IN_STEPS, r >> 1, r & 1, lane);
    return load_w_chained(W + kHidden * 32 * IN_STEPS + kHidden * kHidden, 0, r - 8, lane);
}

// The first IN_STEPS - 1 k-steps of a head's first layer see the encoded ray direction only (LiDAR: 64 of the 96 inputs):
// every sample of the ray adds the same 64 partial sums.  They are formed ONCE per ray by the same MFMA chain the per-tile
// form runs (step 0, then step 1 on its result; all 16 columns equal), parked as 64 floats per head in LDS, and enter the
// per-tile MFMA of the last step as its accumulator input: the same additions in the same order, bit-identical, with a third
// of the first layer's MFMAs, a third of its weight fragments in LDS (50 -> 34 KB for the LiDAR field) and 8 VGPRs less.
template <bool LIDAR>
__device__ __forceinline__ void ray_head_constants(const half8_t* xf, const _Float16* __restrict__ w_a, const _Float16* __restrict__ w_b, int lane,
                                                   float4_t* pre) {
    using FR = OccFrags<LIDAR>;
    constexpr int IN_STEPS = FR::IN_STEPS;
    if constexpr (IN_STEPS > 1) {
#pragma unroll
        for (int h = 0; h < FR::kHeads; ++h) {
            const _Float16* W = h ? w_b : w_a;
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) {
                float4_t c = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < IN_STEPS - 1; ++s) c = mfma16(load_w_natural(W, 32 * IN_STEPS, t, s, lane), xf[s], c);
                if ((lane & 15) == 0) pre[(h * 4 + t) * 4 + (lane >> 4)] = c;
            }
        }
    }
}

template <bool LIDAR, int F = 2, int FH = -1>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_render_occupancy_lds(OccRays rr, const _Float16* __restrict__ table, uint32_t table_bytes,
                                                                 GridMeta meta, uint32_t first_hashed, const _Float16* __restrict__ w_sigma,
                                                                 const _Float16* __restrict__ w_a, const _Float16* __restrict__ w_b,
                                                                 float density_scale, float T_thresh, float bg0, float bg1, float bg2,
                                                                 float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                 float* __restrict__ image) {
    using FR = OccFrags<LIDAR>;
    constexpr int IN_STEPS = FR::IN_STEPS;
    __shared__ uint4 s_lv[kMaxLevels];
    __shared__ half8_t s_frag[FR::kCount * kWave];
    __shared__ float4_t s_pre[kWavesPerBlock * FR::kPre];
    constexpr uint32_t kLutH = 128;  // Morton bit-spread table for grids up to 128^3 (the reference's size); larger: computed
    __shared__ uint32_t s_lut[kLutH];
    if (threadIdx.x < kMaxLevels) {
        const uint32_t l = threadIdx.x;
        s_lv[l] = make_uint4(__builtin_bit_cast(uint32_t, meta.scale[l]), meta.res[l], meta.offset[l] * (uint32_t)(F * sizeof(_Float16)),
                             meta.offset[l + 1] - meta.offset[l]);
    }
    if (threadIdx.x < kLutH) s_lut[threadIdx.x] = spread3(threadIdx.x);
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;
    for (int f = (int)(threadIdx.x >> 6); f < FR::kCount; f += kWavesPerBlock) s_frag[f * kWave + lane] = occ_fragment<LIDAR, F>(f, lane, w_sigma, w_a, w_b);
    __syncthreads();
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (n >= rr.N) return;
    const half8_t* frag = s_frag + lane;

    DensityCtx<F> cx;  // level constants + table descriptor; the weight members stay unused (LDS fragments instead)
    cx.g = g;
    cx.first_hashed = first_hashed;
    cx.lds_lv = s_lv + g;
    cx.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(table), 0, (int)table_bytes, 0x00020000);

    const float rd0 = rr.rays_d[3 * (size_t)n], rd1 = rr.rays_d[3 * (size_t)n + 1], rd2 = rr.rays_d[3 * (size_t)n + 2];
    const float d0 = (rd0 + 1.0f) / 2.0f, d1 = (rd1 + 1.0f) / 2.0f, d2 = (rd2 + 1.0f) / 2.0f;
    half8_t xf[IN_STEPS];
    if constexpr (!LIDAR) {
        float sh[16];
        sh4_basis(d0, d1, d2, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[0][j] = (_Float16)(g == 0 ? sh[j] : sh[8 + j]);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int k = 32 * s + 8 * g + j;
                float sn = 1.0f, cs = 1.0f;
                if (k < 72) {
                    const int i = k / 24, f = (k - 24 * i) >> 1;
                    freq_pair(i == 0 ? d0 : (i == 1 ? d1 : d2), f, sn, cs);
                }
                xf[s][j] = (_Float16)sn;
                xf[s][j + 1] = (_Float16)cs;
            }
    }
    const bool takes_geo = LIDAR ? (g == 1 || g == 2) : (g >= 2);
    const int src_a = (sl + 16 * (LIDAR ? 2 * (g - 1) : 2 * (g - 2))) & 63, src_b = (src_a + 16) & 63;

    float4_t* pre = s_pre + (threadIdx.x >> 6) * FR::kPre;
    ray_head_constants<LIDAR>(xf, w_a, w_b, lane, pre);
    // one head on a tile: last k-step of the first layer on top of the ray's constants, two more layers
    auto head = [&](int hd, const half8_t& x_last) {
        const int base = FR::kHead + hd * FR::kPerHead;
        float4_t acc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
            if constexpr (IN_STEPS > 1) c = pre[(hd * 4 + t) * 4 + g];
            acc[t] = mfma16(frag[(base + t) * kWave], x_last, c);
        }
        half8_t h[kHidSteps];
        pack_hidden(acc, h);
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) c = mfma16(frag[(base + 4 + 2 * t + s) * kWave], h[s], c);
            acc[t] = c;
        }
        pack_hidden(acc, h);
        float4_t c = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s) c = mfma16(frag[(base + 4 + 8 + s) * kWave], h[s], c);
        return c;
    };

    const float o3[3] = {rr.rays_o[3 * (size_t)n], rr.rays_o[3 * (size_t)n + 1], rr.rays_o[3 * (size_t)n + 2]};
    const float dd[3] = {rd0, rd1, rd2};
    Marcher m;
    m.init(o3, dd, rr.grid, rr.bound, rr.dt_gamma, rr.max_steps, rr.C, rr.H);
    if (rr.H <= kLutH) m.use_lut(s_lut);
    const float far = rr.fars[n];
    float t = rr.nears[n];
    float last_t = t, t_comp = t;
    float ws = 0.0f, dep = 0.0f, col[3] = {0.0f, 0.0f, 0.0f};
    uint32_t total = 0;
    const float extent = 2.0f * rr.bound;
    bool alive = true;
    // Marching, 64 chain members at a time (ChainWalker, march_device.h): the parameters a ray visits form one chain
    // t_{k+1} = t_k + step_len(t_k), lane j classifies member j of the current batch on its own (ONE dependent occupancy
    // load per 64 members), and which members are visited -- hence which are samples -- is decided for the whole batch at
    // once (next_samples).  The samples of a batch are then handed to the tile one by one (bit scan over the mask).
    ChainWalker w;
    w.init(t);
    unsigned long long S = 0ull;  // samples of the current batch not yet consumed (wave-uniform)
    float bd1 = 0.0f;             // per lane: second delta (t_after - previous sample's t_after) of this lane's sample
    bool ray_done = false;
    while (alive) {
        uint32_t count = 0;
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, sdt = 0.0f, sd1 = 0.0f;
        while (count < 16u && total + count < rr.max_steps && !ray_done) {
            if (!S) {
                S = w.next_samples(m, far, lane, rr.max_steps - total - count);
                if (!S) { ray_done = true; break; }
                const float t_after = w.bt + w.bdt;
                const unsigned long long lower = S & ((1ull << lane) - 1ull);
                const int prev_lane = lower ? 63 - __builtin_clzll(lower) : lane;
                const float prev_after = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(prev_lane << 2, __builtin_bit_cast(int, t_after)));
                bd1 = t_after - (lower ? prev_after : last_t);
                last_t = readlane_f32(t_after, 63 - __builtin_clzll(S));
            }
            const int ju = __builtin_ctzll(S);
            S &= S - 1ull;
            const float xj = readlane_f32(w.bx, ju), yj = readlane_f32(w.by, ju), zj = readlane_f32(w.bz, ju);
            const float dtj = readlane_f32(w.bdt, ju), d1j = readlane_f32(bd1, ju);
            if (count == (uint32_t)sl) { sx = xj; sy = yj; sz = zj; sdt = dtj; sd1 = d1j; }
            ++count;
        }
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(count);
        if (cnt == 0u) break;
        const float x01[3] = {(sx + rr.bound) / extent, (sy + rr.bound) / extent, (sz + rr.bound) / extent};
        const half8_t feat = density_encode<F, 8 / F, true, FH>(cx, x01);
        float4_t o;
        {
            float4_t acc1[kHidTiles];
#pragma unroll
            for (int tt = 0; tt < kHidTiles; ++tt) {
                const float4_t zero = {0, 0, 0, 0};
                acc1[tt] = mfma16(frag[(FR::kSigma + tt) * kWave], feat, zero);
            }
            half8_t h[kHidSteps];
            pack_hidden(acc1, h);
            float4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) c = mfma16(frag[(FR::kSigma + 4 + s) * kWave], h[s], c);
            o = c;
        }
        const float sigma = expf(o[3]) * density_scale;  // meaningful in lanes g == 3
        const uint32_t p0 = pack_h2(o[0], o[1]), p1 = pack_h2(o[2], g == 3 ? 1.0f : o[3]);
        typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
        u4_t gv;
        gv[0] = (uint32_t)__shfl((int)p0, src_a);
        gv[1] = (uint32_t)__shfl((int)p1, src_a);
        gv[2] = (uint32_t)__shfl((int)p0, src_b);
        gv[3] = (uint32_t)__shfl((int)p1, src_b);
        const half8_t x_last = takes_geo ? __builtin_bit_cast(half8_t, gv) : xf[IN_STEPS - 1];
        float c[3] = {0.0f, 0.0f, 0.0f};  // colour of sample sl, meaningful in lanes g == 0
        const float4_t oa = head(0, x_last);
        if constexpr (LIDAR) {
            const float4_t ob = head(1, x_last);
            c[0] = sigmoid_f32(oa[0]);
            c[1] = sigmoid_f32(ob[0]);
        } else {
            c[0] = sigmoid_f32(oa[0]); c[1] = sigmoid_f32(oa[1]); c[2] = sigmoid_f32(oa[2]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if ((uint32_t)i < cnt && alive) {
                const float s_i = readlane_f32(sigma, 48 + i);
                const float dt_i = readlane_f32(sdt, i), d1_i = readlane_f32(sd1, i);
                const float alpha = 1.0f - expf(-s_i * dt_i);
                const float T = 1.0f - ws;
                const float w = alpha * T;
                ws += w;
                t_comp += d1_i;
                dep += w * t_comp;
#pragma unroll
                for (int k = 0; k < 3; ++k) col[k] += w * readlane_f32(c[k], i);
                if (T < T_thresh) alive = false;
            }
        }
        total += cnt;
        if (cnt < 16u) alive = false;
    }
    if (lane == 0) {
        weights_sum[n] = ws;
        depth[n] = dep;
        if constexpr (LIDAR) {
            image[2 * (size_t)n] = col[0];
            image[2 * (size_t)n + 1] = col[1];
        } else {
            const float rest = 1.0f - ws;
            image[3 * (size_t)n] = col[0] + rest * bg0;
            image[3 * (size_t)n + 1] = col[1] + rest * bg1;
            image[3 * (size_t)n + 2] = col[2] + rest * bg2;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole uniform render of a ray in one wave (evaluation, no_grad): samples -> hash grid -> sigma MLP -> alpha
// compositing -> masked heads -> image, 16 samples (one MFMA tile) at a time in sample order.  sigma and the geometry
// features never reach HBM (the three-kernel form writes and re-reads 48 B per sample); only z_vals and weights, which
// NeRFRenderer.run returns, are stored.  FROM_FEATURES = true is the tail of the level-sliced path: the encoded
// features come from the scratch planes of k_encode_sliced_pairs instead of gathers.  Weight fragments in LDS as in
// k_render_occupancy_lds; arithmetic of the encode / MLPs / sigmoid identical to the separate kernels, the
// transmittance product is scanned per 16 samples instead of per 64 (differences at the 1e-7 level).
// DPP row_shr:n moves data n lanes up inside a row of 16 lanes (= one MFMA tile column group); lanes without a source keep
// `old`.  One VALU instruction per step where __shfl_up is an LDS round trip (ds_bpermute_b32 + s_waitcnt): the scan sits on
// the per-ray critical path of the one-ray-per-wave kernels.
template <int N>
__device__ __forceinline__ float row_shr(float old, float v) {
    static_assert(N >= 1 && N <= 15, "row_shr:1..15");
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x110 + N, 0xF, 0xF, false));
}

__device__ __forceinline__ float row_shl1(float old, float v) {  // lane c takes lane c + 1 of its row, lane 15 keeps `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, false));
}

__device__ __forceinline__ float row_first(float v) {  // lane 0 of the row, in every lane of the row (row_newbcast:0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150, 0xF, 0xF, false));
}

__device__ __forceinline__ float row16_scan_mul(float v) {  // inclusive product over the lanes 0..c of each row; x * 1.0f is exact
    v *= row_shr<1>(1.0f, v);
    v *= row_shr<2>(1.0f, v);
    v *= row_shr<4>(1.0f, v);
    v *= row_shr<8>(1.0f, v);
    return v;
}

// What the TRAINING form of the render kernels keeps beside z_vals / weights (ops.RenderRaysFn): everything the backward of the
// whole render reads -- unit-cube positions (table scatter), the 32 encoded features as fp16 rows in level order (the density MLP's
// backward recomputes its hidden layer from them), sigma (compositor backward), the geometry rows (h1 .. h15, 1.0) the heads'
// backward reads as their per-sample input, and the masked per-sample colours sigmoid(logits) [weight > w_thresh] (image / sigmoid
// backward).  The per-sample logits, the [M, 16] fp32 network outputs and the mask never reach memory.
struct RenderTrainOut {
    float* x01;      // [M, 3]   (written by the kernel that forms the positions: k_render_uniform<*, false> / k_encode_sliced_pairs)
    _Float16* feat;  // [M, 32], column 2 l + f
    _Float16* geo;   // [M, 16]
    float* sigma;    // [M]
    float* rgb;      // [M, C]
};

template <bool LIDAR, bool FROM_FEATURES, bool TRAIN = false, int FH = -1>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_uniform(
    RayBatch rb, const _Float16* __restrict__ table, uint32_t table_bytes, GridMeta meta, uint32_t first_hashed,
    const uint2* __restrict__ feat, const _Float16* __restrict__ w_sigma, const _Float16* __restrict__ w_a, const _Float16* __restrict__ w_b,
    float k_scale, float w_thresh, float bg0, float bg1, float bg2, int use_bg, float* __restrict__ z_vals, float* __restrict__ weights,
    float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image, RenderTrainOut rt = RenderTrainOut()) {
    using FR = OccFrags<LIDAR>;
    constexpr int F = 2;
    constexpr int IN_STEPS = FR::IN_STEPS;
    constexpr int C = LIDAR ? 2 : 3;
    __shared__ uint4 s_lv[kMaxLevels];
    __shared__ half8_t s_frag[FR::kCount * kWave];
    __shared__ float4_t s_pre[kWavesPerBlock * FR::kPre];
    if (threadIdx.x < kMaxLevels) {
        const uint32_t l = threadIdx.x;
        s_lv[l] = make_uint4(__builtin_bit_cast(uint32_t, meta.scale[l]), meta.res[l], meta.offset[l] * (uint32_t)(F * sizeof(_Float16)),
                             meta.offset[l + 1] - meta.offset[l]);
    }
    const int lane = lane_id(), g = lane >> 4, c = lane & 15;
    for (int f = (int)(threadIdx.x >> 6); f < FR::kCount; f += kWavesPerBlock) s_frag[f * kWave + lane] = occ_fragment<LIDAR>(f, lane, w_sigma, w_a, w_b);
    __syncthreads();
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (n >= rb.N) return;
    const half8_t* frag = s_frag + lane;
    const uint32_t T = rb.T;
    const size_t M = (size_t)rb.N * T;

    DensityCtx<F> cx;
    if constexpr (!FROM_FEATURES) {
        cx.g = g;
        cx.first_hashed = first_hashed;
        cx.lds_lv = s_lv + g;
        cx.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(table), 0, (int)table_bytes, 0x00020000);
    }
    const float rd0 = rb.rays_d[3 * (size_t)n], rd1 = rb.rays_d[3 * (size_t)n + 1], rd2 = rb.rays_d[3 * (size_t)n + 2];
    const float d0 = (rd0 + 1.0f) / 2.0f, d1 = (rd1 + 1.0f) / 2.0f, d2 = (rd2 + 1.0f) / 2.0f;
    half8_t xf[IN_STEPS];
    if constexpr (!LIDAR) {
        float sh[16];
        sh4_basis(d0, d1, d2, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[0][j] = (_Float16)(g == 0 ? sh[j] : sh[8 + j]);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int k = 32 * s + 8 * g + j;
                float sn = 1.0f, cs = 1.0f;
                if (k < 72) {
                    const int i = k / 24, f = (k - 24 * i) >> 1;
                    freq_pair(i == 0 ? d0 : (i == 1 ? d1 : d2), f, sn, cs);
                }
                xf[s][j] = (_Float16)sn;
                xf[s][j + 1] = (_Float16)cs;
            }
    }
    const bool takes_geo = LIDAR ? (g == 1 || g == 2) : (g >= 2);
    const int src_a = (c + 16 * (LIDAR ? 2 * (g - 1) : 2 * (g - 2))) & 63, src_b = (src_a + 16) & 63;
    float4_t* pre = s_pre + (threadIdx.x >> 6) * FR::kPre;
    ray_head_constants<LIDAR>(xf, w_a, w_b, lane, pre);
    auto head = [&](int hd, const half8_t& x_last) {
        const int base = FR::kHead + hd * FR::kPerHead;
        float4_t acc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
            if constexpr (IN_STEPS > 1) c = pre[(hd * 4 + t) * 4 + g];
            acc[t] = mfma16(frag[(base + t) * kWave], x_last, c);
        }
        half8_t h[kHidSteps];
        pack_hidden(acc, h);
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) c = mfma16(frag[(base + 4 + 2 * t + s) * kWave], h[s], c);
            acc[t] = c;
        }
        pack_hidden(acc, h);
        float4_t c = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s) c = mfma16(frag[(base + 4 + 8 + s) * kWave], h[s], c);
        return c;
    };

    const float near = rb.nears[n], range = rb.fars[n] - near;
    const float sample_dist = range / (float)T;
    const float ox = rb.rays_o[3 * (size_t)n], oy = rb.rays_o[3 * (size_t)n + 1], oz = rb.rays_o[3 * (size_t)n + 2];
    const size_t row0 = (size_t)n * T;
    float carry = 1.0f, ws = 0.0f, dp = 0.0f, img[C];
#pragma unroll
    for (int k = 0; k < C; ++k) img[k] = 0.0f;
    // the sample fractions (and jitter) of a tile are requested one tile ahead: two dependent L2 round trips at the head of every
    // tile's chain otherwise
    float lin_cur = 0.0f, lin_nxt = 0.0f, nz_cur = 0.5f, nz_nxt = 0.5f;
    auto request = [&](uint32_t i0) {
        const uint32_t i = i0 + (uint32_t)c;
        const uint32_t ic = i < T ? i : T - 1u, in = ic + 1u < T ? ic + 1u : ic;
        lin_cur = rb.lin[ic];
        lin_nxt = rb.lin[in];
        if (rb.noise) {
            nz_cur = rb.noise[row0 + ic];
            nz_nxt = rb.noise[row0 + in];
        }
    };
    if constexpr (!FROM_FEATURES) request(0u);
    for (uint32_t i0 = 0; i0 < T; i0 += 16) {
        const uint32_t i = i0 + (uint32_t)c;
        const bool valid = i < T;
        const uint32_t ic = valid ? i : T - 1u;
        const size_t s = row0 + ic;
        // ---- sample position / features
        float z, z_next;
        half8_t feat8;
        if constexpr (FROM_FEATURES) {
            z = z_vals[s];
            z_next = ic + 1u < T ? z_vals[s + 1] : z;
            const uint2 p0 = feat[(size_t)(2 * g) * M + s], p1 = feat[(size_t)(2 * g + 1) * M + s];
            typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
            const u4_t packed = {p0.x, p1.x, p1.y, p0.y};
            feat8 = __builtin_bit_cast(half8_t, packed);
        } else {
            z = near + range * lin_cur;
            z_next = near + range * lin_nxt;
            if (rb.noise) {
                z = z + (nz_cur - 0.5f) * sample_dist;
                z_next = z_next + (nz_nxt - 0.5f) * sample_dist;
            }
            if (i0 + 16u < T) request(i0 + 16u);
            float x[3];
            x[0] = (fminf(fmaxf(ox + rd0 * z, rb.lo[0]), rb.hi[0]) + rb.bound) * rb.inv_extent;
            x[1] = (fminf(fmaxf(oy + rd1 * z, rb.lo[1]), rb.hi[1]) + rb.bound) * rb.inv_extent;
            x[2] = (fminf(fmaxf(oz + rd2 * z, rb.lo[2]), rb.hi[2]) + rb.bound) * rb.inv_extent;
            feat8 = density_encode<F, 4, true, FH>(cx, x);
            if constexpr (TRAIN) {
                if (g == 0 && valid) { rt.x01[3 * s] = x[0]; rt.x01[3 * s + 1] = x[1]; rt.x01[3 * s + 2] = x[2]; }
            }
        }
        if constexpr (TRAIN) store_feat_row_swap(rt.feat, s, g, __builtin_bit_cast(uint4, feat8), valid);
        // ---- sigma MLP
        float4_t o;
        {
            float4_t acc1[kHidTiles];
#pragma unroll
            for (int tt = 0; tt < kHidTiles; ++tt) acc1[tt] = mfma16(frag[(FR::kSigma + tt) * kWave], feat8, float4_t{0, 0, 0, 0});
            half8_t h[kHidSteps];
            pack_hidden(acc1, h);
            float4_t a = {0, 0, 0, 0};
#pragma unroll
            for (int sx = 0; sx < kHidSteps; ++sx) a = mfma16(frag[(FR::kSigma + 4 + sx) * kWave], h[sx], a);
            o = a;
        }
        // ---- alpha compositing of the tile (renderer_dynamic.py:176-194); lanes g == 3 hold sigma of sample c
        const float delta = (i + 1u < T) ? z_next - z : sample_dist;
        float alpha = 0.0f;
        if (g == 3 && valid) alpha = 1.0f - expf(-delta * k_scale * expf(o[3]));
        const float om = (g == 3 && valid) ? (1.0f - alpha + 1e-15f) : 1.0f;
        const float incl = row16_scan_mul(om);
        const float excl = row_shr<1>(1.0f, incl);
        const float w = alpha * (carry * excl);  // zero outside lane group 3
        carry = carry * readlane_f32(incl, 63);
        if (g == 3 && valid) {
            weights[s] = w;
            if constexpr (!FROM_FEATURES) z_vals[s] = z;
            if constexpr (TRAIN) rt.sigma[s] = expf(o[3]);
        }
        ws += w;
        dp += w * z;
        // ---- heads on the samples that carry weight
        const float w0 = __shfl(w, 48 + c, 64);  // weight of sample c, for the lanes that hold its colour (g == 0); needed after the heads
        const bool on = w0 > w_thresh;
        float cr[C];  // TRAIN: masked colour of sample c (lanes g == 0)
#pragma unroll
        for (int k = 0; k < C; ++k) cr[k] = 0.0f;
        if constexpr (TRAIN) {  // geometry row (h1 .. h15, 1.0) of sample c: lane group g holds its columns 4g .. 4g + 3
            if (valid) *reinterpret_cast<uint2*>(rt.geo + s * 16 + 4 * g) = make_uint2(pack_h2(o[0], o[1]), pack_h2(o[2], g == 3 ? 1.0f : o[3]));
        }
        if (__ballot(w > w_thresh)) {  // w is zero outside lane group 3: the same set of samples
            const uint32_t p0 = pack_h2(o[0], o[1]), p1 = pack_h2(o[2], g == 3 ? 1.0f : o[3]);
            typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
            u4_t gv;
            gv[0] = (uint32_t)__shfl((int)p0, src_a);
            gv[1] = (uint32_t)__shfl((int)p1, src_a);
            gv[2] = (uint32_t)__shfl((int)p0, src_b);
            gv[3] = (uint32_t)__shfl((int)p1, src_b);
            const half8_t x_last = takes_geo ? __builtin_bit_cast(half8_t, gv) : xf[IN_STEPS - 1];
            const float4_t oa = head(0, x_last);
            if constexpr (LIDAR) {
                const float4_t ob = head(1, x_last);
                if (g == 0 && on) {
                    cr[0] = sigmoid_f32(oa[0]);
                    cr[1] = sigmoid_f32(ob[0]);
                    img[0] += w0 * cr[0];
                    img[1] += w0 * cr[1];
                }
            } else {
                if (g == 0 && on) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        cr[k] = sigmoid_f32(oa[k]);
                        img[k] += w0 * cr[k];
                    }
                }
            }
        }
        if constexpr (TRAIN) {
            if (g == 0 && valid) {
#pragma unroll
                for (int k = 0; k < C; ++k) rt.rgb[s * C + k] = cr[k];
            }
        }
    }
    ws = wave_sum(ws);
    dp = wave_sum(dp);
#pragma unroll
    for (int k = 0; k < C; ++k) img[k] = wave_sum(img[k]);
    if (lane == 0) {
        weights_sum[n] = ws;
        depth[n] = dp;
        const float bg[3] = {bg0, bg1, bg2};
        const float rest = use_bg ? 1.0f - ws : 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) image[(size_t)n * C + k] = use_bg ? img[k] + rest * bg[k] : img[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Tail of the level-sliced path, two MFMA tiles (32 samples) per iteration.
//
// k_render_uniform<*, true> walks a ray one 16-sample tile at a time: a single dependent chain per wave (fragment read ->
// 4 MFMAs -> pack -> 2 MFMAs -> exp -> scan -> shuffles -> 4 -> pack -> 8 -> pack -> 2 MFMAs -> sigmoid) with four waves per
// SIMD to cover it, and every MFMA re-reads its 1 KB weight fragment from LDS (20 KB per tile and wave for the camera
// field: per CU as many LDS cycles as MFMA cycles).  Here a wave takes tiles 2j and 2j+1 of its ray together: every
// fragment is read ONCE and feeds two MFMAs (half the LDS traffic), and the two tiles are independent chains up to the
// transmittance carry, which enters only at the scan (tile 2j+1 scans with tile 2j's carry).  Arithmetic per tile is that
// of k_render_uniform (same scans over 16 lanes, same MFMA chains): bit-identical outputs.
template <bool LIDAR, bool TRAIN = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_tail2(
    RayBatch rb, const uint2* __restrict__ feat, const _Float16* __restrict__ w_sigma, const _Float16* __restrict__ w_a,
    const _Float16* __restrict__ w_b, float k_scale, float w_thresh, float bg0, float bg1, float bg2, int use_bg,
    const float* __restrict__ z_vals, float* __restrict__ weights, float* __restrict__ weights_sum, float* __restrict__ depth,
    float* __restrict__ image, RenderTrainOut rt = RenderTrainOut()) {
    using FR = OccFrags<LIDAR>;
    constexpr int IN_STEPS = FR::IN_STEPS;
    constexpr int C = LIDAR ? 2 : 3;
    constexpr int kOut1 = FR::kCount;  // the sigma net's output layer once more, rows rotated by 5 instead of 1 (tile 2j + 1, see below)
    __shared__ half8_t s_frag[(FR::kCount + kHidSteps) * kWave];
    __shared__ float4_t s_pre[kWavesPerBlock * FR::kPre];
    const int lane = lane_id(), g = lane >> 4, c = lane & 15;
    for (int f = (int)(threadIdx.x >> 6); f < FR::kCount + kHidSteps; f += kWavesPerBlock)
        s_frag[f * kWave + lane] = f < FR::kCount ? occ_fragment<LIDAR>(f, lane, w_sigma, w_a, w_b) : load_w_chained(w_sigma + kHidden * 32, 0, f - FR::kCount, lane, 5);
    __syncthreads();
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (n >= rb.N) return;
    const half8_t* frag = s_frag + lane;
    const uint32_t T = rb.T;
    const size_t M = (size_t)rb.N * T;

    const float rd0 = rb.rays_d[3 * (size_t)n], rd1 = rb.rays_d[3 * (size_t)n + 1], rd2 = rb.rays_d[3 * (size_t)n + 2];
    const float d0 = (rd0 + 1.0f) / 2.0f, d1 = (rd1 + 1.0f) / 2.0f, d2 = (rd2 + 1.0f) / 2.0f;
    half8_t xf[IN_STEPS];
    if constexpr (!LIDAR) {
        float sh[16];
        sh4_basis(d0, d1, d2, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[0][j] = (_Float16)(g == 0 ? sh[j] : sh[8 + j]);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int k = 32 * s + 8 * g + j;
                float sn = 1.0f, cs = 1.0f;
                if (k < 72) {
                    const int i = k / 24, f = (k - 24 * i) >> 1;
                    freq_pair(i == 0 ? d0 : (i == 1 ? d1 : d2), f, sn, cs);
                }
                xf[s][j] = (_Float16)sn;
                xf[s][j + 1] = (_Float16)cs;
            }
    }
    const bool takes_geo = LIDAR ? (g == 1 || g == 2) : (g >= 2);
    const int src_a = (c + 16 * (LIDAR ? 2 * (g - 1) : 2 * (g - 2))) & 63, src_b = (src_a + 16) & 63;
    const int src_a1 = (src_a + 48) & 63, src_b1 = (src_b + 48) & 63;  // tile 2j + 1: its output rows sit one lane group lower
    float4_t* pre = s_pre + (threadIdx.x >> 6) * FR::kPre;
    ray_head_constants<LIDAR>(xf, w_a, w_b, lane, pre);
    // one head on both tiles: every fragment read feeds two MFMAs
    auto head2 = [&](int hd, const half8_t (&x_last)[2], float4_t (&out)[2]) {
        const int base = FR::kHead + hd * FR::kPerHead;
        float4_t acc[2][kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
            if constexpr (IN_STEPS > 1) c = pre[(hd * 4 + t) * 4 + g];
            const half8_t w = frag[(base + t) * kWave];
            acc[0][t] = mfma16(w, x_last[0], c);
            acc[1][t] = mfma16(w, x_last[1], c);
        }
        half8_t h[2][kHidSteps];
        pack_hidden(acc[0], h[0]);
        pack_hidden(acc[1], h[1]);
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) {
                const half8_t w = frag[(base + 4 + 2 * t + s) * kWave];
                a0 = mfma16(w, h[0][s], a0);
                a1 = mfma16(w, h[1][s], a1);
            }
            acc[0][t] = a0;
            acc[1][t] = a1;
        }
        pack_hidden(acc[0], h[0]);
        pack_hidden(acc[1], h[1]);
        float4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s) {
            const half8_t w = frag[(base + 4 + 8 + s) * kWave];
            a0 = mfma16(w, h[0][s], a0);
            a1 = mfma16(w, h[1][s], a1);
        }
        out[0] = a0;
        out[1] = a1;
    };

    const float sample_dist = (rb.fars[n] - rb.nears[n]) / (float)T;
    const size_t row0 = (size_t)n * T;
    const uint2* plane0 = feat + (size_t)(2 * g) * M + row0;
    const uint2* plane1 = feat + (size_t)(2 * g + 1) * M + row0;
    const float* zrow = z_vals + row0;
    float carry = 1.0f, ws = 0.0f, dp = 0.0f, img[C];
#pragma unroll
    for (int k = 0; k < C; ++k) img[k] = 0.0f;
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    // features / depths of the NEXT pair of tiles are requested before the current pair is worked on: with one ray per wave and
    // every ray resident (4 waves x 1024 SIMDs = 4096 rays) the launch lasts as long as ONE wave's walk along its ray, so
    // a memory round trip per iteration would sit on the critical path 24 times.
    struct TileIn {
        uint2 p0, p1;
        float z;
    };
    auto fetch = [&](uint32_t i0, int u) {
        const uint32_t i = i0 + 16u * (uint32_t)u + (uint32_t)c;
        const uint32_t k = i < T ? i : T - 1u;
        TileIn t;
        t.p0 = plane0[k];
        t.p1 = plane1[k];
        t.z = zrow[k];
        return t;
    };
    TileIn cur[2] = {fetch(0, 0), fetch(0, 1)};
    for (uint32_t i0 = 0; i0 < T; i0 += 32) {
        const uint32_t j0 = i0 + 32u < T ? i0 + 32u : i0;  // last iteration: an in-range dummy, never used
        const TileIn nxt[2] = {fetch(j0, 0), fetch(j0, 1)};
        bool valid[2];
        uint32_t idx[2];
        float z[2], z_next[2];
        half8_t feat8[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t i = i0 + 16u * u + (uint32_t)c;
            valid[u] = i < T;
            idx[u] = valid[u] ? i : T - 1u;
            z[u] = cur[u].z;
            const u4_t packed = {cur[u].p0.x, cur[u].p1.x, cur[u].p1.y, cur[u].p0.y};
            feat8[u] = __builtin_bit_cast(half8_t, packed);
            if constexpr (TRAIN)  // feature rows in level order (fragment pieces = levels g, g + 4, g + 8, g + 12)
                store_feat_row_swap(rt.feat, row0 + idx[u], g, make_uint4(packed.x, packed.y, packed.z, packed.w), valid[u]);
        }
        // depth of the following sample: lane c + 1 of the tile, lane 0 of the next tile for c == 15 (the same fp32 words the
        // one-tile kernel loads as z_vals[s + 1]); only read where i + 1 < T
        z_next[0] = row_shl1(row_first(z[1]), z[0]);
        z_next[1] = row_shl1(row_first(nxt[0].z), z[1]);
        // ---- sigma MLP on both tiles
        float4_t o[2];
        {
            float4_t acc[2][kHidTiles];
#pragma unroll
            for (int tt = 0; tt < kHidTiles; ++tt) {
                const half8_t w = frag[(FR::kSigma + tt) * kWave];
                acc[0][tt] = mfma16(w, feat8[0], float4_t{0, 0, 0, 0});
                acc[1][tt] = mfma16(w, feat8[1], float4_t{0, 0, 0, 0});
            }
            half8_t h[2][kHidSteps];
            pack_hidden(acc[0], h[0]);
            pack_hidden(acc[1], h[1]);
            float4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
            for (int sx = 0; sx < kHidSteps; ++sx) {
                a0 = mfma16(frag[(FR::kSigma + 4 + sx) * kWave], h[0][sx], a0);
                a1 = mfma16(frag[(kOut1 + sx) * kWave], h[1][sx], a1);
            }
            o[0] = a0;
            o[1] = a1;
        }
        // ---- alpha compositing of both tiles in one pass (renderer_dynamic.py:176-194).  The output layer of tile 2j + 1 ran with its
        // rows rotated by 5, so the density logit of its sample c sits in lane group 2 where tile 2j has it in lane group 3: the two
        // exponentials, the 16-lane scan and the weights are formed once for 32 samples (rows of 16 lanes scan independently), and
        // tile 2j + 1 continues from tile 2j's transmittance: the same products in the same order as tile by tile.
        float w0[2];
        bool on[2];
        float w;
        {
            const bool hi = g == 3;
            const float lg = hi ? o[0][3] : o[1][3], zz = hi ? z[0] : z[1], zn = hi ? z_next[0] : z_next[1];
            const uint32_t i = i0 + (hi ? 0u : 16u) + (uint32_t)c;
            const bool live = g >= 2 && i < T;
            const float delta = (i + 1u < T) ? zn - zz : sample_dist;
            float alpha = 0.0f;
            if (live) alpha = 1.0f - expf(-delta * k_scale * expf(lg));
            const float om = live ? (1.0f - alpha + 1e-15f) : 1.0f;
            const float incl = row16_scan_mul(om);
            const float excl = row_shr<1>(1.0f, incl);
            const float carry1 = carry * readlane_f32(incl, 63);  // transmittance in front of tile 2j + 1
            w = alpha * ((hi ? carry : carry1) * excl);  // zero outside lane groups 2, 3
            carry = carry1 * readlane_f32(incl, 47);
            if (live) {
                weights[row0 + i] = w;
                if constexpr (TRAIN) rt.sigma[row0 + i] = expf(lg);
            }
            w0[0] = __shfl(w, 48 + c, 64);  // weights of sample c of either tile, for the lanes that hold its colour (g == 0)
            w0[1] = __shfl(w, 32 + c, 64);
            on[0] = w0[0] > w_thresh;
            on[1] = w0[1] > w_thresh;
            // sums in lane group 0, tile 2j before tile 2j + 1 per lane as in the one-tile kernel (bit-identical totals)
            const float a0 = g == 0 ? w0[0] : 0.0f, a1 = g == 0 ? w0[1] : 0.0f;
            ws += a0;
            dp += a0 * z[0];
            ws += a1;
            dp += a1 * z[1];
        }
        float cr[2][C];  // TRAIN: masked colours of sample c of either tile (lanes g == 0)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < C; ++k) cr[u][k] = 0.0f;
        if constexpr (TRAIN) {
            // geometry rows (h1 .. h15, 1.0): tile 2j has columns 4g .. 4g + 3 in lane group g; tile 2j + 1 ran its output layer with the
            // rows rotated by 5, so lane group g holds columns 4 (g + 1 mod 4) .. of its sample and the logit sits in lane group 2
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t q0 = pack_h2(o[u][0], o[u][1]), q1 = pack_h2(o[u][2], g == 3 - u ? 1.0f : o[u][3]);
                const int col = u ? 4 * ((g + 1) & 3) : 4 * g;
                if (valid[u]) *reinterpret_cast<uint2*>(rt.geo + (row0 + idx[u]) * 16 + col) = make_uint2(q0, q1);
            }
        }
        // ---- heads on the samples that carry weight (the test reads the weights where they are: zero outside lane groups 2, 3)
        if (__ballot(w > w_thresh)) {
            half8_t x_last[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t p0 = pack_h2(o[u][0], o[u][1]), p1 = pack_h2(o[u][2], g == 3 - u ? 1.0f : o[u][3]);
                u4_t gv;
                gv[0] = (uint32_t)__shfl((int)p0, u ? src_a1 : src_a);
                gv[1] = (uint32_t)__shfl((int)p1, u ? src_a1 : src_a);
                gv[2] = (uint32_t)__shfl((int)p0, u ? src_b1 : src_b);
                gv[3] = (uint32_t)__shfl((int)p1, u ? src_b1 : src_b);
                x_last[u] = takes_geo ? __builtin_bit_cast(half8_t, gv) : xf[IN_STEPS - 1];
            }
            float4_t oa[2];
            head2(0, x_last, oa);
            if constexpr (LIDAR) {
                float4_t ob[2];
                head2(1, x_last, ob);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (g == 0 && on[u]) {
                        cr[u][0] = sigmoid_f32(oa[u][0]);
                        cr[u][1] = sigmoid_f32(ob[u][0]);
                        img[0] += w0[u] * cr[u][0];
                        img[1] += w0[u] * cr[u][1];
                    }
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (g == 0 && on[u]) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            cr[u][k] = sigmoid_f32(oa[u][k]);
                            img[k] += w0[u] * cr[u][k];
                        }
                    }
            }
        }
        if constexpr (TRAIN) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (g == 0 && valid[u]) {
#pragma unroll
                    for (int k = 0; k < C; ++k) rt.rgb[(row0 + idx[u]) * C + k] = cr[u][k];
                }
        }
        cur[0] = nxt[0];
        cur[1] = nxt[1];
    }
    ws = wave_sum(ws);
    dp = wave_sum(dp);
#pragma unroll
    for (int k = 0; k < C; ++k) img[k] = wave_sum(img[k]);
    if (lane == 0) {
        weights_sum[n] = ws;
        depth[n] = dp;
        const float bg[3] = {bg0, bg1, bg2};
        const float rest = use_bg ? 1.0f - ws : 0.0f;
#pragma unroll
        for (int k = 0; k < C; ++k) image[(size_t)n * C + k] = use_bg ? img[k] + rest * bg[k] : img[k];
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

static int density_uniform_impl(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16,
                                float* z_vals, float* sigmas, void* geo_f16, void* feat_scratch, uint32_t sliced_passes, hipStream_t stream,
                                const TrainOut* train = nullptr) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && nears && fars && lin && h_aabb && table_f16 && sigma_weights_f16 && z_vals && sigmas && geo_f16);
    REQUIRE(bound > 0.0f);
    REQUIRE((reinterpret_cast<uintptr_t>(table_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(sigma_weights_f16) & 15u) == 0 &&
            (reinterpret_cast<uintptr_t>(geo_f16) & 15u) == 0);
    if (L * F != 32 || (F != 2 && F != 4)) return NVSF_ERR_UNSUPPORTED;
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    RayBatch rb;
    rb.rays_o = rays_o; rb.rays_d = rays_d; rb.nears = nears; rb.fars = fars; rb.lin = lin; rb.noise = noise;
    for (int k = 0; k < 3; ++k) { rb.lo[k] = h_aabb[k]; rb.hi[k] = h_aabb[3 + k]; }
    rb.bound = bound;
    rb.inv_extent = 1.0f / (2.0f * bound);
    rb.N = N; rb.T = T;
    const unsigned long long n_tiles = ((unsigned long long)N * T + 15) / 16;
    const unsigned long long want = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t blocks = (uint32_t)(want < 4096ull ? want : 4096ull);
    const _Float16* tb = reinterpret_cast<const _Float16*>(table_f16);
    const _Float16* ws = reinterpret_cast<const _Float16*>(sigma_weights_f16);
    _Float16* gp = reinterpret_cast<_Float16*>(geo_f16);
    // v2 needs "dense levels first, hashed levels after" (true for any per_level_scale >= 1) and a table < 2 GiB
    uint32_t first_hashed = L;
    bool monotone = true;
    for (uint32_t l = 0; l < L; ++l) {
        const unsigned long long cells = (unsigned long long)h_res[l] * h_res[l] * h_res[l];
        const bool hashed = cells > (unsigned long long)(h_offsets[l + 1] - h_offsets[l]);
        if (hashed && first_hashed == L) first_hashed = l;
        if (!hashed && first_hashed != L) monotone = false;
        if (hashed && ((h_offsets[l + 1] - h_offsets[l]) & (h_offsets[l + 1] - h_offsets[l] - 1u)) != 0u) monotone = false;
    }
    const unsigned long long table_bytes = (unsigned long long)h_offsets[L] * F * sizeof(_Float16);
    if (sliced_passes) {
        const unsigned long long total = (unsigned long long)N * T;
        const bool f4 = F == 4 && L == 8;  // the reference-default shape: one level per XCD group (k_encode_sliced_f4), same planes
        if (!(((F == 2 && L == 16) || f4) && monotone && table_bytes < (1ull << 31) && total < (f4 ? (1ull << 28) : (1ull << 32)))) return NVSF_ERR_UNSUPPORTED;
        REQUIRE(feat_scratch && (reinterpret_cast<uintptr_t>(feat_scratch) & 15u) == 0);
        const uint32_t M = (uint32_t)total;
        uint2* fp = reinterpret_cast<uint2*>(feat_scratch);
        if ((sliced_passes & 1u) && f4) {
            const uint32_t units32 = (M + 31u) / 32u;
            uint32_t ps = (units32 + kWavesPerBlock - 1) / kWavesPerBlock;
            if (ps > 512u) ps = 512u;
            const SlicePlan plan = slice_plan(units32, T, h_res, first_hashed, false);  // every group its own level
            float* x01 = train ? train->x01 : nullptr;
            if (T % 32u == 0u)
                hipLaunchKernelGGL((k_encode_sliced_f4<true>), dim3(8u * ps), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                                   first_hashed, M, z_vals, reinterpret_cast<uint32_t*>(fp), plan, x01);
            else
                hipLaunchKernelGGL((k_encode_sliced_f4<false>), dim3(8u * ps), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                                   first_hashed, M, z_vals, reinterpret_cast<uint32_t*>(fp), plan, x01);
        } else if (sliced_passes & 1u) {
            const uint32_t units32 = (M + 31u) / 32u;
            uint32_t ps = (units32 + kWavesPerBlock - 1) / kWavesPerBlock;
            if (ps > 512u) ps = 512u;  // 32 CUs per XCD x 8 resident workgroups x 2 (measured: 256 -> 512 gains 1.5 %)
            const SlicePlan plan = slice_plan(units32, T, h_res, first_hashed, nvsf_variant(kVarSlicePlan) == 0);  // 1 (tests): every group its own slice only
            float* x01 = train ? train->x01 : nullptr;
            if (T % 32u == 0u)
                hipLaunchKernelGGL((k_encode_sliced_pairs<2, true>), dim3(8u * ps), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta, L,
                                   first_hashed, M, z_vals, fp, plan, x01);
            else
                hipLaunchKernelGGL((k_encode_sliced_pairs<2, false>), dim3(8u * ps), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                                   L, first_hashed, M, z_vals, fp, plan, x01);
        }
        const uint32_t tiles = (M + 15u) / 16u;
        uint32_t bb = (tiles + 4u * kWavesPerBlock - 1u) / (4u * kWavesPerBlock);
        if (bb > 4096u) bb = 4096u;
        if (sliced_passes & 2u) {
            if (train) hipLaunchKernelGGL((k_density_from_features<2, true>), dim3(bb), dim3(kBlock), 0, stream, fp, M, L, ws, sigmas, gp, *train);
            else hipLaunchKernelGGL((k_density_from_features<2, false>), dim3(bb), dim3(kBlock), 0, stream, fp, M, L, ws, sigmas, gp, TrainOut());
        }
        return nvsf_launch_status();
    }
    const bool use_v2 = F == 2 && monotone && table_bytes < (1ull << 31);  // else: the generic first formulation (any F with L F = 32)
    if (train) {
        if (!use_v2) return NVSF_ERR_UNSUPPORTED;
        if (T % 16u == 0u) {
            const uint32_t seg_tiles = 4u;
            const unsigned long long units = (unsigned long long)N * ((T / 16 + seg_tiles - 1) / seg_tiles);
            const uint32_t segb = (uint32_t)(units < 3072ull ? units : 3072ull);
            hipLaunchKernelGGL((k_density_uniform_v2<2, true, 4, true>), dim3(segb), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                               first_hashed, seg_tiles, ws, z_vals, sigmas, gp, *train);
        } else {
            hipLaunchKernelGGL((k_density_uniform_v2<2, false, 4, true>), dim3(blocks), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                               first_hashed, 0u, ws, z_vals, sigmas, gp, *train);
        }
        return nvsf_launch_status();
    }
    if (use_v2) {
        const bool seg = T % 16u == 0u;
        if (seg) {
            const uint32_t seg_tiles = 4u;  // 4 tiles = one round of the 4 waves (measured best)
            const unsigned long long units = (unsigned long long)N * ((T / 16 + seg_tiles - 1) / seg_tiles);
            const uint32_t segb = (uint32_t)(units < 3072ull ? units : 3072ull);  // 256 CUs x 3 workgroups (VGPR-limited residency) x 4
            hipLaunchKernelGGL((k_density_uniform_v2<2, true, 4, false>), dim3(segb), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                               first_hashed, seg_tiles, ws, z_vals, sigmas, gp, TrainOut());
        } else {
            hipLaunchKernelGGL((k_density_uniform_v2<2, false, 4, false>), dim3(blocks), dim3(kBlock), 0, stream, rb, tb, (uint32_t)table_bytes, meta,
                               first_hashed, 0u, ws, z_vals, sigmas, gp, TrainOut());
        }
    }
    else if (F == 2) hipLaunchKernelGGL(k_density_uniform<2>, dim3(blocks), dim3(kBlock), 0, stream, rb, tb, meta, ws, z_vals, sigmas, gp);
    else hipLaunchKernelGGL(k_density_uniform<4>, dim3(blocks), dim3(kBlock), 0, stream, rb, tb, meta, ws, z_vals, sigmas, gp);
    return nvsf_launch_status();
}

NVSF_API int nvsf_field_density_uniform_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                            const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                            uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                            const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16,
                                            float* z_vals, float* sigmas, void* geo_f16, hipStream_t stream) {
    return density_uniform_impl(rays_o, rays_d, nears, fars, lin, noise, h_aabb, bound, N, T, table_f16, L, F, h_scales, h_res, h_offsets,
                                sigma_weights_f16, z_vals, sigmas, geo_f16, nullptr, 0u, stream);
}

NVSF_API int nvsf_field_density_uniform_train_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                                  const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                                  uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                                  const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16,
                                                  float* z_vals, float* sigmas, void* geo_f16, float* x01, void* feat_rows_f16, float* h32,
                                                  void* feat_scratch, hipStream_t stream) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(x01 && feat_rows_f16 && h32 && (reinterpret_cast<uintptr_t>(feat_rows_f16) & 3u) == 0);
    if (!((L == 16 && F == 2) || (L == 8 && F == 4 && feat_scratch))) return NVSF_ERR_UNSUPPORTED;  // L8 F4: level-sliced form only
    TrainOut tr;
    tr.x01 = x01; tr.feat = reinterpret_cast<_Float16*>(feat_rows_f16); tr.h32 = h32;
    return density_uniform_impl(rays_o, rays_d, nears, fars, lin, noise, h_aabb, bound, N, T, table_f16, L, F, h_scales, h_res, h_offsets,
                                sigma_weights_f16, z_vals, sigmas, geo_f16, feat_scratch, feat_scratch ? 3u : 0u, stream, &tr);
}

NVSF_API int nvsf_field_density_uniform_sliced_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                                   const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                                   uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                                   const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16,
                                                   float* z_vals, float* sigmas, void* geo_f16, void* feat_scratch, uint32_t passes,
                                                   hipStream_t stream) {
    if (passes == 0u || passes > 3u) return NVSF_ERR_INVALID_ARG;
    return density_uniform_impl(rays_o, rays_d, nears, fars, lin, noise, h_aabb, bound, N, T, table_f16, L, F, h_scales, h_res, h_offsets,
                                sigma_weights_f16, z_vals, sigmas, geo_f16, feat_scratch, passes, stream);
}

NVSF_API int nvsf_field_heads_uniform_fwd(const float* weights, const void* geo_f16, const float* rays_d, const float* weights_sum,
                                          int lidar, const void* head_a_weights_f16, const void* head_b_weights_f16, uint32_t N,
                                          uint32_t T, float w_thresh, const float* h_bg_color, float* image, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(T > 0 && weights && geo_f16 && rays_d && head_a_weights_f16 && image);
    REQUIRE(!lidar || head_b_weights_f16);
    REQUIRE(!h_bg_color || weights_sum);
    REQUIRE((reinterpret_cast<uintptr_t>(geo_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(head_a_weights_f16) & 15u) == 0 &&
            (reinterpret_cast<uintptr_t>(head_b_weights_f16) & 15u) == 0);
    const _Float16* gp = reinterpret_cast<const _Float16*>(geo_f16);
    const _Float16* wa = reinterpret_cast<const _Float16*>(head_a_weights_f16);
    const _Float16* wb = reinterpret_cast<const _Float16*>(head_b_weights_f16);
    const float b0 = h_bg_color ? h_bg_color[0] : 0.0f, b1 = h_bg_color ? h_bg_color[1] : 0.0f,
                b2 = (h_bg_color && !lidar) ? h_bg_color[2] : 0.0f;
    const dim3 grid(cdiv(N, kWavesPerBlock)), block(kBlock);
    if (lidar)
        hipLaunchKernelGGL(k_heads_uniform<true>, dim3(grid.x, 2), block, 0, stream, weights, gp, rays_d, weights_sum, wa, wb, N, T, w_thresh, b0,
                           b1, b2, h_bg_color ? 1 : 0, image);
    else
        hipLaunchKernelGGL(k_heads_uniform<false>, grid, block, 0, stream, weights, gp, rays_d, weights_sum, wa, wb, N, T, w_thresh,
                           b0, b1, b2, h_bg_color ? 1 : 0, image);
    return nvsf_launch_status();
}

NVSF_API int nvsf_render_occupancy_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                       const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                       uint32_t N, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                       const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16, int lidar,
                                       const void* head_a_weights_f16, const void* head_b_weights_f16, float density_scale,
                                       float T_thresh, const float* h_bg_color, float* weights_sum, float* depth, float* image,
                                       hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && nears && fars && grid && table_f16 && sigma_weights_f16 && head_a_weights_f16 && weights_sum && depth && image);
    REQUIRE(!lidar || head_b_weights_f16);
    REQUIRE(bound > 0.0f && C >= 1 && C <= 8 && H >= 2 && H <= 1024 && max_steps >= 1);
    REQUIRE((reinterpret_cast<uintptr_t>(table_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(sigma_weights_f16) & 15u) == 0 &&
            (reinterpret_cast<uintptr_t>(head_a_weights_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(head_b_weights_f16) & 15u) == 0);
    if ((F != 2 && F != 4) || L * F != 32) return NVSF_ERR_UNSUPPORTED;  // 16 levels x 2 (BASELINE config 2) or 8 x 4 (the reference default)
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    uint32_t first_hashed = L;
    for (uint32_t l = 0; l < L; ++l) {
        const unsigned long long cells = (unsigned long long)h_res[l] * h_res[l] * h_res[l];
        const uint32_t rows = h_offsets[l + 1] - h_offsets[l];
        const bool hashed = cells > (unsigned long long)rows;
        if (hashed && first_hashed == L) first_hashed = l;
        if (!hashed && first_hashed != L) return NVSF_ERR_UNSUPPORTED;         // dense levels must come first
        if (hashed && (rows & (rows - 1u)) != 0u) return NVSF_ERR_UNSUPPORTED;  // hashed levels: power-of-two rows
    }
    const unsigned long long table_bytes = (unsigned long long)h_offsets[L] * F * sizeof(_Float16);
    if (table_bytes >= (1ull << 31)) return NVSF_ERR_UNSUPPORTED;
    OccRays rr;
    rr.rays_o = rays_o; rr.rays_d = rays_d; rr.nears = nears; rr.fars = fars; rr.grid = grid;
    rr.bound = bound; rr.dt_gamma = dt_gamma; rr.max_steps = max_steps; rr.C = C; rr.H = H; rr.N = N;
    const _Float16* tb = reinterpret_cast<const _Float16*>(table_f16);
    const _Float16* ws = reinterpret_cast<const _Float16*>(sigma_weights_f16);
    const _Float16* wa = reinterpret_cast<const _Float16*>(head_a_weights_f16);
    const _Float16* wb = reinterpret_cast<const _Float16*>(head_b_weights_f16);
    const float b0 = h_bg_color ? h_bg_color[0] : 0.0f, b1 = h_bg_color ? h_bg_color[1] : 0.0f, b2 = h_bg_color ? h_bg_color[2] : 0.0f;
    const dim3 grid_dim(cdiv(N, kWavesPerBlock)), block(kBlock);
#define LAUNCH_OCC(LD, FF, FH)                                                                                                            \
    hipLaunchKernelGGL((k_render_occupancy_lds<LD, FF, FH>), grid_dim, block, 0, stream, rr, tb, (uint32_t)table_bytes, meta, first_hashed, ws, \
                       wa, wb, density_scale, T_thresh, b0, b1, b2, weights_sum, depth, image)
    const bool c2_grid = F == 2 && first_hashed == kFirstHashedC2 && nvsf_variant(kVarLevelKinds) == 0;  // the instance with the level kinds compiled in (density_encode)
    if (lidar) { if (c2_grid) LAUNCH_OCC(true, 2, (int)kFirstHashedC2); else if (F == 2) LAUNCH_OCC(true, 2, -1); else LAUNCH_OCC(true, 4, -1); }
    else { if (c2_grid) LAUNCH_OCC(false, 2, (int)kFirstHashedC2); else if (F == 2) LAUNCH_OCC(false, 2, -1); else LAUNCH_OCC(false, 4, -1); }
#undef LAUNCH_OCC
    return nvsf_launch_status();
}

static int render_uniform_impl(const float* rays_o, const float* rays_d, const float* nears, const float* fars, const float* lin,
                               const float* noise, const float* h_aabb, float bound, uint32_t N, uint32_t T, const void* table_f16,
                               uint32_t L, uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                               const void* sigma_weights_f16, int lidar, const void* head_a_weights_f16,
                               const void* head_b_weights_f16, float k_scale, float w_thresh, const float* h_bg_color,
                               const void* feat_scratch, float* z_vals, float* weights, float* weights_sum, float* depth, float* image,
                               hipStream_t stream, const RenderTrainOut* train) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && nears && fars && h_aabb && table_f16 && sigma_weights_f16 && head_a_weights_f16);
    REQUIRE(z_vals && weights && weights_sum && depth && image && (feat_scratch || lin));
    REQUIRE(!lidar || head_b_weights_f16);
    REQUIRE(bound > 0.0f);
    REQUIRE((reinterpret_cast<uintptr_t>(table_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(sigma_weights_f16) & 15u) == 0 &&
            (reinterpret_cast<uintptr_t>(head_a_weights_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(head_b_weights_f16) & 15u) == 0 &&
            (reinterpret_cast<uintptr_t>(feat_scratch) & 15u) == 0);
    // L16 F2 (BASELINE config 2): gathers in the render kernel or feature planes; L8 F4 (the reference-default grid): feature planes
    // only -- the streaming tails read column pairs and do not know which shape produced them (k_encode_sliced_f4)
    if (!((F == 2 && L == 16) || (F == 4 && L == 8 && feat_scratch))) return NVSF_ERR_UNSUPPORTED;
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    uint32_t first_hashed = L;
    for (uint32_t l = 0; l < L; ++l) {
        const unsigned long long cells = (unsigned long long)h_res[l] * h_res[l] * h_res[l];
        const uint32_t rows = h_offsets[l + 1] - h_offsets[l];
        const bool hashed = cells > (unsigned long long)rows;
        if (hashed && first_hashed == L) first_hashed = l;
        if (!hashed && first_hashed != L) return NVSF_ERR_UNSUPPORTED;
        if (hashed && (rows & (rows - 1u)) != 0u) return NVSF_ERR_UNSUPPORTED;
    }
    const unsigned long long table_bytes = (unsigned long long)h_offsets[L] * F * sizeof(_Float16);
    if (table_bytes >= (1ull << 31)) return NVSF_ERR_UNSUPPORTED;
    RayBatch rb;
    rb.rays_o = rays_o; rb.rays_d = rays_d; rb.nears = nears; rb.fars = fars; rb.lin = lin; rb.noise = noise;
    for (int k = 0; k < 3; ++k) { rb.lo[k] = h_aabb[k]; rb.hi[k] = h_aabb[3 + k]; }
    rb.bound = bound;
    rb.inv_extent = 1.0f / (2.0f * bound);
    rb.N = N; rb.T = T;
    const _Float16* tb = reinterpret_cast<const _Float16*>(table_f16);
    const _Float16* ws = reinterpret_cast<const _Float16*>(sigma_weights_f16);
    const _Float16* wa = reinterpret_cast<const _Float16*>(head_a_weights_f16);
    const _Float16* wb = reinterpret_cast<const _Float16*>(head_b_weights_f16);
    const uint2* fp = reinterpret_cast<const uint2*>(feat_scratch);
    const float b0 = h_bg_color ? h_bg_color[0] : 0.0f, b1 = h_bg_color ? h_bg_color[1] : 0.0f, b2 = h_bg_color ? h_bg_color[2] : 0.0f;
    const int use_bg = (h_bg_color && !lidar) ? 1 : 0;
    const dim3 grid_dim(cdiv(N, kWavesPerBlock)), block(kBlock);
#define LAUNCH_RU_FH(LD, FF, FH)                                                                                                      \
    do {                                                                                                                              \
        if (train)                                                                                                                    \
            hipLaunchKernelGGL((k_render_uniform<LD, FF, true, FH>), grid_dim, block, 0, stream, rb, tb, (uint32_t)table_bytes, meta, \
                               first_hashed, fp, ws, wa, wb, k_scale, w_thresh, b0, b1, b2, use_bg, z_vals, weights, weights_sum,     \
                               depth, image, *train);                                                                                \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_render_uniform<LD, FF, false, FH>), grid_dim, block, 0, stream, rb, tb, (uint32_t)table_bytes,      \
                               meta, first_hashed, fp, ws, wa, wb, k_scale, w_thresh, b0, b1, b2, use_bg, z_vals, weights,            \
                               weights_sum, depth, image, RenderTrainOut());                                                         \
    } while (0)
    // the gathering form has an instance for the grid of BASELINE config 2 (levels 0-4 dense, 5-15 hashed) beside the general one
#define LAUNCH_RU(LD, FF)                                                                                                             \
    do {                                                                                                                              \
        if (!FF && first_hashed == kFirstHashedC2 && nvsf_variant(kVarLevelKinds) == 0) LAUNCH_RU_FH(LD, FF, (FF ? -1 : (int)kFirstHashedC2)); \
        else LAUNCH_RU_FH(LD, FF, -1);                                                                                                \
    } while (0)
#define LAUNCH_TAIL(LD)                                                                                                               \
    do {                                                                                                                              \
        if (train)                                                                                                                    \
            hipLaunchKernelGGL((k_render_tail2<LD, true>), grid_dim, block, 0, stream, rb, fp, ws, wa, wb, k_scale, w_thresh, b0, b1, \
                               b2, use_bg, z_vals, weights, weights_sum, depth, image, *train);                                      \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_render_tail2<LD, false>), grid_dim, block, 0, stream, rb, fp, ws, wa, wb, k_scale, w_thresh, b0,    \
                               b1, b2, use_bg, z_vals, weights, weights_sum, depth, image, RenderTrainOut());                        \
    } while (0)
    const bool tail2 = nvsf_variant(kVarRenderTail) == 0;  // 1 (tests): one tile per iteration (k_render_uniform<*, true>)
    if (lidar) { if (fp) { if (tail2) LAUNCH_TAIL(true); else LAUNCH_RU(true, true); } else LAUNCH_RU(true, false); }
    else { if (fp) { if (tail2) LAUNCH_TAIL(false); else LAUNCH_RU(false, true); } else LAUNCH_RU(false, false); }
#undef LAUNCH_TAIL
#undef LAUNCH_RU
#undef LAUNCH_RU_FH
    return nvsf_launch_status();
}

NVSF_API int nvsf_render_uniform_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars, const float* lin,
                                     const float* noise, const float* h_aabb, float bound, uint32_t N, uint32_t T, const void* table_f16,
                                     uint32_t L, uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                                     const void* sigma_weights_f16, int lidar, const void* head_a_weights_f16,
                                     const void* head_b_weights_f16, float k_scale, float w_thresh, const float* h_bg_color,
                                     const void* feat_scratch, float* z_vals, float* weights, float* weights_sum, float* depth, float* image,
                                     hipStream_t stream) {
    return render_uniform_impl(rays_o, rays_d, nears, fars, lin, noise, h_aabb, bound, N, T, table_f16, L, F, h_scales, h_res, h_offsets,
                               sigma_weights_f16, lidar, head_a_weights_f16, head_b_weights_f16, k_scale, w_thresh, h_bg_color, feat_scratch,
                               z_vals, weights, weights_sum, depth, image, stream, nullptr);
}

// The TRAINING forward of a whole uniform render (ops.RenderRaysFn): nvsf_render_uniform_fwd's launch(es) -- with feat_scratch the
// level-sliced encode pass runs first, here, and writes the positions -- that also keep what the backward of the render reads
// (RenderTrainOut).  Same image / depth / weights as nvsf_render_uniform_fwd bit for bit.
NVSF_API int nvsf_render_uniform_train_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars, const float* lin,
                                           const float* noise, const float* h_aabb, float bound, uint32_t N, uint32_t T,
                                           const void* table_f16, uint32_t L, uint32_t F, const float* h_scales, const uint32_t* h_res,
                                           const uint32_t* h_offsets, const void* sigma_weights_f16, int lidar,
                                           const void* head_a_weights_f16, const void* head_b_weights_f16, float k_scale, float w_thresh,
                                           const float* h_bg_color, void* feat_scratch, float* z_vals, float* weights, float* weights_sum,
                                           float* depth, float* image, float* x01, void* feat_rows_f16, void* geo_f16, float* sigmas,
                                           float* rgbs, hipStream_t stream) {
    if (N == 0 || T == 0) return NVSF_OK;
    REQUIRE(x01 && feat_rows_f16 && geo_f16 && sigmas && rgbs && lin);
    REQUIRE((reinterpret_cast<uintptr_t>(feat_rows_f16) & 15u) == 0 && (reinterpret_cast<uintptr_t>(geo_f16) & 15u) == 0);
    RenderTrainOut rt;
    rt.x01 = x01; rt.feat = reinterpret_cast<_Float16*>(feat_rows_f16); rt.geo = reinterpret_cast<_Float16*>(geo_f16);
    rt.sigma = sigmas; rt.rgb = rgbs;
    if (feat_scratch) {  // level-sliced form: encode pass (positions, feature planes, z_vals), then the streaming tail
        TrainOut tr;
        tr.x01 = x01; tr.feat = rt.feat; tr.h32 = nullptr; tr.tile = nullptr;
        const int st = density_uniform_impl(rays_o, rays_d, nears, fars, lin, noise, h_aabb, bound, N, T, table_f16, L, F, h_scales, h_res, h_offsets,
                                            sigma_weights_f16, z_vals, sigmas, geo_f16, feat_scratch, 1u, stream, &tr);
        if (st != NVSF_OK) return st;
    }
    return render_uniform_impl(rays_o, rays_d, nears, fars, lin, noise, h_aabb, bound, N, T, table_f16, L, F, h_scales, h_res, h_offsets,
                               sigma_weights_f16, lidar, head_a_weights_f16, head_b_weights_f16, k_scale, w_thresh, h_bg_color, feat_scratch,
                               z_vals, weights, weights_sum, depth, image, stream, &rt);
}
