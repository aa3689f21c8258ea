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
// Heads + image accumulation, one wave per ray.
//   LIDAR = false : colour net  [SH16(d) | geo15 | 1] (32)  -> 64 -> 64 -> 3 (padded 16)
//   LIDAR = true  : raydrop net and intensity net  [Freq72(d) | geo15 | 1 x 9] (96) -> 64 -> 64 -> 1;
//                   image channels = (raydrop, intensity)  (network_dynamic.py:317)
// rgb = sigmoid(fp32 logits); samples with weight <= w_thresh contribute 0 (renderer_dynamic.py:202,
// network_dynamic.py:297-307, 325-330); image = sum_i w_i rgb_i (+ (1 - ws) * bg for the camera, :236-237).
template <int IN_STEPS>
struct HeadW {
    half8_t w0[kHidTiles][IN_STEPS];
    HiddenLayerW w1;
    OutLayerW w2;
    __device__ __forceinline__ void load(const _Float16* __restrict__ W, int lane) {
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
            for (int s = 0; s < IN_STEPS; ++s) w0[t][s] = load_w_natural(W, 32 * IN_STEPS, t, s, lane);
        w1.load(W + kHidden * 32 * IN_STEPS, lane);
        w2.load(W + kHidden * 32 * IN_STEPS + kHidden * kHidden, lane);
    }
    __device__ __forceinline__ float4_t apply(const half8_t (&xf)[IN_STEPS]) const {
        float4_t acc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < IN_STEPS; ++s) c = mfma16(w0[t][s], xf[s], c);
            acc[t] = c;
        }
        half8_t h[kHidSteps];
        pack_hidden(acc, h);
        w1.apply(h, acc);
        pack_hidden(acc, h);
        return w2.apply(h);
    }
};

__device__ __forceinline__ float sigmoid_f16(float logit_f32) {  // fp32 logit -> fp32 sigmoid (no fp16 rounding of outputs)
    return 1.0f / (1.0f + expf(-logit_f32));
}

template <bool LIDAR>
__global__ __launch_bounds__(kBlock) void k_heads_uniform(const float* __restrict__ weights, const _Float16* __restrict__ geo,
                                                          const float* __restrict__ rays_d, const float* __restrict__ weights_sum,
                                                          const _Float16* __restrict__ w_a, const _Float16* __restrict__ w_b,
                                                          uint32_t N, uint32_t T, float w_thresh, float bg0, float bg1, float bg2,
                                                          int use_bg, float* __restrict__ image) {
    constexpr int IN_STEPS = LIDAR ? 3 : 1;
    constexpr int C = LIDAR ? 2 : 3;
    const uint32_t n = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;
    HeadW<IN_STEPS> net_a;
    net_a.load(w_a, lane);
    HeadW<IN_STEPS> net_b;  // second head only exists for LiDAR
    if constexpr (LIDAR) net_b.load(w_b, lane);

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
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;

    const float* w_row = weights + (size_t)n * T;
    const _Float16* geo_row = geo + (size_t)n * T * 16;
    for (uint32_t base = 0; base < T; base += 16) {
        const uint32_t i = base + sl;
        const bool valid = i < T;
        const float w = valid ? w_row[i] : 0.0f;
        const bool on = valid && (w > w_thresh);
        if (__ballot(on) == 0ull) continue;  // whole tile below the weight threshold
        const half8_t glo = *reinterpret_cast<const half8_t*>(geo_row + (size_t)(valid ? i : 0) * 16);
        const half8_t ghi = *reinterpret_cast<const half8_t*>(geo_row + (size_t)(valid ? i : 0) * 16 + 8);
        if constexpr (!LIDAR) {
            if (g == 2) xf[0] = glo;
            if (g == 3) xf[0] = ghi;
        } else {
            if (g == 1) xf[2] = glo;
            if (g == 2) xf[2] = ghi;
        }
        const float4_t oa = net_a.apply(xf);
        float4_t ob = {0, 0, 0, 0};
        if constexpr (LIDAR) ob = net_b.apply(xf);
        if (g == 0 && on) {
            if constexpr (LIDAR) {
                acc[0] += w * sigmoid_f16(oa[0]);  // raydrop
                acc[1] += w * sigmoid_f16(ob[0]);  // intensity
            } else {
                acc[0] += w * sigmoid_f16(oa[0]);
                acc[1] += w * sigmoid_f16(oa[1]);
                acc[2] += w * sigmoid_f16(oa[2]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = wave_sum(acc[c]);  // only lanes 0..15 hold non-zero partial sums
    if (lane == 0) {
        const float bg[3] = {bg0, bg1, bg2};
        const float rest = use_bg ? 1.0f - weights_sum[n] : 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) image[(size_t)n * C + c] = use_bg ? acc[c] + rest * bg[c] : acc[c];
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

NVSF_API int nvsf_field_density_uniform_fwd(const float* rays_o, const float* rays_d, const float* nears, const float* fars,
                                            const float* lin, const float* noise, const float* h_aabb, float bound, uint32_t N,
                                            uint32_t T, const void* table_f16, uint32_t L, uint32_t F, const float* h_scales,
                                            const uint32_t* h_res, const uint32_t* h_offsets, const void* sigma_weights_f16,
                                            float* z_vals, float* sigmas, void* geo_f16, hipStream_t stream) {
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
    if (F == 2) hipLaunchKernelGGL(k_density_uniform<2>, dim3(blocks), dim3(kBlock), 0, stream, rb, tb, meta, ws, z_vals, sigmas, gp);
    else hipLaunchKernelGGL(k_density_uniform<4>, dim3(blocks), dim3(kBlock), 0, stream, rb, tb, meta, ws, z_vals, sigmas, gp);
    return nvsf_launch_status();
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
        hipLaunchKernelGGL(k_heads_uniform<true>, grid, block, 0, stream, weights, gp, rays_d, weights_sum, wa, wb, N, T, w_thresh, b0,
                           b1, b2, h_bg_color ? 1 : 0, image);
    else
        hipLaunchKernelGGL(k_heads_uniform<false>, grid, block, 0, stream, weights, gp, rays_d, weights_sum, wa, wb, N, T, w_thresh,
                           b0, b1, b2, h_bg_color ? 1 : 0, image);
    return nvsf_launch_status();
}
