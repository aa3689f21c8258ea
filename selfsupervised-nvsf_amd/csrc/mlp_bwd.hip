// Fused MLP backward for gfx950: data gradients AND weight gradients of the bias-free ReLU MLPs of the NVSF hot path
// (width 64, 1 or 2 hidden layers, up to 128 padded inputs, 16 padded outputs) in one kernel.  It replaces, for the
// training path, what tcnn's FullyFusedMLP backward does behind tcnn.Network (network_dynamic.py:125-161, 180-189).
//
// A workgroup (4 waves) takes 64 samples per iteration, one 16-sample tile per wave.
//   data path (per wave, sample on the lane, units in registers -- the layout of mlp_device.h):
//     recompute the forward activations from x, then  dH_l^T = W_{l+1}^T dP_{l+1}^T  with the TRANSPOSED weights as
//     MFMA A operand, ReLU mask from the recomputed activations (same lane, same register slot: no data movement),
//     down to dX (stored) -- activations and gradients never leave registers between layers;
//   weight gradients: dW_l[o][k] = sum over samples dP_l[s][o] A_{l-1}[s][k] has the SAMPLE as the MFMA K dimension,
//     so both operands need "unit on the lane, 8 consecutive samples in registers".  The four waves transpose their
//     tiles through LDS ([unit][64 samples], 2-byte scatter writes, 16-byte fragment reads), then wave w accumulates
//     the output-tile row w of dW_l over the 64 samples (K = 2 x 32) in registers; after the last iteration the
//     accumulators are added to the fp32 gradient buffer with one atomic per element and workgroup.
// Every weight fragment (forward and transposed) lives in LDS in MFMA-operand order ([fragment][lane] x 16 B).
// Gradients travel in fp16 scaled by `grad_scale` (tcnn's loss_scale); dX and dW are unscaled in fp32 on the way out.
#include "mlp_device.h"
#include <stdlib.h>

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kPitch = 72;  // halfs per LDS row of the transposed staging: 64 samples + 8 (16-B aligned, spreads the banks)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0):
// it would drain the next iteration's prefetched global loads (and this iteration's dX stores) at every staging step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int kappa(int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); }

// dL/d(outputs) of a DENSITY network formed while the operands are fetched (nvsf_mlp_bwd_density) instead of read from a [M, 16] matrix
// a kernel of its own wrote (nvsf_sigma_geo_bwd):  column 0 = grad_sigma * clamp(sigma, lo, hi) (trunc_exp's backward, activation.py:
// 16-20), column 1 + j = geo_a[m][j] (+ geo_b[m][j]: the geometry gradients of two heads that share the features, summed here instead
// of by a read-modify-write in the second head's launch).  geo rows: 16 floats, 16-byte aligned, column 15 unused.
struct GoCompose {
    const float* sigma;  // nullptr: grad_out is read as it is
    const float* grad_sigma;
    const float* geo_a;
    const float* geo_b;
    uint32_t stride;
    float lo, hi;
    // The operands of lane group g of sample row `row` (outputs 8 (g & 1) .. + 7) in two steps.  issue(): LOADS ONLY -- the operands of
    // iteration it + 1 are requested while iteration it computes, and anything here that consumed a returned value (the first form
    // summed the two heads' rows and multiplied sigma inside the fetch) puts an s_waitcnt vmcnt(0) right behind the request: the
    // whole memory latency exposed once per iteration, k_mlp_bwd<1,1> 173 -> 336 us in the config-4 step.
    // compose(): the arithmetic, where the values are consumed one iteration later.  Every lane loads all four scalars (column 7 of
    // both rows, grad_sigma, sigma -- lines its 16-byte loads touch anyway) so that no load sits behind a lane-dependent branch.
    __device__ __forceinline__ void issue(size_t row, int g, float (&raw)[8], float (&ex)[12]) const {
        const uint32_t h = (uint32_t)g & 1u;
        const float* ra = geo_a + row * stride;
        const float4* pa = reinterpret_cast<const float4*>(ra) + 2 * h;
        const float4 a = pa[0], b = pa[1];
        raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w; raw[4] = b.x; raw[5] = b.y; raw[6] = b.z; raw[7] = b.w;
        ex[8] = ra[7];
        ex[10] = (grad_sigma ? grad_sigma : sigma)[row];  // (a select of the POINTER: a select of load-or-constant makes the compiler park the constant in scratch)
        ex[11] = sigma[row];
        if (geo_b) {  // uniform
            const float* rb = geo_b + row * stride;
            const float4* pb = reinterpret_cast<const float4*>(rb) + 2 * h;
            const float4 a2 = pb[0], b2 = pb[1];
            ex[0] = a2.x; ex[1] = a2.y; ex[2] = a2.z; ex[3] = a2.w; ex[4] = b2.x; ex[5] = b2.y; ex[6] = b2.z; ex[7] = b2.w;
            ex[9] = rb[7];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ex[j] = 0.0f;
            ex[9] = 0.0f;
        }
    }
    __device__ __forceinline__ void compose(int g, const float (&raw)[8], const float (&ex)[12], float (&go)[8]) const {
        const bool h = (g & 1) != 0;
        float v[7] = {raw[0], raw[1], raw[2], raw[3], raw[4], raw[5], raw[6]};
        float prev = ex[8];
        if (geo_b) {  // uniform
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] += ex[j];
            prev += ex[9];
        }
        if (!h) prev = grad_sigma ? ex[10] * fminf(fmaxf(ex[11], lo), hi) : 0.0f;
        go[0] = prev;
#pragma unroll
        for (int j = 0; j < 7; ++j) go[1 + j] = v[j];
    }
};

// fp32 accumulator tiles (2s, 2s+1) -> B fragment of k-step s, no activation (cf. relu_pack)
__device__ __forceinline__ half8_t plain_pack(float4_t a, float4_t b) {
    const float8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_convertvector(v, half8_t);
}

template <int IN_STEPS, int N_HIDDEN>
struct BwdFrags {
    static constexpr int IN_TILES = 2 * IN_STEPS;                   // 16-wide tiles of the (padded) input
    static constexpr int kF0 = 0;                                   // forward first layer [t][s]
    static constexpr int kF1 = kF0 + 4 * IN_STEPS;                  // forward hidden layer [t][s] (N_HIDDEN == 2)
    static constexpr int kFO = kF1 + (N_HIDDEN == 2 ? 8 : 0);       // forward output layer [s]
    static constexpr int kBO = kFO + 2;                             // W_out^T [t] (K = 16 outputs, zero-padded to 32)
    static constexpr int kB1 = kBO + 4;                             // W_1^T [t][s]
    static constexpr int kB0 = kB1 + (N_HIDDEN == 2 ? 8 : 0);       // W_0^T [input tile][s]
    static constexpr int kCount = kB0 + 2 * IN_TILES;
};

template <int IN_STEPS, int N_HIDDEN>
__device__ __forceinline__ half8_t bwd_fragment(int f, int lane, const _Float16* __restrict__ W, int in_cols) {
    using FR = BwdFrags<IN_STEPS, N_HIDDEN>;
    const int g = lane >> 4, c = lane & 15;
    const _Float16* W0 = W;
    const _Float16* W1 = W + (size_t)kHidden * in_cols;
    const _Float16* WO = W1 + (N_HIDDEN == 2 ? kHidden * kHidden : 0);
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (f < FR::kF1) return load_w_natural(W0, in_cols, f / IN_STEPS, f % IN_STEPS, lane);
    if (f < FR::kFO) { const int r = f - FR::kF1; return load_w_chained(W1, r >> 1, r & 1, lane); }
    if (f < FR::kBO) return load_w_chained(WO, 0, f - FR::kFO, lane);
    if (f < FR::kB1) {  // A[m = hidden unit 16t + c][k = output 8g + j]
        const int t = f - FR::kBO;
        if (g < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = WO[(size_t)(8 * g + j) * kHidden + 16 * t + c];
        }
        return v;
    }
    if (f < FR::kB0) {  // A[m = hidden unit 16t + c][k <-> hidden unit kappa(s, g, j)] = W1[kappa][16t + c]
        const int r = f - FR::kB1, t = r >> 1, s = r & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W1[(size_t)kappa(s, g, j) * kHidden + 16 * t + c];
        return v;
    }
    {  // A[m = input 16ti + c][k <-> hidden unit kappa(s, g, j)] = W0[kappa][16ti + c]
        const int r = f - FR::kB0, ti = r >> 1, s = r & 1;
        if (16 * ti + c < in_cols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W0[(size_t)kappa(s, g, j) * in_cols + 16 * ti + c];
        }
        return v;
    }
}

// ReLU mask: accumulator tile t, register r of this lane  <->  element (t & 1) * 4 + r of B fragment h[t >> 1]
__device__ __forceinline__ void relu_mask(float4_t (&gacc)[kHidTiles], const half8_t (&h)[kHidSteps]) {
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (!(h[t >> 1][(t & 1) * 4 + r] > (_Float16)0.0f)) gacc[t][r] = 0.0f;
}

template <int IN_STEPS, int N_HIDDEN, bool X_F16, bool FAST, bool COMPOSE>
__global__ __launch_bounds__(kBlock) void k_mlp_bwd(const void* __restrict__ x, uint32_t M, uint32_t n_in, uint32_t x_stride,
                                                    const _Float16* __restrict__ weights, uint32_t in_cols,
                                                    const float* __restrict__ grad_out, uint32_t n_out, uint32_t go_stride, float grad_scale,
                                                    float* __restrict__ grad_x, uint32_t gx_stride, float* __restrict__ grad_w, int vec_ok,
                                                    uint32_t gx_col0, int gx_accumulate, int go_vec, XPrefix pre, GoCompose gc) {
    using FR = BwdFrags<IN_STEPS, N_HIDDEN>;
    constexpr int IN_TILES = FR::IN_TILES;
    __shared__ half8_t s_frag[FR::kCount * kWave];
    __shared__ _Float16 s_g[kHidden * kPitch];           // dP^T  [unit][sample]
    __shared__ _Float16 s_a[(IN_STEPS > 2 ? 32 * IN_STEPS : 64) * kPitch];  // A^T [unit][sample]: hidden (64) or padded input units
    const int lane = lane_id(), g = lane >> 4, c = lane & 15, w = (int)(threadIdx.x >> 6);
    for (int f = w; f < FR::kCount; f += kWavesPerBlock) s_frag[f * kWave + lane] = bwd_fragment<IN_STEPS, N_HIDDEN>(f, lane, weights, (int)in_cols);
    __syncthreads();
    const half8_t* frag = s_frag + lane;
    const float inv_scale = 1.0f / grad_scale;

    // weight-gradient accumulators of this wave: output-tile row w of dW0 and dW1, k-tile w of dW_out
    float4_t dw0[IN_TILES], dw1[kHidTiles], dwo;
#pragma unroll
    for (int i = 0; i < IN_TILES; ++i) dw0[i] = float4_t{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < kHidTiles; ++i) dw1[i] = float4_t{0, 0, 0, 0};
    dwo = float4_t{0, 0, 0, 0};

    const uint32_t n_tiles = (M + 15) / 16;
    const uint32_t tiles_per_iter = gridDim.x * kWavesPerBlock;
    const uint32_t n_iters = (n_tiles + tiles_per_iter - 1) / tiles_per_iter;
    const int col = 16 * w + c;  // this lane's sample column in the staging arrays

    // transposed fragment (unit row0 + c, samples 32*ks + 8g .. +7) of a staging array
    auto t_frag = [&](const _Float16* arr, int row0, int ks) {
        return *reinterpret_cast<const half8_t*>(arr + (size_t)(row0 + c) * kPitch + 32 * ks + 8 * g);
    };

    // operands of one iteration: the x fragments and the raw output gradients of this lane's sample.  Those of iteration
    // it + 1 are requested before iteration it computes (two resident workgroups per CU cannot hide a dependent global
    // load per iteration by themselves).
    XTail tail;
    if constexpr (FAST) tail.init(32 * (IN_STEPS - 1) + 8 * g, (int)n_in, (int)in_cols);
    struct Operands {
        half8_t xf[IN_STEPS];
        float go[8];
        float ex[COMPOSE ? 12 : 1];  // COMPOSE: the second head's row and the four scalars, as loaded
        uint32_t m;
        bool valid;
    };
    auto fetch = [&](uint32_t it, Operands& op) __attribute__((always_inline)) {
        const uint32_t tile = (it * gridDim.x + blockIdx.x) * kWavesPerBlock + (uint32_t)w;
        op.m = tile * 16 + (uint32_t)c;
        op.valid = tile < n_tiles && op.m < M;
        const size_t row = op.valid ? op.m : (M - 1);
        const uint32_t tile_u = __builtin_amdgcn_readfirstlane(tile);
        const _Float16* prow = pre.a ? pre.row_of((tile_u < n_tiles && tile_u * 16u < M) ? tile_u * 16u : M - 1u) : nullptr;
        issue_x_row<IN_STEPS, X_F16, FAST>(op.xf, x, row, x_stride, g, (int)n_in, (int)in_cols, vec_ok != 0, tail, prow, pre.split);
        const float* go_row = grad_out + row * go_stride;
        if constexpr (COMPOSE) {  // density network: the logit gradient composed from (grad_sigma, sigma, geometry gradient rows)
            gc.issue(row, g, op.go, op.ex);
        } else if (go_vec) {  // 16 outputs, 16-byte aligned rows: two 16-byte loads for the lanes that hold outputs (g < 2)
            const float4* p = reinterpret_cast<const float4*>(go_row) + 2 * (g & 1);
            const float4 a = p[0], b = p[1];
            op.go[0] = a.x; op.go[1] = a.y; op.go[2] = a.z; op.go[3] = a.w;
            op.go[4] = b.x; op.go[5] = b.y; op.go[6] = b.z; op.go[7] = b.w;
        } else {  // element loads with the column clamped into the row, issued back to back
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t o = 8 * (g & 1) + j;
                op.go[j] = go_row[o < n_out ? o : n_out - 1];
            }
        }
    };
    // the wide variants have no registers to spare for a second set of operands at two waves per SIMD: they fetch in place
    constexpr bool kPrefetch = FAST || IN_STEPS <= 2;
    Operands next;
    if (kPrefetch && n_iters) fetch(0, next);

    for (uint32_t it = 0; it < n_iters; ++it) {
        if (!kPrefetch) fetch(it, next);
        const Operands cur = next;
        if (kPrefetch && it + 1 < n_iters) fetch(it + 1, next);
        const uint32_t m = cur.m;
        const bool valid = cur.valid;
        half8_t xf[IN_STEPS];
#pragma unroll
        for (int s = 0; s < IN_STEPS; ++s) xf[s] = cur.xf[s];
        if constexpr (FAST) xf[IN_STEPS - 1] = tail.apply(xf[IN_STEPS - 1]);
        // ---- forward recompute
        float4_t acc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t a = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < IN_STEPS; ++s) a = mfma16(frag[(FR::kF0 + t * IN_STEPS + s) * kWave], xf[s], a);
            acc[t] = a;
        }
        half8_t h0[kHidSteps], h1[kHidSteps];
        pack_hidden(acc, h0);
        if constexpr (N_HIDDEN == 2) {
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) {
                float4_t a = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < kHidSteps; ++s) a = mfma16(frag[(FR::kF1 + 2 * t + s) * kWave], h0[s], a);
                acc[t] = a;
            }
            pack_hidden(acc, h1);
        }
        const half8_t(&h_last)[kHidSteps] = N_HIDDEN == 2 ? h1 : h0;
        // ---- output gradient -> B fragment (natural order of the 16 outputs, zero beyond n_out, scaled)
        half8_t go;
        float gov[8];
        if constexpr (COMPOSE) {
            gc.compose(g, cur.go, cur.ex, gov);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gov[j] = cur.go[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) go[j] = (valid && g < 2 && 8u * g + j < n_out) ? (_Float16)(gov[j] * grad_scale) : (_Float16)0.0f;
        // ---- data path backward
        float4_t gacc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) gacc[t] = mfma16(frag[(FR::kBO + t) * kWave], go, float4_t{0, 0, 0, 0});
        relu_mask(gacc, h_last);
        half8_t gp_last[kHidSteps], gp0[kHidSteps];
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s) gp_last[s] = plain_pack(gacc[2 * s], gacc[2 * s + 1]);
        if constexpr (N_HIDDEN == 2) {
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t) {
                float4_t a = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < kHidSteps; ++s) a = mfma16(frag[(FR::kB1 + 2 * t + s) * kWave], gp_last[s], a);
                gacc[t] = a;
            }
            relu_mask(gacc, h0);
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) gp0[s] = plain_pack(gacc[2 * s], gacc[2 * s + 1]);
        } else {
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s) gp0[s] = gp_last[s];
        }
        if (grad_x) {
#pragma unroll
            for (int ti = 0; ti < IN_TILES; ++ti) {
                if (16u * ti < n_in && 16u * ti + 16u > gx_col0) {  // input tiles without a requested column are skipped
                    float4_t a = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < kHidSteps; ++s) a = mfma16(frag[(FR::kB0 + 2 * ti + s) * kWave], gp0[s], a);
                    const uint32_t k0 = 16 * ti + 4 * g;
                    float* row_x = grad_x + (size_t)m * gx_stride;
                    if constexpr (FAST) {  // 16-byte aligned window: this lane's four columns as one store (the last group may
                                           // reach into the row's alignment padding)
                        const uint32_t blk = ((uint32_t)gx_accumulate >> 8) & 0xFFu;  // 0: rows; 2 / 4: column blocks (below)
                        if (blk == 0u) {
                            if (valid && k0 >= gx_col0 && k0 < n_in) {
                                float4_t* p = reinterpret_cast<float4_t*>(row_x + (k0 - gx_col0));
                                float4_t v = a * inv_scale;
                                if (gx_accumulate & 1) v += *p;
                                *p = v;
                            }
                        } else if (valid && k0 < n_in) {
                            // block-major dX: [n_in / blk][M][blk] -- the gradient of a hash grid's features, level by level, so that the
                            // table scatter reads a level's column as one stream (a row-major [M, 32] costs it a 128-byte line per 8 bytes
                            // and level: hashgrid.hip).  16 consecutive samples of a level are 16 * blk * 4 contiguous bytes.
                            float4_t v = a * inv_scale;
                            if (blk == 4u) {
                                float4_t* p = reinterpret_cast<float4_t*>(grad_x + ((size_t)(k0 >> 2) * M + m) * 4);
                                if (gx_accumulate & 1) v += *p;
                                *p = v;
                            } else {
                                float2* p0 = reinterpret_cast<float2*>(grad_x + ((size_t)(k0 >> 1) * M + m) * 2);
                                float2* p1 = reinterpret_cast<float2*>(grad_x + ((size_t)((k0 >> 1) + 1u) * M + m) * 2);
                                float2 v0 = make_float2(v[0], v[1]), v1 = make_float2(v[2], v[3]);
                                if (gx_accumulate & 1) { v0.x += p0->x; v0.y += p0->y; v1.x += p1->x; v1.y += p1->y; }
                                *p0 = v0;
                                *p1 = v1;
                            }
                        }
                    } else if (valid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t k = k0 + r;
                            if (k < n_in && k >= gx_col0) {
                                const float v = a[r] * inv_scale;
                                row_x[k - gx_col0] = (gx_accumulate & 1) ? row_x[k - gx_col0] + v : v;
                            }
                        }
                    }
                }
            }
        }
        // ---- weight gradients, layer by layer through the transposed staging
        // (1) output layer: dW_out[o][k] = sum_s dOut[s][o] * h_last[s][k]
        if (g < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s_g[(size_t)(8 * g + j) * kPitch + col] = go[j];
        }
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) s_a[(size_t)kappa(s, g, j) * kPitch + col] = h_last[s][j];
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dwo = mfma16(t_frag(s_g, 0, ks), t_frag(s_a, 16 * w, ks), dwo);
        lds_barrier();
        // (2) hidden layer: dW1[o][k] = sum_s dP1[s][o] * h0[s][k]
        if constexpr (N_HIDDEN == 2) {
#pragma unroll
            for (int s = 0; s < kHidSteps; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s_g[(size_t)kappa(s, g, j) * kPitch + col] = gp_last[s][j];
                    s_a[(size_t)kappa(s, g, j) * kPitch + col] = h0[s][j];
                }
            lds_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8_t a = t_frag(s_g, 16 * w, ks);
#pragma unroll
                for (int tk = 0; tk < kHidTiles; ++tk) dw1[tk] = mfma16(a, t_frag(s_a, 16 * tk, ks), dw1[tk]);
            }
            lds_barrier();
        }
        // (3) first layer: dW0[o][k] = sum_s dP0[s][o] * x[s][k]
#pragma unroll
        for (int s = 0; s < kHidSteps; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) s_g[(size_t)kappa(s, g, j) * kPitch + col] = gp0[s][j];
#pragma unroll
        for (int s = 0; s < IN_STEPS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) s_a[(size_t)(32 * s + 8 * g + j) * kPitch + col] = xf[s][j];
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8_t a = t_frag(s_g, 16 * w, ks);
#pragma unroll
            for (int tk = 0; tk < IN_TILES; ++tk) dw0[tk] = mfma16(a, t_frag(s_a, 16 * tk, ks), dw0[tk]);
        }
        lds_barrier();
    }
    // ---- flush: accumulator element (row 4g + r, column c) of tile (to, tk)
    float* gw0 = grad_w;
    float* gw1 = grad_w + (size_t)kHidden * in_cols;
    float* gwo = gw1 + (N_HIDDEN == 2 ? kHidden * kHidden : 0);
#pragma unroll
    for (int tk = 0; tk < IN_TILES; ++tk) {
        const uint32_t k = 16 * tk + c;
        if (k < in_cols) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(gw0 + (size_t)(16 * w + 4 * g + r) * in_cols + k, dw0[tk][r] * inv_scale);
        }
    }
    if constexpr (N_HIDDEN == 2) {
#pragma unroll
        for (int tk = 0; tk < kHidTiles; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(gw1 + (size_t)(16 * w + 4 * g + r) * kHidden + 16 * tk + c, dw1[tk][r] * inv_scale);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(gwo + (size_t)(4 * g + r) * kHidden + 16 * w + c, dwo[r] * inv_scale);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-independent form (production): no LDS staging, no barrier inside the loop.
//
// The staged form above transposes every dW operand through LDS with 2-byte scatter writes and pays six barriers per 64 samples;
// its counters show neither the MFMA pipe (11-15 %) nor HBM as the bound, but LDS (weight fragments re-read per 16-sample tile,
// the transposing writes).  Here a WAVE owns two 16-sample tiles per iteration and everything it needs:
//   * transposition on the matrix core: the data path keeps "sample on the lane, units in registers" (mlp_device.h); the weight
//     gradient dW[o][k] = sum_s dP[s][o] A[s][k] has the SAMPLE as the MFMA K dimension and needs "unit on the lane, samples in
//     registers".  Swapping the A and B operands of an MFMA transposes its result, so a 16 x 16 block of any fp16 fragment V is
//     transposed by ONE mfma(A = V, B = selection matrix of 0 / 1): exact (products by 1, sums with 0), 16 cycles, no LDS.  The two
//     tiles give the two halves of a K = 32 operand: k-block g, element j <-> sample 4g + (j & 3) of tile (j >> 2); both operands
//     of a dW product use that same mapping, so the sum over k is the sum over the 32 samples.
//   * every weight fragment read from LDS feeds two tiles (as k_render_tail2 does): half the LDS reads per sample of the staged form.
//   * the whole dW of the network is accumulated by each wave (all output-tile rows: up to 52 accumulator tiles = 208 registers),
//     in ACCUMULATION registers (AGPRs, inline-asm MFMA: the library is built with -amdgpu-mfma-vgpr-form, which would put them in
//     the 256 architected registers), one wave per SIMD (amdgpu_waves_per_eu 1): 512 registers per lane.
//   * after the loop the four waves' accumulators are summed in LDS and leave with ONE float atomic per weight and workgroup
//     (256 workgroups: 2.9 M atomics for the LiDAR head, as contiguous 1-KB instructions).
// MFMA count per 16-sample tile of the 87-64-64-1 head: 20 recompute + 16 data path + 23 transposes + 22 dW = 81 (58 staged).
__device__ __forceinline__ void mfma16_agpr(float4_t& acc, half8_t a, half8_t b) {
    // s_nop 1: a VALU write of an MFMA source operand needs two wait states before the MFMA reads it (gfx90a+: the compiler inserts
    // them for its own MFMAs -- GCNHazardRecognizer, "legacy VALU writes VGPR" -- and cannot see into this statement; without them the
    // v_cvt_pk that packs a transposed operand right in front of the MFMA is read half-written: measured, wrong sums)
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int IN_STEPS, int N_HIDDEN, bool X_F16, bool FAST, bool COMPOSE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mlp_bwd_wave(
    const void* __restrict__ x, uint32_t M, uint32_t n_in, uint32_t x_stride, const _Float16* __restrict__ weights, uint32_t in_cols,
    const float* __restrict__ grad_out, uint32_t n_out, uint32_t go_stride, float grad_scale, float* __restrict__ grad_x, uint32_t gx_stride,
    float* __restrict__ grad_w, int vec_ok, uint32_t gx_col0, int gx_accumulate, int go_vec, XPrefix pre, GoCompose gc) {
    using FR = BwdFrags<IN_STEPS, N_HIDDEN>;
    constexpr int IN_TILES = FR::IN_TILES;
    constexpr int kDwMax = kHidden * 32 * IN_STEPS + (N_HIDDEN == 2 ? kHidden * kHidden : 0) + 16 * kHidden;
    __shared__ half8_t s_frag[FR::kCount * kWave];
    __shared__ float s_dw[kDwMax];
    const int lane = lane_id(), g = lane >> 4, c = lane & 15, w = (int)(threadIdx.x >> 6);
    for (int f = w; f < FR::kCount; f += kWavesPerBlock) s_frag[f * kWave + lane] = bwd_fragment<IN_STEPS, N_HIDDEN>(f, lane, weights, (int)in_cols);
    __syncthreads();
    const half8_t* frag = s_frag + lane;
    const float inv_scale = 1.0f / grad_scale;

    // selection matrices (B operands): P[h] picks accumulator tile 2s + h out of a kappa-ordered k-step, Q[h] the 16 columns 16h .. 16h + 15
    // out of a naturally ordered one
    half8_t P[2], Q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            P[h][j] = ((j >> 2) == h && c == 4 * g + (j & 3)) ? (_Float16)1.0f : (_Float16)0.0f;
            Q[h][j] = ((g >> 1) == h && c == 8 * (g & 1) + j) ? (_Float16)1.0f : (_Float16)0.0f;
        }
    // 16 x 16 block of fragment v, transposed: lane (unit c, g) gets the samples 4g .. 4g + 3
    auto tr = [&](half8_t v, half8_t sel) { return mfma16(v, sel, float4_t{0, 0, 0, 0}); };

    float4_t dw0[kHidTiles][IN_TILES], dw1[kHidTiles][kHidTiles], dwo[kHidTiles];
#pragma unroll
    for (int to = 0; to < kHidTiles; ++to) {
#pragma unroll
        for (int i = 0; i < IN_TILES; ++i) dw0[to][i] = float4_t{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < kHidTiles; ++i) dw1[to][i] = float4_t{0, 0, 0, 0};
        dwo[to] = float4_t{0, 0, 0, 0};
    }

    const uint32_t n_tiles = (M + 15) / 16;
    const uint32_t n_pairs = (n_tiles + 1) / 2;
    const uint32_t pairs_per_iter = gridDim.x * kWavesPerBlock;
    const uint32_t n_iters = (n_pairs + pairs_per_iter - 1) / pairs_per_iter;

    XTail tail;
    if constexpr (FAST) tail.init(32 * (IN_STEPS - 1) + 8 * g, (int)n_in, (int)in_cols);
    struct Operands {
        half8_t xf[IN_STEPS];
        float go[8];
        float ex[COMPOSE ? 12 : 1];  // COMPOSE: the second head's row and the four scalars, as loaded
        uint32_t m;
        bool valid;
    };
    auto fetch = [&](uint32_t it, int u, Operands& op) __attribute__((always_inline)) {
        const uint32_t tile = 2u * ((it * gridDim.x + blockIdx.x) * kWavesPerBlock + (uint32_t)w) + (uint32_t)u;
        op.m = tile * 16 + (uint32_t)c;
        op.valid = tile < n_tiles && op.m < M;
        const size_t row = op.valid ? op.m : (M - 1);
        const uint32_t tile_u = __builtin_amdgcn_readfirstlane(tile);
        const _Float16* prow = pre.a ? pre.row_of((tile_u < n_tiles && tile_u * 16u < M) ? tile_u * 16u : M - 1u) : nullptr;
        issue_x_row<IN_STEPS, X_F16, FAST>(op.xf, x, row, x_stride, g, (int)n_in, (int)in_cols, vec_ok != 0, tail, prow, pre.split);
        const float* go_row = grad_out + row * go_stride;
        if constexpr (COMPOSE) {
            gc.issue(row, g, op.go, op.ex);
        } else if (go_vec) {
            const float4* p = reinterpret_cast<const float4*>(go_row) + 2 * (g & 1);
            const float4 a = p[0], b = p[1];
            op.go[0] = a.x; op.go[1] = a.y; op.go[2] = a.z; op.go[3] = a.w;
            op.go[4] = b.x; op.go[5] = b.y; op.go[6] = b.z; op.go[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t o = 8 * (g & 1) + j;
                op.go[j] = go_row[o < n_out ? o : n_out - 1];
            }
        }
    };
    constexpr bool kPrefetch = FAST && X_F16;  // the next pair's operands are requested before this pair is worked on
    Operands next[2];
    if (kPrefetch && n_iters) { fetch(0, 0, next[0]); fetch(0, 1, next[1]); }

    for (uint32_t it = 0; it < n_iters; ++it) {
        if (!kPrefetch) { fetch(it, 0, next[0]); fetch(it, 1, next[1]); }
        half8_t xf[2][IN_STEPS];
        float gor[2][8];
        uint32_t m[2];
        bool valid[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int s = 0; s < IN_STEPS; ++s) xf[u][s] = next[u].xf[s];
#pragma unroll
            for (int j = 0; j < 8; ++j) gor[u][j] = next[u].go[j];
            if constexpr (COMPOSE) gc.compose(g, next[u].go, next[u].ex, gor[u]);
            m[u] = next[u].m;
            valid[u] = next[u].valid;
            if constexpr (FAST) xf[u][IN_STEPS - 1] = tail.apply(xf[u][IN_STEPS - 1]);
        }
        if (kPrefetch && it + 1 < n_iters) { fetch(it + 1, 0, next[0]); fetch(it + 1, 1, next[1]); }
        // ---- forward recompute (sample on the lane)
        half8_t h0[2][kHidSteps], h1[2][kHidSteps];
        {
            float4_t acc[2][kHidTiles];
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float4_t a = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < IN_STEPS; ++s) a = mfma16(frag[(FR::kF0 + t * IN_STEPS + s) * kWave], xf[u][s], a);
                    acc[u][t] = a;
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) pack_hidden(acc[u], h0[u]);
            if constexpr (N_HIDDEN == 2) {
#pragma unroll
                for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float4_t a = {0, 0, 0, 0};
#pragma unroll
                        for (int s = 0; s < kHidSteps; ++s) a = mfma16(frag[(FR::kF1 + 2 * t + s) * kWave], h0[u][s], a);
                        acc[u][t] = a;
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) pack_hidden(acc[u], h1[u]);
            }
        }
        const half8_t(&h_last)[2][kHidSteps] = N_HIDDEN == 2 ? h1 : h0;
        // ---- output gradient -> B fragment (natural order of the 16 outputs, zero beyond n_out, scaled)
        half8_t go[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) go[u][j] = (valid[u] && g < 2 && 8u * g + j < n_out) ? (_Float16)(gor[u][j] * grad_scale) : (_Float16)0.0f;
        // ---- output layer: data path, then its weight gradient  dW_out[o][k] = sum_s dOut[s][o] h_last[s][k]
        half8_t gp_last[2][kHidSteps], gp0[2][kHidSteps];
        {
            float4_t gacc[2][kHidTiles];
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) gacc[u][t] = mfma16(frag[(FR::kBO + t) * kWave], go[u], float4_t{0, 0, 0, 0});
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                relu_mask(gacc[u], h_last[u]);
#pragma unroll
                for (int s = 0; s < kHidSteps; ++s) gp_last[u][s] = plain_pack(gacc[u][2 * s], gacc[u][2 * s + 1]);
            }
        }
        {
            const half8_t go_n = plain_pack(tr(go[0], Q[0]), tr(go[1], Q[0]));
#pragma unroll
            for (int tk = 0; tk < kHidTiles; ++tk)
                mfma16_agpr(dwo[tk], go_n, plain_pack(tr(h_last[0][tk >> 1], P[tk & 1]), tr(h_last[1][tk >> 1], P[tk & 1])));
        }
        // ---- hidden layer: dP0 and  dW1[o][k] = sum_s dP1[s][o] h0[s][k]
        if constexpr (N_HIDDEN == 2) {
            float4_t gacc[2][kHidTiles];
#pragma unroll
            for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float4_t a = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < kHidSteps; ++s) a = mfma16(frag[(FR::kB1 + 2 * t + s) * kWave], gp_last[u][s], a);
                    gacc[u][t] = a;
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                relu_mask(gacc[u], h0[u]);
#pragma unroll
                for (int s = 0; s < kHidSteps; ++s) gp0[u][s] = plain_pack(gacc[u][2 * s], gacc[u][2 * s + 1]);
            }
            half8_t h0n[kHidTiles];
#pragma unroll
            for (int tk = 0; tk < kHidTiles; ++tk) h0n[tk] = plain_pack(tr(h0[0][tk >> 1], P[tk & 1]), tr(h0[1][tk >> 1], P[tk & 1]));
#pragma unroll
            for (int to = 0; to < kHidTiles; ++to) {
                const half8_t gn = plain_pack(tr(gp_last[0][to >> 1], P[to & 1]), tr(gp_last[1][to >> 1], P[to & 1]));
#pragma unroll
                for (int tk = 0; tk < kHidTiles; ++tk) mfma16_agpr(dw1[to][tk], gn, h0n[tk]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < kHidSteps; ++s) gp0[u][s] = gp_last[u][s];
        }
        // ---- input gradient
        if (grad_x) {
#pragma unroll
            for (int ti = 0; ti < IN_TILES; ++ti) {
                if (16u * ti < n_in && 16u * ti + 16u > gx_col0) {  // input tiles without a requested column are skipped
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float4_t a = {0, 0, 0, 0};
#pragma unroll
                        for (int s = 0; s < kHidSteps; ++s) a = mfma16(frag[(FR::kB0 + 2 * ti + s) * kWave], gp0[u][s], a);
                        const uint32_t k0 = 16 * ti + 4 * g;
                        float* row_x = grad_x + (size_t)m[u] * gx_stride;
                        if constexpr (FAST) {
                            const uint32_t blk = ((uint32_t)gx_accumulate >> 8) & 0xFFu;  // 0: rows; 2 / 4: column blocks
                            if (blk == 0u) {
                                if (valid[u] && k0 >= gx_col0 && k0 < n_in) {
                                    float4_t* p = reinterpret_cast<float4_t*>(row_x + (k0 - gx_col0));
                                    float4_t v = a * inv_scale;
                                    if (gx_accumulate & 1) v += *p;
                                    *p = v;
                                }
                            } else if (valid[u] && k0 < n_in) {
                                float4_t v = a * inv_scale;
                                if (blk == 4u) {
                                    float4_t* p = reinterpret_cast<float4_t*>(grad_x + ((size_t)(k0 >> 2) * M + m[u]) * 4);
                                    if (gx_accumulate & 1) v += *p;
                                    *p = v;
                                } else {
                                    float2* p0 = reinterpret_cast<float2*>(grad_x + ((size_t)(k0 >> 1) * M + m[u]) * 2);
                                    float2* p1 = reinterpret_cast<float2*>(grad_x + ((size_t)((k0 >> 1) + 1u) * M + m[u]) * 2);
                                    float2 v0 = make_float2(v[0], v[1]), v1 = make_float2(v[2], v[3]);
                                    if (gx_accumulate & 1) { v0.x += p0->x; v0.y += p0->y; v1.x += p1->x; v1.y += p1->y; }
                                    *p0 = v0;
                                    *p1 = v1;
                                }
                            }
                        } else if (valid[u]) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const uint32_t k = k0 + r;
                                if (k < n_in && k >= gx_col0) {
                                    const float v = a[r] * inv_scale;
                                    row_x[k - gx_col0] = (gx_accumulate & 1) ? row_x[k - gx_col0] + v : v;
                                }
                            }
                        }
                    }
                }
            }
        }
        // ---- first layer:  dW0[o][k] = sum_s dP0[s][o] x[s][k]
        {
            half8_t gn[kHidTiles];
#pragma unroll
            for (int to = 0; to < kHidTiles; ++to) gn[to] = plain_pack(tr(gp0[0][to >> 1], P[to & 1]), tr(gp0[1][to >> 1], P[to & 1]));
#pragma unroll
            for (int ti = 0; ti < IN_TILES; ++ti) {
                const half8_t xn = plain_pack(tr(xf[0][ti >> 1], Q[ti & 1]), tr(xf[1][ti >> 1], Q[ti & 1]));
#pragma unroll
                for (int to = 0; to < kHidTiles; ++to) mfma16_agpr(dw0[to][ti], gn[to], xn);
            }
        }
    }
    // ---- flush: the four waves' sums are added up in LDS, then one atomic per weight and workgroup.
    // accumulator element (row 4g + r, column c) of tile (to, tk)
    const uint32_t off1 = (uint32_t)kHidden * in_cols, offo = off1 + (N_HIDDEN == 2 ? kHidden * kHidden : 0), n_w = offo + 16 * kHidden;
    for (int ww = 0; ww < kWavesPerBlock; ++ww) {
        if (w == ww) {
            auto put = [&](uint32_t idx, float v) { s_dw[idx] = ww == 0 ? v : s_dw[idx] + v; };
#pragma unroll
            for (int to = 0; to < kHidTiles; ++to) {
#pragma unroll
                for (int ti = 0; ti < IN_TILES; ++ti) {
                    const uint32_t k = 16 * ti + c;
                    if (k < in_cols) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) put((uint32_t)(16 * to + 4 * g + r) * in_cols + k, dw0[to][ti][r]);
                    }
                }
                if constexpr (N_HIDDEN == 2) {
#pragma unroll
                    for (int tk = 0; tk < kHidTiles; ++tk)
#pragma unroll
                        for (int r = 0; r < 4; ++r) put(off1 + (uint32_t)(16 * to + 4 * g + r) * kHidden + 16 * tk + c, dw1[to][tk][r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) put(offo + (uint32_t)(4 * g + r) * kHidden + 16 * to + c, dwo[to][r]);
            }
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n_w; i += kBlock) {
        const float v = s_dw[i] * inv_scale;
        if (v != 0.0f) atomicAdd(grad_w + i, v);
    }
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

static int mlp_bwd_impl(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                        uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_out, uint32_t n_out,
                        uint32_t go_stride, float grad_scale, float* grad_x, uint32_t gx_stride, float* grad_weights_f32,
                        uint32_t gx_col0, int gx_accumulate, XPrefix pre, GoCompose gc, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && weights_f16 && (grad_out || gc.sigma) && grad_weights_f32);
    REQUIRE(n_in >= 1 && n_in <= in_cols && in_cols % 16 == 0 && x_stride >= n_in - pre.split);
    REQUIRE(n_out >= 1 && n_out <= 16 && go_stride >= n_out && grad_scale > 0.0f);
    const uint32_t gx_blk = ((uint32_t)gx_accumulate >> 8) & 0xFFu;
    REQUIRE(!grad_x || gx_blk || (gx_col0 < n_in && gx_stride >= n_in - gx_col0));
    REQUIRE(!gx_blk || (grad_x && (gx_blk == 2u || gx_blk == 4u) && gx_col0 == 0 && n_in % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15u) == 0));
    REQUIRE((reinterpret_cast<uintptr_t>(weights_f16) & 15u) == 0);
    if (hidden != (uint32_t)kHidden || out_cols != 16 || n_hidden < 1 || n_hidden > 2 || in_cols > 128) return NVSF_ERR_UNSUPPORTED;
    const int in_steps = (int)((in_cols + 31) / 32);
    const size_t esz = x_is_f16 ? 2 : 4;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && ((x_stride * esz) % 16 == 0);
    const int go_vec = n_out == 16 && go_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15u) == 0;
    const uint32_t n_tiles = (M + 15) / 16;
    uint32_t blocks = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    // The staged kernel walks its tiles grid-stride.  A grid of exactly the resident capacity (256 CUs x 3 workgroups = 768, the figure of
    // rounds 2-5: alone 0.180 ms, equal at 512 and 768) finishes together only when it has the device to itself; inside the training step
    // the table scatter of the other stream holds slots, some workgroups start late and the launch lasts until the last of them is through
    // its full share (0.46 ms in the step).  Four times the capacity leaves the balancing to the dispatcher: alone 0.186 ms (more weight
    // staging and flush atomics), config-4 step 4.97-5.01 against 5.04-5.08 ms (same box, three alternating runs; 1536: 5.01-5.04,
    // 6144: 5.08-5.12, 12288: 5.27-5.29).  The wave-independent kernel (one workgroup per CU, the whole register file) gains nothing
    // from more workgroups: 5.32 / 5.50 ms at 2 x / 4 x.
    const uint32_t cap = 3072u;
    if (blocks > cap) blocks = cap;
    const _Float16* w = reinterpret_cast<const _Float16*>(weights_f16);
    // FAST: 16-byte loads of x (mlp_device.h) and 16-byte stores of dX (window and rows aligned to four floats, rows wide
    // enough for the last group of four)
    const bool gx_vec = !grad_x || gx_blk || (gx_col0 % 4 == 0 && gx_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15u) == 0 &&
                                              (n_in - gx_col0 + 3u) / 4u * 4u <= gx_stride);
    const bool fast = x_rows_fast(n_in - pre.split, x_stride, vec_ok) && gx_vec;
    REQUIRE(!gx_blk || fast);  // column blocks: the aligned-row form of the kernel only
    if (pre.a) {  // shared-prefix rows (mlp_device.h: XPrefix): aligned fp16 only, whole 8-column groups, a tile inside one group
        REQUIRE(x_is_f16 && fast && pre.split % 8 == 0 && pre.split < n_in && pre.a_stride >= pre.split && pre.a_stride % 8 == 0);
        REQUIRE(pre.rows_per_a >= 16 && pre.rows_per_a % 16 == 0 && (reinterpret_cast<uintptr_t>(pre.a) & 15u) == 0);
    }
    // production: the wave-independent kernel, one workgroup per compute unit; nvsf_test_variant("mlp_bwd", 1): the LDS-staged kernel
    // The 32-wide one-hidden-layer network (the density MLP: 256 B of rows per sample for 18 432 FLOP = 72 FLOP/B, far below the ridge of
    // 312) is bound by its row traffic, which one wave per SIMD cannot keep in flight (measured 0.30 ms against 0.19 ms at 3.1 M rows):
    // that shape stays on the staged kernel (three workgroups per CU).  nvsf_test_variant("mlp_bwd", 1 / 2) forces staged / wave.
    const int forced = nvsf_variant(kVarMlpBwd);
    const bool staged = forced == 1 || (forced == 0 && in_steps == 1 && n_hidden == 1);
    if (!staged) {
        const uint32_t n_pairs = (n_tiles + 1) / 2, cus = (uint32_t)nvsf_cu_count();
        blocks = (n_pairs + kWavesPerBlock - 1) / kWavesPerBlock;
        if (blocks > cus) blocks = cus;
    }
#define LAUNCH(S, H, XF, FA, CO)                                                                                                           \
    do {                                                                                                                                   \
        if (staged)                                                                                                                        \
            hipLaunchKernelGGL((k_mlp_bwd<S, H, XF, FA, CO>), dim3(blocks), dim3(kBlock), 0, stream, x, M, n_in, x_stride, w, in_cols,    \
                               grad_out, n_out, go_stride, grad_scale, grad_x, gx_stride, grad_weights_f32, vec_ok, gx_col0,              \
                               gx_accumulate, go_vec, pre, gc);                                                                            \
        else                                                                                                                               \
            hipLaunchKernelGGL((k_mlp_bwd_wave<S, H, XF, FA, CO>), dim3(blocks), dim3(kBlock), 0, stream, x, M, n_in, x_stride, w,        \
                               in_cols, grad_out, n_out, go_stride, grad_scale, grad_x, gx_stride, grad_weights_f32, vec_ok, gx_col0,     \
                               gx_accumulate, go_vec, pre, gc);                                                                            \
    } while (0)
// the composed operand fetch is built for the density networks' shape only: ONE hidden layer (sigma_net, network_dynamic.py:125-135)
#define BY_C1(S, XF, FA) do { if (gc.sigma) LAUNCH(S, 1, XF, FA, true); else LAUNCH(S, 1, XF, FA, false); } while (0)
#define BY_F1(S, XF) do { if (fast) BY_C1(S, XF, true); else BY_C1(S, XF, false); } while (0)
#define BY_F2(S, XF) do { if (fast) LAUNCH(S, 2, XF, true, false); else LAUNCH(S, 2, XF, false, false); } while (0)
#define BY_H(S)                                                                                                                            \
    do {                                                                                                                                   \
        if (n_hidden == 1) { if (x_is_f16) BY_F1(S, true); else BY_F1(S, false); }                                                         \
        else { if (x_is_f16) BY_F2(S, true); else BY_F2(S, false); }                                                                       \
    } while (0)
    if (gc.sigma && n_hidden != 1) return NVSF_ERR_UNSUPPORTED;
    switch (in_steps) {
        case 1: BY_H(1); break;
        case 2: BY_H(2); break;
        case 3: BY_H(3); break;
        case 4: BY_H(4); break;
        default: return NVSF_ERR_UNSUPPORTED;
    }
    return nvsf_launch_status();
}

NVSF_API int nvsf_mlp_bwd(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                          uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_out, uint32_t n_out,
                          uint32_t go_stride, float grad_scale, float* grad_x, uint32_t gx_stride, float* grad_weights_f32,
                          uint32_t gx_col0, int gx_accumulate, hipStream_t stream) {
    const XPrefix none = {nullptr, 0, 1, 0};
    const GoCompose as_is = {nullptr, nullptr, nullptr, nullptr, 0, 0.0f, 0.0f};
    return mlp_bwd_impl(x, x_is_f16, M, n_in, x_stride, weights_f16, in_cols, hidden, n_hidden, out_cols, grad_out, n_out, go_stride, grad_scale,
                        grad_x, gx_stride, grad_weights_f32, gx_col0, gx_accumulate, none, as_is, stream);
}

// nvsf_mlp_bwd for a density network whose logit gradient is formed from its parts while the operands are fetched (GoCompose above):
// replaces nvsf_sigma_geo_bwd + nvsf_mlp_bwd (one launch and 128 B / sample of matrix traffic less) and sums two heads' geometry gradients.
NVSF_API int nvsf_mlp_bwd_density(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                                  uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_sigma,
                                  const float* sigma, const float* grad_geo_a, const float* grad_geo_b, uint32_t geo_stride, uint32_t n_geo,
                                  float sigma_lo, float sigma_hi, float grad_scale, float* grad_x, uint32_t gx_stride, float* grad_weights_f32,
                                  uint32_t gx_col0, int gx_accumulate, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(sigma && grad_geo_a && n_geo >= 1 && n_geo <= 15 && geo_stride >= 16 && geo_stride % 4 == 0);
    REQUIRE(((reinterpret_cast<uintptr_t>(grad_geo_a) | reinterpret_cast<uintptr_t>(grad_geo_b)) & 15u) == 0);
    const XPrefix none = {nullptr, 0, 1, 0};
    const GoCompose gc = {sigma, grad_sigma, grad_geo_a, grad_geo_b, geo_stride, sigma_lo, sigma_hi};
    return mlp_bwd_impl(x, x_is_f16, M, n_in, x_stride, weights_f16, in_cols, hidden, n_hidden, out_cols, nullptr, 1 + n_geo, 16, grad_scale,
                        grad_x, gx_stride, grad_weights_f32, gx_col0, gx_accumulate, none, gc, stream);
}

// nvsf_mlp_bwd on rows with a shared prefix (see nvsf_mlp_fwd_prefix); the columns of dL/dx are those of the logical row.
NVSF_API int nvsf_mlp_bwd_prefix(const void* prefix_f16, uint32_t prefix_stride, uint32_t rows_per_prefix, uint32_t prefix_cols,
                                 const void* x_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                                 uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, const float* grad_out,
                                 uint32_t n_out, uint32_t go_stride, float grad_scale, float* grad_x, uint32_t gx_stride,
                                 float* grad_weights_f32, uint32_t gx_col0, int gx_accumulate, hipStream_t stream) {
    REQUIRE(prefix_f16 && prefix_cols > 0);
    const XPrefix pre = {reinterpret_cast<const _Float16*>(prefix_f16), prefix_stride, rows_per_prefix, prefix_cols};
    const GoCompose as_is = {nullptr, nullptr, nullptr, nullptr, 0, 0.0f, 0.0f};
    return mlp_bwd_impl(x_f16, 1, M, n_in, x_stride, weights_f16, in_cols, hidden, n_hidden, out_cols, grad_out, n_out, go_stride, grad_scale,
                        grad_x, gx_stride, grad_weights_f32, gx_col0, gx_accumulate, pre, as_is, stream);
}
