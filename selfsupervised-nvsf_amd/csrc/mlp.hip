// Stand-alone fused MLP forward (all layers in one kernel, activations in registers) for gfx950.
// Replaces tcnn.Network("FullyFusedMLP") as instantiated at network_dynamic.py:125-135 (sigma_net
// 120->64->16), :138-161 (intensity / raydrop 87->64->64->1) and :180-189 (color 31->64->64->3).
//
// One wave = one 16-sample tile at a time (grid-stride over tiles); every weight fragment is fetched once
// per wave and stays in VGPRs.  Input rows may be fp32 or fp16; they are rounded to fp16, columns
// n_in..in_cols-1 read as 1.0 (tcnn pads the network input with ones), the rest of the last k-step is 0.
#include "mlp_device.h"

namespace {
constexpr int kBlock = 256;

template <int IN_STEPS, int N_HIDDEN, bool X_F16, bool FAST>
__global__ __launch_bounds__(kBlock) void k_mlp_fwd(const void* __restrict__ x, uint32_t M, uint32_t n_in, uint32_t x_stride,
                                                    const _Float16* __restrict__ weights, uint32_t in_cols,
                                                    float* __restrict__ out, uint32_t out_stride, int vec_ok, XPrefix pre, uint32_t n_store) {
    const int lane = lane_id();
    const int g = lane >> 4, sl = lane & 15;
    // ---- weights -> registers
    half8_t w0[kHidTiles][IN_STEPS];
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
        for (int s = 0; s < IN_STEPS; ++s) w0[t][s] = load_w_natural(weights, (int)in_cols, t, s, lane);
    const _Float16* wp = weights + (size_t)kHidden * in_cols;
    HiddenLayerW hid[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
#pragma unroll
    for (int i = 0; i < N_HIDDEN - 1; ++i) {
        hid[i].load(wp, lane);
        wp += kHidden * kHidden;
    }
    OutLayerW wout;
    wout.load(wp, lane);

    XTail tail;
    if constexpr (FAST) tail.init(32 * (IN_STEPS - 1) + 8 * g, (int)n_in, (int)in_cols);
    const uint32_t n_tiles = (M + 15) / 16;
    const uint32_t wave_global = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const uint32_t wave_count = gridDim.x * (kBlock / kWave);
    for (uint32_t tile = wave_global; tile < n_tiles; tile += wave_count) {
        const uint32_t m = tile * 16 + sl;
        const size_t row = m < M ? m : M - 1;
        half8_t xf[IN_STEPS];
        const uint32_t tile_u = __builtin_amdgcn_readfirstlane(tile);
        const _Float16* prow = pre.a ? pre.row_of(tile_u * 16u < M ? tile_u * 16u : M - 1u) : nullptr;
        issue_x_row<IN_STEPS, X_F16, FAST>(xf, x, row, x_stride, g, (int)n_in, (int)in_cols, vec_ok != 0, tail, prow, pre.split);
        if constexpr (FAST) xf[IN_STEPS - 1] = tail.apply(xf[IN_STEPS - 1]);
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
#pragma unroll
        for (int i = 0; i < N_HIDDEN - 1; ++i) {
            hid[i].apply(h, acc);
            pack_hidden(acc, h);
        }
        const float4_t o = wout.apply(h);
        if (n_store >= 16u) {  // uniform
            if (m < M) *reinterpret_cast<float4_t*>(out + (size_t)m * out_stride + 4 * g) = o;  // fp32 logits (not rounded to fp16)
        } else if (m < M && g == 0) {  // a head with 1 ... 4 outputs: only those columns leave (4 ... 16 B per sample instead of 64)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((uint32_t)r < n_store) out[(size_t)m * out_stride + r] = o[r];
        }
    }
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

// weights: fp16 [64][in_cols] ++ (n_hidden-1) x [64][64] ++ [16][64]; out fp32 [M, out_stride]: all 16 columns (out_cols = 16) or the leading 1 ... 4
static int mlp_fwd_impl(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                        uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, float* out_f32,
                        uint32_t out_stride, XPrefix pre, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && weights_f16 && out_f32);
    REQUIRE(n_in >= 1 && n_in <= in_cols && in_cols % 16 == 0 && x_stride >= n_in - pre.split);
    if (out_cols == 16) REQUIRE(out_stride >= 16 && out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out_f32) & 15u) == 0);
    else REQUIRE(out_cols >= 1 && out_cols <= 4 && out_stride >= out_cols && (reinterpret_cast<uintptr_t>(out_f32) & 3u) == 0);
    REQUIRE((reinterpret_cast<uintptr_t>(weights_f16) & 15u) == 0);
    if (hidden != (uint32_t)kHidden || n_hidden < 1 || n_hidden > 3 || in_cols > 128) return NVSF_ERR_UNSUPPORTED;
    const int in_steps = (int)((in_cols + 31) / 32);
    const size_t esz = x_is_f16 ? 2 : 4;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && ((x_stride * esz) % 16 == 0);
    const uint32_t n_tiles = (M + 15) / 16;
    const uint32_t blocks = n_tiles / 4 + 1 < 2048u ? n_tiles / 4 + 1 : 2048u;
    const _Float16* w = reinterpret_cast<const _Float16*>(weights_f16);
    float* o = out_f32;
    const bool fast = x_rows_fast(n_in - pre.split, x_stride, vec_ok);
    if (pre.a) {  // shared-prefix rows: aligned fp16 only, whole 8-column groups on either side, a tile inside one group
        REQUIRE(x_is_f16 && fast && pre.split % 8 == 0 && pre.split < n_in && pre.a_stride >= pre.split && pre.a_stride % 8 == 0);
        REQUIRE(pre.rows_per_a >= 16 && pre.rows_per_a % 16 == 0 && (reinterpret_cast<uintptr_t>(pre.a) & 15u) == 0);
    }
#define LAUNCH(S, H, XF, FA)                                                                                                   \
    hipLaunchKernelGGL((k_mlp_fwd<S, H, XF, FA>), dim3(blocks), dim3(kBlock), 0, stream, x, M, n_in, x_stride, w, in_cols, o, \
                       out_stride, vec_ok, pre, out_cols)
#define BY_F(S, H, XF) do { if (fast) LAUNCH(S, H, XF, true); else LAUNCH(S, H, XF, false); } while (0)
#define BY_X(S, H) do { if (x_is_f16) BY_F(S, H, true); else BY_F(S, H, false); } while (0)
#define BY_H(S)                                       \
    do {                                              \
        if (n_hidden == 1) BY_X(S, 1);                \
        else if (n_hidden == 2) BY_X(S, 2);           \
        else BY_X(S, 3);                              \
    } while (0)
    switch (in_steps) {
        case 1: BY_H(1); break;
        case 2: BY_H(2); break;
        case 3: BY_H(3); break;
        case 4: BY_H(4); break;
        default: return NVSF_ERR_UNSUPPORTED;
    }
    return nvsf_launch_status();
}

NVSF_API int nvsf_mlp_fwd(const void* x, int x_is_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                          uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, float* out_f32,
                          uint32_t out_stride, hipStream_t stream) {
    const XPrefix none = {nullptr, 0, 1, 0};
    return mlp_fwd_impl(x, x_is_f16, M, n_in, x_stride, weights_f16, in_cols, hidden, n_hidden, out_cols, out_f32, out_stride, none, stream);
}

// The same network on rows with a shared prefix: logical row r = [prefix[r / rows_per_prefix][0 : prefix_cols] | x[r][0 : n_in - prefix_cols]]
// (fp16, 16-byte aligned rows, prefix_cols % 8 == 0, rows_per_prefix % 16 == 0).  Same result as nvsf_mlp_fwd on the assembled rows.
NVSF_API int nvsf_mlp_fwd_prefix(const void* prefix_f16, uint32_t prefix_stride, uint32_t rows_per_prefix, uint32_t prefix_cols,
                                 const void* x_f16, uint32_t M, uint32_t n_in, uint32_t x_stride, const void* weights_f16,
                                 uint32_t in_cols, uint32_t hidden, uint32_t n_hidden, uint32_t out_cols, float* out_f32,
                                 uint32_t out_stride, hipStream_t stream) {
    REQUIRE(prefix_f16 && prefix_cols > 0);
    const XPrefix pre = {reinterpret_cast<const _Float16*>(prefix_f16), prefix_stride, rows_per_prefix, prefix_cols};
    return mlp_fwd_impl(x_f16, 1, M, n_in, x_stride, weights_f16, in_cols, hidden, n_hidden, out_cols, out_f32, out_stride, pre, stream);
}
