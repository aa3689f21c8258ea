// Feature assembly + density MLP of the space-time field for gfx950 (network_dynamic.py:273-287 of the reference):
//   plane_d' = 0.5 plane_d + 0.25 (plane_1 + plane_2)          [M,32] fp32
//   hash_d'  = 0.5 hash_d  + 0.25 (hash_1  + hash_2)           [M,24]  (hash_1/2 fp16 when they come from the 0-dim-t path)
//   features = [plane_s | plane_d' | hash_s | hash_d'] (120)   -> sigma_net 120 -> 64 -> 16
// In the reference this is six elementwise kernels, a 480-byte-per-sample concatenation and a tcnn launch.  Here a
// wave builds the four 32-wide MFMA B fragments of a 16-sample tile straight from the eight source buffers (each
// lane reads 8 consecutive features), applies the same fp32 / fp16 arithmetic PyTorch applies, and runs the MLP on
// the matrix cores; the [M,120] feature matrix is never written.
#include "mlp_device.h"

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

struct DynFeat {
    const float* plane_s;  // [M,32]
    const float* plane_d;  // [M,32]
    const float* plane_1;  // [M,32] (may alias plane_d)
    const float* plane_2;
    const _Float16* hash_s;  // [M,32]
    const float* hash_d;     // [M,24]
    const void* hash_1;      // [M,24] fp16 or fp32
    const void* hash_2;
    int h1_f16, h2_f16;
    int planes_f16;  // plane_s / plane_d are fp16 [M,32] rows and plane_d is already the blend (nvsf_planes_multi_fwd, blend = 2)
    int hash_s_lm;   // hash_s is level-major [8][M][4] (nvsf_hashgrid_fwd_level_major) instead of rows [M,32]
};

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8h(const _Float16* p, float (&v)[8]) {
    const half8_t h = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
}
__device__ __forceinline__ float r16(float v) { return (float)(_Float16)v; }

// ROT = 0: out_h fp32 [M,16] = the 16 network outputs.
// ROT = 1: sigma fp32 [M] = exp(h0) and geo fp16 [M,16] = (h1..h15, 1.0) (the layout the fused head kernel reads).
template <int ROT>
__global__ __launch_bounds__(kBlock) void k_density_dynamic(DynFeat f, uint32_t M, const _Float16* __restrict__ w_sigma,
                                                            float* __restrict__ out_h, float* __restrict__ sigmas,
                                                            _Float16* __restrict__ geo, _Float16* __restrict__ x_out) {
    const int lane = lane_id(), g = lane >> 4, sl = lane & 15;
    half8_t w0[kHidTiles][4];
#pragma unroll
    for (int t = 0; t < kHidTiles; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) w0[t][s] = load_w_natural(w_sigma, 128, t, s, lane);
    OutLayerW wout;
    wout.load(w_sigma + kHidden * 128, lane, ROT);

    const uint32_t n_tiles = (M + 15) / 16;
    const uint32_t wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t wave_count = gridDim.x * kWavesPerBlock;
    for (uint32_t tile = wave_global; tile < n_tiles; tile += wave_count) {
        const uint32_t m_raw = tile * 16 + sl;
        const size_t m = m_raw < M ? m_raw : M - 1;
        half8_t xf[4];
        float a[8], b[8], c[8];
        if (f.planes_f16) {  // (fp16)(0.5 v + 0.25 (v + v)) == (fp16)v: the producer has rounded exactly what this kernel would
            xf[0] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const _Float16*>(f.plane_s) + m * 32 + 8 * g);
            xf[1] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const _Float16*>(f.plane_d) + m * 32 + 8 * g);
        } else {
            load8(f.plane_s + m * 32 + 8 * g, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[0][j] = (_Float16)a[j];
            load8(f.plane_d + m * 32 + 8 * g, a);
            load8(f.plane_1 + m * 32 + 8 * g, b);
            load8(f.plane_2 + m * 32 + 8 * g, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[1][j] = (_Float16)(0.5f * a[j] + 0.25f * (b[j] + c[j]));
        }
        if (f.hash_s_lm) {  // features 8g .. 8g + 7 = levels 2g, 2g + 1
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const uint2 lo = reinterpret_cast<const uint2*>(f.hash_s)[(size_t)(2 * g) * M + m], hi = reinterpret_cast<const uint2*>(f.hash_s)[(size_t)(2 * g + 1) * M + m];
            const u4v v = {lo.x, lo.y, hi.x, hi.y};
            xf[2] = __builtin_bit_cast(half8_t, v);
        } else {
            xf[2] = *reinterpret_cast<const half8_t*>(f.hash_s + m * 32 + 8 * g);
        }
        if (g < 3) {
            load8(f.hash_d + m * 24 + 8 * g, a);
            if (f.h1_f16) load8h(reinterpret_cast<const _Float16*>(f.hash_1) + m * 24 + 8 * g, b);
            else load8(reinterpret_cast<const float*>(f.hash_1) + m * 24 + 8 * g, b);
            if (f.h2_f16) load8h(reinterpret_cast<const _Float16*>(f.hash_2) + m * 24 + 8 * g, c);
            else load8(reinterpret_cast<const float*>(f.hash_2) + m * 24 + 8 * g, c);
            const bool half_sum = f.h1_f16 && f.h2_f16;  // fp16 + fp16 stays fp16 in PyTorch, and so does 0.25 * (fp16)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float s = half_sum ? r16(b[j] + c[j]) : (b[j] + c[j]);
                const float q = half_sum ? r16(0.25f * s) : 0.25f * s;
                xf[3][j] = (_Float16)(0.5f * a[j] + q);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[3][j] = (_Float16)1.0f;  // columns 120..127: tcnn pads the network input with ones
        }
        if (x_out && m_raw < M) {  // the assembled, rounded network input (training keeps it for the backward pass)
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<half8_t*>(x_out + m * 128 + 32 * s + 8 * g) = xf[s];
        }
        float4_t acc[kHidTiles];
#pragma unroll
        for (int t = 0; t < kHidTiles; ++t) {
            float4_t cacc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) cacc = mfma16(w0[t][s], xf[s], cacc);
            acc[t] = cacc;
        }
        half8_t h[kHidSteps];
        pack_hidden(acc, h);
        const float4_t o = wout.apply(h);
        if (m_raw < M) {
            if (ROT == 0) {
                *reinterpret_cast<float4_t*>(out_h + m * 16 + 4 * g) = o;
            } else {
                half4_t ov;
                ov[0] = (_Float16)o[0]; ov[1] = (_Float16)o[1]; ov[2] = (_Float16)o[2]; ov[3] = (_Float16)o[3];
                if (g == 3) {
                    sigmas[m] = expf(o[3]);
                    ov[3] = (_Float16)1.0f;
                }
                *reinterpret_cast<half4_t*>(geo + m * 16 + 4 * g) = ov;
            }
        }
    }
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

static int density_dynamic_impl(const float* plane_s, const float* plane_d, const float* plane_1, const float* plane_2, int planes_f16,
                                      int hash_s_lm, const void* hash_s_f16, const float* hash_d, const void* hash_1, int hash_1_is_f16, const void* hash_2,
                                      int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h, float* sigmas,
                                      void* geo_f16, void* x_f16_out, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(plane_s && plane_d && plane_1 && plane_2 && hash_s_f16 && hash_d && hash_1 && hash_2 && sigma_weights_f16);
    REQUIRE(out_h || (sigmas && geo_f16));
    const void* ptrs[] = {plane_s, plane_d, plane_1, plane_2, hash_s_f16, hash_d, hash_1, hash_2, sigma_weights_f16, out_h, geo_f16};
    for (const void* p : ptrs) REQUIRE((reinterpret_cast<uintptr_t>(p) & 15u) == 0);
    DynFeat f;
    f.plane_s = plane_s; f.plane_d = plane_d; f.plane_1 = plane_1; f.plane_2 = plane_2;
    f.hash_s = reinterpret_cast<const _Float16*>(hash_s_f16);
    f.hash_d = hash_d; f.hash_1 = hash_1; f.hash_2 = hash_2;
    f.h1_f16 = hash_1_is_f16; f.h2_f16 = hash_2_is_f16;
    f.planes_f16 = planes_f16;
    f.hash_s_lm = hash_s_lm;
    const uint32_t n_tiles = (M + 15) / 16;
    const uint32_t blocks = n_tiles / kWavesPerBlock + 1 < 2048u ? n_tiles / kWavesPerBlock + 1 : 2048u;
    const _Float16* w = reinterpret_cast<const _Float16*>(sigma_weights_f16);
    REQUIRE((reinterpret_cast<uintptr_t>(x_f16_out) & 15u) == 0);
    _Float16* xo = reinterpret_cast<_Float16*>(x_f16_out);
    if (out_h)
        hipLaunchKernelGGL(k_density_dynamic<0>, dim3(blocks), dim3(kBlock), 0, stream, f, M, w, out_h, (float*)nullptr, (_Float16*)nullptr, xo);
    else
        hipLaunchKernelGGL(k_density_dynamic<1>, dim3(blocks), dim3(kBlock), 0, stream, f, M, w, (float*)nullptr, sigmas,
                           reinterpret_cast<_Float16*>(geo_f16), xo);
    return nvsf_launch_status();
}

NVSF_API int nvsf_density_dynamic_fwd(const float* plane_s, const float* plane_d, const float* plane_1, const float* plane_2,
                                      const void* hash_s_f16, const float* hash_d, const void* hash_1, int hash_1_is_f16, const void* hash_2,
                                      int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h, float* sigmas,
                                      void* geo_f16, void* x_f16_out, hipStream_t stream) {
    return density_dynamic_impl(plane_s, plane_d, plane_1, plane_2, 0, 0, hash_s_f16, hash_d, hash_1, hash_1_is_f16, hash_2, hash_2_is_f16, M,
                                sigma_weights_f16, out_h, sigmas, geo_f16, x_f16_out, stream);
}

// The same with the K-planes features as fp16 rows (plane_s, plane_d_blended: [M,32] fp16 from nvsf_planes_multi_fwd with blend = 2;
// the neighbour blend of the plane features has been formed by the producer).  Bit-identical outputs, 256 B per sample less traffic.
NVSF_API int nvsf_density_dynamic_f16planes_fwd(const void* plane_s_f16, const void* plane_d_blended_f16, const void* hash_s_f16,
                                                const float* hash_d, const void* hash_1, int hash_1_is_f16, const void* hash_2,
                                                int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h, float* sigmas,
                                                void* geo_f16, void* x_f16_out, hipStream_t stream) {
    const float* ps = reinterpret_cast<const float*>(plane_s_f16);
    const float* pd = reinterpret_cast<const float*>(plane_d_blended_f16);
    return density_dynamic_impl(ps, pd, pd, pd, 1, 0, hash_s_f16, hash_d, hash_1, hash_1_is_f16, hash_2, hash_2_is_f16, M, sigma_weights_f16, out_h,
                                sigmas, geo_f16, x_f16_out, stream);
}

// ... and with the static hash features level-major, fp16 [8][M][4], as nvsf_hashgrid_fwd_level_major writes them.  Bit-identical outputs.
NVSF_API int nvsf_density_dynamic_lm_fwd(const void* plane_s_f16, const void* plane_d_blended_f16, const void* hash_s_level_major_f16,
                                         const float* hash_d, const void* hash_1, int hash_1_is_f16, const void* hash_2,
                                         int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h, float* sigmas,
                                         void* geo_f16, void* x_f16_out, hipStream_t stream) {
    const float* ps = reinterpret_cast<const float*>(plane_s_f16);
    const float* pd = reinterpret_cast<const float*>(plane_d_blended_f16);
    return density_dynamic_impl(ps, pd, pd, pd, 1, 1, hash_s_level_major_f16, hash_d, hash_1, hash_1_is_f16, hash_2, hash_2_is_f16, M,
                                sigma_weights_f16, out_h, sigmas, geo_f16, x_f16_out, stream);
}

// The four-buffer plane form (training forward: fp32 plane rows, the blend formed here) with the static hash features level-major.
NVSF_API int nvsf_density_dynamic_lm32_fwd(const float* plane_s, const float* plane_d, const float* plane_1, const float* plane_2,
                                           const void* hash_s_level_major_f16, const float* hash_d, const void* hash_1, int hash_1_is_f16,
                                           const void* hash_2, int hash_2_is_f16, uint32_t M, const void* sigma_weights_f16, float* out_h,
                                           float* sigmas, void* geo_f16, void* x_f16_out, hipStream_t stream) {
    return density_dynamic_impl(plane_s, plane_d, plane_1, plane_2, 0, 1, hash_s_level_major_f16, hash_d, hash_1, hash_1_is_f16, hash_2, hash_2_is_f16, M,
                                sigma_weights_f16, out_h, sigmas, geo_f16, x_f16_out, stream);
}
