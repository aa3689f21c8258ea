// Internal helpers shared by the HIP translation units of libnvsf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NVSF_API extern "C" __attribute__((visibility("default")))

// Status codes returned by every C-ABI entry point (include/nvsf_hip.h).
enum : int {
    NVSF_OK = 0,
    NVSF_ERR_INVALID_ARG = -1,   // null pointer / inconsistent sizes / unsupported configuration
    NVSF_ERR_UNSUPPORTED = -2,   // template instantiation not built for this shape
};

constexpr int kWave = 64;  // CDNA4 wavefront width

static inline int nvsf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NVSF_OK : (int)e;  // positive values are hipError_t codes
}

static inline unsigned cdiv(unsigned long long a, unsigned b) { return (unsigned)((a + b - 1) / b); }

// Test-only choice of a kernel formulation (include/nvsf_hip.h: nvsf_test_variant; version.hip).  Every operator has ONE production
// form (value 0) and, where a second formulation is kept as the reference the tests compare it with, that one under value 1 (2).
// Launchers read a plain int here -- no environment look-ups on the call path.
enum NvsfVariantKey : int {
    kVarMarch = 0,      // march_rays_train: 0 wave kernels / one launch, 1 one thread per ray, 2 wave kernels walking a batch member by member
    kVarPlanesFwd,      // 0 rows walked along the ray, 1 one thread per (sample, scale)
    kVarPlanesBwd,      // 0 run-merging (multi entry: time planes through an LDS image), 1 one atomic per (sample, texel, channel), 2 as 0 with global atomics only
    kVarHashgridFwd,    // 0 level-per-XCD where eligible, 1 the generic row kernel
    kVarHashgridBwd,    // 0 corner-parallel run merging, 1 one thread per (row, level)
    kVarHash4dBwd,      // 0 LDS accumulation, 1 run-merging global atomics
    kVarSlicePlan,      // 0 balanced slices, 1 every group its own slice
    kVarRenderTail,     // 0 two tiles per iteration, 1 one tile
    kVarMarchSkew,      // one-launch marcher: 0 off, q + 1 = the workgroups of ticket queue q start late (the other queues steal from it)
    kVarMlpBwd,         // nvsf_mlp_bwd: 0 by shape (wave-independent kernel, transposes on the matrix core; LDS-staged kernel for 32-64-16), 1 staged, 2 wave
    kVarLevelKinds,     // gathering render kernels on the config-2 grid: 0 the instance with the level kinds compiled in, 1 the general one
    kVarCount
};
int nvsf_variant(int key);
int nvsf_cu_count();  // compute units of the current device (256 on MI355X)

// ---- wave-level primitives (64 lanes) -------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive scans over the 64 lanes of a wave (Hillis-Steele on cross-lane shuffles)
__device__ __forceinline__ float wave_scan_add(float v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (l >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (l >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(v, o, 64);
        if (l >= o) v += u;
    }
    return v;
}

// inclusive max-scan over the 64 lanes on DPP (row shifts inside the 16-lane rows, then row_bcast:15 / row_bcast:31)
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) {
#define NVSF_DPP_MAX(ctrl, row_mask)                                                                                        \
    {                                                                                                                      \
        const uint32_t u = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xF, false);                   \
        v = u > v ? u : v;                                                                                                 \
    }
    NVSF_DPP_MAX(0x111, 0xF)  // row_shr:1
    NVSF_DPP_MAX(0x112, 0xF)  // row_shr:2
    NVSF_DPP_MAX(0x114, 0xF)  // row_shr:4
    NVSF_DPP_MAX(0x118, 0xF)  // row_shr:8
    NVSF_DPP_MAX(0x142, 0xA)  // row_bcast:15 -> rows 1 and 3
    NVSF_DPP_MAX(0x143, 0xC)  // row_bcast:31 -> rows 2 and 3
#undef NVSF_DPP_MAX
    return v;
}
