// Scatter-add rate by atomic flavour (table-gradient access shape: 32 random 8-byte entries per wave instruction).
// hipcc --offload-arch=gfx950 -O3 tools/exp_atomics.hip -o /tmp/exp_atomics && /tmp/exp_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint32_t rng(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0 f32 agent, 1 i32 agent, 2 u64 agent (one lane per entry), 3 f32 workgroup scope, 4 i32 workgroup scope,
// 5 plain store (upper bound), 6 f64 agent (one lane per entry), 7 pk f16 (one lane per entry, 4 B)
template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(void* table, uint32_t entries_log2, uint32_t iters, int partition) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t mask = (1u << entries_log2) - 1, base = 0;
    if (partition) { mask = (1u << (entries_log2 - 3)) - 1; base = (blockIdx.x & 7) << (entries_log2 - 3); }
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t e = base + (rng((wave * iters + it) * 32u + (lane >> 1)) & mask);  // entry shared by a lane pair
        if (MODE == 0) {
            float* p = (float*)table + 2 * (size_t)e + (lane & 1);
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 1) {
            int* p = (int*)table + 2 * (size_t)e + (lane & 1);
            __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 2) {
            if (!(lane & 1)) __hip_atomic_fetch_add((unsigned long long*)table + e, 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 3) {
            float* p = (float*)table + 2 * (size_t)e + (lane & 1);
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 4) {
            int* p = (int*)table + 2 * (size_t)e + (lane & 1);
            __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 5) {
            ((float*)table)[2 * (size_t)e + (lane & 1)] = 1.0f;
        } else if (MODE == 6) {
            if (!(lane & 1)) __hip_atomic_fetch_add((double*)table + e, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 8) {  // fixed point: one 64-bit integer per (entry, feature), all lanes active (16-byte entries)
            unsigned long long* p = (unsigned long long*)table + 2 * (size_t)e + (lane & 1);
            __hip_atomic_fetch_add(p, 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 7) {
            typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
            if (!(lane & 1)) {
                half2_t v = {(_Float16)1.0f, (_Float16)1.0f};
                __builtin_amdgcn_global_atomic_fadd_v2f16((half2_t*)table + 2 * (size_t)e, v);
            }
        }
    }
}

template <int MODE>
int run(const char* name, void* table, uint32_t entries_log2, int partition) {
    const uint32_t blocks = 256 * 8, iters = 512;  // 8192 waves x 512 wave-instructions x 32 entries = 134 M entry adds
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(table, 0, (size_t)(MODE == 8 ? 16 : 8) << entries_log2));
    hipLaunchKernelGGL(k_scatter<MODE>, dim3(blocks), dim3(256), 0, 0, table, entries_log2, 16u, partition);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_scatter<MODE>, dim3(blocks), dim3(256), 0, 0, table, entries_log2, iters, partition);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double adds = (double)blocks * 4 * iters * 32;
    printf("%-22s table %6.1f MB part %d: %7.3f ms  %6.2f G entries/s  %6.3f TB/s of 8-B entries\n", name, (double)(8ull << entries_log2) / 1e6,
           partition, ms, adds / ms / 1e6, adds * 8 / ms / 1e9);
    fflush(stdout);
    return 0;
}

int main() {
    void* table;
    CK(hipMalloc(&table, (size_t)16 << 24));
    for (uint32_t lg : {19u, 23u}) {  // 4 MB (one level of the hash grid), 64 MB
        for (int part = 0; part < 1; ++part) {
            if (run<0>("f32 agent", table, lg, part)) return 1;
            if (run<1>("i32 agent", table, lg, part)) return 1;
            if (run<2>("u64 agent", table, lg, part)) return 1;
            if (run<8>("u64 x2 per entry", table, lg, part)) return 1;
            if (run<6>("f64 agent", table, lg, part)) return 1;
            if (run<7>("pk_f16 agent", table, lg, part)) return 1;
            if (run<3>("f32 workgroup", table, lg, part)) return 1;
            if (run<4>("i32 workgroup", table, lg, part)) return 1;
            if (run<5>("plain store", table, lg, part)) return 1;
        }
    }
    return 0;
}
