// Store-pattern micro-benchmark: what limits the sample stores of the marcher (12-B / 8-B rows per lane)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// V1: 16 B per lane, aligned
__global__ __launch_bounds__(256) void v1(float4* out, size_t n16, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n16; i += stride) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// V2: wave = "ray": batches of 64 rows of 12 B, contiguous per wave: rows [w * per, (w+1) * per)
__global__ __launch_bounds__(256) void v2(float* out, uint32_t rows_per_wave, uint32_t n_waves, uint32_t shift) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const uint32_t base = w * rows_per_wave + shift;
    for (uint32_t r = lane; r < rows_per_wave; r += 64) {
        float* p = out + 3 * (size_t)(base + r);
        p[0] = 1.f; p[1] = 2.f; p[2] = (float)r;
    }
}
// V4: three arrays 12 / 12 / 8 B per row
__global__ __launch_bounds__(256) void v4(float* a, float* b, float* c, uint32_t rows_per_wave, uint32_t n_waves, uint32_t shift) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const uint32_t base = w * rows_per_wave + shift;
    for (uint32_t r = lane; r < rows_per_wave; r += 64) {
        const size_t s = base + r;
        float* p = a + 3 * s; p[0] = 1.f; p[1] = 2.f; p[2] = (float)r;
        float* q = b + 3 * s; q[0] = 1.f; q[1] = 2.f; q[2] = 3.f;
        float* d = c + 2 * s; d[0] = 1.f; d[1] = (float)r;
    }
}
// V5: the same 32 B per row as ONE array of 32-B rows (two 16-B stores per lane)
__global__ __launch_bounds__(256) void v5(float4* a, uint32_t rows_per_wave, uint32_t n_waves) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const uint32_t base = w * rows_per_wave;
    for (uint32_t r = lane; r < rows_per_wave; r += 64) {
        const size_t s = base + r;
        a[2 * s] = make_float4(1.f, 2.f, 3.f, 4.f);
        a[2 * s + 1] = make_float4(1.f, 2.f, 3.f, (float)r);
    }
}
// V6: like V4 but the wave writes whole 16-B pieces: lane l writes piece l of the batch's 768-B (48 pieces) / 512-B (32 pieces) block
// (data pattern irrelevant here; shows what an LDS-transposed store would reach)
__global__ __launch_bounds__(256) void v6(float* a, float* b, float* c, uint32_t rows_per_wave, uint32_t n_waves) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const uint32_t base = w * rows_per_wave;
    for (uint32_t r0 = 0; r0 < rows_per_wave; r0 += 64) {
        const size_t s = base + r0;
        if (lane < 48) { reinterpret_cast<float4*>(a + 3 * s)[lane] = make_float4(1.f, 2.f, 3.f, 4.f); reinterpret_cast<float4*>(b + 3 * s)[lane] = make_float4(1.f, 2.f, 3.f, 4.f); }
        if (lane < 32) reinterpret_cast<float4*>(c + 2 * s)[lane] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

int main() {
    const uint32_t n_waves = 32768, rows = 704;  // 23 M rows, 738 MB at 32 B per row
    const size_t total_rows = (size_t)n_waves * rows + 64;
    float *a, *b, *c;
    CK(hipMalloc(&a, total_rows * 32)); CK(hipMalloc(&b, total_rows * 12)); CK(hipMalloc(&c, total_rows * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double bytes) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-48s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
        fflush(stdout);
    };
    const double B = (double)n_waves * rows;
    timeit("v1 16 B/lane aligned, 738 MB", [&] { hipLaunchKernelGGL(v1, dim3(8192), dim3(256), 0, 0, (float4*)a, (size_t)(B * 2), 0); }, B * 32);
    timeit("v2 12 B rows one array, aligned start", [&] { hipLaunchKernelGGL(v2, dim3(n_waves / 4), dim3(256), 0, 0, a, rows, n_waves, 0u); }, B * 12);
    timeit("v2 12 B rows one array, shifted 5 rows", [&] { hipLaunchKernelGGL(v2, dim3(n_waves / 4), dim3(256), 0, 0, a, rows, n_waves, 5u); }, B * 12);
    timeit("v2 12 B rows, 701 rows per wave (ragged)", [&] { hipLaunchKernelGGL(v2, dim3(n_waves / 4), dim3(256), 0, 0, a, 701u, n_waves, 0u); }, (double)n_waves * 701 * 12);
    timeit("v4 three arrays 12/12/8", [&] { hipLaunchKernelGGL(v4, dim3(n_waves / 4), dim3(256), 0, 0, a, b, c, rows, n_waves, 0u); }, B * 32);
    timeit("v4 three arrays, 701 rows per wave", [&] { hipLaunchKernelGGL(v4, dim3(n_waves / 4), dim3(256), 0, 0, a, b, c, 701u, n_waves, 0u); }, (double)n_waves * 701 * 32);
    timeit("v5 one array of 32-B rows", [&] { hipLaunchKernelGGL(v5, dim3(n_waves / 4), dim3(256), 0, 0, (float4*)a, rows, n_waves); }, B * 32);
    timeit("v6 three arrays, 16-B pieces per lane", [&] { hipLaunchKernelGGL(v6, dim3(n_waves / 4), dim3(256), 0, 0, a, b, c, rows, n_waves); }, B * 32);
    return 0;
}
