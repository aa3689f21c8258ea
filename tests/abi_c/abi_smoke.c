/* Plain-C client of libnvsf_hip.so: the drop-in boundary is a C ABI (include/nvsf_hip.h) -- raw device pointers, sizes, a
 * hipStream_t -- with no Python, torch or pybind in the way.  This program allocates with hipMalloc, calls the
 * raymarching entry points the reference's `_raymarching` pybind module exposes (bindings.cpp:5-21) and checks the
 * results against the CPU oracle (oracle/liboracle_raymarching.so, test infrastructure), bit for bit.
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include tests/abi_c/abi_smoke.c -o abi_smoke \
 *       -L selfsupervised-nvsf_amd/lib -lnvsf_hip -L oracle -loracle_raymarching -L/opt/rocm/lib -lamdhip64 -lm
 * Exit code 0 = all checks passed.  Run by tests/test_abi_c_gpu.py. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nvsf_hip.h"

void oracle_near_far_from_aabb(const float*, const float*, const float*, uint32_t, float, float*, float*);
void oracle_morton3D(const int32_t*, uint32_t, int32_t*);
void oracle_packbits(const float*, uint32_t, float, uint8_t*);
void oracle_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                             uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                             float* deltas, int32_t* rays, int32_t* counter, const float* noises);
void oracle_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                                         float T_thresh, float* weights_sum, float* depth, float* image);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NVSF(x) do { int s_ = (x); if (s_ != 0) { fprintf(stderr, "%s: status %d\n", #x, s_); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static float frand(void) { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 16777216.0f; }

static void* to_dev(const void* h, size_t n) { void* d = NULL; if (hipMalloc(&d, n) != hipSuccess) return NULL; hipMemcpy(d, h, n, hipMemcpyHostToDevice); return d; }

int main(void) {
    enum { N = 1500, C = 2, H = 128, MAXS = 128 };
    const uint32_t M = N * MAXS;
    float *o = malloc(3 * N * 4), *d = malloc(3 * N * 4), aabb[6] = {-2, -2, -2, 2, 2, 2};
    for (int i = 0; i < N; ++i) {
        float v[3], n2 = 0;
        for (int k = 0; k < 3; ++k) { o[3 * i + k] = (frand() - 0.5f) * (i % 7 == 0 ? 12.0f : 1.0f); v[k] = frand() - 0.5f; n2 += v[k] * v[k]; }
        float inv = 1.0f / __builtin_sqrtf(n2 + 1e-12f);
        for (int k = 0; k < 3; ++k) d[3 * i + k] = v[k] * inv;
    }
    /* occupancy grid: blocky density, packed on the device and on the host */
    const uint32_t cells = C * H * H * H;
    float* grid = malloc((size_t)cells * 4);
    for (uint32_t i = 0; i < cells; ++i) grid[i] = ((i >> 9) % 5 == 0) ? 1.0f : 0.0f;
    uint8_t *bits_ref = malloc(cells / 8), *bits = malloc(cells / 8);
    oracle_packbits(grid, cells / 8, 0.5f, bits_ref);
    float *dg = to_dev(grid, (size_t)cells * 4), *d_o = to_dev(o, 3 * N * 4), *d_d = to_dev(d, 3 * N * 4), *d_aabb = to_dev(aabb, 24);
    uint8_t* d_bits = NULL; HIP(hipMalloc((void**)&d_bits, cells / 8));
    float *d_near = NULL, *d_far = NULL; HIP(hipMalloc((void**)&d_near, N * 4)); HIP(hipMalloc((void**)&d_far, N * 4));
    hipStream_t stream; HIP(hipStreamCreate(&stream));
    printf("%s\n", nvsf_version());
    NVSF(nvsf_packbits(dg, cells / 8, 0.5f, d_bits, stream));
    NVSF(nvsf_near_far_from_aabb(d_o, d_d, d_aabb, N, 0.05f, d_near, d_far, stream));
    HIP(hipStreamSynchronize(stream));
    HIP(hipMemcpy(bits, d_bits, cells / 8, hipMemcpyDeviceToHost));
    if (memcmp(bits, bits_ref, cells / 8)) { fprintf(stderr, "packbits differs\n"); return 1; }
    float *near = malloc(N * 4), *far = malloc(N * 4), *near_r = malloc(N * 4), *far_r = malloc(N * 4);
    HIP(hipMemcpy(near, d_near, N * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(far, d_far, N * 4, hipMemcpyDeviceToHost));
    oracle_near_far_from_aabb(o, d, aabb, N, 0.05f, near_r, far_r);
    if (memcmp(near, near_r, N * 4) || memcmp(far, far_r, N * 4)) { fprintf(stderr, "near_far differs\n"); return 1; }
    /* Morton codes */
    int32_t *co = malloc(3 * N * 4), *mo = malloc(N * 4), *mo_r = malloc(N * 4);
    for (int i = 0; i < 3 * N; ++i) co[i] = (int32_t)(frand() * 1024.0f) & 1023;
    int32_t *d_co = to_dev(co, 3 * N * 4), *d_mo = NULL; HIP(hipMalloc((void**)&d_mo, N * 4));
    NVSF(nvsf_morton3D(d_co, N, d_mo, stream));
    HIP(hipStreamSynchronize(stream));
    HIP(hipMemcpy(mo, d_mo, N * 4, hipMemcpyDeviceToHost));
    oracle_morton3D(co, N, mo_r);
    if (memcmp(mo, mo_r, N * 4)) { fprintf(stderr, "morton3D differs\n"); return 1; }
    /* sample generation + compositing of the packed batch */
    float *xyz = calloc((size_t)M * 3, 4), *dir = calloc((size_t)M * 3, 4), *del = calloc((size_t)M * 2, 4), *noise = calloc(N, 4);
    int32_t *rays = malloc(3 * N * 4), counter[2] = {0, 0};
    oracle_march_rays_train(o, d, bits_ref, 2.0f, 0.0f, MAXS, N, C, H, M, near_r, far_r, xyz, dir, del, rays, counter, noise);
    float *d_xyz = NULL, *d_dir = NULL, *d_del = NULL, *d_noise = to_dev(noise, N * 4);
    int32_t *d_rays = NULL, *d_counter = NULL, zero2[2] = {0, 0};
    HIP(hipMalloc((void**)&d_xyz, (size_t)M * 12)); HIP(hipMalloc((void**)&d_dir, (size_t)M * 12)); HIP(hipMalloc((void**)&d_del, (size_t)M * 8));
    HIP(hipMemset(d_xyz, 0, (size_t)M * 12)); HIP(hipMemset(d_dir, 0, (size_t)M * 12)); HIP(hipMemset(d_del, 0, (size_t)M * 8));
    HIP(hipMalloc((void**)&d_rays, 3 * N * 4)); d_counter = to_dev(zero2, 8);
    NVSF(nvsf_march_rays_train(d_o, d_d, d_bits, 2.0f, 0.0f, MAXS, N, C, H, M, d_near, d_far, d_xyz, d_dir, d_del, d_rays, d_counter, d_noise, stream));
    HIP(hipStreamSynchronize(stream));
    int32_t counter_g[2], *rays_g = malloc(3 * N * 4);
    HIP(hipMemcpy(counter_g, d_counter, 8, hipMemcpyDeviceToHost)); HIP(hipMemcpy(rays_g, d_rays, 3 * N * 4, hipMemcpyDeviceToHost));
    if (counter_g[0] != counter[0] || counter_g[1] != counter[1] || memcmp(rays_g, rays, 3 * N * 4)) { fprintf(stderr, "march_rays_train: counts differ\n"); return 1; }
    const uint32_t m = (uint32_t)counter[0];
    float *xyz_g = malloc((size_t)m * 12 + 4), *del_g = malloc((size_t)m * 8 + 4);
    HIP(hipMemcpy(xyz_g, d_xyz, (size_t)m * 12, hipMemcpyDeviceToHost)); HIP(hipMemcpy(del_g, d_del, (size_t)m * 8, hipMemcpyDeviceToHost));
    if (memcmp(xyz_g, xyz, (size_t)m * 12) || memcmp(del_g, del, (size_t)m * 8)) { fprintf(stderr, "march_rays_train: samples differ\n"); return 1; }
    float *sig = malloc((size_t)m * 4 + 4), *rgb = malloc((size_t)m * 12 + 4);
    for (uint32_t i = 0; i < m; ++i) { sig[i] = frand() * 40.0f; rgb[3 * i] = frand(); rgb[3 * i + 1] = frand(); rgb[3 * i + 2] = frand(); }
    float *ws_r = malloc(N * 4), *dp_r = malloc(N * 4), *im_r = malloc(3 * N * 4), *ws = malloc(N * 4), *dp = malloc(N * 4), *im = malloc(3 * N * 4);
    oracle_composite_rays_train_forward(sig, rgb, del, rays, m, N, 1e-4f, ws_r, dp_r, im_r);
    float *d_sig = to_dev(sig, (size_t)m * 4 + 4), *d_rgb = to_dev(rgb, (size_t)m * 12 + 4), *d_ws = NULL, *d_dp = NULL, *d_im = NULL;
    HIP(hipMalloc((void**)&d_ws, N * 4)); HIP(hipMalloc((void**)&d_dp, N * 4)); HIP(hipMalloc((void**)&d_im, 3 * N * 4));
    NVSF(nvsf_composite_rays_train_forward(d_sig, d_rgb, d_del, d_rays, m, N, 1e-4f, d_ws, d_dp, d_im, stream));
    HIP(hipStreamSynchronize(stream));
    HIP(hipMemcpy(ws, d_ws, N * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(dp, d_dp, N * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(im, d_im, 3 * N * 4, hipMemcpyDeviceToHost));
    float worst = 0;
    for (int i = 0; i < N; ++i) {
        float e = __builtin_fabsf(ws[i] - ws_r[i]); if (e > worst) worst = e;
        e = __builtin_fabsf(dp[i] - dp_r[i]); if (e > worst) worst = e;
        for (int k = 0; k < 3; ++k) { e = __builtin_fabsf(im[3 * i + k] - im_r[3 * i + k]); if (e > worst) worst = e; }
    }
    if (!(worst <= 1e-5f)) { fprintf(stderr, "composite_rays_train_forward: max error %g\n", worst); return 1; }
    /* argument checking: a null pointer is rejected with a status, not a fault */
    if (nvsf_near_far_from_aabb(NULL, d_d, d_aabb, N, 0.05f, d_near, d_far, stream) == 0) { fprintf(stderr, "null pointer accepted\n"); return 1; }
    printf("abi_smoke OK: %u rays, %u samples, compositor max error %.2e\n", (unsigned)N, m, worst);
    return 0;
}
