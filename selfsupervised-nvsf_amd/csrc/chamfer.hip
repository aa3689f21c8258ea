// Chamfer (nearest-neighbour) distance between two point clouds for gfx950, forward + backward.
// Reference: /root/reference/nvsf/nerf/chamfer3D/chamfer3D.cu:9-138 (NmDistanceKernel, <<<(32,16,1),512>>>, one
// 512-point shared tile) and :167-221 (NmDistanceGradKernel); Python wrapper chamfer3D/dist_chamfer_3D.py:42-97.
// Used by the LiDAR training loss (trainer.py:232-233,252-267) and the CD / F-score meter (error_matrices.py:322-343).
//
// Design: a 4096-point ray batch gives only 16 workgroups of queries, so the TARGET cloud is additionally split
// over blockIdx.z; every (query block, target slice) scans its slice through a 1024-point LDS tile and merges into a
// packed (distance bits << 32 | index) word with one 64-bit atomicMin per query -- squared distances are
// non-negative, so their fp32 bit patterns order like unsigned integers and ties resolve to the lowest index (the
// first occurrence, as a sequential scan would).  A second tiny kernel unpacks the words.
#include "common.h"
#include <math.h>

namespace {
constexpr int kBlock = 256;
constexpr int kTile = 1024;

__global__ __launch_bounds__(kBlock) void k_chamfer_scan(const float* __restrict__ q, uint32_t n, const float* __restrict__ t, uint32_t m,
                                                         uint32_t slice_len, unsigned long long* __restrict__ best) {
    __shared__ float tile[kTile * 3];
    const uint32_t b = blockIdx.y;
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t s_begin = blockIdx.z * slice_len, s_end = min(m, s_begin + slice_len);
    float x1 = 0, y1 = 0, z1 = 0;
    if (j < n) {
        const float* p = q + ((size_t)b * n + j) * 3;
        x1 = p[0]; y1 = p[1]; z1 = p[2];
    }
    float bd = INFINITY;
    uint32_t bi = 0;
    for (uint32_t k0 = s_begin; k0 < s_end; k0 += kTile) {
        const uint32_t cnt = min((uint32_t)kTile, s_end - k0);
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < cnt * 3; e += kBlock) tile[e] = t[((size_t)b * m + k0) * 3 + e];
        __syncthreads();
        for (uint32_t k = 0; k < cnt; ++k) {
            const float dx = tile[3 * k] - x1, dy = tile[3 * k + 1] - y1, dz = tile[3 * k + 2] - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < bd) { bd = d; bi = k0 + k; }
        }
    }
    if (j < n && s_begin < s_end) {
        const unsigned long long word = ((unsigned long long)__float_as_uint(bd) << 32) | bi;
        atomicMin(best + (size_t)b * n + j, word);
    }
}

__global__ __launch_bounds__(kBlock) void k_chamfer_unpack(const unsigned long long* __restrict__ best, uint32_t total,
                                                           float* __restrict__ dist, int* __restrict__ idx) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const unsigned long long w = best[i];
    dist[i] = __uint_as_float((uint32_t)(w >> 32));
    idx[i] = (int)(uint32_t)w;
}

// grad_a[j] += 2 g (a_j - b_nn(j)),  grad_b[nn(j)] -= the same  (chamfer3D.cu:167-195).  grad_a / grad_b may be NULL (that cloud needs
// no gradient); STORE_A: grad_a[j] is WRITTEN -- point j of the query cloud is this thread's alone -- so a caller that runs this
// direction first needs no zero fill of grad_a.
template <bool STORE_A>
__global__ __launch_bounds__(kBlock) void k_chamfer_grad(const float* __restrict__ a, uint32_t n, const float* __restrict__ bpts, uint32_t m,
                                                         const float* __restrict__ grad_dist, const int* __restrict__ idx,
                                                         float* __restrict__ grad_a, float* __restrict__ grad_b, uint32_t B) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (size_t)B * n) return;
    const uint32_t bb = (uint32_t)(i / n);
    const int j2 = idx[i];
    const float g = grad_dist[i] * 2.0f;
    const float* p = a + i * 3;
    const float* r = bpts + ((size_t)bb * m + j2) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = g * (p[c] - r[c]);
        if (grad_a) {
            if constexpr (STORE_A) grad_a[i * 3 + c] = v;
            else atomicAdd(grad_a + i * 3 + c, v);
        }
        if (grad_b) atomicAdd(grad_b + ((size_t)bb * m + j2) * 3 + c, -v);
    }
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

static int scan_one_direction(const float* q, uint32_t n, const float* t, uint32_t m, uint32_t B, unsigned long long* ws, float* dist,
                              int* idx, hipStream_t stream) {
    if (n == 0) return NVSF_OK;
    hipError_t e = hipMemsetAsync(ws, 0xFF, (size_t)B * n * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return (int)e;
    const uint32_t qblocks = cdiv(n, kBlock);
    uint32_t slices = 1;
    while (qblocks * B * slices < 1024 && slices * kTile < m) slices *= 2;
    const uint32_t slice_len = (m + slices - 1) / slices;
    hipLaunchKernelGGL(k_chamfer_scan, dim3(qblocks, B, slices), dim3(kBlock), 0, stream, q, n, t, m, slice_len, ws);
    hipLaunchKernelGGL(k_chamfer_unpack, dim3(cdiv((unsigned long long)B * n, kBlock)), dim3(kBlock), 0, stream, ws, B * n, dist, idx);
    return nvsf_launch_status();
}

// workspace: B * max(n, m) 64-bit words
NVSF_API int nvsf_chamfer_forward(const float* xyz1, const float* xyz2, uint32_t B, uint32_t n, uint32_t m, float* dist1, float* dist2,
                                  int32_t* idx1, int32_t* idx2, void* workspace_u64, hipStream_t stream) {
    if (B == 0 || (n == 0 && m == 0)) return NVSF_OK;
    REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2 && workspace_u64 && n > 0 && m > 0);
    REQUIRE((reinterpret_cast<uintptr_t>(workspace_u64) & 7u) == 0 && (unsigned long long)B * (n > m ? n : m) < (1ull << 31));
    unsigned long long* ws = reinterpret_cast<unsigned long long*>(workspace_u64);
    int st = scan_one_direction(xyz1, n, xyz2, m, B, ws, dist1, idx1, stream);
    if (st != NVSF_OK) return st;
    return scan_one_direction(xyz2, m, xyz1, n, B, ws, dist2, idx2, stream);
}

// grad_xyz2 must be zero-initialised by the caller (dist_chamfer_3D.py:79-80) unless it is NULL (the second cloud needs no gradient:
// the measured point cloud of the training loss); grad_xyz1 may hold anything: the first direction writes every row of it before the
// second one adds.  grad_xyz1 NULL: only the second cloud's gradient is formed (then grad_xyz2 is written first / added second).
NVSF_API int nvsf_chamfer_backward(const float* xyz1, const float* xyz2, uint32_t B, uint32_t n, uint32_t m, const float* grad_dist1,
                                   const float* grad_dist2, const int32_t* idx1, const int32_t* idx2, float* grad_xyz1, float* grad_xyz2,
                                   hipStream_t stream) {
    if (B == 0 || n == 0 || m == 0) return NVSF_OK;
    REQUIRE(xyz1 && xyz2 && grad_dist1 && grad_dist2 && idx1 && idx2 && (grad_xyz1 || grad_xyz2));
    hipLaunchKernelGGL(k_chamfer_grad<true>, dim3(cdiv((unsigned long long)B * n, kBlock)), dim3(kBlock), 0, stream, xyz1, n, xyz2, m, grad_dist1, idx1,
                       grad_xyz1, grad_xyz2, B);
    hipLaunchKernelGGL(k_chamfer_grad<false>, dim3(cdiv((unsigned long long)B * m, kBlock)), dim3(kBlock), 0, stream, xyz2, m, xyz1, n, grad_dist2, idx2,
                       grad_xyz2, grad_xyz1, B);
    return nvsf_launch_status();
}
