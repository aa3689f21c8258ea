// Ray generation from pixel indices for gfx950: the step immediately in front of the render.
// Reference: /root/reference/nvsf/nerf/dataset/dataset_utils.py:369-536 (get_lidar_rays) and :539-687 (get_rays).
// The reference builds full H x W meshgrids (67 980 / 529 408 floats x 2) on every step, gathers the N sampled
// pixels, and runs ~15 elementwise torch kernels + a batched matmul.  Here one thread = one sampled pixel:
//   LiDAR : beta = -(i - W/2)/W * fov_hoz/180*pi, alpha = (fov_up - j/H*fov)/180*pi,
//           dir = (cos a cos b, cos a sin b, sin a) (not normalised again), rays_d = dir R^T, rays_o = pose[:3,3]
//   camera: pixel centre +0.5, dir = ((i-cx)/fx, (j-cy)/fy, 1) normalised, rays_d = dir R^T
// Scalar arithmetic follows the reference's operation order in fp32 (torch divides a tensor by a Python scalar as a
// multiplication by the fp32 reciprocal).
#include "common.h"
#include <math.h>

namespace {
constexpr int kBlock = 256;

struct Pose { float r[9]; float t[3]; };

__device__ __forceinline__ void rotate_store(const Pose& p, float dx, float dy, float dz, size_t n, float* __restrict__ rays_o,
                                             float* __restrict__ rays_d) {
    // directions @ R^T  ->  d_world[k] = sum_c dir[c] * R[k][c]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_d[3 * n + k] = (dx * p.r[3 * k] + dy * p.r[3 * k + 1]) + dz * p.r[3 * k + 2];
        rays_o[3 * n + k] = p.t[k];
    }
}

__global__ __launch_bounds__(kBlock) void k_lidar_rays(const float* __restrict__ pose, const long long* __restrict__ inds, uint32_t N,
                                                       uint32_t H, uint32_t W, float fov_up, float fov, float fov_hoz,
                                                       float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    Pose p;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.r[3 * k] = pose[4 * k]; p.r[3 * k + 1] = pose[4 * k + 1]; p.r[3 * k + 2] = pose[4 * k + 2];
        p.t[k] = pose[4 * k + 3];
    }
    const long long id = inds ? inds[n] : (long long)n;
    const float i = (float)(id % W), j = (float)(id / W);
    const float kPi = 3.14159265358979323846f;
    const float beta = (((-(i - (float)W / 2.0f)) * (1.0f / (float)W)) * fov_hoz) * (1.0f / 180.0f) * kPi;
    const float alpha = ((fov_up - (j * (1.0f / (float)H)) * fov) * (1.0f / 180.0f)) * kPi;
    const float ca = cosf(alpha), sa = sinf(alpha), cb = cosf(beta), sb = sinf(beta);
    rotate_store(p, ca * cb, ca * sb, sa, n, rays_o, rays_d);
}

__global__ __launch_bounds__(kBlock) void k_camera_rays(const float* __restrict__ pose, const long long* __restrict__ inds, uint32_t N,
                                                        uint32_t W, float fx, float fy, float cx, float cy, float* __restrict__ rays_o,
                                                        float* __restrict__ rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    Pose p;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.r[3 * k] = pose[4 * k]; p.r[3 * k + 1] = pose[4 * k + 1]; p.r[3 * k + 2] = pose[4 * k + 2];
        p.t[k] = pose[4 * k + 3];
    }
    const long long id = inds ? inds[n] : (long long)n;
    const float i = (float)(id % W) + 0.5f, j = (float)(id / W) + 0.5f;
    const float xs = (i - cx) / fx, ys = (j - cy) / fy, zs = 1.0f;
    const float nrm = sqrtf((xs * xs + ys * ys) + zs * zs);
    rotate_store(p, xs / nrm, ys / nrm, zs / nrm, n, rays_o, rays_d);
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_lidar_rays(const float* pose44, const int64_t* inds, uint32_t N, uint32_t H, uint32_t W, float fov_up, float fov,
                             float fov_hoz, float* rays_o, float* rays_d, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(pose44 && rays_o && rays_d && H > 0 && W > 0);
    hipLaunchKernelGGL(k_lidar_rays, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, pose44, reinterpret_cast<const long long*>(inds), N, H, W,
                       fov_up, fov, fov_hoz, rays_o, rays_d);
    return nvsf_launch_status();
}

NVSF_API int nvsf_camera_rays(const float* pose44, const int64_t* inds, uint32_t N, uint32_t W, float fx, float fy, float cx, float cy,
                              float* rays_o, float* rays_d, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(pose44 && rays_o && rays_d && W > 0 && fx != 0.0f && fy != 0.0f);
    hipLaunchKernelGGL(k_camera_rays, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, pose44, reinterpret_cast<const long long*>(inds), N, W, fx,
                       fy, cx, cy, rays_o, rays_d);
    return nvsf_launch_status();
}
