// K-planes space-time encoder (Planes4D) for gfx950: forward, and backward wrt planes and coordinates.
// Reference: /root/reference/nvsf/nerf/models/planes_field.py:55-140 (24 F.grid_sample launches + products +
// concatenation per call; bilinear, align_corners=True, padding_mode='border').
//
// Layout: the reference keeps each plane channel-first [C][H][W]; the HIP path reads a CHANNEL-LAST copy
// [H][W][C=8] (refreshed by the host when the parameters change), so one texel is 32 contiguous bytes = two
// 16-byte loads instead of eight 4-byte loads at a stride of H*W*4 bytes.  All 24 planes of an encoder are
// one buffer, in (scale, pair) order with pair = (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); static feature =
// (xy * xz) * yz, dynamic = (xt * yt) * zt, per scale, concatenated over scales.
// The whole table (2.2 M parameters = 8.7 MB) lives in L2 / Infinity Cache; one thread = one (sample, scale).
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {
constexpr int kBlock = 256;
constexpr int kC = 8;          // features per plane (n_features_per_level_plane)
constexpr int kMaxScales = 8;

struct PlaneMeta {
    uint32_t res[kMaxScales][4];   // per scale: resolution of x, y, z, t
    uint32_t off[kMaxScales][6];   // float offset of plane (scale, pair) in the channel-last buffer
    uint32_t n_scales;
};

__device__ __constant__ const int kPa[6] = {0, 0, 0, 1, 1, 2};
__device__ __constant__ const int kPb[6] = {1, 2, 3, 2, 3, 3};

struct Tap {
    uint32_t i00, i01, i10, i11;   // texel indices (row-major, clamped into the image)
    float nw, ne, sw, se;          // bilinear weights, torch naming
    float gx, gy;                  // d ix / d p_a, d iy / d p_b (0 where the border clamp is active)
    float ix_f, iy_f, x0, y0;
};

__device__ __forceinline__ Tap make_tap(float pa, float pb, uint32_t W, uint32_t H) {
    Tap t;
    float ix = ((pa * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((pb * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(H - 1);
    // torch's border clamp passes no gradient AT the border either (clip_coordinates_set_grad: in <= 0, in >= max)
    t.gx = (ix <= 0.0f || ix >= (float)(W - 1)) ? 0.0f : (float)(W - 1);
    t.gy = (iy <= 0.0f || iy >= (float)(H - 1)) ? 0.0f : (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    t.nw = (x1 - ix) * (y1 - iy);
    t.ne = (ix - x0) * (y1 - iy);
    t.sw = (x1 - ix) * (iy - y0);
    t.se = (ix - x0) * (iy - y0);
    const uint32_t X0 = (uint32_t)x0, Y0 = (uint32_t)y0;
    const uint32_t X1 = X0 + 1 < W ? X0 + 1 : W - 1, Y1 = Y0 + 1 < H ? Y0 + 1 : H - 1;  // out-of-image taps carry weight 0
    t.i00 = Y0 * W + X0; t.i01 = Y0 * W + X1; t.i10 = Y1 * W + X0; t.i11 = Y1 * W + X1;
    t.ix_f = ix; t.iy_f = iy; t.x0 = x0; t.y0 = y0;
    return t;
}

__device__ __forceinline__ void load_texel(const float* __restrict__ plane, uint32_t idx, float (&v)[kC]) {
    const float4 a = reinterpret_cast<const float4*>(plane + (size_t)idx * kC)[0];
    const float4 b = reinterpret_cast<const float4*>(plane + (size_t)idx * kC)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ void interp(const float* __restrict__ plane, const Tap& t, float (&out)[kC]) {
    float a[kC], b[kC], c[kC], d[kC];
    load_texel(plane, t.i00, a);
    load_texel(plane, t.i01, b);
    load_texel(plane, t.i10, c);
    load_texel(plane, t.i11, d);
#pragma unroll
    for (int k = 0; k < kC; ++k) out[k] = ((a[k] * t.nw + b[k] * t.ne) + c[k] * t.sw) + d[k] * t.se;
}

// want: bit 0 static, bit 1 dynamic
__global__ __launch_bounds__(kBlock) void k_planes_fwd(const float* __restrict__ xt, uint32_t M, const float* __restrict__ planes,
                                                       PlaneMeta meta, int want, float* __restrict__ out_static,
                                                       float* __restrict__ out_dynamic) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x, s = blockIdx.y;
    if (m >= M) return;
    const float4 p4 = reinterpret_cast<const float4*>(xt)[m];
    const float p[4] = {p4.x, p4.y, p4.z, p4.w};
    float fs[kC], fd[kC];
    bool first_s = true, first_d = true;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int a = kPa[q], b = kPb[q];
        const bool is_time = b == 3;
        if ((is_time && !(want & 2)) || (!is_time && !(want & 1))) continue;
        const Tap t = make_tap(p[a], p[b], meta.res[s][a], meta.res[s][b]);
        float v[kC];
        interp(planes + meta.off[s][q], t, v);
        if (is_time) {
#pragma unroll
            for (int k = 0; k < kC; ++k) fd[k] = first_d ? v[k] : fd[k] * v[k];
            first_d = false;
        } else {
#pragma unroll
            for (int k = 0; k < kC; ++k) fs[k] = first_s ? v[k] : fs[k] * v[k];
            first_s = false;
        }
    }
    const uint32_t stride = meta.n_scales * kC;
    if (want & 1) {
        float4* o = reinterpret_cast<float4*>(out_static + (size_t)m * stride + s * kC);
        o[0] = make_float4(fs[0], fs[1], fs[2], fs[3]);
        o[1] = make_float4(fs[4], fs[5], fs[6], fs[7]);
    }
    if (want & 2) {
        float4* o = reinterpret_cast<float4*>(out_dynamic + (size_t)m * stride + s * kC);
        o[0] = make_float4(fd[0], fd[1], fd[2], fd[3]);
        o[1] = make_float4(fd[4], fd[5], fd[6], fd[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// Forward along the rays (the form nvsf_planes_fwd / nvsf_planes_multi_fwd launch; k_planes_fwd above is its test reference).
//
// k_planes_fwd gathers four 32-byte texels per (sample, scale, plane): 3 KB per sample, from a table that sits in L2, i.e.
// the kernel runs at the rate of the vector L1's data path (64 B/clk/CU) no matter how often neighbouring samples ask for
// the same texels -- and rows of the input are consecutive samples of a ray, 0.02 - 0.33 texels apart.  Here an item =
// (chunk of kRun consecutive rows, evaluation) is walked IN ORDER by 8 adjacent lanes (lane = scale x channel half): a lane
// keeps the four half-texels (4 x 16 B) of the current cell of each of its three planes in registers and gathers again
// only when that plane's cell changes -- 4 x to 16 x fewer gather instructions on camera / LiDAR rays.  Per sample the
// eight lanes of an item store one whole 128-byte output row.  Same tap arithmetic, same interpolation and product order
// per channel as k_planes_fwd: bit-identical features (tests/test_dynamic_gpu.py).
//
// An "evaluation" is one group of three planes (static xy, xz, yz | dynamic xt, yt, zt) at positions x (+ an offset read
// from another buffer: the scene flow of the neighbour frames, network_dynamic.py:242-271) with the time either taken from
// column 3 of the positions or given as a scalar -- so the three dynamic evaluations of a density query and the static one
// are ONE launch and no [M,4] position copies are built for the neighbours.
constexpr int kRun = 64;
constexpr int kMaxEval = 4;
struct PlaneEvals {
    const float* x;            // [M, x_stride] positions in [0,1]
    uint32_t x_stride;
    int n;
    int grp[kMaxEval];         // 0: static planes (pairs 0, 1, 3), 1: time planes (pairs 2, 4, 5)
    const float* off[kMaxEval];  // optional offsets added to x (fp32 add, as torch.add): row stride / first column below
    uint32_t off_stride[kMaxEval], off_col[kMaxEval];
    float t[kMaxEval];         // time coordinate ...
    int t_from_x[kMaxEval];    // ... unless taken from column 3 of x
    float* out[kMaxEval];      // [M, n_scales * 8]
    int blend;                 // 2: as 1 with fp16 output rows; 1: the evaluations are (static, dynamic, dynamic at neighbour 1, dynamic at neighbour 2) and out[1]
                               // receives 0.5 d + 0.25 (d1 + d2) (network_dynamic.py:273); out[2], out[3] are not written
};

__global__ __launch_bounds__(kBlock) void k_planes_fwd_runs(PlaneEvals ev, uint32_t M, const float* __restrict__ planes, PlaneMeta meta) {
    __shared__ uint32_t s_res[kMaxScales][4], s_off[kMaxScales][6];
    // positions of the block's 32 items (x + offset, time), staged once: a row per item, padded by one float4 so that the eight
    // items of a wave read different banks (the eight lanes of an item read the same word: a broadcast)
    __shared__ float4 s_pos[kBlock / 8][kRun + 1];
    if (threadIdx.x < kMaxScales * 4) s_res[threadIdx.x >> 2][threadIdx.x & 3] = meta.res[threadIdx.x >> 2][threadIdx.x & 3];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kMaxScales * 6) {
        const uint32_t i = threadIdx.x - 64;
        s_off[i / 6][i % 6] = meta.off[i / 6][i % 6];
    }
    constexpr uint32_t lanes_per_item = 8;  // four scales x two channel halves (the launcher takes other scale counts elsewhere)
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t item = tid / lanes_per_item, li = tid - item * lanes_per_item, item_local = threadIdx.x / lanes_per_item;
    const uint32_t s = li >> 1, half = li & 1u;
    const uint32_t n_chunks = (M + kRun - 1) / kRun;
    uint32_t chunk = item / (uint32_t)ev.n;
    const uint32_t e = item - chunk * (uint32_t)ev.n;  // the evaluations of a chunk are neighbours
    const bool active = chunk < n_chunks;
    if (!active) chunk = 0;
    {   // stage the item's positions: lane li takes rows li, li + 8, ... (independent loads, all in flight together -- read one
        // row per step inside the loop instead and every step waits for a dependent global load)
        const float* soff = ev.off[e];
        const uint32_t so_stride = ev.off_stride[e], so_col = ev.off_col[e];
        const bool tx = ev.t_from_x[e] != 0;
        const float tc = ev.t[e];
        const uint32_t mb = chunk * kRun;
        for (uint32_t k = li; k < (uint32_t)kRun; k += lanes_per_item) {
            const uint32_t m = mb + k;
            if (active && m < M) {
                const float* px = ev.x + (size_t)m * ev.x_stride;
                float4 p = make_float4(px[0], px[1], px[2], tx ? px[3] : tc);
                if (soff) {
                    const float* po = soff + (size_t)m * so_stride + so_col;
                    p.x = p.x + po[0]; p.y = p.y + po[1]; p.z = p.z + po[2];
                }
                s_pos[item_local][k] = p;
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const int grp = ev.grp[e];
    const int pairs[3] = {grp == 0 ? 0 : 2, grp == 0 ? 1 : 4, grp == 0 ? 3 : 5};
    uint32_t W[3], H[3];
    const float* base[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        W[j] = s_res[s][kPa[pairs[j]]];
        H[j] = s_res[s][kPb[pairs[j]]];
        base[j] = planes + s_off[s][pairs[j]] + half * 4u;
    }
    const bool t_from_x = ev.t_from_x[e] != 0;
    const float t_const = ev.t[e];
    float* out = ev.out[e] + (size_t)s * kC + half * 4u;
    const uint32_t stride = meta.n_scales * kC;
    // One coordinate axis of make_tap: u = ((p * 2 - 1 + 1) / 2) * (R - 1) clamped into the image, its cell and the two linear
    // weights (x1 - u, u - x0).  A plane's bilinear weights are products of two axes' weights -- the same products, in the same
    // order, as make_tap forms them -- so the three planes of a group share three (static) or four (dynamic; the time axis is
    // constant along the chunk) axis evaluations instead of six.
    struct Axis { uint32_t c0, c1; float w0, w1; };
    auto axis = [](float pv, uint32_t R) {
        Axis a;
        float u = ((pv * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(R - 1);
        u = fminf((float)(R - 1), fmaxf(u, 0.0f));
        const float f0 = floorf(u), f1 = f0 + 1.0f;
        a.w0 = f1 - u;
        a.w1 = u - f0;
        a.c0 = (uint32_t)f0;
        a.c1 = a.c0 + 1 < R ? a.c0 + 1 : R - 1;
        return a;
    };
    const Axis at = axis(t_const, s_res[s][3]);  // used by the dynamic group unless the time comes from the rows
    uint32_t key[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    float4 tex[3][4];
    const uint32_t m0 = chunk * kRun, m1 = m0 + kRun < M ? m0 + kRun : M;
    for (uint32_t m = m0; m < m1; ++m) {
        const float4 pp = s_pos[item_local][m - m0];
        const float p[3] = {pp.x, pp.y, pp.z};
        Axis ax[4];
#pragma unroll
        for (int d = 0; d < 3; ++d) ax[d] = axis(p[d], s_res[s][d]);
        ax[3] = t_from_x ? axis(pp.w, s_res[s][3]) : at;
        float4 f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const Axis& A = ax[grp == 0 ? (j == 2 ? 1 : 0) : j];   // pairs (x,y) (x,z) (y,z) | (x,t) (y,t) (z,t): first axis = image x
            const Axis& B = ax[grp == 0 ? (j == 0 ? 1 : 2) : 3];   //                                            second axis = image y
            const float nw = A.w0 * B.w0, ne = A.w1 * B.w0, sw = A.w0 * B.w1, se = A.w1 * B.w1;
            const uint32_t i00 = B.c0 * W[j] + A.c0;
            if (i00 != key[j]) {  // (X0, Y0) fixes all four taps: gather the cell's half-texels again
                key[j] = i00;
                const uint32_t i01 = B.c0 * W[j] + A.c1, i10 = B.c1 * W[j] + A.c0, i11 = B.c1 * W[j] + A.c1;
                tex[j][0] = *reinterpret_cast<const float4*>(base[j] + (size_t)i00 * kC);
                tex[j][1] = *reinterpret_cast<const float4*>(base[j] + (size_t)i01 * kC);
                tex[j][2] = *reinterpret_cast<const float4*>(base[j] + (size_t)i10 * kC);
                tex[j][3] = *reinterpret_cast<const float4*>(base[j] + (size_t)i11 * kC);
            }
            float4 v;
            v.x = ((tex[j][0].x * nw + tex[j][1].x * ne) + tex[j][2].x * sw) + tex[j][3].x * se;
            v.y = ((tex[j][0].y * nw + tex[j][1].y * ne) + tex[j][2].y * sw) + tex[j][3].y * se;
            v.z = ((tex[j][0].z * nw + tex[j][1].z * ne) + tex[j][2].z * sw) + tex[j][3].z * se;
            v.w = ((tex[j][0].w * nw + tex[j][1].w * ne) + tex[j][2].w * sw) + tex[j][3].w * se;
            if (j == 0) f = v;
            else { f.x = f.x * v.x; f.y = f.y * v.y; f.z = f.z * v.z; f.w = f.w * v.w; }
        }
        if (ev.blend) {
            // the four evaluations of a chunk sit on 32 adjacent lanes (8 each, same (scale, half) at the same offset): the base
            // evaluation's lanes fetch the neighbours' features of this row from the lanes 8 and 16 further on
            const int src1 = (int)((threadIdx.x & 63u) + 8u) & 63, src2 = (int)((threadIdx.x & 63u) + 16u) & 63;
            float4 a, b;
            a.x = __shfl(f.x, src1); a.y = __shfl(f.y, src1); a.z = __shfl(f.z, src1); a.w = __shfl(f.w, src1);
            b.x = __shfl(f.x, src2); b.y = __shfl(f.y, src2); b.z = __shfl(f.z, src2); b.w = __shfl(f.w, src2);
            if (e == 1u) {
                f.x = 0.5f * f.x + 0.25f * (a.x + b.x);
                f.y = 0.5f * f.y + 0.25f * (a.y + b.y);
                f.z = 0.5f * f.z + 0.25f * (a.z + b.z);
                f.w = 0.5f * f.w + 0.25f * (a.w + b.w);
            }
            if (e >= 2u) continue;
        }
        if (ev.blend == 2) {  // fp16 rows [M, n_scales * 8]: what the density kernel rounds the features to anyway
            typedef _Float16 half4_v __attribute__((ext_vector_type(4)));
            half4_v h;
            h[0] = (_Float16)f.x; h[1] = (_Float16)f.y; h[2] = (_Float16)f.z; h[3] = (_Float16)f.w;
            *reinterpret_cast<half4_v*>(reinterpret_cast<_Float16*>(ev.out[e]) + (size_t)m * stride + (size_t)s * kC + half * 4u) = h;
        } else {
            *reinterpret_cast<float4*>(out + (size_t)m * stride) = f;
        }
    }
}

// Backward: one thread = one sample (loops over scales so that grad_xt needs no atomics); plane gradients are
// fp32 atomics into the channel-last gradient buffer.
__global__ __launch_bounds__(kBlock) void k_planes_bwd(const float* __restrict__ xt, uint32_t M, const float* __restrict__ planes,
                                                       PlaneMeta meta, int want, const float* __restrict__ g_static,
                                                       const float* __restrict__ g_dynamic, float* __restrict__ g_planes,
                                                       float* __restrict__ g_xt) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    const float4 p4 = reinterpret_cast<const float4*>(xt)[m];
    const float p[4] = {p4.x, p4.y, p4.z, p4.w};
    float gp[4] = {0, 0, 0, 0};
    const uint32_t stride = meta.n_scales * kC;
    for (uint32_t s = 0; s < meta.n_scales; ++s) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {  // 0: static planes (pairs 0,1,3), 1: time planes (pairs 2,4,5)
            if (!(want & (1 << grp))) continue;
            const float* gout = (grp == 0 ? g_static : g_dynamic) + (size_t)m * stride + s * kC;
            const int pairs[3] = {grp == 0 ? 0 : 2, grp == 0 ? 1 : 4, grp == 0 ? 3 : 5};
            Tap t[3];
            float v[3][kC], dvx[3][kC], dvy[3][kC];  // interpolated values and their derivatives along the two image axes
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int a = kPa[pairs[j]], b = kPb[pairs[j]];
                t[j] = make_tap(p[a], p[b], meta.res[s][a], meta.res[s][b]);
                // the four texels are gathered ONCE: value and both derivatives come from the same registers (the first form
                // interpolated in a helper and loaded the texels a second time for the derivatives: twice the gathers of a
                // gather-bound kernel)
                const float* plane = planes + meta.off[s][pairs[j]];
                float tex00[kC], tex01[kC], tex10[kC], tex11[kC];
                load_texel(plane, t[j].i00, tex00);
                load_texel(plane, t[j].i01, tex01);
                load_texel(plane, t[j].i10, tex10);
                load_texel(plane, t[j].i11, tex11);
                const float wx1 = t[j].ix_f - t[j].x0, wx0 = 1.0f - wx1, wy1 = t[j].iy_f - t[j].y0, wy0 = 1.0f - wy1;
#pragma unroll
                for (int k = 0; k < kC; ++k) {
                    v[j][k] = ((tex00[k] * t[j].nw + tex01[k] * t[j].ne) + tex10[k] * t[j].sw) + tex11[k] * t[j].se;  // = interp()
                    dvx[j][k] = (tex01[k] - tex00[k]) * wy0 + (tex11[k] - tex10[k]) * wy1;
                    dvy[j][k] = (tex10[k] - tex00[k]) * wx0 + (tex11[k] - tex01[k]) * wx1;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = pairs[j], a = kPa[q], b = kPb[q];
                float* gpl = g_planes ? g_planes + meta.off[s][q] : nullptr;
                float dix = 0.0f, diy = 0.0f;
#pragma unroll
                for (int k = 0; k < kC; ++k) {
                    const float other = v[(j + 1) % 3][k] * v[(j + 2) % 3][k];
                    const float gv = gout[k] * other;  // d L / d (interpolated value of this plane, channel k)
                    if (gpl) {
                        atomicAdd(gpl + (size_t)t[j].i00 * kC + k, gv * t[j].nw);
                        atomicAdd(gpl + (size_t)t[j].i01 * kC + k, gv * t[j].ne);
                        atomicAdd(gpl + (size_t)t[j].i10 * kC + k, gv * t[j].sw);
                        atomicAdd(gpl + (size_t)t[j].i11 * kC + k, gv * t[j].se);
                    }
                    dix += gv * dvx[j][k];
                    diy += gv * dvy[j][k];
                }
                gp[a] += dix * t[j].gx;
                gp[b] += diy * t[j].gy;
            }
        }
    }
    if (g_xt) reinterpret_cast<float4*>(g_xt)[m] = make_float4(gp[0], gp[1], gp[2], gp[3]);
}

// Plane gradients with run merging (the form nvsf_planes_bwd launches; k_planes_bwd above then only produces grad_xt).
// One atomic per (sample, texel, channel) is 1.2 G adds per 1.6 M samples into tables of 8 K - 0.5 M floats, and the time
// planes receive all of them in the two rows around the frame's t: float atomics execute at the memory side and many
// adders on one row are the slowest case (MI355X_MICROARCH.md, Global float atomics) -- 171 ms per call.  Rows of xt are
// consecutive samples of a ray, 0.07 - 0.3 texels apart even at the finest scale, so an item = (chunk of `run` rows,
// scale, static | time group) keeps the sums of the current texel quad of each of its three planes in registers and adds
// them only when that plane's quad changes.  32 lanes per item: lane = (texel of the quad, channel), i.e. a flush is one
// atomic instruction covering four whole 32-byte texels.  The interpolated values the products need are rebuilt from the
// lanes' own gathers (butterfly sum over the four texel lanes).
__global__ __launch_bounds__(kBlock) void k_planes_bwd_runs(const float* __restrict__ xt, uint32_t M, const float* __restrict__ planes,
                                                            PlaneMeta meta, int want, const float* __restrict__ g_static,
                                                            const float* __restrict__ g_dynamic, float* __restrict__ g_planes,
                                                            uint32_t run) {
    // taps of 32 samples x 3 planes of an item: {nw, ne, sw, se, i00, i01, i10, i11}.  All 32 lanes of an item would compute
    // the same taps; instead lane k computes those of sample k of a round of 32 and the round then reads them back as
    // broadcasts (the tap arithmetic was 3/4 of this kernel's instructions).
    __shared__ uint32_t s_taps[kBlock / 32][32][3][8];
    const int lane = lane_id();
    // lane of an item = (texel of the quad, channel): texel = bits 2-3, channel = bits 0-1 and bit 4 -- the four texel lanes of a channel
    // are then 4 apart inside one DPP row of 16 and their sum is two rotate-and-add instructions (no LDS round trip)
    const int half = lane >> 5, k32 = lane & 31, tex = (lane >> 2) & 3, ch = (lane & 3) | ((lane >> 2) & 4);
    const uint32_t n_grp = (want & 1 ? 1u : 0u) + (want & 2 ? 1u : 0u);
    const unsigned long long item = ((unsigned long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * 2ull + (unsigned)half;
    const uint32_t per_chunk = meta.n_scales * n_grp;
    const uint32_t chunk = (uint32_t)(item / per_chunk), rest = (uint32_t)(item - (unsigned long long)chunk * per_chunk);
    const uint32_t s = rest / n_grp;
    const int grp = n_grp == 2 ? (int)(rest - s * n_grp) : ((want & 1) ? 0 : 1);
    const unsigned long long first = (unsigned long long)chunk * run;
    const bool active = first < M;  // inactive halves still execute the shuffles below
    const uint32_t m0 = active ? (uint32_t)first : 0u, m1 = active ? (uint32_t)(first + run < M ? first + run : M) : 0u;
    const uint32_t ss = active ? s : 0u;
    const float* gbase = grp == 0 ? g_static : g_dynamic;
    const uint32_t stride = meta.n_scales * kC;
    const int pairs[3] = {grp == 0 ? 0 : 2, grp == 0 ? 1 : 4, grp == 0 ? 3 : 5};
    uint32_t (*taps)[3][8] = s_taps[(threadIdx.x >> 5)];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    uint32_t cur[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // i00 of the quad being summed (identifies the quad)
    uint32_t dst[3] = {0u, 0u, 0u};
    const uint32_t n_rows = m1 - m0;
    for (uint32_t r0 = 0; r0 < run; r0 += 32) {  // uniform trip count over the wave
        {   // lane k32: taps of row r0 + k32
            const uint32_t r = r0 + (uint32_t)k32;
            const uint32_t m = r < n_rows ? m0 + r : (M - 1);
            const float4 p4 = reinterpret_cast<const float4*>(xt)[m];
            const float p[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int a = kPa[pairs[j]], b = kPb[pairs[j]];
                const Tap t = make_tap(p[a], p[b], meta.res[ss][a], meta.res[ss][b]);
                uint32_t* o = taps[k32][j];
                o[0] = __builtin_bit_cast(uint32_t, t.nw); o[1] = __builtin_bit_cast(uint32_t, t.ne);
                o[2] = __builtin_bit_cast(uint32_t, t.sw); o[3] = __builtin_bit_cast(uint32_t, t.se);
                o[4] = t.i00; o[5] = t.i01; o[6] = t.i10; o[7] = t.i11;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t k = 0; k < 32u; ++k) {
            const uint32_t r = r0 + k;
            const bool row_ok = r < n_rows;
            const uint32_t m = row_ok ? m0 + r : (M - 1);
            const float g = row_ok ? gbase[(size_t)m * stride + ss * kC + ch] : 0.0f;
            float v[3], w[3];
            uint32_t idx[3], quad[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                w[j] = __builtin_bit_cast(float, taps[k][j][tex]);
                idx[j] = taps[k][j][4 + tex];
                quad[j] = taps[k][j][4];
                float part = planes[meta.off[ss][pairs[j]] + (size_t)idx[j] * kC + ch] * w[j];
                part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x124 /* row_ror:4 */, 0xF, 0xF, false));
                part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
                v[j] = part;  // interpolated value of plane j, channel ch (in all four texel lanes, up to the order of the additions)
            }
            if (!row_ok) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float gv = g * (v[(j + 1) % 3] * v[(j + 2) % 3]);
                if (quad[j] != cur[j]) {
                    if (acc[j] != 0.0f) atomicAdd(g_planes + dst[j], acc[j]);
                    acc[j] = 0.0f;
                    cur[j] = quad[j];
                    dst[j] = meta.off[ss][pairs[j]] + idx[j] * kC + ch;
                }
                acc[j] += gv * w[j];
            }
        }
        __builtin_amdgcn_wave_barrier();  // the taps are overwritten by the next round
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (acc[j] != 0.0f) atomicAdd(g_planes + dst[j], acc[j]);
}

// ------------------------------------------------------------------------------------------------
// Backward of SEVERAL evaluations of one position set in one launch (nvsf_planes_multi_bwd; the forward is k_planes_fwd_runs with the same
// PlaneEvals): what a density query of the space-time field evaluates under autograd -- the static planes and the time planes at (x, t),
// and the time planes at the flow-warped positions of the two neighbour frames (network_dynamic.py:220-271) -- used to be three
// k_planes_bwd_runs launches (+ two coordinate-gradient launches) per ray batch.  Here the items of ALL evaluations are one launch:
// item = (chunk of `run` rows, evaluation, scale), each walking its rows exactly as a k_planes_bwd_runs item does (same taps, same
// products, same run sums: the same addends into the texels, in another order across items).  A first form that let one item walk
// its group's evaluations in turn with ONE set of quad sums -- merging the neighbours' addends with the base evaluation's -- was
// 25 % slower (5.4 against 3 x 1.44 ms per 3.1 M rows): three times longer items, a third of the parallelism.
struct PlaneGradEvals {
    const float* x;
    uint32_t x_stride;
    int n;
    int grp[kMaxEval];
    const float* off[kMaxEval];
    uint32_t off_stride[kMaxEval], off_col[kMaxEval];
    float t[kMaxEval];
    const float* g[kMaxEval];      // gradient of the evaluation's features, rows of n_scales * 8 floats g_stride apart (nullptr: contributes nothing)
    uint32_t g_stride[kMaxEval];   // ... a slice of a wider matrix is read in place (the density tail's input gradient)
    float g_scale[kMaxEval];       // ... times this factor: the evaluations of a BLEND 0.5 d + 0.25 (d1 + d2) share one gradient (x 0.5, 0.25, 0.25)
    float* g_off[kMaxEval];        // optional: receives d L / d offset (3 floats per row at column g_off_col, row stride g_off_stride)
    uint32_t g_off_stride[kMaxEval], g_off_col[kMaxEval];
};

__device__ __forceinline__ float4 eval_position(const PlaneGradEvals& ev, int e, uint32_t m) {
    const float* px = ev.x + (size_t)m * ev.x_stride;
    float4 p = make_float4(px[0], px[1], px[2], ev.t[e]);
    if (ev.off[e]) {
        const float* po = ev.off[e] + (size_t)m * ev.off_stride[e] + ev.off_col[e];
        p.x = p.x + po[0]; p.y = p.y + po[1]; p.z = p.z + po[2];  // fp32 adds, as torch.add forms x + flow
    }
    return p;
}

__global__ __launch_bounds__(kBlock) void k_planes_multi_bwd_runs(PlaneGradEvals ev, uint32_t M, const float* __restrict__ planes, PlaneMeta meta,
                                                                  int live, float* __restrict__ g_planes, uint32_t run) {
    __shared__ uint32_t s_taps[kBlock / 32][32][3][8];
    const int lane = lane_id();
    const int half = lane >> 5, k32 = lane & 31, tex = (lane >> 2) & 3, ch = (lane & 3) | ((lane >> 2) & 4);
    // item = (chunk, live evaluation, scale), the scale fastest: the two halves of a wave (consecutive items, n_scales even) share the chunk
    // AND the evaluation, whose constants are therefore wave-uniform (selected below by compares: a run-time index into the by-value
    // struct would send it to scratch).  `live`: bit e set = evaluation e has a gradient.
    const uint32_t n_live = (uint32_t)__builtin_popcount((unsigned)live);
    const unsigned long long item = ((unsigned long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * 2ull + (unsigned)half;
    const uint32_t per_chunk = meta.n_scales * n_live;
    const uint32_t chunk = (uint32_t)(item / per_chunk), rest = (uint32_t)(item - (unsigned long long)chunk * per_chunk);
    const uint32_t slot = rest / meta.n_scales, s = rest - slot * meta.n_scales;
    int grp = 0;
    const float* gbase = nullptr;
    const float* off = nullptr;
    uint32_t off_stride = 0, off_col = 0, g_stride = 0;
    float t_e = 0.0f, g_scale = 1.0f;
    {
        uint32_t seen = 0;
#pragma unroll
        for (int e = 0; e < kMaxEval; ++e) {
            if (!((live >> e) & 1)) continue;
            if (seen == slot) {
                grp = ev.grp[e]; gbase = ev.g[e]; off = ev.off[e]; off_stride = ev.off_stride[e]; off_col = ev.off_col[e]; t_e = ev.t[e];
                g_stride = ev.g_stride[e]; g_scale = ev.g_scale[e];
            }
            ++seen;
        }
    }
    const unsigned long long first = (unsigned long long)chunk * run;
    const bool active = first < M && gbase != nullptr;
    const uint32_t m0 = active ? (uint32_t)first : 0u, m1 = active ? (uint32_t)(first + run < M ? first + run : M) : 0u;
    const uint32_t ss = active ? s : 0u;
    const uint32_t stride = g_stride;
    const int pairs[3] = {grp == 0 ? 0 : 2, grp == 0 ? 1 : 4, grp == 0 ? 3 : 5};
    uint32_t (*taps)[3][8] = s_taps[(threadIdx.x >> 5)];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    uint32_t cur[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t dst[3] = {0u, 0u, 0u};
    const uint32_t n_rows = m1 - m0;
    for (uint32_t r0 = 0; r0 < run; r0 += 32) {  // uniform trip count over the wave
        {   // lane k32: taps of row r0 + k32 at the evaluation's own position (x + offset, t_e)
            const uint32_t r = r0 + (uint32_t)k32;
            const uint32_t m = r < n_rows ? m0 + r : (M - 1);
            const float* px = ev.x + (size_t)m * ev.x_stride;
            float p[4] = {px[0], px[1], px[2], t_e};
            if (off) {
                const float* po = off + (size_t)m * off_stride + off_col;
                p[0] = p[0] + po[0]; p[1] = p[1] + po[1]; p[2] = p[2] + po[2];  // fp32 adds, as torch.add forms x + flow
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int a = kPa[pairs[j]], b = kPb[pairs[j]];
                const Tap t = make_tap(p[a], p[b], meta.res[ss][a], meta.res[ss][b]);
                uint32_t* o = taps[k32][j];
                o[0] = __builtin_bit_cast(uint32_t, t.nw); o[1] = __builtin_bit_cast(uint32_t, t.ne);
                o[2] = __builtin_bit_cast(uint32_t, t.sw); o[3] = __builtin_bit_cast(uint32_t, t.se);
                o[4] = t.i00; o[5] = t.i01; o[6] = t.i10; o[7] = t.i11;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t k = 0; k < 32u; ++k) {
            const uint32_t r = r0 + k;
            const bool row_ok = r < n_rows;
            const uint32_t m = row_ok ? m0 + r : (M - 1);
            const float g = row_ok ? gbase[(size_t)m * stride + ss * kC + ch] * g_scale : 0.0f;  // (x 1 is exact: the unblended form is unchanged)
            float v[3], w[3];
            uint32_t idx[3], quad[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                w[j] = __builtin_bit_cast(float, taps[k][j][tex]);
                idx[j] = taps[k][j][4 + tex];
                quad[j] = taps[k][j][4];
                float part = planes[meta.off[ss][pairs[j]] + (size_t)idx[j] * kC + ch] * w[j];
                part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x124 /* row_ror:4 */, 0xF, 0xF, false));
                part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
                v[j] = part;
            }
            if (!row_ok) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float gv = g * (v[(j + 1) % 3] * v[(j + 2) % 3]);
                if (quad[j] != cur[j]) {
                    if (acc[j] != 0.0f) atomicAdd(g_planes + dst[j], acc[j]);
                    acc[j] = 0.0f;
                    cur[j] = quad[j];
                    dst[j] = meta.off[ss][pairs[j]] + idx[j] * kC + ch;
                }
                acc[j] += gv * w[j];
            }
        }
        __builtin_amdgcn_wave_barrier();  // the taps are overwritten by the next round
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (acc[j] != 0.0f) atomicAdd(g_planes + dst[j], acc[j]);
}

// ------------------------------------------------------------------------------------------------
// The TIME-plane evaluations of nvsf_planes_multi_bwd through on-chip images (the production form; k_planes_multi_bwd_runs keeps the
// spatial planes and is the test reference for these).
//
// Every evaluation of the multi entry has ONE time t_e for all its rows, so the three time planes (xt, yt, zt) of a scale are only ever
// touched in the two rows Y0, Y1 around t_e, with the SAME two time weights wy0, wy1 for every row of the launch.  The run-merging
// kernel nevertheless treats them as 2-D gathers and sends every run sum of every item of every evaluation to those 2 x W texels as
// a memory-side fp32 atomic: at 4096 x 768 rows a texel receives 10^5 - 10^6 addends, (i) at the slowest atomic rate there is (many
// adders on a few lines): 5.2 ms per launch, the largest kernel of the config-5 training step, and (ii) in an order that changes from
// run to run: 1e-4 of the largest entry between two runs of the SAME step (GPUTEST_r05), 6e-4 against an fp64 sum of the step's own
// addends (tests/test_config5_train_full_size_gpu.py).
//
// Here the problem is made 1-D.  A workgroup owns (slice of the chunks, evaluation, scale) and first folds the time axis away:
//     A_j[X][c] = wy0 P_j[Y0][X][c] + wy1 P_j[Y1][X][c]        (LDS, fp32: the interpolated value is then  wx0 A[X0] + wx1 A[X1]),
// walks its rows with 16 lanes per item (x side, channel): per row and plane one LDS read of A, one weight, one DPP add -- no global
// gather, a quarter of the run-merging kernel's instructions per row -- forms g (v v) wx run sums in registers exactly as that
// kernel does, and adds a run sum to an fp64 image G_j[X][c] in LDS (ds_add_f64).  A run sum is an fp32 number, its conversion is
// exact, and an fp64 sum of <= 2^17 of them rounds at 2^-53 of the running sum: whatever order the waves arrive in, the image agrees
// to ~1e-13 of its entries, far below the one fp32 rounding it gets on the way out -- as good as order-independent, with no scale to
// choose.  (The first form of this kernel used 64-bit fixed point with a scale from a bound of the slice's addends: the pass over the
// gradient rows that bound needs cost 0.55 of the 2.2 ms, measured by leaving it out.)  The final pass adds wy0 G and wy1 G to rows Y0,
// Y1 of the global gradient with contiguous fp32 atomics: one addend per (slice, evaluation) and texel, ~10^3 instead of 10^5 - 10^6.
// Arithmetic against the run-merging kernel: the value wx0 (wy0 P00 + wy1 P10) + wx1 (...) instead of the four-weight blend of
// make_tap, the addend wy (sum g v v wx) instead of sum g v v (wx wy): the same real numbers, roundings at 1e-7 relative.
// Non-finite gradients or texels (GradScaler overflow steps) travel through the sums as they do through the memory-side atomics:
// the texels the offending row touches end up non-finite and found_inf skips the step.
constexpr int kTimeBlock = 1024;            // 16 waves = 64 items of 16 lanes sharing one image: the image (72 KB at the finest scale) allows one or two workgroups per CU whatever their size, and the walk needs the waves
constexpr int kTimeItems = kTimeBlock / 16;
constexpr int kTimeRows = 8;                // rows of an item per round (taps and gradient rows staged in LDS)
constexpr int kTimeUnroll = 4;

struct TimeTap {
    uint32_t Y0, Y1;
    float wy0, wy1;   // (y1 - iy), (iy - y0) of make_tap
};
__device__ __forceinline__ TimeTap make_time_tap(float t, uint32_t H) {
    float iy = ((t * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(H - 1);
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float y0 = floorf(iy), y1 = y0 + 1.0f;
    TimeTap r;
    r.wy0 = y1 - iy; r.wy1 = iy - y0;
    r.Y0 = (uint32_t)y0; r.Y1 = r.Y0 + 1 < H ? r.Y0 + 1 : H - 1;
    return r;
}

__global__ __launch_bounds__(kTimeBlock) void k_planes_multi_bwd_time_lds(PlaneGradEvals ev, uint32_t M, const float* __restrict__ planes,
                                                                          PlaneMeta meta, int live, float* __restrict__ g_planes, uint32_t run,
                                                                          uint32_t chunks_per_slice, uint32_t s) {
    extern __shared__ double s_img[];                     // [n_img] fp64 sums, then [n_img] floats: the folded planes A
    __shared__ float s_tap_ix[kTimeItems][kTimeRows][3];  // per item: make_tap's ix, x0 of a round of rows
    __shared__ uint32_t s_tap_x0[kTimeItems][kTimeRows][3];
    __shared__ float s_g[kTimeItems][kTimeRows][kC];      // ... and the rows' gradient values (scaled)
    const int lane = lane_id();
    const int l16 = lane & 15, xs = (lane >> 3) & 1, ch = lane & 7;
    const uint32_t slot = blockIdx.y;
    const float* gbase = nullptr;
    const float* off = nullptr;
    uint32_t off_stride = 0, off_col = 0, stride = 0;
    float t_e = 0.0f, g_scale = 1.0f;
    {
        uint32_t seen = 0;
#pragma unroll
        for (int e = 0; e < kMaxEval; ++e) {
            if (!((live >> e) & 1)) continue;
            if (seen == slot) {
                gbase = ev.g[e]; off = ev.off[e]; off_stride = ev.off_stride[e]; off_col = ev.off_col[e]; t_e = ev.t[e];
                stride = ev.g_stride[e]; g_scale = ev.g_scale[e];
            }
            ++seen;
        }
    }
    const uint32_t n_chunks = (uint32_t)(((unsigned long long)M + run - 1) / run);
    const uint32_t c_lo = blockIdx.x * chunks_per_slice, c_hi = c_lo + chunks_per_slice < n_chunks ? c_lo + chunks_per_slice : n_chunks;
    if (gbase == nullptr || c_lo >= c_hi) return;  // uniform over the workgroup
    const uint32_t W[3] = {meta.res[s][0], meta.res[s][1], meta.res[s][2]}, H = meta.res[s][3];
    const uint32_t poff[3] = {meta.off[s][2], meta.off[s][4], meta.off[s][5]};  // pairs (0,3), (1,3), (2,3)
    const uint32_t ioff[3] = {0u, W[0] * kC, (W[0] + W[1]) * kC};
    const uint32_t n_img = (W[0] + W[1] + W[2]) * kC;
    float* const s_A = reinterpret_cast<float*>(s_img + n_img);
    const TimeTap ty = make_time_tap(t_e, H);

    // ---- the folded planes ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t row_len = W[j] * kC;
        for (uint32_t i = threadIdx.x; i < row_len; i += kTimeBlock)
            s_A[ioff[j] + i] = planes[poff[j] + ty.Y0 * row_len + i] * ty.wy0 + planes[poff[j] + ty.Y1 * row_len + i] * ty.wy1;
    }
    for (uint32_t i = threadIdx.x; i < n_img; i += kTimeBlock) s_img[i] = 0.0;
    __syncthreads();

    // ---- the walk ------------------------------------------------------------------------------------------------------------------
    const uint32_t it = threadIdx.x >> 4;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    uint32_t cur[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, dst[3] = {0u, 0u, 0u};
#define NVSF_TIME_FLUSH(j)                                                                           \
    {                                                                                                \
        if (acc[j] != 0.0f) atomicAdd(&s_img[ioff[j] + dst[j]], (double)acc[j]);  /* ds_add_f64 */  \
        acc[j] = 0.0f;                                                                               \
    }
    // The rows' operands come from global memory (positions, flow offsets, gradient rows): a round trip of several microseconds against
    // ~1 us of work per round of kTimeRows rows, with one 1024-thread workgroup (4 waves per SIMD) per CU at the finest scale to hide it.  So the loads of round
    // q + 1 are ISSUED (into registers, nothing consumed) before round q is worked on, and staged into LDS when their turn comes.
    struct RoundOperands {
        float p[3], o[3], g[kTimeRows * kC / 16];
    };
    auto rows_of = [&](uint32_t c, uint32_t& m0, uint32_t& n_rows) {
        const bool active = c < c_hi;
        m0 = active ? c * run : 0u;
        const uint32_t m1 = active ? (m0 + run < M ? m0 + run : M) : 0u;
        n_rows = m1 - m0;
    };
    auto issue = [&](uint32_t c, uint32_t r0, RoundOperands& op) __attribute__((always_inline)) {
        uint32_t m0, n_rows;
        rows_of(c, m0, n_rows);
        const uint32_t r = r0 + (uint32_t)(l16 & (kTimeRows - 1));
        const uint32_t m = r < n_rows ? m0 + r : (M - 1);
        const float* px = ev.x + (size_t)m * ev.x_stride;
        op.p[0] = px[0]; op.p[1] = px[1]; op.p[2] = px[2];
        if (off) {
            const float* po = off + (size_t)m * off_stride + off_col;
            op.o[0] = po[0]; op.o[1] = po[1]; op.o[2] = po[2];
        }
#pragma unroll
        for (int i = 0; i < kTimeRows * kC / 16; ++i) {
            const uint32_t idx = (uint32_t)i * 16u + (uint32_t)l16, rr = idx >> 3, cc = idx & 7u;
            const uint32_t mr = r0 + rr < n_rows ? m0 + r0 + rr : (M - 1);
            op.g[i] = gbase[(size_t)mr * stride + s * kC + cc];
        }
    };
    RoundOperands nxt;
    issue(c_lo + it, 0u, nxt);
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += kTimeItems) {  // uniform trip count over the workgroup
        const uint32_t c = c0 + it;
        uint32_t m0, n_rows;
        rows_of(c, m0, n_rows);
        for (uint32_t r0 = 0; r0 < run; r0 += kTimeRows) {
            {   // lanes 0..7 of the item: the x taps of row r0 + lane at the evaluation's own position; all 16: the rows' gradient values
                float p[3] = {nxt.p[0], nxt.p[1], nxt.p[2]};
                if (off) { p[0] = p[0] + nxt.o[0]; p[1] = p[1] + nxt.o[1]; p[2] = p[2] + nxt.o[2]; }  // fp32 adds, as torch.add forms x + flow
                if (l16 < kTimeRows) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const Tap t = make_tap(p[j], t_e, W[j], H);
                        s_tap_ix[it][l16][j] = t.ix_f;
                        s_tap_x0[it][l16][j] = (uint32_t)t.x0;
                    }
                }
#pragma unroll
                for (int i = 0; i < kTimeRows * kC / 16; ++i) {
                    const uint32_t idx = (uint32_t)i * 16u + (uint32_t)l16, rr = idx >> 3, cc = idx & 7u;
                    s_g[it][rr][cc] = r0 + rr < n_rows ? nxt.g[i] * g_scale : 0.0f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (r0 + kTimeRows < run) issue(c, r0 + kTimeRows, nxt);       // the next round of this chunk ...
            else if (c0 + kTimeItems < c_hi) issue(c + kTimeItems, 0u, nxt);  // ... or the first one of the item's next chunk (uniform)
            // kTimeUnroll rows at a time: their LDS operands are all requested before the first of them is consumed
            for (uint32_t k0 = 0; k0 < (uint32_t)kTimeRows; k0 += kTimeUnroll) {
                float g_u[kTimeUnroll], val[kTimeUnroll][3], w_u[kTimeUnroll][3];
                uint32_t X0[kTimeUnroll][3], Xl[kTimeUnroll][3];
#pragma unroll
                for (int u = 0; u < kTimeUnroll; ++u) {
                    g_u[u] = s_g[it][k0 + u][ch];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float ix = s_tap_ix[it][k0 + u][j];
                        X0[u][j] = s_tap_x0[it][k0 + u][j];
                        const float x0 = (float)X0[u][j], x1 = x0 + 1.0f;
                        w_u[u][j] = xs ? (ix - x0) : (x1 - ix);
                        Xl[u][j] = xs ? (X0[u][j] + 1 < W[j] ? X0[u][j] + 1 : W[j] - 1) : X0[u][j];
                        val[u][j] = s_A[ioff[j] + Xl[u][j] * kC + ch];
                    }
                }
#pragma unroll
                for (int u = 0; u < kTimeUnroll; ++u) {
                    const bool row_ok = r0 + k0 + (uint32_t)u < n_rows;
                    float v[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float part = val[u][j] * w_u[u][j];
                        v[j] = part + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
                    }
                    if (row_ok) {  // uniform over the item's 16 lanes
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float gv = g_u[u] * (v[(j + 1) % 3] * v[(j + 2) % 3]);
                            if (X0[u][j] != cur[j]) {
                                NVSF_TIME_FLUSH(j)
                                cur[j] = X0[u][j];
                                dst[j] = Xl[u][j] * kC + ch;
                            }
                            acc[j] += gv * w_u[u][j];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // the staged rows are overwritten by the next round
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            NVSF_TIME_FLUSH(j)
            cur[j] = 0xFFFFFFFFu;
        }
    }
#undef NVSF_TIME_FLUSH
    __syncthreads();

    // ---- the image into rows Y0, Y1 of the global gradient ----------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t row_len = W[j] * kC;
        float* r0p = g_planes + poff[j] + (size_t)ty.Y0 * row_len;
        float* r1p = g_planes + poff[j] + (size_t)ty.Y1 * row_len;
        for (uint32_t i = threadIdx.x; i < row_len; i += kTimeBlock) {
            const double G = s_img[ioff[j] + i];
            if (G == 0.0) continue;
            atomicAdd(r0p + i, (float)(G * (double)ty.wy0));
            if (ty.wy1 != 0.0f) atomicAdd(r1p + i, (float)(G * (double)ty.wy1));
        }
    }
}

// d L / d (offset) of the evaluations that carry one (the flow towards the neighbour frames): thread = (row, evaluation); the arithmetic
// of k_planes_bwd's coordinate half on the evaluation's own position (x + offset, t_e), time planes or static planes as the group says.
// TIME_ONLY: every evaluation of the launch is a time-plane group (what a training pass asks for: the offsets are the scene flow of the
// neighbour frames, which warps the time planes only).  The second axis of those planes is t, whose gradient nobody receives: the derivative
// along it and its sums are not formed (a third of the kernel's arithmetic; same values in the three spatial components).
template <bool TIME_ONLY>
__global__ __launch_bounds__(kBlock) void k_planes_multi_coord_bwd(PlaneGradEvals ev, uint32_t M, const float* __restrict__ planes, PlaneMeta meta) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    // the evaluation's constants selected by compares on blockIdx.y (no run-time index into the by-value struct)
    int grp = 0;
    const float* g_e = nullptr;
    float* go_e = nullptr;
    uint32_t go_stride = 0, go_col = 0, g_stride = 0;
    float g_scale = 1.0f;
    float4 p4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int e = 0; e < kMaxEval; ++e)
        if ((int)blockIdx.y == e) {
            grp = ev.grp[e]; g_e = ev.g[e]; go_e = ev.g_off[e]; go_stride = ev.g_off_stride[e]; go_col = ev.g_off_col[e];
            g_stride = ev.g_stride[e]; g_scale = ev.g_scale[e];
            if (g_e && go_e) p4 = eval_position(ev, e, m);
        }
    if (!g_e || !go_e) return;
    const float p[4] = {p4.x, p4.y, p4.z, p4.w};
    float gp[4] = {0, 0, 0, 0};
    const uint32_t stride = g_stride;
    const int pairs[3] = {grp == 0 ? 0 : 2, grp == 0 ? 1 : 4, grp == 0 ? 3 : 5};
    for (uint32_t s = 0; s < meta.n_scales; ++s) {
        const float* gout = g_e + (size_t)m * stride + s * kC;
        Tap t[3];
        float v[3][kC], dvx[3][kC], dvy[3][kC];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int a = kPa[pairs[j]], b = kPb[pairs[j]];
            t[j] = make_tap(p[a], p[b], meta.res[s][a], meta.res[s][b]);
            const float* plane = planes + meta.off[s][pairs[j]];
            float tex00[kC], tex01[kC], tex10[kC], tex11[kC];
            load_texel(plane, t[j].i00, tex00);
            load_texel(plane, t[j].i01, tex01);
            load_texel(plane, t[j].i10, tex10);
            load_texel(plane, t[j].i11, tex11);
            const float wx1 = t[j].ix_f - t[j].x0, wx0 = 1.0f - wx1, wy1 = t[j].iy_f - t[j].y0, wy0 = 1.0f - wy1;
#pragma unroll
            for (int k = 0; k < kC; ++k) {
                v[j][k] = ((tex00[k] * t[j].nw + tex01[k] * t[j].ne) + tex10[k] * t[j].sw) + tex11[k] * t[j].se;
                dvx[j][k] = (tex01[k] - tex00[k]) * wy0 + (tex11[k] - tex10[k]) * wy1;
                if constexpr (!TIME_ONLY) dvy[j][k] = (tex10[k] - tex00[k]) * wx0 + (tex11[k] - tex01[k]) * wx1;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = pairs[j], a = kPa[q], b = kPb[q];
            float dix = 0.0f, diy = 0.0f;
#pragma unroll
            for (int k = 0; k < kC; ++k) {
                const float other = v[(j + 1) % 3][k] * v[(j + 2) % 3][k];
                const float gv = (gout[k] * g_scale) * other;
                dix += gv * dvx[j][k];
                if constexpr (!TIME_ONLY) diy += gv * dvy[j][k];
            }
            gp[a] += dix * t[j].gx;
            if constexpr (!TIME_ONLY) gp[b] += diy * t[j].gy;
        }
    }
    float* o = go_e + (size_t)m * go_stride + go_col;
    o[0] = gp[0]; o[1] = gp[1]; o[2] = gp[2];
}

int fill_plane_meta(PlaneMeta& meta, uint32_t n_scales, const uint32_t* h_res) {
    if (n_scales == 0 || n_scales > (uint32_t)kMaxScales || !h_res) return NVSF_ERR_INVALID_ARG;
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    meta.n_scales = n_scales;
    unsigned long long off = 0;
    for (uint32_t s = 0; s < n_scales; ++s) {
        for (int d = 0; d < 4; ++d) {
            meta.res[s][d] = h_res[4 * s + d];
            if (h_res[4 * s + d] < 2) return NVSF_ERR_INVALID_ARG;
        }
        for (int q = 0; q < 6; ++q) {
            meta.off[s][q] = (uint32_t)off;
            off += (unsigned long long)h_res[4 * s + pa[q]] * h_res[4 * s + pb[q]] * kC;
            if (off >= (1ull << 31)) return NVSF_ERR_INVALID_ARG;
        }
    }
    return NVSF_OK;
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_planes_fwd(const float* xt, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C, const uint32_t* h_res,
                             int want, float* out_static, float* out_dynamic, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(xt && planes_cl && (want & 3) && (!(want & 1) || out_static) && (!(want & 2) || out_dynamic));
    REQUIRE((reinterpret_cast<uintptr_t>(xt) & 15u) == 0 && (reinterpret_cast<uintptr_t>(planes_cl) & 15u) == 0);
    if (C != (uint32_t)kC) return NVSF_ERR_UNSUPPORTED;
    PlaneMeta meta;
    const int st = fill_plane_meta(meta, n_scales, h_res);
    if (st != NVSF_OK) return st;
    const bool runs_ok = n_scales == 4;
    if (nvsf_variant(kVarPlanesFwd) != 0 || !runs_ok) {  // 1 (tests): one thread per (sample, scale) -- the first formulation, the reference form
        hipLaunchKernelGGL(k_planes_fwd, dim3(cdiv(M, kBlock), n_scales), dim3(kBlock), 0, stream, xt, M, planes_cl, meta, want, out_static,
                           out_dynamic);
        return nvsf_launch_status();
    }
    PlaneEvals ev = {};
    ev.x = xt; ev.x_stride = 4; ev.n = 0;
    for (int grp = 0; grp < 2; ++grp) {
        if (!(want & (1 << grp))) continue;
        ev.grp[ev.n] = grp; ev.off[ev.n] = nullptr; ev.t_from_x[ev.n] = 1; ev.out[ev.n] = grp == 0 ? out_static : out_dynamic;
        ++ev.n;
    }
    const unsigned long long threads = (unsigned long long)cdiv(M, kRun) * ev.n * 2u * n_scales;
    hipLaunchKernelGGL(k_planes_fwd_runs, dim3((uint32_t)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ev, M, planes_cl, meta);
    return nvsf_launch_status();
}

NVSF_API int nvsf_planes_multi_fwd(const float* x, uint32_t x_stride, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C,
                                   const uint32_t* h_res, uint32_t n_evals, const int* h_group, const float* const* h_offsets,
                                   const uint32_t* h_offset_stride, const uint32_t* h_offset_col, const float* h_time,
                                   float* const* h_out, int blend, hipStream_t stream) {
    if (M == 0 || n_evals == 0) return NVSF_OK;
    REQUIRE(x && planes_cl && h_group && h_offsets && h_offset_stride && h_offset_col && h_time && h_out && x_stride >= 3);
    REQUIRE(n_evals <= (uint32_t)kMaxEval && (reinterpret_cast<uintptr_t>(planes_cl) & 15u) == 0);
    if (C != (uint32_t)kC || n_scales != 4) return NVSF_ERR_UNSUPPORTED;
    PlaneMeta meta;
    const int st = fill_plane_meta(meta, n_scales, h_res);
    if (st != NVSF_OK) return st;
    PlaneEvals ev = {};
    ev.x = x; ev.x_stride = x_stride; ev.n = (int)n_evals;
    ev.blend = blend == 2 ? 2 : (blend ? 1 : 0);
    if (blend) REQUIRE(n_evals == 4 && h_group[0] == 0 && h_group[1] == 1 && h_group[2] == 1 && h_group[3] == 1);
    for (uint32_t e = 0; e < n_evals; ++e) {
        REQUIRE((h_group[e] == 0 || h_group[e] == 1) && (h_out[e] || (blend && e >= 2)) && (reinterpret_cast<uintptr_t>(h_out[e]) & 15u) == 0);
        REQUIRE(!h_offsets[e] || h_offset_stride[e] >= h_offset_col[e] + 3);
        ev.grp[e] = h_group[e]; ev.off[e] = h_offsets[e]; ev.off_stride[e] = h_offset_stride[e]; ev.off_col[e] = h_offset_col[e];
        ev.t[e] = h_time[e]; ev.t_from_x[e] = 0; ev.out[e] = h_out[e];
    }
    const unsigned long long threads = (unsigned long long)cdiv(M, kRun) * ev.n * 2u * n_scales;
    hipLaunchKernelGGL(k_planes_fwd_runs, dim3((uint32_t)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ev, M, planes_cl, meta);
    return nvsf_launch_status();
}

NVSF_API int nvsf_planes_bwd(const float* xt, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C, const uint32_t* h_res,
                             int want, const float* grad_static, const float* grad_dynamic, float* grad_planes_cl, float* grad_xt,
                             hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(xt && planes_cl && (want & 3) && (!(want & 1) || grad_static) && (!(want & 2) || grad_dynamic));
    REQUIRE(grad_planes_cl || grad_xt);
    if (C != (uint32_t)kC) return NVSF_ERR_UNSUPPORTED;
    PlaneMeta meta;
    const int st = fill_plane_meta(meta, n_scales, h_res);
    if (st != NVSF_OK) return st;
    if (nvsf_variant(kVarPlanesBwd) == 1) {  // 1 (tests): one atomic per (sample, texel, channel), the first formulation
        hipLaunchKernelGGL(k_planes_bwd, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, stream, xt, M, planes_cl, meta, want, grad_static,
                           grad_dynamic, grad_planes_cl, grad_xt);
        return nvsf_launch_status();
    }
    if (grad_xt)  // coordinate gradients: one thread per sample, no atomics
        hipLaunchKernelGGL(k_planes_bwd, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, stream, xt, M, planes_cl, meta, want, grad_static,
                           grad_dynamic, static_cast<float*>(nullptr), grad_xt);
    if (grad_planes_cl) {
        const uint32_t run = 128u;  // rows per item
        const uint32_t n_grp = ((want & 1) ? 1u : 0u) + ((want & 2) ? 1u : 0u);
        const unsigned long long items = (unsigned long long)cdiv(M, run) * n_scales * n_grp;
        const unsigned long long waves = (items + 1) / 2;
        hipLaunchKernelGGL(k_planes_bwd_runs, dim3((uint32_t)((waves + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, stream, xt, M,
                           planes_cl, meta, want, grad_static, grad_dynamic, grad_planes_cl, run);
    }
    return nvsf_launch_status();
}

NVSF_API int nvsf_planes_multi_bwd(const float* x, uint32_t x_stride, uint32_t M, const float* planes_cl, uint32_t n_scales, uint32_t C,
                                   const uint32_t* h_res, uint32_t n_evals, const int* h_group, const float* const* h_offsets,
                                   const uint32_t* h_offset_stride, const uint32_t* h_offset_col, const float* h_time,
                                   const float* const* h_grad_out, const uint32_t* h_grad_stride, const float* h_grad_scale,
                                   float* grad_planes_cl, float* const* h_grad_offsets, const uint32_t* h_grad_offset_stride,
                                   const uint32_t* h_grad_offset_col, hipStream_t stream) {
    if (M == 0 || n_evals == 0) return NVSF_OK;
    REQUIRE(x && planes_cl && h_group && h_offsets && h_offset_stride && h_offset_col && h_time && h_grad_out && x_stride >= 3);
    REQUIRE(n_evals <= (uint32_t)kMaxEval && (reinterpret_cast<uintptr_t>(planes_cl) & 15u) == 0);
    if (C != (uint32_t)kC || n_scales != 4) return NVSF_ERR_UNSUPPORTED;
    PlaneMeta meta;
    const int st = fill_plane_meta(meta, n_scales, h_res);
    if (st != NVSF_OK) return st;
    PlaneGradEvals ev = {};
    ev.x = x; ev.x_stride = x_stride; ev.n = (int)n_evals;
    int groups = 0, any_coord = 0;
    for (uint32_t e = 0; e < n_evals; ++e) {
        REQUIRE(h_group[e] == 0 || h_group[e] == 1);
        REQUIRE(!h_offsets[e] || h_offset_stride[e] >= h_offset_col[e] + 3);
        ev.grp[e] = h_group[e]; ev.off[e] = h_offsets[e]; ev.off_stride[e] = h_offset_stride[e]; ev.off_col[e] = h_offset_col[e];
        ev.t[e] = h_time[e]; ev.g[e] = h_grad_out[e];
        ev.g_stride[e] = h_grad_stride ? h_grad_stride[e] : n_scales * (uint32_t)kC;
        ev.g_scale[e] = h_grad_scale ? h_grad_scale[e] : 1.0f;
        REQUIRE(!ev.g[e] || ev.g_stride[e] >= n_scales * (uint32_t)kC);
        ev.g_off[e] = h_grad_offsets ? h_grad_offsets[e] : nullptr;
        if (ev.g_off[e]) {
            REQUIRE(h_grad_offset_stride && h_grad_offset_col && h_grad_offset_stride[e] >= h_grad_offset_col[e] + 3);
            ev.g_off_stride[e] = h_grad_offset_stride[e]; ev.g_off_col[e] = h_grad_offset_col[e];
            if (ev.g[e]) any_coord = 1;
        }
        if (ev.g[e]) groups |= 1 << h_group[e];
    }
    if (any_coord) {
        bool time_only = true;
        for (uint32_t e = 0; e < n_evals; ++e)
            if (ev.g[e] && ev.g_off[e] && ev.grp[e] != 1) time_only = false;
        if (time_only) hipLaunchKernelGGL(k_planes_multi_coord_bwd<true>, dim3(cdiv(M, kBlock), n_evals), dim3(kBlock), 0, stream, ev, M, planes_cl, meta);
        else hipLaunchKernelGGL(k_planes_multi_coord_bwd<false>, dim3(cdiv(M, kBlock), n_evals), dim3(kBlock), 0, stream, ev, M, planes_cl, meta);
    }
    if (grad_planes_cl && groups) {
        const uint32_t run = 128u;
        int live = 0, live_time = 0;
        for (uint32_t e = 0; e < n_evals; ++e)
            if (ev.g[e]) {
                live |= 1 << e;
                if (ev.grp[e] == 1) live_time |= 1 << e;
            }
        // production: the time-plane evaluations through the LDS image (k_planes_multi_bwd_time_lds), the spatial planes as run sums
        // into global atomics.  Variant != 0 (tests), small batches (the slices would not fill the chip) or planes too wide for LDS:
        // every evaluation through the run-merging kernel
        uint32_t img_floats = 0;
        for (uint32_t s = 0; s < n_scales; ++s) {
            const uint32_t f = (meta.res[s][0] + meta.res[s][1] + meta.res[s][2]) * (uint32_t)kC;
            img_floats = f > img_floats ? f : img_floats;
        }
        if (live_time && nvsf_variant(kVarPlanesBwd) == 0 && M >= (1u << 16) && (size_t)img_floats * 12u <= 72u * 1024u) {
            // One launch per scale, its images sized for that scale (9 ... 72 KB of fp64 sums + folded planes, + 28 KB of staged rows:
            // one 1024-thread workgroup per CU at the finest scale, two at the coarse ones), several times the resident capacity in workgroups
            // so that the dispatcher evens out what the slices differ by (and what other streams' kernels take): a grid sized to the
            // resident capacity that overshoots it by ONE workgroup runs twice as long (measured: 513 workgroups on 512 slots)
            const uint32_t n_time = (uint32_t)__builtin_popcount((unsigned)live_time);
            const uint32_t n_chunks = cdiv(M, run);
            // a slice = a whole number of the workgroup's item rounds (kTimeItems chunks each), ~2 x CUs slices per evaluation
            uint32_t iters = (n_chunks + (uint32_t)kTimeItems * 2u * (uint32_t)nvsf_cu_count() / 2u) / ((uint32_t)kTimeItems * 2u * (uint32_t)nvsf_cu_count());
            iters = iters < 1u ? 1u : iters;
            const uint32_t chunks_per_slice = iters * (uint32_t)kTimeItems;
            for (uint32_t s = 0; s < n_scales; ++s) {
                const size_t lds = (size_t)(meta.res[s][0] + meta.res[s][1] + meta.res[s][2]) * (size_t)kC * (sizeof(unsigned long long) + sizeof(float));
                hipLaunchKernelGGL(k_planes_multi_bwd_time_lds, dim3(cdiv(n_chunks, chunks_per_slice), n_time), dim3(kTimeBlock), lds, stream, ev, M,
                                   planes_cl, meta, live_time, grad_planes_cl, run, chunks_per_slice, s);
            }
            live &= ~live_time;
        }
        if (!live) return nvsf_launch_status();
        const unsigned long long items = (unsigned long long)cdiv(M, run) * n_scales * (unsigned)__builtin_popcount((unsigned)live);
        const unsigned long long waves = (items + 1) / 2;
        hipLaunchKernelGGL(k_planes_multi_bwd_runs, dim3((uint32_t)((waves + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, stream, ev, M,
                           planes_cl, meta, live, grad_planes_cl, run);
    }
    return nvsf_launch_status();
}
