// Occupancy-grid marcher shared by the raymarching entry points (raymarching.hip) and the fused occupancy render
// (fused_field.hip).  Behavioural contract: raymarching.cu:331-534 / 808-928 of the reference (march_rays_train /
// march_rays): the cascade level of a sample, the Morton-ordered occupancy bit test and the empty-cell skip.
// fp32, every operation individually rounded (-ffp-contract=off), so discrete decisions match the CPU oracle.
#pragma once
#include "common.h"
#include <float.h>
#include <math.h>

namespace {

constexpr float kSqrt3 = 1.7320508075688772f;

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

// `Marcher` carries the per-ray constants; probe() classifies the cell at parameter t and, for an empty cell,
// advances t past it.
struct Marcher {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Hm1, Cf;
    const uint8_t* grid;

    __device__ __forceinline__ void init(const float* o, const float* d, const uint8_t* g, float bound_, float dt_gamma_,
                                         uint32_t max_steps, uint32_t C, uint32_t H) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz;
        bound = bound_; dt_gamma = dt_gamma_; grid = g;
        Hf = (float)H; Hm1 = (float)(H - 1); Cf = (float)C;
        rH = 1.0f / Hf;
        H3 = (float)(H * H * H);
        dt_min = 2.0f * kSqrt3 / (float)max_steps;
        dt_max = 2.0f * kSqrt3 * (float)(1 << (C - 1)) / Hf;
    }
    __device__ __forceinline__ float step_len(float t) const { return clampf(t * dt_gamma, dt_min, dt_max); }

    __device__ __forceinline__ int level_of(float x, float y, float z, float dt) const {
        int e0, e1;
        (void)frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e0);
        (void)frexpf(dt * Hf * 0.5f, &e1);
        const int l0 = (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e0));
        const int l1 = (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e1));
        return l0 > l1 ? l0 : l1;
    }
    // The parameters visited along a ray form ONE chain t_{k+1} = t_k + step_len(t_k), whether a step is taken because a
    // sample was emitted or while skipping an empty cell (probe() below: both add step_len(t)); next() is its recurrence.
    __device__ __forceinline__ float next(float t) const {
        const float tn = t + step_len(t);
        return tn == t ? INFINITY : tn;
    }
    // Branch-free classification of chain member t (for lanes that examine many members at once): the sample
    // description, whether its cell is occupied, and the parameter at which the ray leaves that cell (the skip target of
    // probe()).  Same arithmetic as probe().
    __device__ __forceinline__ bool classify(float t, float& x, float& y, float& z, float& dt, float& t_exit) const {
        x = clampf(ox + t * dx, -bound, bound);
        y = clampf(oy + t * dy, -bound, bound);
        z = clampf(oz + t * dz, -bound, bound);
        dt = step_len(t);
        const int level = level_of(x, y, z, dt);
        const float mb = fminf(ldexpf(1.0f, level), bound);
        const float rmb = 1.0f / mb;
        const int nx = (int)clampf(0.5f * (x * rmb + 1.0f) * Hf, 0.0f, Hm1);
        const int ny = (int)clampf(0.5f * (y * rmb + 1.0f) * Hf, 0.0f, Hm1);
        const int nz = (int)clampf(0.5f * (z * rmb + 1.0f) * Hf, 0.0f, Hm1);
        const uint32_t cell = (uint32_t)((float)level * H3 + (float)morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const bool occ = (grid[cell >> 3] & (1u << (cell & 7u))) != 0;
        const float tx = ((((float)nx + 0.5f + 0.5f * copysignf(1.0f, dx)) * rH * 2.0f - 1.0f) * mb - x) * ix;
        const float ty = ((((float)ny + 0.5f + 0.5f * copysignf(1.0f, dy)) * rH * 2.0f - 1.0f) * mb - y) * iy;
        const float tz = ((((float)nz + 0.5f + 0.5f * copysignf(1.0f, dz)) * rH * 2.0f - 1.0f) * mb - z) * iz;
        t_exit = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        return occ;
    }
    // returns true when the cell containing o + t d is occupied; x,y,z,dt describe the sample.
    // Otherwise t is advanced past the empty cell.
    __device__ __forceinline__ bool probe(float& t, float& x, float& y, float& z, float& dt) const {
        x = clampf(ox + t * dx, -bound, bound);
        y = clampf(oy + t * dy, -bound, bound);
        z = clampf(oz + t * dz, -bound, bound);
        dt = step_len(t);
        const int level = level_of(x, y, z, dt);
        const float mb = fminf(ldexpf(1.0f, level), bound);
        const float rmb = 1.0f / mb;
        const int nx = (int)clampf(0.5f * (x * rmb + 1.0f) * Hf, 0.0f, Hm1);
        const int ny = (int)clampf(0.5f * (y * rmb + 1.0f) * Hf, 0.0f, Hm1);
        const int nz = (int)clampf(0.5f * (z * rmb + 1.0f) * Hf, 0.0f, Hm1);
        const uint32_t cell = (uint32_t)((float)level * H3 + (float)morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        if (grid[cell >> 3] & (1u << (cell & 7u))) return true;
        const float tx = ((((float)nx + 0.5f + 0.5f * copysignf(1.0f, dx)) * rH * 2.0f - 1.0f) * mb - x) * ix;
        const float ty = ((((float)ny + 0.5f + 0.5f * copysignf(1.0f, dy)) * rH * 2.0f - 1.0f) * mb - y) * iy;
        const float tz = ((((float)nz + 0.5f + 0.5f * copysignf(1.0f, dz)) * rH * 2.0f - 1.0f) * mb - z) * iz;
        const float t_exit = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do {
            const float t_next = t + step_len(t);
            if (t_next == t) { t = INFINITY; break; }  // step below 1 ulp of t: leave instead of spinning forever
            t = t_next;
        } while (t < t_exit);
        return false;
    }
};

// Wave-cooperative marcher: all 64 lanes of a wave work on ONE ray.  Lane j classifies member j of the current batch of
// 64 consecutive chain members (one dependent occupancy load per 64 members instead of one per probe); next_sample()
// then replays the serial protocol of Marcher::probe from the per-lane results -- an occupied member is a sample and
// the walk moves to the next member, an empty one jumps to the first member at or beyond the cell's exit parameter
// (ballot + find-first), possibly in a later batch.  All results are wave-uniform.
struct ChainWalker {
    float bt, bx, by, bz, bdt, bexit;  // per lane: member parameter, sample position, step, exit parameter of its cell
    bool bocc;
    int j;          // next member of the batch to examine (64 = batch exhausted)
    float pending;  // exit parameter of an empty cell whose skip runs past the end of a batch (-inf: none)
    float t;        // member 0 of the NEXT batch
    __device__ __forceinline__ void init(float t0) {
        t = t0; j = 64; pending = -INFINITY;
        bt = bx = by = bz = bdt = bexit = 0.0f;
        bocc = false;
    }
    // Next sample of the ray with parameter < far: returns false when the ray is finished.
    __device__ __forceinline__ bool next_sample(const Marcher& m, float far, int lane, float& x, float& y, float& z, float& dt,
                                                float& t_sample) {
        while (true) {
            if (j >= 64) {
                float tc = t;
                bt = t;
                for (int k = 1; k < 64; ++k) {
                    tc = m.next(tc);
                    if (lane == k) bt = tc;
                }
                bocc = m.classify(bt, bx, by, bz, bdt, bexit);
                t = m.next(tc);
                j = 0;
                if (pending > -INFINITY) {
                    const unsigned long long reach = __ballot(bt >= pending);
                    if (reach) { j = __builtin_ctzll(reach); pending = -INFINITY; }
                    else j = 64;
                }
                continue;
            }
            const int ju = __builtin_amdgcn_readfirstlane(j);
            const float tj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bt), ju));
            if (!(tj < far)) return false;
            if (__builtin_amdgcn_readlane((int)bocc, ju)) {
                x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bx), ju));
                y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, by), ju));
                z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bz), ju));
                dt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bdt), ju));
                t_sample = tj;
                j = ju + 1;
                return true;
            }
            const float tt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bexit), ju));
            if (!(tt < far)) return false;  // the serial marcher would step to a member >= tt >= far and stop there
            const unsigned long long later = ~0ull << ju << 1;
            const unsigned long long reach = __ballot(bt >= tt) & later;
            if (reach) j = __builtin_ctzll(reach);
            else { j = 64; pending = tt; }
        }
    }
};

}  // namespace
