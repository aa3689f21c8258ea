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
// the same value for lo <= hi as one v_med3_f32 (a NaN x gives lo either way: med3 returns min3 when an input is NaN)
__device__ __forceinline__ float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

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
    float bound, rbound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Hm1, Cf;
    float halfH;               // 0.5 * H (exact)
    float dt_const;            // the step when dt_gamma == 0 (then every member has the same step and dt-level)
    float rdt_const;           // ~1 / dt_const when dt_gamma == 0, else 0 (a guess that is always verified: ChainWalker::walk_parallel)
    int l1_const;
    int Cm1;                   // C - 1
    uint32_t H3i;              // H^3 when the cell index level * H^3 + morton is exact in integers AND in fp32 (all values < 2^24), else 0
    const uint8_t* grid;
    const uint32_t* lut;       // optional LDS table lut[v] = spread3(v), v < H (fill_spread_lut)
    bool lut_on;               // set together with `lut` (use_lut): a literal at every call site, so the choice folds at compile time --
                               // a test of the pointer itself does not (three branches and three serialised LDS reads per batch)

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
        rbound = 1.0f / bound;
        halfH = 0.5f * Hf;
        Cm1 = (int)C - 1;
        // the reference forms the cell index in fp32 (raymarching.cu:404: level * H^3 + morton, converted back); below 2^24 every
        // value involved is an integer fp32 holds exactly, so one integer multiply-add gives the same index
        uint32_t side = 1;
        while (side < H) side <<= 1;  // morton codes of coordinates < H stay below side^3
        H3i = (unsigned long long)(C - 1) * H * H * H + (unsigned long long)side * side * side <= (1ull << 24) ? H * H * H : 0u;
        dt_const = step_len(0.0f);
        rdt_const = dt_gamma == 0.0f ? __builtin_amdgcn_rcpf(dt_const) : 0.0f;
        l1_const = dt_level(dt_const);
        lut = nullptr;
        lut_on = false;
    }
    // What replaying recorded batches needs (positions, step lengths, fill_batch): no inverse directions, no cell constants.
    __device__ __forceinline__ void init_replay(const float* o, const float* d, float bound_, float dt_gamma_, uint32_t max_steps, uint32_t C,
                                                uint32_t H) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        bound = bound_; dt_gamma = dt_gamma_;
        Hf = (float)H;
        dt_min = 2.0f * kSqrt3 / (float)max_steps;
        dt_max = 2.0f * kSqrt3 * (float)(1 << (C - 1)) / Hf;
        dt_const = step_len(0.0f);
        lut = nullptr;
        lut_on = false;
    }
    __device__ __forceinline__ int dt_level(float dt) const {
        int e1;
        (void)frexpf(dt * Hf * 0.5f, &e1);
        return (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e1));
    }
    __device__ __forceinline__ float step_len(float t) const { return clampf(t * dt_gamma, dt_min, dt_max); }

    __device__ __forceinline__ int level_of(float x, float y, float z, float dt) const {
        int e0;
        (void)frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e0);
        const int l0 = (int)fminf(Cf - 1.0f, fmaxf(0.0f, (float)e0));
        const int l1 = dt_gamma == 0.0f ? l1_const : dt_level(dt);  // wave-uniform choice
        return l0 > l1 ? l0 : l1;
    }
    __device__ __forceinline__ void use_lut(const uint32_t* table) { lut = table; lut_on = true; }
    __device__ __forceinline__ uint32_t spread(uint32_t v) const { return lut_on ? lut[v] : spread3(v); }
    // The parameters visited along a ray form ONE chain t_{k+1} = t_k + step_len(t_k), whether a step is taken because a
    // sample was emitted or while skipping an empty cell (probe() below: both add step_len(t)); next() is its recurrence.
    __device__ __forceinline__ float next(float t) const {
        const float tn = t + step_len(t);
        return tn == t ? INFINITY : tn;
    }
    // One batch of consecutive chain members starting at t: lane k receives member k in `bt`; nb (wave-uniform, 1..64)
    // is the number of members in the batch (lanes >= nb hold a harmless copy of member 0) and t_next is member 0 of the
    // following batch.
    // dt_gamma == 0 (the reference's default): the step is the constant c = step_len(.) and the chain is the running
    // fp32 sum t_k = fl(t_{k-1} + c).  Inside one binade of t that sum advances by a fixed number of ulps: with
    // t = m u (u = ulp, 2^23 <= m < 2^24) and c = (q + r) u, 0 <= r < 1, round-to-nearest-even gives m + q (r < 1/2),
    // m + q + 1 (r > 1/2), and for r = 1/2 the even one of the two -- after which m stays even and the step is
    // q + (q & 1).  So member k is m_0 + s_first + (k - 1) s in integer arithmetic, valid as long as the exact sum of
    // the previous step stayed inside the binade (m_{k-1} + q < 2^24); the batch ends there and the next batch starts
    // from a real fp32 addition, which handles the crossing.  Bit-identical to the serial recurrence (checked
    // exhaustively against it on the host: tests/test_oracle_cpu.py::test_constant_step_chain_closed_form).
    __device__ __forceinline__ void fill_batch(float t, int lane, float& bt, int& nb, float& t_next) const {
        if (dt_gamma == 0.0f) {
            // A batch is a run of SEGMENTS, each inside one binade of t and generated from the closed form above; the first
            // member of the next segment is the real fp32 sum fl(t_last + c), exactly what starts the next batch.  Without
            // this a batch would end at every binade crossing (a camera ray from t = 0.01 crosses eight of them: 17.5 batches
            // of 41 members on average instead of 11.7 of 61 at 1024 steps).
            const uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, dt_const));
            const int ec = (int)(cb >> 23);
            const uint32_t mc = (cb & 0x7FFFFFu) | 0x800000u;
            float tc = t;  // wave-uniform: member 0 of the current segment
            int filled = 0;
            bt = t;
            while (ec > 0) {
                const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, tc));
                const int e = (int)(tb >> 23);  // a negative / NaN / inf t gives e >= 255: generic path (or the batch ends here)
                const int d = e - ec;
                if (!(d >= 0 && d <= 24 && e >= 24 && e < 254)) break;
                const uint32_t m0 = (tb & 0x7FFFFFu) | 0x800000u;
                const uint32_t q = mc >> d, rem = mc & ((1u << d) - 1u), half = d > 0 ? 1u << (d - 1) : 0u;
                uint32_t s, s_first;
                if (d > 0 && rem == half) { s = q + (q & 1u); s_first = q + ((m0 + q) & 1u); }
                else { s = q + ((d > 0 && rem > half) ? 1u : 0u); s_first = s; }
                if (s == 0u) break;
                const int k = lane - filled;  // member index inside the segment (negative: a lane of an earlier segment)
                const uint32_t ku = (uint32_t)k;
                // (k - 1) * s: k < 64 and s < 2^24, so the 24-bit multiplier (full rate; the 32-bit one runs at a quarter) is exact
                const uint32_t mk = k == 0 ? m0 : m0 + s_first + __umul24(ku - 1u, s);
                const uint32_t mprev = k <= 1 ? m0 : mk - s;
                const bool valid = k >= 0 && (k == 0 || mprev + q < (1u << 24));
                filled += __builtin_popcountll(__ballot(valid));  // the valid lanes are a run from lane `filled` (m_k increases with k)
                const float scale = __builtin_bit_cast(float, (uint32_t)(e - 23) << 23);
                if (valid) bt = (float)mk * scale;
                const float t_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bt), filled - 1));
                tc = next(t_last);
                if (filled >= 64) break;
            }
            if (filled > 0) {
                nb = filled;
                t_next = tc;
                return;
            }
        }
        float tc = t;
        bt = t;
        for (int k = 1; k < 64; ++k) {
            tc = next(tc);
            if (lane == k) bt = tc;
        }
        nb = 64;
        t_next = next(tc);
    }
    // Branch-free classification of chain member t (for lanes that examine many members at once): the sample
    // description, whether its cell is occupied (classify_cell), and the parameter at which the ray leaves that cell
    // (cell_exit: the skip target of probe(); only needed for empty cells).  Same arithmetic as probe(): 1 / mip_bound is
    // 2^-level (exact) or the ray-constant 1 / bound, the same IEEE quotients probe() computes per sample.
    struct Cell { int nx, ny, nz; float mb; };
    // level_of() with the clamp done on the integer exponent: (int)min(C - 1, max(0, (float)e)) == med3(e, 0, C - 1)
    __device__ __forceinline__ int level_fast(float x, float y, float z, float dt) const {
        int e0, e1;
        (void)frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e0);
        const int l0 = max(0, min(e0, Cm1));
        int l1 = l1_const;
        if (dt_gamma != 0.0f) {  // wave-uniform choice, kept a branch (the empty asm cannot be speculated): ten instructions per member otherwise
            asm volatile("");
            (void)frexpf(dt * Hf * 0.5f, &e1);
            l1 = max(0, min(e1, Cm1));
        }
        return l0 > l1 ? l0 : l1;
    }
    __device__ __forceinline__ bool classify_cell(float t, float& x, float& y, float& z, float& dt, Cell& c) const {
        x = clamp_med3(ox + t * dx, -bound, bound);
        y = clamp_med3(oy + t * dy, -bound, bound);
        z = clamp_med3(oz + t * dz, -bound, bound);
        dt = dt_const;
        if (dt_gamma != 0.0f) {
            asm volatile("");
            dt = step_len(t);
        }
        const int level = level_fast(x, y, z, dt);
        const float p = ldexpf(1.0f, level);
        const bool capped = bound < p;  // mb = fminf(2^level, bound)
        c.mb = capped ? bound : p;
        const float rmb = capped ? rbound : ldexpf(1.0f, -level);
        // 0.5f * (v * rmb + 1.0f) * H of probe(): halving is exact, so (v * rmb + 1.0f) * (0.5f * H) rounds the same real number
        c.nx = (int)clamp_med3((x * rmb + 1.0f) * halfH, 0.0f, Hm1);
        c.ny = (int)clamp_med3((y * rmb + 1.0f) * halfH, 0.0f, Hm1);
        c.nz = (int)clamp_med3((z * rmb + 1.0f) * halfH, 0.0f, Hm1);
        const uint32_t mort = spread((uint32_t)c.nx) | (spread((uint32_t)c.ny) << 1) | (spread((uint32_t)c.nz) << 2);
        const uint32_t cell = H3i ? __umul24((uint32_t)level, H3i) + mort : (uint32_t)((float)level * H3 + (float)mort);  // level * H^3 < 2^24
        return (grid[cell >> 3] & (1u << (cell & 7u))) != 0;
    }
    __device__ __forceinline__ float cell_exit(float t, float x, float y, float z, const Cell& c) const {
        const float tx = ((((float)c.nx + 0.5f + 0.5f * copysignf(1.0f, dx)) * rH * 2.0f - 1.0f) * c.mb - x) * ix;
        const float ty = ((((float)c.ny + 0.5f + 0.5f * copysignf(1.0f, dy)) * rH * 2.0f - 1.0f) * c.mb - y) * iy;
        const float tz = ((((float)c.nz + 0.5f + 0.5f * copysignf(1.0f, dz)) * rH * 2.0f - 1.0f) * c.mb - z) * iz;
        return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    __device__ __forceinline__ bool classify(float t, float& x, float& y, float& z, float& dt, float& t_exit) const {
        Cell c;
        const bool occ = classify_cell(t, x, y, z, dt, c);
        t_exit = cell_exit(t, x, y, z, c);
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

// Fills the workgroup's LDS table of spread3(v), v < H (H <= kSpreadLutMax); the caller synchronises afterwards.
constexpr uint32_t kSpreadLutMax = 1024;
__device__ __forceinline__ void fill_spread_lut(uint32_t* lut, uint32_t H) {
    for (uint32_t v = threadIdx.x; v < H; v += blockDim.x) lut[v] = spread3(v);
}

// Wave-cooperative marcher: all 64 lanes of a wave work on ONE ray.  Lane j classifies member j of the current batch of
// up to 64 consecutive chain members (one dependent occupancy load per batch instead of one per probe); next_samples()
// then decides which members the serial protocol of Marcher::probe would visit -- an occupied member is a sample and
// the walk moves to the next member, an empty one jumps to the first member at or beyond the cell's exit parameter,
// possibly in a later batch -- for the whole batch at once and returns the samples as a lane mask (wave-uniform).
struct ChainWalker {
    float bt, bx, by, bz, bdt, bexit;  // per lane: member parameter, sample position, step, exit parameter of its cell
    bool bocc;
    int j, nb;      // next member of the batch to examine, members in the batch (j >= nb: batch exhausted)
    float pending;  // exit parameter of an empty cell whose skip runs past the end of a batch (-inf: none)
    float t;        // member 0 of the NEXT batch
    float t_batch;  // member 0 of the CURRENT batch (fill_batch(t_batch, ...) regenerates its members)
    bool done;      // next_samples(): the ray is finished
    __device__ __forceinline__ void init(float t0) {
        t = t0; t_batch = t0; j = 64; nb = 64; pending = -INFINITY; done = false;
        bt = bx = by = bz = bdt = bexit = 0.0f;
        bocc = false;
    }
    __device__ __forceinline__ unsigned long long batch_mask() const { return nb >= 64 ? ~0ull : (1ull << nb) - 1ull; }
    __device__ __forceinline__ void refill(const Marcher& m, int lane) {
        const float t0 = t;
        t_batch = t0;
        m.fill_batch(t0, lane, bt, nb, t);
        Marcher::Cell c;
        bocc = m.classify_cell(bt, bx, by, bz, bdt, c);
        if (__ballot(!bocc) & batch_mask()) bexit = m.cell_exit(bt, bx, by, bz, c);  // exits matter for empty cells only
        j = 0;
        if (pending > -INFINITY) {
            const unsigned long long reach = __ballot(bt >= pending) & batch_mask();
            if (reach) { j = __builtin_ctzll(reach); pending = -INFINITY; }
            else j = nb;
        }
    }
    // All samples of the next batch that has any, as a lane mask (lane k set: member k of the batch is a sample; its
    // description sits in bt / bx / by / bz / bdt of that lane; samples are in chain order = lane order).  At most max_n
    // (>= 1) samples.  Returns 0 when the ray is finished.
    //
    // Which members of a batch are visited is decided for all 64 lanes at once.  An empty member j skips to
    // nxt_j = the first later member at or beyond its cell's exit (binary search over the non-decreasing bt), a member
    // at / beyond `far` or an empty one whose exit is beyond `far` ends the ray (nxt = 255), an occupied one moves to
    // j + 1.  With P_k = max nxt_j over the empty members j < k, member k is visited iff P_k <= k -- provided no skipped
    // member would have skipped further than the member that skipped it (nxt_j <= P_j for every covered j; then covered
    // members contribute nothing to P and an induction over k gives visited == uncovered).  Cells are convex, so the
    // proviso only fails through rounding at a cell face; that batch is then walked serially (walk_serial, the protocol of
    // Marcher::probe verbatim).  `serial` forces the serial walk (tests, A/B).
    __device__ __forceinline__ unsigned long long next_samples(const Marcher& m, float far, int lane, uint32_t max_n, bool serial = false) {
        while (true) {
            if (done) return 0ull;
            if (j >= nb) {
                refill(m, lane);
                if (j >= nb) continue;
            }
            unsigned long long S = 0ull;
            if (serial || !walk_parallel(far, lane, S, m.rdt_const)) S = walk_serial(far);
            if (S) {
                if ((uint32_t)__builtin_popcountll(S) > max_n) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(S >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)S, 0u));
                    S = __ballot(((S >> lane) & 1ull) && rank < max_n);
                    done = true;
                }
                return S;
            }
        }
    }
    __device__ __forceinline__ unsigned long long walk_serial(float far) {
        const unsigned long long bm = batch_mask();
        const unsigned long long below = __ballot(bt < far) & bm;
        const unsigned long long occ = __ballot(bocc) & below;
        unsigned long long S = 0ull;
        while (j < nb) {
            const int ju = __builtin_amdgcn_readfirstlane(j);
            if (!((below >> ju) & 1ull)) { done = true; break; }
            if ((occ >> ju) & 1ull) {
                const unsigned long long inv = ~(occ >> ju);
                const int r = inv ? __builtin_ctzll(inv) : 64;
                S |= (r >= 64 ? ~0ull : (1ull << r) - 1ull) << ju;
                j = ju + r;
                continue;
            }
            const float tt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bexit), ju));
            if (!(tt < far)) { done = true; break; }  // the serial marcher would step to a member >= tt >= far and stop there
            const unsigned long long later = (~0ull << ju << 1) & bm;
            const unsigned long long reach = __ballot(bt >= tt) & later;
            if (reach) j = __builtin_ctzll(reach);
            else { j = nb; pending = tt; }
        }
        j = nb;
        return S;
    }
    // returns false (state untouched) when the batch needs the serial walk
    // rdt: 1 / step when the step is the same for every member (dt_gamma == 0), else 0
    __device__ __forceinline__ bool walk_parallel(float far, int lane, unsigned long long& S, float rdt) {
        const int j0 = __builtin_amdgcn_readfirstlane(j);
        const unsigned long long bm = batch_mask();
        const unsigned long long active = bm & (~0ull << j0);
        const bool in = (active >> lane) & 1ull;
        const bool below = bt < far;
        // No empty member below far among the remaining ones: the walk visits member after member until the first one at or beyond
        // far (bt does not decrease along the batch, so the members below far are a run from j0) and stops there.
        const unsigned long long below_m = __ballot(below);
        if (!(active & below_m & ~__ballot(bocc))) {
            S = active & below_m;
            if (active & ~below_m) done = true;
            j = nb;
            return true;
        }
        const bool empty = in && below && !bocc;
        const bool stopper = in && (!below || (!bocc && !(bexit < far)));
        uint32_t nxt = stopper ? 255u : 0u;
        if (__ballot(empty && !stopper)) {
            // cnt = number of batch members below this lane's exit parameter (bt is non-decreasing along the batch)
            int cnt = 0;
            bool found = false;
            if (rdt > 0.0f) {
                // Constant step c: member k sits within rounding of bt + (k - lane) c, so the first member at or beyond the exit is
                // guessed as k* = lane + ceil((bexit - bt) / c) and CHECKED against the members themselves, bt[k* - 1] < bexit <=
                // bt[k*] (two independent cross-lane reads instead of six dependent ones); any lane whose guess fails sends the
                // wave to the search below, so the result is the search's in every case.
                const float e = ceilf((bexit - bt) * rdt);
                int k = lane + (int)fminf(fmaxf(e, -64.0f), 128.0f);
                k = k < 0 ? 0 : (k > nb ? nb : k);
                const float below = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((k > 0 ? k - 1 : 0) << 2, __builtin_bit_cast(int, bt)));
                const float at = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((k < nb ? k : nb - 1) << 2, __builtin_bit_cast(int, bt)));
                const bool ok = (k == 0 || below < bexit) && (k == nb || !(at < bexit));
                cnt = k;
                found = !__ballot(empty && !stopper && !ok);
            }
            if (!found) {
                const float last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bt), nb - 1));
                int lo = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int probe = lo + step - 1;
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(probe << 2, __builtin_bit_cast(int, bt)));
                    if (probe < nb && v < bexit) lo += step;
                }
                cnt = last < bexit ? nb : lo;
            }
            const int jump = cnt > lane + 1 ? cnt : lane + 1;
            if (empty && !stopper) nxt = (uint32_t)jump;
        }
        const uint32_t incl = wave_scan_max_u32(nxt);
        uint32_t P = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) P = 0u;
        const bool covered = P > (uint32_t)lane;
        if (__ballot((empty || stopper) && covered && nxt > P)) return false;
        const bool vis = in && !covered;
        const unsigned long long stop = __ballot(vis && stopper);
        S = __ballot(vis && bocc && below);
        if (stop) {
            S &= (1ull << __builtin_ctzll(stop)) - 1ull;
            done = true;
        } else {
            const unsigned long long visited = __ballot(vis);  // non-zero: member j0 is never covered
            const int v = 63 - __builtin_clzll(visited);
            if (!__builtin_amdgcn_readlane((int)bocc, v)) pending = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bexit), v));
        }
        j = nb;
        return true;
    }
};

}  // namespace
