// Ray-marching operators for gfx950 (MI355X): AABB intersection, background-sphere coords, Morton
// codes, occupancy bit packing, occupancy-grid sample generation (training + inference) and the
// packed-sample compositors.  Behavioural contract: /root/reference/nvsf/nerf/raymarching/src/raymarching.cu
// (kernel line ranges cited per entry point in include/nvsf_hip.h).  Design differences, all MI355X-driven:
//   * wave64 everywhere; 256-thread workgroups;
//   * march_rays_train is three launches (count -> single-workgroup exclusive scan -> write) so the
//     packed sample order is deterministic (ray-index order) instead of atomicAdd arrival order;
//   * the packed compositors give one 64-lane wave to each ray: coalesced sigma/rgb/delta reads and a
//     cross-lane product scan for the transmittance instead of a per-thread serial loop, so that a
//     4096-ray batch fills 4096 waves rather than 64;
//   * packbits builds each output byte from two 16-byte loads per lane (full 128-B lines per wave).
// Arithmetic is fp32 with every operation individually rounded (-ffp-contract=off) so discrete
// decisions (floor, frexp, occupancy bit) agree with the CPU oracle bit for bit.
#include "march_device.h"

namespace {

constexpr float kRPi = 0.3183098861837907f;
constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t compact3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ aabb, uint32_t N, float min_near,
                                                     float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float ix = 1.0f / rays_d[3 * n], iy = 1.0f / rays_d[3 * n + 1], iz = 1.0f / rays_d[3 * n + 2];
    float lo = (aabb[0] - ox) * ix, hi = (aabb[3] - ox) * ix;
    if (lo > hi) { float s = lo; lo = hi; hi = s; }
    float ylo = (aabb[1] - oy) * iy, yhi = (aabb[4] - oy) * iy;
    if (ylo > yhi) { float s = ylo; ylo = yhi; yhi = s; }
    bool miss = (lo > yhi) || (ylo > hi);
    if (!miss) {
        if (ylo > lo) lo = ylo;
        if (yhi < hi) hi = yhi;
        float zlo = (aabb[2] - oz) * iz, zhi = (aabb[5] - oz) * iz;
        if (zlo > zhi) { float s = zlo; zlo = zhi; zhi = s; }
        miss = (lo > zhi) || (zlo > hi);
        if (!miss) {
            if (zlo > lo) lo = zlo;
            if (zhi < hi) hi = zhi;
            if (lo < min_near) lo = min_near;
        }
    }
    nears[n] = miss ? FLT_MAX : lo;
    fars[n] = miss ? FLT_MAX : hi;
}

__global__ __launch_bounds__(kBlock) void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         float radius, uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[2 * n] = 2.0f * theta * kRPi - 1.0f;
    coords[2 * n + 1] = phi * kRPi;
}

__global__ __launch_bounds__(kBlock) void k_morton3D(const int* __restrict__ coords, uint32_t N, int* __restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int)morton_encode((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}

__global__ __launch_bounds__(kBlock) void k_morton3D_invert(const int* __restrict__ indices, uint32_t N, int* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[3 * n] = (int)compact3((uint32_t)(ind >> 0));
    coords[3 * n + 1] = (int)compact3((uint32_t)(ind >> 1));
    coords[3 * n + 2] = (int)compact3((uint32_t)(ind >> 2));
}

// One thread = one output byte = 8 consecutive floats, fetched as two float4 (the wave reads 2 KiB of
// whole cache lines and writes 64 contiguous bytes).
__global__ __launch_bounds__(kBlock) void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh,
                                                     uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[2 * (size_t)n];
    const float4 b = reinterpret_cast<const float4*>(grid)[2 * (size_t)n + 1];
    uint32_t bits = (a.x > thresh ? 1u : 0u) | (a.y > thresh ? 2u : 0u) | (a.z > thresh ? 4u : 0u) |
                    (a.w > thresh ? 8u : 0u) | (b.x > thresh ? 16u : 0u) | (b.y > thresh ? 32u : 0u) |
                    (b.z > thresh ? 64u : 0u) | (b.w > thresh ? 128u : 0u);
    bitfield[n] = (uint8_t)bits;
}

// ------------------------------------------------------------------------------------------------
// Occupancy-grid sample generation (the marcher itself: march_device.h).
// pass 1: per-ray sample count -> rays[n].z
__global__ __launch_bounds__(kBlock) void k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                        const float* __restrict__ nears, const float* __restrict__ fars,
                                                        const float* __restrict__ noises, int* __restrict__ rays) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    Marcher m;
    m.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
    const float far = fars[n];
    float t = nears[n];
    t += m.step_len(t) * noises[n];
    uint32_t count = 0;
    float x, y, z, dt;
    while (t < far && count < max_steps) {
        if (m.probe(t, x, y, z, dt)) { ++count; t += dt; }
    }
    rays[3 * (size_t)n + 2] = (int)count;
}

// The same two passes with a WAVE per ray (ChainWalker, march_device.h): 64 chain members are classified per
// dependent occupancy load and no lane waits for the longest ray of its wave.  Identical samples (same chain, same
// classification arithmetic, same skip protocol) -- tests/test_raymarching_gpu.py pins both forms to the oracle.
__global__ __launch_bounds__(kBlock) void k_march_count_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                             const float* __restrict__ nears, const float* __restrict__ fars,
                                                             const float* __restrict__ noises, int* __restrict__ rays, int serial) {
    __shared__ uint32_t s_lut[kSpreadLutMax];
    fill_spread_lut(s_lut, H);  // H <= 1024 (checked by the launcher)
    __syncthreads();
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    if (n >= N) return;
    const int lane = lane_id();
    Marcher m;
    m.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
    m.use_lut(s_lut);
    const float far = fars[n];
    float t = nears[n];
    t += m.step_len(t) * noises[n];
    ChainWalker w;
    w.init(t);
    uint32_t count = 0;
    while (count < max_steps) {
        const unsigned long long S = w.next_samples(m, far, lane, max_steps - count, serial != 0);
        if (!S) break;
        count += (uint32_t)__builtin_popcountll(S);
    }
    if (lane == 0) rays[3 * (size_t)n + 2] = (int)count;
}

__global__ __launch_bounds__(kBlock) void k_march_write_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                             const float* __restrict__ nears, const float* __restrict__ fars,
                                                             const float* __restrict__ noises, const int* __restrict__ rays,
                                                             float* __restrict__ xyzs, float* __restrict__ dirs,
                                                             float* __restrict__ deltas, int serial) {
    __shared__ uint32_t s_lut[kSpreadLutMax];
    fill_spread_lut(s_lut, H);
    __syncthreads();
    const uint32_t n = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t offset = (uint32_t)rays[3 * (size_t)n + 1], count = (uint32_t)rays[3 * (size_t)n + 2];
    if (count == 0 || offset + count > M) return;
    Marcher m;
    m.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
    m.use_lut(s_lut);
    const float far = fars[n];
    float t = nears[n];
    t += m.step_len(t) * noises[n];
    float last_t = t;
    ChainWalker w;
    w.init(t);
    // the samples of a batch sit in the lanes of the mask S, in chain order: every such lane stores its own sample
    uint32_t step = 0;
    while (step < count) {
        const unsigned long long S = w.next_samples(m, far, lane, count - step, serial != 0);
        if (!S) break;
        const float t_after = w.bt + w.bdt;
        const unsigned long long lower = S & ((1ull << lane) - 1ull);  // samples of this batch before this lane
        const int prev_lane = lower ? 63 - __builtin_clzll(lower) : lane;
        const float prev_after = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(prev_lane << 2, __builtin_bit_cast(int, t_after)));
        if ((S >> lane) & 1ull) {
            const size_t s = (size_t)offset + step + (uint32_t)__builtin_popcountll(lower);
            xyzs[3 * s] = w.bx; xyzs[3 * s + 1] = w.by; xyzs[3 * s + 2] = w.bz;
            dirs[3 * s] = m.dx; dirs[3 * s + 1] = m.dy; dirs[3 * s + 2] = m.dz;
            deltas[2 * s] = w.bdt; deltas[2 * s + 1] = t_after - (lower ? prev_after : last_t);
        }
        last_t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_after), 63 - __builtin_clzll(S)));
        step += (uint32_t)__builtin_popcountll(S);
    }
}

// ------------------------------------------------------------------------------------------------
// march_rays_train in ONE launch that classifies every chain member ONCE (nvsf_march_rays_train_ws).
//
// The reference-shaped three-launch form classifies every chain member twice (count pass, write pass: ~115 VALU instructions per
// 64 members each) because the sample ranges are only known after the scan.  Here a wave counts its ray once and keeps, per batch
// that holds samples, a 16-byte record {first chain parameter, members, 64-bit sample mask} and (for the first kMarchBtCap
// batches) the 64 member parameters in LDS; once the ray's range is known it replays the records -- positions by the marcher's
// own clamp(o + t d) -- and stores the samples.  No second classification, and the stores of one ticket drain while the next
// one is counted (counting is VALU-bound, the stores are not).
// Order of the packed samples: ray-index order, exactly as the three-launch form and the CPU oracle (the reference's order
// depends on the arrival order of its atomics, raymarching.cu:445-446).
//
// Where the ranges come from.  Workgroups 1.. are WORKERS: they draw tickets (4 consecutive rays, one per wave) from an atomic
// counter until none are left; per ticket: count, publish the ticket's sum, and -- one ticket LATER, after counting the next one --
// fetch the ticket's exclusive prefix and store.  Workgroup 0 is the SCANNER: one wave that walks the published sums in ticket
// order (up to 512 per poll) and publishes every ticket's exclusive prefix.  A worker therefore touches two 8-byte flags per
// ticket and never adds up anybody else's sums.  (Two earlier forms of this launch let every workgroup look back over its
// predecessors' sums itself: all workgroups of the chip start together, finish counting together, and each then needs ~2000
// flags through agent-scope loads that miss the L2s -- 0.15 of 0.30 ms was waiting, and counting the next ticket meanwhile did
// not help because the look-back itself was the cost.  Count / scan / replay as three launches with the records in global
// memory: 0.36 ms, the stores no longer overlap the counting.)
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, recipe R2): a ticket is drawn by a RUNNING workgroup, which
// publishes the ticket's sum without waiting for anything, so the scanner (workgroup 0: dispatched first) only ever waits for
// running workgroups, and workers only wait for the scanner.  Each flag is ONE naturally aligned 8-byte {status, value} granule
// written by a single agent-scope atomic store and read by agent-scope atomic loads -- no other data crosses workgroups.  Every
// spin is bounded (kMarchSpinLimit polls); on a timeout the launch still terminates and reports through counter[1] < 0.
constexpr int kMarchRecCap = 24;  // records per ray kept in LDS; a ray with more sample-bearing batches re-marches when writing
// (template parameter BT) batches per ray whose member parameters stay in LDS; the others come from Marcher::fill_batch again
constexpr int kMarchRays = kBlock / kWave;  // rays per ticket: one per wave
constexpr uint32_t kMarchSpinLimit = 1u << 22;
struct MarchRec { float t0; uint32_t nb; unsigned long long S; };

// OFF32: 12 * M < 2^32, so sample byte offsets fit 32 bits and the stores take the scalar base + 32-bit lane offset form (no
// 64-bit multiply-adds per lane).  The per-lane sample index is offset + step + (samples of the batch below the lane); the second
// delta needs t + dt of the previous sample: the neighbouring lane when the batch's samples are a contiguous run (one DPP move),
// the lane below found through the mask otherwise.
struct MarchDirs { float x, y, z; };
template <bool OFF32>
__device__ __forceinline__ void march_store_batch(const MarchDirs& dv, float bt, float bx, float by, float bz, float bdt, unsigned long long S,
                                                  int lane, uint32_t offset, uint32_t& step, float& last_t, float* __restrict__ xyzs,
                                                  float* __restrict__ dirs, float* __restrict__ deltas) {
    const float t_after = bt + bdt;
    const unsigned long long lower = S & ((1ull << lane) - 1ull);  // samples of this batch before this lane
    const int first = __builtin_ctzll(S), top = 63 - __builtin_clzll(S);
    const bool run = (((S >> first) + 1ull) & (S >> first)) == 0ull;  // wave-uniform: the samples are lanes first..top
    float prev_after;
    if (run) {
        prev_after = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t_after), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
    } else {
        const int prev_lane = lower ? 63 - __builtin_clzll(lower) : lane;
        prev_after = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(prev_lane << 2, __builtin_bit_cast(int, t_after)));
    }
    if ((S >> lane) & 1ull) {
        const uint32_t rank = run ? (uint32_t)(lane - first) : (uint32_t)__builtin_popcountll(lower);
        const float d1 = t_after - (rank ? prev_after : last_t);
        if (OFF32) {
            const uint32_t s = offset + step + rank;
            uint32_t s12;  // s * 12 as shift-and-add (the 32-bit multiplier runs at a quarter of the rate)
            asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(s12) : "v"(s), "v"(s << 2));
            float* px = reinterpret_cast<float*>(reinterpret_cast<char*>(xyzs) + (size_t)s12);
            float* pd = reinterpret_cast<float*>(reinterpret_cast<char*>(dirs) + (size_t)s12);
            float* pl = reinterpret_cast<float*>(reinterpret_cast<char*>(deltas) + (size_t)(s * 8u));
            px[0] = bx; px[1] = by; px[2] = bz;
            pd[0] = dv.x; pd[1] = dv.y; pd[2] = dv.z;
            pl[0] = bdt; pl[1] = d1;
        } else {
            const size_t s = (size_t)offset + step + rank;
            xyzs[3 * s] = bx; xyzs[3 * s + 1] = by; xyzs[3 * s + 2] = bz;
            dirs[3 * s] = dv.x; dirs[3 * s + 1] = dv.y; dirs[3 * s + 2] = dv.z;
            deltas[2 * s] = bdt; deltas[2 * s + 1] = d1;
        }
    }
    last_t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_after), top));
    step += (uint32_t)__builtin_popcountll(S);
}

// Everything a wave keeps of a ray between its count and its stores (two per wave: one being counted, one waiting for its range)
template <int BT>
struct MarchRayLds {
    MarchRec rec[kMarchRecCap];
    float bt[BT][kWave];
};

__device__ __forceinline__ float march_t_start(const Marcher& m, const float* __restrict__ nears, const float* __restrict__ noises, uint32_t n) {
    const float t = nears[n];
    return t + m.step_len(t) * noises[n];
}

// wave = ray n: count it once; remember the sample-bearing batches
template <int BT>
__device__ __forceinline__ void march_count_ray(MarchRayLds<BT>& L, int lane, uint32_t n, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                const uint8_t* __restrict__ grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                                uint32_t H, const uint32_t* lut, const float* __restrict__ nears, const float* __restrict__ fars,
                                                const float* __restrict__ noises, int serial, uint32_t& count, uint32_t& nrec) {
    Marcher m;
    m.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
    m.use_lut(lut);
    const float far = fars[n];
    ChainWalker w;
    w.init(march_t_start(m, nears, noises, n));
    count = 0;
    nrec = 0;
    while (count < max_steps) {
        const unsigned long long S = w.next_samples(m, far, lane, max_steps - count, serial != 0);
        if (!S) break;
        if (nrec < (uint32_t)kMarchRecCap && lane == 0) {
            MarchRec r;
            r.t0 = w.t_batch; r.nb = (uint32_t)w.nb; r.S = S;
            L.rec[nrec] = r;
        }
        if (nrec < (uint32_t)BT) L.bt[nrec][lane] = w.bt;
        ++nrec;
        count += (uint32_t)__builtin_popcountll(S);
    }
}

// wave = ray n, whose `count` samples start at `offset`: write rays[n] and replay the records into the sample arrays
template <bool OFF32, int BT>
__device__ __forceinline__ void march_store_ray(const MarchRayLds<BT>& L, uint32_t offset, uint32_t count, uint32_t nrec, int lane, uint32_t n, uint32_t M,
                                                const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                const uint8_t* __restrict__ grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                                uint32_t H, const uint32_t* lut, const float* __restrict__ nears, const float* __restrict__ fars,
                                                const float* __restrict__ noises, int* __restrict__ rays, float* __restrict__ xyzs,
                                                float* __restrict__ dirs, float* __restrict__ deltas, int serial) {
    if (lane == 0) {
        rays[3 * (size_t)n + 0] = (int)n;
        rays[3 * (size_t)n + 1] = (int)offset;
        rays[3 * (size_t)n + 2] = (int)count;
    }
    if (count == 0 || offset + count > M) return;
    Marcher m;
    m.init_replay(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H);
    const float t_start = march_t_start(m, nears, noises, n);
    float last_t = t_start;
    uint32_t step = 0;
    MarchDirs dv = {m.dx, m.dy, m.dz};
    asm volatile("" : "+v"(dv.x), "+v"(dv.y), "+v"(dv.z));  // three registers for the whole loop instead of three moves per batch
    if (nrec <= (uint32_t)kMarchRecCap) {
        for (uint32_t r = 0; r < nrec; ++r) {
            const MarchRec rec = L.rec[r];
            // the record is the same in every lane: as scalars, so that the mask arithmetic of the stores runs on the scalar unit
            const unsigned long long S = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rec.S >> 32)) << 32) |
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rec.S);
            float bt;
            if (r < (uint32_t)BT) {
                bt = L.bt[r][lane];
            } else {
                float tn;
                int nb;
                m.fill_batch(rec.t0, lane, bt, nb, tn);
            }
            // the sample description of Marcher::classify_cell: position clamped to the box, step length of the member
            const float bx = clamp_med3(m.ox + bt * m.dx, -m.bound, m.bound);
            const float by = clamp_med3(m.oy + bt * m.dy, -m.bound, m.bound);
            const float bz = clamp_med3(m.oz + bt * m.dz, -m.bound, m.bound);
            float bdt = m.dt_const;
            if (m.dt_gamma != 0.0f) {
                asm volatile("");
                bdt = m.step_len(bt);
            }
            march_store_batch<OFF32>(dv, bt, bx, by, bz, bdt, S, lane, offset, step, last_t, xyzs, dirs, deltas);
        }
    } else {  // more sample-bearing batches than LDS records: march again, as the three-launch write pass does
        m.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
        m.use_lut(lut);
        const float far = fars[n];
        ChainWalker w;
        w.init(t_start);
        while (step < count) {
            const unsigned long long S = w.next_samples(m, far, lane, count - step, serial != 0);
            if (!S) break;
            march_store_batch<OFF32>(dv, w.bt, w.bx, w.by, w.bz, w.bdt, S, lane, offset, step, last_t, xyzs, dirs, deltas);
        }
    }
}

// The scanner (one wave): sums[t] = {1, samples of ticket t} in, prefix[t] = {1, samples of the tickets before t} out, in ticket
// order as far as the published sums reach, kWindows x 64 tickets per poll (the windows are requested together: one memory
// latency per poll).  Closes the call: counter[0] = total, counter[1] += N -- what the three-launch form leaves.
__device__ __forceinline__ void march_scanner(const unsigned long long* sums, unsigned long long* prefix, uint32_t n_tickets, uint32_t N, int lane,
                                              int* __restrict__ counter, uint32_t spin_limit) {
    constexpr int kWindows = 8;
    constexpr unsigned long long kReady = 1ull << 32;
    uint32_t frontier = 0, running = (uint32_t)counter[0], idle = 0;
    bool expired = false;
    while (frontier < n_tickets) {
        unsigned long long f[kWindows];
#pragma unroll
        for (int k = 0; k < kWindows; ++k) {
            const uint32_t i = frontier + (uint32_t)(64 * k + lane);
            f[k] = 0ull;
            if (i < n_tickets) f[k] = __hip_atomic_load(&sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t advanced = 0;
        bool open = true;  // wave-uniform: every window so far was published completely
#pragma unroll
        for (int k = 0; k < kWindows; ++k) {
            if (open) {
                const uint32_t i = frontier + (uint32_t)(64 * k + lane);
                const bool there = (uint32_t)(f[k] >> 32) == 1u;
                const unsigned long long missing = ~__ballot(there);
                const int lead = missing ? __builtin_ctzll(missing) : 64;  // published tickets at the head of the window
                const uint32_t value = lane < lead ? (uint32_t)f[k] : 0u;
                const uint32_t incl = wave_scan_add_u32(value);
                if (lane < lead) __hip_atomic_store(&prefix[i], kReady | (unsigned long long)(running + incl - value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                advanced += (uint32_t)lead;
                open = lead == 64;  // (a window past the last ticket ends the round as well: its flags read as unpublished)
            }
        }
        frontier += advanced;
        if (advanced) { idle = 0; continue; }
        if (++idle >= spin_limit) { expired = true; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) {
        // counter[1]: rays marched so far, NEGATIVE once a bounded spin expired anywhere (the launch terminates, the result is
        // invalid).  Scanner and workers may finish in any order: the sign bit is set with an atomic OR and the ray count is added
        // with an atomic ADD, so neither loses the other's update.
        if (expired) {
            atomicOr(reinterpret_cast<unsigned int*>(counter + 1), 0x80000000u);
        } else {
            counter[0] = (int)running;
            atomicAdd(reinterpret_cast<unsigned int*>(counter + 1), N);
        }
    }
}

// Tickets come from kMarchQueues counters, each on a cache line of its own, instead of one: a returning atomic add on ONE
// address is served at ~90 per microsecond chip-wide (MI355X_MICROARCH.md, price list: dequeue), which 8192 tickets drawn by ~1500
// workgroups run into (each draw sits in front of the workgroup's barrier).  Queue q hands out tickets q, q + Q, q + 2Q, ...; a
// workgroup starts at queue blockIdx % Q -- with the round-robin placement of workgroups that is one queue per XCD and an equal
// number of workgroups per queue -- and moves on to another queue when its own is exhausted, so every ticket is drawn exactly
// once, by a running workgroup.
// Measured (32 768 rays x 1024 steps, one box): one counter 0.203 / 0.216 ms (dense / 10 % grid), 8 queues 0.191 / 0.205, 16 queues
// 0.185 / 0.210, 32 queues 0.197 / 0.223 (fewer workgroups per queue: the queues drift apart); with 128-step rays -- two batches
// per ray, nothing but the scaffolding -- 0.129 -> 0.082 ms.
//
// Progress invariant (the queues need NOT advance in step).  A worker waits for the exclusive prefix of its previous ticket p
// while it may already HOLD the next, still uncounted ticket t, drawn early so that the draw's round trip hides behind the count
// phase.  The scanner hands out prefixes in ticket order, so prefix(p) needs sum(t) whenever t < p -- which this very worker would
// publish only after the wait.  Therefore a ticket is drawn EARLY only from the queue the pending ticket came from (tickets of one
// queue grow: t > p); whenever the next ticket has to come from another queue -- the own one is exhausted, or the current ticket
// was itself taken from another queue than the pending one -- it is drawn LATE, after the pending ticket's wait and stores, when
// the worker holds nothing uncounted.  Then: the worker holding the smallest unpublished ticket never waits for a larger one, and
// every queue that still has tickets keeps the workgroups that started on it (blockIdx % Q; for fewer than Q workers each queue
// holds at most one ticket and the first, unconditional late draw hands all of them out), whose waits are for smaller tickets
// of that queue -- by induction on the smallest unpublished ticket some worker always advances.  A queue slowed down by a
// concurrent kernel on its XCD delays the launch; it cannot stall it until the spin limit.
constexpr uint32_t kMarchQueues = 8;
constexpr uint32_t kMarchHeadWords = 16 * kMarchQueues;   // 8-byte words in front of the flags: one 128-byte line per queue
constexpr uint32_t kTicketNone = 0xFFFFFFFFu;   // every queue is exhausted
constexpr uint32_t kTicketLate = 0xFFFFFFFEu;   // to be drawn after the pending ticket has been stored
__device__ __forceinline__ uint32_t march_draw_own(unsigned int* heads, uint32_t n_tickets, uint32_t q) {
    const unsigned long long b = (unsigned long long)atomicAdd(&heads[32u * q], 1u) * kMarchQueues + q;
    return b < n_tickets ? (uint32_t)b : kTicketLate;
}
__device__ __forceinline__ uint32_t march_draw_any(unsigned int* heads, uint32_t n_tickets, uint32_t q0) {
    uint32_t head[kMarchQueues];  // looked at together (one memory latency): only queues that show tickets are drawn from
#pragma unroll
    for (uint32_t k = 0; k < kMarchQueues; ++k) head[k] = __hip_atomic_load(&heads[32u * ((q0 + k) % kMarchQueues)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (uint32_t k = 0; k < kMarchQueues; ++k) {
        const uint32_t q = (q0 + k) % kMarchQueues;
        if ((unsigned long long)head[k] * kMarchQueues + q >= n_tickets) continue;
        const unsigned long long b = (unsigned long long)atomicAdd(&heads[32u * q], 1u) * kMarchQueues + q;
        if (b < n_tickets) return (uint32_t)b;
    }
    return kTicketNone;
}

// PLAIN: dt_gamma == 0 and the batch-parallel walk -- the reference's defaults -- are compile-time facts of the instance: the
// step-length recurrences, the per-member dt-level and the serial-walk switch fold away (fewer instructions and fewer live scalars).
template <bool OFF32, bool PLAIN, int BT>
__device__ __forceinline__ void march_train_onepass_body(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, float bound, float dt_gamma_arg,
    uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* __restrict__ nears, const float* __restrict__ fars,
    const float* __restrict__ noises, int* __restrict__ rays, int* __restrict__ counter, unsigned long long* __restrict__ ws,
    float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, int serial_arg, uint32_t spin_limit, uint32_t skew) {
    const float dt_gamma = PLAIN ? 0.0f : dt_gamma_arg;
    const int serial = PLAIN ? 0 : serial_arg;
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
    const uint32_t n_tickets = (N + (uint32_t)kMarchRays - 1u) / (uint32_t)kMarchRays;
    unsigned int* heads = reinterpret_cast<unsigned int*>(ws);
    unsigned long long* sums = ws + kMarchHeadWords;
    unsigned long long* prefix = ws + kMarchHeadWords + n_tickets;
    if (blockIdx.x == 0) {
        if (wid == 0) march_scanner(sums, prefix, n_tickets, N, lane, counter, spin_limit);
        return;
    }
    extern __shared__ uint32_t s_lut[];  // H entries
    __shared__ MarchRayLds<BT> s_ray[kMarchRays][2];
    __shared__ uint32_t s_late;
    __shared__ uint32_t s_cnt[2][kMarchRays], s_next[2], s_arrived[2];  // by parity of the iteration: written before its barrier, read right after it
    fill_spread_lut(s_lut, H);
    if (threadIdx.x < 2) s_arrived[threadIdx.x] = 0u;
    if (skew && blockIdx.x % kMarchQueues == skew - 1u && threadIdx.x == 0)
        for (int i = 0; i < 300; ++i) __builtin_amdgcn_s_sleep(64);  // tests: this queue's workgroups start ~0.5 ms late
    if (threadIdx.x == 0) s_next[1] = march_draw_any(heads, n_tickets, blockIdx.x % kMarchQueues);  // ticket = position in the scan
    __syncthreads();
    uint32_t b = s_next[1];
    int cur = 0;
    bool pending = false;
    uint32_t pend_n = 0, pend_ticket = 0, pend_before = 0, pend_count = 0, pend_nrec = 0;
    while (true) {
        const bool have = b < n_tickets;
        const uint32_t n = __builtin_amdgcn_readfirstlane(b * (uint32_t)kMarchRays + (uint32_t)wid);
        uint32_t count = 0, nrec = 0, before = 0;
        if (have) {
            if (n < N)
                march_count_ray<BT>(s_ray[wid][cur], lane, n, rays_o, rays_d, grid, bound, dt_gamma, max_steps, C, H, s_lut, nears, fars, noises, serial, count, nrec);
            if (lane == 0) s_cnt[cur][wid] = count;
            // the next ticket is drawn late: the scanner works in ticket order, and a ticket drawn long before it is counted holds
            // everybody's ranges back (drawn before the count: 0.274 ms instead of 0.25) -- by the wave that finishes its ray FIRST, so
            // that the draw's round trip runs while the other rays of the ticket are still being counted
            if (lane == 0 && atomicAdd(&s_arrived[cur], 1u) == 0u) {
                // early only from the queue of the ticket this iteration is going to wait for (progress invariant above)
                const uint32_t q = b % kMarchQueues;
                s_next[cur] = (!pending || pend_ticket % kMarchQueues == q) ? march_draw_own(heads, n_tickets, q) : kTicketLate;
                s_arrived[cur ^ 1] = 0u;  // the other parity's counter: last used before the previous barrier, next used after this one
            }
            __syncthreads();  // one barrier per ticket: the four counts -> the ticket's sum and every wave's place inside the ticket
            uint32_t sum = 0;
#pragma unroll
            for (int r = 0; r < kMarchRays; ++r) {
                const uint32_t c = s_cnt[cur][r];
                before += r < wid ? c : 0u;
                sum += c;
            }
            if (threadIdx.x == 0) __hip_atomic_store(&sums[b], (1ull << 32) | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pending) {  // the previous ticket: its predecessors and the scanner had a whole count phase
            unsigned long long f = 0ull;
            if (lane == 0) {
                uint32_t polls = 0;
                while (true) {
                    f = __hip_atomic_load(&prefix[pend_ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(f >> 32) == 1u || ++polls >= spin_limit) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            const uint32_t status = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(f >> 32));
            const uint32_t excl = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)f);
            if (status != 1u) {
                if (lane == 0) atomicOr(reinterpret_cast<unsigned int*>(counter + 1), 0x80000000u);
            } else if (pend_n < N) {
                march_store_ray<OFF32, BT>(s_ray[wid][cur ^ 1], excl + pend_before, pend_count, pend_nrec, lane, pend_n, M, rays_o, rays_d, grid, bound, dt_gamma,
                                       max_steps, C, H, s_lut, nears, fars, noises, rays, xyzs, dirs, deltas, serial);
            }
        }
        if (!have) break;
        pending = true;
        pend_n = n;
        pend_ticket = b;
        pend_before = before;
        pend_count = count;
        pend_nrec = nrec;
        uint32_t next = s_next[cur];
        if (next == kTicketLate) {  // (the same value in every wave) nothing uncounted is held now: any queue may be drawn from
            if (threadIdx.x == 0) s_late = march_draw_any(heads, n_tickets, b % kMarchQueues);
            __syncthreads();  // s_late's previous readers are at least one count-phase barrier behind
            next = s_late;
        }
        b = next;
        cur ^= 1;
    }
}

#define NVSF_MARCH_ONEPASS_KERNEL(NAME, WAVES, BT)                                                                                                  \
    template <bool OFF32, bool PLAIN>                                                                                                       \
    __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void NAME(                                     \
        const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid, float bound, float dt_gamma,  \
        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* __restrict__ nears,                                \
        const float* __restrict__ fars, const float* __restrict__ noises, int* __restrict__ rays, int* __restrict__ counter,                \
        unsigned long long* __restrict__ ws, float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, int serial,    \
        uint32_t spin_limit, uint32_t skew) {                                                                                               \
        march_train_onepass_body<OFF32, PLAIN, BT>(rays_o,       rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, noises, rays,     \
                                               counter, ws, xyzs, dirs, deltas, serial, spin_limit, skew);                                  \
    }
// Six waves per SIMD: 80 registers hold the marcher's state without spills (at eight waves = 64 registers, 12-14 of them went to
// scratch, whose loads and stores queue behind the sample stores on the wave's memory counter), and six workgroups per compute
// unit leave LDS for the member parameters of 11 batches per ray (a camera ray through a full grid has ~12).  Measured (32 768
// rays x 1024 steps, dense grid): 0.233 -> 0.211 ms.
constexpr uint32_t kMarchWgPerCu = 6;
NVSF_MARCH_ONEPASS_KERNEL(k_march_train_onepass, 6, 11)

// pass 2: one workgroup; exclusive scan of the counts in ray order, reserving [counter[0], +total).
// A thread takes kScanPer consecutive rays per round (4096 rays per round: one round at the BASELINE batch size).
constexpr uint32_t kScanPer = 4;
__global__ __launch_bounds__(1024) void k_march_scan(uint32_t N, int* __restrict__ rays, int* __restrict__ counter) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) carry_s = (uint32_t)counter[0];
    __syncthreads();
    for (uint32_t base = 0; base < N; base += 1024 * kScanPer) {
        const uint32_t n0 = base + threadIdx.x * kScanPer;
        uint32_t c[kScanPer], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < kScanPer; ++i) {
            c[i] = n0 + i < N ? (uint32_t)rays[3 * (size_t)(n0 + i) + 2] : 0u;
            sum += c[i];
        }
        const uint32_t incl = wave_scan_add_u32(sum);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t before = carry_s;
        for (int w = 0; w < wid; ++w) before += wave_tot[w];
        uint32_t run = before + incl - sum;
#pragma unroll
        for (uint32_t i = 0; i < kScanPer; ++i) {
            if (n0 + i < N) {
                rays[3 * (size_t)(n0 + i) + 0] = (int)(n0 + i);
                rays[3 * (size_t)(n0 + i) + 1] = (int)run;
            }
            run += c[i];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counter[0] = (int)carry_s;
        counter[1] += (int)N;
    }
}

// pass 3: re-march and write the packed samples
__global__ __launch_bounds__(kBlock) void k_march_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                        const float* __restrict__ nears, const float* __restrict__ fars,
                                                        const float* __restrict__ noises, const int* __restrict__ rays,
                                                        float* __restrict__ xyzs, float* __restrict__ dirs,
                                                        float* __restrict__ deltas) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t offset = (uint32_t)rays[3 * (size_t)n + 1], count = (uint32_t)rays[3 * (size_t)n + 2];
    if (count == 0 || offset + count > M) return;
    Marcher m;
    m.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
    const float far = fars[n];
    float t = nears[n];
    t += m.step_len(t) * noises[n];
    float last_t = t;
    float* px = xyzs + 3 * (size_t)offset;
    float* pd = dirs + 3 * (size_t)offset;
    float* pl = deltas + 2 * (size_t)offset;
    uint32_t step = 0;
    float x, y, z, dt;
    while (t < far && step < count) {
        if (m.probe(t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
            t += dt;
            pl[0] = dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            ++step;
        }
    }
}

// inference marcher: fixed n_step slots per alive ray
__global__ __launch_bounds__(kBlock) void k_march_rays(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                                                       const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                       const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                       uint32_t max_steps, uint32_t C, uint32_t H,
                                                       const uint8_t* __restrict__ grid, const float* __restrict__ fars,
                                                       float* __restrict__ xyzs, float* __restrict__ dirs,
                                                       float* __restrict__ deltas, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    Marcher m;
    m.init(rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid, bound, dt_gamma, max_steps, C, H);
    float* px = xyzs + 3 * (size_t)n * n_step;
    float* pd = dirs + 3 * (size_t)n * n_step;
    float* pl = deltas + 2 * (size_t)n * n_step;
    const float far = fars[index];
    float t = rays_t[index];
    t += m.step_len(t) * noises[n];
    float last_t = t;
    uint32_t step = 0;
    float x, y, z, dt;
    while (t < far && step < n_step) {
        if (m.probe(t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
            t += dt;
            pl[0] = dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            ++step;
        }
    }
}

// n_step == 8: the samples of a ray are collected in registers and leave as whole records (xyz / dirs 96 B, deltas 64 B
// per ray, 16-byte stores).  With one 12-byte store per step the 128-B lines of a ray's records are completed over the
// ~10 us of its march while 134 MB of such lines are in flight -- more than the L2s hold -- and reach HBM in pieces.
__global__ __launch_bounds__(kBlock) void k_march_rays8(uint32_t n_alive, const int* __restrict__ rays_alive,
                                                        const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                        uint32_t max_steps, uint32_t C, uint32_t H,
                                                        const uint8_t* __restrict__ grid, const float* __restrict__ fars,
                                                        float* __restrict__ xyzs, float* __restrict__ dirs,
                                                        float* __restrict__ deltas, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n - (uint32_t)lane_id() >= n_alive) return;  // whole wave past the end
    const bool valid = n < n_alive;                    // lanes past the end of the last wave march nothing but help to store
    const int index = rays_alive[valid ? n : n_alive - 1u];
    Marcher m;
    m.init(rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid, bound, dt_gamma, max_steps, C, H);
    const float far = valid ? fars[index] : -1.0f;
    float t = rays_t[index];
    t += m.step_len(t) * noises[valid ? n : n_alive - 1u];
    float last_t = t;
    float rec[8][5];  // x, y, z, dt, t_new - last_t; unfilled slots stay zero (they signal termination to composite_rays)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 5; ++i) rec[k][i] = 0.0f;
    uint32_t step = 0;
    float x, y, z, dt;
    while (t < far && step < 8u) {
        if (m.probe(t, x, y, z, dt)) {
            t += dt;
            const float d1 = t - last_t;
            last_t = t;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (step == (uint32_t)k) { rec[k][0] = x; rec[k][1] = y; rec[k][2] = z; rec[k][3] = dt; rec[k][4] = d1; }
            ++step;
        }
    }
    // The records of a wave's 64 rays are contiguous in memory (64 x 96 B, 64 x 96 B, 64 x 64 B): staged through LDS, every store
    // instruction writes 1 KB of consecutive bytes (whole 128-byte lines) instead of 64 pieces of 16 bytes 96 bytes apart.
    __shared__ float s_rec[kBlock / kWave][kWave * 24 + 8];
    float* stage = s_rec[threadIdx.x >> 6];
    const int lane = lane_id();
    const uint32_t wave_first = n - (uint32_t)lane;                       // first ray of this wave
    const uint32_t wave_rays = n_alive - wave_first < (uint32_t)kWave ? n_alive - wave_first : (uint32_t)kWave;
    auto flush = [&](float* __restrict__ dst, uint32_t per_ray) {         // stage holds wave_rays x per_ray floats, ray-major
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t n4 = wave_rays * per_ray / 4u;
        float4* out = reinterpret_cast<float4*>(dst + (size_t)per_ray * wave_first);
        for (uint32_t e = (uint32_t)lane; e < n4; e += (uint32_t)kWave) out[e] = *reinterpret_cast<const float4*>(stage + 4u * e);
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        stage[24 * lane + 3 * k] = rec[k][0];
        stage[24 * lane + 3 * k + 1] = rec[k][1];
        stage[24 * lane + 3 * k + 2] = rec[k][2];
    }
    flush(xyzs, 24u);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool filled = (uint32_t)k < step;
        stage[24 * lane + 3 * k] = filled ? m.dx : 0.0f;
        stage[24 * lane + 3 * k + 1] = filled ? m.dy : 0.0f;
        stage[24 * lane + 3 * k + 2] = filled ? m.dz : 0.0f;
    }
    flush(dirs, 24u);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        stage[16 * lane + 2 * k] = rec[k][3];
        stage[16 * lane + 2 * k + 1] = rec[k][4];
    }
    flush(deltas, 16u);
}

// ------------------------------------------------------------------------------------------------
// Packed-sample compositor, one wave per ray.  Lanes take 64 consecutive samples per round; the
// transmittance is a cross-lane exclusive product scan carried between rounds.
constexpr int kRaysPerBlock = kBlock / kWave;

__global__ __launch_bounds__(kBlock) void k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                uint32_t M, uint32_t N, float T_thresh,
                                                                float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                   count = (uint32_t)rays[3 * (size_t)n + 2];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (count != 0 && offset + count <= M) {
        float T_carry = 1.0f, t_carry = 0.0f;
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < count;
            float sg = 0, d0 = 0, d1 = 0, cr = 0, cg = 0, cb = 0;
            if (valid) {
                const size_t p = (size_t)offset + i;
                sg = sigmas[p];
                const float2 dl = reinterpret_cast<const float2*>(deltas)[p];
                d0 = dl.x; d1 = dl.y;
                cr = rgbs[3 * p]; cg = rgbs[3 * p + 1]; cb = rgbs[3 * p + 2];
            }
            const float alpha = valid ? 1.0f - expf(-sg * d0) : 0.0f;
            const float om = 1.0f - alpha;
            const float incl = wave_scan_mul(om);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T_before = T_carry * excl, T_after = T_carry * incl;
            const unsigned long long stop = __ballot(valid && (T_after < T_thresh));
            const int first_stop = stop ? (int)__builtin_ctzll(stop) : 64;
            const float w = (valid && lane <= first_stop) ? alpha * T_before : 0.0f;
            const float t_here = t_carry + wave_scan_add(d1);
            r += w * cr; g += w * cg; b += w * cb;
            d += w * t_here;
            ws += w;
            if (stop) break;
            T_carry = __shfl(T_after, 63, 64);
            t_carry = __shfl(t_here, 63, 64);
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    }
    if (lane == 0) {
        weights_sum[index] = ws;
        depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}

__global__ __launch_bounds__(kBlock) void k_composite_train_bwd(const float* __restrict__ grad_ws, const float* __restrict__ grad_image,
                                                                const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                uint32_t M, uint32_t N, float T_thresh,
                                                                float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                   count = (uint32_t)rays[3 * (size_t)n + 2];
    if (count == 0 || offset + count > M) return;
    const float gws = grad_ws[index];
    const float gr = grad_image[3 * (size_t)index], gg = grad_image[3 * (size_t)index + 1], gb = grad_image[3 * (size_t)index + 2];
    const float rF = image[3 * (size_t)index], gF = image[3 * (size_t)index + 1], bF = image[3 * (size_t)index + 2];
    const float wsF = weights_sum[index];
    float T_carry = 1.0f, r_carry = 0, g_carry = 0, b_carry = 0;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < count;
        const size_t p = (size_t)offset + (valid ? i : 0);
        float sg = 0, d0 = 0, cr = 0, cg = 0, cb = 0;
        if (valid) {
            sg = sigmas[p];
            d0 = deltas[2 * p];
            cr = rgbs[3 * p]; cg = rgbs[3 * p + 1]; cb = rgbs[3 * p + 2];
        }
        const float alpha = valid ? 1.0f - expf(-sg * d0) : 0.0f;
        const float incl = wave_scan_mul(1.0f - alpha);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T_before = T_carry * excl, T_after = T_carry * incl;
        const unsigned long long stop = __ballot(valid && (T_after < T_thresh));
        const int first_stop = stop ? (int)__builtin_ctzll(stop) : 64;
        const bool live = valid && lane <= first_stop;
        const float w = live ? alpha * T_before : 0.0f;
        const float r = r_carry + wave_scan_add(w * cr);
        const float g = g_carry + wave_scan_add(w * cg);
        const float b = b_carry + wave_scan_add(w * cb);
        if (live) {
            grad_rgbs[3 * p] = gr * w; grad_rgbs[3 * p + 1] = gg * w; grad_rgbs[3 * p + 2] = gb * w;
            grad_sigmas[p] = d0 * (gr * (T_after * cr - (rF - r)) + gg * (T_after * cg - (gF - g)) +
                                   gb * (T_after * cb - (bF - b)) + gws * (1.0f - wsF));
        }
        if (stop) break;
        T_carry = __shfl(T_after, 63, 64);
        r_carry = __shfl(r, 63, 64); g_carry = __shfl(g, 63, 64); b_carry = __shfl(b, 63, 64);
    }
}

// inference compositor: n_step is small (1..8 in practice) and n_alive large -> one lane per ray.
__global__ __launch_bounds__(kBlock) void k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                           int* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                           const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                           const float* __restrict__ deltas, float* __restrict__ weights_sum,
                                                           float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float* s = sigmas + (size_t)n * n_step;
    const float* c = rgbs + 3 * (size_t)n * n_step;
    const float* dl = deltas + 2 * (size_t)n * n_step;
    float t = rays_t[index], ws = weights_sum[index], d = depth[index];
    float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0.0f) break;
        const float alpha = 1.0f - expf(-s[0] * dl[0]);
        const float T = 1.0f - ws;
        const float w = alpha * T;
        ws += w;
        t += dl[1];
        d += w * t;
        r += w * c[0]; g += w * c[1]; b += w * c[2];
        if (T < T_thresh) break;
        ++s; c += 3; dl += 2; ++step;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[index] = t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
}

// The same accumulation with EIGHT lanes per ray (n_step <= 8): lane j of a group loads slot j of its ray, so the loads of
// a wave are contiguous (sigma 4 B, rgb 12 B, deltas 8 B per lane) instead of 64 lanes striding over 8-slot records; the
// serial recurrence then runs on values broadcast inside the group (every lane of a group carries the same state, lane 0
// stores it).  Same operations in the same order as k_composite_rays: identical results.
__global__ __launch_bounds__(kBlock) void k_composite_rays_g8(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                              int* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                              const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                              const float* __restrict__ deltas, float* __restrict__ weights_sum,
                                                              float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n = tid >> 3, j = tid & 7u;
    const bool ray_ok = n < n_alive;
    const uint32_t nn = ray_ok ? n : n_alive - 1;
    const bool has = ray_ok && j < n_step;
    const size_t slot = (size_t)nn * n_step + (has ? j : 0u);
    const float sg = has ? sigmas[slot] : 0.0f;
    const float d0 = has ? deltas[2 * slot] : 0.0f, d1 = has ? deltas[2 * slot + 1] : 0.0f;
    const float c0 = has ? rgbs[3 * slot] : 0.0f, c1 = has ? rgbs[3 * slot + 1] : 0.0f, c2 = has ? rgbs[3 * slot + 2] : 0.0f;
    const int index = rays_alive[nn];
    float t = rays_t[index], ws = weights_sum[index], d = depth[index];
    float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
    uint32_t step = 0;
    bool stopped = false;
    for (uint32_t k = 0; k < 8u; ++k) {  // uniform trip count: the shuffles need every lane
        const float sk = __shfl(sg, (int)k, 8), dk0 = __shfl(d0, (int)k, 8), dk1 = __shfl(d1, (int)k, 8);
        const float ck0 = __shfl(c0, (int)k, 8), ck1 = __shfl(c1, (int)k, 8), ck2 = __shfl(c2, (int)k, 8);
        if (stopped || k >= n_step) continue;
        if (dk0 == 0.0f) { stopped = true; continue; }
        const float alpha = 1.0f - expf(-sk * dk0);
        const float T = 1.0f - ws;
        const float w = alpha * T;
        ws += w;
        t += dk1;
        d += w * t;
        r += w * ck0; g += w * ck1; b += w * ck2;
        if (T < T_thresh) { stopped = true; continue; }
        ++step;
    }
    if (ray_ok && j == 0u) {
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;
        weights_sum[index] = ws;
        depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}

}  // namespace

// ================================================================================================
// C-ABI (declared and documented in include/nvsf_hip.h)
// ================================================================================================
#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                     float min_near, float* nears, float* fars, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && aabb && nears && fars);
    hipLaunchKernelGGL(k_near_far, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, rays_o, rays_d, aabb, N, min_near, nears, fars);
    return nvsf_launch_status();
}

NVSF_API int nvsf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                               hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && coords);
    hipLaunchKernelGGL(k_sph_from_ray, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, rays_o, rays_d, radius, N, coords);
    return nvsf_launch_status();
}

NVSF_API int nvsf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(coords && indices);
    hipLaunchKernelGGL(k_morton3D, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, coords, N, indices);
    return nvsf_launch_status();
}

NVSF_API int nvsf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(coords && indices);
    hipLaunchKernelGGL(k_morton3D_invert, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, indices, N, coords);
    return nvsf_launch_status();
}

NVSF_API int nvsf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(grid && bitfield);
    REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15u) == 0);
    hipLaunchKernelGGL(k_packbits, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, grid, N, density_thresh, bitfield);
    return nvsf_launch_status();
}

// count / scan / write as three launches: no inter-workgroup wait, no scratch (the form the one-launch kernel falls back to)
NVSF_API int nvsf_march_rays_train_passes(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                          uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                          const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays,
                                          int32_t* counter, const float* noises, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter && noises);
    REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 1024 && max_steps >= 1);
    const int variant = nvsf_variant(kVarMarch);  // tests: 1 = one thread per ray (first formulation), 2 = wave kernels, batch walked member by member
    if (variant == 1) {
        hipLaunchKernelGGL(k_march_count, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, rays_o, rays_d, grid, bound, dt_gamma,
                           max_steps, N, C, H, nears, fars, noises, rays);
        hipLaunchKernelGGL(k_march_scan, dim3(1), dim3(1024), 0, stream, N, rays, counter);
        hipLaunchKernelGGL(k_march_write, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, stream, rays_o, rays_d, grid, bound, dt_gamma,
                           max_steps, N, C, H, M, nears, fars, noises, rays, xyzs, dirs, deltas);
        return nvsf_launch_status();
    }
    const int serial = variant == 2;
    const dim3 wgrid(cdiv(N, kBlock / kWave));
    hipLaunchKernelGGL(k_march_count_wave, wgrid, dim3(kBlock), 0, stream, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H,
                       nears, fars, noises, rays, serial);
    hipLaunchKernelGGL(k_march_scan, dim3(1), dim3(1024), 0, stream, N, rays, counter);
    hipLaunchKernelGGL(k_march_write_wave, wgrid, dim3(kBlock), 0, stream, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M,
                       nears, fars, noises, rays, xyzs, dirs, deltas, serial);
    return nvsf_launch_status();
}

// compute units of the current device (one process drives one GPU): the persistent launch below fills each with kMarchWgPerCu workgroups
static int march_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// ticket queue heads (one 128-byte line each) + {sum, exclusive prefix} flag per ticket of kMarchRays rays
NVSF_API size_t nvsf_march_rays_train_ws_bytes(uint32_t N) { return 8u * (size_t)(kMarchHeadWords + 2u * cdiv(N, (uint32_t)kMarchRays)); }

NVSF_API int nvsf_march_rays_train_ws(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                      uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                      const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                      const float* noises, void* workspace, size_t workspace_bytes, uint32_t spin_limit, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter && noises && workspace);
    REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 1024 && max_steps >= 1);
    REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0 && workspace_bytes >= nvsf_march_rays_train_ws_bytes(N));
    const int serial = nvsf_variant(kVarMarch) == 2;  // tests: the batch walked member by member
    if (hipMemsetAsync(workspace, 0, nvsf_march_rays_train_ws_bytes(N), stream) != hipSuccess) return (int)hipGetLastError();
    const bool off32 = (unsigned long long)M * 12ull < (1ull << 32), plain = dt_gamma == 0.0f && !serial;
    auto kernel = off32 ? (plain ? k_march_train_onepass<true, true> : k_march_train_onepass<true, false>)
                        : (plain ? k_march_train_onepass<false, true> : k_march_train_onepass<false, false>);
    // the scanner + one worker per ticket until the chip is full (kMarchWgPerCu workgroups per compute unit)
    const uint32_t wanted = 1u + cdiv(N, (uint32_t)kMarchRays), resident = kMarchWgPerCu * (uint32_t)march_cu_count();
    hipLaunchKernelGGL(kernel, dim3(wanted < resident ? wanted : resident), dim3(kBlock), H * sizeof(uint32_t), stream, rays_o, rays_d, grid, bound, dt_gamma,
                       max_steps, N, C, H, M, nears, fars, noises, rays, counter, reinterpret_cast<unsigned long long*>(workspace), xyzs, dirs,
                       deltas, serial, spin_limit ? spin_limit : kMarchSpinLimit, (uint32_t)nvsf_variant(kVarMarchSkew));
    return nvsf_launch_status();
}

// Scratch of the reference-shaped entry below.  SURVEY 8b's ownership rule is "native code never allocates": the entry points that
// take a caller-owned workspace (nvsf_march_rays_train_ws, what the Python wrapper calls) keep it.  The reference's own argument
// list has no scratch, so ITS entry borrows the block (1 KB + 16 B per four rays) from the device's DEFAULT stream-ordered pool
// (hipMallocAsync / hipFreeAsync on the caller's stream).  That pool's release threshold is 0 out of the box: every stream or device
// synchronisation hands the freed block back to the OS and the next call is a real allocation again.  So the first call
// on a device raises the threshold to kScratchPoolHold = 16 MiB (never lowers one the application has set): the pool then keeps up to that
// many reserved bytes across synchronisations and serves the block from them -- no allocation on the call path after the first
// call, nothing visible to (or taken from) PyTorch's caching allocator beyond those few KB.  nvsf_scratch_pool_stats reports the
// pool's reserved / used bytes and threshold (tests/test_raymarching_gpu.py asserts no growth over 1000 calls).
constexpr unsigned long long kScratchPoolHold = 16ull << 20;  // (the pool reserves in chunks of its own granularity, 2 MiB on this runtime: the hold must cover a chunk)
static int scratch_pool_of_current_device(hipMemPool_t* pool) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetDefaultMemPool(pool, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return dev;
}
static void scratch_pool_hold_once() {
    static bool done[64] = {};
    hipMemPool_t pool;
    const int dev = scratch_pool_of_current_device(&pool);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    done[dev] = true;
    unsigned long long thr = 0;
    if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr) == hipSuccess && thr < kScratchPoolHold) {
        thr = kScratchPoolHold;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    }
    (void)hipGetLastError();
}

NVSF_API int nvsf_scratch_pool_stats(uint64_t* reserved_bytes, uint64_t* used_bytes, uint64_t* release_threshold) {
    hipMemPool_t pool;
    if (scratch_pool_of_current_device(&pool) < 0) return NVSF_ERR_UNSUPPORTED;
    unsigned long long v[3] = {0, 0, 0};
    if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &v[0]) != hipSuccess ||
        hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &v[1]) != hipSuccess ||
        hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &v[2]) != hipSuccess)
        return (int)hipGetLastError();
    if (reserved_bytes) *reserved_bytes = v[0];
    if (used_bytes) *used_bytes = v[1];
    if (release_threshold) *release_threshold = v[2];
    return NVSF_OK;
}

// The reference-shaped entry (raymarching.h:27-44: no scratch argument): the one-launch form on a scratch block taken from and
// returned to the device's stream-ordered pool around the launch (see above: no allocation and no synchronisation after the first
// call); the three-launch form when the pool has nothing to give or a test selected one of the first formulations.  Expiry of the
// bounded wait is reported as by nvsf_march_rays_train_ws.
NVSF_API int nvsf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                   uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                   const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays,
                                   int32_t* counter, const float* noises, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    void* scratch = nullptr;
    const size_t bytes = nvsf_march_rays_train_ws_bytes(N);
    scratch_pool_hold_once();
    if (nvsf_variant(kVarMarch) != 0 || hipMallocAsync(&scratch, bytes, stream) != hipSuccess || !scratch) {
        (void)hipGetLastError();
        return nvsf_march_rays_train_passes(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                            counter, noises, stream);
    }
    const int st = nvsf_march_rays_train_ws(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                            counter, noises, scratch, bytes, 0u, stream);
    const hipError_t fr = hipFreeAsync(scratch, stream);
    return st != NVSF_OK ? st : (fr == hipSuccess ? NVSF_OK : (int)fr);
}

NVSF_API int nvsf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                               uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth,
                                               float* image, hipStream_t stream) {
    if (N == 0) return NVSF_OK;
    REQUIRE(rays && weights_sum && depth && image);
    REQUIRE(M == 0 || (sigmas && rgbs && deltas));
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(cdiv(N, kRaysPerBlock)), dim3(kBlock), 0, stream, sigmas, rgbs, deltas, rays,
                       M, N, T_thresh, weights_sum, depth, image);
    return nvsf_launch_status();
}

NVSF_API int nvsf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                                const float* rgbs, const float* deltas, const int32_t* rays,
                                                const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                                float T_thresh, float* grad_sigmas, float* grad_rgbs, hipStream_t stream) {
    if (N == 0 || M == 0) return NVSF_OK;
    REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs);
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(cdiv(N, kRaysPerBlock)), dim3(kBlock), 0, stream, grad_weights_sum, grad_image,
                       sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    return nvsf_launch_status();
}

NVSF_API int nvsf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                             const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                             uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                             float* dirs, float* deltas, const float* noises, hipStream_t stream) {
    if (n_alive == 0 || n_step == 0) return NVSF_OK;
    REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && noises);
    REQUIRE(C >= 1 && C <= 8 && H >= 2 && H <= 1024 && max_steps >= 1);
    if (n_step == 8u && (reinterpret_cast<uintptr_t>(xyzs) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dirs) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(deltas) & 15u) == 0) {
        hipLaunchKernelGGL(k_march_rays8, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, stream, n_alive, rays_alive, rays_t, rays_o, rays_d,
                           bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
        return nvsf_launch_status();
    }
    hipLaunchKernelGGL(k_march_rays, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, stream, n_alive, n_step, rays_alive, rays_t,
                       rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
    return nvsf_launch_status();
}

NVSF_API int nvsf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                 const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                                 float* image, hipStream_t stream) {
    if (n_alive == 0) return NVSF_OK;
    REQUIRE(rays_alive && rays_t && weights_sum && depth && image);
    REQUIRE(n_step == 0 || (sigmas && rgbs && deltas));
    if (n_step <= 8u && n_alive <= (1u << 28))
        hipLaunchKernelGGL(k_composite_rays_g8, dim3(cdiv((unsigned long long)n_alive * 8u, kBlock)), dim3(kBlock), 0, stream, n_alive, n_step,
                           T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    else
        hipLaunchKernelGGL(k_composite_rays, dim3(cdiv(n_alive, kBlock)), dim3(kBlock), 0, stream, n_alive, n_step, T_thresh,
                           rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return nvsf_launch_status();
}
