// Stand-alone multiresolution hash-grid encoding (forward + table gradient) for gfx950.
// Replaces the tcnn.Encoding("HashGrid") modules the reference builds at hash_field.py:47-57 (2-D, per
// time slice), hash_field.py:109-119 (3-D static) and flow_field.py:70-80 (3-D, L16 F8).
//
// Work mapping (forward): a 256-thread workgroup encodes 64 consecutive samples.  Lanes of a wave are
// 64 consecutive samples (consecutive samples along a ray touch neighbouring cells, so the 64 gathers of
// one wave instruction fall into few cache lines); the four waves split the levels (wave w takes levels
// w, w+4, ...).  Features are staged as fp16 in LDS in output order and leave the workgroup as 16-byte
// stores of whole rows, instead of 4..16-byte stores at a stride of L*F*2 bytes.
#include "hashgrid_device.h"

namespace {
constexpr int kBlock = 256;
constexpr int kSamplesPerBlock = 64;

template <int D, int F>
__global__ __launch_bounds__(kBlock) void k_hashgrid_fwd(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                         uint32_t c1, uint32_t c2, const _Float16* __restrict__ table, uint32_t L,
                                                         GridMeta meta, _Float16* __restrict__ out, uint32_t out_stride) {
    // staging tile [64 samples][L*F/2 + 1] dwords (half2 granularity; the odd row stride keeps the
    // per-lane b32 writes and the row-major b32 read-back free of bank conflicts)
    extern __shared__ uint32_t stage[];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform by construction: level tables, pointers and loop control in SGPRs
    const uint32_t m = blockIdx.x * kSamplesPerBlock + lane;
    const uint32_t row_dw = L * F / 2, row_pitch = row_dw + 1;
    float xs[D];
    {
        const uint32_t mm = m < M ? m : M - 1;  // clamp: out-of-range lanes compute a valid sample and are dropped at the store
        const float* px = x + (size_t)mm * x_stride;
        xs[0] = px[c0];
        xs[1] = px[c1];
        if constexpr (D == 3) xs[2] = px[c2];
    }
    for (uint32_t l = wave; l < L; l += 4) {
        float acc[F];
        encode_level<D, F>(xs, table, meta.scale[l], meta.res[l], meta.offset[l], meta.offset[l + 1] - meta.offset[l], acc);
#pragma unroll
        for (int f = 0; f < F; f += 2) {
            h2_t p;
            p[0] = (_Float16)acc[f];
            p[1] = (_Float16)acc[f + 1];
            stage[lane * row_pitch + (l * F + f) / 2] = __builtin_bit_cast(uint32_t, p);
        }
    }
    __syncthreads();
    const uint32_t n_rows = min((uint32_t)kSamplesPerBlock, M - blockIdx.x * kSamplesPerBlock);
    uint32_t* out_dw = reinterpret_cast<uint32_t*>(out);
    for (uint32_t c = threadIdx.x; c < n_rows * row_dw; c += kBlock) {
        const uint32_t row = c / row_dw, col = c - row * row_dw;
        out_dw[(size_t)(blockIdx.x * kSamplesPerBlock + row) * (out_stride / 2) + col] = stage[row * row_pitch + col];
    }
}

// Forward of an all-hashed F = 4 grid with 8 levels (the static hash of the space-time field: T = 2^19, 512 -> 32768), ONE LEVEL
// PER XCD.  The generic kernel above walks the 8 levels of 64 samples inside one workgroup: every XCD's 4 MiB L2 sees all
// 32 MB of tables, nearly every corner is an L2 miss served by the Infinity Cache (measured 0.88-1.30 ms per 3.1 M samples).
// Here workgroup b works on level b % 8 for a share of ALL samples -- workgroups are dealt to the XCDs round-robin, so an XCD
// keeps gathering from the same 4 MB level, which stays in its L2.  Two lanes per sample (lane = 2 * sample + x-bit), as in the
// sliced pass of fused_field.hip: a gather instruction fetches both x-neighbours of 32 samples (the same 16-B pair for even
// cells).  The even lane blends features 0, 1, the odd lane features 2, 3, each over all eight corners in the specification's
// order (the partner's half of every entry arrives by a quad swap): the same fma chain as encode_level<3, 4>, bit-identical
// features.  If the dispatcher placed workgroups differently the result is the same, only slower.
__device__ __forceinline__ uint32_t lane_swap(uint32_t v) {  // value of lane ^ 1
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hashgrid_fwd_levels8(
    const float* __restrict__ x, uint32_t M, uint32_t x_stride, const _Float16* __restrict__ table, uint32_t table_bytes, GridMeta meta,
    _Float16* __restrict__ out, uint32_t out_stride, uint32_t planes = 0) {
    const uint32_t level = blockIdx.x & 7u, sb = blockIdx.x >> 3, n_sb = gridDim.x >> 3;
    const float scale = meta.scale[level];
    const uint32_t boff = meta.offset[level] * 8u, mask = meta.offset[level + 1] - meta.offset[level] - 1u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(table), 0, (int)table_bytes, 0x00020000);
    const int lane = lane_id();
    const uint32_t xb = (uint32_t)(lane & 1), half_lane = (uint32_t)(lane >> 1);
    const uint32_t n_units = (M + 31u) / 32u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(sb * (kBlock / kWave) + (threadIdx.x >> 6)), wave_count = n_sb * (kBlock / kWave);
    if (wave >= n_units) return;
    auto fetch = [&](uint32_t unit, float (&p)[3]) {
        const uint32_t s = unit * 32u + half_lane, sc = s < M ? s : M - 1u;
        const float* px = x + (size_t)sc * x_stride;
        p[0] = px[0];
        p[1] = px[1];
        p[2] = px[2];
    };
    float nxt[3];
    fetch(wave, nxt);
    for (uint32_t unit = wave; unit < n_units; unit += wave_count) {
        float xs[3] = {nxt[0], nxt[1], nxt[2]};
        fetch(unit + wave_count < n_units ? unit + wave_count : unit, nxt);
        float frac[3];
        uint32_t c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pos = fmaf(scale, xs[d], 0.5f);
            const float fl = floorf(pos);
            frac[d] = pos - fl;
            c[d] = (uint32_t)(int32_t)fl;
        }
        const uint32_t hy0 = c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = c[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
        u2v raw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) raw[p] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, boff + ((((c[0] + xb) ^ yz[p]) & mask) << 3), 0, 0);
        // this lane's feature pair of all eight corners: its own corners' dword, and the partner's (the corners with the other x-bit)
        uint32_t mine[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t own = xb ? raw[p][1] : raw[p][0], send = xb ? raw[p][0] : raw[p][1];
            const uint32_t recv = lane_swap(send);
            mine[2 * p] = xb ? recv : own;
            mine[2 * p + 1] = xb ? own : recv;
        }
        const float wx[2] = {1.0f - frac[0], frac[0]}, wy[2] = {1.0f - frac[1], frac[1]}, wz[2] = {1.0f - frac[2], frac[2]};
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float w = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
            const h2_t v = __builtin_bit_cast(h2_t, mine[k]);
            a0 = fmaf(w, (float)v[0], a0);
            a1 = fmaf(w, (float)v[1], a1);
        }
        h2_t o;
        o[0] = (_Float16)a0;
        o[1] = (_Float16)a1;
        const uint32_t packed = __builtin_bit_cast(uint32_t, o), other = lane_swap(packed);
        const uint32_t s = unit * 32u + half_lane;
        if (xb == 0u && s < M) {
            // rows [M, 32]: every XCD group writes 8 of the 64 bytes of every row (partial lines from eight L2s); level-major
            // [8][M][4] (`planes`): a wave instruction writes 256 contiguous bytes -- 0.78 -> 0.53 ms per 3.1 M samples
            if (planes) reinterpret_cast<uint2*>(out)[(size_t)level * M + s] = make_uint2(packed, other);
            else *reinterpret_cast<uint2*>(out + (size_t)s * out_stride + 4u * level) = make_uint2(packed, other);
        }
    }
}

// Table gradient: one thread per (sample, level); fp32 atomics into grad_table.
template <int D, int F, bool GRAD_F16>
__global__ __launch_bounds__(kBlock) void k_hashgrid_bwd(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                         uint32_t c1, uint32_t c2, uint32_t L, GridMeta meta,
                                                         const void* __restrict__ grad_out, uint32_t go_stride, uint32_t go_level,
                                                         float* __restrict__ grad_table) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t l = blockIdx.y;
    if (m >= M) return;
    float xs[D];
    const float* px = x + (size_t)m * x_stride;
    xs[0] = px[c0];
    xs[1] = px[c1];
    if constexpr (D == 3) xs[2] = px[c2];
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if constexpr (GRAD_F16) g[f] = (float)reinterpret_cast<const _Float16*>(grad_out)[(size_t)m * go_stride + (size_t)l * go_level + f];
        else g[f] = reinterpret_cast<const float*>(grad_out)[(size_t)m * go_stride + (size_t)l * go_level + f];
    }
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; ++f) any |= (g[f] != 0.0f);
    if (!any) return;
    const float scale = meta.scale[l];
    const uint32_t res = meta.res[l], row0 = meta.offset[l], hsize = meta.offset[l + 1] - row0;
    float frac[D];
    uint32_t cell[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pos = fmaf(scale, xs[d], 0.5f);
        const float fl = floorf(pos);
        frac[d] = pos - fl;
        cell[d] = (uint32_t)(int32_t)fl;
    }
#pragma unroll
    for (int c = 0; c < (1 << D); ++c) {
        uint32_t cc[D];
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (c & (1 << d)) { w = w * frac[d]; cc[d] = cell[d] + 1u; }
            else { w = w * (1.0f - frac[d]); cc[d] = cell[d]; }
        }
        float* dst = grad_table + ((size_t)row0 + grid_row<D>(cc, res, hsize)) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(dst + f, w * g[f]);
    }
}

// Table gradient, corner-parallel run merging.  The atomic adds are what bounds this pass once the gradients are dense
// (805 M fp32 adds per 3.1 M-sample batch at L16 F2), and MI355X executes float atomics at the memory side at a cost
// per wave-instruction and 64-B segment (MI355X_MICROARCH.md, Global float atomics): a lane that walks the 2^D x F
// floats of a cell one after the other makes 2^D x F instructions whose 64 lanes sit in 64
// unrelated rows, the slowest shape there is.  Here the 2^D x F floats of one (chunk of rows, level) item are spread over
// G = 2^D x F adjacent lanes (lane = corner x F + feature), 64 / G items (consecutive levels of one chunk) per wave:
// every lane keeps ONE running sum, the cell changes for all lanes of an item at once, and a flush is ONE atomic
// instruction whose lanes cover whole table entries (F contiguous floats; the two x-neighbours of a corner pair are
// adjacent entries on dense levels and for even cells on hashed ones).  Same sums as k_hashgrid_bwd up to the order of the
// fp32 additions.
template <int D, int F, bool GRAD_F16>
__global__ __launch_bounds__(kBlock) void k_hashgrid_bwd_corners(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0,
                                                                 uint32_t c1, uint32_t c2, uint32_t L, GridMeta meta,
                                                                 const void* __restrict__ grad_out, uint32_t go_stride, uint32_t go_level,
                                                                 float* __restrict__ grad_table, uint32_t run, uint32_t l_end) {
    auto flush = [&](float* dst, float acc) { atomicAdd(dst, acc); };
    constexpr int G = (1 << D) * F;  // lanes per item
    constexpr int IPW = kWave / G;   // items per wave
    constexpr int B = 8;             // rows requested together: a walk with ONE row in flight lasts a memory round trip per row
    static_assert(G <= kWave && kWave % G == 0, "2^D x F must divide the wave");
    const int lane = lane_id();
    const int sub = lane / G, r = lane - sub * G, c = r / F, f = r - c * F;
    // the IPW items of a wave are consecutive levels of ONE chunk of rows (host: L % IPW == 0): the rows are wave-uniform,
    // their positions come through the scalar cache, only the gradient element differs per lane
    const unsigned long long item0 = ((unsigned long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * IPW;
    const uint32_t chunk = __builtin_amdgcn_readfirstlane((uint32_t)(item0 / L));
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(item0 - (unsigned long long)chunk * L)) + (uint32_t)sub;
    const unsigned long long first = (unsigned long long)chunk * run;
    if (first >= M) return;
    const uint32_t m0 = (uint32_t)first, m1 = (uint32_t)(first + run < M ? first + run : M);
    // levels l_end ... are scattered by the binned kernels below: their items idle (zero gradient = skipped row), a wave all of
    // whose items are such levels leaves
    const bool live = l < l_end;
    if (l - (uint32_t)sub >= l_end) return;  // uniform
    const float scale = meta.scale[l];
    const uint32_t res = meta.res[l], row0 = meta.offset[l], hsize = meta.offset[l + 1] - row0;
    const size_t gcol = (size_t)l * go_level + (uint32_t)f;  // go_level = F: rows [M, L F]; = M F: level-major [L][M][F]
    float acc = 0.0f;
    uint32_t cur[D];
    bool have = false;
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = 0u;
    float* dst = grad_table;
    struct Rows {
        float xs[B][D], g[B];
    };
    auto load_rows = [&](uint32_t m, Rows& rw) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t mj = m + (uint32_t)j < m1 ? m + (uint32_t)j : m1 - 1u;  // uniform; rows past the end are not consumed
            const float* px = x + (size_t)mj * x_stride;
            rw.xs[j][0] = px[c0];
            rw.xs[j][1] = px[c1];
            if constexpr (D == 3) rw.xs[j][2] = px[c2];
            if (!live) rw.g[j] = 0.0f;
            else if constexpr (GRAD_F16) rw.g[j] = (float)reinterpret_cast<const _Float16*>(grad_out)[(size_t)mj * go_stride + gcol];
            else rw.g[j] = reinterpret_cast<const float*>(grad_out)[(size_t)mj * go_stride + gcol];
        }
    };
    Rows nxt;
    load_rows(m0, nxt);
    for (uint32_t m = m0; m < m1; m += B) {
        const Rows rw = nxt;
        if (m + B < m1) load_rows(m + B, nxt);  // next batch in flight while this one is walked
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (m + (uint32_t)j >= m1) break;  // uniform
            const float g = rw.g[j];
            // a row all of whose features have a zero gradient at this level is skipped (the run is not broken); otherwise the
            // whole item moves with the row, lanes of a zero feature add 0
            const unsigned long long nz = __ballot(g != 0.0f);
            const unsigned long long item_bits = G == 64 ? ~0ull : ((1ull << G) - 1ull);
            if (((nz >> (sub * G)) & item_bits) != 0ull) {
                float w = 1.0f;
                uint32_t cell[D];
                bool same = have;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float pos = fmaf(scale, rw.xs[j][d], 0.5f);
                    const float fl = floorf(pos);
                    const float frac = pos - fl;
                    cell[d] = (uint32_t)(int32_t)fl;
                    same = same && (cell[d] == cur[d]);
                    w = w * ((c & (1 << d)) ? frac : (1.0f - frac));
                }
                if (!same) {
                    // A step into a neighbouring cell keeps the grid vertices the two cells share (4 of 8 across a face, 2 across
                    // an edge): their running sums move to the lanes that hold those vertices in the new cell instead of going
                    // to memory -- the scatter is bound by the number of 64-byte atomic pieces, and on the levels whose cells are
                    // a few steps long this is where half of them came from.  `take`: this lane's new vertex was vertex `src_c`
                    // of the old cell; `kept`: this lane's old vertex is a vertex of the new cell (some lane takes its sum).
                    bool take = have, kept = have;
                    int src_c = 0;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const int dl = (int)(cell[d] - cur[d]);
                        const int b = (c >> d) & 1;
                        const int sb = b + dl, tb = b - dl;
                        take = take && (sb == 0 || sb == 1);
                        kept = kept && (tb == 0 || tb == 1);
                        src_c |= (sb & 1) << d;
                    }
                    const float moved = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane - r) + src_c * F + f) << 2, __builtin_bit_cast(int, acc)));
                    if (!kept && acc != 0.0f) flush(dst, acc);
                    acc = take ? moved : 0.0f;
                    have = true;
                    uint32_t cc[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        cur[d] = cell[d];
                        cc[d] = cell[d] + ((c >> d) & 1u);
                    }
                    dst = grad_table + ((size_t)row0 + grid_row<D>(cc, res, hsize)) * F + f;
                }
                acc += w * g;
            }
        }
    }
    if (acc != 0.0f) flush(dst, acc);
}

// ---- Binned scatter of the FINE levels ----------------------------------------------------------------------------------------------
// On the levels whose cells are shorter than a ray step every sample sits in a cell of its own: run merging has nothing to merge and
// the corner-parallel kernel above pays one memory-side atomic (a read-modify-write of a 64-byte segment, ~41 G/s chip-wide) per
// corner pair -- 93 M of the 110 M atomic pieces of a config-2 camera batch.  These levels take two streaming passes instead:
//   k_hashgrid_bwd_bin     (sample tiles): every (sample, corner) contribution {row, w_c * g[0..F)} of a level is appended to the bin
//                          its table row falls into (bin = kBinRows consecutive rows of one level).  A workgroup sorts a tile's pairs by
//                          bin in LDS (LDS counters -> ranks -> exclusive scan), reserves its slice of every bin with ONE returning
//                          atomic per bin and copies the sorted pairs out as contiguous runs; it also records max |g| of the level;
//   k_hashgrid_bwd_reduce  (bins): a workgroup adds the pairs of its bin into an LDS image of the bin and adds the image to the table
//                          gradient with contiguous atomics (512 pieces per bin instead of one per pair).  The image is 64-bit FIXED
//                          POINT (value x 2^(33 - exponent of the level's max |g|), ds_add_u64): LDS float atomics run at a quarter
//                          of the rate of the integer ones on gfx950 (measured: 1.6 ms against 0.38 ms for the five fine levels of a
//                          camera batch), and integer sums do not depend on the order of the pairs.  A contribution (a run of up to 8 rows,
//                          |w g| < 2^(e+1) each) scales below 2^37, a row can take 2^26 of them before the sum leaves 63 bits (the host
//                          limits M x 8 to that); the resolution 2^(e-33) is 2^-10 ulp of the largest gradient.
// A bin holds `cap` pairs (1.25 x the mean of a uniform hash + slack); what does not fit is added to the table directly, so the result is
// the same sum for any input (only slower when the hash is badly skewed).  Same addends as the other kernels, another order.
template <int F> struct BinCfg {
    static constexpr int kTile = F <= 2 ? 512 : 256;      // samples per workgroup tile
    static constexpr int kPairs = kTile * 8;              // (sample, corner) pairs of a tile and level: 4096 / 2048
    static constexpr int kBinRows = 8192 / F;             // rows per bin: 64 KB of 64-bit sums
    static constexpr int kBinShift = F <= 2 ? 12 : 11;
    static constexpr int kMaxBins = 256;                  // rows per level <= 256 bins (checked on the host)
};

// The binned levels of one launch: level, bins, capacity of a bin, where its cursors / pairs / reduce workgroups start.  `merged`: the
// level's contributions are run sums -- up to 8 consecutive rows that fall into one cell add their w_c g in registers first (a
// segmented scan over the lanes of the wave, whose lanes are consecutive rows) and the LAST row of the run emits the eight sums.  On
// the levels whose cells are several ray steps long this leaves a fraction of the contributions (and of the memory-side atomics the
// corner-parallel kernel pays for them).
struct BinLevels {
    uint32_t n;
    uint32_t level[16], nbins[16], bin0[16], cap[16], split[16], wg0[17], merged[16];
    unsigned long long pair0[16];
};

template <int F, bool GRAD_F16>
__global__ __launch_bounds__(kBlock) void k_hashgrid_bwd_bin(const float* __restrict__ x, uint32_t M, uint32_t x_stride, uint32_t c0, uint32_t c1,
                                                             uint32_t c2, GridMeta meta, BinLevels bl, const void* __restrict__ grad_out,
                                                             uint32_t go_stride, uint32_t go_level, float* __restrict__ grad_table, uint32_t* __restrict__ cursors,
                                                             uint32_t* __restrict__ level_max, uint16_t* __restrict__ pair_row,
                                                             float* __restrict__ pair_val) {
    using C = BinCfg<F>;
    constexpr int SPT = C::kTile / kBlock;
    __shared__ uint32_t s_cnt[C::kMaxBins], s_pre[C::kMaxBins], s_gbase[C::kMaxBins], s_total, s_max;
    __shared__ uint32_t s_row[C::kPairs];
    __shared__ float s_val[C::kPairs * F];
    const int tid = (int)threadIdx.x, lane = lane_id();
    float xs[SPT][3];
    uint32_t mrow[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        mrow[s] = blockIdx.x * C::kTile + s * kBlock + tid;  // lanes of a wave = consecutive rows
        const uint32_t mm = mrow[s] < M ? mrow[s] : M - 1;
        const float* px = x + (size_t)mm * x_stride;
        xs[s][0] = px[c0];
        xs[s][1] = px[c1];
        xs[s][2] = px[c2];
    }
    for (uint32_t j = 0; j < bl.n; ++j) {
        const uint32_t l = bl.level[j], nbins = bl.nbins[j], cap = bl.cap[j], bin0 = bl.bin0[j];
        const bool merged = bl.merged[j] != 0u;  // uniform
        const float scale = meta.scale[l];
        const uint32_t res = meta.res[l], row0 = meta.offset[l], hsize = meta.offset[l + 1] - row0;
        const bool lvl_hashed = (unsigned long long)res * res * res > hsize;  // uniform
        for (int b = tid; b < C::kMaxBins; b += kBlock) s_cnt[b] = 0u;
        if (tid == 0) s_max = 0u;
        __syncthreads();
        float val[SPT][8][F];   // contributions of this lane's row (or of the run that ends with it) to the eight vertices of its cell
        uint32_t pk[SPT][8];
        float gmax = 0.0f;
        bool bad = false;
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            float g[F];
            bool any = false;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const size_t at = (size_t)(mrow[s] < M ? mrow[s] : 0u) * go_stride + (size_t)l * go_level + f;
                if constexpr (GRAD_F16) g[f] = (float)reinterpret_cast<const _Float16*>(grad_out)[at];
                else g[f] = reinterpret_cast<const float*>(grad_out)[at];
                if (mrow[s] >= M) g[f] = 0.0f;
                // an inf / NaN gradient (an fp16 overflow under GradScaler) must REACH the table -- the memory-atomic path adds it
                // there and the scaler's found_inf check skips the step -- but it must not enter the fixed-point image: fmaxf drops
                // a NaN, an inf would set the level's exponent to 128 and truncate every finite addend to zero.  The
                // value is taken out of the sums here and the level's first table entry is made NaN below.
                if (!(fabsf(g[f]) <= 3.402823466e38f)) { bad = true; g[f] = 0.0f; }
                any = any || g[f] != 0.0f;
                gmax = fmaxf(gmax, fabsf(g[f]));
            }
            uint32_t cell[3];
            float frac[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale, xs[s][d], 0.5f);
                const float fl = floorf(pos);
                frac[d] = pos - fl;
                cell[d] = (uint32_t)(int32_t)fl;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = ((1.0f * ((c & 1) ? frac[0] : 1.0f - frac[0])) * ((c & 2) ? frac[1] : 1.0f - frac[1])) *
                                ((c & 4) ? frac[2] : 1.0f - frac[2]);
#pragma unroll
                for (int f = 0; f < F; ++f) val[s][c][f] = w * g[f];
            }
            bool emit = any;
            if (merged) {
                // runs: consecutive lanes in one cell, cut every 8 lanes (a run never crosses a DPP row of 16, three scan steps reach
                // its first lane).  head = first lane of a run, dist = lanes since the head, tail = last lane of a run.
                bool head = (lane & 7) == 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cell[d], 0x111 /* row_shr:1 */, 0xF, 0xF, false);
                    head = head || prev != cell[d];
                }
                const unsigned long long hm = __ballot(head);
                const unsigned long long below = hm & ((2ull << lane) - 1ull);
                const int dist = lane - (63 - __builtin_clzll(below));
                const bool tail = (lane & 7) == 7 || ((hm >> ((lane + 1) & 63)) & 1ull) != 0ull;
                bool run_any = false;
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        float v = val[s][c][f];
                        {
                            const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
                            if (dist >= 1) v += t;
                        }
                        {
                            const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, false));
                            if (dist >= 2) v += t;
                        }
                        {
                            const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, false));
                            if (dist >= 4) v += t;
                        }
                        val[s][c][f] = v;
                        run_any = run_any || v != 0.0f;
                    }
                emit = tail && run_any;
            }
            // rows of the eight vertices (grid_row<3> of hashgrid_device.h with the level's kind decided once: per-axis terms of the
            // cell and of its upper neighbour, combined by xor + mask on a hashed level, by sums + one wrap on a dense one)
            uint32_t tx[2], ty[2], tz[2];
            tx[0] = cell[0];
            tx[1] = cell[0] + 1u;
            if (lvl_hashed) {
                ty[0] = cell[1] * 2654435761u;
                ty[1] = ty[0] + 2654435761u;
                tz[0] = cell[2] * 805459861u;
                tz[1] = tz[0] + 805459861u;
            } else {
                ty[0] = cell[1] * res;
                ty[1] = ty[0] + res;
                tz[0] = cell[2] * res * res;
                tz[1] = tz[0] + res * res;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pk[s][c] = 0xFFFFFFFFu;
                if (emit) {
                    uint32_t row;
                    if (lvl_hashed) {
                        row = (tx[c & 1] ^ ty[(c >> 1) & 1] ^ tz[c >> 2]) & (hsize - 1u);  // hashed levels are powers of two (host)
                    } else {
                        row = tx[c & 1] + ty[(c >> 1) & 1] + tz[c >> 2];  // < 2 hsize for a position inside the unit cube
                        if (row >= hsize) {
                            row -= hsize;
                            if (row >= hsize) row %= hsize;  // a position outside [0, 1]^3 (any cell index): grid_row's general modulo
                        }
                    }
                    const uint32_t rank = atomicAdd(&s_cnt[row >> C::kBinShift], 1u);
                    pk[s][c] = row | (rank << 20);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        if (lane == 0 && gmax > 0.0f) atomicMax(&s_max, __float_as_uint(gmax));  // non-negative floats order like their bit patterns
        if (__ballot(bad) != 0ull && lane == 0) atomicAdd(grad_table + (size_t)row0 * F, __builtin_nanf(""));
        __syncthreads();
        if (tid < kWave) {  // exclusive scan of the bin counts, one reservation per bin
            uint32_t run = 0;
            for (uint32_t b0 = 0; b0 < nbins; b0 += kWave) {
                const uint32_t b = b0 + (uint32_t)lane;
                const uint32_t cnt = b < nbins ? s_cnt[b] : 0u;
                uint32_t incl = cnt;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                    if (lane >= o) incl += up;
                }
                if (b < nbins) {
                    s_pre[b] = run + incl - cnt;
                    s_gbase[b] = (cnt ? atomicAdd(&cursors[bin0 + b], cnt) : 0u) - (run + incl - cnt);  // global slot of staged pair p: p + s_gbase[bin] (mod 2^32)
                }
                run += (uint32_t)__shfl((int)incl, kWave - 1);
            }
            if (lane == 0) {
                s_total = run;
                if (s_max) atomicMax(&level_max[j], s_max);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPT; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (pk[s][c] == 0xFFFFFFFFu) continue;
                const uint32_t row = pk[s][c] & 0xFFFFFu, rank = pk[s][c] >> 20;
                const uint32_t at = s_pre[row >> C::kBinShift] + rank;
                s_row[at] = row;
#pragma unroll
                for (int f = 0; f < F; ++f) s_val[at * F + f] = val[s][c][f];
            }
        __syncthreads();
        const uint32_t total = s_total;
        for (uint32_t p = (uint32_t)tid; p < total; p += kBlock) {
            const uint32_t row = s_row[p], b = row >> C::kBinShift;
            const uint32_t q = p + s_gbase[b];
            if (q < cap) {
                const size_t at = (size_t)bl.pair0[j] + (size_t)b * cap + q;
                pair_row[at] = (uint16_t)(row & (uint32_t)(C::kBinRows - 1));  // the row inside its bin: 12 / 11 bits
                if constexpr (F == 2) {
                    *reinterpret_cast<float2*>(pair_val + at * 2) = make_float2(s_val[p * 2], s_val[p * 2 + 1]);
                } else {
                    *reinterpret_cast<float4*>(pair_val + at * 4) = make_float4(s_val[p * 4], s_val[p * 4 + 1], s_val[p * 4 + 2], s_val[p * 4 + 3]);
                }
            } else {  // the bin is full: add straight to the table
#pragma unroll
                for (int f = 0; f < F; ++f)
                    if (s_val[p * F + f] != 0.0f) atomicAdd(grad_table + ((size_t)row0 + row) * F + f, s_val[p * F + f]);
            }
        }
        __syncthreads();
    }
}

constexpr int kReduceBlock = 512;
constexpr int kFixedTop = 33;  // the level's largest |g| has its leading bit at 2^33 in the fixed-point image (a run sum reaches 2^37)
template <int F>
__global__ __launch_bounds__(kReduceBlock) void k_hashgrid_bwd_reduce(const uint32_t* __restrict__ cursors, const uint32_t* __restrict__ level_max,
                                                                      const uint16_t* __restrict__ pair_row, const float* __restrict__ pair_val,
                                                                      BinLevels bl, GridMeta meta, float* __restrict__ grad_table) {
    using C = BinCfg<F>;
    __shared__ unsigned long long s_acc[C::kBinRows * F];
    uint32_t j = 0;
    while (j + 1 < bl.n && blockIdx.x >= bl.wg0[j + 1]) ++j;  // uniform
    const uint32_t local = blockIdx.x - bl.wg0[j], split = bl.split[j], cap = bl.cap[j];
    const uint32_t b = local / split, part = local - b * split;
    const uint32_t l = bl.level[j], rows_l = meta.offset[l + 1] - meta.offset[l];
    uint32_t n = cursors[bl.bin0[j] + b];
    n = n < cap ? n : cap;
    const uint32_t q0 = (uint32_t)((unsigned long long)n * part / split), q1 = (uint32_t)((unsigned long long)n * (part + 1) / split);
    if (q0 == q1) return;  // uniform
    // power-of-two scale: the product is exact, the only rounding is the truncation to an integer
    const int e = (int)((level_max[j] >> 23) & 0xFFu) - 127;
    const float to_fixed = __builtin_ldexpf(1.0f, kFixedTop - e);
    for (int i = (int)threadIdx.x; i < C::kBinRows * F; i += kReduceBlock) s_acc[i] = 0ull;
    __syncthreads();
    const size_t base = (size_t)bl.pair0[j] + (size_t)b * cap;
    const uint16_t* pr = pair_row + base;
    const float* pv = pair_val + base * F;
    constexpr int U = 4;
    for (uint32_t q = q0 + threadIdx.x; q < q1; q += kReduceBlock * U) {
        uint32_t row[U];
        float v[U][F];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t qq = q + (uint32_t)u * kReduceBlock;
            row[u] = qq < q1 ? (uint32_t)pr[qq] : 0xFFFFFFFFu;
            if (qq < q1) {
                if constexpr (F == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(pv + (size_t)qq * 2);
                    v[u][0] = t.x, v[u][1] = t.y;
                } else {
                    const float4 t = *reinterpret_cast<const float4*>(pv + (size_t)qq * 4);
                    v[u][0] = t.x, v[u][1] = t.y, v[u][2] = t.z, v[u][3] = t.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (row[u] == 0xFFFFFFFFu) continue;
            const uint32_t at = (row[u] & (uint32_t)(C::kBinRows - 1)) * F;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const long long fx = (long long)(v[u][f] * to_fixed);
                if (fx != 0) atomicAdd(&s_acc[at + f], (unsigned long long)fx);
            }
        }
    }
    __syncthreads();
    const double from_fixed = (double)__builtin_ldexpf(1.0f, e - kFixedTop);
    const uint32_t first_row = b * (uint32_t)C::kBinRows;
    const uint32_t n_rows = rows_l - first_row < (uint32_t)C::kBinRows ? rows_l - first_row : (uint32_t)C::kBinRows;  // the last bin of a dense level is short
    float* dst = grad_table + ((size_t)meta.offset[l] + first_row) * F;
    for (uint32_t i = threadIdx.x; i < n_rows * F; i += kReduceBlock) {
        const long long sum = (long long)s_acc[i];
        if (sum != 0) atomicAdd(dst + i, (float)((double)sum * from_fixed));
    }
}

int fill_meta(GridMeta& meta, uint32_t L, const float* scales, const uint32_t* res, const uint32_t* offsets) {
    if (L == 0 || L > (uint32_t)kMaxLevels || !scales || !res || !offsets) return NVSF_ERR_INVALID_ARG;
    for (uint32_t l = 0; l < L; ++l) {
        meta.scale[l] = scales[l];
        meta.res[l] = res[l];
        meta.offset[l] = offsets[l];
        if (offsets[l + 1] <= offsets[l] || res[l] == 0) return NVSF_ERR_INVALID_ARG;
    }
    meta.offset[L] = offsets[L];
    return NVSF_OK;
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

#define DISPATCH_DF(D, F, CALL)                                         \
    do {                                                                \
        if (D == 2 && F == 2) { CALL(2, 2); }                           \
        else if (D == 2 && F == 4) { CALL(2, 4); }                      \
        else if (D == 2 && F == 8) { CALL(2, 8); }                      \
        else if (D == 3 && F == 2) { CALL(3, 2); }                      \
        else if (D == 3 && F == 4) { CALL(3, 4); }                      \
        else if (D == 3 && F == 8) { CALL(3, 8); }                      \
        else return NVSF_ERR_UNSUPPORTED;                               \
    } while (0)

NVSF_API int nvsf_hashgrid_fwd(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, const void* table_f16,
                               uint32_t L, uint32_t F, const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets,
                               void* out_f16, uint32_t out_stride, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && cols && table_f16 && out_f16 && (D == 2 || D == 3));
    REQUIRE(out_stride >= L * F && out_stride % 2 == 0);
    REQUIRE((reinterpret_cast<uintptr_t>(out_f16) & 3u) == 0 && (reinterpret_cast<uintptr_t>(table_f16) & 15u) == 0);
    for (uint32_t d = 0; d < D; ++d) REQUIRE(cols[d] < x_stride);
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    const uint32_t c0 = cols[0], c1 = cols[1], c2 = D == 3 ? cols[2] : 0;
    {   // eight hashed levels of F = 4 on a batch large enough to fill the chip: one level per XCD (k_hashgrid_fwd_levels8)
        bool sliced = D == 3 && F == 4 && L == 8 && c0 == 0 && c1 == 1 && c2 == 2 && M >= (1u << 16) && out_stride % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(out_f16) & 7u) == 0 && (unsigned long long)h_offsets[L] * 8ull < (1ull << 31);
        for (uint32_t l = 0; l < L && sliced; ++l) {
            const unsigned long long cells = (unsigned long long)h_res[l] * h_res[l] * h_res[l];
            const uint32_t rows = h_offsets[l + 1] - h_offsets[l];
            sliced = cells > rows && (rows & (rows - 1u)) == 0u;  // hashed, power-of-two table
        }
        if (sliced && nvsf_variant(kVarHashgridFwd) == 0) {  // 1 (tests): the one-workgroup-per-64-samples kernel, the reference form
            const uint32_t units = (M + 31u) / 32u;
            uint32_t ps = (units + kBlock / kWave - 1) / (kBlock / kWave);
            if (ps > 512u) ps = 512u;
            hipLaunchKernelGGL(k_hashgrid_fwd_levels8, dim3(8u * ps), dim3(kBlock), 0, stream, x, M, x_stride,
                               reinterpret_cast<const _Float16*>(table_f16), (uint32_t)(h_offsets[L] * 8u), meta,
                               reinterpret_cast<_Float16*>(out_f16), out_stride, 0u);
            return nvsf_launch_status();
        }
    }
    const size_t lds = (size_t)kSamplesPerBlock * (L * F / 2 + 1) * sizeof(uint32_t);
#define CALL(DD, FF)                                                                                                        \
    hipLaunchKernelGGL((k_hashgrid_fwd<DD, FF>), dim3(cdiv(M, kSamplesPerBlock)), dim3(kBlock), lds, stream, x, M, x_stride, c0, \
                       c1, c2, reinterpret_cast<const _Float16*>(table_f16), L, meta, reinterpret_cast<_Float16*>(out_f16),  \
                       out_stride)
    DISPATCH_DF(D, F, CALL);
#undef CALL
    return nvsf_launch_status();
}

// The same encoder with the output LEVEL-MAJOR, fp16 [L][M][F]: what a consumer that reads whole levels per lane wants
// (nvsf_density_dynamic_lm_fwd) and what the one-level-per-XCD kernel writes best (contiguous 8-byte pieces instead of eight
// partial writes into every 64-byte row).  Built for the grid that kernel is built for: D = 3, F = 4, L = 8, every level hashed
// into a power-of-two table, columns 0..2 of x; NVSF_ERR_UNSUPPORTED otherwise (the caller falls back to rows).
NVSF_API int nvsf_hashgrid_fwd_level_major(const float* x, uint32_t M, uint32_t x_stride, const void* table_f16, uint32_t L, uint32_t F,
                                           const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, void* out_f16,
                                           hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && table_f16 && out_f16 && x_stride >= 3);
    REQUIRE((reinterpret_cast<uintptr_t>(out_f16) & 7u) == 0 && (reinterpret_cast<uintptr_t>(table_f16) & 15u) == 0);
    if (!(F == 4 && L == 8)) return NVSF_ERR_UNSUPPORTED;
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    if ((unsigned long long)h_offsets[L] * 8ull >= (1ull << 31)) return NVSF_ERR_UNSUPPORTED;
    for (uint32_t l = 0; l < L; ++l) {
        const unsigned long long cells = (unsigned long long)h_res[l] * h_res[l] * h_res[l];
        const uint32_t rows = h_offsets[l + 1] - h_offsets[l];
        if (!(cells > rows && (rows & (rows - 1u)) == 0u)) return NVSF_ERR_UNSUPPORTED;
    }
    const uint32_t units = (M + 31u) / 32u;
    uint32_t ps = (units + kBlock / kWave - 1) / (kBlock / kWave);
    if (ps > 512u) ps = 512u;
    hipLaunchKernelGGL(k_hashgrid_fwd_levels8, dim3(8u * ps), dim3(kBlock), 0, stream, x, M, x_stride,
                       reinterpret_cast<const _Float16*>(table_f16), (uint32_t)(h_offsets[L] * 8u), meta,
                       reinterpret_cast<_Float16*>(out_f16), 0u, 1u);
    return nvsf_launch_status();
}

namespace {
struct BinPlan {
    BinLevels bl;
    uint32_t total_bins, total_wgs;
    size_t cursors_bytes, rows_bytes, vals_bytes;
};
// Levels merge_from ... L-1 go through the bins: [merge_from, fine_from) as run sums, [fine_from, L) one contribution per (row, vertex).
// Workspace: [cursors: one u32 per bin, max |g| bits: 16 u32 | rows of the contributions, u16 | their values, F fp32 each].
bool bin_plan(uint32_t M, uint32_t L, uint32_t F, const uint32_t* h_res, const uint32_t* h_offsets, uint32_t merge_from, uint32_t fine_from,
              BinPlan& bp) {
    if ((F != 2 && F != 4) || merge_from > fine_from || fine_from > L || merge_from == L || L - merge_from > 16u) return false;
    if ((unsigned long long)M * 8ull > (1ull << 26)) return false;  // 2^26 contributions per row at most (fixed-point headroom)
    const uint32_t bin_rows = 8192u / F;
    BinLevels& bl = bp.bl;
    bl.n = L - merge_from;
    unsigned long long pairs = 0;
    uint32_t bins = 0, wgs = 0;
    for (uint32_t j = 0; j < bl.n; ++j) {
        const uint32_t l = merge_from + j, rows = h_offsets[l + 1] - h_offsets[l];
        const bool hashed = (unsigned long long)h_res[l] * h_res[l] * h_res[l] > rows;
        if (rows > (1u << 20) || (hashed && (rows & (rows - 1u)) != 0u)) return false;
        const uint32_t nbins = (rows + bin_rows - 1) / bin_rows;
        if (nbins > 256u) return false;
        const unsigned long long all = (unsigned long long)M * 8ull, mean = (all + nbins - 1) / nbins;
        unsigned long long cap;
        if (l >= fine_from) cap = mean + mean / 4 + 2048ull;          // uniform hash, one contribution per (row, vertex)
        else if (hashed) cap = mean + 4096ull;                        // run sums: never more than that
        else cap = (mean < all / 2 ? mean : all / 2) + 4096ull;       // dense level: bins are slabs of space (skewed): the unmerged uniform share
        cap = (cap + 63ull) & ~63ull;
        bl.level[j] = l;
        bl.nbins[j] = nbins;
        bl.bin0[j] = bins;
        bl.cap[j] = (uint32_t)cap;
        bl.merged[j] = l < fine_from ? 1u : 0u;
        bl.pair0[j] = pairs;
        uint32_t split = (uint32_t)((cap + 98303ull) / 98304ull);
        bl.split[j] = split < 1u ? 1u : (split > 128u ? 128u : split);
        if (l >= fine_from && bl.split[j] < 4u) bl.split[j] = 4u;
        bl.wg0[j] = wgs;
        bins += nbins;
        wgs += nbins * bl.split[j];
        pairs += cap * nbins;
        if (cap >= (1ull << 31) || pairs >= (1ull << 40)) return false;
    }
    bl.wg0[bl.n] = wgs;
    bp.total_bins = bins;
    bp.total_wgs = wgs;
    bp.cursors_bytes = (((size_t)(bins + 16u) * sizeof(uint32_t)) + 255u) & ~(size_t)255u;  // cursors + max |g| per level
    bp.rows_bytes = (((size_t)pairs * sizeof(uint16_t)) + 255u) & ~(size_t)255u;
    bp.vals_bytes = (size_t)pairs * F * sizeof(float);
    return true;
}

int hashgrid_bwd_launch(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L, uint32_t F,
                        const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, const void* grad_out, int grad_is_f16,
                        uint32_t go_stride, uint32_t go_level_stride, float* grad_table_f32, uint32_t merge_from, uint32_t fine_from,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M == 0) return NVSF_OK;
    REQUIRE(x && cols && grad_out && grad_table_f32 && (D == 2 || D == 3) && merge_from <= fine_from && fine_from <= L);
    // rows [M, go_stride >= L F] (go_level_stride = 0), or level-major [L][M][F]: go_level_stride elements between levels, go_stride = F
    REQUIRE(go_level_stride ? (go_stride >= F && (unsigned long long)go_level_stride >= (unsigned long long)M * go_stride) : go_stride >= L * F);
    const uint32_t go_level = go_level_stride ? go_level_stride : F;
    for (uint32_t d = 0; d < D; ++d) REQUIRE(cols[d] < x_stride);
    GridMeta meta;
    const int st = fill_meta(meta, L, h_scales, h_res, h_offsets);
    if (st != NVSF_OK) return st;
    const uint32_t c0 = cols[0], c1 = cols[1], c2 = D == 3 ? cols[2] : 0;
    if (nvsf_variant(kVarHashgridBwd) != 0) merge_from = fine_from = L;  // tests: every level through the plain kernel
    if (merge_from < L) {  // levels merge_from ... L-1 through the bins
        REQUIRE(D == 3 && workspace && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0);
        for (uint32_t l = fine_from; l < L; ++l)
            REQUIRE((unsigned long long)h_res[l] * h_res[l] * h_res[l] > h_offsets[l + 1] - h_offsets[l]);  // per-row contributions: hashed levels
        BinPlan bp;
        REQUIRE(bin_plan(M, L, F, h_res, h_offsets, merge_from, fine_from, bp));
        REQUIRE(workspace_bytes >= bp.cursors_bytes + bp.rows_bytes + bp.vals_bytes);
        uint32_t* cursors = reinterpret_cast<uint32_t*>(workspace);
        uint16_t* pair_row = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + bp.cursors_bytes);
        float* pair_val = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + bp.cursors_bytes + bp.rows_bytes);
        {
            const hipError_t e = hipMemsetAsync(cursors, 0, bp.cursors_bytes, stream);
            if (e != hipSuccess) return (int)e;
        }
        uint32_t* level_max = cursors + bp.total_bins;
#define CALLB(FF)                                                                                                                      \
    do {                                                                                                                               \
        const dim3 bgrid(cdiv(M, (uint32_t)BinCfg<FF>::kTile));                                                                        \
        if (grad_is_f16)                                                                                                               \
            hipLaunchKernelGGL((k_hashgrid_bwd_bin<FF, true>), bgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, meta,       \
                               bp.bl, grad_out, go_stride, go_level, grad_table_f32, cursors, level_max, pair_row, pair_val);                    \
        else                                                                                                                           \
            hipLaunchKernelGGL((k_hashgrid_bwd_bin<FF, false>), bgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, meta,      \
                               bp.bl, grad_out, go_stride, go_level, grad_table_f32, cursors, level_max, pair_row, pair_val);                    \
        hipLaunchKernelGGL((k_hashgrid_bwd_reduce<FF>), dim3(bp.total_wgs), dim3(kReduceBlock), 0, stream, cursors, level_max,         \
                           pair_row, pair_val, bp.bl, meta, grad_table_f32);                                                           \
    } while (0)
        if (F == 2) CALLB(2);
        else CALLB(4);
#undef CALLB
        if (merge_from == 0) return nvsf_launch_status();
    }
    const uint32_t l_end = merge_from;  // the levels below go through the atomics
    // production form: corner-parallel run merging.  Variant 1 (tests) selects the plain one-thread-per-(row, level) kernel,
    // which is also the fallback for shapes whose 2^D x F lanes do not divide the levels evenly
    const uint32_t ipw = kWave / ((1u << D) * F);
    if (nvsf_variant(kVarHashgridBwd) == 0 && (L % ipw == 0 || l_end < L)) {
        const uint32_t run = M >= (1u << 20) ? 128u : 32u;  // rows per item: long runs once there is enough work to fill the chip
        const uint32_t l_items = (l_end + ipw - 1) / ipw * ipw;  // items of a wave = ipw consecutive levels of one chunk
        const unsigned long long waves = (unsigned long long)cdiv(M, run) * (l_items / ipw);
        const dim3 cgrid((uint32_t)((waves + kBlock / kWave - 1) / (kBlock / kWave)));
#define CALLC(DD, FF)                                                                                                                \
    do {                                                                                                                             \
        if (grad_is_f16)                                                                                                             \
            hipLaunchKernelGGL((k_hashgrid_bwd_corners<DD, FF, true>), cgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, l_items, meta, \
                               grad_out, go_stride, go_level, grad_table_f32, run, l_end);                                                     \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_hashgrid_bwd_corners<DD, FF, false>), cgrid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, l_items, meta, \
                               grad_out, go_stride, go_level, grad_table_f32, run, l_end);                                                     \
    } while (0)
        DISPATCH_DF(D, F, CALLC);
#undef CALLC
        return nvsf_launch_status();
    }
    const dim3 grid(cdiv(M, kBlock), L);
#define CALL(DD, FF)                                                                                                          \
    do {                                                                                                                         \
        if (grad_is_f16)                                                                                                         \
            hipLaunchKernelGGL((k_hashgrid_bwd<DD, FF, true>), grid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, go_level, grad_table_f32);                                                   \
        else                                                                                                                     \
            hipLaunchKernelGGL((k_hashgrid_bwd<DD, FF, false>), grid, dim3(kBlock), 0, stream, x, M, x_stride, c0, c1, c2, L, meta, \
                               grad_out, go_stride, go_level, grad_table_f32);                                                   \
    } while (0)
    DISPATCH_DF(D, F, CALL);
#undef CALL
    return nvsf_launch_status();
}
}  // namespace

NVSF_API int nvsf_hashgrid_bwd(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L, uint32_t F,
                               const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, const void* grad_out,
                               int grad_is_f16, uint32_t go_stride, float* grad_table_f32, hipStream_t stream) {
    return hashgrid_bwd_launch(x, M, x_stride, cols, D, L, F, h_scales, h_res, h_offsets, grad_out, grad_is_f16, go_stride, 0, grad_table_f32, L, L,
                               nullptr, 0, stream);
}

NVSF_API size_t nvsf_hashgrid_bwd_binned_ws_bytes(uint32_t M, uint32_t L, uint32_t F, const uint32_t* h_res, const uint32_t* h_offsets,
                                                  uint32_t merge_from, uint32_t fine_from) {
    BinPlan bp;
    if (!h_res || !h_offsets || L == 0 || L > (uint32_t)kMaxLevels) return 0;
    return bin_plan(M, L, F, h_res, h_offsets, merge_from, fine_from, bp) ? bp.cursors_bytes + bp.rows_bytes + bp.vals_bytes : 0;
}

NVSF_API int nvsf_hashgrid_bwd_binned(const float* x, uint32_t M, uint32_t x_stride, const uint32_t* cols, uint32_t D, uint32_t L, uint32_t F,
                                      const float* h_scales, const uint32_t* h_res, const uint32_t* h_offsets, const void* grad_out,
                                      int grad_is_f16, uint32_t go_stride, uint32_t go_level_stride, float* grad_table_f32, uint32_t merge_from,
                                      uint32_t fine_from, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return hashgrid_bwd_launch(x, M, x_stride, cols, D, L, F, h_scales, h_res, h_offsets, grad_out, grad_is_f16, go_stride, go_level_stride, grad_table_f32,
                               merge_from, fine_from, workspace, workspace_bytes, stream);
}
