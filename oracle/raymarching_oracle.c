/*
 * oracle/raymarching_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential, single-threaded CPU restatement (plain C99, fp32) of the ten device
 * kernels of the reference's `raymarching` extension
 *   /root/reference/nvsf/nerf/raymarching/src/raymarching.cu
 * Every function below cites the reference lines it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (selfsupervised-nvsf_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" against a *compiled* reference.  The reference is a
 * CUDA translation unit (needs nvcc + CUDA runtime + ATen CUDA headers) and therefore is
 * unbuildable in this image; it ships no tests and no golden vectors.  This restatement
 * is pinned instead by (i) closed-form known answers (slab test in fp64, bit-interleave
 * round trips, numpy.packbits), and (ii) cross-checks of the compositing recurrences
 * against the reference's importable PyTorch compositor (renderer_dynamic.py:181-224),
 * see tests/test_oracle_raymarching.py.
 *
 * Floating-point contract: every +,-,*,/ is an individually rounded IEEE fp32 operation
 * (build with -ffp-contract=off); the HIP kernels are built the same way so that the
 * discrete decisions of the marcher (floor/frexp/bit tests) agree exactly.  The CUDA
 * build of the reference contracts a*b+c into FMA at nvcc's discretion, which is not
 * reproducible anywhere else; differences are bounded by 1 ulp per contracted op.
 *
 * Ordering contract: the reference reserves output ranges with atomicAdd
 * (raymarching.cu:445-446), so its sample order depends on thread scheduling.  The
 * canonical order here (and in the HIP path) is ray-index order, i.e. the order a
 * single thread executing rays 0..N-1 would produce.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>

#define ORACLE_API __attribute__((visibility("default")))

/* MADD(a, b, c) = a * b + c.  Default build: two individually rounded operations (the contract above).  Built a second time with
 * -DORACLE_FMAD (liboracle_raymarching_fmad.so) the same sites are single-rounding fmaf -- the contraction nvcc applies by default
 * (--fmad=true) to a product feeding an add in the marcher of raymarching.cu:375-438 (`ox + t * dx`, `x * mip_rbound + 1`, the
 * cell-exit expressions, the jitter of t0).  That variant exists ONLY to price the deviation of this oracle (and of the HIP build,
 * which follows it bit for bit) from a contracted build of the reference: tests/test_oracle_cpu.py marches the same rays both ways
 * and records how often a discrete decision -- a ray's sample count -- differs. */
#ifdef ORACLE_FMAD
#define MADD(a, b, c) fmaf((a), (b), (c))
#else
#define MADD(a, b, c) ((a) * (b) + (c))
#endif

/* ---- helpers: raymarching.cu:25-95 ------------------------------------------------ */
static const float kSqrt3 = 1.7320508075688772f; /* :25 */
static const float kRPi = 0.3183098861837907f;   /* :28 */

static float sgn1(float x) { return copysignf(1.0f, x); }                    /* :35-37 */
static float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); } /* :39-43 */

/* :51-60  cascade level from position (frexpf exponent of the max |coord|) */
static int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}
/* :62-69  cascade level from step size */
static int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}
/* :71-77 */
static uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* :79-86 */
static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* :88-95 */
static uint32_t morton3_inv(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* ---- a1: kernel_near_far_from_aabb, raymarching.cu:104-157 ------------------------ */
ORACLE_API void oracle_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                                          uint32_t N, float min_near, float *nears, float *fars) {
    for (uint32_t n = 0; n < N; ++n) {
        const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
        const float rdx = 1.0f / d[0], rdy = 1.0f / d[1], rdz = 1.0f / d[2];
        float tn = (aabb[0] - o[0]) * rdx, tf = (aabb[3] - o[0]) * rdx, s;
        if (tn > tf) { s = tn; tn = tf; tf = s; }
        float ny = (aabb[1] - o[1]) * rdy, fy = (aabb[4] - o[1]) * rdy;
        if (ny > fy) { s = ny; ny = fy; fy = s; }
        if (tn > fy || ny > tf) { nears[n] = fars[n] = FLT_MAX; continue; } /* :133-136 */
        if (ny > tn) tn = ny;
        if (fy < tf) tf = fy;
        float nz = (aabb[2] - o[2]) * rdz, fz = (aabb[5] - o[2]) * rdz;
        if (nz > fz) { s = nz; nz = fz; fz = s; }
        if (tn > fz || nz > tf) { nears[n] = fars[n] = FLT_MAX; continue; } /* :145-148 */
        if (nz > tn) tn = nz;
        if (fz < tf) tf = fz;
        if (tn < min_near) tn = min_near; /* :153 */
        nears[n] = tn;
        fars[n] = tf;
    }
}

/* ---- a2: kernel_sph_from_ray, raymarching.cu:182-217 ------------------------------ */
ORACLE_API void oracle_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                                    float *coords) {
    for (uint32_t n = 0; n < N; ++n) {
        const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
        const float A = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float B = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
        const float C = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A; /* :206 */
        const float x = o[0] + t * d[0], y = o[1] + t * d[1], z = o[2] + t * d[2];
        const float theta = atan2f(sqrtf(x * x + z * z), y); /* :211 */
        const float phi = atan2f(z, x);                      /* :212 */
        coords[2 * (size_t)n + 0] = 2.0f * theta * kRPi - 1.0f;
        coords[2 * (size_t)n + 1] = phi * kRPi;
    }
}

/* ---- a3: kernel_morton3D / kernel_morton3D_invert, raymarching.cu:237-272 --------- */
ORACLE_API void oracle_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; ++n)
        indices[n] = (int32_t)morton3((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1],
                                      (uint32_t)coords[3 * (size_t)n + 2]);
}
ORACLE_API void oracle_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; ++n) {
        const int32_t ind = indices[n]; /* arithmetic shifts of a signed int, :267-271 */
        coords[3 * (size_t)n + 0] = (int32_t)morton3_inv((uint32_t)(ind >> 0));
        coords[3 * (size_t)n + 1] = (int32_t)morton3_inv((uint32_t)(ind >> 1));
        coords[3 * (size_t)n + 2] = (int32_t)morton3_inv((uint32_t)(ind >> 2));
    }
}

/* ---- a4: kernel_packbits, raymarching.cu:286-306 ---------------------------------- */
ORACLE_API void oracle_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; ++n) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; ++i)
            if (grid[8 * (size_t)n + i] > density_thresh) bits |= (uint8_t)(1u << i);
        bitfield[n] = bits;
    }
}

/* ---- shared marching state (one DDA-style step), raymarching.cu:384-439 ----------- */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max;
    uint32_t C, H;
    const uint8_t *grid;
} march_ctx;

static void ctx_init(march_ctx *c, const float *o, const float *d, const uint8_t *grid, float bound,
                     float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1.0f / d[0]; c->rdy = 1.0f / d[1]; c->rdz = 1.0f / d[2]; /* :361 */
    c->rH = 1.0f / (float)H;                                          /* :362 */
    c->H3 = (float)(H * H * H);                                       /* :363 */
    c->bound = bound; c->dt_gamma = dt_gamma; c->C = C; c->H = H; c->grid = grid;
    c->dt_min = 2.0f * kSqrt3 / (float)max_steps;                     /* :369 */
    c->dt_max = 2.0f * kSqrt3 * (float)(1 << (C - 1)) / (float)H;     /* :370 */
}

/* Evaluates the sample at parameter t.  Returns 1 if the cell is occupied (sample is emitted,
 * xyz/dt filled) else 0 and *t_skip = the t reached after leaving the empty cell (:414-438). */
static int march_probe(const march_ctx *c, float t, float xyz[3], float *dt_out, float *t_skip) {
    const float x = clampf(MADD(t, c->dx, c->ox), -c->bound, c->bound); /* :386-388 */
    const float y = clampf(MADD(t, c->dy, c->oy), -c->bound, c->bound);
    const float z = clampf(MADD(t, c->dz, c->oz), -c->bound, c->bound);
    const float dt = clampf(t * c->dt_gamma, c->dt_min, c->dt_max); /* :390 */
    const int lp = mip_from_pos(x, y, z, (float)c->C), ld = mip_from_dt(dt, (float)c->H, (float)c->C);
    const int level = lp > ld ? lp : ld;                            /* :393-394 */
    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);  /* :396 */
    const float mip_rbound = 1.0f / mip_bound;
    const float Hf = (float)c->H, Hm1 = (float)(c->H - 1);
    /* :400-405 -- the reference evaluates 0.5*(..)*H in double; both factors are exact there */
    const int nx = (int)clampf((float)(0.5 * (double)MADD(x, mip_rbound, 1.0f) * (double)c->H), 0.0f, Hm1);
    const int ny = (int)clampf((float)(0.5 * (double)MADD(y, mip_rbound, 1.0f) * (double)c->H), 0.0f, Hm1);
    const int nz = (int)clampf((float)(0.5 * (double)MADD(z, mip_rbound, 1.0f) * (double)c->H), 0.0f, Hm1);
    (void)Hf;
    const uint32_t index = (uint32_t)((float)level * c->H3 + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz)); /* :407 */
    const int occ = (c->grid[index / 8] & (1u << (index % 8))) != 0;                                                   /* :408 */
    xyz[0] = x; xyz[1] = y; xyz[2] = z;
    *dt_out = dt;
    if (occ) return 1;
    /* distance to the exit face of the empty cell, :420-433 */
    /* (n + 0.5f + 0.5f * sign) and (.. * rH) * 2 - 1 are exact either way (halves of small integers, a doubling) */
    const float tx = MADD(MADD(MADD(0.5f, sgn1(c->dx), (float)nx + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -x) * c->rdx;
    const float ty = MADD(MADD(MADD(0.5f, sgn1(c->dy), (float)ny + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -y) * c->rdy;
    const float tz = MADD(MADD(MADD(0.5f, sgn1(c->dz), (float)nz + 0.5f) * c->rH, 2.0f, -1.0f), mip_bound, -z) * c->rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do { /* :435-437; the reference spins forever once the step drops below 1 ulp of t -- the restatement
          * (and the HIP path) leave the ray instead, which only differs where the reference never returns */
        const float t_next = t + clampf(t * c->dt_gamma, c->dt_min, c->dt_max);
        if (t_next == t) { t = INFINITY; break; }
        t = t_next;
    } while (t < tt);
    *t_skip = t;
    return 0;
}

/* ---- a5: kernel_march_rays_train, raymarching.cu:331-534 -------------------------- */
ORACLE_API void oracle_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                                        float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                        uint32_t M, const float *nears, const float *fars, float *xyzs,
                                        float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                                        const float *noises) {
    for (uint32_t n = 0; n < N; ++n) {
        march_ctx c;
        ctx_init(&c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, dt_gamma, max_steps, C, H);
        const float far = fars[n];
        float t0 = nears[n];
        t0 = MADD(clampf(t0 * dt_gamma, c.dt_min, c.dt_max), noises[n], t0); /* :375 */

        /* pass 1 (:378-439): count */
        float t = t0, xyz[3], dt, tskip;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {
            if (march_probe(&c, t, xyz, &dt, &tskip)) { num_steps++; t += dt; }
            else t = tskip;
        }
        /* reservation (:445-454); sequential execution == ray-index order */
        const uint32_t point_index = (uint32_t)counter[0];
        counter[0] += (int32_t)num_steps;
        /* the reference's slot is atomicAdd(counter+1, 1): with a zeroed counter and sequential execution that
         * is n; with a carried-over counter it indexes past the [N,3] buffer.  Canonical slot here: n. */
        counter[1] += 1;
        rays[3 * (size_t)n + 0] = (int32_t)n;
        rays[3 * (size_t)n + 1] = (int32_t)point_index;
        rays[3 * (size_t)n + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;               /* :456 */
        if (point_index + num_steps > M) continue;  /* :457 */

        /* pass 2 (:459-533): write */
        float *px = xyzs + 3 * (size_t)point_index, *pd = dirs + 3 * (size_t)point_index,
              *pl = deltas + 2 * (size_t)point_index;
        t = t0;
        float last_t = t;
        uint32_t step = 0;
        while (t < far && step < num_steps) {
            if (march_probe(&c, t, xyz, &dt, &tskip)) {
                px[0] = xyz[0]; px[1] = xyz[1]; px[2] = xyz[2];
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t; /* :506 */
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            } else t = tskip;
        }
    }
}

/* ---- a6: kernel_composite_rays_train_forward, raymarching.cu:577-655 -------------- */
ORACLE_API void oracle_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                                    const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                    float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                       num_steps = (uint32_t)rays[3 * (size_t)n + 2];
        if (num_steps == 0 || offset + num_steps > M) { /* :599-606 */
            weights_sum[index] = 0; depth[index] = 0;
            image[3 * (size_t)index] = image[3 * (size_t)index + 1] = image[3 * (size_t)index + 2] = 0;
            continue;
        }
        const float *s = sigmas + offset, *c = rgbs + 3 * (size_t)offset, *dl = deltas + 2 * (size_t)offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; ++step, ++s, c += 3, dl += 2) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]); /* __expf in the reference, :619 */
            const float w = alpha * T;
            r += w * c[0]; g += w * c[1]; b += w * c[2];
            t += dl[1];
            d += w * t;
            ws += w;
            T *= 1.0f - alpha;
            if (T < T_thresh) break; /* :634 (after accumulating this sample) */
        }
        weights_sum[index] = ws; depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}

/* ---- a7: kernel_composite_rays_train_backward, raymarching.cu:690-772 ------------- */
ORACLE_API void oracle_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                                     const float *sigmas, const float *rgbs, const float *deltas,
                                                     const int32_t *rays, const float *weights_sum,
                                                     const float *image, uint32_t M, uint32_t N, float T_thresh,
                                                     float *grad_sigmas, float *grad_rgbs) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                       num_steps = (uint32_t)rays[3 * (size_t)n + 2];
        if (num_steps == 0 || offset + num_steps > M) continue; /* :714 */
        const float gws = grad_weights_sum[index];
        const float *gi = grad_image + 3 * (size_t)index;
        const float r_final = image[3 * (size_t)index], g_final = image[3 * (size_t)index + 1],
                    b_final = image[3 * (size_t)index + 2], ws_final = weights_sum[index];
        const float *s = sigmas + offset, *c = rgbs + 3 * (size_t)offset, *dl = deltas + 2 * (size_t)offset;
        float *gs = grad_sigmas + offset, *gc = grad_rgbs + 3 * (size_t)offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; ++step, ++s, c += 3, dl += 2, ++gs, gc += 3) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float w = alpha * T;
            r += w * c[0]; g += w * c[1]; b += w * c[2];
            ws += w;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * w; gc[1] = gi[1] * w; gc[2] = gi[2] * w; /* :747-749 */
            gs[0] = dl[0] * (gi[0] * (T * c[0] - (r_final - r)) + gi[1] * (T * c[1] - (g_final - g)) +
                             gi[2] * (T * c[2] - (b_final - b)) + gws * (1.0f - ws_final)); /* :752-756 */
            if (T < T_thresh) break;
        }
        (void)ws;
    }
}

/* ---- a8: kernel_march_rays, raymarching.cu:808-928 -------------------------------- */
ORACLE_API void oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive,
                                  const float *rays_t, const float *rays_o, const float *rays_d, float bound,
                                  float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                  const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                                  float *dirs, float *deltas, const float *noises) {
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int32_t index = rays_alive[n];
        march_ctx c;
        ctx_init(&c, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid, bound, dt_gamma, max_steps, C, H);
        float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step,
              *pl = deltas + 2 * (size_t)n * n_step;
        float t = rays_t[index];
        const float far = fars[index];
        (void)nears;
        t = MADD(clampf(t * dt_gamma, c.dt_min, c.dt_max), noises[n], t); /* :856 */
        float last_t = t, xyz[3], dt, tskip;
        uint32_t step = 0;
        while (t < far && step < n_step) {
            if (march_probe(&c, t, xyz, &dt, &tskip)) {
                px[0] = xyz[0]; px[1] = xyz[1]; px[2] = xyz[2];
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            } else t = tskip;
        }
    }
}

/* ---- a9: kernel_composite_rays, raymarching.cu:966-1053 --------------------------- */
ORACLE_API void oracle_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                      float *rays_t, const float *sigmas, const float *rgbs,
                                      const float *deltas, float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int32_t index = rays_alive[n];
        const float *s = sigmas + (size_t)n * n_step, *c = rgbs + 3 * (size_t)n * n_step,
                    *dl = deltas + 2 * (size_t)n * n_step;
        float t = rays_t[index], ws = weights_sum[index], d = depth[index];
        float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break; /* :1005 */
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float T = 1.0f - ws; /* :1015 */
            const float w = alpha * T;
            ws += w;
            t += dl[1];
            d += w * t;
            r += w * c[0]; g += w * c[1]; b += w * c[2];
            if (T < T_thresh) break; /* :1030 */
            ++s; c += 3; dl += 2; ++step;
        }
        if (step < n_step) rays_alive[n] = -1; /* :1042-1046 */
        else rays_t[index] = t;
        weights_sum[index] = ws; depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}
