/*
 * oracle/chamfer_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of the reference's Chamfer-distance extension
 *   /root/reference/nvsf/nerf/chamfer3D/chamfer3D.cu:9-138 (NmDistanceKernel: brute-force nearest neighbour,
 *   squared Euclidean distance, index of the nearest point) and :167-195 (NmDistanceGradKernel).
 * PARITY STATUS: "parity unpinned" against a compiled reference (CUDA source, unbuildable in this image, no tests in
 * the reference); pinned by the closed form below being checked against numpy (scipy.spatial cdist / argmin) in
 * tests/test_oracle_cpu.py.  Ties: the lowest index wins (a sequential scan with a strict '<').
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API void oracle_chamfer_forward(const float *xyz1, const float *xyz2, uint32_t B, uint32_t n, uint32_t m, float *dist1,
                                       float *dist2, int32_t *idx1, int32_t *idx2) {
    for (int dir = 0; dir < 2; ++dir) {
        const float *q = dir ? xyz2 : xyz1, *t = dir ? xyz1 : xyz2;
        const uint32_t nq = dir ? m : n, nt = dir ? n : m;
        float *dist = dir ? dist2 : dist1;
        int32_t *idx = dir ? idx2 : idx1;
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t j = 0; j < nq; ++j) {
                const float *p = q + ((size_t)b * nq + j) * 3;
                float best = INFINITY;
                int32_t bi = 0;
                for (uint32_t k = 0; k < nt; ++k) {
                    const float *r = t + ((size_t)b * nt + k) * 3;
                    const float dx = r[0] - p[0], dy = r[1] - p[1], dz = r[2] - p[2];
                    const float d = dx * dx + dy * dy + dz * dz;
                    if (d < best) { best = d; bi = (int32_t)k; }
                }
                dist[(size_t)b * nq + j] = best;
                idx[(size_t)b * nq + j] = bi;
            }
    }
}

ORACLE_API void oracle_chamfer_backward(const float *xyz1, const float *xyz2, uint32_t B, uint32_t n, uint32_t m, const float *g1,
                                        const float *g2, const int32_t *idx1, const int32_t *idx2, float *grad1, float *grad2) {
    for (int dir = 0; dir < 2; ++dir) {
        const float *a = dir ? xyz2 : xyz1, *bp = dir ? xyz1 : xyz2, *g = dir ? g2 : g1;
        const int32_t *idx = dir ? idx2 : idx1;
        float *ga = dir ? grad2 : grad1, *gb = dir ? grad1 : grad2;
        const uint32_t na = dir ? m : n, nb = dir ? n : m;
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t j = 0; j < na; ++j) {
                const size_t i = (size_t)b * na + j;
                const size_t r = (size_t)b * nb + (size_t)idx[i];
                const float s = g[i] * 2.0f;
                for (int c = 0; c < 3; ++c) {
                    const float v = s * (a[i * 3 + c] - bp[r * 3 + c]);
                    ga[i * 3 + c] += v;
                    gb[r * 3 + c] -= v;
                }
            }
    }
}
