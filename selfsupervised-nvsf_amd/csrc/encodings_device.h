// Device helpers for the direction encodings (shared by encodings.hip and the fused head kernels).
#pragma once
#include "common.h"
#include <math.h>

typedef _Float16 h2_pair_t __attribute__((ext_vector_type(2)));

// sin / cos of 2^k * pi * x.  ldexp is exact and sinpi/cospi reduce the argument exactly, so the
// result is accurate for all 12 octaves (a plain sinf(x * 2^k * pi) loses bits at k = 11).
__device__ __forceinline__ void freq_pair(float x, int k, float& s, float& c) {
    const float a = ldexpf(x, k);
    s = sinpif(a);
    c = cospif(a);
}

// 16 real spherical-harmonics basis functions (degree 4) of the direction 2*d01 - 1
__device__ __forceinline__ void sh4_basis(float d0, float d1, float d2, float (&o)[16]) {
    const float x = d0 * 2.0f - 1.0f, y = d1 * 2.0f - 1.0f, z = d2 * 2.0f - 1.0f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}
