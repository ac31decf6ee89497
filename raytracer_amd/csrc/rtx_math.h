// The elementary functions of the light-transport path: powf, expf, sinf, cosf, tanf, asinf, acosf, atan2f.
//
// The reference calls the platform's libm for these (std::pow in the Blinn lobe, Shading.cpp:296-530; std::sin / std::cos
// in the light and lens sampling, RayCaster.cpp:729-762; std::exp in ApplyAbsorption, :281-301 ...). A GPU's math library
// and glibc round a few results differently in the last bit, and in this scene - millimetre-sized cereal tori, glass -
// a last-bit difference is a different path a few bounces later: with OCML on one side and glibc on the other, 0.2 % of
// the per-sample radiances of a small scene and 5 % of those of the full 6.57 M triangle scene part from the oracle,
// although every other operation of the two implementations is the same IEEE operation in the same order (with both
// sides calling THESE routines, every sample of every test scene is bit-identical: tools/parity_levels.py).
//
// So the kernels and the CPU oracle evaluate one set of routines, written with IEEE basic operations only (+, -, *, /,
// sqrt, fma, conversions, integer bit operations): identical source, identical results on any conforming CPU or GPU. No
// hardware approximation instruction, no contraction (the translation units are built with -ffp-contract=off; every
// fused operation below is an explicit fma). Accuracy (tests/test_math_cpu.py, against float64): powf and expf below
// 1 ulp (evaluated in double), the trigonometric functions within 2 ulp on the range the path uses (|x| <= 8192 for
// sin / cos / tan). They stand in for the platform libm the way a second platform's libm would.
//
// Included by the HIP translation units (device and host side) and by oracle/ (plain C++).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RTX_MATH_FN __host__ __device__ __forceinline__
#else
#define RTX_MATH_FN inline
#endif

RTX_MATH_FN uint32_t rtx_f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
RTX_MATH_FN float rtx_u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
RTX_MATH_FN uint64_t rtx_d2u(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
RTX_MATH_FN double rtx_u2d(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// ---- x / 255 for an 8-bit texel value -------------------------------------------------------------------------------
// The float the IEEE division (float)x / 255.0f gives, for x = 0, 1, ... 255 and for those values only: the product with the
// rounded reciprocal, corrected once by its exact residual (Markstein's step). tests/test_math_cpu.py checks all 256 values
// against the division, in exact rational arithmetic and as compiled; a kernel pays 3 instructions for it instead of the 11
// of the general division, and a bilinear texture tap makes twelve of them.
RTX_MATH_FN float rtx_unorm8(float x) {
    const float rc = 0x1.010102p-8f;   // the float nearest to 1 / 255
    const float q = x * rc;
    const float r = fmaf(-255.0f, q, x);
    return fmaf(r, rc, q);
}

// ---- 2^t for a double t, rounded to float (shared by powf and expf) ----------------------------------------------------
// k = nearest integer, r = t - k in [-1/2, 1/2], 2^r by its Taylor polynomial of degree 8 (truncation below 2^-32), the
// power of two applied as an exact factor; the conversion to float does the one rounding (and the gradual underflow).
RTX_MATH_FN float rtx_exp2_to_float(double t) {
    if (!(t < 128.0)) return t != t ? (float)t : rtx_u2f(0x7F800000u);   // NaN -> NaN, overflow -> +inf
    if (t < -151.0) return 0.0f;
    const double shift = 6755399441055744.0;   // 1.5 * 2^52: (t + shift) - shift = t rounded to the nearest integer
    const double kd = (t + shift) - shift;
    const double r = t - kd;
    double p = 0x1.62c0223a5c824p-20;
    p = fma(p, r, 0x1.ffcbfc588b0c7p-17);
    p = fma(p, r, 0x1.430912f86c787p-13);
    p = fma(p, r, 0x1.5d87fe78a6731p-10);
    p = fma(p, r, 0x1.3b2ab6fba4e77p-7);
    p = fma(p, r, 0x1.c6b08d704a0c0p-5);
    p = fma(p, r, 0x1.ebfbdff82c58fp-3);
    p = fma(p, r, 0x1.62e42fefa39efp-1);
    p = fma(p, r, 1.0);
    const int k = (int)kd;                                                            // in [-151, 128]
    const double scale = rtx_u2d((uint64_t)(uint32_t)(k + 1023) << 52);               // 2^k, a normal double
    return (float)(p * scale);
}

// expf(x) = 2^(x * log2 e), the product formed in double
RTX_MATH_FN float rtx_expf(float x) { return rtx_exp2_to_float((double)x * 0x1.71547652b82fep+0); }

// log2 of the positive normal float with bit pattern ix, minus `sub`, in double with a relative error below 2^-38:
// x = 2^k * m with m in [0.699, 1.398), so that x near 1 has k = 0 and nothing cancels where powf(x, 1e5) of the Blinn lobe
// is sensitive; s = (m - 1) / (m + 1), |s| < 0.178, log2(m) = (2 / ln 2) atanh(s) = s (c1 + c3 s^2 + ... + c13 s^12).
// No table: the one division costs a kernel less than two dependent loads.
RTX_MATH_FN double rtx_log2_bits(uint32_t ix, int sub) {
    const uint32_t tmp = ix - 0x3F330000u;
    const int k = ((int32_t)tmp >> 23) - sub;
    const double m = (double)rtx_u2f(ix - (tmp & 0xFF800000u));
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 0x1.c68f568d31760p-3;
    p = fma(p, s2, 0x1.0c9a84994022dp-2);
    p = fma(p, s2, 0x1.484b13d7c02a9p-2);
    p = fma(p, s2, 0x1.a61762a7aded9p-2);
    p = fma(p, s2, 0x1.2776c50ef9bfep-1);
    p = fma(p, s2, 0x1.ec709dc3a03fdp-1);
    p = fma(p, s2, 0x1.71547652b82fep+1);
    return fma(p, s, (double)k);
}

// powf(x, y). The special cases of C99 sit in a function of their own behind one test, so that the common call - x a
// positive normal number, y finite and non-zero - is the table lookup and the two polynomials and nothing else.
RTX_MATH_FN float rtx_powf_special(float x, float y) {
    const uint32_t ix = rtx_f2u(x), iy = rtx_f2u(y);
    if ((iy << 1) == 0u) return 1.0f;                       // pow(x, +-0) = 1, also for NaN x
    if (ix == 0x3F800000u) return 1.0f;                     // pow(1, y) = 1, also for NaN y
    if (x != x || y != y) return x + y;
    const bool yint = floorf(y) == y;                       // an integer (every |y| >= 2^23 is)
    const bool yodd = yint && fabsf(y) < 16777216.0f && ((int)y & 1) != 0;
    if ((ix << 1) == 0u) {                                  // x = +-0
        const float mag = (iy >> 31) ? rtx_u2f(0x7F800000u) : 0.0f;
        return (yodd && (ix >> 31)) ? -mag : mag;
    }
    const float ax = fabsf(x);
    if ((iy << 1) == 0xFF000000u) {                         // y = +-inf
        if (ax == 1.0f) return 1.0f;
        return ((ax > 1.0f) == ((iy >> 31) == 0u)) ? rtx_u2f(0x7F800000u) : 0.0f;
    }
    if ((ix << 1) == 0xFF000000u) {                         // x = +-inf: sign only for (-inf)^(odd integer)
        const float mag = (iy >> 31) ? 0.0f : rtx_u2f(0x7F800000u);
        return (yodd && (ix >> 31)) ? -mag : mag;
    }
    float sign = 1.0f;
    if (ix >> 31) {                                         // finite x < 0: defined for integer y only
        if (!yint) return rtx_u2f(0x7FC00000u);
        if (yodd) sign = -1.0f;
    }
    uint32_t ia = rtx_f2u(ax);
    int sub = 0;
    if (ia < 0x00800000u) {                                 // subnormal: scale by 2^23 (exact) first
        ia = rtx_f2u(ax * 8388608.0f);
        sub = 23;
    }
    return sign * rtx_exp2_to_float((double)y * rtx_log2_bits(ia, sub));
}
RTX_MATH_FN float rtx_powf(float x, float y) {
    // Schlick's (1 - cos)^5 (RayCaster.cpp:253, Shading.cpp:57): the fifth power formed in double is the correctly rounded
    // result but for ties the general path would not resolve either (every special case comes out as in C99: NaN, +-inf,
    // +-0 propagate through the products). A call site with the literal exponent folds this test away.
    if (y == 5.0f) {
        const double d = (double)x, d2 = d * d;
        return (float)(d2 * d2 * d);
    }
    const uint32_t ix = rtx_f2u(x), iy = rtx_f2u(y);
    // x in [2^-126, inf) and 0 < |y| < inf
    if (ix - 0x00800000u < 0x7F000000u && (iy << 1) - 1u < 0xFEFFFFFFu) return rtx_exp2_to_float((double)y * rtx_log2_bits(ix, 0));
    return rtx_powf_special(x, y);
}

// ---- sine and cosine ----------------------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three float pieces (each product is exact inside its fma), then the Cephes minimax
// polynomials on [-pi/4, pi/4]. Meant for the path's arguments (angles up to a few turns); beyond |x| ~ 8192 the
// reduction loses accuracy gradually (results stay in [-1, 1] and stay the same on every platform).
RTX_MATH_FN void rtx_sincos_reduced(float x, float& s, float& c, int& q) {
    const float shift = 12582912.0f;   // 1.5 * 2^23
    const float kf = fmaf(x, 0x1.45f306p-1f, shift) - shift;   // nearest integer of x * 2/pi
    float r = fmaf(-kf, 0x1.921fb6p+0f, x);
    r = fmaf(-kf, -0x1.777a5cp-25f, r);
    r = fmaf(-kf, -0x1.ee59dap-50f, r);
    q = (int)kf & 3;
    const float z = r * r;
    float ps = -1.9515295891E-4f;
    ps = fmaf(ps, z, 8.3321608736E-3f);
    ps = fmaf(ps, z, -1.6666654611E-1f);
    s = fmaf(ps * z, r, r);
    float pc = 2.443315711809948E-5f;
    pc = fmaf(pc, z, -1.388731625493765E-3f);
    pc = fmaf(pc, z, 4.166664568298827E-2f);
    c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
}
RTX_MATH_FN float rtx_sinf(float x) {
    if (!(fabsf(x) < 2147483648.0f)) return fabsf(x) == rtx_u2f(0x7F800000u) || x != x ? rtx_u2f(0x7FC00000u) : 0.0f;
    float s, c;
    int q;
    rtx_sincos_reduced(x, s, c, q);
    const float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
RTX_MATH_FN float rtx_cosf(float x) {
    if (!(fabsf(x) < 2147483648.0f)) return fabsf(x) == rtx_u2f(0x7F800000u) || x != x ? rtx_u2f(0x7FC00000u) : 1.0f;
    float s, c;
    int q;
    rtx_sincos_reduced(x, s, c, q);
    const float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}
RTX_MATH_FN float rtx_tanf(float x) { return rtx_sinf(x) / rtx_cosf(x); }

// ---- inverse functions (Cephes single-precision forms) --------------------------------------------------------------
RTX_MATH_FN float rtx_asin_poly(float x, float z) {   // asin(x) for |x| <= 1/2, z = x*x
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    return fmaf(p * z, x, x);
}
RTX_MATH_FN float rtx_asinf(float x) {
    const float a = fabsf(x);
    if (!(a <= 1.0f)) return rtx_u2f(0x7FC00000u);
    float r;
    if (a > 0.5f) {
        const float z = 0.5f * (1.0f - a);
        const float s = sqrtf(z);
        r = 0x1.921fb6p+0f - 2.0f * rtx_asin_poly(s, z);
    } else
        r = rtx_asin_poly(a, a * a);
    return x < 0.0f ? -r : r;
}
RTX_MATH_FN float rtx_acosf(float x) {
    if (!(fabsf(x) <= 1.0f)) return rtx_u2f(0x7FC00000u);
    if (x > 0.5f) {
        const float z = 0.5f * (1.0f - x);
        return 2.0f * rtx_asin_poly(sqrtf(z), z);
    }
    if (x < -0.5f) {
        const float z = 0.5f * (1.0f + x);
        return 0x1.921fb6p+1f - 2.0f * rtx_asin_poly(sqrtf(z), z);
    }
    return 0x1.921fb6p+0f - rtx_asin_poly(x, x * x);
}
RTX_MATH_FN float rtx_atan_pos(float x) {   // atan(x) for x >= 0
    float y = 0.0f;
    if (x > 2.414213562373095f) {           // tan(3 pi / 8)
        y = 0x1.921fb6p+0f;
        x = -1.0f / x;
    } else if (x > 0.4142135623730950f) {   // tan(pi / 8)
        y = 0x1.921fb6p-1f;
        x = (x - 1.0f) / (x + 1.0f);
    }
    const float z = x * x;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z, 1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    return y + fmaf(p * z, x, x);
}
RTX_MATH_FN float rtx_atan2f(float y, float x) {
    if (x != x || y != y) return x + y;
    const float PI = 0x1.921fb6p+1f, PIO2 = 0x1.921fb6p+0f;
    const bool yneg = (rtx_f2u(y) >> 31) != 0u, xneg = (rtx_f2u(x) >> 31) != 0u;
    float r;
    if (y == 0.0f) r = xneg ? PI : 0.0f;
    else if (x == 0.0f) r = PIO2;
    else {
        const float ax = fabsf(x), ay = fabsf(y);
        const float inf = rtx_u2f(0x7F800000u);
        float a;
        if (ax == inf && ay == inf) a = 0x1.921fb6p-1f;
        else if (ay == inf) a = PIO2;
        else if (ax == inf) a = 0.0f;
        else a = rtx_atan_pos(ay / ax);
        r = xneg ? PI - a : a;
    }
    return yneg ? -r : r;
}
