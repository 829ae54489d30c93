/*
 * oracle/colorspaces.c — every float colour space of the reference's src/color.zig (Hsl, Hsv, Lab, Lch, Lms, Oklab,
 * Oklch, Xyb, Xyz, Ycbcr, Gray, Rgb, Rgba) and the routing between them, SURVEY §8f rank 3.
 * TEST INFRASTRUCTURE ONLY (zo.h).
 *
 * The conversions live in colorspaces_impl.h and are instantiated twice:
 *   f32  the image path (Image(Rgba(u8)).convert(Lab(f32)) and friends): maths from zigmath.c (+ atan2f below),
 *        PARITY UNPINNED at the last ulp like every value that flows through Zig's std maths;
 *   f64  only to PIN the restatement: the reference's unit tests hold exact f64 values for Rgb(u8) -> Hsl / Hsv / Lab
 *        (color.zig:1641-1725) and exact round trips. The f64 std.math.pow restated here (Go's algorithm over the
 *        fdlibm exp / log Zig's compiler-rt ports) reproduces those Lab values bit for bit
 *        (tests/test_oracle_color.py) — which is also the evidence for the f32 instance of the same algorithm.
 *        cbrt / atan2 / sin / cos of the f64 instance come from libm: only the u8 round trips depend on them.
 */
#include "zo.h"
#include <math.h>
#include <string.h>

/* ---- f32: Zig std.math.atan2(f32) / atan(f32), ports of musl atan2f / atanf ------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float zo_atanf(float x) {
    static const float atanhi[] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static const float atanlo[] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    static const float aT[] = {3.3333328366e-01f, -1.9999158382e-01f, 1.4253635705e-01f, -1.0648017377e-01f, 6.1687607318e-02f};
    uint32_t ix = f2u(x);
    const uint32_t sign = ix >> 31;
    int id;
    ix &= 0x7fffffff;
    if (ix >= 0x4c800000) { /* |x| >= 2^26 */
        if (isnan(x)) return x;
        const float z = atanhi[3] + 0x1p-120f;
        return sign ? -z : z;
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x39800000) return x; /* |x| < 2^-12 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) { /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * aT[4]));
    const float s2 = w * (aT[1] + w * aT[3]);
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return sign ? -z : z;
}

float zo_atan2f(float y, float x) {
    static const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    if (isnan(x) || isnan(y)) return x + y;
    uint32_t ix = f2u(x), iy = f2u(y);
    if (ix == 0x3f800000) return zo_atanf(y);
    const uint32_t m = ((iy >> 31) & 1) | ((ix >> 30) & 2); /* 2 * sign(x) + sign(y) */
    ix &= 0x7fffffff;
    iy &= 0x7fffffff;
    if (iy == 0) {
        switch (m) { case 0: case 1: return y; case 2: return pi; default: return -pi; }
    }
    if (ix == 0) return (m & 1) ? -pi / 2 : pi / 2;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) { case 0: return pi / 4; case 1: return -pi / 4; case 2: return 3 * pi / 4; default: return -3 * pi / 4; }
        } else {
            switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi; default: return -pi; }
        }
    }
    if (ix + (26u << 23) < iy || iy == 0x7f800000) return (m & 1) ? -pi / 2 : pi / 2; /* |y/x| > 2^26 */
    float z;
    if ((m & 2) && iy + (26u << 23) < ix) z = 0.0f; /* |y/x| < 2^-26, x < 0 */
    else z = zo_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

/* ---- f64: Zig std.math.pow(f64) = Go's math.Pow over compiler-rt exp / log (fdlibm e_exp.c / e_log.c) ------------ */
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static double zo_exp64(double x) {
    static const double half[2] = {0.5, -0.5}, ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    uint32_t hx = (uint32_t)(d2u(x) >> 32);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    double hi, lo;
    int k;
    if (hx >= 0x4086232b) { /* |x| >= 708.39 or nan */
        if (isnan(x)) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023;
        if (x < -708.39641853226410622 && x < -745.13321910194110842) return 0;
    }
    if (hx > 0x3fd62e42) { /* |x| > 0.5 ln2 */
        if (hx >= 0x3ff0a2b2) k = (int)(invln2 * x + half[sign]);
        else k = 1 - sign - sign;
        hi = x - k * ln2hi;
        lo = k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000) { /* |x| > 2^-28 */
        k = 0; hi = x; lo = 0;
    } else {
        return 1 + x;
    }
    const double xx = x * x;
    const double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    const double y = 1 + (x * c / (2 - c) - lo + hi);
    return k == 0 ? y : ldexp(y, k);
}

static double zo_log64(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, Lg1 = 6.666666666666735130e-01,
                        Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    uint64_t u = d2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000 || hx >> 31) {
        if ((u << 1) == 0) return -INFINITY;
        if (hx >> 31) return NAN;
        k -= 54; x *= 0x1p54; u = d2u(x); hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000) {
        return x;
    } else if (hx == 0x3ff00000 && (u << 32) == 0) {
        return 0;
    }
    hx += 0x3ff00000 - 0x3fe6a09e;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffff) + 0x3fe6a09e;
    x = u2d(((uint64_t)hx << 32) | (u & 0xffffffff));
    const double f = x - 1.0, hfsq = 0.5 * f * f, s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1, dk = k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

double zo_pow64(double x, double y) { /* finite x > 0 paths of std.math.pow; the special cases mirror zo_powf */
    if (y == 0 || x == 1) return 1;
    if (isnan(x) || isnan(y)) return NAN;
    if (y == 1) return x;
    if (x == 0) {
        const int odd = (trunc(y) == y) && (fmod(fabs(y), 2.0) == 1.0);
        if (y < 0) return odd ? copysign(INFINITY, x) : INFINITY;
        return odd ? x : 0.0;
    }
    if (isinf(y)) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (isinf(x)) {
        const int odd = (trunc(y) == y) && (fmod(fabs(y), 2.0) == 1.0);
        if (x < 0) {
            if (y < 0) return odd ? -0.0 : 0.0;
            return odd ? -INFINITY : INFINITY;
        }
        return y < 0 ? 0.0 : INFINITY;
    }
    if (y == 0.5) return sqrt(x);
    if (y == -0.5) return 1 / sqrt(x);
    const double ay = fabs(y);
    double yi = trunc(ay), yf = ay - yi;
    if (yf != 0 && x < 0) return NAN;
    if (yi >= 9223372036854775808.0) return zo_exp64(y * zo_log64(x));
    double a1 = 1.0;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5) { yf -= 1; yi += 1; }
        a1 = zo_exp64(yf * zo_log64(x));
    }
    int xe;
    double x1 = frexp(x, &xe);
    int64_t i = (int64_t)yi;
    while (i != 0) {
        if (xe < -(1 << 12) || (1 << 12) < xe) { ae += xe; break; } /* floatExponentBits(f64) + 1 */
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5) { x1 += x1; xe -= 1; }
        i >>= 1;
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return ldexp(a1, ae);
}

/* ---- the two instances -------------------------------------------------------------------------------------------- */
#define ZT float
#define ZN(name) name##_f
#define ZC(x) x##f
#define ZPOW zo_powf
#define ZCBRT zo_cbrtf
#define ZSQRT sqrtf
#define ZATAN2 zo_atan2f
#define ZSIN zo_sinf
#define ZCOS zo_cosf
#define ZFMOD fmodf
#define ZFMA fmaf
#define ZTRUNC truncf
#include "colorspaces_impl.h"
#undef ZT
#undef ZN
#undef ZC
#undef ZPOW
#undef ZCBRT
#undef ZSQRT
#undef ZATAN2
#undef ZSIN
#undef ZCOS
#undef ZFMOD
#undef ZFMA
#undef ZTRUNC

#define ZT double
#define ZN(name) name##_d
#define ZC(x) x
#define ZPOW zo_pow64
#define ZCBRT cbrt
#define ZSQRT sqrt
#define ZATAN2 atan2
#define ZSIN sin
#define ZCOS cos
#define ZFMOD fmod
#define ZFMA fma
#define ZTRUNC trunc
#include "colorspaces_impl.h"

void zo_color_to_f32(int from, const float in[4], int to, float out[4]) { cs_to_f(from, in, to, out); }
void zo_color_to_f64(int from, const double in[4], int to, double out[4]) { cs_to_d(from, in, to, out); }
