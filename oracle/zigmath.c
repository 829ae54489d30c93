/*
 * oracle/zigmath.c — the f32 transcendental functions the reference reaches through Zig's
 * builtins / std.math on the image hot path. TEST INFRASTRUCTURE ONLY (see zo.h).
 *
 * THIRD-PARTY ARITHMETIC NOT UNDER /root/reference: Zig standard library + compiler-rt, pinned only
 * by build.zig.zon:5 `minimum_zig_version = "0.17.0-dev.1441+d5181a9c9"`. Call sites:
 *   @exp f32            src/image.zig:983 (Gaussian taps)
 *   std.math.pow f32    src/color.zig:1255 (gammaToLinear, exponent 2.4)
 *   std.math.cbrt f32   src/color.zig:1391-1393 (xyzToOklab)
 *   @sin/@cos f32       src/image/transforms.zig:139-140,190-191,256-257; interpolation.zig:252
 *
 * Zig's compiler-rt exp/log/sin/cos are ports of musl libc (FreeBSD msun lineage); std.math.cbrt
 * is a port of musl cbrtf; std.math.pow is a port of Go's math.Pow (exp(yf*log x) times an
 * integer power by square-and-multiply on the frexp significand). The functions below restate
 * those PUBLISHED algorithms from memory. No Zig toolchain exists in this image, so their last-ulp
 * agreement with the real Zig functions is PARITY UNPINNED. The product isolates every such value
 * behind a host-supplied argument (taps, cos/sin, the 256-entry sRGB table, the Lanczos table) so a
 * Zig caller keeps Zig's own values; only cbrt and float-typed pow live in device code.
 */
#include "zo.h"
#include <math.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float zo_scalbnf(float x, int n) {
    /* musl scalbnf */
    float y = x;
    if (n > 127) {
        y *= 0x1p127f; n -= 127;
        if (n > 127) { y *= 0x1p127f; n -= 127; if (n > 127) n = 127; }
    } else if (n < -126) {
        y *= 0x1p-126f * 0x1p24f; n += 126 - 24;
        if (n < -126) { y *= 0x1p-126f * 0x1p24f; n += 126 - 24; if (n < -126) n = -126; }
    }
    return y * u2f((uint32_t)(0x7f + n) << 23);
}

/* musl expf (FreeBSD e_expf.c) */
float zo_expf(float x) {
    static const float half[2] = {0.5f, -0.5f};
    static const float ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    static const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    float hi, lo, c, xx, y;
    int k, sign;
    uint32_t hx = f2u(x);
    sign = hx >> 31;
    hx &= 0x7fffffff;
    if (hx >= 0x42aeac50) {
        if (hx > 0x7f800000) return x;
        if (hx >= 0x42b17218 && !sign) { x *= 0x1p127f; return x; }
        if (sign) { if (hx >= 0x42cff1b5) return 0; }
    }
    if (hx > 0x3eb17218) {
        if (hx > 0x3f851592) k = (int)(invln2 * x + half[sign]);
        else k = 1 - sign - sign;
        hi = x - (float)k * ln2hi;
        lo = (float)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000) {
        k = 0; hi = x; lo = 0;
    } else {
        return 1 + x;
    }
    xx = x * x;
    c = x - xx * (P1 + xx * P2);
    y = 1 + (x * c / (2 - c) - lo + hi);
    if (k == 0) return y;
    return zo_scalbnf(y, k);
}

/* musl logf (FreeBSD e_logf.c) */
float zo_logf(float x) {
    static const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    static const float Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    float hfsq, f, s, z, R, w, t1, t2, dk;
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix < 0x00800000 || ix >> 31) {
        if (ix << 1 == 0) return -1 / (x * x);
        if (ix >> 31) return (x - x) / 0.0f;
        k -= 25; x *= 0x1p25f; ix = f2u(x);
    } else if (ix >= 0x7f800000) {
        return x;
    } else if (ix == 0x3f800000) {
        return 0;
    }
    ix += 0x3f800000 - 0x3f3504f3;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffff) + 0x3f3504f3;
    x = u2f(ix);
    f = x - 1.0f;
    s = f / (2.0f + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * Lg4);
    t2 = z * (Lg1 + w * Lg3);
    R = t2 + t1;
    hfsq = 0.5f * f * f;
    dk = (float)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* Zig std.math.pow(f32) — port of Go math.Pow. Only the finite, x > 0 paths that the sRGB
 * transfer function reaches are exercised; the special cases are kept for completeness. */
float zo_powf(float x, float y) {
    if (y == 0 || x == 1) return 1;
    if (isnan(x) || isnan(y)) return NAN;
    if (y == 1) return x;
    if (x == 0) {
        if (y < 0) {
            /* isOddInteger(y) ? copysign(inf, x) : inf */
            float yi = truncf(y);
            int odd = (yi == y) && (fmodf(fabsf(y), 2.0f) == 1.0f);
            return odd ? copysignf(INFINITY, x) : INFINITY;
        } else {
            float yi = truncf(y);
            int odd = (yi == y) && (fmodf(fabsf(y), 2.0f) == 1.0f);
            return odd ? x : 0.0f;
        }
    }
    if (isinf(y)) {
        if (x == -1) return 1;
        if ((fabsf(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (isinf(x)) {
        if (x < 0) {
            float yi = truncf(y);
            int odd = (yi == y) && (fmodf(fabsf(y), 2.0f) == 1.0f);
            if (y < 0) return odd ? -0.0f : 0.0f;
            return odd ? -INFINITY : INFINITY;
        }
        return y < 0 ? 0.0f : INFINITY;
    }
    if (y == 0.5f) return sqrtf(x);
    if (y == -0.5f) return 1 / sqrtf(x);

    float ay = fabsf(y);
    float yi = truncf(ay);
    float yf = ay - yi;
    if (yf != 0 && x < 0) return NAN;
    if (yi >= 2147483648.0f) return zo_expf(y * zo_logf(x));

    float a1 = 1.0f;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5f) { yf -= 1; yi += 1; }
        a1 = zo_expf(yf * zo_logf(x));
    }
    int xe;
    float x1 = frexpf(x, &xe);
    int32_t i = (int32_t)yi;
    while (i != 0) {
        /* overflow_shift = floatExponentBits(f32) + 1 = 9 */
        if (xe < -(1 << 9) || (1 << 9) < xe) { ae += xe; break; }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) { x1 += x1; xe -= 1; }
        i >>= 1;
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return zo_scalbnf(a1, ae);
}

/* Zig std.math.cbrt cbrt32 — port of musl cbrtf (two f64 Newton steps, one final rounding). */
float zo_cbrtf(float x) {
    static const uint32_t B1 = 709958130, B2 = 642849266;
    uint32_t u = f2u(x);
    uint32_t hx = u & 0x7fffffff;
    if (hx >= 0x7f800000) return x + x;
    if (hx < 0x00800000) {
        if (hx == 0) return x;
        u = f2u(x * 0x1p24f);
        hx = u & 0x7fffffff;
        hx = hx / 3 + B2;
    } else {
        hx = hx / 3 + B1;
    }
    u &= 0x80000000;
    u |= hx;
    double t = (double)u2f(u);
    double r = t * t * t;
    t = t * ((double)x + x + r) / (x + r + r);
    r = t * t * t;
    t = t * ((double)x + x + r) / (x + r + r);
    return (float)t;
}

/* musl __sindf / __cosdf / sinf / cosf (argument reduction in double) */
static float k_sindf(double x) {
    static const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59,
                        S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}
static float k_cosdf(double x) {
    static const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57,
                        C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    double z = x * x, w = z * z, r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
static int rem_pio2f(float x, double *y) {
    static const double toint = 1.5 / 2.220446049250313e-16, invpio2 = 6.36619772367581382433e-01,
                        pio2_1 = 1.57079631090164184570e+00, pio2_1t = 1.58932547735281966916e-08;
    uint32_t ix = f2u(x) & 0x7fffffff;
    if (ix < 0x4dc90fdb) { /* |x| ~< 2^28*(pi/2) */
        double fn = (double)x * invpio2 + toint - toint;
        int n = (int)fn;
        *y = x - fn * pio2_1 - fn * pio2_1t;
        return n;
    }
    /* huge arguments: defer to libm's remainder (never reached by the image path's angles) */
    double q = nearbyint((double)x * invpio2);
    *y = (double)x - q * 1.5707963267948966;
    return (int)fmod(q, 4.0);
}
static const double s1pio2 = 1 * 1.5707963267948966, s2pio2 = 2 * 1.5707963267948966,
                    s3pio2 = 3 * 1.5707963267948966, s4pio2 = 4 * 1.5707963267948966;
float zo_sinf(float x) {
    double y;
    uint32_t ix = f2u(x);
    int n, sign = ix >> 31;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {
        if (ix < 0x39800000) return x;
        return k_sindf(x);
    }
    if (ix <= 0x407b53d1) {
        if (ix <= 0x4016cbe3) return sign ? -k_cosdf(x + s1pio2) : k_cosdf(x - s1pio2);
        return k_sindf(sign ? -(x + s2pio2) : -(x - s2pio2));
    }
    if (ix <= 0x40e231d5) {
        if (ix <= 0x40afeddf) return sign ? k_cosdf(x + s3pio2) : -k_cosdf(x - s3pio2);
        return k_sindf(sign ? x + s4pio2 : x - s4pio2);
    }
    if (ix >= 0x7f800000) return x - x;
    n = rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return k_sindf(y);
    case 1: return k_cosdf(y);
    case 2: return k_sindf(-y);
    default: return -k_cosdf(y);
    }
}
float zo_cosf(float x) {
    double y;
    uint32_t ix = f2u(x);
    int n, sign = ix >> 31;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {
        if (ix < 0x39800000) return 1.0f;
        return k_cosdf(x);
    }
    if (ix <= 0x407b53d1) {
        if (ix > 0x4016cbe3) return -k_cosdf(sign ? x + s2pio2 : x - s2pio2);
        return sign ? k_sindf(x + s1pio2) : k_sindf(s1pio2 - x);
    }
    if (ix <= 0x40e231d5) {
        if (ix > 0x40afeddf) return k_cosdf(sign ? x + s4pio2 : x - s4pio2);
        return sign ? k_sindf(-x - s3pio2) : k_sindf(x - s3pio2);
    }
    if (ix >= 0x7f800000) return x - x;
    n = rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return k_cosdf(y);
    case 1: return k_sindf(-y);
    case 2: return -k_cosdf(y);
    default: return k_sindf(y);
    }
}

/* Element-wise application over arrays (the checker's side of tests/test_math_pin.py). fn as in zg_devmath_apply:
 * 0 cbrt, 1 pow(x, 2.4), 2 exp, 3 log, 4 sin, 5 cos, 6 atan2(x, y), 7 pow(x, y), 8 gammaToLinear (color.zig:1252-1258). */
ZO_API int zo_math_apply(int fn, const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const float a = x[i];
        switch (fn) {
        case 0: out[i] = zo_cbrtf(a); break;
        case 1: out[i] = zo_powf(a, 2.4f); break;
        case 2: out[i] = zo_expf(a); break;
        case 3: out[i] = zo_logf(a); break;
        case 4: out[i] = zo_sinf(a); break;
        case 5: out[i] = zo_cosf(a); break;
        case 6: out[i] = zo_atan2f(a, y[i]); break;
        case 7: out[i] = zo_powf(a, y[i]); break;
        case 8: out[i] = a > 0.04045f ? zo_powf((a + 0.055f) / 1.055f, 2.4f) : a / 12.92f; break;
        default: return -1;
        }
    }
    return 0;
}
