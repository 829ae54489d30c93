// zg_hostmath.cpp — see zg_hostmath.h. Host only; nothing here runs on the device.
#include "zg_hostmath.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace zg { namespace hostmath {

namespace {
inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

float scalbn_f32(float x, int n) {
    float y = x;
    if (n > 127) {
        y *= 0x1p127f; n -= 127;
        if (n > 127) { y *= 0x1p127f; n -= 127; if (n > 127) n = 127; }
    } else if (n < -126) {
        y *= 0x1p-126f * 0x1p24f; n += 126 - 24;
        if (n < -126) { y *= 0x1p-126f * 0x1p24f; n += 126 - 24; if (n < -126) n = -126; }
    }
    return y * from_bits((uint32_t)(0x7f + n) << 23);
}

// sin/cos kernels on [-pi/4, pi/4] evaluated in double, as musl __sindf / __cosdf
float sin_kernel(double x) {
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59,
                 S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    const double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}
float cos_kernel(double x) {
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57,
                 C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    const double z = x * x, w = z * z, r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
int reduce_pio2(float x, double *y) {
    const double toint = 1.5 / 2.220446049250313e-16, invpio2 = 6.36619772367581382433e-01,
                 pio2_1 = 1.57079631090164184570e+00, pio2_1t = 1.58932547735281966916e-08;
    if ((bits(x) & 0x7fffffff) < 0x4dc90fdb) {
        const double fn = (double)x * invpio2 + toint - toint;
        *y = x - fn * pio2_1 - fn * pio2_1t;
        return (int)fn;
    }
    const double q = std::nearbyint((double)x * invpio2);
    *y = (double)x - q * 1.5707963267948966;
    return (int)std::fmod(q, 4.0);
}
const double PIO2 = 1.5707963267948966;
} // namespace

float exp_f32(float x) {
    const float ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = bits(x);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx >= 0x42aeac50) {
        if (hx > 0x7f800000) return x;
        if (hx >= 0x42b17218 && !sign) return x * 0x1p127f;
        if (sign && hx >= 0x42cff1b5) return 0;
    }
    float hi, lo;
    int k;
    if (hx > 0x3eb17218) {
        if (hx > 0x3f851592) k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
        else k = 1 - sign - sign;
        hi = x - (float)k * ln2hi;
        lo = (float)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000) {
        k = 0; hi = x; lo = 0;
    } else {
        return 1 + x;
    }
    const float xx = x * x;
    const float c = x - xx * (P1 + xx * P2);
    const float y = 1 + (x * c / (2 - c) - lo + hi);
    return k == 0 ? y : scalbn_f32(y, k);
}

float log_f32(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    uint32_t ix = bits(x);
    int k = 0;
    if (ix < 0x00800000 || ix >> 31) {
        if (ix << 1 == 0) return -1 / (x * x);
        if (ix >> 31) return (x - x) / 0.0f;
        k -= 25; x *= 0x1p25f; ix = bits(x);
    } else if (ix >= 0x7f800000) {
        return x;
    } else if (ix == 0x3f800000) {
        return 0;
    }
    ix += 0x3f800000 - 0x3f3504f3;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffff) + 0x3f3504f3;
    x = from_bits(ix);
    const float f = x - 1.0f, s = f / (2.0f + f), z = s * s, w = z * z;
    const float t1 = w * (Lg2 + w * Lg4), t2 = z * (Lg1 + w * Lg3), R = t2 + t1;
    const float hfsq = 0.5f * f * f, dk = (float)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// Go-style Pow for finite x > 0 and finite y (the only domain the image path reaches).
float pow_f32(float x, float y) {
    if (y == 0 || x == 1) return 1;
    if (std::isnan(x) || std::isnan(y)) return NAN;
    if (y == 1) return x;
    if (!(x > 0) || std::isinf(x) || std::isinf(y)) return std::pow(x, y); // special cases: defer
    if (y == 0.5f) return std::sqrt(x);
    if (y == -0.5f) return 1 / std::sqrt(x);
    float yi = std::trunc(std::fabs(y));
    float yf = std::fabs(y) - yi;
    if (yi >= 2147483648.0f) return exp_f32(y * log_f32(x));
    float a1 = 1.0f;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5f) { yf -= 1; yi += 1; }
        a1 = exp_f32(yf * log_f32(x));
    }
    int xe;
    float x1 = std::frexp(x, &xe);
    for (int32_t i = (int32_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 9) || (1 << 9) < xe) { ae += xe; break; }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) { x1 += x1; xe -= 1; }
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return scalbn_f32(a1, ae);
}

float sin_f32(float x) {
    double y;
    uint32_t ix = bits(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) return ix < 0x39800000 ? x : sin_kernel(x);
    if (ix <= 0x407b53d1) {
        if (ix <= 0x4016cbe3) return sign ? -cos_kernel(x + PIO2) : cos_kernel(x - PIO2);
        return sin_kernel(sign ? -(x + 2 * PIO2) : -(x - 2 * PIO2));
    }
    if (ix <= 0x40e231d5) {
        if (ix <= 0x40afeddf) return sign ? cos_kernel(x + 3 * PIO2) : -cos_kernel(x - 3 * PIO2);
        return sin_kernel(sign ? x + 4 * PIO2 : x - 4 * PIO2);
    }
    if (ix >= 0x7f800000) return x - x;
    switch (reduce_pio2(x, &y) & 3) {
    case 0: return sin_kernel(y);
    case 1: return cos_kernel(y);
    case 2: return sin_kernel(-y);
    default: return -cos_kernel(y);
    }
}

float cos_f32(float x) {
    double y;
    uint32_t ix = bits(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) return ix < 0x39800000 ? 1.0f : cos_kernel(x);
    if (ix <= 0x407b53d1) {
        if (ix > 0x4016cbe3) return -cos_kernel(sign ? x + 2 * PIO2 : x - 2 * PIO2);
        return sign ? sin_kernel(x + PIO2) : sin_kernel(PIO2 - x);
    }
    if (ix <= 0x40e231d5) {
        if (ix > 0x40afeddf) return cos_kernel(sign ? x + 4 * PIO2 : x - 4 * PIO2);
        return sign ? sin_kernel(-x - 3 * PIO2) : sin_kernel(x - 3 * PIO2);
    }
    if (ix >= 0x7f800000) return x - x;
    switch (reduce_pio2(x, &y) & 3) {
    case 0: return cos_kernel(y);
    case 1: return sin_kernel(-y);
    case 2: return -cos_kernel(y);
    default: return sin_kernel(y);
    }
}

float srgb_to_linear(float c) {
    return c > 0.04045f ? pow_f32((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f;
}

const float *srgb_u8_lut() {
    static float lut[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i < 256; ++i) lut[i] = srgb_to_linear((float)i / 255.0f);
    });
    return lut;
}

const float *lanczos3_lut() {
    static float lut[1025];
    static std::once_flag once;
    std::call_once(once, [] {
        const float step = 1024.0f / 3.0f; // comptime: size / max_dist
        for (int i = 0; i < 1025; ++i) {
            const float x = (float)i / step;
            float v;
            if (x == 0) v = 1;
            else if (std::fabs(x) >= 3.0f) v = 0;
            else {
                const float pi_x = 3.14159265358979323846f * x;
                const float pi_x_over_a = pi_x / 3.0f;
                v = (3.0f * sin_f32(pi_x) * sin_f32(pi_x_over_a)) / (pi_x * pi_x);
            }
            lut[i] = v;
        }
    });
    return lut;
}

}} // namespace zg::hostmath
