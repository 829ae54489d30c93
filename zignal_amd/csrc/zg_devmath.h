// zg_devmath.h — Zig std / compiler-rt f32 maths restated for the device (the algorithms Zig ports from musl and Go),
// used where a reference value flows through @exp / @log / std.math.pow / cbrt / atan2 / @sin / @cos on the image path
// (colour conversion). Same restatement as the oracle's zigmath.c / colorspaces.c, written independently here; like
// those it is parity-unpinned against real Zig at the last ulp. No FMA contraction.
#pragma once
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

__device__ inline float dev_cbrtf_musl(float x) { // Zig std.math.cbrt cbrt32 == musl cbrtf, step for step
    const uint32_t B1 = 709958130u, B2 = 642849266u;
    uint32_t u = __float_as_uint(x);
    uint32_t hx = u & 0x7fffffffu;
    if (hx >= 0x7f800000u) return x + x;
    if (hx < 0x00800000u) {
        if (hx == 0) return x;
        u = __float_as_uint(x * 0x1p24f);
        hx = u & 0x7fffffffu;
        hx = hx / 3 + B2;
    } else {
        hx = hx / 3 + B1;
    }
    u &= 0x80000000u;
    u |= hx;
    double t = (double)__uint_as_float(u);
    double r = t * t * t;
    t = t * ((double)x + x + r) / (x + r + r);
    r = t * t * t;
    t = t * ((double)x + x + r) / (x + r + r);
    return (float)t;
}

// The same function at a quarter of the cost. musl's two Halley steps run in f64 with two IEEE divisions (~25 f64 instructions
// each, and xyzToOklab takes three cube roots per pixel). Its result is the f32 nearest to a double that is accurate to ~2^-50, so
// any value within 2^-40 of the true cube root rounds to the same f32 unless it sits within that distance of a rounding midpoint.
// Here: t = exp2(log2 |x| / 3) from the two hardware transcendentals (good to ~2^-20 between 2^-60 and 2^60) and ONE Halley
// correction c = t (x - t^3) / (x + 2 t^3), written so that only the residual x - t^3 needs more than f32: t^2 and t^2 t are split
// into product + rounding error by FMA (both exact), x - fl(t^3) is exact (the two agree to 2^-19), and what remains are 2^-20-sized
// terms for which f32 is plenty. t + c is then ONE f32 addition, i.e. the correctly rounded sum, and its own rounding error
// (c - (s - t), exact) says how far the sum was from a rounding midpoint: within 2^-16 ulp (one lane in 32 768) the lane takes
// musl's own steps instead; so do arguments outside [2^-60, 2^60). Contraction is off in this file: the FMAs below are the ones
// written. v_log_f32 / v_exp_f32 / v_rcp_f32 are gfx950's: tests/test_math_pin.py compares this function with the musl form over
// ALL 2^32 bit patterns on the part itself, which is the proof that the margins above are wide enough — identical.
// `redo` comes back true when the value returned is not to be used and dev_cbrtf_musl must run instead; callers with several
// cube roots (xyz_to_oklab) run all the fast parts back to back and share ONE rare branch.
__device__ inline float dev_cbrtf_fast(float x, bool &redo) {
    const uint32_t u0 = __float_as_uint(x), hx = u0 & 0x7fffffffu;
    const float mag = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(__uint_as_float(hx)) * 0x1.555556p-2f);
    const float t = __uint_as_float(__float_as_uint(mag) | (u0 & 0x80000000u));
    const float t2 = t * t, e2 = __builtin_fmaf(t, t, -t2);    // t^2   = t2 + e2
    const float t3 = t2 * t, e3 = __builtin_fmaf(t2, t, -t3);  // t2 t  = t3 + e3
    const float resid = ((x - t3) - e3) - e2 * t;              // x - t^3 to ~2^-44 x
    const float c = t * (resid * __builtin_amdgcn_rcpf((x + t3) + t3));
    const float s = t + c;
    const float lost = c - (s - t);                            // exact: |t| > |c|
    const uint32_t se = __float_as_uint(s) & 0x7f800000u;
    const float half_ulp = __uint_as_float(se - (24u << 23)), near = __uint_as_float(se - (39u << 23));
    // a tie is |lost| == half_ulp. Zero is its own cube root and musl says so before doing any arithmetic: it takes that road too,
    // with everything else outside the range, in ONE unsigned comparison.
    redo = (hx - 0x21800000u >= 0x5d800000u - 0x21800000u) | (fabsf(fabsf(lost) - half_ulp) < near);
    return s;
}
__device__ inline float dev_cbrtf(float x) {
    bool redo;
    const float s = dev_cbrtf_fast(x, redo);
    if (redo) return dev_cbrtf_musl(x);
    return s;
}

// x / 100.0f (xyzToOklab, color.zig:1381-1384) without the IEEE-division expansion (~12 instructions, three per pixel): the
// quotient by the correctly rounded reciprocal, the exact remainder with one FMA, one correction. For a constant divisor this
// is correctly rounded wherever neither the quotient nor the remainder leaves the normal range; outside (|x| < 2^-100,
// |x| >= 2^120, inf, nan) the division itself runs. tests/test_math_pin.py compares it with x / 100.0f over ALL 2^32 inputs.
__device__ inline float dev_div100_fast(float x, bool &redo) { // `redo`: as in dev_cbrtf_fast. +0 stays here (the steps below give +0); -0 does not
    const uint32_t u = __float_as_uint(x), ax = u & 0x7fffffffu;
    redo = (ax - 0x0d800000u >= 0x7b800000u - 0x0d800000u) & (u != 0);
    const float r = 0.01f;
    const float q = x * r;
    const float e = __builtin_fmaf(-q, 100.0f, x);
    return __builtin_fmaf(e, r, q);
}
__device__ inline float dev_div100(float x) {
    bool redo;
    const float q = dev_div100_fast(x, redo);
    if (redo) return x / 100.0f;
    return q;
}

__device__ inline float dev_scalbnf(float x, int n) {
    float y = x;
    if (n > 127) {
        y *= 0x1p127f; n -= 127;
        if (n > 127) { y *= 0x1p127f; n -= 127; if (n > 127) n = 127; }
    } else if (n < -126) {
        y *= 0x1p-126f * 0x1p24f; n += 126 - 24;
        if (n < -126) { y *= 0x1p-126f * 0x1p24f; n += 126 - 24; if (n < -126) n = -126; }
    }
    return y * __uint_as_float((uint32_t)(0x7f + n) << 23);
}
__device__ inline float dev_expf(float x) { // musl expf
    const float ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = __float_as_uint(x);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    if (hx >= 0x42aeac50u) {
        if (hx > 0x7f800000u) return x;
        if (hx >= 0x42b17218u && !sign) return x * 0x1p127f;
        if (sign && hx >= 0x42cff1b5u) return 0.0f;
    }
    float hi, lo;
    int k;
    if (hx > 0x3eb17218u) {
        if (hx > 0x3f851592u) k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
        else k = 1 - sign - sign;
        hi = x - (float)k * ln2hi;
        lo = (float)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000u) {
        k = 0; hi = x; lo = 0;
    } else {
        return 1 + x;
    }
    const float xx = x * x;
    const float c = x - xx * (P1 + xx * P2);
    const float y = 1 + (x * c / (2 - c) - lo + hi);
    return k == 0 ? y : dev_scalbnf(y, k);
}
__device__ inline float dev_logf(float x) { // musl logf
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    uint32_t ix = __float_as_uint(x);
    int k = 0;
    if (ix < 0x00800000u || (ix >> 31)) {
        if ((ix << 1) == 0) return -1 / (x * x);
        if (ix >> 31) return (x - x) / 0.0f;
        k -= 25; x *= 0x1p25f; ix = __float_as_uint(x);
    } else if (ix >= 0x7f800000u) {
        return x;
    } else if (ix == 0x3f800000u) {
        return 0;
    }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = __uint_as_float(ix);
    const float f = x - 1.0f, s = f / (2.0f + f), z = s * s, w = z * z;
    const float t1 = w * (Lg2 + w * Lg4), t2 = z * (Lg1 + w * Lg3), R = t2 + t1;
    const float hfsq = 0.5f * f * f, dk = (float)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}
// a / b for operands whose quotient and remainder stay in the normal range (the callers' ranges are stated where they call): the IEEE
// division expansion without its scaling and fix-up steps — reciprocal, one Newton step on it, quotient, two exact-remainder corrections.
__device__ inline float dev_div_normal(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r = __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
// x / D for a constant D with R = the correctly rounded 1 / D: quotient by the reciprocal, the exact remainder with one FMA, one
// correction (dev_div100's form). `redo` comes back true where the quotient or the remainder could leave the normal range (|x| < 2^-100,
// |x| >= 2^120, -0, inf, nan): there the caller divides. Checked against x / D over all 2^32 inputs for every D in use (tests/test_math_pin.py).
__device__ inline float dev_div_const_fast(float x, float D, float R, bool &redo) {
    const uint32_t u = __float_as_uint(x), ax = u & 0x7fffffffu;
    redo = (ax - 0x0d800000u >= 0x7b800000u - 0x0d800000u) & (u != 0);
    const float q = x * R;
    return __builtin_fmaf(__builtin_fmaf(-q, D, x), R, q);
}
// std.math.pow(f32, t, yf) for 2^-100 <= t < 2^120 and a fractional exponent 0 < yf <= 0.5 — labForward's 1 / 3 (color.zig:1289-1291) and
// linearToGamma's 1 / 2.4 (:1243-1249) — Go's algorithm with yi = 0: exp(yf * log(t)), musl's expf and logf — with every branch of the
// two functions turned into selects and their two divisions into dev_div_normal (2 + f in (1.7, 2.42), 2 - c in (1.6, 2.4); numerators
// 0 or normal). The same operations in the same order as dev_powf(t, yf): equal bit for bit on every t of the range
// (tests/test_math_pin.py sweeps all of them, through labForward and linearToGamma).
__device__ inline float dev_pow_frac_fast(float t, float yf) {
    // logf(t), t positive and normal
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    uint32_t ix = __float_as_uint(t);
    ix += 0x3f800000u - 0x3f3504f3u;
    const int kl = (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float xm = __uint_as_float(ix);
    const float f = xm - 1.0f, s = dev_div_normal(f, 2.0f + f), z = s * s, w = z * z;
    const float t1 = w * (Lg2 + w * Lg4), t2 = z * (Lg1 + w * Lg3), R = t2 + t1;
    const float hfsq = 0.5f * f * f, dk = (float)kl;
    const float lg = s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
    // expf(yf * lg): |argument| < 80 over the range, so no overflow / underflow exits
    float x = yf * lg;
    const float ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    const uint32_t hx = __float_as_uint(x) & 0x7fffffffu;
    const bool neg = (__float_as_uint(x) >> 31) != 0;
    const int k_far = (int)(invln2 * x + (neg ? -0.5f : 0.5f));
    const int k = hx > 0x3f851592u ? k_far : (hx > 0x3eb17218u ? (neg ? -1 : 1) : 0);
    const float fk = (float)k;
    const float hi = x - fk * ln2hi, lo = fk * ln2lo; // k = 0: hi = x, lo = +0, as musl's middle branch sets them
    const float xr = hi - lo;
    const float xx = xr * xr;
    const float c = xr - xx * (P1 + xx * P2);
    const float y = 1 + (dev_div_normal(xr * c, 2 - c) - lo + hi);
    const float scaled = y * __uint_as_float((uint32_t)(0x7f + k) << 23); // scalbnf(y, k), |k| < 120; k = 0: y * 1
    return hx > 0x39000000u ? scaled : 1 + x; // |argument| <= 2^-13: musl returns 1 + x
}

// labForward (color.zig:1289-1291) as the route walker evaluates it, and the fast form: equal on every f32 (tests/test_math_pin.py).
__device__ inline float dev_powf(float x, float y);
__device__ inline float dev_lab_forward(float t) {
    return t > 0.008856f ? dev_powf(t, 0.33333333333333333333333333333333f) : 7.787f * t + 0.13793103448275862068965517241379f;
}
__device__ inline float dev_lab_forward_fast(float t, bool &redo) { // `redo`: t >= 2^120, inf or nan — the caller takes dev_lab_forward
    redo = !(t < 0x1p120f);
    const float p = dev_pow_frac_fast(t, 0.33333333333333333333333333333333f); // garbage for t <= 0.008856 (zero, negative, tiny), never selected there
    return t > 0.008856f ? p : 7.787f * t + 0.13793103448275862068965517241379f;
}
// linearToGamma (color.zig:1243-1249; 1.0 / 2.4 is a comptime division) plain and fast, likewise
__device__ inline float dev_linear_to_gamma(float c) {
    return c > 0.0031308f ? 1.055f * dev_powf(c, 0.41666666666666666666666666666667f) - 0.055f : c * 12.92f;
}
__device__ inline float dev_linear_to_gamma_fast(float c, bool &redo) {
    redo = !(c < 0x1p120f);
    const float p = dev_pow_frac_fast(c, 0.41666666666666666666666666666667f);
    return c > 0.0031308f ? 1.055f * p - 0.055f : c * 12.92f;
}
constexpr float LAB_XN = 95.047f, LAB_YN = 100.000f, LAB_ZN = 108.883f; // d65 white (color.zig:1296-1298)
constexpr float LAB_XN_R = 1.0f / 95.047f, LAB_ZN_R = 1.0f / 108.883f;   // correctly rounded by the compiler

// pow(x, 2.4) for finite x > 0 the way Zig's std.math.pow computes it: yi = 2, yf = 0.4.
__device__ inline float dev_pow_2p4(float x) {
    if (x == 1) return 1;
    if (!(x > 0) || !isfinite(x)) return powf(x, 2.4f); // outside the sRGB domain: defer
    const float yf = 2.4f - 2.0f; // modf(|y|).fpart in f32 = 0.4000001, not 0.4f
    float a1 = dev_expf(yf * dev_logf(x));
    int xe;
    float x1 = frexpf(x, &xe);
    int ae = 0;
    // i = 2: bit 0 clear -> square; then i = 1: multiply
    x1 *= x1; xe <<= 1;
    if (x1 < 0.5f) { x1 += x1; xe -= 1; }
    a1 *= x1; ae += xe;
    return dev_scalbnf(a1, ae);
}
__device__ inline float dev_gamma_to_linear(float c) { // color.zig:1252-1258
    return c > 0.04045f ? dev_pow_2p4((c + 0.055f) / 1.055f) : c / 12.92f;
}


// std.math.pow(f32, x, y) for finite x > 0 and finite y > 0 (Go's algorithm: x^yi by squaring on the frexp mantissa,
// x^yf = exp(yf * log(x)) with yf in (-0.5, 0.5]); other arguments defer to the runtime's powf (never reached by the
// sRGB / Lab transfer functions, whose arguments are positive).
__device__ inline float dev_powf(float x, float y) {
    if (y == 0 || x == 1) return 1;
    if (y == 1) return x;
    if (!(x > 0) || !isfinite(x) || !(y > 0) || !isfinite(y)) return powf(x, y);
    if (y == 0.5f) return sqrtf(x);
    float yi = truncf(y), yf = y - yi;
    float a1 = 1.0f;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5f) { yf -= 1; yi += 1; }
        a1 = dev_expf(yf * dev_logf(x));
    }
    int xe;
    float x1 = frexpf(x, &xe);
    int i = (int)yi;
    while (i != 0) {
        if (xe < -(1 << 9) || (1 << 9) < xe) { ae += xe; break; }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) { x1 += x1; xe -= 1; }
        i >>= 1;
    }
    return dev_scalbnf(a1, ae);
}

__device__ inline float dev_atanf(float x) { // std.math.atan(f32) == musl atanf
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333328366e-01f, aT1 = -1.9999158382e-01f, aT2 = 1.4253635705e-01f, aT3 = -1.0648017377e-01f, aT4 = 6.1687607318e-02f;
    uint32_t ix = __float_as_uint(x);
    const uint32_t sign = ix >> 31;
    ix &= 0x7fffffffu;
    if (ix >= 0x4c800000u) { // |x| >= 2^26
        if (ix > 0x7f800000u) return x;
        const float z = atanhi[3] + 0x1p-120f;
        return sign ? -z : z;
    }
    int id;
    if (ix < 0x3ee00000u) { // |x| < 0.4375
        if (ix < 0x39800000u) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {
            if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * aT4));
    const float s2 = w * (aT1 + w * aT3);
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0] : (id == 1 ? atanhi[1] : (id == 2 ? atanhi[2] : atanhi[3]));
    const float lo = id == 0 ? atanlo[0] : (id == 1 ? atanlo[1] : (id == 2 ? atanlo[2] : atanlo[3]));
    z = hi - ((x * (s1 + s2) - lo) - x);
    return sign ? -z : z;
}

__device__ inline float dev_atan2f(float y, float x) { // std.math.atan2(f32) == musl atan2f
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    if (x != x || y != y) return x + y;
    uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
    if (ix == 0x3f800000u) return dev_atanf(y);
    const uint32_t m = ((iy >> 31) & 1u) | ((ix >> 30) & 2u);
    ix &= 0x7fffffffu;
    iy &= 0x7fffffffu;
    if (iy == 0) return m == 0 || m == 1 ? y : (m == 2 ? pi : -pi);
    if (ix == 0) return (m & 1u) ? -pi / 2 : pi / 2;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) return m == 0 ? pi / 4 : (m == 1 ? -pi / 4 : (m == 2 ? 3 * pi / 4 : -3 * pi / 4));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi : -pi));
    }
    if (ix + (26u << 23) < iy || iy == 0x7f800000u) return (m & 1u) ? -pi / 2 : pi / 2;
    float z;
    if ((m & 2u) && iy + (26u << 23) < ix) z = 0.0f;
    else z = dev_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// @sin / @cos on f32 == musl sinf / cosf: polynomial kernels and argument reduction in f64
__device__ inline float dev_k_sindf(double x) {
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59, S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    const double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}
__device__ inline float dev_k_cosdf(double x) {
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57, C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    const double z = x * x, w = z * z, r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
__device__ inline int dev_rem_pio2f(float x, double *y) { // |x| < 2^28 * pi/2 (hue angles are < 2 pi); larger: plain remainder
    const double toint = 1.5 / 2.220446049250313e-16, invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079631090164184570e+00,
                 pio2_1t = 1.58932547735281966916e-08;
    const uint32_t ix = __float_as_uint(x) & 0x7fffffffu;
    if (ix < 0x4dc90fdbu) {
        const double fn = (double)x * invpio2 + toint - toint;
        *y = x - fn * pio2_1 - fn * pio2_1t;
        return (int)fn;
    }
    const double q = nearbyint((double)x * invpio2);
    *y = (double)x - q * 1.5707963267948966;
    return (int)fmod(q, 4.0);
}
__device__ inline float dev_sinf(float x) {
    const double p1 = 1 * 1.5707963267948966, p2 = 2 * 1.5707963267948966, p3 = 3 * 1.5707963267948966, p4 = 4 * 1.5707963267948966;
    uint32_t ix = __float_as_uint(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffffu;
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? x : dev_k_sindf(x);
    if (ix <= 0x407b53d1u) {
        if (ix <= 0x4016cbe3u) return sign ? -dev_k_cosdf(x + p1) : dev_k_cosdf(x - p1);
        return dev_k_sindf(sign ? -(x + p2) : -(x - p2));
    }
    if (ix <= 0x40e231d5u) {
        if (ix <= 0x40afeddfu) return sign ? dev_k_cosdf(x + p3) : -dev_k_cosdf(x - p3);
        return dev_k_sindf(sign ? x + p4 : x - p4);
    }
    if (ix >= 0x7f800000u) return x - x;
    double y;
    const int n = dev_rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return dev_k_sindf(y);
    case 1: return dev_k_cosdf(y);
    case 2: return dev_k_sindf(-y);
    default: return -dev_k_cosdf(y);
    }
}
__device__ inline float dev_cosf(float x) {
    const double p1 = 1 * 1.5707963267948966, p2 = 2 * 1.5707963267948966, p3 = 3 * 1.5707963267948966, p4 = 4 * 1.5707963267948966;
    uint32_t ix = __float_as_uint(x);
    const int sign = (int)(ix >> 31);
    ix &= 0x7fffffffu;
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? 1.0f : dev_k_cosdf(x);
    if (ix <= 0x407b53d1u) {
        if (ix > 0x4016cbe3u) return -dev_k_cosdf(sign ? x + p2 : x - p2);
        return sign ? dev_k_sindf(x + p1) : dev_k_sindf(p1 - x);
    }
    if (ix <= 0x40e231d5u) {
        if (ix > 0x40afeddfu) return dev_k_cosdf(sign ? x + p4 : x - p4);
        return sign ? dev_k_sindf(-x - p3) : dev_k_sindf(x - p3);
    }
    if (ix >= 0x7f800000u) return x - x;
    double y;
    const int n = dev_rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return dev_k_cosdf(y);
    case 1: return dev_k_sindf(-y);
    case 2: return -dev_k_cosdf(y);
    default: return dev_k_sindf(y);
    }
}

} // namespace zg
