// zg_colordev.h — the Rgb -> Xyz -> Oklab arithmetic of convertColor (reference src/color.zig:1261-1272 rgbToXyz after
// gammaToLinear, :1381-1400 xyzToOklab) as device functions, shared by k_convert (convert.hip) and the fused
// resize -> convert kernel (resize_planes.hip) so that both produce the same bits by construction.
#pragma once
#include "zg_devmath.h"

#pragma clang fp contract(off)

namespace zg {

// linear RGB -> Xyz scaled by 100 (color.zig:1261-1272)
__device__ inline void linear_rgb_to_xyz(const float lin[3], float &X, float &Y, float &Z) {
    X = (lin[0] * 0.4124f + lin[1] * 0.3576f + lin[2] * 0.1805f) * 100;
    Y = (lin[0] * 0.2126f + lin[1] * 0.7152f + lin[2] * 0.0722f) * 100;
    Z = (lin[0] * 0.0193f + lin[1] * 0.1192f + lin[2] * 0.9505f) * 100;
}

// Xyz -> Oklab (color.zig:1381-1400). IN_RANGE: the caller vouches that X, Y, Z are each +0 or a normal number between 2^-100 and
// 2^120 — true whenever they come from a linearisation table whose entries are +0 or within [2^-60, 2^60] (the sRGB table is;
// the launchers check a caller's table on the host) — and dev_div100's range test is left out.
template <bool IN_RANGE = false>
__device__ inline void xyz_to_oklab(float X, float Y, float Z, float &L, float &A, float &B) {
    float x, y, z;
    if constexpr (IN_RANGE) {
        bool unused;
        x = dev_div100_fast(X, unused); y = dev_div100_fast(Y, unused); z = dev_div100_fast(Z, unused);
    } else {
        bool rx, ry, rz; // the rare inputs the fast forms hand back (zg_devmath.h): one branch for all three
        x = dev_div100_fast(X, rx); y = dev_div100_fast(Y, ry); z = dev_div100_fast(Z, rz); // == X / 100.0f, bit for bit
        if (rx | ry | rz) {
            if (rx) x = X / 100.0f;
            if (ry) y = Y / 100.0f;
            if (rz) z = Z / 100.0f;
        }
    }
    const float l_linear = 0.8189330101f * x + 0.3618667424f * y - 0.1288597137f * z;
    const float m_linear = 0.0329845436f * x + 0.9293118715f * y + 0.0361456387f * z;
    const float s_linear = 0.0482003018f * x + 0.2643662691f * y + 0.6338517070f * z;
    bool rl, rm, rs;
    float l_dash = dev_cbrtf_fast(l_linear, rl), m_dash = dev_cbrtf_fast(m_linear, rm), s_dash = dev_cbrtf_fast(s_linear, rs);
    if (rl | rm | rs) {
        if (rl) l_dash = dev_cbrtf_musl(l_linear);
        if (rm) m_dash = dev_cbrtf_musl(m_linear);
        if (rs) s_dash = dev_cbrtf_musl(s_linear);
    }
    L = 0.2104542553f * l_dash + 0.7936177850f * m_dash - 0.0040720468f * s_dash;
    A = 1.9779984951f * l_dash - 2.4285922050f * m_dash + 0.4505937099f * s_dash;
    B = 0.0259040371f * l_dash + 0.7827717662f * m_dash - 0.8086757660f * s_dash;
}

// Xyz -> Lab (color.zig:1294-1308) with the divisions by the white point and labForward's power in their fast forms (zg_devmath.h: each
// equal to the plain form on every f32). IN_RANGE as in xyz_to_oklab: X, Y, Z each +0 or normal within [2^-100, 2^120), so no form
// ever asks for its plain twin.
template <bool IN_RANGE = false>
__device__ inline void xyz_to_lab(float X, float Y, float Z, float &L, float &A, float &B) {
    bool rx, ry, rz;
    float tx = dev_div_const_fast(X, LAB_XN, LAB_XN_R, rx), ty = dev_div100_fast(Y, ry), tz = dev_div_const_fast(Z, LAB_ZN, LAB_ZN_R, rz);
    if constexpr (!IN_RANGE) {
        if (rx | ry | rz) { // one rare branch for all three
            if (rx) tx = X / LAB_XN;
            if (ry) ty = Y / LAB_YN;
            if (rz) tz = Z / LAB_ZN;
        }
    }
    bool px, py, pz;
    float fx = dev_lab_forward_fast(tx, px), fy = dev_lab_forward_fast(ty, py), fz = dev_lab_forward_fast(tz, pz);
    if constexpr (!IN_RANGE) {
        if (px | py | pz) {
            if (px) fx = dev_lab_forward(tx);
            if (py) fy = dev_lab_forward(ty);
            if (pz) fz = dev_lab_forward(tz);
        }
    }
    L = fmaxf(0.0f, 116.0f * fy - 16.0f);
    A = 500.0f * (fx - fy);
    B = 200.0f * (fy - fz);
}

// Lab -> Xyz -> Rgb in [0, 1] (color.zig:1311-1330 labToXyz: f32 quotients widened to f64, the cubes and the white-point products in f64;
// :1275-1286 xyzToRgb), the constant divisions and linearToGamma's power in their fast forms (zg_devmath.h: each equal to the plain
// form on every f32; where a fast form hands a value back the plain one runs — one rare branch per group).
__device__ inline void lab_to_rgb_unit(float L, float A, float B, float (&rgb)[3]) {
    bool r0, r1, r2;
    const float l16 = L + 16.0f;
    float qy = dev_div_const_fast(l16, 116.0f, 1.0f / 116.0f, r0), qa = dev_div_const_fast(A, 500.0f, 1.0f / 500.0f, r1), qb = dev_div_const_fast(B, 200.0f, 1.0f / 200.0f, r2);
    if (r0 | r1 | r2) {
        if (r0) qy = l16 / 116.0f;
        if (r1) qa = A / 500.0f;
        if (r2) qb = B / 200.0f;
    }
    const double fy = (double)qy, fx = (double)qa + fy, fz = fy - (double)qb;
    const double y3 = fy * fy * fy, x3 = fx * fx * fx, z3 = fz * fz * fz;
    const double eps = 0.008856, delta = 0.13793103448275862068965517241379, kappa = 7.787;
    double y = y3, x = x3, z = z3;
    if (!(y3 > eps) | !(x3 > eps) | !(z3 > eps)) { // dark colours: the linear piece, a true f64 division
        if (!(y3 > eps)) y = (fy - delta) / kappa;
        if (!(x3 > eps)) x = (fx - delta) / kappa;
        if (!(z3 > eps)) z = (fz - delta) / kappa;
    }
    const float X = (float)(x * 95.047), Y = (float)(y * 100.000), Z = (float)(z * 108.883);
    const float sr = X * 3.2406f + Y * -1.5372f + Z * -0.4986f, sg = X * -0.9689f + Y * 1.8758f + Z * 0.0415f, sb = X * 0.0557f + Y * -0.2040f + Z * 1.0570f;
    bool d0, d1, d2;
    float lr = dev_div100_fast(sr, d0), lg = dev_div100_fast(sg, d1), lb = dev_div100_fast(sb, d2);
    if (d0 | d1 | d2) {
        if (d0) lr = sr / 100.0f;
        if (d1) lg = sg / 100.0f;
        if (d2) lb = sb / 100.0f;
    }
    bool g0, g1, g2;
    float cr = dev_linear_to_gamma_fast(lr, g0), cg = dev_linear_to_gamma_fast(lg, g1), cb = dev_linear_to_gamma_fast(lb, g2);
    if (g0 | g1 | g2) {
        if (g0) cr = dev_linear_to_gamma(lr);
        if (g1) cg = dev_linear_to_gamma(lg);
        if (g2) cb = dev_linear_to_gamma(lb);
    }
    rgb[0] = cr; rgb[1] = cg; rgb[2] = cb;
}

} // namespace zg
