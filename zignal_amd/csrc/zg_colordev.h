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

} // namespace zg
