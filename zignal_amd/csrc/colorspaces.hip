// colorspaces.hip — Image(T).convertInto for every colour space of the reference: Hsl, Hsv, Lab, Lch,
// Lms, Oklab, Oklch, Xyb, Xyz, float Ycbcr on either side, and the float -> u8 back-conversions.
//
// Replaces reference src/image.zig:396-407 (loop) + convertColor (src/color.zig:108-151) + the <Space>(T).to / .as
// tables and the conversion functions at src/color.zig:987-1532. The reference routes a conversion through hub spaces
// (`else => self.to(.xyz).to(target)` and friends); the value after each hop is rounded to f32, so the ROUTE is part of
// the arithmetic contract. The kernel walks the same route hop by hop: `next_hop(cur, target)` restates the thirteen
// dispatch tables, `apply_hop` holds each conversion once. Spaces are wave-uniform, so the walk is scalar branching.
// The maths that flows through Zig's std (pow, cbrt, atan2, sin, cos) is in zg_devmath.h — parity-unpinned at the last
// ulp like the oracle's; everything else is plain f32 arithmetic in the reference's operation order, no contraction
// (std.math.lerp IS an fma in Zig's std and is one here).
//
// The legacy fast path (Rgb / Rgba / grey sources to grey, Rgb, Rgba, Oklab, Xyz, u8 Ycbcr: k_convert in convert.hip,
// with the 256-entry sRGB table) stays as it is; this kernel takes every other pair.
#include "zg_common.h"
#include "zg_devmath.h"

#pragma clang fp contract(off)

namespace zg {

__device__ inline float cs_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ inline float cs_lerp(float a, float b, float t) { return __fmaf_rn(b - a, t, a); } // std.math.lerp = @mulAdd
__device__ inline float cs_mod(float x, float y) { // Zig's float @mod
    const float a = fmodf(x, y);
    return x < 0 ? fmodf(a + y, y) : a;
}
__device__ inline uint8_t cs_unit_to_u8(float v) { return (uint8_t)(int)roundf(255.0f * cs_clamp(v, 0.0f, 1.0f)); }
__device__ inline uint8_t cs_clamp_u8(long long v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__device__ inline float cs_gamma_to_linear(float c) { return c > 0.04045f ? dev_powf((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f; } // :1252
__device__ inline float cs_linear_to_gamma(float c) { // :1243 (1.0 / 2.4 is a comptime division)
    return c > 0.0031308f ? 1.055f * dev_powf(c, 0.41666666666666666666666666666667f) - 0.055f : c * 12.92f;
}
__device__ inline float cs_lab_forward(float t) { // :1289
    return t > 0.008856f ? dev_powf(t, 0.33333333333333333333333333333333f) : 7.787f * t + 0.13793103448275862068965517241379f;
}
__device__ inline void cs_linear_to_xyz(float r, float g, float b, float (&o)[4]) {
    o[0] = (r * 0.4124f + g * 0.3576f + b * 0.1805f) * 100;
    o[1] = (r * 0.2126f + g * 0.7152f + b * 0.0722f) * 100;
    o[2] = (r * 0.0193f + g * 0.1192f + b * 0.9505f) * 100;
}
__device__ inline void cs_xyz_to_linear(const float (&v)[4], float (&lin)[3]) { // :1276-1278 == :1437-1439
    lin[0] = (v[0] * 3.2406f + v[1] * -1.5372f + v[2] * -0.4986f) / 100;
    lin[1] = (v[0] * -0.9689f + v[1] * 1.8758f + v[2] * 0.0415f) / 100;
    lin[2] = (v[0] * 0.0557f + v[1] * -0.2040f + v[2] * 1.0570f) / 100;
}
__device__ inline void cs_linear_to_xyb(float r, float g, float b, float (&o)[4]) { // :1441-1454 == :1488-1501
    const float bias = 0.00379307325527544933f, enc = 0.15595420054924863f;
    const float l = fmaxf(0.0f, 0.30f * r + 0.622f * g + 0.078f * b + bias);
    const float m = fmaxf(0.0f, 0.23f * r + 0.692f * g + 0.078f * b + bias);
    const float s = fmaxf(0.0f, 0.24342268924547819f * r + 0.20476744424496821f * g + 0.5518098665095536f * b + bias);
    const float ld = dev_cbrtf(l) - enc, md = dev_cbrtf(m) - enc, sd = dev_cbrtf(s) - enc;
    o[0] = 0.5f * (ld - md);
    o[1] = 0.5f * (ld + md);
    o[2] = sd;
}
__device__ inline void cs_xyb_to_linear(const float (&v)[4], float (&c)[3]) { // :1459-1474 == :1507-1522
    const float bias = 0.00379307325527544933f, dec = 0.15594113236791331f;
    const float lc = (v[1] + v[0]) + dec, mc = (v[1] - v[0]) + dec, sc = v[2] + dec;
    const float l = (lc * lc * lc) - bias, m = (mc * mc * mc) - bias, s = (sc * sc * sc) - bias;
    c[0] = 11.031566901960783f * l - 9.866943921568629f * m - 0.16462299647058826f * s;
    c[1] = -3.254147380392157f * l + 4.418770392156863f * m - 0.16462299647058826f * s;
    c[2] = -3.6588512862745097f * l + 2.7129230470588235f * m + 1.9459282392156863f * s;
}
__device__ inline void cs_cart_to_cyl(float a, float b, float &c, float &h) { // :1333-1338
    c = sqrtf(a * a + b * b);
    h = cs_mod(dev_atan2f(b, a) * 57.295779513082320876798154814105170332405472466564f, 360.0f);
}
__device__ inline void cs_cyl_to_cart(float c, float h, float &a, float &b) { // :1341-1344
    const float h_rad = h * 0.017453292519943295769236907684886127134428718885417f;
    a = c * dev_cosf(h_rad);
    b = c * dev_sinf(h_rad);
}

// <Space>(T).to(target): which space the value is in after the next conversion (color.zig:350-361, 475-480, 533-537,
// 586-592, 624-630, 667-678, 711-717, 750-755, 786-791, 824-830, 863-868, 902-908, 943-948)
__device__ inline int next_hop(int cur, int to) {
    switch (cur) {
    case ZG_CS_RGB:
        switch (to) {
        case ZG_CS_GRAY: case ZG_CS_HSL: case ZG_CS_HSV: case ZG_CS_RGBA: case ZG_CS_XYB: case ZG_CS_XYZ: case ZG_CS_YCBCR: return to;
        default: return ZG_CS_XYZ;
        }
    case ZG_CS_RGBA: case ZG_CS_GRAY: case ZG_CS_YCBCR: return ZG_CS_RGB;
    case ZG_CS_HSV: return to == ZG_CS_HSL ? ZG_CS_HSL : ZG_CS_RGB;
    case ZG_CS_HSL: return to == ZG_CS_HSV ? ZG_CS_HSV : ZG_CS_RGB;
    case ZG_CS_XYZ:
        switch (to) {
        case ZG_CS_LAB: case ZG_CS_LMS: case ZG_CS_OKLAB: case ZG_CS_RGB: case ZG_CS_XYB: return to;
        case ZG_CS_LCH: return ZG_CS_LAB;
        case ZG_CS_OKLCH: return ZG_CS_OKLAB;
        default: return ZG_CS_RGB;
        }
    case ZG_CS_LAB: return to == ZG_CS_LCH ? ZG_CS_LCH : ZG_CS_XYZ;
    case ZG_CS_LCH: return ZG_CS_LAB;
    case ZG_CS_LMS: return ZG_CS_XYZ;
    case ZG_CS_OKLAB: return to == ZG_CS_OKLCH ? ZG_CS_OKLCH : ZG_CS_XYZ;
    case ZG_CS_OKLCH: return ZG_CS_OKLAB;
    case ZG_CS_XYB: return to == ZG_CS_RGB ? ZG_CS_RGB : ZG_CS_XYZ;
    default: return to;
    }
}

// one conversion function of color.zig: v (fields in declaration order) moves from space `cur` to space `nxt`
// `lin`: gammaToLinear of the three channels when they are still the untouched u8 source (256 possible arguments, read
// from the library's table instead of three pows), else null.
__device__ inline void apply_hop(int cur, int nxt, float (&v)[4], const float *lin) {
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    switch (cur) {
    case ZG_CS_RGB:
        switch (nxt) {
        case ZG_CS_GRAY: o[0] = cs_clamp(0.2126f * v[0] + 0.7152f * v[1] + 0.0722f * v[2], 0.0f, 1.0f); break; // :1043-1046
        case ZG_CS_RGBA: o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = 1.0f; break;
        case ZG_CS_XYZ: // :1261-1272
            if (lin) cs_linear_to_xyz(lin[0], lin[1], lin[2], o);
            else cs_linear_to_xyz(cs_gamma_to_linear(v[0]), cs_gamma_to_linear(v[1]), cs_gamma_to_linear(v[2]), o);
            break;
        case ZG_CS_XYB: // :1483-1502
            if (lin) cs_linear_to_xyb(lin[0], lin[1], lin[2], o);
            else cs_linear_to_xyb(cs_gamma_to_linear(v[0]), cs_gamma_to_linear(v[1]), cs_gamma_to_linear(v[2]), o);
            break;
        case ZG_CS_YCBCR: { // :1010-1017
            const float y = cs_clamp(0.299f * v[0] + 0.587f * v[1] + 0.114f * v[2], 0.0f, 1.0f);
            o[0] = y;
            o[1] = cs_clamp((v[2] - y) / 1.772f, -0.5f, 0.5f);
            o[2] = cs_clamp((v[0] - y) / 1.402f, -0.5f, 0.5f);
            break;
        }
        case ZG_CS_HSV: case ZG_CS_HSL: { // :1087-1108, :1150-1174
            const float r = v[0], g = v[1], b = v[2];
            const float mn = fminf(r, fminf(g, b)), mx = fmaxf(r, fmaxf(g, b)), delta = mx - mn;
            if (nxt == ZG_CS_HSV) {
                float h = 0.0f;
                if (delta != 0) {
                    if (mx == r) h = (g - b) / delta * 60;
                    else if (mx == g) h = 120 + (b - r) / delta * 60;
                    else h = 240 + (r - g) / delta * 60;
                }
                o[0] = cs_mod(h, 360.0f);
                o[1] = mx == 0 ? 0.0f : (delta / mx) * 100;
                o[2] = mx * 100;
            } else {
                float hue = 0.0f;
                if (delta != 0) {
                    if (mx == r) hue = (g - b) / delta;
                    else if (mx == g) hue = 2 + (b - r) / delta;
                    else hue = 4 + (r - g) / delta;
                }
                const float l = (mx + mn) / 2.0f;
                const float s = delta == 0 ? 0.0f : (l < 0.5f ? delta / (2 * l) : delta / (2 - 2 * l));
                o[0] = cs_mod(hue * 60.0f, 360.0f);
                o[1] = cs_clamp(s, 0.0f, 1.0f) * 100.0f;
                o[2] = cs_clamp(l, 0.0f, 1.0f) * 100.0f;
            }
            break;
        }
        }
        break;
    case ZG_CS_RGBA: o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; break;   // :478
    case ZG_CS_GRAY: o[0] = o[1] = o[2] = v[0]; break;               // grayToRgb :1050
    case ZG_CS_YCBCR: // :1070-1083
        o[0] = cs_clamp(v[0] + 1.402f * v[2], 0.0f, 1.0f);
        o[1] = cs_clamp(v[0] - 0.344136f * v[1] - 0.714136f * v[2], 0.0f, 1.0f);
        o[2] = cs_clamp(v[0] + 1.772f * v[1], 0.0f, 1.0f);
        break;
    case ZG_CS_HSV:
        if (nxt == ZG_CS_HSL) { // :1211-1224
            const float s_v = v[1] / 100.0f, val = v[2] / 100.0f;
            const float l = val * (1.0f - s_v / 2.0f);
            const float s_l = (l == 0 || l == 1) ? 0.0f : (val - l) / fminf(l, 1 - l);
            o[0] = v[0]; o[1] = s_l * 100.0f; o[2] = l * 100.0f;
        } else { // :1177-1208
            const float hue = cs_clamp(v[0] / 360, 0.0f, 1.0f), sat = cs_clamp(v[1] / 100, 0.0f, 1.0f), val = cs_clamp(v[2] / 100, 0.0f, 1.0f);
            if (sat == 0.0f) { o[0] = o[1] = o[2] = val; break; }
            const float sector = hue * 6;
            const int index = (int)truncf(sector);
            const float f = sector - (float)index;
            const float p = val * (1 - sat), q = val * (1 - (sat * f)), t = val * (1 - sat * (1 - f));
            switch (((index % 6) + 6) % 6) {
            case 0: o[0] = val; o[1] = t; o[2] = p; break;
            case 1: o[0] = q; o[1] = val; o[2] = p; break;
            case 2: o[0] = p; o[1] = val; o[2] = t; break;
            case 3: o[0] = p; o[1] = q; o[2] = val; break;
            case 4: o[0] = t; o[1] = p; o[2] = val; break;
            default: o[0] = val; o[1] = p; o[2] = q; break;
            }
        }
        break;
    case ZG_CS_HSL:
        if (nxt == ZG_CS_HSV) { // :1227-1240
            const float s_l = v[1] / 100.0f, l = v[2] / 100.0f;
            const float val = l + s_l * fminf(l, 1 - l);
            const float s_v = val == 0 ? 0.0f : 2.0f * (1.0f - l / val);
            o[0] = v[0]; o[1] = s_v * 100.0f; o[2] = val * 100.0f;
        } else { // :1111-1147
            const float h = cs_mod(v[0], 360.0f);
            const float s = cs_clamp(v[1] / 100, 0.0f, 1.0f), l = cs_clamp(v[2] / 100, 0.0f, 1.0f);
            const float hue_sector = h / 60.0f;
            const unsigned long long sector = (unsigned long long)truncf(hue_sector);
            const float f = hue_sector - (float)sector;
            float fr, fg, fb;
            switch ((int)(sector % 6)) {
            case 0: fr = 1; fg = f; fb = 0; break;
            case 1: fr = 1 - f; fg = 1; fb = 0; break;
            case 2: fr = 0; fg = 1; fb = f; break;
            case 3: fr = 0; fg = 1 - f; fb = 1; break;
            case 4: fr = f; fg = 0; fb = 1; break;
            default: fr = 1; fg = 0; fb = 1 - f; break;
            }
            const float c0 = cs_lerp(1.0f, 2 * fr, s), c1 = cs_lerp(1.0f, 2 * fg, s), c2 = cs_lerp(1.0f, 2 * fb, s);
            if (l < 0.5f) { o[0] = c0 * l; o[1] = c1 * l; o[2] = c2 * l; }
            else { o[0] = cs_lerp(c0, 2.0f, l) - 1; o[1] = cs_lerp(c1, 2.0f, l) - 1; o[2] = cs_lerp(c2, 2.0f, l) - 1; }
        }
        break;
    case ZG_CS_XYZ:
        switch (nxt) {
        case ZG_CS_LAB: { // :1294-1308
            const float fx = cs_lab_forward(v[0] / 95.047f), fy = cs_lab_forward(v[1] / 100.000f), fz = cs_lab_forward(v[2] / 108.883f);
            o[0] = fmaxf(0.0f, 116.0f * fy - 16.0f);
            o[1] = 500.0f * (fx - fy);
            o[2] = 200.0f * (fy - fz);
            break;
        }
        case ZG_CS_LMS: // :1361-1368
            o[0] = (0.8951f * v[0] + 0.2664f * v[1] - 0.1614f * v[2]) / 100;
            o[1] = (-0.7502f * v[0] + 1.7135f * v[1] + 0.0367f * v[2]) / 100;
            o[2] = (0.0389f * v[0] - 0.0685f * v[1] + 1.0296f * v[2]) / 100;
            break;
        case ZG_CS_OKLAB: { // :1381-1400
            const float x = v[0] / 100.0f, y = v[1] / 100.0f, z = v[2] / 100.0f;
            const float l_linear = 0.8189330101f * x + 0.3618667424f * y - 0.1288597137f * z;
            const float m_linear = 0.0329845436f * x + 0.9293118715f * y + 0.0361456387f * z;
            const float s_linear = 0.0482003018f * x + 0.2643662691f * y + 0.6338517070f * z;
            const float ld = dev_cbrtf(l_linear), md = dev_cbrtf(m_linear), sd = dev_cbrtf(s_linear);
            o[0] = 0.2104542553f * ld + 0.7936177850f * md - 0.0040720468f * sd;
            o[1] = 1.9779984951f * ld - 2.4285922050f * md + 0.4505937099f * sd;
            o[2] = 0.0259040371f * ld + 0.7827717662f * md - 0.8086757660f * sd;
            break;
        }
        case ZG_CS_RGB: { // :1275-1286
            float lin[3];
            cs_xyz_to_linear(v, lin);
            o[0] = cs_clamp(cs_linear_to_gamma(lin[0]), 0.0f, 1.0f);
            o[1] = cs_clamp(cs_linear_to_gamma(lin[1]), 0.0f, 1.0f);
            o[2] = cs_clamp(cs_linear_to_gamma(lin[2]), 0.0f, 1.0f);
            break;
        }
        case ZG_CS_XYB: { // :1435-1454
            float lin[3];
            cs_xyz_to_linear(v, lin);
            cs_linear_to_xyb(lin[0], lin[1], lin[2], o);
            break;
        }
        }
        break;
    case ZG_CS_LAB:
        if (nxt == ZG_CS_LCH) { o[0] = v[0]; cs_cart_to_cyl(v[1], v[2], o[1], o[2]); }
        else { // labToXyz :1311-1330: f64 inside whatever T is
            const double fy = (double)((v[0] + 16.0f) / 116.0f);
            const double fx = (double)(v[1] / 500.0f) + fy;
            const double fz = fy - (double)(v[2] / 200.0f);
            const double y3 = fy * fy * fy, x3 = fx * fx * fx, z3 = fz * fz * fz;
            const double eps = 0.008856, delta = 0.13793103448275862068965517241379, kappa = 7.787;
            const double y = y3 > eps ? y3 : (fy - delta) / kappa;
            const double x = x3 > eps ? x3 : (fx - delta) / kappa;
            const double z = z3 > eps ? z3 : (fz - delta) / kappa;
            o[0] = (float)(x * 95.047);
            o[1] = (float)(y * 100.000);
            o[2] = (float)(z * 108.883);
        }
        break;
    case ZG_CS_LCH: case ZG_CS_OKLCH: o[0] = v[0]; cs_cyl_to_cart(v[1], v[2], o[1], o[2]); break; // :1354-1358, :1428-1432
    case ZG_CS_LMS: // :1371-1378
        o[0] = 100 * (0.9869929f * v[0] - 0.1470543f * v[1] + 0.1599627f * v[2]);
        o[1] = 100 * (0.4323053f * v[0] + 0.5183603f * v[1] + 0.0492912f * v[2]);
        o[2] = 100 * (-0.0085287f * v[0] + 0.0400428f * v[1] + 0.9684867f * v[2]);
        break;
    case ZG_CS_OKLAB:
        if (nxt == ZG_CS_OKLCH) { o[0] = v[0]; cs_cart_to_cyl(v[1], v[2], o[1], o[2]); }
        else { // :1403-1418
            const float ld = v[0] + 0.3963377774f * v[1] + 0.2158037573f * v[2];
            const float md = v[0] - 0.1055613458f * v[1] - 0.0638541728f * v[2];
            const float sd = v[0] - 0.0894841775f * v[1] - 1.2914855480f * v[2];
            const float l = ld * ld * ld, m = md * md * md, s = sd * sd * sd;
            o[0] = 100.0f * (1.2270138511f * l - 0.5577999807f * m + 0.2812561490f * s);
            o[1] = 100.0f * (-0.0405801784f * l + 1.1122568696f * m - 0.0716766787f * s);
            o[2] = 100.0f * (-0.0763812845f * l - 0.4214819784f * m + 1.5861632204f * s);
        }
        break;
    case ZG_CS_XYB: {
        float c[3];
        cs_xyb_to_linear(v, c);
        if (nxt == ZG_CS_RGB) { // :1505-1529
            o[0] = cs_clamp(cs_linear_to_gamma(c[0]), 0.0f, 1.0f);
            o[1] = cs_clamp(cs_linear_to_gamma(c[1]), 0.0f, 1.0f);
            o[2] = cs_clamp(cs_linear_to_gamma(c[2]), 0.0f, 1.0f);
        } else { // :1457-1480
            cs_linear_to_xyz(c[0], c[1], c[2], o);
        }
        break;
    }
    }
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; v[3] = o[3];
}

__device__ inline void cs_to_f32(int from, int to, float (&v)[4], const float *lin = nullptr) {
    int cur = from;
    for (int hop = 0; hop < 8 && cur != to; ++hop) { // longest route: Lch -> Lab -> Xyz -> Rgb -> Hsl
        const int nxt = next_hop(cur, to);
        apply_hop(cur, nxt, v, lin); // grey / Rgba reach Rgb by copies, so `lin` is still valid at the Rgb hop
        cur = nxt;
    }
}

__device__ inline int cs_channels(int space) { return space == ZG_CS_GRAY ? 1 : (space == ZG_CS_RGBA ? 4 : 3); }

// <Space>(u8).to(target) among the u8-backed types (color.zig:350-361, 475-480, 533-537, 943-948)
__device__ inline void cs_u8_to(int from, int to, int (&v)[4]) {
    if (from == to) return;
    int r = v[0], g = v[1], b = v[2];
    if (from == ZG_CS_GRAY) { g = r; b = r; }
    else if (from == ZG_CS_YCBCR) { // ycbcrToRgb :1057-1068
        const long long y = v[0], cb = (long long)v[1] - 128, cr = (long long)v[2] - 128;
        r = cs_clamp_u8((65536 * y + 91881 * cr + 32768) >> 16);
        g = cs_clamp_u8((65536 * y - 22554 * cb - 46802 * cr + 32768) >> 16);
        b = cs_clamp_u8((65536 * y + 116130 * cb + 32768) >> 16);
    }
    switch (to) {
    case ZG_CS_GRAY: { // :1031-1042
        const int y = (13933 * r + 46871 * g + 4732 * b + 32768) >> 16;
        v[0] = y < 0 ? 0 : (y > 255 ? 255 : y);
        break;
    }
    case ZG_CS_RGB: v[0] = r; v[1] = g; v[2] = b; break;
    case ZG_CS_RGBA: v[0] = r; v[1] = g; v[2] = b; v[3] = 255; break;
    case ZG_CS_YCBCR: { // :987-1009
        const long long rr = r, gg = g, bb = b;
        v[0] = cs_clamp_u8((19595ll * rr + 38470ll * gg + 7471ll * bb + 32768) >> 16);
        v[1] = cs_clamp_u8(((-11059ll * rr + -21710ll * gg + 32768ll * bb + 32768) >> 16) + 128);
        v[2] = cs_clamp_u8(((32768ll * rr + -27439ll * gg + -5329ll * bb + 32768) >> 16) + 128);
        break;
    }
    }
}

template <int SPIX, int DPIX>
__global__ __launch_bounds__(256) void k_convert_spaces(DImg src, DImg dst, int src_space, int dst_space, const float *srgb_lut) {
    using SP = Px<SPIX>;
    using DP = Px<DPIX>;
    constexpr bool SF = std::is_same<typename SP::Elem, float>::value;
    constexpr bool DF = std::is_same<typename DP::Elem, float>::value;
    constexpr int SC = SP::C, DC = DP::C;
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    const typename SP::Vec sv = SP::load(src.data, (size_t)r * src.stride + (size_t)c);
    typename DP::Vec dv = DP::zero();
    float f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int u[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < SC; ++i) { if constexpr (SF) f[i] = sv[i]; else u[i] = sv[i]; }

    // convertColor(Dest, source), color.zig:108-151. Scalar images are Zig scalars: converted to the destination's
    // component type first (:122-133); colours go .as(f32).to(space) for float destinations (:146-150) and
    // .to(space).as(u8) for u8 ones (:152).
    if (dst_space == ZG_CS_GRAY) { // colour -> scalar: source.to(.gray).as(Dest).y (the scalar <-> scalar pairs stay in k_convert)
        if constexpr (!SF) {
            cs_u8_to(src_space, ZG_CS_GRAY, u);
            if constexpr (DF) f[0] = (float)u[0] / 255;
        } else {
            cs_to_f32(src_space, ZG_CS_GRAY, f);
            if constexpr (!DF) u[0] = cs_unit_to_u8(f[0]);
        }
    } else if (src_space == ZG_CS_GRAY) {
        if constexpr (DF) {
            float lin[3];
            const float *linp = nullptr;
            if constexpr (!SF) {
                f[0] = (float)u[0] / 255;
                lin[0] = lin[1] = lin[2] = srgb_lut[u[0]];
                linp = lin;
            }
            cs_to_f32(ZG_CS_GRAY, dst_space, f, linp);
        } else {
            if constexpr (SF) u[0] = cs_unit_to_u8(f[0]);
            cs_u8_to(ZG_CS_GRAY, dst_space, u);
        }
    } else if constexpr (DF) {
        if constexpr (!SF) { // <Space>(u8).as(f32): /255, Ycbcr chroma re-centred (:950-981)
#pragma unroll
            for (int i = 0; i < SC; ++i) f[i] = (float)u[i] / 255;
            if (src_space == ZG_CS_YCBCR) { f[1] = ((float)u[1] - 128) / 255; f[2] = ((float)u[2] - 128) / 255; }
        }
        float lin[3];
        const float *linp = nullptr;
        if constexpr (!SF) {
            if (src_space == ZG_CS_RGB || src_space == ZG_CS_RGBA) {
                lin[0] = srgb_lut[u[0]]; lin[1] = srgb_lut[u[1]]; lin[2] = srgb_lut[u[2]];
                linp = lin;
            }
        }
        cs_to_f32(src_space, dst_space, f, linp);
    } else {
        if constexpr (!SF) cs_u8_to(src_space, dst_space, u);
        else {
            cs_to_f32(src_space, dst_space, f);
#pragma unroll
            for (int i = 0; i < DC; ++i) u[i] = cs_unit_to_u8(f[i]);
            if (dst_space == ZG_CS_YCBCR) { u[1] = cs_unit_to_u8(f[1] + 0.5f); u[2] = cs_unit_to_u8(f[2] + 0.5f); }
        }
    }
#pragma unroll
    for (int i = 0; i < DC; ++i) { if constexpr (DF) dv[i] = f[i]; else dv[i] = (uint8_t)u[i]; }
    DP::store(dst.data, (size_t)r * dst.stride + (size_t)c, dv);
}

static bool space_has_u8(int space) { return space == ZG_CS_GRAY || space == ZG_CS_RGB || space == ZG_CS_RGBA || space == ZG_CS_YCBCR; }

// Called by convert_impl (convert.hip) after the shared validation, for the pairs its fast path does not cover.
int convert_spaces_impl(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const float *srgb_lut_dev, hipStream_t s) {
    const bool sf = pixel_is_float(src->pixel), df = pixel_is_float(dst->pixel);
    ZG_REQUIRE(sf || space_has_u8(src_space), ZG_ERR_UNSUPPORTED, "convert: colour space %d has no u8 form (float-only type)", src_space);
    ZG_REQUIRE(df || space_has_u8(dst_space), ZG_ERR_UNSUPPORTED, "convert: colour space %d has no u8 form (float-only type)", dst_space);
    const dim3 grid = row_grid(ceil_div(src->cols, 256), src->rows);
    return dispatch_pixel(src->pixel, [&](auto stag) -> int {
        constexpr int SPIX = decltype(stag)::value;
        return dispatch_pixel(dst->pixel, [&](auto dtag) -> int {
            constexpr int DPIX = decltype(dtag)::value;
            hipLaunchKernelGGL((k_convert_spaces<SPIX, DPIX>), grid, dim3(256), 0, s, dimg(src), dimg(dst), src_space, dst_space, srgb_lut_dev);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
    });
}

} // namespace zg
