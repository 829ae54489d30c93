/*
 * oracle/colorspaces_impl.h — the float colour-space conversions of the reference's src/color.zig, written once and
 * instantiated for f32 (the image path) and f64 (to pin the restatement against the reference's f64 golden values,
 * color.zig:1641-1725). TEST INFRASTRUCTURE ONLY (zo.h). Included by colorspaces.c with
 *   ZT        float | double
 *   ZN(name)  name##_f | name##_d
 *   ZC(x)     x##f | x          a comptime_float constant rounded once to T, as Zig does
 *   ZPOW, ZCBRT, ZSQRT, ZATAN2, ZSIN, ZCOS, ZFMOD, ZFMA, ZTRUNC   the maths of that width
 * Every function cites the reference lines it restates. Operation order is the reference's; nothing is contracted
 * (-ffp-contract=off) except std.math.lerp, which IS @mulAdd in Zig's std.
 */

static inline ZT ZN(clampv)(ZT v, ZT lo, ZT hi) { return v < lo ? lo : (v > hi ? hi : v); } /* std.math.clamp */
static inline ZT ZN(lerp)(ZT a, ZT b, ZT t) { return ZFMA(b - a, t, a); }                     /* std.math.lerp = @mulAdd(b - a, t, a) */
static inline ZT ZN(maxv)(ZT a, ZT b) { return a > b ? a : b; }
static inline ZT ZN(minv)(ZT a, ZT b) { return a < b ? a : b; }
/* Zig's float @mod: r = fmod(x, y); x < 0 ? fmod(r + y, y) : r (LLVM lowering of airMod) */
static inline ZT ZN(zmod)(ZT x, ZT y) {
    const ZT a = ZFMOD(x, y);
    if (x < 0) return ZFMOD(a + y, y);
    return a;
}

static ZT ZN(gamma_to_linear)(ZT c) { /* color.zig:1252-1258 */
    return c > ZC(0.04045) ? ZPOW((c + ZC(0.055)) / ZC(1.055), ZC(2.4)) : c / ZC(12.92);
}
static ZT ZN(linear_to_gamma)(ZT c) { /* color.zig:1243-1249; 1.0 / 2.4 is a comptime division */
    return c > ZC(0.0031308) ? ZC(1.055) * ZPOW(c, ZC(0.41666666666666666666666666666667)) - ZC(0.055) : c * ZC(12.92);
}

static void ZN(rgb_to_xyz)(const ZT in[3], ZT out[3]) { /* color.zig:1261-1272 */
    const ZT r = ZN(gamma_to_linear)(in[0]), g = ZN(gamma_to_linear)(in[1]), b = ZN(gamma_to_linear)(in[2]);
    out[0] = (r * ZC(0.4124) + g * ZC(0.3576) + b * ZC(0.1805)) * 100;
    out[1] = (r * ZC(0.2126) + g * ZC(0.7152) + b * ZC(0.0722)) * 100;
    out[2] = (r * ZC(0.0193) + g * ZC(0.1192) + b * ZC(0.9505)) * 100;
}
static void ZN(xyz_to_linear_rgb)(const ZT xyz[3], ZT rgb[3]) { /* shared by xyzToRgb :1276-1278 and xyzToXyb :1437-1439 */
    rgb[0] = (xyz[0] * ZC(3.2406) + xyz[1] * ZC(-1.5372) + xyz[2] * ZC(-0.4986)) / 100;
    rgb[1] = (xyz[0] * ZC(-0.9689) + xyz[1] * ZC(1.8758) + xyz[2] * ZC(0.0415)) / 100;
    rgb[2] = (xyz[0] * ZC(0.0557) + xyz[1] * ZC(-0.2040) + xyz[2] * ZC(1.0570)) / 100;
}
static void ZN(xyz_to_rgb)(const ZT xyz[3], ZT out[3]) { /* color.zig:1275-1286 */
    ZT lin[3];
    ZN(xyz_to_linear_rgb)(xyz, lin);
    for (int i = 0; i < 3; ++i) out[i] = ZN(clampv)(ZN(linear_to_gamma)(lin[i]), 0, 1);
}

static ZT ZN(lab_forward)(ZT t) { /* color.zig:1289-1291; 1.0 / 3.0 and lab_delta = 16.0 / 116.0 are comptime */
    return t > ZC(0.008856) ? ZPOW(t, ZC(0.33333333333333333333333333333333)) : ZC(7.787) * t + ZC(0.13793103448275862068965517241379);
}
static void ZN(xyz_to_lab)(const ZT xyz[3], ZT out[3]) { /* color.zig:1294-1308 */
    const ZT fx = ZN(lab_forward)(xyz[0] / ZC(95.047)), fy = ZN(lab_forward)(xyz[1] / ZC(100.000)), fz = ZN(lab_forward)(xyz[2] / ZC(108.883));
    out[0] = ZN(maxv)(0, ZC(116.0) * fy - ZC(16.0));
    out[1] = ZC(500.0) * (fx - fy);
    out[2] = ZC(200.0) * (fy - fz);
}
static void ZN(lab_to_xyz)(const ZT lab[3], ZT out[3]) { /* color.zig:1311-1330: f64 inside whatever T is */
    const double fy = (double)((lab[0] + ZC(16.0)) / ZC(116.0));
    const double fx = (double)(lab[1] / ZC(500.0)) + fy;
    const double fz = fy - (double)(lab[2] / ZC(200.0));
    const double y3 = fy * fy * fy, x3 = fx * fx * fx, z3 = fz * fz * fz;
    const double eps = 0.008856, delta = 0.13793103448275862068965517241379, kappa = 7.787;
    const double y = y3 > eps ? y3 : (fy - delta) / kappa;
    const double x = x3 > eps ? x3 : (fx - delta) / kappa;
    const double z = z3 > eps ? z3 : (fz - delta) / kappa;
    out[0] = (ZT)(x * 95.047);
    out[1] = (ZT)(y * 100.000);
    out[2] = (ZT)(z * 108.883);
}

/* color.zig:1333-1344; std.math.radiansToDegrees / degreesToRadians multiply by deg_per_rad / rad_per_deg */
static void ZN(cart_to_cyl)(ZT a, ZT b, ZT *c, ZT *h) {
    *c = ZSQRT(a * a + b * b);
    *h = ZN(zmod)(ZATAN2(b, a) * ZC(57.295779513082320876798154814105170332405472466564), ZC(360.0));
}
static void ZN(cyl_to_cart)(ZT c, ZT h, ZT *a, ZT *b) {
    const ZT h_rad = h * ZC(0.017453292519943295769236907684886127134428718885417);
    *a = c * ZCOS(h_rad);
    *b = c * ZSIN(h_rad);
}

static void ZN(xyz_to_lms)(const ZT xyz[3], ZT out[3]) { /* color.zig:1361-1368 */
    out[0] = (ZC(0.8951) * xyz[0] + ZC(0.2664) * xyz[1] - ZC(0.1614) * xyz[2]) / 100;
    out[1] = (ZC(-0.7502) * xyz[0] + ZC(1.7135) * xyz[1] + ZC(0.0367) * xyz[2]) / 100;
    out[2] = (ZC(0.0389) * xyz[0] - ZC(0.0685) * xyz[1] + ZC(1.0296) * xyz[2]) / 100;
}
static void ZN(lms_to_xyz)(const ZT lms[3], ZT out[3]) { /* color.zig:1371-1378 */
    out[0] = 100 * (ZC(0.9869929) * lms[0] - ZC(0.1470543) * lms[1] + ZC(0.1599627) * lms[2]);
    out[1] = 100 * (ZC(0.4323053) * lms[0] + ZC(0.5183603) * lms[1] + ZC(0.0492912) * lms[2]);
    out[2] = 100 * (ZC(-0.0085287) * lms[0] + ZC(0.0400428) * lms[1] + ZC(0.9684867) * lms[2]);
}

static void ZN(xyz_to_oklab)(const ZT xyz[3], ZT out[3]) { /* color.zig:1381-1400 */
    const ZT x = xyz[0] / ZC(100.0), y = xyz[1] / ZC(100.0), z = xyz[2] / ZC(100.0);
    const ZT l_linear = ZC(0.8189330101) * x + ZC(0.3618667424) * y - ZC(0.1288597137) * z;
    const ZT m_linear = ZC(0.0329845436) * x + ZC(0.9293118715) * y + ZC(0.0361456387) * z;
    const ZT s_linear = ZC(0.0482003018) * x + ZC(0.2643662691) * y + ZC(0.6338517070) * z;
    const ZT l_dash = ZCBRT(l_linear), m_dash = ZCBRT(m_linear), s_dash = ZCBRT(s_linear);
    out[0] = ZC(0.2104542553) * l_dash + ZC(0.7936177850) * m_dash - ZC(0.0040720468) * s_dash;
    out[1] = ZC(1.9779984951) * l_dash - ZC(2.4285922050) * m_dash + ZC(0.4505937099) * s_dash;
    out[2] = ZC(0.0259040371) * l_dash + ZC(0.7827717662) * m_dash - ZC(0.8086757660) * s_dash;
}
static void ZN(oklab_to_xyz)(const ZT lab[3], ZT out[3]) { /* color.zig:1403-1418 */
    const ZT l_dash = lab[0] + ZC(0.3963377774) * lab[1] + ZC(0.2158037573) * lab[2];
    const ZT m_dash = lab[0] - ZC(0.1055613458) * lab[1] - ZC(0.0638541728) * lab[2];
    const ZT s_dash = lab[0] - ZC(0.0894841775) * lab[1] - ZC(1.2914855480) * lab[2];
    const ZT l = l_dash * l_dash * l_dash, m = m_dash * m_dash * m_dash, s = s_dash * s_dash * s_dash;
    out[0] = ZC(100.0) * (ZC(1.2270138511) * l - ZC(0.5577999807) * m + ZC(0.2812561490) * s);
    out[1] = ZC(100.0) * (ZC(-0.0405801784) * l + ZC(1.1122568696) * m - ZC(0.0716766787) * s);
    out[2] = ZC(100.0) * (ZC(-0.0763812845) * l - ZC(0.4214819784) * m + ZC(1.5861632204) * s);
}

/* XYB (JPEG XL): color.zig:1435-1532 */
static void ZN(linear_rgb_to_xyb)(ZT r, ZT g, ZT b, ZT out[3]) { /* :1441-1454 == :1488-1501 */
    const ZT bias = ZC(0.00379307325527544933), enc = ZC(0.15595420054924863);
    const ZT l = ZN(maxv)(0, ZC(0.30) * r + ZC(0.622) * g + ZC(0.078) * b + bias);
    const ZT m = ZN(maxv)(0, ZC(0.23) * r + ZC(0.692) * g + ZC(0.078) * b + bias);
    const ZT s = ZN(maxv)(0, ZC(0.24342268924547819) * r + ZC(0.20476744424496821) * g + ZC(0.5518098665095536) * b + bias);
    const ZT l_dash = ZCBRT(l) - enc, m_dash = ZCBRT(m) - enc, s_dash = ZCBRT(s) - enc;
    out[0] = ZC(0.5) * (l_dash - m_dash);
    out[1] = ZC(0.5) * (l_dash + m_dash);
    out[2] = s_dash;
}
static void ZN(xyb_to_linear_rgb)(const ZT xyb[3], ZT rgb[3]) { /* :1459-1474 == :1507-1522 */
    const ZT bias = ZC(0.00379307325527544933), dec = ZC(0.15594113236791331);
    const ZT l_cbrt = (xyb[1] + xyb[0]) + dec, m_cbrt = (xyb[1] - xyb[0]) + dec, s_cbrt = xyb[2] + dec;
    const ZT l = (l_cbrt * l_cbrt * l_cbrt) - bias, m = (m_cbrt * m_cbrt * m_cbrt) - bias, s = (s_cbrt * s_cbrt * s_cbrt) - bias;
    rgb[0] = ZC(11.031566901960783) * l - ZC(9.866943921568629) * m - ZC(0.16462299647058826) * s;
    rgb[1] = ZC(-3.254147380392157) * l + ZC(4.418770392156863) * m - ZC(0.16462299647058826) * s;
    rgb[2] = ZC(-3.6588512862745097) * l + ZC(2.7129230470588235) * m + ZC(1.9459282392156863) * s;
}
static void ZN(xyz_to_xyb)(const ZT xyz[3], ZT out[3]) { /* :1435-1454 */
    ZT lin[3];
    ZN(xyz_to_linear_rgb)(xyz, lin);
    ZN(linear_rgb_to_xyb)(lin[0], lin[1], lin[2], out);
}
static void ZN(rgb_to_xyb)(const ZT rgb[3], ZT out[3]) { /* :1483-1502 */
    ZN(linear_rgb_to_xyb)(ZN(gamma_to_linear)(rgb[0]), ZN(gamma_to_linear)(rgb[1]), ZN(gamma_to_linear)(rgb[2]), out);
}
static void ZN(xyb_to_xyz)(const ZT xyb[3], ZT out[3]) { /* :1457-1480 */
    ZT c[3];
    ZN(xyb_to_linear_rgb)(xyb, c);
    out[0] = (c[0] * ZC(0.4124) + c[1] * ZC(0.3576) + c[2] * ZC(0.1805)) * 100;
    out[1] = (c[0] * ZC(0.2126) + c[1] * ZC(0.7152) + c[2] * ZC(0.0722)) * 100;
    out[2] = (c[0] * ZC(0.0193) + c[1] * ZC(0.1192) + c[2] * ZC(0.9505)) * 100;
}
static void ZN(xyb_to_rgb)(const ZT xyb[3], ZT out[3]) { /* :1505-1529 */
    ZT c[3];
    ZN(xyb_to_linear_rgb)(xyb, c);
    for (int i = 0; i < 3; ++i) out[i] = ZN(clampv)(ZN(linear_to_gamma)(c[i]), 0, 1);
}

static void ZN(rgb_to_hsv)(const ZT rgb[3], ZT out[3]) { /* color.zig:1087-1108 */
    const ZT r = rgb[0], g = rgb[1], b = rgb[2];
    const ZT mn = ZN(minv)(r, ZN(minv)(g, b)), mx = ZN(maxv)(r, ZN(maxv)(g, b)), delta = mx - mn;
    ZT h = 0;
    if (delta != 0) {
        if (mx == r) h = (g - b) / delta * 60;
        else if (mx == g) h = 120 + (b - r) / delta * 60;
        else h = 240 + (r - g) / delta * 60;
    }
    out[0] = ZN(zmod)(h, ZC(360.0));
    out[1] = mx == 0 ? 0 : (delta / mx) * 100;
    out[2] = mx * 100;
}
static void ZN(rgb_to_hsl)(const ZT rgb[3], ZT out[3]) { /* color.zig:1150-1174 */
    const ZT r = rgb[0], g = rgb[1], b = rgb[2];
    const ZT mn = ZN(minv)(r, ZN(minv)(g, b)), mx = ZN(maxv)(r, ZN(maxv)(g, b)), delta = mx - mn;
    ZT hue = 0;
    if (delta != 0) {
        if (mx == r) hue = (g - b) / delta;
        else if (mx == g) hue = 2 + (b - r) / delta;
        else hue = 4 + (r - g) / delta;
    }
    const ZT l = (mx + mn) / ZC(2.0);
    const ZT s = delta == 0 ? 0 : (l < ZC(0.5) ? delta / (2 * l) : delta / (2 - 2 * l));
    out[0] = ZN(zmod)(hue * ZC(60.0), ZC(360.0));
    out[1] = ZN(clampv)(s, 0, 1) * ZC(100.0);
    out[2] = ZN(clampv)(l, 0, 1) * ZC(100.0);
}
static void ZN(hsl_to_rgb)(const ZT hsl[3], ZT out[3]) { /* color.zig:1111-1147 */
    const ZT h = ZN(zmod)(hsl[0], 360);
    const ZT s = ZN(clampv)(hsl[1] / 100, 0, 1), l = ZN(clampv)(hsl[2] / 100, 0, 1);
    const ZT hue_sector = h / ZC(60.0);
    const uint64_t sector = (uint64_t)ZTRUNC(hue_sector);
    const ZT f = hue_sector - (ZT)sector;
    const ZT factors[6][3] = {{1, f, 0}, {1 - f, 1, 0}, {0, 1, f}, {0, 1 - f, 1}, {f, 0, 1}, {1, 0, 1 - f}};
    const unsigned idx = (unsigned)(sector % 6);
    ZT c[3];
    for (int i = 0; i < 3; ++i) c[i] = ZN(lerp)(1, 2 * factors[idx][i], s);
    if (l < ZC(0.5)) for (int i = 0; i < 3; ++i) out[i] = c[i] * l;
    else for (int i = 0; i < 3; ++i) out[i] = ZN(lerp)(c[i], 2, l) - 1;
}
static void ZN(hsv_to_rgb)(const ZT hsv[3], ZT out[3]) { /* color.zig:1177-1208 */
    const ZT hue = ZN(clampv)(hsv[0] / 360, 0, 1), sat = ZN(clampv)(hsv[1] / 100, 0, 1), val = ZN(clampv)(hsv[2] / 100, 0, 1);
    if (sat == ZC(0.0)) { out[0] = out[1] = out[2] = val; return; }
    const ZT sector = hue * 6;
    const int32_t index = (int32_t)ZTRUNC(sector);
    const ZT f = sector - (ZT)index;
    const ZT p = val * (1 - sat), q = val * (1 - (sat * f)), t = val * (1 - sat * (1 - f));
    const ZT colors[6][3] = {{val, t, p}, {q, val, p}, {p, val, t}, {p, q, val}, {t, p, val}, {val, p, q}};
    const int idx = ((index % 6) + 6) % 6;
    for (int i = 0; i < 3; ++i) out[i] = colors[idx][i];
}
static void ZN(hsv_to_hsl)(const ZT hsv[3], ZT out[3]) { /* color.zig:1211-1224 */
    const ZT s_v = hsv[1] / ZC(100.0), v = hsv[2] / ZC(100.0);
    const ZT l = v * (ZC(1.0) - s_v / ZC(2.0));
    const ZT s_l = (l == 0 || l == 1) ? 0 : (v - l) / ZN(minv)(l, 1 - l);
    out[0] = hsv[0]; out[1] = s_l * ZC(100.0); out[2] = l * ZC(100.0);
}
static void ZN(hsl_to_hsv)(const ZT hsl[3], ZT out[3]) { /* color.zig:1227-1240 */
    const ZT s_l = hsl[1] / ZC(100.0), l = hsl[2] / ZC(100.0);
    const ZT v = l + s_l * ZN(minv)(l, 1 - l);
    const ZT s_v = v == 0 ? 0 : ZC(2.0) * (ZC(1.0) - l / v);
    out[0] = hsl[0]; out[1] = s_v * ZC(100.0); out[2] = v * ZC(100.0);
}

static ZT ZN(rgb_to_gray)(const ZT rgb[3]) { /* color.zig:1043-1046 */
    const ZT y = ZC(0.2126) * rgb[0] + ZC(0.7152) * rgb[1] + ZC(0.0722) * rgb[2];
    return ZN(clampv)(y, 0, 1);
}
static void ZN(rgb_to_ycbcr)(const ZT rgb[3], ZT out[3]) { /* color.zig:1010-1017 */
    const ZT y = ZN(clampv)(ZC(0.299) * rgb[0] + ZC(0.587) * rgb[1] + ZC(0.114) * rgb[2], 0, 1);
    out[0] = y;
    out[1] = ZN(clampv)((rgb[2] - y) / ZC(1.772), ZC(-0.5), ZC(0.5));
    out[2] = ZN(clampv)((rgb[0] - y) / ZC(1.402), ZC(-0.5), ZC(0.5));
}
static void ZN(ycbcr_to_rgb)(const ZT in[3], ZT out[3]) { /* color.zig:1070-1083 */
    const ZT y = in[0], cb = in[1], cr = in[2];
    out[0] = ZN(clampv)(y + ZC(1.402) * cr, 0, 1);
    out[1] = ZN(clampv)(y - ZC(0.344136) * cb - ZC(0.714136) * cr, 0, 1);
    out[2] = ZN(clampv)(y + ZC(1.772) * cb, 0, 1);
}

/* <Space>(T).to(target): the dispatch tables at color.zig:350-361 (Rgb), 475-480 (Rgba), 533-537 (Gray), 586-592 (Hsv),
 * 624-630 (Hsl), 667-678 (Xyz), 711-717 (Lab), 750-755 (Lch), 786-791 (Lms), 824-830 (Oklab), 863-868 (Oklch),
 * 902-908 (Xyb), 943-948 (Ycbcr). in/out hold the struct's fields in declaration order (Gray: 1, Rgba: 4, else 3). */
static void ZN(cs_to)(int from, const ZT in[4], int to, ZT out[4]) {
    ZT t[4] = {0, 0, 0, 0};
    if (from == to) { for (int i = 0; i < 4; ++i) out[i] = in[i]; return; }
    switch (from) {
    case ZO_CS_RGB:
        switch (to) {
        case ZO_CS_GRAY: out[0] = ZN(rgb_to_gray)(in); return;
        case ZO_CS_HSL: ZN(rgb_to_hsl)(in, out); return;
        case ZO_CS_HSV: ZN(rgb_to_hsv)(in, out); return;
        case ZO_CS_RGBA: out[0] = in[0]; out[1] = in[1]; out[2] = in[2]; out[3] = ZC(1.0); return;
        case ZO_CS_XYB: ZN(rgb_to_xyb)(in, out); return;
        case ZO_CS_XYZ: ZN(rgb_to_xyz)(in, out); return;
        case ZO_CS_YCBCR: ZN(rgb_to_ycbcr)(in, out); return;
        default: ZN(rgb_to_xyz)(in, t); ZN(cs_to)(ZO_CS_XYZ, t, to, out); return;
        }
    case ZO_CS_RGBA:
        t[0] = in[0]; t[1] = in[1]; t[2] = in[2];
        ZN(cs_to)(ZO_CS_RGB, t, to, out);
        return;
    case ZO_CS_GRAY:
        t[0] = t[1] = t[2] = in[0];
        ZN(cs_to)(ZO_CS_RGB, t, to, out);
        return;
    case ZO_CS_HSV:
        if (to == ZO_CS_HSL) { ZN(hsv_to_hsl)(in, out); return; }
        ZN(hsv_to_rgb)(in, t);
        ZN(cs_to)(ZO_CS_RGB, t, to, out);
        return;
    case ZO_CS_HSL:
        if (to == ZO_CS_HSV) { ZN(hsl_to_hsv)(in, out); return; }
        ZN(hsl_to_rgb)(in, t);
        ZN(cs_to)(ZO_CS_RGB, t, to, out);
        return;
    case ZO_CS_XYZ:
        switch (to) {
        case ZO_CS_LAB: ZN(xyz_to_lab)(in, out); return;
        case ZO_CS_LCH: ZN(xyz_to_lab)(in, t); ZN(cs_to)(ZO_CS_LAB, t, ZO_CS_LCH, out); return;
        case ZO_CS_LMS: ZN(xyz_to_lms)(in, out); return;
        case ZO_CS_OKLAB: ZN(xyz_to_oklab)(in, out); return;
        case ZO_CS_OKLCH: ZN(xyz_to_oklab)(in, t); ZN(cs_to)(ZO_CS_OKLAB, t, ZO_CS_OKLCH, out); return;
        case ZO_CS_RGB: ZN(xyz_to_rgb)(in, out); return;
        case ZO_CS_XYB: ZN(xyz_to_xyb)(in, out); return;
        default: ZN(xyz_to_rgb)(in, t); ZN(cs_to)(ZO_CS_RGB, t, to, out); return;
        }
    case ZO_CS_LAB:
        if (to == ZO_CS_LCH) { out[0] = in[0]; ZN(cart_to_cyl)(in[1], in[2], &out[1], &out[2]); return; }
        ZN(lab_to_xyz)(in, t);
        ZN(cs_to)(ZO_CS_XYZ, t, to, out);
        return;
    case ZO_CS_LCH:
        t[0] = in[0];
        ZN(cyl_to_cart)(in[1], in[2], &t[1], &t[2]);
        ZN(cs_to)(ZO_CS_LAB, t, to, out);
        return;
    case ZO_CS_LMS:
        ZN(lms_to_xyz)(in, t);
        ZN(cs_to)(ZO_CS_XYZ, t, to, out);
        return;
    case ZO_CS_OKLAB:
        if (to == ZO_CS_OKLCH) { out[0] = in[0]; ZN(cart_to_cyl)(in[1], in[2], &out[1], &out[2]); return; }
        ZN(oklab_to_xyz)(in, t);
        ZN(cs_to)(ZO_CS_XYZ, t, to, out);
        return;
    case ZO_CS_OKLCH:
        t[0] = in[0];
        ZN(cyl_to_cart)(in[1], in[2], &t[1], &t[2]);
        ZN(cs_to)(ZO_CS_OKLAB, t, to, out);
        return;
    case ZO_CS_XYB:
        if (to == ZO_CS_RGB) { ZN(xyb_to_rgb)(in, out); return; }
        ZN(xyb_to_xyz)(in, t);
        ZN(cs_to)(ZO_CS_XYZ, t, to, out);
        return;
    case ZO_CS_YCBCR:
        ZN(ycbcr_to_rgb)(in, t);
        ZN(cs_to)(ZO_CS_RGB, t, to, out);
        return;
    default:
        for (int i = 0; i < 4; ++i) out[i] = in[i];
    }
}
