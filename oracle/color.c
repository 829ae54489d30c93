/*
 * oracle/color.c — restatement of Image.convertInto / convertColor for the colour spaces on the image
 * hot path. TEST INFRASTRUCTURE ONLY (zo.h).
 *   src/image.zig:396-407            convertInto (per-pixel convertColor)
 *   src/color.zig:108-151            convertColor dispatch (scalar<->scalar, scalar<->colour, colour<->colour)
 *   src/color.zig:365-390,484-512    Rgb/Rgba .as (u8 -> /255 ; float -> @round(255 * clamp(v,0,1)))
 *   src/color.zig:987-1009           rgbToYcbcr (u8: BT.601 16.16 fixed point)
 *   src/color.zig:1031-1047          rgbToGray (u8: BT.709 16.16 fixed point; float: clamp(dot, 0, 1))
 *   src/color.zig:1252-1272          gammaToLinear, rgbToXyz
 *   src/color.zig:1381-1400          xyzToOklab
 * Oklab/XYZ forward values flow through Zig's std.math.pow / cbrt: PARITY UNPINNED at the last ulp
 * (the reference has no forward golden values for them either, only round trips: color.zig:1738-1773).
 * The other colour spaces (Hsl, Hsv, Lab, Lch, Lms, Oklch, Xyb, float Ycbcr) and every back-conversion go through
 * colorspaces.c, whose f64 instance IS pinned to the reference's golden values.
 */
#include "zo.h"
#include <math.h>
#include <string.h>

static float gamma_to_linear(float c) { /* color.zig:1252-1258 */
    return c > 0.04045f ? zo_powf((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f;
}

void zo_srgb_to_linear_lut(float lut[256]) {
    for (int i = 0; i < 256; ++i) lut[i] = gamma_to_linear((float)i / 255);
}

static void linear_to_xyz(float r, float g, float b, float xyz[3]) { /* color.zig:1261-1272 after gammaToLinear */
    xyz[0] = (r * 0.4124f + g * 0.3576f + b * 0.1805f) * 100;
    xyz[1] = (r * 0.2126f + g * 0.7152f + b * 0.0722f) * 100;
    xyz[2] = (r * 0.0193f + g * 0.1192f + b * 0.9505f) * 100;
}

static void xyz_to_oklab(const float xyz[3], float lab[3]) { /* color.zig:1381-1400 */
    const float x = xyz[0] / 100.0f, y = xyz[1] / 100.0f, z = xyz[2] / 100.0f;
    const float l_linear = 0.8189330101f * x + 0.3618667424f * y - 0.1288597137f * z;
    const float m_linear = 0.0329845436f * x + 0.9293118715f * y + 0.0361456387f * z;
    const float s_linear = 0.0482003018f * x + 0.2643662691f * y + 0.6338517070f * z;
    const float l_dash = zo_cbrtf(l_linear), m_dash = zo_cbrtf(m_linear), s_dash = zo_cbrtf(s_linear);
    lab[0] = 0.2104542553f * l_dash + 0.7936177850f * m_dash - 0.0040720468f * s_dash;
    lab[1] = 1.9779984951f * l_dash - 2.4285922050f * m_dash + 0.4505937099f * s_dash;
    lab[2] = 0.0259040371f * l_dash + 0.7827717662f * m_dash - 0.8086757660f * s_dash;
}

static uint8_t rgb_to_gray_u8(int32_t r, int32_t g, int32_t b) { /* color.zig:1031-1042 */
    const int32_t yr = 13933, yg = 46871, yb = 4732; /* @round(0.2126 / 0.7152 / 0.0722 * 65536) */
    int32_t v = (yr * r + yg * g + yb * b + 32768) >> 16;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
static float clamp01(float v) { return v < 0 ? 0 : (v > 1 ? 1 : v); }
static float rgb_to_gray_f32(float r, float g, float b) { /* color.zig:1043-1046 */
    const float y = 0.2126f * r + 0.7152f * g + 0.0722f * b;
    return clamp01(y);
}
static uint8_t unit_to_u8(float v) { return (uint8_t)roundf(255 * clamp01(v)); } /* @round(255 * clamp(v, 0, 1)) */

static void rgb_to_ycbcr_u8(int32_t r, int32_t g, int32_t b, uint8_t out[3]) { /* color.zig:987-1009 */
    const int64_t y = (19595LL * r + 38470LL * g + 7471LL * b + 32768) >> 16;
    const int64_t cb = ((-11059LL * r + -21710LL * g + 32768LL * b + 32768) >> 16) + 128;
    const int64_t cr = ((32768LL * r + -27439LL * g + -5329LL * b + 32768) >> 16) + 128;
    out[0] = zo_clamp_u8_i64(y); out[1] = zo_clamp_u8_i64(cb); out[2] = zo_clamp_u8_i64(cr);
}

static int convert_legacy(const zo_image *src, int src_space, const zo_image *dst, int dst_space, const float *srgb_lut) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    float own_lut[256];
    if (!srgb_lut) { zo_srgb_to_linear_lut(own_lut); srgb_lut = own_lut; }
    const int sf = zo_is_float(src->pixel), df = zo_is_float(dst->pixel);
    const int sch = zo_channels(src->pixel), dch = zo_channels(dst->pixel);
    const size_t sps = zo_pixel_size(src->pixel), dps = zo_pixel_size(dst->pixel);
    /* layout / space consistency */
    if (src_space != ZO_CS_GRAY && src_space != ZO_CS_RGB && src_space != ZO_CS_RGBA) return 5;
    if (sch != (src_space == ZO_CS_GRAY ? 1 : (src_space == ZO_CS_RGBA ? 4 : 3))) return 2;
    if (dch != (dst_space == ZO_CS_GRAY ? 1 : (dst_space == ZO_CS_RGBA ? 4 : 3))) return 2;

    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c) {
            const void *sp = (const char *)src->data + (r * src->stride + c) * sps;
            void *dp = (char *)dst->data + (r * dst->stride + c) * dps;
            /* gather source as (r,g,b,a) in its own element type */
            uint8_t su[4] = {0, 0, 0, 255};
            float sfl[4] = {0, 0, 0, 1.0f};
            if (sf) { for (int i = 0; i < sch; ++i) sfl[i] = ((const float *)sp)[i]; if (sch == 1) sfl[1] = sfl[2] = sfl[0]; }
            else { for (int i = 0; i < sch; ++i) su[i] = ((const uint8_t *)sp)[i]; if (sch == 1) su[1] = su[2] = su[0]; }

            switch (dst_space) {
            case ZO_CS_GRAY:
                if (src_space == ZO_CS_GRAY) { /* scalar <-> scalar (color.zig:113-119) */
                    if (!sf && !df) ((uint8_t *)dp)[0] = su[0];
                    else if (!sf && df) ((float *)dp)[0] = (float)su[0] / 255.0f;
                    else if (sf && !df) {
                        double v = (double)sfl[0];
                        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                        ((uint8_t *)dp)[0] = (uint8_t)round(v * 255.0);
                    } else ((float *)dp)[0] = sfl[0];
                } else if (!sf) { /* colour(u8) -> luminance: fixed-point gray, then .as(Dest) */
                    const uint8_t y = rgb_to_gray_u8(su[0], su[1], su[2]);
                    if (df) ((float *)dp)[0] = (float)y / 255; else ((uint8_t *)dp)[0] = y;
                } else {
                    const float y = rgb_to_gray_f32(sfl[0], sfl[1], sfl[2]);
                    if (df) ((float *)dp)[0] = y; else ((uint8_t *)dp)[0] = unit_to_u8(y);
                }
                break;
            case ZO_CS_RGB:
            case ZO_CS_RGBA: {
                if (df) { /* destination floats: source.as(f32) first (u8 -> /255), gray replicated, alpha 1 */
                    float o[4];
                    if (sf) memcpy(o, sfl, sizeof o);
                    else for (int i = 0; i < 4; ++i) o[i] = (float)su[i] / 255;
                    if (!sf && sch != 4) o[3] = 1.0f;
                    for (int i = 0; i < dch; ++i) ((float *)dp)[i] = o[i];
                } else {
                    uint8_t o[4];
                    if (!sf) memcpy(o, su, 4);
                    else { for (int i = 0; i < 4; ++i) o[i] = unit_to_u8(sfl[i]); if (sch != 4) o[3] = 255; }
                    for (int i = 0; i < dch; ++i) ((uint8_t *)dp)[i] = o[i];
                }
                break;
            }
            case ZO_CS_XYZ:
            case ZO_CS_OKLAB: {
                if (!df) return 5;
                float lin[3], xyz[3];
                if (!sf) for (int i = 0; i < 3; ++i) lin[i] = srgb_lut[su[i]];           /* gammaToLinear(u8 / 255) */
                else for (int i = 0; i < 3; ++i) lin[i] = gamma_to_linear(sfl[i]);
                linear_to_xyz(lin[0], lin[1], lin[2], xyz);
                if (dst_space == ZO_CS_XYZ) memcpy(dp, xyz, 12);
                else { float lab[3]; xyz_to_oklab(xyz, lab); memcpy(dp, lab, 12); }
                break;
            }
            case ZO_CS_YCBCR:
                if (sf || df) return 5;
                rgb_to_ycbcr_u8(su[0], su[1], su[2], (uint8_t *)dp);
                break;
            default: return 5;
            }
        }
    return 0;
}

/* ---- the general case: any colour space on either side ---------------------------------------- */
static int space_channels(int space) { return space == ZO_CS_GRAY ? 1 : (space == ZO_CS_RGBA ? 4 : 3); }
static int space_has_u8(int space) { return space == ZO_CS_GRAY || space == ZO_CS_RGB || space == ZO_CS_RGBA || space == ZO_CS_YCBCR; }

/* <Space>(u8).to(target) among the u8-backed types (color.zig:350-361, 475-480, 533-537, 943-948) */
static void u8_to(int from, const uint8_t in[4], int to, uint8_t out[4]) {
    uint8_t rgb[4] = {0, 0, 0, 255};
    if (from == to) { memcpy(out, in, 4); return; }
    switch (from) {
    case ZO_CS_GRAY: rgb[0] = rgb[1] = rgb[2] = in[0]; break;                           /* grayToRgb :1050 */
    case ZO_CS_RGB: case ZO_CS_RGBA: rgb[0] = in[0]; rgb[1] = in[1]; rgb[2] = in[2]; break;
    case ZO_CS_YCBCR: {                                                                  /* ycbcrToRgb :1057-1068 */
        const int64_t y = in[0], cb = (int64_t)in[1] - 128, cr = (int64_t)in[2] - 128;
        rgb[0] = zo_clamp_u8_i64((65536 * y + 91881 * cr + 32768) >> 16);
        rgb[1] = zo_clamp_u8_i64((65536 * y - 22554 * cb - 46802 * cr + 32768) >> 16);
        rgb[2] = zo_clamp_u8_i64((65536 * y + 116130 * cb + 32768) >> 16);
        break;
    }
    }
    switch (to) {
    case ZO_CS_GRAY: out[0] = rgb_to_gray_u8(rgb[0], rgb[1], rgb[2]); break;
    case ZO_CS_RGB: out[0] = rgb[0]; out[1] = rgb[1]; out[2] = rgb[2]; break;
    case ZO_CS_RGBA: out[0] = rgb[0]; out[1] = rgb[1]; out[2] = rgb[2]; out[3] = 255; break; /* Rgb(u8).to(.rgba): alpha 255 */
    case ZO_CS_YCBCR: rgb_to_ycbcr_u8(rgb[0], rgb[1], rgb[2], out); break;
    }
}
/* <Space>(u8).as(f32) / <Space>(f32).as(u8): color.zig:365-390 (Rgb), 484-512 (Rgba), 540-558 (Gray), 950-981 (Ycbcr) */
static void u8_as_f32(int space, const uint8_t in[4], float out[4]) {
    const int n = space_channels(space);
    for (int i = 0; i < n; ++i) out[i] = (float)in[i] / 255;
    if (space == ZO_CS_YCBCR) { out[1] = ((float)in[1] - 128) / 255; out[2] = ((float)in[2] - 128) / 255; }
}
static void f32_as_u8(int space, const float in[4], uint8_t out[4]) {
    const int n = space_channels(space);
    for (int i = 0; i < n; ++i) out[i] = unit_to_u8(in[i]);
    if (space == ZO_CS_YCBCR) { out[1] = unit_to_u8(in[1] + 0.5f); out[2] = unit_to_u8(in[2] + 0.5f); }
}

/* convertColor(Dest, source) for one pixel (color.zig:108-151). Scalar images (Image(u8) / Image(f32), space GRAY) are
 * Zig scalars, not Gray structs: the scalar is converted to the destination's component type FIRST (:128-133). */
static int convert_pixel(int src_space, int sf, const uint8_t su[4], const float sfl[4], int dst_space, int df, uint8_t du[4], float dfl[4]) {
    if (dst_space == ZO_CS_GRAY) { /* colour (or scalar) -> scalar: source.to(.gray).as(Dest).y (:136-138) */
        if (src_space == ZO_CS_GRAY) {
            if (!sf && !df) du[0] = su[0];
            else if (!sf && df) dfl[0] = (float)su[0] / 255.0f;
            else if (sf && !df) { double v = (double)sfl[0]; v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); du[0] = (uint8_t)round(v * 255.0); }
            else dfl[0] = sfl[0];
            return 0;
        }
        if (!sf) {
            if (!space_has_u8(src_space)) return 5;
            uint8_t g[4];
            u8_to(src_space, su, ZO_CS_GRAY, g);
            if (df) dfl[0] = (float)g[0] / 255; else du[0] = g[0];
        } else {
            float g[4];
            zo_color_to_f32(src_space, sfl, ZO_CS_GRAY, g);
            if (df) dfl[0] = g[0]; else du[0] = unit_to_u8(g[0]);
        }
        return 0;
    }
    if (src_space == ZO_CS_GRAY) { /* scalar -> colour: Gray(Src){y}.as(DestT).to(space).as(DestT) (:122-133) */
        if (df) {
            float g[4] = {sf ? sfl[0] : (float)su[0] / 255, 0, 0, 0};
            zo_color_to_f32(ZO_CS_GRAY, g, dst_space, dfl);
        } else {
            if (!space_has_u8(dst_space)) return 5;
            uint8_t g[4] = {sf ? unit_to_u8(sfl[0]) : su[0], 0, 0, 0};
            u8_to(ZO_CS_GRAY, g, dst_space, du);
        }
        return 0;
    }
    if (df) { /* colour -> float colour: source.as(f32).to(space) (:146-150) */
        float v[4] = {0, 0, 0, 0};
        if (sf) memcpy(v, sfl, sizeof v);
        else { if (!space_has_u8(src_space)) return 5; u8_as_f32(src_space, su, v); }
        zo_color_to_f32(src_space, v, dst_space, dfl);
        return 0;
    }
    /* colour -> u8 colour: source.to(space).as(u8) (:152) */
    if (!space_has_u8(dst_space)) return 5;
    if (!sf) {
        if (!space_has_u8(src_space)) return 5;
        u8_to(src_space, su, dst_space, du);
    } else {
        float t[4] = {0, 0, 0, 0};
        zo_color_to_f32(src_space, sfl, dst_space, t);
        f32_as_u8(dst_space, t, du);
    }
    return 0;
}

int zo_convert(const zo_image *src, int src_space, const zo_image *dst, int dst_space, const float *srgb_lut) {
    const int legacy_src = src_space == ZO_CS_GRAY || src_space == ZO_CS_RGB || src_space == ZO_CS_RGBA;
    const int float_ycbcr = dst_space == ZO_CS_YCBCR && (zo_is_float(src->pixel) || zo_is_float(dst->pixel));
    if (legacy_src && dst_space <= ZO_CS_YCBCR && !float_ycbcr) return convert_legacy(src, src_space, dst, dst_space, srgb_lut); /* honours srgb_lut */
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src_space < 0 || src_space > ZO_CS_XYB || dst_space < 0 || dst_space > ZO_CS_XYB) return 5;
    const int sf = zo_is_float(src->pixel), df = zo_is_float(dst->pixel);
    const int sch = zo_channels(src->pixel), dch = zo_channels(dst->pixel);
    if (sch != space_channels(src_space) || dch != space_channels(dst_space)) return 2;
    if ((!sf && !space_has_u8(src_space)) || (!df && !space_has_u8(dst_space))) return 5; /* float-only colour types */
    const size_t sps = zo_pixel_size(src->pixel), dps = zo_pixel_size(dst->pixel);
    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c) {
            const void *sp = (const char *)src->data + (r * src->stride + c) * sps;
            void *dp = (char *)dst->data + (r * dst->stride + c) * dps;
            uint8_t su[4] = {0, 0, 0, 0}, du[4] = {0, 0, 0, 0};
            float sfl[4] = {0, 0, 0, 0}, dfl[4] = {0, 0, 0, 0};
            if (sf) memcpy(sfl, sp, (size_t)sch * 4); else memcpy(su, sp, (size_t)sch);
            const int rc = convert_pixel(src_space, sf, su, sfl, dst_space, df, du, dfl);
            if (rc) return rc;
            if (df) memcpy(dp, dfl, (size_t)dch * 4); else memcpy(dp, du, (size_t)dch);
        }
    return 0;
}
