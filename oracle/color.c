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

int zo_convert(const zo_image *src, int src_space, const zo_image *dst, int dst_space, const float *srgb_lut) {
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
