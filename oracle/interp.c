/*
 * oracle/interp.c — restatement of zignal's point sampling and resize. TEST INFRASTRUCTURE ONLY (zo.h).
 *   src/image/interpolation.zig:72-84     interpolate (finite / range guard, method switch)
 *   src/image/interpolation.zig:89-214    resize, resizeGeneric
 *   src/image/interpolation.zig:222-300   bicubic / Catmull-Rom / Lanczos3 LUT / Mitchell kernels
 *   src/image/interpolation.zig:306-519   interpolateNearest / Bilinear / WithKernel
 *   src/image/channel_ops.zig:144-493     resizePlane{Bilinear,Nearest,Bicubic,CatmullRom,Mitchell,Lanczos}U8
 *   src/image/transforms.zig:49-108       letterbox
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int zo_copy(const zo_image *src, const zo_image *dst);
int zo_set_border(const zo_image *img, const uint32_t rect[4], const void *pixel);

/* ---- kernels (interpolation.zig:222-300) ------------------------------------------------- */
static float bicubic_kernel(float t) {
    const float at = fabsf(t);
    if (at <= 1) return 1 - 2 * at * at + at * at * at;
    if (at <= 2) return 4 - 8 * at + 5 * at * at - at * at * at;
    return 0;
}
static float catmull_rom_kernel(float x) {
    const float ax = fabsf(x);
    if (ax <= 1) return 1.5f * ax * ax * ax - 2.5f * ax * ax + 1;
    if (ax <= 2) return -0.5f * ax * ax * ax + 2.5f * ax * ax - 4 * ax + 2;
    return 0;
}
static float mitchell_kernel(float x, float m_b, float m_c) {
    const float ax = fabsf(x), ax2 = ax * ax, ax3 = ax2 * ax;
    if (ax < 1)
        return ((12 - 9 * m_b - 6 * m_c) * ax3 + (-18 + 12 * m_b + 6 * m_c) * ax2 + (6 - 2 * m_b)) / 6;
    if (ax < 2)
        return ((-m_b - 6 * m_c) * ax3 + (6 * m_b + 30 * m_c) * ax2 + (-12 * m_b - 48 * m_c) * ax + (8 * m_b + 24 * m_c)) / 6;
    return 0;
}
static const float ZIG_PI_F32 = 3.14159265358979323846f;
static float lanczos_kernel(float x, float a) { /* interpolation.zig:246-253 */
    if (x == 0) return 1;
    if (fabsf(x) >= a) return 0;
    const float pi_x = ZIG_PI_F32 * x;
    const float pi_x_over_a = pi_x / a;
    return (a * zo_sinf(pi_x) * zo_sinf(pi_x_over_a)) / (pi_x * pi_x);
}
static float g_lanczos_lut[1025];
static int g_lanczos_ready = 0;
const float *zo_lanczos3_lut(void) { /* interpolation.zig:256-267 (built at comptime in the reference) */
    if (!g_lanczos_ready) {
        const float step = 1024.0f / 3.0f;
        for (int i = 0; i < 1025; ++i) g_lanczos_lut[i] = lanczos_kernel((float)i / step, 3.0f);
        g_lanczos_ready = 1;
    }
    return g_lanczos_lut;
}
static float lanczos3_kernel_lut(const float *lut, float x) { /* interpolation.zig:270-281 */
    const float ax = fabsf(x);
    if (ax >= 3.0f) return 0;
    const float step = (float)(1024.0 / 3.0);
    const float pos = ax * step;
    const size_t idx = (size_t)truncf(pos);
    const float frac = pos - (float)idx;
    return lut[idx] * (1.0f - frac) + lut[idx + 1] * frac;
}

/* ---- pixel access ------------------------------------------------------------------------ */
static inline const void *px_ptr(const zo_image *img, size_t r, size_t c) {
    return (const char *)img->data + (r * img->stride + c) * zo_pixel_size(img->pixel);
}
static inline float ch_as_f32(const zo_image *img, const void *p, int ch) {
    return zo_is_float(img->pixel) ? ((const float *)p)[ch] : (float)((const uint8_t *)p)[ch];
}

static int interp_nearest(const zo_image *img, float x, float y, int border, void *out) {
    const int64_t col = zo_resolve_index((int64_t)roundf(x), img->cols, border);
    if (col < 0) return 0;
    const int64_t row = zo_resolve_index((int64_t)roundf(y), img->rows, border);
    if (row < 0) return 0;
    memcpy(out, px_ptr(img, (size_t)row, (size_t)col), zo_pixel_size(img->pixel));
    return 1;
}

static int interp_bilinear(const zo_image *img, float x, float y, int border, void *out) {
    const int64_t left = (int64_t)floorf(x), top = (int64_t)floorf(y);
    const int64_t r0 = zo_resolve_index(top, img->rows, border), r1 = zo_resolve_index(top + 1, img->rows, border);
    const int64_t c0 = zo_resolve_index(left, img->cols, border), c1 = zo_resolve_index(left + 1, img->cols, border);
    if (border == ZO_MIRROR && (r0 < 0 || r1 < 0 || c0 < 0 || c1 < 0)) return 0;
    const int nch = zo_channels(img->pixel);
    const size_t ps = zo_pixel_size(img->pixel);
    char zero[16] = {0}, tl[16], tr[16], bl[16], br[16];
    memcpy(tl, (r0 >= 0 && c0 >= 0) ? px_ptr(img, r0, c0) : zero, ps);
    memcpy(tr, (r0 >= 0 && c1 >= 0) ? px_ptr(img, r0, c1) : zero, ps);
    memcpy(bl, (r1 >= 0 && c0 >= 0) ? px_ptr(img, r1, c0) : zero, ps);
    memcpy(br, (r1 >= 0 && c1 >= 0) ? px_ptr(img, r1, c1) : zero, ps);
    const float lr = x - (float)left, tb = y - (float)top;
    const int32_t fx = (int32_t)roundf(lr * 256), fy = (int32_t)roundf(tb * 256);
    for (int ch = 0; ch < nch; ++ch) {
        if (zo_is_float(img->pixel)) { /* lerpFloat */
            const float p_tl = ((float *)tl)[ch], p_tr = ((float *)tr)[ch], p_bl = ((float *)bl)[ch], p_br = ((float *)br)[ch];
            ((float *)out)[ch] = (1 - tb) * ((1 - lr) * p_tl + lr * p_tr) + tb * ((1 - lr) * p_bl + lr * p_br);
        } else { /* lerpInt, i32 intermediates for <= 8 bit */
            const int32_t p_tl = ((uint8_t *)tl)[ch], p_tr = ((uint8_t *)tr)[ch], p_bl = ((uint8_t *)bl)[ch], p_br = ((uint8_t *)br)[ch];
            const int32_t top_val = p_tl * (256 - fx) + p_tr * fx;
            const int32_t bottom_val = p_bl * (256 - fx) + p_br * fx;
            const int32_t result = (top_val * (256 - fy) + bottom_val * fy + 32768) / 65536;
            ((uint8_t *)out)[ch] = zo_clamp_u8_i64(result);
        }
    }
    return 1;
}

static float eval_kernel(const zo_method *m, const float *lut, float t) {
    switch (m->kind) {
    case ZO_BICUBIC: return bicubic_kernel(t);
    case ZO_CATMULL_ROM: return catmull_rom_kernel(t);
    case ZO_MITCHELL: return mitchell_kernel(t, m->b, m->c);
    default: return lanczos3_kernel_lut(lut, t);
    }
}

/* interpolateWithKernel (interpolation.zig:426-519) */
static int interp_kernel(const zo_image *img, float x, float y, const zo_method *m, int border, void *out) {
    const int radius = m->kind == ZO_LANCZOS ? 3 : 2, window = 2 * radius;
    const float *lut = m->lanczos_lut ? m->lanczos_lut : zo_lanczos3_lut();
    const int64_t ix = (int64_t)floorf(x), iy = (int64_t)floorf(y);
    const float fx = x - (float)ix, fy = y - (float)iy;
    float xw[6], yw[6];
    for (int i = 0; i < window; ++i) {
        xw[i] = eval_kernel(m, lut, (float)(i - (radius - 1)) - fx);
        yw[i] = eval_kernel(m, lut, (float)(i - (radius - 1)) - fy);
    }
    const int nch = zo_channels(img->pixel);
    float sums[4] = {0, 0, 0, 0}, weight_sum = 0;
    for (int j = 0; j < window; ++j) {
        const int64_t py = zo_resolve_index(iy - (radius - 1) + j, img->rows, border);
        if (py < 0) continue;
        for (int i = 0; i < window; ++i) {
            const int64_t pxi = zo_resolve_index(ix - (radius - 1) + i, img->cols, border);
            if (pxi < 0) continue;
            const void *p = px_ptr(img, (size_t)py, (size_t)pxi);
            const float weight = xw[i] * yw[j];
            for (int ch = 0; ch < nch; ++ch) sums[ch] += ch_as_f32(img, p, ch) * weight;
            weight_sum += weight;
        }
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float val = weight_sum != 0 ? sums[ch] / weight_sum : 0;
        if (zo_is_float(img->pixel)) ((float *)out)[ch] = val;
        else ((uint8_t *)out)[ch] = zo_clamp_u8_f32(val);
    }
    return 1;
}

int zo_interpolate(const zo_image *img, float x, float y, const zo_method *m, int border, void *out) {
    if (!isfinite(x) || !isfinite(y)) return 0;
    const float range_limit = (float)(INT64_MAX / 2);
    if (fabsf(x) > range_limit || fabsf(y) > range_limit) return 0;
    switch (m->kind) {
    case ZO_NEAREST: return interp_nearest(img, x, y, border, out);
    case ZO_BILINEAR: return interp_bilinear(img, x, y, border, out);
    default: return interp_kernel(img, x, y, m, border, out);
    }
}

/* ---- u8 plane resizers (channel_ops.zig:144-493) ------------------------------------------- */
static void plane_bilinear(const uint8_t *src, uint8_t *dst, uint32_t sr, uint32_t sc, uint32_t dr, uint32_t dc) {
    const float x_ratio = (float)sc / (float)dc, y_ratio = (float)sr / (float)dr;
    for (uint32_t r = 0; r < dr; ++r) {
        const float syf = ((float)r + 0.5f) * y_ratio - 0.5f;
        const int64_t syi = (int64_t)floorf(syf);
        const int32_t fy = (int32_t)truncf((syf - floorf(syf)) * 256.0f);
        const size_t y0 = (size_t)zo_resolve_index(syi, sr, ZO_MIRROR), y1 = (size_t)zo_resolve_index(syi + 1, sr, ZO_MIRROR);
        for (uint32_t c = 0; c < dc; ++c) {
            const float sxf = ((float)c + 0.5f) * x_ratio - 0.5f;
            const int64_t sxi = (int64_t)floorf(sxf);
            const int32_t fx = (int32_t)truncf((sxf - floorf(sxf)) * 256.0f);
            const size_t x0 = (size_t)zo_resolve_index(sxi, sc, ZO_MIRROR), x1 = (size_t)zo_resolve_index(sxi + 1, sc, ZO_MIRROR);
            const int32_t tl = src[y0 * sc + x0], tr = src[y0 * sc + x1], bl = src[y1 * sc + x0], br = src[y1 * sc + x1];
            const int32_t top = tl * (256 - fx) + tr * fx, bottom = bl * (256 - fx) + br * fx;
            const int32_t result = (top * (256 - fy) + bottom * fy) / 65536; /* no rounding offset */
            dst[(size_t)r * dc + c] = zo_clamp_u8_i64(result);
        }
    }
}
static void plane_nearest(const uint8_t *src, uint8_t *dst, uint32_t sr, uint32_t sc, uint32_t dr, uint32_t dc) {
    const float x_ratio = (float)sc / (float)dc, y_ratio = (float)sr / (float)dr;
    for (uint32_t r = 0; r < dr; ++r) {
        const float syf = ((float)r + 0.5f) * y_ratio - 0.5f;
        uint32_t sy = (uint32_t)roundf(syf);
        if (sy > sr - 1) sy = sr - 1;
        for (uint32_t c = 0; c < dc; ++c) {
            const float sxf = ((float)c + 0.5f) * x_ratio - 0.5f;
            uint32_t sx = (uint32_t)roundf(sxf);
            if (sx > sc - 1) sx = sc - 1;
            dst[(size_t)r * dc + c] = src[(size_t)sy * sc + sx];
        }
    }
}
static int32_t k_bicubic_i(int32_t t) {
    const int32_t at = t < 0 ? -t : t;
    if (at <= 256) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 256 - 2 * t2 + t3; }
    if (at <= 512) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 4 * 256 - 8 * at + 5 * t2 - t3; }
    return 0;
}
static int32_t k_catmull_i(int32_t t) {
    const int32_t at = t < 0 ? -t : t;
    if (at <= 256) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 256 - (5 * t2) / 2 + (3 * t3) / 2; }
    if (at <= 512) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 2 * 256 - 4 * at + (5 * t2) / 2 - t3 / 2; }
    return 0;
}
static int32_t k_mitchell_i(int32_t t) {
    const int64_t at = t < 0 ? -(int64_t)t : t, s = 256, s2 = s * s, s3 = s2 * s;
    if (at < s) { const int64_t at2 = at * at, at3 = at2 * at; return (int32_t)((21 * at3 - 36 * at2 * s + 16 * s3) / (18 * s2)); }
    if (at < 2 * s) { const int64_t at2 = at * at, at3 = at2 * at; return (int32_t)((-7 * at3 + 36 * at2 * s - 60 * at * s2 + 32 * s3) / (18 * s2)); }
    return 0;
}
static void plane_cubic_int(const uint8_t *src, uint8_t *dst, uint32_t sr, uint32_t sc, uint32_t dr, uint32_t dc,
                            int32_t (*K)(int32_t)) {
    const float x_ratio = (float)sc / (float)dc, y_ratio = (float)sr / (float)dr;
    for (uint32_t r = 0; r < dr; ++r) {
        const float syf = ((float)r + 0.5f) * y_ratio - 0.5f;
        const int64_t sy = (int64_t)floorf(syf);
        const int32_t fy = (int32_t)truncf((syf - floorf(syf)) * 256.0f);
        for (uint32_t c = 0; c < dc; ++c) {
            const float sxf = ((float)c + 0.5f) * x_ratio - 0.5f;
            const int64_t sx = (int64_t)floorf(sxf);
            const int32_t fx = (int32_t)truncf((sxf - floorf(sxf)) * 256.0f);
            int32_t sum = 0, weight_sum = 0;
            for (int ky = 0; ky < 4; ++ky) {
                const size_t py = (size_t)zo_resolve_index(sy + ky - 1, sr, ZO_MIRROR);
                const int32_t wy = K(ky * 256 - 256 - fy);
                for (int kx = 0; kx < 4; ++kx) {
                    const size_t pxi = (size_t)zo_resolve_index(sx + kx - 1, sc, ZO_MIRROR);
                    const int32_t wx = K(kx * 256 - 256 - fx);
                    const int32_t w = (wx * wy) / 256;
                    sum += (int32_t)src[py * sc + pxi] * w;
                    weight_sum += w;
                }
            }
            const int32_t result = weight_sum != 0 ? sum / weight_sum : 0;
            dst[(size_t)r * dc + c] = zo_clamp_u8_i64(result);
        }
    }
}
static float k_lanczos_plane(float x) { /* channel_ops.zig:446-454, runtime @sin */
    if (x == 0) return 1.0f;
    const float a = 3.0f;
    if (fabsf(x) >= a) return 0.0f;
    const float pi_x = ZIG_PI_F32 * x;
    return (a * zo_sinf(pi_x) * zo_sinf(pi_x / a)) / (pi_x * pi_x);
}
static void plane_lanczos(const uint8_t *src, uint8_t *dst, uint32_t sr, uint32_t sc, uint32_t dr, uint32_t dc) {
    const float x_ratio = (float)sc / (float)dc, y_ratio = (float)sr / (float)dr;
    for (uint32_t r = 0; r < dr; ++r) {
        const float syf = ((float)r + 0.5f) * y_ratio - 0.5f;
        const int64_t sy = (int64_t)floorf(syf);
        const float fy = syf - floorf(syf);
        for (uint32_t c = 0; c < dc; ++c) {
            const float sxf = ((float)c + 0.5f) * x_ratio - 0.5f;
            const int64_t sx = (int64_t)floorf(sxf);
            const float fx = sxf - floorf(sxf);
            float sum = 0, weight_sum = 0;
            for (int ky = 0; ky < 6; ++ky) {
                const size_t py = (size_t)zo_resolve_index(sy + ky - 2, sr, ZO_MIRROR);
                const float wy = k_lanczos_plane((float)(ky - 2) - fy);
                for (int kx = 0; kx < 6; ++kx) {
                    const size_t pxi = (size_t)zo_resolve_index(sx + kx - 2, sc, ZO_MIRROR);
                    const float wx = k_lanczos_plane((float)(kx - 2) - fx);
                    const float w = wx * wy;
                    sum += (float)src[py * sc + pxi] * w;
                    weight_sum += w;
                }
            }
            const float result = weight_sum != 0 ? sum / weight_sum : 0;
            dst[(size_t)r * dc + c] = zo_clamp_u8_f32(result);
        }
    }
}

/* resizeGeneric (interpolation.zig:194-214) */
static void resize_generic(const zo_image *src, const zo_image *dst, const zo_method *m) {
    const float scale_x = (float)src->cols / (float)dst->cols, scale_y = (float)src->rows / (float)dst->rows;
    const size_t ps = zo_pixel_size(dst->pixel);
    for (size_t r = 0; r < dst->rows; ++r) {
        const float src_y = ((float)r + 0.5f) * scale_y - 0.5f;
        for (size_t c = 0; c < dst->cols; ++c) {
            const float src_x = ((float)c + 0.5f) * scale_x - 0.5f;
            char px[16] = {0};
            if (!zo_interpolate(src, src_x, src_y, m, ZO_MIRROR, px)) memset(px, 0, sizeof px);
            memcpy((char *)dst->data + (r * dst->stride + c) * ps, px, ps);
        }
    }
}

int zo_resize(const zo_image *src, const zo_image *dst, const zo_method *m) {
    if (src->pixel != dst->pixel) return 2;
    if (dst->rows == 0 || dst->cols == 0) return 0;
    if (src->rows == dst->rows && src->cols == dst->cols) return zo_copy(src, dst); /* interpolation.zig:91-108 */
    if (src->rows == 0 || src->cols == 0) { resize_generic(src, dst, m); return 0; }
    if (src->pixel == ZO_RGB_U8 || src->pixel == ZO_RGBA_U8) { /* meta.isRgb(T): split, per-plane, merge */
        const int nch = zo_channels(src->pixel);
        const size_t sn = (size_t)src->rows * src->cols, dn = (size_t)dst->rows * dst->cols;
        uint8_t *sp[4] = {0}, *dp[4] = {0};
        for (int i = 0; i < nch; ++i) { sp[i] = (uint8_t *)malloc(sn); dp[i] = (uint8_t *)malloc(dn); }
        size_t idx = 0;
        for (size_t r = 0; r < src->rows; ++r)
            for (size_t c = 0; c < src->cols; ++c, ++idx) {
                const uint8_t *p = (const uint8_t *)src->data + (r * src->stride + c) * (size_t)nch;
                for (int i = 0; i < nch; ++i) sp[i][idx] = p[i];
            }
        for (int i = 0; i < nch; ++i) {
            switch (m->kind) {
            case ZO_NEAREST: plane_nearest(sp[i], dp[i], src->rows, src->cols, dst->rows, dst->cols); break;
            case ZO_BILINEAR: plane_bilinear(sp[i], dp[i], src->rows, src->cols, dst->rows, dst->cols); break;
            case ZO_BICUBIC: plane_cubic_int(sp[i], dp[i], src->rows, src->cols, dst->rows, dst->cols, k_bicubic_i); break;
            case ZO_CATMULL_ROM: plane_cubic_int(sp[i], dp[i], src->rows, src->cols, dst->rows, dst->cols, k_catmull_i); break;
            case ZO_MITCHELL: plane_cubic_int(sp[i], dp[i], src->rows, src->cols, dst->rows, dst->cols, k_mitchell_i); break;
            default: plane_lanczos(sp[i], dp[i], src->rows, src->cols, dst->rows, dst->cols); break;
            }
        }
        idx = 0;
        for (size_t r = 0; r < dst->rows; ++r)
            for (size_t c = 0; c < dst->cols; ++c, ++idx) {
                uint8_t *p = (uint8_t *)dst->data + (r * dst->stride + c) * (size_t)nch;
                for (int i = 0; i < nch; ++i) p[i] = dp[i][idx];
            }
        for (int i = 0; i < nch; ++i) { free(sp[i]); free(dp[i]); }
        return 0;
    }
    resize_generic(src, dst, m);
    return 0;
}

/* letterbox (transforms.zig:49-108) */
int zo_letterbox(const zo_image *src, const zo_image *dst, const zo_method *m, uint32_t rect_out[4]) {
    uint32_t rect[4] = {0, 0, 0, 0};
    const char zero[16] = {0};
    if (dst->rows == 0 || dst->cols == 0) goto done;
    if (src->rows == 0 || src->cols == 0) { zo_fill(dst, zero); goto done; }
    if (src->rows == dst->rows && src->cols == dst->cols) {
        zo_copy(src, dst);
        rect[2] = dst->cols; rect[3] = dst->rows;
        goto done;
    }
    {
        const float rows_scale = (float)dst->rows / (float)src->rows, cols_scale = (float)dst->cols / (float)src->cols;
        if (rows_scale == cols_scale) {
            zo_resize(src, dst, m);
            rect[2] = dst->cols; rect[3] = dst->rows;
            goto done;
        }
        const float aspect = rows_scale < cols_scale ? rows_scale : cols_scale;
        const uint32_t scaled_rows = (uint32_t)roundf(aspect * (float)src->rows);
        const uint32_t scaled_cols = (uint32_t)roundf(aspect * (float)src->cols);
        const uint32_t off_r = (dst->rows > scaled_rows ? dst->rows - scaled_rows : 0) / 2;
        const uint32_t off_c = (dst->cols > scaled_cols ? dst->cols - scaled_cols : 0) / 2;
        rect[0] = off_c; rect[1] = off_r; rect[2] = off_c + scaled_cols; rect[3] = off_r + scaled_rows;
        /* out.view(content_rect): clipped to the image, shares memory */
        uint32_t l = rect[0], t = rect[1], r = rect[2] < dst->cols ? rect[2] : dst->cols, b = rect[3] < dst->rows ? rect[3] : dst->rows;
        if (l < r && t < b) {
            zo_image view = *dst;
            view.rows = b - t; view.cols = r - l;
            view.data = (char *)dst->data + ((size_t)t * dst->stride + l) * zo_pixel_size(dst->pixel);
            zo_resize(src, &view, m);
        }
        zo_set_border(dst, rect, zero);
    }
done:
    if (rect_out) memcpy(rect_out, rect, sizeof rect);
    return 0;
}
