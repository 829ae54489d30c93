/*
 * oracle/transforms.c — restatement of zignal's geometric transforms. TEST INFRASTRUCTURE ONLY (zo.h).
 *   src/geometry/transforms.zig:39-42,147-150,224-231   Similarity / Affine / Projective .project
 *   src/matrix/SMatrix.zig:472-567,162-170              gemm (scalar 3-term dot, left to right), scale
 *   src/image/transforms.zig:112-148    rotateBounds
 *   src/image/transforms.zig:163-212    rotateInto (+ exact 0/90/180/270 paths :385-462)
 *   src/image/transforms.zig:216-282    crop, extract
 *   src/image/transforms.zig:293-378    insert (+ assignPixel src/image.zig:67-94, blendColors src/blending.zig:27-157)
 *   src/image/transforms.zig:465-518    copyRect
 *   src/image/transforms.zig:522-531    warp
 *   src/geometry/transforms.zig:242-263 exact 4-point homography (solved in f64, cast to f32)
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int zo_fill(const zo_image *img, const void *pixel);
int zo_set_border(const zo_image *img, const uint32_t rect[4], const void *pixel);

static inline char *px_at(const zo_image *img, size_t r, size_t c) {
    return (char *)img->data + (r * img->stride + c) * zo_pixel_size(img->pixel);
}

/* gemm with a_cols < vec_len: accumulator = 0; accumulator += a*b per k; result = 0 + 1.0 * accumulator */
void zo_project(int kind, const float *m, float x, float y, float *ox, float *oy) {
    if (kind == ZO_PROJECTIVE) {
        float acc, X, Y, W;
        acc = 0; acc += m[0] * x; acc += m[1] * y; acc += m[2] * 1.0f; X = 0.0f + 1.0f * acc;
        acc = 0; acc += m[3] * x; acc += m[4] * y; acc += m[5] * 1.0f; Y = 0.0f + 1.0f * acc;
        acc = 0; acc += m[6] * x; acc += m[7] * y; acc += m[8] * 1.0f; W = 0.0f + 1.0f * acc;
        if (W != 0) {
            const float inv = 1 / W; /* dst.scale(1 / w): value * item */
            X = inv * X;
            Y = inv * Y;
        }
        *ox = X;
        *oy = Y;
    } else { /* similarity / affine: matrix.dot(src).add(bias); m = {a00,a01,a10,a11,b0,b1} */
        float acc, X, Y;
        acc = 0; acc += m[0] * x; acc += m[1] * y; X = 0.0f + 1.0f * acc;
        acc = 0; acc += m[2] * x; acc += m[3] * y; Y = 0.0f + 1.0f * acc;
        *ox = X + m[4];
        *oy = Y + m[5];
    }
}

int zo_warp(const zo_image *src, const zo_image *dst, int kind, const float *m, const zo_method *method) {
    if (src->pixel != dst->pixel) return 2;
    const size_t ps = zo_pixel_size(dst->pixel);
    for (size_t r = 0; r < dst->rows; ++r)
        for (size_t c = 0; c < dst->cols; ++c) {
            float sx, sy;
            zo_project(kind, m, (float)c, (float)r, &sx, &sy);
            char px[16] = {0};
            if (!zo_interpolate(src, sx, sy, method, ZO_MIRROR, px)) memset(px, 0, sizeof px);
            memcpy(px_at(dst, r, c), px, ps);
        }
    return 0;
}

/* @mod(angle, tau) in f32: floored modulo */
static float mod_tau(float a) {
    const float tau = 6.28318530717958647692f;
    float r = fmodf(a, tau);
    if (r < 0) r += tau; /* fmodf is exact; adding tau once gives the floored result */
    return r;
}
static int orthogonal_case(float angle) { /* 0: 0deg, 1: 90, 2: 180, 3: 270, -1: general */
    const float n = mod_tau(angle), eps = 1e-6f;
    const float pi = 3.14159265358979323846f, tau = 6.28318530717958647692f;
    if (fabsf(n) < eps || fabsf(n - tau) < eps) return 0;
    if (fabsf(n - pi / 2.0f) < eps) return 1;
    if (fabsf(n - pi) < eps) return 2;
    if (fabsf(n - 3.0f * pi / 2.0f) < eps) return 3;
    return -1;
}

int zo_rotate_bounds(uint32_t rows, uint32_t cols, float angle, float cos_a, float sin_a, uint32_t *out_rows, uint32_t *out_cols) {
    switch (orthogonal_case(angle)) {
    case 0: case 2: *out_rows = rows; *out_cols = cols; return 0;
    case 1: case 3: *out_rows = cols; *out_cols = rows; return 0;
    }
    const float cos_abs = fabsf(cos_a), sin_abs = fabsf(sin_a), w = (float)cols, h = (float)rows;
    *out_cols = (uint32_t)ceilf(w * cos_abs + h * sin_abs);
    *out_rows = (uint32_t)ceilf(h * cos_abs + w * sin_abs);
    return 0;
}

/* exact permutations (transforms.zig:385-462); which: 0, 1 (90 CCW), 2, 3 (270 CCW) */
static void rotate_orthogonal(const zo_image *src, const zo_image *out, int which) {
    const size_t ps = zo_pixel_size(src->pixel);
    const uint32_t rr = (which & 1) ? src->cols : src->rows, rc = (which & 1) ? src->rows : src->cols;
    const uint32_t off_r = (out->rows > rr ? out->rows - rr : 0) / 2, off_c = (out->cols > rc ? out->cols - rc : 0) / 2;
    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c) {
            size_t nr, nc;
            switch (which) {
            case 0: nr = r; nc = c; break;
            case 1: nr = src->cols - 1 - c; nc = r; break;
            case 2: nr = src->rows - 1 - r; nc = src->cols - 1 - c; break;
            default: nr = c; nc = src->rows - 1 - r; break;
            }
            nr += off_r; nc += off_c;
            if (nr < out->rows && nc < out->cols) memcpy(px_at(out, nr, nc), px_at(src, r, c), ps);
        }
    if (off_r != 0 || off_c != 0) {
        const uint32_t inner[4] = {off_c, off_r, off_c + rc, off_r + rr};
        const char zero[16] = {0};
        zo_set_border(out, inner, zero);
    }
}

int zo_rotate_into(const zo_image *src, const zo_image *out, float angle, float cos_a, float sin_a,
                   const zo_method *m, int border) {
    if (src->pixel != out->pixel) return 2;
    const int oc = orthogonal_case(angle);
    if (oc >= 0) { rotate_orthogonal(src, out, oc); return 0; }
    const float cx = (float)src->cols / 2.0f, cy = (float)src->rows / 2.0f; /* getCenter, image.zig:322-327 */
    const float offset_x = ((float)out->cols - (float)src->cols) / 2.0f;
    const float offset_y = ((float)out->rows - (float)src->rows) / 2.0f;
    const float rcx = cx + offset_x, rcy = cy + offset_y;
    const size_t ps = zo_pixel_size(out->pixel);
    for (size_t r = 0; r < out->rows; ++r) {
        const float y = (float)r;
        for (size_t c = 0; c < out->cols; ++c) {
            const float x = (float)c;
            const float dx = x - rcx, dy = y - rcy;
            const float rdx = cos_a * dx - sin_a * dy;
            const float rdy = sin_a * dx + cos_a * dy;
            const float sx = rdx + cx, sy = rdy + cy;
            char px[16] = {0};
            if (!zo_interpolate(src, sx, sy, m, border, px)) memset(px, 0, sizeof px);
            memcpy(px_at(out, r, c), px, ps);
        }
    }
    return 0;
}

/* Rectangle(f32).width / height (geometry/Rectangle.zig:76-93): 0 when degenerate */
static float rect_w(const float r[4]) { return r[0] >= r[2] ? 0 : r[2] - r[0]; }
static float rect_h(const float r[4]) { return r[1] >= r[3] ? 0 : r[3] - r[1]; }

/* copyRect (transforms.zig:465-518) */
static void copy_rect(const zo_image *src, int32_t rect_top, int32_t rect_left, const zo_image *out, int border) {
    const size_t ps = zo_pixel_size(src->pixel);
    const char zero[16] = {0};
    if (border == ZO_ZERO) {
        const int32_t r_min = rect_top > 0 ? rect_top : 0;
        const int32_t r_max = (int32_t)src->rows < rect_top + (int32_t)out->rows ? (int32_t)src->rows : rect_top + (int32_t)out->rows;
        const int32_t c_min = rect_left > 0 ? rect_left : 0;
        const int32_t c_max = (int32_t)src->cols < rect_left + (int32_t)out->cols ? (int32_t)src->cols : rect_left + (int32_t)out->cols;
        if (r_min < r_max && c_min < c_max) {
            const int covers_all = (uint32_t)(r_max - r_min) == out->rows && (uint32_t)(c_max - c_min) == out->cols;
            if (!covers_all) zo_fill(out, zero);
            const size_t len = (size_t)(c_max - c_min);
            for (int32_t r = r_min; r < r_max; ++r)
                memcpy(px_at(out, (size_t)(r - rect_top), (size_t)(c_min - rect_left)), px_at(src, (size_t)r, (size_t)c_min), len * ps);
        } else {
            zo_fill(out, zero);
        }
        return;
    }
    for (size_t r = 0; r < out->rows; ++r)
        for (size_t c = 0; c < out->cols; ++c) {
            const int64_t rr = zo_resolve_index((int64_t)r + rect_top, src->rows, border);
            const int64_t cc = rr < 0 ? -1 : zo_resolve_index((int64_t)c + rect_left, src->cols, border);
            if (rr >= 0 && cc >= 0) memcpy(px_at(out, r, c), px_at(src, (size_t)rr, (size_t)cc), ps);
            else memcpy(px_at(out, r, c), zero, ps);
        }
}

int zo_extract(const zo_image *src, const zo_image *out, const float rect[4], float angle, float cos_a, float sin_a,
               const zo_method *m, int border) {
    if (src->pixel != out->pixel) return 2;
    if (out->rows == 0 || out->cols == 0) return 0;
    const float frows = (float)out->rows, fcols = (float)out->cols;
    const float width = rect_w(rect), height = rect_h(rect);
    const float eps = 1e-6f;
    if (fabsf(angle) < eps && fabsf(width - fcols) < eps && fabsf(height - frows) < eps) {
        copy_rect(src, (int32_t)roundf(rect[1]), (int32_t)roundf(rect[0]), out, border);
        return 0;
    }
    const float cx = (rect[0] + rect[2]) * 0.5f, cy = (rect[1] + rect[3]) * 0.5f;
    const size_t ps = zo_pixel_size(out->pixel);
    for (size_t r = 0; r < out->rows; ++r) {
        const float ty = out->rows == 1 ? 0.5f : (float)r / (frows - 1);
        const float y_rect = rect[1] + ty * height;
        for (size_t c = 0; c < out->cols; ++c) {
            const float tx = out->cols == 1 ? 0.5f : (float)c / (fcols - 1);
            const float x_rect = rect[0] + tx * width;
            const float dx = x_rect - cx, dy = y_rect - cy;
            const float sx = cx + cos_a * dx - sin_a * dy;
            const float sy = cy + sin_a * dx + cos_a * dy;
            char px[16] = {0};
            if (!zo_interpolate(src, sx, sy, m, border, px)) memset(px, 0, sizeof px);
            memcpy(px_at(out, r, c), px, ps);
        }
    }
    return 0;
}

int zo_crop_dims(const float rect[4], uint32_t *rows, uint32_t *cols) {
    *rows = (uint32_t)roundf(rect_h(rect));
    *cols = (uint32_t)roundf(rect_w(rect));
    return 0;
}

int zo_crop(const zo_image *src, const zo_image *dst, const float rect[4]) {
    zo_method nearest = {ZO_NEAREST, 0, 0, 0};
    return zo_extract(src, dst, rect, 0, 1.0f, 0.0f, &nearest, ZO_ZERO);
}

/* Rgba(u8).blend(overlay, mode) — blendColors, src/blending.zig:27-157. mode: the Blending ordinal (none 0, normal 1,
 * multiply 2, screen 3, overlay 4, soft_light 5, hard_light 6, color_dodge 7, color_burn 8, darken 9, lighten 10,
 * difference 11, exclusion 12). f32 per channel in the reference's operation order (left-associative vector products). */
static float blend_channel(int mode, float b, float o) {
    switch (mode) {
    case 1: return o;
    case 2: return b * o;
    case 3: return 1.0f - (1.0f - b) * (1.0f - o);
    case 4: return b < 0.5f ? (2.0f * b) * o : 1.0f - (2.0f * (1.0f - b)) * (1.0f - o);
    case 5: return o <= 0.5f ? b - ((1.0f - 2.0f * o) * b) * (1.0f - b) : b + (2.0f * o - 1.0f) * (sqrtf(b) - b);
    case 6: return o < 0.5f ? (2.0f * o) * b : 1.0f - (2.0f * (1.0f - o)) * (1.0f - b);
    case 7: { const float r = b / (1.0f - o); return b == 0.0f ? 0.0f : (o >= 1.0f ? 1.0f : fminf(1.0f, r)); }
    case 8: { const float r = 1.0f - (1.0f - b) / o; return b >= 1.0f ? 1.0f : (o <= 0.0f ? 0.0f : fmaxf(0.0f, r)); }
    case 9: return fminf(b, o);
    case 10: return fmaxf(b, o);
    case 11: return fabsf(b - o);
    case 12: return (b + o) - (2.0f * b) * o;
    }
    return o;
}
static void blend_u8(uint8_t *base, const uint8_t *overlay, int mode) {
    if (mode == 0) { memcpy(base, overlay, 4); return; }
    if (overlay[3] == 0) return;
    if (base[3] == 0) { memcpy(base, overlay, 4); return; }
    if (mode == 1 && overlay[3] == 255) { memcpy(base, overlay, 4); return; }
    float b[4], o[4], out[4];
    for (int i = 0; i < 4; ++i) { b[i] = (float)base[i] / 255; o[i] = (float)overlay[i] / 255; }
    float blended[3];
    for (int i = 0; i < 3; ++i) blended[i] = blend_channel(mode, b[i], o[i]);
    if (overlay[3] == 255) {
        for (int i = 0; i < 3; ++i) out[i] = blended[i];
        out[3] = 1.0f;
    } else {
        const float result_a = o[3] + b[3] * (1.0f - o[3]);
        if (result_a <= 0) { memset(base, 0, 4); return; }
        const float base_weight = b[3] * (1.0f - o[3]);
        const float inv = 1.0f / result_a;
        for (int i = 0; i < 3; ++i) out[i] = (blended[i] * o[3] + b[i] * base_weight) * inv;
        out[3] = result_a;
    }
    for (int i = 0; i < 4; ++i) { /* Rgba(f32).as(u8): @round(255 * clamp(v, 0, 1)) */
        float v = out[i] < 0 ? 0 : (out[i] > 1 ? 1 : out[i]);
        base[i] = (uint8_t)roundf(255 * v);
    }
}
ZO_API void zo_blend_rgba_u8(uint8_t base[4], const uint8_t overlay[4], int mode) { blend_u8(base, overlay, mode); }

int zo_convert(const zo_image *src, int src_space, const zo_image *dst, int dst_space, const float *srgb_lut);
/* convertColor between image pixel types (color.zig:108-151): the colour space follows the channel count */
static void convert_px(int spix, const void *s, int dpix, void *d) {
    const zo_image si = {(void *)s, 1, 1, 1, spix}, di = {d, 1, 1, 1, dpix};
    const int sc = zo_channels(spix), dc = zo_channels(dpix);
    zo_convert(&si, sc == 1 ? ZO_CS_GRAY : (sc == 4 ? ZO_CS_RGBA : ZO_CS_RGB), &di, dc == 1 ? ZO_CS_GRAY : (dc == 4 ? ZO_CS_RGBA : ZO_CS_RGB), NULL);
}
/* assignPixel (image.zig:67-94): Rgba(u8) samples with a blend mode composite through Rgba(u8) (the destination is
 * converted to Rgba, blended, converted back); otherwise the sample is converted to the destination type and stored. */
static void assign_pixel(const zo_image *self, char *dest, int spix, const char *sample, int blend_mode) {
    if (spix == ZO_RGBA_U8 && blend_mode != 0) {
        if (self->pixel == ZO_RGBA_U8) { blend_u8((uint8_t *)dest, (const uint8_t *)sample, blend_mode); return; }
        uint8_t rgba[4];
        convert_px(self->pixel, dest, ZO_RGBA_U8, rgba);
        blend_u8(rgba, (const uint8_t *)sample, blend_mode);
        convert_px(ZO_RGBA_U8, rgba, self->pixel, dest);
        return;
    }
    if (spix == self->pixel) memcpy(dest, sample, zo_pixel_size(self->pixel));
    else convert_px(spix, sample, self->pixel, dest);
}

int zo_insert(const zo_image *self, const zo_image *source, const float rect[4], float angle, float cos_a, float sin_a,
              const zo_method *m, int blend_mode) {
    if (source->rows == 0 || source->cols == 0) return 0;
    const float frows = (float)source->rows, fcols = (float)source->cols;
    const float rect_width = rect_w(rect), rect_height = rect_h(rect);
    const float eps = 1e-6f;
    if (fabsf(angle) < eps && fabsf(rect_width - fcols) < eps && fabsf(rect_height - frows) < eps) {
        const int32_t dst_top = (int32_t)roundf(rect[1]), dst_left = (int32_t)roundf(rect[0]);
        for (size_t r = 0; r < source->rows; ++r) {
            const int64_t y = (int64_t)dst_top + (int64_t)r;
            for (size_t c = 0; c < source->cols; ++c) {
                const int64_t x = (int64_t)dst_left + (int64_t)c;
                if (y < 0 || x < 0 || y >= self->rows || x >= self->cols) continue;
                assign_pixel(self, px_at(self, (size_t)y, (size_t)x), source->pixel, px_at(source, r, c), blend_mode);
            }
        }
        return 0;
    }
    const float cx = (rect[0] + rect[2]) * 0.5f, cy = (rect[1] + rect[3]) * 0.5f;
    const float inv_width = 1.0f / rect_width, inv_height = 1.0f / rect_height;
    const float half_width = rect_width * 0.5f, half_height = rect_height * 0.5f;
    const float abs_cos = fabsf(cos_a), abs_sin = fabsf(sin_a);
    const float bound_hw = half_width * abs_cos + half_height * abs_sin;
    const float bound_hh = half_width * abs_sin + half_height * abs_cos;
    const uint32_t min_r = (cy - bound_hh < 0) ? 0 : (uint32_t)floorf(cy - bound_hh);
    uint32_t max_r = (uint32_t)ceilf(cy + bound_hh) + 1;
    if (max_r > self->rows) max_r = self->rows;
    const uint32_t min_c = (cx - bound_hw < 0) ? 0 : (uint32_t)floorf(cx - bound_hw);
    uint32_t max_c = (uint32_t)ceilf(cx + bound_hw) + 1;
    if (max_c > self->cols) max_c = self->cols;
    for (uint32_t r = min_r; r < max_r; ++r) {
        const float dy = (float)r - cy;
        for (uint32_t c = min_c; c < max_c; ++c) {
            const float dx = (float)c - cx;
            const float rect_x = cos_a * dx + sin_a * dy;
            const float rect_y = -sin_a * dx + cos_a * dy;
            if (fabsf(rect_x) > half_width || fabsf(rect_y) > half_height) continue;
            const float norm_x = (rect_x + half_width) * inv_width;
            const float norm_y = (rect_y + half_height) * inv_height;
            const float sx = source->cols == 1 ? 0 : norm_x * (fcols - 1);
            const float sy = source->rows == 1 ? 0 : norm_y * (frows - 1);
            char px[16];
            if (zo_interpolate(source, sx, sy, m, ZO_MIRROR, px)) assign_pixel(self, px_at(self, r, c), source->pixel, px, blend_mode);
        }
    }
    return 0;
}

/* ProjectiveTransform(f64).init with exactly four correspondences (geometry/transforms.zig:242-263):
 * 8x8 system a h = b solved in f64 (partial-pivot Gaussian elimination), h22 = 1, then .as(f32). */
int zo_homography_from_4pts(const double from_xy[8], const double to_xy[8], float m_out[9]) {
    double a[8][9];
    for (int i = 0; i < 4; ++i) {
        const double fx = from_xy[2 * i], fy = from_xy[2 * i + 1], tx = to_xy[2 * i], ty = to_xy[2 * i + 1];
        const double r0[9] = {fx, fy, 1, 0, 0, 0, -tx * fx, -tx * fy, tx};
        const double r1[9] = {0, 0, 0, fx, fy, 1, -ty * fx, -ty * fy, ty};
        memcpy(a[2 * i], r0, sizeof r0);
        memcpy(a[2 * i + 1], r1, sizeof r1);
    }
    for (int col = 0; col < 8; ++col) {
        int piv = col;
        for (int r = col + 1; r < 8; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (fabs(a[piv][col]) < 1e-300) return 2;
        if (piv != col) for (int k = 0; k < 9; ++k) { double t = a[col][k]; a[col][k] = a[piv][k]; a[piv][k] = t; }
        for (int r = col + 1; r < 8; ++r) {
            const double f = a[r][col] / a[col][col];
            for (int k = col; k < 9; ++k) a[r][k] -= f * a[col][k];
        }
    }
    double h[8];
    for (int r = 7; r >= 0; --r) {
        double s = a[r][8];
        for (int k = r + 1; k < 8; ++k) s -= a[r][k] * h[k];
        h[r] = s / a[r][r];
    }
    for (int i = 0; i < 8; ++i) m_out[i] = (float)h[i];
    m_out[8] = 1.0f;
    return 0;
}
