/*
 * oracle/conv.c — restatement of zignal's border handling and convolution filters.
 * TEST INFRASTRUCTURE ONLY (see zo.h). Follows, loop for loop:
 *   src/image/border.zig:46-63            resolveIndex
 *   src/image/convolution.zig:18-22       divClampU8
 *   src/image/convolution.zig:303-309     scaleKernelToInt
 *   src/image/convolution.zig:313-438     convolveSeparable (type switch, split / merge, uniform shortcut)
 *   src/image/convolution.zig:441-655     convolveSeparablePlane, getPixel
 *   src/image/convolution.zig:76-301      convolve (2-D)
 *   src/image/channel_ops.zig:56-136      splitChannelsWithUniform / mergeChannels
 *   src/image.zig:954-994                 gaussianBlur
 * including the reference's memory behaviour (AoS->SoA split, full-image temp plane, 16-column
 * tiled vertical pass, SoA->AoS merge) so that its timing is a fair "port" CPU baseline.
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- border.zig:46-63 ---------------------------------------------------------------- */
int64_t zo_resolve_index(int64_t idx, int64_t length, int border) {
    if (idx >= 0 && idx < length) return idx;
    switch (border) {
    case ZO_ZERO: return -1;
    case ZO_REPLICATE:
        if (length == 0) return -1;
        return idx < 0 ? 0 : (idx > length - 1 ? length - 1 : idx);
    case ZO_MIRROR: {
        if (length <= 0) return -1;
        if (length == 1) return 0;
        int64_t period = 2 * (length - 1);
        int64_t m = idx % period;
        if (m < 0) m += period; /* @mod is floored */
        return m >= length ? period - m : m;
    }
    case ZO_WRAP: {
        if (length == 0) return -1;
        int64_t m = idx % length;
        if (m < 0) m += length;
        return m;
    }
    }
    return -1;
}

/* meta.zig:110-135 — float -> u8: trunc(clamp(round(f64 v), 0, 255)), round half away from zero */
uint8_t zo_clamp_u8_f32(float v) {
    double r = round((double)v);
    /* std.math.clamp = @max(lo, @min(v, hi)); @min/@max return the non-NaN operand, as fmin/fmax */
    return (uint8_t)trunc(fmax(0.0, fmin(r, 255.0)));
}

/* convolution.zig:18-22 */
static inline uint8_t div_clamp_u8(int64_t scale, int64_t accum) {
    int64_t half = scale / 2;
    int64_t rounded = (accum + (accum >= 0 ? half : -half)) / scale; /* C '/' == @divTrunc */
    return zo_clamp_u8_i64(rounded);
}
static inline int32_t clamp_i32(int64_t v) {
    return (int32_t)(v < INT32_MIN ? INT32_MIN : (v > INT32_MAX ? INT32_MAX : v));
}

/* convolution.zig:303-309: result[i] = @round(k * scale) -> i32 */
static void scale_kernel_to_int(const float *k, uint32_t n, int32_t scale, int32_t *out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = (int32_t)roundf(k[i] * (float)scale);
}

/* ---- plane images -------------------------------------------------------------------- */
typedef struct { uint8_t *data; size_t stride; uint32_t rows, cols; } plane_u8;
typedef struct { int32_t *data; size_t stride; uint32_t rows, cols; } plane_i32;
typedef struct { float *data; size_t stride; uint32_t rows, cols; } plane_f32;

/* getPixel (convolution.zig:650-655) */
static inline int32_t get_u8(plane_u8 img, int64_t r, int64_t c, int border) {
    int64_t rr = zo_resolve_index(r, img.rows, border);
    if (rr < 0) return 0;
    int64_t cc = zo_resolve_index(c, img.cols, border);
    if (cc < 0) return 0;
    return img.data[(size_t)rr * img.stride + (size_t)cc];
}
static inline int32_t get_i32(plane_i32 img, int64_t r, int64_t c, int border) {
    int64_t rr = zo_resolve_index(r, img.rows, border);
    if (rr < 0) return 0;
    int64_t cc = zo_resolve_index(c, img.cols, border);
    if (cc < 0) return 0;
    return img.data[(size_t)rr * img.stride + (size_t)cc];
}
static inline float get_f32(plane_f32 img, int64_t r, int64_t c, int border) {
    int64_t rr = zo_resolve_index(r, img.rows, border);
    if (rr < 0) return 0;
    int64_t cc = zo_resolve_index(c, img.cols, border);
    if (cc < 0) return 0;
    return img.data[(size_t)rr * img.stride + (size_t)cc];
}

#define TILE_W 16 /* @max(vec_len, 16) (convolution.zig:579); results do not depend on it */
#define VEC_W 8   /* std.simd.suggestVectorLength(f32 / i32) on an AVX2 host; results do not depend on it */

/* convolveSeparablePlane(u8, i32, ...) — convolution.zig:441-647 */
static void sep_plane_u8(plane_u8 src, plane_u8 dst, plane_i32 tmp, const int32_t *kx, uint32_t nkx,
                         const int32_t *ky, uint32_t nky, int border) {
    const size_t half_x = nkx / 2, half_y = nky / 2;
    const size_t rows = src.rows, cols = src.cols;
    /* horizontal pass */
    for (size_t r = 0; r < rows; ++r) {
        const size_t row_off = r * src.stride, tmp_off = r * tmp.stride;
        size_t c = 0;
        const size_t left_end = half_x < cols ? half_x : cols;
        for (; c < left_end; ++c) {
            int64_t acc = 0;
            for (size_t i = 0; i < nkx; ++i)
                acc += (int64_t)get_u8(src, (int64_t)r, (int64_t)c + (int64_t)i - (int64_t)half_x, border) * (int64_t)kx[i];
            tmp.data[tmp_off + c] = clamp_i32(acc);
        }
        if (cols > 2 * half_x) {
            const size_t interior_end = cols - half_x;
            for (; c < interior_end; ++c) {
                int64_t acc = 0;
                const size_t c0 = c - half_x;
                for (size_t i = 0; i < nkx; ++i) {
                    if (kx[i] == 0) continue; /* isNegligible */
                    acc += (int64_t)src.data[row_off + c0 + i] * (int64_t)kx[i];
                }
                tmp.data[tmp_off + c] = clamp_i32(acc);
            }
        }
        for (; c < cols; ++c) {
            int64_t acc = 0;
            for (size_t i = 0; i < nkx; ++i)
                acc += (int64_t)get_u8(src, (int64_t)r, (int64_t)c + (int64_t)i - (int64_t)half_x, border) * (int64_t)kx[i];
            tmp.data[tmp_off + c] = clamp_i32(acc);
        }
    }
    /* vertical pass, tiled by columns */
    if (rows > 2 * half_y) {
        const size_t safe_end = rows - half_y;
        for (size_t tile_c = 0; tile_c < cols; tile_c += TILE_W) {
            const size_t tile_end = tile_c + TILE_W < cols ? tile_c + TILE_W : cols;
            for (size_t cv = tile_c; cv < tile_end; cv += VEC_W) { /* vec_len-wide column strips, all rows each */
                const size_t cv_end = cv + VEC_W < tile_end ? cv + VEC_W : tile_end;
                for (size_t r = half_y; r < safe_end; ++r) {
                    const size_t r0 = r - half_y;
                    for (size_t c = cv; c < cv_end; ++c) {
                        int64_t acc = 0;
                        for (size_t i = 0; i < nky; ++i) {
                            if (ky[i] == 0) continue;
                            acc += (int64_t)tmp.data[(r0 + i) * tmp.stride + c] * (int64_t)ky[i];
                        }
                        dst.data[r * dst.stride + c] = div_clamp_u8(65536, acc);
                    }
                }
            }
        }
    }
    /* top / bottom border rows */
    const size_t top_end = half_y < rows ? half_y : rows;
    size_t bottom_start = rows;
    if (rows > half_y) bottom_start = top_end > rows - half_y ? top_end : rows - half_y;
    const size_t ranges[2][2] = {{0, top_end}, {bottom_start, rows}};
    for (int k = 0; k < 2; ++k)
        for (size_t r = ranges[k][0]; r < ranges[k][1]; ++r)
            for (size_t c = 0; c < cols; ++c) {
                int64_t acc = 0;
                for (size_t i = 0; i < nky; ++i)
                    acc += (int64_t)get_i32(tmp, (int64_t)r + (int64_t)i - (int64_t)half_y, (int64_t)c, border) * (int64_t)ky[i];
                dst.data[r * dst.stride + c] = div_clamp_u8(65536, acc);
            }
}

/* convolveSeparablePlane(f32, f32, ...) — same structure, f32 mul then add (no FMA) */
static void sep_plane_f32(plane_f32 src, plane_f32 dst, plane_f32 tmp, const float *kx, uint32_t nkx,
                          const float *ky, uint32_t nky, int border) {
    const size_t half_x = nkx / 2, half_y = nky / 2;
    const size_t rows = src.rows, cols = src.cols;
    for (size_t r = 0; r < rows; ++r) {
        const size_t row_off = r * src.stride, tmp_off = r * tmp.stride;
        size_t c = 0;
        const size_t left_end = half_x < cols ? half_x : cols;
        for (; c < left_end; ++c) {
            float acc = 0;
            for (size_t i = 0; i < nkx; ++i)
                acc += get_f32(src, (int64_t)r, (int64_t)c + (int64_t)i - (int64_t)half_x, border) * kx[i];
            tmp.data[tmp_off + c] = acc;
        }
        if (cols > 2 * half_x) {
            const size_t interior_end = cols - half_x;
            for (; c < interior_end; ++c) {
                float acc = 0;
                const size_t c0 = c - half_x;
                for (size_t i = 0; i < nkx; ++i) {
                    if (fabsf(kx[i]) < 1e-10f) continue;
                    acc += src.data[row_off + c0 + i] * kx[i];
                }
                tmp.data[tmp_off + c] = acc;
            }
        }
        for (; c < cols; ++c) {
            float acc = 0;
            for (size_t i = 0; i < nkx; ++i)
                acc += get_f32(src, (int64_t)r, (int64_t)c + (int64_t)i - (int64_t)half_x, border) * kx[i];
            tmp.data[tmp_off + c] = acc;
        }
    }
    if (rows > 2 * half_y) {
        const size_t safe_end = rows - half_y;
        for (size_t tile_c = 0; tile_c < cols; tile_c += TILE_W) {
            const size_t tile_end = tile_c + TILE_W < cols ? tile_c + TILE_W : cols;
            for (size_t cv = tile_c; cv < tile_end; cv += VEC_W) {
                const size_t cv_end = cv + VEC_W < tile_end ? cv + VEC_W : tile_end;
                for (size_t r = half_y; r < safe_end; ++r) {
                    const size_t r0 = r - half_y;
                    for (size_t c = cv; c < cv_end; ++c) {
                        float acc = 0;
                        for (size_t i = 0; i < nky; ++i) {
                            if (fabsf(ky[i]) < 1e-10f) continue;
                            acc += tmp.data[(r0 + i) * tmp.stride + c] * ky[i];
                        }
                        dst.data[r * dst.stride + c] = acc;
                    }
                }
            }
        }
    }
    const size_t top_end = half_y < rows ? half_y : rows;
    size_t bottom_start = rows;
    if (rows > half_y) bottom_start = top_end > rows - half_y ? top_end : rows - half_y;
    const size_t ranges[2][2] = {{0, top_end}, {bottom_start, rows}};
    for (int k = 0; k < 2; ++k)
        for (size_t r = ranges[k][0]; r < ranges[k][1]; ++r)
            for (size_t c = 0; c < cols; ++c) {
                float acc = 0;
                for (size_t i = 0; i < nky; ++i)
                    acc += get_f32(tmp, (int64_t)r + (int64_t)i - (int64_t)half_y, (int64_t)c, border) * ky[i];
                dst.data[r * dst.stride + c] = acc;
            }
}

/* channel_ops.zig:56-112: AoS -> SoA with per-channel uniformity flag */
static int split_u8(const zo_image *img, int nch, uint8_t **planes, int *uniform, uint8_t *uniform_val) {
    const size_t n = (size_t)img->rows * img->cols;
    for (int i = 0; i < nch; ++i) {
        planes[i] = (uint8_t *)malloc(n ? n : 1);
        if (!planes[i]) { for (int j = 0; j < i; ++j) free(planes[j]); return -1; }
        uniform[i] = 1;
    }
    int have = 0;
    size_t idx = 0;
    const uint8_t *base = (const uint8_t *)img->data;
    for (size_t r = 0; r < img->rows; ++r)
        for (size_t c = 0; c < img->cols; ++c) {
            const uint8_t *px = base + (r * img->stride + c) * (size_t)nch;
            for (int i = 0; i < nch; ++i) {
                uint8_t v = px[i];
                planes[i][idx] = v;
                if (!have) uniform_val[i] = v;
                else if (uniform[i] && v != uniform_val[i]) uniform[i] = 0;
            }
            have = 1;
            ++idx;
        }
    if (!have) for (int i = 0; i < nch; ++i) uniform[i] = 0;
    return 0;
}
/* channel_ops.zig:122-136 */
static void merge_u8(const zo_image *out, int nch, uint8_t *const *planes) {
    size_t idx = 0;
    uint8_t *base = (uint8_t *)out->data;
    for (size_t r = 0; r < out->rows; ++r)
        for (size_t c = 0; c < out->cols; ++c) {
            uint8_t *px = base + (r * out->stride + c) * (size_t)nch;
            for (int i = 0; i < nch; ++i) px[i] = planes[i][idx];
            ++idx;
        }
}

static int preserves_uniform(int border) { return border != ZO_ZERO; }

/* f32 struct extension (NOT in the reference, which rejects Rgba(f32) at comptime,
 * convolution.zig:431-435): defined as the F32 plane path applied to each channel. */
static int sep_f32_struct(const zo_image *src, const zo_image *dst, const float *kx, uint32_t nkx,
                          const float *ky, uint32_t nky, int border) {
    const int nch = zo_channels(src->pixel);
    const size_t n = (size_t)src->rows * src->cols;
    float *sp = (float *)malloc((n ? n : 1) * sizeof(float));
    float *dp = (float *)malloc((n ? n : 1) * sizeof(float));
    float *tp = (float *)malloc((n ? n : 1) * sizeof(float));
    if (!sp || !dp || !tp) { free(sp); free(dp); free(tp); return 3; }
    for (int ch = 0; ch < nch; ++ch) {
        for (size_t r = 0; r < src->rows; ++r)
            for (size_t c = 0; c < src->cols; ++c)
                sp[r * src->cols + c] = ((const float *)src->data)[(r * src->stride + c) * nch + ch];
        plane_f32 s = {sp, src->cols, src->rows, src->cols}, d = {dp, src->cols, src->rows, src->cols},
                  t = {tp, src->cols, src->rows, src->cols};
        sep_plane_f32(s, d, t, kx, nkx, ky, nky, border);
        for (size_t r = 0; r < src->rows; ++r)
            for (size_t c = 0; c < src->cols; ++c)
                ((float *)dst->data)[(r * dst->stride + c) * nch + ch] = dp[r * src->cols + c];
    }
    free(sp); free(dp); free(tp);
    return 0;
}

int zo_conv_separable(const zo_image *src, const zo_image *dst, const float *kx, uint32_t nkx,
                      const float *ky, uint32_t nky, int border) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel || nkx == 0 || nky == 0) return 2;
    const size_t n = (size_t)src->rows * src->cols;
    switch (src->pixel) {
    case ZO_U8: {
        int32_t *tmp = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
        int32_t *kxi = (int32_t *)malloc(nkx * sizeof(int32_t)), *kyi = (int32_t *)malloc(nky * sizeof(int32_t));
        if (!tmp || !kxi || !kyi) { free(tmp); free(kxi); free(kyi); return 3; }
        scale_kernel_to_int(kx, nkx, 256, kxi);
        scale_kernel_to_int(ky, nky, 256, kyi);
        plane_u8 s = {(uint8_t *)src->data, src->stride, src->rows, src->cols};
        plane_u8 d = {(uint8_t *)dst->data, dst->stride, dst->rows, dst->cols};
        plane_i32 t = {tmp, src->cols, src->rows, src->cols};
        sep_plane_u8(s, d, t, kxi, nkx, kyi, nky, border);
        free(tmp); free(kxi); free(kyi);
        return 0;
    }
    case ZO_F32: {
        float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
        if (!tmp) return 3;
        plane_f32 s = {(float *)src->data, src->stride, src->rows, src->cols};
        plane_f32 d = {(float *)dst->data, dst->stride, dst->rows, dst->cols};
        plane_f32 t = {tmp, src->cols, src->rows, src->cols};
        sep_plane_f32(s, d, t, kx, nkx, ky, nky, border);
        free(tmp);
        return 0;
    }
    case ZO_RGB_U8:
    case ZO_RGBA_U8: {
        const int nch = zo_channels(src->pixel);
        int32_t *kxi = (int32_t *)malloc(nkx * sizeof(int32_t)), *kyi = (int32_t *)malloc(nky * sizeof(int32_t));
        if (!kxi || !kyi) { free(kxi); free(kyi); return 3; }
        scale_kernel_to_int(kx, nkx, 256, kxi);
        scale_kernel_to_int(ky, nky, 256, kyi);
        int64_t kx_sum = 0, ky_sum = 0;
        for (uint32_t i = 0; i < nkx; ++i) kx_sum += kxi[i];
        for (uint32_t i = 0; i < nky; ++i) ky_sum += kyi[i];
        const int64_t kernel_sum = kx_sum * ky_sum, scale_sq = 65536;
        uint8_t *ch[4] = {0}, *och[4] = {0};
        int uniform[4];
        uint8_t uval[4];
        if (split_u8(src, nch, ch, uniform, uval)) { free(kxi); free(kyi); return 3; }
        enum { NORMALIZED, SCALED, NON_UNIFORM } strat[4];
        for (int i = 0; i < nch; ++i) {
            if (uniform[i] && preserves_uniform(border)) strat[i] = kernel_sum == scale_sq ? NORMALIZED : SCALED;
            else strat[i] = NON_UNIFORM;
        }
        int32_t *tmp = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
        for (int i = 0; i < nch; ++i) {
            if (strat[i] == NORMALIZED) continue;
            och[i] = (uint8_t *)malloc(n ? n : 1);
            if (strat[i] == SCALED) memset(och[i], div_clamp_u8(65536, (int64_t)uval[i] * kernel_sum), n);
        }
        for (int i = 0; i < nch; ++i) {
            if (strat[i] != NON_UNIFORM) continue;
            plane_u8 s = {ch[i], src->cols, src->rows, src->cols}, d = {och[i], src->cols, src->rows, src->cols};
            plane_i32 t = {tmp, src->cols, src->rows, src->cols};
            sep_plane_u8(s, d, t, kxi, nkx, kyi, nky, border);
        }
        uint8_t *fin[4];
        for (int i = 0; i < nch; ++i) fin[i] = strat[i] == NORMALIZED ? ch[i] : och[i];
        merge_u8(dst, nch, fin);
        for (int i = 0; i < nch; ++i) { free(ch[i]); free(och[i]); }
        free(tmp); free(kxi); free(kyi);
        return 0;
    }
    case ZO_RGB_F32:
    case ZO_RGBA_F32:
        return sep_f32_struct(src, dst, kx, nkx, ky, nky, border);
    }
    return 5;
}

/* image.zig:973-990 */
int zo_gaussian_kernel(float sigma, float *taps, uint32_t capacity) {
    if (!(sigma > 0)) return -2;
    const uint32_t radius = (uint32_t)ceilf(3.0f * sigma);
    const uint32_t size = 2 * radius + 1;
    if (!taps) return (int)size;
    if (capacity < size) return -2;
    float sum = 0;
    for (uint32_t i = 0; i < size; ++i) {
        const float x = (float)i - (float)radius;
        taps[i] = zo_expf(-(x * x) / (2.0f * sigma * sigma));
        sum += taps[i];
    }
    for (uint32_t i = 0; i < size; ++i) taps[i] /= sum;
    return (int)size;
}

int zo_copy(const zo_image *src, const zo_image *dst);

/* image.zig:954-994 */
int zo_gaussian_blur(const zo_image *src, const zo_image *dst, float sigma) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (sigma == 0) return zo_copy(src, dst);
    if (sigma < 0 || sigma != sigma) return 2;
    int n = zo_gaussian_kernel(sigma, NULL, 0);
    float *k = (float *)malloc((size_t)n * sizeof(float));
    if (!k) return 3;
    zo_gaussian_kernel(sigma, k, (uint32_t)n);
    int rc = zo_conv_separable(src, dst, k, (uint32_t)n, k, (uint32_t)n, ZO_MIRROR);
    free(k);
    return rc;
}

/* ---- 2-D convolution (convolution.zig:76-195) ---------------------------------------- */
static void conv2d_plane_u8(plane_u8 src, plane_u8 dst, const int32_t *k, uint32_t kh, uint32_t kw, int border) {
    const size_t half_h = kh / 2, half_w = kw / 2;
    for (size_t r = 0; r < src.rows; ++r) {
        const int row_in_band = r >= half_h && r + half_h < src.rows;
        for (size_t c = 0; c < src.cols; ++c) {
            int64_t acc = 0;
            if (row_in_band && c >= half_w && c + half_w < src.cols) {
                for (size_t ky = 0; ky < kh; ++ky)
                    for (size_t kx = 0; kx < kw; ++kx)
                        acc += (int64_t)src.data[(r + ky - half_h) * src.stride + (c + kx - half_w)] * (int64_t)k[ky * kw + kx];
            } else {
                for (size_t ky = 0; ky < kh; ++ky)
                    for (size_t kx = 0; kx < kw; ++kx)
                        acc += (int64_t)get_u8(src, (int64_t)r + (int64_t)ky - (int64_t)half_h,
                                               (int64_t)c + (int64_t)kx - (int64_t)half_w, border) * (int64_t)k[ky * kw + kx];
            }
            dst.data[r * dst.stride + c] = div_clamp_u8(256, acc);
        }
    }
}
static void conv2d_plane_f32(plane_f32 src, plane_f32 dst, const float *k, uint32_t kh, uint32_t kw, int border) {
    const size_t half_h = kh / 2, half_w = kw / 2;
    for (size_t r = 0; r < src.rows; ++r) {
        const int row_in_band = r >= half_h && r + half_h < src.rows;
        for (size_t c = 0; c < src.cols; ++c) {
            float acc = 0;
            if (row_in_band && c >= half_w && c + half_w < src.cols) {
                for (size_t ky = 0; ky < kh; ++ky)
                    for (size_t kx = 0; kx < kw; ++kx)
                        acc += src.data[(r + ky - half_h) * src.stride + (c + kx - half_w)] * k[ky * kw + kx];
            } else {
                for (size_t ky = 0; ky < kh; ++ky)
                    for (size_t kx = 0; kx < kw; ++kx)
                        acc += get_f32(src, (int64_t)r + (int64_t)ky - (int64_t)half_h,
                                       (int64_t)c + (int64_t)kx - (int64_t)half_w, border) * k[ky * kw + kx];
            }
            dst.data[r * dst.stride + c] = acc;
        }
    }
}

int zo_convolve(const zo_image *src, const zo_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel || kh == 0 || kw == 0) return 2;
    const size_t n = (size_t)src->rows * src->cols, ks = (size_t)kh * kw;
    switch (src->pixel) {
    case ZO_U8: {
        int32_t *ki = (int32_t *)malloc(ks * sizeof(int32_t));
        scale_kernel_to_int(kernel, (uint32_t)ks, 256, ki); /* flatten: @round(val * 256) */
        plane_u8 s = {(uint8_t *)src->data, src->stride, src->rows, src->cols};
        plane_u8 d = {(uint8_t *)dst->data, dst->stride, dst->rows, dst->cols};
        conv2d_plane_u8(s, d, ki, kh, kw, border);
        free(ki);
        return 0;
    }
    case ZO_F32: {
        plane_f32 s = {(float *)src->data, src->stride, src->rows, src->cols};
        plane_f32 d = {(float *)dst->data, dst->stride, dst->rows, dst->cols};
        conv2d_plane_f32(s, d, kernel, kh, kw, border);
        return 0;
    }
    case ZO_RGB_U8:
    case ZO_RGBA_U8: {
        const int nch = zo_channels(src->pixel);
        int32_t *ki = (int32_t *)malloc(ks * sizeof(int32_t));
        scale_kernel_to_int(kernel, (uint32_t)ks, 256, ki);
        int32_t kernel_sum = 0;
        for (size_t i = 0; i < ks; ++i) kernel_sum += ki[i];
        uint8_t *ch[4] = {0}, *och[4] = {0};
        int uniform[4];
        uint8_t uval[4];
        if (split_u8(src, nch, ch, uniform, uval)) { free(ki); return 3; }
        enum { NORMALIZED, SCALED, NON_UNIFORM } strat[4];
        for (int i = 0; i < nch; ++i) {
            if (uniform[i] && preserves_uniform(border)) strat[i] = kernel_sum == 256 ? NORMALIZED : SCALED;
            else strat[i] = NON_UNIFORM;
        }
        for (int i = 0; i < nch; ++i) {
            if (strat[i] == NORMALIZED) continue;
            och[i] = (uint8_t *)malloc(n ? n : 1);
            if (strat[i] == SCALED) memset(och[i], div_clamp_u8(256, (int64_t)uval[i] * (int64_t)kernel_sum), n);
            else {
                plane_u8 s = {ch[i], src->cols, src->rows, src->cols}, d = {och[i], src->cols, src->rows, src->cols};
                conv2d_plane_u8(s, d, ki, kh, kw, border);
            }
        }
        uint8_t *fin[4];
        for (int i = 0; i < nch; ++i) fin[i] = strat[i] == NORMALIZED ? ch[i] : och[i];
        merge_u8(dst, nch, fin);
        for (int i = 0; i < nch; ++i) { free(ch[i]); free(och[i]); }
        free(ki);
        return 0;
    }
    case ZO_RGB_F32:
    case ZO_RGBA_F32: { /* extension, per-channel F32 plane path (see sep_f32_struct) */
        const int nch = zo_channels(src->pixel);
        float *sp = (float *)malloc((n ? n : 1) * sizeof(float)), *dp = (float *)malloc((n ? n : 1) * sizeof(float));
        for (int c4 = 0; c4 < nch; ++c4) {
            for (size_t r = 0; r < src->rows; ++r)
                for (size_t c = 0; c < src->cols; ++c)
                    sp[r * src->cols + c] = ((const float *)src->data)[(r * src->stride + c) * nch + c4];
            plane_f32 s = {sp, src->cols, src->rows, src->cols}, d = {dp, src->cols, src->rows, src->cols};
            conv2d_plane_f32(s, d, kernel, kh, kw, border);
            for (size_t r = 0; r < src->rows; ++r)
                for (size_t c = 0; c < src->cols; ++c)
                    ((float *)dst->data)[(r * dst->stride + c) * nch + c4] = dp[r * src->cols + c];
        }
        free(sp); free(dp);
        return 0;
    }
    }
    return 5;
}
