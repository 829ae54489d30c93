/*
 * oracle/integral.c — restatement of zignal's integral-image box blur. TEST INFRASTRUCTURE ONLY (zo.h).
 *   src/image/integral.zig:41-78     plane (f32 row cumulative sums, then column accumulation)
 *   src/image/integral.zig:86-91     sum (sat[r2,c2] - sat[r2,c1-1] - sat[r1-1,c2] + sat[r1-1,c1-1], in that order)
 *   src/image/integral.zig:194-269   boxBlurPlane (window and area clipped at the borders)
 *   src/image.zig:635-648            Image.boxBlur (radius 0 copies; SAT built before any output is written)
 */
#include "zo.h"
#include <stdlib.h>
#include <string.h>

int zo_copy(const zo_image *src, const zo_image *dst);

int zo_integral_plane_f32(const float *src, size_t src_stride, float *sat, uint32_t rows, uint32_t cols) {
    for (size_t r = 0; r < rows; ++r) {
        float tmp = 0;
        for (size_t c = 0; c < cols; ++c) {
            tmp += src[r * src_stride + c];
            sat[r * cols + c] = tmp;
        }
    }
    for (size_t r = 1; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) sat[r * cols + c] += sat[(r - 1) * cols + c];
    return 0;
}

static float sat_sum(const float *sat, size_t stride, size_t r1, size_t c1, size_t r2, size_t c2) {
    return sat[r2 * stride + c2] - (c1 > 0 ? sat[r2 * stride + (c1 - 1)] : 0) - (r1 > 0 ? sat[(r1 - 1) * stride + c2] : 0) +
           ((r1 > 0 && c1 > 0) ? sat[(r1 - 1) * stride + (c1 - 1)] : 0);
}

int zo_box_blur(const zo_image *src, const zo_image *dst, uint32_t radius) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel) return 2;
    if (radius == 0) return zo_copy(src, dst);
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    const int nch = zo_channels(src->pixel), isf = zo_is_float(src->pixel);
    float *plane = (float *)malloc(n * sizeof(float));
    float **sats = (float **)malloc((size_t)nch * sizeof(float *));
    if (!plane || !sats) return 3;
    /* every channel's SAT is complete before the first output pixel is written (src may alias dst) */
    for (int ch = 0; ch < nch; ++ch) {
        sats[ch] = (float *)malloc(n * sizeof(float));
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const size_t i = (r * src->stride + c) * (size_t)nch + (size_t)ch;
                plane[r * cols + c] = isf ? ((const float *)src->data)[i] : (float)((const uint8_t *)src->data)[i];
            }
        zo_integral_plane_f32(plane, cols, sats[ch], (uint32_t)rows, (uint32_t)cols);
    }
    for (int ch = 0; ch < nch; ++ch) {
        for (size_t r = 0; r < rows; ++r) {
            const size_t r1 = r > radius ? r - radius : 0, r2 = r + radius < rows - 1 ? r + radius : rows - 1;
            for (size_t c = 0; c < cols; ++c) {
                const size_t c1 = c > radius ? c - radius : 0, c2 = c + radius < cols - 1 ? c + radius : cols - 1;
                const float area = (float)((r2 - r1 + 1) * (c2 - c1 + 1));
                const float val = sat_sum(sats[ch], cols, r1, c1, r2, c2) / area;
                const size_t i = (r * dst->stride + c) * (size_t)nch + (size_t)ch;
                if (isf) ((float *)dst->data)[i] = val;
                else ((uint8_t *)dst->data)[i] = zo_clamp_u8_f32(val);
            }
        }
        free(sats[ch]);
    }
    free(sats);
    free(plane);
    return 0;
}

/* Image(T).sharpen (image.zig:785-801 -> Integral.sharpen, integral.zig:273-426): 2 * original - blurred, same windows and
 * integral planes as boxBlur; integer fields @round / clamp / @trunc (== meta.clamp), float fields as is. */
int zo_sharpen(const zo_image *src, const zo_image *dst, uint32_t radius) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel) return 2;
    if (radius == 0) return zo_copy(src, dst);
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    const int nch = zo_channels(src->pixel), isf = zo_is_float(src->pixel);
    float *plane = (float *)malloc(n * sizeof(float));
    float **sats = (float **)malloc((size_t)nch * sizeof(float *));
    for (int ch = 0; ch < nch; ++ch) {
        sats[ch] = (float *)malloc(n * sizeof(float));
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const size_t i = (r * src->stride + c) * (size_t)nch + (size_t)ch;
                plane[r * cols + c] = isf ? ((const float *)src->data)[i] : (float)((const uint8_t *)src->data)[i];
            }
        zo_integral_plane_f32(plane, cols, sats[ch], (uint32_t)rows, (uint32_t)cols);
    }
    for (int ch = 0; ch < nch; ++ch) {
        for (size_t r = 0; r < rows; ++r) {
            const size_t r1 = r > radius ? r - radius : 0, r2 = r + radius < rows - 1 ? r + radius : rows - 1;
            for (size_t c = 0; c < cols; ++c) {
                const size_t c1 = c > radius ? c - radius : 0, c2 = c + radius < cols - 1 ? c + radius : cols - 1;
                const float area = (float)((r2 - r1 + 1) * (c2 - c1 + 1));
                const float blurred = sat_sum(sats[ch], cols, r1, c1, r2, c2) / area;
                const size_t si = (r * src->stride + c) * (size_t)nch + (size_t)ch, di = (r * dst->stride + c) * (size_t)nch + (size_t)ch;
                const float original = isf ? ((const float *)src->data)[si] : (float)((const uint8_t *)src->data)[si];
                const float sharpened = 2 * original - blurred;
                if (isf) ((float *)dst->data)[di] = sharpened;
                else ((uint8_t *)dst->data)[di] = zo_clamp_u8_f32(sharpened);
            }
        }
        free(sats[ch]);
    }
    free(sats);
    free(plane);
    return 0;
}

/* Image(T).integral: planes[ch] (rows x cols f32, packed), integral.zig:95-140 */
int zo_integral(const zo_image *src, float *planes) {
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    const int nch = zo_channels(src->pixel), isf = zo_is_float(src->pixel);
    float *plane = (float *)malloc(n * sizeof(float));
    for (int ch = 0; ch < nch; ++ch) {
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const size_t i = (r * src->stride + c) * (size_t)nch + (size_t)ch;
                plane[r * cols + c] = isf ? ((const float *)src->data)[i] : (float)((const uint8_t *)src->data)[i];
            }
        zo_integral_plane_f32(plane, cols, planes + (size_t)ch * n, (uint32_t)rows, (uint32_t)cols);
    }
    free(plane);
    return 0;
}

/* Image(T).invert (image.zig:494-513), in place; Image(f32) has none (returns 5) */
int zo_invert(const zo_image *img) {
    if (img->pixel == ZO_F32) return 5;
    const int nch = zo_channels(img->pixel), isf = zo_is_float(img->pixel), n = nch == 4 ? 3 : nch;
    for (size_t r = 0; r < img->rows; ++r)
        for (size_t c = 0; c < img->cols; ++c)
            for (int ch = 0; ch < n; ++ch) {
                const size_t i = (r * img->stride + c) * (size_t)nch + (size_t)ch;
                if (isf) ((float *)img->data)[i] = 1.0f - ((float *)img->data)[i];
                else ((uint8_t *)img->data)[i] = (uint8_t)(255 - ((uint8_t *)img->data)[i]);
            }
    return 0;
}
