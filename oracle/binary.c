/*
 * oracle/binary.c — restatement of src/image/binary.zig (Image(u8) only). TEST INFRASTRUCTURE ONLY (zo.h).
 *   :38-84    thresholdOtsu: 256-bin histogram, between-class variance in f64, out = src > t ? 255 : 0, returns t
 *   :86-118   thresholdAdaptiveMean: integral image, window mean (clipped area), out = src > mean - c ? 255 : 0
 *   :121-281  dilate / erode / open / close with a structuring element (non-zero = on, centre anchor, not flipped);
 *             dilation ignores out-of-image samples, erosion treats them as background
 */
#include "zo.h"
#include <stdlib.h>
#include <string.h>

int zo_integral_plane_f32(const float *src, size_t src_stride, float *sat, uint32_t rows, uint32_t cols);

ZO_API int zo_threshold_otsu(const zo_image *src, const zo_image *dst, uint8_t *threshold_out) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != ZO_U8 || dst->pixel != ZO_U8) return 2;
    *threshold_out = 0;
    if (src->rows == 0 || src->cols == 0) return 0;
    size_t hist[256] = {0};
    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c) hist[((const uint8_t *)src->data)[r * src->stride + c]] += 1;
    const double total_pixels = (double)((size_t)src->rows * src->cols);
    double sum_total = 0;
    for (int i = 0; i < 256; ++i) sum_total += (double)hist[i] * (double)i;
    double sum_background = 0, weight_background = 0, max_variance = -1;
    uint8_t threshold = 0;
    for (int i = 0; i < 256; ++i) {
        const double count_f = (double)hist[i];
        weight_background += count_f;
        if (weight_background == 0) continue;
        const double weight_foreground = total_pixels - weight_background;
        if (weight_foreground == 0) break;
        sum_background += count_f * (double)i;
        const double mean_background = sum_background / weight_background;
        const double mean_foreground = (sum_total - sum_background) / weight_foreground;
        const double diff = mean_background - mean_foreground;
        const double variance = weight_background * weight_foreground * diff * diff;
        if (variance > max_variance) { max_variance = variance; threshold = (uint8_t)i; }
    }
    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c)
            ((uint8_t *)dst->data)[r * dst->stride + c] = ((const uint8_t *)src->data)[r * src->stride + c] > threshold ? 255 : 0;
    *threshold_out = threshold;
    return 0;
}

ZO_API int zo_threshold_adaptive_mean(const zo_image *src, const zo_image *dst, uint32_t radius, float cc) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != ZO_U8 || dst->pixel != ZO_U8) return 2;
    if (radius == 0) return 3;
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    float *plane = (float *)calloc(n, 4), *sat = (float *)calloc(n, 4);
    if (!plane || !sat) { free(plane); free(sat); return 4; }
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) plane[r * cols + c] = (float)((const uint8_t *)src->data)[r * src->stride + c];
    zo_integral_plane_f32(plane, cols, sat, (uint32_t)rows, (uint32_t)cols);
    for (size_t r = 0; r < rows; ++r) {
        const size_t r1 = r > radius ? r - radius : 0, r2 = r + radius < rows - 1 ? r + radius : rows - 1;
        for (size_t c = 0; c < cols; ++c) {
            const size_t c1 = c > radius ? c - radius : 0, c2 = c + radius < cols - 1 ? c + radius : cols - 1;
            const float area = (float)((r2 - r1 + 1) * (c2 - c1 + 1));
            const float sum = sat[r2 * cols + c2] - (c1 > 0 ? sat[r2 * cols + (c1 - 1)] : 0) - (r1 > 0 ? sat[(r1 - 1) * cols + c2] : 0) +
                              ((r1 > 0 && c1 > 0) ? sat[(r1 - 1) * cols + (c1 - 1)] : 0);
            const float mean = sum / area;
            const float v = (float)((const uint8_t *)src->data)[r * src->stride + c];
            ((uint8_t *)dst->data)[r * dst->stride + c] = v > mean - cc ? 255 : 0;
        }
    }
    free(plane); free(sat);
    return 0;
}

static void apply_morph(const uint8_t *src, uint8_t *dst, size_t rows, size_t cols, const uint8_t *k, uint32_t kr, uint32_t kc, int erode) {
    const int ar = (int)(kr / 2), ac = (int)(kc / 2);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) {
            uint8_t value = erode ? 255 : 0;
            int done = 0;
            for (uint32_t i = 0; i < kr && !done; ++i)
                for (uint32_t j = 0; j < kc; ++j) {
                    if (k[i * kc + j] == 0) continue;
                    const long sr = (long)r + (long)i - ar, sc = (long)c + (long)j - ac;
                    const int inb = sr >= 0 && sc >= 0 && sr < (long)rows && sc < (long)cols;
                    if (!erode) { if (inb && src[(size_t)sr * cols + (size_t)sc] != 0) { value = 255; done = 1; break; } }
                    else if (!inb || src[(size_t)sr * cols + (size_t)sc] == 0) { value = 0; done = 1; break; }
                }
            dst[r * cols + c] = value;
        }
}
static void morph(const uint8_t *src, uint8_t *dst, size_t rows, size_t cols, const uint8_t *k, uint32_t kr, uint32_t kc, uint32_t iterations, int erode) {
    const size_t n = rows * cols;
    if (iterations == 0) { memcpy(dst, src, n); return; }
    uint8_t *a = (uint8_t *)malloc(n), *b = (uint8_t *)malloc(n);
    memcpy(a, src, n);
    for (uint32_t i = 0; i < iterations; ++i) { apply_morph(a, b, rows, cols, k, kr, kc, erode); uint8_t *t = a; a = b; b = t; }
    memcpy(dst, a, n);
    free(a); free(b);
}

/* op: 0 dilate, 1 erode, 2 open (erode then dilate), 3 close (dilate then erode); every stage runs `iterations` times */
ZO_API int zo_morph(const zo_image *src, const zo_image *dst, const uint8_t *kernel, uint32_t krows, uint32_t kcols, uint32_t iterations, int op) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != ZO_U8 || dst->pixel != ZO_U8) return 2;
    if (krows == 0 || kcols == 0 || krows % 2 == 0 || kcols % 2 == 0) return 3; /* InvalidKernelSize */
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    uint8_t *in = (uint8_t *)malloc(n), *mid = (uint8_t *)malloc(n), *res = (uint8_t *)malloc(n);
    for (size_t r = 0; r < rows; ++r) memcpy(in + r * cols, (const uint8_t *)src->data + r * src->stride, cols);
    if (op == 0 || op == 1) morph(in, res, rows, cols, kernel, krows, kcols, iterations, op);
    else if (iterations == 0) memcpy(res, in, n);
    else {
        morph(in, mid, rows, cols, kernel, krows, kcols, iterations, op == 2 ? 1 : 0);
        morph(mid, res, rows, cols, kernel, krows, kcols, iterations, op == 2 ? 0 : 1);
    }
    for (size_t r = 0; r < rows; ++r) memcpy((uint8_t *)dst->data + r * dst->stride, res + r * cols, cols);
    free(in); free(mid); free(res);
    return 0;
}
