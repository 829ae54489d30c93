/*
 * oracle/next.c — restatement of the first "next" rows of the scope table. TEST INFRASTRUCTURE ONLY (zo.h).
 *   src/image/edges.zig:13-70        Sobel: grey f32 -> two 3x3 f32 convolutions (.replicate) -> sqrt(gx^2+gy^2)/4 -> u8
 *   src/image/pyramid.zig:31-102     ImagePyramid.build: per level scale = pow(f, i), dims = trunc(dim / scale),
 *                                    sigma = blur_sigma * sqrt(scale^2 - 1), gaussianBlur if sigma > 0.5, bilinear resize
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int zo_convolve(const zo_image *src, const zo_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border);
int zo_convert(const zo_image *src, int src_space, const zo_image *dst, int dst_space, const float *srgb_lut);

ZO_API int zo_sobel(const zo_image *src, const zo_image *out) {
    if (src->rows != out->rows || src->cols != out->cols) return 1;
    if (out->pixel != ZO_U8) return 2;
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    float *gray = (float *)malloc(n * sizeof(float)), *gx = (float *)malloc(n * sizeof(float)), *gy = (float *)malloc(n * sizeof(float));
    if (src->pixel == ZO_F32) { /* scalar float input is used as is */
        for (size_t r = 0; r < rows; ++r)
            memcpy(gray + r * cols, (const float *)src->data + r * src->stride, cols * sizeof(float));
    } else { /* as(f32, convertColor(u8, pixel)) */
        uint8_t *g8 = (uint8_t *)malloc(n);
        zo_image g = {g8, cols, (uint32_t)rows, (uint32_t)cols, ZO_U8};
        const int ch = zo_channels(src->pixel);
        zo_convert(src, ch == 1 ? ZO_CS_GRAY : (ch == 4 ? ZO_CS_RGBA : ZO_CS_RGB), &g, ZO_CS_GRAY, NULL);
        for (size_t i = 0; i < n; ++i) gray[i] = (float)g8[i];
        free(g8);
    }
    static const float sobel_x[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, sobel_y[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
    zo_image gi = {gray, cols, (uint32_t)rows, (uint32_t)cols, ZO_F32}, xi = gi, yi = gi;
    xi.data = gx; yi.data = gy;
    zo_convolve(&gi, &xi, sobel_x, 3, 3, ZO_REPLICATE);
    zo_convolve(&gi, &yi, sobel_y, 3, 3, ZO_REPLICATE);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) {
            const float a = gx[r * cols + c], b = gy[r * cols + c];
            const float magnitude = sqrtf(a * a + b * b);
            const float scaled = magnitude / 4.0f;
            ((uint8_t *)out->data)[r * out->stride + c] = (uint8_t)truncf(fmaxf(0.0f, fminf(255.0f, scaled)));
        }
    free(gray); free(gx); free(gy);
    return 0;
}

ZO_API float zo_pyramid_scale(float scale_factor, uint32_t level) { return zo_powf(scale_factor, (float)level); }

/* returns 0 = level exists, 1 = pyramid truncated here (level smaller than 8 x 8) */
ZO_API int zo_pyramid_level(uint32_t rows, uint32_t cols, float scale, float blur_sigma, uint32_t *out_rows, uint32_t *out_cols, float *sigma) {
    uint32_t nr = (uint32_t)truncf((float)rows / scale), nc = (uint32_t)truncf((float)cols / scale);
    if (nr < 1) nr = 1;
    if (nc < 1) nc = 1;
    *out_rows = nr; *out_cols = nc;
    *sigma = blur_sigma * sqrtf(scale * scale - 1.0f);
    return (nr < 8 || nc < 8) ? 1 : 0;
}

/*
 * Canny (src/image.zig:1047-1063 -> src/image/edges.zig:212-277):
 *   grey      as(f32, convertColor(u8, px))                              :231-240  (0..255, every pixel type goes through u8)
 *   blur      own Gaussian: radius ceil(3 sigma), exp(-x^2 / (2 sigma^2)) / sum, convolveSeparable .replicate   :663-687
 *             (sigma == 0: copy)
 *   gradient  convolve(sobel_x), convolve(sobel_y), .replicate; magnitude sqrt(gx^2 + gy^2)                     :249-266
 *   NMS       direction quantised with K = tan(22.5 deg) without atan2; border pixels stay 0                      :692-763
 *   hysteresis  strong = nms && mag >= high; grow through 8-neighbours with nms && mag >= low                     :499-576
 * Returns 1 dimension mismatch, 2 bad output type, 3 invalid parameter / sigma / threshold.
 */
int zo_conv_separable(const zo_image *src, const zo_image *dst, const float *kx, uint32_t nkx, const float *ky, uint32_t nky, int border);

ZO_API int zo_canny(const zo_image *src, const zo_image *out, float sigma, float low, float high) {
    if (src->rows != out->rows || src->cols != out->cols) return 1;
    if (out->pixel != ZO_U8) return 2;
    if (!isfinite(sigma) || !isfinite(low) || !isfinite(high)) return 3;
    if (sigma < 0 || low < 0 || high < 0 || low >= high) return 3;
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    float *gray = (float *)malloc(n * 4), *blur = (float *)malloc(n * 4), *gx = (float *)malloc(n * 4), *gy = (float *)malloc(n * 4),
          *mag = (float *)malloc(n * 4);
    uint8_t *g8 = (uint8_t *)malloc(n), *nms = (uint8_t *)calloc(n, 1);
    {
        zo_image g = {g8, cols, (uint32_t)rows, (uint32_t)cols, ZO_U8};
        const int ch = zo_channels(src->pixel);
        zo_convert(src, ch == 1 ? ZO_CS_GRAY : (ch == 4 ? ZO_CS_RGBA : ZO_CS_RGB), &g, ZO_CS_GRAY, NULL);
        for (size_t i = 0; i < n; ++i) gray[i] = (float)g8[i];
    }
    zo_image gi = {gray, cols, (uint32_t)rows, (uint32_t)cols, ZO_F32}, bi = gi, xi = gi, yi = gi;
    bi.data = blur; xi.data = gx; yi.data = gy;
    if (sigma == 0) memcpy(blur, gray, n * 4);
    else {
        const size_t radius = (size_t)ceilf(3.0f * sigma), ks = 2 * radius + 1;
        float *k = (float *)malloc(ks * 4), sum = 0;
        for (size_t i = 0; i < ks; ++i) {
            const float x = (float)i - (float)radius;
            k[i] = zo_expf(-(x * x) / (2.0f * sigma * sigma));
            sum += k[i];
        }
        for (size_t i = 0; i < ks; ++i) k[i] /= sum;
        const int rc = zo_conv_separable(&gi, &bi, k, (uint32_t)ks, k, (uint32_t)ks, ZO_REPLICATE);
        free(k);
        if (rc) { free(gray); free(blur); free(gx); free(gy); free(mag); free(g8); free(nms); return 3; }
    }
    static const float sobel_x[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, sobel_y[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
    zo_convolve(&bi, &xi, sobel_x, 3, 3, ZO_REPLICATE);
    zo_convolve(&bi, &yi, sobel_y, 3, 3, ZO_REPLICATE);
    for (size_t i = 0; i < n; ++i) mag[i] = sqrtf(gx[i] * gx[i] + gy[i] * gy[i]);
    if (rows >= 3 && cols >= 3) {
        const float K = 0.414213562f;
        for (size_t r = 1; r + 1 < rows; ++r)
            for (size_t c = 1; c + 1 < cols; ++c) {
                const float a = gx[r * cols + c], b = gy[r * cols + c], ax = fabsf(a), ay = fabsf(b);
                int dr1, dc1, dr2, dc2;
                if (ay <= K * ax) { dr1 = 0; dc1 = -1; dr2 = 0; dc2 = 1; }
                else if (ax <= K * ay) { dr1 = -1; dc1 = 0; dr2 = 1; dc2 = 0; }
                else if (a * b > 0) { dr1 = -1; dc1 = 1; dr2 = 1; dc2 = -1; }
                else { dr1 = -1; dc1 = -1; dr2 = 1; dc2 = 1; }
                const float m = mag[r * cols + c], n1 = mag[(r + dr1) * cols + (c + dc1)], n2 = mag[(r + dr2) * cols + (c + dc2)];
                if (m >= n1 && m >= n2) nms[r * cols + c] = 255;
            }
    }
    uint8_t *o = (uint8_t *)out->data;
    for (size_t r = 0; r < rows; ++r) memset(o + r * out->stride, 0, cols);
    size_t *queue = (size_t *)malloc(n * sizeof(size_t)), push = 0, pop = 0;
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c)
            if (nms[r * cols + c] > 0 && mag[r * cols + c] >= high) { o[r * out->stride + c] = 255; queue[push++] = r * cols + c; }
    while (pop < push) {
        const size_t cur = queue[pop++], r = cur / cols, c = cur % cols;
        const size_t r0 = r > 0 ? r - 1 : 0, r1 = r + 2 < rows ? r + 2 : rows, c0 = c > 0 ? c - 1 : 0, c1 = c + 2 < cols ? c + 2 : cols;
        for (size_t nr = r0; nr < r1; ++nr)
            for (size_t nc = c0; nc < c1; ++nc) {
                if (nr == r && nc == c) continue;
                if (o[nr * out->stride + nc] > 0) continue;
                if (nms[nr * cols + nc] > 0 && mag[nr * cols + nc] >= low) { o[nr * out->stride + nc] = 255; queue[push++] = nr * cols + nc; }
            }
    }
    free(queue); free(gray); free(blur); free(gx); free(gy); free(mag); free(g8); free(nms);
    return 0;
}
