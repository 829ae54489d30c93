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
