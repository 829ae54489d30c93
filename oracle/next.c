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

/*
 * Motion blur (src/image.zig:1077-1091 -> src/image/motion_blur.zig), SURVEY §8f rank 4.
 *   linear   :65-236   distance 0: copy; |sin| < 0.001 / |cos| < 0.001: convolveSeparable(uniform 1/n taps, identity,
 *                      .replicate); else per pixel, per field: samples t = -d/2, -d/2 + 1, ... <= d/2 along
 *                      (cos, sin), in-bounds ones bilinearly interpolated (x1/y1 clamped), mean; no sample: the pixel
 *   radial   :240-440  zoom: samples on the ray through the centre, scale 1 + t * amount * 0.1; spin: on the circle,
 *                      angle + t * amount; 8 + trunc(strength * 24) samples, t in [-0.5, 0.5]
 * cos_a / sin_a: @cos(angle) / @sin(angle) as the caller computes them. Integer fields: @round, clamp, @trunc.
 */
static float mb_get(const zo_image *im, size_t r, size_t c, int ch) {
    const int n = zo_channels(im->pixel);
    if (zo_is_float(im->pixel)) return ((const float *)im->data)[(r * im->stride + c) * n + ch];
    return (float)((const uint8_t *)im->data)[(r * im->stride + c) * n + ch];
}
static void mb_put(const zo_image *im, size_t r, size_t c, int ch, float v) {
    const int n = zo_channels(im->pixel);
    if (zo_is_float(im->pixel)) ((float *)im->data)[(r * im->stride + c) * n + ch] = v;
    else ((uint8_t *)im->data)[(r * im->stride + c) * n + ch] = (uint8_t)truncf(fmaxf(0.0f, fminf(255.0f, roundf(v))));
}
static int mb_sample(const zo_image *im, float sx, float sy, int ch, float *value) { /* bounds check + bilinear, :136-158 */
    if (!(sx >= 0 && sx < (float)im->cols && sy >= 0 && sy < (float)im->rows)) return 0;
    const size_t x0 = (size_t)floorf(sx), y0 = (size_t)floorf(sy);
    const size_t x1 = x0 + 1 < im->cols - 1 ? x0 + 1 : im->cols - 1, y1 = y0 + 1 < im->rows - 1 ? y0 + 1 : im->rows - 1;
    const float fx = sx - (float)x0, fy = sy - (float)y0;
    const float v00 = mb_get(im, y0, x0, ch), v10 = mb_get(im, y0, x1, ch), v01 = mb_get(im, y1, x0, ch), v11 = mb_get(im, y1, x1, ch);
    const float v0 = v00 * (1 - fx) + v10 * fx;
    const float v1 = v01 * (1 - fx) + v11 * fx;
    *value = v0 * (1 - fy) + v1 * fy;
    return 1;
}

ZO_API int zo_motion_blur_linear(const zo_image *src, const zo_image *dst, float angle, float cos_a, float sin_a, uint32_t distance) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel) return 2;
    const int n = zo_channels(src->pixel);
    if (distance == 0) {
        for (size_t r = 0; r < src->rows; ++r)
            memcpy((char *)dst->data + r * dst->stride * zo_pixel_size(dst->pixel), (const char *)src->data + r * src->stride * zo_pixel_size(src->pixel),
                   (size_t)src->cols * zo_pixel_size(src->pixel));
        return 0;
    }
    const float half_dist = (float)distance / 2.0f;
    if (fabsf(sin_a) < 0.001f || fabsf(cos_a) < 0.001f) {
        float *k = (float *)malloc(distance * sizeof(float));
        const float weight = 1.0f / (float)distance, identity = 1.0f;
        for (uint32_t i = 0; i < distance; ++i) k[i] = weight;
        const int rc = fabsf(sin_a) < 0.001f ? zo_conv_separable(src, dst, k, distance, &identity, 1, ZO_REPLICATE)
                                             : zo_conv_separable(src, dst, &identity, 1, k, distance, ZO_REPLICATE);
        free(k);
        return rc;
    }
    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c)
            for (int ch = 0; ch < n; ++ch) {
                float sum = 0, count = 0, t = -half_dist;
                for (uint32_t it = 0; it < distance + 2; ++it) {
                    if (t > half_dist) break;
                    float v;
                    if (mb_sample(src, (float)c + t * cos_a, (float)r + t * sin_a, ch, &v)) { sum += v; count += 1; }
                    t += 1.0f;
                }
                mb_put(dst, r, c, ch, count > 0 ? sum / count : mb_get(src, r, c, ch));
            }
    return 0;
}

ZO_API int zo_motion_blur_radial(const zo_image *src, const zo_image *dst, float center_x, float center_y, float strength, int spin) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel) return 2;
    const int n = zo_channels(src->pixel);
    if (strength == 0) {
        for (size_t r = 0; r < src->rows; ++r)
            memcpy((char *)dst->data + r * dst->stride * zo_pixel_size(dst->pixel), (const char *)src->data + r * src->stride * zo_pixel_size(src->pixel),
                   (size_t)src->cols * zo_pixel_size(src->pixel));
        return 0;
    }
    if (src->rows == 0 || src->cols == 0) return 0;
    const float cx = center_x * (float)(src->cols - 1), cy = center_y * (float)(src->rows - 1);
    const float s = fmaxf(0.0f, fminf(1.0f, strength));
    const size_t num_samples = 8 + (size_t)truncf(s * 24.0f);
    for (size_t r = 0; r < src->rows; ++r)
        for (size_t c = 0; c < src->cols; ++c) {
            const float dx = (float)c - cx, dy = (float)r - cy;
            const float dist = sqrtf(dx * dx + dy * dy), ang = zo_atan2f(dy, dx);
            const float max_distance = sqrtf(cx * cx + cy * cy);
            const float blur_amount = spin ? s * 0.5f : (dist / max_distance) * s * 20;
            for (int ch = 0; ch < n; ++ch) {
                float sum = 0;
                size_t count = 0;
                for (size_t k = 0; k < num_samples; ++k) {
                    const float t = ((float)k - (float)(num_samples - 1) / 2.0f) / (float)(num_samples - 1);
                    float sx, sy;
                    if (!spin) {
                        const float scale = 1.0f + t * blur_amount * 0.1f;
                        sx = cx + dx * scale;
                        sy = cy + dy * scale;
                    } else {
                        const float new_angle = ang + t * blur_amount;
                        sx = cx + dist * zo_cosf(new_angle);
                        sy = cy + dist * zo_sinf(new_angle);
                    }
                    float v;
                    if (mb_sample(src, sx, sy, ch, &v)) { sum += v; count += 1; }
                }
                mb_put(dst, r, c, ch, count > 0 ? sum / (float)count : mb_get(src, r, c, ch));
            }
        }
    return 0;
}

/*
 * Shen-Castan (src/image.zig:1015-1027 -> src/image/edges.zig:83-196, options src/image/ShenCastan.zig:9-45):
 *   grey (as canny) -> ISEF smoothing, rows then columns, each a forward + backward first-order recursion  :283-349
 *   -> BLI = (smoothed - grey >= 0) -> zero crossings (forward neighbours, or any 4-neighbour for NMS)       :356-415
 *   -> adaptive gradient |mean1 - mean0| of the window split by BLI, from three integral images             :417-497
 *   -> high threshold = first histogram bin whose cumulative count reaches floor(total * high_ratio)         :137-166
 *   -> optional NMS on central differences of the smoothed plane :582-661 -> strong-only emit or hysteresis :179-195
 * Returns 1 dimension mismatch, 2 bad output type, 3 invalid option (InvalidBParameter / WindowSizeMustBeOdd /
 * WindowSizeTooSmall / InvalidThreshold).
 */
int zo_integral_plane_f32(const float *src, size_t src_stride, float *sat, uint32_t rows, uint32_t cols);

static void isef_1d(float *data, size_t n, size_t stride, float b, float *temp) { /* edges.zig:283-305 on a strided line */
    if (n == 0) return;
    const float a = 1.0f - b;
    temp[0] = b * data[0];
    for (size_t i = 1; i < n; ++i) temp[i] = b * data[i * stride] + a * temp[i - 1];
    data[(n - 1) * stride] = temp[n - 1];
    if (n > 1)
        for (size_t i = n - 1; i-- > 0;) data[i * stride] = b * temp[i] + a * data[(i + 1) * stride];
}
/* isefFilter2D (edges.zig:308-349) on a contiguous f32 plane, in place: every row, then every column. Exposed so that the device's segmented
 * recursions can be held to the sequential ones bit for bit (test infrastructure, like the rest of this file). */
ZO_API int zo_isef_plane(float *plane, uint32_t rows, uint32_t cols, float smooth) {
    if (!(smooth > 0 && smooth < 1)) return 3;
    if (rows == 0 || cols == 0) return 0;
    float *tmp = (float *)malloc((size_t)(rows > cols ? rows : cols) * 4);
    for (size_t r = 0; r < rows; ++r) isef_1d(plane + r * cols, cols, 1, smooth, tmp);
    for (size_t c = 0; c < cols; ++c) isef_1d(plane + c, rows, cols, smooth, tmp);
    free(tmp);
    return 0;
}
static float sc_sat_sum(const float *sat, size_t stride, size_t r1, size_t c1, size_t r2, size_t c2) { /* integral.zig:85-90 */
    return sat[r2 * stride + c2] - (c1 > 0 ? sat[r2 * stride + (c1 - 1)] : 0) - (r1 > 0 ? sat[(r1 - 1) * stride + c2] : 0) +
           ((r1 > 0 && c1 > 0) ? sat[(r1 - 1) * stride + (c1 - 1)] : 0);
}

ZO_API int zo_shen_castan(const zo_image *src, const zo_image *out, float smooth, uint32_t window_size, float high_ratio, float low_rel,
                          int hysteresis, int use_nms) {
    if (src->rows != out->rows || src->cols != out->cols) return 1;
    if (out->pixel != ZO_U8) return 2;
    if (!(smooth > 0 && smooth < 1)) return 3;
    if (window_size % 2 == 0 || window_size < 3) return 3;
    if (!(high_ratio > 0 && high_ratio < 1) || !(low_rel > 0 && low_rel < 1)) return 3;
    const size_t rows = src->rows, cols = src->cols, n = rows * cols;
    if (n == 0) return 0;
    float *gray = (float *)malloc(n * 4), *sm = (float *)malloc(n * 4), *grad = (float *)calloc(n, 4), *tmp = (float *)malloc((rows > cols ? rows : cols) * 4);
    float *sat_g = (float *)malloc(n * 4), *sat_m = (float *)malloc(n * 4), *sat_gm = (float *)malloc(n * 4), *plane = (float *)malloc(n * 4);
    uint8_t *g8 = (uint8_t *)malloc(n), *bli = (uint8_t *)malloc(n), *edges = (uint8_t *)calloc(n, 1), *nms = (uint8_t *)calloc(n, 1);
    {
        zo_image g = {g8, cols, (uint32_t)rows, (uint32_t)cols, ZO_U8};
        const int ch = zo_channels(src->pixel);
        zo_convert(src, ch == 1 ? ZO_CS_GRAY : (ch == 4 ? ZO_CS_RGBA : ZO_CS_RGB), &g, ZO_CS_GRAY, NULL);
        for (size_t i = 0; i < n; ++i) gray[i] = (float)g8[i];
    }
    memcpy(sm, gray, n * 4);
    for (size_t r = 0; r < rows; ++r) isef_1d(sm + r * cols, cols, 1, smooth, tmp);
    for (size_t c = 0; c < cols; ++c) isef_1d(sm + c, rows, cols, smooth, tmp);
    for (size_t i = 0; i < n; ++i) bli[i] = (sm[i] - gray[i]) >= 0 ? 1 : 0;
    /* zero crossings */
    if (!use_nms) {
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const uint8_t ce = bli[r * cols + c];
                int mark = 0;
                if (!mark && c + 1 < cols) mark = ce != bli[r * cols + c + 1];
                if (!mark && r + 1 < rows) mark = ce != bli[(r + 1) * cols + c];
                if (!mark && r + 1 < rows && c + 1 < cols) mark = ce != bli[(r + 1) * cols + c + 1];
                if (!mark && r + 1 < rows && c > 0) mark = ce != bli[(r + 1) * cols + c - 1];
                if (mark) edges[r * cols + c] = 255;
            }
    } else if (rows >= 3 && cols >= 3) {
        for (size_t r = 1; r + 1 < rows; ++r)
            for (size_t c = 1; c + 1 < cols; ++c) {
                const uint8_t ce = bli[r * cols + c];
                if (ce != bli[r * cols + c - 1] || ce != bli[r * cols + c + 1] || ce != bli[(r - 1) * cols + c] || ce != bli[(r + 1) * cols + c])
                    edges[r * cols + c] = 255;
            }
    } else {
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) {
                const uint8_t ce = bli[r * cols + c];
                int mark = 0;
                if (!mark && c > 0) mark = ce != bli[r * cols + c - 1];
                if (!mark && c + 1 < cols) mark = ce != bli[r * cols + c + 1];
                if (!mark && r > 0) mark = ce != bli[(r - 1) * cols + c];
                if (!mark && r + 1 < rows) mark = ce != bli[(r + 1) * cols + c];
                if (mark) edges[r * cols + c] = 255;
            }
    }
    /* adaptive gradients from three integral images */
    zo_integral_plane_f32(gray, cols, sat_g, (uint32_t)rows, (uint32_t)cols);
    for (size_t i = 0; i < n; ++i) plane[i] = (float)bli[i];
    zo_integral_plane_f32(plane, cols, sat_m, (uint32_t)rows, (uint32_t)cols);
    for (size_t i = 0; i < n; ++i) plane[i] = gray[i] * (float)bli[i];
    zo_integral_plane_f32(plane, cols, sat_gm, (uint32_t)rows, (uint32_t)cols);
    const size_t hw = window_size / 2;
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) {
            if (edges[r * cols + c] == 0) continue;
            const size_t r1 = r > hw ? r - hw : 0, r2 = r + hw < rows - 1 ? r + hw : rows - 1;
            const size_t c1 = c > hw ? c - hw : 0, c2 = c + hw < cols - 1 ? c + hw : cols - 1;
            const float area = (float)((r2 - r1 + 1) * (c2 - c1 + 1));
            const float count1 = sc_sat_sum(sat_m, cols, r1, c1, r2, c2), count0 = area - count1;
            if (count0 > 0 && count1 > 0) {
                const float sum1 = sc_sat_sum(sat_gm, cols, r1, c1, r2, c2), sum_total = sc_sat_sum(sat_g, cols, r1, c1, r2, c2);
                const float sum0 = sum_total - sum1;
                const float mean0 = sum0 / count0, mean1 = sum1 / count1;
                grad[r * cols + c] = fabsf(mean1 - mean0);
            }
        }
    /* thresholds */
    size_t hist[256] = {0}, total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (edges[i] == 0) continue;
        float g = grad[i];
        if (g < 0) g = 0;
        if (g > 255) g = 255;
        hist[(size_t)roundf(g)] += 1;
        total += 1;
    }
    uint8_t *o = (uint8_t *)out->data;
    for (size_t r = 0; r < rows; ++r) memset(o + r * out->stride, 0, cols);
    int rc = 0;
    if (total != 0) {
        const size_t target = (size_t)floorf((float)total * high_ratio);
        size_t cum = 0, idx = 0;
        while (idx < 256 && cum < target) { cum += hist[idx]; idx += 1; }
        const float t_high = (float)(idx < 255 ? idx : 255), t_low = low_rel * t_high;
        const uint8_t *cand = edges;
        if (use_nms) {
            cand = nms;
            if (rows >= 3 && cols >= 3) {
                const float K = 0.414213562f;
                for (size_t r = 1; r + 1 < rows; ++r)
                    for (size_t c = 1; c + 1 < cols; ++c) {
                        if (edges[r * cols + c] == 0) continue;
                        const float gx = 0.5f * (sm[r * cols + c + 1] - sm[r * cols + c - 1]), gy = 0.5f * (sm[(r + 1) * cols + c] - sm[(r - 1) * cols + c]);
                        const float ax = fabsf(gx), ay = fabsf(gy);
                        int dr1, dc1, dr2, dc2;
                        if (ay <= K * ax) { dr1 = 0; dc1 = -1; dr2 = 0; dc2 = 1; }
                        else if (ax <= K * ay) { dr1 = -1; dc1 = 0; dr2 = 1; dc2 = 0; }
                        else if (gx * gy > 0) { dr1 = -1; dc1 = 1; dr2 = 1; dc2 = -1; }
                        else { dr1 = -1; dc1 = -1; dr2 = 1; dc2 = 1; }
                        const float m = grad[r * cols + c], n1 = grad[(r + dr1) * cols + (c + dc1)], n2 = grad[(r + dr2) * cols + (c + dc2)];
                        if (m >= n1 && m >= n2) nms[r * cols + c] = 255;
                    }
            }
        }
        if (!hysteresis) {
            for (size_t r = 0; r < rows; ++r)
                for (size_t c = 0; c < cols; ++c) o[r * out->stride + c] = (cand[r * cols + c] > 0 && grad[r * cols + c] >= t_high) ? 255 : 0;
        } else { /* applyHysteresis, edges.zig:499-576 */
            size_t *queue = (size_t *)malloc(n * sizeof(size_t)), push = 0, pop = 0;
            for (size_t r = 0; r < rows; ++r)
                for (size_t c = 0; c < cols; ++c)
                    if (cand[r * cols + c] > 0 && grad[r * cols + c] >= t_high) { o[r * out->stride + c] = 255; queue[push++] = r * cols + c; }
            while (pop < push) {
                const size_t cur = queue[pop++], r = cur / cols, c = cur % cols;
                const size_t r0 = r > 0 ? r - 1 : 0, r1 = r + 2 < rows ? r + 2 : rows, c0 = c > 0 ? c - 1 : 0, c1 = c + 2 < cols ? c + 2 : cols;
                for (size_t nr = r0; nr < r1; ++nr)
                    for (size_t nc = c0; nc < c1; ++nc) {
                        if ((nr == r && nc == c) || o[nr * out->stride + nc] > 0) continue;
                        if (cand[nr * cols + nc] > 0 && grad[nr * cols + nc] >= t_low) { o[nr * out->stride + nc] = 255; queue[push++] = nr * cols + nc; }
                    }
            }
            free(queue);
        }
    }
    free(gray); free(sm); free(grad); free(tmp); free(sat_g); free(sat_m); free(sat_gm); free(plane); free(g8); free(bli); free(edges); free(nms);
    return rc;
}
