/*
 * oracle/enhance.c — restatement of src/image/enhancement.zig (u8, Rgb(u8), Rgba(u8); in place). TEST INFRASTRUCTURE ONLY.
 *   autocontrast :11-80   per-channel histogram, cutoff_pixels = trunc(f32(total) * cutoff), min / max from
 *                         Histogram.findCutoffMin / Max (src/image/histogram.zig:123-162; the Max scan never looks at
 *                         bin 0), value -> round(f32(clamp(v) - min) / f32(range) * 255); Rgba keeps alpha
 *   equalize     :84-250  per-channel CDF, lut[i] = (cdf[i] - cdf_min) * 255 / (total - cdf_min) (identity when the
 *                         denominator is 0); Rgba equalises alpha too
 * The reference has no known-answer test for these two (only shape checks in its Python tests): parity rests on this
 * restatement of the integer / f32 arithmetic alone.
 */
#include "zo.h"
#include <math.h>
#include <string.h>

static void histogram(const zo_image *img, uint32_t hist[4][256]) {
    memset(hist, 0, 4 * 256 * sizeof(uint32_t));
    const int nch = zo_channels(img->pixel);
    for (size_t r = 0; r < img->rows; ++r)
        for (size_t c = 0; c < img->cols; ++c)
            for (int ch = 0; ch < nch; ++ch) hist[ch][((const uint8_t *)img->data)[(r * img->stride + c) * nch + ch]] += 1;
}
static uint8_t cutoff_min(const uint32_t bins[256], uint32_t cutoff) {
    if (cutoff == 0) { for (int i = 0; i < 256; ++i) if (bins[i] > 0) return (uint8_t)i; return 0; }
    uint32_t cum = 0;
    for (int i = 0; i < 256; ++i) { cum += bins[i]; if (cum > cutoff) return (uint8_t)i; }
    return 255;
}
static uint8_t cutoff_max(const uint32_t bins[256], uint32_t cutoff) {
    if (cutoff == 0) { for (int i = 255; i > 0; --i) if (bins[i] > 0) return (uint8_t)i; return 0; }
    uint32_t cum = 0;
    for (int i = 255; i > 0; --i) { cum += bins[i]; if (cum > cutoff) return (uint8_t)i; }
    return 0;
}
static void apply_luts(const zo_image *img, uint8_t lut[4][256], int nlut) {
    const int nch = zo_channels(img->pixel);
    for (size_t r = 0; r < img->rows; ++r)
        for (size_t c = 0; c < img->cols; ++c)
            for (int ch = 0; ch < nlut; ++ch) {
                uint8_t *p = (uint8_t *)img->data + (r * img->stride + c) * nch + ch;
                *p = lut[ch][*p];
            }
}

ZO_API int zo_autocontrast(const zo_image *img, float cutoff) {
    if (zo_is_float(img->pixel)) return 5;
    if (cutoff < 0 || cutoff >= 0.5f || cutoff != cutoff) return 3; /* error.InvalidCutoff */
    const size_t total = (size_t)img->rows * img->cols;
    const uint32_t cutoff_pixels = (uint32_t)truncf((float)total * cutoff);
    uint32_t hist[4][256];
    histogram(img, hist);
    const int nch = zo_channels(img->pixel), nlut = nch == 4 ? 3 : nch;
    uint8_t lut[4][256];
    for (int ch = 0; ch < nlut; ++ch) {
        const uint8_t mn = cutoff_min(hist[ch], cutoff_pixels), mx = cutoff_max(hist[ch], cutoff_pixels);
        const uint8_t range = mx > mn ? (uint8_t)(mx - mn) : 1;
        for (int v = 0; v < 256; ++v) {
            const uint8_t lo = (uint8_t)v < mx ? (uint8_t)v : mx;      /* @min(max_val, val) */
            const uint8_t clamped = mn > lo ? mn : lo;                  /* @max(min_val, .) */
            const float normalized = (float)(uint8_t)(clamped - mn) / (float)range;
            lut[ch][v] = (uint8_t)roundf(normalized * 255.0f);
        }
    }
    apply_luts(img, lut, nlut);
    return 0;
}

ZO_API int zo_equalize(const zo_image *img) {
    if (zo_is_float(img->pixel)) return 5;
    const uint32_t total = (uint32_t)((size_t)img->rows * img->cols);
    uint32_t hist[4][256];
    histogram(img, hist);
    const int nch = zo_channels(img->pixel);
    uint8_t lut[4][256];
    for (int ch = 0; ch < nch; ++ch) {
        uint32_t cdf[256];
        cdf[0] = hist[ch][0];
        for (int i = 1; i < 256; ++i) cdf[i] = cdf[i - 1] + hist[ch][i];
        uint32_t cdf_min = 0;
        for (int i = 0; i < 256; ++i) if (cdf[i] > 0) { cdf_min = cdf[i]; break; }
        const uint32_t denominator = total - cdf_min;
        for (int i = 0; i < 256; ++i) {
            if (denominator == 0) lut[ch][i] = (uint8_t)i;
            else if (cdf[i] >= cdf_min) lut[ch][i] = (uint8_t)((uint32_t)((cdf[i] - cdf_min) * 255u) / denominator); /* u32 arithmetic as in the reference */
            else lut[ch][i] = 0;
        }
    }
    apply_luts(img, lut, nch);
    return 0;
}
