/*
 * oracle/order_stat.c — restatement of src/image/order_statistic_blur.zig (u8 and all-u8 struct pixels, per channel).
 * TEST INFRASTRUCTURE ONLY (zo.h).
 *   window histogram of the (2r+1)^2 samples around a pixel, out-of-image samples through border.resolveIndex (a dropped
 *   sample counts as value 0: constantHistogram / getPixel, :291-307), then a reducer:
 *     percentile  stats.percentile (src/image/histogram.zig:586-610): rank = clamp(trunc(floor(p * (total - 1) + 1e-12))),
 *                 first value whose cumulative count exceeds it; median = 0.5 with .mirror, min = 0.0, max = 1.0
 *     midpoint    (min + max + 1) / 2                                                   (:318-325)
 *     alpha-trimmed mean  drop trunc(floor(trim * area)) samples (at most area / 2) from each end, rounded mean (:327-375)
 * The reference maintains the histogram incrementally (column histograms, Huang's sliding window); the window multiset,
 * hence every result, is the same as this direct evaluation.
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint8_t reduce(const uint32_t hist[256], size_t area, int op, double param) {
    if (op == 0) { /* percentile */
        size_t total = 0;
        for (int i = 0; i < 256; ++i) total += hist[i];
        if (total == 0) return 0;
        const size_t tm1 = total - 1;
        const double rank_floor = floor(param * (double)tm1 + 1e-12);
        size_t rank = (size_t)trunc(rank_floor);
        if (rank > tm1) rank = tm1;
        size_t cum = 0;
        for (int v = 0; v < 256; ++v) {
            if (hist[v] == 0) continue;
            cum += hist[v];
            if (cum > rank) return (uint8_t)v;
        }
        return 255;
    }
    if (op == 1) { /* midpoint */
        int mn = -1, mx = -1;
        for (int v = 0; v < 256; ++v) if (hist[v] > 0) { mn = v; break; }
        for (int v = 255; v >= 0; --v) if (hist[v] > 0) { mx = v; break; }
        if (mn < 0) mn = 0;
        if (mx < 0) mx = mn;
        return (uint8_t)((mn + mx + 1) / 2);
    }
    /* alpha-trimmed mean */
    const double total_f = (double)area;
    const size_t trimmed_each = (size_t)trunc(floor(param * total_f));
    const size_t trim_each = trimmed_each < area / 2 ? trimmed_each : area / 2;
    uint64_t total_sum = 0, low_sum = 0, high_sum = 0;
    size_t low_count = 0, high_count = 0, remaining = trim_each;
    for (int v = 0; v < 256; ++v) total_sum += (uint64_t)hist[v] * (uint64_t)v;
    for (int v = 0; v < 256 && remaining > 0; ++v) {
        const size_t take = hist[v] < remaining ? hist[v] : remaining;
        low_sum += (uint64_t)take * (uint64_t)v; low_count += take; remaining -= take;
    }
    remaining = trim_each;
    for (int v = 255; v >= 0 && remaining > 0; --v) {
        if (hist[v] == 0) continue;
        const size_t take = hist[v] < remaining ? hist[v] : remaining;
        high_sum += (uint64_t)take * (uint64_t)v; high_count += take; remaining -= take;
    }
    const size_t kept = area - low_count - high_count;
    if (kept == 0) return 0;
    const uint64_t rounded = ((total_sum - low_sum - high_sum) + (uint64_t)kept / 2) / (uint64_t)kept;
    return (uint8_t)(rounded > 255 ? 255 : rounded);
}

/* op: 0 percentile (param = fraction), 1 midpoint, 2 alpha-trimmed mean (param = trim fraction) */
ZO_API int zo_order_statistic_blur(const zo_image *src, const zo_image *dst, uint32_t radius, int op, double param, int border) {
    if (src->rows != dst->rows || src->cols != dst->cols) return 1;
    if (src->pixel != dst->pixel) return 2;
    if (zo_is_float(src->pixel)) return 5; /* error.UnsupportedPixelType */
    if (src->rows == 0 || src->cols == 0) return 0;
    if (op == 2 && (!isfinite(param) || param < 0.0 || param >= 0.5)) return 3; /* error.InvalidTrim (checked before radius 0) */
    const int nch = zo_channels(src->pixel);
    const size_t rows = src->rows, cols = src->cols;
    if (radius == 0) {
        for (size_t r = 0; r < rows; ++r) memcpy((uint8_t *)dst->data + r * dst->stride * nch, (const uint8_t *)src->data + r * src->stride * nch, cols * nch);
        return 0;
    }
    if (op == 0 && (param < 0.0 || param > 1.0 || param != param)) return 3; /* error.InvalidPercentile */
    uint8_t *in = (uint8_t *)malloc(rows * cols * nch);
    for (size_t r = 0; r < rows; ++r) memcpy(in + r * cols * nch, (const uint8_t *)src->data + r * src->stride * nch, cols * nch);
    const long rad = (long)radius;
    const size_t area = (size_t)(2 * radius + 1) * (2 * radius + 1);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c)
            for (int ch = 0; ch < nch; ++ch) {
                uint32_t hist[256];
                memset(hist, 0, sizeof hist);
                for (long dy = -rad; dy <= rad; ++dy)
                    for (long dx = -rad; dx <= rad; ++dx) {
                        const int64_t rr = zo_resolve_index((int64_t)r + dy, (int64_t)rows, border), cc = zo_resolve_index((int64_t)c + dx, (int64_t)cols, border);
                        hist[(rr >= 0 && cc >= 0) ? in[((size_t)rr * cols + (size_t)cc) * nch + ch] : 0] += 1;
                    }
                ((uint8_t *)dst->data)[(r * dst->stride + c) * nch + ch] = reduce(hist, area, op, param);
            }
    free(in);
    return 0;
}
