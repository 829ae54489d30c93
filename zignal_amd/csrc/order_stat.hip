// order_stat.hip — Image(T).medianBlur / percentileBlur / minBlur / maxBlur / midpointBlur / alphaTrimmedMeanBlur for u8 and
// all-u8 struct pixels (reference src/image.zig:653-783 -> src/image/order_statistic_blur.zig, stats.percentile at
// src/image/histogram.zig:586-610).
//
// Every result is a function of the multiset of the (2r+1)^2 window samples (out-of-image samples through
// border.resolveIndex, a dropped one counting as 0), so the reference's sliding histograms are an implementation detail.
// Here a workgroup stages its 64 x 4 tile plus the radius in LDS (border resolved once per staged pixel) and each lane
// evaluates its window directly: the k-th smallest sample by an 8-step bisection on the VALUE (count of samples <= mid),
// min / max for the midpoint, and for the alpha-trimmed mean the two trim boundaries by bisection plus one pass of sums.
// Cost ~ 8 (2r+1)^2 byte reads per channel: right for the radii these filters are used with (median 3x3 .. 7x7); the
// window area and the rank / trim counts are constants of the call, computed on the host in f64 as the reference does.
#include "zg_common.h"

#include <cmath>

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);

constexpr int OS_MAXR = 15;
enum : int { OS_PERCENTILE = 0, OS_MIDPOINT = 1, OS_ALPHA_TRIMMED = 2 };

template <int PIX>
__global__ __launch_bounds__(256) void k_order_stat(DImg src, DImg dst, int radius, int border, int op, int rank, int trim_each, int tiles_x) {
    using P = Px<PIX>;
    constexpr int C = P::C;
    extern __shared__ uint8_t tile[]; // [C][lh][lw], lw = 64 + 2r, lh = 4 + 2r
    const int lw = 64 + 2 * radius, lh = 4 + 2 * radius;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * 64, y0 = ty * 4;
    for (int i = threadIdx.x; i < lh * lw; i += 256) {
        const int tr = i / lw, tc = i - tr * lw;
        const int gr = resolve_index(y0 - radius + tr, src.rows, border), gc = resolve_index(x0 - radius + tc, src.cols, border);
        typename P::Vec v = P::load(src.data, (size_t)max(gr, 0) * src.stride + (size_t)max(gc, 0)); // clamped, unpredicated
        if (gr < 0 || gc < 0) v = P::zero(); // dropped sample: value 0
#pragma unroll
        for (int ch = 0; ch < C; ++ch) tile[(ch * lh + tr) * lw + tc] = v[ch];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int c = x0 + lx, r = y0 + ly;
    if (c >= dst.cols || r >= dst.rows) return;
    const int w = 2 * radius + 1, area = w * w;
    typename P::Vec out;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const uint8_t *t = tile + (ch * lh + ly) * lw + lx; // window origin of this lane
        auto count_le = [&](int m) { // samples <= m
            int n = 0;
            for (int j = 0; j < w; ++j)
                for (int i = 0; i < w; ++i) n += t[j * lw + i] <= m ? 1 : 0;
            return n;
        };
        auto kth = [&](int k) { // value of the (k+1)-th smallest sample: smallest v with count(<= v) > k
            int lo = 0, hi = 255;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (count_le(mid) > k) hi = mid; else lo = mid + 1;
            }
            return lo;
        };
        int res;
        if (op == OS_PERCENTILE) {
            res = kth(rank);
        } else if (op == OS_MIDPOINT) {
            int mn = 255, mx = 0;
            for (int j = 0; j < w; ++j)
                for (int i = 0; i < w; ++i) { const int v = t[j * lw + i]; mn = min(mn, v); mx = max(mx, v); }
            res = (mn + mx + 1) / 2;
        } else { // alpha-trimmed mean: drop trim_each samples from each end (ties split by count), rounded mean of the rest
            int a = 0, b = 255;
            if (trim_each > 0) { a = kth(trim_each - 1); b = kth(area - trim_each); }
            long long total = 0, sum_lt = 0, sum_gt = 0;
            int cnt_lt = 0, cnt_gt = 0;
            for (int j = 0; j < w; ++j)
                for (int i = 0; i < w; ++i) {
                    const int v = t[j * lw + i];
                    total += v;
                    if (v < a) { sum_lt += v; ++cnt_lt; }
                    if (v > b) { sum_gt += v; ++cnt_gt; }
                }
            long long low = 0, high = 0;
            if (trim_each > 0) { low = sum_lt + (long long)(trim_each - cnt_lt) * a; high = sum_gt + (long long)(trim_each - cnt_gt) * b; }
            const long long kept = area - 2 * trim_each; // >= 1: area is odd and trim_each <= area / 2
            const long long rounded = ((total - low - high) + kept / 2) / kept;
            res = (int)(rounded > 255 ? 255 : rounded);
        }
        out[ch] = (uint8_t)res;
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, out);
}

static int order_stat_impl(const zg_image *src, const zg_image *dst, uint32_t radius, int op, double param, int border, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "order-statistic blur: %ux%u vs %ux%u", src->rows, src->cols,
               dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "order-statistic blur: pixel types differ");
    ZG_REQUIRE(!pixel_is_float(src->pixel), ZG_ERR_UNSUPPORTED, "order-statistic blur: UnsupportedPixelType (u8 and all-u8 structs only)");
    ZG_REQUIRE(border >= ZG_BORDER_ZERO && border <= ZG_BORDER_WRAP, ZG_ERR_INVALID_ARGUMENT, "invalid border %d", border);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    if (op == OS_ALPHA_TRIMMED) ZG_REQUIRE(std::isfinite(param) && param >= 0.0 && param < 0.5, ZG_ERR_INVALID_ARGUMENT, "alphaTrimmedMeanBlur: InvalidTrim (%g)", param);
    if (radius == 0) return copy_impl(src, dst, s);
    if (op == OS_PERCENTILE) ZG_REQUIRE(param >= 0.0 && param <= 1.0, ZG_ERR_INVALID_ARGUMENT, "percentileBlur: InvalidPercentile (%g)", param);
    ZG_REQUIRE(radius <= (uint32_t)OS_MAXR, ZG_ERR_UNSUPPORTED, "order-statistic blur: radius %u (up to %d supported)", radius, OS_MAXR);
    const long long area = (long long)(2 * radius + 1) * (2 * radius + 1);
    int rank = 0, trim_each = 0;
    if (op == OS_PERCENTILE) { // stats.percentile, histogram.zig:595-599 (total = area: every window sample counts)
        const double rank_floor = std::floor(param * (double)(area - 1) + 1e-12);
        long long rk = (long long)std::trunc(rank_floor);
        rank = (int)std::min<long long>(std::max<long long>(rk, 0), area - 1);
    } else if (op == OS_ALPHA_TRIMMED) { // order_statistic_blur.zig:331-334
        const long long trimmed_each = (long long)std::trunc(std::floor(param * (double)area));
        trim_each = (int)std::min<long long>(trimmed_each, area / 2);
    }
    // in place: other workgroups would read pixels this one has already replaced
    const zg_image *in = src;
    zg_image copy{};
    void *scratch = nullptr;
    const size_t ps = pixel_size(src->pixel);
    if (src->data == dst->data) {
        if ((rc = scratch_alloc(&scratch, (size_t)src->rows * src->cols * ps, s))) return rc;
        copy = zg_image{scratch, src->cols, src->rows, src->cols, src->pixel};
        if ((rc = copy_impl(src, &copy, s))) { scratch_free(scratch, s); return rc; }
        in = &copy;
    }
    const int tiles_x = (int)ceil_div(src->cols, 64), tiles_y = (int)ceil_div(src->rows, 4);
    const size_t lds = (size_t)pixel_channels(src->pixel) * (4 + 2 * radius) * (64 + 2 * radius);
    rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if constexpr (!std::is_same<typename Px<PIX>::Elem, float>::value) {
            hipLaunchKernelGGL((k_order_stat<PIX>), dim3((unsigned)(tiles_x * tiles_y)), dim3(256), lds, s, dimg(in), dimg(dst), (int)radius, border, op, rank,
                               trim_each, tiles_x);
            ZG_HIP(hipGetLastError());
        }
        return ZG_OK;
    });
    scratch_free(scratch, s);
    return rc;
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_order_statistic_blur(const zg_image *src, const zg_image *dst, uint32_t radius, int op, double param, int border, zg_stream stream) {
    ZG_REQUIRE(op >= OS_PERCENTILE && op <= OS_ALPHA_TRIMMED, ZG_ERR_INVALID_ARGUMENT, "order-statistic blur: op %d", op);
    return order_stat_impl(src, dst, radius, op, param, border, as_stream(stream));
}
int zg_order_statistic_blur_host(const zg_image *src, const zg_image *dst, uint32_t radius, int op, double param, int border) {
    ZG_REQUIRE(op >= OS_PERCENTILE && op <= OS_ALPHA_TRIMMED, ZG_ERR_INVALID_ARGUMENT, "order-statistic blur: op %d", op);
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = order_stat_impl(&a.dev, &b.dev, radius, op, param, border, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
