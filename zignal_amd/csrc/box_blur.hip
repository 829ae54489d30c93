// box_blur.hip — Image(T).boxBlur through an f32 summed-area table.
//
// Replaces reference src/image.zig:635-648 and src/image/integral.zig:41-78 (plane), :86-91 (sum), :194-269
// (boxBlurPlane). The SAT is built in f32 with the reference's exact association order — a running sum along
// each row, then rows accumulated top to bottom — so that its rounding (inexact above 2^24) is reproduced bit
// for bit; the window and its area are clipped at the borders and the mean goes through meta.clamp for u8.
// All SAT planes are complete before any output is written, so src may alias dst as in the reference.
//
//   k_sat_rows   one wave per (64 rows, channel): 64x64 tiles staged through LDS (coalesced, loads batched), each lane
//                scans its row of the tile sequentially and carries the running sum to the next tile.
//   k_sat_cols   one thread per (column, channel): sequential accumulation down the rows, coalesced across lanes,
//                16 rows prefetched ahead of the chain.
// The two scans are sequential f32 chains by contract (the reference's rounding order), so their parallelism is capped
// at rows x channels and columns x channels chains; they are latency-bound, not bandwidth-bound.
//   k_box_mean   one thread per pixel: four SAT taps, divide by the clipped area, clamp.
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);

// Row pass. The running sum along a row is a sequential f32 chain (f32 addition is not associative and the reference's
// rounding must be reproduced), so the parallelism is rows x channels. One wave owns 64 rows of one channel: per chunk
// of 64 columns it loads the 64 x 64 tile (lanes = consecutive columns: coalesced; the next chunk's loads are issued
// before this chunk is processed), transposes through LDS, each lane runs its row's 64-step chain in registers carrying
// the sum across chunks, and the results go back out transposed (coalesced again).
template <int PIX>
__global__ __launch_bounds__(64) void k_sat_rows(DImg src, float *sat) { // sat: [C][rows][cols]
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C;
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64;
    const int ch = blockIdx.y;
    const size_t plane = (size_t)src.rows * src.cols;
    const Elem *base = (const Elem *)src.data;
    const int nrows = min(64, src.rows - r0);

    auto fetch = [&](int c0, float (&v)[64]) {
        // addresses are clamped into the image instead of predicating the loads: a predicated load makes the compiler
        // wait for each one before issuing the next (measured 14 us per chunk); rows / columns past the edge are never stored
        const int c = min(c0 + lane, src.cols - 1);
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = (float)base[((size_t)(r0 + min(i, nrows - 1)) * src.stride + c) * C + ch];
    };

    float cur[64], nxt[64];
    fetch(0, cur);
    float run = 0.0f; // lane = row within the strip
    for (int c0 = 0; c0 < src.cols; c0 += 64) {
        fetch(c0 + 64, nxt); // clamped: harmless past the last chunk
#pragma unroll
        for (int i = 0; i < 64; ++i) tile[i][lane] = cur[i];
        __syncthreads();
        float rowv[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) rowv[j] = tile[lane][j];
#pragma unroll
        for (int j = 0; j < 64; ++j) { // the chain values produced by columns past the edge are never stored
            run = run + rowv[j];
            rowv[j] = run;
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) tile[lane][j] = rowv[j];
        __syncthreads();
        const int c = c0 + lane;
        if (c < src.cols) {
            float *o = sat + (size_t)ch * plane + (size_t)r0 * src.cols + c;
#pragma unroll
            for (int i = 0; i < 64; ++i)
                if (i < nrows) o[(size_t)i * src.cols] = tile[i][lane];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 64; ++i) cur[i] = nxt[i];
    }
}

// Column pass: sat[r][c] += sat[r-1][c], a sequential chain down each column (coalesced across lanes). Only columns x
// channels / 64 waves exist, so the achieved bandwidth is (bytes in flight) / latency: rows are taken 128 at a time with
// the following 128 already in flight (32 rows: 177 us, 128 rows: 157 us per 4096^2 plane; scalar row addressing: 195 us).
__global__ __launch_bounds__(64) void k_sat_cols(float *sat, int rows, int cols) {
    constexpr int G = 128;
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int ch = blockIdx.y;
    if (c >= cols) return;
    float *p = sat + (size_t)ch * rows * cols + c;
    auto fetch = [&](int r, float (&v)[G]) {
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = p[(size_t)min(r + i, rows - 1) * cols]; // clamped, not predicated (see k_sat_rows)
    };
    float prev = 0.0f;
    auto chain_store = [&](int r, float (&v)[G]) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (r + i < rows) {
                if (r + i > 0) v[i] = v[i] + prev; // dst[curr] += dst[prev], rows 1..
                prev = v[i];
                p[(size_t)(r + i) * cols] = v[i];
            }
        }
    };
    float a[G], b[G];
    fetch(0, a);
    for (int r = 0; r < rows; r += 2 * G) {
        fetch(r + G, b);
        chain_store(r, a);
        fetch(r + 2 * G, a);
        chain_store(r + G, b);
    }
}

// SHARPEN: Integral.sharpen (integral.zig:273-323, 325-426): 2 * original - blurred instead of the mean itself.
template <int PIX, bool SHARPEN>
__global__ __launch_bounds__(256) void k_box_mean(const float *sat, DImg src, DImg dst, int radius) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= dst.cols || r >= dst.rows) return;
    const int rows = dst.rows, cols = dst.cols;
    const int r1 = max(r - radius, 0), r2 = (int)min((long long)r + radius, (long long)rows - 1);
    const int c1 = max(c - radius, 0), c2 = (int)min((long long)c + radius, (long long)cols - 1);
    const float area = (float)((long long)(r2 - r1 + 1) * (long long)(c2 - c1 + 1));
    Vec o, orig = P::zero();
    if constexpr (SHARPEN) orig = P::load(src.data, (size_t)r * src.stride + (size_t)c);
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const float *s = sat + (size_t)ch * rows * cols;
        const float a = s[(size_t)r2 * cols + c2];
        const float b = c1 > 0 ? s[(size_t)r2 * cols + (c1 - 1)] : 0.0f;
        const float d = r1 > 0 ? s[(size_t)(r1 - 1) * cols + c2] : 0.0f;
        const float e = (r1 > 0 && c1 > 0) ? s[(size_t)(r1 - 1) * cols + (c1 - 1)] : 0.0f;
        const float sum = a - b - d + e; // ((a - b) - d) + e, integral.zig:87-90
        float val = sum / area;
        if constexpr (SHARPEN) {
            const float original = (float)orig[ch];
            const float twice = 2 * original;
            val = twice - val;
        }
        if constexpr (std::is_same<typename P::Elem, float>::value) o[ch] = val;
        else o[ch] = clamp_u8_f32(val);
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, o);
}

// Row pass for sources whose row sums are exact in f32 — integer-valued elements with cols * max < 2^24 (every u8 pixel type,
// and the detectors' grey / mask planes): each partial sum is then an integer below 2^24, every association gives the
// same bits as the reference's left-to-right loop, and the row becomes a parallel prefix sum (one workgroup per row and
// channel, 1024 columns per step) at memory speed instead of a 4096-step chain (316 us -> see DESIGN.md).
template <int PIX>
__global__ __launch_bounds__(256) void k_sat_rows_exact(DImg src, float *sat) { // sat: [C][rows][cols]
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C;
    __shared__ float wsum[4];
    const int r = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const Elem *row = (const Elem *)src.data + (size_t)r * src.stride * C + ch;
    float *out = sat + ((size_t)ch * src.rows + r) * src.cols;
    float carry = 0.0f;
    for (int c0 = 0; c0 < src.cols; c0 += 1024) {
        const int c = c0 + 4 * t;
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = (float)row[(size_t)min(c + k, src.cols - 1) * C]; // clamped, unpredicated
#pragma unroll
        for (int k = 0; k < 4; ++k) if (c + k >= src.cols) p[k] = 0.0f;
        p[1] += p[0]; p[2] += p[1]; p[3] += p[2];
        float x = p[3]; // inclusive scan of the per-thread totals across the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        float base = carry + (x - p[3]);
        if (w > 0) base += wsum[0];
        if (w > 1) base += wsum[1];
        if (w > 2) base += wsum[2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c + k < src.cols) out[c + k] = base + p[k];
        carry += ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
        __syncthreads();
    }
}

// Integral image(s) of `src` (Image(T).Integral.compute, integral.zig:95-140): one f32 plane of rows x cols per channel,
// planar, in the reference's association order. Also used by the Shen-Castan detector (edges.hip).
// `integer_valued`: the caller knows every element is an integer in [0, 255] (always true for u8 pixels), which makes the
// row sums exact and lets the row pass run as a parallel scan.
int sat_planes_impl(const zg_image *src, float *sat, hipStream_t s, bool integer_valued) {
    const int C = pixel_channels(src->pixel);
    const bool exact_rows = (integer_valued || !pixel_is_float(src->pixel)) && src->cols <= 65536; // 65536 * 255 < 2^24
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if (exact_rows) hipLaunchKernelGGL((k_sat_rows_exact<PIX>), dim3(src->rows, (unsigned)C), dim3(256), 0, s, dimg(src), sat);
        else hipLaunchKernelGGL((k_sat_rows<PIX>), dim3(ceil_div(src->rows, 64), (unsigned)C), dim3(64), 0, s, dimg(src), sat);
        hipLaunchKernelGGL(k_sat_cols, dim3(ceil_div(src->cols, 64), (unsigned)C), dim3(64), 0, s, sat, (int)src->rows, (int)src->cols);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

static int box_blur_impl(const zg_image *src, const zg_image *dst, uint32_t radius, bool sharpen, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "%s: %ux%u vs %ux%u", sharpen ? "sharpen" : "boxBlur",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "boxBlur / sharpen: pixel types differ");
    if (radius == 0) return copy_impl(src, dst, s); // image.zig:639-642
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    ZG_REQUIRE(radius < (1u << 30), ZG_ERR_INVALID_ARGUMENT, "boxBlur: radius too large");
    const int C = pixel_channels(src->pixel);
    float *sat = nullptr;
    if ((rc = scratch_alloc((void **)&sat, (size_t)C * src->rows * src->cols * sizeof(float), s))) return rc;
    if ((rc = sat_planes_impl(src, sat, s, false)) == ZG_OK)
        rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
            constexpr int PIX = decltype(tag)::value;
            if (sharpen) hipLaunchKernelGGL((k_box_mean<PIX, true>), row_grid(ceil_div(dst->cols, 256), dst->rows), dim3(256), 0, s, (const float *)sat, dimg(src), dimg(dst), (int)radius);
            else hipLaunchKernelGGL((k_box_mean<PIX, false>), row_grid(ceil_div(dst->cols, 256), dst->rows), dim3(256), 0, s, (const float *)sat, dimg(src), dimg(dst), (int)radius);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
    scratch_free(sat, s);
    return rc;
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_box_blur(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream) {
    return box_blur_impl(src, dst, radius, false, as_stream(stream));
}

int zg_box_blur_host(const zg_image *src, const zg_image *dst, uint32_t radius) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = box_blur_impl(&a.dev, &b.dev, radius, false, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

int zg_sharpen(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream) {
    return box_blur_impl(src, dst, radius, true, as_stream(stream));
}

int zg_sharpen_host(const zg_image *src, const zg_image *dst, uint32_t radius) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = box_blur_impl(&a.dev, &b.dev, radius, true, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

// Image(T).integral (image.zig:628-630 -> integral.zig:95-140): planes[ch] is a rows x cols f32 image, ch-major in `planes`.
int zg_integral(const zg_image *src, float *planes, zg_stream stream) {
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    ZG_REQUIRE(planes != nullptr || src->rows == 0 || src->cols == 0, ZG_ERR_INVALID_ARGUMENT, "integral: null output");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    return sat_planes_impl(src, planes, as_stream(stream), false);
}

int zg_integral_host(const zg_image *src, float *planes) {
    HostStage a;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const size_t bytes = (size_t)pixel_channels(src->pixel) * src->rows * src->cols * sizeof(float);
    ZG_REQUIRE(planes != nullptr, ZG_ERR_INVALID_ARGUMENT, "integral: null output");
    float *dev = nullptr;
    ZG_HIP(hipMalloc((void **)&dev, bytes));
    rc = sat_planes_impl(&a.dev, dev, nullptr, false);
    if (rc == ZG_OK) rc = download_pageable(planes, dev, bytes, nullptr);
    (void)hipFree(dev);
    return rc;
}

} // extern "C"
