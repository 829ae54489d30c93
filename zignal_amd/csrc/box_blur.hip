// box_blur.hip — Image(T).boxBlur through an f32 summed-area table.
//
// Replaces reference src/image.zig:635-648 and src/image/integral.zig:41-78 (plane), :86-91 (sum), :194-269
// (boxBlurPlane). The SAT is built in f32 with the reference's exact association order — a running sum along
// each row, then rows accumulated top to bottom — so that its rounding (inexact above 2^24) is reproduced bit
// for bit; the window and its area are clipped at the borders and the mean goes through meta.clamp for u8.
// All SAT planes are complete before any output is written, so src may alias dst as in the reference.
//
//   k_sat_rows   one wave per 64 rows: 64x64 tiles staged through LDS (coalesced loads / stores), each lane
//                scans its row of the tile sequentially and carries the running sum to the next tile.
//   k_sat_cols   one thread per (column, channel): sequential accumulation down the rows, coalesced across lanes.
//   k_box_mean   one thread per pixel: four SAT taps, divide by the clipped area, clamp.
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);

template <int PIX>
__global__ __launch_bounds__(64) void k_sat_rows(DImg src, float *sat) { // sat: [C][rows][cols]
    using P = Px<PIX>;
    constexpr int C = P::C;
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64;
    const int ch = blockIdx.y;
    const size_t plane = (size_t)src.rows * src.cols;
    float run = 0.0f; // lane = row within the strip
    for (int c0 = 0; c0 < src.cols; c0 += 64) {
        for (int i = 0; i < 64; ++i) { // row i of the strip, 64 consecutive columns
            const int r = r0 + i, c = c0 + lane;
            float v = 0.0f;
            if (r < src.rows && c < src.cols) {
                const typename P::Elem *p = (const typename P::Elem *)src.data + ((size_t)r * src.stride + c) * C + ch;
                v = (float)*p;
            }
            tile[i][lane] = v;
        }
        __syncthreads();
        const int ncols = min(64, src.cols - c0);
        for (int j = 0; j < ncols; ++j) {
            run = run + tile[lane][j];
            tile[lane][j] = run;
        }
        __syncthreads();
        for (int i = 0; i < 64; ++i) {
            const int r = r0 + i, c = c0 + lane;
            if (r < src.rows && c < src.cols) sat[(size_t)ch * plane + (size_t)r * src.cols + c] = tile[i][lane];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_sat_cols(float *sat, int rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y;
    if (c >= cols) return;
    float *p = sat + (size_t)ch * rows * cols + c;
    float prev = p[0];
    for (int r = 1; r < rows; ++r) {
        const float cur = p[(size_t)r * cols] + prev; // dst[curr] += dst[prev]
        p[(size_t)r * cols] = cur;
        prev = cur;
    }
}

template <int PIX>
__global__ __launch_bounds__(256) void k_box_mean(const float *sat, DImg dst, int radius) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= dst.cols) return;
    const int rows = dst.rows, cols = dst.cols;
    const int r1 = max(r - radius, 0), r2 = (int)min((long long)r + radius, (long long)rows - 1);
    const int c1 = max(c - radius, 0), c2 = (int)min((long long)c + radius, (long long)cols - 1);
    const float area = (float)((long long)(r2 - r1 + 1) * (long long)(c2 - c1 + 1));
    Vec o;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const float *s = sat + (size_t)ch * rows * cols;
        const float a = s[(size_t)r2 * cols + c2];
        const float b = c1 > 0 ? s[(size_t)r2 * cols + (c1 - 1)] : 0.0f;
        const float d = r1 > 0 ? s[(size_t)(r1 - 1) * cols + c2] : 0.0f;
        const float e = (r1 > 0 && c1 > 0) ? s[(size_t)(r1 - 1) * cols + (c1 - 1)] : 0.0f;
        const float sum = a - b - d + e; // ((a - b) - d) + e, integral.zig:87-90
        const float val = sum / area;
        if constexpr (std::is_same<typename P::Elem, float>::value) o[ch] = val;
        else o[ch] = clamp_u8_f32(val);
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, o);
}

static int box_blur_impl(const zg_image *src, const zg_image *dst, uint32_t radius, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "boxBlur: %ux%u vs %ux%u",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "boxBlur: pixel types differ");
    if (radius == 0) return copy_impl(src, dst, s); // image.zig:639-642
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    ZG_REQUIRE(radius < (1u << 30), ZG_ERR_INVALID_ARGUMENT, "boxBlur: radius too large");
    const int C = pixel_channels(src->pixel);
    float *sat = nullptr;
    ZG_HIP(hipMallocAsync((void **)&sat, (size_t)C * src->rows * src->cols * sizeof(float), s));
    rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        hipLaunchKernelGGL((k_sat_rows<PIX>), dim3(ceil_div(src->rows, 64), (unsigned)C), dim3(64), 0, s, dimg(src), sat);
        hipLaunchKernelGGL(k_sat_cols, dim3(ceil_div(src->cols, 256), (unsigned)C), dim3(256), 0, s, sat, (int)src->rows, (int)src->cols);
        hipLaunchKernelGGL((k_box_mean<PIX>), dim3(ceil_div(dst->cols, 256), dst->rows), dim3(256), 0, s, (const float *)sat, dimg(dst), (int)radius);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
    ZG_HIP(hipFreeAsync(sat, s));
    return rc;
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_box_blur(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream) {
    return box_blur_impl(src, dst, radius, as_stream(stream));
}

int zg_box_blur_host(const zg_image *src, const zg_image *dst, uint32_t radius) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = box_blur_impl(&a.dev, &b.dev, radius, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
