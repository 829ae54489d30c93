// binary.hip — Image(u8) binarisation and binary morphology (reference src/image/binary.zig via src/image.zig:845-914).
//   thresholdOtsu          256-bin histogram (LDS + global integer atomics), between-class variance scanned in f64 by one
//                          lane exactly as the reference's loop (binary.zig:38-84), out = src > t ? 255 : 0; the threshold
//                          stays on the device between the kernels and is copied back only if the caller asks for it.
//   thresholdAdaptiveMean  the integral image of boxBlur (exact parallel row scan: u8 source) and out = src > mean - c
//                          (binary.zig:86-118).
//   dilate/erode/open/close  structuring element in the kernel arguments (<= 15 x 15, non-zero = on, centre anchor, not
//                          flipped); dilation ignores out-of-image samples, erosion treats them as background
//                          (binary.zig:121-281). Iterations ping-pong between scratch planes.
#include "zg_common.h"

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);
int sat_planes_impl(const zg_image *src, float *sat, hipStream_t s, bool integer_valued, size_t plane_stride = 0); // box_blur.hip (0: planes contiguous)

__global__ __launch_bounds__(256) void k_hist_u8(DImg src, unsigned int *hist) {
    __shared__ unsigned int lh[16][256];
#pragma unroll
    for (int k = 0; k < 16; ++k) lh[k][threadIdx.x] = 0;
    __syncthreads();
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    for (int step = 0; step < 16; ++step) { // 64 x 64 pixels per workgroup: few global atomics
        const int r = blockIdx.y * 64 + step * 4 + (int)(threadIdx.x >> 6);
        if (c < src.cols && r < src.rows) atomicAdd(&lh[threadIdx.x & 15][((const uint8_t *)src.data)[(size_t)r * src.stride + c]], 1u);
    }
    __syncthreads();
    unsigned int total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += lh[k][threadIdx.x];
    if (total) atomicAdd(&hist[threadIdx.x], total);
}

__global__ void k_otsu_threshold(const unsigned int *hist, uint8_t *threshold, double total_pixels) { // binary.zig:43-74
    if (threadIdx.x != 0) return;
    double sum_total = 0;
    for (int i = 0; i < 256; ++i) {
        const double term = (double)hist[i] * (double)i;
        sum_total = sum_total + term;
    }
    double sum_background = 0, weight_background = 0, max_variance = -1;
    int best = 0;
    for (int i = 0; i < 256; ++i) {
        const double count_f = (double)hist[i];
        weight_background = weight_background + count_f;
        if (weight_background == 0) continue;
        const double weight_foreground = total_pixels - weight_background;
        if (weight_foreground == 0) break;
        const double term = count_f * (double)i;
        sum_background = sum_background + term;
        const double mean_background = sum_background / weight_background;
        const double mean_foreground = (sum_total - sum_background) / weight_foreground;
        const double diff = mean_background - mean_foreground;
        const double variance = ((weight_background * weight_foreground) * diff) * diff;
        if (variance > max_variance) { max_variance = variance; best = i; }
    }
    *threshold = (uint8_t)best;
}

__global__ __launch_bounds__(256) void k_apply_threshold(DImg src, DImg dst, const uint8_t *threshold) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= src.cols) return;
    const uint8_t t = *threshold;
    ((uint8_t *)dst.data)[(size_t)r * dst.stride + c] = ((const uint8_t *)src.data)[(size_t)r * src.stride + c] > t ? 255 : 0;
}

__global__ __launch_bounds__(256) void k_adaptive_mean(const float *sat, DImg src, DImg dst, int radius, float cc) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    const int rows = src.rows, cols = src.cols;
    if (c >= cols) return;
    const int r1 = max(r - radius, 0), r2 = (int)min((long long)r + radius, (long long)rows - 1);
    const int c1 = max(c - radius, 0), c2 = (int)min((long long)c + radius, (long long)cols - 1);
    const float area = (float)((long long)(r2 - r1 + 1) * (long long)(c2 - c1 + 1));
    const float a = sat[(size_t)r2 * cols + c2];
    const float b = c1 > 0 ? sat[(size_t)r2 * cols + (c1 - 1)] : 0.0f;
    const float d = r1 > 0 ? sat[(size_t)(r1 - 1) * cols + c2] : 0.0f;
    const float e = (r1 > 0 && c1 > 0) ? sat[(size_t)(r1 - 1) * cols + (c1 - 1)] : 0.0f;
    const float mean = (a - b - d + e) / area;
    const float v = (float)((const uint8_t *)src.data)[(size_t)r * src.stride + c];
    ((uint8_t *)dst.data)[(size_t)r * dst.stride + c] = v > mean - cc ? 255 : 0;
}

struct MorphKernel { int rows, cols; uint8_t on[15 * 15]; };

template <bool ERODE>
__global__ __launch_bounds__(256) void k_morph(DImg src, DImg dst, MorphKernel k) { // applyMorph, binary.zig:230-280
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= src.cols || r >= src.rows) return;
    const int ar = k.rows / 2, ac = k.cols / 2;
    bool hit = false; // dilate: an on-element over a set pixel; erode: an on-element over background / outside
    for (int i = 0; i < k.rows && !hit; ++i) {
        const int sr = r + i - ar;
        for (int j = 0; j < k.cols; ++j) {
            if (!k.on[i * k.cols + j]) continue;
            const int sc = c + j - ac;
            const bool inb = sr >= 0 && sr < src.rows && sc >= 0 && sc < src.cols;
            const uint8_t v = inb ? ((const uint8_t *)src.data)[(size_t)sr * src.stride + sc] : (uint8_t)0;
            if (ERODE ? (v == 0) : (inb && v != 0)) { hit = true; break; }
        }
    }
    ((uint8_t *)dst.data)[(size_t)r * dst.stride + c] = ERODE ? (hit ? 0 : 255) : (hit ? 255 : 0);
}

static int check_u8_pair(const zg_image *src, const zg_image *dst, const char *what) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->pixel == ZG_PIXEL_U8 && dst->pixel == ZG_PIXEL_U8, ZG_ERR_UNSUPPORTED, "%s is only available for Image(u8)", what);
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "%s: %ux%u vs %ux%u", what, src->rows, src->cols, dst->rows, dst->cols);
    return ZG_OK;
}

static int otsu_impl(const zg_image *src, const zg_image *dst, uint8_t *threshold_host, hipStream_t s) {
    int rc;
    if ((rc = check_u8_pair(src, dst, "thresholdOtsu"))) return rc;
    if (threshold_host) *threshold_host = 0;
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    char *scratch = nullptr;
    if ((rc = scratch_alloc((void **)&scratch, 256 * sizeof(unsigned int) + 16, s))) return rc;
    unsigned int *hist = (unsigned int *)scratch;
    uint8_t *thr = (uint8_t *)(hist + 256);
    if (hipMemsetAsync(hist, 0, 256 * sizeof(unsigned int), s) != hipSuccess) { scratch_free(scratch, s); ZG_HIP(hipErrorUnknown); }
    hipLaunchKernelGGL(k_hist_u8, dim3(ceil_div(src->cols, 64), ceil_div(src->rows, 64)), dim3(256), 0, s, dimg(src), hist);
    hipLaunchKernelGGL(k_otsu_threshold, dim3(1), dim3(64), 0, s, (const unsigned int *)hist, thr, (double)((size_t)src->rows * src->cols));
    hipLaunchKernelGGL(k_apply_threshold, dim3(ceil_div(src->cols, 256), src->rows), dim3(256), 0, s, dimg(src), dimg(dst), (const uint8_t *)thr);
    rc = hipGetLastError() == hipSuccess ? ZG_OK : ZG_ERR_HIP;
    if (rc == ZG_OK && threshold_host) { // the return value of the reference's method: needs the stream to finish
        rc = download_pageable(threshold_host, thr, 1, s);
    }
    scratch_free(scratch, s);
    if (rc == ZG_ERR_HIP) set_error("thresholdOtsu: HIP failure");
    return rc;
}

static int adaptive_impl(const zg_image *src, const zg_image *dst, uint32_t radius, float c, hipStream_t s) {
    int rc;
    if ((rc = check_u8_pair(src, dst, "thresholdAdaptiveMean"))) return rc;
    ZG_REQUIRE(radius != 0, ZG_ERR_INVALID_ARGUMENT, "thresholdAdaptiveMean: InvalidRadius (0)");
    ZG_REQUIRE(radius < (1u << 30), ZG_ERR_INVALID_ARGUMENT, "thresholdAdaptiveMean: radius too large");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    float *sat = nullptr;
    if ((rc = scratch_alloc((void **)&sat, (size_t)src->rows * src->cols * sizeof(float), s))) return rc;
    if ((rc = sat_planes_impl(src, sat, s, true)) == ZG_OK) {
        hipLaunchKernelGGL(k_adaptive_mean, dim3(ceil_div(src->cols, 256), src->rows), dim3(256), 0, s, (const float *)sat, dimg(src), dimg(dst), (int)radius, c);
        if (hipGetLastError() != hipSuccess) { rc = ZG_ERR_HIP; set_error("thresholdAdaptiveMean: launch failed"); }
    }
    scratch_free(sat, s);
    return rc;
}

static void launch_morph(const zg_image *src, const zg_image *dst, const MorphKernel &k, bool erode, hipStream_t s) {
    const dim3 grid(ceil_div(src->cols, 64), ceil_div(src->rows, 4));
    if (erode) hipLaunchKernelGGL(k_morph<true>, grid, dim3(256), 0, s, dimg(src), dimg(dst), k);
    else hipLaunchKernelGGL(k_morph<false>, grid, dim3(256), 0, s, dimg(src), dimg(dst), k);
}

// `iterations` applications of one operation from src to dst through two scratch planes (src may alias dst)
static void morph_chain(const zg_image *src, const zg_image *dst, const MorphKernel &k, uint32_t iterations, bool erode, uint8_t *pa, uint8_t *pb, hipStream_t s) {
    zg_image a{pa, src->cols, src->rows, src->cols, ZG_PIXEL_U8}, b{pb, src->cols, src->rows, src->cols, ZG_PIXEL_U8};
    const zg_image *cur = src;
    for (uint32_t i = 0; i < iterations; ++i) {
        const bool last = i + 1 == iterations;
        // the last step writes dst; if dst is the buffer being read (in-place, single step) go through scratch first
        const zg_image *out = last ? dst : (cur == &a ? &b : &a);
        if (last && cur->data == dst->data) {
            launch_morph(cur, &a, k, erode, s);
            (void)copy_impl(&a, dst, s);
        } else {
            launch_morph(cur, out, k, erode, s);
        }
        cur = out;
    }
}

static int morph_impl(const zg_image *src, const zg_image *dst, const uint8_t *kernel, uint32_t krows, uint32_t kcols, uint32_t iterations, int op, hipStream_t s) {
    int rc;
    if ((rc = check_u8_pair(src, dst, "binary morphology"))) return rc;
    ZG_REQUIRE(op >= 0 && op <= 3, ZG_ERR_INVALID_ARGUMENT, "morphology: op %d (0 dilate, 1 erode, 2 open, 3 close)", op);
    ZG_REQUIRE(kernel && krows > 0 && kcols > 0 && krows % 2 == 1 && kcols % 2 == 1, ZG_ERR_INVALID_ARGUMENT, "morphology: InvalidKernelSize (%ux%u)", krows, kcols);
    ZG_REQUIRE(krows <= 15 && kcols <= 15, ZG_ERR_UNSUPPORTED, "morphology: structuring element %ux%u (15 x 15 supported)", krows, kcols);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    if (iterations == 0) return copy_impl(src, dst, s);
    MorphKernel k{};
    k.rows = (int)krows; k.cols = (int)kcols;
    for (uint32_t i = 0; i < krows * kcols; ++i) k.on[i] = kernel[i] != 0;
    const size_t n = (size_t)src->rows * src->cols;
    uint8_t *scratch = nullptr;
    if ((rc = scratch_alloc((void **)&scratch, 3 * n, s))) return rc;
    if (op == 0 || op == 1) morph_chain(src, dst, k, iterations, op == 1, scratch, scratch + n, s);
    else {
        const zg_image mid{scratch + 2 * n, src->cols, src->rows, src->cols, ZG_PIXEL_U8};
        morph_chain(src, &mid, k, iterations, op == 2, scratch, scratch + n, s);
        morph_chain(&mid, dst, k, iterations, op != 2, scratch, scratch + n, s);
    }
    rc = hipGetLastError() == hipSuccess ? ZG_OK : ZG_ERR_HIP;
    if (rc) set_error("morphology: launch failed");
    scratch_free(scratch, s);
    return rc;
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_threshold_otsu(const zg_image *src, const zg_image *dst, uint8_t *threshold_out, zg_stream stream) { return otsu_impl(src, dst, threshold_out, as_stream(stream)); }
int zg_threshold_otsu_host(const zg_image *src, const zg_image *dst, uint8_t *threshold_out) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = otsu_impl(&a.dev, &b.dev, threshold_out, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}
int zg_threshold_adaptive_mean(const zg_image *src, const zg_image *dst, uint32_t radius, float c, zg_stream stream) {
    return adaptive_impl(src, dst, radius, c, as_stream(stream));
}
int zg_threshold_adaptive_mean_host(const zg_image *src, const zg_image *dst, uint32_t radius, float c) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = adaptive_impl(&a.dev, &b.dev, radius, c, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}
int zg_morph(const zg_image *src, const zg_image *dst, const uint8_t *kernel, uint32_t kernel_rows, uint32_t kernel_cols, uint32_t iterations, int op,
             zg_stream stream) {
    return morph_impl(src, dst, kernel, kernel_rows, kernel_cols, iterations, op, as_stream(stream));
}
int zg_morph_host(const zg_image *src, const zg_image *dst, const uint8_t *kernel, uint32_t kernel_rows, uint32_t kernel_cols, uint32_t iterations, int op) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = morph_impl(&a.dev, &b.dev, kernel, kernel_rows, kernel_cols, iterations, op, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
