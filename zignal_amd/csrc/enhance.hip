// enhance.hip — Image(T).autocontrast / equalize for u8, Rgb(u8), Rgba(u8) (reference src/image.zig:804-829 ->
// src/image/enhancement.zig, cut-offs from src/image/histogram.zig:123-162). Both are "histogram -> per-channel 256-entry
// table -> remap in place": the histogram is built with LDS + global integer atomics, ONE workgroup turns it into the
// tables with the reference's integer / f32 arithmetic (nothing returns to the host), and a streaming kernel applies them.
//   autocontrast  lut[v] = round(f32(clamp(v, min, max) - min) / f32(max > min ? max - min : 1) * 255), min / max after
//                 dropping cutoff_pixels = trunc(f32(total) * cutoff) from each end; Rgba keeps its alpha
//   equalize      lut[v] = (cdf[v] - cdf_min) * 255 / (total - cdf_min) in u32 (identity when the denominator is 0);
//                 Rgba equalises alpha too
#include "zg_common.h"

#include <cmath>

namespace zg {

template <int PIX>
__global__ __launch_bounds__(256) void k_hist_channels(DImg img, unsigned int *hist) { // hist[C][256]
    using P = Px<PIX>;
    constexpr int C = P::C;
    __shared__ unsigned int lh[4][C][256]; // four copies against same-address serialisation
    for (int i = threadIdx.x; i < 4 * C * 256; i += 256) (&lh[0][0][0])[i] = 0;
    __syncthreads();
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    for (int step = 0; step < 16; ++step) { // 64 x 64 pixels per workgroup
        const int r = blockIdx.y * 64 + step * 4 + (int)(threadIdx.x >> 6);
        if (c < img.cols && r < img.rows) {
            const typename P::Vec v = P::load(img.data, (size_t)r * img.stride + (size_t)c);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) atomicAdd(&lh[threadIdx.x & 3][ch][v[ch]], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const unsigned int t = lh[0][ch][threadIdx.x] + lh[1][ch][threadIdx.x] + lh[2][ch][threadIdx.x] + lh[3][ch][threadIdx.x];
        if (t) atomicAdd(&hist[ch * 256 + threadIdx.x], t);
    }
}

// one workgroup: channel = threadIdx.x >> 6 would waste lanes; the scans are 256 steps, so lane 0 of each of `nlut` waves does one channel
__global__ __launch_bounds__(256) void k_enhance_luts(const unsigned int *hist, uint8_t *lut, int nch, int nlut, int equalize, unsigned int cutoff_pixels,
                                                      unsigned int total) {
    const int ch = threadIdx.x >> 6;
    if ((threadIdx.x & 63) != 0 || ch >= nch) return;
    const unsigned int *bins = hist + ch * 256;
    uint8_t *out = lut + ch * 256;
    if (ch >= nlut) { for (int i = 0; i < 256; ++i) out[i] = (uint8_t)i; return; } // autocontrast leaves alpha alone
    if (!equalize) {
        int mn = 255, mx = 0;
        if (cutoff_pixels == 0) {
            mn = 0;
            for (int i = 0; i < 256; ++i) if (bins[i] > 0) { mn = i; break; }
            for (int i = 255; i > 0; --i) if (bins[i] > 0) { mx = i; break; }
        } else {
            unsigned int cum = 0;
            for (int i = 0; i < 256; ++i) { cum += bins[i]; if (cum > cutoff_pixels) { mn = i; break; } }
            cum = 0;
            for (int i = 255; i > 0; --i) { cum += bins[i]; if (cum > cutoff_pixels) { mx = i; break; } }
        }
        const int range = mx > mn ? mx - mn : 1;
        for (int v = 0; v < 256; ++v) {
            const int lo = v < mx ? v : mx, clamped = mn > lo ? mn : lo;
            const float normalized = (float)((clamped - mn) & 255) / (float)range; // u8 arithmetic in the reference
            out[v] = (uint8_t)(int)roundf(normalized * 255.0f);
        }
    } else {
        unsigned int cdf_min = 0, run = 0;
        for (int i = 0; i < 256; ++i) { run += bins[i]; if (run > 0) { cdf_min = run; break; } }
        const unsigned int denominator = total - cdf_min;
        run = 0;
        for (int i = 0; i < 256; ++i) {
            run += bins[i];
            if (denominator == 0) out[i] = (uint8_t)i;
            else if (run >= cdf_min) out[i] = (uint8_t)(((run - cdf_min) * 255u) / denominator);
            else out[i] = 0;
        }
    }
}

template <int PIX>
__global__ __launch_bounds__(256) void k_apply_luts(DImg img, const uint8_t *lut) {
    using P = Px<PIX>;
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= img.cols) return;
    const size_t i = (size_t)r * img.stride + (size_t)c;
    typename P::Vec v = P::load(img.data, i);
#pragma unroll
    for (int ch = 0; ch < P::C; ++ch) v[ch] = lut[ch * 256 + v[ch]];
    P::store(img.data, i, v);
}

static int enhance_impl(const zg_image *img, bool equalize, float cutoff, hipStream_t s) {
    int rc;
    if ((rc = check_image(img, "img"))) return rc;
    ZG_REQUIRE(!pixel_is_float(img->pixel), ZG_ERR_UNSUPPORTED, "%s only supports u8, Rgb(u8) and Rgba(u8)", equalize ? "equalize" : "autocontrast");
    if (!equalize) ZG_REQUIRE(cutoff >= 0 && cutoff < 0.5f, ZG_ERR_INVALID_ARGUMENT, "autocontrast: InvalidCutoff (%g not in [0, 0.5))", (double)cutoff);
    const size_t total = (size_t)img->rows * img->cols;
    if (total == 0) return ZG_OK;
    ZG_REQUIRE(total <= 0xffffffffu / 255u, ZG_ERR_UNSUPPORTED, "autocontrast / equalize: %zu pixels overflow the reference's u32 arithmetic", total);
    const unsigned int cutoff_pixels = equalize ? 0u : (unsigned int)std::trunc((float)total * cutoff);
    const int nch = pixel_channels(img->pixel), nlut = equalize ? nch : (nch == 4 ? 3 : nch);
    char *scratch = nullptr;
    if ((rc = scratch_alloc((void **)&scratch, 4 * 256 * sizeof(unsigned int) + 4 * 256, s))) return rc;
    unsigned int *hist = (unsigned int *)scratch;
    uint8_t *lut = (uint8_t *)(hist + 4 * 256);
    if (hipMemsetAsync(hist, 0, 4 * 256 * sizeof(unsigned int), s) != hipSuccess) { scratch_free(scratch, s); ZG_HIP(hipErrorUnknown); }
    rc = dispatch_pixel(img->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if constexpr (!std::is_same<typename Px<PIX>::Elem, float>::value) {
            hipLaunchKernelGGL((k_hist_channels<PIX>), dim3(ceil_div(img->cols, 64), ceil_div(img->rows, 64)), dim3(256), 0, s, dimg(img), hist);
            hipLaunchKernelGGL(k_enhance_luts, dim3(1), dim3(256), 0, s, (const unsigned int *)hist, lut, nch, nlut, equalize ? 1 : 0, cutoff_pixels, (unsigned int)total);
            hipLaunchKernelGGL((k_apply_luts<PIX>), dim3(ceil_div(img->cols, 256), img->rows), dim3(256), 0, s, dimg(img), (const uint8_t *)lut);
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

int zg_autocontrast(const zg_image *img, float cutoff, zg_stream stream) { return enhance_impl(img, false, cutoff, as_stream(stream)); }
int zg_equalize(const zg_image *img, zg_stream stream) { return enhance_impl(img, true, 0.0f, as_stream(stream)); }

int zg_autocontrast_host(const zg_image *img, float cutoff) {
    HostStage a;
    int rc;
    if ((rc = a.upload(img, true, true))) return rc;
    if ((rc = enhance_impl(&a.dev, false, cutoff, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}
int zg_equalize_host(const zg_image *img) {
    HostStage a;
    int rc;
    if ((rc = a.upload(img, true, true))) return rc;
    if ((rc = enhance_impl(&a.dev, true, 0.0f, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}

} // extern "C"
