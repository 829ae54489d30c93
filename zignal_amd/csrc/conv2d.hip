// conv2d.hip — Image(T).convolve: dense 2-D convolution with a small kernel.
//
// Replaces reference src/image/convolution.zig:76-301: dst = sum_ky sum_kx src[r+ky-hh, c+kx-hw] * k[ky][kx],
// ky-major, every tap (no skipping); f32: separate mul and add; u8: k = @round(k * 256) as i32, i64-exact
// accumulate, divClampU8(256). Out-of-range taps go through border.resolveIndex (the reference's interior fast
// path computes the same sums). Struct pixels run per channel on interleaved data (the reference's split /
// plane / merge, :213-293, gives the same bytes; its uniform-channel shortcut is value-preserving).
#include "zg_common.h"

#include <cmath>
#include <cstdlib>

#pragma clang fp contract(off)

namespace zg {

constexpr int MAX_K2D = 15 * 15;

struct Kernel2D {
    int kh, kw;
    union { float f[MAX_K2D]; int32_t i[MAX_K2D]; };
};

// MODE: 0 = f32, 1 = u8 with i32 accumulate (host proved it exact), 2 = u8 with i64 accumulate
template <int PIX, int MODE>
__global__ __launch_bounds__(256) void k_conv2d(DImg src, DImg dst, Kernel2D k, int border, int tiles_x) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int c = tx * 64 + (int)(threadIdx.x & 63);
    const int r = ty * 4 + (int)(threadIdx.x >> 6);
    if (c >= dst.cols || r >= dst.rows) return;
    const int hh = k.kh / 2, hw = k.kw / 2;
    using Acc = typename std::conditional<MODE == 0, float, typename std::conditional<MODE == 1, int32_t, int64_t>::type>::type;
    Acc acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
    for (int ky = 0; ky < k.kh; ++ky) {
        const int gr = resolve_index(r + ky - hh, src.rows, border);
        for (int kx = 0; kx < k.kw; ++kx) {
            const int gc = gr < 0 ? -1 : resolve_index(c + kx - hw, src.cols, border);
            Vec v = P::zero();
            if (gr >= 0 && gc >= 0) v = P::load(src.data, (size_t)gr * src.stride + (size_t)gc);
            if constexpr (MODE == 0) {
                const float w = k.f[ky * k.kw + kx];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) { const float p = v[ch] * w; acc[ch] = acc[ch] + p; }
            } else {
                const int32_t w = k.i[ky * k.kw + kx];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) acc[ch] += (Acc)v[ch] * (Acc)w;
            }
        }
    }
    Vec o;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        if constexpr (MODE == 0) o[ch] = acc[ch];
        else { // divClampU8(256): symmetric rounding divide, clamp
            const Acc a = acc[ch];
            if (a < 0) o[ch] = 0; // (a - 128) / 256 truncates to <= 0
            else { const Acc q = (a + 128) >> 8; o[ch] = (uint8_t)(q > 255 ? 255 : q); }
        }
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, o);
}

static int convolve_impl(const zg_image *src, const zg_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "convolve: %ux%u vs %ux%u",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "convolve: pixel types differ");
    ZG_REQUIRE(kernel && kh >= 1 && kw >= 1 && kh * kw <= (uint32_t)MAX_K2D, ZG_ERR_INVALID_ARGUMENT,
               "convolve: kernel %ux%u (at most %d taps)", kh, kw, MAX_K2D);
    ZG_REQUIRE(border >= ZG_BORDER_ZERO && border <= ZG_BORDER_WRAP, ZG_ERR_INVALID_ARGUMENT, "invalid border %d", border);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    Kernel2D k;
    k.kh = (int)kh;
    k.kw = (int)kw;
    const bool is_float = pixel_is_float(src->pixel);
    int mode = 0;
    if (is_float) {
        for (uint32_t i = 0; i < kh * kw; ++i) k.f[i] = kernel[i];
    } else {
        int64_t sum_abs = 0, max_abs = 0;
        for (uint32_t i = 0; i < kh * kw; ++i) {
            const float r = std::round(kernel[i] * 256.0f); // ConvolutionKernel.flatten (convolution.zig:95-113)
            ZG_REQUIRE(std::fabs(r) < 2147483648.0f, ZG_ERR_INVALID_ARGUMENT, "kernel[%u] does not fit i32 after scaling", i);
            k.i[i] = (int32_t)r;
            sum_abs += std::llabs((long long)k.i[i]);
            max_abs = std::max<int64_t>(max_abs, std::llabs((long long)k.i[i]));
        }
        mode = (255 * sum_abs < (int64_t)INT32_MAX - 256) ? 1 : 2;
    }
    const int tiles_x = (int)ceil_div(dst->cols, 64), tiles_y = (int)ceil_div(dst->rows, 4);
    const dim3 grid((unsigned)(tiles_x * tiles_y));
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if constexpr (std::is_same<typename Px<PIX>::Elem, float>::value) {
            hipLaunchKernelGGL((k_conv2d<PIX, 0>), grid, dim3(256), 0, s, dimg(src), dimg(dst), k, border, tiles_x);
        } else {
            if (mode == 1) hipLaunchKernelGGL((k_conv2d<PIX, 1>), grid, dim3(256), 0, s, dimg(src), dimg(dst), k, border, tiles_x);
            else hipLaunchKernelGGL((k_conv2d<PIX, 2>), grid, dim3(256), 0, s, dimg(src), dimg(dst), k, border, tiles_x);
        }
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_convolve(const zg_image *src, const zg_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border, zg_stream stream) {
    return convolve_impl(src, dst, kernel, kh, kw, border, as_stream(stream));
}

int zg_convolve_host(const zg_image *src, const zg_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = convolve_impl(&a.dev, &b.dev, kernel, kh, kw, border, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
