// edges.hip — Image(T).sobel as ONE kernel, plus the host arithmetic of ImagePyramid.build.
//
// Replaces reference src/image.zig:1001-1010 -> src/image/edges.zig:33-70: grey f32 plane (as(f32, convertColor(u8, px)),
// scalar f32 input used as is) -> convolve(sobel_x, .replicate) and convolve(sobel_y, .replicate) in f32 (ky-major, every
// tap including the zero ones, separate mul and add: src/image/convolution.zig:158-169) -> sqrt(gx^2 + gy^2) / 4 ->
// @trunc(@max(0, @min(255, .))). The reference materialises three full f32 planes; here a workgroup stages the grey values
// of its 64 x 4 tile plus a one-pixel replicate halo in LDS and writes only the u8 result: traffic = source once + 1 B/px.
#include "zg_common.h"
#include "zg_hostmath.h"

#include <cmath>

#pragma clang fp contract(off)

namespace zg {

__device__ inline float gray_as_f32(uint8_t v) { return (float)v; }

template <int PIX> __device__ inline float sobel_gray(typename Px<PIX>::Vec v) {
    using P = Px<PIX>;
    if constexpr (PIX == ZG_PIXEL_F32) return v[0];
    else if constexpr (PIX == ZG_PIXEL_U8) return (float)v[0];
    else if constexpr (!std::is_same<typename P::Elem, float>::value) { // Rgb(u8) / Rgba(u8): BT.709 16.16 fixed point (color.zig:1031-1042)
        const int y = (13933 * (int)v[0] + 46871 * (int)v[1] + 4732 * (int)v[2] + 32768) >> 16;
        return (float)(y < 0 ? 0 : (y > 255 ? 255 : y));
    } else { // Rgb(f32) / Rgba(f32): clamp(dot, 0, 1) then .as(u8) = @round(255 * clamp) (color.zig:1043-1046, :528-546)
        float y = 0.2126f * v[0] + 0.7152f * v[1] + 0.0722f * v[2];
        y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
        const float s = 255.0f * y;
        return (float)(int)roundf(s);
    }
}

template <int PIX>
__global__ __launch_bounds__(256) void k_sobel(DImg src, DImg dst, int tiles_x) {
    using P = Px<PIX>;
    __shared__ float g[6][66];
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int x0 = tx * 64, y0 = ty * 4;
    for (int i = threadIdx.x; i < 6 * 66; i += 256) {
        const int r = i / 66, c = i - r * 66;
        int gr = y0 - 1 + r, gc = x0 - 1 + c; // .replicate: clamp
        gr = gr < 0 ? 0 : (gr > src.rows - 1 ? src.rows - 1 : gr);
        gc = gc < 0 ? 0 : (gc > src.cols - 1 ? src.cols - 1 : gc);
        g[r][c] = sobel_gray<PIX>(P::load(src.data, (size_t)gr * src.stride + (size_t)gc));
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int c = x0 + lx, r = y0 + ly;
    if (c >= dst.cols || r >= dst.rows) return;
    const float kx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, ky[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
    float ax = 0.0f, ay = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float p = g[ly + j][lx + i];
            const float px = p * kx[j * 3 + i], py = p * ky[j * 3 + i];
            ax = ax + px;
            ay = ay + py;
        }
    const float sx = ax * ax, sy = ay * ay;
    const float magnitude = sqrtf(sx + sy);
    const float scaled = magnitude / 4.0f;
    const float clamped = fmaxf(0.0f, fminf(255.0f, scaled));
    ((uint8_t *)dst.data)[(size_t)r * dst.stride + (size_t)c] = (uint8_t)(int)truncf(clamped);
}

static int sobel_impl(const zg_image *src, const zg_image *dst, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "sobel: %ux%u vs %ux%u",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(dst->pixel == ZG_PIXEL_U8, ZG_ERR_INVALID_ARGUMENT, "sobel: the output is Image(u8)");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const int tiles_x = (int)ceil_div(src->cols, 64), tiles_y = (int)ceil_div(src->rows, 4);
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        hipLaunchKernelGGL((k_sobel<PIX>), dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, s, dimg(src), dimg(dst), tiles_x);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_sobel(const zg_image *src, const zg_image *dst, zg_stream stream) { return sobel_impl(src, dst, as_stream(stream)); }

int zg_sobel_host(const zg_image *src, const zg_image *dst) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = sobel_impl(&a.dev, &b.dev, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

// ImagePyramid.build's per-level arithmetic (src/image/pyramid.zig:57-80). `scale` is pow(scale_factor, level) as the
// CALLER's maths library computes it (a Zig host passes std.math.pow's value); zg_pyramid_scale is the library's own.
float zg_pyramid_scale(float scale_factor, uint32_t level) { return hostmath::pow_f32(scale_factor, (float)level); }

int zg_pyramid_level(uint32_t rows, uint32_t cols, float scale, float blur_sigma, uint32_t *out_rows, uint32_t *out_cols, float *out_sigma) {
    ZG_REQUIRE(out_rows && out_cols && out_sigma, ZG_ERR_INVALID_ARGUMENT, "pyramid level: null output");
    ZG_REQUIRE(scale > 0, ZG_ERR_INVALID_ARGUMENT, "pyramid level: scale must be positive");
    uint32_t nr = (uint32_t)std::trunc((float)rows / scale), nc = (uint32_t)std::trunc((float)cols / scale);
    *out_rows = nr < 1 ? 1 : nr;
    *out_cols = nc < 1 ? 1 : nc;
    *out_sigma = blur_sigma * std::sqrt(scale * scale - 1.0f);
    return ZG_OK;
}

} // extern "C"
