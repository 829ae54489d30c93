// image_ops.hip — Image(T) container operations that move pixels without arithmetic:
//   copy            reference src/image.zig:375-392 (row-wise, honours views)
//   fill            reference src/image.zig:186-195
//   setBorder       reference src/image.zig:198-229
//   flipLeftRight / flipTopBottom   reference src/image/transforms.zig:28-44 (in place)
// All bit-exact by construction.
#include "zg_common.h"
#include <cstring>

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH,
               "copy: %ux%u vs %ux%u", src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "copy: pixel types differ");
    if (src->rows == 0 || src->cols == 0 || src->data == dst->data) return ZG_OK;
    const size_t ps = pixel_size(src->pixel);
    ZG_HIP(hipMemcpy2DAsync(dst->data, dst->stride * ps, src->data, src->stride * ps, (size_t)src->cols * ps,
                            src->rows, hipMemcpyDeviceToDevice, s));
    return ZG_OK;
}

struct PixelValue { uint8_t b[16]; };

// One thread per pixel; writes `value` where the pixel lies outside [l,r) x [t,b).
__global__ __launch_bounds__(256) void k_fill_outside(DImg img, int ps, PixelValue value, int l, int t, int r, int b) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int y = grid_row();
    if (c >= img.cols || y >= img.rows) return;
    if (y >= t && y < b && c >= l && c < r) return;
    uint8_t *p = (uint8_t *)img.data + ((size_t)y * img.stride + c) * ps;
    for (int i = 0; i < ps; ++i) p[i] = value.b[i];
}

int fill_outside_impl(const zg_image *img, const void *pixel_value, int l, int t, int r, int b, hipStream_t s) {
    int rc;
    if ((rc = check_image(img, "img"))) return rc;
    if (img->rows == 0 || img->cols == 0) return ZG_OK;
    PixelValue v{};
    const int ps = (int)pixel_size(img->pixel);
    if (pixel_value) std::memcpy(v.b, pixel_value, (size_t)ps);
    hipLaunchKernelGGL(k_fill_outside, row_grid(ceil_div(img->cols, 256), img->rows), dim3(256), 0, s, dimg(img), ps, v, l, t, r, b);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// setBorder: rect clipped to the image; an empty intersection fills everything (image.zig:200-204).
int set_border_impl(const zg_image *img, const uint32_t rect[4], const void *pixel_value, hipStream_t s) {
    uint32_t l = rect[0], t = rect[1], r = rect[2], b = rect[3];
    if (r > img->cols) r = img->cols;
    if (b > img->rows) b = img->rows;
    if (l >= r || t >= b) return fill_outside_impl(img, pixel_value, 0, 0, 0, 0, s);
    return fill_outside_impl(img, pixel_value, (int)l, (int)t, (int)r, (int)b, s);
}

template <int PS>
__global__ __launch_bounds__(256) void k_flip_lr(DImg img) {
    struct B { uint8_t b[PS]; };
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int y = grid_row();
    if (c >= img.cols / 2 || y >= img.rows) return;
    B *row = (B *)img.data + (size_t)y * img.stride;
    const B a = row[c], z = row[img.cols - 1 - c];
    row[c] = z;
    row[img.cols - 1 - c] = a;
}

template <int PS>
__global__ __launch_bounds__(256) void k_flip_tb(DImg img) {
    struct B { uint8_t b[PS]; };
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int y = grid_row(); // < rows / 2
    if (c >= img.cols || y >= img.rows / 2) return;
    B *top = (B *)img.data + (size_t)y * img.stride;
    B *bot = (B *)img.data + (size_t)(img.rows - 1 - y) * img.stride;
    const B a = top[c], z = bot[c];
    top[c] = z;
    bot[c] = a;
}


static int flip_impl(const zg_image *img, bool lr, hipStream_t s) {
    int rc;
    if ((rc = check_image(img, "img"))) return rc;
    if (img->rows == 0 || img->cols == 0) return ZG_OK;
    const int ps = (int)pixel_size(img->pixel);
    const dim3 grid_lr = row_grid(ceil_div(img->cols / 2 ? img->cols / 2 : 1, 256), img->rows);
    const dim3 grid_tb = row_grid(ceil_div(img->cols, 256), img->rows / 2);
#define ZG_FLIP(PS)                                                                              \
    case PS:                                                                                     \
        if (lr) hipLaunchKernelGGL(k_flip_lr<PS>, grid_lr, dim3(256), 0, s, dimg(img));          \
        else if (img->rows / 2) hipLaunchKernelGGL(k_flip_tb<PS>, grid_tb, dim3(256), 0, s, dimg(img)); \
        break;
    switch (ps) {
        ZG_FLIP(1) ZG_FLIP(3) ZG_FLIP(4) ZG_FLIP(12) ZG_FLIP(16)
    }
#undef ZG_FLIP
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Image(T).invert (image.zig:494-513): u8 scalars 255 - v; colour structs through their .invert() (color.zig:328-331,
// 441-444): max - channel with max = 255 / 1.0, Rgba keeps alpha. Image(f32) has no invert in the reference (compile error).
template <int PIX>
__global__ __launch_bounds__(256) void k_invert(DImg img) {
    using P = Px<PIX>;
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= img.cols || r >= img.rows) return;
    const size_t i = (size_t)r * img.stride + (size_t)c;
    typename P::Vec v = P::load(img.data, i);
    constexpr int N = P::C == 4 ? 3 : P::C;
#pragma unroll
    for (int ch = 0; ch < N; ++ch) {
        if constexpr (std::is_same<typename P::Elem, float>::value) v[ch] = 1.0f - v[ch];
        else v[ch] = (uint8_t)(255 - v[ch]);
    }
    P::store(img.data, i, v);
}

static int invert_impl(const zg_image *img, hipStream_t s) {
    int rc;
    if ((rc = check_image(img, "img"))) return rc;
    ZG_REQUIRE(img->pixel != ZG_PIXEL_F32, ZG_ERR_UNSUPPORTED, "invert: Image(f32) has no invert() in the reference (u8 scalars and colour structs only)");
    if (img->rows == 0 || img->cols == 0) return ZG_OK;
    return dispatch_pixel(img->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        hipLaunchKernelGGL((k_invert<PIX>), row_grid(ceil_div(img->cols, 256), img->rows), dim3(256), 0, s, dimg(img));
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_copy(const zg_image *src, const zg_image *dst, zg_stream stream) { return copy_impl(src, dst, as_stream(stream)); }

int zg_fill(const zg_image *img, const void *pixel_value, zg_stream stream) {
    return fill_outside_impl(img, pixel_value, 0, 0, 0, 0, as_stream(stream));
}

int zg_set_border(const zg_image *img, const uint32_t rect[4], const void *pixel_value, zg_stream stream) {
    int rc;
    if ((rc = check_image(img, "img"))) return rc;
    ZG_REQUIRE(rect, ZG_ERR_INVALID_ARGUMENT, "setBorder: null rect");
    return set_border_impl(img, rect, pixel_value, as_stream(stream));
}

int zg_fill_host(const zg_image *img, const void *pixel_value) {
    HostStage a;
    int rc;
    if ((rc = a.upload(img, false, true))) return rc;
    if ((rc = fill_outside_impl(&a.dev, pixel_value, 0, 0, 0, 0, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}
int zg_set_border_host(const zg_image *img, const uint32_t rect[4], const void *pixel_value) {
    HostStage a;
    int rc;
    ZG_REQUIRE(rect, ZG_ERR_INVALID_ARGUMENT, "setBorder: null rect");
    if ((rc = a.upload(img, true, true))) return rc; // the inside of rect keeps its pixels
    if ((rc = set_border_impl(&a.dev, rect, pixel_value, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}

int zg_invert(const zg_image *img, zg_stream stream) { return invert_impl(img, as_stream(stream)); }
int zg_invert_host(const zg_image *img) {
    HostStage a;
    int rc;
    if ((rc = a.upload(img, true, true))) return rc;
    if ((rc = invert_impl(&a.dev, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}

int zg_flip_left_right(const zg_image *img, zg_stream stream) { return flip_impl(img, true, as_stream(stream)); }
int zg_flip_top_bottom(const zg_image *img, zg_stream stream) { return flip_impl(img, false, as_stream(stream)); }

int zg_flip_left_right_host(const zg_image *img) {
    HostStage a;
    int rc;
    if ((rc = a.upload(img, true, true))) return rc;
    if ((rc = flip_impl(&a.dev, true, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}

int zg_flip_top_bottom_host(const zg_image *img) {
    HostStage a;
    int rc;
    if ((rc = a.upload(img, true, true))) return rc;
    if ((rc = flip_impl(&a.dev, false, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}

} // extern "C"
