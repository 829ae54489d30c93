// motion.hip — Image(T).motionBlur.
//
// Replaces reference src/image.zig:1077-1091 -> src/image/motion_blur.zig:
//   linear (:65-236)   distance 0: copy. Horizontal / vertical motion (|sin| or |cos| < 0.001): convolveSeparable with a
//                      uniform 1/n kernel and an identity kernel, .replicate — the library's separable convolution.
//                      Otherwise one gather kernel: per pixel the samples t = -d/2, -d/2 + 1, ... <= d/2 along
//                      (cos, sin); the in-bounds ones are bilinearly interpolated and averaged, per field.
//   radial (:240-440)  zoom: samples on the ray through the centre (scale 1 + t * amount * 0.1); spin: on the circle
//                      through the pixel (angle + t * amount, with atan2 / cos / sin per pixel — zg_devmath.h).
// The reference loops fields outermost; sample positions do not depend on the field, so they are computed once per pixel
// and every field accumulates its own sum in the same sample order: same bits. f32 arithmetic as written in the
// reference (separate mul / add). Integer fields: @round, clamp to the type, @trunc.
#include "zg_common.h"
#include "zg_devmath.h"
#include "zg_hostmath.h"

#include <cmath>
#include <vector>

#pragma clang fp contract(off)

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);

template <int PIX> struct MbAcc {
    using P = Px<PIX>;
    static constexpr int C = P::C;
    float sum[C];
    __device__ void clear() {
#pragma unroll
        for (int i = 0; i < C; ++i) sum[i] = 0.0f;
    }
    // bounds check + bilinear (motion_blur.zig:136-158); returns whether the sample counted
    __device__ bool sample(const DImg &src, float sx, float sy) {
        if (!(sx >= 0 && sx < (float)src.cols && sy >= 0 && sy < (float)src.rows)) return false;
        const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
        const int x1 = min(x0 + 1, src.cols - 1), y1 = min(y0 + 1, src.rows - 1);
        const float fx = sx - (float)x0, fy = sy - (float)y0;
        const typename P::Vec p00 = P::load(src.data, (size_t)y0 * src.stride + x0), p10 = P::load(src.data, (size_t)y0 * src.stride + x1);
        const typename P::Vec p01 = P::load(src.data, (size_t)y1 * src.stride + x0), p11 = P::load(src.data, (size_t)y1 * src.stride + x1);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const float v00 = (float)p00[i], v10 = (float)p10[i], v01 = (float)p01[i], v11 = (float)p11[i];
            const float a0 = v00 * (1 - fx), b0 = v10 * fx, v0 = a0 + b0;
            const float a1 = v01 * (1 - fx), b1 = v11 * fx, v1 = a1 + b1;
            const float a = v0 * (1 - fy), b = v1 * fy;
            sum[i] = sum[i] + (a + b);
        }
        return true;
    }
    __device__ void finish(const DImg &src, const DImg &dst, int r, int c, float count) const {
        typename P::Vec o = P::zero();
        const typename P::Vec self = P::load(src.data, (size_t)r * src.stride + c);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const float res = count > 0 ? sum[i] / count : (float)self[i];
            if constexpr (std::is_same<typename P::Elem, float>::value) o[i] = res;
            else o[i] = (uint8_t)(int)truncf(fmaxf(0.0f, fminf(255.0f, roundf(res))));
        }
        P::store(dst.data, (size_t)r * dst.stride + c, o);
    }
};

template <int PIX>
__global__ __launch_bounds__(256) void k_motion_linear(DImg src, DImg dst, float cos_a, float sin_a, float half_dist, uint32_t loop_limit) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= src.cols || r >= src.rows) return;
    MbAcc<PIX> acc;
    acc.clear();
    float count = 0.0f, t = -half_dist;
    for (uint32_t it = 0; it < loop_limit; ++it) { // distance + 2: the reference's guard against a non-advancing t
        if (t > half_dist) break;
        const float ox = t * cos_a, oy = t * sin_a;
        if (acc.sample(src, (float)c + ox, (float)r + oy)) count = count + 1;
        t = t + 1.0f;
    }
    acc.finish(src, dst, r, c, count);
}

template <int PIX, bool SPIN>
__global__ __launch_bounds__(256) void k_motion_radial(DImg src, DImg dst, float cx, float cy, float strength, int num_samples) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= src.cols || r >= src.rows) return;
    const float dx = (float)c - cx, dy = (float)r - cy;
    const float dxx = dx * dx, dyy = dy * dy;
    const float dist = sqrtf(dxx + dyy);
    const float cxx = cx * cx, cyy = cy * cy;
    const float max_distance = sqrtf(cxx + cyy);
    float angle = 0.0f, blur_amount;
    if constexpr (SPIN) { angle = dev_atan2f(dy, dx); blur_amount = strength * 0.5f; }
    else blur_amount = (dist / max_distance) * strength * 20;
    MbAcc<PIX> acc;
    acc.clear();
    int count = 0;
    const float last = (float)(num_samples - 1);
    for (int k = 0; k < num_samples; ++k) {
        const float t = ((float)k - last / 2.0f) / last;
        float sx, sy;
        if constexpr (!SPIN) {
            const float scale = 1.0f + t * blur_amount * 0.1f;
            sx = cx + dx * scale;
            sy = cy + dy * scale;
        } else {
            const float new_angle = angle + t * blur_amount;
            sx = cx + dist * dev_cosf(new_angle);
            sy = cy + dist * dev_sinf(new_angle);
        }
        if (acc.sample(src, sx, sy)) ++count;
    }
    acc.finish(src, dst, r, c, (float)count);
}

static int motion_check(const zg_image *src, const zg_image *dst, const char *what) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "%s: %ux%u vs %ux%u", what, src->rows, src->cols,
               dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "%s: pixel types differ", what);
    return ZG_OK;
}

static int motion_linear_impl(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a, uint32_t distance, zg_stream stream) {
    (void)angle;
    hipStream_t s = as_stream(stream);
    int rc;
    if ((rc = motion_check(src, dst, "motionBlur.linear"))) return rc;
    if (distance == 0) return copy_impl(src, dst, s);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    if (std::fabs(sin_a) < 0.001f || std::fabs(cos_a) < 0.001f) { // motion_blur.zig:77-118
        ZG_REQUIRE(distance <= (1u << 22), ZG_ERR_INVALID_ARGUMENT, "motionBlur.linear: distance %u is out of range", distance); // any length the separable convolution takes
        std::vector<float> k(distance, 1.0f / (float)distance);
        const float identity = 1.0f;
        return std::fabs(sin_a) < 0.001f ? zg_conv_separable(src, dst, k.data(), distance, &identity, 1, ZG_BORDER_REPLICATE, stream)
                                         : zg_conv_separable(src, dst, &identity, 1, k.data(), distance, ZG_BORDER_REPLICATE, stream);
    }
    const dim3 grid(ceil_div(src->cols, 64), ceil_div(src->rows, 4));
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        hipLaunchKernelGGL((k_motion_linear<PIX>), grid, dim3(256), 0, s, dimg(src), dimg(dst), cos_a, sin_a, (float)distance / 2.0f, distance + 2);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

static int motion_radial_impl(const zg_image *src, const zg_image *dst, float center_x, float center_y, float strength, int spin, zg_stream stream) {
    hipStream_t s = as_stream(stream);
    int rc;
    if ((rc = motion_check(src, dst, "motionBlur.radial"))) return rc;
    ZG_REQUIRE(spin == 0 || spin == 1, ZG_ERR_INVALID_ARGUMENT, "motionBlur.radial: type %d (0 zoom, 1 spin)", spin);
    if (strength == 0) return copy_impl(src, dst, s);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const float cx = center_x * (float)(src->cols - 1), cy = center_y * (float)(src->rows - 1);
    const float clamped = std::fmax(0.0f, std::fmin(1.0f, strength));
    const int num_samples = 8 + (int)std::trunc(clamped * 24.0f);
    const dim3 grid(ceil_div(src->cols, 64), ceil_div(src->rows, 4));
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if (spin) hipLaunchKernelGGL((k_motion_radial<PIX, true>), grid, dim3(256), 0, s, dimg(src), dimg(dst), cx, cy, clamped, num_samples);
        else hipLaunchKernelGGL((k_motion_radial<PIX, false>), grid, dim3(256), 0, s, dimg(src), dimg(dst), cx, cy, clamped, num_samples);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_motion_blur_linear(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a, uint32_t distance, zg_stream stream) {
    return motion_linear_impl(src, dst, angle, cos_a, sin_a, distance, stream);
}
int zg_motion_blur_linear_host(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a, uint32_t distance) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = motion_linear_impl(&a.dev, &b.dev, angle, cos_a, sin_a, distance, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}
int zg_motion_blur_radial(const zg_image *src, const zg_image *dst, float center_x, float center_y, float strength, int spin, zg_stream stream) {
    return motion_radial_impl(src, dst, center_x, center_y, strength, spin, stream);
}
int zg_motion_blur_radial_host(const zg_image *src, const zg_image *dst, float center_x, float center_y, float strength, int spin) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = motion_radial_impl(&a.dev, &b.dev, center_x, center_y, strength, spin, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
