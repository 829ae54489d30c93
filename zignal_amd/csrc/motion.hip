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

#include <algorithm>
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

// blockIdx.z: the frame of a batch (the pipeline's motion-blur step, batch.hip), fr.src_frame / fr.dst_frame bytes from one frame to the next
template <int PIX>
__global__ __launch_bounds__(256) void k_motion_linear(DImg src, DImg dst, float cos_a, float sin_a, float half_dist, uint32_t loop_limit, FrameSpan fr) {
    src.data = (char *)src.data + (size_t)blockIdx.z * fr.src_frame;
    dst.data = (char *)dst.data + (size_t)blockIdx.z * fr.dst_frame;
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
__global__ __launch_bounds__(256) void k_motion_radial(DImg src, DImg dst, float cx, float cy, float strength, int num_samples, FrameSpan fr) {
    src.data = (char *)src.data + (size_t)blockIdx.z * fr.src_frame;
    dst.data = (char *)dst.data + (size_t)blockIdx.z * fr.dst_frame;
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

// n equally shaped frames src_frame / dst_frame bytes apart (n = 1: one image). The gather kernels take the batch in gridDim.z (up to 65 535 frames
// per launch); the axis-aligned linear case is the separable convolution, frame by frame.
static int motion_linear_frames_impl(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, float cos_a, float sin_a, uint32_t distance,
                                     zg_stream stream) {
    hipStream_t s = as_stream(stream);
    int rc;
    if ((rc = motion_check(src, dst, "motionBlur.linear"))) return rc;
    auto frame_of = [](const zg_image *im, size_t step, uint32_t f) { zg_image one = *im; one.data = (char *)im->data + (size_t)f * step; return one; };
    const bool axis = std::fabs(sin_a) < 0.001f || std::fabs(cos_a) < 0.001f; // motion_blur.zig:77-118
    if (distance != 0 && axis) ZG_REQUIRE(distance <= (1u << 22), ZG_ERR_INVALID_ARGUMENT, "motionBlur.linear: distance %u is out of range", distance); // any length the separable convolution takes
    if (distance == 0 || axis) {
        std::vector<float> k(distance, distance ? 1.0f / (float)distance : 0.0f);
        const float identity = 1.0f;
        for (uint32_t f = 0; f < n; ++f) {
            const zg_image a = frame_of(src, src_frame, f), b = frame_of(dst, dst_frame, f);
            if (distance == 0) rc = copy_impl(&a, &b, s);
            else if (src->rows == 0 || src->cols == 0) rc = ZG_OK;
            else rc = std::fabs(sin_a) < 0.001f ? zg_conv_separable(&a, &b, k.data(), distance, &identity, 1, ZG_BORDER_REPLICATE, stream)
                                               : zg_conv_separable(&a, &b, &identity, 1, k.data(), distance, ZG_BORDER_REPLICATE, stream);
            if (rc) return rc;
        }
        return ZG_OK;
    }
    if (src->rows == 0 || src->cols == 0 || n == 0) return ZG_OK;
    const FrameSpan fr{src_frame, dst_frame};
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        for (uint32_t f0 = 0; f0 < n; f0 += 65535u) {
            const zg_image a = frame_of(src, src_frame, f0), b = frame_of(dst, dst_frame, f0);
            const dim3 grid(ceil_div(src->cols, 64), ceil_div(src->rows, 4), std::min(n - f0, 65535u));
            hipLaunchKernelGGL((k_motion_linear<PIX>), grid, dim3(256), 0, s, dimg(&a), dimg(&b), cos_a, sin_a, (float)distance / 2.0f, distance + 2, fr);
        }
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

static int motion_radial_frames_impl(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, float center_x, float center_y, float strength,
                                     int spin, zg_stream stream) {
    hipStream_t s = as_stream(stream);
    int rc;
    if ((rc = motion_check(src, dst, "motionBlur.radial"))) return rc;
    ZG_REQUIRE(spin == 0 || spin == 1, ZG_ERR_INVALID_ARGUMENT, "motionBlur.radial: type %d (0 zoom, 1 spin)", spin);
    auto frame_of = [](const zg_image *im, size_t step, uint32_t f) { zg_image one = *im; one.data = (char *)im->data + (size_t)f * step; return one; };
    if (strength == 0) {
        for (uint32_t f = 0; f < n; ++f) {
            const zg_image a = frame_of(src, src_frame, f), b = frame_of(dst, dst_frame, f);
            if ((rc = copy_impl(&a, &b, s))) return rc;
        }
        return ZG_OK;
    }
    if (src->rows == 0 || src->cols == 0 || n == 0) return ZG_OK;
    const float cx = center_x * (float)(src->cols - 1), cy = center_y * (float)(src->rows - 1);
    const float clamped = std::fmax(0.0f, std::fmin(1.0f, strength));
    const int num_samples = 8 + (int)std::trunc(clamped * 24.0f);
    const FrameSpan fr{src_frame, dst_frame};
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        for (uint32_t f0 = 0; f0 < n; f0 += 65535u) {
            const zg_image a = frame_of(src, src_frame, f0), b = frame_of(dst, dst_frame, f0);
            const dim3 grid(ceil_div(src->cols, 64), ceil_div(src->rows, 4), std::min(n - f0, 65535u));
            if (spin) hipLaunchKernelGGL((k_motion_radial<PIX, true>), grid, dim3(256), 0, s, dimg(&a), dimg(&b), cx, cy, clamped, num_samples, fr);
            else hipLaunchKernelGGL((k_motion_radial<PIX, false>), grid, dim3(256), 0, s, dimg(&a), dimg(&b), cx, cy, clamped, num_samples, fr);
        }
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

static int motion_linear_impl(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a, uint32_t distance, zg_stream stream) {
    (void)angle;
    return motion_linear_frames_impl(src, dst, 1, 0, 0, cos_a, sin_a, distance, stream);
}
static int motion_radial_impl(const zg_image *src, const zg_image *dst, float center_x, float center_y, float strength, int spin, zg_stream stream) {
    return motion_radial_frames_impl(src, dst, 1, 0, 0, center_x, center_y, strength, spin, stream);
}

// the pipeline's motion-blur step over a batch (batch.hip)
int motion_linear_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, float cos_a, float sin_a, uint32_t distance, hipStream_t s) {
    return motion_linear_frames_impl(src, dst, n, src_frame, dst_frame, cos_a, sin_a, distance, (zg_stream)s);
}
int motion_radial_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, float center_x, float center_y, float strength, int spin,
                         hipStream_t s) {
    return motion_radial_frames_impl(src, dst, n, src_frame, dst_frame, center_x, center_y, strength, spin, (zg_stream)s);
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
