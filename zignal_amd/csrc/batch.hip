// batch.hip — the `pipeline` recipe [blur gaussian sigma, resize] over a batch of frames (semantics of reference
// src/cli/pipeline.zig:153-179 applied to N independent images laid out back to back). Frames are independent units:
// this is also the shard a rank processes in the multi-GPU bench (no halo, no collective on the data path).
//
// Result == Image.gaussianBlur followed by Image.resize on every frame (tests compare against exactly that).
//   * Rgba(u8), bilinear to exactly half size, Gaussian-sized taps: ONE launch for the whole batch, blur and the 2:1
//     bilinear fused (conv_sep_rgba8.hip); per frame the HBM traffic is the source read once + the small output.
//   * Rgba(u8) otherwise: one batched blur launch into scratch frames, then resize per frame.
//   * anything else: gaussianBlur + resize per frame.
#include "zg_common.h"

#include <cmath>
#include <vector>

namespace zg {
int resize_impl(const zg_image *src, const zg_image *dst, const zg_method *method, hipStream_t s);
struct Rgba8Batch {
    const void *src; void *dst;
    uint32_t n_frames, rows, cols;
    size_t src_stride, dst_stride, src_frame_px, dst_frame_px;
    bool down2;
};
int try_sep_rgba8_batch(const Rgba8Batch &b, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s);
}

using namespace zg;

extern "C" {

int zg_batch_blur_resize(const void *src_frames, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, float sigma,
                         void *dst_frames, uint32_t out_rows, uint32_t out_cols, const zg_method *method, zg_stream stream) {
    ZG_REQUIRE(pixel_valid(pixel), ZG_ERR_INVALID_ARGUMENT, "batch: invalid pixel type %d", pixel);
    ZG_REQUIRE(method != nullptr, ZG_ERR_INVALID_ARGUMENT, "batch: null method");
    ZG_REQUIRE(sigma >= 0, ZG_ERR_INVALID_ARGUMENT, "batch: InvalidSigma (%g)", sigma);
    if (n_frames == 0 || rows == 0 || cols == 0 || out_rows == 0 || out_cols == 0) return ZG_OK;
    ZG_REQUIRE(src_frames && dst_frames, ZG_ERR_INVALID_ARGUMENT, "batch: null frame pointer");
    hipStream_t s = as_stream(stream);
    const size_t ps = pixel_size(pixel);
    const size_t in_px = (size_t)rows * cols, out_px = (size_t)out_rows * out_cols;

    // integer taps exactly as convolveSeparable derives them (convolution.zig:303-309) from gaussianBlur's f32 taps
    std::vector<int32_t> taps;
    if (sigma > 0 && pixel == ZG_PIXEL_RGBA_U8) {
        float f[255];
        int n = zg_gaussian_kernel(sigma, nullptr, 0);
        if (n < 0) return -n;
        if (n <= 255) n = zg_gaussian_kernel(sigma, f, 255); // longer kernels take the general per-frame path below
        if (n < 0) return -n;
        if (n <= 255) taps.resize((size_t)n);
        for (size_t i = 0; i < taps.size(); ++i) taps[i] = (int32_t)std::round(f[i] * 256.0f);
        const bool half = method->kind == ZG_INTERP_BILINEAR && rows == 2 * out_rows && cols == 2 * out_cols;
        if (half && !taps.empty()) {
            const StreamJob job{src_frames, dst_frames, n_frames, rows, cols, 4, (size_t)cols * 4, (size_t)out_cols * 4, in_px * 4, out_px * 4, true};
            const int rcs = try_sep_stream(job, taps.data(), taps.data(), n, ZG_BORDER_MIRROR, s);
            if (rcs >= 0) return rcs;
            const Rgba8Batch b{src_frames, dst_frames, n_frames, rows, cols, cols, out_cols, in_px, out_px, true};
            const int rc = try_sep_rgba8_batch(b, taps.data(), taps.data(), n, ZG_BORDER_MIRROR, s);
            if (rc >= 0) return rc;
        }
    }

    void *scratch = nullptr;
    int rc = ZG_OK;
    if (!taps.empty()) { // batched blur of every frame in one launch, then the reference's resize per frame
        if ((rc = scratch_alloc(&scratch, (size_t)n_frames * in_px * ps, s))) return rc;
        const StreamJob job{src_frames, scratch, n_frames, rows, cols, 4, (size_t)cols * 4, (size_t)cols * 4, in_px * 4, in_px * 4, false};
        rc = try_sep_stream(job, taps.data(), taps.data(), (int)taps.size(), ZG_BORDER_MIRROR, s);
        if (rc < 0) {
            const Rgba8Batch b{src_frames, scratch, n_frames, rows, cols, cols, cols, in_px, in_px, false};
            rc = try_sep_rgba8_batch(b, taps.data(), taps.data(), (int)taps.size(), ZG_BORDER_MIRROR, s);
        }
        if (rc >= 0) {
            for (uint32_t i = 0; i < n_frames && rc == ZG_OK; ++i) {
                zg_image tmp{(char *)scratch + (size_t)i * in_px * ps, cols, rows, cols, pixel};
                zg_image dst{(char *)dst_frames + (size_t)i * out_px * ps, out_cols, out_rows, out_cols, pixel};
                rc = resize_impl(&tmp, &dst, method, s);
            }
            scratch_free(scratch, s);
            return rc;
        }
        scratch_free(scratch, s);
        scratch = nullptr;
    }

    // general path: two scratch frames so frame i+1's blur may start while frame i's resize still reads its scratch
    if ((rc = scratch_alloc(&scratch, 2 * in_px * ps, s))) return rc;
    for (uint32_t i = 0; i < n_frames && rc == ZG_OK; ++i) {
        zg_image src{(char *)src_frames + (size_t)i * in_px * ps, cols, rows, cols, pixel};
        zg_image tmp{(char *)scratch + (size_t)(i & 1) * in_px * ps, cols, rows, cols, pixel};
        zg_image dst{(char *)dst_frames + (size_t)i * out_px * ps, out_cols, out_rows, out_cols, pixel};
        rc = zg_gaussian_blur(&src, &tmp, sigma, stream);
        if (rc == ZG_OK) rc = resize_impl(&tmp, &dst, method, s);
    }
    scratch_free(scratch, s);
    return rc;
}

} // extern "C"
