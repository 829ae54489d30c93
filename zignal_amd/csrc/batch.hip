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

#include <algorithm>
#include <cmath>
#include <functional>
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

namespace zg {
int convert_impl(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const float *srgb_lut, hipStream_t s);

namespace {

struct Frames { // n equally shaped frames back to back
    void *data;
    uint32_t n, rows, cols;
    int pixel, space;
    size_t frame_bytes() const { return (size_t)rows * cols * pixel_size(pixel); }
    zg_image frame(uint32_t i) const { return zg_image{(char *)data + (size_t)i * frame_bytes(), cols, rows, cols, pixel}; }
};

// the integer taps convolveSeparable derives from gaussianBlur's f32 taps (convolution.zig:303-309), when the kernel is short
int gaussian_taps_u8(float sigma, std::vector<int32_t> &taps) {
    taps.clear();
    float f[255];
    int n = zg_gaussian_kernel(sigma, nullptr, 0);
    if (n < 0) return -n;
    if (n > 255) return ZG_OK; // long kernels: the per-frame path
    n = zg_gaussian_kernel(sigma, f, 255);
    if (n < 0) return -n;
    taps.resize((size_t)n);
    for (int i = 0; i < n; ++i) taps[(size_t)i] = (int32_t)std::round(f[i] * 256.0f);
    return ZG_OK;
}

int apply_shape(const zg_step &st, Frames &f) { // what a step does to shape and type; validates the step
    switch (st.kind) {
    case ZG_STEP_GAUSSIAN_BLUR:
        ZG_REQUIRE(st.sigma >= 0, ZG_ERR_INVALID_ARGUMENT, "pipeline: InvalidSigma (%g)", st.sigma);
        return ZG_OK;
    case ZG_STEP_BOX_BLUR: return ZG_OK;
    case ZG_STEP_MEDIAN_BLUR: // the CLI's own limit (src/cli/blur.zig:118-121); the kernel's is lower and reported when the step runs
        ZG_REQUIRE(st.radius <= 256, ZG_ERR_INVALID_ARGUMENT, "pipeline: median blur radius %u exceeds 256", st.radius);
        return ZG_OK;
    case ZG_STEP_MOTION_BLUR:
        ZG_REQUIRE(st.motion >= ZG_MOTION_LINEAR && st.motion <= ZG_MOTION_RADIAL_SPIN, ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid motion blur kind %d", st.motion);
        return ZG_OK;
    case ZG_STEP_EDGES:
        ZG_REQUIRE(st.edges >= ZG_EDGES_SOBEL && st.edges <= ZG_EDGES_SHEN_CASTAN, ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid edge detector %d", st.edges);
        return ZG_OK;
    case ZG_STEP_RESIZE:
    case ZG_STEP_WARP:
        ZG_REQUIRE(st.method.kind >= ZG_INTERP_NEAREST && st.method.kind <= ZG_INTERP_LANCZOS, ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid interpolation method %d", st.method.kind);
        if (st.kind == ZG_STEP_WARP) ZG_REQUIRE(st.transform >= ZG_TRANSFORM_SIMILARITY && st.transform <= ZG_TRANSFORM_PROJECTIVE, ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid transform %d", st.transform);
        f.rows = st.out_rows;
        f.cols = st.out_cols;
        return ZG_OK;
    case ZG_STEP_CONVERT:
        ZG_REQUIRE(pixel_valid(st.dst_pixel), ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid pixel type %d", st.dst_pixel);
        ZG_REQUIRE(st.dst_space >= ZG_CS_GRAY && st.dst_space <= ZG_CS_XYB, ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid colour space %d", st.dst_space);
        f.pixel = st.dst_pixel;
        f.space = st.dst_space;
        return ZG_OK;
    }
    set_error("pipeline: unknown step kind %d", st.kind);
    return ZG_ERR_INVALID_ARGUMENT;
}

int per_frame(const Frames &in, const Frames &out, const std::function<int(const zg_image *, const zg_image *)> &op) {
    for (uint32_t i = 0; i < in.n; ++i) {
        const zg_image a = in.frame(i), b = out.frame(i);
        if (const int rc = op(&a, &b)) return rc;
    }
    return ZG_OK;
}

int run_blur(const Frames &in, const Frames &out, float sigma, hipStream_t s) {
    if (sigma == 0) { ZG_HIP(hipMemcpyAsync(out.data, in.data, in.n * in.frame_bytes(), hipMemcpyDeviceToDevice, s)); return ZG_OK; } // image.zig:966
    if (!pixel_is_float(in.pixel)) {
        std::vector<int32_t> taps;
        if (const int rc = gaussian_taps_u8(sigma, taps)) return rc;
        if (!taps.empty()) {
            const size_t sp = pixel_size(in.pixel);
            const StreamJob job{in.data, out.data, in.n, in.rows, in.cols, (int)sp, in.cols * sp, in.cols * sp, in.frame_bytes(), in.frame_bytes(), false};
            const int rc = try_sep_stream(job, taps.data(), taps.data(), (int)taps.size(), ZG_BORDER_MIRROR, s);
            if (rc >= 0) return rc;
            if (in.n > 1) { // longer kernels: the two passes through the u16 temp planes, the frame in the grid
                const zg_image a = in.frame(0), b = out.frame(0);
                const int rc2 = try_sep_bytes2_frames(&a, &b, in.n, in.frame_bytes(), in.frame_bytes(), taps.data(), (int)taps.size(), taps.data(),
                                                      (int)taps.size(), ZG_BORDER_MIRROR, s);
                if (rc2 >= 0) return rc2;
            }
        }
    }
    return per_frame(in, out, [&](const zg_image *a, const zg_image *b) { return zg_gaussian_blur(a, b, sigma, (zg_stream)s); });
}

int run_resize(const Frames &in, const Frames &out, const zg_method &method, hipStream_t s) {
    if (in.n > 0) { // one launch over the batch wherever the path is a single kernel
        const zg_image a = in.frame(0), b = out.frame(0);
        const int rc = resize_frames(&a, &b, &method, in.n, in.frame_bytes(), out.frame_bytes(), s);
        if (rc >= 0) return rc;
    }
    return per_frame(in, out, [&](const zg_image *a, const zg_image *b) { return resize_impl(a, b, &method, s); });
}

int run_convert(const Frames &in, const Frames &out, const float *lut, hipStream_t s) {
    // per pixel, and the frames are contiguous: the batch is one tall image
    const uint64_t tall = (uint64_t)in.rows * in.n;
    if (tall <= 0x7fffffffu) {
        const zg_image a{in.data, in.cols, (uint32_t)tall, in.cols, in.pixel}, b{out.data, out.cols, (uint32_t)tall, out.cols, out.pixel};
        return convert_impl(&a, in.space, &b, out.space, lut, s);
    }
    return per_frame(in, out, [&](const zg_image *a, const zg_image *b) { return convert_impl(a, in.space, b, out.space, lut, s); });
}

// The CLI's edges step (src/cli/edges.zig:126-135): gray = frame.convert(u8); detector(gray) -> Image(u8); .convert(frame type). The
// detectors take any pixel type and start with that very conversion per pixel (edges.zig:36-45: as(f32, convertColor(u8, px))), so they
// read the frames directly; the edge maps of the whole batch go to one grey scratch plane and come back through ONE conversion launch.
// That shortcut holds for u8 frames in Gray / Rgb / Rgba. Float frames do not: Image(f32).sobel works on the raw floats (edges.zig:36-45 casts, it does
// not scale), where the bridge's frame.convert(u8) maps [0, 1] to [0, 255] first; and frames a CONVERT step left in another colour space (Oklab, Lab, Hsv ...)
// are not RGB at all. Those frames go through the bridge literally: one conversion launch to a grey u8 plane, the detector on that.
int run_edges(const zg_step &st, const Frames &in_frames, const Frames &out, hipStream_t s) {
    void *grey = nullptr, *bridged = nullptr;
    const size_t plane = (size_t)in_frames.rows * in_frames.cols;
    if (const int rc = scratch_alloc(&grey, plane * in_frames.n, s)) return rc;
    Frames in = in_frames;
    const bool direct = !pixel_is_float(in.pixel) && (in.space == ZG_CS_GRAY || in.space == ZG_CS_RGB || in.space == ZG_CS_RGBA);
    if (!direct) {
        if (const int rc = scratch_alloc(&bridged, plane * in.n, s)) { scratch_free(grey, s); return rc; }
        const Frames b8{bridged, in.n, in.rows, in.cols, ZG_PIXEL_U8, ZG_CS_GRAY};
        if (const int rc = run_convert(in, b8, nullptr, s)) { scratch_free(bridged, s); scratch_free(grey, s); return rc; }
        in = b8;
    }
    const Frames g{grey, in.n, in.rows, in.cols, ZG_PIXEL_U8, ZG_CS_GRAY};
    int rc = -1;
    if (st.edges == ZG_EDGES_SOBEL && in.n > 0) {
        const zg_image a = in.frame(0), b = g.frame(0);
        rc = sobel_frames(&a, &b, in.n, in.frame_bytes(), plane, s); // one launch over the batch
    }
    if (rc == -1)
        rc = per_frame(in, g, [&](const zg_image *a, const zg_image *b) {
            if (st.edges == ZG_EDGES_SOBEL) return zg_sobel(a, b, (zg_stream)s);
            if (st.edges == ZG_EDGES_CANNY) return zg_canny(a, b, st.sigma, st.low, st.high, (zg_stream)s);
            return zg_shen_castan(a, b, st.sigma, st.window, st.high, st.low, 1, st.use_nms, (zg_stream)s);
        });
    if (rc == ZG_OK) rc = run_convert(g, out, nullptr, s);
    scratch_free(bridged, s);
    scratch_free(grey, s);
    return rc;
}

// motion blur and box blur over the batch: the frame index is in the grid (round 5; they launched once per frame before)
int run_motion(const zg_step &st, const Frames &in, const Frames &out, hipStream_t s) {
    if (in.n == 0) return ZG_OK;
    const zg_image a = in.frame(0), b = out.frame(0);
    if (st.motion == ZG_MOTION_LINEAR) return motion_linear_frames(&a, &b, in.n, in.frame_bytes(), out.frame_bytes(), st.cos_a, st.sin_a, st.distance, s);
    return motion_radial_frames(&a, &b, in.n, in.frame_bytes(), out.frame_bytes(), st.center_x, st.center_y, st.strength, st.motion == ZG_MOTION_RADIAL_SPIN, s);
}
int run_box(const zg_step &st, const Frames &in, const Frames &out, hipStream_t s) {
    if (in.n == 0) return ZG_OK;
    const zg_image a = in.frame(0), b = out.frame(0);
    return box_blur_frames(&a, &b, in.n, in.frame_bytes(), out.frame_bytes(), st.radius, s);
}

// steps [i, i + 2) as one fused launch over the batch, or -1
int run_fused_pair(const zg_step &s0, const zg_step &s1, const Frames &in, const Frames &out, hipStream_t s) {
    if (s0.kind == ZG_STEP_GAUSSIAN_BLUR && s1.kind == ZG_STEP_RESIZE && in.pixel == ZG_PIXEL_RGBA_U8 && s0.sigma > 0 && s1.method.kind == ZG_INTERP_BILINEAR &&
        in.rows == 2 * out.rows && in.cols == 2 * out.cols) {
        std::vector<int32_t> taps;
        if (const int rc = gaussian_taps_u8(s0.sigma, taps)) return rc;
        if (taps.empty()) return -1;
        const StreamJob job{in.data, out.data, in.n, in.rows, in.cols, 4, (size_t)in.cols * 4, (size_t)out.cols * 4, in.frame_bytes(), out.frame_bytes(), true};
        return try_sep_stream(job, taps.data(), taps.data(), (int)taps.size(), ZG_BORDER_MIRROR, s);
    }
    if (s0.kind == ZG_STEP_RESIZE && s1.kind == ZG_STEP_CONVERT && in.space == ZG_CS_RGBA && s0.method.kind == ZG_INTERP_BILINEAR && in.n > 0) {
        const zg_image a = in.frame(0), b = out.frame(0);
        return resize_convert_rgba8_frames(&a, &b, s1.dst_space, in.n, in.frame_bytes(), out.frame_bytes(), s1.srgb_lut, s);
    }
    return -1;
}

} // namespace
} // namespace zg

using namespace zg;

extern "C" {

size_t zg_sizeof_step(void) { return sizeof(zg_step); }

int zg_batch_pipeline_shape(uint32_t rows, uint32_t cols, int pixel, int space, const zg_step *steps, uint32_t n_steps, uint32_t *out_rows, uint32_t *out_cols,
                            int *out_pixel, int *out_space) {
    ZG_REQUIRE(pixel_valid(pixel), ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid pixel type %d", pixel);
    ZG_REQUIRE(n_steps == 0 || steps != nullptr, ZG_ERR_INVALID_ARGUMENT, "pipeline: null steps");
    Frames f{nullptr, 0, rows, cols, pixel, space};
    for (uint32_t i = 0; i < n_steps; ++i)
        if (const int rc = apply_shape(steps[i], f)) return rc;
    if (out_rows) *out_rows = f.rows;
    if (out_cols) *out_cols = f.cols;
    if (out_pixel) *out_pixel = f.pixel;
    if (out_space) *out_space = f.space;
    return ZG_OK;
}

int zg_batch_pipeline(const void *src_frames, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, int space, const zg_step *steps, uint32_t n_steps,
                      void *dst_frames, zg_stream stream) {
    ZG_REQUIRE(pixel_valid(pixel), ZG_ERR_INVALID_ARGUMENT, "pipeline: invalid pixel type %d", pixel);
    ZG_REQUIRE(n_steps == 0 || steps != nullptr, ZG_ERR_INVALID_ARGUMENT, "pipeline: null steps");
    hipStream_t s = as_stream(stream);
    // shapes after every step (validates all steps before anything is enqueued, like the CLI validates its recipe first: pipeline.zig:108-114)
    std::vector<Frames> shape(n_steps + 1);
    shape[0] = Frames{nullptr, 0, rows, cols, pixel, space};
    size_t widest = 0; // bytes per frame of the largest intermediate
    for (uint32_t i = 0; i < n_steps; ++i) {
        shape[i + 1] = shape[i];
        if (const int rc = apply_shape(steps[i], shape[i + 1])) return rc;
        if (i + 1 < n_steps) widest = std::max(widest, shape[i + 1].frame_bytes());
    }
    if (n_frames == 0 || rows == 0 || cols == 0) return ZG_OK;
    ZG_REQUIRE(src_frames && dst_frames, ZG_ERR_INVALID_ARGUMENT, "pipeline: null frame pointer");
    for (const Frames &f : shape) ZG_REQUIRE(f.rows > 0 && f.cols > 0, ZG_ERR_INVALID_ARGUMENT, "pipeline: a step produces empty frames");
    if (n_steps == 0) {
        ZG_HIP(hipMemcpyAsync(dst_frames, src_frames, (size_t)n_frames * shape[0].frame_bytes(), hipMemcpyDeviceToDevice, s));
        return ZG_OK;
    }
    // Frames go through in groups: two ping-pong scratch blocks hold a group's intermediates (a 1024-frame 1080p batch has 8.5 GB of
    // them; nothing is gained by keeping more than a chip-filling group in flight). A block takes at most a quarter of the scratch
    // cache (512 MiB by default), so both stay cached from call to call beside whatever else lives there.
    const size_t budget = std::min<size_t>((size_t)1 << 30, scratch_block_budget());
    const uint32_t group = widest ? (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, budget / widest)) : n_frames;
    void *ping[2] = {nullptr, nullptr};
    int rc = ZG_OK;
    if (widest) {
        if ((rc = scratch_alloc(&ping[0], (size_t)group * widest, s))) return rc;
        if (n_steps > 2 && (rc = scratch_alloc(&ping[1], (size_t)group * widest, s))) { scratch_free(ping[0], s); return rc; }
    }
    for (uint32_t g0 = 0; g0 < n_frames && rc == ZG_OK; g0 += group) {
        const uint32_t gn = std::min(group, n_frames - g0);
        Frames cur = shape[0];
        cur.n = gn;
        cur.data = (char *)src_frames + (size_t)g0 * shape[0].frame_bytes();
        int flip = 0;
        for (uint32_t i = 0; i < n_steps && rc == ZG_OK;) {
            // try the step together with its successor
            uint32_t take = 1;
            if (i + 1 < n_steps) {
                Frames two = shape[i + 2];
                two.n = gn;
                two.data = i + 2 == n_steps ? (void *)((char *)dst_frames + (size_t)g0 * shape[n_steps].frame_bytes()) : ping[flip];
                const int rf = run_fused_pair(steps[i], steps[i + 1], cur, two, s);
                if (rf >= 0) {
                    rc = rf;
                    take = 2;
                    cur = two;
                    if (i + 2 != n_steps) flip ^= 1;
                }
            }
            if (take == 1) {
                Frames next = shape[i + 1];
                next.n = gn;
                next.data = i + 1 == n_steps ? (void *)((char *)dst_frames + (size_t)g0 * shape[n_steps].frame_bytes()) : ping[flip];
                const zg_step &st = steps[i];
                switch (st.kind) {
                case ZG_STEP_GAUSSIAN_BLUR: rc = run_blur(cur, next, st.sigma, s); break;
                case ZG_STEP_BOX_BLUR: rc = run_box(st, cur, next, as_stream(stream)); break;
                case ZG_STEP_RESIZE: rc = run_resize(cur, next, st.method, s); break;
                case ZG_STEP_CONVERT: rc = run_convert(cur, next, st.srgb_lut, s); break;
                case ZG_STEP_MEDIAN_BLUR:
                    rc = per_frame(cur, next, [&](const zg_image *a, const zg_image *b) { return zg_order_statistic_blur(a, b, st.radius, 0, 0.5, ZG_BORDER_MIRROR, stream); });
                    break;
                case ZG_STEP_MOTION_BLUR: rc = run_motion(st, cur, next, s); break;
                case ZG_STEP_EDGES: rc = run_edges(st, cur, next, s); break;
                default: { // warp: the same map for every frame, one launch
                    const zg_image a = cur.frame(0), b = next.frame(0);
                    rc = warp_frames(&a, &b, st.transform, st.m, &st.method, gn, cur.frame_bytes(), next.frame_bytes(), s);
                    if (rc == -1) // more frames than one launch takes: frame by frame
                        rc = per_frame(cur, next, [&](const zg_image *fa, const zg_image *fb) { return zg_warp(fa, fb, st.transform, st.m, &st.method, stream); });
                    break;
                }
                }
                cur = next;
                if (i + 1 != n_steps) flip ^= 1;
            }
            i += take;
        }
    }
    scratch_free(ping[0], s);
    scratch_free(ping[1], s);
    return rc;
}

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
