// zg_common.h — shared host/device helpers of libzignal_hip (gfx950 only).
//
// Everything numeric in this library is compiled with -ffp-contract=off: the reference computes
// f32 with separate multiply and add (no @mulAdd anywhere under src/image*), and bit parity with
// it depends on never forming an FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <functional>
#include <string>
#include <type_traits>

#include "../../include/zignal_hip.h"

#pragma clang fp contract(off)

namespace zg {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define ZG_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return ::zg::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define ZG_REQUIRE(cond, status, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            ::zg::set_error(__VA_ARGS__);  \
            return (status);               \
        }                                  \
    } while (0)

inline hipStream_t as_stream(zg_stream s) { return reinterpret_cast<hipStream_t>(s); }

// ---- pixel layouts -------------------------------------------------------------------------
__host__ __device__ inline int pixel_channels(int pixel) {
    switch (pixel) {
    case ZG_PIXEL_U8: case ZG_PIXEL_F32: return 1;
    case ZG_PIXEL_RGB_U8: case ZG_PIXEL_RGB_F32: return 3;
    default: return 4;
    }
}
__host__ __device__ inline bool pixel_is_float(int pixel) {
    return pixel == ZG_PIXEL_F32 || pixel == ZG_PIXEL_RGB_F32 || pixel == ZG_PIXEL_RGBA_F32;
}
__host__ __device__ inline size_t pixel_size(int pixel) {
    return (size_t)pixel_channels(pixel) * (pixel_is_float(pixel) ? 4 : 1);
}
inline bool pixel_valid(int pixel) { return pixel >= ZG_PIXEL_U8 && pixel <= ZG_PIXEL_RGBA_F32; }

// Device-side image descriptor (Image(T) fields, reference src/image.zig:97-103).
struct DImg {
    void *data;
    uint64_t stride; // pixels
    int32_t rows, cols;
};
inline DImg dimg(const zg_image *im) {
    return DImg{im->data, (uint64_t)im->stride, (int32_t)im->rows, (int32_t)im->cols};
}

// Compile-time pixel traits. `Vec` is the register / LDS form of one pixel (a clang ext vector, so
// it lives in VGPRs and moves with one instruction); 3-channel pixels pad to 4 lanes in registers
// and LDS but are 3 tightly packed elements in memory (reference src/color.zig:286-290).
template <typename E, int N> struct VecOf { typedef E type __attribute__((ext_vector_type(N))); };

template <typename E, int CH> struct PxBase {
    using Elem = E;
    static constexpr int C = CH;
    static constexpr int BYTES = CH * (int)sizeof(E);
    using Vec = typename VecOf<E, CH>::type;
    __device__ static Vec zero() {
        Vec v;
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = (E)0;
        return v;
    }
    __device__ static Vec load(const void *base, size_t idx) {
        if constexpr (CH == 3) {
            const E *p = (const E *)base + idx * 3;
            Vec v;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
            return v;
        } else {
            return ((const Vec *)base)[idx];
        }
    }
    __device__ static void store(void *base, size_t idx, Vec v) {
        if constexpr (CH == 3) {
            E *p = (E *)base + idx * 3;
            p[0] = v[0]; p[1] = v[1]; p[2] = v[2];
        } else {
            ((Vec *)base)[idx] = v;
        }
    }
};
template <int PIX> struct Px;
template <> struct Px<ZG_PIXEL_U8> : PxBase<uint8_t, 1> {};
template <> struct Px<ZG_PIXEL_F32> : PxBase<float, 1> {};
template <> struct Px<ZG_PIXEL_RGB_U8> : PxBase<uint8_t, 3> {};
template <> struct Px<ZG_PIXEL_RGBA_U8> : PxBase<uint8_t, 4> {};
template <> struct Px<ZG_PIXEL_RGB_F32> : PxBase<float, 3> {};
template <> struct Px<ZG_PIXEL_RGBA_F32> : PxBase<float, 4> {};

// Dispatch a runtime zg_pixel to a compile-time tag: f(std::integral_constant<int, PIX>{}).
template <typename F> inline int dispatch_pixel(int pixel, F &&f) {
    switch (pixel) {
    case ZG_PIXEL_U8: return f(std::integral_constant<int, ZG_PIXEL_U8>{});
    case ZG_PIXEL_F32: return f(std::integral_constant<int, ZG_PIXEL_F32>{});
    case ZG_PIXEL_RGB_U8: return f(std::integral_constant<int, ZG_PIXEL_RGB_U8>{});
    case ZG_PIXEL_RGBA_U8: return f(std::integral_constant<int, ZG_PIXEL_RGBA_U8>{});
    case ZG_PIXEL_RGB_F32: return f(std::integral_constant<int, ZG_PIXEL_RGB_F32>{});
    case ZG_PIXEL_RGBA_F32: return f(std::integral_constant<int, ZG_PIXEL_RGBA_F32>{});
    }
    set_error("invalid pixel type %d", pixel);
    return ZG_ERR_INVALID_ARGUMENT;
}

// ---- border (reference src/image/border.zig:46-63) ------------------------------------------
// Returns the in-range index, or -1 for the reference's `null` (sample is zero).
__host__ __device__ inline int resolve_index(int idx, int length, int border) {
    if (idx >= 0 && idx < length) return idx;
    switch (border) {
    case ZG_BORDER_ZERO: return -1;
    case ZG_BORDER_REPLICATE:
        if (length == 0) return -1;
        return idx < 0 ? 0 : length - 1;
    case ZG_BORDER_MIRROR: {
        if (length <= 0) return -1;
        if (length == 1) return 0;
        const int period = 2 * (length - 1);
        int m = idx % period;
        if (m < 0) m += period; // @mod is floored
        return m >= length ? period - m : m;
    }
    default: { // wrap
        if (length == 0) return -1;
        int m = idx % length;
        if (m < 0) m += length;
        return m;
    }
    }
}

// ---- meta.clamp (reference src/meta.zig:110-135) ---------------------------------------------
// float -> u8: trunc(clamp(round(f64(v)), 0, 255)); rounding an f32 in f64 equals roundf in f32.
// Evaluated as clamp-then-round, which is the same function: u = min(max(v, 0), 255) (NaN -> 255 like the reference's
// @min), then trunc(u) + (frac >= 0.5) — for u >= 0 that is round-half-away, and u <= 255 keeps the result in range.
__device__ inline uint8_t clamp_u8_f32(float v) {
    float u = fmaxf(v, 0.0f);            // negative values round to <= 0 and clamp to 0 anyway; fmaxf(NaN, 0) = 0 ...
    u = (v != v) ? 255.0f : fminf(u, 255.0f); // ... but the reference maps NaN to 255 (std.math.clamp via @min / @max)
    const float t = truncf(u);
    const int r = (int)t + ((u - t) >= 0.5f ? 1 : 0);
    return (uint8_t)r;
}
__device__ inline uint8_t clamp_u8_i32(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Kernels whose neighbouring workgroups share source lines renumber their workgroups XCD-major (block b runs on XCD b % 8, each XCD has
// its own L2). tools/build_variant.sh-style builds with -DZG_XCD_ORDER=0 run every kernel in plain address order instead: the A/B of round 5.
#ifndef ZG_XCD_ORDER
#define ZG_XCD_ORDER 1
#endif

// ---- launch helpers ------------------------------------------------------------------------
inline unsigned ceil_div(unsigned a, unsigned b) { return (a + b - 1) / b; }
// One workgroup row per image row: HIP caps gridDim.y at 65535, so rows past that continue in gridDim.z. Kernels read their
// row with grid_row() and return when it is past the image (the last z slice is usually partial). The reference has no
// such limit (rows: u32, src/image.zig:97-103).
constexpr unsigned GRID_Y_MAX = 65535u;
inline dim3 row_grid(unsigned gx, unsigned rows) { return rows <= GRID_Y_MAX ? dim3(gx, rows, 1) : dim3(gx, GRID_Y_MAX, ceil_div(rows, GRID_Y_MAX)); }
__device__ inline int grid_row() { return (int)(blockIdx.z * GRID_Y_MAX + blockIdx.y); }

// Host-layer scaffolding (zg_runtime.cpp): stage host images to the device, run, copy back.
struct HostStage {
    zg_image dev{};       // device twin (contiguous: stride == cols)
    const zg_image *host{};
    bool writeback = false;
    ~HostStage();
    int upload(const zg_image *h, bool copy_in, bool write_back);
    int finish();         // D2H if writeback
};

int check_image(const zg_image *im, const char *name, bool device_pointer = true);

// Host images through a row-local device op as a banded, full-duplex pipeline (zg_runtime.cpp): `op` is called once per
// band with device views (source rows include up to `halo` real neighbour rows on each side) and the stream to launch on.
// Returns -1 when the call does not qualify (small, overlapping, too few rows): the caller then takes the whole-frame path.
typedef std::function<int(const zg_image *src_view, const zg_image *dst_view, hipStream_t s)> BandOp;
int host_banded(const zg_image *src, const zg_image *dst, uint32_t halo, const BandOp &op);

// Pageable host memory -> device memory on stream s, synchronised before returning (zg_runtime.cpp).
int upload_pageable(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s);
// rows of `width` bytes, `spitch` apart on the host, packed back to back on the device
int upload_pageable_rows(void *dst_dev, const void *src_host, size_t spitch, size_t width, size_t rows, hipStream_t s);
// the reverse trips: device memory (contiguous) into pageable host memory / into host rows `dpitch` apart
int download_pageable(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s);
int download_pageable_rows(void *dst_host, size_t dpitch, const void *src_dev, size_t width, size_t rows, hipStream_t s);

// Frames of a batch inside one launch: bytes from one frame to the next on both sides. The frame index is blockIdx.y (grids are
// tiles-per-frame x frames), so a one-image launch pays nothing for it: round 3 carried the index in blockIdx.x and every workgroup
// divided by the tiles per frame — forty scalar instructions behind an extra kernel-argument load, 12 % of the bicubic warp and
// 15 % of the 2:1 resize (profiles/r04_experiments.txt).
struct FrameSpan {
    size_t src_frame, dst_frame;
};
constexpr uint32_t MAX_FRAMES_PER_LAUNCH = 65535; // gridDim.y
int resize_bilinear_rgba8_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s);              // resize_planes.hip
int resize_planes_frames(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s); // resize_planes.hip
int resize_frames(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s);       // geom.hip
int warp_frames(const zg_image *src, const zg_image *dst, int kind, const float *mat, const zg_method *method, uint32_t n, size_t src_frame, size_t dst_frame,
                hipStream_t s);                                                                                                                        // geom.hip
int try_pyramid_levels_u8(const zg_image *src, const zg_image *levels, const float *sigmas, uint32_t n, uint8_t *handled, int which, hipStream_t s); // conv_sep_bytes2.hip: several levels in three launches
int try_pyramid_tiles_u8(const zg_image *src, const zg_image *levels, const float *sigmas, uint32_t n, uint8_t *handled, hipStream_t s); // pyramid_tile.hip: a level per kernel, nothing but the level written
int try_pyramid_level_u8(const zg_image *src, const zg_image *level, const int32_t *taps, int nk, hipStream_t s); // conv_sep_bytes2.hip: blur + bilinear level, -1 = not this shape
int box_blur_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, uint32_t radius, hipStream_t s);                   // box_blur.hip
int motion_linear_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, float cos_a, float sin_a, uint32_t distance,
                         hipStream_t s);                                                                                                                 // motion.hip
int motion_radial_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, float center_x, float center_y, float strength,
                         int spin, hipStream_t s);                                                                                                       // motion.hip
int sobel_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s);                                        // edges.hip
int resize_convert_rgba8_frames(const zg_image *src, const zg_image *dst, int dst_space, uint32_t n, size_t src_frame, size_t dst_frame, const float *srgb_lut,
                                hipStream_t s);                                                                                                       // convert.hip

// u8 separable convolution of a batch of equally sized frames laid out back to back, one wave per column strip
// (conv_sep_stream.hip). Returns -1 when its preconditions do not hold: the caller falls back to the tiled kernels.
struct StreamJob {
    const void *src; void *dst;
    uint32_t n_frames, rows, cols;
    int sp;                                // bytes per pixel: 1, 3, 4
    size_t src_pitch, dst_pitch;           // bytes between rows
    size_t src_frame, dst_frame;           // bytes between frames
    bool down2;                            // dst is (rows / 2) x (cols / 2): blur then 2:1 bilinear (sp == 4)
};
int try_sep_stream(const StreamJob &j, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s);

// scratch blocks from the library's caching allocator, ordered on stream s (zg_runtime.cpp)
int scratch_alloc(void **out, size_t bytes, hipStream_t s);
int host_threads(); // ZIGNAL_HIP_HOST_THREADS, else min(16, hardware threads)
void scratch_free(void *p, hipStream_t s);
int try_sep_bytes2_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, const int32_t *ix, int nkx,
                          const int32_t *iy, int nky, int border, hipStream_t s); // conv_sep_bytes2.hip
size_t scratch_block_budget(); // bytes one long-lived scratch block may take so that a few of them stay cached (a quarter of the cache limit)

} // namespace zg
