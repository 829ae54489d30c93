/*
 * zignal_hip.h — C ABI of libzignal_hip.so, the MI355X (gfx950) implementation of
 * zignal's per-pixel image hot path (src/image: convolution, resize / rotate / warp
 * sampling, colour conversion).
 *
 * The reference has no FFI seam for this path: `Image(T)` methods forward at comptime
 * into module functions (reference src/image.zig:523-525, :621-623, :917-994, :396-407).
 * The replacement seam is therefore those method bodies; each entry point below names
 * the reference function whose body it replaces. `Image(T)` is not an extern struct
 * (reference src/image.zig:97-103: {rows:u32, cols:u32, data:[]T, stride:usize}), so
 * its fields travel in `zg_image`.
 *
 * Two layers, same semantics:
 *   zg_<op>(..., zg_stream)   device pointers, asynchronous on `stream`
 *   zg_<op>_host(...)         host pointers, synchronous (H2D -> kernel -> D2H)
 *
 * Return value: zg_status. The Zig shim maps 1 -> error.DimensionMismatch,
 * 2 -> error.InvalidArgument (InvalidSigma / InvalidScaleFactor / InvalidDimensions
 * as documented per op), 3 -> error.OutOfMemory; 4 is a HIP runtime failure.
 *
 * Enum ordinals follow the declaration order of the reference enums:
 *   BorderMode     reference src/image/border.zig:10-18
 *   Interpolation  reference src/image/interpolation.zig:53-68
 */
#ifndef ZIGNAL_HIP_H
#define ZIGNAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZG_API __attribute__((visibility("default")))

typedef enum zg_status {
    ZG_OK = 0,
    ZG_ERR_DIMENSION_MISMATCH = 1, /* error.DimensionMismatch (src/image.zig:636,927,947,962) */
    ZG_ERR_INVALID_ARGUMENT = 2,   /* error.InvalidSigma (:970), InvalidScaleFactor / InvalidDimensions (:531-536) */
    ZG_ERR_OUT_OF_MEMORY = 3,      /* error.OutOfMemory (device scratch) */
    ZG_ERR_HIP = 4,                /* HIP runtime error; see zg_last_error() */
    ZG_ERR_UNSUPPORTED = 5,        /* pixel type / op combination the reference rejects at comptime */
    ZG_ERR_CODEC = 6               /* an error of the reference's codec error sets (src/codecs/png.zig); zg_last_error() STARTS WITH the
                                      Zig error name, e.g. "InvalidCrc", "NonConsecutiveIdatChunks", "ImageTooLarge" */
} zg_status;

/* Pixel layouts. Element index is row*stride + col, stride in PIXELS (src/image.zig:426-430). */
typedef enum zg_pixel {
    ZG_PIXEL_U8 = 0,       /* u8 / Gray(u8): 1 B                                  */
    ZG_PIXEL_F32 = 1,      /* f32: 4 B                                            */
    ZG_PIXEL_RGB_U8 = 2,   /* Rgb(u8): r,g,b bytes, 3 B (src/color.zig:286-290)   */
    ZG_PIXEL_RGBA_U8 = 3,  /* Rgba(u8): packed r,g,b,a, 4 B (src/color.zig:400)   */
    ZG_PIXEL_RGB_F32 = 4,  /* any 3 x f32 struct: Rgb(f32), Oklab(f32), Xyz(f32)  */
    ZG_PIXEL_RGBA_F32 = 5  /* Rgba(f32): 16 B                                     */
} zg_pixel;

typedef enum zg_border {   /* src/image/border.zig:10-18 */
    ZG_BORDER_ZERO = 0,
    ZG_BORDER_REPLICATE = 1,
    ZG_BORDER_MIRROR = 2,  /* reflect-101, period 2(L-1) (border.zig:56-59) */
    ZG_BORDER_WRAP = 3
} zg_border;

typedef enum zg_interp {   /* src/image/interpolation.zig:53-68 */
    ZG_INTERP_NEAREST = 0,
    ZG_INTERP_BILINEAR = 1,
    ZG_INTERP_BICUBIC = 2,
    ZG_INTERP_CATMULL_ROM = 3,
    ZG_INTERP_MITCHELL = 4, /* takes (b, c) */
    ZG_INTERP_LANCZOS = 5
} zg_interp;

typedef enum zg_transform_kind { /* src/geometry/transforms.zig:10,118,197 */
    ZG_TRANSFORM_SIMILARITY = 0, /* m = {a00,a01,a10,a11, b0,b1}            */
    ZG_TRANSFORM_AFFINE = 1,     /* m = {a00,a01,a10,a11, b0,b1}            */
    ZG_TRANSFORM_PROJECTIVE = 2  /* m = row-major 3x3                       */
} zg_transform_kind;

typedef enum zg_blending { /* src/blending.zig:8-22 */
    ZG_BLEND_NONE = 0, ZG_BLEND_NORMAL = 1, ZG_BLEND_MULTIPLY = 2, ZG_BLEND_SCREEN = 3, ZG_BLEND_OVERLAY = 4, ZG_BLEND_SOFT_LIGHT = 5,
    ZG_BLEND_HARD_LIGHT = 6, ZG_BLEND_COLOR_DODGE = 7, ZG_BLEND_COLOR_BURN = 8, ZG_BLEND_DARKEN = 9, ZG_BLEND_LIGHTEN = 10,
    ZG_BLEND_DIFFERENCE = 11, ZG_BLEND_EXCLUSION = 12
} zg_blending;

typedef enum zg_colorspace { /* src/color.zig ColorSpace (the ordinals are this library's, not Zig's) */
    ZG_CS_GRAY = 0,  /* Image(u8) / Image(f32) scalars */
    ZG_CS_RGB = 1,
    ZG_CS_RGBA = 2,
    ZG_CS_OKLAB = 3,
    ZG_CS_XYZ = 4,
    ZG_CS_YCBCR = 5, /* u8 (16.16 fixed point) and float forms */
    ZG_CS_HSL = 6,   /* 6..12: float-only colour types, three f32 fields in the struct's declaration order */
    ZG_CS_HSV = 7,
    ZG_CS_LAB = 8,
    ZG_CS_LCH = 9,
    ZG_CS_LMS = 10,
    ZG_CS_OKLCH = 11,
    ZG_CS_XYB = 12
} zg_colorspace;

/* Mirrors Image(T) (src/image.zig:97-103) plus the pixel tag that T carries at comptime. */
typedef struct zg_image {
    void *data;     /* first pixel (device pointer for zg_<op>, host pointer for zg_<op>_host) */
    size_t stride;  /* in pixels; stride >= cols; views keep the parent's stride (image.zig:332) */
    uint32_t rows;
    uint32_t cols;
    int32_t pixel;  /* zg_pixel */
} zg_image;

/* Interpolation method with Mitchell parameters (interpolation.zig:57-66). */
typedef struct zg_method {
    int32_t kind;   /* zg_interp */
    float b, c;     /* used when kind == ZG_INTERP_MITCHELL */
    /* Optional 1025-entry Lanczos3 table (interpolation.zig:256-267 builds it at comptime with
     * Zig's @sin). NULL -> the library's own table. Host pointer in both layers. */
    const float *lanczos_lut;
} zg_method;

typedef void *zg_stream; /* hipStream_t; NULL = default stream */
typedef void *zg_event;  /* hipEvent_t */
typedef void *zg_graph;  /* hipGraphExec_t: an instantiated, launchable graph */

/* ---- runtime ------------------------------------------------------------------------- */
/* The reference has no device: `Image(T)` owns host memory from a std.mem.Allocator (src/image.zig:124-134, deinit :173).
 * A device-resident image (zig/zignal_hip.zig DeviceImage(T), zignal_amd/cpp/zignal_hip.hpp DeviceImage<T>) gets its pixels
 * from zg_malloc, crosses PCIe once with zg_image_upload / zg_image_download, and calls the stream-taking entry points
 * below in between, so that a chain like pipeline.zig:153-179's [blur, resize] never leaves HBM. */
ZG_API int zg_init(int device);               /* hipSetDevice + the gfx950 check; == zg_set_device */
/* The current device belongs to the calling THREAD. A host driving several GPUs runs one thread per GPU, each calling
 * zg_set_device(i) once; that thread's allocations, scratch, cached tables and launches then live on GPU i. */
ZG_API int zg_set_device(int device);
ZG_API int zg_get_device(int *device);
ZG_API void zg_shutdown(void);
ZG_API const char *zg_last_error(void);       /* thread-local message of the last non-OK status */
ZG_API int zg_version(void);
ZG_API int zg_device_count(void);

ZG_API int zg_malloc(void **dev_ptr, size_t bytes);
ZG_API int zg_free(void *dev_ptr);
ZG_API int zg_malloc_host(void **host_ptr, size_t bytes); /* pinned host memory: copies to and from it are asynchronous */
ZG_API int zg_free_host(void *host_ptr);
/* Host pointer <-> device pointer; the call returns when the copy is complete (any host memory). */
ZG_API int zg_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, zg_stream stream);
ZG_API int zg_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, zg_stream stream);
/* The same, only enqueued on `stream` (meant for zg_malloc_host memory, which must stay valid until the stream gets there). */
ZG_API int zg_memcpy_h2d_async(void *dst_dev, const void *src_host, size_t bytes, zg_stream stream);
ZG_API int zg_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, zg_stream stream);
/* A whole image or view across PCIe, strides honoured on both sides (same rows, cols and pixel type, else
 * ZG_ERR_DIMENSION_MISMATCH / ZG_ERR_INVALID_ARGUMENT); complete on return. */
ZG_API int zg_image_upload(const zg_image *dst_dev, const zg_image *src_host, zg_stream stream);
ZG_API int zg_image_download(const zg_image *dst_host, const zg_image *src_dev, zg_stream stream);
ZG_API int zg_stream_create(zg_stream *out);
ZG_API int zg_stream_destroy(zg_stream s);
ZG_API int zg_stream_synchronize(zg_stream s);
ZG_API int zg_stream_wait_event(zg_stream s, zg_event e);
ZG_API int zg_event_create(zg_event *out);
ZG_API int zg_event_destroy(zg_event e);
ZG_API int zg_event_record(zg_event e, zg_stream s);
ZG_API int zg_event_synchronize(zg_event e);
ZG_API int zg_event_elapsed_ms(zg_event start, zg_event stop, float *ms); /* device time between two recorded events */
/* Everything the library enqueues on `stream` (not the default stream) between begin and end becomes one launchable
 * graph; host-side work of the captured calls (taps, tables, checks) is done at capture time. Scratch that captured calls
 * take belongs to the graph and is freed by zg_graph_destroy. Captures ended by somebody else (torch.cuda.graph around Image
 * calls, a caller's own hipStreamEndCapture) cannot be followed: their scratch stays reserved until zg_release_graph_scratch(),
 * to be called once those graphs are destroyed (it skips captures still in progress). */
ZG_API int zg_graph_begin_capture(zg_stream stream);
ZG_API int zg_graph_end_capture(zg_stream stream, zg_graph *out);
ZG_API int zg_graph_launch(zg_graph graph, zg_stream stream);
ZG_API int zg_graph_destroy(zg_graph graph);
ZG_API int zg_release_graph_scratch(void);
/* Idle scratch blocks the library keeps for reuse (at most ZIGNAL_HIP_SCRATCH_CACHE_MB, default 2048 MiB) go back to the driver. The
 * library does this itself when one of its own allocations (zg_malloc included) runs out of memory; another allocator in the same
 * process calls it before giving up. */
ZG_API int zg_trim_scratch(void);
ZG_API size_t zg_pixel_size(int pixel);

/* ---- filters ------------------------------------------------------------------------- */

/* Image(T).convolveSeparable (src/image.zig:935-951 -> src/image/convolution.zig:313-438,
 * worker :441-647). Types: U8, F32, RGB_U8, RGBA_U8 as in the reference; RGB_F32 / RGBA_F32
 * are an extension (the reference rejects them at comptime, convolution.zig:431-435) whose
 * per-channel result equals the F32 plane path. kx/ky are host pointers. */
ZG_API int zg_conv_separable(const zg_image *src, const zg_image *dst,
                             const float *kx, uint32_t nkx, const float *ky, uint32_t nky,
                             int border, zg_stream stream);
ZG_API int zg_conv_separable_host(const zg_image *src, const zg_image *dst,
                                  const float *kx, uint32_t nkx, const float *ky, uint32_t nky,
                                  int border);

/* Image(T).gaussianBlur (src/image.zig:954-994): radius = ceil(3 sigma), taps exp(-x^2/(2 s^2))
 * normalised in f32, then convolveSeparable(.mirror). sigma == 0 copies; sigma < 0 ->
 * ZG_ERR_INVALID_ARGUMENT (error.InvalidSigma). */
ZG_API int zg_gaussian_blur(const zg_image *src, const zg_image *dst, float sigma, zg_stream stream);
ZG_API int zg_gaussian_blur_host(const zg_image *src, const zg_image *dst, float sigma);
/* The 1-D taps gaussianBlur builds (src/image.zig:973-990). Returns the tap count (2*radius+1),
 * or a negative zg_status. `taps` may be NULL to query the count. */
ZG_API int zg_gaussian_kernel(float sigma, float *taps, uint32_t capacity);

/* convolveSeparable / gaussianBlur over n_planes images of one shape in ONE launch where the device has a kernel for it. The
 * reference's f32 route is per plane: Image(f32).convolveSeparable works, Image(Rgba(f32)) is a compile error
 * (src/image/convolution.zig:431-435), so a host that holds RGBA f32 data as four Image(f32) planes (the layout
 * splitChannels, src/image/channel_ops.zig:56-136, produces for the u8 structs) calls these with n_planes = 4 instead of
 * paying four launches. src / dst are host arrays of n_planes descriptors (device pixels); plane i goes src[i] -> dst[i],
 * each pair checked exactly as zg_conv_separable / zg_gaussian_blur check theirs (DimensionMismatch, InvalidSigma, ...)
 * before anything is launched. Any pixel type is accepted; planes that do not share a launch run one after the other on
 * `stream`. The result of every plane equals the single-plane call bit for bit. */
ZG_API int zg_conv_separable_planes(const zg_image *src, const zg_image *dst, uint32_t n_planes,
                                    const float *kx, uint32_t nkx, const float *ky, uint32_t nky,
                                    int border, zg_stream stream);
ZG_API int zg_gaussian_blur_planes(const zg_image *src, const zg_image *dst, uint32_t n_planes, float sigma, zg_stream stream);

/* Image(T).convolve (src/image.zig:917-932 -> src/image/convolution.zig:76-301); kernel is
 * row-major kh x kw f32 on the host. */
ZG_API int zg_convolve(const zg_image *src, const zg_image *dst,
                       const float *kernel, uint32_t kh, uint32_t kw, int border, zg_stream stream);
ZG_API int zg_convolve_host(const zg_image *src, const zg_image *dst,
                            const float *kernel, uint32_t kh, uint32_t kw, int border);

/* Image(T).boxBlur (src/image.zig:635-648 -> src/image/integral.zig:41-90,194-269):
 * f32 summed-area table, window and area clipped at the borders. src may equal dst. */
ZG_API int zg_box_blur(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream);
ZG_API int zg_box_blur_host(const zg_image *src, const zg_image *dst, uint32_t radius);

/* ---- resampling ---------------------------------------------------------------------- */

/* Image(T).resize (src/image.zig:523 -> src/image/interpolation.zig:89-214 and the u8 plane
 * kernels src/image/channel_ops.zig:144-493). Never fails in the reference (void). */
ZG_API int zg_resize(const zg_image *src, const zg_image *dst, const zg_method *method, zg_stream stream);
ZG_API int zg_resize_host(const zg_image *src, const zg_image *dst, const zg_method *method);

/* Image(Rgb(u8) / Rgba(u8)).resize(.lanczos) = resizePlaneLanczosU8 (src/image/channel_ops.zig:438-493) with the plane weights
 * made by the CALLER: the reference evaluates lanczosKernel with @sin for every destination column and row (:446-466), so a Zig
 * host computes wx[d * 6 + k] = lanczosKernel((k - 2) - frac((d + 0.5) * src_cols / dst_cols - 0.5)) for d < dst.cols, k < 6 (and wy
 * likewise over rows) with Zig's own @sin and hands them over; the library supplies only the integer tap indices and the f32
 * accumulation. Host pointers in both layers; NULL takes the library's own weights for that axis (zg_lanczos_plane_weights, which is
 * also what zg_resize uses). Pixel types other than RGB_U8 / RGBA_U8: ZG_ERR_UNSUPPORTED (they go through zg_method.lanczos_lut). */
ZG_API int zg_lanczos_plane_weights(uint32_t src_n, uint32_t dst_n, float *weights /* dst_n * 6 */);
ZG_API int zg_resize_lanczos_weights(const zg_image *src, const zg_image *dst, const float *wx, const float *wy, zg_stream stream);
ZG_API int zg_resize_lanczos_weights_host(const zg_image *src, const zg_image *dst, const float *wx, const float *wy);

/* Image(T).letterbox (src/image.zig:546 -> src/image/transforms.zig:49-108). Writes the content
 * rectangle {l,t,r,b} to rect_out (may be NULL). */
ZG_API int zg_letterbox(const zg_image *src, const zg_image *dst, const zg_method *method,
                        uint32_t rect_out[4], zg_stream stream);
ZG_API int zg_letterbox_host(const zg_image *src, const zg_image *dst, const zg_method *method,
                             uint32_t rect_out[4]);

/* Image(T).warp (src/image.zig:621 -> src/image/transforms.zig:522-531) with the project()
 * of src/geometry/transforms.zig:39-42 / :147-150 / :224-231. m is a host pointer. */
ZG_API int zg_warp(const zg_image *src, const zg_image *dst, int kind, const float *m,
                   const zg_method *method, zg_stream stream);
ZG_API int zg_warp_host(const zg_image *src, const zg_image *dst, int kind, const float *m,
                        const zg_method *method);

/* Image(T).rotateInto (src/image.zig:566 -> src/image/transforms.zig:163-212, exact
 * 0/90/180/270 paths :385-462). cos_a / sin_a are @cos(angle) / @sin(angle) as the caller's
 * maths library computes them (the Zig shim passes Zig's); zg_rotate_into_angle uses the
 * library's own. */
ZG_API int zg_rotate_into(const zg_image *src, const zg_image *dst, float angle,
                          float cos_a, float sin_a, const zg_method *method, int border, zg_stream stream);
ZG_API int zg_rotate_into_host(const zg_image *src, const zg_image *dst, float angle,
                               float cos_a, float sin_a, const zg_method *method, int border);
/* Image(T).rotateBounds (src/image/transforms.zig:112-148). */
ZG_API int zg_rotate_bounds(uint32_t rows, uint32_t cols, float angle, float cos_a, float sin_a,
                            uint32_t *out_rows, uint32_t *out_cols);

/* Image(T).extract (src/image.zig:593 -> src/image/transforms.zig:231-282); rect = {l,t,r,b}. */
ZG_API int zg_extract(const zg_image *src, const zg_image *dst, const float rect[4], float angle,
                      float cos_a, float sin_a, const zg_method *method, int border, zg_stream stream);
ZG_API int zg_extract_host(const zg_image *src, const zg_image *dst, const float rect[4], float angle,
                           float cos_a, float sin_a, const zg_method *method, int border);

/* Image(T).crop (src/image.zig:582 -> src/image/transforms.zig:216-222): dst must be
 * round(rect.height) x round(rect.width) (zg_crop_dims). Bit-exact copy. */
ZG_API int zg_crop(const zg_image *src, const zg_image *dst, const float rect[4], zg_stream stream);
ZG_API int zg_crop_host(const zg_image *src, const zg_image *dst, const float rect[4]);
ZG_API int zg_crop_dims(const float rect[4], uint32_t *out_rows, uint32_t *out_cols);

/* Image(T).flipLeftRight / flipTopBottom (src/image/transforms.zig:28-44), in place. */
ZG_API int zg_flip_left_right(const zg_image *img, zg_stream stream);
ZG_API int zg_flip_top_bottom(const zg_image *img, zg_stream stream);
ZG_API int zg_flip_left_right_host(const zg_image *img);
ZG_API int zg_flip_top_bottom_host(const zg_image *img);

/* Image(T).insert (src/image.zig:606 -> src/image/transforms.zig:293-378). blend_mode is the ordinal of the reference's
 * Blending enum (src/blending.zig:8-22: none 0, normal 1, multiply 2, screen 3, overlay 4, soft_light 5, hard_light 6,
 * color_dodge 7, color_burn 8, darken 9, lighten 10, difference 11, exclusion 12); anything but `.none` composites
 * Rgba(u8) sources with blendColors (blending.zig:27-157), other pixel types store the sample (assignPixel,
 * image.zig:67-94). source may have a different pixel type than self (`source: anytype`): samples are then converted with
 * convertColor, and Rgba(u8) samples composite through Rgba(u8) whatever self's type. self is modified in place. */
ZG_API int zg_insert(const zg_image *self, const zg_image *source, const float rect[4], float angle,
                     float cos_a, float sin_a, const zg_method *method, int blend_mode, zg_stream stream);
ZG_API int zg_insert_host(const zg_image *self, const zg_image *source, const float rect[4], float angle,
                          float cos_a, float sin_a, const zg_method *method, int blend_mode);

/* Image(T).copy (src/image.zig:375-392), Image(T).fill, Image(T).setBorder (:200). */
ZG_API int zg_copy(const zg_image *src, const zg_image *dst, zg_stream stream);
ZG_API int zg_fill(const zg_image *img, const void *pixel_value, zg_stream stream);
ZG_API int zg_set_border(const zg_image *img, const uint32_t rect[4], const void *pixel_value, zg_stream stream);
ZG_API int zg_fill_host(const zg_image *img, const void *pixel_value);
ZG_API int zg_set_border_host(const zg_image *img, const uint32_t rect[4], const void *pixel_value);

/* ---- colour -------------------------------------------------------------------------- */

/* Image(T).convertInto (src/image.zig:396-407 -> convertColor src/color.zig:108-151).
 * The pixel layout comes from the images, the colour space from the arguments, e.g.
 * (RGBA_U8, ZG_CS_RGBA) -> (RGB_F32, ZG_CS_OKLAB) is Image(Rgba(u8)).convert(Oklab(f32)). Any pair of colour spaces is
 * accepted (color.zig:350-948 routing tables, :987-1532 conversion functions): three-field colour types live in
 * RGB_F32 / RGB_U8 pixels, Rgba in RGBA_*, scalars in U8 / F32; float-only types (Hsl, Hsv, Lab, Lch, Lms, Oklab,
 * Oklch, Xyb, Xyz) need f32 pixels (ZG_ERR_UNSUPPORTED otherwise).
 * srgb_lut: optional 256-entry host table of gammaToLinear(i/255) (color.zig:1252-1258), so a
 * Zig caller can supply values made with Zig's std.math.pow; NULL -> the library's own. */
ZG_API int zg_convert(const zg_image *src, int src_space, const zg_image *dst, int dst_space,
                      const float *srgb_lut, zg_stream stream);
ZG_API int zg_convert_host(const zg_image *src, int src_space, const zg_image *dst, int dst_space,
                           const float *srgb_lut);

/* The pipeline steps [resize, convert] (reference src/cli/pipeline.zig:153-179; BASELINE configs[2]: Image(Rgba(u8)).resize(.bilinear)
 * then .convert(Oklab(f32))) in ONE call and, where a fused kernel exists (Rgba(u8) source, bilinear, Oklab / Xyz f32 destination),
 * one pass: the resized pixel never goes to memory. dst has the resized shape and the converted type. The result equals
 * zg_resize into a temporary followed by zg_convert, bit for bit, for every combination (the others run as those two steps
 * through scratch). */
ZG_API int zg_resize_convert(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const zg_method *method,
                             const float *srgb_lut, zg_stream stream);
ZG_API int zg_resize_convert_host(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const zg_method *method,
                                  const float *srgb_lut);

/* Diagnostics, not part of Image(T): the library's device-side maths (the restatements of Zig's std.math.cbrt / pow / exp / log /
 * sin / cos / atan2 that convertColor and motionBlur reach on the device, zignal_amd/csrc/zg_devmath.h) applied element-wise to
 * device arrays of n floats. fn: 0 cbrt, 1 pow(x, 2.4), 2 exp, 3 log, 4 sin, 5 cos, 6 atan2(x, y), 7 pow(x, y), 8 gammaToLinear
 * (src/color.zig:1252-1258), 9 cbrt by musl's steps verbatim (the device's faster cbrt, fn 0, is checked against it over all 2^32 inputs), 10 x / 100.0f, 11 the
 * device's division-free form of it (likewise), 12 / 13 labForward (src/color.zig:1289-1291) plain and in its branch-free form, 14 / 15 x / 95.047f and
 * 16 / 17 x / 108.883f by division and by the reciprocal-and-remainder form, 18 / 19 linearToGamma (src/color.zig:1243-1249) plain and branch-free,
 * 20 / 21, 22 / 23, 24 / 25 x / 116, x / 500, x / 200 likewise (each pair equal on all 2^32 inputs). y_dev may be NULL for the unary functions. This is how the transcendental boundary is swept against
 * the oracle and against correctly rounded values (tests/test_math_pin.py). */
ZG_API int zg_devmath_apply(int fn, const float *x_dev, const float *y_dev, float *out_dev, size_t n, zg_stream stream);

/* ---- next rows of the scope table (callers of the path) ------------------------ */

/* Image(T).sobel (src/image.zig:1001-1010 -> src/image/edges.zig:33-70): grey f32, two 3x3 f32 convolutions
 * (.replicate), sqrt(gx^2 + gy^2) / 4 truncated to u8 — fused into one kernel. dst is Image(u8). */
ZG_API int zg_sobel(const zg_image *src, const zg_image *dst, zg_stream stream);
ZG_API int zg_sobel_host(const zg_image *src, const zg_image *dst);

/* Image(T).sharpen (src/image.zig:785-801 -> Integral.sharpen, src/image/integral.zig:273-426): 2 * original - boxBlur
 * with the same integral image; radius 0 copies. */
ZG_API int zg_sharpen(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream);
ZG_API int zg_sharpen_host(const zg_image *src, const zg_image *dst, uint32_t radius);
/* Image(T).integral (src/image.zig:628-630 -> Integral.compute, integral.zig:95-140): one f32 summed-area plane of
 * rows x cols per channel, channel-major in `planes` (channels * rows * cols floats, packed), in the reference's
 * summation order. */
ZG_API int zg_integral(const zg_image *src, float *planes, zg_stream stream);
ZG_API int zg_integral_host(const zg_image *src, float *planes);
/* Image(T).invert (src/image.zig:494-513), in place: 255 - v / 1 - v per colour channel, alpha kept; Image(f32) is
 * rejected as in the reference (ZG_ERR_UNSUPPORTED). */
ZG_API int zg_invert(const zg_image *img, zg_stream stream);
ZG_API int zg_invert_host(const zg_image *img);

/* Image(T).medianBlur / percentileBlur / minBlur / maxBlur / midpointBlur / alphaTrimmedMeanBlur (src/image.zig:653-783
 * -> src/image/order_statistic_blur.zig) for u8 and all-u8 struct pixels, per channel. op 0: percentile (param in
 * [0, 1]; median = 0.5 with ZG_BORDER_MIRROR, min = 0.0, max = 1.0), 1: midpoint, 2: alpha-trimmed mean (param = trim
 * fraction in [0, 0.5)). InvalidPercentile / InvalidTrim -> ZG_ERR_INVALID_ARGUMENT, UnsupportedPixelType and radius > 15
 * -> ZG_ERR_UNSUPPORTED. radius 0 copies. src may alias dst. */
ZG_API int zg_order_statistic_blur(const zg_image *src, const zg_image *dst, uint32_t radius, int op, double param, int border,
                                   zg_stream stream);
ZG_API int zg_order_statistic_blur_host(const zg_image *src, const zg_image *dst, uint32_t radius, int op, double param, int border);

/* Image(T).autocontrast / equalize (src/image.zig:804-829 -> src/image/enhancement.zig), in place, for u8, Rgb(u8),
 * Rgba(u8) (ZG_ERR_UNSUPPORTED otherwise). cutoff outside [0, 0.5) is error.InvalidCutoff -> ZG_ERR_INVALID_ARGUMENT. */
ZG_API int zg_autocontrast(const zg_image *img, float cutoff, zg_stream stream);
ZG_API int zg_autocontrast_host(const zg_image *img, float cutoff);
ZG_API int zg_equalize(const zg_image *img, zg_stream stream);
ZG_API int zg_equalize_host(const zg_image *img);

/* Image(u8) binarisation and binary morphology (src/image.zig:845-914 -> src/image/binary.zig). Image(u8) only
 * (ZG_ERR_UNSUPPORTED otherwise, a compile error in the reference).
 * thresholdOtsu (:38-84): out = src > t ? 255 : 0; *threshold_out (host pointer, may be NULL) receives t, which
 * synchronises `stream`. thresholdAdaptiveMean (:86-118): out = src > windowMean(radius) - c; radius 0 is
 * error.InvalidRadius -> ZG_ERR_INVALID_ARGUMENT. zg_morph (:121-281): op 0 dilate, 1 erode, 2 open, 3 close; kernel is a
 * host array of kernel_rows x kernel_cols bytes (non-zero = on; odd sizes, else error.InvalidKernelSize); src may alias dst. */
ZG_API int zg_threshold_otsu(const zg_image *src, const zg_image *dst, uint8_t *threshold_out, zg_stream stream);
ZG_API int zg_threshold_otsu_host(const zg_image *src, const zg_image *dst, uint8_t *threshold_out);
ZG_API int zg_threshold_adaptive_mean(const zg_image *src, const zg_image *dst, uint32_t radius, float c, zg_stream stream);
ZG_API int zg_threshold_adaptive_mean_host(const zg_image *src, const zg_image *dst, uint32_t radius, float c);
ZG_API int zg_morph(const zg_image *src, const zg_image *dst, const uint8_t *kernel, uint32_t kernel_rows, uint32_t kernel_cols,
                    uint32_t iterations, int op, zg_stream stream);
ZG_API int zg_morph_host(const zg_image *src, const zg_image *dst, const uint8_t *kernel, uint32_t kernel_rows, uint32_t kernel_cols,
                         uint32_t iterations, int op);

/* Image(T).canny (src/image.zig:1047-1063 -> src/image/edges.zig:212-277): grey -> the detector's own Gaussian
 * (.replicate; sigma == 0 skips it) -> Sobel gradients -> non-maximum suppression -> double threshold + hysteresis.
 * dst is Image(u8), 0 or 255. error.InvalidParameter / InvalidSigma / InvalidThreshold -> ZG_ERR_INVALID_ARGUMENT.
 * Hysteresis is connected-component labelling in a fixed number of launches: the call is asynchronous on `stream`. */
ZG_API int zg_canny(const zg_image *src, const zg_image *dst, float sigma, float low_threshold, float high_threshold, zg_stream stream);
ZG_API int zg_canny_host(const zg_image *src, const zg_image *dst, float sigma, float low_threshold, float high_threshold);

/* Diagnostics, not part of Image(T): shenCastan's smoothing stage on its own — isefFilter2D (src/image/edges.zig:308-349, a private
 * function there): isefFilter1D (:283-305) along every row, then along every column, of a contiguous plane on the device: Image(f32), or
 * Image(u8) taken as as(f32, u8) (what shenCastan feeds it, and how the detector's byte plane reaches the row pass). src -> dst (Image(f32);
 * dst may not alias src). The device runs the recursions in overlapping segments and proves each segment's start against its
 * predecessor's exact value (zignal_amd/csrc/isef.hip); this entry point is how that is held to the sequential recursion bit for bit
 * (tests/test_next_rows.py). */
ZG_API int zg_isef_smooth(const zg_image *src, const zg_image *dst, float smooth, zg_stream stream);

/* Image(T).shenCastan (src/image.zig:1015-1027 -> src/image/edges.zig:83-196; options src/image/ShenCastan.zig:9-45:
 * smooth 0.9, window_size 7, high_ratio 0.99, low_rel 0.5, hysteresis true, use_nms false by default). dst is Image(u8),
 * 0 or 255. InvalidBParameter / WindowSizeMustBeOdd / WindowSizeTooSmall / InvalidThreshold -> ZG_ERR_INVALID_ARGUMENT.
 * Asynchronous on `stream` with or without hysteresis (the thresholds never leave the device). */
ZG_API int zg_shen_castan(const zg_image *src, const zg_image *dst, float smooth, uint32_t window_size, float high_ratio, float low_rel,
                          int hysteresis, int use_nms, zg_stream stream);
ZG_API int zg_shen_castan_host(const zg_image *src, const zg_image *dst, float smooth, uint32_t window_size, float high_ratio, float low_rel,
                               int hysteresis, int use_nms);

/* Image(T).motionBlur (src/image.zig:1077-1091 -> src/image/motion_blur.zig). `.linear` (:65-236): distance 0 copies;
 * |sin| or |cos| < 0.001 is the separable convolution with a uniform kernel (.replicate); anything else averages
 * bilinear samples along the motion line. cos_a / sin_a are @cos(angle) / @sin(angle) as the caller's maths library
 * computes them. `.radial_zoom` / `.radial_spin` (:240-440): spin = 0 / 1; centre normalised to [0, 1]. */
ZG_API int zg_motion_blur_linear(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a, uint32_t distance, zg_stream stream);
ZG_API int zg_motion_blur_linear_host(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a, uint32_t distance);
ZG_API int zg_motion_blur_radial(const zg_image *src, const zg_image *dst, float center_x, float center_y, float strength, int spin, zg_stream stream);
ZG_API int zg_motion_blur_radial_host(const zg_image *src, const zg_image *dst, float center_x, float center_y, float strength, int spin);

/* ImagePyramid.build (src/image/pyramid.zig:31-102) is gaussianBlur + resize(.bilinear) per level; these two give the
 * per-level arithmetic. scale = pow(scale_factor, level): zg_pyramid_scale is the library's restatement of Zig's
 * std.math.pow; a Zig caller passes its own value to zg_pyramid_level. A level below 8 x 8 truncates the pyramid;
 * the level is blurred only when *out_sigma > 0.5. */
ZG_API float zg_pyramid_scale(float scale_factor, uint32_t level);
ZG_API int zg_pyramid_level(uint32_t rows, uint32_t cols, float scale, float blur_sigma,
                            uint32_t *out_rows, uint32_t *out_cols, float *out_sigma);
/* One level in one call (pyramid.zig:76-92): gaussianBlur(source, sigma) when sigma > 0.5 (scratch inside), then
 * resize(.bilinear) into `level` (pre-allocated with zg_pyramid_level's dimensions). Device pointers. */
ZG_API int zg_pyramid_build_level(const zg_image *source, const zg_image *level, float sigma, zg_stream stream);
/* ImagePyramid.build (pyramid.zig:31-102) as ONE device operation: levels[i] / sigmas[i] are pyramid level i + 1 (level 0 is the
 * source itself: no copy, pyramid.zig:54), n_levels of them, shapes and sigmas from zg_pyramid_level, memory from the caller. Every
 * level is made from the original, so they are independent: the call forks them over internal streams and joins them back into
 * `stream` with events — asynchronous, no host round trip between levels, recordable into a graph. */
ZG_API int zg_pyramid_build(const zg_image *source, const zg_image *levels, const float *sigmas, uint32_t n_levels, zg_stream stream);

/* ---- batch (config: N frames, gaussianBlur(sigma) then bilinear resize) ----------------- */

/* Semantics of the `pipeline` recipe [blur gaussian, resize] (src/cli/pipeline.zig:153-179)
 * applied to n_frames images laid out back to back. scratch may be NULL (allocated inside). */
ZG_API int zg_batch_blur_resize(const void *src_frames, uint32_t n_frames,
                                uint32_t rows, uint32_t cols, int pixel, float sigma,
                                void *dst_frames, uint32_t out_rows, uint32_t out_cols,
                                const zg_method *method, zg_stream stream);

/* The `pipeline` command in general (src/cli/pipeline.zig:153-179: `for (steps) |step| current = step.apply(current)` on every input image)
 * over a batch of n_frames equally shaped images laid out back to back: frame f of the result is step[n-1](... step[0](frame f)).
 * Every step runs as ONE launch over the whole batch where a batched kernel exists (the u8 Gaussians, Rgba(u8) bilinear resize,
 * every conversion; a frame is too short a launch to fill the chip), frame by frame otherwise; consecutive steps the library has a
 * fused kernel for ([gaussian blur, resize to half size] and [resize, convert to Oklab / Xyz] on Rgba(u8)) run as that kernel, the
 * rest hands frames on through scratch. Results equal the per-frame calls bit for bit.
 * Steps are the CLI's three (src/cli/pipeline.zig:170-172) with every variant each of them has — resize (src/cli/resize.zig:77-99);
 * blur: box, gaussian, median, motion_linear, motion_zoom, motion_spin (src/cli/blur.zig:98-170); edges: sobel, canny, shen_castan
 * through the grey bridge (src/cli/edges.zig:85-135) — plus convert and warp. */
typedef enum zg_step_kind {
    ZG_STEP_GAUSSIAN_BLUR = 0, /* Image.gaussianBlur(sigma) */
    ZG_STEP_BOX_BLUR = 1,      /* Image.boxBlur(radius) */
    ZG_STEP_RESIZE = 2,        /* Image.resize(out_rows x out_cols, method) */
    ZG_STEP_CONVERT = 3,       /* Image.convert(dst_pixel / dst_space) */
    ZG_STEP_WARP = 4,          /* Image.warp(transform, m, method) into out_rows x out_cols */
    ZG_STEP_MEDIAN_BLUR = 5,   /* Image.medianBlur(radius) */
    ZG_STEP_MOTION_BLUR = 6,   /* Image.motionBlur(.linear / .radial_zoom / .radial_spin) */
    ZG_STEP_EDGES = 7          /* edges.apply (src/cli/edges.zig:126-135): frame.convert(u8) -> detector -> .convert(the frames' own type) */
} zg_step_kind;
typedef enum zg_motion_kind { ZG_MOTION_LINEAR = 0, ZG_MOTION_RADIAL_ZOOM = 1, ZG_MOTION_RADIAL_SPIN = 2 } zg_motion_kind;
typedef enum zg_edges_kind { ZG_EDGES_SOBEL = 0, ZG_EDGES_CANNY = 1, ZG_EDGES_SHEN_CASTAN = 2 } zg_edges_kind;
typedef struct zg_step {
    int kind;                    /* zg_step_kind */
    float sigma;                 /* GAUSSIAN_BLUR; EDGES: canny's sigma, Shen-Castan's smooth */
    uint32_t radius;             /* BOX_BLUR, MEDIAN_BLUR */
    uint32_t out_rows, out_cols; /* RESIZE, WARP: the shape of the step's output frames */
    zg_method method;            /* RESIZE, WARP */
    int dst_pixel, dst_space;    /* CONVERT: zg_pixel, zg_colorspace of the step's output */
    const float *srgb_lut;       /* CONVERT: as zg_convert (host pointer, may be NULL) */
    int transform;               /* WARP: zg_transform */
    float m[9];                  /* WARP: as zg_warp */
    int motion;                  /* MOTION_BLUR: zg_motion_kind */
    float angle, cos_a, sin_a;   /* MOTION_BLUR linear: as zg_motion_blur_linear (the caller's cos / sin of the angle) */
    uint32_t distance;           /* MOTION_BLUR linear */
    float center_x, center_y, strength; /* MOTION_BLUR radial */
    int edges;                   /* EDGES: zg_edges_kind */
    float low, high;             /* EDGES: canny's low / high threshold, Shen-Castan's low_rel / high_ratio */
    uint32_t window;             /* EDGES: Shen-Castan's window_size */
    int use_nms;                 /* EDGES: Shen-Castan's use_nms (hysteresis stays at its default, on, as in the CLI) */
} zg_step;
/* sizeof(zg_step) as this library was built. zg_step grows with the recipe language (round 4 added thirteen fields) and carries no size field of
 * its own: a binding compares its own struct's size with this and refuses to run on a mismatch (zignal_amd/_lib.py at load time, zignal_hip.hpp in Pipeline's
 * constructor, the Zig shim at the head of Pipeline.run / runMulti / outShape), so a caller built against another header can never hand over arrays with the wrong stride. */
ZG_API size_t zg_sizeof_step(void);
/* Host only: shape and type of the frames after the steps (what dst_frames of zg_batch_pipeline must hold, n_frames times). */
ZG_API int zg_batch_pipeline_shape(uint32_t rows, uint32_t cols, int pixel, int space, const zg_step *steps, uint32_t n_steps,
                                   uint32_t *out_rows, uint32_t *out_cols, int *out_pixel, int *out_space);
/* Device pointers; `space` is the colour space of the source frames (only CONVERT steps look at it). Asynchronous on `stream`. */
ZG_API int zg_batch_pipeline(const void *src_frames, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, int space,
                             const zg_step *steps, uint32_t n_steps, void *dst_frames, zg_stream stream);

/* ---- the node's GPUs from one host process (BASELINE configs[4]) --------------------------------------------------------------- */

/* Two routes to several GPUs:
 *   (1) one host thread per GPU, each calling zg_set_device(i) once and then the ordinary entry points — no library state is
 *       shared between devices (scratch, tables and streams are per device);
 *   (2) a zg_multi context: ONE thread drives every GPU of the context. The root (first) device holds the batch; the shards go
 *       to their owners and the results come back over RCCL (grouped ncclSend / ncclRecv: one shard per xGMI peer link, no
 *       ring), each device runs zg_batch_blur_resize on its shard on its own stream; frames are independent, so there is no
 *       halo and no collective on the data path. librccl is loaded at first use (dlopen).
 * devices == NULL means 0 .. n_devices - 1; n_devices <= 0 means every visible device. */
typedef void *zg_multi;
ZG_API int zg_multi_create(const int *devices, int n_devices, zg_multi *out);
ZG_API int zg_multi_destroy(zg_multi m);
ZG_API int zg_multi_device_count(zg_multi m);
/* Orders the context's next batch call behind everything enqueued so far on `producer`, a stream of the ROOT device (NULL = its legacy
 * default stream), without host synchronisation; may be called once per producing stream. A batch call that was NOT preceded by this
 * synchronises the root device instead, so it is safe whatever stream produced its frames. */
ZG_API int zg_multi_wait_stream(zg_multi m, zg_stream producer);
/* zg_batch_blur_resize over the context's devices. src_frames_root / dst_frames_root are device pointers ON THE ROOT DEVICE holding
 * all n_frames input frames / receiving all output frames. Device i owns a contiguous block of frames (sizes differ by at
 * most one) and receives it in up to four pieces: transfer, kernel and return trip of consecutive pieces overlap. Synchronous:
 * results are complete on return. The call waits for whatever produced src_frames_root: the streams named with zg_multi_wait_stream
 * since the previous batch call (stream waits, the host does not block), or, when none was named, the whole root device. times_ms
 * (may be NULL) receives milliseconds of {the scatter stream from first to last send (device events), the busiest device's kernels
 * from first to last piece (device events), the whole call (host clock)} — the three overlap, they do not add up. After a failure
 * inside the exchange the context refuses further calls (ZG_ERR_INVALID_ARGUMENT): destroy and re-create it. */
ZG_API int zg_multi_batch_blur_resize(zg_multi m, const void *src_frames_root, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel,
                                      float sigma, void *dst_frames_root, uint32_t out_rows, uint32_t out_cols, const zg_method *method,
                                      float times_ms[3]);
/* zg_batch_pipeline over the context's devices: every frame through the recipe's steps (src/cli/pipeline.zig:153-179), the frames sharded
 * exactly as above (contiguous blocks, pieces, two communicators). dst_frames_root holds n_frames frames of the shape and type
 * zg_batch_pipeline_shape reports. Same waiting, timing and failure rules as zg_multi_batch_blur_resize. */
ZG_API int zg_multi_batch_pipeline(zg_multi m, const void *src_frames_root, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, int space,
                                   const zg_step *steps, uint32_t n_steps, void *dst_frames_root, float times_ms[3]);
/* Host only, no device touched: frames [*begin, *end) of the batch that form piece `piece` of device `device`'s shard when n_frames frames go
 * to `world` devices in up to `chunks` (1..8) pieces per shard — the arithmetic the two calls above use (zignal_amd/sharding.py cuts the same
 * way). Pieces past a shard's last are empty (*begin == *end). */
ZG_API int zg_multi_piece_range(uint32_t n_frames, int world, int chunks, int device, int piece, uint32_t *begin, uint32_t *end);

/* ---- the host I/O edge: PNG (src/codecs/png.zig) -------------------------- */

/* Where frames come from and go to. The entropy-coded layers stay on the host (the chunk layer with the reference's
 * ordering rules and limits, zlib inflate / deflate, and de-filtering, which is a serial recurrence along and across
 * rows); the per-pixel layers run on the device: sample unpacking (1/2/4/8/16 bit, palette, tRNS, Adam7 placement,
 * conversion to the requested Image(T)) on the way in, filter selection and row filtering on the way out. */
typedef struct zg_png_header { /* png.Header (png.zig:135-149) */
    uint32_t width, height;
    uint8_t bit_depth, color_type /* 0 grey, 2 rgb, 3 palette, 4 grey+alpha, 6 rgba */, compression_method, filter_method, interlace_method;
    uint8_t has_gamma, has_srgb, srgb_intent;
    float gamma;
} zg_png_header;
typedef struct zg_png_limits { /* png.DecodeLimits (png.zig:23-41); a zero disables that limit */
    size_t max_png_bytes, max_chunk_bytes, max_idat_bytes, max_chunks;
    uint32_t max_width, max_height;
    uint64_t max_pixels;
    size_t max_decompressed_bytes;
} zg_png_limits;
typedef struct zg_png_encode_options { /* png.EncodeOptions (png.zig:1296-1316) */
    int filter;            /* ZG_PNG_FILTER_ADAPTIVE (the default), or a fixed FilterType 0 none, 1 sub, 2 up, 3 average, 4 paeth */
    int compression_level; /* zlib level 0..9; negative = the library default (5, Z_FILTERED: the reference's "filtered" preset) */
    int has_gamma;         /* write a gAMA chunk (ignored when srgb_intent >= 0, as in the reference) */
    float gamma;
    int srgb_intent;       /* 0..3 writes an sRGB chunk; negative = none */
} zg_png_encode_options;
#define ZG_PNG_FILTER_ADAPTIVE (-1)

ZG_API void zg_png_default_limits(zg_png_limits *limits);                 /* DecodeLimits{} */
ZG_API void zg_png_default_encode_options(zg_png_encode_options *options); /* EncodeOptions.default */
/* png.getInfo (png.zig:308-410): header + gAMA / sRGB metadata, reading no further than the first IDAT. limits may be NULL. */
ZG_API int zg_png_info(const uint8_t *png, size_t len, const zg_png_limits *limits, zg_png_header *out);
/* png.decode (png.zig:629-794), the chunk layer only: validates every chunk (order, CRC, limits), and reports the header,
 * the pixel type png.toNativeImage would produce (ZG_PIXEL_U8 / RGB_U8 / RGBA_U8, png.zig:852-1146) and whether the file
 * is cut short (missing IEND). Nothing is inflated. */
ZG_API int zg_png_probe(const uint8_t *png, size_t len, const zg_png_limits *limits, zg_png_header *header_out, int *native_pixel_out,
                        int *truncated_out);
/* The host half of a decode on its own (png.decode + the inflate / recovery / de-filter part of png.toNativeImage, png.zig:
 * 801-852, :1721-1803, and the palette-index check of :1080 / :1119): FNV-1a over the de-filtered scan data, filter bytes
 * included. No device is touched: this is how the host layers are compared with the reference's without a GPU. */
ZG_API int zg_png_scan_hash(const uint8_t *png, size_t len, const zg_png_limits *limits, uint64_t *hash_out, int *truncated_out);
/* png.loadFromBytes(T) (png.zig:1151-1186): decode into `dst` (rows x cols must equal the header's height x width, else
 * ZG_ERR_DIMENSION_MISMATCH). dst's pixel type and dst_space name T exactly as in zg_convert: when T is not the native
 * type the native image is converted with Image.convert. Cut pixel data decodes partially (whole rows kept, the rest
 * zero) and sets *truncated_out (may be NULL). `png` is host memory; dst is device memory (zg_png_decode) or host memory
 * (zg_png_decode_host). The host part (inflate, de-filter) completes before the call returns; the upload and the device
 * kernels are ordered on `stream`. */
ZG_API int zg_png_decode(const uint8_t *png, size_t len, const zg_png_limits *limits, const zg_image *dst, int dst_space, int *truncated_out,
                         zg_stream stream);
ZG_API int zg_png_decode_host(const uint8_t *png, size_t len, const zg_png_limits *limits, const zg_image *dst, int dst_space, int *truncated_out);
/* filterScanlines / filterScanlinesAdaptive (png.zig:1265-1294, :1661-1719) on an 8-bit Image(u8 / Rgb / Rgba): writes
 * rows * (1 + cols * channels) bytes to `filtered` (device memory): per row the filter byte, then the filtered bytes.
 * Adaptive: per-row costs of all five filters in one pass, the reference's sampling state machine on the device, then
 * the chosen filter per row. Asynchronous on `stream`. */
ZG_API int zg_png_filter(const zg_image *src, int filter, uint8_t *filtered, zg_stream stream);
/* The IDAT payload of encodeRaw (png.zig:1297-1306, :1372-1391): the zlib stream of `len` bytes of filtered scanlines in
 * host memory (what zg_png_filter wrote, copied back), at EncodeOptions.compression_level (negative: the default, zlib
 * level 5 "filtered"). Host only, no device call: with zg_png_filter it is zg_png_encode taken apart, for callers that
 * filter many frames on the device and deflate elsewhere. Inputs of 4 MiB and more are deflated on up to 16 host threads
 * (ZIGNAL_HIP_HOST_THREADS overrides) as one stream any inflater reads. *out is malloc'd: zg_png_free. */
ZG_API int zg_png_compress(const uint8_t *scanlines, size_t len, int compression_level, uint8_t **out, size_t *out_len);
/* png.encode(T) (png.zig:1400-1425): Image(u8) -> greyscale, Rgb -> RGB, Rgba -> RGBA, anything else (src_space as in
 * zg_convert) is converted to Rgb first. *out is malloc'd host memory holding the file, release it with zg_png_free.
 * options may be NULL (EncodeOptions.default). The device filters, the host deflates: the call synchronises `stream`. */
ZG_API int zg_png_encode(const zg_image *src, int src_space, const zg_png_encode_options *options, uint8_t **out, size_t *out_len, zg_stream stream);
ZG_API int zg_png_encode_host(const zg_image *src, int src_space, const zg_png_encode_options *options, uint8_t **out, size_t *out_len);
ZG_API void zg_png_free(void *p);

/* ---- the host I/O edge: JPEG decode (src/codecs/jpeg.zig) -------------------- */

/* Baseline (SOF0) and progressive (SOF2) DCT JPEG, one component (grey) or three (YCbCr at 4:4:4, 4:2:2, 4:1:1, 4:2:0).
 * Host, inside the library: marker parsing with the reference's errors and limits, Huffman decoding of every scan into
 * coefficient blocks (bit-serial by nature), including the reference's handling of restart markers and of cut streams
 * (the blocks decoded so far are kept). Device: dequantisation, the integer IDCT, the level shift, chroma upsampling,
 * YCbCr -> RGB, cropping to width x height and the conversion to the requested Image(T). Errors come back as
 * ZG_ERR_CODEC with the Zig error name first in zg_last_error() ("InvalidHuffmanCode", "UnsupportedSamplingFactor", ...). */
typedef struct zg_jpeg_header { /* jpeg.Header (jpeg.zig:61-74) */
    uint32_t width, height;
    uint8_t precision, num_components, progressive /* frame_type: 0 baseline, 1 progressive */;
    int8_t subsampling; /* Subsampling (jpeg.zig:260-283): 0 yuv444, 1 yuv422, 2 yuv420, -1 null */
} zg_jpeg_header;
typedef struct zg_jpeg_limits { /* jpeg.DecodeLimits (jpeg.zig:19-33); a zero disables that limit */
    size_t max_jpeg_bytes, max_marker_bytes;
    uint32_t max_width, max_height;
    uint64_t max_pixels;
    size_t max_blocks, max_scans;
} zg_jpeg_limits;
ZG_API void zg_jpeg_default_limits(zg_jpeg_limits *limits);
/* jpeg.getInfo (jpeg.zig:77-179): the first SOFn's header. The native pixel type of the file is ZG_PIXEL_U8 for one
 * component and ZG_PIXEL_RGB_U8 otherwise (jpeg.toNativeImage, :2786-2821). limits may be NULL. */
ZG_API int zg_jpeg_info(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, zg_jpeg_header *out);
/* jpeg.decode (jpeg.zig:2035-2151) on the host only: every marker is parsed and validated, the scans of a progressive
 * file are entropy-decoded (a baseline file stops at its SOS, as in the reference), nothing is rendered. Reports the frame
 * header and JpegState.scan_limit_reached. */
ZG_API int zg_jpeg_probe(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, zg_jpeg_header *header_out, int *scan_limit_reached_out);
/* The host half of a decode on its own: jpeg.decode, plus performBlockScan (jpeg.zig:2397-2479) for a baseline file, then
 * FNV-1a over the coefficient blocks (one i32 per step, component by component) — the state jpeg.toNativeImage starts
 * from. No device is touched: this is how the entropy decoders are compared with the reference's without a GPU. */
ZG_API int zg_jpeg_coefficient_hash(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, uint64_t *hash_out);
/* jpeg.loadFromBytes(T) (jpeg.zig:2825-2851): dst is rows x cols == height x width of the frame header
 * (ZG_ERR_DIMENSION_MISMATCH otherwise); its pixel type and dst_space name T as in zg_convert (the native image goes
 * through Image.convert when T differs). *scan_limit_reached_out (may be NULL) reports JpegState.scan_limit_reached.
 * `jpeg` is host memory; dst is device memory (zg_jpeg_decode) or host memory (zg_jpeg_decode_host). */
ZG_API int zg_jpeg_decode(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, const zg_image *dst, int dst_space,
                          int *scan_limit_reached_out, zg_stream stream);
ZG_API int zg_jpeg_decode_host(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, const zg_image *dst, int dst_space,
                               int *scan_limit_reached_out);
/* jpeg.encode(T) (jpeg.zig:307-329): baseline SOF0 with the reference's fixed Huffman tables. Image(u8) becomes a one-component
 * file, Rgb a YCbCr one at 4:4:4 / 4:2:2 / 4:2:0, any other T (src_space as in zg_convert) is converted to Rgb first.
 * Device: colour conversion, edge replication, chroma averaging, the LLM forward DCT, reciprocal quantisation; host: the
 * Huffman coder. The file is a deterministic function of pixels and options, byte for byte the reference's. *out is
 * malloc'd host memory, release it with zg_jpeg_free. options may be NULL (EncodeOptions.default). A 0 x n image is
 * error.InvalidImageDimensions, more than 65535 rows or columns error.ImageTooLarge (ZG_ERR_CODEC). Synchronises `stream`. */
typedef struct zg_jpeg_encode_options { /* jpeg.EncodeOptions (jpeg.zig:284-290) */
    int quality;            /* 1..100 (clamped), default 90 */
    int subsampling;        /* 0 yuv444, 1 yuv422, 2 yuv420 (default) */
    int density_dpi;        /* JFIF density, default 72 */
    const uint8_t *comment; /* COM segment, or NULL */
    size_t comment_len;
} zg_jpeg_encode_options;
ZG_API void zg_jpeg_default_encode_options(zg_jpeg_encode_options *options);
ZG_API int zg_jpeg_encode(const zg_image *src, int src_space, const zg_jpeg_encode_options *options, uint8_t **out, size_t *out_len, zg_stream stream);
ZG_API int zg_jpeg_encode_host(const zg_image *src, int src_space, const zg_jpeg_encode_options *options, uint8_t **out, size_t *out_len);
/* The host half of zg_jpeg_encode on its own: the container and the entropy-coded scan (encodeRgb :929-975,
 * encodeGrayscale :977-1043, encodeBlock :771-817) around quantised coefficient blocks in host memory, laid out as the
 * device half writes them: 64 int16 per block in natural order; luma blocks row-major on the
 * (mcus_y * v) x (mcus_x * h) grid, then (gray == 0) all Cb blocks, then all Cr blocks on the mcus_y x mcus_x grid, where
 * h x v is 1 x 1 / 2 x 1 / 2 x 2 for yuv444 / yuv422 / yuv420 and mcus = ceil(size / (8 h, 8 v)). Quantisation tables in
 * the file come from options->quality. Host only, no device call. Large frames are coded in bands of MCU rows on up to
 * 16 host threads (ZIGNAL_HIP_HOST_THREADS overrides) and spliced bit-exactly: the bytes do not depend on the thread
 * count. *out is malloc'd: zg_jpeg_free. */
ZG_API int zg_jpeg_encode_blocks(const int16_t *blocks, uint32_t rows, uint32_t cols, int gray, const zg_jpeg_encode_options *options, uint8_t **out,
                                 size_t *out_len);
ZG_API void zg_jpeg_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* ZIGNAL_HIP_H */
