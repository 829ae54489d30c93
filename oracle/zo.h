/*
 * oracle/zo.h — CPU restatement of zignal's per-pixel image hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (libzignal_hip.so) never links, loads or calls it.
 *
 * Parity pinning: the reference (Zig) cannot be compiled in this image (no zig toolchain, see
 * DESIGN.md), so this oracle is pinned against the known-answer values of the reference's own
 * unit tests (tests/test_oracle_*.py cite them file:line). Anything that flows through Zig's
 * std maths (@exp, @sin, @cos, std.math.pow, std.math.cbrt) is restated from memory of the
 * musl/Go algorithms Zig ports and is PARITY UNPINNED at the last ulp (zigmath.c header).
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, no fast-math: the reference uses
 * strict IEEE f32 with separate mul and add everywhere on this path).
 */
#ifndef ZO_H
#define ZO_H
#include <stddef.h>
#include <stdint.h>

#define ZO_API __attribute__((visibility("default")))

/* same ordinals as include/zignal_hip.h */
enum { ZO_U8 = 0, ZO_F32 = 1, ZO_RGB_U8 = 2, ZO_RGBA_U8 = 3, ZO_RGB_F32 = 4, ZO_RGBA_F32 = 5 };
enum { ZO_ZERO = 0, ZO_REPLICATE = 1, ZO_MIRROR = 2, ZO_WRAP = 3 };
enum { ZO_NEAREST = 0, ZO_BILINEAR = 1, ZO_BICUBIC = 2, ZO_CATMULL_ROM = 3, ZO_MITCHELL = 4, ZO_LANCZOS = 5 };
enum { ZO_SIMILARITY = 0, ZO_AFFINE = 1, ZO_PROJECTIVE = 2 };
enum { ZO_CS_GRAY = 0, ZO_CS_RGB = 1, ZO_CS_RGBA = 2, ZO_CS_OKLAB = 3, ZO_CS_XYZ = 4, ZO_CS_YCBCR = 5,
       ZO_CS_HSL = 6, ZO_CS_HSV = 7, ZO_CS_LAB = 8, ZO_CS_LCH = 9, ZO_CS_LMS = 10, ZO_CS_OKLCH = 11, ZO_CS_XYB = 12 };

typedef struct zo_image {
    void *data;
    size_t stride; /* pixels */
    uint32_t rows, cols;
    int32_t pixel;
} zo_image;

static inline int zo_channels(int pixel) {
    switch (pixel) {
    case ZO_U8: case ZO_F32: return 1;
    case ZO_RGB_U8: case ZO_RGB_F32: return 3;
    default: return 4;
    }
}
static inline int zo_is_float(int pixel) { return pixel == ZO_F32 || pixel == ZO_RGB_F32 || pixel == ZO_RGBA_F32; }
static inline size_t zo_pixel_size(int pixel) { return (size_t)zo_channels(pixel) * (zo_is_float(pixel) ? 4 : 1); }

/* border.c — reference src/image/border.zig:46-63. Returns -1 for `null`. */
ZO_API int64_t zo_resolve_index(int64_t idx, int64_t length, int border);

/* meta.c-ish helpers (reference src/meta.zig:110-135) */
static inline uint8_t zo_clamp_u8_i64(int64_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
ZO_API uint8_t zo_clamp_u8_f32(float v);

/* zigmath.c — restated Zig std / compiler-rt maths (parity unpinned, see file header) */
ZO_API float zo_expf(float x);
ZO_API float zo_logf(float x);
ZO_API float zo_powf(float x, float y);
ZO_API float zo_cbrtf(float x);
ZO_API float zo_sinf(float x);
ZO_API float zo_cosf(float x);
ZO_API float zo_atanf(float x);
ZO_API float zo_atan2f(float y, float x);
ZO_API double zo_pow64(double x, double y);
ZO_API int zo_math_apply(int fn, const float *x, const float *y, float *out, size_t n); /* element-wise, fn as in zg_devmath_apply */

/* colorspaces.c — <Space>(T).to(target) for every float colour space (src/color.zig), fields in declaration order */
ZO_API void zo_color_to_f32(int from, const float in[4], int to, float out[4]);
ZO_API void zo_color_to_f64(int from, const double in[4], int to, double out[4]);

/* conv.c */
ZO_API int zo_gaussian_kernel(float sigma, float *taps, uint32_t capacity);
ZO_API int zo_conv_separable(const zo_image *src, const zo_image *dst, const float *kx, uint32_t nkx,
                             const float *ky, uint32_t nky, int border);
ZO_API int zo_gaussian_blur(const zo_image *src, const zo_image *dst, float sigma);
ZO_API int zo_convolve(const zo_image *src, const zo_image *dst, const float *kernel, uint32_t kh,
                       uint32_t kw, int border);
/* integral.c */
ZO_API int zo_integral_plane_f32(const float *src, size_t src_stride, float *sat, uint32_t rows, uint32_t cols);
ZO_API int zo_box_blur(const zo_image *src, const zo_image *dst, uint32_t radius);
ZO_API int zo_sharpen(const zo_image *src, const zo_image *dst, uint32_t radius);
ZO_API int zo_integral(const zo_image *src, float *planes);
ZO_API int zo_invert(const zo_image *img);

/* interp.c */
typedef struct zo_method { int32_t kind; float b, c; const float *lanczos_lut; } zo_method;
/* returns 1 and writes one pixel to out, or 0 for `null` */
ZO_API int zo_interpolate(const zo_image *img, float x, float y, const zo_method *m, int border, void *out);
ZO_API const float *zo_lanczos3_lut(void);
ZO_API int zo_resize(const zo_image *src, const zo_image *dst, const zo_method *m);
ZO_API int zo_letterbox(const zo_image *src, const zo_image *dst, const zo_method *m, uint32_t rect_out[4]);

/* transforms.c */
ZO_API void zo_project(int kind, const float *m, float x, float y, float *ox, float *oy);
ZO_API int zo_warp(const zo_image *src, const zo_image *dst, int kind, const float *m, const zo_method *method);
ZO_API int zo_rotate_bounds(uint32_t rows, uint32_t cols, float angle, float cos_a, float sin_a,
                            uint32_t *out_rows, uint32_t *out_cols);
ZO_API int zo_rotate_into(const zo_image *src, const zo_image *dst, float angle, float cos_a, float sin_a,
                          const zo_method *m, int border);
ZO_API int zo_extract(const zo_image *src, const zo_image *dst, const float rect[4], float angle,
                      float cos_a, float sin_a, const zo_method *m, int border);
ZO_API int zo_crop_dims(const float rect[4], uint32_t *rows, uint32_t *cols);
ZO_API int zo_crop(const zo_image *src, const zo_image *dst, const float rect[4]);
ZO_API int zo_flip_left_right(const zo_image *img);
ZO_API int zo_flip_top_bottom(const zo_image *img);
ZO_API int zo_insert(const zo_image *self, const zo_image *source, const float rect[4], float angle,
                     float cos_a, float sin_a, const zo_method *m, int blend_mode);
ZO_API int zo_copy(const zo_image *src, const zo_image *dst);
ZO_API int zo_fill(const zo_image *img, const void *pixel);
ZO_API int zo_set_border(const zo_image *img, const uint32_t rect[4], const void *pixel);

/* color.c */
ZO_API int zo_convert(const zo_image *src, int src_space, const zo_image *dst, int dst_space, const float *srgb_lut);
ZO_API void zo_srgb_to_linear_lut(float lut[256]);

/* homography (geometry/transforms.zig:242-263, exact 4-point solve in f64 then cast) */
ZO_API int zo_homography_from_4pts(const double from_xy[8], const double to_xy[8], float m_out[9]);

/* png.c — src/codecs/png.zig (the host I/O edge of the path). Status codes are the reference's error
 * set in the order of zo_png_error_name(); 0 = ok. */
typedef struct zo_png_header { /* png.zig:135-149 */
    uint32_t width, height;
    uint8_t bit_depth, color_type, compression_method, filter_method, interlace_method;
    uint8_t has_gamma, has_srgb, srgb_intent;
    float gamma;
} zo_png_header;
typedef struct zo_png_limits { /* png.zig:23-41; 0 disables a limit */
    size_t max_png_bytes, max_chunk_bytes, max_idat_bytes, max_chunks;
    uint32_t max_width, max_height;
    uint64_t max_pixels;
    size_t max_decompressed_bytes;
} zo_png_limits;
ZO_API const char *zo_png_error_name(int code);
ZO_API void zo_png_default_limits(zo_png_limits *l);
ZO_API uint32_t zo_png_crc(const uint8_t *p, size_t n);
ZO_API uint8_t zo_png_paeth(int a, int b, int c);
ZO_API int zo_png_info(const uint8_t *png, size_t len, const zo_png_limits *limits, zo_png_header *out);
ZO_API int zo_png_decode_chunks(const uint8_t *png, size_t len, const zo_png_limits *limits, zo_png_header *header_out, int *truncated_out,
                                int *palette_len, int *trns_len);
/* decode + toNativeImage: *pixels_out is malloc'd (rows * cols of the native pixel type ZO_U8 / ZO_RGB_U8 / ZO_RGBA_U8), free with zo_png_free */
ZO_API int zo_png_decode_native(const uint8_t *png, size_t len, const zo_png_limits *limits, zo_png_header *header_out, int *native_out,
                                uint8_t **pixels_out, int *truncated_out);
ZO_API void zo_png_free(void *p);
ZO_API int zo_png_scan_hash(const uint8_t *png, size_t len, const zo_png_limits *limits, uint64_t *hash_out, int *truncated_out);
ZO_API int zo_png_filter(const uint8_t *raw, uint32_t rows, size_t row_bytes, int bpp, int mode, uint8_t *filtered);
ZO_API int zo_png_encode_stored(const zo_image *img, int mode, uint8_t **out, size_t *out_len);

/* jpeg.c — the decoder of src/codecs/jpeg.zig (baseline + progressive). Status codes follow zo_jpeg_error_name(); 0 = ok. */
typedef struct zo_jpeg_header { /* jpeg.zig:61-74 */
    uint32_t width, height;
    uint8_t precision, num_components, progressive;
    int8_t subsampling; /* 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0, -1 = null (getInfo only) */
} zo_jpeg_header;
typedef struct zo_jpeg_limits { /* jpeg.zig:19-33; 0 disables a limit */
    size_t max_jpeg_bytes, max_marker_bytes;
    uint32_t max_width, max_height;
    uint64_t max_pixels;
    size_t max_blocks, max_scans;
} zo_jpeg_limits;
ZO_API const char *zo_jpeg_error_name(int code);
ZO_API void zo_jpeg_default_limits(zo_jpeg_limits *l);
ZO_API int zo_jpeg_info(const uint8_t *data, size_t len, const zo_jpeg_limits *limits, zo_jpeg_header *out);
ZO_API void zo_jpeg_idct8x8(int32_t block[64]);
/* decode (+ toNativeImage when pixels_out != NULL): *pixels_out is malloc'd rows * cols of ZO_U8 or ZO_RGB_U8; free with zo_jpeg_free */
ZO_API int zo_jpeg_decode_native(const uint8_t *data, size_t len, const zo_jpeg_limits *limits, zo_jpeg_header *header_out, int *native_out,
                                 uint8_t **pixels_out, int *scan_limit_reached_out);
ZO_API void zo_jpeg_free(void *p);
ZO_API int zo_jpeg_get_bits(const uint8_t *data, size_t len, const int *counts, int n, uint32_t *out);
ZO_API int zo_jpeg_coefficient_hash(const uint8_t *data, size_t len, const zo_jpeg_limits *limits, uint64_t *hash_out);
ZO_API void zo_jpeg_fdct8x8(const int32_t src[64], int32_t dst[64]);
/* jpeg.encode for Image(u8) / Image(Rgb(u8)): 0 ok, 1 InvalidImageDimensions, 2 ImageTooLarge, 3 other; *out from malloc (zo_jpeg_free) */
ZO_API int zo_jpeg_encode(const zo_image *img, int quality, int subsampling, int density_dpi, const uint8_t *comment, size_t comment_len,
                          uint8_t **out, size_t *out_len);

#endif
