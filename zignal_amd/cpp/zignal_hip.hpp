// zignal_hip.hpp — C++ host-side mirror of zignal's `Image(T)` over the C ABI of libzignal_hip.so.
//
// The reference is compiled code (Zig) and no Zig toolchain exists in this build image, so this header is the
// compiled-language face of the drop-in: same method names, argument meaning and error behaviour as reference
// src/image.zig (line numbers per method). Zig error unions become exceptions:
//   error.DimensionMismatch -> zignal::DimensionMismatch, error.InvalidSigma / InvalidScaleFactor /
//   InvalidDimensions -> zignal::InvalidArgument, error.OutOfMemory -> std::bad_alloc.
// Header only; link with -lzignal_hip. Host pixels (std::vector-backed or borrowed), synchronous calls through
// the zg_<op>_host entry points, exactly like the reference's synchronous CPU methods.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/zignal_hip.h"

namespace zignal {

struct Error : std::runtime_error { int status; Error(int s, const std::string &m) : std::runtime_error(m), status(s) {} };
struct DimensionMismatch : Error { using Error::Error; };
struct InvalidArgument : Error { using Error::Error; };
// an error of a codec's error set (src/codecs/png.zig); name() is the Zig error name, e.g. "InvalidCrc"
struct CodecError : Error {
    using Error::Error;
    std::string name() const { const std::string m = what(); return m.substr(0, m.find(' ')); }
};

inline void check(int status) {
    if (status == ZG_OK) return;
    const std::string msg = zg_last_error();
    if (status == ZG_ERR_DIMENSION_MISMATCH) throw DimensionMismatch(status, msg);
    if (status == ZG_ERR_INVALID_ARGUMENT) throw InvalidArgument(status, msg);
    if (status == ZG_ERR_OUT_OF_MEMORY) throw std::bad_alloc();
    if (status == ZG_ERR_CODEC) throw CodecError(status, msg);
    throw Error(status, msg);
}

// pixel structs with the reference's memory layout (src/color.zig:286-290, :400)
template <typename T> struct Rgb { T r, g, b; };
template <typename T> struct Rgba { T r, g, b, a; };
template <typename T> struct Oklab { T l, a, b; };

template <typename T> struct PixelTraits;
template <> struct PixelTraits<uint8_t> { static constexpr int pixel = ZG_PIXEL_U8, space = ZG_CS_GRAY; };
template <> struct PixelTraits<float> { static constexpr int pixel = ZG_PIXEL_F32, space = ZG_CS_GRAY; };
template <> struct PixelTraits<Rgb<uint8_t>> { static constexpr int pixel = ZG_PIXEL_RGB_U8, space = ZG_CS_RGB; };
template <> struct PixelTraits<Rgba<uint8_t>> { static constexpr int pixel = ZG_PIXEL_RGBA_U8, space = ZG_CS_RGBA; };
template <> struct PixelTraits<Rgb<float>> { static constexpr int pixel = ZG_PIXEL_RGB_F32, space = ZG_CS_RGB; };
template <> struct PixelTraits<Rgba<float>> { static constexpr int pixel = ZG_PIXEL_RGBA_F32, space = ZG_CS_RGBA; };
template <> struct PixelTraits<Oklab<float>> { static constexpr int pixel = ZG_PIXEL_RGB_F32, space = ZG_CS_OKLAB; };

enum class BorderMode : int { zero = 0, replicate = 1, mirror = 2, wrap = 3 };          // border.zig:10-18
struct Interpolation {                                                                   // interpolation.zig:53-68
    int kind; float b = 0, c = 0;
    static Interpolation nearest() { return {ZG_INTERP_NEAREST}; }
    static Interpolation bilinear() { return {ZG_INTERP_BILINEAR}; }
    static Interpolation bicubic() { return {ZG_INTERP_BICUBIC}; }
    static Interpolation catmull_rom() { return {ZG_INTERP_CATMULL_ROM}; }
    static Interpolation mitchell(float b, float c) { return {ZG_INTERP_MITCHELL, b, c}; }
    static Interpolation lanczos() { return {ZG_INTERP_LANCZOS}; }
    zg_method c_method() const { return zg_method{kind, b, c, nullptr}; }
};
template <typename T> struct Rectangle { T l, t, r, b; T width() const { return l >= r ? T(0) : r - l; } T height() const { return t >= b ? T(0) : b - t; } };
struct ProjectiveTransform { float m[9]; };                                              // geometry/transforms.zig:197

// Image(T): rows, cols, stride (pixels), data — owning (init) or borrowed (initFromSlice / view).
template <typename T> class Image {
  public:
    uint32_t rows = 0, cols = 0;
    size_t stride = 0;
    T *data = nullptr;

    Image() = default;
    static Image init(uint32_t rows, uint32_t cols) {                                    // image.zig:124-134
        Image im;
        im.owned_.resize((size_t)rows * cols);
        im.rows = rows; im.cols = cols; im.stride = cols; im.data = im.owned_.data();
        return im;
    }
    static Image initFromSlice(uint32_t rows, uint32_t cols, T *data) {                  // image.zig:161-170
        Image im; im.rows = rows; im.cols = cols; im.stride = cols; im.data = data; return im;
    }
    Image view(Rectangle<uint32_t> rect) const {                                         // image.zig:332-352
        const uint32_t l = rect.l, t = rect.t, r = std::min(rect.r, cols), b = std::min(rect.b, rows);
        Image v;
        if (l >= r || t >= b) return v;
        v.rows = b - t; v.cols = r - l; v.stride = stride; v.data = data + (size_t)t * stride + l;
        return v;
    }
    bool hasSameShape(const Image &o) const { return rows == o.rows && cols == o.cols; }
    bool isContiguous() const { return cols == stride; }
    T &at(size_t r, size_t c) const { return data[r * stride + c]; }                     // image.zig:426-430

    // ---- filters ----
    void convolveSeparable(const Image &out, const std::vector<float> &kx, const std::vector<float> &ky, BorderMode border) const { // image.zig:935
        if (!hasSameShape(out)) throw DimensionMismatch(1, "convolveSeparable");
        const zg_image s = desc(), d = out.desc();
        check(zg_conv_separable_host(&s, &d, kx.data(), (uint32_t)kx.size(), ky.data(), (uint32_t)ky.size(), (int)border));
    }
    void gaussianBlur(const Image &out, float sigma) const {                             // image.zig:954
        if (!hasSameShape(out)) throw DimensionMismatch(1, "gaussianBlur");
        const zg_image s = desc(), d = out.desc();
        check(zg_gaussian_blur_host(&s, &d, sigma));
    }
    template <size_t KH, size_t KW> void convolve(const Image &out, const float (&kernel)[KH][KW], BorderMode border) const { // image.zig:917
        if (!hasSameShape(out)) throw DimensionMismatch(1, "convolve");
        const zg_image s = desc(), d = out.desc();
        check(zg_convolve_host(&s, &d, &kernel[0][0], (uint32_t)KH, (uint32_t)KW, (int)border));
    }
    void boxBlur(const Image &out, uint32_t radius) const {                              // image.zig:635
        if (!hasSameShape(out)) throw DimensionMismatch(1, "boxBlur");
        const zg_image s = desc(), d = out.desc();
        check(zg_box_blur_host(&s, &d, radius));
    }
    void medianBlur(const Image &out, uint32_t radius) const {                            // image.zig:653
        if (!hasSameShape(out)) throw DimensionMismatch(1, "medianBlur");
        const zg_image s = desc(), d = out.desc();
        check(zg_order_statistic_blur_host(&s, &d, radius, 0, 0.5, ZG_BORDER_MIRROR));
    }
    void equalize() const { const zg_image s = desc(); check(zg_equalize_host(&s)); }     // image.zig:824 (in place)
    void sharpen(const Image &out, uint32_t radius) const {                               // image.zig:785
        if (!hasSameShape(out)) throw DimensionMismatch(1, "sharpen");
        const zg_image s = desc(), d = out.desc();
        check(zg_sharpen_host(&s, &d, radius));
    }
    void invert() const { const zg_image s = desc(); check(zg_invert_host(&s)); }        // image.zig:494 (in place)
    Image<uint8_t> sobel() const {                                                       // image.zig:1001 (out allocated here)
        auto out = Image<uint8_t>::init(rows, cols);
        const zg_image s = desc(), d = out.desc();
        check(zg_sobel_host(&s, &d));
        return out;
    }
    void canny(const Image<uint8_t> &out, float sigma, float low_threshold, float high_threshold) const {   // image.zig:1047
        if (rows != out.rows || cols != out.cols) throw DimensionMismatch(1, "canny");
        const zg_image s = desc(), d = out.desc();
        check(zg_canny_host(&s, &d, sigma, low_threshold, high_threshold));
    }
    struct ShenCastan { float smooth = 0.9f; uint32_t window_size = 7; float high_ratio = 0.99f, low_rel = 0.5f; bool hysteresis = true, use_nms = false; }; // ShenCastan.zig:9-32
    void shenCastan(const Image<uint8_t> &out, const ShenCastan &o = {}) const {         // image.zig:1015
        if (rows != out.rows || cols != out.cols) throw DimensionMismatch(1, "shenCastan");
        const zg_image s = desc(), d = out.desc();
        check(zg_shen_castan_host(&s, &d, o.smooth, o.window_size, o.high_ratio, o.low_rel, o.hysteresis ? 1 : 0, o.use_nms ? 1 : 0));
    }
    void motionBlurLinear(const Image &out, float angle, uint32_t distance) const {     // image.zig:1077 (.linear)
        if (!hasSameShape(out)) throw DimensionMismatch(1, "motionBlur");
        const zg_image s = desc(), d = out.desc();
        check(zg_motion_blur_linear_host(&s, &d, angle, std::cos(angle), std::sin(angle), distance));
    }
    void motionBlurRadial(const Image &out, float center_x, float center_y, float strength, bool spin) const {   // (.radial_zoom / .radial_spin)
        if (!hasSameShape(out)) throw DimensionMismatch(1, "motionBlur");
        const zg_image s = desc(), d = out.desc();
        check(zg_motion_blur_radial_host(&s, &d, center_x, center_y, strength, spin ? 1 : 0));
    }
    // ---- resampling ----
    void resize(const Image &out, Interpolation method) const {                          // image.zig:523
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        check(zg_resize_host(&s, &d, &m));
    }
    Image scale(float factor, Interpolation method) const {                              // image.zig:530
        if (factor <= 0) throw InvalidArgument(2, "InvalidScaleFactor");
        const uint32_t nr = (uint32_t)std::round((float)rows * factor), nc = (uint32_t)std::round((float)cols * factor);
        if (nr == 0 || nc == 0) throw InvalidArgument(2, "InvalidDimensions");
        Image out = init(nr, nc);
        resize(out, method);
        return out;
    }
    Rectangle<uint32_t> letterbox(const Image &out, Interpolation method) const {        // image.zig:546
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        uint32_t r[4];
        check(zg_letterbox_host(&s, &d, &m, r));
        return {r[0], r[1], r[2], r[3]};
    }
    void warp(const Image &out, const ProjectiveTransform &t, Interpolation method) const { // image.zig:621
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        check(zg_warp_host(&s, &d, ZG_TRANSFORM_PROJECTIVE, t.m, &m));
    }
    void rotateInto(const Image &out, float angle, Interpolation method, BorderMode border) const { // image.zig:566
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        check(zg_rotate_into_host(&s, &d, angle, std::cos(angle), std::sin(angle), &m, (int)border));
    }
    Image rotate(float angle, Interpolation method, BorderMode border) const {           // image.zig:558
        uint32_t r, c;
        check(zg_rotate_bounds(rows, cols, angle, std::cos(angle), std::sin(angle), &r, &c));
        Image out = init(r, c);
        rotateInto(out, angle, method, border);
        return out;
    }
    void extract(const Image &out, Rectangle<float> rect, float angle, Interpolation method, BorderMode border) const { // image.zig:593
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        const float r[4] = {rect.l, rect.t, rect.r, rect.b};
        check(zg_extract_host(&s, &d, r, angle, std::cos(angle), std::sin(angle), &m, (int)border));
    }
    Image crop(Rectangle<float> rect) const {                                            // image.zig:582
        const float r[4] = {rect.l, rect.t, rect.r, rect.b};
        uint32_t nr, nc;
        check(zg_crop_dims(r, &nr, &nc));
        Image out = init(nr, nc);
        const zg_image s = desc(), d = out.desc();
        check(zg_crop_host(&s, &d, r));
        return out;
    }
    void fill(const T &value) const { const zg_image s = desc(); check(zg_fill_host(&s, &value)); } // image.zig:187
    void setBorder(Rectangle<uint32_t> rect, const T &value) const {                      // image.zig:200
        const zg_image s = desc();
        const uint32_t r[4] = {rect.l, rect.t, rect.r, rect.b};
        check(zg_set_border_host(&s, r, &value));
    }
    void flipLeftRight() const { const zg_image s = desc(); check(zg_flip_left_right_host(&s)); }   // transforms.zig:28
    void flipTopBottom() const { const zg_image s = desc(); check(zg_flip_top_bottom_host(&s)); }   // transforms.zig:36
    // ---- colour ----
    template <typename Target> void convertInto(const Image<Target> &out) const {       // image.zig:396
        const zg_image s = desc(), d = out.desc();
        check(zg_convert_host(&s, PixelTraits<T>::space, &d, PixelTraits<Target>::space, nullptr));
    }
    template <typename Target> Image<Target> convert() const {                           // image.zig:418
        Image<Target> out = Image<Target>::init(rows, cols);
        convertInto(out);
        return out;
    }

    // ---- file I/O (image.zig:239-287: the format comes from the signature; src/codecs/png.zig, src/codecs/jpeg.zig) ----
    static Image loadFromBytes(const uint8_t *bytes, size_t len) {                                          // image.zig:265
        if (len >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {                                             // jpeg.zig:2825
            zg_jpeg_header h;
            check(zg_jpeg_probe(bytes, len, nullptr, &h, nullptr));
            Image out = init(h.height, h.width);
            const zg_image d = out.desc();
            check(zg_jpeg_decode_host(bytes, len, nullptr, &d, PixelTraits<T>::space, nullptr));
            return out;
        }
        zg_png_header h;                                                                                    // png.zig:1151
        check(zg_png_probe(bytes, len, nullptr, &h, nullptr, nullptr));
        Image out = init(h.height, h.width);
        const zg_image d = out.desc();
        check(zg_png_decode_host(bytes, len, nullptr, &d, PixelTraits<T>::space, nullptr));
        return out;
    }
    std::vector<uint8_t> encodeJpeg(const zg_jpeg_encode_options *options = nullptr) const {                // jpeg.zig:307
        uint8_t *mem = nullptr;
        size_t n = 0;
        const zg_image s = desc();
        check(zg_jpeg_encode_host(&s, PixelTraits<T>::space, options, &mem, &n));
        std::vector<uint8_t> out(mem, mem + n);
        zg_jpeg_free(mem);
        return out;
    }
    std::vector<uint8_t> encodePng(const zg_png_encode_options *options = nullptr) const {                  // png.zig:1400
        uint8_t *mem = nullptr;
        size_t n = 0;
        const zg_image s = desc();
        check(zg_png_encode_host(&s, PixelTraits<T>::space, options, &mem, &n));
        std::vector<uint8_t> out(mem, mem + n);
        zg_png_free(mem);
        return out;
    }

    zg_image desc() const { return zg_image{(void *)data, stride, rows, cols, PixelTraits<T>::pixel}; }

  private:
    std::vector<T> owned_;
};

} // namespace zignal
