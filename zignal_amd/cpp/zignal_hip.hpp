// zignal_hip.hpp — C++ host-side mirror of zignal's `Image(T)` over the C ABI of libzignal_hip.so.
//
// The reference is compiled code (Zig) and no Zig toolchain exists in this build image, so this header is the
// compiled-language face of the drop-in: same method names, argument meaning and error behaviour as reference
// src/image.zig (line numbers per method). Zig error unions become exceptions:
//   error.DimensionMismatch -> zignal::DimensionMismatch, error.InvalidSigma / InvalidScaleFactor /
//   InvalidDimensions -> zignal::InvalidArgument, error.OutOfMemory -> std::bad_alloc.
// Header only; link with -lzignal_hip.
//
// Two image classes share one set of methods (detail::Ops, written once):
//   Image<T>        pixels in host memory (std::vector-backed or borrowed). Every call is synchronous and crosses
//                   PCIe twice (zg_<op>_host), exactly like calling the reference's CPU method — the drop-in for code
//                   that touches pixels between calls.
//   DeviceImage<T>  pixels in HBM (zg_malloc), a stream to order work on. Calls go to the stream-taking entry points
//                   (zg_<op>) and return as soon as the work is enqueued; a chain such as the CLI's
//                   `pipeline [blur, resize]` (reference src/cli/pipeline.zig:153-179) keeps its intermediate images
//                   on the GPU and crosses PCIe once on the way in (upload / fromHost) and once on the way out
//                   (download / toHost). This is the one that runs at the measured kernel rates.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
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

// A HIP stream owned by the caller; DeviceImage work is ordered on one. The default-constructed handle is the default stream.
class Stream {
  public:
    Stream() = default;
    static Stream create() { Stream s; check(zg_stream_create(&s.s_)); s.owned_ = true; return s; }
    Stream(Stream &&o) noexcept : s_(o.s_), owned_(o.owned_) { o.s_ = nullptr; o.owned_ = false; }
    Stream &operator=(Stream &&o) noexcept { if (this != &o) { release(); s_ = o.s_; owned_ = o.owned_; o.s_ = nullptr; o.owned_ = false; } return *this; }
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;
    ~Stream() { release(); }
    zg_stream handle() const { return s_; }
    void synchronize() const { check(zg_stream_synchronize(s_)); }
  private:
    void release() { if (owned_ && s_) (void)zg_stream_destroy(s_); s_ = nullptr; owned_ = false; }
    zg_stream s_ = nullptr;
    bool owned_ = false;
};

template <typename T> class Image;
template <typename T> class DeviceImage;

namespace detail {

// Every hot-path method of Image(T), written once. `run(dev_fn, host_fn, args...)` calls zg_<op>(args..., stream) for a
// DeviceImage and zg_<op>_host(args...) for an Image; `Of<U>` is the same kind of image with pixel type U.
template <template <typename> class Img, typename T> class Ops {
    using Derived = Img<T>;
    template <typename U> using Of = Img<U>;
    const Derived &self() const { return static_cast<const Derived &>(*this); }
    template <class DevFn, class HostFn, class... A> void run(DevFn dev, HostFn host, A... a) const {
        if constexpr (Derived::on_device) check(dev(a..., self().stream()));
        else check(host(a...));
    }

  public:
    uint32_t rows = 0, cols = 0;
    size_t stride = 0;
    T *data = nullptr; // host pointer (Image) or device pointer (DeviceImage): never dereference the latter on the host

    template <class O> bool hasSameShape(const O &o) const { return rows == o.rows && cols == o.cols; }
    bool isContiguous() const { return cols == stride; }
    zg_image desc() const { return zg_image{(void *)data, stride, rows, cols, PixelTraits<T>::pixel}; }

    // ---- filters ----
    void convolveSeparable(const Derived &out, const std::vector<float> &kx, const std::vector<float> &ky, BorderMode border) const { // image.zig:935
        if (!hasSameShape(out)) throw DimensionMismatch(1, "convolveSeparable");
        const zg_image s = desc(), d = out.desc();
        run(zg_conv_separable, zg_conv_separable_host, &s, &d, kx.data(), (uint32_t)kx.size(), ky.data(), (uint32_t)ky.size(), (int)border);
    }
    void gaussianBlur(const Derived &out, float sigma) const {                           // image.zig:954
        if (!hasSameShape(out)) throw DimensionMismatch(1, "gaussianBlur");
        const zg_image s = desc(), d = out.desc();
        run(zg_gaussian_blur, zg_gaussian_blur_host, &s, &d, sigma);
    }
    template <size_t KH, size_t KW> void convolve(const Derived &out, const float (&kernel)[KH][KW], BorderMode border) const { // image.zig:917
        if (!hasSameShape(out)) throw DimensionMismatch(1, "convolve");
        const zg_image s = desc(), d = out.desc();
        run(zg_convolve, zg_convolve_host, &s, &d, &kernel[0][0], (uint32_t)KH, (uint32_t)KW, (int)border);
    }
    void boxBlur(const Derived &out, uint32_t radius) const {                            // image.zig:635
        if (!hasSameShape(out)) throw DimensionMismatch(1, "boxBlur");
        const zg_image s = desc(), d = out.desc();
        run(zg_box_blur, zg_box_blur_host, &s, &d, radius);
    }
    void medianBlur(const Derived &out, uint32_t radius) const {                          // image.zig:653
        if (!hasSameShape(out)) throw DimensionMismatch(1, "medianBlur");
        const zg_image s = desc(), d = out.desc();
        run(zg_order_statistic_blur, zg_order_statistic_blur_host, &s, &d, radius, 0, 0.5, (int)ZG_BORDER_MIRROR);
    }
    void equalize() const { const zg_image s = desc(); run(zg_equalize, zg_equalize_host, &s); } // image.zig:824 (in place)
    void sharpen(const Derived &out, uint32_t radius) const {                             // image.zig:785
        if (!hasSameShape(out)) throw DimensionMismatch(1, "sharpen");
        const zg_image s = desc(), d = out.desc();
        run(zg_sharpen, zg_sharpen_host, &s, &d, radius);
    }
    void invert() const { const zg_image s = desc(); run(zg_invert, zg_invert_host, &s); }  // image.zig:494 (in place)
    Of<uint8_t> sobel() const {                                                          // image.zig:1001 (out allocated here)
        auto out = Of<uint8_t>::like(self(), rows, cols);
        const zg_image s = desc(), d = out.desc();
        run(zg_sobel, zg_sobel_host, &s, &d);
        return out;
    }
    void canny(const Of<uint8_t> &out, float sigma, float low_threshold, float high_threshold) const {      // image.zig:1047
        if (rows != out.rows || cols != out.cols) throw DimensionMismatch(1, "canny");
        const zg_image s = desc(), d = out.desc();
        run(zg_canny, zg_canny_host, &s, &d, sigma, low_threshold, high_threshold);
    }
    struct ShenCastan { float smooth = 0.9f; uint32_t window_size = 7; float high_ratio = 0.99f, low_rel = 0.5f; bool hysteresis = true, use_nms = false; }; // ShenCastan.zig:9-32
    void shenCastan(const Of<uint8_t> &out, const ShenCastan &o = {}) const {            // image.zig:1015
        if (rows != out.rows || cols != out.cols) throw DimensionMismatch(1, "shenCastan");
        const zg_image s = desc(), d = out.desc();
        run(zg_shen_castan, zg_shen_castan_host, &s, &d, o.smooth, o.window_size, o.high_ratio, o.low_rel, o.hysteresis ? 1 : 0, o.use_nms ? 1 : 0);
    }
    void motionBlurLinear(const Derived &out, float angle, uint32_t distance) const {    // image.zig:1077 (.linear)
        if (!hasSameShape(out)) throw DimensionMismatch(1, "motionBlur");
        const zg_image s = desc(), d = out.desc();
        run(zg_motion_blur_linear, zg_motion_blur_linear_host, &s, &d, angle, std::cos(angle), std::sin(angle), distance);
    }
    void motionBlurRadial(const Derived &out, float center_x, float center_y, float strength, bool spin) const {  // (.radial_zoom / .radial_spin)
        if (!hasSameShape(out)) throw DimensionMismatch(1, "motionBlur");
        const zg_image s = desc(), d = out.desc();
        run(zg_motion_blur_radial, zg_motion_blur_radial_host, &s, &d, center_x, center_y, strength, spin ? 1 : 0);
    }
    // ---- resampling ----
    void resize(const Derived &out, Interpolation method) const {                        // image.zig:523
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        run(zg_resize, zg_resize_host, &s, &d, &m);
    }
    Derived scale(float factor, Interpolation method) const {                            // image.zig:530
        if (factor <= 0) throw InvalidArgument(2, "InvalidScaleFactor");
        const uint32_t nr = (uint32_t)std::round((float)rows * factor), nc = (uint32_t)std::round((float)cols * factor);
        if (nr == 0 || nc == 0) throw InvalidArgument(2, "InvalidDimensions");
        Derived out = Derived::like(self(), nr, nc);
        resize(out, method);
        return out;
    }
    Rectangle<uint32_t> letterbox(const Derived &out, Interpolation method) const {      // image.zig:546
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        uint32_t r[4];
        run(zg_letterbox, zg_letterbox_host, &s, &d, &m, r);
        return {r[0], r[1], r[2], r[3]};
    }
    void warp(const Derived &out, const ProjectiveTransform &t, Interpolation method) const { // image.zig:621
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        run(zg_warp, zg_warp_host, &s, &d, (int)ZG_TRANSFORM_PROJECTIVE, t.m, &m);
    }
    void rotateInto(const Derived &out, float angle, Interpolation method, BorderMode border) const { // image.zig:566
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        run(zg_rotate_into, zg_rotate_into_host, &s, &d, angle, std::cos(angle), std::sin(angle), &m, (int)border);
    }
    Derived rotate(float angle, Interpolation method, BorderMode border) const {         // image.zig:558
        uint32_t r, c;
        check(zg_rotate_bounds(rows, cols, angle, std::cos(angle), std::sin(angle), &r, &c));
        Derived out = Derived::like(self(), r, c);
        rotateInto(out, angle, method, border);
        return out;
    }
    void extract(const Derived &out, Rectangle<float> rect, float angle, Interpolation method, BorderMode border) const { // image.zig:593
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        const float r[4] = {rect.l, rect.t, rect.r, rect.b};
        run(zg_extract, zg_extract_host, &s, &d, r, angle, std::cos(angle), std::sin(angle), &m, (int)border);
    }
    Derived crop(Rectangle<float> rect) const {                                          // image.zig:582
        const float r[4] = {rect.l, rect.t, rect.r, rect.b};
        uint32_t nr, nc;
        check(zg_crop_dims(r, &nr, &nc));
        Derived out = Derived::like(self(), nr, nc);
        const zg_image s = desc(), d = out.desc();
        run(zg_crop, zg_crop_host, &s, &d, r);
        return out;
    }
    void fill(const T &value) const { const zg_image s = desc(); run(zg_fill, zg_fill_host, &s, (const void *)&value); } // image.zig:187
    void setBorder(Rectangle<uint32_t> rect, const T &value) const {                      // image.zig:200
        const zg_image s = desc();
        const uint32_t r[4] = {rect.l, rect.t, rect.r, rect.b};
        run(zg_set_border, zg_set_border_host, &s, r, (const void *)&value);
    }
    void flipLeftRight() const { const zg_image s = desc(); run(zg_flip_left_right, zg_flip_left_right_host, &s); }   // transforms.zig:28
    void flipTopBottom() const { const zg_image s = desc(); run(zg_flip_top_bottom, zg_flip_top_bottom_host, &s); }   // transforms.zig:36
    // ---- colour ----
    template <typename Target> void convertInto(const Of<Target> &out) const {          // image.zig:396
        const zg_image s = desc(), d = out.desc();
        run(zg_convert, zg_convert_host, &s, (int)PixelTraits<T>::space, &d, (int)PixelTraits<Target>::space, (const float *)nullptr);
    }
    template <typename Target> Of<Target> convert() const {                              // image.zig:418
        Of<Target> out = Of<Target>::like(self(), rows, cols);
        convertInto<Target>(out);
        return out;
    }
    // resize then convert in one call (zg_resize_convert): the pipeline steps [resize, convert] without the intermediate image
    template <typename Target> void resizeConvertInto(const Of<Target> &out, Interpolation method) const {
        const zg_image s = desc(), d = out.desc(); const zg_method m = method.c_method();
        run(zg_resize_convert, zg_resize_convert_host, &s, (int)PixelTraits<T>::space, &d, (int)PixelTraits<Target>::space, &m, (const float *)nullptr);
    }
    // ---- file output (src/codecs/jpeg.zig:307, src/codecs/png.zig:1400); the file lands in host memory either way ----
    std::vector<uint8_t> encodeJpeg(const zg_jpeg_encode_options *options = nullptr) const {
        uint8_t *mem = nullptr;
        size_t n = 0;
        const zg_image s = desc();
        run(zg_jpeg_encode, zg_jpeg_encode_host, &s, (int)PixelTraits<T>::space, options, &mem, &n);
        std::vector<uint8_t> out(mem, mem + n);
        zg_jpeg_free(mem);
        return out;
    }
    std::vector<uint8_t> encodePng(const zg_png_encode_options *options = nullptr) const {
        uint8_t *mem = nullptr;
        size_t n = 0;
        const zg_image s = desc();
        run(zg_png_encode, zg_png_encode_host, &s, (int)PixelTraits<T>::space, options, &mem, &n);
        std::vector<uint8_t> out(mem, mem + n);
        zg_png_free(mem);
        return out;
    }
};

} // namespace detail

// Image(T): rows, cols, stride (pixels), data — owning (init) or borrowed (initFromSlice / view), in host memory.
template <typename T> class Image : public detail::Ops<Image, T> {
    using Base = detail::Ops<Image, T>;

  public:
    static constexpr bool on_device = false;
    using Base::rows; using Base::cols; using Base::stride; using Base::data;

    Image() = default;
    Image(Image &&o) noexcept { *this = std::move(o); }
    Image &operator=(Image &&o) noexcept {
        rows = o.rows; cols = o.cols; stride = o.stride; data = o.data; owned_ = std::move(o.owned_); // a moved vector keeps its buffer
        o.rows = o.cols = 0; o.stride = 0; o.data = nullptr;
        return *this;
    }
    Image(const Image &o) { *this = o; }
    Image &operator=(const Image &o) { // an owning image is copied with its pixels, a borrowed one stays a borrow
        if (this == &o) return *this;
        rows = o.rows; cols = o.cols; stride = o.stride; owned_ = o.owned_;
        data = o.owned_.empty() ? o.data : owned_.data() + (o.data - o.owned_.data());
        return *this;
    }
    static Image init(uint32_t rows, uint32_t cols) {                                    // image.zig:124-134
        Image im;
        im.owned_.resize((size_t)rows * cols);
        im.rows = rows; im.cols = cols; im.stride = cols; im.data = im.owned_.data();
        return im;
    }
    template <class Proto> static Image like(const Proto &, uint32_t rows, uint32_t cols) { return init(rows, cols); }
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
    T &at(size_t r, size_t c) const { return data[r * stride + c]; }                     // image.zig:426-430
    zg_stream stream() const { return nullptr; }

    // ---- file input (image.zig:239-287: the format comes from the signature; src/codecs/png.zig, src/codecs/jpeg.zig) ----
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

  private:
    std::vector<T> owned_;
};

// Image(T) whose pixels live in HBM. Owning (init / fromHost / results of scale, crop, convert ...) or a non-owning view.
// Move-only. All methods of detail::Ops enqueue on stream() and return; call synchronize() (or download, which does)
// before looking at results from the host.
template <typename T> class DeviceImage : public detail::Ops<DeviceImage, T> {
    using Base = detail::Ops<DeviceImage, T>;

  public:
    static constexpr bool on_device = true;
    using Base::rows; using Base::cols; using Base::stride; using Base::data;

    DeviceImage() = default;
    DeviceImage(DeviceImage &&o) noexcept { *this = std::move(o); }
    DeviceImage &operator=(DeviceImage &&o) noexcept {
        if (this == &o) return *this;
        deinit();
        rows = o.rows; cols = o.cols; stride = o.stride; data = o.data; stream_ = o.stream_; owned_ = o.owned_;
        o.rows = o.cols = 0; o.stride = 0; o.data = nullptr; o.owned_ = nullptr;
        return *this;
    }
    DeviceImage(const DeviceImage &) = delete;
    DeviceImage &operator=(const DeviceImage &) = delete;
    ~DeviceImage() { deinit(); }

    static DeviceImage init(uint32_t rows, uint32_t cols, zg_stream stream = nullptr) {  // image.zig:124-134, in HBM
        DeviceImage im;
        check(zg_malloc(&im.owned_, (size_t)rows * cols * sizeof(T)));
        im.rows = rows; im.cols = cols; im.stride = cols; im.data = (T *)im.owned_; im.stream_ = stream;
        return im;
    }
    template <class Proto> static DeviceImage like(const Proto &proto, uint32_t rows, uint32_t cols) { return init(rows, cols, proto.stream()); }
    void deinit() {                                                                       // image.zig:173
        if (owned_) (void)zg_free(owned_); // hipFree waits for work that still uses the block
        owned_ = nullptr; data = nullptr; rows = cols = 0; stride = 0;
    }
    static DeviceImage fromHost(const Image<T> &host, zg_stream stream = nullptr) {
        DeviceImage im = init(host.rows, host.cols, stream);
        im.upload(host);
        return im;
    }
    void upload(const Image<T> &host) const {   // one trip across PCIe, strides honoured; complete on return
        const zg_image d = this->desc(), s = host.desc();
        check(zg_image_upload(&d, &s, stream_));
    }
    void download(const Image<T> &host) const { // waits for the stream's work on this image, then one trip back
        const zg_image d = host.desc(), s = this->desc();
        check(zg_image_download(&d, &s, stream_));
    }
    // convolveSeparable / gaussianBlur over several planes of one shape in one launch (zg_conv_separable_planes / zg_gaussian_blur_planes):
    // the reference's f32 route is per plane (Image(Rgba(f32)).convolveSeparable is a compile error, convolution.zig:431-435), so RGBA f32
    // data held as four Image(f32) planes goes through here. Asynchronous on planes[0]'s stream.
    static void convolveSeparablePlanes(const std::vector<const DeviceImage *> &planes, const std::vector<const DeviceImage *> &outs,
                                        const std::vector<float> &kx, const std::vector<float> &ky, BorderMode border) {
        if (planes.size() != outs.size()) throw DimensionMismatch(1, "convolveSeparablePlanes");
        if (planes.empty()) return;
        std::vector<zg_image> s, d;
        for (size_t i = 0; i < planes.size(); ++i) { s.push_back(planes[i]->desc()); d.push_back(outs[i]->desc()); }
        check(zg_conv_separable_planes(s.data(), d.data(), (uint32_t)s.size(), kx.data(), (uint32_t)kx.size(), ky.data(), (uint32_t)ky.size(), (int)border,
                                       planes[0]->stream()));
    }
    static void gaussianBlurPlanes(const std::vector<const DeviceImage *> &planes, const std::vector<const DeviceImage *> &outs, float sigma) {
        if (planes.size() != outs.size()) throw DimensionMismatch(1, "gaussianBlurPlanes");
        if (planes.empty()) return;
        std::vector<zg_image> s, d;
        for (size_t i = 0; i < planes.size(); ++i) { s.push_back(planes[i]->desc()); d.push_back(outs[i]->desc()); }
        check(zg_gaussian_blur_planes(s.data(), d.data(), (uint32_t)s.size(), sigma, planes[0]->stream()));
    }
    Image<T> toHost() const {
        Image<T> out = Image<T>::init(rows, cols);
        download(out);
        return out;
    }
    DeviceImage view(Rectangle<uint32_t> rect) const {                                   // image.zig:332-352 (non-owning)
        const uint32_t l = rect.l, t = rect.t, r = std::min(rect.r, cols), b = std::min(rect.b, rows);
        DeviceImage v;
        v.stream_ = stream_;
        if (l >= r || t >= b) return v;
        v.rows = b - t; v.cols = r - l; v.stride = stride; v.data = data + (size_t)t * stride + l;
        return v;
    }
    zg_stream stream() const { return stream_; }
    void setStream(zg_stream s) { stream_ = s; }
    void synchronize() const { check(zg_stream_synchronize(stream_)); }

    // decode a PNG / JPEG file straight into HBM (host: entropy layers; device: everything per pixel)
    static DeviceImage loadFromBytes(const uint8_t *bytes, size_t len, zg_stream stream = nullptr) {        // image.zig:265
        if (len >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
            zg_jpeg_header h;
            check(zg_jpeg_probe(bytes, len, nullptr, &h, nullptr));
            DeviceImage out = init(h.height, h.width, stream);
            const zg_image d = out.desc();
            check(zg_jpeg_decode(bytes, len, nullptr, &d, PixelTraits<T>::space, nullptr, stream));
            return out;
        }
        zg_png_header h;
        check(zg_png_probe(bytes, len, nullptr, &h, nullptr, nullptr));
        DeviceImage out = init(h.height, h.width, stream);
        const zg_image d = out.desc();
        check(zg_png_decode(bytes, len, nullptr, &d, PixelTraits<T>::space, nullptr, stream));
        return out;
    }

  private:
    void *owned_ = nullptr;
    zg_stream stream_ = nullptr;
};

// ImagePyramid(T) (src/image/pyramid.zig:11-170) resident on the device: level 0 is the source itself (not copied, pyramid.zig:54), level i
// the source blurred with sigma_i = blur_sigma * sqrt(scale_i^2 - 1) (only if > 0.5) and resized bilinearly to trunc(dim / scale_i),
// scale_i = pow(scale_factor, i); a level below 8 x 8 truncates the pyramid. `build` is ONE call of the C ABI: every level is enqueued
// without a host round trip (they fork over internal streams and join back into the source's stream).
template <typename T> struct ImagePyramid {
    std::vector<DeviceImage<T>> levels; // levels[0] is a view of the source
    float scale_factor = 0, blur_sigma = 0;
    static ImagePyramid build(const DeviceImage<T> &source, uint8_t n_levels, float scale_factor, float blur_sigma) {       // pyramid.zig:31-102
        ImagePyramid p;
        p.scale_factor = scale_factor;
        p.blur_sigma = blur_sigma;
        p.levels.push_back(source.view(Rectangle<uint32_t>{0, 0, source.cols, source.rows}));
        std::vector<zg_image> descs;
        std::vector<float> sigmas;
        for (uint32_t i = 1; i < n_levels; ++i) {
            uint32_t r = 0, c = 0;
            float sigma = 0;
            check(zg_pyramid_level(source.rows, source.cols, zg_pyramid_scale(scale_factor, i), blur_sigma, &r, &c, &sigma));
            if (r < 8 || c < 8) break;                                                                                       // pyramid.zig:63-73
            p.levels.push_back(DeviceImage<T>::init(r, c, source.stream()));
            descs.push_back(p.levels.back().desc());
            sigmas.push_back(sigma);
        }
        const zg_image s = source.desc();
        check(zg_pyramid_build(&s, descs.data(), sigmas.data(), (uint32_t)descs.size(), source.stream()));
        return p;
    }
    static ImagePyramid buildDefault(const DeviceImage<T> &source) { return build(source, 8, 1.2f, 1.6f); }                 // pyramid.zig:105-107
    size_t nLevels() const { return levels.size(); }
    float getScale(size_t level) const { return zg_pyramid_scale(scale_factor, (uint32_t)level); }                          // pyramid.zig:122-125
};

// The `pipeline` command (src/cli/pipeline.zig:153-179) over a batch of equally shaped frames resident on the device: a recipe is a
// list of steps, `run` sends every frame through them in order with one zg_batch_pipeline call (a launch per step over the whole
// batch where the library has a batched kernel, fused neighbours where it has a fused one; equal to the per-frame methods bit for bit).
struct Pipeline {
    std::vector<zg_step> steps;
    // zg_step carries no size field: a library built from another header must not be handed arrays of this header's struct (zignal_hip.h: zg_sizeof_step)
    Pipeline() {
        if (zg_sizeof_step() != sizeof(zg_step))
            throw Error(ZG_ERR_INVALID_ARGUMENT, "libzignal_hip: sizeof(zg_step) is " + std::to_string(zg_sizeof_step()) + " in the library, " + std::to_string(sizeof(zg_step)) +
                                                     " in this header: rebuild one of them");
    }
    Pipeline &gaussianBlur(float sigma) { zg_step s{}; s.kind = ZG_STEP_GAUSSIAN_BLUR; s.sigma = sigma; steps.push_back(s); return *this; }              // cli/blur.zig:113-120
    Pipeline &boxBlur(uint32_t radius) { zg_step s{}; s.kind = ZG_STEP_BOX_BLUR; s.radius = radius; steps.push_back(s); return *this; }                  // cli/blur.zig:109-112
    Pipeline &medianBlur(uint32_t radius = 1) { zg_step s{}; s.kind = ZG_STEP_MEDIAN_BLUR; s.radius = radius; steps.push_back(s); return *this; }          // cli/blur.zig:116-123
    // cli/blur.zig:124-146; the angle is in radians, cos_a / sin_a are the caller's own cosine and sine of it (as in Image::motionBlurLinear)
    Pipeline &motionBlurLinear(float angle, float cos_a, float sin_a, uint32_t distance) {
        zg_step s{}; s.kind = ZG_STEP_MOTION_BLUR; s.motion = ZG_MOTION_LINEAR; s.angle = angle; s.cos_a = cos_a; s.sin_a = sin_a; s.distance = distance; steps.push_back(s); return *this;
    }
    Pipeline &motionBlurRadial(float center_x = 0.5f, float center_y = 0.5f, float strength = 0.5f, bool spin = false) {                             // cli/blur.zig:147-170
        zg_step s{}; s.kind = ZG_STEP_MOTION_BLUR; s.motion = spin ? ZG_MOTION_RADIAL_SPIN : ZG_MOTION_RADIAL_ZOOM;
        s.center_x = center_x; s.center_y = center_y; s.strength = strength; steps.push_back(s); return *this;
    }
    // edges.apply (cli/edges.zig:126-135): convert(u8) -> detector -> convert back; the frames keep their type
    Pipeline &edgesSobel() { zg_step s{}; s.kind = ZG_STEP_EDGES; s.edges = ZG_EDGES_SOBEL; steps.push_back(s); return *this; }
    Pipeline &edgesCanny(float sigma = 1.0f, float low = 50.0f, float high = 100.0f) {                                                                // cli/edges.zig:98-104
        zg_step s{}; s.kind = ZG_STEP_EDGES; s.edges = ZG_EDGES_CANNY; s.sigma = sigma; s.low = low; s.high = high; steps.push_back(s); return *this;
    }
    Pipeline &edgesShenCastan(float smooth = 0.9f, uint32_t window_size = 7, float high_ratio = 0.99f, float low_rel = 0.5f, bool use_nms = false) { // cli/edges.zig:105-118
        zg_step s{}; s.kind = ZG_STEP_EDGES; s.edges = ZG_EDGES_SHEN_CASTAN; s.sigma = smooth; s.window = window_size; s.high = high_ratio; s.low = low_rel; s.use_nms = use_nms;
        steps.push_back(s); return *this;
    }
    Pipeline &resize(uint32_t rows, uint32_t cols, Interpolation method = Interpolation::bilinear()) {                                                // cli/resize.zig:77-99
        zg_step s{}; s.kind = ZG_STEP_RESIZE; s.out_rows = rows; s.out_cols = cols; s.method = method.c_method(); steps.push_back(s); return *this;
    }
    template <typename Target> Pipeline &convert() {                                                                                                  // image.zig:418
        zg_step s{}; s.kind = ZG_STEP_CONVERT; s.dst_pixel = PixelTraits<Target>::pixel; s.dst_space = PixelTraits<Target>::space; steps.push_back(s); return *this;
    }
    Pipeline &warp(int transform, const float *m, int n_coefficients, uint32_t rows, uint32_t cols, Interpolation method = Interpolation::bilinear()) {  // image.zig:621
        zg_step s{}; s.kind = ZG_STEP_WARP; s.transform = transform; s.out_rows = rows; s.out_cols = cols; s.method = method.c_method();
        for (int i = 0; i < n_coefficients && i < 9; ++i) s.m[i] = m[i];
        steps.push_back(s); return *this;
    }
    // shape and type of the frames after the steps
    void outShape(uint32_t rows, uint32_t cols, int pixel, int space, uint32_t &out_rows, uint32_t &out_cols, int &out_pixel) const {
        check(zg_batch_pipeline_shape(rows, cols, pixel, space, steps.data(), (uint32_t)steps.size(), &out_rows, &out_cols, &out_pixel, nullptr));
    }
    // n frames of Src, rows x cols each, back to back at `src` (device memory) -> frames of Dst at `dst` (device memory, n * outShape)
    template <typename Src> void run(const Src *src, uint32_t n_frames, uint32_t rows, uint32_t cols, void *dst, zg_stream stream = nullptr) const {
        check(zg_batch_pipeline(src, n_frames, rows, cols, PixelTraits<Src>::pixel, PixelTraits<Src>::space, steps.data(), (uint32_t)steps.size(), dst, stream));
    }
    // the same over every device of a zg_multi context: src / dst live on the context's root device, the call returns when the results are complete
    template <typename Src> void runMulti(zg_multi ctx, const Src *src_root, uint32_t n_frames, uint32_t rows, uint32_t cols, void *dst_root, float times_ms[3] = nullptr) const {
        check(zg_multi_batch_pipeline(ctx, src_root, n_frames, rows, cols, PixelTraits<Src>::pixel, PixelTraits<Src>::space, steps.data(), (uint32_t)steps.size(), dst_root,
                                      times_ms));
    }
};

} // namespace zignal
