// Known-answer tests of the C++ host mirror, transcribed from the reference's own unit tests
// (paths relative to /root/reference/src). Needs a GPU: built by build(), run by tests/test_cpp_mirror.py.
#include <cstdio>
#include <cstdlib>

#include "../../zignal_amd/cpp/zignal_hip.hpp"

using namespace zignal;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

int main() {
    if (zg_init(0) != ZG_OK) { std::printf("no gfx950 device: %s\n", zg_last_error()); return 77; }

    { // image/tests/filters.zig:370-398 "convolve identity kernel"
        auto image = Image<uint8_t>::init(3, 3), result = Image<uint8_t>::init(3, 3);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) image.at(r, c) = (uint8_t)(r * 3 + c + 10);
        const float identity[3][3] = {{0, 0, 0}, {0, 1, 0}, {0, 0, 0}};
        image.convolve(result, identity, BorderMode::zero);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) EXPECT(image.at(r, c) == result.at(r, c));
    }
    { // image/tests/filters.zig:602-632 "stride bug in f32 separable convolution"
        auto base = Image<float>::init(5, 5), out = Image<float>::init(3, 3);
        for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) base.at(r, c) = (float)(r * 10 + c);
        auto view = base.view({1, 1, 4, 4});
        view.convolveSeparable(out, {1.0f}, {1.0f}, BorderMode::zero);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) EXPECT(view.at(r, c) == out.at(r, c));
    }
    { // image.zig:962, :970
        auto a = Image<float>::init(4, 4), b = Image<float>::init(4, 5);
        bool threw = false;
        try { a.gaussianBlur(b, 1.0f); } catch (const DimensionMismatch &) { threw = true; }
        EXPECT(threw);
        threw = false;
        try { a.gaussianBlur(a, -1.0f); } catch (const InvalidArgument &) { threw = true; }
        EXPECT(threw);
    }
    { // image/tests/interpolation.zig:422-449 via resize semantics; image/tests/resize.zig:140-161
        auto src = Image<uint8_t>::init(1, 1), out = Image<uint8_t>::init(10, 10);
        src.at(0, 0) = 128;
        auto rect = src.letterbox(out, Interpolation::nearest());
        EXPECT(rect.width() == 10 && rect.height() == 10);
        for (int r = 0; r < 10; ++r) for (int c = 0; c < 10; ++c) EXPECT(out.at(r, c) == 128);
    }
    { // image/tests/transforms.zig:231-278 "extract rotated rectangle basic and 90deg"
        auto image = Image<uint8_t>::init(5, 5), out = Image<uint8_t>::init(3, 3);
        for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) image.at(r, c) = (uint8_t)(r * 10 + c);
        image.extract(out, {1, 1, 3, 3}, 0.0f, Interpolation::nearest(), BorderMode::mirror);
        EXPECT(out.at(0, 0) == 11 && out.at(1, 1) == 22 && out.at(2, 2) == 33 && out.at(0, 2) == 13);
        image.extract(out, {1, 1, 3, 3}, 3.14159265358979f / 2, Interpolation::nearest(), BorderMode::mirror);
        EXPECT(out.at(0, 0) == 13 && out.at(0, 2) == 33 && out.at(2, 0) == 11 && out.at(1, 1) == 22);
    }
    { // image/tests/transforms.zig:427-456 flips; color.zig:1556-1562 grey
        uint8_t d[6] = {1, 2, 3, 4, 5, 6};
        auto im = Image<uint8_t>::initFromSlice(2, 3, d);
        im.flipLeftRight();
        EXPECT(d[0] == 3 && d[2] == 1 && d[3] == 6 && d[5] == 4);
        auto rgb = Image<Rgb<uint8_t>>::init(1, 2);
        rgb.at(0, 0) = {128, 128, 128};
        rgb.at(0, 1) = {255, 0, 0};
        auto gray = rgb.convert<uint8_t>();
        EXPECT(gray.at(0, 0) == 128 && gray.at(0, 1) == 54);
        auto lab = rgb.convert<Oklab<float>>();
        EXPECT(std::fabs(lab.at(0, 1).l - 0.628f) < 4e-3f);
    }
    { // image.zig:187-227: fill, then setBorder outside a rectangle (and with no overlap: everything)
        auto img = Image<uint8_t>::init(5, 6);
        img.fill(7);
        img.setBorder({1, 1, 4, 3}, 9);
        bool ok = true;
        for (uint32_t r = 0; r < 5; ++r)
            for (uint32_t c = 0; c < 6; ++c) ok = ok && img.at(r, c) == ((r >= 1 && r < 3 && c >= 1 && c < 4) ? 7 : 9);
        EXPECT(ok);
        img.setBorder({10, 10, 12, 12}, 3);
        EXPECT(img.at(0, 0) == 3 && img.at(2, 2) == 3 && img.at(4, 5) == 3);
    }
    { // image/tests/resize.zig:258-298 "scale image"
        auto img = Image<uint8_t>::init(100, 100);
        EXPECT(img.scale(0.5f, Interpolation::bilinear()).rows == 50);
        bool threw = false;
        try { img.scale(0.0f, Interpolation::bilinear()); } catch (const InvalidArgument &) { threw = true; }
        EXPECT(threw);
    }
    { // codecs/png.zig:2586-2642 round trip, :2073-2077 signature check
        auto img = Image<Rgb<uint8_t>>::init(4, 4);
        for (uint32_t r = 0; r < 4; ++r)
            for (uint32_t c = 0; c < 4; ++c) img.at(r, c) = {(uint8_t)(r * 60 + c), (uint8_t)(255 - c * 40), (uint8_t)(r * c * 17)};
        const std::vector<uint8_t> file = img.encodePng();
        EXPECT(file.size() > 8 && file[0] == 137 && file[1] == 'P');
        auto back = Image<Rgb<uint8_t>>::loadFromBytes(file.data(), file.size());
        bool same = back.rows == 4 && back.cols == 4;
        for (uint32_t r = 0; r < 4 && same; ++r)
            for (uint32_t c = 0; c < 4; ++c) same = same && back.at(r, c).r == img.at(r, c).r && back.at(r, c).g == img.at(r, c).g && back.at(r, c).b == img.at(r, c).b;
        EXPECT(same);
        auto rgba = Image<Rgba<uint8_t>>::loadFromBytes(file.data(), file.size()); // loadFromBytes(T) converts (png.zig:1160-1184)
        EXPECT(rgba.at(1, 2).a == 255 && rgba.at(1, 2).g == img.at(1, 2).g);
        const uint8_t junk[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        std::string name;
        try { Image<uint8_t>::loadFromBytes(junk, 8); } catch (const CodecError &e) { name = e.name(); }
        EXPECT(name == "InvalidPngSignature");
    }
    { // codecs/jpeg.zig:3055-3074: the reference's hand-built 8 x 8 progressive stream renders as a flat 143
        std::vector<uint8_t> j = {0xFF, 0xD8, 0xFF, 0xDB, 0x00, 0x43, 0x00};
        j.insert(j.end(), 64, 0x08);
        const uint8_t rest[] = {0xFF, 0xC2, 0x00, 0x0B, 0x08, 0x00, 0x08, 0x00, 0x08, 0x01, 0x01, 0x11, 0x00,
                                0xFF, 0xC4, 0x00, 0x14, 0x00, 0x01, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x02,
                                0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x02, 0x7F,
                                0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x21, 0xFF, 0x00,
                                0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x10, 0xFF, 0x00, 0xFF, 0xD9};
        j.insert(j.end(), rest, rest + sizeof rest);
        auto img = Image<uint8_t>::loadFromBytes(j.data(), j.size());
        bool flat = img.rows == 8 && img.cols == 8;
        for (uint32_t r = 0; r < 8 && flat; ++r)
            for (uint32_t c = 0; c < 8; ++c) flat = flat && img.at(r, c) == 143;
        EXPECT(flat);
        auto ramp = Image<Rgb<uint8_t>>::init(16, 16); // codecs/jpeg.zig:2860-2889: encode -> decode keeps PSNR above 40 dB
        for (uint32_t r = 0; r < 16; ++r)
            for (uint32_t c = 0; c < 16; ++c) ramp.at(r, c) = {(uint8_t)(c * 255 / 15), (uint8_t)(r * 255 / 15), (uint8_t)((r + c) * 255 / 30)};
        zg_jpeg_encode_options jo;
        zg_jpeg_default_encode_options(&jo);
        jo.quality = 85;
        const std::vector<uint8_t> jf = ramp.encodeJpeg(&jo);
        auto back = Image<Rgb<uint8_t>>::loadFromBytes(jf.data(), jf.size());
        double mse = 0;
        for (uint32_t r = 0; r < 16; ++r)
            for (uint32_t c = 0; c < 16; ++c) {
                const double dr = (double)ramp.at(r, c).r - back.at(r, c).r, dg = (double)ramp.at(r, c).g - back.at(r, c).g, db = (double)ramp.at(r, c).b - back.at(r, c).b;
                mse += dr * dr + dg * dg + db * db;
            }
        EXPECT(10.0 * std::log10(255.0 * 255.0 / (mse / (16 * 16 * 3))) > 40.0);
        std::string name;
        const uint8_t no_scan[] = {0xFF, 0xD8, 0xFF, 0xD9};
        try { Image<uint8_t>::loadFromBytes(no_scan, 4); } catch (const CodecError &e) { name = e.name(); }
        EXPECT(name == "NoScanData");
    }
    std::printf(failures ? "%d FAILED\n" : "cpp mirror ok\n", failures);
    return failures ? 1 : 0;
}
