// Device-resident Image(T) in a bare process (the configuration a Zig or C++ host has: the image's system ROCm runtime,
// no PyTorch anywhere): BASELINE config 5's `pipeline [blur, resize]` (reference src/cli/pipeline.zig:153-179) on
// DeviceImage<T> without leaving HBM between the two, checked bit for bit against the CPU oracle (oracle/liboracle.so is
// linked here as the checker — this file is test infrastructure), timed with device events; the host-pointer layer's banded
// pipeline against the whole-frame trip; and graph capture of multi-kernel ops that take scratch.
// Needs a GPU: built by tests/test_cpp_mirror.py, run there. Prints `key=value` lines the Python side reads.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <vector>

extern "C" {
#include "../../oracle/zo.h"
}
#include "../../zignal_amd/cpp/zignal_hip.hpp"

using namespace zignal;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
template <typename T> static zo_image zo_of(const Image<T> &im) { return zo_image{(void *)im.data, im.stride, im.rows, im.cols, PixelTraits<T>::pixel}; }
template <typename T> static bool same_bits(const Image<T> &a, const Image<T> &b) {
    if (a.rows != b.rows || a.cols != b.cols) return false;
    for (uint32_t r = 0; r < a.rows; ++r)
        if (std::memcmp(&a.at(r, 0), &b.at(r, 0), (size_t)a.cols * sizeof(T)) != 0) return false;
    return true;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Events {
    zg_event a = nullptr, b = nullptr;
    Events() { check(zg_event_create(&a)); check(zg_event_create(&b)); }
    ~Events() { (void)zg_event_destroy(a); (void)zg_event_destroy(b); }
    float ms() const { float t = 0; check(zg_event_synchronize(b)); check(zg_event_elapsed_ms(a, b, &t)); return t; }
};

int main(int argc, char **argv) {
    const bool virt = argc > 1 && std::strcmp(argv[1], "virtual") == 0; // quick sizes, but seven frames for the zg_multi section (two pieces per shard at world 3)
    const bool quick = virt || (argc > 1 && std::strcmp(argv[1], "quick") == 0);
    if (zg_init(0) != ZG_OK) { std::printf("no gfx950 device: %s\n", zg_last_error()); return 77; }
    const float sigma = 0.6f; // gaussianBlur(0.6): the 5 x 5 kernel BASELINE.json names
    Stream stream = Stream::create();
    uint32_t seed = 12345;

    { // config 5, one frame at a time as pipeline.zig runs it: blur -> resize on a device image, one upload, one download
        const uint32_t rows = 1080, cols = 1920, n_frames = quick ? 2 : 6;
        auto blurred = DeviceImage<Rgba<uint8_t>>::init(rows, cols, stream.handle());
        auto small = DeviceImage<Rgba<uint8_t>>::init(rows / 2, cols / 2, stream.handle());
        auto frame = DeviceImage<Rgba<uint8_t>>::init(rows, cols, stream.handle());
        for (uint32_t f = 0; f < n_frames; ++f) {
            auto host = Image<Rgba<uint8_t>>::init(rows, cols);
            for (size_t i = 0; i < (size_t)rows * cols; ++i) { const uint32_t v = lcg(seed); host.data[i] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 5)}; }
            frame.upload(host);
            frame.gaussianBlur(blurred, sigma);
            blurred.resize(small, Interpolation::bilinear());
            const Image<Rgba<uint8_t>> got = small.toHost();
            auto want_blur = Image<Rgba<uint8_t>>::init(rows, cols), want = Image<Rgba<uint8_t>>::init(rows / 2, cols / 2);
            const zo_image zs = zo_of(host), zb = zo_of(want_blur), zw = zo_of(want);
            const zo_method bil = {ZO_BILINEAR, 0, 0, nullptr};
            EXPECT(zo_gaussian_blur(&zs, &zb, sigma) == 0 && zo_resize(&zb, &zw, &bil) == 0);
            EXPECT(same_bits(got, want));
        }
        // device time of the resident chain (what bench.py's per-frame leg measures from Python)
        const int reps = 200;
        for (int i = 0; i < 20; ++i) { frame.gaussianBlur(blurred, sigma); blurred.resize(small, Interpolation::bilinear()); }
        Events ev;
        check(zg_event_record(ev.a, stream.handle()));
        for (int i = 0; i < reps; ++i) { frame.gaussianBlur(blurred, sigma); blurred.resize(small, Interpolation::bilinear()); }
        check(zg_event_record(ev.b, stream.handle()));
        std::printf("chain_1080p_rgba8_blur_resize_us=%.2f\n", ev.ms() * 1000.0 / reps);
    }

    { // BASELINE config 2 through the compiled-language mirror: 4096^2 Rgba(f32), resident, and the same through host pointers
        const uint32_t n = quick ? 1024 : 4096;
        auto host = Image<Rgba<float>>::init(n, n), want = Image<Rgba<float>>::init(n, n), out = Image<Rgba<float>>::init(n, n);
        for (size_t i = 0; i < (size_t)n * n; ++i) {
            const uint32_t v = lcg(seed), w = lcg(seed);
            host.data[i] = {(float)(v & 0xffff) / 65535.0f, (float)(v >> 16 & 0xff) / 255.0f, (float)(w & 0xffff) / 4096.0f - 3.0f, (float)(w >> 12 & 0xfff) / 4095.0f};
        }
        const zo_image zs = zo_of(host), zw = zo_of(want);
        EXPECT(zo_gaussian_blur(&zs, &zw, sigma) == 0);
        auto dsrc = DeviceImage<Rgba<float>>::fromHost(host, stream.handle());
        auto ddst = DeviceImage<Rgba<float>>::init(n, n, stream.handle());
        dsrc.gaussianBlur(ddst, sigma);
        ddst.download(out);
        EXPECT(same_bits(out, want));
        const int reps = 100;
        for (int i = 0; i < 30; ++i) dsrc.gaussianBlur(ddst, sigma);
        Events ev;
        check(zg_event_record(ev.a, stream.handle()));
        for (int i = 0; i < reps; ++i) dsrc.gaussianBlur(ddst, sigma);
        check(zg_event_record(ev.b, stream.handle()));
        std::printf("resident_blur_rgba_f32_%u_us=%.2f\n", n, ev.ms() * 1000.0 / reps);

        // host pointers: the banded full-duplex pipeline, then the plain upload -> kernel -> download trip
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) setenv("ZIGNAL_HIP_NO_BANDS", "1", 1);
            std::memset((void *)out.data, 0, (size_t)n * n * sizeof(Rgba<float>));
            host.gaussianBlur(out, sigma); // first call of a size pays the device allocations
            EXPECT(same_bits(out, want));
            double best = 1e30;
            for (int i = 0; i < 5; ++i) {
                const double t0 = now_ms();
                host.gaussianBlur(out, sigma);
                const double t = now_ms() - t0;
                if (t < best) best = t;
            }
            EXPECT(same_bits(out, want));
            std::printf(pass == 0 ? "host_blur_rgba_f32_%u_banded_ms=%.3f\n" : "host_blur_rgba_f32_%u_whole_ms=%.3f\n", n, best);
        }
        unsetenv("ZIGNAL_HIP_NO_BANDS");
        // views on both sides (strides differ from cols) through the banded path, and a convert (halo 0) with a type change
        auto big = Image<Rgba<float>>::init(n, n + 8);
        auto hv = big.view({3, 0, n + 3, n});
        for (uint32_t r = 0; r < n; ++r) std::memcpy(&hv.at(r, 0), &host.at(r, 0), (size_t)n * sizeof(Rgba<float>));
        auto obig = Image<Rgba<float>>::init(n, n + 5);
        auto ov = obig.view({5, 0, n + 5, n});
        hv.gaussianBlur(ov, sigma);
        EXPECT(same_bits(ov, want));
        auto lab = Image<Oklab<float>>::init(n, n), lab_want = Image<Oklab<float>>::init(n, n);
        host.convertInto<Oklab<float>>(lab);
        const zo_image zl = zo_of(lab_want);
        EXPECT(zo_convert(&zs, ZO_CS_RGBA, &zl, ZO_CS_OKLAB, nullptr) == 0);
        EXPECT(same_bits(lab, lab_want));
    }

    { // graph capture under the system runtime: multi-kernel ops that take scratch (17-tap two-pass blur, canny), replayed
      // while eager calls of the same sizes run in between; every replay must equal the eager result
        const uint32_t rows = 600, cols = 800;
        auto host = Image<Rgba<uint8_t>>::init(rows, cols);
        for (size_t i = 0; i < (size_t)rows * cols; ++i) { const uint32_t v = lcg(seed); host.data[i] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), 255}; }
        auto src = DeviceImage<Rgba<uint8_t>>::fromHost(host, stream.handle());
        auto eager_blur = DeviceImage<Rgba<uint8_t>>::init(rows, cols, stream.handle()), graph_blur = DeviceImage<Rgba<uint8_t>>::init(rows, cols, stream.handle());
        auto eager_edges = DeviceImage<uint8_t>::init(rows, cols, stream.handle()), graph_edges = DeviceImage<uint8_t>::init(rows, cols, stream.handle());
        const float big_sigma = 2.6f; // ceil(7.8) = 8 -> 17 taps: the packed two-pass path with a temp plane
        src.gaussianBlur(eager_blur, big_sigma);
        src.canny(eager_edges, 1.4f, 50.0f, 100.0f);
        const Image<Rgba<uint8_t>> want_blur = eager_blur.toHost();
        const Image<uint8_t> want_edges = eager_edges.toHost();
        { // the eager results themselves against the oracle
            auto ob = Image<Rgba<uint8_t>>::init(rows, cols);
            const zo_image zs = zo_of(host), zb = zo_of(ob);
            EXPECT(zo_gaussian_blur(&zs, &zb, big_sigma) == 0);
            EXPECT(same_bits(want_blur, ob));
        }
        check(zg_graph_begin_capture(stream.handle()));
        src.gaussianBlur(graph_blur, big_sigma);
        src.canny(graph_edges, 1.4f, 50.0f, 100.0f);
        zg_graph graph = nullptr;
        check(zg_graph_end_capture(stream.handle(), &graph));
        for (int replay = 0; replay < 4; ++replay) {
            const Rgba<uint8_t> junk = {1, 2, 3, 4};
            graph_blur.fill(junk);
            graph_edges.fill((uint8_t)7);
            check(zg_graph_launch(graph, stream.handle()));
            // eager calls of the same sizes between replays: they must not be handed the graph's scratch
            src.gaussianBlur(eager_blur, big_sigma);
            src.canny(eager_edges, 1.4f, 50.0f, 100.0f);
            EXPECT(same_bits(graph_blur.toHost(), want_blur));
            EXPECT(same_bits(graph_edges.toHost(), want_edges));
            EXPECT(same_bits(eager_blur.toHost(), want_blur));
            EXPECT(same_bits(eager_edges.toHost(), want_edges));
        }
        check(zg_graph_destroy(graph));
        check(zg_release_graph_scratch());
        src.gaussianBlur(eager_blur, big_sigma); // and the library is still in working order afterwards
        EXPECT(same_bits(eager_blur.toHost(), want_blur));
    }

    { // ImagePyramid.build as one device operation (zg_pyramid_build behind zignal::ImagePyramid), each level against the oracle
        const uint32_t prow = quick ? 160 : 540, pcol = quick ? 200 : 960;
        Image<uint8_t> g = Image<uint8_t>::init(prow, pcol);
        for (size_t i = 0; i < (size_t)prow * pcol; ++i) g.data[i] = (uint8_t)lcg(seed);
        DeviceImage<uint8_t> dg = DeviceImage<uint8_t>::fromHost(g);
        ImagePyramid<uint8_t> pyr = ImagePyramid<uint8_t>::buildDefault(dg);
        EXPECT(pyr.nLevels() == 8 && pyr.levels[0].data == dg.data);
        for (size_t i = 1; i < pyr.nLevels(); ++i) {
            uint32_t r = 0, c = 0; float sg = 0;
            check(zg_pyramid_level(prow, pcol, pyr.getScale(i), 1.6f, &r, &c, &sg));
            EXPECT(pyr.levels[i].rows == r && pyr.levels[i].cols == c);
            auto blurred = Image<uint8_t>::init(prow, pcol), want_level = Image<uint8_t>::init(r, c);
            const zo_image zs = zo_of(g), zb = zo_of(blurred), zw = zo_of(want_level);
            const zo_method zbil = {ZO_BILINEAR, 0, 0, nullptr};
            EXPECT(zo_gaussian_blur(&zs, &zb, sg) == 0 && zo_resize(&zb, &zw, &zbil) == 0);
            EXPECT(same_bits(pyr.levels[i].toHost(), want_level));
        }
    }

    { // the node's GPUs from this one thread (zg_multi): on a one-GPU box the context has one device; with ZIGNAL_HIP_MULTI_LOOPBACK the
      // root's shard makes its round trip through an RCCL communicator (ncclCommInitAll, grouped ncclSend / ncclRecv) all the same
        const uint32_t n = quick && !virt ? 3 : 7, rows = 270, cols = 480;
        std::vector<uint8_t> frames((size_t)n * rows * cols * 4);
        for (auto &b : frames) b = (uint8_t)lcg(seed);
        void *dsrc = nullptr, *dref = nullptr, *dout = nullptr;
        const size_t in_bytes = frames.size(), out_bytes = (size_t)n * (rows / 2) * (cols / 2) * 4;
        check(zg_malloc(&dsrc, in_bytes)); check(zg_malloc(&dref, out_bytes)); check(zg_malloc(&dout, out_bytes));
        check(zg_memcpy_h2d(dsrc, frames.data(), in_bytes, nullptr));
        const zg_method bil = {ZG_INTERP_BILINEAR, 0, 0, nullptr};
        check(zg_batch_blur_resize(dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dref, rows / 2, cols / 2, &bil, nullptr));
        std::vector<uint8_t> want(out_bytes), got(out_bytes);
        check(zg_memcpy_d2h(want.data(), dref, out_bytes, nullptr));
        { // one frame of the batch against the oracle, so `want` is not merely self-consistent
            Image<Rgba<uint8_t>> f0 = Image<Rgba<uint8_t>>::initFromSlice(rows, cols, (Rgba<uint8_t> *)frames.data());
            auto b0 = Image<Rgba<uint8_t>>::init(rows, cols), s0 = Image<Rgba<uint8_t>>::init(rows / 2, cols / 2);
            const zo_image zs = zo_of(f0), zb = zo_of(b0), zw = zo_of(s0);
            const zo_method zbil = {ZO_BILINEAR, 0, 0, nullptr};
            EXPECT(zo_gaussian_blur(&zs, &zb, sigma) == 0 && zo_resize(&zb, &zw, &zbil) == 0);
            EXPECT(std::memcmp(want.data(), s0.data, (size_t)(rows / 2) * (cols / 2) * 4) == 0);
        }
        { // the same recipe through the general batched pipeline (zg_batch_pipeline behind zignal::Pipeline), and a three-step one
            Pipeline recipe;
            recipe.gaussianBlur(sigma).resize(rows / 2, cols / 2);
            uint32_t orows = 0, ocols = 0; int opix = -1;
            recipe.outShape(rows, cols, ZG_PIXEL_RGBA_U8, ZG_CS_RGBA, orows, ocols, opix);
            EXPECT(orows == rows / 2 && ocols == cols / 2 && opix == ZG_PIXEL_RGBA_U8);
            std::vector<uint8_t> fill(out_bytes, 0x3C);
            check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
            recipe.run((const Rgba<uint8_t> *)dsrc, n, rows, cols, dout);
            check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
            EXPECT(got == want);
            Pipeline three;
            three.resize(rows / 2, cols / 2).gaussianBlur(sigma).convert<Rgb<float>>();
            three.outShape(rows, cols, ZG_PIXEL_RGBA_U8, ZG_CS_RGBA, orows, ocols, opix);
            EXPECT(opix == ZG_PIXEL_RGB_F32);
            void *dlab = nullptr;
            const size_t lab_bytes = (size_t)n * orows * ocols * 12;
            check(zg_malloc(&dlab, lab_bytes));
            three.run((const Rgba<uint8_t> *)dsrc, n, rows, cols, dlab);
            std::vector<float> lab(lab_bytes / 4);
            check(zg_memcpy_d2h(lab.data(), dlab, lab_bytes, nullptr));
            // frame 0 against the oracle: resize, blur, Rgba(u8) -> Rgb(f32) (c / 255)
            Image<Rgba<uint8_t>> f0 = Image<Rgba<uint8_t>>::initFromSlice(rows, cols, (Rgba<uint8_t> *)frames.data());
            auto r0 = Image<Rgba<uint8_t>>::init(rows / 2, cols / 2), b0 = Image<Rgba<uint8_t>>::init(rows / 2, cols / 2);
            const zo_image zs = zo_of(f0), zr = zo_of(r0), zb = zo_of(b0);
            const zo_method zbil = {ZO_BILINEAR, 0, 0, nullptr};
            EXPECT(zo_resize(&zs, &zr, &zbil) == 0 && zo_gaussian_blur(&zr, &zb, sigma) == 0);
            bool same = true;
            for (size_t i = 0; i < (size_t)orows * ocols && same; ++i) {
                const Rgba<uint8_t> px = b0.data[i];
                same = lab[3 * i] == (float)px.r / 255.0f && lab[3 * i + 1] == (float)px.g / 255.0f && lab[3 * i + 2] == (float)px.b / 255.0f;
            }
            EXPECT(same);
            check(zg_free(dlab));
        }
        // one device, then the same with the shard looped through both RCCL communicators in 1, 3 and 8 pieces (communicator set-up
        // costs seconds, so the loop-back runs only in the full mode)
        struct Mode { int loop; const char *chunks; };
        const Mode modes[] = {{0, "4"}, {1, "1"}, {1, "3"}, {1, "8"}};
        for (const Mode &mode : modes) {
            if (mode.loop && quick) continue;
            if (mode.loop) setenv("ZIGNAL_HIP_MULTI_LOOPBACK", "1", 1); else unsetenv("ZIGNAL_HIP_MULTI_LOOPBACK");
            setenv("ZIGNAL_HIP_MULTI_CHUNKS", mode.chunks, 1);
            zg_multi ctx = nullptr;
            check(zg_multi_create(nullptr, 1, &ctx));
            EXPECT(zg_multi_device_count(ctx) == 1);
            const uint8_t junk = 0xA5;
            std::vector<uint8_t> fill(out_bytes, junk);
            for (int call = 0; call < 2; ++call) { // a context is reusable
                check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
                check(zg_multi_wait_stream(ctx, nullptr));
                float t[3] = {0, 0, 0};
                check(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, call ? nullptr : t));
                check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                EXPECT(got == want);
                if (!call) std::printf(mode.loop ? "multi_1gpu_rccl_loopback_%s_pieces_ms=%.3f %.3f %.3f\n" : "multi_1gpu_%s_ms=%.3f %.3f %.3f\n", mode.chunks, t[0], t[1], t[2]);
            }
            { // any recipe, not just [blur, resize]: zg_multi_batch_pipeline behind Pipeline::runMulti, against the one-device zg_batch_pipeline
                Pipeline three;
                three.resize(rows / 2, cols / 2).gaussianBlur(sigma).convert<Rgb<float>>();
                uint32_t orows = 0, ocols = 0; int opix = -1;
                three.outShape(rows, cols, ZG_PIXEL_RGBA_U8, ZG_CS_RGBA, orows, ocols, opix);
                const size_t lab_bytes = (size_t)n * orows * ocols * 12;
                void *dwant = nullptr, *dgot = nullptr;
                check(zg_malloc(&dwant, lab_bytes)); check(zg_malloc(&dgot, lab_bytes));
                three.run((const Rgba<uint8_t> *)dsrc, n, rows, cols, dwant);
                std::vector<uint8_t> a(lab_bytes), b(lab_bytes, 0x77);
                check(zg_memcpy_d2h(a.data(), dwant, lab_bytes, nullptr));
                check(zg_memcpy_h2d(dgot, b.data(), lab_bytes, nullptr));
                float t[3] = {0, 0, 0};
                three.runMulti(ctx, (const Rgba<uint8_t> *)dsrc, n, rows, cols, dgot, t);
                check(zg_memcpy_d2h(b.data(), dgot, lab_bytes, nullptr));
                EXPECT(a == b);
                std::printf(mode.loop ? "multi_1gpu_rccl_loopback_%s_pieces_pipeline_ms=%.3f %.3f %.3f\n" : "multi_1gpu_%s_pipeline_ms=%.3f %.3f %.3f\n", mode.chunks, t[0], t[1], t[2]);
                // a recipe the shape function rejects fails before anything is enqueued and leaves the context usable
                Pipeline bad;
                bad.gaussianBlur(-1.0f);
                EXPECT(zg_multi_batch_pipeline(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, ZG_CS_RGBA, bad.steps.data(), 1, dgot, nullptr) != ZG_OK);
                three.runMulti(ctx, (const Rgba<uint8_t> *)dsrc, n, rows, cols, dgot);
                check(zg_memcpy_d2h(b.data(), dgot, lab_bytes, nullptr));
                EXPECT(a == b);
                check(zg_free(dwant)); check(zg_free(dgot));
            }
            { // frames produced on a NON-BLOCKING stream (zg_stream_create makes those): unnamed, the call synchronises the root device;
              // named with zg_multi_wait_stream, it waits for that stream. Either way it must not read the source before the copy has run.
                void *dsrc2 = nullptr, *big = nullptr;
                check(zg_malloc(&dsrc2, in_bytes)); check(zg_malloc(&big, (size_t)2048 * 2048 * 4 * 2));
                zg_stream st = nullptr;
                check(zg_stream_create(&st));
                const zg_image all = {dsrc, cols, n * rows, cols, ZG_PIXEL_RGBA_U8}, all2 = {dsrc2, cols, n * rows, cols, ZG_PIXEL_RGBA_U8};
                const zg_image b0 = {big, 2048, 2048, 2048, ZG_PIXEL_RGBA_U8}, b1 = {(char *)big + (size_t)2048 * 2048 * 4, 2048, 2048, 2048, ZG_PIXEL_RGBA_U8};
                for (int named = 0; named < 2; ++named) {
                    std::vector<uint8_t> zeros(in_bytes, 0);
                    check(zg_memcpy_h2d(dsrc2, zeros.data(), in_bytes, nullptr));
                    check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
                    for (int k = 0; k < 20; ++k) check(zg_gaussian_blur(k & 1 ? &b1 : &b0, k & 1 ? &b0 : &b1, 2.5f, st)); // keeps the stream busy for a while
                    check(zg_copy(&all, &all2, st));
                    if (named) check(zg_multi_wait_stream(ctx, st));
                    check(zg_multi_batch_blur_resize(ctx, dsrc2, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, nullptr));
                    check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                    EXPECT(got == want);
                }
                check(zg_stream_synchronize(st));
                check(zg_stream_destroy(st));
                check(zg_free(dsrc2)); check(zg_free(big));
            }
            // argument errors leave the context usable
            EXPECT(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, 99, sigma, dout, rows / 2, cols / 2, &bil, nullptr) == ZG_ERR_INVALID_ARGUMENT);
            check(zg_multi_batch_blur_resize(ctx, dsrc, 1, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, nullptr));
            check(zg_multi_destroy(ctx));
        }
        unsetenv("ZIGNAL_HIP_MULTI_CHUNKS");
        // every visible device (the world > 1 branches: shard ownership, staging, both communicators): needs a box with >= 2 GPUs
        if (zg_device_count() >= 2 && !quick) {
            for (const char *chunks : {"1", "4"}) {
                setenv("ZIGNAL_HIP_MULTI_CHUNKS", chunks, 1);
                unsetenv("ZIGNAL_HIP_MULTI_LOOPBACK");
                zg_multi ctx = nullptr;
                check(zg_multi_create(nullptr, 0, &ctx));
                const int world = zg_multi_device_count(ctx);
                EXPECT(world == zg_device_count());
                std::vector<uint8_t> fill(out_bytes, 0x5A);
                check(zg_set_device(0));
                check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
                float t[3] = {0, 0, 0};
                check(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, t));
                check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                EXPECT(got == want);
                std::printf("multi_%dgpu_%s_pieces_ms=%.3f %.3f %.3f\n", world, chunks, t[0], t[1], t[2]);
                { // the general recipe over every device
                    Pipeline recipe;
                    recipe.gaussianBlur(sigma).resize(rows / 2, cols / 2);
                    check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
                    recipe.runMulti(ctx, (const Rgba<uint8_t> *)dsrc, n, rows, cols, dout, t);
                    check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                    EXPECT(got == want);
                    std::printf("multi_%dgpu_%s_pieces_pipeline_ms=%.3f %.3f %.3f\n", world, chunks, t[0], t[1], t[2]);
                }
                check(zg_multi_destroy(ctx));
            }
            unsetenv("ZIGNAL_HIP_MULTI_CHUNKS");
        } else {
            std::printf("multi_world_gt_1=skipped (%d device%s visible)\n", zg_device_count(), zg_device_count() == 1 ? "" : "s");
        }
        unsetenv("ZIGNAL_HIP_MULTI_LOOPBACK");
        // Virtual worlds: with ZIGNAL_HIP_MULTI_VIRTUAL the same device may be listed N times, and ZIGNAL_HIP_RCCL_LIBRARY binds
        // tests/c/rccl_double.cpp instead of librccl — every world > 1 line of zg_multi.cpp (owners' staging buffers and offsets, the grouped send / receive
        // pairs of both communicators, per-piece events, piece arithmetic) then runs on this one GPU, against the one-device result; then a send fails
        // half-way through a batch: the call must fail, the context must refuse further work, and a fresh context must be fine.
        if (getenv("ZIGNAL_HIP_MULTI_VIRTUAL") && getenv("ZIGNAL_HIP_RCCL_LIBRARY")) {
            typedef void (*stats_fn)(uint64_t *);
            typedef void (*reset_fn)();
            void *dbl = dlopen(getenv("ZIGNAL_HIP_RCCL_LIBRARY"), RTLD_NOW | RTLD_LOCAL);
            EXPECT(dbl != nullptr);
            stats_fn stats = dbl ? (stats_fn)dlsym(dbl, "rccl_double_stats") : nullptr;
            reset_fn reset = dbl ? (reset_fn)dlsym(dbl, "rccl_double_reset") : nullptr;
            EXPECT(stats && reset);
            for (int world : {2, 3, 8}) {
                for (const char *chunks : {"1", "4", "8"}) {
                    setenv("ZIGNAL_HIP_MULTI_CHUNKS", chunks, 1);
                    std::vector<int> devs((size_t)world, 0);
                    zg_multi ctx = nullptr;
                    check(zg_multi_create(devs.data(), world, &ctx));
                    EXPECT(zg_multi_device_count(ctx) == world);
                    if (reset) reset();
                    std::vector<uint8_t> fill(out_bytes, 0x5A);
                    check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
                    float t[3] = {0, 0, 0};
                    check(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, t));
                    check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                    EXPECT(got == want);
                    uint64_t st[5] = {0, 0, 0, 0, 0};
                    if (stats) stats(st);
                    // every frame that is not the root's crosses once in each direction
                    const uint32_t root_frames = n / (uint32_t)world + (n % (uint32_t)world ? 1u : 0u);
                    EXPECT(st[0] == st[1] && st[4] == st[0] && st[0] >= 2);
                    EXPECT(st[2] == (uint64_t)(n - root_frames) * ((uint64_t)rows * cols * 4 + (uint64_t)(rows / 2) * (cols / 2) * 4));
                    Pipeline recipe;
                    recipe.gaussianBlur(sigma).resize(rows / 2, cols / 2);
                    check(zg_memcpy_h2d(dout, fill.data(), out_bytes, nullptr));
                    recipe.runMulti(ctx, (const Rgba<uint8_t> *)dsrc, n, rows, cols, dout, t);
                    check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                    EXPECT(got == want);
                    check(zg_multi_destroy(ctx));
                    std::printf("multi_virtual_world%d_%s_pieces=ok sends=%llu bytes=%llu\n", world, chunks, (unsigned long long)st[0], (unsigned long long)st[2]);
                }
            }
            { // a send of piece 1 fails: world 3, four pieces per shard -> sends 1, 2 are piece 0's scatter, 3, 4 its gather, 5 piece 1's scatter
                setenv("ZIGNAL_HIP_MULTI_CHUNKS", "4", 1);
                const int devs[3] = {0, 0, 0};
                zg_multi ctx = nullptr;
                check(zg_multi_create(devs, 3, &ctx));
                if (reset) reset();
                setenv("RCCL_DOUBLE_FAIL_SEND", n >= 7 ? "5" : "3", 1); // three frames: one piece per shard, so its gather fails instead
                EXPECT(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, nullptr) == ZG_ERR_HIP);
                EXPECT(std::strstr(zg_last_error(), "RCCL error") != nullptr);
                unsetenv("RCCL_DOUBLE_FAIL_SEND");
                // poisoned: no later call may trust the streams or the communicators
                EXPECT(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, nullptr) == ZG_ERR_INVALID_ARGUMENT);
                EXPECT(std::strstr(zg_last_error(), "failed half-way") != nullptr);
                check(zg_multi_destroy(ctx)); // drains what was enqueued
                check(zg_multi_create(devs, 3, &ctx));
                check(zg_multi_batch_blur_resize(ctx, dsrc, n, rows, cols, ZG_PIXEL_RGBA_U8, sigma, dout, rows / 2, cols / 2, &bil, nullptr));
                check(zg_memcpy_d2h(got.data(), dout, out_bytes, nullptr));
                EXPECT(got == want);
                check(zg_multi_destroy(ctx));
                std::printf("multi_virtual_failure_injection=ok\n");
            }
            unsetenv("ZIGNAL_HIP_MULTI_CHUNKS");
        }
        zg_multi bad = nullptr;
        EXPECT(zg_multi_create(nullptr, 99, &bad) == ZG_ERR_INVALID_ARGUMENT && bad == nullptr); // more devices than the box has
        check(zg_free(dsrc)); check(zg_free(dref)); check(zg_free(dout));
    }

    std::printf(failures ? "%d FAILED\n" : "device image ok\n", failures);
    return failures ? 1 : 0;
}
