// isef_bench.hip — isef.hip's two launches on an f32 plane: bit-for-bit against the recursion written out on the host (-ffp-contract=off on both
// sides), and timed. -DZG_ISEF_NOCHAIN / NOLOAD / NOSTORE compile one role's work out (the barriers stay). Includes the product source.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I zignal_amd/csrc -o tools/exp/isef_bench tools/exp/isef_bench.hip
// usage: isef_bench [rows cols]...   (default: a list of shapes, then 4096 x 4096 timed)
#include "../../zignal_amd/csrc/isef.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace zg { void set_error(const char *, ...) {} int hip_fail(hipError_t e, const char *w, const char *f, int l) { printf("HIP error %d %s %s:%d\n", (int)e, w, f, l); return 4; } }
#pragma clang fp contract(off)
static void isef1d(float *d, int n, int stride, float b, std::vector<float> &t) {
    const float a = 1.0f - b;
    t[0] = b * d[0];
    for (int i = 1; i < n; ++i) t[i] = b * d[(size_t)i * stride] + a * t[i - 1];
    d[(size_t)(n - 1) * stride] = t[n - 1];
    for (int i = n - 2; i >= 0; --i) d[(size_t)i * stride] = b * t[i] + a * d[(size_t)(i + 1) * stride];
}
int main(int argc, char **argv) {
    std::vector<std::pair<int, int>> shapes = {{16, 64}, {64, 64}, {65, 68}, {3, 4}, {1, 8}, {200, 132}, {130, 256}, {257, 1028}, {1080, 1920}};
    if (argc > 2) { shapes.clear(); for (int i = 1; i + 1 < argc; i += 2) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1])}); }
    const float b = getenv("ISEF_B") ? (float)atof(getenv("ISEF_B")) : 0.9f;
    int bad_shapes = 0;
    for (auto [rows, cols] : shapes) {
        const size_t n = (size_t)rows * cols;
        std::vector<float> h(n), want, got(n), t((size_t)std::max(rows, cols));
        unsigned s = 12345u + (unsigned)n;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 24); }
        want = h;
        for (int r = 0; r < rows; ++r) isef1d(&want[(size_t)r * cols], cols, 1, b, t);
        for (int c = 0; c < cols; ++c) isef1d(&want[c], rows, cols, b, t);
        float *dg, *ds, *dt;
        uint32_t *dk;
        (void)hipMalloc(&dk, zg::isef_check_bytes(rows, cols));
        (void)hipMalloc(&dg, n * 4 + 256); (void)hipMalloc(&ds, n * 4 + 256); (void)hipMalloc(&dt, n * 4 + 256);
        (void)hipMemcpy(dg, h.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemset(ds, 0xff, n * 4); (void)hipMemset(dt, 0xff, n * 4);
        { // each direction alone first
            std::vector<float> wr = h, wc = h, g2(n);
            for (int r = 0; r < rows; ++r) isef1d(&wr[(size_t)r * cols], cols, 1, b, t);
            for (int c = 0; c < cols; ++c) isef1d(&wc[c], rows, cols, b, t);
            const dim3 block(64 * (1 + zg::ISEF_NL + zg::ISEF_NS));
            hipLaunchKernelGGL(zg::k_isef<true>, dim3((rows + 63) / 64), block, 0, 0, (const void *)dg, dt, ds, rows, cols, b, zg::SpecCheck{});
            (void)hipMemcpy(g2.data(), ds, n * 4, hipMemcpyDeviceToHost);
            size_t br = 0; for (size_t i = 0; i < n; ++i) br += memcmp(&g2[i], &wr[i], 4) != 0;
            hipLaunchKernelGGL(zg::k_isef<false>, dim3((cols + 63) / 64), block, 0, 0, (const void *)dg, dt, ds, rows, cols, b, zg::SpecCheck{});
            (void)hipMemcpy(g2.data(), ds, n * 4, hipMemcpyDeviceToHost);
            size_t bc = 0, fc = 0; for (size_t i = 0; i < n; ++i) if (memcmp(&g2[i], &wc[i], 4) && !bc++) fc = i;
            if (n <= 16) {
                std::vector<float> tt(n);
                (void)hipMemcpy(tt.data(), dt, n * 4, hipMemcpyDeviceToHost);
                printf("   src:"); for (size_t i = 0; i < n; ++i) printf(" %g", h[i]);
                printf("\n   tmp (device, forward):"); for (size_t i = 0; i < n; ++i) printf(" %g", tt[i]);
                printf("\n   dst (device):"); for (size_t i = 0; i < n; ++i) printf(" %g", g2[i]);
                printf("\n   want:"); for (size_t i = 0; i < n; ++i) printf(" %g", wc[i]);
                printf("\n");
            }
            printf("   rows alone: %zu differ; columns alone: %zu differ", br, bc);
            if (bc) printf(" (first row %zu col %zu got %g want %g)", fc / cols, fc % cols, g2[fc], wc[fc]);
            printf("\n");
        }
        const int rc = zg::isef_2d(dg, false, ds, dt, dk, (uint32_t)rows, (uint32_t)cols, b, nullptr);
        (void)hipMemcpy(got.data(), ds, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < n; ++i) if (memcmp(&got[i], &want[i], 4) && !bad++) first = i;
        printf("%4d x %4d: rc %d, %zu of %zu values differ%s", rows, cols, rc, bad, n, bad ? "" : "\n");
        if (bad) { printf(" (first at row %zu col %zu: got %g want %g)\n", first / cols, first % cols, got[first], want[first]); ++bad_shapes; }
        (void)hipFree(dg); (void)hipFree(ds); (void)hipFree(dt); (void)hipFree(dk);
    }
    { // timing
        const int R = 4096;
        const size_t n = (size_t)R * R;
        float *dg, *ds, *dt;
        uint32_t *dk;
        (void)hipMalloc(&dk, zg::isef_check_bytes(R, R));
        (void)hipMalloc(&dg, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dt, n * 4);
        { // noise: the repair launch's rate is part of what is timed
            std::vector<float> h(n);
            unsigned sd = 99u;
            for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = (float)(sd >> 24); }
            (void)hipMemcpy(dg, h.data(), n * 4, hipMemcpyHostToDevice);
        }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 30; ++i) zg::isef_2d(dg, false, ds, dt, dk, R, R, b, nullptr);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) zg::isef_2d(dg, false, ds, dt, dk, R, R, b, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("isef_2d 4096 x 4096: %.1f us (rows + columns, forward + backward)\n", ms * 100);
    }
    return bad_shapes ? 2 : 0;
}
