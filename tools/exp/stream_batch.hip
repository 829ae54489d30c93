// stream_batch.hip — k_sep_stream in the batched geometry of BASELINE configs[4] (N x 1080p Rgba(u8), 5 taps, 2:1 resize fused), timed with
// HIP events over a ring of batches, with pieces of the kernel compiled out: -DZG_STREAM_NOLOAD (constants instead of loads),
// -DZG_STREAM_NOSTORE (results kept alive, nothing written), both; -DZG_STREAM_NOARITH (the loads and the stores alone). Includes the product kernel source.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DZG_STREAM_NOLOAD] [-DZG_STREAM_NOSTORE] -I zignal_amd/csrc -o tools/exp/stream_batch_<v> tools/exp/stream_batch.hip
// usage: stream_batch [frames=128] [rows=1080] [cols=1920] [down2=1] [strip_rows=44] [unit=1: the folded row pass of round 5]
#include "../../zignal_amd/csrc/conv_sep_stream.hip"
#include <cstdio>
#include <vector>

namespace zg { void set_error(const char *, ...) {} int hip_fail(hipError_t e, const char *w, const char *f, int l) { printf("HIP error %d %s %s:%d\n", (int)e, w, f, l); return 4; } }
using namespace zg;

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 128, rows = argc > 2 ? atoi(argv[2]) : 1080, cols = argc > 3 ? atoi(argv[3]) : 1920;
    const bool down2 = argc > 4 ? atoi(argv[4]) != 0 : true;
    const int strip_rows = argc > 5 ? atoi(argv[5]) : 44;
    const bool unit = argc > 6 ? atoi(argv[6]) != 0 : true;
    const size_t in_frame = (size_t)rows * cols * 4, out_frame = down2 ? in_frame / 4 : in_frame;
    const int ring = 3;
    uint8_t *src, *dst;
    if (hipMalloc(&src, in_frame * n * ring) != hipSuccess || hipMalloc(&dst, out_frame * n * ring) != hipSuccess) { printf("alloc failed\n"); return 1; }
    std::vector<uint8_t> h(in_frame);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 2654435761u >> 24);
    for (int i = 0; i < n * ring; ++i) (void)hipMemcpy(src + (size_t)i * in_frame, h.data(), in_frame, hipMemcpyHostToDevice);
    StreamArgs a{};
    a.src_pitch = (uint64_t)cols * 4; a.dst_pitch = down2 ? (uint64_t)cols * 2 : (uint64_t)cols * 4;
    a.src_frame = in_frame; a.dst_frame = out_frame;
    a.rows = rows; a.row_bytes = cols * 4; a.strips_x = (cols * 4 + 1023) / 1024; a.strip_rows = strip_rows;
    a.strips_y = (rows + strip_rows - 1) / strip_rows; a.border = ZG_BORDER_MIRROR;
    a.src_span = (uint32_t)((uint64_t)(rows - 1) * a.src_pitch + a.row_bytes);
    a.dst_span = (uint32_t)(down2 ? (uint64_t)(rows / 2 - 1) * a.dst_pitch + a.row_bytes / 2 : (uint64_t)(rows - 1) * a.dst_pitch + a.row_bytes);
    a.fast_ok = 1;
    const unsigned items = (unsigned)(a.strips_x * a.strips_y * n);
    TapsU8<5> k; const uint32_t t[5] = {1, 42, 170, 42, 1}; for (int i = 0; i < 5; ++i) k.k[i] = t[i];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&](int r) {
        a.src = src + (size_t)(r % ring) * in_frame * n; a.dst = dst + (size_t)(r % ring) * out_frame * n;
        if (down2 && unit) hipLaunchKernelGGL((k_sep_stream<4, 5, false, true, 1, true>), dim3(items), dim3(64), 0, 0, a, k, k);
        else if (down2) hipLaunchKernelGGL((k_sep_stream<4, 5, false, true, 1, false>), dim3(items), dim3(64), 0, 0, a, k, k);
        else if (unit) hipLaunchKernelGGL((k_sep_stream<4, 5, false, false, 1, true>), dim3(items), dim3(64), 0, 0, a, k, k);
        else hipLaunchKernelGGL((k_sep_stream<4, 5, false, false, 1, false>), dim3(items), dim3(64), 0, 0, a, k, k);
    };
    for (int r = 0; r < 900; ++r) launch(r); // ~0.3 s: the clocks have ramped
    (void)hipDeviceSynchronize();
    const int reps = 12;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch(r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const char *variant =
#if defined(ZG_STREAM_NOARITH)
        "no arithmetic";
#elif defined(ZG_STREAM_NOLOAD) && defined(ZG_STREAM_NOSTORE)
        "no loads, no stores";
#elif defined(ZG_STREAM_NOLOAD)
        "no loads";
#elif defined(ZG_STREAM_NOSTORE)
        "no stores";
#else
        "whole kernel";
#endif
    printf("%-20s %d x %d x %d down2=%d strip_rows=%d fold=%d: %u waves, %.1f us per launch\n", variant, n, rows, cols, (int)down2, strip_rows, (int)unit, items, ms * 1e3 / reps);
    return 0;
}
