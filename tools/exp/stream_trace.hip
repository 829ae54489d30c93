// stream_trace.hip — per-wave timeline of k_sep_stream on a 4096 x 4096 Rgba(u8) frame: when every wave starts and ends
// (100 MHz clock), where it ran (XCC / CU), shader cycles. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   -DZG_STREAM_TRACE -I zignal_amd/csrc -o stream_trace tools/exp/stream_trace.hip    (includes the product kernel source)
#define ZG_STREAM_TRACE 1
#include "../../zignal_amd/csrc/conv_sep_stream.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>

namespace zg { void set_error(const char *, ...) {} int hip_fail(hipError_t e, const char *w, const char *f, int l) { printf("HIP error %d %s %s:%d\n", (int)e, w, f, l); return 4; } }
using namespace zg;

int main(int argc, char **argv) {
    const int R = 4096, rows_per = argc > 1 ? atoi(argv[1]) : 32;
    uint8_t *src, *dst; hipMalloc(&src, (size_t)R * R * 4 * 4); hipMalloc(&dst, (size_t)R * R * 4 * 4);
    hipMemset(src, 7, (size_t)R * R * 4 * 4);
    StreamArgs a{};
    a.src_pitch = a.dst_pitch = (uint64_t)R * 4; a.rows = R; a.row_bytes = R * 4; a.strips_x = R * 4 / 1024; a.strip_rows = rows_per;
    a.strips_y = (R + rows_per - 1) / rows_per; a.border = ZG_BORDER_MIRROR;
    const unsigned items = (unsigned)(a.strips_x * a.strips_y);
    hipMalloc(&a.trace, (size_t)items * 32);
    TapsU8<5> k; const uint32_t t[5] = {1, 42, 170, 42, 1}; for (int i = 0; i < 5; ++i) k.k[i] = t[i];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> h((size_t)items * 4);
    for (int rep = 0; rep < 12; ++rep) {
        a.src = src + (size_t)(rep & 3) * R * R * 4; a.dst = dst + (size_t)(rep & 3) * R * R * 4;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_sep_stream<4, 5, false, false, 1, false>), dim3(items), dim3(64), 0, 0, a, k, k);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), a.trace, (size_t)items * 32, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (unsigned i = 0; i < items; ++i) { t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 1]); }
        std::vector<double> st, du, mhz;
        std::map<unsigned long long, int> per_cu;
        for (unsigned i = 0; i < items; ++i) {
            st.push_back((h[4 * i] - t0) / 100.0); du.push_back((h[4 * i + 1] - h[4 * i]) / 100.0);
            mhz.push_back((double)h[4 * i + 2] / (double)(h[4 * i + 1] - h[4 * i]) * 100.0);
            const unsigned hw = (unsigned)h[4 * i + 3];
            per_cu[(h[4 * i + 3] >> 32 << 16) | ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4)]++; // xcc, cu_id [11:8], se_id [15:13]
        }
        // waves sharing a SIMD, and how a wave's life depends on that
        std::map<unsigned long long, int> per_simd;
        auto simd_key = [&](unsigned i) { const unsigned hw = (unsigned)h[4 * i + 3]; return (h[4 * i + 3] >> 32 << 20) | (((hw >> 13) & 0x7) << 12) | (((hw >> 8) & 0xf) << 4) | ((hw >> 4) & 0x3); };
        for (unsigned i = 0; i < items; ++i) per_simd[simd_key(i)]++;
        double life_by_share[8] = {0}; int n_by_share[8] = {0};
        for (unsigned i = 0; i < items; ++i) { const int sh = std::min(per_simd[simd_key(i)], 7); life_by_share[sh] += du[i]; n_by_share[sh]++; }
        double early = 0, late = 0; for (unsigned i = 0; i < items; ++i) (i < items / 2 ? early : late) += du[i];
        if (rep < 8) continue;
        printf("        mean life: first half of the grid %.1f us, second half %.1f us\n", early / (items / 2), late / (items - items / 2));
        if (rep == 11) {
            FILE *f = fopen("gpurun_out/r03/stream_trace.csv", "w");
            if (f) {
                fprintf(f, "block,start_us,life_us,xcc,se,cu,simd\n");
                for (unsigned i = 0; i < items; ++i) { const unsigned hw = (unsigned)h[4 * i + 3];
                    fprintf(f, "%u,%.2f,%.2f,%llu,%u,%u,%u\n", i, st[i], du[i], h[4 * i + 3] >> 32, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3); }
                fclose(f);
            }
        }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        int mn = 1 << 30, mx = 0; for (auto &kv : per_cu) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
        printf("rep %d: event %.1f us, first start -> last end %.1f us | wave start p0/p50/p90/p100 %.1f/%.1f/%.1f/%.1f us | wave life p0/p50/p100 %.1f/%.1f/%.1f us | clock %.0f MHz | %zu CUs, waves per CU %d..%d\n",
               rep, ms * 1e3, (t1 - t0) / 100.0, pct(st, 0), pct(st, .5), pct(st, .9), pct(st, 1), pct(du, 0), pct(du, .5), pct(du, 1), pct(mhz, .5), per_cu.size(), mn, mx);
    }
    return 0;
}
