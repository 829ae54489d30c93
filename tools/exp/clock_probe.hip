// clock_probe.hip — what the shader clock really is while a kernel runs: every wave reads s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) at its start and end. Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_valu(unsigned long long *out, int iters, float *sink) {
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) {
        a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f;
        a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f;
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a + c + d == 12345.678f) *sink = a;
}
// streaming copy with a little arithmetic per element (the u8 blur's mix: memory + VALU)
__global__ void k_mix(unsigned long long *out, const uint4 *src, uint4 *dst, size_t n, int valu) {
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        for (int k = 0; k < valu; ++k) { v.x = v.x * 1664525u + v.y; v.y = v.y * 22695477u + v.z; v.z = v.z * 1103515245u + v.w; v.w = v.w * 134775813u + v.x; }
        dst[i] = v;
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

static void report(const char *tag, const std::vector<unsigned long long> &h, int blocks, float ms) {
    std::vector<double> mhz;
    for (int i = 0; i < blocks; ++i) if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    printf("%-28s kernel %.3f ms  shader clock MHz: min %.0f median %.0f max %.0f  (wave life median %.1f us)\n", tag, ms, mhz.front(), mhz[mhz.size() / 2], mhz.back(),
           (double)h[2 * (blocks / 2) + 1] / 100.0);
}

int main() {
    const int blocks = 256 * 8;
    unsigned long long *d; hipMalloc(&d, blocks * 16);
    float *sink; hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, 20000, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost); report("pure VALU (fma chain)", h, blocks, ms);
    }
    const size_t n = (size_t)64 << 20; // 1 GiB in, 1 GiB out
    uint4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    for (int valu : {0, 4, 16, 64}) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, d, a, b, n, valu); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
        char tag[64]; snprintf(tag, sizeof tag, "copy + %d x4 int ops / 16 B", valu); report(tag, h, blocks, ms);
        printf("    -> %.2f TB/s\n", 2.0 * n * 16 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
