// sqrt_probe.hip — sobel_stream.hip's sobel_byte (raw v_sqrt_f32 + 2^-12, / 4, floor, saturating pack) against the reference's
// @trunc(@max(0, @min(255, @sqrt(m) / 4))) with a correctly rounded square root, for EVERY integer m the Sobel sums can produce
// (gx^2 + gy^2 <= 2 * 1020^2 = 2 080 800) and a margin beyond. Includes the product source.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I zignal_amd/csrc -o tools/exp/sqrt_probe tools/exp/sqrt_probe.hip
#include "../../zignal_amd/csrc/sobel_stream.hip"
#include <cmath>
#include <cstdio>
#include <vector>
namespace zg { void set_error(const char *, ...) {} int hip_fail(hipError_t e, const char *w, const char *f, int l) { printf("HIP error %d %s %s:%d\n", (int)e, w, f, l); return 4; } }
__global__ void k_probe(uint8_t *out, unsigned n) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint8_t)zg::sobel_byte((float)i, 0u, 0u);
}
int main() {
    const unsigned n = 1u << 22; // 4 194 304 > 2 080 800
    uint8_t *d;
    if (hipMalloc(&d, n) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_probe, dim3(n / 256), dim3(256), 0, 0, d, n);
    std::vector<uint8_t> h(n);
    if (hipMemcpy(h.data(), d, n, hipMemcpyDeviceToHost) != hipSuccess) { printf("sqrt_probe: HIP error\n"); return 1; }
    unsigned bad = 0, first = 0;
    for (unsigned m = 0; m < n; ++m) {
        const float s = std::sqrt((float)m) / 4.0f; // IEEE: correctly rounded square root, exact division by 4
        const float c = std::fmax(0.0f, std::fmin(255.0f, s));
        const uint8_t want = (uint8_t)std::trunc(c);
        if (h[m] != want && !bad++) first = m;
    }
    printf("sqrt_probe: %u of %u integers differ from trunc(min(255, sqrt(m) / 4))%s\n", bad, n, bad ? "" : " — none");
    if (bad) printf("  first at m = %u: got %u\n", first, h[first]);
    return bad ? 2 : 0;
}
