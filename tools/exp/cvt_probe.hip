// cvt_probe.hip — what v_cvt_pk_u8_f32 does with fractions, ties, negatives, values past 255, infinities and NaN on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/cvt_probe tools/exp/cvt_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *in, unsigned *out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned d;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(d) : "v"(in[i]), "v"(0xAABBCCDDu));
    out[i] = d;
}
int main() {
    const float v[] = {-1e9f, -5.0f, -0.6f, -0.5f, -0.4f, 0.0f, 0.4f, 0.5f, 0.6f, 1.4f, 1.5f, 1.6f, 2.5f, 3.5f, 127.5f, 254.4f, 254.5f, 254.6f, 255.0f, 255.4f, 255.5f, 255.6f, 256.0f, 300.0f, 1e9f,
                       INFINITY, -INFINITY, NAN};
    const int n = sizeof(v) / sizeof(v[0]);
    float *di; unsigned *dout; unsigned h[64];
    (void)hipMalloc(&di, sizeof v); (void)hipMalloc(&dout, sizeof h);
    (void)hipMemcpy(di, v, sizeof v, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, n);
    if (hipMemcpy(h, dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("cvt_probe: HIP error\n"); return 1; }
    for (int i = 0; i < n; ++i) printf("v_cvt_pk_u8_f32(%g) into byte 1 of 0xAABBCCDD -> 0x%08X (byte %u)\n", v[i], h[i], (h[i] >> 8) & 255);
    return 0;
}
