// Round 6: cycles per DEPENDENT pair (cross-lane move + v_add) in one wave, nothing else on the CU: v_permlane16_swap / v_permlane32_swap / v_mov_dpp (quad) /
// ds_swizzle (xor 16) / ds_bpermute. Result on an MI355X box (profiles/r06_box_blur.txt): 21.5 / 21.5 / 17.3 / 60.8 / 68.8 cycles — a wave alone issues a dependent vector
// instruction every ~8 cycles. build + run: hipcc --offload-arch=gfx950 -O3 -o lane_ops_latency lane_ops_latency.hip && ./lane_ops_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(unsigned *p, int n, unsigned long long *cyc) {
    unsigned a = p[threadIdx.x], b = a * 3 + 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0] + 1; b = r[1]; }
            if (KIND == 1) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0] + 1; b = r[1]; }
            if (KIND == 2) { a = (unsigned)__builtin_amdgcn_mov_dpp((int)a, 0xb1, 0xf, 0xf, false) + 1; }
            if (KIND == 3) { a = (unsigned)__builtin_amdgcn_ds_swizzle((int)a, 0x401f) + 1; } // xor 16
            if (KIND == 4) { a = (unsigned)__builtin_amdgcn_ds_bpermute((int)(threadIdx.x ^ 32) * 4, (int)a) + 1; }
            if (KIND == 5) { a = a * 5 + 1; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    p[threadIdx.x] = a + b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned *d; unsigned long long *c, h;
    hipMalloc(&d, 4096); hipMalloc(&c, 8);
    const char *names[] = {"permlane16_swap", "permlane32_swap", "mov_dpp quad", "ds_swizzle xor16", "ds_bpermute", "mad"};
    const int n = 1000;
#define RUN(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, d, n, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-18s %.1f cycles per dependent op (+1 add)\n", names[K], (double)h / (n * 16));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
