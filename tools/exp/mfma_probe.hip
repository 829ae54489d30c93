// mfma_probe.hip — which lane / register / byte of v_mfma_i32_16x16x64_i8's operands is which matrix element on gfx950.
// Random int8 fragments go in, the 4 result registers per lane come out, and the host checks the layout conv_sep_mfma.hip assumes:
//   A: row = lane & 15, K slot (lane >> 4, byte e);  B: column = lane & 15, the same K slot;  D: column = lane & 15, row = 4 (lane >> 4) + reg.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_probe tools/exp/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i *a, const v4i *b, v4i *d) {
    const v4i c = {0, 0, 0, 0};
    d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
}
int main() {
    int8_t ha[64][16], hb[64][16];
    int hd[64][4];
    srand(3);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) { ha[l][e] = (int8_t)(rand() % 256 - 128); hb[l][e] = (int8_t)(rand() % 256 - 128); }
    v4i *da, *db, *dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    if (hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("probe: HIP error\n"); return 1; }
    // hypothesis: D[i][j] = sum over (g, e) of A[lane 16 g + i][e] * B[lane 16 g + j][e], found in lane j + 16 (i >> 2), register i & 3
    int bad = 0, bad_t = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        int s = 0;
        for (int g = 0; g < 4; ++g) for (int e = 0; e < 16; ++e) s += (int)ha[16 * g + i][e] * (int)hb[16 * g + j][e];
        if (hd[j + 16 * (i >> 2)][i & 3] != s) ++bad;
        if (hd[i + 16 * (j >> 2)][j & 3] != s) ++bad_t;
    }
    printf("probe: assumed layout %s (%d of 256 differ); transposed D layout %s (%d differ)\n", bad ? "WRONG" : "ok", bad, bad_t ? "wrong" : "OK", bad_t);
    return bad ? 2 : 0;
}
