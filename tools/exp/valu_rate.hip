// valu_rate.hip — cycles per wave64 instruction for the integer / packed / cross-lane instructions the u8 kernels are made of,
// measured with s_memtime around a long dependent-free block, 1 and 2 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, insn)                                                                                                   \
    __global__ __launch_bounds__(512) void k_##name(unsigned long long *out, unsigned *sink, int iters) {                \
        unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        unsigned b = a0 ^ 0x01020304u, c = 0x002a00aau;                                                                    \
        const unsigned long long t0 = clock64();                                                                          \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(REP16(insn) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                                                  \
        const unsigned long long t1 = clock64();                                                                          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                    \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345678u) *sink = a0;                                              \
    }
// eight independent chains per block of 8 instructions (x16 = 128 instructions per asm)
#define I8(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"
BODY(perm, I8("v_perm_b32", ", %8, %9"))
BODY(pk_mad_u16, I8("v_pk_mad_u16", ", %8, %9"))
BODY(pk_mul_lo_u16, I8("v_pk_mul_lo_u16", ", %8"))
BODY(pk_add_u16, I8("v_pk_add_u16", ", %8"))
BODY(dot2_u32_u16, I8("v_dot2_u32_u16", ", %8, %9"))
BODY(dot4_u32_u8, I8("v_dot4_u32_u8", ", %8, %9"))
BODY(mad_u32_u16, I8("v_mad_u32_u16", ", %8, %9"))
BODY(mad_u32_u24, I8("v_mad_u32_u24", ", %8, %9"))
BODY(add_u32, I8("v_add_u32", ", %8"))
BODY(and_or, I8("v_and_or_b32", ", %8, %9"))
BODY(lshl_or, I8("v_lshl_or_b32", ", %8, %9"))
BODY(alignbyte, I8("v_alignbyte_b32", ", %8, 1"))
BODY(mov_dpp_wave_shr, I8("v_mov_b32_dpp", " wave_shr:1 row_mask:0xf bank_mask:0xf"))
BODY(mov_dpp_row_shr, I8("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"))
BODY(fma_f32, I8("v_fma_f32", ", %8, %9"))
BODY(mul_lo_u32, I8("v_mul_lo_u32", ", %8"))
BODY(cvt_f32_ubyte0, I8("v_cvt_f32_ubyte0", ""))
BODY(mul_u32_u24_sdwa, I8("v_mul_u32_u24_sdwa", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"))

template <typename K> static void run(const char *name, K kern, unsigned long long *d, unsigned *sink) {
    const int iters = 200;
    for (int waves : {4, 8}) { // per CU: one, two per SIMD
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * waves);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        const double per = sum / h.size() / (iters * 128.0);
        printf("%-22s %d waves/SIMD: %.2f cycles per instruction per wave -> %.2f cycles of SIMD time per instruction\n", name, waves / 4, per, per / (waves / 4));
    }
}
#define RUN(name) run(#name, k_##name, d, sink)
int main() {
    unsigned long long *d; hipMalloc(&d, 256 * 8 * 8 * 2);
    unsigned *sink; hipMalloc(&sink, 4);
    RUN(add_u32); RUN(perm); RUN(pk_mad_u16); RUN(pk_mul_lo_u16); RUN(pk_add_u16); RUN(dot2_u32_u16); RUN(dot4_u32_u8); RUN(mad_u32_u16); RUN(mad_u32_u24);
    RUN(and_or); RUN(lshl_or); RUN(alignbyte); RUN(mov_dpp_wave_shr); RUN(mov_dpp_row_shr); RUN(fma_f32); RUN(mul_lo_u32); RUN(cvt_f32_ubyte0);
    RUN(mul_u32_u24_sdwa);
    return 0;
}
