// valu_rate.hip — cycles per wave64 instruction for the integer / packed / cross-lane instructions the u8 kernels are made of,
// measured with s_memtime around a long dependent-free block, 1 and 2 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, insn)                                                                                                   \
    __global__ __launch_bounds__(1024) void k_##name(unsigned long long *out, unsigned *sink, int iters) {                \
        unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        unsigned b = a0 ^ 0x01020304u, c = 0x002a00aau;                                                                    \
        const unsigned long long t0 = clock64();                                                                          \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(REP16(insn) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"); \
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
BODY(dot2c_i32_i16, I8("v_dot2c_i32_i16", ", %8"))
BODY(dot4c_i32_i8, I8("v_dot4c_i32_i8", ", %8"))
BODY(fmac_f32, I8("v_fmac_f32", ", %8"))
BODY(mad_i32_i16, I8("v_mad_i32_i16", ", %8, %9"))
BODY(and_b32, I8("v_and_b32", ", %8"))
BODY(lshrrev_b32, I8("v_lshrrev_b32", ", 8"))
BODY(xor_b32, I8("v_xor_b32", ", %8"))
BODY(sqrt_f32, I8("v_sqrt_f32", ""))
BODY(rsq_f32, I8("v_rsq_f32", ""))

// packed f32: eight independent 64-bit chains
#define PK8(op) op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7\n"
#define PK8B(op) op " %0, %0, %0\n" op " %1, %1, %1\n" op " %2, %2, %2\n" op " %3, %3, %3\n" op " %4, %4, %4\n" op " %5, %5, %5\n" op " %6, %6, %6\n" op " %7, %7, %7\n"
#define BODY64(name, insn)                                                                                                 \
    __global__ __launch_bounds__(1024) void k_##name(unsigned long long *out, unsigned *sink, int iters) {                \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                              \
        f2 a0 = {1.0f, 1.0f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;                                 \
        const unsigned long long t0 = clock64();                                                                          \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(REP16(insn) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));     \
        }                                                                                                                  \
        const unsigned long long t1 = clock64();                                                                          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                    \
        if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y == 12345.0f) *sink = 1;                                  \
    }
BODY64(pk_fma_f32, PK8("v_pk_fma_f32"))
BODY64(pk_mul_f32, PK8B("v_pk_mul_f32"))
BODY64(pk_add_f32, PK8B("v_pk_add_f32"))
BODY64(fma_f64, PK8("v_fma_f64"))
BODY64(mul_f64, PK8B("v_mul_f64"))
BODY64(add_f64, PK8B("v_add_f64"))
BODY(rcp_f32, I8("v_rcp_f32", ""))
BODY(log_f32, I8("v_log_f32", ""))
BODY(exp_f32, I8("v_exp_f32", ""))


// what a real f32 kernel looks like to the issue logic: one dependent chain, literal constants, mixed opcodes
#define D8(op, tail) op " %0, %0" tail "\n" op " %0, %0" tail "\n" op " %0, %0" tail "\n" op " %0, %0" tail "\n" op " %0, %0" tail "\n" op " %0, %0" tail "\n" op " %0, %0" tail "\n" op " %0, %0" tail "\n"
BODY(add_f32_chain1, D8("v_add_f32", ", %8"))
#define D2x4(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %0, %0" tail "\n" op " %1, %1" tail "\n"
BODY(add_f32_chain2, D2x4("v_add_f32", ", %8"))
BODY(mul_f32_literal, "v_mul_f32 %0, 0x3f7fff00, %0\nv_mul_f32 %1, 0x3f7fff00, %1\nv_mul_f32 %2, 0x3f7fff00, %2\nv_mul_f32 %3, 0x3f7fff00, %3\nv_mul_f32 %4, 0x3f7fff00, %4\nv_mul_f32 %5, 0x3f7fff00, %5\nv_mul_f32 %6, 0x3f7fff00, %6\nv_mul_f32 %7, 0x3f7fff00, %7\n")
BODY(add_f32, I8("v_add_f32", ", %8"))
BODY(mul_f32, I8("v_mul_f32", ", %8"))
BODY(mix_f32, "v_mul_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_fmac_f32 %2, %2, %8\nv_sub_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_fmac_f32 %6, %6, %8\nv_sub_f32 %7, %7, %8\n")
BODY(mix_f32_dep, "v_mul_f32 %0, %1, %8\nv_add_f32 %1, %0, %8\nv_mul_f32 %2, %1, %0\nv_sub_f32 %3, %2, %1\nv_mul_f32 %4, %3, %2\nv_add_f32 %5, %4, %3\nv_mul_f32 %6, %5, %4\nv_sub_f32 %7, %6, %5\n")
BODY(cndmask, I8("v_cndmask_b32", ", %8, vcc"))
BODY(cmp_f32, "v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\nv_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n")
BODY(mov_b32, I8("v_mov_b32", ""))
BODY(mul_hi_u32, I8("v_mul_hi_u32", ", %8"))
BODY(bfi_b32, I8("v_bfi_b32", ", %8, %9"))
BODY(cvt_f32_ubyte0, I8("v_cvt_f32_ubyte0", ""))
BODY(mul_u32_u24_sdwa, I8("v_mul_u32_u24_sdwa", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"))

template <typename K> static void run(const char *name, K kern, unsigned long long *d, unsigned *sink) {
    const int iters = 200;
    for (int waves : {4, 8, 16}) { // per CU: one, two, four per SIMD
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * waves);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        const double per = sum / h.size() / (iters * 128.0);
        printf("%-22s %d waves/SIMD: %.2f cycles per instruction per wave -> %.2f cycles of SIMD time per instruction\n", name, waves / 4, per, per / (waves / 4));
    }
}
// wall-clock form: the whole chip full (8 waves per SIMD), hipEvent time / instructions per SIMD — no on-chip counter involved
template <typename K> static void run_wall(const char *name, K kern, unsigned long long *d, unsigned *sink) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) { // x 16 waves: 4 / 8 / 16 waves per SIMD on average (the last one in two rounds)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, d, sink, 10);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, d, sink, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)blocks * 16 / 1024 * iters * 128.0;
        printf("WALL %-22s %4d blocks x 16 waves: %.3f ms -> %.3f ns of SIMD time per wave64 instruction (= %.2f cycles at 2.4 GHz)\n", name, blocks, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
}
#define RUN(name) run(#name, k_##name, d, sink); run_wall(#name, k_##name, d, sink)
int main() {
    unsigned long long *d; hipMalloc(&d, 1024 * 16 * 8 * 2);
    unsigned *sink; hipMalloc(&sink, 4);
    RUN(add_u32); RUN(perm); RUN(pk_mad_u16); RUN(pk_mul_lo_u16); RUN(pk_add_u16); RUN(dot2_u32_u16); RUN(dot4_u32_u8); RUN(mad_u32_u16); RUN(mad_u32_u24);
    RUN(and_or); RUN(lshl_or); RUN(alignbyte); RUN(mov_dpp_wave_shr); RUN(mov_dpp_row_shr); RUN(fma_f32); RUN(mul_lo_u32); RUN(dot2c_i32_i16); RUN(dot4c_i32_i8); RUN(fmac_f32); RUN(mad_i32_i16); RUN(and_b32); RUN(lshrrev_b32); RUN(xor_b32); RUN(pk_fma_f32); RUN(pk_mul_f32); RUN(pk_add_f32); RUN(fma_f64); RUN(mul_f64); RUN(add_f64); RUN(rcp_f32); RUN(sqrt_f32); RUN(rsq_f32); RUN(log_f32); RUN(exp_f32); RUN(cvt_f32_ubyte0); RUN(add_f32_chain1); RUN(add_f32_chain2); RUN(mul_f32_literal); RUN(add_f32); RUN(mul_f32); RUN(mix_f32); RUN(mix_f32_dep); RUN(cndmask); RUN(cmp_f32); RUN(mov_b32); RUN(mul_hi_u32); RUN(bfi_b32);
    RUN(mul_u32_u24_sdwa);
    return 0;
}
