// fmac_sgpr.hip — does a scalar-register operand change the issue rate of v_fmac_f32 / v_fma_f32? (the column pass of conv_sep_bytes2.hip keeps its
// taps in SGPRs). Wall-clock form of valu_rate.hip: the whole chip full, hipEvent time / instructions per SIMD. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define I8(op, tail) op " %0, " tail ", %0x\n"
#define BODY(name, insn)                                                                                                   \
    __global__ __launch_bounds__(1024) void k_##name(unsigned *sink, int iters, float sk) {                               \
        float a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        float b = a0 * 0.5f, c = a0 * 0.25f + 1.0f;                                                                                               \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(REP16(insn) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sk), "v"(c)); \
        }                                                                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) *sink = 1;                                                  \
    }
#define EIGHT(fmt_a, fmt_b) fmt_a "0" fmt_b fmt_a "1" fmt_b fmt_a "2" fmt_b fmt_a "3" fmt_b fmt_a "4" fmt_b fmt_a "5" fmt_b fmt_a "6" fmt_b fmt_a "7" fmt_b
BODY(fmac_vv, EIGHT("v_fmac_f32 %", ", %8, %8\n"))
BODY(fmac_sv, EIGHT("v_fmac_f32 %", ", %9, %8\n"))
BODY(fma_svv, "v_fma_f32 %0, %9, %8, %0\nv_fma_f32 %1, %9, %8, %1\nv_fma_f32 %2, %9, %8, %2\nv_fma_f32 %3, %9, %8, %3\nv_fma_f32 %4, %9, %8, %4\nv_fma_f32 %5, %9, %8, %5\nv_fma_f32 %6, %9, %8, %6\nv_fma_f32 %7, %9, %8, %7\n")
BODY(fmac_self, "v_fmac_f32 %0, %0, %8\nv_fmac_f32 %1, %1, %8\nv_fmac_f32 %2, %2, %8\nv_fmac_f32 %3, %3, %8\nv_fmac_f32 %4, %4, %8\nv_fmac_f32 %5, %5, %8\nv_fmac_f32 %6, %6, %8\nv_fmac_f32 %7, %7, %8\n")
BODY(fmac_two_v, "v_fmac_f32 %0, %8, %10\nv_fmac_f32 %1, %8, %10\nv_fmac_f32 %2, %8, %10\nv_fmac_f32 %3, %8, %10\nv_fmac_f32 %4, %8, %10\nv_fmac_f32 %5, %8, %10\nv_fmac_f32 %6, %8, %10\nv_fmac_f32 %7, %8, %10\n")
BODY(fmac_lit, "v_fmac_f32 %0, 0x3fc00000, %8\nv_fmac_f32 %1, 0x3fc00000, %8\nv_fmac_f32 %2, 0x3fc00000, %8\nv_fmac_f32 %3, 0x3fc00000, %8\nv_fmac_f32 %4, 0x3fc00000, %8\nv_fmac_f32 %5, 0x3fc00000, %8\nv_fmac_f32 %6, 0x3fc00000, %8\nv_fmac_f32 %7, 0x3fc00000, %8\n")
BODY(fmac_inl, "v_fmac_f32 %0, 2.0, %8\nv_fmac_f32 %1, 2.0, %8\nv_fmac_f32 %2, 2.0, %8\nv_fmac_f32 %3, 2.0, %8\nv_fmac_f32 %4, 2.0, %8\nv_fmac_f32 %5, 2.0, %8\nv_fmac_f32 %6, 2.0, %8\nv_fmac_f32 %7, 2.0, %8\n")
BODY(mul_self, "v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8\n")
BODY(mul_two_v, "v_mul_f32 %0, %8, %10\nv_mul_f32 %1, %8, %10\nv_mul_f32 %2, %8, %10\nv_mul_f32 %3, %8, %10\nv_mul_f32 %4, %8, %10\nv_mul_f32 %5, %8, %10\nv_mul_f32 %6, %8, %10\nv_mul_f32 %7, %8, %10\n")
BODY(mul_sv, EIGHT("v_mul_f32 %", ", %9, %8\n"))
BODY(cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\nv_cvt_f32_u32_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\nv_cvt_f32_u32_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\nv_cvt_f32_u32_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_u32_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n")

template <typename K> static void run_wall(const char *name, K kern, unsigned *sink) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 1024}) { // 1 / 4 waves per SIMD
        hipLaunchKernelGGL(kern, dim3(512), dim3(threads), 0, 0, sink, 10, 1.5f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(512), dim3(threads), 0, 0, sink, iters, 1.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = 512.0 * (threads / 64) / 1024 * iters * 128.0;
        printf("WALL %-12s 512 blocks x %2d waves: %.3f ms -> %.2f cycles of SIMD time per wave64 instruction at 2.4 GHz\n", name, threads / 64, ms, ms * 1e6 / per_simd * 2.4);
    }
}
int main() {
    unsigned *sink; hipMalloc(&sink, 4);
    run_wall("fmac v,v,v", k_fmac_vv, sink); run_wall("fmac v,s,v", k_fmac_sv, sink); run_wall("fma v,s,v,v", k_fma_svv, sink); run_wall("mul v,s,v", k_mul_sv, sink);
    run_wall("cvt_u32 sdwa", k_cvt_sdwa, sink);
    run_wall("fmac a,a,b", k_fmac_self, sink); run_wall("fmac a,b,c", k_fmac_two_v, sink); run_wall("fmac a,lit,b", k_fmac_lit, sink); run_wall("fmac a,2.0,b", k_fmac_inl, sink);
    run_wall("mul a,a,b", k_mul_self, sink); run_wall("mul a,b,c", k_mul_two_v, sink);
    return 0;
}
