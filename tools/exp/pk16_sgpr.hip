// pk16_sgpr.hip — issue rate of the packed-u16 / integer instructions of the u8 kernels with a scalar-register operand. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, insn)                                                                                                   \
    __global__ __launch_bounds__(1024) void k_##name(unsigned *sink, int iters, unsigned sk) {                            \
        unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        unsigned b = a0 ^ 0x01020304u, c = 0x002a00aau;                                                                    \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(REP16(insn) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sk), "v"(c) : "vcc"); \
        }                                                                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345678u) *sink = a0;                                              \
    }
#define E8(pre, post) pre "0" post "\n" pre "1" post "\n" pre "2" post "\n" pre "3" post "\n" pre "4" post "\n" pre "5" post "\n" pre "6" post "\n" pre "7" post "\n"
#define E8A(pre, mid, post) pre "0" mid "0" post "\n" pre "1" mid "1" post "\n" pre "2" mid "2" post "\n" pre "3" mid "3" post "\n" pre "4" mid "4" post "\n" pre "5" mid "5" post "\n" pre "6" mid "6" post "\n" pre "7" mid "7" post "\n"
BODY(pk_mad_vvv, E8A("v_pk_mad_u16 %", ", %8, %10, %", ""))
BODY(pk_mad_vsv, E8A("v_pk_mad_u16 %", ", %8, %9, %", ""))
BODY(pk_mul_vv, E8("v_pk_mul_lo_u16 %", ", %8, %10"))
BODY(pk_mul_vs, E8("v_pk_mul_lo_u16 %", ", %8, %9"))
BODY(pk_add_vs, E8A("v_pk_add_u16 %", ", %", ", %9"))
BODY(pk_add_vv, E8A("v_pk_add_u16 %", ", %", ", %8"))
BODY(mad_u16_vsv, E8A("v_mad_u32_u16 %", ", %8, %9, %", ""))
BODY(mad_u24_vsv, E8A("v_mad_u32_u24 %", ", %8, %9, %", ""))
BODY(dot4_vsv, E8A("v_dot4_u32_u8 %", ", %8, %9, %", ""))
BODY(dot4_vvv, E8A("v_dot4_u32_u8 %", ", %8, %10, %", ""))
BODY(perm_s, E8A("v_perm_b32 %", ", %", ", %8, %9"))
BODY(add_u32_s, E8A("v_add_u32 %", ", %9, %", ""))
BODY(add_u32_v, E8A("v_add_u32 %", ", %8, %", ""))
BODY(and_s, E8A("v_and_b32 %", ", %9, %", ""))
BODY(lshl_add_s, E8A("v_lshl_add_u32 %", ", %", ", 3, %9"))
template <typename K> static void run_wall(const char *name, K kern, unsigned *sink) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, sink, 10, 0x00030005u);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, sink, iters, 0x00030005u);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 512.0 * 16 / 1024 * iters * 128.0;
    printf("WALL %-26s 512 blocks x 16 waves: %.3f ms -> %.2f cycles of SIMD time per wave64 instruction at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    unsigned *sink; (void)hipMalloc(&sink, 4);
    run_wall("pk_mad_u16 a,v,v,a", k_pk_mad_vvv, sink); run_wall("pk_mad_u16 a,v,s,a", k_pk_mad_vsv, sink);
    run_wall("pk_mul_lo_u16 a,v,v", k_pk_mul_vv, sink); run_wall("pk_mul_lo_u16 a,v,s", k_pk_mul_vs, sink);
    run_wall("pk_add_u16 a,a,v", k_pk_add_vv, sink); run_wall("pk_add_u16 a,a,s", k_pk_add_vs, sink);
    run_wall("mad_u32_u16 a,v,s,a", k_mad_u16_vsv, sink); run_wall("mad_u32_u24 a,v,s,a", k_mad_u24_vsv, sink);
    run_wall("dot4_u32_u8 a,v,v,a", k_dot4_vvv, sink); run_wall("dot4_u32_u8 a,v,s,a", k_dot4_vsv, sink);
    run_wall("perm a,a,v,s", k_perm_s, sink);
    run_wall("add_u32 a,v,a", k_add_u32_v, sink); run_wall("add_u32 a,s,a", k_add_u32_s, sink); run_wall("and_b32 a,s,a", k_and_s, sink); run_wall("lshl_add_u32 a,a,3,s", k_lshl_add_s, sink);
    return 0;
}
