// pk_sgpr.hip — issue rate of v_pk_fma_f32 with a vector-pair, a scalar-pair and a broadcast scalar (op_sel_hi) multiplier. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, insn)                                                                                                   \
    __global__ __launch_bounds__(1024) void k_##name(unsigned *sink, int iters, unsigned long long sk) {                  \
        const float t = threadIdx.x;                                                                                       \
        f2 a0 = {t, t + 1}, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;  \
        f2 b = a0 * 0.5f, c = a0 * 0.25f + 1.0f;                                                                           \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(REP16(insn) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sk), "v"(c)); \
        }                                                                                                                  \
        const f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                \
        if (r.x + r.y == 123.456f) *sink = 1;                                                                              \
    }
#define E8(pre, post) pre "0" post "0\n" pre "1" post "1\n" pre "2" post "2\n" pre "3" post "3\n" pre "4" post "4\n" pre "5" post "5\n" pre "6" post "6\n" pre "7" post "7\n"
BODY(pk_vvv, E8("v_pk_fma_f32 %", ", %8, %10, %"))
BODY(pk_svv, E8("v_pk_fma_f32 %", ", %9, %8, %"))
#define E8B(pre, post, tail) pre "0" post "0" tail "\n" pre "1" post "1" tail "\n" pre "2" post "2" tail "\n" pre "3" post "3" tail "\n" pre "4" post "4" tail "\n" pre "5" post "5" tail "\n" pre "6" post "6" tail "\n" pre "7" post "7" tail "\n"
BODY(pk_sbcast, E8B("v_pk_fma_f32 %", ", %9, %8, %", " op_sel_hi:[0,1,1]"))
BODY(pk_vbcast, E8B("v_pk_fma_f32 %", ", %10, %8, %", " op_sel_hi:[0,1,1]"))
template <typename K> static void run_wall(const char *name, K kern, unsigned *sink) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned long long sk = 0x3fc000003fc00000ull;
    hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, sink, 10, sk);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, sink, iters, sk);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 512.0 * 16 / 1024 * iters * 128.0;
    printf("WALL %-28s 512 blocks x 16 waves: %.3f ms -> %.2f cycles of SIMD time per wave64 instruction at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    unsigned *sink; (void)hipMalloc(&sink, 4);
    run_wall("pk_fma a, v2, v2, a", k_pk_vvv, sink); run_wall("pk_fma a, s2, v2, a", k_pk_svv, sink);
    run_wall("pk_fma a, s(bcast lo), v2, a", k_pk_sbcast, sink); run_wall("pk_fma a, v(bcast lo), v2, a", k_pk_vbcast, sink);
    return 0;
}
