// Round 6: can the fused box blur's window mean use three operations instead of five?
//   five:  q0 = s * y; q1 = fma(fma(-a, q0, s), y, q0); q = fma(fma(-a, q1, s), y, q1)   (the IEEE division sequence, = s / a bit for bit: box_blur.hip)
//   three: q0 = s * y; q = fma(fma(-a, q0, s), y, q0)                                    (Markstein's correction step; y = RN(1 / a))
//   one:   q = fma(s, y, 2^-10) straight into v_cvt_pk_u8_f32: no correction step at all — only the BYTE has to match, and s / a = k + f / a is either an
//          exact tie or at least (for INTEGER s, which a window sum always is; counted over the integer-valued s only)
//           1 / (2 a) >= 1 / 98 away from one, far more than the error of s * y (<= 2^-15 for quotients up to 256)
// Checked over EVERY f32 bit pattern s and every area a = h * w, h, w in 1..7: (1) the quotients themselves, (2) the byte
// v_cvt_pk_u8_f32(q + 2^-10) that leaves the kernel, with and without sharpen's 2 * original - q for original in {0, 77, 255}.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o box_quot_check box_quot_check.hip && ./box_quot_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#pragma clang fp contract(off)

__device__ __forceinline__ float five(float s, float a, float y) {
    const float nd = -a, q0 = s * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(nd, q0, s), y, q0);
    return __builtin_fmaf(__builtin_fmaf(nd, q1, s), y, q1);
}
__device__ __forceinline__ float three(float s, float a, float y) {
    const float nd = -a, q0 = s * y;
    return __builtin_fmaf(__builtin_fmaf(nd, q0, s), y, q0);
}

__global__ void k_check(int area, unsigned long long *out) { // out: [0] quotient mismatches, [1] vs true division, [2] byte mismatches, [3] first bad bits
    const float a = (float)area;
    const float r0 = __builtin_amdgcn_rcpf(a);
    const float ynewton = __builtin_fmaf(__builtin_fmaf(-a, r0, 1.0f), r0, r0); // what box_blur.hip uses
    const float y = 1.0f / a;                                                   // RN(1 / a)
    unsigned long long bad_q = 0, bad_div = 0, bad_b = 0, bad_1 = 0, bad_y = ynewton != y;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float s = __builtin_bit_cast(float, (uint32_t)i);
        if (!(s == s) || __builtin_isinf(s)) continue;
        const float q5 = five(s, a, ynewton), q3 = three(s, a, y), qd = s / a;
        const bool fin = !__builtin_isinf(q5) && q5 == q5;
        if (fin && __builtin_bit_cast(uint32_t, q5) != __builtin_bit_cast(uint32_t, qd)) ++bad_div;
        if (fin && __builtin_bit_cast(uint32_t, q5) != __builtin_bit_cast(uint32_t, q3)) {
            ++bad_q;
            if (out[3] == 0) out[3] = i | (1ull << 40);
        }
        // the bytes (NaN / inf cannot reach the kernel: s is a finite sum, |s| < 2^34)
        if (!(__builtin_fabsf(s) < 0x1p34f)) continue;
        uint32_t b5 = __builtin_amdgcn_cvt_pk_u8_f32(q5 + 0x1p-10f, 0u, 0u), b3 = __builtin_amdgcn_cvt_pk_u8_f32(q3 + 0x1p-10f, 0u, 0u);
        bad_b += b5 != b3;
        uint32_t b1 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(s, y, 0x1p-10f), 0u, 0u);
        const bool integer = s == __builtin_truncf(s); // a window sum is a sum / difference of integer-valued floats: an integer, always
        bad_1 += integer && b5 != b1;
        for (float orig : {0.0f, 1.0f, 77.0f, 128.0f, 254.0f, 255.0f}) {
            const float t = 2 * orig;
            b5 = __builtin_amdgcn_cvt_pk_u8_f32((t - q5) + 0x1p-10f, 0u, 0u);
            b3 = __builtin_amdgcn_cvt_pk_u8_f32((t - q3) + 0x1p-10f, 0u, 0u);
            b1 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(-s, y, t + 0x1p-10f), 0u, 0u);
            bad_b += b5 != b3;
            bad_1 += integer && b5 != b1;
        }
    }
    atomicAdd(&out[0], bad_q);
    atomicAdd(&out[1], bad_div);
    atomicAdd(&out[2], bad_b);
    atomicAdd(&out[5], bad_1);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[4] = bad_y;
}

int main() {
    unsigned long long *out;
    hipMalloc(&out, 6 * sizeof(*out));
    bool seen[50] = {};
    int total_bad = 0, total_bad1 = 0;
    for (int h = 1; h <= 7; ++h)
        for (int w = 1; w <= 7; ++w) {
            const int area = h * w;
            if (seen[area]) continue;
            seen[area] = true;
            hipMemset(out, 0, 6 * sizeof(*out));
            hipLaunchKernelGGL(k_check, dim3(256 * 16), dim3(256), 0, 0, area, out);
            unsigned long long r[6];
            hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
            printf("area %2d: five != s/a: %llu   three != five (finite quotients): %llu   bytes differ, three: %llu   bytes differ, one fma: %llu   newton reciprocal != RN(1/a): %llu%s\n", area,
                   r[1], r[0], r[2], r[5], r[4], r[0] ? "   (first bad s bits below)" : "");
            if (r[0]) printf("         first bad s = 0x%08llx\n", r[3] & 0xffffffffull);
            total_bad += r[2] != 0;
            total_bad1 += r[5] != 0;
        }
    printf(total_bad ? "RESULT: the three-operation quotient changes output bytes for %d areas\n" : "RESULT: the three-operation quotient gives the same output byte for every finite f32 sum and every area (%d bad)\n", total_bad);
    printf(total_bad1 ? "RESULT: the single fma changes output bytes for %d areas\n" : "RESULT: the single fma gives the same output byte for every finite f32 sum with |s| < 2^34, every area and blur / sharpen (%d bad)\n", total_bad1);
    return 0;
}
