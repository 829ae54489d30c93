// zg_bilinear_u8.h — the u8 plane bilinear resizer's per-pixel arithmetic (reference src/image/channel_ops.zig:144-190:
// s = (d + 0.5) * ratio - 0.5 in f32, f = trunc(frac * 256), mirror-resolved taps, two 8.8 lerps and a truncating shift),
// shared by k_resize_bilinear_rgba8 (resize_planes.hip) and the fused resize -> convert kernel (convert.hip).
#pragma once
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

__device__ inline void bilinear_taps(int d, float ratio, int n, int &i0, int &i1, int &f) {
    const float sf = ((float)d + 0.5f) * ratio - 0.5f;
    const float fl = floorf(sf);
    const int base = (int)fl;
    f = (int)truncf((sf - fl) * 256.0f);
    i0 = base;
    i1 = base + 1;
    if (base < 0 || base + 1 >= n) { i0 = resolve_index(base, n, ZG_BORDER_MIRROR); i1 = resolve_index(base + 1, n, ZG_BORDER_MIRROR); }
}
// Four channels at once on packed 16-bit halves: the horizontal lerps a (256 - fx) + b fx stay below 2^16 (255 * 256), so
// channel pairs (0, 2) and (1, 3) go through v_pk_mul_lo_u16 / v_pk_mad_u16; the vertical lerp needs 24 bits and takes its
// 16-bit operands straight out of the packed halves (v_mad_u32_u16 with op_sel); the result of each channel is byte 2 of its
// accumulator (@divTrunc(.., 65536), <= 255). Integer arithmetic throughout: the same values as the scalar form, in half the
// instructions (the up-scaling resizes are VALU-bound).
__device__ inline uint32_t bilinear_rgba8(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, int fx, int fy) {
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    auto even = [](uint32_t px) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, px, 0x0c020c00u)); }; // channels 0, 2
    auto odd = [](uint32_t px) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, px, 0x0c030c01u)); };  // channels 1, 3
    const uint16_t w1 = (uint16_t)fx, w0 = (uint16_t)(256 - fx);
    const u16x2 k0 = {w0, w0}, k1 = {w1, w1};
    const uint32_t top02 = __builtin_bit_cast(uint32_t, even(tl) * k0 + even(tr) * k1), top13 = __builtin_bit_cast(uint32_t, odd(tl) * k0 + odd(tr) * k1);
    const uint32_t bot02 = __builtin_bit_cast(uint32_t, even(bl) * k0 + even(br) * k1), bot13 = __builtin_bit_cast(uint32_t, odd(bl) * k0 + odd(br) * k1);
    const uint32_t v0 = (uint32_t)(256 - fy), v1 = (uint32_t)fy;
    uint32_t c0, c1, c2, c3;
    asm("v_mad_u32_u16 %0, %1, %2, 0" : "=v"(c0) : "v"(top02), "v"(v0));
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(c0) : "v"(bot02), "v"(v1), "v"(c0));
    asm("v_mad_u32_u16 %0, %1, %2, 0" : "=v"(c1) : "v"(top13), "v"(v0));
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(c1) : "v"(bot13), "v"(v1), "v"(c1));
    asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(c2) : "v"(top02), "v"(v0));
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(c2) : "v"(bot02), "v"(v1), "v"(c2));
    asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(c3) : "v"(top13), "v"(v0));
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(c3) : "v"(bot13), "v"(v1), "v"(c3));
    // byte 2 of c0 -> byte 0, of c1 -> byte 1, of c2 -> byte 2, of c3 -> byte 3
    return __builtin_amdgcn_perm(c1, c0, 0x0c0c0602u) | __builtin_amdgcn_perm(c3, c2, 0x06020c0cu);
}

} // namespace zg
