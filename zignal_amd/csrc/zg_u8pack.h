// zg_u8pack.h — packed-u16 building blocks shared by the u8 separable fast paths (conv_sep_rgba8.hip, conv_sep_bytes.hip).
#pragma once
#include "zg_common.h"

namespace zg {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

template <int N> struct TapsU8 { uint32_t k[N]; }; // plain integer taps (0..255)

constexpr int R8_TW = 256;      // tile width in pixels = 64 lanes x 4 pixels
constexpr int R8_UNITS = 66;    // 16-byte units per LDS row: one halo unit left, 64, one right

__device__ inline u16x2 pair_lo(uint32_t px) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, px, 0x0c010c00u)); } // (r, g)
__device__ inline u16x2 pair_hi(uint32_t px) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, px, 0x0c030c02u)); } // (b, a)

// acc + half(packed) * k with the 16-bit half picked by op_sel: one full-rate VALU op per (channel, tap), no unpacking.
// (hipcc otherwise lowers the u32 column pass to v_mul_u32_u24 + v_add3_u32 pairs plus and/shift extractions.)
__device__ inline uint32_t mad_lo16(uint32_t packed, uint32_t k, uint32_t acc) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(packed), "s"(k), "v"(acc));
    return d;
}
__device__ inline uint32_t mad_hi16(uint32_t packed, uint32_t k, uint32_t acc) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(packed), "s"(k), "v"(acc));
    return d;
}

} // namespace zg
