// zg_stream.h — building blocks of the register-resident stream kernels (conv_sep_stream.hip, conv2d_stream.hip): one wave walks a
// 1024-byte-wide column strip from top to bottom, a lane owns 16 consecutive bytes of every row, neighbours' bytes cross the wave
// with DPP wave shifts, the image's left / right border is synthesised in the wave's outer lanes.
#pragma once
#include "zg_common.h"
#include "zg_u8pack.h"

namespace zg {

// v of the lane below / above; the lane that has none (0 / 63) keeps `old`. Never call these under divergent control flow: a
// lane switched off by EXEC is no source either.
__device__ __forceinline__ uint32_t from_lane_below(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false); } // wave_shr:1
__device__ __forceinline__ uint32_t from_lane_above(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130, 0xf, 0xf, false); } // wave_shl:1

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
template <int HB> struct HaloLoad; // HB dwords at byte offset `off` (per lane) + `soff` (wave-uniform)
template <> struct HaloLoad<1> { __device__ static __forceinline__ void run(__amdgpu_buffer_rsrc_t r, int off, int soff, uint32_t (&h)[1]) { h[0] = __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0); } };
template <> struct HaloLoad<2> { __device__ static __forceinline__ void run(__amdgpu_buffer_rsrc_t r, int off, int soff, uint32_t (&h)[2]) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0); h[0] = v[0]; h[1] = v[1]; } };
template <> struct HaloLoad<3> { __device__ static __forceinline__ void run(__amdgpu_buffer_rsrc_t r, int off, int soff, uint32_t (&h)[3]) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, off, soff, 0); h[0] = v[0]; h[1] = v[1]; h[2] = v[2]; } };
template <> struct HaloLoad<4> { __device__ static __forceinline__ void run(__amdgpu_buffer_rsrc_t r, int off, int soff, uint32_t (&h)[4]) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0); h[0] = v[0]; h[1] = v[1]; h[2] = v[2]; h[3] = v[3]; } };
// Stores are nt: written once, never read here. A store of more than 64 bits keeps reading its data registers for a cycle after
// it issues, and a VALU instruction that overwrites them right behind it corrupts what the last lanes of each 16 write (seen on
// gfx950: an even row's store came out with dword 0 of the odd row in lanes 12-15, 28-31, ...). hipcc knows the hazard and pads
// it — except when the store takes its scalar offset from a register, which its rule (inherited from older parts) exempts, wrongly
// for this one. So stores never put the row into soffset: the fast path adds it to the lane's offset instead (one v_add), and
// hipcc pads as usual. Loads have no such window and do use soffset.
__device__ __forceinline__ void st_unit(u32x4 o, __amdgpu_buffer_rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b128(o, r, off, 0, 2); }
__device__ __forceinline__ void st_unit(u32x2 o, __amdgpu_buffer_rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b64(o, r, off, 0, 2); }
__device__ __forceinline__ void st_unit(uint32_t o, __amdgpu_buffer_rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b32(o, r, off, 0, 2); }

// Byte j (0..15) of the outer lane's own unit that supplies halo byte `pos` of a border halo, or -1 when no tap reaches it.
// Left halo: HB dwords covering stream positions -4 HB .. -1. Right halo: positions rb .. rb + 4 HB - 1, the own unit being
// bytes rb - 16 .. rb - 1. Mirror is reflect-101 (border.zig:46-63): pixel -1 - m -> 1 + m, pixel cols + m -> cols - 2 - m.
template <int SP, int H> constexpr int halo_source(bool right, bool mirror, int byte_in_halo, int hb) {
    if (!right) {
        const int p = -4 * hb + byte_in_halo; // < 0
        const int k = -1 - p;
        if (k >= H * SP) return -1;
        const int m = k / SP, c = SP - 1 - (k % SP);
        return mirror ? (1 + m) * SP + c : c;
    }
    const int k = byte_in_halo;
    if (k >= H * SP) return -1;
    const int m = k / SP, c = k % SP;
    return mirror ? 16 - (2 + m) * SP + c : 16 - SP + c;
}
template <int SP, int H, int HB, bool RIGHT, bool MIRROR> __device__ __forceinline__ void synth_halo(const u32x4 &own, uint32_t (&out)[HB]) {
#pragma unroll
    for (int d = 0; d < HB; ++d) {
        uint32_t r = 0;
#pragma unroll
        for (int D = 0; D < 4; ++D) {
            uint32_t sel = 0;
            bool any = false;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = halo_source<SP, H>(RIGHT, MIRROR, 4 * d + b, HB);
                if (j >= 0 && (j >> 2) == D) {
                    sel |= (uint32_t)(j & 3) << (8 * b);
                    any = true;
                } else {
                    sel |= (uint32_t)(4 + b) << (8 * b); // keep what r holds
                }
            }
            if (any) r = __builtin_amdgcn_perm(r, own[D], sel);
        }
        out[d] = r;
    }
}

// border.resolveIndex for a row at most H outside the image, without the general rule's division (rows > 2H is a
// precondition of this kernel): one reflection / one wrap is the whole story there.
__device__ __forceinline__ int resolve_row_near(int y, int rows, int border) {
    if (y >= 0 && y < rows) return y;
    if (border == ZG_BORDER_ZERO) return -1;
    if (border == ZG_BORDER_REPLICATE) return y < 0 ? 0 : rows - 1;
    if (border == ZG_BORDER_MIRROR) return y < 0 ? -y : 2 * rows - 2 - y;
    return y < 0 ? y + rows : y - rows;
}

template <int HB> struct RowIn {
    u32x4 v;        // this lane's sixteen bytes
    uint32_t h[HB]; // lane 0: the 4 HB bytes before the strip's 1024; every other lane: the 4 HB bytes after them
};


} // namespace zg
