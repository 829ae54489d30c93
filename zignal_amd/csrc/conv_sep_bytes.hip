// conv_sep_bytes.hip — fast path of Image(u8) and Image(Rgb(u8)) convolveSeparable / gaussianBlur for small
// non-negative integer kernels (every Gaussian the reference builds: taps round(k*256) in [0,255], sum <= 257).
//
// Same arithmetic contract as conv_separable.hip (reference src/image/convolution.zig:441-647, u8 path):
// temp = sum src*kx (exact), out = divClampU8(65536, sum temp*ky), every channel with the same taps. That last point
// is what this kernel is built on: a row of SP-byte pixels is a plain BYTE STREAM in which tap i of byte b is byte
// b + SP*(i - H), whatever channel b belongs to (SP = 1 for a grey plane, 3 for Rgb). So a lane owns 16 consecutive
// bytes of a row (one 16-byte load / LDS access / store — 16 grey pixels or 5 1/3 Rgb pixels) instead of one pixel,
// and pixel structure only matters where the border rule resolves an out-of-range pixel, i.e. in the edge tiles'
// staging. Arithmetic as in conv_sep_rgba8.hip: the row pass on packed u16 pairs of ADJACENT bytes (even pairs come
// from one v_perm of a dword, odd pairs from one v_perm of a dword and its neighbour), the column pass with
// v_mad_u32_u16 straight from the packed halves, byte-select finalisation.
//
// Tile = 1024 bytes x 4*RPT rows per workgroup, staged in LDS as 16-byte units with one halo unit on each side
// (H*SP <= 12 bytes). Preconditions (else the general kernel runs): u8 or Rgb(u8), row length and strides multiples of
// 16 bytes, 16-byte aligned bases, odd equal tap counts <= 9, taps as above.
#include "zg_common.h"
#include "zg_u8pack.h"

namespace zg {

template <int SP, int NK, int RPT> struct StageB {
    static constexpr int H = NK / 2;
    static constexpr int LH = 4 * RPT + 2 * H;
    static constexpr int RW = (LH + 3) / 4;
    static constexpr int NEXTRA = LH * 2; // units 64 and 65 of every row
    u32x4 main_v[RW];
    u32x4 extra_v;

    // tile row r, unit u: bytes xb0 - 16 + 16u .. +15 of image row y0 - H + r (byte b = channel b % SP of pixel b / SP)
    __device__ static __forceinline__ u32x4 load_unit(const DImg &src, int xb0, int y0, int border, int r, int u) {
        const int gr = resolve_index(y0 - H + r, src.rows, border);
        const int gb = xb0 - 16 + 16 * u;
        const int row_bytes = src.cols * SP;
        // Units are all inside or all outside the row (row length % 16 == 0). Outside ones (and rows the zero border
        // drops) become 0 here; the bytes of them that the taps can reach are filled in by patch_edges. The load itself
        // is unconditional from a clamped address: predicated loads would be issued one at a time.
        const bool ok = gr >= 0 && gb >= 0 && gb + 16 <= row_bytes;
        const uint8_t *row = (const uint8_t *)src.data + (size_t)max(gr, 0) * src.stride * SP;
        u32x4 v = *(const u32x4 *)(row + min(max(gb, 0), row_bytes - 16)); // 16-byte aligned by the preconditions
        if (!ok) v = u32x4{0u, 0u, 0u, 0u};
        return v;
    }
    __device__ __forceinline__ void load(const DImg &src, int xb0, int y0, int border, int lx, int wave) {
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int r = wave + 4 * k;
            main_v[k] = load_unit(src, xb0, y0, border, min(r, LH - 1), lx);
        }
        const int e = min((int)threadIdx.x, NEXTRA - 1); // lanes past NEXTRA load a duplicate and do not spill it
        extra_v = load_unit(src, xb0, y0, border, e >> 1, 64 + (e & 1));
    }
    // Border rule for the columns: the H*SP bytes left of byte 0 and right of the last byte of the row, where this tile
    // covers them, one byte per lane straight from global memory into the LDS tile (edge tiles only).
    __device__ static void patch_edges(u32x4 *tile, const DImg &src, int xb0, int y0, int border) {
        constexpr int PB = H * SP; // bytes per side a tap can reach
        const int row_bytes = src.cols * SP;
        for (int idx = (int)threadIdx.x; idx < LH * 2 * PB; idx += 256) {
            const int r = idx / (2 * PB), k = idx - r * (2 * PB);
            const int b = k < PB ? -1 - k : row_bytes + (k - PB); // byte position in the row's stream
            const int t = b - (xb0 - 16);                         // byte position in the tile row
            if (t < 0 || t >= R8_UNITS * 16) continue;
            const int gr = resolve_index(y0 - H + r, src.rows, border);
            const int px = b >= 0 ? b / SP : -((SP - 1 - b) / SP); // floor
            const int gc = resolve_index(px, src.cols, border);
            if (gr < 0 || gc < 0) continue; // zero border: already 0
            ((uint8_t *)tile)[(size_t)r * R8_UNITS * 16 + t] = ((const uint8_t *)src.data)[((size_t)gr * src.stride + gc) * SP + (b - px * SP)];
        }
    }
    __device__ void spill(u32x4 *tile, int lx, int wave) const {
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int r = wave + 4 * k;
            if (r < LH) tile[r * R8_UNITS + lx] = main_v[k];
        }
        const int e = (int)threadIdx.x;
        if (e < NEXTRA) tile[(e >> 1) * R8_UNITS + 64 + (e & 1)] = extra_v;
    }
};

// Arithmetic on ROW PAIRS: the u16 pair (byte S of tile row r, byte S of tile row r + 1) of two 48-byte windows held as 12 dwords
// each (S is a compile-time constant after unrolling). A position is unpacked once (one v_perm) and every tap that reaches it reuses the
// register — tap i of output byte b reads position b + SP * (i - H), whatever SP is — so the row pass is one v_pk_mad_u16 per tap and
// byte for two rows; and the temps come out as (row 2q, row 2q + 1) pairs, which is what v_dot2_u32_u16 wants in the column pass: two
// taps per instruction. (Pairs of adjacent bytes of one row needed a v_perm per tap, and the column pass one v_mad_u32_u16 per tap:
// ten instructions per byte against six and a half.)
template <int S> __device__ __forceinline__ u16x2 row_pair(const uint32_t (&q0)[12], const uint32_t (&q1)[12]) {
    constexpr int d = S >> 2, o = S & 3;
    constexpr uint32_t sel = 0x0c000c00u | ((4u + o) << 16) | (uint32_t)o; // byte o of q0[d] -> low half, byte o of q1[d] -> high half
    return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q1[d], q0[d], sel));
}
template <int SP, int NK, int S0, int N, int I = 0> struct UnpackRows { // P[I] = position S0 + I, I < N
    __device__ static __forceinline__ void run(const uint32_t (&q0)[12], const uint32_t (&q1)[12], u16x2 (&P)[N]) {
        if constexpr (I < N) {
            P[I] = row_pair<S0 + I>(q0, q1);
            UnpackRows<SP, NK, S0, N, I + 1>::run(q0, q1, P);
        }
    }
};
template <int SP, int NK, int N, int T, int I> struct RowTaps2 { // unrolled over taps: register index is a compile-time constant
    __device__ static __forceinline__ u16x2 run(const u16x2 (&P)[N], const TapsU8<NK> &kx, u16x2 acc) {
        if constexpr (I == NK) return acc;
        else {
            const uint16_t k = (uint16_t)kx.k[I];
            const u16x2 kk = {k, k};
            acc += P[T + SP * I] * kk; // P[0] is position 16 - H * SP; <= 65535 by the preconditions: exact
            return RowTaps2<SP, NK, N, T, I + 1>::run(P, kx, acc);
        }
    }
};
template <int SP, int NK, int N, int T> struct RowBytes2 { // unrolled over the lane's sixteen output bytes
    __device__ static __forceinline__ void run(const u16x2 (&P)[N], const TapsU8<NK> &kx, u16x2 (&out)[16]) {
        if constexpr (T < 16) {
            out[T] = RowTaps2<SP, NK, N, T, 0>::run(P, kx, u16x2{0, 0});
            RowBytes2<SP, NK, N, T + 1>::run(P, kx, out);
        }
    }
};
__device__ __forceinline__ uint32_t dot2_u16(uint32_t packed, uint32_t kpair, uint32_t acc) { // acc + lo * klo + hi * khi
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, packed), __builtin_bit_cast(u16x2, kpair), acc, false);
}

template <int SP, int NK, int RPT, bool CLAMP>
__global__ __launch_bounds__(256) void k_sep_bytes(DImg src, DImg dst, TapsU8<NK> kx, TapsU8<NK> ky, int border, int tiles_x) {
    using Stage = StageB<SP, NK, RPT>;
    constexpr int H = NK / 2;
    constexpr int TH = 4 * RPT;
    __shared__ u32x4 tile[Stage::LH * R8_UNITS];

    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3); // XCD-major order
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int xb0 = tx * 1024, y0 = ty * TH;
    const int lx = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    Stage st;
    st.load(src, xb0, y0, border, lx, wave);
    st.spill(tile, lx, wave);
    if (xb0 == 0 || xb0 + 1024 + 16 > src.cols * SP) { // workgroup-uniform: this tile sees the left or right border
        __syncthreads();
        Stage::patch_edges(tile, src, xb0, y0, border);
    }
    __syncthreads();

    static_assert(RPT % 2 == 0, "rows are processed in pairs");
    constexpr int NQ = H + 1;                 // row pairs the column pass of one output pair reaches
    constexpr int NP = 16 + 2 * H * SP;       // unpacked positions: the lane's sixteen bytes and the taps' reach on both sides
    u16x2 win[NQ][16];                        // [row pair][byte]: (temp of tile row 2q, temp of tile row 2q + 1)
#pragma unroll
    for (int q = 0; q < (RPT + 2 * H) / 2; ++q) {
        const int lr = wave * RPT + 2 * q;
        const u32x4 a0 = tile[lr * R8_UNITS + lx], b0 = tile[lr * R8_UNITS + lx + 1], c0 = tile[lr * R8_UNITS + lx + 2];
        const u32x4 a1 = tile[(lr + 1) * R8_UNITS + lx], b1 = tile[(lr + 1) * R8_UNITS + lx + 1], c1 = tile[(lr + 1) * R8_UNITS + lx + 2];
        const uint32_t q0[12] = {a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3], c0[0], c0[1], c0[2], c0[3]}; // byte 16 = this lane's first
        const uint32_t q1[12] = {a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3], c1[0], c1[1], c1[2], c1[3]};
        u16x2 P[NP];
        UnpackRows<SP, NK, 16 - H * SP, NP>::run(q0, q1, P);
        RowBytes2<SP, NK, NP, 0>::run(P, kx, win[q % NQ]);
        if (q >= H) { // output rows 2m, 2m + 1 of the strip, m = q - H: tile rows 2m .. 2m + 2H + 1 = row pairs m .. m + H
            const int m = q - H;
            uint32_t ve[16], vo[16]; // one per output byte, even row and odd row
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                // divClampU8(65536, a) for a >= 0 is min(255, (a + 32768) >> 16): the rounding term seeds the accumulator
                uint32_t e = 32768u, o = 32768u;
                o = mad_hi16(__builtin_bit_cast(uint32_t, win[m % NQ][t]), ky.k[0], o);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    e = dot2_u16(__builtin_bit_cast(uint32_t, win[(m + h) % NQ][t]), ky.k[2 * h] | (ky.k[2 * h + 1] << 16), e);
                    o = dot2_u16(__builtin_bit_cast(uint32_t, win[(m + 1 + h) % NQ][t]), ky.k[2 * h + 1] | (ky.k[2 * h + 2] << 16), o);
                }
                e = mad_lo16(__builtin_bit_cast(uint32_t, win[(m + H) % NQ][t]), ky.k[NK - 1], e);
                if constexpr (CLAMP) {
                    e >>= 16; o >>= 16;
                    ve[t] = e > 255u ? 255u : e;
                    vo[t] = o > 255u ? 255u : o;
                } else { // host proved acc < 2^24: the value is byte 2, extracted below
                    ve[t] = e;
                    vo[t] = o;
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint32_t(&v)[16] = half == 0 ? ve : vo;
                u32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if constexpr (CLAMP) o[d] = v[4 * d] | (v[4 * d + 1] << 8) | (v[4 * d + 2] << 16) | (v[4 * d + 3] << 24);
                    else o[d] = __builtin_amdgcn_perm(v[4 * d + 1], v[4 * d], 0x0c0c0602u) | __builtin_amdgcn_perm(v[4 * d + 3], v[4 * d + 2], 0x06020c0cu);
                }
                const int gy = y0 + wave * RPT + 2 * m + half;
                const bool row_ok = gy < dst.rows;
                char *row = (char *)dst.data + (row_ok ? (size_t)gy * dst.stride * SP : (size_t)0);
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, row_ok ? dst.cols * SP : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, xb0 + 16 * lx, 0, 2); // row bytes % 16 == 0: a unit is all in or all out
            }
        }
    }
}

template <int SP, int NK, int RPT, bool CLAMP>
static int launch_bytes(const zg_image *src, const zg_image *dst, const int32_t *ix, const int32_t *iy, int border, hipStream_t s) {
    TapsU8<NK> kx, ky;
    for (int i = 0; i < NK; ++i) { kx.k[i] = (uint32_t)ix[i]; ky.k[i] = (uint32_t)iy[i]; }
    const int tiles_x = (int)ceil_div((uint32_t)(src->cols * SP), 1024u), tiles_y = (int)ceil_div(src->rows, 4 * RPT);
    hipLaunchKernelGGL((k_sep_bytes<SP, NK, RPT, CLAMP>), dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, s, dimg(src), dimg(dst), kx,
                       ky, border, tiles_x);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int SP>
static int dispatch_bytes(const zg_image *src, const zg_image *dst, const int32_t *ix, const int32_t *iy, int nk, bool clamp, int border,
                          hipStream_t s) {
    // Rows per lane: 8 amortises the 2H halo rows of the row pass better (measured 33.2 / 26.9 / 11.5 us against
    // 33.8 / 28.9 / 11.9 us at 4 for 4096^2 Rgba / Rgb / grey) but halves the workgroup count, so only for frames that
    // still give every CU several workgroups (and not for 9 taps, whose 16-step strip loop the compiler stops unrolling).
    const bool tall = (uint64_t)ceil_div((uint32_t)(src->cols * SP), 1024u) * ceil_div(src->rows, 32u) >= 2048;
#define ZG_B8(NK) case NK: \
        if (tall && NK < 9) return clamp ? launch_bytes<SP, NK, (NK < 9 ? 8 : 4), true>(src, dst, ix, iy, border, s) : launch_bytes<SP, NK, (NK < 9 ? 8 : 4), false>(src, dst, ix, iy, border, s); \
        return clamp ? launch_bytes<SP, NK, 4, true>(src, dst, ix, iy, border, s) : launch_bytes<SP, NK, 4, false>(src, dst, ix, iy, border, s);
    switch (nk) { ZG_B8(3) ZG_B8(5) ZG_B8(7) ZG_B8(9) }
#undef ZG_B8
    return -1;
}

// Returns -1 when the preconditions do not hold (caller falls back to the general kernels).
int try_sep_bytes(const zg_image *src, const zg_image *dst, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_U8 && src->pixel != ZG_PIXEL_RGB_U8 && src->pixel != ZG_PIXEL_RGBA_U8) return -1;
    if (nk != 3 && nk != 5 && nk != 7 && nk != 9) return -1;
    const size_t sp = pixel_size(src->pixel);
    if ((src->cols * sp) % 16 || (src->stride * sp) % 16 || (dst->stride * sp) % 16 || ((uintptr_t)src->data & 15) || ((uintptr_t)dst->data & 15)) return -1;
    if (src->cols * sp < 256) return -1; // tiny images: the 1024-byte tile is mostly padding
    if ((uint64_t)src->cols * sp > 0x7fffffffu) return -1;
    int64_t sx = 0, sy = 0;
    for (int i = 0; i < nk; ++i) {
        if (ix[i] < 0 || ix[i] > 255 || iy[i] < 0 || iy[i] > 255) return -1;
        sx += ix[i];
        sy += iy[i];
    }
    if (sx > 257 || sy > 257) return -1; // temp must fit u16: 255 * 257 = 65535
    const bool clamp = sx * sy * 255 + 32768 >= 256 * 65536; // only then can (acc >> 16) exceed 255
    if (sp == 1) return dispatch_bytes<1>(src, dst, ix, iy, nk, clamp, border, s);
    if (sp == 3) return dispatch_bytes<3>(src, dst, ix, iy, nk, clamp, border, s);
    return dispatch_bytes<4>(src, dst, ix, iy, nk, clamp, border, s);
}

} // namespace zg
