// conv_sep_rgba8.hip — fast path of Image(Rgba(u8)).convolveSeparable / gaussianBlur for small non-negative
// integer kernels (every Gaussian the reference builds: taps round(k*256) in [0,255], sum <= 257).
//
// Same arithmetic contract as conv_separable.hip (reference src/image/convolution.zig:441-647, u8 path):
// temp = sum src*kx (exact), out = divClampU8(65536, sum temp*ky). All integer, so any evaluation order is exact;
// this kernel exploits that: with taps in [0,255] and sum <= 257 the horizontal temp fits 16 bits (<= 65535), so
//   * the row pass runs on packed u16 pairs (two channels per VALU lane-op, v_pk_mul_lo_u16 / v_pk_add_u16),
//   * the column pass accumulates in u32 straight from the packed halves,
//   * and each lane owns FOUR adjacent pixels: 16-byte global loads, LDS reads and stores instead of 4-byte ones.
// Tile = 256 x 4*RPT pixels per workgroup, staged in LDS as 16-byte units with a 4-pixel (one unit) halo on
// each side so every access stays 16-byte aligned.
//
// Preconditions (checked by the caller, else the general kernel runs): Rgba(u8), cols % 4 == 0, strides % 4 == 0,
// 16-byte aligned bases, odd equal tap counts <= 9, taps as above.
#include "zg_common.h"
#include "zg_u8pack.h"
#include <algorithm>
#include <cstdlib>

namespace zg {

// Register staging of one (4*RPT + 2H) x 66-unit source tile: all loads first, LDS writes later. Single frames now run
// on the byte-stream kernel (conv_sep_bytes.hip, same arithmetic); this one keeps the batched and blur+half-resize forms.
template <int NK, int RPT> struct Stage8 {
    static constexpr int H = NK / 2;
    static constexpr int LH = 4 * RPT + 2 * H;
    static constexpr int RW = (LH + 3) / 4;
    static constexpr int NEXTRA = LH * 2; // units 64 and 65 of every row
    u32x4 main_v[RW];
    u32x4 extra_v;

    // tile row r, unit u: pixels x0 - 4 + 4u .. +3 of image row y0 - H + r. Units are all inside or all outside the row
    // (cols % 4 == 0); outside ones (and rows the zero border drops) become 0 here and the pixels of them that the taps
    // can reach are filled in by patch_edges. The load itself is unconditional from a clamped address: predicated loads
    // would be issued one at a time.
    __device__ static __forceinline__ u32x4 load_unit(const DImg &src, int x0, int y0, int border, int r, int u) {
        const int gr = resolve_index(y0 - H + r, src.rows, border);
        const int gx = x0 - 4 + 4 * u;
        const bool ok = gr >= 0 && gx >= 0 && gx + 4 <= src.cols;
        const uint32_t *row = (const uint32_t *)src.data + (size_t)max(gr, 0) * src.stride;
        u32x4 v = *(const u32x4 *)(row + min(max(gx, 0), src.cols - 4)); // 16-byte aligned by the preconditions
        if (!ok) v = u32x4{0u, 0u, 0u, 0u};
        return v;
    }
    __device__ __forceinline__ void load(const DImg &src, int x0, int y0, int border, int lx, int wave) {
#pragma unroll
        for (int k = 0; k < RW; ++k) main_v[k] = load_unit(src, x0, y0, border, min(wave + 4 * k, LH - 1), lx);
        const int e = min((int)threadIdx.x, NEXTRA - 1); // lanes past NEXTRA load a duplicate and do not spill it
        extra_v = load_unit(src, x0, y0, border, e >> 1, 64 + (e & 1));
    }
    // Border rule for the columns: the H pixels left of column 0 and right of the last column, where this tile covers
    // them, one pixel per lane straight from global memory into the LDS tile (edge tiles only).
    __device__ static void patch_edges(u32x4 *tile, const DImg &src, int x0, int y0, int border) {
        for (int idx = (int)threadIdx.x; idx < LH * 2 * H; idx += 256) {
            const int r = idx / (2 * H), k = idx - r * (2 * H);
            const int px = k < H ? -1 - k : src.cols + (k - H);
            const int t = px - (x0 - 4); // pixel position in the tile row
            if (t < 0 || t >= R8_UNITS * 4) continue;
            const int gr = resolve_index(y0 - H + r, src.rows, border);
            const int gc = resolve_index(px, src.cols, border);
            if (gr < 0 || gc < 0) continue; // zero border: already 0
            ((uint32_t *)tile)[(size_t)r * R8_UNITS * 4 + t] = ((const uint32_t *)src.data)[(size_t)gr * src.stride + gc];
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

// Row pass (packed u16) into a sliding window, column pass (u32), 16-byte row-clipped stores.
// DOWN2 fuses Image.resize(.bilinear) at exactly half size behind the blur (the `pipeline` recipe [blur, resize x0.5],
// reference src/cli/pipeline.zig:153-179): with ratio 2 the plane kernel's taps are 2d and 2d+1 with fx = fy = 128
// (src/image/channel_ops.zig:144-190), so out = ((tl + tr) * 128 * 128 + (bl + br) * 128 * 128) >> 16 = (tl + tr + bl + br) >> 2
// of the BLURRED pixels — two adjacent pixels of a lane and two consecutive rows of its strip. The blurred frame
// never touches HBM.
// Arithmetic on ROW PAIRS, as in conv_sep_bytes.hip: the u16 pair (byte S of tile row r, byte S of tile row r + 1) is unpacked once per
// position (one v_perm) and reused by every tap that reaches it (taps are 4 bytes apart: whole pixels); the row pass is one
// v_pk_mad_u16 per tap and byte for two rows, and its results are (row 2q, row 2q + 1) pairs, which v_dot2_u32_u16 consumes two taps at a
// time in the column pass (it used to be one v_mad_u32_u16 per tap and channel). An output row pair is also exactly what DOWN2 averages.
template <int S> __device__ __forceinline__ u16x2 row_pair8(const uint32_t (&q0)[12], const uint32_t (&q1)[12]) {
    constexpr int d = S >> 2, o = S & 3;
    constexpr uint32_t sel = 0x0c000c00u | ((4u + o) << 16) | (uint32_t)o; // byte o of q0[d] -> low half, byte o of q1[d] -> high half
    return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q1[d], q0[d], sel));
}
template <int S0, int N, int I = 0> struct UnpackRows8 { // P[I] = position S0 + I, I < N
    __device__ static __forceinline__ void run(const uint32_t (&q0)[12], const uint32_t (&q1)[12], u16x2 (&P)[N]) {
        if constexpr (I < N) {
            P[I] = row_pair8<S0 + I>(q0, q1);
            UnpackRows8<S0, N, I + 1>::run(q0, q1, P);
        }
    }
};
__device__ __forceinline__ uint32_t dot2_u16_8(uint32_t packed, uint32_t kpair, uint32_t acc) { // acc + lo * klo + hi * khi
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, packed), __builtin_bit_cast(u16x2, kpair), acc, false);
}

template <int NK, int RPT, bool NT, bool CLAMP, bool DOWN2>
__device__ __forceinline__ void convolve_tile8(const u32x4 *tile, const DImg &dst, const TapsU8<NK> &kx, const TapsU8<NK> &ky,
                                               int x0, int y0, int lx, int wave) {
    constexpr int H = NK / 2;
    static_assert(RPT % 2 == 0, "rows are processed in pairs");
    constexpr int NQ = H + 1;           // row pairs the column pass of one output pair reaches
    constexpr int NP = 16 + 8 * H;      // unpacked positions: the lane's sixteen bytes and H pixels on both sides
    const int gx = x0 + 4 * lx;         // first of this lane's four pixels
    u16x2 win[NQ][16];                  // [row pair][byte]: (temp of tile row 2q, temp of tile row 2q + 1)
#pragma unroll
    for (int q = 0; q < (RPT + 2 * H) / 2; ++q) {
        const int lr = wave * RPT + 2 * q;
        const u32x4 a0 = tile[lr * R8_UNITS + lx], b0 = tile[lr * R8_UNITS + lx + 1], c0 = tile[lr * R8_UNITS + lx + 2];
        const u32x4 a1 = tile[(lr + 1) * R8_UNITS + lx], b1 = tile[(lr + 1) * R8_UNITS + lx + 1], c1 = tile[(lr + 1) * R8_UNITS + lx + 2];
        const uint32_t q0[12] = {a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3], c0[0], c0[1], c0[2], c0[3]}; // q0[4] = pixel gx
        const uint32_t q1[12] = {a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3], c1[0], c1[1], c1[2], c1[3]};
        u16x2 P[NP];
        UnpackRows8<16 - 4 * H, NP>::run(q0, q1, P);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            u16x2 acc = {0, 0};
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const uint16_t k = (uint16_t)kx.k[i];
                const u16x2 kk = {k, k};
                acc += P[t + 4 * i] * kk; // P[0] is position 16 - 4 H; <= 65535 by the preconditions: exact
            }
            win[q % NQ][t] = acc;
        }
        if (q >= H) { // output rows 2m, 2m + 1 of the strip, m = q - H: tile rows 2m .. 2m + 2H + 1 = row pairs m .. m + H
            const int m = q - H;
            const int gy = y0 + wave * RPT + 2 * m;
            uint32_t ve[16], vo[16]; // one per output byte, even row and odd row
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                // divClampU8(65536, a) for a >= 0 is min(255, (a + 32768) >> 16): the rounding term seeds the accumulator
                uint32_t e = 32768u, o = 32768u;
                o = mad_hi16(__builtin_bit_cast(uint32_t, win[m % NQ][t]), ky.k[0], o);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    e = dot2_u16_8(__builtin_bit_cast(uint32_t, win[(m + h) % NQ][t]), ky.k[2 * h] | (ky.k[2 * h + 1] << 16), e);
                    o = dot2_u16_8(__builtin_bit_cast(uint32_t, win[(m + 1 + h) % NQ][t]), ky.k[2 * h + 1] | (ky.k[2 * h + 2] << 16), o);
                }
                e = mad_lo16(__builtin_bit_cast(uint32_t, win[(m + H) % NQ][t]), ky.k[NK - 1], e);
                if constexpr (CLAMP) {
                    e >>= 16; o >>= 16;
                    ve[t] = e > 255u ? 255u : e;
                    vo[t] = o > 255u ? 255u : o;
                } else if constexpr (DOWN2) { // host proved acc < 2^24: the value is byte 2
                    ve[t] = e >> 16;
                    vo[t] = o >> 16;
                } else { // ... extracted by the packing below
                    ve[t] = e;
                    vo[t] = o;
                }
            }
            if constexpr (!DOWN2) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const uint32_t(&v)[16] = half == 0 ? ve : vo;
                    u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if constexpr (CLAMP) o[d] = v[4 * d] | (v[4 * d + 1] << 8) | (v[4 * d + 2] << 16) | (v[4 * d + 3] << 24);
                        else o[d] = __builtin_amdgcn_perm(v[4 * d + 1], v[4 * d], 0x0c0c0602u) | __builtin_amdgcn_perm(v[4 * d + 3], v[4 * d + 2], 0x06020c0cu);
                    }
                    const bool row_ok = gy + half < dst.rows;
                    char *row = (char *)dst.data + (row_ok ? (size_t)(gy + half) * dst.stride * 4 : (size_t)0);
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, row_ok ? dst.cols * 4 : 0, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, gx * 4, 0, NT ? 2 : 0); // cols % 4 == 0: a unit is all in or all out
                }
            } else {
                // 2 x 2 means of the blurred pixels: pixels (0, 1) -> output 0, (2, 3) -> output 1, rows 2m and 2m + 1 (y0 and RPT are even)
                uint32_t opx[2];
#pragma unroll
                for (int o2 = 0; o2 < 2; ++o2) {
                    uint32_t c[4];
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) c[ch] = (ve[8 * o2 + ch] + ve[8 * o2 + 4 + ch] + vo[8 * o2 + ch] + vo[8 * o2 + 4 + ch]) >> 2;
                    opx[o2] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
                }
                const int oy = gy >> 1;
                const bool row_ok = oy < dst.rows;
                char *row = (char *)dst.data + (row_ok ? (size_t)oy * dst.stride * 4 : (size_t)0);
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, row_ok ? dst.cols * 4 : 0, 0x00020000);
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 o = {opx[0], opx[1]};
                __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, (gx >> 1) * 4, 0, NT ? 2 : 0); // dst.cols even: a pair is all in or all out
            }
        }
    }
}

// One workgroup per (frame, tile). Frames of a batch are laid out back to back (frame strides in pixels); with many
// frames in one launch the workgroups of different frames drift apart and load / compute phases overlap, which a single
// small frame (a few hundred workgroups, ~2 rounds) cannot do.
template <int NK, int RPT, bool NT, bool CLAMP, bool DOWN2>
__global__ __launch_bounds__(256) void k_sep_rgba8(DImg src, DImg dst, size_t src_frame_px, size_t dst_frame_px,
                                                   TapsU8<NK> kx, TapsU8<NK> ky, int border, int tiles_x, int tiles_per_frame) {
    using Stage = Stage8<NK, RPT>;
    constexpr int TH = 4 * RPT;
    __shared__ u32x4 tile[Stage::LH * R8_UNITS];

    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3); // XCD-major order
    const int frame = wg / tiles_per_frame, t = wg - frame * tiles_per_frame;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    src.data = (uint32_t *)src.data + (size_t)frame * src_frame_px;
    dst.data = (uint32_t *)dst.data + (size_t)frame * dst_frame_px;
    const int lx = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    Stage st;
    st.load(src, tx * R8_TW, ty * TH, border, lx, wave);
    st.spill(tile, lx, wave);
    if (tx == 0 || tx * R8_TW + R8_TW + 4 > src.cols) { // workgroup-uniform: this tile sees the left or right border
        __syncthreads();
        Stage::patch_edges(tile, src, tx * R8_TW, ty * TH, border);
    }
    __syncthreads();
    convolve_tile8<NK, RPT, NT, CLAMP, DOWN2>(tile, dst, kx, ky, tx * R8_TW, ty * TH, lx, wave);
}

struct Rgba8Batch { // frames laid out back to back
    const void *src; void *dst;
    uint32_t n_frames, rows, cols;
    size_t src_stride, dst_stride;         // row strides in pixels
    size_t src_frame_px, dst_frame_px;     // frame strides in pixels
    bool down2;                            // dst is (rows/2) x (cols/2): blur then 2:1 bilinear
};

template <int NK, int RPT, bool CLAMP, bool DOWN2>
static int launch_rgba8(const Rgba8Batch &b, const int32_t *ix, const int32_t *iy, int border, hipStream_t s) {
    TapsU8<NK> kx, ky;
    for (int i = 0; i < NK; ++i) { kx.k[i] = (uint32_t)ix[i]; ky.k[i] = (uint32_t)iy[i]; }
    const int tiles_x = (int)ceil_div(b.cols, R8_TW), tiles_y = (int)ceil_div(b.rows, 4 * RPT);
    const int tiles_per_frame = tiles_x * tiles_y;
    const DImg src{(void *)b.src, b.src_stride, (int32_t)b.rows, (int32_t)b.cols};
    const DImg dst{b.dst, b.dst_stride, (int32_t)(DOWN2 ? b.rows / 2 : b.rows), (int32_t)(DOWN2 ? b.cols / 2 : b.cols)};
    hipLaunchKernelGGL((k_sep_rgba8<NK, RPT, true, CLAMP, DOWN2>), dim3((unsigned)(tiles_per_frame * b.n_frames)), dim3(256), 0, s,
                       src, dst, b.src_frame_px, b.dst_frame_px, kx, ky, border, tiles_x, tiles_per_frame);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Returns -1 when the preconditions do not hold (caller falls back to the general kernels).
int try_sep_rgba8_batch(const Rgba8Batch &b, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s) {
    if (nk != 3 && nk != 5 && nk != 7 && nk != 9) return -1;
    if (b.cols % 4 || b.src_stride % 4 || b.src_frame_px % 4 || ((uintptr_t)b.src & 15)) return -1;
    if (b.down2) {
        if (b.rows % 2 || b.cols % 4 || b.dst_stride % 2 || b.dst_frame_px % 2 || ((uintptr_t)b.dst & 7)) return -1;
    } else {
        if (b.dst_stride % 4 || b.dst_frame_px % 4 || ((uintptr_t)b.dst & 15)) return -1;
    }
    if (b.cols < 64) return -1; // tiny images: the 256-wide tile is mostly padding
    int64_t sx = 0, sy = 0;
    for (int i = 0; i < nk; ++i) {
        if (ix[i] < 0 || ix[i] > 255 || iy[i] < 0 || iy[i] > 255) return -1;
        sx += ix[i];
        sy += iy[i];
    }
    if (sx > 257 || sy > 257) return -1; // temp must fit u16: 255 * 257 = 65535
    const bool clamp = sx * sy * 255 + 32768 >= 256 * 65536; // only then can (acc >> 16) exceed 255
    // RPT 4 (21 KB of LDS, 7 workgroups / CU) measured best on MI355X: profiles/r01_sep_variant_sweep.txt
#define ZG_R8(NK) case NK: \
        if (b.down2) return clamp ? launch_rgba8<NK, (NK < 9 ? 8 : 4), true, true>(b, ix, iy, border, s) : launch_rgba8<NK, (NK < 9 ? 8 : 4), false, true>(b, ix, iy, border, s); \
        return clamp ? launch_rgba8<NK, 4, true, false>(b, ix, iy, border, s) : launch_rgba8<NK, 4, false, false>(b, ix, iy, border, s);
    switch (nk) { ZG_R8(3) ZG_R8(5) ZG_R8(7) ZG_R8(9) }
#undef ZG_R8
    return -1;
}

int try_sep_rgba8(const zg_image *src, const zg_image *dst, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_RGBA_U8) return -1;
    Rgba8Batch b{src->data, dst->data, 1, src->rows, src->cols, src->stride, dst->stride, 0, 0, false};
    return try_sep_rgba8_batch(b, ix, iy, nk, border, s);
}

} // namespace zg
