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
template <int NK, int RPT, bool NT, bool CLAMP, bool DOWN2>
__device__ __forceinline__ void convolve_tile8(const u32x4 *tile, const DImg &dst, const TapsU8<NK> &kx, const TapsU8<NK> &ky,
                                               int x0, int y0, int lx, int wave) {
    constexpr int H = NK / 2;
    const int gx = x0 + 4 * lx; // first of this lane's four pixels
    u16x2 win[NK][4][2];        // [slot][pixel][rg | ba]
    uint32_t prev_sum[2][2];    // DOWN2: horizontal pair sums of the previous (even) row, as packed u16 pairs (rg | ba)
#pragma unroll
    for (int j = 0; j < RPT + 2 * H; ++j) {
        const int lr = wave * RPT + j;
        const u32x4 a = tile[lr * R8_UNITS + lx], b = tile[lr * R8_UNITS + lx + 1], c = tile[lr * R8_UNITS + lx + 2];
        const uint32_t q[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]}; // q[4] = pixel gx
        u16x2 lo[4 + 2 * H], hi[4 + 2 * H];
#pragma unroll
        for (int i = 0; i < 4 + 2 * H; ++i) { lo[i] = pair_lo(q[4 - H + i]); hi[i] = pair_hi(q[4 - H + i]); }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u16x2 tl = {0, 0}, th = {0, 0};
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const uint16_t k = (uint16_t)kx.k[i];
                const u16x2 kk = {k, k};
                tl += lo[p + i] * kk; // <= 65535 by the preconditions: exact
                th += hi[p + i] * kk;
            }
            win[j % NK][p][0] = tl;
            win[j % NK][p][1] = th;
        }
        if (j >= 2 * H) {
            const int orow = j - 2 * H;             // output row within this wave's strip
            const int gy = y0 + wave * RPT + orow;
            uint32_t v[4][4];                       // [pixel][channel], each 0..255
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // divClampU8(65536, a) for a >= 0 is min(255, (a + 32768) >> 16): the rounding term seeds the accumulator
                uint32_t acc[4] = {32768u, 32768u, 32768u, 32768u};
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const uint32_t wl = __builtin_bit_cast(uint32_t, win[(j + 1 + i) % NK][p][0]);
                    const uint32_t wh = __builtin_bit_cast(uint32_t, win[(j + 1 + i) % NK][p][1]);
                    const uint32_t k = ky.k[i]; // wave-uniform (kernel argument): an SGPR operand
                    acc[0] = mad_lo16(wl, k, acc[0]);
                    acc[1] = mad_hi16(wl, k, acc[1]);
                    acc[2] = mad_lo16(wh, k, acc[2]);
                    acc[3] = mad_hi16(wh, k, acc[3]);
                }
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    if constexpr (CLAMP) { const uint32_t t = acc[ch] >> 16; v[p][ch] = t > 255u ? 255u : t; }
                    else v[p][ch] = acc[ch]; // host proved acc < 2^24: the value is byte 2, extracted below
                }
            }
            if constexpr (!DOWN2) {
                u32x4 o;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if constexpr (CLAMP) o[p] = v[p][0] | (v[p][1] << 8) | (v[p][2] << 16) | (v[p][3] << 24);
                    else o[p] = __builtin_amdgcn_perm(v[p][1], v[p][0], 0x0c0c0602u) | __builtin_amdgcn_perm(v[p][3], v[p][2], 0x06020c0cu);
                }
                const bool row_ok = gy < dst.rows;
                char *row = (char *)dst.data + (row_ok ? (size_t)gy * dst.stride * 4 : (size_t)0);
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, row_ok ? dst.cols * 4 : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, gx * 4, 0, NT ? 2 : 0); // cols % 4 == 0: a unit is all in or all out
            } else {
                // horizontal pair sums of this blurred row: pixels (0,1) -> output 0, (2,3) -> output 1; u16 lanes (c0 | c1 << 16)
                uint32_t hs[2][2];
#pragma unroll
                for (int o2 = 0; o2 < 2; ++o2) {
                    uint32_t c[4];
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const uint32_t a0 = CLAMP ? v[2 * o2][ch] : ((v[2 * o2][ch] >> 16) & 0xffu);
                        const uint32_t a1 = CLAMP ? v[2 * o2 + 1][ch] : ((v[2 * o2 + 1][ch] >> 16) & 0xffu);
                        c[ch] = a0 + a1;
                    }
                    hs[o2][0] = c[0] | (c[1] << 16);
                    hs[o2][1] = c[2] | (c[3] << 16);
                }
                if ((orow & 1) == 0) { // even row: keep (rows pair up inside the strip because RPT and y0 are even)
                    prev_sum[0][0] = hs[0][0]; prev_sum[0][1] = hs[0][1]; prev_sum[1][0] = hs[1][0]; prev_sum[1][1] = hs[1][1];
                } else {
                    uint32_t opx[2];
#pragma unroll
                    for (int o2 = 0; o2 < 2; ++o2) {
                        const uint32_t rg = ((prev_sum[o2][0] + hs[o2][0]) >> 2) & 0x00ff00ffu; // per-lane sums <= 1020: no carry across lanes
                        const uint32_t ba = ((prev_sum[o2][1] + hs[o2][1]) >> 2) & 0x00ff00ffu;
                        opx[o2] = (rg & 0xffu) | ((rg >> 8) & 0xff00u) | ((ba & 0xffu) << 16) | ((ba >> 16) << 24);
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
    if (wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3); // XCD-major order
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
        if (b.down2) return clamp ? launch_rgba8<NK, 4, true, true>(b, ix, iy, border, s) : launch_rgba8<NK, 4, false, true>(b, ix, iy, border, s); \
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
