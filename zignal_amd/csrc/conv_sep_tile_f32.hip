// conv_sep_tile_f32.hip — Image(f32).convolveSeparable / gaussianBlur on single-channel f32 planes, one WAVE per tile (round 5).
//
// Same arithmetic contract as conv_sep_f32x4.hip / conv_separable.hip (reference src/image/convolution.zig:441-647, f32 path): per
// pixel temp = sum_i src[c + i - h] * kx[i] and out = sum_i temp[r + i - h] * ky[i], ascending i from an accumulator of 0, separate
// multiply and add (-ffp-contract=off), out-of-range taps through border.resolveIndex (src/image/border.zig:46-63) on both passes.
//
// What a 64 MiB -> 64 MiB transfer costs on this part depends on the ORDER the bytes are asked for (tools/exp/copy_floor.hip,
// profiles/r05_copy_floor.txt): waves that each load a kilobyte-wide tile of a few rows, store it and exit — thousands of them, handed
// out by the dispatcher in address order — move a 4096^2 plane in 23.4 us (5.75 TB/s) even when every tile re-reads four halo rows
// through the L2; long-lived waves that walk column strips need 25.8 us for the bare copy (the stream kernels; an f32 stream kernel was
// built first this round and ended at 30.5 us), 256-thread tiles with a barrier 24.8 us (conv_sep_f32x4.hip: 29.3 us with its LDS
// staging). Long-lived waves drift apart and scatter the request stream over DRAM pages; short-lived ones keep it a moving front.
// So: a 256-pixel x R-row tile per wave; a lane owns four consecutive pixels of every row; all R + 2H source rows are asked for up
// front (one buffer_load_dwordx4 each); the H pixels a lane needs from each neighbour cross the wave with DPP wave shifts, the wave's
// two outer lanes take theirs from a narrow second load that arrives as the DPP move's `old` operand (at the plane's left / right edge
// the outer lane synthesises the border rule from its own four pixels, or takes the other end of the row for .wrap); the row pass runs
// as the rows arrive, the column pass out of a register window of NK rows of temps; R stores; exit. No LDS, no barrier, no XCD remap:
// with 16 tiles across a 16 KiB row the vertical neighbours of a tile share its XCD (block b runs on XCD b % 8), which is where the
// halo rows are re-read from. One launch takes up to 8 equally shaped planes (blockIdx.y): BASELINE configs[1] in the one form a
// zignal caller can express for f32 data — four Image(f32) planes, convolveSeparable rejects Rgba(f32) at comptime
// (convolution.zig:431-435).
// Measured (MI355X, bench-style graph replay over 1 GiB rings, profiles/r05_experiments.txt): one 4096^2 plane 29.3 -> 26.5 us
// (0.63 of 8 TB/s with the launch gap, 0.68 on the kernel's own time), four planes in one launch 116 -> 92.2 us (0.73).
//
// Preconditions (else conv_sep_f32x4.hip / the general kernels run): f32 planes, cols % 4 == 0, strides % 4 == 0, 16-byte aligned
// bases, cols >= 64, rows >= 16, odd equal tap counts 3 / 5 / 7, no tap below the reference's skip threshold (|k| < 1e-10 is
// skipped for interior pixels only, convolution.zig:459-467 — the tiled kernel carries that mask).
#include "zg_common.h"
#include "zg_stream.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#pragma clang fp contract(off)

namespace zg {

constexpr int SF_MAX_PLANES = 8;

template <int N> struct TapsSF { float k[N]; };

struct TileF32Args {
    const uint8_t *src[SF_MAX_PLANES];
    uint8_t *dst[SF_MAX_PLANES];
    uint64_t src_pitch, dst_pitch; // bytes between rows
    int32_t rows, row_bytes;
    int32_t strips_x;              // tiles across a row
    int32_t border;
    uint32_t src_span, dst_span;   // bytes from a plane's first byte to the end of its last row
    int32_t fast_ok;               // both spans fit 32 bits: whole-plane descriptors may be used
};

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// OUTER: only the lanes that keep their halo (the tile's first and last) ask for it — a load's cost in the address path goes by its
// active lanes, and in a batch of planes that pays (four planes 95.6 -> 92.2 us); for a lone plane the branch around the loads costs
// more than it saves (26.5 vs 27.0 us), so one-plane launches let every lane ask (the lanes in between read one shared address).
template <int NK, int R, bool OUTER>
__global__ __launch_bounds__(64) void k_sep_tile_f32(TileF32Args a, TapsSF<NK> kx, TapsSF<NK> ky) {
    constexpr int H = NK / 2;
    constexpr int HB = H;
    constexpr int NR = R + 2 * H; // source rows of a tile
    static_assert(H + 1 <= 4, "the border halo must come out of the outer lane's own four pixels");

    const uint32_t w = blockIdx.x; // address order: tiles left to right, then down
    const int ty = (int)(w / (uint32_t)a.strips_x), tx = (int)(w - (uint32_t)ty * (uint32_t)a.strips_x);
    const uint8_t *srcf = a.src[blockIdx.y];
    uint8_t *dstf = a.dst[blockIdx.y];

    const int lx = (int)threadIdx.x;
    const int rb = a.row_bytes, x0 = tx * 1024;
    const int voff = x0 + 16 * lx;
    const int last_lane = (min(rb - x0, 1024) >> 4) - 1;
    // What the lanes past the row's end load through the whole-plane descriptor (its range check sees the plane, not the row): the last unit again.
    // Their values are never used, but an offset past the row reads the next row — and, in the plane's last row, up to 1 KiB past the plane's end.
    const int voff_ld = min(voff, x0 + 16 * last_lane);
    const bool left_edge = tx == 0, right_edge = x0 + 1024 >= rb;
    const int border = a.border;
    const int off_left = left_edge ? rb - 4 * HB : x0 - 4 * HB, off_right = right_edge ? 0 : x0 + 1024; // at the plane's edges: the other end (.wrap)
    const int y0 = ty * R;

    auto widen = [&](auto edge_tag, const RowIn<HB> &r, float (&q)[12]) {
#pragma unroll
        for (int d = 0; d < 4; ++d) q[4 + d] = u2f(r.v[d]);
#pragma unroll
        for (int d = 0; d < HB; ++d) {
            q[4 - HB + d] = u2f(from_lane_below(r.h[d], r.v[4 - HB + d]));
            q[8 + d] = u2f(from_lane_above(r.h[d], r.v[d]));
        }
        if constexpr (!decltype(edge_tag)::value) return;
        if (last_lane != 63) {
#pragma unroll
            for (int d = 0; d < HB; ++d) q[8 + d] = lx == last_lane ? u2f(r.h[d]) : q[8 + d];
        }
        if (left_edge && border != ZG_BORDER_WRAP) {
#pragma unroll
            for (int d = 0; d < HB; ++d) {
                const float g = border == ZG_BORDER_MIRROR ? u2f(r.v[H - d]) : (border == ZG_BORDER_REPLICATE ? u2f(r.v[0]) : 0.0f);
                q[4 - HB + d] = lx == 0 ? g : q[4 - HB + d];
            }
        }
        if (right_edge && border != ZG_BORDER_WRAP) {
#pragma unroll
            for (int d = 0; d < HB; ++d) {
                const float g = border == ZG_BORDER_MIRROR ? u2f(r.v[2 - d]) : (border == ZG_BORDER_REPLICATE ? u2f(r.v[3]) : 0.0f);
                q[8 + d] = lx == last_lane ? g : q[8 + d];
            }
        }
    };

    const auto src_all = __builtin_amdgcn_make_buffer_rsrc((void *)srcf, (short)0, (int)a.src_span, 0x00020000);
    const auto dst_all = __builtin_amdgcn_make_buffer_rsrc((void *)dstf, (short)0, (int)a.dst_span, 0x00020000);
    const bool fast = a.fast_ok && y0 - H >= 0 && y0 + R + H <= a.rows; // every source and destination row of the tile exists
    const bool full = last_lane == 63;

    auto run = [&](auto fast_tag, auto edge_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        RowIn<HB> in[NR];
        if constexpr (FAST) {
            const uint32_t s0 = (uint32_t)(y0 - H) * (uint32_t)a.src_pitch;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int so = (int)(s0 + (uint32_t)i * (uint32_t)a.src_pitch);
                in[i].v = __builtin_amdgcn_raw_buffer_load_b128(src_all, voff_ld, so, 0);
                if constexpr (!OUTER) HaloLoad<HB>::run(src_all, lx == 0 ? off_left : off_right, so, in[i].h);
            }
            if constexpr (OUTER) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    for (int d = 0; d < HB; ++d) in[i].h[d] = 0;
                if (lx == 0 || lx == last_lane) {
#pragma unroll
                    for (int i = 0; i < NR; ++i) HaloLoad<HB>::run(src_all, lx == 0 ? off_left : off_right, (int)(s0 + (uint32_t)i * (uint32_t)a.src_pitch), in[i].h);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int y = y0 - H + i;
                int gr = y;
                uint32_t keep = ~0u; // 0 for a row the zero border drops
                if (y < 0 || y >= a.rows) { // wave-uniform
                    gr = resolve_row_near(min(y, a.rows - 1 + H), a.rows, border); // rows further out belong to output rows past the plane
                    keep = gr >= 0 ? ~0u : 0u;
                    gr = max(gr, 0);
                }
                const uint8_t *row = srcf + (size_t)(uint32_t)gr * a.src_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)((uint32_t)rb & keep), 0x00020000);
                in[i].v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                HaloLoad<HB>::run(rsrc, lx == 0 ? off_left : off_right, 0, in[i].h);
            }
        }
        auto store_row = [&](u32x4 o, int gy) {
            if constexpr (FAST) {
                const int off = voff + (int)((uint32_t)gy * (uint32_t)a.dst_pitch);
                if (full) st_unit(o, dst_all, off);
                else if (lx <= last_lane) st_unit(o, dst_all, off); // the plane's descriptor does not clip a row
            } else {
                const uint32_t row_ok = (uint32_t)gy < (uint32_t)a.rows ? ~0u : 0u;
                uint8_t *row = dstf + (size_t)((uint32_t)gy & row_ok) * a.dst_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)((uint32_t)rb & row_ok), 0x00020000);
                st_unit(o, rsrc, voff);
            }
        };
        float win[NK][4];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float q[12];
            widen(edge_tag, in[j], q);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const float prod = q[4 - H + p + i] * kx.k[i];
                    acc = acc + prod;
                }
                win[j % NK][p] = acc;
            }
            if (j < 2 * H) continue;
            u32x4 o;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const float prod = win[(j + 1 + i) % NK][p] * ky.k[i];
                    acc = acc + prod;
                }
                o[p] = f2u(acc);
            }
            store_row(o, y0 + j - 2 * H);
        }
    };
    const bool edges = left_edge || right_edge || last_lane != 63;
    if (fast && !edges) run(std::true_type{}, std::false_type{});
    else if (fast) run(std::true_type{}, std::true_type{});
    else run(std::false_type{}, std::true_type{});
}

template <int NK, int R, bool OUTER>
static int launch_tile_f32(const zg_image *src, const zg_image *dst, uint32_t n, const float *fx, const float *fy, int border, hipStream_t s) {
    TapsSF<NK> kx, ky;
    for (int i = 0; i < NK; ++i) { kx.k[i] = fx[i]; ky.k[i] = fy[i]; }
    TileF32Args a{};
    for (uint32_t p = 0; p < SF_MAX_PLANES; ++p) {
        a.src[p] = (const uint8_t *)src[p < n ? p : 0].data;
        a.dst[p] = (uint8_t *)dst[p < n ? p : 0].data;
    }
    a.src_pitch = (uint64_t)src->stride * 4;
    a.dst_pitch = (uint64_t)dst->stride * 4;
    a.rows = (int32_t)src->rows;
    a.row_bytes = (int32_t)(src->cols * 4u);
    a.strips_x = (int32_t)ceil_div((unsigned)a.row_bytes, 1024u);
    const unsigned tiles_y = ceil_div(src->rows, (unsigned)R);
    a.border = border;
    const uint64_t sspan = (uint64_t)(src->rows - 1) * a.src_pitch + (uint64_t)a.row_bytes;
    const uint64_t dspan = (uint64_t)(src->rows - 1) * a.dst_pitch + (uint64_t)a.row_bytes;
    a.fast_ok = sspan <= 0xffffff00u && dspan <= 0xffffff00u;
    a.src_span = (uint32_t)sspan;
    a.dst_span = (uint32_t)dspan;
    const uint64_t items = (uint64_t)a.strips_x * tiles_y;
    if (items > 0x7fffffffu) return -1;
    hipLaunchKernelGGL((k_sep_tile_f32<NK, R, OUTER>), dim3((unsigned)items, n), dim3(64), 0, s, a, kx, ky);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// n <= SF_MAX_PLANES planes of one shape, stride and alignment class in one launch. Returns -1 when the preconditions do not hold
// (the caller falls back to the tiled kernel, plane by plane).
int try_sep_tile_f32(const zg_image *src, const zg_image *dst, uint32_t n, const float *fx, const float *fy, int nk, uint32_t skipx, uint32_t skipy,
                       int border, hipStream_t s) {
    if (n == 0 || n > SF_MAX_PLANES || (nk != 3 && nk != 5 && nk != 7) || (skipx | skipy)) return -1;
    static const bool no_tile = getenv("ZIGNAL_HIP_NO_TILE_F32") != nullptr; // A/B hook of round 5 (the LDS-tiled kernel), read once
    if (no_tile) return -1;
    for (uint32_t p = 0; p < n; ++p) {
        if (src[p].pixel != ZG_PIXEL_F32 || dst[p].pixel != ZG_PIXEL_F32) return -1;
        if (src[p].rows != src->rows || src[p].cols != src->cols || dst[p].rows != src->rows || dst[p].cols != src->cols) return -1;
        if (src[p].stride != src->stride || dst[p].stride != dst->stride) return -1;
        if (((uintptr_t)src[p].data & 15) || ((uintptr_t)dst[p].data & 15)) return -1;
    }
    if (src->cols % 4 || src->stride % 4 || dst->stride % 4) return -1;
    if (src->cols < 64 || src->rows < 16 || (uint64_t)src->cols * 4 > 0x3fffffffu) return -1;
    if ((uint64_t)src->stride * 4 > 0x7fffffffu || (uint64_t)dst->stride * 4 > 0x7fffffffu) return -1;
    if ((src->cols * 4u) % 1024u == 16u) return -1; // the last strip would be one lane wide: that lane is first and last at once
    const bool outer = n > 1; // profiles/r05_experiments.txt section 1: each form is the better one where it is used (the tests reach both through the plane count)
    switch (nk) {
    case 3: return outer ? launch_tile_f32<3, 8, true>(src, dst, n, fx, fy, border, s) : launch_tile_f32<3, 8, false>(src, dst, n, fx, fy, border, s);
    case 5: return outer ? launch_tile_f32<5, 8, true>(src, dst, n, fx, fy, border, s) : launch_tile_f32<5, 8, false>(src, dst, n, fx, fy, border, s);
    case 7: return outer ? launch_tile_f32<7, 8, true>(src, dst, n, fx, fy, border, s) : launch_tile_f32<7, 8, false>(src, dst, n, fx, fy, border, s);
    }
    return -1;
}

} // namespace zg
