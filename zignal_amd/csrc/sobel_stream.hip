// sobel_stream.hip — Image(u8).sobel and Image(Rgba(u8)).sobel as a register-resident stream: one WAVE walks a 1024-byte-wide column
// strip from top to bottom (the skeleton of conv_sep_stream.hip / conv2d_stream.hip), no LDS, no barrier.
//
// Contract (reference src/image/edges.zig:33-70): grey = as(f32, convertColor(u8, px)); gx, gy = the 3 x 3 Sobel sums of the grey plane
// with .replicate borders, in f32; out = @trunc(@max(0, @min(255, @sqrt(gx * gx + gy * gy) / 4))).
// Everything up to the square root is small integers in f32 — grey <= 255, |gx|, |gy| <= 1020, gx^2 + gy^2 <= 2 080 800 < 2^24 — so it is
// exact in any order and this kernel is free to
//   * take Rgba(u8) -> grey as three v_fma_f32 on bytes converted by v_cvt_f32_ubyteN: the BT.709 16.16 sum
//     (13933 R + 46871 G + 4732 B + 32768) <= 16 744 448 is exact in f32, and >> 16 is a multiplication by 2^-16 and a floor
//     (src/color.zig:1031-1042);
//   * form the Sobel sums separably: s = g[r-1] + 2 g[r] + g[r+1] and d = g[r+1] - g[r-1] per column, gx = s[x+1] - s[x-1],
//     gy = d[x-1] + 2 d[x] + d[x+1] (the reference's term order gives the same integers; a zero's sign disappears in the squares);
//   * replace the correctly rounded square root: the result byte is floor(sqrt(m) / 4) for an integer m, and sqrt(m) is never closer
//     than 1 / (8 * 255) = 4.9e-4 below a multiple of 4 unless it IS one; v_sqrt_f32 is good to one ulp (6.1e-5 below 1024), so
//     floor((v_sqrt_f32(m) + 2^-12) / 4) is that byte for every m (checked for all 2 080 801 values: tools/exp/sqrt_probe.hip,
//     profiles/r04_experiments.txt), and v_cvt_pk_u8_f32 saturates it at 255 and packs it.
// A lane owns 16 bytes of every source row: 4 Rgba pixels -> 4 output bytes (one dword store per lane), or 16 grey pixels -> 16
// output bytes. The neighbours' border columns cross the wave as grey FLOATS by DPP; the strip's outer columns come from a narrow
// second load that arrives as the DPP move's `old` operand (zg_stream.h).
//
// Preconditions (else k_sobel runs): source u8 or Rgba(u8), row bytes a multiple of 16, 16-byte aligned rows on both sides of the
// Rgba form (4-byte aligned destination rows), at least 64 pixels per row and 16 rows.
#include "zg_common.h"
#include "zg_u8pack.h"
#include "zg_stream.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace zg {

struct SobelStreamArgs {
    const uint8_t *src;
    uint8_t *dst;
    uint64_t src_pitch, dst_pitch; // bytes between rows
    uint64_t src_frame, dst_frame; // bytes between frames (blockIdx.y)
    int32_t rows, row_bytes;       // source rows, bytes per source row
    int32_t strips_x, strips_y;
    int32_t strip_rows;
    uint32_t src_span, dst_span;
    int32_t fast_ok;
};

template <int B> __device__ __forceinline__ float ubyte_f32(uint32_t dword) { return (float)((dword >> (8 * B)) & 0xffu); } // v_cvt_f32_ubyteB
__device__ __forceinline__ float dpp_below(float old, float v) { return __builtin_bit_cast(float, from_lane_below(__builtin_bit_cast(uint32_t, old), __builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ float dpp_above(float old, float v) { return __builtin_bit_cast(float, from_lane_above(__builtin_bit_cast(uint32_t, old), __builtin_bit_cast(uint32_t, v))); }

// convertColor(u8, Rgba(u8)) as a float (color.zig:1031-1042): exact in f32, see the file comment
__device__ __forceinline__ float grey_of_rgba(uint32_t px) {
    float acc = __builtin_fmaf(ubyte_f32<0>(px), 13933.0f, 32768.0f);
    acc = __builtin_fmaf(ubyte_f32<1>(px), 46871.0f, acc);
    acc = __builtin_fmaf(ubyte_f32<2>(px), 4732.0f, acc);
    return __builtin_floorf(acc * 0x1p-16f);
}
// @trunc(@max(0, @min(255, @sqrt(m) / 4))) for an integer m in [0, 2^24) as a float, packed into byte `b` of `old`
__device__ __forceinline__ uint32_t sobel_byte(float m, uint32_t b, uint32_t old) {
    const float q = __builtin_floorf(__builtin_fmaf(__builtin_amdgcn_sqrtf(m), 0.25f, 0x1p-14f));
    return __builtin_amdgcn_cvt_pk_u8_f32(q, b, old);
}

template <int SP, int DM>
__global__ __launch_bounds__(64) void k_sobel_stream(SobelStreamArgs a) {
    static_assert(SP == 1 || SP == 4, "grey or Rgba(u8) sources");
    constexpr int NPX = 16 / SP;  // pixels a lane owns per row
    constexpr int HB = 1;         // one halo dword per side: one Rgba pixel, or the grey pixel in its byte 3 / byte 0
    constexpr int D = 3 * DM;     // source rows in flight = rows per unrolled block
    constexpr int NG = NPX + 2;   // grey columns a lane holds per row: x - 1 .. x + NPX

    const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3;
    uint32_t w = blockIdx.x;
    if (ZG_XCD_ORDER && w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3);
    const int sy = (int)(w / (uint32_t)a.strips_x), sx = (int)(w - (uint32_t)sy * (uint32_t)a.strips_x);
    const uint8_t *srcf = a.src + (size_t)blockIdx.y * a.src_frame;
    uint8_t *dstf = a.dst + (size_t)blockIdx.y * a.dst_frame;

    const int lx = (int)threadIdx.x;
    const int rb = a.row_bytes, x0 = sx * 1024;
    const int voff = x0 + 16 * lx;
    const int dvoff = voff / SP; // one output byte per source pixel
    const int last_lane = (min(rb - x0, 1024) >> 4) - 1;
    const bool left_edge = sx == 0, right_edge = x0 + 1024 >= rb;
    // lane 0: the dword before the strip, every other lane: the dword after it (the last lane's); clamped into the row at the image's ends,
    // where .replicate takes the lane's own outer pixel instead
    const int off_h = lx == 0 ? (left_edge ? 0 : x0 - 4) : (right_edge ? rb - 4 : x0 + 1024);
    const int y0 = sy * a.strip_rows;
    const int out_rows = min(y0 + a.strip_rows, a.rows) - y0;
    const int n_in = (out_rows + 2 + D - 1) / D * D;

    const uint32_t dst_rb = (uint32_t)rb / SP;
    const auto src_all = __builtin_amdgcn_make_buffer_rsrc((void *)srcf, (short)0, (int)a.src_span, 0x00020000);
    const auto dst_all = __builtin_amdgcn_make_buffer_rsrc((void *)dstf, (short)0, (int)a.dst_span, 0x00020000);
    const bool fast = a.fast_ok && y0 - 1 >= 0 && y0 - 1 + n_in + D <= a.rows;
    const bool full = last_lane == 63;

    auto run = [&](auto fast_tag, auto edge_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr bool EDGE = decltype(edge_tag)::value;
        uint32_t s_next = (uint32_t)(y0 - 1) * (uint32_t)a.src_pitch;
        uint32_t d_next = (uint32_t)y0 * (uint32_t)a.dst_pitch;
        auto load_row = [&](int y) -> RowIn<HB> {
            RowIn<HB> r;
            if constexpr (FAST) {
                r.v = __builtin_amdgcn_raw_buffer_load_b128(src_all, voff, (int)s_next, 0);
                HaloLoad<HB>::run(src_all, off_h, (int)s_next, r.h);
                s_next += (uint32_t)a.src_pitch;
            } else {
                const int gr = min(max(y, 0), a.rows - 1); // .replicate
                const uint8_t *row = srcf + (size_t)(uint32_t)gr * a.src_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, rb, 0x00020000);
                r.v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                HaloLoad<HB>::run(rsrc, off_h, 0, r.h);
            }
            return r;
        };
        auto store_row = [&](auto o, int gy) {
            if constexpr (FAST) {
                if (full) st_unit(o, dst_all, dvoff + (int)d_next);
                else if (lx <= last_lane) st_unit(o, dst_all, dvoff + (int)d_next);
                d_next += (uint32_t)a.dst_pitch;
            } else {
                const uint32_t row_ok = (uint32_t)gy < (uint32_t)a.rows ? ~0u : 0u;
                uint8_t *row = dstf + (size_t)((uint32_t)gy & row_ok) * a.dst_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)(dst_rb & row_ok), 0x00020000);
                st_unit(o, rsrc, dvoff); // a lane's unit is all inside the row or all outside it
            }
        };
        // grey columns x - 1 .. x + NPX of one source row
        auto grey_row = [&](const RowIn<HB> &r, float (&g)[NG]) {
            float gh; // the strip's outer column this lane can supply: lane 0 its left one, the others the right one
            if constexpr (SP == 4) {
#pragma unroll
                for (int p = 0; p < 4; ++p) g[1 + p] = grey_of_rgba(r.v[p]);
                gh = grey_of_rgba(r.h[0]);
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    g[1 + 4 * d] = ubyte_f32<0>(r.v[d]); g[2 + 4 * d] = ubyte_f32<1>(r.v[d]); g[3 + 4 * d] = ubyte_f32<2>(r.v[d]); g[4 + 4 * d] = ubyte_f32<3>(r.v[d]);
                }
                gh = lx == 0 ? ubyte_f32<3>(r.h[0]) : ubyte_f32<0>(r.h[0]);
            }
            g[0] = dpp_below(gh, g[NPX]);
            g[NG - 1] = dpp_above(gh, g[1]);
            if constexpr (EDGE) { // .replicate at the row's ends; a row that ends inside the strip
                if (left_edge) g[0] = lx == 0 ? g[1] : g[0];
                if (right_edge) g[NG - 1] = lx == last_lane ? g[NPX] : g[NG - 1];
            }
        };

        RowIn<HB> ahead[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ahead[i] = load_row(y0 - 1 + i);
            __builtin_amdgcn_sched_barrier(0);
        }
        float G[3][NG]; // grey rows, source row q of the strip (row y0 - 1 + q of the image) in slot q % 3
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < NG; ++t) G[s][t] = 0.0f;
        for (int qb = 0; qb < n_in; qb += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int q = qb + u;
                grey_row(ahead[u], G[u % 3]);
                ahead[u] = load_row(y0 - 1 + q + D);
                if (q < 2) continue; // wave-uniform
                const float(&top)[NG] = G[(u + 1) % 3], (&mid)[NG] = G[(u + 2) % 3], (&bot)[NG] = G[u % 3]; // rows q - 2, q - 1, q
                float sm[NG], df[NG];
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    sm[t] = __builtin_fmaf(mid[t], 2.0f, top[t]) + bot[t];
                    df[t] = bot[t] - top[t];
                }
                uint32_t o[NPX / 4];
#pragma unroll
                for (int d = 0; d < NPX / 4; ++d) {
                    uint32_t pk = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int t = 1 + 4 * d + b; // column x + 4 d + b
                        const float gx = sm[t + 1] - sm[t - 1];
                        const float gy = __builtin_fmaf(df[t], 2.0f, df[t - 1]) + df[t + 1];
                        pk = sobel_byte(__builtin_fmaf(gx, gx, gy * gy), (uint32_t)b, pk);
                    }
                    o[d] = pk;
                }
                if constexpr (SP == 4) store_row(o[0], y0 + q - 2);
                else store_row(u32x4{o[0], o[1], o[2], o[3]}, y0 + q - 2);
            }
        }
    };
    const bool edges = left_edge || right_edge || last_lane != 63;
    if (fast && !edges) run(std::true_type{}, std::false_type{});
    else if (fast) run(std::true_type{}, std::true_type{});
    else run(std::false_type{}, std::true_type{});
}

template <int SP>
static int launch_sobel_stream(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    SobelStreamArgs a;
    a.src = (const uint8_t *)src->data;
    a.dst = (uint8_t *)dst->data;
    a.src_pitch = src->stride * (size_t)SP;
    a.dst_pitch = dst->stride;
    a.src_frame = src_frame;
    a.dst_frame = dst_frame;
    a.rows = (int32_t)src->rows;
    a.row_bytes = (int32_t)(src->cols * (uint32_t)SP);
    a.strips_x = (int32_t)ceil_div((unsigned)a.row_bytes, 1024u);
    int r = (int)std::min<uint64_t>(std::max<uint64_t>(((uint64_t)src->rows * a.strips_x * n + 4095) / 4096, 16), 64);
    while ((r + 2) % 3) ++r; // (strip rows + 2) % D == 0
    a.strip_rows = r;
    a.strips_y = (int32_t)ceil_div(src->rows, (unsigned)a.strip_rows);
    const uint64_t sspan = (uint64_t)(src->rows - 1) * a.src_pitch + (uint64_t)a.row_bytes, dspan = (uint64_t)(src->rows - 1) * a.dst_pitch + src->cols;
    a.fast_ok = sspan <= 0xffffffffu && dspan <= 0xffffffffu;
    a.src_span = (uint32_t)sspan;
    a.dst_span = (uint32_t)dspan;
    const uint64_t items = (uint64_t)a.strips_x * a.strips_y;
    if (items > 0x7fffffffu || n > MAX_FRAMES_PER_LAUNCH) return -1;
    hipLaunchKernelGGL((k_sobel_stream<SP, 1>), dim3((unsigned)items, n), dim3(64), 0, s, a);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Returns -1 when the preconditions do not hold (the caller runs k_sobel).
int try_sobel_stream(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_U8 && src->pixel != ZG_PIXEL_RGBA_U8) return -1;
    const int sp = src->pixel == ZG_PIXEL_U8 ? 1 : 4;
    const uint64_t rb = (uint64_t)src->cols * (uint64_t)sp, sp_pitch = (uint64_t)src->stride * sp;
    const unsigned dalign = sp == 1 ? 16 : 4; // the lane's output unit
    if (rb % 16 || sp_pitch % 16 || src_frame % 16 || ((uintptr_t)src->data & 15)) return -1;
    if (dst->stride % dalign || dst_frame % dalign || ((uintptr_t)dst->data & (dalign - 1))) return -1;
    if (rb % 1024 == 16) return -1; // the last strip would be one lane wide
    if (src->cols < 64 || src->rows < 16 || rb > 0x3fffffffu || sp_pitch > 0x7fffffffu || dst->stride > 0x7fffffffu) return -1;
    return sp == 1 ? launch_sobel_stream<1>(src, dst, n, src_frame, dst_frame, s) : launch_sobel_stream<4>(src, dst, n, src_frame, dst_frame, s);
}

} // namespace zg
