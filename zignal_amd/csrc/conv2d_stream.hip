// conv2d_stream.hip — Image(T).convolve with a 3 x 3 or 5 x 5 kernel on u8 pixel types as a register-resident stream: one WAVE walks a
// 1024-byte-wide column strip from top to bottom, nothing is staged in LDS, there is no barrier (the skeleton of conv_sep_stream.hip).
//
// Contract (reference src/image/convolution.zig:76-195, u8 path): taps @round(k * 256) as i32, dst = divClampU8(256) of the exact integer
// sum over all taps, out-of-range taps through border.resolveIndex. Only kernels with 255 * sum|k| < 2^24 come here: every partial sum
// is then an integer an f32 holds exactly, so the sums may be formed in any order and every tap is one v_fmac_f32 (k_conv2d's MODE 3).
//   * a lane owns 16 consecutive bytes of every row (one buffer_load_dwordx4, D rows ahead of the arithmetic), its neighbours' H * SP
//     bytes come across the wave by DPP, the row's ends are synthesised in the wave's outer lanes (zg_stream.h);
//   * every byte is converted to f32 ONCE per source row (v_cvt_f32_ubyteN straight out of the loaded dwords) and feeds the K output
//     rows it belongs to: K x 16 accumulators per lane rotate through K slots, the loop is unrolled over whole turns so that every slot
//     index is a compile-time constant;
//   * a finished row leaves as v_fma(acc, 2^-8, 2^-9) -> v_cvt_pk_u8_f32 per byte: the conversion rounds to nearest even and saturates
//     to [0, 255] (tools/exp/cvt_probe.hip), (a + 0.5) / 256 is never a tie and rounds to floor((a + 128) / 256) for a >= 0 and to 0 for
//     a < 0 — divClampU8(256) — and the instruction packs the byte into its dword on the way: two instructions per output byte.
// The LDS-tiled k_conv2d spent a quarter of its instructions on tile staging and border selects and ran at 0.24 of the HBM roofline;
// this form issues K * K + 3.5 VALU instructions per output byte, all but 2.5 of them at the full rate.
//
// Preconditions (else k_conv2d runs): u8 / Rgb(u8) / Rgba(u8), square kernel of 3 or 5, 255 * sum|round(256 k)| < 2^24, row length and
// pitches multiples of 16 bytes, 16-byte aligned bases, at least 64 pixels per row and 16 rows, spans below 4 GiB for the fast path.
#include "zg_common.h"
#include "zg_u8pack.h"
#include "zg_stream.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace zg {

struct C2StreamArgs {
    const uint8_t *src;
    uint8_t *dst;
    uint64_t src_pitch, dst_pitch; // bytes between rows
    int32_t rows, row_bytes;
    int32_t strips_x, strips_y;
    int32_t strip_rows;
    int32_t border;
    uint32_t src_span, dst_span; // bytes from the first byte to the end of the last row
    int32_t fast_ok;             // both spans fit 32 bits: whole-image descriptors may be used
};
#ifndef C2S_PLAIN_FMA
#define C2S_PLAIN_FMA 0 // 1: v_fmac_f32 per byte for every pixel stride (the form before round 5; tools/build_variant.sh for A/B)
#endif
#ifndef C2S_WAVES
#define C2S_WAVES (K == 3 ? 4 : 2)
#endif
template <int K> struct TapsF2D { float w[K * K]; }; // round(k * 256), integer-valued

template <int B> __device__ __forceinline__ float byte_to_f32(uint32_t dword) { return (float)((dword >> (8 * B)) & 0xffu); } // v_cvt_f32_ubyteB

template <int SP, int K, int DM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(C2S_WAVES))) void k_conv2d_stream(C2StreamArgs a, TapsF2D<K> k) {
    constexpr int H = K / 2;
    constexpr int HB = (H * SP + 3) / 4; // halo dwords per side
    constexpr int NP = 16 + 2 * H * SP;  // byte positions a lane converts per row
    constexpr int D = K * DM;            // source rows in flight ahead of the arithmetic = rows per unrolled block
    static_assert((H + 1) * SP <= 16, "the border halo must come out of the outer lane's own unit");

    const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3;
    uint32_t w = blockIdx.x;
    if (ZG_XCD_ORDER && w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3); // XCD-major: an XCD's L2 sees whole bands of neighbouring strips
    const int sy = (int)(w / (uint32_t)a.strips_x), sx = (int)(w - (uint32_t)sy * (uint32_t)a.strips_x);

    const int lx = (int)threadIdx.x;
    const int rb = a.row_bytes, x0 = sx * 1024;
    const int voff = x0 + 16 * lx;
    const int last_lane = (min(rb - x0, 1024) >> 4) - 1;
    const bool left_edge = sx == 0, right_edge = x0 + 1024 >= rb;
    const int border = a.border;
    const int off_h = lx == 0 ? (left_edge ? rb - 4 * HB : x0 - 4 * HB) : (right_edge ? 0 : x0 + 1024); // always inside the row; at the ends: what .wrap wants
    const int y0 = sy * a.strip_rows;
    const int out_rows = min(y0 + a.strip_rows, a.rows) - y0;
    const int n_in = (out_rows + 2 * H + D - 1) / D * D; // whole blocks: one loop exit (see conv_sep_stream.hip)

    // the twelve dwords around a lane's unit: [4 - HB, 4) from the lane below, [4, 8) own, [8, 8 + HB) from the lane above
    auto widen = [&](auto edge_tag, const RowIn<HB> &r, uint32_t (&q)[12]) {
#pragma unroll
        for (int d = 0; d < 12; ++d) q[d] = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) q[4 + d] = r.v[d];
#pragma unroll
        for (int d = 0; d < HB; ++d) {
            q[4 - HB + d] = from_lane_below(r.h[d], r.v[4 - HB + d]);
            q[8 + d] = from_lane_above(r.h[d], r.v[d]);
        }
        if (decltype(edge_tag)::value && last_lane != 63) { // wave-uniform: the row ends inside this strip
#pragma unroll
            for (int d = 0; d < HB; ++d) q[8 + d] = lx == last_lane ? r.h[d] : q[8 + d];
        }
        if constexpr (!decltype(edge_tag)::value) return;
        if (left_edge && border != ZG_BORDER_WRAP) {
            uint32_t g[HB];
            if (border == ZG_BORDER_MIRROR) synth_halo<SP, H, HB, false, true>(r.v, g);
            else if (border == ZG_BORDER_REPLICATE) synth_halo<SP, H, HB, false, false>(r.v, g);
            else {
#pragma unroll
                for (int d = 0; d < HB; ++d) g[d] = 0;
            }
#pragma unroll
            for (int d = 0; d < HB; ++d) q[4 - HB + d] = lx == 0 ? g[d] : q[4 - HB + d];
        }
        if (right_edge && border != ZG_BORDER_WRAP) {
            uint32_t g[HB];
            if (border == ZG_BORDER_MIRROR) synth_halo<SP, H, HB, true, true>(r.v, g);
            else if (border == ZG_BORDER_REPLICATE) synth_halo<SP, H, HB, true, false>(r.v, g);
            else {
#pragma unroll
                for (int d = 0; d < HB; ++d) g[d] = 0;
            }
#pragma unroll
            for (int d = 0; d < HB; ++d) q[8 + d] = lx == last_lane ? g[d] : q[8 + d];
        }
    };

    const auto src_all = __builtin_amdgcn_make_buffer_rsrc((void *)a.src, (short)0, (int)a.src_span, 0x00020000);
    const auto dst_all = __builtin_amdgcn_make_buffer_rsrc((void *)a.dst, (short)0, (int)a.dst_span, 0x00020000);
    const bool fast = a.fast_ok && y0 - H >= 0 && y0 - H + n_in + D <= a.rows; // every row this strip touches, read-ahead included, is a real row
    const bool full = last_lane == 63;

    auto run = [&](auto fast_tag, auto edge_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        uint32_t s_next = (uint32_t)(y0 - H) * (uint32_t)a.src_pitch; // FAST: byte offset of the next source row to ask for
        uint32_t d_next = (uint32_t)y0 * (uint32_t)a.dst_pitch;       // FAST: ... of the next destination row
        auto load_row = [&](int y) -> RowIn<HB> {
            RowIn<HB> r;
            if constexpr (FAST) { // rows are asked for in ascending order
                r.v = __builtin_amdgcn_raw_buffer_load_b128(src_all, voff, (int)s_next, 0);
                HaloLoad<HB>::run(src_all, off_h, (int)s_next, r.h);
                s_next += (uint32_t)a.src_pitch;
            } else {
                int gr = y;
                uint32_t keep = ~0u; // 0 for a row the zero border drops
                if (y < 0 || y >= a.rows) { // wave-uniform
                    gr = resolve_row_near(min(y, a.rows - 1 + H), a.rows, border); // rows past that are read ahead and never used
                    keep = gr >= 0 ? ~0u : 0u;
                    gr = max(gr, 0);
                }
                const uint8_t *row = a.src + (size_t)(uint32_t)gr * a.src_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)((uint32_t)rb & keep), 0x00020000);
                r.v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                HaloLoad<HB>::run(rsrc, off_h, 0, r.h);
            }
            return r;
        };
        auto store_row = [&](u32x4 o, int gy) {
            if constexpr (FAST) {
                if (full) st_unit(o, dst_all, voff + (int)d_next);
                else if (lx <= last_lane) st_unit(o, dst_all, voff + (int)d_next); // the image's descriptor does not clip a row
                d_next += (uint32_t)a.dst_pitch;
            } else {
                const uint32_t row_ok = (uint32_t)gy < (uint32_t)a.rows ? ~0u : 0u;
                uint8_t *row = a.dst + (size_t)((uint32_t)gy & row_ok) * a.dst_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)((uint32_t)rb & row_ok), 0x00020000);
                st_unit(o, rsrc, voff); // row bytes % 16 == 0: a unit is all in or all out
            }
        };

        RowIn<HB> ahead[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ahead[i] = load_row(y0 - H + i);
            __builtin_amdgcn_sched_barrier(0); // requests in row order: a wave's loads return in order
        }
        // acc[slot][byte]: output row m of the strip lives in slot m % K from the source row that brings its first kernel row (q = m) to
        // the one that brings its last (q = m + K - 1); source row q = qb + u of the strip is row y0 - H + q of the image.
        // Even pixel strides (Rgba): two bytes per v_pk_fma_f32, the weight broadcast from one half of a scalar-register PAIR (op_sel). A packed
        // multiply-add issues every 4.5 cycles whatever the multiplier's register file, a plain v_fmac_f32 with an SGPR operand every 4.1 — for ONE
        // byte (tools/exp/pk_sgpr.hip, fmac_sgpr.hip) — and the weights keep costing no vector registers.
        constexpr bool PACKED = SP % 2 == 0 && !C2S_PLAIN_FMA;
        typedef float f2 __attribute__((ext_vector_type(2)));
        uint64_t wpair[(K * K + 1) / 2];
        if constexpr (PACKED) {
#pragma unroll
            for (int i = 0; i < (K * K + 1) / 2; ++i)
                wpair[i] = (uint64_t)__float_as_uint(k.w[2 * i]) | ((uint64_t)(2 * i + 1 < K * K ? __float_as_uint(k.w[2 * i + 1]) : 0u) << 32);
        }
        auto pk_mul = [&](f2 p, int wi) -> f2 {
            f2 r;
            if (wi & 1) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "s"(wpair[wi >> 1]), "v"(p));
            else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "s"(wpair[wi >> 1]), "v"(p));
            return r;
        };
        auto pk_fma = [&](f2 p, int wi, f2 c) -> f2 {
            if (wi & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "s"(wpair[wi >> 1]), "v"(p));
            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "s"(wpair[wi >> 1]), "v"(p));
            return c;
        };
        float acc[K][16];
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[s][t] = 0.0f;
        for (int qb = 0; qb < n_in; qb += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int q = qb + u;
                uint32_t qd[12];
                widen(edge_tag, ahead[u], qd);
                float P[NP]; // position p is byte p - H * SP of the lane's unit = byte 16 + p - H * SP of qd
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    constexpr int base = 16 - H * SP;
                    const int b = base + p;
                    const uint32_t dw = qd[b >> 2];
                    P[p] = (b & 3) == 0 ? byte_to_f32<0>(dw) : (b & 3) == 1 ? byte_to_f32<1>(dw) : (b & 3) == 2 ? byte_to_f32<2>(dw) : byte_to_f32<3>(dw);
                }
                ahead[u] = load_row(y0 - H + q + D); // the slot's registers are dead: the row D ahead lands in them
#pragma unroll
                for (int ky = 0; ky < K; ++ky) { // this source row is kernel row ky of output row m = q - ky
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int s = ((u - ky) % K + K) % K;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        if constexpr (PACKED) {
#pragma unroll
                            for (int t = 0; t < 16; t += 2) {
                                const f2 pp = f2{P[t + SP * kx], P[t + 1 + SP * kx]};
                                f2 c = f2{acc[s][t], acc[s][t + 1]};
                                c = ky == 0 && kx == 0 ? pk_mul(pp, 0) : pk_fma(pp, ky * K + kx, c);
                                acc[s][t] = c.x;
                                acc[s][t + 1] = c.y;
                            }
                        } else {
                            const float wt = k.w[ky * K + kx];
#pragma unroll
                            for (int t = 0; t < 16; ++t) {
                                if (ky == 0 && kx == 0) acc[s][t] = P[t] * wt; // the row's first tap starts the slot afresh
                                else acc[s][t] = __builtin_fmaf(P[t + SP * kx], wt, acc[s][t]); // integers below 2^24: exact, any order
                            }
                        }
                    }
                }
                if (q < K - 1) continue; // wave-uniform: the strip's first K - 1 rows complete no output row of this strip
                const int sdone = ((u + 1) % K + K) % K; // output row m = q - (K - 1)
                u32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t pk = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(acc[sdone][4 * d + b], 0x1p-8f, 0x1p-9f), (uint32_t)b, pk);
                    o[d] = pk;
                }
                store_row(o, y0 + q - (K - 1));
            }
        }
    };
    const bool edges = left_edge || right_edge || last_lane != 63;
    if (fast && !edges) run(std::true_type{}, std::false_type{}); // wave-uniform: nearly every strip
    else if (fast) run(std::true_type{}, std::true_type{});
    else run(std::false_type{}, std::true_type{});
}

static int c2s_strip_rows(uint32_t rows, uint32_t strips_x, int k, int d) {
    int r;
    {
        const uint64_t all_rows = (uint64_t)rows * strips_x;
        r = (int)std::min<uint64_t>(std::max<uint64_t>((all_rows + 4095) / 4096, 16), 64); // ~four waves per SIMD of the chip when one image has to fill it
    }
    while ((r + k - 1) % d) ++r; // (strip rows + 2H) % D == 0: no padded row at the end of a strip
    return r;
}

template <int SP, int K>
static int launch_c2s(const zg_image *src, const zg_image *dst, const float *taps, int border, hipStream_t s) {
    TapsF2D<K> k;
    for (int i = 0; i < K * K; ++i) k.w[i] = taps[i];
    C2StreamArgs a;
    a.src = (const uint8_t *)src->data;
    a.dst = (uint8_t *)dst->data;
    a.src_pitch = src->stride * (size_t)SP;
    a.dst_pitch = dst->stride * (size_t)SP;
    a.rows = (int32_t)src->rows;
    a.row_bytes = (int32_t)(src->cols * (uint32_t)SP);
    a.strips_x = (int32_t)ceil_div((unsigned)a.row_bytes, 1024u);
    a.strip_rows = c2s_strip_rows(src->rows, (uint32_t)a.strips_x, K, K);
    a.strips_y = (int32_t)ceil_div(src->rows, (unsigned)a.strip_rows);
    a.border = border;
    const uint64_t sspan = (uint64_t)(src->rows - 1) * a.src_pitch + (uint64_t)a.row_bytes, dspan = (uint64_t)(src->rows - 1) * a.dst_pitch + (uint64_t)a.row_bytes;
    a.fast_ok = sspan <= 0xffffffffu && dspan <= 0xffffffffu;
    a.src_span = (uint32_t)sspan;
    a.dst_span = (uint32_t)dspan;
    const uint64_t items = (uint64_t)a.strips_x * a.strips_y;
    if (items > 0x7fffffffu) return -1;
    hipLaunchKernelGGL((k_conv2d_stream<SP, K, 1>), dim3((unsigned)items), dim3(64), 0, s, a, k);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// taps: round(k * 256) as floats, kh x kw. Returns -1 when the preconditions do not hold (the caller runs k_conv2d).
int try_conv2d_stream(const zg_image *src, const zg_image *dst, const float *taps, int kh, int kw, int border, hipStream_t s) {
    if (kh != kw || (kh != 3 && kh != 5)) return -1; // 7 x 7 wants 261 to 373 registers in this form (one wave per SIMD): it stays on k_conv2d
    const int sp = (int)pixel_size(src->pixel);
    if (pixel_is_float(src->pixel) || (sp != 1 && sp != 3 && sp != 4)) return -1;
    if ((kh / 2 + 1) * sp > 16) return -1;
    const uint64_t rb = (uint64_t)src->cols * (uint64_t)sp, sp_pitch = (uint64_t)src->stride * sp, dp_pitch = (uint64_t)dst->stride * sp;
    if (rb % 16 || sp_pitch % 16 || dp_pitch % 16 || ((uintptr_t)src->data & 15) || ((uintptr_t)dst->data & 15)) return -1;
    if (rb % 1024 == 16) return -1; // the last strip would be one lane wide: that lane is first and last at once
    if (src->cols < 64 || src->rows < 16 || rb > 0x3fffffffu || sp_pitch > 0x7fffffffu || dp_pitch > 0x7fffffffu) return -1;
#define ZG_C2S(SP, K) if (sp == SP && kh == K) return launch_c2s<SP, K>(src, dst, taps, border, s);
    ZG_C2S(1, 3) ZG_C2S(1, 5) ZG_C2S(3, 3) ZG_C2S(3, 5) ZG_C2S(4, 3) ZG_C2S(4, 5)
#undef ZG_C2S
    return -1;
}

} // namespace zg
