// conv_sep_stream.hip — the u8 separable convolution as a register-resident stream: one WAVE walks a 1024-byte-wide column
// strip of the image from top to bottom; nothing is staged in LDS and there is no barrier anywhere.
//
// Same arithmetic contract as conv_sep_bytes.hip (reference src/image/convolution.zig:441-647, u8 path; taps round(k * 256)
// in [0, 255] with sum <= 257, every channel the same taps, so a row of SP-byte pixels is a plain byte stream in which tap i
// of byte b is byte b + SP * (i - H)). What differs is where the data lives:
//   * a lane owns 16 consecutive bytes of the strip and loads them once per source row (one buffer_load_dwordx4, D row
//     pairs ahead of the arithmetic; the row's descriptor carries the row length, so lanes past the row and rows the zero
//     border drops read 0 without a branch);
//   * the H * SP bytes a lane needs from each neighbour come across the wave with DPP wave shifts (wave_shr:1 / wave_shl:1),
//     the two outer lanes of the wave take theirs from a second, narrow load (lane 0 the 4 * HB bytes before the strip, every
//     other lane the 4 * HB bytes after it: two cache lines per row) that arrives as the DPP move's `old` operand — a wave
//     shift has no source for the outer lane, which then keeps `old` — so the halo costs no instruction of its own;
//   * the row pass (packed u16 on row pairs, one v_perm per position, one v_pk_mad_u16 per byte and tap for two rows) fills a
//     sliding window of H + 1 row pairs in registers and the column pass (v_dot2_u32_u16, two taps per instruction) reads it;
//     only the first 2H rows of a strip are convolved twice (by this strip and the one above), against 2H rows of every
//     4 * RPT in the LDS-tiled kernel (half of the row pass at RPT = 8);
//   * the image's left / right border is synthesised in the two outer lanes from their own 16 bytes (mirror, replicate: byte
//     permutes with compile-time selectors; wrap: the scalar load points at the other end of the row; zero: nothing to do).
// Waves share nothing, so they drift apart freely: loads, arithmetic and stores of different waves overlap instead of
// meeting at workgroup barriers, which is what kept the tiled kernel at the sum of its HBM and VALU times.
//
// One launch covers a batch of frames laid out back to back (frame index in the work-item number), optionally with the 2:1
// bilinear resize of the `pipeline` recipe [blur, resize x0.5] fused behind the blur for Rgba(u8) (DOWN2; see
// conv_sep_rgba8.hip for the identity (tl + tr + bl + br) >> 2 that makes the fusion exact).
//
// Preconditions (else the tiled kernels run): u8 / Rgb(u8) / Rgba(u8), row length and pitches multiples of 16 bytes,
// 16-byte aligned bases, odd equal tap counts with (H + 1) * SP <= 16, taps as above, at least 64 pixels per row and 16 rows.
#include "zg_common.h"
#include "zg_u8pack.h"
#include "zg_stream.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace zg {

struct StreamArgs {
    const uint8_t *src;
    uint8_t *dst;
    uint64_t src_pitch, dst_pitch; // bytes between rows
    uint64_t src_frame, dst_frame; // bytes between frames
    int32_t rows, row_bytes;       // source rows, bytes per source row
    int32_t strips_x, strips_y;    // work items per frame
    int32_t strip_rows;            // source rows per strip (even)
    int32_t border;
    uint32_t src_span, dst_span;   // bytes from a frame's first byte to the end of its last row
    int32_t fast_ok;               // both spans fit 32 bits: whole-frame descriptors may be used
#ifdef ZG_STREAM_TRACE
    unsigned long long *trace;     // tools/exp/stream_trace.hip: per wave {start, end} of the 100 MHz clock + shader cycles
#endif
};

template <int S> __device__ __forceinline__ u16x2 row_pair_s(const uint32_t (&q0)[12], const uint32_t (&q1)[12]) {
    constexpr int d = S >> 2, o = S & 3;
    constexpr uint32_t sel = 0x0c000c00u | ((4u + o) << 16) | (uint32_t)o; // byte o of q0[d] -> low half, byte o of q1[d] -> high half
    return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q1[d], q0[d], sel));
}
template <int S0, int N, int I = 0> struct UnpackRowsS { // P[I] = position S0 + I, I < N
    __device__ static __forceinline__ void run(const uint32_t (&q0)[12], const uint32_t (&q1)[12], u16x2 (&P)[N]) {
        if constexpr (I < N) {
            P[I] = row_pair_s<S0 + I>(q0, q1);
            UnpackRowsS<S0, N, I + 1>::run(q0, q1, P);
        }
    }
};
template <int SP, int NK, int N, int T, int I> struct RowTapsS {
    __device__ static __forceinline__ u16x2 run(const u16x2 (&P)[N], const TapsU8<NK> &kx, u16x2 acc) {
        if constexpr (I == NK) return acc;
        else {
            const uint16_t k = (uint16_t)kx.k[I];
            const u16x2 kk = {k, k};
            acc += P[T + SP * I] * kk; // <= 65535 by the preconditions: exact
            return RowTapsS<SP, NK, N, T, I + 1>::run(P, kx, acc);
        }
    }
};
template <int SP, int NK, int N, int T> struct RowBytesS {
    __device__ static __forceinline__ void run(const u16x2 (&P)[N], const TapsU8<NK> &kx, u16x2 (&out)[16]) {
        if constexpr (T < 16) {
            out[T] = RowTapsS<SP, NK, N, T, 0>::run(P, kx, u16x2{0, 0});
            RowBytesS<SP, NK, N, T + 1>::run(P, kx, out);
        }
    }
};
template <int SP, int NK, int N> __device__ __forceinline__ void row_pass_tap_major(const u16x2 (&P)[N], const TapsU8<NK> &kx, u16x2 (&out)[16]) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const uint16_t k = (uint16_t)kx.k[i];
        const u16x2 kk = {k, k};
#pragma unroll
        for (int t = 0; t < 16; ++t) out[t] = i == 0 ? P[t] * kk : out[t] + P[t + SP * i] * kk; // <= 65535 by the preconditions: exact
    }
}
// Symmetric taps with unit ends (gaussianBlur(0.6) is [1, 42, 170, 42, 1], the metric's own kernel): the mirrored positions are added
// first — bytes, so the sums stay far inside 16 bits — and the end pair needs no multiply at all: four packed operations per byte pair
// and row pair instead of five (5 taps), two instead of three (3 taps), six instead of seven (7 taps). Integer sums: same value, bit for bit.
template <int SP, int NK, int N> __device__ __forceinline__ void row_pass_unit_sym(const u16x2 (&P)[N], const TapsU8<NK> &kx, u16x2 (&out)[16]) {
    constexpr int H = NK / 2;
#pragma unroll
    for (int t = 0; t < 16; ++t) out[t] = P[t] + P[t + SP * (NK - 1)];
#pragma unroll
    for (int i = 1; i < H; ++i) {
        const uint16_t k = (uint16_t)kx.k[i];
        const u16x2 kk = {k, k};
        u16x2 s[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = P[t + SP * i] + P[t + SP * (NK - 1 - i)];
#pragma unroll
        for (int t = 0; t < 16; ++t) out[t] = out[t] + s[t] * kk;
    }
    const uint16_t kc = (uint16_t)kx.k[H];
    const u16x2 kkc = {kc, kc};
#pragma unroll
    for (int t = 0; t < 16; ++t) out[t] = out[t] + P[t + SP * H] * kkc;
}
__device__ __forceinline__ uint32_t dot2_u16_s(uint32_t packed, uint32_t kpair, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, packed), __builtin_bit_cast(u16x2, kpair), acc, false);
}

template <int SP, int NK, bool CLAMP, bool DOWN2, int DM, bool UNIT>
__global__ __launch_bounds__(64) void k_sep_stream(StreamArgs a, TapsU8<NK> kx, TapsU8<NK> ky) {
    constexpr int H = NK / 2;
    constexpr int HB = (H * SP + 3) / 4;  // halo dwords per side
    constexpr int NQ = H + 1;             // row pairs the column pass of one output pair reaches
    constexpr int NP = 16 + 2 * H * SP;   // unpacked positions
    constexpr int D = NQ * DM;            // source row pairs in flight ahead of the arithmetic
    static_assert((H + 1) * SP <= 16, "the border halo must come out of the outer lane's own unit");

#ifdef ZG_STREAM_TRACE
    const unsigned long long trace_r0 = wall_clock64(), trace_c0 = clock64();
#endif
    const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3;
    uint32_t w = blockIdx.x;
    if (ZG_XCD_ORDER && w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3); // XCD-major: an XCD's L2 sees whole bands of neighbouring strips
    const uint32_t per_frame = (uint32_t)(a.strips_x * a.strips_y);
    const uint32_t frame = w / per_frame, t = w - frame * per_frame;
    const int sy = (int)(t / (uint32_t)a.strips_x), sx = (int)(t - (uint32_t)sy * (uint32_t)a.strips_x);
    const uint8_t *srcf = a.src + (size_t)frame * a.src_frame;
    uint8_t *dstf = a.dst + (size_t)frame * a.dst_frame;

    const int lx = (int)threadIdx.x;
    const int rb = a.row_bytes, x0 = sx * 1024;
    const int voff = x0 + 16 * lx;
    const int last_lane = (min(rb - x0, 1024) >> 4) - 1;
    const bool left_edge = sx == 0, right_edge = x0 + 1024 >= rb;
    const int border = a.border;
    // the halo loads always stay inside the row; at the image's edges they point at the other end (what .wrap wants)
    const int off_h = lx == 0 ? (left_edge ? rb - 4 * HB : x0 - 4 * HB) : (right_edge ? 0 : x0 + 1024);
    const int y0 = sy * a.strip_rows;
    const int out_pairs = (min(y0 + a.strip_rows, a.rows) - y0 + 1) >> 1;
    // source row pairs: rows y0 - H + 2q, y0 - H + 2q + 1, q < out_pairs + H — rounded up to whole blocks of D pairs, so that the
    // loop below has one exit (with an early exit inside the block hipcc's wait-count pass sees a path on which the block's newest
    // loads are the next ones needed, and waits for everything at the top of every block). The extra pairs are real rows of the
    // strip below (or resolved / dropped ones past the image): what they produce is what that strip produces, or is clipped.
    const int n_in = (out_pairs + H + D - 1) / D * D;

    // the twelve dwords the row pass reads: [4 - HB, 4) from the lane below, [4, 8) own, [8, 8 + HB) from the lane above
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
        if (decltype(edge_tag)::value && last_lane != 63) { // wave-uniform: the row ends inside this strip, its last lane has a (zero) neighbour above
#pragma unroll
            for (int d = 0; d < HB; ++d) q[8 + d] = lx == last_lane ? r.h[d] : q[8 + d];
        }
        if constexpr (!decltype(edge_tag)::value) return; // strips that touch neither end of the rows are done here
        if (left_edge && border != ZG_BORDER_WRAP) { // wave-uniform
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

    // Two ways to reach a row. GENERAL: the border rule per row and a descriptor per row whose length clips the lanes past the
    // row's end (and is 0 for a row the zero border drops) — a dozen scalar instructions per row, which matters: this kernel is
    // bound by instruction issue, scalar instructions included (profiles/r03_stream_kernel.txt). FAST, for the strips whose rows —
    // read-ahead included — all lie inside the image (every strip but a frame's first and last): one descriptor for the whole frame,
    // the row in the instruction's scalar offset, which advances by one addition per row.
    const uint32_t dst_rb = DOWN2 ? (uint32_t)rb >> 1 : (uint32_t)rb, dst_rows = DOWN2 ? (uint32_t)a.rows >> 1 : (uint32_t)a.rows;
    const auto src_all = __builtin_amdgcn_make_buffer_rsrc((void *)srcf, (short)0, (int)a.src_span, 0x00020000);
    const auto dst_all = __builtin_amdgcn_make_buffer_rsrc((void *)dstf, (short)0, (int)a.dst_span, 0x00020000);
    const bool fast = a.fast_ok && y0 - H >= 0 && y0 - H + 2 * (n_in + D) <= a.rows;
    const bool full = last_lane == 63;
    const int dvoff = DOWN2 ? voff >> 1 : voff;

    auto run = [&](auto fast_tag, auto edge_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        uint32_t s_next = (uint32_t)(y0 - H) * (uint32_t)a.src_pitch; // FAST: byte offset of the next source row to ask for
        uint32_t d_next = (uint32_t)(DOWN2 ? y0 >> 1 : y0) * (uint32_t)a.dst_pitch; // FAST: ... of the next destination row
        auto load_row = [&](int y) -> RowIn<HB> {
            RowIn<HB> r;
#ifdef ZG_STREAM_NOLOAD // tools/exp/stream_trace.hip: the arithmetic and the stores alone
            r.v = u32x4{(uint32_t)(voff + y), (uint32_t)(voff ^ y), (uint32_t)y * 2654435761u, (uint32_t)lx};
            for (int d = 0; d < HB; ++d) r.h[d] = (uint32_t)(y + d);
            return r;
#endif
            if constexpr (FAST) { // rows are asked for in ascending order, so the running offset is row y's
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
                const uint8_t *row = srcf + (size_t)(uint32_t)gr * a.src_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)((uint32_t)rb & keep), 0x00020000);
                r.v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                HaloLoad<HB>::run(rsrc, off_h, 0, r.h);
            }
            return r;
        };
        // one destination row: `o` is the lane's unit (16 bytes, or 8 behind the fused half-size resize)
        auto store_row = [&](auto o, int gy) {
#ifdef ZG_STREAM_NOSTORE // tools/exp/stream_trace.hip: the loads and the arithmetic alone (the test keeps the results alive)
            if (o[0] != 0x12345678u || o[1] != 0x9abcdef0u) return;
#endif
            if constexpr (FAST) {
                if (full) st_unit(o, dst_all, dvoff + (int)d_next);
                else if (lx <= last_lane) st_unit(o, dst_all, dvoff + (int)d_next); // the frame's descriptor does not clip a row
                d_next += (uint32_t)a.dst_pitch;
            } else {
                const uint32_t row_ok = (uint32_t)gy < dst_rows ? ~0u : 0u;
                uint8_t *row = dstf + (size_t)((uint32_t)gy & row_ok) * a.dst_pitch;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, (int)(dst_rb & row_ok), 0x00020000);
                st_unit(o, rsrc, dvoff); // row bytes % 16 == 0: a unit is all in or all out
            }
        };

        RowIn<HB> ahead[D][2]; // D row pairs in flight ahead of the arithmetic
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ahead[i][0] = load_row(y0 - H + 2 * i);
            ahead[i][1] = load_row(y0 - H + 2 * i + 1);
            // keep the requests in this order: hipcc's scheduler otherwise issues the first pair last, and since a wave's loads
            // return in order the loop's first wait (a merge of this path and the back edge) becomes "wait for everything", every
            // time round
            __builtin_amdgcn_sched_barrier(0);
        }

        // [row pair][byte]: (temp of source row 2q, temp of row 2q + 1). Pair q lives in slot q % NQ and its source rows arrive in
        // ahead[q % D]; the loop below is unrolled over D (a multiple of NQ) pairs so that every slot index is a compile-time constant
        // and nothing is ever moved between registers.
        u16x2 win[NQ][16];
        for (int qb = 0; qb < n_in; qb += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int q = qb + u;
#ifdef ZG_STREAM_NOARITH // tools/exp/stream_batch.hip: the loads and the stores alone, in the kernel's order and with its read-ahead
                {
                    const RowIn<HB> c0 = ahead[u % D][0], c1 = ahead[u % D][1];
                    ahead[u % D][0] = load_row(y0 - H + 2 * (q + D));
                    ahead[u % D][1] = load_row(y0 - H + 2 * (q + D) + 1);
                    if (q < H) continue;
                    const int gy = y0 + 2 * (q - H);
                    if constexpr (DOWN2) store_row(u32x2{c0.v[0] ^ c1.v[1] ^ c0.h[0], c0.v[2] ^ c1.v[3] ^ c1.h[HB - 1]}, gy >> 1);
                    else { store_row(u32x4{c0.v[0] ^ c0.h[0], c0.v[1], c0.v[2], c0.v[3]}, gy); store_row(u32x4{c1.v[0] ^ c1.h[HB - 1], c1.v[1], c1.v[2], c1.v[3]}, gy + 1); }
                    continue;
                }
#endif
                uint32_t q0[12], q1[12];
                widen(edge_tag, ahead[u % D][0], q0);
                widen(edge_tag, ahead[u % D][1], q1);
                u16x2 P[NP];
                UnpackRowsS<16 - H * SP, NP>::run(q0, q1, P);
                // The slot's rows are dead from here on, so the pair D ahead is asked for now and lands in the very registers it
                // replaces. (Asked for before the unpacking, it would need registers of its own, and the loop would rotate every
                // slot by moves at its back edge — moves that have to wait for the loads, which is the read-ahead gone.)
                ahead[u % D][0] = load_row(y0 - H + 2 * (q + D)); // past the strip's last pair: a resolved (valid) row nobody uses
                ahead[u % D][1] = load_row(y0 - H + 2 * (q + D) + 1);
                // tap-major: consecutive instructions write different accumulators. (Byte-major — one byte's five taps in a row — is a
                // chain of dependent packed operations, and gfx950 needs a wait state between those: hipcc put an s_nop after every one,
                // ~50 issue slots per row pair.)
                if constexpr (UNIT) row_pass_unit_sym<SP, NK, NP>(P, kx, win[u % NQ]);
                else row_pass_tap_major<SP, NK, NP>(P, kx, win[u % NQ]);
                if (q < H) continue; // wave-uniform: the strip's first H pairs only feed the window

                // output rows 2m, 2m + 1 of the strip, m = q - H: source row pairs m .. m + H = slots (u + 1) % NQ .. u % NQ, oldest first
                const int gy = y0 + 2 * (q - H);
                // divClampU8(65536, a) for a >= 0 is min(255, (a + 32768) >> 16): the rounding term seeds the accumulator. Tap-major here too.
                uint32_t ve[16], vo[16];
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) {
                    // unit end taps (UNIT: the host checked both kernels): the end rows are plain 32-bit additions with a word select (SDWA)
                    // where the general form is a 16 x 16 multiply-add — half the issue time (profiles/r03_valu_rates.txt)
                    if constexpr (UNIT) vo[t2] = (__builtin_bit_cast(uint32_t, win[(u + 1) % NQ][t2]) >> 16) + 32768u;
                    else vo[t2] = mad_hi16(__builtin_bit_cast(uint32_t, win[(u + 1) % NQ][t2]), ky.k[0], 32768u);
                }
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) ve[t2] = dot2_u16_s(__builtin_bit_cast(uint32_t, win[(u + 1) % NQ][t2]), ky.k[0] | (ky.k[1] << 16), 32768u);
#pragma unroll
                for (int h = 0; h < H; ++h) {
#pragma unroll
                    for (int t2 = 0; t2 < 16; ++t2) vo[t2] = dot2_u16_s(__builtin_bit_cast(uint32_t, win[(u + 2 + h) % NQ][t2]), ky.k[2 * h + 1] | (ky.k[2 * h + 2] << 16), vo[t2]);
                    if (h + 1 < H) {
#pragma unroll
                        for (int t2 = 0; t2 < 16; ++t2) ve[t2] = dot2_u16_s(__builtin_bit_cast(uint32_t, win[(u + 2 + h) % NQ][t2]), ky.k[2 * h + 2] | (ky.k[2 * h + 3] << 16), ve[t2]);
                    }
                }
#pragma unroll
                for (int t2 = 0; t2 < 16; ++t2) {
                    uint32_t e, o = vo[t2];
                    if constexpr (UNIT) e = (__builtin_bit_cast(uint32_t, win[u % NQ][t2]) & 0xffffu) + ve[t2];
                    else e = mad_lo16(__builtin_bit_cast(uint32_t, win[u % NQ][t2]), ky.k[NK - 1], ve[t2]);
                    if constexpr (CLAMP) {
                        e >>= 16; o >>= 16;
                        ve[t2] = e > 255u ? 255u : e;
                        vo[t2] = o > 255u ? 255u : o;
                    } else { // host proved acc < 2^24: the value is byte 2, extracted by the packing (or by the 2 x 2 mean's word selects) below
                        ve[t2] = e;
                        vo[t2] = o;
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
                        store_row(o, gy + half);
                    }
                } else {
                    static_assert(!DOWN2 || SP == 4, "the fused half-size resize is Rgba(u8) only");
                    // 2 x 2 means of the blurred pixels: pixels (0, 1) -> output 0, (2, 3) -> output 1, rows 2m and 2m + 1 (y0 is even)
                    u32x2 o;
#pragma unroll
                    for (int o2 = 0; o2 < 2; ++o2) {
                        if constexpr (CLAMP) {
                            uint32_t c[4];
#pragma unroll
                            for (int ch = 0; ch < 4; ++ch) c[ch] = (ve[8 * o2 + ch] + ve[8 * o2 + 4 + ch] + vo[8 * o2 + ch] + vo[8 * o2 + 4 + ch]) >> 2;
                            o[o2] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
                        } else {
                            // The four values of a channel are the HIGH WORDS of their accumulators (acc < 2^24): a row's two pixels are one addition
                            // with word selects on both operands (v_add_u32_sdwa, a plain-rate instruction; kept apart from the second addition — the
                            // empty asm — or hipcc folds three shifts and a v_add3_u32 out of it: 10 cycles a channel against 6), the channels are
                            // then paired (0, 2) and (1, 3) in 16-bit halves so that the divide by four and the byte packing each happen once per
                            // pair instead of once per channel. 5.5 instructions a channel -> 3.4; config 5 is bound by exactly this arithmetic.
                            uint32_t t[4];
#pragma unroll
                            for (int ch = 0; ch < 4; ++ch) {
                                uint32_t se = (ve[8 * o2 + ch] >> 16) + (ve[8 * o2 + 4 + ch] >> 16);
                                uint32_t so = (vo[8 * o2 + ch] >> 16) + (vo[8 * o2 + 4 + ch] >> 16);
                                asm volatile("" : "+v"(se), "+v"(so));
                                t[ch] = se + so; // <= 1020
                            }
                            const uint32_t p02 = ((t[0] | (t[2] << 16)) >> 2) & 0x00ff00ffu, p13 = ((t[1] | (t[3] << 16)) >> 2) & 0x00ff00ffu;
                            o[o2] = p02 | (p13 << 8);
                        }
                    }
                    store_row(o, gy >> 1);
                }
            }
        }
    };
    const bool edges = left_edge || right_edge || last_lane != 63;
    if (fast && !edges) run(std::true_type{}, std::false_type{}); // wave-uniform: nearly every strip of a frame
    else if (fast) run(std::true_type{}, std::true_type{});
    else run(std::false_type{}, std::true_type{});
#ifdef ZG_STREAM_TRACE
    if (a.trace && lx == 0) {
        a.trace[4 * blockIdx.x] = trace_r0;
        a.trace[4 * blockIdx.x + 1] = wall_clock64();
        a.trace[4 * blockIdx.x + 2] = clock64() - trace_c0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        a.trace[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// Rows per strip. Enough strips for about two waves per SIMD of the chip (2048) when one frame has to fill it — more waves do not
// help a kernel bound by instruction issue, fewer leave SIMDs idle — and taller strips for batches, where the 2H rows a strip
// convolves for its upper neighbour are the only waste (measured on MI355X, profiles/r03_stream_kernel.txt: 4096^2 Rgba 32 rows,
// 128 x 1080p 44). The count is then nudged so that a strip's row pairs fill whole blocks of the kernel's unrolled loop.
static int stream_strip_rows(const StreamJob &j, uint32_t strips_x, int h, int d) {
    const uint64_t all_rows = (uint64_t)j.rows * strips_x * j.n_frames;
    uint64_t r = (all_rows + 2047) / 2048;
    r = std::min<uint64_t>(std::max<uint64_t>(r, 16), 44);
    int pairs = (int)((r + 1) / 2);
    while ((pairs + h) % d) ++pairs; // (strip_rows / 2 + H) % D == 0: no padded pair at the end of a strip
    return 2 * pairs;
}

template <int SP, int NK, bool CLAMP, bool DOWN2>
static int launch_stream(const StreamJob &j, const int32_t *ix, const int32_t *iy, int border, hipStream_t s) {
    TapsU8<NK> kx, ky;
    for (int i = 0; i < NK; ++i) { kx.k[i] = (uint32_t)ix[i]; ky.k[i] = (uint32_t)iy[i]; }
    StreamArgs a;
    a.src = (const uint8_t *)j.src;
    a.dst = (uint8_t *)j.dst;
    a.src_pitch = j.src_pitch; a.dst_pitch = j.dst_pitch;
    a.src_frame = j.src_frame; a.dst_frame = j.dst_frame;
    a.rows = (int32_t)j.rows;
    a.row_bytes = (int32_t)(j.cols * (uint32_t)SP);
    a.strips_x = (int32_t)ceil_div((unsigned)a.row_bytes, 1024u);
    a.strip_rows = stream_strip_rows(j, (uint32_t)a.strips_x, NK / 2, NK / 2 + 1);
    a.strips_y = (int32_t)ceil_div(j.rows, (unsigned)a.strip_rows);
    a.border = border;
    const uint64_t sspan = (uint64_t)(j.rows - 1) * j.src_pitch + (uint64_t)a.row_bytes;
    const uint64_t dspan = j.down2 ? (uint64_t)(j.rows / 2 - 1) * j.dst_pitch + (uint64_t)(a.row_bytes / 2) : (uint64_t)(j.rows - 1) * j.dst_pitch + (uint64_t)a.row_bytes;
    a.fast_ok = sspan <= 0xffffffffu && dspan <= 0xffffffffu;
    a.src_span = (uint32_t)sspan;
    a.dst_span = (uint32_t)dspan;
    const uint64_t items = (uint64_t)a.strips_x * a.strips_y * j.n_frames;
    if (items > 0x7fffffffu) return -1;
    // row taps symmetric with unit ends: the folded row pass (the CLAMP instantiations — tap sums past 256 — keep the plain one)
    static const bool no_fold = getenv("ZIGNAL_HIP_STREAM_NO_FOLD") != nullptr; // A/B hook of round 5, read once
    bool unit = ix[0] == 1 && iy[0] == 1 && iy[NK - 1] == 1 && !no_fold;
    for (int i = 0; i < NK; ++i) unit = unit && ix[i] == ix[NK - 1 - i];
    if constexpr (!CLAMP) {
        if (unit) {
            hipLaunchKernelGGL((k_sep_stream<SP, NK, CLAMP, DOWN2, 1, true>), dim3((unsigned)items), dim3(64), 0, s, a, kx, ky);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        }
    }
    hipLaunchKernelGGL((k_sep_stream<SP, NK, CLAMP, DOWN2, 1, false>), dim3((unsigned)items), dim3(64), 0, s, a, kx, ky);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Returns -1 when the preconditions do not hold (caller falls back to the tiled kernels).
int try_sep_stream(const StreamJob &j, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s) {
    if (j.sp != 1 && j.sp != 3 && j.sp != 4) return -1;
    // a single grey plane is the one case the LDS-tiled kernel still wins (10.4 against 11.9 us at 4096^2): 16 pixels per lane leave
    // it little halo to re-convolve, and a 4 KiB-wide row gives this kernel only four strips across
    if (j.sp == 1 && j.n_frames == 1) return -1;
    if (nk != 3 && nk != 5 && nk != 7 && nk != 9) return -1;
    if ((nk / 2 + 1) * j.sp > 16) return -1;
    const uint64_t rb = (uint64_t)j.cols * (uint64_t)j.sp;
    if (rb % 16 || j.src_pitch % 16 || j.src_frame % 16 || ((uintptr_t)j.src & 15)) return -1;
    if (rb % 1024 == 16) return -1; // the last strip would be one lane wide: that lane is first and last at once
    if (j.cols < 64 || j.rows < 16 || rb > 0x3fffffffu || j.src_pitch > 0x7fffffffu || j.dst_pitch > 0x7fffffffu) return -1;
    if (j.down2) {
        if (j.sp != 4 || j.rows % 2 || rb % 32 || j.dst_pitch % 8 || j.dst_frame % 8 || ((uintptr_t)j.dst & 7)) return -1;
    } else {
        if (j.dst_pitch % 16 || j.dst_frame % 16 || ((uintptr_t)j.dst & 15)) return -1;
    }
    int64_t sx = 0, sy = 0;
    for (int i = 0; i < nk; ++i) {
        if (ix[i] < 0 || ix[i] > 255 || iy[i] < 0 || iy[i] > 255) return -1;
        sx += ix[i];
        sy += iy[i];
    }
    if (sx > 257 || sy > 257) return -1; // temp must fit u16: 255 * 257 = 65535
    const bool clamp = sx * sy * 255 + 32768 >= 256 * 65536; // only then can (acc >> 16) exceed 255
#define ZG_ST(SP, NK) \
    if (j.sp == SP && nk == NK) { \
        if constexpr (SP == 4) { if (j.down2) return clamp ? launch_stream<SP, NK, true, true>(j, ix, iy, border, s) : launch_stream<SP, NK, false, true>(j, ix, iy, border, s); } \
        return clamp ? launch_stream<SP, NK, true, false>(j, ix, iy, border, s) : launch_stream<SP, NK, false, false>(j, ix, iy, border, s); \
    }
    ZG_ST(1, 3) ZG_ST(1, 5) ZG_ST(1, 7) ZG_ST(1, 9)
    ZG_ST(3, 3) ZG_ST(3, 5) ZG_ST(3, 7) ZG_ST(3, 9)
    ZG_ST(4, 3) ZG_ST(4, 5) ZG_ST(4, 7)
#undef ZG_ST
    return -1;
}

} // namespace zg
