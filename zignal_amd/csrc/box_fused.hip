// box_fused.hip — Image(T).boxBlur / sharpen of u8 pixel types with the summed-area table kept on the chip (round 6).
//
// Replaces, for Image(u8) / Rgba(u8) and radius 1..3, the three-kernel route of box_blur.hip (strip carries, SAT chain, window means), which
// wrote 4 B of f32 SAT per source byte and read it back four times: 859 MB of traffic for 134 MB of algorithmic bytes on a 4096^2 Rgba(u8) frame.
// Reference: src/image.zig:635-648 (boxBlur), :785-801 (sharpen), src/image/integral.zig:41-78 (Integral.plane), :86-91 (sum), :194-269
// (boxBlurPlane), :273-426 (sharpen).
//
// What has to be reproduced bit for bit is the reference's f32 SAT: sat[r][c] = sat[r-1][c] + rowprefix[r][c], one f32 addition per row and column,
// top to bottom (inexact from 2^24 on), and the window sum ((a - b) - d) + e of four of its entries. A column's chain depends on nothing but that
// column's row prefixes, and for u8 sources those are exact integers below 2^24 in any association: the sum left of a strip (its CARRY, a small
// table written by k_box_carries) plus a prefix inside the strip. So a workgroup can own a strip of pixel columns for the whole height, chain the
// columns its windows touch — its W output columns plus R + 1 to the left and R to the right, re-running those 2R + 1 neighbours' chains itself —
// and never store a SAT row anywhere but in LDS:
//
//   k_box_carries   one wave per row: K[r][k][ch] = sum of row r left of strip k's first chained column (rows x (strips + 1) x C floats)
//   k_box_fused     a strip of 16 pixel columns per workgroup (4096 columns: 256 workgroups, one per CU, one round); a byte column (pixel column x
//                   channel) per chain lane — 96 for Rgba(u8): 16 + R + 1 + R pixels rounded up to quarters of 16 bytes, 32 for Image(u8) — and the
//                   waves dealt out by role (waves w and w + 4 share a SIMD):
//       LOADERS     lane = (group of four rows, row pair, 16 adjacent bytes of it): one load per row, bytes -> f32, prefix inside the lane along each
//                   channel, scan over the lanes of a row (v_permlane16/32_swap), + carry; the pair goes to LDS as half of 16 chain lanes' vectors.
//       CHAIN       per four rows: ds_read_b128 of row prefixes, four dependent v_add_f32 (THE sequential part: rows x 1 addition), ds_write_b128
//                   of SAT values into a ring of three 64-row blocks. Rgba(u8): two waves (64 + 32 lanes) on two SIMDs.
//       MEANS       corners a, b of a row are one ds_read_b128 per four rows and lane, corners d, e are the a, b of 2R + 1 rows earlier (the groups
//                   above, read again); ((a - b) - d) + e, then division, rounding, clamp and pack as ONE fma and one v_cvt_pk_u8_f32 (below),
//                   a 4 x 4 byte transpose across the quad (two DPP moves, two v_perm) so that every lane stores one aligned dword of one row.
//   Steps are separated by one s_barrier: at step s the loaders work on blocks s and s - 1, the chain turns block s - 2 into SAT rows, the means
//   finish block s - 3. The first R + 1 .. and the last R rows, whose windows are clipped, take a generic path (any row, LDS reads per corner).
//   History of the geometry (profiles/r06_box_blur.txt): the first version chained 64 byte columns per workgroup — 56 grey outputs (74 workgroups on 256
//   CUs) or 11 Rgba pixels (373 workgroups: two rounds, 44 of a mean wave's 64 lanes live).
//
// meta.clamp(u8, v) is round-half-away then clamp. v_cvt_pk_u8_f32 rounds to nearest EVEN and saturates, so it is fed v + 2^-10: v = n / area with an
// integer n and area <= 49, so a v that is not exactly k + 1/2 is at least 1 / 98 away from it (2^-10 cannot carry it across, nor can the quotient's
// rounding error of <= 2^-16), and k + 1/2 + 2^-10 is representable and rounds up, which is what half-away does for v >= 0; negative v saturate to 0
// either way. The same holds for sharpen's 2 * original - v.
#include "zg_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstdio>

#pragma clang fp contract(off)

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);

constexpr int BF_W = 16;        // output pixel columns per strip: 4096 columns = 256 strips = one workgroup per CU
constexpr int BF_NP = 2;        // row-prefix ring: block s - 1 is written (at the end of its loaders' second step) while block s - 2 is read
constexpr int BF_NS = 3;        // SAT ring: block s - 2 is written while the means read s - 3 and its predecessor (history rows)
constexpr int BF_MAX_R = 3;
#ifndef BF_DEPTH
#define BF_DEPTH 1
#endif
constexpr int BF_MAX_HG = 2;    // (2 R + 1 + 3) / 4 groups of history at most

// A strip is NQ QUARTERS of 16 byte columns of the image's rows (a byte column = pixel column x channel): its 16 output pixels, LEFT columns left of them —
// the R + 1 a window reaches back, rounded up to a whole dword for one-channel images so that every loader lane's 16 bytes start on a dword — and R to
// the right. Rgba(u8): 6 quarters = 24 pixels (16 + 4 + 3 at most); Image(u8): 2 quarters = 32 columns.
template <int C> struct BoxGeo;
template <> struct BoxGeo<4> {
    static constexpr int B = 64 /* rows per block = per step */, NQ = 6, THREADS = 1024, NM = 8 /* mean waves */, NG = 2 /* groups of four rows a mean lane takes at once */,
                         LPG = 64 /* lanes per group: 16 pixels x 4 */, NLOAD = 4;
};
template <> struct BoxGeo<1> {
    // a step costs ~600 cycles whatever is done in it (a barrier, an LDS round trip or two: removing any one role's work saved 4 .. 12 % of the kernel's time), and
    // one channel has the LDS for twice the rows per step
#ifndef BF_THREADS1
#define BF_THREADS1 576
#endif
    static constexpr int B = 128, NQ = 2, THREADS = BF_THREADS1, NM = 4, NG = 1, LPG = 16, NLOAD = 4;
};
__host__ __device__ constexpr int box_strip_left(int C, int R) { return C == 4 ? R + 1 : 4; }

// LDS layout (both rings): [four-row group][chain lane] float4 = the lane's column in the group's four rows, RS = lanes + 1 slots from one group to the
// next, NO padding inside a group. Under the bank rules of the LDS (MI355X_MICROARCH.md: a ds_read_b128 is served in four fixed sets of 16 lanes on 64
// banks, stores in sets of 8 / 16 / 32 contiguous lanes on 32 banks) every access of this kernel is then free of conflicts — tools/exp/box_lds_conflicts.py
// counts them per layout — provided that the 16 (32) lanes a store serves together differ in (group mod 8, row pair) and NOT in the quarter of the row:
// a loader's quarter sits in its lane's top bits and the scan over a row's quarters crosses rows of 16 lanes (v_permlane16_swap / v_permlane32_swap).
// The first version had a slot of padding per quarter and the quarter in the low lane bits (a scan by quad DPP moves): 42 % of its LDS cycles were conflicts.
__device__ __forceinline__ int bf_pos(int l) { return l; }

struct BfRoleList { int r[16]; };
struct BfRoles { unsigned long long lo, hi; }; // 8 bits per wave
constexpr BfRoles bf_roles(BfRoleList l) {
    BfRoles v{0, 0};
    for (int w = 0; w < 8; ++w) v.lo |= (unsigned long long)l.r[w] << (8 * w);
    for (int w = 8; w < 16; ++w) v.hi |= (unsigned long long)l.r[w] << (8 * (w - 8));
    return v;
}

struct BoxFusedArgs {
    DImg src, dst;
    const float *carries; // [frame][row][strip 0 .. nwg][C]
    int nwg;              // strips
    int nk;               // carries of a row: nwg + 1
    int nwg8;             // ceil(nwg / 8): strips per XCD
    size_t src_frame, dst_frame;
#ifdef BF_TIMING
    unsigned long long *timing; // [strip < 1024][role < 16][2]: cycles a wave spent between barriers, cycles it waited in them (tools/build_variant.sh ... -DBF_TIMING)
#endif
};

// ---- carries ---------------------------------------------------------------------------------------------------------------------------------------
// K[(r * nk + k) * C + ch] = sum over columns < 16 k - LEFT (strip k's first chained column) of row r, channel ch (0 for k = 0), exact in f32; k = nwg too
// for four channels (the strip's last two quarters take the NEXT strip's carry: they start where it starts).
// One WAVE per row, no LDS and no barrier. A lane takes one PIECE of the row — the 16 columns from one strip's first chained column to the next one's — and
// sums it by itself (packed 16-bit pairs for four channels: 16 x 255 < 2^16; v_sad_u8 for one channel), so that the wave needs one scan per 64 pieces
// (1024 pixels) instead of one per 256 pixels: the first versions (a block scan per 1024 pixels with a barrier; then a wave scan per 256 pixels) took 22 us
// for a 4096^2 Rgba(u8) frame whatever their loads did — 85 instructions per 256 pixels, bound by issue.
template <int C, int LEFT>
__global__ __launch_bounds__(256) void k_box_carries(DImg src, float *K, int nwg, int nk, size_t src_frame) {
    src.data = (char *)src.data + (size_t)blockIdx.y * src_frame;
    K += (size_t)blockIdx.y * src.rows * nk * C;
    const int lane = threadIdx.x & 63;
    const int r = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= src.rows) return;
    const uint8_t *row = (const uint8_t *)src.data + (size_t)r * src.stride * C; // 4-byte aligned (checked by the host)
    float *Kr = K + (size_t)r * nk * C;
    if (lane < C) Kr[lane] = 0.0f; // strip 0
    uint32_t carry[C];                // the row's total left of the batch (wave-uniform)
#pragma unroll
    for (int ch = 0; ch < C; ++ch) carry[ch] = 0;
    constexpr int ND = BF_W * C / 4; // dwords of a piece
    static_assert((LEFT * C) % 4 == 0, "pieces are whole dwords");
    // piece p = columns [16 p - LEFT, 16 (p + 1) - LEFT) gives K[p + 1]. One channel: p <= nwg - 2, which lies inside the row; four channels: p <= nwg - 1,
    // which may reach past the row's end. Piece 0 starts left of the row. Both ends are masked (whole dwords: a pixel of four channels; for one channel only
    // the left end, at a multiple of four columns).
    const int npieces = C == 4 ? nwg : nwg - 1;
    const int row_dwords = src.cols * C / 4;
    constexpr int LD = LEFT * C / 4;                              // dwords of piece 0 that lie left of the row
    const bool overhang = npieces * BF_W - LEFT > src.cols;       // (never for one channel: its last piece ends inside the row)
    for (int p0 = 0; p0 < npieces; p0 += 64) {
        const int p = min(p0 + lane, npieces - 1);
        const int d0 = (p * BF_W - LEFT) * C / 4; // first dword of my piece
        const uint32_t *d = (const uint32_t *)row + d0;
        uint32_t v[ND];
        if (overhang && p0 + 64 >= npieces) { // wave-uniform: the batch holds the row's last piece and that reaches past the row's end (cols % 16 != 0)
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const bool real = d0 + i >= 0 && d0 + i < row_dwords;
                v[i] = ((const uint32_t *)row)[min(max(d0 + i, 0), row_dwords - 1)] & (real ? 0xffffffffu : 0u);
            }
        } else if (p0 == 0) { // piece 0 starts LD dwords left of the row: its lane reads the row's first ND dwords instead and drops the last LD of them
            d = (const uint32_t *)row + max(d0, 0);
#pragma unroll
            for (int i = 0; i < ND; ++i) v[i] = d[i] & ((i >= ND - LD && p == 0) ? 0u : 0xffffffffu);
        } else {
#pragma unroll
            for (int i = 0; i < ND; ++i) v[i] = d[i];
        }
        uint32_t x[C]; // my piece's sum per channel
        if constexpr (C == 4) { // a dword is a pixel: ch0 | ch2 << 16 and ch1 | ch3 << 16 as packed 16-bit sums
            uint32_t e = 0, o = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                e += __builtin_amdgcn_perm(0, v[i], 0x0c020c00);
                o += __builtin_amdgcn_perm(0, v[i], 0x0c030c01);
            }
            x[0] = e & 0xffffu; x[1] = o & 0xffffu; x[2] = e >> 16; x[3] = o >> 16;
        } else {
            uint32_t t = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) t = __builtin_amdgcn_sad_u8(v[i], 0u, t);
            x[0] = t;
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) { // inclusive scan over the wave
            uint32_t t = x[ch];
            t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xf, 0xf, true);  // row_shr:1
            t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xf, 0xf, true);  // row_shr:2
            t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xf, 0xf, true);  // row_shr:4
            t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xf, 0xf, true);  // row_shr:8
            t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
            t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
            x[ch] = t + carry[ch];                                                         // everything left of strip p + 1: < 2^24, exact as f32
            carry[ch] += (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
        }
        if (p0 + lane < npieces) {
            if constexpr (C == 4) *(float4 *)(Kr + (size_t)(p + 1) * 4) = make_float4((float)x[0], (float)x[1], (float)x[2], (float)x[3]);
            else Kr[p + 1] = (float)x[0];
        }
    }
}

// ---- the division and the pack ----------------------------------------------------------------------------------------------------------------------
struct BoxDiv { float nd, rcp; }; // -area and the refined reciprocal of the IEEE division sequence (v_rcp + one Newton step)
__device__ __forceinline__ BoxDiv box_div_of(float area) {
    const float nd = -area, r0 = __builtin_amdgcn_rcpf(area);
    return BoxDiv{nd, __builtin_fmaf(__builtin_fmaf(nd, r0, 1.0f), r0, r0)};
}
// sum / area, bit for bit what `/` expands to when v_div_scale has nothing to scale (a finite sum over an area >= 1): box_blur.hip k_box_mean
__device__ __forceinline__ float box_quot(float sum, const BoxDiv &d) {
    const float q0 = sum * d.rcp;
    const float q1 = __builtin_fmaf(__builtin_fmaf(d.nd, q0, sum), d.rcp, q0);
    return __builtin_fmaf(__builtin_fmaf(d.nd, q1, sum), d.rcp, q1);
}
constexpr float BF_BIAS = 0x1p-10f; // see the header: turns v_cvt_pk_u8_f32's nearest-even into meta.clamp's half-away on this value set

#ifdef BF_TIMING
struct BfTimer {
    unsigned long long busy = 0, wait = 0, last = 0;
    __device__ __forceinline__ void sync() {
        const unsigned long long t0 = __builtin_readcyclecounter();
        __syncthreads();
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (last) busy += t0 - last;
        wait += t1 - t0;
        last = t1;
    }
};
#define BF_SYNC() bf_timer.sync()
#else
#define BF_SYNC() __syncthreads()
#endif

// ---- the loaders' pieces ------------------------------------------------------------------------------------------------------------------------------
// The 16 bytes of a row a loader lane owns. In edge strips (some chained columns lie outside the image): clamped offsets and AND masks per dword — a
// select on the loaded value would be turned into a branch around the load, and loads under a branch are waited for one by one.
// EDGE: 0 the strip lies inside the rows; 1 dwords are inside or outside as a whole (always so with four channels); 2 one channel and a row that does not end on a dword
template <int C, int EDGE>
struct BfCols {
    int byte0;
    uint32_t eoff[4], emask[4], emask2[4], poff[3], pmask[3];
    // col0: my first pixel column (C = 4: four pixels) / byte column (C = 1: sixteen, a multiple of four)
    __device__ __forceinline__ void init(int col0, int cols) {
        byte0 = col0 * C;
        if constexpr (EDGE) {
            const int cols4 = cols & ~3;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if constexpr (C == 4) {
                    const int cc = col0 + d; // my pixel
                    eoff[d] = (uint32_t)min(max(cc, 0), cols - 1) * 4u;
                    emask[d] = cc >= 0 && cc < cols ? 0xffffffffu : 0u;
                    emask2[d] = 0;
                } else {
                    const int cc = col0 + 4 * d;
                    eoff[d] = (uint32_t)min(max(cc, 0), cols4 - 4);
                    emask[d] = cc >= 0 && cc + 4 <= cols ? 0xffffffffu : 0u;
                    emask2[d] = EDGE == 2 && cc == cols4 ? 0xffffffffu : 0u;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                poff[i] = (uint32_t)min(cols4 + i, cols - 1);
                pmask[i] = cols4 + i < cols ? 0xffu : 0u;
            }
            // keep the masks as register values: turned back into compares they take eight scalar registers that the kernel does not have
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                asm volatile("" : "+v"(emask[d]));
                if constexpr (EDGE == 2) asm volatile("" : "+v"(emask2[d]));
            }
        }
    }
    __device__ __forceinline__ void load(const uint8_t *rowp, uint32_t (&raw)[4]) const { // unpredicated, unconditional
        if constexpr (!EDGE) {
            const uint32_t *p4 = (const uint32_t *)(rowp + byte0); // 4-byte aligned; the 16 bytes may straddle a line
            raw[0] = p4[0]; raw[1] = p4[1]; raw[2] = p4[2]; raw[3] = p4[3];
        } else if constexpr (EDGE == 1) { // dwords inside or outside as a whole
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[i] = *(const uint32_t *)(rowp + eoff[i]) & emask[i];
        } else { // whole dwords where they lie inside the row; the row's last, partial dword (the same for every lane) from its bytes
            uint32_t part = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) part |= ((uint32_t)rowp[poff[i]] & pmask[i]) << (8 * i);
#pragma unroll
            for (int d = 0; d < 4; ++d) raw[d] = (*(const uint32_t *)(rowp + eoff[d]) & emask[d]) | (part & emask2[d]);
        }
    }
};

// the row prefixes of a lane's 16 bytes: bytes -> f32, prefix inside the lane along each channel (bytes C apart), + what lies left of the lane: the carry and
// the totals of the row's quarters to the left, which sit in the lanes 16 / 32 / 48 below. SCAN: 1 two quarters at lane bit 4; 2 four quarters at lane
// bits 4, 5; 3 two quarters at lane bit 5. m1 / m2: 1 where that bit of my quarter is set. All sums are integers below 2^24: exact in any association.
// v_permlane16_swap(x, x) = ({r0, r0, r2, r2}, {r1, r1, r3, r3}) and v_permlane32_swap(x, x) = ({r0, r1, r0, r1}, {r2, r3, r2, r3}) of x's rows of 16 lanes.
template <int C, int SCAN>
__device__ __forceinline__ void bf_prefixes(const uint32_t (&raw)[4], const float (&kk)[C], float m1, float m2, float (&val)[16]) {
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = (float)((raw[i >> 2] >> (8 * (i & 3))) & 0xffu); // v_cvt_f32_ubyteN
#pragma unroll
    for (int i = C; i < 16; ++i) e[i] = e[i - C] + e[i];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const uint32_t t = __builtin_bit_cast(uint32_t, e[16 - C + ch]); // my total of this channel
        float base = kk[ch];
        if constexpr (SCAN == 1 || SCAN == 2) {
            const auto s16 = __builtin_amdgcn_permlane16_swap(t, t, false, false);
            uint32_t s16lo = s16[0], s16hi = s16[1];
            // hipcc 7.2 folds the sum of the swap's two results into twice the first (lo + hi became lo + lo): the results pass through an empty asm
            if constexpr (SCAN == 2) asm volatile("" : "+v"(s16lo), "+v"(s16hi));
            const float lo = __builtin_bit_cast(float, s16lo), hi = __builtin_bit_cast(float, s16hi); // the even and the odd quarter of my pair
            base = __builtin_fmaf(lo, m1, base);
            if constexpr (SCAN == 2) {
                const uint32_t pair = __builtin_bit_cast(uint32_t, lo + hi);
                const auto s32 = __builtin_amdgcn_permlane32_swap(pair, pair, false, false);
                base = __builtin_fmaf(__builtin_bit_cast(float, s32[0]), m2, base); // quarters 0 + 1
            }
        } else {
            const auto s32 = __builtin_amdgcn_permlane32_swap(t, t, false, false);
            base = __builtin_fmaf(__builtin_bit_cast(float, s32[0]), m1, base);
        }
#pragma unroll
        for (int i = ch; i < 16; i += C) val[i] = base + e[i];
    }
}

// ---- the fused kernel -------------------------------------------------------------------------------------------------------------------------------
// The byte of a mean. Only the BYTE has to equal the reference's, and a window sum is an integer (sums and differences of integer-valued floats): sum / area
// is then an exact tie k + 1/2 or at least 1 / (2 area) >= 1 / 98 away from one, far more than the error of one multiplication by the correctly rounded
// reciprocal. So the division is ONE fma, the rounding bias riding in its addend: v_cvt_pk_u8_f32(fma(sum, 1 / area, 2^-10)). tools/exp/box_quot_check.hip
// compares it with the IEEE division sequence over EVERY integer-valued f32 sum below 2^34, every area h x w (h, w <= 7), blur and sharpen: no byte differs
// (profiles/r06_box_quot_check.txt; Markstein's three-operation quotient, also checked there, is not needed).
//
// A loader of the strip's first four quarters (four channels) or of the whole strip (one channel). Four channels: wave li owns the four-row groups
// 8 hh .. 8 hh + 7 (hh = li >> 1) of blocks par, par + 2, ... (par = li & 1), lane = (group, row pair h, quarter); one channel: wave li owns blocks li,
// li + 2, ..., whole: lane = (group, row pair, quarter of two). The first row of a pair is worked at step blk (kept in registers), the second at step
// blk + 1, then both go to LDS as the low or the high half of 16 chain lanes' vectors (the slot is read by the chain until step blk). Right behind each row
// the load of the same row of the wave's next block. No condition inside the loop: with one the compiler waits for the loads at the loop's end.
// What bounds a step is its LONGEST wave — a wave of this kernel issues an instruction every ~10 cycles whatever its neighbours do — so the work is cut
// into many waves of ~90 instructions per step rather than few long ones (the last two quarters have waves of their own: box_loader_b).
template <int C, int EDGE>
__device__ __forceinline__ void box_loader(const BoxFusedArgs &A, float4 (*Pr)[BoxGeo<C>::B / 4][16 * BoxGeo<C>::NQ + 1], int li, int lane, int k, const uint8_t *src, size_t spitch, const float *K,
                                           int a0, int nblocks, int nsteps) {
#ifdef BF_TIMING
    BfTimer bf_timer;
#endif
    constexpr int SCAN = C == 4 ? 2 : 1, BF_B = BoxGeo<C>::B;
    const int rows = A.src.rows, cols = A.src.cols;
    const int par = li & 1;
    const int q = C == 4 ? lane >> 4 : (lane >> 4) & 1;
    const int rg = C == 4 ? 8 * (li >> 1) + (lane & 7) : 16 * (li >> 1) + (lane & 7) + 8 * (lane >> 5);
    const int h = (lane >> 3) & 1;
    const float fm0 = (q & 1) ? 1.0f : 0.0f, fm1 = (q & 2) ? 1.0f : 0.0f;
    BfCols<C, EDGE> mine;
    mine.init(C == 4 ? a0 + 4 * q : a0 + 16 * q, cols);
    const uint32_t krow = (uint32_t)A.nk * C; // floats of a row of carries
    constexpr int DEPTH = BF_DEPTH; // blocks of mine whose rows are in flight or in registers
    uint32_t raw[DEPTH][2][4]; // [block][row][dword]
    float kk[DEPTH][2][C];
    // rows and carries are reached by 32-bit offsets from wave-uniform bases (the host keeps both below 2^32 bytes): half the registers and half the
    // additions of per-lane pointers
    auto fetch_at = [&](uint32_t row_off, uint32_t k_off, int d, int u) {
        mine.load(src + row_off, raw[d][u]);
        const float *kp = (const float *)((const char *)K + k_off);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) kk[d][u][ch] = kp[ch];
    };
    auto row_off_of = [&](int r) { return (uint32_t)r * (uint32_t)spitch; };
    auto k_off_of = [&](int r) { return ((uint32_t)r * (uint32_t)A.nk + (uint32_t)k) * (uint32_t)(C * sizeof(float)); };
    auto fetch = [&](int blk, int d, int u) { // row 2h + u of my group in block blk, clamped into the image
#if defined(BF_NO_LOADERS) || defined(BF_NO_FETCH)
        return;
#endif
        const int r = min(blk * BF_B + rg * 4 + 2 * h + u, rows - 1);
        fetch_at(row_off_of(r), k_off_of(r), d, u);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        fetch(par + 2 * d, d, 0);
        fetch(par + 2 * d, d, 1);
    }
    int done = 0;
    if (par == 1) { BF_SYNC(); done = 1; }
    auto step_a = [&](int d, float (&va)[16]) {
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_PUBLISH)
        bf_prefixes<C, SCAN>(raw[d][0], kk[d][0], fm0, fm1, va);
#endif
    };
    auto step_b = [&](int d, const float (&va)[16]) {
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_PUBLISH)
        float vb[16];
        bf_prefixes<C, SCAN>(raw[d][1], kk[d][1], fm0, fm1, vb);
        float2 *o = (float2 *)&Pr[par][rg][16 * q] + h; // my blocks sit in slot blk & 1 = par
#pragma unroll
        for (int i = 0; i < 16; ++i) o[2 * i] = make_float2(va[i], vb[i]);
#endif
    };
    int blk = par;
    // while the blocks fetched next are whole, their rows are reached by moving offsets on (the clamped form multiplies: quarter-rate instructions)
    const int nfull = rows / BF_B;
    const uint32_t row_step = 2u * BF_B * (uint32_t)spitch, k_step = 2u * BF_B * (uint32_t)krow * (uint32_t)sizeof(float);
    const int r_next = min((par + 2 * DEPTH) * BF_B + rg * 4 + 2 * h, rows - 1); // my first row of the next block to fetch (used only when that block is whole)
    uint32_t ro = row_off_of(r_next), ko = k_off_of(r_next);
    for (; blk + 4 * DEPTH - 2 < nfull; blk += 2 * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            float va[16];
            BF_SYNC(); // step blk + 2 d
            step_a(d, va);
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_FETCH)
            fetch_at(ro, ko, d, 0);
#endif
            BF_SYNC(); // step blk + 2 d + 1
            step_b(d, va);
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_FETCH)
            fetch_at(ro + (uint32_t)spitch, ko + (uint32_t)(krow * sizeof(float)), d, 1);
#endif
            ro += row_step;
            ko += k_step;
            done += 2;
        }
    }
    for (; blk < nblocks; blk += 2 * DEPTH) { // the last blocks: what they fetch is partial or past the end (clamped, never used)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (blk + 2 * d < nblocks) {
                float va[16];
                BF_SYNC();
                step_a(d, va);
                fetch(blk + 2 * d + 2 * DEPTH, d, 0);
                BF_SYNC();
                step_b(d, va);
                fetch(blk + 2 * d + 2 * DEPTH, d, 1);
                done += 2;
            }
        }
    }
    for (; done < nsteps; ++done) BF_SYNC();
#ifdef BF_TIMING
    if (lane == 0 && k < 1024) { const int w = 3 + li; A.timing[(k * 16 + w) * 2] = bf_timer.busy; A.timing[(k * 16 + w) * 2 + 1] = bf_timer.wait; }
#endif
}

// A loader of the strip's last two quarters (four channels: pixels 16 .. 23, carry of the NEXT strip, which starts where they start). Wave hh owns the groups
// 8 hh .. 8 hh + 7 of EVERY block: lane = (group, row of four, quarter), one row each, worked and stored at step blk + 1 (until step blk the chain still reads
// the slot); its loads run two blocks ahead, in two register sets.
template <int EDGE>
__device__ __forceinline__ void box_loader_b(const BoxFusedArgs &A, float4 (*Pr)[BoxGeo<4>::B / 4][16 * BoxGeo<4>::NQ + 1], int hh, int lane, int k, const uint8_t *src, size_t spitch, const float *K,
                                             int a0, int nblocks, int nsteps) {
#ifdef BF_TIMING
    BfTimer bf_timer;
#endif
    constexpr int C = 4, BF_B = BoxGeo<4>::B;
    const int rows = A.src.rows, cols = A.src.cols;
    const int q2 = lane >> 5, rg2 = 8 * hh + (lane & 7), rin = (lane >> 3) & 3, row2 = 4 * rg2 + rin; // my row inside a block
    const float fm2 = q2 ? 1.0f : 0.0f;
    BfCols<C, EDGE> mine;
    mine.init(a0 + 16 + 4 * q2, cols);
    uint32_t raw[2][4]; // [block parity][dword]
    float kk[2][C];
    auto fetch_at = [&](uint32_t row_off, uint32_t k_off, int u) {
        mine.load(src + row_off, raw[u]);
        const float *kp = (const float *)((const char *)K + k_off);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) kk[u][ch] = kp[ch];
    };
    auto row_off_of = [&](int r) { return (uint32_t)r * (uint32_t)spitch; };
    auto k_off_of = [&](int r) { return ((uint32_t)r * (uint32_t)A.nk + (uint32_t)(k + 1)) * (uint32_t)(C * sizeof(float)); };
    auto fetch = [&](int blk, int u) {
#if defined(BF_NO_LOADERS) || defined(BF_NO_FETCH)
        return;
#endif
        const int r = min(blk * BF_B + row2, rows - 1);
        fetch_at(row_off_of(r), k_off_of(r), u);
    };
    auto work = [&](int u) { // block parity = LDS slot = register set
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_PUBLISH)
        float v[16];
        bf_prefixes<C, 3>(raw[u], kk[u], fm2, 0.0f, v);
        float *o = (float *)&Pr[u][rg2][16 * (4 + q2)] + rin;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[4 * i] = v[i];
#endif
    };
    fetch(0, 0);
    fetch(1, 1);
    BF_SYNC(); // step 0
    int done = 1, blk = 0;
    const int nfull = rows / BF_B;
    const uint32_t row_step = 2u * BF_B * (uint32_t)spitch, k_step = 2u * BF_B * (uint32_t)A.nk * (uint32_t)(C * sizeof(float));
    const int r_next = min(2 * BF_B + row2, rows - 1); // my row of block 2 (used only when that block is whole)
    uint32_t ro = row_off_of(r_next), ko = k_off_of(r_next);
    const uint32_t row_blk = (uint32_t)BF_B * (uint32_t)spitch, k_blk = (uint32_t)BF_B * (uint32_t)A.nk * (uint32_t)(C * sizeof(float));
    for (; blk + 3 < nfull; blk += 2) { // the blocks fetched, blk + 2 and blk + 3, are whole
        BF_SYNC(); // step blk + 1
        work(0);
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_FETCH)
        fetch_at(ro, ko, 0);
#endif
        BF_SYNC(); // step blk + 2
        work(1);
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_FETCH)
        fetch_at(ro + row_blk, ko + k_blk, 1);
#endif
        ro += row_step;
        ko += k_step;
        done += 2;
    }
    for (; blk < nblocks; blk += 2) { // the last blocks: what they fetch is partial or past the end (clamped, never used)
        BF_SYNC(); // step blk + 1
        work(0);
        fetch(blk + 2, 0);
        ++done;
        if (blk + 1 < nblocks) {
            BF_SYNC(); // step blk + 2
            work(1);
            fetch(blk + 3, 1);
            ++done;
        }
    }
    for (; done < nsteps; ++done) BF_SYNC();
#ifdef BF_TIMING
    if (lane == 0 && k < 1024 && hh == 0) { const int w = 15; A.timing[(k * 16 + w) * 2] = bf_timer.busy; A.timing[(k * 16 + w) * 2 + 1] = bf_timer.wait; }
#endif
}

template <int C, int R, bool SHARPEN>
__global__ __launch_bounds__(BoxGeo<C>::THREADS) void k_box_fused(BoxFusedArgs A) {
    using Geo = BoxGeo<C>;
    constexpr int W = BF_W, LEFT = box_strip_left(C, R);
    constexpr int CL = 16 * Geo::NQ, RS = CL + 1; // chain lanes; 16-byte units from one group to the next in LDS (odd: see bf_pos)
    constexpr int HG = (2 * R + 1 + 3) / 4;             // groups of history a group's d / e rows reach back into
    constexpr int BF_B = Geo::B, BF_G = BF_B / 4; // rows and four-row groups per block
    constexpr int NM = Geo::NM, NG = Geo::NG, LPG = Geo::LPG, GPW = NG * (64 / LPG), NPASS = BF_G / (NM * GPW); // groups a mean wave takes at once; times per step
    static_assert(NM * GPW * NPASS == BF_G && HG <= BF_MAX_HG && LEFT + W + R <= CL / C, "geometry");
    __shared__ float4 Pr[BF_NP][BF_G][RS];
    // the SAT ring, with a header: the last HG groups of the last slot once more, so that the groups above slot 0 are found where those above any other slot
    // are — HG groups before it
    __shared__ float4 Sr[HG + BF_NS * BF_G][RS];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // role of a wave (SIMD = wave % 4). Four channels: 0, 4 the chain, 8, 12 leave at once (SIMD 0 is the chain's); 1, 5, 2, 6 loaders, the other eight are means.
    // One channel: 0 the chain, 4 leaves; 1, 2 loaders; 3, 5, 6, 7 means
    // role of wave w: 0 leaves at once, 1 / 2 the chain's waves, 3 .. 6 loaders 0 .. 3, 7 .. 14 mean waves 0 .. 7, 15 / 16 the loaders of the last two quarters.
    // Waves w and w + 4 share a SIMD (which roles share one made no difference: +-1 % over three placements).
#ifndef BF_ROLES4 //          SIMD: 0  1  2  3   0  1  2   3  0  1  2   3   0   1   2   3
#define BF_ROLES4 bf_roles({1, 3, 2, 5, 15, 4, 16, 6, 7, 8, 9, 10, 11, 12, 13, 14}) // 0, 2: a chain wave, a loader of the last quarters, two mean waves; 1, 3: two loaders, two mean waves
#endif
#ifndef BF_ROLES1
#define BF_ROLES1 bf_roles({1, 3, 4, 5, 7, 6, 8, 9, 10, 0, 0, 0, 0, 0, 0, 0})
#endif
    constexpr BfRoles ROLES = C == 4 ? BF_ROLES4 : BF_ROLES1; // one channel: 9 waves
    const int role = (int)(((wave < 8 ? ROLES.lo : ROLES.hi) >> (8 * (wave & 7))) & 255);
    if (role == 0) return;
    // strips of one XCD are neighbours: they share source lines (the re-chained columns) and the halves of output lines in that XCD's L2
    const int b = (int)blockIdx.x;
    const int k = ZG_XCD_ORDER ? (b & 7) * A.nwg8 + (b >> 3) : b;
    if (k >= A.nwg) return;
    const int frame = (int)blockIdx.y;
    const int rows = A.src.rows, cols = A.src.cols;
    const uint8_t *src = (const uint8_t *)A.src.data + (size_t)frame * A.src_frame;
    uint8_t *dst = (uint8_t *)A.dst.data + (size_t)frame * A.dst_frame;
    const size_t spitch = (size_t)A.src.stride * C, dpitch = (size_t)A.dst.stride * C;
    const float *K = A.carries + (size_t)frame * rows * A.nk * C;
    const int nblocks = (rows + BF_B - 1) / BF_B;
    constexpr int LAG = 2;
    const int nsteps = nblocks + LAG + 1; // loaders at s = blk, blk + 1; chain at blk + LAG; means at blk + LAG + 1
    const int x0 = k * W;          // first output column
    const int a0 = x0 - LEFT;      // first chained column (may be negative: those columns hold zeros, which is what the reference's c1 == 0 case reads)

    if (role <= 2) { // ---- a chain wave ------------------------------------------------------------------------------------------------------
        // lane l of wave 0 chains byte column l of the strip; with four channels the 32 columns from 64 on are wave 4's (its upper half idles: an LDS
        // instruction costs what its live lanes move). One channel: 32 columns, wave 0's lower half. The two waves share SIMD 0 and nothing else does:
        // a chain's additions wait for each other, the other chain's fill the gaps.
#ifdef BF_TIMING
        BfTimer bf_timer;
#endif
        static_assert(CL == 32 || CL == 96, "lane masks below");
        const int first = role == 1 ? 0 : 64, nl = min(CL - first, 64); // wave-uniform
        const int pl = bf_pos(first + (lane & (nl - 1)));
        float run = 0.0f;
        // every group's SAT values replace its row prefixes in the SAME registers and the stores trail the additions by half a block: a register
        // that a store still reads is never the target of an addition
        auto get = [&](float4 (&v)[BF_G], int pslot) {
#pragma unroll
            for (int g = 0; g < BF_G; ++g) v[g] = Pr[pslot][g][pl];
        };
        auto chain = [&](float4 (&v)[BF_G], int sslot) {
            auto add = [&](int g) {
#ifdef BF_NO_CHAIN_ADD
                return;
#endif
                run = run + v[g].x; v[g].x = run;
                run = run + v[g].y; v[g].y = run;
                run = run + v[g].z; v[g].z = run;
                run = run + v[g].w; v[g].w = run;
            };
            auto put = [&](int g) {
#ifdef BF_NO_CHAIN_PUT
                return;
#endif
                Sr[HG + sslot * BF_G + g][pl] = v[g];
                if (g >= BF_G - HG && sslot == BF_NS - 1) Sr[g - (BF_G - HG)][pl] = v[g]; // the header (wave-uniform)
            };
#pragma unroll
            for (int g = 0; g < BF_G / 2; ++g) add(g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = BF_G / 2; g < BF_G; ++g) { add(g); put(g - BF_G / 2); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = BF_G / 2; g < BF_G; ++g) put(g);
        };
        auto step = [&](int pslot, int sslot) {
#ifdef BF_NO_CHAIN // removal timings (tools/build_variant.sh): profiles/r06_box_blur.txt
            return;
#endif
            if (lane < nl) {
                float4 v[BF_G];
                get(v, pslot);
                chain(v, sslot);
            }
        };
        BF_SYNC();
        BF_SYNC();
        int sslot = 0, blk = 0;
        static_assert(BF_NP == 2, "slots by parity below");
        for (; blk + 1 < nblocks; blk += 2) {
            BF_SYNC(); // step blk + 2
            step(0, sslot);
            sslot = sslot + 1 == BF_NS ? 0 : sslot + 1;
            BF_SYNC(); // step blk + 3
            step(1, sslot);
            sslot = sslot + 1 == BF_NS ? 0 : sslot + 1;
        }
        if (blk < nblocks) {
            BF_SYNC();
            step(0, sslot);
        }
        BF_SYNC(); // the last step: the means' last block
#ifdef BF_TIMING
        if (lane == 0 && k < 1024) { A.timing[(k * 16 + role) * 2] = bf_timer.busy; A.timing[(k * 16 + role) * 2 + 1] = bf_timer.wait; }
#endif
        return;
    }

    const bool edge = a0 < 0 || a0 + CL / C > cols; // workgroup-uniform: some chained columns lie outside the image
    if (role <= 6) { // ---- a loader ----------------------------------------------------------------------------------------------------------------
        const int li = role - 3;
        if (!edge) box_loader<C, 0>(A, Pr, li, lane, k, src, spitch, K, a0, nblocks, nsteps);
        else if (C == 4 || (cols & 3) == 0) box_loader<C, 1>(A, Pr, li, lane, k, src, spitch, K, a0, nblocks, nsteps);
        else if constexpr (C == 1) box_loader<C, 2>(A, Pr, li, lane, k, src, spitch, K, a0, nblocks, nsteps);
        return;
    }
    if (role >= 15) { // ---- a loader of the last two quarters -----------------------------------------------------------------------------------------
        if constexpr (C == 4) {
            if (!edge) box_loader_b<0>(A, Pr, role - 15, lane, k, src, spitch, K, a0, nblocks, nsteps);
            else box_loader_b<1>(A, Pr, role - 15, lane, k, src, spitch, K, a0, nblocks, nsteps);
        }
        return;
    }

    // ---- a mean wave -------------------------------------------------------------------------------------------------------------------------------
    // GPW of a block's sixteen groups each. Four channels: a lane = an output byte column, two groups per step. One channel: a lane = (one of four groups,
    // output column).
    const int mi = role - 7;
    const int gbase = mi * GPW * NPASS;  // the wave's first group inside a block
    // my group among the 64 / LPG the wave works on at once. One channel: the 16 lanes one cycle of a ds_read_b128 serves — {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31}, the same + 32 — work on one group (whole quads each, and lane & 15 takes every value once: the columns)
    const int l31 = lane & 31;
    const int gsel = C == 4 ? 0 : 2 * (lane >> 5) + (((l31 >= 4 && l31 < 12) || (l31 >= 16 && l31 < 20) || l31 >= 28) ? 1 : 0);
    const int gmine = gbase + gsel * NG; // my first group inside a block
    const int ml = lane % LPG;
    // my output byte column inside the strip. In the image's last strip the byte columns past its edge have nothing to store: their lanes repeat what
    // quad 0 does, address included, so that a store needs no execution mask (the same dword several times into one line). Only a row whose last bytes
    // are not a whole dword (one channel, cols % 4 != 0) takes the careful path.
    const int live_bytes = min(W, cols - x0) * C;      // workgroup-uniform
    const bool tail_strip = (live_bytes & 3) != 0;
    const int m = (ml >= live_bytes && !tail_strip) ? (ml & 3) : ml;
    const int pxi = m / C, ch = m - pxi * C;  // its pixel and channel
    const int c = x0 + pxi;
    const bool live = c < cols;
    const int c1 = max(c - R, 0), c2 = min(c + R, cols - 1);
    const int cw = max(c2 - c1 + 1, 1);
    // chain lanes of the two corner columns: c - R - 1 (zeros when negative: the reference's c1 == 0) and c2
    const int lb = min((pxi + LEFT - R - 1) * C + ch, CL - 1), la = min((max(c2, 0) - a0) * C + ch, CL - 1);
    const int pa = bf_pos(la), pb = bf_pos(lb);
    const float yrcp = 1.0f / (float)((2 * R + 1) * cw), nyrcp = -yrcp; // RN(1 / area) of the unclipped rows
    // after the transpose lane (quad, j) holds row j of the quad's four byte columns
    const int tj = m & 3, tq = m & ~3;
    const int row_bytes_left = live_bytes - tq; // bytes of the strip's output row from my quad on
    const bool store_dword = row_bytes_left >= 4;
    const int nbytes = max(min(row_bytes_left, 4), 0);           // 1..3: the image's last columns, when they are not a whole dword
    // from the first row of the wave's groups of a block (dpitch * 16 < 2^32: checked by the host)
    const uint32_t out_off = (uint32_t)(tj + 4 * NG * gsel) * (uint32_t)dpitch + (uint32_t)(x0 * C + tq);
    const uint32_t in_off = (uint32_t)(4 * NG * gsel) * (uint32_t)spitch + (uint32_t)min(x0 * C + m, cols * C - 1); // sharpen: my byte of a source row
    const uint32_t gen_in_off = (uint32_t)min(x0 * C + m, cols * C - 1);
    const uint32_t gen_off = (uint32_t)(x0 * C + m);
    // the 4 x 4 byte transpose across a quad: selectors of the two v_perm steps. v_perm_b32(D, X, sel): selector bytes 0..3 pick from X, 4..7 from D.
    const uint32_t sel1 = (lane & 1) ? 0x07030501u : 0x02060004u; // odd: {X1, D1, X3, D3}; even: {D0, X0, D2, X2}
    const uint32_t sel2 = (lane & 2) ? 0x07060302u : 0x01000504u; // lanes 2, 3: {Y2, Y3, T2, T3}; lanes 0, 1: {T0, T1, Y0, Y1}

    auto sat_at = [&](int row, int p) -> float { // any SAT row still in the ring
        const int blk = row / BF_B;
        const float *v = (const float *)&Sr[HG + (blk % BF_NS) * BF_G + ((row >> 2) & (BF_G - 1))][p];
        return v[row & 3];
    };
    auto generic_row = [&](int r) { // clipped windows: integral.zig:203-205, 254-266
        const int r1 = max(r - R, 0), r2 = min(r + R, rows - 1);
        const float a = sat_at(r2, pa), bb = sat_at(r2, pb);
        float d = 0.0f, e = 0.0f;
        if (r1 > 0) { d = sat_at(r1 - 1, pa); e = sat_at(r1 - 1, pb); }
        const float sum = ((a - bb) - d) + e;
        float val = box_quot(sum, box_div_of((float)((r2 - r1 + 1) * cw)));
        if constexpr (SHARPEN) {
            const float twice = 2 * (float)src[(size_t)r * spitch + gen_in_off];
            val = twice - val;
        }
        const uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(val + BF_BIAS, 0u, 0u);
        if (live && gsel == 0 && ml == m) dst[(size_t)r * dpitch + gen_off] = (uint8_t)pk;
    };

    // fast groups: SAT rows 4G .. 4G + 3 all exist and so do the d / e rows 2R + 1 above them; they give output rows 4G - R .. 4G - R + 3
    const int g_lo = HG, g_hi = rows / 4 - 1; // inclusive
    const bool any_fast = g_hi >= g_lo;
    const int top_end = any_fast ? 4 * g_lo - R : 0;          // generic rows [0, top_end)
    const int bot_start = any_fast ? 4 * (g_hi + 1) - R : 0;  // generic rows [bot_start, rows)

    auto groups = [&](int blk, int sslot, int gb) { // GPW groups of block blk from group gb on, one after the other, every lane on the same group (the first and the last
                                            // blocks, strips whose rows do not end on a dword): the lanes of gsel 0 store
        const int G0 = blk * BF_G + gb;
        if (G0 + GPW - 1 < g_lo || G0 > g_hi) return;
        float a[(HG + GPW) * 4], bq[(HG + GPW) * 4];
        const int first = HG + sslot * BF_G + gb - HG; // ring row of the first history group (the header when sslot == 0 and gb == 0)
#pragma unroll
        for (int h = 0; h < HG + GPW; ++h) {
            const float4 va = Sr[first + h][pa], vb = Sr[first + h][pb];
            a[4 * h] = va.x; a[4 * h + 1] = va.y; a[4 * h + 2] = va.z; a[4 * h + 3] = va.w;
            bq[4 * h] = vb.x; bq[4 * h + 1] = vb.y; bq[4 * h + 2] = vb.z; bq[4 * h + 3] = vb.w;
        }
#pragma unroll
        for (int t = 0; t < GPW; ++t) {
            const int Gt = G0 + t;
            if (Gt >= g_lo && Gt <= g_hi) { // wave-uniform
                const int r0 = 4 * Gt - R; // first output row of the group
                uint32_t pk = 0;
                float orig[4];
                if constexpr (SHARPEN) {
                    const uint8_t *srow = src + (size_t)r0 * spitch; // uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) orig[j] = (float)srow[(size_t)j * spitch + gen_in_off];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    constexpr int back = 2 * R + 1;
                    const int i = 4 * (HG + t) + j;
                    const float sum = ((a[i] - bq[i]) - a[i - back]) + bq[i - back]; // ((a - b) - d) + e, integral.zig:87-90
                    float val;
                    if constexpr (SHARPEN) val = __builtin_fmaf(sum, nyrcp, __builtin_fmaf(orig[j], 2.0f, BF_BIAS)); // 2 * original - sum / area (+ bias): integral.zig:308
                    else val = __builtin_fmaf(sum, yrcp, BF_BIAS);
                    pk = __builtin_amdgcn_cvt_pk_u8_f32(val, (uint32_t)j, pk);
                }
                // bytes = rows of my column -> bytes = the quad's columns of row tj
                const uint32_t x1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0xb1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
                const uint32_t t1 = __builtin_amdgcn_perm(pk, x1, sel1);
                const uint32_t y1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t1, 0x4e, 0xf, 0xf, false); // quad_perm [2,3,0,1]
                const uint32_t rowv = __builtin_amdgcn_perm(t1, y1, sel2);
                uint8_t *o = dst + (size_t)(r0 + tj) * dpitch + (size_t)(x0 * C + tq);
                if (gsel == 0) {
                    if (store_dword) *(uint32_t *)o = rowv;
                    else
                        for (int i = 0; i < nbytes; ++i) o[i] = (uint8_t)(rowv >> (8 * i));
                }
            }
        }
    };

    // The common step, trimmed to what it has to do: all of the wave's groups unclipped, the strip's rows ending on a dword. LDS addresses are byte offsets
    // into Sr (one addition per step for the slot), rows of the image are reached from wave-uniform pointers that move on by a block per step.
    constexpr uint32_t GROUP_BYTES = RS * 16, SLOT_BYTES = BF_G * GROUP_BYTES;
    const char *sr0 = (const char *)&Sr[0][0]; // the header's first group: HG groups above slot 0
    const uint32_t own_a = (uint32_t)pa * 16u + (uint32_t)gmine * GROUP_BYTES, own_b = (uint32_t)pb * 16u + (uint32_t)gmine * GROUP_BYTES; // my first HISTORY group in slot 0
    auto fast = [&](uint32_t so, uint8_t *orow, const uint8_t *srow) {
        const char *qa = sr0 + (own_a + so), *qb = sr0 + (own_b + so);
        float a[(HG + NG) * 4], bq[(HG + NG) * 4];
#pragma unroll
        for (int h = 0; h < HG + NG; ++h) {
            const float4 va = *(const float4 *)(qa + h * GROUP_BYTES);
            const float4 vb = *(const float4 *)(qb + h * GROUP_BYTES);
            a[4 * h] = va.x; a[4 * h + 1] = va.y; a[4 * h + 2] = va.z; a[4 * h + 3] = va.w;
            bq[4 * h] = vb.x; bq[4 * h + 1] = vb.y; bq[4 * h + 2] = vb.z; bq[4 * h + 3] = vb.w;
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            uint32_t pk = 0;
            float orig[4];
            if constexpr (SHARPEN) {
#pragma unroll
                for (int j = 0; j < 4; ++j) orig[j] = (float)srow[(size_t)(4 * t + j) * spitch + in_off];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                constexpr int back = 2 * R + 1;
                const int i = 4 * (HG + t) + j;
                const float sum = ((a[i] - bq[i]) - a[i - back]) + bq[i - back]; // ((a - b) - d) + e, integral.zig:87-90
                float val;
                if constexpr (SHARPEN) val = __builtin_fmaf(sum, nyrcp, __builtin_fmaf(orig[j], 2.0f, BF_BIAS)); // 2 * original - sum / area (+ bias): integral.zig:308
                else val = __builtin_fmaf(sum, yrcp, BF_BIAS);
                pk = __builtin_amdgcn_cvt_pk_u8_f32(val, (uint32_t)j, pk);
            }
            const uint32_t x1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0xb1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
            const uint32_t t1 = __builtin_amdgcn_perm(pk, x1, sel1);
            const uint32_t y1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t1, 0x4e, 0xf, 0xf, false); // quad_perm [2,3,0,1]
            *(uint32_t *)(orow + (out_off + (uint32_t)(4 * t) * (uint32_t)dpitch)) = __builtin_amdgcn_perm(t1, y1, sel2);
        }
    };

#ifdef BF_TIMING
    BfTimer bf_timer;
#endif
    for (int i = 0; i <= LAG; ++i) BF_SYNC();
    int sslot = 0;
    uint32_t so = 0;
    uint8_t *orow = dst + ((ptrdiff_t)(4 * gbase) - R) * (ptrdiff_t)dpitch; // first output row of the wave's groups of block 0 (negative rows are never touched)
    const uint8_t *srow = src + ((ptrdiff_t)(4 * gbase) - R) * (ptrdiff_t)spitch;
    auto advance = [&]() {
        sslot = sslot + 1 == BF_NS ? 0 : sslot + 1;
        so = so + SLOT_BYTES == BF_NS * SLOT_BYTES ? 0 : so + SLOT_BYTES;
        orow += (size_t)BF_B * dpitch;
        srow += (size_t)BF_B * spitch;
    };
    // block 0 (it holds the clipped rows at the top: the SAT rows they read, < 4 HG + 4, are all in it), then the blocks whose groups are all unclipped in a
    // loop without a condition, then what is left (the last block or two; every block of a strip whose rows do not end on a dword)
    const int fast_num = g_hi - gbase - GPW * NPASS + 1; // block blk of mine is unclipped iff 1 <= blk <= fast_num / groups per block
    const int last_fast = tail_strip || fast_num < 0 ? 0 : min(fast_num / BF_G, nblocks - 1);
    int blk = 0;
#ifndef BF_NO_MEANS
    BF_SYNC(); // step LAG + 1
    if (mi == NM - 1)
        for (int r = 0; r < min(top_end, rows); ++r) generic_row(r);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) groups(0, sslot, gbase + ps * GPW);
    advance();
    for (blk = 1; blk <= last_fast; ++blk) {
        BF_SYNC(); // step blk + LAG + 1
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) fast(so + (uint32_t)(ps * GPW) * GROUP_BYTES, orow + (size_t)(ps * 4 * GPW) * dpitch, srow + (size_t)(ps * 4 * GPW) * spitch);
        advance();
    }
    for (; blk < nblocks; ++blk) {
        BF_SYNC();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) groups(blk, sslot, gbase + ps * GPW);
        advance();
    }
#else
    for (; blk < nblocks; ++blk) BF_SYNC();
#endif
    // the clipped rows at the bottom: the ring still holds the last two blocks (every row they read is >= bot_start - R - 1)
#ifdef BF_TIMING
    if (lane == 0 && k < 1024) { A.timing[(k * 16 + role) * 2] = bf_timer.busy; A.timing[(k * 16 + role) * 2 + 1] = bf_timer.wait; }
#endif
    for (int r = max(bot_start, 0) + mi; r < rows; r += NM) generic_row(r);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------------------------
static bool box_fused_off() {
    static const bool off = getenv("ZIGNAL_HIP_BOX_UNFUSED") != nullptr; // the A/B of round 6 (profiles/r06_*)
    return off;
}

// -1: not this shape (the caller keeps the integral-image route)
int try_box_fused(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, uint32_t radius, bool sharpen, hipStream_t s) {
    if (box_fused_off()) return -1;
    if (src->pixel != ZG_PIXEL_U8 && src->pixel != ZG_PIXEL_RGBA_U8) return -1;
    if (radius < 1 || radius > (uint32_t)BF_MAX_R) return -1;
    const int C = pixel_channels(src->pixel);
    if (src->rows < 64 || src->cols < 64 || src->cols > 65536 || n == 0 || n > MAX_FRAMES_PER_LAUNCH) return -1; // 65536 * 255 < 2^24: exact row sums
    if (((uintptr_t)dst->data & 3) != 0 || ((size_t)dst->stride * C) % 4 != 0 || (dst_frame % 4) != 0) return -1;  // dword stores
    if (std::max((size_t)dst->stride, (size_t)src->stride) * C * 16 + (size_t)src->cols * C >= (1ull << 32)) return -1;    // 32-bit offsets inside a wave's sixteen rows
    const int nwg = (int)ceil_div(src->cols, (unsigned)BF_W), nk = nwg + 1;
    const size_t kbytes = (size_t)n * src->rows * nk * C * sizeof(float);
    // the in-place call (examples/src/face_alignment.zig:95): a strip's outputs would be read by its neighbours' chains, so the source is copied first
    const size_t px = pixel_size(src->pixel);
    const char *sb = (const char *)src->data, *se = sb + (size_t)(n - 1) * src_frame + ((size_t)(src->rows - 1) * src->stride + src->cols) * px;
    const char *db = (const char *)dst->data, *de = db + (size_t)(n - 1) * dst_frame + ((size_t)(dst->rows - 1) * dst->stride + dst->cols) * px;
    // dword loads: a source whose rows do not start on dwords is copied too (a view of a grey image at an odd column)
    const bool overlap = !(se <= db || de <= sb) || ((uintptr_t)src->data & 3) != 0 || ((size_t)src->stride * C) % 4 != 0 || (src_frame % 4) != 0;
    // 32-bit offsets inside a frame's rows and carries
    if ((size_t)src->rows * (overlap ? (size_t)src->cols + 3 : (size_t)src->stride) * C + 64 >= (1ull << 32) || (size_t)src->rows * nk * C * sizeof(float) >= (1ull << 32)) return -1;
    zg_image from = *src;
    size_t from_frame = src_frame;
    char *copy = nullptr;
    int rc;
    if (overlap) {
        const size_t cstride = ((size_t)src->cols + 3) & ~(size_t)3; // pixels: rows start on dwords
        const size_t frame_bytes = (size_t)src->rows * cstride * px;
        if ((rc = scratch_alloc((void **)&copy, (size_t)n * frame_bytes, s))) return rc;
        for (uint32_t f = 0; f < n; ++f) {
            zg_image a = *src, b2 = *src;
            a.data = (char *)src->data + (size_t)f * src_frame;
            b2.data = copy + (size_t)f * frame_bytes;
            b2.stride = (uint32_t)cstride;
            if ((rc = copy_impl(&a, &b2, s))) { scratch_free(copy, s); return rc; }
        }
        from.data = copy;
        from.stride = (uint32_t)cstride;
        from_frame = frame_bytes;
    }
    float *K = nullptr;
    if ((rc = scratch_alloc((void **)&K, kbytes, s))) { if (copy) scratch_free(copy, s); return rc; }
#ifdef BF_TIMING
    static unsigned long long *timing = nullptr;
    if (!timing) (void)hipMalloc((void **)&timing, 1024 * 16 * 2 * sizeof(unsigned long long));
    (void)hipMemsetAsync(timing, 0, 1024 * 16 * 2 * sizeof(unsigned long long), s);
    BoxFusedArgs A{dimg(&from), dimg(dst), K, nwg, nk, (int)ceil_div((unsigned)nwg, 8u), from_frame, dst_frame, timing};
#else
    BoxFusedArgs A{dimg(&from), dimg(dst), K, nwg, nk, (int)ceil_div((unsigned)nwg, 8u), from_frame, dst_frame};
#endif
    const dim3 grid(ZG_XCD_ORDER ? (unsigned)A.nwg8 * 8u : (unsigned)nwg, n);
    auto launch = [&](auto ctag, auto rtag) {
        constexpr int CC = decltype(ctag)::value, RR = decltype(rtag)::value;
        hipLaunchKernelGGL((k_box_carries<CC, box_strip_left(CC, RR)>), dim3(ceil_div(from.rows, 4u), n), dim3(256), 0, s, dimg(&from), K, nwg, nk, from_frame);
        if (sharpen) hipLaunchKernelGGL((k_box_fused<CC, RR, true>), grid, dim3(BoxGeo<CC>::THREADS), 0, s, A);
        else hipLaunchKernelGGL((k_box_fused<CC, RR, false>), grid, dim3(BoxGeo<CC>::THREADS), 0, s, A);
    };
    auto by_radius = [&](auto ctag) {
        switch (radius) {
        case 1: launch(ctag, std::integral_constant<int, 1>{}); break;
        case 2: launch(ctag, std::integral_constant<int, 2>{}); break;
        default: launch(ctag, std::integral_constant<int, 3>{}); break;
        }
    };
    if (C == 4) by_radius(std::integral_constant<int, 4>{});
    else by_radius(std::integral_constant<int, 1>{});
    const hipError_t e = hipGetLastError();
#ifdef BF_TIMING
    {
        static int calls = 0;
        if (++calls % 50 == 0) { // a warm call
            (void)hipStreamSynchronize(s);
            static unsigned long long host[1024 * 16 * 2];
            (void)hipMemcpy(host, timing, sizeof(host), hipMemcpyDeviceToHost);
            for (int kk : {0, 1, nwg / 2, nwg - 1}) {
                printf("strip %d of %d (C=%d r=%u): busy / wait cycles per role (1, 2 chain; 3.. loaders; 7.. means):", kk, nwg, C, radius);
                for (int w = 0; w < 16; ++w) printf(" %d:%llu/%llu", w, host[(kk * 16 + w) * 2], host[(kk * 16 + w) * 2 + 1]);
                printf("\n");
            }
        }
    }
#endif
    scratch_free(K, s);
    if (copy) scratch_free(copy, s);
    if (e != hipSuccess) { set_error("boxBlur: launch failed: %s", hipGetErrorString(e)); return ZG_ERR_HIP; }
    return ZG_OK;
}

} // namespace zg
