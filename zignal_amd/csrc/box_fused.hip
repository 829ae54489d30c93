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
//   k_box_carries   one workgroup per row: K[r][k][ch] = sum of row r left of workgroup k's first chained column (rows x strips x C floats)
//   k_box_fused     a byte column (pixel column x channel) per chain lane, 64 lanes per workgroup; sixteen waves dealt out by role (wave w runs on
//                   SIMD w % 4; waves 4, 8, 12 leave at once so that the chain has SIMD 0 to itself):
//       4 LOADERS   lane = (group of four rows, 16 adjacent bytes of them): one load per row, bytes -> f32, prefix inside the lane along each channel,
//                   scan over the four lanes of a row (DPP), + carry. A loader owns two of a group's four rows in every other 64-row block, one row
//                   per step, and stores the pair as half of the chain lanes' LDS vectors.
//       1 CHAIN     per four rows: ds_read_b128 of row prefixes, four dependent v_add_f32 (THE sequential part: rows x 1 addition), ds_write_b128
//                   of SAT values into a ring of three 64-row blocks.
//       8 MEANS     two groups of four rows per step each: corners a, b of a row are one ds_read_b128 per four rows and lane, corners d, e are
//                   the a, b of 2R + 1 rows earlier and stay in registers; ((a - b) - d) + e, then division, rounding, clamp and pack as ONE fma and
//                   one v_cvt_pk_u8_f32 (below),
//                   a 4 x 4 byte transpose across the quad (two DPP moves, two v_perm) so that every lane stores one aligned dword of one row.
//   Steps are separated by one s_barrier: at step s the loaders work on blocks s and s - 1, the chain turns block s - 2 into SAT rows, the means
//   finish block s - 3. What bounds a step is the instruction count of its longest wave (a wave issues about one instruction per ten cycles whatever
//   its neighbours do: profiles/r06_box_removal.txt), hence many waves with ~100 instructions per step each. The first R + 1 .. and the last R rows, whose windows are clipped, take a generic path (any row, LDS reads per corner).
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

constexpr int BF_B = 64;        // rows per block
constexpr int BF_G = BF_B / 4;  // four-row groups per block
constexpr int BF_RS = 68;       // 16-byte units from one group to the next in LDS: four quarters of 16 chain lanes + 1 (see bf_pos)
constexpr int BF_NP = 2;        // row-prefix ring: block s - 1 is written (at the end of its loaders' second step) while block s - 2 is read
constexpr int BF_NS = 3;        // SAT ring: block s - 2 is written while the means read s - 3 and its predecessor (history rows)
constexpr int BF_THREADS = 1024;
constexpr int BF_NM = 8;        // mean waves
constexpr int BF_MAX_R = 3;

// A strip is 64 byte columns of the image's rows (a byte column = pixel column x channel): 16 Rgba(u8) pixels, 64 grey ones. LEFT of them lie left of
// the first output column: the R + 1 a window reaches back (rounded up to a whole dword for one-channel images, so that every loader lane's 16 bytes start
// on a dword); W = outputs per strip, a whole number of dwords.
__host__ __device__ constexpr int box_strip_left(int C, int R) { return C == 4 ? R + 1 : 4; }
__host__ __device__ constexpr int box_strip_w(int C, int R) { return C == 4 ? 16 - (2 * R + 1) : 56; }

// chain lane -> 16-byte slot inside a group row. A loader lane (four-row group rg, quarter q) stores byte i of its 16 at rg * 68 + q * 17 + i: for a fixed
// i the eight lanes one LDS store serves together (two groups x four quarters) fall into eight different bank groups (4 (rg & 1) + q + i mod 8), and the
// sixteen stores of a lane differ by immediate offsets only (a permutation by XOR needed sixteen address registers and spilled).
__device__ __forceinline__ int bf_pos(int l) { return l + (l >> 4); }

struct BoxFusedArgs {
    DImg src, dst;
    const float *carries; // [frame][row][strip][C]
    int nwg;              // strips
    int nwg8;             // ceil(nwg / 8): strips per XCD
    size_t src_frame, dst_frame;
#ifdef BF_TIMING
    unsigned long long *timing; // [strip][wave][2]: cycles a wave spent between barriers, cycles it waited in them (tools/build_variant.sh ... -DBF_TIMING)
#endif
};

// ---- carries ---------------------------------------------------------------------------------------------------------------------------------------
// K[(r * nwg + k) * C + ch] = sum over columns < k * W - LEFT (strip k's first chained column) of row r, channel ch (0 for k = 0), exact in f32.
// One WAVE per row, no LDS and no barrier. A lane takes one PIECE of the row — the W columns from one strip's first chained column to the next one's — and
// sums it by itself (packed 16-bit pairs for four channels: 13 x 255 < 2^16; v_sad_u8 for one channel), so that the wave needs one scan per 64 pieces
// (64 W pixels) instead of one per 256 pixels: the first versions (a block scan per 1024 pixels with a barrier; then a wave scan per 256 pixels) took 22 us
// for a 4096^2 Rgba(u8) frame whatever their loads did — 85 instructions per 256 pixels, bound by issue.
template <int C, int W, int LEFT>
__global__ __launch_bounds__(256) void k_box_carries(DImg src, float *K, int nwg, size_t src_frame) {
    src.data = (char *)src.data + (size_t)blockIdx.y * src_frame;
    K += (size_t)blockIdx.y * src.rows * nwg * C;
    const int lane = threadIdx.x & 63;
    const int r = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= src.rows) return;
    const uint8_t *row = (const uint8_t *)src.data + (size_t)r * src.stride * C; // 4-byte aligned (checked by the host)
    float *Kr = K + (size_t)r * nwg * C;
    if (lane < C) Kr[lane] = 0.0f; // strip 0
    uint32_t carry[C];                // the row's total left of the batch (wave-uniform)
#pragma unroll
    for (int ch = 0; ch < C; ++ch) carry[ch] = 0;
    constexpr int ND = W * C / 4; // dwords of a piece
    static_assert((W * C) % 4 == 0 && (LEFT * C) % 4 == 0, "pieces are whole dwords");
    // piece p = columns [p W - LEFT, (p + 1) W - LEFT): it ends where strip p + 1 starts, and the last one needed (p = nwg - 2) lies inside the row;
    // only piece 0 starts left of the row (its first LEFT columns do not exist: masked)
    for (int p0 = 0; p0 < nwg - 1; p0 += 64) {
        const int p = min(p0 + lane, nwg - 2);
        const uint32_t *d = (const uint32_t *)(row + ((ptrdiff_t)p * W - LEFT) * C);
        uint32_t v[ND];
        if (p0 == 0) { // wave-uniform: lane 0's piece starts at column -LEFT
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const bool real = p > 0 || i >= LEFT * C / 4;
                v[i] = d[real ? i : LEFT * C / 4] & (real ? 0xffffffffu : 0u);
            }
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
        if (p0 + lane < nwg - 1) {
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

// ---- the fused kernel -------------------------------------------------------------------------------------------------------------------------------
// The byte of a mean. Only the BYTE has to equal the reference's, and a window sum is an integer (sums and differences of integer-valued floats): sum / area
// is then an exact tie k + 1/2 or at least 1 / (2 area) >= 1 / 98 away from one, far more than the error of one multiplication by the correctly rounded
// reciprocal. So the division is ONE fma, the rounding bias riding in its addend: v_cvt_pk_u8_f32(fma(sum, 1 / area, 2^-10)). tools/exp/box_quot_check.hip
// compares it with the IEEE division sequence over EVERY integer-valued f32 sum below 2^34, every area h x w (h, w <= 7), blur and sharpen: no byte differs
// (profiles/r06_box_quot_check.txt; Markstein's three-operation quotient, also checked there, is not needed).
template <int C, bool EDGE>
__device__ __forceinline__ void box_loader(const BoxFusedArgs &A, float4 (*Pr)[BF_G][BF_RS], int li, int lane, int k, const uint8_t *src, size_t spitch, const float *K, int a0,
                                           int nblocks, int nsteps) {
#ifdef BF_TIMING
    BfTimer bf_timer;
#endif
    const int rows = A.src.rows, cols = A.src.cols;
    const int par = li & 1, h = li >> 1;    // my blocks: par, par + 2, ...; my rows of every four-row group: 2h, 2h + 1
    const int q = lane & 3, rg = lane >> 2; // my 16 bytes of the strip's rows, my four-row group of a block
    const int byte0 = a0 * C + 16 * q;      // first byte column (of the row) of mine
    float fm[2];                            // lane-scan masks: 1 where the lane 1 / 2 to the left belongs to the same row
    fm[0] = q >= 1 ? 1.0f : 0.0f;
    fm[1] = q >= 2 ? 1.0f : 0.0f;
    uint32_t raw[2][4]; // [row][dword]
    float kk[2][C];
    // edge strips: clamped offsets and AND masks per dword (a select on the loaded value would be turned into a branch around the load, and loads
    // under a branch are waited for one by one)
    uint32_t eoff[4], emask[4], emask2[4], poff[3], pmask[3];
    if constexpr (EDGE) {
        const int cols4 = cols & ~3;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if constexpr (C == 4) {
                const int cc = a0 + 4 * q + d; // my pixel
                eoff[d] = (uint32_t)min(max(cc, 0), cols - 1) * 4u;
                emask[d] = cc >= 0 && cc < cols ? 0xffffffffu : 0u;
                emask2[d] = 0;
            } else {
                const int cc = byte0 + 4 * d; // a multiple of 4: a0 and W are
                eoff[d] = (uint32_t)min(max(cc, 0), cols4 - 4);
                emask[d] = cc >= 0 && cc + 4 <= cols ? 0xffffffffu : 0u;
                emask2[d] = cc == cols4 ? 0xffffffffu : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            poff[i] = (uint32_t)min(cols4 + i, cols - 1);
            pmask[i] = cols4 + i < cols ? 0xffu : 0u;
        }
        // keep the masks as register values: turned back into compares they take eight scalar registers that the kernel does not have
#pragma unroll
        for (int d = 0; d < 4; ++d) asm volatile("" : "+v"(emask[d]), "+v"(emask2[d]));
    }
    auto fetch_at = [&](const uint8_t *rowp, const float *kp, int u) { // unpredicated, unconditional
        if constexpr (!EDGE) {
            const uint32_t *p4 = (const uint32_t *)(rowp + byte0); // 4-byte aligned; the 16 bytes may straddle a line
            raw[u][0] = p4[0]; raw[u][1] = p4[1]; raw[u][2] = p4[2]; raw[u][3] = p4[3];
        } else if constexpr (C == 4) { // a pixel is a dword: inside or outside as a whole
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[u][i] = *(const uint32_t *)(rowp + eoff[i]) & emask[i];
        } else { // whole dwords where they lie inside the row; the row's last, partial dword (the same for every lane) from its bytes
            uint32_t part = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) part |= ((uint32_t)rowp[poff[i]] & pmask[i]) << (8 * i);
#pragma unroll
            for (int d = 0; d < 4; ++d) raw[u][d] = (*(const uint32_t *)(rowp + eoff[d]) & emask[d]) | (part & emask2[d]);
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) kk[u][ch] = kp[ch];
    };
    auto fetch = [&](int blk, int u) { // row 2h + u of my group in block blk, clamped into the image
#if defined(BF_NO_LOADERS) || defined(BF_NO_FETCH)
        return;
#endif
        const int r = min(blk * BF_B + rg * 4 + 2 * h + u, rows - 1);
        fetch_at(src + (size_t)r * spitch, K + ((size_t)r * A.nwg + k) * C, u);
    };
    auto prefixes = [&](int u, float (&val)[16]) { // the row prefixes of my 16 bytes of row 2h + u
        float e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = (float)((raw[u][i >> 2] >> (8 * (i & 3))) & 0xffu); // v_cvt_f32_ubyteN
        // prefix inside the lane along each channel (bytes C apart): all sums are integers below 2^24, exact in any order
#pragma unroll
        for (int i = C; i < 16; ++i) e[i] = e[i - C] + e[i];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const float t = e[16 - C + ch]; // my total of this channel
            float y = t;                    // inclusive scan over the four lanes of my row
            // (every lane of a quad is live: the move's `old` operand is never used, and naming the source itself saves a v_mov 0 per move)
            y = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, y), __builtin_bit_cast(int, y), 0x90, 0xf, 0xf, false)), fm[0], y); // quad_perm [0,0,1,2]
            y = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, y), __builtin_bit_cast(int, y), 0x44, 0xf, 0xf, false)), fm[1], y); // quad_perm [0,1,0,1]
            const float base = (y - t) + kk[u][ch]; // the lanes to my left + everything left of the strip
#pragma unroll
            for (int i = ch; i < 16; i += C) val[i] = base + e[i];
        }
    };
    // Loader li owns rows 2h, 2h + 1 of the groups of blocks par, par + 2, ...: the first row at step blk (kept in registers), the second at step blk + 1,
    // then both as the low or the high half of 16 chain lanes' LDS vectors; right behind each row the load of the same row of its next block. No
    // condition inside the loop: with one the compiler waits for the loads at the loop's end.
    fetch(par, 0);
    fetch(par, 1);
    int done = 0;
    if (par == 1) { BF_SYNC(); done = 1; }
    auto step_a = [&](float (&va)[16]) {
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_PUBLISH)
        prefixes(0, va);
#endif
    };
    auto step_b = [&](const float (&va)[16]) {
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_PUBLISH)
        float vb[16];
        prefixes(1, vb);
        float2 *o = (float2 *)&Pr[par][rg][17 * q] + h; // = bf_pos(16 q + i) - i; my blocks sit in slot blk & 1 = par
#pragma unroll
        for (int i = 0; i < 16; ++i) o[2 * i] = make_float2(va[i], vb[i]);
#endif
    };
    int blk = par;
    // while the block fetched next (blk + 2) is whole, its rows are reached by moving two pointers on (the clamped form multiplies: quarter-rate instructions)
    const int nfull = rows / BF_B;
    const size_t row_step = (size_t)2 * BF_B * spitch, k_step = (size_t)2 * BF_B * A.nwg * C;
    const int r_next = (par + 2) * BF_B + rg * 4 + 2 * h; // my first row of block par + 2 (used only when that block is whole)
    const uint8_t *rp = src + (size_t)min(r_next, rows - 1) * spitch;
    const float *kq = K + ((size_t)min(r_next, rows - 1) * A.nwg + k) * C;
    for (; blk + 2 < nfull; blk += 2) {
        float va[16];
        BF_SYNC(); // step blk
        step_a(va);
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_FETCH)
        fetch_at(rp, kq, 0);
#endif
        BF_SYNC(); // step blk + 1
        step_b(va);
#if !defined(BF_NO_LOADERS) && !defined(BF_NO_FETCH)
        fetch_at(rp + spitch, kq + (size_t)A.nwg * C, 1);
#endif
        rp += row_step;
        kq += k_step;
        done += 2;
    }
    for (; blk < nblocks; blk += 2) { // the last blocks: what they fetch is partial or past the end (clamped, never used)
        float va[16];
        BF_SYNC(); // step blk
        step_a(va);
        fetch(blk + 2, 0);
        BF_SYNC(); // step blk + 1
        step_b(va);
        fetch(blk + 2, 1);
        done += 2;
    }
    for (; done < nsteps; ++done) BF_SYNC();
#ifdef BF_TIMING
    if (lane == 0) { A.timing[(k * 16 + (li == 3 ? 7 : li + 1)) * 2] = bf_timer.busy; A.timing[(k * 16 + (li == 3 ? 7 : li + 1)) * 2 + 1] = bf_timer.wait; }
#endif
}

template <int C, int R, bool SHARPEN>
__global__ __launch_bounds__(BF_THREADS) void k_box_fused(BoxFusedArgs A) {
    constexpr int W = box_strip_w(C, R), LEFT = box_strip_left(C, R);
    constexpr int HG = (2 * R + 1 + 3) / 4; // groups of history a group's d / e rows reach back into
    __shared__ float4 Pr[BF_NP][BF_G][BF_RS];
    __shared__ float4 Sr[BF_NS][BF_G][BF_RS];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // role of a wave (SIMD = wave % 4): 0 the chain; 4, 8, 12 leave (same SIMD); 1, 2, 3, 7 loaders; the other eight are means
    if (wave == 4 || wave == 8 || wave == 12) return;
    // strips of one XCD are neighbours: they share source lines (the re-chained columns) and the halves of output lines in that XCD's L2
    const int b = (int)blockIdx.x;
    const int k = ZG_XCD_ORDER ? (b & 7) * A.nwg8 + (b >> 3) : b;
    if (k >= A.nwg) return;
    const int frame = (int)blockIdx.y;
    const int rows = A.src.rows, cols = A.src.cols;
    const uint8_t *src = (const uint8_t *)A.src.data + (size_t)frame * A.src_frame;
    uint8_t *dst = (uint8_t *)A.dst.data + (size_t)frame * A.dst_frame;
    const size_t spitch = (size_t)A.src.stride * C, dpitch = (size_t)A.dst.stride * C;
    const float *K = A.carries + (size_t)frame * rows * A.nwg * C;
    const int nblocks = (rows + BF_B - 1) / BF_B;
    const int nsteps = nblocks + 3; // loaders at s = blk, blk + 1; chain at blk + 2; means at blk + 3
    const int x0 = k * W;          // first output column
    const int a0 = x0 - LEFT;      // first chained column (may be negative: those columns hold zeros, which is what the reference's c1 == 0 case reads)

    if (wave == 0) { // ---- the chain ---------------------------------------------------------------------------------------------------------------
#ifdef BF_TIMING
        BfTimer bf_timer;
#endif
        float run = 0.0f;
        const int pl = bf_pos(lane);
        BF_SYNC();
        BF_SYNC();
        int pslot = 0, sslot = 0;
        for (int blk = 0; blk < nblocks; ++blk) {
            BF_SYNC(); // step blk + 2
#ifdef BF_NO_CHAIN // removal timings (tools/build_variant.sh): profiles/r06_box_removal.txt
            continue;
#endif
            const float4 *p = &Pr[pslot][0][pl];
            float4 *o = &Sr[sslot][0][pl];
            float4 v[BF_G];
#pragma unroll
            for (int g = 0; g < BF_G; ++g) v[g] = p[g * BF_RS];
#pragma unroll
            for (int g = 0; g < BF_G; ++g) {
                float4 qv;
                run = run + v[g].x; qv.x = run;
                run = run + v[g].y; qv.y = run;
                run = run + v[g].z; qv.z = run;
                run = run + v[g].w; qv.w = run;
                o[g * BF_RS] = qv;
            }
            pslot = pslot + 1 == BF_NP ? 0 : pslot + 1;
            sslot = sslot + 1 == BF_NS ? 0 : sslot + 1;
        }
        BF_SYNC(); // step nblocks + 2
#ifdef BF_TIMING
        if (lane == 0) { A.timing[(k * 16) * 2] = bf_timer.busy; A.timing[(k * 16) * 2 + 1] = bf_timer.wait; }
#endif
        return;
    }

    if (wave == 1 || wave == 2 || wave == 3 || wave == 7) { // ---- a loader --------------------------------------------------------------------------------
        const int li = wave == 7 ? 3 : wave - 1;
        const bool edge = a0 < 0 || a0 + 64 / C > cols; // workgroup-uniform: some chained columns lie outside the image
        if (edge) box_loader<C, true>(A, Pr, li, lane, k, src, spitch, K, a0, nblocks, nsteps);
        else box_loader<C, false>(A, Pr, li, lane, k, src, spitch, K, a0, nblocks, nsteps);
        return;
    }

    // ---- a mean wave -------------------------------------------------------------------------------------------------------------------------------
    // two of a block's sixteen groups each. SIMD 1: waves 5, 9, 13 (+ loader 1); SIMD 2: 6, 10, 14 (+ loader 2); SIMD 3: 11, 15 (+ loaders 3, 7)
    const int mi = wave == 5 ? 0 : wave == 9 ? 1 : wave == 13 ? 2 : wave == 6 ? 3 : wave == 10 ? 4 : wave == 14 ? 5 : wave == 11 ? 6 : 7;
    const int gfirst = 2 * mi;
    // my output byte column inside the strip. The byte columns past the strip's outputs (the re-chained neighbours; in the image's last strip also the
    // columns past its edge) have nothing to store: their lanes repeat what quad 0 does, address included, so that a store needs no execution mask
    // (the same dword several times into one line). Only a row whose last bytes are not a whole dword (one channel, cols % 4 != 0) takes the careful path.
    const int live_bytes = min(W, cols - x0) * C;      // workgroup-uniform
    const bool tail_strip = (live_bytes & 3) != 0;
    const int m = (lane >= live_bytes && !tail_strip) ? (lane & 3) : lane;
    const int pxi = m / C, ch = m - pxi * C;  // its pixel and channel
    const int c = x0 + pxi;
    const bool live = m < W * C && c < cols;
    const int c1 = max(c - R, 0), c2 = min(c + R, cols - 1);
    const int cw = max(c2 - c1 + 1, 1);
    // chain lanes of the two corner columns: c - R - 1 (zeros when negative: the reference's c1 == 0) and c2
    const int lb = min((pxi + LEFT - R - 1) * C + ch, 63), la = min((max(c2, 0) - a0) * C + ch, 63);
    const int pa = bf_pos(la), pb = bf_pos(lb);
    const float yrcp = 1.0f / (float)((2 * R + 1) * cw), nyrcp = -yrcp; // RN(1 / area) of the unclipped rows
    // after the transpose lane (quad, j) holds row j of the quad's four byte columns
    const int tj = m & 3, tq = m & ~3;
    const int row_bytes_left = min(W * C, (cols - x0) * C) - tq; // bytes of the strip's output row from my quad on
    const bool store_dword = row_bytes_left >= 4;
    const int nbytes = max(min(row_bytes_left, 4), 0);           // 1..3: the image's last columns, when they are not a whole dword
    const uint32_t out_off = (uint32_t)tj * (uint32_t)dpitch + (uint32_t)(x0 * C + tq); // from the first row of a group (dpitch * 8 < 2^32: checked by the host)
    const uint32_t in_off = (uint32_t)min(x0 * C + m, cols * C - 1);                   // sharpen: my byte of a source row
    const uint32_t gen_off = (uint32_t)(x0 * C + m);
    // the 4 x 4 byte transpose across a quad: selectors of the two v_perm steps. v_perm_b32(D, X, sel): selector bytes 0..3 pick from X, 4..7 from D.
    const uint32_t sel1 = (lane & 1) ? 0x07030501u : 0x02060004u; // odd: {X1, D1, X3, D3}; even: {D0, X0, D2, X2}
    const uint32_t sel2 = (lane & 2) ? 0x07060302u : 0x01000504u; // lanes 2, 3: {Y2, Y3, T2, T3}; lanes 0, 1: {T0, T1, Y0, Y1}

    auto sat_at = [&](int row, int p) -> float { // any SAT row still in the ring
        const int blk = row >> 6;
        const float *v = (const float *)&Sr[blk % BF_NS][(row >> 2) & (BF_G - 1)][p];
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
            const float twice = 2 * (float)src[(size_t)r * spitch + in_off];
            val = twice - val;
        }
        const uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(val + BF_BIAS, 0u, 0u);
        if (live) dst[(size_t)r * dpitch + gen_off] = (uint8_t)pk;
    };

    // fast groups: SAT rows 4G .. 4G + 3 all exist and so do the d / e rows 2R + 1 above them; they give output rows 4G - R .. 4G - R + 3
    const int g_lo = HG, g_hi = rows / 4 - 1; // inclusive
    const bool any_fast = g_hi >= g_lo;
    const int top_end = any_fast ? 4 * g_lo - R : 0;          // generic rows [0, top_end)
    const int bot_start = any_fast ? 4 * (g_hi + 1) - R : 0;  // generic rows [bot_start, rows)

    auto groups = [&](auto ntag, int blk, int sslot) { // my ntag groups of block blk
        constexpr int NGR = decltype(ntag)::value;
        const int G0 = blk * BF_G + gfirst;
        if (G0 + NGR - 1 < g_lo || G0 > g_hi) return;
        float a[(HG + NGR) * 4], bq[(HG + NGR) * 4];
        const int prev = sslot == 0 ? BF_NS - 1 : sslot - 1;
#pragma unroll
        for (int h = 0; h < HG + NGR; ++h) {
            const int gi = gfirst - HG + h; // group inside the block; negative: the block before
            const int sl = gi < 0 ? prev : sslot;
            const float4 va = Sr[sl][gi & (BF_G - 1)][pa], vb = Sr[sl][gi & (BF_G - 1)][pb];
            a[4 * h] = va.x; a[4 * h + 1] = va.y; a[4 * h + 2] = va.z; a[4 * h + 3] = va.w;
            bq[4 * h] = vb.x; bq[4 * h + 1] = vb.y; bq[4 * h + 2] = vb.z; bq[4 * h + 3] = vb.w;
        }
#pragma unroll
        for (int t = 0; t < NGR; ++t) {
            const int Gt = G0 + t;
            if (Gt >= g_lo && Gt <= g_hi) { // wave-uniform
                const int r0 = 4 * Gt - R; // first output row of the group
                uint32_t pk = 0;
                float orig[4];
                if constexpr (SHARPEN) {
                    const uint8_t *srow = src + (size_t)r0 * spitch; // uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) orig[j] = (float)srow[(size_t)j * spitch + in_off];
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
                const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp((int)pk, (int)pk, 0xb1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
                const uint32_t t1 = __builtin_amdgcn_perm(pk, x1, sel1);
                const uint32_t y1 = (uint32_t)__builtin_amdgcn_update_dpp((int)t1, (int)t1, 0x4e, 0xf, 0xf, false); // quad_perm [2,3,0,1]
                const uint32_t rowv = __builtin_amdgcn_perm(t1, y1, sel2);
                uint8_t *orow = dst + (size_t)r0 * dpitch; // uniform
                if (store_dword) *(uint32_t *)(orow + out_off) = rowv;
                else
                    for (int i = 0; i < nbytes; ++i) orow[out_off + i] = (uint8_t)(rowv >> (8 * i));
            }
        }
    };

    // The common step, trimmed to what it has to do: both groups unclipped, the strip inside the image. LDS addresses are byte offsets into Sr (one
    // addition per step for the slot), rows of the image are reached from wave-uniform pointers that move on by a block per step, the history rows of
    // wave 0 (groups 14, 15 of the block before) sit in the previous slot.
    constexpr uint32_t GROUP_BYTES = BF_RS * 16, SLOT_BYTES = BF_G * GROUP_BYTES;
    const char *sr0 = (const char *)&Sr[0][0][0];
    const uint32_t own_a = (uint32_t)pa * 16u + (uint32_t)gfirst * GROUP_BYTES, own_b = (uint32_t)pb * 16u + (uint32_t)gfirst * GROUP_BYTES;
    const uint32_t out_off2 = out_off + 4u * (uint32_t)dpitch;
    auto fast2 = [&](uint32_t so, uint32_t so_prev, uint8_t *orow, const uint8_t *srow) {
        const uint32_t oa = own_a + so, ob = own_b + so;
        const uint32_t ha = mi == 0 ? (uint32_t)pa * 16u + (BF_G - HG) * GROUP_BYTES + so_prev : oa - HG * GROUP_BYTES;
        const uint32_t hb = mi == 0 ? (uint32_t)pb * 16u + (BF_G - HG) * GROUP_BYTES + so_prev : ob - HG * GROUP_BYTES;
        float a[(HG + 2) * 4], bq[(HG + 2) * 4];
#pragma unroll
        for (int h = 0; h < HG + 2; ++h) {
            const float4 va = *(const float4 *)(sr0 + (h < HG ? ha + h * GROUP_BYTES : oa + (h - HG) * GROUP_BYTES));
            const float4 vb = *(const float4 *)(sr0 + (h < HG ? hb + h * GROUP_BYTES : ob + (h - HG) * GROUP_BYTES));
            a[4 * h] = va.x; a[4 * h + 1] = va.y; a[4 * h + 2] = va.z; a[4 * h + 3] = va.w;
            bq[4 * h] = vb.x; bq[4 * h + 1] = vb.y; bq[4 * h + 2] = vb.z; bq[4 * h + 3] = vb.w;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
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
            const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp((int)pk, (int)pk, 0xb1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
            const uint32_t t1 = __builtin_amdgcn_perm(pk, x1, sel1);
            const uint32_t y1 = (uint32_t)__builtin_amdgcn_update_dpp((int)t1, (int)t1, 0x4e, 0xf, 0xf, false); // quad_perm [2,3,0,1]
            *(uint32_t *)(orow + (t == 0 ? out_off : out_off2)) = __builtin_amdgcn_perm(t1, y1, sel2);
        }
    };

#ifdef BF_TIMING
    BfTimer bf_timer;
#endif
    BF_SYNC();
    BF_SYNC();
    BF_SYNC();
    int sslot = 0;
    uint32_t so = 0, so_prev = (BF_NS - 1) * SLOT_BYTES;
    uint8_t *orow = dst + ((ptrdiff_t)(4 * gfirst) - R) * (ptrdiff_t)dpitch; // first output row of my groups of block 0 (negative rows are never touched)
    const uint8_t *srow = src + ((ptrdiff_t)(4 * gfirst) - R) * (ptrdiff_t)spitch;
    for (int blk = 0; blk < nblocks; ++blk) {
        BF_SYNC(); // step blk + 3
#ifdef BF_NO_MEANS
        continue;
#endif
        if (blk == 0 && mi == BF_NM - 1) { // the clipped rows at the top: the SAT rows they read (< 4 HG + 4) are all in block 0
            for (int r = 0; r < min(top_end, rows); ++r) generic_row(r);
        }
        const int G0 = blk * BF_G + gfirst;
        if (!tail_strip && G0 >= g_lo && G0 + 1 <= g_hi) fast2(so, so_prev, orow, srow);
        else groups(std::integral_constant<int, 2>{}, blk, sslot);
        sslot = sslot + 1 == BF_NS ? 0 : sslot + 1;
        so_prev = so;
        so = so + SLOT_BYTES == BF_NS * SLOT_BYTES ? 0 : so + SLOT_BYTES;
        orow += (size_t)BF_B * dpitch;
        srow += (size_t)BF_B * spitch;
    }
    // the clipped rows at the bottom: the ring still holds the last two blocks (every row they read is >= bot_start - R - 1)
#ifdef BF_TIMING
    if (lane == 0) { A.timing[(k * 16 + wave) * 2] = bf_timer.busy; A.timing[(k * 16 + wave) * 2 + 1] = bf_timer.wait; }
#endif
    for (int r = max(bot_start, 0) + mi; r < rows; r += BF_NM) generic_row(r);
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
    const int W = box_strip_w(C, (int)radius);
    if (src->rows < 64 || src->cols < 64 || src->cols > 65536 || n == 0 || n > MAX_FRAMES_PER_LAUNCH) return -1; // 65536 * 255 < 2^24: exact row sums
    if (((uintptr_t)dst->data & 3) != 0 || ((size_t)dst->stride * C) % 4 != 0 || (dst_frame % 4) != 0) return -1;  // dword stores
    if ((size_t)dst->stride * C * 8 + (size_t)src->cols * C >= (1ull << 32)) return -1;                            // 32-bit offsets inside a wave's two groups of four rows
    const int nwg = (int)ceil_div(src->cols, (unsigned)W);
    const size_t kbytes = (size_t)n * src->rows * nwg * C * sizeof(float);
    // the in-place call (examples/src/face_alignment.zig:95): a strip's outputs would be read by its neighbours' chains, so the source is copied first
    const size_t px = pixel_size(src->pixel);
    const char *sb = (const char *)src->data, *se = sb + (size_t)(n - 1) * src_frame + ((size_t)(src->rows - 1) * src->stride + src->cols) * px;
    const char *db = (const char *)dst->data, *de = db + (size_t)(n - 1) * dst_frame + ((size_t)(dst->rows - 1) * dst->stride + dst->cols) * px;
    // dword loads: a source whose rows do not start on dwords is copied too (a view of a grey image at an odd column)
    const bool overlap = !(se <= db || de <= sb) || ((uintptr_t)src->data & 3) != 0 || ((size_t)src->stride * C) % 4 != 0 || (src_frame % 4) != 0;
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
    BoxFusedArgs A{dimg(&from), dimg(dst), K, nwg, (int)ceil_div((unsigned)nwg, 8u), from_frame, dst_frame, timing};
#else
    BoxFusedArgs A{dimg(&from), dimg(dst), K, nwg, (int)ceil_div((unsigned)nwg, 8u), from_frame, dst_frame};
#endif
    const dim3 grid(ZG_XCD_ORDER ? (unsigned)A.nwg8 * 8u : (unsigned)nwg, n);
    auto launch = [&](auto ctag, auto rtag) {
        constexpr int CC = decltype(ctag)::value, RR = decltype(rtag)::value;
        hipLaunchKernelGGL((k_box_carries<CC, box_strip_w(CC, RR), box_strip_left(CC, RR)>), dim3(ceil_div(from.rows, 4u), n), dim3(256), 0, s, dimg(&from), K, nwg, from_frame);
        if (sharpen) hipLaunchKernelGGL((k_box_fused<CC, RR, true>), grid, dim3(BF_THREADS), 0, s, A);
        else hipLaunchKernelGGL((k_box_fused<CC, RR, false>), grid, dim3(BF_THREADS), 0, s, A);
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
                printf("strip %d of %d (C=%d r=%u): busy / wait cycles per wave:", kk, nwg, C, radius);
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
