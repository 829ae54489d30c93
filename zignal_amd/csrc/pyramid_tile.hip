// pyramid_tile.hip — the levels of ImagePyramid.build on an Image(u8), each as ONE kernel with nothing but the level written to memory (round 6).
//
// Reference: src/image/pyramid.zig:55-93 — every level blurs the ORIGINAL (gaussianBlur with the level's sigma, .mirror borders: src/image.zig:954-994,
// src/image/convolution.zig:441-647 with integer taps round(k * 256) and divClampU8(65536)), rounds to u8, then resize(.bilinear)
// (src/image/interpolation.zig:313-407: fx = round(frac * 256), ((tl (256 - fx) + tr fx)(256 - fy) + (bl (256 - fx) + br fx) fy + 32768) >> 16, mirror-resolved taps).
//
// Round 5 ran it as row pass -> u16 temp plane in HBM -> column pass (dense, then a resize launch; or evaluated where the resize looks): 827 MB of counted traffic
// for 33.5 MB of algorithmic bytes, and the row pass dense although a level reduced by s reads 2 of every s blurred columns. The blur's double sum is an exact
// integer below 2^24 (2^25 for tap sums of 257) in ANY order, so it can be evaluated only where the resize looks, in the cheaper order, on the chip:
//
//   a workgroup owns a tile of 64 output columns x TH output rows of one level (TH chosen so that the source rows it needs fill the staging buffer)
//   1. STAGE   the source bytes the tile's windows reach — rows and columns resolved through the mirror rule on the way in, so that nothing after this step
//              knows about borders — into LDS (dword loads where the tile lies inside the image).
//   2. ROWS    output column C taps the two neighbouring source columns cb, cb + 1. For every staged row the row pass is evaluated at those two columns only:
//              a lane owns one C and walks row pairs; the taps, shifted to the byte phase of its window, are packed four to a register once, and each group of
//              four taps is one v_dot4_u32_u8 on an aligned dword of the staged row (no byte extraction, no alignment instruction). Results (<= 255 x 257 =
//              65 535) leave as 16-bit pairs of two rows: H[row pair][C] = {column cb: row 2q | row 2q + 1 << 16, column cb + 1: the same}.
//   3. COLUMNS output row R taps the two neighbouring blurred rows rb, rb + 1. A wave owns one R (so the phase of its window in the row pairs is uniform and
//              the tap pairs stay in scalar registers): per lane HM + 1 or HM + 2 LDS reads of 8 bytes and 4 (HM + 1) v_dot2_u32_u16 give the four sums,
//              divClampU8 gives the four blurred bytes, and Image(u8).resize's integer expression the output byte.
//
// Instructions per output pixel ~ (s / 2) x (8 + 4 ND + ND reads) for step 2 and ~ 4 (HM + 1) + HM + 30 for step 3 (HM = the padded half width of the taps,
// ND = ceil((2 HM + 5) / 4) dwords per window) against ~ (1 + 2 / s) x K x s^2 multiply-adds plus two trips through HBM before.
#include "zg_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace zg {

constexpr int PT_TW = 64;        // output columns per tile: one output row per wave instruction
#ifndef PT_SR_N
#define PT_SR_N 128
#endif
constexpr int PT_SR = PT_SR_N;   // staged source rows (halo included)
constexpr int PT_SP = 304;       // bytes per staged row: 64 x 3.9 + 2 + 2 x 17 + 3 of alignment, rounded up to a multiple of 16
#ifndef PT_THREADS_N
#define PT_THREADS_N 512
#endif
constexpr int PT_THREADS = PT_THREADS_N;
constexpr int PT_MAX_JOBS = 8;
constexpr int PT_MAX_HM = 17;    // taps <= 35

struct PyrTileJob {
    uint8_t *dst;
    uint32_t dst_pitch;
    int32_t drows, dcols;
    float rx, ry;                 // (float)cols / dcols, (float)rows / drows: Image.resize's ratios
    int32_t hm;                   // the padded half width of the taps: 5, 9, 13 or 17 (which body runs)
    int32_t th;                   // output rows per tile
    int32_t wide;                 // the taps sum to 257: a blurred value can pass 255 and divClampU8 clamps it (sums <= 256 never do)
    int32_t tiles_x, block0;      // tiles across; this job's first workgroup in the launch
    int32_t ntiles;               // this job's workgroups
    uint32_t tapb[12];            // the taps as bytes: tap j at byte 4 + j + (HM - half), zeros around (48 bytes)
    uint32_t pair_a[PT_MAX_HM + 1]; // (k[2q], k[2q + 1]): a window that starts on an even row of the pairs
    uint32_t pair_b[PT_MAX_HM + 1]; // (k[2q - 1], k[2q]): one that starts on an odd row
};
struct PyrTileJobs {
    DImg src;
    int32_t n;
    PyrTileJob j[PT_MAX_JOBS];
};

__device__ __forceinline__ uint32_t pt_udot2(uint32_t a, uint32_t b, uint32_t c) { // v_dot2_u32_u16: a.lo * b.lo + a.hi * b.hi + c
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}
__device__ __forceinline__ int pt_mirror(int i, int n) { // border.zig:46-63, .mirror
    if (i >= 0 && i < n) return i;
    return resolve_index(i, n, ZG_BORDER_MIRROR);
}

template <int HM>
__device__ __forceinline__ void pyr_tile_body(const PyrTileJobs &jobs, const PyrTileJob &job, uint32_t (*srcT)[PT_SP / 4], uint2 (*Hp)[PT_TW], uint32_t *tapw, uint2 *rowtab) {
    constexpr int ND = (2 * HM + 5 + 3) / 4; // dwords a lane's two windows (columns cb, cb + 1, any byte phase) span
    constexpr int NQ = HM + 1;               // row pairs a column window spans when it starts on an even row
    // workgroup b runs on XCD b % 8: the tiles of one XCD are a run of neighbours (they share source lines — a tile's rows are 90 .. 270 bytes of 128-byte
    // lines — in that XCD's L2 instead of each L2 fetching them for itself)
    int tile = (int)blockIdx.x - job.block0;
    {
        const int per_xcd = job.ntiles >> 3;
        if (ZG_XCD_ORDER && tile < (per_xcd << 3)) tile = (tile & 7) * per_xcd + (tile >> 3);
    }
    const int ty = tile / job.tiles_x, tx = tile - ty * job.tiles_x;
    const int rows = jobs.src.rows, cols = jobs.src.cols;
    const uint8_t *src = (const uint8_t *)jobs.src.data;
    const size_t spitch = (size_t)jobs.src.stride;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Image(u8).resize's taps of an output column / row (k_resize_bilinear_u8's expressions: geom.hip)
    auto taps_of = [](int o, float ratio, int n, int &base, bool &swapped, int &frac) {
        const float sp = ((float)o + 0.5f) * ratio - 0.5f;
        const float fl = floorf(sp);
        const int lo = (int)fl;
        const int a = pt_mirror(lo, n), b = pt_mirror(lo + 1, n);
        base = min(a, b); // neighbours (the plane is at least two wide / high), the first tap the higher one only at a mirrored edge
        swapped = a > b;
        frac = (int)roundf((sp - fl) * 256);
    };
    const int C0 = tx * PT_TW, R0 = ty * job.th;
    const int Clast = min(C0 + PT_TW, job.dcols) - 1, Rlast = min(R0 + job.th, job.drows) - 1;
    // The tile's geometry is the same for every wave: wave 0 works it out (two hundred scalar instructions of resize taps and mirror arithmetic) and leaves
    // it in LDS for the others — eight copies of it were a fifth of the kernel's instructions.
    if (wave == 0) {
        int xmin, xmax, ymin, ymax, f_;
        bool s_;
        taps_of(C0, job.rx, cols, xmin, s_, f_);
        taps_of(Clast, job.rx, cols, xmax, s_, f_);
        taps_of(R0, job.ry, rows, ymin, s_, f_);
        taps_of(Rlast, job.ry, rows, ymax, s_, f_);
        xmin = min(xmin, xmax); // (a mirrored last column can sit one below its predecessor)
        ymin = min(ymin, ymax);
        const int xa = (xmin - HM) & ~3, ya = (ymin - HM) & ~1;
        const float y_first = ((float)R0 + 0.5f) * job.ry - 0.5f, y_last = ((float)Rlast + 0.5f) * job.ry - 0.5f;
        const float x_first = ((float)C0 + 0.5f) * job.rx - 0.5f, x_last = ((float)Clast + 0.5f) * job.rx - 0.5f;
        if (lane == 0) {
            tapw[12] = (uint32_t)xa;
            tapw[13] = (uint32_t)ya;
            tapw[14] = (uint32_t)min(ymax + 1 + HM - ya + 1, PT_SR);              // staged rows (the host sized th so that they fit)
            tapw[15] = (uint32_t)min((xmax + 1 + HM - xa + 4) >> 2, PT_SP / 4);   // staged dwords per row
            // inside the image (no tap of the tile is mirrored) rows and columns come in order: no swaps, no mirror arithmetic per row
            tapw[16] = y_first >= 0.0f && (int)floorf(y_last) + 1 < rows && x_first >= 0.0f && (int)floorf(x_last) + 1 < cols;
        }
    }
    if (tid < 12) tapw[tid] = job.tapb[tid];
    __syncthreads();
    const int XA = (int)tapw[12];  // image column of staged byte 0 (may be negative; & ~3 rounds towards minus infinity)
    const int YA = (int)tapw[13];  // image row of staged row 0: row pairs are (YA + 2q, YA + 2q + 1)
    const int nsr = __builtin_amdgcn_readfirstlane((int)tapw[14]), nsd = __builtin_amdgcn_readfirstlane((int)tapw[15]);
    const bool plain = __builtin_amdgcn_readfirstlane((int)tapw[16]) != 0;

    // the resize's taps of the tile's output rows, one lane per row, once: as scalar work per output row and wave (floor, round, the mirror rule, ~60
    // instructions) they were a sixth of the kernel
    for (int t = tid; t <= Rlast - R0; t += PT_THREADS) {
        int rb, fy;
        bool swy;
        taps_of(R0 + t, job.ry, rows, rb, swy, fy);
        rowtab[t] = make_uint2((uint32_t)(rb - HM - YA) | (swy ? 0x80000000u : 0u), (uint32_t)fy);
    }
    // ---- 1. stage ------------------------------------------------------------------------------------------------------------------------------------
#ifndef PT_NO_STAGE // removal timings (tools/build_variant.sh): profiles/r06_pyramid.txt
    {
        // a wave takes whole rows (the row's mirror rule and pointer are scalar work), a lane dwords lane and lane + 64 of them
        const bool inside_x = XA >= 0 && XA + 4 * nsd <= cols; // workgroup-uniform: whole dwords of the image
        const bool second = lane + 64 < nsd;
        if (inside_x) {
            // every load of the wave's rows first, then the stores: a loop of load - wait - store left the memory's latency in the open once per row (sixteen
            // times per tile; the kernel spent most of its time there: profiles/r06_pyramid.txt). Rows past the tile's last are clamped, loaded and dropped.
            constexpr int NR = (PT_SR + PT_THREADS / 64 - 1) / (PT_THREADS / 64); // rows per wave at most
            const uint32_t o0 = (uint32_t)(XA + 4 * min(lane, nsd - 1)), o1 = (uint32_t)(XA + 4 * min(lane + 64, nsd - 1)); // clamped, unpredicated
            const bool inside_y = YA >= 0 && YA + nsr <= rows; // workgroup-uniform: no row of the tile is mirrored
            uint32_t v0[NR], v1[NR];
            if (inside_y) { // (two copies of the loop: the mirror rule's scalar arithmetic stays out of the interior tiles' code)
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const uint8_t *rowp = src + (size_t)(YA + min(wave + i * (PT_THREADS / 64), nsr - 1)) * spitch;
                    v0[i] = *(const uint32_t *)(rowp + o0);
                    v1[i] = *(const uint32_t *)(rowp + o1);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const uint8_t *rowp = src + (size_t)pt_mirror(YA + min(wave + i * (PT_THREADS / 64), nsr - 1), rows) * spitch;
                    v0[i] = *(const uint32_t *)(rowp + o0);
                    v1[i] = *(const uint32_t *)(rowp + o1);
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + i * (PT_THREADS / 64);
                if (r < nsr) { // wave-uniform
                    srcT[r][lane] = v0[i]; // (lanes past the staged dwords write the row's unused tail: a row has room for 76)
                    if (second) srcT[r][lane + 64] = v1[i];
                }
            }
        } else { // the image's left or right edge runs through the tile: byte by byte through the mirror rule
            uint32_t bo[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < 4; ++b) bo[h][b] = (uint32_t)pt_mirror(XA + 4 * min(lane + 64 * h, nsd - 1) + b, cols);
            for (int r = wave; r < nsr; r += PT_THREADS / 64) {
                const uint8_t *rowp = src + (size_t)pt_mirror(YA + r, rows) * spitch;
                uint32_t v[2] = {0, 0};
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int b = 0; b < 4; ++b) v[h] |= (uint32_t)rowp[bo[h][b]] << (8 * b);
                if (lane < nsd) srcT[r][lane] = v[0];
                if (second) srcT[r][lane + 64] = v[1];
            }
        }
    }
#endif
    __syncthreads();

    // ---- 2. the row pass at the tapped columns ------------------------------------------------------------------------------------------------------------
    int cb, fx;
    bool swx;
    taps_of(min(C0 + lane, job.dcols - 1), job.rx, cols, cb, swx, fx);
    {
        const int off = cb - HM - XA; // staged byte of the first tap of column cb (>= 0)
        const int p = off & 3, d0 = off >> 2;
        // the taps at the window's byte phase, four to a register: column cb's start at byte p of dword d0, column cb + 1's at byte p + 1
        uint32_t ka[ND], kb[ND];
#pragma unroll
        for (int m = 0; m < ND; ++m) {
            const uint32_t lo = tapw[m], hi = tapw[m + 1];
            // bytes 4 (m + 1) - p ... of the table = taps 4 m - p ...; and one byte earlier for the next column
            ka[m] = p == 0 ? hi : __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(4 - p));
            kb[m] = p == 3 ? lo : __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(3 - p));
        }
#ifdef PT_NO_ROWS
        const int npairs = 0;
#else
        const int npairs = (nsr + 1) >> 1;
#endif
        for (int q = wave; q < npairs; q += PT_THREADS / 64) {
            const uint32_t *r0 = &srcT[2 * q][d0], *r1 = &srcT[2 * q + 1][d0];
            uint32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
#pragma unroll
            for (int m = 0; m < ND; ++m) {
                const uint32_t v0 = r0[m], v1 = r1[m];
                a0 = __builtin_amdgcn_udot4(v0, ka[m], a0, false);
                b0 = __builtin_amdgcn_udot4(v0, kb[m], b0, false);
                a1 = __builtin_amdgcn_udot4(v1, ka[m], a1, false);
                b1 = __builtin_amdgcn_udot4(v1, kb[m], b1, false);
            }
            Hp[q][lane] = make_uint2(a0 | (a1 << 16), b0 | (b1 << 16)); // each <= 255 x 257 = 65 535
        }
    }
    __syncthreads();

    // ---- 3. the column pass at the tapped rows, divClampU8, the bilinear taps ---------------------------------------------------------------------------------
    const bool live_c = C0 + lane < job.dcols;
    const bool wide = job.wide != 0; // workgroup-uniform
    // the tap pairs, resident in vector registers for the whole loop (as scalar operands they were re-loaded from the kernel arguments for every output row)
    uint32_t pa[NQ], pb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { pa[q] = job.pair_a[q]; pb[q] = job.pair_b[q]; asm volatile("" : "+v"(pa[q]), "+v"(pb[q])); }
    const uint32_t wx1 = (uint32_t)fx, wx0 = 256u - wx1;
    uint8_t *out_col = job.dst + (size_t)(C0 + lane);
    const uint32_t hp_lane = (uint32_t)lane * 8u;
    const char *hp0 = (const char *)&Hp[0][0];
#ifdef PT_NO_COLS
    if (false)
#endif
    for (int R = R0 + wave; R <= Rlast; R += PT_THREADS / 64) { // wave-uniform
        const uint2 rt = rowtab[R - R0];                                  // one address for the wave: a broadcast read
        const int a = __builtin_amdgcn_readfirstlane((int)(rt.x & 0x7fffffffu)); // staged row of the first tap of blurred row rb (>= 0)
        const bool swy = !plain && __builtin_amdgcn_readfirstlane((int)rt.x) < 0;
        const int fy = (int)rt.y;
        const uint2 *hq = (const uint2 *)(hp0 + (uint32_t)(a >> 1) * (uint32_t)(PT_TW * 8) + hp_lane);
        uint32_t va = 32768u, vb = 32768u, wa = 32768u, wb = 32768u; // column cb: rows rb, rb + 1; column cb + 1: the same; divClampU8's rounding term rides along
        if ((a & 1) == 0) { // row rb's window starts a pair, row rb + 1's one row later: the same NQ pairs
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint2 h = hq[q * PT_TW];
                va = pt_udot2(h.x, pa[q], va);
                vb = pt_udot2(h.x, pb[q], vb);
                wa = pt_udot2(h.y, pa[q], wa);
                wb = pt_udot2(h.y, pb[q], wb);
            }
        } else { // row rb's window starts on the second row of pair q0, row rb + 1's on pair q0 + 1
#pragma unroll
            for (int q = 0; q <= NQ; ++q) {
                const uint2 h = hq[q * PT_TW];
                if (q < NQ) {
                    va = pt_udot2(h.x, pb[q], va);
                    wa = pt_udot2(h.y, pb[q], wa);
                }
                if (q > 0) {
                    vb = pt_udot2(h.x, pa[q - 1], vb);
                    wb = pt_udot2(h.y, pa[q - 1], wb);
                }
            }
        }
        // divClampU8(65536) (convolution.zig:18-22): sums are non-negative, so the quotient is the upper half of sum + 32768; only tap sums of 257 can pass 255
        uint32_t b_rb_cb = va >> 16, b_rb1_cb = vb >> 16, b_rb_cb1 = wa >> 16, b_rb1_cb1 = wb >> 16;
        if (wide) { b_rb_cb = min(b_rb_cb, 255u); b_rb1_cb = min(b_rb1_cb, 255u); b_rb_cb1 = min(b_rb_cb1, 255u); b_rb1_cb1 = min(b_rb1_cb1, 255u); }
        uint32_t top_l = b_rb_cb, top_r = b_rb_cb1, bot_l = b_rb1_cb, bot_r = b_rb1_cb1;
        if (!plain) { // the resize's four taps: top / bottom = rows r0, r1 (swapped at a mirrored edge), left / right = columns cl, cr
            top_l = swy ? (swx ? b_rb1_cb1 : b_rb1_cb) : (swx ? b_rb_cb1 : b_rb_cb);
            top_r = swy ? (swx ? b_rb1_cb : b_rb1_cb1) : (swx ? b_rb_cb : b_rb_cb1);
            bot_l = swy ? (swx ? b_rb_cb1 : b_rb_cb) : (swx ? b_rb1_cb1 : b_rb1_cb);
            bot_r = swy ? (swx ? b_rb_cb : b_rb_cb1) : (swx ? b_rb1_cb : b_rb1_cb1);
        }
        const uint32_t top_val = top_l * wx0 + top_r * wx1;
        const uint32_t bottom_val = bot_l * wx0 + bot_r * wx1;
        const uint32_t v = (top_val * (uint32_t)(256 - fy) + bottom_val * (uint32_t)fy + 32768u) >> 16; // < 256: a convex combination of bytes
        if (live_c) out_col[(size_t)R * job.dst_pitch] = (uint8_t)v;
    }
}

// Every level of the pyramid in ONE launch (the levels are independent and each one alone leaves a tail of idle CUs): the workgroup finds its job and
// runs the body built for that job's tap-length class.
__global__ __launch_bounds__(PT_THREADS) void k_pyr_tile(PyrTileJobs jobs) {
    __shared__ uint32_t srcT[PT_SR + 1][PT_SP / 4]; // + 1, + 2: slack rows that only ever meet zero taps (no clamp in the loops)
    __shared__ uint2 Hp[PT_SR / 2 + 2][PT_TW];
    __shared__ uint32_t tapw[20]; // the taps as bytes (12 dwords), then the tile's geometry
    __shared__ uint2 rowtab[PT_SR];  // per output row of the tile: staged row of its window's first tap | swapped << 31, fy
    int ji = 0; // jobs are few: a scalar walk
    for (int i = 1; i < jobs.n; ++i)
        if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
    const PyrTileJob &job = jobs.j[ji];
    switch (job.hm) { // workgroup-uniform
    case 5: pyr_tile_body<5>(jobs, job, srcT, Hp, tapw, rowtab); break;
    case 9: pyr_tile_body<9>(jobs, job, srcT, Hp, tapw, rowtab); break;
    case 13: pyr_tile_body<13>(jobs, job, srcT, Hp, tapw, rowtab); break;
    default: pyr_tile_body<17>(jobs, job, srcT, Hp, tapw, rowtab); break;
    }
}

static bool pyr_tile_off() {
    static const bool off = getenv("ZIGNAL_HIP_NO_PYRAMID_TILE") != nullptr; // the A/B of round 6: round 5's temp-plane route
    return off;
}

// Levels i with handled[i] set on return were enqueued here (on `s`, one launch per tap-length class); the others are the caller's. -1: nothing was done.
int try_pyramid_tiles_u8(const zg_image *src, const zg_image *levels, const float *sigmas, uint32_t n, uint8_t *handled, hipStream_t s) {
    if (pyr_tile_off()) return -1;
    if (src->pixel != ZG_PIXEL_U8 || src->rows < 2 || src->cols < 2 || (src->stride & 3) || ((uintptr_t)src->data & 3) || src->cols > 0x3fffffffu || src->rows > 0x3fffffffu) return -1;
    static const int classes[4] = {5, 9, 13, 17};
    PyrTileJobs J;
    J.src = dimg(src);
    J.n = 0;
    int grid = 0;
    // the longest taps first: their tiles take longest, and a launch ends with its last workgroup
    uint32_t order[64];
    uint32_t no = 0;
    for (uint32_t i = 0; i < n && no < 64; ++i) order[no++] = i;
    std::stable_sort(order, order + no, [&](uint32_t a, uint32_t b) { return sigmas[a] > sigmas[b]; });
    for (uint32_t oi = 0; oi < no; ++oi) {
        const uint32_t i = order[oi];
        const zg_image &lv = levels[i];
        if (handled[i] || !(sigmas[i] > 0.5f) || lv.pixel != ZG_PIXEL_U8 || lv.rows == 0 || lv.cols == 0 || lv.rows > src->rows || lv.cols > src->cols) continue;
        const float rx = (float)src->cols / (float)lv.cols, ry = (float)src->rows / (float)lv.rows;
        if (!(rx < 3.9f) || !(ry < 3.9f) || (size_t)lv.stride > 0xffffffffu) continue;
        // From a reduction by 2 the resize looks at fewer blurred pixels than there are (4 / s^2 of them) and evaluating the blur only there wins on arithmetic;
        // below it the tile kernel computes pixels twice, but it also never writes a temp or a blurred plane. With the first staging loop the best threshold was
        // 2.0 (thresholds 0 / 1.4 / 1.7 / 2.0 / 2.4 / 2.9 -> 239 / 231 / 223 / 212 / 215 / 231 us for ORB's default pyramid of a 4096^2 plane); with every load
        // of a tile in flight at once it is a tie in the graph replay (0 / 1.3 / 1.6 / 2.0 -> 182 - 186 / 177 - 183 / 182 / 177 us) and the tile kernel's side
        // wins the eager call (183 - 186 / 187 - 190 / 211 / 208 us) and the traffic (156 / 190 / - / 405 MB): 1.3, i.e. every level but a 1.2 x one
        // (profiles/r06_pyramid.txt). ZIGNAL_HIP_PYRAMID_TILE_MIN_RATIO moves the threshold (read once).
        static const float min_ratio = getenv("ZIGNAL_HIP_PYRAMID_TILE_MIN_RATIO") ? (float)atof(getenv("ZIGNAL_HIP_PYRAMID_TILE_MIN_RATIO")) : 1.3f;
        if (rx < min_ratio) continue;
        const int nfull = zg_gaussian_kernel(sigmas[i], nullptr, 0);
        if (nfull < 1 || nfull > 129) continue;
        float ft[129];
        if (zg_gaussian_kernel(sigmas[i], ft, 129) != nfull) continue;
        int32_t it[129];
        int64_t sum = 0;
        bool ok = true;
        for (int j = 0; j < nfull; ++j) { it[j] = (int32_t)std::round(ft[j] * 256.0f); ok = ok && it[j] >= 0 && it[j] <= 255; sum += it[j]; } // scaleKernelToInt (convolution.zig:303-309)
        if (!ok || sum > 257) continue; // a row sum must fit 16 bits: 255 x 257 = 65 535
        int z = 0; // outer taps that rounded to zero add nothing
        while (nfull - 2 * z > 2 && it[z] == 0 && it[nfull - 1 - z] == 0) ++z;
        const int nk = nfull - 2 * z, half = nk / 2;
        if (!(nk & 1) || half > PT_MAX_HM || J.n == PT_MAX_JOBS) continue;
        int c = 0;
        while (classes[c] < half) ++c;
        const int HM = classes[c];
        PyrTileJob &job = J.j[J.n];
        memset(&job, 0, sizeof(job));
        job.dst = (uint8_t *)lv.data;
        job.dst_pitch = (uint32_t)lv.stride;
        job.drows = (int32_t)lv.rows;
        job.dcols = (int32_t)lv.cols;
        job.rx = rx;
        job.ry = ry;
        job.hm = HM;
        job.wide = sum > 256;
        // the staged rows of a tile: the tapped rows of th output rows (<= th ry + 2) + HM above and below + one to make the first row even
        job.th = std::max(1, (int)std::floor((float)(PT_SR - 2 * HM - 4) / ry));
        job.tiles_x = (int)ceil_div(lv.cols, (unsigned)PT_TW);
        const uint64_t tiles = (uint64_t)job.tiles_x * ceil_div(lv.rows, (unsigned)job.th);
        if (tiles + (uint64_t)grid > 0x3fffffffu) continue;
        job.block0 = grid;
        job.ntiles = (int32_t)tiles;
        uint8_t tb[48];
        memset(tb, 0, sizeof(tb));
        int32_t k[2 * PT_MAX_HM + 3]; // k[j + 1] = padded tap j, zeros around
        memset(k, 0, sizeof(k));
        for (int j = 0; j < nk; ++j) { tb[4 + (HM - half) + j] = (uint8_t)it[z + j]; k[1 + (HM - half) + j] = it[z + j]; }
        memcpy(job.tapb, tb, sizeof(tb));
        for (int q = 0; q <= HM; ++q) {
            const int j = 2 * q; // padded tap index
            job.pair_a[q] = (uint32_t)k[1 + j] | ((uint32_t)(j + 1 <= 2 * HM ? k[2 + j] : 0) << 16);
            job.pair_b[q] = (uint32_t)k[j] | ((uint32_t)k[1 + j] << 16); // k[0] = 0: the tap before the first
        }
        grid += (int)tiles;
        ++J.n;
        handled[i] = 1;
    }
    if (!J.n) return -1;
    hipLaunchKernelGGL(k_pyr_tile, dim3((unsigned)grid), dim3(PT_THREADS), 0, s, J);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

} // namespace zg
