// box_blur.hip — Image(T).boxBlur through an f32 summed-area table.
//
// Replaces reference src/image.zig:635-648 and src/image/integral.zig:41-78 (plane), :86-91 (sum), :194-269
// (boxBlurPlane). The SAT is built in f32 with the reference's exact association order — a running sum along
// each row, then rows accumulated top to bottom — so that its rounding (inexact above 2^24) is reproduced bit
// for bit; the window and its area are clipped at the borders and the mean goes through meta.clamp for u8.
// All SAT planes are complete before any output is written, so src may alias dst as in the reference.
//
// f32 sources (the general case):
//   k_sat_rows   one wave per (64 rows, channel): 64x64 tiles staged through LDS (coalesced, loads batched), each lane
//                scans its row of the tile sequentially and carries the running sum to the next tile.
//   k_sat_cols   one thread per (column, channel): sequential accumulation down the rows, coalesced across lanes,
//                128 rows prefetched ahead of the chain.
// The two scans are sequential f32 chains by contract (the reference's rounding order), so their parallelism is capped
// at rows x channels and columns x channels chains; they are latency-bound, not bandwidth-bound.
// Integer-valued sources (every u8 pixel type; the detectors' planes) have exact row sums in any order:
//   k_strip_carries  the sum left of every 16-column strip of every row (a block scan per row), 1/16 of a plane
//   k_sat_chain      reads the source once, rebuilds the row prefixes from the carries, runs the column chain and writes the SAT
//                    once: a chain wave alone on its SIMD, eight loader waves, four storer waves per 64 columns of one channel
// then, for both:
//   k_box_mean   one thread per pixel: four SAT taps (buffer loads), divide by the clipped area, clamp.
#include "zg_common.h"

#include <algorithm>
#include <cstdlib>

#pragma clang fp contract(off)

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);
int try_box_fused(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, uint32_t radius, bool sharpen, hipStream_t s); // box_fused.hip

// Row pass. The running sum along a row is a sequential f32 chain (f32 addition is not associative and the reference's
// rounding must be reproduced), so the parallelism is rows x channels. One wave owns 64 rows of one channel: per chunk
// of 64 columns it loads the 64 x 64 tile (lanes = consecutive columns: coalesced; the next chunk's loads are issued
// before this chunk is processed), transposes through LDS, each lane runs its row's 64-step chain in registers carrying
// the sum across chunks, and the results go back out transposed (coalesced again).
template <int PIX>
__global__ __launch_bounds__(64) void k_sat_rows(DImg src, float *sat) { // sat: [C][rows][cols]
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C;
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64;
    const int ch = blockIdx.y;
    const size_t plane = (size_t)src.rows * src.cols;
    const Elem *base = (const Elem *)src.data;
    const int nrows = min(64, src.rows - r0);

    auto fetch = [&](int c0, float (&v)[64]) {
        // addresses are clamped into the image instead of predicating the loads: a predicated load makes the compiler
        // wait for each one before issuing the next (measured 14 us per chunk); rows / columns past the edge are never stored
        const int c = min(c0 + lane, src.cols - 1);
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = (float)base[((size_t)(r0 + min(i, nrows - 1)) * src.stride + c) * C + ch];
    };

    float cur[64], nxt[64];
    fetch(0, cur);
    float run = 0.0f; // lane = row within the strip
    for (int c0 = 0; c0 < src.cols; c0 += 64) {
        fetch(c0 + 64, nxt); // clamped: harmless past the last chunk
#pragma unroll
        for (int i = 0; i < 64; ++i) tile[i][lane] = cur[i];
        __syncthreads();
        float rowv[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) rowv[j] = tile[lane][j];
#pragma unroll
        for (int j = 0; j < 64; ++j) { // the chain values produced by columns past the edge are never stored
            run = run + rowv[j];
            rowv[j] = run;
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) tile[lane][j] = rowv[j];
        __syncthreads();
        const int c = c0 + lane;
        if (c < src.cols) {
            float *o = sat + (size_t)ch * plane + (size_t)r0 * src.cols + c;
#pragma unroll
            for (int i = 0; i < 64; ++i)
                if (i < nrows) o[(size_t)i * src.cols] = tile[i][lane];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 64; ++i) cur[i] = nxt[i];
    }
}

// Column pass: sat[r][c] += sat[r-1][c], a sequential chain down each column (coalesced across lanes). Only columns x
// channels / 64 waves exist, so the achieved bandwidth is (bytes in flight) / latency: rows are taken 128 at a time with
// the following 128 already in flight (32 rows: 177 us, 128 rows: 157 us per 4096^2 plane; scalar row addressing: 195 us).
__global__ __launch_bounds__(64) void k_sat_cols(float *sat, int rows, int cols) {
    constexpr int G = 128;
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int ch = blockIdx.y;
    if (c >= cols) return;
    float *p = sat + (size_t)ch * rows * cols + c;
    auto fetch = [&](int r, float (&v)[G]) {
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = p[(size_t)min(r + i, rows - 1) * cols]; // clamped, not predicated (see k_sat_rows)
    };
    float prev = 0.0f;
    auto chain_store = [&](int r, float (&v)[G]) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (r + i < rows) {
                if (r + i > 0) v[i] = v[i] + prev; // dst[curr] += dst[prev], rows 1..
                prev = v[i];
                p[(size_t)(r + i) * cols] = v[i];
            }
        }
    };
    float a[G], b[G];
    fetch(0, a);
    for (int r = 0; r < rows; r += 2 * G) {
        fetch(r + G, b);
        chain_store(r, a);
        fetch(r + 2 * G, a);
        chain_store(r + G, b);
    }
}

// SHARPEN: Integral.sharpen (integral.zig:273-323, 325-426): 2 * original - blurred instead of the mean itself.
// BUF: the planes are below 4 GiB together, so the four corners of a channel are buffer loads (the row and the plane in the
// scalar offset, the column per lane, corners that do not exist pointed out of range where a buffer load reads zero): no address
// arithmetic and no selects. For integer pixels the C quotients share their divisor: the refined reciprocal of the IEEE division
// sequence (v_rcp + one Newton step) is computed once, each quotient is the sequence's remaining five operations, bit for bit what
// `/` expands to when v_div_scale has nothing to scale (a finite sum of 8-bit samples over an area >= 1).
// (One thread per pixel with 64-bit addresses and four full divisions was VALU-bound: 81 us per 4096^2 Rgba(u8) frame.)
// A launch covers a batch of equally shaped frames (the pipeline's box-blur step, batch.hip): frame f = blockIdx.z / zpf, its SAT planes
// sat_frame elements and its pixels fr.src_frame / fr.dst_frame bytes after the previous frame's; zpf = gridDim.z slices per frame (1 below 65 536
// workgroup rows).
template <int PIX, bool SHARPEN, bool BUF>
__global__ __launch_bounds__(256) void k_box_mean(const float *sat, size_t plane, DImg src, DImg dst, int radius, size_t sat_frame, FrameSpan fr, int zpf) { // plane: elements from one channel's SAT to the next
    using P = Px<PIX>;
    const int frame = (int)blockIdx.z / zpf, zrow = (int)blockIdx.z - frame * zpf;
    sat += (size_t)frame * sat_frame;
    src.data = (char *)src.data + (size_t)frame * fr.src_frame;
    dst.data = (char *)dst.data + (size_t)frame * fr.dst_frame;
    const int grow = zrow * (int)GRID_Y_MAX + (int)blockIdx.y; // grid_row() of this frame
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    constexpr bool IS_F = std::is_same<typename P::Elem, float>::value;
    // One plane: a workgroup is 64 columns x 16 rows (a wave per row, four steps). The two SAT rows an output row needs are needed
    // again 2 * radius + 1 rows further down, and the tile keeps them in the CU's L1 in between (34 -> 25 us per 4096^2 plane).
    // Several planes: 256 columns of one row; their tiles no longer fit (64 x 16: 108 us, 64 x 4: 96 us, 256 x 1: 77 us for Rgba(u8)).
    constexpr int STEPS = C == 1 ? 4 : 1;
    const int c = C == 1 ? blockIdx.x * 64 + (int)(threadIdx.x & 63) : blockIdx.x * 256 + (int)threadIdx.x;
    const int wrow = C == 1 ? grow * 16 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : grow;
    if (c >= dst.cols) return;
    const int rows = dst.rows, cols = dst.cols;
    const int c1 = max(c - radius, 0), c2 = (int)min((long long)c + radius, (long long)cols - 1);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
    const int r = wrow + step * 4;
    if (r >= rows) return; // wave-uniform
    const int r1 = max(r - radius, 0), r2 = (int)min((long long)r + radius, (long long)rows - 1);
    const float area = (float)((long long)(r2 - r1 + 1) * (long long)(c2 - c1 + 1));
    Vec o, orig = P::zero();
    if constexpr (SHARPEN) orig = P::load(src.data, (size_t)r * src.stride + (size_t)c);
    float sum[C];
    if constexpr (BUF) {
        constexpr uint32_t OOR = 0xfffffff0u;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sat, (short)0, (int)(uint32_t)(plane * C * 4), 0x00020000);
        const uint32_t oa = (uint32_t)c2 * 4u, ob = c1 > 0 ? (uint32_t)(c1 - 1) * 4u : OOR;
        const uint32_t od = r1 > 0 ? oa : OOR, oe = r1 > 0 ? ob : OOR;
        const uint32_t bot = (uint32_t)((size_t)r2 * cols * 4), top = (uint32_t)((size_t)(r1 > 0 ? r1 - 1 : 0) * cols * 4);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const uint32_t pl = (uint32_t)(plane * ch * 4);
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)oa, (int)(pl + bot), 0));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)ob, (int)(pl + bot), 0));
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)od, (int)(pl + top), 0));
            const float e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)oe, (int)(pl + top), 0));
            sum[ch] = a - b - d + e; // ((a - b) - d) + e, integral.zig:87-90
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const float *s = sat + (size_t)ch * plane;
            const float a = s[(size_t)r2 * cols + c2];
            const float b = c1 > 0 ? s[(size_t)r2 * cols + (c1 - 1)] : 0.0f;
            const float d = r1 > 0 ? s[(size_t)(r1 - 1) * cols + c2] : 0.0f;
            const float e = (r1 > 0 && c1 > 0) ? s[(size_t)(r1 - 1) * cols + (c1 - 1)] : 0.0f;
            sum[ch] = a - b - d + e;
        }
    }
    const float nd = -area, r0 = __builtin_amdgcn_rcpf(area);
    const float r1f = __builtin_fmaf(__builtin_fmaf(nd, r0, 1.0f), r0, r0);
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        float val;
        if constexpr (IS_F) {
            val = sum[ch] / area;
        } else {
            const float q0 = sum[ch] * r1f;
            const float q1 = __builtin_fmaf(__builtin_fmaf(nd, q0, sum[ch]), r1f, q0);
            val = __builtin_fmaf(__builtin_fmaf(nd, q1, sum[ch]), r1f, q1);
        }
        if constexpr (SHARPEN) {
            const float original = (float)orig[ch];
            const float twice = 2 * original;
            val = twice - val;
        }
        if constexpr (IS_F) {
            o[ch] = val;
        } else { // finite: no NaN case to map
            const float u = fminf(fmaxf(val, 0.0f), 255.0f), t = truncf(u);
            o[ch] = (uint8_t)((int)t + ((u - t) >= 0.5f ? 1 : 0));
        }
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, o);
    }
}

// Row pass for sources whose row sums are exact in f32 — integer-valued elements with cols * max < 2^24 (every u8 pixel type,
// and the detectors' grey / mask planes): each partial sum is then an integer below 2^24, every association gives the
// same bits as the reference's left-to-right loop, and the row becomes a parallel prefix sum (one workgroup per row and
// channel, 1024 columns per step) at memory speed instead of a 4096-step chain (316 us -> see DESIGN.md).
template <int PIX>
__global__ __launch_bounds__(256) void k_sat_rows_exact(DImg src, float *sat) { // sat: [C][rows][cols]
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C;
    __shared__ float wsum[4];
    const int r = blockIdx.x, ch = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const Elem *row = (const Elem *)src.data + (size_t)r * src.stride * C + ch;
    float *out = sat + ((size_t)ch * src.rows + r) * src.cols;
    float carry = 0.0f;
    for (int c0 = 0; c0 < src.cols; c0 += 1024) {
        const int c = c0 + 4 * t;
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = (float)row[(size_t)min(c + k, src.cols - 1) * C]; // clamped, unpredicated
#pragma unroll
        for (int k = 0; k < 4; ++k) if (c + k >= src.cols) p[k] = 0.0f;
        p[1] += p[0]; p[2] += p[1]; p[3] += p[2];
        float x = p[3]; // inclusive scan of the per-thread totals across the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        float base = carry + (x - p[3]);
        if (w > 0) base += wsum[0];
        if (w > 1) base += wsum[1];
        if (w > 2) base += wsum[2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c + k < src.cols) out[c + k] = base + p[k];
        carry += ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
        __syncthreads();
    }
}

// ---- exact-row sources: strip carries, then prefix + chain + store in one pass ------------------------------------------------
// The three-kernel form writes the row prefixes (4 B per element), reads them back for the column chain and writes the SAT:
// 12 B of traffic per element for 4 B of result, and the chain kernel has only columns x channels / 64 waves to pull it with.
// For sources whose row sums are exact (see k_sat_rows_exact) the row prefix of any element is "the sum of everything left of
// its 16-column strip" (an exact integer, the strip's CARRY) plus a prefix inside the strip, so the column chain can rebuild
// it on the fly: k_strip_carries writes one value per row, strip and channel (1/16 of a plane), k_sat_chain reads the source
// once and writes the SAT once.

// carry[(r * nstrips + s) * C + ch] = sum of row r, channel ch, over columns < 16 s, as an exact f32. One workgroup per row,
// every channel at once (the source is read once): 4 pixels per thread and step, integer block scan.
template <int PIX, int LOG2G = 4> // strips of 2^LOG2G columns (16 for the SAT kernels, 8 for the fused box blur)
__device__ __forceinline__ void strip_carries_body(const DImg &src, float *carries, int nstrips, uint32_t (*wsum)[4][Px<PIX>::C]) {
    static_assert(LOG2G >= 2, "a thread covers four columns");
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C;
    const int r = blockIdx.x, t = threadIdx.x, w = t >> 6;
    // four adjacent pixels in one load when they are contiguous and aligned (always, for images this library allocated)
    bool vec = false;
    if constexpr (PIX == ZG_PIXEL_U8) vec = (src.stride & 3) == 0 && ((uintptr_t)src.data & 3) == 0;
    if constexpr (PIX == ZG_PIXEL_RGBA_U8 || PIX == ZG_PIXEL_F32) vec = ((size_t)src.stride * sizeof(Elem) * C) % 16 == 0 && ((uintptr_t)src.data & 15) == 0;
    const Elem *row = (const Elem *)src.data + (size_t)r * src.stride * C;
    uint32_t carry[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) carry[ch] = 0;
    for (int c0 = 0, it = 0; c0 < src.cols; c0 += 1024, ++it) {
        const int c = c0 + 4 * t;
        uint32_t x[C]; // the sum of my four pixels, then its inclusive scan over the wave
        if (vec && c0 + 1024 <= src.cols) { // workgroup-uniform
            if constexpr (PIX == ZG_PIXEL_U8) {
                const uint32_t v = *(const uint32_t *)(row + c);
                x[0] = (v & 0xffu) + ((v >> 8) & 0xffu) + ((v >> 16) & 0xffu) + (v >> 24);
            } else if constexpr (PIX == ZG_PIXEL_RGBA_U8) {
                const uint4 v = *(const uint4 *)(row + (size_t)c * 4);
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                    x[ch] = ((v.x >> (8 * ch)) & 0xffu) + ((v.y >> (8 * ch)) & 0xffu) + ((v.z >> (8 * ch)) & 0xffu) + ((v.w >> (8 * ch)) & 0xffu);
            } else if constexpr (PIX == ZG_PIXEL_F32) {
                const float4 v = *(const float4 *)(row + c);
                x[0] = (uint32_t)v.x + (uint32_t)v.y + (uint32_t)v.z + (uint32_t)v.w; // integer-valued by contract
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) x[ch] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const typename P::Vec v = P::load(src.data, (size_t)r * src.stride + (size_t)min(c + k, src.cols - 1)); // clamped, unpredicated
#pragma unroll
                for (int ch = 0; ch < C; ++ch) x[ch] += c + k < src.cols ? (uint32_t)v[ch] : 0u; // integer-valued elements (u8, or f32 holding 0..255)
            }
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) { // inclusive scan over 64 lanes: four steps inside the 16-lane rows, then the row totals broadcast
            uint32_t y = x[ch];
            y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x111, 0xf, 0xf, true);  // row_shr:1
            y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x112, 0xf, 0xf, true);  // row_shr:2
            y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x114, 0xf, 0xf, true);  // row_shr:4
            y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x118, 0xf, 0xf, true);  // row_shr:8
            y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
            y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
            x[ch] = y;
            if ((t & 63) == 63) wsum[it & 1][w][ch] = y;
        }
        __syncthreads(); // one barrier per step: the totals alternate between two buffers
        const int next = c + 4; // the prefix through column c + 3 is the carry of the strip that starts at column c + 4
        const bool boundary = (next & ((1 << LOG2G) - 1)) == 0 && next < src.cols;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const uint32_t w0 = wsum[it & 1][0][ch], w1 = wsum[it & 1][1][ch], w2 = wsum[it & 1][2][ch], w3 = wsum[it & 1][3][ch];
            uint32_t base = carry[ch] + x[ch];
            if (w > 0) base += w0;
            if (w > 1) base += w1;
            if (w > 2) base += w2;
            if (boundary) carries[((size_t)r * nstrips + (next >> LOG2G)) * C + ch] = (float)base; // < 2^24: exact
            carry[ch] += w0 + w1 + w2 + w3;
        }
        if (c0 == 0 && t < C) carries[(size_t)r * nstrips * C + t] = 0.0f; // strip 0
    }
}
template <int PIX, int LOG2G = 4>
__global__ __launch_bounds__(256) void k_strip_carries(DImg src, float *carries, int nstrips, size_t src_frame) { // blockIdx.y: the frame of a batch
    __shared__ uint32_t wsum[2][4][Px<PIX>::C];
    src.data = (char *)src.data + (size_t)blockIdx.y * src_frame;
    strip_carries_body<PIX, LOG2G>(src, carries + (size_t)blockIdx.y * src.rows * nstrips * Px<PIX>::C, nstrips, wsum);
}

// A workgroup owns four adjacent 16-column strips of ONE channel for the whole height (64 columns: 256 contiguous bytes of SAT
// per row; strips of all channels in one workgroup were tried first and wrote 64-byte pieces into four planes: 0.8 TB/s).
// The row prefix of an element is its strip's carry plus a prefix inside the strip, exact; the SAT is the reference's column
// recurrence sat[r][c] = sat[r-1][c] + rowprefix[r][c] (integral.zig:60-77): one f32 addition per row, top to bottom.
// Only that addition is sequential, but every workgroup walks the whole height, and a lone wave issues one instruction per
// ~5 ns whatever it is (measured by switching roles and instructions off one at a time, profiles/r02_experiments.txt): the wave
// that chains must do nothing else. So the work is dealt out by role and by SIMD (wave w runs on SIMD w % 4):
//   wave 0         the CHAIN, alone on its SIMD (waves 4, 8, 12 leave at once): per row half an LDS read, one addition, half an
//                  LDS write: 41 us for 4096 rows, the floor of this kernel for a single plane
//   8 LOADERS      lane = (row of four, four adjacent columns): one load brings four elements of a row, their prefix costs three
//                  additions, the four lanes of a strip are scanned with two quad permutes, and a row-major float4 goes into the
//                  ring: about seven instructions per 64-column row where a lane per column needed sixteen; six blocks ahead
//   4 STORERS      float4 out of the chain's ring, four rows (4 x 256 contiguous bytes) per store instruction
// With four channels there are 256 workgroups and the kernel sits on HBM's write rate instead (268 MB of SAT: 86 us).
constexpr int SAT_SB = 64;         // rows per block of k_sat_chain
constexpr int SAT_THREADS = 1024;  // sixteen waves: the chain, three that leave, eight loaders, four storers

template <int PIX, bool VEC>
__device__ __forceinline__ void sat_loader(const DImg &src, const float *carries, int nstrips, int ch, float (*ring)[SAT_SB][64], int L, int lane,
                                           int nblocks) {
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C, SB = SAT_SB;
    constexpr int D = (VEC && PIX == ZG_PIXEL_U8) ? 6 : 5; // blocks in flight: 2 x (1 or 4) + 2 loads per block; five keep the wider forms out of scratch
    constexpr bool IS_F32 = sizeof(Elem) == 4;
    const int row4 = lane >> 4, q = lane & 15;
    const int rows = src.rows, cols = src.cols;
    const int x0 = blockIdx.x * 64, col0 = x0 + 4 * q;
    const bool all_live = x0 + 64 <= cols; // workgroup-uniform
    const Elem *elems = (const Elem *)src.data;
    const size_t row_elems = (size_t)src.stride * C, carry_step = (size_t)nstrips * C;
    size_t coff[4];
    uint32_t cmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        coff[j] = (size_t)min(col0 + j, cols - 1) * C + ch;
        cmask[j] = col0 + j < cols ? 0xffffffffu : 0u;
    }
    const size_t cidx = (size_t)min((int)blockIdx.x * 4 + (q >> 2), nstrips - 1) * C + ch;
    const uint32_t m1 = (lane & 3) >= 1 ? 0xffffffffu : 0u, m2 = (lane & 3) >= 2 ? 0xffffffffu : 0u;

    struct Regs { uint32_t w[2][VEC && PIX == ZG_PIXEL_U8 ? 1 : 4]; float k[2]; };
    auto load_row = [&](const Elem *rowp, const float *kp, Regs &g, int u) {
        if constexpr (VEC && PIX == ZG_PIXEL_U8) {
            g.w[u][0] = *(const uint32_t *)(rowp + col0);
        } else if constexpr (VEC) { // Rgba(u8): four pixels; f32: four elements
            const uint4 v = *(const uint4 *)(rowp + (size_t)col0 * C);
            g.w[u][0] = v.x; g.w[u][1] = v.y; g.w[u][2] = v.z; g.w[u][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (IS_F32) g.w[u][j] = __builtin_bit_cast(uint32_t, rowp[coff[j]]);
                else g.w[u][j] = (uint32_t)rowp[coff[j]];
            }
        }
        g.k[u] = kp[cidx];
    };
    auto fetch = [&](int blk, Regs &g) { // clamped, unpredicated; blocks past the end re-read the last row and are never used
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = min(blk * SB + (2 * L + u) * 4 + row4, rows - 1);
            load_row(elems + (size_t)r * row_elems, carries + (size_t)r * carry_step, g, u);
        }
    };
    // whole blocks: the row pointers just move on by 64 rows (the multiplications of the clamped form are quarter-rate
    // instructions, and with them the three loaders of a SIMD took longer over a step than the chain)
    const Elem *rowp[2];
    const float *kp[2];
    auto fetch_next = [&](Regs &g) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            load_row(rowp[u], kp[u], g, u);
            rowp[u] += (size_t)SB * row_elems;
            kp[u] += (size_t)SB * carry_step;
        }
    };
    auto publish = [&](int blk, const Regs &g) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (VEC && PIX == ZG_PIXEL_U8) e[j] = (g.w[u][0] >> (8 * j)) & 0xffu;
                else if constexpr (VEC && PIX == ZG_PIXEL_RGBA_U8) e[j] = (g.w[u][j] >> (8 * ch)) & 0xffu;
                else if constexpr (IS_F32) e[j] = (uint32_t)__builtin_bit_cast(float, g.w[u][j]); // integer-valued by contract
                else e[j] = g.w[u][j];
            }
            if (!all_live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] &= cmask[j];
            }
            const uint32_t p1 = e[0] + e[1], p2 = p1 + e[2], p3 = p2 + e[3];
            // inclusive scan of the lane totals over the four lanes of the strip: quad_perm [0,0,1,2] then [0,1,0,1]
            const uint32_t y = p3 + ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)p3, 0x90, 0xf, 0xf, false) & m1);
            const uint32_t z = y + ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x44, 0xf, 0xf, false) & m2);
            const uint32_t base = (uint32_t)g.k[u] + (z - p3); // carry of the strip + the lanes to my left: integers below 2^24
            *(float4 *)&ring[blk & 1][(2 * L + u) * 4 + row4][4 * q] = make_float4((float)(base + e[0]), (float)(base + p1), (float)(base + p2), (float)(base + p3));
        }
    };
    Regs g[D];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) fetch(d, g[d]);
    // whole groups of D steps run without a condition in sight: with one, the compiler can no longer count the loads in flight
    // at the loop head and waits for all but the newest, emptying the pipe once per group
    int blk0 = 0;
    const int nfull = rows / SB;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = min((D - 1) * SB + (2 * L + u) * 4 + row4, rows - 1);
        rowp[u] = elems + (size_t)r * row_elems;
        kp[u] = carries + (size_t)r * carry_step;
    }
    for (; blk0 + 2 * D - 1 <= nfull; blk0 += D) { // every block fetched here is whole
#pragma unroll
        for (int d = 0; d < D; ++d) { // static register rotation
            fetch_next(g[(d + D - 1) % D]);
            publish(blk0 + d, g[d]);
            __syncthreads(); // barrier blk0 + d
        }
    }
    for (; blk0 + D <= nblocks; blk0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) { // static register rotation
            fetch(blk0 + d + D - 1, g[(d + D - 1) % D]);
            publish(blk0 + d, g[d]);
            __syncthreads(); // barrier blk0 + d
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (blk0 + d < nblocks) { // workgroup-uniform; the loads are all in flight already
            publish(blk0 + d, g[d]);
            __syncthreads();
        }
    }
    __syncthreads(); // the two draining steps
    __syncthreads();
}

template <int PIX>
__device__ __forceinline__ void sat_chain_body(const DImg &src, const float *carries, float *sat, size_t pstride, int nstrips, int ch,
                                               float (*ring)[SAT_SB][64], float (*oring)[SAT_SB][64]) {
    using P = Px<PIX>;
    using Elem = typename P::Elem;
    constexpr int C = P::C, SB = SAT_SB, NS = 4, RS = SB / NS; // storer waves, rows per storer and block
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, simd = wave & 3;
    if (simd == 0 && wave != 0) return; // the chain keeps its SIMD to itself
    const int rows = src.rows, cols = src.cols;
    const int nblocks = (rows + SB - 1) / SB;
    const int x0 = blockIdx.x * 64;
    const bool all_live = x0 + 64 <= cols; // workgroup-uniform

    // Step k (between barrier k and barrier k + 1): the loaders publish block k + 1 into ring[(k + 1) & 1], the chain turns
    // ring[k & 1] into oring[k & 1], the storers write block k - 1 out of oring[(k - 1) & 1]. Every slot is rewritten one
    // step after its last reader finished. Two extra steps drain the pipe.
    if (wave == 0) { // ---- the chain ------------------------------------------------------------------------------------------
        float run = 0.0f;
        for (int blk = 0; blk < nblocks + 2; ++blk) {
            __syncthreads(); // block blk is in ring[blk & 1]
            if (blk < nblocks) {
                float p[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) p[i] = ring[blk & 1][i][lane];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    run = run + p[i];
                    oring[blk & 1][i][lane] = run;
                }
            }
        }
        return;
    }
    const int role = (wave >> 2) * 3 + simd - 1; // 0..11
    float *plane = sat + (size_t)ch * pstride;
    if (role >= 8) { // ---- a storer: rows sub * RS .. + RS of every block, out of the chain's ring -----------------------------------
        const int sub = role - 8, row4 = lane >> 4, q = lane & 15;
        const bool vec = all_live && (cols & 3) == 0 && ((uintptr_t)plane & 15) == 0; // 16-byte stores: four rows per instruction
        float *o = plane + (size_t)(sub * RS + row4) * cols + x0 + 4 * q; // row sub * RS + row4 of block 0; the four pieces are 4 rows apart
        const size_t piece = (size_t)4 * cols, block_step = (size_t)SB * cols;
        for (int blk = 0; blk < nblocks + 2; ++blk) {
            if (blk >= 2) {
                const int ob = blk - 2, r0 = ob * SB + sub * RS;
                float4 v[RS / 4];
#pragma unroll
                for (int h = 0; h < RS / 4; ++h) v[h] = *(const float4 *)&oring[ob & 1][sub * RS + h * 4 + row4][4 * q];
                if (vec && r0 + RS <= rows) { // the common case: no predicate, no multiplication
                    // Nontemporal: the stores are what bounds this kernel (41.7 us without them against 77.8 us for the four planes of a
                    // 4096^2 Rgba(u8) frame), nothing reads the SAT before the kernel ends, and kept out of the caches' way the 268 MB go
                    // out in 53.8 us — and leave k_box_mean, which runs next, a cleaner cache: 79.6 -> 69.0 us. (A strip-major SAT, one
                    // contiguous stream per workgroup, was measured too: no better here, and 96 us in k_box_mean.)
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int h = 0; h < RS / 4; ++h) __builtin_nontemporal_store(f32x4{v[h].x, v[h].y, v[h].z, v[h].w}, (f32x4 *)(o + h * piece));
                } else {
#pragma unroll
                    for (int h = 0; h < RS / 4; ++h) {
                        if (r0 + h * 4 + row4 < rows) {
                            const float e[4] = {v[h].x, v[h].y, v[h].z, v[h].w};
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (x0 + 4 * q + j < cols) o[h * piece + j] = e[j];
                        }
                    }
                }
                o += block_step;
            }
            __syncthreads(); // barrier blk
        }
        return;
    }
    // ---- a loader -------------------------------------------------------------------------------------------------------------
    // one wide load per lane and row when the four elements are contiguous and aligned (always, for images this library allocated)
    bool vec = false;
    if constexpr (PIX == ZG_PIXEL_U8) vec = all_live && (src.stride & 3) == 0 && ((uintptr_t)src.data & 3) == 0;
    if constexpr (PIX == ZG_PIXEL_RGBA_U8 || PIX == ZG_PIXEL_F32) vec = all_live && (src.stride * sizeof(Elem) * C) % 16 == 0 && ((uintptr_t)src.data & 15) == 0;
    if constexpr (PIX == ZG_PIXEL_U8 || PIX == ZG_PIXEL_RGBA_U8 || PIX == ZG_PIXEL_F32) {
        if (vec) { sat_loader<PIX, true>(src, carries, nstrips, ch, ring, role, lane, nblocks); return; }
    }
    sat_loader<PIX, false>(src, carries, nstrips, ch, ring, role, lane, nblocks);
}
template <int PIX>
__global__ __launch_bounds__(SAT_THREADS) void k_sat_chain(DImg src, const float *carries, float *sat, size_t pstride, int nstrips, size_t src_frame, size_t sat_frame) {
    __shared__ float ring[2][SAT_SB][64], oring[2][SAT_SB][64];
    src.data = (char *)src.data + (size_t)blockIdx.z * src_frame; // blockIdx.z: the frame of a batch
    sat_chain_body<PIX>(src, carries + (size_t)blockIdx.z * src.rows * nstrips * Px<PIX>::C, sat + (size_t)blockIdx.z * sat_frame, pstride, nstrips, (int)blockIdx.y, ring, oring);
}

// Several single-channel planes of one size in one launch (blockIdx.y picks the plane): a lone plane gives the chain kernel only
// cols / 64 workgroups, and it is latency-bound, so the planes of the Shen-Castan detector cost the time of one.
struct SatPlanes {
    DImg src[3];
    float *sat[3];
    int f32[3]; // element type of the plane: f32 (holding integers 0..255) or u8
};
__global__ __launch_bounds__(256) void k_strip_carries_planes(SatPlanes pl, float *carries, int nstrips) {
    __shared__ uint32_t wsum[2][4][1];
    const int p = blockIdx.y;
    float *table = carries + (size_t)p * pl.src[0].rows * nstrips;
    if (pl.f32[p]) strip_carries_body<ZG_PIXEL_F32>(pl.src[p], table, nstrips, wsum);
    else strip_carries_body<ZG_PIXEL_U8>(pl.src[p], table, nstrips, wsum);
}
__global__ __launch_bounds__(SAT_THREADS) void k_sat_chain_planes(SatPlanes pl, const float *carries, int nstrips) {
    __shared__ float ring[2][SAT_SB][64], oring[2][SAT_SB][64];
    const int p = blockIdx.y;
    const float *table = carries + (size_t)p * pl.src[0].rows * nstrips;
    if (pl.f32[p]) sat_chain_body<ZG_PIXEL_F32>(pl.src[p], table, pl.sat[p], 0, nstrips, 0, ring, oring);
    else sat_chain_body<ZG_PIXEL_U8>(pl.src[p], table, pl.sat[p], 0, nstrips, 0, ring, oring);
}

// Integral image(s) of `src` (Image(T).Integral.compute, integral.zig:95-140): one f32 plane of rows x cols per channel,
// planar, in the reference's association order. Also used by the Shen-Castan detector (edges.hip).
// `integer_valued`: the caller knows every element is an integer in [0, 255] (always true for u8 pixels), which makes the
// row sums exact and lets the row pass run as a parallel scan.
// Exact-row sources take the fused pair of kernels, which can leave a gap between the planes (see sat_plane_stride).
static bool sat_fused_applies(const zg_image *src, bool integer_valued) {
    static const bool fused_off = getenv("ZIGNAL_HIP_SAT_UNFUSED") != nullptr;
    return (integer_valued || !pixel_is_float(src->pixel)) && src->cols <= 65536 && !fused_off; // 65536 * 255 < 2^24
}
// Planes exactly rows * cols apart are a power of two apart for the usual frame sizes: the same element of the C planes then sits in
// the same L2 channel and the same cache set, and the kernels that walk the planes together (k_sat_chain's channel workgroups,
// k_box_mean's corner reads) queue up there. Scratch planes are spread by 4352 bytes per channel instead.
static size_t sat_plane_stride(const zg_image *src, bool padded) { return (size_t)src->rows * src->cols + (padded ? 1088 : 0); }

// n_frames equally shaped frames src_frame bytes apart (their SATs sat_frame elements apart) in one launch pair where the fused kernels apply;
// the unfused kernels (f32 sources) go frame by frame.
static int sat_planes_frames_impl(const zg_image *src, float *sat, hipStream_t s, bool integer_valued, size_t pstride, uint32_t n_frames, size_t src_frame, size_t sat_frame) {
    const int C = pixel_channels(src->pixel);
    if (pstride == 0) pstride = (size_t)src->rows * src->cols;
    if (n_frames > 1 && !sat_fused_applies(src, integer_valued)) {
        for (uint32_t f = 0; f < n_frames; ++f) {
            zg_image one = *src;
            one.data = (char *)src->data + (size_t)f * src_frame;
            if (int rc = sat_planes_frames_impl(&one, sat + (size_t)f * sat_frame, s, integer_valued, pstride, 1, 0, 0)) return rc;
        }
        return ZG_OK;
    }
    const bool exact_rows = (integer_valued || !pixel_is_float(src->pixel)) && src->cols <= 65536; // 65536 * 255 < 2^24
    // exact rows: carries of the 16-column strips (a small table), then prefix + chain + store in one pass over the source
    if (sat_fused_applies(src, integer_valued)) {
        const int nstrips = (int)ceil_div(src->cols, 16u);
        float *carries = nullptr;
        if (int rc = scratch_alloc((void **)&carries, (size_t)n_frames * src->rows * nstrips * C * sizeof(float), s)) return rc;
        const int rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
            constexpr int PIX = decltype(tag)::value;
            constexpr int PC = Px<PIX>::C;
            hipLaunchKernelGGL((k_strip_carries<PIX>), dim3(src->rows, n_frames), dim3(256), 0, s, dimg(src), carries, nstrips, src_frame);
            hipLaunchKernelGGL((k_sat_chain<PIX>), dim3(ceil_div((unsigned)nstrips, 4u), (unsigned)PC, n_frames), dim3(SAT_THREADS), 0, s, dimg(src), (const float *)carries, sat, pstride,
                               nstrips, src_frame, sat_frame);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
        scratch_free(carries, s);
        return rc;
    }
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if (exact_rows) hipLaunchKernelGGL((k_sat_rows_exact<PIX>), dim3(src->rows, (unsigned)C), dim3(256), 0, s, dimg(src), sat);
        else hipLaunchKernelGGL((k_sat_rows<PIX>), dim3(ceil_div(src->rows, 64), (unsigned)C), dim3(64), 0, s, dimg(src), sat);
        hipLaunchKernelGGL(k_sat_cols, dim3(ceil_div(src->cols, 64), (unsigned)C), dim3(64), 0, s, sat, (int)src->rows, (int)src->cols);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}

int sat_planes_impl(const zg_image *src, float *sat, hipStream_t s, bool integer_valued, size_t pstride) {
    return sat_planes_frames_impl(src, sat, s, integer_valued, pstride, 1, 0, 0);
}

// Integral images of up to three single-channel planes (u8 or integer-valued f32) of one size, one launch pair for all of them.
int sat_planes_multi(const zg_image *const *srcs, float *const *sats, int count, hipStream_t s) {
    const zg_image *a = srcs[0];
    static const bool fused_off = getenv("ZIGNAL_HIP_SAT_UNFUSED") != nullptr;
    bool ok = count >= 1 && count <= 3 && a->cols <= 65536 && !fused_off;
    for (int i = 0; i < count && ok; ++i)
        ok = srcs[i]->rows == a->rows && srcs[i]->cols == a->cols && (srcs[i]->pixel == ZG_PIXEL_U8 || srcs[i]->pixel == ZG_PIXEL_F32);
    if (!ok) { // one at a time
        for (int i = 0; i < count; ++i)
            if (int rc = sat_planes_impl(srcs[i], sats[i], s, true, 0)) return rc;
        return ZG_OK;
    }
    const int nstrips = (int)ceil_div(a->cols, 16u);
    float *carries = nullptr;
    if (int rc = scratch_alloc((void **)&carries, (size_t)count * a->rows * nstrips * sizeof(float), s)) return rc;
    SatPlanes pl{};
    for (int i = 0; i < count; ++i) {
        pl.src[i] = dimg(srcs[i]);
        pl.sat[i] = sats[i];
        pl.f32[i] = srcs[i]->pixel == ZG_PIXEL_F32;
    }
    hipLaunchKernelGGL(k_strip_carries_planes, dim3(a->rows, (unsigned)count), dim3(256), 0, s, pl, carries, nstrips);
    hipLaunchKernelGGL(k_sat_chain_planes, dim3(ceil_div((unsigned)nstrips, 4u), (unsigned)count), dim3(SAT_THREADS), 0, s, pl, (const float *)carries, nstrips);
    const hipError_t e = hipGetLastError();
    scratch_free(carries, s);
    if (e != hipSuccess) { set_error("integral image: launch failed: %s", hipGetErrorString(e)); return ZG_ERR_HIP; }
    return ZG_OK;
}


// Small u8 images — rows * cols * 255 < 2^24, BASELINE configs[0] itself (256 x 256) — need no integral image at all: every SAT value is an
// integer below 2^24, so the reference's f32 SAT is exact, ((a - b) - d) + e is the window's integer sum whatever the order, and the sum can be taken
// straight from the pixels: one launch, the source read once through L1, no scratch (three launches and 2 x 4 B per element of SAT before). The
// mean and the rounding are k_box_mean's. One thread per pixel; windows up to 15 x 15 (225 loads a pixel at most, on at most 65 793 pixels).
template <int PIX, bool SHARPEN>
__global__ __launch_bounds__(256) void k_box_direct(DImg src, DImg dst, int radius) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    const int c = blockIdx.x * 64 + (int)(threadIdx.x & 63), r = blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    if (c >= dst.cols || r >= dst.rows) return;
    const int r1 = max(r - radius, 0), r2 = min(r + radius, dst.rows - 1), c1 = max(c - radius, 0), c2 = min(c + radius, dst.cols - 1);
    uint32_t acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
    for (int y = r1; y <= r2; ++y)
        for (int x = c1; x <= c2; ++x) {
            const Vec v = P::load(src.data, (size_t)y * src.stride + (size_t)x);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] += (uint32_t)v[ch];
        }
    const float area = (float)((r2 - r1 + 1) * (c2 - c1 + 1));
    const float nd = -area, r0 = __builtin_amdgcn_rcpf(area);
    const float r1f = __builtin_fmaf(__builtin_fmaf(nd, r0, 1.0f), r0, r0);
    Vec o, orig = P::zero();
    if constexpr (SHARPEN) orig = P::load(src.data, (size_t)r * src.stride + (size_t)c);
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const float sum = (float)acc[ch]; // < 2^24: exact, and equal to the reference's ((a - b) - d) + e on its exact SAT
        const float q0 = sum * r1f;
        const float q1 = __builtin_fmaf(__builtin_fmaf(nd, q0, sum), r1f, q0);
        float val = __builtin_fmaf(__builtin_fmaf(nd, q1, sum), r1f, q1); // sum / area, bit for bit (see k_box_mean)
        if constexpr (SHARPEN) {
            const float original = (float)orig[ch];
            const float twice = 2 * original;
            val = twice - val;
        }
        const float u = fminf(fmaxf(val, 0.0f), 255.0f), t = truncf(u);
        o[ch] = (uint8_t)((int)t + ((u - t) >= 0.5f ? 1 : 0));
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, o);
}

// n equally shaped frames, src_frame / dst_frame bytes apart (n = 1: one image). Batches of frames go through the three kernels in groups whose SATs fit
// a scratch block; one-plane images taller than 65 535 x 16 rows, and f32 sources, whose SAT kernels are per image, go frame by frame.
static int box_blur_frames_impl(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, uint32_t radius, bool sharpen, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "%s: %ux%u vs %ux%u", sharpen ? "sharpen" : "boxBlur",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "boxBlur / sharpen: pixel types differ");
    auto frame_of = [](const zg_image *im, size_t step, uint32_t f) { zg_image one = *im; one.data = (char *)im->data + (size_t)f * step; return one; };
    if (radius == 0) { // image.zig:639-642
        for (uint32_t f = 0; f < n; ++f) {
            const zg_image a = frame_of(src, src_frame, f), b = frame_of(dst, dst_frame, f);
            if ((rc = copy_impl(&a, &b, s))) return rc;
        }
        return ZG_OK;
    }
    if (src->rows == 0 || src->cols == 0 || n == 0) return ZG_OK;
    ZG_REQUIRE(radius < (1u << 30), ZG_ERR_INVALID_ARGUMENT, "boxBlur: radius too large");
    const int C = pixel_channels(src->pixel);
    if (n == 1 && !pixel_is_float(src->pixel) && radius <= 7 && (uint64_t)src->rows * src->cols * 255u < (1u << 24)) {
        // the in-place call (examples/src/face_alignment.zig:95) keeps the integral-image route: there every output is written after every input is read
        const char *sb = (const char *)src->data, *se = sb + ((size_t)(src->rows - 1) * src->stride + src->cols) * pixel_size(src->pixel);
        const char *db = (const char *)dst->data, *de = db + ((size_t)(dst->rows - 1) * dst->stride + dst->cols) * pixel_size(dst->pixel);
        if (se <= db || de <= sb) {
            const dim3 grid(ceil_div(dst->cols, 64), ceil_div(dst->rows, 4));
            return dispatch_pixel(src->pixel, [&](auto tag) -> int {
                constexpr int PIX = decltype(tag)::value;
                if constexpr (!std::is_same<typename Px<PIX>::Elem, float>::value) {
                    if (sharpen) hipLaunchKernelGGL((k_box_direct<PIX, true>), grid, dim3(256), 0, s, dimg(src), dimg(dst), (int)radius);
                    else hipLaunchKernelGGL((k_box_direct<PIX, false>), grid, dim3(256), 0, s, dimg(src), dimg(dst), (int)radius);
                    ZG_HIP(hipGetLastError());
                }
                return ZG_OK;
            });
        }
    }
    // u8 planes and Rgba(u8), radius 1..3: the SAT stays in LDS (box_fused.hip)
    if ((rc = try_box_fused(src, dst, n, src_frame, dst_frame, radius, sharpen, s)) >= 0) return rc;
    const size_t plane = sat_plane_stride(src, sat_fused_applies(src, false));
    const size_t sat_frame = (size_t)C * plane; // elements
    const bool buf = sat_frame * sizeof(float) < (1ull << 32);
    const unsigned grid_rows = C == 1 ? ceil_div(dst->rows, 16) : dst->rows;
    const unsigned zpf = ceil_div(grid_rows, GRID_Y_MAX);
    // frames per group: their SATs in one scratch block; gridDim.z <= 65 535
    uint32_t group = (uint32_t)std::max<size_t>(1, std::min<size_t>(n, scratch_block_budget() / (sat_frame * sizeof(float))));
    group = std::min<uint32_t>(group, 65535u / zpf);
    if (!sat_fused_applies(src, false)) group = 1;
    float *sat = nullptr;
    if ((rc = scratch_alloc((void **)&sat, (size_t)group * sat_frame * sizeof(float), s))) return rc;
    for (uint32_t f0 = 0; f0 < n && rc == ZG_OK; f0 += group) {
        const uint32_t k = std::min(group, n - f0);
        const zg_image a = frame_of(src, src_frame, f0), b = frame_of(dst, dst_frame, f0);
        if ((rc = sat_planes_frames_impl(&a, sat, s, false, plane, k, src_frame, sat_frame)) != ZG_OK) break;
        rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
            constexpr int PIX = decltype(tag)::value;
            const dim3 grid(C == 1 ? ceil_div(dst->cols, 64) : ceil_div(dst->cols, 256), std::min(grid_rows, GRID_Y_MAX), zpf * k);
            const FrameSpan fr{src_frame, dst_frame};
            if (buf && sharpen) hipLaunchKernelGGL((k_box_mean<PIX, true, true>), grid, dim3(256), 0, s, (const float *)sat, plane, dimg(&a), dimg(&b), (int)radius, sat_frame, fr, (int)zpf);
            else if (buf) hipLaunchKernelGGL((k_box_mean<PIX, false, true>), grid, dim3(256), 0, s, (const float *)sat, plane, dimg(&a), dimg(&b), (int)radius, sat_frame, fr, (int)zpf);
            else if (sharpen) hipLaunchKernelGGL((k_box_mean<PIX, true, false>), grid, dim3(256), 0, s, (const float *)sat, plane, dimg(&a), dimg(&b), (int)radius, sat_frame, fr, (int)zpf);
            else hipLaunchKernelGGL((k_box_mean<PIX, false, false>), grid, dim3(256), 0, s, (const float *)sat, plane, dimg(&a), dimg(&b), (int)radius, sat_frame, fr, (int)zpf);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
    }
    scratch_free(sat, s);
    return rc;
}

static int box_blur_impl(const zg_image *src, const zg_image *dst, uint32_t radius, bool sharpen, hipStream_t s) {
    return box_blur_frames_impl(src, dst, 1, 0, 0, radius, sharpen, s);
}

// the pipeline's box-blur step over a batch (batch.hip)
int box_blur_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, uint32_t radius, hipStream_t s) {
    return box_blur_frames_impl(src, dst, n, src_frame, dst_frame, radius, false, s);
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_box_blur(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream) {
    return box_blur_impl(src, dst, radius, false, as_stream(stream));
}

int zg_box_blur_host(const zg_image *src, const zg_image *dst, uint32_t radius) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = box_blur_impl(&a.dev, &b.dev, radius, false, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

int zg_sharpen(const zg_image *src, const zg_image *dst, uint32_t radius, zg_stream stream) {
    return box_blur_impl(src, dst, radius, true, as_stream(stream));
}

int zg_sharpen_host(const zg_image *src, const zg_image *dst, uint32_t radius) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = box_blur_impl(&a.dev, &b.dev, radius, true, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

// Image(T).integral (image.zig:628-630 -> integral.zig:95-140): planes[ch] is a rows x cols f32 image, ch-major in `planes`.
int zg_integral(const zg_image *src, float *planes, zg_stream stream) {
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    ZG_REQUIRE(planes != nullptr || src->rows == 0 || src->cols == 0, ZG_ERR_INVALID_ARGUMENT, "integral: null output");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    return sat_planes_impl(src, planes, as_stream(stream), false, 0);
}

int zg_integral_host(const zg_image *src, float *planes) {
    HostStage a;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const size_t bytes = (size_t)pixel_channels(src->pixel) * src->rows * src->cols * sizeof(float);
    ZG_REQUIRE(planes != nullptr, ZG_ERR_INVALID_ARGUMENT, "integral: null output");
    float *dev = nullptr;
    ZG_HIP(hipMalloc((void **)&dev, bytes));
    rc = sat_planes_impl(&a.dev, dev, nullptr, false, 0);
    if (rc == ZG_OK) rc = download_pageable(planes, dev, bytes, nullptr);
    (void)hipFree(dev);
    return rc;
}

} // extern "C"
