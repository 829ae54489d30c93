// conv_sep_bytes2.hip — Image(u8 / Rgb(u8) / Rgba(u8)).convolveSeparable / gaussianBlur for LONG non-negative integer
// kernels (the 11..65-tap Gaussians ImagePyramid builds: sigma = blur_sigma * sqrt(scale^2 - 1), reference
// src/image/pyramid.zig:76-85), as two packed passes over the row as a byte stream.
//
// Same arithmetic contract as conv_separable.hip (reference src/image/convolution.zig:441-647, u8 path):
// temp = sum src*kx (exact), out = divClampU8(65536, sum temp*ky), every channel with the same taps, tap j of an n-tap
// kernel reading offset j - n/2. With taps in [0,255] summing to <= 257 the temp fits u16, so
//   k_rows_u16   byte stream -> u16 temp plane (2 bytes per source byte, in HBM). A wave stages one row segment in LDS
//                (its own buffer: no workgroup barrier) and walks the taps FOUR AT A TIME in a run-time loop: with the
//                half width padded to a multiple of 4 (zero taps) every group's byte offset SP*(4g - hpad) is a whole
//                number of dwords, so the window of a group is a few aligned LDS dwords and all byte-pair extraction
//                (v_perm) and packed multiply-adds (v_pk_mad_u16) inside the group are compile-time. One kernel per
//                pixel stride, any tap count. Lanes own the dwords l, l+64, l+128, l+192 of the 1024-byte tile so that
//                dword-granular LDS reads hit consecutive banks.
//   k_cols_u16   u16 temp -> bytes. A lane owns four adjacent bytes (two packed pairs) of 32 output rows as u32
//                accumulators and streams the 32 + n - 1 temp rows past them once; the taps slide through 32 SGPRs, so
//                each (row, output) step is four v_mad_u32_u16 with a scalar tap. No LDS, no window registers.
// Border rule: columns in k_rows_u16 (edge tiles patch the out-of-row bytes of their LDS row), rows in k_cols_u16 (the
// source row index of each streamed temp row is resolved on the scalar unit).
//
// Preconditions (else the general two-pass kernels run): u8 pixel types, row length and strides multiples of 16 bytes,
// 16-byte aligned bases, row >= 256 bytes, taps as above, both tap counts <= 65.
#include "zg_common.h"
#include "zg_u8pack.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace zg {

constexpr int B2_HMAX = 32;  // padded half width of the row pass (a multiple of 4)
constexpr int B2_NKMAX = 65; // longest kernel
constexpr int B2_R = 32;     // output rows per lane in the column pass
constexpr int B2_LEFT = 32;  // LDS halo dwords left of the tile (128 bytes >= hpad * SP)
constexpr int B2_ROW = B2_LEFT + 256 + 36; // + right halo: hpad * SP <= 128 bytes and the 3 * SP + 3 bytes a group overshoots

// Batches: frame f of the launch reads src + f * src_frame bytes, goes through temp + f * temp_frame dwords, writes dst + f * dst_frame.
struct B2Frames { size_t src_frame, temp_frame, dst_frame; };

struct TapsRows { uint32_t kk[4 * (B2_HMAX / 2 + 1)]; }; // tap | tap << 16, index = offset + hpad, zero padded
struct TapsCols { uint32_t k[B2_NKMAX + 2 * B2_R]; };    // k[B2_R + j] = tap j, zeros around

template <int S, int NW> __device__ __forceinline__ u16x2 window_pair(const uint32_t (&w)[NW]) { // bytes (S, S + 1) of the window
    constexpr int d = S >> 2, o = S & 3;
    if constexpr (o == 0) return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, w[d], 0x0c010c00u));
    else if constexpr (o == 1) return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, w[d], 0x0c020c01u));
    else if constexpr (o == 2) return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, w[d], 0x0c030c02u));
    else return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(w[d + 1], w[d], 0x0c040c03u));
}

// One wave stages bytes xb0 - 128 .. xb0 + 1024 + 144 of a row in its own LDS row (the row passes below share this).
template <int SP>
__device__ __forceinline__ void stage_row(uint32_t *buf, const uint8_t *row, int xb0, int row_bytes, int lane, int hpad, bool edge, int cols, int border) {
    // stage bytes xb0 - 128 .. xb0 + 1024 + 144 of the row: 64 main units and 17 halo units of 16 bytes; units are all
    // inside or all outside the row (length % 16 == 0), outside ones are zeroed here and patched below
    {
        const int gb = xb0 + 16 * lane;
        u32x4 v = *(const u32x4 *)(row + min(gb, row_bytes - 16));
        if (gb + 16 > row_bytes) v = u32x4{0u, 0u, 0u, 0u};
        const int hu = min(lane, 16);                              // halo unit: 0..7 left, 8..16 right
        const int hb = hu < 8 ? xb0 - 128 + 16 * hu : xb0 + 1024 + 16 * (hu - 8);
        u32x4 h = *(const u32x4 *)(row + min(max(hb, 0), row_bytes - 16));
        if (hb < 0 || hb + 16 > row_bytes) h = u32x4{0u, 0u, 0u, 0u};
        *(u32x4 *)(buf + B2_LEFT + 4 * lane) = v;
        if (lane < 17) *(u32x4 *)(buf + (hu < 8 ? 4 * hu : B2_LEFT + 256 + 4 * (hu - 8))) = h;
    }
    if (edge) { // border rule for the columns, one byte per lane
        const int reach = hpad * SP + 16;
        for (int k = lane; k < 2 * reach; k += 64) {
            const int b = k < reach ? -1 - k : row_bytes + (k - reach); // byte position in the row's stream
            const int t = b - (xb0 - 128);                               // byte position in the LDS row
            if (t < 0 || t >= B2_ROW * 4) continue;
            const int px = b >= 0 ? b / SP : -((SP - 1 - b) / SP); // floor
            const int gc = resolve_index(px, cols, border);
            if (gc < 0) continue; // zero border: already 0
            ((uint8_t *)buf)[t] = row[gc * SP + (b - px * SP)];
        }
    }
}

template <int SP>
__global__ __launch_bounds__(256) void k_rows_u16(DImg src, uint32_t *temp, TapsRows taps, int hpad, int ngroups, int border,
                                                  int tiles_x, int rows_per_wave, B2Frames fr) {
    src.data = (uint8_t *)src.data + (size_t)blockIdx.y * fr.src_frame; // the frame is the grid's y
    temp += (size_t)blockIdx.y * fr.temp_frame;
    constexpr int NW = SP == 1 ? 2 : 4; // window dwords of a group: 4 output bytes + 3 * SP bytes of tap reach
    __shared__ uint32_t lds[4][B2_ROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int xb0 = tx * 1024;
    const int row_bytes = src.cols * SP;
    uint32_t *buf = lds[wave];
    const bool edge = xb0 == 0 || xb0 + 1024 + hpad * SP + 16 > row_bytes; // this tile's taps reach past a row end

    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int y = (ty * 4 + wave) * rows_per_wave + rr; // wave-uniform
        if (y >= src.rows) break;
        const uint8_t *row = (const uint8_t *)src.data + (size_t)y * src.stride * SP;
        stage_row<SP>(buf, row, xb0, row_bytes, lane, hpad, edge, src.cols, border);
        __builtin_amdgcn_wave_barrier(); // LDS is in order within a wave; this only stops the compiler from reordering

        u16x2 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = u16x2{0, 0};
        for (int g = 0; g < ngroups; ++g) {
            const uint32_t k0 = taps.kk[4 * g], k1 = taps.kk[4 * g + 1], k2 = taps.kk[4 * g + 2], k3 = taps.kk[4 * g + 3];
            const int cdw = (SP * (4 * g - hpad)) / 4; // exact: SP == 4 or hpad % 4 == 0
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t w[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) w[j] = buf[B2_LEFT + lane + 64 * m + cdw + j];
                // output bytes (0,1) and (2,3) of this dword; tap t reads SP * t bytes further on
                acc[m][0] += window_pair<0, NW>(w) * __builtin_bit_cast(u16x2, k0);
                acc[m][1] += window_pair<2, NW>(w) * __builtin_bit_cast(u16x2, k0);
                acc[m][0] += window_pair<SP, NW>(w) * __builtin_bit_cast(u16x2, k1);
                acc[m][1] += window_pair<2 + SP, NW>(w) * __builtin_bit_cast(u16x2, k1);
                acc[m][0] += window_pair<2 * SP, NW>(w) * __builtin_bit_cast(u16x2, k2);
                acc[m][1] += window_pair<2 + 2 * SP, NW>(w) * __builtin_bit_cast(u16x2, k2);
                acc[m][0] += window_pair<3 * SP, NW>(w) * __builtin_bit_cast(u16x2, k3);
                acc[m][1] += window_pair<2 + 3 * SP, NW>(w) * __builtin_bit_cast(u16x2, k3);
            }
        }
        __builtin_amdgcn_wave_barrier(); // the next row's staging must not overtake these reads

        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        uint32_t *trow = temp + ((size_t)y * row_bytes + xb0) / 2; // two bytes of the stream per packed u32
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int b = xb0 + 4 * (lane + 64 * m);
            if (b < row_bytes)
                *(u32x2 *)(trow + 2 * (lane + 64 * m)) = u32x2{__builtin_bit_cast(uint32_t, acc[m][0]), __builtin_bit_cast(uint32_t, acc[m][1])};
        }
    }
}

// The row pass of a grey plane in f32 (exact: a temp is at most 255 * 257). A lane owns SIXTEEN consecutive bytes of the tile; it
// converts the 16 + 2 HP bytes its taps reach once (v_cvt_f32_ubyteN) and every tap of every output is one v_fmac_f32 with the tap
// in an SGPR — no byte-pair extraction, and an f32 multiply-add issues in half the time of a packed u16 one (tools/exp/valu_rate.hip).
// HP (a multiple of 4, the kernel's half width rounded up) is a template parameter so that all register indices are compile-time;
// taps past the kernel are zeros.
struct TapsRowsU8F { float k[2 * B2_HMAX + 1]; }; // k[offset + HP]
// tap(j) = the tap at offset j - HP (a scalar load out of the kernel arguments, compile-time j); bx = the workgroup's index within its plane
template <int HP, typename TapFn>
__device__ __forceinline__ void rows_u8f_body(DImg src, uint32_t *temp, TapFn tap, int border, int tiles_x, int rows_per_wave, uint32_t (*lds)[B2_ROW], int bx) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int ND = 4 + HP / 2; // dwords of a lane's window
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int xb0 = tx * 1024;
    const int row_bytes = src.cols;
    uint32_t *buf = lds[wave];
    const bool edge = xb0 == 0 || xb0 + 1024 + HP + 16 > row_bytes;
    float k[2 * HP + 1];
#pragma unroll
    for (int j = 0; j <= 2 * HP; ++j) k[j] = tap(j); // scalar loads: the taps live in SGPRs
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int y = (ty * 4 + wave) * rows_per_wave + rr; // wave-uniform
        if (y >= src.rows) break;
        const uint8_t *row = (const uint8_t *)src.data + (size_t)y * src.stride;
        stage_row<1>(buf, row, xb0, row_bytes, lane, HP, edge, src.cols, border);
        __builtin_amdgcn_wave_barrier();
        float w[4 * ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const uint32_t v = buf[B2_LEFT + 4 * lane - HP / 4 + d];
            w[4 * d] = (float)(v & 0xffu); w[4 * d + 1] = (float)((v >> 8) & 0xffu); w[4 * d + 2] = (float)((v >> 16) & 0xffu); w[4 * d + 3] = (float)(v >> 24);
        }
        __builtin_amdgcn_wave_barrier(); // the next row's staging must not overtake these reads
        float acc[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = 0.0f;
#pragma unroll
        for (int j = 0; j <= 2 * HP; ++j) {
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(w[o + j], k[j], acc[o]);
        }
        if (xb0 + 16 * lane < row_bytes) {
            uint32_t *trow = temp + ((size_t)y * row_bytes + xb0) / 2 + 8 * lane; // two bytes of the stream per packed u32
            uint32_t pk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = (uint32_t)acc[2 * i] | ((uint32_t)acc[2 * i + 1] << 16);
            *(u32x4 *)trow = u32x4{pk[0], pk[1], pk[2], pk[3]};
            *(u32x4 *)(trow + 4) = u32x4{pk[4], pk[5], pk[6], pk[7]};
        }
    }
}
template <int HP>
__global__ __launch_bounds__(256) void k_rows_u8f(DImg src, uint32_t *temp, TapsRowsU8F taps, int border, int tiles_x, int rows_per_wave, B2Frames fr) {
    src.data = (uint8_t *)src.data + (size_t)blockIdx.y * fr.src_frame;
    temp += (size_t)blockIdx.y * fr.temp_frame;
    __shared__ uint32_t lds[4][B2_ROW];
    rows_u8f_body<HP>(src, temp, [&](int j) { return taps.k[j]; }, border, tiles_x, rows_per_wave, lds, (int)blockIdx.x);
}
// The row passes of SEVERAL Gaussians over ONE source in one launch (the levels of a pyramid: every level blurs the original with its own sigma):
// blockIdx.y picks the job. One launch of n x 4 096 waves instead of n launches of 4 096: a launch of one round of waves is a memory phase and a
// compute phase one after the other plus 4 us of ramp (profiles/r05_experiments.txt, section 5), several rounds overlap them.
constexpr int PYR_MAX_JOBS = 8;
constexpr int PYR_HMAX = 16; // padded half width of a batched row pass (taps <= 33)
struct RowsJob { uint32_t *temp; int hp; float k[2 * PYR_HMAX + 1]; }; // k[offset + hp]
struct RowsJobs { RowsJob j[PYR_MAX_JOBS]; };
__global__ __launch_bounds__(256) void k_rows_u8f_multi(DImg src, RowsJobs jobs, int border, int tiles_x, int rows_per_wave) {
    __shared__ uint32_t lds[4][B2_ROW];
    const RowsJob &job = jobs.j[blockIdx.y];
    auto tap = [&](int j) { return job.k[j]; };
    switch (job.hp) { // workgroup-uniform
    case 4: rows_u8f_body<4>(src, job.temp, tap, border, tiles_x, rows_per_wave, lds, (int)blockIdx.x); break;
    case 8: rows_u8f_body<8>(src, job.temp, tap, border, tiles_x, rows_per_wave, lds, (int)blockIdx.x); break;
    case 12: rows_u8f_body<12>(src, job.temp, tap, border, tiles_x, rows_per_wave, lds, (int)blockIdx.x); break;
    default: rows_u8f_body<16>(src, job.temp, tap, border, tiles_x, rows_per_wave, lds, (int)blockIdx.x); break;
    }
}
template <int HP>
static void launch_rows_u8f(const zg_image *src, uint32_t *temp, const int32_t *ix, int nkx, int border, int tiles_x, int rows_per_wave, dim3 grid, const B2Frames &fr, hipStream_t s) {
    TapsRowsU8F t{};
    for (int j = 0; j < nkx; ++j) t.k[j - nkx / 2 + HP] = (float)ix[j];
    hipLaunchKernelGGL((k_rows_u8f<HP>), grid, dim3(256), 0, s, dimg(src), temp, t, border, tiles_x, rows_per_wave, fr);
}

template <bool CLAMP, bool INSIDE>
__device__ __forceinline__ void cols_strip(const uint32_t *temp, uint8_t *dst, size_t dst_pitch, int rows, int row_bytes,
                                           const TapsCols &taps, int nk, int half, int border, int tx, int ty) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const int xd = tx * 256 + (int)threadIdx.x; // this lane's dword of the row (4 bytes, 2 packed temp pairs)
    const bool live = xd * 4 < row_bytes;
    const int xdc = live ? xd : 0;
    const int y0 = ty * B2_R;
    const size_t trow = (size_t)row_bytes / 2;

    uint32_t acc[B2_R][4];
#pragma unroll
    for (int o = 0; o < B2_R; ++o) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 32768u; // divClampU8's rounding term
    const int nrows_in = B2_R + nk - 1;
    // Scalar instructions share the SIMD's issue slots with vector ones, so the per-row bookkeeping is kept minimal:
    // strips whose 32 + nk - 1 rows all lie inside the image skip the border resolution, and the "is this tap inside
    // the kernel" test is made once per 4 rows x 4 outputs.
    const uint32_t *tcol = temp + 2 * (size_t)xdc;
    const uint32_t *next_row = tcol + (size_t)(INSIDE ? y0 - half : 0) * trow; // INSIDE: a running pointer, one add per row
    auto fetch = [&](int r) -> u32x2 { // temp row y0 - half + r; called with r = 0, 1, 2, ... in order
        if constexpr (INSIDE) { // rows past the strip's last (prefetch overshoot) fall in the temp plane's slack rows
            const u32x2 p = *(const u32x2 *)next_row;
            next_row += trow;
            return p;
        } else {
            const int gr = resolve_index(y0 - half + min(r, nrows_in - 1), rows, border); // scalar
            u32x2 p = *(const u32x2 *)(tcol + (size_t)max(gr, 0) * trow);
            if (gr < 0) p = u32x2{0u, 0u};
            return p;
        }
    };
    // rows are taken eight at a time with the next eight already in flight
    u32x2 cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = fetch(i);
    for (int r0 = 0; r0 < nrows_in; r0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = fetch(r0 + 8 + i);
        // the taps of this chunk: output row o takes k[r - o] from temp row r = r0 + i, i.e. entry 32 + i - o of the
        // 40-entry slice of the zero-padded tap table that starts at r0 (scalar loads, compile-time indices below).
        // Rows past the last one and outputs outside the kernel's reach meet zero taps or are skipped.
        uint32_t kw[40];
#pragma unroll
        for (int c = 0; c < 40; ++c) kw[c] = taps.k[r0 + c];
#pragma unroll
        for (int ib = 0; ib < 8; ib += 4) {
#pragma unroll
            for (int ob = 0; ob < B2_R; ob += 4) {
                // rows r0+ib .. +3 against outputs ob .. +3: some tap index r - o in [0, nk)?
                if (r0 + ib + 3 >= ob && r0 + ib - ob - 3 < nk) {
#pragma unroll
                    for (int i = ib; i < ib + 4; ++i) {
#pragma unroll
                        for (int o = ob; o < ob + 4; ++o) {
                            const uint32_t k = __builtin_amdgcn_readfirstlane(kw[32 + i - o]);
                            acc[o][0] = mad_lo16(cur[i][0], k, acc[o][0]);
                            acc[o][1] = mad_hi16(cur[i][0], k, acc[o][1]);
                            acc[o][2] = mad_lo16(cur[i][1], k, acc[o][2]);
                            acc[o][3] = mad_hi16(cur[i][1], k, acc[o][3]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    if (!live) return;
#pragma unroll
    for (int o = 0; o < B2_R; ++o) {
        const int y = y0 + o;
        if (y >= rows) break;
        uint32_t out;
        if constexpr (CLAMP) {
            uint32_t v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const uint32_t t = acc[o][c] >> 16; v[c] = t > 255u ? 255u : t; }
            out = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        } else { // host proved acc < 2^24: the value is byte 2
            out = __builtin_amdgcn_perm(acc[o][1], acc[o][0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[o][3], acc[o][2], 0x06020c0cu);
        }
        *(uint32_t *)(dst + (size_t)y * dst_pitch + 4 * (size_t)xd) = out;
    }
}

template <bool CLAMP>
__global__ __launch_bounds__(256) void k_cols_u16(const uint32_t *temp, uint8_t *dst, size_t dst_pitch, int rows, int row_bytes,
                                                  TapsCols taps, int nk, int half, int border, int tiles_x, B2Frames fr) {
    temp += (size_t)blockIdx.y * fr.temp_frame;
    dst += (size_t)blockIdx.y * fr.dst_frame;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * B2_R;
    if (y0 - half >= 0 && y0 - half + B2_R + nk - 1 <= rows) // workgroup-uniform: every streamed row is inside the image
        cols_strip<CLAMP, true>(temp, dst, dst_pitch, rows, row_bytes, taps, nk, half, border, tx, ty);
    else
        cols_strip<CLAMP, false>(temp, dst, dst_pitch, rows, row_bytes, taps, nk, half, border, tx, ty);
}

// The column pass of every normalised kernel. When the host proved sum(kx) * sum(ky) * 255 + 32768 < 2^24 each partial sum is an
// integer below 2^24 and f32 holds it exactly, so the multiply-adds run as (packed) f32 FMAs with the taps as f32 in SGPRs — two
// per instruction where v_mad_u32_u16 does one at the same issue cost (tools/exp/valu_rate.hip) — and the temps are converted once
// per streamed row. A lane owns TWO adjacent bytes (one packed temp dword) of 32 output rows: 64 accumulator registers instead of
// 128, so four waves fit a SIMD where the integer form fits two, and a 4096-byte row gives 4 096 waves instead of 2 048.
//
// WIDE: tap sums of 257 (what rounding the taps to 1/256 can leave: ORB's pyramid level 4) take the sum to 2^24 + 98 047 at most, where
// divClampU8 starts to clamp and f32 stops being exact. The accumulators then start 2^23 lower — every partial sum lies in
// [-2^23, 2^23 + 98 047], all exact — the clamp is a v_min against 255.99.. (as an integer: 2^24 - 1 - 2^23) before the conversion, and
// adding the 2^23 back is flipping the top bit of byte 2, the byte that is the result.
constexpr float U8F_BIAS = 8388608.0f;
template <bool INSIDE, bool WIDE, int R>
__device__ __forceinline__ void cols_strip_u8f(const uint32_t *temp, uint8_t *dst, size_t dst_pitch, int rows, int row_bytes,
                                               const TapsCols &taps, int nk, int half, int border, int tx, int ty) {
    const int xd = tx * 256 + (int)threadIdx.x; // this lane's packed temp dword of the row (2 bytes of output)
    const bool live = xd * 2 < row_bytes;
    const int xdc = live ? xd : 0;
    const int y0 = ty * R;
    const size_t trow = (size_t)row_bytes / 2;
    float acc[R][2];
#pragma unroll
    for (int o = 0; o < R; ++o) acc[o][0] = acc[o][1] = WIDE ? 32768.0f - U8F_BIAS : 32768.0f; // divClampU8's rounding term
    const int nrows_in = R + nk - 1;
    const uint32_t *tcol = temp + (size_t)xdc;
    const uint32_t *next_row = tcol + (size_t)(INSIDE ? y0 - half : 0) * trow;
    auto fetch = [&](int r) -> uint32_t { // temp row y0 - half + r; called with r = 0, 1, 2, ... in order
        if constexpr (INSIDE) { // rows past the strip's last (prefetch overshoot) fall in the temp plane's slack rows
            const uint32_t p = *next_row;
            next_row += trow;
            return p;
        } else {
            const int gr = resolve_index(y0 - half + min(r, nrows_in - 1), rows, border); // scalar
            const uint32_t p = tcol[(size_t)max(gr, 0) * trow];
            return gr < 0 ? 0u : p;
        }
    };
    uint32_t cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = fetch(i);
    for (int r0 = 0; r0 < nrows_in; r0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = fetch(r0 + 8 + i);
        uint32_t kw[R + 8]; // as in cols_strip: entry R + i - o is the tap output row o takes from temp row r0 + i
#pragma unroll
        for (int c = 0; c < R + 8; ++c) kw[c] = taps.k[r0 + (B2_R - R) + c];
#pragma unroll
        for (int ib = 0; ib < 8; ib += 4) {
#pragma unroll
            for (int ob = 0; ob < R; ob += 4) {
                if (r0 + ib + 3 >= ob && r0 + ib - ob - 3 < nk) {
#pragma unroll
                    for (int i = ib; i < ib + 4; ++i) {
                        const float t0 = (float)(cur[i] & 0xffffu), t1 = (float)(cur[i] >> 16);
#pragma unroll
                        for (int o = ob; o < ob + 4; ++o) {
                            const float k = __uint_as_float(__builtin_amdgcn_readfirstlane(kw[R + i - o]));
                            acc[o][0] = __builtin_fmaf(t0, k, acc[o][0]);
                            acc[o][1] = __builtin_fmaf(t1, k, acc[o][1]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    if (!live) return;
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const int y = y0 + o;
        if (y >= rows) break;
        uint32_t out;
        if constexpr (WIDE) {
            const int v0 = (int)fminf(acc[o][0], 16777215.0f - U8F_BIAS), v1 = (int)fminf(acc[o][1], 16777215.0f - U8F_BIAS);
            out = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x0c0c0602u) ^ 0x8080u;
        } else { // acc < 2^24: the value is byte 2 of the (exact) integer
            out = __builtin_amdgcn_perm((uint32_t)acc[o][1], (uint32_t)acc[o][0], 0x0c0c0602u);
        }
        *(uint16_t *)(dst + (size_t)y * dst_pitch + 2 * (size_t)xd) = (uint16_t)out;
    }
}

template <bool WIDE, int R>
__global__ __launch_bounds__(256) void k_cols_u8f(const uint32_t *temp, uint8_t *dst, size_t dst_pitch, int rows, int row_bytes,
                                                  TapsCols taps, int nk, int half, int border, int tiles_x, B2Frames fr) {
    temp += (size_t)blockIdx.y * fr.temp_frame;
    dst += (size_t)blockIdx.y * fr.dst_frame;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * R;
    if (y0 - half >= 0 && y0 - half + R + nk - 1 <= rows) // workgroup-uniform: every streamed row is inside the image
        cols_strip_u8f<true, WIDE, R>(temp, dst, dst_pitch, rows, row_bytes, taps, nk, half, border, tx, ty);
    else
        cols_strip_u8f<false, WIDE, R>(temp, dst, dst_pitch, rows, row_bytes, taps, nk, half, border, tx, ty);
}

// The same column pass with the tap count a template parameter (rounded up to 4 m + 1, zero taps behind the kernel): k_cols_u8f spends two of
// every five issue slots on scalar bookkeeping — "does this 4 x 4 block of (row, output) pairs meet a tap" sixteen times per eight rows, the tap
// slice reloaded per chunk (PMC: 820 SALU against 1 220 VALU instructions per wave, and SALU shares the SIMD's issue slots: profiles/r05_experiments.txt).
// With NK4 fixed every loop unrolls, the taps sit in SGPRs from the start and the bands inside the plane run the multiply-adds and nothing else.
// Bands that touch the top or bottom rows take cols_strip_u8f's general form.
template <int NK4, bool WIDE>
__device__ __forceinline__ void cols_strip_u8f_static(const uint32_t *temp, uint8_t *dst, size_t dst_pitch, int row_bytes, const TapsCols &taps, int half,
                                                      int tx, int ty) {
    constexpr int R = B2_R, NR = R + NK4 - 1;
    const int xd = tx * 256 + (int)threadIdx.x;
    const bool live = xd * 2 < row_bytes;
    const int xdc = live ? xd : 0;
    const int y0 = ty * R;
    const size_t trow = (size_t)row_bytes / 2;
    float k[NK4];
#pragma unroll
    for (int j = 0; j < NK4; ++j) k[j] = __uint_as_float(taps.k[B2_R + j]); // kernel arguments: scalar loads, the taps live in SGPRs
    float acc[R][2];
#pragma unroll
    for (int o = 0; o < R; ++o) acc[o][0] = acc[o][1] = WIDE ? 32768.0f - U8F_BIAS : 32768.0f;
    const uint32_t *next_row = temp + (size_t)xdc + (size_t)(y0 - half) * trow;
    uint32_t cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { cur[i] = *next_row; next_row += trow; }
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (r0 + 8 + i < NR) { nxt[i] = *next_row; next_row += trow; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = r0 + i; // temp row y0 - half + r meets output row o with tap r - o
            if (r < NR) {
                const float t0 = (float)(cur[i] & 0xffffu), t1 = (float)(cur[i] >> 16);
#pragma unroll
                for (int o = 0; o < R; ++o) {
                    if (r - o >= 0 && r - o < NK4) {
                        acc[o][0] = __builtin_fmaf(t0, k[r - o], acc[o][0]);
                        acc[o][1] = __builtin_fmaf(t1, k[r - o], acc[o][1]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    if (!live) return;
    uint8_t *out_row = dst + (size_t)y0 * dst_pitch + 2 * (size_t)xd;
#pragma unroll
    for (int o = 0; o < R; ++o) {
        uint32_t out;
        if constexpr (WIDE) {
            const int v0 = (int)fminf(acc[o][0], 16777215.0f - U8F_BIAS), v1 = (int)fminf(acc[o][1], 16777215.0f - U8F_BIAS);
            out = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x0c0c0602u) ^ 0x8080u;
        } else {
            out = __builtin_amdgcn_perm((uint32_t)acc[o][1], (uint32_t)acc[o][0], 0x0c0c0602u);
        }
        *(uint16_t *)out_row = (uint16_t)out;
        out_row += dst_pitch;
    }
}

template <int NK4, bool WIDE>
__global__ __launch_bounds__(256) void k_cols_u8f_static(const uint32_t *temp, uint8_t *dst, size_t dst_pitch, int rows, int row_bytes, TapsCols taps, int nk,
                                                         int half, int border, int tiles_x, B2Frames fr) {
    temp += (size_t)blockIdx.y * fr.temp_frame;
    dst += (size_t)blockIdx.y * fr.dst_frame;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * B2_R;
    if (y0 - half >= 0 && y0 - half + B2_R + NK4 - 1 <= rows) // workgroup-uniform: every streamed row (the padded ones too) is inside the plane
        cols_strip_u8f_static<NK4, WIDE>(temp, dst, dst_pitch, row_bytes, taps, half, tx, ty);
    else
        cols_strip_u8f<false, WIDE, B2_R>(temp, dst, dst_pitch, rows, row_bytes, taps, nk, half, border, tx, ty);
}

// ---- one pyramid level: gaussianBlur then resize(.bilinear) of an Image(u8), the column pass fused with the resize (round 5) -------------------------
// ImagePyramid.build blurs the WHOLE source for every level and then samples a shrinking part of it (reference src/image/pyramid.zig:76-92): the bilinear
// resize of level i reads two blurred rows and two blurred columns per output row / column — 4 / scale^2 of the blurred pixels (31 % at level 7), and the
// blurred plane itself is only ever an intermediate. So the row pass runs as it is (k_rows_u8f: u8 -> u16 temp), and the column pass is evaluated where the
// resize looks: a lane owns the two source COLUMNS its output column taps (adjacent: one unaligned 4-byte gather per streamed temp row), keeps 32 consecutive
// blurred rows of them as f32 accumulators exactly as cols_strip_u8f does (the reference's integer sums, exact in f32 below 2^24), leaves the 32 x 2 bytes in
// its own LDS column and interpolates every output row whose two source rows lie in the band — Image(u8).resize's expressions (interpolation.zig:313-407 via
// k_resize_bilinear_u8: fx = round(frac * 256), (top (256 - fy) + bottom fy + 32768) >> 16, mirror-resolved neighbours). No blurred plane, no resize launch,
// 2 / scale of the column pass's arithmetic. Bands of 32 source rows advance by 31 so that both rows of every output row share a band (the last band is
// anchored at the bottom). Integer arithmetic throughout, so evaluating the passes where they are needed changes no bit.
struct PyrLevelArgs {
    const uint16_t *temp; // k_rows_u8f's plane: u16 per source byte, rows x cols (+ slack rows)
    uint8_t *dst;
    size_t dst_pitch;
    int rows, cols, drows, dcols;
    float rx, ry;         // (float)cols / dcols, (float)rows / drows: Image.resize's ratios
    int nk, half, tiles_x, nbands;
};

template <bool INSIDE, bool WIDE>
__device__ __forceinline__ void cols_bilinear_band(const PyrLevelArgs &a, const TapsCols &taps, uint16_t (*bt)[256], int tx, int band) {
    constexpr int BAND = B2_R - 1;
    const int tid = (int)threadIdx.x;
    const int c = tx * 256 + tid;
    const bool live = c < a.dcols;
    const int cc = live ? c : a.dcols - 1;
    const int y0 = band == a.nbands - 1 ? max(0, a.rows - B2_R) : band * BAND;
    // the output column's taps (k_resize_bilinear_u8's expressions)
    const float sx = ((float)cc + 0.5f) * a.rx - 0.5f;
    const float fl = floorf(sx);
    const int left = (int)fl;
    int cl = left, cr = left + 1;
    if (left < 0 || left + 1 >= a.cols) {
        cl = resolve_index(left, a.cols, ZG_BORDER_MIRROR);
        cr = resolve_index(left + 1, a.cols, ZG_BORDER_MIRROR);
    }
    const int fx = (int)roundf((sx - fl) * 256);
    const int base = min(cl, cr); // the two columns are neighbours (a reduction of a plane at least two columns wide): one 4-byte gather
    const bool swapped = cl > cr; // at the mirrored right edge the left tap is the higher column

    float acc[B2_R][2];
#pragma unroll
    for (int o = 0; o < B2_R; ++o) acc[o][0] = acc[o][1] = WIDE ? 32768.0f - U8F_BIAS : 32768.0f; // divClampU8's rounding term (WIDE: cols_strip_u8f)
    const int nrows_in = B2_R + a.nk - 1;
    const size_t trow = (size_t)a.cols; // u16 per temp row
    const uint16_t *tcol = a.temp + base;
    const uint16_t *next_row = tcol + (size_t)(INSIDE ? y0 - a.half : 0) * trow;
    auto fetch = [&](int r) -> uint32_t { // temp row y0 - half + r; called with r = 0, 1, 2, ... in order
        uint32_t p;
        if constexpr (INSIDE) { // rows past the band's last (prefetch overshoot) fall in the temp plane's slack rows
            __builtin_memcpy(&p, next_row, 4);
            next_row += trow;
        } else {
            const int gr = resolve_index(y0 - a.half + min(r, nrows_in - 1), a.rows, ZG_BORDER_MIRROR); // scalar; gaussianBlur's border rule
            __builtin_memcpy(&p, tcol + (size_t)max(gr, 0) * trow, 4);
        }
        return p;
    };
    uint32_t cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = fetch(i);
    for (int r0 = 0; r0 < nrows_in; r0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = fetch(r0 + 8 + i);
        uint32_t kw[40]; // as in cols_strip: entry 32 + i - o is the tap blurred row o takes from temp row r0 + i
#pragma unroll
        for (int k = 0; k < 40; ++k) kw[k] = taps.k[r0 + k];
#pragma unroll
        for (int ib = 0; ib < 8; ib += 4) {
#pragma unroll
            for (int ob = 0; ob < B2_R; ob += 4) {
                if (r0 + ib + 3 >= ob && r0 + ib - ob - 3 < a.nk) {
#pragma unroll
                    for (int i = ib; i < ib + 4; ++i) {
                        const float t0 = (float)(cur[i] & 0xffffu), t1 = (float)(cur[i] >> 16);
#pragma unroll
                        for (int o = ob; o < ob + 4; ++o) {
                            const float k = __uint_as_float(__builtin_amdgcn_readfirstlane(kw[32 + i - o]));
                            acc[o][0] = __builtin_fmaf(t0, k, acc[o][0]);
                            acc[o][1] = __builtin_fmaf(t1, k, acc[o][1]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    // the band's blurred values of this lane's two columns: acc < 2^24 (the host checked), so the value is bits 16..23 of the exact integer
#pragma unroll
    for (int o = 0; o < B2_R; ++o) {
        uint32_t lo, hi;
        if constexpr (WIDE) {
            lo = (((uint32_t)(int)fminf(acc[o][0], 16777215.0f - U8F_BIAS) >> 16) & 255u) ^ 0x80u;
            hi = (((uint32_t)(int)fminf(acc[o][1], 16777215.0f - U8F_BIAS) >> 16) & 255u) ^ 0x80u;
        } else {
            lo = (uint32_t)acc[o][0] >> 16;
            hi = (uint32_t)acc[o][1] >> 16;
        }
        bt[o][tid] = (uint16_t)(swapped ? hi | (lo << 8) : lo | (hi << 8)); // left tap | right tap << 8
    }
    // every output row whose two source rows lie in this band (a lane reads back only its own column: no barrier)
    const int first_owned = band * BAND;
    int r = (int)floorf(((float)first_owned + 0.5f) / a.ry - 0.5f) - 2;
    r = __builtin_amdgcn_readfirstlane(max(r, 0));
    for (; r < a.drows; ++r) { // wave-uniform
        const float sy = ((float)r + 0.5f) * a.ry - 0.5f;
        const float ft = floorf(sy);
        const int top = (int)ft;
        int r0 = top, r1 = top + 1;
        if (top < 0 || top + 1 >= a.rows) {
            r0 = resolve_index(top, a.rows, ZG_BORDER_MIRROR);
            r1 = resolve_index(top + 1, a.rows, ZG_BORDER_MIRROR);
        }
        const int owner = min(min(r0, r1) / BAND, a.nbands - 1);
        if (owner < band) continue;
        if (owner > band) break;
        const int fy = (int)roundf((sy - ft) * 256);
        const uint32_t p0 = bt[r0 - y0][tid], p1 = bt[r1 - y0][tid];
        const int tl = (int)(p0 & 255u), tr = (int)(p0 >> 8), bl = (int)(p1 & 255u), br = (int)(p1 >> 8);
        const int top_val = tl * (256 - fx) + tr * fx;
        const int bottom_val = bl * (256 - fx) + br * fx;
        const uint32_t v = (uint32_t)((top_val * (256 - fy) + bottom_val * fy + 32768) >> 16); // < 256: a convex combination of bytes
        if (live) a.dst[(size_t)r * a.dst_pitch + (size_t)c] = (uint8_t)v;
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_cols_bilinear_u8(PyrLevelArgs a, TapsCols taps) {
    __shared__ uint16_t bt[B2_R][256];
    const int band = blockIdx.x / a.tiles_x, tx = blockIdx.x - band * a.tiles_x;
    const int y0 = band == a.nbands - 1 ? max(0, a.rows - B2_R) : band * (B2_R - 1);
    if (y0 - a.half >= 0 && y0 - a.half + B2_R + a.nk - 1 <= a.rows) // workgroup-uniform: every streamed row is inside the plane
        cols_bilinear_band<true, WIDE>(a, taps, bt, tx, band);
    else
        cols_bilinear_band<false, WIDE>(a, taps, bt, tx, band);
}

// gaussianBlur(taps, .mirror) then resize(.bilinear) of an Image(u8) into `level`, without the blurred plane. -1 when the preconditions do not hold
// (the caller blurs and resizes): contiguous-enough u8 planes as k_rows_u8f wants them, a reduction, odd tap counts <= 65, the f32-exact case.
int try_pyramid_level_u8(const zg_image *src, const zg_image *level, const int32_t *taps, int nk, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_U8 || level->pixel != ZG_PIXEL_U8 || nk < 1 || nk > B2_NKMAX || !(nk & 1)) return -1;
    static const bool no_fuse = getenv("ZIGNAL_HIP_NO_PYRAMID_FUSE") != nullptr; // A/B hook of round 5, read once
    if (no_fuse) return -1;
    // A lane here owns the two columns of ONE output column, the dense column pass the two columns of a dword: 2 / scale of its arithmetic. Below
    // a reduction by 2 the dense pass and the separate resize are cheaper (thresholds 1 / 1.7 / 2 / 2.4 / 2.9 measured: profiles/r05_experiments.txt).
    if (src->cols < 2 * level->cols) return -1;
    if (src->cols % 16 || src->stride % 16 || ((uintptr_t)src->data & 15) || src->cols < 256 || (uint64_t)src->cols > 0x3fffffffu) return -1;
    if (level->rows == 0 || level->cols == 0 || level->rows > src->rows || level->cols > src->cols || src->rows < 2) return -1;
    if ((uint64_t)(src->rows + 16) * src->cols * 2 > 0x7fffffffull) return -1;
    int64_t sum = 0;
    for (int i = 0; i < nk; ++i) { if (taps[i] < 0 || taps[i] > 255) return -1; sum += taps[i]; }
    if (sum > 257) return -1; // temp fits u16
    const bool wide = sum * sum * 255 + 32768 >= 256 * 65536; // a tap sum of 257: the biased accumulators of cols_strip_u8f<., true>

    const int row_bytes = (int)src->cols, half = nk / 2;
    const size_t temp_bytes = ((size_t)src->rows + 16) * row_bytes * 2; // + 16 slack rows: the column pass prefetches past a band's last row
    uint32_t *temp = nullptr;
    if (int rc = scratch_alloc((void **)&temp, temp_bytes, s)) return rc;
    const int tiles_x = (int)ceil_div((uint32_t)row_bytes, 1024u);
    const int rows_per_wave = 4; // 1, 2 and 8 measured the same or worse (profiles/r05_experiments.txt)
    const B2Frames fr{0, 0, 0};
    const dim3 grid_rows((unsigned)(tiles_x * ceil_div(src->rows, 4u * rows_per_wave)), 1);
    switch ((half + 3) / 4) {
    case 0: case 1: launch_rows_u8f<4>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    case 2: launch_rows_u8f<8>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    case 3: launch_rows_u8f<12>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    case 4: launch_rows_u8f<16>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    case 5: launch_rows_u8f<20>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    case 6: launch_rows_u8f<24>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    case 7: launch_rows_u8f<28>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    default: launch_rows_u8f<32>(src, temp, taps, nk, ZG_BORDER_MIRROR, tiles_x, rows_per_wave, grid_rows, fr, s); break;
    }
    TapsCols tc{};
    for (int j = 0; j < nk; ++j) {
        const float f = (float)taps[j];
        memcpy(&tc.k[B2_R + j], &f, 4);
    }
    PyrLevelArgs a{};
    a.temp = (const uint16_t *)temp;
    a.dst = (uint8_t *)level->data;
    a.dst_pitch = level->stride;
    a.rows = (int)src->rows; a.cols = (int)src->cols; a.drows = (int)level->rows; a.dcols = (int)level->cols;
    a.rx = (float)src->cols / (float)level->cols;
    a.ry = (float)src->rows / (float)level->rows;
    a.nk = nk; a.half = half;
    a.tiles_x = (int)ceil_div(level->cols, 256u);
    a.nbands = src->rows <= (uint32_t)B2_R ? 1 : (int)ceil_div(src->rows - 1, (uint32_t)(B2_R - 1));
    if (wide) hipLaunchKernelGGL((k_cols_bilinear_u8<true>), dim3((unsigned)(a.tiles_x * a.nbands)), dim3(256), 0, s, a, tc);
    else hipLaunchKernelGGL((k_cols_bilinear_u8<false>), dim3((unsigned)(a.tiles_x * a.nbands)), dim3(256), 0, s, a, tc);
    const hipError_t e = hipGetLastError();
    scratch_free(temp, s);
    ZG_HIP(e);
    return ZG_OK;
}

// ---- the levels of a pyramid in three launches -----------------------------------------------------------------------------------------------------------
// ImagePyramid.build makes every level from the ORIGINAL, so all the row passes read one source: k_rows_u8f_multi runs them as one launch, then the dense
// column passes (levels reduced by less than 2: a blurred plane, resized afterwards) as one and the fused ones (k_cols_bilinear_u8) as one. Per job:
struct DenseJob { const uint32_t *temp; uint8_t *dst; size_t dst_pitch; int nk, half, nk4; TapsCols taps; };
struct DenseJobs { DenseJob j[4]; };
__global__ __launch_bounds__(256) void k_cols_u8f_multi(DenseJobs jobs, int rows, int row_bytes, int tiles_x) {
    const DenseJob &job = jobs.j[blockIdx.y];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * B2_R;
    if (!(y0 - job.half >= 0 && y0 - job.half + B2_R + job.nk4 - 1 <= rows)) { // bands at the top and bottom rows: the general form
        cols_strip_u8f<false, false, B2_R>(job.temp, job.dst, job.dst_pitch, rows, row_bytes, job.taps, job.nk, job.half, ZG_BORDER_MIRROR, tx, ty);
        return;
    }
    switch (job.nk4) { // workgroup-uniform
    case 9: cols_strip_u8f_static<9, false>(job.temp, job.dst, job.dst_pitch, row_bytes, job.taps, job.half, tx, ty); break;
    case 13: cols_strip_u8f_static<13, false>(job.temp, job.dst, job.dst_pitch, row_bytes, job.taps, job.half, tx, ty); break;
    case 17: cols_strip_u8f_static<17, false>(job.temp, job.dst, job.dst_pitch, row_bytes, job.taps, job.half, tx, ty); break;
    default: cols_strip_u8f_static<21, false>(job.temp, job.dst, job.dst_pitch, row_bytes, job.taps, job.half, tx, ty); break; // the host sends no longer kernel
    }
}
struct FusedJob { PyrLevelArgs a; int wide; TapsCols taps; };
struct FusedJobs { FusedJob j[4]; };
__global__ __launch_bounds__(256) void k_cols_bilinear_u8_multi(FusedJobs jobs) {
    __shared__ uint16_t bt[B2_R][256];
    const FusedJob &job = jobs.j[blockIdx.y];
    const PyrLevelArgs &a = job.a;
    if ((int)blockIdx.x >= a.tiles_x * a.nbands) return; // the launch is as wide as its largest level
    const int band = blockIdx.x / a.tiles_x, tx = blockIdx.x - band * a.tiles_x;
    const int y0 = band == a.nbands - 1 ? max(0, a.rows - B2_R) : band * (B2_R - 1);
    const bool inside = y0 - a.half >= 0 && y0 - a.half + B2_R + a.nk - 1 <= a.rows;
    if (job.wide) {
        if (inside) cols_bilinear_band<true, true>(a, job.taps, bt, tx, band);
        else cols_bilinear_band<false, true>(a, job.taps, bt, tx, band);
    } else {
        if (inside) cols_bilinear_band<true, false>(a, job.taps, bt, tx, band);
        else cols_bilinear_band<false, false>(a, job.taps, bt, tx, band);
    }
}

int resize_impl_bilinear_u8(const zg_image *src, const zg_image *dst, hipStream_t s); // edges.hip: zg_resize(.bilinear)

// Levels i with handled[i] set on return were enqueued here (on `s`); the others are the caller's. A caller with several streams asks for the fused levels on
// one and the dense levels on another (two independent batches, each with its own row-pass launch and scratch block: no event between the streams). sigmas[i] <= 0.5 (a plain resize), other pixel types,
// shapes the packed kernels exclude, kernels of <= 7 taps (the one-pass stream kernel is better there) and > 33 are left alone. -1: nothing was done.
int try_pyramid_levels_u8(const zg_image *src, const zg_image *levels, const float *sigmas, uint32_t n, uint8_t *handled, int which, hipStream_t s) {
    static const bool no_batch = getenv("ZIGNAL_HIP_NO_PYRAMID_BATCH") != nullptr; // A/B hook of round 5, read once
    if (no_batch) return -1;
    if (src->pixel != ZG_PIXEL_U8 || src->cols % 16 || src->stride % 16 || ((uintptr_t)src->data & 15) || src->cols < 256 || (uint64_t)src->cols > 0x3fffffffu || src->rows < 2) return -1;
    struct Plan { uint32_t level; int nk, half, hp, wide; bool fused; int32_t taps[33]; };
    Plan plan[PYR_MAX_JOBS];
    int np = 0, n_dense = 0, n_fused = 0;
    for (uint32_t i = 0; i < n && np < PYR_MAX_JOBS; ++i) {
        const zg_image &lv = levels[i];
        if (!(sigmas[i] > 0.5f) || lv.pixel != ZG_PIXEL_U8 || lv.rows == 0 || lv.cols == 0 || lv.rows > src->rows || lv.cols > src->cols) continue;
        const int nfull = zg_gaussian_kernel(sigmas[i], nullptr, 0);
        if (nfull < 1 || nfull > 65) continue;
        float ft[65];
        if (zg_gaussian_kernel(sigmas[i], ft, 65) != nfull) continue;
        int32_t it[65];
        int64_t sum = 0;
        bool ok = true;
        for (int j = 0; j < nfull; ++j) { it[j] = (int32_t)std::round(ft[j] * 256.0f); ok = ok && it[j] >= 0 && it[j] <= 255; sum += it[j]; }
        if (!ok || sum > 257) continue;
        int z = 0;
        while (nfull - 2 * z > 2 && it[z] == 0 && it[nfull - 1 - z] == 0) ++z;
        const int nk = nfull - 2 * z;
        if (nk <= 7 || nk > 33 || !(nk & 1)) continue;
        Plan &p = plan[np];
        p.level = i; p.nk = nk; p.half = nk / 2; p.hp = std::max(4, (p.half + 3) / 4 * 4);
        p.wide = sum * sum * 255 + 32768 >= 256 * 65536;
        p.fused = src->cols >= 2 * lv.cols;
        if (handled[i] || (which == 0 && !p.fused) || (which == 1 && p.fused)) continue; // which: 0 = the fused levels, 1 = the dense ones, 2 = both
        if (!p.fused && (p.wide || nk > 21)) continue;      // the batched dense column pass: sums <= 256, <= 21 taps (a reduction below 2 has sigma < 2.8 x blur_sigma;
                                                            // longer instantiations would cost every job of the launch its occupancy)
        if (p.fused ? n_fused == 4 : n_dense == 4) continue;
        for (int j = 0; j < nk; ++j) p.taps[j] = it[z + j];
        (p.fused ? n_fused : n_dense)++;
        ++np;
    }
    if (np < 2) return -1; // one level gains nothing from a batch (the caller's level-by-level path takes it)

    const int row_bytes = (int)src->cols;
    const size_t temp_bytes = ((size_t)src->rows + 16) * row_bytes * 2, plane_bytes = (size_t)src->rows * src->cols;
    uint8_t *block = nullptr;
    if (int rc = scratch_alloc((void **)&block, temp_bytes * np + plane_bytes * n_dense, s)) return rc;
    RowsJobs rj{};
    DenseJobs dj{};
    FusedJobs fj{};
    uint8_t *blurred[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t dense_level[4] = {0, 0, 0, 0};
    int di = 0, fi = 0, fused_grid = 0;
    for (int q = 0; q < np; ++q) {
        const Plan &p = plan[q];
        uint32_t *temp = (uint32_t *)(block + temp_bytes * q);
        rj.j[q].temp = temp;
        rj.j[q].hp = p.hp;
        for (int j = 0; j < p.nk; ++j) rj.j[q].k[j - p.half + p.hp] = (float)p.taps[j];
        TapsCols tc{};
        for (int j = 0; j < p.nk; ++j) { const float f = (float)p.taps[j]; memcpy(&tc.k[B2_R + j], &f, 4); }
        const zg_image &lv = levels[p.level];
        if (p.fused) {
            FusedJob &f = fj.j[fi++];
            f.a.temp = (const uint16_t *)temp; f.a.dst = (uint8_t *)lv.data; f.a.dst_pitch = lv.stride;
            f.a.rows = (int)src->rows; f.a.cols = (int)src->cols; f.a.drows = (int)lv.rows; f.a.dcols = (int)lv.cols;
            f.a.rx = (float)src->cols / (float)lv.cols; f.a.ry = (float)src->rows / (float)lv.rows;
            f.a.nk = p.nk; f.a.half = p.half;
            f.a.tiles_x = (int)ceil_div(lv.cols, 256u);
            f.a.nbands = src->rows <= (uint32_t)B2_R ? 1 : (int)ceil_div(src->rows - 1, (uint32_t)(B2_R - 1));
            f.wide = p.wide; f.taps = tc;
            fused_grid = std::max(fused_grid, f.a.tiles_x * f.a.nbands);
        } else {
            blurred[di] = block + temp_bytes * np + plane_bytes * di;
            dense_level[di] = p.level;
            DenseJob &d = dj.j[di++];
            d.temp = temp; d.dst = blurred[di - 1]; d.dst_pitch = src->cols; d.nk = p.nk; d.half = p.half; d.nk4 = std::max(9, (p.nk + 2) / 4 * 4 + 1); d.taps = tc;
        }
        handled[p.level] = 1;
    }
    const int tiles_x = (int)ceil_div((uint32_t)row_bytes, 1024u), rows_per_wave = 4;
    hipLaunchKernelGGL(k_rows_u8f_multi, dim3((unsigned)(tiles_x * ceil_div(src->rows, 4u * rows_per_wave)), (unsigned)np), dim3(256), 0, s, dimg(src), rj, (int)ZG_BORDER_MIRROR, tiles_x,
                       rows_per_wave);
    int rc = ZG_OK;
    if (fi) hipLaunchKernelGGL(k_cols_bilinear_u8_multi, dim3((unsigned)fused_grid, (unsigned)fi), dim3(256), 0, s, fj);
    if (di) {
        const int tiles_x2 = (int)ceil_div((uint32_t)row_bytes, 512u);
        hipLaunchKernelGGL(k_cols_u8f_multi, dim3((unsigned)(tiles_x2 * ceil_div(src->rows, (uint32_t)B2_R)), (unsigned)di), dim3(256), 0, s, dj, (int)src->rows, row_bytes, tiles_x2);
        for (int d = 0; d < di && rc == ZG_OK; ++d) {
            const zg_image tmp{blurred[d], src->cols, src->rows, src->cols, ZG_PIXEL_U8};
            rc = resize_impl_bilinear_u8(&tmp, &levels[dense_level[d]], s);
        }
    }
    const hipError_t e = hipGetLastError();
    scratch_free(block, s);
    ZG_HIP(e);
    return rc;
}

// Returns -1 when the preconditions do not hold (caller falls back to the general kernels).
int try_sep_bytes2(const zg_image *src, const zg_image *dst, const int32_t *ix, int nkx, const int32_t *iy, int nky, int border, hipStream_t s) {
    return try_sep_bytes2_frames(src, dst, 1, 0, 0, ix, nkx, iy, nky, border, s);
}

// n frames of src's geometry, src_frame / dst_frame BYTES apart: the two passes run once per chunk of frames whose temp planes fit the
// scratch budget, frame = blockIdx.y.
int try_sep_bytes2_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, const int32_t *ix, int nkx,
                          const int32_t *iy, int nky, int border, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_U8 && src->pixel != ZG_PIXEL_RGB_U8 && src->pixel != ZG_PIXEL_RGBA_U8) return -1;
    if (nkx < 1 || nky < 1 || nkx > B2_NKMAX || nky > B2_NKMAX) return -1;
    const size_t sp = pixel_size(src->pixel);
    if ((src->cols * sp) % 16 || (src->stride * sp) % 16 || (dst->stride * sp) % 16 || ((uintptr_t)src->data & 15) || ((uintptr_t)dst->data & 15)) return -1;
    if (src->cols * sp < 256 || (uint64_t)src->cols * sp > 0x3fffffffu) return -1;
    int64_t sx = 0, sy = 0;
    for (int i = 0; i < nkx; ++i) { if (ix[i] < 0 || ix[i] > 255) return -1; sx += ix[i]; }
    for (int i = 0; i < nky; ++i) { if (iy[i] < 0 || iy[i] > 255) return -1; sy += iy[i]; }
    if (sx > 257 || sy > 257) return -1; // temp must fit u16: 255 * 257 = 65535
    const bool over = sx * sy * 255 + 32768 >= 256 * 65536; // only then can (acc >> 16) exceed 255
    const bool wide = over && sx * sy * 255 + 32768 - (int64_t)U8F_BIAS <= 16777216; // ... and the biased f32 form still holds every sum (always: sums <= 257)
    const bool clamp = over && !wide; // the integer column pass

    // row pass taps: tap j reads offset j - nkx / 2; the padded window starts at offset -hpad
    const int halfx = nkx / 2, halfy = nky / 2;
    // (4-byte pixels keep every group dword-aligned whatever hpad is; 1- and 3-byte pixels need hpad % 4 == 0)
    const int hpad = sp == 4 ? halfx : std::max(4, (halfx + 3) / 4 * 4);
    const int ngroups = (2 * hpad + 4) / 4; // taps at offsets -hpad .. +hpad, four per group
    TapsRows tr{};
    for (int j = 0; j < nkx; ++j) tr.kk[j - halfx + hpad] = (uint32_t)ix[j] | ((uint32_t)ix[j] << 16);
    TapsCols tc{};
    for (int j = 0; j < nky; ++j) { // k_cols_u8f (!clamp) reads its taps as floats; a zero is a zero either way
        const float f = (float)iy[j];
        uint32_t bits;
        memcpy(&bits, &f, 4);
        tc.k[B2_R + j] = clamp ? (uint32_t)iy[j] : bits;
    }

    const int row_bytes = (int)(src->cols * sp);
    // + 16 slack rows: the column pass prefetches up to 15 rows past a strip's last one (their taps are zero)
    const size_t temp_bytes = ((size_t)src->rows + 16) * row_bytes * 2;
    if (n > 1 && ((src_frame | dst_frame) & 15)) return -1;
    const uint32_t per_launch = (uint32_t)std::min<size_t>(std::min(n, MAX_FRAMES_PER_LAUNCH), std::max<size_t>(1, scratch_block_budget() / 2 / temp_bytes));
    uint32_t *temp = nullptr;
    if (int rc = scratch_alloc((void **)&temp, temp_bytes * per_launch, s)) return rc;
    const int tiles_x = (int)ceil_div((uint32_t)row_bytes, 1024u);
    const int rows_per_wave = 4; // 1, 2 and 8 measured the same or worse (profiles/r05_experiments.txt)
    for (uint32_t f0 = 0; f0 < n; f0 += per_launch) {
        const uint32_t nf = std::min(per_launch, n - f0);
        zg_image a = *src;
        a.data = (uint8_t *)src->data + (size_t)f0 * src_frame;
        uint8_t *out = (uint8_t *)dst->data + (size_t)f0 * dst_frame;
        const B2Frames fr{src_frame, temp_bytes / 4, dst_frame};
        const dim3 grid_rows((unsigned)(tiles_x * ceil_div(src->rows, 4u * rows_per_wave)), nf);
        if (sp == 1) { // grey planes: the f32 row pass (the packed-u16 one lost by a third: profiles/r03_experiments.txt)
            switch ((halfx + 3) / 4) {
            case 0: case 1: launch_rows_u8f<4>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            case 2: launch_rows_u8f<8>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            case 3: launch_rows_u8f<12>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            case 4: launch_rows_u8f<16>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            case 5: launch_rows_u8f<20>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            case 6: launch_rows_u8f<24>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            case 7: launch_rows_u8f<28>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            default: launch_rows_u8f<32>(&a, temp, ix, nkx, border, tiles_x, rows_per_wave, grid_rows, fr, s); break;
            }
        } else if (sp == 3) hipLaunchKernelGGL((k_rows_u16<3>), grid_rows, dim3(256), 0, s, dimg(&a), temp, tr, hpad, ngroups, border, tiles_x, rows_per_wave, fr);
        else hipLaunchKernelGGL((k_rows_u16<4>), grid_rows, dim3(256), 0, s, dimg(&a), temp, tr, hpad, ngroups, border, tiles_x, rows_per_wave, fr);
        if (clamp) {
            const dim3 grid_cols((unsigned)(tiles_x * ceil_div(src->rows, (uint32_t)B2_R)), nf);
            hipLaunchKernelGGL((k_cols_u16<true>), grid_cols, dim3(256), 0, s, (const uint32_t *)temp, out, dst->stride * sp, (int)src->rows, row_bytes, tc, nky, halfy, border, tiles_x, fr);
        } else {
            const int tiles_x2 = (int)ceil_div((uint32_t)row_bytes, 512u);
            static const bool generic = getenv("ZIGNAL_HIP_B2_GENERIC_COLS") != nullptr; // A/B hook of round 5
            if (!generic && nky <= 33) {
                const dim3 grid2((unsigned)(tiles_x2 * ceil_div(src->rows, (uint32_t)B2_R)), nf);
                auto st = [&](auto nk_tag) {
                    constexpr int NK4 = decltype(nk_tag)::value;
                    if (wide) hipLaunchKernelGGL((k_cols_u8f_static<NK4, true>), grid2, dim3(256), 0, s, (const uint32_t *)temp, out, dst->stride * sp, (int)src->rows, row_bytes, tc, nky, halfy, border, tiles_x2, fr);
                    else hipLaunchKernelGGL((k_cols_u8f_static<NK4, false>), grid2, dim3(256), 0, s, (const uint32_t *)temp, out, dst->stride * sp, (int)src->rows, row_bytes, tc, nky, halfy, border, tiles_x2, fr);
                };
                switch ((nky + 2) / 4) { // 4 m + 1 >= nky
                case 0: case 1: case 2: st(std::integral_constant<int, 9>{}); break;
                case 3: st(std::integral_constant<int, 13>{}); break;
                case 4: st(std::integral_constant<int, 17>{}); break;
                case 5: st(std::integral_constant<int, 21>{}); break;
                case 6: st(std::integral_constant<int, 25>{}); break;
                case 7: st(std::integral_constant<int, 29>{}); break;
                default: st(std::integral_constant<int, 33>{}); break;
                }
                continue;
            }
            const dim3 grid2((unsigned)(tiles_x2 * ceil_div(src->rows, (uint32_t)B2_R)), nf);
            if (wide) hipLaunchKernelGGL((k_cols_u8f<true, B2_R>), grid2, dim3(256), 0, s, (const uint32_t *)temp, out, dst->stride * sp, (int)src->rows, row_bytes, tc, nky, halfy, border, tiles_x2, fr);
            else hipLaunchKernelGGL((k_cols_u8f<false, B2_R>), grid2, dim3(256), 0, s, (const uint32_t *)temp, out, dst->stride * sp, (int)src->rows, row_bytes, tc, nky, halfy, border, tiles_x2, fr);
        }
    }
    const hipError_t e = hipGetLastError();
    scratch_free(temp, s);
    ZG_HIP(e);
    return ZG_OK;
}

} // namespace zg
