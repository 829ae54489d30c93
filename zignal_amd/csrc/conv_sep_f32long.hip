// conv_sep_f32long.hip — Image(f32 / Rgb(f32) / Rgba(f32)).convolveSeparable / gaussianBlur for LONG kernels (the 11..65-tap
// Gaussians of ImagePyramid levels and of Canny's own blur), as two coalesced passes through an f32 temp plane.
//
// Same arithmetic contract as conv_separable.hip (reference src/image/convolution.zig:441-647, f32 path): per element
// temp = sum_i src[c + i - h] * kx[i], out = sum_i temp[r + i - h] * ky[i], ascending i from an accumulator of 0, separate
// multiply and add, out-of-range taps through border.resolveIndex (a dropped tap adds 0 * k). Interior pixels would skip
// taps with |k| < 1e-10; this path is only taken when no tap is that small, so the question does not arise.
// All channels share the taps, so — as in conv_sep_bytes2.hip — a row of SP-float pixels is a plain float stream in which
// tap i of element e is element e + SP * (i - h):
//   k_rows_f32<SP>  a wave stages one row segment (256 elements + the taps' reach) in its own LDS buffer and walks the
//                   taps in a run-time loop; lanes own elements l, l + 64, l + 128, l + 192 so the element-granular LDS
//                   reads hit consecutive banks. One kernel per pixel stride, any tap count.
//   k_cols_f32      a lane owns four adjacent elements of 16 output rows as accumulators and streams the 16 + n - 1 temp
//                   rows past them once (eight rows per step, the next eight in flight, taps in SGPRs). Each
//                   accumulator receives its taps in ascending order, as the reference's inner loop does.
// Preconditions (else the general kernels run): f32 pixel types, row length and strides multiples of 4 elements, 16-byte
// aligned bases, row >= 256 elements, no negligible tap, both tap counts <= 65.
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

constexpr int FL_HMAX = 32;
constexpr int FL_NKMAX = 65;
constexpr int FL_R = 16;
constexpr int FL_LEFT = FL_HMAX * 4;              // elements of reach on each side (half width x widest pixel)
constexpr int FL_ROW = FL_LEFT + 256 + FL_LEFT;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TapsRowsF { float k[FL_NKMAX]; };
struct TapsColsF { float k[FL_NKMAX + 2 * FL_R + 8]; }; // k[FL_R + j] = tap j, zeros around

// GREY (SP == 1 only): the source is not the f32 plane itself but the image the detectors take their grey from — 1: Image(u8), as(f32, u8);
// 4: Image(Rgba(u8)), BT.709 in 16.16 fixed point (color.zig:1031-1042, edges.zig:231-240) — converted as it is staged, so canny's grey plane is
// never written or read back (k_canny_gray4's 22 us per 4096^2 frame).
template <int GREY>
__device__ __forceinline__ float grey_of_rgba(uint32_t px) {
    const int y = (13933 * (int)(px & 255u) + 46871 * (int)((px >> 8) & 255u) + 4732 * (int)((px >> 16) & 255u) + 32768) >> 16;
    return (float)(y < 0 ? 0 : (y > 255 ? 255 : y));
}
template <int GREY>
__device__ __forceinline__ f32x4 row_load4(const void *row, int e) { // elements e .. e + 3 of the row's f32 stream
    if constexpr (GREY == 0) return *(const f32x4 *)((const float *)row + e);
    else if constexpr (GREY == 1) {
        const uint32_t v = *(const uint32_t *)((const uint8_t *)row + e);
        return f32x4{(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24)};
    } else {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = *(const u32x4 *)((const uint8_t *)row + 4 * (size_t)e);
        return f32x4{grey_of_rgba<GREY>(q[0]), grey_of_rgba<GREY>(q[1]), grey_of_rgba<GREY>(q[2]), grey_of_rgba<GREY>(q[3])};
    }
}
template <int GREY>
__device__ __forceinline__ float row_load1(const void *row, int e) {
    if constexpr (GREY == 0) return ((const float *)row)[e];
    else if constexpr (GREY == 1) return (float)((const uint8_t *)row)[e];
    else return grey_of_rgba<GREY>(((const uint32_t *)row)[e]);
}
template <int SP, int GREY = 0>
__global__ __launch_bounds__(256) void k_rows_f32(DImg src, float *temp, TapsRowsF taps, int nk, int half, int border, int tiles_x, int rows_per_wave) {
    static_assert(GREY == 0 || SP == 1, "a grey source is one plane");
    __shared__ float lds[4][FL_ROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int xf0 = tx * 256;
    const int row_f = src.cols * SP;
    float *buf = lds[wave];
    const int reach = half * SP;
    const bool edge = xf0 - reach < 0 || xf0 + 256 + reach > row_f; // this tile's taps reach past a row end (workgroup-uniform)

    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int y = (ty * 4 + wave) * rows_per_wave + rr; // wave-uniform
        if (y >= src.rows) break;
        const void *row = (const uint8_t *)src.data + (size_t)y * src.stride * (GREY == 0 ? SP * sizeof(float) : (size_t)GREY);
        {   // 64 main units of four elements and 64 halo units (32 left, 32 right); units are all inside or all outside the
            // row (length % 4 == 0); outside ones are zeroed here and patched below. Unpredicated loads from clamped addresses.
            const int gm = xf0 + 4 * lane;
            f32x4 v = row_load4<GREY>(row, min(gm, row_f - 4));
            if (gm + 4 > row_f) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const int gh = lane < 32 ? xf0 - FL_LEFT + 4 * lane : xf0 + 256 + 4 * (lane - 32);
            f32x4 h = row_load4<GREY>(row, min(max(gh, 0), row_f - 4));
            if (gh < 0 || gh + 4 > row_f) h = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *(f32x4 *)(buf + FL_LEFT + 4 * lane) = v;
            *(f32x4 *)(buf + (lane < 32 ? 4 * lane : FL_LEFT + 256 + 4 * (lane - 32))) = h;
        }
        if (edge) { // border rule for the columns, one element per lane
            for (int k = lane; k < 2 * reach; k += 64) {
                const int e = k < reach ? -1 - k : row_f + (k - reach); // element position in the row's stream
                const int t = e - (xf0 - FL_LEFT);                       // position in the LDS row
                if (t < 0 || t >= FL_ROW) continue;
                const int px = e >= 0 ? e / SP : -((SP - 1 - e) / SP);   // floor
                const int gc = resolve_index(px, src.cols, border);
                if (gc < 0) continue; // zero border: already 0
                buf[t] = row_load1<GREY>(row, gc * SP + (e - px * SP));
            }
        }
        __builtin_amdgcn_wave_barrier(); // LDS is in order within a wave; this only stops the compiler from reordering

        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
        const float *p = buf + FL_LEFT + lane - reach;
        for (int i = 0; i < nk; ++i) {
            const float k = taps.k[i];
            const float v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];
            const float m0 = v0 * k, m1 = v1 * k, m2 = v2 * k, m3 = v3 * k;
            acc0 = acc0 + m0; acc1 = acc1 + m1; acc2 = acc2 + m2; acc3 = acc3 + m3;
            p += SP;
        }
        __builtin_amdgcn_wave_barrier(); // the next row's staging must not overtake these reads

        float *trow = temp + (size_t)y * row_f + xf0 + lane;
        if (xf0 + lane < row_f) trow[0] = acc0;
        if (xf0 + lane + 64 < row_f) trow[64] = acc1;
        if (xf0 + lane + 128 < row_f) trow[128] = acc2;
        if (xf0 + lane + 192 < row_f) trow[192] = acc3;
    }
}

template <bool INSIDE>
__device__ __forceinline__ void cols_strip_f32(const float *temp, float *dst, size_t dst_pitch_f, int rows, int row_f, const TapsColsF &taps, int nk,
                                               int half, int border, int tx, int ty) {
    const int xu = tx * 256 + (int)threadIdx.x; // this lane's unit of four elements
    const bool live = xu * 4 < row_f;
    const int xuc = live ? xu : 0;
    const int y0 = ty * FL_R;
    const size_t trow = (size_t)row_f;
    f32x4 acc[FL_R];
#pragma unroll
    for (int o = 0; o < FL_R; ++o) acc[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int nrows_in = FL_R + nk - 1;
    const float *tcol = temp + 4 * (size_t)xuc;
    const float *next_row = tcol + (size_t)(INSIDE ? y0 - half : 0) * trow; // INSIDE: a running pointer, one add per row
    auto fetch = [&](int r) -> f32x4 { // temp row y0 - half + r; called with r = 0, 1, 2, ... in order
        if constexpr (INSIDE) { // rows past the strip's last (prefetch overshoot) fall in the temp plane's slack rows
            const f32x4 p = *(const f32x4 *)next_row;
            next_row += trow;
            return p;
        } else {
            const int gr = resolve_index(y0 - half + min(r, nrows_in - 1), rows, border); // scalar
            f32x4 p = *(const f32x4 *)(tcol + (size_t)max(gr, 0) * trow);
            if (gr < 0) p = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            return p;
        }
    };
    f32x4 cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = fetch(i);
    for (int r0 = 0; r0 < nrows_in; r0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = fetch(r0 + 8 + i);
        // output row o takes k[r - o] from temp row r = r0 + i: entry 16 + i - o of the 24-entry slice of the zero-padded
        // tap table that starts at r0. A row is only applied to the outputs whose tap lies inside the kernel (a tap outside
        // it is not "times zero" in the reference, it does not exist), tested once per 4 rows x 4 outputs and per element below.
        float kw[24];
#pragma unroll
        for (int c = 0; c < 24; ++c) kw[c] = taps.k[r0 + c];
#pragma unroll
        for (int ib = 0; ib < 8; ib += 4) {
#pragma unroll
            for (int ob = 0; ob < FL_R; ob += 4) {
                const int jlo = r0 + ib - ob - 3, jhi = r0 + ib - ob + 3; // range of tap indices r - o inside this block
                if (jlo >= 0 && jhi < nk) { // the whole block lies inside the kernel: no per-element test
#pragma unroll
                    for (int i = ib; i < ib + 4; ++i) {
#pragma unroll
                        for (int o = ob; o < ob + 4; ++o) {
                            const f32x4 m = cur[i] * kw[FL_R + i - o];
                            acc[o] = acc[o] + m;
                        }
                    }
                } else if (jhi >= 0 && jlo < nk) { // straddles an end of the kernel
#pragma unroll
                    for (int i = ib; i < ib + 4; ++i) {
#pragma unroll
                        for (int o = ob; o < ob + 4; ++o) {
                            const int j = r0 + i - o; // wave-uniform
                            if (j >= 0 && j < nk) {
                                const f32x4 m = cur[i] * kw[FL_R + i - o];
                                acc[o] = acc[o] + m;
                            }
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
    for (int o = 0; o < FL_R; ++o) {
        const int y = y0 + o;
        if (y >= rows) break;
        *(f32x4 *)(dst + (size_t)y * dst_pitch_f + 4 * (size_t)xu) = acc[o];
    }
}

__global__ __launch_bounds__(256) void k_cols_f32(const float *temp, float *dst, size_t dst_pitch_f, int rows, int row_f, TapsColsF taps, int nk, int half,
                                                  int border, int tiles_x) {
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * FL_R;
    if (y0 - half >= 0 && y0 - half + FL_R + nk - 1 <= rows) // workgroup-uniform: every streamed row is inside the image
        cols_strip_f32<true>(temp, dst, dst_pitch_f, rows, row_f, taps, nk, half, border, tx, ty);
    else
        cols_strip_f32<false>(temp, dst, dst_pitch_f, rows, row_f, taps, nk, half, border, tx, ty);
}

// grey: 0 = src is the f32 image itself (sp channels); 1 / 4 = src is Image(u8) / Image(Rgba(u8)) and the plane convolved is its grey (sp = 1).
static int run_sep_f32long(const zg_image *src, int grey, size_t sp, const zg_image *dst, const float *fx, int nkx, const float *fy, int nky, int border, hipStream_t s) {
    if (nkx < 1 || nky < 1 || nkx > FL_NKMAX || nky > FL_NKMAX) return -1;
    const size_t src_bytes = grey ? (size_t)grey : sp * sizeof(float);
    if ((src->cols * sp) % 4 || (src->stride * src_bytes) % 16 || (dst->stride * sp) % 4 || ((uintptr_t)src->data & 15) || ((uintptr_t)dst->data & 15)) return -1;
    if (src->cols * sp < 256 || (uint64_t)src->cols * sp > 0x1fffffffu) return -1;
    for (int i = 0; i < nkx; ++i) if (!(std::fabs(fx[i]) >= 1e-10f)) return -1; // negligible (or NaN) taps: the general kernels know the skip rule
    for (int i = 0; i < nky; ++i) if (!(std::fabs(fy[i]) >= 1e-10f)) return -1;
    const int halfx = nkx / 2, halfy = nky / 2;
    if (halfx > FL_HMAX) return -1;
    TapsRowsF tr{};
    for (int j = 0; j < nkx; ++j) tr.k[j] = fx[j];
    TapsColsF tc{};
    for (int j = 0; j < nky; ++j) tc.k[FL_R + j] = fy[j];

    const int row_f = (int)(src->cols * sp);
    float *temp = nullptr;
    // + 16 slack rows: the column pass prefetches up to 15 rows past a strip's last one (never applied)
    if (int rc = scratch_alloc((void **)&temp, ((size_t)src->rows + 16) * row_f * sizeof(float), s)) return rc;
    const int tiles_rx = (int)ceil_div((uint32_t)row_f, 256u);
    const int rows_per_wave = 4;
    const dim3 grid_rows((unsigned)(tiles_rx * ceil_div(src->rows, 4u * rows_per_wave)));
    if (grey == 1) hipLaunchKernelGGL((k_rows_f32<1, 1>), grid_rows, dim3(256), 0, s, dimg(src), temp, tr, nkx, halfx, border, tiles_rx, rows_per_wave);
    else if (grey == 4) hipLaunchKernelGGL((k_rows_f32<1, 4>), grid_rows, dim3(256), 0, s, dimg(src), temp, tr, nkx, halfx, border, tiles_rx, rows_per_wave);
    else if (sp == 1) hipLaunchKernelGGL((k_rows_f32<1>), grid_rows, dim3(256), 0, s, dimg(src), temp, tr, nkx, halfx, border, tiles_rx, rows_per_wave);
    else if (sp == 3) hipLaunchKernelGGL((k_rows_f32<3>), grid_rows, dim3(256), 0, s, dimg(src), temp, tr, nkx, halfx, border, tiles_rx, rows_per_wave);
    else hipLaunchKernelGGL((k_rows_f32<4>), grid_rows, dim3(256), 0, s, dimg(src), temp, tr, nkx, halfx, border, tiles_rx, rows_per_wave);
    const int tiles_cx = (int)ceil_div((uint32_t)row_f, 1024u);
    hipLaunchKernelGGL(k_cols_f32, dim3((unsigned)(tiles_cx * ceil_div(src->rows, (uint32_t)FL_R))), dim3(256), 0, s, (const float *)temp, (float *)dst->data,
                       dst->stride * sp, (int)src->rows, row_f, tc, nky, halfy, border, tiles_cx);
    const hipError_t e = hipGetLastError();
    scratch_free(temp, s);
    ZG_HIP(e);
    return ZG_OK;
}

// Returns -1 when the preconditions do not hold (caller falls back to the general kernels).
int try_sep_f32long(const zg_image *src, const zg_image *dst, const float *fx, int nkx, const float *fy, int nky, int border, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_F32 && src->pixel != ZG_PIXEL_RGB_F32 && src->pixel != ZG_PIXEL_RGBA_F32) return -1;
    return run_sep_f32long(src, 0, pixel_channels(src->pixel), dst, fx, nkx, fy, nky, border, s);
}

// The detectors' "grey of src as f32, convolved" in one go: src is Image(u8) or Image(Rgba(u8)), dst an Image(f32) plane of the same size; the
// row pass converts as it stages. -1 when the form does not apply (the caller converts first).
int try_sep_f32long_grey(const zg_image *src, const zg_image *dst, const float *fx, int nkx, const float *fy, int nky, int border, hipStream_t s) {
    if ((src->pixel != ZG_PIXEL_U8 && src->pixel != ZG_PIXEL_RGBA_U8) || dst->pixel != ZG_PIXEL_F32) return -1;
    return run_sep_f32long(src, src->pixel == ZG_PIXEL_U8 ? 1 : 4, 1, dst, fx, nkx, fy, nky, border, s);
}

} // namespace zg
