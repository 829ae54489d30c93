// conv2d.hip — Image(T).convolve: dense 2-D convolution with a small kernel.
//
// Replaces reference src/image/convolution.zig:76-301: dst = sum_ky sum_kx src[r+ky-hh, c+kx-hw] * k[ky][kx],
// ky-major, every tap (no skipping); f32: separate mul and add; u8: k = @round(k * 256) as i32, i64-exact
// accumulate, divClampU8(256). Out-of-range taps go through border.resolveIndex (the reference's interior fast
// path computes the same sums). Struct pixels run per channel on interleaved data (the reference's split /
// plane / merge, :213-293, gives the same bytes; its uniform-channel shortcut is value-preserving).
#include "zg_common.h"

#include <cmath>
#include <cstdlib>
#include <vector>

#pragma clang fp contract(off)

namespace zg {

int try_conv2d_stream(const zg_image *src, const zg_image *dst, const float *taps, int kh, int kw, int border, hipStream_t s); // conv2d_stream.hip

constexpr int MAX_K2D = 15 * 15;

struct Kernel2D {
    int kh, kw;
    union { float f[MAX_K2D]; int32_t i[MAX_K2D]; };
};

// MODE: 0 = f32, 1 = u8 with i32 accumulate (host proved it exact), 2 = u8 with i64 accumulate,
//       3 = u8 with f32 accumulate: 255 * sum|k| < 2^24, so every partial sum is an integer f32 holds exactly — the tile is converted to f32
//           once per staged pixel and each tap is one v_fmac_f32 (half the issue time of an integer multiply-add, no byte extraction:
//           tools/exp/valu_rate.hip); taps travel as floats
//
// One workgroup per 64 x 16 output tile. The (16 + kh - 1) x (64 + kw - 1) source tile is staged in LDS once — the
// border rule is evaluated per staged pixel (and not at all for tiles whose halo lies inside the image), with
// unpredicated loads from a clamped address — and every lane then produces four vertically adjacent outputs, reading
// each tile pixel of its columns once for all four (a tile row j feeds output o with kernel row j - o). For a given
// output the taps are still accumulated ky-major, kx ascending, so the f32 sums are the reference's bit for bit.
// KH / KW > 0: compile-time kernel size (3x3, 5x5, 7x7: fully unrolled, weights in SGPRs); 0: run-time size up to 15x15.
constexpr int C2_TW = 64, C2_TH = 16, C2_RPT = 4;

template <int PIX, int MODE, int KH, int KW>
__global__ __launch_bounds__(256) void k_conv2d(DImg src, DImg dst, Kernel2D k, int border, int tiles_x) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    constexpr int MAXK = KH > 0 ? KH : 15, MAXKW = KW > 0 ? KW : 15;
    constexpr int LW = C2_TW + MAXKW - 1, LH = C2_TH + MAXK - 1;
    using TVec = typename std::conditional<MODE == 3, typename VecOf<float, C>::type, Vec>::type; // what the tile holds
    __shared__ TVec tile[LH * LW];
    const int kh = KH > 0 ? KH : k.kh, kw = KW > 0 ? KW : k.kw;
    const int hh = kh / 2, hw = kw / 2;
    const int lw = C2_TW + kw - 1, lh = C2_TH + kh - 1;

    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int x0 = tx * C2_TW, y0 = ty * C2_TH;

    const bool inside = x0 - hw >= 0 && x0 - hw + lw <= src.cols && y0 - hh >= 0 && y0 - hh + lh <= src.rows; // workgroup-uniform
    for (int i = threadIdx.x; i < lh * lw; i += 256) {
        const int tr = i / lw, tc = i - tr * lw;
        int gr = y0 - hh + tr, gc = x0 - hw + tc;
        bool ok = true;
        if (!inside) {
            gr = resolve_index(gr, src.rows, border);
            gc = resolve_index(gc, src.cols, border);
            ok = gr >= 0 && gc >= 0;
            gr = max(gr, 0);
            gc = max(gc, 0);
        }
        Vec v = P::load(src.data, (size_t)gr * src.stride + (size_t)gc);
        if (!ok) v = P::zero();
        if constexpr (MODE == 3) {
            TVec f;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) f[ch] = (float)v[ch];
            tile[tr * LW + tc] = f;
        } else {
            tile[tr * LW + tc] = v;
        }
    }
    __syncthreads();

    const int lx = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = x0 + lx, r0 = y0 + wave * C2_RPT;
    using Acc = typename std::conditional<MODE == 0 || MODE == 3, float, typename std::conditional<MODE == 1, int32_t, int64_t>::type>::type;
    Acc acc[C2_RPT][C];
#pragma unroll
    for (int o = 0; o < C2_RPT; ++o)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) acc[o][ch] = 0;

    auto tap = [&](int o, int ky, int kx, const TVec &v) {
        if constexpr (MODE == 0) {
            const float w = k.f[ky * kw + kx];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) { const float p = v[ch] * w; acc[o][ch] = acc[o][ch] + p; }
        } else if constexpr (MODE == 3) {
            const float w = k.f[ky * kw + kx];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[o][ch] = __builtin_fmaf(v[ch], w, acc[o][ch]); // exact either way: integers below 2^24
        } else {
            const int32_t w = k.i[ky * kw + kx];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[o][ch] += (Acc)v[ch] * (Acc)w;
        }
    };
    if constexpr (KH > 0) {
#pragma unroll
        for (int j = 0; j < C2_RPT + KH - 1; ++j)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const TVec v = tile[(wave * C2_RPT + j) * LW + lx + kx];
#pragma unroll
                for (int o = 0; o < C2_RPT; ++o)
                    if (j - o >= 0 && j - o < KH) tap(o, j - o, kx, v);
            }
    } else {
        for (int j = 0; j < C2_RPT + kh - 1; ++j)
            for (int kx = 0; kx < kw; ++kx) {
                const TVec v = tile[(wave * C2_RPT + j) * LW + lx + kx];
#pragma unroll
                for (int o = 0; o < C2_RPT; ++o)
                    if (j - o >= 0 && j - o < kh) tap(o, j - o, kx, v); // wave-uniform
            }
    }
    if (c >= dst.cols) return;
#pragma unroll
    for (int o = 0; o < C2_RPT; ++o) {
        const int r = r0 + o;
        if (r >= dst.rows) break;
        Vec out;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            if constexpr (MODE == 0) out[ch] = acc[o][ch];
            else { // divClampU8(256): symmetric rounding divide, clamp
                using IAcc = typename std::conditional<MODE == 3, int32_t, Acc>::type;
                const IAcc a = (IAcc)acc[o][ch]; // MODE 3: an integer-valued float, exact
                if (a < 0) out[ch] = 0; // (a - 128) / 256 truncates to <= 0
                else { const IAcc q = (a + 128) >> 8; out[ch] = (uint8_t)(q > 255 ? 255 : q); }
            }
        }
        P::store(dst.data, (size_t)r * dst.stride + (size_t)c, out);
    }
}

// Kernels larger than 15 x 15 (the reference takes any comptime size, convolution.zig:76): one lane per output pixel, taps
// from device memory, every source pixel fetched through the border rule. No tile, no size limit; the accumulation order
// is the same (ky-major, kx ascending).
template <int PIX, int MODE>
__global__ __launch_bounds__(256) void k_conv2d_big(DImg src, DImg dst, const void *taps, int kh, int kw, int border) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= dst.cols || r >= dst.rows) return;
    using Acc = typename std::conditional<MODE == 0, float, int64_t>::type;
    Acc acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
    const int hh = kh / 2, hw = kw / 2;
    for (int ky = 0; ky < kh; ++ky) {
        const int gr = resolve_index(r + ky - hh, src.rows, border);
        for (int kx = 0; kx < kw; ++kx) {
            const int gc = gr < 0 ? -1 : resolve_index(c + kx - hw, src.cols, border);
            Vec v = P::load(src.data, (size_t)max(gr, 0) * src.stride + (size_t)max(gc, 0)); // clamped address, unpredicated
            if (gc < 0) v = P::zero();
            if constexpr (MODE == 0) {
                const float w = ((const float *)taps)[ky * kw + kx];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) { const float p = v[ch] * w; acc[ch] = acc[ch] + p; }
            } else {
                const int32_t w = ((const int32_t *)taps)[ky * kw + kx];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) acc[ch] += (Acc)v[ch] * (Acc)w;
            }
        }
    }
    Vec out;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        if constexpr (MODE == 0) out[ch] = acc[ch];
        else {
            const Acc a = acc[ch];
            if (a < 0) out[ch] = 0;
            else { const Acc q = (a + 128) >> 8; out[ch] = (uint8_t)(q > 255 ? 255 : q); }
        }
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, out);
}

template <int PIX, int MODE>
static int launch_conv2d(const zg_image *src, const zg_image *dst, const Kernel2D &k, int border, hipStream_t s) {
    const int tiles_x = (int)ceil_div(dst->cols, C2_TW), tiles_y = (int)ceil_div(dst->rows, C2_TH);
    const dim3 grid((unsigned)(tiles_x * tiles_y));
#define ZG_C2(KH, KW) hipLaunchKernelGGL((k_conv2d<PIX, MODE, KH, KW>), grid, dim3(256), 0, s, dimg(src), dimg(dst), k, border, tiles_x)
    if (k.kh == 3 && k.kw == 3) ZG_C2(3, 3);
    else if (k.kh == 5 && k.kw == 5) ZG_C2(5, 5);
    else if (k.kh == 7 && k.kw == 7) {
        if constexpr (MODE == 3 && Px<PIX>::C > 1) ZG_C2(0, 0); // never launched (convolve_impl): unrolled, this form wants 288 registers
        else ZG_C2(7, 7);
    } else ZG_C2(0, 0);
#undef ZG_C2
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int convolve_impl(const zg_image *src, const zg_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "convolve: %ux%u vs %ux%u",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "convolve: pixel types differ");
    ZG_REQUIRE(kernel && kh >= 1 && kw >= 1 && kh <= 4096 && kw <= 4096, ZG_ERR_INVALID_ARGUMENT, "convolve: kernel %ux%u (sides of 1..4096)", kh, kw);
    ZG_REQUIRE(border >= ZG_BORDER_ZERO && border <= ZG_BORDER_WRAP, ZG_ERR_INVALID_ARGUMENT, "invalid border %d", border);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const size_t nk = (size_t)kh * kw;
    const bool is_float = pixel_is_float(src->pixel);
    std::vector<int32_t> ik;
    int mode = 0;
    if (!is_float) {
        ik.resize(nk);
        int64_t sum_abs = 0;
        for (size_t i = 0; i < nk; ++i) {
            const float r = std::round(kernel[i] * 256.0f); // ConvolutionKernel.flatten (convolution.zig:95-113)
            ZG_REQUIRE(std::fabs(r) < 2147483648.0f, ZG_ERR_INVALID_ARGUMENT, "kernel[%zu] does not fit i32 after scaling", i);
            ik[i] = (int32_t)r;
            sum_abs += std::llabs((long long)ik[i]);
        }
        mode = (255 * sum_abs < (int64_t)INT32_MAX - 256) ? 1 : 2;
        // every partial sum an integer below 2^24: f32 multiply-adds are exact. Used where it was measured faster — the unrolled 3 x 3 and
        // 5 x 5 kernels and the one-channel 7 x 7 (4096^2 Rgba(u8): 78.6 -> 72.9 and 155 -> 127 us; grey 31.9 / 43.2 / 59.1 -> 30.6 / 39.0 / 54.2 us);
        // the run-time-size kernel and 7 x 7 on three or four channels (288 registers when unrolled) keep the integer form
        const bool unrolled = (kh == 3 && kw == 3) || (kh == 5 && kw == 5) || (kh == 7 && kw == 7 && pixel_channels(src->pixel) == 1);
        if (255 * sum_abs < (1 << 24) && unrolled) mode = 3;
        if (255 * sum_abs < (1 << 24) && kh == kw && (kh == 3 || kh == 5)) {
            // every partial sum an integer below 2^24: one wave per column strip, rows resident in registers (conv2d_stream.hip)
            float fk[25];
            for (size_t i = 0; i < nk; ++i) fk[i] = (float)ik[i];
            const int rcs = try_conv2d_stream(src, dst, fk, (int)kh, (int)kw, border, s);
            if (rcs >= 0) return rcs;
        }
    }
    if (kh <= 15 && kw <= 15) { // tiled kernels, taps as a kernel argument
        Kernel2D k;
        k.kh = (int)kh;
        k.kw = (int)kw;
        for (size_t i = 0; i < nk; ++i) { if (is_float) k.f[i] = kernel[i]; else if (mode == 3) k.f[i] = (float)ik[i]; else k.i[i] = ik[i]; }
        return dispatch_pixel(src->pixel, [&](auto tag) -> int {
            constexpr int PIX = decltype(tag)::value;
            if constexpr (std::is_same<typename Px<PIX>::Elem, float>::value) return launch_conv2d<PIX, 0>(src, dst, k, border, s);
            else return mode == 3 ? launch_conv2d<PIX, 3>(src, dst, k, border, s) : mode == 1 ? launch_conv2d<PIX, 1>(src, dst, k, border, s) : launch_conv2d<PIX, 2>(src, dst, k, border, s);
        });
    }
    void *taps = nullptr; // larger: taps from device memory (uploaded synchronously: not capturable)
    if ((rc = scratch_alloc(&taps, nk * 4, s))) return rc;
    rc = upload_pageable(taps, is_float ? (const void *)kernel : (const void *)ik.data(), nk * 4, s);
    if (rc == ZG_OK)
        rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
            constexpr int PIX = decltype(tag)::value;
            const dim3 grid = row_grid(ceil_div(dst->cols, 256), dst->rows);
            if constexpr (std::is_same<typename Px<PIX>::Elem, float>::value)
                hipLaunchKernelGGL((k_conv2d_big<PIX, 0>), grid, dim3(256), 0, s, dimg(src), dimg(dst), (const void *)taps, (int)kh, (int)kw, border);
            else
                hipLaunchKernelGGL((k_conv2d_big<PIX, 2>), grid, dim3(256), 0, s, dimg(src), dimg(dst), (const void *)taps, (int)kh, (int)kw, border);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
    scratch_free(taps, s);
    return rc;
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_convolve(const zg_image *src, const zg_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border, zg_stream stream) {
    return convolve_impl(src, dst, kernel, kh, kw, border, as_stream(stream));
}

int zg_convolve_host(const zg_image *src, const zg_image *dst, const float *kernel, uint32_t kh, uint32_t kw, int border) {
    if (border != ZG_BORDER_WRAP && kh >= 1) {
        const int brc = host_banded(src, dst, kh / 2, [&](const zg_image *sv, const zg_image *dv, hipStream_t s) { return convolve_impl(sv, dv, kernel, kh, kw, border, s); });
        if (brc >= 0) return brc;
    }
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = convolve_impl(&a.dev, &b.dev, kernel, kh, kw, border, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
