// resize_planes.hip — Image(Rgb(u8) / Rgba(u8)).resize: the reference's per-plane u8 resizers.
//
// Replaces reference src/image/interpolation.zig:111-186 (split -> plane kernel -> merge) and the plane
// kernels src/image/channel_ops.zig:144-190 (bilinear, fx = trunc(frac*256), no rounding offset), :193-214
// (nearest, clamp not mirror), :217-435 (bicubic / Catmull-Rom / Mitchell in 8.8 integers) and :438-493
// (Lanczos3, f32 weights from @sin).
//
// Every one of those kernels is separable in its *coordinates*: the source indices and tap weights of an
// output pixel depend only on its column (x taps) and its row (y taps). For nearest / bilinear / the three
// integer cubics the taps are plain IEEE f32 and integer arithmetic and are recomputed per lane in registers
// (a table load in front of the pixel gathers would double the dependent memory latency of this few-microsecond
// kernel: 9.7 -> see profiles). Lanczos weights need sin, so the host builds two small tables for it with the
// reference's arithmetic and caches them per geometry. The device kernel is pure integer (or f32 mul / add)
// accumulation over interleaved pixels: channels are independent, so no split / merge pass exists here.
// Per output pixel the kernel reads T x T source pixels and writes one: that is the algorithmic traffic.
#include "zg_common.h"
#include "zg_hostmath.h"
#include "zg_bilinear_u8.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#pragma clang fp contract(off)

namespace zg {

enum : int { RC_NEAREST = 0, RC_BILINEAR = 1, RC_CUBIC_INT = 2, RC_LANCZOS = 3 };

// One axis of taps: for destination index d, source indices idx[d*T + k] and weights w[d*T + k]
// (int32, or f32 bits for Lanczos).
struct AxisTable {
    const int32_t *idx;
    const int32_t *w;
};

// Integer 8.8 kernels of channel_ops.zig:217-435 (device copies of the host functions below).
__device__ inline int dk_bicubic(int t) {
    const int at = t < 0 ? -t : t;
    if (at <= 256) { const int t2 = at * at / 256, t3 = t2 * at / 256; return 256 - 2 * t2 + t3; }
    if (at <= 512) { const int t2 = at * at / 256, t3 = t2 * at / 256; return 4 * 256 - 8 * at + 5 * t2 - t3; }
    return 0;
}
__device__ inline int dk_catmull(int t) {
    const int at = t < 0 ? -t : t;
    if (at <= 256) { const int t2 = at * at / 256, t3 = t2 * at / 256; return 256 - (5 * t2) / 2 + (3 * t3) / 2; }
    if (at <= 512) { const int t2 = at * at / 256, t3 = t2 * at / 256; return 2 * 256 - 4 * at + (5 * t2) / 2 - t3 / 2; }
    return 0;
}
__device__ inline int dk_mitchell(int t) {
    const long long at = t < 0 ? -(long long)t : t, s = 256, s2 = s * s, s3 = s2 * s;
    if (at < s) { const long long at2 = at * at, at3 = at2 * at; return (int)((21 * at3 - 36 * at2 * s + 16 * s3) / (18 * s2)); }
    if (at < 2 * s) { const long long at2 = at * at, at3 = at2 * at; return (int)((-7 * at3 + 36 * at2 * s - 60 * at * s2 + 32 * s3) / (18 * s2)); }
    return 0;
}

// Taps of one axis for destination index d, computed in registers with the reference's f32 / integer arithmetic
// (s = (d + 0.5) * ratio - 0.5; mirror indices; f = trunc(frac * 256)). Plain IEEE mul / sub / floor / trunc, so the
// device reproduces the host bit for bit and no table load sits in front of the pixel gathers.
template <int CLS, int KIND, int T>
__device__ inline void axis_taps(int d, float ratio, int n, int (&idx)[T], int (&w)[T]) {
    const float sf = ((float)d + 0.5f) * ratio - 0.5f;
    if constexpr (CLS == RC_NEAREST) {
        unsigned i = (unsigned)(int)roundf(sf);
        if (i > (unsigned)(n - 1)) i = (unsigned)(n - 1);
        idx[0] = (int)i;
        w[0] = 0;
    } else {
        const float fl = floorf(sf);
        const int base = (int)fl;
        const int f = (int)truncf((sf - fl) * 256.0f);
        if constexpr (CLS == RC_BILINEAR) {
            idx[0] = resolve_index(base, n, ZG_BORDER_MIRROR);
            idx[1] = resolve_index(base + 1, n, ZG_BORDER_MIRROR);
            w[0] = 256 - f;
            w[1] = f;
        } else {
#pragma unroll
            for (int k = 0; k < T; ++k) {
                idx[k] = resolve_index(base + k - 1, n, ZG_BORDER_MIRROR);
                const int t = k * 256 - 256 - f;
                w[k] = KIND == ZG_INTERP_BICUBIC ? dk_bicubic(t) : (KIND == ZG_INTERP_CATMULL_ROM ? dk_catmull(t) : dk_mitchell(t));
            }
        }
    }
}

template <int PIX, int CLS, int KIND, int T>
__global__ __launch_bounds__(256) void k_resize_planes(DImg src, DImg dst, AxisTable tx, AxisTable ty, float ratio_x, float ratio_y, int tiles_x, FrameSpan fr) {
    using P = Px<PIX>;
    src.data = (uint8_t *)src.data + (size_t)blockIdx.y * fr.src_frame; // the frame is the grid's y
    dst.data = (uint8_t *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    using Vec = typename P::Vec;
    constexpr int C = P::C;
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    const int tyi = wg / tiles_x, txi = wg - tyi * tiles_x;
    const int c = txi * 64 + (int)(threadIdx.x & 63);
    const int r = __builtin_amdgcn_readfirstlane(tyi * 4 + (int)(threadIdx.x >> 6)); // one row per wave
    if (r >= dst.rows || c >= dst.cols) return;

    int xi[T], yi[T], wxi[T], wyi[T];
    if constexpr (CLS == RC_LANCZOS) { // f32 weights need sin: host tables (channel_ops.zig:446-454)
#pragma unroll
        for (int k = 0; k < T; ++k) { xi[k] = tx.idx[c * T + k]; yi[k] = ty.idx[r * T + k]; wxi[k] = tx.w[c * T + k]; wyi[k] = ty.w[r * T + k]; }
    } else {
        axis_taps<CLS, KIND, T>(c, ratio_x, src.cols, xi, wxi);
        axis_taps<CLS, KIND, T>(r, ratio_y, src.rows, yi, wyi);
    }

    // Rgba(u8), four or six taps: a row's taps are 16 / 24 adjacent bytes unless the row end clamps them, and a gather costs the texture path per lane
    // ADDRESS — 36 dword gathers per output bound the Lanczos kernel (540 us for 64 frames of 1080p -> 450 x 800, twice its arithmetic). Where every
    // lane of the wave has adjacent taps, a row is one 16-byte buffer load (+ one of 8), the row's offset scalar.
    [[maybe_unused]] bool wide_rows = false;
    [[maybe_unused]] uint32_t row_px[T][T]; // [j][i]: the pixel of tap (i, j) as a dword
    if constexpr ((CLS == RC_LANCZOS || CLS == RC_CUBIC_INT) && PIX == ZG_PIXEL_RGBA_U8 && (T == 4 || T == 6)) {
        const size_t frame_bytes = (size_t)src.rows * src.stride * 4;
        const bool adjacent = xi[T - 1] - xi[0] == T - 1;
        if (frame_bytes < (1ull << 32) && __builtin_amdgcn_ballot_w64(adjacent) == __builtin_amdgcn_ballot_w64(true)) { // wave-uniform
            typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src.data, (short)0, (int)(uint32_t)frame_bytes, 0x00020000);
            const int voff = xi[0] * 4;
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int soff = __builtin_amdgcn_readfirstlane((int)(uint32_t)((size_t)yi[j] * src.stride * 4)); // the row of a tap is the wave's
                const u32x4s q = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3]; // (a bit_cast of an element expression reads element 0: clang)
                row_px[j][0] = q0; row_px[j][1] = q1; row_px[j][2] = q2; row_px[j][3] = q3;
                if constexpr (T == 6) {
                    const u32x2s h = __builtin_bit_cast(u32x2s, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + 16, soff, 0));
                    const uint32_t h0 = h[0], h1 = h[1];
                    row_px[j][4] = h0; row_px[j][5] = h1;
                }
            }
            wide_rows = true;
        }
    }
    Vec out;
    if constexpr (CLS == RC_NEAREST) {
        out = P::load(src.data, (size_t)yi[0] * src.stride + (size_t)xi[0]);
    } else if constexpr (CLS == RC_BILINEAR) {
        const int fx = wxi[1], fy = wyi[1]; // w = {256 - f, f}
        const Vec tl = P::load(src.data, (size_t)yi[0] * src.stride + (size_t)xi[0]);
        const Vec tr = P::load(src.data, (size_t)yi[0] * src.stride + (size_t)xi[1]);
        const Vec bl = P::load(src.data, (size_t)yi[1] * src.stride + (size_t)xi[0]);
        const Vec br = P::load(src.data, (size_t)yi[1] * src.stride + (size_t)xi[1]);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const int top = (int)tl[ch] * (256 - fx) + (int)tr[ch] * fx;
            const int bottom = (int)bl[ch] * (256 - fx) + (int)br[ch] * fx;
            const int result = (top * (256 - fy) + bottom * fy) >> 16; // @divTrunc(.., 65536), operand >= 0
            out[ch] = (uint8_t)result;                                  // <= 255 by construction
        }
    } else if constexpr (CLS == RC_CUBIC_INT) {
        int wx[T], wy[T];
#pragma unroll
        for (int k = 0; k < T; ++k) { wx[k] = wxi[k]; wy[k] = wyi[k]; }
        auto run = [&](auto tap) {
            int sum[C], weight_sum = 0;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) sum[ch] = 0;
#pragma unroll
            for (int j = 0; j < T; ++j) {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const int w = (wx[i] * wy[j]) / 256; // @divTrunc: signed, toward zero
                    const Vec p = tap(i, j);
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) sum[ch] += (int)p[ch] * w;
                    weight_sum += w;
                }
            }
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out[ch] = clamp_u8_i32(weight_sum != 0 ? sum[ch] / weight_sum : 0);
        };
        if (wide_rows) run([&](int i, int j) -> Vec { return __builtin_bit_cast(Vec, row_px[j][i]); });
        else run([&](int i, int j) -> Vec { return P::load(src.data, (size_t)yi[j] * src.stride + (size_t)xi[i]); });
    } else {
        float wx[T], wy[T];
#pragma unroll
        for (int k = 0; k < T; ++k) { wx[k] = __int_as_float(wxi[k]); wy[k] = __int_as_float(wyi[k]); }
        auto run = [&](auto tap) {
            float sum[C], weight_sum = 0;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) sum[ch] = 0;
#pragma unroll
            for (int j = 0; j < T; ++j) {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const float w = wx[i] * wy[j];
                    const Vec p = tap(i, j);
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float prod = (float)p[ch] * w;
                        sum[ch] = sum[ch] + prod;
                    }
                    weight_sum = weight_sum + w;
                }
            }
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out[ch] = clamp_u8_f32(weight_sum != 0 ? sum[ch] / weight_sum : 0.0f);
        };
        if (wide_rows) run([&](int i, int j) -> Vec { return __builtin_bit_cast(Vec, row_px[j][i]); });
        else run([&](int i, int j) -> Vec { return P::load(src.data, (size_t)yi[j] * src.stride + (size_t)xi[i]); });
    }
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, out);
}

// ---- host-side tap tables (channel_ops.zig arithmetic, verbatim) --------------------------------------
static int32_t k_bicubic_i(int32_t t) {
    const int32_t at = t < 0 ? -t : t;
    if (at <= 256) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 256 - 2 * t2 + t3; }
    if (at <= 512) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 4 * 256 - 8 * at + 5 * t2 - t3; }
    return 0;
}
static int32_t k_catmull_i(int32_t t) {
    const int32_t at = t < 0 ? -t : t;
    if (at <= 256) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 256 - (5 * t2) / 2 + (3 * t3) / 2; }
    if (at <= 512) { const int32_t t2 = at * at / 256, t3 = t2 * at / 256; return 2 * 256 - 4 * at + (5 * t2) / 2 - t3 / 2; }
    return 0;
}
static int32_t k_mitchell_i(int32_t t) {
    const int64_t at = t < 0 ? -(int64_t)t : t, s = 256, s2 = s * s, s3 = s2 * s;
    if (at < s) { const int64_t at2 = at * at, at3 = at2 * at; return (int32_t)((21 * at3 - 36 * at2 * s + 16 * s3) / (18 * s2)); }
    if (at < 2 * s) { const int64_t at2 = at * at, at3 = at2 * at; return (int32_t)((-7 * at3 + 36 * at2 * s - 60 * at * s2 + 32 * s3) / (18 * s2)); }
    return 0;
}
static float k_lanczos_plane(float x) { // channel_ops.zig:446-454
    if (x == 0) return 1.0f;
    const float a = 3.0f;
    if (std::fabs(x) >= a) return 0.0f;
    const float pi_x = 3.14159265358979323846f * x;
    return (a * hostmath::sin_f32(pi_x) * hostmath::sin_f32(pi_x / a)) / (pi_x * pi_x);
}

static void build_axis(int kind, uint32_t src_n, uint32_t dst_n, int taps, std::vector<int32_t> &idx, std::vector<int32_t> &w) {
    idx.assign((size_t)dst_n * taps, 0);
    w.assign((size_t)dst_n * taps, 0);
    const float ratio = (float)src_n / (float)dst_n;
    for (uint32_t d = 0; d < dst_n; ++d) {
        const float sf = ((float)d + 0.5f) * ratio - 0.5f;
        if (kind == ZG_INTERP_NEAREST) { // @max(0, @min(n - 1, @as(u32, @round(s))))
            uint32_t i = (uint32_t)std::round(sf);
            if (i > src_n - 1) i = src_n - 1;
            idx[d] = (int32_t)i;
            continue;
        }
        const float fl = std::floor(sf);
        const int base = (int)fl;
        if (kind == ZG_INTERP_BILINEAR) {
            const int32_t f = (int32_t)std::trunc((sf - fl) * 256.0f);
            idx[d * 2 + 0] = resolve_index(base, (int)src_n, ZG_BORDER_MIRROR);
            idx[d * 2 + 1] = resolve_index(base + 1, (int)src_n, ZG_BORDER_MIRROR);
            w[d * 2 + 0] = 256 - f;
            w[d * 2 + 1] = f;
        } else if (kind == ZG_INTERP_LANCZOS) {
            const float f = sf - fl;
            for (int k = 0; k < 6; ++k) {
                idx[d * 6 + k] = resolve_index(base + k - 2, (int)src_n, ZG_BORDER_MIRROR);
                const float wk = k_lanczos_plane((float)(k - 2) - f);
                int32_t bits;
                std::memcpy(&bits, &wk, 4);
                w[d * 6 + k] = bits;
            }
        } else {
            const int32_t f = (int32_t)std::trunc((sf - fl) * 256.0f);
            for (int k = 0; k < 4; ++k) {
                idx[d * 4 + k] = resolve_index(base + k - 1, (int)src_n, ZG_BORDER_MIRROR);
                const int32_t t = k * 256 - 256 - f;
                w[d * 4 + k] = kind == ZG_INTERP_BICUBIC ? k_bicubic_i(t) : (kind == ZG_INTERP_CATMULL_ROM ? k_catmull_i(t) : k_mitchell_i(t));
            }
        }
    }
}

// Device-resident tables, cached per (device, kind, src_n, dst_n): repeated resizes of the same geometry
// (video frames, batches) upload nothing and are graph-capturable after the first call.
// Lifetime: the cache and every caller that has fetched a table share ownership (shared_ptr). The cache is LRU (a hit
// moves the entry to the back), eviction only drops the cache's reference, and a caller keeps its reference until its
// kernel has been launched — so a table that a launch still points at is never freed under it, whatever other host
// threads insert meanwhile. The last owner's release is a hipFree, which waits for the device: a kernel already in
// flight finishes before the memory goes away.
struct AxisBuf {
    int32_t *dev = nullptr;
    size_t n = 0;
    ~AxisBuf() { if (dev) (void)hipFree(dev); }
};
typedef std::tuple<int, int, uint32_t, uint32_t> AxisKey;
static std::mutex g_cache_mu;
static std::map<AxisKey, std::pair<std::shared_ptr<AxisBuf>, std::list<AxisKey>::iterator>> g_cache;
static std::list<AxisKey> g_cache_order; // least recently used first

static int axis_table(int kind, uint32_t src_n, uint32_t dst_n, int taps, AxisTable &out, std::shared_ptr<AxisBuf> &hold) {
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));
    const AxisKey key = std::make_tuple(dev, kind, src_n, dst_n);
    std::shared_ptr<AxisBuf> evicted; // released after the lock is dropped (hipFree may block)
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) {
            g_cache_order.splice(g_cache_order.end(), g_cache_order, it->second.second); // most recently used
            hold = it->second.first;
        } else {
            std::vector<int32_t> idx, w;
            build_axis(kind, src_n, dst_n, taps, idx, w);
            auto buf = std::make_shared<AxisBuf>(); // frees its device block on every error path below
            buf->n = idx.size();
            ZG_HIP(hipMalloc((void **)&buf->dev, 2 * buf->n * sizeof(int32_t)));
            if (int rc = upload_pageable(buf->dev, idx.data(), buf->n * sizeof(int32_t), nullptr)) return rc;
            if (int rc = upload_pageable(buf->dev + buf->n, w.data(), buf->n * sizeof(int32_t), nullptr)) return rc;
            if (g_cache.size() >= 64) { // bounded: the least recently used geometry leaves the cache
                auto old = g_cache.find(g_cache_order.front());
                evicted = std::move(old->second.first);
                g_cache.erase(old);
                g_cache_order.pop_front();
            }
            g_cache_order.push_back(key);
            g_cache.emplace(key, std::make_pair(buf, std::prev(g_cache_order.end())));
            hold = buf;
        }
    }
    out.idx = hold->dev;
    out.w = hold->dev + hold->n;
    return ZG_OK;
}

// Bilinear Rgba(u8), the pixel type and method BASELINE's resize is quoted on: same arithmetic as the RC_BILINEAR branch
// above (channel_ops.zig:144-190) with the per-pixel overheads taken out:
//   * the mirror rule is only evaluated by lanes whose taps leave the image (everything else is base, base + 1);
//   * the two horizontal taps of a row are adjacent pixels, so they come from ONE 8-byte load;
//   * NPX = 4 (upscales and mild downscales, where neighbouring destination pixels read neighbouring source pixels):
//     four destination pixels per lane, the row taps computed once, one 16-byte store.
//     For strong downscales NPX = 4 is slower (12.2 us against 8.8 us for 4096^2 -> 1024^2: the lanes of one gather
//     instruction then sit 256 bytes apart instead of 16), so those keep one pixel per lane.
// A launch covers `frames` equally shaped images laid out `src_frame` / `dst_frame` bytes apart (a batch of the pipeline, batch.hip):
// a 4096^2 -> 1024^2 frame is 4 096 workgroups of one gather each, i.e. launch ramp and tail; sixteen of them in one grid keep the chip
// full (profiles/r03_batched_resize.txt).
template <int NPX, int WAVES, bool XCD, int RPW>
__global__ __launch_bounds__(64 * WAVES) void k_resize_bilinear_rgba8(DImg src, DImg dst, float ratio_x, float ratio_y, int tiles_x, FrameSpan fr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    int wg = blockIdx.x;
    if constexpr (XCD) {
        const int nwg = gridDim.x, per_xcd = nwg >> 3;
        if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    }
    src.data = (char *)src.data + (size_t)blockIdx.y * fr.src_frame;
    dst.data = (char *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    const int tyi = wg / tiles_x, txi = wg - tyi * tiles_x;
    const int c0 = (txi * 64 + (int)(threadIdx.x & 63)) * NPX;
    const int r0 = __builtin_amdgcn_readfirstlane((tyi * WAVES + (int)(threadIdx.x >> 6)) * RPW); // RPW consecutive rows per wave
    if (r0 >= dst.rows || c0 >= dst.cols) return;
    int x0[NPX], x1[NPX], fx[NPX], xp[NPX];
#pragma unroll
    for (int p = 0; p < NPX; ++p) {
        bilinear_taps(c0 + p, ratio_x, src.cols, x0[p], x1[p], fx[p]);
        xp[p] = min(x0[p], src.cols - 2); // the pair (xp, xp + 1) is always inside the row (cols >= 2)
    }
    const uint32_t *row0[RPW], *row1[RPW];
    int fy[RPW];
    u32x2 p0[RPW][NPX], p1[RPW][NPX];
#pragma unroll
    for (int k = 0; k < RPW; ++k) { // every load of the wave is asked for before the first is used
        int y0, y1;
        bilinear_taps(min(r0 + k, dst.rows - 1), ratio_y, src.rows, y0, y1, fy[k]);
        row0[k] = (const uint32_t *)src.data + (size_t)y0 * src.stride;
        row1[k] = (const uint32_t *)src.data + (size_t)y1 * src.stride;
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
            p0[k][p] = *(const u32x2 *)(row0[k] + xp[p]); // 4-byte aligned 8-byte loads
            p1[k][p] = *(const u32x2 *)(row1[k] + xp[p]);
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        if (r0 + k >= dst.rows) break; // wave-uniform
        uint32_t out[NPX];
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
            uint32_t tl = p0[k][p][0], tr = p0[k][p][1], bl = p1[k][p][0], br = p1[k][p][1];
            if (x1[p] != x0[p] + 1 || x0[p] > src.cols - 2) { tl = row0[k][x0[p]]; tr = row0[k][x1[p]]; bl = row1[k][x0[p]]; br = row1[k][x1[p]]; } // mirrored taps
            out[p] = bilinear_rgba8(tl, tr, bl, br, fx[p], fy[k]);
        }
        uint32_t *o = (uint32_t *)dst.data + (size_t)(r0 + k) * dst.stride + c0;
        if constexpr (NPX == 4) *(u32x4 *)o = u32x4{out[0], out[1], out[2], out[3]}; // dst.cols % 4 == 0, 16-byte aligned rows
        else o[0] = out[0];
    }
}

// Image(Rgba(u8)).resize(.bilinear) of n frames in one launch; -1 when the fast kernel does not apply (the caller goes frame by frame).
int resize_bilinear_rgba8_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_RGBA_U8 || dst->pixel != ZG_PIXEL_RGBA_U8 || src->cols < 2 || src->rows == 0 || dst->rows == 0 || dst->cols == 0 || n == 0) return -1;
    if (src->rows == dst->rows && src->cols == dst->cols) return -1; // equal sizes are a copy (interpolation.zig:100-108)
    // Launch form (profiles/r05_experiments.txt; tools/exp/copy_floor.hip is why): strong reductions read a few bytes per output pixel, and there
    // a grid of one-wave workgroups handed out in address order keeps the request stream a moving front — 4096^2 -> 1024^2: 9.6 -> 8.4 us per
    // launch; two rows per wave for 2 : 1 (61 -> 59 us at 8192^2 -> 4096^2). Enlargements keep round 4's four-row workgroups in XCD-major
    // order (their neighbouring rows share source rows: 35.9 against 37-40 us at 2048^2 -> 4096^2). 0 / 1 / 2 / 4 force a form (the tests do).
    const float ry = (float)src->rows / (float)dst->rows, rx = (float)src->cols / (float)dst->cols;
    int form = rx <= 1.5f ? 0 : (ry >= 3.0f ? 1 : 2);
    static const int forced_form = getenv("ZIGNAL_HIP_RESIZE_FORM") ? atoi(getenv("ZIGNAL_HIP_RESIZE_FORM")) : -1; // read once
    if (forced_form == 0 || forced_form == 1 || forced_form == 2 || forced_form == 4) form = forced_form; // the forms the launch switch has: anything else would leave rows unwritten
    const int waves = form == 0 ? 4 : 1, rpw = form == 0 ? 1 : form;
    const int tiles_x = (int)ceil_div(dst->cols, 64), tiles_y = (int)ceil_div(dst->rows, (unsigned)(waves * rpw));
    const float ratio_x = (float)src->cols / (float)dst->cols, ratio_y = (float)src->rows / (float)dst->rows;
    const bool x4 = ratio_x <= 1.5f && dst->cols % 4 == 0 && dst->stride % 4 == 0 && ((uintptr_t)dst->data & 15) == 0 && dst_frame % 16 == 0;
    const int tx = x4 ? (int)ceil_div(dst->cols, 256) : tiles_x;
    const uint64_t tiles = (uint64_t)tx * tiles_y;
    if (tiles > 0x7fffffffu || n > MAX_FRAMES_PER_LAUNCH) return -1;
    const dim3 grid((unsigned)tiles, n);
    const FrameSpan fr{src_frame, dst_frame};
#define ZG_RB(NPX) \
    switch (form) { \
    case 1: hipLaunchKernelGGL((k_resize_bilinear_rgba8<NPX, 1, false, 1>), grid, dim3(64), 0, s, dimg(src), dimg(dst), ratio_x, ratio_y, tx, fr); break; \
    case 2: hipLaunchKernelGGL((k_resize_bilinear_rgba8<NPX, 1, false, 2>), grid, dim3(64), 0, s, dimg(src), dimg(dst), ratio_x, ratio_y, tx, fr); break; \
    case 4: hipLaunchKernelGGL((k_resize_bilinear_rgba8<NPX, 1, false, 4>), grid, dim3(64), 0, s, dimg(src), dimg(dst), ratio_x, ratio_y, tx, fr); break; \
    default: hipLaunchKernelGGL((k_resize_bilinear_rgba8<NPX, 4, true, 1>), grid, dim3(256), 0, s, dimg(src), dimg(dst), ratio_x, ratio_y, tx, fr); \
    }
    if (x4) { ZG_RB(4) } else { ZG_RB(1) }
#undef ZG_RB
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int PIX, int CLS, int KIND, int T>
static int launch_planes(const zg_image *src, const zg_image *dst, const AxisTable &tx, const AxisTable &ty, uint32_t n, const FrameSpan &fr, hipStream_t s) {
    const int tiles_x = (int)ceil_div(dst->cols, 64), tiles_y = (int)ceil_div(dst->rows, 4);
    const float ratio_x = (float)src->cols / (float)dst->cols, ratio_y = (float)src->rows / (float)dst->rows;
    if constexpr (PIX == ZG_PIXEL_RGBA_U8 && CLS == RC_BILINEAR) {
        const int rcb = resize_bilinear_rgba8_frames(src, dst, n, fr.src_frame, fr.dst_frame, s);
        if (rcb >= 0) return rcb;
    }
    if (n > MAX_FRAMES_PER_LAUNCH) return -1;
    hipLaunchKernelGGL((k_resize_planes<PIX, CLS, KIND, T>), dim3((unsigned)(tiles_x * tiles_y), n), dim3(256), 0, s, dimg(src), dimg(dst), tx, ty,
                       ratio_x, ratio_y, tiles_x, fr);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int PIX>
static int resize_planes_pix(const zg_image *src, const zg_image *dst, int kind, const AxisTable &tx, const AxisTable &ty, uint32_t n, const FrameSpan &fr, hipStream_t s) {
    switch (kind) {
    case ZG_INTERP_NEAREST: return launch_planes<PIX, RC_NEAREST, ZG_INTERP_NEAREST, 1>(src, dst, tx, ty, n, fr, s);
    case ZG_INTERP_BILINEAR: return launch_planes<PIX, RC_BILINEAR, ZG_INTERP_BILINEAR, 2>(src, dst, tx, ty, n, fr, s);
    case ZG_INTERP_LANCZOS: return launch_planes<PIX, RC_LANCZOS, ZG_INTERP_LANCZOS, 6>(src, dst, tx, ty, n, fr, s);
    case ZG_INTERP_BICUBIC: return launch_planes<PIX, RC_CUBIC_INT, ZG_INTERP_BICUBIC, 4>(src, dst, tx, ty, n, fr, s);
    case ZG_INTERP_CATMULL_ROM: return launch_planes<PIX, RC_CUBIC_INT, ZG_INTERP_CATMULL_ROM, 4>(src, dst, tx, ty, n, fr, s);
    default: return launch_planes<PIX, RC_CUBIC_INT, ZG_INTERP_MITCHELL, 4>(src, dst, tx, ty, n, fr, s);
    }
}

// Image(Rgb(u8) / Rgba(u8)).resize for differing sizes; caller (resize_impl) has validated the images.
int resize_planes_impl(const zg_image *src, const zg_image *dst, const zg_method *method, hipStream_t s) {
    return resize_planes_frames(src, dst, method, 1, 0, 0, s);
}

// n frames of the same two shapes in one launch (frame = blockIdx.y); -1 past the grid's limit.
int resize_planes_frames(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    const FrameSpan fr{src_frame, dst_frame};
    const int kind = method->kind;
    const int taps = kind == ZG_INTERP_NEAREST ? 1 : (kind == ZG_INTERP_BILINEAR ? 2 : (kind == ZG_INTERP_LANCZOS ? 6 : 4));
    AxisTable tx{}, ty{};
    std::shared_ptr<AxisBuf> hold_x, hold_y; // keep both tables alive until the kernel below has been launched
    int rc;
    if (kind == ZG_INTERP_LANCZOS) { // the only kernel whose weights need a transcendental: tables from the host
        if ((rc = axis_table(kind, src->cols, dst->cols, taps, tx, hold_x))) return rc;
        if ((rc = axis_table(kind, src->rows, dst->rows, taps, ty, hold_y))) return rc;
    }
    if (src->pixel == ZG_PIXEL_RGB_U8) return resize_planes_pix<ZG_PIXEL_RGB_U8>(src, dst, kind, tx, ty, n, fr, s);
    return resize_planes_pix<ZG_PIXEL_RGBA_U8>(src, dst, kind, tx, ty, n, fr, s);
}


// The Lanczos3 plane weights of one axis (channel_ops.zig:446-466): for destination index d, weight k is
// lanczosKernel((k - 2) - f) with f = frac((d + 0.5) * ratio - 0.5). dst_n * 6 floats.
void lanczos_plane_weights(uint32_t src_n, uint32_t dst_n, float *w) {
    std::vector<int32_t> idx, bits;
    build_axis(ZG_INTERP_LANCZOS, src_n, dst_n, 6, idx, bits);
    std::memcpy(w, bits.data(), bits.size() * sizeof(float));
}

// resizePlaneLanczosU8 with caller-made weights: the taps' source indices are the library's (integer arithmetic), the
// weights are whatever the caller's @sin produced. Tables go up per call (KBs), outside the geometry cache.
int resize_lanczos_weights_impl(const zg_image *src, const zg_image *dst, const float *wx, const float *wy, hipStream_t s) {
    const size_t nx = (size_t)dst->cols * 6, ny = (size_t)dst->rows * 6;
    std::vector<int32_t> ix, iy, own;
    build_axis(ZG_INTERP_LANCZOS, src->cols, dst->cols, 6, ix, own);
    std::vector<int32_t> host(2 * nx + 2 * ny);
    std::memcpy(host.data(), ix.data(), nx * 4);
    std::memcpy(host.data() + nx, wx ? (const void *)wx : (const void *)own.data(), nx * 4);
    build_axis(ZG_INTERP_LANCZOS, src->rows, dst->rows, 6, iy, own);
    std::memcpy(host.data() + 2 * nx, iy.data(), ny * 4);
    std::memcpy(host.data() + 2 * nx + ny, wy ? (const void *)wy : (const void *)own.data(), ny * 4);
    int32_t *dev = nullptr;
    int rc;
    if ((rc = scratch_alloc((void **)&dev, host.size() * 4, s))) return rc;
    if ((rc = upload_pageable(dev, host.data(), host.size() * 4, s)) == ZG_OK) {
        const AxisTable tx{dev, dev + nx}, ty{dev + 2 * nx, dev + 2 * nx + ny};
        rc = src->pixel == ZG_PIXEL_RGB_U8 ? resize_planes_pix<ZG_PIXEL_RGB_U8>(src, dst, ZG_INTERP_LANCZOS, tx, ty, 1, FrameSpan{0, 0}, s)
                                           : resize_planes_pix<ZG_PIXEL_RGBA_U8>(src, dst, ZG_INTERP_LANCZOS, tx, ty, 1, FrameSpan{0, 0}, s);
    }
    scratch_free(dev, s);
    return rc;
}

} // namespace zg
