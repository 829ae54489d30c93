// geom.hip — backward-mapped resampling: one thread per destination pixel computes a source coordinate and
// samples it with zg_sample.h. Covers
//   resizeGeneric       reference src/image/interpolation.zig:194-214 (u8, f32, Rgb(f32), Rgba(f32) ...)
//   Image.warp          reference src/image/transforms.zig:522-531 + project() of
//                       src/geometry/transforms.zig:39-42,147-150,224-231 (SMatrix.gemm 3-term scalar dot,
//                       left to right, then scale by the reciprocal of w)
//   Image.rotateInto    reference src/image/transforms.zig:163-212 (general angle; 0/90/180/270 are exact
//                       permutations, :385-462)
//   Image.extract       reference src/image/transforms.zig:231-282 (general path; the aligned case is copyRect)
//   Image.crop          reference src/image/transforms.zig:216-222 -> copyRect :465-518 (bit-exact copy)
//   Image.insert        reference src/image/transforms.zig:293-378 + assignPixel src/image.zig:67-94
//   Image.letterbox     reference src/image/transforms.zig:49-108 (host logic around resize)
//
// Every coordinate is computed in f32 with the reference's operation order and no FMA; cos/sin of the angle
// are host-supplied scalars. A wave covers 64 consecutive destination pixels of one row (coalesced stores);
// the source side is a gather served by L1/L2 (neighbouring lanes read neighbouring source pixels).
// Measured and rejected: staging each workgroup's source bounding box in LDS (wave/LDS min-max reduction, coalesced
// rectangle load, taps via ds_read_b128) — 268-284 us vs 236 us for the direct gather on the 4096^2 RGBA f32 bicubic
// warp: the reduction, two extra barriers and a dependent load phase cost more than the L1 traffic they remove.
#include "zg_common.h"
#include "zg_hostmath.h"
#include "zg_sample.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#pragma clang fp contract(off)

namespace zg {

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);
int fill_outside_impl(const zg_image *img, const void *pixel_value, int l, int t, int r, int b, hipStream_t s);
int set_border_impl(const zg_image *img, const uint32_t rect[4], const void *pixel_value, hipStream_t s);
int resize_planes_impl(const zg_image *src, const zg_image *dst, const zg_method *method, hipStream_t s);
int resize_lanczos_weights_impl(const zg_image *src, const zg_image *dst, const float *wx, const float *wy, hipStream_t s);
void lanczos_plane_weights(uint32_t src_n, uint32_t dst_n, float *w);

enum : int { GEOM_RESIZE = 0, GEOM_PROJECTIVE = 1, GEOM_AFFINE = 2, GEOM_ROTATE = 3, GEOM_EXTRACT = 4 };

struct GeomParams {
    int mode;
    float p[12];
    int stage; // wave-level staging of shared taps where the kernel has it (a tuning hook turns it off)
};

// Source coordinate of destination pixel (c, r).
__device__ inline void source_coord(const GeomParams &g, int c, int r, float &sx, float &sy) {
    const float x = (float)c, y = (float)r;
    switch (g.mode) {
    case GEOM_RESIZE: // src = (dst + 0.5) * scale - 0.5
        sx = (x + 0.5f) * g.p[0] - 0.5f;
        sy = (y + 0.5f) * g.p[1] - 0.5f;
        break;
    case GEOM_PROJECTIVE: { // gemm: acc = 0; acc += a*b (k ascending); result = 0 + 1.0 * acc
        float acc, X, Y, W;
        acc = 0.0f; acc = acc + g.p[0] * x; acc = acc + g.p[1] * y; acc = acc + g.p[2] * 1.0f; X = 0.0f + 1.0f * acc;
        acc = 0.0f; acc = acc + g.p[3] * x; acc = acc + g.p[4] * y; acc = acc + g.p[5] * 1.0f; Y = 0.0f + 1.0f * acc;
        acc = 0.0f; acc = acc + g.p[6] * x; acc = acc + g.p[7] * y; acc = acc + g.p[8] * 1.0f; W = 0.0f + 1.0f * acc;
        if (W != 0.0f) {
            const float inv = 1.0f / W;
            X = inv * X;
            Y = inv * Y;
        }
        sx = X;
        sy = Y;
        break;
    }
    case GEOM_AFFINE: { // matrix.dot(src).add(bias)
        float acc, X, Y;
        acc = 0.0f; acc = acc + g.p[0] * x; acc = acc + g.p[1] * y; X = 0.0f + 1.0f * acc;
        acc = 0.0f; acc = acc + g.p[2] * x; acc = acc + g.p[3] * y; Y = 0.0f + 1.0f * acc;
        sx = X + g.p[4];
        sy = Y + g.p[5];
        break;
    }
    case GEOM_ROTATE: { // p = {cos, sin, rotated_center_x, rotated_center_y, center_x, center_y}
        const float dx = x - g.p[2], dy = y - g.p[3];
        const float rdx = g.p[0] * dx - g.p[1] * dy;
        const float rdy = g.p[1] * dx + g.p[0] * dy;
        sx = rdx + g.p[4];
        sy = rdy + g.p[5];
        break;
    }
    default: { // extract: p = {cos, sin, rect.l, rect.t, width, height, cx, cy, fcols - 1, frows - 1, cols == 1, rows == 1}
        const float ty = g.p[11] != 0.0f ? 0.5f : y / g.p[9];
        const float y_rect = g.p[3] + ty * g.p[5];
        const float tx = g.p[10] != 0.0f ? 0.5f : x / g.p[8];
        const float x_rect = g.p[2] + tx * g.p[4];
        const float dx = x_rect - g.p[6], dy = y_rect - g.p[7];
        sx = g.p[6] + g.p[0] * dx - g.p[1] * dy;
        sy = g.p[7] + g.p[1] * dx + g.p[0] * dy;
        break;
    }
    }
}

template <int PIX, int KIND, bool WAVE_STAGE = false>
__global__ __launch_bounds__(256) void k_geom(DImg src, DImg dst, GeomParams g, MethodArg m, int border, int tiles_x, FrameSpan fr) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    // 64 x 4 destination tile per workgroup; workgroups numbered XCD-major so one XCD's L2 serves a
    // contiguous band of destination (hence, for smooth maps, of source) rows.
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    src.data = (char *)src.data + (size_t)blockIdx.y * fr.src_frame; // a batch of equally shaped frames, the same map for each (zg_batch_pipeline)
    dst.data = (char *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int c = tx * 64 + (int)(threadIdx.x & 63);
    const int r = ty * 4 + (int)(threadIdx.x >> 6);
    if (c >= dst.cols || r >= dst.rows) return;
    float sx, sy;
    source_coord(g, c, r, sx, sy);
    Vec v;
    // WAVE_STAGE (Rgba(f32) through a radius-2 kernel): each wave stages the source pixels its 64 windows share (zg_sample.h)
    __shared__ Vec stage_mem[WAVE_STAGE ? 4 * WAVE_STAGE_ROWS * 64 : 1];
    if (!interpolate<PIX, KIND, WAVE_STAGE>(src, sx, sy, m, border, v, stage_mem + (WAVE_STAGE ? (threadIdx.x >> 6) * (WAVE_STAGE_ROWS * 64) : 0))) v = P::zero();
    P::store(dst.data, (size_t)r * dst.stride + (size_t)c, v);
}

// ---- Lanczos table on the device ---------------------------------------------------------------
struct LutHolder {
    const float *dev = nullptr;   // what the kernel reads
    float *owned = nullptr;       // per-call upload of a caller-supplied table (freed on the stream)
};
static int device_lanczos_lut(const zg_method *method, hipStream_t s, LutHolder &h) {
    if (method->kind != ZG_INTERP_LANCZOS) return ZG_OK;
    if (method->lanczos_lut) {
        if (int rc = scratch_alloc((void **)&h.owned, 1025 * sizeof(float), s)) return rc;
        if (int rc = upload_pageable(h.owned, method->lanczos_lut, 1025 * sizeof(float), s)) return rc; // the caller's table may be pageable / short-lived
        h.dev = h.owned;
        return ZG_OK;
    }
    static std::mutex mu;
    static float *per_device[64] = {nullptr};
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && !per_device[dev]) {
        float *p = nullptr;
        ZG_HIP(hipMalloc((void **)&p, 1025 * sizeof(float)));
        if (int rc = upload_pageable(p, hostmath::lanczos3_lut(), 1025 * sizeof(float), nullptr)) return rc;
        per_device[dev] = p;
    }
    h.dev = per_device[dev];
    return ZG_OK;
}
static void release_lut(LutHolder &h, hipStream_t s) {
    if (h.owned) scratch_free(h.owned, s);
    h.owned = nullptr;
}

static int check_method(const zg_method *method) {
    ZG_REQUIRE(method != nullptr, ZG_ERR_INVALID_ARGUMENT, "null interpolation method");
    ZG_REQUIRE(method->kind >= ZG_INTERP_NEAREST && method->kind <= ZG_INTERP_LANCZOS, ZG_ERR_INVALID_ARGUMENT,
               "invalid interpolation kind %d", method->kind);
    return ZG_OK;
}

struct FrameBatch { // n frames src_frame / dst_frame bytes apart (defaults: the one image)
    uint32_t n = 1;
    size_t src_frame = 0, dst_frame = 0;
};
template <int PIX, int KIND>
static int launch_geom_k(const zg_image *src, const zg_image *dst, const GeomParams &g, const MethodArg &m, int border, const FrameBatch &fb, hipStream_t s) {
    const int tiles_x = (int)ceil_div(dst->cols, 64), tiles_y = (int)ceil_div(dst->rows, 4);
    const uint64_t tiles = (uint64_t)tiles_x * tiles_y;
    ZG_REQUIRE(tiles <= 0x7fffffffu, ZG_ERR_INVALID_ARGUMENT, "too many tiles in one launch (%llu)", (unsigned long long)tiles);
    if (fb.n > MAX_FRAMES_PER_LAUNCH) return -1; // the caller goes frame by frame
    const dim3 grid((unsigned)tiles, fb.n);
    const FrameSpan fr{fb.src_frame, fb.dst_frame};
    constexpr bool CAN_STAGE = PIX == ZG_PIXEL_RGBA_F32 && (KIND == ZG_INTERP_BICUBIC || KIND == ZG_INTERP_CATMULL_ROM || KIND == ZG_INTERP_MITCHELL);
    if constexpr (CAN_STAGE) {
        // for every map: where a wave cannot stage, this kernel's row-at-a-time gather (seven waves per SIMD) still beats the sixteen gathers
        // in flight of the plain one — 2:1 reduction 101 -> 80 us, a 10-degree rotation 350 -> 304 us (profiles/r03_experiments.txt)
        hipLaunchKernelGGL((k_geom<PIX, KIND, true>), grid, dim3(256), 0, s, dimg(src), dimg(dst), g, m, border, tiles_x, fr);
    } else {
        hipLaunchKernelGGL((k_geom<PIX, KIND>), grid, dim3(256), 0, s, dimg(src), dimg(dst), g, m, border, tiles_x, fr);
    }
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int launch_geom(const zg_image *src, const zg_image *dst, const GeomParams &g, const zg_method *method, int border, hipStream_t s, const FrameBatch &fb = FrameBatch{}) {
    int rc;
    if ((rc = check_method(method))) return rc;
    ZG_REQUIRE(border >= ZG_BORDER_ZERO && border <= ZG_BORDER_WRAP, ZG_ERR_INVALID_ARGUMENT, "invalid border %d", border);
    if (dst->rows == 0 || dst->cols == 0) return ZG_OK;
    LutHolder lut;
    if ((rc = device_lanczos_lut(method, s, lut))) return rc;
    const MethodArg m{method->kind, method->b, method->c, lut.dev};
    GeomParams gp = g;
    gp.stage = 1;
    rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        switch (method->kind) {
        case ZG_INTERP_NEAREST: return launch_geom_k<PIX, ZG_INTERP_NEAREST>(src, dst, gp, m, border, fb, s);
        case ZG_INTERP_BILINEAR: return launch_geom_k<PIX, ZG_INTERP_BILINEAR>(src, dst, gp, m, border, fb, s);
        case ZG_INTERP_BICUBIC: return launch_geom_k<PIX, ZG_INTERP_BICUBIC>(src, dst, gp, m, border, fb, s);
        case ZG_INTERP_CATMULL_ROM: return launch_geom_k<PIX, ZG_INTERP_CATMULL_ROM>(src, dst, gp, m, border, fb, s);
        case ZG_INTERP_MITCHELL: return launch_geom_k<PIX, ZG_INTERP_MITCHELL>(src, dst, gp, m, border, fb, s);
        default: return launch_geom_k<PIX, ZG_INTERP_LANCZOS>(src, dst, gp, m, border, fb, s);
        }
    });
    release_lut(lut, s);
    return rc;
}

static int check_pair(const zg_image *src, const zg_image *dst, const char *op) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "%s: pixel types differ", op);
    return ZG_OK;
}

// ---- Image(u8).resize(.bilinear): a grey plane (what ImagePyramid resizes, src/image/pyramid.zig:88-99) -------------------------------
// The reference sends a scalar u8 image through the generic sampler (interpolation.zig:313-407: f32 coordinates, fractions rounded
// to 8 bits, integer lerp; resize hands it .mirror, :171). k_geom<0, bilinear> spends ~115 VALU instructions per pixel on
// generality: a switch on the map, the 64-bit index fall-back of far-away coordinates, four byte loads behind four index
// resolutions. A resize's coordinates stay within half a pixel of the image, so the only neighbours that can leave it are -1 and
// `length`, the 32-bit resolveIndex covers them, and it is not needed at all for a pixel whose neighbours are inside — every pixel
// when shrinking. A lane owns FOUR consecutive pixels of a row; the row taps are the wave's (scalar). Same expressions as the
// generic path, in the same order. Same XCD-major tile order and frame index as k_geom.
// (Staging the wave's two source rows in LDS instead of gathering bytes through L1 was built and measured: slower, 335 against
// 312 us for the pyramid, profiles/r03_experiments.txt.)
__global__ __launch_bounds__(256) void k_resize_bilinear_u8(DImg src, DImg dst, float rx, float ry, int tiles_x, FrameSpan fr, int dword_rows) {
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    src.data = (char *)src.data + (size_t)blockIdx.y * fr.src_frame;
    dst.data = (char *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int c0 = tx * 256 + (int)(threadIdx.x & 63) * 4;
    const int r = __builtin_amdgcn_readfirstlane(ty * 4 + (int)(threadIdx.x >> 6));
    if (r >= dst.rows || c0 >= dst.cols) return;
    const float sy = ((float)r + 0.5f) * ry - 0.5f; // source_coord's GEOM_RESIZE, the same expression
    const float ft = floorf(sy);
    const int top = (int)ft;
    int r0 = top, r1 = top + 1;
    if (top < 0 || top + 1 >= src.rows) { // wave-uniform
        r0 = resolve_index(top, src.rows, ZG_BORDER_MIRROR);
        r1 = resolve_index(top + 1, src.rows, ZG_BORDER_MIRROR);
    }
    const int fy = (int)roundf((sy - ft) * 256);
    const uint8_t *row0 = (const uint8_t *)src.data + (size_t)r0 * src.stride, *row1 = (const uint8_t *)src.data + (size_t)r1 * src.stride;
    // Loads are what bounds this kernel (every gather instruction is a full pass through the texture-address unit, whatever its width),
    // so a row's taps of TWO neighbouring pixels come from one 8-byte load where they fit (both inside the row, at most 6 columns apart,
    // 8 readable bytes left in the row) and from 2-byte / 1-byte loads otherwise. Global memory takes any alignment.
    uint32_t packed = 0;
    const int n = dst.cols - c0 < 4 ? dst.cols - c0 : 4;
    float sx[4], fl[4];
    int left[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        sx[p] = ((float)(c0 + p) + 0.5f) * rx - 0.5f;
        fl[p] = floorf(sx[p]);
        left[p] = (int)fl[p];
    }
    int tl[4], tr[4], bl[4], br[4];
#pragma unroll
    for (int p = 0; p < 4; p += 2) {
        const int d = left[p + 1] - left[p];
        if (p + 1 < n && left[p] >= 0 && d >= 0 && d <= 6 && left[p] + 8 <= src.cols) {
            uint64_t a, b;
            __builtin_memcpy(&a, row0 + left[p], 8);
            __builtin_memcpy(&b, row1 + left[p], 8);
            const uint32_t a1 = (uint32_t)(a >> (8 * d)), b1 = (uint32_t)(b >> (8 * d));
            tl[p] = (int)(a & 255); tr[p] = (int)((a >> 8) & 255); bl[p] = (int)(b & 255); br[p] = (int)((b >> 8) & 255);
            tl[p + 1] = (int)(a1 & 255); tr[p + 1] = (int)((a1 >> 8) & 255); bl[p + 1] = (int)(b1 & 255); br[p + 1] = (int)((b1 >> 8) & 255);
        } else {
#pragma unroll
            for (int q = p; q < p + 2; ++q) {
                tl[q] = tr[q] = bl[q] = br[q] = 0;
                if (q < n) {
                    int cl = left[q], cr = left[q] + 1;
                    if (left[q] < 0 || left[q] + 1 >= src.cols) {
                        cl = resolve_index(left[q], src.cols, ZG_BORDER_MIRROR);
                        cr = resolve_index(left[q] + 1, src.cols, ZG_BORDER_MIRROR);
                    }
                    if (cr == cl + 1) {
                        uint16_t a, b;
                        __builtin_memcpy(&a, row0 + cl, 2);
                        __builtin_memcpy(&b, row1 + cl, 2);
                        tl[q] = a & 255; tr[q] = a >> 8; bl[q] = b & 255; br[q] = b >> 8;
                    } else {
                        tl[q] = row0[cl]; tr[q] = row0[cr]; bl[q] = row1[cl]; br[q] = row1[cr];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int fx = (int)roundf((sx[p] - fl[p]) * 256);
        const int top_val = tl[p] * (256 - fx) + tr[p] * fx;
        const int bottom_val = bl[p] * (256 - fx) + br[p] * fx;
        const uint32_t v = (uint32_t)((top_val * (256 - fy) + bottom_val * fy + 32768) >> 16); // < 256: a convex combination of bytes
        packed |= v << (8 * p);
    }
    uint8_t *o = (uint8_t *)dst.data + (size_t)r * dst.stride + (size_t)c0;
    if (n == 4 && dword_rows) *(uint32_t *)o = packed;
    else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p < n) o[p] = (uint8_t)(packed >> (8 * p));
    }
}

// The same resize for horizontal ratios below 2 (what ImagePyramid's first levels and most thumbnails-of-thumbnails ask for), R output rows per wave: the
// taps of a lane's four columns depend on the column only, so they are computed once and used for R rows (14 of the 45 instructions a pixel of the kernel
// above costs), and with a ratio below 2 the four pixels' taps span at most 8 source bytes: ONE unaligned 8-byte load per source row and lane. A pixel's
// two bytes of a row come out of the pair of dwords by v_perm_b32 with a per-lane selector, as the u16 pair (left, right); the horizontal lerp is a
// v_dot2_u32_u16 against (256 - fx, fx), the vertical one two v_mad_u32_u24: 7 instructions a pixel. Lanes whose taps leave the row (the image's edges)
// take the byte-by-byte form. Integer arithmetic, the same expressions: bit-identical to the kernel above.
template <int R>
__global__ __launch_bounds__(256) void k_resize_bilinear_u8_rows(DImg src, DImg dst, float rx, float ry, int tiles_x, FrameSpan fr, int dword_rows) {
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    src.data = (char *)src.data + (size_t)blockIdx.y * fr.src_frame;
    dst.data = (char *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int c0 = tx * 256 + (int)(threadIdx.x & 63) * 4;
    const int rbase = __builtin_amdgcn_readfirstlane((ty * 4 + (int)(threadIdx.x >> 6)) * R);
    if (rbase >= dst.rows || c0 >= dst.cols) return;
    const int n = dst.cols - c0 < 4 ? dst.cols - c0 : 4;
    int left[4], fx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float sx = ((float)(c0 + p) + 0.5f) * rx - 0.5f;
        const float fl = floorf(sx);
        left[p] = (int)fl;
        fx[p] = (int)roundf((sx - fl) * 256);
    }
    // every tap of the four pixels inside the row, within 8 bytes of the first, and 8 readable bytes there
    const bool fast = n == 4 && left[0] >= 0 && left[3] + 2 <= src.cols && left[3] - left[0] <= 6 && left[1] >= left[0] && left[2] >= left[1] && left[3] >= left[2] &&
                      left[0] + 8 <= src.cols;
    uint32_t sel[4], wpair[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t o = (uint32_t)(left[p] - left[0]);
        sel[p] = o * 0x00010001u + 0x0c010c00u;          // bytes (o, zero, o + 1, zero): the u16 pair (left tap, right tap)
        wpair[p] = (uint32_t)fx[p] * 65535u + 256u;       // (256 - fx) | fx << 16
    }
    auto row_taps = [&](int r, int &r0, int &r1, int &fy) { // wave-uniform
        const float sy = ((float)r + 0.5f) * ry - 0.5f;
        const float ft = floorf(sy);
        const int top = (int)ft;
        r0 = top;
        r1 = top + 1;
        if (top < 0 || top + 1 >= src.rows) {
            r0 = resolve_index(top, src.rows, ZG_BORDER_MIRROR);
            r1 = resolve_index(top + 1, src.rows, ZG_BORDER_MIRROR);
        }
        fy = (int)roundf((sy - ft) * 256);
    };
    auto store = [&](int r, uint32_t packed) {
        uint8_t *o = (uint8_t *)dst.data + (size_t)r * dst.stride + (size_t)c0;
        if (n == 4 && dword_rows) *(uint32_t *)o = packed;
        else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p < n) o[p] = (uint8_t)(packed >> (8 * p));
        }
    };
    if (fast) {
        // every load of the wave's R rows first, then the arithmetic: a loop of load - wait - use leaves the memory's latency in the open once per row
        // (rows past the image's last are clamped, loaded and dropped)
        uint32_t a[R][2], b[R][2];
        int fys[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            int r0, r1;
            row_taps(min(rbase + rr, dst.rows - 1), r0, r1, fys[rr]);
            __builtin_memcpy(a[rr], (const uint8_t *)src.data + (size_t)r0 * src.stride + left[0], 8);
            __builtin_memcpy(b[rr], (const uint8_t *)src.data + (size_t)r1 * src.stride + left[0], 8);
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int r = rbase + rr; // wave-uniform
            if (r >= dst.rows) break;
            const int fy = fys[rr];
            uint32_t v[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t tp = __builtin_amdgcn_perm(a[rr][1], a[rr][0], sel[p]), bp = __builtin_amdgcn_perm(b[rr][1], b[rr][0], sel[p]);
                uint32_t top_val, bottom_val;
                asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(top_val) : "v"(tp), "v"(wpair[p]));
                asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(bottom_val) : "v"(bp), "v"(wpair[p]));
                v[p] = top_val * (uint32_t)(256 - fy) + bottom_val * (uint32_t)fy + 32768u; // < 2^24 + 2^15; the pixel is byte 2
            }
            store(r, __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu));
        }
        return;
    }
    for (int rr = 0; rr < R; ++rr) {
        const int r = rbase + rr; // wave-uniform
        if (r >= dst.rows) break;
        int r0, r1, fy;
        row_taps(r, r0, r1, fy);
        const uint8_t *row0 = (const uint8_t *)src.data + (size_t)r0 * src.stride, *row1 = (const uint8_t *)src.data + (size_t)r1 * src.stride;
        uint32_t packed = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p >= n) break;
            int cl = left[p], cr = left[p] + 1;
            if (left[p] < 0 || left[p] + 1 >= src.cols) {
                cl = resolve_index(left[p], src.cols, ZG_BORDER_MIRROR);
                cr = resolve_index(left[p] + 1, src.cols, ZG_BORDER_MIRROR);
            }
            const int tl = row0[cl], tr = row0[cr], bl = row1[cl], br = row1[cr];
            const int top_val = tl * (256 - fx[p]) + tr * fx[p];
            const int bottom_val = bl * (256 - fx[p]) + br * fx[p];
            packed |= (uint32_t)((top_val * (256 - fy) + bottom_val * fy + 32768) >> 16) << (8 * p);
        }
        store(r, packed);
    }
}

// n equally shaped u8 planes (n = 1: the one image); sizes the caller has already checked
static int launch_resize_bilinear_u8(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    const int tiles_x = (int)ceil_div(dst->cols, 256), tiles_y = (int)ceil_div(dst->rows, 4);
    const uint64_t tiles = (uint64_t)tiles_x * tiles_y;
    ZG_REQUIRE(tiles <= 0x7fffffffu, ZG_ERR_INVALID_ARGUMENT, "too many tiles in one launch (%llu)", (unsigned long long)tiles);
    if (n > MAX_FRAMES_PER_LAUNCH) return -1; // the caller goes frame by frame
    const FrameSpan fr{src_frame, dst_frame};
    const int dword_rows = ((uintptr_t)dst->data % 4 == 0 && dst->stride % 4 == 0 && dst_frame % 4 == 0) ? 1 : 0;
    const float rx = (float)src->cols / (float)dst->cols;
    static const int rows_form = getenv("ZIGNAL_HIP_RESIZE_U8_ROWS") ? atoi(getenv("ZIGNAL_HIP_RESIZE_U8_ROWS")) : 2; // A/B hook of round 5: 0 = the one-row kernel; 4096^2 -> 3413^2: 21.3 / 18.2 / 20.4 us for 0 / 2 / 4 rows
    if (rows_form > 0 && rx < 2.0f && src->cols >= 8) { // taps computed once for several rows, one 8-byte load per source row and lane
        const int R = rows_form >= 8 ? 8 : rows_form >= 4 ? 4 : 2;
        const uint64_t tiles_r = (uint64_t)tiles_x * ceil_div(dst->rows, (uint32_t)(4 * R));
        if (R == 8) hipLaunchKernelGGL((k_resize_bilinear_u8_rows<8>), dim3((unsigned)tiles_r, n), dim3(256), 0, s, dimg(src), dimg(dst), rx, (float)src->rows / (float)dst->rows, tiles_x, fr, dword_rows);
        else if (R == 4) hipLaunchKernelGGL((k_resize_bilinear_u8_rows<4>), dim3((unsigned)tiles_r, n), dim3(256), 0, s, dimg(src), dimg(dst), rx, (float)src->rows / (float)dst->rows, tiles_x, fr, dword_rows);
        else hipLaunchKernelGGL((k_resize_bilinear_u8_rows<2>), dim3((unsigned)tiles_r, n), dim3(256), 0, s, dimg(src), dimg(dst), rx, (float)src->rows / (float)dst->rows, tiles_x, fr, dword_rows);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    hipLaunchKernelGGL(k_resize_bilinear_u8, dim3((unsigned)tiles, n), dim3(256), 0, s, dimg(src), dimg(dst), rx, (float)src->rows / (float)dst->rows, tiles_x, fr, dword_rows);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}
static bool resize_u8_plane_applies(const zg_image *src, const zg_image *dst, const zg_method *method) {
    return src->pixel == ZG_PIXEL_U8 && method->kind == ZG_INTERP_BILINEAR && src->rows > 0 && src->cols > 0 && dst->rows > 0 && dst->cols > 0;
}

// ---- resize (interpolation.zig:89-191) -----------------------------------------------------------
int resize_impl(const zg_image *src, const zg_image *dst, const zg_method *method, hipStream_t s) {
    int rc;
    if ((rc = check_pair(src, dst, "resize")) || (rc = check_method(method))) return rc;
    if (dst->rows == 0 || dst->cols == 0) return ZG_OK;
    if (src->rows == dst->rows && src->cols == dst->cols) return copy_impl(src, dst, s); // :91-108
    const bool is_rgb_u8 = src->pixel == ZG_PIXEL_RGB_U8 || src->pixel == ZG_PIXEL_RGBA_U8; // meta.isRgb(T)
    if (is_rgb_u8 && src->rows > 0 && src->cols > 0) return resize_planes_impl(src, dst, method, s);
    if (resize_u8_plane_applies(src, dst, method)) return launch_resize_bilinear_u8(src, dst, 1, 0, 0, s);
    GeomParams g{};
    g.mode = GEOM_RESIZE;
    g.p[0] = (float)src->cols / (float)dst->cols;
    g.p[1] = (float)src->rows / (float)dst->rows;
    return launch_geom(src, dst, g, method, ZG_BORDER_MIRROR, s);
}

// resize of n equally shaped frames in one launch where the path is a single kernel with a frame index (the generic interpolators of every
// type, and the Rgb(u8) / Rgba(u8) plane resizers); -1 otherwise (zg_batch_pipeline then goes frame by frame).
int resize_frames(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    int rc;
    if ((rc = check_pair(src, dst, "resize")) || (rc = check_method(method))) return rc;
    if (n == 0 || dst->rows == 0 || dst->cols == 0 || src->rows == 0 || src->cols == 0) return -1;
    if (src->rows == dst->rows && src->cols == dst->cols) return -1; // a copy per frame
    const bool is_rgb_u8 = src->pixel == ZG_PIXEL_RGB_U8 || src->pixel == ZG_PIXEL_RGBA_U8;
    if (is_rgb_u8) return resize_planes_frames(src, dst, method, n, src_frame, dst_frame, s); // every method: the frame is the grid's y
    if (resize_u8_plane_applies(src, dst, method)) return launch_resize_bilinear_u8(src, dst, n, src_frame, dst_frame, s);
    GeomParams g{};
    g.mode = GEOM_RESIZE;
    g.p[0] = (float)src->cols / (float)dst->cols;
    g.p[1] = (float)src->rows / (float)dst->rows;
    FrameBatch fb;
    fb.n = n; fb.src_frame = src_frame; fb.dst_frame = dst_frame;
    return launch_geom(src, dst, g, method, ZG_BORDER_MIRROR, s, fb);
}

// Image(Rgb(u8) / Rgba(u8)).resize(.lanczos) with the caller's plane weights (channel_ops.zig:438-493)
static int resize_lanczos_weights_checked(const zg_image *src, const zg_image *dst, const float *wx, const float *wy, hipStream_t s) {
    int rc;
    if ((rc = check_pair(src, dst, "resize"))) return rc;
    ZG_REQUIRE(src->pixel == ZG_PIXEL_RGB_U8 || src->pixel == ZG_PIXEL_RGBA_U8, ZG_ERR_UNSUPPORTED,
               "resize with Lanczos plane weights is the Rgb(u8) / Rgba(u8) path; other pixel types interpolate through zg_method.lanczos_lut");
    if (dst->rows == 0 || dst->cols == 0) return ZG_OK;
    if (src->rows == dst->rows && src->cols == dst->cols) return copy_impl(src, dst, s); // interpolation.zig:91-108
    if (src->rows == 0 || src->cols == 0) { // no source pixel to resolve an index to: what resize_impl does
        const zg_method m{ZG_INTERP_LANCZOS, 0, 0, nullptr};
        return resize_impl(src, dst, &m, s);
    }
    return resize_lanczos_weights_impl(src, dst, wx, wy, s);
}

// ---- letterbox (transforms.zig:49-108) -------------------------------------------------------------
static int letterbox_impl(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t rect_out[4], hipStream_t s) {
    int rc;
    if ((rc = check_pair(src, dst, "letterbox")) || (rc = check_method(method))) return rc;
    uint32_t rect[4] = {0, 0, 0, 0};
    const uint8_t zero[16] = {0};
    auto done = [&](int status) {
        if (rect_out) std::memcpy(rect_out, rect, sizeof rect);
        return status;
    };
    if (dst->rows == 0 || dst->cols == 0) return done(ZG_OK);
    if (src->rows == 0 || src->cols == 0) return done(fill_outside_impl(dst, zero, 0, 0, 0, 0, s));
    if (src->rows == dst->rows && src->cols == dst->cols) {
        rect[2] = dst->cols; rect[3] = dst->rows;
        return done(copy_impl(src, dst, s));
    }
    const float rows_scale = (float)dst->rows / (float)src->rows, cols_scale = (float)dst->cols / (float)src->cols;
    if (rows_scale == cols_scale) {
        rect[2] = dst->cols; rect[3] = dst->rows;
        return done(resize_impl(src, dst, method, s));
    }
    const float aspect = std::fmin(rows_scale, cols_scale);
    const uint32_t scaled_rows = (uint32_t)std::round(aspect * (float)src->rows);
    const uint32_t scaled_cols = (uint32_t)std::round(aspect * (float)src->cols);
    const uint32_t off_r = (dst->rows > scaled_rows ? dst->rows - scaled_rows : 0) / 2;
    const uint32_t off_c = (dst->cols > scaled_cols ? dst->cols - scaled_cols : 0) / 2;
    rect[0] = off_c; rect[1] = off_r; rect[2] = off_c + scaled_cols; rect[3] = off_r + scaled_rows;
    const uint32_t l = rect[0], t = rect[1], r = std::min(rect[2], dst->cols), b = std::min(rect[3], dst->rows);
    if (l < r && t < b) { // out.view(content_rect)
        zg_image view = *dst;
        view.rows = b - t;
        view.cols = r - l;
        view.data = (char *)dst->data + ((size_t)t * dst->stride + l) * pixel_size(dst->pixel);
        if ((rc = resize_impl(src, &view, method, s))) return done(rc);
    }
    return done(set_border_impl(dst, rect, zero, s));
}

// ---- warp (transforms.zig:522-531) ------------------------------------------------------------------
static int warp_impl(const zg_image *src, const zg_image *dst, int kind, const float *mat, const zg_method *method, hipStream_t s, const FrameBatch &fb = FrameBatch{}) {
    int rc;
    if ((rc = check_pair(src, dst, "warp"))) return rc;
    ZG_REQUIRE(mat != nullptr, ZG_ERR_INVALID_ARGUMENT, "warp: null transform coefficients");
    ZG_REQUIRE(kind >= ZG_TRANSFORM_SIMILARITY && kind <= ZG_TRANSFORM_PROJECTIVE, ZG_ERR_INVALID_ARGUMENT, "warp: invalid transform kind %d", kind);
    GeomParams g{};
    if (kind == ZG_TRANSFORM_PROJECTIVE) {
        g.mode = GEOM_PROJECTIVE;
        for (int i = 0; i < 9; ++i) g.p[i] = mat[i];
    } else {
        g.mode = GEOM_AFFINE;
        for (int i = 0; i < 6; ++i) g.p[i] = mat[i];
    }
    return launch_geom(src, dst, g, method, ZG_BORDER_MIRROR, s, fb);
}
// the same warp applied to n equally shaped frames in one launch
int warp_frames(const zg_image *src, const zg_image *dst, int kind, const float *mat, const zg_method *method, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    FrameBatch fb;
    fb.n = n; fb.src_frame = src_frame; fb.dst_frame = dst_frame;
    return warp_impl(src, dst, kind, mat, method, s, fb);
}

// ---- rotate (transforms.zig:112-212, :385-462) -----------------------------------------------------
static float mod_tau(float a) { // @mod(angle, tau): floored
    const float tau = 6.28318530717958647692f;
    float r = std::fmod(a, tau);
    if (r < 0) r += tau;
    return r;
}
static int orthogonal_case(float angle) {
    const float n = mod_tau(angle), eps = 1e-6f, pi = 3.14159265358979323846f, tau = 6.28318530717958647692f;
    if (std::fabs(n) < eps || std::fabs(n - tau) < eps) return 0;
    if (std::fabs(n - pi / 2.0f) < eps) return 1;
    if (std::fabs(n - pi) < eps) return 2;
    if (std::fabs(n - 3.0f * pi / 2.0f) < eps) return 3;
    return -1;
}

// Exact index permutation: out[new_r + off_r, new_c + off_c] = src[r, c]; one thread per source pixel.
template <int PS>
__global__ __launch_bounds__(256) void k_rotate_orthogonal(DImg src, DImg out, int which, int off_r, int off_c) {
    struct B { uint8_t b[PS]; };
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    int nr, nc;
    switch (which) {
    case 0: nr = r; nc = c; break;
    case 1: nr = src.cols - 1 - c; nc = r; break;
    case 2: nr = src.rows - 1 - r; nc = src.cols - 1 - c; break;
    default: nr = c; nc = src.rows - 1 - r; break;
    }
    nr += off_r;
    nc += off_c;
    if (nr < out.rows && nc < out.cols)
        ((B *)out.data)[(size_t)nr * out.stride + nc] = ((const B *)src.data)[(size_t)r * src.stride + c];
}

static int rotate_into_impl(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a,
                            const zg_method *method, int border, hipStream_t s) {
    int rc;
    if ((rc = check_pair(src, dst, "rotateInto")) || (rc = check_method(method))) return rc;
    const int oc = orthogonal_case(angle);
    if (oc >= 0) {
        const uint32_t rr = (oc & 1) ? src->cols : src->rows, rcols = (oc & 1) ? src->rows : src->cols;
        const uint32_t off_r = (dst->rows > rr ? dst->rows - rr : 0) / 2, off_c = (dst->cols > rcols ? dst->cols - rcols : 0) / 2;
        if (src->rows && src->cols) {
            const dim3 grid = row_grid(ceil_div(src->cols, 256), src->rows);
#define ZG_ROT(PS) case PS: hipLaunchKernelGGL(k_rotate_orthogonal<PS>, grid, dim3(256), 0, s, dimg(src), dimg(dst), oc, (int)off_r, (int)off_c); break;
            switch ((int)pixel_size(src->pixel)) { ZG_ROT(1) ZG_ROT(3) ZG_ROT(4) ZG_ROT(12) ZG_ROT(16) }
#undef ZG_ROT
            ZG_HIP(hipGetLastError());
        }
        if (off_r != 0 || off_c != 0) {
            const uint32_t inner[4] = {off_c, off_r, off_c + rcols, off_r + rr};
            const uint8_t zero[16] = {0};
            return set_border_impl(dst, inner, zero, s);
        }
        return ZG_OK;
    }
    const float cx = (float)src->cols / 2.0f, cy = (float)src->rows / 2.0f; // getCenter
    const float offset_x = ((float)dst->cols - (float)src->cols) / 2.0f;
    const float offset_y = ((float)dst->rows - (float)src->rows) / 2.0f;
    GeomParams g{};
    g.mode = GEOM_ROTATE;
    g.p[0] = cos_a; g.p[1] = sin_a; g.p[2] = cx + offset_x; g.p[3] = cy + offset_y; g.p[4] = cx; g.p[5] = cy;
    return launch_geom(src, dst, g, method, border, s);
}

// ---- extract / crop (transforms.zig:216-282, copyRect :465-518) ---------------------------------------
static float rect_w(const float r[4]) { return r[0] >= r[2] ? 0.0f : r[2] - r[0]; }
static float rect_h(const float r[4]) { return r[1] >= r[3] ? 0.0f : r[3] - r[1]; }

template <int PS>
__global__ __launch_bounds__(256) void k_copy_rect(DImg src, DImg out, int rect_top, int rect_left, int border) {
    struct B { uint8_t b[PS]; };
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= out.cols || r >= out.rows) return;
    const int rr = resolve_index(r + rect_top, src.rows, border);
    const int cc = rr < 0 ? -1 : resolve_index(c + rect_left, src.cols, border);
    B v = {};
    if (rr >= 0 && cc >= 0) v = ((const B *)src.data)[(size_t)rr * src.stride + cc];
    ((B *)out.data)[(size_t)r * out.stride + c] = v;
}

static int copy_rect_impl(const zg_image *src, int rect_top, int rect_left, const zg_image *out, int border, hipStream_t s) {
    if (out->rows == 0 || out->cols == 0) return ZG_OK;
    const dim3 grid = row_grid(ceil_div(out->cols, 256), out->rows);
#define ZG_CR(PS) case PS: hipLaunchKernelGGL(k_copy_rect<PS>, grid, dim3(256), 0, s, dimg(src), dimg(out), rect_top, rect_left, border); break;
    switch ((int)pixel_size(src->pixel)) { ZG_CR(1) ZG_CR(3) ZG_CR(4) ZG_CR(12) ZG_CR(16) }
#undef ZG_CR
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int extract_impl(const zg_image *src, const zg_image *dst, const float rect[4], float angle, float cos_a, float sin_a,
                        const zg_method *method, int border, hipStream_t s) {
    int rc;
    if ((rc = check_pair(src, dst, "extract")) || (rc = check_method(method))) return rc;
    ZG_REQUIRE(rect != nullptr, ZG_ERR_INVALID_ARGUMENT, "extract: null rect");
    if (dst->rows == 0 || dst->cols == 0) return ZG_OK;
    const float frows = (float)dst->rows, fcols = (float)dst->cols;
    const float width = rect_w(rect), height = rect_h(rect), eps = 1e-6f;
    if (std::fabs(angle) < eps && std::fabs(width - fcols) < eps && std::fabs(height - frows) < eps)
        return copy_rect_impl(src, (int)std::round(rect[1]), (int)std::round(rect[0]), dst, border, s);
    GeomParams g{};
    g.mode = GEOM_EXTRACT;
    g.p[0] = cos_a; g.p[1] = sin_a; g.p[2] = rect[0]; g.p[3] = rect[1]; g.p[4] = width; g.p[5] = height;
    g.p[6] = (rect[0] + rect[2]) * 0.5f; g.p[7] = (rect[1] + rect[3]) * 0.5f;
    g.p[8] = fcols - 1; g.p[9] = frows - 1;
    g.p[10] = dst->cols == 1 ? 1.0f : 0.0f; g.p[11] = dst->rows == 1 ? 1.0f : 0.0f;
    return launch_geom(src, dst, g, method, border, s);
}

// ---- insert (transforms.zig:293-378) -------------------------------------------------------------------
struct InsertParams {
    int aligned;             // fast path: axis aligned, no resampling
    int dst_top, dst_left;
    int min_r, max_r, min_c, max_c;
    float cx, cy, cos_a, sin_a, inv_width, inv_height, half_width, half_height, fcols_m1, frows_m1;
    int blend;
    uint8_t *mask; // mixed-type insert: one byte per pixel of the bounding box, set where a sample was produced (else null)
    int mask_w;
};

// Rgba(u8).blend(overlay, mode) — blendColors, reference src/blending.zig:27-157; mode = the Blending ordinal (none 0 ...
// exclusion 12). f32 per channel in the reference's operation order (its vector products are left-associative).
__device__ inline float blend_channel(int mode, float b, float o) {
    switch (mode) {
    case 1: return o;
    case 2: return b * o;
    case 3: return 1.0f - (1.0f - b) * (1.0f - o);
    case 4: return b < 0.5f ? (2.0f * b) * o : 1.0f - (2.0f * (1.0f - b)) * (1.0f - o);
    case 5: return o <= 0.5f ? b - ((1.0f - 2.0f * o) * b) * (1.0f - b) : b + (2.0f * o - 1.0f) * (sqrtf(b) - b);
    case 6: return o < 0.5f ? (2.0f * o) * b : 1.0f - (2.0f * (1.0f - o)) * (1.0f - b);
    case 7: { const float r = b / (1.0f - o); return b == 0.0f ? 0.0f : (o >= 1.0f ? 1.0f : fminf(1.0f, r)); }
    case 8: { const float r = 1.0f - (1.0f - b) / o; return b >= 1.0f ? 1.0f : (o <= 0.0f ? 0.0f : fmaxf(0.0f, r)); }
    case 9: return fminf(b, o);
    case 10: return fmaxf(b, o);
    case 11: return fabsf(b - o);
    case 12: return (b + o) - (2.0f * b) * o;
    }
    return o;
}
__device__ inline void blend_u8(typename Px<ZG_PIXEL_RGBA_U8>::Vec &base, typename Px<ZG_PIXEL_RGBA_U8>::Vec overlay, int mode) {
    if (overlay[3] == 0) return;
    if (base[3] == 0 || (mode == 1 && overlay[3] == 255)) { base = overlay; return; }
    float b[4], o[4], out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = (float)base[i] / 255.0f; o[i] = (float)overlay[i] / 255.0f; }
    float blended[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) blended[i] = blend_channel(mode, b[i], o[i]);
    if (overlay[3] == 255) {
        out[0] = blended[0]; out[1] = blended[1]; out[2] = blended[2]; out[3] = 1.0f;
    } else {
        const float result_a = o[3] + b[3] * (1.0f - o[3]);
        if (result_a <= 0) { base = Px<ZG_PIXEL_RGBA_U8>::zero(); return; }
        const float base_weight = b[3] * (1.0f - o[3]);
        const float inv = 1.0f / result_a;
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = (blended[i] * o[3] + b[i] * base_weight) * inv;
        out[3] = result_a;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = out[i] < 0.0f ? 0.0f : (out[i] > 1.0f ? 1.0f : out[i]);
        base[i] = (uint8_t)(int)roundf(255.0f * v);
    }
}

template <int PIX, int KIND>
__global__ __launch_bounds__(256) void k_insert(DImg self, DImg source, InsertParams q, MethodArg m) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    const int c = q.min_c + (int)(blockIdx.x * 64 + (threadIdx.x & 63));
    const int r = q.min_r + (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (c >= q.max_c || r >= q.max_r) return;
    Vec sample;
    if (q.aligned) { // iterate destination pixels covered by the source rectangle
        const int sr = r - q.dst_top, sc = c - q.dst_left;
        sample = P::load(source.data, (size_t)sr * source.stride + (size_t)sc);
    } else {
        const float dy = (float)r - q.cy, dx = (float)c - q.cx;
        const float rect_x = q.cos_a * dx + q.sin_a * dy;
        const float rect_y = -q.sin_a * dx + q.cos_a * dy;
        if (fabsf(rect_x) > q.half_width || fabsf(rect_y) > q.half_height) return;
        const float norm_x = (rect_x + q.half_width) * q.inv_width;
        const float norm_y = (rect_y + q.half_height) * q.inv_height;
        const float sx = source.cols == 1 ? 0.0f : norm_x * q.fcols_m1;
        const float sy = source.rows == 1 ? 0.0f : norm_y * q.frows_m1;
        if (!interpolate<PIX, KIND>(source, sx, sy, m, ZG_BORDER_MIRROR, sample)) return;
    }
    const size_t di = (size_t)r * self.stride + (size_t)c;
    if (q.mask) q.mask[(size_t)(r - q.min_r) * q.mask_w + (c - q.min_c)] = 1;
    if constexpr (PIX == ZG_PIXEL_RGBA_U8) {
        if (q.blend != 0) {
            Vec d = P::load(self.data, di);
            blend_u8(d, sample, q.blend);
            P::store(self.data, di, d);
            return;
        }
    }
    P::store(self.data, di, sample);
}

// convertColor between the six image pixel types (color.zig:108-151 with the scalar / Rgb / Rgba rules of :365-390, :484-512,
// :1031-1047): what assignPixel applies when source and destination types differ.
template <int SPIX, int DPIX> __device__ inline typename Px<DPIX>::Vec convert_px(typename Px<SPIX>::Vec sv) {
    using SP = Px<SPIX>;
    using DP = Px<DPIX>;
    constexpr bool SF = std::is_same<typename SP::Elem, float>::value, DF = std::is_same<typename DP::Elem, float>::value;
    constexpr int SC = SP::C, DC = DP::C;
    typename DP::Vec d = DP::zero();
    auto u8_of = [](float v) -> uint8_t { const float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); return (uint8_t)(int)roundf(255.0f * c); };
    if constexpr (DC == 1) {
        if constexpr (SC == 1) { // scalar <-> scalar (:113-119)
            if constexpr (SF == DF) d[0] = sv[0];
            else if constexpr (!SF) d[0] = (float)sv[0] / 255.0f;
            else { double v = (double)sv[0]; v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); d[0] = (uint8_t)(int)round(v * 255.0); }
        } else if constexpr (!SF) { // colour(u8) -> luminance, BT.709 16.16 fixed point
            const int y0 = (13933 * (int)sv[0] + 46871 * (int)sv[1] + 4732 * (int)sv[2] + 32768) >> 16;
            const int y = y0 < 0 ? 0 : (y0 > 255 ? 255 : y0);
            if constexpr (DF) d[0] = (float)y / 255.0f; else d[0] = (uint8_t)y;
        } else {
            float y = 0.2126f * sv[0] + 0.7152f * sv[1] + 0.0722f * sv[2];
            y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
            if constexpr (DF) d[0] = y; else d[0] = u8_of(y);
        }
    } else if constexpr (SC == 1) { // scalar -> colour: Gray(Src).as(DestT), replicated, opaque
        typename DP::Elem g;
        if constexpr (SF == DF) g = sv[0];
        else if constexpr (!SF) g = (float)sv[0] / 255.0f;
        else g = u8_of(sv[0]);
        d[0] = g; d[1] = g; d[2] = g;
        if constexpr (DC == 4) d[3] = DF ? (typename DP::Elem)1 : (typename DP::Elem)255;
    } else { // colour -> colour: component type first (.as), then Rgb <-> Rgba
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if constexpr (SF == DF) d[i] = sv[i];
            else if constexpr (!SF) d[i] = (float)sv[i] / 255.0f;
            else d[i] = u8_of(sv[i]);
        }
        if constexpr (DC == 4) {
            if constexpr (SC == 4) {
                if constexpr (SF == DF) d[3] = sv[3];
                else if constexpr (!SF) d[3] = (float)sv[3] / 255.0f;
                else d[3] = u8_of(sv[3]);
            } else d[3] = DF ? (typename DP::Elem)1 : (typename DP::Elem)255;
        }
    }
    return d;
}

// assignPixel (image.zig:67-94) for differing source / destination types, over the samples of the bounding box:
// Rgba(u8) samples with a blend mode composite through Rgba(u8); everything else is convertColor(Dest, sample).
template <int SPIX, int DPIX>
__global__ __launch_bounds__(256) void k_insert_assign(DImg self, const void *samples, const uint8_t *mask, int min_r, int max_r, int min_c, int max_c, int blend) {
    using SP = Px<SPIX>;
    using DP = Px<DPIX>;
    const int c = min_c + (int)(blockIdx.x * 64 + (threadIdx.x & 63)), r = min_r + (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (c >= max_c || r >= max_r) return;
    const size_t bi = (size_t)(r - min_r) * (max_c - min_c) + (c - min_c);
    if (!mask[bi]) return;
    const typename SP::Vec sample = SP::load(samples, bi);
    const size_t di = (size_t)r * self.stride + (size_t)c;
    if constexpr (SPIX == ZG_PIXEL_RGBA_U8) {
        if (blend != 0) {
            typename SP::Vec d = convert_px<DPIX, ZG_PIXEL_RGBA_U8>(DP::load(self.data, di));
            blend_u8(d, sample, blend);
            DP::store(self.data, di, convert_px<ZG_PIXEL_RGBA_U8, DPIX>(d));
            return;
        }
    }
    DP::store(self.data, di, convert_px<SPIX, DPIX>(sample));
}

static int insert_impl(const zg_image *self, const zg_image *source, const float rect[4], float angle, float cos_a, float sin_a,
                       const zg_method *method, int blend_mode, hipStream_t s) {
    int rc;
    if ((rc = check_image(self, "self")) || (rc = check_image(source, "source")) || (rc = check_method(method))) return rc;
    ZG_REQUIRE(blend_mode >= 0 && blend_mode <= 12, ZG_ERR_INVALID_ARGUMENT, "insert: invalid Blending ordinal %d", blend_mode);
    ZG_REQUIRE(rect != nullptr, ZG_ERR_INVALID_ARGUMENT, "insert: null rect");
    if (source->rows == 0 || source->cols == 0 || self->rows == 0 || self->cols == 0) return ZG_OK;
    const float frows = (float)source->rows, fcols = (float)source->cols;
    const float rect_width = rect_w(rect), rect_height = rect_h(rect), eps = 1e-6f;
    InsertParams q{};
    q.blend = blend_mode;
    if (std::fabs(angle) < eps && std::fabs(rect_width - fcols) < eps && std::fabs(rect_height - frows) < eps) {
        q.aligned = 1;
        q.dst_top = (int)std::round(rect[1]);
        q.dst_left = (int)std::round(rect[0]);
        q.min_r = std::max(0, q.dst_top);
        q.max_r = (int)std::min<long long>(self->rows, (long long)q.dst_top + source->rows);
        q.min_c = std::max(0, q.dst_left);
        q.max_c = (int)std::min<long long>(self->cols, (long long)q.dst_left + source->cols);
    } else {
        q.cx = (rect[0] + rect[2]) * 0.5f; q.cy = (rect[1] + rect[3]) * 0.5f;
        q.cos_a = cos_a; q.sin_a = sin_a;
        q.inv_width = 1.0f / rect_width; q.inv_height = 1.0f / rect_height;
        q.half_width = rect_width * 0.5f; q.half_height = rect_height * 0.5f;
        q.fcols_m1 = fcols - 1; q.frows_m1 = frows - 1;
        const float abs_cos = std::fabs(cos_a), abs_sin = std::fabs(sin_a);
        const float bound_hw = q.half_width * abs_cos + q.half_height * abs_sin;
        const float bound_hh = q.half_width * abs_sin + q.half_height * abs_cos;
        q.min_r = (q.cy - bound_hh < 0) ? 0 : (int)std::floor(q.cy - bound_hh);
        q.max_r = (int)std::min<double>(self->rows, (double)std::ceil(q.cy + bound_hh) + 1);
        q.min_c = (q.cx - bound_hw < 0) ? 0 : (int)std::floor(q.cx - bound_hw);
        q.max_c = (int)std::min<double>(self->cols, (double)std::ceil(q.cx + bound_hw) + 1);
    }
    if (q.min_r >= q.max_r || q.min_c >= q.max_c) return ZG_OK;
    LutHolder lut;
    if ((rc = device_lanczos_lut(method, s, lut))) return rc;
    const MethodArg m{method->kind, method->b, method->c, lut.dev};
    const dim3 grid(ceil_div((unsigned)(q.max_c - q.min_c), 64), ceil_div((unsigned)(q.max_r - q.min_r), 4));
    // Differing types (the reference's `source: anytype`): sample in the SOURCE type into a scratch copy of the bounding box
    // with the kernels below (blend off, a mask of the pixels that got a sample), then assignPixel converts / composites.
    const bool mixed = self->pixel != source->pixel;
    const int bw = q.max_c - q.min_c, bh = q.max_r - q.min_r;
    zg_image target = *self;
    char *scratch = nullptr;
    if (mixed) {
        const size_t sps = pixel_size(source->pixel), samples_bytes = ((size_t)bw * bh * sps + 15) / 16 * 16;
        if ((rc = scratch_alloc((void **)&scratch, samples_bytes + (size_t)bw * bh, s))) { release_lut(lut, s); return rc; }
        q.mask = (uint8_t *)(scratch + samples_bytes);
        q.mask_w = bw;
        q.blend = 0;
        if (hipMemsetAsync(q.mask, 0, (size_t)bw * bh, s) != hipSuccess) { scratch_free(scratch, s); release_lut(lut, s); ZG_HIP(hipErrorUnknown); }
        // a view of the scratch addressed with self's coordinates: pixel (min_r, min_c) is scratch[0], row pitch = box width
        target.pixel = source->pixel;
        target.stride = (size_t)bw;
        target.data = scratch - ((ptrdiff_t)q.min_r * bw + q.min_c) * (ptrdiff_t)sps;
    }
    rc = dispatch_pixel(source->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
#define ZG_INS(K) case K: hipLaunchKernelGGL((k_insert<PIX, K>), grid, dim3(256), 0, s, dimg(&target), dimg(source), q, m); break;
        switch (method->kind) {
            ZG_INS(ZG_INTERP_NEAREST) ZG_INS(ZG_INTERP_BILINEAR) ZG_INS(ZG_INTERP_BICUBIC)
            ZG_INS(ZG_INTERP_CATMULL_ROM) ZG_INS(ZG_INTERP_MITCHELL) ZG_INS(ZG_INTERP_LANCZOS)
        }
#undef ZG_INS
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
    if (mixed) {
        if (rc == ZG_OK)
            rc = dispatch_pixel(source->pixel, [&](auto stag) -> int {
                constexpr int SPIX = decltype(stag)::value;
                return dispatch_pixel(self->pixel, [&](auto dtag) -> int {
                    constexpr int DPIX = decltype(dtag)::value;
                    if constexpr (SPIX != DPIX) {
                        hipLaunchKernelGGL((k_insert_assign<SPIX, DPIX>), grid, dim3(256), 0, s, dimg(self), (const void *)scratch, (const uint8_t *)q.mask, q.min_r,
                                           q.max_r, q.min_c, q.max_c, blend_mode);
                        ZG_HIP(hipGetLastError());
                    }
                    return ZG_OK;
                });
            });
        scratch_free(scratch, s);
    }
    release_lut(lut, s);
    return rc;
}

} // namespace zg

using namespace zg;

// host-pointer wrappers: stage, run on the default stream, copy back
#define ZG_HOST2(call)                                                   \
    HostStage a, b;                                                      \
    int rc;                                                              \
    if ((rc = a.upload(src, true, false))) return rc;                    \
    if ((rc = b.upload(dst, false, true))) return rc;                    \
    if ((rc = (call))) return rc;                                        \
    ZG_HIP(hipStreamSynchronize(nullptr));                               \
    return b.finish();

extern "C" {

int zg_resize(const zg_image *src, const zg_image *dst, const zg_method *method, zg_stream stream) {
    return resize_impl(src, dst, method, as_stream(stream));
}
int zg_resize_host(const zg_image *src, const zg_image *dst, const zg_method *method) {
    ZG_HOST2(resize_impl(&a.dev, &b.dev, method, nullptr))
}

int zg_lanczos_plane_weights(uint32_t src_n, uint32_t dst_n, float *weights) {
    ZG_REQUIRE(weights != nullptr && src_n > 0 && dst_n > 0, ZG_ERR_INVALID_ARGUMENT, "zg_lanczos_plane_weights: null output or empty axis");
    lanczos_plane_weights(src_n, dst_n, weights);
    return ZG_OK;
}
int zg_resize_lanczos_weights(const zg_image *src, const zg_image *dst, const float *wx, const float *wy, zg_stream stream) {
    return resize_lanczos_weights_checked(src, dst, wx, wy, as_stream(stream));
}
int zg_resize_lanczos_weights_host(const zg_image *src, const zg_image *dst, const float *wx, const float *wy) {
    ZG_HOST2(resize_lanczos_weights_checked(&a.dev, &b.dev, wx, wy, nullptr))
}

int zg_letterbox(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t rect_out[4], zg_stream stream) {
    return letterbox_impl(src, dst, method, rect_out, as_stream(stream));
}
int zg_letterbox_host(const zg_image *src, const zg_image *dst, const zg_method *method, uint32_t rect_out[4]) {
    ZG_HOST2(letterbox_impl(&a.dev, &b.dev, method, rect_out, nullptr))
}

int zg_warp(const zg_image *src, const zg_image *dst, int kind, const float *m, const zg_method *method, zg_stream stream) {
    return warp_impl(src, dst, kind, m, method, as_stream(stream));
}
int zg_warp_host(const zg_image *src, const zg_image *dst, int kind, const float *m, const zg_method *method) {
    ZG_HOST2(warp_impl(&a.dev, &b.dev, kind, m, method, nullptr))
}

int zg_rotate_into(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a,
                   const zg_method *method, int border, zg_stream stream) {
    return rotate_into_impl(src, dst, angle, cos_a, sin_a, method, border, as_stream(stream));
}
int zg_rotate_into_host(const zg_image *src, const zg_image *dst, float angle, float cos_a, float sin_a,
                        const zg_method *method, int border) {
    ZG_HOST2(rotate_into_impl(&a.dev, &b.dev, angle, cos_a, sin_a, method, border, nullptr))
}

int zg_rotate_bounds(uint32_t rows, uint32_t cols, float angle, float cos_a, float sin_a, uint32_t *out_rows, uint32_t *out_cols) {
    ZG_REQUIRE(out_rows && out_cols, ZG_ERR_INVALID_ARGUMENT, "rotateBounds: null output");
    switch (orthogonal_case(angle)) {
    case 0: case 2: *out_rows = rows; *out_cols = cols; return ZG_OK;
    case 1: case 3: *out_rows = cols; *out_cols = rows; return ZG_OK;
    }
    const float cos_abs = std::fabs(cos_a), sin_abs = std::fabs(sin_a), w = (float)cols, h = (float)rows;
    *out_cols = (uint32_t)std::ceil(w * cos_abs + h * sin_abs);
    *out_rows = (uint32_t)std::ceil(h * cos_abs + w * sin_abs);
    return ZG_OK;
}

int zg_extract(const zg_image *src, const zg_image *dst, const float rect[4], float angle, float cos_a, float sin_a,
               const zg_method *method, int border, zg_stream stream) {
    return extract_impl(src, dst, rect, angle, cos_a, sin_a, method, border, as_stream(stream));
}
int zg_extract_host(const zg_image *src, const zg_image *dst, const float rect[4], float angle, float cos_a, float sin_a,
                    const zg_method *method, int border) {
    ZG_HOST2(extract_impl(&a.dev, &b.dev, rect, angle, cos_a, sin_a, method, border, nullptr))
}

int zg_crop_dims(const float rect[4], uint32_t *out_rows, uint32_t *out_cols) {
    ZG_REQUIRE(rect && out_rows && out_cols, ZG_ERR_INVALID_ARGUMENT, "crop: null argument");
    *out_rows = (uint32_t)std::round(rect_h(rect));
    *out_cols = (uint32_t)std::round(rect_w(rect));
    return ZG_OK;
}
static int crop_impl(const zg_image *src, const zg_image *dst, const float rect[4], hipStream_t s) {
    ZG_REQUIRE(rect != nullptr, ZG_ERR_INVALID_ARGUMENT, "crop: null rect");
    uint32_t r, c;
    zg_crop_dims(rect, &r, &c);
    ZG_REQUIRE(dst && dst->rows == r && dst->cols == c, ZG_ERR_DIMENSION_MISMATCH, "crop: destination must be %ux%u", r, c);
    const zg_method nearest{ZG_INTERP_NEAREST, 0, 0, nullptr};
    return extract_impl(src, dst, rect, 0.0f, 1.0f, 0.0f, &nearest, ZG_BORDER_ZERO, s);
}
int zg_crop(const zg_image *src, const zg_image *dst, const float rect[4], zg_stream stream) {
    return crop_impl(src, dst, rect, as_stream(stream));
}
int zg_crop_host(const zg_image *src, const zg_image *dst, const float rect[4]) {
    ZG_HOST2(crop_impl(&a.dev, &b.dev, rect, nullptr))
}

int zg_insert(const zg_image *self, const zg_image *source, const float rect[4], float angle, float cos_a, float sin_a,
              const zg_method *method, int blend_mode, zg_stream stream) {
    return insert_impl(self, source, rect, angle, cos_a, sin_a, method, blend_mode, as_stream(stream));
}
int zg_insert_host(const zg_image *self, const zg_image *source, const float rect[4], float angle, float cos_a, float sin_a,
                   const zg_method *method, int blend_mode) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(self, true, true))) return rc;
    if ((rc = b.upload(source, true, false))) return rc;
    if ((rc = insert_impl(&a.dev, &b.dev, rect, angle, cos_a, sin_a, method, blend_mode, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return a.finish();
}

} // extern "C"
