// convert.hip — Image(T).convertInto: per-pixel colour conversion.
//
// Replaces reference src/image.zig:396-407 (loop) and the parts of convertColor (src/color.zig:108-151) that
// lie on the image hot path:
//   Rgb/Rgba(u8|f32) -> Oklab(f32) / Xyz(f32)   color.zig:1252-1272 (gammaToLinear, rgbToXyz), :1381-1400 (xyzToOklab)
//   Rgb/Rgba(u8|f32) -> u8 / f32 luminance        color.zig:1031-1047 (u8: BT.709 16.16 fixed point)
//   Rgb <-> Rgba, u8 <-> f32 backing              color.zig:365-390, :484-512 (.as: /255 ; @round(255 * clamp))
//   u8 / f32 scalar -> Rgb / Rgba (grey replicate) color.zig:121-131, :1050-1052
//   Rgb/Rgba(u8) -> Ycbcr(u8)                     color.zig:987-1009 (BT.601 16.16 fixed point)
//
// Zig-std transcendentals are isolated: for u8 sources gammaToLinear has 256 possible arguments, so the kernel
// reads a 256-entry host-supplied table (std.math.pow stays on the caller's side); cbrt (continuous argument)
// is the musl cbrtf algorithm Zig ports — two Newton steps in f64, one final rounding — written out here.
// For float-typed RGB sources gammaToLinear needs pow on the device: restated below (exp/log based, as Zig's
// port of Go's Pow); like the oracle's it is parity-unpinned against real Zig at the last ulp.
#include "zg_common.h"
#include <algorithm>
#include "zg_hostmath.h"
#include "zg_colordev.h"
#include "zg_bilinear_u8.h"

#include <cmath>
#include <cstring>
#include <mutex>

#pragma clang fp contract(off)

namespace zg {

__device__ inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
__device__ inline uint8_t unit_to_u8(float v) { return (uint8_t)(int)roundf(255.0f * clamp01(v)); }
__device__ inline uint8_t gray_u8(int r, int g, int b) { // color.zig:1031-1042
    const int v = (13933 * r + 46871 * g + 4732 * b + 32768) >> 16;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct ConvertArgs {
    int src_space, dst_space;
    const float *srgb_lut; // device, 256 entries
};

template <int SPIX, int DPIX>
__global__ __launch_bounds__(256) void k_convert(DImg src, DImg dst, ConvertArgs a) {
    using SP = Px<SPIX>;
    using DP = Px<DPIX>;
    constexpr bool SF = std::is_same<typename SP::Elem, float>::value;
    constexpr bool DF = std::is_same<typename DP::Elem, float>::value;
    constexpr int SC = SP::C, DC = DP::C;
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    const typename SP::Vec sv = SP::load(src.data, (size_t)r * src.stride + (size_t)c);
    typename DP::Vec dv = DP::zero();

    // source as (r, g, b, a) in its own element type; grey replicated, missing alpha opaque
    float sf[4] = {0, 0, 0, 1.0f};
    int su[4] = {0, 0, 0, 255};
    if constexpr (SF) {
#pragma unroll
        for (int i = 0; i < SC; ++i) sf[i] = sv[i];
        if constexpr (SC == 1) { sf[1] = sf[0]; sf[2] = sf[0]; }
    } else {
#pragma unroll
        for (int i = 0; i < SC; ++i) su[i] = sv[i];
        if constexpr (SC == 1) { su[1] = su[0]; su[2] = su[0]; }
    }

    switch (a.dst_space) {
    case ZG_CS_GRAY:
        if constexpr (DC == 1) {
            if constexpr (SC == 1) { // scalar <-> scalar (color.zig:113-119)
                if constexpr (!SF && !DF) dv[0] = (uint8_t)su[0];
                else if constexpr (!SF && DF) dv[0] = (float)su[0] / 255.0f;
                else if constexpr (SF && !DF) {
                    double v = (double)sf[0];
                    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                    dv[0] = (uint8_t)(int)round(v * 255.0);
                } else dv[0] = sf[0];
            } else if constexpr (!SF) {
                const uint8_t y = gray_u8(su[0], su[1], su[2]);
                if constexpr (DF) dv[0] = (float)y / 255.0f; else dv[0] = y;
            } else {
                const float y = clamp01(0.2126f * sf[0] + 0.7152f * sf[1] + 0.0722f * sf[2]);
                if constexpr (DF) dv[0] = y; else dv[0] = unit_to_u8(y);
            }
        }
        break;
    case ZG_CS_RGB:
    case ZG_CS_RGBA:
        if constexpr (DC >= 3) {
            if constexpr (DF) {
                float o[4];
                if constexpr (SF) { o[0] = sf[0]; o[1] = sf[1]; o[2] = sf[2]; o[3] = sf[3]; }
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (float)su[i] / 255.0f;
                    if constexpr (SC != 4) o[3] = 1.0f;
                }
#pragma unroll
                for (int i = 0; i < DC; ++i) dv[i] = o[i];
            } else {
                uint8_t o[4];
                if constexpr (!SF) { o[0] = (uint8_t)su[0]; o[1] = (uint8_t)su[1]; o[2] = (uint8_t)su[2]; o[3] = (uint8_t)su[3]; }
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = unit_to_u8(sf[i]);
                    if constexpr (SC != 4) o[3] = 255;
                }
#pragma unroll
                for (int i = 0; i < DC; ++i) dv[i] = o[i];
            }
        }
        break;
    case ZG_CS_XYZ:
    case ZG_CS_OKLAB:
        if constexpr (DF && DC == 3) {
            float lin[3];
            if constexpr (!SF) {
#pragma unroll
                for (int i = 0; i < 3; ++i) lin[i] = a.srgb_lut[su[i]];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) lin[i] = dev_gamma_to_linear(sf[i]);
            }
            float X, Y, Z;
            linear_rgb_to_xyz(lin, X, Y, Z);
            if (a.dst_space == ZG_CS_XYZ) {
                dv[0] = X; dv[1] = Y; dv[2] = Z;
            } else {
                float L, A, B;
                xyz_to_oklab(X, Y, Z, L, A, B);
                dv[0] = L; dv[1] = A; dv[2] = B;
            }
        }
        break;
    case ZG_CS_YCBCR:
        if constexpr (!SF && !DF && DC == 3) { // color.zig:987-1009
            const long long rr = su[0], gg = su[1], bb = su[2];
            const long long y = (19595ll * rr + 38470ll * gg + 7471ll * bb + 32768) >> 16;
            const long long cb = ((-11059ll * rr + -21710ll * gg + 32768ll * bb + 32768) >> 16) + 128;
            const long long cr = ((32768ll * rr + -27439ll * gg + -5329ll * bb + 32768) >> 16) + 128;
            dv[0] = (uint8_t)(y < 0 ? 0 : (y > 255 ? 255 : y));
            dv[1] = (uint8_t)(cb < 0 ? 0 : (cb > 255 ? 255 : cb));
            dv[2] = (uint8_t)(cr < 0 ? 0 : (cr > 255 ? 255 : cr));
        }
        break;
    }
    DP::store(dst.data, (size_t)r * dst.stride + (size_t)c, dv);
}

// Rgb(u8) / Rgba(u8) -> Xyz / Oklab (f32), four pixels per lane. k_convert's one pixel per lane leaves a wave with a single chain
// load -> three table look-ups -> ~220 VALU -> store and nothing to overlap it with: at eight waves per SIMD the VALU sat idle a third
// of the time (SQ_WAIT_INST_ANY 37 % of wave-cycles). Here a lane loads four pixels at once (one dwordx4 / dwordx3), looks its twelve
// bytes up in an LDS copy of the sRGB table, and stores 48 contiguous bytes as three dwordx4. Same device functions as k_convert
// (zg_colordev.h), so the bits are the same by construction. Needs 16-byte aligned rows on the f32 side and 4 * SC-byte aligned
// rows on the u8 side (the launcher checks; anything else stays on k_convert).
typedef uint32_t lab4_u32x4 __attribute__((ext_vector_type(4)));
typedef float lab4_f32x4 __attribute__((ext_vector_type(4)));
#ifndef LAB4_NT
#define LAB4_NT 1 // streaming stores for the three 1 KiB rows a wave writes (A/B: tools/build_variant.sh lab4_plain convert.hip -DLAB4_NT=0)
#endif
template <int SC, int MODE> // MODE 0: Xyz, 1: Oklab, 2: Oklab from a table whose entries are +0 or within [2^-60, 2^60] (xyz_to_oklab<true>), 3 / 4: Lab likewise
__global__ __launch_bounds__(256) void k_u8_to_lab4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t src_pitch, uint64_t dst_pitch,
                                                    int rows, int cols, const float *__restrict__ lut_global) {
    __shared__ float lut[256];
    __shared__ __attribute__((aligned(16))) float turn[4 * 768];
    lut[threadIdx.x] = lut_global[threadIdx.x];
    __syncthreads();
    const int c0 = (int)(blockIdx.x * 256 + threadIdx.x) * 4, r = grid_row();
    if (c0 >= cols || r >= rows) return;
    const uint8_t *sp = src + (uint64_t)r * src_pitch + (uint64_t)c0 * SC;
    float *dp = (float *)(dst + (uint64_t)r * dst_pitch) + (uint64_t)c0 * 3;
    const int n = cols - c0 < 4 ? cols - c0 : 4;
    uint32_t w[SC] = {};
    if (n == 4) {
        if constexpr (SC == 4) {
            const lab4_u32x4 v = *(const lab4_u32x4 *)sp;
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            const uint32_t *q = (const uint32_t *)sp;
            w[0] = q[0]; w[1] = q[1]; w[2] = q[2];
        }
    } else { // the ragged end of a row, byte by byte
        for (int i = 0; i < n * SC; ++i) {
            const uint32_t b = (uint32_t)sp[i] << (8 * (i & 3));
#pragma unroll
            for (int k = 0; k < SC; ++k) w[k] |= (i >> 2) == k ? b : 0u;
        }
    }
    float out[12];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float lin[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int byte = p * SC + i;
            lin[i] = lut[(w[byte >> 2] >> (8 * (byte & 3))) & 255u];
        }
        float X, Y, Z;
        linear_rgb_to_xyz(lin, X, Y, Z);
        if constexpr (MODE == 0) { out[3 * p] = X; out[3 * p + 1] = Y; out[3 * p + 2] = Z; }
        else if constexpr (MODE <= 2) xyz_to_oklab<MODE == 2>(X, Y, Z, out[3 * p], out[3 * p + 1], out[3 * p + 2]);
        else xyz_to_lab<MODE == 4>(X, Y, Z, out[3 * p], out[3 * p + 1], out[3 * p + 2]);
    }
    // a lane's 48 bytes are contiguous but the next lane's start 48 bytes on: stored as they are, every dwordx4 store of the wave touches
    // 24 cache lines and fills a third of each. When all 64 lanes hold four pixels the wave turns its 3 KiB through LDS instead (each
    // wave its own 3 KiB, LDS operations of one wave execute in order) and stores three times 1 KiB of consecutive addresses.
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int wave_c0 = (int)(blockIdx.x * 256 + (threadIdx.x & ~63u)) * 4;
    if (wave_c0 + 256 <= cols) {
        lab4_f32x4 *mine = (lab4_f32x4 *)(turn + wave * 768);
#pragma unroll
        for (int k = 0; k < 3; ++k) mine[lane * 3 + k] = lab4_f32x4{out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lab4_f32x4 *wave_dp = (lab4_f32x4 *)(dp - (size_t)lane * 12);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (LAB4_NT) __builtin_nontemporal_store(mine[k * 64 + lane], &wave_dp[k * 64 + lane]); // written once, never read here: 12 of the 16 bytes a pixel moves
            else wave_dp[k * 64 + lane] = mine[k * 64 + lane];
        }
    } else if (n == 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) ((lab4_f32x4 *)dp)[k] = lab4_f32x4{out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]};
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (i < 3 * n) dp[i] = out[i];
    }
}

static bool lab4_applies(const zg_image *src, const zg_image *dst, int src_space, int dst_space) {
    if (dst->pixel != ZG_PIXEL_RGB_F32 || (dst_space != ZG_CS_XYZ && dst_space != ZG_CS_OKLAB && dst_space != ZG_CS_LAB)) return false;
    if (!((src->pixel == ZG_PIXEL_RGBA_U8 && src_space == ZG_CS_RGBA) || (src->pixel == ZG_PIXEL_RGB_U8 && src_space == ZG_CS_RGB))) return false;
    const uint64_t sc = src->pixel == ZG_PIXEL_RGBA_U8 ? 4 : 3, need = sc == 4 ? 16 : 4;
    if (((uintptr_t)src->data % need) || ((uint64_t)src->stride * sc % need)) return false;
    if (((uintptr_t)dst->data % 16) || ((uint64_t)dst->stride * 12 % 16)) return false;
    return true;
}

static int launch_lab4(const zg_image *src, const zg_image *dst, int dst_space, const float *lut, bool plain_table, hipStream_t s) {
    const dim3 grid = row_grid(ceil_div(ceil_div((unsigned)src->cols, 4u), 256u), (unsigned)src->rows);
    const uint8_t *sp = (const uint8_t *)src->data;
    uint8_t *dp = (uint8_t *)dst->data;
    const bool four = src->pixel == ZG_PIXEL_RGBA_U8, oklab = dst_space == ZG_CS_OKLAB;
    const uint64_t spitch = (uint64_t)src->stride * (four ? 4 : 3), dpitch = (uint64_t)dst->stride * 12;
    const int rows = (int)src->rows, cols = (int)src->cols;
    const int mode = dst_space == ZG_CS_LAB ? (plain_table ? 4 : 3) : !oklab ? 0 : (plain_table ? 2 : 1);
#define ZG_LAB4(SC, MODE) hipLaunchKernelGGL((k_u8_to_lab4<SC, MODE>), grid, dim3(256), 0, s, sp, dp, spitch, dpitch, rows, cols, lut)
    if (four) { if (mode == 0) ZG_LAB4(4, 0); else if (mode == 1) ZG_LAB4(4, 1); else if (mode == 2) ZG_LAB4(4, 2); else if (mode == 3) ZG_LAB4(4, 3); else ZG_LAB4(4, 4); }
    else { if (mode == 0) ZG_LAB4(3, 0); else if (mode == 1) ZG_LAB4(3, 1); else if (mode == 2) ZG_LAB4(3, 2); else if (mode == 3) ZG_LAB4(3, 3); else ZG_LAB4(3, 4); }
#undef ZG_LAB4
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// The way back, Lab(f32) -> Rgb(u8) / Rgba(u8) (route Lab -> Xyz -> Rgb -> u8, colorspaces.hip's hops: color.zig:1311-1330, :1275-1286, then
// @round(255 * clamp(c, 0, 1))), four pixels per lane: a wave's 3 KiB of Lab values arrive as three coalesced 1 KiB loads and are turned
// through LDS so that every lane holds its own four pixels; 16 (or 12) bytes out per lane.
template <int DC>
__global__ __launch_bounds__(256) void k_lab4_to_u8(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t src_pitch, uint64_t dst_pitch, int rows, int cols) {
    __shared__ __attribute__((aligned(16))) float turn[4 * 768];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int wave_c0 = (int)(blockIdx.x * 256 + (threadIdx.x & ~63u)) * 4, c0 = wave_c0 + lane * 4, r = grid_row();
    if (r >= rows || wave_c0 >= cols) return; // wave-uniform
    const float *row = (const float *)(src + (uint64_t)r * src_pitch);
    float in[12];
    if (wave_c0 + 256 <= cols) {
        lab4_f32x4 *mine = (lab4_f32x4 *)(turn + wave * 768);
        const lab4_f32x4 *wave_sp = (const lab4_f32x4 *)(row + (size_t)wave_c0 * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) mine[k * 64 + lane] = wave_sp[k * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const lab4_f32x4 v = mine[lane * 3 + k];
            in[4 * k] = v.x; in[4 * k + 1] = v.y; in[4 * k + 2] = v.z; in[4 * k + 3] = v.w;
        }
    } else { // the ragged end of a row
        const int n = min(max(cols - c0, 0), 4);
#pragma unroll
        for (int i = 0; i < 12; ++i) in[i] = i < 3 * n ? row[(size_t)c0 * 3 + i] : 0.0f;
    }
    const int n = min(max(cols - c0, 0), 4);
    uint32_t px[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float rgb[3];
        lab_to_rgb_unit(in[3 * p], in[3 * p + 1], in[3 * p + 2], rgb);
        uint32_t w = DC == 4 ? 0xff000000u : 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v = rgb[i];
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); // xyzToRgb's clamp, then the u8 conversion's own (the same function twice)
            w |= (uint32_t)(uint8_t)(int)roundf(255.0f * v) << (8 * i);
        }
        px[p] = w;
    }
    if constexpr (DC == 4) {
        uint32_t *dp = (uint32_t *)(dst + (uint64_t)r * dst_pitch) + c0;
        if (n == 4) *(lab4_u32x4 *)dp = lab4_u32x4{px[0], px[1], px[2], px[3]};
        else for (int p = 0; p < n; ++p) dp[p] = px[p];
    } else {
        uint8_t *dp = dst + (uint64_t)r * dst_pitch + (uint64_t)c0 * 3;
        if (n == 4) { // twelve bytes as three dwords (rows are 4-byte aligned)
            uint32_t *d32 = (uint32_t *)dp;
            d32[0] = (px[0] & 0xffffffu) | (px[1] << 24);
            d32[1] = ((px[1] >> 8) & 0xffffu) | (px[2] << 16);
            d32[2] = ((px[2] >> 16) & 0xffu) | (px[3] << 8);
        } else {
            for (int p = 0; p < n; ++p) { dp[3 * p] = (uint8_t)px[p]; dp[3 * p + 1] = (uint8_t)(px[p] >> 8); dp[3 * p + 2] = (uint8_t)(px[p] >> 16); }
        }
    }
}
static bool lab4_back_applies(const zg_image *src, const zg_image *dst, int src_space, int dst_space) {
    if (src->pixel != ZG_PIXEL_RGB_F32 || src_space != ZG_CS_LAB) return false;
    if (!((dst->pixel == ZG_PIXEL_RGBA_U8 && dst_space == ZG_CS_RGBA) || (dst->pixel == ZG_PIXEL_RGB_U8 && dst_space == ZG_CS_RGB))) return false;
    const uint64_t dc = dst->pixel == ZG_PIXEL_RGBA_U8 ? 4 : 3, need = dc == 4 ? 16 : 4;
    if (((uintptr_t)dst->data % need) || ((uint64_t)dst->stride * dc % need)) return false;
    if (((uintptr_t)src->data % 16) || ((uint64_t)src->stride * 12 % 16)) return false;
    return true;
}
static int launch_lab4_back(const zg_image *src, const zg_image *dst, hipStream_t s) {
    const dim3 grid = row_grid(ceil_div(ceil_div((unsigned)src->cols, 4u), 256u), (unsigned)src->rows);
    const uint64_t spitch = (uint64_t)src->stride * 12;
    if (dst->pixel == ZG_PIXEL_RGBA_U8)
        hipLaunchKernelGGL(k_lab4_to_u8<4>, grid, dim3(256), 0, s, (const uint8_t *)src->data, (uint8_t *)dst->data, spitch, (uint64_t)dst->stride * 4, (int)src->rows, (int)src->cols);
    else
        hipLaunchKernelGGL(k_lab4_to_u8<3>, grid, dim3(256), 0, s, (const uint8_t *)src->data, (uint8_t *)dst->data, spitch, (uint64_t)dst->stride * 3, (int)src->rows, (int)src->cols);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// `plain` (optional) comes back true when every entry of the table is +0 or a positive number within [2^-60, 2^60]: what
// xyz_to_oklab<true> asks for. The library's own sRGB table is (entry 1 is 3.0e-4).
static int device_srgb_lut(const float *host_lut, hipStream_t s, const float **out, float **owned, bool *plain = nullptr) {
    *owned = nullptr;
    if (plain) {
        *plain = true;
        if (host_lut)
            for (int i = 0; i < 256; ++i) {
                uint32_t u;
                memcpy(&u, &host_lut[i], 4);
                if (u != 0 && (u < 0x21800000u || u >= 0x5d800000u)) *plain = false; // -0, negatives, tiny, huge, inf, nan
            }
    }
    if (host_lut) {
        if (int rc = scratch_alloc((void **)owned, 256 * sizeof(float), s)) return rc;
        if (int rc = upload_pageable(*owned, host_lut, 256 * sizeof(float), s)) return rc;
        *out = *owned;
        return ZG_OK;
    }
    static std::mutex mu;
    static float *per_device[64] = {nullptr};
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && !per_device[dev]) {
        float *p = nullptr;
        ZG_HIP(hipMalloc((void **)&p, 256 * sizeof(float)));
        if (int rc = upload_pageable(p, hostmath::srgb_u8_lut(), 256 * sizeof(float), nullptr)) return rc;
        per_device[dev] = p;
    }
    *out = per_device[dev];
    return ZG_OK;
}

static int space_channels(int space) { return space == ZG_CS_GRAY ? 1 : (space == ZG_CS_RGBA ? 4 : 3); }

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);
int convert_spaces_impl(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const float *srgb_lut_dev, hipStream_t s);

// Gray(u8) -> Rgba(u8) (convertColor: r = g = b = grey, a = 255; the last step of the CLI's edges bridge, src/cli/edges.zig:133-135), four pixels per lane:
// a dword in, sixteen bytes out. (k_convert<U8, Rgba(u8)>, a pixel per lane: 62 us for the 23 M pixels of 64 x 450 x 800 edge maps = 1.9 TB/s.)
__global__ __launch_bounds__(256) void k_gray8_to_rgba8_4(DImg src, DImg dst) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int c = (int)(blockIdx.x * 256 + threadIdx.x) * 4, r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    const uint32_t g = *(const uint32_t *)((const uint8_t *)src.data + (size_t)r * src.stride + c);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = ((g >> (8 * k)) & 0xffu) * 0x00010101u | 0xff000000u;
    __builtin_nontemporal_store(o, (u32x4 *)((uint8_t *)dst.data + ((size_t)r * dst.stride + c) * 4));
}

int convert_impl(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const float *srgb_lut, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "convert: shapes differ");
    ZG_REQUIRE(src_space >= ZG_CS_GRAY && src_space <= ZG_CS_XYB, ZG_ERR_INVALID_ARGUMENT, "convert: invalid source space %d", src_space);
    ZG_REQUIRE(dst_space >= ZG_CS_GRAY && dst_space <= ZG_CS_XYB, ZG_ERR_INVALID_ARGUMENT, "convert: invalid destination space %d", dst_space);
    ZG_REQUIRE(pixel_channels(src->pixel) == space_channels(src_space), ZG_ERR_INVALID_ARGUMENT, "convert: source layout does not match its colour space");
    ZG_REQUIRE(pixel_channels(dst->pixel) == space_channels(dst_space), ZG_ERR_INVALID_ARGUMENT, "convert: destination layout does not match its colour space");
    const bool df = pixel_is_float(dst->pixel), sf = pixel_is_float(src->pixel);
    const bool fast_pair = (src_space == ZG_CS_GRAY || src_space == ZG_CS_RGB || src_space == ZG_CS_RGBA) && dst_space <= ZG_CS_YCBCR &&
                           !(dst_space == ZG_CS_YCBCR && (df || sf));
    if (!fast_pair) { // every other pair of colour spaces: the route-walking kernel (colorspaces.hip)
        if (src->rows == 0 || src->cols == 0) return ZG_OK;
        if (src_space == dst_space && src->pixel == dst->pixel) return copy_impl(src, dst, s);
        if (lab4_back_applies(src, dst, src_space, dst_space)) return launch_lab4_back(src, dst, s); // Lab(f32) -> Rgb(u8) / Rgba(u8), four pixels per lane
        const float *lut_dev = nullptr; // gammaToLinear(u8 / 255): the caller's table if given, else the library's
        float *lut_owned = nullptr;
        bool plain_lut = false;
        if (!sf && (rc = device_srgb_lut(srgb_lut, s, &lut_dev, &lut_owned, &plain_lut))) return rc;
        if (lab4_applies(src, dst, src_space, dst_space)) { // Rgb(u8) / Rgba(u8) -> Lab(f32): the route Rgb -> Xyz -> Lab, four pixels per lane
            rc = launch_lab4(src, dst, dst_space, lut_dev, plain_lut, s);
            if (lut_owned) scratch_free(lut_owned, s);
            return rc;
        }
        rc = convert_spaces_impl(src, src_space, dst, dst_space, lut_dev, s);
        if (lut_owned) scratch_free(lut_owned, s);
        return rc;
    }
    if (dst_space == ZG_CS_XYZ || dst_space == ZG_CS_OKLAB) ZG_REQUIRE(df, ZG_ERR_UNSUPPORTED, "convert: Xyz / Oklab need a float destination");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    if (src_space == dst_space && src->pixel == dst->pixel) return copy_impl(src, dst, s); // T == TargetType

    if (src->pixel == ZG_PIXEL_U8 && dst->pixel == ZG_PIXEL_RGBA_U8 && src_space == ZG_CS_GRAY && dst_space == ZG_CS_RGBA && src->cols % 4 == 0 && src->stride % 4 == 0 &&
        dst->stride % 4 == 0 && ((uintptr_t)src->data & 3) == 0 && ((uintptr_t)dst->data & 15) == 0) {
        hipLaunchKernelGGL(k_gray8_to_rgba8_4, row_grid(ceil_div(src->cols, 1024), src->rows), dim3(256), 0, s, dimg(src), dimg(dst));
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    ConvertArgs a{src_space, dst_space, nullptr};
    float *owned = nullptr;
    bool plain_table = false;
    if (!sf && (dst_space == ZG_CS_XYZ || dst_space == ZG_CS_OKLAB)) {
        if ((rc = device_srgb_lut(srgb_lut, s, &a.srgb_lut, &owned, &plain_table))) return rc;
    }
    if (lab4_applies(src, dst, src_space, dst_space)) {
        rc = launch_lab4(src, dst, dst_space, a.srgb_lut, plain_table, s);
        if (owned) scratch_free(owned, s);
        return rc;
    }
    const dim3 grid = row_grid(ceil_div(src->cols, 256), src->rows);
    rc = dispatch_pixel(src->pixel, [&](auto stag) -> int {
        constexpr int SPIX = decltype(stag)::value;
        return dispatch_pixel(dst->pixel, [&](auto dtag) -> int {
            constexpr int DPIX = decltype(dtag)::value;
            hipLaunchKernelGGL((k_convert<SPIX, DPIX>), grid, dim3(256), 0, s, dimg(src), dimg(dst), a);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
    });
    if (owned) scratch_free(owned, s);
    return rc;
}


// ---- resize -> convert in one pass (BASELINE configs[2]: bilinear 4096^2 -> 1024^2 Rgba(u8), then Rgb -> Oklab) -----------------
// The pipeline [resize, convert] writes a resized Rgba(u8) image only for convert to read it back; per output pixel that is
// 4 B written and 4 B read again, and a second launch. Fused: the resized pixel stays in a register and goes straight into
// convertColor. Same arithmetic as k_resize_bilinear_rgba8 followed by k_convert (the two headers both kernels share), so the
// result equals the two calls bit for bit. Algorithmic bytes: 16 read + 12 written per output pixel (28 B).
int resize_impl(const zg_image *src, const zg_image *dst, const zg_method *method, hipStream_t s);

template <int MODE, int WAVES> // MODE as k_u8_to_lab4's; WAVES 4: four rows per workgroup in XCD-major order (round 4), 1: one-wave workgroups in address order
__global__ __launch_bounds__(64 * WAVES) void k_resize_bilinear_rgba8_to_lab(DImg src, DImg dst, float ratio_x, float ratio_y, int tiles_x, const float *srgb_lut, FrameSpan fr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    int wg = blockIdx.x;
    if constexpr (WAVES == 4) {
        const int nwg = gridDim.x, per_xcd = nwg >> 3;
        if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    }
    src.data = (char *)src.data + (size_t)blockIdx.y * fr.src_frame; // a batch of equally shaped frames in one launch (batch.hip)
    dst.data = (char *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    const int tyi = wg / tiles_x, txi = wg - tyi * tiles_x;
    const int c_raw = txi * 64 + (int)(threadIdx.x & 63);
    const int r = __builtin_amdgcn_readfirstlane(tyi * WAVES + (int)(threadIdx.x >> 6));
    if (r >= dst.rows) return; // wave-uniform
    if constexpr (WAVES == 4) {
        if (c_raw >= dst.cols) return;
    }
    // WAVES == 1 (round 5): the wave works as a whole. Its sRGB table comes into LDS with ONE 1 KiB load issued beside the tap loads (three
    // table gathers from memory behind the taps were a second dependent round trip per wave), and its 768 bytes of results are turned
    // through LDS into three coalesced 256-byte stores (three dword stores 12 bytes apart per lane were three partial sweeps of the same
    // six lines). Lanes past the row's end compute a duplicate of its last pixel and store nothing. LDS operations of one wave execute in
    // order; the fences keep the compiler from reordering them.
    __shared__ float lds[WAVES == 1 ? 256 + 192 : 1];
    const int lane = (int)(threadIdx.x & 63);
    const int c = WAVES == 1 ? min(c_raw, dst.cols - 1) : c_raw;
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    f32x4s lut4 = {0, 0, 0, 0};
    if constexpr (WAVES == 1) lut4 = *(const f32x4s *)(srgb_lut + 4 * lane);
    int y0, y1, fy, x0, x1, fx;
    bilinear_taps(r, ratio_y, src.rows, y0, y1, fy);
    bilinear_taps(c, ratio_x, src.cols, x0, x1, fx);
    const uint32_t *row0 = (const uint32_t *)src.data + (size_t)y0 * src.stride, *row1 = (const uint32_t *)src.data + (size_t)y1 * src.stride;
    const int xp = min(x0, src.cols - 2);
    const u32x2 p0 = *(const u32x2 *)(row0 + xp), p1 = *(const u32x2 *)(row1 + xp);
    if constexpr (WAVES == 1) {
        *(f32x4s *)(lds + 4 * lane) = lut4;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t tl = p0[0], tr = p0[1], bl = p1[0], br = p1[1];
    if (x1 != x0 + 1 || x0 > src.cols - 2) { tl = row0[x0]; tr = row0[x1]; bl = row1[x0]; br = row1[x1]; }
    const uint32_t px = bilinear_rgba8(tl, tr, bl, br, fx, fy);
    const float *table = WAVES == 1 ? (const float *)lds : srgb_lut;
    const float lin[3] = {table[px & 0xffu], table[(px >> 8) & 0xffu], table[(px >> 16) & 0xffu]};
    float X, Y, Z, o0, o1, o2;
    linear_rgb_to_xyz(lin, X, Y, Z);
    if constexpr (MODE == 0) { o0 = X; o1 = Y; o2 = Z; }
    else xyz_to_oklab<MODE == 2>(X, Y, Z, o0, o1, o2);
    if constexpr (WAVES == 1) {
        float *stage = lds + 256;
        stage[3 * lane] = o0; stage[3 * lane + 1] = o1; stage[3 * lane + 2] = o2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int valid = 3 * min(64, dst.cols - txi * 64); // floats of this wave's row segment that exist
        float *o = (float *)dst.data + ((size_t)r * dst.stride + (size_t)txi * 64) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane + 64 * k < valid) o[lane + 64 * k] = stage[lane + 64 * k];
    } else {
        float *o = (float *)dst.data + ((size_t)r * dst.stride + (size_t)c) * 3;
        o[0] = o0; o[1] = o1; o[2] = o2;
    }
}

// [resize(.bilinear), convert(Oklab | Xyz f32)] of n Rgba(u8) frames in one launch; -1 when the fused kernel does not apply.
int resize_convert_rgba8_frames(const zg_image *src, const zg_image *dst, int dst_space, uint32_t n, size_t src_frame, size_t dst_frame, const float *srgb_lut,
                                hipStream_t s) {
    const bool fused = src->pixel == ZG_PIXEL_RGBA_U8 && dst->pixel == ZG_PIXEL_RGB_F32 && (dst_space == ZG_CS_OKLAB || dst_space == ZG_CS_XYZ) && src->rows > 0 &&
                       src->cols >= 2 && dst->rows > 0 && dst->cols > 0 && n > 0 && !(src->rows == dst->rows && src->cols == dst->cols);
    if (!fused) return -1;
    // reductions: one-wave workgroups in address order, as the plain resize (resize_planes.hip); 0 / 1 in ZIGNAL_HIP_RESIZE_FORM force a form
    bool one_wave = (float)src->cols / (float)dst->cols > 1.5f;
    static const int forced_form = getenv("ZIGNAL_HIP_RESIZE_FORM") ? atoi(getenv("ZIGNAL_HIP_RESIZE_FORM")) : -1; // read once
    if (forced_form >= 0) one_wave = forced_form != 0;
    const int tiles_x = (int)ceil_div(dst->cols, 64), tiles_y = (int)ceil_div(dst->rows, one_wave ? 1u : 4u);
    const uint64_t tiles = (uint64_t)tiles_x * tiles_y;
    if (tiles > 0x7fffffffu || n > MAX_FRAMES_PER_LAUNCH) return -1;
    const dim3 grid((unsigned)tiles, n);
    const float *lut_dev = nullptr;
    float *owned = nullptr;
    bool plain_table = false;
    if (int rc = device_srgb_lut(srgb_lut, s, &lut_dev, &owned, &plain_table)) return rc;
    const float ratio_x = (float)src->cols / (float)dst->cols, ratio_y = (float)src->rows / (float)dst->rows;
    const FrameSpan fr{src_frame, dst_frame};
    const int mode = dst_space != ZG_CS_OKLAB ? 0 : (plain_table ? 2 : 1);
#define ZG_RL(MODE) \
    if (mode == MODE) { \
        if (one_wave) hipLaunchKernelGGL((k_resize_bilinear_rgba8_to_lab<MODE, 1>), grid, dim3(64), 0, s, dimg(src), dimg(dst), ratio_x, ratio_y, tiles_x, lut_dev, fr); \
        else hipLaunchKernelGGL((k_resize_bilinear_rgba8_to_lab<MODE, 4>), grid, dim3(256), 0, s, dimg(src), dimg(dst), ratio_x, ratio_y, tiles_x, lut_dev, fr); \
    }
    ZG_RL(0) ZG_RL(1) ZG_RL(2)
#undef ZG_RL
    const hipError_t e = hipGetLastError();
    if (owned) scratch_free(owned, s);
    ZG_HIP(e);
    return ZG_OK;
}

static int resize_convert_impl(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const zg_method *method,
                               const float *srgb_lut, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(method != nullptr && method->kind >= ZG_INTERP_NEAREST && method->kind <= ZG_INTERP_LANCZOS, ZG_ERR_INVALID_ARGUMENT, "resize + convert: invalid interpolation method");
    ZG_REQUIRE(src_space >= ZG_CS_GRAY && src_space <= ZG_CS_XYB && dst_space >= ZG_CS_GRAY && dst_space <= ZG_CS_XYB, ZG_ERR_INVALID_ARGUMENT, "resize + convert: invalid colour space");
    if (dst->rows == 0 || dst->cols == 0) return ZG_OK;
    if (src_space == ZG_CS_RGBA && method->kind == ZG_INTERP_BILINEAR) {
        const int rcf = resize_convert_rgba8_frames(src, dst, dst_space, 1, 0, 0, srgb_lut, s);
        if (rcf >= 0) return rcf;
    }
    // every other combination: the two steps as they are, through a resized image that lives in scratch for the call
    zg_image mid = *src;
    mid.rows = dst->rows;
    mid.cols = dst->cols;
    mid.stride = dst->cols;
    mid.data = nullptr;
    if ((rc = scratch_alloc(&mid.data, (size_t)mid.rows * mid.cols * pixel_size(mid.pixel), s))) return rc;
    rc = resize_impl(src, &mid, method, s);
    if (rc == ZG_OK) rc = convert_impl(&mid, src_space, dst, dst_space, srgb_lut, s);
    scratch_free(mid.data, s);
    return rc;
}


__global__ __launch_bounds__(256) void k_devmath_apply(int fn, const float *x, const float *y, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float a = x[i];
        float r;
        switch (fn) { // wave-uniform
        case 0: r = dev_cbrtf(a); break;
        case 1: r = dev_pow_2p4(a); break;
        case 2: r = dev_expf(a); break;
        case 3: r = dev_logf(a); break;
        case 4: r = dev_sinf(a); break;
        case 5: r = dev_cosf(a); break;
        case 6: r = dev_atan2f(a, y[i]); break;
        case 7: r = dev_powf(a, y[i]); break;
        case 8: r = dev_gamma_to_linear(a); break;
        case 9: r = dev_cbrtf_musl(a); break;
        case 10: r = a / 100.0f; break;
        case 11: r = dev_div100(a); break;
        case 12: r = dev_lab_forward(a); break;
        case 13: { bool redo; r = dev_lab_forward_fast(a, redo); if (redo) r = dev_lab_forward(a); break; }
        case 14: r = a / LAB_XN; break;
        case 15: { bool redo; r = dev_div_const_fast(a, LAB_XN, LAB_XN_R, redo); if (redo) r = a / LAB_XN; break; }
        case 16: r = a / LAB_ZN; break;
        case 17: { bool redo; r = dev_div_const_fast(a, LAB_ZN, LAB_ZN_R, redo); if (redo) r = a / LAB_ZN; break; }
        case 18: r = dev_linear_to_gamma(a); break;
        case 19: { bool redo; r = dev_linear_to_gamma_fast(a, redo); if (redo) r = dev_linear_to_gamma(a); break; }
        case 20: r = a / 116.0f; break;
        case 21: { bool redo; r = dev_div_const_fast(a, 116.0f, 1.0f / 116.0f, redo); if (redo) r = a / 116.0f; break; }
        case 22: r = a / 500.0f; break;
        case 23: { bool redo; r = dev_div_const_fast(a, 500.0f, 1.0f / 500.0f, redo); if (redo) r = a / 500.0f; break; }
        case 24: r = a / 200.0f; break;
        default: { bool redo; r = dev_div_const_fast(a, 200.0f, 1.0f / 200.0f, redo); if (redo) r = a / 200.0f; break; }
        }
        out[i] = r;
    }
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_convert(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const float *srgb_lut, zg_stream stream) {
    return convert_impl(src, src_space, dst, dst_space, srgb_lut, as_stream(stream));
}

int zg_convert_host(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const float *srgb_lut) {
    { // per-pixel: no halo at all
        const int brc = host_banded(src, dst, 0, [&](const zg_image *sv, const zg_image *dv, hipStream_t s) { return convert_impl(sv, src_space, dv, dst_space, srgb_lut, s); });
        if (brc >= 0) return brc;
    }
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = convert_impl(&a.dev, src_space, &b.dev, dst_space, srgb_lut, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}


// Diagnostics: the device's own maths functions (zg_devmath.h) applied element-wise to device arrays, so that the
// transcendental boundary of DESIGN.md section 4 can be swept densely against the oracle's restatement (tests/test_math_pin.py).
// fn: 0 cbrt, 1 pow(x, 2.4), 2 exp, 3 log, 4 sin, 5 cos, 6 atan2(x, y), 7 pow(x, y), 8 gammaToLinear, 9 cbrt by musl's own steps (what 0 is
// checked against over all 2^32 inputs). y may be NULL for unary fn.
int zg_devmath_apply(int fn, const float *x_dev, const float *y_dev, float *out_dev, size_t n, zg_stream stream) {
    ZG_REQUIRE(fn >= 0 && fn <= 25, ZG_ERR_INVALID_ARGUMENT, "zg_devmath_apply: unknown function %d", fn);
    ZG_REQUIRE((x_dev && out_dev) || n == 0, ZG_ERR_INVALID_ARGUMENT, "zg_devmath_apply: null array");
    ZG_REQUIRE((fn != 6 && fn != 7) || y_dev || n == 0, ZG_ERR_INVALID_ARGUMENT, "zg_devmath_apply: function %d needs y", fn);
    if (n == 0) return ZG_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_devmath_apply, dim3(blocks), dim3(256), 0, as_stream(stream), fn, x_dev, y_dev, out_dev, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}


int zg_resize_convert(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const zg_method *method, const float *srgb_lut,
                      zg_stream stream) {
    return resize_convert_impl(src, src_space, dst, dst_space, method, srgb_lut, as_stream(stream));
}
int zg_resize_convert_host(const zg_image *src, int src_space, const zg_image *dst, int dst_space, const zg_method *method, const float *srgb_lut) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = resize_convert_impl(&a.dev, src_space, &b.dev, dst_space, method, srgb_lut, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
