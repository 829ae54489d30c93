// zg_sample.h — device-side point sampling: Image(T).interpolate for every pixel layout.
//
// Replaces reference src/image/interpolation.zig:72-84 (dispatch + finite/range guard), :306-311 (nearest),
// :313-407 (bilinear: fixed-point for u8 fields, lerpFloat for f32 fields), :426-519 (kernel interpolation)
// and the kernels :222-300. Arithmetic order, rounding and the absence of FMA follow the reference exactly;
// the only table (Lanczos3, built at comptime with Zig's @sin there) comes in from the host.
#pragma once
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

constexpr int WAVE_STAGE_ROWS = 5; // rows a wave stages for its 64 windows (interpolate<..., WAVE_STAGE>): windows at most one row apart

struct MethodArg {
    int kind;          // zg_interp
    float b, c;        // Mitchell
    const float *lut;  // device pointer, 1025-entry Lanczos3 table (nullptr unless kind == lanczos)
};

// resolveIndex on a 64-bit index (sampling coordinates can be astronomically large near a homography's
// horizon; the reference resolves them in isize). 32-bit fast path for everything realistic.
__device__ inline int resolve_index64(long long idx, int length, int border) {
    if (idx >= 0 && idx < (long long)length) return (int)idx;
    if (idx > -(1ll << 30) && idx < (1ll << 30)) return resolve_index((int)idx, length, border);
    switch (border) {
    case ZG_BORDER_ZERO: return -1;
    case ZG_BORDER_REPLICATE:
        if (length == 0) return -1;
        return idx < 0 ? 0 : length - 1;
    case ZG_BORDER_MIRROR: {
        if (length <= 0) return -1;
        if (length == 1) return 0;
        const long long period = 2ll * (length - 1);
        long long m = idx % period;
        if (m < 0) m += period;
        return (int)(m >= length ? period - m : m);
    }
    default: {
        if (length == 0) return -1;
        long long m = idx % (long long)length;
        if (m < 0) m += length;
        return (int)m;
    }
    }
}

// An integral-valued float coordinate (already floored / rounded) plus a small tap offset.
struct BaseIdx {
    float value; // integral-valued
    int narrow;
    bool is_narrow;
    __device__ explicit BaseIdx(float integral) {
        value = integral;
        is_narrow = fabsf(integral) < 1.0e9f;
        narrow = is_narrow ? (int)integral : 0;
    }
    __device__ int resolve(int offset, int length, int border) const {
        if (is_narrow) return resolve_index(narrow + offset, length, border);
        return resolve_index64((long long)value + offset, length, border); // f32 -> i64 only on this cold path
    }
};

// ---- kernels (interpolation.zig:222-300) ----------------------------------------------------
__device__ inline float bicubic_kernel(float t) {
    const float at = fabsf(t);
    if (at <= 1) return 1 - 2 * at * at + at * at * at;
    if (at <= 2) return 4 - 8 * at + 5 * at * at - at * at * at;
    return 0;
}
__device__ inline float catmull_rom_kernel(float x) {
    const float ax = fabsf(x);
    if (ax <= 1) return 1.5f * ax * ax * ax - 2.5f * ax * ax + 1;
    if (ax <= 2) return -0.5f * ax * ax * ax + 2.5f * ax * ax - 4 * ax + 2;
    return 0;
}
__device__ inline float mitchell_kernel(float x, float m_b, float m_c) {
    const float ax = fabsf(x), ax2 = ax * ax, ax3 = ax2 * ax;
    if (ax < 1) return ((12 - 9 * m_b - 6 * m_c) * ax3 + (-18 + 12 * m_b + 6 * m_c) * ax2 + (6 - 2 * m_b)) / 6;
    if (ax < 2)
        return ((-m_b - 6 * m_c) * ax3 + (6 * m_b + 30 * m_c) * ax2 + (-12 * m_b - 48 * m_c) * ax + (8 * m_b + 24 * m_c)) / 6;
    return 0;
}
__device__ inline float lanczos3_kernel_lut(const float *lut, float x) {
    const float ax = fabsf(x);
    if (ax >= 3.0f) return 0;
    const float step = (float)(1024.0 / 3.0);
    const float pos = ax * step;
    const int idx = (int)truncf(pos);
    const float frac = pos - (float)idx;
    return lut[idx] * (1.0f - frac) + lut[idx + 1] * frac;
}

// Taps of a radius-2 kernel sit at t = (i - 1) - f with f in [0, 1): |t| is in [1, 2] for i = 0, 3 and in [0, 1] for
// i = 1, 2, so for the two kernels whose branches are closed intervals (`<= 1`, `<= 2`) the branch taken is known per
// tap — except i = 0 at f == 0 (|t| == 1 takes the inner branch in the reference), where both polynomials evaluate to
// exactly +0. Evaluating only the polynomial that applies halves the weight arithmetic; the result is bit-identical.
// The x and y weights of a tap come from the same polynomial: evaluated as a pair (v_pk_mul_f32 / v_pk_add_f32), each half with
// the scalar expression's operations in the scalar expression's order.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND> __device__ inline f32x2 cubic_tap_weight2(int i, f32x2 t) {
    const f32x2 at = {fabsf(t.x), fabsf(t.y)};
    if constexpr (KIND == ZG_INTERP_BICUBIC) {
        if (i == 1 || i == 2) return (1.0f - (2.0f * at) * at) + (at * at) * at;
        return ((4.0f - 8.0f * at) + (5.0f * at) * at) - (at * at) * at;
    } else {
        if (i == 1 || i == 2) return (((1.5f * at) * at) * at - (2.5f * at) * at) + 1.0f;
        return ((((-0.5f * at) * at) * at + (2.5f * at) * at) - 4.0f * at) + 2.0f;
    }
}

template <int KIND> __device__ inline float eval_kernel(const MethodArg &m, float t) {
    if constexpr (KIND == ZG_INTERP_BICUBIC) return bicubic_kernel(t);
    else if constexpr (KIND == ZG_INTERP_CATMULL_ROM) return catmull_rom_kernel(t);
    else if constexpr (KIND == ZG_INTERP_MITCHELL) return mitchell_kernel(t, m.b, m.c);
    else return lanczos3_kernel_lut(m.lut, t);
}

// ---- interpolate ---------------------------------------------------------------------------
// Returns false for the reference's `null` (caller stores a zeroed pixel).
// `stage` (WAVE_STAGE kernels: Rgba(f32) with the radius-2 kernels): WAVE_STAGE_ROWS x 64 pixels of LDS owned by the calling WAVE. When all
// 64 lanes are here with their 4 x 4 windows inside the image, inside 64 consecutive columns and inside 5 consecutive rows (neighbouring
// pixels of a row of a mild warp or an enlargement), the wave loads those 5 x 64 pixels once — five coalesced 16-byte loads per lane instead
// of sixteen gathers — and every lane takes its taps out of LDS, a row at a time. Only where the pixels come from changes; measured, the sixteen gathers are what
// bounds the f32 bicubic warp (221 us; 210 us with the arithmetic halved, 160 us with the loads removed: profiles/r03_experiments.txt).
// WAVE_STAGE kernels keep few registers: where the wave cannot stage (a partial wave, windows too far apart, the image's rim) the taps
// are gathered a row at a time.
template <int PIX, int KIND, bool WAVE_STAGE = false>
__device__ inline bool interpolate(const DImg &img, float x, float y, const MethodArg &m, int border,
                                   typename Px<PIX>::Vec &out, typename Px<PIX>::Vec *stage = nullptr) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    using Elem = typename P::Elem;
    constexpr int C = P::C;
    constexpr bool IS_F = std::is_same<Elem, float>::value;

    const float range_limit = 4611686018427387904.0f; // @floatFromInt(maxInt(isize) / 2)
    if (!(fabsf(x) <= range_limit) || !(fabsf(y) <= range_limit)) return false; // also rejects NaN and +-inf

    if constexpr (KIND == ZG_INTERP_NEAREST) {
        const int col = BaseIdx(roundf(x)).resolve(0, img.cols, border);
        if (col < 0) return false;
        const int row = BaseIdx(roundf(y)).resolve(0, img.rows, border);
        if (row < 0) return false;
        out = P::load(img.data, (size_t)row * img.stride + (size_t)col);
        return true;
    } else if constexpr (KIND == ZG_INTERP_BILINEAR) {
        const float fl = floorf(x), ft = floorf(y);
        const BaseIdx left(fl), top(ft);
        Vec tl = P::zero(), tr = P::zero(), bl = P::zero(), br = P::zero();
        if (left.is_narrow && top.is_narrow && left.narrow >= 0 && left.narrow + 1 < img.cols && top.narrow >= 0 &&
            top.narrow + 1 < img.rows) { // all four neighbours inside: no index resolution, four independent loads
            const size_t o = (size_t)top.narrow * img.stride + (size_t)left.narrow;
            tl = P::load(img.data, o);
            tr = P::load(img.data, o + 1);
            bl = P::load(img.data, o + img.stride);
            br = P::load(img.data, o + img.stride + 1);
        } else {
            const int r0 = top.resolve(0, img.rows, border), r1 = top.resolve(1, img.rows, border);
            const int c0 = left.resolve(0, img.cols, border), c1 = left.resolve(1, img.cols, border);
            if (border == ZG_BORDER_MIRROR && (r0 < 0 || r1 < 0 || c0 < 0 || c1 < 0)) return false;
            if (r0 >= 0 && c0 >= 0) tl = P::load(img.data, (size_t)r0 * img.stride + (size_t)c0);
            if (r0 >= 0 && c1 >= 0) tr = P::load(img.data, (size_t)r0 * img.stride + (size_t)c1);
            if (r1 >= 0 && c0 >= 0) bl = P::load(img.data, (size_t)r1 * img.stride + (size_t)c0);
            if (r1 >= 0 && c1 >= 0) br = P::load(img.data, (size_t)r1 * img.stride + (size_t)c1);
        }
        const float lr = x - fl, tb = y - ft; // as(f32, left) == floorf(x) for every finite in-range x
        if constexpr (IS_F) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                out[ch] = (1 - tb) * ((1 - lr) * tl[ch] + lr * tr[ch]) + tb * ((1 - lr) * bl[ch] + lr * br[ch]);
        } else {
            const int fx = (int)roundf(lr * 256), fy = (int)roundf(tb * 256);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const int top_val = (int)tl[ch] * (256 - fx) + (int)tr[ch] * fx;
                const int bottom_val = (int)bl[ch] * (256 - fx) + (int)br[ch] * fx;
                const int result = (top_val * (256 - fy) + bottom_val * fy + 32768) / 65536; // >= 0: trunc == floor
                out[ch] = clamp_u8_i32(result);
            }
        }
        return true;
    } else {
        constexpr int R = KIND == ZG_INTERP_LANCZOS ? 3 : 2;
        constexpr int W = 2 * R;
        const float flx = floorf(x), fly = floorf(y);
        const BaseIdx ix(flx), iy(fly);
        const float fx = x - flx, fy = y - fly;
        float xw[W], yw[W];
        auto make_weights = [&](float wfx, float wfy) {
#pragma unroll
            for (int i = 0; i < W; ++i) {
                if constexpr (KIND == ZG_INTERP_BICUBIC || KIND == ZG_INTERP_CATMULL_ROM) {
                    const f32x2 f = {wfx, wfy};
                    const f32x2 w = cubic_tap_weight2<KIND>(i, (float)(i - (R - 1)) - f);
                    xw[i] = w.x;
                    yw[i] = w.y;
                } else {
                    xw[i] = eval_kernel<KIND>(m, (float)(i - (R - 1)) - wfx);
                    yw[i] = eval_kernel<KIND>(m, (float)(i - (R - 1)) - wfy);
                }
            }
        };
        // WAVE_STAGE: the staged path makes them AFTER it has issued its loads (they do not depend on the pixels), the others where they start
        if constexpr (!WAVE_STAGE) make_weights(fx, fy);
        float sums[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) sums[ch] = 0;
        float weight_sum = 0;
        const int bx = ix.narrow - (R - 1), by = iy.narrow - (R - 1);
        // pixels of 4 or 16 bytes in images below 4 GiB are gathered with buffer loads; larger images take the general path
        constexpr int PB = P::BYTES;
        constexpr bool BUF = PB == 4 || PB == 16;
        const size_t img_bytes = (size_t)img.rows * img.stride * PB;
        if (ix.is_narrow && iy.is_narrow && bx >= 0 && bx + W <= img.cols && by >= 0 && by + W <= img.rows && (!BUF || img_bytes < (1ull << 32))) {
            // the whole W x W window is inside the image (every pixel but a thin rim): straight-line code, all
            // W*W gathers independent and in flight together, same accumulation order as the general path
            Vec px[W][W];
            bool staged = false;
            if constexpr (WAVE_STAGE && BUF && PB == 16 && W == 4) {
                if (__builtin_amdgcn_ballot_w64(true) == ~0ull) { // the whole wave is here
                    const int bx0 = min(__builtin_amdgcn_readlane(bx, 0), __builtin_amdgcn_readlane(bx, 63));
                    const int by0 = min(__builtin_amdgcn_readlane(by, 0), __builtin_amdgcn_readlane(by, 63));
                    const bool fits = (unsigned)(bx - bx0) <= 60u && (unsigned)(by - by0) <= (unsigned)(WAVE_STAGE_ROWS - 4);
                    if (__builtin_amdgcn_ballot_w64(fits) == ~0ull) { // wave-uniform
                        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img.data, (short)0, (int)(uint32_t)img_bytes, 0x00020000);
                        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                        // columns / rows past the image are clamped: no window reaches them (every window is inside)
                        const int voff = (by0 * (int)img.stride + min(bx0 + lane, img.cols - 1)) * PB;
                        Vec fetched[WAVE_STAGE_ROWS];
#pragma unroll
                        for (int j = 0; j < WAVE_STAGE_ROWS; ++j) {
                            const int soff = (min(by0 + j, img.rows - 1) - by0) * (int)img.stride * PB;
                            fetched[j] = __builtin_bit_cast(Vec, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
                        }
                        {
                            float wfx = fx, wfy = fy;
                            asm volatile("" : "+v"(wfx), "+v"(wfy)); // after the loads above, in program order
                            make_weights(wfx, wfy);
                        }
#pragma unroll
                        for (int j = 0; j < WAVE_STAGE_ROWS; ++j) stage[j * 64 + lane] = fetched[j];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier(); // LDS operations of one wave execute in order; this keeps the compiler from reordering them
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const Vec *mine = stage + (by - by0) * 64 + (bx - bx0);
                        // row by row, a row's four taps read just before they are used (the scheduling barrier keeps hipcc from hoisting all
                        // sixteen reads to the top: 64 registers of taps cost two waves per SIMD of occupancy)
#pragma unroll
                        for (int j = 0; j < W; ++j) {
                            Vec row[W];
#pragma unroll
                            for (int i = 0; i < W; ++i) row[i] = mine[j * 64 + i];
#pragma unroll
                            for (int i = 0; i < W; ++i) {
                                const float weight = xw[i] * yw[j];
#pragma unroll
                                for (int ch = 0; ch < C; ++ch) {
                                    const float prod = (float)row[i][ch] * weight;
                                    sums[ch] = sums[ch] + prod;
                                }
                                weight_sum = weight_sum + weight;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        staged = true;
                    }
                }
            }
            if constexpr (WAVE_STAGE && BUF && PB == 16 && W == 4) {
                if (!staged) {
                    make_weights(fx, fy);
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img.data, (short)0, (int)(uint32_t)img_bytes, 0x00020000);
                    const int voff = (by * img.stride + bx) * PB;
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const int soff = j * img.stride * PB;
                        Vec row[W];
#pragma unroll
                        for (int i = 0; i < W; ++i) row[i] = __builtin_bit_cast(Vec, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + i * PB, soff, 0));
#pragma unroll
                        for (int i = 0; i < W; ++i) {
                            const float weight = xw[i] * yw[j];
#pragma unroll
                            for (int ch = 0; ch < C; ++ch) {
                                const float prod = (float)row[i][ch] * weight;
                                sums[ch] = sums[ch] + prod;
                            }
                            weight_sum = weight_sum + weight;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    staged = true; // the sums are made
                }
            }
            if (staged) {
            } else if constexpr (BUF) {
                // buffer loads: one 32-bit offset per lane, the row of a tap in the scalar offset, its column in the immediate:
                // sixteen gathers without a single address instruction
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img.data, (short)0, (int)(uint32_t)img_bytes, 0x00020000);
                const int voff = (by * img.stride + bx) * PB;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const int soff = j * img.stride * PB;
                    if constexpr (PB == 4 && W == 4) {
                        // the four taps of a window row are sixteen adjacent bytes: ONE gather per row (dword-aligned 16-byte buffer load) instead of
                        // four — a gather costs the texture path per lane ADDRESS, not per byte
                        typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
                        const u32x4s q = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
#pragma unroll
                        for (int i = 0; i < W; ++i) {
                            const uint32_t tap = q[i]; // (a bit_cast of the element expression itself reads element 0: clang)
                            px[j][i] = __builtin_bit_cast(Vec, tap);
                        }
                    } else if constexpr (PB == 4 && W == 6) { // Lanczos3: sixteen bytes and eight
                        typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
                        typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
                        const u32x4s q = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
                        const u32x2s q2 = __builtin_bit_cast(u32x2s, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + 16, soff, 0));
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t tap = q[i];
                            px[j][i] = __builtin_bit_cast(Vec, tap);
                        }
                        const uint32_t t4 = q2[0], t5 = q2[1];
                        px[j][4] = __builtin_bit_cast(Vec, t4);
                        px[j][5] = __builtin_bit_cast(Vec, t5);
                    } else {
#pragma unroll
                        for (int i = 0; i < W; ++i) {
                            if constexpr (PB == 4) px[j][i] = __builtin_bit_cast(Vec, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + i * PB, soff, 0));
                            else px[j][i] = __builtin_bit_cast(Vec, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + i * PB, soff, 0));
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < W; ++j)
#pragma unroll
                    for (int i = 0; i < W; ++i) px[j][i] = P::load(img.data, (size_t)(by + j) * img.stride + (size_t)(bx + i));
            }
            if (!staged) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const float weight = xw[i] * yw[j];
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float prod = (float)px[j][i][ch] * weight;
                        sums[ch] = sums[ch] + prod;
                    }
                    weight_sum = weight_sum + weight;
                }
            }
            }
            if constexpr (!IS_F) {
                // The C quotients share their divisor: the refined reciprocal of the IEEE division sequence (v_rcp + one Newton
                // step) is computed once and each quotient is the sequence's remaining five operations, bit for bit what `/`
                // expands to when v_div_scale has nothing to scale (sums of 8-bit samples over a weight sum near 1).
                if (weight_sum > 0.5f && weight_sum < 2.0f) {
                    const float nd = -weight_sum, r0 = __builtin_amdgcn_rcpf(weight_sum);
                    const float r1 = __builtin_fmaf(__builtin_fmaf(nd, r0, 1.0f), r0, r0);
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float q0 = sums[ch] * r1;
                        const float q1 = __builtin_fmaf(__builtin_fmaf(nd, q0, sums[ch]), r1, q0);
                        const float val = __builtin_fmaf(__builtin_fmaf(nd, q1, sums[ch]), r1, q1); // finite: no NaN case to map
                        const float u = fminf(fmaxf(val, 0.0f), 255.0f), t = truncf(u);
                        out[ch] = (uint8_t)((int)t + ((u - t) >= 0.5f ? 1 : 0));
                    }
                    return true;
                }
            }
        } else {
            if constexpr (WAVE_STAGE) make_weights(fx, fy);
            int cols_idx[W];
#pragma unroll
            for (int i = 0; i < W; ++i) cols_idx[i] = ix.resolve(i - (R - 1), img.cols, border);
#pragma unroll 1
            for (int j = 0; j < W; ++j) {
                const int py = iy.resolve(j - (R - 1), img.rows, border);
                if (py < 0) continue;
                const size_t rowoff = (size_t)py * img.stride;
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    if (cols_idx[i] < 0) continue;
                    const Vec p = P::load(img.data, rowoff + (size_t)cols_idx[i]);
                    const float weight = xw[i] * yw[j];
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float prod = (float)p[ch] * weight;
                        sums[ch] = sums[ch] + prod;
                    }
                    weight_sum = weight_sum + weight;
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const float val = weight_sum != 0 ? sums[ch] / weight_sum : 0.0f;
            if constexpr (IS_F) out[ch] = val;
            else out[ch] = clamp_u8_f32(val);
        }
        return true;
    }
}

} // namespace zg
