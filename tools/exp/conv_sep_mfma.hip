// ARCHIVED EXPERIMENT (round 4, out of libzignal_hip.so since round 5): the u8 Gaussian with both passes on the matrix pipe. It is bit-exact and loses
// (92 us against 30: profiles/r04_mfma_blur.txt), and north_star rules MFMA out for this path. To run it again, copy it back into zignal_amd/csrc/ and
// call try_sep_mfma from conv_separable.hip ahead of try_sep_stream, as commit 4d02f76 did (exp_mfma.py beside this file is its driver).
// conv_sep_mfma.hip — the u8 separable convolution with both passes on the matrix pipe (v_mfma_i32_16x16x64_i8), exact in i32.
//
// Same arithmetic contract as conv_sep_stream.hip (reference src/image/convolution.zig:441-647, u8 path: taps round(k * 256) in
// [0, 255], every channel the same taps, i32-exact sums, divClampU8(65536)): integer sums are order-free, so a pass of the
// convolution is a product with a banded Toeplitz matrix, and the result bits cannot depend on who multiplies. What the VALU
// kernels spend 7.6 instructions per byte on (unpack, one packed multiply-add per byte and tap, repack) becomes:
//   row pass     D1[r][c] = sum_k  A1[r][k] * B1[k][c]     A1 = 16 source rows x 64 bytes exactly as they lie in memory (a lane's
//                dwordx4 IS its fragment of A: row = lane & 15, bytes 16 * (lane >> 4) ..), pixels - 128 (one v_xor per dword);
//                B1 = the x taps on a band, NT tiles of 16 output bytes per 64-byte window (the window's halo bytes have no
//                output); the i32 accumulator starts at 128 * sum(kx) - 32768, so D1 = temp - 32768 fits i16: X = 256 Hs + L
//   column pass  D2[c][r] = sum_k  A2[c][k] * B2[k][r]     A2 = D1's own registers (a lane holds four rows of one column as
//                four dwords [L, Hs, sign, sign]; with v_xor 0x80 the low byte is L - 128): K runs over 16 rows x 4 bytes and
//                B2 = the y taps placed on the L bytes (one MFMA) or on the Hs bytes (a second one), zeros elsewhere;
//                acc = (D2_hi << 8) + D2_lo, every constant (the 128s, 32768 * sum, the rounding half) in D2_lo's start value;
//   the result byte is bits 16..23 of acc: one v_lshl_add per byte, two v_perm and a v_or per four bytes, one store per lane.
// A tap above 127 does not fit the signed operand: such kernels run every product as two MFMAs on the halves (SPLIT).
// That is ~3 VALU lane-instructions per byte; the MFMAs (0.5 cycles of the matrix pipe per byte) run beside them.
//
// Work decomposition: a unit is 16 source rows x one 64-byte window -> (16 - 2H) output rows x 16 NT output bytes. A wave walks
// a run of units along a band of rows; border columns are folded into the first / last unit's Toeplitz matrix (mirror,
// replicate and zero resolve inside the window), border rows are resolved per lane when the band is set up.
// Column index i of tile n stands for output byte 4 NT (i >> 2) + 4 n + (i & 3): after the column pass lane (row, g) then holds
// 4 NT consecutive bytes of its row, one store.
//
// Preconditions (else the other kernels run): u8 pixel types, odd equal tap counts with H * SP <= 16, taps in [0, 255] with
// folded weights <= 254 and sums <= 257, border != wrap, row bytes % 16 == 0 and >= 64, spans below 2 GiB.
#include "zg_common.h"
#include "zg_u8pack.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace zg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2m __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3m __attribute__((ext_vector_type(3)));

struct MfmaArgs {
    const uint8_t *src;
    uint8_t *dst;
    uint64_t src_frame, dst_frame; // bytes between frames
    uint32_t src_pitch, dst_pitch; // bytes between rows
    uint32_t src_span, dst_span;   // bytes from a frame's first byte to the end of its last row
    int32_t rows, row_bytes;
    int32_t units_x;   // units across a row
    int32_t segs_x;    // waves across a row
    int32_t seg_units; // units per wave
    int32_t bands;     // bands per frame
    int32_t border;
    int32_t c2;        // start value of the low column-pass accumulator
    int32_t half;      // H: taps reach this many rows / pixels to each side
    const v4i *tab;    // Toeplitz fragments, see MfmaTables
};

// Table layout in v4i units (one entry per lane):
//   B1 [variant 3: LEFT, MID, RIGHT][half 2][tile NT][64]     x taps, K slot (g, e) = window byte 16 g + e, column lane & 15
//   T2 [select 2: L bytes, Hs bytes][half 2][64]              y taps, K slot (g, 4 v + b) = temp row 4 g + v byte b, column = output row
//   C1 [variant 3][tile NT][64] as int32 (16 entries per v4i slot row)  row-pass accumulator start per column
template <int NT> struct MfmaLayout {
    static constexpr int B1 = 0;
    static constexpr int T2 = 3 * 2 * NT * 64;
    static constexpr int C1 = T2 + 2 * 2 * 64;        // in v4i units; holds 3 * NT * 64 int32 = 3 * NT * 16 v4i
    static constexpr int TOTAL = C1 + 3 * NT * 16;
};

__device__ __forceinline__ v4i mfma_i8(v4i a, v4i b, v4i c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }

template <int NT> __device__ __forceinline__ void st_lane(const uint32_t (&o)[NT], __amdgpu_buffer_rsrc_t r, int off) {
    if constexpr (NT == 3) __builtin_amdgcn_raw_buffer_store_b96(u32x3m{o[0], o[1], o[2]}, r, off, 0, 2);
    else if constexpr (NT == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2m{o[0], o[1]}, r, off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(o[0], r, off, 0, 2);
}

template <int NT, bool SPLIT, bool CLAMP, int PD, int EXP = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_sep_mfma(MfmaArgs a) {
    constexpr int NS = SPLIT ? 2 : 1;
    const int H = a.half, OR = 16 - 2 * H; // output rows of a band
    constexpr int UW = 16 * NT, OFF_MID = (64 - UW) / 2;
    typedef MfmaLayout<NT> L;

    const int lane = (int)threadIdx.x, li = lane & 15, lg = lane >> 4;
    const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3;
    uint32_t w = blockIdx.x;
    if (ZG_XCD_ORDER && w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3); // XCD-major: an XCD's L2 sees neighbouring bands
    const uint32_t per_frame = (uint32_t)(a.bands * a.segs_x);
    const uint32_t frame = w / per_frame, t = w - frame * per_frame;
    const int band = (int)(t / (uint32_t)a.segs_x), seg = (int)(t - (uint32_t)band * (uint32_t)a.segs_x);
    const uint8_t *srcf = a.src + (size_t)frame * a.src_frame;
    uint8_t *dstf = a.dst + (size_t)frame * a.dst_frame;
    const auto src_all = __builtin_amdgcn_make_buffer_rsrc((void *)srcf, (short)0, (int)a.src_span, 0x00020000);
    const auto dst_all = __builtin_amdgcn_make_buffer_rsrc((void *)dstf, (short)0, (int)a.dst_span, 0x00020000);
    const auto tab_all = __builtin_amdgcn_make_buffer_rsrc((void *)a.tab, (short)0, L::TOTAL * 16, 0x00020000);

    const int U = a.units_x, rb = a.row_bytes;
    const int u_begin = seg * a.seg_units, u_end = min(u_begin + a.seg_units, U);
    const int yb = band * OR;

    // this lane's source row (fragment row lane & 15) and its 16-byte chunk of the window
    const int y = yb - H + li;
    int ry = y;
    uint32_t keep = ~0u; // 0 for a row the zero border drops
    const bool edge_band = yb - H < 0 || yb - H + 16 > a.rows; // wave-uniform
    if (edge_band) {
        const int r = resolve_index(y, a.rows, a.border);
        keep = r >= 0 ? ~0u : 0u;
        ry = max(r, 0);
    }
    const int ld_off = (int)((uint32_t)ry * a.src_pitch) + 16 * lg;
    // ... and the output row (column lane & 15 of the column pass) with this lane group's 4 NT bytes of the unit
    const int orow = yb + li;
    const bool st_ok = li < OR && orow < a.rows;
    const int st_off = st_ok ? (int)((uint32_t)orow * a.dst_pitch) + 4 * NT * lg : (int)0x80000000; // out of range: the store is dropped

    auto tab_v4 = [&](int idx) -> v4i { return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(tab_all, (idx * 64 + lane) * 16, 0, 0)); };
    auto tab_c1 = [&](int var, int n) -> int { return (int)__builtin_amdgcn_raw_buffer_load_b32(tab_all, L::C1 * 16 + ((var * NT + n) * 64 + lane) * 4, 0, 0); };

    v4i T2[2][NS];
#pragma unroll
    for (int sel = 0; sel < 2; ++sel)
#pragma unroll
        for (int s = 0; s < NS; ++s) T2[sel][s] = tab_v4(L::T2 / 64 + sel * 2 + s);
    const v4i c2 = {a.c2, a.c2, a.c2, a.c2};
    const v4i zero = {0, 0, 0, 0};

    // one unit: `av` is the lane's chunk of the window (pixels - 128), B1 / C1 the fragments of the window's variant, c0 the unit's first output byte
    auto prepare = [&](u32x4 raw) -> v4i {
        if (edge_band) { raw[0] &= keep; raw[1] &= keep; raw[2] &= keep; raw[3] &= keep; } // wave-uniform
        return __builtin_bit_cast(v4i, raw ^ 0x80808080u);
    };
    auto unit = [&](v4i av, const v4i (&B1)[NS][NT], const int (&C1)[NT], int c0) {
        uint32_t o[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            v4i d1 = mfma_i8(av, B1[0][n], v4i{C1[n], C1[n], C1[n], C1[n]});
            if constexpr (SPLIT) d1 = mfma_i8(av, B1[1][n], d1);
            const v4i a2 = d1 ^ 0x80;
            v4i lo = mfma_i8(a2, T2[0][0], c2);
            if constexpr (SPLIT) lo = mfma_i8(a2, T2[0][1], lo);
            v4i hi = mfma_i8(a2, T2[1][0], zero);
            if constexpr (SPLIT) hi = mfma_i8(a2, T2[1][1], hi);
            uint32_t x[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                x[v] = ((uint32_t)hi[v] << 8) + (uint32_t)lo[v];
                if constexpr (CLAMP) x[v] = min(x[v], 0x00ffffffu);
            }
            o[n] = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0602u) | __builtin_amdgcn_perm(x[3], x[2], 0x06020c0cu);
        }
        if constexpr (EXP & 1) { if (o[0] != 0x12345678u || o[NT - 1] != 0x9abcdef0u) return; } // experiments: the kernel without its stores
        st_lane<NT>(o, dst_all, st_off + c0);
    };
    auto load_window = [&](int xs) -> u32x4 {
        if constexpr (EXP & 2) return u32x4{(uint32_t)(xs + lane), (uint32_t)(xs ^ lane), (uint32_t)xs * 2654435761u, (uint32_t)lane};
        return __builtin_amdgcn_raw_buffer_load_b128(src_all, ld_off, xs, 0);
    };
    auto edge_unit = [&](int var, int xs, int c0) {
        v4i B1[NS][NT];
        int C1[NT];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n) B1[s][n] = tab_v4((var * 2 + s) * NT + n);
#pragma unroll
        for (int n = 0; n < NT; ++n) C1[n] = tab_c1(var, n);
        unit(prepare(load_window(xs)), B1, C1, c0);
    };

    if (u_begin == 0) edge_unit(0, 0, 0);
    const int ui0 = max(u_begin, 1), ui1 = min(u_end, U - 1); // the units whose windows touch neither end of the rows
    if (ui0 < ui1) {
        v4i B1[NS][NT];
        int C1[NT];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n) B1[s][n] = tab_v4((1 * 2 + s) * NT + n);
#pragma unroll
        for (int n = 0; n < NT; ++n) C1[n] = tab_c1(1, n);
        const int last = ui1 - 1;
        auto xs_of = [&](int u) { return UW * min(u, last) - OFF_MID; };
        u32x4 ring[PD]; // the windows of the next PD units, in flight
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            ring[p] = load_window(xs_of(ui0 + p));
            __builtin_amdgcn_sched_barrier(0);
        }
        const int n_units = ui1 - ui0, n_main = n_units / PD * PD;
        auto block = [&](int u) {
#pragma unroll
            for (int p = 0; p < PD; ++p) {
                const v4i av = prepare(ring[p]);
                // the slot's registers are dead from here on: the window PD units ahead is asked for now and lands in the very registers it
                // replaces (asked for earlier it would need registers of its own, and the loop would rotate the slots by moves at its back
                // edge, moves that wait for the loads)
                __builtin_amdgcn_sched_barrier(0);
                ring[p] = load_window(xs_of(u + p + PD));
                __builtin_amdgcn_sched_barrier(0);
                unit(av, B1, C1, UW * (u + p));
            }
        };
        // The first block is peeled off the loop: the wait-count pass merges what is in flight at the loop's entry with what is in flight
        // at its back edge and waits for the worse of the two, and after PD bare loads a slot has PD - 1 younger requests where the
        // steady state has 2 PD - 1 (the stores count too) — the loop would run with half its read-ahead.
        int u = ui0;
        if (n_main > 0) {
            block(u);
            for (u += PD; u < ui0 + n_main; u += PD) block(u);
        }
#pragma unroll
        for (int p = 0; p < PD - 1; ++p)
            if (p < n_units - n_main) unit(prepare(ring[p]), B1, C1, UW * (u + p)); // wave-uniform
    }
    if (u_end == U) edge_unit(2, rb - 64, rb - UW);
}

// ---- host side: the Toeplitz fragments, cached on the device per (taps, pixel stride, border, row length) --------------------------------
struct MfmaKey {
    int dev, sp, nk, nt, border;
    uint32_t cols;
    uint8_t kx[9], ky[9];
    bool operator<(const MfmaKey &o) const { return memcmp(this, &o, sizeof(MfmaKey)) < 0; }
};
struct MfmaBuf {
    v4i *dev = nullptr;
    bool split = false;
    ~MfmaBuf() { if (dev) (void)hipFree(dev); }
};
static std::mutex g_mfma_mu;
static std::map<MfmaKey, std::pair<std::shared_ptr<MfmaBuf>, std::list<MfmaKey>::iterator>> g_mfma_cache;
static std::list<MfmaKey> g_mfma_order;

// Returns false when a folded weight does not fit the (split) signed operand.
template <int NT> static bool build_mfma_tables(int sp, int nk, const int32_t *ix, const int32_t *iy, int border, uint32_t cols, std::vector<int32_t> &out, bool &split) {
    typedef MfmaLayout<NT> L;
    const int h = nk / 2, uw = 16 * NT, off_mid = (64 - uw) / 2, rb = (int)(cols * (uint32_t)sp);
    // W[variant][tile][k][column]
    std::vector<int> W((size_t)3 * NT * 64 * 16, 0);
    auto Wat = [&](int var, int n, int k, int jc) -> int & { return W[(((size_t)var * NT + n) * 64 + k) * 16 + jc]; };
    for (int var = 0; var < 3; ++var) {
        // the window's first byte and the unit's first output byte, as row offsets; MID stands for any unit away from both ends
        const int c0 = var == 0 ? 0 : var == 2 ? rb - uw : 4096 * sp * 16, xs = var == 0 ? 0 : var == 2 ? rb - 64 : c0 - off_mid;
        for (int n = 0; n < NT; ++n)
            for (int jc = 0; jc < 16; ++jc) {
                const int ob = 4 * NT * (jc >> 2) + 4 * n + (jc & 3);
                for (int t = 0; t < nk; ++t) {
                    const int p = c0 + ob + sp * (t - h);
                    int px = p >= 0 ? p / sp : -((-p + sp - 1) / sp);
                    const int ch = p - px * sp;
                    if (var != 1) {
                        px = resolve_index(px, (int)cols, border);
                        if (px < 0) continue; // the zero border drops the tap
                    }
                    const int k = px * sp + ch - xs;
                    if (k < 0 || k >= 64) return false; // cannot happen for mirror / replicate / zero with cols >= 64
                    Wat(var, n, k, jc) += ix[t];
                }
            }
    }
    int wmax = 0;
    for (int v : W) wmax = std::max(wmax, v);
    for (int t = 0; t < nk; ++t) wmax = std::max(wmax, iy[t]);
    if (wmax > 254) return false;
    split = wmax > 127;
    out.assign((size_t)L::TOTAL * 4, 0);
    auto put = [&](int v4_index, int ln, int e, int wv) { // byte e of lane ln's fragment
        int32_t &d = out[((size_t)v4_index * 64 + ln) * 4 + (e >> 2)];
        d |= (int32_t)((uint32_t)(uint8_t)(int8_t)wv << (8 * (e & 3)));
    };
    for (int var = 0; var < 3; ++var)
        for (int n = 0; n < NT; ++n)
            for (int ln = 0; ln < 64; ++ln) {
                const int jc = ln & 15, g = ln >> 4;
                int sum = 0;
                for (int k = 0; k < 64; ++k) sum += Wat(var, n, k, jc);
                for (int e = 0; e < 16; ++e) {
                    const int wv = Wat(var, n, 16 * g + e, jc), wa = split ? wv / 2 : wv, wb = wv - wa;
                    put((var * 2 + 0) * NT + n, ln, e, wa);
                    put((var * 2 + 1) * NT + n, ln, e, wb);
                }
                out[(size_t)L::C1 * 4 + ((size_t)var * NT + n) * 64 + ln] = 128 * sum - 32768;
            }
    for (int sel = 0; sel < 2; ++sel)
        for (int ln = 0; ln < 64; ++ln) {
            const int j2 = ln & 15, g = ln >> 4;
            for (int v = 0; v < 4; ++v) {
                const int d = 4 * g + v - j2;
                if (d < 0 || d >= nk) continue;
                const int wv = iy[d], wa = split ? wv / 2 : wv, wb = wv - wa;
                put(L::T2 / 64 + sel * 2 + 0, ln, 4 * v + sel, wa);
                put(L::T2 / 64 + sel * 2 + 1, ln, 4 * v + sel, wb);
            }
        }
    return true;
}

template <int NT> static int mfma_tables(const MfmaKey &key, const int32_t *ix, const int32_t *iy, hipStream_t s, std::shared_ptr<MfmaBuf> &hold) {
    std::shared_ptr<MfmaBuf> evicted;
    std::lock_guard<std::mutex> lock(g_mfma_mu);
    auto it = g_mfma_cache.find(key);
    if (it != g_mfma_cache.end()) {
        g_mfma_order.splice(g_mfma_order.end(), g_mfma_order, it->second.second);
        hold = it->second.first;
        return hold->dev ? ZG_OK : -1;
    }
    auto buf = std::make_shared<MfmaBuf>();
    std::vector<int32_t> host;
    const bool ok = build_mfma_tables<NT>(key.sp, key.nk, ix, iy, key.border, key.cols, host, buf->split);
    if (ok) {
        // a first call inside a stream capture cannot allocate or synchronise: it takes the other kernels, and so does every later
        // call of that key until one arrives outside a capture (nothing is cached for it here)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return -1; }
        ZG_HIP(hipMalloc((void **)&buf->dev, host.size() * sizeof(int32_t)));
        if (int rc = upload_pageable(buf->dev, host.data(), host.size() * sizeof(int32_t), nullptr)) return rc;
    }
    if (g_mfma_cache.size() >= 64) {
        auto old = g_mfma_cache.find(g_mfma_order.front());
        evicted = std::move(old->second.first);
        g_mfma_cache.erase(old);
        g_mfma_order.pop_front();
    }
    g_mfma_order.push_back(key);
    g_mfma_cache.emplace(key, std::make_pair(buf, std::prev(g_mfma_order.end())));
    hold = buf;
    return ok ? ZG_OK : -1;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int NT> static int launch_mfma(const StreamJob &j, const int32_t *ix, const int32_t *iy, int NK, int border, bool clamp, hipStream_t s) {
    constexpr int UW = 16 * NT;
    const int OR = 16 - 2 * (NK / 2);
    MfmaKey key;
    memset(&key, 0, sizeof key);
    ZG_HIP(hipGetDevice(&key.dev));
    key.sp = j.sp; key.nk = NK; key.nt = NT; key.border = border; key.cols = j.cols;
    for (int i = 0; i < NK; ++i) { key.kx[i] = (uint8_t)ix[i]; key.ky[i] = (uint8_t)iy[i]; }
    std::shared_ptr<MfmaBuf> hold;
    if (int rc = mfma_tables<NT>(key, ix, iy, s, hold)) return rc;

    MfmaArgs a;
    a.src = (const uint8_t *)j.src; a.dst = (uint8_t *)j.dst;
    a.src_frame = j.src_frame; a.dst_frame = j.dst_frame;
    a.src_pitch = (uint32_t)j.src_pitch; a.dst_pitch = (uint32_t)j.dst_pitch;
    a.rows = (int32_t)j.rows;
    a.row_bytes = (int32_t)(j.cols * (uint32_t)j.sp);
    a.src_span = (uint32_t)((uint64_t)(j.rows - 1) * j.src_pitch + (uint64_t)a.row_bytes);
    a.dst_span = (uint32_t)((uint64_t)(j.rows - 1) * j.dst_pitch + (uint64_t)a.row_bytes);
    a.units_x = (int32_t)ceil_div((unsigned)a.row_bytes, (unsigned)UW);
    a.bands = (int32_t)ceil_div(j.rows, (unsigned)OR);
    // units per wave: enough waves for ~6 per SIMD of the chip when one frame has to fill it, long runs for batches
    const uint64_t all_units = (uint64_t)a.units_x * a.bands * j.n_frames;
    int su = (int)std::min<uint64_t>(std::max<uint64_t>((all_units + 6143) / 6144, 8), 64);
    su = env_int("ZIGNAL_HIP_MFMA_SEG", su);
    a.seg_units = std::max(1, su);
    a.segs_x = (int32_t)ceil_div((unsigned)a.units_x, (unsigned)a.seg_units);
    a.border = border;
    int64_t sy = 0;
    for (int i = 0; i < NK; ++i) sy += iy[i];
    a.c2 = (int32_t)(sy * (128 + 32768) + 32768);
    a.tab = hold->dev;
    a.half = NK / 2;
    const uint64_t items = (uint64_t)a.bands * a.segs_x * j.n_frames;
    if (items > 0x7fffffffu) return -1;
    const dim3 grid((unsigned)items), block(64);
    if (const int ex = env_int("ZIGNAL_HIP_MFMA_EXP", 0)) { // experiments (profiles/r04_mfma_blur.txt): 1 = no stores, 2 = no loads, 3 = neither
        if constexpr (NT == 3) if (hold->split && !clamp) {
            if (ex == 1) hipLaunchKernelGGL((k_sep_mfma<3, true, false, 4, 1>), grid, block, 0, s, a);
            else if (ex == 2) hipLaunchKernelGGL((k_sep_mfma<3, true, false, 4, 2>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_sep_mfma<3, true, false, 4, 3>), grid, block, 0, s, a);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        }
    }
#define ZG_MF(SPLIT, CLAMP) hipLaunchKernelGGL((k_sep_mfma<NT, SPLIT, CLAMP, 4>), grid, block, 0, s, a)
    if (hold->split) { if (clamp) ZG_MF(true, true); else ZG_MF(true, false); }
    else { if (clamp) ZG_MF(false, true); else ZG_MF(false, false); }
#undef ZG_MF
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Returns -1 when the preconditions do not hold (the caller falls back to the VALU kernels).
int try_sep_mfma(const StreamJob &j, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s) {
    // Tuning hook, off by default: bit-exact, but the fragment layout makes its loads and stores 16- and 12-byte pieces, one row per lane, and it runs at a
    // third of k_sep_stream's speed (profiles/r04_mfma_blur.txt). Kept as the record of that experiment and as the base of an LDS-staged form.
    static const bool on = getenv("ZIGNAL_HIP_MFMA") != nullptr;
    if (!on) return -1;
    if (j.down2) return -1;
    if (j.sp != 1 && j.sp != 3 && j.sp != 4) return -1;
    if (nk != 3 && nk != 5 && nk != 7 && nk != 9) return -1;
    if (border == ZG_BORDER_WRAP) return -1;
    const int hs = nk / 2 * j.sp;
    if (hs > 16) return -1;
    const uint64_t rb = (uint64_t)j.cols * (uint64_t)j.sp;
    if (rb % 16 || rb < 64 || j.cols < 64 || j.rows < 16) return -1;
    if (j.src_pitch % 4 || j.dst_pitch % 4 || j.src_frame % 4 || j.dst_frame % 4 || ((uintptr_t)j.src & 3) || ((uintptr_t)j.dst & 3)) return -1;
    const uint64_t sspan = (uint64_t)(j.rows - 1) * j.src_pitch + rb, dspan = (uint64_t)(j.rows - 1) * j.dst_pitch + rb;
    if (sspan > 0x7fffffffu || dspan > 0x7fffffffu) return -1;
    int64_t sx = 0, sy = 0;
    for (int i = 0; i < nk; ++i) {
        if (ix[i] < 0 || ix[i] > 255 || iy[i] < 0 || iy[i] > 255) return -1;
        sx += ix[i];
        sy += iy[i];
    }
    if (sx > 257 || sy > 257) return -1;
    const bool clamp = sx * sy * 255 + 32768 >= 256 * 65536;
    const bool nt3 = hs <= 8 && !getenv("ZIGNAL_HIP_MFMA_NT2");
    return nt3 ? launch_mfma<3>(j, ix, iy, nk, border, clamp, s) : launch_mfma<2>(j, ix, iy, nk, border, clamp, s);
}

} // namespace zg
