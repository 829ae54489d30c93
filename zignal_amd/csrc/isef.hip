// isef.hip — Shen-Castan's ISEF smoothing (reference src/image/edges.zig:283-349): along every row, then along every column, a forward
// recursion temp[i] = b * data[i] + a * temp[i - 1] (temp[0] = b * data[0]) and a backward one data[i] = b * temp[i] + a * data[i + 1]
// (data[n - 1] = temp[n - 1]), a = 1 - b, separate f32 multiplies and additions in exactly that order.
//
// Two kernels, one result:
//   k_isef_spec   (further down) the recursions cut into overlapping SEGMENTS that run side by side — the recursion contracts by a per step, so
//                 a segment started a few steps early from zero arrives with the sequential value — each leaving behind what the next kernel
//                 needs to PROVE that it did: 83 us for a 4096^2 plane;
//   k_isef        the sequential formulation, one dependent chain per row (column) from end to end, role-split so that the chain's wave
//                 issues nothing but the chain: 251 us. It is the repair launch behind k_isef_spec (it checks every segment's start bit for bit
//                 and redoes the 64-chain groups that fail, normally none) and the route for smoothing factors whose warm-up outgrows a window.
//
// k_isef. A pass is one dependent chain per row (or column): 4 096 chains of 4 096 steps for a 4096^2 plane, two dependent operations a step,
// and nothing may be re-associated. The time of a pass is therefore steps x (instructions the chain's wave issues per step) x ~4.2
// cycles, whatever else the chip does — so the chain's wave must issue nothing but the chain. One workgroup owns 64 chains:
//   wave 0          the CHAIN: lane = chain. Per four steps one ds_read_b128 (the products b * x, already formed), four times
//                   { ar = a * run; run = bx + ar }, one ds_write_b128 of the results: 2.75 instructions a step;
//   waves 1 .. 4    LOADERS, each a quarter of every block: read the block's 64 chains x 64 steps from memory with whatever lane mapping is
//                   coalesced for the direction (a chain per lane down the columns; 16-byte chunks along the rows), three blocks ahead in
//                   registers, multiply by b and lay the products down in LDS chain-major — the transposition a row pass needs happens here;
//   waves 5 .. 8    STORERS, each a quarter of every block: the chain's results out of LDS, written with the same coalesced mapping.
// One barrier per 64-step block hands slot k & 1 of the in-ring to the chain and slot (k - 1) & 1 of the out-ring to a storer.
// The forward and the backward pass of a direction run back to back in one launch (the backward pass reads what this workgroup's
// storers wrote: a device-scope fence and a barrier between the two).
// Round 3 ran the row passes as two full-plane transposes around a column kernel whose chain wave did its own LDS reads (64 per
// block), the b * x products and a global store per row: 137 us per direction + 23 us per transpose for a 4096^2 plane.
#include "zg_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

#pragma clang fp contract(off)

namespace zg {

// The hand-over barrier: this wave's LDS operations are complete (lgkmcnt(0)), its loads and stores stay in flight — __syncthreads() would also
// wait for every outstanding global load and store (vmcnt(0)), i.e. put a memory round trip into every 64-step block.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0), vmcnt and expcnt at their maxima
    __builtin_amdgcn_s_barrier();
}

constexpr int ISEF_SB = 64;    // steps per block
constexpr int ISEF_PITCH = 68; // floats per chain in a ring slot: 16-byte aligned rows, lanes 17 banks apart
constexpr int ISEF_NL = 4, ISEF_NS = 4, ISEF_D = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4i __attribute__((ext_vector_type(4)));

// What the segmented kernel (k_isef_spec below) leaves for the repair launch: four values per (segment, chain), as bits.
struct SpecCheck {
    uint32_t *v = nullptr;
    int n_seg = 0, n_chains = 0;
    __device__ uint32_t &at(int what, int seg, int chain) const { return v[((size_t)what * n_seg + seg) * n_chains + chain]; }
};
enum { SC_FWD_SPEC = 0, SC_FWD_TRUE = 1, SC_BWD_SPEC = 2, SC_BWD_TRUE = 3 };

// ROWS: a chain per image row, steps along the columns (cols % 4 == 0, 16-byte aligned planes); otherwise a chain per column.
// Pass 0: src -> tmp (forward), pass 1: tmp -> dst (backward). dst may be src.
// SRC8 (rows only): src is the plane as bytes (integers 0 .. 255, what the detector's grey is); the forward pass converts as it loads.
__device__ __forceinline__ f32x4 f32_of_bytes(uint32_t v) {
    return f32x4{(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24)};
}
template <bool ROWS, bool SRC8 = false>
__global__ __launch_bounds__(64 * (1 + ISEF_NL + ISEF_NS)) void k_isef(const void *src, float *tmp, float *dst, int rows, int cols, float b, SpecCheck chk) {
    static_assert(ROWS || !SRC8, "bytes come in along the rows");
    constexpr int SB = ISEF_SB, P = ISEF_PITCH, NL = ISEF_NL, D = ISEF_D;
    if (chk.v != nullptr) { // the REPAIR launch behind k_isef_spec: this workgroup's 64 chains are redone only if a segment of theirs started wrong
        const int nc = ROWS ? rows : cols, c0 = (int)blockIdx.x * 64;
        int bad = 0;
        const int total = chk.n_seg * 64, step = (int)blockDim.x;
        for (int i0 = (int)threadIdx.x; i0 < total; i0 += 4 * step) { // four entries a round: their sixteen loads go out together (the launch is one round trip, not n_seg / 9)
            uint32_t fs[4], ft[4], bs[4], bt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * step, total - 1), j = i >> 6, c = min(c0 + (i & 63), nc - 1);
                fs[u] = chk.at(SC_FWD_SPEC, j, c);
                ft[u] = chk.at(SC_FWD_TRUE, max(j - 1, 0), c);
                bs[u] = chk.at(SC_BWD_SPEC, j, c);
                bt[u] = chk.at(SC_BWD_TRUE, min(j + 1, chk.n_seg - 1), c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * step, j = i >> 6;
                if (i >= total || c0 + (i & 63) >= nc) continue;
                if (j >= 1) bad |= fs[u] != ft[u];
                if (j + 1 < chk.n_seg) bad |= bs[u] != bt[u];
            }
        }
        if (!__syncthreads_or(bad)) return;
    }
    constexpr int PER = SB / NL; // COLS: steps a loader / storer wave owns per block; ROWS: chains it owns
    const size_t ld = (size_t)cols;
    __shared__ __attribute__((aligned(16))) float in_ring[2][64 * P];
    __shared__ __attribute__((aligned(16))) float out_ring[2][64 * P];
    __shared__ float first_raw[64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int n_chains = ROWS ? rows : cols, n_steps = ROWS ? cols : rows;
    const int ch0 = (int)blockIdx.x * 64;
    const int nb = (n_steps + SB - 1) / SB;
    const float a = 1.0f - b;

    for (int pass = 0; pass < 2; ++pass) {
        const float *in = pass == 0 ? (const float *)src : tmp;
        float *out = pass == 0 ? tmp : dst;
        auto block_of = [&](int k) { return pass == 0 ? k : nb - 1 - k; }; // the k-th block this pass processes

        if (wave == 0) { // ---- the chain -------------------------------------------------------------------------------------------
            float run = 0.0f;
            for (int k = 0; k <= nb; ++k) {
                lds_barrier();
                if (k == nb) break;
                const int blk = block_of(k), s0 = blk * SB;
                const float *ib = &in_ring[k & 1][lane * P];
                float *ob = &out_ring[k & 1][lane * P];
                const bool whole = s0 + SB <= n_steps; // every step of the block exists
#ifdef ZG_ISEF_NOCHAIN // tools/exp/isef_bench.hip: the hand-overs without the chain's arithmetic
                if (run == 0.0f) continue;
#endif
                if (pass == 0 && whole && s0 > 0) {
#pragma unroll
                    for (int g = 0; g < SB / 4; ++g) { // unrolled: the sixteen reads go out together and the arithmetic follows them
                        const f32x4 v = *(const f32x4 *)(ib + 4 * g);
                        f32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float ar = a * run; run = v[e] + ar; r[e] = run; }
                        *(f32x4 *)(ob + 4 * g) = r;
                    }
                } else if (pass == 1 && whole && s0 + SB < n_steps) {
#pragma unroll
                    for (int g = SB / 4 - 1; g >= 0; --g) {
                        const f32x4 v = *(const f32x4 *)(ib + 4 * g);
                        f32x4 r;
#pragma unroll
                        for (int e = 3; e >= 0; --e) { const float ar = a * run; run = v[e] + ar; r[e] = run; }
                        *(f32x4 *)(ob + 4 * g) = r;
                    }
                } else if (pass == 0) { // the first block (the chain starts at step 0) and / or a partial last one
                    for (int g = 0; g < SB / 4; ++g) {
                        const f32x4 v = *(const f32x4 *)(ib + 4 * g);
                        f32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int st = s0 + 4 * g + e; // wave-uniform
                            if (st == 0) run = v[e];
                            else if (st < n_steps) { const float ar = a * run; run = v[e] + ar; }
                            r[e] = run;
                        }
                        *(f32x4 *)(ob + 4 * g) = r;
                    }
                } else { // the block the backward chain starts in (step n - 1 is not multiplied by b) and / or a partial one
                    const float raw = first_raw[lane];
                    for (int g = SB / 4 - 1; g >= 0; --g) {
                        const f32x4 v = *(const f32x4 *)(ib + 4 * g);
                        f32x4 r;
#pragma unroll
                        for (int e = 3; e >= 0; --e) {
                            const int st = s0 + 4 * g + e;
                            if (st == n_steps - 1) run = raw;
                            else if (st < n_steps - 1) { const float ar = a * run; run = v[e] + ar; }
                            r[e] = run;
                        }
                        *(f32x4 *)(ob + 4 * g) = r;
                    }
                }
            }
        } else if (wave <= NL) { // ---- loaders: every one of them a part of every block, D blocks ahead in registers ---------------------
            const int sub = wave - 1;
            struct Regs { f32x4 v[4]; };
            // COLS: lane = chain, this wave's steps sub * PER .. + PER of the block (dword loads, each 256 contiguous bytes across the wave).
            // ROWS: this wave's chains sub * PER .. + PER; lane l of load j holds chunk q = 64 j + l: chain sub * PER + (q >> 4), steps 4 (q & 15) .. + 4.
            auto fetch = [&](int k, Regs &g) { // clamped, unpredicated: what lies past the plane is re-read from inside it and never used
                const int blk = min(max(block_of(min(k, nb - 1)), 0), nb - 1), s0 = blk * SB;
                if constexpr (ROWS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = 64 * j + lane, chain = min(ch0 + sub * PER + (q >> 4), n_chains - 1), st = min(s0 + 4 * (q & 15), n_steps - 4);
                        if (SRC8 && pass == 0) g.v[j] = f32_of_bytes(*(const uint32_t *)((const uint8_t *)src + (size_t)chain * ld + st));
                        else g.v[j] = *(const f32x4 *)(in + (size_t)chain * ld + st);
                    }
                } else {
                    const int chain = min(ch0 + lane, n_chains - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) g.v[j][e] = in[(size_t)min(s0 + sub * PER + 4 * j + e, n_steps - 1) * ld + chain];
                }
            };
            auto publish = [&](int k, const Regs &g) {
                float *ring = in_ring[k & 1];
                const int blk = block_of(k), s0 = blk * SB;
                const bool holds_last = pass == 1 && s0 <= n_steps - 1 && n_steps - 1 < s0 + SB; // wave-uniform: the backward chain's first value goes over raw
                if constexpr (ROWS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = 64 * j + lane, cl = sub * PER + (q >> 4), sl = 4 * (q & 15);
                        *(f32x4 *)(ring + cl * P + sl) = g.v[j] * b;
                        if (holds_last && s0 + sl + 3 == n_steps - 1) first_raw[cl] = g.v[j][3]; // cols % 4 == 0: step n - 1 ends its chunk
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *(f32x4 *)(ring + lane * P + sub * PER + 4 * j) = g.v[j] * b;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (holds_last && s0 + sub * PER + 4 * j + e == n_steps - 1) first_raw[lane] = g.v[j][e];
                    }
                }
            };
            Regs g[D];
#pragma unroll
            for (int d = 0; d < D - 1; ++d) fetch(d, g[d]);
            for (int k0 = 0; k0 <= nb; k0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int k = k0 + d;
                    if (k <= nb) { // workgroup-uniform
#ifdef ZG_ISEF_NOLOAD
                        if (b == 1.2345f)
#endif
                        if (k < nb) {
                            publish(k, g[d]);
                            fetch(k + D - 1, g[(d + D - 1) % D]);
                        }
                        lds_barrier();
                    }
                }
            }
        } else { // ---- storers: every one of them a part of every block -------------------------------------------------------------
            const int sub = wave - 1 - NL;
            for (int k = 0; k <= nb; ++k) {
                lds_barrier();
                if (k == 0) continue;
#ifdef ZG_ISEF_NOSTORE
                if (out_ring[0][lane] != 1.2345f) continue;
#endif
                const float *ring = out_ring[(k - 1) & 1];
                const int blk = block_of(k - 1), s0 = blk * SB;
                if constexpr (ROWS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = 64 * j + lane, cl = sub * PER + (q >> 4), sl = 4 * (q & 15);
                        const f32x4 r = *(const f32x4 *)(ring + cl * P + sl);
                        if (ch0 + cl < n_chains && s0 + sl < n_steps) __builtin_nontemporal_store(r, (f32x4 *)(out + (size_t)(ch0 + cl) * ld + s0 + sl));
                    }
                } else {
                    const bool live = ch0 + lane < n_chains;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 r = *(const f32x4 *)(ring + lane * P + sub * PER + 4 * j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int st = s0 + sub * PER + 4 * j + e;
                            if (live && st < n_steps) out[(size_t)st * ld + ch0 + lane] = r[e];
                        }
                    }
                }
            }
        }
        __threadfence(); // pass 0's results are visible to this workgroup's loaders before they read them back
        __syncthreads();
    }
}


// ---- the recursions cut into segments ----------------------------------------------------------------------------------------------------
// a = 1 - b < 1 makes the recursion a contraction: two runs over the same inputs from different starting values close in on each other by
// the factor a per step, and once their f32 values are equal they stay equal for good (same inputs, same operations). So a segment of a
// chain need not wait for the one before it: it starts W steps early from zero, and by the time it reaches its first own step its value IS
// the sequential one — unless the two runs still straddle a rounding boundary, which after W steps has probability ~ a^(W - log_a 2^-24) per
// segment (measured: tools/exp/isef_merge.py). That is not left to chance: every segment records the value it arrived with at its first own
// step and the value it hands to the next segment, for both recursions; the REPAIR launch (k_isef with a SpecCheck) compares them bit for
// bit and redoes a workgroup's 64 chains sequentially if any segment of theirs started from anything but its predecessor's exact value. By
// induction from the chain's true first (last) step every segment that passes holds exactly the sequential values; W is chosen so that a
// repair is expected once in ~10^4 planes of 4096^2. What the segments buy: a pass is no longer 4 096 dependent steps (the 135 us a direction above)
// but (S + 2 W) of them per wave with thousands of waves in flight, i.e. the plane's bytes through HBM.
//
// One wave owns a window of up to SPEC_WIN steps of 64 chains in LDS, in place: the products b * x come in (coalesced along the rows, a
// chain per lane afterwards), the forward recursion overwrites them with temp, the backward one with the results, and the S own steps go out.
constexpr int SPEC_WIN = 128;                 // steps a window holds
constexpr uint32_t SPEC_WAVES = 4 * 256;      // resident one-wave workgroups: four windows of 33 KB fit a CU's LDS
constexpr int SPEC_PITCH = SPEC_WIN + 4;      // ROWS layout [chain][step]: 16-byte aligned rows, lanes 4 banks apart (b128 accesses conflict-free)

template <bool ROWS, bool SRC8 = false>
__global__ __launch_bounds__(64) void k_isef_spec(const void *src, float *out, int rows, int cols, float b, int W, int S, SpecCheck chk) {
    static_assert(ROWS || !SRC8, "bytes come in along the rows");
    const float *in = (const float *)src;
    __shared__ __attribute__((aligned(16))) float buf[ROWS ? 64 * SPEC_PITCH : SPEC_WIN * 64];
    const int lane = (int)threadIdx.x;
    const int n_chains = ROWS ? rows : cols, n = ROWS ? cols : rows;
    const int n_groups = (n_chains + 63) / 64, n_win = chk.n_seg * n_groups;
    const size_t ld = (size_t)cols;
    const float a = 1.0f - b;
    struct Win { int seg, ch0, w0, w1, len, own0, own1; };
    auto window = [&](int t) { // rows: the segments of a chain group follow each other (their windows overlap); columns: the groups of a segment do
        Win w;
        w.seg = ROWS ? t % chk.n_seg : t / n_groups;
        w.ch0 = (ROWS ? t / chk.n_seg : t % n_groups) * 64;
        const int s0 = w.seg * S, e0 = min(s0 + S, n); // the own steps
        w.w0 = max(s0 - W, 0), w.w1 = min(e0 + W, n);
        w.len = w.w1 - w.w0, w.own0 = s0 - w.w0, w.own1 = e0 - w.w0;
        return w;
    };
    // A window's 32 KB come in as 32 loads of 16 bytes per lane, all in flight at once, and wait in registers while the window before it is
    // worked on: the wave's memory latency is hidden behind its own recursions.
    // ROWS: lane l of load i: chain 2 i + (l >> 5), steps 4 (l & 31) .. + 4 of the window (512 contiguous bytes per chain);
    // columns: step 4 i + (l >> 4), chains 4 (l & 15) .. + 4 (256 contiguous bytes per step).
    // SRC8: the same lanes take the same four steps as four bytes.
    f32x4 pre[SRC8 ? 1 : 32];
    uint32_t pre8[SRC8 ? 32 : 1];
    auto fetch = [&](const Win &w) {
        if constexpr (SRC8) {
            const uint8_t *p = (const uint8_t *)src + w.w0 + min(4 * (lane & 31), w.len - 4);
#pragma unroll
            for (int i = 0; i < 32; ++i) pre8[i] = *(const uint32_t *)(p + (size_t)min(w.ch0 + 2 * i + (lane >> 5), n_chains - 1) * ld);
        } else if constexpr (ROWS) {
            const int st = 4 * (lane & 31);
            const float *p = in + w.w0 + min(st, w.len - 4); // clamped: what lies past the window is re-read from inside it and never used
#pragma unroll
            for (int i = 0; i < 32; ++i) pre[i] = *(const f32x4 *)(p + (size_t)min(w.ch0 + 2 * i + (lane >> 5), n_chains - 1) * ld);
        } else {
            const float *p = in + min(w.ch0 + 4 * (lane & 15), n_chains - 4); // cols % 4 == 0
#pragma unroll
            for (int i = 0; i < 32; ++i) pre[i] = *(const f32x4 *)(p + (size_t)(w.w0 + min(4 * i + (lane >> 4), w.len - 1)) * ld);
        }
    };
    auto publish = [&]() { // b * x into the window, chain-major (ROWS) or step-major (columns)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if constexpr (SRC8) *(f32x4 *)(buf + (2 * i + (lane >> 5)) * SPEC_PITCH + 4 * (lane & 31)) = f32_of_bytes(pre8[i]) * b;
            else if constexpr (ROWS) *(f32x4 *)(buf + (2 * i + (lane >> 5)) * SPEC_PITCH + 4 * (lane & 31)) = pre[i] * b;
            else *(f32x4 *)(buf + (4 * i + (lane >> 4)) * 64 + 4 * (lane & 15)) = pre[i] * b;
        }
    };

    int t = (int)blockIdx.x;
    if (t >= n_win) return;
    Win nxt = window(t);
    fetch(nxt);
    for (; t < n_win; t += (int)gridDim.x) {
    const Win w = nxt;
    const int seg = w.seg, ch0 = w.ch0, w0 = w.w0, w1 = w.w1, len = w.len, own0 = w.own0, own1 = w.own1;
    publish();
    if (t + (int)gridDim.x < n_win) { nxt = window(t + (int)gridDim.x); fetch(nxt); }
    __builtin_amdgcn_s_waitcnt(0xc07f); // this wave's LDS writes have landed (the LDS is in order within a wave; one wave per workgroup)
    __builtin_amdgcn_wave_barrier();

    // ---- the two recursions: lane = chain. Sixteen steps at a time: their LDS reads go out together, the dependent arithmetic follows -------
    // What the repair launch compares is read back from the window: temp[s0 - 1] is never overwritten (the backward recursion stops at s0),
    // temp[e0 - 1] is read between the two recursions.
    uint32_t fwd_spec = 0, fwd_true = 0, bwd_spec = 0, bwd_true = 0;
    if constexpr (ROWS) { // len, own0, own1 are multiples of 4 (cols, S, W are)
        float *row = buf + lane * SPEC_PITCH;
        const int ng = len / 4;
        float run = 0.0f;
        int g = 0;
        if (w0 == 0) { // the chain's first step: temp[0] = b * data[0]
            const f32x4 v = *(const f32x4 *)row;
            f32x4 r;
            run = v[0];
            r[0] = run;
#pragma unroll
            for (int e = 1; e < 4; ++e) { const float ar = a * run; run = v[e] + ar; r[e] = run; }
            *(f32x4 *)row = r;
            g = 1;
        }
        for (; g + 4 <= ng; g += 4) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const f32x4 *)(row + 4 * (g + k));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float ar = a * run; run = v[k][e] + ar; r[e] = run; }
                *(f32x4 *)(row + 4 * (g + k)) = r;
            }
        }
        for (; g < ng; ++g) {
            const f32x4 v = *(const f32x4 *)(row + 4 * g);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float ar = a * run; run = v[e] + ar; r[e] = run; }
            *(f32x4 *)(row + 4 * g) = r;
        }
        fwd_true = __float_as_uint(row[own1 - 1]);
        run = 0.0f;
        g = ng - 1;
        const int g_lo = own0 / 4;
        if (w1 == n) { // the chain's last step: data[n - 1] = temp[n - 1]
            const f32x4 v = *(const f32x4 *)(row + 4 * g);
            f32x4 r;
            run = v[3];
            r[3] = run;
#pragma unroll
            for (int e = 2; e >= 0; --e) { const float bt = b * v[e], ar = a * run; run = bt + ar; r[e] = run; }
            *(f32x4 *)(row + 4 * g) = r;
            --g;
        }
        for (; g - 3 >= g_lo; g -= 4) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const f32x4 *)(row + 4 * (g - k));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 r;
#pragma unroll
                for (int e = 3; e >= 0; --e) { const float bt = b * v[k][e], ar = a * run; run = bt + ar; r[e] = run; }
                *(f32x4 *)(row + 4 * (g - k)) = r;
            }
        }
        for (; g >= g_lo; --g) {
            const f32x4 v = *(const f32x4 *)(row + 4 * g);
            f32x4 r;
#pragma unroll
            for (int e = 3; e >= 0; --e) { const float bt = b * v[e], ar = a * run; run = bt + ar; r[e] = run; }
            *(f32x4 *)(row + 4 * g) = r;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        fwd_spec = own0 > 0 ? __float_as_uint(row[own0 - 1]) : 0u;
        bwd_spec = own1 < len ? __float_as_uint(row[own1]) : 0u;
        bwd_true = __float_as_uint(row[own0]);
    } else {
        float *col = buf + lane;
        float run = 0.0f;
        int i = 0;
        if (w0 == 0) { run = col[0]; i = 1; } // the chain's first step: temp[0] = b * data[0]
        for (; i + 16 <= len; i += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = col[(i + k) * 64];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float ar = a * run; run = v[k] + ar; col[(i + k) * 64] = run; }
        }
        for (; i < len; ++i) { const float ar = a * run; run = col[i * 64] + ar; col[i * 64] = run; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        fwd_true = __float_as_uint(col[(own1 - 1) * 64]);
        run = 0.0f;
        i = len - 1;
        if (w1 == n) { run = col[i * 64]; --i; } // the chain's last step: data[n - 1] = temp[n - 1]
        for (; i - 15 >= own0; i -= 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = col[(i - k) * 64];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float bt = b * v[k], ar = a * run; run = bt + ar; col[(i - k) * 64] = run; }
        }
        for (; i >= own0; --i) { const float bt = b * col[i * 64], ar = a * run; run = bt + ar; col[i * 64] = run; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        fwd_spec = own0 > 0 ? __float_as_uint(col[(own0 - 1) * 64]) : 0u;
        bwd_spec = own1 < len ? __float_as_uint(col[own1 * 64]) : 0u;
        bwd_true = __float_as_uint(col[own0 * 64]);
    }
    if (ch0 + lane < n_chains) {
        chk.at(SC_FWD_SPEC, seg, ch0 + lane) = fwd_spec;
        chk.at(SC_FWD_TRUE, seg, ch0 + lane) = fwd_true;
        chk.at(SC_BWD_SPEC, seg, ch0 + lane) = bwd_spec;
        chk.at(SC_BWD_TRUE, seg, ch0 + lane) = bwd_true;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- out: the own steps ---------------------------------------------------------------------------------------------------------------
    if constexpr (ROWS) {
        const int st = 4 * (lane & 31);
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int cl = 2 * i + (lane >> 5);
            if (st >= own0 && st < own1 && ch0 + cl < n_chains)
                __builtin_nontemporal_store(*(const f32x4 *)(buf + cl * SPEC_PITCH + st), (f32x4 *)(out + (size_t)(ch0 + cl) * ld + w0 + st));
        }
    } else {
        const int c4 = 4 * (lane & 15);
        const bool live = ch0 + c4 < n_chains;
        for (int i = own0 / 4; i < (own1 + 3) / 4; ++i) {
            const int st = 4 * i + (lane >> 4);
            if (st >= own0 && st < own1 && live)
                __builtin_nontemporal_store(*(const f32x4 *)(buf + st * 64 + c4), (f32x4 *)(out + (size_t)(w0 + st) * ld + ch0 + c4));
        }
    }
    __builtin_amdgcn_wave_barrier(); // the next window's b * x follow these reads (in order within the wave)
    }
}

// Warm-up steps for the contraction factor a: a^W <= 2^-24 (the runs within an ulp) x 10^-10 (then still apart), a multiple of 4:
// ~10^6 checks a 4096^2 plane -> one repair (the time of the sequential kernel for that direction) in ~10^4 planes.
static int spec_warmup(float a) {
    if (!(a > 0.0f)) return 4;
    const double w = std::ceil((std::log(0x1p-24) + std::log(1e-10)) / std::log((double)a));
    return w > 1e6 ? 1 << 20 : ((int)w + 3) / 4 * 4;
}

size_t isef_check_bytes(uint32_t rows, uint32_t cols) { // the SpecCheck plane isef_2d wants: four words per (segment, chain), S >= 32
    const size_t by_rows = (size_t)(ceil_div(cols, 32u) + 1) * rows, by_cols = (size_t)(ceil_div(rows, 32u) + 1) * cols;
    return 4 * sizeof(uint32_t) * (by_rows > by_cols ? by_rows : by_cols);
}

// The smoothing of a rows x cols f32 plane: gray -> sm, with `tmp` (same size) between the passes. Returns -1 when the row kernel's
// preconditions do not hold (cols % 4, alignment): the caller then takes the transposing route.
bool isef_2d_applies(uint32_t rows, uint32_t cols) {
    static const bool off = getenv("ZIGNAL_HIP_ISEF_TRANSPOSE") != nullptr; // tuning hook: round 3's route
    return !off && cols % 4 == 0 && cols >= 4 && (uint64_t)rows * cols * 4 < 0x40000000u; // 32-bit buffer offsets (the masked lanes' 0x80000000 stays out of range of whatever is added)
}

template <bool SRC8>
static void isef_2d_launch(const void *gray, float *sm, float *tmp, uint32_t *check, uint32_t rows, uint32_t cols, float smooth, hipStream_t s) {
    static const bool serial = getenv("ZIGNAL_HIP_ISEF_SERIAL") != nullptr; // tuning hook: one chain per row / column from end to end
    static const int w_env = getenv("ZIGNAL_HIP_ISEF_W") ? atoi(getenv("ZIGNAL_HIP_ISEF_W")) : 0; // tests: a short warm-up makes the repair launch work
    const dim3 block(64 * (1 + ISEF_NL + ISEF_NS));
    const int W = w_env >= 4 ? w_env / 4 * 4 : spec_warmup(1.0f - smooth);
    if (serial || check == nullptr || 2 * W > SPEC_WIN - 32) { // a close to 1: the windows would be mostly warm-up
        hipLaunchKernelGGL((k_isef<true, SRC8>), dim3(ceil_div(rows, 64)), block, 0, s, gray, tmp, sm, (int)rows, (int)cols, smooth, SpecCheck{});
        hipLaunchKernelGGL((k_isef<false>), dim3(ceil_div(cols, 64)), block, 0, s, (const void *)sm, tmp, sm, (int)rows, (int)cols, smooth, SpecCheck{});
        return;
    }
    const int S = SPEC_WIN - 2 * W;
    // rows: gray -> tmp (repair: gray -> sm -> tmp); columns: tmp -> sm (repair: tmp -> sm -> sm, the backward recursion in place)
    const SpecCheck cr{check, (int)ceil_div(cols, (uint32_t)S), (int)rows}, cc{check, (int)ceil_div(rows, (uint32_t)S), (int)cols};
    hipLaunchKernelGGL((k_isef_spec<true, SRC8>), dim3(std::min<uint32_t>(cr.n_seg * ceil_div(rows, 64), SPEC_WAVES)), dim3(64), 0, s, gray, tmp, (int)rows, (int)cols, smooth, W, S, cr);
    hipLaunchKernelGGL((k_isef<true, SRC8>), dim3(ceil_div(rows, 64)), block, 0, s, gray, sm, tmp, (int)rows, (int)cols, smooth, cr);
    hipLaunchKernelGGL((k_isef_spec<false>), dim3(std::min<uint32_t>(cc.n_seg * ceil_div(cols, 64), SPEC_WAVES)), dim3(64), 0, s, (const void *)tmp, sm, (int)rows, (int)cols, smooth, W, S, cc);
    hipLaunchKernelGGL((k_isef<false>), dim3(ceil_div(cols, 64)), block, 0, s, (const void *)tmp, sm, sm, (int)rows, (int)cols, smooth, cc);
}

// gray: the plane as f32, or — gray_is_bytes — as bytes (the detector's grey: integers 0 .. 255), converted as the row pass loads it.
int isef_2d(const void *gray, bool gray_is_bytes, float *sm, float *tmp, uint32_t *check, uint32_t rows, uint32_t cols, float smooth, hipStream_t s) {
    if (!isef_2d_applies(rows, cols) || ((uintptr_t)gray & 15) || ((uintptr_t)sm & 15) || ((uintptr_t)tmp & 15)) return -1;
    if (gray_is_bytes) isef_2d_launch<true>(gray, sm, tmp, check, rows, cols, smooth, s);
    else isef_2d_launch<false>(gray, sm, tmp, check, rows, cols, smooth, s);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

} // namespace zg
