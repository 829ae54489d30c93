// isef.hip — Shen-Castan's ISEF smoothing (reference src/image/edges.zig:283-349): along every row, then along every column, a forward
// recursion temp[i] = b * data[i] + a * temp[i - 1] (temp[0] = b * data[0]) and a backward one data[i] = b * temp[i] + a * data[i + 1]
// (data[n - 1] = temp[n - 1]), a = 1 - b, separate f32 multiplies and additions in exactly that order.
//
// A pass is one dependent chain per row (or column): 4 096 chains of 4 096 steps for a 4096^2 plane, two dependent operations a step,
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

// ROWS: a chain per image row, steps along the columns (cols % 4 == 0, 16-byte aligned planes); otherwise a chain per column.
// Pass 0: src -> tmp (forward), pass 1: tmp -> dst (backward). dst may be src.
template <bool ROWS>
__global__ __launch_bounds__(64 * (1 + ISEF_NL + ISEF_NS)) void k_isef(const float *src, float *tmp, float *dst, int rows, int cols, float b) {
    constexpr int SB = ISEF_SB, P = ISEF_PITCH, NL = ISEF_NL, NS = ISEF_NS, D = ISEF_D;
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
        const float *in = pass == 0 ? src : tmp;
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
                        g.v[j] = *(const f32x4 *)(in + (size_t)chain * ld + st);
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

// The smoothing of a rows x cols f32 plane: gray -> sm, with `tmp` (same size) between the passes. Returns -1 when the row kernel's
// preconditions do not hold (cols % 4, alignment): the caller then takes the transposing route.
int isef_2d(const float *gray, float *sm, float *tmp, uint32_t rows, uint32_t cols, float smooth, hipStream_t s) {
    static const bool off = getenv("ZIGNAL_HIP_ISEF_TRANSPOSE") != nullptr; // tuning hook: round 3's route
    if (off || cols % 4 || cols < 4 || ((uintptr_t)gray & 15) || ((uintptr_t)sm & 15) || ((uintptr_t)tmp & 15)) return -1;
    if ((uint64_t)rows * cols * 4 >= 0x40000000u) return -1; // 32-bit buffer offsets (the masked lanes' 0x80000000 stays out of range of whatever is added)
    const dim3 block(64 * (1 + ISEF_NL + ISEF_NS));
    hipLaunchKernelGGL(k_isef<true>, dim3(ceil_div(rows, 64)), block, 0, s, gray, tmp, sm, (int)rows, (int)cols, smooth);
    hipLaunchKernelGGL(k_isef<false>, dim3(ceil_div(cols, 64)), block, 0, s, (const float *)sm, tmp, sm, (int)rows, (int)cols, smooth);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

} // namespace zg
