// conv_separable.hip — Image(T).convolveSeparable / gaussianBlur on gfx950.
//
// Replaces reference src/image/convolution.zig:313-438 (type switch) and :441-647
// (convolveSeparablePlane) plus src/image.zig:954-994 (gaussianBlur).
//
// Arithmetic contract (what makes the output bit-identical to the reference CPU path):
//   horizontal  temp[r,c] = sum_i  src[r, c+i-hx] * kx[i]   ascending i, acc starts at 0
//   vertical    dst[r,c]  = sum_i temp[r+i-hy, c] * ky[i]   ascending i, acc starts at 0
//   f32: separate multiply and add (no FMA);  u8: taps round(k*256) as i32, i64-exact accumulate,
//   temp clamped to i32, output divClampU8(65536) = round-half-away then clamp.
//   Out-of-range taps go through border.resolveIndex on BOTH passes; `temp` at a resolved row is the
//   horizontal result of that row, so the two passes fuse without changing a single bit.
//   Interior pixels skip taps with |k| < 1e-10 (f32) (convolution.zig:459-467,541,594); border pixels
//   do not. For integers skipping a zero tap is a no-op, so only the f32 path carries the skip mask.
//
// The reference de-interleaves struct pixels into planes and runs the plane kernel per channel
// (convolution.zig:358-430). Channels are independent, so the kernels here stay interleaved (one
// coalesced read and one coalesced write of the image) and produce the same bytes. The uniform-
// channel shortcut (:367-412) yields the same values as the full computation for every border mode
// it is enabled for, so it is not reproduced.
//
// Kernels:
//   k_sep_fused<PIX,NK,MODE,SKIP>  one launch: (TW+2h)x(TH+2h) source tile -> LDS (border resolved at
//       load time), row pass from LDS into a register sliding window of NK temps, column pass from
//       the window, one coalesced store. HBM traffic = read once + write once.
//   k_sep_h / k_sep_v              general two-pass fallback for long or asymmetric kernels
//       (temp plane in HBM, as the reference does).
#include "zg_common.h"
#include "zg_hostmath.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#pragma clang fp contract(off)

namespace zg {

constexpr int MAX_TAPS = 255;          // as a kernel argument
constexpr int MAX_TAPS_MEM = 1 << 22;  // from device memory (radius up to two million: far past any image)

enum : int { MODE_F32 = 0, MODE_I24 = 1, MODE_I64 = 2 };

template <int N> struct TapsArg {
    union { float f[N]; int32_t i[N]; };
};

template <int MODE> struct Arith;
template <> struct Arith<MODE_F32> {
    using Temp = float; using Acc = float;
    __device__ static Acc mac(Acc a, Temp v, float k) { const float p = v * k; return a + p; }
    __device__ static Temp to_temp(Acc a) { return a; }
};
template <> struct Arith<MODE_I24> { // host proved every product and sum fits: exact in i32
    using Temp = int32_t; using Acc = int32_t;
    __device__ static Acc mac(Acc a, Temp v, int32_t k) { return a + __mul24(v, k); }
    __device__ static Temp to_temp(Acc a) { return a; }
};
template <> struct Arith<MODE_I64> { // reference widths: i64 accumulate, temp clamped to i32
    using Temp = int32_t; using Acc = int64_t;
    __device__ static Acc mac(Acc a, Temp v, int32_t k) { return a + (int64_t)v * (int64_t)k; }
    __device__ static Temp to_temp(Acc a) {
        return (int32_t)(a < INT32_MIN ? (int64_t)INT32_MIN : (a > INT32_MAX ? (int64_t)INT32_MAX : a));
    }
};

// divClampU8(65536, acc) (convolution.zig:18-22): symmetric rounding divide, clamp to u8.
template <typename Acc> __device__ inline uint8_t div_clamp_u8_sq(Acc acc) {
    if (acc < 0) return 0; // (acc - 32768) / 65536 truncates to <= 0
    const Acc q = (acc + 32768) >> 16;
    return (uint8_t)(q > 255 ? 255 : q);
}

template <int MODE, typename TapsT> __device__ inline auto tap(const TapsT &t, int i) {
    if constexpr (MODE == MODE_F32) return t.f[i]; else return t.i[i];
}

constexpr int TW = 64;  // tile width = one wavefront of columns; tile height = 4 waves x RPT rows

// Kernel shape parameters (measured on MI355X, profiles/r01_sep_variant_sweep.txt):
//   RPT      output rows per thread (tile height = 4 * RPT); 4 for 16-byte pixels, 8 otherwise
//   PERSIST  persistent workgroups that prefetch tile t+1 into registers while convolving tile t (measured slower; kept
//            selectable for future geometries)
//   NT       non-temporal stores for the output (always on; non-temporal LOADS measured slower)

// Tile staging shared by the persistent kernel: load one (TH+2H) x (TW+2H) source tile into registers
// (every load issued back to back), and later spill those registers to LDS.
template <int PIX, int NK, int RPT> struct TileStage {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    static constexpr int H = NK / 2;
    static constexpr int TH = 4 * RPT;
    static constexpr int LW = TW + 2 * H;
    static constexpr int LH = TH + 2 * H;
    static constexpr int RW = (LH + 3) / 4;          // tile rows per wave, interleaved by wave
    static constexpr int NEXTRA = LH * 2 * H;        // halo columns TW .. TW+2H-1 of every row
    static constexpr int EX = (NEXTRA + 255) / 256;
    Vec main_v[RW];
    Vec extra_v[EX > 0 ? EX : 1];

    __device__ void load(const DImg &src, int x0, int y0, int border, int lx, int wave) {
        const bool inside = x0 - H >= 0 && x0 + TW + H <= src.cols && y0 - H >= 0 && y0 + TH + H <= src.rows;
        if (inside) { // all but the frame's rim: no index resolution at all
            const size_t base = (size_t)(y0 - H) * src.stride + (size_t)(x0 - H);
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const int r = wave + 4 * k;
                if (r < LH) main_v[k] = P::load(src.data, base + (size_t)r * src.stride + (size_t)lx);
            }
            if constexpr (H > 0) {
#pragma unroll
                for (int k = 0; k < EX; ++k) {
                    const int e = (int)threadIdx.x + 256 * k;
                    if (e < NEXTRA) {
                        const int r = e / (2 * H), c = TW + e - r * (2 * H);
                        extra_v[k] = P::load(src.data, base + (size_t)r * src.stride + (size_t)c);
                    }
                }
            }
        } else {
            const int gc_main = resolve_index(x0 - H + lx, src.cols, border);
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const int r = wave + 4 * k;
                main_v[k] = P::zero();
                if (r < LH) {
                    const int gr = resolve_index(y0 - H + r, src.rows, border); // wave-uniform
                    if (gr >= 0 && gc_main >= 0) main_v[k] = P::load(src.data, (size_t)gr * src.stride + (size_t)gc_main);
                }
            }
            if constexpr (H > 0) {
#pragma unroll
                for (int k = 0; k < EX; ++k) {
                    const int e = (int)threadIdx.x + 256 * k;
                    extra_v[k] = P::zero();
                    if (e < NEXTRA) {
                        const int r = e / (2 * H), c = TW + e - r * (2 * H);
                        const int gr = resolve_index(y0 - H + r, src.rows, border);
                        const int gc = resolve_index(x0 - H + c, src.cols, border);
                        if (gr >= 0 && gc >= 0) extra_v[k] = P::load(src.data, (size_t)gr * src.stride + (size_t)gc);
                    }
                }
            }
        }
    }

    __device__ void spill(Vec *tile, int lx, int wave) const {
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int r = wave + 4 * k;
            if (r < LH) tile[r * LW + lx] = main_v[k];
        }
        if constexpr (H > 0) {
#pragma unroll
            for (int k = 0; k < EX; ++k) {
                const int e = (int)threadIdx.x + 256 * k;
                if (e < NEXTRA) {
                    const int r = e / (2 * H), c = TW + e - r * (2 * H);
                    tile[r * LW + c] = extra_v[k];
                }
            }
        }
    }
};

// Row-clipped pixel store. The row's buffer descriptor (wave-uniform, SGPRs) carries the row length, so the
// hardware drops lanes that fall outside the image and the store needs no branch. Unconditional stores let
// the compiler count them exactly in s_waitcnt, which is what keeps the next tile's loads from being
// serialised behind this tile's stores (gfx9 has one vmcnt for loads and stores).
template <int PIX, bool NT> struct RowStore {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __device__ static void store(const DImg &dst, int gy, int gx, Vec v) {
        const bool row_ok = gy >= 0 && gy < dst.rows;
        char *row = (char *)dst.data + (row_ok ? (size_t)gy * dst.stride * P::BYTES : (size_t)0);
        const int bytes = row_ok ? dst.cols * P::BYTES : 0;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, bytes, 0x00020000);
        const int off = gx * P::BYTES;
        constexpr int AUX = NT ? 2 : 0; // bit 1 = nt (streaming): the output is never re-read by this kernel
        if constexpr (P::BYTES == 16) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, off, 0, AUX);
        } else if constexpr (P::BYTES == 12) {
            const float e0 = v[0], e1 = v[1], e2 = v[2];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e0), rsrc, off, 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e1), rsrc, off + 4, 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e2), rsrc, off + 8, 0, AUX);
        } else if constexpr (P::BYTES == 4) {
            if constexpr (std::is_same<typename P::Elem, float>::value) {
                const float e0 = v[0];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e0), rsrc, off, 0, AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rsrc, off, 0, AUX);
            }
        } else if constexpr (P::BYTES == 3) {
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v[0], rsrc, off, 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v[1], rsrc, off + 1, 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v[2], rsrc, off + 2, 0, AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v[0], rsrc, off, 0, AUX);
        }
    }
};

// Convolve one staged tile out of LDS: row pass into a register sliding window of NK temps, column pass
// from the window, one row-clipped store per output row.
template <int PIX, int NK, int MODE, bool SKIP, int RPT, bool NT>
__device__ __forceinline__ void convolve_tile(const typename Px<PIX>::Vec *tile, const DImg &src, const DImg &dst,
                                              const TapsArg<NK> &kx, const TapsArg<NK> &ky, uint32_t skipx,
                                              uint32_t skipy, int x0, int y0, int lx, int wave) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    using A = Arith<MODE>;
    using Temp = typename A::Temp;
    using Acc = typename A::Acc;
    constexpr int C = P::C;
    constexpr int H = NK / 2;
    constexpr int LW = TW + 2 * H;

    const int gx = x0 + lx;
    const bool col_interior = (src.cols > 2 * H) && gx >= H && gx < src.cols - H;
    const bool rows_have_interior = src.rows > 2 * H;

    Temp win[NK][C];
#pragma unroll
    for (int j = 0; j < RPT + 2 * H; ++j) {
        const int lr = wave * RPT + j; // tile row whose horizontal result enters the window
        Acc acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            if (SKIP && col_interior && ((skipx >> i) & 1u)) continue;
            const Vec v = tile[lr * LW + lx + i];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = A::mac(acc[ch], (Temp)v[ch], tap<MODE>(kx, i));
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) win[j % NK][ch] = A::to_temp(acc[ch]);

        if (j >= 2 * H) {
            const int gy = y0 + wave * RPT + (j - 2 * H);
            const bool row_interior = rows_have_interior && gy >= H && gy < src.rows - H;
            Acc out[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out[ch] = 0;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                if (SKIP && row_interior && ((skipy >> i) & 1u)) continue;
                // tap i reads temp row (j - 2H + i) of this thread's strip = window slot (j + 1 + i) % NK
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                    out[ch] = A::mac(out[ch], win[(j + 1 + i) % NK][ch], tap<MODE>(ky, i));
            }
            Vec o;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                if constexpr (MODE == MODE_F32) o[ch] = out[ch];
                else o[ch] = div_clamp_u8_sq<Acc>(out[ch]);
            }
            RowStore<PIX, NT>::store(dst, gy, gx, o);
        }
    }
}

// Persistent, software-pipelined fused kernel. Each workgroup owns a contiguous run of tiles; while it
// convolves tile t out of LDS, the global loads of tile t+1 are already in flight into registers, so a CU
// never drains its memory queue between tiles. grid = min(#tiles, resident workgroups).
template <int PIX, int NK, int MODE, bool SKIP, int RPT, bool PERSIST, bool NT>
__global__ __launch_bounds__(256) void k_sep_fused(DImg src, DImg dst, TapsArg<NK> kx, TapsArg<NK> ky,
                                                   int border, uint32_t skipx, uint32_t skipy, int tiles_x,
                                                   int n_tiles) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    using Stage = TileStage<PIX, NK, RPT>;
    constexpr int TH = Stage::TH;

    __shared__ Vec tile[Stage::LH * Stage::LW];

    const int lx = threadIdx.x & 63;
    // wave-uniform by construction; readfirstlane tells the compiler, so row indices, bounds tests and the
    // store descriptors live in SGPRs (otherwise hipcc wraps every buffer store in a waterfall loop)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // Workgroup b sits on XCD b % 8 (private L2 per XCD): number the workgroups XCD-major so that each XCD
    // sweeps one contiguous band of the image and neighbouring tiles (which share halo lines) hit the same L2.
    const int nwg = gridDim.x;
    const int per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);

    Stage st;
    if constexpr (!PERSIST) { // one tile per workgroup; the hardware dispatcher overlaps load and compute phases
        const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
        st.load(src, tx * TW, ty * TH, border, lx, wave);
        st.spill(tile, lx, wave);
        __syncthreads();
        convolve_tile<PIX, NK, MODE, SKIP, RPT, NT>(tile, src, dst, kx, ky, skipx, skipy, tx * TW, ty * TH, lx, wave);
    } else {
        // Tiles [t_begin, t_end) of this workgroup; tile t+1 is prefetched into registers while tile t is convolved.
        const int t_begin = (int)(((long long)n_tiles * wg) / nwg);
        const int t_end = (int)(((long long)n_tiles * (wg + 1)) / nwg);
        if (t_begin >= t_end) return;
        int ty = t_begin / tiles_x, tx = t_begin - ty * tiles_x;
        st.load(src, tx * TW, ty * TH, border, lx, wave);
        st.spill(tile, lx, wave);
        __syncthreads();
        // Steady state, straight-line per iteration: [loads of t+1] [convolve t: LDS reads, NK-row window, stores]
        // [barrier] [spill t+1 -> LDS] [barrier]. The spill waits on the loads only (vmcnt = #stores behind them).
        for (int t = t_begin; t + 1 < t_end; ++t) {
            int ny = ty, nx = tx + 1;
            if (nx == tiles_x) { nx = 0; ++ny; }
            st.load(src, nx * TW, ny * TH, border, lx, wave);
            convolve_tile<PIX, NK, MODE, SKIP, RPT, NT>(tile, src, dst, kx, ky, skipx, skipy, tx * TW, ty * TH, lx, wave);
            __syncthreads(); // every wave is done reading tile t
            st.spill(tile, lx, wave);
            __syncthreads();
            tx = nx;
            ty = ny;
        }
        convolve_tile<PIX, NK, MODE, SKIP, RPT, NT>(tile, src, dst, kx, ky, skipx, skipy, tx * TW, ty * TH, lx, wave);
    }
}

// ---- general two-pass fallback ----------------------------------------------------------------
// The temp plane of the two-pass path holds C lanes of Temp per pixel ([rows][cols][C], 3-channel pixels padded to 4
// lanes) so that one pixel of temps moves with one 4/16-byte access.
template <typename T, int C> struct TempVec { typedef T type __attribute__((ext_vector_type(C == 3 ? 4 : C))); };
template <int C> constexpr int temp_lanes() { return C == 3 ? 4 : C; }

// Up to MAX_TAPS taps travel as a kernel argument (1 KB): no table upload, no synchronisation, graph-capturable. Longer
// kernels (the reference's gaussianBlur has radius ceil(3 sigma), unbounded: image.zig:973) come from device memory.
struct TapsBig {
    union { float f[MAX_TAPS]; int32_t i[MAX_TAPS]; };
};

template <int PIX, int MODE>
__global__ __launch_bounds__(256) void k_sep_h(DImg src, typename Arith<MODE>::Temp *temp, TapsBig taps, const void *taps_mem,
                                               int nk, int border) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    using A = Arith<MODE>;
    constexpr int C = P::C;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    const int h = nk / 2;
    const bool interior = (src.cols > 2 * h) && c >= h && c < src.cols - h;
    typename A::Acc acc[C];
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
    const size_t row = (size_t)r * src.stride;
    for (int i = 0; i < nk; ++i) {
        if constexpr (MODE == MODE_F32) {
            const float k = taps_mem ? ((const float *)taps_mem)[i] : taps.f[i]; // wave-uniform either way
            if (interior && fabsf(k) < 1e-10f) continue;
            const int gc = resolve_index(c + i - h, src.cols, border);
            Vec v = P::zero();
            if (gc >= 0) v = P::load(src.data, row + gc);
            for (int ch = 0; ch < C; ++ch) acc[ch] = A::mac(acc[ch], v[ch], k);
        } else {
            const int32_t k = taps_mem ? ((const int32_t *)taps_mem)[i] : taps.i[i];
            const int gc = resolve_index(c + i - h, src.cols, border);
            Vec v = P::zero();
            if (gc >= 0) v = P::load(src.data, row + gc);
            for (int ch = 0; ch < C; ++ch) acc[ch] = A::mac(acc[ch], (int32_t)v[ch], k);
        }
    }
    using TV = typename TempVec<typename A::Temp, C>::type;
    TV tv;
#pragma unroll
    for (int ch = 0; ch < temp_lanes<C>(); ++ch) tv[ch] = ch < C ? A::to_temp(acc[ch < C ? ch : 0]) : (typename A::Temp)0;
    ((TV *)temp)[(size_t)r * src.cols + c] = tv;
}

template <int PIX, int MODE>
__global__ __launch_bounds__(256) void k_sep_v(const typename Arith<MODE>::Temp *temp, DImg dst, TapsBig taps, const void *taps_mem,
                                               int nk, int border) {
    using P = Px<PIX>;
    using Vec = typename P::Vec;
    using A = Arith<MODE>;
    constexpr int C = P::C;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = grid_row();
    if (c >= dst.cols || r >= dst.rows) return;
    const int h = nk / 2;
    const bool interior = (dst.rows > 2 * h) && r >= h && r < dst.rows - h;
    typename A::Acc acc[C];
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0;
    using TV = typename TempVec<typename A::Temp, C>::type;
    for (int i = 0; i < nk; ++i) {
        const int gr = resolve_index(r + i - h, dst.rows, border);
        TV tv;
#pragma unroll
        for (int ch = 0; ch < temp_lanes<C>(); ++ch) tv[ch] = (typename A::Temp)0;
        if (gr >= 0) tv = ((const TV *)temp)[(size_t)gr * dst.cols + c];
        if constexpr (MODE == MODE_F32) {
            const float k = taps_mem ? ((const float *)taps_mem)[i] : taps.f[i]; // wave-uniform either way
            if (interior && fabsf(k) < 1e-10f) continue;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = A::mac(acc[ch], tv[ch], k);
        } else {
            const int32_t k = taps_mem ? ((const int32_t *)taps_mem)[i] : taps.i[i];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = A::mac(acc[ch], tv[ch], k);
        }
    }
    Vec o;
    for (int ch = 0; ch < C; ++ch) {
        if constexpr (MODE == MODE_F32) o[ch] = acc[ch];
        else o[ch] = div_clamp_u8_sq<typename A::Acc>(acc[ch]);
    }
    P::store(dst.data, (size_t)r * dst.stride + c, o);
}

// ---- host dispatch ----------------------------------------------------------------------------
struct SepPlan {
    int nkx, nky;
    std::vector<float> fx, fy;
    std::vector<int32_t> ix, iy;
    uint32_t skipx = 0, skipy = 0;
    int mode = MODE_F32;
};

// Workgroups that are resident at once for `kernel` (256 threads, static LDS): CUs x blocks per CU, cached.
static int persistent_grid(const void *kernel) {
    static std::mutex mu;
    static std::unordered_map<const void *, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(kernel);
    if (it != cache.end()) return it->second;
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    const int grid = cus * std::min(per_cu, 8);
    cache[kernel] = grid;
    return grid;
}

template <int PIX, int NK, int MODE, bool SKIP, int RPT, bool PERSIST, bool NT>
static int launch_fused_v(const zg_image *src, const zg_image *dst, const SepPlan &p, int border, hipStream_t s) {
    TapsArg<NK> kx, ky;
    for (int i = 0; i < NK; ++i) {
        if constexpr (MODE == MODE_F32) { kx.f[i] = p.fx[i]; ky.f[i] = p.fy[i]; }
        else { kx.i[i] = p.ix[i]; ky.i[i] = p.iy[i]; }
    }
    const int tiles_x = (int)ceil_div(src->cols, TW), tiles_y = (int)ceil_div(src->rows, 4 * RPT);
    const int n_tiles = tiles_x * tiles_y;
    auto kernel = k_sep_fused<PIX, NK, MODE, SKIP, RPT, PERSIST, NT>;
    const int grid = PERSIST ? std::min(n_tiles, persistent_grid((const void *)kernel)) : n_tiles;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), 0, s,
                       dimg(src), dimg(dst), kx, ky, border, p.skipx, p.skipy, tiles_x, n_tiles);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

template <int PIX, int NK, int MODE, bool SKIP>
static int launch_fused(const zg_image *src, const zg_image *dst, const SepPlan &p, int border, hipStream_t s) {
    // measured on MI355X (profiles/r01_sep_variant_sweep.txt): 16-byte pixels like many small tiles (more
    // workgroups per CU overlap each other's load and compute phases); every type likes streaming stores.
    if constexpr (Px<PIX>::BYTES >= 12) return launch_fused_v<PIX, NK, MODE, SKIP, 4, false, true>(src, dst, p, border, s);
    else return launch_fused_v<PIX, NK, MODE, SKIP, 8, false, true>(src, dst, p, border, s);
}

template <int PIX, int NK, int MODE>
static int launch_fused_skip(const zg_image *src, const zg_image *dst, const SepPlan &p, int border, hipStream_t s) {
    if constexpr (MODE == MODE_F32) {
        if (p.skipx | p.skipy) return launch_fused<PIX, NK, MODE, true>(src, dst, p, border, s);
    }
    return launch_fused<PIX, NK, MODE, false>(src, dst, p, border, s);
}

template <int PIX, int MODE>
static int launch_fused_nk(const zg_image *src, const zg_image *dst, const SepPlan &p, int border, hipStream_t s) {
    switch (p.nkx) {
    case 1: return launch_fused_skip<PIX, 1, MODE>(src, dst, p, border, s);
    case 3: return launch_fused_skip<PIX, 3, MODE>(src, dst, p, border, s);
    case 5: return launch_fused_skip<PIX, 5, MODE>(src, dst, p, border, s);
    case 7: return launch_fused_skip<PIX, 7, MODE>(src, dst, p, border, s);
    case 9: if constexpr (MODE != MODE_I64) return launch_fused_skip<PIX, 9, MODE>(src, dst, p, border, s); else break;
    case 11: if constexpr (MODE != MODE_I64 && Px<PIX>::BYTES < 12) return launch_fused_skip<PIX, 11, MODE>(src, dst, p, border, s); else break;
    case 13: if constexpr (MODE != MODE_I64 && Px<PIX>::BYTES < 12) return launch_fused_skip<PIX, 13, MODE>(src, dst, p, border, s); else break;
    }
    return -1;
}

template <int PIX, int MODE>
static int launch_two_pass(const zg_image *src, const zg_image *dst, const SepPlan &p, int border, hipStream_t s) {
    using Temp = typename Arith<MODE>::Temp;
    constexpr int C = Px<PIX>::C;
    const size_t temp_bytes = (size_t)src->rows * src->cols * temp_lanes<C>() * sizeof(Temp);
    Temp *temp = nullptr;
    if (int rc = scratch_alloc((void **)&temp, temp_bytes, s)) return rc;
    TapsBig tx{}, ty{};
    void *taps_dev = nullptr; // kernels longer than MAX_TAPS: [kx | ky] in device memory (uploaded synchronously: not capturable)
    const void *mx = nullptr, *my = nullptr;
    if (p.nkx > MAX_TAPS || p.nky > MAX_TAPS) {
        const size_t n = (size_t)p.nkx + (size_t)p.nky;
        std::vector<uint32_t> host(n);
        if constexpr (MODE == MODE_F32) { std::memcpy(host.data(), p.fx.data(), (size_t)p.nkx * 4); std::memcpy(host.data() + p.nkx, p.fy.data(), (size_t)p.nky * 4); }
        else { std::memcpy(host.data(), p.ix.data(), (size_t)p.nkx * 4); std::memcpy(host.data() + p.nkx, p.iy.data(), (size_t)p.nky * 4); }
        int rc = scratch_alloc(&taps_dev, n * 4, s);
        if (rc == ZG_OK) rc = upload_pageable(taps_dev, host.data(), n * 4, s);
        if (rc) { scratch_free(taps_dev, s); scratch_free(temp, s); return rc; }
        mx = taps_dev;
        my = (const uint32_t *)taps_dev + p.nkx;
    } else {
        for (int i = 0; i < p.nkx; ++i) { if constexpr (MODE == MODE_F32) tx.f[i] = p.fx[i]; else tx.i[i] = p.ix[i]; }
        for (int i = 0; i < p.nky; ++i) { if constexpr (MODE == MODE_F32) ty.f[i] = p.fy[i]; else ty.i[i] = p.iy[i]; }
    }
    const dim3 grid = row_grid(ceil_div(src->cols, 256), src->rows);
    hipLaunchKernelGGL((k_sep_h<PIX, MODE>), grid, dim3(256), 0, s, dimg(src), temp, tx, mx, p.nkx, border);
    hipLaunchKernelGGL((k_sep_v<PIX, MODE>), grid, dim3(256), 0, s, temp, dimg(dst), ty, my, p.nky, border);
    const hipError_t launch_error = hipGetLastError();
    scratch_free(taps_dev, s);
    if (launch_error != hipSuccess) { scratch_free(temp, s); ZG_HIP(launch_error); }
    ZG_HIP(hipGetLastError());
    scratch_free(temp, s);
    return ZG_OK;
}

template <int PIX, int MODE>
static int run_sep(const zg_image *src, const zg_image *dst, const SepPlan &p, int border, hipStream_t s) {
    // The fused kernel keeps NK rows of temps per lane in registers; where that window does not fit the register file hipcc spills,
    // and the one spilling instantiation ever exercised (13 taps, 16-byte pixels: 512 registers + 676 bytes of scratch) came back with
    // a wrong third channel under hipcc 7.2 (tests/test_gpu_aligned_shapes.py pins the case). So no spilling instantiation exists:
    // 12- and 16-byte pixels stop at 9 taps, the i64 path at 7; longer kernels take the two-pass kernels (images of 64 columns and
    // more have gone to conv_sep_f32long.hip / conv_sep_bytes2.hip before they get here anyway). The build checks it
    // (tools/isa_report.py --check).
    constexpr int NK_MAX = MODE == MODE_I64 ? 7 : (Px<PIX>::BYTES >= 12 ? 9 : 13);
    const bool fused_ok = p.nkx <= NK_MAX;
    if (p.nkx == p.nky && p.nkx <= 13 && (p.nkx & 1) && fused_ok) {
        const int rc = launch_fused_nk<PIX, MODE>(src, dst, p, border, s);
        if (rc >= 0) return rc;
    }
    return launch_two_pass<PIX, MODE>(src, dst, p, border, s);
}

int try_sep_rgba8(const zg_image *src, const zg_image *dst, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s);
int try_sep_bytes(const zg_image *src, const zg_image *dst, const int32_t *ix, const int32_t *iy, int nk, int border, hipStream_t s);
int try_sep_f32long(const zg_image *src, const zg_image *dst, const float *fx, int nkx, const float *fy, int nky, int border, hipStream_t s);
int try_sep_bytes2(const zg_image *src, const zg_image *dst, const int32_t *ix, int nkx, const int32_t *iy, int nky, int border, hipStream_t s);
int try_sep_f32x4(const zg_image *src, const zg_image *dst, const float *fx, const float *fy, int nk, uint32_t skipx, uint32_t skipy,
                  int border, hipStream_t s);
constexpr uint32_t SF_MAX_PLANES = 8; // planes per launch of conv_sep_tile_f32.hip
int try_sep_tile_f32(const zg_image *src, const zg_image *dst, uint32_t n, const float *fx, const float *fy, int nk, uint32_t skipx, uint32_t skipy,
                       int border, hipStream_t s);

static int conv_separable_impl(const zg_image *src, const zg_image *dst, const float *kx, uint32_t nkx,
                               const float *ky, uint32_t nky, int border, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH,
               "convolveSeparable: %ux%u vs %ux%u", src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->pixel == dst->pixel, ZG_ERR_INVALID_ARGUMENT, "convolveSeparable: pixel types differ");
    ZG_REQUIRE(kx && ky && nkx >= 1 && nky >= 1 && nkx <= MAX_TAPS_MEM && nky <= MAX_TAPS_MEM, ZG_ERR_INVALID_ARGUMENT,
               "convolveSeparable: kernel lengths %u, %u (1..%d supported)", nkx, nky, MAX_TAPS_MEM);
    ZG_REQUIRE(border >= ZG_BORDER_ZERO && border <= ZG_BORDER_WRAP, ZG_ERR_INVALID_ARGUMENT, "invalid border %d", border);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;

    SepPlan p;
    p.nkx = (int)nkx;
    p.nky = (int)nky;
    const bool is_float = pixel_is_float(src->pixel);
    if (is_float) {
        p.mode = MODE_F32;
        p.fx.assign(kx, kx + nkx);
        p.fy.assign(ky, ky + nky);
        for (uint32_t i = 0; i < nkx && i < 32; ++i) if (std::fabs(kx[i]) < 1e-10f) p.skipx |= 1u << i;
        for (uint32_t i = 0; i < nky && i < 32; ++i) if (std::fabs(ky[i]) < 1e-10f) p.skipy |= 1u << i;
        if (src->pixel == ZG_PIXEL_F32 && p.nkx == p.nky) { // single-channel planes: four pixels per lane
            const int rcs = try_sep_tile_f32(src, dst, 1, p.fx.data(), p.fy.data(), p.nkx, p.skipx, p.skipy, border, s); // one wave per tile, no LDS
            if (rcs >= 0) return rcs;
            const int rc4 = try_sep_f32x4(src, dst, p.fx.data(), p.fy.data(), p.nkx, p.skipx, p.skipy, border, s);
            if (rc4 >= 0) return rc4;
        }
        if (!(p.nkx == p.nky && p.nkx <= 9 && (p.nkx & 1))) { // long (or unequal) kernels: two coalesced passes through an f32 temp plane
            const int rcl = try_sep_f32long(src, dst, p.fx.data(), p.nkx, p.fy.data(), p.nky, border, s);
            if (rcl >= 0) return rcl;
        }
    } else {
        // scaleKernelToInt (convolution.zig:303-309): @round(k * 256) -> i32
        p.ix.resize(nkx);
        p.iy.resize(nky);
        int64_t sax = 0, say = 0, mx = 0, my = 0;
        for (uint32_t i = 0; i < nkx; ++i) {
            const float r = std::round(kx[i] * 256.0f);
            ZG_REQUIRE(std::fabs(r) < 2147483648.0f, ZG_ERR_INVALID_ARGUMENT, "kernel_x[%u] does not fit i32 after scaling", i);
            p.ix[i] = (int32_t)r;
            sax += std::llabs((long long)p.ix[i]);
            mx = std::max<int64_t>(mx, std::llabs((long long)p.ix[i]));
        }
        for (uint32_t i = 0; i < nky; ++i) {
            const float r = std::round(ky[i] * 256.0f);
            ZG_REQUIRE(std::fabs(r) < 2147483648.0f, ZG_ERR_INVALID_ARGUMENT, "kernel_y[%u] does not fit i32 after scaling", i);
            p.iy[i] = (int32_t)r;
            say += std::llabs((long long)p.iy[i]);
            my = std::max<int64_t>(my, std::llabs((long long)p.iy[i]));
        }
        // Outer taps that round to zero add exactly nothing to an integer sum whatever the border rule hands them: drop them in pairs (the kernel stays
        // centred). gaussianBlur's 3-sigma radius leaves such taps from sigma 2.3 up (ORB's pyramid levels: 35 taps -> 29, 29 -> 25, 19 -> 17, 9 -> 7).
        auto trim_zero_ends = [](std::vector<int32_t> &k, int &nk) {
            int z = 0;
            while (nk - 2 * z > 2 && k[(size_t)z] == 0 && k[(size_t)(nk - 1 - z)] == 0) ++z;
            if (z) {
                k.erase(k.begin(), k.begin() + z);
                k.resize((size_t)(nk - 2 * z));
                nk -= 2 * z;
            }
        };
        static const bool keep_zero_taps = getenv("ZIGNAL_HIP_KEEP_ZERO_TAPS") != nullptr; // A/B hook of round 5, read once
        if (!keep_zero_taps) {
            trim_zero_ends(p.ix, p.nkx);
            trim_zero_ends(p.iy, p.nky);
        }
        // i24 multiplies are exact when both operands fit 24 signed bits and no sum leaves i32.
        const int64_t max_temp = 255 * sax;
        const bool fits = mx < (1 << 23) && my < (1 << 23) && max_temp < (1 << 23) &&
                          max_temp * say < (int64_t)INT32_MAX - 65536;
        p.mode = fits ? MODE_I24 : MODE_I64;
        if (p.nkx == p.nky) { // small non-negative taps: packed-u16 arithmetic on the row as a byte stream, 16 bytes / lane
            const size_t sp = pixel_size(src->pixel);
            const StreamJob job{src->data, dst->data, 1, src->rows, src->cols, (int)sp, src->stride * sp, dst->stride * sp, 0, 0, false};
            const int rcs = try_sep_stream(job, p.ix.data(), p.iy.data(), p.nkx, border, s); // one wave per column strip, no LDS
            if (rcs >= 0) return rcs;
            const int rcb = try_sep_bytes(src, dst, p.ix.data(), p.iy.data(), p.nkx, border, s);
            if (rcb >= 0) return rcb;
        }
        // longer (or unequal) non-negative kernels: two packed passes through a u16 temp plane
        const int rcb2 = try_sep_bytes2(src, dst, p.ix.data(), p.nkx, p.iy.data(), p.nky, border, s);
        if (rcb2 >= 0) return rcb2;
    }

    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        if constexpr (std::is_same<typename Px<PIX>::Elem, float>::value) {
            return run_sep<PIX, MODE_F32>(src, dst, p, border, s);
        } else {
            if (p.mode == MODE_I24) return run_sep<PIX, MODE_I24>(src, dst, p, border, s);
            return run_sep<PIX, MODE_I64>(src, dst, p, border, s);
        }
    });
}

// n planes of one shape through convolveSeparable: one launch per SF_MAX_PLANES of them where the f32 tile kernel applies (the planes
// then only have to agree in shape, strides and alignment), plane by plane otherwise. Every plane is validated before anything is launched.
static int conv_separable_planes_impl(const zg_image *src, const zg_image *dst, uint32_t n, const float *kx, uint32_t nkx,
                                      const float *ky, uint32_t nky, int border, hipStream_t s) {
    ZG_REQUIRE(n == 0 || (src && dst), ZG_ERR_INVALID_ARGUMENT, "convolveSeparable (planes): null plane array");
    ZG_REQUIRE(kx && ky && nkx >= 1 && nky >= 1 && nkx <= MAX_TAPS_MEM && nky <= MAX_TAPS_MEM, ZG_ERR_INVALID_ARGUMENT,
               "convolveSeparable: kernel lengths %u, %u (1..%d supported)", nkx, nky, MAX_TAPS_MEM);
    ZG_REQUIRE(border >= ZG_BORDER_ZERO && border <= ZG_BORDER_WRAP, ZG_ERR_INVALID_ARGUMENT, "invalid border %d", border);
    for (uint32_t p = 0; p < n; ++p) {
        int rc;
        if ((rc = check_image(&src[p], "src")) || (rc = check_image(&dst[p], "dst"))) return rc;
        ZG_REQUIRE(src[p].rows == dst[p].rows && src[p].cols == dst[p].cols, ZG_ERR_DIMENSION_MISMATCH,
                   "convolveSeparable: plane %u is %ux%u vs %ux%u", p, src[p].rows, src[p].cols, dst[p].rows, dst[p].cols);
        ZG_REQUIRE(src[p].pixel == dst[p].pixel, ZG_ERR_INVALID_ARGUMENT, "convolveSeparable: pixel types of plane %u differ", p);
    }
    uint32_t p = 0;
    while (p < n) {
        uint32_t run = 1; // planes p .. p + run - 1 share a launch
        if (src[p].pixel == ZG_PIXEL_F32 && nkx == nky && src[p].rows && src[p].cols) {
            uint32_t skipx = 0, skipy = 0;
            for (uint32_t i = 0; i < nkx && i < 32; ++i) {
                if (std::fabs(kx[i]) < 1e-10f) skipx |= 1u << i;
                if (std::fabs(ky[i]) < 1e-10f) skipy |= 1u << i;
            }
            while (run < SF_MAX_PLANES && p + run < n && src[p + run].pixel == ZG_PIXEL_F32 && src[p + run].rows == src[p].rows &&
                   src[p + run].cols == src[p].cols && src[p + run].stride == src[p].stride && dst[p + run].stride == dst[p].stride)
                ++run;
            const int rcs = try_sep_tile_f32(src + p, dst + p, run, kx, ky, (int)nkx, skipx, skipy, border, s);
            if (rcs >= 0) {
                if (rcs != ZG_OK) return rcs;
                p += run;
                continue;
            }
            run = 1;
        }
        if (int rc = conv_separable_impl(&src[p], &dst[p], kx, nkx, ky, nky, border, s)) return rc;
        p += run;
    }
    return ZG_OK;
}

int copy_impl(const zg_image *src, const zg_image *dst, hipStream_t s);

} // namespace zg

using namespace zg;

extern "C" {

int zg_conv_separable(const zg_image *src, const zg_image *dst, const float *kx, uint32_t nkx,
                      const float *ky, uint32_t nky, int border, zg_stream stream) {
    return conv_separable_impl(src, dst, kx, nkx, ky, nky, border, as_stream(stream));
}

int zg_conv_separable_host(const zg_image *src, const zg_image *dst, const float *kx, uint32_t nkx,
                           const float *ky, uint32_t nky, int border) {
    if (border != ZG_BORDER_WRAP && nky >= 1) { // row-local with a halo of nky / 2 rows: upload, kernel and download overlap band by band
        const int brc = host_banded(src, dst, nky / 2, [&](const zg_image *sv, const zg_image *dv, hipStream_t s) {
            return conv_separable_impl(sv, dv, kx, nkx, ky, nky, border, s);
        });
        if (brc >= 0) return brc;
    }
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = conv_separable_impl(&a.dev, &b.dev, kx, nkx, ky, nky, border, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

int zg_conv_separable_planes(const zg_image *src, const zg_image *dst, uint32_t n_planes, const float *kx, uint32_t nkx,
                             const float *ky, uint32_t nky, int border, zg_stream stream) {
    return conv_separable_planes_impl(src, dst, n_planes, kx, nkx, ky, nky, border, as_stream(stream));
}

// image.zig:973-990
int zg_gaussian_kernel(float sigma, float *taps, uint32_t capacity) {
    if (!(sigma > 0)) { set_error("gaussian kernel: sigma must be > 0"); return -ZG_ERR_INVALID_ARGUMENT; }
    const float rf = std::ceil(3.0f * sigma);
    if (!(rf < (float)(MAX_TAPS_MEM / 2))) { set_error("gaussian kernel: sigma %g too large", sigma); return -ZG_ERR_INVALID_ARGUMENT; }
    const uint32_t radius = (uint32_t)rf, size = 2 * radius + 1;
    if (!taps) return (int)size;
    if (capacity < size) { set_error("gaussian kernel: capacity %u < %u", capacity, size); return -ZG_ERR_INVALID_ARGUMENT; }
    float sum = 0;
    for (uint32_t i = 0; i < size; ++i) {
        const float x = (float)i - (float)radius;
        taps[i] = hostmath::exp_f32(-(x * x) / (2.0f * sigma * sigma));
        sum += taps[i];
    }
    for (uint32_t i = 0; i < size; ++i) taps[i] /= sum;
    return (int)size;
}

static int gaussian_impl(const zg_image *src, const zg_image *dst, float sigma, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH,
               "gaussianBlur: %ux%u vs %ux%u", src->rows, src->cols, dst->rows, dst->cols);
    if (sigma == 0) return copy_impl(src, dst, s);                       // image.zig:966
    ZG_REQUIRE(sigma > 0, ZG_ERR_INVALID_ARGUMENT, "gaussianBlur: InvalidSigma (%g)", sigma); // image.zig:970
    const int n = zg_gaussian_kernel(sigma, nullptr, 0);
    if (n < 0) return -n;
    std::vector<float> taps((size_t)n);
    if (zg_gaussian_kernel(sigma, taps.data(), (uint32_t)n) != n) return ZG_ERR_INVALID_ARGUMENT;
    return conv_separable_impl(src, dst, taps.data(), (uint32_t)n, taps.data(), (uint32_t)n, ZG_BORDER_MIRROR, s);
}

int zg_gaussian_blur(const zg_image *src, const zg_image *dst, float sigma, zg_stream stream) {
    return gaussian_impl(src, dst, sigma, as_stream(stream));
}

int zg_gaussian_blur_planes(const zg_image *src, const zg_image *dst, uint32_t n_planes, float sigma, zg_stream stream) {
    hipStream_t s = as_stream(stream);
    ZG_REQUIRE(n_planes == 0 || (src && dst), ZG_ERR_INVALID_ARGUMENT, "gaussianBlur (planes): null plane array");
    if (!(sigma > 0)) { // sigma == 0 copies, anything else is error.InvalidSigma: per plane, exactly as zg_gaussian_blur
        for (uint32_t p = 0; p < n_planes; ++p)
            if (int rc = gaussian_impl(&src[p], &dst[p], sigma, s)) return rc;
        return ZG_OK;
    }
    for (uint32_t p = 0; p < n_planes; ++p) {
        int rc;
        if ((rc = check_image(&src[p], "src")) || (rc = check_image(&dst[p], "dst"))) return rc;
        ZG_REQUIRE(src[p].rows == dst[p].rows && src[p].cols == dst[p].cols, ZG_ERR_DIMENSION_MISMATCH,
                   "gaussianBlur: plane %u is %ux%u vs %ux%u", p, src[p].rows, src[p].cols, dst[p].rows, dst[p].cols);
    }
    const int n = zg_gaussian_kernel(sigma, nullptr, 0);
    if (n < 0) return -n;
    std::vector<float> taps((size_t)n);
    if (zg_gaussian_kernel(sigma, taps.data(), (uint32_t)n) != n) return ZG_ERR_INVALID_ARGUMENT;
    return conv_separable_planes_impl(src, dst, n_planes, taps.data(), (uint32_t)n, taps.data(), (uint32_t)n, ZG_BORDER_MIRROR, s);
}

int zg_gaussian_blur_host(const zg_image *src, const zg_image *dst, float sigma) {
    if (sigma > 0) { // gaussianBlur is convolveSeparable(.mirror) with ceil(3 sigma) rows of halo (image.zig:973-994)
        const int n = zg_gaussian_kernel(sigma, nullptr, 0);
        if (n > 0) {
            const int brc = host_banded(src, dst, (uint32_t)n / 2, [&](const zg_image *sv, const zg_image *dv, hipStream_t s) { return gaussian_impl(sv, dv, sigma, s); });
            if (brc >= 0) return brc;
        }
    }
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = gaussian_impl(&a.dev, &b.dev, sigma, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

} // extern "C"
