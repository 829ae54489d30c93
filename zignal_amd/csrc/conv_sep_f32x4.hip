// conv_sep_f32x4.hip — Image(f32).convolveSeparable / gaussianBlur on single-channel f32 planes, four pixels per lane.
//
// Same arithmetic contract as conv_separable.hip (reference src/image/convolution.zig:441-647, f32 path): per pixel
// temp = sum_i src[c+i-h] * kx[i] and out = sum_i temp[r+i-h] * ky[i], ascending i from an accumulator of 0, separate
// multiply and add, interior pixels skipping taps with |k| < 1e-10. Only the work distribution differs from the general
// kernel: a 4-byte pixel per lane leaves the memory system three quarters idle, so here each lane owns FOUR adjacent
// pixels (one 16-byte load / LDS access / store) and a workgroup covers a 256 x 4*RPT tile. BASELINE configs[1] in its
// "four Image(f32) planes" form runs on this kernel.
//
// Preconditions (else the general kernel runs): f32, cols % 4 == 0, strides % 4 == 0, 16-byte aligned bases,
// cols >= 64, odd equal tap counts <= 9.
#include "zg_common.h"

#pragma clang fp contract(off)

namespace zg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

template <int N> struct TapsF32 { float k[N]; };

constexpr int F4_TW = 256;   // tile width in pixels = 64 lanes x 4 pixels
constexpr int F4_UNITS = 66; // 16-byte units per LDS row: one halo unit left, 64, one right

template <int NK, int RPT> struct StageF4 {
    static constexpr int H = NK / 2;
    static constexpr int LH = 4 * RPT + 2 * H;
    static constexpr int RW = (LH + 3) / 4;
    static constexpr int NEXTRA = LH * 2;
    f32x4 main_v[RW];
    f32x4 extra_v;

    // tile row r, unit u: pixels x0 - 4 + 4u .. +3 of image row y0 - H + r. Units are all inside or all outside the row
    // (cols % 4 == 0); outside ones (and rows the zero border drops) become 0 here and the pixels of them that the taps
    // can reach are filled in by patch_edges. The load itself is unconditional from a clamped address: predicated loads
    // would be issued one at a time.
    __device__ static __forceinline__ f32x4 load_unit(const DImg &src, int x0, int y0, int border, int r, int u) {
        const int gr = resolve_index(y0 - H + r, src.rows, border);
        const int gx = x0 - 4 + 4 * u;
        const bool ok = gr >= 0 && gx >= 0 && gx + 4 <= src.cols;
        const float *row = (const float *)src.data + (size_t)max(gr, 0) * src.stride;
        f32x4 v = *(const f32x4 *)(row + min(max(gx, 0), src.cols - 4)); // 16-byte aligned by the preconditions
        if (!ok) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        return v;
    }
    __device__ __forceinline__ void load(const DImg &src, int x0, int y0, int border, int lx, int wave) {
#pragma unroll
        for (int k = 0; k < RW; ++k) main_v[k] = load_unit(src, x0, y0, border, min(wave + 4 * k, LH - 1), lx);
        const int e = min((int)threadIdx.x, NEXTRA - 1); // lanes past NEXTRA load a duplicate and do not spill it
        extra_v = load_unit(src, x0, y0, border, e >> 1, 64 + (e & 1));
    }
    // Border rule for the columns: the H pixels left of column 0 and right of the last column, where this tile covers
    // them, one pixel per lane straight from global memory into the LDS tile (edge tiles only).
    __device__ static void patch_edges(f32x4 *tile, const DImg &src, int x0, int y0, int border) {
        for (int idx = (int)threadIdx.x; idx < LH * 2 * H; idx += 256) {
            const int r = idx / (2 * H), k = idx - r * (2 * H);
            const int px = k < H ? -1 - k : src.cols + (k - H);
            const int t = px - (x0 - 4); // pixel position in the tile row
            if (t < 0 || t >= F4_UNITS * 4) continue;
            const int gr = resolve_index(y0 - H + r, src.rows, border);
            const int gc = resolve_index(px, src.cols, border);
            if (gr < 0 || gc < 0) continue; // zero border: already 0
            ((float *)tile)[(size_t)r * F4_UNITS * 4 + t] = ((const float *)src.data)[(size_t)gr * src.stride + gc];
        }
    }
    __device__ void spill(f32x4 *tile, int lx, int wave) const {
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int r = wave + 4 * k;
            if (r < LH) tile[r * F4_UNITS + lx] = main_v[k];
        }
        const int e = (int)threadIdx.x;
        if (e < NEXTRA) tile[(e >> 1) * F4_UNITS + 64 + (e & 1)] = extra_v;
    }
};

template <int NK, int RPT, bool SKIP>
__global__ __launch_bounds__(256) void k_sep_f32x4(DImg src, DImg dst, TapsF32<NK> kx, TapsF32<NK> ky, int border,
                                                   uint32_t skipx, uint32_t skipy, int tiles_x) {
    using Stage = StageF4<NK, RPT>;
    constexpr int H = NK / 2;
    constexpr int TH = 4 * RPT;
    __shared__ f32x4 tile[Stage::LH * F4_UNITS];

    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3); // XCD-major tile order
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int x0 = tx * F4_TW, y0 = ty * TH;
    const int lx = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    Stage st;
    st.load(src, x0, y0, border, lx, wave);
    st.spill(tile, lx, wave);
    if (x0 == 0 || x0 + F4_TW + 4 > src.cols) { // workgroup-uniform: this tile sees the left or right border
        __syncthreads();
        Stage::patch_edges(tile, src, x0, y0, border);
    }
    __syncthreads();

    const int gx = x0 + 4 * lx;
    bool col_interior[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) col_interior[p] = (src.cols > 2 * H) && (gx + p) >= H && (gx + p) < src.cols - H;
    const bool rows_have_interior = src.rows > 2 * H;

    float win[NK][4];
#pragma unroll
    for (int j = 0; j < RPT + 2 * H; ++j) {
        const int lr = wave * RPT + j;
        const f32x4 a = tile[lr * F4_UNITS + lx], b = tile[lr * F4_UNITS + lx + 1], c = tile[lr * F4_UNITS + lx + 2];
        const float q[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]}; // q[4] = pixel gx
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                if (SKIP && col_interior[p] && ((skipx >> i) & 1u)) continue;
                const float prod = q[4 - H + p + i] * kx.k[i];
                acc = acc + prod;
            }
            win[j % NK][p] = acc;
        }
        if (j >= 2 * H) {
            const int gy = y0 + wave * RPT + (j - 2 * H);
            const bool row_interior = rows_have_interior && gy >= H && gy < src.rows - H;
            f32x4 o;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    if (SKIP && row_interior && ((skipy >> i) & 1u)) continue;
                    const float prod = win[(j + 1 + i) % NK][p] * ky.k[i];
                    acc = acc + prod;
                }
                o[p] = acc;
            }
            const bool row_ok = gy < dst.rows;
            char *row = (char *)dst.data + (row_ok ? (size_t)gy * dst.stride * 4 : (size_t)0);
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, row_ok ? dst.cols * 4 : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o), rsrc, gx * 4, 0, 2); // nt; cols % 4 == 0
        }
    }
}

template <int NK, int RPT>
static int launch_f32x4(const zg_image *src, const zg_image *dst, const float *fx, const float *fy, uint32_t skipx, uint32_t skipy,
                        int border, hipStream_t s) {
    TapsF32<NK> kx, ky;
    for (int i = 0; i < NK; ++i) { kx.k[i] = fx[i]; ky.k[i] = fy[i]; }
    const int tiles_x = (int)ceil_div(src->cols, F4_TW), tiles_y = (int)ceil_div(src->rows, 4 * RPT);
    const dim3 grid((unsigned)(tiles_x * tiles_y));
    if (skipx | skipy)
        hipLaunchKernelGGL((k_sep_f32x4<NK, RPT, true>), grid, dim3(256), 0, s, dimg(src), dimg(dst), kx, ky, border, skipx, skipy, tiles_x);
    else
        hipLaunchKernelGGL((k_sep_f32x4<NK, RPT, false>), grid, dim3(256), 0, s, dimg(src), dimg(dst), kx, ky, border, skipx, skipy, tiles_x);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// Returns -1 when the preconditions do not hold (caller falls back to the general kernel).
int try_sep_f32x4(const zg_image *src, const zg_image *dst, const float *fx, const float *fy, int nk, uint32_t skipx, uint32_t skipy,
                  int border, hipStream_t s) {
    if (src->pixel != ZG_PIXEL_F32 || (nk != 3 && nk != 5 && nk != 7 && nk != 9)) return -1;
    if (src->cols % 4 || src->stride % 4 || dst->stride % 4 || ((uintptr_t)src->data & 15) || ((uintptr_t)dst->data & 15)) return -1;
    if (src->cols < 64) return -1;
    switch (nk) { // RPT 4 measured best on MI355X (RPT 1 / 2 / 4 / 8: 47.1 / 36.5 / 34.9 / 40.0 us per 4096^2 plane)
    case 3: return launch_f32x4<3, 4>(src, dst, fx, fy, skipx, skipy, border, s);
    case 5: return launch_f32x4<5, 4>(src, dst, fx, fy, skipx, skipy, border, s);
    case 7: return launch_f32x4<7, 4>(src, dst, fx, fy, skipx, skipy, border, s);
    case 9: return launch_f32x4<9, 4>(src, dst, fx, fy, skipx, skipy, border, s);
    }
    return -1;
}

} // namespace zg
