// edges.hip — Image(T).sobel as ONE kernel, plus the host arithmetic of ImagePyramid.build.
//
// Replaces reference src/image.zig:1001-1010 -> src/image/edges.zig:33-70: grey f32 plane (as(f32, convertColor(u8, px)),
// scalar f32 input used as is) -> convolve(sobel_x, .replicate) and convolve(sobel_y, .replicate) in f32 (ky-major, every
// tap including the zero ones, separate mul and add: src/image/convolution.zig:158-169) -> sqrt(gx^2 + gy^2) / 4 ->
// @trunc(@max(0, @min(255, .))). The reference materialises three full f32 planes; here a workgroup stages the grey values
// of its 64 x 4 tile plus a one-pixel replicate halo in LDS and writes only the u8 result: traffic = source once + 1 B/px.
#include "zg_common.h"
#include "zg_hostmath.h"

#include <cmath>
#include <map>
#include <vector>

#pragma clang fp contract(off)

namespace zg {

int try_sobel_stream(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s); // sobel_stream.hip

int try_sep_f32long_grey(const zg_image *src, const zg_image *dst, const float *fx, int nkx, const float *fy, int nky, int border, hipStream_t s); // conv_sep_f32long.hip

__device__ inline float gray_as_f32(uint8_t v) { return (float)v; }

template <int PIX> __device__ inline float sobel_gray(typename Px<PIX>::Vec v) {
    using P = Px<PIX>;
    if constexpr (PIX == ZG_PIXEL_F32) return v[0];
    else if constexpr (PIX == ZG_PIXEL_U8) return (float)v[0];
    else if constexpr (!std::is_same<typename P::Elem, float>::value) { // Rgb(u8) / Rgba(u8): BT.709 16.16 fixed point (color.zig:1031-1042)
        const int y = (13933 * (int)v[0] + 46871 * (int)v[1] + 4732 * (int)v[2] + 32768) >> 16;
        return (float)(y < 0 ? 0 : (y > 255 ? 255 : y));
    } else { // Rgb(f32) / Rgba(f32): clamp(dot, 0, 1) then .as(u8) = @round(255 * clamp) (color.zig:1043-1046, :528-546)
        float y = 0.2126f * v[0] + 0.7152f * v[1] + 0.0722f * v[2];
        y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
        const float s = 255.0f * y;
        return (float)(int)roundf(s);
    }
}

// Sobel sums of a 3 x 3 window in the reference's term order (edges.zig:255-262 / image.zig sobel): the zero weights are skipped
// and the +-1 / +-2 weights folded into additions; the products are exact, so only the sign of a zero can differ.
__device__ inline void sobel_3x3(const float (&p)[3][3], float &gx, float &gy) {
    gx = ((((-p[0][0] + p[0][2]) - 2.0f * p[1][0]) + 2.0f * p[1][2]) - p[2][0]) + p[2][2];
    gy = ((((-p[0][0] - 2.0f * p[0][1]) - p[0][2]) + p[2][0]) + 2.0f * p[2][1]) + p[2][2];
}

// Tile 64 x 16 outputs, grey values one pixel around it in LDS (a wave per row, .replicate by clamped coordinates); a thread owns
// four consecutive rows of one column, whose windows share six rows of three values; output bytes leave as dwords through LDS.
// (Round 1: 64 x 4 tile, nine clamped taps per pixel, byte stores: 67 us per 4096^2 Rgba(u8) frame.)
template <int PIX>
__global__ __launch_bounds__(256) void k_sobel(DImg src, DImg dst, int tiles_x, FrameSpan fr) {
    using P = Px<PIX>;
    constexpr int TH = 16;
    __shared__ float g[TH + 2][68];
    __shared__ uint8_t ob[TH][64];
    src.data = (char *)src.data + (size_t)blockIdx.y * fr.src_frame; // a batch of equally shaped frames (zg_batch_pipeline's edges step)
    dst.data = (char *)dst.data + (size_t)blockIdx.y * fr.dst_frame;
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int x0 = tx * 64, y0 = ty * TH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const int gc = min(max(x0 - 1 + lane, 0), src.cols - 1), gc2 = min(max(x0 + 63 + lane, 0), src.cols - 1);
        for (int rr = w; rr < TH + 2; rr += 4) {
            const size_t row = (size_t)min(max(y0 - 1 + rr, 0), src.rows - 1) * src.stride;
            g[rr][lane] = sobel_gray<PIX>(P::load(src.data, row + (size_t)gc));
            if (lane < 2) g[rr][64 + lane] = sobel_gray<PIX>(P::load(src.data, row + (size_t)gc2));
        }
    }
    __syncthreads();
    float p[6][3];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) p[j][i] = g[4 * w + j][lane + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float win[3][3] = {{p[k][0], p[k][1], p[k][2]}, {p[k + 1][0], p[k + 1][1], p[k + 1][2]}, {p[k + 2][0], p[k + 2][1], p[k + 2][2]}};
        float ax, ay;
        sobel_3x3(win, ax, ay);
        const float sx = ax * ax, sy = ay * ay;
        const float magnitude = sqrtf(sx + sy);
        const float scaled = magnitude / 4.0f;
        const float clamped = fmaxf(0.0f, fminf(255.0f, scaled));
        ob[4 * w + k][lane] = (uint8_t)(int)truncf(clamped);
    }
    __syncthreads();
    const int lr = threadIdx.x >> 4, q = (threadIdx.x & 15) * 4, r = y0 + lr;
    if (r < dst.rows && x0 + q < dst.cols) {
        uint8_t *o = (uint8_t *)dst.data + (size_t)r * dst.stride + x0 + q;
        if (x0 + q + 4 <= dst.cols && (dst.stride & 3) == 0 && ((uintptr_t)dst.data & 3) == 0) {
            *(uint32_t *)o = *(const uint32_t *)&ob[lr][q];
        } else {
            for (int j = 0; j < 4 && x0 + q + j < dst.cols; ++j) o[j] = ob[lr][q + j];
        }
    }
}

static int sobel_impl(const zg_image *src, const zg_image *dst, hipStream_t s, uint32_t n = 1, size_t src_frame = 0, size_t dst_frame = 0) {
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "sobel: %ux%u vs %ux%u",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(dst->pixel == ZG_PIXEL_U8, ZG_ERR_INVALID_ARGUMENT, "sobel: the output is Image(u8)");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    if (const int rcs = try_sobel_stream(src, dst, n, src_frame, dst_frame, s); rcs >= 0) return rcs; // u8 / Rgba(u8): one wave per column strip
    const int tiles_x = (int)ceil_div(src->cols, 64), tiles_y = (int)ceil_div(src->rows, 16);
    if (n > MAX_FRAMES_PER_LAUNCH) return -1; // the caller goes frame by frame
    const FrameSpan fr{src_frame, dst_frame};
    return dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        hipLaunchKernelGGL((k_sobel<PIX>), dim3((unsigned)(tiles_x * tiles_y), n), dim3(256), 0, s, dimg(src), dimg(dst), tiles_x, fr);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
}
// Image.sobel of n equally shaped frames in one launch (-1: too many frames for one grid)
int sobel_frames(const zg_image *src, const zg_image *dst, uint32_t n, size_t src_frame, size_t dst_frame, hipStream_t s) {
    return sobel_impl(src, dst, s, n, src_frame, dst_frame);
}


// ---- Canny (src/image.zig:1047-1063 -> src/image/edges.zig:212-277) --------------------------------------------------
// grey plane as(f32, convertColor(u8, px)) -> the detector's own Gaussian (.replicate) -> Sobel gradients -> magnitude ->
// non-maximum suppression -> double threshold + hysteresis. The reference materialises six full planes and walks a BFS
// queue; here: one grey kernel, the library's separable convolution, ONE fused kernel for gradients + magnitude + NMS +
// classification (a 2-pixel halo of the blurred plane in LDS; its output is a single byte per pixel: 0 none, 1 weak, 2
// strong), and hysteresis as connected-component labelling (see run_hysteresis): the BFS's reachable set, in a fixed
// number of kernels.

template <int PIX> __device__ inline float canny_gray(typename Px<PIX>::Vec v) { // as(f32, convertColor(u8, px)), edges.zig:231-240
    if constexpr (PIX == ZG_PIXEL_F32) { // scalar float -> u8 in f64 (color.zig:114-118)
        double d = (double)v[0];
        d = d < 0.0 ? 0.0 : (d > 1.0 ? 1.0 : d);
        return (float)(int)round(d * 255.0);
    } else return sobel_gray<PIX>(v);
}

template <int PIX>
__global__ __launch_bounds__(256) void k_canny_gray(DImg src, float *gray) {
    using P = Px<PIX>;
    const int c = blockIdx.x * 256 + threadIdx.x, r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    gray[(size_t)r * src.cols + c] = canny_gray<PIX>(P::load(src.data, (size_t)r * src.stride + (size_t)c));
}

// four pixels per lane: one 16-byte load of Rgba(u8) (when the rows are 16-byte aligned), one float4 store
template <int PIX>
__global__ __launch_bounds__(256) void k_canny_gray4(DImg src, float *gray, uint8_t *gray8, int wide) { // cols % 4 == 0, gray 16-byte aligned; gray8: the same values as bytes; either may be null
    using P = Px<PIX>;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4, r = grid_row();
    if (c >= src.cols || r >= src.rows) return;
    typename P::Vec v[4];
    if constexpr (PIX == ZG_PIXEL_RGBA_U8) {
        if (wide) {
            const uint4 q = *(const uint4 *)((const uint8_t *)src.data + ((size_t)r * src.stride + (size_t)c) * 4);
            v[0] = __builtin_bit_cast(typename P::Vec, q.x); v[1] = __builtin_bit_cast(typename P::Vec, q.y);
            v[2] = __builtin_bit_cast(typename P::Vec, q.z); v[3] = __builtin_bit_cast(typename P::Vec, q.w);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = P::load(src.data, (size_t)r * src.stride + (size_t)(c + k));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = P::load(src.data, (size_t)r * src.stride + (size_t)(c + k));
    }
    const float4 g = make_float4(canny_gray<PIX>(v[0]), canny_gray<PIX>(v[1]), canny_gray<PIX>(v[2]), canny_gray<PIX>(v[3]));
    if (gray) *(float4 *)(gray + (size_t)r * src.cols + c) = g;
    if (gray8) *(uint32_t *)(gray8 + (size_t)r * src.cols + c) = (uint32_t)g.x | ((uint32_t)g.y << 8) | ((uint32_t)g.z << 16) | ((uint32_t)g.w << 24); // integers 0 .. 255
}
template <int PIX>
static void launch_canny_gray(const zg_image *src, float *gray, hipStream_t s, uint8_t *gray8 = nullptr) { // gray8 (and a null gray) only where cols % 4 == 0
    if (src->cols % 4 == 0 && ((uintptr_t)gray & 15) == 0) {
        const int wide = ((size_t)src->stride * 4) % 16 == 0 && ((uintptr_t)src->data & 15) == 0;
        hipLaunchKernelGGL((k_canny_gray4<PIX>), row_grid(ceil_div(src->cols, 1024), src->rows), dim3(256), 0, s, dimg(src), gray, gray8, wide);
    } else {
        hipLaunchKernelGGL((k_canny_gray<PIX>), row_grid(ceil_div(src->cols, 256), src->rows), dim3(256), 0, s, dimg(src), gray);
    }
}

// blurred plane -> state plane. Tile 64 x 16 outputs; magnitudes are needed one pixel around it, blurred values two.
// A thread owns four consecutive rows of one column: their 3 x 3 windows share six rows of three blurred values, and the Sobel sums
// skip the zero weights and fold the +-1 / +-2 weights into additions (exact products, so the reference's nine-term sums in the
// reference's order, edges.zig:255-262, up to the sign of a zero); only the magnitude goes back to LDS, plus the ring of magnitudes
// around the tile (164 positions, one each for the first threads). The round-1 form (64 x 4 tile, every gradient through a clamped
// nine-tap loop and three LDS planes) took 111 us per 4096^2 frame.
__global__ __launch_bounds__(256) void k_canny_nms(const float *blur, uint8_t *state, int rows, int cols, float low, float high, int tiles_x) {
    constexpr int TH = 16;
    __shared__ float b[TH + 4][68];   // blurred: tile row r, column c at [r + 2][c + 2]
    __shared__ float mag[TH + 2][68]; // magnitude: at [r + 1][c + 1]
    __shared__ uint8_t st[TH][64];
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    int wg = blockIdx.x;
    if (ZG_XCD_ORDER && wg < (per_xcd << 3)) wg = (wg & 7) * per_xcd + (wg >> 3);
    const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
    const int x0 = tx * 64, y0 = ty * TH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    { // .replicate of the 3 x 3 convolutions: clamped coordinates; a wave per row, the four columns beyond 64 by its first lanes
        const int gc = min(max(x0 - 2 + lane, 0), cols - 1), gc2 = min(max(x0 + 62 + lane, 0), cols - 1);
        for (int rr = w; rr < TH + 4; rr += 4) {
            const size_t row = (size_t)min(max(y0 - 2 + rr, 0), rows - 1) * cols;
            b[rr][lane] = blur[row + gc];
            if (lane < 4) b[rr][64 + lane] = blur[row + gc2];
        }
    }
    __syncthreads();
    float gx[4], gy[4], m[4];
    {
        float p[6][3];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) p[j][i] = b[4 * w + 1 + j][lane + 1 + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float win[3][3] = {{p[k][0], p[k][1], p[k][2]}, {p[k + 1][0], p[k + 1][1], p[k + 1][2]}, {p[k + 2][0], p[k + 2][1], p[k + 2][2]}};
            sobel_3x3(win, gx[k], gy[k]);
            const float sx = gx[k] * gx[k], sy = gy[k] * gy[k];
            m[k] = sqrtf(sx + sy);
            mag[4 * w + k + 1][lane + 1] = m[k];
        }
    }
    if (threadIdx.x < 2 * 66 + 2 * TH) { // the ring: row -1, row TH (columns -1 .. 64), column -1, column 64 (rows 0 .. TH - 1)
        const int t = threadIdx.x;
        int hr, hc;
        if (t < 66) { hr = -1; hc = t - 1; }
        else if (t < 132) { hr = TH; hc = t - 67; }
        else if (t < 132 + TH) { hr = t - 132; hc = -1; }
        else { hr = t - 132 - TH; hc = 64; }
        float win[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) win[j][i] = b[hr + 1 + j][hc + 1 + i];
        float hx, hy;
        sobel_3x3(win, hx, hy);
        const float sx = hx * hx, sy = hy * hy;
        mag[hr + 1][hc + 1] = sqrtf(sx + sy);
    }
    __syncthreads();
    const int c = x0 + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lr = 4 * w + k, r = y0 + lr;
        uint8_t s = 0;
        if (rows >= 3 && cols >= 3 && r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1) { // edges.zig:710-715
            const float K = 0.414213562f; // tan(22.5 deg)
            const float ax = fabsf(gx[k]), ay = fabsf(gy[k]);
            int dr1, dc1, dr2, dc2;
            if (ay <= K * ax) { dr1 = 0; dc1 = -1; dr2 = 0; dc2 = 1; }
            else if (ax <= K * ay) { dr1 = -1; dc1 = 0; dr2 = 1; dc2 = 0; }
            else if (gx[k] * gy[k] > 0) { dr1 = -1; dc1 = 1; dr2 = 1; dc2 = -1; }
            else { dr1 = -1; dc1 = -1; dr2 = 1; dc2 = 1; }
            const float n1 = mag[lr + 1 + dr1][lane + 1 + dc1], n2 = mag[lr + 1 + dr2][lane + 1 + dc2];
            if (m[k] >= n1 && m[k] >= n2) s = m[k] >= high ? 2 : (m[k] >= low ? 1 : 0); // edges.zig:541, :566
        }
        st[lr][lane] = s;
    }
    __syncthreads();
    { // the tile's state bytes, four columns per thread
        const int lr = threadIdx.x >> 4, q = (threadIdx.x & 15) * 4, r = y0 + lr;
        if (r < rows && x0 + q < cols) {
            uint8_t *o = state + (size_t)r * cols + x0 + q;
            if ((cols & 3) == 0 && ((uintptr_t)state & 3) == 0) {
                *(uint32_t *)o = *(const uint32_t *)&st[lr][q];
            } else {
                for (int j = 0; j < 4 && x0 + q + j < cols; ++j) o[j] = st[lr][q + j];
            }
        }
    }
}

// Hysteresis (edges.zig:499-576): a weak candidate becomes an edge iff it is 8-connected, through candidates, to a strong
// one. The reference grows the set with a BFS queue; the set itself is "the connected components of the candidate mask
// that contain a strong pixel", which a lock-free union-find labels in a fixed number of kernels (no convergence loop, no
// host synchronisation, so the detectors stay asynchronous and graph-capturable):
//   k_cc_tile    a workgroup labels one 64 x 64 tile entirely in LDS (runs from a ballot, unions with LDS atomics). A component that
//                does not reach the tile's edge is finished there and then: its pixels are written to the output (strong, or weak with
//                a strong pixel in the component) and leave the state plane. Only the components on a tile edge stay PENDING: their
//                pixels keep their state and get a label (the global index of the tile root), their roots a cleared flag and CC_ROOT (| CC_ROOT_STRONG) in their state byte.
//   k_cc_border  the pixels on tile edges are united with their neighbours in the next tile through global memory: 1/32 of
//                the pixels; roots only ever move to smaller indices (atomicMin), so the structure stays a forest whatever
//                the interleaving
//   k_cc_mark    the root of every pending tile component that holds a strong pixel marks the root of its global component
//   k_cc_emit_tile  a tile's roots look their global verdicts up; every pending weak pixel reads its own root's in LDS
// The label plane is only touched where components cross tiles, and the two last passes read a byte per pixel.
// (Round 1 united every pixel pair through global memory: 177 - 296 us of the detectors' time on 4096^2 noise; labelling tiles but
// still writing a label per pixel and resolving every weak pixel in a separate pass took 149 us on canny's frame.)
__device__ inline int cc_find(int *label, int x) {
    int p = label[x];
    while (p != x) { // path halving (plain stores: another lane can only have written a smaller ancestor)
        const int gp = label[p];
        if (gp != p) label[x] = gp;
        x = p;
        p = gp;
    }
    return x;
}
template <bool PAIRED>
__device__ inline void cc_unite_t(int *label, int a, int b) {
    for (;;) {
        if constexpr (PAIRED) { // cc_find of both at once: through global memory the two walks are independent chains of loads, and a union's
                                // time is their latency (k_cc_border 91 -> 74 us on noise; in LDS the extra instructions cost more than they hide)
            int pa = label[a], pb = label[b];
            while (pa != a || pb != b) {
                const int ga = label[pa], gb = label[pb];
                if (pa != a) { if (ga != pa) label[a] = ga; a = pa; pa = ga; }
                if (pb != b) { if (gb != pb) label[b] = gb; b = pb; pb = gb; }
            }
        } else {
            a = cc_find(label, a);
            b = cc_find(label, b);
        }
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; } // a > b: hang a under b
        const int old = atomicMin(&label[a], b);
        if (old == a) return; // a was still a root: linked
        a = old;              // someone re-rooted a in the meantime: retry from its new parent
    }
}
__device__ inline void cc_unite(int *label, int a, int b) { cc_unite_t<false>(label, a, b); }        // LDS
__device__ inline void cc_unite_global(int *label, int a, int b) { cc_unite_t<true>(label, a, b); }
// Links to the row below (SW / S / SE; the upward directions are the same pairs seen from the other side). Links the run
// structure already implies are skipped: with S a candidate, SW and SE hang off S's run, and S itself is implied when W and
// SW are both candidates (the pixel to the left makes the same link); without S, SW is implied by W and SE by E.
constexpr int CC_T = 64; // tile edge
// state bytes after k_cc_tile: 0 resolved / not a candidate, 1 weak and pending, 2 strong and pending; on the root pixel of a pending tile component also:
constexpr uint8_t CC_ROOT = 0x80, CC_ROOT_STRONG = 0x40;
__global__ __launch_bounds__(256) void k_cc_tile(uint8_t *state, int *label, uint16_t *label16, uint8_t *flag, DImg dst, int rows, int cols) {
    __shared__ int lab[CC_T * CC_T];
    __shared__ uint8_t st[CC_T + 1][CC_T]; // one spare row of zeros below
    __shared__ uint8_t out[CC_T][CC_T];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int x0 = blockIdx.x * CC_T, y0 = blockIdx.y * CC_T;
    const bool whole = x0 + CC_T <= cols && y0 + CC_T <= rows && (((uintptr_t)state | (uintptr_t)cols) & 3) == 0;
    if (whole) { // 16 rows of 64 bytes per instruction
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = k * 16 + (t >> 4), c = (t & 15) * 4;
            *(uint32_t *)&st[r][c] = *(const uint32_t *)(state + (size_t)(y0 + r) * cols + x0 + c);
        }
    } else {
        for (int i = t; i < CC_T * CC_T; i += 256) {
            const int r = i >> 6, c = i & 63;
            st[r][c] = (y0 + r < rows && x0 + c < cols) ? state[(size_t)(y0 + r) * cols + x0 + c] : 0;
        }
    }
    if (t < CC_T) st[CC_T][t] = 0;
    __syncthreads();
    // a candidate starts under the first pixel of its horizontal run (ballot + count-leading-zeros)
    uint32_t mine = 0; // bit k: my pixel of row w * 16 + k is a candidate
    unsigned long long rowm[17]; // the candidates of my sixteen rows and of the row below them, a bit a column: wave-uniform
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = w * 16 + k;
        const bool cand = st[r][lane] != 0;
        rowm[k] = __ballot(cand);
        const unsigned long long gaps = ~rowm[k] & ((1ull << lane) - 1); // non-candidates to my left
        const int start = gaps ? 64 - __clzll(gaps) : 0;
        lab[r * CC_T + lane] = r * CC_T + (cand ? start : lane);
        mine |= (uint32_t)cand << k;
    }
    rowm[16] = __ballot(st[w * 16 + 16][lane] != 0); // row 64 is zeros
    __syncthreads();
    // The links of a row are bit operations on two row masks (scalar): which pixels link to S, to SW, to SE under the skip rules above. Rows
    // without a link cost nothing, and no pixel reads its neighbours' bytes.
    const unsigned long long me = 1ull << lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long C = rowm[k], S = rowm[k + 1], W = C << 1, E = C >> 1, SW = S << 1, SE = S >> 1;
        const unsigned long long to_s = C & S & ~(W & SW), to_sw = C & ~S & SW & ~W, to_se = C & ~S & SE & ~E;
        if ((to_s | to_sw | to_se) == 0) continue; // wave-uniform
        const int i = (w * 16 + k) * CC_T + lane;
        // one pass of the union loop serves all three directions (a lane has S, or SW and / or SE); the few lanes with both diagonals go again
        const bool d_s = (to_s & me) != 0, d_sw = (to_sw & me) != 0, d_se = (to_se & me) != 0;
        if (d_s | d_sw | d_se) cc_unite(lab, i, i + CC_T + (d_s ? 0 : (d_sw ? -1 : 1)));
        if ((to_sw & to_se) != 0 && d_sw && d_se) cc_unite(lab, i, i + CC_T + 1);
    }
    __syncthreads();
    // every candidate straight under its root ...
    int root[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) root[k] = ((mine >> k) & 1) ? cc_find(lab, (w * 16 + k) * CC_T + lane) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if ((mine >> k) & 1) lab[(w * 16 + k) * CC_T + lane] = root[k];
    __syncthreads();
    // ... then what the component holds goes into the top bits of the root's entry (entries are < 4096): STRONG, EDGE (of the tile)
    constexpr int STRONG = 1 << 16, EDGE = 1 << 17;
    if (mine) {
        for (int k = 0; k < 16; ++k) {
            if (!((mine >> k) & 1)) continue;
            const int r = w * 16 + k;
            int a = st[r][lane] == 2 ? STRONG : 0;
            if (r == 0 || r == CC_T - 1 || lane == 0 || lane == CC_T - 1) a |= EDGE;
            if (a) atomicOr(&lab[root[k]], a);
        }
    }
    __syncthreads();
    const bool col_ok = x0 + lane < cols;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int r = w * 16 + k;
        uint8_t o = 0, ns = 0;
        if ((mine >> k) & 1) {
            const int a = lab[root[k]];
            const uint8_t s = st[r][lane];
            if (a & EDGE) { // pending: keeps its state, gets a label; a strong pixel is an edge whatever happens
                ns = s;
                o = s == 2 ? 255 : 0;
                if (col_ok && y0 + r < rows) {
                    const size_t gi = (size_t)(y0 + r) * cols + x0 + lane;
                    const int groot = (y0 + (root[k] >> 6)) * cols + x0 + (root[k] & 63);
                    // every pending pixel: its root's place IN THE TILE (two bytes; what k_cc_emit_tile reads). The global label — a node of the forest the
                    // border links build — only where the forest can touch it: the tile's edge pixels (a find starts there) and the root itself.
                    label16[gi] = (uint16_t)root[k];
                    const bool is_root = (int)gi == groot;
                    if (is_root || r == 0 || r == CC_T - 1 || lane == 0 || lane == CC_T - 1) label[gi] = groot;
                    if (is_root) { // the root pixel of a pending tile component says so in its state byte, and whether the component holds a strong pixel
                        ns = s | CC_ROOT | ((a & STRONG) ? CC_ROOT_STRONG : 0);
                        flag[gi] = 0;
                    }
                }
            } else {
                o = (s == 2 || (a & STRONG)) ? 255 : 0;
            }
        }
        st[r][lane] = ns; // my own pixel: nobody else reads it any more
        out[r][lane] = o;
    }
    __syncthreads();
    const bool vec = whole && (dst.stride & 3) == 0 && ((uintptr_t)dst.data & 3) == 0;
    if (vec) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = k * 16 + (t >> 4), c = (t & 15) * 4;
            *(uint32_t *)(state + (size_t)(y0 + r) * cols + x0 + c) = *(const uint32_t *)&st[r][c];
            *(uint32_t *)((uint8_t *)dst.data + (size_t)(y0 + r) * dst.stride + x0 + c) = *(const uint32_t *)&out[r][c];
        }
    } else {
        for (int i = t; i < CC_T * CC_T; i += 256) {
            const int r = i >> 6, c = i & 63;
            if (y0 + r < rows && x0 + c < cols) {
                state[(size_t)(y0 + r) * cols + x0 + c] = st[r][c];
                ((uint8_t *)dst.data)[(size_t)(y0 + r) * dst.stride + x0 + c] = out[r][c];
            }
        }
    }
}
// blockIdx.y < nvb: the right-hand column of a tile column (pixel with E / NE / SE in the next tile); otherwise the bottom row
// of a tile row (SW / S / SE below). The skip rules above hold across tiles for the same reasons: the link they rely on is
// either inside a tile or another border link. A candidate on a tile edge is pending by construction, so `state` still has it.
__global__ __launch_bounds__(256) void k_cc_border(const uint8_t *state, int *label, int rows, int cols, int nvb) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    int b = blockIdx.y;
    if (b < nvb) {
        const int x = b * CC_T + CC_T - 1, r = j; // x + 1 < cols by construction
        if (r >= rows) return;
        const int i = r * cols + x;
        if (!state[i]) return;
        if (state[i + 1]) { cc_unite_global(label, i, i + 1); return; } // NE and SE hang off E
        if (r > 0 && state[i - cols + 1] && !state[i - cols]) cc_unite_global(label, i, i - cols + 1);
        if (r + 1 < rows && state[i + cols + 1] && !state[i + cols]) cc_unite_global(label, i, i + cols + 1);
    } else {
        b -= nvb;
        const int y = b * CC_T + CC_T - 1, c = j; // y + 1 < rows by construction
        if (c >= cols) return;
        const int i = y * cols + c;
        if (!state[i]) return;
        const bool wc = c > 0 && state[i - 1], ec = c + 1 < cols && state[i + 1];
        const bool sw = c > 0 && state[i + cols - 1], so = state[i + cols], se = c + 1 < cols && state[i + cols + 1];
        if (so) {
            if (!(wc && sw)) cc_unite_global(label, i, i + cols);
        } else {
            if (sw && !wc) cc_unite_global(label, i, i + cols - 1);
            if (se && !ec) cc_unite_global(label, i, i + cols + 1);
        }
    }
}
// After the border links: every pending tile component that holds a strong pixel marks the root of its global component (flag[root] = 1; the flags of
// all tile roots were cleared by k_cc_tile). Four state bytes per lane; most dwords have no root.
__global__ __launch_bounds__(256) void k_cc_mark(const uint8_t *state, int *label, uint8_t *flag, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    uint32_t s4 = 0;
    if (i0 + 4 <= n && ((uintptr_t)state & 3) == 0) s4 = *(const uint32_t *)(state + i0);
    else for (size_t k = i0; k < n; ++k) s4 |= (uint32_t)state[k] << (8 * (k - i0));
    if ((s4 & (0x01010101u * CC_ROOT_STRONG)) == 0) return;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((s4 >> (8 * j)) & CC_ROOT_STRONG) flag[cc_find(label, (int)(i0 + j))] = 1;
}
// The weak pending pixels of a 64 x 64 tile become edges where their global component is marked. The roots of the tile's pending components look
// their global roots up once (that is where the tree walks are: a few dozen per tile instead of one per pixel) and leave the verdicts in LDS at
// the roots' places; a weak pixel reads the verdict at the place its two-byte label names. The pixels' labels are asked for before the roots walk,
// so the two round trips overlap. (One find per weak pixel through global memory: 56 us per 4096^2 frame of noise, 39 on a photo-like frame;
// with four-byte global labels decoded per pixel: 37 / 22.)
__global__ __launch_bounds__(256) void k_cc_emit_tile(const uint8_t *state, int *label, const uint16_t *label16, const uint8_t *flag, DImg dst, int rows, int cols) {
    __shared__ uint8_t verdict[CC_T][CC_T];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * CC_T, y0 = blockIdx.y * CC_T;
    const bool whole = x0 + CC_T <= cols && y0 + CC_T <= rows && (((uintptr_t)state | (uintptr_t)cols) & 3) == 0;
    const int c = (t & 15) * 4; // this lane's pixels: rows k * 16 + (t >> 4), columns c .. c + 4
    uint32_t st4[4];
    uint16_t l16[4][4];
    bool any_weak = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = k * 16 + (t >> 4);
        const size_t gi = (size_t)(y0 + r) * cols + x0 + c;
        st4[k] = 0;
        if (whole) st4[k] = *(const uint32_t *)(state + gi);
        else if (y0 + r < rows)
            for (int j = 0; j < 4 && x0 + c + j < cols; ++j) st4[k] |= (uint32_t)state[gi + j] << (8 * j);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = k * 16 + (t >> 4);
        const size_t gi = (size_t)(y0 + r) * cols + x0 + c;
        const uint32_t weak = (st4[k] & 0x01010101u) & ~((st4[k] >> 1) & 0x01010101u); // a byte is 1 exactly where the pixel is weak and pending
#pragma unroll
        for (int j = 0; j < 4; ++j) l16[k][j] = 0;
        if (weak) {
            any_weak = true;
            if (whole && ((uintptr_t)label16 & 7) == 0) { // four labels in one load (cols % 4 == 0)
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 v = *(const u32x2 *)(label16 + gi);
                const uint32_t v0 = v[0], v1 = v[1];
                l16[k][0] = (uint16_t)v0; l16[k][1] = (uint16_t)(v0 >> 16); l16[k][2] = (uint16_t)v1; l16[k][3] = (uint16_t)(v1 >> 16);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((weak >> (8 * j)) & 1u) l16[k][j] = label16[gi + j];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((st4[k] & (0x01010101u * CC_ROOT)) == 0) continue;
        const int r = k * 16 + (t >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((st4[k] >> (8 * j)) & CC_ROOT) verdict[r][c + j] = flag[cc_find(label, (y0 + r) * cols + x0 + c + j)];
    }
    if (!__syncthreads_or(any_weak)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t weak = (st4[k] & 0x01010101u) & ~((st4[k] >> 1) & 0x01010101u);
        if (!weak) continue;
        const int r = k * 16 + (t >> 4);
        uint8_t *drow = (uint8_t *)dst.data + (size_t)(y0 + r) * dst.stride + x0 + c;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((weak >> (8 * j)) & 1u) && (&verdict[0][0])[l16[k][j]] != 0) drow[j] = 255;
    }
}
// Writes the edge map of `state` (0 none / 1 weak / 2 strong) into dst; `state` is consumed. `work` holds an int label, a two-byte in-tile
// label and a flag byte per pixel (touched only where components cross tiles).
static int run_hysteresis(uint8_t *state, uint32_t rows, uint32_t cols, char *work, const zg_image *dst, hipStream_t s, const char *who) {
    const size_t n = (size_t)rows * cols;
    if (n > 0x7fffffffu) { set_error("%s: hysteresis labels are 32-bit (rows * cols must stay below 2^31)", who); return ZG_ERR_UNSUPPORTED; }
    int *label = (int *)work;
    uint16_t *label16 = (uint16_t *)(label + n);
    uint8_t *flag = (uint8_t *)(label16 + (n + 3) / 4 * 4);
    const unsigned nb = (unsigned)((n + 1023) / 1024);
    const unsigned nvb = (cols - 1) / CC_T, nhb = (rows - 1) / CC_T;
    const dim3 tiles(ceil_div(cols, (unsigned)CC_T), ceil_div(rows, (unsigned)CC_T));
    hipLaunchKernelGGL(k_cc_tile, tiles, dim3(256), 0, s, state, label, label16, flag, dimg(dst), (int)rows, (int)cols);
    if (nvb + nhb)
        hipLaunchKernelGGL(k_cc_border, dim3(ceil_div(rows > cols ? rows : cols, 256u), nvb + nhb), dim3(256), 0, s, (const uint8_t *)state, label, (int)rows, (int)cols, (int)nvb);
    hipLaunchKernelGGL(k_cc_mark, dim3(nb), dim3(256), 0, s, (const uint8_t *)state, label, flag, n);
    hipLaunchKernelGGL(k_cc_emit_tile, tiles, dim3(256), 0, s, (const uint8_t *)state, label, (const uint16_t *)label16, (const uint8_t *)flag, dimg(dst), (int)rows, (int)cols);
    if (hipGetLastError() != hipSuccess) { set_error("%s: hysteresis launch failed", who); return ZG_ERR_HIP; }
    return ZG_OK;
}
static size_t hysteresis_work_bytes(uint32_t rows, uint32_t cols) {
    const size_t n = (size_t)rows * cols;
    return n * sizeof(int) + (n + 3) / 4 * 4 * sizeof(uint16_t) + n + 64; // global labels | two-byte in-tile labels | flags
}

// out = 255 on edges, 0 elsewhere (edges.zig:511-515). A strong pixel is an edge; a weak one is an edge when hysteresis
// ran (label != nullptr) and its component's root is flagged. Four pixels per lane; VEC moves them as one dword.
template <bool VEC>
__global__ __launch_bounds__(256) void k_canny_emit(const uint8_t *state, int *label, const uint8_t *flag, DImg dst) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4, r = grid_row();
    if (c0 >= dst.cols || r >= dst.rows) return;
    const size_t i0 = (size_t)r * dst.cols + c0;
    uint8_t *out = (uint8_t *)dst.data + (size_t)r * dst.stride + c0;
    uint8_t st[4];
    const int nvalid = dst.cols - c0 < 4 ? dst.cols - c0 : 4;
    if (VEC) {
        *(uint32_t *)st = *(const uint32_t *)(state + i0);
    } else {
        for (int j = 0; j < 4; ++j) st[j] = j < nvalid ? state[i0 + j] : 0;
    }
    uint8_t px[4];
    for (int j = 0; j < 4; ++j) {
        bool edge = st[j] == 2;
        if (st[j] == 1 && label) edge = flag[cc_find(label, (int)(i0 + j))] != 0;
        px[j] = edge ? 255 : 0;
    }
    if (VEC) {
        *(uint32_t *)out = *(const uint32_t *)px;
    } else {
        for (int j = 0; j < nvalid; ++j) out[j] = px[j];
    }
}
static int launch_emit(const uint8_t *state, int *label, const uint8_t *flag, const zg_image *dst, hipStream_t s) {
    const dim3 grid = row_grid(ceil_div(dst->cols, 1024), dst->rows);
    const bool vec = dst->cols % 4 == 0 && dst->stride % 4 == 0 && (uintptr_t)dst->data % 4 == 0;
    if (vec) hipLaunchKernelGGL(k_canny_emit<true>, grid, dim3(256), 0, s, state, label, flag, dimg(dst));
    else hipLaunchKernelGGL(k_canny_emit<false>, grid, dim3(256), 0, s, state, label, flag, dimg(dst));
    return hipGetLastError() == hipSuccess ? ZG_OK : ZG_ERR_HIP;
}

static int canny_impl(const zg_image *src, const zg_image *dst, float sigma, float low, float high, zg_stream stream) {
    hipStream_t s = as_stream(stream);
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "canny: %ux%u vs %ux%u",
               src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(dst->pixel == ZG_PIXEL_U8, ZG_ERR_INVALID_ARGUMENT, "canny: the output is Image(u8)");
    ZG_REQUIRE(std::isfinite(sigma) && std::isfinite(low) && std::isfinite(high), ZG_ERR_INVALID_ARGUMENT, "canny: InvalidParameter (non-finite)");
    ZG_REQUIRE(sigma >= 0, ZG_ERR_INVALID_ARGUMENT, "canny: InvalidSigma (%g)", (double)sigma);
    ZG_REQUIRE(low >= 0 && high >= 0 && low < high, ZG_ERR_INVALID_ARGUMENT, "canny: InvalidThreshold (low %g, high %g)", (double)low, (double)high);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const uint32_t rows = src->rows, cols = src->cols;
    const size_t n = (size_t)rows * cols;

    // scratch: grey f32 | blurred f32 | state u8 | hysteresis work
    char *scratch = nullptr;
    const size_t state_off = 2 * n * sizeof(float), work_off = (state_off + n + 255) / 256 * 256;
    if ((rc = scratch_alloc((void **)&scratch, work_off + hysteresis_work_bytes(rows, cols), s))) return rc;
    float *gray = (float *)scratch, *blur = gray + n;
    uint8_t *state = (uint8_t *)(scratch + state_off);
    char *work = scratch + work_off;

    const float *blurred = gray;
    bool have_gray = false;
    auto make_gray = [&]() -> int {
        have_gray = true;
        return dispatch_pixel(src->pixel, [&](auto tag) -> int {
            constexpr int PIX = decltype(tag)::value;
            launch_canny_gray<PIX>(src, gray, s);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        });
    };
    if (sigma == 0) rc = make_gray();
    if (rc == ZG_OK && sigma != 0) { // blurGaussian (edges.zig:663-687): its own taps, .replicate
        const size_t radius = (size_t)std::ceil(3.0f * sigma), ks = 2 * radius + 1;
        // any length the separable convolution takes (kernels past 255 taps are read from device memory); the reference has no limit
        if (!(3.0f * sigma < 2000000.0f)) { scratch_free(scratch, s); ZG_REQUIRE(false, ZG_ERR_INVALID_ARGUMENT, "canny: sigma %g is out of range", (double)sigma); }
        std::vector<float> k(ks);
        float sum = 0;
        for (size_t i = 0; i < ks; ++i) {
            const float x = (float)i - (float)radius;
            k[i] = hostmath::exp_f32(-(x * x) / (2.0f * sigma * sigma));
            sum += k[i];
        }
        for (size_t i = 0; i < ks; ++i) k[i] /= sum;
        const zg_image gi{gray, cols, rows, cols, ZG_PIXEL_F32}, bi{blur, cols, rows, cols, ZG_PIXEL_F32};
        // Image(u8) / Image(Rgba(u8)) sources, up to 65 taps: the blur's row pass takes the grey straight from the source
        rc = ks <= 65 ? try_sep_f32long_grey(src, &bi, k.data(), (int)ks, k.data(), (int)ks, ZG_BORDER_REPLICATE, s) : -1;
        if (rc == -1) {
            rc = make_gray();
            if (rc == ZG_OK) rc = zg_conv_separable(&gi, &bi, k.data(), (uint32_t)ks, k.data(), (uint32_t)ks, ZG_BORDER_REPLICATE, stream);
        }
        blurred = blur;
    }
    if (rc == ZG_OK) {
        const int tiles_x = (int)ceil_div(cols, 64), tiles_y = (int)ceil_div(rows, 16);
        hipLaunchKernelGGL(k_canny_nms, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, s, blurred, state, (int)rows, (int)cols, low, high, tiles_x);
        rc = run_hysteresis(state, rows, cols, work, dst, s, "canny");
    }
    scratch_free(scratch, s);
    return rc;
}


// ---- Shen-Castan (src/image.zig:1015-1027 -> src/image/edges.zig:83-196) -----------------------------------------------
// grey -> ISEF smoothing (rows, then columns; each a forward and a backward first-order recursion, edges.zig:283-349) ->
// BLI = (smoothed - grey >= 0) -> zero crossings -> adaptive gradient from three integral images (grey, BLI, grey * BLI)
// -> percentile threshold from a 256-bin histogram -> optional NMS -> strong-only emit or hysteresis.
// The recursions and the integral images are sequential f32 chains by contract (the reference's rounding order), one
// chain per row / column: they run as one lane per chain with LDS transposes (rows) or coalesced strided walks (columns),
// latency-bound like boxBlur's SAT. The histogram, its percentile and the thresholds stay on the device (integer atomics
// and a one-workgroup kernel), so nothing but the hysteresis fixed-point test synchronises the stream.

int sat_planes_impl(const zg_image *src, float *sat, hipStream_t s, bool integer_valued, size_t plane_stride = 0); // box_blur.hip (0: planes contiguous)
int isef_2d(const void *gray, bool gray_is_bytes, float *sm, float *tmp, uint32_t *check, uint32_t rows, uint32_t cols, float smooth, hipStream_t s); // isef.hip
bool isef_2d_applies(uint32_t rows, uint32_t cols);
size_t isef_check_bytes(uint32_t rows, uint32_t cols);
int sat_planes_multi(const zg_image *const *srcs, float *const *sats, int count, hipStream_t s); // box_blur.hip

// The recursions along ROWS run as the column kernel on the transposed plane: a row chain needs lanes = rows, i.e. a transpose
// through LDS per 64-column chunk inside a kernel with one wave per 64 rows (354 + 464 us per 4096^2 plane that way); two plain
// transposes at memory speed around the role-split column kernel do the same arithmetic, element for element, in a third of it.
__global__ __launch_bounds__(256) void k_transpose_f32(const float *in, float *out, int rows, int cols) { // out[c][r] = in[r][c]
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ly + 4 * k, c = c0 + lx;
        tile[ly + 4 * k][lx] = in[(size_t)min(r, rows - 1) * cols + min(c, cols - 1)]; // clamped, unpredicated
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ly + 4 * k, r = r0 + lx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[lx][ly + 4 * k];
    }
}
// The same two recursions down / up the columns, in place on `data` with `temp` between them. One chain per column and two
// dependent operations per row (a * run, then + b * x): a wave that also does its own loads and stores is latency-bound (one
// wave per 64 columns: 379 us per 4096^2 plane). Roles are split as in box_blur.hip's k_sat_chain: a workgroup owns 64 columns;
// waves 1..8 are LOADERS that keep six 64-row blocks in flight and hand rows over through an LDS ring, wave 0 is the CHAIN:
// per row one LDS read, the recurrence, one store, nothing in its memory queue but stores. The up pass reads what the down
// pass wrote (same workgroup, other waves): the chain wave publishes with a device-scope fence before the barrier that
// separates the passes.
__global__ __launch_bounds__(576) void k_isef_cols(float *data, float *temp, int rows, int cols, float b) {
    constexpr int SB = 64, NL = 8, RL = SB / NL, D = 6;
    __shared__ float ring[2][SB][64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < cols;
    const int cc = min(c, cols - 1);
    const int nblocks = (rows + SB - 1) / SB;
    const float a = 1.0f - b;
    const bool all_live = blockIdx.x * 64 + 64 <= (unsigned)cols; // workgroup-uniform

    for (int pass = 0; pass < 2; ++pass) { // 0: down, data -> temp; 1: up, temp -> data. Block k of the up pass is block nblocks - 1 - k.
        const float *in = pass == 0 ? data : temp;
        float *out = pass == 0 ? temp : data;
        if (wave == 0) {
            float run = 0.0f;
            for (int k = 0; k < nblocks; ++k) {
                __syncthreads(); // block k of this pass is in ring[k & 1]
                const int blk = pass == 0 ? k : nblocks - 1 - k, r0 = blk * SB;
                float p[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) p[i] = ring[k & 1][i][lane];
                float *o = out + (size_t)r0 * cols + cc;
                const bool plain = all_live && r0 + SB <= rows; // every row of the block exists, every lane stores: no predicate
                if (pass == 0 && plain && r0 > 0) {
#pragma unroll
                    for (int i = 0; i < SB; ++i) {
                        const float bx = b * p[i], ar = a * run;
                        run = bx + ar;
                        o[(size_t)i * cols] = run;
                    }
                } else if (pass == 1 && plain && r0 + SB < rows) {
#pragma unroll
                    for (int i = SB - 1; i >= 0; --i) {
                        const float bt = b * p[i], ar = a * run;
                        run = bt + ar;
                        o[(size_t)i * cols] = run;
                    }
                } else if (pass == 0) {
#pragma unroll
                    for (int i = 0; i < SB; ++i) {
                        const float bx = b * p[i];
                        if (r0 + i == 0) run = bx;
                        else { const float ar = a * run; run = bx + ar; }
                        if (live && r0 + i < rows) o[(size_t)i * cols] = run;
                    }
                } else {
#pragma unroll
                    for (int i = SB - 1; i >= 0; --i) {
                        if (r0 + i < rows) { // rows past the end do not exist: the chain starts at rows - 1
                            if (r0 + i == rows - 1) run = p[i];
                            else { const float bt = b * p[i], ar = a * run; run = bt + ar; }
                            if (live) o[(size_t)i * cols] = run;
                        }
                    }
                }
            }
            if (pass == 0) __threadfence(); // temp is complete and visible before any loader of this workgroup reads it back
        } else {
            const int sub = wave - 1;
            struct Regs { float v[RL]; };
            auto fetch = [&](int k, Regs &g) { // clamped, unpredicated; steps past the last block re-read a valid row and are never used
                const int blk = pass == 0 ? k : nblocks - 1 - k;
                const int r0 = min(max(blk, 0), nblocks - 1) * SB + sub * RL;
#pragma unroll
                for (int i = 0; i < RL; ++i) g.v[i] = in[(size_t)min(r0 + i, rows - 1) * cols + cc];
            };
            auto publish = [&](int k, const Regs &g) {
#pragma unroll
                for (int i = 0; i < RL; ++i) ring[k & 1][sub * RL + i][lane] = g.v[i];
            };
            Regs g[D];
#pragma unroll
            for (int d = 0; d < D - 1; ++d) fetch(d, g[d]);
            for (int k0 = 0; k0 < nblocks; k0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int k = k0 + d;
                    if (k < nblocks) {
                        fetch(k + D - 1, g[(d + D - 1) % D]);
                        publish(k, g[d]);
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads(); // the pass is over for every wave (and the ring is free again)
    }
}

// BLI, grey * BLI and the zero-crossing candidates (edges.zig:111-128, 356-415). FORWARD: east / south / south-east /
// south-west neighbours; otherwise (NMS mode) any 4-neighbour, interior pixels only when the image is at least 3 x 3.
__global__ __launch_bounds__(256) void k_sc_bli(const float *gray, const float *sm, uint8_t *bli, float *gm, uint8_t *cand, int rows, int cols, int forward) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= cols || r >= rows) return;
    auto B = [&](int rr, int cc) -> int { const size_t i = (size_t)rr * cols + cc; return (sm[i] - gray[i]) >= 0 ? 1 : 0; };
    const int ce = B(r, c);
    const size_t idx = (size_t)r * cols + c;
    bli[idx] = (uint8_t)ce;
    gm[idx] = gray[idx] * (float)ce;
    bool mark = false;
    if (forward) {
        if (!mark && c + 1 < cols) mark = ce != B(r, c + 1);
        if (!mark && r + 1 < rows) mark = ce != B(r + 1, c);
        if (!mark && r + 1 < rows && c + 1 < cols) mark = ce != B(r + 1, c + 1);
        if (!mark && r + 1 < rows && c > 0) mark = ce != B(r + 1, c - 1);
    } else if (rows >= 3 && cols >= 3) {
        if (r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1) mark = ce != B(r, c - 1) || ce != B(r, c + 1) || ce != B(r - 1, c) || ce != B(r + 1, c);
    } else {
        if (!mark && c > 0) mark = ce != B(r, c - 1);
        if (!mark && c + 1 < cols) mark = ce != B(r, c + 1);
        if (!mark && r > 0) mark = ce != B(r - 1, c);
        if (!mark && r + 1 < rows) mark = ce != B(r + 1, c);
    }
    cand[idx] = mark ? 255 : 0;
}

constexpr int SC_HIST_COPIES = 64;
constexpr int SC_BLI_ROWS = 16; // rows a wave of k_sc_bli4 walks (+ one above and two below): 8 -> 31.8 us, 16 -> 31.2, 32 -> 49.3 (unrolled: code and registers)
// The same, four pixels per lane (cols % 4 == 0): a lane loads float4s, keeps the BLI of its columns and of the two beside
// them as six bits per row (the neighbours' come over with DPP wave shifts, a workgroup's outer columns with one extra load),
// and the marks of four pixels are a few bitwise operations on the rows above, at and below; dword and float4 stores.
// A wave walks 16 rows, two rows of loads ahead. MODE 0: forward; 1: four-neighbour, interior only; 2: four-neighbour, bounded.
// (The lane-per-pixel kernel evaluates BLI five times per pixel, ten loads: 97 us per 4096^2 frame against 45 us.)
template <int MODE>
__global__ __launch_bounds__(256) void k_sc_bli4(const uint8_t *gray, const float *sm, uint8_t *bli, uint8_t *gm, uint8_t *cand, int rows, int cols, unsigned int *hist_to_clear) { // gray, gm: bytes
    if (blockIdx.x == 0 && blockIdx.y == 0) // the gradient kernel's histogram copies start from zero: cleared here, two launches earlier, instead of by a memset of its own
        for (int i = threadIdx.x; i < SC_HIST_COPIES * 256; i += 256) hist_to_clear[i] = 0;
    constexpr int RW = SC_BLI_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 256 + lane * 4;
    const int y0 = (blockIdx.y * 4 + wave) * RW;
    if (y0 >= rows) return; // wave-uniform
    const bool live = x0 < cols;
    const int xc = live ? x0 : cols - 4;
    const bool is_edge = live && ((lane == 0 && x0 > 0) || (lane == 63 && x0 + 4 < cols));
    const int ecol = lane == 0 ? x0 - 1 : x0 + 4;
    uint32_t mE = 0, mW = 0, mI = 0; // bit j: column x0 + j has an east neighbour / a west neighbour / is interior
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (x0 + j + 1 < cols) mE |= 1u << j;
        if (x0 + j > 0) mW |= 1u << j;
        if (x0 + j >= 1 && x0 + j < cols - 1) mI |= 1u << j;
    }
    struct Raw { float4 s, g; float es, eg; uint32_t g8; };
    auto load = [&](int r, Raw &w) { // row clamped into the image: rows outside are masked where they are used
        r = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r);
        const size_t i = (size_t)r * cols;
        w.s = *(const float4 *)(sm + i + xc);
        const uint32_t g4 = *(const uint32_t *)(gray + i + xc);
        w.g8 = g4;
        w.g = make_float4((float)(g4 & 255u), (float)((g4 >> 8) & 255u), (float)((g4 >> 16) & 255u), (float)(g4 >> 24));
        w.es = 0.0f; w.eg = 0.0f;
        if (is_edge) { w.es = sm[i + ecol]; w.eg = (float)gray[i + ecol]; }
    };
    auto ext_of = [&](const Raw &w) -> uint32_t { // bit k: BLI of column x0 - 1 + k
        const uint32_t nib = ((w.s.x - w.g.x) >= 0 ? 1u : 0u) | ((w.s.y - w.g.y) >= 0 ? 2u : 0u) | ((w.s.z - w.g.z) >= 0 ? 4u : 0u) | ((w.s.w - w.g.w) >= 0 ? 8u : 0u);
        const uint32_t e = (w.es - w.eg) >= 0 ? 1u : 0u;
        uint32_t left = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0x138, 0xf, 0xf, true) >> 3) & 1u; // wave_shr:1: lane - 1
        uint32_t right = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0x130, 0xf, 0xf, true) & 1u;      // wave_shl:1: lane + 1
        if (lane == 0) left = e;
        if (lane == 63) right = e;
        return left | (nib << 1) | (right << 5);
    };
    Raw q0, q1;
    load(y0 - 1, q0);
    load(y0, q1);
    uint32_t ext_n = ext_of(q0), ext_c = ext_of(q1);
    uint32_t g_c = q1.g8;
    load(y0 + 1, q0);
    load(y0 + 2, q1);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int r = y0 + k;
        if (r >= rows) break; // wave-uniform
        Raw &cur = (k & 1) ? q1 : q0; // row r + 1
        const uint32_t ext_s = ext_of(cur);
        const uint32_t g_s = cur.g8;
        load(r + 3, cur);
        const uint32_t C = (ext_c >> 1) & 15u, Eb = (ext_c >> 2) & 15u, Wb = ext_c & 15u;
        const uint32_t Sb = (ext_s >> 1) & 15u, Nb = (ext_n >> 1) & 15u;
        const uint32_t vS = r + 1 < rows ? 15u : 0u, vN = r > 0 ? 15u : 0u;
        uint32_t mark;
        if (MODE == 0) {
            const uint32_t SEb = (ext_s >> 2) & 15u, SWb = ext_s & 15u;
            mark = ((C ^ Eb) & mE) | (((C ^ Sb) | ((C ^ SEb) & mE) | ((C ^ SWb) & mW)) & vS);
        } else if (MODE == 1) {
            mark = ((C ^ Wb) | (C ^ Eb) | (C ^ Nb) | (C ^ Sb)) & mI & (vS & vN);
        } else {
            mark = ((C ^ Wb) & mW) | ((C ^ Eb) & mE) | ((C ^ Nb) & vN) | ((C ^ Sb) & vS);
        }
        if (live) {
            const size_t i = (size_t)r * cols + x0;
            const uint32_t cb = (C * 0x00204081u) & 0x01010101u; // bit j -> byte j
            *(uint32_t *)(bli + i) = cb;
            *(uint32_t *)(cand + i) = ((mark * 0x00204081u) & 0x01010101u) * 255u;
            *(uint32_t *)(gm + i) = g_c & (cb * 255u); // grey * BLI
        }
        ext_n = ext_c;
        ext_c = ext_s;
        g_c = g_s;
    }
}

// adaptive gradient at the candidates (edges.zig:462-496) + the histogram of its rounded values (:139-150)
// BUF: the planes are below 4 GiB, so a corner read is a buffer load (scalar row offset + per-lane column offset: no vector
// address arithmetic at all; with 64-bit pointers a third of the kernel's instructions computed addresses).
// CNT: sat_m is not an integral image but the window's count itself, one byte per pixel (k_sc_count below): one byte load per pixel instead of four
// corner loads, and one integral image fewer to build.
template <bool BUF, bool CNT = false>
__global__ __launch_bounds__(256) void k_sc_gradient(const uint8_t *cand, const float *sat_g, const float *sat_m, const float *sat_gm, float *grad,
                                                     unsigned int *hist, int rows, int cols, int hw) {
    // sixteen copies of the block histogram: neighbouring pixels have similar gradients, and 64 lanes hitting one LDS counter
    // serialise (measured 1086 us per 4096^2 frame of noise with a single copy)
    __shared__ unsigned int lh[16][256];
#pragma unroll
    for (int k = 0; k < 16; ++k) lh[k][threadIdx.x] = 0;
    __syncthreads();
    // a workgroup covers 64 columns x 64 rows (sixteen steps of four rows), so the 256 global histogram updates it ends
    // with are amortised over 4096 pixels (one update per 4-row block serialised 16.7 M atomics on 256 counters: 1.09 ms).
    // The kernel is latency-bound (one step at a time, the corner reads behind two branches: 150 us per 4096^2 frame), so the
    // sixteen candidate bytes of a thread are read first and the steps go four at a time, the 48 corner reads of a group
    // issued before anything is computed. The loads of a wave row are whole cache lines whichever lanes want them, so they are not
    // predicated on the candidate bit, only skipped when no lane of the wave has a candidate in the group.
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), cc = min(c, cols - 1);
    const int wrow = blockIdx.y * 64 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // a wave is one row: row arithmetic is scalar
    const int c1 = cc > hw ? cc - hw : 0, c2 = min(cc + hw, cols - 1);
    const int cl = c1 > 0 ? c1 - 1 : 0; // column of the two left-hand corners (read anyway, dropped when c1 == 0)
    const float *const planes[3] = {sat_m, sat_gm, sat_g};
    const uint32_t oc2 = (uint32_t)c2 * 4u, ocl = (uint32_t)cl * 4u;
    const uint32_t plane_bytes = BUF ? (uint32_t)((size_t)rows * cols * 4) : 0u;
    uint8_t cb[16];
    [[maybe_unused]] uint8_t cnt[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        cb[st] = cand[(size_t)min(wrow + st * 4, rows - 1) * cols + cc];
        if constexpr (CNT) cnt[st] = ((const uint8_t *)sat_m)[(size_t)min(wrow + st * 4, rows - 1) * cols + cc];
    }
    // out-of-range buffer offsets read as zero: the corners that do not exist (c1 == 0, r1 == 0) need no select afterwards
    constexpr uint32_t OOR = 0xfffffff0u;
    const uint32_t ocl_z = c1 > 0 ? ocl : OOR;
    constexpr int U = CNT ? 4 : 2, NG = 16 / U; // steps per group (two groups in flight): the counted form has the registers for four
    struct Group { float v[U][12]; };
    auto issue = [&](int grp, Group &g) { // the 24 corner reads of steps grp * U .. + U
        bool some = false;
#pragma unroll
        for (int u = 0; u < U; ++u) some = some || (c < cols && wrow + (grp * U + u) * 4 < rows && cb[grp * U + u] != 0);
        if (__ballot(some) != 0) { // wave-uniform
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = min(wrow + (grp * U + u) * 4, rows - 1);
                const int r1 = r > hw ? r - hw : 0, r2 = min(r + hw, rows - 1);
                const size_t bot = (size_t)r2 * cols, top = (size_t)(r1 > 0 ? r1 - 1 : 0) * cols;
#pragma unroll
                for (int p = CNT ? 1 : 0; p < 3; ++p) {
                    if constexpr (BUF) {
                        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)planes[p], (short)0, (int)plane_bytes, 0x00020000);
                        // readfirstlane: the row offsets are wave-uniform, and saying so spares a waterfall loop around every load
                        const int sb = __builtin_amdgcn_readfirstlane((int)(uint32_t)(bot * 4)), stp = __builtin_amdgcn_readfirstlane((int)(uint32_t)(top * 4));
                        const uint32_t t2 = r1 > 0 ? oc2 : OOR, tl = r1 > 0 ? ocl_z : OOR;
                        g.v[u][p * 4 + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)oc2, sb, 0));
                        g.v[u][p * 4 + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)ocl_z, sb, 0));
                        g.v[u][p * 4 + 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)t2, stp, 0));
                        g.v[u][p * 4 + 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)tl, stp, 0));
                    } else {
                        g.v[u][p * 4 + 0] = planes[p][bot + c2];
                        g.v[u][p * 4 + 1] = c1 > 0 ? planes[p][bot + cl] : 0.0f;
                        g.v[u][p * 4 + 2] = r1 > 0 ? planes[p][top + c2] : 0.0f;
                        g.v[u][p * 4 + 3] = (r1 > 0 && c1 > 0) ? planes[p][top + cl] : 0.0f;
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 12; ++k) g.v[u][k] = 0.0f;
        }
    };
    auto finish = [&](int grp, const Group &g) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = wrow + (grp * U + u) * 4;
            if (c >= cols || r >= rows) continue;
            float gr = 0.0f;
            if (cb[grp * U + u] != 0) {
                const int r1 = r > hw ? r - hw : 0, r2 = min(r + hw, rows - 1);
                const float area = (float)((size_t)(r2 - r1 + 1) * (size_t)(c2 - c1 + 1));
                float sum[3];
#pragma unroll
                for (int p = CNT ? 1 : 0; p < 3; ++p) sum[p] = ((g.v[u][p * 4 + 0] - g.v[u][p * 4 + 1]) - g.v[u][p * 4 + 2]) + g.v[u][p * 4 + 3]; // integral.zig:85-90, in that order
                if constexpr (CNT) sum[0] = (float)cnt[grp * U + u];
                const float count1 = sum[0], count0 = area - count1;
                if (count0 > 0 && count1 > 0) {
                    const float sum1 = sum[1], sum_total = sum[2];
                    const float sum0 = sum_total - sum1;
                    const float mean0 = sum0 / count0, mean1 = sum1 / count1;
                    gr = fabsf(mean1 - mean0);
                }
                float hgv = gr;
                if (hgv < 0) hgv = 0;
                if (hgv > 255) hgv = 255;
                atomicAdd(&lh[threadIdx.x & 15][(int)roundf(hgv)], 1u);
            }
            grad[(size_t)r * cols + c] = gr;
        }
    };
    Group ga, gb; // two groups in flight: the reads of the next one are out before this one is computed
    issue(0, ga);
#pragma unroll
    for (int grp = 0; grp < NG; grp += 2) {
        issue(grp + 1, gb);
        finish(grp, ga);
        if (grp + 2 < NG) issue(grp + 2, ga);
        finish(grp + 1, gb);
    }
    __syncthreads();
    unsigned int total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += lh[k][threadIdx.x];
    // SC_HIST_COPIES copies of the frame histogram, picked by workgroup: 4096 workgroups adding into one set of 256 counters queued
    // up at the L2 atomic units (about half of this kernel's time on a 4096^2 frame); k_sc_thresholds sums the copies
    if (total) atomicAdd(&hist[((blockIdx.y * gridDim.x + blockIdx.x) % SC_HIST_COPIES) * 256 + threadIdx.x], total);
}
// The number of BLI pixels in each pixel's clipped (2 hw + 1)^2 window, as a byte (hw <= 7: at most 225). The reference takes it from the integral
// image of the mask, ((A - B) - C) + D (integral.zig:85-90): on frames of at most 2^24 pixels every value of that image is an integer f32 holds
// exactly, so the four-corner expression IS the count — computed here directly (two 1-D sums through LDS), which spares the detector one of its
// three integral images (a third of k_sat_chain_planes' stores) and k_sc_gradient four of its twelve corner loads per pixel.
__global__ __launch_bounds__(256) void k_sc_count(const uint8_t *bli, uint8_t *cnt, int rows, int cols, int hw) {
    constexpr int TW = 64, TH = 32, HMAX = 7;
    __shared__ uint8_t in[TH + 2 * HMAX][TW + 2 * HMAX + 2];
    __shared__ uint8_t hs[TH + 2 * HMAX][TW];
    const int t = threadIdx.x, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int nr = TH + 2 * hw, nc = TW + 2 * hw;
    for (int i = t; i < nr * nc; i += 256) {
        const int rr = i / nc, cc = i - rr * nc, gy = y0 - hw + rr, gx = x0 - hw + cc;
        in[rr][cc] = (gy >= 0 && gy < rows && gx >= 0 && gx < cols) ? bli[(size_t)gy * cols + gx] : 0; // outside the frame: not in any clipped window
    }
    __syncthreads();
    for (int i = t; i < nr * TW; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        int sum = 0;
        for (int k = 0; k <= 2 * hw; ++k) sum += in[rr][cc + k];
        hs[rr][cc] = (uint8_t)sum;
    }
    __syncthreads();
    for (int i = t; i < TH * TW; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        int sum = 0;
        for (int k = 0; k <= 2 * hw; ++k) sum += hs[rr + k][cc];
        if (y0 + rr < rows && x0 + cc < cols) cnt[(size_t)(y0 + rr) * cols + x0 + cc] = (uint8_t)sum;
    }
}

// The same for windows up to 7 x 7 (cols % 4 == 0), streaming: a lane owns four adjacent columns as the bytes of a dword and walks down a 32-row
// segment. The horizontal sums of its four pixels are 2 HW + 1 byte-shifted copies of { left neighbour's dword, its own, right neighbour's } added
// as packed bytes (sums <= 7: no carry between bytes; the neighbours' dwords come over with DPP wave shifts, a wave's outer lanes fetch theirs);
// the vertical sum is a running packed sum over a ring of 2 HW + 1 rows (<= 49 per byte). One dword in, one dword out per four pixels.
constexpr int SC_COUNT_SEG = 32; // rows a wave walks (+ 2 HW of run-in)
template <int HW>
__global__ __launch_bounds__(256) void k_sc_count_stream(const uint8_t *bli, uint8_t *cnt, int rows, int cols) {
    constexpr int N = 2 * HW + 1, SEG = SC_COUNT_SEG;
    const int lane = threadIdx.x & 63, strip = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int x0 = strip * 256 + lane * 4;
    if (strip * 256 >= cols) return; // wave-uniform
    const bool live = x0 < cols;
    const int y0 = (int)blockIdx.y * SEG, y1 = min(y0 + SEG, rows);
    const int ex = lane == 0 ? x0 - 4 : x0 + 4; // the dword beside the wave's columns that this lane fetches if it is an outer lane
    const bool outer = (lane == 0 || lane == 63) && ex >= 0 && ex < cols;
    uint32_t ring[N];
#pragma unroll
    for (int k = 0; k < N; ++k) ring[k] = 0;
    uint32_t v = 0;
    for (int rb = y0 - HW; rb < y1 + HW; rb += N) {
        uint32_t curs[N], sides[N]; // a ring's worth of rows asked for together: the walk is one memory round trip per N rows, not per row
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int r = rb + j;
            curs[j] = sides[j] = 0;
            if (r >= 0 && r < rows && r < y1 + HW) {
                if (live) curs[j] = *(const uint32_t *)(bli + (size_t)r * cols + x0);
                if (outer) sides[j] = *(const uint32_t *)(bli + (size_t)r * cols + ex);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) { // row rb + j goes into ring slot j: (rb - (y0 - HW)) is a multiple of N
            const int r = rb + j;
            const uint32_t cur = curs[j];
            uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x138, 0xf, 0xf, true); // wave_shr:1: lane - 1's dword
            uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x130, 0xf, 0xf, true); // wave_shl:1: lane + 1's
            if (lane == 0) prev = sides[j];
            if (lane == 63) next = sides[j];
            uint32_t h = cur;
#pragma unroll
            for (int k = 1; k <= HW; ++k) h += __builtin_amdgcn_alignbyte(next, cur, k) + __builtin_amdgcn_alignbyte(cur, prev, 4 - k);
            v += h - ring[j]; // the row 2 HW + 1 above leaves the window, per byte: it was in the sum, so no borrow
            ring[j] = h;
            const int ro = r - HW; // the row whose window is now complete
            if (ro >= y0 && ro < y1 && live) *(uint32_t *)(cnt + (size_t)ro * cols + x0) = v;
        }
    }
}

// thr[0] = t_high, thr[1] = t_low (edges.zig:160-166): the reference walks the histogram until the running count reaches
// floor(total * high_ratio); the number of steps it takes is the number of bins whose EXCLUSIVE prefix is below that target.
// One workgroup of 256 threads, a bin each (a single thread walking 256 global loads took 12 us).
__global__ __launch_bounds__(256) void k_sc_thresholds(const unsigned int *hist, float *thr, float high_ratio, float low_rel) {
    __shared__ unsigned int wtot[4], below[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned int h = 0;
#pragma unroll 8
    for (int k = 0; k < SC_HIST_COPIES; ++k) h += hist[k * 256 + t];
    unsigned int y = h; // inclusive scan over the wave (counts sum to the pixel count, below 2^31)
    y += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)y, 0x111, 0xf, 0xf, true);
    y += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)y, 0x112, 0xf, 0xf, true);
    y += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)y, 0x114, 0xf, 0xf, true);
    y += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)y, 0x118, 0xf, 0xf, true);
    y += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)y, 0x142, 0xa, 0xf, false);
    y += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)y, 0x143, 0xc, 0xf, false);
    if (lane == 63) wtot[w] = y;
    __syncthreads();
    unsigned int before = 0;
    for (int k = 0; k < w; ++k) before += wtot[k];
    const unsigned long long total = (unsigned long long)wtot[0] + wtot[1] + wtot[2] + wtot[3];
    const unsigned long long target = (unsigned long long)floorf((float)total * high_ratio);
    const unsigned long long excl = (unsigned long long)before + (y - h);
    const unsigned long long m = __ballot(excl < target);
    if (lane == 0) below[w] = (unsigned int)__popcll(m);
    __syncthreads();
    if (t == 0) {
        const int idx = (int)(below[0] + below[1] + below[2] + below[3]);
        const float t_high = (float)min(idx, 255);
        thr[0] = t_high;
        thr[1] = low_rel * t_high;
    }
}
// NMS on central differences of the smoothed plane (edges.zig:582-661)
__global__ __launch_bounds__(256) void k_sc_nms(const float *sm, const float *grad, const uint8_t *cand, uint8_t *out, int rows, int cols) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= cols || r >= rows) return;
    const size_t idx = (size_t)r * cols + c;
    uint8_t keep = 0;
    if (rows >= 3 && cols >= 3 && r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1 && cand[idx] != 0) {
        const float K = 0.414213562f;
        const float gx = 0.5f * (sm[idx + 1] - sm[idx - 1]), gy = 0.5f * (sm[idx + cols] - sm[idx - cols]);
        const float ax = fabsf(gx), ay = fabsf(gy);
        int dr1, dc1, dr2, dc2;
        if (ay <= K * ax) { dr1 = 0; dc1 = -1; dr2 = 0; dc2 = 1; }
        else if (ax <= K * ay) { dr1 = -1; dc1 = 0; dr2 = 1; dc2 = 0; }
        else if (gx * gy > 0) { dr1 = -1; dc1 = 1; dr2 = 1; dc2 = -1; }
        else { dr1 = -1; dc1 = -1; dr2 = 1; dc2 = 1; }
        const float m = grad[idx], n1 = grad[(size_t)(r + dr1) * cols + (c + dc1)], n2 = grad[(size_t)(r + dr2) * cols + (c + dc2)];
        if (m >= n1 && m >= n2) keep = 255;
    }
    out[idx] = keep;
}
// 0 none / 1 weak / 2 strong against the device-resident thresholds; without hysteresis only strong survives. Flat planes,
// four pixels per lane (dword / float4 accesses; the planes start 16 bytes aligned).
__global__ __launch_bounds__(256) void k_sc_classify4(const uint8_t *cand, const float *grad, const float *thr, uint8_t *state, size_t n, int hysteresis) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float hi = thr[0], lo = thr[1];
    if (i + 4 <= n) {
        const uint32_t c = *(const uint32_t *)(cand + i);
        const float4 g = *(const float4 *)(grad + i);
        const float e[4] = {g.x, g.y, g.z, g.w};
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t st = 0;
            if ((c >> (8 * j)) & 0xffu) st = e[j] >= hi ? 2u : ((hysteresis && e[j] >= lo) ? 1u : 0u);
            out |= st << (8 * j);
        }
        *(uint32_t *)(state + i) = out;
    } else {
        for (size_t k = i; k < n; ++k) {
            uint8_t st = 0;
            if (cand[k] != 0) st = grad[k] >= hi ? 2 : ((hysteresis && grad[k] >= lo) ? 1 : 0);
            state[k] = st;
        }
    }
}

// isefFilter2D (edges.zig:308-349): gray -> sm. The segmented recursions of isef.hip where its preconditions hold; planes whose rows are
// not whole 16-byte chunks take round 3's route: transpose, column recursions on the cols x rows plane (in place on `t1`, `t2` between the
// passes), transpose back into `sm`, then the columns proper. tmp, t1, t2: planes of the same size; check: isef_check_bytes().
// gray8: the same plane as bytes, or null; gray (f32) may be null when gray8 is given and isef_2d_applies().
static void isef_plane(const float *gray, const uint8_t *gray8, float *sm, float *tmp, float *t1, float *t2, uint32_t *check, uint32_t rows, uint32_t cols, float smooth,
                       hipStream_t s) {
    if (gray8 && isef_2d((const void *)gray8, true, sm, tmp, check, rows, cols, smooth, s) >= 0) return;
    if (isef_2d((const void *)gray, false, sm, tmp, check, rows, cols, smooth, s) >= 0) return;
    hipLaunchKernelGGL(k_transpose_f32, dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, s, gray, t1, (int)rows, (int)cols);
    hipLaunchKernelGGL(k_isef_cols, dim3(ceil_div(rows, 64)), dim3(576), 0, s, t1, t2, (int)cols, (int)rows, smooth);
    hipLaunchKernelGGL(k_transpose_f32, dim3(ceil_div(rows, 64), ceil_div(cols, 64)), dim3(256), 0, s, (const float *)t1, sm, (int)cols, (int)rows);
    hipLaunchKernelGGL(k_isef_cols, dim3(ceil_div(cols, 64)), dim3(576), 0, s, sm, tmp, (int)rows, (int)cols, smooth);
}

static int shen_castan_impl(const zg_image *src, const zg_image *dst, float smooth, uint32_t window_size, float high_ratio, float low_rel, int hysteresis,
                            int use_nms, zg_stream stream) {
    hipStream_t s = as_stream(stream);
    int rc;
    if ((rc = check_image(src, "src")) || (rc = check_image(dst, "dst"))) return rc;
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "shenCastan: %ux%u vs %ux%u", src->rows, src->cols,
               dst->rows, dst->cols);
    ZG_REQUIRE(dst->pixel == ZG_PIXEL_U8, ZG_ERR_INVALID_ARGUMENT, "shenCastan: the output is Image(u8)");
    ZG_REQUIRE(smooth > 0 && smooth < 1, ZG_ERR_INVALID_ARGUMENT, "shenCastan: InvalidBParameter (smooth %g not in (0, 1))", (double)smooth);
    ZG_REQUIRE(window_size % 2 == 1, ZG_ERR_INVALID_ARGUMENT, "shenCastan: WindowSizeMustBeOdd (%u)", window_size);
    ZG_REQUIRE(window_size >= 3, ZG_ERR_INVALID_ARGUMENT, "shenCastan: WindowSizeTooSmall (%u)", window_size);
    ZG_REQUIRE(high_ratio > 0 && high_ratio < 1 && low_rel > 0 && low_rel < 1, ZG_ERR_INVALID_ARGUMENT, "shenCastan: InvalidThreshold (high_ratio %g, low_rel %g)",
               (double)high_ratio, (double)low_rel);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    const uint32_t rows = src->rows, cols = src->cols;
    const size_t n = (size_t)rows * cols, nf = (n + 3) / 4 * 4;

    // scratch: f32 planes grey | smoothed | temp (ISEF) then grey*BLI | gradient | three SATs; u8 planes BLI | candidates | NMS | state;
    // histogram (256) | thresholds (2) | hysteresis work
    char *scratch = nullptr;
    const size_t f32_bytes = 7 * nf * sizeof(float), u8_off = f32_bytes, small_off = (u8_off + 6 * nf + 255) / 256 * 256;
    constexpr size_t small_bytes = SC_HIST_COPIES * 256 * sizeof(unsigned int) + 256; // histogram copies | thresholds
    const size_t check_off = (small_off + small_bytes + hysteresis_work_bytes(rows, cols) + 255) / 256 * 256; // the segmented ISEF's check words
    if ((rc = scratch_alloc((void **)&scratch, check_off + isef_check_bytes(rows, cols), s))) return rc;
    float *gray = (float *)scratch, *sm = gray + nf, *temp = sm + nf, *grad = temp + nf, *sat_g = grad + nf, *sat_m = sat_g + nf, *sat_gm = sat_m + nf;
    uint8_t *bli = (uint8_t *)(scratch + u8_off), *cand = bli + nf, *nms = cand + nf, *state = nms + nf, *gray8 = state + nf, *gm8 = gray8 + nf;
    unsigned int *hist = (unsigned int *)(scratch + small_off);
    float *thr = (float *)(hist + SC_HIST_COPIES * 256);
    char *work = scratch + small_off + small_bytes;
    // Rows of whole dwords: grey and grey * BLI (integers 0 .. 255) also live as BYTES, and that is what the BLI kernel and the three integral
    // images read — a quarter of the f32 planes' traffic (the recursions keep the f32 plane).
    const bool bytes = cols % 4 == 0;
    const bool gray_f32 = !(bytes && isef_2d_applies(rows, cols)); // nothing reads the f32 grey when the recursions take the bytes too

    rc = dispatch_pixel(src->pixel, [&](auto tag) -> int {
        constexpr int PIX = decltype(tag)::value;
        launch_canny_gray<PIX>(src, gray_f32 ? gray : nullptr, s, bytes ? gray8 : nullptr);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    });
    const dim3 g64(ceil_div(cols, 64), ceil_div(rows, 4));
    if (rc == ZG_OK) {
        // rows: transpose, column recursions on the cols x rows plane (in place on `grad`, `sat_g` between the passes: both are
        // free until the gradient stage), transpose back into `sm`; then the columns proper
        // the recursions in segments along the rows and then the columns (isef.hip); planes whose rows are not whole 16-byte chunks take round 3's
        // route: transpose, column recursions on the cols x rows plane (in place on `grad`, `sat_g` between the passes: both are free until
        // the gradient stage), transpose back into `sm`, then the columns proper
        isef_plane(gray_f32 ? gray : nullptr, bytes ? gray8 : nullptr, sm, temp, grad, sat_g, (uint32_t *)(scratch + check_off), rows, cols, smooth, s);
        if (cols % 4 == 0) { // four pixels per lane (the planes start 16 bytes aligned)
            const dim3 g4(ceil_div(cols, 256), ceil_div(rows, 4u * SC_BLI_ROWS));
            if (!use_nms) hipLaunchKernelGGL(k_sc_bli4<0>, g4, dim3(256), 0, s, (const uint8_t *)gray8, (const float *)sm, bli, gm8 /* grey * BLI */, cand, (int)rows, (int)cols, hist);
            else if (rows >= 3 && cols >= 3) hipLaunchKernelGGL(k_sc_bli4<1>, g4, dim3(256), 0, s, (const uint8_t *)gray8, (const float *)sm, bli, gm8, cand, (int)rows, (int)cols, hist);
            else hipLaunchKernelGGL(k_sc_bli4<2>, g4, dim3(256), 0, s, (const uint8_t *)gray8, (const float *)sm, bli, gm8, cand, (int)rows, (int)cols, hist);
        } else {
            hipLaunchKernelGGL(k_sc_bli, g64, dim3(256), 0, s, (const float *)gray, (const float *)sm, bli, temp /* grey * BLI */, cand, (int)rows, (int)cols, use_nms ? 0 : 1);
        }
        if (hipGetLastError() != hipSuccess) rc = ZG_ERR_HIP;
    }
    // frames of at most 2^24 pixels, windows up to 15 x 15: the mask's window count is exact in the reference's f32 integral image, so it is
    // counted directly (k_sc_count) and only grey and grey * BLI go through integral images
    static const bool three_sats = getenv("ZIGNAL_HIP_SC_THREE_SATS") != nullptr; // tuning hook: the mask's integral image as well
    const bool counted = bytes && !three_sats && n <= ((size_t)1 << 24) && window_size / 2 <= 7 && n < (1u << 30);
    uint8_t *cnt8 = (uint8_t *)sat_m; // the plane the mask's integral image would take
    if (rc == ZG_OK) {
        const zg_image gi{bytes ? (void *)gray8 : (void *)gray, cols, rows, cols, bytes ? ZG_PIXEL_U8 : ZG_PIXEL_F32}, mi{bli, cols, rows, cols, ZG_PIXEL_U8},
            gmi{bytes ? (void *)gm8 : (void *)temp, cols, rows, cols, bytes ? ZG_PIXEL_U8 : ZG_PIXEL_F32};
        // grey is as(f32, u8), BLI is 0 / 1, grey * BLI is their product: integer-valued planes, exact row sums
        if (counted) {
            const zg_image *srcs[2] = {&gi, &gmi};
            float *sats[2] = {sat_g, sat_gm};
            rc = sat_planes_multi(srcs, sats, 2, s);
            const dim3 gs(ceil_div(cols, 1024), ceil_div(rows, (uint32_t)SC_COUNT_SEG));
            switch (window_size / 2) { // cols % 4 == 0 here
            case 1: hipLaunchKernelGGL(k_sc_count_stream<1>, gs, dim3(256), 0, s, (const uint8_t *)bli, cnt8, (int)rows, (int)cols); break;
            case 2: hipLaunchKernelGGL(k_sc_count_stream<2>, gs, dim3(256), 0, s, (const uint8_t *)bli, cnt8, (int)rows, (int)cols); break;
            case 3: hipLaunchKernelGGL(k_sc_count_stream<3>, gs, dim3(256), 0, s, (const uint8_t *)bli, cnt8, (int)rows, (int)cols); break;
            default: hipLaunchKernelGGL(k_sc_count, dim3(ceil_div(cols, 64), ceil_div(rows, 32)), dim3(256), 0, s, (const uint8_t *)bli, cnt8, (int)rows, (int)cols, (int)(window_size / 2));
            }
        } else {
            const zg_image *srcs[3] = {&gi, &mi, &gmi};
            float *sats[3] = {sat_g, sat_m, sat_gm};
            rc = sat_planes_multi(srcs, sats, 3, s);
        }
    }
    if (rc == ZG_OK) {
        if (!bytes && hipMemsetAsync(hist, 0, SC_HIST_COPIES * 256 * sizeof(unsigned int), s) != hipSuccess) rc = ZG_ERR_HIP; // (k_sc_bli4 cleared it otherwise)
        if (counted)
            hipLaunchKernelGGL((k_sc_gradient<true, true>), dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, s, (const uint8_t *)cand, (const float *)sat_g, (const float *)cnt8, (const float *)sat_gm, grad, hist,
                               (int)rows, (int)cols, (int)(window_size / 2));
        else if (n < (1u << 30))
            hipLaunchKernelGGL(k_sc_gradient<true>, dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, s, (const uint8_t *)cand, (const float *)sat_g, (const float *)sat_m, (const float *)sat_gm, grad, hist,
                               (int)rows, (int)cols, (int)(window_size / 2));
        else
            hipLaunchKernelGGL(k_sc_gradient<false>, dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, s, (const uint8_t *)cand, (const float *)sat_g, (const float *)sat_m, (const float *)sat_gm, grad, hist,
                               (int)rows, (int)cols, (int)(window_size / 2));
        hipLaunchKernelGGL(k_sc_thresholds, dim3(1), dim3(256), 0, s, (const unsigned int *)hist, thr, high_ratio, low_rel);
        const uint8_t *final_cand = cand;
        if (use_nms) {
            hipLaunchKernelGGL(k_sc_nms, g64, dim3(256), 0, s, (const float *)sm, (const float *)grad, (const uint8_t *)cand, nms, (int)rows, (int)cols);
            final_cand = nms;
        }
        hipLaunchKernelGGL(k_sc_classify4, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, final_cand, (const float *)grad, (const float *)thr, state, n, hysteresis ? 1 : 0);
        if (hipGetLastError() != hipSuccess) rc = ZG_ERR_HIP;
    }
    if (rc == ZG_OK) rc = hysteresis ? run_hysteresis(state, rows, cols, work, dst, s, "shenCastan") : launch_emit(state, nullptr, nullptr, dst, s);
    scratch_free(scratch, s);
    return rc;
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_sobel(const zg_image *src, const zg_image *dst, zg_stream stream) { return sobel_impl(src, dst, as_stream(stream)); }

int zg_sobel_host(const zg_image *src, const zg_image *dst) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = sobel_impl(&a.dev, &b.dev, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

int zg_canny(const zg_image *src, const zg_image *dst, float sigma, float low_threshold, float high_threshold, zg_stream stream) {
    return canny_impl(src, dst, sigma, low_threshold, high_threshold, stream);
}

int zg_canny_host(const zg_image *src, const zg_image *dst, float sigma, float low_threshold, float high_threshold) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = canny_impl(&a.dev, &b.dev, sigma, low_threshold, high_threshold, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

int zg_isef_smooth(const zg_image *src, const zg_image *dst, float smooth, zg_stream stream) {
    ZG_REQUIRE(src && dst && src->data && dst->data, ZG_ERR_INVALID_ARGUMENT, "isef: null image");
    ZG_REQUIRE((src->pixel == ZG_PIXEL_F32 || src->pixel == ZG_PIXEL_U8) && dst->pixel == ZG_PIXEL_F32, ZG_ERR_INVALID_ARGUMENT, "isef: an f32 or u8 plane into an f32 plane");
    ZG_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, ZG_ERR_DIMENSION_MISMATCH, "isef: %ux%u vs %ux%u", src->rows, src->cols, dst->rows, dst->cols);
    ZG_REQUIRE(src->stride == src->cols && dst->stride == dst->cols, ZG_ERR_UNSUPPORTED, "isef: contiguous planes");
    ZG_REQUIRE(smooth > 0 && smooth < 1, ZG_ERR_INVALID_ARGUMENT, "isef: InvalidBParameter (smooth %g not in (0, 1))", (double)smooth);
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    hipStream_t s = as_stream(stream);
    const uint32_t rows = src->rows, cols = src->cols;
    const size_t nf = ((size_t)rows * cols + 3) / 4 * 4;
    char *scratch = nullptr;
    if (const int rc = scratch_alloc((void **)&scratch, 4 * nf * sizeof(float) + isef_check_bytes(rows, cols), s)) return rc;
    float *tmp = (float *)scratch, *t1 = tmp + nf, *t2 = t1 + nf, *as_f32 = t2 + nf;
    const float *gray = (const float *)src->data;
    const uint8_t *gray8 = nullptr;
    if (src->pixel == ZG_PIXEL_U8) { // as the detector has it: bytes for the segmented row pass, f32 only where that does not apply
        gray8 = (const uint8_t *)src->data;
        gray = nullptr;
        // isef_2d also wants a 16-byte aligned destination; when it is not, isef_plane falls through to the transposing route, which reads the
        // f32 plane: it has to exist then (the null plane was an illegal access)
        if (!isef_2d_applies(rows, cols) || ((uintptr_t)gray8 & 15) || ((uintptr_t)dst->data & 15)) {
            launch_canny_gray<ZG_PIXEL_U8>(src, as_f32, s); // as(f32, u8)
            gray = as_f32;
            gray8 = nullptr;
        }
    }
    isef_plane(gray, gray8, (float *)dst->data, tmp, t1, t2, (uint32_t *)(as_f32 + nf), rows, cols, smooth, s);
    const hipError_t e = hipGetLastError();
    scratch_free(scratch, s);
    ZG_HIP(e);
    return ZG_OK;
}

int zg_shen_castan(const zg_image *src, const zg_image *dst, float smooth, uint32_t window_size, float high_ratio, float low_rel, int hysteresis,
                   int use_nms, zg_stream stream) {
    return shen_castan_impl(src, dst, smooth, window_size, high_ratio, low_rel, hysteresis, use_nms, stream);
}

int zg_shen_castan_host(const zg_image *src, const zg_image *dst, float smooth, uint32_t window_size, float high_ratio, float low_rel, int hysteresis,
                        int use_nms) {
    HostStage a, b;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    if ((rc = b.upload(dst, false, true))) return rc;
    if ((rc = shen_castan_impl(&a.dev, &b.dev, smooth, window_size, high_ratio, low_rel, hysteresis, use_nms, nullptr))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return b.finish();
}

// ImagePyramid.build's per-level arithmetic (src/image/pyramid.zig:57-80). `scale` is pow(scale_factor, level) as the
// CALLER's maths library computes it (a Zig host passes std.math.pow's value); zg_pyramid_scale is the library's own.
float zg_pyramid_scale(float scale_factor, uint32_t level) { return hostmath::pow_f32(scale_factor, (float)level); }

// One level of ImagePyramid.build (pyramid.zig:76-92) in a single call: blur the ORIGINAL with `sigma` when sigma > 0.5
// (library scratch holds the blurred copy), then bilinear resize into `level`. `sigma` is zg_pyramid_level's output.
} // extern "C"
namespace zg {
int resize_impl_bilinear_u8(const zg_image *src, const zg_image *dst, hipStream_t s) {
    const zg_method bilinear{ZG_INTERP_BILINEAR, 0.0f, 0.0f, nullptr};
    return zg_resize(src, dst, &bilinear, (zg_stream)s);
}
} // namespace zg
extern "C" {
int zg_pyramid_build_level(const zg_image *source, const zg_image *level, float sigma, zg_stream stream) {
    int rc;
    if ((rc = check_image(source, "source")) || (rc = check_image(level, "level"))) return rc;
    ZG_REQUIRE(source->pixel == level->pixel, ZG_ERR_INVALID_ARGUMENT, "pyramid level: pixel types differ");
    const zg_method bilinear{ZG_INTERP_BILINEAR, 0.0f, 0.0f, nullptr};
    if (!(sigma > 0.5f) || source->rows == 0 || source->cols == 0) return zg_resize(source, level, &bilinear, stream);
    hipStream_t s = as_stream(stream);
    if (source->pixel == ZG_PIXEL_U8 && sigma > 0) { // Image(u8), what ORB builds its pyramid on: the column pass evaluated where the resize looks (conv_sep_bytes2.hip)
        const int n = zg_gaussian_kernel(sigma, nullptr, 0);
        if (n > 0 && n <= 65) {
            float taps[65];
            int32_t itaps[65];
            if (zg_gaussian_kernel(sigma, taps, 65) == n) {
                for (int i = 0; i < n; ++i) itaps[i] = (int32_t)std::round(taps[i] * 256.0f); // scaleKernelToInt (convolution.zig:303-309)
                int z = 0; // outer taps that rounded to zero add nothing to an integer sum (conv_separable.hip drops them the same way)
                while (n - 2 * z > 2 && itaps[z] == 0 && itaps[n - 1 - z] == 0) ++z;
                const int rcf = try_pyramid_level_u8(source, level, itaps + z, n - 2 * z, s);
                if (rcf >= 0) return rcf;
            }
        }
    }
    void *blurred = nullptr;
    if ((rc = scratch_alloc(&blurred, (size_t)source->rows * source->cols * pixel_size(source->pixel), s))) return rc;
    const zg_image tmp{blurred, source->cols, source->rows, source->cols, source->pixel};
    rc = zg_gaussian_blur(source, &tmp, sigma, stream);
    if (rc == ZG_OK) rc = zg_resize(&tmp, level, &bilinear, stream);
    scratch_free(blurred, s);
    return rc;
}

// ImagePyramid.build (pyramid.zig:31-102) as one device operation: every level is made from the ORIGINAL (blur with its own sigma, then
// bilinear resize), so the levels are independent of each other — they are enqueued on a few internal streams forked from `stream`
// and joined back into it (events; no host synchronisation, and the fork / join is what stream capture expects, so the whole build
// records into a graph). levels[i] / sigmas[i] are pyramid level i + 1: shapes from zg_pyramid_level, memory from the caller.
extern "C++" {
namespace {
struct LevelStreams {
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
};
// The helper streams the levels fork over: per thread (a stream can sit in one capture at a time, so two threads recording pyramids must not
// share them) and per device, created the first time a build on that device wants to fork, destroyed when the thread ends. A stream that
// cannot be created leaves its lane out (the build then forks over fewer lanes, or not at all): forking is an optimisation.
struct LevelStreamPool {
    std::map<int, LevelStreams> by_device;
    ~LevelStreamPool() {
        for (auto &kv : by_device)
            for (hipStream_t st : kv.second.s)
                if (st) (void)hipStreamDestroy(st);
    }
};
LevelStreams &level_streams() {
    static thread_local LevelStreamPool pool;
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto it = pool.by_device.find(dev);
    if (it == pool.by_device.end()) {
        it = pool.by_device.emplace(dev, LevelStreams{}).first;
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed; // creating a stream is an "unsafe" call while a capture is in progress
        const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        for (hipStream_t &st : it->second.s)
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
        if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
    }
    return it->second;
}
} // namespace
} // extern "C++"

int zg_pyramid_build(const zg_image *source, const zg_image *levels, const float *sigmas, uint32_t n_levels, zg_stream stream) {
    ZG_REQUIRE(source && (n_levels == 0 || (levels && sigmas)), ZG_ERR_INVALID_ARGUMENT, "pyramid build: null argument");
    if (n_levels == 0) return ZG_OK;
    hipStream_t main = as_stream(stream);
    // Forking pays when the device is the bottleneck — a graph replay: 445 us on one stream, 351 us on four for ORB's default pyramid of
    // a 4096^2 plane — and costs when the host is (eager launches: 508 us on one stream, 661 us on four: every cross-stream hand-off
    // is host work). So the levels fork under stream capture and stay on `stream` otherwise.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
    int want = capturing == hipStreamCaptureStatusActive ? 4 : 1;
    if (n_levels < (uint32_t)want) want = (int)n_levels;
    hipStream_t lanes[4] = {main, nullptr, nullptr, nullptr};
    int n_lanes = 1;
    if (want > 1) { // the pool is only touched (and its streams only created) by a build that forks
        const LevelStreams &pool = level_streams();
        for (int i = 1; i < want && pool.s[i - 1]; ++i) { lanes[i] = pool.s[i - 1]; n_lanes = i + 1; }
    }
    hipEvent_t fork = nullptr, join[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = ZG_OK;
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == ZG_OK) rc = hip_fail(e, what, __FILE__, __LINE__);
    };
    if (n_lanes > 1) {
        hip(hipEventCreateWithFlags(&fork, hipEventDisableTiming), "hipEventCreateWithFlags");
        if (rc == ZG_OK) hip(hipEventRecord(fork, main), "hipEventRecord");
        for (int l = 1; l < n_lanes && rc == ZG_OK; ++l) hip(hipStreamWaitEvent(lanes[l], fork, 0), "hipStreamWaitEvent");
    }
    // Image(u8): the long-tap levels as one batch on the last lane (three launches + the resizes of the dense levels), the rest level by level
    std::vector<uint8_t> handled(n_levels, 0);
    if (rc == ZG_OK && n_levels <= 64 && check_image(source, "source") == ZG_OK) {
        bool all_ok = true;
        for (uint32_t i = 0; i < n_levels; ++i) all_ok = all_ok && check_image(&levels[i], "level") == ZG_OK && levels[i].pixel == source->pixel;
        if (all_ok) {
            // round 6: every level whose taps fit (<= 35, each <= 255) as one tile kernel with no plane in between (pyramid_tile.hip); what it leaves goes on as before
            const int rct = try_pyramid_tiles_u8(source, levels, sigmas, n_levels, handled.data(), lanes[n_lanes - 1]);
            if (rct > 0) rc = rct;
        }
        if (all_ok && rc == ZG_OK) {
            if (n_lanes >= 3) { // two independent batches on two lanes: levels reduced by 2 and more (fused column pass), and the others (dense + resize)
                const int rcf = try_pyramid_levels_u8(source, levels, sigmas, n_levels, handled.data(), 0, lanes[n_lanes - 1]);
                const int rcd = try_pyramid_levels_u8(source, levels, sigmas, n_levels, handled.data(), 1, lanes[n_lanes - 2]);
                if (rcf > 0) rc = rcf; else if (rcd > 0) rc = rcd;
            } else {
                const int rcb = try_pyramid_levels_u8(source, levels, sigmas, n_levels, handled.data(), 2, lanes[n_lanes - 1]);
                if (rcb > 0) rc = rcb;
            }
        }
    }
    // the largest levels (the longest kernels) first, dealt round the lanes
    uint32_t dealt = 0;
    for (uint32_t i = 0; i < n_levels && rc == ZG_OK; ++i) {
        if (handled[i]) continue;
        rc = zg_pyramid_build_level(source, &levels[i], sigmas[i], (zg_stream)lanes[dealt++ % (uint32_t)n_lanes]);
    }
    for (int l = 1; l < n_lanes; ++l) { // always joined, also after a failure: a forked stream must not be left inside a capture
        if (hipEventCreateWithFlags(&join[l], hipEventDisableTiming) == hipSuccess && hipEventRecord(join[l], lanes[l]) == hipSuccess)
            hip(hipStreamWaitEvent(main, join[l], 0), "hipStreamWaitEvent");
        else
            hip(hipGetLastError(), "pyramid build: join");
    }
    if (fork) (void)hipEventDestroy(fork);
    for (hipEvent_t e : join)
        if (e) (void)hipEventDestroy(e);
    return rc;
}

int zg_pyramid_level(uint32_t rows, uint32_t cols, float scale, float blur_sigma, uint32_t *out_rows, uint32_t *out_cols, float *out_sigma) {
    ZG_REQUIRE(out_rows && out_cols && out_sigma, ZG_ERR_INVALID_ARGUMENT, "pyramid level: null output");
    ZG_REQUIRE(scale > 0, ZG_ERR_INVALID_ARGUMENT, "pyramid level: scale must be positive");
    uint32_t nr = (uint32_t)std::trunc((float)rows / scale), nc = (uint32_t)std::trunc((float)cols / scale);
    *out_rows = nr < 1 ? 1 : nr;
    *out_cols = nc < 1 ? 1 : nc;
    *out_sigma = blur_sigma * std::sqrt(scale * scale - 1.0f);
    return ZG_OK;
}

} // extern "C"
