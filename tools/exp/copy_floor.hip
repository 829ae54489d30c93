// copy_floor.hip — what a 64 MiB -> 64 MiB transfer (one 4096 x 4096 f32 plane / Rgba(u8) frame: the bytes of every "134 MB kernel" of the
// library) can cost on MI355X when nothing but the transfer is done, by access pattern. Ring of planes >= 1 GiB on each side (no Infinity Cache
// hits), 2000 warm-up launches, HIP events over 48 launches. Patterns:
//   linear   thread i copies float4 i, i + T, ... (T = threads in the grid), U of them in flight           [grid, U, nt loads, nt stores]
//   strip    one WAVE per 1 KiB x R rows column strip, top to bottom, D rows in flight (the stream kernels)  [R, D]
//   tile     one 256-thread workgroup per 256 px x 16 row tile: load all, then store all (the LDS-tiled kernels without their LDS)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/exp/copy_floor tools/exp/copy_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 4096, COLS = 4096; // f32
constexpr size_t PLANE = (size_t)ROWS * COLS * 4;

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_linear(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n) {
    const size_t T = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * T < n; i += U * T) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * T) : src[i + u * T];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * T);
            else dst[i + u * T] = v[u];
        }
    }
    for (; i < n; i += T) dst[i] = src[i];
}

template <int D>
__global__ __launch_bounds__(64) void k_strip(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int strip_rows) {
    const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3;
    uint32_t w = blockIdx.x;
    if (w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3);
    const int sx = w & 15, sy = w >> 4; // 16 strips across a 16 KiB row
    const int y0 = sy * strip_rows;
    const int n = min(strip_rows, ROWS - y0);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, (int)PLANE, 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, (short)0, (int)PLANE, 0x00020000);
    const int voff = sx * 1024 + 16 * (int)threadIdx.x;
    uint32_t so = (uint32_t)y0 * (COLS * 4), dof = so;
    u32x4 a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)so, 0); so += COLS * 4; }
    for (int jb = 0; jb < n; jb += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const u32x4 c = a[u];
            a[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)so, 0); // past the plane: reads 0
            so += COLS * 4;
            if (jb + u < n) __builtin_amdgcn_raw_buffer_store_b128(c, rd, voff + (int)dof, 0, 2);
            dof += COLS * 4;
        }
    }
}

__global__ __launch_bounds__(256) void k_tile(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst) {
    const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3;
    uint32_t w = blockIdx.x;
    if (w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3);
    const int tx = w & 15, ty = w >> 4; // 16 tiles of 256 px across, 16 rows each
    const int lx = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = src[(size_t)(ty * 16 + wave + 4 * k) * (COLS / 4) + tx * 64 + lx];
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], dst + (size_t)(ty * 16 + wave + 4 * k) * (COLS / 4) + tx * 64 + lx);
}


// One wave per 256 px x R row tile, the conv's real traffic shape: R + 4 source rows (two halo rows above and below, re-read by the
// vertical neighbours: L2 / Infinity Cache hits), optionally the narrow left / right halo load of every row (HALO 1: a second VMEM
// load, lane 0 the 8 bytes before the tile's 1 KiB, the other lanes the 8 bytes after it; HALO 2: the same bytes by two scalar loads),
// everything asked for up front, R stores, exit. WAVES tiles side by side per workgroup. XCD 0: dispatch order = address order.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int R, int HALO, int WAVES, bool XCD>
__global__ __launch_bounds__(64 * WAVES) void k_tilewave(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst) {
    uint32_t w = blockIdx.x;
    if (XCD) { const uint32_t nwg = gridDim.x, per_xcd = nwg >> 3; if (w < (per_xcd << 3)) w = (w & 7) * per_xcd + (w >> 3); }
    constexpr int TX = 16 / WAVES; // workgroups across a 16 KiB row
    const int tx = (int)(w % TX) * WAVES + (int)(threadIdx.x >> 6), ty = (int)(w / TX);
    const int lx = threadIdx.x & 63;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, (short)0, (int)PLANE, 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, (short)0, (int)PLANE, 0x00020000);
    const int voff = tx * 1024 + 16 * lx;
    const int hoff = lx == 0 ? max(tx * 1024 - 8, 0) : min(tx * 1024 + 1024, COLS * 4 - 8);
    const int y0 = ty * R - 2;
    u32x4 a[R + 4];
    u32x2 h[R + 4];
#pragma unroll
    for (int i = 0; i < R + 4; ++i) {
        const int y = min(max(y0 + i, 0), ROWS - 1);
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, y * (COLS * 4), 0);
        if (HALO == 1) h[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, hoff, y * (COLS * 4), 0);
        if (HALO == 2) {
            const uint32_t *row = (const uint32_t *)(src + (size_t)y * (COLS * 4));
            const int l = __builtin_amdgcn_readfirstlane(max(tx * 256 - 2, 0)), r = __builtin_amdgcn_readfirstlane(min(tx * 256 + 256, COLS - 2));
            h[i] = u32x2{row[l] ^ row[r], row[l + 1] ^ row[r + 1]};
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        u32x4 o = a[i + 2];
        o[0] ^= a[i][1] ^ a[i + 4][2] ^ a[i + 1][3] ^ a[i + 3][0]; // every loaded row is used
        if (HALO) o[1] ^= h[i][0] ^ h[i + 1][1] ^ h[i + 2][0] ^ h[i + 3][1] ^ h[i + 4][0];
        __builtin_amdgcn_raw_buffer_store_b128(o, rd, voff, (ty * R + i) * (COLS * 4), 2);
    }
}

int main() {
    const int ring = 9;
    uint8_t *src, *dst;
    if (hipMalloc(&src, PLANE * ring) != hipSuccess || hipMalloc(&dst, PLANE * ring) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, PLANE * ring);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        for (int r = 0; r < 2000; ++r) launch(r % ring);
        (void)hipDeviceSynchronize();
        const int reps = 48;
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch(r % ring);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %6.2f us  %5.2f TB/s\n", name, ms * 1e3 / reps, 2.0 * PLANE / (ms * 1e-3 / reps) / 1e12);
    };
    const size_t n4 = PLANE / 16;
    char name[128];
#define LIN(G, U, NTL, NTS) snprintf(name, sizeof name, "linear grid=%d U=%d nt_load=%d nt_store=%d", G, U, NTL, NTS); \
    time(name, [&](int r) { hipLaunchKernelGGL((k_linear<U, NTL, NTS>), dim3(G), dim3(256), 0, 0, (const u32x4 *)(src + PLANE * r), (u32x4 *)(dst + PLANE * r), n4); });
    LIN(1024, 4, false, false) LIN(2048, 4, false, false) LIN(4096, 4, false, false) LIN(8192, 4, false, false) LIN(16384, 1, false, false)
    LIN(2048, 8, false, false) LIN(2048, 4, false, true) LIN(2048, 4, true, true) LIN(2048, 8, false, true) LIN(4096, 4, false, true) LIN(2048, 16, false, true)
    LIN(16384, 1, false, true) LIN(8192, 2, false, true)
#define STR(R, D) snprintf(name, sizeof name, "strip rows=%d in flight=%d (%d waves)", R, D, 16 * ((ROWS + R - 1) / R)); \
    time(name, [&](int r) { hipLaunchKernelGGL((k_strip<D>), dim3(16 * ((ROWS + R - 1) / R)), dim3(64), 0, 0, src + PLANE * r, dst + PLANE * r, R); });
    STR(16, 4) STR(32, 4) STR(32, 8) STR(64, 8) STR(64, 16) STR(128, 16) STR(16, 8) STR(8, 8) STR(8, 4)
    time("tile 256 px x 16 rows per 256-thread workgroup (4096 WGs)", [&](int r) { hipLaunchKernelGGL(k_tile, dim3(4096), dim3(256), 0, 0, (const u32x4 *)(src + PLANE * r), (u32x4 *)(dst + PLANE * r)); });
    time("hipMemcpyAsync device to device", [&](int r) { (void)hipMemcpyAsync(dst + PLANE * r, src + PLANE * r, PLANE, hipMemcpyDeviceToDevice, 0); });

#define TW(R, HALO, WAVES, XCD) snprintf(name, sizeof name, "tilewave %d rows, halo=%d, %d wave(s)/WG, %s order", R, HALO, WAVES, XCD ? "XCD-major" : "address"); \
    time(name, [&](int r) { hipLaunchKernelGGL((k_tilewave<R, HALO, WAVES, XCD>), dim3((16 / WAVES) * (ROWS / R)), dim3(64 * WAVES), 0, 0, src + PLANE * r, dst + PLANE * r); });
    TW(4, 0, 1, false) TW(8, 0, 1, false) TW(16, 0, 1, false) TW(8, 0, 1, true) TW(8, 0, 4, false) TW(8, 0, 4, true) TW(16, 0, 4, false)
    TW(8, 1, 1, false) TW(8, 1, 4, false) TW(8, 2, 1, false) TW(8, 2, 4, false) TW(16, 1, 1, false) TW(16, 1, 4, false) TW(16, 2, 4, false)
    TW(4, 1, 4, false) TW(4, 2, 4, false) TW(12, 1, 4, false) TW(32, 1, 1, false) TW(32, 1, 4, false)
    // the same patterns on four planes per launch are the library's batched geometry: linear copy of 4 planes
    return 0;
}
