// zg_runtime.cpp — runtime half of the C ABI: device selection, memory, streams, events, graphs, error text and
// the host-pointer staging used by every zg_<op>_host entry point.
#include "zg_common.h"
#include <utility>
#include <vector>
#include <unordered_map>
#include <mutex>
#include <new>
#include <condition_variable>
#include <string.h>
#include <stdlib.h>

#include <cstdarg>
#include <thread>
#include <cstdio>
#include <cstring>

namespace zg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    if (e == hipErrorOutOfMemory) return ZG_ERR_OUT_OF_MEMORY;
    return ZG_ERR_HIP;
}

int check_image(const zg_image *im, const char *name, bool device_pointer) {
    ZG_REQUIRE(im != nullptr, ZG_ERR_INVALID_ARGUMENT, "%s: null image descriptor", name);
    ZG_REQUIRE(pixel_valid(im->pixel), ZG_ERR_INVALID_ARGUMENT, "%s: invalid pixel type %d", name, im->pixel);
    if (im->rows == 0 || im->cols == 0) return ZG_OK; // Image.empty is legal
    ZG_REQUIRE(im->data != nullptr, ZG_ERR_INVALID_ARGUMENT, "%s: null data", name);
    ZG_REQUIRE(im->stride >= im->cols, ZG_ERR_INVALID_ARGUMENT, "%s: stride %zu < cols %u", name, im->stride, im->cols);
    ZG_REQUIRE(im->rows <= 0x3fffffffu && im->cols <= 0x3fffffffu, ZG_ERR_INVALID_ARGUMENT, "%s: image too large", name);
    if (!device_pointer) return ZG_OK; // host pixels are staged with hipMemcpy2D: any alignment
    // kernels move whole pixels with one instruction: the first pixel must be naturally aligned (it always is for an
    // allocation or a view of one; only a hand-built pointer can violate it)
    const size_t align = im->pixel == ZG_PIXEL_RGBA_F32 ? 16 : (im->pixel == ZG_PIXEL_U8 || im->pixel == ZG_PIXEL_RGB_U8 ? 1 : 4);
    ZG_REQUIRE(((uintptr_t)im->data % align) == 0, ZG_ERR_INVALID_ARGUMENT, "%s: data pointer is not %zu-byte aligned", name, align);
    return ZG_OK;
}

// Scratch for the multi-kernel ops (two-pass separable temp, integral images, detector planes, codec staging): a small
// caching allocator over hipMalloc. A block goes back to the cache at scratch_free together with an event recorded on the
// stream that used it; the next scratch_alloc of a fitting size takes it, and waits for that event only when it runs on a
// different stream (on the same stream, stream order already separates the two uses). Nothing is returned to the driver
// while the library lives (a 256 MB remap costs milliseconds), beyond a cap of 64 cached blocks.
// Why not hipMallocAsync / the device's default memory pool, which is this exact service: with the image's ROCm 7.2.0
// runtime (a bare process; PyTorch processes load their own bundled runtime first) blocks from that pool lost data — the
// first PNG / JPEG decode of a process read back zeros, 20 runs in 20 on an affected host, 0 in 20 with plain allocations,
// 0 in 20 with the pool under PyTorch's bundled runtime. The pool is therefore not used at all, graph capture included.
//
// Under stream capture a scratch block has to outlive the call — the captured kernels run at every replay — and nobody
// else may touch it between replays. Such blocks are "graph-owned": taken from hipMalloc (with the thread's capture mode
// switched to relaxed for the duration, as any caching allocator under a capture has to) and never handed to anybody
// outside the capture that took them. scratch_free inside the capture makes the block reusable by later calls of the SAME
// capture on the SAME stream (graph order separates the two uses); after the capture ends the blocks stay reserved until
// zg_release_graph_scratch(), which the owner of the graphs calls once they are destroyed.
namespace {
struct CachedBlock {
    void *p;
    size_t bytes;
    int device;
    hipStream_t last;
    hipEvent_t done;
};
struct GraphBlock {
    void *p;
    size_t bytes;
    int device;
    unsigned long long capture_id;
    hipStream_t stream;
    bool in_use;
};
std::mutex g_scratch_mu;
std::vector<CachedBlock> g_scratch_free;                           // oldest first
size_t g_scratch_cached_bytes = 0;                                 // sum over g_scratch_free
std::unordered_map<void *, std::vector<void *>> g_graph_owned;     // zg_graph -> the scratch blocks its capture took

// Idle blocks are kept up to this many bytes (ZIGNAL_HIP_SCRATCH_CACHE_MB, default 2048): enough for the temp planes of a few 4096^2
// calls in flight, small against 288 GB, and a bound — the host-pointer layer stages whole frames through this cache, and a process
// that shares the device with another allocator (PyTorch's) must not find gigabytes parked here. zg_trim_scratch() empties it.
size_t scratch_cache_limit() {
    static const size_t limit = [] {
        const char *e = getenv("ZIGNAL_HIP_SCRATCH_CACHE_MB");
        const long mb = e ? strtol(e, nullptr, 10) : 2048;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return limit;
}
void release_blocks(std::vector<CachedBlock> &drop); // defined below RelaxedCapture
void release_blocks_impl(std::vector<CachedBlock> &drop) {
    for (const CachedBlock &b : drop) {
        (void)hipEventSynchronize(b.done);
        (void)hipEventDestroy(b.done);
        (void)hipFree(b.p);
    }
    drop.clear();
}
void drop_scratch_cache() {
    std::vector<CachedBlock> drop;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        drop.swap(g_scratch_free);
        g_scratch_cached_bytes = 0;
    }
    release_blocks(drop);
}
std::unordered_map<void *, std::pair<size_t, int>> g_scratch_live; // ptr -> (bytes, device)
std::vector<GraphBlock> g_graph_blocks;

bool stream_capture_id(hipStream_t s, unsigned long long *id) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    *id = 0;
    if (hipStreamGetCaptureInfo(s, &st, id) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}
struct RelaxedCapture { // allocation calls are "unsafe" while ANY thread captures in global mode: relax for this thread
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    RelaxedCapture() { if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
    ~RelaxedCapture() { if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
};
// hipEventSynchronize / hipFree are "unsafe" calls while another thread captures in global mode: relaxed for the duration
void release_blocks(std::vector<CachedBlock> &drop) {
    if (drop.empty()) return;
    RelaxedCapture relaxed;
    release_blocks_impl(drop);
}
// Oldest idle blocks out until the cache holds at most `limit` bytes and 64 blocks. Called where the caller is about to pay a
// hipMalloc anyway (a cache miss) and from scratch_free only past TWICE the limit: freeing is a device-wide synchronisation, and a
// steady state whose working set sits a little above the limit (the pipeline's two ping-pong blocks plus a table) must not pay one
// per call.
void evict_down_to(size_t limit) {
    std::vector<CachedBlock> evict;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        while (!g_scratch_free.empty() && (g_scratch_free.size() > 64 || g_scratch_cached_bytes > limit)) {
            evict.push_back(g_scratch_free.front());
            g_scratch_cached_bytes -= g_scratch_free.front().bytes;
            g_scratch_free.erase(g_scratch_free.begin());
        }
    }
    release_blocks(evict);
}
size_t scratch_round(size_t bytes) {
    const size_t unit = (size_t)1 << 20;
    return (bytes + unit - 1) / unit * unit + (bytes == 0 ? unit : 0);
}

int scratch_alloc_captured(void **out, size_t need, int dev, unsigned long long id, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (GraphBlock &g : g_graph_blocks) // a block this capture has already finished with, on this stream
            if (!g.in_use && g.capture_id == id && g.stream == s && g.device == dev && g.bytes >= need && g.bytes <= 2 * need) {
                g.in_use = true;
                *out = g.p;
                return ZG_OK;
            }
    }
    // Nothing from the general cache: whether a cached block is idle can only be learnt from its event, and querying an
    // event invalidates a capture in progress. A capture happens once; a fresh allocation per scratch request is cheap
    // against that.
    void *p = nullptr;
    {
        RelaxedCapture relaxed;
        ZG_HIP(hipMalloc(&p, need));
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    g_graph_blocks.push_back(GraphBlock{p, need, dev, id, s, true});
    *out = p;
    return ZG_OK;
}
} // namespace

int scratch_alloc(void **out, size_t bytes, hipStream_t s) {
    *out = nullptr;
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));
    const size_t need = scratch_round(bytes);
    unsigned long long capture_id = 0;
    if (stream_capture_id(s, &capture_id)) return scratch_alloc_captured(out, need, dev, capture_id, s);
    CachedBlock take{};
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        size_t best = g_scratch_free.size();
        for (size_t i = 0; i < g_scratch_free.size(); ++i) { // best fit, never more than twice what is asked for
            const CachedBlock &b = g_scratch_free[i];
            if (b.device == dev && b.bytes >= need && b.bytes <= 2 * need && (best == g_scratch_free.size() || b.bytes < g_scratch_free[best].bytes)) best = i;
        }
        if (best != g_scratch_free.size()) {
            take = g_scratch_free[best];
            g_scratch_free.erase(g_scratch_free.begin() + (long)best);
            g_scratch_cached_bytes -= take.bytes;
        }
    }
    if (take.p) {
        const hipError_t e = take.last != s ? hipStreamWaitEvent(s, take.done, 0) : hipSuccess;
        if (e != hipSuccess) (void)hipEventSynchronize(take.done); // still ordered, just not asynchronously
        (void)hipEventDestroy(take.done);
    } else {
        evict_down_to(scratch_cache_limit() > need ? scratch_cache_limit() - need : 0); // make room for the block this call will bring back
        hipError_t e = hipMalloc(&take.p, need);
        if (e == hipErrorOutOfMemory) { // give the cache back to the driver and try once more
            (void)hipGetLastError();
            drop_scratch_cache();
            e = hipMalloc(&take.p, need);
        }
        ZG_HIP(e);
        take.bytes = need;
        take.device = dev;
    }
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        g_scratch_live[take.p] = {take.bytes, take.device};
    }
    *out = take.p;
    return ZG_OK;
}

void scratch_free(void *p, hipStream_t s) {
    if (!p) return;
    CachedBlock b{p, 0, 0, s, nullptr};
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        auto it = g_scratch_live.find(p);
        if (it == g_scratch_live.end()) { // a graph-owned block: reusable inside its capture, reserved afterwards
            for (GraphBlock &g : g_graph_blocks)
                if (g.p == p) g.in_use = false;
            return;
        }
        b.bytes = it->second.first;
        b.device = it->second.second;
        g_scratch_live.erase(it);
    }
    if (hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess || hipEventRecord(b.done, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);
        (void)hipFree(p);
        return;
    }
    bool far_over;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        g_scratch_free.push_back(b);
        g_scratch_cached_bytes += b.bytes;
        far_over = g_scratch_free.size() > 64 || g_scratch_cached_bytes > 2 * scratch_cache_limit();
    }
    if (far_over) evict_down_to(scratch_cache_limit()); // the hard bound; between the limit and twice the limit the next cache miss trims
}

// What one scratch block of a long-lived working set may take so that a few of them stay inside the cache (batch.hip's ping-pong blocks).
size_t scratch_block_budget() { return std::max<size_t>(scratch_cache_limit() / 4, (size_t)64 << 20); }

// How many host threads the codecs' host halves may use for one call (deflate pieces, entropy-coding bands).
int host_threads() {
    if (const char *e = getenv("ZIGNAL_HIP_HOST_THREADS")) {
        const long n = strtol(e, nullptr, 10);
        if (n >= 1) return n > 256 ? 256 : (int)n;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 1 : (hw > 16 ? 16 : (int)hw);
}

// Pageable host memory <-> device memory, synchronised before returning (the callers' host buffers are short-lived).
int upload_pageable_rows(void *dst_dev, const void *src_host, size_t spitch, size_t width, size_t rows, hipStream_t s) {
    if (width == 0 || rows == 0) return ZG_OK;
    ZG_HIP(hipMemcpy2DAsync(dst_dev, width, src_host, spitch, width, rows, hipMemcpyHostToDevice, s));
    ZG_HIP(hipStreamSynchronize(s));
    return ZG_OK;
}
int upload_pageable(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s) {
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    ZG_HIP(hipStreamSynchronize(s));
    return ZG_OK;
}
int download_pageable_rows(void *dst_host, size_t dpitch, const void *src_dev, size_t width, size_t rows, hipStream_t s) {
    if (width == 0 || rows == 0) return ZG_OK;
    ZG_HIP(hipMemcpy2DAsync(dst_host, dpitch, src_dev, width, width, rows, hipMemcpyDeviceToHost, s));
    ZG_HIP(hipStreamSynchronize(s));
    return ZG_OK;
}
int download_pageable(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
    ZG_HIP(hipStreamSynchronize(s));
    return ZG_OK;
}

// The device twin of a host image comes from the scratch cache: a second call of the same size pays no allocation.
HostStage::~HostStage() {
    if (dev.data) scratch_free(dev.data, nullptr);
}

int HostStage::upload(const zg_image *h, bool copy_in, bool write_back) {
    int rc = check_image(h, "host image", false);
    if (rc) return rc;
    host = h;
    writeback = write_back;
    dev = *h;
    dev.stride = h->cols;
    dev.data = nullptr;
    const size_t ps = pixel_size(h->pixel);
    const size_t bytes = (size_t)h->rows * h->cols * ps;
    if (bytes == 0) return ZG_OK;
    if ((rc = scratch_alloc(&dev.data, bytes, nullptr))) return rc;
    if (copy_in) {
        if ((rc = upload_pageable_rows(dev.data, h->data, h->stride * ps, (size_t)h->cols * ps, h->rows, nullptr))) return rc;
    }
    return ZG_OK;
}

int HostStage::finish() {
    if (!writeback || !dev.data) return ZG_OK;
    const size_t ps = pixel_size(host->pixel);
    return download_pageable_rows(host->data, host->stride * ps, dev.data, (size_t)host->cols * ps, host->rows, nullptr);
}

// ---- banded host pipeline -----------------------------------------------------------------------------------------------
// A host-pointer call is PCIe time: 256 MB up, 256 MB down for one 4096^2 Rgba(f32) frame, the kernel is 1 % of it. The
// link is full duplex, so a row-local op (each output row needs source rows within `halo` of it) is cut into row bands and
// run as a three-stage pipeline: this thread uploads band k+1 while the device computes band k and a helper thread
// downloads band k-1. The whole source frame lives in one device block (bands read their halo rows out of their
// neighbours' uploads); each band's result goes to one of three rotating band buffers that carry halo rows of their own,
// so the op sees an ordinary image whose top / bottom edge is the frame's edge for the first / last band (the border rule
// applies there exactly as in the whole-frame call) and real neighbour rows elsewhere; rows computed from a band's
// artificial inner edge are simply not downloaded. Bands and the last band's remainder are kept >= halo rows, which makes
// "is this row within `halo` of the edge" (the reference's interior / border classification) agree between a band view
// and the frame for every row that is kept.
namespace {
struct BandStreams {
    hipStream_t up = nullptr, run = nullptr, down = nullptr;
    int device = -1;
    int ensure() {
        int dev = 0;
        ZG_HIP(hipGetDevice(&dev));
        if (device == dev && up) return ZG_OK;
        release();
        ZG_HIP(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        ZG_HIP(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
        ZG_HIP(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
        device = dev;
        return ZG_OK;
    }
    void release() {
        if (up) (void)hipStreamDestroy(up);
        if (run) (void)hipStreamDestroy(run);
        if (down) (void)hipStreamDestroy(down);
        up = run = down = nullptr;
        device = -1;
    }
    ~BandStreams() { release(); }
};
thread_local BandStreams t_band;
} // namespace

int host_banded(const zg_image *src, const zg_image *dst, uint32_t halo, const BandOp &op) {
    if (getenv("ZIGNAL_HIP_NO_BANDS")) return -1;
    if (check_image(src, "src", false) || check_image(dst, "dst", false)) return -1;
    if (src->rows != dst->rows || src->cols != dst->cols || src->rows == 0 || src->cols == 0) return -1;
    const size_t ps_s = pixel_size(src->pixel), ps_d = pixel_size(dst->pixel);
    const size_t row_s = (size_t)src->cols * ps_s, row_d = (size_t)dst->cols * ps_d;
    const size_t bytes_s = row_s * src->rows, bytes_d = row_d * dst->rows;
    if (bytes_s + bytes_d < ((size_t)24 << 20)) return -1; // small frames: one trip each way is as good
    { // an in-place call (or any overlap) would have later bands read rows that earlier bands already overwrote
        const uintptr_t s0 = (uintptr_t)src->data, s1 = s0 + ((size_t)(src->rows - 1) * src->stride + src->cols) * ps_s;
        const uintptr_t d0 = (uintptr_t)dst->data, d1 = d0 + ((size_t)(dst->rows - 1) * dst->stride + dst->cols) * ps_d;
        if (s0 < d1 && d0 < s1) return -1;
    }
    const uint32_t rows = src->rows, min_band = halo > 0 ? 2 * halo : 1;
    size_t band_mib = 16; // ~16 MiB of the larger side per band
    if (const char *e = getenv("ZIGNAL_HIP_BAND_MIB")) { const long v = strtol(e, nullptr, 10); if (v >= 1 && v <= 1024) band_mib = (size_t)v; }
    size_t want = (bytes_s > bytes_d ? bytes_s : bytes_d) / (band_mib << 20);
    if (want < 4) want = 4;
    if (want > 96) want = 96;
    uint32_t band = (uint32_t)((rows + want - 1) / want);
    if (band < min_band) band = min_band;
    uint32_t nb = (rows + band - 1) / band;
    if (nb >= 2 && rows - (nb - 1) * band < (halo ? halo : 1)) --nb; // a short remainder joins the band before it
    if (nb < 3) return -1;
    const uint32_t last_rows = rows - (nb - 1) * band, max_rows = (last_rows > band ? last_rows : band) + 2 * halo;

    int rc = t_band.ensure();
    if (rc) return rc;
    const hipStream_t up = t_band.up, run = t_band.run, down = t_band.down;
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));

    constexpr int NB = 3;
    uint8_t *dsrc = nullptr, *dband = nullptr;
    if ((rc = scratch_alloc((void **)&dsrc, bytes_s, run))) return rc;
    const size_t band_bytes = (row_d * max_rows + 255) / 256 * 256;
    if ((rc = scratch_alloc((void **)&dband, band_bytes * NB, run))) { scratch_free(dsrc, run); return rc; }
    std::vector<hipEvent_t> ev_up(nb, nullptr), ev_run(nb, nullptr);
    hipEvent_t ev_alloc = nullptr;
    auto cleanup = [&](int status) {
        (void)hipStreamSynchronize(up);
        (void)hipStreamSynchronize(run);
        (void)hipStreamSynchronize(down);
        for (hipEvent_t e : ev_up) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_run) if (e) (void)hipEventDestroy(e);
        if (ev_alloc) (void)hipEventDestroy(ev_alloc);
        scratch_free(dband, run);
        scratch_free(dsrc, run);
        return status;
    };
    for (uint32_t k = 0; k < nb; ++k)
        if (hipEventCreateWithFlags(&ev_up[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_run[k], hipEventDisableTiming) != hipSuccess)
            return cleanup(hip_fail(hipGetLastError(), "hipEventCreate", __FILE__, __LINE__));
    // the blocks may have been handed over with a wait queued on `run`: the upload stream starts behind it
    if (hipEventCreateWithFlags(&ev_alloc, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev_alloc, run) != hipSuccess ||
        hipStreamWaitEvent(up, ev_alloc, 0) != hipSuccess)
        return cleanup(hip_fail(hipGetLastError(), "band pipeline setup", __FILE__, __LINE__));

    std::mutex mu;
    std::condition_variable cv;
    uint32_t launched = 0, downloaded = 0;
    bool abort_flag = false;
    int down_rc = ZG_OK;
    char down_err[512] = "";
    auto band_range = [&](uint32_t k, uint32_t *r0, uint32_t *r1, uint32_t *v0, uint32_t *v1) {
        *r0 = k * band;
        *r1 = k + 1 == nb ? rows : (k + 1) * band;
        *v0 = *r0 > halo ? *r0 - halo : 0;
        *v1 = *r1 + halo < rows ? *r1 + halo : rows;
    };

    std::thread downloader([&] {
        if (hipSetDevice(dev) != hipSuccess) {
            std::lock_guard<std::mutex> lock(mu);
            down_rc = ZG_ERR_HIP;
            snprintf(down_err, sizeof(down_err), "band pipeline: hipSetDevice(%d) failed on the download thread", dev);
            abort_flag = true;
            cv.notify_all();
            return;
        }
        for (uint32_t k = 0; k < nb; ++k) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return launched > k || abort_flag; });
                if (launched <= k) return; // aborted before band k was launched
            }
            uint32_t r0, r1, v0, v1;
            band_range(k, &r0, &r1, &v0, &v1);
            int st = ZG_OK;
            if (hipEventSynchronize(ev_run[k]) != hipSuccess) st = hip_fail(hipGetLastError(), "hipEventSynchronize(band)", __FILE__, __LINE__);
            if (!st)
                st = download_pageable_rows((uint8_t *)dst->data + (size_t)r0 * dst->stride * ps_d, dst->stride * ps_d,
                                            dband + band_bytes * (k % NB) + (size_t)(r0 - v0) * row_d, row_d, r1 - r0, down);
            std::lock_guard<std::mutex> lock(mu);
            if (st) {
                down_rc = st;
                snprintf(down_err, sizeof(down_err), "%s", g_err); // this thread's message, for the caller's thread
                abort_flag = true;
                cv.notify_all();
                return;
            }
            downloaded = k + 1;
            cv.notify_all();
        }
    });

    uint32_t uploaded = 0;
    for (uint32_t k = 0; k < nb && rc == ZG_OK; ++k) {
        uint32_t r0, r1, v0, v1;
        band_range(k, &r0, &r1, &v0, &v1);
        if (v1 > uploaded) {
            rc = upload_pageable_rows(dsrc + (size_t)uploaded * row_s, (const uint8_t *)src->data + (size_t)uploaded * src->stride * ps_s,
                                      src->stride * ps_s, row_s, v1 - uploaded, up);
            uploaded = v1;
        }
        if (rc) break;
        if (hipEventRecord(ev_up[k], up) != hipSuccess || hipStreamWaitEvent(run, ev_up[k], 0) != hipSuccess) {
            rc = hip_fail(hipGetLastError(), "band pipeline: upload -> kernel hand-off", __FILE__, __LINE__);
            break;
        }
        { // band buffer k % NB is free once band k - NB has left it
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&] { return downloaded + NB > k || abort_flag; });
            if (abort_flag) break;
        }
        zg_image sv = *src, dv = *dst;
        sv.data = dsrc + (size_t)v0 * row_s;
        sv.stride = src->cols;
        sv.rows = v1 - v0;
        dv.data = dband + band_bytes * (k % NB);
        dv.stride = dst->cols;
        dv.rows = v1 - v0;
        try { // an exception (std::bad_alloc from a tap vector, say) must not unwind past the live downloader thread: that is std::terminate
            rc = op(&sv, &dv, run);
        } catch (const std::bad_alloc &) {
            set_error("band pipeline: out of host memory");
            rc = ZG_ERR_OUT_OF_MEMORY;
        } catch (...) {
            set_error("band pipeline: the operation threw");
            rc = ZG_ERR_INVALID_ARGUMENT;
        }
        if (rc) break;
        if (hipEventRecord(ev_run[k], run) != hipSuccess) {
            rc = hip_fail(hipGetLastError(), "band pipeline: kernel -> download hand-off", __FILE__, __LINE__);
            break;
        }
        std::lock_guard<std::mutex> lock(mu);
        launched = k + 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lock(mu);
        if (rc) abort_flag = true;
        cv.notify_all();
    }
    downloader.join();
    if (!rc && down_rc) {
        rc = down_rc;
        set_error("%s", down_err);
    }
    return cleanup(rc);
}

} // namespace zg

using namespace zg;

extern "C" {

int zg_init(int device) {
    int n = 0;
    ZG_HIP(hipGetDeviceCount(&n));
    ZG_REQUIRE(device >= 0 && device < n, ZG_ERR_INVALID_ARGUMENT, "device %d out of range (%d visible)", device, n);
    ZG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ZG_HIP(hipGetDeviceProperties(&prop, device));
    // This library carries gfx950 code objects only: fail loudly anywhere else.
    ZG_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, ZG_ERR_UNSUPPORTED,
               "device %d is %s; libzignal_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return ZG_OK;
}

// The current device is a property of the calling THREAD (as in HIP): a host that drives eight GPUs runs one thread per
// GPU, each of which calls zg_set_device(i) once; allocations, scratch, tables and launches of that thread then belong to
// GPU i. Everything the library caches is keyed by device.
int zg_set_device(int device) { return zg_init(device); }

int zg_get_device(int *device) {
    ZG_REQUIRE(device, ZG_ERR_INVALID_ARGUMENT, "zg_get_device: null out pointer");
    ZG_HIP(hipGetDevice(device));
    return ZG_OK;
}

void zg_shutdown(void) { (void)hipDeviceSynchronize(); }

const char *zg_last_error(void) { return g_err; }

int zg_version(void) { return 200; } // 0.2.0

int zg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zg_malloc(void **dev_ptr, size_t bytes) {
    ZG_REQUIRE(dev_ptr, ZG_ERR_INVALID_ARGUMENT, "zg_malloc: null out pointer");
    *dev_ptr = nullptr;
    if (bytes == 0) return ZG_OK;
    RelaxedCapture relaxed; // legal while another thread captures
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e == hipErrorOutOfMemory) { // the library's own idle scratch must never be the reason an allocation fails
        (void)hipGetLastError();
        drop_scratch_cache();
        e = hipMalloc(dev_ptr, bytes);
    }
    ZG_HIP(e);
    return ZG_OK;
}

int zg_trim_scratch(void) {
    drop_scratch_cache();
    return ZG_OK;
}

int zg_free(void *dev_ptr) {
    if (dev_ptr) ZG_HIP(hipFree(dev_ptr));
    return ZG_OK;
}

int zg_malloc_host(void **host_ptr, size_t bytes) {
    ZG_REQUIRE(host_ptr, ZG_ERR_INVALID_ARGUMENT, "zg_malloc_host: null out pointer");
    *host_ptr = nullptr;
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return ZG_OK;
}

int zg_free_host(void *host_ptr) {
    if (host_ptr) ZG_HIP(hipHostFree(host_ptr));
    return ZG_OK;
}

int zg_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, zg_stream stream) {
    if (bytes == 0) return ZG_OK;
    return upload_pageable(dst_dev, src_host, bytes, as_stream(stream));
}

int zg_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, zg_stream stream) {
    if (bytes == 0) return ZG_OK;
    return download_pageable(dst_host, src_dev, bytes, as_stream(stream));
}

int zg_memcpy_h2d_async(void *dst_dev, const void *src_host, size_t bytes, zg_stream stream) {
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return ZG_OK;
}

int zg_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, zg_stream stream) {
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return ZG_OK;
}

int zg_image_upload(const zg_image *dst_dev, const zg_image *src_host, zg_stream stream) {
    int rc;
    if ((rc = check_image(dst_dev, "dst")) || (rc = check_image(src_host, "src", false))) return rc;
    ZG_REQUIRE(dst_dev->rows == src_host->rows && dst_dev->cols == src_host->cols, ZG_ERR_DIMENSION_MISMATCH, "upload: %ux%u into %ux%u",
               src_host->rows, src_host->cols, dst_dev->rows, dst_dev->cols);
    ZG_REQUIRE(dst_dev->pixel == src_host->pixel, ZG_ERR_INVALID_ARGUMENT, "upload: pixel types differ");
    if (dst_dev->rows == 0 || dst_dev->cols == 0) return ZG_OK;
    const size_t ps = pixel_size(dst_dev->pixel);
    ZG_HIP(hipMemcpy2DAsync(dst_dev->data, dst_dev->stride * ps, src_host->data, src_host->stride * ps, (size_t)dst_dev->cols * ps, dst_dev->rows,
                            hipMemcpyHostToDevice, as_stream(stream)));
    ZG_HIP(hipStreamSynchronize(as_stream(stream))); // the host rows may be pageable and short-lived
    return ZG_OK;
}

int zg_image_download(const zg_image *dst_host, const zg_image *src_dev, zg_stream stream) {
    int rc;
    if ((rc = check_image(dst_host, "dst", false)) || (rc = check_image(src_dev, "src"))) return rc;
    ZG_REQUIRE(dst_host->rows == src_dev->rows && dst_host->cols == src_dev->cols, ZG_ERR_DIMENSION_MISMATCH, "download: %ux%u into %ux%u",
               src_dev->rows, src_dev->cols, dst_host->rows, dst_host->cols);
    ZG_REQUIRE(dst_host->pixel == src_dev->pixel, ZG_ERR_INVALID_ARGUMENT, "download: pixel types differ");
    if (dst_host->rows == 0 || dst_host->cols == 0) return ZG_OK;
    const size_t ps = pixel_size(dst_host->pixel);
    ZG_HIP(hipMemcpy2DAsync(dst_host->data, dst_host->stride * ps, src_dev->data, src_dev->stride * ps, (size_t)dst_host->cols * ps, dst_host->rows,
                            hipMemcpyDeviceToHost, as_stream(stream)));
    ZG_HIP(hipStreamSynchronize(as_stream(stream)));
    return ZG_OK;
}

int zg_stream_create(zg_stream *out) {
    ZG_REQUIRE(out, ZG_ERR_INVALID_ARGUMENT, "zg_stream_create: null out pointer");
    hipStream_t s;
    ZG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (zg_stream)s;
    return ZG_OK;
}

int zg_stream_destroy(zg_stream s) {
    if (s) ZG_HIP(hipStreamDestroy(as_stream(s)));
    return ZG_OK;
}

int zg_stream_synchronize(zg_stream s) {
    ZG_HIP(hipStreamSynchronize(as_stream(s)));
    return ZG_OK;
}

int zg_event_create(zg_event *out) {
    ZG_REQUIRE(out, ZG_ERR_INVALID_ARGUMENT, "zg_event_create: null out pointer");
    hipEvent_t e;
    ZG_HIP(hipEventCreate(&e));
    *out = (zg_event)e;
    return ZG_OK;
}

int zg_event_destroy(zg_event e) {
    if (e) ZG_HIP(hipEventDestroy((hipEvent_t)e));
    return ZG_OK;
}

int zg_event_record(zg_event e, zg_stream s) {
    ZG_REQUIRE(e, ZG_ERR_INVALID_ARGUMENT, "zg_event_record: null event");
    ZG_HIP(hipEventRecord((hipEvent_t)e, as_stream(s)));
    return ZG_OK;
}

int zg_event_synchronize(zg_event e) {
    ZG_REQUIRE(e, ZG_ERR_INVALID_ARGUMENT, "zg_event_synchronize: null event");
    ZG_HIP(hipEventSynchronize((hipEvent_t)e));
    return ZG_OK;
}

int zg_event_elapsed_ms(zg_event start, zg_event stop, float *ms) {
    ZG_REQUIRE(start && stop && ms, ZG_ERR_INVALID_ARGUMENT, "zg_event_elapsed_ms: null argument");
    ZG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ZG_OK;
}

int zg_stream_wait_event(zg_stream s, zg_event e) {
    ZG_REQUIRE(e, ZG_ERR_INVALID_ARGUMENT, "zg_stream_wait_event: null event");
    ZG_HIP(hipStreamWaitEvent(as_stream(s), (hipEvent_t)e, 0));
    return ZG_OK;
}

// Graphs: everything the library enqueues on `stream` between begin and end becomes one launchable object. Host-side
// work of a call (taps, tables, argument checks) happens at capture time and is baked in.
int zg_graph_begin_capture(zg_stream stream) {
    ZG_REQUIRE(stream, ZG_ERR_INVALID_ARGUMENT, "zg_graph_begin_capture: the default stream cannot be captured; create one with zg_stream_create");
    ZG_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return ZG_OK;
}

int zg_graph_end_capture(zg_stream stream, zg_graph *out) {
    ZG_REQUIRE(stream && out, ZG_ERR_INVALID_ARGUMENT, "zg_graph_end_capture: null argument");
    *out = nullptr;
    unsigned long long capture_id = 0;
    const bool capturing = stream_capture_id(as_stream(stream), &capture_id);
    hipGraph_t g = nullptr;
    ZG_HIP(hipStreamEndCapture(as_stream(stream), &g));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    // the scratch this capture took now belongs to the graph (or goes back at once if there is no graph to own it)
    std::vector<void *> mine;
    if (capturing) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (size_t i = 0; i < g_graph_blocks.size();) {
            if (g_graph_blocks[i].capture_id == capture_id) {
                mine.push_back(g_graph_blocks[i].p);
                g_graph_blocks.erase(g_graph_blocks.begin() + (long)i);
            } else {
                ++i;
            }
        }
        if (e == hipSuccess && !mine.empty()) g_graph_owned[(void *)exec] = mine;
    }
    if (e != hipSuccess)
        for (void *p : mine) (void)hipFree(p);
    ZG_HIP(e);
    *out = (zg_graph)exec;
    return ZG_OK;
}

int zg_graph_launch(zg_graph graph, zg_stream stream) {
    ZG_REQUIRE(graph, ZG_ERR_INVALID_ARGUMENT, "zg_graph_launch: null graph");
    ZG_HIP(hipGraphLaunch((hipGraphExec_t)graph, as_stream(stream)));
    return ZG_OK;
}

int zg_graph_destroy(zg_graph graph) {
    if (!graph) return ZG_OK;
    std::vector<void *> mine;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        auto it = g_graph_owned.find((void *)graph);
        if (it != g_graph_owned.end()) {
            mine.swap(it->second);
            g_graph_owned.erase(it);
        }
    }
    const hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph);
    for (void *p : mine) (void)hipFree(p); // hipFree waits for a replay that is still running
    ZG_HIP(e);
    return ZG_OK;
}

// Scratch of captures the library did not end itself (torch.cuda.graph around Image calls, a caller's own hipStreamEndCapture):
// nobody tells the library when those graphs die, so their blocks stay reserved until the caller says so here — once the graphs
// are destroyed (hipFree waits for whatever is still running). Graphs made with zg_graph_end_capture own their scratch and are
// not touched; neither are blocks of a capture that is still in progress or of a call that is still running.
int zg_release_graph_scratch(void) {
    std::vector<GraphBlock> drop;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (size_t i = 0; i < g_graph_blocks.size();) {
            const GraphBlock &g = g_graph_blocks[i];
            unsigned long long id = 0;
            const bool still_capturing = stream_capture_id(g.stream, &id) && id == g.capture_id;
            if (g.in_use || still_capturing) {
                ++i;
                continue;
            }
            drop.push_back(g);
            g_graph_blocks.erase(g_graph_blocks.begin() + (long)i);
        }
    }
    for (const GraphBlock &g : drop) ZG_HIP(hipFree(g.p));
    return ZG_OK;
}

size_t zg_pixel_size(int pixel) { return pixel_valid(pixel) ? pixel_size(pixel) : 0; }

} // extern "C"
