// zg_runtime.cpp — runtime half of the C ABI: device selection, memory, streams, error text and
// the host-pointer staging used by every zg_<op>_host entry point.
#include "zg_common.h"
#include <utility>
#include <vector>
#include <unordered_map>
#include <mutex>
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
// 0 in 20 with the pool under PyTorch's bundled runtime. The pool is still used while a stream is being captured into a
// graph, because there the allocation has to belong to the graph.
namespace {
struct CachedBlock {
    void *p;
    size_t bytes;
    int device;
    hipStream_t last;
    hipEvent_t done;
};
std::mutex g_scratch_mu;
std::vector<CachedBlock> g_scratch_free;                       // oldest first
std::unordered_map<void *, std::pair<size_t, int>> g_scratch_live; // ptr -> (bytes, device)
bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}
} // namespace

int scratch_alloc(void **out, size_t bytes, hipStream_t s) {
    *out = nullptr;
    if (stream_is_capturing(s)) {
        ZG_HIP(hipMallocAsync(out, bytes ? bytes : 1, s));
        return ZG_OK;
    }
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));
    const size_t unit = (size_t)1 << 20, need = (bytes + unit - 1) / unit * unit + (bytes == 0 ? unit : 0);
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
        }
    }
    if (take.p) {
        const hipError_t e = take.last != s ? hipStreamWaitEvent(s, take.done, 0) : hipSuccess;
        if (e != hipSuccess) (void)hipEventSynchronize(take.done); // still ordered, just not asynchronously
        (void)hipEventDestroy(take.done);
    } else {
        hipError_t e = hipMalloc(&take.p, need);
        if (e == hipErrorOutOfMemory) { // give the cache back to the driver and try once more
            (void)hipGetLastError();
            std::vector<CachedBlock> drop;
            {
                std::lock_guard<std::mutex> lock(g_scratch_mu);
                drop.swap(g_scratch_free);
            }
            for (const CachedBlock &b : drop) {
                (void)hipEventSynchronize(b.done);
                (void)hipEventDestroy(b.done);
                (void)hipFree(b.p);
            }
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
        if (it == g_scratch_live.end()) { // a graph-owned (captured) allocation
            (void)hipFreeAsync(p, s);
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
    CachedBlock evict{};
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        g_scratch_free.push_back(b);
        if (g_scratch_free.size() > 64) {
            evict = g_scratch_free.front();
            g_scratch_free.erase(g_scratch_free.begin());
        }
    }
    if (evict.p) {
        (void)hipEventSynchronize(evict.done);
        (void)hipEventDestroy(evict.done);
        (void)hipFree(evict.p);
    }
}

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

HostStage::~HostStage() {
    if (dev.data) (void)hipFree(dev.data);
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
    ZG_HIP(hipMalloc(&dev.data, bytes));
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

void zg_shutdown(void) { (void)hipDeviceSynchronize(); }

const char *zg_last_error(void) { return g_err; }

int zg_version(void) { return 100; } // 0.1.0

int zg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zg_malloc(void **dev_ptr, size_t bytes) {
    ZG_REQUIRE(dev_ptr, ZG_ERR_INVALID_ARGUMENT, "zg_malloc: null out pointer");
    *dev_ptr = nullptr;
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMalloc(dev_ptr, bytes));
    return ZG_OK;
}

int zg_free(void *dev_ptr) {
    if (dev_ptr) ZG_HIP(hipFree(dev_ptr));
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

size_t zg_pixel_size(int pixel) { return pixel_valid(pixel) ? pixel_size(pixel) : 0; }

} // extern "C"
