// zg_runtime.cpp — runtime half of the C ABI: device selection, memory, streams, error text and
// the host-pointer staging used by every zg_<op>_host entry point.
#include "zg_common.h"
#include <string.h>
#include <stdlib.h>

#include <cstdarg>
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

// Scratch for the multi-kernel ops (two-pass separable temp, integral image, batch intermediates): stream-ordered from
// the device's default pool. The pool's release threshold is raised once so that a freed scratch block stays mapped for
// the next call instead of going back to the driver at every synchronisation point (a 256 MB remap costs milliseconds).
int scratch_alloc(void **out, size_t bytes, hipStream_t s) {
    static thread_local int tuned_device = -1;
    int dev = 0;
    ZG_HIP(hipGetDevice(&dev));
    if (tuned_device != dev) {
        hipMemPool_t pool;
        ZG_HIP(hipDeviceGetDefaultMemPool(&pool, dev));
        uint64_t keep = ~(uint64_t)0;
        ZG_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
        tuned_device = dev;
        // Settle the pool's first pages before anything depends on them. On some hosts a block taken from a pool that has
        // just grown was seen to lose what the first microseconds of work stored in it (the first zg_png_decode_host /
        // zg_jpeg_decode_host of a fresh process read back zeros, only ever the first, only in processes that reach their
        // first scratch use within a millisecond of creating the HIP context): the driver's clear of newly mapped memory is
        // not ordered against user-queue work. One memset of a block that is then handed back to the pool (the release
        // threshold above keeps it mapped) costs ~1 ms once per thread and device.
        if (!getenv("ZG_NO_POOL_WARMUP")) {
            void *warm = nullptr;
            const size_t warm_bytes = (size_t)64 << 20;
            if (hipMallocAsync(&warm, warm_bytes, s) == hipSuccess) {
                (void)hipMemsetAsync(warm, 0, warm_bytes, s);
                (void)hipFreeAsync(warm, s);
                (void)hipStreamSynchronize(s);
            } else {
                (void)hipGetLastError();
            }
        }
    }
    *out = nullptr;
    ZG_HIP(hipMallocAsync(out, bytes, s));
    return ZG_OK;
}

// Small hipMemcpy / hipMemcpyAsync calls from PAGEABLE host memory were seen, on some hosts, to be invisible to a kernel
// launched right after the copy had been synchronised: the kernel read zeros where the 52 bytes of a PNG's scan data or the
// 256 bytes of a JPEG block had just been "copied" (zg_png_decode_host / zg_jpeg_decode_host in a bare C++ process, up to
// 29 runs in 30 on an affected host, never on others). Copies whose source is pinned memory are DMA reads issued by the GPU
// and ordered in the stream like any other command, so uploads that feed kernels go through a pinned staging buffer.
namespace {
struct PinnedStage {
    void *p = nullptr;
    size_t bytes = 0;
    ~PinnedStage() { if (p) (void)hipHostFree(p); }
};
} // namespace
static const size_t kStageBytes = (size_t)8 << 20, kStageLimit = (size_t)4 << 20;
static int stage_buffer(void **out) {
    static thread_local PinnedStage stage;
    if (!stage.p) {
        ZG_HIP(hipHostMalloc(&stage.p, kStageBytes, hipHostMallocDefault));
        stage.bytes = kStageBytes;
    }
    *out = stage.p;
    return ZG_OK;
}
int upload_pageable_rows(void *dst_dev, const void *src_host, size_t spitch, size_t width, size_t rows, hipStream_t s) {
    if (width == 0 || rows == 0) return ZG_OK;
    // large transfers keep the runtime's own pipelined pageable path (tens of GB/s); the staging buffer is for the small ones,
    // where the copy is a CPU write into device memory rather than a DMA (ZG_NO_PINNED_UPLOAD: diagnostic switch, never stage)
    if (width * rows > kStageLimit || getenv("ZG_NO_PINNED_UPLOAD")) {
        ZG_HIP(hipMemcpy2DAsync(dst_dev, width, src_host, spitch, width, rows, hipMemcpyHostToDevice, s));
        ZG_HIP(hipStreamSynchronize(s));
        return ZG_OK;
    }
    void *stage = nullptr;
    int rc;
    if ((rc = stage_buffer(&stage))) return rc;
    if (width > kStageBytes) { // absurdly wide rows: piecewise
        for (size_t r = 0; r < rows; ++r)
            for (size_t at = 0; at < width; at += kStageBytes) {
                const size_t n = width - at < kStageBytes ? width - at : kStageBytes;
                memcpy(stage, (const char *)src_host + r * spitch + at, n);
                ZG_HIP(hipMemcpyAsync((char *)dst_dev + r * width + at, stage, n, hipMemcpyHostToDevice, s));
                ZG_HIP(hipStreamSynchronize(s));
            }
        return ZG_OK;
    }
    const size_t per = kStageBytes / width;
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t n = rows - r0 < per ? rows - r0 : per;
        if (spitch == width) memcpy(stage, (const char *)src_host + r0 * spitch, n * width);
        else for (size_t r = 0; r < n; ++r) memcpy((char *)stage + r * width, (const char *)src_host + (r0 + r) * spitch, width);
        ZG_HIP(hipMemcpyAsync((char *)dst_dev + r0 * width, stage, n * width, hipMemcpyHostToDevice, s));
        ZG_HIP(hipStreamSynchronize(s)); // the staging buffer is reused by the next chunk / call
    }
    return ZG_OK;
}
int upload_pageable(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s) {
    if (bytes > kStageLimit) {
        ZG_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        ZG_HIP(hipStreamSynchronize(s));
        return ZG_OK;
    }
    const size_t row = (size_t)1 << 20; // as rows of 1 MiB plus a tail
    int rc = upload_pageable_rows(dst_dev, src_host, row, row, bytes / row, s);
    if (rc == ZG_OK && bytes % row) rc = upload_pageable_rows((char *)dst_dev + bytes / row * row, (const char *)src_host + bytes / row * row, bytes % row, bytes % row, 1, s);
    return rc;
}
int download_pageable_rows(void *dst_host, size_t dpitch, const void *src_dev, size_t width, size_t rows, hipStream_t s) {
    if (width == 0 || rows == 0) return ZG_OK;
    if (width * rows > kStageLimit || getenv("ZG_NO_PINNED_UPLOAD")) {
        ZG_HIP(hipMemcpy2DAsync(dst_host, dpitch, src_dev, width, width, rows, hipMemcpyDeviceToHost, s));
        ZG_HIP(hipStreamSynchronize(s));
        return ZG_OK;
    }
    void *stage = nullptr;
    int rc;
    if ((rc = stage_buffer(&stage))) return rc;
    if (width > kStageBytes) {
        for (size_t r = 0; r < rows; ++r)
            for (size_t at = 0; at < width; at += kStageBytes) {
                const size_t n = width - at < kStageBytes ? width - at : kStageBytes;
                ZG_HIP(hipMemcpyAsync(stage, (const char *)src_dev + r * width + at, n, hipMemcpyDeviceToHost, s));
                ZG_HIP(hipStreamSynchronize(s));
                memcpy((char *)dst_host + r * dpitch + at, stage, n);
            }
        return ZG_OK;
    }
    const size_t per = kStageBytes / width;
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t n = rows - r0 < per ? rows - r0 : per;
        ZG_HIP(hipMemcpyAsync(stage, (const char *)src_dev + r0 * width, n * width, hipMemcpyDeviceToHost, s));
        ZG_HIP(hipStreamSynchronize(s));
        if (dpitch == width) memcpy((char *)dst_host + r0 * dpitch, stage, n * width);
        else for (size_t r = 0; r < n; ++r) memcpy((char *)dst_host + (r0 + r) * dpitch, (const char *)stage + r * width, width);
    }
    return ZG_OK;
}
int download_pageable(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s) {
    if (bytes > kStageLimit) {
        ZG_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
        ZG_HIP(hipStreamSynchronize(s));
        return ZG_OK;
    }
    const size_t row = (size_t)1 << 20;
    int rc = download_pageable_rows(dst_host, row, src_dev, row, bytes / row, s);
    if (rc == ZG_OK && bytes % row) rc = download_pageable_rows((char *)dst_host + bytes / row * row, bytes % row, (const char *)src_dev + bytes / row * row, bytes % row, 1, s);
    return rc;
}

void scratch_free(void *p, hipStream_t s) {
    if (p) (void)hipFreeAsync(p, s);
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
    } else {
        // A destination-only twin is still touched once through the runtime before any kernel writes it: on a fresh
        // hipMalloc block a small kernel's stores were observed to be overwritten by the allocation's own (deferred) clear
        // — the first zg_png_decode_host of a process came back all zero in ~1 of 4 runs — and never after a memset.
        ZG_HIP(hipMemset(dev.data, 0, bytes));
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
