// zg_runtime.cpp — runtime half of the C ABI: device selection, memory, streams, error text and
// the host-pointer staging used by every zg_<op>_host entry point.
#include "zg_common.h"

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
    }
    *out = nullptr;
    ZG_HIP(hipMallocAsync(out, bytes, s));
    return ZG_OK;
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
        ZG_HIP(hipMemcpy2D(dev.data, (size_t)h->cols * ps, h->data, h->stride * ps, (size_t)h->cols * ps,
                           h->rows, hipMemcpyHostToDevice));
    }
    return ZG_OK;
}

int HostStage::finish() {
    if (!writeback || !dev.data) return ZG_OK;
    const size_t ps = pixel_size(host->pixel);
    ZG_HIP(hipMemcpy2D(host->data, host->stride * ps, dev.data, (size_t)host->cols * ps,
                       (size_t)host->cols * ps, host->rows, hipMemcpyDeviceToHost));
    return ZG_OK;
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
    ZG_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    ZG_HIP(hipStreamSynchronize(as_stream(stream)));
    return ZG_OK;
}

int zg_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, zg_stream stream) {
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
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

size_t zg_pixel_size(int pixel) { return pixel_valid(pixel) ? pixel_size(pixel) : 0; }

} // extern "C"
