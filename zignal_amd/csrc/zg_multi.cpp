// zg_multi.cpp — the node's GPUs behind the C ABI: BASELINE configs[4] (a batch of frames through [blur, resize], sharded over
// the GPUs of one node) for a host that is ONE process in the reference's language (Zig / C++), not a launcher of one Python
// process per GPU.
//
// Frames are independent units (reference src/cli/pipeline.zig:153-179 runs one image at a time): device i of the context
// owns a contiguous block of frames, there is no halo and no collective on the data path. The only exchange is
// distribution: the root device holds the batch, the shards travel to their owners and the results travel back. That is a
// scatter and a gather of point-to-point transfers — over RCCL (ncclSend / ncclRecv grouped into one launch per direction: each
// xGMI peer link carries exactly one shard, nothing rings through third devices) — issued from this one thread against a
// communicator per device (ncclCommInitAll). Each device then runs the same zg_batch_blur_resize on its shard on its own
// stream; the root works on its slice of the caller's buffers in place.
//
// librccl is bound at first use with dlopen, not linked: a Python process already carries PyTorch's own copy (same SONAME,
// found first), a bare process gets the system's, and a host that never asks for more than one GPU never loads it.
#include "zg_common.h"

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

namespace zg {
namespace {

// the slice of the RCCL API this file uses (rccl.h: ncclResult_t is an int with 0 = success, ncclUint8 = 1)
typedef void *nccl_comm;
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(nccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_UINT8 = 1;

Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return ZG_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    ZG_REQUIRE(h != nullptr, ZG_ERR_UNSUPPORTED, "multi-GPU: librccl.so not found (%s)", dlerror());
    Rccl r;
    r.handle = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    ZG_REQUIRE(r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString, ZG_ERR_UNSUPPORTED,
               "multi-GPU: librccl.so lacks an expected entry point");
    g_rccl = r;
    return ZG_OK;
}

#define ZG_NCCL(expr)                                                                                   \
    do {                                                                                                \
        const int _r = (expr);                                                                          \
        if (_r != 0) {                                                                                  \
            set_error("RCCL error %d (%s) at %s:%d: %s", _r, g_rccl.GetErrorString(_r), __FILE__, __LINE__, #expr); \
            return ZG_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

struct DeviceSlot {
    int device = -1;
    hipStream_t stream = nullptr;
    nccl_comm comm = nullptr;
    void *in = nullptr, *out = nullptr; // shard staging on a non-root device (grow-only)
    size_t in_bytes = 0, out_bytes = 0;
};

struct Multi {
    std::vector<DeviceSlot> dev; // dev[0] is the root
    bool loopback = false;       // one device: still pass the root's shard through ncclSend / ncclRecv (diagnostics)
};

struct DeviceScope { // the calling thread's current device, restored on exit
    int saved = 0;
    DeviceScope() { (void)hipGetDevice(&saved); }
    ~DeviceScope() { (void)hipSetDevice(saved); }
};

void shard_range(uint32_t n, int i, int world, uint32_t *begin, uint32_t *end) { // as zignal_amd/sharding.py: blocks differ by at most one frame
    const uint32_t base = n / (uint32_t)world, extra = n % (uint32_t)world;
    *begin = (uint32_t)i * base + ((uint32_t)i < extra ? (uint32_t)i : extra);
    *end = *begin + base + ((uint32_t)i < extra ? 1u : 0u);
}

int ensure(void **p, size_t *have, size_t need) {
    if (*have >= need) return ZG_OK;
    if (*p) ZG_HIP(hipFree(*p));
    *p = nullptr;
    *have = 0;
    ZG_HIP(hipMalloc(p, need));
    *have = need;
    return ZG_OK;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace
} // namespace zg

using namespace zg;

extern "C" {

int zg_multi_create(const int *devices, int n_devices, zg_multi *out) {
    ZG_REQUIRE(out != nullptr, ZG_ERR_INVALID_ARGUMENT, "zg_multi_create: null out pointer");
    *out = nullptr;
    int visible = 0;
    ZG_HIP(hipGetDeviceCount(&visible));
    if (n_devices <= 0) n_devices = visible; // all of them
    ZG_REQUIRE(n_devices >= 1 && n_devices <= visible, ZG_ERR_INVALID_ARGUMENT, "zg_multi_create: %d devices asked for, %d visible", n_devices, visible);
    std::vector<int> list((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        list[(size_t)i] = devices ? devices[i] : i;
        ZG_REQUIRE(list[(size_t)i] >= 0 && list[(size_t)i] < visible, ZG_ERR_INVALID_ARGUMENT, "zg_multi_create: device %d out of range", list[(size_t)i]);
        for (int j = 0; j < i; ++j) ZG_REQUIRE(list[(size_t)j] != list[(size_t)i], ZG_ERR_INVALID_ARGUMENT, "zg_multi_create: device %d listed twice", list[(size_t)i]);
    }
    DeviceScope scope;
    Multi *m = new Multi();
    m->dev.resize((size_t)n_devices);
    m->loopback = getenv("ZIGNAL_HIP_MULTI_LOOPBACK") != nullptr;
    int rc = ZG_OK;
    for (int i = 0; i < n_devices && rc == ZG_OK; ++i) {
        m->dev[(size_t)i].device = list[(size_t)i];
        rc = zg_init(list[(size_t)i]); // gfx950 check + current device
        if (rc == ZG_OK && hipStreamCreateWithFlags(&m->dev[(size_t)i].stream, hipStreamNonBlocking) != hipSuccess)
            rc = hip_fail(hipGetLastError(), "hipStreamCreateWithFlags", __FILE__, __LINE__);
    }
    if (rc == ZG_OK && (n_devices > 1 || m->loopback)) {
        rc = load_rccl();
        if (rc == ZG_OK) {
            std::vector<nccl_comm> comms((size_t)n_devices, nullptr);
            const int r = g_rccl.CommInitAll(comms.data(), n_devices, list.data());
            if (r != 0) {
                set_error("RCCL error %d (%s) in ncclCommInitAll over %d devices", r, g_rccl.GetErrorString(r), n_devices);
                rc = ZG_ERR_HIP;
            } else {
                for (int i = 0; i < n_devices; ++i) m->dev[(size_t)i].comm = comms[(size_t)i];
            }
        }
    }
    if (rc != ZG_OK) {
        (void)zg_multi_destroy((zg_multi)m);
        return rc;
    }
    *out = (zg_multi)m;
    return ZG_OK;
}

int zg_multi_destroy(zg_multi handle) {
    if (!handle) return ZG_OK;
    Multi *m = (Multi *)handle;
    DeviceScope scope;
    for (DeviceSlot &d : m->dev) {
        if (d.device < 0) continue;
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm) (void)g_rccl.CommDestroy(d.comm);
        if (d.in) (void)hipFree(d.in);
        if (d.out) (void)hipFree(d.out);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete m;
    return ZG_OK;
}

int zg_multi_device_count(zg_multi handle) { return handle ? (int)((Multi *)handle)->dev.size() : 0; }

int zg_multi_batch_blur_resize(zg_multi handle, const void *src_frames_root, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, float sigma,
                               void *dst_frames_root, uint32_t out_rows, uint32_t out_cols, const zg_method *method, float times_ms[3]) {
    ZG_REQUIRE(handle != nullptr, ZG_ERR_INVALID_ARGUMENT, "zg_multi_batch_blur_resize: null context");
    ZG_REQUIRE(pixel_valid(pixel), ZG_ERR_INVALID_ARGUMENT, "batch: invalid pixel type %d", pixel);
    ZG_REQUIRE(method != nullptr, ZG_ERR_INVALID_ARGUMENT, "batch: null method");
    if (times_ms) times_ms[0] = times_ms[1] = times_ms[2] = 0.0f;
    if (n_frames == 0 || rows == 0 || cols == 0 || out_rows == 0 || out_cols == 0) return ZG_OK;
    ZG_REQUIRE(src_frames_root && dst_frames_root, ZG_ERR_INVALID_ARGUMENT, "batch: null frame pointer");
    Multi *m = (Multi *)handle;
    const int world = (int)m->dev.size();
    const size_t ps = pixel_size(pixel), in_frame = (size_t)rows * cols * ps, out_frame = (size_t)out_rows * out_cols * ps;
    DeviceScope scope;
    DeviceSlot &root = m->dev[0];
    ZG_HIP(hipSetDevice(root.device));
    ZG_HIP(hipDeviceSynchronize()); // whatever produced the caller's frames on the root, on any of its streams, is complete

    // shard staging on the owners
    for (int i = 1; i < world; ++i) {
        uint32_t b, e;
        shard_range(n_frames, i, world, &b, &e);
        ZG_HIP(hipSetDevice(m->dev[(size_t)i].device));
        int rc;
        if ((rc = ensure(&m->dev[(size_t)i].in, &m->dev[(size_t)i].in_bytes, (size_t)(e - b) * in_frame))) return rc;
        if ((rc = ensure(&m->dev[(size_t)i].out, &m->dev[(size_t)i].out_bytes, (size_t)(e - b) * out_frame))) return rc;
    }
    void *loop_in = nullptr; // one-device diagnostics: the root's shard makes a round trip through the communicator first
    if (world == 1 && m->loopback) {
        ZG_HIP(hipSetDevice(root.device));
        int rc;
        if ((rc = ensure(&root.in, &root.in_bytes, (size_t)n_frames * in_frame))) return rc;
        loop_in = root.in;
    }

    auto sync_all = [&]() -> int {
        for (DeviceSlot &d : m->dev) {
            ZG_HIP(hipSetDevice(d.device));
            ZG_HIP(hipStreamSynchronize(d.stream));
        }
        return ZG_OK;
    };

    // ---- scatter: one grouped launch, root sends shard i to device i, device i receives it --------------------------------
    int rc;
    double t0 = now_ms();
    if (world > 1 || loop_in) {
        ZG_NCCL(g_rccl.GroupStart());
        for (int i = (loop_in ? 0 : 1); i < world; ++i) {
            uint32_t b, e;
            shard_range(n_frames, i, world, &b, &e);
            if (e == b) continue;
            const size_t bytes = (size_t)(e - b) * in_frame;
            ZG_NCCL(g_rccl.Send((const char *)src_frames_root + (size_t)b * in_frame, bytes, NCCL_UINT8, i, root.comm, root.stream));
            ZG_NCCL(g_rccl.Recv(i == 0 ? loop_in : m->dev[(size_t)i].in, bytes, NCCL_UINT8, 0, m->dev[(size_t)i].comm, m->dev[(size_t)i].stream));
        }
        ZG_NCCL(g_rccl.GroupEnd());
        if (times_ms) {
            if ((rc = sync_all())) return rc;
            times_ms[0] = (float)(now_ms() - t0);
        }
    }

    // ---- compute: every device runs the batch kernel on its shard, on its own stream -----------------------------------
    t0 = now_ms();
    for (int i = 0; i < world; ++i) {
        uint32_t b, e;
        shard_range(n_frames, i, world, &b, &e);
        if (e == b) continue;
        DeviceSlot &d = m->dev[(size_t)i];
        ZG_HIP(hipSetDevice(d.device));
        const void *in = i == 0 ? (loop_in ? loop_in : (const void *)((const char *)src_frames_root + (size_t)b * in_frame)) : d.in;
        void *outp = i == 0 ? (void *)((char *)dst_frames_root + (size_t)b * out_frame) : d.out;
        if ((rc = zg_batch_blur_resize(in, e - b, rows, cols, pixel, sigma, outp, out_rows, out_cols, method, (zg_stream)d.stream))) return rc;
    }
    if (times_ms) {
        if ((rc = sync_all())) return rc;
        times_ms[1] = (float)(now_ms() - t0);
    }

    // ---- gather: the results travel back the same way (stream order on each device puts them behind its kernel) ----------
    t0 = now_ms();
    if (world > 1) {
        ZG_NCCL(g_rccl.GroupStart());
        for (int i = 1; i < world; ++i) {
            uint32_t b, e;
            shard_range(n_frames, i, world, &b, &e);
            if (e == b) continue;
            const size_t bytes = (size_t)(e - b) * out_frame;
            ZG_NCCL(g_rccl.Send(m->dev[(size_t)i].out, bytes, NCCL_UINT8, 0, m->dev[(size_t)i].comm, m->dev[(size_t)i].stream));
            ZG_NCCL(g_rccl.Recv((char *)dst_frames_root + (size_t)b * out_frame, bytes, NCCL_UINT8, i, root.comm, root.stream));
        }
        ZG_NCCL(g_rccl.GroupEnd());
    }
    if ((rc = sync_all())) return rc;
    if (times_ms) times_ms[2] = (float)(now_ms() - t0);
    return ZG_OK;
}

} // extern "C"
