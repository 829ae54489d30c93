// zg_multi.cpp — the node's GPUs behind the C ABI: BASELINE configs[4] (a batch of frames through [blur, resize], sharded over
// the GPUs of one node) for a host that is ONE process in the reference's language (Zig / C++), not a launcher of one Python
// process per GPU.
//
// Frames are independent units (reference src/cli/pipeline.zig:153-179 runs one image at a time): device i of the context
// owns a contiguous block of frames, there is no halo and no collective on the data path. The only exchange is
// distribution: the root device holds the batch, the shards travel to their owners and the results travel back. That is a
// scatter and a gather of point-to-point transfers over RCCL (grouped ncclSend / ncclRecv: each xGMI peer link carries its own
// peer's frames, nothing rings through third devices), issued from this one thread against two communicators per device
// (ncclCommInitAll twice: shards out on one, results back on the other, so both directions of a link are busy at once). A shard
// travels in pieces: a device convolves piece c (zg_batch_blur_resize on its compute stream) while piece c + 1 is arriving and
// piece c - 1 is on its way back; the root works on its slice of the caller's buffers in place. Nothing blocks the host until
// the final wait; a failure half-way closes any open RCCL group, drains every stream and marks the context unusable.
//
// librccl is bound at first use with dlopen, not linked: a Python process already carries PyTorch's own copy (same SONAME,
// found first), a bare process gets the system's, and a host that never asks for more than one GPU never loads it.
#include "zg_common.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
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
    // ZIGNAL_HIP_RCCL_LIBRARY names the library to bind instead (tests: tests/c/rccl_double.cpp, a stand-in that turns grouped ncclSend / ncclRecv pairs
    // into event-ordered copies, so that the world > 1 branches of this file run on a box with one GPU); if it is set, nothing else is tried
    if (const char *forced = getenv("ZIGNAL_HIP_RCCL_LIBRARY")) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        ZG_REQUIRE(h != nullptr, ZG_ERR_UNSUPPORTED, "multi-GPU: ZIGNAL_HIP_RCCL_LIBRARY=%s cannot be loaded (%s)", forced, dlerror());
    } else {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
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

constexpr int MAX_CHUNKS = 8;

struct DeviceSlot {
    int device = -1;
    // three streams so that a shard's pieces can arrive, be convolved and travel back at the same time
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    nccl_comm comm_in = nullptr, comm_out = nullptr; // scatter and gather on communicators of their own: the two directions of a link overlap
    hipEvent_t arrived[MAX_CHUNKS] = {}, done[MAX_CHUNKS] = {};
    hipEvent_t t0 = nullptr, t_in = nullptr, t_run0 = nullptr, t_run1 = nullptr; // timing (root: scatter; every device: its kernels)
    void *in = nullptr, *out = nullptr; // shard staging on a non-root device (grow-only)
    size_t in_bytes = 0, out_bytes = 0;
};

struct Multi {
    std::vector<DeviceSlot> dev; // dev[0] is the root
    hipEvent_t ready = nullptr;  // root: whatever produced the caller's frames
    bool loopback = false;       // one device: still pass the root's shard through ncclSend / ncclRecv (diagnostics)
    bool poisoned = false;       // a call failed half-way: streams and communicators are in an unknown state
    bool producer_named = false; // zg_multi_wait_stream was called since the last batch: the caller has said what produces its frames
    int chunks = 4;
};

struct DeviceScope { // the calling thread's current device, restored on exit
    int saved = 0;
    DeviceScope() { (void)hipGetDevice(&saved); }
    ~DeviceScope() { (void)hipSetDevice(saved); }
};

struct GroupScope { // an RCCL group that is closed on every path out of the scope
    bool open = false;
    int begin() {
        const int r = g_rccl.GroupStart();
        open = r == 0;
        return r;
    }
    int end() {
        open = false;
        return g_rccl.GroupEnd();
    }
    ~GroupScope() {
        if (open) (void)g_rccl.GroupEnd();
    }
};

void shard_range(uint32_t n, int i, int world, uint32_t *begin, uint32_t *end) { // as zignal_amd/sharding.py: blocks differ by at most one frame
    const uint32_t base = n / (uint32_t)world, extra = n % (uint32_t)world;
    *begin = (uint32_t)i * base + ((uint32_t)i < extra ? (uint32_t)i : extra);
    *end = *begin + base + ((uint32_t)i < extra ? 1u : 0u);
}

// Frames [*begin, *end) of the batch: piece c of device i's shard, when every shard travels in (up to) `chunks` pieces. Pieces of a shard are
// contiguous, ordered and differ by at most one frame; a shard shorter than `chunks` frames has one frame per piece and empty pieces after them.
void piece_range(uint32_t n_frames, int world, int chunks, int i, int c, uint32_t *begin, uint32_t *end) {
    uint32_t sb, se, cb, ce;
    shard_range(n_frames, i, world, &sb, &se);
    const uint32_t k = se - sb;
    const int pieces = (int)std::min<uint32_t>((uint32_t)chunks, std::max<uint32_t>(k, 1));
    if (c >= pieces || k == 0) { *begin = *end = sb; return; }
    shard_range(k, c, pieces, &cb, &ce);
    *begin = sb + cb;
    *end = sb + ce;
}

// What a device does with a piece of its shard once it has arrived: n frames at `src` -> n frames at `dst`, enqueued on `s` (the device is current).
typedef std::function<int(const void *src, uint32_t n, void *dst, zg_stream s)> PieceOp;

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
    // Tests only (ZIGNAL_HIP_MULTI_VIRTUAL): one device may stand in for several — the same id listed N times makes a context of world N whose shards,
    // staging buffers, streams, events and both "communicators" are all real and all on that device. RCCL itself refuses such a list; the stand-in
    // of ZIGNAL_HIP_RCCL_LIBRARY does not.
    const bool virtual_world = getenv("ZIGNAL_HIP_MULTI_VIRTUAL") != nullptr;
    ZG_REQUIRE(n_devices >= 1 && (n_devices <= visible || (virtual_world && devices && n_devices <= 64)), ZG_ERR_INVALID_ARGUMENT,
               "zg_multi_create: %d devices asked for, %d visible", n_devices, visible);
    std::vector<int> list((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        list[(size_t)i] = devices ? devices[i] : i;
        ZG_REQUIRE(list[(size_t)i] >= 0 && list[(size_t)i] < visible, ZG_ERR_INVALID_ARGUMENT, "zg_multi_create: device %d out of range", list[(size_t)i]);
        for (int j = 0; j < i && !virtual_world; ++j)
            ZG_REQUIRE(list[(size_t)j] != list[(size_t)i], ZG_ERR_INVALID_ARGUMENT, "zg_multi_create: device %d listed twice", list[(size_t)i]);
    }
    DeviceScope scope;
    Multi *m = new Multi();
    m->dev.resize((size_t)n_devices);
    m->loopback = getenv("ZIGNAL_HIP_MULTI_LOOPBACK") != nullptr;
    if (const char *e = getenv("ZIGNAL_HIP_MULTI_CHUNKS")) m->chunks = std::max(1, std::min(MAX_CHUNKS, atoi(e)));
    int rc = ZG_OK;
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == ZG_OK) rc = hip_fail(e, what, __FILE__, __LINE__);
    };
    for (int i = 0; i < n_devices && rc == ZG_OK; ++i) {
        DeviceSlot &d = m->dev[(size_t)i];
        d.device = list[(size_t)i];
        rc = zg_init(d.device); // gfx950 check + current device
        if (rc != ZG_OK) break;
        hip(hipStreamCreateWithFlags(&d.s_in, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        hip(hipStreamCreateWithFlags(&d.s_run, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        hip(hipStreamCreateWithFlags(&d.s_out, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            hip(hipEventCreateWithFlags(&d.arrived[c], hipEventDisableTiming), "hipEventCreateWithFlags");
            hip(hipEventCreateWithFlags(&d.done[c], hipEventDisableTiming), "hipEventCreateWithFlags");
        }
        hip(hipEventCreate(&d.t0), "hipEventCreate");
        hip(hipEventCreate(&d.t_in), "hipEventCreate");
        hip(hipEventCreate(&d.t_run0), "hipEventCreate");
        hip(hipEventCreate(&d.t_run1), "hipEventCreate");
        if (i == 0) hip(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    if (rc == ZG_OK && (n_devices > 1 || m->loopback)) {
        rc = load_rccl();
        for (int pass = 0; pass < 2 && rc == ZG_OK; ++pass) {
            std::vector<nccl_comm> comms((size_t)n_devices, nullptr);
            const int r = g_rccl.CommInitAll(comms.data(), n_devices, list.data());
            if (r != 0) {
                set_error("RCCL error %d (%s) in ncclCommInitAll over %d devices", r, g_rccl.GetErrorString(r), n_devices);
                rc = ZG_ERR_HIP;
            } else {
                for (int i = 0; i < n_devices; ++i) (pass == 0 ? m->dev[(size_t)i].comm_in : m->dev[(size_t)i].comm_out) = comms[(size_t)i];
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
        for (hipStream_t s : {d.s_in, d.s_run, d.s_out})
            if (s) (void)hipStreamSynchronize(s);
        if (d.comm_in) (void)g_rccl.CommDestroy(d.comm_in);
        if (d.comm_out) (void)g_rccl.CommDestroy(d.comm_out);
        if (d.in) (void)hipFree(d.in);
        if (d.out) (void)hipFree(d.out);
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            if (d.arrived[c]) (void)hipEventDestroy(d.arrived[c]);
            if (d.done[c]) (void)hipEventDestroy(d.done[c]);
        }
        for (hipEvent_t e : {d.t0, d.t_in, d.t_run0, d.t_run1})
            if (e) (void)hipEventDestroy(e);
        for (hipStream_t s : {d.s_in, d.s_run, d.s_out})
            if (s) (void)hipStreamDestroy(s);
    }
    if (m->ready) {
        (void)hipSetDevice(m->dev[0].device);
        (void)hipEventDestroy(m->ready);
    }
    delete m;
    return ZG_OK;
}

int zg_multi_device_count(zg_multi handle) { return handle ? (int)((Multi *)handle)->dev.size() : 0; }

int zg_multi_wait_stream(zg_multi handle, zg_stream producer) {
    ZG_REQUIRE(handle != nullptr, ZG_ERR_INVALID_ARGUMENT, "zg_multi_wait_stream: null context");
    Multi *m = (Multi *)handle;
    DeviceScope scope;
    DeviceSlot &root = m->dev[0];
    ZG_HIP(hipSetDevice(root.device));
    ZG_HIP(hipEventRecord(m->ready, as_stream(producer)));
    for (hipStream_t s : {root.s_in, root.s_run, root.s_out}) ZG_HIP(hipStreamWaitEvent(s, m->ready, 0));
    m->producer_named = true;
    return ZG_OK;
}

namespace {

// The whole exchange, asynchronous: nothing here blocks the host. Piece c of device i's shard: root -> i on the scatter communicator
// (streams s_in), the kernel on i's s_run once the piece has arrived, the result i -> root on the gather communicator (streams s_out)
// once the kernel is done. One grouped launch per piece index and direction, so every xGMI link carries its own peer's pieces back to
// back and no transfer rings through a third device. Issue order = piece order on every device, which is what RCCL needs from a single
// thread driving several communicators.
int run_batch(Multi *m, const void *src_root, uint32_t n_frames, size_t in_frame, void *dst_root, size_t out_frame, const PieceOp &op, bool timed) {
    const int world = (int)m->dev.size();
    DeviceSlot &root = m->dev[0];
    const bool loop = world == 1 && m->loopback;
    const int first_peer = loop ? 0 : 1;
    auto piece = [&](int i, int c, uint32_t *b, uint32_t *e) { piece_range(n_frames, world, m->chunks, i, c, b, e); };
    auto staged = [&](int i, uint32_t frame, bool input) -> char * { // where device i keeps `frame` of its shard
        uint32_t sb, se;
        shard_range(n_frames, i, world, &sb, &se);
        DeviceSlot &d = m->dev[(size_t)i];
        return (char *)(input ? d.in : d.out) + (size_t)(frame - sb) * (input ? in_frame : out_frame);
    };

    if (timed) {
        ZG_HIP(hipSetDevice(root.device));
        ZG_HIP(hipEventRecord(root.t0, root.s_in));
    }
    // the root's own frames need no transfer: one launch on its compute stream, in place
    if (!loop) {
        uint32_t b, e;
        shard_range(n_frames, 0, world, &b, &e);
        ZG_HIP(hipSetDevice(root.device));
        if (timed) ZG_HIP(hipEventRecord(root.t_run0, root.s_run));
        if (e > b) {
            const int rc = op((const char *)src_root + (size_t)b * in_frame, e - b, (char *)dst_root + (size_t)b * out_frame, (zg_stream)root.s_run);
            if (rc) return rc;
        }
        if (timed) ZG_HIP(hipEventRecord(root.t_run1, root.s_run));
    }
    for (int c = 0; c < m->chunks; ++c) {
        // ---- scatter piece c ----
        {
            GroupScope g;
            bool any = false;
            for (int i = first_peer; i < world; ++i) {
                uint32_t b, e;
                piece(i, c, &b, &e);
                if (e == b) continue;
                if (!any) { ZG_NCCL(g.begin()); any = true; }
                DeviceSlot &d = m->dev[(size_t)i];
                const size_t bytes = (size_t)(e - b) * in_frame;
                ZG_NCCL(g_rccl.Send((const char *)src_root + (size_t)b * in_frame, bytes, NCCL_UINT8, i, root.comm_in, root.s_in));
                ZG_NCCL(g_rccl.Recv(staged(i, b, true), bytes, NCCL_UINT8, 0, d.comm_in, d.s_in));
            }
            if (any) ZG_NCCL(g.end());
        }
        // ---- compute piece c where it landed ----
        for (int i = first_peer; i < world; ++i) {
            uint32_t b, e;
            piece(i, c, &b, &e);
            if (e == b) continue;
            DeviceSlot &d = m->dev[(size_t)i];
            ZG_HIP(hipSetDevice(d.device));
            ZG_HIP(hipEventRecord(d.arrived[c], d.s_in));
            ZG_HIP(hipStreamWaitEvent(d.s_run, d.arrived[c], 0));
            if (timed && c == 0) ZG_HIP(hipEventRecord(d.t_run0, d.s_run));
            const int rc = op(staged(i, b, true), e - b, staged(i, b, false), (zg_stream)d.s_run);
            if (rc) return rc;
            ZG_HIP(hipEventRecord(d.done[c], d.s_run));
            ZG_HIP(hipStreamWaitEvent(d.s_out, d.done[c], 0));
        }
        // ---- gather piece c ----
        {
            GroupScope g;
            bool any = false;
            for (int i = first_peer; i < world; ++i) {
                uint32_t b, e;
                piece(i, c, &b, &e);
                if (e == b) continue;
                if (!any) { ZG_NCCL(g.begin()); any = true; }
                DeviceSlot &d = m->dev[(size_t)i];
                const size_t bytes = (size_t)(e - b) * out_frame;
                ZG_NCCL(g_rccl.Send(staged(i, b, false), bytes, NCCL_UINT8, 0, d.comm_out, d.s_out));
                ZG_NCCL(g_rccl.Recv((char *)dst_root + (size_t)b * out_frame, bytes, NCCL_UINT8, i, root.comm_out, root.s_out));
            }
            if (any) ZG_NCCL(g.end());
        }
    }
    if (timed) {
        ZG_HIP(hipSetDevice(root.device));
        ZG_HIP(hipEventRecord(root.t_in, root.s_in));
        for (int i = first_peer; i < world; ++i) {
            uint32_t b, e;
            shard_range(n_frames, i, world, &b, &e);
            if (e == b || (i == 0 && !loop)) continue;
            ZG_HIP(hipSetDevice(m->dev[(size_t)i].device));
            ZG_HIP(hipEventRecord(m->dev[(size_t)i].t_run1, m->dev[(size_t)i].s_run));
        }
    }
    return ZG_OK;
}

int sync_all(Multi *m) {
    int rc = ZG_OK;
    for (DeviceSlot &d : m->dev) {
        if (hipSetDevice(d.device) != hipSuccess) { rc = ZG_ERR_HIP; continue; }
        for (hipStream_t s : {d.s_in, d.s_run, d.s_out}) {
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess && rc == ZG_OK) rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
        }
    }
    return rc;
}

} // namespace

namespace {

// The part every batch entry point shares: staging on the owners, the wait for whatever produced the caller's frames, the exchange, the final
// wait, the timings. Synchronous: results are complete on return.
int multi_batch(Multi *m, const void *src_frames_root, uint32_t n_frames, size_t in_frame, void *dst_frames_root, size_t out_frame, const PieceOp &op,
                float times_ms[3]) {
    // The caller's statement about its producer stream(s) covers exactly ONE batch call: take it and clear it first, so that a call that fails
    // its checks below cannot leave it behind for a later batch whose frames come from some other stream.
    const bool producer_named = m->producer_named;
    m->producer_named = false;
    ZG_REQUIRE(!m->poisoned, ZG_ERR_INVALID_ARGUMENT, "zg_multi: an earlier call on this context failed half-way; destroy it and create a new one");
    const int world = (int)m->dev.size();
    DeviceScope scope;
    DeviceSlot &root = m->dev[0];
    const double wall0 = now_ms();

    // shard staging on the owners (before anything is enqueued: a failure here leaves the context usable)
    for (int i = 1; i < world; ++i) {
        uint32_t b, e;
        shard_range(n_frames, i, world, &b, &e);
        ZG_HIP(hipSetDevice(m->dev[(size_t)i].device));
        int rc;
        if ((rc = ensure(&m->dev[(size_t)i].in, &m->dev[(size_t)i].in_bytes, (size_t)(e - b) * in_frame))) return rc;
        if ((rc = ensure(&m->dev[(size_t)i].out, &m->dev[(size_t)i].out_bytes, (size_t)(e - b) * out_frame))) return rc;
    }
    if (world == 1 && m->loopback) { // one-device diagnostics: the root's shard makes the round trip through both communicators
        ZG_HIP(hipSetDevice(root.device));
        int rc;
        if ((rc = ensure(&root.in, &root.in_bytes, (size_t)n_frames * in_frame))) return rc;
        if ((rc = ensure(&root.out, &root.out_bytes, (size_t)n_frames * out_frame))) return rc;
    }
    // Whatever produced the caller's frames. A caller that named its producer stream(s) with zg_multi_wait_stream since the last batch has
    // already ordered the three root streams behind them: nothing more to do, and no host synchronisation. Otherwise the producer is unknown —
    // it may be a non-blocking stream (this library's own zg_stream_create makes those, and so do PyTorch's side streams), which an event on
    // the legacy default stream would NOT order behind — so the root device is synchronised: the documented "synchronous call" stays safe
    // whatever stream filled src_frames_root.
    if (!producer_named) {
        ZG_HIP(hipSetDevice(root.device));
        ZG_HIP(hipDeviceSynchronize());
    }
    int rc = run_batch(m, src_frames_root, n_frames, in_frame, dst_frames_root, out_frame, op, times_ms != nullptr);
    const int src = sync_all(m); // results are complete on return; after a failure this also drains what was already enqueued
    if (rc != ZG_OK || src != ZG_OK) {
        m->poisoned = true; // shards may be half-way: no later call may trust the streams or the communicators
        return rc != ZG_OK ? rc : src;
    }
    if (times_ms) {
        float ms = 0;
        (void)hipSetDevice(root.device);
        if (hipEventElapsedTime(&ms, root.t0, root.t_in) == hipSuccess) times_ms[0] = ms; // the scatter stream, first send to last send
        for (int i = 0; i < world; ++i) {
            uint32_t b, e;
            shard_range(n_frames, i, world, &b, &e);
            if (e == b) continue;
            (void)hipSetDevice(m->dev[(size_t)i].device);
            if (hipEventElapsedTime(&ms, m->dev[(size_t)i].t_run0, m->dev[(size_t)i].t_run1) == hipSuccess) times_ms[1] = std::max(times_ms[1], ms);
        }
        times_ms[2] = (float)(now_ms() - wall0);
    }
    return ZG_OK;
}

} // namespace

int zg_multi_batch_blur_resize(zg_multi handle, const void *src_frames_root, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, float sigma,
                               void *dst_frames_root, uint32_t out_rows, uint32_t out_cols, const zg_method *method, float times_ms[3]) {
    ZG_REQUIRE(handle != nullptr, ZG_ERR_INVALID_ARGUMENT, "zg_multi_batch_blur_resize: null context");
    Multi *m = (Multi *)handle;
    if (times_ms) times_ms[0] = times_ms[1] = times_ms[2] = 0.0f;
    const bool named = m->producer_named; // an argument error or an empty batch consumes the caller's producer statement too
    m->producer_named = false;
    ZG_REQUIRE(pixel_valid(pixel), ZG_ERR_INVALID_ARGUMENT, "batch: invalid pixel type %d", pixel);
    ZG_REQUIRE(method != nullptr, ZG_ERR_INVALID_ARGUMENT, "batch: null method");
    if (n_frames == 0 || rows == 0 || cols == 0 || out_rows == 0 || out_cols == 0) return ZG_OK;
    ZG_REQUIRE(src_frames_root && dst_frames_root, ZG_ERR_INVALID_ARGUMENT, "batch: null frame pointer");
    m->producer_named = named;
    const size_t ps = pixel_size(pixel);
    const zg_method mt = *method;
    return multi_batch(m, src_frames_root, n_frames, (size_t)rows * cols * ps, dst_frames_root, (size_t)out_rows * out_cols * ps,
                       [=](const void *src, uint32_t n, void *dst, zg_stream s) {
                           return zg_batch_blur_resize(src, n, rows, cols, pixel, sigma, dst, out_rows, out_cols, &mt, s);
                       },
                       times_ms);
}

// zg_batch_pipeline over the context's devices: any recipe, not just [blur, resize] (the reference's unit of batch work is "every
// image through the recipe's steps", src/cli/pipeline.zig:153-179).
int zg_multi_batch_pipeline(zg_multi handle, const void *src_frames_root, uint32_t n_frames, uint32_t rows, uint32_t cols, int pixel, int space,
                            const zg_step *steps, uint32_t n_steps, void *dst_frames_root, float times_ms[3]) {
    ZG_REQUIRE(handle != nullptr, ZG_ERR_INVALID_ARGUMENT, "zg_multi_batch_pipeline: null context");
    Multi *m = (Multi *)handle;
    if (times_ms) times_ms[0] = times_ms[1] = times_ms[2] = 0.0f;
    const bool named = m->producer_named;
    m->producer_named = false;
    uint32_t out_rows = 0, out_cols = 0;
    int out_pixel = 0, out_space = 0;
    if (int rc = zg_batch_pipeline_shape(rows, cols, pixel, space, steps, n_steps, &out_rows, &out_cols, &out_pixel, &out_space)) return rc; // validates the recipe
    if (n_frames == 0 || rows == 0 || cols == 0) return ZG_OK;
    ZG_REQUIRE(src_frames_root && dst_frames_root, ZG_ERR_INVALID_ARGUMENT, "batch: null frame pointer");
    m->producer_named = named;
    const std::vector<zg_step> recipe(steps, steps + n_steps); // the caller's array need not outlive the call's set-up
    return multi_batch(m, src_frames_root, n_frames, (size_t)rows * cols * pixel_size(pixel), dst_frames_root, (size_t)out_rows * out_cols * pixel_size(out_pixel),
                       [&](const void *src, uint32_t n, void *dst, zg_stream s) {
                           return zg_batch_pipeline(src, n, rows, cols, pixel, space, recipe.data(), (uint32_t)recipe.size(), dst, s);
                       },
                       times_ms);
}

// Host only: the frames of piece `piece` of device `device`'s shard when n_frames frames go to `world` devices in up to `chunks` pieces per shard
// (the context's own arithmetic; zignal_amd/sharding.py cuts the same way). A CPU test holds the two to each other and to the partition
// properties (disjoint, covering, ordered, sizes within one frame) without a GPU.
int zg_multi_piece_range(uint32_t n_frames, int world, int chunks, int device, int piece, uint32_t *begin, uint32_t *end) {
    ZG_REQUIRE(begin && end, ZG_ERR_INVALID_ARGUMENT, "zg_multi_piece_range: null output");
    ZG_REQUIRE(world >= 1 && device >= 0 && device < world && chunks >= 1 && chunks <= MAX_CHUNKS && piece >= 0, ZG_ERR_INVALID_ARGUMENT,
               "zg_multi_piece_range: world %d, device %d, chunks %d (1..%d), piece %d", world, device, chunks, MAX_CHUNKS, piece);
    piece_range(n_frames, world, chunks, device, piece, begin, end);
    return ZG_OK;
}

} // extern "C"
