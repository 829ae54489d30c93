// rccl_double.cpp — a stand-in for librccl, for tests: zg_multi.cpp binds it instead of the real library when
// ZIGNAL_HIP_RCCL_LIBRARY names it, so that its world > 1 branches (staging offsets, piece arithmetic, event ordering, groups closed on
// every path, the poisoned context after a failure) execute on a box with ONE GPU, where every "device" of the context is the same one.
//
// Only what zg_multi.cpp uses, with RCCL's semantics for a single thread that drives several communicators:
//   ncclCommInitAll      n communicators of one world (rank i on devices[i]; the same device may appear any number of times)
//   ncclGroupStart / End calls between them are queued; the outermost End matches every send (rank a -> peer b) with the first unmatched receive
//                        of that world on rank b from peer a, in issue order, and turns the pair into: an event on the sender's stream, the
//                        receiver's stream waits for it, a device-to-device copy on the receiver's stream, an event behind it that the sender's
//                        stream waits for (a send buffer may be reused once the send's stream has passed it)
//   ncclSend / ncclRecv  outside a group: queued and flushed at once (zg_multi.cpp never does that)
// Failure injection: RCCL_DOUBLE_FAIL_SEND=k makes the k-th ncclSend of the process (1-based, counted from the last rccl_double_reset) fail with
// ncclUnhandledCudaError before it is queued; what the caller's group has queued so far is dropped at its GroupEnd with ncclInvalidUsage.
// rccl_double_stats reports what went through, so a test can tell that the exchange really ran here.
// build: hipcc -shared -fPIC -O1 -o librccl_double.so rccl_double.cpp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <vector>

namespace {
struct World;
struct Comm { World *world; int rank, device; };
struct World { std::vector<Comm *> comms; int alive; };
struct Op { bool send; const void *src; void *dst; size_t bytes; int peer; Comm *comm; hipStream_t stream; bool matched; };
int g_depth = 0;
bool g_group_failed = false;
std::vector<Op> g_pending;
uint64_t g_stats[5] = {0, 0, 0, 0, 0}; // sends, receives, bytes copied, groups flushed, pairs matched
uint64_t g_send_calls = 0;

size_t type_size(int t) { // ncclDataType_t
    switch (t) {
    case 0: case 1: return 1;             // int8, uint8
    case 2: case 3: case 7: return 4;     // int32, uint32, float32
    case 4: case 5: case 8: return 8;     // int64, uint64, float64
    case 6: case 9: return 2;             // float16, bfloat16
    default: return 0;
    }
}

int flush() {
    int saved = 0;
    (void)hipGetDevice(&saved);
    int rc = 0;
    if (g_group_failed) rc = 5; // ncclInvalidUsage: a call of the group failed
    for (size_t i = 0; i < g_pending.size() && rc == 0; ++i) {
        Op &s = g_pending[i];
        if (!s.send) continue;
        Op *r = nullptr;
        for (Op &c : g_pending)
            if (!c.send && !c.matched && c.comm->world == s.comm->world && c.comm->rank == s.peer && c.peer == s.comm->rank) { r = &c; break; }
        if (!r || r->bytes != s.bytes) { rc = 5; break; } // a send without its receive would hang real RCCL
        r->matched = s.matched = true;
        hipEvent_t sent = nullptr, copied = nullptr;
        if (hipSetDevice(s.comm->device) != hipSuccess || hipEventCreateWithFlags(&sent, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(sent, s.stream) != hipSuccess) { rc = 1; break; }
        if (hipSetDevice(r->comm->device) != hipSuccess || hipStreamWaitEvent(r->stream, sent, 0) != hipSuccess ||
            hipMemcpyAsync(r->dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r->stream) != hipSuccess ||
            hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess || hipEventRecord(copied, r->stream) != hipSuccess) { rc = 1; break; }
        if (hipSetDevice(s.comm->device) != hipSuccess || hipStreamWaitEvent(s.stream, copied, 0) != hipSuccess) { rc = 1; break; }
        (void)hipEventDestroy(sent);   // released once it has completed
        (void)hipEventDestroy(copied);
        g_stats[2] += s.bytes;
        g_stats[4] += 1;
    }
    for (const Op &o : g_pending)
        if (rc == 0 && !o.matched) rc = 5; // a receive nobody sends to
    g_pending.clear();
    g_group_failed = false;
    g_stats[3] += 1;
    (void)hipSetDevice(saved);
    return rc;
}
} // namespace

extern "C" {

int ncclCommInitAll(void **comms, int n, const int *devices) {
    if (!comms || n < 1) return 4; // ncclInvalidArgument
    World *w = new World();
    w->alive = n;
    for (int i = 0; i < n; ++i) {
        Comm *c = new Comm{w, i, devices ? devices[i] : i};
        w->comms.push_back(c);
        comms[i] = c;
    }
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return 4;
    World *w = c->world;
    delete c;
    if (--w->alive == 0) delete w;
    return 0;
}

int ncclGroupStart() { ++g_depth; return 0; }

int ncclGroupEnd() {
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    return flush();
}

int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    ++g_send_calls;
    if (const char *e = getenv("RCCL_DOUBLE_FAIL_SEND")) {
        if ((uint64_t)atoll(e) == g_send_calls) { g_group_failed = g_depth > 0; return 1; } // ncclUnhandledCudaError
    }
    const size_t ts = type_size(type);
    if (!c || !buf || ts == 0 || peer < 0 || peer >= (int)c->world->comms.size()) return 4;
    g_pending.push_back(Op{true, buf, nullptr, count * ts, peer, c, stream, false});
    g_stats[0] += 1;
    return g_depth > 0 ? 0 : flush();
}

int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    const size_t ts = type_size(type);
    if (!c || !buf || ts == 0 || peer < 0 || peer >= (int)c->world->comms.size()) return 4;
    g_pending.push_back(Op{false, nullptr, buf, count * ts, peer, c, stream, false});
    g_stats[1] += 1;
    return g_depth > 0 ? 0 : flush();
}

const char *ncclGetErrorString(int r) {
    switch (r) {
    case 0: return "no error";
    case 1: return "unhandled cuda error (rccl_double: injected or a HIP call failed)";
    case 4: return "invalid argument";
    case 5: return "invalid usage (rccl_double: unmatched send / receive, or a call of the group failed)";
    default: return "rccl_double: unknown error";
    }
}

// test hooks (not part of RCCL)
void rccl_double_stats(uint64_t out[5]) { for (int i = 0; i < 5; ++i) out[i] = g_stats[i]; }
void rccl_double_reset() { for (uint64_t &v : g_stats) v = 0; g_send_calls = 0; }

} // extern "C"
