// stub_coll.cpp — TEST INFRASTRUCTURE, not product: a stand-in for the five RCCL entry points the sharded search
// binds with dlopen (yams_amd/csrc/sharded_api.cpp: ncclGetVersion / ncclCommInitAll / ncclAllGather /
// ncclCommDestroy / ncclGetErrorString), with which the kRccl code path — turnstile, per-rank worker threads,
// empty-record participation of a failed rank, the exchange fence, destroy with batches in flight — can run with
// 2, 4 or 8 ranks on a box that has ONE GPU.  RCCL itself refuses two ranks on one device; this library does not
// care where its ranks live.  It is reached only through `yams_scan_sharded_options_t.rccl_library` (or the
// plugin's "rccl_library" config key); nothing under yams_amd/ or include/ refers to it.
//
// Semantics kept from the real thing, because they are what the product's ordering logic has to get right:
//   * an all-gather is ONE collective of the communicator: call number c of rank i pairs with call number c of
//     every other rank.  Ranks that arrive with different byte counts (= different batches: the records of two
//     batches differ in size whenever their query counts do) get ncclInvalidArgument — an ordering bug in the
//     caller shows up as an error (or, with a missing rank, as a hang that the tests' timeouts catch), never as
//     silently mixed-up data;
//   * stream order: rank i's receive buffer is complete, on its stream, only after every rank's send buffer was
//     ready on ITS stream; and no rank's stream proceeds past the collective before every peer has read its send
//     buffer (the product reuses the send buffer for the lane's next batch);
//   * the host call returns once the work is enqueued (after a host-side rendezvous of the n ranks: every rank has
//     its own worker thread in the product, as in any thread-per-rank use of RCCL).
// Fault injection for the tests (environment, read per communicator at ncclCommInitAll):
//   YAMS_STUB_COLL_FAIL_AT=c     collective number c (0-based) moves no data and returns ncclSystemError on every rank.
//   YAMS_STUB_COLL_STALL_AT=c    collective number c never completes on any rank's stream (a host function parked on the
//                                stream, as a collective kernel spins when a peer never joins) — until ncclCommAbort
//                                or ncclCommDestroy, or two minutes, whichever comes first.
// Also exported, optional for the caller: ncclCommCount, ncclCommAbort (RCCL's signatures).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <vector>

extern "C" {
typedef struct stubComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t; // (only byte types are used by the caller; sizes below)
}

namespace {

struct Group {
    int n = 0;
    std::vector<int> device;
    std::mutex mu;
    std::condition_variable cv;
    // rendezvous of the current collective
    int arrived = 0; uint64_t generation = 0;
    std::vector<const void*> send; std::vector<void*> recv; std::vector<size_t> bytes; std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> ready, done;   // per rank
    bool mismatch = false;
    uint64_t calls = 0;                    // collectives completed
    long fail_at = -1, stall_at = -1;
    int alive = 0;
    bool released = false;                 // abort / destroy: parked streams go on
    int parked = 0;                        // host functions still inside park()
};

// the stalled collective: parked on the stream until the communicator is aborted or destroyed
void park(void* p) {
    Group* g = static_cast<Group*>(p);
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->released; });
    --g->parked;
    g->cv.notify_all();
}

void release_parked(Group* g, std::unique_lock<std::mutex>& lk) {
    g->released = true;
    g->cv.notify_all();
    g->cv.wait_for(lk, std::chrono::seconds(10), [&] { return g->parked == 0; });
}

// all n ranks meet here; returns after the last one has arrived (classic generation barrier)
void barrier(Group& g, std::unique_lock<std::mutex>& lk) {
    const uint64_t gen = g.generation;
    if (++g.arrived == g.n) { g.arrived = 0; ++g.generation; g.cv.notify_all(); }
    else g.cv.wait(lk, [&] { return g.generation != gen; });
}

size_t type_bytes(ncclDataType_t t) { return (t == ncclInt8 || t == ncclUint8) ? 1u : 0u; }

} // namespace

struct stubComm { Group* g; int rank; };

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetVersion(int* v) {
    if (!v) return ncclInvalidArgument;
    *v = 9900001; // recognisable: no RCCL release carries it
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "stub: unhandled HIP error";
        case ncclSystemError: return "stub: injected system error";
        case ncclInvalidArgument: return "stub: ranks disagree on the collective (byte counts differ)";
        case ncclInvalidUsage: return "stub: invalid usage";
        default: return "stub: internal error";
    }
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return ncclUnhandledCudaError;
    auto* g = new Group();
    g->n = ndev; g->alive = ndev;
    g->device.resize(ndev);
    g->send.assign(ndev, nullptr); g->recv.assign(ndev, nullptr); g->bytes.assign(ndev, 0); g->stream.assign(ndev, nullptr);
    g->ready.assign(ndev, nullptr); g->done.assign(ndev, nullptr);
    if (const char* f = std::getenv("YAMS_STUB_COLL_FAIL_AT")) g->fail_at = std::atol(f);
    if (const char* f = std::getenv("YAMS_STUB_COLL_STALL_AT")) g->stall_at = std::atol(f);
    int before = 0;
    (void)hipGetDevice(&before);
    for (int i = 0; i < ndev; ++i) {
        g->device[i] = devlist ? devlist[i] : i;
        if (g->device[i] < 0 || g->device[i] >= have) { delete g; return ncclInvalidArgument; }
        (void)hipSetDevice(g->device[i]);
        if (hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming) != hipSuccess) { (void)hipSetDevice(before); return ncclUnhandledCudaError; }
    }
    (void)hipSetDevice(before);
    for (int i = 0; i < ndev; ++i) comms[i] = new stubComm{g, i};
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    Group* g = comm->g;
    bool last;
    { std::unique_lock<std::mutex> lk(g->mu); release_parked(g, lk); last = --g->alive == 0; }
    delete comm;
    if (last) {
        for (int i = 0; i < g->n; ++i) {
            (void)hipSetDevice(g->device[i]);
            if (g->ready[i]) (void)hipEventDestroy(g->ready[i]);
            if (g->done[i]) (void)hipEventDestroy(g->done[i]);
        }
        delete g;
    }
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->g->n;
    return ncclSuccess;
}

// RCCL's abort frees the communicator as well: so does this one
__attribute__((visibility("default"))) ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                                                                  ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff || type_bytes(datatype) == 0) return ncclInvalidArgument;
    Group& g = *comm->g;
    const int me = comm->rank;
    const size_t bytes = sendcount * type_bytes(datatype);
    std::unique_lock<std::mutex> lk(g.mu);
    // phase 1: publish this rank's part; its send buffer is ready when `ready` fires on its stream
    g.send[me] = sendbuff; g.recv[me] = recvbuff; g.bytes[me] = bytes; g.stream[me] = stream;
    if (hipEventRecord(g.ready[me], stream) != hipSuccess) { (void)hipGetLastError(); g.mismatch = true; }
    barrier(g, lk);
    const uint64_t call = g.calls; // (stable until the last barrier below)
    bool bad = g.mismatch;
    for (int j = 0; j < g.n; ++j) bad |= g.bytes[j] != bytes;
    const bool inject = g.fail_at >= 0 && static_cast<uint64_t>(g.fail_at) == call;
    if (g.stall_at >= 0 && static_cast<uint64_t>(g.stall_at) == call && !g.released) {
        ++g.parked;
        if (hipLaunchHostFunc(stream, park, &g) != hipSuccess) { (void)hipGetLastError(); --g.parked; }
    }
    // phase 2: every rank copies every part into its own receive buffer, on its own stream
    if (!bad && !inject) {
        lk.unlock();
        for (int j = 0; j < g.n; ++j) {
            if (hipStreamWaitEvent(stream, g.ready[j], 0) != hipSuccess) { (void)hipGetLastError(); bad = true; }
            unsigned char* dst = static_cast<unsigned char*>(recvbuff) + static_cast<size_t>(j) * bytes;
            hipError_t e = g.device[j] == g.device[me]
                               ? hipMemcpyAsync(dst, g.send[j], bytes, hipMemcpyDeviceToDevice, stream)
                               : hipMemcpyPeerAsync(dst, g.device[me], g.send[j], g.device[j], bytes, stream);
            if (e != hipSuccess) { (void)hipGetLastError(); bad = true; }
        }
        if (hipEventRecord(g.done[me], stream) != hipSuccess) { (void)hipGetLastError(); bad = true; }
        lk.lock();
        if (bad) g.mismatch = true;
    }
    barrier(g, lk);
    bad |= g.mismatch;
    // phase 3: nobody's stream leaves the collective before every peer has read its send buffer
    if (!bad && !inject) {
        lk.unlock();
        for (int j = 0; j < g.n; ++j)
            if (j != me && hipStreamWaitEvent(stream, g.done[j], 0) != hipSuccess) { (void)hipGetLastError(); }
        lk.lock();
    }
    barrier(g, lk);
    if (me == 0) { ++g.calls; g.mismatch = false; }
    barrier(g, lk); // (the next collective's phase 1 must not see this one's state)
    if (inject) return ncclSystemError;
    return bad ? ncclInvalidArgument : ncclSuccess;
}

} // extern "C"
