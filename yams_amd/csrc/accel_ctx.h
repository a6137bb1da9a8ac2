// accel_ctx.h — the context object behind yams_accel_ctx (host side, C++17).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/yams_mi355x_accel.h"

// Orders the filter sweeps of the contexts that share it (see yams_accel_gate_create).
struct yams_accel_gate {
    int device = 0;
    std::mutex mu;
    std::condition_variable cv;
    hipEvent_t last = nullptr; // end of the most recently enqueued sweep (or of what a holder put behind it)
    bool armed = false;
    yams_accel_ctx* holder = nullptr; // a context that keeps the gate closed behind its sweep (yams_accel_ctx_set_sweep_hold)
};

struct yams_accel_ctx {
    int device = 0;
    yams_accel_gate* gate = nullptr;
    // Called on the host right before a filter sweep is enqueued on `stream` (sharded_api.cpp: the fence that keeps
    // a shard's next sweep behind its previous exchange).  May block; must not throw.
    std::function<void(hipStream_t)> before_sweep;
    bool sweep_hold = false; // keep the gate closed behind this context's sweeps until yams_accel_ctx_release_sweep_hold
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipStream_t aux_stream = nullptr;   // high-priority side stream (whole-blob digest chains)
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    std::string last_error;

    // growable named device buffers (workspace); never shrinks
    struct Buf { void* p = nullptr; size_t cap = 0; };
    std::map<std::string, Buf> bufs;
    // name prefix of ws_get: a nested scan (the split-filter escalation of a batch) works in its
    // own namespace, so the buffers of the call that is still in flight around it stay intact
    std::string ws_ns;
    // What this context has learnt about a corpus's first filter tier (scan_api.cpp, "tier hint"): the int8 tier's bound is as
    // wide as the shadow's quantisation residue — on strongly anisotropic rows (a few large components, a long tail of small
    // ones) five times wider than on isotropic ones, and every query then fails its proof and pays a split-bf16 sweep on top.
    // Keyed by the int8 shadow's address; `bf16_first` batches start on the bf16 tier, every 256th one probes int8 again.
    // depth: what the int8 tier's batches on this corpus needed — 0: the plan's stage 1 (3k + 64 candidates) proves them;
    // 1: most proofs needed the whole list (stage 1 re-scores all of it at once); 2: and many lists were too short (deeper lists)
    // (depth per tier — [0] the int8 tier, [1] the single-pass bf16 tier: the same corpus may crowd one bound and not the other)
    struct TierHint { uint64_t n_rows = 0; bool bf16_first = false; uint32_t served = 0; uint8_t depth[2] = {0, 0}; uint32_t served_deep[2] = {0, 0}; };
    std::map<const void*, TierHint> tier_hints;
    uint32_t emu_calls = 0; // measurement build: batches this context has served (emulation knobs of scan_api.cpp)
    // pinned host staging
    void* pinned = nullptr;
    size_t pinned_cap = 0;

    // optional HIP-event timing of the dominant kernels
    bool timing = false;
    struct Span { hipEvent_t a, b; };
    std::map<std::string, std::vector<Span>> spans;
    std::vector<hipEvent_t> event_pool;

    // where the last yams_ingest_host call spent its set-up and tear-down (device_info_json: "last_host_ingest")
    struct HostIngestStats { double alloc_ms = 0, release_ms = 0, total_ms = 0; uint64_t batch_bytes = 0, bytes = 0; uint32_t batches = 0, slots = 0; } host_ingest;
    // last ingest result (device arrays live in bufs)
    yams_ingest_result_t ingest{};
};

namespace yams_accel {

yams_status_t fail(yams_accel_ctx* ctx, yams_status_t st, const std::string& msg);
yams_status_t hip_fail(yams_accel_ctx* ctx, hipError_t e, const char* what);

// Workspace: returns a device pointer of at least `bytes` (contents undefined).
yams_status_t ws_get(yams_accel_ctx* ctx, const char* name, size_t bytes, void** out);
// Frees every workspace buffer of the context larger than `keep_bytes` (after a synchronisation of its stream); returns
// the bytes given back.  For calls whose buffers scale with the CALL (GiB-sized ingest batches), not with the device.
size_t ws_trim(yams_accel_ctx* ctx, size_t keep_bytes);
yams_status_t pinned_get(yams_accel_ctx* ctx, size_t bytes, void** out);

// Host -> device copy of `bytes` at `src` (any host memory) to `dst` on `stream`, ordered on the stream like
// hipMemcpyAsync (the caller synchronises the stream before it reads `dst` from another one).  Large pageable sources:
// the call RETURNS ONLY AFTER the data has left host memory (the source may be reused at once; the host is blocked for
// the length of the transfer) and uploads of one process take turns on one ring — in exchange they run at the link's
// rate from PAGEABLE memory too: the runtime's own staging of a
// pageable source is one thread and one bounce buffer (measured: 6.2 GB/s for a 38 GB corpus upload), this one fills a
// ring of pinned buffers with several threads while the previous buffer is on its way.  Pinned / registered sources and
// small copies go straight to hipMemcpyAsync.  `dst` must lie inside ONE allocation (callers split at chunk borders).
hipError_t staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t stream);

// Brackets a filter sweep: the stream waits for the previous sweep of the gate, the sweep's end becomes the
// gate's new tail.  The gate's mutex is held from enter() to leave() (launches only, no host waits).
struct GatedSweep {
    yams_accel_ctx* ctx; hipStream_t st; std::unique_lock<std::mutex> lk;
    GatedSweep(yams_accel_ctx* c, hipStream_t s) : ctx(c), st(s) {
        if (ctx->before_sweep) ctx->before_sweep(st); // (before the gate's mutex: it may wait for another lane)
        if (!ctx->gate) return;
        lk = std::unique_lock<std::mutex>(ctx->gate->mu);
        // another context holds the gate closed behind its sweep (its collective has not been enqueued yet)
        ctx->gate->cv.wait(lk, [&] { return ctx->gate->holder == nullptr || ctx->gate->holder == ctx; });
        if (ctx->gate->armed) (void)hipStreamWaitEvent(st, ctx->gate->last, 0);
    }
    void leave() {
        if (!lk.owns_lock()) return;
        (void)hipEventRecord(ctx->gate->last, st);
        ctx->gate->armed = true;
        if (ctx->sweep_hold) ctx->gate->holder = ctx;
        lk.unlock();
    }
    ~GatedSweep() { leave(); }
};

struct TimedRegion { // RAII-less helper: begin/end record events when timing is enabled
    yams_accel_ctx* ctx; const char* name; hipEvent_t a = nullptr, b = nullptr;
    TimedRegion(yams_accel_ctx* c, const char* n, hipStream_t on = nullptr);
    hipStream_t stream = nullptr;
    void end();
};

// Allocation fault injection (yams_accel_debug_fail_alloc_after): every allocation of device, pinned or VMM-backed memory
// this library makes goes through one of the three doors below, which fail with hipErrorOutOfMemory once armed.
#ifdef YAMS_ACCEL_MEASURE
bool alloc_fault();
#else
constexpr bool alloc_fault() { return false; }   // (the product build: no injection, no atomic load on the allocation paths)
#endif
size_t big_trim(int device);
inline hipError_t ya_malloc(void** p, size_t bytes) {
    if (alloc_fault()) { *p = nullptr; return hipErrorOutOfMemory; }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && big_trim(-1) > 0) { (void)hipGetLastError(); e = hipMalloc(p, bytes); } // the pool's memory first
    return e;
}
inline hipError_t ya_host_malloc(void** p, size_t bytes, unsigned flags) {
    if (alloc_fault()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(p, bytes, flags);
}
inline hipError_t ya_mem_create(hipMemGenericAllocationHandle_t* h, size_t bytes, const hipMemAllocationProp* prop) {
    if (alloc_fault()) return hipErrorOutOfMemory;
    hipError_t e = hipMemCreate(h, bytes, prop, 0);
    if (e == hipErrorOutOfMemory && big_trim(-1) > 0) { (void)hipGetLastError(); e = hipMemCreate(h, bytes, prop, 0); }
    return e;
}

// Call-sized device buffers (the slot buffers of yams_ingest_host: up to four of 8 GiB) come from a process-wide, per-device
// POOL instead of being allocated and freed by every call: allocating 32 GiB costs the driver hundreds of milliseconds to
// seconds (round 5: the same 32 GiB stream measured 9 and 46 GB/s), and a context must not keep them either (one such share
// per pooled context next to mirrors sized as shares of the device).  big_take returns a cached buffer of at least `bytes`
// (at most twice that) or allocates one; big_give puts it back, unless the pool already holds kBigPoolMax bytes.  The pool
// gives its memory back when an allocation of this library fails (ya_malloc and ya_mem_create retry once after a trim) and
// when the host says so (yams_accel_trim).
constexpr size_t kBigPoolMax = 40ull << 30;
hipError_t big_take(int device, size_t bytes, void** p, size_t* cap);
void big_give(int device, void* p, size_t cap);
size_t big_trim(int device); // device < 0: every device; returns the bytes freed
size_t big_held(int device); // bytes the pool holds for this device (free memory as far as a caller sizing its batches is concerned)

#define YA_HIP(ctx, expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) \
    return ::yams_accel::hip_fail((ctx), e__, #expr); } while (0)
#define YA_TRY(expr) do { yams_status_t s__ = (expr); if (s__ != YAMS_OK) return s__; } while (0)

} // namespace yams_accel
