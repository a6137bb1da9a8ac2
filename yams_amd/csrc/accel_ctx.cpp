// accel_ctx.cpp — context lifecycle, workspace, timing, memory helpers of the C ABI.
#include "accel_ctx.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

namespace yams_accel {

yams_status_t fail(yams_accel_ctx* ctx, yams_status_t st, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return st;
}

yams_status_t hip_fail(yams_accel_ctx* ctx, hipError_t e, const char* what) {
    std::string m = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) +
                    ") in " + what;
    (void)hipGetLastError(); // clear the sticky error
    // out-of-memory is the only error a caller can act on (ErrorCode::ResourceExhausted, core/types.h:49 of the
    // reference); everything else is internal
    return fail(ctx, e == hipErrorOutOfMemory ? YAMS_ERR_RESOURCE_EXHAUSTED : YAMS_ERR_INTERNAL, m);
}

#ifdef YAMS_ACCEL_MEASURE   // allocation-failure injection exists in the measurement build only (the product's doors are inert)
namespace {
std::atomic<long long> g_fail_after{-1};     // allocations that may still succeed; < 0: injection off
std::atomic<unsigned long long> g_faults{0}; // allocations failed by injection
} // namespace
bool alloc_fault() {
    long long v = g_fail_after.load(std::memory_order_relaxed);
    while (v >= 0) {
        if (v == 0) { ++g_faults; return true; }
        if (g_fail_after.compare_exchange_weak(v, v - 1)) return false;
    }
    return false;
}
#endif

namespace {
struct BigBuf { void* p; size_t cap; int device; };
std::mutex g_big_mu;
std::vector<BigBuf> g_big;      // cached, unused
}
hipError_t big_take(int device, size_t bytes, void** p, size_t* cap) {
    {
        std::lock_guard<std::mutex> lk(g_big_mu);
        int best = -1;
        for (size_t i = 0; i < g_big.size(); ++i)
            if (g_big[i].device == device && g_big[i].cap >= bytes && g_big[i].cap <= 2 * bytes + (64u << 20) &&
                (best < 0 || g_big[i].cap < g_big[static_cast<size_t>(best)].cap)) best = static_cast<int>(i);
        if (best >= 0) {
            *p = g_big[static_cast<size_t>(best)].p; *cap = g_big[static_cast<size_t>(best)].cap;
            g_big.erase(g_big.begin() + best);
            return hipSuccess;
        }
    }
    int before = device;
    (void)hipGetDevice(&before);
    (void)hipSetDevice(device);
    const hipError_t e = ya_malloc(p, bytes);
    (void)hipSetDevice(before);
    if (e != hipSuccess) { *p = nullptr; *cap = 0; return e; }
    *cap = bytes;
    return hipSuccess;
}
void big_give(int device, void* p, size_t cap) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_big_mu);
        size_t held = 0;
        for (const BigBuf& b : g_big) held += b.cap;
        if (held + cap <= kBigPoolMax) { g_big.push_back({p, cap, device}); return; }
    }
    int before = device;
    (void)hipGetDevice(&before);
    (void)hipSetDevice(device);
    if (hipFree(p) != hipSuccess) (void)hipGetLastError();
    (void)hipSetDevice(before);
}
size_t big_held(int device) {
    std::lock_guard<std::mutex> lk(g_big_mu);
    size_t held = 0;
    for (const BigBuf& b : g_big) if (b.device == device) held += b.cap;
    return held;
}
size_t big_trim(int device) {
    std::vector<BigBuf> drop;
    {
        std::lock_guard<std::mutex> lk(g_big_mu);
        for (size_t i = 0; i < g_big.size();) {
            if (device < 0 || g_big[i].device == device) { drop.push_back(g_big[i]); g_big.erase(g_big.begin() + static_cast<long>(i)); }
            else ++i;
        }
    }
    size_t freed = 0;
    int before = 0;
    (void)hipGetDevice(&before);
    for (const BigBuf& b : drop) {
        (void)hipSetDevice(b.device);
        if (hipFree(b.p) != hipSuccess) (void)hipGetLastError();
        freed += b.cap;
    }
    (void)hipSetDevice(before);
    return freed;
}

yams_status_t ws_get(yams_accel_ctx* ctx, const char* name, size_t bytes, void** out) {
    auto& b = ctx->ws_ns.empty() ? ctx->bufs[name] : ctx->bufs[ctx->ws_ns + name];
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        if (b.p) {
            YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
            YA_HIP(ctx, hipFree(b.p));
            b.p = nullptr; b.cap = 0;
        }
        size_t want = bytes + bytes / 8; // a little headroom so steady-state calls never realloc
        want = (want + 255) & ~static_cast<size_t>(255);
        hipError_t e = ya_malloc(&b.p, want);
        if (e != hipSuccess) {
            b.p = nullptr;
            return hip_fail(ctx, e, (std::string("hipMalloc workspace '") + name + "' of " +
                                     std::to_string(want) + " bytes").c_str());
        }
        b.cap = want;
    }
    *out = b.p;
    return YAMS_OK;
}

size_t ws_trim(yams_accel_ctx* ctx, size_t keep_bytes) {
    size_t freed = 0;
    bool synced = false;
    for (auto& kv : ctx->bufs) {
        auto& b = kv.second;
        if (!b.p || b.cap <= keep_bytes) continue;
        if (!synced) { (void)hipStreamSynchronize(ctx->stream); synced = true; }
        if (hipFree(b.p) != hipSuccess) (void)hipGetLastError();
        freed += b.cap;
        b.p = nullptr; b.cap = 0;
    }
    return freed;
}

yams_status_t pinned_get(yams_accel_ctx* ctx, size_t bytes, void** out) {
    if (ctx->pinned_cap < bytes) {
        if (ctx->pinned) {
            YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
            YA_HIP(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr; ctx->pinned_cap = 0;
        }
        size_t want = (bytes * 2 + 4095) & ~static_cast<size_t>(4095);
        YA_HIP(ctx, ya_host_malloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_cap = want;
    }
    *out = ctx->pinned;
    return YAMS_OK;
}

static hipEvent_t get_event(yams_accel_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

TimedRegion::TimedRegion(yams_accel_ctx* c, const char* n, hipStream_t on) : ctx(c), name(n) {
    stream = on ? on : ctx->stream;
    if (!ctx->timing) return;
    a = get_event(ctx);
    b = get_event(ctx);
    if (a) (void)hipEventRecord(a, stream);
}
void TimedRegion::end() {
    if (!ctx->timing || !a || !b) return;
    (void)hipEventRecord(b, stream);
    ctx->spans[name].push_back({a, b});
}

// ---- staged host -> device copies -------------------------------------------------------------------------------
namespace {

class CopyCrew { // a few persistent threads that memcpy slices of one piece in parallel
public:
    explicit CopyCrew(unsigned n) {
        for (unsigned i = 0; i < n; ++i) threads_.emplace_back([this] { run(); });
    }
    ~CopyCrew() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // copies [src, src + bytes) to dst with every thread of the crew plus the caller; returns when all of it is there
    void copy(unsigned char* dst, const unsigned char* src, size_t bytes) {
        if (bytes == 0) return;
        const size_t parts = threads_.size() + 1;
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = dst; src_ = src; bytes_ = bytes; next_ = 0;
            slice_ = ((bytes + parts - 1) / parts + 4095) & ~static_cast<size_t>(4095);
            remaining_ = (bytes + slice_ - 1) / slice_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return remaining_ == 0; });
    }
private:
    void work() { // slices of the current piece until none is left
        for (;;) {
            size_t off, len;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (next_ >= bytes_) return;
                off = next_; len = std::min(slice_, bytes_ - off); next_ += slice_;
            }
            std::memcpy(dst_ + off, src_ + off, len);
            std::lock_guard<std::mutex> lk(mu_);
            if (--remaining_ == 0) done_.notify_all();
        }
    }
    void run() {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || next_ < bytes_; });
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    unsigned char* dst_ = nullptr; const unsigned char* src_ = nullptr;
    size_t bytes_ = 0, slice_ = 1, next_ = 0, remaining_ = 0;
    bool stop_ = false;
};

struct StageRing {
    static constexpr int kBufs = 3;
    static constexpr size_t kPiece = size_t(32) << 20;
    std::mutex mu;                 // one staged upload at a time (the crew and the ring are shared)
    unsigned char* buf[kBufs] = {nullptr, nullptr, nullptr};
    // events belong to a device: one set per device that ever uploaded (a multi-shard corpus_append alternates devices
    // with every stripe — re-making the set on each change cost three event destroy / create pairs per stripe)
    std::map<int, std::array<hipEvent_t, kBufs>> landed_by_device;
    hipEvent_t guard[kBufs] = {nullptr, nullptr, nullptr};   // the event behind the copy that last read buf[i] (null: idle)
    std::unique_ptr<CopyCrew> crew;
    bool broken = false;
};
StageRing& ring() { static StageRing* r = new StageRing(); return *r; } // (leaked on purpose: worker threads at exit)

bool source_is_pinned(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; } // plain malloc'd memory: "invalid value"
    return a.type == hipMemoryTypeHost;
}

} // namespace

hipError_t staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes < (size_t(8) << 20) || source_is_pinned(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    StageRing& R = ring();
    std::unique_lock<std::mutex> lk(R.mu, std::try_to_lock);
    if (!lk.owns_lock() || R.broken) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream); // busy: the plain path
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!R.crew) {
        unsigned hw = std::thread::hardware_concurrency();
        R.crew.reset(new CopyCrew(std::max(1u, std::min(7u, hw > 2 ? hw / 2 - 1 : 1u))));
    }
    auto found = R.landed_by_device.find(dev);
    if (found == R.landed_by_device.end()) {
        std::array<hipEvent_t, StageRing::kBufs> ev{};
        for (int i = 0; i < StageRing::kBufs; ++i)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); R.broken = true; }
        found = R.landed_by_device.emplace(dev, ev).first;
    }
    hipEvent_t* const landed = found->second.data();
    for (int i = 0; i < StageRing::kBufs && !R.broken; ++i)
        if (!R.buf[i] && ya_host_malloc(reinterpret_cast<void**>(&R.buf[i]), StageRing::kPiece, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError(); R.buf[i] = nullptr; R.broken = true;
        }
    if (R.broken) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    const unsigned char* sp = static_cast<const unsigned char*>(src);
    unsigned char* dp = static_cast<unsigned char*>(dst);
    int b = 0;
    for (size_t off = 0; off < bytes; off += StageRing::kPiece, b = (b + 1) % StageRing::kBufs) {
        const size_t len = std::min(StageRing::kPiece, bytes - off);
        if (R.guard[b]) { const hipError_t e = hipEventSynchronize(R.guard[b]); R.guard[b] = nullptr; if (e != hipSuccess) return e; }
        R.crew->copy(R.buf[b], sp + off, len);
        hipError_t e = hipMemcpyAsync(dp + off, R.buf[b], len, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(landed[b], stream);
        if (e != hipSuccess) return e;
        R.guard[b] = landed[b];
    }
    // the ring is reused by the next call (possibly of another device): everything it holds must have left host memory
    // before the lock goes — the call returns with the SOURCE free to reuse and the last pieces landed
    for (int i = 0; i < StageRing::kBufs; ++i)
        if (R.guard[i]) { const hipError_t e = hipEventSynchronize(R.guard[i]); R.guard[i] = nullptr; if (e != hipSuccess) return e; }
    return hipSuccess;
}

} // namespace yams_accel

using namespace yams_accel;

extern "C" {

int yams_accel_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

yams_status_t yams_accel_ctx_create(int device, void* hip_stream, yams_accel_ctx** out_ctx) {
    if (!out_ctx) return YAMS_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int n = yams_accel_device_count();
    if (n <= 0) return YAMS_ERR_UNSUPPORTED; // no GPU: there is deliberately no CPU fallback
    if (device < 0 || device >= n) return YAMS_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return YAMS_ERR_INTERNAL; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return YAMS_ERR_INTERNAL; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return YAMS_ERR_UNSUPPORTED;
    auto* ctx = new yams_accel_ctx();
    ctx->device = device;
    if (hip_stream) {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
        ctx->owns_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            delete ctx;
            return YAMS_ERR_INTERNAL;
        }
        ctx->owns_stream = true;
    }
    // side stream for work that must not queue behind the main stream (long SHA-256 chains)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        yams_accel_ctx_destroy(ctx);
        return YAMS_ERR_INTERNAL;
    }
    *out_ctx = ctx;
    return YAMS_OK;
}

void yams_accel_ctx_destroy(yams_accel_ctx* ctx) {
    if (!ctx) return;
    (void)yams_accel_ctx_release_sweep_hold(ctx, nullptr); // (a context must not take its gate with it)
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
    for (auto& kv : ctx->bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (auto& kv : ctx->spans)
        for (auto& s : kv.second) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

yams_status_t yams_accel_gate_create(int device, yams_accel_gate** out_gate) {
    if (!out_gate) return YAMS_ERR_INVALID_ARG;
    *out_gate = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return YAMS_ERR_UNSUPPORTED; }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return YAMS_ERR_INTERNAL; }
    auto* g = new yams_accel_gate();
    g->device = device;
    if (hipEventCreateWithFlags(&g->last, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); delete g; return YAMS_ERR_INTERNAL; }
    *out_gate = g;
    return YAMS_OK;
}

void yams_accel_gate_destroy(yams_accel_gate* gate) {
    if (!gate) return;
    (void)hipSetDevice(gate->device);
    if (gate->last) (void)hipEventDestroy(gate->last);
    delete gate;
}

yams_status_t yams_accel_ctx_set_gate(yams_accel_ctx* ctx, yams_accel_gate* gate) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (ctx->gate && ctx->gate != gate) (void)yams_accel_ctx_release_sweep_hold(ctx, nullptr);
    if (gate && gate->device != ctx->device) return yams_accel::fail(ctx, YAMS_ERR_INVALID_ARG, "gate and context are on different devices");
    ctx->gate = gate;
    return YAMS_OK;
}

yams_status_t yams_accel_ctx_set_sweep_hold(yams_accel_ctx* ctx, int on) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    ctx->sweep_hold = on != 0;
    if (!on) return yams_accel_ctx_release_sweep_hold(ctx, nullptr);
    return YAMS_OK;
}

yams_status_t yams_accel_ctx_release_sweep_hold(yams_accel_ctx* ctx, void* hip_stream) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    yams_accel_gate* g = ctx->gate;
    if (!g) return YAMS_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->holder != ctx) return YAMS_OK; // (no sweep ran under the hold: nothing to release)
        (void)hipSetDevice(ctx->device);
        // the next sweep of any context of this gate starts behind everything enqueued on that stream so far
        if (hipEventRecord(g->last, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream) != hipSuccess) (void)hipGetLastError();
        g->armed = true;
        g->holder = nullptr;
    }
    g->cv.notify_all();
    return YAMS_OK;
}

yams_status_t yams_accel_ctx_synchronize(yams_accel_ctx* ctx) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return YAMS_OK;
}

const char* yams_accel_last_error(const yams_accel_ctx* ctx) {
    return ctx ? ctx->last_error.c_str() : "null context";
}

yams_status_t yams_accel_device_info_json(yams_accel_ctx* ctx, char** out_json) {
    if (!ctx || !out_json) return YAMS_ERR_INVALID_ARG;
    hipDeviceProp_t p;
    YA_HIP(ctx, hipGetDeviceProperties(&p, ctx->device));
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    std::ostringstream os;
    os << "{\"backend\":\"hip\",\"arch\":\"" << p.gcnArchName << "\",\"name\":\"" << p.name
       << "\",\"compute_units\":" << p.multiProcessorCount << ",\"clock_khz\":" << p.clockRate
       << ",\"memory_clock_khz\":" << p.memoryClockRate << ",\"memory_bus_bits\":" << p.memoryBusWidth
       << ",\"hbm_bytes\":" << total_b << ",\"hbm_free_bytes\":" << free_b
       << ",\"lds_per_block\":" << p.sharedMemPerBlock << ",\"wavefront\":" << p.warpSize
       << ",\"l2_bytes\":" << p.l2CacheSize << ",\"version\":\"" << YAMS_ACCEL_VERSION_STRING << "\"";
    const auto& hi = ctx->host_ingest;
    if (hi.batches)
        os << ",\"last_host_ingest\":{\"bytes\":" << hi.bytes << ",\"batches\":" << hi.batches << ",\"batch_bytes\":" << hi.batch_bytes
           << ",\"slots\":" << hi.slots << ",\"alloc_ms\":" << hi.alloc_ms
           << ",\"release_ms\":" << hi.release_ms << ",\"total_ms\":" << hi.total_ms << "}";
    os << "}";
    const std::string s = os.str();
    char* buf = static_cast<char*>(std::malloc(s.size() + 1));
    if (!buf) return YAMS_ERR_INTERNAL;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    *out_json = buf;
    return YAMS_OK;
}

void yams_accel_free_string(char* s) { std::free(s); }

uint64_t yams_accel_trim(int device) { return static_cast<uint64_t>(yams_accel::big_trim(device)); }

yams_status_t yams_accel_malloc(yams_accel_ctx* ctx, size_t bytes, void** out_dev) {
    if (!ctx || !out_dev) return YAMS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    YA_HIP(ctx, hipMalloc(out_dev, bytes ? bytes : 16));
    return YAMS_OK;
}
void yams_accel_free(yams_accel_ctx* ctx, void* dev) {
    if (!ctx || !dev) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dev);
}
yams_status_t yams_accel_upload(yams_accel_ctx* ctx, void* dst_dev, const void* src_host,
                                size_t bytes) {
    if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return YAMS_ERR_INVALID_ARG;
    if (!bytes) return YAMS_OK;
    (void)hipSetDevice(ctx->device);
    YA_HIP(ctx, staged_h2d(dst_dev, src_host, bytes, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return YAMS_OK;
}
yams_status_t yams_accel_download(yams_accel_ctx* ctx, void* dst_host, const void* src_dev,
                                  size_t bytes) {
    if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return YAMS_ERR_INVALID_ARG;
    if (!bytes) return YAMS_OK;
    (void)hipSetDevice(ctx->device);
    YA_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return YAMS_OK;
}

yams_status_t yams_accel_enable_kernel_timing(yams_accel_ctx* ctx, int enable) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->spans) {
        for (auto& s : kv.second) { ctx->event_pool.push_back(s.a); ctx->event_pool.push_back(s.b); }
        kv.second.clear();
    }
    ctx->timing = enable != 0;
    return YAMS_OK;
}

yams_status_t yams_accel_last_kernel_ms(const yams_accel_ctx* cctx, const char* kernel,
                                        double* out_ms_per_launch, uint64_t* out_launches) {
    auto* ctx = const_cast<yams_accel_ctx*>(cctx);
    if (!ctx || !kernel || !out_ms_per_launch || !out_launches) return YAMS_ERR_INVALID_ARG;
    *out_ms_per_launch = 0.0;
    *out_launches = 0;
    auto it = ctx->spans.find(kernel);
    if (it == ctx->spans.end() || it->second.empty()) return YAMS_ERR_NOT_FOUND;
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    for (auto& s : it->second) {
        float ms = 0.f;
        YA_HIP(ctx, hipEventElapsedTime(&ms, s.a, s.b));
        total += ms;
    }
    *out_launches = it->second.size();
    *out_ms_per_launch = total / static_cast<double>(it->second.size());
    return YAMS_OK;
}

} // extern "C"

// ---- allocation fault injection (tests of the out-of-memory paths; see the header) ---------------------------------
#ifdef YAMS_ACCEL_MEASURE
extern "C" void yams_accel_debug_fail_alloc_after(int64_t n) { yams_accel::g_fail_after.store(n < 0 ? -1 : n); }
extern "C" uint64_t yams_accel_debug_alloc_faults(void) { return yams_accel::g_faults.load(); }
extern "C" int yams_accel_debug_alloc_injection_compiled(void) { return 1; }
#else   // the product library: the symbols stay (one ABI for both builds), nothing can be armed
extern "C" void yams_accel_debug_fail_alloc_after(int64_t) {}
extern "C" uint64_t yams_accel_debug_alloc_faults(void) { return 0; }
extern "C" int yams_accel_debug_alloc_injection_compiled(void) { return 0; }
#endif
