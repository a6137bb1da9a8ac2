// sharded_api.cpp — one search over a corpus that is row-sharded across the devices of this node, behind the
// C ABI: ONE process, one RCCL communicator over the shard devices, one ncclAllGather of the packed per-shard
// record per batch, the k-way merge kernel behind it.  This is what a C++ host calling
// SqliteVecBackend::searchSimilarBatch (src/vector/sqlite_vec_backend.cpp:1612-1647; caller
// src/daemon/components/EmbeddingService.cpp:572) reaches; bench.py's one-process-per-GPU form
// (yams_amd/dist.py) uses the same record layout and the same merge kernel behind torch.distributed.
//
// Shape of a handle:
//   * `lanes` batches in flight (submit / wait).  A lane owns, on every shard, a context (stream + workspace),
//     a record buffer and a side stream, plus pinned staging for its queries and its merged result.
//   * one persistent worker thread per (shard, lane), bound to the shard's device for its whole life: no
//     thread is created per call.  submit() copies the queries into the lane's pinned staging ONCE and
//     wakes the lane's workers; each uploads the batch to its device, runs the exact per-shard top-k
//     (yams_scan_topk_device, fp64-exact: the gathered scores are final, the merge is a pure comparison)
//     into the lane's record, then joins the exchange on the lane's SIDE stream: ncclAllGather of the
//     record, and on the root shard merge_topk_kernel + the download into pinned memory behind it.
//   * exchanges of one communicator must be issued in the same order on every rank: a turnstile per shard
//     admits them in submit order, whichever lane finishes first.  Every batch that was given a place in that
//     order takes part in its exchange on every exit path (a failed scan contributes an empty record): a rank
//     that skipped one would hang the others.
//   * THE FENCE.  The filter sweep is a persistent grid that owns every CU of its device, and a collective is a
//     CU-resident kernel that spins on its peers.  Left to float, the all-gather of batch i on GPU g would have to
//     find a CU under the sweep of batch i + 1 and then wait for GPU h's copy of the kernel, itself queued behind
//     h's sweep — the sweeps of different GPUs get coupled through the few CUs the collective holds.  So with two
//     or more shards a shard's sweep of batch i + 1 is enqueued only after its part of the exchange of batch i
//     has been enqueued, and waits (on the device) for that part to complete: on the root shard all-gather +
//     merge + download, elsewhere the all-gather.  What still overlaps the sweep of batch i is everything in
//     front of the sweep of batch i + 1: query upload, preparation, the sample pass (in its half-tile form while a
//     fence or a sweep hold is installed — scan_api.cpp, i8_sample_small_grid: the resident-query sample form would be
//     a second grid that owns every CU, and the exchange of batch i may still be on the device when it starts).  The price is the gap
//     between two sweeps (host hand-over + the all-gather of ~1 MB per rank + on the root the merge); what it
//     buys is that no collective ever shares a device with a sweep.  (accel_ctx.h: before_sweep.)
//   * the contexts of one device share a sweep gate (yams_accel_gate): big filter sweeps run one after the
//     other, everything around them overlaps.
//   * THE DEADLINE.  A collective whose peer never arrives spins on the device for ever; a host that waits for it
//     with hipEventSynchronize hangs with it.  wait() therefore polls under a deadline (options.exchange_timeout_ms,
//     30 s by default): when a batch's workers or its exchange have not finished by then the handle is declared
//     STUCK — wait() returns YAMS_ERR_TIMEOUT with a one-line diagnosis (which shards, which batch, how many
//     exchanges completed before it), the communicator is aborted (ncclCommAbort, when the library has it) so that
//     the spinning kernels leave the devices, every later submit() fails at once, and destroy() returns without
//     waiting for work that cannot finish (what cannot be freed safely is leaked, and said so on stderr).
//   * what the exchange costs is measured per batch: events on the root shard's side stream around all-gather +
//     merge + download (`exchange_ms` in ..._info_json; it includes the wait for the slowest peer's scan).
// The collective library is bound at run time (dlopen of librccl.so.1, or of the library the options name, when
// the first communicator is needed): a host that hashes files or searches one GPU never maps the 570 MB of RCCL.
// Shards that SHARE a device (the parity tests on a one-GPU box) cannot form an RCCL communicator — RCCL refuses
// duplicate devices — and use device-to-device copies of the records instead (or, for the tests of the kRccl
// code path, the stand-in library under tests/stub_coll); that path is not a second production backend.
// The corpus itself never moves.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <vector>

#include "accel_ctx.h"
#include "scan_launch.h"

using namespace yams_accel;

namespace {
uint64_t align16(uint64_t v) { return (v + 15) & ~static_cast<uint64_t>(15); }
} // namespace

extern "C" void yams_scan_record_layout(uint32_t n_queries, uint32_t k, int with_dist, int with_ranks,
                                        yams_scan_record_layout_t* out) {
    if (!out) return;
    const uint64_t qk = static_cast<uint64_t>(n_queries) * std::max<uint32_t>(k, 1);
    uint64_t off = 0;
    out->scores_off = off; off = align16(off + qk * 4);
    out->rows_off = off;   off = align16(off + qk * 8);
    out->counts_off = off; off = align16(off + static_cast<uint64_t>(n_queries) * 4);
    out->dist_off = with_dist ? off : UINT64_MAX;
    if (with_dist) off = align16(off + qk * 4);
    out->ranks_off = with_ranks ? off : UINT64_MAX;
    if (with_ranks) off = align16(off + qk * 4);
    out->bytes = off;
}

extern "C" yams_status_t yams_scan_merge_records_device(
    yams_accel_ctx* ctx, uint32_t n_shards, uint32_t n_queries, const yams_scan_params_t* params,
    const void* records, uint64_t record_stride, const yams_scan_record_layout_t* lay,
    const uint32_t* rank_of_row, int64_t rank_row_base, float* out_scores, int64_t* out_rows,
    uint32_t* out_counts, float* out_dist) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!params || !lay || n_shards == 0) return fail(ctx, YAMS_ERR_INVALID_ARG, "bad merge arguments");
    if (n_queries == 0) return YAMS_OK;
    if (!records || !out_counts) return fail(ctx, YAMS_ERR_INVALID_ARG, "null records/counts");
    if ((record_stride & 7u) || record_stride < lay->bytes)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "record_stride must be a multiple of 8 and hold a record");
    (void)hipSetDevice(ctx->device);
    if (params->k == 0) {
        YA_HIP(ctx, hipMemsetAsync(out_counts, 0, static_cast<size_t>(n_queries) * 4, ctx->stream));
        return YAMS_OK;
    }
    if (!out_scores || !out_rows) return fail(ctx, YAMS_ERR_INVALID_ARG, "null merge outputs");
    if (params->metric == YAMS_SCAN_L2 && lay->dist_off == UINT64_MAX)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "L2 merge needs distances in the records");
    if (static_cast<uint64_t>(n_shards) * params->k > 8192)
        return fail(ctx, YAMS_ERR_UNSUPPORTED, "n_shards * k exceeds 8192");
    const unsigned char* base = static_cast<const unsigned char*>(records);
    MergeLaunch M{};
    M.n_shards = n_shards; M.n_queries = n_queries; M.k = params->k; M.metric = params->metric;
    // (a caller that cuts its own result further — rounds of a k above YAMS_SCAN_MAX_K — defers the vec0 threshold)
    M.threshold = (params->flags & YAMS_SCAN_FLAG_DEFER_THRESHOLD) ? -__builtin_inff() : params->similarity_threshold;
    M.in_scores = reinterpret_cast<const float*>(base + lay->scores_off);
    M.in_rows = reinterpret_cast<const int64_t*>(base + lay->rows_off);
    M.in_counts = reinterpret_cast<const uint32_t*>(base + lay->counts_off);
    M.in_dist = lay->dist_off != UINT64_MAX ? reinterpret_cast<const float*>(base + lay->dist_off) : nullptr;
    M.in_ranks = lay->ranks_off != UINT64_MAX ? reinterpret_cast<const uint32_t*>(base + lay->ranks_off) : nullptr;
    M.st_scores = record_stride / 4; M.st_rows = record_stride / 8; M.st_counts = record_stride / 4;
    M.st_dist = record_stride / 4; M.st_ranks = record_stride / 4;
    M.rank_of_row = rank_of_row; M.rank_row_base = rank_row_base;
    M.out_scores = out_scores; M.out_rows = out_rows; M.out_counts = out_counts; M.out_dist = out_dist;
    TimedRegion tr(ctx, "merge_topk");
    YA_HIP(ctx, launch_merge(ctx->stream, M));
    tr.end();
    return YAMS_OK;
}


namespace {

// ---- RCCL, bound at run time -----------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;   // optional
    decltype(&ncclCommAbort) CommAbort = nullptr;   // optional
    int version = 0;
    std::string path, error;
    bool ok() const { return handle != nullptr; }
};

// One binding per library path ("" = the default search for RCCL itself); bindings live for the process.
Rccl& rccl(const std::string& wanted) {
    static std::mutex mu;
    static std::map<std::string, std::unique_ptr<Rccl>> bound;
    std::lock_guard<std::mutex> lk(mu);
    std::unique_ptr<Rccl>& slot = bound[wanted];
    if (slot) return *slot;
    slot.reset(new Rccl());
    Rccl& r = *slot;
    if (!wanted.empty()) {
        r.handle = dlopen(wanted.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) { if (const char* e = dlerror()) r.error = e; return r; }
    } else {
        // the SONAME first: a process that already maps RCCL (e.g. through torch) gets that very copy
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            if (const char* e = dlerror()) r.error = e;
        }
        if (!r.handle) return r;
    }
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(r.handle, name);
        if (!p) { r.error = std::string("the collective library lacks ") + name; }
        return p;
    };
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.GetVersion || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        dlclose(r.handle); r.handle = nullptr; return r;
    }
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.handle, "ncclCommAbort"));
    r.error.clear();
    (void)r.GetVersion(&r.version);
    Dl_info info{};
    if (dladdr(reinterpret_cast<void*>(r.AllGather), &info) && info.dli_fname) r.path = info.dli_fname;
    return r;
}

enum Mode : int { kNone = 0, kRccl = 1, kPeer = 2 };

struct ShardLane {                  // what one lane owns on one shard
    yams_accel_ctx* ctx = nullptr;  // the scan's context: its own stream and workspace
    hipStream_t side = nullptr;     // exchange (and, on the root shard, merge + download) of this lane
    float* queries = nullptr; size_t queries_cap = 0;           // the batch on this device
    unsigned char* rec = nullptr; size_t rec_cap = 0;           // this shard's packed record (the send buffer)
    unsigned char* gathered = nullptr; size_t gathered_cap = 0; // [n shards][record stride]
    std::thread worker;
};

struct Lane {
    // --- state, under yams_scan_sharded::mu
    bool acquired = false, submitted = false, trivial = false, want_diag = false;
    bool ordered = false;           // the batch has a place in the exchange order (seq)
    uint64_t seq = 0;               // position in the exchange order (turnstile)
    uint64_t job = 0;               // generation: the lane's workers run when it moves
    uint32_t pending = 0;           // workers that have not reported yet
    uint32_t peer_left = 0;         // kPeer: records that have not landed on the root's device yet
    // --- the batch
    uint32_t nq = 0, dim = 0; yams_scan_params_t prm{}; bool l2 = false;
    const uint32_t* rank_of_row = nullptr; int64_t rank_row_base = 0;
    std::vector<yams_scan_corpus_t> views;
    yams_scan_record_layout_t lay{}; uint64_t stride = 0;
    std::vector<yams_status_t> st; std::vector<yams_scan_diag_t> dg; std::vector<std::string> err;
    yams_status_t merge_st = YAMS_OK; std::string merge_err; bool merge_issued = false;
    // --- resources
    float* h_queries = nullptr; size_t h_queries_cap = 0;   // pinned: the batch, uploaded from here by every shard
    unsigned char* h_out = nullptr; size_t h_out_cap = 0;   // pinned: counts | scores | rows | dist of the merged result
    yams_accel_ctx* merge_ctx = nullptr;                    // root device, bound to the root's side stream
    hipEvent_t done = nullptr;                              // merged result has landed in h_out
    hipEvent_t ex_begin = nullptr;                          // head of the exchange on the root shard's side stream
    bool ex_timed = false;
    std::vector<char> reported;                             // per shard: its worker is through with the batch
    std::vector<ShardLane> sh;
    // ONE deadline per batch, fixed at submit: the handle's exchange timeout + an allowance for the scan itself that grows
    // with the work the batch asks for (a huge FORCE_EXACT batch is slow, not stuck).  Both phases of wait() look at it.
    std::chrono::steady_clock::time_point deadline{};
};

size_t align16s(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }
struct OutLayout { size_t counts, scores, rows, dist, bytes; };
OutLayout out_layout(size_t nq, size_t k) {
    OutLayout o;
    size_t off = 0;
    o.counts = off; off = align16s(off + nq * 4);
    o.scores = off; off = align16s(off + nq * k * 4);
    o.rows = off;   off = align16s(off + nq * k * 8);
    o.dist = off;   off = align16s(off + nq * k * 4);
    o.bytes = off;
    return o;
}

} // namespace

struct yams_scan_sharded {
    std::vector<int> device;
    uint32_t n = 0, n_lanes = 0;
    int mode = kNone;
    bool fenced = false;
    Rccl* R = nullptr;                           // kRccl: the bound collective library
    std::vector<ncclComm_t> comm;                // kRccl: one per shard, one communicator
    std::map<int, yams_accel_gate*> gates;       // one per distinct device
    std::vector<std::unique_ptr<Lane>> lanes;
    std::mutex mu;
    std::condition_variable cv_job, cv_done, cv_lane, cv_turn, cv_peer;
    std::vector<uint64_t> coll_next;             // per shard: the seq whose exchange is due next
    std::vector<hipEvent_t> fence_ev;            // per shard: end of its part of the most recent exchange
    std::vector<char> fence_armed;
    uint64_t next_seq = 0;
    bool stop = false;
    std::string last_error, fallback_reason;
    std::atomic<uint64_t> batches{0}, collectives{0}, fence_waits{0};
    // the exchange as the root shard's side stream saw it (events): all-gather + merge + download, per batch
    std::atomic<uint64_t> exchanges_timed{0}, exchange_us_sum{0}, exchange_us_max{0};
    uint32_t timeout_ms = 30000;                 // the deadline of wait(); UINT32_MAX: none
    bool stuck = false;                          // a batch missed the deadline (under mu)
    std::string stuck_why;
    // kRccl: a shard's communicator is used (exchange_shard) and aborted (declare_stuck) under its own mutex — an abort
    // frees the communicator, and a worker of another lane may be inside ncclAllGather on it at that moment
    std::vector<std::unique_ptr<std::mutex>> comm_mu;
};

namespace {

yams_status_t set_error(yams_scan_sharded* s, yams_status_t st, const std::string& m) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->last_error = m;
    return st;
}

// merge + download of lane L's batch, enqueued on the root shard's side stream of that lane (the current
// device must be the root's)
void issue_merge(yams_scan_sharded* s, Lane& L) {
    yams_accel_ctx* m = L.merge_ctx;
    hipStream_t st = L.sh[0].side;
    const size_t nq = L.nq, k = L.prm.k;
    const OutLayout o = out_layout(nq, k);
    float* d_s; int64_t* d_r; uint32_t* d_c; float* d_d;
    L.merge_issued = true;
    yams_status_t r = ws_get(m, "merged_scores", nq * k * 4, (void**)&d_s);
    if (r == YAMS_OK) r = ws_get(m, "merged_rows", nq * k * 8, (void**)&d_r);
    if (r == YAMS_OK) r = ws_get(m, "merged_counts", nq * 4, (void**)&d_c);
    if (r == YAMS_OK) r = ws_get(m, "merged_dist", nq * k * 4, (void**)&d_d);
    if (r == YAMS_OK)
        r = yams_scan_merge_records_device(m, s->n, L.nq, &L.prm, L.sh[0].gathered, L.stride, &L.lay, L.rank_of_row,
                                           L.rank_row_base, d_s, d_r, d_c, d_d);
    if (r == YAMS_OK) {
        hipError_t e = hipMemcpyAsync(L.h_out + o.counts, d_c, nq * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(L.h_out + o.scores, d_s, nq * k * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(L.h_out + o.rows, d_r, nq * k * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(L.h_out + o.dist, d_d, nq * k * 4, hipMemcpyDeviceToHost, st); // (cosine: 1 - similarity)
        if (e != hipSuccess) { (void)hipGetLastError(); r = fail(m, YAMS_ERR_INTERNAL, "download of the merged result failed"); }
    }
    (void)hipEventRecord(L.done, st);
    L.merge_st = r;
    if (r != YAMS_OK) L.merge_err = yams_accel_last_error(m);
}

// The scan of shard i's part of lane L's batch into the lane's record (worker thread of (shard i, lane L); current
// device = the shard's).  Returns with the scan's stream idle; never throws.
yams_status_t scan_shard(yams_scan_sharded* s, Lane& L, uint32_t i, std::string& err) {
    ShardLane& SL = L.sh[i];
    yams_accel_ctx* c = SL.ctx;
    const size_t nq = L.nq, dim = L.dim;
    yams_status_t st = YAMS_OK;
    // The fence (see the header comment): called by the scan right before it enqueues a sweep on `sst`.  Installed for
    // the duration of this batch only — between batches the lane-0 contexts are the caller's (uploads, shadow builds,
    // direct scans through yams_scan_sharded_ctx).
    struct Hook {
        yams_accel_ctx* c;
        ~Hook() { c->before_sweep = nullptr; }
    } hook{c};
    if (s->fenced && L.ordered)
        c->before_sweep = [s, &L, i](hipStream_t sst) {
            bool armed;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                if (s->coll_next[i] != L.seq) { ++s->fence_waits; s->cv_turn.wait(lk, [&] { return s->coll_next[i] == L.seq; }); }
                armed = s->fence_armed[i] != 0;
            }
            if (armed && hipStreamWaitEvent(sst, s->fence_ev[i], 0) != hipSuccess) (void)hipGetLastError();
        };
    try {
        // this lane's previous exchange finished long ago (its wait() returned); a stream query, not a stall
        (void)hipStreamSynchronize(SL.side);
        if (hipMemcpyAsync(SL.queries, L.h_queries, nq * dim * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            st = fail(c, YAMS_ERR_INTERNAL, "query upload failed");
        }
        if (st == YAMS_OK) {
            yams_scan_params_t prm = L.prm;
            if (L.l2 && s->mode != kNone) prm.flags |= YAMS_SCAN_FLAG_DEFER_THRESHOLD; // vec0: the k nearest first, the threshold after the merge
            unsigned char* d_rec = SL.rec;
            st = yams_scan_topk_device(c, &L.views[i], SL.queries, L.nq, &prm, reinterpret_cast<float*>(d_rec + L.lay.scores_off),
                                       reinterpret_cast<int64_t*>(d_rec + L.lay.rows_off),
                                       reinterpret_cast<uint32_t*>(d_rec + L.lay.counts_off),
                                       L.lay.dist_off != UINT64_MAX ? reinterpret_cast<float*>(d_rec + L.lay.dist_off) : nullptr, nullptr,
                                       L.want_diag ? &L.dg[i] : nullptr);
        }
        if (st != YAMS_OK) err = yams_accel_last_error(c);
    } catch (const std::exception& e) { st = YAMS_ERR_INTERNAL; err = e.what(); }
    catch (...) { st = YAMS_ERR_INTERNAL; err = "unknown exception in a shard worker"; }
    if (st != YAMS_OK) {
        // the exchange still happens (a rank that skips a collective hangs the others): an empty record
        (void)hipMemsetAsync(SL.rec, 0, static_cast<size_t>(L.stride), c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();
    }
    return st;
}

// Shard i's part of the exchange of lane L's batch, in the handle's exchange order.  Runs on EVERY exit path of a
// batch that was given a seq; never throws.
yams_status_t exchange_shard(yams_scan_sharded* s, Lane& L, uint32_t i, yams_status_t st, std::string& err) {
    ShardLane& SL = L.sh[i];
    {   // exchanges of one communicator go out in submit order on every rank
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_turn.wait(lk, [&] { return s->coll_next[i] == L.seq; });
    }
    if (i == 0) L.ex_timed = L.ex_begin && hipEventRecord(L.ex_begin, SL.side) == hipSuccess;
    try {
        if (s->mode == kRccl) {
            bool gone = false;
            ncclResult_t r = ncclSuccess;
            {
                std::lock_guard<std::mutex> ck(*s->comm_mu[i]);
                if (s->comm[i]) r = s->R->AllGather(SL.rec, SL.gathered, static_cast<size_t>(L.stride), ncclUint8, s->comm[i], SL.side);
                else gone = true;   // the handle was declared stuck and its communicator aborted: no collective any more
            }
            if (gone) {
                if (st == YAMS_OK) { err = "the communicator was aborted (the handle is stuck)"; st = YAMS_ERR_TIMEOUT; }
            } else if (r != ncclSuccess) {
                (void)hipGetLastError();
                if (st == YAMS_OK) { err = std::string("ncclAllGather failed: ") + s->R->GetErrorString(r); st = YAMS_ERR_INTERNAL; }
            }
            if (i == 0) { issue_merge(s, L); ++s->collectives; }
        } else {
            // shards that share a device (or a host without RCCL): the record is copied next to the others on the
            // root's device; the root's worker enqueues the merge once the last one has landed
            unsigned char* dst = L.sh[0].gathered + L.stride * i;
            hipError_t e;
            if (s->device[i] == s->device[0]) e = hipMemcpyAsync(dst, SL.rec, static_cast<size_t>(L.stride), hipMemcpyDeviceToDevice, SL.side);
            else e = hipMemcpyPeerAsync(dst, s->device[0], SL.rec, s->device[i], static_cast<size_t>(L.stride), SL.side);
            if (e == hipSuccess && i != 0) e = hipStreamSynchronize(SL.side);
            if (e != hipSuccess) { (void)hipGetLastError(); if (st == YAMS_OK) { err = "record copy to the merge device failed"; st = YAMS_ERR_INTERNAL; } }
            {
                std::unique_lock<std::mutex> lk(s->mu);
                --L.peer_left;
                if (i == 0) s->cv_peer.wait(lk, [&] { return L.peer_left == 0; });
                else s->cv_peer.notify_all();
            }
            if (i == 0) { issue_merge(s, L); ++s->collectives; }
        }
    } catch (const std::exception& e) { if (st == YAMS_OK) { st = YAMS_ERR_INTERNAL; err = e.what(); } }
    catch (...) { if (st == YAMS_OK) { st = YAMS_ERR_INTERNAL; err = "unknown exception in the record exchange"; } }
    // the fence: this shard's next sweep waits for everything enqueued on the side stream up to here
    const bool fence = s->fenced && hipEventRecord(s->fence_ev[i], SL.side) == hipSuccess;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (fence) s->fence_armed[i] = 1;
        ++s->coll_next[i];
    }
    s->cv_turn.notify_all();
    return st;
}

// One shard's part of lane L's batch (worker thread of (shard i, lane L); current device = the shard's).
yams_status_t run_shard(yams_scan_sharded* s, Lane& L, uint32_t i, std::string& err) {
    ShardLane& SL = L.sh[i];
    yams_accel_ctx* c = SL.ctx;
    const size_t nq = L.nq, k = L.prm.k;
    yams_status_t st = scan_shard(s, L, i, err);
    if (s->mode != kNone) return exchange_shard(s, L, i, st, err);
    // one shard: its own ordering (tie ranks included) is final, nothing to merge
    if (st != YAMS_OK) return st;
    const OutLayout o = out_layout(nq, k);
    unsigned char* d_rec = SL.rec;
    hipError_t e = hipMemcpyAsync(L.h_out + o.counts, d_rec + L.lay.counts_off, nq * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(L.h_out + o.scores, d_rec + L.lay.scores_off, nq * k * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(L.h_out + o.rows, d_rec + L.lay.rows_off, nq * k * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(L.h_out + o.dist, d_rec + L.lay.dist_off, nq * k * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); err = "result download failed"; return YAMS_ERR_INTERNAL; }
    return YAMS_OK;
}

void worker_main(yams_scan_sharded* s, uint32_t i, uint32_t li) {
    (void)hipSetDevice(s->device[i]);
    Lane& L = *s->lanes[li];
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_job.wait(lk, [&] { return s->stop || L.job != seen; });
            if (L.job == seen) return; // stop
            seen = L.job;
        }
        yams_status_t st; std::string err;
        try { st = run_shard(s, L, i, err); }
        catch (const std::exception& e) { st = YAMS_ERR_INTERNAL; err = e.what(); }
        catch (...) { st = YAMS_ERR_INTERNAL; err = "unknown exception in a shard worker"; }
        {
            std::lock_guard<std::mutex> lk(s->mu);
            L.st[i] = st; L.err[i] = std::move(err);
            L.reported[i] = 1;
            if (--L.pending == 0) s->cv_done.notify_all();
        }
    }
}

// A batch missed the deadline: the handle is stuck.  Abort the communicator so that spinning collective kernels leave
// the devices (RCCL: ncclCommAbort); the streams behind them drain, the workers report, destroy() can run.
void declare_stuck(yams_scan_sharded* s, const std::string& why) {
    bool first;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        first = !s->stuck;
        s->stuck = true;
        if (first) s->stuck_why = why;
        s->last_error = why;
    }
    s->cv_lane.notify_all(); // callers blocked in lane_acquire learn about it
    if (!first) return;
    std::fprintf(stderr, "[yams_mi355x_accel] sharded search STUCK: %s\n", why.c_str());
    if (s->mode == kRccl && s->R && s->R->CommAbort) {
        int before = 0;
        (void)hipGetDevice(&before);
        for (uint32_t i = 0; i < s->comm.size(); ++i) {
            std::lock_guard<std::mutex> ck(*s->comm_mu[i]); // (an AllGather call only ENQUEUES: the lock is held for microseconds)
            if (s->comm[i]) { (void)hipSetDevice(s->device[i]); (void)s->R->CommAbort(s->comm[i]); s->comm[i] = nullptr; }
        }
        (void)hipSetDevice(before);
    }
}

void destroy_handle(yams_scan_sharded* s) {
    {
        std::unique_lock<std::mutex> lk(s->mu);
        // batches in flight run to their end first (their collectives need every rank) — unless the handle is stuck:
        // then they get a few seconds to drain behind the aborted communicator, and what has not drained is leaked
        auto idle = [&] { for (auto& L : s->lanes) if (L->submitted && L->pending) return false; return true; };
        if (!s->stuck) s->cv_done.wait(lk, idle);
        else if (!s->cv_done.wait_for(lk, std::chrono::seconds(5), idle)) {
            std::fprintf(stderr, "[yams_mi355x_accel] destroy of a stuck sharded handle: workers still blocked, resources leaked\n");
            s->stop = true;
            lk.unlock();
            s->cv_job.notify_all();
            for (auto& L : s->lanes)
                for (auto& SL : L->sh) if (SL.worker.joinable()) SL.worker.detach();
            return; // (the handle and what hangs on it stay allocated: threads may still touch them)
        }
        s->stop = true;
    }
    s->cv_job.notify_all();
    for (auto& L : s->lanes)
        for (auto& SL : L->sh) if (SL.worker.joinable()) SL.worker.join();
    if (s->stuck) {
        // The workers have reported, but the DEVICES may still hang (peer-copy mode, or a collective library without
        // ncclCommAbort: nothing took the spinning kernels off them).  Synchronising or freeing behind such a stream blocks
        // for good: look first (a query, not a wait), and leak the handle's device resources when anything is incomplete.
        bool quiet = true;
        for (auto& L : s->lanes)
            for (uint32_t i = 0; i < L->sh.size() && quiet; ++i) {
                ShardLane& SL = L->sh[i];
                (void)hipSetDevice(s->device[i]);
                if (SL.side && hipStreamQuery(SL.side) != hipSuccess) { (void)hipGetLastError(); quiet = false; }
                if (quiet && SL.ctx && SL.ctx->stream && hipStreamQuery(SL.ctx->stream) != hipSuccess) { (void)hipGetLastError(); quiet = false; }
            }
        if (!quiet) {
            std::fprintf(stderr, "[yams_mi355x_accel] destroy of a stuck sharded handle: device work still pending, resources leaked\n");
            return; // (buffers, streams, contexts stay: whatever hangs may still touch them)
        }
    }
    for (auto& L : s->lanes) {
        for (uint32_t i = 0; i < L->sh.size(); ++i) {
            ShardLane& SL = L->sh[i];
            (void)hipSetDevice(s->device[i]);
            if (SL.side) (void)hipStreamSynchronize(SL.side);
        }
    }
    if (s->mode == kRccl && s->R)
        for (uint32_t i = 0; i < s->comm.size(); ++i)
            if (s->comm[i]) { (void)hipSetDevice(s->device[i]); (void)s->R->CommDestroy(s->comm[i]); }
    for (auto& L : s->lanes) {
        (void)hipSetDevice(s->device[0]);
        if (L->merge_ctx) yams_accel_ctx_destroy(L->merge_ctx); // bound to the root's side stream, which it does not own
        if (L->done) (void)hipEventDestroy(L->done);
        if (L->ex_begin) (void)hipEventDestroy(L->ex_begin);
        if (L->h_queries) (void)hipHostFree(L->h_queries);
        if (L->h_out) (void)hipHostFree(L->h_out);
        for (uint32_t i = 0; i < L->sh.size(); ++i) {
            ShardLane& SL = L->sh[i];
            (void)hipSetDevice(s->device[i]);
            if (SL.ctx) yams_accel_ctx_destroy(SL.ctx);
            if (SL.queries) (void)hipFree(SL.queries);
            if (SL.rec) (void)hipFree(SL.rec);
            if (SL.gathered) (void)hipFree(SL.gathered);
            if (SL.side) (void)hipStreamDestroy(SL.side);
        }
    }
    for (uint32_t i = 0; i < s->fence_ev.size(); ++i)
        if (s->fence_ev[i]) { (void)hipSetDevice(s->device[i]); (void)hipEventDestroy(s->fence_ev[i]); }
    for (auto& kv : s->gates) yams_accel_gate_destroy(kv.second);
    delete s;
}

yams_status_t create_impl(const int* devices, uint32_t n_shards, const yams_scan_sharded_options_t* opt, yams_scan_sharded** out) {
    if (!out) return YAMS_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_shards == 0 || n_shards > 64) return YAMS_ERR_INVALID_ARG;
    uint32_t n_lanes = 2, collective = YAMS_SHARDED_COLLECTIVE_AUTO, fence = YAMS_SHARDED_FENCE_AUTO, timeout_ms = 30000;
    std::string library;
    if (opt) {
        if (opt->struct_size < 16) return YAMS_ERR_INVALID_ARG; // (16: the round-3 struct — lanes, collective, one reserved word)
        if (opt->lanes) n_lanes = opt->lanes;
        collective = opt->collective;
        fence = opt->fence;
        if (opt->struct_size >= offsetof(yams_scan_sharded_options_t, exchange_timeout_ms) && opt->rccl_library) library = opt->rccl_library;
        if (opt->struct_size >= sizeof(yams_scan_sharded_options_t) && opt->exchange_timeout_ms) timeout_ms = opt->exchange_timeout_ms;
    }
    if (n_lanes > 16 || collective > YAMS_SHARDED_COLLECTIVE_PEER || fence > YAMS_SHARDED_FENCE_OFF) return YAMS_ERR_INVALID_ARG;
    const int n_dev = yams_accel_device_count();
    if (n_dev <= 0) return YAMS_ERR_UNSUPPORTED; // no GPU: there is deliberately no CPU fallback
    bool distinct = true;
    for (uint32_t i = 0; i < n_shards; ++i) {
        if (devices[i] < 0 || devices[i] >= n_dev) return YAMS_ERR_INVALID_ARG;
        for (uint32_t j = 0; j < i; ++j) distinct &= devices[i] != devices[j];
    }
    // RCCL refuses two ranks on one device; a library the caller names decides for itself
    if (collective == YAMS_SHARDED_COLLECTIVE_RCCL && !distinct && library.empty()) return YAMS_ERR_INVALID_ARG;

    auto* s = new yams_scan_sharded();
    s->n = n_shards; s->n_lanes = n_lanes; s->timeout_ms = timeout_ms;
    s->device.assign(devices, devices + n_shards);
    s->coll_next.assign(n_shards, 0);
    auto bail = [&](yams_status_t st) { destroy_handle(s); return st; };

    // the communicator: one process, one ncclCommInitAll over the shard devices
    const bool want_rccl = collective == YAMS_SHARDED_COLLECTIVE_RCCL ||
                           (collective == YAMS_SHARDED_COLLECTIVE_AUTO && (distinct || !library.empty()) && n_shards >= 2);
    s->mode = n_shards >= 2 ? kPeer : kNone;
    if (want_rccl) {
        Rccl& R = rccl(library);
        std::string why;
        if (!R.ok()) why = (library.empty() ? std::string("librccl.so.1") : library) + " could not be loaded: " + R.error;
        else {
            s->comm.assign(n_shards, nullptr);
            s->comm_mu.clear();
            for (uint32_t i = 0; i < n_shards; ++i) s->comm_mu.emplace_back(new std::mutex());
            const ncclResult_t r = R.CommInitAll(s->comm.data(), static_cast<int>(n_shards), s->device.data());
            if (r != ncclSuccess) { why = std::string("ncclCommInitAll failed: ") + R.GetErrorString(r); s->comm.clear(); (void)hipGetLastError(); }
        }
        if (why.empty()) { s->mode = kRccl; s->R = &R; }
        else if (collective == YAMS_SHARDED_COLLECTIVE_RCCL) {
            std::fprintf(stderr, "[yams_mi355x_accel] %s\n", why.c_str());
            return bail(YAMS_ERR_UNSUPPORTED);
        } else s->fallback_reason = why; // AUTO: device-to-device copies of the records; reported by ..._info_json
    }
    s->fenced = n_shards >= 2 && fence == YAMS_SHARDED_FENCE_AUTO;
    s->fence_ev.assign(n_shards, nullptr);
    s->fence_armed.assign(n_shards, 0);

    for (uint32_t i = 0; i < n_shards; ++i) {
        yams_accel_gate*& g = s->gates[devices[i]];
        if (!g && yams_accel_gate_create(devices[i], &g) != YAMS_OK) return bail(YAMS_ERR_INTERNAL);
        if (s->fenced) {
            (void)hipSetDevice(devices[i]);
            if (hipEventCreateWithFlags(&s->fence_ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return bail(YAMS_ERR_INTERNAL); }
        }
    }
    for (uint32_t li = 0; li < n_lanes; ++li) {
        s->lanes.emplace_back(new Lane());
        Lane& L = *s->lanes.back();
        L.sh.resize(n_shards);
        for (uint32_t i = 0; i < n_shards; ++i) {
            ShardLane& SL = L.sh[i];
            const yams_status_t st = yams_accel_ctx_create(devices[i], nullptr, &SL.ctx);
            if (st != YAMS_OK) return bail(st);
            (void)yams_accel_ctx_set_gate(SL.ctx, s->gates[devices[i]]);
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&SL.side, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); return bail(YAMS_ERR_INTERNAL); }
        }
        (void)hipSetDevice(devices[0]);
        if (yams_accel_ctx_create(devices[0], L.sh[0].side, &L.merge_ctx) != YAMS_OK) return bail(YAMS_ERR_INTERNAL);
        if (hipEventCreate(&L.done) != hipSuccess || hipEventCreate(&L.ex_begin) != hipSuccess) { (void)hipGetLastError(); return bail(YAMS_ERR_INTERNAL); }
    }
    if (s->mode == kPeer) {
        // records travel device-to-device: enable peer access towards the merge device where the
        // hardware offers it (xGMI); without it hipMemcpyPeerAsync stages through the host
        const int d0 = s->device[0];
        for (uint32_t i = 1; i < n_shards; ++i) {
            const int di = s->device[i];
            if (di == d0) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, d0) == hipSuccess && can) {
                (void)hipSetDevice(di);
                if (hipDeviceEnablePeerAccess(d0, 0) != hipSuccess) (void)hipGetLastError(); // already enabled, or not permitted: staged copies still work
            } else (void)hipGetLastError();
        }
    }
    for (uint32_t li = 0; li < n_lanes; ++li)
        for (uint32_t i = 0; i < n_shards; ++i) s->lanes[li]->sh[i].worker = std::thread(worker_main, s, i, li);
    (void)hipSetDevice(s->device[0]);
    *out = s;
    return YAMS_OK;
}

// pinned staging of a lane grows on the caller's thread while the lane is idle
bool grow_pinned(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return true;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = (bytes + bytes / 4 + 4095) & ~static_cast<size_t>(4095);
    if (ya_host_malloc(p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
    *cap = want;
    return true;
}
// ... and so do its device buffers (the current device is the buffer's; `quiet` = a stream whose work may still use it)
bool grow_device(void** p, size_t* cap, size_t bytes, hipStream_t quiet) {
    if (*cap >= bytes && *p) return true;
    if (*p) { (void)hipStreamSynchronize(quiet); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = ((bytes ? bytes : 16) + bytes / 4 + 255) & ~static_cast<size_t>(255);
    if (ya_malloc(p, want) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
    *cap = want;
    return true;
}

// When must lane L's batch be through?  The exchange timeout of the handle, counted from submit, plus what the scan of the
// largest shard may honestly take: (queries x rows x dim) at a tenth of the filter's measured rate (1e14 multiply-adds a
// second), or at the exhaustive fp64 kernel's (2e11) when the caller asks for that path.
std::chrono::steady_clock::time_point batch_deadline(const yams_scan_sharded* s, const Lane& L) {
    double work = 0.0;
    for (const yams_scan_corpus_t& v : L.views) work = std::max(work, static_cast<double>(v.n_rows) * L.nq * std::max<uint32_t>(L.dim, 1u));
    const bool exhaustive = (L.prm.flags & YAMS_SCAN_FLAG_FORCE_EXACT) != 0;
    const double allowance_ms = L.trivial ? 0.0 : work / (exhaustive ? 2.0e8 : 1.0e11);
    const double total = static_cast<double>(s->timeout_ms) + std::min(allowance_ms, 3.6e6);
    return std::chrono::steady_clock::now() + std::chrono::milliseconds(static_cast<int64_t>(total));
}

yams_status_t submit_impl(yams_scan_sharded* s, uint32_t lane, const yams_scan_corpus_t* shards, const float* queries_host,
                          uint32_t n_queries, const yams_scan_params_t* params, const uint32_t* rank_of_row, int64_t rank_row_base,
                          uint32_t flags) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    if (lane >= s->n_lanes) return set_error(s, YAMS_ERR_INVALID_ARG, "no such lane");
    Lane& L = *s->lanes[lane];
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->stuck) { s->last_error = "the handle is stuck (" + s->stuck_why + "): destroy it"; return YAMS_ERR_TIMEOUT; }
        if (!L.acquired || L.submitted) { s->last_error = "lane is not acquired, or its batch has not been waited for"; return YAMS_ERR_INVALID_ARG; }
    }
    if (!shards || !params) return set_error(s, YAMS_ERR_INVALID_ARG, "null shards/params");
    if (n_queries && !queries_host) return set_error(s, YAMS_ERR_INVALID_ARG, "null queries");
    const uint32_t n = s->n;
    const uint32_t dim = shards[0].dim;
    for (uint32_t i = 1; i < n; ++i)
        if (shards[i].dim != dim) return set_error(s, YAMS_ERR_INVALID_ARG, "shards disagree on the dimension");
    L.nq = n_queries; L.dim = dim; L.prm = *params; L.l2 = params->metric == YAMS_SCAN_L2;
    L.rank_of_row = rank_of_row; L.rank_row_base = rank_row_base;
    L.want_diag = (flags & YAMS_SHARDED_SUBMIT_DIAG) != 0;
    L.merge_issued = false; L.merge_st = YAMS_OK; L.merge_err.clear();
    // empty results before the query is validated (:4123-4126): nothing runs on the devices
    L.trivial = n_queries == 0 || params->k == 0 || dim == 0;
    if (!L.trivial) {
        if (params->k > YAMS_SCAN_MAX_K) return set_error(s, YAMS_ERR_UNSUPPORTED, "k exceeds YAMS_SCAN_MAX_K");
        if (s->mode != kNone && static_cast<uint64_t>(n) * params->k > 8192) return set_error(s, YAMS_ERR_UNSUPPORTED, "n_shards * k exceeds 8192");
        const size_t nq = n_queries, k = params->k;
        // records that travel carry distances only under L2 (the merge derives 1 - similarity for cosine); a lone
        // shard's record is the result itself and always has them
        yams_scan_record_layout(n_queries, params->k, (L.l2 || s->mode == kNone) ? 1 : 0, 0, &L.lay);
        L.stride = L.lay.bytes;
        int dev0 = s->device[0];
        (void)hipGetDevice(&dev0); // the caller's current device is restored below
        if (!grow_pinned(reinterpret_cast<void**>(&L.h_queries), &L.h_queries_cap, nq * dim * 4) ||
            !grow_pinned(reinterpret_cast<void**>(&L.h_out), &L.h_out_cap, out_layout(nq, k).bytes))
            return set_error(s, YAMS_ERR_RESOURCE_EXHAUSTED, "pinned staging could not be allocated");
        // device buffers of the batch are sized HERE, on the caller's thread while the lane is idle, so that no rank
        // can drop out of an exchange later for want of memory
        const size_t need = static_cast<size_t>(L.stride) * n;
        for (uint32_t i = 0; i < n; ++i) {
            ShardLane& SL = L.sh[i];
            (void)hipSetDevice(s->device[i]);
            bool ok = grow_device(reinterpret_cast<void**>(&SL.queries), &SL.queries_cap, nq * dim * 4, SL.ctx->stream) &&
                      grow_device(reinterpret_cast<void**>(&SL.rec), &SL.rec_cap, static_cast<size_t>(L.stride), SL.side);
            // receive buffers of the exchange: every rank under RCCL, the root's device otherwise
            if (ok && s->mode != kNone && (s->mode == kRccl || i == 0))
                ok = grow_device(reinterpret_cast<void**>(&SL.gathered), &SL.gathered_cap, need, SL.side);
            if (!ok) {
                (void)hipSetDevice(dev0);
                return set_error(s, YAMS_ERR_RESOURCE_EXHAUSTED, "device buffers of the batch could not be allocated");
            }
        }
        (void)hipSetDevice(dev0);
        std::memcpy(L.h_queries, queries_host, nq * dim * 4);
        L.views.assign(shards, shards + n);
        L.st.assign(n, YAMS_OK); L.err.assign(n, std::string());
        L.reported.assign(n, 0); L.ex_timed = false;
        L.dg.assign(n, yams_scan_diag_t{});
    }
    // One shard and a handful of queries (the reference's everyday call is ONE query): latency is the product, and two
    // thread hand-offs (wake the worker, wake the waiter) cost more than the search.  The lane is this caller's, its
    // worker is idle: the batch runs right here, on the caller's thread; wait() finds it finished.
    if (!L.trivial && s->mode == kNone && n_queries <= 16) {
        int dev_before = s->device[0];
        (void)hipGetDevice(&dev_before);
        (void)hipSetDevice(s->device[0]);
        std::string err;
        yams_status_t st;
        try { st = run_shard(s, L, 0, err); }
        catch (...) { st = YAMS_ERR_INTERNAL; err = "exception in the shard scan"; }
        (void)hipSetDevice(dev_before);
        std::lock_guard<std::mutex> lk(s->mu);
        L.st[0] = st; L.err[0] = std::move(err);
        L.pending = 0;
        L.ordered = false;
        L.submitted = true;
        ++s->batches;
        return YAMS_OK;
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        L.submitted = true;
        L.ordered = false;
        L.deadline = batch_deadline(s, L);
        if (!L.trivial) {
            L.pending = n;
            L.peer_left = n;
            if (s->mode != kNone) { L.seq = s->next_seq++; L.ordered = true; }
            ++L.job;
        }
    }
    if (!L.trivial) { ++s->batches; s->cv_job.notify_all(); }
    return YAMS_OK;
}

yams_status_t wait_impl(yams_scan_sharded* s, uint32_t lane, float* out_scores_host, int64_t* out_rows_host, uint32_t* out_counts_host,
                        float* out_dist_host, yams_scan_diag_t* diag) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    if (lane >= s->n_lanes) return set_error(s, YAMS_ERR_INVALID_ARG, "no such lane");
    Lane& L = *s->lanes[lane];
    {
        std::unique_lock<std::mutex> lk(s->mu);
        if (!L.acquired || !L.submitted) { s->last_error = "nothing was submitted on this lane"; return YAMS_ERR_INVALID_ARG; }
        auto through = [&] { return L.trivial || L.pending == 0; };
        if (s->timeout_ms == UINT32_MAX) s->cv_done.wait(lk, through);
        else if (!s->cv_done.wait_until(lk, L.deadline, through)) {
            // the lane stays acquired + submitted: its buffers are still in use by whatever hangs
            std::ostringstream os;
            os << "batch " << (L.ordered ? L.seq : 0) << " on lane " << lane << ": shard(s)";
            for (uint32_t i = 0; i < s->n; ++i) if (!L.reported[i]) os << ' ' << i << "(device " << s->device[i] << ")";
            os << " not through scan + exchange " << s->timeout_ms << " ms (+ the scan's allowance) after submit (collective "
               << (s->mode == kRccl ? "rccl" : (s->mode == kPeer ? "peer_copy" : "none")) << ", " << s->n << " ranks, "
               << s->collectives.load() << " exchanges issued before): a rank is missing from a collective, or a device hangs";
            lk.unlock();
            declare_stuck(s, os.str());
            return YAMS_ERR_TIMEOUT;
        }
    }
    // from here on the lane goes back to the pool on EVERY path, an exception included (a lane that stays acquired
    // would, once all of them are gone, block the next caller in lane_acquire for good)
    struct Release {
        yams_scan_sharded* s; Lane& L; yams_status_t st = YAMS_ERR_INTERNAL; std::string msg = "exception while a batch was collected";
        bool keep = false;
        ~Release() {
            {
                std::lock_guard<std::mutex> lk(s->mu);
                if (st != YAMS_OK) s->last_error = msg;
                if (!keep) L.acquired = L.submitted = false;
            }
            s->cv_lane.notify_one();
        }
    } rel{s, L};
    auto release = [&](yams_status_t st, const std::string& m) { rel.st = st; rel.msg = m; return st; };
    if (diag) std::memset(diag, 0, sizeof(*diag));
    const size_t nq = L.nq, k = L.prm.k;
    if (L.trivial) {
        if (nq && !out_counts_host) return release(YAMS_ERR_INVALID_ARG, "null out_counts");
        if (nq) std::memset(out_counts_host, 0, nq * 4);
        return release(YAMS_OK, "");
    }
    if (L.merge_issued) {   // also when a shard failed: the lane's buffers must be quiet before reuse
        if (s->timeout_ms == UINT32_MAX) (void)hipEventSynchronize(L.done);
        else {
            // (a batch is through in milliseconds: the first 50 ms are polled without sleeping — what hipEventSynchronize
            //  does as well — so that the deadline costs the caller's turn-around nothing; only a wait that is already
            //  long sleeps between looks)
            // (the batch's ONE deadline; a batch whose scans used most of their allowance still gets a quarter of the exchange
            //  timeout for the exchange itself: the caller waits at most 1.25 x what submit promised, not 2 x)
            const auto t_wait = std::chrono::steady_clock::now();
            const auto deadline = std::max(L.deadline, t_wait + std::chrono::milliseconds(s->timeout_ms / 4 + 1));
            const auto spin_until = t_wait + std::chrono::milliseconds(50);
            for (;;) {
                const hipError_t e = hipEventQuery(L.done);
                if (e == hipSuccess) break;
                if (e != hipErrorNotReady) { (void)hipGetLastError(); break; }
                const auto now = std::chrono::steady_clock::now();
                if (now > deadline) {
                    std::ostringstream os;
                    os << "batch " << L.seq << " on lane " << lane << ": all-gather + merge not complete on the root shard (device "
                       << s->device[0] << ") by the batch's deadline (" << s->timeout_ms << " ms exchange timeout; collective " << (s->mode == kRccl ? "rccl" : "peer_copy")
                       << ", " << s->n << " ranks, " << s->exchanges_timed.load() << " exchanges completed before): a peer never "
                          "joined the collective, or the link is down";
                    rel.keep = true; // buffers still in use: the lane is not handed out again
                    declare_stuck(s, os.str());
                    return release(YAMS_ERR_TIMEOUT, os.str());
                }
                if (now < spin_until) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        float ms = 0.f;
        if (L.ex_timed && hipEventElapsedTime(&ms, L.ex_begin, L.done) == hipSuccess && ms >= 0.f) {
            const uint64_t us = static_cast<uint64_t>(ms * 1000.0f + 0.5f);
            ++s->exchanges_timed; s->exchange_us_sum += us;
            uint64_t prev = s->exchange_us_max.load();
            while (us > prev && !s->exchange_us_max.compare_exchange_weak(prev, us)) {}
        } else (void)hipGetLastError();
    }
    for (uint32_t i = 0; i < s->n; ++i)
        if (L.st[i] != YAMS_OK) return release(L.st[i], L.err[i]); // a batch fails as a whole (:1635-1647)
    if (s->mode != kNone && L.merge_st != YAMS_OK) return release(L.merge_st, L.merge_err);
    if (!out_counts_host || !out_scores_host || !out_rows_host) return release(YAMS_ERR_INVALID_ARG, "null outputs");
    const OutLayout o = out_layout(nq, k);
    std::memcpy(out_counts_host, L.h_out + o.counts, nq * 4);
    std::memcpy(out_scores_host, L.h_out + o.scores, nq * k * 4);
    std::memcpy(out_rows_host, L.h_out + o.rows, nq * k * 8);
    if (out_dist_host) std::memcpy(out_dist_host, L.h_out + o.dist, nq * k * 4); // L2 distance, or 1 - similarity
    if (diag && L.want_diag) {
        if (s->n == 1) *diag = L.dg[0];
        diag->used_exact_scan = 1; diag->rows_visited_observed = 1;
        if (s->n > 1)
            for (uint32_t i = 0; i < s->n; ++i) {
                const yams_scan_diag_t& d = L.dg[i];
                diag->rows_visited += d.rows_visited;
                diag->exact_distance_evaluations += d.exact_distance_evaluations;
                diag->filter_candidates += d.filter_candidates;
                diag->rescored_rows += d.rescored_rows;
                diag->widened_queries += d.widened_queries;
                diag->exact_fallback_queries += d.exact_fallback_queries;
                diag->escalated_queries += d.escalated_queries;
                diag->path = std::max(diag->path, d.path);
                diag->filter_tier = std::max(diag->filter_tier, d.filter_tier);
            }
        uint64_t ret = 0;
        for (size_t q = 0; q < nq; ++q) ret += out_counts_host[q];
        diag->returned_rows = ret;
    }
    return release(YAMS_OK, "");
}

} // namespace

// No exception crosses the C ABI (model_provider_v1.h:44-49; the reference's plugins wrap every entry,
// plugins/onnx/model_provider.cpp:51-105).
#define YAMS_GUARD(expr) try { return (expr); } catch (const std::bad_alloc&) { return YAMS_ERR_INTERNAL; } catch (...) { return YAMS_ERR_INTERNAL; }

extern "C" yams_status_t yams_scan_sharded_create_ex(const int* devices, uint32_t n_shards, const yams_scan_sharded_options_t* options,
                                                     yams_scan_sharded** out) {
    YAMS_GUARD(create_impl(devices, n_shards, options, out));
}

extern "C" yams_status_t yams_scan_sharded_create(const int* devices, uint32_t n_shards, yams_scan_sharded** out) {
    YAMS_GUARD(create_impl(devices, n_shards, nullptr, out));
}

extern "C" void yams_scan_sharded_destroy(yams_scan_sharded* s) {
    if (!s) return;
    try { destroy_handle(s); } catch (...) {}
}

extern "C" uint32_t yams_scan_sharded_count(const yams_scan_sharded* s) { return s ? s->n : 0u; }
extern "C" uint32_t yams_scan_sharded_lanes(const yams_scan_sharded* s) { return s ? s->n_lanes : 0u; }

extern "C" yams_accel_ctx* yams_scan_sharded_lane_ctx(yams_scan_sharded* s, uint32_t shard, uint32_t lane) {
    return (s && shard < s->n && lane < s->n_lanes) ? s->lanes[lane]->sh[shard].ctx : nullptr;
}
extern "C" yams_accel_ctx* yams_scan_sharded_ctx(yams_scan_sharded* s, uint32_t shard) { return yams_scan_sharded_lane_ctx(s, shard, 0); }

extern "C" const char* yams_scan_sharded_last_error(const yams_scan_sharded* s) {
    return s ? s->last_error.c_str() : "null sharded handle";
}

extern "C" yams_status_t yams_scan_sharded_info_json(yams_scan_sharded* s, char** out_json) {
    if (!s || !out_json) return YAMS_ERR_INVALID_ARG;
    try {
        std::ostringstream os;
        os << "{\"shards\":" << s->n << ",\"lanes\":" << s->n_lanes << ",\"devices\":[";
        for (uint32_t i = 0; i < s->n; ++i) os << (i ? "," : "") << s->device[i];
        os << "],\"collective\":\"" << (s->mode == kRccl ? "rccl" : (s->mode == kPeer ? "peer_copy" : "none")) << "\""
           << ",\"fenced\":" << (s->fenced ? "true" : "false");
        if (s->mode == kRccl && s->R) {
            const Rccl& R = *s->R;
            int ranks = static_cast<int>(s->comm.size());
            const char* src = "handle";
            if (R.CommCount && !s->comm.empty() && s->comm[0]) { int c = 0; if (R.CommCount(s->comm[0], &c) == ncclSuccess) { ranks = c; src = "ncclCommCount"; } }
            os << ",\"rccl_version\":" << R.version << ",\"rccl_library\":\"" << R.path << "\",\"communicator_ranks\":" << ranks
               << ",\"communicator_ranks_source\":\"" << src << "\"";
        }
        if (!s->fallback_reason.empty()) {
            std::string r = s->fallback_reason;
            for (char& ch : r) if (ch == '"' || ch == '\\' || ch == '\n') ch = ' ';
            os << ",\"rccl_unavailable\":\"" << r << "\"";
        }
        const uint64_t ex = s->exchanges_timed.load();
        os << ",\"batches\":" << s->batches.load() << ",\"collectives\":" << s->collectives.load()
           << ",\"fence_waits\":" << s->fence_waits.load() << ",\"exchanges_timed\":" << ex
           << ",\"exchange_ms\":" << (ex ? static_cast<double>(s->exchange_us_sum.load()) / 1000.0 / static_cast<double>(ex) : 0.0)
           << ",\"exchange_ms_max\":" << static_cast<double>(s->exchange_us_max.load()) / 1000.0
           << ",\"exchange_timeout_ms\":" << s->timeout_ms;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            os << ",\"stuck\":" << (s->stuck ? "true" : "false");
        }
        os << "}";
        const std::string str = os.str();
        char* buf = static_cast<char*>(std::malloc(str.size() + 1));
        if (!buf) return YAMS_ERR_INTERNAL;
        std::memcpy(buf, str.c_str(), str.size() + 1);
        *out_json = buf;
        return YAMS_OK;
    } catch (...) { return YAMS_ERR_INTERNAL; }
}

extern "C" yams_status_t yams_scan_sharded_lane_acquire(yams_scan_sharded* s, int wait, uint32_t* out_lane) {
    if (!s || !out_lane) return YAMS_ERR_INVALID_ARG;
    try {
        std::unique_lock<std::mutex> lk(s->mu);
        for (;;) {
            // a stuck handle hands out no lanes: the lanes of its hung batches never come back, a caller must not wait for them
            if (s->stuck) { s->last_error = "the handle is stuck (" + s->stuck_why + "): destroy it"; return YAMS_ERR_TIMEOUT; }
            for (uint32_t li = 0; li < s->n_lanes; ++li)
                if (!s->lanes[li]->acquired) { s->lanes[li]->acquired = true; s->lanes[li]->submitted = false; *out_lane = li; return YAMS_OK; }
            if (!wait) return YAMS_ERR_NOT_FOUND; // every lane has a batch in flight: wait() for one
            s->cv_lane.wait(lk);
        }
    } catch (...) { return YAMS_ERR_INTERNAL; }
}

extern "C" void yams_scan_sharded_lane_release(yams_scan_sharded* s, uint32_t lane) {
    if (!s || lane >= s->n_lanes) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        Lane& L = *s->lanes[lane];
        if (L.submitted) return; // a submitted batch is released by wait()
        L.acquired = false;
    }
    s->cv_lane.notify_one();
}

extern "C" yams_status_t yams_scan_sharded_submit(yams_scan_sharded* s, uint32_t lane, const yams_scan_corpus_t* shards,
                                                  const float* queries_host, uint32_t n_queries, const yams_scan_params_t* params,
                                                  const uint32_t* rank_of_row, int64_t rank_row_base, uint32_t flags) {
    YAMS_GUARD(submit_impl(s, lane, shards, queries_host, n_queries, params, rank_of_row, rank_row_base, flags));
}

extern "C" yams_status_t yams_scan_sharded_wait(yams_scan_sharded* s, uint32_t lane, float* out_scores_host, int64_t* out_rows_host,
                                                uint32_t* out_counts_host, float* out_dist_host, yams_scan_diag_t* diag) {
    YAMS_GUARD(wait_impl(s, lane, out_scores_host, out_rows_host, out_counts_host, out_dist_host, diag));
}

extern "C" yams_status_t yams_scan_sharded_topk_host(
    yams_scan_sharded* s, const yams_scan_corpus_t* shards, const float* queries_host, uint32_t n_queries,
    const yams_scan_params_t* params, const uint32_t* rank_of_row, int64_t rank_row_base,
    float* out_scores_host, int64_t* out_rows_host, uint32_t* out_counts_host, float* out_dist_host,
    yams_scan_diag_t* diag) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    if (diag) std::memset(diag, 0, sizeof(*diag));
    if (n_queries == 0 && shards && params) return YAMS_OK; // searchSimilarBatch on an empty batch (:1615-1617)
    uint32_t lane = 0;
    yams_status_t st = yams_scan_sharded_lane_acquire(s, 1, &lane);
    if (st != YAMS_OK) return st;
    st = yams_scan_sharded_submit(s, lane, shards, queries_host, n_queries, params, rank_of_row, rank_row_base,
                                  diag ? YAMS_SHARDED_SUBMIT_DIAG : 0u);
    if (st != YAMS_OK) { yams_scan_sharded_lane_release(s, lane); return st; }
    return yams_scan_sharded_wait(s, lane, out_scores_host, out_rows_host, out_counts_host, out_dist_host, diag);
}
