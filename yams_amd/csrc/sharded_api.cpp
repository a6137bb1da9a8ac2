// sharded_api.cpp — one search over a corpus that is row-sharded across several devices of this node,
// behind ONE C call (the in-process form of the multi-GPU split; bench.py's one-process-per-GPU form
// uses the same record layout and the same merge kernel behind an RCCL all-gather).
//
// Reference: SqliteVecBackend::searchSimilarBatch (src/vector/sqlite_vec_backend.cpp:1612-1647) sees
// one corpus; here every shard runs the exact scan on its own device with its own context and host
// thread, writes its per-query top-k as one packed record, the records are copied device-to-device
// (peer-to-peer over xGMI where enabled) next to each other on the first shard's device, and the
// k-way merge kernel produces the answer there.  The corpus itself never moves.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "accel_ctx.h"
#include "scan_launch.h"

using namespace yams_accel;

struct yams_scan_sharded {
    std::vector<yams_accel_ctx*> ctx; // one per shard (several shards may share a device)
    std::vector<int> device;
    std::string last_error;
};

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

extern "C" yams_status_t yams_scan_sharded_create(const int* devices, uint32_t n_shards, yams_scan_sharded** out) {
    if (!out) return YAMS_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_shards == 0 || n_shards > 64) return YAMS_ERR_INVALID_ARG;
    auto* s = new yams_scan_sharded();
    for (uint32_t i = 0; i < n_shards; ++i) {
        yams_accel_ctx* c = nullptr;
        const yams_status_t st = yams_accel_ctx_create(devices[i], nullptr, &c);
        if (st != YAMS_OK) {
            for (auto* p : s->ctx) yams_accel_ctx_destroy(p);
            delete s;
            return st;
        }
        s->ctx.push_back(c);
        s->device.push_back(devices[i]);
    }
    // records travel device-to-device: enable peer access towards the merge device where the
    // hardware offers it (xGMI); without it hipMemcpyPeerAsync stages through the host
    const int d0 = s->device[0];
    for (uint32_t i = 1; i < n_shards; ++i) {
        const int di = s->device[i];
        if (di == d0) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, di, d0) == hipSuccess && can) {
            (void)hipSetDevice(di);
            const hipError_t e = hipDeviceEnablePeerAccess(d0, 0);
            if (e != hipSuccess) (void)hipGetLastError(); // already enabled, or not permitted: staged copies still work
        } else {
            (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(d0);
    *out = s;
    return YAMS_OK;
}

extern "C" void yams_scan_sharded_destroy(yams_scan_sharded* s) {
    if (!s) return;
    for (auto* c : s->ctx) yams_accel_ctx_destroy(c);
    delete s;
}

extern "C" uint32_t yams_scan_sharded_count(const yams_scan_sharded* s) {
    return s ? static_cast<uint32_t>(s->ctx.size()) : 0u;
}

extern "C" yams_accel_ctx* yams_scan_sharded_ctx(yams_scan_sharded* s, uint32_t shard) {
    return (s && shard < s->ctx.size()) ? s->ctx[shard] : nullptr;
}

extern "C" const char* yams_scan_sharded_last_error(const yams_scan_sharded* s) {
    return s ? s->last_error.c_str() : "null sharded handle";
}

extern "C" yams_status_t yams_scan_sharded_topk_host(
    yams_scan_sharded* s, const yams_scan_corpus_t* shards, const float* queries_host, uint32_t n_queries,
    const yams_scan_params_t* params, const uint32_t* rank_of_row, int64_t rank_row_base,
    float* out_scores_host, int64_t* out_rows_host, uint32_t* out_counts_host, float* out_dist_host,
    yams_scan_diag_t* diag) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    auto failed = [&](yams_status_t st, const std::string& m) { s->last_error = m; return st; };
    if (!shards || !params) return failed(YAMS_ERR_INVALID_ARG, "null shards/params");
    if (diag) std::memset(diag, 0, sizeof(*diag));
    if (n_queries == 0) return YAMS_OK;
    if (!queries_host || !out_counts_host) return failed(YAMS_ERR_INVALID_ARG, "null queries/out_counts");
    const uint32_t n = static_cast<uint32_t>(s->ctx.size());
    if (n == 1) { // one shard: its own ordering (tie ranks included) is final, nothing to merge
        const yams_status_t st1 = yams_scan_topk_host(s->ctx[0], &shards[0], queries_host, n_queries, params, out_scores_host,
                                                      out_rows_host, out_counts_host, out_dist_host, diag);
        if (st1 != YAMS_OK) s->last_error = yams_accel_last_error(s->ctx[0]);
        return st1;
    }
    const uint32_t dim = shards[0].dim;
    for (uint32_t i = 1; i < n; ++i)
        if (shards[i].dim != dim) return failed(YAMS_ERR_INVALID_ARG, "shards disagree on the dimension");
    const size_t nq = n_queries, k = params->k;
    if (k == 0 || dim == 0) { // empty result before the query is validated (:4123-4126)
        std::memset(out_counts_host, 0, nq * 4);
        return YAMS_OK;
    }
    if (!out_scores_host || !out_rows_host) return failed(YAMS_ERR_INVALID_ARG, "null outputs");
    const bool l2 = params->metric == YAMS_SCAN_L2;
    yams_scan_record_layout_t lay;
    yams_scan_record_layout(n_queries, params->k, l2 ? 1 : 0, 0, &lay);
    const uint64_t stride = lay.bytes;

    yams_accel_ctx* c0 = s->ctx[0];
    (void)hipSetDevice(c0->device);
    unsigned char* d_gather = nullptr;
    if (ws_get(c0, "shard_gather", static_cast<size_t>(stride) * n, (void**)&d_gather) != YAMS_OK)
        return failed(YAMS_ERR_INTERNAL, yams_accel_last_error(c0));

    std::vector<yams_status_t> st(n, YAMS_OK);
    std::vector<yams_scan_diag_t> dg(n);
    auto work = [&](uint32_t i) {
        yams_accel_ctx* c = s->ctx[i];
        (void)hipSetDevice(c->device);
        float* d_q = nullptr; unsigned char* d_rec = nullptr;
        if ((st[i] = ws_get(c, "shard_queries", nq * dim * 4, (void**)&d_q)) != YAMS_OK) return;
        if ((st[i] = ws_get(c, "shard_record", static_cast<size_t>(stride), (void**)&d_rec)) != YAMS_OK) return;
        if (hipMemcpyAsync(d_q, queries_host, nq * dim * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            (void)hipGetLastError(); st[i] = fail(c, YAMS_ERR_INTERNAL, "query upload failed"); return;
        }
        yams_scan_params_t prm = *params;
        if (l2) prm.flags |= YAMS_SCAN_FLAG_DEFER_THRESHOLD; // vec0: the k nearest first, the threshold after the merge
        st[i] = yams_scan_topk_device(c, &shards[i], d_q, n_queries, &prm,
                                      reinterpret_cast<float*>(d_rec + lay.scores_off),
                                      reinterpret_cast<int64_t*>(d_rec + lay.rows_off),
                                      reinterpret_cast<uint32_t*>(d_rec + lay.counts_off),
                                      l2 ? reinterpret_cast<float*>(d_rec + lay.dist_off) : nullptr, nullptr, &dg[i]);
        if (st[i] != YAMS_OK) return;
        hipError_t e;
        if (c->device == c0->device)
            e = hipMemcpyAsync(d_gather + stride * i, d_rec, stride, hipMemcpyDeviceToDevice, c->stream);
        else
            e = hipMemcpyPeerAsync(d_gather + stride * i, c0->device, d_rec, c->device, stride, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); st[i] = fail(c, YAMS_ERR_INTERNAL, "record copy to the merge device failed"); }
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        th.reserve(n);
        for (uint32_t i = 0; i < n; ++i) th.emplace_back(work, i);
        for (auto& t : th) t.join();
    }
    for (uint32_t i = 0; i < n; ++i)
        if (st[i] != YAMS_OK) return failed(st[i], yams_accel_last_error(s->ctx[i])); // a batch fails as a whole (:1635-1647)

    (void)hipSetDevice(c0->device);
    float* d_s; int64_t* d_r; uint32_t* d_c; float* d_d;
    if (ws_get(c0, "shard_out_scores", nq * k * 4, (void**)&d_s) != YAMS_OK ||
        ws_get(c0, "shard_out_rows", nq * k * 8, (void**)&d_r) != YAMS_OK ||
        ws_get(c0, "shard_out_counts", nq * 4, (void**)&d_c) != YAMS_OK ||
        ws_get(c0, "shard_out_dist", nq * k * 4, (void**)&d_d) != YAMS_OK)
        return failed(YAMS_ERR_INTERNAL, yams_accel_last_error(c0));
    yams_status_t ms = yams_scan_merge_records_device(c0, n, n_queries, params, d_gather, stride, &lay, rank_of_row,
                                                      rank_row_base, d_s, d_r, d_c, d_d);
    if (ms != YAMS_OK) return failed(ms, yams_accel_last_error(c0));
    hipError_t e = hipMemcpyAsync(out_counts_host, d_c, nq * 4, hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores_host, d_s, nq * k * 4, hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_rows_host, d_r, nq * k * 8, hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess && out_dist_host) e = hipMemcpyAsync(out_dist_host, d_d, nq * k * 4, hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return failed(YAMS_ERR_INTERNAL, "result download failed"); }
    if (diag) {
        diag->used_exact_scan = 1; diag->rows_visited_observed = 1;
        for (uint32_t i = 0; i < n; ++i) {
            diag->rows_visited += dg[i].rows_visited;
            diag->exact_distance_evaluations += dg[i].exact_distance_evaluations;
            diag->filter_candidates += dg[i].filter_candidates;
            diag->rescored_rows += dg[i].rescored_rows;
            diag->widened_queries += dg[i].widened_queries;
            diag->exact_fallback_queries += dg[i].exact_fallback_queries;
            diag->escalated_queries += dg[i].escalated_queries;
            diag->path = std::max(diag->path, dg[i].path);
            diag->filter_tier = std::max(diag->filter_tier, dg[i].filter_tier);
        }
        uint64_t ret = 0;
        for (size_t q = 0; q < nq; ++q) ret += out_counts_host[q];
        diag->returned_rows = ret;
    }
    return YAMS_OK;
}
