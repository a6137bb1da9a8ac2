// scan_api.cpp — host orchestration of the exact vector scan behind the C ABI.
//
// Mirrors the control flow of SqliteVecBackend::Impl::searchSimilarBatch ->
// bruteForceSearchUnlocked (src/vector/sqlite_vec_backend.cpp:1612-1647, 4115-4331) with the
// corpus read ONCE per batch instead of once per query:
//   prep (validate, fp64 norms)  -> MFMA sample pass -> thresholds -> MFMA filter pass
//   -> per-query candidate select -> fp64 re-score + verification (-> widen -> exhaustive fp64).
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "accel_ctx.h"
#include "scan_launch.h"

using namespace yams_accel;

namespace {

constexpr uint64_t kMfmaMinRows = 4096;      // below this the exhaustive fp64 kernel is used
constexpr uint64_t kExactKeyBudget = 1ull << 31; // bytes of fp64-path keys per batch of queries

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// passes: MFMA passes of the filter (0 = exact f32 kernel, 1 = RNE bf16, 3 = split bf16).  A looser
// filter needs more candidates re-scored before the proof can succeed, not a different threshold.
// l2_band: the single-pass L2 filter of the bf16 tier (its score carries +E itself, the band around the k-th best
// is 2E wide); the int8 tier's L2 bound is as tight as its cosine bound and plans like it.
// depth (int8 tier, learnt per corpus: TierHint): 1 = stage 1 re-scores the whole list, 2 = and the lists are half as long again.
ScanPlan make_plan(uint64_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, bool bf16, int passes, bool l2_band, int depth = 0) {
    ScanPlan p;
    p.n_rows = n_rows; p.dim = dim; p.n_queries = nq;
    p.tile_rows = bf16 ? 256 : kTileRows;
    p.tile_queries = bf16 ? 256 : kTileQueries;
    p.n_tiles = static_cast<uint32_t>((n_rows + p.tile_rows - 1) / p.tile_rows);
    p.n_qtiles = (nq + p.tile_queries - 1) / p.tile_queries;
    p.kprime = (passes == 1)
                   ? std::min<uint32_t>(round_up(l2_band ? 6 * k + 128 : 3 * k + 64, 32), kRescoreMax)
                   : std::min<uint32_t>(round_up(k + std::max<uint32_t>(16, k / 4), 32), kRescoreMax);
    const uint64_t s_target = std::min<uint64_t>(n_rows, std::max<uint64_t>(n_rows / 64, 8192));
    uint32_t want_tiles = static_cast<uint32_t>((s_target + p.tile_rows - 1) / p.tile_rows);
    if (want_tiles == 0) want_tiles = 1;
    p.sample_stride = std::max<uint32_t>(1, p.n_tiles / want_tiles);
    p.n_sample_tiles = (p.n_tiles + p.sample_stride - 1) / p.sample_stride;
    p.n_filter_tiles = p.n_tiles - p.n_sample_tiles;
    p.sample_rows = static_cast<uint64_t>(p.n_sample_tiles) * p.tile_rows;
    p.n_groups = static_cast<uint32_t>(p.sample_rows / kGroupRows);
    // tau = the tau_rank-th best sample value, i.e. about the (tau_rank * stride)-th best overall:
    // the lists must hold comfortably more than the kprime candidates stage 1 wants
    // (list size ~ Gamma(tau_rank) scaled to tau_rank * stride: with rank 16 and an expectation of
    // 2.7x what the proof needs, the chance of a list too short to prove is ~1e-7 per query)
    const uint32_t need_rank = (2 * p.kprime + p.kprime / 2 + 32 + p.sample_stride - 1) / p.sample_stride;
    p.tau_rank = std::min<uint32_t>(std::max<uint32_t>(16, round_up(need_rank, 16)), kRescoreMax);
    // (a list longer than what can be re-scored does not hurt the proof: the best kRescoreMax bounds are re-scored and the next
    // one is the proof's threshold; a list cut short by a threshold the sample put too high does)
    if (depth >= 2 && p.tau_rank * 2 <= 256 && p.n_groups >= p.tau_rank * 2) p.tau_rank *= 2;
    if (depth >= 1) p.kprime = kRescoreMax;
    // expected list length tau_rank * stride (relative spread ~1/sqrt(tau_rank)): 4x is > 15 sigma
    uint64_t cap = std::max<uint64_t>(4096, 4ull * p.tau_rank * p.sample_stride);
    if (p.n_groups < p.tau_rank) cap = std::max<uint64_t>(cap, n_rows); // threshold is -inf
    cap = std::min<uint64_t>(cap, std::max<uint64_t>(n_rows, 4096));
    p.list_cap = round_up(static_cast<uint32_t>(cap), 256);
    return p;
}

struct ScanIo {
    const yams_scan_corpus_t* corpus;
    const float* queries; uint32_t nq;
    yams_scan_params_t prm;
    float* out_scores; int64_t* out_rows; uint32_t* out_counts; float* out_dist; uint32_t* out_ranks;
};

// Exhaustive fp64 path for a subset of queries (slots -> query indices in qmap_host; nullptr = all).
yams_status_t run_exact(yams_accel_ctx* ctx, const ScanIo& io, const double* d_qnorm,
                        const std::vector<uint32_t>* subset, uint32_t* d_status,
                        unsigned long long* d_stat, const uint32_t* d_rows_sel = nullptr,
                        uint64_t n_sel = 0) {
    const auto& c = *io.corpus;
    const uint32_t total = subset ? static_cast<uint32_t>(subset->size()) : io.nq;
    if (total == 0) return YAMS_OK;
    const uint32_t keep = io.prm.k; // exact keys: the best k ARE the answer
    const uint64_t n_items = d_rows_sel ? n_sel : c.n_rows; // rows that get a key
    const uint64_t key_stride = std::max<uint64_t>(n_items, 1);
    uint32_t batch = static_cast<uint32_t>(std::max<uint64_t>(1, kExactKeyBudget / (key_stride * 8)));
    batch = std::min(batch, total);
    uint64_t* d_keys; uint64_t* d_work; uint32_t* d_qmap;
    const uint32_t chunks = static_cast<uint32_t>((key_stride + kSelectCap - 1) / kSelectCap);
    YA_TRY(ws_get(ctx, "exact_keys", static_cast<size_t>(batch) * key_stride * 8, (void**)&d_keys));
    YA_TRY(ws_get(ctx, "exact_work", static_cast<size_t>(2) * batch * chunks * keep * 8, (void**)&d_work));
    YA_TRY(ws_get(ctx, "exact_qmap", static_cast<size_t>(total) * 4, (void**)&d_qmap));
    std::vector<uint32_t> ident;
    const uint32_t* qm_host;
    if (subset) qm_host = subset->data();
    else { ident.resize(total); for (uint32_t i = 0; i < total; ++i) ident[i] = i; qm_host = ident.data(); }
    YA_HIP(ctx, hipMemcpyAsync(d_qmap, qm_host, static_cast<size_t>(total) * 4, hipMemcpyHostToDevice, ctx->stream));
    const int metric = static_cast<int>(io.prm.metric);
    // In the L2 (vec0) contract the cosine threshold applies after the top-k (:4506-4510).
    for (uint32_t b0 = 0; b0 < total; b0 += batch) {
        const uint32_t nb = std::min(batch, total - b0);
        if (n_items > 0) {
            TimedRegion tr(ctx, "exact_keys");
            YA_HIP(ctx, launch_exact_keys(ctx->stream, metric, c.rows, c.n_rows, c.dim, io.queries,
                                          d_qnorm, c.tie_rank, c.row_mask, d_rows_sel, n_sel,
                                          d_qmap + b0, nb, io.prm.similarity_threshold, io.prm.flags,
                                          d_keys, key_stride));
            tr.end();
        }
        const uint64_t* res; uint64_t res_stride;
        YA_HIP(ctx, launch_topk_keys(ctx->stream, d_keys, key_stride,
                                     static_cast<uint32_t>(n_items), nb, keep, d_work, &res,
                                     &res_stride));
        RescoreLaunch R{};
        R.rows = c.rows; R.n_rows = c.n_rows; R.dim = c.dim; R.queries = io.queries; R.qnorm = d_qnorm;
        R.tie_rank = c.tie_rank; R.rank_row = c.tie_rank ? c.rank_row : nullptr; R.row_base = c.row_base;
        R.stripe_rows = c.stripe_rows; R.n_stripes = c.n_stripes; R.stripe_index = c.stripe_index;
        R.cand = res; R.cand_stride = res_stride; R.n_cand = std::min<uint32_t>(keep, kRescoreMax);
        R.tau = nullptr; R.list_count = nullptr; R.list_cap = 0; R.all_rows_listed = 1;
        R.qmap = d_qmap + b0; R.n_slots = nb; R.k = io.prm.k; R.threshold = io.prm.similarity_threshold;
        R.flags = io.prm.flags & ~kRescoreFlagPqRerank; R.err_bound = 0.0;
        R.out_scores = io.out_scores; R.out_rows = io.out_rows; R.out_counts = io.out_counts;
        R.out_dist = io.out_dist; R.out_ranks = io.out_ranks; R.out_status = d_status;
        R.stat_rescored = d_stat;
        YA_HIP(ctx, launch_rescore(ctx->stream, metric, R));
    }
    return YAMS_OK;
}

// Small corpora and few queries (BASELINE config 1; the reference's most common call is ONE query,
// search_vector_pipeline.cpp:221 -> sqlite_vec_backend.cpp:1436-1454): every row scored in fp64 in the reference's
// order and reduced to the final top-k by ONE launch (scan_small_kernel.hip), one look at the query flags.
constexpr uint64_t kSmallRows = 16384;
constexpr uint32_t kSmallQueries = 16;
bool small_scan_applies(const yams_scan_corpus_t& c, uint32_t nq, const yams_scan_params_t& p) {
    if (c.n_rows == 0 || c.n_rows > kSmallRows || nq > kSmallQueries) return false;
    // the fused kernel scores in fp64 only: L2 under an fp32 accumulation goes through the general pipeline
    if (p.metric == YAMS_SCAN_L2 && (p.flags & YAMS_SCAN_FLAG_L2_ACC_MASK)) return false;
    // callers that name a filter tier or the exhaustive pipeline get what they name
    if (p.flags & (YAMS_SCAN_FLAG_FORCE_EXACT | YAMS_SCAN_FLAG_F32_FILTER | YAMS_SCAN_FLAG_SPLIT_FILTER | YAMS_SCAN_FLAG_WIDE_TILE |
                   YAMS_SCAN_FLAG_NO_I8_FILTER | YAMS_SCAN_FLAG_RESIDENT_QUERIES)) return false;
    if ((c.dim & 31u) || c.dim > 1024 || (reinterpret_cast<uintptr_t>(c.rows) & 15u)) return false;
    const uint32_t n_wg = static_cast<uint32_t>((c.n_rows + 255) / 256), kk = std::min<uint32_t>(p.k, 256);
    return p.k <= 256 && static_cast<uint64_t>(n_wg) * kk <= small_scan_max_survivors();
}

yams_status_t small_scan(yams_accel_ctx* ctx, const yams_scan_corpus_t* corpus, const float* queries, uint32_t nq,
                         const yams_scan_params_t* params, float* out_scores, int64_t* out_rows, uint32_t* out_counts,
                         float* out_dist, uint32_t* out_ranks, yams_scan_diag_t* diag) {
    hipStream_t st = ctx->stream;
    const uint32_t n_wg = static_cast<uint32_t>((corpus->n_rows + 255) / 256), kk = std::min<uint32_t>(params->k, 256);
    const uint32_t qb = nq == 1 ? 1 : 4;
    SmallScanArgs a{};
    a.rows = corpus->rows; a.n_rows = static_cast<uint32_t>(corpus->n_rows); a.dim = corpus->dim; a.queries = queries; a.nq = nq;
    a.tie_rank = corpus->tie_rank; a.rank_row = corpus->rank_row; a.row_mask = corpus->row_mask; a.row_base = corpus->row_base;
    a.stripe_rows = corpus->stripe_rows; a.n_stripes = corpus->n_stripes; a.stripe_index = corpus->stripe_index;
    a.k = params->k; a.kk = kk; a.sort_cap = 0;
    a.threshold = params->similarity_threshold; a.flags = params->flags;
    a.out_scores = out_scores; a.out_rows = out_rows; a.out_counts = out_counts; a.out_dist = out_dist; a.out_ranks = out_ranks;
    YA_TRY(ws_get(ctx, "small_keys", static_cast<size_t>(nq) * n_wg * kk * 8, (void**)&a.part_key));
    YA_TRY(ws_get(ctx, "small_aux", static_cast<size_t>(nq) * n_wg * kk * 4, (void**)&a.part_aux));
    YA_TRY(ws_get(ctx, "small_flags", static_cast<size_t>(nq) * 4, (void**)&a.qflags));
    {   // the ticket counters are zero between launches (the kernel leaves them so): cleared when first allocated
        const std::string name = ctx->ws_ns + "small_counter";
        const bool fresh = ctx->bufs.find(name) == ctx->bufs.end();
        YA_TRY(ws_get(ctx, "small_counter", 64, (void**)&a.counter));
        if (fresh) YA_HIP(ctx, hipMemsetAsync(a.counter, 0, 64, st));
    }
    uint32_t* h_pin;
    YA_TRY(pinned_get(ctx, static_cast<size_t>(nq) * 8 + 64, (void**)&h_pin));
#ifdef YAMS_ACCEL_MEASURE
    if (std::getenv("YAMS_ACCEL_SMALL_STAMPS")) YA_TRY(ws_get(ctx, "small_dbg", static_cast<size_t>(n_wg) * 64, (void**)&a.dbg));
#endif
    { TimedRegion tr(ctx, "small_scan");
      YA_HIP(ctx, launch_small_scan(st, static_cast<int>(params->metric), a, qb));
      tr.end(); }
    YA_HIP(ctx, hipMemcpyAsync(h_pin, a.qflags, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
    if (diag) YA_HIP(ctx, hipMemcpyAsync(h_pin + nq, out_counts, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
    YA_HIP(ctx, hipStreamSynchronize(st));
#ifdef YAMS_ACCEL_MEASURE
    if (a.dbg) if (const char* dump = std::getenv("YAMS_ACCEL_SMALL_STAMPS")) {
        std::vector<unsigned long long> h(static_cast<size_t>(n_wg) * 8);
        YA_HIP(ctx, hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    }
#endif
    for (uint32_t i = 0; i < nq; ++i) {
        const bool bad = (params->metric == YAMS_SCAN_COSINE) ? (h_pin[i] != 0) : ((h_pin[i] & 1u) != 0);
        if (bad) return fail(ctx, YAMS_ERR_INVALID_ARG, "Exact vector search requires a finite, non-zero query embedding");
    }
    if (diag) {
        const uint64_t n_eff = corpus->row_mask ? corpus->row_mask_count : corpus->n_rows;
        diag->used_exact_scan = 1; diag->rows_visited_observed = 1;
        diag->rows_visited = static_cast<uint64_t>(nq) * n_eff;
        diag->exact_distance_evaluations = static_cast<uint64_t>(nq) * n_eff;
        uint64_t ret = 0;
        for (uint32_t i = 0; i < nq; ++i) ret += h_pin[nq + i];
        diag->returned_rows = ret;
        diag->rescored_rows = static_cast<uint64_t>(nq) * n_eff; // every row was scored in fp64
        diag->path = 1; diag->filter_tier = 0;
    }
    return YAMS_OK;
}

// The RETRY run of the int8 tier (round 6): queries whose proof failed are filtered again with a threshold nobody has to
// estimate — the k-th best EXACT score stage 1 found for them (a lower bound of the final k-th best: no row whose upper bound
// lies below it can enter the result).  On clustered corpora the sampled threshold sits inside a cloud of near-equal scores
// and every query failed its proof; one more int8 sweep with this threshold lists exactly the rows that matter.
struct TauRetry {
    const float* forced_tau = nullptr;      // device [n_queries]: use instead of the sampled threshold
    std::vector<uint32_t>* unproven = nullptr; // out: queries (of this run) still unproven — the caller escalates them
};

// split_only: this is the escalation run of a batch whose single-pass filter left queries unproven.
yams_status_t scan_impl(yams_accel_ctx* ctx, const yams_scan_corpus_t* corpus, const float* queries,
                        uint32_t n_queries, const yams_scan_params_t* params, float* out_scores,
                        int64_t* out_rows, uint32_t* out_counts, float* out_dist,
                        uint32_t* out_ranks, yams_scan_diag_t* diag, bool split_only, const TauRetry* retry = nullptr) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!corpus || !params) return fail(ctx, YAMS_ERR_INVALID_ARG, "null corpus/params");
    if (diag) std::memset(diag, 0, sizeof(*diag));
    if (n_queries == 0) return YAMS_OK; // searchSimilarBatch on an empty batch (:1615-1617)
    if (!queries || !out_counts) return fail(ctx, YAMS_ERR_INVALID_ARG, "null queries/out_counts");
    if (params->metric != YAMS_SCAN_COSINE && params->metric != YAMS_SCAN_L2)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "unknown metric");
    if (corpus->dim == 0) { // query_embedding.empty() -> empty result (:4123-4126)
        YA_HIP(ctx, hipMemsetAsync(out_counts, 0, static_cast<size_t>(n_queries) * 4, ctx->stream));
        YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return YAMS_OK;
    }
    if (params->k == 0) { // k == 0 returns empty BEFORE the query is validated (:4123-4126)
        YA_HIP(ctx, hipMemsetAsync(out_counts, 0, static_cast<size_t>(n_queries) * 4, ctx->stream));
        YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return YAMS_OK;
    }
    if (!out_scores || !out_rows) return fail(ctx, YAMS_ERR_INVALID_ARG, "null outputs");
    if (params->k > YAMS_SCAN_MAX_K) return fail(ctx, YAMS_ERR_UNSUPPORTED, "k exceeds YAMS_SCAN_MAX_K");
    if (corpus->dim > YAMS_SCAN_MAX_DIM) return fail(ctx, YAMS_ERR_UNSUPPORTED, "dim exceeds YAMS_SCAN_MAX_DIM (8192)");
    if (corpus->n_rows >= (1ull << 32)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "shard must hold < 2^32 rows");
    if (corpus->n_rows > 0 && !corpus->rows) return fail(ctx, YAMS_ERR_INVALID_ARG, "null corpus rows");
    if ((corpus->tie_rank == nullptr) != (corpus->rank_row == nullptr))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "tie_rank and rank_row must be given together");
    if ((corpus->rows_bf16 == nullptr) != (corpus->rows_nsq == nullptr))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "rows_bf16 and rows_nsq must be given together");
    if ((corpus->rows_i8 == nullptr) != (corpus->rows_i8_meta == nullptr))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "rows_i8 and rows_i8_meta must be given together");
    if (corpus->row_mask && corpus->row_mask_count > corpus->n_rows)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "row_mask_count exceeds n_rows");
    if (corpus->stripe_rows && (corpus->n_stripes == 0 || corpus->stripe_index >= corpus->n_stripes))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "striped shard needs stripe_index < n_stripes");
    (void)hipSetDevice(ctx->device);
    // L2 (vec0) ties: the reference's statement is `ORDER BY distance` alone (:4473) and SQLite's sorter keeps rows of equal
    // distance in the order the vec0 table handed them over — rowid order, i.e. the order of this mirror — whatever their
    // chunk ids (pinned by the reference's own vec0SearchUnlocked compiled over SQLite: tests/test_scan_ref_l2_pin.py).  The
    // chunk_id ranking belongs to the cosine comparator (:4218-4223) only.
    yams_scan_corpus_t l2_view;
    if (params->metric == YAMS_SCAN_L2 && corpus->tie_rank) {
        l2_view = *corpus; l2_view.tie_rank = nullptr; l2_view.rank_row = nullptr;
        corpus = &l2_view;
    }
    if (!split_only && !retry && small_scan_applies(*corpus, n_queries, *params))
        return small_scan(ctx, corpus, queries, n_queries, params, out_scores, out_rows, out_counts, out_dist, out_ranks, diag);

    const uint32_t nq = n_queries, dim = corpus->dim, k = params->k;
    const int metric = static_cast<int>(params->metric);
    hipStream_t st = ctx->stream;
    if (metric == YAMS_SCAN_L2 && !out_dist && corpus->rows_i8 && k) {
        // (the int8 tier's second pass takes its threshold from the k-th exact DISTANCE found: kept even when the caller does not ask)
        YA_TRY(ws_get(ctx, "l2_dist_own", static_cast<size_t>(nq) * k * 4, (void**)&out_dist));
    }
    ScanIo io{corpus, queries, nq, *params, out_scores, out_rows, out_counts, out_dist, out_ranks};

    // ---- prep --------------------------------------------------------------------------------
    float* d_qprep; double* d_qnorm; float* d_qnorm_up; uint32_t* d_qflags; uint32_t* d_status;
    unsigned long long* d_stat;
    YA_TRY(ws_get(ctx, "qprep", static_cast<size_t>(nq) * dim * 4, (void**)&d_qprep));
    YA_TRY(ws_get(ctx, "qnorm", static_cast<size_t>(nq) * 8, (void**)&d_qnorm));
    YA_TRY(ws_get(ctx, "qnorm_up", static_cast<size_t>(nq) * 4, (void**)&d_qnorm_up));
    // The batch's small state words live in ONE block — [query flags nq | counters 16 | status nq | list counts nq]: everything
    // behind the flags starts at zero, and prep_queries (the first launch of every batch, which writes the flags) clears it —
    // three fills were three launches of their own; the host reads the whole block back with one copy where it read three.
    constexpr size_t kStatWords = 16;
    uint32_t* d_qstate; uint32_t* d_lcount;
    const size_t nq_al = (static_cast<size_t>(nq) + 3) & ~static_cast<size_t>(3); // (the 64-bit counters stay 16-byte aligned)
    const size_t qstate_words = 3 * nq_al + kStatWords;
    YA_TRY(ws_get(ctx, "qstate", qstate_words * 4, (void**)&d_qstate));
    d_qflags = d_qstate;
    d_stat = reinterpret_cast<unsigned long long*>(d_qstate + nq_al);
    d_status = d_qstate + nq_al + kStatWords;
    d_lcount = d_status + nq_al;
    YA_HIP(ctx, launch_prep_queries(st, queries, nq, dim, metric, d_qprep, d_qnorm, d_qnorm_up, d_qflags,
                                    d_qstate + nq_al, static_cast<uint32_t>(qstate_words - nq_al)));

    uint32_t* h_pin;
    YA_TRY(pinned_get(ctx, (nq_al * 4 + kStatWords) * 4 + 128, (void**)&h_pin));
    uint32_t* h_flags = h_pin;                                   // (the first three mirror the device block)
    uint32_t* h_status = h_pin + nq_al + kStatWords;
    uint32_t* h_lcount = h_status + nq_al;
    float* h_qnup = reinterpret_cast<float*>(h_lcount + nq_al);

    const bool aligned = (reinterpret_cast<uintptr_t>(corpus->rows) & 15u) == 0 && (dim & 3u) == 0;
    // rows that take part in the scan: all of them, or the set bits of the allow-mask
    const uint64_t n_eff = corpus->row_mask ? corpus->row_mask_count : corpus->n_rows;
    // a sparse allow-mask (document_hash / small candidate sets) is gathered and scored in fp64
    const bool sparse_mask = corpus->row_mask && n_eff < 4 * kMfmaMinRows;
    bool use_mfma = !(params->flags & YAMS_SCAN_FLAG_FORCE_EXACT) && aligned &&
                    corpus->n_rows >= kMfmaMinRows && !sparse_mask;
    // L2 on the int8 tier (scan_i8_kernel.hip, "L2 on the int8 tier") needs the shard's norm statistics: every
    // squared norm inside the filter's range and a norm spread the per-query line can follow.  They ride on the
    // sync the L2 path has anyway.
    bool l2_i8_ok = false;
    float* d_l2_nmin = nullptr; uint32_t* d_l2_stats = nullptr; uint32_t* d_l2_special = nullptr;
    uint32_t l2_n_special = 0;
    float l2_nsq_hi = 0.f;      // largest squared row norm of the shard (L2 on the int8 tier)
    const bool l2_i8_wanted = use_mfma && metric == YAMS_SCAN_L2 && corpus->rows_i8 && corpus->rows_i8_meta && corpus->rows_nsq &&
                              (dim & 63u) == 0 && dim >= 256 && corpus->n_rows >= 4096 &&
                              (reinterpret_cast<uintptr_t>(corpus->rows_i8) & 15u) == 0 &&
                              !(params->flags & (YAMS_SCAN_FLAG_NO_I8_FILTER | YAMS_SCAN_FLAG_F32_FILTER | YAMS_SCAN_FLAG_SPLIT_FILTER)) &&
                              !split_only;
    uint32_t* h_l2_stats = h_pin + 4 * nq_al + kStatWords + 8;
    if (use_mfma && metric == YAMS_SCAN_L2) {
        // The L2 filter works on raw magnitudes; queries far outside the fp32 comfort zone take
        // the fp64 path (needs the norms on the host: one small sync).
        YA_HIP(ctx, hipMemcpyAsync(h_qnup, d_qnorm_up, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
        if (l2_i8_wanted) {
            const uint64_t n_blocks = (corpus->n_rows + 63) / 64;
            YA_TRY(ws_get(ctx, "i8_l2_nmin", static_cast<size_t>(n_blocks) * 4, (void**)&d_l2_nmin));
            YA_TRY(ws_get(ctx, "i8_l2_stats", 32, (void**)&d_l2_stats));
            YA_TRY(ws_get(ctx, "i8_l2_special", static_cast<size_t>(i8_l2_max_special()) * 4, (void**)&d_l2_special));
            YA_HIP(ctx, hipMemsetAsync(d_l2_stats, 0, 32, st));
            YA_HIP(ctx, launch_i8_l2_norm_stats(st, corpus->rows_nsq, corpus->rows_i8_meta, corpus->n_rows, d_l2_nmin, d_l2_stats,
                                                d_l2_special));
            YA_HIP(ctx, hipMemcpyAsync(h_l2_stats, d_l2_stats, 32, hipMemcpyDeviceToHost, st));
        }
        YA_HIP(ctx, hipStreamSynchronize(st));
        for (uint32_t i = 0; i < nq; ++i)
            if (!(h_qnup[i] < 1e15f) || (h_qnup[i] != 0.f && h_qnup[i] < 1e-15f)) use_mfma = false;
        if (l2_i8_wanted) {
            float lo, hi;
            const uint32_t lo_bits = ~h_l2_stats[0], hi_bits = h_l2_stats[1];
            std::memcpy(&lo, &lo_bits, 4); std::memcpy(&hi, &hi_bits, 4);
            // a few rows without a usable norm ride along as unconditional candidates; the others within a factor of two
            l2_n_special = h_l2_stats[3];
            l2_i8_ok = l2_n_special <= i8_l2_max_special() && h_l2_stats[1] != 0 && hi <= 4.0f * lo;
            l2_nsq_hi = hi;
        }
    }

#ifdef YAMS_ACCEL_MEASURE
    if (std::getenv("YAMS_ACCEL_HONESTY_PRINT")) { const unsigned long long magic = 0x5eed; YA_HIP(ctx, hipMemcpyAsync(d_stat + 6, &magic, 8, hipMemcpyHostToDevice, st)); YA_HIP(ctx, hipStreamSynchronize(st)); }
#endif
    uint64_t filter_candidates = 0, rescored_nested = 0;
    uint32_t widened = 0, exact_fb = 0, escalated = 0, filter_tier = 0, retried = 0;
#ifdef YAMS_ACCEL_MEASURE
    // YAMS_ACCEL_TRACE_STAGES: host time at every stage boundary of one call (each boundary follows a stream synchronize)
    const bool trace_stages = std::getenv("YAMS_ACCEL_TRACE_STAGES") != nullptr;
    const auto trace_t0 = std::chrono::steady_clock::now();
    auto stage_mark = [&](const char* what, size_t n) {
        if (trace_stages) std::fprintf(stderr, "stage[%s%s] %-18s %8.3f ms  n=%zu\n", ctx->ws_ns.c_str(), split_only ? "split" : (retry ? "retry" : ""), what,
                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - trace_t0).count(), n);
    };
#else
    auto stage_mark = [](const char*, size_t) {};
#endif
    std::vector<uint32_t> flags_keep; // h_flags survives a nested (escalation) call through this copy
    if (!use_mfma) {
        const uint32_t* d_rows_sel = nullptr;
        uint64_t n_sel = 0;
        if (corpus->row_mask && corpus->n_rows > 0) {
            uint32_t* d_sel; unsigned long long* d_cnt;
            YA_TRY(ws_get(ctx, "mask_rows", static_cast<size_t>(corpus->n_rows) * 4, (void**)&d_sel));
            YA_TRY(ws_get(ctx, "mask_count", 64, (void**)&d_cnt));
            YA_HIP(ctx, launch_compact_mask(st, corpus->row_mask, corpus->n_rows, d_sel, d_cnt));
            unsigned long long* h_cnt = reinterpret_cast<unsigned long long*>(h_pin + 4 * nq_al + kStatWords);
            YA_HIP(ctx, hipMemcpyAsync(h_cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            n_sel = *h_cnt;
            d_rows_sel = d_sel;
        }
        YA_TRY(run_exact(ctx, io, d_qnorm, nullptr, d_status, d_stat, d_rows_sel, n_sel));
        YA_HIP(ctx, hipMemcpyAsync(h_flags, d_qflags, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        if (diag) diag->path = 1;
    } else {
        // bf16 matrix-core filter unless the caller asks for exact f32.  One RNE-bf16 pass is the
        // default: a third of the matrix work of the split filter for a looser bound (2^-7 |x||q|),
        // paid for by re-scoring ~3k instead of ~1.25k candidates per query.  Large k (where the
        // extra candidates would not fit the re-score stage: k > 661, L2 k > 319) and escalation runs
        // use the split filter.
        const bool bf16 = !(params->flags & YAMS_SCAN_FLAG_F32_FILTER) && (dim & 15u) == 0;
        // 2 = the library's own choice of kernel form; 3 = keep the 256-query tile for small batches
        int bf16_version = (params->flags & YAMS_SCAN_FLAG_WIDE_TILE) ? 3 : 2;
        int passes = 0;
        if (bf16) {
            // the single-pass tier needs 3k + 64 (L2: 6k + 128) candidates re-scored in stage 1
            const uint32_t need1 = (metric == YAMS_SCAN_L2) ? 6 * k + 128 : 3 * k + 64;
            passes = (split_only || (params->flags & YAMS_SCAN_FLAG_SPLIT_FILTER) || need1 > kRescoreMax) ? 3 : 1;
        }
        // The INT8 tier (cosine, dim % 64 == 0, dim >= 256, int8 shadow in the view): the tile loop on
        // v_mfma_i32_16x16x64_i8 (scan_i8_kernel.hip) — more than twice the sustained matrix rate,
        // half the shadow bytes, exact integer accumulation; its filter score is an upper bound of the
        // similarity built from the MEASURED quantisation residues, so the proof needs no extra error term.
        // Batches of <= 128 queries take it when the shard is large enough for the resident-query kernel form
        // (decided below, once the plan is known); on smaller shards they stay on the narrow bf16 form when a
        // bf16 shadow is there too.
        // L2 batches take it too when the shard's norms allow it (l2_i8_ok above).
        bool i8 = bf16 && passes == 1 && (metric == YAMS_SCAN_COSINE || l2_i8_ok) && (dim & 63u) == 0 && dim >= 256 && corpus->rows_i8 &&
                  corpus->rows_i8_meta && (reinterpret_cast<uintptr_t>(corpus->rows_i8) & 15u) == 0 &&
                  !(params->flags & YAMS_SCAN_FLAG_NO_I8_FILTER);
        if (i8 && (corpus->i8_flags & ~YAMS_SCAN_I8_ROTATED)) return fail(ctx, YAMS_ERR_INVALID_ARG, "unknown bits in yams_scan_corpus_t.i8_flags");
        if (i8 && (corpus->i8_flags & YAMS_SCAN_I8_ROTATED) && !i8_rotation_window(dim))
            return fail(ctx, YAMS_ERR_INVALID_ARG, "no rotated int8 layout exists for this dimension");
        // Tier hint: batches of more than 128 cosine queries on a corpus whose int8 batches keep escalating start on the bf16
        // tier (anisotropic rows, 12.5M x 768, 1024 queries: 59.9 ms per step on the int8 tier — all 1024 queries escalate —
        // 17.3 ms on the bf16 tier, no query widened; profiles/r06_non_uniform.json).  Learnt per context from the batches it
        // has served, probed again every 256th batch; results are identical on every tier.
        if (ctx->tier_hints.size() > 4096) ctx->tier_hints.clear();   // (keyed by shadow address: a long-lived context that has seen thousands of mirrors forgets)
        yams_accel_ctx::TierHint* hint = nullptr;
        if (i8 && metric == YAMS_SCAN_COSINE && !split_only && !retry && corpus->rows_bf16 && corpus->rows_nsq && nq > 128 &&
            !(params->flags & (YAMS_SCAN_FLAG_RESIDENT_QUERIES | YAMS_SCAN_FLAG_WIDE_TILE))) {
            hint = &ctx->tier_hints[corpus->rows];
            // (a mirror that GROWS keeps its address and its character: what was learnt stays; a row count that halved or
            // more than doubled is another corpus at this address)
            if (corpus->n_rows * 2 < hint->n_rows || corpus->n_rows > hint->n_rows * 2 || hint->n_rows == 0) { *hint = yams_accel_ctx::TierHint{}; }
            hint->n_rows = corpus->n_rows;
            if (hint->bf16_first && (++hint->served & 255u) != 0) i8 = false;
        }
#ifdef YAMS_ACCEL_MEASURE
        // Measurement build only (libyams_mi355x_accel_measure.so, scripts/): kernel-form and
        // ablation selection from the environment.  The product library never reads it.
        if (const char* kv = std::getenv("YAMS_ACCEL_BF16_KERNEL")) bf16_version = std::atoi(kv);
        if (const char* pv = std::getenv("YAMS_ACCEL_BF16_PASSES"))
            if (bf16 && !split_only) passes = std::atoi(pv) == 3 ? 3 : 1;
        if (passes != 1 || (bf16_version != 2 && bf16_version != 3 && bf16_version != 30 && bf16_version != 31 && bf16_version != 32 && bf16_version != 37 && bf16_version != 38 && !(bf16_version >= 40 && bf16_version <= 99))) i8 = false;
#endif
        // Depth hint (round 6): on rows with Gaussian components — what embedding models emit, and what the rotated layout makes
        // of any corpus — the int8 bound is 2.4x as wide as on the bench's uniform rows and the proof of a top-100 over 12.5M
        // rows needs ~850 candidates re-scored, not the plan's 384: every query failed stage 1 and was widened, a third found
        // its list too short and went through a second sweep (11.4 ms per batch instead of 7.6).  The context remembers per
        // corpus what its batches needed and plans the next ones for it (probed without the hint every 256th batch).
        // The single-pass bf16 tier (dims that are not a multiple of 64, views without an int8 shadow) learns the same way, under
        // cosine: its lists are cut by the same sampled threshold.
        yams_accel_ctx::TierHint* dhint = nullptr;
        int depth = 0, hint_tier = 0;
        auto depth_for = [&](int tier) {          // what the context has learnt for this tier of this corpus (0 every 256th batch: a probe)
            return dhint && dhint->depth[tier] && (++dhint->served_deep[tier] & 255u) != 0 ? static_cast<int>(dhint->depth[tier]) : 0;
        };
        if ((i8 || (bf16 && passes == 1 && metric == YAMS_SCAN_COSINE)) && !split_only && !retry) {   // (int8 tier: both metrics — its L2 batches plan the same lists)
            dhint = hint ? hint : &ctx->tier_hints[corpus->rows];
            if (corpus->n_rows * 2 < dhint->n_rows || corpus->n_rows > dhint->n_rows * 2 || dhint->n_rows == 0) { *dhint = yams_accel_ctx::TierHint{}; }
            dhint->n_rows = corpus->n_rows;
            hint_tier = i8 ? 0 : 1;
            depth = depth_for(hint_tier);
        }
        ScanPlan plan = make_plan(corpus->n_rows, dim, nq, k, bf16, passes, metric == YAMS_SCAN_L2, depth);
        if (retry) plan.kprime = kRescoreMax;   // (the retry's lists are what the proof needs: all of a list is re-scored)
        ScanLaunch L;
        L.plan = plan; L.rows = corpus->rows; L.row_mask = corpus->row_mask;
        if (corpus->rows_bf16 && corpus->rows_nsq && (reinterpret_cast<uintptr_t>(corpus->rows_bf16) & 15u) == 0) {
            L.rows_bf16 = corpus->rows_bf16; L.rows_nsq = corpus->rows_nsq; // used by the single-pass kernel
        }
        L.i8_form = (params->flags & YAMS_SCAN_FLAG_WIDE_TILE) ? 1 : ((params->flags & YAMS_SCAN_FLAG_RESIDENT_QUERIES) ? 2 : 0);
#ifdef YAMS_ACCEL_MEASURE
        if (bf16_version == 40) L.i8_form = 1; // A/B runs: half tiles where the library would pick the resident-query form
#endif
        // Multi-GPU modes (the sharded handle's exchange fence, or a caller that holds the gate for its collective): the
        // exchange of the previous batch may still be on this device when this batch's SAMPLE pass starts — only the filter
        // sweep is fenced behind it.  The resident-query sample form is a grid of one 160 KiB workgroup per CU: a collective
        // kernel would have to wait for it.  The half-tile form (two small workgroups per CU) leaves it room.
        L.i8_sample_small_grid = static_cast<bool>(ctx->before_sweep) || ctx->sweep_hold;
        if (i8 && nq <= 128 && corpus->rows_bf16 && !retry && !i8_takes_resident_form(L)) i8 = false; // small batch on a small shard: narrow bf16 (a second pass stays: its threshold is the int8 tier's)
        if (dhint && !i8 && hint_tier == 0) {   // the batch left the int8 tier after it was planned: the other tier's lesson applies (none under L2)
            hint_tier = 1;
            if (metric != YAMS_SCAN_COSINE) dhint = nullptr;
            const int d2 = depth_for(1);
            if (d2 != depth) { depth = d2; plan = make_plan(corpus->n_rows, dim, nq, k, bf16, passes, metric == YAMS_SCAN_L2, depth); L.plan = plan; }
        }
        if (i8) { L.rows_i8 = corpus->rows_i8; L.rows_i8_meta = corpus->rows_i8_meta; }
        if (i8 && metric == YAMS_SCAN_L2) {
            L.i8_l2 = true; L.rows_nsq = corpus->rows_nsq; L.l2_eps = i8_l2_eps(dim);
            plan = make_plan(corpus->n_rows, dim, nq, k, bf16, passes, false, depth); // (tile geometry unchanged: the form decision above stands)
            if (retry) plan.kprime = kRescoreMax;
            L.plan = plan;
        }
        L.qprep = d_qprep; L.qnorm_up = d_qnorm_up;
        // relative error of the filter's dot product, in units of |x||q|:
        //   exact f32 : fp32 FMA chain over dim terms
        //   split bf16: 3*dim fp32 accumulations (x2 safety for the MFMA adder tree) + the split residue:
        //               corpus head truncated (tail error 2^-16), query split RNE (2^-18), lo*lo dropped (2^-16)
        const double u24 = 5.9604644775390625e-8;
        //   RNE bf16  : both operands rounded to 8 significant bits (u = 2^-8 each): |x^q^ - xq| <=
        //               (2u + u^2)|x||q| summed with Cauchy-Schwarz, + dim fp32 accumulations (x2)
        const double dot_rel = passes == 3   ? (6.0 * dim + 64.0) * u24 + 3.0 / 65536.0
                               : passes == 1 ? (2.0 * dim + 64.0) * u24 + 2.0 / 256.0 + 2.0 / 65536.0
                                             : (dim + 8.0) * u24;
        // with the (pre-normalised) shadow the row-norm rounding sits inside the dot product
        const bool use_shadow = passes == 1 && L.rows_bf16 && bf16_slab_k(passes, dim) == 32;
        const double norm_rel = (dim + 32.0) * u24;
        L.err_coef = static_cast<float>((dot_rel + (use_shadow ? norm_rel : 0.0)) * 1.01);
        if (i8) {
            int8_t* d_qi8; float* d_qmeta;
            const uint32_t q_pad = plan.n_qtiles * plan.tile_queries;
            YA_TRY(ws_get(ctx, "q_i8", static_cast<size_t>(q_pad) * dim, (void**)&d_qi8));
            YA_TRY(ws_get(ctx, "q_meta", static_cast<size_t>(q_pad) * 16, (void**)&d_qmeta));
            float* d_qthr;
            YA_TRY(ws_get(ctx, "q_thr", static_cast<size_t>(q_pad) * 8, (void**)&d_qthr));
            L.q_i8 = d_qi8; L.q_meta = d_qmeta; L.q_thr = d_qthr; L.q_pad = q_pad; L.sample_layout = 1;
            if (metric == YAMS_SCAN_COSINE || metric == YAMS_SCAN_L2) { // proof-aware threshold (tau_select_kernel)
                // (under L2 the sample values are g = n (x~ . q) - n^2 / 2: the bound's width in those units is n E, taken at the
                // shard's largest norm — L.i8_l2 is decided below, the fields are harmless on the other tiers)
                L.tau_e_scale = metric == YAMS_SCAN_L2 ? std::sqrt(std::max(l2_nsq_hi, 0.f)) * 1.0001f : 1.0f;
                L.tau_rows_meta = corpus->rows_i8_meta; L.tau_n_blocks = (corpus->n_rows + 63) / 64;
                L.tau_rank2 = (k + plan.sample_stride - 1) / plan.sample_stride + 4;     // P(fewer than k rows reach that sample value) < 1 %
                L.tau_max_groups = kRescoreMax * 3u / 2 / plan.sample_stride;               // what the crowd is estimated at must fit the list (cap: 4096 rows or more)
            }
            if (L.i8_l2) { // the per-batch tables of the L2 threshold: built after the sample pass (below)
                const uint64_t n_blocks = (corpus->n_rows + 63) / 64;
                float* d_l2meta; uint8_t* d_rbias; uint32_t* d_qbias;
                YA_TRY(ws_get(ctx, "i8_l2_meta", static_cast<size_t>(n_blocks) * 8, (void**)&d_l2meta));
                YA_TRY(ws_get(ctx, "i8_l2_rbias", static_cast<size_t>(n_blocks) * 64, (void**)&d_rbias));
                YA_TRY(ws_get(ctx, "i8_l2_qbias", static_cast<size_t>(q_pad) * 4, (void**)&d_qbias));
                L.i8_l2_meta = d_l2meta; L.i8_row_bias = d_rbias; L.i8_q_bias = d_qbias;
            }
        } else if (bf16) {
            uint16_t* d_qhi; uint16_t* d_qlo;
            const uint32_t q_pad = plan.n_qtiles * plan.tile_queries;
            YA_TRY(ws_get(ctx, "q_hi", static_cast<size_t>(q_pad) * dim * 2, (void**)&d_qhi));
            YA_TRY(ws_get(ctx, "q_lo", static_cast<size_t>(q_pad) * dim * 2, (void**)&d_qlo)); // unused by the 1-pass kernel
            YA_HIP(ctx, launch_prep_split(st, d_qprep, nq, q_pad, dim, bf16_slab_k(passes, dim), d_qhi, d_qlo));
            L.q_hi = d_qhi; L.q_lo = d_qlo; L.q_pad = q_pad;
        }
        if (retry && !i8 && !(bf16 && passes == 1 && metric == YAMS_SCAN_COSINE)) {    // (the forced threshold is a value of the caller's tier's score)
            for (uint32_t i = 0; i < nq; ++i) retry->unproven->push_back(i);
            return YAMS_OK;
        }
        float* d_tau; uint64_t* d_list; uint32_t* d_work32; uint64_t* d_work64;
        if (i8) L.dense = nullptr; // (the int8 sample pass keeps group maxima only, launch_i8_collect_sample)
        else YA_TRY(ws_get(ctx, "dense", static_cast<size_t>(nq) * plan.sample_rows * 4, (void**)&L.dense));
        YA_TRY(ws_get(ctx, "gmax", static_cast<size_t>(nq) * plan.n_groups * 4, (void**)&L.gmax));
        YA_TRY(ws_get(ctx, "tau", static_cast<size_t>(nq) * 4, (void**)&d_tau));
        YA_TRY(ws_get(ctx, "list", static_cast<size_t>(nq) * plan.list_cap * 8, (void**)&d_list));
        const uint32_t gchunks = (plan.n_groups + kSelectCap - 1) / kSelectCap;
        YA_TRY(ws_get(ctx, "work32", static_cast<size_t>(2) * nq * std::max(1u, gchunks) * plan.tau_rank * 4, (void**)&d_work32));
        const uint32_t lchunks = (plan.list_cap + kSelectCap - 1) / kSelectCap;
        const uint32_t keep_max = kRescoreMax + 1;
        YA_TRY(ws_get(ctx, "work64", static_cast<size_t>(2) * nq * lchunks * keep_max * 8, (void**)&d_work64));
        L.tau = d_tau; L.tau_out = d_tau; L.list_count = d_lcount; L.list = d_list;
        uint32_t* d_qover = nullptr;
        if (i8) { // the int8 filter writes its survivors to a log (scan_i8_kernel.hip), one region per (workgroup, wave)
            L.i8_q_form = i8_takes_q_form(L, bf16_version);
            if (corpus->row_mask && corpus->row_mask_count) L.i8_mask_inflation = static_cast<double>(corpus->n_rows) / static_cast<double>(corpus->row_mask_count);
            const uint64_t regions = i8_log_regions(L);
            L.log_cap = i8_log_capacity(L);
            YA_TRY(ws_get(ctx, "i8_log_key", static_cast<size_t>(regions) * L.log_cap * i8_log_entry_bytes(L), (void**)&L.log_key));
            YA_TRY(ws_get(ctx, "i8_log_q", static_cast<size_t>(regions) * L.log_cap * 4, (void**)&L.log_q));
            // the three small tables the filter launch needs zeroed — region counts, per-query overflow marks, pacing
            // counters — share ONE buffer and one fill (each fill is a 5 us launch of its own in front of the sample pass)
            const uint64_t sync_words = i8_sync_words(L);
            const size_t z_cnt = (static_cast<size_t>(regions) * 4 + 255) & ~size_t(255), z_over = (static_cast<size_t>(nq) * 4 + 255) & ~size_t(255);
            unsigned char* zeroed;
            YA_TRY(ws_get(ctx, "i8_zeroed", z_cnt + z_over + static_cast<size_t>(sync_words) * 4, (void**)&zeroed));
            // (cleared by the query preparation of the int8 tier: one launch where there were a fill and a launch)
            YA_HIP(ctx, launch_prep_i8(st, d_qprep, nq, L.q_pad, dim, const_cast<int8_t*>(L.q_i8), const_cast<float*>(L.q_meta), L.i8_l2,
                                       reinterpret_cast<uint32_t*>(zeroed), (z_cnt + z_over) / 4 + sync_words,
                                       (corpus->i8_flags & YAMS_SCAN_I8_ROTATED) != 0));
            L.log_cnt = reinterpret_cast<uint32_t*>(zeroed);
            d_qover = reinterpret_cast<uint32_t*>(zeroed + z_cnt);
            L.q_over = d_qover;
            if (sync_words) L.i8_sync = reinterpret_cast<uint32_t*>(zeroed + z_cnt + z_over);
        }

        bool emu_no_sample = false;
#ifdef YAMS_ACCEL_MEASURE
        // EMULATION (kill criterion): what would the two-lane step cost if the sample pass, the tau selection
        // and the collect kernel were not launches of their own (folded into the head of the sweep)?  From a context's third
        // batch on they are skipped and the previous batch's tau / thresholds / group maxima are used — exact when the same
        // query batch comes again (bench.py --query-batches 1), timing only otherwise.  Never in the product build.
        if (std::getenv("YAMS_ACCEL_EMU_NO_SAMPLE") && i8 && !L.i8_l2) emu_no_sample = ++ctx->emu_calls > 2;
#endif
        if (!emu_no_sample) {
        { TimedRegion tr(ctx, "scan_sample");
          if (i8) YA_HIP(ctx, launch_scan_i8(st, L, 0, bf16_version));
          else if (bf16) YA_HIP(ctx, launch_scan_bf16(st, L, metric, 0, passes, bf16_version)); else YA_HIP(ctx, launch_scan_sample(st, L, metric));
          tr.end(); }
#ifdef YAMS_ACCEL_MEASURE
        if (L.dense && ctx->ws_ns.empty()) if (const char* dump = std::getenv("YAMS_ACCEL_DUMP_DENSE")) {     // the sample pass's scores, [sample_row / 4][query][4]
            const size_t nf = static_cast<size_t>(nq) * plan.sample_rows;
            std::vector<float> h(nf);
            YA_HIP(ctx, hipMemcpyAsync(h.data(), L.dense, nf * 4, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            if (FILE* f = std::fopen(dump, "wb")) {
                const uint64_t hdr[4] = {nq, plan.sample_rows, plan.tile_rows, plan.sample_stride};
                std::fwrite(hdr, 8, 4, f); std::fwrite(h.data(), 4, nf, f); std::fclose(f);
            }
        }
#endif
        if (retry) YA_HIP(ctx, hipMemcpyAsync(d_tau, retry->forced_tau, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToDevice, st));
        else YA_HIP(ctx, launch_select_tau(st, L, d_work32));
        }
        if (emu_no_sample) {
            // (the lists lose the sample rows' candidates: results of the emulation are not checked)
        } else if (i8 && L.i8_l2) {
            YA_HIP(ctx, launch_i8_l2_thresholds(st, d_tau, L.q_meta, nq, L.q_pad, dim, d_l2_stats, const_cast<float*>(L.q_thr),
                                                const_cast<uint32_t*>(L.i8_q_bias)));
            YA_HIP(ctx, launch_i8_l2_rows(st, corpus->rows_nsq, corpus->rows_i8_meta, d_l2_nmin, corpus->n_rows, d_l2_stats,
                                          const_cast<float*>(L.i8_l2_meta), const_cast<uint8_t*>(L.i8_row_bias)));
        } else if (i8) YA_HIP(ctx, launch_i8_thresholds(st, d_tau, L.q_meta, nq, L.q_pad, const_cast<float*>(L.q_thr)));
        if (emu_no_sample) {} else
        if (i8) YA_HIP(ctx, launch_i8_collect_sample(st, L)); else YA_HIP(ctx, launch_collect_sample(st, L));
        if (i8 && L.i8_l2) YA_HIP(ctx, launch_i8_l2_add_special(st, L, d_l2_special, l2_n_special));
        { GatedSweep gs(ctx, st); // sweeps of contexts that share a gate run one after the other
          TimedRegion tr(ctx, "scan_filter");
          if (i8) YA_HIP(ctx, launch_scan_i8(st, L, 1, bf16_version));
          else if (bf16) YA_HIP(ctx, launch_scan_bf16(st, L, metric, 1, passes, bf16_version)); else YA_HIP(ctx, launch_scan_filter(st, L, metric));
          tr.end();
          gs.leave(); }
        if (i8) YA_HIP(ctx, launch_i8_log_gather(st, L));
#ifdef YAMS_ACCEL_MEASURE
        if (i8 && L.i8_sync) if (const char* dump = std::getenv("YAMS_ACCEL_DUMP_SYNC")) { // per-wave begin / end ticks of the resident-query kernel
            std::vector<uint32_t> h(i8_sync_words(L));
            YA_HIP(ctx, hipMemcpyAsync(h.data(), L.i8_sync, h.size() * 4, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(h.data(), 4, h.size(), f); std::fclose(f); }
        }
#endif

#ifdef YAMS_ACCEL_MEASURE
        if (bf16_version != 2 && bf16_version != 3 && bf16_version != 4 && bf16_version != 20 && bf16_version != 30 && bf16_version != 40 && bf16_version != 50 && bf16_version != 70 && bf16_version != 80 && !(bf16_version >= 81 && bf16_version <= 99)) { // ablated kernels produce no candidates: stop here
            YA_HIP(ctx, hipStreamSynchronize(st));
            YA_HIP(ctx, hipMemsetAsync(out_counts, 0, static_cast<size_t>(nq) * 4, st));
            return YAMS_OK;
        }
#endif
        // stage 1: re-score the best kprime filter survivors of every query
        // cosine: |s32 - cos| <= dot_rel + norm (dim/2 u) + rsqrt/product/unit-query rounding
        // (the int8 tier's filter score already is an upper bound of the similarity)
        const double err_bound = (metric == YAMS_SCAN_COSINE && !i8)
                                     ? dot_rel + (dim + 24.0) * u24 + (use_shadow ? norm_rel : 0.0) : 0.0;
        filter_tier = i8 ? 1u : (!bf16 ? 4u : (passes == 3 ? 3u : 2u));
        auto rescore_stage = [&](uint32_t n_slots, const uint32_t* d_qmap, uint32_t n_cand) -> yams_status_t {
            const uint64_t* res; uint64_t res_stride;
            YA_HIP(ctx, launch_select_lists(st, d_list, d_lcount, plan.list_cap, n_slots, d_qmap,
                                            n_cand + 1, d_work64, &res, &res_stride));
            RescoreLaunch R{};
            R.rows = corpus->rows; R.n_rows = corpus->n_rows; R.dim = dim; R.queries = queries;
            R.qnorm = d_qnorm; R.tie_rank = corpus->tie_rank; R.rank_row = nullptr;
            R.row_base = corpus->row_base; R.stripe_rows = corpus->stripe_rows;
            R.n_stripes = corpus->n_stripes; R.stripe_index = corpus->stripe_index; R.cand = res; R.cand_stride = res_stride;
            R.n_cand = n_cand; R.tau = d_tau; R.list_count = d_lcount; R.list_cap = plan.list_cap;
            R.all_rows_listed = 0; R.qmap = d_qmap; R.n_slots = n_slots; R.k = k;
            R.threshold = params->similarity_threshold; R.flags = params->flags & ~(kRescoreFlagPqRerank | kRescoreFlagNoEarlyClose);
            R.err_bound = err_bound; R.out_scores = out_scores; R.out_rows = out_rows;
            R.out_counts = out_counts; R.out_dist = out_dist; R.out_ranks = out_ranks;
            R.out_status = d_status; R.stat_rescored = d_stat; R.q_over = d_qover;
            YA_HIP(ctx, launch_rescore(st, metric, R));
            return YAMS_OK;
        };
        uint32_t kprime1 = plan.kprime;
#ifdef YAMS_ACCEL_MEASURE
        if (const char* kv = std::getenv("YAMS_ACCEL_EMU_KPRIME")) kprime1 = std::min<uint32_t>(plan.kprime, std::max<uint32_t>(k, static_cast<uint32_t>(std::atoi(kv)))); // (a tighter bound would re-score this many)
#endif
        YA_TRY(rescore_stage(nq, nullptr, kprime1));
        YA_HIP(ctx, hipMemcpyAsync(h_pin, d_qstate, qstate_words * 4, hipMemcpyDeviceToHost, st)); // flags, status, list counts
        YA_HIP(ctx, hipStreamSynchronize(st));
#ifdef YAMS_ACCEL_MEASURE
        if (const char* dump = std::getenv("YAMS_ACCEL_DUMP_LCOUNT")) // per-query candidate counts of the filter pass
            if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(h_lcount, 4, nq, f); std::fclose(f); }
#endif
        std::vector<uint32_t> failed, overflowed;
        stage_mark("stage1 done", nq);
        for (uint32_t i = 0; i < nq; ++i) {
            filter_candidates += std::min<uint32_t>(h_lcount[i], plan.list_cap);
            if (h_status[i] != 0 && h_flags[i] == 0) {
                // a list that overflowed is incomplete: widening cannot help, go exhaustive
                if (h_lcount[i] > plan.list_cap) overflowed.push_back(i); else failed.push_back(i);
            }
        }
        if (retry) {    // the caller decides what happens to what is still unproven (its lists overflowed, or hold more than can be re-scored)
            retry->unproven->insert(retry->unproven->end(), failed.begin(), failed.end());
            retry->unproven->insert(retry->unproven->end(), overflowed.begin(), overflowed.end());
            if (diag) { diag->filter_candidates = filter_candidates; diag->filter_tier = filter_tier; }
            return YAMS_OK;
        }
        if (!failed.empty() && plan.kprime < kRescoreMax) {
            // stage 2: widen to everything the list holds (up to kRescoreMax candidates)
            widened = static_cast<uint32_t>(failed.size());
            uint32_t* d_qmap;
            YA_TRY(ws_get(ctx, "widen_qmap", failed.size() * 4, (void**)&d_qmap));
            YA_HIP(ctx, hipMemcpyAsync(d_qmap, failed.data(), failed.size() * 4, hipMemcpyHostToDevice, st));
            YA_TRY(rescore_stage(static_cast<uint32_t>(failed.size()), d_qmap, kRescoreMax));
            YA_HIP(ctx, hipMemcpyAsync(h_status, d_status, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            std::vector<uint32_t> still;
            for (uint32_t q : failed) if (h_status[q] != 0) still.push_back(q);
            failed.swap(still);
            stage_mark("widen done, left", failed.size());
        }
        const bool bf16_single = !i8 && bf16 && passes == 1 && metric == YAMS_SCAN_COSINE;
        if (!failed.empty() && ((i8 && (!L.i8_l2 || out_dist)) || bf16_single) && !split_only && k <= 1024) {
            // stage 2a (round 6): the int8 tier once more, with the threshold the proof asks for.  For every unproven query
            // stage 1 / 2 left the k best EXACT scores it found: no row whose upper bound lies below the k-th of them can be in
            // the answer, so tau' = that score (one ulp down: the proof is a strict comparison) lists exactly what matters.
            // The sample's group maxima say beforehand how many rows that will be: queries whose list would not fit go
            // straight to the escalation below.
            const size_t nf = failed.size();
            float* d_rtau; uint32_t* d_rest; uint32_t* d_fmap;
            YA_TRY(ws_get(ctx, "retry_tau", nf * 4, (void**)&d_rtau));
            YA_TRY(ws_get(ctx, "retry_est", nf * 4, (void**)&d_rest));
            YA_TRY(ws_get(ctx, "retry_fmap", nf * 4, (void**)&d_fmap));
            YA_HIP(ctx, hipMemcpyAsync(d_fmap, failed.data(), nf * 4, hipMemcpyHostToDevice, st));
            if (L.i8_l2) {
                const bool f32acc = (params->flags & YAMS_SCAN_FLAG_L2_ACC_MASK) != 0;
                const double margin = 1e-6 + (f32acc ? 2.0 * (static_cast<double>(dim) + 8.0) * 5.9604644775390625e-8 * 1.01 : 0.0);
                YA_HIP(ctx, launch_retry_tau_l2(st, out_dist, out_counts, k, d_fmap, static_cast<uint32_t>(nf), d_qnorm, margin, L.gmax, plan.n_groups, d_rtau, d_rest));
            } else
            // (bf16 tier: its score is within err_bound of the similarity either way: a row that can still enter scores >= s_k - err_bound)
            YA_HIP(ctx, launch_retry_tau(st, out_scores, out_counts, k, d_fmap, static_cast<uint32_t>(nf), L.gmax, plan.n_groups, d_rtau, d_rest,
                                         bf16_single ? static_cast<float>(err_bound * 1.000001 + 1e-7) : 0.f));
            std::vector<uint32_t> est(nf);
            YA_HIP(ctx, hipMemcpyAsync(est.data(), d_rest, nf * 4, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            std::vector<uint32_t> sub, rest; // indices into `failed`
            for (uint32_t i = 0; i < nf; ++i) {
                // est = sample groups that reach tau' (0xffffffff: stage 1 found fewer than k rows): each stands for `stride` rows
                const uint64_t rows_est = est[i] == 0xffffffffu ? ~0ull : static_cast<uint64_t>(est[i]) * plan.sample_stride;
                (rows_est <= kRescoreMax * 5ull / 4 ? sub : rest).push_back(i);  // (the estimate's spread is ~ 1 / sqrt(groups): a list a quarter over still has an even chance to fit)
            }
            stage_mark("retry: fits", sub.size());
            if (!sub.empty()) {
                const size_t ns = sub.size(), kk = k;
                std::vector<uint32_t> sub_q(ns);
                for (size_t i = 0; i < ns; ++i) sub_q[i] = failed[sub[i]];
                float* s_q; float* s_scores; int64_t* s_rows; uint32_t* s_counts; float* s_dist = nullptr; uint32_t* s_ranks = nullptr;
                uint32_t* d_submap; float* s_tau;
                YA_TRY(ws_get(ctx, "sub_queries", ns * dim * 4, (void**)&s_q));
                YA_TRY(ws_get(ctx, "sub_scores", ns * kk * 4, (void**)&s_scores));
                YA_TRY(ws_get(ctx, "sub_rows", ns * kk * 8, (void**)&s_rows));
                YA_TRY(ws_get(ctx, "sub_counts", ns * 4, (void**)&s_counts));
                if (out_dist) YA_TRY(ws_get(ctx, "sub_dist", ns * kk * 4, (void**)&s_dist));
                if (out_ranks) YA_TRY(ws_get(ctx, "sub_ranks", ns * kk * 4, (void**)&s_ranks));
                YA_TRY(ws_get(ctx, "sub_qmap", ns * 4, (void**)&d_submap));
                YA_TRY(ws_get(ctx, "sub_tau", ns * 4, (void**)&s_tau));
                YA_HIP(ctx, hipMemcpyAsync(d_submap, sub_q.data(), ns * 4, hipMemcpyHostToDevice, st));
                YA_HIP(ctx, launch_gather_queries(st, queries, d_submap, static_cast<uint32_t>(ns), dim, s_q));
                std::vector<uint32_t> sub_slots(sub.begin(), sub.end());
                uint32_t* d_subslots;
                YA_TRY(ws_get(ctx, "sub_slots", ns * 4, (void**)&d_subslots));
                YA_HIP(ctx, hipMemcpyAsync(d_subslots, sub_slots.data(), ns * 4, hipMemcpyHostToDevice, st));
                YA_HIP(ctx, launch_gather_queries(st, d_rtau, d_subslots, static_cast<uint32_t>(ns), 1, s_tau));
                YA_HIP(ctx, hipStreamSynchronize(st)); // (the host vectors above are pageable)
                flags_keep.assign(h_flags, h_flags + nq);
                std::vector<uint32_t> unproven;
                TauRetry rt{s_tau, &unproven};
                yams_scan_diag_t subd{};
                {
                    const std::string outer_ns = ctx->ws_ns;
                    ctx->ws_ns = outer_ns + "retry/";
                    const yams_status_t r_st = scan_impl(ctx, corpus, s_q, static_cast<uint32_t>(ns), params, s_scores, s_rows, s_counts, s_dist, s_ranks,
                                                         &subd, false, &rt);
                    ctx->ws_ns = outer_ns;
                    YA_TRY(r_st);
                }
                retried = static_cast<uint32_t>(ns);
                filter_candidates += subd.filter_candidates;
                // the proven ones go back to their places; the others join the queries the escalation takes
                std::vector<uint8_t> bad(ns, 0);
                for (uint32_t u : unproven) bad[u] = 1;
                std::vector<uint32_t> good_src, good_dst;
                for (uint32_t i = 0; i < ns; ++i) {
                    if (bad[i]) rest.push_back(sub[i]);
                    else { good_src.push_back(i); good_dst.push_back(sub_q[i]); }
                }
                if (!good_src.empty()) {
                    uint32_t* d_src; uint32_t* d_dst;
                    YA_TRY(ws_get(ctx, "retry_src", good_src.size() * 4, (void**)&d_src));
                    YA_TRY(ws_get(ctx, "retry_dst", good_dst.size() * 4, (void**)&d_dst));
                    YA_HIP(ctx, hipMemcpyAsync(d_src, good_src.data(), good_src.size() * 4, hipMemcpyHostToDevice, st));
                    YA_HIP(ctx, hipMemcpyAsync(d_dst, good_dst.data(), good_dst.size() * 4, hipMemcpyHostToDevice, st));
                    YA_HIP(ctx, launch_scatter_results_from(st, d_src, d_dst, static_cast<uint32_t>(good_src.size()), k, s_scores, s_rows, s_counts, s_dist,
                                                            s_ranks, out_scores, out_rows, out_counts, out_dist, out_ranks));
                    YA_HIP(ctx, hipStreamSynchronize(st));
                }
            }
            std::vector<uint32_t> still;
            std::sort(rest.begin(), rest.end());
            for (uint32_t i : rest) still.push_back(failed[i]);
            failed.swap(still);
            stage_mark("retry done, left", failed.size());
        }
        if (!failed.empty() && passes == 1) {
            // stage 2b: precision escalation.  The unproven queries become their own small batch
            // under the split (3-pass) filter, whose bound is ~170x tighter; that run widens and
            // falls back to the exhaustive scan on its own.  Results are scattered back.
            escalated = static_cast<uint32_t>(failed.size());
            if (flags_keep.empty()) flags_keep.assign(h_flags, h_flags + nq); // (a retry run above has already taken the pinned words over)
            unsigned long long h_stat0 = 0;
            YA_HIP(ctx, hipMemcpyAsync(&h_stat0, d_stat, 8, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            rescored_nested = h_stat0;
            const size_t ns = failed.size(), kk = k;
            float* s_q; float* s_scores; int64_t* s_rows; uint32_t* s_counts; float* s_dist = nullptr;
            uint32_t* s_ranks = nullptr; uint32_t* d_submap;
            YA_TRY(ws_get(ctx, "sub_queries", ns * dim * 4, (void**)&s_q));
            YA_TRY(ws_get(ctx, "sub_scores", ns * kk * 4, (void**)&s_scores));
            YA_TRY(ws_get(ctx, "sub_rows", ns * kk * 8, (void**)&s_rows));
            YA_TRY(ws_get(ctx, "sub_counts", ns * 4, (void**)&s_counts));
            if (out_dist) YA_TRY(ws_get(ctx, "sub_dist", ns * kk * 4, (void**)&s_dist));
            if (out_ranks) YA_TRY(ws_get(ctx, "sub_ranks", ns * kk * 4, (void**)&s_ranks));
            YA_TRY(ws_get(ctx, "sub_qmap", ns * 4, (void**)&d_submap));
            YA_HIP(ctx, hipMemcpyAsync(d_submap, failed.data(), ns * 4, hipMemcpyHostToDevice, st));
            YA_HIP(ctx, launch_gather_queries(st, queries, d_submap, escalated, dim, s_q));
            YA_HIP(ctx, hipStreamSynchronize(st)); // `failed` is pageable
            yams_scan_diag_t sub{};
            {   // the nested run gets its own workspace namespace: this call's qnorm / status /
                // candidate lists are still needed by the exhaustive pass below
                const std::string outer_ns = ctx->ws_ns;
                ctx->ws_ns = outer_ns + "esc/";
                const yams_status_t ns_st = scan_impl(ctx, corpus, s_q, escalated, params, s_scores, s_rows,
                                                      s_counts, s_dist, s_ranks, &sub, true);
                ctx->ws_ns = outer_ns;
                YA_TRY(ns_st);
            }
            YA_HIP(ctx, launch_scatter_results(st, d_submap, escalated, k, s_scores, s_rows, s_counts,
                                               s_dist, s_ranks, out_scores, out_rows, out_counts,
                                               out_dist, out_ranks));
            YA_HIP(ctx, hipStreamSynchronize(st));
            filter_candidates += sub.filter_candidates;
            rescored_nested += sub.rescored_rows;
            widened += sub.widened_queries;
            exact_fb += sub.exact_fallback_queries;
            failed.clear();
            // d_stat was read into rescored_nested above; the nested call counted in its own buffer
            YA_HIP(ctx, hipMemsetAsync(d_stat, 0, 64, st));
            stage_mark("escalation done", escalated);
        }
        if (hint && i8) hint->bf16_first = static_cast<uint64_t>(escalated) * 2 > nq;    // (an int8 batch — first or probe — decides for the next 255)
        if (dhint && (i8 ? hint_tier == 0 : hint_tier == 1)) {
            // (lists cut short show as second passes, or as escalations where there is no second pass)
            const uint64_t cut_short = static_cast<uint64_t>(retried) + escalated;
            if (depth == 0) dhint->depth[hint_tier] = cut_short * 8 > nq ? 2 : (static_cast<uint64_t>(widened) * 4 > nq ? 1 : 0); // (a plain batch decides)
            else if (depth == 1 && cut_short * 8 > nq) dhint->depth[hint_tier] = 2;
        }
        failed.insert(failed.end(), overflowed.begin(), overflowed.end());
        if (!failed.empty()) {
            // stage 3: exhaustive fp64 for the queries that could not be proven complete
            exact_fb += static_cast<uint32_t>(failed.size());
            YA_TRY(run_exact(ctx, io, d_qnorm, &failed, d_status, d_stat));
            YA_HIP(ctx, hipStreamSynchronize(st));
        }
    }

    // ---- query validity (:4127-4130): a batch fails as a whole (:1635-1647) ---------------------
    for (uint32_t i = 0; i < nq; ++i) {
        const uint32_t f = flags_keep.empty() ? h_flags[i] : flags_keep[i];
        const bool bad = (metric == YAMS_SCAN_COSINE) ? (f != 0) : ((f & 1u) != 0);
        if (bad)
            return fail(ctx, YAMS_ERR_INVALID_ARG,
                        "Exact vector search requires a finite, non-zero query embedding");
    }
    if (diag) {
        unsigned long long h_stat = 0;
#ifdef YAMS_ACCEL_MEASURE
        if (std::getenv("YAMS_ACCEL_DUMP_NEEDED")) { // candidates the proof needed per query (rescore_select_kernel)
            unsigned long long h4[6] = {0, 0, 0, 0, 0, 0};
            YA_HIP(ctx, hipMemcpyAsync(h4, d_stat, 48, hipMemcpyDeviceToHost, st));
            YA_HIP(ctx, hipStreamSynchronize(st));
            if (h4[5]) std::fprintf(stderr, "bound honesty: %llu of %llu re-scored candidates outside their filter bound (tier %u)\n", h4[4], h4[5], filter_tier);
            if (h4[3]) std::fprintf(stderr, "candidates needed per query: mean %.1f, max %llu over %llu queries (k = %u)\n",
                                    static_cast<double>(h4[1]) / static_cast<double>(h4[3]), h4[2], h4[3], k);
        }
#endif
        YA_HIP(ctx, hipMemcpyAsync(&h_stat, d_stat, 8, hipMemcpyDeviceToHost, st));
        uint32_t* h_counts = h_status; // reuse pinned space
        YA_HIP(ctx, hipMemcpyAsync(h_counts, out_counts, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        diag->used_exact_scan = 1;
        diag->rows_visited_observed = 1;
        diag->rows_visited = static_cast<uint64_t>(nq) * n_eff;
        diag->exact_distance_evaluations = static_cast<uint64_t>(nq) * n_eff;
        uint64_t ret = 0;
        for (uint32_t i = 0; i < nq; ++i) ret += h_counts[i];
        diag->returned_rows = ret;
        diag->filter_candidates = filter_candidates;
        diag->rescored_rows = h_stat + rescored_nested;
        diag->widened_queries = widened;
        diag->exact_fallback_queries = exact_fb;
        diag->escalated_queries = escalated;
        diag->retried_queries = retried;
        diag->filter_tier = filter_tier;
    }
    return YAMS_OK;
}

} // namespace

extern "C" yams_status_t yams_scan_topk_device(yams_accel_ctx* ctx,
                                               const yams_scan_corpus_t* corpus,
                                               const float* queries, uint32_t n_queries,
                                               const yams_scan_params_t* params, float* out_scores,
                                               int64_t* out_rows, uint32_t* out_counts,
                                               float* out_dist, uint32_t* out_ranks,
                                               yams_scan_diag_t* diag) {
    // Very large batches run as slices: the per-batch workspace (sample scores, candidate lists)
    // grows with the query count, and one corpus pass already amortises over 4096 queries.
    uint32_t kBatchMax = 4096;
    // Calls of more than 1024 queries on a shard whose 1024-query batches take the int8 tier's resident-query form run
    // as slices of 1024: that form holds at most eight 128-query tiles per row stream, and the half-tile form it would
    // otherwise fall back to is slower than two resident sweeps (2048 queries on the 12.5M x 768 shard: 18.1 ms in
    // one half-tile batch against 2 x 8.1 ms).
    if (n_queries > 1024 && ctx && corpus && params && queries && corpus->rows_i8 && corpus->rows_i8_meta &&
        (corpus->dim & 63u) == 0 && corpus->dim >= 256 && params->k <= YAMS_SCAN_MAX_K &&
        !(params->flags & (YAMS_SCAN_FLAG_NO_I8_FILTER | YAMS_SCAN_FLAG_F32_FILTER | YAMS_SCAN_FLAG_SPLIT_FILTER |
                           YAMS_SCAN_FLAG_WIDE_TILE | YAMS_SCAN_FLAG_FORCE_EXACT)) &&
        3 * params->k + 64 <= kRescoreMax && corpus->n_rows >= kMfmaMinRows && corpus->n_rows < (1ull << 32)) {
        (void)hipSetDevice(ctx->device);
        ScanLaunch probe;
        probe.plan = make_plan(corpus->n_rows, corpus->dim, 1024, params->k, true, 1, false);
        probe.i8_form = (params->flags & YAMS_SCAN_FLAG_RESIDENT_QUERIES) ? 2 : 0;
        if (i8_takes_resident_form(probe)) kBatchMax = 1024;
    }
    if (n_queries <= kBatchMax || !ctx || !corpus || !params || !queries)
        return scan_impl(ctx, corpus, queries, n_queries, params, out_scores, out_rows, out_counts,
                         out_dist, out_ranks, diag, false);
    yams_scan_diag_t total{};
    const size_t k = params->k, dim = corpus->dim;
    for (uint32_t q0 = 0; q0 < n_queries; q0 += kBatchMax) {
        const uint32_t nq = std::min(kBatchMax, n_queries - q0);
        yams_scan_diag_t d{};
        const yams_status_t s = scan_impl(ctx, corpus, queries + static_cast<size_t>(q0) * dim, nq, params,
                                          out_scores ? out_scores + q0 * k : nullptr,
                                          out_rows ? out_rows + q0 * k : nullptr,
                                          out_counts ? out_counts + q0 : nullptr,
                                          out_dist ? out_dist + q0 * k : nullptr,
                                          out_ranks ? out_ranks + q0 * k : nullptr, diag ? &d : nullptr, false);
        if (s != YAMS_OK) return s; // a batch fails as a whole (:1635-1647)
        total.used_exact_scan = 1; total.rows_visited_observed = 1;
        total.rows_visited += d.rows_visited;
        total.exact_distance_evaluations += d.exact_distance_evaluations;
        total.returned_rows += d.returned_rows;
        total.filter_candidates += d.filter_candidates;
        total.rescored_rows += d.rescored_rows;
        total.widened_queries += d.widened_queries;
        total.exact_fallback_queries += d.exact_fallback_queries;
        total.escalated_queries += d.escalated_queries;
        total.path = std::max(total.path, d.path);
        total.filter_tier = d.filter_tier;
    }
    if (diag) *diag = total;
    return YAMS_OK;
}

extern "C" yams_status_t yams_scan_build_shadow_device(yams_accel_ctx* ctx, const float* rows,
                                                       uint64_t n_rows, uint32_t dim,
                                                       uint16_t* out_rows_bf16, float* out_rows_nsq) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (n_rows == 0) return YAMS_OK;
    if (!rows || !out_rows_bf16 || !out_rows_nsq) return fail(ctx, YAMS_ERR_INVALID_ARG, "null shadow buffers");
    if (dim == 0 || (dim & 3u) || (reinterpret_cast<uintptr_t>(rows) & 15u) ||
        (reinterpret_cast<uintptr_t>(out_rows_bf16) & 7u))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "shadow needs dim % 4 == 0 and 16-byte aligned rows");
    (void)hipSetDevice(ctx->device);
    TimedRegion tr(ctx, "shadow_build");
    YA_HIP(ctx, launch_shadow_build(ctx->stream, rows, n_rows, dim, out_rows_bf16, out_rows_nsq));
    tr.end();
    return YAMS_OK;
}

extern "C" yams_status_t yams_scan_build_shadow_i8_device(yams_accel_ctx* ctx, const float* rows,
                                                          uint64_t first_row, uint64_t n_rows, uint32_t dim,
                                                          int8_t* out_rows_i8, float* out_meta,
                                                          double* out_mean_err) {
    return yams_scan_build_shadow_i8_layout_device(ctx, rows, first_row, n_rows, dim, 0u, out_rows_i8, out_meta, out_mean_err);
}

extern "C" yams_status_t yams_scan_choose_i8_layout_device(yams_accel_ctx* ctx, const float* rows, uint64_t n_rows, uint32_t dim,
                                                           uint32_t* out_i8_flags, double* out_mean_err_plain,
                                                           double* out_mean_err_rotated) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (out_i8_flags) *out_i8_flags = 0;
    if (out_mean_err_plain) *out_mean_err_plain = 0.0;
    if (out_mean_err_rotated) *out_mean_err_rotated = 0.0;
    if (!out_i8_flags) return fail(ctx, YAMS_ERR_INVALID_ARG, "null out_i8_flags");
    if (n_rows == 0) return YAMS_OK;
    if (!rows || dim < 256 || (dim & 63u) || (reinterpret_cast<uintptr_t>(rows) & 15u))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "the int8 shadow needs dim % 64 == 0, dim >= 256 and 16-byte aligned rows");
    if (!i8_rotation_window(dim)) return YAMS_OK;     // only the plain layout exists
    (void)hipSetDevice(ctx->device);
    // the residues of up to 256 blocks spread over the mirror, under both layouts (nothing is written but the two sums)
    const uint64_t n_blocks = (n_rows + 63) / 64;
    const uint64_t stride = std::max<uint64_t>(1, n_blocks / 256);
    double* d_stats;
    YA_TRY(ws_get(ctx, "i8_layout_stats", 32, (void**)&d_stats));
    YA_HIP(ctx, hipMemsetAsync(d_stats, 0, 32, ctx->stream));
    YA_HIP(ctx, launch_shadow_build_i8(ctx->stream, rows, 0, n_rows, dim, nullptr, nullptr, nullptr, false, stride, d_stats));
    YA_HIP(ctx, launch_shadow_build_i8(ctx->stream, rows, 0, n_rows, dim, nullptr, nullptr, nullptr, true, stride, d_stats + 2));
    double h[4] = {0, 0, 0, 0};
    YA_HIP(ctx, hipMemcpyAsync(h, d_stats, 32, hipMemcpyDeviceToHost, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long c0, c1;
    std::memcpy(&c0, &h[1], 8); std::memcpy(&c1, &h[3], 8);
    const double plain = c0 ? h[0] / static_cast<double>(c0) : 0.0, rotated = c1 ? h[2] / static_cast<double>(c1) : 0.0;
    if (out_mean_err_plain) *out_mean_err_plain = plain;
    if (out_mean_err_rotated) *out_mean_err_rotated = rotated;
    // the rotation has to pay for itself: a fifth less residue at least (isotropic Gaussian rows measure the same both ways)
    if (c0 && c1 && rotated < 0.8 * plain) *out_i8_flags = YAMS_SCAN_I8_ROTATED;
    return YAMS_OK;
}

extern "C" yams_status_t yams_scan_build_shadow_i8_layout_device(yams_accel_ctx* ctx, const float* rows,
                                                                 uint64_t first_row, uint64_t n_rows, uint32_t dim, uint32_t i8_flags,
                                                                 int8_t* out_rows_i8, float* out_meta,
                                                                 double* out_mean_err) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (i8_flags & ~YAMS_SCAN_I8_ROTATED) return fail(ctx, YAMS_ERR_INVALID_ARG, "unknown i8_flags");
    const bool rotated = (i8_flags & YAMS_SCAN_I8_ROTATED) != 0;
    if (rotated && !i8_rotation_window(dim)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "no rotated int8 layout exists for this dimension (256 <= dim <= 4096)");
    if (out_mean_err) *out_mean_err = 0.0;
    if (n_rows == 0) return YAMS_OK;
    if (!rows || !out_rows_i8 || !out_meta) return fail(ctx, YAMS_ERR_INVALID_ARG, "null shadow buffers");
    if (dim < 256 || (dim & 63u) || (reinterpret_cast<uintptr_t>(rows) & 15u) ||
        (reinterpret_cast<uintptr_t>(out_rows_i8) & 15u) || (reinterpret_cast<uintptr_t>(out_meta) & 7u))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "the int8 shadow needs dim % 64 == 0, dim >= 256 and 16-byte aligned rows");
    (void)hipSetDevice(ctx->device);
    double* d_stats = nullptr;
    if (out_mean_err) {
        YA_TRY(ws_get(ctx, "i8_stats", 16, (void**)&d_stats));
        YA_HIP(ctx, hipMemsetAsync(d_stats, 0, 16, ctx->stream));
    }
    TimedRegion tr(ctx, "shadow_build_i8");
    YA_HIP(ctx, launch_shadow_build_i8(ctx->stream, rows, first_row, n_rows, dim, out_rows_i8, out_meta, d_stats, rotated));
    tr.end();
    if (out_mean_err) {
        double h[2] = {0.0, 0.0};
        YA_HIP(ctx, hipMemcpyAsync(h, d_stats, 16, hipMemcpyDeviceToHost, ctx->stream));
        YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long cnt;
        std::memcpy(&cnt, &h[1], 8);
        *out_mean_err = cnt ? h[0] / static_cast<double>(cnt) : 0.0;
    }
    return YAMS_OK;
}

extern "C" yams_status_t yams_scan_topk_host(yams_accel_ctx* ctx, const yams_scan_corpus_t* corpus,
                                             const float* queries_host, uint32_t n_queries,
                                             const yams_scan_params_t* params,
                                             float* out_scores_host, int64_t* out_rows_host,
                                             uint32_t* out_counts_host, float* out_dist_host,
                                             yams_scan_diag_t* diag) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!corpus || !params) return fail(ctx, YAMS_ERR_INVALID_ARG, "null corpus/params");
    if (diag) std::memset(diag, 0, sizeof(*diag));
    if (n_queries == 0) return YAMS_OK;
    if (!queries_host || !out_counts_host) return fail(ctx, YAMS_ERR_INVALID_ARG, "null queries/out_counts");
    (void)hipSetDevice(ctx->device);
    const size_t nq = n_queries, k = params->k, dim = corpus->dim;
    float* d_q; float* d_s; int64_t* d_r; uint32_t* d_c; float* d_d;
    YA_TRY(ws_get(ctx, "h_queries", nq * std::max<size_t>(dim, 1) * 4, (void**)&d_q));
    YA_TRY(ws_get(ctx, "h_scores", nq * std::max<size_t>(k, 1) * 4, (void**)&d_s));
    YA_TRY(ws_get(ctx, "h_rows", nq * std::max<size_t>(k, 1) * 8, (void**)&d_r));
    YA_TRY(ws_get(ctx, "h_counts", nq * 4, (void**)&d_c));
    YA_TRY(ws_get(ctx, "h_dist", nq * std::max<size_t>(k, 1) * 4, (void**)&d_d));
    if (dim)
        YA_HIP(ctx, hipMemcpyAsync(d_q, queries_host, nq * dim * 4, hipMemcpyHostToDevice, ctx->stream));
    yams_status_t s = yams_scan_topk_device(ctx, corpus, d_q, n_queries, params, d_s, d_r, d_c,
                                            d_d, nullptr, diag);
    if (s != YAMS_OK) return s;
    YA_HIP(ctx, hipMemcpyAsync(out_counts_host, d_c, nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (k && out_scores_host)
        YA_HIP(ctx, hipMemcpyAsync(out_scores_host, d_s, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (k && out_rows_host)
        YA_HIP(ctx, hipMemcpyAsync(out_rows_host, d_r, nq * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (k && out_dist_host && dim)
        YA_HIP(ctx, hipMemcpyAsync(out_dist_host, d_d, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return YAMS_OK;
}

extern "C" yams_status_t yams_scan_merge_topk_device(
    yams_accel_ctx* ctx, uint32_t n_shards, uint32_t n_queries, const yams_scan_params_t* params,
    const float* in_scores, const int64_t* in_rows, const uint32_t* in_counts, const float* in_dist,
    const uint32_t* in_ranks, float* out_scores, int64_t* out_rows, uint32_t* out_counts,
    float* out_dist) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!params || n_shards == 0) return fail(ctx, YAMS_ERR_INVALID_ARG, "bad merge arguments");
    if (n_queries == 0) return YAMS_OK;
    if (!in_counts || !out_counts) return fail(ctx, YAMS_ERR_INVALID_ARG, "null counts");
    (void)hipSetDevice(ctx->device);
    if (params->k == 0) {
        YA_HIP(ctx, hipMemsetAsync(out_counts, 0, static_cast<size_t>(n_queries) * 4, ctx->stream));
        return YAMS_OK;
    }
    if (!in_scores || !in_rows || !out_scores || !out_rows)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "null merge buffers");
    if (params->metric == YAMS_SCAN_L2 && !in_dist)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "L2 merge needs distances");
    if (static_cast<uint64_t>(n_shards) * params->k > 8192)
        return fail(ctx, YAMS_ERR_UNSUPPORTED, "n_shards * k exceeds 8192");
    MergeLaunch M{};
    M.n_shards = n_shards; M.n_queries = n_queries; M.k = params->k; M.metric = params->metric;
    M.threshold = params->similarity_threshold; M.in_scores = in_scores; M.in_rows = in_rows;
    M.in_counts = in_counts; M.in_dist = in_dist; M.in_ranks = in_ranks; M.out_scores = out_scores;
    M.out_rows = out_rows; M.out_counts = out_counts; M.out_dist = out_dist;
    TimedRegion tr(ctx, "merge_topk");
    YA_HIP(ctx, launch_merge(ctx->stream, M));
    tr.end();
    return YAMS_OK;
}

extern "C" yams_status_t yams_synth_rows_device(yams_accel_ctx* ctx, uint64_t seed, uint64_t row0,
                                                uint64_t n_rows, uint32_t dim, float* out_dev) {
    if (!ctx || (!out_dev && n_rows) || dim == 0) return YAMS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    YA_HIP(ctx, launch_synth_rows(ctx->stream, seed, row0, n_rows, dim, out_dev));
    return YAMS_OK;
}

extern "C" yams_status_t yams_synth_bytes_device(yams_accel_ctx* ctx, uint64_t seed,
                                                 uint64_t blob_id0, uint64_t n_blobs,
                                                 uint64_t blob_len, uint8_t* out_dev) {
    if (!ctx || (!out_dev && n_blobs && blob_len)) return YAMS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    YA_HIP(ctx, launch_synth_bytes(ctx->stream, seed, blob_id0, n_blobs, blob_len, out_dev));
    return YAMS_OK;
}
