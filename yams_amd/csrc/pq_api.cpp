// pq_api.cpp — yams_scan_pq_topk_device: the product-quantised engine's search behind the C ABI.
//
// Mirrors SqliteVecBackend::Impl::simeonPqSearchUnlocked (src/vector/sqlite_vec_backend.cpp:3868-4056):
//   candidates (all indexed rows, or the host's sorted candidate indices :3910-3937)
//   -> ADC score of every candidate from the host-built table (:3962-3977)
//   -> best approxK = min(candidates, max(k, k * rerank_factor)) by (score desc, tie key asc) (:3952-3997)
//   -> exact re-score with computeCosineSimilarity against the RAW query (:4023-4034), threshold (:4036-4038)
//   -> sorted by (similarity desc, chunk_id asc), cut to k (:4041-4051).
// The host keeps what simeon owns: training, encoding, and building the per-query table.
#include <algorithm>
#include <cstring>
#include <vector>

#include "accel_ctx.h"
#include "scan_launch.h"

using namespace yams_accel;

namespace yams_accel {
hipError_t launch_pq_adc_keys(hipStream_t st, const uint8_t* codes, uint64_t n_codes, uint32_t m, const float* luts, const uint32_t* qmap,
                              uint32_t n_slots, int lanes, const uint32_t* tie_rank, const uint32_t* candidates, uint64_t n_items,
                              uint64_t* keys, uint64_t key_stride);
hipError_t launch_pq_adc_filter(hipStream_t st, int mode, const uint8_t* codes, uint64_t n_codes, uint32_t m, const float* luts,
                                uint32_t n_slots, int lanes, const uint32_t* tie_rank, const uint32_t* candidates, uint64_t n_items,
                                uint32_t stride, uint32_t* sample_out, uint32_t n_sample, const float* tau, uint32_t* list_count,
                                uint64_t* list, uint32_t list_cap);
}

extern "C" yams_status_t yams_scan_pq_topk_device(yams_accel_ctx* ctx, const yams_scan_corpus_t* corpus, const yams_scan_pq_index_t* pq,
                                                  const float* queries, const float* luts, uint32_t n_queries,
                                                  const yams_scan_pq_params_t* prm, const uint32_t* candidates, uint64_t n_candidates,
                                                  float* out_scores, int64_t* out_rows, uint32_t* out_counts, yams_scan_diag_t* diag) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!corpus || !pq || !prm) return fail(ctx, YAMS_ERR_INVALID_ARG, "null corpus / index / params");
    if (diag) std::memset(diag, 0, sizeof(*diag));
    if (n_queries == 0) return YAMS_OK;
    if (!out_counts) return fail(ctx, YAMS_ERR_INVALID_ARG, "null out_counts");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const uint32_t nq = n_queries, dim = corpus->dim, k = prm->k;
    const uint64_t n_items = candidates ? n_candidates : pq->n_codes;
    // empty query / k == 0 / no index / an empty candidate set: an empty result, nothing is validated (:3873-3880, :3946-3948)
    if (dim == 0 || k == 0 || pq->n_codes == 0 || n_items == 0) {
        YA_HIP(ctx, hipMemsetAsync(out_counts, 0, static_cast<size_t>(nq) * 4, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        return YAMS_OK;
    }
    if (!queries || !luts || !out_scores || !out_rows) return fail(ctx, YAMS_ERR_INVALID_ARG, "null queries / tables / outputs");
    if (!pq->codes || pq->m == 0) return fail(ctx, YAMS_ERR_INVALID_ARG, "null codes or m == 0");
    if (pq->m > 128) return fail(ctx, YAMS_ERR_UNSUPPORTED, "more than 128 sub-quantisers (a query's table must fit the LDS)");
    if (k > YAMS_SCAN_MAX_K) return fail(ctx, YAMS_ERR_UNSUPPORTED, "k exceeds YAMS_SCAN_MAX_K");
    if (pq->n_codes >= (1ull << 32) || corpus->n_rows >= (1ull << 32)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "index must hold < 2^32 rows");
    if (!corpus->rows) return fail(ctx, YAMS_ERR_INVALID_ARG, "null corpus rows");
    if ((corpus->tie_rank == nullptr) != (corpus->rank_row == nullptr))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "tie_rank and rank_row must be given together");
    if ((reinterpret_cast<uintptr_t>(pq->codes) & 3u) || (reinterpret_cast<uintptr_t>(luts) & 15u))
        return fail(ctx, YAMS_ERR_INVALID_ARG, "codes must be 4-byte and tables 16-byte aligned");
    const uint32_t lanes_flag = prm->flags & YAMS_PQ_SUM_MASK;
    const int lanes = lanes_flag == YAMS_PQ_SUM_X4 ? 4 : (lanes_flag == YAMS_PQ_SUM_X8 ? 8 : (lanes_flag == YAMS_PQ_SUM_X16 ? 16 : 1));
    // approxK (:3952-3960): max(k, k * rerank) unless that overflows, at most the candidates
    const uint64_t rf = std::max<uint32_t>(1u, prm->rerank_factor);
    const uint64_t budget = std::max<uint64_t>(k, static_cast<uint64_t>(k) * rf);
    const uint64_t approx = std::min<uint64_t>(n_items, budget);
    if (approx > kRescoreMax) return fail(ctx, YAMS_ERR_UNSUPPORTED, "k * rerank_factor exceeds 2047 candidates per query");
    const uint32_t approx_k = static_cast<uint32_t>(approx);

    // ---- the queries: fp64 norms for the re-score; a query the host's normalisation would refuse (norm^2 <= 1e-20, :213-226,
    //      or not finite) gets an empty result (:3895-3898), not an error
    float* d_qprep; double* d_qnorm; uint32_t* d_qflags; uint32_t* d_status; unsigned long long* d_stat;
    YA_TRY(ws_get(ctx, "pq_qprep", static_cast<size_t>(nq) * dim * 4, (void**)&d_qprep));
    YA_TRY(ws_get(ctx, "pq_qnorm", static_cast<size_t>(nq) * 8, (void**)&d_qnorm));
    YA_TRY(ws_get(ctx, "pq_qflags", static_cast<size_t>(nq) * 4, (void**)&d_qflags));
    YA_TRY(ws_get(ctx, "pq_status", static_cast<size_t>(nq) * 4, (void**)&d_status));
    YA_TRY(ws_get(ctx, "pq_stat", 64, (void**)&d_stat));
    YA_HIP(ctx, hipMemsetAsync(d_stat, 0, 64, st));
    YA_HIP(ctx, hipMemsetAsync(d_status, 0, static_cast<size_t>(nq) * 4, st));
    YA_HIP(ctx, launch_prep_queries(st, queries, nq, dim, YAMS_SCAN_L2 /* raw: no normalised copy is needed */, d_qprep, d_qnorm, nullptr, d_qflags));
    uint32_t* h_pin;
    YA_TRY(pinned_get(ctx, static_cast<size_t>(nq) * 12 + 64, (void**)&h_pin));

    auto rescore = [&](const uint64_t* res, uint64_t res_stride, const uint32_t* d_map, uint32_t n_slots) -> yams_status_t {
        // ---- exact re-score of the approxK best, final order, cut to k ---------------------------------------------------
        RescoreLaunch R{};
        R.rows = corpus->rows; R.n_rows = corpus->n_rows; R.dim = dim; R.queries = queries; R.qnorm = d_qnorm;
        R.tie_rank = corpus->tie_rank;           // the final order: (similarity desc, chunk_id asc) (:4041-4051)
        R.rank_row = pq->key_row;                // key index (tie rank of the code, or its index) -> corpus row (rowids[idx], :4009)
        R.row_base = corpus->row_base; R.stripe_rows = corpus->stripe_rows; R.n_stripes = corpus->n_stripes; R.stripe_index = corpus->stripe_index;
        R.cand = res; R.cand_stride = res_stride; R.n_cand = approx_k;
        R.tau = nullptr; R.list_count = nullptr; R.list_cap = 0; R.all_rows_listed = 1;
        R.qmap = d_map; R.n_slots = n_slots; R.k = k; R.threshold = prm->similarity_threshold;
        R.flags = kRescoreFlagPqRerank; R.err_bound = 0.0;
        R.out_scores = out_scores; R.out_rows = out_rows; R.out_counts = out_counts; R.out_dist = nullptr; R.out_ranks = nullptr;
        R.out_status = d_status; R.stat_rescored = d_stat;
        YA_HIP(ctx, launch_rescore(st, YAMS_SCAN_COSINE, R));
        return YAMS_OK;
    };
    uint32_t* d_qmap;
    YA_TRY(ws_get(ctx, "pq_qmap", static_cast<size_t>(nq) * 4, (void**)&d_qmap));
    std::vector<uint32_t> todo(nq);           // the queries the unfiltered form below serves
    for (uint32_t i = 0; i < nq; ++i) todo[i] = i;

    // ---- the filtered form (pq_kernels.hip): sample -> tau -> keys of the codes that reach it -> best approxK of each list ---
    uint32_t filtered_queries = 0;
    if (n_items >= 65536) {
        const uint32_t stride = static_cast<uint32_t>(std::min<uint64_t>(64, std::max<uint64_t>(1, n_items / 16384)));
        const uint32_t n_sample = static_cast<uint32_t>((n_items + stride - 1) / stride);
        const uint32_t want = std::max<uint32_t>(4 * approx_k, 1024);                   // expected list length
        const uint32_t rank = std::max<uint32_t>(16, (want + stride - 1) / stride);     // (spread ~ 1 / sqrt(rank): approxK is > 3 sigma below)
        const uint32_t list_cap = (4 * rank * stride + 255u) & ~255u;
        const uint32_t chunks_l = (list_cap + kSelectCap - 1) / kSelectCap;
        uint32_t* d_sample; float* d_tau; uint32_t* d_lcount; uint64_t* d_list; uint64_t* d_lwork;
        YA_TRY(ws_get(ctx, "pq_sample", static_cast<size_t>(nq) * n_sample * 4, (void**)&d_sample));
        YA_TRY(ws_get(ctx, "pq_tau", static_cast<size_t>(nq) * 4, (void**)&d_tau));
        YA_TRY(ws_get(ctx, "pq_lcount", static_cast<size_t>(nq) * 4, (void**)&d_lcount));
        YA_TRY(ws_get(ctx, "pq_list", static_cast<size_t>(nq) * list_cap * 8, (void**)&d_list));
        YA_TRY(ws_get(ctx, "pq_lwork", static_cast<size_t>(2) * nq * chunks_l * (approx_k + 1) * 8, (void**)&d_lwork));
        YA_HIP(ctx, hipMemsetAsync(d_lcount, 0, static_cast<size_t>(nq) * 4, st));
        { TimedRegion tr(ctx, "pq_adc_sample");
          YA_HIP(ctx, launch_pq_adc_filter(st, 1, pq->codes, pq->n_codes, pq->m, luts, nq, lanes, pq->tie_rank, candidates, n_items, stride,
                                           d_sample, n_sample, nullptr, nullptr, nullptr, 0));
          tr.end(); }
        ScanLaunch T; T.plan.n_queries = nq; T.plan.n_groups = n_sample; T.plan.tau_rank = rank; T.gmax = d_sample; T.tau_out = d_tau;
        YA_HIP(ctx, launch_select_tau(st, T, nullptr));
        { TimedRegion tr(ctx, "pq_adc");
          YA_HIP(ctx, launch_pq_adc_filter(st, 2, pq->codes, pq->n_codes, pq->m, luts, nq, lanes, pq->tie_rank, candidates, n_items, stride,
                                           nullptr, n_sample, d_tau, d_lcount, d_list, list_cap));
          tr.end(); }
        uint32_t* h_lc = h_pin + 2 * static_cast<size_t>(nq);
        YA_HIP(ctx, hipMemcpyAsync(h_lc, d_lcount, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        std::vector<uint32_t> good;
        todo.clear();
        for (uint32_t i = 0; i < nq; ++i)       // a list too short to hold the best approxK, or cut off, is no use
            (h_lc[i] < approx_k || h_lc[i] > list_cap ? todo : good).push_back(i);
        filtered_queries = static_cast<uint32_t>(good.size());
        if (!good.empty()) {
            YA_HIP(ctx, hipMemcpyAsync(d_qmap, good.data(), good.size() * 4, hipMemcpyHostToDevice, st));
            const uint64_t* res; uint64_t res_stride;
            YA_HIP(ctx, launch_select_lists(st, d_list, d_lcount, list_cap, filtered_queries, d_qmap, approx_k, d_lwork, &res, &res_stride));
            YA_TRY(rescore(res, res_stride, d_qmap, filtered_queries));
            YA_HIP(ctx, hipStreamSynchronize(st)); // (`good` is pageable; d_qmap is reused below)
        }
    }

    // ---- the unfiltered form: ADC keys of every code + top approxK, in batches of queries that keep the key array within its budget
    if (!todo.empty()) {
        const uint32_t nt = static_cast<uint32_t>(todo.size());
        const uint64_t key_stride = n_items;
        constexpr uint64_t kKeyBudget = 1ull << 31;
        uint32_t batch = static_cast<uint32_t>(std::max<uint64_t>(1, kKeyBudget / (key_stride * 8)));
        batch = std::min(batch, nt);
        const uint32_t chunks = static_cast<uint32_t>((key_stride + kSelectCap - 1) / kSelectCap);
        uint64_t* d_keys; uint64_t* d_work;
        YA_TRY(ws_get(ctx, "pq_keys", static_cast<size_t>(batch) * key_stride * 8, (void**)&d_keys));
        YA_TRY(ws_get(ctx, "pq_work", static_cast<size_t>(2) * batch * chunks * (approx_k + 1) * 8, (void**)&d_work));
        YA_HIP(ctx, hipMemcpyAsync(d_qmap, todo.data(), static_cast<size_t>(nt) * 4, hipMemcpyHostToDevice, st));
        for (uint32_t b0 = 0; b0 < nt; b0 += batch) {
            const uint32_t nb = std::min(batch, nt - b0);
            { TimedRegion tr(ctx, "pq_adc_keys");
              YA_HIP(ctx, launch_pq_adc_keys(st, pq->codes, pq->n_codes, pq->m, luts, d_qmap + b0, nb, lanes, pq->tie_rank, candidates, n_items,
                                             d_keys, key_stride));
              tr.end(); }
            const uint64_t* res; uint64_t res_stride;
            YA_HIP(ctx, launch_topk_keys(st, d_keys, key_stride, static_cast<uint32_t>(n_items), nb, approx_k, d_work, &res, &res_stride));
            YA_TRY(rescore(res, res_stride, d_qmap + b0, nb));
        }
        YA_HIP(ctx, hipStreamSynchronize(st)); // (`todo` is pageable)
    }
    double* h_qn = reinterpret_cast<double*>(h_pin);
    YA_HIP(ctx, hipMemcpyAsync(h_qn, d_qnorm, static_cast<size_t>(nq) * 8, hipMemcpyDeviceToHost, st));
    YA_HIP(ctx, hipStreamSynchronize(st));
    // queries the host's normalisation refuses (norm^2 <= 1e-20 or not finite): empty results
    std::vector<uint32_t> refused;
    for (uint32_t i = 0; i < nq; ++i) if (!(h_qn[i] * h_qn[i] > 1e-20) || !(h_qn[i] < 1e300)) refused.push_back(i);
    for (uint32_t q : refused) YA_HIP(ctx, hipMemsetAsync(out_counts + q, 0, 4, st));
    if (diag) {
        unsigned long long h_stat = 0;
        YA_HIP(ctx, hipMemcpyAsync(&h_stat, d_stat, 8, hipMemcpyDeviceToHost, st));
        uint32_t* h_counts = h_pin + 2 * static_cast<size_t>(nq);
        YA_HIP(ctx, hipMemcpyAsync(h_counts, out_counts, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        uint64_t ret = 0;
        for (uint32_t i = 0; i < nq; ++i) ret += h_counts[i];
        diag->used_exact_scan = 0; diag->rows_visited_observed = 1;
        diag->rows_visited = static_cast<uint64_t>(nq) * n_items;      // (:3938-3945: annCandidateBudget = rowsVisited = candidateCount)
        diag->exact_distance_evaluations = h_stat;                      // (:4027: one per materialised candidate)
        diag->rescored_rows = h_stat; diag->returned_rows = ret; diag->filter_candidates = static_cast<uint64_t>(nq) * approx_k;
        diag->path = 2; diag->filter_tier = 5;                          // 2 / 5: the product-quantised engine
        diag->exact_fallback_queries = nq - filtered_queries;           // (here: queries served by the unfiltered form)
    } else if (!refused.empty()) {
        YA_HIP(ctx, hipStreamSynchronize(st));
    }
    return YAMS_OK;
}
