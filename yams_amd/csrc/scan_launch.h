// scan_launch.h — host-visible launch descriptors for scan_kernels.hip.
#pragma once
#include "common.h"

namespace yams_accel {

struct ScanLaunch {
    ScanPlan plan;
    const float* rows = nullptr;
    const uint16_t* rows_bf16 = nullptr; // nullable shadow (with rows_nsq)
    const float* rows_nsq = nullptr;
    const int8_t* rows_i8 = nullptr;     // nullable INT8 shadow (with rows_i8_meta)
    const float* rows_i8_meta = nullptr;
    const int8_t* q_i8 = nullptr;
    const float* q_meta = nullptr;
    const float* q_thr = nullptr;
    uint64_t* log_key = nullptr; uint32_t* log_q = nullptr; uint32_t* log_cnt = nullptr; uint32_t log_cap = 0;
    uint32_t* q_over = nullptr;
    uint32_t* i8_sync = nullptr;         // resident-query form of the int8 filter: pacing counters (i8_sync_words())
    // int8 tier under L2 (launch_i8_l2_*): thresholds meta {s_b, -nmin_b}, row / query biases
    bool i8_l2 = false;
    const float* i8_l2_meta = nullptr;
    const uint8_t* i8_row_bias = nullptr;
    const uint32_t* i8_q_bias = nullptr;
    float l2_eps = 0.f;
    bool i8_sample_small_grid = false;   // a collective may be in flight on this device (sharded fence / sweep hold): the sample
                                         // pass must not own every CU — it takes the half-tile form, which leaves room
    int i8_form = 0;                     // int8 filter pass: 0 = the library's choice, 1 = half tiles, 2 = resident queries when possible
    bool i8_q_form = false;              // resident queries with 128 x 128 wave tiles (i8_takes_q_form): the log holds BLOCK entries
    double i8_mask_inflation = 1.0;      // n_rows / rows the allow-mask lets through (block entries are written before the mask is applied)
    int sample_layout = 0;               // rows of a sample group: 0 = 32x32 accumulator layout, 1 = 16x16 (int8 tier)
    const uint32_t* row_mask = nullptr;
    const float* qprep = nullptr;
    const uint16_t* q_hi = nullptr;
    const uint16_t* q_lo = nullptr;
    uint32_t q_pad = 0;
    float* dense = nullptr;
    uint32_t* gmax = nullptr;
    const float* tau = nullptr;   // read by filter / collect
    float* tau_out = nullptr;     // written by select_tau
    // proof-aware threshold (tau_select_kernel): int8 tier under cosine only
    const float* tau_rows_meta = nullptr; uint64_t tau_n_blocks = 0; uint32_t tau_rank2 = 0, tau_max_groups = 0; float tau_e_scale = 1.0f;
    uint32_t* list_count = nullptr;
    uint64_t* list = nullptr;
    const float* qnorm_up = nullptr;
    float err_coef = 0.f;
};

struct RescoreLaunch {
    const float* rows; uint64_t n_rows; uint32_t dim;
    const float* queries; const double* qnorm;
    const uint32_t* tie_rank; const uint32_t* rank_row; int64_t row_base;
    uint32_t stripe_rows, n_stripes, stripe_index;
    const uint64_t* cand; uint64_t cand_stride; uint32_t n_cand;
    const float* tau; const uint32_t* list_count; uint32_t list_cap; uint32_t all_rows_listed;
    const uint32_t* qmap; uint32_t n_slots;
    uint32_t k; float threshold; uint32_t flags; double err_bound;
    float* out_scores; int64_t* out_rows; uint32_t* out_counts; float* out_dist;
    uint32_t* out_ranks; uint32_t* out_status; unsigned long long* stat_rescored;
    const uint32_t* q_over = nullptr; // nullable: queries whose candidate list is known to be incomplete
};

struct MergeLaunch {
    uint32_t n_shards, n_queries, k, metric; float threshold;
    const float* in_scores; const int64_t* in_rows; const uint32_t* in_counts;
    const float* in_dist; const uint32_t* in_ranks;
    float* out_scores; int64_t* out_rows; uint32_t* out_counts; float* out_dist;
    // 0 = dense arrays; else the distance between consecutive shards in elements of each input
    uint64_t st_scores = 0, st_rows = 0, st_counts = 0, st_dist = 0, st_ranks = 0;
    const uint32_t* rank_of_row = nullptr; int64_t rank_row_base = 0;
};

// scan_small_kernel.hip: the whole exact search of a small corpus in one launch
struct SmallScanArgs {
    const float* rows; uint32_t n_rows; uint32_t dim;
    const float* queries; uint32_t nq;
    const uint32_t* tie_rank; const uint32_t* rank_row; const uint32_t* row_mask;
    int64_t row_base; uint32_t stripe_rows, n_stripes, stripe_index;
    uint32_t k, kk, sort_cap;   // kk = keys a workgroup keeps per query = min(k, 256); n_wg * kk <= small_scan_max_survivors()
    float threshold; uint32_t flags;
    uint64_t* part_key; float* part_aux;   // [nq][n_wg][kk]
    uint32_t* counter;                     // [query chunks], zero between launches (the kernel resets it)
    uint32_t* qflags;                      // [nq]: bit0 non-finite element, bit1 norm^2 < 1e-10
    float* out_scores; int64_t* out_rows; uint32_t* out_counts; float* out_dist; uint32_t* out_ranks;
    unsigned long long* dbg = nullptr;     // measurement build only: [n_wg][8] phase stamps (100 MHz wall clock)
};
uint32_t small_scan_max_survivors();
hipError_t launch_small_scan(hipStream_t st, int metric, const SmallScanArgs& a, uint32_t qb);

constexpr uint32_t kRescoreMax = 2047; // + 1 boundary key == kSelectCap / 2 (select convergence)

hipError_t launch_prep_queries(hipStream_t st, const float* q, uint32_t nq, uint32_t dim,
                               int metric, float* qprep, double* qnorm, float* qnorm_up,
                               uint32_t* qflags, uint32_t* zero_words = nullptr, uint32_t n_zero_words = 0);
hipError_t launch_prep_split(hipStream_t st, const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                             uint32_t slab_k, uint16_t* q_hi, uint16_t* q_lo);
// k-extent of one LDS stage of the bf16 filter kernels (= the slab size of the query planes):
// the single-pass kernel uses 32-wide slabs when the dimension allows it.
inline uint32_t bf16_slab_k(int passes, uint32_t dim) { return (passes == 1 && (dim & 31u) == 0) ? 32u : 16u; }
hipError_t launch_scan_bf16(hipStream_t st, const ScanLaunch& L, int metric, int mode, int passes, int version);
// The INT8 tier (cosine, dim % 64 == 0, int8 shadow present): sample (mode 0) or filter (mode 1) pass.
hipError_t launch_scan_i8(hipStream_t st, const ScanLaunch& L, int mode, int version);
// Quantises the prepared (unit) queries of a batch: k-slab-major int8 plane + {t_q, c_q, f_q} per query.
hipError_t launch_prep_i8(hipStream_t st, const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                          int8_t* q_i8, float* q_meta, bool raw_queries = false, uint32_t* zero_words = nullptr,
                          uint64_t n_zero_words = 0, bool rotated = false);
// L2 on the int8 tier.  Per batch and shard: (1) norm statistics of the shard — per 64-row block the smallest row norm,
// shard-wide the norm range, the largest in-block spread in units of the block scale, and the rows whose squared
// norm lies outside norm_in_range() (counted and listed: unconditional candidates); stats = 8 words, zeroed first.
// (2) after the sample pass: the per-query threshold halves and biases; (3) the per-block thresholds meta and the
// per-row biases.  nmin = [ceil(n_rows / 64)] floats of workspace.  stats words: ~bits(min |x|^2), bits(max |x|^2),
// bits(max in-block norm spread / s_b), rows out of range, bits(max e_b).
hipError_t launch_i8_l2_norm_stats(hipStream_t st, const float* rows_nsq, const float* rows_i8_meta, uint64_t n_rows,
                                   float* nmin, uint32_t* stats, uint32_t* special);
// rows without a usable norm (stats word 3 counts them, `special` lists the first i8_l2_max_special()): every query's
// candidate list gets them unconditionally; more than that many keep the batch on the bf16 tier
uint32_t i8_l2_max_special();
hipError_t launch_i8_l2_add_special(hipStream_t st, const ScanLaunch& L, const uint32_t* special, uint32_t n_special);
hipError_t launch_i8_l2_thresholds(hipStream_t st, const float* tau, const float* q_meta, uint32_t nq, uint32_t q_pad,
                                   uint32_t dim, const uint32_t* stats, float* q_thr, uint32_t* q_bias);
hipError_t launch_i8_l2_rows(hipStream_t st, const float* rows_nsq, const float* rows_i8_meta, const float* nmin,
                             uint64_t n_rows, const uint32_t* stats, float* l2_meta, uint8_t* row_bias);
float i8_l2_eps(uint32_t dim);
// (Re)builds the INT8 shadow of every 16-row block that intersects rows [first_row, first_row + n_rows)
// of the mirror at `rows`; stats (nullable) = {sum of e_b (double), blocks counted (u64 bits)}.
hipError_t launch_shadow_build_i8(hipStream_t st, const float* rows, uint64_t first_row, uint64_t n_rows, uint32_t dim,
                                  int8_t* out_i8, float* out_meta, double* stats, bool rotated = false, uint64_t block_stride = 1,
                                  double* dry = nullptr);
uint32_t i8_rotation_window(uint32_t dim);   // 0: no rotated layout for this dimension
// After the sample pass: q_thr[q] = per-query halves of the filter's integer thresholds.
// int8 tier: workgroups of the filter launch (the survivor log has 8 regions of log_cap entries per group)
uint64_t i8_log_regions(const ScanLaunch& L);
// entries per survivor-log region of this launch (depends on the kernel form the launch takes)
uint32_t i8_log_capacity(const ScanLaunch& L);
// whether the int8 filter pass of this launch would take the resident-query kernel form
bool i8_takes_resident_form(const ScanLaunch& L);
// ... and of those launches, which take its 128 x 128 wave-tile form (scan_tiles_i8q_kernel; `version`: the measurement
// build's kernel selector, 0 otherwise).  Set ScanLaunch::i8_q_form from it before sizing the log.
bool i8_takes_q_form(const ScanLaunch& L, int version);
// bytes per entry of the survivor log (8 + 4 in two arrays: log_key gets the 8; block entries: 144, all in log_key)
uint32_t i8_log_entry_bytes(const ScanLaunch& L);
// pacing counters of the resident-query form (0 when the launch takes the half-tile form); zero them before the launch
uint64_t i8_sync_words(const ScanLaunch& L);
// after the filter pass: log entries -> per-query candidate lists (the returning atomics live here, where
// thousands of independent threads hide their latency)
hipError_t launch_i8_log_gather(hipStream_t st, const ScanLaunch& L);
// int8 tier: sample rows that reach tau join the lists, their scores re-derived from the group maxima's groups
hipError_t launch_i8_collect_sample(hipStream_t st, const ScanLaunch& L);
hipError_t launch_i8_thresholds(hipStream_t st, const float* tau, const float* q_meta, uint32_t nq, uint32_t q_pad,
                                float* q_thr);
hipError_t launch_scan_sample(hipStream_t st, const ScanLaunch& L, int metric);
hipError_t launch_scan_filter(hipStream_t st, const ScanLaunch& L, int metric);
hipError_t launch_select_tau(hipStream_t st, const ScanLaunch& L, uint32_t* work32);
hipError_t launch_collect_sample(hipStream_t st, const ScanLaunch& L);
hipError_t launch_select_lists(hipStream_t st, const uint64_t* list, const uint32_t* list_count,
                               uint32_t list_cap, uint32_t n_slots, const uint32_t* qmap,
                               uint32_t keep, uint64_t* work, const uint64_t** result,
                               uint64_t* result_stride);
// rows_sel: nullable list of n_sel row ordinals to score (sparse allow-mask); else all n_rows rows,
// skipping rows whose bit in row_mask (nullable) is clear.  keys[slot][i], i < (rows_sel ? n_sel : n_rows).
hipError_t launch_exact_keys(hipStream_t st, int metric, const float* rows, uint64_t n_rows,
                             uint32_t dim, const float* queries, const double* qnorm,
                             const uint32_t* tie_rank, const uint32_t* row_mask,
                             const uint32_t* rows_sel, uint64_t n_sel, const uint32_t* qmap,
                             uint32_t n_slots, float threshold, uint32_t flags, uint64_t* keys,
                             uint64_t key_stride);
hipError_t launch_compact_mask(hipStream_t st, const uint32_t* row_mask, uint64_t n_rows,
                               uint32_t* rows_sel, unsigned long long* counter);
hipError_t launch_topk_keys(hipStream_t st, const uint64_t* keys, uint64_t key_stride,
                            uint32_t n_per_slot, uint32_t n_slots, uint32_t keep, uint64_t* work,
                            const uint64_t** result, uint64_t* result_stride);
hipError_t launch_rescore(hipStream_t st, int metric, const RescoreLaunch& R);
hipError_t launch_merge(hipStream_t st, const MergeLaunch& M);
hipError_t launch_shadow_build(hipStream_t st, const float* rows, uint64_t n_rows, uint32_t dim,
                               uint16_t* out_bf16, float* out_nsq);
hipError_t launch_retry_tau_l2(hipStream_t st, const float* dist, const uint32_t* counts, uint32_t k, const uint32_t* fmap, uint32_t n_slots,
                               const double* qnorm, double margin, const uint32_t* gmax, uint32_t n_groups, float* tau_out, uint32_t* est_out);
hipError_t launch_retry_tau(hipStream_t st, const float* scores, const uint32_t* counts, uint32_t k, const uint32_t* fmap, uint32_t n_slots,
                            const uint32_t* gmax, uint32_t n_groups, float* tau_out, uint32_t* est_out, float slack = 0.f);
hipError_t launch_scatter_results_from(hipStream_t st, const uint32_t* src, const uint32_t* dst, uint32_t n, uint32_t k, const float* s_scores,
                                       const int64_t* s_rows, const uint32_t* s_counts, const float* s_dist, const uint32_t* s_ranks,
                                       float* scores, int64_t* rows, uint32_t* counts, float* dist, uint32_t* ranks);
hipError_t launch_gather_queries(hipStream_t st, const float* queries, const uint32_t* qmap,
                                 uint32_t n_slots, uint32_t dim, float* out);
hipError_t launch_scatter_results(hipStream_t st, const uint32_t* qmap, uint32_t n_slots, uint32_t k,
                                  const float* s_scores, const int64_t* s_rows,
                                  const uint32_t* s_counts, const float* s_dist,
                                  const uint32_t* s_ranks, float* scores, int64_t* rows,
                                  uint32_t* counts, float* dist, uint32_t* ranks);
hipError_t launch_synth_rows(hipStream_t st, uint64_t seed, uint64_t row0, uint64_t n_rows,
                             uint32_t dim, float* out);
hipError_t launch_synth_bytes(hipStream_t st, uint64_t seed, uint64_t blob_id0, uint64_t n_blobs,
                              uint64_t blob_len, uint8_t* out);

} // namespace yams_accel
