/*
 * yams_mi355x_accel.h — C ABI of libyams_mi355x_accel.so, the MI355X (gfx950) drop-in for the
 * YAMS vector-scan / SHA-256 / content-defined-chunking hot path.
 *
 * Two layers, both plain C (pointers + sizes, no C++ or torch types):
 *
 *   1. The flat `yams_accel_*` / `yams_scan_*` / `yams_sha256_*` / `yams_cdc_*` functions below.
 *      Data pointers marked "device" are HIP device addresses (e.g. a torch tensor's data_ptr());
 *      pointers marked "host" are ordinary memory.  All work is enqueued on the context's HIP
 *      stream; functions that hand results back to the host synchronise that stream themselves.
 *
 *   2. The YAMS plugin surface (include/yams/plugins/abi.h:26-34 in the reference): the eight
 *      `yams_plugin_*` entry points plus three vtables — `vector_scan_v1`, `content_hash_v1`,
 *      `chunker_v1` — written to the conventions of the reference's
 *      include/yams/plugins/model_provider_v1.h:44-49 (first field abi_version, second `self`,
 *      every function returns yams_status_t, plugin-allocated buffers are released only through
 *      the paired free_* of the same vtable).  The reference has no plugin interface for these
 *      three seams today (its seams are the C++ classes IVectorBackend / IContentHasher /
 *      IChunker); INTEGRATION.md shows the host-side adapter a maintainer adds.
 *
 * What each entry point replaces in the reference (paths under /root/reference):
 *   yams_scan_topk_*          SqliteVecBackend::Impl::bruteForceSearchUnlocked fast path,
 *                             src/vector/sqlite_vec_backend.cpp:4115-4135,4204-4331, batched as
 *                             searchSimilarBatch (:1612-1647), and for YAMS_SCAN_L2 the vec0 path
 *                             vec0SearchUnlocked (:4450-4530).
 *   yams_sha256_*             crypto::SHA256Hasher::hash / init / update / finalize,
 *                             src/crypto/sha256_hasher.cpp:81-109,167-195.
 *   yams_cdc_chunk_*          RabinChunker::chunkDataLazy (src/chunking/rabin_chunker.cpp:63-152)
 *                             and StreamingChunker::chunkData (include/yams/chunking/
 *                             streaming_chunker.h:146-204, src/chunking/streaming_chunker.cpp:37-137).
 *   yams_ingest_*             the hash + chunk_file phases of ContentStore::store,
 *                             src/api/content_store_impl.cpp:199-231 (caller contract only).
 *
 * There is NO CPU fallback behind this ABI: without a gfx950 device every compute entry point
 * returns YAMS_ERR_UNSUPPORTED.
 */
#ifndef YAMS_MI355X_ACCEL_H
#define YAMS_MI355X_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__) || defined(__clang__)
#define YAMS_ACCEL_API __attribute__((visibility("default")))
#else
#define YAMS_ACCEL_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: numerically identical to yams_status_e, model_provider_v1.h:17-25. */
typedef int yams_status_t;
#ifndef YAMS_PLUGINS_MODEL_PROVIDER_V1_H
enum yams_status_e {
    YAMS_OK = 0,
    YAMS_ERR_INVALID_ARG = 1,
    YAMS_ERR_NOT_FOUND = 2,
    YAMS_ERR_IO = 3,
    YAMS_ERR_INTERNAL = 4,
    YAMS_ERR_UNSUPPORTED = 5,
    YAMS_ERR_TIMEOUT = 6,            /* a sharded batch missed its deadline: the handle is stuck (see yams_scan_sharded_wait) */
    YAMS_ERR_RESOURCE_EXHAUSTED = 7  /* device (or pinned host) memory could not be had; the object is left as it was */
};
#endif

#define YAMS_ACCEL_VERSION_STRING "0.6.0" /* round 6: yams_scan_pq_topk_device, vector_scan_v1 v2 (pq_index_set / search_pq), yams_accel_trim, strict configuration */

/* ------------------------------------------------------------------------------------------ */
/* Context                                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct yams_accel_ctx yams_accel_ctx;

/* Number of visible HIP devices (0 when there is no GPU or no driver). */
YAMS_ACCEL_API int yams_accel_device_count(void);
/* Create a context bound to `device`.  `hip_stream` may be NULL (the context creates its own
 * non-blocking stream) or an existing hipStream_t on which all work of this context is enqueued.
 * NOTE: the handle of HIP's legacy default stream IS the null pointer — a caller that passes it
 * (torch.cuda.current_stream().cuda_stream is 0 unless a torch.cuda.Stream is current) gets the
 * context's own stream, and work it queued on the default stream (tensors still being filled) is
 * NOT ordered with the context's: synchronise first, or pass a created stream.  A context is
 * single-threaded; create one per host thread (mirrors "one SHA256Hasher per thread",
 * src/crypto/sha256_hasher.cpp:34). */
YAMS_ACCEL_API yams_status_t yams_accel_ctx_create(int device, void* hip_stream,
                                                   yams_accel_ctx** out_ctx);
YAMS_ACCEL_API void yams_accel_ctx_destroy(yams_accel_ctx* ctx);
YAMS_ACCEL_API yams_status_t yams_accel_ctx_synchronize(yams_accel_ctx* ctx);
/* Sweep gate: contexts of ONE device that serve concurrent search calls share a gate, and their big filter
 * sweeps then run one after the other on the GPU (stream-ordered: the host never waits) while everything
 * around them — query preparation, candidate selection, the fp64 re-score — overlaps the other context's
 * sweep.  Without a gate two concurrent sweeps interleave workgroup by workgroup and evict each other's
 * tiles from L2.  The plugin's pool of search contexts shares one gate per device (plugin.cpp); the
 * reference serves concurrent searches under a shared lock (vector_database.cpp:539,618).
 * A gate must outlive the contexts attached to it; attach with NULL to detach. */
typedef struct yams_accel_gate yams_accel_gate;
YAMS_ACCEL_API yams_status_t yams_accel_gate_create(int device, yams_accel_gate** out_gate);
YAMS_ACCEL_API void yams_accel_gate_destroy(yams_accel_gate* gate);
YAMS_ACCEL_API yams_status_t yams_accel_ctx_set_gate(yams_accel_ctx* ctx, yams_accel_gate* gate);
/* Sweep hold — for callers that put a COLLECTIVE behind every batch themselves (one process per GPU:
 * yams_scan_topk_device, then an RCCL all-gather of the record on a side stream, then the merge).  The filter sweep is
 * a persistent grid that owns every CU of the device; a collective kernel enqueued while another context's sweep runs
 * would wait for a CU under it and keep its peers on the other GPUs spinning meanwhile.  With the hold on, the gate
 * stays CLOSED behind this context's sweeps: no other context of the gate starts a sweep until this one calls
 * yams_accel_ctx_release_sweep_hold(ctx, stream) — after it has enqueued its collective (+ merge) on `stream`; the next
 * sweep then starts behind that work.  (yams_scan_sharded_* does the same for its own lanes: YAMS_SHARDED_FENCE_AUTO.)
 * Release on every path, also after a failed scan; releasing without a hold is a no-op. */
YAMS_ACCEL_API yams_status_t yams_accel_ctx_set_sweep_hold(yams_accel_ctx* ctx, int on);
YAMS_ACCEL_API yams_status_t yams_accel_ctx_release_sweep_hold(yams_accel_ctx* ctx, void* hip_stream);
/* Human-readable description of the last failure on this context (static storage, never NULL). */
YAMS_ACCEL_API const char* yams_accel_last_error(const yams_accel_ctx* ctx);
/* Device properties as a JSON string (malloc'd; release with yams_accel_free_string). */
YAMS_ACCEL_API yams_status_t yams_accel_device_info_json(yams_accel_ctx* ctx, char** out_json);
YAMS_ACCEL_API void yams_accel_free_string(char* s);
/* The library keeps the call-sized device buffers of its host-streaming entry points (yams_ingest_host: up to four of 8 GiB)
 * in a process-wide pool between calls — allocating them anew cost every call hundreds of milliseconds — up to 40 GiB per
 * process; the pool is emptied when one of the library's own allocations fails, and by this call.  device < 0: every device.
 * Returns the bytes handed back to the driver. */
YAMS_ACCEL_API uint64_t yams_accel_trim(int device);
/* Plain device memory helpers for hosts that do not bring their own allocator. */
YAMS_ACCEL_API yams_status_t yams_accel_malloc(yams_accel_ctx* ctx, size_t bytes, void** out_dev);
YAMS_ACCEL_API void yams_accel_free(yams_accel_ctx* ctx, void* dev);
YAMS_ACCEL_API yams_status_t yams_accel_upload(yams_accel_ctx* ctx, void* dst_dev,
                                               const void* src_host, size_t bytes);
YAMS_ACCEL_API yams_status_t yams_accel_download(yams_accel_ctx* ctx, void* dst_host,
                                                 const void* src_dev, size_t bytes);
/* HIP-event timing of the most recent call's dominant kernel on the context's stream:
 * average milliseconds per launch and number of launches (bench.py's roofline leg). */
YAMS_ACCEL_API yams_status_t yams_accel_last_kernel_ms(const yams_accel_ctx* ctx,
                                                       const char* kernel, double* out_ms_per_launch,
                                                       uint64_t* out_launches);
YAMS_ACCEL_API yams_status_t yams_accel_enable_kernel_timing(yams_accel_ctx* ctx, int enable);

/* ------------------------------------------------------------------------------------------ */
/* Exact vector scan (batched cosine / L2 top-k)                                                */
/* ------------------------------------------------------------------------------------------ */
typedef enum yams_scan_metric_e {
    YAMS_SCAN_COSINE = 0, /* exact_scan engine, sqlite_vec_backend.cpp:4204-4331 */
    YAMS_SCAN_L2 = 1      /* vec0_l2 engine,    sqlite_vec_backend.cpp:4450-4530 */
} yams_scan_metric_t;

/* A dense device mirror of `vectors WHERE embedding_dim = dim ORDER BY rowid`
 * (sqlite_vec_backend.cpp:4140-4175): row-major fp32, raw (not normalised). */
typedef struct yams_scan_corpus_s {
    const float* rows;         /* device, [n_rows][dim], 16-byte aligned                          */
    uint64_t n_rows;           /* < 2^32 per shard                                                */
    uint32_t dim;              /* embedding dimension                                             */
    uint32_t reserved;
    const uint32_t* tie_rank;  /* device, nullable: tie_rank[row] = rank of the row's chunk_id in
                                  lexicographic order (the reference's secondary sort key,
                                  :4218-4223).  NULL means rank == row_base + row.               */
    const uint32_t* rank_row;  /* device, nullable iff tie_rank is: inverse permutation within this
                                  shard for single-shard corpora (rank_row[tie_rank[r]] == r)     */
    int64_t row_base;          /* added to local row ordinals in the outputs (shard base)         */
    const uint32_t* row_mask;  /* device, nullable: bit (r & 31) of word (r >> 5) set = row r takes
                                  part in the search.  This is the `AND document_hash = ?` /
                                  `AND document_hash IN (...)` restriction of the reference's scan
                                  (sqlite_vec_backend.cpp:4137-4175): masked-out rows are neither
                                  visited nor evaluated.  ceil(n_rows / 32) words.               */
    uint64_t row_mask_count;   /* number of set bits (the host built the mask, it knows)          */
    const uint16_t* rows_bf16; /* device, nullable: the SHADOW of `rows` built by
                                  yams_scan_build_shadow_device — [n_rows][dim] bf16 (round to
                                  nearest even) of the unit-normalised rows, 16-byte aligned.  Only the MFMA filter reads it
                                  (half the bytes, no conversion in the loop); the fp64 re-score
                                  that decides the result always reads `rows`.  Results are
                                  bit-identical with and without it.                              */
    const float* rows_nsq;     /* device, nullable iff rows_bf16 is: [n_rows] fp32 squared norms  */
    const int8_t* rows_i8;     /* device, nullable: the INT8 SHADOW built by
                                  yams_scan_build_shadow_i8_device — YAMS_SCAN_I8_SHADOW_BYTES(n_rows,
                                  dim) bytes of int8 = round(unit-normalised row / s_b), in the
                                  blocked layout that call writes (opaque to callers), 16-byte
                                  aligned, dim % 64 == 0, dim >= 256.
                                  Read by the first filter tier of cosine searches (half the bytes
                                  of the bf16 shadow, twice its matrix rate); like every filter
                                  tier it only proposes candidates, the fp64 re-score over `rows`
                                  decides: results are bit-identical with and without it.
                                  L2 searches take the same tier when the view also carries rows_nsq
                                  (i.e. both shadows), the row norms of the shard are within a factor
                                  of two of each other and at most 64 rows have a squared norm outside
                                  (1e-30, 1e30) (zero / overflowing / non-finite rows: carried along as
                                  unconditional candidates); checked per call, otherwise L2 stays on
                                  the bf16 tier.                                                      */
    const float* rows_i8_meta; /* device, nullable iff rows_i8 is: [ceil(n_rows / 64)][2] =
                                  {s_b, e_b} per block of 64 rows: the block's quantisation scale
                                  and the largest measured residue |x/|x| - s_b * int8 row| of
                                  its rows                                                         */
    uint32_t stripe_rows;      /* 0: a contiguous shard, global id = row_base + local row.  Else the
                                  corpus is dealt to n_stripes shards in stripes of stripe_rows
                                  rows (how a growing mirror stays balanced over devices): local
                                  row L of this shard is global row  row_base +
                                  ((L / stripe_rows) * n_stripes + stripe_index) * stripe_rows +
                                  L % stripe_rows.  Local order == global order within a shard,
                                  so the default tie-break (row id) stays the reference's.        */
    uint32_t n_stripes;
    uint32_t stripe_index;
    uint32_t i8_flags;         /* YAMS_SCAN_I8_*: the layout rows_i8 was built in — what
                                  yams_scan_build_shadow_i8_layout_device was given (0 for
                                  yams_scan_build_shadow_i8_device).  (was: reserved2, 0)            */
} yams_scan_corpus_t;

/* int8 shadow layouts.  ROTATED: rows (and, per batch, queries) go through a fixed orthogonal map — sign flips and two
 * overlapping Walsh-Hadamard transforms — before they are quantised: dot products and norms stay where they were, the
 * energy of a few large components (outlier dimensions of an embedding model, a power-law spectrum) spreads over all of
 * them and the measured residue — the width of the filter's bound — drops from ~0.05 to ~0.01.  Rows with uniform
 * components quantise better WITHOUT it; yams_scan_choose_i8_layout_device measures both on a sample.  256 <= dim <= 4096.
 * Like every filter property it changes candidate counts, never results. */
#define YAMS_SCAN_I8_ROTATED 1u

#define YAMS_SCAN_FLAG_DEFER_THRESHOLD 1u /* L2 only: do not apply similarity_threshold (a sharded
                                             caller applies it after merging per-shard lists)  */
#define YAMS_SCAN_FLAG_FORCE_EXACT 2u     /* skip the MFMA filter, score every row in fp64       */
#define YAMS_SCAN_FLAG_F32_FILTER 4u      /* use the exact-f32 MFMA filter instead of the bf16 ones */
#define YAMS_SCAN_FLAG_SPLIT_FILTER 8u    /* start with the split-bf16 (3-pass) filter instead of
                                             the single-pass bf16 one                            */
#define YAMS_SCAN_FLAG_RECORD_PATH 16u    /* cosine: the reference's record path, taken when a search
                                             carries metadata_filters (sqlite_vec_backend.cpp:
                                             4333-4409).  Same fp64 arithmetic
                                             (computeCosineSimilarity), but rows are dropped when
                                             norm^2 < 1e-10 (isZeroNormEmbedding, :204-211) instead of
                                             <= 1e-12 (:4267-4269).  The caller turns the metadata
                                             predicate into the row allow-mask.                  */
#define YAMS_SCAN_FLAG_WIDE_TILE 32u      /* keep the 256-query MFMA tile for batches of <= 128
                                             queries (default: the narrow, HBM-bound kernel form);
                                             results are identical, only the kernel form differs  */
#define YAMS_SCAN_FLAG_NO_I8_FILTER 64u   /* do not use the int8 shadow even when the view carries one */
#define YAMS_SCAN_FLAG_RESIDENT_QUERIES 128u /* int8 tier: take the resident-query kernel form whenever its
                                             preconditions hold (dim % 128 == 0, dim <= 768), also on shards
                                             too small for it to balance (default: the library chooses);
                                             with YAMS_SCAN_FLAG_WIDE_TILE: never take it.  Results are
                                             identical, only the kernel form differs              */
/* L2 only — the arithmetic of vec0's distance.  It lives in the ABSENT third_party/sqlite-vec-cpp, so it cannot be
 * pinned from the reference checkout; every definition that dependency can plausibly have is served and
 * the HOST picks the one its build of the library uses (plugin config "l2_accumulate"):
 *   F64    (default) sum (x_i - q_i)^2 in fp64, sequentially, sqrt, round to fp32 — this repository's own definition;
 *   F32    fp32 difference, fp32 product, fp32 sequential sum, sqrtf — the public sqlite-vec's scalar loop;
 *   F32X8  / F32X16  the same with 8 / 16 round-robin partial sums (element i goes to lane i % 8 / 16) added left to
 *          right at the end — its AVX / AVX-512 forms;
 *   ... | FUSED  the three fp32 forms with the square accumulated by one fused multiply-add — those loops as a compiler
 *          emits them under -mfma, the flag the reference's build gives that dependency on x86.
 * include/yams_accel/l2_calibration.hpp finds out which one a host's build uses by asking its own distance function.
 * Each is tested bit for bit against a CPU restatement of that arithmetic (tests/): the order (distance
 * asc, chunk_id asc), the cosine re-score and the threshold-after-top-k of sqlite_vec_backend.cpp:4464-4512 are
 * unchanged.  The filter tiers are the same; the completeness proof widens its margin by the fp32 summation bound. */
#define YAMS_SCAN_FLAG_L2_ACC_F64 0u
#define YAMS_SCAN_FLAG_L2_ACC_F32 256u
#define YAMS_SCAN_FLAG_L2_ACC_F32X8 512u
#define YAMS_SCAN_FLAG_L2_ACC_F32X16 768u
#define YAMS_SCAN_FLAG_L2_ACC_MASK 768u
#define YAMS_SCAN_FLAG_L2_ACC_FUSED 2048u  /* with F32 / F32X8 / F32X16: every square is accumulated by ONE fused multiply-add,
                                                p = fma(d, d, p) with d = fl(x - q) — what `sum += d * d` and
                                                _mm256_add_ps(sum, _mm256_mul_ps(d, d)) compile to under -mfma, which is how the
                                                reference builds sqlite-vec-cpp on x86 (src/vector/meson.build:80-88).  Ignored with F64. */
#define YAMS_SCAN_FLAG_L2_ACC_EXPLICIT 1024u /* vtable callers: the L2_ACC bits of THIS call are the caller's decision even when
                                                they read F64 (= 0) — the plugin's configured default is not applied.  Set by the
                                                adapters after l2_calibration.hpp has asked the host's own distance function. */
#define YAMS_SCAN_MAX_K 1024u  /* results per query and call; larger k: rounds behind the allow-mask, as
                                  AccelVectorIndex::searchPeeled does (include/yams_accel/vector_index.hpp) */
#define YAMS_SCAN_MAX_DIM 8192u /* the fp64 re-score stages a query and its candidate rows in LDS */

typedef struct yams_scan_params_s {
    uint32_t k;                 /* results per query; 0 => empty result (:4123-4126)             */
    float similarity_threshold; /* rows with similarity < threshold are dropped (:4277-4279)     */
    uint32_t metric;            /* yams_scan_metric_t                                            */
    uint32_t flags;
} yams_scan_params_t;

/* Per-call work counters: the VectorSearchDiagnostics subset the exact scan fills
 * (include/yams/vector/vector_types.h:181-204; set at sqlite_vec_backend.cpp:4131-4135,
 * 4229-4251,4327-4329) plus this implementation's own filter statistics. */
typedef struct yams_scan_diag_s {
    uint32_t used_exact_scan;             /* always 1                                            */
    uint32_t rows_visited_observed;       /* always 1                                            */
    uint64_t rows_visited;                /* per query: n_rows (summed over the batch)           */
    uint64_t exact_distance_evaluations;  /* per query: n_rows (summed over the batch)           */
    uint64_t returned_rows;
    uint64_t filter_candidates;           /* rows that passed the MFMA filter (any tier)           */
    uint64_t rescored_rows;               /* rows re-scored in fp64                              */
    uint32_t widened_queries;             /* queries whose candidate set had to be widened       */
    uint32_t exact_fallback_queries;      /* queries that took the full fp64 scan                */
    uint32_t path;                        /* 0 = mfma filter + fp64 re-score, 1 = full fp64 scan */
    uint32_t escalated_queries;           /* queries re-filtered with the split (3-pass) filter  */
    uint32_t filter_tier;                 /* first filter tier of the call: 0 none (fp64 scan),
                                             1 int8, 2 bf16, 3 split bf16, 4 f32                  */
    uint32_t retried_queries;             /* int8 tier: queries filtered a second time with the threshold their proof asked for
                                             (the k-th best exact score found), before any escalation (was: reserved)  */
} yams_scan_diag_t;

/* Builds the filter shadow of `n_rows` rows (call it when rows are uploaded or appended; pass
 * pointers offset to the first new row to extend an existing shadow).  One pass: reads 4*dim bytes
 * and writes 2*dim + 4 bytes per row.  dim must be a multiple of 4; rows 16-byte aligned. */
YAMS_ACCEL_API yams_status_t yams_scan_build_shadow_device(yams_accel_ctx* ctx, const float* rows,
                                                           uint64_t n_rows, uint32_t dim,
                                                           uint16_t* out_rows_bf16,
                                                           float* out_rows_nsq);

/* (Re)builds the INT8 shadow of the mirror at `rows` for every block of 64 rows that intersects
 * [first_row, first_row + n_rows) — call it when those rows were uploaded or appended (the rows of a
 * block share one quantisation scale, so an append that starts inside a block re-quantises that
 * block's earlier rows too; all pointers are the BASES of the mirror's arrays, not offset ones).
 * dim must be a multiple of 64 and at least 256, rows 16-byte aligned.  For a mirror of n rows:
 * out_rows_i8 holds YAMS_SCAN_I8_SHADOW_BYTES(n, dim) bytes — the shadow is padded to whole blocks of 64 rows
 * and stored in the order the filter's DMA reads it ([row / 16][dim / 64][16 rows][64 bytes], the 16-byte
 * chunks of a row permuted for conflict-free LDS reads): opaque, build it with this call only; out_meta:
 * [ceil(n / 64)][2] fp32.  out_mean_err (host, nullable): mean residue bound
 * over the rebuilt blocks — a host that sees a large value (say > 0.02: heavy-tailed rows quantise
 * badly) may leave the int8 shadow out of the view and keep the bf16 one; asking for it
 * synchronises the stream. */
#define YAMS_SCAN_I8_SHADOW_ROWS(n_rows) ((((uint64_t)(n_rows)) + 63u) / 64u * 64u)
#define YAMS_SCAN_I8_SHADOW_BYTES(n_rows, dim) (YAMS_SCAN_I8_SHADOW_ROWS(n_rows) * (uint64_t)(dim))
YAMS_ACCEL_API yams_status_t yams_scan_build_shadow_i8_device(yams_accel_ctx* ctx, const float* rows,
                                                              uint64_t first_row, uint64_t n_rows,
                                                              uint32_t dim, int8_t* out_rows_i8,
                                                              float* out_meta, double* out_mean_err);

/* The same with the layout named (YAMS_SCAN_I8_* bits; the view must carry the same bits in i8_flags).  Blocks of one mirror
 * must all be built in one layout: a host that changes its mind rebuilds from row 0. */
YAMS_ACCEL_API yams_status_t yams_scan_build_shadow_i8_layout_device(yams_accel_ctx* ctx, const float* rows,
                                                                     uint64_t first_row, uint64_t n_rows,
                                                                     uint32_t dim, uint32_t i8_flags, int8_t* out_rows_i8,
                                                                     float* out_meta, double* out_mean_err);

/* Which layout quantises these rows better: the mean residue of up to 256 blocks of 64 rows spread over [0, n_rows) under
 * both layouts (host, nullable outputs; nothing on the device is written), *out_i8_flags = YAMS_SCAN_I8_ROTATED when the
 * rotated one is at least a fifth smaller, else 0.  Synchronises the stream.  A host decides once per mirror (first upload). */
YAMS_ACCEL_API yams_status_t yams_scan_choose_i8_layout_device(yams_accel_ctx* ctx, const float* rows, uint64_t n_rows,
                                                               uint32_t dim, uint32_t* out_i8_flags,
                                                               double* out_mean_err_plain, double* out_mean_err_rotated);

/* Batched exact top-k, everything device-resident.  Any batch size: more than 4096 queries run as
 * slices of 4096 (the per-batch workspace grows with the query count); diagnostics are summed.
 *   queries      device [n_queries][dim] fp32 (raw)
 *   out_scores   device [n_queries][k] fp32: cosine similarity (relevance_score, :4323-4326),
 *                best first; unused slots hold -inf
 *   out_rows     device [n_queries][k] int64: row_base + row ordinal; unused slots hold -1
 *   out_counts   device [n_queries] uint32
 *   out_dist     device [n_queries][k] fp32, nullable: L2 distance (YAMS_SCAN_L2 only)
 *   out_ranks    device [n_queries][k] uint32, nullable: tie rank of each hit = corpus.tie_rank[row]
 *                (the row ordinal without a rank table).  SHARD-LOCAL unless the shard's tie_rank table
 *                holds corpus-wide ranks: see yams_scan_merge_topk_device before merging by it
 * Returns YAMS_ERR_INVALID_ARG if any query is non-finite or has norm^2 < 1e-10 (:4127-4130;
 * a batch fails as a whole, :1635-1647), or on a dimension / alignment violation.
 * The call synchronises the context's stream before returning. */
YAMS_ACCEL_API yams_status_t yams_scan_topk_device(yams_accel_ctx* ctx,
                                                   const yams_scan_corpus_t* corpus,
                                                   const float* queries, uint32_t n_queries,
                                                   const yams_scan_params_t* params,
                                                   float* out_scores, int64_t* out_rows,
                                                   uint32_t* out_counts, float* out_dist,
                                                   uint32_t* out_ranks, yams_scan_diag_t* diag);

/* Same, with host-resident queries and outputs (the corpus stays a device mirror). */
YAMS_ACCEL_API yams_status_t yams_scan_topk_host(yams_accel_ctx* ctx,
                                                 const yams_scan_corpus_t* corpus,
                                                 const float* queries_host, uint32_t n_queries,
                                                 const yams_scan_params_t* params,
                                                 float* out_scores_host, int64_t* out_rows_host,
                                                 uint32_t* out_counts_host, float* out_dist_host,
                                                 yams_scan_diag_t* diag);

/* k-way merge of per-shard top-k lists (the step after an RCCL all-gather): inputs are
 * device [n_shards][n_queries][k] arrays laid out exactly as yams_scan_topk_device writes them
 * (ranks nullable => row ids break ties).  Order: similarity desc / distance asc, then rank asc.
 * in_ranks MUST be comparable ACROSS shards, i.e. positions in one corpus-wide chunk_id ordering
 * (:4218-4223) — ranks that each shard numbered on its own (a per-shard tie_rank table, or the row
 * ordinals a shard reports without one) order exact cross-shard ties wrongly.  Callers without a
 * corpus-wide ranking pass in_ranks = NULL (global row ids break ties: right whenever rows were
 * appended in chunk_id order) or use yams_scan_merge_records_device with rank_of_row.
 * For YAMS_SCAN_L2 the similarity_threshold is applied after the merge (:4508-4510). */
YAMS_ACCEL_API yams_status_t yams_scan_merge_topk_device(
    yams_accel_ctx* ctx, uint32_t n_shards, uint32_t n_queries, const yams_scan_params_t* params,
    const float* in_scores, const int64_t* in_rows, const uint32_t* in_counts,
    const float* in_dist, const uint32_t* in_ranks, float* out_scores, int64_t* out_rows,
    uint32_t* out_counts, float* out_dist);

/* The packed per-shard result record: what one shard's search writes and what travels between
 * devices (RCCL all-gather in the one-process-per-GPU form, peer copies in the in-process form):
 *   scores f32 [n_queries][k] | rows i64 [n_queries][k] | counts u32 [n_queries]
 *   (| dist f32 [n_queries][k]) (| ranks u32 [n_queries][k]),   every part 16-byte aligned.
 * An absent part has offset UINT64_MAX. */
typedef struct yams_scan_record_layout_s {
    uint64_t scores_off, rows_off, counts_off, dist_off, ranks_off, bytes;
} yams_scan_record_layout_t;
YAMS_ACCEL_API void yams_scan_record_layout(uint32_t n_queries, uint32_t k, int with_dist,
                                            int with_ranks, yams_scan_record_layout_t* out);

/* k-way merge of `n_shards` records lying `record_stride` bytes apart in device memory (e.g. the
 * output of one all-gather, untouched).  Order: similarity desc / distance asc, then the tie rank:
 * the records' own ranks if present, else rank_of_row[row - rank_row_base] if given (device; the
 * corpus-wide chunk_id ranking, :4218-4223), else the global row id. */
YAMS_ACCEL_API yams_status_t yams_scan_merge_records_device(
    yams_accel_ctx* ctx, uint32_t n_shards, uint32_t n_queries, const yams_scan_params_t* params,
    const void* records, uint64_t record_stride, const yams_scan_record_layout_t* layout,
    const uint32_t* rank_of_row, int64_t rank_row_base, float* out_scores, int64_t* out_rows,
    uint32_t* out_counts, float* out_dist);

/* One search over a corpus row-sharded across the devices of this node — what a host's
 * searchSimilarBatch (sqlite_vec_backend.cpp:1612-1647) calls when the corpus does not fit one GPU.
 * ONE process, one RCCL communicator over the shard devices (ncclCommInitAll; librccl.so.1 is bound
 * with dlopen when the first communicator is needed), per batch: every shard's exact top-k into a packed
 * record, ONE ncclAllGather of the records on a side stream, merge_topk_kernel on the first shard's
 * device, the merged result downloaded into pinned memory.  A persistent worker thread per (shard, lane)
 * drives its device; `lanes` batches are in flight: the upload, the query preparation and the sample pass of
 * batch i + 1 run under the sweep of batch i, and the exchange of batch i runs in the gap before the sweep of
 * batch i + 1 (see YAMS_SHARDED_FENCE_* below; submit / wait below; yams_scan_sharded_topk_host is submit + wait).
 * `devices[i]` is the HIP device of shard i.  Shards that SHARE a device cannot form a communicator (RCCL
 * refuses two ranks on one device): such handles — the parity tests of a one-GPU box — move their records
 * with device-to-device copies instead.  shards[i] is shard i's view (device pointers valid on devices[i];
 * row_base / stripe fields give global ids); upload and build shadows through yams_scan_sharded_ctx(s, i).
 * rank_of_row (nullable): device array on devices[0], the global tie ranking — needed only when
 * the shards carry tie_rank arrays (then each shard's tie_rank must be a local permutation that
 * preserves the global order). */
typedef struct yams_scan_sharded yams_scan_sharded;
#define YAMS_SHARDED_COLLECTIVE_AUTO 0u /* RCCL when there are >= 2 shards, each on its own device, and the
                                           communicator can be formed; else device-to-device copies of the
                                           records (one shard: nothing to exchange)                         */
#define YAMS_SHARDED_COLLECTIVE_RCCL 1u /* require the communicator — also for ONE shard (a communicator of one
                                           rank: all-gather + merge still run); creation fails with
                                           YAMS_ERR_UNSUPPORTED when the library cannot be loaded or initialised,
                                           YAMS_ERR_INVALID_ARG when shards share a device and the default
                                           library is used (RCCL refuses two ranks on one device; a library named
                                           in `rccl_library` decides for itself)                            */
#define YAMS_SHARDED_COLLECTIVE_PEER 2u /* never RCCL: device-to-device copies                               */
/* The filter sweep is a persistent grid that owns every CU of its device (one 160 KiB-LDS, 8 x 256-VGPR
 * workgroup per CU), and a collective is a CU-resident kernel that spins on its peers.  Left to float, the
 * all-gather of batch i on GPU g would have to find a CU under the sweep of batch i + 1 and then wait for GPU
 * h's copy of the kernel, itself queued behind h's sweep: the sweeps of different GPUs get coupled through the
 * few CUs the collective holds.  So with >= 2 shards the exchange is FENCED: on every shard the sweep of batch
 * i + 1 begins only after that shard's part of the exchange of batch i (on the root shard also the merge and the
 * download) has completed — the collective never shares a device with a sweep.  Everything in front of the sweep
 * (query upload, preparation, the sample pass) still overlaps — the sample pass in its half-tile form while a
 * fence (or a sweep hold) is installed: its resident-query form would be a second grid that owns every CU, and
 * the exchange of batch i may still be on the device when it starts.  YAMS_SHARDED_FENCE_OFF lifts the fence
 * (measurements only). */
#define YAMS_SHARDED_FENCE_AUTO 0u
#define YAMS_SHARDED_FENCE_OFF 1u
typedef struct yams_scan_sharded_options_s {
    uint32_t struct_size; /* sizeof(yams_scan_sharded_options_t); the 16-byte round-3 struct is accepted */
    uint32_t lanes;       /* batches in flight, 1..16; 0 = 2                                      */
    uint32_t collective;  /* YAMS_SHARDED_COLLECTIVE_*                                            */
    uint32_t fence;       /* YAMS_SHARDED_FENCE_*                                                 */
    const char* rccl_library; /* nullable: the collective library to dlopen instead of librccl.so.1 — any library
                                 that exports ncclGetVersion / ncclCommInitAll / ncclAllGather / ncclCommDestroy /
                                 ncclGetErrorString with RCCL's signatures (a site build of RCCL; the test suite's
                                 stream-ordered stand-in, which lets several ranks share one device).  The string
                                 is copied. */
    uint32_t exchange_timeout_ms; /* deadline of wait(): a batch whose scan + exchange has not completed by then makes
                                 wait() return YAMS_ERR_TIMEOUT with a diagnosis in ..._last_error, the communicator is
                                 aborted and the handle is STUCK (submit fails, destroy does not block).  0 = 30000;
                                 UINT32_MAX = wait for ever (the round-4 behaviour) */
    uint32_t reserved0;
} yams_scan_sharded_options_t;
YAMS_ACCEL_API yams_status_t yams_scan_sharded_create(const int* devices, uint32_t n_shards,
                                                      yams_scan_sharded** out);
YAMS_ACCEL_API yams_status_t yams_scan_sharded_create_ex(const int* devices, uint32_t n_shards,
                                                         const yams_scan_sharded_options_t* options,
                                                         yams_scan_sharded** out);
YAMS_ACCEL_API uint32_t yams_scan_sharded_lanes(const yams_scan_sharded* s);
/* {"shards":n,"lanes":l,"devices":[..],"collective":"rccl"|"peer_copy"|"none","fenced":true|false,"rccl_version":..,
 *  "rccl_library":"<path the symbols came from>","communicator_ranks":<ncclCommCount of the root's communicator>,
 *  "batches":..,"collectives":..,"exchanges_timed":..,"exchange_ms":<mean device time of all-gather + merge + download
 *  on the root shard's side stream, the wait for the slowest peer included>,"exchange_ms_max":..,
 *  "exchange_timeout_ms":..,"stuck":true|false}; malloc'd, release with yams_accel_free_string. */
YAMS_ACCEL_API yams_status_t yams_scan_sharded_info_json(yams_scan_sharded* s, char** out_json);
/* The context lane `lane` uses on shard `shard` (kernel timings of a pipelined run; the plugin's per-call
 * row masks live in its workspace).  yams_scan_sharded_ctx(s, i) is lane 0's. */
YAMS_ACCEL_API yams_accel_ctx* yams_scan_sharded_lane_ctx(yams_scan_sharded* s, uint32_t shard, uint32_t lane);
/* Pipelined use: acquire a lane (wait != 0: block until one is free; else YAMS_ERR_NOT_FOUND when every lane
 * has a batch in flight), submit a batch on it — the queries are copied into the lane's pinned staging, the
 * call returns at once —, submit the next batch on another lane, wait() for the first: it returns the merged
 * result and frees the lane.  The shard views, and rank_of_row, must stay valid until wait() returns.
 * A batch fails as a whole (:1635-1647); a failed batch still takes part in its collective, so the handle
 * stays usable.  Lanes are independent: different host threads may drive different lanes concurrently. */
#define YAMS_SHARDED_SUBMIT_DIAG 1u /* fill the diagnostics wait() returns (costs one more look at the result counts) */
YAMS_ACCEL_API yams_status_t yams_scan_sharded_lane_acquire(yams_scan_sharded* s, int wait, uint32_t* out_lane);
YAMS_ACCEL_API void yams_scan_sharded_lane_release(yams_scan_sharded* s, uint32_t lane); /* a lane acquired but not submitted */
YAMS_ACCEL_API yams_status_t yams_scan_sharded_submit(
    yams_scan_sharded* s, uint32_t lane, const yams_scan_corpus_t* shards, const float* queries_host,
    uint32_t n_queries, const yams_scan_params_t* params, const uint32_t* rank_of_row,
    int64_t rank_row_base, uint32_t flags);
YAMS_ACCEL_API yams_status_t yams_scan_sharded_wait(
    yams_scan_sharded* s, uint32_t lane, float* out_scores_host, int64_t* out_rows_host,
    uint32_t* out_counts_host, float* out_dist_host, yams_scan_diag_t* diag);
YAMS_ACCEL_API void yams_scan_sharded_destroy(yams_scan_sharded* s);
YAMS_ACCEL_API uint32_t yams_scan_sharded_count(const yams_scan_sharded* s);
YAMS_ACCEL_API yams_accel_ctx* yams_scan_sharded_ctx(yams_scan_sharded* s, uint32_t shard);
YAMS_ACCEL_API const char* yams_scan_sharded_last_error(const yams_scan_sharded* s);
YAMS_ACCEL_API yams_status_t yams_scan_sharded_topk_host(
    yams_scan_sharded* s, const yams_scan_corpus_t* shards, const float* queries_host,
    uint32_t n_queries, const yams_scan_params_t* params, const uint32_t* rank_of_row,
    int64_t rank_row_base, float* out_scores_host, int64_t* out_rows_host,
    uint32_t* out_counts_host, float* out_dist_host, yams_scan_diag_t* diag);

/* Allocation-failure injection, for tests of the out-of-memory paths (ErrorCode::ResourceExhausted,
 * include/yams/core/types.h:49 of the reference).  After yams_accel_debug_fail_alloc_after(n) the next n allocations of
 * device / pinned / VMM-backed memory THIS LIBRARY makes for its own objects (workspaces, mirrors behind corpus_append,
 * lane buffers, staging rings, digest sets) succeed and every later one fails as hipErrorOutOfMemory does, until
 * yams_accel_debug_fail_alloc_after(-1).  Process-wide; never armed unless called (the library reads no environment).
 * yams_accel_alloc — the caller's own buffers — is exempt.  ..._alloc_faults: allocations failed by injection so far.
 * What the library promises under exhaustion: the failing call returns YAMS_ERR_RESOURCE_EXHAUSTED, the object it
 * was growing is left as it was (a corpus keeps its rows and answers searches), nothing is leaked (health JSON:
 * "mirror_bytes_mapped" / "mirror_bytes_parked"), and the same call succeeds once memory is there again.
 * MEASUREMENT BUILD ONLY (libyams_mi355x_accel_measure.so, -DYAMS_ACCEL_MEASURE): in the product library the three entry points
 * exist — one ABI for both builds — and do nothing: no code in a host process can arm allocation failures there, and the
 * allocation paths carry no check.  yams_accel_debug_alloc_injection_compiled() says which build this is (1 / 0). */
YAMS_ACCEL_API void yams_accel_debug_fail_alloc_after(int64_t n);
YAMS_ACCEL_API uint64_t yams_accel_debug_alloc_faults(void);
YAMS_ACCEL_API int yams_accel_debug_alloc_injection_compiled(void);

/* Fill a device matrix with the synthetic embedding recipe: Philox4x32-10
 * (seed, row, col/4) -> U[-1,1) -> fp32 L2-normalise.  Used by bench.py and tests so that a
 * corpus larger than host RAM can be generated in HBM and any slice regenerated on the CPU. */
YAMS_ACCEL_API yams_status_t yams_synth_rows_device(yams_accel_ctx* ctx, uint64_t seed,
                                                    uint64_t row0, uint64_t n_rows, uint32_t dim,
                                                    float* out_dev);
YAMS_ACCEL_API yams_status_t yams_synth_bytes_device(yams_accel_ctx* ctx, uint64_t seed,
                                                     uint64_t blob_id0, uint64_t n_blobs,
                                                     uint64_t blob_len, uint8_t* out_dev);

/* ------------------------------------------------------------------------------------------ */
/* The product-quantised engine: ADC scan over host-supplied codes + exact re-rank (round 6)  */
/* ------------------------------------------------------------------------------------------ */
/* Replaces the scoring, selection and re-rank of SqliteVecBackend::Impl::simeonPqSearchUnlocked
 * (src/vector/sqlite_vec_backend.cpp:3868-4056; VectorSearchEngine::SimeonPqAdc is the product's DEFAULT engine,
 * include/yams/vector/vector_types.h:80).  The host keeps what third_party/simeon owns — training the product quantiser,
 * encoding rows (SimeonPqIndexState::codes, :53, :3648-3660) and building the per-query table of inner products
 * (simeon::PQInnerProductQuery, :3901: lut[j][c] = <sub-vector j of the normalised query, centroid c of sub-quantiser j>) —
 * and hands over the codes once (a device mirror, appended like rows) and the tables per batch.  Per query the device
 *   1. scores every indexed row, or every index of `candidates` (:3910-3937, sorted ascending by the host), with
 *      approxScore = sum over j of lut[j][codes[index * m + j]] in fp32 (:3965-3977; the ORDER of the additions is
 *      simeon's: YAMS_PQ_SUM_* below — PARITY UNPINNED, simeon is absent from the reference checkout);
 *   2. keeps the best approxK = min(candidates, max(k, k * rerank_factor)) by (score desc, tie key asc) (:3952-3997);
 *   3. re-scores them with VectorDatabase::computeCosineSimilarity(raw query, row) in fp64, the reference's summation
 *      order (:4023-4034), drops similarity < threshold (:4036-4038);
 *   4. sorts by (similarity desc, chunk_id asc) and cuts to k (:4041-4051).
 * `exact_fallback` indexes (:3882-3893) are the host's call to yams_scan_topk_device.  DocumentTopK selection
 * (retainBestRecordPerDocument) stays on the host: ask for k = the candidate count. */
typedef struct yams_scan_pq_index_s {
    const uint8_t* codes;       /* device [n_codes][m], 4-byte aligned: code of indexed row i = bytes [i * m, (i + 1) * m)          */
    uint64_t n_codes;           /* indexed rows (SimeonPqIndexState::rowids.size()), < 2^32                                        */
    uint32_t m;                 /* sub-quantisers = bytes per code (simeon_pq_subquantizers, default 32), <= 128; 256 centroids each */
    uint32_t reserved;
    const uint32_t* tie_rank;   /* device [n_codes], nullable: rank of tie_break_keys[i] (= stableStringKey(chunk_id), :3337) among
                                   all indexed rows — ascending key, equal keys by index — the second key of the comparator
                                   :3985-3990.  Null: index order.                                                                 */
    const uint32_t* key_row;    /* device [n_codes], nullable: the corpus row (yams_scan_corpus_t::rows) of the code whose KEY INDEX
                                   is r, where the key index of code i is tie_rank[i] (or i without a tie_rank table): the
                                   rowids[idx] lookup of :4009.  Null: identity.                                                    */
} yams_scan_pq_index_t;

#define YAMS_PQ_SUM_SEQUENTIAL 0u /* one fp32 sum over j = 0 .. m - 1 (a scalar loop)                                             */
#define YAMS_PQ_SUM_X4 1u         /* 4 partial sums (element j -> lane j % 4), lanes added left to right (SSE-shaped)            */
#define YAMS_PQ_SUM_X8 2u         /* 8 partial sums (AVX-shaped)                                                                  */
#define YAMS_PQ_SUM_X16 3u        /* 16 partial sums (AVX-512-shaped)                                                             */
#define YAMS_PQ_SUM_MASK 3u

typedef struct yams_scan_pq_params_s {
    uint32_t k;
    float similarity_threshold; /* on the EXACT similarity of the re-rank (:4036-4038)                                           */
    uint32_t rerank_factor;     /* SimeonPqIndexState::rerank_factor (>= 1; 0 is read as 1); k * rerank_factor <= 2047            */
    uint32_t flags;             /* YAMS_PQ_SUM_*                                                                                  */
} yams_scan_pq_params_t;

/* corpus: the fp32 rows the re-rank reads (+ tie_rank / rank_row: the chunk_id order of the final sort; row_base / stripes:
 * how a row becomes the id in out_rows).  queries: device [n_queries][dim] RAW queries (the re-rank's first argument);
 * luts: device [n_queries][m][256] fp32, 16-byte aligned; candidates: device, nullable — n_candidates ascending indices
 * into the PQ index (a restriction to documents, :3910-3930; an EMPTY list returns nothing, :3946-3948).
 * out_scores / out_rows: device [n_queries][k]; out_counts: device [n_queries].  A query whose norm^2 is <= 1e-20 or not
 * finite returns nothing (:3895-3898).  Synchronous.  diag (nullable): rows_visited = candidates per query summed,
 * exact_distance_evaluations = rows re-scored, path = 2, filter_tier = 5. */
YAMS_ACCEL_API yams_status_t yams_scan_pq_topk_device(yams_accel_ctx* ctx, const yams_scan_corpus_t* corpus,
                                                      const yams_scan_pq_index_t* pq, const float* queries, const float* luts,
                                                      uint32_t n_queries, const yams_scan_pq_params_t* params,
                                                      const uint32_t* candidates, uint64_t n_candidates, float* out_scores,
                                                      int64_t* out_rows, uint32_t* out_counts, yams_scan_diag_t* diag);

/* ------------------------------------------------------------------------------------------ */
/* SHA-256                                                                                      */
/* ------------------------------------------------------------------------------------------ */
/* Digest n_msgs byte ranges of one device buffer: message i = data[offsets[i] .. +lengths[i]).
 * offsets/lengths are DEVICE arrays; digests is device [n_msgs][32] (raw bytes, big-endian words
 * as FIPS 180-4 prints them).  Asynchronous on the context's stream. */
YAMS_ACCEL_API yams_status_t yams_sha256_batch_device(yams_accel_ctx* ctx, const uint8_t* data,
                                                      const uint64_t* offsets,
                                                      const uint64_t* lengths, uint64_t n_msgs,
                                                      uint8_t* digests);
/* Batched integrity check (ChunkValidator::validateChunks / validateManifest,
 * src/integrity/chunk_validator.cpp:36-120,160-212,230-262): re-hash n_chunks byte ranges of one
 * device buffer and compare with the expected raw 32-byte digests (device, any alignment).
 * out_valid[i] = 1 iff SHA-256(data[offsets[i] .. +lengths[i])) == expected_digests[i];
 * *out_n_invalid = number of mismatches.  Synchronises the context's stream. */
YAMS_ACCEL_API yams_status_t yams_verify_chunks_device(yams_accel_ctx* ctx, const uint8_t* data,
                                                       const uint64_t* offsets,
                                                       const uint64_t* lengths, uint64_t n_chunks,
                                                       const uint8_t* expected_digests,
                                                       uint8_t* out_valid, uint64_t* out_n_invalid);
/* One-shot over host memory: SHA256Hasher::hash(span), sha256_hasher.cpp:167-195.
 * out_hex receives 64 lower-case hex characters + NUL (bytesToHex, :19-30). */
YAMS_ACCEL_API yams_status_t yams_sha256_host(yams_accel_ctx* ctx, const uint8_t* data_host,
                                              size_t n, char out_hex[65]);
/* Many host buffers at once (one message per lane on the device). */
YAMS_ACCEL_API yams_status_t yams_sha256_many_host(yams_accel_ctx* ctx,
                                                   const uint8_t* const* msgs_host,
                                                   const size_t* lens, size_t n_msgs,
                                                   char* out_hex /* [n_msgs][65] */);

/* ------------------------------------------------------------------------------------------ */
/* Content-defined chunking                                                                     */
/* ------------------------------------------------------------------------------------------ */
typedef enum yams_cdc_mode_e {
    YAMS_CDC_RABIN = 0,     /* RabinChunker: first tested byte is start+min, sizes in [min+1,max] */
    YAMS_CDC_STREAMING = 1  /* StreamingChunker (product default): first tested byte is
                               start+min-1, sizes in [min,max]                                   */
} yams_cdc_mode_t;

/* ChunkingConfig, include/yams/chunking/chunker.h:44-51 (target size does not enter the
 * boundary logic and is omitted).  polynomial == 0 selects the default (rabin_chunker.cpp:29-37). */
typedef struct yams_cdc_config_s {
    uint64_t window_size; /* 1..48 (RabinWindow is a 48-byte ring, chunker.h:151-155)            */
    uint64_t min_size;
    uint64_t max_size;
    uint64_t polynomial;
    uint64_t mask;
    uint32_t mode;        /* yams_cdc_mode_t */
    uint32_t flags;       /* YAMS_CDC_FLAG_*; 0 = defaults */
} yams_cdc_config_t;
/* Boundary detection has two kernel forms with identical results: a narrow one for window 48 and
 * masks below 2^31 (the product defaults) and a generic one (any window 1..48, any 64-bit mask).
 * This flag selects the generic form even where the narrow one applies (parity tests use it). */
#define YAMS_CDC_FLAG_GENERIC_KERNEL 1u

YAMS_ACCEL_API void yams_cdc_default_config(yams_cdc_config_t* cfg, uint32_t mode);

/* Result of chunking (and optionally hashing) a set of blobs; device arrays owned by the
 * context's workspace, valid until the next ingest/cdc call on the same context. */
typedef struct yams_ingest_result_s {
    uint64_t n_chunks;            /* total over all blobs                                        */
    const uint64_t* chunk_offset; /* device [n_chunks]: offset within its blob                   */
    const uint64_t* chunk_size;   /* device [n_chunks]                                           */
    const uint32_t* chunk_blob;   /* device [n_chunks]: blob index                               */
    const uint64_t* blob_first;   /* device [n_blobs + 1]: first chunk of each blob (prefix)     */
    const uint8_t* chunk_digest;  /* device [n_chunks][32] or NULL                               */
    const uint8_t* blob_digest;   /* device [n_blobs][32] or NULL                                */
} yams_ingest_result_t;

/* Chunk boundaries only.  data: device bytes; blob_offsets/blob_lengths: HOST arrays locating
 * each blob inside `data`. */
YAMS_ACCEL_API yams_status_t yams_cdc_chunk_device(yams_accel_ctx* ctx, const uint8_t* data,
                                                   const uint64_t* blob_offsets_host,
                                                   const uint64_t* blob_lengths_host,
                                                   uint64_t n_blobs, const yams_cdc_config_t* cfg,
                                                   yams_ingest_result_t* out);
/* The ingest hot path: chunk boundaries + per-chunk SHA-256 + whole-blob SHA-256
 * (content_store_impl.cpp:199-231).  flags: bit0 = chunk digests, bit1 = blob digests. */
#define YAMS_INGEST_CHUNK_DIGESTS 1u
#define YAMS_INGEST_BLOB_DIGESTS 2u
/* With YAMS_INGEST_BLOB_DIGESTS: do NOT compute the whole-blob digest of blobs whose chain would outlast the rest
 * of the call.  One SHA-256 chain is sequential — ~35 MB/s on a device lane, > 1 GB/s on a host core with SHA
 * extensions — so ONE 64 MiB blob in a batch holds the call for 1.9 s while everything else finishes in tens of
 * milliseconds.  A blob is deferred iff its length exceeds the threshold below (a pure function of the call's total
 * bytes: the caller evaluates the same predicate, starts its own hasher — ContentStore::store's SHA256Hasher,
 * content_store_impl.cpp:199-231 — on those blobs BEFORE the call and joins after it); the digest entry of a
 * deferred blob is 32 zero bytes.  Chunk boundaries and per-chunk digests of deferred blobs are computed as usual.
 * Thresholds: chain time L / 35 MB/s against the rest of the call at its measured rate — device-resident input
 * ~340 GB/s (ratio 2^12, rounded towards keeping work on the device), host-streamed input ~25 GB/s (ratio 2^9);
 * never below the lone-chain limit of content_hash_v1 (1 MiB). */
#define YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS 4u
static inline uint64_t yams_ingest_defer_threshold_device(uint64_t total_bytes) {
    const uint64_t t = total_bytes >> 12;
    return t > (1ull << 20) ? t : (1ull << 20);
}
static inline uint64_t yams_ingest_defer_threshold_host(uint64_t total_bytes) {
    const uint64_t t = total_bytes >> 9;
    return t > (1ull << 20) ? t : (1ull << 20);
}
YAMS_ACCEL_API yams_status_t yams_ingest_device(yams_accel_ctx* ctx, const uint8_t* data,
                                                const uint64_t* blob_offsets_host,
                                                const uint64_t* blob_lengths_host,
                                                uint64_t n_blobs, const yams_cdc_config_t* cfg,
                                                uint32_t flags, yams_ingest_result_t* out);

/* The same path for blobs in HOST memory (what ContentStore::store has after reading a file,
 * content_store_impl.cpp:199-231): blobs cross PCIe in batches of ~batch_bytes (0 = chosen from the call: about
 * 2048 of its longest blob, 1 to 8 GiB, at least four batches — the digest chains of a batch take (longest blob) /
 * 35 MB/s and three batches' chains are in flight; a blob is never split; and never more than a tenth of the device
 * memory that is free when the call starts, so that four slot buffers + tables take at most half of it) through two to
 * four device buffers, batch i + 1 uploading while batch i is chunked and hashed.  Device footprint: during the call up
 * to 4 x batch_bytes + ~25 % of tables; AFTER the call the context keeps no buffer above 1.25 GiB (larger ones are
 * freed on return: they belong to the call, not to the context).
 * Pinned (page-locked) blob memory uploads at link speed, pageable memory through the runtime's
 * staging.  Results go to caller arrays: out_blob_first[n_blobs + 1] (prefix of chunk counts),
 * out_chunk_offset / out_chunk_size [chunk_cap], out_chunk_digest [chunk_cap][32] (nullable),
 * out_blob_digest [n_blobs][32] (required with YAMS_INGEST_BLOB_DIGESTS).  If the chunks do not fit
 * chunk_cap the status is YAMS_ERR_INVALID_ARG and *out_n_chunks holds the required size (blob digests
 * and out_blob_first are complete even then). */
YAMS_ACCEL_API yams_status_t yams_ingest_host(yams_accel_ctx* ctx, const uint8_t* const* blobs_host,
                                              const uint64_t* blob_lengths, uint64_t n_blobs,
                                              const yams_cdc_config_t* cfg, uint32_t flags,
                                              uint64_t batch_bytes, uint64_t* out_blob_first,
                                              uint64_t* out_chunk_offset, uint64_t* out_chunk_size,
                                              uint8_t* out_chunk_digest, uint64_t chunk_cap,
                                              uint8_t* out_blob_digest, uint64_t* out_n_chunks);

/* IChunker::chunkDataLazy over host memory (chunker.h:84-86): fills caller arrays; hex may be
 * NULL.  Returns the chunk count in *out_count (cap = capacity of the arrays; if the count
 * exceeds cap the status is YAMS_ERR_INVALID_ARG and *out_count holds the required size). */
YAMS_ACCEL_API yams_status_t yams_cdc_chunk_host(yams_accel_ctx* ctx, const uint8_t* data_host,
                                                 size_t n, const yams_cdc_config_t* cfg,
                                                 uint64_t* offsets, uint64_t* sizes,
                                                 char* hex /* [cap][65] */, size_t cap,
                                                 size_t* out_count);

/* One WINDOW of a stream (StreamingChunker::processStream / processFileStream, streaming_chunker.h:54-121: the
 * bounded-memory callback form).  The first context_len bytes of `data_host` are history only — they feed the
 * rolling hash (its state is a function of the last 56 bytes: 8 inside the 64-bit shift, 48..55 through the byte
 * that leaves the ring; >= 64 bytes of true history reproduce it exactly), the first chunk starts at offset
 * context_len; offsets are relative to data_host.  The LAST chunk returned ends at the end of the buffer whether or
 * not a boundary falls there: a caller that has more data carries that chunk (and 64 bytes in front of it) into the
 * next window.  Streaming mode only (RabinChunker restarts its hash at every chunk): YAMS_ERR_INVALID_ARG otherwise.
 * context_len = 0 is yams_cdc_chunk_host. */
YAMS_ACCEL_API yams_status_t yams_cdc_chunk_window_host(yams_accel_ctx* ctx, const uint8_t* data_host,
                                                        size_t n, size_t context_len, const yams_cdc_config_t* cfg,
                                                        uint64_t* offsets, uint64_t* sizes,
                                                        char* hex /* [cap][65] */, size_t cap,
                                                        size_t* out_count);

/* ------------------------------------------------------------------------------------------------
 * Chunk dedup lookup: a device-resident set of SHA-256 digests.
 *
 * Replaces the per-chunk `storage_->exists(chunk.hash)` round trips of ContentStore::store
 * (src/api/content_store_impl.cpp:246-287).  The reference walks the chunk list in order: a chunk
 * is stored iff its hash is neither in the store nor carried by an EARLIER chunk of the same walk;
 * yams_dedup_insert_* answers exactly that per chunk (is_new[i]; first occurrence = lowest index)
 * and adds the new digests to the set, so one call per ingest batch replaces n exists() calls.
 * Digests are raw 32-byte values (the `chunk_digest` array of yams_ingest_result_t), 8-byte
 * aligned.  The set grows by rehashing; calls synchronise the context's stream.
 * ---------------------------------------------------------------------------------------------- */
typedef struct yams_dedup_set yams_dedup_set;
YAMS_ACCEL_API yams_status_t yams_dedup_set_create(yams_accel_ctx* ctx, uint64_t expected_entries,
                                                   yams_dedup_set** out_set);
YAMS_ACCEL_API void yams_dedup_set_destroy(yams_dedup_set* set);
YAMS_ACCEL_API yams_status_t yams_dedup_set_size(const yams_dedup_set* set, uint64_t* out_entries);
/* chunk_sizes (device, nullable): when given, out_bytes_new / out_bytes_deduped receive the
 * bytesStored / bytesDeduped sums of content_store_impl.cpp:255,274. */
YAMS_ACCEL_API yams_status_t yams_dedup_insert_device(yams_dedup_set* set, const uint8_t* digests,
                                                      uint64_t n, const uint64_t* chunk_sizes,
                                                      uint8_t* out_is_new, uint64_t* out_n_new,
                                                      uint64_t* out_bytes_new,
                                                      uint64_t* out_bytes_deduped);
YAMS_ACCEL_API yams_status_t yams_dedup_probe_device(yams_dedup_set* set, const uint8_t* digests,
                                                     uint64_t n, uint8_t* out_exists);
YAMS_ACCEL_API yams_status_t yams_dedup_insert_host(yams_dedup_set* set, const uint8_t* digests_host,
                                                    uint64_t n, uint8_t* out_is_new_host,
                                                    uint64_t* out_n_new);
YAMS_ACCEL_API yams_status_t yams_dedup_probe_host(yams_dedup_set* set, const uint8_t* digests_host,
                                                   uint64_t n, uint8_t* out_exists_host);

/* ------------------------------------------------------------------------------------------ */
/* Plugin vtables (obtained through yams_plugin_get_interface)                                  */
/* ------------------------------------------------------------------------------------------ */
#define YAMS_IFACE_VECTOR_SCAN_V1 "vector_scan_v1"
#define YAMS_IFACE_VECTOR_SCAN_V1_VERSION 2u /* 2 appends pq_index_set / search_pq; a host that asks for version 1 gets the same table */
#define YAMS_IFACE_CONTENT_HASH_V1 "content_hash_v1"
#define YAMS_IFACE_CONTENT_HASH_V1_VERSION 1u
#define YAMS_IFACE_CHUNKER_V1 "chunker_v1"
#define YAMS_IFACE_CHUNKER_V1_VERSION 3u /* 2: chunk_many / free_chunk_batch appended; 3: chunk_window appended (hosts
                                           that asked for an older version see the same leading fields) */

/* The YAMS plugin entry points (include/yams/plugins/abi.h:18-34 in the reference; the declarations
 * are identical, so this header and the reference's can be included together). */
#ifndef YAMS_PLUGIN_ABI_VERSION
#define YAMS_PLUGIN_ABI_VERSION 1
#define YAMS_PLUGIN_OK 0
#define YAMS_PLUGIN_ERR_INCOMPATIBLE -1
#define YAMS_PLUGIN_ERR_NOT_FOUND -2
#define YAMS_PLUGIN_ERR_INIT_FAILED -3
#define YAMS_PLUGIN_ERR_INVALID -4
#endif
YAMS_ACCEL_API int yams_plugin_get_abi_version(void);
YAMS_ACCEL_API const char* yams_plugin_get_name(void);
YAMS_ACCEL_API const char* yams_plugin_get_version(void);
YAMS_ACCEL_API const char* yams_plugin_get_manifest_json(void);
/* config_json: {"device": n} | {"devices": [..]} (a corpus is dealt to all of them in stripes and
 * searched behind one call), "search_slots": n (concurrent searches, default 2),
 * "shadows": "both" | "bf16" | "i8" | "none". */
YAMS_ACCEL_API int yams_plugin_init(const char* config_json, const void* host_context);
YAMS_ACCEL_API void yams_plugin_shutdown(void);
YAMS_ACCEL_API int yams_plugin_get_interface(const char* iface_id, uint32_t version, void** out_iface);
YAMS_ACCEL_API int yams_plugin_get_health_json(char** out_json);

typedef struct yams_scan_hit_s {
    int64_t row;      /* row ordinal in the mirror (host maps it to its VectorRecord)            */
    float similarity; /* relevance_score */
    float distance;   /* L2 distance for YAMS_SCAN_L2, else 1 - similarity
                         (utils::similarityToDistance, vector_database.cpp:1867-1875)           */
} yams_scan_hit_t;

typedef struct yams_vector_scan_v1 {
    uint32_t abi_version; /* YAMS_IFACE_VECTOR_SCAN_V1_VERSION */
    void* self;
    /* Device mirror lifecycle.  `rows` is host memory [n_rows][dim]; tie_ranks nullable. */
    yams_status_t (*corpus_create)(void* self, uint32_t dim, uint64_t* out_corpus_id);
    yams_status_t (*corpus_append)(void* self, uint64_t corpus_id, const float* rows,
                                   uint64_t n_rows);
    yams_status_t (*corpus_set_tie_ranks)(void* self, uint64_t corpus_id,
                                          const uint32_t* tie_ranks, uint64_t n_rows);
    yams_status_t (*corpus_clear)(void* self, uint64_t corpus_id);
    yams_status_t (*corpus_destroy)(void* self, uint64_t corpus_id);
    yams_status_t (*corpus_size)(void* self, uint64_t corpus_id, uint64_t* out_rows,
                                 uint32_t* out_dim);
    /* Batched search: one contiguous row-major result buffer [n_queries][k] (padded; counts say
     * how many entries of each row are valid), released with free_hits. */
    yams_status_t (*search_batch)(void* self, uint64_t corpus_id, const float* queries,
                                  uint32_t n_queries, uint32_t dim, uint32_t k,
                                  float similarity_threshold, uint32_t metric,
                                  yams_scan_hit_t** out_hits, uint32_t** out_counts,
                                  yams_scan_diag_t* out_diag /* nullable */);
    void (*free_hits)(void* self, yams_scan_hit_t* hits, uint32_t* counts);
    yams_status_t (*get_runtime_info_json)(void* self, char** out_json);
    void (*free_string)(void* self, char* s);
    /* Filtered search (document_hash / candidate_hashes, vector_store.h:44-49): `row_mask` is a
     * HOST bitmap over the mirror's rows (bit r & 31 of word r >> 5), NULL = all rows. */
    yams_status_t (*search_batch_masked)(void* self, uint64_t corpus_id, const float* queries,
                                         uint32_t n_queries, uint32_t dim, uint32_t k,
                                         float similarity_threshold, uint32_t metric,
                                         const uint32_t* row_mask, yams_scan_hit_t** out_hits,
                                         uint32_t** out_counts, yams_scan_diag_t* out_diag);
    /* Same, with YAMS_SCAN_FLAG_* bits (e.g. YAMS_SCAN_FLAG_RECORD_PATH for searches that carry
     * metadata_filters, whose predicate the host has already folded into `row_mask`). */
    yams_status_t (*search_batch_ex)(void* self, uint64_t corpus_id, const float* queries,
                                     uint32_t n_queries, uint32_t dim, uint32_t k,
                                     float similarity_threshold, uint32_t metric, uint32_t flags,
                                     const uint32_t* row_mask, yams_scan_hit_t** out_hits,
                                     uint32_t** out_counts, yams_scan_diag_t* out_diag);
    /* ---- version 2: the product-quantised engine (VectorSearchEngine::SimeonPqAdc; simeonPqSearchUnlocked,
     * src/vector/sqlite_vec_backend.cpp:3868-4056) over the same mirror; see yams_scan_pq_topk_device.
     * pq_index_set: the host's SimeonPqIndexState (:48-62) for this corpus, all HOST arrays: codes [n_codes][m],
     * tie_keys [n_codes] (tie_break_keys = stableStringKey(chunk_id)), row_of_index [n_codes] (nullable = identity: which
     * mirror row code i belongs to — rowids[i] mapped to the mirror's ordinal).  Replaces a previous index; n_codes == 0
     * drops it.  corpus_clear / corpus_append do NOT touch it: the host sets it again after it re-encodes (a PQ index that
     * names rows the mirror has lost skips them, :4010-4012).  Corpora dealt to several devices: YAMS_ERR_UNSUPPORTED. */
    yams_status_t (*pq_index_set)(void* self, uint64_t corpus_id, const uint8_t* codes, uint64_t n_codes, uint32_t m,
                                  const uint64_t* tie_keys, const uint32_t* row_of_index);
    /* queries: host [n_queries][dim] RAW queries; luts: host [n_queries][m][256] (what simeon::PQInnerProductQuery holds for
     * the NORMALISED query, :3895-3901); candidates (nullable): n_candidates ascending indices into the PQ index (:3910-3937);
     * flags: YAMS_PQ_SUM_*.  Hits as search_batch (distance = 1 - similarity); the final order uses the corpus's tie ranks
     * (corpus_set_tie_ranks: chunk_id order, :4041-4051). */
    yams_status_t (*search_pq)(void* self, uint64_t corpus_id, const float* queries, const float* luts, uint32_t n_queries,
                               uint32_t dim, uint32_t k, float similarity_threshold, uint32_t rerank_factor, uint32_t flags,
                               const uint32_t* candidates, uint64_t n_candidates, yams_scan_hit_t** out_hits,
                               uint32_t** out_counts, yams_scan_diag_t* out_diag);
} yams_vector_scan_v1;

/* WHAT THE DEVICE IS WORSE AT IS REFUSED, NOT SERVED SLOWLY.  SHA-256 of one message is one sequential chain: a
 * GPU lane advances it at ~35 MB/s, a host core with SHA-NI at > 1 GB/s.  The device wins only with many
 * chains in flight.  So:
 *   - hash() of more than YAMS_HASH_LONE_CHAIN_MAX bytes returns YAMS_ERR_UNSUPPORTED;
 *   - hash_many() / verify_many() return YAMS_ERR_UNSUPPORTED when the call cannot finish before ONE host core
 *     would: longest message > max(YAMS_HASH_LONE_CHAIN_MAX, total bytes / YAMS_HASH_CHAIN_RATIO);
 * and the host hashes those with its own SHA256Hasher — the reference's convention for optional features
 * (abi_model_provider_adapter.cpp:121-122,159-170; UNSUPPORTED -> ErrorCode::NotImplemented, :528-552).
 * AccelSHA256Hasher (include/yams_accel/hasher.hpp) takes the host's hasher for exactly that.  The stream_*
 * functions are a compatibility door for hosts without one: correct, never fast. */
#define YAMS_HASH_LONE_CHAIN_MAX (1u << 20)
#define YAMS_HASH_CHAIN_RATIO 37u /* ~ 1.3 GB/s per host core / 35 MB/s per device chain */
typedef struct yams_content_hash_v1 {
    uint32_t abi_version; /* YAMS_IFACE_CONTENT_HASH_V1_VERSION */
    void* self;
    /* IContentHasher::hash one-shot: 64 hex chars + NUL into out_hex.  More than YAMS_HASH_LONE_CHAIN_MAX bytes:
     * YAMS_ERR_UNSUPPORTED (see above). */
    yams_status_t (*hash)(void* self, const uint8_t* data, size_t n, char out_hex[65]);
    /* Many messages per call (what makes a GPU worthwhile). */
    yams_status_t (*hash_many)(void* self, const uint8_t* const* msgs, const size_t* lens,
                               size_t n_msgs, char* out_hex /* [n_msgs][65] */);
    /* Streaming init/update/finalize (sha256_hasher.cpp:81-109): state lives in an opaque handle;
     * finalize re-initialises the handle like the reference (:103-106). */
    yams_status_t (*stream_create)(void* self, void** out_stream);
    yams_status_t (*stream_init)(void* self, void* stream);
    yams_status_t (*stream_update)(void* self, void* stream, const uint8_t* data, size_t n);
    yams_status_t (*stream_finalize)(void* self, void* stream, char out_hex[65]);
    void (*stream_destroy)(void* self, void* stream);
    /* Batched integrity check (ChunkValidator::validateChunks, src/integrity/chunk_validator.cpp:
     * 160-212): out_valid[i] = 1 iff SHA-256(msgs[i]) == expected_hex[i] (64 hex chars, either case;
     * a malformed expectation is simply a mismatch). */
    yams_status_t (*verify_many)(void* self, const uint8_t* const* msgs, const size_t* lens,
                                 const char* expected_hex /* [n_msgs][65] */, size_t n_msgs,
                                 uint8_t* out_valid);
    /* Device-resident set of known chunk hashes (the batched `storage_->exists` of
     * ContentStore::store, src/api/content_store_impl.cpp:246-287).  dedup_insert answers, per hash
     * and in order, "is this one new?" (neither in the set nor earlier in the same call) and adds
     * the new ones; dedup_contains only looks. */
    yams_status_t (*dedup_create)(void* self, uint64_t expected_entries, uint64_t* out_set_id);
    yams_status_t (*dedup_insert)(void* self, uint64_t set_id, const char* hashes_hex /* [n][65] */,
                                  size_t n, uint8_t* out_is_new);
    yams_status_t (*dedup_contains)(void* self, uint64_t set_id, const char* hashes_hex, size_t n,
                                    uint8_t* out_exists);
    yams_status_t (*dedup_size)(void* self, uint64_t set_id, uint64_t* out_entries);
    yams_status_t (*dedup_destroy)(void* self, uint64_t set_id);
} yams_content_hash_v1;

typedef struct yams_chunk_ref_s { /* ChunkRef, chunker.h:32-41 (hash as hex) */
    uint64_t offset;
    uint64_t size;
    char hash_hex[65];
    char pad[7];
} yams_chunk_ref_t;

/* Result of chunk_many: the chunk tables of all buffers back to back. */
typedef struct yams_chunk_batch_s {
    size_t n_buffers;
    size_t n_chunks;
    size_t* first_chunk;       /* [n_buffers + 1]: buffer b owns chunks[first_chunk[b] .. first_chunk[b + 1])     */
    yams_chunk_ref_t* chunks;  /* [n_chunks]: offset within its buffer, size, SHA-256 of the chunk (hex)          */
    char* buffer_hash_hex;     /* [n_buffers][65]: SHA-256 of each whole buffer (the file hash of
                                  ContentStore::store, content_store_impl.cpp:199-231), or NULL if not asked for  */
} yams_chunk_batch_t;
#define YAMS_CHUNK_MANY_BUFFER_HASHES 1u
/* With YAMS_CHUNK_MANY_BUFFER_HASHES: buffers longer than yams_ingest_defer_threshold_host(sum of lens) come back with
 * an EMPTY buffer_hash_hex entry (first byte 0) — their whole-buffer chain would outlast the rest of the call; the
 * adapter's host hasher computes them while the device works (AccelChunker::chunkMany).  See
 * YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS. */
#define YAMS_CHUNK_MANY_DEFER_LONG_BUFFER_HASHES 2u

typedef struct yams_chunker_v1 {
    uint32_t abi_version; /* YAMS_IFACE_CHUNKER_V1_VERSION */
    void* self;
    yams_status_t (*get_default_config)(void* self, uint32_t mode, yams_cdc_config_t* out_cfg);
    /* IChunker::chunkDataLazy: offsets, sizes and per-chunk SHA-256. */
    yams_status_t (*chunk_data)(void* self, const uint8_t* data, size_t n,
                                const yams_cdc_config_t* cfg, yams_chunk_ref_t** out_chunks,
                                size_t* out_count);
    void (*free_chunks)(void* self, yams_chunk_ref_t* chunks, size_t count);
    /* ---- version 2 (NULL in older builds: "not implemented", abi_model_provider_adapter.cpp:121-122) --------
     * MANY buffers per call — the shape that lets the device run at its ingest rate (yams_ingest_host: uploads
     * of batch i + 1 under the kernels of batch i; ~500 GB/s device-resident, PCIe-bound from host memory)
     * instead of one launch sequence per file: boundaries + per-chunk SHA-256 of every buffer and, with
     * YAMS_CHUNK_MANY_BUFFER_HASHES, the whole-buffer digests — everything ContentStore::store needs for a
     * batch of files (content_store_impl.cpp:199-231) in one call.  Release with free_chunk_batch. */
    yams_status_t (*chunk_many)(void* self, const uint8_t* const* buffers, const size_t* lens, size_t n_buffers,
                                const yams_cdc_config_t* cfg, uint32_t flags, yams_chunk_batch_t** out_batch);
    void (*free_chunk_batch)(void* self, yams_chunk_batch_t* batch);
    /* ---- version 3 -----------------------------------------------------------------------------------------
     * One window of a stream (yams_cdc_chunk_window_host): the first context_len bytes are history, chunks start
     * behind them; what AccelChunker::processStream calls once per window.  Release with free_chunks. */
    yams_status_t (*chunk_window)(void* self, const uint8_t* data, size_t n, size_t context_len,
                                  const yams_cdc_config_t* cfg, yams_chunk_ref_t** out_chunks, size_t* out_count);
} yams_chunker_v1;

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* YAMS_MI355X_ACCEL_H */
