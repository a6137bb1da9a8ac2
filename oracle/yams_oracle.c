/*
 * oracle/yams_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  Nothing under yams_amd/ or include/ links, loads or calls it.
 *
 * Every function cites the reference lines (under /root/reference) it follows.  The SHA-256 and
 * CDC restatements are pinned against the reference's own translation units (oracle/_ref, built by
 * oracle/Makefile from the sources where they lie) and against the reference's known-answer
 * vectors (tests/unit/crypto/crypto_test.cpp:92-99) in tests/test_oracle.py.  The exact cosine
 * scan is a line-for-line restatement (the reference file cannot be compiled here: its
 * sqlite-vec-cpp / simeon submodules are empty) pinned by the reference's vector known-answer
 * tests (tests/unit/vector/vector_smoke_catch2_test.cpp:188-353).  The L2 scan restates an ABSENT
 * dependency (trvon/sqlite-vec-cpp, unpinned revision): parity unpinned for L2.
 *
 * Plain C, scalar, single-threaded by design: this is the "port" cpu_baseline.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * SHA-256 (FIPS 180-4).  Reference: src/crypto/sha256_hasher.cpp:167-195 (one-shot hash through
 * OpenSSL EVP_sha256, pinned openssl/3.2.0 in conanfile.py:95) and :19-30 (lower-case hex).
 * ---------------------------------------------------------------------------------------------- */
static const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u,
    0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu,
    0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu,
    0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u,
    0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu,
    0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu,
    0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u,
    0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,
    0xc67178f2u};

static inline uint32_t ror32(uint32_t x, unsigned n) { return (x >> n) | (x << (32u - n)); }

static void sha256_compress(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) {
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | (uint32_t)blk[4 * i + 3];
    }
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6],
             h = st[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

ORACLE_API void oracle_sha256(const uint8_t* data, size_t n, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                      0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    size_t full = n / 64;
    for (size_t i = 0; i < full; ++i) sha256_compress(st, data + 64 * i);
    uint8_t tail[128];
    size_t rem = n - 64 * full;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)n * 8u;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_compress(st, tail);
    if (tl == 128) sha256_compress(st, tail + 64);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

/* bytesToHex, src/crypto/sha256_hasher.cpp:19-30: lower-case, 64 chars + NUL. */
ORACLE_API void oracle_sha256_hex(const uint8_t* data, size_t n, char out[65]) {
    static const char hexd[] = "0123456789abcdef";
    uint8_t dg[32];
    oracle_sha256(data, n, dg);
    for (int i = 0; i < 32; ++i) { out[2 * i] = hexd[dg[i] >> 4]; out[2 * i + 1] = hexd[dg[i] & 15]; }
    out[64] = 0;
}

/* ------------------------------------------------------------------------------------------------
 * Rolling-hash CDC.  Table: src/chunking/rabin_fingerprint_table.h:12-28.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t window_size;  /* ChunkingConfig::windowSize, chunker.h:45 (default 48)            */
    uint64_t min_size;     /* minChunkSize  (default 16 KiB, core/types.h:282)                 */
    uint64_t max_size;     /* maxChunkSize  (default 1 MiB,  core/types.h:284)                 */
    uint64_t polynomial;   /* default 0x3DA3358B4DC173, chunker.h:49                            */
    uint64_t mask;         /* chunkMask, default 0x1FFF, chunker.h:50                           */
} oracle_cdc_config;

ORACLE_API void oracle_rabin_table(uint64_t polynomial, uint64_t out[256]) {
    /* rabin_fingerprint_table.h:17-26 */
    for (int byte = 0; byte < 256; ++byte) {
        uint64_t h = 0;
        for (int bit = 0; bit < 8; ++bit)
            if (byte & (1 << bit)) h ^= polynomial << bit;
        out[byte] = h;
    }
}

/* RabinChunker::chunkDataImpl + findChunkBoundary, src/chunking/rabin_chunker.cpp:63-152.
 * One RabinWindow (48-byte ring, pos, hash) for the whole buffer, never reset (:126).
 * Returns the number of chunks; writes at most cap (offset,size) pairs. */
ORACLE_API size_t oracle_chunk_rabin(const uint8_t* data, size_t n, const oracle_cdc_config* cfg,
                                     uint64_t* offsets, uint64_t* sizes, size_t cap) {
    uint64_t table[256];
    uint64_t poly = cfg->polynomial ? cfg->polynomial : 0x3DA3358B4DC173ULL; /* :29-37 */
    oracle_rabin_table(poly, table);
    uint8_t ring[48];
    memset(ring, 0, sizeof ring);
    size_t ring_pos = 0;
    uint64_t hash = 0;
    const size_t wsz = (size_t)cfg->window_size;
    size_t count = 0, pos = 0;
    while (pos < n) { /* :128 */
        const size_t start = pos;
        size_t min_b = start + cfg->min_size; if (min_b > n) min_b = n;   /* :67 */
        size_t max_b = start + cfg->max_size; if (max_b > n) max_b = n;   /* :68 */
        size_t p = start, end;
        int found = 0;
        while (p < min_b) { /* :78-88, no boundary test */
            uint8_t nb = data[p++], ob = ring[ring_pos];
            ring[ring_pos] = nb;
            if (++ring_pos == wsz) ring_pos = 0;
            hash = ((hash - table[ob]) << 8) ^ table[nb];
        }
        while (p < max_b) { /* :90-105 */
            uint8_t nb = data[p], ob = ring[ring_pos];
            ring[ring_pos] = nb;
            if (++ring_pos == wsz) ring_pos = 0;
            hash = ((hash - table[ob]) << 8) ^ table[nb];
            if ((hash & cfg->mask) == cfg->mask) { found = 1; break; }
            ++p;
        }
        end = found ? p + 1 : p; /* :101-109 */
        if (count < cap) { offsets[count] = start; sizes[count] = end - start; }
        ++count;
        pos = end;
    }
    return count;
}

/* StreamingChunker::processBuffer / emitChunk / updateRabinHash,
 * include/yams/chunking/streaming_chunker.h:146-204, src/chunking/streaming_chunker.cpp:37-69.
 * The result does not depend on how the stream is fragmented into buffers, so one pass. */
ORACLE_API size_t oracle_chunk_streaming(const uint8_t* data, size_t n,
                                         const oracle_cdc_config* cfg, uint64_t* offsets,
                                         uint64_t* sizes, size_t cap) {
    uint64_t table[256];
    uint64_t poly = cfg->polynomial ? cfg->polynomial : 0x3DA3358B4DC173ULL;
    oracle_rabin_table(poly, table);
    uint8_t ring[48];
    memset(ring, 0, sizeof ring);
    size_t wsz = (size_t)cfg->window_size; /* .cpp:44-49: 0 -> 1, > 48 -> 48 */
    if (wsz == 0) wsz = 1; else if (wsz > 48) wsz = 48;
    size_t ring_pos = 0;
    uint64_t hash = 0;
    size_t count = 0, acc = 0, chunk_start = 0;
    for (size_t off = 0; off < n; ++off) {
        ++acc; /* accumulator.push_back, .h:153 */
        if (ring_pos >= wsz) ring_pos = 0; /* .cpp:50-52 */
        uint8_t nb = data[off], ob = ring[ring_pos];
        ring[ring_pos] = nb;
        if (++ring_pos >= wsz) ring_pos = 0;
        hash = ((hash - table[ob]) << 8) ^ table[nb]; /* .cpp:68 */
        int emit = 0;
        if (acc >= cfg->min_size) { /* .h:162-170 */
            if ((hash & cfg->mask) == cfg->mask) emit = 1;
            else if (acc >= cfg->max_size) emit = 1;
        }
        if (emit) { /* .h:184-204 */
            if (count < cap) { offsets[count] = chunk_start; sizes[count] = acc; }
            ++count;
            chunk_start = off + 1;
            acc = 0;
        }
    }
    if (acc) { /* finalizeChunk, .h:116-118,207-211 */
        if (count < cap) { offsets[count] = chunk_start; sizes[count] = acc; }
        ++count;
    }
    return count;
}

/* ------------------------------------------------------------------------------------------------
 * Exact cosine scan.  Reference: SqliteVecBackend::Impl::bruteForceSearchUnlocked fast path,
 * src/vector/sqlite_vec_backend.cpp:4123-4135 (pre-checks), :204-211 (zero-norm threshold 1e-10),
 * :229-236 (finite), :4204-4211 (query norm), :4228-4307 (row loop, skips, bounded heap),
 * :4315-4326 (final order, relevance_score).  The corpus is a dense row-major fp32 matrix in
 * rowid order (:4175 ORDER BY rowid); the chunk_id string tie-break (:4218-4223) is modelled by
 * tie_rank[row] = rank of that row's chunk_id in lexicographic order (NULL: rank = row index).
 * Return: number of results (<= k), or -1 for the InvalidArgument case (:4127-4130).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float sim; uint64_t rank; int64_t row; } oracle_hit;

static int hit_better(const oracle_hit* a, const oracle_hit* b) { /* :4218-4223 */
    if (a->sim != b->sim) return a->sim > b->sim;
    return a->rank < b->rank;
}
/* std::push_heap / pop_heap with comparator `better` keep the WORST retained row at the front. */
static void heap_sift_up(oracle_hit* h, size_t i) {
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (hit_better(&h[p], &h[i])) { oracle_hit t = h[p]; h[p] = h[i]; h[i] = t; i = p; }
        else break;
    }
}
static void heap_sift_down(oracle_hit* h, size_t n, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && hit_better(&h[m], &h[l])) m = l;
        if (r < n && hit_better(&h[m], &h[r])) m = r;
        if (m == i) break;
        oracle_hit t = h[m]; h[m] = h[i]; h[i] = t; i = m;
    }
}
static int hit_cmp_best_first(const void* pa, const void* pb) {
    const oracle_hit* a = (const oracle_hit*)pa; const oracle_hit* b = (const oracle_hit*)pb;
    if (hit_better(a, b)) return -1;
    if (hit_better(b, a)) return 1;
    return 0;
}

ORACLE_API int oracle_query_invalid(const float* q, size_t dim) {
    /* isFiniteEmbedding :229-236 + isZeroNormEmbedding :204-211 */
    double nsq = 0.0;
    for (size_t i = 0; i < dim; ++i) {
        if (!isfinite(q[i])) return 1;
        nsq += (double)q[i] * (double)q[i];
    }
    return nsq < 1e-10;
}

ORACLE_API long oracle_exact_scan_cosine(const float* corpus, size_t n_rows, size_t dim,
                                         const float* query, size_t k, float similarity_threshold,
                                         const uint64_t* tie_rank, int64_t* out_rows,
                                         float* out_sims, uint64_t* rows_visited,
                                         uint64_t* evaluations) {
    if (rows_visited) *rows_visited = 0;
    if (evaluations) *evaluations = 0;
    if (dim == 0 || k == 0) return 0;            /* :4123-4126 */
    if (oracle_query_invalid(query, dim)) return -1; /* :4127-4130 */
    double qn_sq = 0.0;
    for (size_t i = 0; i < dim; ++i) { double v = (double)query[i]; qn_sq += v * v; } /* :4206-4210 */
    const double qn = sqrt(qn_sq);               /* :4211 */
    oracle_hit* heap = (oracle_hit*)malloc(sizeof(oracle_hit) * (k + 1));
    size_t hs = 0;
    for (size_t r = 0; r < n_rows; ++r) {
        if (rows_visited) ++*rows_visited;       /* :4229-4231 */
        const float* e = corpus + r * dim;
        if (evaluations) ++*evaluations;         /* :4249-4251 */
        double nsq = 0.0, dot = 0.0; int finite = 1;
        for (size_t i = 0; i < dim; ++i) {       /* :4256-4266 */
            float v = e[i];
            if (!isfinite(v)) { finite = 0; break; }
            double sv = (double)v, qv = (double)query[i];
            nsq += sv * sv; dot += sv * qv;
        }
        if (!finite || nsq <= 1e-12) continue;   /* :4267-4269 */
        double denom = sqrt(nsq) * qn;           /* :4271 */
        double sd = denom > 0.0 ? dot / denom : 0.0;
        if (!isfinite(sd)) continue;             /* :4273-4275 */
        float sim = (float)sd;                   /* :4276 */
        if (sim < similarity_threshold) continue;/* :4277-4279 */
        oracle_hit h = {sim, tie_rank ? tie_rank[r] : (uint64_t)r, (int64_t)r};
        if (hs < k) { heap[hs] = h; heap_sift_up(heap, hs); ++hs; }           /* :4289-4295 */
        else if (hit_better(&h, &heap[0])) { heap[0] = h; heap_sift_down(heap, hs, 0); } /* :4296-4306 */
    }
    qsort(heap, hs, sizeof(oracle_hit), hit_cmp_best_first); /* :4315-4319 */
    for (size_t i = 0; i < hs; ++i) { out_rows[i] = heap[i].row; out_sims[i] = heap[i].sim; }
    free(heap);
    return (long)hs;
}

/* The record path of the same function, src/vector/sqlite_vec_backend.cpp:4333-4409: taken when a
 * search carries metadata_filters.  `allow` (nullable, one byte per row) stands for everything the
 * reference decides before it scores a row: the SQL restriction (:4137-4175), the candidate_hashes
 * re-check (:4339-4345) and the metadata predicate (:4350-4360).  Then: skip rows with
 * isZeroNormEmbedding (norm^2 < 1e-10, :204-211) or a non-finite element (:4365-4367); similarity =
 * float(computeCosineSimilarity(query, row)) (:4373-4374); skip if < threshold (:4375-4377); full
 * sort by (similarity desc, chunk_id asc) (:4391-4396); first k (or all of them when k == 0 stands
 * for ExactRowSelection::AllMatching, :4398-4400).  evaluations counts the rows that got a score
 * (:4369-4371). */
static double oracle_cosine_similarity_impl(const float* a, const float* b, size_t dim);
typedef struct { float sim; uint64_t rank; int64_t row; } oracle_rhit;
static int rhit_cmp(const void* pa, const void* pb) {
    const oracle_rhit* a = (const oracle_rhit*)pa; const oracle_rhit* b = (const oracle_rhit*)pb;
    if (a->sim != b->sim) return a->sim > b->sim ? -1 : 1;
    if (a->rank != b->rank) return a->rank < b->rank ? -1 : 1;
    return 0;
}
ORACLE_API long oracle_exact_scan_cosine_records(const float* corpus, size_t n_rows, size_t dim,
                                                 const float* query, size_t k, int all_matching,
                                                 float similarity_threshold,
                                                 const uint64_t* tie_rank, const uint8_t* allow,
                                                 int64_t* out_rows, float* out_sims,
                                                 uint64_t* evaluations) {
    if (evaluations) *evaluations = 0;
    if (dim == 0 || (k == 0 && !all_matching)) return 0;   /* :4123-4126 */
    if (oracle_query_invalid(query, dim)) return -1;       /* :4127-4130 */
    oracle_rhit* all = (oracle_rhit*)malloc(sizeof(oracle_rhit) * (n_rows ? n_rows : 1));
    size_t m = 0;
    for (size_t r = 0; r < n_rows; ++r) {
        if (allow && !allow[r]) continue;
        const float* e = corpus + r * dim;
        double nsq = 0.0; int finite = 1;
        for (size_t i = 0; i < dim; ++i) { nsq += (double)e[i] * (double)e[i]; if (!isfinite(e[i])) finite = 0; }
        if (nsq < 1e-10 || !finite) continue;              /* :4365-4367 */
        if (evaluations) ++*evaluations;                   /* :4369-4371 */
        float sim = (float)oracle_cosine_similarity_impl(query, e, dim); /* :4373-4374 */
        if (sim < similarity_threshold) continue;          /* :4375-4377 */
        all[m].sim = sim; all[m].rank = tie_rank ? tie_rank[r] : (uint64_t)r; all[m].row = (int64_t)r; ++m;
    }
    qsort(all, m, sizeof(oracle_rhit), rhit_cmp);          /* :4391-4396 */
    size_t cnt = all_matching ? m : (k < m ? k : m);       /* :4398-4400 */
    for (size_t i = 0; i < cnt; ++i) { out_rows[i] = all[i].row; out_sims[i] = all[i].sim; }
    free(all);
    return (long)cnt;
}

/* VectorDatabase::computeCosineSimilarity, src/vector/vector_database.cpp:1786-1810. */
static double oracle_cosine_similarity_impl(const float* a, const float* b, size_t dim) {
    double dp = 0.0, na = 0.0, nb = 0.0;
    for (size_t i = 0; i < dim; ++i) {
        dp += (double)a[i] * (double)b[i];
        na += (double)a[i] * (double)a[i];
        nb += (double)b[i] * (double)b[i];
    }
    na = sqrt(na); nb = sqrt(nb);
    if (na == 0.0 || nb == 0.0) return 0.0;
    return dp / (na * nb);
}

/* VectorDatabase::computeCosineSimilarity, src/vector/vector_database.cpp:1786-1810. */
ORACLE_API double oracle_cosine_similarity(const float* a, const float* b, size_t dim) {
    return oracle_cosine_similarity_impl(a, b, dim);
}

/* ------------------------------------------------------------------------------------------------
 * L2 (vec0) scan — PARITY UNPINNED.  The arithmetic lives in the absent third_party/sqlite-vec-cpp
 * (.gitmodules:4-6, no recoverable revision).  What the reference pins at this boundary:
 * tests/unit/vector/sqlite_vec_c_api_smoke_catch2_test.cpp:20-43 (Euclidean distance WITH sqrt),
 * and the caller src/vector/sqlite_vec_backend.cpp:4450-4530: k nearest by distance ascending,
 * each hit re-scored with computeCosineSimilarity (:4506), dropped if below the threshold (:4508),
 * returned in distance order with relevance_score = cosine (:4512).  Restated here as:
 * distance = (float)sqrt(sum_i ((double)a_i - (double)b_i)^2), ascending, ties in ROW (= rowid) order.
 * vec0's `k = ?2` bounds the candidate list BEFORE the cosine threshold filter (:4464-4473), so
 * fewer than k rows may come back.
 * Round 6: everything of this function EXCEPT the distance arithmetic is pinned by the reference's own
 * vec0SearchUnlocked (+ getVectorByRowidUnlocked, rebuildVec0DimUnlocked), compiled over SQLite with a harness `vec0`
 * module (oracle/scan_ref_wrap.cpp, tests/test_scan_ref_l2_pin.py): k nearest THEN the threshold, rows the vectors table
 * no longer holds skipped, non-finite / wrong-size rows never indexed, k > n, the candidate-rowid branch, and the order
 * of EQUAL distances — the statement is `ORDER BY distance` alone (:4473) and SQLite's sorter returns ties in the order
 * the table handed the rows over, i.e. rowid order, whatever the chunk ids.  `tie_rank` (the chunk_id ranking of the
 * cosine comparator, :4218-4223) is therefore IGNORED here; the parameter stays for the callers' convenience.
 * ---------------------------------------------------------------------------------------------- */
/* the default definition of the distance, with the signature the harness's vec0 module calls (mode unused) */
ORACLE_API float oracle_l2_distance_f64(const float* a, const float* b, size_t dim, int mode) {
    (void)mode;
    double acc = 0.0;
    for (size_t i = 0; i < dim; ++i) { const double d = (double)a[i] - (double)b[i]; acc += d * d; }
    return (float)sqrt(acc);
}
typedef struct { float dist; uint64_t rank; int64_t row; } oracle_l2hit;
static int l2_cmp(const void* pa, const void* pb) {
    const oracle_l2hit* a = (const oracle_l2hit*)pa; const oracle_l2hit* b = (const oracle_l2hit*)pb;
    if (a->dist != b->dist) return a->dist < b->dist ? -1 : 1;
    if (a->rank != b->rank) return a->rank < b->rank ? -1 : 1;
    return 0;
}
ORACLE_API long oracle_exact_scan_l2(const float* corpus, size_t n_rows, size_t dim,
                                     const float* query, size_t k, float similarity_threshold,
                                     const uint64_t* tie_rank, int64_t* out_rows, float* out_dist,
                                     float* out_sims) {
    if (dim == 0 || k == 0) return 0; /* :4453-4455 */
    oracle_l2hit* all = (oracle_l2hit*)malloc(sizeof(oracle_l2hit) * (n_rows ? n_rows : 1));
    size_t m = 0;
    for (size_t r = 0; r < n_rows; ++r) {
        const float* e = corpus + r * dim;
        double acc = 0.0; int finite = 1;
        for (size_t i = 0; i < dim; ++i) {
            if (!isfinite(e[i])) { finite = 0; break; }
            double d = (double)e[i] - (double)query[i];
            acc += d * d;
        }
        if (!finite) continue; /* insert-time validity, src/vector/vector_database.cpp:1771-1784 */
        double dd = sqrt(acc);
        if (!isfinite(dd)) continue;
        all[m].dist = (float)dd; all[m].rank = (uint64_t)r; (void)tie_rank; /* ties: rowid order (see above) */
        all[m].row = (int64_t)r; ++m;
    }
    qsort(all, m, sizeof(oracle_l2hit), l2_cmp);
    size_t take = m < k ? m : k, outn = 0;
    for (size_t i = 0; i < take; ++i) {
        float sim = (float)oracle_cosine_similarity(query, corpus + (size_t)all[i].row * dim, dim);
        if (sim < similarity_threshold) continue; /* :4508-4510 */
        out_rows[outn] = all[i].row; out_dist[outn] = all[i].dist; out_sims[outn] = sim; ++outn;
    }
    free(all);
    return (long)outn;
}

/* ------------------------------------------------------------------------------------------------
 * The OTHER plausible definition of the vec0 distance: fp32 accumulation.  The public sqlite-vec (and, most
 * likely, the absent trvon/sqlite-vec-cpp) accumulates (a_i - b_i)^2 in float — sequentially in its scalar form,
 * in 4 / 8 / 16 independent lanes that are summed at the end in its NEON / AVX forms (src/vector/meson.build:76-106
 * sets the SIMD flags for that dependency).  The fp64 definition above is THIS repository's choice; these variants
 * exist so that tests and bench.py can REPORT how far the choice matters (distances agree to ~1e-6 relative, the
 * top-k index SET differs only when two rows sit within that of each other at the cut): `lanes` = 1 sequential,
 * else that many round-robin partial sums (element i goes to lane i % lanes) added left to right.
 * ---------------------------------------------------------------------------------------------- */
/* lanes < 0 (round 5): |lanes| lanes with every square accumulated by ONE fused multiply-add, part = fmaf(d, d, part) — what a
 * compiler makes of the loops above when the translation unit is built with -mfma, as the reference builds sqlite-vec-cpp on
 * x86 (src/vector/meson.build:80-88: '-mavx', '-mfma'; GCC contracts `sum += d * d` and _mm256_add_ps(sum, _mm256_mul_ps(d, d))
 * by default).  fmaf is the correctly rounded fused operation whether or not this host has the instruction. */
ORACLE_API float oracle_l2_distance_f32acc(const float* a, const float* b, size_t dim, int lanes) {
    const int fused = lanes < 0;
    if (fused) lanes = -lanes;
    if (lanes <= 1) {
        float acc = 0.0f;
        for (size_t i = 0; i < dim; ++i) { const float d = a[i] - b[i]; acc = fused ? fmaf(d, d, acc) : acc + d * d; }
        return sqrtf(acc);
    }
    float part[64];
    if (lanes > 64) lanes = 64;
    for (int l = 0; l < lanes; ++l) part[l] = 0.0f;
    for (size_t i = 0; i < dim; ++i) {
        const float d = a[i] - b[i];
        float* p = &part[i % (size_t)lanes];
        *p = fused ? fmaf(d, d, *p) : *p + d * d;
    }
    float acc = 0.0f;
    for (int l = 0; l < lanes; ++l) acc += part[l];
    return sqrtf(acc);
}
/* rows[m][dim] against one query */
ORACLE_API void oracle_l2_distance_f32acc_many(const float* rows, const float* query, size_t m, size_t dim, int lanes,
                                               float* out) {
    for (size_t r = 0; r < m; ++r) out[r] = oracle_l2_distance_f32acc(rows + r * dim, query, dim, lanes);
}
/* the whole scan under that definition (same contract as oracle_exact_scan_l2) */
ORACLE_API long oracle_exact_scan_l2_f32acc(const float* corpus, size_t n_rows, size_t dim, const float* query, size_t k,
                                            float similarity_threshold, const uint64_t* tie_rank, int lanes,
                                            int64_t* out_rows, float* out_dist, float* out_sims) {
    if (dim == 0 || k == 0) return 0;
    oracle_l2hit* all = (oracle_l2hit*)malloc(sizeof(oracle_l2hit) * (n_rows ? n_rows : 1));
    size_t m = 0;
    for (size_t r = 0; r < n_rows; ++r) {
        const float* e = corpus + r * dim;
        int finite = 1;
        for (size_t i = 0; i < dim; ++i) if (!isfinite(e[i])) { finite = 0; break; }
        if (!finite) continue;
        const float dd = oracle_l2_distance_f32acc(e, query, dim, lanes);
        if (!isfinite(dd)) continue;
        all[m].dist = dd; all[m].rank = (uint64_t)r; (void)tie_rank; all[m].row = (int64_t)r; ++m; /* ties: rowid order */
    }
    qsort(all, m, sizeof(oracle_l2hit), l2_cmp);
    size_t take = m < k ? m : k, outn = 0;
    for (size_t i = 0; i < take; ++i) {
        float sim = (float)oracle_cosine_similarity(query, corpus + (size_t)all[i].row * dim, dim);
        if (sim < similarity_threshold) continue;
        out_rows[outn] = all[i].row; out_dist[outn] = all[i].dist; out_sims[outn] = sim; ++outn;
    }
    free(all);
    return (long)outn;
}

/* ------------------------------------------------------------------------------------------------
 * The product-quantised engine: SqliteVecBackend::Impl::simeonPqSearchUnlocked,
 * src/vector/sqlite_vec_backend.cpp:3868-4056, restated.  PARITY UNPINNED: third_party/simeon (ProductQuantizer,
 * PQInnerProductQuery::inner_product) is absent from the reference checkout — the ORDER of the fp32 additions of the ADC
 * score is simeon's.  `sum_lanes` = 1: one sequential sum over the sub-quantisers; 4 / 8 / 16: that many partial sums
 * (element j -> lane j % lanes) added left to right.  Everything around that sum is the reference's own text:
 *   :3873-3880  empty query / k == 0 / no index -> nothing            :3895-3898  query norm^2 <= 1e-20 -> nothing
 *   :3938-3948  candidateCount = all indexed rows or the candidate indices; 0 -> nothing
 *   :3952-3960  approxK = min(candidateCount, max(k, k * rerank_factor)) (k * rerank saturating)
 *   :3985-3997  best approxK by (score desc, tie key asc); equal (score, key) pairs are ordered by INDEX here (the
 *               reference leaves that to nth_element / sort)
 *   :4006-4016  rows the vectors table no longer holds are skipped (row_of_index >= n_rows here)
 *   :4023-4038  similarity = (float)computeCosineSimilarity(query, row), dropped when < threshold
 *   :4041-4051  sorted by (similarity desc, chunk_id asc), cut to k.
 * A NaN similarity (a non-finite row, which cannot be stored: vector_database.cpp:1771-1784) is left out.
 * out_stats (nullable) = {candidateCount, rows materialised (exactDistanceEvaluations)}.
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API float oracle_pq_adc_score(const uint8_t* code, size_t m, const float* lut, int sum_lanes) {
    if (sum_lanes <= 1) {
        float acc = 0.0f;
        for (size_t j = 0; j < m; ++j) acc = acc + lut[j * 256 + code[j]];
        return acc;
    }
    float part[16];
    if (sum_lanes > 16) sum_lanes = 16;
    for (int l = 0; l < sum_lanes; ++l) part[l] = 0.0f;
    for (size_t j = 0; j < m; ++j) part[j % (size_t)sum_lanes] += lut[j * 256 + code[j]];
    float acc = 0.0f;
    for (int l = 0; l < sum_lanes; ++l) acc += part[l];
    return acc;
}
typedef struct { float score; uint64_t key; size_t index; } oracle_pqhit;
static int pq_cmp(const void* pa, const void* pb) {
    const oracle_pqhit* a = (const oracle_pqhit*)pa; const oracle_pqhit* b = (const oracle_pqhit*)pb;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    if (a->index != b->index) return a->index < b->index ? -1 : 1;
    return 0;
}
typedef struct { float sim; uint64_t rank; int64_t row; } oracle_pqrec;
static int pqrec_cmp(const void* pa, const void* pb) {
    const oracle_pqrec* a = (const oracle_pqrec*)pa; const oracle_pqrec* b = (const oracle_pqrec*)pb;
    if (a->sim != b->sim) return a->sim > b->sim ? -1 : 1;
    if (a->rank != b->rank) return a->rank < b->rank ? -1 : 1;
    return 0;
}
ORACLE_API long oracle_pq_search(const float* corpus, size_t n_rows, size_t dim, const uint8_t* codes, size_t n_codes, size_t m,
                                 const float* lut, const uint64_t* tie_keys, const uint32_t* row_of_index, const uint64_t* chunk_rank,
                                 const float* query, size_t k, float threshold, size_t rerank_factor, const uint32_t* candidates,
                                 size_t n_candidates, int sum_lanes, int64_t* out_rows, float* out_sims, uint64_t* out_stats) {
    if (out_stats) { out_stats[0] = 0; out_stats[1] = 0; }
    if (dim == 0 || k == 0 || n_codes == 0) return 0;
    double nsq = 0.0;
    for (size_t i = 0; i < dim; ++i) nsq += (double)query[i] * (double)query[i];
    if (!(nsq > 1e-20) || !isfinite(nsq)) return 0;            /* normalizeEmbeddingInPlace fails (:213-226) */
    const size_t count = candidates ? n_candidates : n_codes;
    if (out_stats) out_stats[0] = count;
    if (count == 0) return 0;
    if (rerank_factor == 0) rerank_factor = 1;
    const size_t budget = k > (size_t)-1 / rerank_factor ? (size_t)-1 : k * rerank_factor;
    size_t approx = k > budget ? k : budget;
    if (approx > count) approx = count;
    oracle_pqhit* all = (oracle_pqhit*)malloc(sizeof(oracle_pqhit) * count);
    size_t n = 0;
    for (size_t c = 0; c < count; ++c) {
        const size_t idx = candidates ? candidates[c] : c;
        if (idx >= n_codes) continue;
        all[n].score = oracle_pq_adc_score(codes + idx * m, m, lut, sum_lanes);
        all[n].key = tie_keys ? tie_keys[idx] : (uint64_t)idx; all[n].index = idx; ++n;
    }
    qsort(all, n, sizeof(oracle_pqhit), pq_cmp);
    if (approx > n) approx = n;
    oracle_pqrec* recs = (oracle_pqrec*)malloc(sizeof(oracle_pqrec) * (approx ? approx : 1));
    size_t nr = 0, materialised = 0;
    for (size_t i = 0; i < approx; ++i) {
        const size_t row = row_of_index ? row_of_index[all[i].index] : all[i].index;
        if (row >= n_rows) continue;                              /* getVectorByRowidUnlocked finds nothing (:4010-4012) */
        ++materialised;
        const float sim = (float)oracle_cosine_similarity_impl(query, corpus + row * dim, dim);
        if (sim != sim || sim < threshold) continue;              /* (:4036-4038; a NaN similarity: see above) */
        recs[nr].sim = sim; recs[nr].rank = chunk_rank ? chunk_rank[row] : (uint64_t)row; recs[nr].row = (int64_t)row; ++nr;
    }
    if (out_stats) out_stats[1] = materialised;
    qsort(recs, nr, sizeof(oracle_pqrec), pqrec_cmp);
    if (nr > k) nr = k;
    for (size_t i = 0; i < nr; ++i) { out_rows[i] = recs[i].row; out_sims[i] = recs[i].sim; }
    free(all); free(recs);
    return (long)nr;
}

/* ------------------------------------------------------------------------------------------------
 * MANY queries against one corpus slice: what lets a test check EVERY query of a BASELINE-size batch (1024 queries x
 * 12.5M rows) instead of two or four of them.  Not a second definition — the same arithmetic as
 * oracle_exact_scan_cosine / oracle_exact_scan_l2 above, per (row, query) the very same sequence of IEEE double
 * operations (products of two floats are exact in double; the sums run left to right over the elements), only
 *   * interleaved across QB queries, whose accumulation chains are independent (the single-query loop is bound by the
 *     latency of ONE dependent chain of additions; here the queries sit in the lanes of a vector),
 *   * with |row|^2 and the finite check, which do not depend on the query, computed once per row,
 *   * with a bounded heap under the same total order (similarity desc / distance asc, then row asc) in place of the
 *     L2 function's full sort: the k best under a total order do not depend on how they are found.
 * tests/test_oracle.py pins both against the single-query functions bit for bit (ties, zero rows, non-finite rows,
 * thresholds, ragged query counts).  Built with -ffp-contract=off like everything here; the vector types are GCC's
 * generic ones (lane-wise IEEE multiply and add: no reassociation, no fused operations).
 * ---------------------------------------------------------------------------------------------- */
#define ORACLE_QB 8
typedef double oracle_v2d __attribute__((vector_size(16)));
typedef double oracle_v4d __attribute__((vector_size(32), aligned(32)));
typedef struct { double v[ORACLE_QB]; } __attribute__((aligned(64))) oracle_vq; /* element i of QB queries, side by side */

/* dot[j] = sum_i row[i] * qT[i][j], left to right over i (the :4256-4266 loop, QB queries at once);
 * d2[j] = sum_i (row[i] - q_j[i])^2 likewise (oracle_exact_scan_l2's loop).  Two builds of the same statements: 128-bit
 * lanes (every x86-64 / aarch64) and 256-bit lanes where the host has AVX2 — lane-wise IEEE either way. */
#define ORACLE_RB_MAX 8 /* rows per kernel call: RB x QB independent chains hide the latency of the additions */
#define ORACLE_QB_KERNELS(SUFFIX, VT, NV, RB, ATTR)                                                                 \
    ATTR static void dots_qb_##SUFFIX(const float* const* row, const oracle_vq* qT, size_t dim, double* out) {        \
        VT acc[RB][NV];                                                                                               \
        for (int r = 0; r < RB; ++r) for (int v = 0; v < NV; ++v) acc[r][v] = (VT){0};                                \
        for (size_t i = 0; i < dim; ++i) {                                                                            \
            const VT* q = (const VT*)qT[i].v;                                                                         \
            for (int r = 0; r < RB; ++r) {                                                                            \
                const double sv = (double)row[r][i];                                                                  \
                for (int v = 0; v < NV; ++v) acc[r][v] = acc[r][v] + sv * q[v];                                       \
            }                                                                                                         \
        }                                                                                                             \
        for (int r = 0; r < RB; ++r) for (int v = 0; v < NV; ++v)                                                     \
            for (int l = 0; l < ORACLE_QB / NV; ++l) out[r * ORACLE_QB + v * (ORACLE_QB / NV) + l] = acc[r][v][l];    \
    }                                                                                                                 \
    ATTR static void sqdist_qb_##SUFFIX(const float* const* row, const oracle_vq* qT, size_t dim, double* out) {      \
        VT acc[RB][NV];                                                                                               \
        for (int r = 0; r < RB; ++r) for (int v = 0; v < NV; ++v) acc[r][v] = (VT){0};                                \
        for (size_t i = 0; i < dim; ++i) {                                                                            \
            const VT* q = (const VT*)qT[i].v;                                                                         \
            for (int r = 0; r < RB; ++r) {                                                                            \
                const double sv = (double)row[r][i];                                                                  \
                for (int v = 0; v < NV; ++v) { const VT d = sv - q[v]; acc[r][v] = acc[r][v] + d * d; }               \
            }                                                                                                         \
        }                                                                                                             \
        for (int r = 0; r < RB; ++r) for (int v = 0; v < NV; ++v)                                                     \
            for (int l = 0; l < ORACLE_QB / NV; ++l) out[r * ORACLE_QB + v * (ORACLE_QB / NV) + l] = acc[r][v][l];    \
    }
ORACLE_QB_KERNELS(v2, oracle_v2d, 4, 2, )
#if defined(__x86_64__)
typedef double oracle_v8d __attribute__((vector_size(64), aligned(64)));
ORACLE_QB_KERNELS(v4, oracle_v4d, 2, 4, __attribute__((target("avx2"))))
ORACLE_QB_KERNELS(v8, oracle_v8d, 1, 8, __attribute__((target("avx512f"))))
/* 0: 128-bit lanes, 1: AVX2, 2: AVX-512 — the widest the host has, unless oracle_set_lanes() pinned a narrower one (the
 * tests run all the host offers against the single-query functions) */
static int oracle_width_pin = -1;
static int oracle_width(void) {
    static int have = -1;
    if (have < 0) {
        __builtin_cpu_init();
        have = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
    }
    return oracle_width_pin >= 0 && oracle_width_pin < have ? oracle_width_pin : have;
}
/* bits: 128 / 256 / 512, anything else = the widest available; returns the width in effect (not thread-safe: call it
 * between scans) */
ORACLE_API int oracle_set_lanes(int bits) {
    oracle_width_pin = bits == 128 ? 0 : (bits == 256 ? 1 : (bits == 512 ? 2 : -1));
    return 128 << oracle_width();
}
static int oracle_rb(void) { const int w = oracle_width(); return w == 2 ? 8 : (w == 1 ? 4 : 2); }
static void dots_qb(const float* const* row, const oracle_vq* qT, size_t dim, double* out) {
    const int w = oracle_width();
    if (w == 2) dots_qb_v8(row, qT, dim, out); else if (w == 1) dots_qb_v4(row, qT, dim, out); else dots_qb_v2(row, qT, dim, out);
}
static void sqdist_qb(const float* const* row, const oracle_vq* qT, size_t dim, double* out) {
    const int w = oracle_width();
    if (w == 2) sqdist_qb_v8(row, qT, dim, out); else if (w == 1) sqdist_qb_v4(row, qT, dim, out); else sqdist_qb_v2(row, qT, dim, out);
}
#else
ORACLE_API int oracle_set_lanes(int bits) { (void)bits; return 128; }
static int oracle_rb(void) { return 2; }
#define dots_qb dots_qb_v2
#define sqdist_qb sqdist_qb_v2
#endif

#define ORACLE_ROW_BLOCK 64 /* rows per pass over the query groups: 64 x 768 floats = 192 KiB, stays in the core's L2 */

/* out_rows / out_sims: [nq][k], out_counts: [nq].  Returns 0, or -1 when a query is invalid (:4127-4130: that call
 * fails as a whole in the reference; check such a batch query by query). */
ORACLE_API long oracle_exact_scan_cosine_many(const float* corpus, size_t n_rows, size_t dim, const float* queries,
                                              size_t nq, size_t k, float similarity_threshold, int64_t* out_rows,
                                              float* out_sims, uint32_t* out_counts) {
    for (size_t q = 0; q < nq; ++q) out_counts[q] = 0;
    if (dim == 0 || k == 0 || nq == 0) return 0;
    for (size_t q = 0; q < nq; ++q) if (oracle_query_invalid(queries + q * dim, dim)) return -1;
    const size_t ng = (nq + ORACLE_QB - 1) / ORACLE_QB;
    oracle_vq* qT = (oracle_vq*)aligned_alloc(64, sizeof(oracle_vq) * ng * dim);
    double* qn = (double*)malloc(sizeof(double) * ng * ORACLE_QB);
    oracle_hit* heaps = (oracle_hit*)malloc(sizeof(oracle_hit) * nq * (k + 1));
    size_t* hs = (size_t*)calloc(nq, sizeof(size_t));
    for (size_t g = 0; g < ng; ++g)
        for (int j = 0; j < ORACLE_QB; ++j) {
            const size_t q = g * ORACLE_QB + j;
            double s = 0.0;
            for (size_t i = 0; i < dim; ++i) {
                const double v = q < nq ? (double)queries[q * dim + i] : 0.0;
                qT[g * dim + i].v[j] = v;
                s += v * v;                                  /* :4206-4210 */
            }
            qn[q] = sqrt(s);                                 /* :4211 */
        }
    double nsq[ORACLE_ROW_BLOCK]; int live[ORACLE_ROW_BLOCK];
    const size_t rb = (size_t)oracle_rb();
    for (size_t r0 = 0; r0 < n_rows; r0 += ORACLE_ROW_BLOCK) {
        const size_t nb = n_rows - r0 < ORACLE_ROW_BLOCK ? n_rows - r0 : ORACLE_ROW_BLOCK;
        for (size_t b = 0; b < nb; ++b) {
            const float* e = corpus + (r0 + b) * dim;
            double s = 0.0; int finite = 1;
            for (size_t i = 0; i < dim; ++i) {               /* :4256-4266 */
                if (!isfinite(e[i])) { finite = 0; break; }
                const double sv = (double)e[i];
                s += sv * sv;
            }
            nsq[b] = s; live[b] = finite && !(s <= 1e-12);   /* :4267-4269 */
        }
        for (size_t g = 0; g < ng; ++g)
            for (size_t b0 = 0; b0 < nb; b0 += rb) {
                const float* rp[ORACLE_RB_MAX]; int any = 0;
                for (int t = 0; t < rb; ++t) {              /* (a short tail repeats its last row; dead rows are computed and dropped) */
                    const size_t b = b0 + t < nb ? b0 + t : nb - 1;
                    rp[t] = corpus + (r0 + b) * dim;
                    any |= b0 + t < nb && live[b0 + t];
                }
                if (!any) continue;
                double dot[ORACLE_RB_MAX * ORACLE_QB];
                dots_qb(rp, qT + g * dim, dim, dot);
                for (int t = 0; t < rb && b0 + t < nb; ++t) {
                    const size_t b = b0 + t;
                    if (!live[b]) continue;
                    const double root = sqrt(nsq[b]);
                    for (int j = 0; j < ORACLE_QB; ++j) {
                        const size_t q = g * ORACLE_QB + j;
                        if (q >= nq) break;
                        const double denom = root * qn[q];       /* :4271 */
                        const double sd = denom > 0.0 ? dot[t * ORACLE_QB + j] / denom : 0.0;
                        if (!isfinite(sd)) continue;             /* :4273-4275 */
                        const float sim = (float)sd;             /* :4276 */
                        if (sim < similarity_threshold) continue;/* :4277-4279 */
                        oracle_hit h = {sim, (uint64_t)(r0 + b), (int64_t)(r0 + b)};
                        oracle_hit* heap = heaps + q * (k + 1);
                        if (hs[q] < k) { heap[hs[q]] = h; heap_sift_up(heap, hs[q]); ++hs[q]; }
                        else if (hit_better(&h, &heap[0])) { heap[0] = h; heap_sift_down(heap, hs[q], 0); }
                    }
                }
            }
    }
    for (size_t q = 0; q < nq; ++q) {
        oracle_hit* heap = heaps + q * (k + 1);
        qsort(heap, hs[q], sizeof(oracle_hit), hit_cmp_best_first);
        for (size_t i = 0; i < hs[q]; ++i) { out_rows[q * k + i] = heap[i].row; out_sims[q * k + i] = heap[i].sim; }
        out_counts[q] = (uint32_t)hs[q];
    }
    free(hs); free(heaps); free(qn); free(qT);
    return 0;
}

static int l2hit_worse(const oracle_l2hit* a, const oracle_l2hit* b) { return l2_cmp(a, b) > 0; }
static void l2heap_up(oracle_l2hit* h, size_t i) {          /* the WORST retained row at the front */
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (l2hit_worse(&h[i], &h[p])) { oracle_l2hit t = h[p]; h[p] = h[i]; h[i] = t; i = p; }
        else break;
    }
}
static void l2heap_down(oracle_l2hit* h, size_t n, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && l2hit_worse(&h[l], &h[m])) m = l;
        if (r < n && l2hit_worse(&h[r], &h[m])) m = r;
        if (m == i) break;
        oracle_l2hit t = h[m]; h[m] = h[i]; h[i] = t; i = m;
    }
}
/* The k nearest of every query (distance asc, row asc) with the cosine of each; the cosine threshold of :4508-4510 is
 * the caller's (it comes AFTER the k-nearest cut, and after the merge when the corpus is scanned slice by slice). */
ORACLE_API long oracle_exact_scan_l2_many(const float* corpus, size_t n_rows, size_t dim, const float* queries, size_t nq,
                                          size_t k, int64_t* out_rows, float* out_dist, float* out_sims,
                                          uint32_t* out_counts) {
    for (size_t q = 0; q < nq; ++q) out_counts[q] = 0;
    if (dim == 0 || k == 0 || nq == 0) return 0;
    const size_t ng = (nq + ORACLE_QB - 1) / ORACLE_QB;
    oracle_vq* qT = (oracle_vq*)aligned_alloc(64, sizeof(oracle_vq) * ng * dim);
    oracle_l2hit* heaps = (oracle_l2hit*)malloc(sizeof(oracle_l2hit) * nq * (k + 1));
    size_t* hs = (size_t*)calloc(nq, sizeof(size_t));
    for (size_t g = 0; g < ng; ++g)
        for (int j = 0; j < ORACLE_QB; ++j) {
            const size_t q = g * ORACLE_QB + j;
            for (size_t i = 0; i < dim; ++i) qT[g * dim + i].v[j] = q < nq ? (double)queries[q * dim + i] : 0.0;
        }
    int live[ORACLE_ROW_BLOCK];
    const size_t rb = (size_t)oracle_rb();
    for (size_t r0 = 0; r0 < n_rows; r0 += ORACLE_ROW_BLOCK) {
        const size_t nb = n_rows - r0 < ORACLE_ROW_BLOCK ? n_rows - r0 : ORACLE_ROW_BLOCK;
        for (size_t b = 0; b < nb; ++b) {
            const float* e = corpus + (r0 + b) * dim;
            int finite = 1;
            for (size_t i = 0; i < dim; ++i) if (!isfinite(e[i])) { finite = 0; break; }
            live[b] = finite;
        }
        for (size_t g = 0; g < ng; ++g)
            for (size_t b0 = 0; b0 < nb; b0 += rb) {
                const float* rp[ORACLE_RB_MAX]; int any = 0;
                for (int t = 0; t < rb; ++t) {
                    const size_t b = b0 + t < nb ? b0 + t : nb - 1;
                    rp[t] = corpus + (r0 + b) * dim;
                    any |= b0 + t < nb && live[b0 + t];
                }
                if (!any) continue;
                double d2[ORACLE_RB_MAX * ORACLE_QB];
                sqdist_qb(rp, qT + g * dim, dim, d2);
                for (int t = 0; t < rb && b0 + t < nb; ++t) {
                    const size_t b = b0 + t;
                    if (!live[b]) continue;
                    for (int j = 0; j < ORACLE_QB; ++j) {
                        const size_t q = g * ORACLE_QB + j;
                        if (q >= nq) break;
                        const double dd = sqrt(d2[t * ORACLE_QB + j]);
                        if (!isfinite(dd)) continue;
                        oracle_l2hit h = {(float)dd, (uint64_t)(r0 + b), (int64_t)(r0 + b)};
                        oracle_l2hit* heap = heaps + q * (k + 1);
                        if (hs[q] < k) { heap[hs[q]] = h; l2heap_up(heap, hs[q]); ++hs[q]; }
                        else if (l2_cmp(&h, &heap[0]) < 0) { heap[0] = h; l2heap_down(heap, hs[q], 0); }
                    }
                }
            }
    }
    for (size_t q = 0; q < nq; ++q) {
        oracle_l2hit* heap = heaps + q * (k + 1);
        qsort(heap, hs[q], sizeof(oracle_l2hit), l2_cmp);
        for (size_t i = 0; i < hs[q]; ++i) {
            out_rows[q * k + i] = heap[i].row; out_dist[q * k + i] = heap[i].dist;
            out_sims[q * k + i] = (float)oracle_cosine_similarity(queries + q * dim, corpus + (size_t)heap[i].row * dim, dim);
        }
        out_counts[q] = (uint32_t)hs[q];
    }
    free(hs); free(heaps); free(qT);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic-data recipes shared by tests and bench (so the CPU side can regenerate any slice).
 * Philox4x32-10 counter-based generator (Salmon et al., SC'11): key = (seed_lo, seed_hi),
 * counter = (i0, i1, i2, i3).  Used for corpora too large to hold on the host.
 * ---------------------------------------------------------------------------------------------- */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
ORACLE_API void oracle_philox4x32(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t out[4]) {
    uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi,
                     (uint32_t)(ctr_hi >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
/* Row r of the synthetic embedding matrix: U(-1,1) from the top 24 bits of each Philox word
 * (counter = (r, j/4)), then fp32 L2-normalised with the norm accumulated in float, as the
 * reference's generateNormalizedVector does (tests/benchmarks/vector_backend_engine_compare.cpp:
 * 83-99). */
ORACLE_API void oracle_synth_rows(uint64_t seed, uint64_t row0, size_t n_rows, size_t dim,
                                  float* out) {
    for (size_t r = 0; r < n_rows; ++r) {
        float* o = out + r * dim;
        float nsq = 0.0f;
        for (size_t j = 0; j < dim; j += 4) {
            uint32_t w[4];
            oracle_philox4x32(seed, row0 + r, (uint64_t)(j / 4), w);
            for (size_t t = 0; t < 4 && j + t < dim; ++t) {
                float u = (float)(w[t] >> 8) * (1.0f / 8388608.0f) - 1.0f; /* [-1, 1) */
                o[j + t] = u;
            }
        }
        for (size_t j = 0; j < dim; ++j) nsq += o[j] * o[j];
        float nrm = sqrtf(nsq);
        if (nrm > 0.0f) for (size_t j = 0; j < dim; ++j) o[j] /= nrm;
    }
}
/* Synthetic blob bytes: 16 bytes per Philox call, counter = (blob_id, byte_offset / 16). */
ORACLE_API void oracle_synth_bytes(uint64_t seed, uint64_t blob_id, uint64_t off0, size_t n,
                                   uint8_t* out) {
    for (size_t i = 0; i < n;) {
        uint64_t off = off0 + i;
        uint32_t w[4];
        oracle_philox4x32(seed, blob_id, off / 16, w);
        size_t b = (size_t)(off % 16);
        for (; b < 16 && i < n; ++b, ++i) out[i] = (uint8_t)(w[b / 4] >> (8 * (b % 4)));
    }
}

/* ================================================================================================
 * The reference's own synthetic-vector recipe (tests/benchmarks/vector_backend_engine_compare.cpp:
 * 83-107, 251-253; fixture tests/unit/vector/sqlite_vec_backend_comprehensive_catch2_test.cpp:
 * 84-101): std::mt19937 rng(seed); every component uniform_real_distribution<float>(-1, 1); then an
 * fp32 L2-normalisation with norm_sq accumulated in float; the corpus is drawn first, the queries
 * after it from the same stream.  Restated here so BASELINE config 1 (10k x 384, seed 42) can be
 * regenerated on the GPU box: MT19937 (Matsumoto & Nishimura 1998, the parameters std::mt19937 is
 * specified with, [rand.predef]) and libstdc++'s uniform_real_distribution<float> =
 * generate_canonical<float, 24>: one 32-bit draw, float(draw) * 2^-32, a result that rounds to
 * 1.0f is replaced by nextafter(1, 0); value = canonical * (b - a) + a.
 * Pinned by tests/golden/mt_recipe.json (made with the real std:: classes, oracle/ref_wrap.cpp).
 * ============================================================================================== */
typedef struct { uint32_t mt[624]; int idx; } mt19937_state;

static void mt19937_seed(mt19937_state* s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static uint32_t mt19937_next(mt19937_state* s) {
    if (s->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* `count` normalised vectors drawn consecutively from mt19937(seed) after skipping `skip_vectors`
 * vectors of the same dimension (the queries follow the corpus in the stream). */
ORACLE_API void oracle_mt19937_rows(uint32_t seed, size_t skip_vectors, size_t count, size_t dim, float* out) {
    mt19937_state st;
    mt19937_seed(&st, seed);
    for (size_t i = 0; i < skip_vectors * dim; ++i) (void)mt19937_next(&st);
    for (size_t r = 0; r < count; ++r) {
        float* v = out + r * dim;
        float norm_sq = 0.0f;
        for (size_t j = 0; j < dim; ++j) {
            float canon = (float)mt19937_next(&st) * 2.3283064365386963e-10f; /* / 2^32 (exact scaling) */
            if (canon >= 1.0f) canon = 0.99999994f;                            /* nextafterf(1, 0) */
            const float value = canon * 2.0f + -1.0f;                          /* * (b - a) + a */
            v[j] = value;
            norm_sq += value * value;
        }
        const float norm = sqrtf(norm_sq);
        if (norm > 0.0f)
            for (size_t j = 0; j < dim; ++j) v[j] /= norm;
    }
}
