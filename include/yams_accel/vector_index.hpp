// vector_index.hpp — the yams::vector query API over the vector_scan_v1 vtable.
//
// Mirrors the search half of IVectorStore (include/yams/vector/vector_store.h:23-77 in the
// reference): searchSimilar / searchSimilarBatch with the reference's argument meaning, result
// order and error behaviour, plus the CRUD subset needed to keep a device mirror of the
// `vectors` table (insertVector, insertVectorsBatch, deleteVector, getVectorCount).  The records
// stay on the host (chunk_id, document_hash, content, metadata are opaque to the scan); the
// embeddings live in HBM as one dense matrix per dimension, extended lazily after a mutation
// (appends + tombstones; generation idea of sqlite_vec_backend.cpp:389-411).
//
// A patched VectorDatabase::Impl flips ONE line to use it (vector_database.cpp:56 constructs
// SqliteVecBackend unconditionally); see INTEGRATION.md.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <map>
#include <memory>
#include <numeric>
#include <optional>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "l2_calibration.hpp"
#include "plugin.hpp"

#ifdef YAMS_ACCEL_USE_HOST_TYPES
#include <yams/vector/vector_store.h> // the host's own VectorRecord / VectorSearchDiagnostics / ... / IVectorStore
#endif

namespace yams::vector {

#ifndef YAMS_ACCEL_USE_HOST_TYPES
// restated for builds outside the YAMS tree (the subset of vector_types.h the scan path touches)
struct VectorRecord { // the fields of vector_types.h:104-138 the scan path reads or fills
    std::string chunk_id;
    std::string document_hash;
    std::vector<float> embedding;
    std::string content;
    std::map<std::string, std::string> metadata;
    float relevance_score = 0.0f;
    size_t embedding_dim = 0;
};

struct VectorSearchDiagnostics { // vector_types.h:181-204 (exact-scan subset)
    bool usedAnn = false;
    bool usedExactScan = false;
    bool collectVisitedDocumentHashes = false;
    bool rowsVisitedObserved = false;
    bool exactDistanceEvaluationsObserved = false;
    size_t rowsVisited = 0;
    size_t exactDistanceEvaluations = 0;
    size_t returnedRows = 0;
    std::unordered_set<std::string> visitedDocumentHashes;
};

struct VectorSearchParams { // vector_types.h:216-227
    size_t k = 10;
    float similarity_threshold = 0.7f;
    VectorSearchDiagnostics* diagnostics = nullptr;
};

enum class VectorSearchEngine { Vec0L2, ExactScan }; // vector_types.h:31-35 (the engines served here)

#endif

enum class ExactRowSelection { TopK, AllMatching }; // sqlite_vec_backend.cpp:342-371 / :4398-4400

// vec0 L2: the arithmetic of the host's distance, as settled by calibrateL2 (l2_calibration.hpp).  Uncalibrated: the
// plugin's configured "l2_accumulate" applies.  Calibrated and matched: the matching YAMS_SCAN_FLAG_L2_ACC_* goes with
// every L2 search.  Calibrated and NOT matched: L2 searches are refused (ErrorCode::NotSupported).
struct L2Setting {
    bool calibrated = false, matched = false;
    uint32_t flags = 0;
    std::string detail;
};

// One dense device matrix per AccelVectorIndex (one embedding dimension).  Row r of the mirror is
// records_[r]; rows are only ever APPENDED to the device mirror:
//   * insert of a new chunk_id      -> new row (corpus_append of the tail on the next search)
//   * insert of an existing chunk_id -> the old row becomes a tombstone, the new value a new row
//     (delete + insert with a new rowid inside one transaction, :1086-1226)
//   * deleteVector                  -> tombstone
// Tombstones are excluded through the row allow-mask of the scan; when they exceed a quarter of
// the mirror it is compacted and re-uploaded.  The chunk_id ranking (secondary sort key) is
// recomputed after every mutation batch.
class AccelVectorIndex {
public:
    AccelVectorIndex(std::shared_ptr<accel::Plugin> plugin, yams_vector_scan_v1* vt, size_t embeddingDim,
                     VectorSearchEngine engine = VectorSearchEngine::ExactScan)
        : plugin_(std::move(plugin)), vt_(vt), dim_(embeddingDim), engine_(engine) {}
    ~AccelVectorIndex() { if (corpus_) vt_->corpus_destroy(vt_->self, corpus_); }

    Result<void> initialize() {
        if (dim_ == 0) return Error{ErrorCode::InvalidArgument, "embedding_dim must be set"};
        const yams_status_t st = vt_->corpus_create(vt_->self, static_cast<uint32_t>(dim_), &corpus_);
        if (st != YAMS_OK) return Error{accel::mapStatus(st), "corpus_create failed"};
        initialized_ = true;
        return {};
    }
    bool isInitialized() const { return initialized_; }

    // validity on insert: size == dim and all finite (vector_database.cpp:1771-1784)
    Result<void> insertVector(const VectorRecord& record) { return insertVectorsBatch({record}); }
    Result<void> insertVectorsBatch(const std::vector<VectorRecord>& records) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        for (const auto& r : records) {
            if (r.embedding.size() != dim_) return Error{ErrorCode::InvalidArgument, "embedding dimension mismatch"};
            for (float v : r.embedding) if (!std::isfinite(v)) return Error{ErrorCode::InvalidArgument, "non-finite embedding"};
        }
        // chunk_ids repeated inside one batch: the last write wins (:1086-1226)
        std::unordered_map<std::string, size_t> last;
        for (size_t i = 0; i < records.size(); ++i) last[records[i].chunk_id] = i;
        for (size_t i = 0; i < records.size(); ++i) {
            const auto& r = records[i];
            if (last[r.chunk_id] != i) continue;
            auto it = byId_.find(r.chunk_id);
            if (it != byId_.end()) kill(it->second);
            // rows appended in ascending chunk_id order need no rank table: row order IS the
            // secondary sort key (the scan's default when no tie ranks are set)
            if (idOrdered_ && (records_.empty() || r.chunk_id > maxId_)) maxId_ = r.chunk_id;
            else idOrdered_ = false;
            byId_[r.chunk_id] = records_.size();
            byDoc_[r.document_hash].push_back(static_cast<uint32_t>(records_.size()));
            records_.push_back(r);
            alive_.push_back(1);
            zeroNorm_.push_back(isZeroNorm(r.embedding) ? 1 : 0);
        }
        ranksDirty_ = true;
        return {};
    }
    Result<void> deleteVector(const std::string& chunkId) {
        auto it = byId_.find(chunkId);
        if (it == byId_.end()) return Error{ErrorCode::NotFound, "chunk not found"};
        kill(it->second);
        byId_.erase(it);
        return {};
    }
    // updateVector replaces the row of chunk_id (it must exist, sqlite_vec_backend.cpp CRUD contract)
    Result<void> updateVector(const std::string& chunkId, const VectorRecord& record) {
        if (byId_.find(chunkId) == byId_.end()) return Error{ErrorCode::NotFound, "chunk not found"};
        VectorRecord r = record;
        r.chunk_id = chunkId;
        return insertVectorsBatch({r});
    }
    Result<void> deleteVectorsByDocument(const std::string& documentHash) {
        auto it = byDoc_.find(documentHash);
        if (it == byDoc_.end()) return {};
        for (uint32_t row : it->second)
            if (alive_[row]) (void)deleteVector(records_[row].chunk_id);
        return {};
    }
    Result<std::optional<VectorRecord>> getVector(const std::string& chunkId) const {
        auto it = byId_.find(chunkId);
        if (it == byId_.end()) return std::optional<VectorRecord>{};
        return std::optional<VectorRecord>{records_[it->second]};
    }
    Result<std::vector<VectorRecord>> getVectorsByDocument(const std::string& documentHash) const {
        std::vector<VectorRecord> out;
        if (auto it = byDoc_.find(documentHash); it != byDoc_.end())
            for (uint32_t r : it->second) if (alive_[r]) out.push_back(records_[r]);
        return out;
    }
    Result<bool> hasEmbedding(const std::string& documentHash) const {
        if (auto it = byDoc_.find(documentHash); it != byDoc_.end())
            for (uint32_t r : it->second) if (alive_[r]) return true;
        return false;
    }
    Result<size_t> getVectorCount() const { return records_.size() - dead_; }
    size_t mirrorRows() const { return records_.size(); }   // incl. tombstones (for tests)
    size_t uploadedRows() const { return deviceRows_; }

    Result<std::vector<VectorRecord>> searchSimilar(const std::vector<float>& query, size_t k,
                                                    float similarityThreshold = 0.0f,
                                                    VectorSearchDiagnostics* diagnostics = nullptr) {
        return searchSimilar(query, k, similarityThreshold, std::nullopt, {}, {}, diagnostics);
    }
    // The full IVectorStore::searchSimilar (vector_store.h:44-49).  Only rows whose document_hash
    // equals `document_hash` (if given) AND is in `candidate_hashes` (if non-empty) take part — the
    // SQL restriction of sqlite_vec_backend.cpp:4137-4175 — and, when `metadata_filters` is not
    // empty, whose metadata holds every (key, value) pair (:4350-4360): the host evaluates the
    // predicates on its records, the scan gets a row allow-mask, and the record-path row rules of
    // the reference apply (YAMS_SCAN_FLAG_RECORD_PATH).
    Result<std::vector<VectorRecord>>
    searchSimilar(const std::vector<float>& query, size_t k, float similarityThreshold,
                  const std::optional<std::string>& document_hash,
                  const std::unordered_set<std::string>& candidate_hashes,
                  const std::map<std::string, std::string>& metadata_filters = {},
                  VectorSearchDiagnostics* diagnostics = nullptr,
                  ExactRowSelection rowSelection = ExactRowSelection::TopK) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (auto sy = syncMirror(); !sy) return sy.error(); // may compact: row numbers change BEFORE masks are built
        const bool filtered = document_hash || !candidate_hashes.empty() || !metadata_filters.empty();
        if (!filtered && dead_ == 0 && rowSelection == ExactRowSelection::TopK) {
            auto r = searchSimilarBatchImpl({query}, k, similarityThreshold, diagnostics, nullptr, 0, 0, 0);
            if (!r) return r.error();
            return std::move(r.value().front());
        }
        std::vector<uint32_t> mask((records_.size() + 31) / 32, 0u);
        size_t visited = 0, evaluated = 0;
        // `r` passed the document restriction (the SQL WHERE of :4137-4175)
        auto consider = [&](size_t r) {
            if (!alive_[r]) return;
            const auto& rec = records_[r];
            ++visited; // rows the reference's statement steps over (:4336-4338)
            for (const auto& [key, value] : metadata_filters) {
                auto it = rec.metadata.find(key);
                if (it == rec.metadata.end() || it->second != value) return;
            }
            mask[r >> 5] |= 1u << (r & 31);
            if (!zeroNorm_[r]) ++evaluated; // rows that get a score on the record path (:4365-4371)
        };
        // document restrictions go through the document_hash index (the reference's statement is an
        // indexed lookup too), only a pure metadata filter walks every record
        auto considerDoc = [&](const std::string& h) {
            if (auto it = byDoc_.find(h); it != byDoc_.end())
                for (uint32_t r : it->second) consider(r);
        };
        if (document_hash) {
            if (candidate_hashes.empty() || candidate_hashes.count(*document_hash)) considerDoc(*document_hash);
        } else if (!candidate_hashes.empty()) {
            for (const auto& h : candidate_hashes) considerDoc(h);
        } else {
            for (size_t r = 0; r < records_.size(); ++r) consider(r);
        }
        if (mask.empty()) mask.push_back(0u);
        const bool recordPath = !metadata_filters.empty();
        size_t kk = k;
        if (rowSelection == ExactRowSelection::AllMatching) {
            kk = std::max<size_t>(visited, 1);
            if (kk > YAMS_SCAN_MAX_K) return allMatchingSliced(query, similarityThreshold, mask, recordPath, visited, evaluated, diagnostics);
        }
        if (diagnostics && diagnostics->collectVisitedDocumentHashes) // (:4189-4197)
            for (size_t r = 0; r < records_.size(); ++r)
                if ((mask[r >> 5] >> (r & 31)) & 1u) diagnostics->visitedDocumentHashes.insert(records_[r].document_hash);
        auto r = searchSimilarBatchImpl({query}, kk, similarityThreshold, diagnostics, mask.data(),
                                        recordPath ? YAMS_SCAN_FLAG_RECORD_PATH : 0u, visited,
                                        recordPath ? evaluated : visited);
        if (!r) return r.error();
        return std::move(r.value().front());
    }
    // Every live record, in mirror order (retrieval methods of a backend built on this index).
    template <typename Fn> void forEachRecord(Fn&& fn) const {
        for (size_t r = 0; r < records_.size(); ++r) if (alive_[r]) fn(records_[r]);
    }
    // num_threads is accepted and ignored, exactly like the reference (:1627)
    Result<std::vector<std::vector<VectorRecord>>>
    searchSimilarBatch(const std::vector<std::vector<float>>& queries, size_t k,
                       float similarityThreshold = 0.0f, size_t /*num_threads*/ = 0) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (auto sy = syncMirror(); !sy) return sy.error();
        if (dead_ == 0) return searchSimilarBatchImpl(queries, k, similarityThreshold, nullptr, nullptr, 0, 0, 0);
        std::vector<uint32_t> mask((records_.size() + 31) / 32, 0u);
        for (size_t r = 0; r < records_.size(); ++r) if (alive_[r]) mask[r >> 5] |= 1u << (r & 31);
        if (mask.empty()) mask.push_back(0u);
        return searchSimilarBatchImpl(queries, k, similarityThreshold, nullptr, mask.data(), 0, 0, 0);
    }

private:
    // ExactRowSelection::AllMatching over more rows than one device call returns (YAMS_SCAN_MAX_K):
    // the allowed rows are cut into slices of at most YAMS_SCAN_MAX_K, every slice returns ALL of its
    // matching rows, and the slices are merged with the reference's comparator (similarity desc,
    // chunk_id asc, sqlite_vec_backend.cpp:4218-4223) — exact, because nothing is ever dropped.
    Result<std::vector<VectorRecord>> allMatchingSliced(const std::vector<float>& query, float thr,
                                                        const std::vector<uint32_t>& mask, bool recordPath,
                                                        size_t visited, size_t evaluated,
                                                        VectorSearchDiagnostics* diagnostics) {
        std::vector<VectorRecord> all;
        std::vector<uint32_t> part(mask.size(), 0u);
        size_t inPart = 0;
        auto flush = [&]() -> Result<void> {
            if (inPart == 0) return {};
            auto r = searchSimilarBatchImpl({query}, inPart, thr, nullptr, part.data(),
                                            recordPath ? YAMS_SCAN_FLAG_RECORD_PATH : 0u, 0, 0);
            if (!r) return r.error();
            for (auto& rec : r.value().front()) all.push_back(std::move(rec));
            std::fill(part.begin(), part.end(), 0u);
            inPart = 0;
            return {};
        };
        for (size_t r = 0; r < records_.size(); ++r) {
            if (!((mask[r >> 5] >> (r & 31)) & 1u)) continue;
            part[r >> 5] |= 1u << (r & 31);
            if (++inPart == YAMS_SCAN_MAX_K)
                if (auto f = flush(); !f) return f.error();
        }
        if (auto f = flush(); !f) return f.error();
        std::stable_sort(all.begin(), all.end(), [](const VectorRecord& x, const VectorRecord& y) {
            if (x.relevance_score != y.relevance_score) return x.relevance_score > y.relevance_score;
            return x.chunk_id < y.chunk_id;
        });
        if (diagnostics) {
            diagnostics->usedExactScan = true; diagnostics->rowsVisitedObserved = true;
            diagnostics->exactDistanceEvaluationsObserved = true;
            diagnostics->rowsVisited += visited;
            diagnostics->exactDistanceEvaluations += recordPath ? evaluated : visited;
            diagnostics->returnedRows = all.size();
        }
        return all;
    }
    static bool isZeroNorm(const std::vector<float>& e) { // isZeroNormEmbedding, sqlite_vec_backend.cpp:204-211
        double n = 0.0;
        for (float v : e) n += static_cast<double>(v) * static_cast<double>(v);
        return n < 1e-10;
    }
    void kill(size_t row) {
        if (alive_[row]) { alive_[row] = 0; ++dead_; ranksDirty_ = true; }
    }
    Result<void> upload(size_t first) {
        const size_t n = records_.size() - first;
        if (n == 0) return {};
        std::vector<float> flat(n * dim_);
        for (size_t i = 0; i < n; ++i)
            std::copy(records_[first + i].embedding.begin(), records_[first + i].embedding.end(), flat.begin() + i * dim_);
        // Device memory exhausted (YAMS_ERR_RESOURCE_EXHAUSTED -> ErrorCode::ResourceExhausted, core/types.h:49): the device
        // mirror keeps the rows it had, the rows stay pending HERE, and every search fails with that code (a search over
        // part of the rows would not be the reference's answer) until an upload succeeds — the next search retries it.
        if (const yams_status_t st = vt_->corpus_append(vt_->self, corpus_, flat.data(), n); st != YAMS_OK)
            return Error{accel::mapStatus(st), st == YAMS_ERR_RESOURCE_EXHAUSTED ? "device memory exhausted while the mirror grew" : "corpus_append failed"};
        deviceRows_ = records_.size();
        return {};
    }
public:
    // Pending work for the device mirror (rows appended since the last upload, tombstones worth compacting, a stale
    // chunk_id ranking)?  Searches synchronise lazily; a host that serves concurrent searches under a shared lock
    // (vector_database.cpp:539,618) calls sync() under its exclusive lock first — with nothing pending a search only
    // reads this object.
    bool needsSync() const {
        return (dead_ > 1024 && dead_ * 4 > records_.size()) || records_.size() > deviceRows_ ||
               (ranksDirty_ && !records_.empty() && !idOrdered_);
    }
    Result<void> sync() { return syncMirror(); }
private:
    Result<void> syncMirror() {
        if (!needsSync()) return {};
        if (dead_ > 1024 && dead_ * 4 > records_.size()) { // compact: drop the tombstones, re-upload
            std::vector<VectorRecord> keep;
            std::vector<uint8_t> zn;
            keep.reserve(records_.size() - dead_);
            for (size_t r = 0; r < records_.size(); ++r)
                if (alive_[r]) { keep.push_back(std::move(records_[r])); zn.push_back(zeroNorm_[r]); }
            records_.swap(keep); zeroNorm_.swap(zn);
            alive_.assign(records_.size(), 1);
            dead_ = 0;
            byId_.clear(); byDoc_.clear();
            for (size_t i = 0; i < records_.size(); ++i) {
                byId_[records_[i].chunk_id] = i;
                byDoc_[records_[i].document_hash].push_back(static_cast<uint32_t>(i));
            }
            if (vt_->corpus_clear(vt_->self, corpus_) != YAMS_OK) return Error{ErrorCode::InternalError, "corpus_clear failed"};
            deviceRows_ = 0;
            ranksDirty_ = true;
        }
        const bool appended = records_.size() > deviceRows_;
        if (appended) if (auto u = upload(deviceRows_); !u) return u.error();
        if ((ranksDirty_ || appended) && !records_.empty() && !idOrdered_) {
            // secondary sort key = chunk_id string order (:4218-4223); tombstones keep a rank too
            const size_t n = records_.size();
            std::vector<uint32_t> order(n), rank(n);
            std::iota(order.begin(), order.end(), 0u);
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                const int c = records_[a].chunk_id.compare(records_[b].chunk_id);
                return c != 0 ? c < 0 : a < b; // a replaced chunk_id exists twice (old row is dead)
            });
            for (uint32_t r = 0; r < n; ++r) rank[order[r]] = r;
            if (vt_->corpus_set_tie_ranks(vt_->self, corpus_, rank.data(), n) != YAMS_OK)
                return Error{ErrorCode::InternalError, "corpus_set_tie_ranks failed"};
        }
        ranksDirty_ = false;
        return {};
    }

    // visited / evaluated: per-query diagnostics of a filtered search computed by the caller (0, 0 =
    // take them from the scan)
    Result<std::vector<std::vector<VectorRecord>>>
    searchSimilarBatchImpl(const std::vector<std::vector<float>>& queries, size_t k, float thr,
                           VectorSearchDiagnostics* diagnostics, const uint32_t* rowMask,
                           uint32_t flags, size_t visited, size_t evaluated) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (queries.empty()) return std::vector<std::vector<VectorRecord>>{};
        for (const auto& q : queries) // vector_database.cpp:545-550, 626-633
            if (q.size() != dim_)
                return Error{ErrorCode::InvalidArgument, "Query embedding dimension mismatch (expected=" +
                                                             std::to_string(dim_) + ", got=" + std::to_string(q.size()) + ")"};
        if (auto s = syncMirror(); !s) return s.error();
        if (engine_ == VectorSearchEngine::Vec0L2 && !l2_.calibrated) {
            // Served in the plugin's CONFIGURED arithmetic ("l2_accumulate", fp64 unless set) — which the host's vec0 build may
            // not use (a reference built with '-mavx', '-mfma' calibrates as f32x8_fma): top-k sets can then differ from the
            // host's own at the cut on a fraction of a percent of queries.  Said once per index, counted always.
            ++l2UncalibratedSearches_;
            if (!l2Warned_) {
                l2Warned_ = true;
                std::fprintf(stderr, "[yams_mi355x_accel] vec0 L2 index (dim %zu) is searched WITHOUT calibration: the plugin's configured "
                                     "\"l2_accumulate\" is used as is; call calibrateL2(host distance function) or setL2() to pin the arithmetic\n", dim_);
            }
        }
        if (engine_ == VectorSearchEngine::Vec0L2 && l2_.calibrated) {
            // a top-k set computed in arithmetic the host's vec0 does not use would differ from the host's own on a fraction
            // of queries: refuse rather than serve it
            if (!l2_.matched) return Error{ErrorCode::NotSupported, "vec0 L2 refused: " + l2_.detail};
            flags |= l2_.flags;
        }
        if (k > YAMS_SCAN_MAX_K) return searchPeeled(queries, k, thr, diagnostics, rowMask, flags, visited, evaluated);
        std::vector<float> flat(queries.size() * dim_);
        for (size_t i = 0; i < queries.size(); ++i) std::copy(queries[i].begin(), queries[i].end(), flat.begin() + i * dim_);
        yams_scan_hit_t* hits = nullptr; uint32_t* counts = nullptr; yams_scan_diag_t diag{};
        const uint32_t metric = engine_ == VectorSearchEngine::Vec0L2 ? YAMS_SCAN_L2 : YAMS_SCAN_COSINE;
        yams_status_t st;
        if (flags != 0) {
            if (!vt_->search_batch_ex) return Error{ErrorCode::NotImplemented, "plugin lacks search_batch_ex"};
            st = vt_->search_batch_ex(vt_->self, corpus_, flat.data(), static_cast<uint32_t>(queries.size()),
                                      static_cast<uint32_t>(dim_), static_cast<uint32_t>(k), thr, metric, flags,
                                      rowMask, &hits, &counts, &diag);
        } else {
            st = vt_->search_batch_masked(vt_->self, corpus_, flat.data(), static_cast<uint32_t>(queries.size()),
                                          static_cast<uint32_t>(dim_), static_cast<uint32_t>(k), thr, metric,
                                          rowMask, &hits, &counts, &diag);
        }
        if (st == YAMS_ERR_INVALID_ARG)
            return Error{ErrorCode::InvalidArgument, "Exact vector search requires a finite, non-zero query embedding"};
        if (st != YAMS_OK) return Error{accel::mapStatus(st), "vector scan failed"};
        std::vector<std::vector<VectorRecord>> out(queries.size());
        for (size_t q = 0; q < queries.size(); ++q)
            for (uint32_t i = 0; i < counts[q]; ++i) {
                const auto& h = hits[q * k + i];
                VectorRecord rec = records_[static_cast<size_t>(h.row)];
                rec.relevance_score = h.similarity; // :4323-4326
                rec.embedding_dim = dim_;
                out[q].push_back(std::move(rec));
            }
        vt_->free_hits(vt_->self, hits, counts);
        if (diagnostics) {
            diagnostics->usedExactScan = true; diagnostics->rowsVisitedObserved = true;
            diagnostics->exactDistanceEvaluationsObserved = true;
            const bool own = rowMask != nullptr && (visited != 0 || evaluated != 0 || flags != 0);
            diagnostics->rowsVisited += own ? visited * queries.size() : diag.rows_visited;
            diagnostics->exactDistanceEvaluations += own ? evaluated * queries.size() : diag.exact_distance_evaluations;
            diagnostics->returnedRows = diag.returned_rows;
        }
        return out;
    }

public:
    // ---- the product-quantised engine (VectorSearchEngine::SimeonPqAdc — the product's default, vector_types.h:80) ---------
    // SqliteVecBackend::Impl::simeonPqSearchUnlocked (sqlite_vec_backend.cpp:3868-4056) with its scan, selection and re-rank
    // on the device.  The host keeps simeon (training, encoding, the per-query table).  After every (re)build of its
    // SimeonPqIndexState for this dimension it hands over the codes and, per indexed row, the chunk id (rowids[i] resolved):
    // the tie-break keys (stableStringKey, :141-148, :3337) and the index -> mirror-row table are derived here.  Rows the
    // mirror does not (or no longer) hold are skipped by the search, as :4010-4012 does.
    static uint64_t stableStringKey(const std::string& s) { // FNV-1a 64 (:141-148)
        uint64_t h = 1469598103934665603ULL;
        for (const unsigned char b : s) { h ^= b; h *= 1099511628211ULL; }
        return h;
    }
    Result<void> setPqIndex(const std::vector<uint8_t>& codes, size_t m, const std::vector<std::string>& chunkIdOfIndex) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (!vt_->pq_index_set) return Error{ErrorCode::NotImplemented, "plugin lacks pq_index_set (vector_scan_v1 version 2)"};
        const size_t n = chunkIdOfIndex.size();
        if (m == 0 || codes.size() != n * m) return Error{ErrorCode::InvalidArgument, "codes must hold m bytes per indexed row"};
        if (auto sy = syncMirror(); !sy) return sy.error();   // (may compact: row numbers are final after it)
        std::vector<uint64_t> keys(n);
        std::vector<uint32_t> rowOf(n);
        for (size_t i = 0; i < n; ++i) {
            keys[i] = stableStringKey(chunkIdOfIndex[i]);
            const auto it = byId_.find(chunkIdOfIndex[i]);
            rowOf[i] = it == byId_.end() ? 0xffffffffu : static_cast<uint32_t>(it->second);
        }
        pqChunkIds_ = chunkIdOfIndex; pqCodes_ = codes; pqM_ = m; pqRows_ = records_.size();
        const yams_status_t st = vt_->pq_index_set(vt_->self, corpus_, codes.data(), n, static_cast<uint32_t>(m), keys.data(), rowOf.data());
        if (st != YAMS_OK) return Error{accel::mapStatus(st), "pq_index_set failed"};
        return {};
    }
    // queries: RAW query embeddings; luts[q]: m x 256 floats, simeon::PQInnerProductQuery's table for the NORMALISED query
    // (:3895-3901); candidateIndices (nullable): ascending indices into the PQ index (:3910-3937); sumFlags: YAMS_PQ_SUM_*.
    Result<std::vector<std::vector<VectorRecord>>>
    searchPqBatch(const std::vector<std::vector<float>>& queries, const std::vector<std::vector<float>>& luts, size_t k, float thr,
                  size_t rerankFactor, const std::vector<uint32_t>* candidateIndices = nullptr, uint32_t sumFlags = YAMS_PQ_SUM_SEQUENTIAL,
                  VectorSearchDiagnostics* diagnostics = nullptr) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (!vt_->search_pq) return Error{ErrorCode::NotImplemented, "plugin lacks search_pq (vector_scan_v1 version 2)"};
        if (queries.empty()) return std::vector<std::vector<VectorRecord>>{};
        if (luts.size() != queries.size()) return Error{ErrorCode::InvalidArgument, "one table per query"};
        for (size_t i = 0; i < queries.size(); ++i)
            if (queries[i].size() != dim_ || luts[i].size() != pqM_ * 256)
                return Error{ErrorCode::InvalidArgument, "query / table size mismatch"};
        if (k > YAMS_SCAN_MAX_K) return Error{ErrorCode::NotSupported, "k exceeds YAMS_SCAN_MAX_K for the product-quantised engine"};
        if (auto sy = syncMirror(); !sy) return sy.error();
        if (records_.size() != pqRows_ && !pqChunkIds_.empty()) { // the mirror was compacted or grew: the row table is re-derived
            if (auto r = setPqIndex(pqCodes_, pqM_, pqChunkIds_); !r) return r.error();
        }
        std::vector<float> fq(queries.size() * dim_), fl(queries.size() * pqM_ * 256);
        for (size_t i = 0; i < queries.size(); ++i) {
            std::copy(queries[i].begin(), queries[i].end(), fq.begin() + i * dim_);
            std::copy(luts[i].begin(), luts[i].end(), fl.begin() + i * pqM_ * 256);
        }
        yams_scan_hit_t* hits = nullptr; uint32_t* counts = nullptr; yams_scan_diag_t diag{};
        static const uint32_t kNoIndex = 0;     // (an EMPTY candidate list is a list — nothing is searched —, not "no restriction")
        const uint32_t* cand = candidateIndices ? (candidateIndices->empty() ? &kNoIndex : candidateIndices->data()) : nullptr;
        const yams_status_t st = vt_->search_pq(vt_->self, corpus_, fq.data(), fl.data(), static_cast<uint32_t>(queries.size()),
                                                static_cast<uint32_t>(dim_), static_cast<uint32_t>(k), thr, static_cast<uint32_t>(rerankFactor),
                                                sumFlags, cand, candidateIndices ? candidateIndices->size() : 0, &hits, &counts, &diag);
        if (st != YAMS_OK) return Error{accel::mapStatus(st), "product-quantised search failed"};
        std::vector<std::vector<VectorRecord>> out(queries.size());
        for (size_t q = 0; q < queries.size(); ++q)
            for (uint32_t i = 0; i < counts[q]; ++i) {
                const auto& h = hits[q * k + i];
                VectorRecord rec = records_[static_cast<size_t>(h.row)];
                if (!alive_[static_cast<size_t>(h.row)]) continue;      // (a tombstone the PQ index still names: the table lost the row)
                rec.relevance_score = h.similarity; // :4039
                rec.embedding_dim = dim_;
                out[q].push_back(std::move(rec));
            }
        vt_->free_hits(vt_->self, hits, counts);
        if (diagnostics) {      // :3938-3945, :4027
            diagnostics->usedAnn = true; diagnostics->rowsVisitedObserved = true; diagnostics->exactDistanceEvaluationsObserved = true;
            diagnostics->rowsVisited += diag.rows_visited; diagnostics->exactDistanceEvaluations += diag.exact_distance_evaluations;
            diagnostics->returnedRows = diag.returned_rows;
        }
        return out;
    }

private:
    std::vector<std::string> pqChunkIds_; std::vector<uint8_t> pqCodes_; size_t pqM_ = 0, pqRows_ = 0;

    // k above what one device call returns (YAMS_SCAN_MAX_K): the reference takes any k
    // (sqlite_vec_backend.cpp:4299-4303 keeps a heap of k), callers that over-fetch for fusion or re-ranking
    // use thousands.  Per query, rounds of at most YAMS_SCAN_MAX_K: every round excludes the rows already
    // returned through the allow-mask, so it yields exactly the next best rows in the reference's order
    // (similarity desc, chunk_id asc) — the concatenation IS the top k.  The vec0 engine ranks by distance and
    // applies the similarity threshold after the cut (:4506-4510): its rounds defer the threshold.
    Result<std::vector<std::vector<VectorRecord>>>
    searchPeeled(const std::vector<std::vector<float>>& queries, size_t k, float thr, VectorSearchDiagnostics* diagnostics,
                 const uint32_t* rowMask, uint32_t flags, size_t visited, size_t evaluated) {
        const bool l2 = engine_ == VectorSearchEngine::Vec0L2;
        const size_t words = std::max<size_t>((records_.size() + 31) / 32, 1);
        std::vector<std::vector<VectorRecord>> out(queries.size());
        for (size_t qi = 0; qi < queries.size(); ++qi) {
            std::vector<uint32_t> mask(words, 0u);
            if (rowMask) std::copy(rowMask, rowMask + words, mask.begin());
            else for (size_t r = 0; r < records_.size(); ++r) if (alive_[r]) mask[r >> 5] |= 1u << (r & 31);
            size_t remaining = k;
            bool first = true;
            while (remaining > 0) {
                const size_t kk = std::min<size_t>(remaining, YAMS_SCAN_MAX_K);
                auto r = searchSimilarBatchImpl({queries[qi]}, kk, thr, first ? diagnostics : nullptr, mask.data(),
                                                flags | (l2 ? YAMS_SCAN_FLAG_DEFER_THRESHOLD : 0u), visited, evaluated);
                if (!r) return r.error();
                auto& got = r.value().front();
                const size_t n = got.size();
                for (auto& rec : got) {
                    const size_t row = byId_.at(rec.chunk_id);
                    mask[row >> 5] &= ~(1u << (row & 31));
                    out[qi].push_back(std::move(rec));
                }
                first = false;
                if (n < kk) break; // nothing left (above the threshold)
                remaining -= n;
            }
            if (l2) out[qi].erase(std::remove_if(out[qi].begin(), out[qi].end(), [&](const VectorRecord& x) { return x.relevance_score < thr; }),
                                  out[qi].end());
        }
        if (diagnostics) diagnostics->returnedRows = out.empty() ? 0 : out.back().size();
        return out;
    }

public:
    // Ask the HOST's own vec0 distance which arithmetic it uses on vectors of THIS index's dimension (l2_calibration.hpp): on
    // a match every later L2 search of this index carries that YAMS_SCAN_FLAG_L2_ACC_*; on no match L2 searches fail with
    // NotSupported.  Returns the calibration record either way (Error only when no function was given).
    Result<accel_l2::L2Calibration> calibrateL2(const accel_l2::L2DistanceFn& fn) {
        if (!fn) return Error{ErrorCode::InvalidArgument, "calibrateL2 needs the host's L2 distance function"};
        const auto c = accel_l2::calibrateL2(fn, dim_);
        setL2(L2Setting{true, c.matched, c.flags, c.detail});
        return c;
    }
    void setL2(const L2Setting& s) { l2_ = s; }
    const L2Setting& l2() const { return l2_; }
    // L2 search calls answered in an arithmetic nobody confirmed (no calibrateL2 / setL2 before them)
    uint64_t l2UncalibratedSearches() const { return l2UncalibratedSearches_; }

private:
    std::shared_ptr<accel::Plugin> plugin_;
    yams_vector_scan_v1* vt_;
    size_t dim_;
    VectorSearchEngine engine_;
    L2Setting l2_;
    uint64_t l2UncalibratedSearches_ = 0;
    bool l2Warned_ = false;
    uint64_t corpus_ = 0;
    bool initialized_ = false, ranksDirty_ = false;
    std::vector<VectorRecord> records_;
    std::vector<uint8_t> alive_, zeroNorm_;
    size_t dead_ = 0, deviceRows_ = 0;
    std::unordered_map<std::string, size_t> byId_;
    std::unordered_map<std::string, std::vector<uint32_t>> byDoc_; // document_hash -> rows (incl. tombstones)
    bool idOrdered_ = true;   // every row was appended with a chunk_id above all earlier ones
    std::string maxId_;
};

inline Result<std::unique_ptr<AccelVectorIndex>> createAccelVectorIndex(std::shared_ptr<accel::Plugin> plugin, size_t dim,
                                                                        VectorSearchEngine engine = VectorSearchEngine::ExactScan) {
    auto vt = plugin->getInterface<yams_vector_scan_v1>(YAMS_IFACE_VECTOR_SCAN_V1, YAMS_IFACE_VECTOR_SCAN_V1_VERSION);
    if (!vt) return vt.error();
    return std::make_unique<AccelVectorIndex>(std::move(plugin), vt.value(), dim, engine);
}

// The `vectors` table holds rows of ANY dimension; a search only sees the rows whose embedding_dim
// equals the query's (`WHERE embedding_dim = ?`, sqlite_vec_backend.cpp:4147).  One dense mirror per
// dimension, created on first use; a query of a dimension that has no rows returns no results.
class AccelVectorTable {
public:
    explicit AccelVectorTable(std::shared_ptr<accel::Plugin> plugin,
                              VectorSearchEngine engine = VectorSearchEngine::ExactScan)
        : plugin_(std::move(plugin)), engine_(engine) {}

    Result<void> insertVectorsBatch(const std::vector<VectorRecord>& records) {
        std::map<size_t, std::vector<VectorRecord>> byDim;
        for (const auto& r : records) {
            if (r.embedding.empty()) return Error{ErrorCode::InvalidArgument, "empty embedding"};
            byDim[r.embedding.size()].push_back(r);
        }
        for (auto& [dim, recs] : byDim) {
            // a chunk_id lives in one dimension only: re-inserting it with another size moves it
            for (const auto& r : recs) {
                auto it = dimOf_.find(r.chunk_id);
                if (it != dimOf_.end() && it->second != dim) (void)byDim_[it->second]->deleteVector(r.chunk_id);
                dimOf_[r.chunk_id] = dim;
            }
            auto idx = indexFor(dim);
            if (!idx) return idx.error();
            if (auto s = idx.value()->insertVectorsBatch(recs); !s) return s;
        }
        return {};
    }
    Result<void> insertVector(const VectorRecord& record) { return insertVectorsBatch({record}); }
    // The product-quantised engine per dimension (AccelVectorIndex::setPqIndex / searchPqBatch): the host's
    // SimeonPqIndexState of `dim` (simeon_pq_indices_[dim], sqlite_vec_backend.cpp:3877-3880)
    Result<void> setPqIndex(size_t dim, const std::vector<uint8_t>& codes, size_t m, const std::vector<std::string>& chunkIdOfIndex) {
        auto idx = indexFor(dim);
        if (!idx) return idx.error();
        return idx.value()->setPqIndex(codes, m, chunkIdOfIndex);
    }
    Result<std::vector<std::vector<VectorRecord>>>
    searchPqBatch(const std::vector<std::vector<float>>& queries, const std::vector<std::vector<float>>& luts, size_t k, float thr,
                  size_t rerankFactor, const std::vector<uint32_t>* candidateIndices = nullptr, uint32_t sumFlags = YAMS_PQ_SUM_SEQUENTIAL,
                  VectorSearchDiagnostics* diagnostics = nullptr) {
        if (queries.empty()) return std::vector<std::vector<VectorRecord>>{};
        const auto it = byDim_.find(queries.front().size());
        if (it == byDim_.end()) return std::vector<std::vector<VectorRecord>>(queries.size());   // no index of this dimension (:3877-3880)
        return it->second->searchPqBatch(queries, luts, k, thr, rerankFactor, candidateIndices, sumFlags, diagnostics);
    }
    Result<void> deleteVector(const std::string& chunkId) {
        auto it = dimOf_.find(chunkId);
        if (it == dimOf_.end()) return Error{ErrorCode::NotFound, "chunk not found"};
        auto s = byDim_[it->second]->deleteVector(chunkId);
        dimOf_.erase(it);
        return s;
    }
    Result<size_t> getVectorCount() const {
        size_t n = 0;
        for (const auto& [dim, idx] : byDim_) n += idx->getVectorCount().value();
        return n;
    }
    Result<std::vector<VectorRecord>>
    searchSimilar(const std::vector<float>& query, size_t k, float similarityThreshold = 0.0f,
                  const std::optional<std::string>& document_hash = std::nullopt,
                  const std::unordered_set<std::string>& candidate_hashes = {},
                  const std::map<std::string, std::string>& metadata_filters = {},
                  VectorSearchDiagnostics* diagnostics = nullptr) {
        auto it = byDim_.find(query.size());
        if (it == byDim_.end()) { // no row of this dimension: the statement steps over nothing
            if (query.empty() || k == 0) return std::vector<VectorRecord>{};
            for (float v : query)
                if (!std::isfinite(v)) return Error{ErrorCode::InvalidArgument, "Exact vector search requires a finite, non-zero query embedding"};
            if (diagnostics) { diagnostics->usedExactScan = true; diagnostics->rowsVisitedObserved = true; }
            return std::vector<VectorRecord>{};
        }
        return it->second->searchSimilar(query, k, similarityThreshold, document_hash, candidate_hashes,
                                         metadata_filters, diagnostics);
    }

    // all queries of a batch share one dimension (:1619-1626); num_threads is ignored (:1627)
    Result<std::vector<std::vector<VectorRecord>>>
    searchSimilarBatch(const std::vector<std::vector<float>>& queries, size_t k, float similarityThreshold = 0.0f,
                       size_t num_threads = 0) {
        if (queries.empty()) return std::vector<std::vector<VectorRecord>>{};
        const size_t dim = queries.front().size();
        for (const auto& q : queries)
            if (q.size() != dim) return Error{ErrorCode::InvalidArgument, "All query embeddings in a batch must share one dimension"};
        auto it = byDim_.find(dim);
        if (it == byDim_.end()) {
            std::vector<std::vector<VectorRecord>> out;
            for (const auto& q : queries) {
                auto one = searchSimilar(q, k, similarityThreshold);
                if (!one) return one.error();
                out.push_back(std::move(one.value()));
            }
            return out;
        }
        return it->second->searchSimilarBatch(queries, k, similarityThreshold, num_threads);
    }
    Result<std::vector<VectorRecord>>
    searchSimilarRows(const std::vector<float>& query, size_t k, float similarityThreshold,
                      const std::unordered_set<std::string>& candidate_hashes, VectorSearchDiagnostics* diagnostics,
                      ExactRowSelection selection) {
        auto it = byDim_.find(query.size());
        if (it == byDim_.end()) return searchSimilar(query, selection == ExactRowSelection::AllMatching ? 1 : k,
                                                     similarityThreshold, std::nullopt, candidate_hashes, {}, diagnostics);
        return it->second->searchSimilar(query, k, similarityThreshold, std::nullopt, candidate_hashes, {}, diagnostics, selection);
    }
    template <typename Fn> void forEachRecord(Fn&& fn) const {
        for (const auto& [dim, idx] : byDim_) idx->forEachRecord(fn);
    }
    bool needsSync() const {
        for (const auto& [dim, idx] : byDim_) if (idx->needsSync()) return true;
        return false;
    }
    Result<void> sync() {
        for (auto& [dim, idx] : byDim_) if (auto s = idx->sync(); !s) return s;
        return {};
    }
    // Drops every mirror (the device memory is parked by the plugin for the next corpus): what a re-warm from the
    // durable store starts from.
    void clear() { byDim_.clear(); dimOf_.clear(); }
    Result<std::optional<VectorRecord>> getVector(const std::string& chunkId) const {
        auto it = dimOf_.find(chunkId);
        if (it == dimOf_.end()) return std::optional<VectorRecord>{};
        return byDim_.at(it->second)->getVector(chunkId);
    }
    Result<void> deleteVectorsByDocument(const std::string& documentHash) {
        for (auto& [dim, idx] : byDim_) {
            auto recs = idx->getVectorsByDocument(documentHash);
            if (recs) for (const auto& r : recs.value()) dimOf_.erase(r.chunk_id);
            if (auto s = idx->deleteVectorsByDocument(documentHash); !s) return s;
        }
        return {};
    }

private:
    Result<AccelVectorIndex*> indexFor(size_t dim) {
        auto it = byDim_.find(dim);
        if (it != byDim_.end()) return it->second.get();
        auto made = createAccelVectorIndex(plugin_, dim, engine_);
        if (!made) return made.error();
        if (auto s = made.value()->initialize(); !s) return s.error();
        if (l2Fn_) (void)made.value()->calibrateL2(l2Fn_); // every dimension is calibrated on its own probes
        auto* raw = made.value().get();
        byDim_[dim] = std::move(made.value());
        return raw;
    }
public:
    // One function for the table: every index (dimension), present and future, is calibrated with it on probes of its
    // own dimension and searches L2 in the arithmetic the host's function shows there (or refuses).  Returns the
    // calibration at `dim` (the table's main dimension; 0: the largest one present, else 768) for the log.
    Result<accel_l2::L2Calibration> calibrateL2(const accel_l2::L2DistanceFn& fn, size_t dim = 0) {
        if (!fn) return Error{ErrorCode::InvalidArgument, "calibrateL2 needs the host's L2 distance function"};
        l2Fn_ = fn;
        for (auto& kv : byDim_) (void)kv.second->calibrateL2(fn);
        if (dim == 0) dim = byDim_.empty() ? 768 : byDim_.rbegin()->first;
        auto it = byDim_.find(dim);
        l2_ = it != byDim_.end() ? it->second->l2() : [&] { const auto c = accel_l2::calibrateL2(fn, dim); return L2Setting{true, c.matched, c.flags, c.detail}; }();
        return accel_l2::calibrateL2(fn, dim);
    }
    const L2Setting& l2() const { return l2_; }   // (of the dimension the last calibrateL2 call reported)

private:
    std::shared_ptr<accel::Plugin> plugin_;
    VectorSearchEngine engine_;
    L2Setting l2_;
    accel_l2::L2DistanceFn l2Fn_;
    std::map<size_t, std::unique_ptr<AccelVectorIndex>> byDim_;
    std::unordered_map<std::string, size_t> dimOf_;
};

} // namespace yams::vector
