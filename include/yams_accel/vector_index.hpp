// vector_index.hpp — the yams::vector query API over the vector_scan_v1 vtable.
//
// Mirrors the search half of IVectorStore (include/yams/vector/vector_store.h:23-77 in the
// reference): searchSimilar / searchSimilarBatch with the reference's argument meaning, result
// order and error behaviour, plus the CRUD subset needed to keep a device mirror of the
// `vectors` table (insertVector, insertVectorsBatch, deleteVector, getVectorCount).  The records
// stay on the host (chunk_id, document_hash, content, metadata are opaque to the scan); the
// embeddings live in HBM as one dense matrix per dimension, rebuilt lazily after a mutation
// (generation counter, the idea of sqlite_vec_backend.cpp:389-411).
//
// A patched VectorDatabase::Impl flips ONE line to use it (vector_database.cpp:56 constructs
// SqliteVecBackend unconditionally); see INTEGRATION.md.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <numeric>
#include <optional>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "plugin.hpp"

namespace yams::vector {

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
    bool rowsVisitedObserved = false;
    bool exactDistanceEvaluationsObserved = false;
    size_t rowsVisited = 0;
    size_t exactDistanceEvaluations = 0;
    size_t returnedRows = 0;
};

struct VectorSearchParams { // vector_types.h:216-227
    size_t k = 10;
    float similarity_threshold = 0.7f;
    VectorSearchDiagnostics* diagnostics = nullptr;
};

enum class VectorSearchEngine { Vec0L2, ExactScan }; // vector_types.h:31-35 (the engines served here)

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
        for (const auto& r : records) { // an existing chunk_id is replaced (delete + insert, :1086-1226)
            auto it = byId_.find(r.chunk_id);
            if (it != byId_.end()) { records_[it->second] = r; }
            else { byId_[r.chunk_id] = records_.size(); records_.push_back(r); }
        }
        dirty_ = true;
        return {};
    }
    Result<void> deleteVector(const std::string& chunkId) {
        auto it = byId_.find(chunkId);
        if (it == byId_.end()) return Error{ErrorCode::NotFound, "chunk not found"};
        records_.erase(records_.begin() + static_cast<std::ptrdiff_t>(it->second));
        byId_.clear();
        for (size_t i = 0; i < records_.size(); ++i) byId_[records_[i].chunk_id] = i;
        dirty_ = true;
        return {};
    }
    Result<size_t> getVectorCount() const { return records_.size(); }

    Result<std::vector<VectorRecord>> searchSimilar(const std::vector<float>& query, size_t k,
                                                    float similarityThreshold = 0.0f,
                                                    VectorSearchDiagnostics* diagnostics = nullptr) {
        auto r = searchSimilarBatchImpl({query}, k, similarityThreshold, diagnostics, nullptr);
        if (!r) return r.error();
        return std::move(r.value().front());
    }
    // The filtered form of IVectorStore::searchSimilar (vector_store.h:44-49): only rows whose
    // document_hash equals `document_hash` (if given) AND is in `candidate_hashes` (if non-empty)
    // take part — the SQL restriction of sqlite_vec_backend.cpp:4137-4175, as a row allow-mask.
    // (metadata_filters need the parsed record per row and stay with the SQLite backend.)
    Result<std::vector<VectorRecord>>
    searchSimilar(const std::vector<float>& query, size_t k, float similarityThreshold,
                  const std::optional<std::string>& document_hash,
                  const std::unordered_set<std::string>& candidate_hashes,
                  VectorSearchDiagnostics* diagnostics = nullptr) {
        if (!document_hash && candidate_hashes.empty()) return searchSimilar(query, k, similarityThreshold, diagnostics);
        std::vector<uint32_t> mask((records_.size() + 31) / 32, 0u);
        for (size_t r = 0; r < records_.size(); ++r) {
            const auto& h = records_[r].document_hash;
            if (document_hash && h != *document_hash) continue;
            if (!candidate_hashes.empty() && !candidate_hashes.count(h)) continue;
            mask[r >> 5] |= 1u << (r & 31);
        }
        if (mask.empty()) mask.push_back(0u);
        auto r = searchSimilarBatchImpl({query}, k, similarityThreshold, diagnostics, mask.data());
        if (!r) return r.error();
        return std::move(r.value().front());
    }
    // num_threads is accepted and ignored, exactly like the reference (:1627)
    Result<std::vector<std::vector<VectorRecord>>>
    searchSimilarBatch(const std::vector<std::vector<float>>& queries, size_t k,
                       float similarityThreshold = 0.0f, size_t /*num_threads*/ = 0) {
        return searchSimilarBatchImpl(queries, k, similarityThreshold, nullptr, nullptr);
    }

private:
    Result<void> syncMirror() {
        if (!dirty_) return {};
        if (vt_->corpus_clear(vt_->self, corpus_) != YAMS_OK) return Error{ErrorCode::InternalError, "corpus_clear failed"};
        const size_t n = records_.size();
        if (n) {
            std::vector<float> flat(n * dim_);
            for (size_t i = 0; i < n; ++i) std::copy(records_[i].embedding.begin(), records_[i].embedding.end(), flat.begin() + i * dim_);
            if (vt_->corpus_append(vt_->self, corpus_, flat.data(), n) != YAMS_OK) return Error{ErrorCode::InternalError, "corpus_append failed"};
            // secondary sort key = chunk_id string order (:4218-4223)
            std::vector<uint32_t> order(n), rank(n);
            std::iota(order.begin(), order.end(), 0u);
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return records_[a].chunk_id < records_[b].chunk_id; });
            for (uint32_t r = 0; r < n; ++r) rank[order[r]] = r;
            if (vt_->corpus_set_tie_ranks(vt_->self, corpus_, rank.data(), n) != YAMS_OK) return Error{ErrorCode::InternalError, "corpus_set_tie_ranks failed"};
        }
        dirty_ = false;
        return {};
    }

    Result<std::vector<std::vector<VectorRecord>>>
    searchSimilarBatchImpl(const std::vector<std::vector<float>>& queries, size_t k, float thr,
                           VectorSearchDiagnostics* diagnostics, const uint32_t* rowMask) {
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (queries.empty()) return std::vector<std::vector<VectorRecord>>{};
        for (const auto& q : queries) // vector_database.cpp:545-550, 626-633
            if (q.size() != dim_)
                return Error{ErrorCode::InvalidArgument, "Query embedding dimension mismatch (expected=" +
                                                             std::to_string(dim_) + ", got=" + std::to_string(q.size()) + ")"};
        if (auto s = syncMirror(); !s) return s.error();
        std::vector<float> flat(queries.size() * dim_);
        for (size_t i = 0; i < queries.size(); ++i) std::copy(queries[i].begin(), queries[i].end(), flat.begin() + i * dim_);
        yams_scan_hit_t* hits = nullptr; uint32_t* counts = nullptr; yams_scan_diag_t diag{};
        const uint32_t metric = engine_ == VectorSearchEngine::Vec0L2 ? YAMS_SCAN_L2 : YAMS_SCAN_COSINE;
        const yams_status_t st = vt_->search_batch_masked(vt_->self, corpus_, flat.data(), static_cast<uint32_t>(queries.size()),
                                                          static_cast<uint32_t>(dim_), static_cast<uint32_t>(k), thr, metric,
                                                          rowMask, &hits, &counts, &diag);
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
            diagnostics->rowsVisited += diag.rows_visited;
            diagnostics->exactDistanceEvaluations += diag.exact_distance_evaluations;
            diagnostics->returnedRows = diag.returned_rows;
        }
        return out;
    }

    std::shared_ptr<accel::Plugin> plugin_;
    yams_vector_scan_v1* vt_;
    size_t dim_;
    VectorSearchEngine engine_;
    uint64_t corpus_ = 0;
    bool initialized_ = false, dirty_ = false;
    std::vector<VectorRecord> records_;
    std::unordered_map<std::string, size_t> byId_;
};

inline Result<std::unique_ptr<AccelVectorIndex>> createAccelVectorIndex(std::shared_ptr<accel::Plugin> plugin, size_t dim,
                                                                        VectorSearchEngine engine = VectorSearchEngine::ExactScan) {
    auto vt = plugin->getInterface<yams_vector_scan_v1>(YAMS_IFACE_VECTOR_SCAN_V1, YAMS_IFACE_VECTOR_SCAN_V1_VERSION);
    if (!vt) return vt.error();
    return std::make_unique<AccelVectorIndex>(std::move(plugin), vt.value(), dim, engine);
}

} // namespace yams::vector
