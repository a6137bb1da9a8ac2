// exact_scan_backend.hpp — yams::vector::IVectorStore (+ the four optional capability seams) over the
// accelerator: the class a patched VectorDatabase::Impl constructs instead of SqliteVecBackend
// (src/vector/vector_database.cpp:45-56 builds the backend unconditionally — a one-line seam).
//
// Only compiled inside a YAMS tree: define YAMS_ACCEL_USE_HOST_TYPES and put the host's include/
// on the path; the class then derives from the host's OWN interfaces
//   IVectorStore                      include/yams/vector/vector_store.h:23-77
//   IDiagnosticVectorStore            :95-107    IDocumentCandidateVectorStore  :109-120
//   IExactCandidateVectorStore        :122-133   IAllExactCandidateVectorStore  :135-142
// which VectorDatabase discovers with dynamic_cast (vector_database.cpp:553-609).
//
// What lives where: the embeddings live in HBM (AccelVectorTable: one dense device mirror per
// dimension, row-sharded over the plugin's devices) and every search — plain, filtered, exact
// candidate, all-rows, per-document — is the device's exact scan.  The records (ids, content,
// metadata) stay on the host.  Persistence is NOT this class's job: when a `durable` store is given
// (the host's SqliteVecBackend with search disabled, or any IVectorStore), every mutation and every
// SQL-only method is forwarded to it and the mirror is warmed from it at initialize(); without one
// the backend is an in-memory store (tests, caches).
//
// Transactions (vector_store.h:71-76; SqliteVecBackend: BEGIN IMMEDIATE ... COMMIT / ROLLBACK): the device mirror
// holds COMMITTED state.  While a transaction is open, mutations go to the durable store at once (it is what can
// roll back) and are JOURNALED for the mirror; commit applies the journal in order, rollback drops it — a search
// never returns a chunk_id that getVector() cannot resolve afterwards.  Without a durable store the mirror is the
// store: mutations apply at once and an undo log restores the previous rows on rollback.  If the mirror fails after
// the durable store succeeded (device out of memory), the mirror is marked stale and re-warmed from the durable
// store before the next search.
#pragma once
#ifndef YAMS_ACCEL_USE_HOST_TYPES
#error "exact_scan_backend.hpp binds to the host's headers: define YAMS_ACCEL_USE_HOST_TYPES"
#endif
#include <algorithm>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <variant>

#include "vector_index.hpp"

namespace yams::vector {

class AccelExactScanBackend final : public IVectorStore,
                                    public IDiagnosticVectorStore,
                                    public IDocumentCandidateVectorStore,
                                    public IExactCandidateVectorStore,
                                    public IAllExactCandidateVectorStore {
public:
    explicit AccelExactScanBackend(std::shared_ptr<accel::Plugin> plugin, std::shared_ptr<IVectorStore> durable = nullptr,
                                   VectorSearchEngine engine = VectorSearchEngine::ExactScan)
        : plugin_(std::move(plugin)), durable_(std::move(durable)), table_(plugin_, engine) {}

    // ---- vec0 L2: settle the distance arithmetic with the host's own function (l2_calibration.hpp) ---------------
    // Call once after construction when the engine is Vec0L2, e.g.
    //     backend.calibrateL2(accel_l2::fromCApi(&sqlite3_vec_distance_l2));
    // Matched: every L2 search runs in that arithmetic (overrides the plugin's "l2_accumulate").  Not matched: the
    // returned record says so and every L2 search fails with ErrorCode::NotSupported — the host keeps its CPU vec0.
    Result<accel_l2::L2Calibration> calibrateL2(const accel_l2::L2DistanceFn& fn) {
        std::unique_lock lk(mu_);
        return table_.calibrateL2(fn, dim_);   // (reported for the schema's dimension; every dimension present is calibrated on its own)
    }

    // ---- lifecycle / schema --------------------------------------------------------------------------
    Result<void> initialize(const std::string& db_path) override {
        std::unique_lock lk(mu_);
        if (durable_) {
            if (auto s = durable_->initialize(db_path); !s) return s;
            if (auto s = warmLocked(); !s) return s;
        }
        initialized_ = true;
        return {};
    }
    void close() override {
        std::unique_lock lk(mu_);
        if (durable_) durable_->close();
        initialized_ = false;
    }
    bool isInitialized() const override { return initialized_; }
    Result<void> createTables(size_t embedding_dim) override {
        dim_ = embedding_dim;
        tables_ = true;
        return durable_ ? durable_->createTables(embedding_dim) : Result<void>{};
    }
    bool tablesExist() const override { return durable_ ? durable_->tablesExist() : tables_; }

    // ---- CRUD: durable store first (it validates and persists), then the device mirror — at once outside a
    // transaction, at commit inside one -------------------------------------------------------------------------
    Result<void> insertVector(const VectorRecord& record) override { return insertVectorsBatch({record}); }
    Result<void> insertVectorsBatch(const std::vector<VectorRecord>& records) override {
        std::unique_lock lk(mu_);
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (durable_) if (auto s = durable_->insertVectorsBatch(records); !s) return s;
        return mirror(Op{InsertOp{records}});
    }
    Result<void> updateVector(const std::string& chunk_id, const VectorRecord& record) override {
        std::unique_lock lk(mu_);
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (durable_) {
            if (auto s = durable_->updateVector(chunk_id, record); !s) return s; // (it knows whether the chunk exists)
        } else {
            auto have = table_.getVector(chunk_id);
            if (!have || !have.value()) return Error{ErrorCode::NotFound, "chunk not found"};
        }
        VectorRecord r = record;
        r.chunk_id = chunk_id;
        return mirror(Op{InsertOp{{std::move(r)}}});
    }
    Result<void> deleteVector(const std::string& chunk_id) override {
        std::unique_lock lk(mu_);
        if (durable_) if (auto s = durable_->deleteVector(chunk_id); !s) return s;
        return mirror(Op{DeleteOp{chunk_id}});
    }
    Result<void> deleteVectorsByDocument(const std::string& document_hash) override {
        std::unique_lock lk(mu_);
        if (durable_) if (auto s = durable_->deleteVectorsByDocument(document_hash); !s) return s;
        return mirror(Op{DeleteDocOp{document_hash}});
    }

    // ---- search: always the device -------------------------------------------------------------------
    Result<std::vector<VectorRecord>>
    searchSimilar(const std::vector<float>& query_embedding, size_t k, float similarity_threshold = 0.0f,
                  const std::optional<std::string>& document_hash = std::nullopt,
                  const std::unordered_set<std::string>& candidate_hashes = {},
                  const std::map<std::string, std::string>& metadata_filters = {}) override {
        return search(query_embedding, k, similarity_threshold, document_hash, candidate_hashes, metadata_filters,
                      nullptr, ExactRowSelection::TopK);
    }
    Result<std::vector<std::vector<VectorRecord>>>
    searchSimilarBatch(const std::vector<std::vector<float>>& query_embeddings, size_t k,
                       float similarity_threshold = 0.0f, size_t num_threads = 0) override {
        // searches share the lock (vector_database.cpp:539,618): the plugin's search lanes serve them side by side
        return withSyncedMirror([&] { return table_.searchSimilarBatch(query_embeddings, k, similarity_threshold, num_threads); });
    }
    // ---- the product-quantised engine (VectorSearchEngine::SimeonPqAdc, the default: vector_types.h:80) ----------------------
    // simeonPqSearchUnlocked (sqlite_vec_backend.cpp:3868-4056) with its ADC scan, candidate selection and exact re-rank on the
    // device.  A patched SqliteVecBackend keeps simeon — it trains, encodes (:3540-3690) and builds the per-query table
    // (:3895-3901) — and calls these two where it called its own loop: setSimeonPqIndex after every rebuild / load of
    // simeon_pq_indices_[dim] (codes + the chunk id of every indexed row), searchSimeonPq from searchSimilar when
    // usesSimeonPqSearchEngine() (:1506-1560).  exact_fallback indexes (:3882-3893) keep calling searchSimilar.
    Result<void> setSimeonPqIndex(size_t dim, const std::vector<uint8_t>& codes, size_t m, const std::vector<std::string>& chunkIdOfIndex) {
        std::unique_lock lk(mu_);
        return table_.setPqIndex(dim, codes, m, chunkIdOfIndex);
    }
    Result<std::vector<VectorRecord>>
    searchSimeonPq(const std::vector<float>& query_embedding, const std::vector<float>& lut, size_t k, float similarity_threshold,
                   size_t rerank_factor, const std::vector<uint32_t>* candidate_indices = nullptr,
                   uint32_t sum_flags = YAMS_PQ_SUM_SEQUENTIAL, VectorSearchDiagnostics* diagnostics = nullptr) {
        if (query_embedding.empty() || k == 0) return std::vector<VectorRecord>{};                  // :3873-3875
        return withSyncedMirror([&]() -> Result<std::vector<VectorRecord>> {
            auto r = table_.searchPqBatch({query_embedding}, {lut}, k, similarity_threshold, rerank_factor, candidate_indices, sum_flags, diagnostics);
            if (!r) return r.error();
            return std::move(r.value().front());
        });
    }
    // sqlite_vec_backend.cpp:4650-4661: the diagnostics are reset, the caller's collect flag survives
    Result<std::vector<VectorRecord>>
    searchSimilarWithDiagnostics(const std::vector<float>& query_embedding, size_t k, float similarity_threshold,
                                 const std::optional<std::string>& document_hash,
                                 const std::unordered_set<std::string>& candidate_hashes,
                                 const std::map<std::string, std::string>& metadata_filters,
                                 VectorSearchDiagnostics& diagnostics) override {
        resetKeepingCollectFlag(diagnostics);
        return search(query_embedding, k, similarity_threshold, document_hash, candidate_hashes, metadata_filters,
                      &diagnostics, ExactRowSelection::TopK);
    }
    // sqlite_vec_backend.cpp:1578-1593, 4673-4681
    Result<std::vector<VectorRecord>>
    searchExactCandidatesWithDiagnostics(const std::vector<float>& query_embedding, size_t k, float similarity_threshold,
                                         const std::unordered_set<std::string>& candidate_hashes,
                                         VectorSearchDiagnostics& diagnostics) override {
        resetKeepingCollectFlag(diagnostics);
        if (candidate_hashes.empty())
            return Error{ErrorCode::InvalidArgument, "Exact candidate search requires candidate hashes"};
        return search(query_embedding, k, similarity_threshold, std::nullopt, candidate_hashes, {}, &diagnostics,
                      ExactRowSelection::TopK);
    }
    // sqlite_vec_backend.cpp:1595-1610, 4683-4691: every matching row of the candidate documents
    Result<std::vector<VectorRecord>>
    searchAllExactCandidateRowsWithDiagnostics(const std::vector<float>& query_embedding, float similarity_threshold,
                                               const std::unordered_set<std::string>& candidate_hashes,
                                               VectorSearchDiagnostics& diagnostics) override {
        resetKeepingCollectFlag(diagnostics);
        if (candidate_hashes.empty())
            return Error{ErrorCode::InvalidArgument, "Exact candidate search requires candidate hashes"};
        return search(query_embedding, 0, similarity_threshold, std::nullopt, candidate_hashes, {}, &diagnostics,
                      ExactRowSelection::AllMatching);
    }
    // The exact arm of document-level selection (sqlite_vec_backend.cpp:1508-1518: all matching rows,
    // then the best row per document, :86-125) — what the product's PQ engine falls back to whenever
    // its compressed state is missing, stale or bypassed by a filter.
    Result<std::vector<VectorRecord>>
    searchDocumentCandidatesWithDiagnostics(const std::vector<float>& query_embedding, size_t k, float similarity_threshold,
                                            const std::unordered_set<std::string>& candidate_hashes,
                                            VectorSearchDiagnostics& diagnostics) override {
        resetKeepingCollectFlag(diagnostics);
        if (candidate_hashes.empty())
            return Error{ErrorCode::InvalidArgument, "Document candidate search requires candidate hashes"};
        auto rows = search(query_embedding, 0, similarity_threshold, std::nullopt, candidate_hashes, {}, &diagnostics,
                           ExactRowSelection::AllMatching);
        if (!rows) return rows;
        return bestRecordPerDocument(std::move(rows.value()), k);
    }

    // ---- retrieval / existence / stats: from the durable store when there is one, else the mirror ---------
    Result<std::optional<VectorRecord>> getVector(const std::string& chunk_id) override {
        std::shared_lock lk(mu_);
        return durable_ ? durable_->getVector(chunk_id) : table_.getVector(chunk_id);
    }
    Result<std::map<std::string, VectorRecord>> getVectorsBatch(const std::vector<std::string>& chunk_ids) override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->getVectorsBatch(chunk_ids);
        std::map<std::string, VectorRecord> out;
        for (const auto& id : chunk_ids)
            if (auto r = table_.getVector(id); r && r.value()) out.emplace(id, *r.value());
        return out;
    }
    Result<std::vector<VectorRecord>> getVectorsByDocument(const std::string& document_hash) override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->getVectorsByDocument(document_hash);
        std::vector<VectorRecord> out;
        table_.forEachRecord([&](const VectorRecord& r) { if (r.document_hash == document_hash) out.push_back(r); });
        return out;
    }
    Result<std::unordered_map<std::string, VectorRecord>> getDocumentLevelVectorsAll() override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->getDocumentLevelVectorsAll();
        std::unordered_map<std::string, VectorRecord> out;
        table_.forEachRecord([&](const VectorRecord& r) { if (r.level == EmbeddingLevel::DOCUMENT) out[r.document_hash] = r; });
        return out;
    }
    Result<size_t> forEachDocumentLevelVector(const std::function<bool(VectorRecord&&)>& visitor) override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->forEachDocumentLevelVector(visitor);
        size_t n = 0; bool go = true;
        table_.forEachRecord([&](const VectorRecord& r) {
            if (!go || r.level != EmbeddingLevel::DOCUMENT) return;
            ++n;
            go = visitor(VectorRecord(r));
        });
        return n;
    }
    Result<bool> hasEmbedding(const std::string& document_hash) override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->hasEmbedding(document_hash);
        bool any = false;
        table_.forEachRecord([&](const VectorRecord& r) { any = any || r.document_hash == document_hash; });
        return any;
    }
    Result<std::unordered_set<std::string>> getEmbeddedDocumentHashes() override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->getEmbeddedDocumentHashes();
        std::unordered_set<std::string> out;
        table_.forEachRecord([&](const VectorRecord& r) { out.insert(r.document_hash); });
        return out;
    }
    Result<size_t> getVectorCount() override {
        std::shared_lock lk(mu_);
        return durable_ ? durable_->getVectorCount() : table_.getVectorCount();
    }
    Result<VectorDatabaseStats> getStats() override {
        std::shared_lock lk(mu_);
        if (durable_) return durable_->getStats();
        VectorDatabaseStats st;
        std::unordered_set<std::string> docs;
        double mag = 0.0;
        table_.forEachRecord([&](const VectorRecord& r) {
            ++st.total_vectors; docs.insert(r.document_hash);
            double n = 0.0;
            for (float v : r.embedding) n += static_cast<double>(v) * v;
            mag += std::sqrt(n);
            st.index_size_bytes += r.embedding.size() * sizeof(float);
        });
        st.total_documents = docs.size();
        st.avg_embedding_magnitude = st.total_vectors ? mag / static_cast<double>(st.total_vectors) : 0.0;
        return st;
    }
    // ---- transactions: the durable store's; the mirror holds committed state (see the header comment) ---------
    // The durable store was changed by someone other than this object (another process, a restore): the mirror is
    // rebuilt from it before the next search.
    void invalidateMirror() { std::unique_lock lk(mu_); if (durable_) mirrorStale_ = true; }
    Result<void> beginTransaction() override {
        std::unique_lock lk(mu_);
        if (inTxn_) return Error{ErrorCode::InvalidState, "transaction already open"};
        if (durable_) if (auto s = durable_->beginTransaction(); !s) return s;
        inTxn_ = true; rewarmedInTxn_ = false; journal_.clear(); undo_.clear();
        return {};
    }
    Result<void> commitTransaction() override {
        std::unique_lock lk(mu_);
        if (!inTxn_) return durable_ ? durable_->commitTransaction() : Result<void>{};
        if (durable_) if (auto s = durable_->commitTransaction(); !s) return s; // (still open: the caller rolls back)
        inTxn_ = false; rewarmedInTxn_ = false;
        undo_.clear();
        std::vector<Op> ops;
        ops.swap(journal_);
        for (auto& op : ops)
            if (auto s = applyToMirror(op); !s) { mirrorStale_ = durable_ != nullptr; if (!durable_) return s; break; }
        return {};
    }
    Result<void> rollbackTransaction() override {
        std::unique_lock lk(mu_);
        if (!inTxn_) return durable_ ? durable_->rollbackTransaction() : Result<void>{};
        inTxn_ = false;
        journal_.clear(); // (durable mode: the mirror never saw these — unless it was re-warmed inside the transaction)
        if (rewarmedInTxn_) { mirrorStale_ = true; rewarmedInTxn_ = false; }
        Result<void> rc{};
        if (durable_) rc = durable_->rollbackTransaction();
        else {
            // in-memory mode: put the previous rows back, newest change first
            for (auto it = undo_.rbegin(); it != undo_.rend(); ++it) {
                for (const auto& id : it->inserted) (void)table_.deleteVector(id);
                if (!it->previous.empty()) if (auto s = table_.insertVectorsBatch(it->previous); !s) rc = s;
            }
        }
        undo_.clear();
        return rc;
    }

private:
    struct InsertOp { std::vector<VectorRecord> records; };
    struct DeleteOp { std::string chunk_id; };
    struct DeleteDocOp { std::string document_hash; };
    using Op = std::variant<InsertOp, DeleteOp, DeleteDocOp>;
    struct Undo { std::vector<std::string> inserted; std::vector<VectorRecord> previous; };

    Result<void> applyToMirror(const Op& op) {
        if (const auto* i = std::get_if<InsertOp>(&op)) return table_.insertVectorsBatch(i->records);
        if (const auto* d = std::get_if<DeleteOp>(&op)) {
            auto s = table_.deleteVector(d->chunk_id);
            // (the durable store accepted the delete of a chunk the mirror does not hold: nothing to do)
            if (!s && durable_ && s.error().code == ErrorCode::NotFound) return {};
            return s;
        }
        return table_.deleteVectorsByDocument(std::get<DeleteDocOp>(op).document_hash);
    }
    // One mutation on its way to the mirror (mu_ held exclusively; the durable store has already accepted it).
    Result<void> mirror(Op op) {
        if (inTxn_ && durable_) { journal_.push_back(std::move(op)); return {}; } // applied at commit
        if (inTxn_) { // in-memory mode: apply now, remember how to undo it
            Undo u;
            if (const auto* i = std::get_if<InsertOp>(&op)) {
                for (const auto& r : i->records) {
                    auto have = table_.getVector(r.chunk_id);
                    if (have && have.value()) u.previous.push_back(*have.value());
                    u.inserted.push_back(r.chunk_id);
                }
            } else if (const auto* d = std::get_if<DeleteOp>(&op)) {
                auto have = table_.getVector(d->chunk_id);
                if (have && have.value()) u.previous.push_back(*have.value());
            } else {
                const auto& h = std::get<DeleteDocOp>(op).document_hash;
                table_.forEachRecord([&](const VectorRecord& r) { if (r.document_hash == h) u.previous.push_back(r); });
            }
            auto s = applyToMirror(op);
            if (s) undo_.push_back(std::move(u));
            return s;
        }
        auto s = applyToMirror(op);
        if (!s && durable_) { mirrorStale_ = true; return {}; } // the durable store HAS the change: re-warm before the next search
        return s;
    }
    // (re)builds the mirror from the durable rows (document-level rows included)
    Result<void> warmLocked() {
        std::vector<VectorRecord> all;
        auto hashes = durable_->getEmbeddedDocumentHashes();
        if (!hashes) return hashes.error();
        for (const auto& h : hashes.value()) {
            auto rows = durable_->getVectorsByDocument(h);
            if (!rows) return rows.error();
            for (auto& r : rows.value()) all.push_back(std::move(r));
        }
        if (!all.empty())
            if (auto s = table_.insertVectorsBatch(all); !s) return s;
        return {};
    }
    // Everything a search may have to WRITE happens here, under the exclusive lock: re-warming a stale mirror and
    // uploading rows appended since the last search.  The search itself then only reads (shared lock).
    Result<void> syncForSearch() {
        {
            std::shared_lock rd(mu_);
            if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
            if (!mirrorStale_ && !table_.needsSync()) return {};
        }
        std::unique_lock lk(mu_);
        if (auto s = rewarmIfStaleLocked(); !s) return s;
        return table_.sync();
    }
    Result<void> rewarmIfStaleLocked() {
        if (!mirrorStale_ || !durable_) return {};
        table_.clear();
        if (auto s = warmLocked(); !s) return s; // (still stale: the next search tries again)
        mirrorStale_ = false;
        // Inside an open durable transaction the rows just read include its uncommitted changes (the journal will be
        // applied over them at commit: inserts replace, deletes are idempotent).  Should the transaction roll back, the
        // mirror would keep rows getVector() cannot resolve — so a rollback marks it stale again (rollbackTransaction).
        if (inTxn_) rewarmedInTxn_ = true;
        return {};
    }
    static void resetKeepingCollectFlag(VectorSearchDiagnostics& d) {
        const bool collect = d.collectVisitedDocumentHashes;
        d = {};
        d.collectVisitedDocumentHashes = collect;
    }
    Result<std::vector<VectorRecord>>
    search(const std::vector<float>& q, size_t k, float thr, const std::optional<std::string>& document_hash,
           const std::unordered_set<std::string>& candidate_hashes, const std::map<std::string, std::string>& metadata_filters,
           VectorSearchDiagnostics* diagnostics, ExactRowSelection selection) {
        return withSyncedMirror([&]() -> Result<std::vector<VectorRecord>> {
            if (q.empty() || (selection == ExactRowSelection::TopK && k == 0)) return std::vector<VectorRecord>{}; // :4123-4126
            if (selection == ExactRowSelection::AllMatching)
                return table_.searchSimilarRows(q, k, thr, candidate_hashes, diagnostics, selection);
            return table_.searchSimilar(q, k, thr, document_hash, candidate_hashes, metadata_filters, diagnostics);
        });
    }
    // Runs `scan` with the mirror in step with the committed state.  The scan is the long part: searches share the
    // lock, only the mirror upload excludes.  A writer may slip in between the exclusive synchronisation and the shared
    // lock; that is retried a bounded number of times, then the search runs UNDER the exclusive lock (no recursion,
    // no livelock under a steady stream of writers).
    template <class F>
    auto withSyncedMirror(F&& scan) -> decltype(scan()) {
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (auto s = syncForSearch(); !s) return s.error();
            std::shared_lock lk(mu_);
            if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
            if (!mirrorStale_ && !table_.needsSync()) return scan();
        }
        std::unique_lock lk(mu_);
        if (!initialized_) return Error{ErrorCode::NotInitialized, "Database not initialized"};
        if (auto s = rewarmIfStaleLocked(); !s) return s.error();
        if (auto s = table_.sync(); !s) return s.error();
        return scan();
    }
    // retainBestRecordPerDocument, sqlite_vec_backend.cpp:86-125
    static std::vector<VectorRecord> bestRecordPerDocument(std::vector<VectorRecord> records, size_t limit) {
        std::unordered_map<std::string, VectorRecord> best;
        for (auto& r : records) {
            if (r.document_hash.empty()) continue;
            auto it = best.find(r.document_hash);
            if (it == best.end()) { best.emplace(r.document_hash, std::move(r)); continue; }
            if (r.relevance_score > it->second.relevance_score ||
                (r.relevance_score == it->second.relevance_score && r.chunk_id < it->second.chunk_id))
                it->second = std::move(r);
        }
        records.clear();
        for (auto& [h, r] : best) records.push_back(std::move(r));
        std::sort(records.begin(), records.end(), [](const VectorRecord& a, const VectorRecord& b) {
            if (a.relevance_score != b.relevance_score) return a.relevance_score > b.relevance_score;
            if (a.document_hash != b.document_hash) return a.document_hash < b.document_hash;
            return a.chunk_id < b.chunk_id;
        });
        if (records.size() > limit) records.resize(limit);
        return records;
    }

    std::shared_ptr<accel::Plugin> plugin_;
    std::shared_ptr<IVectorStore> durable_;
    AccelVectorTable table_;
    mutable std::shared_mutex mu_;
    bool initialized_ = false, tables_ = false;
    size_t dim_ = 0;
    bool inTxn_ = false, mirrorStale_ = false, rewarmedInTxn_ = false;
    std::vector<Op> journal_;   // durable mode: mutations of the open transaction, applied to the mirror at commit
    std::vector<Undo> undo_;    // in-memory mode: how to put the previous rows back on rollback
};

// The C-linkage object factory a C++ host may prefer over the vtable door (precedent:
// plugins/object_storage_s3/s3_plugin.cpp:875-881 exports yams_plugin_create_object_storage()).
inline std::unique_ptr<IVectorStore> createAccelExactScanBackend(std::shared_ptr<accel::Plugin> plugin,
                                                                 std::shared_ptr<IVectorStore> durable = nullptr) {
    return std::make_unique<AccelExactScanBackend>(std::move(plugin), std::move(durable));
}

} // namespace yams::vector
