// oracle/scan_ref_wrap.cpp — TEST INFRASTRUCTURE, never on the product path.  C-linkage doors into the REFERENCE's own
// exact-scan loop: SqliteVecBackend::Impl::bruteForceSearchUnlocked (src/vector/sqlite_vec_backend.cpp:4115-4409, both
// its fast path and its metadata-filter path), its helpers (:204-236), recordFromStatement (:3102-3160), the vectors
// table (:342-371) and VectorDatabase::computeCosineSimilarity (src/vector/vector_database.cpp:1786-1810).
//
// sqlite_vec_backend.cpp cannot be compiled whole here (it includes the absent third_party/sqlite-vec-cpp and simeon,
// nlohmann/json, spdlog), so oracle/gen_scan_ref.py cuts those functions VERBATIM out of the sources where they lie
// into oracle/_ref/*.inc at build time (git-ignored; no reference text is committed) and this file gives them the
// few things they touch: a `db_` member over an in-memory SQLite (the image's /opt/conda sqlite3), the reference's own
// headers for VectorRecord / VectorSearchDiagnostics / Result / ErrorCode, and stand-ins for the two JSON readers
// recordFromStatement calls (the harness writes the metadata column itself, as a flat {"k":"v"} object).
// What this pins: tests/test_oracle.py compares oracle/yams_oracle.c's oracle_exact_scan_cosine / _records /
// oracle_cosine_similarity with THIS code, bit for bit, and tests/golden/make_scan_golden.py writes its outputs to
// tests/golden/scan.json for the GPU box (where /root/reference does not exist).
#include <sqlite3.h>

#include <yams/common/time_utils.h>
#include <yams/core/types.h>
#include <yams/vector/vector_database.h>
#include <yams/vector/vector_types.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <optional>
#include <span>
#include <string>
#include <string_view>
#include <unordered_set>
#include <vector>

namespace yams::vector {
namespace {

// ---- verbatim reference text: ExactRowSelection, isZeroNormEmbedding, isFiniteEmbedding, kCreateVectorsTable --------
#include "_ref/scan_ref_file_scope.inc"

// ---- harness stand-ins for :540-575 (nlohmann::json readers): flat {"key":"value",...} / ["a","b"] written below ----
std::string unescape(std::string_view s, size_t& i) { // s[i] == '"' on entry; leaves i past the closing quote
    std::string out;
    for (++i; i < s.size() && s[i] != '"'; ++i) {
        if (s[i] == '\\' && i + 1 < s.size()) ++i;
        out.push_back(s[i]);
    }
    ++i;
    return out;
}
std::map<std::string, std::string> deserializeMetadata(const std::string& json_str) {
    std::map<std::string, std::string> m;
    std::string_view s = json_str;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && s[i] != '"') ++i;
        if (i >= s.size()) break;
        std::string k = unescape(s, i);
        while (i < s.size() && s[i] != '"') ++i;
        if (i >= s.size()) break;
        m[k] = unescape(s, i);
    }
    return m;
}
std::vector<std::string> deserializeStringVector(const std::string& json_str) {
    std::vector<std::string> v;
    std::string_view s = json_str;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && s[i] != '"') ++i;
        if (i >= s.size()) break;
        v.push_back(unescape(s, i));
    }
    return v;
}

struct RefScan {
    sqlite3* db_ = nullptr;
    sqlite3_stmt* insert_ = nullptr;
    // ---- verbatim reference text: recordFromStatement, bruteForceSearchUnlocked ----------------------------------------
#include "_ref/scan_ref_members.inc"
};

} // namespace

// ---- verbatim reference text: VectorDatabase::computeCosineSimilarity (a static member declared in vector_database.h)
#include "_ref/scan_ref_cosine.inc"

} // namespace yams::vector

using yams::vector::RefScan;

extern "C" {

__attribute__((visibility("default"))) void* scanref_open(void) {
    auto* h = new RefScan();
    if (sqlite3_open(":memory:", &h->db_) != SQLITE_OK) { delete h; return nullptr; }
    char* err = nullptr;
    if (sqlite3_exec(h->db_, yams::vector::kCreateVectorsTable, nullptr, nullptr, &err) != SQLITE_OK) {
        sqlite3_free(err); sqlite3_close(h->db_); delete h; return nullptr;
    }
    const char* ins = "INSERT INTO vectors (chunk_id, document_hash, embedding, embedding_dim, content, start_offset, "
                      "metadata) VALUES (?1, ?2, ?3, ?4, '', ?5, ?6)";
    if (sqlite3_prepare_v2(h->db_, ins, -1, &h->insert_, nullptr) != SQLITE_OK) { sqlite3_close(h->db_); delete h; return nullptr; }
    sqlite3_exec(h->db_, "BEGIN", nullptr, nullptr, nullptr);
    return h;
}

__attribute__((visibility("default"))) void scanref_close(void* hv) {
    auto* h = static_cast<RefScan*>(hv);
    if (!h) return;
    if (h->insert_) sqlite3_finalize(h->insert_);
    if (h->db_) sqlite3_close(h->db_);
    delete h;
}

// One row, as the backend stores it (:3002-3003: the raw fp32 blob, not normalised).  `ordinal` goes into start_offset
// (VectorRecord carries no rowid) so that results can be mapped back to the caller's row numbers.  blob_bytes need not
// be embedding_dim * 4 (the scan skips such rows, :4244-4246); metadata_json may be null.
__attribute__((visibility("default"))) int scanref_insert(void* hv, const char* chunk_id, const char* document_hash, const void* blob,
                                                          int blob_bytes, long long embedding_dim, long long ordinal,
                                                          const char* metadata_json) {
    auto* h = static_cast<RefScan*>(hv);
    sqlite3_stmt* s = h->insert_;
    sqlite3_reset(s);
    sqlite3_bind_text(s, 1, chunk_id, -1, SQLITE_TRANSIENT);
    sqlite3_bind_text(s, 2, document_hash ? document_hash : "", -1, SQLITE_TRANSIENT);
    if (blob && blob_bytes > 0) sqlite3_bind_blob(s, 3, blob, blob_bytes, SQLITE_TRANSIENT);
    else sqlite3_bind_null(s, 3);
    sqlite3_bind_int64(s, 4, embedding_dim);
    sqlite3_bind_int64(s, 5, ordinal);
    if (metadata_json) sqlite3_bind_text(s, 6, metadata_json, -1, SQLITE_TRANSIENT);
    else sqlite3_bind_null(s, 6);
    return sqlite3_step(s) == SQLITE_DONE ? 0 : 1;
}

// n rows of `dim` floats; chunk ids from `chunk_ids` (n C strings) or, when null, "c" + the zero-padded ordinal (so
// that chunk-id order == row order, the case the oracle's default tie rank restates).
__attribute__((visibility("default"))) int scanref_insert_rows_docs(void* hv, const float* rows, size_t n, size_t dim, const char* const* chunk_ids,
                                                                    const char* const* document_hashes, long long first_ordinal) {
    char id[32];
    for (size_t r = 0; r < n; ++r) {
        const char* cid = chunk_ids ? chunk_ids[r] : id;
        if (!chunk_ids) std::snprintf(id, sizeof id, "c%018lld", first_ordinal + static_cast<long long>(r));
        if (scanref_insert(hv, cid, document_hashes ? document_hashes[r] : "doc", rows + r * dim, static_cast<int>(dim * sizeof(float)), static_cast<long long>(dim),
                           first_ordinal + static_cast<long long>(r), nullptr) != 0)
            return 1;
    }
    return 0;
}
__attribute__((visibility("default"))) int scanref_insert_rows(void* hv, const float* rows, size_t n, size_t dim, const char* const* chunk_ids,
                                                               long long first_ordinal) {
    return scanref_insert_rows_docs(hv, rows, n, dim, chunk_ids, nullptr, first_ordinal);
}

// bruteForceSearchUnlocked(query, k, threshold, document_hash, candidate_hashes, metadata_filters, &diagnostics, rowSelection).
// document_hash (nullable) and candidate_hashes restrict the rows the statement visits (:4151-4195: pushed into SQL).
// meta_kv: n_meta (key, value) pairs, flattened; a NON-EMPTY filter map selects the reference's record path (:4333-4409).
// Returns 0 and fills out_* (ordinals = start_offset of the returned records, scores = relevance_score; *out_n may
// exceed cap: the count the reference returned), or the reference's ErrorCode as a negative number.
// diag[4] = rowsVisited, exactDistanceEvaluations, returnedRows, usedExactScan.
__attribute__((visibility("default"))) long scanref_search_ex(void* hv, const float* query, size_t dim, size_t k, float threshold,
                                                              const char* document_hash, const char* const* candidate_hashes, size_t n_candidates,
                                                              const char* const* meta_kv, size_t n_meta, int all_matching,
                                                              long long* out_ordinals, float* out_scores, size_t cap, size_t* out_n,
                                                              unsigned long long* diag) {
    auto* h = static_cast<RefScan*>(hv);
    sqlite3_exec(h->db_, "COMMIT", nullptr, nullptr, nullptr); // (no-op when no transaction is open)
    std::vector<float> q(query, query + dim);
    std::map<std::string, std::string> filters;
    for (size_t i = 0; i < n_meta; ++i) filters[meta_kv[2 * i]] = meta_kv[2 * i + 1];
    std::optional<std::string> doc;
    if (document_hash) doc = document_hash;
    std::unordered_set<std::string> cands;
    for (size_t i = 0; i < n_candidates; ++i) cands.insert(candidate_hashes[i]);
    yams::vector::VectorSearchDiagnostics dg;
    auto r = h->bruteForceSearchUnlocked(q, k, threshold, doc, cands, filters, &dg,
                                         all_matching ? yams::vector::ExactRowSelection::AllMatching
                                                      : yams::vector::ExactRowSelection::TopK);
    if (diag) { diag[0] = dg.rowsVisited; diag[1] = dg.exactDistanceEvaluations; diag[2] = dg.returnedRows; diag[3] = dg.usedExactScan ? 1 : 0; }
    if (!r) return -static_cast<long>(r.error().code);
    const auto& recs = r.value();
    if (out_n) *out_n = recs.size();
    for (size_t i = 0; i < recs.size() && i < cap; ++i) {
        out_ordinals[i] = static_cast<long long>(recs[i].start_offset);
        out_scores[i] = recs[i].relevance_score;
    }
    return 0;
}

// (the plain form: no document / candidate restriction)
__attribute__((visibility("default"))) long scanref_search(void* hv, const float* query, size_t dim, size_t k, float threshold,
                                                           const char* const* meta_kv, size_t n_meta, int all_matching,
                                                           long long* out_ordinals, float* out_scores, size_t cap, size_t* out_n,
                                                           unsigned long long* diag) {
    return scanref_search_ex(hv, query, dim, k, threshold, nullptr, nullptr, 0, meta_kv, n_meta, all_matching, out_ordinals, out_scores, cap,
                             out_n, diag);
}

__attribute__((visibility("default"))) double scanref_cosine(const float* a, size_t na, const float* b, size_t nb) {
    return yams::vector::VectorDatabase::computeCosineSimilarity(std::vector<float>(a, a + na), std::vector<float>(b, b + nb));
}

__attribute__((visibility("default"))) int scanref_error_code_invalid_argument(void) { return static_cast<int>(yams::ErrorCode::InvalidArgument); }

} // extern "C"
