// integrity.hpp — the callers either side of the hash / chunk path, over the content_hash_v1 vtable:
//   * AccelChunkValidator: yams::integrity::ChunkValidator's validateChunk / validateChunks
//     (include/yams/integrity/chunk_validator.h:96-132, src/integrity/chunk_validator.cpp:30-34,
//     160-212, 230-262 of the reference): isValid = (SHA-256 hex of the bytes == expected hash),
//     the same error text on a mismatch.
//   * AccelDedupIndex: the chunk loop of ContentStore::store (src/api/content_store_impl.cpp:246-287)
//     asks storage_->exists(chunk.hash) once per chunk; this answers a whole chunk list at once with
//     the same in-order rule (a chunk is new iff its hash is neither stored nor carried by an
//     earlier chunk of the list) and remembers the new hashes.
#pragma once
#include <chrono>
#include <cstddef>
#include <memory>
#include <span>
#include <string>
#include <utility>
#include <vector>

#include "plugin.hpp"

namespace yams::integrity {

struct ChunkValidationResult { // chunk_validator.h (reference)
    std::string chunkHash;
    bool isValid = false;
    std::string errorMessage;
    size_t chunkOffset = 0;
    size_t chunkSize = 0;
    std::chrono::milliseconds validationTime{0};
};

class AccelChunkValidator {
public:
    AccelChunkValidator(std::shared_ptr<accel::Plugin> plugin, yams_content_hash_v1* vt)
        : plugin_(std::move(plugin)), vt_(vt) {}

    ChunkValidationResult validateChunk(std::span<const std::byte> chunkData, const std::string& expectedHash) {
        return validateChunks({{chunkData, expectedHash}}).front();
    }
    // one device pass for the whole list
    std::vector<ChunkValidationResult>
    validateChunks(const std::vector<std::pair<std::span<const std::byte>, std::string>>& chunks) {
        const auto t0 = std::chrono::high_resolution_clock::now();
        const size_t n = chunks.size();
        std::vector<ChunkValidationResult> out(n);
        if (n == 0) return out;
        std::vector<const uint8_t*> ptrs(n); std::vector<size_t> lens(n);
        std::vector<char> expected(n * 65, 0), actual(n * 65, 0);
        for (size_t i = 0; i < n; ++i) {
            ptrs[i] = reinterpret_cast<const uint8_t*>(chunks[i].first.data());
            lens[i] = chunks[i].first.size();
            const std::string& e = chunks[i].second;
            if (e.size() == 64) e.copy(expected.data() + 65 * i, 64); // anything else can never match
        }
        std::vector<uint8_t> valid(n, 0);
        const bool ok = vt_->verify_many && vt_->verify_many(vt_->self, ptrs.data(), lens.data(), expected.data(), n, valid.data()) == YAMS_OK;
        // the mismatch text quotes the actual hash (chunk_validator.cpp:250-253): fetch it for the failures only
        std::vector<size_t> bad;
        for (size_t i = 0; i < n; ++i) if (!ok || !valid[i]) bad.push_back(i);
        std::vector<std::string> got(n);
        if (!bad.empty()) {
            std::vector<const uint8_t*> bp; std::vector<size_t> bl;
            for (size_t i : bad) { bp.push_back(ptrs[i]); bl.push_back(lens[i]); }
            std::vector<char> hx(bad.size() * 65);
            if (vt_->hash_many(vt_->self, bp.data(), bl.data(), bad.size(), hx.data()) == YAMS_OK)
                for (size_t j = 0; j < bad.size(); ++j) got[bad[j]].assign(hx.data() + 65 * j, 64);
        }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0);
        for (size_t i = 0; i < n; ++i) {
            auto& r = out[i];
            r.chunkHash = chunks[i].second; r.chunkOffset = 0; r.chunkSize = lens[i]; r.validationTime = ms;
            if (got[i].empty() && ok && valid[i]) { r.isValid = true; continue; }
            if (got[i].empty()) { r.errorMessage = "Hash calculation failed: accelerator error"; continue; }
            r.isValid = got[i] == chunks[i].second;
            if (!r.isValid)
                r.errorMessage = "Hash mismatch: expected " + chunks[i].second.substr(0, 8) + ", got " + got[i].substr(0, 8);
        }
        return out;
    }

private:
    std::shared_ptr<accel::Plugin> plugin_;
    yams_content_hash_v1* vt_;
};

class AccelDedupIndex {
public:
    AccelDedupIndex(std::shared_ptr<accel::Plugin> plugin, yams_content_hash_v1* vt, uint64_t expectedEntries = 0)
        : plugin_(std::move(plugin)), vt_(vt) {
        if (!vt_->dedup_create || vt_->dedup_create(vt_->self, expectedEntries, &id_) != YAMS_OK)
            throw std::runtime_error("Failed to create the dedup set on the accelerator");
    }
    ~AccelDedupIndex() { if (id_) vt_->dedup_destroy(vt_->self, id_); }
    AccelDedupIndex(const AccelDedupIndex&) = delete;
    AccelDedupIndex& operator=(const AccelDedupIndex&) = delete;

    // isNew[i]: chunk i has to be stored (its hash was unknown and no earlier chunk of the list
    // carries it); every new hash is known afterwards.  Hashes are 64-char hex strings.
    Result<std::vector<bool>> insertAndClassify(const std::vector<std::string>& hashes) { return run(hashes, true); }
    Result<std::vector<bool>> contains(const std::vector<std::string>& hashes) { return run(hashes, false); }
    Result<size_t> size() const {
        uint64_t n = 0;
        if (vt_->dedup_size(vt_->self, id_, &n) != YAMS_OK) return Error{ErrorCode::InternalError, "dedup_size failed"};
        return static_cast<size_t>(n);
    }

private:
    Result<std::vector<bool>> run(const std::vector<std::string>& hashes, bool insert) {
        std::vector<char> hex(hashes.size() * 65, 0);
        for (size_t i = 0; i < hashes.size(); ++i) {
            if (hashes[i].size() != 64) return Error{ErrorCode::InvalidArgument, "chunk hash must be 64 hex characters"};
            hashes[i].copy(hex.data() + 65 * i, 64);
        }
        std::vector<uint8_t> flags(hashes.size(), 0);
        const yams_status_t st = insert ? vt_->dedup_insert(vt_->self, id_, hex.data(), hashes.size(), flags.data())
                                        : vt_->dedup_contains(vt_->self, id_, hex.data(), hashes.size(), flags.data());
        if (st != YAMS_OK) return Error{accel::mapStatus(st), "dedup lookup failed"};
        return std::vector<bool>(flags.begin(), flags.end());
    }
    std::shared_ptr<accel::Plugin> plugin_;
    yams_content_hash_v1* vt_;
    uint64_t id_ = 0;
};

} // namespace yams::integrity
