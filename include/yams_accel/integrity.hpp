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
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <functional>
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
    // One SHA-256 chain of a buffer, as lower-case hex — the host's own hasher (e.g. a lambda around
    // crypto::SHA256Hasher::hash).  Used for the chains the device REFUSES (YAMS_ERR_UNSUPPORTED: a lone or dominating
    // chain above 1 MiB): whole blobs, or chunks of a store whose maxChunkSize exceeds 1 MiB.
    using HostHash = std::function<std::string(std::span<const std::byte>)>;
    AccelChunkValidator(std::shared_ptr<accel::Plugin> plugin, yams_content_hash_v1* vt, HostHash hostHash = nullptr)
        : plugin_(std::move(plugin)), vt_(vt), hostHash_(std::move(hostHash)) {}

    ChunkValidationResult validateChunk(std::span<const std::byte> chunkData, const std::string& expectedHash) {
        return validateChunks({{chunkData, expectedHash}}).front();
    }
    // one device pass for the whole list (plus one chain at a time for what the device refuses)
    std::vector<ChunkValidationResult>
    validateChunks(const std::vector<std::pair<std::span<const std::byte>, std::string>>& chunks) {
        const auto t0 = std::chrono::high_resolution_clock::now();
        const size_t n = chunks.size();
        std::vector<ChunkValidationResult> out(n);
        if (n == 0) return out;
        std::vector<const uint8_t*> ptrs(n); std::vector<size_t> lens(n);
        std::vector<char> expected(n * 65, 0);
        for (size_t i = 0; i < n; ++i) {
            ptrs[i] = reinterpret_cast<const uint8_t*>(chunks[i].first.data());
            lens[i] = chunks[i].first.size();
            const std::string& e = chunks[i].second;
            if (e.size() == 64) e.copy(expected.data() + 65 * i, 64); // anything else can never match
        }
        enum : uint8_t { kUnknown = 0, kValid = 1, kMismatch = 2 };
        std::vector<uint8_t> state(n, kUnknown);
        std::vector<std::string> got(n);      // the actual hash, where it had to be computed
        // verify `idx` in one device call; false = the device did not take the set
        auto verify = [&](const std::vector<size_t>& idx) -> yams_status_t {
            if (idx.empty()) return YAMS_OK;
            if (!vt_->verify_many) return YAMS_ERR_UNSUPPORTED;
            std::vector<const uint8_t*> p; std::vector<size_t> l; std::vector<char> e(idx.size() * 65, 0); std::vector<uint8_t> v(idx.size(), 0);
            for (size_t j = 0; j < idx.size(); ++j) { p.push_back(ptrs[idx[j]]); l.push_back(lens[idx[j]]); std::copy_n(expected.data() + 65 * idx[j], 65, e.data() + 65 * j); }
            const yams_status_t st = vt_->verify_many(vt_->self, p.data(), l.data(), e.data(), idx.size(), v.data());
            if (st == YAMS_OK) for (size_t j = 0; j < idx.size(); ++j) state[idx[j]] = v[j] ? kValid : kMismatch;
            return st;
        };
        // one chain on its own: the device's one-shot door, else the host's hasher, else the device's streaming door
        auto chain = [&](size_t i) {
            char hex[65];
            yams_status_t st = vt_->hash(vt_->self, ptrs[i], lens[i], hex);
            if (st == YAMS_OK) { got[i].assign(hex, 64); return; }
            if (st != YAMS_ERR_UNSUPPORTED) return;
            if (hostHash_) { got[i] = hostHash_(chunks[i].first); return; }
            void* h = nullptr;
            if (vt_->stream_create(vt_->self, &h) != YAMS_OK) return;
            st = vt_->stream_update(vt_->self, h, ptrs[i], lens[i]);
            if (st == YAMS_OK) st = vt_->stream_finalize(vt_->self, h, hex);
            vt_->stream_destroy(vt_->self, h);
            if (st == YAMS_OK) got[i].assign(hex, 64);
        };
        std::vector<size_t> all(n);
        for (size_t i = 0; i < n; ++i) all[i] = i;
        yams_status_t st = verify(all);
        if (st == YAMS_ERR_UNSUPPORTED) {
            // A lone or dominating long chain in the set (plugin.cpp, chains_suit_the_device): intact data must not be
            // reported corrupt for that.  The chains above 1 MiB go one at a time, the rest is a set the device takes.
            std::vector<size_t> small;
            for (size_t i = 0; i < n; ++i) if (lens[i] > (size_t(1) << 20)) chain(i); else small.push_back(i);
            if (verify(small) != YAMS_OK) for (size_t i : small) chain(i);
        } else if (st != YAMS_OK) {
            for (size_t i = 0; i < n; ++i) chain(i); // (a failed batch: every chunk still gets its own verdict where possible)
        }
        // the mismatch text quotes the actual hash (chunk_validator.cpp:250-253): fetch it for the failures only
        std::vector<size_t> bad;
        for (size_t i = 0; i < n; ++i) if (state[i] == kMismatch && got[i].empty()) bad.push_back(i);
        if (!bad.empty()) {
            std::vector<const uint8_t*> bp; std::vector<size_t> bl;
            for (size_t i : bad) { bp.push_back(ptrs[i]); bl.push_back(lens[i]); }
            std::vector<char> hx(bad.size() * 65);
            if (vt_->hash_many(vt_->self, bp.data(), bl.data(), bad.size(), hx.data()) == YAMS_OK)
                for (size_t j = 0; j < bad.size(); ++j) got[bad[j]].assign(hx.data() + 65 * j, 64);
            else for (size_t i : bad) chain(i);
        }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0);
        for (size_t i = 0; i < n; ++i) {
            auto& r = out[i];
            r.chunkHash = chunks[i].second; r.chunkOffset = 0; r.chunkSize = lens[i]; r.validationTime = ms;
            if (state[i] == kValid) { r.isValid = true; continue; }
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
    HostHash hostHash_;
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
