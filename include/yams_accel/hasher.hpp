// hasher.hpp — yams::crypto::IContentHasher over the content_hash_v1 vtable.
// Mirrors include/yams/crypto/hasher.h:14-80 of the reference (init / update / finalize /
// hashFile / static one-shot hash) so it can be handed to ContentStoreBuilder::withHasher
// (src/api/content_store_builder.cpp:433-441).  Instances are single-threaded objects, like the
// reference's (src/crypto/sha256_hasher.cpp:34).
#pragma once
#include <algorithm>
#include <cstddef>
#include <filesystem>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <span>
#include <string>
#include <vector>

#include "plugin.hpp"

#ifdef YAMS_ACCEL_USE_HOST_TYPES
#include <yams/crypto/hasher.h> // the host's own yams::crypto::IContentHasher (hasher.h:14-47)
#endif

namespace yams::crypto {

#ifndef YAMS_ACCEL_USE_HOST_TYPES
class IContentHasher { // hasher.h:14-47 (reference), restated for builds outside the YAMS tree
public:
    virtual ~IContentHasher() = default;
    virtual void init() = 0;
    virtual void update(std::span<const std::byte> data) = 0;
    virtual std::string finalize() = 0;
    virtual std::string hashFile(const std::filesystem::path& path) = 0;
    // hasher.h:41-46
    virtual std::future<Result<std::string>> hashFileAsync(const std::filesystem::path& path) = 0;
    using ProgressCallback = std::function<void(uint64_t, uint64_t)>;
    virtual void setProgressCallback(ProgressCallback callback) = 0;
};
#endif

class AccelSHA256Hasher final : public IContentHasher {
public:
    AccelSHA256Hasher(std::shared_ptr<accel::Plugin> plugin, yams_content_hash_v1* vt)
        : plugin_(std::move(plugin)), vt_(vt) {
        if (vt_->stream_create(vt_->self, &stream_) != YAMS_OK)
            throw std::runtime_error("Failed to create SHA256 stream on the accelerator");
    }
    ~AccelSHA256Hasher() override { if (stream_) vt_->stream_destroy(vt_->self, stream_); }
    AccelSHA256Hasher(const AccelSHA256Hasher&) = delete;
    AccelSHA256Hasher& operator=(const AccelSHA256Hasher&) = delete;

    void init() override { check(vt_->stream_init(vt_->self, stream_), "Failed to initialize SHA256"); }
    void update(std::span<const std::byte> data) override {
        check(vt_->stream_update(vt_->self, stream_, reinterpret_cast<const uint8_t*>(data.data()), data.size()),
              "Failed to update SHA256");
    }
    std::string finalize() override { // re-initialises for reuse, sha256_hasher.cpp:103-106
        char hex[65];
        check(vt_->stream_finalize(vt_->self, stream_, hex), "Failed to finalize SHA256");
        return std::string(hex, 64);
    }
    // sha256_hasher.cpp:111-150.  The reference streams 64 KiB reads through update(); a device has
    // nothing to gain from a launch per read, so the file is read whole and hashed with ONE upload
    // and ONE kernel (content_hash_v1.hash).  A single SHA-256 chain is sequential by definition: one
    // file at a time is a job for the host's hasher — the device pays off on batches (hashMany,
    // hashFiles, chunker_v1, yams_ingest_device); see INTEGRATION.md.
    std::string hashFile(const std::filesystem::path& path) override {
        std::ifstream file(path, std::ios::binary);
        if (!file) throw std::runtime_error("Failed to open file: " + path.string());
        const uint64_t fileSize = std::filesystem::file_size(path);
        std::vector<std::byte> data(static_cast<size_t>(fileSize));
        uint64_t processed = 0;
        while (file && processed < fileSize) {
            const size_t want = static_cast<size_t>(std::min<uint64_t>(fileSize - processed, 64u << 10)); // the reference's read size (:120)
            file.read(reinterpret_cast<char*>(data.data() + processed), static_cast<std::streamsize>(want));
            const auto n = file.gcount();
            if (n <= 0) break;
            processed += static_cast<uint64_t>(n);
            if (progress_) progress_(processed, fileSize); // sha256_hasher.cpp:136-138
        }
        data.resize(static_cast<size_t>(processed));
        return hash(std::span<const std::byte>(data.data(), data.size()));
    }
    // Many files per call: read them all, one device call for all digests.
    std::vector<std::string> hashFiles(const std::vector<std::filesystem::path>& paths) {
        std::vector<std::vector<std::byte>> bufs(paths.size());
        std::vector<std::span<const std::byte>> spans;
        for (size_t i = 0; i < paths.size(); ++i) {
            std::ifstream file(paths[i], std::ios::binary);
            if (!file) throw std::runtime_error("Failed to open file: " + paths[i].string());
            bufs[i].resize(static_cast<size_t>(std::filesystem::file_size(paths[i])));
            file.read(reinterpret_cast<char*>(bufs[i].data()), static_cast<std::streamsize>(bufs[i].size()));
            bufs[i].resize(static_cast<size_t>(file.gcount()));
            spans.emplace_back(bufs[i].data(), bufs[i].size());
        }
        return hashMany(spans);
    }
    // sha256_hasher.cpp:152-161: any failure becomes ErrorCode::FileNotFound
    std::future<Result<std::string>> hashFileAsync(const std::filesystem::path& path) override {
        return std::async(std::launch::async, [this, path]() -> Result<std::string> {
            try { return hashFile(path); }
            catch (const std::exception&) { return Error{ErrorCode::FileNotFound, "hashFileAsync failed"}; }
        });
    }
    void setProgressCallback(ProgressCallback callback) override { progress_ = std::move(callback); }
    // SHA256Hasher::hash(span) one-shot, sha256_hasher.cpp:167-195
    std::string hash(std::span<const std::byte> data) {
        char hex[65];
        check(vt_->hash(vt_->self, reinterpret_cast<const uint8_t*>(data.data()), data.size(), hex),
              "Failed to hash");
        return std::string(hex, 64);
    }
    // Many buffers per call — the shape that suits a GPU (one message per lane).
    std::vector<std::string> hashMany(const std::vector<std::span<const std::byte>>& msgs) {
        std::vector<const uint8_t*> ptrs; std::vector<size_t> lens;
        for (auto& m : msgs) { ptrs.push_back(reinterpret_cast<const uint8_t*>(m.data())); lens.push_back(m.size()); }
        std::vector<char> hex(msgs.size() * 65);
        check(vt_->hash_many(vt_->self, ptrs.data(), lens.data(), msgs.size(), hex.data()), "Failed to hash batch");
        std::vector<std::string> out;
        for (size_t i = 0; i < msgs.size(); ++i) out.emplace_back(hex.data() + 65 * i, 64);
        return out;
    }
private:
    static void check(yams_status_t st, const char* what) { if (st != YAMS_OK) throw std::runtime_error(what); }
    std::shared_ptr<accel::Plugin> plugin_;
    yams_content_hash_v1* vt_;
    void* stream_ = nullptr;
    ProgressCallback progress_;
};

inline Result<std::unique_ptr<AccelSHA256Hasher>> createAccelSHA256Hasher(std::shared_ptr<accel::Plugin> plugin) {
    auto vt = plugin->getInterface<yams_content_hash_v1>(YAMS_IFACE_CONTENT_HASH_V1, YAMS_IFACE_CONTENT_HASH_V1_VERSION);
    if (!vt) return vt.error();
    return std::make_unique<AccelSHA256Hasher>(std::move(plugin), vt.value());
}

} // namespace yams::crypto
