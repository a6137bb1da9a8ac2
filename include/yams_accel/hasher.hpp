// hasher.hpp — yams::crypto::IContentHasher over the content_hash_v1 vtable.
// Mirrors include/yams/crypto/hasher.h:14-80 of the reference (init / update / finalize /
// hashFile / static one-shot hash) so it can be handed to ContentStoreBuilder::withHasher
// (src/api/content_store_builder.cpp:433-441).  Instances are single-threaded objects, like the
// reference's (src/crypto/sha256_hasher.cpp:34).
#pragma once
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

namespace yams::crypto {

class IContentHasher { // hasher.h:14-47 (reference)
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
    std::string hashFile(const std::filesystem::path& path) override { // sha256_hasher.cpp:111-150
        std::ifstream file(path, std::ios::binary);
        if (!file) throw std::runtime_error("Failed to open file: " + path.string());
        init();
        std::vector<std::byte> buffer(1 << 20); // larger reads than the reference's 64 KiB: one launch each
        const uint64_t fileSize = std::filesystem::file_size(path);
        uint64_t processed = 0;
        while (file) {
            file.read(reinterpret_cast<char*>(buffer.data()), static_cast<std::streamsize>(buffer.size()));
            const auto n = file.gcount();
            if (n > 0) {
                update(std::span<const std::byte>(buffer.data(), static_cast<size_t>(n)));
                processed += static_cast<uint64_t>(n);
                if (progress_) progress_(processed, fileSize); // sha256_hasher.cpp:136-138
            }
        }
        return finalize();
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
