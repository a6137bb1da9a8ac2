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
#include <mutex>
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

// One SHA-256 chain is sequential: a device lane advances it at ~35 MB/s, a host core with SHA-NI at > 1 GB/s.
// The device is for MANY chains at once (hashMany / hashFiles / chunker_v1 / yams_ingest_*).  The vtable therefore
// REFUSES lone long chains (YAMS_ERR_UNSUPPORTED, see content_hash_v1 in yams_mi355x_accel.h) and this adapter hands
// them to the hasher the HOST supplies — its own SHA256Hasher (src/crypto/sha256_hasher.cpp) — which also serves the
// streaming calls and hashFile.  Without a host hasher everything still works through the plugin's streaming door:
// correct, slow, and documented as such.
class AccelSHA256Hasher final : public IContentHasher {
public:
    // hostHasher serves THIS object's streaming calls (init / update / finalize / hashFile).  oneShotHasher, a SECOND
    // instance of the host's hasher, serves the one-shot chains the device refuses: the reference's
    // SHA256Hasher::hash(span) is static and stateless (sha256_hasher.cpp:167-195), so a one-shot hash() between two
    // update() calls must not touch the chain in progress, and concurrent hash() calls must not share a context
    // unguarded.  Without one, refused chains take a private stream on the device (correct, slow).
    AccelSHA256Hasher(std::shared_ptr<accel::Plugin> plugin, yams_content_hash_v1* vt,
                      std::unique_ptr<IContentHasher> hostHasher = nullptr, std::unique_ptr<IContentHasher> oneShotHasher = nullptr)
        : plugin_(std::move(plugin)), vt_(vt), host_(std::move(hostHasher)), oneShot_(std::move(oneShotHasher)) {
        if (!host_ && vt_->stream_create(vt_->self, &stream_) != YAMS_OK)
            throw std::runtime_error("Failed to create SHA256 stream on the accelerator");
    }
    ~AccelSHA256Hasher() override { if (stream_) vt_->stream_destroy(vt_->self, stream_); }
    AccelSHA256Hasher(const AccelSHA256Hasher&) = delete;
    AccelSHA256Hasher& operator=(const AccelSHA256Hasher&) = delete;

    // ---- the streaming interface: one chain — the host's hasher when there is one --------------------------
    void init() override {
        if (host_) return host_->init();
        check(vt_->stream_init(vt_->self, stream_), "Failed to initialize SHA256");
    }
    void update(std::span<const std::byte> data) override {
        if (host_) return host_->update(data);
        check(vt_->stream_update(vt_->self, stream_, reinterpret_cast<const uint8_t*>(data.data()), data.size()),
              "Failed to update SHA256");
    }
    std::string finalize() override { // re-initialises for reuse, sha256_hasher.cpp:103-106
        if (host_) return host_->finalize();
        char hex[65];
        check(vt_->stream_finalize(vt_->self, stream_, hex), "Failed to finalize SHA256");
        return std::string(hex, 64);
    }
    // sha256_hasher.cpp:111-150: reads until EOF (not until file_size(): growing and special files hash as the
    // reference hashes them) in bounded pieces through update(); progress after every read (:136-138).
    std::string hashFile(const std::filesystem::path& path) override {
        if (host_) { host_->setProgressCallback(progress_); return host_->hashFile(path); }
        std::ifstream file(path, std::ios::binary);
        if (!file) throw std::runtime_error("Failed to open file: " + path.string());
        std::error_code ec;
        const uint64_t fileSize = std::filesystem::file_size(path, ec); // for the progress callback only
        init();
        std::vector<std::byte> buf(size_t(1) << 20);
        uint64_t processed = 0;
        while (file.read(reinterpret_cast<char*>(buf.data()), static_cast<std::streamsize>(buf.size())) || file.gcount() > 0) {
            const auto n = static_cast<size_t>(file.gcount());
            update(std::span<const std::byte>(buf.data(), n));
            processed += n;
            if (progress_) progress_(processed, ec ? processed : fileSize);
        }
        return finalize();
    }
    // Many files per call: groups of at most ~maxBytes are read and hashed with ONE device call each; a group the
    // device refuses (a lone long chain in it) goes through hash() file by file.
    std::vector<std::string> hashFiles(const std::vector<std::filesystem::path>& paths, size_t maxBytes = size_t(1) << 30) {
        std::vector<std::string> out;
        size_t i = 0;
        while (i < paths.size()) {
            std::vector<std::vector<std::byte>> bufs;
            size_t bytes = 0;
            while (i < paths.size() && (bufs.empty() || bytes < maxBytes)) {
                std::ifstream file(paths[i], std::ios::binary);
                if (!file) throw std::runtime_error("Failed to open file: " + paths[i].string());
                std::vector<std::byte> data;
                char block[64 << 10];
                while (file.read(block, sizeof block) || file.gcount() > 0)
                    data.insert(data.end(), reinterpret_cast<std::byte*>(block), reinterpret_cast<std::byte*>(block) + file.gcount());
                bytes += data.size();
                bufs.push_back(std::move(data));
                ++i;
            }
            std::vector<std::span<const std::byte>> spans;
            for (auto& b : bufs) spans.emplace_back(b.data(), b.size());
            for (auto& h : hashMany(spans)) out.push_back(std::move(h));
        }
        return out;
    }
    // sha256_hasher.cpp:152-161: any failure becomes ErrorCode::FileNotFound
    std::future<Result<std::string>> hashFileAsync(const std::filesystem::path& path) override {
        return std::async(std::launch::async, [this, path]() -> Result<std::string> {
            try { return hashFile(path); }
            catch (const std::exception&) { return Error{ErrorCode::FileNotFound, "hashFileAsync failed"}; }
        });
    }
    void setProgressCallback(ProgressCallback callback) override { progress_ = std::move(callback); }
    // SHA256Hasher::hash(span) one-shot, sha256_hasher.cpp:167-195.  Short messages: one device call.  A long lone
    // chain is refused by the device (YAMS_ERR_UNSUPPORTED): the host's hasher takes it, else the streaming door.
    std::string hash(std::span<const std::byte> data) {
        char hex[65];
        const yams_status_t st = vt_->hash(vt_->self, reinterpret_cast<const uint8_t*>(data.data()), data.size(), hex);
        if (st == YAMS_ERR_UNSUPPORTED) return oneChain(data);
        check(st, "Failed to hash");
        return std::string(hex, 64);
    }
    // Many buffers per call — the shape that suits a GPU (one message per lane).
    std::vector<std::string> hashMany(const std::vector<std::span<const std::byte>>& msgs) {
        std::vector<const uint8_t*> ptrs; std::vector<size_t> lens;
        for (auto& m : msgs) { ptrs.push_back(reinterpret_cast<const uint8_t*>(m.data())); lens.push_back(m.size()); }
        std::vector<char> hex(msgs.size() * 65);
        const yams_status_t st = vt_->hash_many(vt_->self, ptrs.data(), lens.data(), msgs.size(), hex.data());
        std::vector<std::string> out;
        if (st == YAMS_ERR_UNSUPPORTED) { // a batch dominated by one long chain: every message on its own
            for (auto& m : msgs) out.push_back(hash(m));
            return out;
        }
        check(st, "Failed to hash batch");
        for (size_t i = 0; i < msgs.size(); ++i) out.emplace_back(hex.data() + 65 * i, 64);
        return out;
    }
    bool hasHostHasher() const { return host_ != nullptr; }
    bool hasOneShotHostHasher() const { return oneShot_ != nullptr; }
private:
    std::string oneChain(std::span<const std::byte> data) {
        if (oneShot_) { // never host_: that one carries the chain of init() / update() / finalize()
            std::lock_guard<std::mutex> lk(oneShotMu_);
            oneShot_->init(); oneShot_->update(data); return oneShot_->finalize();
        }
        // no host hasher for one-shot chains: a private stream on the device (the handle of init/update/finalize stays untouched)
        void* st = nullptr;
        check(vt_->stream_create(vt_->self, &st), "Failed to create SHA256 stream on the accelerator");
        char hex[65];
        yams_status_t rc = vt_->stream_update(vt_->self, st, reinterpret_cast<const uint8_t*>(data.data()), data.size());
        if (rc == YAMS_OK) rc = vt_->stream_finalize(vt_->self, st, hex);
        vt_->stream_destroy(vt_->self, st);
        check(rc, "Failed to hash");
        return std::string(hex, 64);
    }
    static void check(yams_status_t st, const char* what) { if (st != YAMS_OK) throw std::runtime_error(what); }
    std::shared_ptr<accel::Plugin> plugin_;
    yams_content_hash_v1* vt_;
    std::unique_ptr<IContentHasher> host_, oneShot_;
    std::mutex oneShotMu_;
    void* stream_ = nullptr;
    ProgressCallback progress_;
};

// hostHasher / oneShotHasher: two instances of the host's own IContentHasher (e.g. crypto::createSHA256Hasher()) — the
// first for this object's streaming chain, the second for the one-shot chains the device refuses; either may be null.
inline Result<std::unique_ptr<AccelSHA256Hasher>> createAccelSHA256Hasher(std::shared_ptr<accel::Plugin> plugin,
                                                                          std::unique_ptr<IContentHasher> hostHasher = nullptr,
                                                                          std::unique_ptr<IContentHasher> oneShotHasher = nullptr) {
    auto vt = plugin->getInterface<yams_content_hash_v1>(YAMS_IFACE_CONTENT_HASH_V1, YAMS_IFACE_CONTENT_HASH_V1_VERSION);
    if (!vt) return vt.error();
    return std::make_unique<AccelSHA256Hasher>(std::move(plugin), vt.value(), std::move(hostHasher), std::move(oneShotHasher));
}
// The usual form: a factory for the host's hasher, called twice.
inline Result<std::unique_ptr<AccelSHA256Hasher>> createAccelSHA256Hasher(std::shared_ptr<accel::Plugin> plugin,
                                                                          const std::function<std::unique_ptr<IContentHasher>()>& hostHasherFactory) {
    return createAccelSHA256Hasher(std::move(plugin), hostHasherFactory ? hostHasherFactory() : nullptr,
                                   hostHasherFactory ? hostHasherFactory() : nullptr);
}

} // namespace yams::crypto
