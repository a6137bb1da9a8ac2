// chunker.hpp — yams::chunking::IChunker over the chunker_v1 vtable.
// Mirrors include/yams/chunking/chunker.h:18-172 of the reference (Chunk, ChunkRef,
// ChunkingConfig, IChunker) so it can be handed to ContentStoreBuilder::withChunker
// (src/api/content_store_builder.cpp:433-441).  chunkData/chunkFile are re-entrant (the vtable
// serialises on the device), as ContentStore shares one chunker across workers
// (src/api/content_store_impl.cpp:1412).
#pragma once
#include <cstddef>
#include <cstdint>
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
#include <yams/chunking/chunker.h> // the host's own Chunk / ChunkRef / ChunkingConfig / IChunker (chunker.h:18-92)
#endif

namespace yams::chunking {

#ifndef YAMS_ACCEL_USE_HOST_TYPES
// restated for builds outside the YAMS tree
struct Chunk { // chunker.h:18-30
    std::vector<std::byte> data;
    Hash hash;
    size_t offset = 0;
    size_t size = 0;
};
struct ChunkRef { Hash hash; size_t offset = 0; size_t size = 0; }; // chunker.h:32-41

struct ChunkingConfig { // chunker.h:44-51; defaults core/types.h:280-285
    size_t windowSize = 48;
    size_t minChunkSize = 16 * 1024;
    size_t targetChunkSize = 256 * 1024; // not part of the boundary logic
    size_t maxChunkSize = 1024 * 1024;
    uint64_t polynomial = 0x3DA3358B4DC173ULL;
    uint64_t chunkMask = 0x1FFF;
};

class IChunker { // chunker.h:65-92
public:
    virtual ~IChunker() = default;
    virtual const ChunkingConfig& getConfig() const = 0;
    virtual std::vector<Chunk> chunkFile(const std::filesystem::path& path) = 0;
    virtual std::vector<Chunk> chunkData(std::span<const std::byte> data) = 0;
    virtual std::vector<Chunk> chunkDataLazy(std::span<const std::byte> data) { return chunkData(data); }
    virtual std::future<Result<std::vector<Chunk>>> chunkFileAsync(const std::filesystem::path& path) = 0;
    using ProgressCallback = std::function<void(uint64_t, uint64_t)>;
    virtual void setProgressCallback(ProgressCallback callback) = 0;
};

#endif

enum class AccelChunkerKind { Rabin, Streaming };

class AccelChunker final : public IChunker {
public:
    AccelChunker(std::shared_ptr<accel::Plugin> plugin, yams_chunker_v1* vt, AccelChunkerKind kind,
                 ChunkingConfig config = {})
        : plugin_(std::move(plugin)), vt_(vt), kind_(kind), config_(std::move(config)) {
        if (config_.polynomial == 0) config_.polynomial = 0x3DA3358B4DC173ULL; // rabin_chunker.cpp:29-37
    }
    const ChunkingConfig& getConfig() const override { return config_; }
    std::vector<Chunk> chunkData(std::span<const std::byte> data) override { return run(data, false); }
    std::vector<Chunk> chunkDataLazy(std::span<const std::byte> data) override { return run(data, true); }
    std::vector<Chunk> chunkFile(const std::filesystem::path& path) override { // rabin_chunker.cpp:154-180
        std::ifstream file(path, std::ios::binary);
        if (!file) throw std::runtime_error("Failed to open file: " + path.string());
        file.seekg(0, std::ios::end);
        const auto endPos = file.tellg();
        if (endPos < std::streampos{0}) throw std::runtime_error("Failed to determine file size");
        std::vector<std::byte> data(static_cast<size_t>(endPos));
        file.seekg(0, std::ios::beg);
        file.read(reinterpret_cast<char*>(data.data()), static_cast<std::streamsize>(data.size()));
        if (!file && !data.empty()) throw std::runtime_error("Failed to read file");
        return chunkData(data);
    }
    // rabin_chunker.cpp:182-192 / streaming_chunker.cpp:139-150: any failure -> FileNotFound
    std::future<Result<std::vector<Chunk>>> chunkFileAsync(const std::filesystem::path& path) override {
        return std::async(std::launch::async, [this, path]() -> Result<std::vector<Chunk>> {
            try { return chunkFile(path); }
            catch (const std::exception&) { return Error{ErrorCode::FileNotFound, "chunkFileAsync failed"}; }
        });
    }
    // Called with (end offset of the chunk, total bytes) once per emitted chunk, in order — the
    // sequence RabinChunker produces (rabin_chunker.cpp:144-147); the boundaries all come back from
    // one device call, so the calls happen after it.
    void setProgressCallback(ProgressCallback callback) override { progress_ = std::move(callback); }
private:
    std::vector<Chunk> run(std::span<const std::byte> data, bool lazy) {
        yams_cdc_config_t cfg{};
        cfg.window_size = config_.windowSize; cfg.min_size = config_.minChunkSize;
        cfg.max_size = config_.maxChunkSize; cfg.polynomial = config_.polynomial;
        cfg.mask = config_.chunkMask;
        cfg.mode = kind_ == AccelChunkerKind::Streaming ? YAMS_CDC_STREAMING : YAMS_CDC_RABIN;
        yams_chunk_ref_t* refs = nullptr; size_t n = 0;
        const yams_status_t st = vt_->chunk_data(vt_->self, reinterpret_cast<const uint8_t*>(data.data()),
                                                 data.size(), &cfg, &refs, &n);
        if (st != YAMS_OK) throw std::runtime_error("Failed to chunk data on the accelerator");
        std::vector<Chunk> chunks(n);
        for (size_t i = 0; i < n; ++i) {
            chunks[i].offset = refs[i].offset; chunks[i].size = refs[i].size;
            chunks[i].hash.assign(refs[i].hash_hex, 64);
            if (!lazy) { auto s = data.subspan(chunks[i].offset, chunks[i].size); chunks[i].data.assign(s.begin(), s.end()); }
        }
        vt_->free_chunks(vt_->self, refs, n); // paired free, never host free() (model_provider_v1.h:46-49)
        if (progress_)
            for (const auto& c : chunks) progress_(c.offset + c.size, data.size());
        return chunks;
    }
    std::shared_ptr<accel::Plugin> plugin_;
    yams_chunker_v1* vt_;
    AccelChunkerKind kind_;
    ChunkingConfig config_;
    ProgressCallback progress_;
};

inline Result<std::unique_ptr<IChunker>> createAccelChunker(std::shared_ptr<accel::Plugin> plugin,
                                                            AccelChunkerKind kind, ChunkingConfig config = {}) {
    auto vt = plugin->getInterface<yams_chunker_v1>(YAMS_IFACE_CHUNKER_V1, YAMS_IFACE_CHUNKER_V1_VERSION);
    if (!vt) return vt.error();
    return std::unique_ptr<IChunker>(new AccelChunker(std::move(plugin), vt.value(), kind, std::move(config)));
}

} // namespace yams::chunking
