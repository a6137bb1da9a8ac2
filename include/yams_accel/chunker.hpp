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
#include <istream>
#include <memory>
#include <span>
#include <string>
#include <vector>

#include <atomic>
#include <exception>
#include <thread>

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

    // MANY buffers per call (chunker_v1.chunk_many): what ContentStore::store needs for a batch of files
    // (content_store_impl.cpp:199-231) — chunk lists with per-chunk hashes and, if asked for, the whole-buffer
    // (file) hashes — from ONE device call that runs at the batched ingest rate instead of one launch sequence
    // per file.  A plugin build without chunk_many (NULL entry: "not implemented",
    // abi_model_provider_adapter.cpp:121-122) is served buffer by buffer through chunk_data.
    struct BatchResult {
        std::vector<std::vector<Chunk>> chunks;   // per buffer, in order
        std::vector<Hash> bufferHashes;           // per buffer; empty unless asked for
    };
    // The host's own one-shot SHA-256 (e.g. a lambda around crypto::SHA256Hasher::hash).  With one, chunkMany asks the
    // device NOT to compute the whole-buffer hash of buffers whose chain would outlast the rest of the call
    // (YAMS_CHUNK_MANY_DEFER_LONG_BUFFER_HASHES: one 64 MiB file in a batch of small ones holds the device for 1.9 s) and
    // hashes those on host threads WHILE the device call runs.  Thread-safe callables only.
    using HostHash = std::function<std::string(std::span<const std::byte>)>;
    void setHostHash(HostHash h, unsigned maxThreads = 0) { hostHash_ = std::move(h); hostThreads_ = maxThreads; }
    BatchResult chunkMany(const std::vector<std::span<const std::byte>>& buffers, bool withBufferHashes = false, bool lazy = true) {
        BatchResult out;
        out.chunks.resize(buffers.size());
        const bool batched = vt_->abi_version >= 2 && vt_->chunk_many && vt_->free_chunk_batch;
        if (!batched) {
            if (withBufferHashes) throw std::runtime_error("this chunker_v1 build has no chunk_many: buffer hashes need a hasher");
            for (size_t b = 0; b < buffers.size(); ++b) out.chunks[b] = run(buffers[b], lazy);
            return out;
        }
        const yams_cdc_config_t cfg = config();
        std::vector<const uint8_t*> ptrs(buffers.size());
        std::vector<size_t> lens(buffers.size());
        for (size_t b = 0; b < buffers.size(); ++b) { ptrs[b] = reinterpret_cast<const uint8_t*>(buffers[b].data()); lens[b] = buffers[b].size(); }
        yams_chunk_batch_t* batch = nullptr;
        // buffers the device will leave to the host (the same predicate the plugin applies): started NOW, joined below
        const bool defer = withBufferHashes && static_cast<bool>(hostHash_);
        std::vector<size_t> deferred;
        std::vector<std::string> deferredHash;
        std::vector<std::future<void>> workers;
        if (defer) {
            uint64_t total = 0;
            for (size_t l : lens) total += l;
            const uint64_t above = yams_ingest_defer_threshold_host(total);
            for (size_t b = 0; b < buffers.size(); ++b) if (lens[b] > above) deferred.push_back(b);
            deferredHash.resize(deferred.size());
            if (!deferred.empty()) {
                unsigned nt = hostThreads_ ? hostThreads_ : std::max(1u, std::thread::hardware_concurrency());
                nt = static_cast<unsigned>(std::min<size_t>(nt, deferred.size()));
                auto next = std::make_shared<std::atomic<size_t>>(0);
                for (unsigned t = 0; t < nt; ++t)
                    workers.push_back(std::async(std::launch::async, [&, next] {
                        for (size_t j; (j = next->fetch_add(1)) < deferred.size();) deferredHash[j] = hostHash_(buffers[deferred[j]]);
                    }));
            }
        }
        const yams_status_t st = vt_->chunk_many(vt_->self, ptrs.data(), lens.data(), buffers.size(), &cfg,
                                                 (withBufferHashes ? YAMS_CHUNK_MANY_BUFFER_HASHES : 0u) |
                                                     (defer ? YAMS_CHUNK_MANY_DEFER_LONG_BUFFER_HASHES : 0u), &batch);
        std::exception_ptr hostError; // (workers are joined on every path; an exception of the host hasher surfaces here)
        for (auto& w : workers) try { w.get(); } catch (...) { if (!hostError) hostError = std::current_exception(); }
        if (hostError) { if (batch) vt_->free_chunk_batch(vt_->self, batch); std::rethrow_exception(hostError); }
        if (st != YAMS_OK || !batch) throw std::runtime_error("Failed to chunk the batch on the accelerator");
        try {
            for (size_t b = 0; b < buffers.size(); ++b) {
                auto& dst = out.chunks[b];
                dst.resize(batch->first_chunk[b + 1] - batch->first_chunk[b]);
                for (size_t i = 0; i < dst.size(); ++i) {
                    const yams_chunk_ref_t& r = batch->chunks[batch->first_chunk[b] + i];
                    dst[i].offset = r.offset; dst[i].size = r.size; dst[i].hash.assign(r.hash_hex, 64);
                    if (!lazy) { auto s = buffers[b].subspan(dst[i].offset, dst[i].size); dst[i].data.assign(s.begin(), s.end()); }
                }
                if (withBufferHashes) out.bufferHashes.emplace_back(batch->buffer_hash_hex + 65 * b, batch->buffer_hash_hex[65 * b] ? 64 : 0);
            }
            for (size_t j = 0; j < deferred.size(); ++j) out.bufferHashes[deferred[j]] = deferredHash[j];
            if (withBufferHashes)
                for (auto& h : out.bufferHashes) if (h.size() != 64) throw std::runtime_error("a buffer hash is missing from the batch");
        } catch (...) { vt_->free_chunk_batch(vt_->self, batch); throw; } // the paired free, also on the exception path (:131-149)
        vt_->free_chunk_batch(vt_->self, batch);
        return out;
    }
    // ---- the bounded-memory callback form (StreamingChunker::processStream / processFileStream,
    //      include/yams/chunking/streaming_chunker.h:54-121) ---------------------------------------------------------
    // The stream is consumed in windows of `windowBytes` (read in 64 KiB pieces exactly as the reference reads: state
    // cleared, read(), gcount() until it returns 0 — short reads of segmenting streambufs included).  Each window is
    // ONE device pass (chunker_v1.chunk_window): boundaries + per-chunk SHA-256; every complete chunk goes to
    // `processor(ChunkRef, bytes)` in stream order with offsets relative to the stream.  What crosses a window: the
    // open chunk (the last one of a window ends where the window ends, boundary or not — it is carried and chunked
    // again with the bytes that follow) and 64 bytes in front of it, from which the device rebuilds the rolling hash
    // exactly (its state is a function of the last 56 bytes; StreamingChunker never resets it at a chunk boundary).
    // Host memory: windowBytes + maxChunkSize + 64, whatever the stream's length.  The progress callback is called
    // with (bytes read so far, totalSize) after every read when totalSize > 0 (streaming_chunker.h:103-106) — reads of a
    // window come before its chunks here, where the reference interleaves them.  Streaming kind only: RabinChunker has
    // no such entry point in the reference (and restarts its hash at every chunk).
    template <class F>
    Result<void> processStream(std::istream& stream, size_t totalSize, F&& processor, size_t windowBytes = size_t(32) << 20) {
        if (kind_ != AccelChunkerKind::Streaming)
            return Error{ErrorCode::InvalidOperation, "processStream is the StreamingChunker's entry point"};
        if (!(vt_->abi_version >= 3 && vt_->chunk_window))
            return Error{ErrorCode::NotSupported, "this chunker_v1 build has no chunk_window"};
        constexpr size_t kRead = 64 * 1024, kHistory = 64;
        if (windowBytes < kRead) windowBytes = kRead;
        const yams_cdc_config_t cfg = config();
        std::vector<std::byte> buf;     // [history: h bytes][the open chunk so far + what was read since]
        size_t h = 0;                   // leading bytes of buf that are history only
        size_t base = 0;                // stream offset of buf[h]
        size_t consumed = 0;
        bool eof = false;
        while (!eof) {
            size_t got = 0;
            while (got < windowBytes) {
                const size_t at = buf.size();
                buf.resize(at + kRead);
                stream.clear(); // (a previous short read may have set failbit / eofbit: rely on gcount() alone)
                stream.read(reinterpret_cast<char*>(buf.data() + at), static_cast<std::streamsize>(kRead));
                const std::streamsize rc = stream.gcount();
                const size_t n = rc > 0 ? static_cast<size_t>(rc) : 0u;
                buf.resize(at + n);
                if (n == 0) { eof = true; break; }
                got += n; consumed += n;
                if (progress_ && totalSize > 0) progress_(consumed, totalSize);
            }
            if (buf.size() == h) break; // nothing pending
            yams_chunk_ref_t* refs = nullptr; size_t n = 0;
            const yams_status_t st = vt_->chunk_window(vt_->self, reinterpret_cast<const uint8_t*>(buf.data()), buf.size(), h, &cfg, &refs, &n);
            if (st != YAMS_OK) return Error{accel::mapStatus(st), "Failed to chunk a window of the stream on the accelerator"};
            size_t keepFrom = buf.size();
            try {
                const size_t emit = eof ? n : (n ? n - 1 : 0);
                for (size_t i = 0; i < emit; ++i) {
                    ChunkRef ref;
                    ref.hash.assign(refs[i].hash_hex, 64);
                    ref.offset = base + (static_cast<size_t>(refs[i].offset) - h);
                    ref.size = static_cast<size_t>(refs[i].size);
                    processor(static_cast<const ChunkRef&>(ref), std::span<const std::byte>(buf.data() + refs[i].offset, ref.size));
                }
                if (!eof && n) { // carry the open chunk and the history in front of it
                    const size_t open = static_cast<size_t>(refs[n - 1].offset);
                    keepFrom = open >= kHistory ? open - kHistory : 0; // (open < 64 only while buf still begins at the stream's first byte)
                    base += open - h;
                    h = open - keepFrom;
                }
            } catch (...) { vt_->free_chunks(vt_->self, refs, n); throw; }
            vt_->free_chunks(vt_->self, refs, n);
            if (!eof) buf.erase(buf.begin(), buf.begin() + static_cast<std::ptrdiff_t>(keepFrom));
        }
        return Result<void>();
    }
    template <class F>
    Result<void> processFileStream(const std::filesystem::path& path, F&& processor, size_t windowBytes = size_t(32) << 20) {
        std::ifstream file(path, std::ios::binary);
        if (!file) return Error{ErrorCode::FileNotFound, "Failed to open file: " + path.string()};
        file.seekg(0, std::ios::end);
        const auto pos = file.tellg();
        const size_t fileSize = pos == std::ios::pos_type(-1) ? 0 : static_cast<size_t>(pos); // (progress reporting only)
        file.seekg(0, std::ios::beg);
        return processStream(file, fileSize, std::forward<F>(processor), windowBytes);
    }

    // Files: read, then one chunkMany over all of them (bounded: files are grouped so that at most ~maxBytes are in
    // host memory at once).
    BatchResult chunkFiles(const std::vector<std::filesystem::path>& paths, bool withFileHashes = true,
                           size_t maxBytes = size_t(1) << 30) {
        BatchResult out;
        size_t i = 0;
        while (i < paths.size()) {
            std::vector<std::vector<std::byte>> bufs;
            size_t bytes = 0;
            while (i < paths.size() && (bufs.empty() || bytes < maxBytes)) {
                std::ifstream file(paths[i], std::ios::binary);
                if (!file) throw std::runtime_error("Failed to open file: " + paths[i].string());
                std::vector<std::byte> data;
                char block[64 << 10];
                while (file.read(block, sizeof block) || file.gcount() > 0) // until EOF, like the reference's readers
                    data.insert(data.end(), reinterpret_cast<std::byte*>(block), reinterpret_cast<std::byte*>(block) + file.gcount());
                bytes += data.size();
                bufs.push_back(std::move(data));
                ++i;
            }
            std::vector<std::span<const std::byte>> spans;
            for (auto& b : bufs) spans.emplace_back(b.data(), b.size());
            BatchResult part = chunkMany(spans, withFileHashes, /*lazy=*/false);
            for (auto& c : part.chunks) out.chunks.push_back(std::move(c));
            for (auto& h : part.bufferHashes) out.bufferHashes.push_back(std::move(h));
        }
        return out;
    }
private:
    yams_cdc_config_t config() const {
        yams_cdc_config_t cfg{};
        cfg.window_size = config_.windowSize; cfg.min_size = config_.minChunkSize;
        cfg.max_size = config_.maxChunkSize; cfg.polynomial = config_.polynomial;
        cfg.mask = config_.chunkMask;
        cfg.mode = kind_ == AccelChunkerKind::Streaming ? YAMS_CDC_STREAMING : YAMS_CDC_RABIN;
        return cfg;
    }
    std::vector<Chunk> run(std::span<const std::byte> data, bool lazy) {
        const yams_cdc_config_t cfg = config();
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
    HostHash hostHash_;
    unsigned hostThreads_ = 0;
};

inline Result<std::unique_ptr<IChunker>> createAccelChunker(std::shared_ptr<accel::Plugin> plugin,
                                                            AccelChunkerKind kind, ChunkingConfig config = {}) {
    // version 1 is all this adapter NEEDS (a plugin refuses versions above its own): what a newer vtable adds — chunk_many
    // (2), chunk_window (3) — is used when vt->abi_version says it is there, and has a fallback when it is not
    auto vt = plugin->getInterface<yams_chunker_v1>(YAMS_IFACE_CHUNKER_V1, 1);
    if (!vt) return vt.error();
    return std::unique_ptr<IChunker>(new AccelChunker(std::move(plugin), vt.value(), kind, std::move(config)));
}
// the concrete type, for hosts that batch (chunkMany / chunkFiles)
inline Result<std::unique_ptr<AccelChunker>> createAccelBatchChunker(std::shared_ptr<accel::Plugin> plugin,
                                                                     AccelChunkerKind kind, ChunkingConfig config = {}) {
    // version 1 is all this adapter NEEDS (a plugin refuses versions above its own): what a newer vtable adds — chunk_many
    // (2), chunk_window (3) — is used when vt->abi_version says it is there, and has a fallback when it is not
    auto vt = plugin->getInterface<yams_chunker_v1>(YAMS_IFACE_CHUNKER_V1, 1);
    if (!vt) return vt.error();
    return std::make_unique<AccelChunker>(std::move(plugin), vt.value(), kind, std::move(config));
}

} // namespace yams::chunking
