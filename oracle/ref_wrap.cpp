// oracle/ref_wrap.cpp — C-linkage doors into the REFERENCE's own translation units, compiled
// from /root/reference where they lie (see oracle/Makefile).  TEST INFRASTRUCTURE ONLY: used to
// pin oracle/yams_oracle.c and (bench.py cpu_baseline.kind == "reference") as the timed CPU leg.
// No reference source is copied into this repository; this file only calls the public classes
// declared in include/yams/crypto/hasher.h and include/yams/chunking/{chunker,streaming_chunker}.h.
#include <yams/chunking/chunker.h>
#include <yams/chunking/streaming_chunker.h>
#include <yams/crypto/hasher.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <span>
#include <vector>

namespace {
yams::chunking::ChunkingConfig makeConfig(uint64_t window, uint64_t minSize, uint64_t maxSize,
                                          uint64_t polynomial, uint64_t mask) {
    yams::chunking::ChunkingConfig cfg;
    cfg.windowSize = static_cast<size_t>(window);
    cfg.minChunkSize = static_cast<size_t>(minSize);
    cfg.maxChunkSize = static_cast<size_t>(maxSize);
    cfg.polynomial = polynomial;
    cfg.chunkMask = mask;
    return cfg;
}
size_t emit(const std::vector<yams::chunking::Chunk>& chunks, uint64_t* offsets, uint64_t* sizes,
            char* hex, size_t cap) {
    for (size_t i = 0; i < chunks.size() && i < cap; ++i) {
        offsets[i] = chunks[i].offset;
        sizes[i] = chunks[i].size;
        if (hex) std::memcpy(hex + 65 * i, chunks[i].hash.c_str(), 65);
    }
    return chunks.size();
}
} // namespace

extern "C" {
__attribute__((visibility("default"))) void ref_sha256_hex(const uint8_t* data, size_t n,
                                                           char out[65]) {
    auto h = yams::crypto::SHA256Hasher::hash(
        std::span<const std::byte>(reinterpret_cast<const std::byte*>(data), n));
    std::memcpy(out, h.c_str(), 65);
}
// Streaming interface (init/update/finalize) with an explicit split point list.
__attribute__((visibility("default"))) void ref_sha256_hex_split(const uint8_t* data, size_t n,
                                                                 const size_t* cuts, size_t ncuts,
                                                                 char out[65]) {
    yams::crypto::SHA256Hasher hasher;
    hasher.init();
    size_t prev = 0;
    for (size_t i = 0; i <= ncuts; ++i) {
        size_t end = (i < ncuts) ? cuts[i] : n;
        hasher.update(std::span<const std::byte>(
            reinterpret_cast<const std::byte*>(data) + prev, end - prev));
        prev = end;
    }
    auto h = hasher.finalize();
    std::memcpy(out, h.c_str(), 65);
}
// RabinChunker::chunkDataLazy (offset, size, per-chunk SHA-256 hex).
__attribute__((visibility("default"))) size_t
ref_chunk_rabin(const uint8_t* data, size_t n, uint64_t window, uint64_t minSize, uint64_t maxSize,
                uint64_t polynomial, uint64_t mask, uint64_t* offsets, uint64_t* sizes, char* hex,
                size_t cap) {
    yams::chunking::RabinChunker chunker(makeConfig(window, minSize, maxSize, polynomial, mask));
    auto chunks = chunker.chunkDataLazy(
        std::span<const std::byte>(reinterpret_cast<const std::byte*>(data), n));
    return emit(chunks, offsets, sizes, hex, cap);
}
// StreamingChunker::chunkData — the product default chunker.
__attribute__((visibility("default"))) size_t
ref_chunk_streaming(const uint8_t* data, size_t n, uint64_t window, uint64_t minSize,
                    uint64_t maxSize, uint64_t polynomial, uint64_t mask, uint64_t* offsets,
                    uint64_t* sizes, char* hex, size_t cap) {
    yams::chunking::StreamingChunker chunker(
        makeConfig(window, minSize, maxSize, polynomial, mask));
    auto chunks = chunker.chunkData(
        std::span<const std::byte>(reinterpret_cast<const std::byte*>(data), n));
    return emit(chunks, offsets, sizes, hex, cap);
}
// The reference's synthetic-embedding recipe drawn with the REAL std::mt19937 and
// std::uniform_real_distribution<float> of this toolchain's libstdc++ (the classes
// tests/benchmarks/vector_backend_engine_compare.cpp:83-107 uses); pins oracle_mt19937_rows.
__attribute__((visibility("default"))) void ref_mt19937_rows(uint32_t seed, size_t count, size_t dim,
                                                             float* out) {
    std::mt19937 rng(seed);
    for (size_t r = 0; r < count; ++r) {
        std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
        float* v = out + r * dim;
        float norm_sq = 0.0f;
        for (size_t j = 0; j < dim; ++j) {
            v[j] = dist(rng);
            norm_sq += v[j] * v[j];
        }
        const float norm = std::sqrt(norm_sq);
        if (norm > 0.0f)
            for (size_t j = 0; j < dim; ++j) v[j] /= norm;
    }
}
}
