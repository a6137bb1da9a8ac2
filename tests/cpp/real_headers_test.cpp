// real_headers_test.cpp — the adapters compiled against the REFERENCE's own headers
// (-DYAMS_ACCEL_USE_HOST_TYPES -I/root/reference/include) and used only through the reference's
// own base classes: yams::crypto::IContentHasher (hasher.h:14-47), yams::chunking::IChunker
// (chunker.h:65-92), yams::vector::IVectorStore + the four capability seams found by dynamic_cast
// exactly as VectorDatabase::Impl does (vector_database.cpp:553-609).  Linked with the reference's
// OWN translation units (sha256_hasher.cpp, rabin_chunker.cpp, streaming_chunker.cpp — the recipe
// of oracle/Makefile), so every accelerated result is compared with the reference class next to it.
// Built only where /root/reference exists (tests/test_cpp_host.py); the binary travels to the GPU box.
// Usage: real_headers_test <path/to/libyams_mi355x_accel.so> [--expect-no-gpu]
#include <yams/chunking/chunker.h>
#include <yams/chunking/streaming_chunker.h>
#include <atomic>
#include <istream>
#include <streambuf>
#include <sstream>
#include <csignal>
#include <execinfo.h>
#include <thread>
#include <unistd.h>
#include <yams/crypto/hasher.h>
#include <yams/vector/vector_store.h>

#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <random>

#include "yams_accel/chunker.hpp"
#include "yams_accel/exact_scan_backend.hpp"
#include "yams_accel/hasher.hpp"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

using namespace yams;

static std::vector<float> unit(std::mt19937& rng, size_t dim) {
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> v(dim);
    float n = 0.f;
    for (auto& x : v) { x = d(rng); n += x * x; }
    n = std::sqrt(n);
    for (auto& x : v) x /= n;
    return v;
}
// the reference's exact arithmetic (sqlite_vec_backend.cpp:4253-4276): fp64 accumulate, cast to float
static float refCosine(const std::vector<float>& q, const std::vector<float>& x) {
    double dot = 0.0, nsq = 0.0, qsq = 0.0;
    for (size_t i = 0; i < q.size(); ++i) { nsq += double(x[i]) * x[i]; dot += double(x[i]) * q[i]; }
    for (float v : q) qsq += double(v) * v;
    return static_cast<float>(dot / (std::sqrt(nsq) * std::sqrt(qsq)));
}

// A durable IVectorStore for the transaction tests: rows in a map, a snapshot taken at BEGIN and put back at ROLLBACK
// (what SqliteVecBackend's BEGIN IMMEDIATE ... COMMIT / ROLLBACK amounts to for these calls).  It never searches.
struct FakeDurableStore final : vector::IVectorStore {
    std::map<std::string, vector::VectorRecord> rows, snapshot;
    bool inTxn = false, init = false;
    int failNextCommit = 0;
    Result<void> initialize(const std::string&) override { init = true; return {}; }
    void close() override { init = false; }
    bool isInitialized() const override { return init; }
    Result<void> createTables(size_t) override { return {}; }
    bool tablesExist() const override { return true; }
    Result<void> insertVector(const vector::VectorRecord& r) override { rows[r.chunk_id] = r; return {}; }
    Result<void> insertVectorsBatch(const std::vector<vector::VectorRecord>& rs) override { for (auto& r : rs) rows[r.chunk_id] = r; return {}; }
    Result<void> updateVector(const std::string& id, const vector::VectorRecord& r) override {
        if (!rows.count(id)) return Error{ErrorCode::NotFound, "chunk not found"};
        rows[id] = r; rows[id].chunk_id = id; return {};
    }
    Result<void> deleteVector(const std::string& id) override { rows.erase(id); return {}; }
    Result<void> deleteVectorsByDocument(const std::string& h) override {
        for (auto it = rows.begin(); it != rows.end();) it = it->second.document_hash == h ? rows.erase(it) : std::next(it);
        return {};
    }
    Result<std::vector<vector::VectorRecord>> searchSimilar(const std::vector<float>&, size_t, float, const std::optional<std::string>&,
                                                            const std::unordered_set<std::string>&,
                                                            const std::map<std::string, std::string>&) override {
        return Error{ErrorCode::NotImplemented, "the durable store does not search"};
    }
    Result<std::vector<std::vector<vector::VectorRecord>>> searchSimilarBatch(const std::vector<std::vector<float>>&, size_t, float, size_t) override {
        return Error{ErrorCode::NotImplemented, "the durable store does not search"};
    }
    Result<std::optional<vector::VectorRecord>> getVector(const std::string& id) override {
        auto it = rows.find(id);
        return it == rows.end() ? std::optional<vector::VectorRecord>{} : std::optional<vector::VectorRecord>{it->second};
    }
    Result<std::map<std::string, vector::VectorRecord>> getVectorsBatch(const std::vector<std::string>& ids) override {
        std::map<std::string, vector::VectorRecord> out;
        for (auto& id : ids) if (rows.count(id)) out[id] = rows[id];
        return out;
    }
    Result<std::vector<vector::VectorRecord>> getVectorsByDocument(const std::string& h) override {
        std::vector<vector::VectorRecord> out;
        for (auto& [id, r] : rows) if (r.document_hash == h) out.push_back(r);
        return out;
    }
    Result<std::unordered_map<std::string, vector::VectorRecord>> getDocumentLevelVectorsAll() override { return std::unordered_map<std::string, vector::VectorRecord>{}; }
    Result<size_t> forEachDocumentLevelVector(const std::function<bool(vector::VectorRecord&&)>&) override { return size_t(0); }
    Result<bool> hasEmbedding(const std::string& h) override { for (auto& [id, r] : rows) if (r.document_hash == h) return true; return false; }
    Result<std::unordered_set<std::string>> getEmbeddedDocumentHashes() override {
        std::unordered_set<std::string> out;
        for (auto& [id, r] : rows) out.insert(r.document_hash);
        return out;
    }
    Result<size_t> getVectorCount() override { return rows.size(); }
    Result<vector::VectorDatabaseStats> getStats() override { vector::VectorDatabaseStats st; st.total_vectors = rows.size(); return st; }
    Result<void> beginTransaction() override { if (inTxn) return Error{ErrorCode::InvalidState, "nested"}; inTxn = true; snapshot = rows; return {}; }
    Result<void> commitTransaction() override {
        if (failNextCommit) { --failNextCommit; return Error{ErrorCode::DatabaseError, "commit failed"}; }
        inTxn = false; return {};
    }
    Result<void> rollbackTransaction() override { if (inTxn) rows = snapshot; inTxn = false; return {}; }
};

int main(int argc, char** argv) {
    std::setvbuf(stdout, nullptr, _IONBF, 0); // (a crash must not swallow the failures printed before it)
    std::signal(SIGSEGV, [](int) { void* bt[48]; const int n = backtrace(bt, 48); backtrace_symbols_fd(bt, n, 1); _exit(139); });
    if (argc < 2) { std::printf("usage: %s <plugin.so> [--expect-no-gpu]\n", argv[0]); return 2; }
    const bool expectNoGpu = argc > 2 && std::strcmp(argv[2], "--expect-no-gpu") == 0;
    auto loaded = accel::Plugin::load(argv[1], "{\"device\":0}");
    if (expectNoGpu) {
        CHECK(!loaded.has_value());
        std::printf("%s\n", failures ? "FAILED" : "OK (refused without a GPU)");
        return failures ? 1 : 0;
    }
    if (!loaded) { std::printf("load failed: %s\n", loaded.error().message.c_str()); return 1; }
    auto plugin = loaded.value();
    static_assert(static_cast<int>(ErrorCode::Unknown) == 36, "the host's ErrorCode (core/types.h:25-63)");

    std::printf("[section] hasher\n");
    // ---- IContentHasher: the accelerator next to the reference's SHA256Hasher -----------------------------
    {
        auto made = crypto::createAccelSHA256Hasher(plugin);
        CHECK(made.has_value());
        std::unique_ptr<crypto::IContentHasher> acc = std::move(made.value());
        std::unique_ptr<crypto::IContentHasher> ref = std::make_unique<crypto::SHA256Hasher>();
        std::mt19937 rng(3);
        for (size_t n : {size_t(0), size_t(1), size_t(55), size_t(56), size_t(64), size_t(1000), size_t(70001), size_t(3 << 20)}) {
            std::vector<std::byte> data(n);
            for (auto& b : data) b = static_cast<std::byte>(rng());
            acc->init(); ref->init();
            size_t pos = 0;
            while (pos < n) { // ragged updates
                const size_t take = std::min<size_t>(n - pos, 1 + rng() % 50000);
                acc->update({data.data() + pos, take}); ref->update({data.data() + pos, take});
                pos += take;
            }
            CHECK(acc->finalize() == ref->finalize());
            CHECK(acc->hash(data) == crypto::SHA256Hasher::hash(std::span<const std::byte>(data)));   // the HashableData template of the base
        }
        const auto path = std::filesystem::temp_directory_path() / "yams_accel_real_headers.bin";
        { std::ofstream f(path, std::ios::binary); std::vector<char> d(5 * 1000 * 1000 + 3); for (auto& c : d) c = static_cast<char>(rng()); f.write(d.data(), d.size()); }
        uint64_t seen = 0;
        acc->setProgressCallback([&](uint64_t done, uint64_t total) { seen = done; CHECK(total == 5 * 1000 * 1000 + 3); });
        CHECK(acc->hashFile(path) == ref->hashFile(path));
        CHECK(seen == 5 * 1000 * 1000 + 3);
        auto fut = acc->hashFileAsync(path);
        CHECK(fut.get().value() == ref->hashFile(path));
        auto bad = acc->hashFileAsync(path.string() + ".missing").get();
        CHECK(!bad.has_value() && bad.error().code == ErrorCode::FileNotFound);   // sha256_hasher.cpp:152-161
        std::filesystem::remove(path);
    }
    std::printf("[section] lone chains\n");
    // ---- lone long chains are the HOST's: the vtable refuses them, the adapter hands them to the host's hasher ----
    {
        auto vt = plugin->getInterface<yams_content_hash_v1>(YAMS_IFACE_CONTENT_HASH_V1, YAMS_IFACE_CONTENT_HASH_V1_VERSION);
        CHECK(vt.has_value());
        std::mt19937 rng(5);
        std::vector<std::byte> big(3 << 20), small(4096);
        for (auto& b : big) b = static_cast<std::byte>(rng());
        for (auto& b : small) b = static_cast<std::byte>(rng());
        char hex[65];
        CHECK(vt.value()->hash(vt.value()->self, reinterpret_cast<const uint8_t*>(big.data()), big.size(), hex) == YAMS_ERR_UNSUPPORTED);
        CHECK(vt.value()->hash(vt.value()->self, reinterpret_cast<const uint8_t*>(small.data()), small.size(), hex) == YAMS_OK);
        // one 3 MiB message among a few small ones: the call could not finish before one host core would -> refused;
        // among 4096 small ones (16 MiB in total... still dominated) -> refused; 200 x 1 MiB: served
        {
            std::vector<const uint8_t*> ptrs{reinterpret_cast<const uint8_t*>(big.data())};
            std::vector<size_t> lens{big.size()};
            for (int i = 0; i < 8; ++i) { ptrs.push_back(reinterpret_cast<const uint8_t*>(small.data())); lens.push_back(small.size()); }
            std::vector<char> out(ptrs.size() * 65);
            CHECK(vt.value()->hash_many(vt.value()->self, ptrs.data(), lens.data(), ptrs.size(), out.data()) == YAMS_ERR_UNSUPPORTED);
            ptrs.clear(); lens.clear();
            for (int i = 0; i < 200; ++i) { ptrs.push_back(reinterpret_cast<const uint8_t*>(big.data()) + (i % 3) * 1000); lens.push_back(1 << 20); } // (inside the 3 MiB)
            out.resize(ptrs.size() * 65);
            CHECK(vt.value()->hash_many(vt.value()->self, ptrs.data(), lens.data(), ptrs.size(), out.data()) == YAMS_OK);
            CHECK(std::string(out.data(), 64) == crypto::SHA256Hasher::hash(std::span<const std::byte>(big.data(), size_t(1) << 20)));
        }
        // (the factory form: one host hasher for the streaming chain, a second one for refused one-shot chains)
        auto made = crypto::createAccelSHA256Hasher(plugin, [] { return std::unique_ptr<crypto::IContentHasher>(std::make_unique<crypto::SHA256Hasher>()); });
        CHECK(made.has_value() && made.value()->hasHostHasher() && made.value()->hasOneShotHostHasher());
        auto& acc = *made.value();
        {   // a one-shot hash() of a refused (> 1 MiB) chain between two update() calls must not disturb the
            // chain in progress — SHA256Hasher::hash(span) is static and stateless in the reference (sha256_hasher.cpp:167-195)
            acc.init(); acc.update({big.data(), 1000});
            CHECK(acc.hash(big) == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));
            acc.update({big.data() + 1000, big.size() - 1000});
            CHECK(acc.finalize() == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));
            // ... also with ONE host hasher only (refused chains then take a private device stream) and with none
            auto single = crypto::createAccelSHA256Hasher(plugin, std::make_unique<crypto::SHA256Hasher>());
            CHECK(single.has_value() && single.value()->hasHostHasher() && !single.value()->hasOneShotHostHasher());
            single.value()->init(); single.value()->update({big.data(), 1000});
            CHECK(single.value()->hash(big) == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));
            single.value()->update({big.data() + 1000, big.size() - 1000});
            CHECK(single.value()->finalize() == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));
        }
        CHECK(acc.hash(big) == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));       // host chain
        CHECK(acc.hash(small) == crypto::SHA256Hasher::hash(std::span<const std::byte>(small)));   // device
        acc.init(); acc.update({big.data(), 1000}); acc.update({big.data() + 1000, big.size() - 1000});
        CHECK(acc.finalize() == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));
        // hashMany: a skewed batch (refused as a whole) falls apart into single chains; a regular batch is one device call
        std::vector<std::span<const std::byte>> skew{{big.data(), big.size()}, {small.data(), small.size()}, {small.data(), 0}};
        auto hs = acc.hashMany(skew);
        CHECK(hs.size() == 3 && hs[0] == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)) &&
              hs[1] == crypto::SHA256Hasher::hash(std::span<const std::byte>(small)) &&
              hs[2] == crypto::SHA256Hasher::hash(std::span<const std::byte>(small.data(), 0)));
        // hashFiles: bounded groups, read until EOF
        std::vector<std::filesystem::path> paths;
        std::vector<std::string> want;
        for (int i = 0; i < 12; ++i) {
            const auto pth = std::filesystem::temp_directory_path() / ("yams_accel_files_" + std::to_string(i) + ".bin");
            const size_t n = i == 0 ? 0 : (size_t(1) << (8 + i)) + i * 37;
            { std::ofstream f(pth, std::ios::binary); f.write(reinterpret_cast<const char*>(big.data()) + i * 101, n); }
            paths.push_back(pth);
            want.push_back(crypto::SHA256Hasher::hash(std::span<const std::byte>(big.data() + i * 101, n)));
        }
        CHECK(acc.hashFiles(paths, /*maxBytes=*/300000) == want);
        // ... and without a host hasher the same calls go through the plugin's streaming door
        auto bare = crypto::createAccelSHA256Hasher(plugin);
        CHECK(bare.has_value() && !bare.value()->hasHostHasher());
        CHECK(bare.value()->hash(big) == crypto::SHA256Hasher::hash(std::span<const std::byte>(big)));
        CHECK(bare.value()->hashFiles(paths) == want);
        for (auto& pth : paths) std::filesystem::remove(pth);
    }

    std::printf("[section] chunker\n");
    // ---- IChunker: both chunkers next to the reference's ---------------------------------------------------
    for (int kind = 0; kind < 2; ++kind) {
        chunking::ChunkingConfig cfg;
        cfg.minChunkSize = 2048; cfg.maxChunkSize = 65536;
        auto made = chunking::createAccelChunker(plugin, kind ? chunking::AccelChunkerKind::Streaming : chunking::AccelChunkerKind::Rabin, cfg);
        CHECK(made.has_value());
        std::unique_ptr<chunking::IChunker> acc = std::move(made.value());
        std::unique_ptr<chunking::IChunker> ref;
        if (kind) ref = std::make_unique<chunking::StreamingChunker>(cfg); else ref = std::make_unique<chunking::RabinChunker>(cfg);
        std::mt19937 rng(17 + kind);
        std::vector<std::byte> data((3 << 20) + 12345);
        for (auto& b : data) b = static_cast<std::byte>(rng());
        for (size_t n : {size_t(0), size_t(47), size_t(2048), size_t(2049), size_t(100000), data.size()}) {
            auto a = acc->chunkData({data.data(), n});
            auto r = ref->chunkData({data.data(), n});
            CHECK(a.size() == r.size());
            for (size_t i = 0; i < a.size() && i < r.size(); ++i) CHECK(a[i] == r[i]);   // Chunk::operator== : data, hash, offset, size
            auto lazy = acc->chunkDataLazy({data.data(), n});
            CHECK(lazy.size() == r.size());
            for (size_t i = 0; i < lazy.size() && i < r.size(); ++i)
                CHECK(lazy[i].data.empty() && lazy[i].hash == r[i].hash && lazy[i].offset == r[i].offset && lazy[i].size == r[i].size);
        }
        CHECK(acc->getConfig().minChunkSize == 2048 && acc->getConfig().chunkMask == 0x1FFF);
        // chunk_many: a batch of buffers in ONE device call — chunk tables, chunk hashes and whole-buffer hashes
        auto batcher = chunking::createAccelBatchChunker(plugin, kind ? chunking::AccelChunkerKind::Streaming : chunking::AccelChunkerKind::Rabin, cfg);
        CHECK(batcher.has_value());
        std::vector<std::span<const std::byte>> bufs;
        for (size_t n : {size_t(100000), size_t(0), size_t(47), size_t(2048), size_t(2049), data.size() - 1000, size_t(65537), size_t(1)})
            bufs.emplace_back(data.data() + (n % 977), n);   // (unaligned starts, overlapping ranges of one array)
        for (bool lazy : {true, false}) {
            auto res = batcher.value()->chunkMany(bufs, /*withBufferHashes=*/true, lazy);
            CHECK(res.chunks.size() == bufs.size() && res.bufferHashes.size() == bufs.size());
            for (size_t b = 0; b < bufs.size(); ++b) {
                auto r = ref->chunkData(bufs[b]);
                CHECK(res.chunks[b].size() == r.size());
                for (size_t i = 0; i < r.size() && i < res.chunks[b].size(); ++i) {
                    const auto& c = res.chunks[b][i];
                    CHECK(c.hash == r[i].hash && c.offset == r[i].offset && c.size == r[i].size);
                    CHECK(lazy ? c.data.empty() : c.data == r[i].data);
                }
                CHECK(res.bufferHashes[b] == crypto::SHA256Hasher::hash(bufs[b]));
            }
        }
        CHECK(batcher.value()->chunkMany({}, true).chunks.empty());
        if (kind) {
            // StreamingChunker::processStream / processFileStream (streaming_chunker.h:54-121,
            // tests/unit/chunking/chunking_test.cpp:403,496,540) — the bounded-memory callback form, windowed on the
            // device.  Same ChunkRefs (hash, offset, size) and the same bytes as the reference's, for streams that
            // hand out 64 KiB, odd-sized and 7-byte pieces, windows from 64 KiB (dozens of windows, the open chunk
            // carried across many of them) to larger than the stream, and the default configuration (1 MiB chunks).
            struct SegBuf : std::streambuf { // at most `seg` bytes per refill (the reference test's SegmentingStringBuf idea)
                const char* p; size_t n, at = 0, seg; std::vector<char> cur;
                SegBuf(const std::byte* d, size_t len, size_t s) : p(reinterpret_cast<const char*>(d)), n(len), seg(s) {}
                int_type underflow() override {
                    if (at >= n) return traits_type::eof();
                    const size_t k = std::min(seg, n - at);
                    cur.assign(p + at, p + at + k); at += k;
                    setg(cur.data(), cur.data(), cur.data() + k);
                    return traits_type::to_int_type(cur[0]);
                }
            };
            for (int which = 0; which < 2; ++which) {
                chunking::ChunkingConfig scfg = cfg;
                if (which) scfg = chunking::ChunkingConfig{};                     // min 16 KiB / max 1 MiB
                chunking::StreamingChunker refStream(scfg);
                auto accStream = chunking::createAccelBatchChunker(plugin, chunking::AccelChunkerKind::Streaming, scfg);
                CHECK(accStream.has_value());
                for (size_t len : {size_t(0), size_t(1), size_t(100000), data.size()}) {
                    std::vector<chunking::ChunkRef> want; size_t wantBytes = 0;
                    {
                        SegBuf sb(data.data(), len, len ? len : 1); std::istream is(&sb);
                        CHECK(refStream.processStream(is, len, [&](const chunking::ChunkRef& r, std::span<const std::byte> b) { want.push_back(r); wantBytes += b.size(); }).has_value());
                    }
                    CHECK(wantBytes == len);
                    for (size_t seg : {size_t(64) << 10, size_t(7), size_t(100003), len ? len : size_t(1)})
                        for (size_t window : {size_t(64) << 10, size_t(1) << 20, size_t(64) << 20}) {
                            if (seg == 7 && (len > 200000 || window != (size_t(64) << 10))) continue; // (7-byte reads: short streams only)
                            SegBuf sb(data.data(), len, seg); std::istream is(&sb);
                            std::vector<chunking::ChunkRef> got; bool bytesOk = true;
                            auto rc = accStream.value()->processStream(is, len, [&](const chunking::ChunkRef& r, std::span<const std::byte> b) {
                                got.push_back(r);
                                bytesOk = bytesOk && b.size() == r.size && r.offset + r.size <= len && std::equal(b.begin(), b.end(), data.begin() + r.offset);
                            }, window);
                            CHECK(rc.has_value() && bytesOk && got.size() == want.size());
                            for (size_t i = 0; i < got.size() && i < want.size(); ++i) CHECK(got[i] == want[i]);
                        }
                }
                // processFileStream + progress; a missing file is FileNotFound (streaming_chunker.h:54-63)
                const auto pth = std::filesystem::temp_directory_path() / "yams_accel_stream_test.bin";
                { std::ofstream f(pth, std::ios::binary); f.write(reinterpret_cast<const char*>(data.data()), static_cast<std::streamsize>(data.size())); }
                std::vector<chunking::ChunkRef> want, got;
                CHECK(refStream.processFileStream(pth, [&](const chunking::ChunkRef& r, std::span<const std::byte>) { want.push_back(r); }).has_value());
                uint64_t lastDone = 0, lastTotal = 0;
                accStream.value()->setProgressCallback([&](uint64_t d, uint64_t t) { lastDone = d; lastTotal = t; });
                CHECK(accStream.value()->processFileStream(pth, [&](const chunking::ChunkRef& r, std::span<const std::byte>) { got.push_back(r); }, size_t(256) << 10).has_value());
                CHECK(got == want && lastDone == data.size() && lastTotal == data.size());
                CHECK(accStream.value()->processFileStream("/tmp/yams_accel_no_such_stream", [](const chunking::ChunkRef&, std::span<const std::byte>) {}).error().code == ErrorCode::FileNotFound);
                std::filesystem::remove(pth);
            }
            auto rabinOnly = chunking::createAccelBatchChunker(plugin, chunking::AccelChunkerKind::Rabin, cfg);
            std::istringstream empty;
            CHECK(rabinOnly.value()->processStream(empty, 0, [](const chunking::ChunkRef&, std::span<const std::byte>) {}).error().code == ErrorCode::InvalidOperation);
        }
        {   // a batch must not wait for ONE long chain.  With a host hash the 3 MiB buffer (above
            // yams_ingest_defer_threshold_host of this 3.3 MiB batch = 1 MiB) is left to the host — hashed on a host
            // thread while the device call runs — and every buffer hash still equals the reference's.
            std::atomic<int> hostCalls{0};
            batcher.value()->setHostHash([&](std::span<const std::byte> d) { ++hostCalls; return crypto::SHA256Hasher::hash(d); });
            auto res = batcher.value()->chunkMany(bufs, /*withBufferHashes=*/true, true);
            CHECK(hostCalls.load() == 1 && res.bufferHashes.size() == bufs.size());
            for (size_t b = 0; b < bufs.size(); ++b) {
                CHECK(res.bufferHashes[b] == crypto::SHA256Hasher::hash(bufs[b]));
                auto r = ref->chunkData(bufs[b]);
                CHECK(res.chunks[b].size() == r.size() && (r.empty() || res.chunks[b].back().hash == r.back().hash));
            }
            // the raw door: the deferred entry comes back empty, the others filled
            auto cvt = plugin->getInterface<yams_chunker_v1>(YAMS_IFACE_CHUNKER_V1, YAMS_IFACE_CHUNKER_V1_VERSION).value();
            std::vector<const uint8_t*> ptrs; std::vector<size_t> lens;
            for (auto& b : bufs) { ptrs.push_back(reinterpret_cast<const uint8_t*>(b.data())); lens.push_back(b.size()); }
            yams_cdc_config_t ccfg{}; cvt->get_default_config(cvt->self, kind ? YAMS_CDC_STREAMING : YAMS_CDC_RABIN, &ccfg);
            yams_chunk_batch_t* batch = nullptr;
            CHECK(cvt->chunk_many(cvt->self, ptrs.data(), lens.data(), ptrs.size(), &ccfg,
                                  YAMS_CHUNK_MANY_BUFFER_HASHES | YAMS_CHUNK_MANY_DEFER_LONG_BUFFER_HASHES, &batch) == YAMS_OK);
            if (batch) {
                for (size_t b = 0; b < bufs.size(); ++b) CHECK((batch->buffer_hash_hex[65 * b] == 0) == (lens[b] > (size_t(1) << 20)));
                cvt->free_chunk_batch(cvt->self, batch);
            }
            // min_size == 0 (legal when max_size > 0) and min_size > max_size through chunk_many
            for (auto mm : {std::pair<uint64_t, uint64_t>{0, 4096}, std::pair<uint64_t, uint64_t>{8192, 1024}}) {
                yams_cdc_config_t c2 = ccfg; c2.min_size = mm.first; c2.max_size = mm.second;
                const uint8_t* p1[2] = {reinterpret_cast<const uint8_t*>(data.data()), reinterpret_cast<const uint8_t*>(data.data()) + 5000};
                const size_t l1[2] = {300000, 70001};
                yams_chunk_batch_t* b2 = nullptr;
                CHECK(cvt->chunk_many(cvt->self, p1, l1, 2, &c2, 0, &b2) == YAMS_OK);
                if (b2) {
                    for (int j = 0; j < 2; ++j) {
                        yams_chunk_ref_t* one = nullptr; size_t n1 = 0;
                        CHECK(cvt->chunk_data(cvt->self, p1[j], l1[j], &c2, &one, &n1) == YAMS_OK);
                        CHECK(n1 == b2->first_chunk[j + 1] - b2->first_chunk[j]);
                        for (size_t i = 0; i < n1 && n1 == b2->first_chunk[j + 1] - b2->first_chunk[j]; ++i) {
                            const auto& m = b2->chunks[b2->first_chunk[j] + i];
                            CHECK(m.offset == one[i].offset && m.size == one[i].size && std::string(m.hash_hex, 64) == std::string(one[i].hash_hex, 64));
                        }
                        cvt->free_chunks(cvt->self, one, n1);
                    }
                    cvt->free_chunk_batch(cvt->self, b2);
                }
            }
        }
    }

    std::printf("[section] vector store\n");
    // ---- IVectorStore + capability seams ---------------------------------------------------------------------
    {
        std::unique_ptr<vector::IVectorStore> backend = vector::createAccelExactScanBackend(plugin);
        CHECK(backend->initialize(":memory:").has_value());
        CHECK(backend->createTables(256).has_value() && backend->tablesExist() && backend->isInitialized());
        auto* diagnosticStore = dynamic_cast<vector::IDiagnosticVectorStore*>(backend.get());
        auto* exactStore = dynamic_cast<vector::IExactCandidateVectorStore*>(backend.get());
        auto* allRowsStore = dynamic_cast<vector::IAllExactCandidateVectorStore*>(backend.get());
        auto* documentStore = dynamic_cast<vector::IDocumentCandidateVectorStore*>(backend.get());
        CHECK(diagnosticStore && exactStore && allRowsStore && documentStore);

        const size_t dim = 256, n = 6000, perDoc = 4;   // doc_<i/4>, chunk_%09zu: string order == row order
        std::mt19937 rng(42);
        std::vector<vector::VectorRecord> recs;
        for (size_t i = 0; i < n; ++i) {
            char id[32], doc[32];
            std::snprintf(id, sizeof id, "chunk_%09zu", i);
            std::snprintf(doc, sizeof doc, "doc_%06zu", i / perDoc);
            vector::VectorRecord r(id, doc, unit(rng, dim), "content");
            r.metadata["lang"] = (i % 3 == 0) ? "en" : "de";
            r.level = (i % perDoc == 0) ? vector::EmbeddingLevel::DOCUMENT : vector::EmbeddingLevel::CHUNK;
            recs.push_back(std::move(r));
        }
        CHECK(backend->insertVectorsBatch(recs).has_value());
        CHECK(backend->getVectorCount().value() == n);
        const auto q = unit(rng, dim);

        // plain top-k against the reference arithmetic, order (similarity desc, chunk_id asc)
        std::vector<std::pair<float, size_t>> exact;
        for (size_t i = 0; i < n; ++i) exact.emplace_back(refCosine(q, recs[i].embedding), i);
        std::sort(exact.begin(), exact.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
        auto top = backend->searchSimilar(q, 25, -1.0f);
        CHECK(top.has_value() && top.value().size() == 25);
        for (size_t i = 0; i < 25 && i < top.value().size(); ++i) {
            CHECK(top.value()[i].chunk_id == recs[exact[i].second].chunk_id);
            CHECK(top.value()[i].relevance_score == exact[i].first);   // bit-identical fp32
        }
        // diagnostics seam: the collect flag survives the reset (:4650-4661)
        vector::VectorSearchDiagnostics diag;
        diag.collectVisitedDocumentHashes = true; diag.rowsVisited = 999;
        std::unordered_set<std::string> cands = {"doc_000003", "doc_000700", "doc_001499"};
        auto viaDiag = diagnosticStore->searchSimilarWithDiagnostics(q, 5, -1.0f, std::nullopt, cands, {}, diag);
        CHECK(viaDiag.has_value() && viaDiag.value().size() == 5);
        CHECK(diag.usedExactScan && diag.rowsVisitedObserved && diag.rowsVisited == 12 && diag.exactDistanceEvaluations == 12);
        CHECK(diag.collectVisitedDocumentHashes && diag.visitedDocumentHashes == cands);
        // exact candidates (:1578-1593) and the InvalidArgument rule
        auto noCands = exactStore->searchExactCandidatesWithDiagnostics(q, 5, -1.0f, {}, diag);
        CHECK(!noCands.has_value() && noCands.error().code == ErrorCode::InvalidArgument);
        auto ex = exactStore->searchExactCandidatesWithDiagnostics(q, 3, -1.0f, cands, diag);
        CHECK(ex.has_value() && ex.value().size() == 3 && diag.returnedRows == 3);
        for (const auto& r : ex.value()) CHECK(cands.count(r.document_hash) == 1);
        // all rows of MANY candidate documents: more than one device call returns (no cap, :1595-1610)
        std::unordered_set<std::string> many;
        for (size_t dd = 0; dd < 400; ++dd) { char doc[32]; std::snprintf(doc, sizeof doc, "doc_%06zu", dd * 3); many.insert(doc); }
        auto all = allRowsStore->searchAllExactCandidateRowsWithDiagnostics(q, -1.0f, many, diag);
        CHECK(all.has_value() && all.value().size() == 400 * perDoc && diag.rowsVisited == 400 * perDoc);
        for (size_t i = 1; i < all.value().size(); ++i) {
            const auto &a = all.value()[i - 1], &b = all.value()[i];
            CHECK(a.relevance_score > b.relevance_score || (a.relevance_score == b.relevance_score && a.chunk_id < b.chunk_id));
        }
        // one best row per document (:86-125), k documents
        auto docs = documentStore->searchDocumentCandidatesWithDiagnostics(q, 7, -1.0f, many, diag);
        CHECK(docs.has_value() && docs.value().size() == 7);
        {
            std::unordered_map<std::string, float> best;
            for (const auto& r : all.value()) { auto it = best.find(r.document_hash); if (it == best.end() || r.relevance_score > it->second) best[r.document_hash] = r.relevance_score; }
            std::unordered_set<std::string> seen;
            float prev = 2.f;
            for (const auto& r : docs.value()) {
                CHECK(seen.insert(r.document_hash).second && r.relevance_score == best[r.document_hash] && r.relevance_score <= prev);
                prev = r.relevance_score;
            }
        }
        // metadata filters take the record path (:4333-4409); threshold; batch; invalid query; k = 0
        auto en = backend->searchSimilar(q, 10, -1.0f, std::nullopt, {}, {{"lang", "en"}});
        CHECK(en.has_value() && en.value().size() == 10);
        for (const auto& r : en.value()) CHECK(r.metadata.at("lang") == "en");
        CHECK(backend->searchSimilar(q, 0, -1.0f).value().empty());
        auto zero = backend->searchSimilar(std::vector<float>(dim, 0.f), 5, -1.0f);
        CHECK(!zero.has_value() && zero.error().code == ErrorCode::InvalidArgument);
        auto batch = backend->searchSimilarBatch({q, recs[17].embedding}, 4, -1.0f, 0);
        CHECK(batch.has_value() && batch.value().size() == 2 && batch.value()[1].front().chunk_id == "chunk_000000017");
        // CRUD through the base class: replace, delete, document delete; retrieval methods
        vector::VectorRecord repl = recs[17]; repl.embedding = q;
        CHECK(backend->updateVector("chunk_000000017", repl).has_value());
        CHECK(backend->searchSimilar(q, 1, -1.0f).value().front().chunk_id == "chunk_000000017");
        CHECK(backend->deleteVector("chunk_000000017").has_value());
        CHECK(backend->searchSimilar(q, 1, -1.0f).value().front().chunk_id == recs[exact[0].second].chunk_id);
        CHECK(backend->deleteVectorsByDocument("doc_000001").has_value());
        CHECK(!backend->hasEmbedding("doc_000001").value() && backend->hasEmbedding("doc_000002").value());
        CHECK(backend->getVectorCount().value() == n - 1 - perDoc);
        CHECK(backend->getVector("chunk_000000020").value().has_value() && !backend->getVector("chunk_000000017").value().has_value());
        CHECK(backend->getVectorsByDocument("doc_000002").value().size() == perDoc);
        CHECK(backend->getDocumentLevelVectorsAll().value().size() == n / perDoc - 1);
        CHECK(backend->getEmbeddedDocumentHashes().value().size() == n / perDoc - 1);
        CHECK(backend->getStats().value().total_vectors == n - 1 - perDoc);
        CHECK(backend->beginTransaction().has_value() && backend->commitTransaction().has_value());
        backend->close();
        CHECK(!backend->isInitialized());
        CHECK(backend->searchSimilar(q, 1, -1.0f).error().code == ErrorCode::NotInitialized);
    }

    std::printf("[section] transactions\n");
    // ---- transactions: the device mirror holds COMMITTED state (phantom chunk_ids after a rollback) -----
    for (int durableMode = 0; durableMode < 2; ++durableMode) {
        auto durable = std::make_shared<FakeDurableStore>();
        std::unique_ptr<vector::IVectorStore> backend =
            vector::createAccelExactScanBackend(plugin, durableMode ? std::static_pointer_cast<vector::IVectorStore>(durable) : nullptr);
        const size_t dim = 64;
        std::mt19937 rng(7);
        auto rec = [&](const std::string& id, const std::string& doc) { return vector::VectorRecord(id, doc, unit(rng, dim), "c"); };
        if (durableMode) { // rows that exist before initialize(): the mirror is warmed from the durable store
            (void)durable->insertVector(rec("chunk_old_1", "doc_old"));
            (void)durable->insertVector(rec("chunk_old_2", "doc_old"));
        }
        CHECK(backend->initialize(":memory:").has_value() && backend->createTables(dim).has_value());
        if (!durableMode) { CHECK(backend->insertVector(rec("chunk_old_1", "doc_old")).has_value()); CHECK(backend->insertVector(rec("chunk_old_2", "doc_old")).has_value()); }
        const auto target = rec("chunk_new", "doc_new");
        const auto old1 = backend->getVector("chunk_old_1").value().value();
        auto top = [&](const std::vector<float>& qq) { auto r = backend->searchSimilar(qq, 1, -1.0f); return r && !r.value().empty() ? r.value().front().chunk_id : std::string("<none>"); };
        CHECK(top(old1.embedding) == "chunk_old_1");
        // rollback: an inserted row never becomes searchable, a deleted row is back, a replaced row has its old vector
        CHECK(backend->beginTransaction().has_value());
        CHECK(!backend->beginTransaction().has_value());                          // one transaction at a time
        CHECK(backend->insertVector(target).has_value());
        CHECK(backend->deleteVector("chunk_old_2").has_value());
        vector::VectorRecord repl = old1; repl.embedding = target.embedding;
        CHECK(backend->updateVector("chunk_old_1", repl).has_value());
        CHECK(backend->rollbackTransaction().has_value());
        CHECK(top(target.embedding) != "chunk_new");
        {
            const auto found = backend->searchSimilar(target.embedding, 10, -1.0f);
            CHECK(found.has_value());
            for (const auto& r : found.value()) CHECK(backend->getVector(r.chunk_id).value().has_value());   // no phantom ids
        }
        CHECK(backend->getVector("chunk_old_2").value().has_value() && !backend->getVector("chunk_new").value().has_value());
        CHECK(top(old1.embedding) == "chunk_old_1" && backend->getVectorCount().value() == 2);
        // commit: everything of the transaction becomes searchable, in order (insert, then replace, then a document delete)
        CHECK(backend->beginTransaction().has_value());
        CHECK(backend->insertVector(target).has_value());
        CHECK(backend->insertVector(rec("chunk_gone", "doc_gone")).has_value());
        CHECK(backend->deleteVectorsByDocument("doc_gone").has_value());
        if (durableMode) CHECK(top(target.embedding) != "chunk_new");             // durable mode: not before the commit
        CHECK(backend->commitTransaction().has_value());
        CHECK(top(target.embedding) == "chunk_new" && backend->getVectorCount().value() == 3);
        CHECK(!backend->hasEmbedding("doc_gone").value());
        if (durableMode) {
            // a commit the durable store refuses leaves the transaction open; the rollback that follows drops the journal
            CHECK(backend->beginTransaction().has_value());
            CHECK(backend->insertVector(rec("chunk_never", "doc_never")).has_value());
            durable->failNextCommit = 1;
            CHECK(!backend->commitTransaction().has_value());
            CHECK(backend->rollbackTransaction().has_value());
            CHECK(!backend->getVector("chunk_never").value().has_value());
            const auto found = backend->searchSimilar(target.embedding, 10, -1.0f);
            CHECK(found.has_value());
            for (const auto& r : found.value()) CHECK(r.chunk_id != "chunk_never");
        }
        if (durableMode) {
            // a stale mirror re-warmed INSIDE an open transaction reads the transaction's uncommitted rows;
            // after the rollback those must not stay searchable (the mirror is marked stale again and rebuilt)
            auto* concrete = dynamic_cast<vector::AccelExactScanBackend*>(backend.get());
            CHECK(concrete != nullptr);
            CHECK(backend->beginTransaction().has_value());
            const auto ghost = rec("chunk_ghost", "doc_ghost");
            CHECK(backend->insertVector(ghost).has_value());
            if (concrete) concrete->invalidateMirror();
            (void)top(ghost.embedding);                                            // re-warms from the durable rows, ghost included
            CHECK(backend->rollbackTransaction().has_value());
            CHECK(!backend->getVector("chunk_ghost").value().has_value());
            const auto found = backend->searchSimilar(ghost.embedding, 10, -1.0f);
            CHECK(found.has_value());
            for (const auto& r : found.value()) CHECK(r.chunk_id != "chunk_ghost" && backend->getVector(r.chunk_id).value().has_value());
            // ... and when such a transaction COMMITS the journal is applied over the re-warmed rows without duplicates
            CHECK(backend->beginTransaction().has_value());
            CHECK(backend->insertVector(ghost).has_value());
            if (concrete) concrete->invalidateMirror();
            (void)top(ghost.embedding);
            CHECK(backend->commitTransaction().has_value());
            CHECK(top(ghost.embedding) == "chunk_ghost" && backend->getVectorCount().value() == 4);
            const auto again = backend->searchSimilar(ghost.embedding, 10, -1.0f);
            CHECK(again.has_value() && again.value().size() == 4);
            CHECK(backend->deleteVector("chunk_ghost").has_value());
        }
        // searches share the lock: four threads at once, same answers
        {
            std::vector<std::thread> th; std::atomic<int> ok{0};
            for (int t = 0; t < 4; ++t) th.emplace_back([&] { for (int i = 0; i < 5; ++i) if (top(target.embedding) == "chunk_new") ++ok; });
            for (auto& t : th) t.join();
            CHECK(ok == 20);
        }
        backend->close();
    }
    std::printf("[section] vec0 L2 through the host's own classes, calibrated\n");
    // ---- the Vec0L2 engine behind IVectorStore: the HOST's distance function settles the arithmetic (l2_calibration.hpp) --
    {
        namespace l2n = vector::accel_l2;
        vector::AccelExactScanBackend backend(plugin, nullptr, vector::VectorSearchEngine::Vec0L2);
        CHECK(backend.initialize(":memory:").has_value() && backend.createTables(256).has_value());
        const size_t dim = 256, n = 3000;
        std::mt19937 rng(7);
        std::vector<vector::VectorRecord> recs;
        for (size_t i = 0; i < n; ++i) {
            char id[32];
            std::snprintf(id, sizeof id, "l2_%06zu", i);
            auto e = unit(rng, dim);
            for (auto& v : e) v *= 1.7f;
            recs.emplace_back(id, "doc", std::move(e), "content");
        }
        CHECK(backend.insertVectorsBatch(recs).has_value());
        const auto q = unit(rng, dim);
        // a host built like the reference builds its dependency on x86 (-mavx -mfma: eight lanes, fused): recognised, and
        // from then on the backend's answer is that host's own brute force, row for row
        const auto def = l2n::L2Accumulate::F32x8Fma;
        auto host = [def](const float* a, const float* b, size_t d, float* out) { *out = l2n::distance(def, a, b, d); return true; };
        auto cal = backend.calibrateL2(host);
        CHECK(cal.has_value() && cal.value().matched && cal.value().accumulate == def && cal.value().dim == dim);
        auto got = backend.searchSimilar(q, 40, -1.0f);
        CHECK(got.has_value() && got.value().size() == 40);
        std::vector<std::pair<float, std::string>> brute;
        for (const auto& r : recs) brute.emplace_back(l2n::distance(def, r.embedding.data(), q.data(), dim), r.chunk_id);
        std::sort(brute.begin(), brute.end());
        if (got) for (size_t i = 0; i < 40; ++i) CHECK(got.value()[i].chunk_id == brute[i].second);
        // a host that matches nothing is refused with the reference's own error vocabulary
        auto odd = [](const float* a, const float* b, size_t d, float* out) {
            float p[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < d; ++i) { const float x = a[i] - b[i]; p[i % 4] += x * x; }
            *out = std::sqrt((p[0] + p[1]) + (p[2] + p[3]));
            return true;
        };
        auto none = backend.calibrateL2(odd);
        CHECK(none.has_value() && !none.value().matched);
        auto refused = backend.searchSimilar(q, 5, -1.0f);
        CHECK(!refused.has_value() && refused.error().code == ErrorCode::NotSupported);
        backend.close();
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
