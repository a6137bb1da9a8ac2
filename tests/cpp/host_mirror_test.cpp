// host_mirror_test.cpp — the C++ host-side mirror (include/yams_accel/*.hpp) exercised the way the
// reference's Catch2 tests exercise the seams it replaces:
//   tests/unit/vector/vector_smoke_catch2_test.cpp:188-353   (exact-scan contract, ties, invalid queries)
//   tests/unit/crypto/crypto_test.cpp:92-99,134-228          (SHA-256 known answers, split updates)
//   tests/unit/chunking/chunking_test.cpp:146-228            (chunk invariants, hash == SHA-256 of slice)
// Usage: host_mirror_test <path/to/libyams_mi355x_accel.so> [--expect-no-gpu | --config '<init json>']
// (e.g. --config '{"devices":[0,0],"stripe_rows":64}': every corpus dealt to two shards — two contexts
// on one device — in stripes of 64 rows; the suite must pass unchanged, bit for bit)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <random>
#include <string>

#include "yams_accel/chunker.hpp"
#include "yams_accel/hasher.hpp"
#include "yams_accel/vector_index.hpp"
#include "yams_accel/integrity.hpp"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

using namespace yams;

static std::span<const std::byte> bytes(const std::string& s) {
    return {reinterpret_cast<const std::byte*>(s.data()), s.size()};
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: %s <plugin.so> [--expect-no-gpu | --config <json>]\n", argv[0]); return 2; }
    const bool expectNoGpu = argc > 2 && std::strcmp(argv[2], "--expect-no-gpu") == 0;
    const std::string config = (argc > 3 && std::strcmp(argv[2], "--config") == 0) ? argv[3] : "{\"device\":0}";
    auto loaded = accel::Plugin::load(argv[1], config);
    if (expectNoGpu) {
        CHECK(!loaded.has_value());
        if (!loaded.has_value()) CHECK(loaded.error().code == ErrorCode::NotInitialized);
        std::printf("%s\n", failures ? "FAILED" : "OK (refused without a GPU)");
        return failures ? 1 : 0;
    }
    if (!loaded) { std::printf("load failed: %s\n", loaded.error().message.c_str()); return 1; }
    auto plugin = loaded.value();
    CHECK(plugin->manifestJson().find("vector_scan_v1") != std::string::npos);

    // ---- crypto ----------------------------------------------------------------------------
    auto hasherR = crypto::createAccelSHA256Hasher(plugin);
    CHECK(hasherR.has_value());
    auto& hasher = *hasherR.value();
    hasher.init();
    CHECK(hasher.finalize() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855");
    hasher.init(); hasher.update(bytes("abc"));
    CHECK(hasher.finalize() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad");
    CHECK(hasher.hash(bytes("Hello World")) == "a591a6d40bf420404a011733cfb7b190d62c65bf0bcda32b57b277d9ad9f146e");
    {
        std::mt19937 rng(7); std::string data(1000, '\0');
        for (auto& c : data) c = static_cast<char>(rng());
        hasher.init();
        hasher.update(bytes(data).subspan(0, 100)); hasher.update(bytes(data).subspan(100, 400));
        hasher.update(bytes(data).subspan(500, 500));
        const auto h1 = hasher.finalize();
        CHECK(h1 == hasher.hash(bytes(data)));
        auto many = hasher.hashMany({bytes(data), bytes("abc")});
        CHECK(many.size() == 2 && many[0] == h1 && many[1] == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad");
    }

    // ---- chunking --------------------------------------------------------------------------
    for (auto kind : {chunking::AccelChunkerKind::Rabin, chunking::AccelChunkerKind::Streaming}) {
        chunking::ChunkingConfig cfg; cfg.minChunkSize = 4096; cfg.maxChunkSize = 32768;
        auto chunkerR = chunking::createAccelChunker(plugin, kind, cfg);
        CHECK(chunkerR.has_value());
        auto& chunker = *chunkerR.value();
        CHECK(chunker.chunkData({}).empty()); // "Empty input produces no chunks"
        std::mt19937 rng(11); std::string data((1 << 20) + 777, '\0');
        for (auto& c : data) c = static_cast<char>(rng());
        auto chunks = chunker.chunkData(bytes(data));
        CHECK(!chunks.empty());
        size_t pos = 0;
        const size_t lo = kind == chunking::AccelChunkerKind::Rabin ? cfg.minChunkSize + 1 : cfg.minChunkSize;
        for (size_t i = 0; i < chunks.size(); ++i) {
            CHECK(chunks[i].offset == pos);                       // contiguity / coverage
            CHECK(chunks[i].size <= cfg.maxChunkSize);
            if (i + 1 < chunks.size()) CHECK(chunks[i].size >= lo);
            CHECK(chunks[i].data.size() == chunks[i].size);
            pos += chunks[i].size;
        }
        CHECK(pos == data.size());
        for (size_t i = 0; i < chunks.size(); i += 5)          // chunk hash == SHA-256 of the slice
            CHECK(chunks[i].hash == hasher.hash(bytes(data).subspan(chunks[i].offset, chunks[i].size)));
        {   // progress callback + async file chunking (rabin_chunker.cpp:144-147, 182-192)
            uint64_t last = 0, total = 0; size_t calls = 0;
            chunker.setProgressCallback([&](uint64_t p, uint64_t t) { CHECK(p > last); last = p; total = t; ++calls; });
            auto again = chunker.chunkDataLazy(bytes(data));
            CHECK(calls == again.size() && last == data.size() && total == data.size());
            chunker.setProgressCallback(nullptr);
            const std::string path = "/tmp/yams_accel_chunkfile_test.bin";
            { std::ofstream f(path, std::ios::binary); f.write(data.data(), static_cast<std::streamsize>(data.size())); }
            auto viaFile = chunker.chunkFileAsync(path).get();
            CHECK(viaFile.has_value() && viaFile.value().size() == chunks.size() && viaFile.value().back().hash == chunks.back().hash);
            auto missing = chunker.chunkFileAsync("/tmp/yams_accel_no_such_file").get();
            CHECK(!missing.has_value() && missing.error().code == ErrorCode::FileNotFound);
            std::remove(path.c_str());
        }
        auto lazy = chunker.chunkDataLazy(bytes(data));
        CHECK(lazy.size() == chunks.size() && lazy.back().hash == chunks.back().hash && lazy.front().data.empty());
    }

    {   // hashFile / hashFileAsync / progress callback (sha256_hasher.cpp:111-165)
        const std::string path = "/tmp/yams_accel_hashfile_test.bin";
        std::string payload(3 * (1 << 20) + 12345, 'x');
        for (size_t i = 0; i < payload.size(); i += 997) payload[i] = static_cast<char>(i);
        { std::ofstream f(path, std::ios::binary); f.write(payload.data(), static_cast<std::streamsize>(payload.size())); }
        uint64_t lastDone = 0, lastTotal = 0; int calls = 0;
        hasher.setProgressCallback([&](uint64_t d, uint64_t t) { lastDone = d; lastTotal = t; ++calls; });
        CHECK(hasher.hashFile(path) == hasher.hash(bytes(payload)));
        CHECK(calls >= 3 && lastDone == payload.size() && lastTotal == payload.size());
        auto fut = hasher.hashFileAsync(path);
        auto r = fut.get();
        CHECK(r.has_value() && r.value() == hasher.hash(bytes(payload)));
        auto missing = hasher.hashFileAsync("/tmp/yams_accel_no_such_file").get();
        CHECK(!missing.has_value() && missing.error().code == ErrorCode::FileNotFound);
        hasher.setProgressCallback(nullptr);
        std::remove(path.c_str());
    }

    {   // the restated vocabulary behaves as include/yams/core/types.h:147-244 (Error() / error())
        Error none;
        CHECK(none.code == ErrorCode::Success && none.message.empty());
        CHECK(Error(ErrorCode::Timeout).message == "Operation timed out" && Error(std::string("boom")).code == ErrorCode::Unknown);
        CHECK(Error(ErrorCode::NotFound) == ErrorCode::NotFound && ErrorCode::NotFound != Error(ErrorCode::IOError));
        Result<int> v(7), e(ErrorCode::InvalidState), d;
        CHECK(v.has_value() && v.value() == 7 && !e.has_value() && e.error().message == "Invalid state");
        CHECK(!d.has_value() && d.error().code == ErrorCode::InternalError && d.error().message == "Uninitialized Result");
        bool threw = false;
        try { (void)v.error(); } catch (const std::runtime_error& x) { threw = std::string(x.what()) == "Result contains value"; }
        CHECK(threw);
        threw = false;
        try { (void)e.value(); } catch (const std::runtime_error& x) { threw = std::string(x.what()) == "Result contains error"; }
        CHECK(threw);
        Result<void> ok, bad(ErrorCode::WriteError);
        CHECK(ok.has_value() && !bad.has_value() && bad.error().message == "Write error");
        threw = false;
        try { (void)ok.error(); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    }

    // ---- integrity check + dedup lookup (the callers either side of the hash path) -----------------
    {
        auto* hvt = plugin->getInterface<yams_content_hash_v1>(YAMS_IFACE_CONTENT_HASH_V1, 1).value();
        integrity::AccelChunkValidator validator(plugin, hvt);
        std::string a(5000, 'a'), b(70000, 'b'), c;
        const std::string ha = hasher.hash(bytes(a)), hb = hasher.hash(bytes(b)), hc = hasher.hash(bytes(c));
        auto one = validator.validateChunk(bytes(a), ha);
        CHECK(one.isValid && one.errorMessage.empty() && one.chunkSize == 5000 && one.chunkHash == ha);
        auto res = validator.validateChunks({{bytes(a), ha}, {bytes(b), ha}, {bytes(c), hc}, {bytes(b), "short"}});
        CHECK(res.size() == 4 && res[0].isValid && !res[1].isValid && res[2].isValid && !res[3].isValid);
        CHECK(res[1].errorMessage == "Hash mismatch: expected " + ha.substr(0, 8) + ", got " + hb.substr(0, 8));
        // a chunk above 1 MiB is a lone long chain — the device REFUSES it (YAMS_ERR_UNSUPPORTED) and the
        // validator must not report intact data as corrupt for that: it takes such chains one at a time (host hash
        // where one is given, else the device's streaming door) and verifies the rest as a set.
        {
            std::string big(3 * (1 << 20) + 77, 'q');
            for (size_t i = 0; i < big.size(); i += 811) big[i] = static_cast<char>(i * 7);
            const std::string hbig = hasher.hash(bytes(big));
            char hex[65];
            CHECK(hvt->hash(hvt->self, reinterpret_cast<const uint8_t*>(big.data()), big.size(), hex) == YAMS_ERR_UNSUPPORTED); // (the premise)
            auto lone = validator.validateChunk(bytes(big), hbig);
            CHECK(lone.isValid && lone.errorMessage.empty() && lone.chunkSize == big.size());
            auto wrong = validator.validateChunk(bytes(big), ha);
            CHECK(!wrong.isValid && wrong.errorMessage == "Hash mismatch: expected " + ha.substr(0, 8) + ", got " + hbig.substr(0, 8));
            // a small batch dominated by the long chain: refused as a whole, answered chunk by chunk
            auto mixed = validator.validateChunks({{bytes(a), ha}, {bytes(big), hbig}, {bytes(b), ha}, {bytes(big), hb}});
            CHECK(mixed.size() == 4 && mixed[0].isValid && mixed[1].isValid && !mixed[2].isValid && !mixed[3].isValid);
            CHECK(mixed[2].errorMessage == "Hash mismatch: expected " + ha.substr(0, 8) + ", got " + hb.substr(0, 8));
            CHECK(mixed[3].errorMessage == "Hash mismatch: expected " + hb.substr(0, 8) + ", got " + hbig.substr(0, 8));
            // with a host hash supplied, the refused chains go to it (and only those)
            int hostCalls = 0;
            integrity::AccelChunkValidator withHost(plugin, hvt, [&](std::span<const std::byte> d) { ++hostCalls; return hasher.hash(d); });
            auto viaHost = withHost.validateChunks({{bytes(a), ha}, {bytes(big), hbig}});
            CHECK(viaHost[0].isValid && viaHost[1].isValid && hostCalls == 1);
        }
        integrity::AccelDedupIndex known(plugin, hvt);
        auto first = known.insertAndClassify({ha, hb, ha, hc, hb});
        CHECK(first.has_value() && first.value() == std::vector<bool>({true, true, false, true, false}));
        auto again = known.insertAndClassify({hc, ha});
        CHECK(again.has_value() && again.value() == std::vector<bool>({false, false}) && known.size().value() == 3);
        const std::string hd = hasher.hash(bytes(std::string("d")));
        auto look = known.contains({hd, hb});
        CHECK(look.has_value() && look.value() == std::vector<bool>({false, true}) && known.size().value() == 3);
        CHECK(!known.insertAndClassify({"nothex"}).has_value());
    }

    // ---- vector ----------------------------------------------------------------------------
    {
        auto idxR = vector::createAccelVectorIndex(plugin, 4);
        CHECK(idxR.has_value());
        auto& db = *idxR.value();
        CHECK(!db.searchSimilar({1, 0, 0, 0}, 3).has_value());           // NotInitialized
        CHECK(db.initialize().has_value());
        for (int i = 0; i < 6; ++i) {
            vector::VectorRecord r; r.chunk_id = "exact_" + std::to_string(i); r.document_hash = "doc_" + std::to_string(i);
            r.embedding = {1.0f, static_cast<float>(i), 0.0f, 0.0f};
            CHECK(db.insertVector(r).has_value());
        }
        vector::VectorSearchDiagnostics diag;
        auto res = db.searchSimilar({1, 0, 0, 0}, 3, -1.0f, &diag);
        CHECK(res.has_value());
        if (res) {
            CHECK(res.value().size() == 3 && res.value().front().chunk_id == "exact_0");
            CHECK(res.value().front().relevance_score == 1.0f);
        }
        CHECK(diag.usedExactScan && !diag.usedAnn && diag.rowsVisited == 6 && diag.exactDistanceEvaluations == 6);
        auto zero = db.searchSimilar({0, 0, 0, 0}, 1, -1.0f);
        CHECK(!zero.has_value() && zero.error().code == ErrorCode::InvalidArgument);
        auto nan = db.searchSimilar({1, std::numeric_limits<float>::quiet_NaN(), 0, 0}, 1, -1.0f);
        CHECK(!nan.has_value() && nan.error().code == ErrorCode::InvalidArgument);
        auto dimErr = db.searchSimilar({1, 0, 0}, 1, -1.0f);
        CHECK(!dimErr.has_value() && dimErr.error().code == ErrorCode::InvalidArgument);
        auto batch = db.searchSimilarBatch({{1, 0, 0, 0}, {1, 5, 0, 0}}, 2, -1.0f);
        CHECK(batch.has_value() && batch.value().size() == 2 && batch.value()[1].front().chunk_id == "exact_5");
        CHECK(db.deleteVector("exact_0").has_value());
        auto after = db.searchSimilar({1, 0, 0, 0}, 1, -1.0f);
        CHECK(after.has_value() && after.value().front().chunk_id == "exact_1");
        // retrieval / update / delete-by-document on the mirror
        CHECK(db.getVector("exact_3").value().has_value() && !db.getVector("exact_0").value().has_value());
        CHECK(db.hasEmbedding("doc_2").value() && !db.hasEmbedding("doc_0").value());
        vector::VectorRecord up; up.document_hash = "doc_5"; up.embedding = {1.0f, 0.0f, 0.0f, 0.0f};
        CHECK(db.updateVector("exact_5", up).has_value() && !db.updateVector("nope", up).has_value());
        auto upd = db.searchSimilar({1, 0, 0, 0}, 1, -1.0f);
        CHECK(upd.has_value() && upd.value().front().chunk_id == "exact_5" && upd.value().front().relevance_score == 1.0f);
        CHECK(db.deleteVectorsByDocument("doc_5").has_value() && db.getVectorsByDocument("doc_5").value().empty());
        CHECK(db.getVectorCount().value() == 4);
        auto gone5 = db.searchSimilar({1, 0, 0, 0}, 1, -1.0f);
        CHECK(gone5.has_value() && gone5.value().front().chunk_id == "exact_1");
    }
    for (auto order : {std::vector<std::string>{"tie_c", "tie_a", "tie_b"}, std::vector<std::string>{"tie_b", "tie_a", "tie_c"}}) {
        auto idxR = vector::createAccelVectorIndex(plugin, 4);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        for (auto& id : order) { vector::VectorRecord r; r.chunk_id = id; r.embedding = {1, 0, 0, 0}; CHECK(db.insertVector(r).has_value()); }
        auto res = db.searchSimilar({1, 0, 0, 0}, 2, -1.0f);
        CHECK(res.has_value() && res.value().size() == 2 && res.value()[0].chunk_id == "tie_a" && res.value()[1].chunk_id == "tie_b");
    }
    {   // "exact candidate mode scores only allowed documents" (vector_smoke_catch2_test.cpp:355-401)
        auto idxR = vector::createAccelVectorIndex(plugin, 4);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        auto insert = [&](const char* id, const char* doc, std::vector<float> e) {
            vector::VectorRecord r; r.chunk_id = id; r.document_hash = doc; r.embedding = std::move(e);
            return db.insertVector(r);
        };
        CHECK(insert("allowed_best", "allowed", {1.0f, 0.0f, 0.0f, 0.0f}).has_value());
        CHECK(insert("allowed_second", "allowed", {0.8f, 0.6f, 0.0f, 0.0f}).has_value());
        CHECK(insert("blocked", "blocked", {1.0f, 0.0f, 0.0f, 0.0f}).has_value());
        vector::VectorSearchDiagnostics diag;
        auto res = db.searchSimilar({1, 0, 0, 0}, 4, -1.0f, std::nullopt, {"allowed"}, {}, &diag);
        CHECK(res.has_value() && res.value().size() == 2);
        if (res && res.value().size() == 2) {
            CHECK(res.value()[0].chunk_id == "allowed_best" && res.value()[1].chunk_id == "allowed_second");
        }
        CHECK(diag.usedExactScan && diag.rowsVisited == 2 && diag.exactDistanceEvaluations == 2 && diag.returnedRows == 2);
        auto one = db.searchSimilar({1, 0, 0, 0}, 4, -1.0f, std::optional<std::string>("blocked"), {});
        CHECK(one.has_value() && one.value().size() == 1 && one.value()[0].chunk_id == "blocked");
        auto none = db.searchSimilar({1, 0, 0, 0}, 4, -1.0f, std::optional<std::string>("blocked"), {"allowed"});
        CHECK(none.has_value() && none.value().empty());
    }
    for (auto order : {std::vector<std::string>{"tie_c", "tie_a", "tie_b"}, std::vector<std::string>{"tie_b", "tie_a", "tie_c"}}) {
        // "breaks score ties deterministically", the useMetadataFilter arm (vector_smoke_catch2_test.cpp:304-353)
        auto idxR = vector::createAccelVectorIndex(plugin, 4);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        for (auto& id : order) {
            vector::VectorRecord r; r.chunk_id = id; r.document_hash = "doc_" + id; r.embedding = {1, 0, 0, 0};
            r.metadata["lane"] = "tie";
            CHECK(db.insertVector(r).has_value());
        }
        auto res = db.searchSimilar({1, 0, 0, 0}, 2, -1.0f, std::nullopt, {}, {{"lane", "tie"}});
        CHECK(res.has_value() && res.value().size() == 2 && res.value()[0].chunk_id == "tie_a" && res.value()[1].chunk_id == "tie_b");
    }
    {   // metadata_filters: the record path (sqlite_vec_backend.cpp:4333-4409), AllMatching (:4398-4400)
        auto idxR = vector::createAccelVectorIndex(plugin, 4);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        auto insert = [&](const char* id, const char* lane, std::vector<float> e) {
            vector::VectorRecord r; r.chunk_id = id; r.document_hash = "d"; r.embedding = std::move(e);
            r.metadata["lane"] = lane; r.metadata["kind"] = "x";
            return db.insertVector(r);
        };
        CHECK(insert("a_best", "a", {1.0f, 0.0f, 0.0f, 0.0f}).has_value());
        CHECK(insert("a_second", "a", {0.8f, 0.6f, 0.0f, 0.0f}).has_value());
        CHECK(insert("b_best", "b", {1.0f, 0.0f, 0.0f, 0.0f}).has_value());
        CHECK(insert("a_tiny", "a", {5e-6f, 0.0f, 0.0f, 0.0f}).has_value());   // norm^2 = 2.5e-11
        vector::VectorSearchDiagnostics diag;
        auto res = db.searchSimilar({1, 0, 0, 0}, 4, -1.0f, std::nullopt, {}, {{"lane", "a"}, {"kind", "x"}}, &diag);
        CHECK(res.has_value() && res.value().size() == 2);   // a_tiny is a zero-norm row on this path (< 1e-10)
        if (res && res.value().size() == 2) CHECK(res.value()[0].chunk_id == "a_best" && res.value()[1].chunk_id == "a_second");
        CHECK(diag.rowsVisited == 4 && diag.exactDistanceEvaluations == 2 && diag.returnedRows == 2);
        auto plain = db.searchSimilar({1, 0, 0, 0}, 4, -1.0f);       // the fast path keeps it (norm^2 > 1e-12)
        CHECK(plain.has_value() && plain.value().size() == 4);
        auto miss = db.searchSimilar({1, 0, 0, 0}, 4, -1.0f, std::nullopt, {}, {{"lane", "a"}, {"kind", "y"}});
        CHECK(miss.has_value() && miss.value().empty());
        auto all = db.searchSimilar({1, 0, 0, 0}, 1, -1.0f, std::nullopt, {}, {{"kind", "x"}}, nullptr,
                                    vector::ExactRowSelection::AllMatching);
        CHECK(all.has_value() && all.value().size() == 3);           // k is ignored; a_tiny dropped
        if (all && all.value().size() == 3) CHECK(all.value()[0].chunk_id == "a_best" && all.value()[1].chunk_id == "b_best");
    }
    {   // incremental mirror: appends extend the device mirror, replace/delete leave tombstones
        auto idxR = vector::createAccelVectorIndex(plugin, 8);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        auto rec = [](int i, float tilt) {
            vector::VectorRecord r; r.chunk_id = "c" + std::to_string(1000 + i); r.document_hash = "doc";
            r.embedding = {1.0f, tilt * static_cast<float>(i + 1), 0, 0, 0, 0, 0, 0};
            return r;
        };
        std::vector<vector::VectorRecord> first;
        for (int i = 0; i < 100; ++i) first.push_back(rec(i, 0.01f));
        CHECK(db.insertVectorsBatch(first).has_value());
        auto r1 = db.searchSimilar({1, 0, 0, 0, 0, 0, 0, 0}, 1, -1.0f);
        CHECK(r1.has_value() && r1.value().front().chunk_id == "c1000" && db.uploadedRows() == 100);
        std::vector<vector::VectorRecord> more;
        for (int i = 100; i < 150; ++i) more.push_back(rec(i, 0.01f));
        more.push_back(rec(0, 5.0f));                                 // replaces c1000: now far from the query
        more.push_back(rec(0, 7.0f));                                 // same id twice in one batch: last write wins
        CHECK(db.insertVectorsBatch(more).has_value());
        CHECK(db.getVectorCount().value() == 150 && db.mirrorRows() == 151);
        auto r2 = db.searchSimilar({1, 0, 0, 0, 0, 0, 0, 0}, 2, -1.0f);
        CHECK(r2.has_value() && r2.value().size() == 2 && r2.value()[0].chunk_id == "c1001" && r2.value()[1].chunk_id == "c1002");
        CHECK(db.uploadedRows() == 151);                              // only the 51 new rows went up
        CHECK(db.deleteVector("c1001").has_value());
        auto r3 = db.searchSimilarBatch({{1, 0, 0, 0, 0, 0, 0, 0}}, 1, -1.0f);
        CHECK(r3.has_value() && r3.value()[0].front().chunk_id == "c1002" && db.uploadedRows() == 151);
        auto gone = db.searchSimilar({1, 7.0f, 0, 0, 0, 0, 0, 0}, 1, -1.0f);
        CHECK(gone.has_value() && gone.value().front().chunk_id == "c1000" && gone.value().front().relevance_score > 0.9999f);
        // compaction: more than 1024 tombstones and more than a quarter of the mirror
        std::vector<vector::VectorRecord> bulk;
        for (int i = 200; i < 3200; ++i) bulk.push_back(rec(i, 0.001f));
        CHECK(db.insertVectorsBatch(bulk).has_value());
        for (int i = 200; i < 1800; ++i) CHECK(db.deleteVector("c" + std::to_string(1000 + i)).has_value());
        auto r4 = db.searchSimilar({1, 0, 0, 0, 0, 0, 0, 0}, 1, -1.0f);
        CHECK(r4.has_value() && r4.value().front().chunk_id == "c1002");
        CHECK(db.mirrorRows() == db.getVectorCount().value() && db.uploadedRows() == db.mirrorRows());
    }
    {   // mixed dimensions in one table: a search only sees rows of the query's dimension (:4147)
        vector::AccelVectorTable table(plugin);
        auto rec = [](const char* id, std::vector<float> e) { vector::VectorRecord r; r.chunk_id = id; r.document_hash = "d"; r.embedding = std::move(e); return r; };
        CHECK(table.insertVectorsBatch({rec("a4", {1, 0, 0, 0}), rec("b8", {1, 0, 0, 0, 0, 0, 0, 0}), rec("c4", {0.6f, 0.8f, 0, 0})}).has_value());
        CHECK(table.getVectorCount().value() == 3);
        auto r4 = table.searchSimilar({1, 0, 0, 0}, 5, -1.0f);
        CHECK(r4.has_value() && r4.value().size() == 2 && r4.value()[0].chunk_id == "a4" && r4.value()[1].chunk_id == "c4");
        auto r8 = table.searchSimilar({1, 0, 0, 0, 0, 0, 0, 0}, 5, -1.0f);
        CHECK(r8.has_value() && r8.value().size() == 1 && r8.value()[0].chunk_id == "b8");
        auto r3 = table.searchSimilar({1, 0, 0}, 5, -1.0f);
        CHECK(r3.has_value() && r3.value().empty());
        CHECK(table.insertVector(rec("a4", {0, 1, 0, 0, 0, 0, 0, 0})).has_value());   // a4 moves to dimension 8
        CHECK(table.getVectorCount().value() == 3);
        auto again = table.searchSimilar({1, 0, 0, 0}, 5, -1.0f);
        CHECK(again.has_value() && again.value().size() == 1 && again.value()[0].chunk_id == "c4");
        CHECK(table.deleteVector("b8").has_value() && !table.deleteVector("b8").has_value());
    }
    {   // document restrictions over a larger mirror: the document_hash index must select exactly
        // the rows a second index holding ONLY those rows would score; ids arrive in ascending
        // order (no rank table is uploaded: row order is the tie order) until one does not
        const int N = 6000, DOCS = 40;
        auto a = vector::createAccelVectorIndex(plugin, 32), b = vector::createAccelVectorIndex(plugin, 32);
        auto &all = *a.value(), &only = *b.value();
        CHECK(all.initialize().has_value() && only.initialize().has_value());
        uint32_t x = 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return static_cast<float>((x >> 8) & 0xffff) / 32768.0f - 1.0f; };
        std::vector<vector::VectorRecord> recs, sub;
        const std::unordered_set<std::string> want = {"doc_3", "doc_17", "doc_39", "doc_none"};
        char id[32];
        for (int i = 0; i < N; ++i) {
            vector::VectorRecord r;
            std::snprintf(id, sizeof id, "chunk_%06d", i);
            r.chunk_id = id; r.document_hash = "doc_" + std::to_string(i % DOCS);
            r.embedding.resize(32);
            for (auto& v : r.embedding) v = rnd();
            if (i % 500 == 7) r.embedding = recs[i - 7].embedding;        // exact ties, broken by chunk_id
            recs.push_back(r);
            if (want.count(r.document_hash)) sub.push_back(r);
        }
        CHECK(all.insertVectorsBatch(recs).has_value() && only.insertVectorsBatch(sub).has_value());
        auto same = [&](const std::vector<vector::VectorRecord>& p, const std::vector<vector::VectorRecord>& q) {
            if (p.size() != q.size()) return false;
            for (size_t i = 0; i < p.size(); ++i)
                if (p[i].chunk_id != q[i].chunk_id || p[i].relevance_score != q[i].relevance_score) return false;
            return true;
        };
        for (int t = 0; t < 3; ++t) {
            std::vector<float> qv(32);
            for (auto& v : qv) v = rnd();
            if (t == 2) qv = recs[500].embedding;                         // a query sitting on a tie pair
            vector::VectorSearchDiagnostics d1, d2;
            auto f = all.searchSimilar(qv, 25, -1.0f, std::nullopt, want, {}, &d1);
            auto g = only.searchSimilar(qv, 25, -1.0f, &d2);
            CHECK(f.has_value() && g.has_value() && same(f.value(), g.value()));
            CHECK(d1.rowsVisited == sub.size() && d1.exactDistanceEvaluations == sub.size() && d2.rowsVisited == sub.size());
            auto one = all.searchSimilar(qv, 10, -1.0f, std::optional<std::string>("doc_17"), want);
            CHECK(one.has_value() && one.value().size() == 10);
            if (one) for (const auto& r : one.value()) CHECK(r.document_hash == "doc_17");
            auto excl = all.searchSimilar(qv, 10, -1.0f, std::optional<std::string>("doc_5"), want); // doc_5 not a candidate
            CHECK(excl.has_value() && excl.value().empty());
        }
        {   // k above YAMS_SCAN_MAX_K (the reference takes any k, :4299-4303): rounds behind the allow-mask must
            // give exactly the first k of the full ordering — checked against an AllMatching search (every row,
            // sorted by the reference's comparator) and across the round boundaries, ties included
            std::vector<float> qv = recs[500].embedding;
            auto full = all.searchSimilar(qv, 1, -1.0f, std::nullopt, {}, {}, nullptr, vector::ExactRowSelection::AllMatching);
            CHECK(full.has_value() && full.value().size() == static_cast<size_t>(N));
            for (size_t kk : {size_t(1025), size_t(2500), size_t(N + 10)}) {
                auto big = all.searchSimilar(qv, kk, -1.0f);
                CHECK(big.has_value() && big.value().size() == std::min<size_t>(kk, N));
                if (big && full) {
                    auto head = full.value(); head.resize(std::min<size_t>(kk, N));
                    CHECK(same(big.value(), head));
                }
            }
            auto thr = all.searchSimilar(qv, 3000, 0.2f);             // fewer rows above the threshold than k
            size_t above = 0;
            if (full) for (const auto& r : full.value()) above += r.relevance_score >= 0.2f;
            CHECK(thr.has_value() && thr.value().size() == std::min<size_t>(above, 3000));
            auto batch = all.searchSimilarBatch({qv, recs[3].embedding}, 1500, -1.0f);
            CHECK(batch.has_value() && batch.value().size() == 2 && batch.value()[0].size() == 1500 && batch.value()[1].size() == 1500);
            if (batch && full) { auto head = full.value(); head.resize(1500); CHECK(same(batch.value()[0], head)); }
            // vec0 engine: ranked by distance, the similarity threshold applied after the cut at k (:4506-4510)
            auto l2r = vector::createAccelVectorIndex(plugin, 32, vector::VectorSearchEngine::Vec0L2);
            auto& l2 = *l2r.value();
            CHECK(l2.initialize().has_value() && l2.insertVectorsBatch(recs).has_value());
            auto near1 = l2.searchSimilar(qv, 1024, -1.0f), near2 = l2.searchSimilar(qv, 2000, -1.0f);
            CHECK(near1.has_value() && near2.has_value() && near2.value().size() == 2000);
            if (near1 && near2) { auto head = near2.value(); head.resize(1024); CHECK(same(near1.value(), head)); }
            auto cut = l2.searchSimilar(qv, 2000, 0.1f);
            size_t keep = 0;
            if (near2) for (const auto& r : near2.value()) keep += r.relevance_score >= 0.1f;
            CHECK(cut.has_value() && cut.value().size() == keep && keep > 0 && keep < 2000);
            // vec0 L2 self-calibration (l2_calibration.hpp): the HOST's distance function decides the arithmetic.  Each of the
            // seven served definitions (fp64; fp32 in 1 / 8 / 16 lanes, plain and with a fused multiply-add) plays the host in turn: the index must recognise it and from then on return exactly
            // the rows a host-side brute force UNDER THAT DEFINITION returns (distance asc, chunk_id asc) — whatever
            // "l2_accumulate" the plugin was configured with.
            namespace l2n = vector::accel_l2;
            for (auto def : l2n::kDefinitions) {
                auto host = [def](const float* a, const float* b, size_t dim, float* out) { *out = l2n::distance(def, a, b, dim); return true; };
                auto cal = l2.calibrateL2(host);
                CHECK(cal.has_value() && cal.value().matched && cal.value().accumulate == def);
                CHECK(l2.l2().calibrated && l2.l2().matched && l2.l2().flags == (static_cast<uint32_t>(def) | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT));
                auto got = l2.searchSimilar(qv, 60, -1.0f);
                CHECK(got.has_value() && got.value().size() == 60);
                std::vector<std::pair<float, std::string>> brute;
                for (const auto& r : recs) brute.emplace_back(l2n::distance(def, r.embedding.data(), qv.data(), qv.size()), r.chunk_id);
                std::sort(brute.begin(), brute.end());
                if (got) for (size_t i = 0; i < 60; ++i) CHECK(got.value()[i].chunk_id == brute[i].second);
            }
            // a host whose arithmetic is none of them (pairwise summation): calibration says so and L2 is REFUSED, not guessed
            auto pairwise = [](const float* a, const float* b, size_t dim, float* out) {
                std::vector<float> v(dim);
                for (size_t i = 0; i < dim; ++i) { const float d = a[i] - b[i]; v[i] = d * d; }
                for (size_t n = dim; n > 1; n = (n + 1) / 2) for (size_t i = 0; i < n / 2; ++i) v[i] = v[i] + v[n - 1 - i];
                *out = std::sqrt(v[0]);
                return true;
            };
            auto none = l2.calibrateL2(pairwise);
            CHECK(none.has_value() && !none.value().matched && !l2.l2().matched);
            auto refused = l2.searchSimilar(qv, 10, -1.0f);
            CHECK(!refused.has_value() && refused.error().code == ErrorCode::NotSupported);
            auto refusedBig = l2.searchSimilar(qv, 2000, -1.0f);
            CHECK(!refusedBig.has_value() && refusedBig.error().code == ErrorCode::NotSupported);
            // ... while the cosine engine of the same plugin is untouched
            CHECK(all.searchSimilar(qv, 5, -1.0f).has_value());
        }
        auto ties = all.searchSimilar(recs[500].embedding, 2, -1.0f);
        CHECK(ties.has_value() && ties.value().size() == 2 && ties.value()[0].chunk_id == "chunk_000500" && ties.value()[1].chunk_id == "chunk_000507");
        // an id below the current maximum ends the append-order shortcut: ranks are uploaded and the
        // new row sorts in front of its twins
        vector::VectorRecord early = recs[500];
        early.chunk_id = "chunk_000000_b"; early.document_hash = "doc_17";
        CHECK(all.insertVector(early).has_value());
        auto ties2 = all.searchSimilar(recs[500].embedding, 3, -1.0f);
        CHECK(ties2.has_value() && ties2.value().size() == 3 && ties2.value()[0].chunk_id == "chunk_000000_b" &&
              ties2.value()[1].chunk_id == "chunk_000500");
        CHECK(all.deleteVectorsByDocument("doc_17").has_value() && !all.hasEmbedding("doc_17").value());
        CHECK(all.getVectorsByDocument("doc_3").value().size() == static_cast<size_t>(N / DOCS));
        auto gone = all.searchSimilar(recs[17].embedding, 5, -1.0f, std::optional<std::string>("doc_17"), {});
        CHECK(gone.has_value() && gone.value().empty());
    }
    {   // device memory exhausted while a mirror grows (allocation-failure injection, yams_accel_debug_fail_alloc_after):
        // ErrorCode::ResourceExhausted (core/types.h:49), nothing lost on the host side, the same search succeeds — new rows
        // included — once memory is there again
        using fail_after_t = void (*)(int64_t);
        void* self = dlopen(argv[1], RTLD_NOW | RTLD_NOLOAD);
        auto fail_after = self ? reinterpret_cast<fail_after_t>(dlsym(self, "yams_accel_debug_fail_alloc_after")) : nullptr;
        auto compiled = self ? reinterpret_cast<int (*)()>(dlsym(self, "yams_accel_debug_alloc_injection_compiled")) : nullptr;
        CHECK(fail_after != nullptr && compiled != nullptr);
        // (the injection exists in the measurement build only: the product library's doors do nothing, and this block is
        // run by tests/test_cpp_host.py against libyams_mi355x_accel_measure.so)
        if (fail_after && compiled && compiled() == 0) std::printf("allocation-failure injection: not compiled into this library, block skipped\n");
        if (fail_after && compiled && compiled() == 1) {
            std::printf("allocation-failure injection: exercised\n");
            auto idxR = vector::createAccelVectorIndex(plugin, 64);
            auto& db = *idxR.value();
            CHECK(db.initialize().has_value());
            auto row = [](size_t i) {
                vector::VectorRecord r;
                char id[32]; std::snprintf(id, sizeof id, "oom_%07zu", i);
                r.chunk_id = id; r.document_hash = "doc_oom"; r.embedding.assign(64, 0.0f);
                r.embedding[i % 64] = 1.0f; r.embedding[(i / 64) % 64] += 0.5f; r.embedding[(i / 4096) % 64] += 0.25f;
                return r;
            };
            std::vector<vector::VectorRecord> first, more;
            for (size_t i = 0; i < 1000; ++i) first.push_back(row(i));
            for (size_t i = 1000; i < 201000; ++i) more.push_back(row(i));   // 51 MB of rows: the mirror must map more memory
            CHECK(db.insertVectorsBatch(first).has_value());
            auto before = db.searchSimilar(first[7].embedding, 5, -1.0f);
            CHECK(before.has_value() && before.value().size() == 5 && before.value()[0].chunk_id == "oom_0000007");
            fail_after(0);
            CHECK(db.insertVectorsBatch(more).has_value());                 // (host side only: the upload is lazy)
            auto starved = db.searchSimilar(more[12345].embedding, 5, -1.0f);
            CHECK(!starved.has_value() && starved.error().code == ErrorCode::ResourceExhausted);
            auto again = db.searchSimilar(more[12345].embedding, 5, -1.0f);  // still exhausted: still refused, nothing corrupted
            CHECK(!again.has_value() && again.error().code == ErrorCode::ResourceExhausted);
            fail_after(-1);
            auto fed = db.searchSimilar(more[12345].embedding, 5, -1.0f);
            CHECK(fed.has_value() && fed.value().size() == 5 && fed.value()[0].chunk_id == more[12345].chunk_id);
            auto old = db.searchSimilar(first[7].embedding, 5, -1.0f);
            CHECK(old.has_value() && old.value().size() == 5 && old.value()[0].chunk_id == "oom_0000007");
        }
    }
    {   // large finite scores (+-FLT_MAX/4) stay finite
        const float L = std::numeric_limits<float>::max() / 4.0f;
        auto idxR = vector::createAccelVectorIndex(plugin, 4);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        vector::VectorRecord r; r.chunk_id = "large_finite"; r.embedding = {L, -L, L, -L};
        CHECK(db.insertVector(r).has_value());
        auto res = db.searchSimilar(r.embedding, 1, -1.0f);
        CHECK(res.has_value() && res.value().size() == 1 && std::isfinite(res.value()[0].relevance_score) && res.value()[0].relevance_score > 0.999f);
    }
    if (config.find("devices") == std::string::npos) {
        // ---- the product-quantised engine through the adapter (round 6): AccelVectorIndex::setPqIndex / searchPqBatch against
        //      a host restatement of simeonPqSearchUnlocked (sqlite_vec_backend.cpp:3868-4056).  The quantiser is a stand-in
        //      (random codes and tables: third_party/simeon is absent) — what is checked is everything the reference's function
        //      does around it: ADC sum, (score desc, tie key asc), approxK, the exact re-rank, rows the mirror lost, the final
        //      (similarity desc, chunk_id asc) order.  (Single-device corpora: a striped corpus answers NotImplemented.)
        const size_t n = 3000, d = 64, m = 8, k = 10, rf = 3;
        std::mt19937 rng(99);
        std::normal_distribution<float> nd(0.f, 1.f);
        auto idxR = vector::createAccelVectorIndex(plugin, d);
        auto& db = *idxR.value();
        CHECK(db.initialize().has_value());
        std::vector<vector::VectorRecord> recs(n);
        for (size_t i = 0; i < n; ++i) {
            char id[32]; std::snprintf(id, sizeof id, "pq_%05zu", (i * 7919) % 100003);
            recs[i].chunk_id = id; recs[i].document_hash = "doc";
            recs[i].embedding.resize(d);
            for (auto& v : recs[i].embedding) v = nd(rng) * (1.0f + float(i % 7));
        }
        for (size_t i = 40; i < 48; ++i) recs[i].embedding = recs[7].embedding;            // equal exact similarities: chunk ids decide
        CHECK(db.insertVectorsBatch(recs).has_value());
        std::vector<uint8_t> codes(n * m);
        for (auto& c : codes) c = static_cast<uint8_t>(rng() & 255u);
        for (size_t i = 40; i < 48; ++i) std::copy(codes.begin() + 7 * m, codes.begin() + 8 * m, codes.begin() + i * m); // equal ADC scores: tie keys decide
        std::vector<std::string> ids(n);
        for (size_t i = 0; i < n; ++i) ids[i] = recs[i].chunk_id;
        ids.push_back("pq_gone_1"); ids.push_back("pq_gone_2");                             // indexed rows the table no longer holds
        codes.resize((n + 2) * m, 3);
        CHECK(db.setPqIndex(codes, m, ids).has_value());
        std::vector<std::vector<float>> queries(3, std::vector<float>(d)), luts(3, std::vector<float>(m * 256));
        for (auto& q : queries) for (auto& v : q) v = nd(rng);
        for (auto& l : luts) for (auto& v : l) v = nd(rng);
        for (auto& v : luts[0]) v = 0.f;                                                    // every ADC score equal: the tie keys alone order the candidates
        std::vector<uint32_t> cand;
        for (uint32_t i = 1; i < n + 2; i += 3) cand.push_back(i);
        for (int pass = 0; pass < 2; ++pass) {
            const std::vector<uint32_t>* cd = pass ? &cand : nullptr;
            auto got = db.searchPqBatch(queries, luts, k, -1.0f, rf, cd);
            CHECK(got.has_value());
            if (!got.has_value()) continue;
            for (size_t qi = 0; qi < queries.size(); ++qi) {
                struct Hit { float s; uint64_t key; size_t idx; };
                std::vector<Hit> hits;
                const size_t cnt = cd ? cd->size() : n + 2;
                for (size_t c = 0; c < cnt; ++c) {
                    const size_t i = cd ? (*cd)[c] : c;
                    float acc = 0.f;
                    for (size_t j = 0; j < m; ++j) acc = acc + luts[qi][j * 256 + codes[i * m + j]];
                    hits.push_back({acc, vector::AccelVectorIndex::stableStringKey(ids[i]), i});
                }
                std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.s != b.s ? a.s > b.s : (a.key != b.key ? a.key < b.key : a.idx < b.idx); });
                hits.resize(std::min(hits.size(), std::max(k, k * rf)));
                std::vector<std::pair<float, std::string>> want;
                for (const auto& h : hits) {
                    if (h.idx >= n) continue;                                                // getVectorByRowidUnlocked finds nothing
                    double dp = 0, na = 0, nb = 0;
                    for (size_t t = 0; t < d; ++t) { const double a = queries[qi][t], b = recs[h.idx].embedding[t]; dp += a * b; na += a * a; nb += b * b; }
                    na = std::sqrt(na); nb = std::sqrt(nb);
                    want.emplace_back(static_cast<float>((na == 0 || nb == 0) ? 0.0 : dp / (na * nb)), recs[h.idx].chunk_id);
                }
                std::sort(want.begin(), want.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
                if (want.size() > k) want.resize(k);
                const auto& g = got.value()[qi];
                CHECK(g.size() == want.size());
                for (size_t i = 0; i < std::min(g.size(), want.size()); ++i) {
                    CHECK(g[i].chunk_id == want[i].second);
                    CHECK(std::memcmp(&g[i].relevance_score, &want[i].first, 4) == 0);
                }
            }
        }
        std::vector<uint32_t> none;
        auto empty = db.searchPqBatch(queries, luts, k, -1.0f, rf, &none);                  // an empty candidate list: nothing (:3946-3948)
        CHECK(empty.has_value() && empty.value().size() == 3 && empty.value()[0].empty());
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
