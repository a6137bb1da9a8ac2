// plugin.cpp — the YAMS plugin surface of libyams_mi355x_accel.so.
//
// Exports the eight entry points of the reference's include/yams/plugins/abi.h:26-34 and serves
// three interface vtables (vector_scan_v1, content_hash_v1, chunker_v1) written to the
// conventions of include/yams/plugins/model_provider_v1.h:44-49.  The host side that would load
// this file is AbiPluginLoader::load / getInterface (src/daemon/resource/abi_plugin_loader.cpp:
// 270-442, 657-681): dlopen(RTLD_LAZY|RTLD_LOCAL), yams_plugin_init(config_json, host_context),
// yams_plugin_get_manifest_json, then yams_plugin_get_interface(id, version, &vtable).
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "accel_ctx.h"

#define YAMS_PLUGIN_API __attribute__((visibility("default")))
#define YAMS_PLUGIN_ABI_VERSION 1          // abi.h:18
#define YAMS_PLUGIN_OK 0                   // abi.h:20-24
#define YAMS_PLUGIN_ERR_INCOMPATIBLE -1
#define YAMS_PLUGIN_ERR_NOT_FOUND -2
#define YAMS_PLUGIN_ERR_INIT_FAILED -3
#define YAMS_PLUGIN_ERR_INVALID -4

namespace {

struct Corpus {
    uint32_t dim = 0;
    uint64_t n_rows = 0, cap_rows = 0;
    float* d_rows = nullptr;
    uint16_t* d_bf16 = nullptr; // filter shadow of d_rows (same capacity), kept in step by corpus_append
    float* d_nsq = nullptr;
    uint32_t* d_tie = nullptr;
    uint32_t* d_inv = nullptr;
};

struct PluginState {
    std::mutex mu; // vtable functions must be thread-safe (model_provider_v1.h:45); one stream
    yams_accel_ctx* ctx = nullptr;
    int device = 0;
    bool initialised = false;
    std::string init_error;
    std::map<uint64_t, Corpus> corpora;
    uint64_t next_id = 1;
    uint64_t searches = 0, hashes = 0, chunk_calls = 0;
};
PluginState g;

const char kManifest[] =
    "{\"name\":\"yams_mi355x_accel\",\"version\":\"" YAMS_ACCEL_VERSION_STRING "\",\"abi\":1,"
    "\"description\":\"MI355X (gfx950) exact vector scan, SHA-256 and content-defined chunking\","
    "\"interfaces\":[{\"id\":\"vector_scan_v1\",\"version\":1},"
    "{\"id\":\"content_hash_v1\",\"version\":1},{\"id\":\"chunker_v1\",\"version\":1}]}";

int parse_device(const char* json) {
    if (!json) return 0;
    const char* p = std::strstr(json, "\"device\"");
    if (!p) return 0;
    p = std::strchr(p, ':');
    if (!p) return 0;
    return std::atoi(p + 1);
}

void free_corpus(Corpus& c) {
    if (c.d_rows) (void)hipFree(c.d_rows);
    if (c.d_bf16) (void)hipFree(c.d_bf16);
    if (c.d_nsq) (void)hipFree(c.d_nsq);
    if (c.d_tie) (void)hipFree(c.d_tie);
    if (c.d_inv) (void)hipFree(c.d_inv);
    c = Corpus{};
}

#define NEED_CTX() do { if (!g.ctx) return YAMS_ERR_UNSUPPORTED; } while (0)

// ---- vector_scan_v1 ---------------------------------------------------------------------------
yams_status_t vs_corpus_create(void*, uint32_t dim, uint64_t* out_id) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!out_id || dim == 0) return YAMS_ERR_INVALID_ARG;
    Corpus c; c.dim = dim;
    const uint64_t id = g.next_id++;
    g.corpora[id] = c;
    *out_id = id;
    return YAMS_OK;
}

yams_status_t vs_corpus_append(void*, uint64_t id, const float* rows, uint64_t n_rows) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    auto it = g.corpora.find(id);
    if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
    if (n_rows == 0) return YAMS_OK;
    if (!rows) return YAMS_ERR_INVALID_ARG;
    Corpus& c = it->second;
    (void)hipSetDevice(g.ctx->device);
    (void)hipStreamSynchronize(g.ctx->stream);
    const uint64_t need = c.n_rows + n_rows;
    if (need > c.cap_rows) {
        uint64_t cap = std::max<uint64_t>(need, c.cap_rows + c.cap_rows / 2);
        float* nd = nullptr;
        if (hipMalloc(&nd, cap * c.dim * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return YAMS_ERR_INTERNAL; }
        if (c.n_rows && hipMemcpy(nd, c.d_rows, c.n_rows * c.dim * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(nd); return YAMS_ERR_INTERNAL;
        }
        if (c.d_rows) (void)hipFree(c.d_rows);
        c.d_rows = nd;
        if ((c.dim & 3u) == 0) { // the shadow grows with the mirror
            uint16_t* nb = nullptr; float* nn = nullptr;
            if (hipMalloc(&nb, cap * c.dim * sizeof(uint16_t)) != hipSuccess || hipMalloc(&nn, cap * sizeof(float)) != hipSuccess) {
                (void)hipGetLastError(); if (nb) (void)hipFree(nb); return YAMS_ERR_INTERNAL;
            }
            if (c.n_rows && (hipMemcpy(nb, c.d_bf16, c.n_rows * c.dim * sizeof(uint16_t), hipMemcpyDeviceToDevice) != hipSuccess ||
                             hipMemcpy(nn, c.d_nsq, c.n_rows * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess)) {
                (void)hipGetLastError(); (void)hipFree(nb); (void)hipFree(nn); return YAMS_ERR_INTERNAL;
            }
            if (c.d_bf16) (void)hipFree(c.d_bf16);
            if (c.d_nsq) (void)hipFree(c.d_nsq);
            c.d_bf16 = nb; c.d_nsq = nn;
        }
        c.cap_rows = cap;
    }
    if (hipMemcpy(c.d_rows + c.n_rows * c.dim, rows, n_rows * c.dim * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError(); return YAMS_ERR_INTERNAL;
    }
    if (c.d_bf16) {
        if (yams_scan_build_shadow_device(g.ctx, c.d_rows + c.n_rows * c.dim, n_rows, c.dim,
                                          c.d_bf16 + c.n_rows * c.dim, c.d_nsq + c.n_rows) != YAMS_OK ||
            yams_accel_ctx_synchronize(g.ctx) != YAMS_OK)
            return YAMS_ERR_INTERNAL;
    }
    c.n_rows = need;
    // appended rows invalidate a previously supplied chunk_id ranking
    if (c.d_tie) { (void)hipFree(c.d_tie); c.d_tie = nullptr; }
    if (c.d_inv) { (void)hipFree(c.d_inv); c.d_inv = nullptr; }
    return YAMS_OK;
}

yams_status_t vs_corpus_set_tie_ranks(void*, uint64_t id, const uint32_t* ranks, uint64_t n_rows) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    auto it = g.corpora.find(id);
    if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
    Corpus& c = it->second;
    if (n_rows != c.n_rows || (!ranks && n_rows)) return YAMS_ERR_INVALID_ARG;
    std::vector<uint32_t> inv(n_rows, 0xffffffffu);
    for (uint64_t r = 0; r < n_rows; ++r) {
        if (ranks[r] >= n_rows || inv[ranks[r]] != 0xffffffffu) return YAMS_ERR_INVALID_ARG; // not a permutation
        inv[ranks[r]] = static_cast<uint32_t>(r);
    }
    (void)hipSetDevice(g.ctx->device);
    (void)hipStreamSynchronize(g.ctx->stream);
    if (c.d_tie) { (void)hipFree(c.d_tie); c.d_tie = nullptr; }
    if (c.d_inv) { (void)hipFree(c.d_inv); c.d_inv = nullptr; }
    if (n_rows == 0) return YAMS_OK;
    if (hipMalloc(&c.d_tie, n_rows * 4) != hipSuccess || hipMalloc(&c.d_inv, n_rows * 4) != hipSuccess) {
        (void)hipGetLastError(); return YAMS_ERR_INTERNAL;
    }
    if (hipMemcpy(c.d_tie, ranks, n_rows * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c.d_inv, inv.data(), n_rows * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError(); return YAMS_ERR_INTERNAL;
    }
    return YAMS_OK;
}

yams_status_t vs_corpus_clear(void*, uint64_t id) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    auto it = g.corpora.find(id);
    if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
    (void)hipStreamSynchronize(g.ctx->stream);
    const uint32_t dim = it->second.dim;
    free_corpus(it->second);
    it->second.dim = dim;
    return YAMS_OK;
}

yams_status_t vs_corpus_destroy(void*, uint64_t id) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    auto it = g.corpora.find(id);
    if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
    (void)hipStreamSynchronize(g.ctx->stream);
    free_corpus(it->second);
    g.corpora.erase(it);
    return YAMS_OK;
}

yams_status_t vs_corpus_size(void*, uint64_t id, uint64_t* out_rows, uint32_t* out_dim) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    auto it = g.corpora.find(id);
    if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
    if (out_rows) *out_rows = it->second.n_rows;
    if (out_dim) *out_dim = it->second.dim;
    return YAMS_OK;
}

yams_status_t vs_search_batch_ex(void*, uint64_t id, const float* queries, uint32_t nq, uint32_t dim,
                                 uint32_t k, float threshold, uint32_t metric, uint32_t flags,
                                 const uint32_t* row_mask_host, yams_scan_hit_t** out_hits,
                                 uint32_t** out_counts, yams_scan_diag_t* out_diag) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!out_hits || !out_counts) return YAMS_ERR_INVALID_ARG;
    *out_hits = nullptr; *out_counts = nullptr;
    auto it = g.corpora.find(id);
    if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
    const Corpus& c = it->second;
    // dimension mismatch -> InvalidArgument (vector_database.cpp:545-550, 626-633)
    if (dim != c.dim) return YAMS_ERR_INVALID_ARG;
    if (nq && !queries) return YAMS_ERR_INVALID_ARG;
    yams_scan_corpus_t view{};
    view.rows = c.d_rows; view.n_rows = c.n_rows; view.dim = c.dim;
    view.tie_rank = c.d_tie; view.rank_row = c.d_inv; view.row_base = 0;
    view.rows_bf16 = c.d_bf16; view.rows_nsq = c.d_bf16 ? c.d_nsq : nullptr;
    if (row_mask_host && c.n_rows) { // document_hash / candidate_hashes restriction (:4137-4175)
        const size_t words = (c.n_rows + 31) / 32;
        uint64_t bits = 0;
        for (size_t i = 0; i < words; ++i) {
            uint32_t w = row_mask_host[i];
            if (i == words - 1 && (c.n_rows & 31)) w &= (1u << (c.n_rows & 31)) - 1u;
            bits += static_cast<uint64_t>(__builtin_popcount(w));
        }
        uint32_t* d_mask = nullptr;
        if (yams_accel::ws_get(g.ctx, "plugin_row_mask", words * 4, (void**)&d_mask) != YAMS_OK) return YAMS_ERR_INTERNAL;
        if (yams_accel_upload(g.ctx, d_mask, row_mask_host, words * 4) != YAMS_OK) return YAMS_ERR_INTERNAL;
        view.row_mask = d_mask; view.row_mask_count = bits;
    }
    // only semantic flags cross the vtable; filter selection stays with the library
    yams_scan_params_t prm{k, threshold, metric, flags & (YAMS_SCAN_FLAG_RECORD_PATH | YAMS_SCAN_FLAG_FORCE_EXACT)};
    const size_t slots = static_cast<size_t>(nq) * std::max<uint32_t>(k, 1);
    std::vector<float> scores(slots), dist(slots);
    std::vector<int64_t> rows(slots);
    auto* counts = static_cast<uint32_t*>(std::calloc(std::max<uint32_t>(nq, 1), sizeof(uint32_t)));
    auto* hits = static_cast<yams_scan_hit_t*>(std::calloc(std::max<size_t>(slots, 1), sizeof(yams_scan_hit_t)));
    if (!counts || !hits) { std::free(counts); std::free(hits); return YAMS_ERR_INTERNAL; }
    yams_status_t s = yams_scan_topk_host(g.ctx, &view, queries, nq, &prm, scores.data(), rows.data(),
                                          counts, dist.data(), out_diag);
    if (s != YAMS_OK) { std::free(counts); std::free(hits); return s; }
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t i = 0; i < k; ++i) {
            const size_t o = static_cast<size_t>(q) * k + i;
            if (i < counts[q]) { hits[o].row = rows[o]; hits[o].similarity = scores[o]; hits[o].distance = dist[o]; }
            else { hits[o].row = -1; hits[o].similarity = 0.f; hits[o].distance = 0.f; }
        }
    ++g.searches;
    *out_hits = hits; *out_counts = counts;
    return YAMS_OK;
}

yams_status_t vs_search_batch_masked(void* self, uint64_t id, const float* queries, uint32_t nq, uint32_t dim,
                                     uint32_t k, float threshold, uint32_t metric,
                                     const uint32_t* row_mask_host, yams_scan_hit_t** out_hits,
                                     uint32_t** out_counts, yams_scan_diag_t* out_diag) {
    return vs_search_batch_ex(self, id, queries, nq, dim, k, threshold, metric, 0, row_mask_host, out_hits,
                              out_counts, out_diag);
}

yams_status_t vs_search_batch(void* self, uint64_t id, const float* queries, uint32_t nq, uint32_t dim,
                              uint32_t k, float threshold, uint32_t metric,
                              yams_scan_hit_t** out_hits, uint32_t** out_counts,
                              yams_scan_diag_t* out_diag) {
    return vs_search_batch_masked(self, id, queries, nq, dim, k, threshold, metric, nullptr, out_hits,
                                  out_counts, out_diag);
}

void vs_free_hits(void*, yams_scan_hit_t* hits, uint32_t* counts) { std::free(hits); std::free(counts); }

yams_status_t vs_runtime_info(void*, char** out_json) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    return yams_accel_device_info_json(g.ctx, out_json);
}
void vs_free_string(void*, char* s) { std::free(s); }

yams_vector_scan_v1 g_vector_scan = {
    YAMS_IFACE_VECTOR_SCAN_V1_VERSION, nullptr, vs_corpus_create, vs_corpus_append,
    vs_corpus_set_tie_ranks, vs_corpus_clear, vs_corpus_destroy, vs_corpus_size, vs_search_batch,
    vs_free_hits, vs_runtime_info, vs_free_string, vs_search_batch_masked, vs_search_batch_ex};

// ---- content_hash_v1 --------------------------------------------------------------------------
yams_status_t ch_hash(void*, const uint8_t* data, size_t n, char out_hex[65]) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    ++g.hashes;
    return yams_sha256_host(g.ctx, data, n, out_hex);
}
yams_status_t ch_hash_many(void*, const uint8_t* const* msgs, const size_t* lens, size_t n, char* out_hex) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    g.hashes += n;
    return yams_sha256_many_host(g.ctx, msgs, lens, n, out_hex);
}

// Streaming state (sha256_hasher.cpp:81-109): the compression function runs on the device over
// whole 64-byte blocks; the host only buffers the partial block and builds the FIPS 180-4 padding.
struct HashStream {
    uint32_t state[8];
    uint8_t partial[64];
    size_t partial_len = 0;
    uint64_t total = 0;
};
const uint32_t kShaInit[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                              0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
}  // namespace

namespace yams_accel { // from ingest_kernels.hip
hipError_t launch_sha256(hipStream_t st, const uint8_t* data, const uint64_t* offs,
                         const uint64_t* lens, uint64_t n_long, uint64_t n_msgs, uint8_t* digests,
                         unsigned long long* queue_heads, const uint32_t* init_state,
                         uint32_t* out_state, int raw_blocks_only, uint32_t max_blocks, int slots);
}

namespace {
// Runs the compression function over `n` bytes (a multiple of 64) starting from hs->state.
yams_status_t stream_blocks(HashStream* hs, const uint8_t* bytes, size_t n) {
    using namespace yams_accel;
    if (n == 0) return YAMS_OK;
    yams_accel_ctx* ctx = g.ctx;
    (void)hipSetDevice(ctx->device);
    uint8_t* d_data; uint64_t* d_tab; uint32_t* d_state; unsigned long long* d_head;
    YA_TRY(ws_get(ctx, "hs_data", n + 64, (void**)&d_data));
    YA_TRY(ws_get(ctx, "hs_tab", 64, (void**)&d_tab));
    YA_TRY(ws_get(ctx, "hs_state", 64, (void**)&d_state));
    YA_TRY(ws_get(ctx, "ing_queue", 64, (void**)&d_head));
    const uint64_t tab[2] = {0, n};
    YA_HIP(ctx, hipMemcpyAsync(d_data, bytes, n, hipMemcpyHostToDevice, ctx->stream));
    YA_HIP(ctx, hipMemcpyAsync(d_tab, tab, 16, hipMemcpyHostToDevice, ctx->stream));
    YA_HIP(ctx, hipMemcpyAsync(d_state, hs->state, 32, hipMemcpyHostToDevice, ctx->stream));
    YA_HIP(ctx, launch_sha256(ctx->stream, d_data, d_tab, d_tab + 1, 0, 1, nullptr, d_head, d_state,
                              d_state + 8, 1, 1, 1));
    YA_HIP(ctx, hipMemcpyAsync(hs->state, d_state + 8, 32, hipMemcpyDeviceToHost, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return YAMS_OK;
}

yams_status_t ch_stream_create(void*, void** out) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!out) return YAMS_ERR_INVALID_ARG;
    auto* hs = new HashStream();
    std::memcpy(hs->state, kShaInit, 32);
    *out = hs;
    return YAMS_OK;
}
yams_status_t ch_stream_init(void*, void* s) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!s) return YAMS_ERR_INVALID_ARG;
    auto* hs = static_cast<HashStream*>(s);
    std::memcpy(hs->state, kShaInit, 32);
    hs->partial_len = 0; hs->total = 0;
    return YAMS_OK;
}
yams_status_t ch_stream_update(void*, void* s, const uint8_t* data, size_t n) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!s || (n && !data)) return YAMS_ERR_INVALID_ARG;
    auto* hs = static_cast<HashStream*>(s);
    hs->total += n;
    size_t pos = 0;
    if (hs->partial_len) {
        const size_t take = std::min(n, 64 - hs->partial_len);
        std::memcpy(hs->partial + hs->partial_len, data, take);
        hs->partial_len += take; pos = take;
        if (hs->partial_len < 64) return YAMS_OK;
        YA_TRY(stream_blocks(hs, hs->partial, 64));
        hs->partial_len = 0;
    }
    const size_t whole = (n - pos) / 64 * 64;
    if (whole) { YA_TRY(stream_blocks(hs, data + pos, whole)); pos += whole; }
    if (pos < n) { std::memcpy(hs->partial, data + pos, n - pos); hs->partial_len = n - pos; }
    return YAMS_OK;
}
yams_status_t ch_stream_finalize(void*, void* s, char out_hex[65]) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!s || !out_hex) return YAMS_ERR_INVALID_ARG;
    auto* hs = static_cast<HashStream*>(s);
    uint8_t tail[128];
    std::memset(tail, 0, sizeof tail);
    std::memcpy(tail, hs->partial, hs->partial_len);
    tail[hs->partial_len] = 0x80;
    const size_t tl = hs->partial_len < 56 ? 64 : 128;
    const uint64_t bits = hs->total * 8;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = static_cast<uint8_t>(bits >> (8 * i));
    YA_TRY(stream_blocks(hs, tail, tl));
    static const char kHex[] = "0123456789abcdef";
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) {
            const uint8_t v = static_cast<uint8_t>(hs->state[i] >> (24 - 8 * b));
            out_hex[8 * i + 2 * b] = kHex[v >> 4]; out_hex[8 * i + 2 * b + 1] = kHex[v & 15];
        }
    out_hex[64] = 0;
    // "Reset for potential reuse" (sha256_hasher.cpp:103-106)
    std::memcpy(hs->state, kShaInit, 32);
    hs->partial_len = 0; hs->total = 0;
    ++g.hashes;
    return YAMS_OK;
}
void ch_stream_destroy(void*, void* s) { delete static_cast<HashStream*>(s); }

// hex (either case) -> 32 raw bytes; false on anything that is not 64 hex digits
bool parse_hex32(const char* hex, uint8_t out[32]) {
    for (int i = 0; i < 32; ++i) {
        int v = 0;
        for (int j = 0; j < 2; ++j) {
            const char c = hex[2 * i + j];
            int d;
            if (c >= '0' && c <= '9') d = c - '0';
            else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
            else return false;
            v = v * 16 + d;
        }
        out[i] = static_cast<uint8_t>(v);
    }
    return hex[64] == 0;
}

yams_status_t ch_verify_many(void*, const uint8_t* const* msgs, const size_t* lens, const char* expected_hex,
                             size_t n, uint8_t* out_valid) {
    if (n == 0) return YAMS_OK;
    if (!msgs || !lens || !expected_hex || !out_valid) return YAMS_ERR_INVALID_ARG;
    std::vector<char> hex(n * 65);
    {
        std::lock_guard<std::mutex> lk(g.mu);
        NEED_CTX();
        g.hashes += n;
        yams_status_t s = yams_sha256_many_host(g.ctx, msgs, lens, n, hex.data());
        if (s != YAMS_OK) return s;
    }
    for (size_t i = 0; i < n; ++i) { // the reference compares the lower-case hex strings (:243-249)
        uint8_t a[32], b[32];
        out_valid[i] = (parse_hex32(hex.data() + 65 * i, a) && parse_hex32(expected_hex + 65 * i, b) &&
                        std::memcmp(a, b, 32) == 0) ? 1 : 0;
    }
    return YAMS_OK;
}

std::map<uint64_t, yams_dedup_set*> g_dedup;
uint64_t g_next_dedup = 1;

yams_status_t ch_dedup_create(void*, uint64_t expected, uint64_t* out_id) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!out_id) return YAMS_ERR_INVALID_ARG;
    yams_dedup_set* s = nullptr;
    yams_status_t st = yams_dedup_set_create(g.ctx, expected, &s);
    if (st != YAMS_OK) return st;
    *out_id = g_next_dedup++;
    g_dedup[*out_id] = s;
    return YAMS_OK;
}
yams_status_t dedup_call(uint64_t id, const char* hashes_hex, size_t n, uint8_t* out, bool insert) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    auto it = g_dedup.find(id);
    if (it == g_dedup.end()) return YAMS_ERR_NOT_FOUND;
    if (n == 0) return YAMS_OK;
    if (!hashes_hex || !out) return YAMS_ERR_INVALID_ARG;
    std::vector<uint8_t> raw(n * 32);
    for (size_t i = 0; i < n; ++i)
        if (!parse_hex32(hashes_hex + 65 * i, raw.data() + 32 * i)) return YAMS_ERR_INVALID_ARG;
    return insert ? yams_dedup_insert_host(it->second, raw.data(), n, out, nullptr)
                  : yams_dedup_probe_host(it->second, raw.data(), n, out);
}
yams_status_t ch_dedup_insert(void*, uint64_t id, const char* hex, size_t n, uint8_t* out) { return dedup_call(id, hex, n, out, true); }
yams_status_t ch_dedup_contains(void*, uint64_t id, const char* hex, size_t n, uint8_t* out) { return dedup_call(id, hex, n, out, false); }
yams_status_t ch_dedup_size(void*, uint64_t id, uint64_t* out_entries) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g_dedup.find(id);
    if (it == g_dedup.end()) return YAMS_ERR_NOT_FOUND;
    return yams_dedup_set_size(it->second, out_entries);
}
yams_status_t ch_dedup_destroy(void*, uint64_t id) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g_dedup.find(id);
    if (it == g_dedup.end()) return YAMS_ERR_NOT_FOUND;
    yams_dedup_set_destroy(it->second);
    g_dedup.erase(it);
    return YAMS_OK;
}

yams_content_hash_v1 g_content_hash = {YAMS_IFACE_CONTENT_HASH_V1_VERSION, nullptr, ch_hash,
                                       ch_hash_many, ch_stream_create, ch_stream_init,
                                       ch_stream_update, ch_stream_finalize, ch_stream_destroy,
                                       ch_verify_many, ch_dedup_create, ch_dedup_insert,
                                       ch_dedup_contains, ch_dedup_size, ch_dedup_destroy};

// ---- chunker_v1 -------------------------------------------------------------------------------
yams_status_t ck_default_config(void*, uint32_t mode, yams_cdc_config_t* out_cfg) {
    if (!out_cfg || (mode != YAMS_CDC_RABIN && mode != YAMS_CDC_STREAMING)) return YAMS_ERR_INVALID_ARG;
    yams_cdc_default_config(out_cfg, mode);
    return YAMS_OK;
}
yams_status_t ck_chunk_data(void*, const uint8_t* data, size_t n, const yams_cdc_config_t* cfg,
                            yams_chunk_ref_t** out_chunks, size_t* out_count) {
    std::lock_guard<std::mutex> lk(g.mu);
    NEED_CTX();
    if (!out_chunks || !out_count || !cfg) return YAMS_ERR_INVALID_ARG;
    *out_chunks = nullptr; *out_count = 0;
    const uint64_t floor = std::max<uint64_t>(1, cfg->min_size);
    size_t cap = n / floor + 2;
    std::vector<uint64_t> off(cap), sz(cap);
    std::vector<char> hex(cap * 65);
    size_t cnt = 0;
    yams_status_t s = yams_cdc_chunk_host(g.ctx, data, n, cfg, off.data(), sz.data(), hex.data(), cap, &cnt);
    if (s != YAMS_OK) return s;
    auto* chunks = static_cast<yams_chunk_ref_t*>(std::calloc(std::max<size_t>(cnt, 1), sizeof(yams_chunk_ref_t)));
    if (!chunks) return YAMS_ERR_INTERNAL;
    for (size_t i = 0; i < cnt; ++i) {
        chunks[i].offset = off[i]; chunks[i].size = sz[i];
        std::memcpy(chunks[i].hash_hex, hex.data() + 65 * i, 65);
    }
    ++g.chunk_calls;
    *out_chunks = chunks; *out_count = cnt;
    return YAMS_OK;
}
void ck_free_chunks(void*, yams_chunk_ref_t* chunks, size_t) { std::free(chunks); }

yams_chunker_v1 g_chunker = {YAMS_IFACE_CHUNKER_V1_VERSION, nullptr, ck_default_config,
                             ck_chunk_data, ck_free_chunks};

} // namespace

extern "C" {

YAMS_PLUGIN_API int yams_plugin_get_abi_version(void) { return YAMS_PLUGIN_ABI_VERSION; }
YAMS_PLUGIN_API const char* yams_plugin_get_name(void) { return "yams_mi355x_accel"; }
YAMS_PLUGIN_API const char* yams_plugin_get_version(void) { return YAMS_ACCEL_VERSION_STRING; }
YAMS_PLUGIN_API const char* yams_plugin_get_manifest_json(void) { return kManifest; }

// host_context is a yams_plugin_host_context_v1* (host_services_v1.h:19-26); unused here.  The
// legacy one-argument form (abi_plugin_loader.cpp:329-357) is tolerated: the second argument is
// never dereferenced.
YAMS_PLUGIN_API int yams_plugin_init(const char* config_json, const void* host_context) {
    (void)host_context;
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.initialised) return YAMS_PLUGIN_OK;
    g.device = parse_device(config_json);
    yams_accel_ctx* ctx = nullptr;
    const yams_status_t s = yams_accel_ctx_create(g.device, nullptr, &ctx);
    if (s != YAMS_OK) {
        g.init_error = (s == YAMS_ERR_UNSUPPORTED) ? "no gfx950 device visible" : "context creation failed";
        return YAMS_PLUGIN_ERR_INIT_FAILED; // the host keeps its built-in CPU backends
    }
    g.ctx = ctx;
    g.initialised = true;
    g.init_error.clear();
    return YAMS_PLUGIN_OK;
}

YAMS_PLUGIN_API void yams_plugin_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ctx) {
        (void)hipSetDevice(g.ctx->device);
        (void)hipStreamSynchronize(g.ctx->stream);
        for (auto& kv : g.corpora) free_corpus(kv.second);
        g.corpora.clear();
        for (auto& kv : g_dedup) yams_dedup_set_destroy(kv.second);
        g_dedup.clear();
        yams_accel_ctx_destroy(g.ctx);
        g.ctx = nullptr;
    }
    g.initialised = false;
}

// Returns a pointer to a static vtable; unknown id or version -> NOT_FOUND, null args -> INVALID
// (plugins/glint/plugin.cpp:338-359, tools/fuzzing/fuzz_abi_test_plugin.c:48-60).
YAMS_PLUGIN_API int yams_plugin_get_interface(const char* iface_id, uint32_t version, void** out_iface) {
    if (!iface_id || !out_iface) return YAMS_PLUGIN_ERR_INVALID;
    *out_iface = nullptr;
    if (std::strcmp(iface_id, YAMS_IFACE_VECTOR_SCAN_V1) == 0) {
        if (version < 1 || version > YAMS_IFACE_VECTOR_SCAN_V1_VERSION) return YAMS_PLUGIN_ERR_NOT_FOUND;
        *out_iface = &g_vector_scan; return YAMS_PLUGIN_OK;
    }
    if (std::strcmp(iface_id, YAMS_IFACE_CONTENT_HASH_V1) == 0) {
        if (version < 1 || version > YAMS_IFACE_CONTENT_HASH_V1_VERSION) return YAMS_PLUGIN_ERR_NOT_FOUND;
        *out_iface = &g_content_hash; return YAMS_PLUGIN_OK;
    }
    if (std::strcmp(iface_id, YAMS_IFACE_CHUNKER_V1) == 0) {
        if (version < 1 || version > YAMS_IFACE_CHUNKER_V1_VERSION) return YAMS_PLUGIN_ERR_NOT_FOUND;
        *out_iface = &g_chunker; return YAMS_PLUGIN_OK;
    }
    return YAMS_PLUGIN_ERR_NOT_FOUND;
}

// malloc'd; the host free()s it (abi_plugin_loader.cpp:481-500).
YAMS_PLUGIN_API int yams_plugin_get_health_json(char** out_json) {
    if (!out_json) return YAMS_PLUGIN_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g.mu);
    std::ostringstream os;
    os << "{\"status\":\"" << (g.ctx ? "ok" : (g.initialised ? "degraded" : "not_initialised"))
       << "\",\"device\":" << g.device << ",\"corpora\":" << g.corpora.size()
       << ",\"searches\":" << g.searches << ",\"hashes\":" << g.hashes
       << ",\"chunk_calls\":" << g.chunk_calls;
    if (!g.init_error.empty()) os << ",\"error\":\"" << g.init_error << "\"";
    os << "}";
    const std::string s = os.str();
    char* buf = static_cast<char*>(std::malloc(s.size() + 1));
    if (!buf) return YAMS_PLUGIN_ERR_INIT_FAILED;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    *out_json = buf;
    return YAMS_PLUGIN_OK;
}

} // extern "C"
