// plugin.cpp — the YAMS plugin surface of libyams_mi355x_accel.so.
//
// Exports the eight entry points of the reference's include/yams/plugins/abi.h:26-34 and serves
// three interface vtables (vector_scan_v1, content_hash_v1, chunker_v1) written to the
// conventions of include/yams/plugins/model_provider_v1.h:44-49.  The host side that would load
// this file is AbiPluginLoader::load / getInterface (src/daemon/resource/abi_plugin_loader.cpp:
// 270-442, 657-681): dlopen(RTLD_LAZY|RTLD_LOCAL), yams_plugin_init(config_json, host_context),
// yams_plugin_get_manifest_json, then yams_plugin_get_interface(id, version, &vtable).
#include <algorithm>
#include <atomic>
#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <set>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

#include "accel_ctx.h"

// The eight entry points and their return codes are declared in include/yams_mi355x_accel.h exactly as
// the reference's include/yams/plugins/abi.h:18-34 declares them (no local restatement here).

namespace {

// No exception may cross the C ABI (model_provider_v1.h:44-49; the reference's own plugins wrap every entry point,
// plugins/onnx/model_provider.cpp:51,94,105): every function pointer of the three vtables is the guarded form of
// its implementation — std::bad_alloc from an absurd batch size, or anything else thrown below, becomes
// YAMS_ERR_INTERNAL (void functions just return).
template <auto Fn> struct Guard;
template <typename R, typename... A, R (*Fn)(A...)> struct Guard<Fn> {
    static R call(A... a) noexcept {
        try { return Fn(a...); }
        catch (...) { if constexpr (!std::is_void_v<R>) return static_cast<R>(YAMS_ERR_INTERNAL); }
    }
};
#define GUARDED(fn) (&Guard<&fn>::call)

// ------------------------------------------------------------------------------------------------
// A device buffer that grows IN PLACE: one virtual range reserved up front, physical chunks mapped
// behind it as rows arrive (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess).
// Appending to a 38 GB mirror maps a few more chunks; nothing is reallocated or copied, pointers
// handed to running searches stay valid.  (Probed on MI355X: scripts/ubench/vmm_probe.hip.)
//
// A virtual address is NEVER reused for a different mapping: scripts/ubench/vmm_stale.hip shows that
// after hipMemUnmap + a new hipMemMap at the same address, kernels read through stale translations
// (wrong data in 1-60 % of the trials, even on the stream that did the copy), while re-filling a range
// that stays mapped is always seen.  So a buffer whose corpus is destroyed is PARKED, mapping intact,
// and handed to the next corpus on that device (park / unpark below); physical memory goes back to the
// driver only when the process exits.
// Parked mappings are adopted BEST-FIT at the next buffer's first ensure() (the smallest one that holds the
// request; a mapping more than 4x larger than the request is left for a corpus that needs it, a fresh range is
// reserved instead).  Physical memory goes back to the driver (a) when a device allocation fails while
// mappings are parked (evict_parked: unmap + RETIRE the range — it stays reserved and is never mapped
// again, so no stale translation can be hit), and (b) at yams_plugin_shutdown, when no kernel can still be
// reading through the translations.  Only address space is spent on retired ranges.
// Falls back to allocate-copy-free growth if the driver refuses the virtual-memory calls.
// ------------------------------------------------------------------------------------------------
#ifdef YAMS_ACCEL_MEASURE
#define GROW_TRACE(what, err) std::fprintf(stderr, "[yams_mi355x_accel] GrowBuf role %d: %s failed (%s); want %zu mapped %zu reserved %zu\n", role, what, hipGetErrorString(err), bytes, mapped, reserved)
#else
#define GROW_TRACE(what, err) ((void)0)
#endif
struct GrowBuf {
    int device = 0;
    int role = 0;               // which array of a shard this is (buffers are parked and reused per role)
    unsigned char* base = nullptr;
    size_t reserved = 0, mapped = 0;
    bool plain = false; // fallback mode: `base` is a hipMalloc allocation of `mapped` bytes
    std::vector<size_t> chunk_end; // end offset of every physical chunk mapped so far

    static constexpr size_t kGran = 2ull << 20;
    template <typename T> T* as() const { return reinterpret_cast<T*>(base); }

    // reserve_bytes: the size of the virtual range if this is the buffer's first use (per role: a fixed
    // fraction of the device's memory, so parked buffers fit every later corpus)
    bool ensure(size_t bytes, size_t reserve_bytes) {
        if (bytes <= mapped) return true;
        (void)hipSetDevice(device);
        if (!plain && !base) adopt(bytes); // a parked mapping of this (device, role) that fits, if there is one
        if (bytes <= mapped) return true;
        if (!plain && !base) {
            size_t want = std::max(reserve_bytes, bytes);
            want = (want + kGran - 1) / kGran * kGran;
            void* va = nullptr;
            if (hipMemAddressReserve(&va, want, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); plain = true; }
            else { base = static_cast<unsigned char*>(va); reserved = want; }
        }
        if (!plain && bytes > reserved) { GROW_TRACE("beyond reservation", hipSuccess); return false; } // (sized to the device's memory)
        if (!plain) {
            // geometric chunks: at least 32 MiB (small side arrays: 2 MiB), at least a quarter of what is mapped
            size_t add = std::max<size_t>(bytes - mapped, std::max<size_t>(reserved >= (1ull << 30) ? 32ull << 20 : kGran, mapped / 4));
            add = (add + kGran - 1) / kGran * kGran;
            if (mapped + add > reserved) add = reserved - mapped;
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = device;
            hipMemGenericAllocationHandle_t h;
            hipError_t e = yams_accel::ya_mem_create(&h, add, &prop);
            if (e != hipSuccess && evict_parked(device)) { // parked mirrors of destroyed corpora give their memory back first
                (void)hipGetLastError();
                if (mapped + (bytes - mapped + kGran - 1) / kGran * kGran <= reserved) add = (bytes - mapped + kGran - 1) / kGran * kGran;
                e = yams_accel::ya_mem_create(&h, add, &prop);
            }
            if (e != hipSuccess) { GROW_TRACE("hipMemCreate", e); (void)hipGetLastError(); return false; }
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemMap(base + mapped, add, 0, h, 0);
            // access is set over the WHOLE mapped range: on a sub-range that starts at a later chunk the call
            // returns "invalid argument" as soon as chunk sizes differ (scripts/ubench/vmm_map.hip)
            const bool was_mapped = e == hipSuccess;
            if (e == hipSuccess) e = hipMemSetAccess(base, mapped + add, &acc, 1);
            if (e != hipSuccess && was_mapped) (void)hipMemUnmap(base + mapped, add); // never touched: safe to unmap
            if (e != hipSuccess) { GROW_TRACE("hipMemMap/SetAccess", e); (void)hipGetLastError(); (void)hipMemRelease(h); return false; }
            (void)hipMemRelease(h); // the mapping keeps the memory alive; it is never unmapped (see above)
            mapped += add;
            chunk_end.push_back(mapped);
            return true;
        }
        // fallback: allocate, copy, free
        const size_t want = std::max(bytes, mapped + mapped / 2);
        void* nd = nullptr;
        if (yams_accel::ya_malloc(&nd, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (mapped && hipMemcpy(nd, base, mapped, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(nd); return false; }
        if (base) (void)hipFree(base);
        base = static_cast<unsigned char*>(nd);
        mapped = want;
        return true;
    }
    // Host -> device copy into [off, off+bytes).  The runtime's copy wants its destination inside ONE
    // allocation, so a range that crosses physical chunks goes as one copy per chunk.
    bool h2d(size_t off, const void* src, size_t bytes, hipStream_t stream) const {
        if (off + bytes > mapped) return false;
        const unsigned char* sp = static_cast<const unsigned char*>(src);
        size_t ci = 0;
        while (bytes) {
            size_t end = off + bytes;
            if (!plain) {
                while (ci < chunk_end.size() && chunk_end[ci] <= off) ++ci;
                if (ci < chunk_end.size()) end = std::min(end, chunk_end[ci]);
            }
            if (yams_accel::staged_h2d(base + off, sp, end - off, stream) != hipSuccess) { (void)hipGetLastError(); return false; } // (pinned ring: link rate from pageable memory)
            sp += end - off; bytes -= end - off; off = end;
        }
        return true;
    }
    void release();            // parks the mapping for the next corpus (or frees a fallback allocation)
    void adopt(size_t bytes);  // takes the best-fitting parked mapping of the same (device, role), if there is one
    void unmap_and_retire();   // gives the physical memory back; the virtual range stays reserved, never mapped again
    static bool evict_parked(int device);   // all parked mappings of a device; true if anything was freed
    static void evict_all_parked();
};

// parked mappings, per (device, role), until a corpus adopts them, memory runs short or the plugin shuts down
std::mutex g_park_mu;
std::map<std::pair<int, int>, std::vector<GrowBuf>> g_parked;

void GrowBuf::release() {
    if (plain) { (void)hipSetDevice(device); if (base) (void)hipFree(base); }
    else if (base) {
        std::lock_guard<std::mutex> lk(g_park_mu);
        g_parked[{device, role}].push_back(*this);
    }
    base = nullptr; reserved = mapped = 0; plain = false; chunk_end.clear();
}
void GrowBuf::adopt(size_t bytes) {
    if (base) return;
    std::lock_guard<std::mutex> lk(g_park_mu);
    auto& v = g_parked[{device, role}];
    if (v.empty()) return;
    // best fit: the smallest parked mapping that holds the request — but not one more than 4x larger (a small
    // corpus must not pin the mirror of a destroyed 100 GB one; that one waits for a corpus of its size).
    // Nothing large enough: the largest one that is not larger than needed, it grows in place.
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].mapped >= bytes && v[i].mapped / 4 <= std::max<size_t>(bytes, 64ull << 20) && (best == v.size() || v[i].mapped < v[best].mapped)) best = i;
    if (best == v.size())
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].mapped < bytes && v[i].reserved >= bytes && (best == v.size() || v[i].mapped > v[best].mapped)) best = i;
    if (best == v.size()) return;
    base = v[best].base; reserved = v[best].reserved; mapped = v[best].mapped; chunk_end = std::move(v[best].chunk_end);
    v.erase(v.begin() + static_cast<std::ptrdiff_t>(best));
}
void GrowBuf::unmap_and_retire() {
    if (plain || !base) return;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize(); // nothing may still read through these translations
    size_t off = 0;
    for (size_t end : chunk_end) { if (hipMemUnmap(base + off, end - off) != hipSuccess) (void)hipGetLastError(); off = end; }
    // the range itself is NOT freed: a later reservation could land on it, and a mapping at an address that
    // was mapped before is read through stale translations on this driver (scripts/ubench/vmm_stale.hip)
    base = nullptr; reserved = mapped = 0; chunk_end.clear();
}
bool GrowBuf::evict_parked(int device) {
    std::vector<GrowBuf> victims;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        for (auto& kv : g_parked)
            if (kv.first.first == device) { for (auto& b : kv.second) victims.push_back(std::move(b)); kv.second.clear(); }
    }
    size_t freed = 0;
    for (auto& b : victims) { freed += b.mapped; b.unmap_and_retire(); }
    return freed != 0;
}
void GrowBuf::evict_all_parked() {
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        for (auto& kv : g_parked) if (!kv.second.empty()) devs.push_back(kv.first.first);
    }
    for (int d : devs) (void)evict_parked(d);
}

// One shard of a corpus: the rows dealt to one device, plus their filter shadows.
struct ShardStore {
    int device = 0;
    uint64_t n_rows = 0;
    GrowBuf rows, bf16, nsq, i8, i8meta, tie, inv;
    bool has_tie = false;
    void release() { rows.release(); bf16.release(); nsq.release(); i8.release(); i8meta.release(); tie.release(); inv.release(); n_rows = 0; has_tie = false; }
    // virtual range per array if freshly reserved: a fixed share of the device's memory per role
    static size_t share(size_t dev_total, int role) {
        switch (role) {
            case 0: return dev_total;            // rows (fp32)
            case 1: return dev_total / 2;        // bf16 shadow
            case 2: return dev_total / 4;        // int8 shadow
            case 3: return dev_total / 16;       // squared norms
            case 4: return dev_total / 256;      // int8 block scales
            default: return dev_total / 16;      // tie ranks, inverse, corpus-wide ranking
        }
    }
};

// The host's product-quantiser index of a corpus (SimeonPqIndexState, sqlite_vec_backend.cpp:48-62), on the corpus's device:
// codes, the rank of every code's tie-break key, and the mirror row behind every key index.
struct PqIndex {
    uint8_t* codes = nullptr; uint32_t* tie_rank = nullptr; uint32_t* key_row = nullptr;
    uint64_t n = 0; uint32_t m = 0; int device = 0;
    void release() {
        if (codes || tie_rank || key_row) (void)hipSetDevice(device);
        if (codes) (void)hipFree(codes);
        if (tie_rank) (void)hipFree(tie_rank);
        if (key_row) (void)hipFree(key_row);
        codes = nullptr; tie_rank = key_row = nullptr; n = 0; m = 0;
    }
};

struct Corpus {
    uint32_t dim = 0;
    uint64_t n_rows = 0;
    int i8_flags = -1;               // layout of the int8 shadow (YAMS_SCAN_I8_*), decided at the first append; -1: not yet
    uint64_t i8_decided_rows = 0;    // rows the "auto" decision looked at (one taken from fewer than 4096 rows is taken again once they are there)
    PqIndex pq;                      // (version 2 of the vtable: pq_index_set / search_pq)
    std::vector<ShardStore> sh;      // one per plugin device
    GrowBuf rank_of_row;             // device 0: the corpus-wide chunk_id ranking (cross-shard ties)
    bool has_ranks = false;
    std::shared_mutex mu;            // searches share, mutations exclude (vector_database.cpp:539,618)
};

// rows of a stripe when a corpus is dealt to several devices (config "stripe_rows", a multiple of 64;
// fixed at init: it is part of every corpus's row -> shard map)
uint32_t kStripeRows = 65536;

// A pool of N identical resources handed out to concurrent calls.
template <typename T> class Pool {
public:
    void add(T v) { std::lock_guard<std::mutex> lk(mu_); free_.push_back(v); all_.push_back(v); }
    T acquire() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !free_.empty(); });
        T v = free_.back(); free_.pop_back();
        return v;
    }
    void release(T v) { { std::lock_guard<std::mutex> lk(mu_); free_.push_back(v); } cv_.notify_one(); }
    std::vector<T> drain() { std::lock_guard<std::mutex> lk(mu_); auto a = all_; all_.clear(); free_.clear(); return a; }
    size_t size() const { return all_.size(); }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<T> free_, all_;
};
template <typename T> struct Lease {
    Pool<T>& pool; T v;
    explicit Lease(Pool<T>& p) : pool(p), v(p.acquire()) {}
    ~Lease() { pool.release(v); }
};

struct PluginState {
    std::shared_mutex mu;            // init / shutdown exclude everything else; calls share
    bool initialised = false;
    std::vector<int> devices;        // config "devices": [..] (or "device": n); rows are striped over them
    bool want_bf16 = true, want_i8 = true;
    int i8_layout = 0;               // configuration "i8_layout": 0 auto (measured at a corpus' first append), 1 plain, 2 rotated
    std::string init_error;
    // searches: ONE sharded handle over the plugin's devices (one RCCL communicator when there are several),
    // "search_slots" lanes = concurrent searches in flight, each on its own contexts / streams / worker threads;
    // the lanes of one device share a sweep gate inside the handle (sharded_api.cpp)
    yams_scan_sharded* sharded = nullptr;
    uint32_t search_slots = 0;
    uint32_t l2_acc = YAMS_SCAN_FLAG_L2_ACC_F64;   // config "l2_accumulate": "f64" | "f32" | "f32x8" | "f32x16"
    // the last corpus_append: bytes, ms spent mapping device memory / copying / building shadows (written under upload_mu,
    // read by the health call without it: atomics), and the slowest append so far with its split
    std::atomic<uint64_t> append_bytes{0}, appends{0};
    std::atomic<double> append_map_ms{0}, append_copy_ms{0}, append_shadow_ms{0};
    std::atomic<double> slow_append_ms{0}, slow_map_ms{0}, slow_copy_ms{0}, slow_shadow_ms{0};
    std::atomic<uint64_t> slow_append_bytes{0}, exhausted_appends{0};
    Pool<yams_accel_ctx*> work_ctx;          // hashing / chunking contexts on devices[0]
    std::vector<yams_accel_ctx*> upload_ctx; // one per device, used under a corpus's exclusive lock
    std::mutex upload_mu;                    // (upload contexts are shared by all corpora)
    std::mutex corpora_mu;
    std::map<uint64_t, std::shared_ptr<Corpus>> corpora;
    uint64_t next_id = 1;
    std::atomic<uint64_t> searches{0}, hashes{0}, chunk_calls{0}, refused_chains{0}, deferred_chains{0};
};
PluginState g;

const char kManifest[] =
    "{\"name\":\"yams_mi355x_accel\",\"version\":\"" YAMS_ACCEL_VERSION_STRING "\",\"abi\":1,"
    "\"description\":\"MI355X (gfx950) exact vector scan, SHA-256 and content-defined chunking\","
    "\"interfaces\":[{\"id\":\"vector_scan_v1\",\"version\":1},"
    "{\"id\":\"content_hash_v1\",\"version\":1},{\"id\":\"chunker_v1\",\"version\":3}]}";

// ---- the plugin's configuration: a strict reader of ONE flat JSON object ----------------------------------------------------
// {"key": "string" | integer | [integers] | true | false | null | {...} | [...]}: keys the plugin does not know are skipped
// (whatever their value, nested or not); a key it knows with a value of the wrong TYPE, an enumerated value it does not
// list, or text that is not a JSON object fails yams_plugin_init — a host's typo must not silently serve another arithmetic
// (round 5 read its keys with strstr: {"shadows":"none","note":"both"} enabled both shadows).
struct Config {
    std::map<std::string, std::string> strings;
    std::map<std::string, long> ints;
    std::map<std::string, std::vector<long>> int_lists;
    std::set<std::string> other;       // keys present with a value of another type (booleans, null, objects, nested arrays, floats)
    std::string error;                 // non-empty: the text did not parse

    static void ws(const char*& p) { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') ++p; }
    static bool str(const char*& p, std::string& out) {
        if (*p != '"') return false;
        out.clear();
        for (++p; *p && *p != '"'; ++p) {
            if (*p == '\\') { ++p; if (!*p) return false; out.push_back(*p == 'n' ? '\n' : (*p == 't' ? '\t' : *p)); }
            else out.push_back(*p);
        }
        if (*p != '"') return false;
        ++p;
        return true;
    }
    static bool integer(const char*& p, long& v) {
        char* e = nullptr;
        v = std::strtol(p, &e, 10);
        if (e == p || *e == '.' || *e == 'e' || *e == 'E') return false;
        p = e;
        return true;
    }
    static bool skip(const char*& p, int depth = 0) {   // any JSON value
        ws(p);
        if (depth > 32) return false;
        std::string t;
        if (*p == '"') return str(p, t);
        if (*p == '{' || *p == '[') {
            const char close = *p == '{' ? '}' : ']';
            const bool object = *p == '{';
            ++p; ws(p);
            if (*p == close) { ++p; return true; }
            for (;;) {
                ws(p);
                if (object) { if (!str(p, t)) return false; ws(p); if (*p++ != ':') return false; }
                if (!skip(p, depth + 1)) return false;
                ws(p);
                if (*p == ',') { ++p; continue; }
                if (*p == close) { ++p; return true; }
                return false;
            }
        }
        const char* b = p;
        while (*p && (std::isalnum(static_cast<unsigned char>(*p)) || *p == '-' || *p == '+' || *p == '.')) ++p;
        return p != b;
    }
    explicit Config(const char* json) {
        if (!json) return;
        const char* p = json;
        ws(p);
        if (!*p) return;                                // "" = no configuration
        if (*p != '{') { error = "the configuration is not a JSON object"; return; }
        ++p; ws(p);
        if (*p == '}') { ++p; ws(p); if (*p) error = "text after the configuration object"; return; }
        for (;;) {
            std::string key;
            ws(p);
            if (!str(p, key)) { error = "expected a key"; return; }
            ws(p);
            if (*p++ != ':') { error = "expected ':' after \"" + key + "\""; return; }
            ws(p);
            if (*p == '"') { std::string v; if (!str(p, v)) { error = "unterminated string for \"" + key + "\""; return; } strings[key] = v; }
            else if (*p == '-' || std::isdigit(static_cast<unsigned char>(*p))) {
                const char* q = p; long v;
                if (integer(q, v)) { ints[key] = v; p = q; }
                else { if (!skip(p)) { error = "bad number for \"" + key + "\""; return; } other.insert(key); }
            } else if (*p == '[') {
                const char* q = p + 1; std::vector<long> lst; bool ok = true;
                ws(q);
                if (*q == ']') ++q;
                else for (;;) {
                    long v; ws(q);
                    if (!integer(q, v)) { ok = false; break; }
                    lst.push_back(v); ws(q);
                    if (*q == ',') { ++q; continue; }
                    if (*q == ']') { ++q; break; }
                    ok = false; break;
                }
                if (ok) { int_lists[key] = lst; p = q; }
                else { if (!skip(p)) { error = "bad array for \"" + key + "\""; return; } other.insert(key); }
            } else { if (!skip(p)) { error = "bad value for \"" + key + "\""; return; } other.insert(key); }
            ws(p);
            if (*p == ',') { ++p; continue; }
            if (*p == '}') { ++p; break; }
            error = "expected ',' or '}' after \"" + key + "\""; return;
        }
        ws(p);
        if (*p) error = "text after the configuration object";
    }
    bool has(const std::string& k) const { return strings.count(k) || ints.count(k) || int_lists.count(k) || other.count(k); }
    // typed reads: false (with `error` set) when the key is there with another type
    bool get_int(const std::string& k, long dflt, long& out) {
        out = dflt;
        if (!has(k)) return true;
        const auto it = ints.find(k);
        if (it == ints.end()) { error = "\"" + k + "\" must be an integer"; return false; }
        out = it->second;
        return true;
    }
    bool get_string(const std::string& k, std::string& out, bool& present) {
        present = false;
        if (!has(k)) return true;
        const auto it = strings.find(k);
        if (it == strings.end()) { error = "\"" + k + "\" must be a string"; return false; }
        out = it->second; present = true;
        return true;
    }
    // an enumerated string: index into `allowed`, dflt when absent; false on any other value
    bool get_choice(const std::string& k, std::initializer_list<const char*> allowed, int dflt, int& out) {
        out = dflt;
        std::string v; bool present;
        if (!get_string(k, v, present)) return false;
        if (!present) return true;
        int i = 0;
        for (const char* a : allowed) { if (v == a) { out = i; return true; } ++i; }
        error = "\"" + k + "\": \"" + v + "\" is not one of";
        for (const char* a : allowed) error += std::string(" \"") + a + "\"";
        return false;
    }
};

std::shared_ptr<Corpus> find_corpus(uint64_t id) {
    std::lock_guard<std::mutex> lk(g.corpora_mu);
    auto it = g.corpora.find(id);
    return it == g.corpora.end() ? nullptr : it->second;
}

#define NEED_INIT() std::shared_lock<std::shared_mutex> init_lk__(g.mu); do { if (!g.initialised) return YAMS_ERR_UNSUPPORTED; } while (0)

// global row -> (shard, local row) under the stripe dealing
inline uint32_t shard_of(uint64_t row, uint32_t n_sh) { return n_sh == 1 ? 0u : static_cast<uint32_t>((row / kStripeRows) % n_sh); }
inline uint64_t local_of(uint64_t row, uint32_t n_sh) {
    return n_sh == 1 ? row : (row / kStripeRows / n_sh) * kStripeRows + row % kStripeRows;
}
// rows of a corpus of n rows that live on shard i
inline uint64_t shard_rows(uint64_t n, uint32_t n_sh, uint32_t i) {
    if (n_sh == 1) return n;
    const uint64_t full = n / kStripeRows, rem = n % kStripeRows;
    uint64_t r = (full / n_sh) * kStripeRows + ((full % n_sh) > i ? kStripeRows : 0);
    if (full % n_sh == i) r += rem;
    return r;
}

// ---- vector_scan_v1 ---------------------------------------------------------------------------
yams_status_t vs_corpus_create(void*, uint32_t dim, uint64_t* out_id) {
    NEED_INIT();
    if (!out_id || dim == 0) return YAMS_ERR_INVALID_ARG;
    auto c = std::make_shared<Corpus>();
    c->dim = dim;
    c->sh.resize(g.devices.size());
    for (size_t i = 0; i < g.devices.size(); ++i) {
        auto& s = c->sh[i];
        s.device = g.devices[i];
        int role = 0;
        for (GrowBuf* b : {&s.rows, &s.bf16, &s.i8, &s.nsq, &s.i8meta, &s.tie, &s.inv}) { b->device = s.device; b->role = role++; }
    }
    c->rank_of_row.device = g.devices[0]; c->rank_of_row.role = 7; // (parked mappings are adopted at the first ensure(), best fit)
    std::lock_guard<std::mutex> lk(g.corpora_mu);
    *out_id = g.next_id++;
    g.corpora[*out_id] = c;
    return YAMS_OK;
}

// Appends rows: they take the next global ids, are dealt to the devices in stripes, copied to the end of
// each shard's growing mirror, and the shadows of the touched local ranges are (re)built.
// measure builds say where an internal error came from
// device memory behind a mirror could not be had (out of memory, or beyond the mirror's share of the address space):
// the corpus is left exactly as it was — same rows, same shadows, searches go on — and the caller can act on it
// (ErrorCode::ResourceExhausted, include/yams/core/types.h:49 of the reference)
inline yams_status_t exhausted(const char* where) {
#ifdef YAMS_ACCEL_MEASURE
    std::fprintf(stderr, "[yams_mi355x_accel] device memory exhausted at %s\n", where);
#else
    (void)where;
#endif
    return YAMS_ERR_RESOURCE_EXHAUSTED;
}
inline yams_status_t internal_error(const char* where) {
#ifdef YAMS_ACCEL_MEASURE
    std::fprintf(stderr, "[yams_mi355x_accel] internal error at %s (hip: %s)\n", where, hipGetErrorString(hipGetLastError()));
#else
    (void)where;
#endif
    return YAMS_ERR_INTERNAL;
}

yams_status_t vs_corpus_append(void*, uint64_t id, const float* rows, uint64_t n_rows) {
    NEED_INIT();
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    if (n_rows == 0) return YAMS_OK;
    if (!rows) return YAMS_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> lk(c->mu);
    std::lock_guard<std::mutex> up(g.upload_mu);
    const uint32_t n_sh = static_cast<uint32_t>(c->sh.size());
    const uint64_t n0 = c->n_rows, n1 = n0 + n_rows;
    if (shard_rows(n1, n_sh, 0) >= (1ull << 32)) return YAMS_ERR_UNSUPPORTED;
    const size_t rb = static_cast<size_t>(c->dim) * 4;
    const bool bf16 = g.want_bf16 && (c->dim & 3u) == 0;
    const bool i8 = g.want_i8 && (c->dim & 63u) == 0 && c->dim >= 256;
    size_t dev_total = 0, dev_free = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < n_sh; ++i) {
        ShardStore& s = c->sh[i];
        const uint64_t old = s.n_rows, now = shard_rows(n1, n_sh, i);
        if (now == old) continue;
        (void)hipSetDevice(s.device);
        if (hipMemGetInfo(&dev_free, &dev_total) != hipSuccess || dev_total == 0) {
            (void)hipGetLastError();
            dev_total = 288ull << 30; // the reservation is address space only: size it for the part this library is written for
        }
        if (!s.rows.ensure(now * rb, ShardStore::share(dev_total, 0))) { ++g.exhausted_appends; return exhausted("append:1"); }
        if (bf16 && (!s.bf16.ensure(now * rb / 2, ShardStore::share(dev_total, 1)) || !s.nsq.ensure(now * 4, ShardStore::share(dev_total, 3)))) { ++g.exhausted_appends; return exhausted("append:2"); }
        if (i8 && (!s.i8.ensure((now + 63) / 64 * 64 * rb / 4 /* whole 64-row blocks: the shadow is stored blocked */, ShardStore::share(dev_total, 2)) || !s.i8meta.ensure(((now + 63) / 64) * 8, ShardStore::share(dev_total, 4)))) { ++g.exhausted_appends; return exhausted("append:3"); }
    }
    const auto t_mapped = std::chrono::steady_clock::now();
    // copy: runs of consecutive global rows inside one stripe are consecutive local rows
    for (uint64_t r = n0; r < n1;) {
        const uint64_t run = std::min<uint64_t>(n1 - r, n_sh == 1 ? n1 - r : kStripeRows - r % kStripeRows);
        ShardStore& s = c->sh[shard_of(r, n_sh)];
        yams_accel_ctx* uc = g.upload_ctx[shard_of(r, n_sh)];
        (void)hipSetDevice(s.device);
        if (!s.rows.h2d(local_of(r, n_sh) * rb, rows + (r - n0) * c->dim, run * rb, uc->stream)) return internal_error("append:4");
        r += run;
    }
    for (uint32_t i = 0; i < n_sh; ++i) if (yams_accel_ctx_synchronize(g.upload_ctx[i]) != YAMS_OK) return internal_error("append:4b");
    const auto t_copied = std::chrono::steady_clock::now();
    bool relayout = false;          // the shadow is rebuilt from row 0 (the measured layout changed once enough rows were there)
    if (i8 && (c->i8_flags < 0 || (g.i8_layout == 0 && c->i8_decided_rows < 4096 && n1 >= 4096))) {
        // the layout of this corpus' int8 shadow: the configured one, or whichever quantises the rows better — measured at the
        // first append and, if that one brought fewer than 4096 rows (a host that inserts a few vectors, then searches), once
        // more when 4096 are there
        const int before = c->i8_flags;
        c->i8_flags = 0;
        c->i8_decided_rows = n1;
        if (g.i8_layout == 2) {
            if (c->dim <= 4096) c->i8_flags = static_cast<int>(YAMS_SCAN_I8_ROTATED);   // (the rotated layout exists for 256 <= dim <= 4096)
        } else if (g.i8_layout == 0) {
            for (uint32_t i = 0; i < n_sh; ++i) {
                const uint64_t now = shard_rows(n1, n_sh, i);
                if (!now) continue;
                uint32_t fl = 0;
                (void)hipSetDevice(c->sh[i].device);
                if (yams_scan_choose_i8_layout_device(g.upload_ctx[i], c->sh[i].rows.as<float>(), now, c->dim, &fl, nullptr, nullptr) != YAMS_OK)
                    return internal_error("append:4c");
                c->i8_flags = static_cast<int>(fl);
                break;
            }
        }
        relayout = before >= 0 && before != c->i8_flags;
    }
    for (uint32_t i = 0; i < n_sh; ++i) {
        ShardStore& s = c->sh[i];
        const uint64_t old = s.n_rows, now = shard_rows(n1, n_sh, i);
        yams_accel_ctx* uc = g.upload_ctx[i];
        if (relayout && i8 && now) {     // every block of this shard again, in the layout the rows turned out to want
            if (yams_scan_build_shadow_i8_layout_device(uc, s.rows.as<float>(), 0, old, c->dim, static_cast<uint32_t>(c->i8_flags),
                                                        s.i8.as<int8_t>(), s.i8meta.as<float>(), nullptr) != YAMS_OK)
                return internal_error("append:6b");
        }
        if (now != old) {
            if (bf16 && yams_scan_build_shadow_device(uc, s.rows.as<float>() + old * c->dim, now - old, c->dim,
                                                      s.bf16.as<uint16_t>() + old * c->dim, s.nsq.as<float>() + old) != YAMS_OK)
                return internal_error("append:5");
            if (i8 && yams_scan_build_shadow_i8_layout_device(uc, s.rows.as<float>(), old, now - old, c->dim, static_cast<uint32_t>(c->i8_flags),
                                                              s.i8.as<int8_t>(), s.i8meta.as<float>(), nullptr) != YAMS_OK)
                return internal_error("append:6");
        }
        if (yams_accel_ctx_synchronize(uc) != YAMS_OK) return internal_error("append:7");
        s.n_rows = now;
        s.has_tie = false; // appended rows invalidate a previously supplied chunk_id ranking
    }
    c->n_rows = n1;
    c->has_ranks = false;
    {   // where the call's time went (health JSON, "last_append"): fresh device memory behind the mirrors, the copy, the shadows
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        g.append_bytes = n_rows * rb; g.append_map_ms = ms(t_begin, t_mapped); g.append_copy_ms = ms(t_mapped, t_copied); g.append_shadow_ms = ms(t_copied, t_end);
        ++g.appends;
        if (ms(t_begin, t_end) > g.slow_append_ms.load()) {   // an outlier among many appends says where its time went
            g.slow_append_ms = ms(t_begin, t_end); g.slow_map_ms = ms(t_begin, t_mapped); g.slow_copy_ms = ms(t_mapped, t_copied);
            g.slow_shadow_ms = ms(t_copied, t_end); g.slow_append_bytes = n_rows * rb;
        }
    }
    return YAMS_OK;
}

yams_status_t vs_corpus_set_tie_ranks(void*, uint64_t id, const uint32_t* ranks, uint64_t n_rows) {
    NEED_INIT();
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    std::unique_lock<std::shared_mutex> lk(c->mu);
    if (n_rows != c->n_rows || (!ranks && n_rows)) return YAMS_ERR_INVALID_ARG;
    {
        std::vector<uint8_t> seen(n_rows, 0);
        for (uint64_t r = 0; r < n_rows; ++r) {
            if (ranks[r] >= n_rows || seen[ranks[r]]) return YAMS_ERR_INVALID_ARG; // not a permutation
            seen[ranks[r]] = 1;
        }
    }
    if (n_rows == 0) return YAMS_OK;
    std::lock_guard<std::mutex> up(g.upload_mu);
    const uint32_t n_sh = static_cast<uint32_t>(c->sh.size());
    for (uint32_t i = 0; i < n_sh; ++i) {
        ShardStore& s = c->sh[i];
        const uint64_t nl = s.n_rows;
        if (nl == 0) continue;
        // local tie ranks: a permutation of 0..nl-1 that preserves the global order (the scan sorts ties
        // by it inside the shard; the merge compares the global ranks through rank_of_row)
        std::vector<uint32_t> glob(nl), order(nl), lrank(nl), linv(nl);
        for (uint64_t l = 0; l < nl; ++l)
            glob[l] = ranks[n_sh == 1 ? l : ((l / kStripeRows) * n_sh + i) * kStripeRows + l % kStripeRows];
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return glob[x] < glob[y]; });
        for (uint32_t p = 0; p < nl; ++p) { lrank[order[p]] = p; linv[p] = order[p]; }
        if (!s.tie.ensure(nl * 4, s.rows.reserved / 16) || !s.inv.ensure(nl * 4, s.rows.reserved / 16)) return YAMS_ERR_RESOURCE_EXHAUSTED;
        (void)hipSetDevice(s.device);
        if (!s.tie.h2d(0, lrank.data(), nl * 4, g.upload_ctx[i]->stream) || !s.inv.h2d(0, linv.data(), nl * 4, g.upload_ctx[i]->stream) ||
            yams_accel_ctx_synchronize(g.upload_ctx[i]) != YAMS_OK) return YAMS_ERR_INTERNAL;
        s.has_tie = true;
    }
    if (n_sh > 1) {
        (void)hipSetDevice(c->rank_of_row.device);
        if (!c->rank_of_row.ensure(n_rows * 4, c->sh[0].rows.reserved / 16) ||
            !c->rank_of_row.h2d(0, ranks, n_rows * 4, g.upload_ctx[0]->stream) ||
            yams_accel_ctx_synchronize(g.upload_ctx[0]) != YAMS_OK) return YAMS_ERR_INTERNAL;
    }
    c->has_ranks = true;
    return YAMS_OK;
}

void release_corpus(Corpus& c) {
    c.pq.release();
    for (auto& s : c.sh) s.release();
    c.rank_of_row.release();
    c.n_rows = 0; c.has_ranks = false; c.i8_flags = -1; c.i8_decided_rows = 0;
}

// The rows are gone, the mirror's memory stays mapped for the re-upload that usually follows (compaction).
yams_status_t vs_corpus_clear(void*, uint64_t id) {
    NEED_INIT();
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    std::unique_lock<std::shared_mutex> lk(c->mu);
    for (auto& s : c->sh) { s.n_rows = 0; s.has_tie = false; }
    c->n_rows = 0; c->has_ranks = false; c->i8_flags = -1; c->i8_decided_rows = 0;
    return YAMS_OK;
}

yams_status_t vs_corpus_destroy(void*, uint64_t id) {
    NEED_INIT();
    std::shared_ptr<Corpus> c;
    {
        std::lock_guard<std::mutex> lk(g.corpora_mu);
        auto it = g.corpora.find(id);
        if (it == g.corpora.end()) return YAMS_ERR_NOT_FOUND;
        c = it->second;
        g.corpora.erase(it);
    }
    std::unique_lock<std::shared_mutex> lk(c->mu);
    release_corpus(*c);
    return YAMS_OK;
}

yams_status_t vs_corpus_size(void*, uint64_t id, uint64_t* out_rows, uint32_t* out_dim) {
    NEED_INIT();
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    std::shared_lock<std::shared_mutex> lk(c->mu);
    if (out_rows) *out_rows = c->n_rows;
    if (out_dim) *out_dim = c->dim;
    return YAMS_OK;
}

yams_status_t vs_search_batch_ex(void*, uint64_t id, const float* queries, uint32_t nq, uint32_t dim,
                                 uint32_t k, float threshold, uint32_t metric, uint32_t flags,
                                 const uint32_t* row_mask_host, yams_scan_hit_t** out_hits,
                                 uint32_t** out_counts, yams_scan_diag_t* out_diag) {
    NEED_INIT();
    if (!out_hits || !out_counts) return YAMS_ERR_INVALID_ARG;
    *out_hits = nullptr; *out_counts = nullptr;
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    // dimension mismatch -> InvalidArgument (vector_database.cpp:545-550, 626-633)
    if (dim != c->dim) return YAMS_ERR_INVALID_ARG;
    if (nq && !queries) return YAMS_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> lk(c->mu);      // concurrent searches share the corpus
    struct LaneLease {                                  // ... and each runs on its own lane of the sharded handle
        uint32_t lane = 0; bool held = false;
        ~LaneLease() { if (held) yams_scan_sharded_lane_release(g.sharded, lane); }
    } slot;
    if (yams_scan_sharded_lane_acquire(g.sharded, 1, &slot.lane) != YAMS_OK) return YAMS_ERR_INTERNAL;
    slot.held = true;
    const uint32_t n_sh = static_cast<uint32_t>(c->sh.size());
    std::vector<yams_scan_corpus_t> views(n_sh);
    for (uint32_t i = 0; i < n_sh; ++i) {
        const ShardStore& s = c->sh[i];
        yams_scan_corpus_t& v = views[i];
        std::memset(&v, 0, sizeof v);
        v.rows = s.rows.as<float>(); v.n_rows = s.n_rows; v.dim = c->dim;
        if (s.has_tie) { v.tie_rank = s.tie.as<uint32_t>(); v.rank_row = s.inv.as<uint32_t>(); }
        if (g.want_bf16 && (c->dim & 3u) == 0 && s.n_rows) { v.rows_bf16 = s.bf16.as<uint16_t>(); v.rows_nsq = s.nsq.as<float>(); }
        if (g.want_i8 && (c->dim & 63u) == 0 && c->dim >= 256 && s.n_rows) { v.rows_i8 = s.i8.as<int8_t>(); v.rows_i8_meta = s.i8meta.as<float>(); v.i8_flags = c->i8_flags > 0 ? static_cast<uint32_t>(c->i8_flags) : 0u; }
        if (n_sh > 1) { v.stripe_rows = kStripeRows; v.n_stripes = n_sh; v.stripe_index = i; }
        if (row_mask_host && s.n_rows) { // document_hash / candidate_hashes restriction (:4137-4175), dealt like the rows
            const size_t words = (s.n_rows + 31) / 32;
            std::vector<uint32_t> local(words, 0u);
            uint64_t bits = 0;
            for (size_t w = 0; w < words; ++w) {
                const uint64_t l0 = static_cast<uint64_t>(w) * 32;  // kStripeRows % 32 == 0: a local word is a global word
                const uint64_t g0 = n_sh == 1 ? l0 : ((l0 / kStripeRows) * n_sh + i) * kStripeRows + l0 % kStripeRows;
                uint32_t m = row_mask_host[g0 >> 5];
                const uint64_t left = s.n_rows - l0;
                if (left < 32) m &= (1u << left) - 1u;
                local[w] = m;
                bits += static_cast<uint64_t>(__builtin_popcount(m));
            }
            yams_accel_ctx* sc = yams_scan_sharded_lane_ctx(g.sharded, i, slot.lane);
            (void)hipSetDevice(s.device);
            uint32_t* d_mask = nullptr;
            if (const yams_status_t ws = yams_accel::ws_get(sc, "plugin_row_mask", words * 4, (void**)&d_mask); ws != YAMS_OK) return ws;
            if (yams_accel_upload(sc, d_mask, local.data(), words * 4) != YAMS_OK) return YAMS_ERR_INTERNAL;
            v.row_mask = d_mask; v.row_mask_count = bits;
        }
    }
    // only semantic flags cross the vtable; filter selection stays with the library
    yams_scan_params_t prm{k, threshold, metric, flags & (YAMS_SCAN_FLAG_RECORD_PATH | YAMS_SCAN_FLAG_FORCE_EXACT | YAMS_SCAN_FLAG_DEFER_THRESHOLD |
                                                          YAMS_SCAN_FLAG_L2_ACC_MASK | YAMS_SCAN_FLAG_L2_ACC_FUSED)};
    // vec0's distance arithmetic: the call's own choice (any L2_ACC bit, or L2_ACC_EXPLICIT for a deliberate F64), else
    // the plugin's ("l2_accumulate" in the init config)
    if (metric == YAMS_SCAN_L2 && !(flags & (YAMS_SCAN_FLAG_L2_ACC_MASK | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT))) prm.flags |= g.l2_acc;
    const size_t slots = static_cast<size_t>(nq) * std::max<uint32_t>(k, 1);
    std::vector<float> scores(slots), dist(slots);
    std::vector<int64_t> rows(slots);
    auto* counts = static_cast<uint32_t*>(std::calloc(std::max<uint32_t>(nq, 1), sizeof(uint32_t)));
    auto* hits = static_cast<yams_scan_hit_t*>(std::calloc(std::max<size_t>(slots, 1), sizeof(yams_scan_hit_t)));
    if (!counts || !hits) { std::free(counts); std::free(hits); return YAMS_ERR_INTERNAL; }
    yams_status_t s = yams_scan_sharded_submit(g.sharded, slot.lane, views.data(), queries, nq, &prm,
                                               c->has_ranks && n_sh > 1 ? c->rank_of_row.as<uint32_t>() : nullptr, 0,
                                               out_diag ? YAMS_SHARDED_SUBMIT_DIAG : 0u);
    if (s == YAMS_OK) {
        slot.held = false; // wait() frees the lane
        s = yams_scan_sharded_wait(g.sharded, slot.lane, scores.data(), rows.data(), counts, dist.data(), out_diag);
    }
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

// ---- version 2: the product-quantised engine over the mirror (yams_scan_pq_topk_device) ------------------------------------
yams_status_t vs_pq_index_set(void*, uint64_t id, const uint8_t* codes, uint64_t n_codes, uint32_t m, const uint64_t* tie_keys,
                              const uint32_t* row_of_index) {
    NEED_INIT();
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    std::unique_lock<std::shared_mutex> lk(c->mu);
    if (c->sh.size() != 1) return YAMS_ERR_UNSUPPORTED;      // (a PQ index over a striped corpus: not built)
    c->pq.release();
    if (n_codes == 0) return YAMS_OK;
    if (!codes || m == 0 || m > 128 || n_codes >= (1ull << 32)) return YAMS_ERR_INVALID_ARG;
    // rank of every tie-break key (ascending key, equal keys by index: the comparator of :3985-3990) and the mirror row
    // behind every key index — once per index build, on the host
    std::vector<uint32_t> order(n_codes), rank(n_codes), key_row(n_codes);
    std::iota(order.begin(), order.end(), 0u);
    if (tie_keys)
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tie_keys[a] != tie_keys[b] ? tie_keys[a] < tie_keys[b] : a < b; });
    for (uint64_t r = 0; r < n_codes; ++r) { rank[order[r]] = static_cast<uint32_t>(r); key_row[r] = row_of_index ? row_of_index[order[r]] : order[r]; }
    PqIndex p; p.device = c->sh[0].device; p.n = n_codes; p.m = m;
    (void)hipSetDevice(p.device);
    Lease<yams_accel_ctx*> w(g.work_ctx);
    const size_t code_bytes = (static_cast<size_t>(n_codes) * m + 15) & ~static_cast<size_t>(15);
    bool ok = yams_accel::ya_malloc(reinterpret_cast<void**>(&p.codes), code_bytes) == hipSuccess &&
              yams_accel::ya_malloc(reinterpret_cast<void**>(&p.tie_rank), n_codes * 4) == hipSuccess &&
              yams_accel::ya_malloc(reinterpret_cast<void**>(&p.key_row), n_codes * 4) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); p.release(); return YAMS_ERR_RESOURCE_EXHAUSTED; }
    ok = yams_accel_upload(w.v, p.codes, codes, static_cast<size_t>(n_codes) * m) == YAMS_OK &&
         yams_accel_upload(w.v, p.tie_rank, rank.data(), n_codes * 4) == YAMS_OK &&
         yams_accel_upload(w.v, p.key_row, key_row.data(), n_codes * 4) == YAMS_OK;
    if (!ok) { p.release(); return YAMS_ERR_INTERNAL; }
    c->pq = p;
    return YAMS_OK;
}

yams_status_t vs_search_pq(void*, uint64_t id, const float* queries, const float* luts, uint32_t nq, uint32_t dim, uint32_t k,
                           float threshold, uint32_t rerank_factor, uint32_t flags, const uint32_t* candidates, uint64_t n_candidates,
                           yams_scan_hit_t** out_hits, uint32_t** out_counts, yams_scan_diag_t* out_diag) {
    NEED_INIT();
    if (!out_hits || !out_counts) return YAMS_ERR_INVALID_ARG;
    *out_hits = nullptr; *out_counts = nullptr;
    if (k > YAMS_SCAN_MAX_K) return YAMS_ERR_UNSUPPORTED;
    auto c = find_corpus(id);
    if (!c) return YAMS_ERR_NOT_FOUND;
    if (dim != c->dim) return YAMS_ERR_INVALID_ARG;
    if (nq && (!queries || !luts)) return YAMS_ERR_INVALID_ARG;
    if (candidates == nullptr) n_candidates = 0;
    std::shared_lock<std::shared_mutex> lk(c->mu);
    if (c->sh.size() != 1) return YAMS_ERR_UNSUPPORTED;
    const ShardStore& s = c->sh[0];
    const PqIndex& p = c->pq;
    const size_t slots = static_cast<size_t>(nq) * std::max<uint32_t>(k, 1);
    auto* counts = static_cast<uint32_t*>(std::calloc(std::max<uint32_t>(nq, 1), sizeof(uint32_t)));
    auto* hits = static_cast<yams_scan_hit_t*>(std::calloc(std::max<size_t>(slots, 1), sizeof(yams_scan_hit_t)));
    if (!counts || !hits) { std::free(counts); std::free(hits); return YAMS_ERR_INTERNAL; }
    auto done = [&](yams_status_t st) { if (st != YAMS_OK) { std::free(counts); std::free(hits); } else { *out_hits = hits; *out_counts = counts; } return st; };
    for (size_t o = 0; o < slots; ++o) hits[o].row = -1;
    // no index (or an empty one), no rows, k == 0, an empty candidate list: nothing (:3873-3880, :3946-3948)
    if (nq == 0 || k == 0 || p.n == 0 || s.n_rows == 0 || (candidates && n_candidates == 0)) { if (out_diag) std::memset(out_diag, 0, sizeof *out_diag); return done(YAMS_OK); }
    (void)hipSetDevice(s.device);
    Lease<yams_accel_ctx*> w(g.work_ctx);
    yams_accel_ctx* x = w.v;
    float* d_q; float* d_l; uint32_t* d_c = nullptr; float* d_s; int64_t* d_r; uint32_t* d_n;
    yams_status_t st;
    if ((st = yams_accel::ws_get(x, "plugin_pq_queries", static_cast<size_t>(nq) * dim * 4, (void**)&d_q)) != YAMS_OK) return done(st);
    if ((st = yams_accel::ws_get(x, "plugin_pq_luts", static_cast<size_t>(nq) * p.m * 1024, (void**)&d_l)) != YAMS_OK) return done(st);
    if (candidates && (st = yams_accel::ws_get(x, "plugin_pq_candidates", n_candidates * 4, (void**)&d_c)) != YAMS_OK) return done(st);
    if ((st = yams_accel::ws_get(x, "plugin_pq_scores", slots * 4, (void**)&d_s)) != YAMS_OK) return done(st);
    if ((st = yams_accel::ws_get(x, "plugin_pq_rows", slots * 8, (void**)&d_r)) != YAMS_OK) return done(st);
    if ((st = yams_accel::ws_get(x, "plugin_pq_counts", static_cast<size_t>(nq) * 4, (void**)&d_n)) != YAMS_OK) return done(st);
    if (yams_accel_upload(x, d_q, queries, static_cast<size_t>(nq) * dim * 4) != YAMS_OK ||
        yams_accel_upload(x, d_l, luts, static_cast<size_t>(nq) * p.m * 1024) != YAMS_OK ||
        (candidates && yams_accel_upload(x, d_c, candidates, n_candidates * 4) != YAMS_OK)) return done(YAMS_ERR_INTERNAL);
    yams_scan_corpus_t v;
    std::memset(&v, 0, sizeof v);
    v.rows = s.rows.as<float>(); v.n_rows = s.n_rows; v.dim = c->dim;
    if (s.has_tie) { v.tie_rank = s.tie.as<uint32_t>(); v.rank_row = s.inv.as<uint32_t>(); }
    yams_scan_pq_index_t pi{p.codes, p.n, p.m, 0, p.tie_rank, p.key_row};
    yams_scan_pq_params_t prm{k, threshold, rerank_factor, flags & YAMS_PQ_SUM_MASK};
    if ((st = yams_scan_pq_topk_device(x, &v, &pi, d_q, d_l, nq, &prm, d_c, n_candidates, d_s, d_r, d_n, out_diag)) != YAMS_OK) return done(st);
    std::vector<float> scores(slots); std::vector<int64_t> rows(slots);
    if (yams_accel_download(x, counts, d_n, static_cast<size_t>(nq) * 4) != YAMS_OK || yams_accel_download(x, scores.data(), d_s, slots * 4) != YAMS_OK ||
        yams_accel_download(x, rows.data(), d_r, slots * 8) != YAMS_OK) return done(YAMS_ERR_INTERNAL);
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t i = 0; i < k && i < counts[q]; ++i) {
            const size_t o = static_cast<size_t>(q) * k + i;
            hits[o].row = rows[o]; hits[o].similarity = scores[o]; hits[o].distance = 1.0f - scores[o];
        }
    ++g.searches;
    return done(YAMS_OK);
}

void vs_free_hits(void*, yams_scan_hit_t* hits, uint32_t* counts) { std::free(hits); std::free(counts); }

yams_status_t vs_runtime_info(void*, char** out_json) {
    NEED_INIT();
    Lease<yams_accel_ctx*> w(g.work_ctx);
    return yams_accel_device_info_json(w.v, out_json);
}
void vs_free_string(void*, char* s) { std::free(s); }

yams_vector_scan_v1 g_vector_scan = {
    YAMS_IFACE_VECTOR_SCAN_V1_VERSION, nullptr, GUARDED(vs_corpus_create), GUARDED(vs_corpus_append),
    GUARDED(vs_corpus_set_tie_ranks), GUARDED(vs_corpus_clear), GUARDED(vs_corpus_destroy), GUARDED(vs_corpus_size),
    GUARDED(vs_search_batch), GUARDED(vs_free_hits), GUARDED(vs_runtime_info), GUARDED(vs_free_string),
    GUARDED(vs_search_batch_masked), GUARDED(vs_search_batch_ex), GUARDED(vs_pq_index_set), GUARDED(vs_search_pq)};

// ---- content_hash_v1 --------------------------------------------------------------------------
// Every call leases one of the plugin's work contexts (own stream, own workspace), so hashing, chunking
// and searches of different host threads overlap on the device instead of queueing on one mutex.
// One SHA-256 chain is sequential: ~35 MB/s on a device lane, > 1 GB/s on a host core.  Work the device is worse
// at is refused (YAMS_ERR_UNSUPPORTED: the host hashes it itself), not served slowly — see the header.
bool chains_suit_the_device(const size_t* lens, size_t n) {
    size_t longest = 0, total = 0;
    for (size_t i = 0; i < n; ++i) { longest = std::max(longest, lens[i]); total += lens[i]; }
    return longest <= std::max<size_t>(YAMS_HASH_LONE_CHAIN_MAX, total / YAMS_HASH_CHAIN_RATIO);
}
yams_status_t ch_hash(void*, const uint8_t* data, size_t n, char out_hex[65]) {
    NEED_INIT();
    if (n > YAMS_HASH_LONE_CHAIN_MAX) { ++g.refused_chains; return YAMS_ERR_UNSUPPORTED; }
    Lease<yams_accel_ctx*> w(g.work_ctx);
    ++g.hashes;
    return yams_sha256_host(w.v, data, n, out_hex);
}
yams_status_t ch_hash_many(void*, const uint8_t* const* msgs, const size_t* lens, size_t n, char* out_hex) {
    NEED_INIT();
    if (n && lens && !chains_suit_the_device(lens, n)) { ++g.refused_chains; return YAMS_ERR_UNSUPPORTED; }
    Lease<yams_accel_ctx*> w(g.work_ctx);
    g.hashes += n;
    return yams_sha256_many_host(w.v, msgs, lens, n, out_hex);
}

// Streaming state (sha256_hasher.cpp:81-109).  update() only BUFFERS on the host: one SHA-256 chain is
// sequential, a launch + sync per update would cost more than the hashing.  Whole 64-byte blocks are
// pushed through the device in pieces of kStreamFlush bytes (one upload + one kernel each), the tail
// and the FIPS 180-4 padding at finalize().  A handle is a single-threaded object, like the reference's
// hasher (sha256_hasher.cpp:34).
struct HashStream {
    uint32_t state[8];
    std::vector<uint8_t> pending; // bytes not yet compressed
    uint64_t total = 0;           // bytes compressed so far + pending
};
constexpr size_t kStreamFlush = 64u << 20;
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
yams_status_t stream_blocks(yams_accel_ctx* ctx, HashStream* hs, const uint8_t* bytes, size_t n) {
    using namespace yams_accel;
    if (n == 0) return YAMS_OK;
    (void)hipSetDevice(ctx->device);
    uint8_t* d_data; uint64_t* d_tab; uint32_t* d_state; unsigned long long* d_head;
    YA_TRY(ws_get(ctx, "hs_data", n + 64, (void**)&d_data));
    YA_TRY(ws_get(ctx, "hs_tab", 64, (void**)&d_tab));
    YA_TRY(ws_get(ctx, "hs_state", 64, (void**)&d_state));
    YA_TRY(ws_get(ctx, "ing_queue", 64, (void**)&d_head));
    const uint64_t tab[2] = {0, n};
    uint32_t out[8];
    YA_HIP(ctx, hipMemcpyAsync(d_data, bytes, n, hipMemcpyHostToDevice, ctx->stream));
    YA_HIP(ctx, hipMemcpyAsync(d_tab, tab, 16, hipMemcpyHostToDevice, ctx->stream));
    YA_HIP(ctx, hipMemcpyAsync(d_state, hs->state, 32, hipMemcpyHostToDevice, ctx->stream));
    YA_HIP(ctx, launch_sha256(ctx->stream, d_data, d_tab, d_tab + 1, 0, 1, nullptr, d_head, d_state,
                              d_state + 8, 1, 1, 1));
    YA_HIP(ctx, hipMemcpyAsync(out, d_state + 8, 32, hipMemcpyDeviceToHost, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(hs->state, out, 32); // the handle changes only after the device call succeeded
    return YAMS_OK;
}
yams_status_t flush_whole_blocks(HashStream* hs) {
    const size_t whole = hs->pending.size() / 64 * 64;
    if (whole == 0) return YAMS_OK;
    Lease<yams_accel_ctx*> w(g.work_ctx);
    YA_TRY(stream_blocks(w.v, hs, hs->pending.data(), whole));
    hs->pending.erase(hs->pending.begin(), hs->pending.begin() + static_cast<std::ptrdiff_t>(whole));
    return YAMS_OK;
}

yams_status_t ch_stream_create(void*, void** out) {
    NEED_INIT();
    if (!out) return YAMS_ERR_INVALID_ARG;
    auto* hs = new HashStream();
    std::memcpy(hs->state, kShaInit, 32);
    *out = hs;
    return YAMS_OK;
}
yams_status_t ch_stream_init(void*, void* s) {
    NEED_INIT();
    if (!s) return YAMS_ERR_INVALID_ARG;
    auto* hs = static_cast<HashStream*>(s);
    std::memcpy(hs->state, kShaInit, 32);
    hs->pending.clear(); hs->total = 0;
    return YAMS_OK;
}
yams_status_t ch_stream_update(void*, void* s, const uint8_t* data, size_t n) {
    NEED_INIT();
    if (!s || (n && !data)) return YAMS_ERR_INVALID_ARG;
    auto* hs = static_cast<HashStream*>(s);
    size_t pos = 0;
    while (pos < n) { // bounded host buffer: flush whole blocks every kStreamFlush bytes
        const size_t take = std::min(n - pos, kStreamFlush - std::min(kStreamFlush, hs->pending.size()) + 64);
        hs->pending.insert(hs->pending.end(), data + pos, data + pos + take);
        hs->total += take;
        pos += take;
        if (hs->pending.size() >= kStreamFlush) {
            const yams_status_t st = flush_whole_blocks(hs);
            if (st != YAMS_OK) return st;
        }
    }
    return YAMS_OK;
}
yams_status_t ch_stream_finalize(void*, void* s, char out_hex[65]) {
    NEED_INIT();
    if (!s || !out_hex) return YAMS_ERR_INVALID_ARG;
    auto* hs = static_cast<HashStream*>(s);
    // padding: 0x80, zeros, the bit length as a big-endian 64-bit word
    const uint64_t bits = hs->total * 8;
    hs->pending.push_back(0x80);
    while (hs->pending.size() % 64 != 56) hs->pending.push_back(0);
    for (int i = 7; i >= 0; --i) hs->pending.push_back(static_cast<uint8_t>(bits >> (8 * i)));
    const yams_status_t st = flush_whole_blocks(hs);
    if (st != YAMS_OK) return st;
    static const char kHex[] = "0123456789abcdef";
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) {
            const uint8_t v = static_cast<uint8_t>(hs->state[i] >> (24 - 8 * b));
            out_hex[8 * i + 2 * b] = kHex[v >> 4]; out_hex[8 * i + 2 * b + 1] = kHex[v & 15];
        }
    out_hex[64] = 0;
    // "Reset for potential reuse" (sha256_hasher.cpp:103-106)
    std::memcpy(hs->state, kShaInit, 32);
    hs->pending.clear(); hs->total = 0;
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
    if (!chains_suit_the_device(lens, n)) { ++g.refused_chains; return YAMS_ERR_UNSUPPORTED; }
    std::vector<char> hex(n * 65);
    {
        NEED_INIT();
        Lease<yams_accel_ctx*> w(g.work_ctx);
        g.hashes += n;
        yams_status_t s = yams_sha256_many_host(w.v, msgs, lens, n, hex.data());
        if (s != YAMS_OK) return s;
    }
    for (size_t i = 0; i < n; ++i) { // the reference compares the lower-case hex strings (:243-249)
        uint8_t a[32], b[32];
        out_valid[i] = (parse_hex32(hex.data() + 65 * i, a) && parse_hex32(expected_hex + 65 * i, b) &&
                        std::memcmp(a, b, 32) == 0) ? 1 : 0;
    }
    return YAMS_OK;
}

// Dedup sets: each owns a context (its table lives in that context's device memory) and a mutex.
struct DedupEntry { yams_accel_ctx* ctx = nullptr; yams_dedup_set* set = nullptr; std::mutex mu; };
std::mutex g_dedup_mu;
std::map<uint64_t, std::shared_ptr<DedupEntry>> g_dedup;
uint64_t g_next_dedup = 1;

std::shared_ptr<DedupEntry> find_dedup(uint64_t id) {
    std::lock_guard<std::mutex> lk(g_dedup_mu);
    auto it = g_dedup.find(id);
    return it == g_dedup.end() ? nullptr : it->second;
}
void destroy_dedup(DedupEntry& e) {
    if (e.set) yams_dedup_set_destroy(e.set);
    if (e.ctx) yams_accel_ctx_destroy(e.ctx);
    e.set = nullptr; e.ctx = nullptr;
}

yams_status_t ch_dedup_create(void*, uint64_t expected, uint64_t* out_id) {
    NEED_INIT();
    if (!out_id) return YAMS_ERR_INVALID_ARG;
    auto e = std::make_shared<DedupEntry>();
    yams_status_t st = yams_accel_ctx_create(g.devices[0], nullptr, &e->ctx);
    if (st != YAMS_OK) return st;
    st = yams_dedup_set_create(e->ctx, expected, &e->set);
    if (st != YAMS_OK) { destroy_dedup(*e); return st; }
    std::lock_guard<std::mutex> lk(g_dedup_mu);
    *out_id = g_next_dedup++;
    g_dedup[*out_id] = e;
    return YAMS_OK;
}
yams_status_t dedup_call(uint64_t id, const char* hashes_hex, size_t n, uint8_t* out, bool insert) {
    NEED_INIT();
    auto e = find_dedup(id);
    if (!e) return YAMS_ERR_NOT_FOUND;
    if (n == 0) return YAMS_OK;
    if (!hashes_hex || !out) return YAMS_ERR_INVALID_ARG;
    std::vector<uint8_t> raw(n * 32);
    for (size_t i = 0; i < n; ++i)
        if (!parse_hex32(hashes_hex + 65 * i, raw.data() + 32 * i)) return YAMS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->set) return YAMS_ERR_NOT_FOUND;
    return insert ? yams_dedup_insert_host(e->set, raw.data(), n, out, nullptr)
                  : yams_dedup_probe_host(e->set, raw.data(), n, out);
}
yams_status_t ch_dedup_insert(void*, uint64_t id, const char* hex, size_t n, uint8_t* out) { return dedup_call(id, hex, n, out, true); }
yams_status_t ch_dedup_contains(void*, uint64_t id, const char* hex, size_t n, uint8_t* out) { return dedup_call(id, hex, n, out, false); }
yams_status_t ch_dedup_size(void*, uint64_t id, uint64_t* out_entries) {
    NEED_INIT();
    auto e = find_dedup(id);
    if (!e) return YAMS_ERR_NOT_FOUND;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->set) return YAMS_ERR_NOT_FOUND;
    return yams_dedup_set_size(e->set, out_entries);
}
yams_status_t ch_dedup_destroy(void*, uint64_t id) {
    NEED_INIT();
    std::shared_ptr<DedupEntry> e;
    {
        std::lock_guard<std::mutex> lk(g_dedup_mu);
        auto it = g_dedup.find(id);
        if (it == g_dedup.end()) return YAMS_ERR_NOT_FOUND;
        e = it->second;
        g_dedup.erase(it);
    }
    std::lock_guard<std::mutex> lk(e->mu);
    destroy_dedup(*e);
    return YAMS_OK;
}

yams_content_hash_v1 g_content_hash = {YAMS_IFACE_CONTENT_HASH_V1_VERSION, nullptr, GUARDED(ch_hash),
                                       GUARDED(ch_hash_many), GUARDED(ch_stream_create), GUARDED(ch_stream_init),
                                       GUARDED(ch_stream_update), GUARDED(ch_stream_finalize), GUARDED(ch_stream_destroy),
                                       GUARDED(ch_verify_many), GUARDED(ch_dedup_create), GUARDED(ch_dedup_insert),
                                       GUARDED(ch_dedup_contains), GUARDED(ch_dedup_size), GUARDED(ch_dedup_destroy)};

// ---- chunker_v1 -------------------------------------------------------------------------------
yams_status_t ck_default_config(void*, uint32_t mode, yams_cdc_config_t* out_cfg) {
    if (!out_cfg || (mode != YAMS_CDC_RABIN && mode != YAMS_CDC_STREAMING)) return YAMS_ERR_INVALID_ARG;
    yams_cdc_default_config(out_cfg, mode);
    return YAMS_OK;
}
// re-entrant: ContentStore shares one chunker across its workers (content_store_impl.cpp:1412)
// One buffer; the first context_len bytes are history (chunk_window: a window of a stream), 0 for chunk_data.
yams_status_t ck_chunk_window(void*, const uint8_t* data, size_t n, size_t context_len, const yams_cdc_config_t* cfg,
                              yams_chunk_ref_t** out_chunks, size_t* out_count) {
    NEED_INIT();
    if (!out_chunks || !out_count || !cfg) return YAMS_ERR_INVALID_ARG;
    *out_chunks = nullptr; *out_count = 0;
    const uint64_t floor = std::max<uint64_t>(1, cfg->min_size);
    size_t cap = n / floor + 2;
    std::vector<uint64_t> off(cap), sz(cap);
    std::vector<char> hex(cap * 65);
    size_t cnt = 0;
    yams_status_t s;
    {
        Lease<yams_accel_ctx*> w(g.work_ctx);
        s = yams_cdc_chunk_window_host(w.v, data, n, context_len, cfg, off.data(), sz.data(), hex.data(), cap, &cnt);
    }
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
yams_status_t ck_chunk_data(void* self, const uint8_t* data, size_t n, const yams_cdc_config_t* cfg,
                            yams_chunk_ref_t** out_chunks, size_t* out_count) {
    return ck_chunk_window(self, data, n, 0, cfg, out_chunks, out_count);
}
void ck_free_chunks(void*, yams_chunk_ref_t* chunks, size_t) { std::free(chunks); }

void to_hex(const uint8_t* d, char out[65]) {
    static const char kHexDigits[] = "0123456789abcdef";
    for (int i = 0; i < 32; ++i) { out[2 * i] = kHexDigits[d[i] >> 4]; out[2 * i + 1] = kHexDigits[d[i] & 15]; }
    out[64] = 0;
}
void ck_free_chunk_batch(void*, yams_chunk_batch_t* b) {
    if (!b) return;
    std::free(b->first_chunk); std::free(b->chunks); std::free(b->buffer_hash_hex); std::free(b);
}
// Many buffers per call: the batched ingest path (yams_ingest_host) behind the plugin door.
yams_status_t ck_chunk_many(void*, const uint8_t* const* buffers, const size_t* lens, size_t n, const yams_cdc_config_t* cfg,
                            uint32_t flags, yams_chunk_batch_t** out_batch) {
    NEED_INIT();
    if (!out_batch) return YAMS_ERR_INVALID_ARG;
    *out_batch = nullptr;
    if (!cfg || (n && (!buffers || !lens))) return YAMS_ERR_INVALID_ARG;
    const bool want_blob = (flags & YAMS_CHUNK_MANY_BUFFER_HASHES) != 0;
    const bool defer = want_blob && (flags & YAMS_CHUNK_MANY_DEFER_LONG_BUFFER_HASHES) != 0;
    // First guess of the chunk count: one chunk per `floor` bytes, where floor is the smallest chunk the configuration
    // can emit in the common case — min(min, max), but never below 256 bytes (min_size == 0 is a legal configuration:
    // a per-byte bound would be ~48 bytes of host vectors per input byte).  Should the guess be too low (min > max, or
    // a degenerate mask) yams_ingest_host reports the required size and the call runs once more with exactly that.
    uint64_t floor = std::min<uint64_t>(cfg->min_size ? cfg->min_size : cfg->max_size, cfg->max_size ? cfg->max_size : cfg->min_size);
    floor = std::max<uint64_t>(floor, 256);
    std::vector<uint64_t> len64(n);
    uint64_t cap = 0, total = 0;
    for (size_t i = 0; i < n; ++i) { len64[i] = lens[i]; cap += lens[i] / floor + 2; total += lens[i]; }
    std::vector<uint64_t> first(n + 1, 0), off, sz;
    std::vector<uint8_t> dig, bdig(want_blob ? n * 32 : 0);
    uint64_t cnt = 0;
    if (n) {
        Lease<yams_accel_ctx*> w(g.work_ctx);
        for (int attempt = 0; ; ++attempt) {
            off.assign(cap, 0); sz.assign(cap, 0); dig.assign(cap * 32, 0);
            const yams_status_t s = yams_ingest_host(w.v, buffers, len64.data(), n, cfg,
                                                     YAMS_INGEST_CHUNK_DIGESTS | (want_blob ? YAMS_INGEST_BLOB_DIGESTS : 0u) |
                                                         (defer ? YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS : 0u), 0,
                                                     first.data(), off.data(), sz.data(), dig.data(), cap,
                                                     want_blob ? bdig.data() : nullptr, &cnt);
            if (s == YAMS_OK) break;
            if (s == YAMS_ERR_INVALID_ARG && attempt == 0 && cnt > cap) { cap = cnt; continue; } // "chunk arrays too small": cnt is the required size
            return s;
        }
    }
    const uint64_t defer_above = defer ? yams_ingest_defer_threshold_host(total) : UINT64_MAX;
    auto* b = static_cast<yams_chunk_batch_t*>(std::calloc(1, sizeof(yams_chunk_batch_t)));
    if (!b) return YAMS_ERR_INTERNAL;
    b->n_buffers = n; b->n_chunks = static_cast<size_t>(cnt);
    b->first_chunk = static_cast<size_t*>(std::calloc(n + 1, sizeof(size_t)));
    b->chunks = static_cast<yams_chunk_ref_t*>(std::calloc(std::max<size_t>(b->n_chunks, 1), sizeof(yams_chunk_ref_t)));
    b->buffer_hash_hex = want_blob ? static_cast<char*>(std::calloc(std::max<size_t>(n, 1), 65)) : nullptr;
    if (!b->first_chunk || !b->chunks || (want_blob && !b->buffer_hash_hex)) { ck_free_chunk_batch(nullptr, b); return YAMS_ERR_INTERNAL; }
    for (size_t i = 0; i <= n; ++i) b->first_chunk[i] = static_cast<size_t>(first[i]);
    for (size_t i = 0; i < b->n_chunks; ++i) {
        b->chunks[i].offset = off[i]; b->chunks[i].size = sz[i];
        to_hex(dig.data() + 32 * i, b->chunks[i].hash_hex);
    }
    if (want_blob)
        for (size_t i = 0; i < n; ++i) {
            if (len64[i] > defer_above) { ++g.deferred_chains; continue; } // (calloc'd: the entry stays empty — the host's hasher fills it)
            to_hex(bdig.data() + 32 * i, b->buffer_hash_hex + 65 * i);
        }
    g.chunk_calls += n;
    *out_batch = b;
    return YAMS_OK;
}

yams_chunker_v1 g_chunker = {YAMS_IFACE_CHUNKER_V1_VERSION, nullptr, GUARDED(ck_default_config),
                             GUARDED(ck_chunk_data), GUARDED(ck_free_chunks), GUARDED(ck_chunk_many),
                             GUARDED(ck_free_chunk_batch), GUARDED(ck_chunk_window)};

void teardown_locked() { // g.mu held exclusively
    {
        std::lock_guard<std::mutex> lk(g.corpora_mu);
        for (auto& kv : g.corpora) { std::unique_lock<std::shared_mutex> cl(kv.second->mu); release_corpus(*kv.second); }
        g.corpora.clear();
    }
    {
        std::lock_guard<std::mutex> lk(g_dedup_mu);
        for (auto& kv : g_dedup) { std::lock_guard<std::mutex> el(kv.second->mu); destroy_dedup(*kv.second); }
        g_dedup.clear();
    }
    if (g.sharded) yams_scan_sharded_destroy(g.sharded);
    g.sharded = nullptr;
    for (auto* c : g.work_ctx.drain()) yams_accel_ctx_destroy(c);
    for (auto* c : g.upload_ctx) yams_accel_ctx_destroy(c);
    g.upload_ctx.clear();
    g.initialised = false;
    // nothing reads the parked mirrors of destroyed corpora any more: their physical memory goes back to the driver
    GrowBuf::evict_all_parked();
}

} // namespace

extern "C" {

int yams_plugin_get_abi_version(void) { return YAMS_PLUGIN_ABI_VERSION; }
const char* yams_plugin_get_name(void) { return "yams_mi355x_accel"; }
const char* yams_plugin_get_version(void) { return YAMS_ACCEL_VERSION_STRING; }
const char* yams_plugin_get_manifest_json(void) { return kManifest; }

// host_context is a yams_plugin_host_context_v1* (host_services_v1.h:19-26); unused here.  The
// legacy one-argument form (abi_plugin_loader.cpp:329-357) is tolerated: the second argument is
// never dereferenced.  config_json: {"device": n} or {"devices": [..]} (a corpus is dealt to all of
// them in stripes and searched behind one call), "search_slots": concurrent searches (default 2),
// "shadows": "both" (default) | "bf16" | "i8" | "none", "collective": "auto" | "rccl" | "peer",
// "rccl_library": "<path>", "fence": "off", "l2_accumulate": "f64" (default) | "f32" | "f32x8" | "f32x16" | "f32_fma" |
// "f32x8_fma" | "f32x16_fma".  Read by a strict tokenizer (struct Config): unknown keys are ignored, a known key with a value
// of the wrong type or an enumerated value that is not listed fails the init (YAMS_PLUGIN_ERR_INIT_FAILED).
static int plugin_init_impl(const char* config_json, const void* host_context) {
    (void)host_context;
    std::unique_lock<std::shared_mutex> lk(g.mu);
    if (g.initialised) return YAMS_PLUGIN_OK;
    auto failed = [&](const char* why) {
        g.init_error = why;
        for (char& ch : g.init_error) if (ch == '"' || ch == '\\') ch = '\'';   // (the health JSON quotes it as is)
        teardown_locked();
        return YAMS_PLUGIN_ERR_INIT_FAILED; // the host keeps its built-in CPU backends
    };
    Config cfg(config_json);
    if (!cfg.error.empty()) return failed(("configuration: " + cfg.error).c_str());
    long v_device = 0, slots = 2, sr = 65536, ex_timeout = 0;
    int shadows = 0, l2 = 0, collective = 0, fence = 0;
    std::string library; bool has_library = false;
    const bool cfg_ok =
        cfg.get_int("device", 0, v_device) && cfg.get_int("search_slots", 2, slots) && cfg.get_int("stripe_rows", 65536, sr) &&
        cfg.get_int("exchange_timeout_ms", 0, ex_timeout) &&
        cfg.get_choice("shadows", {"both", "bf16", "i8", "none"}, 0, shadows) &&
        // layout of the int8 shadow (YAMS_SCAN_I8_ROTATED in the header): measured per corpus at its first append, or fixed
        cfg.get_choice("i8_layout", {"auto", "plain", "rotated"}, 0, g.i8_layout) &&
        // the arithmetic of vec0's L2 distance the host's sqlite-vec-cpp build uses (YAMS_SCAN_FLAG_L2_ACC_* in the header;
        // "..._fma": the same lanes accumulated with ONE fused multiply-add per element, what '-mavx', '-mfma' builds do)
        cfg.get_choice("l2_accumulate", {"f64", "f32", "f32x8", "f32x16", "f32_fma", "f32x8_fma", "f32x16_fma"}, 0, l2) &&
        cfg.get_choice("collective", {"auto", "rccl", "peer"}, 0, collective) &&
        cfg.get_choice("fence", {"auto", "on", "off"}, 0, fence) &&
        cfg.get_string("rccl_library", library, has_library);
    if (!cfg_ok) return failed(("configuration: " + cfg.error).c_str());
    g.devices.clear();
    if (cfg.has("devices")) {
        const auto it = cfg.int_lists.find("devices");
        if (it == cfg.int_lists.end() || it->second.empty()) return failed("configuration: \"devices\" must be a non-empty array of integers");
        for (long d : it->second) g.devices.push_back(static_cast<int>(d));
    } else g.devices.push_back(static_cast<int>(v_device));
    slots = std::max<long>(1, std::min<long>(16, slots));
    kStripeRows = static_cast<uint32_t>(std::max<long>(64, std::min<long>(1 << 24, sr)) / 64 * 64);
    g.want_bf16 = shadows == 0 || shadows == 1;
    g.want_i8 = shadows == 0 || shadows == 2;
    static const uint32_t kL2Acc[7] = {YAMS_SCAN_FLAG_L2_ACC_F64, YAMS_SCAN_FLAG_L2_ACC_F32, YAMS_SCAN_FLAG_L2_ACC_F32X8, YAMS_SCAN_FLAG_L2_ACC_F32X16,
                                       YAMS_SCAN_FLAG_L2_ACC_F32 | YAMS_SCAN_FLAG_L2_ACC_FUSED, YAMS_SCAN_FLAG_L2_ACC_F32X8 | YAMS_SCAN_FLAG_L2_ACC_FUSED,
                                       YAMS_SCAN_FLAG_L2_ACC_F32X16 | YAMS_SCAN_FLAG_L2_ACC_FUSED};
    g.l2_acc = kL2Acc[l2];
    g.append_bytes = 0; g.appends = 0; g.exhausted_appends = 0; g.append_map_ms = 0; g.append_copy_ms = 0; g.append_shadow_ms = 0;
    g.slow_append_ms = 0; g.slow_map_ms = 0; g.slow_copy_ms = 0; g.slow_shadow_ms = 0; g.slow_append_bytes = 0;
    for (size_t i = 0; i < g.devices.size(); ++i) {
        yams_accel_ctx* c = nullptr;
        const yams_status_t s = yams_accel_ctx_create(g.devices[i], nullptr, &c);
        if (s != YAMS_OK) return failed(s == YAMS_ERR_UNSUPPORTED ? "no gfx950 device visible" : "context creation failed");
        g.upload_ctx.push_back(c);
    }
    {
        yams_scan_sharded_options_t so{};
        so.struct_size = sizeof so; so.lanes = static_cast<uint32_t>(slots); so.collective = YAMS_SHARDED_COLLECTIVE_AUTO;
        // "collective": "rccl" (require it) | "peer" | "auto"; "rccl_library": "<path>" — the collective library to bind instead of
        // librccl.so.1 (a site build; the test suite's stand-in, with which several shards may share a device); "fence": "off"
        // lifts the exchange fence; "exchange_timeout_ms": deadline of a sharded batch (default 30000; a batch that misses it
        // fails with YAMS_ERR_TIMEOUT and the sharded handle is stuck until the plugin is shut down and initialised again)
        so.collective = collective == 1 ? YAMS_SHARDED_COLLECTIVE_RCCL : (collective == 2 ? YAMS_SHARDED_COLLECTIVE_PEER : YAMS_SHARDED_COLLECTIVE_AUTO);
        if (has_library && !library.empty()) so.rccl_library = library.c_str();
        so.exchange_timeout_ms = static_cast<uint32_t>(std::max<long>(0, ex_timeout));
        if (fence == 2) so.fence = YAMS_SHARDED_FENCE_OFF;
        if (yams_scan_sharded_create_ex(g.devices.data(), static_cast<uint32_t>(g.devices.size()), &so, &g.sharded) != YAMS_OK)
            return failed("sharded search handle creation failed");
        g.search_slots = static_cast<uint32_t>(slots);
    }
    for (long i = 0; i < std::max<long>(2, slots); ++i) {
        yams_accel_ctx* c = nullptr;
        if (yams_accel_ctx_create(g.devices[0], nullptr, &c) != YAMS_OK) return failed("work context creation failed");
        g.work_ctx.add(c);
    }
    g.initialised = true;
    g.init_error.clear();
    return YAMS_PLUGIN_OK;
}

int yams_plugin_init(const char* config_json, const void* host_context) {
    try { return plugin_init_impl(config_json, host_context); }
    catch (...) { return YAMS_PLUGIN_ERR_INIT_FAILED; } // the host keeps its built-in CPU backends
}

void yams_plugin_shutdown(void) {
    try {
        std::unique_lock<std::shared_mutex> lk(g.mu);
        if (g.initialised) teardown_locked();
    } catch (...) {}
}

// Returns a pointer to a static vtable; unknown id or version -> NOT_FOUND, null args -> INVALID
// (plugins/glint/plugin.cpp:338-359, tools/fuzzing/fuzz_abi_test_plugin.c:48-60).
int yams_plugin_get_interface(const char* iface_id, uint32_t version, void** out_iface) {
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
static int plugin_health_impl(char** out_json) {
    if (!out_json) return YAMS_PLUGIN_ERR_INVALID;
    std::shared_lock<std::shared_mutex> lk(g.mu);
    std::ostringstream os;
    size_t n_corpora;
    // device memory behind the mirrors: mapped into live corpora, and parked (mappings of destroyed corpora waiting for
    // the next one) — what an allocation-failure test watches for leaks
    uint64_t mirror_mapped = 0, mirror_parked = 0;
    size_t rotated_corpora = 0;
    {
        std::lock_guard<std::mutex> cl(g.corpora_mu);
        n_corpora = g.corpora.size();
        for (auto& kv : g.corpora) {
            rotated_corpora += kv.second->i8_flags > 0 && (kv.second->i8_flags & static_cast<int>(YAMS_SCAN_I8_ROTATED));
            for (auto& s : kv.second->sh)
                for (const GrowBuf* b : {&s.rows, &s.bf16, &s.i8, &s.nsq, &s.i8meta, &s.tie, &s.inv}) mirror_mapped += b->mapped;
            mirror_mapped += kv.second->rank_of_row.mapped;
        }
    }
    {
        std::lock_guard<std::mutex> pl(g_park_mu);
        for (auto& kv : g_parked) for (auto& b : kv.second) mirror_parked += b.mapped;
    }
    os << "{\"status\":\"" << (g.initialised ? "ok" : "not_initialised") << "\",\"devices\":[";
    for (size_t i = 0; i < g.devices.size(); ++i) os << (i ? "," : "") << g.devices[i];
    os << "],\"device\":" << (g.devices.empty() ? 0 : g.devices[0]) << ",\"search_slots\":" << g.search_slots
       << ",\"l2_accumulate\":\"" << ((g.l2_acc & YAMS_SCAN_FLAG_L2_ACC_MASK) == YAMS_SCAN_FLAG_L2_ACC_F32X16 ? "f32x16" : (g.l2_acc & YAMS_SCAN_FLAG_L2_ACC_MASK) == YAMS_SCAN_FLAG_L2_ACC_F32X8 ? "f32x8" : (g.l2_acc & YAMS_SCAN_FLAG_L2_ACC_MASK) == YAMS_SCAN_FLAG_L2_ACC_F32 ? "f32" : "f64")
       << ((g.l2_acc & YAMS_SCAN_FLAG_L2_ACC_FUSED) ? "_fma" : "") << "\""
       << ",\"i8_layout\":\"" << (g.i8_layout == 2 ? "rotated" : g.i8_layout == 1 ? "plain" : "auto") << "\",\"corpora_with_rotated_i8_shadow\":" << rotated_corpora
       << ",\"last_append\":{\"bytes\":" << g.append_bytes.load() << ",\"map_ms\":" << g.append_map_ms.load() << ",\"copy_ms\":" << g.append_copy_ms.load()
       << ",\"shadow_ms\":" << g.append_shadow_ms.load() << "}"
       << ",\"appends\":" << g.appends.load() << ",\"exhausted_appends\":" << g.exhausted_appends.load()
       << ",\"slowest_append\":{\"bytes\":" << g.slow_append_bytes.load() << ",\"ms\":" << g.slow_append_ms.load() << ",\"map_ms\":" << g.slow_map_ms.load()
       << ",\"copy_ms\":" << g.slow_copy_ms.load() << ",\"shadow_ms\":" << g.slow_shadow_ms.load() << "}"
       << ",\"mirror_bytes_mapped\":" << mirror_mapped << ",\"mirror_bytes_parked\":" << mirror_parked
       << ",\"alloc_faults_injected\":" << yams_accel_debug_alloc_faults()
       << ",\"corpora\":" << n_corpora << ",\"searches\":" << g.searches.load() << ",\"hashes\":" << g.hashes.load()
       << ",\"chunk_calls\":" << g.chunk_calls.load()
       << ",\"refused_lone_chains\":" << g.refused_chains.load() << ",\"deferred_buffer_hashes\":" << g.deferred_chains.load();
    if (g.sharded) { // how the shards exchange their records: "collective":"rccl" | "peer_copy" | "none" (one device)
        char* info = nullptr;
        if (yams_scan_sharded_info_json(g.sharded, &info) == YAMS_OK && info) { os << ",\"sharded\":" << info; std::free(info); }
    }
    if (!g.init_error.empty()) os << ",\"error\":\"" << g.init_error << "\"";
    os << "}";
    const std::string s = os.str();
    char* buf = static_cast<char*>(std::malloc(s.size() + 1));
    if (!buf) return YAMS_PLUGIN_ERR_INIT_FAILED;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    *out_json = buf;
    return YAMS_PLUGIN_OK;
}

int yams_plugin_get_health_json(char** out_json) {
    try { return plugin_health_impl(out_json); }
    catch (...) { return YAMS_PLUGIN_ERR_INIT_FAILED; }
}

} // extern "C"
