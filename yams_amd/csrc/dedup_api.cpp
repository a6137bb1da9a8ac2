// dedup_api.cpp — C ABI of the device-resident digest set (chunk dedup lookup).
#include <algorithm>
#include <cstring>

#include "accel_ctx.h"
#include "dedup_launch.h"

using namespace yams_accel;

struct yams_dedup_set {
    yams_accel_ctx* ctx = nullptr;
    DedupTable t{};
    unsigned long long* d_count = nullptr; // device: number of entries
    uint64_t count = 0;                    // host copy (exact after every call)
};

namespace {

void free_table(DedupTable& t) {
    if (t.tags) (void)hipFree(t.tags);
    if (t.keys) (void)hipFree(t.keys);
    if (t.owner) (void)hipFree(t.owner);
    if (t.fresh) (void)hipFree(t.fresh);
    t = DedupTable{};
}

yams_status_t alloc_table(yams_accel_ctx* ctx, uint32_t capacity, DedupTable* out) {
    DedupTable t{};
    t.capacity = capacity;
    const size_t c = capacity;
    if (ya_malloc(reinterpret_cast<void**>(&t.tags), c * 8) != hipSuccess || ya_malloc(reinterpret_cast<void**>(&t.keys), c * 32) != hipSuccess ||
        ya_malloc(reinterpret_cast<void**>(&t.owner), c * 4) != hipSuccess || ya_malloc(reinterpret_cast<void**>(&t.fresh), c) != hipSuccess) {
        (void)hipGetLastError();
        free_table(t);
        return fail(ctx, YAMS_ERR_RESOURCE_EXHAUSTED, "out of device memory for the digest set");
    }
    YA_HIP(ctx, hipMemsetAsync(t.tags, 0, c * 8, ctx->stream));
    YA_HIP(ctx, hipMemsetAsync(t.fresh, 0, c, ctx->stream));
    YA_HIP(ctx, launch_dedup_fill_owner(ctx->stream, t.owner, capacity));
    *out = t;
    return YAMS_OK;
}

uint32_t capacity_for(uint64_t entries) {
    uint64_t want = std::max<uint64_t>(1024, entries * 2);
    uint64_t c = 1024;
    while (c < want) c <<= 1;
    return static_cast<uint32_t>(std::min<uint64_t>(c, 1ull << 31));
}

yams_status_t ensure_room(yams_dedup_set* s, uint64_t incoming) {
    const uint64_t need = s->count + incoming;
    if (need * 2 <= s->t.capacity) return YAMS_OK;
    if (need * 2 > (1ull << 31)) return fail(s->ctx, YAMS_ERR_UNSUPPORTED, "digest set would exceed 2^30 entries");
    DedupTable nt{};
    YA_TRY(alloc_table(s->ctx, capacity_for(need * 2), &nt));
    YA_HIP(s->ctx, launch_dedup_rehash(s->ctx->stream, s->t, nt));
    YA_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    free_table(s->t);
    s->t = nt;
    return YAMS_OK;
}

} // namespace

extern "C" {

yams_status_t yams_dedup_set_create(yams_accel_ctx* ctx, uint64_t expected_entries, yams_dedup_set** out) {
    if (!ctx || !out) return YAMS_ERR_INVALID_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    auto* s = new yams_dedup_set();
    s->ctx = ctx;
    yams_status_t st = alloc_table(ctx, capacity_for(expected_entries), &s->t);
    if (st != YAMS_OK) { delete s; return st; }
    if (ya_malloc(reinterpret_cast<void**>(&s->d_count), 8) != hipSuccess) { (void)hipGetLastError(); free_table(s->t); delete s; return YAMS_ERR_RESOURCE_EXHAUSTED; }
    YA_HIP(ctx, hipMemsetAsync(s->d_count, 0, 8, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return YAMS_OK;
}

void yams_dedup_set_destroy(yams_dedup_set* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    free_table(s->t);
    if (s->d_count) (void)hipFree(s->d_count);
    delete s;
}

yams_status_t yams_dedup_set_size(const yams_dedup_set* s, uint64_t* out_entries) {
    if (!s || !out_entries) return YAMS_ERR_INVALID_ARG;
    *out_entries = s->count;
    return YAMS_OK;
}

yams_status_t yams_dedup_insert_device(yams_dedup_set* s, const uint8_t* digests, uint64_t n,
                                       const uint64_t* chunk_sizes, uint8_t* out_is_new,
                                       uint64_t* out_n_new, uint64_t* out_bytes_new,
                                       uint64_t* out_bytes_deduped) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    yams_accel_ctx* ctx = s->ctx;
    if (out_n_new) *out_n_new = 0;
    if (out_bytes_new) *out_bytes_new = 0;
    if (out_bytes_deduped) *out_bytes_deduped = 0;
    if (n == 0) return YAMS_OK;
    if (!digests || !out_is_new) return fail(ctx, YAMS_ERR_INVALID_ARG, "null digests / out_is_new");
    if (reinterpret_cast<uintptr_t>(digests) & 7u) return fail(ctx, YAMS_ERR_INVALID_ARG, "digests must be 8-byte aligned");
    if (n >= (1ull << 31)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "more than 2^31 digests per call");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    YA_TRY(ensure_room(s, n));
    const uint32_t nn = static_cast<uint32_t>(n);
    uint8_t* d_pending; uint32_t* d_start; uint32_t* d_slot; unsigned int* d_unres; unsigned long long* d_bytes;
    YA_TRY(ws_get(ctx, "dd_pending", n, (void**)&d_pending));
    YA_TRY(ws_get(ctx, "dd_start", n * 4, (void**)&d_start));
    YA_TRY(ws_get(ctx, "dd_slot", n * 4, (void**)&d_slot));
    YA_TRY(ws_get(ctx, "dd_unres", 64, (void**)&d_unres));
    YA_TRY(ws_get(ctx, "dd_bytes", 64, (void**)&d_bytes));
    unsigned int* h_unres; // pinned
    YA_TRY(pinned_get(ctx, 64, (void**)&h_unres));
    const uint64_t* d64 = reinterpret_cast<const uint64_t*>(digests);
    TimedRegion tr(ctx, "dedup_insert");
    for (int round = 0;; ++round) {
        YA_HIP(ctx, hipMemsetAsync(d_unres, 0, 4, st));
        YA_HIP(ctx, launch_dedup_round(st, s->t, d64, nn, d_pending, d_start, d_slot, out_is_new, d_unres, round == 0));
        YA_HIP(ctx, hipMemcpyAsync(h_unres, d_unres, 4, hipMemcpyDeviceToHost, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        if (*h_unres == 0) break;
        if (round > 64) return fail(ctx, YAMS_ERR_INTERNAL, "digest set did not converge");
    }
    YA_HIP(ctx, launch_dedup_settle(st, s->t, nn, d_slot, out_is_new, s->d_count));
    tr.end();
    unsigned long long h[3] = {0, 0, 0};
    if (chunk_sizes && (out_bytes_new || out_bytes_deduped)) {
        YA_HIP(ctx, hipMemsetAsync(d_bytes, 0, 16, st));
        YA_HIP(ctx, launch_dedup_bytes(st, out_is_new, chunk_sizes, nn, d_bytes));
        YA_HIP(ctx, hipMemcpyAsync(h + 1, d_bytes, 16, hipMemcpyDeviceToHost, st));
    }
    YA_HIP(ctx, hipMemcpyAsync(h, s->d_count, 8, hipMemcpyDeviceToHost, st));
    YA_HIP(ctx, hipStreamSynchronize(st));
    if (out_n_new) *out_n_new = h[0] - s->count;
    s->count = h[0];
    if (out_bytes_new) *out_bytes_new = h[1];
    if (out_bytes_deduped) *out_bytes_deduped = h[2];
    return YAMS_OK;
}

yams_status_t yams_dedup_probe_device(yams_dedup_set* s, const uint8_t* digests, uint64_t n,
                                      uint8_t* out_exists) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    yams_accel_ctx* ctx = s->ctx;
    if (n == 0) return YAMS_OK;
    if (!digests || !out_exists) return fail(ctx, YAMS_ERR_INVALID_ARG, "null digests / out_exists");
    if (reinterpret_cast<uintptr_t>(digests) & 7u) return fail(ctx, YAMS_ERR_INVALID_ARG, "digests must be 8-byte aligned");
    if (n >= (1ull << 31)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "more than 2^31 digests per call");
    (void)hipSetDevice(ctx->device);
    TimedRegion tr(ctx, "dedup_probe");
    YA_HIP(ctx, launch_dedup_probe(ctx->stream, s->t, reinterpret_cast<const uint64_t*>(digests),
                                   static_cast<uint32_t>(n), out_exists));
    tr.end();
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return YAMS_OK;
}

yams_status_t yams_dedup_insert_host(yams_dedup_set* s, const uint8_t* digests_host, uint64_t n,
                                     uint8_t* out_is_new_host, uint64_t* out_n_new) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    if (out_n_new) *out_n_new = 0;
    if (n == 0) return YAMS_OK;
    if (!digests_host || !out_is_new_host) return fail(s->ctx, YAMS_ERR_INVALID_ARG, "null host buffers");
    yams_accel_ctx* ctx = s->ctx;
    uint8_t* d_dg; uint8_t* d_new;
    YA_TRY(ws_get(ctx, "dd_h_digests", n * 32, (void**)&d_dg));
    YA_TRY(ws_get(ctx, "dd_h_flags", n, (void**)&d_new));
    YA_TRY(yams_accel_upload(ctx, d_dg, digests_host, n * 32));
    YA_TRY(yams_dedup_insert_device(s, d_dg, n, nullptr, d_new, out_n_new, nullptr, nullptr));
    return yams_accel_download(ctx, out_is_new_host, d_new, n);
}

yams_status_t yams_dedup_probe_host(yams_dedup_set* s, const uint8_t* digests_host, uint64_t n,
                                    uint8_t* out_exists_host) {
    if (!s) return YAMS_ERR_INVALID_ARG;
    if (n == 0) return YAMS_OK;
    if (!digests_host || !out_exists_host) return fail(s->ctx, YAMS_ERR_INVALID_ARG, "null host buffers");
    yams_accel_ctx* ctx = s->ctx;
    uint8_t* d_dg; uint8_t* d_ex;
    YA_TRY(ws_get(ctx, "dd_h_digests", n * 32, (void**)&d_dg));
    YA_TRY(ws_get(ctx, "dd_h_flags", n, (void**)&d_ex));
    YA_TRY(yams_accel_upload(ctx, d_dg, digests_host, n * 32));
    YA_TRY(yams_dedup_probe_device(s, d_dg, n, d_ex));
    return yams_accel_download(ctx, out_exists_host, d_ex, n);
}

} // extern "C"
