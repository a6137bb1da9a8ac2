// ingest_api.cpp — host orchestration of the content-ingest path behind the C ABI:
// chunk boundaries (RabinChunker / StreamingChunker semantics), per-chunk SHA-256 and whole-blob
// SHA-256, i.e. the hash + chunk_file phases of ContentStore::store
// (src/api/content_store_impl.cpp:199-231 in the reference).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <string>
#include <vector>

#include "accel_ctx.h"
#include "ingest_launch.h"

using namespace yams_accel;

namespace {

constexpr uint64_t kDefaultPoly = 0x3DA3358B4DC173ULL; // rabin_fingerprint_table.h:12

yams_status_t make_params(yams_accel_ctx* ctx, const yams_cdc_config_t* cfg, CdcParams* cp) {
    if (!cfg) return fail(ctx, YAMS_ERR_INVALID_ARG, "null chunking config");
    if (cfg->mode != YAMS_CDC_RABIN && cfg->mode != YAMS_CDC_STREAMING)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "unknown chunker mode");
    cp->polynomial = cfg->polynomial ? cfg->polynomial : kDefaultPoly; // rabin_chunker.cpp:29-37
    cp->mask = cfg->mask;
    cp->min_size = cfg->min_size;
    cp->max_size = cfg->max_size;
    cp->streaming = cfg->mode == YAMS_CDC_STREAMING;
    cp->generic = (cfg->flags & YAMS_CDC_FLAG_GENERIC_KERNEL) ? 1u : 0u;
    cp->context = 0;
    uint64_t w = cfg->window_size;
    if (cp->streaming) { // streaming_chunker.cpp:44-49 clamps the ring
        if (w == 0) w = 1; else if (w > 48) w = 48;
    } else if (w == 0 || w > 48) {
        // RabinWindow is a fixed 48-byte ring (chunker.h:151-155); other sizes index out of it
        return fail(ctx, YAMS_ERR_INVALID_ARG, "RabinChunker windowSize must be in [1, 48]");
    }
    cp->window = static_cast<uint32_t>(w);
    if (std::max(cfg->min_size, cfg->max_size) == 0)
        return fail(ctx, YAMS_ERR_INVALID_ARG, "minChunkSize and maxChunkSize are both zero");
    return YAMS_OK;
}

// A chain lane (yams_ingest_host): the whole-blob digest chains of one batch run on a stream of their own and are
// NOT joined at the end of the call — the caller joins `join` when it needs the digests (or the batch's data buffer
// back), so the chains of a batch run beside the NEXT batches' uploads and kernels.  Their tables live in
// workspace buffers named after the lane.
struct IngestLane { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; int index = 0; };

yams_status_t ingest_impl(yams_accel_ctx* ctx, const uint8_t* data, const uint64_t* blob_off_h,
                          const uint64_t* blob_len_h, uint64_t n_blobs64,
                          const yams_cdc_config_t* cfg, uint32_t flags, bool do_chunks,
                          yams_ingest_result_t* out, const IngestLane* lane = nullptr, uint64_t defer_above = 0,
                          uint64_t context = 0) {
    if (!out) return fail(ctx, YAMS_ERR_INVALID_ARG, "null result");
    std::memset(out, 0, sizeof(*out));
    if (n_blobs64 >= (1ull << 31)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "too many blobs");
    const uint32_t n_blobs = static_cast<uint32_t>(n_blobs64);
    if (n_blobs && (!blob_off_h || !blob_len_h)) return fail(ctx, YAMS_ERR_INVALID_ARG, "null blob table");
    CdcParams cp{};
    if (do_chunks) YA_TRY(make_params(ctx, cfg, &cp));
    if (context) {
        // only the StreamingChunker carries its rolling hash across chunk boundaries (streaming_chunker.h:146-204);
        // RabinChunker starts every chunk afresh (rabin_chunker.cpp:63-110): history in front of a chunk means nothing there
        if (!do_chunks || !cp.streaming) return fail(ctx, YAMS_ERR_INVALID_ARG, "a context prefix needs the streaming chunker");
        cp.context = context;
    }
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;

    // ---- host metadata -----------------------------------------------------------------------
    std::vector<uint64_t> meta(static_cast<size_t>(n_blobs) * 2 + 2 * (static_cast<size_t>(n_blobs) + 1));
    uint64_t* h_off = meta.data();
    uint64_t* h_len = h_off + n_blobs;
    uint64_t* h_piece = h_len + n_blobs;        // [n_blobs + 1]
    uint64_t* h_slot = h_piece + n_blobs + 1;   // [n_blobs + 1]
    uint64_t pieces = 0, slots = 0, total_bytes = 0;
    const uint64_t min_eff = std::max<uint64_t>(cp.min_size, 1);
    for (uint32_t b = 0; b < n_blobs; ++b) {
        h_off[b] = blob_off_h[b]; h_len[b] = blob_len_h[b];
        if (h_len[b] && !data) return fail(ctx, YAMS_ERR_INVALID_ARG, "null data");
        h_piece[b] = pieces; h_slot[b] = slots;
        pieces += (h_len[b] + kCdcPiece - 1) / kCdcPiece;
        slots += h_len[b] / min_eff + 2;
        total_bytes += h_len[b];
    }
    h_piece[n_blobs] = pieces; h_slot[n_blobs] = slots;
    if (pieces >= (1ull << 31)) return fail(ctx, YAMS_ERR_UNSUPPORTED, "more than 2^31 pieces in one call");

    uint64_t* d_meta;
    YA_TRY(ws_get(ctx, "ing_meta", meta.size() * 8, (void**)&d_meta));
    if (!meta.empty())
        YA_HIP(ctx, hipMemcpyAsync(d_meta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, st));
    const uint64_t* d_off = d_meta;
    const uint64_t* d_len = d_off + n_blobs;
    const uint64_t* d_piece = d_len + n_blobs;
    const uint64_t* d_slot = d_piece + n_blobs + 1;

    uint64_t* d_blob_first; uint64_t* d_blob_count;
    YA_TRY(ws_get(ctx, "ing_blob_first", (static_cast<size_t>(n_blobs) + 1) * 8, (void**)&d_blob_first));
    YA_TRY(ws_get(ctx, "ing_blob_count", (static_cast<size_t>(n_blobs) + 1) * 8, (void**)&d_blob_count));

    // Whole-blob digests are long sequential chains (65 536 blocks for 4 MiB) that bound the call:
    // they start NOW on the high-priority side stream and run concurrently with boundary detection
    // and chunk hashing (which alone reach their VALU / HBM rooflines in a fraction of the time).
    const bool fork_blobs = (flags & YAMS_INGEST_BLOB_DIGESTS) && n_blobs > 0;
    uint8_t* d_blob_digests = nullptr;
    uint64_t* d_sorted_blobs = nullptr;
    // YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS: blobs above the threshold get NO whole-blob chain here (one chain is
    // sequential, ~35 MB/s on a device lane: a 64 MiB blob holds the whole call for 1.9 s while a host core hashes it in
    // 40 ms) — their digest entries are 32 zero bytes and the caller's host hasher fills them while the device works.
    if ((flags & YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS) && defer_above == 0) defer_above = yams_ingest_defer_threshold_device(total_bytes);
    if (!(flags & YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS)) defer_above = 0;
    uint32_t n_chains = n_blobs;
    if (fork_blobs) {
        // Lanes of one workgroup advance in lock step, so messages are grouped by length
        // (longest first); out_slot maps the sorted position back to the blob index.
        std::vector<uint32_t> order(n_blobs);
        for (uint32_t b = 0; b < n_blobs; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t x, uint32_t y) { return h_len[x] > h_len[y]; });
        std::vector<uint64_t> sorted(static_cast<size_t>(n_blobs) * 2 + (static_cast<size_t>(n_blobs) + 1) / 2);
        uint32_t* h_slot32 = reinterpret_cast<uint32_t*>(sorted.data() + static_cast<size_t>(n_blobs) * 2);
        if (defer_above) { // (sorted longest first: the deferred blobs are a prefix)
            uint32_t skip = 0;
            while (skip < n_blobs && h_len[order[skip]] > defer_above) ++skip;
            order.erase(order.begin(), order.begin() + skip);
            n_chains = n_blobs - skip;
        }
        for (uint32_t i = 0; i < n_chains; ++i) {
            sorted[i] = h_off[order[i]];
            sorted[static_cast<size_t>(n_blobs) + i] = h_len[order[i]];
            h_slot32[i] = order[i];
        }
        uint64_t* d_sorted;
        const std::string lane_tag = lane ? "_lane" + std::to_string(lane->index) : std::string();
        YA_TRY(ws_get(ctx, ("ing_meta_blobs" + lane_tag).c_str(), sorted.size() * 8, (void**)&d_sorted));
        YA_HIP(ctx, hipMemcpyAsync(d_sorted, sorted.data(), sorted.size() * 8, hipMemcpyHostToDevice, st));
        YA_HIP(ctx, hipStreamSynchronize(st)); // `sorted` is pageable and dies with this scope
        YA_TRY(ws_get(ctx, ("ing_blob_digests" + lane_tag).c_str(), static_cast<size_t>(n_blobs) * 32, (void**)&d_blob_digests));
        if (n_chains != n_blobs) YA_HIP(ctx, hipMemsetAsync(d_blob_digests, 0, static_cast<size_t>(n_blobs) * 32, st)); // (in front of the fork below)
        d_sorted_blobs = d_sorted;
    }
    // the side stream's launch: at once, or (`long_after_cdc`) behind boundary detection
    const hipStream_t aux = lane ? lane->stream : ctx->aux_stream;
    const hipEvent_t aux_fork = lane ? lane->fork : ctx->aux_fork, aux_join = lane ? lane->join : ctx->aux_join;
    auto fork_long = [&]() -> yams_status_t {
        YA_HIP(ctx, hipEventRecord(aux_fork, st));
        YA_HIP(ctx, hipStreamWaitEvent(aux, aux_fork, 0));
        TimedRegion tr(ctx, "sha256_blobs", aux);
        if (n_chains)
            YA_HIP(ctx, launch_sha256_long(aux, data, d_sorted_blobs, d_sorted_blobs + n_blobs,
                                           reinterpret_cast<const uint32_t*>(d_sorted_blobs + static_cast<size_t>(n_blobs) * 2),
                                           n_chains, d_blob_digests));
        tr.end();
        YA_HIP(ctx, hipEventRecord(aux_join, aux));
        return YAMS_OK;
    };
    bool long_after_cdc = false;
#ifdef YAMS_ACCEL_MEASURE
    if (const char* e = std::getenv("YAMS_ACCEL_INGEST_LONG_AFTER_CDC")) long_after_cdc = std::atoi(e) != 0;
#endif
    if (fork_blobs && !(long_after_cdc && do_chunks)) YA_TRY(fork_long());

    uint64_t n_chunks = 0;
    uint64_t* d_msg_off = nullptr; uint64_t* d_msg_len = nullptr;
    uint64_t* d_chunk_off = nullptr; uint64_t* d_chunk_size = nullptr; uint32_t* d_chunk_blob = nullptr;
    const bool want_chunk_dg = do_chunks && (flags & YAMS_INGEST_CHUNK_DIGESTS);
    const uint64_t msg_cap = static_cast<uint64_t>(n_blobs) + (do_chunks ? slots : 0);
    YA_TRY(ws_get(ctx, "ing_msg_off", msg_cap * 8, (void**)&d_msg_off));
    YA_TRY(ws_get(ctx, "ing_msg_len", msg_cap * 8, (void**)&d_msg_len));
    if (n_blobs) {
        // messages [0, n_blobs) are the whole blobs (longest first in the queue)
        YA_HIP(ctx, hipMemcpyAsync(d_msg_off, d_off, static_cast<size_t>(n_blobs) * 8, hipMemcpyDeviceToDevice, st));
        YA_HIP(ctx, hipMemcpyAsync(d_msg_len, d_len, static_cast<size_t>(n_blobs) * 8, hipMemcpyDeviceToDevice, st));
    }

    if (do_chunks) {
        uint32_t* d_bitmap; uint64_t* d_slot_off; uint64_t* d_slot_size;
        YA_TRY(ws_get(ctx, "ing_bitmap", pieces * (kCdcPiece / 32) * 4, (void**)&d_bitmap));
        YA_TRY(ws_get(ctx, "ing_slot_off", slots * 8, (void**)&d_slot_off));
        YA_TRY(ws_get(ctx, "ing_slot_size", slots * 8, (void**)&d_slot_size));
        YA_TRY(ws_get(ctx, "ing_chunk_off", slots * 8, (void**)&d_chunk_off));
        YA_TRY(ws_get(ctx, "ing_chunk_size", slots * 8, (void**)&d_chunk_size));
        YA_TRY(ws_get(ctx, "ing_chunk_blob", slots * 4, (void**)&d_chunk_blob));
        {
            TimedRegion tr(ctx, "cdc_candidates");
            YA_HIP(ctx, launch_cdc_candidates(st, data, d_off, d_len, d_piece, n_blobs, pieces, cp, d_bitmap));
            tr.end();
        }
        if (fork_blobs && long_after_cdc) YA_TRY(fork_long());
        {
            TimedRegion tr(ctx, "cdc_walk");
            YA_HIP(ctx, launch_cdc_walk(st, d_bitmap, d_len, d_piece, d_slot, n_blobs, cp, d_slot_off,
                                        d_slot_size, d_blob_count));
            tr.end();
        }
        YA_HIP(ctx, launch_chunk_compact(st, d_slot, d_slot_off, d_slot_size, d_blob_count,
                                         d_blob_first, d_off, n_blobs, d_chunk_off, d_chunk_size,
                                         d_chunk_blob, d_msg_off + n_blobs, d_msg_len + n_blobs));
        uint64_t* h_total;
        YA_TRY(pinned_get(ctx, 64, (void**)&h_total));
        YA_HIP(ctx, hipMemcpyAsync(h_total, d_blob_first + n_blobs, 8, hipMemcpyDeviceToHost, st));
        YA_HIP(ctx, hipStreamSynchronize(st));
        n_chunks = *h_total;
    }

    uint8_t* d_digests = nullptr;
    if (want_chunk_dg && n_chunks) {
        unsigned long long* d_head;
        YA_TRY(ws_get(ctx, "ing_queue", 64, (void**)&d_head));
        YA_TRY(ws_get(ctx, "ing_digests", static_cast<size_t>(n_chunks) * 32 + 32, (void**)&d_digests));
        TimedRegion tr(ctx, "sha256");
        YA_HIP(ctx, launch_sha256(st, data, d_msg_off + n_blobs, d_msg_len + n_blobs, 0, n_chunks,
                                  d_digests, d_head, nullptr, nullptr, 0, 4096, 1));
        tr.end();
    }
    if (fork_blobs && !lane) YA_HIP(ctx, hipStreamWaitEvent(st, aux_join, 0)); // join the side stream (a lane's caller joins later)
    out->n_chunks = n_chunks;
    out->chunk_offset = d_chunk_off;
    out->chunk_size = d_chunk_size;
    out->chunk_blob = d_chunk_blob;
    out->blob_first = do_chunks ? d_blob_first : nullptr;
    out->chunk_digest = want_chunk_dg ? d_digests : nullptr;
    out->blob_digest = fork_blobs ? d_blob_digests : nullptr;
    ctx->ingest = *out;
    return YAMS_OK;
}

void to_hex(const uint8_t* dg, char* out) { // bytesToHex, sha256_hasher.cpp:19-30
    static const char kHex[] = "0123456789abcdef";
    for (int i = 0; i < 32; ++i) { out[2 * i] = kHex[dg[i] >> 4]; out[2 * i + 1] = kHex[dg[i] & 15]; }
    out[64] = 0;
}

} // namespace

extern "C" {

void yams_cdc_default_config(yams_cdc_config_t* cfg, uint32_t mode) {
    if (!cfg) return;
    cfg->window_size = 48;              // chunker.h:45
    cfg->min_size = 16 * 1024;          // MIN_CHUNK_SIZE, core/types.h:282
    cfg->max_size = 1024 * 1024;        // MAX_CHUNK_SIZE, core/types.h:284
    cfg->polynomial = kDefaultPoly;     // chunker.h:49
    cfg->mask = 0x1FFF;                 // chunker.h:50
    cfg->mode = mode;
    cfg->flags = 0;
}

yams_status_t yams_cdc_chunk_device(yams_accel_ctx* ctx, const uint8_t* data,
                                    const uint64_t* blob_offsets_host,
                                    const uint64_t* blob_lengths_host, uint64_t n_blobs,
                                    const yams_cdc_config_t* cfg, yams_ingest_result_t* out) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    return ingest_impl(ctx, data, blob_offsets_host, blob_lengths_host, n_blobs, cfg, 0, true, out);
}

yams_status_t yams_ingest_device(yams_accel_ctx* ctx, const uint8_t* data,
                                 const uint64_t* blob_offsets_host,
                                 const uint64_t* blob_lengths_host, uint64_t n_blobs,
                                 const yams_cdc_config_t* cfg, uint32_t flags,
                                 yams_ingest_result_t* out) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    return ingest_impl(ctx, data, blob_offsets_host, blob_lengths_host, n_blobs, cfg, flags, true, out);
}

// Host-sourced ingest: the blobs live in host memory (files just read, network buffers).  They cross
// PCIe in batches through two device buffers: while the kernels work on batch i, batch i + 1 is
// already on its way on a copy stream of its own; the results of a batch (a few bytes per chunk) go
// back before the next one starts.  Small blobs: bound by the link.  Large blobs: a batch cannot end
// before the SHA-256 chain of its longest blob has (one lane, ~35 MB/s), so batches are GiB-sized.
yams_status_t yams_ingest_host(yams_accel_ctx* ctx, const uint8_t* const* blobs_host,
                               const uint64_t* blob_lengths, uint64_t n_blobs,
                               const yams_cdc_config_t* cfg, uint32_t flags, uint64_t batch_bytes,
                               uint64_t* out_blob_first, uint64_t* out_chunk_offset,
                               uint64_t* out_chunk_size, uint8_t* out_chunk_digest, uint64_t chunk_cap,
                               uint8_t* out_blob_digest, uint64_t* out_n_chunks) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (out_n_chunks) *out_n_chunks = 0;
    if (n_blobs && (!blobs_host || !blob_lengths)) return fail(ctx, YAMS_ERR_INVALID_ARG, "null blob list");
    if (!out_blob_first) return fail(ctx, YAMS_ERR_INVALID_ARG, "null blob_first");
    if ((flags & YAMS_INGEST_BLOB_DIGESTS) && n_blobs && !out_blob_digest) return fail(ctx, YAMS_ERR_INVALID_ARG, "null blob digests");
    CdcParams cp{};
    YA_TRY(make_params(ctx, cfg, &cp)); // (before any allocation: bad configurations fail like the device entry point)
    (void)hipSetDevice(ctx->device);
    out_blob_first[0] = 0;
    if (n_blobs == 0) return YAMS_OK;
    if (batch_bytes == 0) {
        // A batch's whole-blob digest chains take (longest blob) / 35 MB/s whatever the batch holds, and the chains of up
        // to three batches are in flight: the stream moves at 3 x batch_bytes per chain time until the link takes over.
        // Measured on 4 MiB blobs, 32 GiB per call (scripts/dbg/host_stream_batches.py): 1 / 2 / 4 / 8 GiB batches =
        // 17.5 / 32 / 41 / 46 GB/s.  So: about 2048 of the longest blob per batch, between 1 and 8 GiB, and never fewer
        // than four batches per call (the upload of one must run under the kernels of another).
        uint64_t longest = 0, total = 0;
        for (uint64_t b = 0; b < n_blobs; ++b) { longest = std::max<uint64_t>(longest, blob_lengths[b]); total += blob_lengths[b]; }
        if (!(flags & YAMS_INGEST_BLOB_DIGESTS)) longest = 0; // no chains: the link is the only bound
        else if (flags & YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS) longest = std::min(longest, yams_ingest_defer_threshold_host(total));
        const uint64_t cap = std::max<uint64_t>(1ull << 30, std::min<uint64_t>(8ull << 30, total / 4));
        batch_bytes = std::min(cap, std::max<uint64_t>(1ull << 30, longest * 2048));
        // ... within what the device has to spare: four slot buffers + the per-batch tables (about a quarter more) may take
        // half of the memory that is free NOW — a device that holds a 67 GB mirror still has room, one that is nearly
        // full gets smaller batches (slower, not an out-of-memory failure of this call or of the next corpus_append)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) {
            free_b += big_held(ctx->device);    // (what the pool of call-sized buffers holds is this call's to use)
            const uint64_t per_slot = static_cast<uint64_t>(free_b) / 2 / 5;
            batch_bytes = std::min<uint64_t>(batch_bytes, std::max<uint64_t>(256ull << 20, per_slot));
        } else (void)hipGetLastError();
    }
    // batches of consecutive blobs (a blob never straddles two: its digest is one chain); every blob
    // starts on a 16-byte boundary of the device buffer
    struct Batch { uint64_t first, count, bytes; };
    std::vector<Batch> batches;
    uint64_t largest = 0;
    for (uint64_t b = 0; b < n_blobs;) {
        Batch bt{b, 0, 0};
        while (b < n_blobs) {
            if (blob_lengths[b] && !blobs_host[b]) return fail(ctx, YAMS_ERR_INVALID_ARG, "null blob");
            const uint64_t padded = (blob_lengths[b] + 15) & ~15ull;
            if (bt.count && bt.bytes + padded > batch_bytes) break;
            bt.bytes += padded; ++bt.count; ++b;
        }
        largest = std::max(largest, bt.bytes);
        batches.push_back(bt);
    }
    // Device buffers: two when there are no whole-blob digests (batch i + 1 travels while batch i is worked on); up to
    // kSlots with them, because a batch's digest chains (one lane per blob, ~35 MB/s: 120 ms for a 4 MiB blob) outlast
    // its upload and its other kernels many times over — each batch gets a chain lane (a stream of its own) and is
    // joined only when its buffer is needed again, kSlots - 1 batches later, so the chains of up to three batches run
    // beside each other and beside the uploads.  (One buffer pair, joined per batch: 2 GiB per 120 ms = 17 GB/s on 4 MiB
    // blobs however fast the link.)
    constexpr int kSlots = 4;
    const bool chains = (flags & YAMS_INGEST_BLOB_DIGESTS) != 0;
    // the deferral threshold is a property of the whole call (the caller computes the same one from the same lengths)
    uint64_t defer_above = 0;
    if (chains && (flags & YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS)) {
        uint64_t total = 0;
        for (uint64_t b = 0; b < n_blobs; ++b) total += blob_lengths[b];
        defer_above = yams_ingest_defer_threshold_host(total);
    }
    const int n_slots = static_cast<int>(std::min<size_t>(chains ? kSlots : 2, batches.size()));
    // The slot buffers belong to the CALL, not to the context: they come from the process-wide pool of call-sized buffers
    // (accel_ctx.h: big_take / big_give) and go back to it on every exit path — a second call finds them there instead of
    // paying the driver for 32 GiB again (round 5: the same stream measured 9 and 46 GB/s; the difference was this).
    uint8_t* d_buf[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t d_cap[kSlots] = {0, 0, 0, 0};
    timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto ms_since = [](const timespec& a) { timespec b; clock_gettime(CLOCK_MONOTONIC, &b); return (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6; };
    auto& stats = ctx->host_ingest;
    stats = {};
    auto give_back = [&]() {
        timespec tr; clock_gettime(CLOCK_MONOTONIC, &tr);
        for (int i = 0; i < kSlots; ++i) { if (d_buf[i]) big_give(ctx->device, d_buf[i], d_cap[i]); d_buf[i] = nullptr; }
        stats.release_ms += ms_since(tr);
    };
    for (int i = 0; i < n_slots; ++i) {
        void* p = nullptr;
        const size_t before_cap = d_cap[i];
        (void)before_cap;
        const hipError_t e = big_take(ctx->device, largest + 64, &p, &d_cap[i]);
        if (e != hipSuccess) {
            // slot i could not be had: the slots before it go back to the pool — which ya_malloc empties before it gives up
            (void)hipGetLastError();
            give_back();
            return hip_fail(ctx, e, "device buffers of the host-streamed ingest");
        }
        d_buf[i] = static_cast<uint8_t*>(p);
    }
    stats.alloc_ms = ms_since(ts0);
    stats.slots = static_cast<uint32_t>(n_slots); stats.batches = static_cast<uint32_t>(batches.size()); stats.batch_bytes = largest;
    for (const Batch& bt : batches) stats.bytes += bt.bytes;
    hipStream_t copy_st = nullptr;
    hipEvent_t landed[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    IngestLane lanes[kSlots];
    yams_status_t rc = YAMS_OK;
    auto cleanup = [&]() {
        // Footprint: the slot buffers and batch tables of GiB-sized batches (up to 4 x 8 GiB + tables for 4 MiB blobs with
        // whole-blob digests) belong to the CALL — a context that kept them (workspaces only ever grow) would sit on
        // ~36 GiB for good, in the plugin one such share per pooled context, next to mirrors sized as shares of the
        // device.  What stays with the context after the call: buffers up to 1 GiB + headroom, as before round 4.
        struct Trim { yams_accel_ctx* c; bool on; ~Trim() { if (on) (void)ws_trim(c, (1ull << 30) + (1ull << 28)); } } trim{ctx, largest > (1ull << 30)};
        if (copy_st) { (void)hipStreamSynchronize(copy_st); (void)hipStreamDestroy(copy_st); }
        (void)hipStreamSynchronize(ctx->stream);    // nothing of this call may still read a slot buffer when it goes back to the pool
        for (hipEvent_t e : landed) if (e) (void)hipEventDestroy(e);
        for (IngestLane& l : lanes) {
            if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
            if (l.fork) (void)hipEventDestroy(l.fork);
            if (l.join) (void)hipEventDestroy(l.join);
        }
        give_back();
        stats.total_ms = ms_since(ts0);
    };
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        rc = hip_fail(ctx, e, what);    // (hipErrorOutOfMemory -> YAMS_ERR_RESOURCE_EXHAUSTED, everything else internal)
        return false;
    };
    // Streams of one priority share a small pool of hardware queues (four by default), handed out by how many streams
    // already sit on each: in a process that has created many streams the copy stream, the chain lanes and the
    // context's stream can land on ONE hardware queue, and then the next upload waits behind a 120 ms chain kernel
    // (measured inside bench.py's process: 22 GB/s for a stream that does 46 on its own).  The three roles therefore
    // live in three priority classes, whose queue pools are disjoint: uploads high, kernels normal (the context's
    // stream), chains low — which is also what they are: background work that must not delay anything.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (!hip_ok(hipStreamCreateWithPriority(&copy_st, hipStreamNonBlocking, prio_hi), "copy stream")) { cleanup(); return rc; }
    for (int i = 0; i < n_slots; ++i) {
        lanes[i].index = i;
        if (!hip_ok(hipEventCreateWithFlags(&landed[i], hipEventDisableTiming), "event") ||
            (chains && (!hip_ok(hipStreamCreateWithPriority(&lanes[i].stream, hipStreamNonBlocking, prio_lo), "chain stream") ||
                        !hip_ok(hipEventCreateWithFlags(&lanes[i].fork, hipEventDisableTiming), "event") ||
                        !hip_ok(hipEventCreateWithFlags(&lanes[i].join, hipEventDisableTiming), "event")))) { cleanup(); return rc; }
    }
    std::vector<uint64_t> offs, lens;
    auto upload = [&](size_t bi) -> bool { // batch bi -> its slot's device buffer, on the copy stream
        const Batch& bt = batches[bi];
        uint8_t* const dst = d_buf[bi % n_slots];
        // blobs that are neighbours in host memory (one mapped file, one receive buffer) and in the device
        // buffer travel as ONE copy: a copy call costs the host ~10 us whatever its size
        uint64_t at = 0, run_dst = 0, run_len = 0;
        const uint8_t* run_src = nullptr;
        auto flush = [&]() -> bool {
            const bool ok = run_len == 0 || hip_ok(hipMemcpyAsync(dst + run_dst, run_src, run_len, hipMemcpyHostToDevice, copy_st), "upload");
            run_len = 0;
            return ok;
        };
        for (uint64_t j = 0; j < bt.count; ++j) {
            const uint64_t n = blob_lengths[bt.first + j];
            const uint8_t* src = blobs_host[bt.first + j];
            if (n) {
                if (run_len && src == run_src + run_len && at == run_dst + run_len) {
                    run_len += n;
                } else {
                    if (!flush()) return false;
                    run_src = src; run_dst = at; run_len = n;
                }
            }
            at += (n + 15) & ~15ull;
        }
        if (!flush()) return false;
        return hip_ok(hipEventRecord(landed[bi % n_slots], copy_st), "event record");
    };
    // a batch whose chains are still running: what is needed to fetch its blob digests later
    struct Pending { bool open = false; const uint8_t* d_digests = nullptr; uint64_t first = 0, count = 0; };
    Pending pending[kSlots];
    auto settle = [&](int slot) -> bool { // join the slot's chains, bring their digests home; its buffer is free afterwards
        Pending& pd = pending[slot];
        if (!pd.open) return true;
        pd.open = false;
        hipStream_t st = ctx->stream;
        if (!hip_ok(hipStreamWaitEvent(st, lanes[slot].join, 0), "join")) return false;
        unsigned char* stage = nullptr;
        const bool take = out_blob_digest && pd.d_digests;
        if (take && (rc = pinned_get(ctx, pd.count * 32 + 64, (void**)&stage)) != YAMS_OK) return false;
        if (take && !hip_ok(hipMemcpyAsync(stage, pd.d_digests, pd.count * 32, hipMemcpyDeviceToHost, st), "results")) return false;
        if (!hip_ok(hipStreamSynchronize(st), "sync")) return false;
        if (take) std::memcpy(out_blob_digest + pd.first * 32, stage, pd.count * 32);
        return true;
    };
    uint64_t chunk_base = 0;
    bool too_small = false;
#ifdef YAMS_ACCEL_MEASURE
    const bool trace = std::getenv("YAMS_ACCEL_INGEST_TRACE") != nullptr; // host-side phase times of every batch, to stderr
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
#define YAMS_TRACE_T(var) const double var = trace ? now_ms() : 0.0
#else
#define YAMS_TRACE_T(var) do {} while (0)
#endif
    if (!upload(0)) { cleanup(); return rc; }
    for (size_t bi = 0; bi < batches.size() && rc == YAMS_OK; ++bi) {
        const Batch& bt = batches[bi];
        const int slot = static_cast<int>(bi % n_slots);
        YAMS_TRACE_T(t0);
        // the buffer batch bi + 1 will go into was batch bi + 1 - n_slots's: its kernels and result copies have completed
        // (every batch ends with a synchronisation of the context's stream); its chains are joined here, while the
        // stream is idle
        if (bi + 1 < batches.size() && !settle(static_cast<int>((bi + 1) % n_slots))) break;
        YAMS_TRACE_T(t1);
        if (!hip_ok(hipStreamWaitEvent(ctx->stream, landed[slot], 0), "wait")) break;
        offs.resize(bt.count); lens.resize(bt.count);
        uint64_t at = 0;
        for (uint64_t j = 0; j < bt.count; ++j) {
            offs[j] = at; lens[j] = blob_lengths[bt.first + j];
            at += (lens[j] + 15) & ~15ull;
        }
        yams_ingest_result_t r;
        rc = ingest_impl(ctx, d_buf[slot], offs.data(), lens.data(), bt.count, cfg, flags, true, &r, chains ? &lanes[slot] : nullptr, defer_above);
        if (rc != YAMS_OK) break;
        YAMS_TRACE_T(t1b);
        // Batch bi + 1 starts its journey only NOW, behind this batch's small table uploads: copies of all streams
        // share the DMA engines first come first served, and issued up front the 512 MiB of the next batch stood in
        // front of a few KiB of tables for 9 ms per batch (host-side trace, YAMS_ACCEL_INGEST_TRACE).
        if (bi + 1 < batches.size() && !upload(bi + 1)) break;
        YAMS_TRACE_T(t2);
        hipStream_t st = ctx->stream;
        // results: one pinned staging area, then plain copies into the caller's (pageable) arrays — four asynchronous
        // copies into pageable memory were four blocking round trips of ~1 ms each
        if (chunk_base + r.n_chunks > chunk_cap || (r.n_chunks && (!out_chunk_offset || !out_chunk_size))) too_small = true;
        const bool take = !too_small && r.n_chunks;
        const bool take_dg = take && out_chunk_digest && r.chunk_digest;
        const size_t b_first = (bt.count + 1) * 8, b_tab = take ? r.n_chunks * 8 : 0, b_dg = take_dg ? r.n_chunks * 32 : 0;
        unsigned char* stage;
        if ((rc = pinned_get(ctx, b_first + 2 * b_tab + b_dg + 64, (void**)&stage)) != YAMS_OK) break;
        unsigned char* s_first = stage; unsigned char* s_off = s_first + b_first; unsigned char* s_size = s_off + b_tab; unsigned char* s_dg = s_size + b_tab;
        if (!hip_ok(hipMemcpyAsync(s_first, r.blob_first, b_first, hipMemcpyDeviceToHost, st), "results")) break;
        if (take && (!hip_ok(hipMemcpyAsync(s_off, r.chunk_offset, b_tab, hipMemcpyDeviceToHost, st), "results") ||
                     !hip_ok(hipMemcpyAsync(s_size, r.chunk_size, b_tab, hipMemcpyDeviceToHost, st), "results"))) break;
        if (take_dg && !hip_ok(hipMemcpyAsync(s_dg, r.chunk_digest, b_dg, hipMemcpyDeviceToHost, st), "results")) break;
        if (chains) { pending[slot].open = true; pending[slot].d_digests = r.blob_digest; pending[slot].first = bt.first; pending[slot].count = bt.count; }
        YAMS_TRACE_T(t3);
        if (!hip_ok(hipStreamSynchronize(st), "sync")) break;
        const uint64_t* first = reinterpret_cast<const uint64_t*>(s_first);
        for (uint64_t j = 0; j <= bt.count; ++j) out_blob_first[bt.first + j] = chunk_base + first[j];
        if (take) {
            std::memcpy(out_chunk_offset + chunk_base, s_off, b_tab);
            std::memcpy(out_chunk_size + chunk_base, s_size, b_tab);
            if (take_dg) std::memcpy(out_chunk_digest + chunk_base * 32, s_dg, b_dg);
        }
        chunk_base += r.n_chunks;
#ifdef YAMS_ACCEL_MEASURE
        if (trace) std::fprintf(stderr, "batch %zu: settle %.2f ms, ingest_impl %.2f ms, next upload issued %.2f ms, result copies issued %.2f ms, sync + copy out %.2f ms\n", bi, t1 - t0, t1b - t1, t2 - t1b, t3 - t2, now_ms() - t3);
#endif
    }
#undef YAMS_TRACE_T
    for (int i = 0; i < n_slots && rc == YAMS_OK; ++i)
        if (!settle(i)) break;
    cleanup();
    if (rc != YAMS_OK) return rc;
    if (out_n_chunks) *out_n_chunks = chunk_base;
    if (too_small) return fail(ctx, YAMS_ERR_INVALID_ARG, "chunk arrays too small (out_n_chunks holds the required size)");
    return YAMS_OK;
}

yams_status_t yams_sha256_batch_device(yams_accel_ctx* ctx, const uint8_t* data,
                                       const uint64_t* offsets, const uint64_t* lengths,
                                       uint64_t n_msgs, uint8_t* digests) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (n_msgs == 0) return YAMS_OK;
    if (!offsets || !lengths || !digests) return fail(ctx, YAMS_ERR_INVALID_ARG, "null message table");
    (void)hipSetDevice(ctx->device);
    unsigned long long* d_head;
    YA_TRY(ws_get(ctx, "ing_queue", 64, (void**)&d_head));
    TimedRegion tr(ctx, "sha256");
    YA_HIP(ctx, launch_sha256(ctx->stream, data, offsets, lengths, 0, n_msgs, digests, d_head, nullptr,
                              nullptr, 0, 4096, 2));
    tr.end();
    return YAMS_OK;
}

yams_status_t yams_verify_chunks_device(yams_accel_ctx* ctx, const uint8_t* data,
                                        const uint64_t* offsets, const uint64_t* lengths,
                                        uint64_t n_chunks, const uint8_t* expected_digests,
                                        uint8_t* out_valid, uint64_t* out_n_invalid) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (out_n_invalid) *out_n_invalid = 0;
    if (n_chunks == 0) return YAMS_OK;
    if (!expected_digests || !out_valid) return fail(ctx, YAMS_ERR_INVALID_ARG, "null expected digests / out_valid");
    (void)hipSetDevice(ctx->device);
    uint8_t* d_actual; unsigned long long* d_bad;
    YA_TRY(ws_get(ctx, "verify_digests", n_chunks * 32, (void**)&d_actual));
    YA_TRY(ws_get(ctx, "verify_count", 64, (void**)&d_bad));
    YA_TRY(yams_sha256_batch_device(ctx, data, offsets, lengths, n_chunks, d_actual));
    YA_HIP(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    YA_HIP(ctx, launch_digest_compare(ctx->stream, d_actual, expected_digests, n_chunks, out_valid, d_bad));
    unsigned long long h = 0;
    YA_HIP(ctx, hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    YA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (out_n_invalid) *out_n_invalid = h;
    return YAMS_OK;
}

yams_status_t yams_sha256_many_host(yams_accel_ctx* ctx, const uint8_t* const* msgs_host,
                                    const size_t* lens, size_t n_msgs, char* out_hex) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (n_msgs == 0) return YAMS_OK;
    if (!msgs_host || !lens || !out_hex) return fail(ctx, YAMS_ERR_INVALID_ARG, "null message list");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<uint64_t> table(n_msgs * 2);
    uint64_t total = 0;
    for (size_t i = 0; i < n_msgs; ++i) {
        if (lens[i] && !msgs_host[i]) return fail(ctx, YAMS_ERR_INVALID_ARG, "null message");
        table[i] = total; table[n_msgs + i] = lens[i];
        total += (lens[i] + 15) & ~static_cast<uint64_t>(15); // keep messages 16-byte aligned
    }
    uint8_t* d_data; uint64_t* d_table; uint8_t* d_dg;
    YA_TRY(ws_get(ctx, "sha_data", total + 64, (void**)&d_data));
    YA_TRY(ws_get(ctx, "sha_table", table.size() * 8, (void**)&d_table));
    YA_TRY(ws_get(ctx, "sha_dg", n_msgs * 32, (void**)&d_dg));
    for (size_t i = 0; i < n_msgs; ++i)
        if (lens[i])
            YA_HIP(ctx, hipMemcpyAsync(d_data + table[i], msgs_host[i], lens[i], hipMemcpyHostToDevice, st));
    YA_HIP(ctx, hipMemcpyAsync(d_table, table.data(), table.size() * 8, hipMemcpyHostToDevice, st));
    YA_TRY(yams_sha256_batch_device(ctx, d_data, d_table, d_table + n_msgs, n_msgs, d_dg));
    std::vector<uint8_t> dg(n_msgs * 32);
    YA_HIP(ctx, hipMemcpyAsync(dg.data(), d_dg, dg.size(), hipMemcpyDeviceToHost, st));
    YA_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < n_msgs; ++i) to_hex(dg.data() + 32 * i, out_hex + 65 * i);
    return YAMS_OK;
}

yams_status_t yams_sha256_host(yams_accel_ctx* ctx, const uint8_t* data_host, size_t n,
                               char out_hex[65]) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!out_hex || (n && !data_host)) return fail(ctx, YAMS_ERR_INVALID_ARG, "null buffer");
    const uint8_t* msgs[1] = {data_host};
    const size_t lens[1] = {n};
    return yams_sha256_many_host(ctx, msgs, lens, 1, out_hex);
}

yams_status_t yams_cdc_chunk_host(yams_accel_ctx* ctx, const uint8_t* data_host, size_t n,
                                  const yams_cdc_config_t* cfg, uint64_t* offsets, uint64_t* sizes,
                                  char* hex, size_t cap, size_t* out_count) {
    return yams_cdc_chunk_window_host(ctx, data_host, n, 0, cfg, offsets, sizes, hex, cap, out_count);
}

yams_status_t yams_cdc_chunk_window_host(yams_accel_ctx* ctx, const uint8_t* data_host, size_t n, size_t context_len,
                                         const yams_cdc_config_t* cfg, uint64_t* offsets, uint64_t* sizes,
                                         char* hex, size_t cap, size_t* out_count) {
    if (!ctx) return YAMS_ERR_INVALID_ARG;
    if (!out_count || (n && !data_host)) return fail(ctx, YAMS_ERR_INVALID_ARG, "null buffer");
    if (context_len > n) return fail(ctx, YAMS_ERR_INVALID_ARG, "context_len exceeds the buffer");
    *out_count = 0;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    uint8_t* d_data;
    YA_TRY(ws_get(ctx, "cdc_host_data", n + 64, (void**)&d_data));
    if (n) YA_HIP(ctx, hipMemcpyAsync(d_data, data_host, n, hipMemcpyHostToDevice, st));
    const uint64_t off0 = 0, len0 = n;
    yams_ingest_result_t r;
    // "Empty input produces no chunks" (tests/unit/chunking/chunking_test.cpp:108-113)
    YA_TRY(ingest_impl(ctx, d_data, &off0, &len0, n ? 1 : 0, cfg, hex ? YAMS_INGEST_CHUNK_DIGESTS : 0,
                       true, &r, nullptr, 0, context_len));
    *out_count = static_cast<size_t>(r.n_chunks);
    if (r.n_chunks > cap) return fail(ctx, YAMS_ERR_INVALID_ARG, "chunk arrays too small");
    if (r.n_chunks == 0) return YAMS_OK;
    if (offsets) YA_HIP(ctx, hipMemcpyAsync(offsets, r.chunk_offset, r.n_chunks * 8, hipMemcpyDeviceToHost, st));
    if (sizes) YA_HIP(ctx, hipMemcpyAsync(sizes, r.chunk_size, r.n_chunks * 8, hipMemcpyDeviceToHost, st));
    std::vector<uint8_t> dg;
    if (hex) {
        dg.resize(r.n_chunks * 32);
        YA_HIP(ctx, hipMemcpyAsync(dg.data(), r.chunk_digest, dg.size(), hipMemcpyDeviceToHost, st));
    }
    YA_HIP(ctx, hipStreamSynchronize(st));
    if (hex) for (uint64_t i = 0; i < r.n_chunks; ++i) to_hex(dg.data() + 32 * i, hex + 65 * i);
    return YAMS_OK;
}

} // extern "C"
