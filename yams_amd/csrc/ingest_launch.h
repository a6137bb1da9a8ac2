// ingest_launch.h — host-visible launch descriptors for ingest_kernels.hip.
#pragma once
#include "common.h"

namespace yams_accel {

constexpr int kCdcPiece = 32768; // bytes of one blob handled by one candidate workgroup

struct CdcParams {
    uint64_t polynomial;
    uint64_t mask;
    uint64_t min_size;
    uint64_t max_size;
    uint32_t window;    // effective ring size (1..48)
    uint32_t streaming; // 1 = StreamingChunker semantics, 0 = RabinChunker
    uint32_t generic;   // 1 = use the any-window candidates kernel even where the narrow one applies
    uint64_t context;   // leading bytes of every blob that are HISTORY only: they feed the rolling hash, the first chunk starts behind them
};

hipError_t launch_cdc_candidates(hipStream_t st, const uint8_t* data, const uint64_t* blob_off,
                                 const uint64_t* blob_len, const uint64_t* piece_prefix,
                                 uint32_t n_blobs, uint64_t n_pieces, const CdcParams& cp,
                                 uint32_t* bitmap);
hipError_t launch_cdc_walk(hipStream_t st, const uint32_t* bitmap, const uint64_t* blob_len,
                           const uint64_t* piece_prefix, const uint64_t* slot_prefix,
                           uint32_t n_blobs, const CdcParams& cp, uint64_t* slot_off,
                           uint64_t* slot_size, uint64_t* blob_count);
hipError_t launch_chunk_compact(hipStream_t st, const uint64_t* slot_prefix,
                                const uint64_t* slot_off, const uint64_t* slot_size,
                                const uint64_t* blob_count, uint64_t* blob_first,
                                const uint64_t* blob_off, uint32_t n_blobs, uint64_t* chunk_offset,
                                uint64_t* chunk_size, uint32_t* chunk_blob, uint64_t* msg_off,
                                uint64_t* msg_len);
// Messages [0, n_long) are long (whole blobs), the rest short; queue_heads is 2 x u64 of scratch.
hipError_t launch_sha256(hipStream_t st, const uint8_t* data, const uint64_t* offs,
                         const uint64_t* lens, uint64_t n_long, uint64_t n_msgs, uint8_t* digests,
                         unsigned long long* queue_heads, const uint32_t* init_state,
                         uint32_t* out_state, int raw_blocks_only, uint32_t max_blocks, int slots);
// Long messages: two waves (schedule producer + round consumer) per 64 messages; digest of
// message i is written to digests[32 * (out_slot ? out_slot[i] : i)].
hipError_t launch_sha256_long(hipStream_t st, const uint8_t* data, const uint64_t* offs,
                              const uint64_t* lens, const uint32_t* out_slot, uint64_t n_msgs,
                              uint8_t* digests);
hipError_t launch_digest_compare(hipStream_t st, const uint8_t* actual, const uint8_t* expected, uint64_t n,
                                 uint8_t* valid, unsigned long long* n_invalid);

} // namespace yams_accel
