// dedup_launch.h — host-visible descriptors for dedup_kernels.hip.
#pragma once
#include "common.h"

namespace yams_accel {

struct DedupTable {
    uint64_t* tags;   // [capacity] first 8 digest bytes (0 = empty slot)
    uint64_t* keys;   // [capacity][4] full digests
    uint32_t* owner;  // [capacity] lowest input index that touched the slot in the running batch
    uint8_t* fresh;   // [capacity] 1 = claimed by the running batch
    uint32_t capacity; // power of two
};

hipError_t launch_dedup_round(hipStream_t st, const DedupTable& t, const uint64_t* digests, uint32_t n,
                              uint8_t* pending, uint32_t* probe_start, uint32_t* slot_of, uint8_t* is_new,
                              unsigned int* n_unresolved, int first_round);
hipError_t launch_dedup_settle(hipStream_t st, const DedupTable& t, uint32_t n, const uint32_t* slot_of,
                               const uint8_t* is_new, unsigned long long* count);
hipError_t launch_dedup_probe(hipStream_t st, const DedupTable& t, const uint64_t* digests, uint32_t n, uint8_t* exists);
hipError_t launch_dedup_rehash(hipStream_t st, const DedupTable& old_t, const DedupTable& new_t);
hipError_t launch_dedup_fill_owner(hipStream_t st, uint32_t* owner, uint32_t n);
hipError_t launch_dedup_bytes(hipStream_t st, const uint8_t* is_new, const uint64_t* sizes, uint32_t n,
                              unsigned long long* out2);

} // namespace yams_accel
