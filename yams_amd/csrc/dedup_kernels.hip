// dedup_kernels.hip — device-resident set of SHA-256 digests for the chunk dedup lookup.
//
// Reference semantics (paths under /root/reference): ContentStore::store walks the chunk list in
// order and asks storage_->exists(chunk.hash) for every chunk (src/api/content_store_impl.cpp:
// 246-287): an existing chunk only gets a reference increment, a new one is stored — and is found
// by the exists() of any later chunk with the same hash.  So chunk i is NEW iff its digest is
// neither in the store nor carried by an earlier chunk of the same walk.  Here: an open-addressing
// table keyed by the full 32-byte digest; a batch insert answers is_new[i] with exactly that rule
// (first occurrence = lowest index, made deterministic with atomicMin), in three race-free steps:
//   claim   every digest finds the slot that carries its 64-bit tag or claims an empty one (CAS)
//   write   the lowest-index claimer of a fresh slot writes the full key
//   verify  every digest compares its 32 bytes with the slot's key; a tag collision with a
//           different key (2^-64 per pair) resumes probing one slot further in another round.
// Byte/integer work, HBM-latency bound (one random 64-byte slot per digest); no MFMA.
#include "common.h"
#include "dedup_launch.h"

namespace yams_accel {

__device__ __forceinline__ uint64_t digest_tag(const uint64_t* d) {
    const uint64_t t = d[0];
    return t == 0 ? 1ull : t; // 0 marks an empty slot
}

__global__ __launch_bounds__(256) void dedup_claim_kernel(DedupTable t, const uint64_t* digests, uint32_t n,
                                                          const uint8_t* pending, uint32_t* probe_start,
                                                          uint32_t* slot_of, int first_round) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (!first_round && !pending[i])) return;
    const uint64_t* d = digests + 4ull * i;
    const uint64_t tag = digest_tag(d);
    const uint32_t mask = t.capacity - 1u;
    uint32_t s = first_round ? static_cast<uint32_t>((d[1] ^ (d[0] >> 17)) & mask) : probe_start[i];
    for (;;) {
        unsigned long long* tp = reinterpret_cast<unsigned long long*>(&t.tags[s]);
        unsigned long long cur = *reinterpret_cast<volatile unsigned long long*>(tp);
        if (cur == 0ull) {
            cur = atomicCAS(tp, 0ull, static_cast<unsigned long long>(tag));
            if (cur == 0ull) { t.fresh[s] = 1; cur = tag; }
        }
        if (cur == tag) {
            atomicMin(&t.owner[s], i);
            slot_of[i] = s;
            return;
        }
        s = (s + 1u) & mask;
    }
}

__global__ __launch_bounds__(256) void dedup_write_kernel(DedupTable t, const uint64_t* digests, uint32_t n,
                                                          const uint8_t* pending, const uint32_t* slot_of,
                                                          int first_round) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (!first_round && !pending[i])) return;
    const uint32_t s = slot_of[i];
    if (t.fresh[s] == 1 && t.owner[s] == i) {
        const uint64_t* d = digests + 4ull * i;
        uint64_t* k = t.keys + 4ull * s;
        k[0] = d[0]; k[1] = d[1]; k[2] = d[2]; k[3] = d[3];
    }
}

__global__ __launch_bounds__(256) void dedup_verify_kernel(DedupTable t, const uint64_t* digests, uint32_t n,
                                                           uint8_t* pending, uint32_t* probe_start,
                                                           const uint32_t* slot_of, uint8_t* is_new,
                                                           unsigned int* n_unresolved, int first_round) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (!first_round && !pending[i])) return;
    const uint32_t s = slot_of[i];
    const uint64_t* d = digests + 4ull * i;
    const uint64_t* k = t.keys + 4ull * s;
    if (k[0] == d[0] && k[1] == d[1] && k[2] == d[2] && k[3] == d[3]) {
        is_new[i] = (t.fresh[s] == 1 && t.owner[s] == i) ? 1 : 0;
        pending[i] = 0;
    } else { // same tag, different digest: keep probing behind this slot
        pending[i] = 1;
        probe_start[i] = (s + 1u) & (t.capacity - 1u);
        atomicAdd(n_unresolved, 1u);
    }
}

// After the last round: slots claimed by this batch become ordinary entries.
__global__ __launch_bounds__(256) void dedup_settle_kernel(DedupTable t, uint32_t n, const uint32_t* slot_of,
                                                           const uint8_t* is_new, unsigned long long* count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    if (is_new[i]) { t.fresh[s] = 0; atomicAdd(count, 1ull); }
    t.owner[s] = 0xffffffffu; // benign same-value races
}

__global__ __launch_bounds__(256) void dedup_probe_kernel(DedupTable t, const uint64_t* digests, uint32_t n,
                                                          uint8_t* exists) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t* d = digests + 4ull * i;
    const uint64_t tag = digest_tag(d);
    const uint32_t mask = t.capacity - 1u;
    uint32_t s = static_cast<uint32_t>((d[1] ^ (d[0] >> 17)) & mask);
    for (;;) {
        const uint64_t cur = t.tags[s];
        if (cur == 0) { exists[i] = 0; return; }
        if (cur == tag) {
            const uint64_t* k = t.keys + 4ull * s;
            if (k[0] == d[0] && k[1] == d[1] && k[2] == d[2] && k[3] == d[3]) { exists[i] = 1; return; }
        }
        s = (s + 1u) & mask;
    }
}

// Growth: every live key of the old table claims an empty slot of the new one (keys are distinct,
// so an occupied slot is simply skipped — no comparison, no race).
__global__ __launch_bounds__(256) void dedup_rehash_kernel(DedupTable old_t, DedupTable new_t) {
    const uint32_t s0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (s0 >= old_t.capacity || old_t.tags[s0] == 0) return;
    const uint64_t* d = old_t.keys + 4ull * s0;
    const uint64_t tag = old_t.tags[s0];
    const uint32_t mask = new_t.capacity - 1u;
    uint32_t s = static_cast<uint32_t>((d[1] ^ (d[0] >> 17)) & mask);
    for (;;) {
        unsigned long long* tp = reinterpret_cast<unsigned long long*>(&new_t.tags[s]);
        if (atomicCAS(tp, 0ull, static_cast<unsigned long long>(tag)) == 0ull) {
            uint64_t* k = new_t.keys + 4ull * s;
            k[0] = d[0]; k[1] = d[1]; k[2] = d[2]; k[3] = d[3];
            return;
        }
        s = (s + 1u) & mask;
    }
}

__global__ void dedup_fill_owner_kernel(uint32_t* owner, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) owner[i] = 0xffffffffu;
}

// Chunk bytes that are new / deduplicated (bytesStored / bytesDeduped, content_store_impl.cpp:255,274).
__global__ __launch_bounds__(256) void dedup_bytes_kernel(const uint8_t* is_new, const uint64_t* sizes, uint32_t n,
                                                          unsigned long long* out2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long a = 0, b = 0;
    if (i < n) { if (is_new[i]) a = sizes[i]; else b = sizes[i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if ((threadIdx.x & 63) == 0) { if (a) atomicAdd(&out2[0], a); if (b) atomicAdd(&out2[1], b); }
}

#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)
static inline dim3 g1(uint32_t n) { return dim3((n + 255) / 256); }

hipError_t launch_dedup_round(hipStream_t st, const DedupTable& t, const uint64_t* digests, uint32_t n,
                              uint8_t* pending, uint32_t* probe_start, uint32_t* slot_of, uint8_t* is_new,
                              unsigned int* n_unresolved, int first_round) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dedup_claim_kernel, g1(n), dim3(256), 0, st, t, digests, n, pending, probe_start, slot_of, first_round);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(dedup_write_kernel, g1(n), dim3(256), 0, st, t, digests, n, pending, slot_of, first_round);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(dedup_verify_kernel, g1(n), dim3(256), 0, st, t, digests, n, pending, probe_start, slot_of,
                       is_new, n_unresolved, first_round);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_dedup_settle(hipStream_t st, const DedupTable& t, uint32_t n, const uint32_t* slot_of,
                               const uint8_t* is_new, unsigned long long* count) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dedup_settle_kernel, g1(n), dim3(256), 0, st, t, n, slot_of, is_new, count);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_dedup_probe(hipStream_t st, const DedupTable& t, const uint64_t* digests, uint32_t n, uint8_t* exists) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dedup_probe_kernel, g1(n), dim3(256), 0, st, t, digests, n, exists);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_dedup_rehash(hipStream_t st, const DedupTable& old_t, const DedupTable& new_t) {
    hipLaunchKernelGGL(dedup_rehash_kernel, g1(old_t.capacity), dim3(256), 0, st, old_t, new_t);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_dedup_fill_owner(hipStream_t st, uint32_t* owner, uint32_t n) {
    hipLaunchKernelGGL(dedup_fill_owner_kernel, g1(n), dim3(256), 0, st, owner, n);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_dedup_bytes(hipStream_t st, const uint8_t* is_new, const uint64_t* sizes, uint32_t n,
                              unsigned long long* out2) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dedup_bytes_kernel, g1(n), dim3(256), 0, st, is_new, sizes, n, out2);
    LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace yams_accel
