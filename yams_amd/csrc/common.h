// common.h — shared declarations for the gfx950 kernels and the C-ABI host layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/yams_mi355x_accel.h"

namespace yams_accel {

// ---- order-preserving float <-> uint32 keys ----------------------------------------------------
// Larger key == better (larger) score.  NaN maps to the top key so that a row whose fp32 filter
// score is not trustworthy is always kept as a candidate (it is then scored exactly in fp64).
// Key 0 is never produced and marks an empty slot.
__host__ __device__ inline uint32_t f2ord(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu; // NaN
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    __builtin_memcpy(&f, &u, 4);
#endif
    return f;
}
__host__ __device__ inline uint64_t pack_key(float score, uint32_t idx) {
    return (static_cast<uint64_t>(f2ord(score)) << 32) | static_cast<uint64_t>(0xffffffffu - idx);
}
__host__ __device__ inline uint32_t key_idx(uint64_t k) {
    return 0xffffffffu - static_cast<uint32_t>(k);
}
__host__ __device__ inline float key_score(uint64_t k) {
    return ord2f(static_cast<uint32_t>(k >> 32));
}

// ---- dense sample scores ---------------------------------------------------------------------------
// Layout [sample_row / 4][query][4]: in the MFMA accumulator layout a lane owns ONE query and four
// consecutive rows per register quad, so the 32 lanes of a half wave store 32 consecutive queries x
// 16 bytes = 512 contiguous bytes (query-major rows would scatter every lane 4 * sample_rows bytes
// apart).  sample_row is a multiple of 4 at every store / load site.
__host__ __device__ inline uint64_t dense_index(uint32_t q, uint64_t sample_row, uint32_t n_queries) {
    return ((sample_row >> 2) * n_queries + q) * 4ull + (sample_row & 3ull);
}

// ---- scan geometry (shared by kernels and host planner) ----------------------------------------
constexpr int kTileRows = 128;    // corpus rows per workgroup tile
constexpr int kTileQueries = 128; // queries per workgroup tile
constexpr int kSlabK = 32;        // k-extent of one LDS stage
constexpr int kGroupRows = 16;    // rows summarised by one "group maximum" in the sample pass
constexpr int kSelectCap = 4096;  // elements one select workgroup sorts
// RescoreLaunch::flags, internal (never a caller's bit): the re-score of the product-quantised engine — the similarity is
// VectorDatabase::computeCosineSimilarity's (0 on a zero norm, no small-norm skip: sqlite_vec_backend.cpp:4023-4034)
constexpr uint32_t kRescoreFlagPqRerank = 1u << 30;
constexpr uint32_t kRescoreFlagNoEarlyClose = 1u << 29;   // (measurement build: the re-score walk never closes early)

struct ScanPlan {
    uint64_t n_rows = 0;
    uint32_t dim = 0;
    uint32_t n_queries = 0;
    uint32_t tile_rows = kTileRows;        // 128 (exact-f32 kernel) or 256 (split-bf16 kernel)
    uint32_t tile_queries = kTileQueries;
    uint32_t n_tiles = 0;        // ceil(n_rows / tile_rows)
    uint32_t sample_stride = 1;  // every sample_stride-th tile is a sample tile
    uint32_t n_sample_tiles = 0;
    uint32_t n_filter_tiles = 0;
    uint32_t n_qtiles = 0;       // ceil(n_queries / tile_queries)
    uint64_t sample_rows = 0;    // n_sample_tiles * tile_rows (padded)
    uint32_t n_groups = 0;       // sample_rows / kGroupRows
    uint32_t list_cap = 0;       // per-query candidate list capacity
    uint32_t kprime = 0;         // candidates re-scored per query in stage 1
    uint32_t tau_rank = 0;       // tau = the tau_rank-th best sample group maximum
};

} // namespace yams_accel
