// scan_args.h — kernel argument block shared by the exact-f32 and the split-bf16 scan kernels.
#pragma once
#include "common.h"

namespace yams_accel {

enum { MODE_SAMPLE = 0, MODE_FILTER = 1 };

struct ScanArgs {
    const float* rows;      // [n_rows][dim]
    const uint16_t* rows_bf16; // nullable shadow: [n_rows][dim] RNE bf16 of rows
    const float* rows_nsq;     // with it: [n_rows] fp32 squared norms
    const int8_t* rows_i8;     // nullable INT8 shadow: round(unit row / s_b), blocked (i8_blocked_offset), padded to 64 rows
    const float* rows_i8_meta; // with it: [ceil(n_rows / 64)][2] = {s_b, e_b} per block of 64 rows
    const int8_t* q_i8;        // [dim/64][q_pad][64] int8 queries of the batch (int8 tier, k-slab-major)
    const float* q_meta;       // [q_pad][4] = {t_q, c_q, f_q, 0}
    const float* q_thr;        // [q_pad][2] = {A_lo, B_hi}: per-query halves of the integer thresholds
    // int8 tier, filter pass: survivors go to a per-(workgroup, wave) log instead of reserving list slots
    // with returning global atomics (whose round trip sat on the critical path of every tile)
    uint64_t* log_key;         // [grid * 8][log_cap] candidate keys
    uint32_t* log_q;           // [grid * 8][log_cap] their queries
    uint32_t* log_cnt;         // [grid * 8] entries written (zeroed before the launch)
    uint32_t log_cap;
    uint32_t* q_over;          // [n_queries] set when a log region was too small for a query's survivors
    uint32_t* i8_sync;         // resident-query form: [n_streams][4 wave pairs][32 query tiles] strips drawn so far (zeroed before the launch)
    // int8 tier, L2 batches (scan_i8_kernel.hip, "L2 on the int8 tier"): the per-row part of the integer threshold
    const uint8_t* i8_row_bias; // [ceil(n_rows / 64)][4 lq][4 rb][4 r] a_r of row 16 rb + 4 lq + r of the block; null for cosine
    const uint32_t* i8_q_bias;  // [q_pad] m_q: a survivor needs I >= T(block, q) + a_r m_q
    float l2_eps;               // relative slack of the L2 score bound (i8_l2_eps)
    const float* qprep;     // [n_queries][dim] prepared queries (unit-norm for cosine, raw for L2)
    const uint16_t* q_hi;   // [dim/16][q_pad][16] bf16 head of qprep (split-bf16 kernel, k-slab-major)
    const uint16_t* q_lo;   // same layout: bf16 of (qprep - head)
    uint32_t q_pad;         // padded query count of the k-slab-major planes (n_qtiles * 256)
    const uint32_t* row_mask; // nullable allow-bitmap over rows
    uint64_t n_rows;
    uint32_t dim;
    uint32_t n_queries;
    uint32_t n_sel_tiles;   // tiles this launch covers
    uint32_t stride;        // sample stride (tile % stride == 0 is a sample tile)
    uint32_t n_qtiles;
    // sample mode
    float* dense;           // [n_queries][sample_rows]
    uint32_t* gmax;         // [n_queries][n_groups] order-preserving keys
    uint64_t sample_rows;
    uint32_t n_groups;
    // filter mode
    const float* tau;       // [n_queries]
    uint32_t* list_count;   // [n_queries]
    uint64_t* list;         // [n_queries][list_cap]
    uint32_t list_cap;
    // L2
    const float* qnorm_up;  // [n_queries] fp32 upper bound of ||q||
    float err_coef;         // (dim + 8) * 2^-24 * 1.01
};

__device__ __forceinline__ bool norm_in_range(float nsq) { return nsq > 1e-30f && nsq < 1e30f; }

// allow-mask word of the 32 rows starting at row32 (a multiple of 32); all ones without a mask
__device__ __forceinline__ uint32_t mask_word(const uint32_t* row_mask, uint64_t row32, uint64_t n_rows) {
    if (!row_mask) return 0xffffffffu;
    return row32 < n_rows ? row_mask[row32 >> 5] : 0u;
}

} // namespace yams_accel
