// scan_i8_kernel.hip — the INT8 tier of the filter pass (cosine).
//
// Same contract as the bf16 tiers (scan_bf16_kernel.hip): score every (row, query) pair with a
// rigorous bound, keep what can reach the top k; the fp64 re-score + proof that follow make the
// result bit-identical to the reference (sqlite_vec_backend.cpp:4204-4331).  What changes is the
// arithmetic of the contraction: the rows and the queries are quantised to int8 and multiplied on
// v_mfma_i32_16x16x64_i8 — exact integer accumulation, half the operand bytes of bf16, and the
// highest sustained matrix rate this part offers (scripts/ubench/mfma_i8_rate.hip, random operands,
// two waves per SIMD: 4.4 POP/s, against 3.6 for v_mfma_i32_32x32x32_i8 and 1.77 PFLOP/s for
// v_mfma_f32_32x32x16_bf16 — the 16 x 16 shape moves a quarter of the accumulator registers per
// multiply-add and sustains a higher clock).
//
// Quantisation (shadow_build_i8_kernel): the unit-normalised row x~ is stored as int8 xi with ONE
// scale per block of 64 rows, s_b = max |x~_i| over the block / 127, together with e_b = the largest
// MEASURED residue |x~ - s_b xi| of the block's rows (not a worst-case figure).  Queries get their own
// scale t_q per batch (prep_i8_kernel), with c_q >= |t_q qi| and f_q >= |q~ - t_q qi| + slop.  Then
//     cos(x, q) = x~ . q~ = s_b t_q (xi . qi) + d . (t_q qi) + x~ . p   <=   s_b t_q I + e_b c_q + f_q =: u
// (Cauchy-Schwarz; I = xi . qi exactly).  The filter score of this tier IS the upper bound u, so the
// completeness proof of the re-score needs no further error term.  A row survives iff u >= tau_q, i.e.
//     I >= (tau_q - f_q) / t_q * (1 / s_b) - (c_q / t_q) * (e_b / s_b) = A_q IS_b - B_q G_b,
// an INTEGER threshold T per (64-row block = the rows of one wave, query): eight per lane.  The
// accumulators START at -T (computed while the
// first k-slabs are still on their way from HBM), so after the k loop "survives" is a sign bit: the
// epilogue is one v_max3 tree per query block, and only lanes that hold a survivor (~1 %) look at
// single elements.
#include <cstdlib>
#include <type_traits>

#include "lds_dma.h"
#include "scan_args.h"

namespace yams_accel {

using i32x4v = __attribute__((ext_vector_type(4))) int;

constexpr int I8_ROWS = 256, I8_QUERIES = 256, I8_THREADS = 512;
constexpr int I8_SLAB = 64;                        // bytes (= int8 k-values) per row per ring stage
constexpr int I8_NST = 4;
constexpr int I8_A_BYTES = I8_ROWS * I8_SLAB;      // 16 KiB
constexpr int I8_B_BYTES = I8_QUERIES * I8_SLAB;   // 16 KiB
constexpr int I8_STAGE = I8_A_BYTES + I8_B_BYTES;
constexpr int I8_BLOCK_ROWS = 64;                  // rows that share one quantisation scale (= a wave's rows of a tile)

// LDS image of a 64-byte row slab: four 16-byte chunks; logical chunk c of row R sits at position
// c ^ g((R >> 2) & 3), g = {0, 2, 3, 1}.  A v_mfma_i32_16x16x64_i8 operand is "row (lane & 15),
// chunk (lane >> 4)"; ds_read_b128 serves a wave in four fixed 16-lane groups ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, ...): with this g every group touches each of the 16 bank quads exactly once.
__device__ __forceinline__ int i8_swz(int row) {
    const int j = (row >> 2) & 3;
    return (((j ^ (j >> 1)) & 1) << 1) | (j >> 1);
}

// The int8 shadow in memory = the LDS image of its DMA pieces: [row / 16][slab][16 rows][4 positions][16 B],
// position p of row r holding the row's logical chunk p ^ i8_swz(r).  A piece (16 rows x one 64-byte slab) is
// one contiguous KiB — eight full 128-byte lines — and lane i of the wave that stages it fetches bytes
// [16 i, +16): with the rows stored row-major a piece was sixteen 64-byte segments, i.e. half of every line it
// touched belonged to the neighbouring slab, and the row stream cost twice the L2 -> L1 line traffic (measured
// on the bench shard, same kernel: 8.4-8.6 ms row-major, 7.7 ms blocked, identical results).  The shadow is
// padded to whole blocks of 64 rows (one quantisation scale = four pieces).
__host__ __device__ __forceinline__ uint64_t i8_blocked_offset(uint64_t row, uint32_t col, uint32_t dim) {
    const uint32_t j = static_cast<uint32_t>(row & 15u), slab = col >> 6, chunk = (col >> 4) & 3u;
    const uint32_t g = (j >> 2) & 3u, swz = (((g ^ (g >> 1)) & 1u) << 1) | (g >> 1);
    return ((row >> 4) * (dim >> 6) + slab) * 1024u + j * 64u + ((chunk ^ swz) << 4) + (col & 15u);
}

// -T(row block, query) for T = A_lo IS_b - B_hi G_b - 2: what an accumulator starts at.  A_lo / B_hi
// carry the relative slack for this fp32 evaluation, the 2 covers the truncation towards zero of the
// conversion; clamped to +-2^30 (-2^30 = nothing survives, +2^30 = everything does — |xi . qi| <=
// 127^2 dim stays far below either).  A_lo is never NaN (i8_query_thresholds_kernel), G_b is finite.
__device__ __forceinline__ int i8_neg_threshold(float A, float is, float B, float g) {
    const float t = fmaf(-A, is, fmaf(B, g, 2.0f));
    return static_cast<int>(__builtin_amdgcn_fmed3f(t, -1.0737418e9f, 1.0737418e9f));
}

// -------------------------------------------------------------------------------------------------
// L2 on the int8 tier (vec0 order: ascending |x - q|, i.e. descending g = q.x - |x|^2 / 2).
// The shadow is the cosine tier's (unit rows), the queries stay raw, so u = s_b t_q I + e_b c_q + f_q
// bounds x~ . q from above and, with n = |x| (sqrt of the fp32 squared norm of the bf16 shadow's
// rows_nsq, relative error far below eps),
//     g = n (x~ . q) - n^2 / 2  <=  n u - n^2 / 2 + eps (|n u| + n^2 / 2) =: G(u, n^2)      (i8_l2_bound)
// is the filter score: sample maxima, tau, the candidate keys and the re-score's proof all live in
// units of g, exactly as on the bf16 tier.  What is new is the integer threshold.  G >= tau needs
//     u >= h(n) = tau' / n + n / 2,      tau' = tau - (the most the eps term can add for this query),
// and h is concave (tau' < 0) or convex in n, never linear: a block threshold from the smallest norm
// of the block alone would let through everything within beta (n_r - nmin_b) of the bound — for rows
// whose norms spread by 1.6 % (uniform components, dim 768) that is a thousand times the survivors.
// So per query a LINE below h over the shard's norm range [Nmin, Nmax],
//     h(n) >= alpha_q + beta_q n,   beta_q = max(1/2 - tau' / (Nmin Nmax), 0),
//     alpha_q = min over [Nmin, Nmax] of h(n) - beta_q n   (the end points, or the stationary point of
//     a convex h) — the chord of a concave h, the tangent at the geometric mean of a convex one,
// which splits the condition into a block part and a row part, n_r = nmin_b + d_r:
//     I >= [(alpha_q - f_q - c_q e_max) / t_q] / s_b + (beta_q / t_q) nmin_b / s_b + (beta_q / t_q) d_r / s_b
//          `----------------- T(block, query): A_q IS_b - B_q G_b -----------------'   `--- >= a_r m_q ---'
// with e_max = the largest e_b of the shard (e_b is the largest of 64 measured residues: it hardly varies
// from block to block, and c_q (e_max - e_b) is all this costs).  T has the cosine tier's form with the
// block's second meta word replaced by -nmin_b: the filter kernel gets a second meta array and is otherwise unchanged.  The
// row part is an 8-bit a_r = floor(d_r / (s_b W)) (W = the largest d_r / s_b of the shard / 255) times
// a per-query integer m_q = floor(beta_q W / t_q): accumulators start at -T - a_r m_q, one v_mad_i24
// per element where the cosine kernel has a v_mov.  The gather kernel undoes both, forms G and keeps
// the rows with G >= tau.  Rows whose squared norm lies outside norm_in_range() (zero rows — valid under L2 —,
// overflowing or non-finite ones) have no usable bound: they are left out of every table and statistic, score
// -inf in the sample pass, and join EVERY query's candidate list unconditionally (i8_l2_add_special_kernel; at
// most I8_L2_MAX_SPECIAL of them, else the batch stays on the bf16 tier, as it does when Nmax / Nmin > 2).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float i8_l2_bound(float u, float nsq, float eps) {
    const float t1 = sqrtf(nsq) * u;
    return fmaf(eps, fmaf(0.5f, nsq, fabsf(t1)), fmaf(-0.5f, nsq, t1));
}
// d_r: how far row norm n lies above its block's smallest, made safe against the rounding of both square roots
__device__ __forceinline__ float i8_l2_spread(float n, float nmin) {
    return fmaxf(0.f, (n - nmin) - n * 2.3841858e-7f);
}
// W from the largest d_r / s_b of the shard (the same expression wherever a_r or m_q is formed)
__device__ __forceinline__ float i8_l2_unit(uint32_t spread_max_bits) {
    return __uint_as_float(spread_max_bits) * (1.0f / 255.0f) * (1.0f + 9.5367432e-7f);
}


// -------------------------------------------------------------------------------------------------
// The filter on HALF tiles: 128 rows x 256 queries per workgroup of FOUR waves (wave tile still
// 64 x 128, 3-stage ring of 24 KiB), two workgroups per CU.  r02 counters of the 8-wave kernel: MFMA
// pipe 43 % busy, wave slots 17 % empty (one workgroup per CU: nothing runs while it turns over) and
// both waves of a SIMD meet at the same barrier.  Two independent workgroups per CU overlap one's
// barrier / epilogue / prologue with the other's k loop.
// -------------------------------------------------------------------------------------------------
constexpr int H_ROWS = 128, H_THREADS = 256, H_NST = 3;
constexpr int H_A_BYTES = H_ROWS * I8_SLAB;        // 8 KiB
constexpr int H_STAGE = H_A_BYTES + I8_B_BYTES;    // 24 KiB

// per-query thresholds of the filter pass: qthr[q] = {A_lo, B_hi} (see i8_query_thresholds_kernel)
// MODE_SAMPLE writes group maxima (groups of 16 rows: the rows one lane holds for a query block — 4 row
// blocks x 4 consecutive rows) and, only when a dense buffer is given, all upper bounds (the int8 tier runs
// without one: i8_collect_sample_kernel re-derives the scores of the groups that reach tau).
// METRIC = L2 ("L2 on the int8 tier" above): the sample pass scores a row with G(u, |x|^2) (i8_l2_bound); the
// filter pass starts its accumulators at -T - a_r m_q (a.rows_i8_meta is then the thresholds meta of the shard).
template <int MODE, int ABL = 0, int METRIC = YAMS_SCAN_COSINE>
__global__ __launch_bounds__(H_THREADS, 2) void scan_tiles_i8h_kernel(ScanArgs a) {
    static_assert(METRIC == YAMS_SCAN_COSINE || ABL == 0, "the measurement forms exist for the cosine kernels only");
    __shared__ __attribute__((aligned(16))) unsigned char lds[H_NST * H_STAGE];
    __shared__ uint32_t wave_log[4]; // survivors each wave has logged (wave-private slots)

    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u;
    const uint32_t w = bid >> 3;
    const uint32_t qt = w % a.n_qtiles;
    const uint32_t sel2 = (w / a.n_qtiles) * 8u + xcd; // (selected 256-row tile, which half of it)
    const uint32_t sel = sel2 >> 1, hf = sel2 & 1u;
    if (sel >= a.n_sel_tiles) return;
    uint32_t tile;
    if (MODE == MODE_SAMPLE) tile = sel * a.stride;
    else tile = sel + sel / (a.stride - 1u) + 1u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1; // wave tile: rows [64 wr, +64) of this half x queries [128 wc, +128)
    const int l15 = lane & 15, lq = lane >> 4;
    const uint64_t row0 = static_cast<uint64_t>(tile) * I8_ROWS + hf * H_ROWS; // may lie past the end (ragged last tile)
    const uint32_t q0 = qt * I8_QUERIES;
    const uint32_t dim = a.dim;
    const int nslab = dim / I8_SLAB; // dim % 64 == 0 and dim >= 256 are preconditions of this tier
    if (MODE == MODE_FILTER && lane == 0) wave_log[wid] = 0u;

    // ---- DMA sources: every wave stages 32 rows (2 pieces of 1 KiB) and 64 queries (4 pieces) per slab ----
    // Row pieces are contiguous KiBs of the blocked shadow (i8_blocked_offset); a 64-row strip of a ragged
    // last tile that lies past the (64-row padded) end re-reads the shadow's last strip.
    const uint64_t padded_rows = (a.n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS * I8_BLOCK_ROWS;
    const uint32_t piece_row_stride = static_cast<uint32_t>(nslab) * 1024u;
    uint64_t stripA = row0 + static_cast<uint32_t>(wid >> 1) * 64u;   // the 64-row strip this wave's two pieces belong to
    if (stripA >= padded_rows) stripA = padded_rows - 64;
    uint32_t voffA[2], voffB[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        voffA[i] = static_cast<uint32_t>(((wid & 1) * 2 + i)) * piece_row_stride + static_cast<uint32_t>(lane) * 16u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rowB = (wid * 4 + i) * 16 + (lane >> 2);
        voffB[i] = static_cast<uint32_t>(rowB) * 64u + ((lane & 3) ^ i8_swz(rowB)) * 16u;
    }
    const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.rows_i8) + (stripA / 16) * piece_row_stride; // (wave-uniform)
    const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.q_i8) + static_cast<uint64_t>(q0) * 64;
    const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    const uint32_t ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * 2048);
    const uint32_t ldsB = __builtin_amdgcn_readfirstlane(lds0 + H_A_BYTES + wid * 4096);
    // piece p of slab s into ring stage `st` (0..2): p = 0, 1 rows, p = 2..5 queries
    auto piece = [&](int s, int st, int p) __attribute__((always_inline)) {
        if ((ABL == 1 || ABL == 3) && s >= H_NST) return; // measurement build: no refills after the prologue
        if (ABL == 5 && s >= H_NST && p >= 2) return;     // measurement build: row pieces only (queries "resident")
        if (ABL == 6 && s >= H_NST && p < 2) return;      // measurement build: query pieces only
        const uint32_t so = static_cast<uint32_t>(st) * H_STAGE;
        if (p < 2) lds_dma16_s(baseA + s * 1024, voffA[p < 2 ? p : 0], ldsA + so + p * 1024);
        else lds_dma16_s(baseB + s * qslab_bytes, voffB[p >= 2 ? p - 2 : 0], ldsB + so + (p - 2) * 1024);
    };

    i32x4v acc[4][8];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][cb][r] = 0;

    // fragment offsets inside a stage: row block rb adds rb * 1 KiB, query block cb adds cb * 1 KiB
    // (block starts are multiples of 16 rows, so the swizzle term depends on the lane only)
    const int offA = (wr * 64 + l15) * 64 + ((lq ^ i8_swz(l15)) << 4);
    const int offB = H_A_BYTES + (wc * 128 + l15) * 64 + ((lq ^ i8_swz(l15)) << 4);

    // epilogue inputs, requested now (older than every DMA piece, so the counted waits stay valid;
    // the compiler waits for them at their first use, after the loop): a load issued at the end would
    // sit on the critical path of every tile — the block scales stream from HBM
    const uint64_t strip = row0 + static_cast<uint32_t>(wr * 64);
    float sb, eb;                       // wave-uniform: scale and residue bound of this wave's 64 rows
    {
        const uint64_t n_blocks = (a.n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
        const uint64_t blk = strip / I8_BLOCK_ROWS;
        const float2 m = blk < n_blocks ? reinterpret_cast<const float2*>(a.rows_i8_meta)[blk] : make_float2(1.f, 0.f);
        sb = m.x; eb = m.y;
    }
    // The per-query threshold halves {A_lo, B_hi}: loaded from inline asm so that the compiler does
    // not wait for them with a conservative vmcnt(0) (it cannot see the DMA pieces that follow); they
    // are older than every piece, so "at most 14 younger operations outstanding" means they landed.
    typedef float qthr_t __attribute__((ext_vector_type(2)));
    qthr_t qthr[8];
    constexpr bool THR = MODE == MODE_FILTER && (ABL == 0 || ABL == 8);
    if (THR) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const float* p = a.q_thr + 2ull * (q0 + wc * 128 + cb * 16 + l15); // < q_pad: the table is padded
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(qthr[cb]) : "v"(p) : "memory");
        }
    }
    // L2: m_q of this lane's eight queries, a_r of its sixteen rows (byte r of word rb = row 16 rb + 4 lq + r) — like the
    // threshold halves requested before the first DMA piece, so the same counted wait covers them
    constexpr bool L2F = THR && METRIC == YAMS_SCAN_L2;
    int qbias[8];
    i32x4v rbias;
    if (L2F) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const uint32_t* p = a.i8_q_bias + (q0 + wc * 128 + cb * 16 + l15);
            asm volatile("global_load_dword %0, %1, off" : "=v"(qbias[cb]) : "v"(p) : "memory");
        }
        const uint64_t n_blocks = (a.n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
        uint64_t blk = strip / I8_BLOCK_ROWS;
        if (blk >= n_blocks) blk = n_blocks - 1; // a strip past the end: nothing of it is ever emitted
        const uint8_t* p = a.i8_row_bias + blk * 64u + static_cast<uint32_t>(lq) * 16u;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rbias) : "v"(p) : "memory");
    }

    {   // prologue: slabs 0 and 1 and the first half of slab 2 in flight (nslab >= 4), slab 0 landed
        for (int s = 0; s < 2; ++s) for (int p = 0; p < 6; ++p) piece(s, s, p);
        piece(2, 2, 0); piece(2, 2, 1); piece(2, 2, 2);
        if (THR) {
            // accumulators start at -T(row block, query block) while the slabs are in flight
            if (L2F)
                asm volatile("s_waitcnt vmcnt(15)" : "+v"(qthr[0]), "+v"(qthr[1]), "+v"(qthr[2]), "+v"(qthr[3]),
                                                     "+v"(qthr[4]), "+v"(qthr[5]), "+v"(qthr[6]), "+v"(qthr[7]), "+v"(rbias),
                                                     "+v"(qbias[0]), "+v"(qbias[1]), "+v"(qbias[2]), "+v"(qbias[3]),
                                                     "+v"(qbias[4]), "+v"(qbias[5]), "+v"(qbias[6]), "+v"(qbias[7]) :: "memory");
            else
                asm volatile("s_waitcnt vmcnt(15)" : "+v"(qthr[0]), "+v"(qthr[1]), "+v"(qthr[2]), "+v"(qthr[3]),
                                                     "+v"(qthr[4]), "+v"(qthr[5]), "+v"(qthr[6]), "+v"(qthr[7]) :: "memory");
            const float is = 1.0f / sb, g = eb * is;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int nt = i8_neg_threshold(qthr[cb][0], is, qthr[cb][1], g);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[rb][cb][r] = L2F ? nt - __mul24(static_cast<int>((static_cast<uint32_t>(rbias[rb]) >> (8 * r)) & 255u), qbias[cb])
                                             : nt;
            }
        }
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); // 15 pieces issued, slab 0's six have landed
        __builtin_amdgcn_s_barrier();
    }
    // Fragments: A (4 row blocks) double-buffered across slabs, B in two halves of 4 query blocks.
    i32x4v fa[2][4], fb[2][4];
    auto ld = [&](const unsigned char* base, int off) __attribute__((always_inline)) -> i32x4v {
        if (ABL == 4) { i32x4v z = {off, 0, 0, 0}; return z; } // measurement build: no fragment reads
        return *reinterpret_cast<const i32x4v*>(base + off);
    };
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) fa[0][rb] = ld(lds, offA + rb * 1024);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) fb[0][cb] = ld(lds, offB + cb * 1024);
    int stage = 0;

    // One slab = two halves of 16 MFMAs (all four row blocks x four query blocks each).
    //   half 1: multiplies fa[cur] x fb[0]; requests the other four query blocks of THIS slab (fb[1])
    //           and issues the second half of slab s+2's DMA pieces (its stage was released by the
    //           barrier of the previous iteration);
    //   barrier: slab s+1 has landed, every wave is done reading slab s;
    //   half 2: multiplies fa[cur] x fb[1]; requests slab s+1's row blocks (fa[nxt]) and first four
    //           query blocks (fb[0]) and issues the first half of slab s+3's pieces into the stage
    //           slab s just left.
    // Two workgroups of four waves share a CU (72 KiB of LDS and 256 VGPRs each): while one sits at
    // its barrier, in its epilogue or in the prologue of its next tile, the other keeps the MFMA pipe fed.
    auto half = [&](const i32x4v (&A)[4], const i32x4v (&B)[4], int cb0, auto&& filler) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rb = i >> 2, c = i & 3;
            if (ABL != 2 && ABL != 3 && ABL != 4)
                acc[rb][cb0 + c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rb], B[c], acc[rb][cb0 + c], 0, 0, 0);
            else if (i == 0)
                asm volatile("" :: "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]));
            __builtin_amdgcn_sched_barrier(0);
            filler(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto pin4 = [&](i32x4v (&F)[4]) __attribute__((always_inline)) { asm volatile("" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3])); };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // REM = slabs left including this one (4 = four or more): decides which DMA halves are still to be
    // issued, how many newer pieces may be in flight at the barrier, and whether a next slab exists.
    // CUR = which of the two row-fragment buffers this slab uses (compile time: a runtime index would
    // push the fragment arrays into scratch memory).
    auto body = [&](int s, auto cur_tag, auto rem_tag) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr int REM = decltype(rem_tag)::value;
        constexpr bool H1 = REM >= 3;                  // slab s+2 exists: issue its second half
        constexpr bool H2 = REM >= 4;                  // slab s+3 exists: issue its first half
        constexpr bool MORE = REM >= 2;
        const int st0 = stage;                                   // slab s (and, after the barrier, slab s+3)
        const int st1 = stage == 2 ? 0 : stage + 1;              // slab s+1
        const int st2 = st1 == 2 ? 0 : st1 + 1;                  // slab s+2
        const unsigned char* base = lds + st0 * H_STAGE;
        const unsigned char* nbase = lds + st1 * H_STAGE;
        stage = st1;
        pin4(fa[CUR]); pin4(fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        half(fa[CUR], fb[0], 0, [&](int i) __attribute__((always_inline)) {
            if (i < 4) fb[1][i < 4 ? i : 0] = ld(base, offB + (4 + (i < 4 ? i : 0)) * 1024);
            if (H1 && (i == 5 || i == 9 || i == 13)) piece(s + 2, st2, i == 5 ? 3 : (i == 9 ? 4 : 5));
        });
        // slab s+1 is older than slab s+2's six pieces
        if (H1 && ABL == 5 && s >= 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if (H1 && ABL == 6 && s >= 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (H1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        pin4(fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        half(fa[CUR], fb[1], 4, [&](int i) __attribute__((always_inline)) {
            if (MORE && i < 4) fa[CUR ^ 1][i < 4 ? i : 0] = ld(nbase, offA + (i < 4 ? i : 0) * 1024);
            if (MORE && i >= 4 && i < 8) fb[0][(i - 4) & 3] = ld(nbase, offB + ((i - 4) & 3) * 1024);
            if (H2 && (i == 9 || i == 11 || i == 13)) piece(s + 3, st0, i == 9 ? 0 : (i == 11 ? 1 : 2));
        });
    };
    using R4 = std::integral_constant<int, 4>;
    using R3 = std::integral_constant<int, 3>;
    using R2 = std::integral_constant<int, 2>;
    using R1 = std::integral_constant<int, 1>;
    // nslab >= 4 (dim >= 256, checked by the host).  The last three slabs have their own bodies; the
    // nslab - 3 steady-state slabs run two per trip with the buffer parity fixed at compile time.  An
    // odd count runs one steady body first and then renames the prefetched row fragments into buffer
    // 0, so that a single code path leads into the pair loop and the tail.
    const int n_steady = nslab - 3;
    int s = 0;
    if (n_steady & 1) {
        body(0, C0{}, R4{});
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) fa[0][rb] = fa[1][rb];
        s = 1;
    }
    for (; s < n_steady; s += 2) { body(s, C0{}, R4{}); body(s + 1, C1{}, R4{}); }
    body(s, C0{}, R3{}); body(s + 1, C1{}, R2{}); body(s + 2, C0{}, R1{});

    if (ABL != 0 && ABL != 8) { // measurement builds: keep the accumulators alive, emit nothing
        int t = 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) t += acc[rb][cb][r];
        if (t == 123456789 && a.list_cap == 0xffffffffu) a.list_count[0] = 1;
        return;
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    // accumulator element acc[rb][cb][r]: row = strip + 16 rb + 4 lq + r, query = qb + 16 cb + l15
    const uint32_t qb = q0 + wc * 128;
    if (MODE == MODE_SAMPLE) {
        const float ninf = -__builtin_inff();
        float nsqv[4][4]; // L2: the squared norms of this lane's sixteen rows
        if (METRIC == YAMS_SCAN_L2) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t row = strip + 16 * rb + 4 * lq + r;
                    nsqv[rb][r] = row < a.n_rows ? a.rows_nsq[row] : 1.f;
                }
        }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const uint32_t qi = qb + cb * 16 + l15;
            const bool qok = qi < a.n_queries;
            const float4 qm = reinterpret_cast<const float4*>(a.q_meta)[qi]; // {t_q, c_q, f_q, 0}
            float m = ninf;
            const float S = sb * qm.x, K = fmaf(eb, qm.y, qm.z);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const uint64_t rbase = strip + 16 * rb + 4 * lq;
                const uint32_t mw = a.row_mask ? mask_word(a.row_mask, rbase & ~31ull, a.n_rows) >> (rbase & 31u) : 0xfu;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float u = fmaf(static_cast<float>(acc[rb][cb][r]), S, K);
                    if (METRIC == YAMS_SCAN_L2) u = norm_in_range(nsqv[rb][r]) ? i8_l2_bound(u, nsqv[rb][r], a.l2_eps) : ninf; // (unbounded rows are candidates of every query anyway)
                    v[r] = (rbase + r < a.n_rows && ((mw >> r) & 1u)) ? u : ninf;
                    m = fmaxf(m, v[r]);
                }
                if (qok && a.dense) { // (no dense buffer: i8_collect_sample_kernel re-derives the scores of the few groups that matter)
                    const uint64_t srow = static_cast<uint64_t>(sel) * I8_ROWS + hf * H_ROWS + static_cast<uint32_t>(wr * 64 + 16 * rb + 4 * lq);
                    *reinterpret_cast<float4*>(a.dense + dense_index(qi, srow, a.n_queries)) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            if (qok) {
                const uint32_t gid = (sel * I8_ROWS + hf * H_ROWS + static_cast<uint32_t>(wr * 64)) / 16u + lq;
                a.gmax[static_cast<uint64_t>(qi) * a.n_groups + gid] = (m != m) ? 0xffffffffu : f2ord(m);
            }
        }
        return;
    }
    // FILTER: the accumulators hold I - T, a survivor is a non-negative one
    uint32_t hot = 0;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        int m = acc[0][cb][0];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = acc[rb][cb][r] > m ? acc[rb][cb][r] : m;
        if (m >= 0 && qb + cb * 16 + l15 < a.n_queries) hot |= 1u << cb;
    }
    if (hot == 0) return; // ~99 % of the lanes
    // the lane holds survivors: which elements (row bound and allow-mask checked here), one reservation
    // per query block (all of them issued before the first store), then the stores
    uint32_t pass[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        pass[cb] = 0u;
        if ((hot >> cb) & 1u) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const uint64_t rbase = strip + 16 * rb + 4 * lq;
                const uint32_t mw = a.row_mask ? mask_word(a.row_mask, rbase & ~31ull, a.n_rows) >> (rbase & 31u) : 0xfu;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (acc[rb][cb][r] >= 0 && rbase + r < a.n_rows && ((mw >> r) & 1u)) pass[cb] |= 1u << (4 * rb + r);
            }
        }
    }
    if (ABL == 8) { // measurement build: the whole epilogue up to here, but nothing is emitted
        uint32_t t = 0;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) t |= pass[cb];
        if (t == 0x1234u && a.list_cap == 0xffffffffu) a.list_count[0] = 1;
        return;
    }
    // Survivors go to this wave's region of the log: slots come from an LDS counter (a ~100-cycle round
    // trip; a returning GLOBAL atomic per query block took microseconds at the end of every tile), the
    // stores are fire-and-forget.  i8_log_gather_kernel moves the log into the per-query lists.
    uint32_t mine = 0;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) mine += static_cast<uint32_t>(__builtin_popcount(pass[cb]));
    uint32_t pos = mine ? atomicAdd(&wave_log[wid], mine) : 0u;
    const uint64_t region = (static_cast<uint64_t>(bid) * 4u + static_cast<uint32_t>(wid)) * a.log_cap;
    // An entry is (accumulator, row) + the query: no global load sits between the k loop and the end of
    // the tile; the gather kernel turns the accumulator back into the score bound u.
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        if (!pass[cb]) continue;
        const uint32_t qi = qb + cb * 16 + l15;
        bool lost = false;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!((pass[cb] >> (4 * rb + r)) & 1u)) continue;
                const uint64_t row = strip + 16 * rb + 4 * lq + r;
                if (pos < a.log_cap) {
                    a.log_key[region + pos] = (static_cast<uint64_t>(static_cast<uint32_t>(acc[rb][cb][r])) << 32) | static_cast<uint32_t>(row);
                    a.log_q[region + pos] = qi;
                } else {
                    lost = true;
                }
                ++pos;
            }
        }
        if (lost) atomicOr(&a.q_over[qi], 1u); // no room: this query's list is incomplete -> exhaustive path (no return value used)
    }
    // the wave's total (all hot lanes ran the LDS add in the same instruction): one lane publishes it
    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
    if (lane == static_cast<int>(__builtin_ctzll(act))) {
        const uint32_t total = wave_log[wid];
        a.log_cnt[static_cast<uint64_t>(bid) * 4u + static_cast<uint32_t>(wid)] = total < a.log_cap ? total : a.log_cap;
    }
}

// -------------------------------------------------------------------------------------------------
// The filter with the QUERY TILE RESIDENT in LDS (dim <= 768, dim % 128 == 0): one workgroup of eight
// waves per CU stays for the whole launch, holds 128 queries x dim int8 (<= 96 KiB) and streams row
// strips past them.  Ablations of the half-tile kernel on the bench shard (12.5M x 768, 1024 queries):
// 8.9 ms as shipped, 7.1 ms with the query pieces of the ring left out, 6.1 ms without any refill —
// two thirds of its LDS-DMA pieces (and of its L2 -> LDS bytes) re-stage queries that never change.
// Here
//   * a query slab is staged ONCE per workgroup; the only steady-state DMA traffic is the rows: 1/128
//     byte per multiply-add instead of 1/128 + 1/256, four pieces per wave and slab instead of six;
//   * every wave stages and multiplies its OWN 64 rows (a ring of two 4 KiB slabs in LDS plus a
//     register double buffer), so there is no workgroup barrier after the prologue, no shared ring to
//     drain at a tile boundary and no tile prologue: the row stream runs straight through the units;
//   * the eight workgroups that hold the eight query tiles of a batch sit on the same XCD and walk the
//     same row units in the same order, so a row comes from HBM once and seven times from that L2.
// A unit = 512 rows (two consecutive filter tiles = eight 64-row strips; the waves w and w + 4 of a SIMD
// draw the strips w & 3 of both tiles from one counter, see "work sharing"), the accumulator layout and the
// thresholds are those of the half-tile kernel; survivors go to one log region per wave.
// -------------------------------------------------------------------------------------------------
constexpr int R_QUERIES = 128, R_THREADS = 512, R_MAX_SLABS = 12;
constexpr int R_B_SLAB = R_QUERIES * I8_SLAB;      // 8 KiB of queries per k-slab
constexpr int R_RING = 2 * 64 * I8_SLAB;           // per wave: two slabs of its 64 rows
constexpr int R_LDS = R_MAX_SLABS * R_B_SLAB + 8 * R_RING; // 160 KiB: everything a CU has

// L2: the accumulators start at -T(block, query) - a_r m_q ("L2 on the int8 tier" above); a.rows_i8_meta is the
// batch's thresholds meta, a.i8_row_bias / a.i8_q_bias hold a_r and m_q.  Nothing else differs.
// DIRECT (round 3): the row fragments are loaded straight from global memory into the register double buffer — the
// blocked shadow already is the fragment image, and a strip's rows are read by one wave only, so the LDS ring bought
// nothing but latency cover (1.6 slabs against the one slab a register pair gives) at the price of a DMA write and a
// fragment read per row byte: half of the kernel's LDS traffic, which ran at the LDS's 128 B/clk.
// ZS (round 4; cosine, DIRECT only): what a wave does BETWEEN two strips is not hidden by its partner on the SIMD — a
// wave runs one slab ahead of its loads, so it cannot take over the matrix pipe while the other one is busy elsewhere
// (measured: the same launch without epilogue 6.8 ms against 7.4, at dim 384 3.5 against 4.8).  So the strip boundary
// is made short: (1) the accumulators are BORN in the strip's first slab (C operand = the constant 0) instead of being
// set to -T one register at a time (128 moves), the thresholds enter the sign test; (2) the per-query threshold halves
// live in the LDS the direct form's ring no longer needs (a 100-clock read instead of eight global loads and a drain
// of the vector-memory counter); (3) survivors go to a per-wave buffer in that same LDS and reach the log in one
// flush at the end (or when it fills): no global store at a strip boundary, so every wait of the k loop counts loads only.
constexpr int R_ZS_THR = R_MAX_SLABS * R_B_SLAB;        // {A_lo, B_hi} of the tile's 128 queries: 1 KiB
constexpr int R_ZS_LOG = R_ZS_THR + 1024;               // per wave: keys u64[R_ZS_ENTRIES] then queries u32[R_ZS_ENTRIES]
constexpr int R_ZS_ENTRIES = 640;                       // 7.5 KiB per wave, 60 KiB of the ring's 64
static_assert(R_ZS_LOG + 8 * R_ZS_ENTRIES * 12 <= R_LDS, "the survivor buffers must fit the LDS the ring has left");
// ZSM: which parts of the short strip boundary a launch uses — 2 the tile's threshold halves resident in LDS, 4 survivors to
// the LDS buffer, 64 the boundary's memory-independent work in front of the drain.  (Two more forms were built and measured
// slower in round 4 — accumulators born from the strip's first multiply-add with the thresholds applied at the end, and a
// strip end without a drain: profiles/r04_filter_forms.json, docs/LAB_NOTES.md; their code left the kernel in round 6.)
// SAMPLE (round 4; cosine, DIRECT, plain strip boundary): the SAMPLE pass in this form — the strips are those of the sample
// tiles (every stride-th tile), the accumulators start at zero and the epilogue writes the group maxima the half-tile
// kernel's MODE_SAMPLE writes (the same two fmaf per element, the same group numbering: gmax feeds tau_select and
// i8_collect_sample_kernel unchanged).  The half-tile form re-stages the query tile for every 128 rows — 1.2 GB of
// queries for 0.15 GB of sample rows at the bench shape.
template <int ABL = 0, bool L2 = false, bool DIRECT = false, int ZSM = 0, bool SAMPLE = false>
__global__ __launch_bounds__(R_THREADS, 1) void scan_tiles_i8r_kernel(ScanArgs a, uint32_t n_units, uint32_t n_qt, uint32_t n_streams, uint32_t window) {
    static_assert(ZSM == 0 || (DIRECT && !L2 && ABL == 0), "the short strip boundary exists for the direct cosine kernel");
    static_assert(!SAMPLE || (DIRECT && !L2 && ABL == 0 && ZSM == 0), "the sample pass exists in the plain direct cosine form");
    static_assert((ZSM & ~(2 | 4 | 64)) == 0, "ZSM bits: 2 thresholds in LDS, 4 survivors in LDS, 64 early boundary");
    constexpr bool ZT = (ZSM & 2) != 0, ZL = (ZSM & 4) != 0;
    // Round 6 — the strip boundary again (it costs 8 % of the launch at dim 768 and 24 % at dim 384, and the partner wave of
    // the SIMD does not hide it).  64 (EARLY): what the boundary does NOT need the memory for — the survivor emission, the next
    // strip's thresholds and the accumulator set-up — runs BEFORE the drain of the vector-memory counter instead of behind it,
    // i.e. under the latency of the next strip's first fragments (requested in the last slab) instead of after it: 7.25 ->
    // 7.14 ms at dim 768, 4.05 -> 3.92 at dim 384 (12.5M rows, 1024 queries), 146 -> 142 us on BASELINE config 2, same
    // candidate sets (profiles/r06_filter_forms.json).  Tried with it and dropped: setting only row block 0's accumulators
    // to -T and letting the first slab's multiply-adds of the other row blocks take them as their C operand (32 moves for
    // 128) — the second code path of the first slab cost four spills and 2 % of the launch.
    constexpr bool EARLY = (ZSM & 64) != 0;
    static_assert(!EARLY || (ZT && ZL), "the early boundary builds on the plain form with its threshold halves and its survivors in LDS");
    // `window`: bits 0-15 the strips a pair may run ahead of its slowest sibling, bits 16-23 log2 of the pacing interval —
    // the siblings' counters are looked at when (strip number & mask) == 0 only: each look is a system-scope load whose
    // latency the strip boundary pays (every strip: 7.43 / 4.61 ms at dim 768 / 384; every eighth: 7.35 / 4.18)
    const uint32_t pace_mask = (1u << ((window >> 16) & 255u)) - 1u;
    __shared__ __attribute__((aligned(16))) unsigned char lds[R_LDS];

    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, slot = bid >> 3;       // workgroup b runs on XCD b % 8
    const uint32_t qt = slot % n_qt, st = slot / n_qt;    // the query tile it holds, its row stream on this XCD
    if (st >= (n_streams >> 3)) return;
    const uint32_t stream = st * 8u + xcd;
    if (stream >= n_units) return;

#ifdef YAMS_ACCEL_MEASURE
    const uint64_t t_begin = wall_clock64();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const uint32_t dim = a.dim;
    const int nslab = dim / I8_SLAB; // even, 4..12 (checked by the host)
    const uint32_t q0 = qt * R_QUERIES;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    const uint32_t ringW = __builtin_amdgcn_readfirstlane(lds0 + R_MAX_SLABS * R_B_SLAB + wid * R_RING);
    const unsigned char* ring = lds + R_MAX_SLABS * R_B_SLAB + wid * R_RING;
    const uint64_t n_blocks = (a.n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
    const uint64_t past_end = n_blocks * I8_BLOCK_ROWS;

    // ---- where this wave's strip of a unit lives ----------------------------------------------------
    // The shadow is stored as the LDS image of its DMA pieces (i8_blocked_offset): a piece — 16 rows x one
    // 64-byte slab — is ONE contiguous KiB, lane i fetches bytes [16 i, +16).  A strip is 64 rows = four
    // consecutive 16-row blocks; strips never straddle the end of the (64-row padded) shadow.
    struct Geo {
        uint64_t row0;               // first row of the strip; >= n_rows when the strip does not exist
        const unsigned char* base;   // first piece of the strip (uniform); an absent strip reads the last one
    };
    const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
    const uint32_t piece_row_stride = static_cast<uint32_t>(nslab) * 1024u; // bytes between the pieces of consecutive 16-row blocks
    // Strip k of this wave's PAIR (waves w and w + 4 share a SIMD and a strip sequence, see "work sharing"
    // below): unit (k >> 1) of the stream, tile (k & 1) of that unit, rows [64 (w & 3), +64) of the tile.
    auto unit_of = [&](uint32_t k) __attribute__((always_inline)) -> uint32_t { return stream + (k >> 1) * n_streams; };
    auto locate = [&](uint32_t k, Geo& g) __attribute__((always_inline)) {
        const uint32_t un_ = unit_of(k);
        const uint32_t sel = un_ < n_units ? 2u * un_ + (k & 1u) : 0xffffffffu;
        uint64_t row0 = past_end;
        if (sel < a.n_sel_tiles) {
            const uint32_t tile = SAMPLE ? sel * a.stride : sel + sel / (a.stride - 1u) + 1u;
            row0 = static_cast<uint64_t>(tile) * I8_ROWS + static_cast<uint32_t>((wid & 3) * 64);
        }
        g.row0 = row0;
        const uint64_t rowb = row0 < past_end ? row0 : past_end - 64; // (n_rows >= 4096 on this path)
        g.base = reinterpret_cast<const unsigned char*>(a.rows_i8) + (rowb / 16) * piece_row_stride;
    };
    // row piece rb of slab ss of the strip at `sbase` into ring stage P
    auto piece = [&](const unsigned char* sbase, int ss, int P, int rb) __attribute__((always_inline)) {
        lds_dma16_s(sbase + static_cast<uint32_t>(rb) * piece_row_stride + static_cast<uint32_t>(ss) * 1024u, lane16,
                    ringW + P * 4096 + rb * 1024);
    };
    typedef float f2_t __attribute__((ext_vector_type(2)));
    auto meta_ptr = [&](uint64_t row0) __attribute__((always_inline)) -> const float* {
        uint64_t blk = row0 / I8_BLOCK_ROWS;
        if (blk >= n_blocks) blk = n_blocks - 1; // a strip past the end: nothing of it is ever emitted
        return a.rows_i8_meta + 2ull * blk;
    };
    // {A_lo, B_hi} of this lane's eight queries: requested from inline asm (the compiler would drain the
    // ring with a vmcnt(0) of its own at a place of its choosing) at the end of a unit, when the second
    // set of query fragments is dead — sixteen registers that are free exactly then
    f2_t qthr[8];
    const float* qthr_p = a.q_thr + 2ull * (q0 + l15); // < q_pad: the table is padded
    bool zt_ready = false; // ZT: the resident copy exists only behind the prologue's barrier (the first request reads global memory)
    auto qthr_request = [&]() __attribute__((always_inline)) {
        if (ZT && zt_ready) { // 2: the tile's halves are resident in LDS — a 100-clock read, nothing for the drain to wait for
            const f2_t* lt = reinterpret_cast<const f2_t*>(lds + R_ZS_THR) + l15;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) qthr[cb] = lt[cb * 16];
            return;
        }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
            asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(qthr[cb]) : "v"(qthr_p), "n"(cb * 128) : "memory");
    };
    // L2: m_q of this lane's eight queries and a_r of its sixteen rows of a strip (one 16-byte load: byte r of word
    // rb = row 16 rb + 4 lq + r), requested and completed together with the thresholds — like them they live only
    // from the end of one strip to the accumulator set-up of the next, in registers the k loop has no use for then
    int qbias[8];
    i32x4v rbias;
    // (uniform base + 32-bit lane offset: no 64-bit per-lane address has to stay alive across the k loop)
    auto qbias_request = [&]() __attribute__((always_inline)) {
        const uint32_t* sb_ = a.i8_q_bias + q0; // < q_pad with the lane offsets
        const uint32_t voff = static_cast<uint32_t>(lane & 15) * 4u;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
            asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(qbias[cb]) : "v"(voff), "s"(sb_), "n"(cb * 64) : "memory");
    };
    auto rbias_request = [&](uint64_t row0) __attribute__((always_inline)) {
        uint64_t blk = row0 / I8_BLOCK_ROWS;
        if (blk >= n_blocks) blk = n_blocks - 1; // a strip past the end: nothing of it is ever emitted
        const uint8_t* sb_ = a.i8_row_bias + blk * 64u;
        const uint32_t voff = static_cast<uint32_t>(lane >> 4) * 16u;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rbias) : "v"(voff), "s"(sb_) : "memory");
    };
    auto qthr_wait = [&]() __attribute__((always_inline)) {
        if (L2)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(qthr[0]), "+v"(qthr[1]), "+v"(qthr[2]), "+v"(qthr[3]),
                                                "+v"(qthr[4]), "+v"(qthr[5]), "+v"(qthr[6]), "+v"(qthr[7]), "+v"(rbias),
                                                "+v"(qbias[0]), "+v"(qbias[1]), "+v"(qbias[2]), "+v"(qbias[3]),
                                                "+v"(qbias[4]), "+v"(qbias[5]), "+v"(qbias[6]), "+v"(qbias[7]) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(qthr[0]), "+v"(qthr[1]), "+v"(qthr[2]), "+v"(qthr[3]),
                                                "+v"(qthr[4]), "+v"(qthr[5]), "+v"(qthr[6]), "+v"(qthr[7]) :: "memory");
    };

    // ---- prologue: the resident query tile (wave w stages 16 queries of every slab), the first two
    //      slabs of the first unit, its thresholds ---------------------------------------------------
    {
        const int prow = lane >> 2;                        // the query of a piece this lane fetches 16 bytes of
        const int rowB = wid * 16 + prow;
        const uint32_t voffB = static_cast<uint32_t>(rowB) * 64u + ((lane & 3) ^ i8_swz(prow)) * 16u; // i8_swz(rowB) == i8_swz(prow)
        const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.q_i8) + static_cast<uint64_t>(q0) * 64;
        const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
        for (int s = 0; s < nslab; ++s)
            lds_dma16_s(baseB + s * qslab_bytes, voffB, __builtin_amdgcn_readfirstlane(lds0 + s * R_B_SLAB + wid * 1024));
    }
    if (ZT && tid < R_QUERIES) // the tile's threshold halves, resident (q_thr is padded to q_pad queries)
        reinterpret_cast<f2_t*>(lds + R_ZS_THR)[tid] = reinterpret_cast<const f2_t*>(a.q_thr)[q0 + static_cast<uint32_t>(tid)];
    // ---- work sharing ----------------------------------------------------------------------------------
    // The two waves of a SIMD do not run at the same speed: the arbiter favours the older one, which then
    // moves at the pace its own DMA latency allows while the younger one gets what is left.  With a fixed
    // strip per wave, waves 0-3 finished their streams 20 % ahead of waves 4-7 and every launch ended with
    // one wave per SIMD (alternating s_setprio made both slower).  So a pair (w, w + 4) draws its strips
    // from ONE counter in global memory (there is no LDS left): a returning atomic, requested two strips
    // before its value is needed and completed by a drain that is there anyway.  The same counters pace
    // the query tiles of a stream (below).
    uint32_t* const pair_cnt = a.i8_sync + (static_cast<uint64_t>(stream) * 4u + static_cast<uint32_t>(wid & 3)) * 32u;
    uint32_t take_v; // lane 0: the counter's value before this wave's increment
    auto take_request = [&]() __attribute__((always_inline)) {
        unsigned long long keep;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                     : "=&v"(take_v), "=&s"(keep) : "v"(pair_cnt + qt), "v"(1u) : "memory");
    };
    auto take_result = [&]() __attribute__((always_inline)) -> uint32_t { // after a vmcnt(0)
        asm volatile("" : "+v"(take_v));
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(take_v)));
    };
    uint32_t k_cur, k_nxt, k_fut; // the strip in the accumulators, the one whose first slabs are on their way, the one after
    take_request(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k_cur = take_result();
    take_request(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k_nxt = take_result();
    take_request(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k_fut = take_result();
    Geo cur, nxt;
    locate(k_cur, cur);
    if (!DIRECT) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) piece(cur.base, s, s, rb);
    }
    if (!SAMPLE) qthr_request();
    if (L2) { qbias_request(); rbias_request(cur.row0); }
    float sb, eb; // block scale and residue bound of the current strip (wave-uniform)
    {
        const float* mp = meta_ptr(cur.row0);
        const float m0 = mp[0], m1 = mp[1];
        sb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m0)));
        eb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m1)));
    }
    // SAMPLE: {t_q, c_q, f_q, 0} of the tile's 128 queries, resident in the LDS the direct form's ring does not use (the
    // epilogue reads a query block's four floats back: holding them in registers for the launch cost seven spill slots)
    if (SAMPLE && tid < R_QUERIES)
        reinterpret_cast<float4*>(lds + R_ZS_THR)[tid] = reinterpret_cast<const float4*>(a.q_meta)[q0 + static_cast<uint32_t>(tid)]; // < q_pad: the table is padded
    if (SAMPLE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else qthr_wait();
    if (ZT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the stash of the threshold halves)
    __builtin_amdgcn_s_barrier(); // the only one: the query tile is shared, everything after it is wave-private
    zt_ready = true;

    // fragment offsets: row block rb adds rb * 1 KiB inside a ring slab, query block cb adds cb * 1 KiB
    const int offF = l15 * 64 + ((lq ^ i8_swz(l15)) << 4);
    i32x4v acc[4][8];
    i32x4v fa[2][4], fb[2][4];
    auto ld = [&](const unsigned char* base, int off) __attribute__((always_inline)) -> i32x4v {
        if (ABL == 4) { i32x4v z = {off, 0, 0, 0}; return z; } // measurement build: no fragment reads
        return *reinterpret_cast<const i32x4v*>(base + off);
    };
    auto half = [&](const i32x4v (&A)[4], const i32x4v (&B)[4], int cb0, auto&& filler, bool first = false) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rb = i >> 2, c = i & 3;
            if (ABL != 2 && ABL != 4)
                acc[rb][cb0 + c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rb], B[c], acc[rb][cb0 + c], 0, 0, 0);
            else if (i == 0)
                asm volatile("" :: "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]));
            __builtin_amdgcn_sched_barrier(0);
            filler(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto pin4 = [&](i32x4v (&F)[4]) __attribute__((always_inline)) { asm volatile("" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3])); };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    // One slab (parity P = slab & 1 = its ring stage) of the current strip:
    //   half 1: fa[P] x query blocks 0-3; requests query blocks 4-7 of this slab;
    //   wait:   the slab one ahead has landed (only the four pieces of the slab two ahead are younger);
    //   half 2: fa[P] x query blocks 4-7; requests the next slab's row fragments (fa[P ^ 1], from stage P ^ 1)
    //           and its query blocks 0-3, and — as soon as those fragment reads have returned — refills that
    //           stage with the slab THREE ahead: a stage is empty for a third of a slab instead of a whole
    //           one, so a piece has 1.6 slabs to land instead of 1.2 (the ring is what bounds the loop: 8 KiB
    //           in flight per wave against ~1 us of loaded L2 -> LDS latency).
    // `sbase` / `ss` name the strip and slab the DMA pieces belong to, `sn` the next slab of the query
    // tile, `early` the first two slabs of a strip: slabs 1 and 2 landed before the strip began (the drain in
    // front of the thresholds), so they wait for no DMA, and the survivor stores of the previous strip's
    // epilogue — in the same in-order counter — get two slabs to complete before a counted wait looks at them.
    auto dfetch = [](i32x4v& dst, uint32_t vo, const unsigned char* src) __attribute__((always_inline)) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(vo), "s"(src) : "memory");
    };
    // DIRECT: `dnext` = the strip's piece of the NEXT slab (the next strip's slab 0 after the last one): its four row
    // fragments are requested under half 1, into the registers the previous slab has just left, and waited for at
    // the beginning of the next slab (`early`: the strip's first slab, whose fragments the drain at the end of the
    // previous strip has covered)
    auto body_impl = [&](int sn, bool early, const unsigned char* sbase, int ss, int s, auto par_tag, const unsigned char* dnext,
                         auto first_tag, auto&& extra) __attribute__((always_inline)) {
        constexpr int P = decltype(par_tag)::value;
        constexpr bool first = decltype(first_tag)::value;
        const unsigned char* bq = lds + s * R_B_SLAB;
        const unsigned char* bqn = lds + sn * R_B_SLAB;
        // (ZS: every slab waits — a strip's first fragments were requested in the previous strip's last slab and nothing
        // younger than them is in flight: no store, and the strip counters are requested a slab earlier)
        if (DIRECT && ABL != 1 && !early)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[P][0]), "+v"(fa[P][1]), "+v"(fa[P][2]), "+v"(fa[P][3]) :: "memory");
        pin4(fa[P]); pin4(fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        half(fa[P], fb[0], 0, [&](int i) __attribute__((always_inline)) {
            if (DIRECT && ABL != 1 && i < 4) dfetch(fa[P ^ 1][i & 3], static_cast<uint32_t>(offF), dnext + static_cast<uint32_t>(i & 3) * piece_row_stride);
            if (DIRECT ? (i >= 4 && i < 8) : i < 4) fb[1][i & 3] = ld(bq, offF + (4 + (i & 3)) * 1024);
        }, first);
        if (!DIRECT && ABL != 1 && !early) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pin4(fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        half(fa[P], fb[1], 4, [&](int i) __attribute__((always_inline)) {
            if (!DIRECT && i < 4) fa[P ^ 1][i < 4 ? i : 0] = ld(ring, (P ^ 1) * 4096 + offF + (i < 4 ? i : 0) * 1024);
            if (DIRECT ? i < 4 : (i >= 4 && i < 8)) fb[0][i & 3] = ld(bqn, offF + (i & 3) * 1024);
            if (!DIRECT && ABL != 1 && i == 10) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); // the row fragments have left stage P ^ 1
            if (!DIRECT && ABL != 1 && i >= 11 && i < 15) piece(sbase, ss, P ^ 1, (i - 11) & 3);
            extra(i);
        }, first);
        if (ABL == 10) { // measurement build: 20 more MFMAs per slab on the SAME fragments — the multiply-adds per row byte of a
                         // 208-query tile (13 query blocks against 8); with 640 queries (5 tiles, 30 of 32 CUs per XCD) a launch
                         // has the matrix work, the row traffic and the CU count of a 1040-query batch in that form
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                acc[i & 3][i >> 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[P][i & 3], fb[1][(i >> 2) & 3], acc[i & 3][i >> 2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto body = [&](int sn, bool early, const unsigned char* sbase, int ss, int s, auto par_tag, const unsigned char* dnext = nullptr) __attribute__((always_inline)) {
        body_impl(sn, early, sbase, ss, s, par_tag, dnext, std::false_type{}, [](int) {});
    };

    if (DIRECT) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) dfetch(fa[0][rb], static_cast<uint32_t>(offF), cur.base + static_cast<uint32_t>(rb) * piece_row_stride);
    } else {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) fa[0][rb] = ld(ring, offF + rb * 1024);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) fb[0][cb] = ld(lds, offF + cb * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!DIRECT) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) piece(cur.base, 2, 0, rb); // slab 2 into the stage slab 0 just left
    }
    if (DIRECT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]) :: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (a strip begins with its slabs 1 and 2 landed)

    // ---- pacing ------------------------------------------------------------------------------------------
    // Pair p of the n_qt workgroups of a row stream reads the same strips in the same order; they come from
    // HBM once only while those pairs stay within what this XCD's L2 keeps of the stream (4 MiB for four
    // streams: two to three units each).  Nothing else couples them — a survivor here, a slow store there —
    // and measured without pacing they drift apart until most strips are fetched again: 34 GB from HBM per
    // launch instead of 9.4.  So a wave looks at its siblings' strip counters at the end of a strip (the
    // load rides on the drain that is there anyway) and lets the slowest one come within R_WINDOW strips of
    // the one it just drew before it goes on.  Best effort: a bounded number of polls (then the wave stops
    // pacing altogether), no correctness depends on it, no workgroup waits for one that is not resident.
    constexpr uint32_t R_POLLS = 1024;
    bool pacing = true;
    const uint32_t R_WINDOW = window & 0xffffu;
    const uint32_t* sync_sib = pair_cnt + (static_cast<uint32_t>(lane) < n_qt ? static_cast<uint32_t>(lane) : qt);
#ifdef YAMS_ACCEL_MEASURE
    uint32_t units_read = 0;
#endif

    constexpr bool THR = (ABL == 0 || ABL == 8 || ABL == 9) && !SAMPLE;
    uint32_t q_live = 0; // bit cb: this lane's query of block cb exists (q0 + 16 cb + l15 < n_queries)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) q_live |= (q0 + static_cast<uint32_t>(cb * 16 + l15) < a.n_queries) ? 1u << cb : 0u;
    int nt[8]; // -T(this strip, query block cb): what the accumulators of the unit start at
    auto thresholds = [&]() __attribute__((always_inline)) {
        const float is = 1.0f / sb, g = eb * is; // (the same expressions as in i8_log_gather_kernel)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) nt[cb] = THR ? i8_neg_threshold(qthr[cb][0], is, qthr[cb][1], g) : 0;
    };
    thresholds();
    // Survivors go to ONE log region per wave and launch ((stream, query tile, wave): ~500 entries at the bench
    // shape), filled front to back: no per-strip region, count or memset, and the gather kernel gets 2048 dense
    // regions of one query tile each instead of 1.5 million mostly empty ones.
    const uint32_t log_region = (stream * n_qt + qt) * 8u + static_cast<uint32_t>(wid);
    const uint64_t region = static_cast<uint64_t>(log_region) * a.log_cap;
    uint32_t log_pos = 0;
    // ZS: the wave's survivor buffer in LDS and its flush into the log region (entries keep their order)
    uint64_t* const zs_key = reinterpret_cast<uint64_t*>(lds + R_ZS_LOG + wid * (R_ZS_ENTRIES * 12));
    uint32_t* const zs_q = reinterpret_cast<uint32_t*>(zs_key + R_ZS_ENTRIES);
    uint32_t zs_n = 0;
    auto zs_flush = [&]() __attribute__((always_inline)) {
        for (uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); i < zs_n; i += 64u) {
            const uint64_t key = zs_key[i];
            const uint32_t qi = zs_q[i];
            const uint32_t pos = log_pos + i;
            if (pos < a.log_cap) { a.log_key[region + pos] = key; a.log_q[region + pos] = qi; }
            else atomicOr(&a.q_over[qi], 1u); // no room: this query's list is incomplete -> exhaustive path
        }
        log_pos += zs_n;
        zs_n = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the k loop's waits count loads only)
    };

    // what a strip's accumulators start at: -T(strip, query block) (L2: - a_r m_q per element)
    auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[rb][cb][r] = L2 ? nt[cb] - __mul24(static_cast<int>((static_cast<uint32_t>(rbias[rb]) >> (8 * r)) & 255u), qbias[cb])
                                        : nt[cb];
    };
    if (EARLY) acc_init(); // (the first strip's; every later one is set up at the end of the strip before it, in front of the drain)
    if (unit_of(k_cur) < n_units) for (;;) {
        const uint32_t u = unit_of(k_cur);
        const bool more = unit_of(k_nxt) < n_units;
        locate(k_nxt, nxt); // (past the end of the stream: the spare DMA slots read the shard's last row; nobody consumes them)
        take_request();    // the strip after the next two; older than every piece of this strip
        unsigned long long meta_n;  // the next strip's block scale, on its way through the scalar cache
        {
            const float* mp = meta_ptr(nxt.row0);
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(meta_n) : "s"(mp) : "memory");
        }
        if (!EARLY) acc_init();
        uint32_t sib = 0; // pacing: the siblings' strip counters
        // two slabs per trip (the buffer parity is a compile-time constant); the pieces issued during the last
        // three slabs already belong to the next strip: uniform selects, not a second copy of the loop body
        int s = 0;
        do { // (nslab >= 4: a loop the compiler knows to run at least once keeps one register assignment)
            int so = s;
            asm volatile("" : "+s"(so)); // (opaque: the compiler must not peel the first trip off the loop for `early`)
            const bool early = so == 0;
            const bool n0 = s + 3 >= nslab, n1 = s + 4 >= nslab;
            body(s + 1, early, n0 ? nxt.base : cur.base, n0 ? s + 3 - nslab : s + 3, s, C0{}, cur.base + static_cast<uint32_t>(s + 1) * 1024u);
            body(s + 2 >= nslab ? 0 : s + 2, DIRECT ? false : early, n1 ? nxt.base : cur.base, n1 ? s + 4 - nslab : s + 4, s + 1, C1{},
                 s + 2 >= nslab ? nxt.base : cur.base + static_cast<uint32_t>(s + 2) * 1024u);
            s += 2;
        } while (s < nslab);
        asm volatile("" : "+s"(meta_n)); // (every slab waits lgkmcnt(0): the scalar load has long returned)
#ifdef YAMS_ACCEL_MEASURE
        ++units_read;
#endif
        const bool pace_now = (k_cur & pace_mask) == 0u;
        if (THR && n_qt > 1 && pace_now) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(sib) : "v"(sync_sib) : "memory");
        if (THR) qthr_request();  // for the NEXT unit's thresholds; in flight under the sign test below
        if (THR && L2) { qbias_request(); rbias_request(nxt.row0); }

        // ---- epilogue of the unit: acc[rb][cb][r] is row = row0 + 16 rb + 4 lq + r, query = q0 + 16 cb + l15;
        //      the accumulators hold I - T, a survivor is a non-negative one ---------------------------------
        const uint64_t strip = cur.row0;
        const float sb_cur = sb, eb_cur = eb;
        (void)sb_cur; (void)eb_cur;
        uint32_t hot = 0, k_new;
        auto emit_survivors = [&]() __attribute__((always_inline)) {
                // One pass over the query blocks that hold a survivor in some lane (one, typically): every element
                // is tested by the whole wave at once, a ballot hands out the slots of this strip's log region (no
                // LDS is left for a counter) and the lanes that hold a survivor store it.  An entry is
                // (accumulator, row) + the query; i8_log_gather_kernel turns the accumulator back into the score
                // bound u and moves the entry into its query's candidate list.
                const uint32_t rows_left = strip < a.n_rows ? static_cast<uint32_t>(a.n_rows - strip < 64 ? a.n_rows - strip : 64) : 0u;
                uint32_t base = ZL ? zs_n : log_pos; // entries of this wave so far (wave-uniform): ONE log region per wave and launch
    #pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const bool hot_cb = (hot >> cb) & 1u;
                    if (__builtin_amdgcn_ballot_w64(hot_cb) == 0) continue;
                    const uint32_t qi = q0 + cb * 16 + l15;
                    // this lane's 16 elements of the block that survive (straight-line code) ...
                    uint32_t pm = 0;
    #pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        const uint32_t off0 = 16 * rb + 4 * lq; // row of element r of this lane: strip + off0 + r
                        uint32_t mw = 0xfu;
                        if (a.row_mask) { const uint64_t rbase = strip + off0; mw = mask_word(a.row_mask, rbase & ~31ull, a.n_rows) >> (rbase & 31u); }
    #pragma unroll
                        for (int r = 0; r < 4; ++r)
                            pm |= (acc[rb][cb][r] >= 0 && off0 + r < rows_left && ((mw >> r) & 1u)) ? 1u << (4 * rb + r) : 0u;
                    }
                    if (!hot_cb) pm = 0;
                    // ... then one trip per survivor of the busiest lane (one, typically): compact code — the fully
                    // unrolled form (a store block per element, 35 KB of it) ran from a cold instruction cache every time
                    bool lost = false;
                    for (;;) {
                        const bool p = pm != 0;
                        const uint64_t m = __builtin_amdgcn_ballot_w64(p);
                        if (m == 0) break;
                        if (ZL && base + 64u > static_cast<uint32_t>(R_ZS_ENTRIES)) { zs_n = base; zs_flush(); base = 0; } // (a trip adds at most 64)
                        const int e = p ? __builtin_ctz(pm) : 0;
                        int val = acc[0][cb][0];
    #pragma unroll
                        for (int i = 1; i < 16; ++i) val = e == i ? acc[i >> 2][cb][i & 3] : val;
                        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                        base += static_cast<uint32_t>(__builtin_popcountll(m));
                        if (p && ABL != 8) {
                            const uint64_t row = strip + static_cast<uint32_t>(16 * (e >> 2) + 4 * lq + (e & 3));
                            if (ZL) {
                                zs_key[pos] = (static_cast<uint64_t>(static_cast<uint32_t>(val)) << 32) | static_cast<uint32_t>(row);
                                zs_q[pos] = qi;
                            } else if (ABL == 9) { // measurement build: slots, but no stores
                                if (pos == 0x7fffffffu && val == 1) a.list_count[0] = 1;
                            } else if (pos < a.log_cap) {
                                a.log_key[region + pos] = (static_cast<uint64_t>(static_cast<uint32_t>(val)) << 32) | static_cast<uint32_t>(row);
                                a.log_q[region + pos] = qi;
                            } else {
                                lost = true;
                            }
                        }
                        pm &= pm - 1u;
                    }
                    if (lost) atomicOr(&a.q_over[qi], 1u); // no room: this query's list is incomplete -> exhaustive path
                }
                if (ZL) zs_n = base; else log_pos = base;
                if (ABL == 8 && base == 0x12345u && a.list_cap == 0xffffffffu) a.list_count[0] = 1;
        };
        if (THR) {
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                int m = acc[0][cb][0];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = acc[rb][cb][r] > m ? acc[rb][cb][r] : m;
                if (m >= 0) hot |= 1u << cb;
            }
            hot &= q_live; // (queries past the end of the batch: ONE mask, not a per-block constant held in a register each)
            if (strip >= a.n_rows) hot = 0;
            if (EARLY && __builtin_amdgcn_ballot_w64(hot != 0) != 0) emit_survivors(); // (LDS only: nothing the drain below would wait for)
            // the next unit's thresholds (needs nothing of this unit's accumulators: eight registers)
            sb = __uint_as_float(static_cast<uint32_t>(meta_n));
            eb = __uint_as_float(static_cast<uint32_t>(meta_n >> 32));
            if (EARLY) {
                // thresholds (from the LDS copy of the halves) and the next strip's accumulators FIRST, the drain after them:
                // its wait — the next strip's first fragments, the strip counters — runs under these ~60 / ~160 instructions
                thresholds();
                acc_init();
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else { qthr_wait(); thresholds(); }
            k_new = take_result(); // (landed with the drain above, like the siblings' counters)
            if (n_qt > 1 && more && pacing && pace_now) {
                asm volatile("" : "+v"(sib));
                uint32_t polls = 0;
                for (; polls < R_POLLS; ++polls) {
                    if (__builtin_amdgcn_ballot_w64(sib + R_WINDOW < k_new) == 0) break;
                    __builtin_amdgcn_s_sleep(8);
                    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(sib) : "v"(sync_sib) : "memory");
                }
                // a sibling that did not move for ~1 ms is not resident (another kernel holds its CU): this wave
                // stops pacing for the rest of the launch rather than pay the timeout at every strip
                if (polls == R_POLLS) pacing = false;
            }
        } else if (SAMPLE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the drain of the plain form: strip counters, the next strip's first fragments
            k_new = take_result();
            const uint32_t sel_c = 2u * u + (k_cur & 1u);    // the sample tile of this strip
            if (sel_c < a.n_sel_tiles) {
                const float ninf = -__builtin_inff();
                const uint32_t gid = (sel_c * static_cast<uint32_t>(I8_ROWS) + static_cast<uint32_t>((wid & 3) * 64)) / 16u + static_cast<uint32_t>(lq);
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const uint32_t qi = q0 + static_cast<uint32_t>(cb * 16 + l15);
                    const float4 qm = reinterpret_cast<const float4*>(lds + R_ZS_THR)[cb * 16 + l15];
                    const float S = sb_cur * qm.x, K = fmaf(eb_cur, qm.y, qm.z); // (as in the half-tile kernel's MODE_SAMPLE)
                    float m = ninf;
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        const uint64_t rbase = strip + static_cast<uint32_t>(16 * rb + 4 * lq);
                        const uint32_t mw = a.row_mask ? mask_word(a.row_mask, rbase & ~31ull, a.n_rows) >> (rbase & 31u) : 0xfu;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float uu = fmaf(static_cast<float>(acc[rb][cb][r]), S, K);
                            m = fmaxf(m, (rbase + r < a.n_rows && ((mw >> r) & 1u)) ? uu : ninf);
                        }
                    }
                    if (qi < a.n_queries) a.gmax[static_cast<uint64_t>(qi) * a.n_groups + gid] = (m != m) ? 0xffffffffu : f2ord(m);
                }
            }
            sb = __uint_as_float(static_cast<uint32_t>(meta_n));
            eb = __uint_as_float(static_cast<uint32_t>(meta_n >> 32));
        } else { // measurement builds: keep the accumulators alive, emit nothing
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            k_new = take_result();
            int t = 0;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += acc[rb][cb][r];
            if (t == 123456789 && a.list_cap == 0xffffffffu) a.list_count[0] = 1;
        }
        if (!EARLY && THR && __builtin_amdgcn_ballot_w64(hot != 0) != 0) emit_survivors(); // about half of the strips hold a survivor somewhere
        if (!more) break;
        cur = nxt;
        k_cur = k_nxt; k_nxt = k_fut; k_fut = k_new;
    }
    if (ZL) zs_flush();
    // (the lane id is derived again: holding it over the launch cost the direct form its 256th register and a scratch slot)
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u && ABL == 0 && !SAMPLE) a.log_cnt[log_region] = log_pos < a.log_cap ? log_pos : a.log_cap;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the ring's last (unused) pieces must land before the LDS is handed back
#ifdef YAMS_ACCEL_MEASURE
    if (lane == 0 && n_qt <= 8) { // measurement build: when each wave started and ended (100 MHz ticks) and how many strips it took
        uint32_t* const dbg = a.i8_sync + (static_cast<uint64_t>(n_streams) * 4u + static_cast<uint64_t>(stream) * 8u + static_cast<uint32_t>(wid)) * 32u;
        dbg[8 + qt] = static_cast<uint32_t>(t_begin);
        dbg[16 + qt] = static_cast<uint32_t>(wall_clock64());
        dbg[24 + qt] = units_read;
    }
#endif
}



#include "scan_i8d_kernel.h" // the same form with two slabs of row fragments in flight per wave (dims 384 / 768), scan_tiles_i8d_kernel

#ifdef YAMS_ACCEL_MEASURE
#include "scan_i8q_kernel.h" // measurement build only: the 128 x 128 wave-tile form, scan_tiles_i8q_kernel
#endif

// The survivor log -> per-query candidate lists.  An entry carries the accumulator of a survivor (I - T; under L2
// I - T - a_r m_q) and its row; the score bound is u = s_b t_q I + e_b c_q + f_q with T re-derived exactly as the
// filter kernel derived it (same inputs, same instructions).  L2: T comes from the shard's thresholds meta, u from
// the shadow's own meta, the score is G(u, |x|^2) and only rows with G >= tau are kept (the integer test is a
// necessary condition only).
struct I8GatherL2 {
    const float* l2_meta;       // [blocks][2] thresholds meta (what the filter kernel saw)
    const uint8_t* row_bias; const uint32_t* q_bias; const float* rows_nsq; const float* tau; float eps;
};
template <bool L2>
__device__ __forceinline__ bool i8_entry_score(uint64_t e, uint32_t q, const float* rows_meta, const float* q_meta,
                                               const float* q_thr, const I8GatherL2& l2, uint32_t& row, float& u) {
    row = static_cast<uint32_t>(e);
    const int accv = static_cast<int>(static_cast<uint32_t>(e >> 32));
    const uint32_t blk = row / I8_BLOCK_ROWS;
    const float2 m = reinterpret_cast<const float2*>(rows_meta)[blk];
    const float4 qm = reinterpret_cast<const float4*>(q_meta)[q];
    const float2 qt = reinterpret_cast<const float2*>(q_thr)[q];
    if (!L2) {
        const float is = 1.0f / m.x;
        const int nt = i8_neg_threshold(qt.x, is, qt.y, m.y * is);
        u = fmaf(static_cast<float>(accv - nt), m.x * qm.x, fmaf(m.y, qm.y, qm.z));
        return true;
    }
    const float2 mt = reinterpret_cast<const float2*>(l2.l2_meta)[blk];
    const float is = 1.0f / mt.x;
    const int nt = i8_neg_threshold(qt.x, is, qt.y, mt.y * is);
    const uint32_t j = row & 63u;
    const int ar = l2.row_bias[static_cast<uint64_t>(blk) * 64u + ((j >> 2) & 3u) * 16u + (j >> 4) * 4u + (j & 3u)];
    const int dot = accv - nt + ar * static_cast<int>(l2.q_bias[q]);
    const float ub = fmaf(static_cast<float>(dot), m.x * qm.x, fmaf(m.y, qm.y, qm.z));
    const float nsq = l2.rows_nsq[row];
    if (!norm_in_range(nsq)) return false; // already in every list (i8_l2_add_special_kernel)
    u = i8_l2_bound(ub, nsq, l2.eps);
    return !(u < l2.tau[q]);
}

// Half-tile kernel's logs: one thread per log region.
template <bool L2>
__global__ __launch_bounds__(256) void i8_log_gather_kernel(const uint64_t* log_key, const uint32_t* log_q, const uint32_t* log_cnt,
                                                            uint32_t log_cap, uint64_t n_regions, const float* rows_meta,
                                                            const float* q_meta, const float* q_thr, uint32_t* list_count,
                                                            uint64_t* list, uint32_t list_cap, I8GatherL2 l2) {
    const uint64_t r = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    const uint32_t n = log_cnt[r];
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t q = log_q[r * log_cap + i];
        uint32_t row; float u;
        if (!i8_entry_score<L2>(log_key[r * log_cap + i], q, rows_meta, q_meta, q_thr, l2, row, u)) continue;
        const uint32_t pos = atomicAdd(&list_count[q], 1u);
        if (pos < list_cap) list[static_cast<uint64_t>(q) * list_cap + pos] = pack_key(u, row);
    }
}

// The resident-query kernel's logs: one region per (row stream, query tile, wave), i.e. all entries of a region
// belong to ONE 128-query tile.  A workgroup per region: the entries are counted per query in LDS, every query
// present reserves its list slots with ONE global atomic (the per-entry atomics of the form above piled 1000
// increments on each of 1024 addresses), then the entries are placed.
template <bool L2>
__global__ __launch_bounds__(256) void i8_log_gather_wave_kernel(const uint64_t* log_key, const uint32_t* log_q, const uint32_t* log_cnt,
                                                                 uint32_t log_cap, uint32_t n_qt, const float* rows_meta,
                                                                 const float* q_meta, const float* q_thr, uint32_t* list_count,
                                                                 uint64_t* list, uint32_t list_cap, I8GatherL2 l2) {
    __shared__ uint32_t hist[R_QUERIES], slot0[R_QUERIES];
    const uint32_t r = blockIdx.x;
    const uint32_t n = log_cnt[r];
    if (n == 0) return;
    const uint32_t q0 = ((r >> 3) % n_qt) * R_QUERIES;
    const int tid = threadIdx.x;
    if (tid < R_QUERIES) hist[tid] = 0u;
    __syncthreads();
    const uint64_t* keys = log_key + static_cast<uint64_t>(r) * log_cap;
    const uint32_t* qs = log_q + static_cast<uint64_t>(r) * log_cap;
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t q = qs[i];
        uint32_t row; float u;
        if (!L2 || i8_entry_score<L2>(keys[i], q, rows_meta, q_meta, q_thr, l2, row, u)) atomicAdd(&hist[q - q0], 1u);
    }
    __syncthreads();
    if (tid < R_QUERIES) {
        const uint32_t c = hist[tid];
        slot0[tid] = c ? atomicAdd(&list_count[q0 + tid], c) : 0u;
        hist[tid] = 0u;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t q = qs[i];
        uint32_t row; float u;
        if (!i8_entry_score<L2>(keys[i], q, rows_meta, q_meta, q_thr, l2, row, u)) continue;
        const uint32_t pos = slot0[q - q0] + atomicAdd(&hist[q - q0], 1u);
        if (pos < list_cap) list[static_cast<uint64_t>(q) * list_cap + pos] = pack_key(u, row);
    }
}

#ifdef YAMS_ACCEL_MEASURE
#include "scan_i8q_gather.h" // measurement build only: the block entries of scan_tiles_i8q_kernel -> candidate lists
#endif

// Sample rows that reach the threshold join the candidate lists (the int8 tier's form of collect_sample_kernel).
// The sample pass keeps only the maximum of every group of 16 rows (the rows one lane holds for a query block:
// 64 (gid >> 2) + 4 (gid & 3) + {0..3} + 16 {0..3} of the sample) — writing all scores densely was 800 MB and
// 0.13 ms per 1024-query batch of the bench shard.  tau is the tau_rank-th largest group maximum, so only about
// tau_rank groups per query reach it: their 16 scores are re-derived here, exactly (the same integer dot product,
// the same two fmaf as the sample pass), from the int8 shadow and the int8 query.
__global__ __launch_bounds__(256) void i8_collect_sample_kernel(const uint32_t* gmax, uint32_t n_groups, uint32_t n_queries,
                                                                const int8_t* rows_i8, const float* rows_meta, const int8_t* q_i8,
                                                                const float* q_meta, uint32_t q_pad, uint32_t dim,
                                                                uint32_t stride, uint64_t n_rows, const uint32_t* row_mask,
                                                                const float* tau, uint32_t* list_count, uint64_t* list,
                                                                uint32_t list_cap, const float* l2_rows_nsq, float l2_eps) {
    const uint32_t q = blockIdx.y;
    const float t = tau[q];
    const uint32_t* gm = gmax + static_cast<uint64_t>(q) * n_groups;
    const uint32_t nchunk = dim / 16;                    // 16-byte chunks per row
    const int lane = threadIdx.x & 63;
    const float4 qm = reinterpret_cast<const float4*>(q_meta)[q]; // {t_q, c_q, f_q, 0}
    // a wave looks at 64 groups at a time (one per lane); every group that reaches tau is then scored by the WHOLE
    // wave: lane l takes row l & 15 of the group and every fourth 16-byte chunk from chunk l >> 4 on
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u, wstep = gridDim.x * blockDim.x;
    for (uint32_t g0 = wave0; g0 < n_groups; g0 += wstep) {
        const uint32_t gl = g0 + static_cast<uint32_t>(lane);
        const bool hot = gl < n_groups && !(ord2f(gm[gl]) < t);
        for (uint64_t hm = __builtin_amdgcn_ballot_w64(hot); hm; hm &= hm - 1) {
            const uint32_t g = g0 + static_cast<uint32_t>(__builtin_ctzll(hm));
            const int i = lane & 15;
            const uint64_t sidx = static_cast<uint64_t>(g >> 2) * 64 + 4 * (g & 3) + 16 * (i >> 2) + (i & 3);
            const uint64_t row = (sidx / I8_ROWS) * stride * I8_ROWS + (sidx % I8_ROWS);
            const bool ok = row < n_rows && (!row_mask || ((row_mask[row >> 5] >> (row & 31u)) & 1u));
            const uint64_t rr = row < n_rows ? row : n_rows - 1;
            int dot = 0;
            for (uint32_t ch = static_cast<uint32_t>(lane >> 4); ch < nchunk; ch += 4) {
                const int4 xv = *reinterpret_cast<const int4*>(rows_i8 + i8_blocked_offset(rr, ch * 16, dim));
                const int4 qv = *reinterpret_cast<const int4*>(q_i8 + (static_cast<uint64_t>(ch >> 2) * q_pad + q) * 64 + (ch & 3u) * 16);
                dot = __builtin_amdgcn_sdot4(xv.x, qv.x, dot, false);
                dot = __builtin_amdgcn_sdot4(xv.y, qv.y, dot, false);
                dot = __builtin_amdgcn_sdot4(xv.z, qv.z, dot, false);
                dot = __builtin_amdgcn_sdot4(xv.w, qv.w, dot, false);
            }
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (lane < 16 && ok) {
                const float2 bm = reinterpret_cast<const float2*>(rows_meta)[row / I8_BLOCK_ROWS];
                const float S = bm.x * qm.x, K = fmaf(bm.y, qm.y, qm.z);   // (as in the sample pass's epilogue)
                float u = fmaf(static_cast<float>(dot), S, K);
                bool unbounded = false;
                if (l2_rows_nsq) { // (L2 batches: as in the sample pass's epilogue)
                    const float nsq = l2_rows_nsq[row];
                    unbounded = !norm_in_range(nsq);
                    u = i8_l2_bound(u, nsq, l2_eps);
                }
                if (!unbounded && !(u < t)) {
                    const uint32_t pos = atomicAdd(&list_count[q], 1u);
                    if (pos < list_cap) list[static_cast<uint64_t>(q) * list_cap + pos] = pack_key(u, static_cast<uint32_t>(row));
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// The INT8 shadow.  One wave per block of 64 rows.  A row is first scaled by its own largest
// component (so tiny and huge rows normalise without under- or overflow), normalised in fp32, and
// the block's scale is s_b = max |x~_i| / 127 over its usable rows.  e_b = the largest measured
// residue |x~ - s_b xi| of those rows, inflated for the fp32 evaluation of the sum of squares and for
// the distance between the fp32-normalised row and the true unit row.  Rows that are all zero or
// hold a non-finite component get an all-zero int8 row: the reference never returns them
// (:4258-4269), so they need no score.
// `first_row` / `n_rows` name the rows that changed: every 64-row block that intersects
// [first_row, first_row + n_rows) is rebuilt from its first row on (an append that starts inside a
// block re-quantises that block's earlier rows with the new common scale).
// -------------------------------------------------------------------------------------------------
// ---- the ROTATED layout (round 6) ----------------------------------------------------------------------------------------
// Rows whose energy sits in a few components (an embedding model's outlier dimensions, a power-law spectrum) quantise badly
// under ONE scale per block: the large components set the step, the many small ones round to nothing — measured residue
// 0.047 against 0.010 for isotropic rows, a bound too loose to prove anything (the tier escalated every query).  An
// orthogonal map of rows AND queries leaves every dot product and norm where it was and spreads the energy: R = H_B S_2 H_A S_1,
// S_1 / S_2 fixed pseudo-random sign flips, H_A / H_B Walsh-Hadamard transforms (scaled by 1 / sqrt(P)) over the first and the
// last P = 2^floor(log2 dim) components (two overlapping windows mix all of a dimension that is not a power of two; S_2 keeps
// the second transform from undoing the first: the Hadamard transform of a Walsh function is a spike).  Residue of the
// anisotropic bench corpus 0.047 -> 0.011, of rows with four outlier dimensions 0.045 -> 0.010; uniform components get WORSE
// (0.004 -> 0.010: the rotated components are Gaussian), so the layout is chosen per corpus from the measured residues of a
// sample (yams_scan_choose_i8_layout_device) and recorded in the view (yams_scan_corpus_t.i8_flags).
// Rounding: the butterflies are fp32 adds; s stages leave a relative error of at most s u in the 2-norm (each stage is
// sqrt(2) x orthogonal with componentwise relative error u), the scale adds 1.5 u per window: with P <= 4096 a rotated unit
// vector is within 1.75e-6 of the exact rotation.  The residues are MEASURED on the computed vectors; the distance between
// computed and exact rotation of row and query enters the bound as an absolute 1e-5 in the query's slop (prep_i8_kernel).
__device__ inline uint32_t rot_sign(uint32_t i, uint32_t salt) {  // bit 31: flip component i (salt 1 / 2: S_1 / S_2)
    uint32_t v = i * 2654435761u + salt * 0x9E3779B9u;
    v ^= v >> 15; v *= 2246822519u; v ^= v >> 13; v *= 3266489917u; v ^= v >> 16;
    return v << 31;
}
__device__ inline void wave_lds_fence() {   // LDS written by some lanes of this wave, read by others: order the accesses for the compiler too
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// One Walsh-Hadamard transform of P = 64 EPL floats at `w` (wave-private LDS) by one wave; sign_salt != 0: component i of the
// window (global index g0 + i) is flipped first.
template <int EPL>
__device__ inline void wave_fwht(float* w, int lane, uint32_t g0, uint32_t sign_salt, float scale) {
    float v[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const uint32_t idx = static_cast<uint32_t>(lane) * EPL + i;
        float x = w[idx];
        if (sign_salt) x = __uint_as_float(__float_as_uint(x) ^ rot_sign(g0 + idx, sign_salt));
        v[i] = x;
    }
#pragma unroll
    for (int len = 1; len < EPL; len <<= 1)
#pragma unroll
        for (int i = 0; i < EPL; ++i)
            if (!(i & len)) { const float a = v[i], b = v[i + len]; v[i] = a + b; v[i + len] = a - b; }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const float o = __shfl_xor(v[i], m);
            v[i] = (lane & m) ? o - v[i] : v[i] + o;
        }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < EPL; ++i) w[static_cast<uint32_t>(lane) * EPL + i] = v[i] * scale;
    wave_lds_fence();
}

// EPL = 0: the plain layout.  Else: rotated, P = 64 EPL.  block_stride > 1 / dry != nullptr: the residues of a SAMPLE of blocks
// only (nothing but dry[0] += e_b, dry[1] (uint64) += 1 is written).
template <int EPL>
__global__ __launch_bounds__(256) void shadow_build_i8_kernel(const float* rows, uint64_t first_block, uint64_t end_row,
                                                              uint32_t dim, int8_t* out_i8, float* out_meta, float rot_scale,
                                                              uint64_t block_stride, double* dry) {
    __shared__ float rinv_s[4][I8_BLOCK_ROWS]; // per wave, per row: the normalising factor, 0 for an unusable row
    extern __shared__ float rot_s[];             // rotated layout: [4 waves][dim] floats
    const uint64_t blk = first_block + (static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * block_stride;
    const uint64_t r0 = blk * I8_BLOCK_ROWS;
    if (r0 >= end_row) return;
    const int lane = threadIdx.x & 63;
    const int nr = static_cast<int>(end_row - r0 < I8_BLOCK_ROWS ? end_row - r0 : I8_BLOCK_ROWS);
    float* rinv = rinv_s[threadIdx.x >> 6];  // (wave-private; written and read by the same wave in program order)
    float* buf = rot_s + static_cast<size_t>(threadIdx.x >> 6) * dim;
    // rotated layout: the normalised row, rotated, into buf
    auto stage_row = [&](const float* src, float inv) {
        if constexpr (EPL != 0) {
            wave_lds_fence();
            for (uint32_t c = lane * 4; c < dim; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(src + c);
                float4 o;
                o.x = __uint_as_float(__float_as_uint(v.x * inv) ^ rot_sign(c + 0, 1));
                o.y = __uint_as_float(__float_as_uint(v.y * inv) ^ rot_sign(c + 1, 1));
                o.z = __uint_as_float(__float_as_uint(v.z * inv) ^ rot_sign(c + 2, 1));
                o.w = __uint_as_float(__float_as_uint(v.w * inv) ^ rot_sign(c + 3, 1));
                if (inv == 0.f) o = make_float4(0.f, 0.f, 0.f, 0.f);     // (an unusable row may hold NaN / inf)
                *reinterpret_cast<float4*>(buf + c) = o;
            }
            wave_lds_fence();
            constexpr uint32_t P = 64u * EPL;
            wave_fwht<EPL>(buf, lane, 0, 0, rot_scale);
            if (P < dim) wave_fwht<EPL>(buf + (dim - P), lane, dim - P, 2, rot_scale);
        }
    };
    float umax = 0.f;                   // largest |x~_i| of the block
    for (int rr = 0; rr < nr; ++rr) {
        const float* src = rows + (r0 + rr) * dim;
        float amax = 0.f; bool bad = false;
        for (uint32_t c = lane * 4; c < dim; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            bad = bad || !(m <= 3.4e38f) || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w; // inf or NaN component
            amax = fmaxf(amax, m);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
        bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        // (a subnormal largest component — squared norm < 1e-74 — would overflow 1 / amax: such a row is unusable like
        // an all-zero one; the reference never returns it under cosine (:4267-4269), under L2 it travels with the
        // rows whose norm is outside norm_in_range())
        const bool ok = !bad && amax >= 1.17549435e-38f;
        const float ia = ok ? 1.0f / amax : 0.f;
        float nsq = 0.f;
        for (uint32_t c = lane * 4; c < dim; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            const float x0 = v.x * ia, x1 = v.y * ia, x2 = v.z * ia, x3 = v.w * ia;
            nsq = fmaf(x0, x0, nsq); nsq = fmaf(x1, x1, nsq); nsq = fmaf(x2, x2, nsq); nsq = fmaf(x3, x3, nsq);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nsq += __shfl_xor(nsq, d);
        const float inv = ok ? ia * rsqrtf(nsq) : 0.f;  // x~ = x * inv (nsq in [1, dim]: no range trouble)
        if (lane == 0) rinv[rr] = inv;
        if constexpr (EPL == 0) {
            if (ok) umax = fmaxf(umax, rsqrtf(nsq));         // |x~|_max = (amax * ia) * rsqrt(nsq)
        } else if (ok) {
            stage_row(src, inv);
            float m = 0.f;
            for (uint32_t c = lane * 4; c < dim; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(buf + c);
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
            umax = fmaxf(umax, m);
        }
    }
    wave_lds_fence();   // (rinv)
    const bool any = umax > 0.f;
    const float sc = any ? umax / 127.0f : 1.0f;
    const float isc = any ? 127.0f / umax : 0.f;
    float emax = 0.f;
    for (int rr = 0; rr < nr; ++rr) {
        const float* src = rows + (r0 + rr) * dim;
        const float inv = rinv[rr];
        if constexpr (EPL != 0) stage_row(src, inv);
        float esq = 0.f;
        for (uint32_t c = lane * 4; c < dim; c += 256) {
            float x[4];
            if constexpr (EPL == 0) {
                const float4 v = *reinterpret_cast<const float4*>(src + c);
                x[0] = v.x * inv; x[1] = v.y * inv; x[2] = v.z * inv; x[3] = v.w * inv;
            } else {
                const float4 v = *reinterpret_cast<const float4*>(buf + c);
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            }
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float qf = rintf(x[e] * isc);
                qf = fminf(fmaxf(qf, -127.f), 127.f);
                const float d = fmaf(-sc, qf, x[e]);
                esq = fmaf(d, d, esq);
                packed |= (static_cast<uint32_t>(static_cast<int>(qf)) & 0xffu) << (8 * e);
            }
            if (!dry) *reinterpret_cast<uint32_t*>(out_i8 + i8_blocked_offset(r0 + rr, c, dim)) = inv != 0.f ? packed : 0u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) esq += __shfl_xor(esq, d);
        if (inv != 0.f) emax = fmaxf(emax, esq);
    }
    if (!dry)
        for (int rr = nr; rr < I8_BLOCK_ROWS; ++rr) // the padding rows of the last block: defined (zero), never emitted
            for (uint32_t c = lane * 4; c < dim; c += 256)
                *reinterpret_cast<uint32_t*>(out_i8 + i8_blocked_offset(r0 + rr, c, dim)) = 0u;
    if (lane == 0) {
        const float fd = static_cast<float>(dim);
        const float e = any ? sqrtf(emax) * (1.0f + (fd + 16.f) * 5.9604645e-8f) + (fd + 64.f) * 5.9604645e-8f : 0.f;
        if (dry) {
            if (e > 0.f) { atomicAdd(dry, static_cast<double>(e)); atomicAdd(reinterpret_cast<unsigned long long*>(dry) + 1, 1ull); }
        } else {
            out_meta[2 * blk] = sc;
            out_meta[2 * blk + 1] = e;
        }
    }
}

// stats[0] += sum of e_b, stats[1] (as uint64) += number of blocks, over blocks [first_block, end_block)
// (one atomic pair per workgroup: a per-block atomic on one address serialises the whole build)
__global__ __launch_bounds__(256) void i8_meta_stats_kernel(const float* meta, uint64_t first_block, uint64_t end_block,
                                                            double* stats) {
    __shared__ double ssum[256];
    __shared__ unsigned int scnt[256];
    double sum = 0.0; unsigned int cnt = 0;
    for (uint64_t b = first_block + blockIdx.x * 256ull + threadIdx.x; b < end_block; b += 256ull * gridDim.x) {
        const float e = meta[2 * b + 1];
        if (e > 0.f) { sum += static_cast<double>(e); ++cnt; }
    }
    ssum[threadIdx.x] = sum; scnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) { ssum[threadIdx.x] += ssum[threadIdx.x + st]; scnt[threadIdx.x] += scnt[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && scnt[0]) {
        atomicAdd(stats, ssum[0]);
        atomicAdd(reinterpret_cast<unsigned long long*>(stats) + 1, static_cast<unsigned long long>(scnt[0]));
    }
}

// Quantises the prepared (unit-norm) queries of a batch.  One workgroup per padded query:
// qi = rint(q~ / t_q), t_q = max |q~_i| / 127, written k-slab-major ([dim/64][q_pad][64] int8, the
// layout the filter's DMA pieces expect); meta[q] = {t_q, c_q, f_q, 0} with c_q >= |t_q qi| and
// f_q >= |q~ - t_q qi| + 1e-6 (the absolute slop covers the fp32 evaluation of the score bound and the
// fp64 -> fp32 rounding of the exact similarity).  Padding queries are zero with t_q = 1.
// raw (L2 batches): the queries are not unit vectors, the absolute slop scales with their norm.
// rot_p != 0: the corpus' shadow is in the ROTATED layout (shadow_build_i8_kernel): the query goes through the same map first
// (one workgroup, the Walsh-Hadamard stages through LDS), and its slop takes the distance between the computed and the exact
// rotations of row and query (1e-5 of the query's norm: see the bound above wave_fwht).
__global__ __launch_bounds__(256) void prep_i8_kernel(const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                                                      int8_t* q_i8, float* q_meta, int raw, uint32_t* zero_words, uint32_t n_zero_words,
                                                      uint32_t rot_p, float rot_scale) {
    const uint32_t q = blockIdx.x;
    // the filter launch's zero-initialised tables (log region counts, overflow marks, strip counters: scan_api.cpp "i8_zeroed")
    for (uint32_t i = q * 256u + threadIdx.x; i < n_zero_words; i += gridDim.x * 256u) zero_words[i] = 0u;
    __shared__ float red[256];
    extern __shared__ float qrot[];     // rotated layout: [dim]
    const bool live = q < nq;
    const float* src = qprep + static_cast<uint64_t>(q) * dim;
    if (rot_p) {
        for (uint32_t i = threadIdx.x; i < dim; i += 256) qrot[i] = live ? __uint_as_float(__float_as_uint(src[i]) ^ rot_sign(i, 1)) : 0.f;
        __syncthreads();
        for (int win = 0; win < 2; ++win) {
            if (win == 1 && rot_p >= dim) break;
            float* w = qrot + (win ? dim - rot_p : 0u);
            if (win) {
                for (uint32_t i = threadIdx.x; i < rot_p; i += 256) w[i] = __uint_as_float(__float_as_uint(w[i]) ^ rot_sign(dim - rot_p + i, 2));
                __syncthreads();
            }
            for (uint32_t len = 1; len < rot_p; len <<= 1) {
                for (uint32_t t = threadIdx.x; t < rot_p / 2; t += 256) {
                    const uint32_t i = ((t & ~(len - 1)) << 1) | (t & (len - 1));
                    const float a = w[i], b = w[i + len];
                    w[i] = a + b; w[i + len] = a - b;
                }
                __syncthreads();
            }
            for (uint32_t i = threadIdx.x; i < rot_p; i += 256) w[i] *= rot_scale;
            __syncthreads();
        }
        src = qrot;
    }
    float amax = 0.f;
    if (live)
        for (uint32_t i = threadIdx.x; i < dim; i += 256) amax = fmaxf(amax, fabsf(src[i]));
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    const float am = red[0];
    __syncthreads();
    const bool ok = live && am > 0.f && am < __builtin_inff(); // an invalid query was zeroed by prep_queries
    const float t = ok ? am / 127.0f : 1.0f;
    const float it = ok ? 127.0f / am : 0.f;
    float csq = 0.f, fsq = 0.f;
    for (uint32_t i = threadIdx.x; i < dim; i += 256) {
        const float x = live ? src[i] : 0.f;
        float qf = rintf(x * it);
        qf = fminf(fmaxf(qf, -127.f), 127.f);
        const float qq = t * qf;
        const float d = fmaf(-t, qf, x);
        csq = fmaf(qq, qq, csq);
        fsq = fmaf(d, d, fsq);
        q_i8[(static_cast<uint64_t>(i >> 6) * q_pad + q) * 64 + (i & 63)] = ok ? static_cast<int8_t>(static_cast<int>(qf)) : int8_t(0);
    }
    red[threadIdx.x] = csq;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const float csum = red[0];
    __syncthreads();
    red[threadIdx.x] = fsq;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float fd = static_cast<float>(dim);
        const float up = 1.0f + (fd + 16.f) * 5.9604645e-8f;
        q_meta[4 * q + 0] = t;
        q_meta[4 * q + 1] = ok ? sqrtf(csum) * up : 0.f;
        // (raw queries, rotated: |q| <= |t qi| + |q - t qi|, both measured on the rotated vector, which keeps |q| to 2e-6)
        const float unit = raw ? (rot_p ? (sqrtf(csum) + sqrtf(red[0])) * up * 1.00001f : sqrtf(csum) * up) : 1.0f;
        q_meta[4 * q + 2] = ok ? sqrtf(red[0]) * up + ((fd + 32.f) * 5.9604645e-8f + 1e-6f + (rot_p ? 1e-5f : 0.f)) * unit : 0.f;
        q_meta[4 * q + 3] = 0.f;
    }
}

// After the sample pass: the per-query halves of the integer thresholds of the filter pass,
//   A_lo <= (tau_q - f_q) / t_q,  B_hi >= c_q / t_q   (relative slack 2^-19 for the fp32 products the
// kernel forms with them).  No threshold (tau = -inf or NaN) keeps everything; padding queries keep
// nothing.
__global__ void i8_query_thresholds_kernel(const float* tau, const float* q_meta, uint32_t nq, uint32_t q_pad, float* q_thr) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q_pad) return;
    float A = __builtin_inff(), B = 0.f;
    if (q < nq) {
        const float t = q_meta[4 * q], c = q_meta[4 * q + 1], f = q_meta[4 * q + 2], ta = tau[q];
        if (ta != ta) A = -__builtin_inff();
        else {
            A = (ta - f) / t;
            A -= fabsf(A) * 1.9073486e-6f;
        }
        B = (c / t) * (1.0f + 1.9073486e-6f);
    }
    q_thr[2 * q] = A;
    q_thr[2 * q + 1] = B;
}

// ---- L2 on the int8 tier: the per-batch tables ("L2 on the int8 tier" at the top of this file) --------------
// stats words: [0] ~bits(smallest squared norm), [1] bits(largest), [2] bits(largest d_r / s_b), [3] rows whose
// squared norm is outside norm_in_range() (the first I8_L2_MAX_SPECIAL of them are listed in `special`), [4]
// bits(largest e_b).  All values are positive floats (their bit patterns order like the values), the buffer starts
// zeroed.  The statistics cover the rows inside the range only.
// (1) One wave per 64-row block, grid-stride.  Only the block's smallest norm needs a wave reduction per block
// (it is stored, and the spread of every row is measured against it); everything shard-wide is folded per lane
// over the wave's blocks, across the wave and the workgroup at the end, and then touches the shared words once.
// (a workgroup per CU, sixteen waves each: with 2048 small workgroups the kernel spent 0.1 ms queueing their
// atomics on the five shared words)
constexpr int L2S_WAVES = 16;
constexpr uint32_t I8_L2_MAX_SPECIAL = 64; // rows without a usable norm that a batch carries as unconditional candidates
__global__ __launch_bounds__(L2S_WAVES * 64) void i8_l2_norm_stats_kernel(const float* rows_nsq, const float* rows_meta, uint64_t n_rows,
                                                                         uint64_t n_blocks, float* nmin_out, uint32_t* stats,
                                                                         uint32_t* special) {
    __shared__ uint32_t red[L2S_WAVES][5];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float qlo = __builtin_inff(), qhi = 0.f, sp = 0.f, em = 0.f; // per lane: squared-norm range, largest d_r / s_b, largest e_b
    // four blocks per trip: their loads are in flight together (a trip costs one memory latency, and a wave has
    // ~20 blocks of a 10M-row shard)
    constexpr int NB = 4;
    const uint64_t step = static_cast<uint64_t>(gridDim.x) * L2S_WAVES;
    for (uint64_t blk0 = static_cast<uint64_t>(blockIdx.x) * L2S_WAVES + wid; blk0 < n_blocks; blk0 += step * NB) {
        float nsq[NB]; float2 m[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint64_t b = blk0 + j * step;
            const uint64_t r = b * I8_BLOCK_ROWS + static_cast<uint32_t>(lane);
            nsq[j] = (b < n_blocks && r < n_rows) ? rows_nsq[r] : 1.f; // (rows past the end are never "ok")
            m[j] = b < n_blocks ? reinterpret_cast<const float2*>(rows_meta)[b] : make_float2(1.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint64_t b = blk0 + j * step;
            if (b >= n_blocks) break;
            const bool live = b * I8_BLOCK_ROWS + static_cast<uint32_t>(lane) < n_rows;
            const bool ok = live && norm_in_range(nsq[j]);
            // rare: slots in the list of unbounded rows — one atomic per block that has any, and none at all once the
            // count is past the cap (a shard full of zero rows must not queue millions of atomics on one word)
            if (const uint64_t um = __builtin_amdgcn_ballot_w64(live && !ok)) {
                uint32_t base = 0u;
                if (lane == 0 && __atomic_load_n(&stats[3], __ATOMIC_RELAXED) <= I8_L2_MAX_SPECIAL)
                    base = atomicAdd(&stats[3], static_cast<uint32_t>(__builtin_popcountll(um)));
                else if (lane == 0) base = I8_L2_MAX_SPECIAL + 1u; // (already over: the batch leaves this tier, the list is moot)
                base = static_cast<uint32_t>(__shfl(static_cast<int>(base), 0));
                const uint32_t slot = base + static_cast<uint32_t>(__builtin_popcountll(um & ((1ull << lane) - 1ull)));
                if (live && !ok && slot < I8_L2_MAX_SPECIAL) special[slot] = static_cast<uint32_t>(b * I8_BLOCK_ROWS + static_cast<uint32_t>(lane));
            }
            const float n = sqrtf(nsq[j]);
            float mn = ok ? n : __builtin_inff();
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mn = fminf(mn, __shfl_xor(mn, d));
            if (ok) {
                qlo = fminf(qlo, nsq[j]); qhi = fmaxf(qhi, nsq[j]);
                sp = fmaxf(sp, i8_l2_spread(n, mn) / m[j].x);
            }
            em = fmaxf(em, m[j].y); // (e_b >= 0)
            if (lane == 0) nmin_out[b] = mn; // (+inf for a block without a usable row: the batch leaves this tier anyway)
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        qlo = fminf(qlo, __shfl_xor(qlo, d)); qhi = fmaxf(qhi, __shfl_xor(qhi, d));
        sp = fmaxf(sp, __shfl_xor(sp, d)); em = fmaxf(em, __shfl_xor(em, d));
    }
    if (lane == 0) {
        red[wid][0] = __float_as_uint(qlo); red[wid][1] = __float_as_uint(qhi); red[wid][2] = __float_as_uint(sp);
        red[wid][3] = 0u; red[wid][4] = __float_as_uint(em);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t lo = red[0][0], hi = red[0][1], s2 = red[0][2], e2 = red[0][4]; // (positive floats: bit patterns order like the values)
        for (int w = 1; w < L2S_WAVES; ++w) {
            lo = min(lo, red[w][0]); hi = max(hi, red[w][1]); s2 = max(s2, red[w][2]); e2 = max(e2, red[w][4]);
        }
        if (lo != 0x7f800000u) { atomicMax(&stats[0], ~lo); atomicMax(&stats[1], hi); atomicMax(&stats[2], s2); }
        atomicMax(&stats[4], e2);
    }
}

// (2) After the sample pass, one thread per (padded) query: q_thr = {A_lo, B}, q_bias = m_q.  fp64 for the line
// under h; the fp32 values the filter kernel multiplies carry the cosine tier's slack (A: 2^-19 downwards; B is
// rounded to nearest, its slack sits in the thresholds meta, i8_l2_rows_kernel).
__global__ void i8_l2_thresholds_kernel(const float* tau, const float* q_meta, uint32_t nq, uint32_t q_pad, float eps,
                                        const uint32_t* stats, float* q_thr, uint32_t* q_bias) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q_pad) return;
    float A = __builtin_inff(), B = 0.f;
    uint32_t mq = 0u;
    if (q < nq) {
        const float t = q_meta[4 * q], c = q_meta[4 * q + 1], f = q_meta[4 * q + 2], ta = tau[q];
        if (!(ta > -__builtin_inff())) A = -__builtin_inff(); // no threshold (or NaN): keep everything
        else {
            const double n_lo = sqrt(static_cast<double>(__uint_as_float(~stats[0])));
            const double n_hi = sqrt(static_cast<double>(__uint_as_float(stats[1])));
            const double W = static_cast<double>(i8_l2_unit(stats[2]));
            // |u| <= (1 + 2 e_b) c + f < 3 c + f; the eps term of G is at most eps (n |u| + n^2 / 2)
            const double u_max = 3.0 * c + f;
            const double tp = static_cast<double>(ta) - 1.5 * eps * (n_hi * u_max + 0.5 * n_hi * n_hi);
            double beta = 0.5 - tp / (n_lo * n_hi);
            if (!(beta >= 0.0)) beta = 0.0; // (the block and row parts use the smallest norms: the slope may not be negative)
            const double slope = 0.5 - beta; // phi(n) = h(n) - beta n = tp / n + slope n
            double alpha = fmin(tp / n_lo + slope * n_lo, tp / n_hi + slope * n_hi);
            if (tp > 0.0 && slope > 0.0) { // convex h: the stationary point of phi, when it lies inside the range
                const double ns = sqrt(tp / slope);
                if (ns > n_lo && ns < n_hi) alpha = fmin(alpha, 2.0 * sqrt(tp * slope));
            }
            alpha -= fabs(alpha) * 1e-12 + 1e-300;
            const double e_max = static_cast<double>(__uint_as_float(stats[4])) * (1.0 + 1e-6);
            A = static_cast<float>((alpha - static_cast<double>(f) - static_cast<double>(c) * e_max) / static_cast<double>(t));
            A -= fabsf(A) * 1.9073486e-6f;
            B = static_cast<float>(beta / static_cast<double>(t));
            const double m = floor(beta / static_cast<double>(t) * W * (1.0 - 1e-6));
            mq = m > 0.0 ? (m < 2097152.0 ? static_cast<uint32_t>(m) : 2097152u) : 0u;
            if (!(fabsf(A) <= 1e15f) || !(B <= 1e15f)) { A = -__builtin_inff(); B = 0.f; mq = 0u; } // far outside any sane range: keep everything rather than overflow a product
        }
    }
    q_thr[2 * q] = A;
    q_thr[2 * q + 1] = B;
    q_bias[q] = mq;
}

// (3) One wave per block: the thresholds meta {s_b, -nmin_b (rounded up)} and the row biases
// a_r = floor(d_r / (s_b W)), stored where the filter kernel's lanes fetch them: [block][lq][rb][r].
__global__ __launch_bounds__(256) void i8_l2_rows_kernel(const float* rows_nsq, const float* rows_meta, const float* nmin_in,
                                                         uint64_t n_rows, uint64_t n_blocks, const uint32_t* stats,
                                                         float* l2_meta, uint8_t* row_bias) {
    const uint64_t blk = static_cast<uint64_t>(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const float2 m = reinterpret_cast<const float2*>(rows_meta)[blk];
    float nmin = nmin_in[blk];
    if (!(nmin < __builtin_inff())) nmin = 0.f; // a block without a bounded row: nothing of it needs to pass (its rows are unconditional candidates)
    const float W = i8_l2_unit(stats[2]);
    const uint64_t r = blk * I8_BLOCK_ROWS + static_cast<uint32_t>(lane);
    uint32_t ar = 0u;
    if (r < n_rows && W > 0.f) {
        const float nsq = rows_nsq[r];
        const float d_over_s = norm_in_range(nsq) ? i8_l2_spread(sqrtf(nsq), nmin) / m.x : 0.f; // (as in i8_l2_norm_stats_kernel)
        const float v = floorf(d_over_s / W * (1.0f - 9.5367432e-7f));
        ar = static_cast<uint32_t>(fminf(fmaxf(v, 0.f), 255.f));
    }
    const uint32_t j = static_cast<uint32_t>(lane);
    row_bias[blk * 64u + ((j >> 2) & 3u) * 16u + (j >> 4) * 4u + (j & 3u)] = static_cast<uint8_t>(ar);
    if (lane == 0) {
        l2_meta[2 * blk] = m.x;
        l2_meta[2 * blk + 1] = -nmin * (1.0f - 4.0531158e-6f); // (2^-22 for the square root, 2^-18 for the products the filter kernel forms with it)
    }
}

// (4) The rows without a usable norm join every query's candidate list with the largest key (they are re-scored
// first; the reference's own rules then keep or drop them: a zero row is a valid L2 neighbour, a non-finite one is not).
__global__ void i8_l2_add_special_kernel(const uint32_t* special, uint32_t n_special, const uint32_t* row_mask,
                                         uint32_t n_queries, uint32_t* list_count, uint64_t* list, uint32_t list_cap) {
    const uint32_t q = blockIdx.x;
    if (q >= n_queries) return;
    for (uint32_t i = threadIdx.x; i < n_special; i += blockDim.x) {
        const uint32_t row = special[i];
        if (row_mask && !((row_mask[row >> 5] >> (row & 31u)) & 1u)) continue;
        const uint32_t pos = atomicAdd(&list_count[q], 1u);
        if (pos < list_cap) list[static_cast<uint64_t>(q) * list_cap + pos] = pack_key(__builtin_inff(), row);
    }
}

} // namespace yams_accel

#include "scan_launch.h"

namespace yams_accel {

ScanArgs make_scan_args(const ScanLaunch& L); // scan_kernels.hip

uint32_t i8_rotation_window(uint32_t dim) { // P = 2^floor(log2 dim); 0: the rotated layout is not offered for this dimension
    if (dim < 256 || dim > 4096) return 0;
    uint32_t p = 256;
    while (p * 2 <= dim) p *= 2;
    return p;
}
static float i8_rotation_scale(uint32_t p) { return static_cast<float>(1.0 / std::sqrt(static_cast<double>(p))); }

// dry != nullptr: the residues of every block_stride-th block only ({sum of e_b, count} at dry), nothing else is written
hipError_t launch_shadow_build_i8(hipStream_t st, const float* rows, uint64_t first_row, uint64_t n_rows, uint32_t dim,
                                  int8_t* out_i8, float* out_meta, double* stats, bool rotated, uint64_t block_stride, double* dry) {
    if (n_rows == 0) return hipSuccess;
    if (block_stride == 0) block_stride = 1;
    const uint64_t first_block = first_row / I8_BLOCK_ROWS;
    const uint64_t end_row = first_row + n_rows;
    const uint64_t n_blocks = ((end_row + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS - first_block + block_stride - 1) / block_stride;
    const dim3 grid(static_cast<uint32_t>((n_blocks + 3) / 4));
    const uint32_t p = rotated ? i8_rotation_window(dim) : 0;
    if (rotated && !p) return hipErrorInvalidValue;
    const float sc = p ? i8_rotation_scale(p) : 0.f;
    const size_t lds = p ? static_cast<size_t>(dim) * 16 : 0;
#define YA_BUILD_I8(EPL) hipLaunchKernelGGL((shadow_build_i8_kernel<EPL>), grid, dim3(256), lds, st, rows, first_block, end_row, dim, out_i8, out_meta, sc, block_stride, dry)
    switch (p / 64) {
        case 0: YA_BUILD_I8(0); break;
        case 4: YA_BUILD_I8(4); break;
        case 8: YA_BUILD_I8(8); break;
        case 16: YA_BUILD_I8(16); break;
        case 32: YA_BUILD_I8(32); break;
        case 64: {
            static bool attr = false;   // 64 KiB of dynamic LDS: above the default limit
            if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shadow_build_i8_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; }
            YA_BUILD_I8(64); break;
        }
        default: return hipErrorInvalidValue;
    }
#undef YA_BUILD_I8
    if (stats && !dry) {
        uint32_t g = static_cast<uint32_t>((n_blocks + 255) / 256);
        if (g > 256) g = 256;
        hipLaunchKernelGGL(i8_meta_stats_kernel, dim3(g), dim3(256), 0, st, out_meta, first_block, first_block + n_blocks, stats);
    }
    return hipGetLastError();
}

hipError_t launch_prep_i8(hipStream_t st, const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                          int8_t* q_i8, float* q_meta, bool raw_queries, uint32_t* zero_words, uint64_t n_zero_words, bool rotated) {
    if (q_pad == 0) return hipSuccess;
    if (n_zero_words >> 32) return hipErrorInvalidValue;
    const uint32_t p = rotated ? i8_rotation_window(dim) : 0;
    if (rotated && !p) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prep_i8_kernel, dim3(q_pad), dim3(256), p ? static_cast<size_t>(dim) * 4 : 0, st, qprep, nq, q_pad, dim, q_i8, q_meta, raw_queries ? 1 : 0,
                       zero_words, zero_words ? static_cast<uint32_t>(n_zero_words) : 0u, p, p ? i8_rotation_scale(p) : 0.f);
    return hipGetLastError();
}

float i8_l2_eps(uint32_t dim) { return (static_cast<float>(dim) + 64.f) * 5.9604645e-8f; }

uint32_t i8_l2_max_special() { return I8_L2_MAX_SPECIAL; }

hipError_t launch_i8_l2_add_special(hipStream_t st, const ScanLaunch& L, const uint32_t* special, uint32_t n_special) {
    if (n_special == 0 || L.plan.n_queries == 0) return hipSuccess;
    hipLaunchKernelGGL(i8_l2_add_special_kernel, dim3(L.plan.n_queries), dim3(64), 0, st, special, n_special, L.row_mask,
                       L.plan.n_queries, L.list_count, L.list, L.plan.list_cap);
    return hipGetLastError();
}

hipError_t launch_i8_l2_norm_stats(hipStream_t st, const float* rows_nsq, const float* rows_i8_meta, uint64_t n_rows,
                                   float* nmin, uint32_t* stats, uint32_t* special) {
    const uint64_t n_blocks = (n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
    if (n_blocks == 0) return hipSuccess;
    const uint64_t want = (n_blocks + L2S_WAVES - 1) / L2S_WAVES;
    hipLaunchKernelGGL(i8_l2_norm_stats_kernel, dim3(static_cast<uint32_t>(want < 256 ? want : 256)), dim3(L2S_WAVES * 64), 0, st,
                       rows_nsq, rows_i8_meta, n_rows, n_blocks, nmin, stats, special);
    return hipGetLastError();
}

hipError_t launch_i8_l2_thresholds(hipStream_t st, const float* tau, const float* q_meta, uint32_t nq, uint32_t q_pad,
                                   uint32_t dim, const uint32_t* stats, float* q_thr, uint32_t* q_bias) {
    if (q_pad == 0) return hipSuccess;
    hipLaunchKernelGGL(i8_l2_thresholds_kernel, dim3((q_pad + 255) / 256), dim3(256), 0, st, tau, q_meta, nq, q_pad,
                       i8_l2_eps(dim), stats, q_thr, q_bias);
    return hipGetLastError();
}

hipError_t launch_i8_l2_rows(hipStream_t st, const float* rows_nsq, const float* rows_i8_meta, const float* nmin,
                             uint64_t n_rows, const uint32_t* stats, float* l2_meta, uint8_t* row_bias) {
    const uint64_t n_blocks = (n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(i8_l2_rows_kernel, dim3(static_cast<uint32_t>((n_blocks + 3) / 4)), dim3(256), 0, st, rows_nsq,
                       rows_i8_meta, nmin, n_rows, n_blocks, stats, l2_meta, row_bias);
    return hipGetLastError();
}

hipError_t launch_i8_thresholds(hipStream_t st, const float* tau, const float* q_meta, uint32_t nq, uint32_t q_pad,
                                float* q_thr) {
    if (q_pad == 0) return hipSuccess;
    hipLaunchKernelGGL(i8_query_thresholds_kernel, dim3((q_pad + 255) / 256), dim3(256), 0, st, tau, q_meta, nq, q_pad, q_thr);
    return hipGetLastError();
}

// Does the filter pass of this launch run with the query tile resident (scan_tiles_i8r_kernel)?
// Needs the whole 128-query tile in LDS next to the row rings (dim <= 768), an even slab count, at most
// one query tile per CU of an XCD, and enough 512-row units per row stream that the streams end together.
struct ResidentPlan { bool use; uint32_t n_units, n_qt, n_streams, grid; };
static ResidentPlan i8_resident_plan(const ScanLaunch& L) {
    static const int n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        return v;
    }();
    ResidentPlan r{false, 0, 0, 0, 0};
    const uint32_t dim = L.plan.dim, nq = L.plan.n_queries;
    if (n_cu < 8 || (dim & 127u) || dim < 256 || dim > static_cast<uint32_t>(R_MAX_SLABS * I8_SLAB)) return r;
    const uint32_t per_xcd = static_cast<uint32_t>(n_cu) / 8u;
    r.n_qt = (nq + R_QUERIES - 1) / R_QUERIES;
    if (r.n_qt == 0 || r.n_qt > per_xcd) return r;
    if (L.i8_form == 1) return r;                                  // the caller keeps the per-tile kernels
    const uint32_t streams_x = per_xcd / r.n_qt;
    r.n_streams = streams_x * 8u;
    r.n_units = (L.plan.n_filter_tiles + 1u) / 2u;
    if (r.n_units == 0) return r;
    if (L.i8_form != 2) { // the library's choice (the caller's RESIDENT_QUERIES flag skips these)
        if (streams_x * r.n_qt * 10u < per_xcd * 9u) return r;   // more than a tenth of the CUs would idle
        if (r.n_units < 12u * r.n_streams) return r;              // short streams: the half-tile kernel balances better
        // measured on 12.5M-row shards (scripts/dbg/c2_forms.py, step times resident / half tiles): dim 384
        // 5.96 / 6.09 ms, 512 6.39 / 7.23, 640 7.5 / 8.6, 768 8.75 / 9.7 at 1024 queries; 768 with 256 queries
        // 2.70 / 3.0, 384: 3.8 / 5.1, 512: 4.8 / 5.3; sixteen sibling workgroups per stream (2048 queries) lose:
        // 20.9 / 18.1
        if (dim < 384 || r.n_qt > 8) return r;
    }
    r.grid = per_xcd * 8u;
    r.use = true;
    return r;
}

// Would the filter pass of this launch take the resident-query form?  (scan_api.cpp: batches of <= 128 queries
// take the int8 tier only then — the persistent kernel streams the int8 shadow at 6.1 TB/s, 1.55 ms for the
// 12.5M x 768 shard against 3.1 ms for the narrow bf16 form over the twice as large bf16 shadow; on shards too
// small for it the narrow bf16 form stays ahead of int8 half tiles.)
bool i8_takes_resident_form(const ScanLaunch& L) { return i8_resident_plan(L).use; }

#ifdef YAMS_ACCEL_MEASURE
bool i8_takes_q_form(const ScanLaunch& L, int version) {
    if (L.i8_l2 || L.plan.dim % (64u * Q_DEPTH) != 0 || L.plan.dim < 128u * Q_DEPTH) return false;
    if (!i8_resident_plan(L).use) return false;
    return version >= 70 && version <= 72;
}
uint32_t i8_log_entry_bytes(const ScanLaunch& L) { return L.i8_q_form ? Q_BLK_DWORDS * 4u : 8u; }
#else // the product has one form of log entry and no 128 x 128 wave-tile kernel
bool i8_takes_q_form(const ScanLaunch&, int) { return false; }
uint32_t i8_log_entry_bytes(const ScanLaunch&) { return 8u; }
#endif

// survivor-log regions of the filter launch: one per (workgroup, wave) of the half-tile kernel, one per
// (unit, query tile, wave) of the resident-query kernel — a 64 x 128 wave tile either way
uint64_t i8_log_regions(const ScanLaunch& L) {
    const ResidentPlan r = i8_resident_plan(L);
    if (r.use && L.i8_q_form) return static_cast<uint64_t>(r.n_streams) * r.n_qt * 4u;
    if (r.use) return static_cast<uint64_t>(r.n_streams) * r.n_qt * 8u;
    return static_cast<uint64_t>((2u * L.plan.n_filter_tiles + 7) / 8) * L.plan.n_qtiles * 8u * 4u;
}

hipError_t launch_i8_collect_sample(hipStream_t st, const ScanLaunch& L) {
    if (L.plan.sample_rows == 0 || L.plan.n_groups == 0) return hipSuccess;
    uint32_t gx = (L.plan.n_groups + 255) / 256;
    if (gx > 64) gx = 64;
#ifdef YAMS_ACCEL_MEASURE
    if (const char* e = std::getenv("YAMS_ACCEL_COLLECT_GX")) { const uint32_t v = static_cast<uint32_t>(std::atoi(e)); if (v >= 1 && v < gx) gx = v; }
#endif
    hipLaunchKernelGGL(i8_collect_sample_kernel, dim3(gx, L.plan.n_queries), dim3(256), 0, st, L.gmax, L.plan.n_groups,
                       L.plan.n_queries, L.rows_i8, L.rows_i8_meta, L.q_i8, L.q_meta, L.q_pad, L.plan.dim, L.plan.sample_stride,
                       L.plan.n_rows, L.row_mask, L.tau, L.list_count, L.list, L.plan.list_cap,
                       L.i8_l2 ? L.rows_nsq : nullptr, L.l2_eps);
    return hipGetLastError();
}

uint64_t i8_sync_words(const ScanLaunch& L) {
    const ResidentPlan r = i8_resident_plan(L);
    // [n_streams][4 pairs][32] strip counters (+ [n_streams][8][32] for the measurement build's timestamps), twice: the
    // filter launch's, then those of a sample pass in the resident form
    return r.use ? static_cast<uint64_t>(r.n_streams) * 12u * 32u * 2u : 0u;
}

// entries per log region.  Half tiles: a wave tile is 64 rows x 128 queries and the threshold admits about
// tau_rank * stride rows per query: 4x the expectation + 16 (12.5M rows: 0.6 expected, 16 slots; a 300k-row
// shard: 16 expected, 80).  Resident queries: everything a wave finds in a launch, 4x its share of the
// expected nq * tau_rank * stride survivors + 256 (waves draw 60-140 % of the mean number of strips).  A region
// that still overflows marks its queries (q_over) and they take the exhaustive path.
uint32_t i8_log_capacity(const ScanLaunch& L) {
    const ScanPlan& p = L.plan;
    const ResidentPlan r = i8_resident_plan(L);
    if (r.use && L.i8_q_form) {
        // block entries: one per (lane, query block) that holds a survivor — about one per survivor, but found BEFORE
        // the allow-mask is applied (the gather kernel applies it): as many more as the mask lets rows through
        const double total = static_cast<double>(p.n_queries) * p.tau_rank * p.sample_stride * (L.i8_mask_inflation < 8.0 ? L.i8_mask_inflation : 8.0);
        const double share = total / (static_cast<double>(r.n_streams) * r.n_qt * 4.0);
        const double cap = 4.0 * share + 512.0;
        const uint64_t c = static_cast<uint64_t>(cap < 1.0e6 ? cap : 1.0e6);
        return static_cast<uint32_t>((c + 15) / 16 * 16);
    }
    if (r.use) {
        const double total = static_cast<double>(p.n_queries) * p.tau_rank * p.sample_stride;
        const double share = total / (static_cast<double>(r.n_streams) * r.n_qt * 8.0);
        const double cap = (L.i8_l2 ? 8.0 : 4.0) * share + 256.0; // (L2: the integer test lets through somewhat more than reach tau)
        const uint64_t c = static_cast<uint64_t>(cap < 4.0e6 ? cap : 4.0e6);
        return static_cast<uint32_t>((c + 15) / 16 * 16);
    }
    const double per_wave = 8192.0 * p.tau_rank * p.sample_stride / static_cast<double>(p.n_rows ? p.n_rows : 1) * (L.i8_l2 ? 2.0 : 1.0);
    const uint32_t c = static_cast<uint32_t>(per_wave * 4.0 < 8192.0 ? per_wave * 4.0 : 8192.0) + 16;
    return (c + 15) / 16 * 16 < 8192u ? (c + 15) / 16 * 16 : 8192u;
}

hipError_t launch_i8_log_gather(hipStream_t st, const ScanLaunch& L) {
    const uint64_t regions = i8_log_regions(L);
    if (regions == 0) return hipSuccess;
    const ResidentPlan rp = i8_resident_plan(L);
    const I8GatherL2 l2{L.i8_l2_meta, L.i8_row_bias, L.i8_q_bias, L.rows_nsq, L.tau, L.l2_eps};
#ifdef YAMS_ACCEL_MEASURE
    if (rp.use && L.i8_q_form) {
        hipLaunchKernelGGL(i8_log_gather_blocks_kernel, dim3(static_cast<uint32_t>(regions)), dim3(256), 0, st,
                           reinterpret_cast<const int32_t*>(L.log_key), L.log_cnt, L.log_cap, rp.n_qt, L.rows_i8_meta, L.q_meta, L.q_thr,
                           L.plan.n_rows, L.row_mask, L.list_count, L.list, L.plan.list_cap);
        return hipGetLastError();
    }
#endif
    if (rp.use) {
        if (L.i8_l2)
            hipLaunchKernelGGL((i8_log_gather_wave_kernel<true>), dim3(static_cast<uint32_t>(regions)), dim3(256), 0, st,
                               L.log_key, L.log_q, L.log_cnt, L.log_cap, rp.n_qt, L.rows_i8_meta, L.q_meta, L.q_thr, L.list_count,
                               L.list, L.plan.list_cap, l2);
        else
            hipLaunchKernelGGL((i8_log_gather_wave_kernel<false>), dim3(static_cast<uint32_t>(regions)), dim3(256), 0, st,
                               L.log_key, L.log_q, L.log_cnt, L.log_cap, rp.n_qt, L.rows_i8_meta, L.q_meta, L.q_thr, L.list_count,
                               L.list, L.plan.list_cap, l2);
        return hipGetLastError();
    }
    if (L.i8_l2)
        hipLaunchKernelGGL((i8_log_gather_kernel<true>), dim3(static_cast<uint32_t>((regions + 255) / 256)), dim3(256), 0, st,
                           L.log_key, L.log_q, L.log_cnt, L.log_cap, regions, L.rows_i8_meta, L.q_meta, L.q_thr, L.list_count, L.list,
                           L.plan.list_cap, l2);
    else
        hipLaunchKernelGGL((i8_log_gather_kernel<false>), dim3(static_cast<uint32_t>((regions + 255) / 256)), dim3(256), 0, st,
                           L.log_key, L.log_q, L.log_cnt, L.log_cap, regions, L.rows_i8_meta, L.q_meta, L.q_thr, L.list_count, L.list,
                           L.plan.list_cap, l2);
    return hipGetLastError();
}

// Resident-query form: row fragments through the LDS ring (false) or straight into registers (true; DIRECT)?
// Measured on 12.5M-row shards (scripts/dbg/direct_sweep.sh: filter launch in ms, ring / direct, mean of three
// alternating runs of six launches each):
//   1024 queries (8 query tiles per stream): dim 256 4.43 / 4.42, 384 5.01 / 4.64, 512 5.43 / 5.45, 640 6.37 / 6.45,
//                                            768 7.53 / 7.45;
//    256 queries (2 query tiles per stream): dim 256 1.28 / 1.28, 384 1.34 / 1.36, 512 1.69 / 1.73, 640 2.04 / 2.04,
//                                            768 2.37 / 2.38.
// Bytes from HBM per launch at 768 x 1024: 17.4 GB / 10.2 GB (algorithmic 9.6: the ring runs three slabs ahead into
// the next strip and the sibling workgroups of a stream drift further apart).  So: direct when a stream has four or
// more query tiles — the rows then come from L2 seven times out of eight and one slab of latency cover is enough —
// and the ring for the small batches whose rows come from HBM.  That rule still stands for L2 batches (round 4, config 3:
// 6.21 / 6.20 ms at 1024 queries, 1.25 / 1.24 at 64).  The COSINE direct form has since moved its thresholds and its
// survivor buffer into the LDS the rings had used and looks at its siblings every 4th strip only, and wins at every
// shape measured (round 4, same script, ring / direct): 64 queries dim 256 1.79 / 1.78, 384 0.82 / 0.77, 512 1.10 /
// 1.08, 640 1.33 / 1.30, 768 1.58 / 1.54; 256 queries 1.29 / 1.29, 1.36 / 1.29, 1.69 / 1.63, 2.03 / 1.99, 2.38 / 2.30;
// 384 queries 2.30 / 2.28, 1.86 / 1.80, 2.34 / 2.25, 2.82 / 2.71, 3.32 / 3.17 — cosine batches always take it.
static bool i8r_direct_rows(uint32_t dim, uint32_t n_qt, bool l2) { (void)dim; return l2 ? n_qt >= 4 : true; }

// Half tiles (128 rows x 256 queries), XCD-aware block -> tile map as for the bf16 tier.  version:
// measurement build only — 40 = half tiles where the library would pick the resident-query form;
// 41..48 ablations of the half-tile kernel (41 no DMA refills after the prologue, 42 no MFMAs, 43 neither,
// 44 no fragment reads, 45 / 46 row / query pieces only, 47 no epilogue, 48 no emission); 61..68 of the
// resident-query kernel (61 no refills, 62 no MFMAs, 64 no fragment reads, 66 no survivor stores, 67 no
// epilogue, 68 no emission).
hipError_t launch_scan_i8(hipStream_t st, const ScanLaunch& L, int mode, int version) {
    (void)version;
    ScanArgs a = make_scan_args(L);
    a.n_sel_tiles = mode == MODE_SAMPLE ? L.plan.n_sample_tiles : L.plan.n_filter_tiles;
    if (a.n_sel_tiles == 0) return hipSuccess;
    const uint32_t hgrid = ((2u * a.n_sel_tiles + 7) / 8) * a.n_qtiles * 8;
    const ResidentPlan rp = i8_resident_plan(L);
    if (mode == MODE_SAMPLE) {
        // Cosine batches whose filter pass takes the resident-query form sample in it too, when every row stream gets at
        // least four units of sample tiles (the bench: 763 sample tiles = 382 units over 32 streams per query tile; a
        // batch of one query tile has 256 streams and keeps the half-tile kernel, as do short shards and L2 batches)
        const uint32_t s_units = (L.plan.n_sample_tiles + 1u) / 2u;
        // (a caller that forces the resident form — YAMS_SCAN_FLAG_RESIDENT_QUERIES, the tests' small shards — gets its
        //  sample pass in that form whatever the stream length)
        bool resident_sample = rp.use && !L.i8_l2 && L.i8_sync && L.plan.sample_stride >= 2 && !L.i8_sample_small_grid &&
                               (L.i8_form == 2 ? s_units >= 1u : s_units >= 4u * rp.n_streams);
#ifdef YAMS_ACCEL_MEASURE
        if (const char* sv = std::getenv("YAMS_ACCEL_I8R_SAMPLE")) resident_sample = resident_sample && std::atoi(sv) != 0;
        if (version != 2 && version != 3 && version != 0) resident_sample = false; // (the A/B kernel forms keep the half-tile sample pass)
#endif
        if (resident_sample) {
            a.i8_sync = L.i8_sync + i8_sync_words(L) / 2u; // its own strip counters (zeroed with the filter's)
            hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, false, true, 0, true>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, s_units, rp.n_qt, rp.n_streams, 1u);
        } else if (L.i8_l2) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_SAMPLE, 0, YAMS_SCAN_L2>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_SAMPLE>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        return hipGetLastError();
    }
    bool direct = i8r_direct_rows(L.plan.dim, rp.n_qt, L.i8_l2);
#ifdef YAMS_ACCEL_MEASURE
    if (const char* dv = std::getenv("YAMS_ACCEL_I8R_DIRECT")) direct = std::atoi(dv) != 0;
#endif
    if (L.i8_l2) { // L2 batches: the same two kernel forms with the row / query biases (no measurement forms)
        a.rows_i8_meta = L.i8_l2_meta; // the thresholds meta of the shard; the shadow's own meta is the gather kernel's
        // (the siblings' counters every 4th strip in the direct form, as for cosine below: config 3, 1024 queries, every strip /
        //  4th / 8th: launch 6.17-6.18 / 6.05-6.15 / 6.16 ms)
        uint32_t l2_pace_log2 = direct ? 2u : 0u;
#ifdef YAMS_ACCEL_MEASURE
        if (const char* pv = std::getenv("YAMS_ACCEL_I8R_L2_PACE_LOG2")) l2_pace_log2 = static_cast<uint32_t>(std::atoi(pv)) & 31u;
#endif
        const uint32_t l2_window = 1u | (l2_pace_log2 << 16);
        if (rp.use && direct) hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, true, true>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, l2_window);
        else if (rp.use) hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, true>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, 1u);
        else hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 0, YAMS_SCAN_L2>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        return hipGetLastError();
    }
    // the siblings' counters are looked at every 2^pace_log2-th strip (see the kernel).  Direct form, 12.5M rows, 1024 queries,
    // every strip / 4th / 8th / 32nd / never: dim 768 7.55 / 7.53 / 7.51 / 7.55 / 7.66 ms and 11.2 / 12.4 / 13.2 / 15.2 / 25.0 GB
    // from HBM; dim 384 4.79 / 4.32 / 4.26 / 4.27 / 4.26 ms and 4.8 GB throughout (scripts/dbg/pace_sweep.sh)
    uint32_t pace_log2 = direct ? 2u : 0u;
    uint32_t window = 1; // strips a pair may run ahead of its slowest sibling (see "pacing" in the kernel; measured on the bench launch: window 1 17 GB from HBM and 7.6-7.8 ms, 2 / 3 20 GB and 7.9 ms, unpaced 19+ GB)
#ifdef YAMS_ACCEL_MEASURE
    if (version >= 41 && version <= 48) {
        if (version == 41) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 1>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else if (version == 42) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 2>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else if (version == 43) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 3>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else if (version == 44) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 4>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else if (version == 45) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 5>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else if (version == 46) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 6>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else if (version == 48) hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 8>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        else hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER, 7>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        return hipGetLastError();
    }
    if (const char* wv = std::getenv("YAMS_ACCEL_I8R_WINDOW")) window = static_cast<uint32_t>(std::atoi(wv)) & 0xffffu;
    if (const char* pv = std::getenv("YAMS_ACCEL_I8R_PACE_LOG2")) pace_log2 = static_cast<uint32_t>(std::atoi(pv)) & 31u;
    window |= pace_log2 << 16; pace_log2 = 0;
    if (rp.use && version >= 61 && version <= 69) {
        if (version == 61) hipLaunchKernelGGL((scan_tiles_i8r_kernel<1>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else if (version == 62) hipLaunchKernelGGL((scan_tiles_i8r_kernel<2>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else if (version == 64) hipLaunchKernelGGL((scan_tiles_i8r_kernel<4>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else if (version == 68) hipLaunchKernelGGL((scan_tiles_i8r_kernel<8>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else if (version == 66) hipLaunchKernelGGL((scan_tiles_i8r_kernel<9>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else if (version == 69) hipLaunchKernelGGL((scan_tiles_i8r_kernel<10, false, true>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else if (version == 65) hipLaunchKernelGGL((scan_tiles_i8r_kernel<7, false, true>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        else hipLaunchKernelGGL((scan_tiles_i8r_kernel<7>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        return hipGetLastError();
    }
    if (version == 40) {
        hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
        return hipGetLastError();
    }
    if (rp.use && version == 90 && (L.plan.dim == 384 || L.plan.dim == 768)) { // two slabs of row fragments in flight per wave
        hipLaunchKernelGGL((scan_tiles_i8d_kernel<0>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        return hipGetLastError();
    }
    if (rp.use && version == 87) { // the direct form as shipped: thresholds + survivors in LDS, boundary work before the drain (ZSM 2 | 4 | 64)
        hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, false, true, 70>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        return hipGetLastError();
    }
    if (rp.use && (version == 84 || version == 85)) { // ... with parts of the short strip boundary (ZSM bits, see the kernel)
#define YAMS_ZS_LAUNCH(M) hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, false, true, M>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window)
        if (version == 84) YAMS_ZS_LAUNCH(2);               // the plain form with its threshold halves resident in LDS
        else YAMS_ZS_LAUNCH(2 | 4);                         // ... and its survivors in the LDS buffer
#undef YAMS_ZS_LAUNCH
        return hipGetLastError();
    }
    if (rp.use && version == 80) { // the resident-query form with direct row loads
        hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, false, true>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
        return hipGetLastError();
    }
    if (rp.use && L.i8_q_form && version != 71 && version != 72) { // 128 x 128 wave tiles, one wave per SIMD
        hipLaunchKernelGGL((scan_tiles_i8q_kernel<0>), dim3(rp.grid), dim3(Q_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window & 0xffffu);
        return hipGetLastError();
    }
    if (rp.use && L.i8_q_form && version == 72) { // no MFMAs (ablation)
        hipLaunchKernelGGL((scan_tiles_i8q_kernel<2>), dim3(rp.grid), dim3(Q_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window & 0xffffu);
        return hipGetLastError();
    }
    if (rp.use && L.i8_q_form && version == 71) { // the same without row loads after the prologue (ablation)
        hipLaunchKernelGGL((scan_tiles_i8q_kernel<1>), dim3(rp.grid), dim3(Q_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window & 0xffffu);
        return hipGetLastError();
    }
#endif
    window |= pace_log2 << 16;
    // the direct form ships with its threshold halves and its survivors in LDS (ZSM 2 | 4: 7.47 -> 7.28 ms at dim 768,
    // 4.26 -> 4.04 at 384, same candidates; profiles/r04_filter_forms.json) and, since round 6, with the boundary's
    // memory-independent work in front of the drain (| 64: 7.25 -> 7.14, 4.05 -> 3.92; profiles/r06_filter_forms.json)
    // dims 384 and 768 (slab counts that are multiples of three) take the form with two slabs of row fragments in flight per
    // wave (scan_i8d_kernel.h): 12.5M rows, direct form / this form — 256 queries dim 384 1.256 / 1.112 ms, dim 768 2.276 / 2.077;
    // BASELINE config 2 0.140 / 0.133; 1024 queries 3.91 / 3.88 and 7.13 / 7.11 (power-bound: the bytes in flight are not what
    // limits it); one query tile (HBM-bound) unchanged.  Same candidate sets (profiles/r06_filter_forms.json).
    if (rp.use && direct && (L.plan.dim == 384 || L.plan.dim == 768))
        hipLaunchKernelGGL((scan_tiles_i8d_kernel<0>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
    else if (rp.use && direct) hipLaunchKernelGGL((scan_tiles_i8r_kernel<0, false, true, 70>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
    else if (rp.use) hipLaunchKernelGGL((scan_tiles_i8r_kernel<0>), dim3(rp.grid), dim3(R_THREADS), 0, st, a, rp.n_units, rp.n_qt, rp.n_streams, window);
    else hipLaunchKernelGGL((scan_tiles_i8h_kernel<MODE_FILTER>), dim3(hgrid), dim3(H_THREADS), 0, st, a);
    return hipGetLastError();
}

} // namespace yams_accel
