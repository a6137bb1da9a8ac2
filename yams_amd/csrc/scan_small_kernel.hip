// scan_small_kernel.hip — the whole exact search of a SMALL corpus in ONE launch.
//
// The reference's most common call is one query (search_vector_pipeline.cpp:221 ->
// sqlite_vec_backend.cpp:1436-1454 -> bruteForceSearchUnlocked :4204-4331), often against a few thousand rows
// (BASELINE config 1: 10k x 384, k = 10).  The filter pipeline of the large-corpus path is a dozen dependent
// launches there — launch latency, not work.  This kernel does the reference's arithmetic for EVERY row instead,
// fp64 in the reference's summation order (no filter, nothing to prove), and reduces to the final top-k itself:
//   * a workgroup of four waves owns 256 consecutive rows and a chunk of QB queries; a lane walks ITS row once and
//     accumulates norm, dot (and squared distance under L2) for the QB queries.  There are too few rows to hide
//     memory latency behind other waves (10k rows = 157 waves for 1024 SIMDs), so a wave brings its rows in with
//     LDS-DMA, 128 dims of all 64 rows per pass (32 x global_load_lds_dwordx4 back to back, no registers in
//     between: lane L fetches 16 bytes of ITS row per instruction, the LDS image of an instruction is lane-linear,
//     so the reads back are conflict-free), pays the latency once per pass and then sums out of LDS;
//   * query validity + ||q|| (fp64, sequential, :4206-4211) are computed by every workgroup for its own queries
//     (a 384-element chain beside the row walk of the other waves) — no separate prep launch;
//   * per query the workgroup keeps its best kk = min(k, 256) of its 256 keys (similarity desc | distance asc, then
//     the tie rank) by RANK COUNTING: a key's place is the number of keys greater than it, counted by its thread
//     against the list in LDS — no sorting network, no barrier per step;
//   * the workgroup that finishes LAST for a query chunk (one ticket counter per chunk, reset by that workgroup)
//     selects from the n_wg * kk <= 1024 survivors the same way and writes the result: scores, rows, counts, distances, ranks — the vec0 rule under L2 (the k nearest first, THEN
//     the cosine threshold, :4506-4510).
// Host side: one launch, one look at the query flags (scan_api.cpp, small_scan).
#include "common.h"
#include "lds_dma.h"
#include "scan_launch.h"

namespace yams_accel {
namespace {

constexpr int SM_THREADS = 256;
constexpr int SM_PASS_CHUNKS = 4;             // chunks of 32 dims (128 B of each of the wave's 64 rows = 8 KiB) staged per pass
constexpr int SM_RING_BYTES = SM_PASS_CHUNKS * 8 * 1024; // per wave

__device__ __forceinline__ int64_t small_global_row(const SmallScanArgs& a, uint32_t row) {
    if (a.stripe_rows == 0) return a.row_base + static_cast<int64_t>(row);
    const uint64_t t = row / a.stripe_rows, w = row % a.stripe_rows;
    return a.row_base + static_cast<int64_t>((t * a.n_stripes + a.stripe_index) * a.stripe_rows + w);
}

// Selection by RANK COUNTING instead of sorting.  Keys are unique (they carry the row's tie rank) or zero (empty
// slot), so the place of a key in the descending order is the number of keys greater than it: every thread counts
// that for its own key(s) against the list in LDS — broadcast reads, 16 bytes = two keys at a time, no bank
// conflicts, no barrier inside — and the keys whose rank is below the cut write themselves straight to their place.
// n^2 / 256 comparisons per thread: 256 for a workgroup's own 256 keys, <= 4096 for the <= 1024 survivors of the
// final selection.  (A bitonic network over the same keys — in LDS with a barrier per step, or in registers with a
// shuffle per step — took 10-25 us per list at the clocks such a small launch runs at: more than the scoring pass.)
__device__ __forceinline__ uint32_t rank_of(const uint64_t* list, uint32_t n_pow2 /* multiple of 16 */, uint64_t mine) {
    uint32_t r = 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2* l2 = reinterpret_cast<const u64x2*>(list);
#pragma unroll 8
    for (uint32_t i = 0; i < n_pow2 / 2; ++i) { // (n a multiple of 16 keys, zero-padded: zeros are never greater)
        const u64x2 v = l2[i];
        r += (v.x > mine ? 1u : 0u) + (v.y > mine ? 1u : 0u);
    }
    return r;
}

// The final selection of ONE query by a group of NT threads (the whole workgroup, or one wave when the chunk holds
// several queries: then the selections of a chunk run side by side): every workgroup's list is sorted, so list b
// alone holds kk keys >= its last one and the k best overall are all >= tau = the LARGEST of the lists' last keys;
// only those candidates are compacted and ranked — a few dozen for ordinary data instead of all n_wg * kk (any
// number stays correct).  `region`: SM_FINAL_REGION bytes of LDS of the group's own.
constexpr int SM_FINAL_REGION = 23040;
template <int NT> __device__ __forceinline__ void group_sync() {
    if (NT == SM_THREADS) __syncthreads();
    else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } // (a wave's LDS operations complete in order)
}
template <int METRIC, int NT>
__device__ __forceinline__ void final_select(const SmallScanArgs& a, uint32_t q, uint32_t total, uint32_t kk,
                                             unsigned char* region, int t) {
    uint64_t* fkey = reinterpret_cast<uint64_t*>(region);                    // [1024] the survivors
    uint64_t* fcand = fkey + 1024;                                          // [1024] the candidates (>= tau)
    uint32_t* fsrc = reinterpret_cast<uint32_t*>(fcand + 1024);             // [1024] their index among the survivors
    uint32_t* frank_row = fsrc + 1024;                                      // [256]: (L2) survivor index of rank r
    float* fcs = reinterpret_cast<float*>(frank_row + 256);                 // [256]: (L2) its cosine
    unsigned long long* s_tau = reinterpret_cast<unsigned long long*>(fcs + 256);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_tau + 1);               // [0] candidates, [1] valid survivors, [2] passing (L2)
    const uint64_t base = static_cast<uint64_t>(q) * total;
    uint32_t total2 = 64;
    while (total2 < total) total2 <<= 1;
    for (uint32_t i = t; i < total2; i += NT) fkey[i] = i < total ? __builtin_nontemporal_load(a.part_key + base + i) : 0ull;
    for (uint32_t i = t; i < 1024; i += NT) fcand[i] = 0;
    if (t == 0) { *s_tau = 0; s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; }
    group_sync<NT>();
    const uint32_t n_lists = total / kk;
    if (a.k <= kk)
        for (uint32_t b2 = t; b2 < n_lists; b2 += NT) atomicMax(s_tau, static_cast<unsigned long long>(fkey[b2 * kk + kk - 1]));
    group_sync<NT>();
    const uint64_t tau = *s_tau;                                            // (0 when no list is full: nothing is pruned)
    uint32_t nv_mine = 0;
    for (uint32_t i = t; i < total; i += NT) {
        const uint64_t mine = fkey[i];
        if (!mine) continue;
        ++nv_mine;
        if (mine < tau) continue;
        const uint32_t slot = atomicAdd(&s_cnt[0], 1u);
        fcand[slot] = mine; fsrc[slot] = i;
    }
    if (nv_mine) atomicAdd(&s_cnt[1], nv_mine);
    group_sync<NT>();
    const uint32_t n_cand = s_cnt[0];
    const uint32_t take = s_cnt[1] < a.k ? s_cnt[1] : a.k;
    for (uint32_t c = t; c < n_cand; c += NT) {
        const uint64_t mine = fcand[c];
        const uint32_t r = rank_of(fcand, (n_cand + 15u) & ~15u, mine);
        if (r >= a.k) continue;
        if (METRIC == YAMS_SCAN_COSINE) {
            const uint64_t o = static_cast<uint64_t>(q) * a.k + r;
            const uint32_t rk = key_idx(mine);
            const uint32_t rowi = a.tie_rank ? a.rank_row[rk] : rk;
            const float sim = key_score(mine);
            a.out_scores[o] = sim;
            a.out_rows[o] = small_global_row(a, rowi);
            if (a.out_ranks) a.out_ranks[o] = rk;
            if (a.out_dist) a.out_dist[o] = 1.0f - sim;
        } else {
            frank_row[r] = c;                                               // (k <= 256)
            fcs[r] = __builtin_nontemporal_load(a.part_aux + base + fsrc[c]);
        }
    }
    if (METRIC == YAMS_SCAN_COSINE) {
        for (uint32_t r = take + t; r < a.k; r += NT) {
            const uint64_t o = static_cast<uint64_t>(q) * a.k + r;
            a.out_scores[o] = -__builtin_inff(); a.out_rows[o] = -1;
            if (a.out_ranks) a.out_ranks[o] = 0xffffffffu;
            if (a.out_dist) a.out_dist[o] = __builtin_inff();
        }
        if (t == 0) a.out_counts[q] = take;
        return;
    }
    // vec0 semantics: the k nearest, THEN the cosine threshold (:4506-4510), order preserved: the place of rank r is
    // the number of passing ranks before it
    group_sync<NT>();
    const bool defer = (a.flags & YAMS_SCAN_FLAG_DEFER_THRESHOLD) != 0;
    for (uint32_t r = t; r < take; r += NT) {
        if (!(defer || !(fcs[r] < a.threshold))) continue;
        uint32_t pos = 0;
        for (uint32_t i = 0; i < r; ++i) pos += (defer || !(fcs[i] < a.threshold)) ? 1u : 0u;
        const uint64_t mine = fcand[frank_row[r]];
        const uint64_t o = static_cast<uint64_t>(q) * a.k + pos;
        const uint32_t rk = key_idx(mine);
        const uint32_t rowi = a.tie_rank ? a.rank_row[rk] : rk;
        a.out_scores[o] = fcs[r];
        a.out_rows[o] = small_global_row(a, rowi);
        if (a.out_dist) a.out_dist[o] = -key_score(mine);
        if (a.out_ranks) a.out_ranks[o] = rk;
        atomicAdd(&s_cnt[2], 1u);
    }
    group_sync<NT>();
    const uint32_t outn = s_cnt[2];
    for (uint32_t rr = outn + t; rr < a.k; rr += NT) {
        const uint64_t o = static_cast<uint64_t>(q) * a.k + rr;
        a.out_scores[o] = -__builtin_inff(); a.out_rows[o] = -1;
        if (a.out_dist) a.out_dist[o] = __builtin_inff();
        if (a.out_ranks) a.out_ranks[o] = 0xffffffffu;
    }
    if (t == 0) a.out_counts[q] = outn;
}

#ifdef YAMS_ACCEL_MEASURE
#define SM_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SM_STAMP(i) ((void)0)
#endif

template <int METRIC, int QB>
__global__ __launch_bounds__(SM_THREADS) void small_scan_kernel(SmallScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t dim = a.dim;
    const uint32_t dim4 = (dim + 3u) & ~3u;
    unsigned char* sring = smem;                                         // [4 waves][SM_RING_BYTES]; the key staging reuses it after the walk
    uint64_t* skey = reinterpret_cast<uint64_t*>(smem);                  // [QB][256]: this workgroup's keys per query
    float* saux = reinterpret_cast<float*>(skey + QB * SM_THREADS);      // [QB][256]: cosine of its rows (L2)
    float* sq = reinterpret_cast<float*>(smem + 4 * SM_RING_BYTES);      // [QB][dim4]
    __shared__ double s_qn[QB];
    __shared__ uint32_t s_last;

    SM_STAMP(0);
    const uint32_t q0 = blockIdx.y * QB;
    const uint32_t nqc = min(static_cast<uint32_t>(QB), a.nq - q0);      // queries of this chunk
    for (uint32_t i = threadIdx.x; i < QB * dim4; i += SM_THREADS) {
        const uint32_t j = i / dim4, e = i % dim4;
        sq[i] = (j < nqc && e < dim) ? a.queries[static_cast<uint64_t>(q0 + j) * dim + e] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- the row walk -----------------------------------------------------------------------------------------
    const uint32_t row_raw = blockIdx.x * SM_THREADS + threadIdx.x;
    bool live = row_raw < a.n_rows;
    const uint32_t row = live ? row_raw : a.n_rows - 1;                   // (a valid address for the staged loads)
    if (live && a.row_mask && !((a.row_mask[row >> 5] >> (row & 31)) & 1u)) live = false;
    double nsq = 0.0, dot[QB], dsq[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) { dot[j] = 0.0; dsq[j] = 0.0; }
    const uint32_t wrow0 = blockIdx.x * SM_THREADS + wave * 64;          // the wave's 64 consecutive rows
    const bool wave_live = wrow0 < a.n_rows;
    const uint32_t ring = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
                              (__attribute__((address_space(3))) unsigned char*)sring)) + wave * SM_RING_BYTES; // LDS byte address
    const uint32_t n_chunks = dim / 32;
    // One DMA instruction fetches a 128-byte line of EIGHT rows (lane = row-in-8 * 8 + slot): eight memory requests,
    // not the sixty-four of a lane-per-row fetch (the walk was bound by that: ~1 request per cycle per CU).  The LDS
    // image of an instruction is lane-linear, so the eight 16-byte pieces of a row land next to each other, 128 bytes
    // from the next row — the 64 lanes reading "piece p of my row" would hit four bank groups.  The rotation is
    // therefore applied at the SOURCE: slot s of row R receives piece (s - rho(R)) mod 8, rho(R) = ((R >> 1) & 3) +
    // 4 * ((R >> 3) & 1), so that lane R finds piece p in slot (p + rho(R)) mod 8 and sixteen consecutive lanes
    // touch sixteen different bank groups; the registers a lane sums from keep their natural order.
    auto rho = [](uint32_t r64) { return ((r64 >> 1) & 3u) + 4u * ((r64 >> 3) & 1u); };
    const uint32_t dma_row8 = static_cast<uint32_t>(lane) >> 3, dma_slot = static_cast<uint32_t>(lane) & 7u;
    auto issue_pass = [&](uint32_t c0, uint32_t nc) {
        for (uint32_t c = 0; c < nc; ++c) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const uint32_t r64 = 8u * g + dma_row8;                    // row of the wave this lane fetches for
                uint32_t rr = wrow0 + r64;
                if (rr >= a.n_rows) rr = a.n_rows - 1;                    // (a valid address; the lane's key is dropped later)
                const uint32_t piece = (dma_slot - rho(r64)) & 7u;
                lds_dma16(a.rows + static_cast<uint64_t>(rr) * dim + (c0 + c) * 32 + piece * 4,
                          __builtin_amdgcn_readfirstlane(ring + (c * 8 + g) * 1024));
            }
        }
    };
    SM_STAMP(1);
    if (wave_live) issue_pass(0, min(static_cast<uint32_t>(SM_PASS_CHUNKS), n_chunks));
    SM_STAMP(2);
    // ---- query validity + norm: lanes 0..QB-1 of the last wave, while its first pass is in flight ------------
    if (wave == 3 && lane < QB) {
        double acc = 0.0;
        const float4* s4 = reinterpret_cast<const float4*>(sq + lane * dim4);
        for (uint32_t i = 0; i < dim / 32; ++i) { // 32 elements per step: the eight reads go out together
            float4 v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = s4[i * 8 + t];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float vv[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double d = static_cast<double>(vv[e]); acc = fma(d, d, acc); }
            }
        }
        uint32_t f = 0;
        if (!isfinite(acc)) f |= 1u;             // a non-finite element (fp64 cannot overflow on fp32 squares)
        if (!(acc >= 1e-10)) f |= 2u;            // isZeroNormEmbedding, :204-211
        s_qn[lane] = sqrt(acc);
        if (blockIdx.x == 0 && static_cast<uint32_t>(lane) < nqc) a.qflags[q0 + lane] = f;
    }
    if (wave_live) {
        // lane L = row L of the wave: block (L >> 3) of a chunk, row-in-8 (L & 7), piece p in slot (p + rho(L)) & 7
        const unsigned char* mine = sring + wave * SM_RING_BYTES + (lane >> 3) * 1024 + (lane & 7) * 128;
        const uint32_t my_rho = rho(static_cast<uint32_t>(lane));
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += SM_PASS_CHUNKS) {
            const uint32_t nc = min(static_cast<uint32_t>(SM_PASS_CHUNKS), n_chunks - c0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this pass has landed
            for (uint32_t c = 0; c < nc; ++c) {
                // all LDS reads of a chunk go out before the first sum (one latency per chunk, not one per 16 bytes)
                float4 rv[8], qv[QB][8];
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) rv[pc] = *reinterpret_cast<const float4*>(mine + c * 8192 + ((pc + my_rho) & 7u) * 16);
#pragma unroll
                for (int j = 0; j < QB; ++j)
#pragma unroll
                    for (int pc = 0; pc < 8; ++pc)
                        qv[j][pc] = *reinterpret_cast<const float4*>(sq + j * dim4 + (c0 + c) * 32 + pc * 4);
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) {
                    const float vv[4] = {rv[pc].x, rv[pc].y, rv[pc].z, rv[pc].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double sv = static_cast<double>(vv[e]);
                        nsq = fma(sv, sv, nsq);                           // :4253-4266, sequential i, fp64
#pragma unroll
                        for (int j = 0; j < QB; ++j) {
                            const float qq[4] = {qv[j][pc].x, qv[j][pc].y, qv[j][pc].z, qv[j][pc].w};
                            const double qd = static_cast<double>(qq[e]);
                            dot[j] = fma(sv, qd, dot[j]);
                            if (METRIC == YAMS_SCAN_L2) { const double d = sv - qd; dsq[j] = fma(d, d, dsq[j]); }
                        }
                    }
                }
            }
            if (c0 + SM_PASS_CHUNKS < n_chunks) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the reads of this pass are done: its slots are free
                issue_pass(c0 + SM_PASS_CHUNKS, min(static_cast<uint32_t>(SM_PASS_CHUNKS), n_chunks - c0 - SM_PASS_CHUNKS));
            }
        }
    }
    SM_STAMP(3);
    __syncthreads(); // the norms are there, and nobody reads the rings any more (the sort arrays live there)
    const uint32_t rank = live ? (a.tie_rank ? a.tie_rank[row] : row) : 0u;
    const uint32_t kk = a.kk;

    // ---- per query: this workgroup's best kk.  Every thread parks its keys; a key's rank is counted by its thread ----
    for (uint32_t j = 0; j < nqc; ++j) {
        double dj = 0.0, qj = 0.0;
#pragma unroll
        for (int t = 0; t < QB; ++t) if (static_cast<uint32_t>(t) == j) { dj = dot[t]; qj = dsq[t]; }
        const double qn = s_qn[j];
        uint64_t key = 0;
        float aux = 0.f;
        if (live) {
            if (METRIC == YAMS_SCAN_COSINE) {
                // :4258-4269 (the record path drops norm^2 < 1e-10 instead, isZeroNormEmbedding :204-211)
                if (isfinite(nsq) && ((a.flags & YAMS_SCAN_FLAG_RECORD_PATH) ? nsq >= 1e-10 : nsq > 1e-12)) {
                    const double denom = sqrt(nsq) * qn;                 // :4271
                    const double sd = denom > 0.0 ? dj / denom : 0.0;
                    if (isfinite(sd)) {                                  // :4273-4275
                        const float sim = static_cast<float>(sd);        // :4276
                        if (!(sim < a.threshold)) key = pack_key(sim, rank); // :4277-4279
                    }
                }
            } else if (isfinite(nsq)) { // non-finite rows cannot be stored (vector_database.cpp:1771-1784)
                const double dd = sqrt(qj);
                if (isfinite(dd)) {
                    key = pack_key(-static_cast<float>(dd), rank);       // ascending distance == descending -dist
                    // computeCosineSimilarity (vector_database.cpp:1786-1810): sqrt each norm, 0 on zero norm
                    const double nb = sqrt(nsq);
                    aux = static_cast<float>((qn == 0.0 || nb == 0.0) ? 0.0 : dj / (qn * nb));
                }
            }
        }
        skey[j * SM_THREADS + threadIdx.x] = key;
        if (METRIC == YAMS_SCAN_L2) saux[j * SM_THREADS + threadIdx.x] = aux;
    }
    __syncthreads();
    // Two levels: a key that is not among the best kk of its own WAVE's 64 cannot be among the workgroup's best kk,
    // so each thread first counts within its wave (64 comparisons); the survivors (<= 4 kk) are compacted and
    // ranked among themselves.  (kk >= 64 keeps everything: one level, 256 comparisons.)
    uint64_t* scomp = skey + QB * SM_THREADS + (QB * SM_THREADS) / 2;    // [QB][256] compacted survivors (behind saux)
    __shared__ uint32_t s_ncomp[QB];
    if (threadIdx.x < QB) s_ncomp[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < QB * SM_THREADS; i += SM_THREADS) scomp[i] = 0;
    __syncthreads();
    uint64_t mine_k[QB];
    bool surv[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        mine_k[j] = 0; surv[j] = false;
        if (static_cast<uint32_t>(j) < nqc) {
            const uint64_t mine = skey[j * SM_THREADS + threadIdx.x];
            mine_k[j] = mine;
            if (mine) {
                const uint32_t rl = kk >= 64 ? 0u : rank_of(skey + j * SM_THREADS + wave * 64, 64, mine);
                if (rl < kk) { surv[j] = true; scomp[j * SM_THREADS + atomicAdd(&s_ncomp[j], 1u)] = mine; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        if (static_cast<uint32_t>(j) >= nqc) break;
        const uint32_t nc = s_ncomp[j];                                   // valid keys kept (all of them when kk >= 64)
        const uint64_t base = (static_cast<uint64_t>(q0 + j) * gridDim.x + blockIdx.x) * kk;
        if (surv[j]) {
            const uint32_t r = rank_of(scomp + j * SM_THREADS, (nc + 15u) & ~15u, mine_k[j]);
            if (r < kk) {
                a.part_key[base + r] = mine_k[j];
                if (METRIC == YAMS_SCAN_L2) a.part_aux[base + r] = saux[j * SM_THREADS + threadIdx.x];
            }
        }
        if (threadIdx.x >= nc && threadIdx.x < kk) a.part_key[base + threadIdx.x] = 0;   // fewer valid keys than kk: empty slots
    }

    SM_STAMP(4);
    // ---- the last workgroup of this query chunk finishes the job ----------------------------------------------
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(&a.counter[blockIdx.y], 1u);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) a.counter[blockIdx.y] = 0; // ready for the next call
    }
    __syncthreads();
    SM_STAMP(5);
    if (!s_last) return;
    __threadfence();
    SM_STAMP(6);
    const uint32_t total = gridDim.x * kk;                                // <= small_scan_max_survivors()
    // one query: the whole workgroup selects; several: wave j selects query j, the selections run side by side
    if (nqc == 1) final_select<METRIC, SM_THREADS>(a, q0, total, kk, smem, threadIdx.x);
    else if (static_cast<uint32_t>(wave) < nqc) final_select<METRIC, 64>(a, q0 + wave, total, kk, smem + wave * SM_FINAL_REGION, lane);
    SM_STAMP(7);
}

} // namespace

uint32_t small_scan_max_survivors() { return 1024; } // n_wg * kk the last workgroup ranks in LDS

size_t small_scan_lds_bytes(uint32_t dim, uint32_t qb, uint32_t sort_cap) {
    const size_t dim4 = (dim + 3u) & ~3u;
    // rings (the key staging of the selection aliases them) + the queries
    (void)sort_cap;
    return static_cast<size_t>(4) * SM_RING_BYTES + qb * dim4 * 4;
}

hipError_t launch_small_scan(hipStream_t st, int metric, const SmallScanArgs& a, uint32_t qb) {
    const uint32_t n_wg = (a.n_rows + SM_THREADS - 1) / SM_THREADS;
    const dim3 grid(n_wg, (a.nq + qb - 1) / qb);
    const size_t lds = small_scan_lds_bytes(a.dim, qb, a.sort_cap);
    if (metric == YAMS_SCAN_COSINE) {
        if (qb == 1) small_scan_kernel<YAMS_SCAN_COSINE, 1><<<grid, SM_THREADS, lds, st>>>(a);
        else small_scan_kernel<YAMS_SCAN_COSINE, 4><<<grid, SM_THREADS, lds, st>>>(a);
    } else {
        if (qb == 1) small_scan_kernel<YAMS_SCAN_L2, 1><<<grid, SM_THREADS, lds, st>>>(a);
        else small_scan_kernel<YAMS_SCAN_L2, 4><<<grid, SM_THREADS, lds, st>>>(a);
    }
    return hipGetLastError();
}

} // namespace yams_accel
