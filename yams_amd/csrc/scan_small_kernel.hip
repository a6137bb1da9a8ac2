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
//   * per query the workgroup sorts its 256 keys (similarity desc | distance asc, then the tie rank) and writes
//     its best kk = min(k, 256);
//   * the workgroup that finishes LAST for a query chunk (one ticket counter per chunk, reset by that workgroup)
//     sorts the n_wg * kk survivors and writes the result: scores, rows, counts, distances, ranks — the vec0
//     rule under L2 (the k nearest first, THEN the cosine threshold, :4506-4510).
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

// descending bitonic sort of (key, idx) pairs s[0, m), m a power of two, by all threads of the workgroup
__device__ __forceinline__ void sort_pairs_desc(uint64_t* skey, uint32_t* sidx, int m) {
    for (int kk = 2; kk <= m; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += SM_THREADS) {
                const int i = 2 * t - (t & (j - 1));
                const int ixj = i + j;
                const uint64_t x = skey[i], y = skey[ixj];
                const bool up = (i & kk) == 0;
                if (up ? (x < y) : (x > y)) {
                    skey[i] = y; skey[ixj] = x;
                    const uint32_t tt = sidx[i]; sidx[i] = sidx[ixj]; sidx[ixj] = tt;
                }
            }
            __syncthreads();
        }
    }
}

template <int METRIC, int QB>
__global__ __launch_bounds__(SM_THREADS) void small_scan_kernel(SmallScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t dim = a.dim;
    const uint32_t dim4 = (dim + 3u) & ~3u;
    unsigned char* sring = smem;                                         // [4 waves][SM_RING_BYTES]; the sort arrays reuse it after the walk
    uint64_t* skey = reinterpret_cast<uint64_t*>(smem);                  // [sort_cap]
    uint32_t* sidx = reinterpret_cast<uint32_t*>(skey + a.sort_cap);     // [sort_cap]
    float* saux = reinterpret_cast<float*>(sidx + a.sort_cap);           // [256]: cosine of this workgroup's rows (L2)
    float* sq = reinterpret_cast<float*>(smem + 4 * SM_RING_BYTES);      // [QB][dim4]
    __shared__ double s_qn[QB];
    __shared__ uint32_t s_last;
    __shared__ uint32_t s_outn;

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
    const float* xrow = a.rows + static_cast<uint64_t>(row) * dim;
    const uint32_t ring = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
                              (__attribute__((address_space(3))) unsigned char*)sring)) + wave * SM_RING_BYTES; // LDS byte address
    const uint32_t n_chunks = dim / 32;
    auto issue_pass = [&](uint32_t c0, uint32_t nc) {
        for (uint32_t c = 0; c < nc; ++c) {
#pragma unroll
            for (int pc = 0; pc < 8; ++pc)
                lds_dma16(xrow + (c0 + c) * 32 + pc * 4, __builtin_amdgcn_readfirstlane(ring + (c * 8 + pc) * 1024));
        }
    };
    if (wave_live) issue_pass(0, min(static_cast<uint32_t>(SM_PASS_CHUNKS), n_chunks));
    // ---- query validity + norm: lanes 0..QB-1 of the last wave, while its first pass is in flight ------------
    if (wave == 3 && lane < QB) {
        double acc = 0.0;
        const float* s = sq + lane * dim4;
        for (uint32_t i = 0; i < dim; ++i) { const double d = static_cast<double>(s[i]); acc = fma(d, d, acc); }
        uint32_t f = 0;
        if (!isfinite(acc)) f |= 1u;             // a non-finite element (fp64 cannot overflow on fp32 squares)
        if (!(acc >= 1e-10)) f |= 2u;            // isZeroNormEmbedding, :204-211
        s_qn[lane] = sqrt(acc);
        if (blockIdx.x == 0 && static_cast<uint32_t>(lane) < nqc) a.qflags[q0 + lane] = f;
    }
    if (wave_live) {
        const unsigned char* mine = sring + wave * SM_RING_BYTES + lane * 16;
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += SM_PASS_CHUNKS) {
            const uint32_t nc = min(static_cast<uint32_t>(SM_PASS_CHUNKS), n_chunks - c0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this pass has landed
            for (uint32_t c = 0; c < nc; ++c) {
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) {
                    const float4 v = *reinterpret_cast<const float4*>(mine + (c * 8 + pc) * 1024);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double sv = static_cast<double>(vv[e]);
                        nsq = fma(sv, sv, nsq);                           // :4253-4266, sequential i, fp64
#pragma unroll
                        for (int j = 0; j < QB; ++j) {
                            const double qv = static_cast<double>(sq[j * dim4 + (c0 + c) * 32 + pc * 4 + e]);
                            dot[j] = fma(sv, qv, dot[j]);
                            if (METRIC == YAMS_SCAN_L2) { const double d = sv - qv; dsq[j] = fma(d, d, dsq[j]); }
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
    __syncthreads(); // the norms are there, and nobody reads the rings any more (the sort arrays live there)
    const uint32_t rank = live ? (a.tie_rank ? a.tie_rank[row] : row) : 0u;
    const uint32_t kk = a.kk;

    // ---- per query: this workgroup's best kk ----------------------------------------------------------------
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
        skey[threadIdx.x] = key; sidx[threadIdx.x] = threadIdx.x; saux[threadIdx.x] = aux;
        __syncthreads();
        sort_pairs_desc(skey, sidx, SM_THREADS);
        const uint64_t base = (static_cast<uint64_t>(q0 + j) * gridDim.x + blockIdx.x) * kk;
        for (uint32_t i = threadIdx.x; i < kk; i += SM_THREADS) {
            a.part_key[base + i] = skey[i];
            if (METRIC == YAMS_SCAN_L2) a.part_aux[base + i] = saux[sidx[i]];
        }
        __syncthreads();
    }

    // ---- the last workgroup of this query chunk finishes the job ---------------------------------------------
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(&a.counter[blockIdx.y], 1u);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) a.counter[blockIdx.y] = 0; // ready for the next call
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const uint32_t total = gridDim.x * kk;
    int m = 64;
    while (m < static_cast<int>(total)) m <<= 1;
    for (uint32_t j = 0; j < nqc; ++j) {
        const uint32_t q = q0 + j;
        const uint64_t base = static_cast<uint64_t>(q) * total;
        for (int i = threadIdx.x; i < m; i += SM_THREADS) {
            skey[i] = static_cast<uint32_t>(i) < total ? __builtin_nontemporal_load(a.part_key + base + i) : 0ull;
            sidx[i] = i;
        }
        __syncthreads();
        sort_pairs_desc(skey, sidx, m);
        uint32_t nv;
        { uint32_t lo = 0, hi = static_cast<uint32_t>(m);   // valid (non-zero) keys: binary search on the sorted array
          while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skey[mid] != 0) lo = mid + 1; else hi = mid; }
          nv = lo; }
        const uint32_t take = nv < a.k ? nv : a.k;
        if (METRIC == YAMS_SCAN_COSINE) {
            for (uint32_t i = threadIdx.x; i < a.k; i += SM_THREADS) {
                const uint64_t o = static_cast<uint64_t>(q) * a.k + i;
                if (i < take) {
                    const uint32_t rk = key_idx(skey[i]);
                    const uint32_t r = a.tie_rank ? a.rank_row[rk] : rk;
                    const float sim = key_score(skey[i]);
                    a.out_scores[o] = sim;
                    a.out_rows[o] = small_global_row(a, r);
                    if (a.out_ranks) a.out_ranks[o] = rk;
                    if (a.out_dist) a.out_dist[o] = 1.0f - sim;
                } else {
                    a.out_scores[o] = -__builtin_inff();
                    a.out_rows[o] = -1;
                    if (a.out_ranks) a.out_ranks[o] = 0xffffffffu;
                    if (a.out_dist) a.out_dist[o] = __builtin_inff();
                }
            }
            if (threadIdx.x == 0) a.out_counts[q] = take;
        } else {
            // vec0 semantics: the k nearest, THEN the cosine threshold (:4506-4510), order preserved
            if (threadIdx.x == 0) {
                uint32_t outn = 0;
                const bool defer = (a.flags & YAMS_SCAN_FLAG_DEFER_THRESHOLD) != 0;
                for (uint32_t i = 0; i < take; ++i) {
                    const float cs = __builtin_nontemporal_load(a.part_aux + base + sidx[i]);
                    if (!defer && cs < a.threshold) continue;
                    const uint64_t o = static_cast<uint64_t>(q) * a.k + outn;
                    const uint32_t rk = key_idx(skey[i]);
                    const uint32_t r = a.tie_rank ? a.rank_row[rk] : rk;
                    a.out_scores[o] = cs;
                    a.out_rows[o] = small_global_row(a, r);
                    if (a.out_dist) a.out_dist[o] = -key_score(skey[i]);
                    if (a.out_ranks) a.out_ranks[o] = rk;
                    ++outn;
                }
                s_outn = outn;
                a.out_counts[q] = outn;
            }
            __syncthreads();
            for (uint32_t i = s_outn + threadIdx.x; i < a.k; i += SM_THREADS) {
                const uint64_t o = static_cast<uint64_t>(q) * a.k + i;
                a.out_scores[o] = -__builtin_inff();
                a.out_rows[o] = -1;
                if (a.out_dist) a.out_dist[o] = __builtin_inff();
                if (a.out_ranks) a.out_ranks[o] = 0xffffffffu;
            }
        }
        __syncthreads();
    }
}

} // namespace

uint32_t small_scan_sort_cap(uint32_t n_wg, uint32_t kk) {
    uint32_t m = SM_THREADS;
    while (m < n_wg * kk) m <<= 1;
    return m;
}

size_t small_scan_lds_bytes(uint32_t dim, uint32_t qb, uint32_t sort_cap) {
    const size_t dim4 = (dim + 3u) & ~3u;
    // rings (the sort arrays alias them: sort_cap * 12 + 1 KiB <= 128 KiB) + the queries
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
