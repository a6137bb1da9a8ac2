// pq_kernels.hip — the ADC scan of the product-quantised engine.
//
// Reference: SqliteVecBackend::Impl::simeonPqSearchUnlocked, src/vector/sqlite_vec_backend.cpp:3868-4056.  Per query the
// host builds a look-up table lut[j][c] = <query sub-vector j, centroid c of sub-quantiser j> (simeon::PQInnerProductQuery,
// :3901; m sub-quantisers x 256 centroids) and scores EVERY indexed row (or every candidate index) with
//     approxScore = sum_j lut[j][codes[index * m + j]]                                              (:3965-3977)
// keeps the best approxK = min(candidates, max(k, k * rerank_factor)) by (score desc, tie key asc) (:3952-3997), and
// re-scores those exactly (computeCosineSimilarity, :4023-4034).  This file is the first two steps: one byte per
// sub-quantiser per row from HBM (n * m bytes per batch: the roofline), 256-entry table rows resident in LDS.
//
// The ORDER of the fp32 additions is simeon's (third_party/simeon is absent from the checkout): PARITY UNPINNED.  Served:
// one sequential sum over j (LANES = 1) and 4 / 8 / 16 partial sums (element j -> lane j % LANES, lanes added left to
// right) — the shapes a scalar loop and its SSE / AVX / AVX-512 forms take; the host picks the one its build reproduces
// (the same idea as the L2 calibration: a crafted LUT separates them).
#include "common.h"
#include "scan_launch.h"

namespace yams_accel {

// One workgroup: QG queries (their tables in LDS) x a run of indices.  Thread t scores indices t, t + 256, ... of the run
// against all QG tables: a code byte is read once per QG queries.
template <int LANES>
__global__ __launch_bounds__(256) void pq_adc_keys_kernel(const uint8_t* codes, uint64_t n_codes, uint32_t m, const float* luts,
                                                          const uint32_t* qmap, uint32_t n_slots, uint32_t qg,
                                                          const uint32_t* tie_rank, const uint32_t* candidates, uint64_t n_items,
                                                          uint32_t run, uint64_t* keys, uint64_t key_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem);                 // [qg][m][256]
    const uint32_t slot0 = blockIdx.y * qg;
    const uint32_t nq_here = slot0 + qg <= n_slots ? qg : n_slots - slot0;
    const uint32_t lut_floats = m * 256u;
    for (uint32_t s = 0; s < nq_here; ++s) {
        const uint32_t q = qmap ? qmap[slot0 + s] : slot0 + s;
        const float4* src = reinterpret_cast<const float4*>(luts + static_cast<uint64_t>(q) * lut_floats);
        float4* dst = reinterpret_cast<float4*>(lut + static_cast<uint64_t>(s) * lut_floats);
        for (uint32_t i = threadIdx.x; i < lut_floats / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * run;
    const uint64_t i1 = i0 + run < n_items ? i0 + run : n_items;
    for (uint64_t it = i0 + threadIdx.x; it < i1; it += 256) {
        const uint64_t idx = candidates ? candidates[it] : it;
        const bool live = idx < n_codes;
        const uint8_t* code = codes + (live ? idx : 0) * m;
        float part[4][LANES];                                    // [query of the group][lane]
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int l = 0; l < LANES; ++l) part[s][l] = 0.f;
        // sixteen sub-quantisers at a time: the partial-sum lane of element j is j % LANES — a compile-time constant inside the
        // unrolled group (LANES divides 16), so the partial sums stay in registers
        const bool words = (m & 3u) == 0;
        for (uint32_t j0 = 0; j0 < m; j0 += 16) {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (words) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (j0 + 4u * t < m) w[t] = *reinterpret_cast<const uint32_t*>(code + j0 + 4u * t);
            } else {
#pragma unroll
                for (int jj = 0; jj < 16; ++jj)
                    if (j0 + jj < m) w[jj >> 2] |= static_cast<uint32_t>(code[j0 + jj]) << (8 * (jj & 3));
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                if (j0 + jj >= m) break;
                const uint32_t c = (w[jj >> 2] >> (8 * (jj & 3))) & 255u;
                const uint32_t o = (j0 + jj) * 256u + c;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (static_cast<uint32_t>(s) < nq_here) {
                        float& p = part[s][jj % LANES];
                        p = __fadd_rn(p, lut[static_cast<uint64_t>(s) * lut_floats + o]);
                    }
            }
        }
        const uint32_t kidx = tie_rank ? tie_rank[live ? idx : 0] : static_cast<uint32_t>(idx);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (static_cast<uint32_t>(s) < nq_here) {
                float sc = part[s][0];
                if (LANES > 1) {
                    sc = 0.f;
#pragma unroll
                    for (int l = 0; l < LANES; ++l) sc = __fadd_rn(sc, part[s][l]);
                }
                // (a score that is not a number sorts nowhere in the reference's comparator: such a row is left out)
                keys[static_cast<uint64_t>(slot0 + s) * key_stride + it] = (live && sc == sc) ? pack_key(sc, kidx) : 0ull;
            }
    }
}

// ---- the filtered form (round 6) -------------------------------------------------------------------------------------------
// One 64-bit key per (query, code) is 2 GB written and read again for 256 queries over 1M codes, and the selection of the best
// approxK out of a million keys per query cost twice the scan (bench leg "pq": 6.8 ms of scan + 4.7 ms of selection per batch,
// 46x the EXACT search of the same rows).  Only about approxK keys per query matter.  As in the exact scan: the scores of every
// stride-th code (MODE 1) give a per-query threshold tau (the rank-th best sample score: about rank * stride codes reach it),
// the scan proper (MODE 2) appends a key only where score >= tau.  The scores are the reference's fp32 sums themselves, not
// bounds: whenever a list holds approxK keys its best approxK ARE the best approxK of all codes (every code left out scored
// below every code listed); a list that came out too short or too long sends its query through the unfiltered form above.
// The tables of a query group are INTERLEAVED in LDS ([m][256][QG]): one ds_read_b128 per code byte serves four queries; a
// workgroup is 1024 threads (the LDS holds one group's tables, so latency is hidden by waves, not by workgroups).
template <int LANES, int QG, int MODE>
__global__ __launch_bounds__(1024) void pq_adc_filter_kernel(const uint8_t* codes, uint64_t n_codes, uint32_t m, const float* luts,
                                                             uint32_t n_slots, const uint32_t* tie_rank, const uint32_t* candidates,
                                                             uint64_t n_items, uint32_t run, uint32_t stride, uint32_t* sample_out,
                                                             uint32_t n_sample, const float* tau, uint32_t* list_count, uint64_t* list,
                                                             uint32_t list_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem);                 // [m][256][QG]
    const uint32_t slot0 = blockIdx.y * QG;
    const uint32_t nq_here = slot0 + QG <= n_slots ? QG : n_slots - slot0;
    const uint32_t lut_floats = m * 256u;
    for (uint32_t s = 0; s < QG; ++s) {
        const float4* src = reinterpret_cast<const float4*>(luts + static_cast<uint64_t>(slot0 + (s < nq_here ? s : 0)) * lut_floats);
        for (uint32_t i = threadIdx.x; i < lut_floats / 4; i += 1024) {
            const float4 v = s < nq_here ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            lut[(4 * i + 0) * QG + s] = v.x; lut[(4 * i + 1) * QG + s] = v.y; lut[(4 * i + 2) * QG + s] = v.z; lut[(4 * i + 3) * QG + s] = v.w;
        }
    }
    float t[QG];
#pragma unroll
    for (int s = 0; s < QG; ++s) t[s] = (MODE == 2 && static_cast<uint32_t>(s) < nq_here) ? tau[slot0 + s] : 0.f;
    __syncthreads();
    // MODE 1 walks the sample (item it * stride is sample element it), MODE 2 every item
    const uint64_t n_walk = MODE == 1 ? n_sample : n_items;
    const uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * run;
    const uint64_t i1 = i0 + run < n_walk ? i0 + run : n_walk;
    const bool words = (m & 3u) == 0;
    for (uint64_t it = i0 + threadIdx.x; it < i1; it += 1024) {
        const uint64_t item = MODE == 1 ? it * stride : it;
        const uint64_t idx = candidates ? candidates[item] : item;
        const bool live = idx < n_codes;
        const uint8_t* code = codes + (live ? idx : 0) * m;
        // partial sums as PAIRS of queries (QG = 4: two packed fp32 adds per table entry instead of four scalar ones — the same
        // IEEE additions, v_pk_add_f32)
        typedef float pq_f2 __attribute__((ext_vector_type(2)));
        constexpr int NP = (QG + 1) / 2;
        pq_f2 part2[NP][LANES];
#pragma unroll
        for (int s = 0; s < NP; ++s)
#pragma unroll
            for (int l = 0; l < LANES; ++l) part2[s][l] = pq_f2{0.f, 0.f};
        const bool quads = words && (m & 15u) == 0 && ((reinterpret_cast<uintptr_t>(codes) & 15u) == 0);
        for (uint32_t j0 = 0; j0 < m; j0 += 16) {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (quads) {
                const uint4 q4 = *reinterpret_cast<const uint4*>(code + j0);
                w[0] = q4.x; w[1] = q4.y; w[2] = q4.z; w[3] = q4.w;
            } else if (words) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    if (j0 + 4u * tt < m) w[tt] = *reinterpret_cast<const uint32_t*>(code + j0 + 4u * tt);
            } else {
#pragma unroll
                for (int jj = 0; jj < 16; ++jj)
                    if (j0 + jj < m) w[jj >> 2] |= static_cast<uint32_t>(code[j0 + jj]) << (8 * (jj & 3));
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                if (j0 + jj >= m) break;
                const uint32_t c = (w[jj >> 2] >> (8 * (jj & 3))) & 255u;
                const float* e = lut + static_cast<size_t>((j0 + jj) * 256u + c) * QG;
                if constexpr (QG == 4) {
                    typedef float pq_f4 __attribute__((ext_vector_type(4)));
                    const pq_f4 x = *reinterpret_cast<const pq_f4*>(e);
                    part2[0][jj % LANES] += pq_f2{x.x, x.y};
                    part2[1][jj % LANES] += pq_f2{x.z, x.w};
                } else if constexpr (QG == 2) {
                    part2[0][jj % LANES] += *reinterpret_cast<const pq_f2*>(e);
                } else {
                    part2[0][jj % LANES].x = __fadd_rn(part2[0][jj % LANES].x, e[0]);
                }
            }
        }
        float part[QG][LANES];
#pragma unroll
        for (int s = 0; s < QG; ++s)
#pragma unroll
            for (int l = 0; l < LANES; ++l) part[s][l] = (s & 1) ? part2[s >> 1][l].y : part2[s >> 1][l].x;
        uint32_t kidx = 0;
        if (MODE == 2) kidx = tie_rank ? tie_rank[live ? idx : 0] : static_cast<uint32_t>(idx);
#pragma unroll
        for (int s = 0; s < QG; ++s)
            if (static_cast<uint32_t>(s) < nq_here) {
                float sc = part[s][0];
                if (LANES > 1) {
                    sc = 0.f;
#pragma unroll
                    for (int l = 0; l < LANES; ++l) sc = __fadd_rn(sc, part[s][l]);
                }
                // (a score that is not a number sorts nowhere in the reference's comparator: such a row is left out)
                const bool ok = live && sc == sc;
                if (MODE == 1) sample_out[static_cast<uint64_t>(slot0 + s) * n_sample + it] = ok ? f2ord(sc) : 0u;
                else if (ok && sc >= t[s]) {
                    const uint32_t pos = atomicAdd(&list_count[slot0 + s], 1u);
                    if (pos < list_cap) list[static_cast<uint64_t>(slot0 + s) * list_cap + pos] = pack_key(sc, kidx);
                }
            }
    }
}

// mode 1: sample scores (sample_out [n_slots][n_sample], element i = item i * stride); mode 2: keys of the items that reach tau.
// `luts` is the tables of slot 0 on (the caller offsets it): [n_slots][m][256].
hipError_t launch_pq_adc_filter(hipStream_t st, int mode, const uint8_t* codes, uint64_t n_codes, uint32_t m, const float* luts,
                                uint32_t n_slots, int lanes, const uint32_t* tie_rank, const uint32_t* candidates, uint64_t n_items,
                                uint32_t stride, uint32_t* sample_out, uint32_t n_sample, const float* tau, uint32_t* list_count,
                                uint64_t* list, uint32_t list_cap) {
    if (n_items == 0 || n_slots == 0) return hipSuccess;
    const uint32_t qg = static_cast<size_t>(m) * 4096u <= 144u * 1024u ? 4u : (static_cast<size_t>(m) * 2048u <= 144u * 1024u ? 2u : 1u);
    if (static_cast<size_t>(qg) * m * 1024u > 144u * 1024u) return hipErrorInvalidValue; // m > 128 (checked by the caller)
    const size_t sh = static_cast<size_t>(qg) * m * 1024u;
    const uint64_t n_walk = mode == 1 ? n_sample : n_items;
    // runs of 64 K items (64 per thread) amortise loading the tables; shorter ones when that would leave CUs idle
    uint32_t run = 65536;
    while (run > 4096 && (n_walk + run - 1) / run * ((n_slots + qg - 1) / qg) < 512) run >>= 1;
    const dim3 grid(static_cast<uint32_t>((n_walk + run - 1) / run), (n_slots + qg - 1) / qg);
#define YAMS_PQF_LAUNCH(L, Q, M)                                                                                                   \
    do {                                                                                                                           \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&pq_adc_filter_kernel<L, Q, M>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sh)); \
        if (e_ != hipSuccess) return e_;                                                                                            \
        hipLaunchKernelGGL((pq_adc_filter_kernel<L, Q, M>), grid, dim3(1024), sh, st, codes, n_codes, m, luts, n_slots, tie_rank,  \
                           candidates, n_items, run, stride, sample_out, n_sample, tau, list_count, list, list_cap);              \
    } while (0)
#define YAMS_PQF_Q(L, M) do { if (qg == 4) YAMS_PQF_LAUNCH(L, 4, M); else if (qg == 2) YAMS_PQF_LAUNCH(L, 2, M); else YAMS_PQF_LAUNCH(L, 1, M); } while (0)
#define YAMS_PQF_M(L) do { if (mode == 1) YAMS_PQF_Q(L, 1); else YAMS_PQF_Q(L, 2); } while (0)
    if (lanes == 4) YAMS_PQF_M(4);
    else if (lanes == 8) YAMS_PQF_M(8);
    else if (lanes == 16) YAMS_PQF_M(16);
    else YAMS_PQF_M(1);
#undef YAMS_PQF_M
#undef YAMS_PQF_Q
#undef YAMS_PQF_LAUNCH
    return hipGetLastError();
}

hipError_t launch_pq_adc_keys(hipStream_t st, const uint8_t* codes, uint64_t n_codes, uint32_t m, const float* luts, const uint32_t* qmap,
                              uint32_t n_slots, int lanes, const uint32_t* tie_rank, const uint32_t* candidates, uint64_t n_items,
                              uint64_t* keys, uint64_t key_stride) {
    if (n_items == 0 || n_slots == 0) return hipSuccess;
    // tables of up to four queries per workgroup: 1 KiB per sub-quantiser and query, 128 KiB of the CU's LDS at most
    uint32_t qg = 4;
    while (qg > 1 && static_cast<size_t>(qg) * m * 1024u > 128u * 1024u) qg >>= 1;
    if (static_cast<size_t>(qg) * m * 1024u > 128u * 1024u) return hipErrorInvalidValue; // m > 128 (checked by the caller)
    if (qg > n_slots) qg = n_slots >= 2 ? 2 : 1;
    const size_t sh = static_cast<size_t>(qg) * m * 1024u;
    // runs of 16 K indices: 64 per thread — enough to amortise loading the tables, short enough to fill 256 CUs
    uint32_t run = 16384;
    while (run > 1024 && (n_items + run - 1) / run * ((n_slots + qg - 1) / qg) < 1024) run >>= 1;
    const dim3 grid(static_cast<uint32_t>((n_items + run - 1) / run), (n_slots + qg - 1) / qg);
#define YAMS_PQ_LAUNCH(L)                                                                                                          \
    do {                                                                                                                           \
        if (sh > 48u * 1024u) {                                                                                                    \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&pq_adc_keys_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sh)); \
            if (e_ != hipSuccess) return e_;                                                                                        \
        }                                                                                                                          \
        hipLaunchKernelGGL((pq_adc_keys_kernel<L>), grid, dim3(256), sh, st, codes, n_codes, m, luts, qmap, n_slots, qg, tie_rank, \
                           candidates, n_items, run, keys, key_stride);                                                           \
    } while (0)
    if (lanes == 4) YAMS_PQ_LAUNCH(4);
    else if (lanes == 8) YAMS_PQ_LAUNCH(8);
    else if (lanes == 16) YAMS_PQ_LAUNCH(16);
    else YAMS_PQ_LAUNCH(1);
#undef YAMS_PQ_LAUNCH
    return hipGetLastError();
}

} // namespace yams_accel
