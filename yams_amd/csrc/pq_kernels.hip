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
