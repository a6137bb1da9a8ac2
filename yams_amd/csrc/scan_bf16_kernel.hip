// scan_bf16_kernel.hip — the filter pass of the exact scan on the bf16 matrix cores.
//
// Same contract as scan_tiles_kernel (scan_kernels.hip): score every (row, query) pair with a
// rigorously bounded error, keep what can reach the top k; the fp64 re-score + proof that follow
// make the final result bit-identical to the reference (sqlite_vec_backend.cpp:4204-4331).
// The fp32 MFMA roof (157 TFLOP/s) caps the scan below the 10 k QPS target, so this kernel
// splits each fp32 operand into a bf16 head and a bf16 tail (x = hi + lo + O(2^-18 |x|)) and
// accumulates  hi*hi + hi*lo + lo*hi  with v_mfma_f32_32x32x16_bf16 (3 passes at the 2.5 PFLOP/s
// rate; products of bf16 values are exact in fp32, accumulation is fp32).  The error bound E grows
// by the dropped lo*lo term and the split residue (3 * 2^-18) — the proof absorbs it.
//
// Tile: 256 corpus rows x 256 queries per workgroup, 8 waves as 2 (rows) x 4 (queries), each wave
// 128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers).  The corpus is read from HBM as fp32
// (that is the input contract) and split on the VALU while it is staged; queries are split once
// per batch by prep_split_kernel.  LDS: 4 bf16 planes (A_hi, A_lo, B_hi, B_lo) x 256 rows x 32 k,
// double-buffered (128 KiB), 16-byte slots XOR-swizzled by (row >> 2) & 3 so that the
// ds_read_b128 fragment reads of a 16-lane group hit 16 distinct 4-bank groups.
#include "scan_args.h"

namespace yams_accel {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int BT_ROWS = 256, BT_QUERIES = 256, BT_K = 32, BT_THREADS = 512;
constexpr int PLANE_BYTES = BT_ROWS * BT_K * 2;     // 16 KiB
constexpr int STAGE_BYTES = 4 * PLANE_BYTES;        // A_hi, A_lo, B_hi, B_lo
constexpr int NORM_OFF = 2 * STAGE_BYTES;

__device__ __forceinline__ int swz(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int MODE, int METRIC>
__global__ __launch_bounds__(BT_THREADS, 2) void scan_tiles_bf16_kernel(ScanArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE_BYTES + BT_ROWS * 4];

    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u;
    const uint32_t w = bid >> 3;
    const uint32_t qt = w % a.n_qtiles;
    const uint32_t sel = (w / a.n_qtiles) * 8u + xcd;
    if (sel >= a.n_sel_tiles) return;
    uint32_t tile;
    if (MODE == MODE_SAMPLE) tile = sel * a.stride;
    else tile = sel + sel / (a.stride - 1u) + 1u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 2, wq = wid & 3;
    const int h = lane >> 5, l31 = lane & 31;
    const uint64_t row0 = static_cast<uint64_t>(tile) * BT_ROWS;
    const uint32_t q0 = qt * BT_QUERIES;
    const uint32_t dim = a.dim;
    const int nslab = (dim + BT_K - 1) / BT_K;

    // ---- staging map --------------------------------------------------------------------------
    // corpus: 4 float4 per thread per slab: row (tid>>3) + 64 j, k columns (tid&7)*4 .. +3
    const int c4 = tid & 7;
    const float* cptr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint64_t r = row0 + (tid >> 3) + 64 * j;
        if (r >= a.n_rows) r = a.n_rows - 1;
        cptr[j] = a.rows + r * dim + c4 * 4;
    }
    // queries: 4 uint4 (8 bf16) per thread per slab over [plane][row][slot]
    const uint16_t* qptr[4];
    int qdst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + BT_THREADS * j;
        const int s = idx & 3, row = (idx >> 2) & 255, plane = idx >> 10;
        uint32_t q = q0 + row;
        if (q >= a.n_queries) q = a.n_queries - 1;
        qptr[j] = (plane ? a.q_lo : a.q_hi) + static_cast<uint64_t>(q) * dim + s * 8;
        qdst[j] = (2 + plane) * PLANE_BYTES + swz(row, s);
    }
    const bool q_in_k = true;
    (void)q_in_k;

    float4 cst[4];
    uint4 qst[4];
    float nacc[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_slab = [&](int s) {
        const int k0 = s * BT_K;
        const bool cin = k0 + c4 * 4 < static_cast<int>(dim);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cst[j] = cin ? *reinterpret_cast<const float4*>(cptr[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sidx = (tid + BT_THREADS * j) & 3;
            const bool qin = k0 + sidx * 8 < static_cast<int>(dim);
            qst[j] = qin ? *reinterpret_cast<const uint4*>(qptr[j] + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_slab = [&](int buf) {
        unsigned char* base = lds + buf * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = cst[j];
            nacc[j] = fmaf(v.x, v.x, nacc[j]); nacc[j] = fmaf(v.y, v.y, nacc[j]);
            nacc[j] = fmaf(v.z, v.z, nacc[j]); nacc[j] = fmaf(v.w, v.w, nacc[j]);
            const f32x4v f = {v.x, v.y, v.z, v.w};
            const bf16x4 hi = __builtin_convertvector(f, bf16x4);
            const f32x4v back = __builtin_convertvector(hi, f32x4v);
            const f32x4v res = f - back;                       // exact in fp32
            const bf16x4 lo = __builtin_convertvector(res, bf16x4);
            const int R = (tid >> 3) + 64 * j;
            const int off = swz(R, c4 >> 1) + (c4 & 1) * 8;
            *reinterpret_cast<bf16x4*>(base + off) = hi;
            *reinterpret_cast<bf16x4*>(base + PLANE_BYTES + off) = lo;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(base + qdst[j]) = qst[j];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int arow = wr * 128 + l31;
    const int brow = wq * 64 + l31;

    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        if (s + 1 < nslab) load_slab(s + 1);
        const unsigned char* base = lds + (s & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = kk * 2 + h;
            bf16x8 ahi[4], alo[4], bhi[2], blo[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = swz(arow + t * 32, slot);
                ahi[t] = *reinterpret_cast<const bf16x8*>(base + o);
                alo[t] = *reinterpret_cast<const bf16x8*>(base + PLANE_BYTES + o);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int o = swz(brow + u * 32, slot);
                bhi[u] = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_BYTES + o);
                blo[u] = *reinterpret_cast<const bf16x8*>(base + 3 * PLANE_BYTES + o);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], bhi[u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[t], blo[u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[t], bhi[u], acc[t][u], 0, 0, 0);
                }
        }
        if (s + 1 < nslab) store_slab((s + 1) & 1);
        __syncthreads();
    }

    // ---- row norms: reduce the 8 threads of a row, publish through LDS --------------------------
    float* s_norm = reinterpret_cast<float*>(lds + NORM_OFF);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = nacc[j];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        if (c4 == 0) s_norm[(tid >> 3) + 64 * j] = v;
    }
    __syncthreads();

    // ---- epilogue ------------------------------------------------------------------------------
    uint32_t qidx[2];
    bool qok[2];
    float qn_up[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qidx[u] = q0 + wq * 64 + u * 32 + l31;
        qok[u] = qidx[u] < a.n_queries;
        if (METRIC == YAMS_SCAN_L2) qn_up[u] = qok[u] ? a.qnorm_up[qidx[u]] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 n4 = *reinterpret_cast<const float4*>(&s_norm[wr * 128 + t * 32 + 8 * g4 + 4 * h]);
            const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const bool ok = norm_in_range(nn[e]);
                if (METRIC == YAMS_SCAN_COSINE) {
                    const float p0 = ok ? rsqrtf(nn[e]) : __builtin_nanf("");
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u][r] = acc[t][u][r] * p0;
                } else {
                    const float p0 = ok ? nn[e] * (-0.5f + 0.5f * a.err_coef) : __builtin_nanf("");
                    const float p1 = ok ? a.err_coef * sqrtf(nn[e]) * 1.000001f : 0.f;
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u][r] = acc[t][u][r] + p0 + p1 * qn_up[u];
                }
            }
        }

    const uint64_t wave_row0 = row0 + wr * 128;
    if (MODE == MODE_SAMPLE) {
        const float ninf = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float m = ninf;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uint64_t rbase = wave_row0 + t * 32 + 8 * g4 + 4 * h;
                    float4 v;
                    v.x = (rbase + 0 < a.n_rows) ? acc[t][u][4 * g4 + 0] : ninf;
                    v.y = (rbase + 1 < a.n_rows) ? acc[t][u][4 * g4 + 1] : ninf;
                    v.z = (rbase + 2 < a.n_rows) ? acc[t][u][4 * g4 + 2] : ninf;
                    v.w = (rbase + 3 < a.n_rows) ? acc[t][u][4 * g4 + 3] : ninf;
                    m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                    if (qok[u]) {
                        const uint64_t srow = static_cast<uint64_t>(sel) * BT_ROWS + wr * 128 + t * 32 + 8 * g4 + 4 * h;
                        *reinterpret_cast<float4*>(a.dense + qidx[u] * a.sample_rows + srow) = v;
                    }
                }
                if (qok[u]) {
                    const uint32_t gid = ((sel * 8u + wr * 4u + t) << 1) + h;
                    a.gmax[static_cast<uint64_t>(qidx[u]) * a.n_groups + gid] = f2ord(m);
                }
            }
    } else {
        float tau[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) tau[u] = qok[u] ? a.tau[qidx[u]] : __builtin_inff();
        bool any = false;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) any |= !(acc[t][u][r] < tau[u]);
        if (any) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sc = acc[t][u][r];
                        if (!(sc < tau[u])) {
                            const uint64_t row = wave_row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (row < a.n_rows && qok[u]) {
                                const uint32_t pos = atomicAdd(&a.list_count[qidx[u]], 1u);
                                if (pos < a.list_cap)
                                    a.list[static_cast<uint64_t>(qidx[u]) * a.list_cap + pos] =
                                        pack_key(sc, static_cast<uint32_t>(row));
                            }
                        }
                    }
        }
    }
}

// Split the prepared fp32 queries into bf16 head + tail planes.
__global__ void prep_split_kernel(const float* qprep, uint64_t n_elems, uint16_t* q_hi, uint16_t* q_lo) {
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_elems;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float x = qprep[i];
        const __bf16 hi = static_cast<__bf16>(x);
        const float res = x - static_cast<float>(hi);
        const __bf16 lo = static_cast<__bf16>(res);
        q_hi[i] = __builtin_bit_cast(uint16_t, hi);
        q_lo[i] = __builtin_bit_cast(uint16_t, lo);
    }
}

} // namespace yams_accel

#include "scan_launch.h"

namespace yams_accel {

#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_prep_split(hipStream_t st, const float* qprep, uint64_t n_elems, uint16_t* q_hi, uint16_t* q_lo) {
    if (n_elems == 0) return hipSuccess;
    uint32_t grid = static_cast<uint32_t>((n_elems + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(prep_split_kernel, dim3(grid), dim3(256), 0, st, qprep, n_elems, q_hi, q_lo);
    LAUNCH_CHECK();
    return hipSuccess;
}

ScanArgs make_scan_args(const ScanLaunch& L); // scan_kernels.hip

hipError_t launch_scan_bf16(hipStream_t st, const ScanLaunch& L, int metric, int mode) {
    ScanArgs a = make_scan_args(L);
    a.n_sel_tiles = mode == MODE_SAMPLE ? L.plan.n_sample_tiles : L.plan.n_filter_tiles;
    if (a.n_sel_tiles == 0) return hipSuccess;
    const uint32_t groups = (a.n_sel_tiles + 7) / 8;
    const uint32_t grid = groups * a.n_qtiles * 8;
    if (mode == MODE_SAMPLE) {
        if (metric == YAMS_SCAN_COSINE)
            hipLaunchKernelGGL((scan_tiles_bf16_kernel<MODE_SAMPLE, YAMS_SCAN_COSINE>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL((scan_tiles_bf16_kernel<MODE_SAMPLE, YAMS_SCAN_L2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
    } else {
        if (metric == YAMS_SCAN_COSINE)
            hipLaunchKernelGGL((scan_tiles_bf16_kernel<MODE_FILTER, YAMS_SCAN_COSINE>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL((scan_tiles_bf16_kernel<MODE_FILTER, YAMS_SCAN_L2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
    }
    LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace yams_accel
