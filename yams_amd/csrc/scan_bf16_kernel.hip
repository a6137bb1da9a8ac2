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
// Tile: 256 corpus rows x 256 queries per workgroup (8 waves).  The corpus is read from HBM as
// fp32 (that is the input contract) and split on the VALU at fragment time; queries are split once
// per batch by prep_split_kernel into k-slab-major bf16 planes.  (A first, register-staged version
// of this kernel is in the history of this file; the measurements that replaced it are below.)
#include "scan_args.h"

namespace yams_accel {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int BT_ROWS = 256, BT_QUERIES = 256, BT_THREADS = 512;
// =================================================================================================
// v2: LDS-DMA staged, 4-deep ring, one A tile per wave.
//
// Measurements that shaped it (profiles/r01b_pmc.json):
//   * v1 (register-staged, 2x4 waves): 46 % of wave time parked on s_waitcnt/barrier, MFMA 37 %.
//   * an ablation of the first DMA version with the MFMAs removed still took 36 of 58 ms: the
//     fp32 -> bf16 head/tail split cost ~150 VALU instructions per wave per 16-wide k-slab (done
//     redundantly by the waves that shared rows) and never overlapped the matrix pipe.
// So: RAW operands are staged with LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no
// ds_write pass), three slabs ahead, into a 4-stage ring of 16-wide k-slabs; every wave owns 32
// corpus rows x ALL 256 queries (acc[8]), so each corpus element is split exactly once (8 floats
// per lane per slab, ~28 VALU) and the rest of the loop is LDS reads + 24 MFMAs.  The head is the
// truncated upper half of the fp32 (one v_perm_b32 packs two of them), the tail is the RNE bf16 of
// the exact remainder.  DMAs are issued from inline asm so hipcc does not drain them (it waits
// vmcnt(0) before any ds_read while a builtin LDS-DMA is in flight); completion is tracked by hand
// with counted s_waitcnt vmcnt(N) + s_barrier.  The LDS image is lane-linear per DMA instruction,
// so the bank swizzle is applied to the SOURCE chunk index and again on the fragment read.
// =================================================================================================
constexpr int V2_K = 16, V2_NST = 4;
constexpr int V2_A_BYTES = BT_ROWS * V2_K * 4;          // raw fp32 rows: 64 B per row
constexpr int V2_B_BYTES = BT_QUERIES * V2_K * 2;       // one bf16 plane: 32 B per row
constexpr int V2_STAGE = V2_A_BYTES + 2 * V2_B_BYTES;   // 32 KiB
static_assert(V2_STAGE == 32768, "stage size");
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ABL is a measurement knob (never set by the product path): 1 = no DMA refills after the prologue
// (compute-only time), 2 = no MFMAs (staging-only time); the epilogue then appends nothing.
template <int MODE, int METRIC, int ABL = 0>
__global__ __launch_bounds__(BT_THREADS, 2) void scan_tiles_bf16v2_kernel(ScanArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[V2_NST * V2_STAGE];

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
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const uint64_t row0 = static_cast<uint64_t>(tile) * BT_ROWS;
    const uint32_t q0 = qt * BT_QUERIES;
    const uint32_t dim = a.dim;
    const int nslab = dim / V2_K; // dim % 16 == 0 is a precondition of this kernel

    // ---- DMA source pointers (per lane) and destinations (per wave) -----------------------------
    const unsigned char* srcA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rowA = (wid * 2 + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((rowA >> 2) & 3);
        uint64_t r = row0 + rowA;
        if (r >= a.n_rows) r = a.n_rows - 1;
        srcA[i] = reinterpret_cast<const unsigned char*>(a.rows + r * dim + c * 4);
    }
    const int rowB = wid * 32 + (lane >> 1);
    const int sB = (lane & 1) ^ ((rowB >> 3) & 1);
    // prepared query planes are k-slab-major: [slab][q_pad][16] bf16, so the 32 rows x 32 B one DMA
    // instruction fetches are 1 KiB contiguous (8 full cache lines instead of 32 quarter lines)
    const unsigned char* srcBh = reinterpret_cast<const unsigned char*>(a.q_hi + (static_cast<uint64_t>(q0 + rowB)) * 16 + sB * 8);
    const unsigned char* srcBl = reinterpret_cast<const unsigned char*>(a.q_lo + (static_cast<uint64_t>(q0 + rowB)) * 16 + sB * 8);
    const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 32;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    auto issue = [&](int s) {
        const uint32_t st = lds0 + (s & (V2_NST - 1)) * V2_STAGE;
        const int kb = s * V2_K;
        lds_dma16(srcA[0] + kb * 4, __builtin_amdgcn_readfirstlane(st + (wid * 2 + 0) * 1024));
        lds_dma16(srcA[1] + kb * 4, __builtin_amdgcn_readfirstlane(st + (wid * 2 + 1) * 1024));
        lds_dma16(srcBh + s * qslab_bytes, __builtin_amdgcn_readfirstlane(st + V2_A_BYTES + wid * 1024));
        lds_dma16(srcBl + s * qslab_bytes, __builtin_amdgcn_readfirstlane(st + V2_A_BYTES + V2_B_BYTES + wid * 1024));
    };

    f32x16 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    float nsq = 0.f;

    // fragment read offsets inside a stage: this wave's 32 rows, every query tile
    int offA[2], offB[8];
    {
        const int rf = wid * 32 + l31;
        const int f = (rf >> 2) & 3;
        offA[0] = rf * 64 + (((2 * h) ^ f) << 4);
        offA[1] = rf * 64 + (((2 * h + 1) ^ f) << 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int rq = u * 32 + l31;
        offB[u] = V2_A_BYTES + rq * 32 + ((h ^ ((rq >> 3) & 1)) << 4);
    }

    {
        const int npre = nslab < 3 ? nslab : 3;
        for (int s = 0; s < npre; ++s) issue(s);
    }
    for (int s = 0; s < nslab; ++s) {
        const int rem = nslab - 1 - s; // slabs issued after s that may still be in flight: min(2, rem)
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ABL != 1 && s + 3 < nslab) issue(s + 3); // refills the stage every wave finished reading last iteration
        const unsigned char* base = lds + (s & (V2_NST - 1)) * V2_STAGE;
        // A: 8 raw floats -> head (truncated upper halves, packed by v_perm) + tail (RNE of the remainder)
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(base + offA[0]);
        const u32x4 x1 = *reinterpret_cast<const u32x4*>(base + offA[1]);
        bf16x8 bhi[8], blo[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bhi[u] = *reinterpret_cast<const bf16x8*>(base + offB[u]);
            blo[u] = *reinterpret_cast<const bf16x8*>(base + offB[u] + V2_B_BYTES);
        }
        uint32_t xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        u32x4 hi_p, lo_p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float f0 = __uint_as_float(xs[2 * j]), f1 = __uint_as_float(xs[2 * j + 1]);
            nsq = fmaf(f0, f0, nsq);
            nsq = fmaf(f1, f1, nsq);
            hi_p[j] = __builtin_amdgcn_perm(xs[2 * j + 1], xs[2 * j], 0x07060302u);
            const float r0 = f0 - __uint_as_float(xs[2 * j] & 0xffff0000u);      // exact
            const float r1 = f1 - __uint_as_float(xs[2 * j + 1] & 0xffff0000u);
            typedef __attribute__((ext_vector_type(2))) float f32x2v;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
            const f32x2v rr = {r0, r1};
            const bf16x2v lp = __builtin_convertvector(rr, bf16x2v);
            lo_p[j] = __builtin_bit_cast(uint32_t, lp);
        }
        const bf16x8 ahi = __builtin_bit_cast(bf16x8, hi_p);
        const bf16x8 alo = __builtin_bit_cast(bf16x8, lo_p);
        if (ABL == 2) {
            asm volatile("" :: "v"(ahi), "v"(alo));
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(bhi[u]), "v"(blo[u]));
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[u], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[u], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[u], acc[u], 0, 0, 0);
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    uint32_t qidx[8];
    bool qok[8];
    float qn_up[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        qidx[u] = q0 + u * 32 + l31;
        qok[u] = qidx[u] < a.n_queries;
        qn_up[u] = (METRIC == YAMS_SCAN_L2 && qok[u]) ? a.qnorm_up[qidx[u]] : 0.f;
    }
    {
        const float nfull = nsq + __shfl_xor(nsq, 32);
        float p0, p1 = 0.f;
        const bool ok = norm_in_range(nfull);
        if (METRIC == YAMS_SCAN_COSINE) {
            p0 = ok ? rsqrtf(nfull) : __builtin_nanf("");
        } else {
            p0 = ok ? nfull * (-0.5f + 0.5f * a.err_coef) : __builtin_nanf("");
            p1 = ok ? a.err_coef * sqrtf(nfull) * 1.000001f : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float rp0 = __shfl(p0, i);
            if (METRIC == YAMS_SCAN_COSINE) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u][r] = acc[u][r] * rp0;
            } else {
                const float rp1 = __shfl(p1, i);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u][r] = acc[u][r] + rp0 + rp1 * qn_up[u];
            }
        }
    }

    const uint64_t wave_row0 = row0 + wid * 32;
    if (a.row_mask) { // rows outside the allow-mask never score (they are not part of the scan)
        const uint32_t mw = mask_word(a.row_mask, wave_row0, a.n_rows);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (!((mw >> i) & 1u)) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u][r] = -__builtin_inff();
            }
        }
    }
    if (MODE == MODE_SAMPLE) {
        const float ninf = -__builtin_inff();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float m = ninf;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint64_t rbase = wave_row0 + 8 * g4 + 4 * h;
                float4 v;
                v.x = (rbase + 0 < a.n_rows) ? acc[u][4 * g4 + 0] : ninf;
                v.y = (rbase + 1 < a.n_rows) ? acc[u][4 * g4 + 1] : ninf;
                v.z = (rbase + 2 < a.n_rows) ? acc[u][4 * g4 + 2] : ninf;
                v.w = (rbase + 3 < a.n_rows) ? acc[u][4 * g4 + 3] : ninf;
                m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                if (qok[u]) {
                    const uint64_t srow = static_cast<uint64_t>(sel) * BT_ROWS + wid * 32 + 8 * g4 + 4 * h;
                    *reinterpret_cast<float4*>(a.dense + qidx[u] * a.sample_rows + srow) = v;
                }
            }
            if (qok[u]) {
                const uint32_t gid = ((sel * 8u + wid) << 1) + h;
                a.gmax[static_cast<uint64_t>(qidx[u]) * a.n_groups + gid] = f2ord(m);
            }
        }
    } else {
        float tau[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tau[u] = (qok[u] && ABL == 0) ? a.tau[qidx[u]] : __builtin_inff();
        bool any = false;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) any |= !(acc[u][r] < tau[u]);
        if (any) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sc = acc[u][r];
                    if (!(sc < tau[u])) {
                        const uint64_t row = wave_row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
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

// Split the prepared fp32 queries into bf16 head + tail planes, k-slab-major:
// plane[(k / 16) * q_pad + q][k % 16]; rows q >= n_queries are zero.
__global__ void prep_split_kernel(const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                                  uint16_t* q_hi, uint16_t* q_lo) {
    const uint64_t total = static_cast<uint64_t>(q_pad) * dim;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t kk = static_cast<uint32_t>(i & 15u);
        const uint64_t qs = i >> 4;
        const uint32_t q = static_cast<uint32_t>(qs % q_pad);
        const uint32_t slab = static_cast<uint32_t>(qs / q_pad);
        const float x = q < nq ? qprep[static_cast<uint64_t>(q) * dim + slab * 16 + kk] : 0.f;
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

hipError_t launch_prep_split(hipStream_t st, const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                             uint16_t* q_hi, uint16_t* q_lo) {
    const uint64_t n_elems = static_cast<uint64_t>(q_pad) * dim;
    if (n_elems == 0) return hipSuccess;
    uint32_t grid = static_cast<uint32_t>((n_elems + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(prep_split_kernel, dim3(grid), dim3(256), 0, st, qprep, nq, q_pad, dim, q_hi, q_lo);
    LAUNCH_CHECK();
    return hipSuccess;
}

ScanArgs make_scan_args(const ScanLaunch& L); // scan_kernels.hip

#define LAUNCH_BF16(KERNEL) do { \
    if (mode == MODE_SAMPLE) { \
        if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((KERNEL<MODE_SAMPLE, YAMS_SCAN_COSINE>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
        else hipLaunchKernelGGL((KERNEL<MODE_SAMPLE, YAMS_SCAN_L2>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
    } else { \
        if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((KERNEL<MODE_FILTER, YAMS_SCAN_COSINE>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
        else hipLaunchKernelGGL((KERNEL<MODE_FILTER, YAMS_SCAN_L2>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
    } } while (0)

hipError_t launch_scan_bf16(hipStream_t st, const ScanLaunch& L, int metric, int mode, int version) {
    ScanArgs a = make_scan_args(L);
    a.n_sel_tiles = mode == MODE_SAMPLE ? L.plan.n_sample_tiles : L.plan.n_filter_tiles;
    if (a.n_sel_tiles == 0) return hipSuccess;
    const uint32_t groups = (a.n_sel_tiles + 7) / 8;
    const uint32_t grid = groups * a.n_qtiles * 8;
    if (version == 11 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE)
        hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 1>), dim3(grid), dim3(BT_THREADS), 0, st, a);
    else if (version == 12 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE)
        hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
    else LAUNCH_BF16(scan_tiles_bf16v2_kernel);
    LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace yams_accel
