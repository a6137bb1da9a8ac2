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
#include <type_traits>

#include "scan_args.h"
#include "lds_dma.h"

namespace yams_accel {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4v = __attribute__((ext_vector_type(4))) int;

constexpr int BT_ROWS = 256, BT_QUERIES = 256, BT_THREADS = 512;

// Shared epilogue of the bf16 filter kernels: norms -> scores, allow-mask, then either the sample
// outputs (dense scores + group maxima) or the threshold test + candidate append.
// One call handles a block of 32 rows x (32 * NCB) queries held by one wave:
// acc[u][r] = dot(row row0 + row_in_tile + (r&3) + 8(r>>2) + 4h, query q0 + 32u + l31); nfull = the
// fp32 squared norm of row (lane & 31) of the block.
// PRENORM: the A operand was the unit-normalised shadow row, so a cosine score needs no scaling
// (rows whose norm is out of range still turn into NaN = "always a candidate"); L2 scales the dot
// back by the row norm.
// PHASE (FILTER mode): 0 = everything; 1 = up to the list reservations (survivor masks and reserved
// positions are returned in pend_pass / pend_base); 2 = only the stores of a call that ran phase 1.
// A wave that holds several row blocks runs phase 1 for all of them before the first phase 2, so
// their reservation round trips overlap.
template <int MODE, int METRIC, int ABL, int NCB, bool PRENORM = false, int PHASE = 0>
__device__ __forceinline__ void bf16_epilogue(const ScanArgs& a, f32x16 (&acc)[NCB], float nfull,
                                              uint64_t row0, uint32_t row_in_tile, uint32_t q0,
                                              uint32_t sel, int h, int l31,
                                              const float* tau_pre = nullptr /* [NCB], preloaded */,
                                              uint32_t* pend_pass = nullptr, uint32_t* pend_base = nullptr) {
    static_assert(PHASE == 0 || MODE == MODE_FILTER, "phases exist for the filter epilogue only");
    if (ABL != 0) { // measurement builds: keep the accumulators alive, emit nothing
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < NCB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[u][r];
        if (t + nfull == 12345.678f && a.list_cap == 0xffffffffu) a.list_count[0] = 1;
        return;
    }
    // ---- epilogue ------------------------------------------------------------------------------
    uint32_t qidx[NCB];
    bool qok[NCB];
    float qn_up[NCB];
#pragma unroll
    for (int u = 0; u < NCB; ++u) {
        qidx[u] = q0 + u * 32 + l31;
        qok[u] = qidx[u] < a.n_queries;
        qn_up[u] = (METRIC == YAMS_SCAN_L2 && qok[u]) ? a.qnorm_up[qidx[u]] : 0.f;
    }
    const bool ok = norm_in_range(nfull);
    const bool any_bad = __builtin_amdgcn_ballot_w64(!ok) != 0; // wave-uniform
    if (PHASE != 2 && (!(PRENORM && METRIC == YAMS_SCAN_COSINE) || any_bad)) {
        float p0, p1 = 0.f, ps = 1.f;
        if (METRIC == YAMS_SCAN_COSINE) {
            p0 = ok ? (PRENORM ? 1.f : rsqrtf(nfull)) : __builtin_nanf("");
        } else {
            p0 = ok ? nfull * (-0.5f + 0.5f * a.err_coef) : __builtin_nanf("");
            p1 = ok ? a.err_coef * sqrtf(nfull) * 1.000001f : 0.f;
            if (PRENORM) ps = ok ? sqrtf(nfull) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float rp0 = __shfl(p0, i);
            if (METRIC == YAMS_SCAN_COSINE) {
#pragma unroll
                for (int u = 0; u < NCB; ++u) acc[u][r] = acc[u][r] * rp0;
            } else {
                const float rp1 = __shfl(p1, i);
                if (PRENORM) {
                    const float rps = __shfl(ps, i);
#pragma unroll
                    for (int u = 0; u < NCB; ++u) acc[u][r] = acc[u][r] * rps + rp0 + rp1 * qn_up[u];
                } else {
#pragma unroll
                    for (int u = 0; u < NCB; ++u) acc[u][r] = acc[u][r] + rp0 + rp1 * qn_up[u];
                }
            }
        }
    }

    const uint64_t wave_row0 = row0 + row_in_tile;
    if (PHASE != 2 && a.row_mask) { // rows outside the allow-mask never score (they are not part of the scan)
        const uint32_t mw = mask_word(a.row_mask, wave_row0, a.n_rows);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (!((mw >> i) & 1u)) {
#pragma unroll
                for (int u = 0; u < NCB; ++u) acc[u][r] = -__builtin_inff();
            }
        }
    }
    if (MODE == MODE_SAMPLE) {
        const float ninf = -__builtin_inff();
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            float m = ninf;
            bool bad = false; // a NaN score marks the whole group (see scan_kernels.hip)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint64_t rbase = wave_row0 + 8 * g4 + 4 * h;
                float4 v;
                v.x = (rbase + 0 < a.n_rows) ? acc[u][4 * g4 + 0] : ninf;
                v.y = (rbase + 1 < a.n_rows) ? acc[u][4 * g4 + 1] : ninf;
                v.z = (rbase + 2 < a.n_rows) ? acc[u][4 * g4 + 2] : ninf;
                v.w = (rbase + 3 < a.n_rows) ? acc[u][4 * g4 + 3] : ninf;
                m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
                if (qok[u]) {
                    const uint64_t srow = static_cast<uint64_t>(sel) * BT_ROWS + row_in_tile + 8 * g4 + 4 * h;
                    *reinterpret_cast<float4*>(a.dense + dense_index(qidx[u], srow, a.n_queries)) = v;
                }
            }
            if (qok[u]) {
                const uint32_t gid = (sel * BT_ROWS + row_in_tile) / 16u + h;
                a.gmax[static_cast<uint64_t>(qidx[u]) * a.n_groups + gid] = bad ? 0xffffffffu : f2ord(m);
            }
        }
    } else {
        uint32_t pass[NCB], base[NCB];
        if (PHASE == 2) {
#pragma unroll
            for (int u = 0; u < NCB; ++u) { pass[u] = pend_pass[u]; base[u] = pend_base[u]; }
        } else {
            float tau[NCB];
#pragma unroll
            for (int u = 0; u < NCB; ++u)
                tau[u] = tau_pre ? tau_pre[u] : ((qok[u] && ABL == 0) ? a.tau[qidx[u]] : __builtin_inff());
            // cheap reject: the maximum of a lane's 16 scores per query block (v_max3 tree); a NaN score
            // (ignored by max) can only come from a row flagged above, which forces the full scan
            uint32_t hot = 0;
#pragma unroll
            for (int u = 0; u < NCB; ++u) {
                float m = acc[u][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[u][r]);
                if (!(m < tau[u]) || any_bad) hot |= 1u << u;
            }
            // A lane reserves room for ALL its survivors of a query block with one atomic, and the
            // reservations of its NCB blocks are issued back to back before the first one is
            // needed: one memory round trip per call instead of one per survivor.  (Small corpora
            // have the same ~1000 survivors per query spread over far fewer tiles: with one
            // round trip per survivor the epilogue was 2/3 of a 1M-row scan.)
#pragma unroll
            for (int u = 0; u < NCB; ++u) {
                pass[u] = 0u;
                if (((hot >> u) & 1u) && qok[u]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint64_t row = wave_row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (!(acc[u][r] < tau[u]) && row < a.n_rows) pass[u] |= 1u << r;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NCB; ++u) {
                base[u] = 0u;
                if (pass[u]) base[u] = atomicAdd(&a.list_count[qidx[u]], static_cast<uint32_t>(__builtin_popcount(pass[u])));
            }
            if (PHASE == 1) {
#pragma unroll
                for (int u = 0; u < NCB; ++u) { pend_pass[u] = pass[u]; pend_base[u] = base[u]; }
                return;
            }
        }
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            if (!pass[u]) continue;
            uint32_t pos = base[u];
            uint64_t* lst = a.list + static_cast<uint64_t>(qidx[u]) * a.list_cap;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!((pass[u] >> r) & 1u)) continue;
                const uint64_t row = wave_row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (pos < a.list_cap) lst[pos] = pack_key(acc[u][r], static_cast<uint32_t>(row));
                ++pos;
            }
        }
    }
}


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
// stage = raw A rows + PASSES==3 ? (query head + tail planes) : (query head plane)
constexpr int v2_stage_bytes(int passes) { return V2_A_BYTES + (passes == 3 ? 2 : 1) * V2_B_BYTES; }
static_assert(v2_stage_bytes(3) == 32768 && v2_stage_bytes(1) == 24576, "stage size");
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// ABL is a measurement knob (never set by the product path): 2 = no MFMAs (staging-only time);
// the epilogue then appends nothing.
// PASSES: 3 = split operands, hi*hi + hi*lo + lo*hi (error ~2^-14.4 |x||q|);
//         1 = RNE bf16 operands, one MFMA pass (error ~2^-7 |x||q|: more candidates to re-score,
//             a third of the matrix work and no tail plane in LDS).
template <int MODE, int METRIC, int PASSES, int ABL = 0>
__global__ __launch_bounds__(BT_THREADS, 2) void scan_tiles_bf16v2_kernel(ScanArgs a) {
    constexpr int V2_STAGE = v2_stage_bytes(PASSES);
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
        if (PASSES == 3)
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
        // DMA instructions per slab: 4 (PASSES == 3) or 3
        if (PASSES == 3) {
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + 3 < nslab) issue(s + 3); // refills the stage every wave finished reading last iteration
        const unsigned char* base = lds + (s & (V2_NST - 1)) * V2_STAGE;
        // A: 8 raw floats -> head (truncated upper halves, packed by v_perm) + tail (RNE of the remainder)
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(base + offA[0]);
        const u32x4 x1 = *reinterpret_cast<const u32x4*>(base + offA[1]);
        bf16x8 bhi[8], blo[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bhi[u] = *reinterpret_cast<const bf16x8*>(base + offB[u]);
            if (PASSES == 3) blo[u] = *reinterpret_cast<const bf16x8*>(base + offB[u] + V2_B_BYTES);
        }
        uint32_t xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        typedef __attribute__((ext_vector_type(2))) float f32x2v;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
        u32x4 hi_p, lo_p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float f0 = __uint_as_float(xs[2 * j]), f1 = __uint_as_float(xs[2 * j + 1]);
            nsq = fmaf(f0, f0, nsq);
            nsq = fmaf(f1, f1, nsq);
            if (PASSES == 3) {
                hi_p[j] = __builtin_amdgcn_perm(xs[2 * j + 1], xs[2 * j], 0x07060302u);
                const float r0 = f0 - __uint_as_float(xs[2 * j] & 0xffff0000u);      // exact
                const float r1 = f1 - __uint_as_float(xs[2 * j + 1] & 0xffff0000u);
                const f32x2v rr = {r0, r1};
                const bf16x2v lp = __builtin_convertvector(rr, bf16x2v);
                lo_p[j] = __builtin_bit_cast(uint32_t, lp);
            } else {
                const f32x2v ff = {f0, f1};
                const bf16x2v hp = __builtin_convertvector(ff, bf16x2v); // RNE, one v_cvt_pk_bf16_f32
                hi_p[j] = __builtin_bit_cast(uint32_t, hp);
                lo_p[j] = 0;
            }
        }
        const bf16x8 ahi = __builtin_bit_cast(bf16x8, hi_p);
        const bf16x8 alo = __builtin_bit_cast(bf16x8, lo_p);
        if (ABL == 2) {
            asm volatile("" :: "v"(ahi), "v"(alo));
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(bhi[u]));
        } else {
            if (PASSES == 3) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[u], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[u], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[u], acc[u], 0, 0, 0);
        }
    }

    bf16_epilogue<MODE, METRIC, ABL, 8>(a, acc, nsq + __shfl_xor(nsq, 32), row0, static_cast<uint32_t>(wid) * 32u, q0, sel, h, l31);
}

// =================================================================================================
// Single-pass kernel, 32-wide k-slabs, 2x4 wave tiles, software-pipelined fragment reads.
//
// With one MFMA pass the matrix pipe needs only ~8 ms for the bench shard; what bounds the kernel is
// LDS: staging writes plus fragment reads.  Measurements that shaped it:
//   * 16-wide slabs fetched HALF a 128-byte line of a corpus row per DMA request (the other half
//     again one slab later).  Here a slab is 32 k-values: one DMA instruction covers 8 rows x 128 B
//     (whole lines); a stage is 32 KiB of raw fp32 rows + 16 KiB of RNE-bf16 query heads, 3 stages.
//   * one 32-row block x all 256 queries per wave re-read the whole query tile in every wave
//     (20 KiB of LDS reads per wave per slab, 160 KiB per workgroup: more LDS time than MFMA time).
//     Here a wave owns 64 rows x 128 queries (acc[2][4]): 16 KiB per wave per slab.
//   * hipcc interleaved `2 ds_read, wait, 2 MFMA` — every MFMA pair waited a full LDS latency.
//     Here the fragments of the NEXT 16-wide step (also across the slab boundary: the barrier sits
//     in the middle of the iteration) are requested before the MFMAs of the current one, with
//     sched_barrier fences so the compiler keeps that order.
// LDS image per row: 8 chunks of 16 B, logical chunk c stored at position c ^ ((row >> 1) & 7);
// per query: 4 chunks, logical chunk c at c ^ ((q >> 2) & 3) — both make the b128 fragment reads
// of 16 consecutive rows / queries hit 16 distinct bank groups.
// =================================================================================================
constexpr int K32 = 32, K32_NST = 3;
constexpr int K32_A_BYTES = BT_ROWS * K32 * 4;     // 32 KiB
constexpr int K32_B_BYTES = BT_QUERIES * K32 * 2;  // 16 KiB
constexpr int K32_STAGE = K32_A_BYTES + K32_B_BYTES;

struct K32Frags { u32x4 a[2][2]; bf16x8 b[4]; }; // one 16-wide k-step: 2 row blocks (raw fp32), 4 query blocks

template <int MODE, int METRIC, int ABL = 0>
__global__ __launch_bounds__(BT_THREADS, 2) void scan_tiles_bf16k32_kernel(ScanArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[K32_NST * K32_STAGE];

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
    const int wr = wid >> 1, wc = wid & 1; // wave tile: rows [64 wr, +64) x queries [128 wc, +128)
    const int h = lane >> 5, l31 = lane & 31;
    const uint64_t row0 = static_cast<uint64_t>(tile) * BT_ROWS;
    const uint32_t q0 = qt * BT_QUERIES;
    const uint32_t dim = a.dim;
    const int nslab = dim / K32; // dim % 32 == 0 is a precondition of this kernel

    // ---- DMA sources: every wave stages 32 rows (4 instructions) and 32 queries (2) -------------
    const unsigned char* srcA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rowA = wid * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((rowA >> 1) & 7);
        uint64_t r = row0 + rowA;
        if (r >= a.n_rows) r = a.n_rows - 1;
        srcA[i] = reinterpret_cast<const unsigned char*>(a.rows + r * dim + c * 4);
    }
    const unsigned char* srcB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rowB = wid * 32 + i * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((rowB >> 2) & 3);
        // k-slab-major plane: [slab][q_pad][32] bf16 -> 16 queries x 64 B are 1 KiB contiguous
        srcB[i] = reinterpret_cast<const unsigned char*>(a.q_hi + (static_cast<uint64_t>(q0 + rowB)) * 32 + c * 8);
    }
    const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    // measurement builds: ABL 1/3/4 refill nothing after the prologue (stages keep valid data);
    // 2/3 skip the MFMAs; 4 skips the fragment reads
    auto issue = [&](int s) {
        if ((ABL == 1 || ABL == 3 || ABL == 4) && s >= K32_NST) return;
        const uint32_t st = lds0 + (s % K32_NST) * K32_STAGE;
        const int kb = s * K32;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            lds_dma16(srcA[i] + kb * 4, __builtin_amdgcn_readfirstlane(st + (wid * 4 + i) * 1024));
#pragma unroll
        for (int i = 0; i < 2; ++i)
            lds_dma16(srcB[i] + s * qslab_bytes, __builtin_amdgcn_readfirstlane(st + K32_A_BYTES + (wid * 2 + i) * 1024));
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][u][r] = 0.f;
    float nsq[2] = {0.f, 0.f};

    // fragment offsets inside a stage, per 16-wide step t: A logical chunks 4t + 2h (+1), B chunk 2t + h
    int offA[2][2][2], offB[4][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int rf = wr * 64 + rb * 32 + l31;
        const int f = (rf >> 1) & 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            offA[rb][t][0] = rf * 128 + (((4 * t + 2 * h) ^ f) << 4);
            offA[rb][t][1] = rf * 128 + (((4 * t + 2 * h + 1) ^ f) << 4);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rq = wc * 128 + u * 32 + l31;
        const int f = (rq >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) offB[u][t] = K32_A_BYTES + rq * 64 + (((2 * t + h) ^ f) << 4);
    }
    auto load = [&](K32Frags& fr, const unsigned char* base, int t) {
        if (ABL == 4 && base != lds) return;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            fr.a[rb][0] = *reinterpret_cast<const u32x4*>(base + offA[rb][t][0]);
            fr.a[rb][1] = *reinterpret_cast<const u32x4*>(base + offA[rb][t][1]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) fr.b[u] = *reinterpret_cast<const bf16x8*>(base + offB[u][t]);
    };
    typedef __attribute__((ext_vector_type(2))) float f32x2v;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
    auto compute = [&](const K32Frags& fr) {
        bf16x8 av[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const uint32_t xs[8] = {fr.a[rb][0][0], fr.a[rb][0][1], fr.a[rb][0][2], fr.a[rb][0][3],
                                    fr.a[rb][1][0], fr.a[rb][1][1], fr.a[rb][1][2], fr.a[rb][1][3]};
            u32x4 ap;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f0 = __uint_as_float(xs[2 * j]), f1 = __uint_as_float(xs[2 * j + 1]);
                nsq[rb] = fmaf(f0, f0, nsq[rb]);
                nsq[rb] = fmaf(f1, f1, nsq[rb]);
                const f32x2v ff = {f0, f1};
                const bf16x2v hp = __builtin_convertvector(ff, bf16x2v); // RNE, one v_cvt_pk_bf16_f32
                ap[j] = __builtin_bit_cast(uint32_t, hp);
            }
            av[rb] = __builtin_bit_cast(bf16x8, ap);
        }
        if (ABL == 2 || ABL == 3) {
            asm volatile("" :: "v"(av[0]), "v"(av[1]));
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" :: "v"(fr.b[u]));
        } else {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[rb][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[rb], fr.b[u], acc[rb][u], 0, 0, 0);
        }
    };

    // ---- prologue: all three stages in flight, slab 0 landed, its first step requested -----------
    {
        const int npre = nslab < K32_NST ? nslab : K32_NST;
        for (int s = 0; s < npre; ++s) issue(s);
        if (npre == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (npre == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    K32Frags f0, f1;
    load(f0, lds, 0);
    int stage = 0;
    // One iteration = one slab: second step requested, first step computed, barrier for the next
    // slab in the MIDDLE (so fragment reads cross the slab boundary), refill, next slab's first
    // step requested, second step computed.  The steady-state loop body is a single basic block
    // (ISSUE / VM are compile-time) so the compiler's lgkmcnt bookkeeping stays exact: it waits for
    // the step it needs, not for the prefetch behind it.
    auto body = [&](int s, auto issue_tag, auto vm_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int VM = decltype(vm_tag)::value;
        const unsigned char* base = lds + stage * K32_STAGE;
        stage = stage + 1 == K32_NST ? 0 : stage + 1;
        load(f1, base, 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(f0);
        __builtin_amdgcn_sched_barrier(0);
        // slab s+1 must have landed (VM newer DMA instructions may still be in flight) and every
        // wave must be done reading slab s (f1 is complete at lgkmcnt(0)) before its stage is
        // refilled with slab s+3
        if (VM == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ISSUE) issue(s + 3);
        load(f0, lds + stage * K32_STAGE, 0);
        __builtin_amdgcn_sched_barrier(0);
        compute(f1);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;
    using V6 = std::integral_constant<int, 6>;
    using V0 = std::integral_constant<int, 0>;
    int s = 0;
    for (; s + 3 < nslab; ++s) body(s, T{}, V6{});
    if (s + 2 < nslab) { body(s, F{}, V6{}); ++s; }
    if (s + 1 < nslab) { body(s, F{}, V0{}); ++s; }
    {   // last slab: nothing left to wait for or to prefetch
        load(f1, lds + stage * K32_STAGE, 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(f0);
        compute(f1);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
        bf16_epilogue<MODE, METRIC, ABL, 4>(a, acc[rb], nsq[rb] + __shfl_xor(nsq[rb], 32), row0,
                                            static_cast<uint32_t>(wr * 64 + rb * 32), q0 + wc * 128, sel, h, l31);
}

// =================================================================================================
// Single-pass kernel over the bf16 SHADOW of the corpus.
//
// Ablations of the kernel above on the bench shard (scripts/filter_ablation.py): product 24.8 ms;
// MFMAs + barriers alone 17.6 ms (the fp32 -> bf16 conversion and the norm FMAs sit in front of
// every MFMA group and both waves of a SIMD do them at the same time, so the matrix pipe idles);
// staging + fragment reads alone 15.3 ms (48 KiB per slab through a ~22 B/clk/CU load path).
// Both go away when the corpus side is prepared once, like the query side: a row-major bf16 (RNE)
// copy of the unit-normalised rows plus their fp32 squared norms, built by shadow_build_kernel
// when the mirror is uploaded (so a cosine score needs no scaling in the epilogue either).  The filter then stages 64 B per row per slab instead of 128, reads MFMA operands
// straight from LDS and has no VALU work in its loop; the fp64 re-score still reads the original
// fp32 rows, so results stay bit-identical (the error bound is the same 2^-7 |x||q|).
// Stage = 16 KiB rows + 16 KiB queries, four stages (three slabs = 96 k-values in flight).
// =================================================================================================
constexpr int SH_K = 32, SH_NST = 4;
constexpr int SH_A_BYTES = BT_ROWS * SH_K * 2;     // 16 KiB
constexpr int SH_B_BYTES = BT_QUERIES * SH_K * 2;  // 16 KiB
constexpr int SH_STAGE = SH_A_BYTES + SH_B_BYTES;

struct ShFrags { bf16x8 a[2]; bf16x8 b[4]; };

// NW = waves per workgroup: 8 -> 4x2 waves of 64 rows x 128 queries (two waves per SIMD),
//                            4 -> 2x2 waves of 128 rows x 128 queries (one wave per SIMD, 256
//                                 accumulator registers): a third less fragment-read traffic
//                                 (64 instead of 96 KiB per slab), twice the DMA pieces per wave.
template <int MODE, int METRIC, int ABL = 0, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void scan_tiles_bf16s_kernel(ScanArgs a) {
    constexpr int RB = 16 / NW;       // 32-row blocks per wave (2 or 4)
    constexpr int PA = 16 / NW;       // DMA pieces per wave per operand per slab (2 or 4)
    constexpr int P = 2 * PA;         // DMA pieces per wave per slab
    constexpr int M = RB * 4;         // MFMAs per 16-wide step
    __shared__ __attribute__((aligned(16))) unsigned char lds[SH_NST * SH_STAGE];

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
    const int wr = wid >> 1, wc = wid & 1; // wave tile: rows [32 RB wr, +32 RB) x queries [128 wc, +128)
    const int h = lane >> 5, l31 = lane & 31;
    const uint64_t row0 = static_cast<uint64_t>(tile) * BT_ROWS;
    const uint32_t q0 = qt * BT_QUERIES;
    const uint32_t dim = a.dim;
    const int nslab = dim / SH_K; // dim % 32 == 0 is a precondition of this kernel

    // ---- DMA sources: every wave stages 16 PA rows and 16 PA queries per slab --------------------
    // address = uniform base (tile / query-tile start + slab offset, SGPRs) + per-lane byte offset
    uint32_t voffA[PA], voffB[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int rowA = (wid * PA + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((rowA >> 2) & 3);
        uint64_t r = row0 + rowA;
        if (r >= a.n_rows) r = a.n_rows - 1; // row0 < n_rows, so r - row0 >= 0
        voffA[i] = static_cast<uint32_t>(r - row0) * dim * 2u + c * 16u;
        voffB[i] = static_cast<uint32_t>(rowA) * 64u + c * 16u;
    }
    const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.rows_bf16 + row0 * dim);
    const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.q_hi + static_cast<uint64_t>(q0) * 32);
    const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    const uint32_t ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * PA * 1024);
    const uint32_t ldsB = __builtin_amdgcn_readfirstlane(lds0 + SH_A_BYTES + wid * PA * 1024);
    // One slab = P DMA pieces per wave (p < PA: rows; else queries).  All 32 pieces of a slab
    // issued in one burst after the barrier fill the CU's load queue and stall every wave in front
    // of its MFMAs (measured: adding the refills to an LDS-read + MFMA loop added their whole
    // stand-alone time), so the pieces are issued one at a time between MFMAs, half a slab per step.
    auto piece = [&](int s, int p) {
        if ((ABL == 1 || ABL == 3 || ABL == 4 || ABL == 5) && s >= SH_NST) return; // measurement builds, see above
        const uint32_t st = (s & (SH_NST - 1)) * SH_STAGE;
        if (p < PA) lds_dma16_s(baseA + s * (SH_K * 2), voffA[p < PA ? p : 0], ldsA + st + p * 1024);
        else lds_dma16_s(baseB + s * qslab_bytes, voffB[p >= PA ? p - PA : 0], ldsB + st + (p - PA) * 1024);
    };

    f32x16 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][u][r] = 0.f;

    // fragment offsets inside a stage, per 16-wide step t: logical chunk 2t + h of the row / query,
    // stored at position chunk ^ ((index >> 2) & 3)
    int offA[RB][2], offB[4][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int rf = wr * (32 * RB) + rb * 32 + l31;
        const int f = (rf >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) offA[rb][t] = rf * 64 + (((2 * t + h) ^ f) << 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rq = wc * 128 + u * 32 + l31;
        const int f = (rq >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) offB[u][t] = SH_A_BYTES + rq * 64 + (((2 * t + h) ^ f) << 4);
    }
    struct Frags { bf16x8 a[RB]; bf16x8 b[4]; };
    auto load = [&](Frags& fr, const unsigned char* base, int t) {
        if ((ABL == 4 || ABL == 5) && base != lds) return;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fr.a[rb] = *reinterpret_cast<const bf16x8*>(base + offA[rb][t]);
#pragma unroll
        for (int u = 0; u < 4; ++u) fr.b[u] = *reinterpret_cast<const bf16x8*>(base + offB[u][t]);
    };
    auto pin = [&](Frags& fr) {
        if constexpr (RB == 2)
            asm volatile("" : "+v"(fr.a[0]), "+v"(fr.a[1]), "+v"(fr.b[0]), "+v"(fr.b[1]), "+v"(fr.b[2]), "+v"(fr.b[3]));
        else
            asm volatile("" : "+v"(fr.a[0]), "+v"(fr.a[1]), "+v"(fr.a[RB - 2]), "+v"(fr.a[RB - 1]), "+v"(fr.b[0]),
                         "+v"(fr.b[1]), "+v"(fr.b[2]), "+v"(fr.b[3]));
    };
    // One 16-wide step: M MFMAs on `cur`, with the RB + 4 fragment reads of the NEXT step and (when
    // slab_dma >= 0) DMA pieces p0 .. p0 + PA - 1 of that slab placed one per MFMA gap.  A burst of
    // reads in front of the MFMAs leaves the matrix pipe idle when both waves of a SIMD are in it at
    // the same time (a bare MFMA stream runs 32 cycles per instruction, micro-benchmark
    // scripts/ubench/mfma_rate.hip; this loop ran 43); up to ~5 single-issue instructions hide in
    // each 32-cycle gap.
    auto step = [&](const Frags& cur, Frags& nxt, const unsigned char* nbase, int nt, bool do_load,
                    int slab_dma, int p0) {
        const bool rd = do_load && !((ABL == 4 || ABL == 5) && nbase != lds);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int rb = i >> 2, u = i & 3;
            if (ABL == 2 || ABL == 3) {
                if (i == 0) {
#pragma unroll
                    for (int v = 0; v < RB; ++v) asm volatile("" :: "v"(cur.a[v]));
#pragma unroll
                    for (int v = 0; v < 4; ++v) asm volatile("" :: "v"(cur.b[v]));
                }
            } else {
                acc[rb][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[rb], cur.b[u], acc[rb][u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (rd && i < RB) nxt.a[i < RB ? i : 0] = *reinterpret_cast<const bf16x8*>(nbase + offA[i < RB ? i : 0][nt]);
            if (rd && i >= RB && i < RB + 4) nxt.b[i - RB] = *reinterpret_cast<const bf16x8*>(nbase + offB[i - RB][nt]);
            // pieces after MFMA 3, 6 (8-MFMA steps) / 3, 7, 11, 14 (16-MFMA steps)
            if (slab_dma >= 0) {
                if constexpr (M == 8) {
                    if (i == 3 || i == 6) piece(slab_dma, p0 + (i == 6 ? 1 : 0));
                } else {
                    if (i == 3 || i == 7 || i == 11 || i == 14) piece(slab_dma, p0 + (i == 3 ? 0 : i == 7 ? 1 : i == 11 ? 2 : 3));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // epilogue inputs, requested now: a load issued at the end would sit on the critical path of
    // every tile (rows_nsq streams from HBM), here it hides under the whole k loop
    float nfull_pre[RB], tau_pre[4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const uint64_t r = row0 + static_cast<uint32_t>(wr * (32 * RB) + rb * 32) + l31;
        nfull_pre[rb] = r < a.n_rows ? a.rows_nsq[r] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t qi = q0 + wc * 128 + u * 32 + l31;
        tau_pre[u] = (MODE == MODE_FILTER && ABL == 0 && qi < a.n_queries) ? a.tau[qi] : __builtin_inff();
    }
    // (these older loads complete before any DMA piece issued below — vmcnt retires in order — so the
    // counted waits in the loop stay valid; the compiler waits for them at their first use, the
    // epilogue)

    // ---- prologue: slabs 0..2 and the first half of slab 3 in flight, slab 0 landed ---------------
    auto wait_vm = [&](int pieces) { // s_waitcnt vmcnt(pieces) for the few values the schedule needs
        if (pieces >= 5 * PA) { if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
        else if (pieces >= 4 * PA) { if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        else if (pieces >= 2 * PA) { if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    {
        int issued = 0;
        for (int s = 0; s < 3 && s < nslab; ++s) { for (int p = 0; p < P; ++p) piece(s, p); issued += P; }
        if (nslab > 3) { for (int p = 0; p < PA; ++p) piece(3, p); issued += PA; }
        wait_vm(issued - P);
        __builtin_amdgcn_s_barrier();
    }
    Frags f0, f1;
    load(f0, lds, 0);
    int stage = 0;
    // Same half-shifted pipeline as above: the barrier for slab s+1 sits between the two steps of
    // slab s; the steady-state body is one basic block.  First step: second half of slab s+3 goes
    // out (its stage was released by the barrier of the previous iteration); second step: first
    // half of slab s+4 (released by this iteration's barrier).
    auto body = [&](int s, auto h1_tag, auto h2_tag, auto vm_tag) {
        constexpr bool H1 = decltype(h1_tag)::value;
        constexpr bool H2 = decltype(h2_tag)::value;
        constexpr int VM = decltype(vm_tag)::value; // in slabs: 2, 1 or 0 newer slabs may be in flight
        const unsigned char* base = lds + stage * SH_STAGE;
        stage = (stage + 1) & (SH_NST - 1);
        // f0 was requested during the previous step: make the compiler place its (conservative,
        // whole-counter) LDS wait HERE, before the next reads go out, not in front of an MFMA
        pin(f0);
        __builtin_amdgcn_sched_barrier(0);
        step(f0, f1, base, 1, true, H1 ? s + 3 : -1, PA);
        // slab s+1 landed (VM slabs of newer DMA pieces may be in flight); all reads of slab s
        // (f1 last) complete before its stage is refilled with slab s+4
        if constexpr (P == 4) {
            if (VM == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (VM == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            if (VM == 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (VM == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        if (ABL != 5) __builtin_amdgcn_s_barrier();
        pin(f1);
        __builtin_amdgcn_sched_barrier(0);
        step(f1, f0, lds + stage * SH_STAGE, 0, true, H2 ? s + 4 : -1, 0);
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;
    using V2 = std::integral_constant<int, 2>;
    using V1 = std::integral_constant<int, 1>;
    using V0 = std::integral_constant<int, 0>;
    int s = 0;
    for (; s + 4 < nslab; ++s) body(s, T{}, T{}, V2{});
    if (s + 3 < nslab) { body(s, T{}, F{}, V2{}); ++s; }
    if (s + 2 < nslab) { body(s, F{}, F{}, V1{}); ++s; }
    if (s + 1 < nslab) { body(s, F{}, F{}, V0{}); ++s; }
    {   // last slab: nothing left to wait for or to prefetch after its second step
        pin(f0);
        __builtin_amdgcn_sched_barrier(0);
        step(f0, f1, lds + stage * SH_STAGE, 1, true, -1, 0);
        pin(f1);
        __builtin_amdgcn_sched_barrier(0);
        step(f1, f0, lds, 0, false, -1, 0);
    }
    if constexpr (MODE == MODE_FILTER && ABL == 0 && METRIC == YAMS_SCAN_COSINE) {
        // reservations of all row blocks first, then their stores (overlapping round trips); the L2
        // epilogue has no registers to spare for the pending masks and keeps one round trip per block
        uint32_t ppass[RB][4], pbase[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            bf16_epilogue<MODE, METRIC, ABL, 4, true, 1>(a, acc[rb], nfull_pre[rb], row0, static_cast<uint32_t>(wr * (32 * RB) + rb * 32),
                                                         q0 + wc * 128, sel, h, l31, tau_pre, ppass[rb], pbase[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            bf16_epilogue<MODE, METRIC, ABL, 4, true, 2>(a, acc[rb], nfull_pre[rb], row0, static_cast<uint32_t>(wr * (32 * RB) + rb * 32),
                                                         q0 + wc * 128, sel, h, l31, tau_pre, ppass[rb], pbase[rb]);
    } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const uint32_t rit = static_cast<uint32_t>(wr * (32 * RB) + rb * 32);
            bf16_epilogue<MODE, METRIC, ABL, 4, true>(a, acc[rb], nfull_pre[rb], row0, rit, q0 + wc * 128, sel, h, l31,
                                                      MODE == MODE_FILTER ? tau_pre : nullptr);
        }
    }
}

// =================================================================================================
// Persistent form of the shadow kernel (FILTER mode, 8 waves).
//
// One workgroup per CU walks a list of (row tile, query tile) visits; the k-slab pipeline runs
// straight through the visit boundaries: while the last four slabs of a visit are multiplied, the
// first slabs of the next visit are already being staged, and its first fragments are read before
// the epilogue of the current one.  What this removes (measured on the bench shard: a dim sweep at
// constant work): one workgroup dispatch + one cold DMA prologue per tile, 3-5 us of
// the ~23 us a tile takes.  Visit order = the block order of the non-persistent kernel (block b on
// XCD b % 8, consecutive blocks of an XCD walk the query tiles of one row tile), so the L2 sharing
// of a row tile between its query tiles is kept.
// Precondition: dim % 32 == 0 and dim >= 256 (at least 8 slabs per visit).
// =================================================================================================
// Cosine only: with the unit-normalised shadow the epilogue needs the row norms just to flag rows
// whose norm is out of range — and those have an all-zero shadow row, i.e. scores of exactly 0.0
// against every query.  So nothing is preloaded per visit: a lane that sees 0.0 in all four of its
// query blocks for some row raises a (wave-uniform, practically never taken) slow path that loads
// the norms.  Thresholds are loaded once while the query tile stays the same.
__global__ __launch_bounds__(BT_THREADS, 2) void scan_tiles_bf16p_kernel(ScanArgs a) {
    constexpr int METRIC = YAMS_SCAN_COSINE;
    constexpr int RB = 2, PA = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[SH_NST * SH_STAGE];

    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t per_xcd = gridDim.x >> 3;              // gridDim.x is a multiple of 8
    const uint32_t total_w = ((a.n_sel_tiles + 7u) / 8u) * a.n_qtiles;
    uint32_t w = blockIdx.x >> 3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const uint32_t dim = a.dim;
    const int nslab = dim / SH_K;
    const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;

    // The current and the next visit, as plain scalars (uniform: bases, row0, q0; per lane: the two
    // row-piece offsets) — a struct handed around by pointer ends up in scratch memory.
    const unsigned char *cA = nullptr, *cB = nullptr, *nA = nullptr, *nB = nullptr;
    uint64_t crow0 = 0, nrow0 = 0;
    uint32_t cq0 = 0, nq0 = 0;
    uint32_t cvo0 = 0, cvo1 = 0, nvo0 = 0, nvo1 = 0;
    const int rowA0 = (wid * PA + 0) * 16 + (lane >> 2), rowA1 = (wid * PA + 1) * 16 + (lane >> 2);
    const uint32_t chk0 = ((lane & 3) ^ ((rowA0 >> 2) & 3)) * 16u, chk1 = ((lane & 3) ^ ((rowA1 >> 2) & 3)) * 16u;
    // visit ww of this XCD -> tile coordinates and DMA bases of the NEXT slot
    auto open_next = [&](uint32_t ww) __attribute__((always_inline)) -> bool {
        if (ww >= total_w) return false;
        const uint32_t qt = ww % a.n_qtiles;
        const uint32_t sel = (ww / a.n_qtiles) * 8u + xcd;
        if (sel >= a.n_sel_tiles) return false;           // only in the last group; nothing valid follows
        const uint32_t tile = sel + sel / (a.stride - 1u) + 1u;
        nrow0 = static_cast<uint64_t>(tile) * BT_ROWS;
        nq0 = qt * BT_QUERIES;
        nA = reinterpret_cast<const unsigned char*>(a.rows_bf16 + nrow0 * dim);
        nB = reinterpret_cast<const unsigned char*>(a.q_hi + static_cast<uint64_t>(nq0) * 32);
        uint64_t r0 = nrow0 + rowA0, r1 = nrow0 + rowA1;
        if (r0 >= a.n_rows) r0 = a.n_rows - 1;
        if (r1 >= a.n_rows) r1 = a.n_rows - 1;
        nvo0 = static_cast<uint32_t>(r0 - nrow0) * dim * 2u + chk0;
        nvo1 = static_cast<uint32_t>(r1 - nrow0) * dim * 2u + chk1;
        return true;
    };
    auto advance = [&]() __attribute__((always_inline)) { cA = nA; cB = nB; crow0 = nrow0; cq0 = nq0; cvo0 = nvo0; cvo1 = nvo1; };
    const uint32_t voffB0 = static_cast<uint32_t>(rowA0) * 64u + chk0, voffB1 = static_cast<uint32_t>(rowA1) * 64u + chk1;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    const uint32_t ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * PA * 1024);
    const uint32_t ldsB = __builtin_amdgcn_readfirstlane(lds0 + SH_A_BYTES + wid * PA * 1024);
    // DMA piece p (0,1: rows; 2,3: queries) of slab `slab` of the current visit, or — when the slab
    // index runs past the visit (slab >= nslab) — of the next one; ring stage `stg`.  The choice is a
    // scalar select, so the slab loop below stays one basic block from the first slab of a visit to
    // its last.
    auto piece = [&](int slab, int stg, int p) __attribute__((always_inline)) {
        const bool cross = slab >= nslab;
        const int sl = cross ? slab - nslab : slab;
        const uint32_t st = static_cast<uint32_t>(stg) * SH_STAGE;
        const unsigned char* bA = cross ? nA : cA;
        const unsigned char* bB = cross ? nB : cB;
        if (p == 0) lds_dma16_s(bA + sl * (SH_K * 2), cross ? nvo0 : cvo0, ldsA + st);
        else if (p == 1) lds_dma16_s(bA + sl * (SH_K * 2), cross ? nvo1 : cvo1, ldsA + st + 1024);
        else if (p == 2) lds_dma16_s(bB + sl * qslab_bytes, voffB0, ldsB + st);
        else lds_dma16_s(bB + sl * qslab_bytes, voffB1, ldsB + st + 1024);
    };

    f32x16 acc[RB][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][u][r] = 0.f;
    };
    zero_acc();

    int offA[RB][2], offB[4][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int rf = wr * 64 + rb * 32 + l31;
        const int f = (rf >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) offA[rb][t] = rf * 64 + (((2 * t + h) ^ f) << 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rq = wc * 128 + u * 32 + l31;
        const int f = (rq >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) offB[u][t] = SH_A_BYTES + rq * 64 + (((2 * t + h) ^ f) << 4);
    }
    struct Frags { bf16x8 a[RB]; bf16x8 b[4]; };
    auto pin = [&](Frags& fr) __attribute__((always_inline)) {
        asm volatile("" : "+v"(fr.a[0]), "+v"(fr.a[1]), "+v"(fr.b[0]), "+v"(fr.b[1]), "+v"(fr.b[2]), "+v"(fr.b[3]));
    };
    // one 16-wide step (see scan_tiles_bf16s_kernel): 8 MFMAs on `cur`, the 6 fragment reads of the
    // next step and 2 DMA pieces (p0, p0+1 of slab dslab into stage dstg) in the gaps
    auto step = [&](const Frags& cur, Frags& nxt, const unsigned char* nbase, int nt, int dslab, int dstg,
                    int p0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rb = i >> 2, u = i & 3;
            acc[rb][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[rb], cur.b[u], acc[rb][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < RB) nxt.a[i < RB ? i : 0] = *reinterpret_cast<const bf16x8*>(nbase + offA[i < RB ? i : 0][nt]);
            if (i >= RB && i < RB + 4) nxt.b[i - RB] = *reinterpret_cast<const bf16x8*>(nbase + offB[i - RB][nt]);
            if (i == 3 || i == 6) piece(dslab, dstg, p0 + (i == 6 ? 1 : 0));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (!open_next(w)) return;
    advance();
    // without a next visit the "next" slot points at the current one: the tail of the slab loop then
    // re-stages this visit's first slabs (never read) instead of branching around the DMA
    auto alias_next = [&]() __attribute__((always_inline)) { nA = cA; nB = cB; nvo0 = cvo0; nvo1 = cvo1; };
    bool has_next = open_next(w + per_xcd);
    if (!has_next) alias_next();
    float tau[4];
    uint32_t tau_q0 = cq0;
    auto load_tau = [&](uint32_t q0v) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t qi = q0v + wc * 128 + u * 32 + l31;
            tau[u] = qi < a.n_queries ? a.tau[qi] : __builtin_inff();
        }
    };
    load_tau(tau_q0); // older than every DMA piece below: retires first, the counted waits stay valid

    // ---- prologue (first visit only): slabs 0..2 and the first half of slab 3 -------------------
    for (int sl = 0; sl < 3; ++sl)
        for (int p = 0; p < 4; ++p) piece(sl, sl, p);
    piece(3, 3, 0); piece(3, 3, 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) f0.a[rb] = *reinterpret_cast<const bf16x8*>(lds + offA[rb][0]);
#pragma unroll
    for (int u = 0; u < 4; ++u) f0.b[u] = *reinterpret_cast<const bf16x8*>(lds + offB[u][0]);
    int stage = 0; // ring stage of the slab being multiplied

    for (;;) {
        // one slab per iteration: [second step requested | first step multiplied | barrier for the
        // next slab | next slab's first step requested | second step multiplied]; DMA: second half
        // of slab s+3, first half of slab s+4 (ring stages stage+3 and stage+4 = stage).  Slabs past
        // the end of the visit are the first slabs of the next one, so the pipeline — DMA, the
        // barrier, the fragment prefetch — runs straight through the visit boundary.
        for (int s = 0; s < nslab; ++s) {
            const unsigned char* base = lds + stage * SH_STAGE;
            const int stg3 = (stage + 3) & (SH_NST - 1), stg4 = stage;
            stage = (stage + 1) & (SH_NST - 1);
            pin(f0);
            __builtin_amdgcn_sched_barrier(0);
            step(f0, f1, base, 1, s + 3, stg3, 2);
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            pin(f1);
            __builtin_amdgcn_sched_barrier(0);
            step(f1, f0, lds + stage * SH_STAGE, 0, s + 4, stg4, 0);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const uint32_t rit = static_cast<uint32_t>(wr * 64 + rb * 32);
            bool suspicious = false;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                suspicious |= (acc[rb][0][r] == 0.f) & (acc[rb][1][r] == 0.f) & (acc[rb][2][r] == 0.f) & (acc[rb][3][r] == 0.f);
            float nfull = 1.f;
            if (__builtin_amdgcn_ballot_w64(suspicious) != 0 || crow0 + BT_ROWS > a.n_rows) {
                const uint64_t r = crow0 + rit + l31;
                nfull = r < a.n_rows ? a.rows_nsq[r] : 1.f;
            }
            bf16_epilogue<MODE_FILTER, METRIC, 0, 4, true>(a, acc[rb], nfull, crow0, rit, cq0 + wc * 128, 0u, h,
                                                           l31, tau);
        }
        // stores / atomics of the epilogue may retire out of order with loads: drain, so that the
        // counted waits of the next visit only ever see DMA pieces (by now the pieces issued during
        // the last four slabs have landed or are about to); also the exit condition of the kernel
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!has_next) break;
        zero_acc();
        advance();
        if (cq0 != tau_q0) { // query tile changed (n_qtiles does not divide the per-XCD stride)
            tau_q0 = cq0;
            load_tau(tau_q0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        w += per_xcd;
        has_next = open_next(w + per_xcd);
        if (!has_next) alias_next();
    }
}

// =================================================================================================
// Narrow form of the shadow kernel for batches of at most 32 NQB queries (64 or 128).
//
// With one query tile the pass streams the shadow once and the 256-query tile of the kernels above
// multiplies mostly padding: a 12.5M x 768 shard took 4.3 ms whether 1 or 256 queries rode along
// (4.5 TB/s).  Here a workgroup still owns 256 rows, but only 32 NQB query columns: a wave
// multiplies ITS OWN 32 rows (staged by itself, two 1 KiB DMA pieces per 32-wide k-slab — no other
// wave reads them) against the query slab the workgroup shares (2 NQB pieces, one per wave for the
// first 2 NQB waves).  4 NQB MFMAs per wave per slab leave the matrix pipe mostly idle.
// (The non-temporal policy on the row pieces, `global_load_lds_dwordx4 ... nt`, was measured:
// 4.5 ms instead of 3.1 — the default policy stays.)
// What matters is bytes in flight: a 3-stage ring of 20 / 24 KiB per workgroup, TWO workgroups per CU
// (<= 128 VGPRs), i.e. 4 row slabs = 64 KiB of HBM reads in flight per CU at all times, and one
// workgroup's prologue / epilogue overlaps the other's stream.  Measured on a 12.5M x 768 shard:
// 3.10 ms (6.1 TB/s, 0.76 of the HBM peak) up to 64 queries, 3.39 ms at 128, against 4.1-4.5 ms of
// the 256-query form.  The same structure stretched to a whole 256-query tile (NQB = 8, one
// workgroup per CU, 4-stage ring) was measured too and loses to the interleaved kernel above
// (5.4 vs 4.7 ms): from 129 queries on the matrix pipe matters again.
// Precondition: dim % 32 == 0, n_queries <= 32 NQB, shadow present.
// =================================================================================================
template <int MODE, int METRIC, int NQB>
__global__ __launch_bounds__(BT_THREADS, 4) void scan_tiles_bf16n_kernel(ScanArgs a) {
    static_assert(NQB == 2 || NQB == 4, "64 or 128 queries");
    constexpr int NST = 3;                         // ring stages
    constexpr int Q_BYTES = NQB * 32 * SH_K * 2;   // query part of a stage (2 KiB per 32 queries)
    constexpr int STAGE = SH_A_BYTES + Q_BYTES;
    constexpr int QPIECES = Q_BYTES / 1024;        // 4 or 8 query pieces per slab
    constexpr int QPW = QPIECES >= 8 ? QPIECES / 8 : 1; // ... per staging wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];

    const uint32_t sel = blockIdx.x;
    const uint32_t tile = MODE == MODE_SAMPLE ? sel * a.stride : sel + sel / (a.stride - 1u) + 1u;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const uint64_t row0 = static_cast<uint64_t>(tile) * BT_ROWS;
    const uint32_t dim = a.dim;
    const int nslab = dim / SH_K;

    uint32_t voffA[2], voffQ[QPW];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rowA = wid * 32 + i * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((rowA >> 2) & 3);
        uint64_t r = row0 + rowA;
        if (r >= a.n_rows) r = a.n_rows - 1;
        voffA[i] = static_cast<uint32_t>(r - row0) * dim * 2u + c * 16u;
    }
    const int qpiece0 = QPIECES >= 8 ? wid * QPW : (wid & (QPIECES - 1));
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        const int rowQ = (qpiece0 + j) * 16 + (lane >> 2);
        voffQ[j] = static_cast<uint32_t>(rowQ) * 64u + (((lane & 3) ^ ((rowQ >> 2) & 3)) * 16u);
    }
    const unsigned char* baseA = reinterpret_cast<const unsigned char*>(a.rows_bf16 + row0 * dim);
    const unsigned char* baseQ = reinterpret_cast<const unsigned char*>(a.q_hi);
    const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    const uint32_t ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * 2048);
    const uint32_t ldsQ = __builtin_amdgcn_readfirstlane(lds0 + SH_A_BYTES + qpiece0 * 1024);

    const int rf = wid * 32 + l31;
    // query block u sits 2 KiB further (same swizzle: 32 | 16), an immediate offset of the read
    int offA[2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        offA[t] = rf * 64 + (((2 * t + h) ^ ((rf >> 2) & 3)) << 4);
        offB[t] = SH_A_BYTES + l31 * 64 + (((2 * t + h) ^ ((l31 >> 2) & 3)) << 4);
    }

    f32x16 acc[NQB];
#pragma unroll
    for (int u = 0; u < NQB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    // epilogue inputs first: older than every DMA piece, so the counted waits below stay valid
    float tau_pre[NQB];
    const uint64_t rme = row0 + static_cast<uint32_t>(rf);
    const float nfull = rme < a.n_rows ? a.rows_nsq[rme] : 1.f;
#pragma unroll
    for (int u = 0; u < NQB; ++u) {
        const uint32_t qi = u * 32 + l31;
        tau_pre[u] = (MODE == MODE_FILTER && qi < a.n_queries) ? a.tau[qi] : __builtin_inff();
    }

    // HASQ: this wave also stages QPW pieces of the shared query slab (2 + QPW pieces per slab, else 2)
    auto run = [&](auto hasq_tag) __attribute__((always_inline)) {
        constexpr bool HASQ = decltype(hasq_tag)::value;
        constexpr int PPW = 2 + (HASQ ? QPW : 0);
        auto issue = [&](int sl, int stg) __attribute__((always_inline)) {
            const uint32_t st = static_cast<uint32_t>(stg) * STAGE;
            lds_dma16_s(baseA + sl * (SH_K * 2), voffA[0], ldsA + st);
            lds_dma16_s(baseA + sl * (SH_K * 2), voffA[1], ldsA + st + 1024);
            if (HASQ) {
#pragma unroll
                for (int j = 0; j < QPW; ++j) lds_dma16_s(baseQ + sl * qslab_bytes, voffQ[j], ldsQ + st + j * 1024);
            }
        };
        // s_waitcnt for "at most `newer` slabs younger than the one needed are still in flight"
        auto wait_newer = [&](int newer) __attribute__((always_inline)) {
            const int c = newer * PPW;
            if (c >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (c >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if (c >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else if (c >= 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            else if (c >= 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        };
        for (int sl = 0; sl < NST - 1 && sl < nslab; ++sl) issue(sl, sl);
        int stage = 0;
        for (int sl = 0; sl < nslab; ++sl) {
            // slab sl landed (up to NST-2 younger ones may still be in flight); every wave is past
            // its reads of slab sl-1, whose stage the refill below reuses
            const int left = nslab - 1 - sl;
            wait_newer(left < NST - 2 ? left : NST - 2);
            __builtin_amdgcn_s_barrier();
            const int nstage = stage == 0 ? NST - 1 : stage - 1; // (sl + NST - 1) % NST
            if (sl + NST - 1 < nslab) issue(sl + NST - 1, nstage);
            const unsigned char* base = lds + stage * STAGE;
            stage = stage + 1 == NST ? 0 : stage + 1;
            bf16x8 fa[2], fb[NQB][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const bf16x8*>(base + offA[t]);
#pragma unroll
                for (int u = 0; u < NQB; ++u) fb[u][t] = *reinterpret_cast<const bf16x8*>(base + offB[t] + u * 2048);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < NQB; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t], fb[u][t], acc[u], 0, 0, 0);
        }
    };
    if constexpr (QPIECES >= 8) {
        run(std::integral_constant<bool, true>{});
    } else {
        if (wid < QPIECES) run(std::integral_constant<bool, true>{});
        else run(std::integral_constant<bool, false>{});
    }
    bf16_epilogue<MODE, METRIC, 0, NQB, true>(a, acc, nfull, row0, static_cast<uint32_t>(wid) * 32u, 0u, sel, h, l31,
                                              MODE == MODE_FILTER ? tau_pre : nullptr);
}

// The shadow: bf16 (RNE) of the UNIT-NORMALISED rows + their fp32 squared norms; one wave per row.
// Rows whose squared norm is outside (1e-30, 1e30) or not finite get an all-zero shadow row; the
// filter epilogue turns their scores into NaN (= always a candidate) from the stored norm.
__global__ __launch_bounds__(256) void shadow_build_kernel(const float* rows, uint64_t n_rows, uint32_t dim,
                                                           uint16_t* out_bf16, float* out_nsq) {
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const float* src = rows + row * dim;
    uint16_t* dst = out_bf16 + row * dim;
    float nsq = 0.f;
    typedef __attribute__((ext_vector_type(2))) float f32x2v;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
    for (uint32_t c = lane * 4; c < dim; c += 256) { // dim % 4 == 0 (rows are 16-byte aligned)
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        nsq = fmaf(v.x, v.x, nsq); nsq = fmaf(v.y, v.y, nsq);
        nsq = fmaf(v.z, v.z, nsq); nsq = fmaf(v.w, v.w, nsq);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nsq += __shfl_xor(nsq, d);
    const float inv = norm_in_range(nsq) ? rsqrtf(nsq) : 0.f;
    for (uint32_t c = lane * 4; c < dim; c += 256) { // second read of the row: L1/L2 hit
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        const f32x2v lo = {inv != 0.f ? v.x * inv : 0.f, inv != 0.f ? v.y * inv : 0.f};
        const f32x2v hi = {inv != 0.f ? v.z * inv : 0.f, inv != 0.f ? v.w * inv : 0.f};
        uint2 o;
        o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2v));
        o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2v));
        *reinterpret_cast<uint2*>(dst + c) = o;
    }
    if (lane == 0) out_nsq[row] = nsq;
}

// Split the prepared fp32 queries into bf16 head + tail planes, k-slab-major with slabs of
// `slab` (16 or 32) k-values: plane[(k / slab) * q_pad + q][k % slab]; rows q >= n_queries are zero.
__global__ void prep_split_kernel(const float* qprep, uint32_t nq, uint32_t q_pad, uint32_t dim,
                                  uint32_t slab_k, uint16_t* q_hi, uint16_t* q_lo) {
    const uint64_t total = static_cast<uint64_t>(q_pad) * dim;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t kk = static_cast<uint32_t>(i % slab_k);
        const uint64_t qs = i / slab_k;
        const uint32_t q = static_cast<uint32_t>(qs % q_pad);
        const uint32_t slab = static_cast<uint32_t>(qs / q_pad);
        const float x = q < nq ? qprep[static_cast<uint64_t>(q) * dim + slab * slab_k + kk] : 0.f;
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
                             uint32_t slab_k, uint16_t* q_hi, uint16_t* q_lo) {
    const uint64_t n_elems = static_cast<uint64_t>(q_pad) * dim;
    if (n_elems == 0) return hipSuccess;
    uint32_t grid = static_cast<uint32_t>((n_elems + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(prep_split_kernel, dim3(grid), dim3(256), 0, st, qprep, nq, q_pad, dim, slab_k, q_hi, q_lo);
    LAUNCH_CHECK();
    return hipSuccess;
}

ScanArgs make_scan_args(const ScanLaunch& L); // scan_kernels.hip

hipError_t launch_shadow_build(hipStream_t st, const float* rows, uint64_t n_rows, uint32_t dim,
                               uint16_t* out_bf16, float* out_nsq) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(shadow_build_kernel, dim3(static_cast<uint32_t>((n_rows + 3) / 4)), dim3(256), 0, st,
                       rows, n_rows, dim, out_bf16, out_nsq);
    hipError_t e_ = hipGetLastError();
    return e_;
}

#define LAUNCH_BF16(PASSES) do { \
    if (mode == MODE_SAMPLE) { \
        if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_SAMPLE, YAMS_SCAN_COSINE, PASSES>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
        else hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_SAMPLE, YAMS_SCAN_L2, PASSES>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
    } else { \
        if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_FILTER, YAMS_SCAN_COSINE, PASSES>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
        else hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_FILTER, YAMS_SCAN_L2, PASSES>), dim3(grid), dim3(BT_THREADS), 0, st, a); \
    } } while (0)

// passes: 1 or 3 (see the kernel).  version: 2 = default forms, 3 = wide tile for small batches
// (YAMS_SCAN_FLAG_WIDE_TILE); other values exist in the measurement build only (ablations, -DYAMS_ACCEL_MEASURE).
hipError_t launch_scan_bf16(hipStream_t st, const ScanLaunch& L, int metric, int mode, int passes, int version) {
    ScanArgs a = make_scan_args(L);
    a.n_sel_tiles = mode == MODE_SAMPLE ? L.plan.n_sample_tiles : L.plan.n_filter_tiles;
    if (a.n_sel_tiles == 0) return hipSuccess;
    const uint32_t groups = (a.n_sel_tiles + 7) / 8;
    const uint32_t grid = groups * a.n_qtiles * 8;
#ifdef YAMS_ACCEL_MEASURE
    if (version == 12 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE && bf16_slab_k(passes, L.plan.dim) == 16) {
        if (passes == 3) hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 3, 2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else hipLaunchKernelGGL((scan_tiles_bf16v2_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 1, 2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        LAUNCH_CHECK();
        return hipSuccess;
    }
#else
    if (version != 3) version = 2; // the product library knows two forms: its own choice, or the wide tile
#endif
    if (passes == 3) LAUNCH_BF16(3);
    else if (a.rows_bf16 && bf16_slab_k(passes, L.plan.dim) == 32) {
        // persistent form where tiles are short (many visits per CU): measured -3 % at dim 384,
        // equal at 768, +5 % (worse) at 1536; version 4 forces it, 3 forces the per-tile form
        const bool persistent = (version == 4) || (version == 2 && L.plan.dim <= 512);
        if (mode == MODE_FILTER && persistent && L.plan.dim >= 256 && metric == YAMS_SCAN_COSINE) {
            // persistent form: one workgroup per CU (multiple of 8 so that XCD = block % 8 holds)
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            uint32_t pg = static_cast<uint32_t>(cus) & ~7u;
            if (pg < 8) pg = 8;
            const uint32_t total_w = groups * a.n_qtiles * 8u;
            if (pg > total_w) pg = (total_w + 7u) & ~7u;
            hipLaunchKernelGGL(scan_tiles_bf16p_kernel, dim3(pg), dim3(BT_THREADS), 0, st, a);
            LAUNCH_CHECK();
            return hipSuccess;
        }
        // small batches: the narrow form (HBM-bound; see the kernel).  version 3 keeps the 256-query form.
        // (L2 with 128 queries would need more than the 128 VGPRs two workgroups per CU leave a wave —
        // its epilogue rescales every score by the row norm — so L2 goes narrow up to 64 queries only.)
        // (the sample epilogue of a 128-query block does not fit 128 VGPRs either: the sample pass
        // of 65..128-query batches stays on the 256-query form)
        const uint32_t narrow_max = (metric == YAMS_SCAN_COSINE && mode == MODE_FILTER) ? 128u : 64u;
        if (version == 2 && L.plan.dim >= 64 && a.n_queries <= narrow_max) {
            const dim3 ng(a.n_sel_tiles), nt(BT_THREADS);
            if (mode == MODE_SAMPLE) {
                if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((scan_tiles_bf16n_kernel<MODE_SAMPLE, YAMS_SCAN_COSINE, 2>), ng, nt, 0, st, a);
                else hipLaunchKernelGGL((scan_tiles_bf16n_kernel<MODE_SAMPLE, YAMS_SCAN_L2, 2>), ng, nt, 0, st, a);
            } else if (metric != YAMS_SCAN_COSINE) {
                hipLaunchKernelGGL((scan_tiles_bf16n_kernel<MODE_FILTER, YAMS_SCAN_L2, 2>), ng, nt, 0, st, a);
            } else if (a.n_queries <= 64) {
                hipLaunchKernelGGL((scan_tiles_bf16n_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 2>), ng, nt, 0, st, a);
            } else {
                hipLaunchKernelGGL((scan_tiles_bf16n_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 4>), ng, nt, 0, st, a);
            }
            LAUNCH_CHECK();
            return hipSuccess;
        }
#ifdef YAMS_ACCEL_MEASURE
        const bool abl = mode == MODE_FILTER && metric == YAMS_SCAN_COSINE;
        const bool four = version >= 20;       // 4-wave (4x4 tiles) form; 2x = its ablations
        const int v = four ? version - 10 : version;
#define LAUNCH_SH(MODE_, METRIC_, ABL_) do { \
            if (four) hipLaunchKernelGGL((scan_tiles_bf16s_kernel<MODE_, METRIC_, ABL_, 4>), dim3(grid), dim3(256), 0, st, a); \
            else hipLaunchKernelGGL((scan_tiles_bf16s_kernel<MODE_, METRIC_, ABL_, 8>), dim3(grid), dim3(512), 0, st, a); } while (0)
        if (abl && v == 11) LAUNCH_SH(MODE_FILTER, YAMS_SCAN_COSINE, 1);
        else if (abl && v == 12) LAUNCH_SH(MODE_FILTER, YAMS_SCAN_COSINE, 2);
        else if (abl && v == 13) LAUNCH_SH(MODE_FILTER, YAMS_SCAN_COSINE, 3);
        else if (abl && v == 14) LAUNCH_SH(MODE_FILTER, YAMS_SCAN_COSINE, 4);
        else if (abl && v == 15) LAUNCH_SH(MODE_FILTER, YAMS_SCAN_COSINE, 5);
        else
#else
#define LAUNCH_SH(MODE_, METRIC_, ABL_) \
            hipLaunchKernelGGL((scan_tiles_bf16s_kernel<MODE_, METRIC_, ABL_, 8>), dim3(grid), dim3(512), 0, st, a)
#endif
        if (mode == MODE_SAMPLE) {
            if (metric == YAMS_SCAN_COSINE) LAUNCH_SH(MODE_SAMPLE, YAMS_SCAN_COSINE, 0);
            else LAUNCH_SH(MODE_SAMPLE, YAMS_SCAN_L2, 0);
        } else {
            if (metric == YAMS_SCAN_COSINE) LAUNCH_SH(MODE_FILTER, YAMS_SCAN_COSINE, 0);
            else LAUNCH_SH(MODE_FILTER, YAMS_SCAN_L2, 0);
        }
#undef LAUNCH_SH
    } else if (bf16_slab_k(passes, L.plan.dim) == 32) {
#ifdef YAMS_ACCEL_MEASURE
        if (version == 12 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE)
            hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else if (version == 11 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE)
            hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 1>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else if (version == 13 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE)
            hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 3>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else if (version == 14 && mode == MODE_FILTER && metric == YAMS_SCAN_COSINE)
            hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_FILTER, YAMS_SCAN_COSINE, 4>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        else
#endif
        if (mode == MODE_SAMPLE) {
            if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_SAMPLE, YAMS_SCAN_COSINE>), dim3(grid), dim3(BT_THREADS), 0, st, a);
            else hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_SAMPLE, YAMS_SCAN_L2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        } else {
            if (metric == YAMS_SCAN_COSINE) hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_FILTER, YAMS_SCAN_COSINE>), dim3(grid), dim3(BT_THREADS), 0, st, a);
            else hipLaunchKernelGGL((scan_tiles_bf16k32_kernel<MODE_FILTER, YAMS_SCAN_L2>), dim3(grid), dim3(BT_THREADS), 0, st, a);
        }
    } else LAUNCH_BF16(1);
    LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace yams_accel
