// scan_kernels.hip — gfx950 kernels of the exact vector scan (K1..K4 of SURVEY.md §2).
//
// Reference semantics being reproduced (paths under /root/reference):
//   src/vector/sqlite_vec_backend.cpp:4204-4331  exact cosine scan, fp64 per-row dot / norm,
//                                                bounded top-k with (similarity desc, chunk_id asc)
//   src/vector/sqlite_vec_backend.cpp:4450-4530  vec0 L2 top-k + cosine re-score
//
// Strategy: the fp64 arithmetic of the reference is only needed for rows that can
// reach the top k.  An exact-f32 MFMA contraction (v_mfma_f32_32x32x2_f32) scores every
// (row, query) pair with a rigorously bounded error E, a threshold filter keeps the rows whose
// f32 score is within reach of the top k, and only those survivors are re-scored in fp64 in the
// reference's own summation order.  A per-query verification proves that no discarded row could
// have entered the result; queries that fail it are widened and finally scored exhaustively in
// fp64 — on the device, never on the CPU.
#include <algorithm>

#include "common.h"
#include "scan_args.h"

namespace yams_accel {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int LDP = kSlabK + 4;                                  // padded LDS row stride (floats)
constexpr int STAGE_FLOATS = (kTileRows + kTileQueries) * LDP;   // one LDS stage
constexpr int SCAN_THREADS = 256;


// One workgroup = 128 corpus rows x 128 queries, 4 waves as 2 (rows) x 2 (queries), each wave a
// 64 x 64 block = 2 x 2 MFMA 32x32 tiles.  A operand = corpus rows (accumulator rows, across
// registers), B operand = queries (accumulator columns, across lanes): a lane owns one query per
// B tile, so the per-query threshold test in the epilogue is lane-local.
template <int MODE, int METRIC>
__global__ __launch_bounds__(SCAN_THREADS, 2) void scan_tiles_kernel(ScanArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE_FLOATS];

    // XCD-aware mapping: block b runs on XCD b % 8 (observed dispatch); consecutive blocks of one
    // XCD walk the query tiles of ONE row tile, so the row tile is fetched from HBM once and
    // re-read from that XCD's L2.
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
    const int wr = wid >> 1, wq = wid & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const uint64_t row0 = static_cast<uint64_t>(tile) * kTileRows;
    const uint32_t q0 = qt * kTileQueries;
    const uint32_t dim = a.dim;
    const int nslab = (dim + kSlabK - 1) / kSlabK;

    // ---- staging addresses: thread -> (row lr + 32*j, float4 column c4) -------------------------
    const int lr = tid >> 3, c4 = tid & 7;
    const float* gptr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int R = lr + 32 * j;
        if (R < kTileRows) {
            uint64_t r = row0 + R;
            if (r >= a.n_rows) r = a.n_rows - 1;
            gptr[j] = a.rows + r * dim + c4 * 4;
        } else {
            uint32_t q = q0 + (R - kTileRows);
            if (q >= a.n_queries) q = a.n_queries - 1;
            gptr[j] = a.qprep + static_cast<uint64_t>(q) * dim + c4 * 4;
        }
    }
    float4 stage[8];
    auto load_slab = [&](int s) {
        const int k = s * kSlabK + c4 * 4;
        const bool in = k < static_cast<int>(dim);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            stage[j] = in ? *reinterpret_cast<const float4*>(gptr[j] + s * kSlabK)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_slab = [&](int buf) {
        float* base = lds + buf * STAGE_FLOATS;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<float4*>(base + (lr + 32 * j) * LDP + c4 * 4) = stage[j];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    float nsq[2] = {0.f, 0.f};

    // LDS fragment offsets: lane reads 4 consecutive k of its row; lanes 0-31 take k 0..3 and
    // lanes 32-63 k 4..7 of each 8-wide k group (the same permutation of k on both operands).
    const int a_off = (wr * 64 + l31) * LDP + 4 * h;
    const int b_off = (kTileRows + wq * 64 + l31) * LDP + 4 * h;

    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        if (s + 1 < nslab) load_slab(s + 1);
        const float* base = lds + (s & 1) * STAGE_FLOATS;
#pragma unroll
        for (int kg = 0; kg < kSlabK / 8; ++kg) {
            float4 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                af[t] = *reinterpret_cast<const float4*>(base + a_off + t * 32 * LDP + kg * 8);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                bf[u] = *reinterpret_cast<const float4*>(base + b_off + u * 32 * LDP + kg * 8);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                nsq[t] = fmaf(af[t].x, af[t].x, nsq[t]);
                nsq[t] = fmaf(af[t].y, af[t].y, nsq[t]);
                nsq[t] = fmaf(af[t].z, af[t].z, nsq[t]);
                nsq[t] = fmaf(af[t].w, af[t].w, nsq[t]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].x, bf[u].x, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].y, bf[u].y, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].z, bf[u].z, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].w, bf[u].w, acc[t][u], 0, 0, 0);
                }
        }
        if (s + 1 < nslab) store_slab((s + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // Accumulator layout (32x32): column j = lane & 31 (query), row i = (r&3) + 8*(r>>2) + 4*h.
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
    for (int t = 0; t < 2; ++t) {
        const float nfull = nsq[t] + __shfl_xor(nsq[t], 32);
        float p0, p1 = 0.f;
        const bool ok = norm_in_range(nfull);
        if (METRIC == YAMS_SCAN_COSINE) {
            p0 = ok ? rsqrtf(nfull) : __builtin_nanf("");
        } else {
            // upper bound of g = q.x - |x|^2/2:  dot + nsq*(-0.5 + 0.5 c) + (c*|x|)*|q|
            p0 = ok ? nfull * (-0.5f + 0.5f * a.err_coef) : __builtin_nanf("");
            p1 = ok ? a.err_coef * sqrtf(nfull) * 1.000001f : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float rp0 = __shfl(p0, i);
            if (METRIC == YAMS_SCAN_COSINE) {
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u][r] = acc[t][u][r] * rp0;
            } else {
                const float rp1 = __shfl(p1, i);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u][r] = acc[t][u][r] + rp0 + rp1 * qn_up[u];
            }
        }
    }

    const uint64_t wave_row0 = row0 + wr * 64;
    if (a.row_mask) { // rows outside the allow-mask never score (they are not part of the scan)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t mw = mask_word(a.row_mask, wave_row0 + t * 32, a.n_rows);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (!((mw >> i) & 1u)) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u][r] = -__builtin_inff();
                }
            }
        }
    }
    if (MODE == MODE_SAMPLE) {
        const float ninf = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float m = ninf;
                bool bad = false; // a NaN score marks the whole group (key 0xffffffff): its rows are
                                  // untrustworthy for the filter and must reach the fp64 re-score
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uint64_t rbase = wave_row0 + t * 32 + 8 * g4 + 4 * h;
                    float4 v;
                    v.x = (rbase + 0 < a.n_rows) ? acc[t][u][4 * g4 + 0] : ninf;
                    v.y = (rbase + 1 < a.n_rows) ? acc[t][u][4 * g4 + 1] : ninf;
                    v.z = (rbase + 2 < a.n_rows) ? acc[t][u][4 * g4 + 2] : ninf;
                    v.w = (rbase + 3 < a.n_rows) ? acc[t][u][4 * g4 + 3] : ninf;
                    m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                    bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
                    if (qok[u]) {
                        const uint64_t srow = static_cast<uint64_t>(sel) * kTileRows + wr * 64 +
                                              t * 32 + 8 * g4 + 4 * h;
                        *reinterpret_cast<float4*>(a.dense + dense_index(qidx[u], srow, a.n_queries)) = v;
                    }
                }
                if (qok[u]) {
                    const uint32_t gid = ((sel * 4u + wr * 2u + t) << 1) + h;
                    a.gmax[static_cast<uint64_t>(qidx[u]) * a.n_groups + gid] = bad ? 0xffffffffu : f2ord(m);
                }
            }
    } else {
        float tau[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) tau[u] = qok[u] ? a.tau[qidx[u]] : __builtin_inff();
        bool any = false;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) any |= !(acc[t][u][r] < tau[u]);
        if (any) {
            // one list reservation per (lane, query block) for all 32 rows of the two row blocks,
            // both reservations issued before the first is needed (see bf16_epilogue)
            uint32_t pass[2], base[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pass[u] = 0u;
                if (qok[u]) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint64_t row = wave_row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (!(acc[t][u][r] < tau[u]) && row < a.n_rows) pass[u] |= 1u << (t * 16 + r);
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                base[u] = 0u;
                if (pass[u]) base[u] = atomicAdd(&a.list_count[qidx[u]], static_cast<uint32_t>(__builtin_popcount(pass[u])));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!pass[u]) continue;
                uint32_t pos = base[u];
                uint64_t* lst = a.list + static_cast<uint64_t>(qidx[u]) * a.list_cap;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (!((pass[u] >> (t * 16 + r)) & 1u)) continue;
                        const uint64_t row = wave_row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (pos < a.list_cap) lst[pos] = pack_key(acc[t][u][r], static_cast<uint32_t>(row));
                        ++pos;
                    }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// prep_queries: validity + fp64 norm in the reference's summation order (:4206-4211) + the
// operand the MFMA pass consumes (unit-norm fp32 for cosine, raw for L2).
//   flags: bit0 = non-finite element, bit1 = norm^2 < 1e-10 (isZeroNormEmbedding, :204-211)
// -------------------------------------------------------------------------------------------------
__global__ void prep_queries_kernel(const float* q, uint32_t nq, uint32_t dim, int metric,
                                    float* qprep, double* qnorm, float* qnorm_up,
                                    uint32_t* qflags, int staged, uint32_t* zero_words, uint32_t n_zero_words) {
    const uint32_t qi = blockIdx.x;
    // the batch's state words (counters, status, list counts: scan_api.cpp "qstate") start at zero
    for (uint32_t i = qi * blockDim.x + threadIdx.x; i < n_zero_words; i += gridDim.x * blockDim.x) zero_words[i] = 0u;
    __shared__ double s_norm;
    __shared__ uint32_t s_flag;
    extern __shared__ float s_stage[]; // dim floats when the launch provides them (staged != 0)
    const float* src = q + static_cast<uint64_t>(qi) * dim;
    if (staged) {
        // the norm below is ONE sequential fp64 chain (reference order): feed it from LDS, not from
        // dependent global loads (40 us -> a few us for a lone 768-wide query)
        for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) s_stage[i] = src[i];
        __syncthreads();
        src = s_stage;
    }
    if (threadIdx.x == 0) {
        double acc = 0.0;
        // one sequential chain in the reference's order; fp64 cannot overflow on fp32 squares, so
        // "every element finite" <=> "the sum is finite" (inf*inf = inf, NaN propagates)
        uint32_t i = 0;
        for (; i + 8 <= dim; i += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[i + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double d = static_cast<double>(v[e]);
                acc = fma(d, d, acc); // v*v is exact in fp64, so fma == mul then add
            }
        }
        for (; i < dim; ++i) {
            const double d = static_cast<double>(src[i]);
            acc = fma(d, d, acc);
        }
        const bool finite = isfinite(acc);
        uint32_t f = 0;
        if (!finite) f |= 1u;
        if (!(acc >= 1e-10)) f |= 2u;
        s_flag = f;
        s_norm = sqrt(acc);
        qflags[qi] = f;
        qnorm[qi] = s_norm;
        if (qnorm_up) {
            float up = static_cast<float>(s_norm);
            if (static_cast<double>(up) < s_norm) up = nextafterf(up, __builtin_inff());
            qnorm_up[qi] = up;
        }
    }
    __syncthreads();
    const double n = s_norm;
    const bool bad = s_flag != 0;
    float* dst = qprep + static_cast<uint64_t>(qi) * dim;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) {
        const float v = src[i];
        float o;
        if (bad) o = 0.f;
        else if (metric == YAMS_SCAN_COSINE) o = static_cast<float>(static_cast<double>(v) / n);
        else o = v;
        dst[i] = o;
    }
}

// -------------------------------------------------------------------------------------------------
// Block top-k by bitonic sort in LDS.  Grid (n_chunks, n_queries).  Sorts one chunk of up to CAP
// keys of one query in descending order and writes its best `keep` keys (0-padded).
// -------------------------------------------------------------------------------------------------
// Sorts s[0, m) in descending order; m is a power of two <= CAP (the same for the whole block).
// One compare-exchange per thread per step: pair t is (i, i + j) with i = 2t - (t & (j - 1)).
template <typename K, int CAP, int NT>
__device__ __forceinline__ void bitonic_desc(K* s, int m = CAP) {
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += NT) {
                const int i = 2 * t - (t & (j - 1));
                const int ixj = i + j;
                const K x = s[i], y = s[ixj];
                const bool up = (i & k) == 0; // descending overall
                if (up ? (x < y) : (x > y)) { s[i] = y; s[ixj] = x; }
            }
            __syncthreads();
        }
    }
}

// The same network with the keys in REGISTERS (blocked layout: thread t holds elements [t E, t E + E) of M = 256 E): a
// compare-exchange at distance j < E stays inside a thread, one at E <= j < 64 E is a lane exchange inside a wave
// (ds_bpermute, no barrier), and only j >= 64 E — three passes of the 55 of a 1024-key sort, three of the 78 of a 4096-key
// one — goes through LDS and a workgroup barrier.  The LDS form above paid a barrier and an LDS round trip for EVERY pass
// with one workgroup of four waves on a CU: 47 us for the 256 lists of a BASELINE config 2 batch (~1000 keys each), 0.11 ms
// of every headline step.  Element e keeps the larger key of (e, e ^ j) iff ((e & j) == 0) == ((e & k) == 0).
template <typename K, int E, int J>
__device__ __forceinline__ void bitonic_regs_inthread(K (&v)[E], int base, int k) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (i & J) continue;
        const K a = v[i], b = v[i | J];
        const bool up = ((base + i) & k) == 0;
        const K hi = a > b ? a : b, lo = a > b ? b : a;
        v[i] = up ? hi : lo;
        v[i | J] = up ? lo : hi;
    }
}
template <typename K>
__device__ __forceinline__ K lane_xor(K x, int m) {
    if constexpr (sizeof(K) == 8) {
        const uint32_t lo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(static_cast<uint32_t>(x)), m));
        const uint32_t hi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(static_cast<uint32_t>(static_cast<uint64_t>(x) >> 32)), m));
        return static_cast<K>((static_cast<uint64_t>(hi) << 32) | lo);
    } else {
        return static_cast<K>(__shfl_xor(static_cast<int>(x), m));
    }
}
// v: this thread's E keys; xch: LDS, 256 E keys (slot-major: [i][thread], conflict-free for the exchange)
template <typename K, int E>
__device__ __forceinline__ void bitonic_desc_regs(K (&v)[E], K* xch) {
    constexpr int M = 256 * E;
    const int t = threadIdx.x, base = t * E;
#pragma unroll 1
    for (int k = 2; k <= M; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * E) {          // another wave's keys: through LDS
                const int pt = t ^ (j / E);
#pragma unroll
                for (int i = 0; i < E; ++i) xch[i * 256 + t] = v[i];
                __syncthreads();
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const K y = xch[i * 256 + pt];
                    const int e = base + i;
                    const bool keep_max = ((e & j) == 0) == ((e & k) == 0);
                    v[i] = (keep_max == (v[i] > y)) ? v[i] : y;
                }
                __syncthreads();
            } else if (j >= E) {        // another lane of this wave
                const int lm = j / E;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const K y = lane_xor<K>(v[i], lm);
                    const int e = base + i;
                    const bool keep_max = ((e & j) == 0) == ((e & k) == 0);
                    v[i] = (keep_max == (v[i] > y)) ? v[i] : y;
                }
            } else {                    // this thread's own keys (j a compile-time constant per branch: no dynamic register index)
                if (E > 1 && j == 1) bitonic_regs_inthread<K, E, 1>(v, base, k);
                else if (E > 2 && j == 2) bitonic_regs_inthread<K, E, 2>(v, base, k);
                else if (E > 4 && j == 4) bitonic_regs_inthread<K, E, 4>(v, base, k);
                else if (E > 8 && j == 8) bitonic_regs_inthread<K, E, 8>(v, base, k);
            }
        }
    }
}
template <typename K, int E>
__device__ __forceinline__ void topk_block_sorted(const K* src, uint64_t c0, uint64_t n, K* dst, uint32_t keep, K* xch) {
    K v[E];
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < E; ++i) {       // (which key starts where does not matter to a sort: coalesced loads)
        const uint64_t e = c0 + static_cast<uint64_t>(i) * 256 + t;
        v[i] = (e < n) ? src[e] : K(0);
    }
    bitonic_desc_regs<K, E>(v, xch);
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const uint32_t e = static_cast<uint32_t>(t * E + i);
        if (e < keep) dst[e] = v[i];
    }
    for (uint32_t i = 256u * E + t; i < keep; i += 256) dst[i] = K(0); // (keep beyond what was sorted: 0-padded)
}

template <typename K, int CAP>
__global__ __launch_bounds__(256) void topk_block_kernel(const K* in, const uint32_t* in_counts,
                                                         uint64_t in_stride, uint32_t n_fixed,
                                                         uint32_t count_clip, K* out,
                                                         uint64_t out_stride, uint32_t keep,
                                                         const uint32_t* qmap) {
    static_assert(CAP == 4096, "the register forms below cover 256 .. 4096 keys");
    __shared__ K s[CAP];
    const uint32_t qslot = blockIdx.y;
    const uint32_t q = qmap ? qmap[qslot] : qslot;
    uint32_t n = in_counts ? in_counts[q] : n_fixed;
    if (n > count_clip) n = count_clip;
    const uint64_t c0 = static_cast<uint64_t>(blockIdx.x) * CAP;
    const K* src = in + static_cast<uint64_t>(q) * in_stride;
    K* dst = out + static_cast<uint64_t>(qslot) * out_stride + static_cast<uint64_t>(blockIdx.x) * keep;
    if (c0 >= n) { // empty chunk: nothing to sort
        for (uint32_t i = threadIdx.x; i < keep; i += 256) dst[i] = K(0);
        return;
    }
    // only the occupied part of the chunk is sorted (keys are non-zero, the 0 padding sorts last)
    const uint64_t left = n - c0;
    const uint64_t end = left < static_cast<uint64_t>(CAP) ? n : c0 + CAP;
    if (left <= 256) topk_block_sorted<K, 1>(src, c0, end, dst, keep, s);
    else if (left <= 512) topk_block_sorted<K, 2>(src, c0, end, dst, keep, s);
    else if (left <= 1024) topk_block_sorted<K, 4>(src, c0, end, dst, keep, s);
    else if (left <= 2048) topk_block_sorted<K, 8>(src, c0, end, dst, keep, s);
    else topk_block_sorted<K, 16>(src, c0, end, dst, keep, s);
}

// tau[q] = score of the rank-th best group maximum (or -inf when there are fewer groups).  One workgroup per query.
//
// Fast path (rank <= 256, the plans' 16 .. 64): every thread takes the maximum of its strided share of the keys; the
// rank-th largest of those 256 maxima, L, is a LOWER bound of the answer (rank different threads hold a key >= L), so
// the answer is the rank-th largest of the keys >= L — a few dozen, collected into LDS (only threads whose own maximum
// reaches L can hold one) and ranked by counting.  Two passes over the keys, no histogram: the radix select below
// spent its time on LDS atomics that all hit the two or three bins the top 11 bits of similar scores fall into
// (72-87 us per 1024-query batch of the bench, 12 207 groups per query; this form: see DESIGN 8).
// General path: radix select over the order-preserving keys, three histogram passes (11 + 11 + 10 bits).
constexpr int TAU_LIST = 2048;
//
// Proof-aware threshold (round 6, int8 tier under cosine: rows_meta != nullptr).  The rank rule sizes the LIST (about
// rank x stride rows); whether the proof then succeeds depends on how the scores crowd around the k-th best: the filter's
// score is an upper bound u with  u - 2 E <= exact <= u,  E = e_b c_q + f_q,  so the k best exact scores are certainly found
// — and certainly proven — once everything with  u >= u_k - 2 E  is listed (u_k: the k-th best bound).  On a corpus of tight
// clusters (scores of a whole cluster within E of each other) the rank rule's threshold sits INSIDE the cluster and every
// query failed its proof, was widened, filtered again and escalated (29 ms per batch of the bench shard instead of 8).
// The rank2-th best sample maximum minus 2 E is taken instead of the rank-th when (1) the sample says the scores ARE crowded —
// the (rank2 - 4)-th best sample maximum, which stands for u_k, lies less than E above the rank rule's threshold: the actual
// error is a small fraction of E, so exact ~ u - E and a list that ends E below u_k proves itself — and (2) the deeper list
// still fits (at most max_groups sample groups reach the new threshold).  Otherwise the rank rule stands (uniform data: not
// crowded, or the crowd is far too large).  A heuristic in front of a proof: it changes list sizes, never results.
__global__ __launch_bounds__(256) void tau_select_kernel(const uint32_t* gmax, uint32_t n_groups,
                                                         uint32_t rank, float* tau, const float* rows_meta, uint64_t n_blocks,
                                                         const float* q_meta, uint32_t rank2, uint32_t max_groups, float e_scale) {
    __shared__ uint32_t hist[2048]; // fast path: [0, 256) the thread maxima, then the collected keys
    __shared__ uint32_t s_prefix, s_rank, s_count, s_v1, s_v2, s_va;
    __shared__ float s_red[4];
    const uint32_t q = blockIdx.x;
    const uint32_t* keys = gmax + static_cast<uint64_t>(q) * n_groups;
    if (n_groups < rank) { // fewer groups than the rank: no threshold
        if (threadIdx.x == 0) tau[q] = -__builtin_inff();
        return;
    }
    if (rank >= 1 && rank <= 256) {
        // (issued in front of the key passes, used behind them)
        const float e_mine = rows_meta != nullptr && n_blocks ? rows_meta[2 * ((n_blocks - 1) * threadIdx.x / 255) + 1] : 0.f;
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < n_groups; i += 256) { const uint32_t k = keys[i]; mine = k > mine ? k : mine; }
        hist[threadIdx.x] = mine;
        if (threadIdx.x == 0) { s_count = 0; s_v1 = 0; s_v2 = 0; s_va = 0; }
        __syncthreads();
        // L: a maximum with fewer than `rank` maxima strictly above it and at least `rank` at or above it
        uint32_t above = 0, at_or_above = 0;
        for (int j = 0; j < 256; ++j) { const uint32_t o = hist[j]; above += o > mine; at_or_above += o >= mine; }
        if (above < rank && rank <= at_or_above) s_prefix = mine; // (every such thread writes the same value)
        __syncthreads();
        const uint32_t L = s_prefix;
        __syncthreads(); // (hist is reused below)
        bool overflow = false;
        if (mine >= L)
            for (uint32_t i = threadIdx.x; i < n_groups; i += 256) {
                const uint32_t k = keys[i];
                if (k >= L) {
                    const uint32_t pos = atomicAdd(&s_count, 1u);
                    if (pos < TAU_LIST) hist[pos] = k; else overflow = true;
                }
            }
        if (!__syncthreads_or(overflow)) {
            const uint32_t c = s_count;
            // (a group is the maximum of 16 sample rows: counting groups counts rows only while hits are rare among them)
            if (max_groups > n_groups / 64) max_groups = n_groups / 64;
            const bool aware = rows_meta != nullptr && rank2 >= 5 && rank2 < rank && max_groups >= rank2;
            const uint32_t rank_a = rank2 - 4;
            for (uint32_t i = threadIdx.x; i < c; i += 256) {
                const uint32_t k = hist[i];
                uint32_t gt = 0, ge = 0;
                for (uint32_t j = 0; j < c; ++j) { const uint32_t o = hist[j]; gt += o > k; ge += o >= k; }
                if (gt < rank && rank <= ge) { // (equal keys write the same value)
                    if (aware) s_v1 = k; else tau[q] = k ? ord2f(k) : -__builtin_inff();
                }
                if (aware && gt < rank2 && rank2 <= ge) s_v2 = k;
                if (aware && gt < rank_a && rank_a <= ge) s_va = k;
            }
            if (!aware) return;
            // e: the largest residue among 256 blocks spread over the shadow (a representative, not a bound)
            float e = e_mine;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) e = fmaxf(e, __shfl_xor(e, d));
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = e;
            __syncthreads();
            const uint32_t v1 = s_v1, v2 = s_v2, va = s_va;
            float t_out = v1 ? ord2f(v1) : -__builtin_inff();
            bool want = false;
            float tp = 0.f;
            if (v1 && v2 && va) {
                e = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
                const float4 qm = reinterpret_cast<const float4*>(q_meta)[q]; // {t_q, c_q, f_q, 0}
                const float E = fmaf(e, qm.y, qm.z) * e_scale;   // (e_scale: the largest row norm under L2, 1 under cosine)
                tp = ord2f(v2) - 2.0f * E;
                want = ord2f(va) - t_out < E && tp < t_out;   // crowded, and the rank rule does not already list deeper than the proof needs
            }
            if (want) {
                // cheap refusal first: every thread whose own maximum reaches tp holds at least one such group
                const uint32_t lb = __syncthreads_count(mine != 0 && !(ord2f(mine) < tp));
                uint32_t n_reach = 0xffffffffu;
                if (lb <= max_groups) {
                    uint32_t cnt = 0;
                    for (uint32_t i = threadIdx.x; i < n_groups; i += 256) { const uint32_t k = keys[i]; cnt += k != 0 && !(ord2f(k) < tp); }
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
                    __syncthreads();
                    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = __uint_as_float(cnt);
                    __syncthreads();
                    n_reach = __float_as_uint(s_red[0]) + __float_as_uint(s_red[1]) + __float_as_uint(s_red[2]) + __float_as_uint(s_red[3]);
                }
                if (n_reach <= max_groups) t_out = tp;
            }
            if (threadIdx.x == 0) tau[q] = t_out;
            return;
        }
        __syncthreads(); // more than TAU_LIST keys tie at the top: the general path
    }
    if (threadIdx.x == 0) { s_prefix = 0; s_rank = rank; }
    // pass p examines bits [shift, shift + bits) of the keys whose higher bits equal s_prefix
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int p = 0; p < 3; ++p) {
        const int shift = shifts[p], bins = 1 << widths[p];
        for (int i = threadIdx.x; i < 2048; i += 256) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const int hi_shift = shift + widths[p];
        for (uint32_t i = threadIdx.x; i < n_groups; i += 256) {
            const uint32_t k = keys[i];
            if (hi_shift >= 32 || (k >> hi_shift) == prefix) atomicAdd(&hist[(k >> shift) & (bins - 1)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) { // one wave walks the bins from the top
            const int lane = threadIdx.x;
            const int per = bins / 64; // 32 or 16 bins per lane, lane 0 owns the TOP bins
            uint32_t mine = 0;
            for (int j = 0; j < per; ++j) mine += hist[bins - 1 - (lane * per + j)];
            uint32_t incl = mine; // inclusive prefix over lanes (= keys in this lane's bins or above)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const uint32_t excl = incl - mine;
            const uint32_t r = s_rank;
            if (r > excl && r <= incl) { // the rank-th key falls in one of my bins
                uint32_t above = excl;
                for (int j = 0; j < per; ++j) {
                    const int b = bins - 1 - (lane * per + j);
                    const uint32_t c = hist[b];
                    if (r <= above + c) { s_prefix = (prefix << widths[p]) | static_cast<uint32_t>(b); s_rank = r - above; break; }
                    above += c;
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t key = s_prefix;
        tau[q] = key ? ord2f(key) : -__builtin_inff();
    }
}

// Sample rows that reach the threshold join the candidate lists.  Driven by the group maxima:
// only groups whose maximum reaches tau (a NaN group always does) have their 16 dense scores read.
// Group gid covers rows  32 * (gid >> 1) + 4 * (gid & 1) + {0..3} + 8 * {0..3}  of the sample
// (the 32x32 MFMA accumulator layout of the f32 / bf16 sample kernels; layout16 == 0), or rows
// 64 * (gid >> 2) + 4 * (gid & 3) + {0..3} + 16 * {0..3}  (the int8 tier's 16x16 layout).
__global__ __launch_bounds__(256) void collect_sample_kernel(const float* dense, const uint32_t* gmax,
                                                             uint32_t n_groups, uint32_t n_queries,
                                                             uint64_t sample_rows, uint32_t tile_rows,
                                                             uint32_t stride,
                                                             uint64_t n_rows, const float* tau,
                                                             uint32_t* list_count, uint64_t* list,
                                                             uint32_t list_cap, int layout16) {
    const uint32_t q = blockIdx.y;
    const float t = tau[q];
    const uint32_t* gm = gmax + static_cast<uint64_t>(q) * n_groups;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
        const float m = ord2f(gm[g]); // 0xffffffff decodes to NaN
        if (m < t) continue;
        const uint64_t s0 = layout16 ? static_cast<uint64_t>(g >> 2) * 64 + 4 * (g & 3) : static_cast<uint64_t>(g >> 1) * 32 + 4 * (g & 1);
        const uint32_t gstep = layout16 ? 16u : 8u;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v4 = *reinterpret_cast<const float4*>(dense + dense_index(q, s0 + gstep * g4, n_queries));
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = vv[e];
                if (!(v < t)) {
                    const uint64_t sidx = s0 + gstep * g4 + e;
                    const uint64_t row = (sidx / tile_rows) * stride * tile_rows + (sidx % tile_rows);
                    if (row < n_rows) {
                        const uint32_t pos = atomicAdd(&list_count[q], 1u);
                        if (pos < list_cap)
                            list[static_cast<uint64_t>(q) * list_cap + pos] = pack_key(v, static_cast<uint32_t>(row));
                    }
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// K3: exact fp64 re-score of the candidates, final ordering, verification.
//
// One workgroup per query.  Thread c walks candidate c's row in the reference's order
// (sequential i, fp64 accumulate; v*q and v*v are exact products in fp64 so fma == mul+add),
// applies the reference's skips (:4258-4279), then the block sorts by (similarity desc, rank asc)
// and emits the best k.  `bound_key` carries the best f32 filter score among everything that was
// NOT re-scored; the result is proven complete when even bound + E cannot reach the k-th result.
// -------------------------------------------------------------------------------------------------
// The arithmetic of vec0's L2 distance.  It lives in the ABSENT third_party/sqlite-vec-cpp (parity
// unpinned), so every definition that dependency can plausibly have is served and the host picks
// (YAMS_SCAN_FLAG_L2_ACC_*): ACC = 0 fp64 sequential (this repository's own definition);
// ACC = 1 fp32 sequential (the public sqlite-vec's scalar loop); ACC = 8 / 16 fp32 in that many round-robin lanes summed
// left to right at the end (its AVX / AVX-512 forms) — the CPU checker's restatement of each, bit for bit: separate rounded
// subtract, multiply and add (no contraction), the root rounded once (sqrt in fp64 of an fp32 value, rounded to fp32,
// IS the correctly rounded fp32 root: 53 >= 2 * 24 + 2).  `i` must be a compile-time constant after unrolling.
// FUSED (round 5): the same lanes with the square accumulated by ONE fused multiply-add, p = fma(d, d, p) — what a
// compiler makes of `sum += d * d` (and of _mm256_add_ps(sum, _mm256_mul_ps(d, d))) when the translation unit is built
// with -mfma, which is how the reference builds sqlite-vec-cpp on x86 (src/vector/meson.build:80-88: '-mavx', '-mfma').
// The kernels take the pair as one template number: ACCF = lanes | (fused ? 64 : 0); 0 = fp64.
constexpr int l2_lanes(int accf) { return accf & 63; }
constexpr bool l2_fused(int accf) { return (accf & 64) != 0; }
template <int ACC, bool FUSED = false> struct L2Sum {
    float p[ACC];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int l = 0; l < ACC; ++l) p[l] = 0.f;
    }
    __device__ __forceinline__ void add(float x, float q, int i) {
        const float d = __fsub_rn(x, q);
        p[i & (ACC - 1)] = FUSED ? __fmaf_rn(d, d, p[i & (ACC - 1)]) : __fadd_rn(p[i & (ACC - 1)], __fmul_rn(d, d));
    }
    __device__ __forceinline__ double root() const {
        float s = p[0];
        if (ACC > 1) {
            s = 0.f;
#pragma unroll
            for (int l = 0; l < ACC; ++l) s = __fadd_rn(s, p[l]);
        }
        return static_cast<double>(static_cast<float>(sqrt(static_cast<double>(s))));
    }
};
template <> struct L2Sum<0, false> {
    double s;
    __device__ __forceinline__ void clear() { s = 0.0; }
    __device__ __forceinline__ void add(float x, float q, int) { const double d = static_cast<double>(x) - static_cast<double>(q); s = fma(d, d, s); }
    __device__ __forceinline__ double root() const { return sqrt(s); }
};
// elements [i0, dim) of a row, i0 a multiple of 16: groups of 16 with static lane indices (tails and unaligned rows)
template <int ACC, bool FUSED>
__device__ __forceinline__ void l2_sum_tail(L2Sum<ACC, FUSED>& l2, const float* x, const float* q, uint32_t i0, uint32_t dim) {
    for (uint32_t i = i0; i < dim; i += 16) {
#pragma unroll
        for (int l = 0; l < 16; ++l)
            if (i + l < dim) l2.add(x[i + l], q[i + l], l);
    }
}

struct RescoreArgs {
    const float* rows;
    uint64_t n_rows;
    uint32_t dim;
    const float* queries;       // raw queries [nq][dim]
    const double* qnorm;        // [nq]
    const uint32_t* tie_rank;   // nullable
    const uint32_t* rank_row;   // nullable: candidate keys carry ranks, not rows (exact path)
    int64_t row_base;
    uint32_t stripe_rows, n_stripes, stripe_index; // striped shard: local row -> global id (0 = contiguous)
    const uint64_t* cand;       // [slots][cand_stride] keys sorted best-first (0 = empty)
    uint64_t cand_stride;
    uint32_t n_cand;            // candidates to re-score per query
    const float* tau;           // nullable: list completeness threshold per query
    const uint32_t* list_count; // nullable
    uint32_t list_cap;
    uint32_t all_rows_listed;   // 1: the candidate set is the whole corpus -> always verified
    const uint32_t* qmap;       // nullable: slot -> query index
    uint32_t k;
    float threshold;
    uint32_t flags;
    double err_bound;           // E (cosine) ; for L2 the bound is folded into the filter score
    double l2_acc_slack;        // L2 with fp32 accumulation: relative error of the summed square an outside row may carry
    float* out_scores; int64_t* out_rows; uint32_t* out_counts; float* out_dist;
    uint32_t* out_ranks;
    uint32_t* out_status;       // [nq]: 0 verified, 1 needs widening
    unsigned long long* stat_rescored;
    const uint32_t* q_over;     // nullable: != 0 -> the list of this query is incomplete
};

// local row ordinal -> the id the caller sees (yams_scan_corpus_t: row_base, stripes)
__device__ __forceinline__ int64_t global_row(const RescoreArgs& a, uint32_t row) {
    if (a.stripe_rows == 0) return a.row_base + static_cast<int64_t>(row);
    const uint64_t t = row / a.stripe_rows, w = row % a.stripe_rows;
    return a.row_base + static_cast<int64_t>((t * a.n_stripes + a.stripe_index) * a.stripe_rows + w);
}

constexpr int RS_MAX = 2048; // max candidates per query per launch
constexpr int RS_STAGE_STRIDE = 36; // floats per staged row chunk (32 + 4 pad: b128 reads of 16 lanes hit 16 bank groups)

template <int METRIC, int ACC = 0>
__global__ __launch_bounds__(512) void rescore_select_kernel(RescoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // rs = slots the block sorts = next power of two >= n_cand (384 candidates sort as 512)
    int rs = 64;
    while (rs < static_cast<int>(a.n_cand)) rs <<= 1;
    if (rs > RS_MAX) rs = RS_MAX;
    uint64_t* skey = reinterpret_cast<uint64_t*>(smem);              // [rs]
    uint32_t* sidx = reinterpret_cast<uint32_t*>(skey + rs);         // [rs] candidate slot
    float* saux = reinterpret_cast<float*>(sidx + rs);               // [rs] cosine (L2 mode)
    float* sq = saux + rs;                                           // [dim]
    float* sstage = sq + ((a.dim + 3u) & ~3u);                       // [waves][64 rows][RS_STAGE_STRIDE] (staged walk)

    const uint32_t slot = blockIdx.x;
    const uint32_t q = a.qmap ? a.qmap[slot] : slot;
    const uint32_t dim = a.dim;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x)
        sq[i] = a.queries[static_cast<uint64_t>(q) * dim + i];
    for (int i = threadIdx.x; i < rs; i += blockDim.x) { skey[i] = 0; sidx[i] = 0; saux[i] = 0.f; }
    __syncthreads();
    const double qn = a.qnorm[q];
    const uint64_t* cand = a.cand + static_cast<uint64_t>(slot) * a.cand_stride;

    uint32_t local_rescored = 0;
    // Staged walk (dim % 32 == 0, 16-byte aligned rows): a lane still walks its own row in the
    // reference's summation order, but the bytes arrive line by line: the wave fetches one 128-byte
    // line of each of its 64 candidate rows with 8 coalesced instructions (8 rows x 8 lanes x 16 B
    // each — every line is requested exactly once), parks them in LDS, and every lane then reads its
    // own row's 32 floats back.  The lane-per-row loads of the plain walk fetch each line in eight
    // 16-byte pieces from 64 different lines per instruction: 4-8x the traffic between L1, L2 and HBM.
    const bool staged = (dim & 31u) == 0 && ((reinterpret_cast<uintptr_t>(a.rows) & 15u) == 0);
    const uint32_t n_rounds = (a.n_cand + blockDim.x - 1) / blockDim.x;
    // Early close (round 6; cosine, lists longer than one round — the widened and the deep stage): the candidates come in
    // the order of their bounds, so once k of the rows re-scored so far beat the bound of the NEXT candidate (by the proof's
    // own margin) no later candidate can enter the result and the proof will hold with what has been re-scored: the walk
    // stops there (Gaussian rows, 12.5M x 768: a top-100 needs ~850 of the ~1900 listed candidates).
    __shared__ uint32_t s_beat, s_ncand;
    const bool may_close = n_rounds > 1 && !a.all_rows_listed && !(a.flags & (kRescoreFlagPqRerank | kRescoreFlagNoEarlyClose));
    if (threadIdx.x == 0) { s_beat = 0; s_ncand = a.n_cand; }
    for (uint32_t round = 0; round < n_rounds; ++round) {
        if (may_close && round > 0) {
            const uint32_t n_done = round * blockDim.x;             // candidates [0, n_done) are re-scored, their keys in skey
            __syncthreads();
            const uint64_t nextk = cand[n_done];
            bool stop = nextk == 0;                                  // (the list ends here: nothing left to walk)
            if (!stop) {
                const double ob = static_cast<double>(key_score(nextk));
                uint32_t mine = 0;
                if (METRIC == YAMS_SCAN_COSINE) {
                    const float reach = static_cast<float>(ob + a.err_bound + 1e-12);
                    for (uint32_t c = threadIdx.x; c < n_done; c += blockDim.x) mine += skey[c] != 0 && reach < key_score(skey[c]);
                } else {    // (the verification's own bound: the distance no outside row can undercut)
                    double d2 = qn * qn - 2.0 * ob;
                    if (ACC != 0) d2 = d2 * (1.0 - a.l2_acc_slack) - 1e-30;
                    if (d2 < 0.0) d2 = 0.0;
                    const float dmin = static_cast<float>(sqrt(d2) * (ACC != 0 ? 1.0 - 1.2e-7 : 1.0 - 1e-12));
                    for (uint32_t c = threadIdx.x; c < n_done; c += blockDim.x) mine += skey[c] != 0 && dmin > -key_score(skey[c]);
                }
                if (mine) atomicAdd(&s_beat, mine);
                __syncthreads();
                stop = ob == ob && s_beat >= a.k;
                __syncthreads();
                if (threadIdx.x == 0) s_beat = 0;
            }
            if (stop) {
                if (threadIdx.x == 0) s_ncand = n_done;
                break;
            }
        }
        const uint32_t c = round * blockDim.x + threadIdx.x;
        const uint64_t ck = c < a.n_cand ? cand[c] : 0;
        bool live = ck != 0;
        uint32_t row = live ? (a.rank_row ? a.rank_row[key_idx(ck)] : key_idx(ck)) : 0u;
        // (the product-quantised engine: an indexed row the vectors table no longer holds is skipped, sqlite_vec_backend.cpp:4010-4012)
        if ((a.flags & kRescoreFlagPqRerank) && row >= a.n_rows) { live = false; row = 0u; }
        const float* x = a.rows + static_cast<uint64_t>(row) * dim;
        if (live) ++local_rescored;
        double nsq = 0.0, dot = 0.0;
        L2Sum<l2_lanes(ACC), l2_fused(ACC)> l2; l2.clear();
        const bool vec4 = (dim & 3u) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
        if (staged) {
            const int lane = threadIdx.x & 63;
            float* stg = sstage + (threadIdx.x >> 6) * (64 * RS_STAGE_STRIDE);
            if (__builtin_amdgcn_ballot_w64(live) != 0) { // whole wave idle in a partial round: skip
                // The lines of chunk c+1 are requested before chunk c is summed: a lone query has
                // nothing else to hide the miss latency of its dim/32 dependent chunk rounds behind.
                // (No fence between the staging stores and the reads: DS operations of one wave
                // execute in order, and a release fence would drain the prefetch as well.)
                const float* lrow[8];
                typedef float rs_f4 __attribute__((ext_vector_type(4))); // (HIP's float4 struct keeps the array in scratch)
                rs_f4 nx[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t r = static_cast<uint32_t>(__shfl(static_cast<int>(row), 8 * j + (lane >> 3)));
                    lrow[j] = a.rows + static_cast<uint64_t>(r) * dim + (lane & 7) * 4;
                    nx[j] = *reinterpret_cast<const rs_f4*>(lrow[j]);
                }
                for (uint32_t ch = 0; ch < dim; ch += 32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        *reinterpret_cast<rs_f4*>(stg + (8 * j + (lane >> 3)) * RS_STAGE_STRIDE + (lane & 7) * 4) = nx[j];
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (ch + 32 < dim) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) nx[j] = *reinterpret_cast<const rs_f4*>(lrow[j] + ch + 32);
                    }
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const float4 v = *reinterpret_cast<const float4*>(stg + lane * RS_STAGE_STRIDE + 4 * m);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double sv = static_cast<double>(vv[e]);
                            const double qv = static_cast<double>(sq[ch + 4 * m + e]);
                            nsq = fma(sv, sv, nsq);
                            dot = fma(sv, qv, dot);
                            if (METRIC == YAMS_SCAN_L2) l2.add(vv[e], sq[ch + 4 * m + e], 4 * m + e);
                        }
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier(); // the next chunk overwrites the staging rows
                }
            }
            if (!live) continue;
        } else if (!live) {
            continue;
        } else if (vec4) {
            // A lane walks its own row (the summation order is the reference's, so a row cannot be
            // split across lanes): fetch a whole 128-byte line (8 x float4) before consuming it,
            // otherwise every 16 bytes pay a full memory latency.
            uint32_t i = 0;
            for (; i + 32 <= dim; i += 32) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(x + i + 4 * j);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double sv = static_cast<double>(vv[e]);
                        const double qv = static_cast<double>(sq[i + 4 * j + e]);
                        nsq = fma(sv, sv, nsq);
                        dot = fma(sv, qv, dot);
                        if (METRIC == YAMS_SCAN_L2) l2.add(vv[e], sq[i + 4 * j + e], 4 * j + e);
                    }
                }
            }
            if (METRIC == YAMS_SCAN_L2) l2_sum_tail(l2, x, sq, i, dim); // (i is a multiple of 32 here)
            for (; i < dim; i += 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + i);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double sv = static_cast<double>(vv[e]);
                    const double qv = static_cast<double>(sq[i + e]);
                    nsq = fma(sv, sv, nsq);
                    dot = fma(sv, qv, dot);
                }
            }
        } else {
            for (uint32_t i = 0; i < dim; ++i) {
                const double sv = static_cast<double>(x[i]);
                const double qv = static_cast<double>(sq[i]);
                nsq = fma(sv, sv, nsq);
                dot = fma(sv, qv, dot);
            }
            if (METRIC == YAMS_SCAN_L2) l2_sum_tail(l2, x, sq, 0, dim);
        }
        const uint32_t rank = a.tie_rank ? a.tie_rank[row] : row;
        if (METRIC == YAMS_SCAN_COSINE && (a.flags & kRescoreFlagPqRerank)) {
            // the product-quantised engine's re-rank (:4023-4040): computeCosineSimilarity(query, embedding) — each norm's own
            // square root, 0 when either is zero, NO small-norm rule — then the threshold.  (A row with a non-finite component
            // cannot be stored, vector_database.cpp:1771-1784: it is left out rather than ranked by a NaN.)
            if (!isfinite(nsq)) continue;
            const double na = qn, nb = sqrt(nsq);
            const double cs = (na == 0.0 || nb == 0.0) ? 0.0 : dot / (na * nb);
            const float sim = static_cast<float>(cs);
            if (sim < a.threshold || sim != sim) continue;
            skey[c] = pack_key(sim, rank);
            sidx[c] = c;
        } else if (METRIC == YAMS_SCAN_COSINE) {
            // all elements finite <=> nsq finite (fp64 cannot overflow on fp32 squares)
            // :4258-4269; the record path drops norm^2 < 1e-10 instead (isZeroNormEmbedding, :204-211)
            if (!isfinite(nsq) || ((a.flags & YAMS_SCAN_FLAG_RECORD_PATH) ? nsq < 1e-10 : nsq <= 1e-12)) continue;
            const double denom = sqrt(nsq) * qn;                    // :4271
            const double sd = denom > 0.0 ? dot / denom : 0.0;
            if (!isfinite(sd)) continue;                            // :4273-4275
            const float sim = static_cast<float>(sd);               // :4276
#ifdef YAMS_ACCEL_MEASURE
            // bound honesty (YAMS_ACCEL_DUMP_NEEDED prints the counts): the filter's score of a re-scored candidate against its exact
            // similarity — |score - cos| <= err_bound on the bf16 / f32 tiers, cos <= score on the int8 tier (err_bound 0)
            if (a.stat_rescored && !a.all_rows_listed && !a.rank_row) {
                const double fs = static_cast<double>(key_score(ck));
                if (a.err_bound > 0.0 ? fabs(sd - fs) > a.err_bound + 1e-7 : sd > fs + 1e-7) {
                    const unsigned long long nth = atomicAdd(a.stat_rescored + 4, 1ull);
                    if (nth < 6 && a.stat_rescored[6] == 0x5eed) printf("  honesty: query %u cand %u row %u exact %.7f filter %.7f bound %.5f nsq %.6g\n", q, c, row, sd, fs, a.err_bound, nsq);
                }
                atomicAdd(a.stat_rescored + 5, 1ull);
            }
#endif
            if (sim < a.threshold) continue;                        // :4277-4279
            skey[c] = pack_key(sim, rank);
            sidx[c] = c;
        } else {
            if (!isfinite(nsq)) continue; // non-finite rows cannot be stored (vector_database.cpp:1771-1784)
            const double dd = l2.root();
            if (!isfinite(dd)) continue;
            const float dist = static_cast<float>(dd);
#ifdef YAMS_ACCEL_MEASURE
            // bound honesty under L2: on every tier the filter's score is an UPPER bound of g = q.x - |x|^2 / 2 (the error term is
            // folded into it); rows the int8 tier lists unconditionally (no usable norm) carry +inf
            if (a.stat_rescored && !a.all_rows_listed && !a.rank_row) {
                const double g = dot - 0.5 * nsq, fs = static_cast<double>(key_score(ck));
                if (g > fs + 1e-6 * (fabs(dot) + 0.5 * nsq) + 1e-30) {
                    const unsigned long long nth = atomicAdd(a.stat_rescored + 4, 1ull);
                    if (nth < 6 && a.stat_rescored[6] == 0x5eed) printf("  honesty (L2): query %u cand %u row %u g %.9g filter %.9g nsq %.6g\n", q, c, row, g, fs, nsq);
                }
                atomicAdd(a.stat_rescored + 5, 1ull);
            }
#endif
            // computeCosineSimilarity (vector_database.cpp:1786-1810): sqrt each norm, 0 on zero norm
            const double na = qn, nb = sqrt(nsq);
            const double cs = (na == 0.0 || nb == 0.0) ? 0.0 : dot / (na * nb);
            saux[c] = static_cast<float>(cs);
            skey[c] = pack_key(-dist, rank); // ascending distance == descending -dist
            sidx[c] = c;
        }
    }
    if (a.stat_rescored && local_rescored) atomicAdd(a.stat_rescored, (unsigned long long)local_rescored);
    __syncthreads();

    // bitonic sort (key desc) of rs pairs
    for (int kk = 2; kk <= rs; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (rs >> 1); t += blockDim.x) { // pair t = (i, i + j)
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

    // ---- verification ---------------------------------------------------------------------------
    // Everything outside the re-scored set has filter score <= `ob`:
    //   * the first list entry that was not re-scored (cand[n_cand]) if there is one,
    //   * else the list threshold tau (rows below tau never entered the list),
    //   * or nothing at all when the whole corpus was listed.
    __shared__ uint32_t s_nvalid;
    __shared__ uint32_t s_status;
    if (threadIdx.x == 0) {
        uint32_t nv = 0;
        // count valid (non-zero) keys by binary search on the sorted array
        uint32_t lo = 0, hi = static_cast<uint32_t>(rs);
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skey[mid] != 0) lo = mid + 1; else hi = mid; }
        nv = lo;
        s_nvalid = nv;
        uint32_t status = 0;
        if (!a.all_rows_listed) {
            const bool overflow = (a.list_count && a.list_count[q] > a.list_cap) || (a.q_over && a.q_over[q] != 0);
            const float ninf = -__builtin_inff();
            double ob = 0.0;
            bool has_outside = false;
            const uint32_t n_walked = s_ncand;      // (< n_cand when the walk closed early)
            const uint64_t nextk = (n_walked < a.cand_stride) ? cand[n_walked] : 0;
            if (nextk != 0) { has_outside = true; ob = static_cast<double>(key_score(nextk)); }
            else if (a.tau && a.tau[q] > ninf) { has_outside = true; ob = static_cast<double>(a.tau[q]); }
            // (a NaN tau admitted every row to the list, so nothing is outside)
            if (overflow) status = 1;
            else if (has_outside) {
                if (ob != ob) status = 1; // NaN bound: cannot prove anything
                else if (METRIC == YAMS_SCAN_COSINE) {
                    const float reach = static_cast<float>(ob + a.err_bound + 1e-12);
                    if (nv >= a.k) {
                        const float sim_w = key_score(skey[a.k - 1]);
                        if (!(reach < sim_w)) status = 1;
                    } else if (!(reach < a.threshold)) {
                        status = 1; // an outside row might still pass the threshold
                    }
                } else {
                    // ob bounds g = q.x - |x|^2/2 of every outside row from above:
                    // d^2 = |q|^2 - 2 g >= |q|^2 - 2 ob
                    if (nv >= a.k) {
                        double d2 = qn * qn - 2.0 * ob;
                        // fp32 accumulation: the distance an outside row would be GIVEN may undercut its true one by the
                        // summation error (nonnegative terms: relative), a possible underflow of tiny squares and the
                        // rounding of the root
                        if (ACC != 0) d2 = d2 * (1.0 - a.l2_acc_slack) - 1e-30;
                        if (d2 < 0.0) d2 = 0.0;
                        const float dmin = static_cast<float>(sqrt(d2) * (ACC != 0 ? 1.0 - 1.2e-7 : 1.0 - 1e-12));
                        const float dist_w = -key_score(skey[a.k - 1]);
                        if (!(dmin > dist_w)) status = 1;
                    } else {
                        status = 1; // fewer than k valid rows re-scored while others exist
                    }
                }
            }
        }
        s_status = status;
        a.out_status[q] = status;
#ifdef YAMS_ACCEL_MEASURE
        // measurement: how many candidates (in the order of their bounds) this query NEEDED — the top k by exact score
        // all lie among the first m, and the (m + 1)-th bound is below the k-th exact score; summed / maximised over
        // the batch in stat[1] / stat[2] / queries counted in stat[3] (YAMS_ACCEL_DUMP_NEEDED prints them)
        if (METRIC == YAMS_SCAN_COSINE && a.stat_rescored && status == 0 && nv >= a.k && !a.all_rows_listed) {
            uint32_t m = 0;
            for (uint32_t i = 0; i < a.k; ++i) m = sidx[i] + 1 > m ? sidx[i] + 1 : m;
            const float sim_w = key_score(skey[a.k - 1]);
            while (m < a.n_cand && !(static_cast<float>(static_cast<double>(key_score(cand[m])) + a.err_bound + 1e-12) < sim_w)) ++m;
            atomicAdd(a.stat_rescored + 1, static_cast<unsigned long long>(m));
            atomicMax(a.stat_rescored + 2, static_cast<unsigned long long>(m));
            atomicAdd(a.stat_rescored + 3, 1ull);
        }
#endif
    }
    __syncthreads();
    const uint32_t nv = s_nvalid;
    const uint32_t take = nv < a.k ? nv : a.k;
    if (METRIC == YAMS_SCAN_COSINE) {
        for (uint32_t i = threadIdx.x; i < a.k; i += blockDim.x) {
            const uint64_t o = static_cast<uint64_t>(q) * a.k + i;
            if (i < take) {
                const uint32_t row = a.rank_row ? a.rank_row[key_idx(cand[sidx[i]])] : key_idx(cand[sidx[i]]);
                a.out_scores[o] = key_score(skey[i]);
                a.out_rows[o] = global_row(a, row);
                if (a.out_ranks) a.out_ranks[o] = key_idx(skey[i]);
            } else {
                a.out_scores[o] = -__builtin_inff();
                a.out_rows[o] = -1;
                if (a.out_ranks) a.out_ranks[o] = 0xffffffffu;
            }
            if (a.out_dist) a.out_dist[o] = (i < take) ? 1.0f - key_score(skey[i]) : __builtin_inff();
        }
        if (threadIdx.x == 0) a.out_counts[q] = take;
    } else {
        // vec0 semantics: the k nearest, THEN the cosine threshold (:4506-4510), order preserved.
        __shared__ uint32_t s_outn;
        if (threadIdx.x == 0) {
            uint32_t outn = 0;
            const bool defer = (a.flags & YAMS_SCAN_FLAG_DEFER_THRESHOLD) != 0;
            for (uint32_t i = 0; i < take; ++i) {
                const float cs = saux[sidx[i]];
                if (!defer && cs < a.threshold) continue;
                const uint64_t o = static_cast<uint64_t>(q) * a.k + outn;
                const uint32_t row = a.rank_row ? a.rank_row[key_idx(cand[sidx[i]])] : key_idx(cand[sidx[i]]);
                a.out_scores[o] = cs;
                a.out_rows[o] = global_row(a, row);
                if (a.out_dist) a.out_dist[o] = -key_score(skey[i]);
                if (a.out_ranks) a.out_ranks[o] = key_idx(skey[i]);
                ++outn;
            }
            s_outn = outn;
            a.out_counts[q] = outn;
        }
        __syncthreads();
        for (uint32_t i = s_outn + threadIdx.x; i < a.k; i += blockDim.x) {
            const uint64_t o = static_cast<uint64_t>(q) * a.k + i;
            a.out_scores[o] = -__builtin_inff();
            a.out_rows[o] = -1;
            if (a.out_dist) a.out_dist[o] = __builtin_inff();
            if (a.out_ranks) a.out_ranks[o] = 0xffffffffu;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Exhaustive fp64 scoring (small corpora, and the last-resort path for queries that cannot be
// verified).  One thread per (row, query slot); key = (score, row) so a multi-level block top-k
// can reduce it; the survivors then go through rescore_select_kernel like any candidate list.
// -------------------------------------------------------------------------------------------------
template <int METRIC, int ACC = 0>
__global__ __launch_bounds__(256) void exact_keys_kernel(const float* rows, uint64_t n_rows,
                                                         uint32_t dim, const float* queries,
                                                         const double* qnorm,
                                                         const uint32_t* tie_rank,
                                                         const uint32_t* row_mask,
                                                         const uint32_t* rows_sel, uint64_t n_sel,
                                                         const uint32_t* qmap, float threshold,
                                                         uint32_t flags,
                                                         uint64_t* keys, uint64_t key_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sq = reinterpret_cast<float*>(smem);
    const uint32_t slot = blockIdx.y;
    const uint32_t q = qmap ? qmap[slot] : slot;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x)
        sq[i] = queries[static_cast<uint64_t>(q) * dim + i];
    __syncthreads();
    const uint64_t slot_i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    uint64_t row = slot_i;
    if (rows_sel) {
        if (slot_i >= n_sel) return;
        row = rows_sel[slot_i];
    } else {
        if (row >= n_rows) return;
        if (row_mask && !((row_mask[row >> 5] >> (row & 31)) & 1u)) {
            keys[static_cast<uint64_t>(slot) * key_stride + slot_i] = 0;
            return;
        }
    }
    const float* x = rows + row * dim;
    double nsq = 0.0, dot = 0.0;
    L2Sum<l2_lanes(ACC), l2_fused(ACC)> l2; l2.clear();
    for (uint32_t i = 0; i < dim; ++i) {
        const double sv = static_cast<double>(x[i]);
        const double qv = static_cast<double>(sq[i]);
        nsq = fma(sv, sv, nsq);
        dot = fma(sv, qv, dot);
        if (METRIC == YAMS_SCAN_L2 && ACC == 0) l2.add(x[i], sq[i], 0);
    }
    if (METRIC == YAMS_SCAN_L2 && ACC != 0) l2_sum_tail(l2, x, sq, 0, dim);
    uint64_t key = 0;
    const uint32_t kidx = tie_rank ? tie_rank[row] : static_cast<uint32_t>(row);
    if (METRIC == YAMS_SCAN_COSINE) {
        if (isfinite(nsq) && ((flags & YAMS_SCAN_FLAG_RECORD_PATH) ? nsq >= 1e-10 : nsq > 1e-12)) {
            const double denom = sqrt(nsq) * qnorm[q];
            const double sd = denom > 0.0 ? dot / denom : 0.0;
            if (isfinite(sd)) {
                const float sim = static_cast<float>(sd);
                if (!(sim < threshold)) key = pack_key(sim, kidx);
            }
        }
    } else {
        if (isfinite(nsq)) {
            const double dd = l2.root();
            if (isfinite(dd)) key = pack_key(-static_cast<float>(dd), kidx);
        }
    }
    keys[static_cast<uint64_t>(slot) * key_stride + slot_i] = key;
}

// Row ordinals of the set bits of an allow-mask (any order: the keys carry the row / rank).
__global__ __launch_bounds__(256) void compact_mask_kernel(const uint32_t* row_mask, uint64_t n_rows,
                                                           uint32_t* rows_sel,
                                                           unsigned long long* counter) {
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool on = row < n_rows && ((row_mask[row >> 5] >> (row & 31)) & 1u);
    const unsigned long long ball = __ballot(on);
    if (ball == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, static_cast<unsigned long long>(__popcll(ball)));
    base = __shfl(base, 0);
    if (on) rows_sel[base + __popcll(ball & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(row);
}

// -------------------------------------------------------------------------------------------------
// Shard merge (after the all-gather): per query, the union of n_shards sorted lists -> top k.
// -------------------------------------------------------------------------------------------------
struct MergeArgs {
    uint32_t n_shards, n_queries, k, metric;
    float threshold;
    const float* in_scores; const int64_t* in_rows; const uint32_t* in_counts;
    const float* in_dist; const uint32_t* in_ranks;
    // distance between consecutive shards of each input, in ELEMENTS of that input (dense arrays:
    // n_queries * k resp. n_queries; packed per-shard records: record bytes / element size)
    uint64_t st_scores, st_rows, st_counts, st_dist, st_ranks;
    const uint32_t* rank_of_row; // nullable: global tie rank of global row id (row - rank_row_base)
    int64_t rank_row_base;
    float* out_scores; int64_t* out_rows; uint32_t* out_counts; float* out_dist;
};

__global__ __launch_bounds__(256) void merge_topk_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // keys: (primary score, tie) packed as 96 bits is awkward; sort indices by comparing tuples.
    const uint32_t q = blockIdx.x;
    const uint32_t total = a.n_shards * a.k;
    uint32_t cap = 1;
    while (cap < total) cap <<= 1;
    uint32_t* sidx = reinterpret_cast<uint32_t*>(smem); // [cap]
    auto valid = [&](uint32_t e) -> bool {
        if (e >= total) return false;
        const uint32_t sh = e / a.k, i = e % a.k;
        return i < a.in_counts[static_cast<uint64_t>(sh) * a.st_counts + q];
    };
    // element e = (shard, i) of query q: offset inside its shard's [n_queries][k] array
    auto within = [&](uint32_t e) -> uint64_t { return static_cast<uint64_t>(q) * a.k + e % a.k; };
    auto score_of = [&](uint32_t e) -> float { return a.in_scores[(e / a.k) * a.st_scores + within(e)]; };
    auto dist_of = [&](uint32_t e) -> float { return a.in_dist[(e / a.k) * a.st_dist + within(e)]; };
    auto row_of = [&](uint32_t e) -> int64_t { return a.in_rows[(e / a.k) * a.st_rows + within(e)]; };
    // better(x, y): x sorts before y
    auto better = [&](uint32_t x, uint32_t y) -> bool {
        const bool vx = valid(x), vy = valid(y);
        if (vx != vy) return vx;
        if (!vx) return x < y;
        if (a.metric == YAMS_SCAN_L2) {
            const float dx = dist_of(x), dy = dist_of(y);
            if (dx != dy) return dx < dy;
        } else {
            const float sx = score_of(x), sy = score_of(y);
            if (sx != sy) return sx > sy;
        }
        const int64_t ix = row_of(x), iy = row_of(y);
        if (a.in_ranks) { // ranks the caller guarantees to be comparable across shards
            const uint32_t rx = a.in_ranks[(x / a.k) * a.st_ranks + within(x)], ry = a.in_ranks[(y / a.k) * a.st_ranks + within(y)];
            if (rx != ry) return rx < ry;
        } else if (a.rank_of_row) { // the corpus-wide chunk_id ranking, looked up by global row id
            const uint32_t rx = a.rank_of_row[ix - a.rank_row_base], ry = a.rank_of_row[iy - a.rank_row_base];
            if (rx != ry) return rx < ry;
        }
        if (ix != iy) return ix < iy;
        return x < y;
    };
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) sidx[i] = i;
    __syncthreads();
    for (uint32_t kk = 2; kk <= cap; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t x = sidx[i], y = sidx[ixj];
                    const bool up = (i & kk) == 0;
                    // "up" segments want best first
                    const bool swap = up ? better(y, x) : better(x, y);
                    if (swap) { sidx[i] = y; sidx[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    __shared__ uint32_t s_outn;
    if (threadIdx.x == 0) {
        uint32_t outn = 0;
        for (uint32_t i = 0; i < a.k && i < cap; ++i) {
            const uint32_t e = sidx[i];
            if (!valid(e)) break;
            const float sc = score_of(e);
            if (a.metric == YAMS_SCAN_L2 && sc < a.threshold) continue; // :4508-4510
            const uint64_t d = static_cast<uint64_t>(q) * a.k + outn;
            a.out_scores[d] = sc;
            a.out_rows[d] = row_of(e);
            if (a.out_dist) a.out_dist[d] = a.in_dist ? dist_of(e) : 1.0f - sc;
            ++outn;
        }
        s_outn = outn;
        a.out_counts[q] = outn;
    }
    __syncthreads();
    for (uint32_t i = s_outn + threadIdx.x; i < a.k; i += blockDim.x) {
        const uint64_t d = static_cast<uint64_t>(q) * a.k + i;
        a.out_scores[d] = -__builtin_inff();
        a.out_rows[d] = -1;
        if (a.out_dist) a.out_dist[d] = __builtin_inff();
    }
}

// -------------------------------------------------------------------------------------------------
// Synthetic data: Philox4x32-10, the same generator the CPU checker restates.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t c_lo, uint64_t c_hi,
                                              uint32_t out[4]) {
    uint32_t c0 = static_cast<uint32_t>(c_lo), c1 = static_cast<uint32_t>(c_lo >> 32);
    uint32_t c2 = static_cast<uint32_t>(c_hi), c3 = static_cast<uint32_t>(c_hi >> 32);
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
        const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = static_cast<uint32_t>(p1);
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = static_cast<uint32_t>(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One wave per row: lanes generate 4 columns each per step, the norm is accumulated in float in
// COLUMN ORDER (a single lane walks the row) so that a CPU regeneration of the same rows matches bit for bit.
__global__ __launch_bounds__(64) void synth_rows_kernel(uint64_t seed, uint64_t row0,
                                                        uint64_t n_rows, uint32_t dim, float* out) {
    extern __shared__ float srow[];
    const uint64_t r = blockIdx.x;
    if (r >= n_rows) return;
    for (uint32_t j4 = threadIdx.x; j4 * 4 < dim; j4 += 64) {
        uint32_t w[4];
        philox4x32_10(seed, row0 + r, j4, w);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (j4 * 4 + t < dim)
                srow[j4 * 4 + t] = static_cast<float>(w[t] >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    __syncthreads();
    __shared__ float s_nrm;
    if (threadIdx.x == 0) {
        float nsq = 0.f;
        for (uint32_t j = 0; j < dim; ++j) nsq = __fadd_rn(nsq, __fmul_rn(srow[j], srow[j]));
        s_nrm = __fsqrt_rn(nsq);
    }
    __syncthreads();
    const float nrm = s_nrm;
    for (uint32_t j = threadIdx.x; j < dim; j += 64) {
        const float v = srow[j];
        out[r * dim + j] = nrm > 0.f ? __fdiv_rn(v, nrm) : v;
    }
}

__global__ void synth_bytes_kernel(uint64_t seed, uint64_t blob_id0, uint64_t n_blobs,
                                   uint64_t blob_len, uint8_t* out) {
    const uint64_t words_per_blob = (blob_len + 15) / 16;
    const uint64_t total = words_per_blob * n_blobs;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint64_t b = i / words_per_blob, w16 = i % words_per_blob;
        uint32_t w[4];
        philox4x32_10(seed, blob_id0 + b, w16, w);
        uint8_t* dst = out + b * blob_len + w16 * 16;
        const uint64_t left = blob_len - w16 * 16;
        if (left >= 16 && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (uint64_t t = 0; t < 16 && t < left; ++t)
                dst[t] = static_cast<uint8_t>(w[t / 4] >> (8 * (t % 4)));
        }
    }
}

// =================================================================================================
// Host-side launchers (called from accel_api.cpp through scan_launch.h)
// =================================================================================================
} // namespace yams_accel

#include "scan_launch.h"

namespace yams_accel {

#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_prep_queries(hipStream_t st, const float* q, uint32_t nq, uint32_t dim,
                               int metric, float* qprep, double* qnorm, float* qnorm_up,
                               uint32_t* qflags, uint32_t* zero_words, uint32_t n_zero_words) {
    const bool staged = static_cast<size_t>(dim) * 4 <= 48 * 1024;
    hipLaunchKernelGGL(prep_queries_kernel, dim3(nq), dim3(128), staged ? static_cast<size_t>(dim) * 4 : 0, st, q,
                       nq, dim, metric, qprep, qnorm, qnorm_up, qflags, staged ? 1 : 0, zero_words, zero_words ? n_zero_words : 0u);
    LAUNCH_CHECK();
    return hipSuccess;
}

ScanArgs make_scan_args(const ScanLaunch& L) {
    ScanArgs a{};
    a.rows = L.rows; a.rows_bf16 = L.rows_bf16; a.rows_nsq = L.rows_nsq;
    a.rows_i8 = L.rows_i8; a.rows_i8_meta = L.rows_i8_meta; a.q_i8 = L.q_i8; a.q_meta = L.q_meta; a.q_thr = L.q_thr;
    a.log_key = L.log_key; a.log_q = L.log_q; a.log_cnt = L.log_cnt; a.log_cap = L.log_cap; a.q_over = L.q_over; a.i8_sync = L.i8_sync; a.i8_row_bias = L.i8_row_bias; a.i8_q_bias = L.i8_q_bias; a.l2_eps = L.l2_eps; a.row_mask = L.row_mask; a.qprep = L.qprep; a.q_hi = L.q_hi; a.q_lo = L.q_lo; a.q_pad = L.q_pad; a.n_rows = L.plan.n_rows; a.dim = L.plan.dim;
    a.n_queries = L.plan.n_queries; a.stride = L.plan.sample_stride; a.n_qtiles = L.plan.n_qtiles;
    a.dense = L.dense; a.gmax = L.gmax; a.sample_rows = L.plan.sample_rows;
    a.n_groups = L.plan.n_groups; a.tau = L.tau; a.list_count = L.list_count; a.list = L.list;
    a.list_cap = L.plan.list_cap; a.qnorm_up = L.qnorm_up; a.err_coef = L.err_coef;
    return a;
}

hipError_t launch_scan_sample(hipStream_t st, const ScanLaunch& L, int metric) {
    ScanArgs a = make_scan_args(L);
    a.n_sel_tiles = L.plan.n_sample_tiles;
    if (a.n_sel_tiles == 0) return hipSuccess;
    const uint32_t groups = (a.n_sel_tiles + 7) / 8;
    const uint32_t grid = groups * a.n_qtiles * 8;
    if (metric == YAMS_SCAN_COSINE)
        hipLaunchKernelGGL((scan_tiles_kernel<MODE_SAMPLE, YAMS_SCAN_COSINE>), dim3(grid),
                           dim3(SCAN_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((scan_tiles_kernel<MODE_SAMPLE, YAMS_SCAN_L2>), dim3(grid),
                           dim3(SCAN_THREADS), 0, st, a);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_scan_filter(hipStream_t st, const ScanLaunch& L, int metric) {
    ScanArgs a = make_scan_args(L);
    a.n_sel_tiles = L.plan.n_filter_tiles;
    if (a.n_sel_tiles == 0) return hipSuccess;
    const uint32_t groups = (a.n_sel_tiles + 7) / 8;
    const uint32_t grid = groups * a.n_qtiles * 8;
    if (metric == YAMS_SCAN_COSINE)
        hipLaunchKernelGGL((scan_tiles_kernel<MODE_FILTER, YAMS_SCAN_COSINE>), dim3(grid),
                           dim3(SCAN_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((scan_tiles_kernel<MODE_FILTER, YAMS_SCAN_L2>), dim3(grid),
                           dim3(SCAN_THREADS), 0, st, a);
    LAUNCH_CHECK();
    return hipSuccess;
}

// Multi-level descending top-`keep` of per-query key arrays.  `work` must hold
// 2 * n_slots * ceil(n_max / kSelectCap) * keep keys.  The result (sorted, 0-padded, `keep` keys
// per slot) ends up at `*result` with stride `*result_stride`.
template <typename K>
static hipError_t topk_multilevel(hipStream_t st, const K* in, const uint32_t* in_counts,
                                  uint64_t in_stride, uint32_t n_max, uint32_t count_clip,
                                  uint32_t n_slots, const uint32_t* qmap, uint32_t keep, K* work,
                                  const K** result, uint64_t* result_stride) {
    // each level must shrink the problem: 2 chunks * keep has to fit one chunk
    if (keep == 0 || keep > static_cast<uint32_t>(kSelectCap) / 2) return hipErrorInvalidValue;
    const K* cur = in;
    const uint32_t* cur_counts = in_counts;
    uint64_t cur_stride = in_stride;
    uint32_t cur_n = n_max < count_clip ? n_max : count_clip;
    const uint32_t* cur_qmap = qmap;
    K* bufs[2];
    uint32_t chunks0 = (cur_n + kSelectCap - 1) / kSelectCap;
    if (chunks0 == 0) chunks0 = 1;
    bufs[0] = work;
    bufs[1] = work + static_cast<uint64_t>(n_slots) * chunks0 * keep;
    int which = 0;
    for (;;) {
        uint32_t chunks = (cur_n + kSelectCap - 1) / kSelectCap;
        if (chunks == 0) chunks = 1;
        K* dst = bufs[which];
        const uint64_t dst_stride = static_cast<uint64_t>(chunks) * keep;
        hipLaunchKernelGGL((topk_block_kernel<K, kSelectCap>), dim3(chunks, n_slots), dim3(256), 0,
                           st, cur, cur_counts, cur_stride, cur_n, count_clip, dst, dst_stride,
                           keep, cur_qmap);
        LAUNCH_CHECK();
        cur = dst; cur_counts = nullptr; cur_stride = dst_stride; cur_n = chunks * keep;
        cur_qmap = nullptr; // outputs are slot-indexed from now on
        count_clip = 0xffffffffu;
        which ^= 1;
        if (chunks == 1) break;
    }
    *result = cur;
    *result_stride = cur_stride;
    return hipSuccess;
}

hipError_t launch_select_tau(hipStream_t st, const ScanLaunch& L, uint32_t* /*work32*/) {
    const uint32_t nq = L.plan.n_queries;
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(tau_select_kernel, dim3(nq), dim3(256), 0, st, L.gmax, L.plan.n_groups,
                       L.plan.tau_rank, L.tau_out, L.tau_rows_meta, L.tau_n_blocks, L.q_meta, L.tau_rank2, L.tau_max_groups, L.tau_e_scale);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_collect_sample(hipStream_t st, const ScanLaunch& L) {
    if (L.plan.sample_rows == 0 || L.plan.n_groups == 0) return hipSuccess;
    uint32_t gx = (L.plan.n_groups + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(collect_sample_kernel, dim3(gx, L.plan.n_queries), dim3(256), 0, st, L.dense,
                       L.gmax, L.plan.n_groups, L.plan.n_queries, L.plan.sample_rows, L.plan.tile_rows,
                       L.plan.sample_stride, L.plan.n_rows, L.tau, L.list_count, L.list, L.plan.list_cap,
                       L.sample_layout);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_select_lists(hipStream_t st, const uint64_t* list, const uint32_t* list_count,
                               uint32_t list_cap, uint32_t n_slots, const uint32_t* qmap,
                               uint32_t keep, uint64_t* work, const uint64_t** result,
                               uint64_t* result_stride) {
    return topk_multilevel<uint64_t>(st, list, list_count, list_cap, list_cap, list_cap, n_slots,
                                     qmap, keep, work, result, result_stride);
}

hipError_t launch_exact_keys(hipStream_t st, int metric, const float* rows, uint64_t n_rows,
                             uint32_t dim, const float* queries, const double* qnorm,
                             const uint32_t* tie_rank, const uint32_t* row_mask,
                             const uint32_t* rows_sel, uint64_t n_sel, const uint32_t* qmap,
                             uint32_t n_slots, float threshold, uint32_t flags, uint64_t* keys,
                             uint64_t key_stride) {
    const uint64_t n_items = rows_sel ? n_sel : n_rows;
    if (n_items == 0) return hipSuccess;
    const uint32_t gx = static_cast<uint32_t>((n_items + 255) / 256);
    const size_t sh = static_cast<size_t>(dim) * sizeof(float);
    const uint32_t acc = flags & YAMS_SCAN_FLAG_L2_ACC_MASK; // the host's choice of vec0's distance arithmetic (L2 only)
    const bool fused = acc != 0 && (flags & YAMS_SCAN_FLAG_L2_ACC_FUSED) != 0; // ... accumulated with one fused multiply-add
#define YAMS_EXACT_KEYS(ACCF) hipLaunchKernelGGL((exact_keys_kernel<YAMS_SCAN_L2, ACCF>), dim3(gx, n_slots), dim3(256), sh, st, \
    rows, n_rows, dim, queries, qnorm, tie_rank, row_mask, rows_sel, n_sel, qmap, threshold, flags, keys, key_stride)
    if (metric == YAMS_SCAN_COSINE)
        hipLaunchKernelGGL((exact_keys_kernel<YAMS_SCAN_COSINE>), dim3(gx, n_slots), dim3(256), sh,
                           st, rows, n_rows, dim, queries, qnorm, tie_rank, row_mask, rows_sel, n_sel, qmap, threshold, flags, keys, key_stride);
    else if (fused && acc == YAMS_SCAN_FLAG_L2_ACC_F32) YAMS_EXACT_KEYS(64 | 1);
    else if (fused && acc == YAMS_SCAN_FLAG_L2_ACC_F32X8) YAMS_EXACT_KEYS(64 | 8);
    else if (fused && acc == YAMS_SCAN_FLAG_L2_ACC_F32X16) YAMS_EXACT_KEYS(64 | 16);
    else if (acc == YAMS_SCAN_FLAG_L2_ACC_F32)
        hipLaunchKernelGGL((exact_keys_kernel<YAMS_SCAN_L2, 1>), dim3(gx, n_slots), dim3(256), sh, st,
                           rows, n_rows, dim, queries, qnorm, tie_rank, row_mask, rows_sel, n_sel, qmap, threshold, flags, keys, key_stride);
    else if (acc == YAMS_SCAN_FLAG_L2_ACC_F32X8)
        hipLaunchKernelGGL((exact_keys_kernel<YAMS_SCAN_L2, 8>), dim3(gx, n_slots), dim3(256), sh, st,
                           rows, n_rows, dim, queries, qnorm, tie_rank, row_mask, rows_sel, n_sel, qmap, threshold, flags, keys, key_stride);
    else if (acc == YAMS_SCAN_FLAG_L2_ACC_F32X16)
        hipLaunchKernelGGL((exact_keys_kernel<YAMS_SCAN_L2, 16>), dim3(gx, n_slots), dim3(256), sh, st,
                           rows, n_rows, dim, queries, qnorm, tie_rank, row_mask, rows_sel, n_sel, qmap, threshold, flags, keys, key_stride);
    else
        hipLaunchKernelGGL((exact_keys_kernel<YAMS_SCAN_L2>), dim3(gx, n_slots), dim3(256), sh, st,
                           rows, n_rows, dim, queries, qnorm, tie_rank, row_mask, rows_sel, n_sel, qmap, threshold, flags, keys, key_stride);
#undef YAMS_EXACT_KEYS
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_compact_mask(hipStream_t st, const uint32_t* row_mask, uint64_t n_rows,
                               uint32_t* rows_sel, unsigned long long* counter) {
    if (n_rows == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(compact_mask_kernel, dim3(static_cast<uint32_t>((n_rows + 255) / 256)),
                       dim3(256), 0, st, row_mask, n_rows, rows_sel, counter);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_topk_keys(hipStream_t st, const uint64_t* keys, uint64_t key_stride,
                            uint32_t n_per_slot, uint32_t n_slots, uint32_t keep, uint64_t* work,
                            const uint64_t** result, uint64_t* result_stride) {
    return topk_multilevel<uint64_t>(st, keys, nullptr, key_stride, n_per_slot, 0xffffffffu,
                                     n_slots, nullptr, keep, work, result, result_stride);
}

hipError_t launch_rescore(hipStream_t st, int metric, const RescoreLaunch& R) {
    RescoreArgs a{};
    a.rows = R.rows; a.n_rows = R.n_rows; a.dim = R.dim; a.queries = R.queries; a.qnorm = R.qnorm;
    a.tie_rank = R.tie_rank; a.rank_row = R.rank_row; a.row_base = R.row_base; a.cand = R.cand;
    a.stripe_rows = R.stripe_rows; a.n_stripes = R.n_stripes; a.stripe_index = R.stripe_index;
    a.cand_stride = R.cand_stride; a.n_cand = R.n_cand; a.tau = R.tau;
    a.list_count = R.list_count; a.list_cap = R.list_cap; a.all_rows_listed = R.all_rows_listed;
    a.qmap = R.qmap; a.k = R.k; a.threshold = R.threshold; a.flags = R.flags;
    a.err_bound = R.err_bound; a.out_scores = R.out_scores; a.out_rows = R.out_rows;
    a.out_counts = R.out_counts; a.out_dist = R.out_dist; a.out_ranks = R.out_ranks;
    a.out_status = R.out_status; a.stat_rescored = R.stat_rescored; a.q_over = R.q_over;
    size_t rs = 64;
    while (rs < R.n_cand) rs <<= 1;
    if (rs > static_cast<size_t>(RS_MAX)) rs = RS_MAX;
    // one round of the candidate walk when the candidates fit a block (384 candidates: 6 waves)
    uint32_t threads = (std::min<uint32_t>(std::max<uint32_t>(R.n_cand, 64u), 512u) + 63u) & ~63u;
    // Rounds of 128 candidates wherever the walk may close early (every proof-carrying re-score): the plan's stage 1 (384
    // candidates) mostly closes after two of its three rounds — the uniform bench rows need ~250 — and several workgroups fit a
    // CU's LDS where one of 384 / 512 threads did (headline step 7.71 -> 7.62 ms, config 2 0.268 -> 0.244 on one box); the
    // widened / deep stage closes after any round.
    // (lists of more than 512: rounds of 256 measured 3 % better on the non-uniform legs than rounds of 128)
    if (R.n_cand > 128 && !(R.flags & kRescoreFlagPqRerank)) threads = R.n_cand > 512 ? 256 : 128;
#ifdef YAMS_ACCEL_MEASURE
    if (const char* e = std::getenv("YAMS_ACCEL_RESCORE_FORM")) {   // 1: no early close, 2: 512-thread rounds, 3: both
        const int v = std::atoi(e);
        if (v & 1) a.flags |= kRescoreFlagNoEarlyClose;
        if (v & 2) threads = (std::min<uint32_t>(std::max<uint32_t>(R.n_cand, 64u), 512u) + 63u) & ~63u;
        if ((v & 4) && R.n_cand > 256 && !(R.flags & kRescoreFlagPqRerank)) threads = 256;   // 4 / 8: rounds of 256 / 192 candidates
        if ((v & 8) && R.n_cand > 192 && !(R.flags & kRescoreFlagPqRerank)) threads = 192;
    }
#endif
    const size_t sh = rs * (sizeof(uint64_t) + sizeof(uint32_t) + sizeof(float)) +
                      ((static_cast<size_t>(R.dim) + 3) & ~static_cast<size_t>(3)) * sizeof(float) +
                      (threads / 64) * 64 * RS_STAGE_STRIDE * sizeof(float);
    const uint32_t acc = R.flags & YAMS_SCAN_FLAG_L2_ACC_MASK;
    // fp32 summation of dim nonnegative squares, each from a rounded difference and a rounded product: relative error
    // below (dim + 2) u of the sum for ANY order of the additions (sequential is the worst case); 8 spare u, 1 % slack
    a.l2_acc_slack = (static_cast<double>(R.dim) + 8.0) * 5.9604644775390625e-8 * 1.01;
    const bool fused = acc != 0 && (R.flags & YAMS_SCAN_FLAG_L2_ACC_FUSED) != 0;
    if (metric == YAMS_SCAN_COSINE)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_COSINE>), dim3(R.n_slots), dim3(threads),
                           sh, st, a);
    else if (fused && acc == YAMS_SCAN_FLAG_L2_ACC_F32)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2, 64 | 1>), dim3(R.n_slots), dim3(threads), sh, st, a);
    else if (fused && acc == YAMS_SCAN_FLAG_L2_ACC_F32X8)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2, 64 | 8>), dim3(R.n_slots), dim3(threads), sh, st, a);
    else if (fused && acc == YAMS_SCAN_FLAG_L2_ACC_F32X16)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2, 64 | 16>), dim3(R.n_slots), dim3(threads), sh, st, a);
    else if (acc == YAMS_SCAN_FLAG_L2_ACC_F32)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2, 1>), dim3(R.n_slots), dim3(threads), sh, st, a);
    else if (acc == YAMS_SCAN_FLAG_L2_ACC_F32X8)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2, 8>), dim3(R.n_slots), dim3(threads), sh, st, a);
    else if (acc == YAMS_SCAN_FLAG_L2_ACC_F32X16)
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2, 16>), dim3(R.n_slots), dim3(threads), sh, st, a);
    else
        hipLaunchKernelGGL((rescore_select_kernel<YAMS_SCAN_L2>), dim3(R.n_slots), dim3(threads), sh,
                           st, a);
    LAUNCH_CHECK();
    return hipSuccess;
}

// Precision escalation helpers: queries whose 1-pass filter could not be proven complete are
// gathered, re-run through the split (3-pass) filter as their own small batch, and scattered back.
__global__ void gather_queries_kernel(const float* queries, const uint32_t* qmap, uint32_t n_slots,
                                      uint32_t dim, float* out) {
    const uint64_t total = static_cast<uint64_t>(n_slots) * dim;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t s = static_cast<uint32_t>(i / dim), c = static_cast<uint32_t>(i % dim);
        out[i] = queries[static_cast<uint64_t>(qmap[s]) * dim + c];
    }
}
__global__ void scatter_results_kernel(const uint32_t* qmap, uint32_t n_slots, uint32_t k,
                                       const float* s_scores, const int64_t* s_rows,
                                       const uint32_t* s_counts, const float* s_dist,
                                       const uint32_t* s_ranks, float* scores, int64_t* rows,
                                       uint32_t* counts, float* dist, uint32_t* ranks) {
    const uint32_t s = blockIdx.x;
    if (s >= n_slots) return;
    const uint64_t q = qmap[s];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        scores[q * k + i] = s_scores[static_cast<uint64_t>(s) * k + i];
        rows[q * k + i] = s_rows[static_cast<uint64_t>(s) * k + i];
        if (dist) dist[q * k + i] = s_dist[static_cast<uint64_t>(s) * k + i];
        if (ranks) ranks[q * k + i] = s_ranks[static_cast<uint64_t>(s) * k + i];
    }
    if (threadIdx.x == 0) counts[q] = s_counts[s];
}

// tau' of the int8 tier's retry (scan_api.cpp, stage 2a): for every unproven query the k-th best EXACT score the first stages
// found, one ulp down (the proof compares strictly), and — from the sample pass's group maxima — how many sample groups reach
// it (each stands for `stride` rows of the shard: the size of the list the retry will build).  Fewer than k rows found: no
// usable threshold (est = 0xffffffff).  One workgroup per unproven query.
__global__ __launch_bounds__(256) void retry_tau_kernel(const float* scores, const uint32_t* counts, uint32_t k, const uint32_t* fmap,
                                                        const uint32_t* gmax, uint32_t n_groups, float* tau_out, uint32_t* est_out, float slack) {
    __shared__ uint32_t s_cnt;
    const uint32_t slot = blockIdx.x;
    const uint32_t q = fmap[slot];
    if (counts[q] < k) {
        if (threadIdx.x == 0) { tau_out[slot] = __builtin_inff(); est_out[slot] = 0xffffffffu; }
        return;
    }
    const float sk = scores[static_cast<uint64_t>(q) * k + (k - 1)] - slack;     // (slack: the tier's error bound — 0 where the score IS an upper bound)
    const float t = __uint_as_float(sk > 0.f ? __float_as_uint(sk) - 1u : (sk < 0.f ? __float_as_uint(sk) + 1u : 0x80000001u)); // the next float below
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t key = f2ord(t);
    const uint32_t* g = gmax + static_cast<uint64_t>(q) * n_groups;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < n_groups; i += 256) mine += (g[i] != 0xffffffffu && g[i] >= key) ? 1u : 0u;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { tau_out[slot] = t; est_out[slot] = s_cnt; }
}
// The same under L2 (the filter's score is g = q.x - |x|^2 / 2 = (|q|^2 - d^2) / 2): a row that can still enter the k nearest has
// d <= d_k, the k-th smallest exact distance found, i.e. g >= (|q|^2 - d_k^2) / 2.  `margin` keeps the proof's strict comparison
// (and, under fp32 accumulation, its summation slack) on the safe side; two ulps down for the cast.
__global__ __launch_bounds__(256) void retry_tau_l2_kernel(const float* dist, const uint32_t* counts, uint32_t k, const uint32_t* fmap,
                                                           const double* qnorm, double margin, const uint32_t* gmax, uint32_t n_groups,
                                                           float* tau_out, uint32_t* est_out) {
    __shared__ uint32_t s_cnt;
    const uint32_t slot = blockIdx.x;
    const uint32_t q = fmap[slot];
    const float dk = counts[q] < k ? __builtin_inff() : dist[static_cast<uint64_t>(q) * k + (k - 1)];
    const double qn = qnorm[q];
    const double g = 0.5 * (qn * qn - static_cast<double>(dk) * static_cast<double>(dk) * (1.0 + margin));
    if (!(dk < __builtin_inff()) || !(g == g) || !(g > -3.0e38 && g < 3.0e38)) {
        if (threadIdx.x == 0) { tau_out[slot] = __builtin_inff(); est_out[slot] = 0xffffffffu; }
        return;
    }
    float t = static_cast<float>(g);
#pragma unroll
    for (int i = 0; i < 2; ++i)
        t = __uint_as_float(t > 0.f ? __float_as_uint(t) - 1u : (t < 0.f ? __float_as_uint(t) + 1u : 0x80000001u)); // the next float below
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t key = f2ord(t);
    const uint32_t* gm = gmax + static_cast<uint64_t>(q) * n_groups;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < n_groups; i += 256) mine += (gm[i] != 0xffffffffu && gm[i] >= key) ? 1u : 0u;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { tau_out[slot] = t; est_out[slot] = s_cnt; }
}
hipError_t launch_retry_tau_l2(hipStream_t st, const float* dist, const uint32_t* counts, uint32_t k, const uint32_t* fmap, uint32_t n_slots,
                               const double* qnorm, double margin, const uint32_t* gmax, uint32_t n_groups, float* tau_out, uint32_t* est_out) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(retry_tau_l2_kernel, dim3(n_slots), dim3(256), 0, st, dist, counts, k, fmap, qnorm, margin, gmax, n_groups, tau_out, est_out);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_retry_tau(hipStream_t st, const float* scores, const uint32_t* counts, uint32_t k, const uint32_t* fmap, uint32_t n_slots,
                            const uint32_t* gmax, uint32_t n_groups, float* tau_out, uint32_t* est_out, float slack) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(retry_tau_kernel, dim3(n_slots), dim3(256), 0, st, scores, counts, k, fmap, gmax, n_groups, tau_out, est_out, slack);
    LAUNCH_CHECK();
    return hipSuccess;
}
// results of run slots src[i] -> the caller's queries dst[i]
__global__ void scatter_results_from_kernel(const uint32_t* src, const uint32_t* dst, uint32_t n, uint32_t k, const float* s_scores,
                                            const int64_t* s_rows, const uint32_t* s_counts, const float* s_dist, const uint32_t* s_ranks,
                                            float* scores, int64_t* rows, uint32_t* counts, float* dist, uint32_t* ranks) {
    const uint32_t i0 = blockIdx.x;
    if (i0 >= n) return;
    const uint64_t s = src[i0], q = dst[i0];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        scores[q * k + i] = s_scores[s * k + i];
        rows[q * k + i] = s_rows[s * k + i];
        if (dist) dist[q * k + i] = s_dist[s * k + i];
        if (ranks) ranks[q * k + i] = s_ranks[s * k + i];
    }
    if (threadIdx.x == 0) counts[q] = s_counts[s];
}
hipError_t launch_scatter_results_from(hipStream_t st, const uint32_t* src, const uint32_t* dst, uint32_t n, uint32_t k, const float* s_scores,
                                       const int64_t* s_rows, const uint32_t* s_counts, const float* s_dist, const uint32_t* s_ranks,
                                       float* scores, int64_t* rows, uint32_t* counts, float* dist, uint32_t* ranks) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_results_from_kernel, dim3(n), dim3(128), 0, st, src, dst, n, k, s_scores, s_rows, s_counts, s_dist, s_ranks,
                       scores, rows, counts, dist, ranks);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gather_queries(hipStream_t st, const float* queries, const uint32_t* qmap,
                                 uint32_t n_slots, uint32_t dim, float* out) {
    if (n_slots == 0 || dim == 0) return hipSuccess;
    const uint64_t total = static_cast<uint64_t>(n_slots) * dim;
    uint32_t grid = static_cast<uint32_t>((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gather_queries_kernel, dim3(grid), dim3(256), 0, st, queries, qmap, n_slots, dim, out);
    LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_scatter_results(hipStream_t st, const uint32_t* qmap, uint32_t n_slots, uint32_t k,
                                  const float* s_scores, const int64_t* s_rows,
                                  const uint32_t* s_counts, const float* s_dist,
                                  const uint32_t* s_ranks, float* scores, int64_t* rows,
                                  uint32_t* counts, float* dist, uint32_t* ranks) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_results_kernel, dim3(n_slots), dim3(128), 0, st, qmap, n_slots, k,
                       s_scores, s_rows, s_counts, s_dist, s_ranks, scores, rows, counts, dist, ranks);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_merge(hipStream_t st, const MergeLaunch& M) {
    MergeArgs a{};
    a.n_shards = M.n_shards; a.n_queries = M.n_queries; a.k = M.k; a.metric = M.metric;
    a.threshold = M.threshold; a.in_scores = M.in_scores; a.in_rows = M.in_rows;
    a.in_counts = M.in_counts; a.in_dist = M.in_dist; a.in_ranks = M.in_ranks;
    const uint64_t dense = static_cast<uint64_t>(M.n_queries) * M.k;
    a.st_scores = M.st_scores ? M.st_scores : dense; a.st_rows = M.st_rows ? M.st_rows : dense;
    a.st_counts = M.st_counts ? M.st_counts : M.n_queries; a.st_dist = M.st_dist ? M.st_dist : dense;
    a.st_ranks = M.st_ranks ? M.st_ranks : dense;
    a.rank_of_row = M.rank_of_row; a.rank_row_base = M.rank_row_base;
    // L2 (vec0): equal distances come back in rowid order (`ORDER BY distance` alone, :4473) — the chunk_id ranking belongs
    // to the cosine comparator only (scan_api.cpp scan_impl; tests/test_scan_ref_l2_pin.py)
    if (M.metric == YAMS_SCAN_L2) { a.in_ranks = nullptr; a.rank_of_row = nullptr; }
    a.out_scores = M.out_scores; a.out_rows = M.out_rows; a.out_counts = M.out_counts;
    a.out_dist = M.out_dist;
    uint32_t cap = 1;
    while (cap < M.n_shards * M.k) cap <<= 1;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(M.n_queries), dim3(256), cap * sizeof(uint32_t), st, a);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_synth_rows(hipStream_t st, uint64_t seed, uint64_t row0, uint64_t n_rows,
                             uint32_t dim, float* out) {
    if (n_rows == 0) return hipSuccess;
    // grid.x is limited to 2^31-1 blocks; split very large requests
    const uint64_t kMax = 1u << 30;
    for (uint64_t done = 0; done < n_rows; done += kMax) {
        const uint64_t n = (n_rows - done < kMax) ? n_rows - done : kMax;
        hipLaunchKernelGGL(synth_rows_kernel, dim3(static_cast<uint32_t>(n)), dim3(64),
                           dim * sizeof(float), st, seed, row0 + done, n, dim,
                           out + done * dim);
        LAUNCH_CHECK();
    }
    return hipSuccess;
}

hipError_t launch_synth_bytes(hipStream_t st, uint64_t seed, uint64_t blob_id0, uint64_t n_blobs,
                              uint64_t blob_len, uint8_t* out) {
    if (n_blobs == 0 || blob_len == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_bytes_kernel, dim3(4096), dim3(256), 0, st, seed, blob_id0, n_blobs,
                       blob_len, out);
    LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace yams_accel
