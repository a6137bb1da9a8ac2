// Micro-benchmark: the MFMA skeleton of the filter loop — 8 waves per workgroup, per iteration two
// bursts of 8 MFMAs on 8 accumulators with 6 operand registers per burst, optional s_barrier between
// bursts, optional LDS fragment reads.  Reports cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// 0 bare, 1 + s_barrier per burst pair, 2 + barrier + 12 ds_read_b128 per iteration,
// 4 = 3 with a blocked source (contiguous 1 KiB pieces instead of 16 rows x 64 B);
// 3 = 2 + four 1 KiB LDS-DMA pieces per wave per iteration (a 4 x 32 KiB ring, counted vmcnt) from `src`
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned long long* cyc, const unsigned char* src,
                                            unsigned long long src_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[MODE >= 3 ? 131072 : 65536];
    f32x16 acc[8];
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    bf16x8 fa[2][2], fb[2][4];
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int s = 0; s < 2; ++s) {
        for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; fa[s][j][i] = (__bf16)(((int)(x >> 8) & 0xffff) / 32768.0f - 1.0f); }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; fb[s][j][i] = (__bf16)(((int)(x >> 8) & 0xffff) / 32768.0f - 1.0f); }
    }
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned char* base = lds + (lane * 16) + (threadIdx.x >> 6) * 4096;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds));
    // every workgroup streams its own 32 KiB per iteration, rows 1536 B apart like the shadow rows
    const unsigned long long wg_span = src_bytes >= (1ull << 28) ? 32768ull * 64 : src_bytes / 2;
    // (a power-of-two workgroup stride would put every workgroup on the same memory channel at the same time)
    const unsigned char* wg_src = src + (static_cast<unsigned long long>(blockIdx.x) * (src_bytes >= (1ull << 28) ? wg_span + 4352 + 65536 * 3 : 40192ull)) % (src_bytes - wg_span);
    const unsigned voff = (wid * 32 + (lane >> 2)) * 1536u + (lane & 3) * 16u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {  // blocked source: every piece is one contiguous 1 KiB (8 full cache lines)
            const unsigned st = lds0 + (it & 3) * 32768u + wid * 4096u;
            const unsigned char* bb = wg_src + (static_cast<unsigned long long>(it) * 32768ull) % (wg_span - 32768ull) + wid * 4096u;
            for (int p = 0; p < 4; ++p) lds_dma16_s(bb + p * 1024u, lane * 16u, st + p * 1024u);
        }
        if (MODE == 3) {
            const unsigned st = lds0 + (it & 3) * 32768u + wid * 4096u;
            const unsigned char* b = wg_src + (static_cast<unsigned long long>(it) * 64) % 1472;
            const unsigned char* bb = b + (static_cast<unsigned long long>(it >> 4) * 393216ull) % (wg_span - 393216ull - 1536);
            for (int p = 0; p < 4; ++p) lds_dma16_s(bb + p * 24576u, voff, st + p * 1024u);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i >> 2], fb[s][i & 3], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE >= 2 && i < 6) {
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(base + ((it * 2 + s) & 1) * 1024 + i * 16 * 64 % 4096);
                    if (i < 2) fa[s ^ 1][i] = v; else fb[s ^ 1][i - 2] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 1 && s == 0) {
                if (MODE >= 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) sum += acc[u][r];
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// Mode 5: the query operand never touches the LDS — its fragments are loaded straight from global
// memory (L2-resident [slab][256 queries][64 B] image) into a double-buffered register set, so an
// iteration carries 2 DMA pieces (rows) + 8 global_load_dwordx4 + 4 ds_read_b128 per wave.
__device__ __forceinline__ void gload16(bf16x8& dst, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__global__ __launch_bounds__(512, 2) void k5(float* out, int iters, unsigned long long* cyc, const unsigned char* src,
                                             unsigned long long src_bytes, const unsigned char* qsrc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    f32x16 acc[8];
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    bf16x8 fa[2][2], fq[2][2][4];
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; fa[s][j][i] = (__bf16)(((int)(x >> 8) & 0xffff) / 32768.0f - 1.0f); }
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
    const unsigned char* base = lds + (lane * 16) + (threadIdx.x >> 6) * 4096;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wid & 1;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds));
    const unsigned long long wg_span = src_bytes >= (1ull << 28) ? 32768ull * 64 : src_bytes / 2;
    const unsigned char* wg_src = src + (static_cast<unsigned long long>(blockIdx.x) * (src_bytes >= (1ull << 28) ? wg_span + 4352 + 65536 * 3 : 40192ull)) % (src_bytes - wg_span);
    const unsigned voff = (wid * 32 + (lane >> 2)) * 1536u + (lane & 3) * 16u;
    unsigned qoff[2][4];
    for (int s = 0; s < 2; ++s) for (int u = 0; u < 4; ++u) qoff[s][u] = (wc * 128 + u * 32 + l31) * 64u + (2 * s + h) * 16u;
    const unsigned char* qt = qsrc + (blockIdx.x & 3) * (24u * 16384u); // four query tiles
    auto loadq = [&](auto P, int slab) __attribute__((always_inline)) {
        const unsigned char* qb = qt + static_cast<unsigned>(slab % 24) * 16384u;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) gload16(fq[decltype(P)::value][s][u], qb, qoff[s][u]);
    };
    loadq(std::integral_constant<int, 0>{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    auto body = [&](auto P, int it) __attribute__((always_inline)) {
        constexpr int cur = decltype(P)::value, nxt = cur ^ 1;
        loadq(std::integral_constant<int, nxt>{}, it + 1);
        {
            const unsigned st = lds0 + (it & 3) * 16384u + wid * 2048u;
            const unsigned char* b = wg_src + (static_cast<unsigned long long>(it) * 64) % 1472;
            const unsigned char* bb = b + (static_cast<unsigned long long>(it >> 4) * 393216ull) % (wg_span - 393216ull - 1536);
            for (int p = 0; p < 2; ++p) lds_dma16_s(bb + p * 24576u, voff, st + p * 1024u);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i >> 2], fq[cur][s][i & 3], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 2) fa[s ^ 1][i] = *reinterpret_cast<const bf16x8*>(base + ((it * 2 + s) & 1) * 1024 + i * 16 * 64 % 4096);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s == 0) {
                asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        asm volatile("" : "+v"(fq[nxt][0][0]), "+v"(fq[nxt][0][1]), "+v"(fq[nxt][0][2]), "+v"(fq[nxt][0][3]),
                          "+v"(fq[nxt][1][0]), "+v"(fq[nxt][1][1]), "+v"(fq[nxt][1][2]), "+v"(fq[nxt][1][3]));
    };
    for (int it = 0; it < iters; it += 2) {
        body(std::integral_constant<int, 0>{}, it);
        body(std::integral_constant<int, 1>{}, it + 1);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) sum += acc[u][r];
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
void run5(int iters, const char* tag, unsigned long long src_bytes) {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    unsigned char *src, *q; (void)hipMalloc(&src, src_bytes + (1u << 20)); (void)hipMemset(src, 0x3c, src_bytes + (1u << 20));
    (void)hipMalloc(&q, 4u * 24u * 16384u); (void)hipMemset(q, 0x3c, 4u * 24u * 16384u);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k5, dim3(256), dim3(512), 0, 0, out, 1000, cyc, src, src_bytes, q); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k5, dim3(256), dim3(512), 0, 0, out, iters, cyc, src, src_bytes, q); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipFree(src); (void)hipFree(q);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = double(iters) * 16 * 2;
    printf("%s: %.2f ms, %.2f ns/MFMA/SIMD, %.0f TF/s\n", tag, ms, ms * 1e6 / n, 256.0 * 8 * double(iters) * 16 * 32768.0 / (ms * 1e-3) / 1e12);
}

// Modes 6/7: the product's DMA pattern.  Rows come from a 24 GiB image (first touch from HBM, shared
// by the four sibling workgroups of a row tile through their XCD's L2), queries from an L2-resident
// [slab][256][64 B] image; random bf16 everywhere, so the fragments multiplied are realistic.
// BLOCKED = the row image is stored [row/16][slab][16 rows][64 B] (a piece = one contiguous KiB)
// instead of row-major (a piece = 16 rows x 64 B, 1536 B apart).
__global__ void fill_bf16(unsigned short* p, unsigned long long n) {
    unsigned long long i = (blockIdx.x * 256ull + threadIdx.x) * 8ull;
    unsigned x = static_cast<unsigned>(i * 2654435761ull) ^ 0x9e3779b9u;
    for (int j = 0; j < 8 && i + j < n; ++j) {
        x = x * 1664525u + 1013904223u;
        const float f = ((x >> 8) & 0xffff) / 32768.0f - 1.0f;
        p[i + j] = static_cast<unsigned short>(__float_as_uint(f) >> 16);
    }
}
// NODMA: after four warm-up slabs (the ring holds random data) no further DMA is issued — what the
// loop costs with the same operand bits but no data movement.
template <bool BLOCKED, bool NODMA = false>
__global__ __launch_bounds__(512, 2) void k67(float* out, int iters, unsigned long long* cyc, const unsigned char* rows,
                                              unsigned long long n_tiles, const unsigned char* qimg) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
    f32x16 acc[8];
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    bf16x8 fa[2][2], fb[2][4];
    for (int s = 0; s < 2; ++s) { for (int j = 0; j < 2; ++j) fa[s][j] = bf16x8{}; for (int j = 0; j < 4; ++j) fb[s][j] = bf16x8{}; }
    for (int i = threadIdx.x; i < 131072 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds));
    const unsigned b = blockIdx.x, grp = (b & 7u) + 8u * (b >> 5), qt = (b >> 3) & 3u;
    int offA[2][2], offB[4][2];
    for (int rb = 0; rb < 2; ++rb) { const int rf = wr * 64 + rb * 32 + l31, f = (rf >> 2) & 3;
        for (int t = 0; t < 2; ++t) offA[rb][t] = rf * 64 + (((2 * t + h) ^ f) << 4); }
    for (int u = 0; u < 4; ++u) { const int rq = wc * 128 + u * 32 + l31, f = (rq >> 2) & 3;
        for (int t = 0; t < 2; ++t) offB[u][t] = 16384 + rq * 64 + (((2 * t + h) ^ f) << 4); }
    const unsigned vo_rows = BLOCKED ? lane * 16u : ((lane >> 2) * 1536u + (lane & 3) * 16u);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int slab = 0; unsigned long long tile = grp;
    for (int it = 0; it < iters; ++it) {
        const unsigned st = lds0 + (it & 3) * 32768u + wid * 2048u;
        const unsigned char* tb = rows + (tile % n_tiles) * 393216ull;
        if (!NODMA || it < 4) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned blk = wid * 2 + p;
                lds_dma16_s(BLOCKED ? tb + blk * 24576u + slab * 1024u : tb + blk * 24576u + slab * 64u, vo_rows, st + p * 1024u);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
                lds_dma16_s(qimg + qt * 393216u + slab * 16384u + (wid * 2 + p) * 1024u, lane * 16u, st + 16384u + p * 1024u);
        }
        if (++slab == 24) { slab = 0; tile += 64; }
        const unsigned char* rbase = lds + ((it + 2) & 3) * 32768;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i >> 2], fb[s][i & 3], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 2) fa[s ^ 1][i] = *reinterpret_cast<const bf16x8*>(rbase + offA[i][s ^ 1]);
                else if (i < 6) fb[s ^ 1][i - 2] = *reinterpret_cast<const bf16x8*>(rbase + offB[i - 2][s ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s == 0) {
                if (NODMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) sum += acc[u][r];
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <bool BLOCKED, bool NODMA = false> void run67(int iters, const char* tag, unsigned long long n_tiles = 48000) {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    // n_tiles = 48000: 18.9 GB of rows, every tile a first touch; n_tiles = 64: 25 MB, L2 / MALL resident
    unsigned char *rows, *q; (void)hipMalloc(&rows, n_tiles * 393216ull); (void)hipMalloc(&q, 4u * 393216u);
    hipLaunchKernelGGL(fill_bf16, dim3(static_cast<unsigned>(n_tiles * 393216ull / 2 / 8 / 256)), dim3(256), 0, 0, reinterpret_cast<unsigned short*>(rows), n_tiles * 393216ull / 2);
    hipLaunchKernelGGL(fill_bf16, dim3(4u * 393216u / 2 / 8 / 256), dim3(256), 0, 0, reinterpret_cast<unsigned short*>(q), 4ull * 393216u / 2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k67<BLOCKED, NODMA>), dim3(256), dim3(512), 0, 0, out, 480, cyc, rows, n_tiles, q); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k67<BLOCKED, NODMA>), dim3(256), dim3(512), 0, 0, out, iters, cyc, rows, n_tiles, q); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipFree(rows); (void)hipFree(q);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.2f ms, %.0f shader cycles per slab iteration (%.2f GHz), %.0f TF/s\n", tag, ms, double(c) / iters,
           double(c) / (ms * 1e6), 256.0 * 8 * double(iters) * 16 * 32768.0 / (ms * 1e-3) / 1e12);
}
template <int MODE> void run(int iters, const char* tag, unsigned long long src_bytes = 1ull << 26) {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    unsigned char* src; (void)hipMalloc(&src, src_bytes + (1u << 20)); (void)hipMemset(src, 0x3c, src_bytes + (1u << 20));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 1000, cyc, src, src_bytes); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, cyc, src, src_bytes); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipFree(src);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = double(iters) * 16 * 2; // MFMAs per SIMD (two waves)
    printf("%s: %.2f ms, %.1f shader cycles per MFMA per SIMD, %.2f ns/MFMA/SIMD, %.0f TF/s\n", tag, ms, double(c) / n, ms * 1e6 / n,
           256.0 * 8 * double(iters) * 16 * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
    run<0>(100000, "bare 2x8 MFMA bursts      ");
    run<1>(100000, "+ s_barrier per iteration ");
    run<2>(100000, "+ barrier + 12 ds_read    ");
    run<3>(100000, "+ 4 DMA pieces, 2 MiB src ", 2ull << 20);
    run<3>(100000, "+ 4 DMA pieces, 64 MiB src", 64ull << 20);
    run<3>(100000, "+ 4 DMA pieces, 24 GiB src", 24ull << 30);
    run<4>(100000, "+ 4 contiguous pieces 2MiB", 2ull << 20);
    run<4>(100000, "+ 4 contiguous pieces 64M ", 64ull << 20);
    run<4>(100000, "+ 4 contiguous pieces 24G ", 24ull << 30);
    run67<false, true>(17160, "product loop, random operands, NO DMA   ");
    run67<false>(17160, "product pattern, rows L2/MALL-resident  ", 64);
    run67<false>(17160, "product pattern, rows from HBM          ");
    run67<true>(17160, "product pattern, blocked rows from HBM  ");
    run67<false, true>(17160, "product loop, random operands, NO DMA   ");
    run67<false>(17160, "product pattern, rows L2/MALL-resident  ", 64);
    run67<false>(17160, "product pattern, rows from HBM          ");
    if (0) run5(100000, "2 DMA + 8 global q-frag loads, 2 MiB ", 2ull << 20);

    return 0;
}
