// Micro-benchmark: the MFMA skeleton of the filter loop — 8 waves per workgroup, per iteration two
// bursts of 8 MFMAs on 8 accumulators with 6 operand registers per burst, optional s_barrier between
// bursts, optional LDS fragment reads.  Reports cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
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
                if (MODE == 2 && i < 6) {
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
    return 0;
}
