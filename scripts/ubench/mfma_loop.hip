// Micro-benchmark: the MFMA skeleton of the filter loop — 8 waves per workgroup, per iteration two
// bursts of 8 MFMAs on 8 accumulators with 6 operand registers per burst, optional s_barrier between
// bursts, optional LDS fragment reads.  Reports cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int MODE> // 0 bare, 1 + s_barrier per burst pair, 2 + barrier + 12 ds_read_b128 per iteration
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
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
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
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
            if (MODE >= 1 && s == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) sum += acc[u][r];
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(int iters, const char* tag) {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 1000, cyc); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, cyc); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
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
    return 0;
}
