// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 issue rate with 1 or 2 waves per SIMD,
// independent accumulators, no memory traffic.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[NACC];
    for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    bf16x8 a, b;
    // RANDOM != 0: operands with full-entropy mantissas (data-dependent power -> sustained clock)
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i) {
        if (iters < 0) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
        else {
            x = x * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(x >> 8) & 0xffff) / 32768.0f - 1.0f);
            x = x * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(x >> 8) & 0xffff) / 32768.0f - 1.0f);
        }
    }
    if (iters < 0) iters = -iters;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
void run(int threads, int iters, const char* tag) {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 1000, cyc); hipDeviceSynchronize();
    const int it_arg = iters; hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, it_arg, cyc); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (iters < 0) iters = -iters;
    const double n_mfma_per_simd = double(iters) * NACC * (threads / 256);
    const double flops = 256.0 * (threads / 64) * double(iters) * NACC * 32768.0;
    printf("%s threads=%d nacc=%d: %.2f ms, %.0f TF/s, %.1f memtime-ticks per MFMA per SIMD (tick=100MHz?), ns/MFMA/SIMD %.2f\n", tag, threads, NACC, ms,
           flops / (ms * 1e-3) / 1e12, double(c) / n_mfma_per_simd, ms * 1e6 / n_mfma_per_simd);
}
int main() {
    run<8>(512, -200000, "2 waves/SIMD, tiny operands");
    run<8>(512, 200000, "2 waves/SIMD, random operands");
    run<8>(256, 200000, "1 wave/SIMD, random operands");
    return 0;
}
