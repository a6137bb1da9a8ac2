// Micro-benchmark: sustained issue rate of the int8 MFMA shapes next to the bf16 one, 2 waves per
// SIMD, independent accumulators, no memory traffic, random operands.
//   hipcc --offload-arch=gfx950 -O3 mfma_i8_rate.hip -o mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ unsigned lcg(unsigned& x) { x = x * 1664525u + 1013904223u; return x; }

// SHAPE 0: v_mfma_f32_32x32x16_bf16 (32768 flop), 1: v_mfma_i32_32x32x32_i8 (65536 op),
//       2: v_mfma_i32_16x16x64_i8 (32768 op)
template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void k(int* out, int iters) {
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    if constexpr (SHAPE == 0) {
        f32x16 acc[NACC];
        for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(((int)(lcg(x) >> 8) & 0xffff) / 32768.0f - 1.0f); b[i] = (__bf16)(((int)(lcg(x) >> 8) & 0xffff) / 32768.0f - 1.0f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
        }
        float s = 0.f;
        for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
        if (s == 123.456f) out[0] = 1;
    } else if constexpr (SHAPE == 1) {
        i32x16 acc[NACC];
        for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0;
        i32x4 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (int)lcg(x); b[i] = (int)lcg(x); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u], 0, 0, 0);
        }
        int s = 0;
        for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
        if (s == 123456789) out[0] = 1;
    } else {
        i32x4 acc[NACC];
        for (int u = 0; u < NACC; ++u) for (int r = 0; r < 4; ++r) acc[u][r] = 0;
        i32x4 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (int)lcg(x); b[i] = (int)lcg(x); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[u], 0, 0, 0);
        }
        int s = 0;
        for (int u = 0; u < NACC; ++u) for (int r = 0; r < 4; ++r) s += acc[u][r];
        if (s == 123456789) out[0] = 1;
    }
}
template <int SHAPE, int NACC>
void run(int threads, int iters, double ops_per_mfma, const char* tag) {
    int* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, 1000); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_per_simd = double(iters) * NACC * (threads / 256);
    const double ops = 256.0 * (threads / 64) * double(iters) * NACC * ops_per_mfma;
    printf("%-34s threads=%d nacc=%d: %.2f ms, %.0f Top/s, ns per MFMA per SIMD %.2f (= %.1f cycles at 2.4 GHz)\n", tag, threads, NACC, ms,
           ops / (ms * 1e-3) / 1e12, ms * 1e6 / n_per_simd, ms * 1e6 / n_per_simd * 2.4);
}
int main() {
    run<0, 8>(512, 100000, 32768.0, "bf16 32x32x16");
    run<1, 8>(512, 100000, 65536.0, "i8 32x32x32");
    run<2, 8>(512, 100000, 32768.0, "i8 16x16x64 (8 acc)");
    run<2, 16>(512, 100000, 32768.0, "i8 16x16x64 (16 acc)");
    run<1, 8>(256, 100000, 65536.0, "i8 32x32x32, 1 wave/SIMD");
    return 0;
}
