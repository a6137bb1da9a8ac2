#include <hip/hip_runtime.h>
// rate of v_mov_b32 vs v_mov_b64 vs v_pk_mov_b32: N moves per loop trip, 2 waves per SIMD
template <int MODE> __global__ __launch_bounds__(512) void k(long* o, int iters) {
    long v[16]; int w[32];
    long s = o[threadIdx.x]; int t = (int)s;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(t));
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" :: "v"(w[i]));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b64 %0, %1" : "=v"(v[i]) : "v"(s));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(v[i]));
        }
    }
    o[threadIdx.x + 512] = s;
}
int main() {
    long* d; hipMalloc(&d, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, 100000);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, 100000);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("mode %d: %.3f ms for 100000 x 32 dwords moved per lane -> %.2f clk per dword-move-instr-equivalent at 2.4GHz\n", mode, ms, ms * 1e-3 * 2.4e9 / (100000.0 * 32 * 2));
        }
    }
}
