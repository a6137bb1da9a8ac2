// Probe: visibility of data written into HIP-VMM-mapped device memory, read by a kernel on ANOTHER stream.
//   mode 0: fresh mapping every iteration, hipMemcpyAsync(H2D) + stream sync, then the kernel
//   mode 1: ONE mapping reused, new contents every iteration via hipMemcpyAsync
//   mode 2: fresh mapping, contents written by a copy kernel from a hipMalloc staging buffer
//   mode 3: fresh mapping, hipMemcpyAsync + hipDeviceSynchronize
//   mode 4: fresh mapping, copy and kernel on the SAME stream
//   mode 5: fresh mapping, a hipMemsetAsync of the whole range first (touch), then as mode 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void sum(const unsigned* p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}
__global__ void copyk(unsigned* dst, const unsigned* src, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }
int run(int mode) {
    hipStream_t s_up, s_k; CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long* d_out; CK(hipMalloc(&d_out, 8));
    unsigned* stage; CK(hipMalloc(&stage, 1 << 20));
    const size_t bytes = 2ull << 20;
    void* va = nullptr; hipMemGenericAllocationHandle_t h{};
    int bad = 0;
    for (int it = 0; it < 300; ++it) {
        if (mode != 1 || it == 0) {
            CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
            CK(hipMemCreate(&h, bytes, &prop, 0));
            CK(hipMemMap(va, bytes, 0, h, 0)); CK(hipMemSetAccess(va, bytes, &acc, 1));
        }
        const size_t n = (it % 3 == 0) ? 3 + (it % 5) : 1000 + it * 37;
        std::vector<unsigned> host(n);
        unsigned long long want = 0;
        for (size_t i = 0; i < n; ++i) { host[i] = (unsigned)(it * 7 + i * 13 + 1); want += host[i]; }
        hipStream_t sk = mode == 4 ? s_up : s_k;
        if (mode == 5) { CK(hipMemsetAsync(va, 0xff, bytes, s_up)); }
        if (mode == 2) {
            CK(hipMemcpyAsync(stage, host.data(), n * 4, hipMemcpyHostToDevice, s_up));
            copyk<<<(unsigned)((n + 255) / 256), 256, 0, s_up>>>((unsigned*)va, stage, n);
        } else {
            CK(hipMemcpyAsync(va, host.data(), n * 4, hipMemcpyHostToDevice, s_up));
        }
        if (mode == 3) CK(hipDeviceSynchronize()); else CK(hipStreamSynchronize(s_up));
        CK(hipMemsetAsync(d_out, 0, 8, sk));
        sum<<<64, 256, 0, sk>>>((const unsigned*)va, n, d_out);
        unsigned long long got = 0; CK(hipMemcpyAsync(&got, d_out, 8, hipMemcpyDeviceToHost, sk)); CK(hipStreamSynchronize(sk));
        if (got != want) ++bad;
        if (mode != 1) { CK(hipMemUnmap(va, bytes)); CK(hipMemRelease(h)); CK(hipMemAddressFree(va, bytes)); }
    }
    printf("mode %d: wrong reads %d of 300\n", mode, bad);
    return 0;
}
int main() { CK(hipSetDevice(0)); for (int m = 0; m < 6; ++m) if (run(m)) return 1; return 0; }
