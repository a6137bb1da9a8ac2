// Probe: can a device mirror grow in place?  Reserve a large virtual range once, map physical chunks
// behind it as rows arrive (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess), run a
// kernel across chunk boundaries.   hipcc --offload-arch=gfx950 -O2 vmm_probe.hip -o vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)(i * 2654435761u); }
__global__ void check(const unsigned* p, size_t n, unsigned long long* bad) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n && p[i] != (unsigned)(i * 2654435761u)) atomicAdd(bad, 1ull); }
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity: recommended %zu, minimum %zu\n", gran, gmin);
    const size_t reserve = 64ull << 30, chunk = ((256ull << 20) + gran - 1) / gran * gran;
    void* va = nullptr; CK(hipMemAddressReserve(&va, reserve, 0, nullptr, 0));
    printf("reserved %zu GiB at %p\n", reserve >> 30, va);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long* bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    for (int c = 0; c < 3; ++c) {   // grow: map one more chunk, the data written before stays where it is
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char*)va + c * chunk, chunk, 0, h, 0));
        CK(hipMemSetAccess((char*)va + c * chunk, chunk, &acc, 1));
        hs.push_back(h);
        const size_t n0 = c * chunk / 4, n1 = (c + 1) * chunk / 4;
        fill<<<(unsigned)((n1 - n0 + 255) / 256), 256>>>((unsigned*)va + n0, 0), hipDeviceSynchronize();
        // fill the new chunk with values that continue the global index
        struct L { static __global__ void k(unsigned* p, size_t a, size_t b) { size_t i = a + blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < b) p[i] = (unsigned)(i * 2654435761u); } };
        hipLaunchKernelGGL(L::k, dim3((unsigned)((n1 - n0 + 255) / 256)), dim3(256), 0, 0, (unsigned*)va, n0, n1);
        CK(hipDeviceSynchronize());
        check<<<(unsigned)((n1 + 255) / 256), 256>>>((unsigned*)va, n1, bad);
        CK(hipDeviceSynchronize());
        unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        printf("after chunk %d: %zu MiB mapped, mismatches %llu\n", c, (size_t)((c + 1) * chunk >> 20), hb);
    }
    // host copies into the mapped range work like on any allocation
    std::vector<unsigned> hbuf(1 << 20, 7u);
    CK(hipMemcpy((char*)va + chunk - 2 * (1 << 20), hbuf.data(), 4 << 20, hipMemcpyHostToDevice)); // straddles two chunks
    unsigned back[4]; CK(hipMemcpy(back, (char*)va + chunk - 8, 16, hipMemcpyDeviceToHost));
    printf("straddling memcpy: %u %u %u %u\n", back[0], back[1], back[2], back[3]);
    for (size_t c = 0; c < hs.size(); ++c) { CK(hipMemUnmap((char*)va + c * chunk, chunk)); CK(hipMemRelease(hs[c])); }
    CK(hipMemAddressFree(va, reserve));
    printf("VMM OK\n");
    return 0;
}
