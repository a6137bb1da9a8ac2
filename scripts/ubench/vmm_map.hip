// Which hipMemMap / hipMemSetAccess shapes does the driver take?  (GrowBuf in plugin.cpp relies on the answer.)
//   variant 0: release the handle right after mapping, set access on the new chunk only
//   variant 1: keep every handle alive
//   variant 2: release, but set access on the whole mapped range each time
//   variant 3: keep handles + whole range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static int g_variant = 0;
static bool map_at(unsigned char* base, size_t off, size_t bytes) {
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, bytes, &prop, 0);
    if (e != hipSuccess) { printf("  create(%zu) -> %s\n", bytes, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
    e = hipMemMap(base + off, bytes, 0, h, 0);
    if (e != hipSuccess) { printf("  map(off %zu MiB, %zu) -> %s\n", off >> 20, bytes, hipGetErrorString(e)); (void)hipGetLastError(); (void)hipMemRelease(h); return false; }
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (g_variant & 2) e = hipMemSetAccess(base, off + bytes, &acc, 1);
    else e = hipMemSetAccess(base + off, bytes, &acc, 1);
    if (e != hipSuccess) {
        printf("  setaccess(off %zu MiB, %zu MiB) -> %s\n", off >> 20, bytes >> 20, hipGetErrorString(e)); (void)hipGetLastError();
        (void)hipMemUnmap(base + off, bytes); (void)hipMemRelease(h); return false;
    }
    if (!(g_variant & 1)) (void)hipMemRelease(h);
    return true;
}
int main() {
    for (g_variant = 0; g_variant < 4; ++g_variant) {
        for (int trial = 0; trial < 4; ++trial) {
            void* va = nullptr; size_t res = 16ull << 30;
            if (hipMemAddressReserve(&va, res, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
            unsigned char* b = (unsigned char*)va;
            size_t off = 0;
            std::vector<size_t> sizes;
            if (trial == 0) sizes = {32ull << 20, 32ull << 20, 32ull << 20, 148ull << 21, 2ull << 20, 6ull << 20, 1ull << 30};
            if (trial == 1) sizes = {196ull << 21, 148ull << 21, 64ull << 20, 3ull << 21};
            if (trial == 2) sizes = {2ull << 20, 2ull << 20, 148ull << 21, 512ull << 20, 147ull << 21};
            if (trial == 3) sizes = std::vector<size_t>(24, 32ull << 20);
            int ok = 0, bad = 0;
            for (size_t s : sizes) { if (map_at(b, off, s)) { off += s; ++ok; } else ++bad; }
            printf("variant %d trial %d base %p: %d ok %d failed\n", g_variant, trial, va, ok, bad);
        }
    }
    return 0;
}
