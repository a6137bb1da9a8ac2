// standalone check of topk_block_kernel (register-blocked bitonic): random key lists of every size class against std::sort
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I yams_amd/csrc -I include scripts/dbg/topk_check.hip -o scripts/dbg/topk_check
#include "../../yams_amd/csrc/scan_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace yams_accel;
int main() {
    std::mt19937_64 rng(7);
    const uint32_t nq = 64, cap = 4096;
    int bad = 0;
    for (uint32_t keep : {11u, 97u, 385u, 2048u}) {
        std::vector<uint64_t> h(static_cast<size_t>(nq) * cap);
        std::vector<uint32_t> cnt(nq);
        for (uint32_t q = 0; q < nq; ++q) {
            const uint32_t sizes[] = {0, 1, 5, 255, 256, 257, 511, 512, 513, 1000, 1024, 1025, 2047, 2048, 2049, 3000, 4095, 4096, 5000};
            cnt[q] = sizes[q % 19];
            for (uint32_t i = 0; i < cap; ++i) h[static_cast<size_t>(q) * cap + i] = (rng() | 1ull) >> (q & 3);
        }
        uint64_t *d_in, *d_out; uint32_t* d_cnt;
        hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, static_cast<size_t>(nq) * keep * 8 * 2); hipMalloc(&d_cnt, nq * 4);
        hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_cnt, cnt.data(), nq * 4, hipMemcpyHostToDevice);
        const uint64_t* res; uint64_t stride;
        hipError_t e = launch_select_lists(nullptr, d_in, d_cnt, cap, nq, nullptr, keep, d_out, &res, &stride);
        hipDeviceSynchronize();
        std::vector<uint64_t> out(static_cast<size_t>(nq) * stride);
        hipMemcpy(out.data(), res, out.size() * 8, hipMemcpyDeviceToHost);
        for (uint32_t q = 0; q < nq; ++q) {
            const uint32_t n = std::min(cnt[q], cap);
            std::vector<uint64_t> ref(h.begin() + static_cast<size_t>(q) * cap, h.begin() + static_cast<size_t>(q) * cap + n);
            std::sort(ref.begin(), ref.end(), std::greater<uint64_t>());
            for (uint32_t i = 0; i < keep; ++i) {
                const uint64_t want = i < n ? ref[i] : 0, got = out[static_cast<size_t>(q) * stride + i];
                if (want != got) { if (bad < 10) std::printf("keep %u q %u n %u i %u: want %llx got %llx\n", keep, q, n, i, (unsigned long long)want, (unsigned long long)got); ++bad; }
            }
        }
        std::printf("keep %u: err %d (%s)\n", keep, (int)e, hipGetErrorString(e));
        hipFree(d_in); hipFree(d_out); hipFree(d_cnt);
    }
    std::printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad != 0;
}
