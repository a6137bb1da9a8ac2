// event_hop.hip — what a cross-stream dependency costs on this device: kernel A on stream 1, event, stream 2 waits for it,
// kernel B; the time between A's last instruction and B's first (wall_clock64 inside the kernels), against A and B
// back to back on ONE stream.  Event flavours: default, hipEventDisableTiming, hipEventDisableTiming | hipEventReleaseToDevice.
//   hipcc --offload-arch=gfx950 -O3 -o event_hop scripts/ubench/event_hop.hip && ./event_hop
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void stamp_kernel(unsigned long long* out, int spin) {
    unsigned long long t0 = wall_clock64();
    if (spin) { while (wall_clock64() - t0 < static_cast<unsigned long long>(spin)) {} }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long* d; CK(hipMalloc(&d, 64)); unsigned long long h[4];
    int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double us_per_tick = 1e3 / rate_khz;
    const int spin = static_cast<int>(200.0 / us_per_tick); // A runs 200 us: B's launch is long enqueued when A ends
    struct V { const char* name; unsigned flags; int mode; };
    const V vs[] = {{"same stream", 0, 0}, {"event default", hipEventDefault, 1}, {"event disable_timing", hipEventDisableTiming, 1},
                    {"event disable_timing|release_to_device", hipEventDisableTiming | hipEventReleaseToDevice, 1}};
    std::printf("{\"wall_clock_khz\": %d, \"grid\": \"A: 256 x 256 threads spinning 200 us, B: 256 x 256\", \"hops_us\": {", rate_khz);
    bool first = true;
    for (const V& v : vs) {
        hipEvent_t ev = nullptr;
        if (v.mode) CK(hipEventCreateWithFlags(&ev, v.flags));
        std::vector<double> gaps;
        for (int rep = 0; rep < 40; ++rep) {
            hipLaunchKernelGGL(stamp_kernel, dim3(256), dim3(256), 0, s1, d, spin);
            if (v.mode) { CK(hipEventRecord(ev, s1)); CK(hipStreamWaitEvent(s2, ev, 0)); hipLaunchKernelGGL(stamp_kernel, dim3(256), dim3(256), 0, s2, d + 2, 0); }
            else hipLaunchKernelGGL(stamp_kernel, dim3(256), dim3(256), 0, s1, d + 2, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
            if (rep >= 5) gaps.push_back((static_cast<double>(h[2]) - static_cast<double>(h[1])) * us_per_tick);
        }
        std::sort(gaps.begin(), gaps.end());
        std::printf("%s\"%s\": {\"median\": %.1f, \"min\": %.1f, \"p90\": %.1f}", first ? "" : ", ", v.name, gaps[gaps.size() / 2], gaps.front(), gaps[gaps.size() * 9 / 10]);
        first = false;
        if (ev) CK(hipEventDestroy(ev));
    }
    std::printf("}}\n");
    return 0;
}
