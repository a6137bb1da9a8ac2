// l2_calibration.hpp — which arithmetic does the HOST's vec0 L2 distance use?  Asked of the host's own function.
//
// The vec0 engine's distance lives in third_party/sqlite-vec-cpp, which is ABSENT from the reference checkout
// (.gitmodules:4-6; caller: SqliteVecBackend::Impl::vec0SearchUnlocked, src/vector/sqlite_vec_backend.cpp:4450-4530),
// so this library cannot pin it and serves every definition that dependency can plausibly have
// (YAMS_SCAN_FLAG_L2_ACC_* in yams_mi355x_accel.h): fp64 sequential, fp32 sequential, fp32 in 8 or 16 round-robin
// lanes.  The definitions agree to a few 1e-7 relative, yet on ~0.5 % of 1024-query batches over 10M rows a top-100
// SET differs at the cut — so the host must not guess.  calibrateL2() runs the host's distance function
// (sqlite3_vec_distance_l2, tests/unit/vector/sqlite_vec_c_api_smoke_catch2_test.cpp:11-43 shows its C signature; or
// sqlite_vec_cpp::distances::l2 wrapped in a lambda) on crafted vector pairs whose fp32 result is DIFFERENT under each
// definition and reports which definition reproduces every probe bit for bit.  Exactly one: use it.  None: the host's
// build does something else (FMA contraction, another lane count, pairwise sums) and L2 must be REFUSED
// (ErrorCode::NotSupported) rather than served with a top-k set that differs from the host's own — see
// AccelVectorIndex::calibrateL2 / AccelExactScanBackend::calibrateL2.
//
// Header-only, std-only, no device: calibration is host arithmetic.  The definitions below are the ones the device
// kernels implement (scan_kernels.hip, L2Sum); the test suite checks them against an independent C restatement.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../yams_mi355x_accel.h"

namespace yams::vector::accel_l2 {

enum class L2Accumulate : uint32_t {
    F64 = YAMS_SCAN_FLAG_L2_ACC_F64,       // sum (a_i - b_i)^2 in fp64, sequentially; sqrt; round to fp32
    F32 = YAMS_SCAN_FLAG_L2_ACC_F32,       // the same in fp32, sequentially; sqrtf
    F32x8 = YAMS_SCAN_FLAG_L2_ACC_F32X8,   // fp32, element i into partial sum i % 8, partial sums added left to right; sqrtf
    F32x16 = YAMS_SCAN_FLAG_L2_ACC_F32X16, // the same with 16 partial sums
};
inline constexpr std::array<L2Accumulate, 4> kDefinitions{L2Accumulate::F64, L2Accumulate::F32, L2Accumulate::F32x8, L2Accumulate::F32x16};
inline const char* name(L2Accumulate d) {
    switch (d) {
        case L2Accumulate::F64: return "f64";
        case L2Accumulate::F32: return "f32";
        case L2Accumulate::F32x8: return "f32x8";
        default: return "f32x16";
    }
}

// The host's distance: true and *out = the fp32 distance, or false when the call failed.
using L2DistanceFn = std::function<bool(const float* a, const float* b, size_t dim, float* out)>;

// sqlite-vec-cpp's C API: int sqlite3_vec_distance_l2(const void*, size_t bytes, const void*, size_t bytes, float*)
inline L2DistanceFn fromCApi(int (*fn)(const void*, size_t, const void*, size_t, float*)) {
    return [fn](const float* a, const float* b, size_t dim, float* out) { return fn && fn(a, dim * sizeof(float), b, dim * sizeof(float), out) == 0; };
}

// One definition, on the host.  volatile keeps a compiler with -ffp-contract=fast / -ffast-math from fusing or
// re-associating what is being DEFINED here (each product and each sum rounds on its own).
inline float distance(L2Accumulate def, const float* a, const float* b, size_t dim) {
    if (def == L2Accumulate::F64) {
        volatile double acc = 0.0;
        for (size_t i = 0; i < dim; ++i) {
            const double d = static_cast<double>(a[i]) - static_cast<double>(b[i]);
            volatile double sq = d * d;
            acc = acc + sq;
        }
        return static_cast<float>(std::sqrt(static_cast<double>(acc)));
    }
    const size_t lanes = def == L2Accumulate::F32 ? 1 : (def == L2Accumulate::F32x8 ? 8 : 16);
    volatile float part[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < dim; ++i) {
        volatile float d = a[i] - b[i];
        volatile float sq = d * d;
        part[i % lanes] = part[i % lanes] + sq;
    }
    volatile float acc = 0.0f;
    for (size_t l = 0; l < lanes; ++l) acc = acc + part[l];
    return std::sqrt(static_cast<float>(acc));
}

struct Probe { std::vector<float> a, b; };

// The probes: deterministic (a 64-bit LCG, no <random>: the same bits on every standard library), dimensions that are
// and are not multiples of the lane counts, magnitudes spread over six binades so that partial sums of different shapes
// round differently.  Built once; `distinguishing()` checks what the scheme relies on — every pair of definitions is
// told apart by at least `minSeparating` probes.
inline const std::vector<Probe>& probes() {
    static const std::vector<Probe> P = [] {
        std::vector<Probe> v;
        uint64_t s = 0x9E3779B97F4A7C15ull;
        auto next = [&s] { s = s * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(s >> 33); };
        const size_t dims[] = {768, 384, 1024, 100, 257, 33, 1536, 64};
        for (size_t rep = 0; rep < 6; ++rep)
            for (size_t dim : dims) {
                Probe p;
                p.a.resize(dim); p.b.resize(dim);
                for (size_t i = 0; i < dim; ++i) {
                    const float ua = static_cast<float>(next() & 0xFFFFFF) / 16777216.0f - 0.5f;
                    const float ub = static_cast<float>(next() & 0xFFFFFF) / 16777216.0f - 0.5f;
                    const float scale = std::ldexp(1.0f, static_cast<int>(next() % 6) - 3);
                    p.a[i] = ua * scale;
                    p.b[i] = ub * scale * 0.75f;
                }
                v.push_back(std::move(p));
            }
        return v;
    }();
    return P;
}

// pairwise: how many probes give different fp32 distances under definitions x and y
inline size_t separating(L2Accumulate x, L2Accumulate y) {
    size_t n = 0;
    for (const Probe& p : probes()) {
        const float dx = distance(x, p.a.data(), p.b.data(), p.a.size()), dy = distance(y, p.a.data(), p.b.data(), p.a.size());
        uint32_t bx, by;
        static_assert(sizeof bx == sizeof dx, "fp32");
        std::memcpy(&bx, &dx, 4); std::memcpy(&by, &dy, 4);
        n += bx != by;
    }
    return n;
}
inline bool distinguishing(size_t minSeparating = 4) {
    for (size_t i = 0; i < kDefinitions.size(); ++i)
        for (size_t j = i + 1; j < kDefinitions.size(); ++j)
            if (separating(kDefinitions[i], kDefinitions[j]) < minSeparating) return false;
    return true;
}

struct L2Calibration {
    bool matched = false;                 // exactly one definition reproduced every probe bit for bit
    L2Accumulate accumulate = L2Accumulate::F64;
    uint32_t flags = 0;                   // YAMS_SCAN_FLAG_L2_ACC_* | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT, for search_batch_ex
    size_t probes = 0, failedCalls = 0;
    std::array<size_t, 4> agreed{};       // per definition (kDefinitions order): probes it reproduced
    std::string detail;                   // one line for the log
};

inline L2Calibration calibrateL2(const L2DistanceFn& fn) {
    L2Calibration c;
    const auto& P = probes();
    c.probes = P.size();
    if (!fn) { c.detail = "no distance function given"; return c; }
    if (!distinguishing()) { c.detail = "the probes do not separate the definitions on this host (should not happen)"; return c; }
    for (const Probe& p : P) {
        float got = 0.0f;
        if (!fn(p.a.data(), p.b.data(), p.a.size(), &got)) { ++c.failedCalls; continue; }
        uint32_t gb;
        std::memcpy(&gb, &got, 4);
        for (size_t d = 0; d < kDefinitions.size(); ++d) {
            const float want = distance(kDefinitions[d], p.a.data(), p.b.data(), p.a.size());
            uint32_t wb;
            std::memcpy(&wb, &want, 4);
            c.agreed[d] += wb == gb;
        }
    }
    size_t full = 0, which = 0;
    for (size_t d = 0; d < kDefinitions.size(); ++d)
        if (c.failedCalls == 0 && c.agreed[d] == c.probes) { ++full; which = d; }
    c.matched = full == 1;
    if (c.matched) {
        c.accumulate = kDefinitions[which];
        c.flags = static_cast<uint32_t>(c.accumulate) | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT;
    }
    c.detail = std::string(c.matched ? "host L2 = " : "host L2 matches NO served definition: ") + (c.matched ? name(c.accumulate) : "") +
               " (probes reproduced: f64 " + std::to_string(c.agreed[0]) + ", f32 " + std::to_string(c.agreed[1]) + ", f32x8 " +
               std::to_string(c.agreed[2]) + ", f32x16 " + std::to_string(c.agreed[3]) + " of " + std::to_string(c.probes) +
               (c.failedCalls ? ", failed calls " + std::to_string(c.failedCalls) : std::string()) + ")";
    return c;
}

} // namespace yams::vector::accel_l2
