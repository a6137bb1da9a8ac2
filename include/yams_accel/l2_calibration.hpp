// l2_calibration.hpp — which arithmetic does the HOST's vec0 L2 distance use?  Asked of the host's own function.
//
// The vec0 engine's distance lives in third_party/sqlite-vec-cpp, which is ABSENT from the reference checkout
// (.gitmodules:4-6; caller: SqliteVecBackend::Impl::vec0SearchUnlocked, src/vector/sqlite_vec_backend.cpp:4450-4530),
// so this library cannot pin it and serves every definition that dependency can plausibly have
// (YAMS_SCAN_FLAG_L2_ACC_* in yams_mi355x_accel.h): fp64 sequential, fp32 sequential, fp32 in 8 or 16 round-robin
// lanes — and each fp32 form with its squares accumulated by a FUSED multiply-add, which is what `sum += d * d` and
// _mm256_add_ps(sum, _mm256_mul_ps(d, d)) become under -mfma: the reference's own build passes '-mavx', '-mfma' to that
// dependency on x86 (src/vector/meson.build:80-88), so the fused forms are the likely ones there.  The definitions agree to
// a few 1e-7 relative, yet on ~0.5 % of 1024-query batches over 10M rows a top-100 SET differs at the cut — so the host
// must not guess.  calibrateL2() runs the host's distance function
// (sqlite3_vec_distance_l2, tests/unit/vector/sqlite_vec_c_api_smoke_catch2_test.cpp:11-43 shows its C signature; or
// sqlite_vec_cpp::distances::l2 wrapped in a lambda) on crafted vector pairs OF THE INDEX'S OWN DIMENSION (a build that sums a tail of dim % 16
// elements its own way is no concern of a 768-wide index) whose fp32 result differs between the definitions, and reports
// which definition reproduces every probe bit for bit.  One (or several that coincide at this dimension): use it.  None:
// the host's build does something else (another lane count, pairwise sums, fp16 partials) and L2 must be REFUSED
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
    F32Fma = YAMS_SCAN_FLAG_L2_ACC_F32 | YAMS_SCAN_FLAG_L2_ACC_FUSED,       // as F32 / F32x8 / F32x16 with p = fma(d, d, p)
    F32x8Fma = YAMS_SCAN_FLAG_L2_ACC_F32X8 | YAMS_SCAN_FLAG_L2_ACC_FUSED,
    F32x16Fma = YAMS_SCAN_FLAG_L2_ACC_F32X16 | YAMS_SCAN_FLAG_L2_ACC_FUSED,
};
inline constexpr size_t kNumDefinitions = 7;
inline constexpr std::array<L2Accumulate, kNumDefinitions> kDefinitions{L2Accumulate::F64,      L2Accumulate::F32,      L2Accumulate::F32x8,
                                                                       L2Accumulate::F32x16,   L2Accumulate::F32Fma,   L2Accumulate::F32x8Fma,
                                                                       L2Accumulate::F32x16Fma};
inline const char* name(L2Accumulate d) {
    switch (d) {
        case L2Accumulate::F64: return "f64";
        case L2Accumulate::F32: return "f32";
        case L2Accumulate::F32x8: return "f32x8";
        case L2Accumulate::F32x16: return "f32x16";
        case L2Accumulate::F32Fma: return "f32_fma";
        case L2Accumulate::F32x8Fma: return "f32x8_fma";
        default: return "f32x16_fma";
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
    const uint32_t bits = static_cast<uint32_t>(def);
    const bool fused = (bits & YAMS_SCAN_FLAG_L2_ACC_FUSED) != 0;
    const uint32_t width = bits & YAMS_SCAN_FLAG_L2_ACC_MASK;
    const size_t lanes = width == YAMS_SCAN_FLAG_L2_ACC_F32 ? 1 : (width == YAMS_SCAN_FLAG_L2_ACC_F32X8 ? 8 : 16);
    volatile float part[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < dim; ++i) {
        volatile float d = a[i] - b[i];
        if (fused) { // one rounding: std::fmaf is the correctly rounded fused operation with or without the instruction
            const float dd = d;
            part[i % lanes] = std::fmaf(dd, dd, part[i % lanes]);
        } else {
            volatile float sq = d * d;
            part[i % lanes] = part[i % lanes] + sq;
        }
    }
    volatile float acc = 0.0f;
    for (size_t l = 0; l < lanes; ++l) acc = acc + part[l];
    return std::sqrt(static_cast<float>(acc));
}

struct Probe { std::vector<float> a, b; };

// The probes: kProbes vector pairs of the given dimension, deterministic (a 64-bit LCG, no <random>: the same bits on
// every standard library), magnitudes spread over six binades so that partial sums of different shapes round differently.
inline constexpr size_t kProbes = 128;
inline std::vector<Probe> probes(size_t dim) {
    std::vector<Probe> v;
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (static_cast<uint64_t>(dim) * 0xD1B54A32D192ED03ull);
    auto next = [&s] { s = s * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(s >> 33); };
    for (size_t rep = 0; rep < kProbes; ++rep) {
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
}

// pairwise: how many probes of this dimension give different fp32 distances under definitions x and y (0: the two
// coincide at this dimension — e.g. 8 and 16 lanes for dim <= 8 — and either serves)
inline size_t separating(L2Accumulate x, L2Accumulate y, size_t dim) {
    size_t n = 0;
    for (const Probe& p : probes(dim)) {
        const float dx = distance(x, p.a.data(), p.b.data(), p.a.size()), dy = distance(y, p.a.data(), p.b.data(), p.a.size());
        uint32_t bx, by;
        static_assert(sizeof bx == sizeof dx, "fp32");
        std::memcpy(&bx, &dx, 4); std::memcpy(&by, &dy, 4);
        n += bx != by;
    }
    return n;
}
// every pair of definitions is either told apart by at least `minSeparating` probes or not at all (a property the tests
// assert for the dimensions embeddings have; calibration itself only needs ONE separating probe per distinct pair)
inline bool distinguishing(size_t dim, size_t minSeparating = 3) {
    for (size_t i = 0; i < kDefinitions.size(); ++i)
        for (size_t j = i + 1; j < kDefinitions.size(); ++j) {
            const size_t n = separating(kDefinitions[i], kDefinitions[j], dim);
            if (n != 0 && n < minSeparating) return false;
        }
    return true;
}

struct L2Calibration {
    bool matched = false;                 // a definition reproduced every probe bit for bit (several only if they coincide here)
    L2Accumulate accumulate = L2Accumulate::F64;
    uint32_t flags = 0;                   // YAMS_SCAN_FLAG_L2_ACC_* | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT, for search_batch_ex
    size_t dim = 0, probes = 0, failedCalls = 0;
    std::array<size_t, kNumDefinitions> agreed{}; // per definition (kDefinitions order): probes it reproduced
    std::string detail;                   // one line for the log
};

// `dim`: the dimension of the vectors the index holds (the probes have it).
inline L2Calibration calibrateL2(const L2DistanceFn& fn, size_t dim) {
    L2Calibration c;
    c.dim = dim;
    if (!fn) { c.detail = "no distance function given"; return c; }
    if (dim == 0) { c.detail = "dimension 0"; return c; }
    const auto P = probes(dim);
    c.probes = P.size();
    // (two definitions that are different functions at this dimension differ on >= 5 % of such probes: the chance that 128 of
    //  them separate a pair nowhere is below 1e-3 of 1e-3; pairs that coincide as functions — tiny dimensions — never separate)
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
    // the first definition that reproduced everything; any other that did so coincides with it on every probe of this
    // dimension (the probes separate what can be separated), so the choice among them changes no result
    size_t which = kDefinitions.size();
    for (size_t d = 0; d < kDefinitions.size(); ++d)
        if (c.failedCalls == 0 && c.agreed[d] == c.probes) { which = d; break; }
    c.matched = which != kDefinitions.size();
    if (c.matched) {
        c.accumulate = kDefinitions[which];
        c.flags = static_cast<uint32_t>(c.accumulate) | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT;
    }
    c.detail = std::string(c.matched ? "host L2 at dim " : "host L2 matches NO served definition at dim ") + std::to_string(dim) +
               (c.matched ? std::string(" = ") + name(c.accumulate) : std::string()) + " (probes reproduced:";
    for (size_t d = 0; d < kDefinitions.size(); ++d) c.detail += std::string(" ") + name(kDefinitions[d]) + " " + std::to_string(c.agreed[d]);
    c.detail += " of " + std::to_string(c.probes) + (c.failedCalls ? ", failed calls " + std::to_string(c.failedCalls) : std::string()) + ")";
    return c;
}

} // namespace yams::vector::accel_l2
