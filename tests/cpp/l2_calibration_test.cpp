// l2_calibration_test.cpp — TEST: include/yams_accel/l2_calibration.hpp picks the host's vec0 L2 arithmetic by asking the
// host's own function.  The "host functions" here are the ORACLE's four definitions (oracle/yams_oracle.c:
// oracle_exact_scan_l2's fp64 distance, oracle_l2_distance_f32acc with 1 / 8 / 16 lanes) — an independent restatement, in
// C, built with -ffp-contract=off —, plus hosts that match NO served definition (4 lanes, pairwise summation, an
// FMA-style fused accumulate, a function that fails): each of the four must be recognised as itself, each of the others
// must be refused.  No GPU involved: calibration is host arithmetic.  Linked against oracle/_build/libyams_oracle.so
// (tests may; the product never does).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "yams_accel/l2_calibration.hpp"

extern "C" {
long oracle_exact_scan_l2(const float* corpus, size_t n_rows, size_t dim, const float* query, size_t k, float thr,
                          const uint64_t* tie_rank, int64_t* out_rows, float* out_dist, float* out_sims);
float oracle_l2_distance_f32acc(const float* a, const float* b, size_t dim, int lanes);
}

namespace l2 = yams::vector::accel_l2;
static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static bool oracle_f64(const float* a, const float* b, size_t dim, float* out) {
    int64_t row = -1; float dist = 0.f, sim = 0.f;
    if (oracle_exact_scan_l2(a, 1, dim, b, 1, -2.0f, nullptr, &row, &dist, &sim) != 1) return false;
    *out = dist;
    return true;
}
static l2::L2DistanceFn oracle_f32(int lanes) {
    return [lanes](const float* a, const float* b, size_t dim, float* out) { *out = oracle_l2_distance_f32acc(a, b, dim, lanes); return true; };
}
// a C-API shaped host function (sqlite3_vec_distance_l2's signature): sizes in BYTES, 0 on success
static int c_api_f32x8(const void* a, size_t na, const void* b, size_t nb, float* out) {
    if (na != nb) return 1;
    *out = oracle_l2_distance_f32acc(static_cast<const float*>(a), static_cast<const float*>(b), na / sizeof(float), 8);
    return 0;
}
static bool pairwise(const float* a, const float* b, size_t dim, float* out) {
    std::vector<float> v(dim);
    for (size_t i = 0; i < dim; ++i) { const float d = a[i] - b[i]; v[i] = d * d; }
    for (size_t n = dim; n > 1; n = (n + 1) / 2)
        for (size_t i = 0; i < n / 2; ++i) v[i] = v[i] + v[n - 1 - i];
    *out = std::sqrt(v[0]);
    return true;
}
static bool fused(const float* a, const float* b, size_t dim, float* out) { // what -ffp-contract=fast makes of the scalar loop
    float acc = 0.f;
    for (size_t i = 0; i < dim; ++i) { const float d = a[i] - b[i]; acc = std::fmaf(d, d, acc); }
    *out = std::sqrt(acc);
    return true;
}

int main() {
    CHECK(l2::distinguishing());
    for (size_t i = 0; i < l2::kDefinitions.size(); ++i)
        for (size_t j = i + 1; j < l2::kDefinitions.size(); ++j) {
            const size_t sep = l2::separating(l2::kDefinitions[i], l2::kDefinitions[j]);
            std::printf("probes separating %-6s from %-6s: %zu of %zu\n", l2::name(l2::kDefinitions[i]), l2::name(l2::kDefinitions[j]), sep,
                        l2::probes().size());
            CHECK(sep >= 4);
        }
    struct Host { const char* what; l2::L2DistanceFn fn; bool match; l2::L2Accumulate want; };
    const Host hosts[] = {
        {"oracle fp64", oracle_f64, true, l2::L2Accumulate::F64},
        {"oracle f32 sequential", oracle_f32(1), true, l2::L2Accumulate::F32},
        {"oracle f32 8 lanes", oracle_f32(8), true, l2::L2Accumulate::F32x8},
        {"oracle f32 16 lanes", oracle_f32(16), true, l2::L2Accumulate::F32x16},
        {"C API shaped, 8 lanes", l2::fromCApi(&c_api_f32x8), true, l2::L2Accumulate::F32x8},
        {"f32 4 lanes (not served)", oracle_f32(4), false, l2::L2Accumulate::F64},
        {"pairwise summation", pairwise, false, l2::L2Accumulate::F64},
        {"fused multiply-add accumulate", fused, false, l2::L2Accumulate::F64},
        {"a function that fails", [](const float*, const float*, size_t, float*) { return false; }, false, l2::L2Accumulate::F64},
    };
    for (const Host& h : hosts) {
        const l2::L2Calibration c = l2::calibrateL2(h.fn);
        std::printf("%-32s -> %s\n", h.what, c.detail.c_str());
        CHECK(c.matched == h.match);
        if (h.match) {
            CHECK(c.accumulate == h.want);
            CHECK(c.flags == (static_cast<uint32_t>(h.want) | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT));
        } else CHECK(c.flags == 0);
    }
    CHECK(!l2::calibrateL2(nullptr).matched);
    std::printf(failures ? "FAILED (%d)\n" : "OK (0 failures)\n", failures);
    return failures ? 1 : 0;
}
