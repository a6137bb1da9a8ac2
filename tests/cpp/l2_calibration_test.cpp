// l2_calibration_test.cpp — TEST: include/yams_accel/l2_calibration.hpp picks the host's vec0 L2 arithmetic by asking the
// host's own function.  The "host functions" here are the ORACLE's seven definitions (oracle/yams_oracle.c:
// oracle_exact_scan_l2's fp64 distance, oracle_l2_distance_f32acc with 1 / 8 / 16 lanes, and the same lanes with a fused
// multiply-add: lanes -1 / -8 / -16) — an independent restatement, in C, built with -ffp-contract=off —, a host that is
// literally the AVX loop of the public sqlite-vec compiled the way the reference compiles its dependency (-mavx -mfma),
// a host with fp64 partial sums (served by F64: any order of double accumulation rounds to the same float), plus hosts
// that match NO served definition (4 lanes, pairwise fp32 summation, a function that fails): each of the seven must be recognised as itself at dims 768 / 384 / 1024 / 100, each of the others refused; at a
// dimension where definitions coincide (dim 8: one element per lane) any of the coinciding ones is accepted.  No GPU involved: calibration is host arithmetic.  Linked against oracle/_build/libyams_oracle.so
// (tests may; the product never does).
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
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
static bool late_rounding(const float* a, const float* b, size_t dim, float* out) { // fp64 partials, 8 lanes, rounded once at the end
    double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < dim; ++i) { const double d = static_cast<double>(a[i]) - b[i]; part[i % 8] += d * d; }
    double s = 0;
    for (double p : part) s += p;
    *out = static_cast<float>(std::sqrt(s));
    return true;
}
// The public sqlite-vec's AVX loop shape (l2_sqr_float_avx: one 8-lane accumulator, sum = add(sum, mul(diff, diff)), the
// lanes added left to right), written with GCC vector types so that THIS translation unit's flags decide the arithmetic:
// the test is compiled with -mavx -mfma -ffp-contract=fast like the reference's dependency, so the compiler fuses it.
typedef float v8f __attribute__((vector_size(32)));
static bool avx_shape(const float* a, const float* b, size_t dim, float* out) {
    if (dim % 8) return false;
    v8f sum = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < dim; i += 8) {
        v8f x, y;
        std::memcpy(&x, a + i, 32); std::memcpy(&y, b + i, 32);
        const v8f d = x - y;
        sum = sum + d * d;
    }
    float r = 0.f;
    for (int l = 0; l < 8; ++l) r += sum[l];
    *out = std::sqrt(r);
    return true;
}

int main() {
    const size_t dims[] = {768, 384, 1024, 100};
    for (size_t dim : dims) {
        CHECK(l2::distinguishing(dim));
        size_t least = 1000;
        for (size_t i = 0; i < l2::kDefinitions.size(); ++i)
            for (size_t j = i + 1; j < l2::kDefinitions.size(); ++j) least = std::min(least, l2::separating(l2::kDefinitions[i], l2::kDefinitions[j], dim));
        std::printf("dim %zu: every two of the %zu definitions are separated by >= %zu of %zu probes\n", dim, l2::kDefinitions.size(), least, l2::kProbes);
        CHECK(least >= 3);
    }
    struct Host { const char* what; l2::L2DistanceFn fn; bool match; l2::L2Accumulate want; };
    const Host hosts[] = {
        {"oracle fp64", oracle_f64, true, l2::L2Accumulate::F64},
        {"oracle f32 sequential", oracle_f32(1), true, l2::L2Accumulate::F32},
        {"oracle f32 8 lanes", oracle_f32(8), true, l2::L2Accumulate::F32x8},
        {"oracle f32 16 lanes", oracle_f32(16), true, l2::L2Accumulate::F32x16},
        {"oracle f32 sequential, fused", oracle_f32(-1), true, l2::L2Accumulate::F32Fma},
        {"oracle f32 8 lanes, fused", oracle_f32(-8), true, l2::L2Accumulate::F32x8Fma},
        {"oracle f32 16 lanes, fused", oracle_f32(-16), true, l2::L2Accumulate::F32x16Fma},
        {"C API shaped, 8 lanes", l2::fromCApi(&c_api_f32x8), true, l2::L2Accumulate::F32x8},
        {"f32 4 lanes (not served)", oracle_f32(4), false, l2::L2Accumulate::F64},
        {"pairwise summation", pairwise, false, l2::L2Accumulate::F64},
        // fp64 partial sums in ANY association are F64 to this scheme: their differences (~1e-16 relative) vanish in the final
        // rounding to fp32 on every probe — a host that accumulates in double, however it orders the sum, is served by F64
        {"fp64 partials in 8 lanes", late_rounding, true, l2::L2Accumulate::F64},
        {"a function that fails", [](const float*, const float*, size_t, float*) { return false; }, false, l2::L2Accumulate::F64},
    };
    for (size_t dim : dims)
        for (const Host& h : hosts) {
            const l2::L2Calibration c = l2::calibrateL2(h.fn, dim);
            if (dim == 768) std::printf("%-32s -> %s\n", h.what, c.detail.c_str());
            CHECK(c.matched == h.match && c.dim == dim);
            if (h.match) {
                CHECK(c.accumulate == h.want);
                CHECK(c.flags == (static_cast<uint32_t>(h.want) | YAMS_SCAN_FLAG_L2_ACC_EXPLICIT));
            } else CHECK(c.flags == 0);
        }
    // the AVX loop shape under this translation unit's own flags (-mavx -mfma -ffp-contract=fast: as the reference's build)
    {
        const l2::L2Calibration c = l2::calibrateL2(avx_shape, 768);
        std::printf("%-32s -> %s\n", "AVX loop shape, this TU's flags", c.detail.c_str());
        CHECK(c.matched && (c.accumulate == l2::L2Accumulate::F32x8Fma || c.accumulate == l2::L2Accumulate::F32x8));
#if defined(__FMA__)
        CHECK(c.accumulate == l2::L2Accumulate::F32x8Fma); // what the reference's x86 build of its dependency would be served with
#endif
        CHECK(!l2::calibrateL2(avx_shape, 100).matched);   // (that host does not serve dim % 8 != 0 at all)
    }
    // where definitions coincide any of them serves: dim 8 puts one element in every lane of the 8- and 16-lane forms
    {
        CHECK(l2::separating(l2::L2Accumulate::F32x8, l2::L2Accumulate::F32x16, 8) == 0);
        const l2::L2Calibration c = l2::calibrateL2(oracle_f32(16), 8);
        // (and both equal the sequential sum there: 0 + p0 + p1 + ... — the first coinciding definition in the list is reported)
        CHECK(l2::separating(l2::L2Accumulate::F32, l2::L2Accumulate::F32x8, 8) == 0);
        CHECK(c.matched && (c.accumulate == l2::L2Accumulate::F32 || c.accumulate == l2::L2Accumulate::F32x8 || c.accumulate == l2::L2Accumulate::F32x16));
    }
    CHECK(!l2::calibrateL2(nullptr, 768).matched && !l2::calibrateL2(oracle_f64, 0).matched);
    std::printf(failures ? "FAILED (%d)\n" : "OK (0 failures)\n", failures);
    return failures ? 1 : 0;
}
