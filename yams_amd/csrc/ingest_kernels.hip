// ingest_kernels.hip — gfx950 integer kernels of the content-ingest path (K5, K6 of SURVEY.md §2).
//
// Reference semantics (paths under /root/reference):
//   src/chunking/rabin_fingerprint_table.h:12-28   out-table
//   src/chunking/rabin_chunker.cpp:63-152          RabinChunker boundaries
//   include/yams/chunking/streaming_chunker.h:146-204 + src/chunking/streaming_chunker.cpp:37-69
//                                                  StreamingChunker boundaries (product default)
//   src/crypto/sha256_hasher.cpp:167-195           SHA-256 (OpenSSL EVP, FIPS 180-4)
//
// The reference's rolling hash  h' = ((h - out[old]) << 8) ^ out[new]  on a uint64_t loses all
// state after 8 steps (each step shifts it left by 8 bits; subtraction and xor only propagate
// upward), so the hash at byte n is a pure function of bytes [n-7..n] and [n-W-7..n-W]
// (SURVEY.md F2).  Candidate detection is therefore a byte-parallel map: every thread warms the
// recurrence up over the 8 positions before its span and then rolls forward.  Only the min/max
// selection is sequential, over a sparse bitmap.  No MFMA anywhere: this is byte/integer work.
#include <cstdlib>

#include "common.h"
#include "ingest_launch.h"

namespace yams_accel {

// 16 bytes with dword alignment only: lets the compiler emit global_load_dwordx4 on addresses that
// are merely 4-byte aligned.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// ------------------------------------------------------------------------------------------------
// K6a: candidate bitmap.  One workgroup = one piece (kCdcPiece bytes) of one blob.
// ------------------------------------------------------------------------------------------------
constexpr int CDC_THREADS = 256;
constexpr int CDC_SPAN = kCdcPiece / CDC_THREADS; // bytes per thread (128)
constexpr int CDC_LOOKBACK = 64;                  // >= window(48) + 8 warm-up steps
static_assert(CDC_SPAN == 128, "bitmap packing below assumes 128-byte spans");

// skewed LDS byte address: 4 bytes of padding per 128 bytes so that lane-strided spans hit
// distinct banks.
__device__ __forceinline__ uint32_t skew(uint32_t b) { return b + ((b >> 7) << 2); }

__global__ __launch_bounds__(CDC_THREADS) void cdc_candidates_kernel(
    const uint8_t* data, const uint64_t* blob_off, const uint64_t* blob_len,
    const uint64_t* piece_prefix, uint32_t n_blobs, CdcParams cp, uint32_t* bitmap) {
    __shared__ uint64_t s_table[256];
    // lookback + piece + 16 bytes of alignment slack, skewed
    __shared__ __attribute__((aligned(16))) uint8_t s_data[(CDC_LOOKBACK + kCdcPiece + 32) / 128 * 132 + 264];
    __shared__ uint32_t s_blob;

    const uint64_t piece = blockIdx.x;
    if (threadIdx.x == 0) { // binary search: last blob with piece_prefix[b] <= piece
        uint32_t lo = 0, hi = n_blobs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (piece_prefix[mid] <= piece) lo = mid; else hi = mid;
        }
        s_blob = lo;
    }
    // out-table (rabin_fingerprint_table.h:17-26): xor of (poly << bit) over the set bits
    {
        const int byte = threadIdx.x;
        uint64_t hsh = 0;
#pragma unroll
        for (int bit = 0; bit < 8; ++bit)
            if (byte & (1 << bit)) hsh ^= cp.polynomial << bit;
        s_table[byte] = hsh;
    }
    __syncthreads();
    const uint32_t b = s_blob;
    const uint64_t blen = blob_len[b];
    const uint64_t p0 = (piece - piece_prefix[b]) * kCdcPiece; // piece start within the blob
    const uint8_t* bptr = data + blob_off[b];

    // ---- stage [p0 - LOOKBACK, p0 + piece) into LDS with aligned 16-byte loads -----------------
    // LDS logical index i corresponds to blob byte (p0 - LOOKBACK - shift + i), where shift makes
    // the first global address 16-byte aligned.
    const int64_t first = static_cast<int64_t>(p0) - CDC_LOOKBACK;
    const uintptr_t gaddr = reinterpret_cast<uintptr_t>(bptr) + first;
    const uint32_t shift = static_cast<uint32_t>(gaddr & 15u);
    const uint8_t* gbase = reinterpret_cast<const uint8_t*>(gaddr - shift);
    const uint32_t total = CDC_LOOKBACK + kCdcPiece + shift; // logical bytes needed
    const int64_t valid_lo = -first + shift;                          // logical index of blob byte 0
    const int64_t valid_hi = static_cast<int64_t>(blen) - first + shift; // one past the last byte
    for (uint32_t i = threadIdx.x * 16; i < total; i += CDC_THREADS * 16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        // a 16-byte granule is loaded only if it overlaps the blob
        if (static_cast<int64_t>(i) + 16 > valid_lo && static_cast<int64_t>(i) < valid_hi)
            v = *reinterpret_cast<const uint4*>(gbase + i);
        // zero bytes that lie outside the blob (before its start or past its end)
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t bi = static_cast<int64_t>(i) + 4 * k;
            if (bi < valid_lo || bi + 4 > valid_hi) {
                uint32_t m = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (bi + e >= valid_lo && bi + e < valid_hi) m |= 0xffu << (8 * e);
                w[k] &= m;
            }
            *reinterpret_cast<uint32_t*>(&s_data[skew(i + 4 * k)]) = w[k];
        }
    }
    __syncthreads();

    // ---- each thread rolls over its 128-byte span ----------------------------------------------
    const uint32_t W = cp.window;
    const uint64_t mask = cp.mask;
    const uint32_t span0 = CDC_LOOKBACK + shift + threadIdx.x * CDC_SPAN; // logical index
    uint64_t h = 0;
#pragma unroll
    for (int wstep = 8; wstep >= 1; --wstep) {
        const uint32_t i = span0 - wstep;
        const uint8_t nb = s_data[skew(i)];
        const uint8_t ob = s_data[skew(i - W)];
        h = ((h - s_table[ob]) << 8) ^ s_table[nb];
    }
    uint32_t bits[4] = {0, 0, 0, 0};
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        uint32_t acc = 0;
        for (int j = 0; j < 32; ++j) {
            const uint32_t i = span0 + wd * 32 + j;
            const uint8_t nb = s_data[skew(i)];
            const uint8_t ob = s_data[skew(i - W)];
            h = ((h - s_table[ob]) << 8) ^ s_table[nb];
            acc |= static_cast<uint32_t>((h & mask) == mask) << j;
        }
        bits[wd] = acc;
    }
    // positions past the end of the blob carry no candidates
    const uint64_t pos0 = p0 + static_cast<uint64_t>(threadIdx.x) * CDC_SPAN;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        const uint64_t wp = pos0 + wd * 32;
        if (wp >= blen) bits[wd] = 0;
        else if (wp + 32 > blen) bits[wd] &= (1u << (blen - wp)) - 1u;
    }
    uint4* dst = reinterpret_cast<uint4*>(bitmap + piece * (kCdcPiece / 32) + threadIdx.x * 4);
    *dst = make_uint4(bits[0], bits[1], bits[2], bits[3]);
}

// ------------------------------------------------------------------------------------------------
// K6b: candidate bitmap, narrow form (window == 48 and mask < 2^31 — the product defaults).
// The match test only looks at the low bits of h, and subtraction, left shift and xor only carry
// upward, so the low 32 bits of h follow the same recurrence on uint32_t and forget their history
// after 4 steps.  One lane = one 32-byte unit = one bitmap word: it loads the unit and the 32 bytes
// 48 behind it straight from global memory (consecutive lanes -> consecutive units, fully
// coalesced; no data staging in LDS), realigns with v_alignbyte, warms up over 4 bytes and rolls.
// Per byte: 2 LDS table reads + ~9 VALU, about half of the generic kernel above.
// ------------------------------------------------------------------------------------------------
// (five waves per SIMD: 96 registers instead of 110 at no spill — a wave issues at most one instruction per ~4.8 clocks, so
//  issue slots are filled by WAVES, and this kernel shares its SIMDs with the long-chain kernel's waves for most of a call)
__global__ __launch_bounds__(CDC_THREADS) __attribute__((amdgpu_waves_per_eu(5))) void cdc_candidates_w48_kernel(
    const uint8_t* data, const uint64_t* blob_off, const uint64_t* blob_len,
    const uint64_t* piece_prefix, uint32_t n_blobs, CdcParams cp, uint32_t* bitmap) {
    __shared__ uint32_t s_t32[256];
    __shared__ uint32_t s_blob;
    const uint64_t piece = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = n_blobs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (piece_prefix[mid] <= piece) lo = mid; else hi = mid;
        }
        s_blob = lo;
    }
    {
        const int byte = threadIdx.x;
        const uint32_t poly = static_cast<uint32_t>(cp.polynomial);
        uint32_t hsh = 0;
#pragma unroll
        for (int bit = 0; bit < 8; ++bit)
            if (byte & (1 << bit)) hsh ^= poly << bit;
        s_t32[byte] = hsh;
    }
    __syncthreads();
    const uint32_t b = s_blob;
    const uint64_t blen = blob_len[b];
    const uint64_t p0 = (piece - piece_prefix[b]) * kCdcPiece;
    const uint8_t* bptr = data + blob_off[b];
    const uint32_t mask = static_cast<uint32_t>(cp.mask);
    uint32_t* out = bitmap + piece * (kCdcPiece / 32);

#pragma unroll 1
    for (int it = 0; it < kCdcPiece / 32 / CDC_THREADS; ++it) {
        const uint32_t unit = it * CDC_THREADS + threadIdx.x;
        const uint64_t pos = p0 + 32ull * unit;
        uint32_t bits = 0;
        if (pos >= 56 && pos + 36 <= blen) {
            const uintptr_t addr = reinterpret_cast<uintptr_t>(bptr + pos);
            const uint32_t sh = static_cast<uint32_t>(addr & 3u);
            const uint32_t* a = reinterpret_cast<const uint32_t*>(addr - sh);
            uint32_t N[10], O[10];
            N[0] = a[-1];
            O[0] = a[-13];
            {
                const u32x4_a4 n0 = *reinterpret_cast<const u32x4_a4*>(a);
                const u32x4_a4 n1 = *reinterpret_cast<const u32x4_a4*>(a + 4);
                const u32x4_a4 o0 = *reinterpret_cast<const u32x4_a4*>(a - 12);
                const u32x4_a4 o1 = *reinterpret_cast<const u32x4_a4*>(a - 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { N[1 + j] = n0[j]; N[5 + j] = n1[j]; O[1 + j] = o0[j]; O[5 + j] = o1[j]; }
            }
            N[9] = a[8];
            O[9] = a[-4];
            uint32_t n[9], o[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                n[j] = __builtin_amdgcn_alignbyte(N[j + 1], N[j], sh);
                o[j] = __builtin_amdgcn_alignbyte(O[j + 1], O[j], sh);
            }
            uint32_t h = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                h = ((h - s_t32[(o[0] >> (8 * e)) & 0xffu]) << 8) ^ s_t32[(n[0] >> (8 * e)) & 0xffu];
            uint32_t acc = 0; // first position ends up in bit 31
#pragma unroll
            for (int j = 1; j < 9; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h = ((h - s_t32[(o[j] >> (8 * e)) & 0xffu]) << 8) ^ s_t32[(n[j] >> (8 * e)) & 0xffu];
                    const uint32_t x = ~h & mask;               // 0 iff every mask bit is set
                    acc = __builtin_amdgcn_alignbit(acc, x - 1u, 31); // (acc << 1) | (x == 0)
                }
            }
            bits = __builtin_bitreverse32(acc);
        } else if (pos < blen) {
            // units touching either end of the blob: bytes outside it read as zero
            uint32_t h = 0;
            for (int i = -4; i < 32; ++i) {
                const int64_t q = static_cast<int64_t>(pos) + i;
                const int64_t qo = q - 48;
                const uint32_t nb = (q >= 0 && q < static_cast<int64_t>(blen)) ? bptr[q] : 0u;
                const uint32_t ob = (qo >= 0 && qo < static_cast<int64_t>(blen)) ? bptr[qo] : 0u;
                h = ((h - s_t32[ob]) << 8) ^ s_t32[nb];
                if (i >= 0 && q < static_cast<int64_t>(blen) && (h & mask) == mask) bits |= 1u << i;
            }
        }
        out[unit] = bits;
    }
}

// ------------------------------------------------------------------------------------------------
// K6c: boundary walk.  One wave per blob walks the sparse bitmap:
//   s = 0; while s < N: lo = s + min - delta (delta = 1 for Streaming, 0 for Rabin);
//   p = first candidate >= lo with p < min(s + max, N); end = found ? p + 1 : min(s + max, N).
// (rabin_chunker.cpp:63-110; streaming_chunker.h:146-181.)  Chunks go to per-blob slots.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cdc_walk_kernel(const uint32_t* bitmap,
                                                      const uint64_t* blob_len,
                                                      const uint64_t* piece_prefix,
                                                      const uint64_t* slot_prefix,
                                                      uint32_t n_blobs, CdcParams cp,
                                                      uint64_t* slot_off, uint64_t* slot_size,
                                                      uint64_t* blob_count) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blobs) return;
    const int lane = threadIdx.x;
    const uint64_t N = blob_len[b];
    const uint32_t* bm = bitmap + piece_prefix[b] * (kCdcPiece / 32);
    const uint64_t nwords = (N + 31) / 32;
    const uint64_t slot0 = slot_prefix[b];
    const uint64_t minsz = cp.min_size;
    const uint64_t maxe = cp.max_size > cp.min_size ? cp.max_size : cp.min_size;
    const uint64_t delta = cp.streaming ? 1 : 0;
    uint64_t s = cp.context < N ? cp.context : N, count = 0; // (a windowed stream: the bytes in front of `context` are history)
    while (s < N) {
        uint64_t lo = s + minsz;
        lo = lo >= delta ? lo - delta : 0;
        if (lo < s) lo = s;
        uint64_t hi = s + maxe; // exclusive bound on tested positions
        if (hi > N) hi = N;
        uint64_t end = hi;
        if (lo < hi) {
            // scan words [lo/32, (hi-1)/32], 64 words per step
            const uint64_t w_first = lo >> 5, w_last = (hi - 1) >> 5;
            for (uint64_t wb = w_first; wb <= w_last; wb += 64) {
                const uint64_t wi = wb + lane;
                uint32_t word = 0;
                if (wi <= w_last && wi < nwords) word = bm[wi];
                if (wi == w_first) word &= 0xffffffffu << (lo & 31);
                if (wi == w_last) {
                    const uint32_t keep = static_cast<uint32_t>(((hi - 1) & 31) + 1);
                    if (keep < 32) word &= (1u << keep) - 1u;
                }
                const unsigned long long ball = __ballot(word != 0);
                if (ball) {
                    const int src = __ffsll(static_cast<long long>(ball)) - 1;
                    const uint32_t wsel = __shfl(word, src);
                    const uint64_t p = ((wb + src) << 5) + (__ffs(static_cast<int>(wsel)) - 1);
                    end = p + 1;
                    break;
                }
            }
        }
        if (lane == 0) { slot_off[slot0 + count] = s; slot_size[slot0 + count] = end - s; }
        ++count;
        s = end;
    }
    if (lane == 0) blob_count[b] = count;
}

// Exclusive prefix over per-blob chunk counts (single workgroup; n_blobs is small metadata).
__global__ __launch_bounds__(1024) void chunk_prefix_kernel(const uint64_t* blob_count,
                                                            uint32_t n_blobs, uint64_t* blob_first) {
    __shared__ uint64_t s_part[1024];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blobs; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n_blobs ? blob_count[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
            uint64_t add = 0;
            if (static_cast<int>(threadIdx.x) >= d) add = s_part[threadIdx.x - d];
            __syncthreads();
            s_part[threadIdx.x] += add;
            __syncthreads();
        }
        const uint64_t incl = s_part[threadIdx.x];
        if (i < n_blobs) blob_first[i] = s_carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) blob_first[n_blobs] = s_carry;
}

// Scatter per-blob slots into the dense chunk arrays + build the SHA message list
// (absolute byte offset of every chunk inside `data`).
__global__ __launch_bounds__(256) void chunk_compact_kernel(
    const uint64_t* slot_prefix, const uint64_t* slot_off, const uint64_t* slot_size,
    const uint64_t* blob_count, const uint64_t* blob_first, const uint64_t* blob_off,
    uint32_t n_blobs, uint64_t* chunk_offset, uint64_t* chunk_size, uint32_t* chunk_blob,
    uint64_t* msg_off, uint64_t* msg_len) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blobs) return;
    const uint64_t n = blob_count[b], src = slot_prefix[b], dst = blob_first[b];
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t o = slot_off[src + i], z = slot_size[src + i];
        chunk_offset[dst + i] = o;
        chunk_size[dst + i] = z;
        chunk_blob[dst + i] = b;
        if (msg_off) { msg_off[dst + i] = blob_off[b] + o; msg_len[dst + i] = z; }
    }
}

// ------------------------------------------------------------------------------------------------
// K5: SHA-256, one message per lane, lanes pull messages from a shared queue so that ragged
// message lengths do not idle the wave.  32-bit integer VALU only.
// ------------------------------------------------------------------------------------------------
__constant__ uint32_t kSha256K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u,
    0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu,
    0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu,
    0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u,
    0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu,
    0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu,
    0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u,
    0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,
    0xc67178f2u};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) {
    return __builtin_amdgcn_alignbit(x, x, n); // v_alignbit_b32
}

__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = h + S1 + ch + kSha256K[i] + w[i & 15];
        const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__device__ __forceinline__ void sha256_init(uint32_t st[8]) {
    st[0] = 0x6a09e667u; st[1] = 0xbb67ae85u; st[2] = 0x3c6ef372u; st[3] = 0xa54ff53au;
    st[4] = 0x510e527fu; st[5] = 0x9b05688cu; st[6] = 0x1f83d9abu; st[7] = 0x5be0cd19u;
}

// Two-message round function: the two states are independent dependency chains, so the scheduler
// interleaves them and a wave that holds long messages (whole-blob digests: 65 536 sequential
// blocks for 4 MiB) is no longer bound by the VALU result latency of a single chain.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); // one v_bitop3_b32 instead of two v_xor_b32
}
template <int NS>
__device__ __forceinline__ void sha256_compress_n(uint32_t st[NS][8], uint32_t w[NS][16]) {
    uint32_t a[NS], b[NS], c[NS], d[NS], e[NS], f[NS], g[NS], h[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        a[k] = st[k][0]; b[k] = st[k][1]; c[k] = st[k][2]; d[k] = st[k][3];
        e[k] = st[k][4]; f[k] = st[k][5]; g[k] = st[k][6]; h[k] = st[k][7];
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (i >= 16) {
                const uint32_t w15 = w[k][(i - 15) & 15], w2 = w[k][(i - 2) & 15];
                const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
                const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
                w[k][i & 15] = w[k][i & 15] + s0 + w[k][(i - 7) & 15] + s1;
            }
            const uint32_t S1 = xor3(rotr(e[k], 6), rotr(e[k], 11), rotr(e[k], 25));
            const uint32_t ch = __builtin_amdgcn_bitop3_b32(e[k], f[k], g[k], 0xCA);  // e ? f : g
            const uint32_t t1 = h[k] + S1 + ch + kSha256K[i] + w[k][i & 15];
            const uint32_t S0 = xor3(rotr(a[k], 2), rotr(a[k], 13), rotr(a[k], 22));
            const uint32_t mj = __builtin_amdgcn_bitop3_b32(a[k], b[k], c[k], 0xE8);  // majority
            const uint32_t t2 = S0 + mj;
            h[k] = g[k]; g[k] = f[k]; f[k] = e[k]; e[k] = d[k] + t1;
            d[k] = c[k]; c[k] = b[k]; b[k] = a[k]; a[k] = t1 + t2;
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        st[k][0] += a[k]; st[k][1] += b[k]; st[k][2] += c[k]; st[k][3] += d[k];
        st[k][4] += e[k]; st[k][5] += f[k]; st[k][6] += g[k]; st[k][7] += h[k];
    }
}

// Fetch of the 17 dwords that cover the 64 bytes at p (any alignment) as four 16-byte loads + one
// dword from the 4-byte-aligned address below p (global_load_dwordx4 only needs dword alignment):
// 5 requests per block instead of 17 (the lanes of a wave walk different messages, so every
// request is its own cache-line access).  Nothing beyond the dword that holds the last message
// byte is ever touched: a 16-byte piece that would cross it falls back to guarded dword loads
// (only in the last block or two of a message).
struct ShaWindow { u32x4_a4 v[4]; uint32_t tail; };
__device__ __forceinline__ void fetch_window(const uint8_t* p, const uint8_t* limit, ShaWindow& win) {
    const uint8_t* q = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~static_cast<uintptr_t>(3));
    const uint8_t* limit4 = reinterpret_cast<const uint8_t*>((reinterpret_cast<uintptr_t>(limit) + 3) & ~static_cast<uintptr_t>(3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint8_t* a = q + 16 * i;
        if (a + 16 <= limit4) {
            win.v[i] = *reinterpret_cast<const u32x4_a4*>(a);
        } else {
            u32x4_a4 t = {0, 0, 0, 0};
            if (a < limit) t[0] = *reinterpret_cast<const uint32_t*>(a);
            if (a + 4 < limit) t[1] = *reinterpret_cast<const uint32_t*>(a + 4);
            if (a + 8 < limit) t[2] = *reinterpret_cast<const uint32_t*>(a + 8);
            if (a + 12 < limit) t[3] = *reinterpret_cast<const uint32_t*>(a + 12);
            win.v[i] = t;
        }
    }
    win.tail = (q + 64 < limit) ? *reinterpret_cast<const uint32_t*>(q + 64) : 0u;
}
// Realign + byte-swap with one v_perm_b32 per word: big-endian word i = bytes [sh+4i, sh+4i+4).
__device__ __forceinline__ void window_words_be(const ShaWindow& win, const uint8_t* p, uint32_t w[16]) {
    const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p) & 3u);
    const uint32_t sel = 0x00010203u + sh * 0x01010101u;
    const uint32_t d[17] = {win.v[0][0], win.v[0][1], win.v[0][2], win.v[0][3], win.v[1][0], win.v[1][1],
                            win.v[1][2], win.v[1][3], win.v[2][0], win.v[2][1], win.v[2][2], win.v[2][3],
                            win.v[3][0], win.v[3][1], win.v[3][2], win.v[3][3], win.tail};
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = __builtin_amdgcn_perm(d[i + 1], d[i], sel);
}

// Lane state machine, NS message slots per lane.  phase: 0 = data blocks, 1 = length-only block
// pending, 2 = needs a new message, 3 = queue drained.  `win` always holds the window of the block
// at `p` (fetched one iteration ahead so the memory latency hides under the 64 rounds).
// Messages [0, n_long) are "long" (whole blobs), the rest "short" (chunks): slot 0 of a lane
// prefers long messages, the other slots short ones, so long chains spread over as many lanes as
// possible and every lane that holds one also has independent work to overlap with it.
template <int NS>
__global__ __launch_bounds__(256) void sha256_batch_kernel(const uint8_t* data,
                                                           const uint64_t* offs,
                                                           const uint64_t* lens, uint64_t n_long,
                                                           uint64_t n_msgs, uint8_t* digests,
                                                           unsigned long long* heads /*[2]*/,
                                                           const uint32_t* init_state /*nullable*/,
                                                           uint32_t* out_state /*nullable*/,
                                                           int raw_blocks_only) {
    const uint8_t* p[NS];
    const uint8_t* end[NS];
    uint64_t total[NS], index[NS];
    uint32_t st[NS][8];
    ShaWindow win[NS];
    int phase[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        p[k] = nullptr; end[k] = nullptr; total[k] = 0; index[k] = 0; phase[k] = 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) st[k][i] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) win[k].v[i] = u32x4_a4{0, 0, 0, 0};
        win[k].tail = 0;
    }
    const uint64_t n_short = n_msgs - n_long;
    for (;;) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (phase[k] == 2) {
                // two queues: [0, n_long) and [n_long, n_msgs); preferred one first
                unsigned long long idx = ~0ull;
                const bool prefer_long = (k == 0);
                if (prefer_long && n_long) {
                    const unsigned long long i = atomicAdd(&heads[0], 1ull);
                    if (i < n_long) idx = i;
                }
                if (idx == ~0ull && n_short) {
                    const unsigned long long i = atomicAdd(&heads[1], 1ull);
                    if (i < n_short) idx = n_long + i;
                }
                if (idx == ~0ull && !prefer_long && n_long) {
                    const unsigned long long i = atomicAdd(&heads[0], 1ull);
                    if (i < n_long) idx = i;
                }
                if (idx != ~0ull) {
                    // a whole-blob digest is one long dependent chain: let the waves that carry
                    // one win issue arbitration against the short-message waves on their SIMD
                    if (__any(idx < n_long)) __builtin_amdgcn_s_setprio(3);
                    index[k] = idx;
                    p[k] = data + offs[idx];
                    total[k] = lens[idx];
                    end[k] = p[k] + total[k];
                    if (init_state) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) st[k][i] = init_state[idx * 8 + i];
                    } else sha256_init(st[k]);
                    fetch_window(p[k], end[k], win[k]); // the only un-hidden fetch of this message
                    phase[k] = 0;
                } else {
                    phase[k] = 3;
                }
            }
        }
        bool drained = true;
#pragma unroll
        for (int k = 0; k < NS; ++k) drained &= (phase[k] == 3);
        if (__all(drained)) break;

        uint32_t w[NS][16];
        bool final_block[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            final_block[k] = false;
            if (phase[k] == 0) {
                const uint64_t rem = static_cast<uint64_t>(end[k] - p[k]);
                if (raw_blocks_only && rem < 64) {
                    final_block[k] = true; // no data at all (total == 0): state passes through
#pragma unroll
                    for (int i = 0; i < 16; ++i) w[k][i] = 0;
                    phase[k] = 4;          // skip the compression below for this slot
                } else {
                    window_words_be(win[k], p[k], w[k]);
                    if (rem >= 64) {
                        p[k] += 64;
                        if (raw_blocks_only && p[k] == end[k]) final_block[k] = true;
                        if (p[k] < end[k]) fetch_window(p[k], end[k], win[k]); // prefetch for the next iteration
                    } else {
                        const uint32_t r = static_cast<uint32_t>(rem);
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const uint32_t lo = 4u * j;
                            if (r <= lo) w[k][j] = 0;
                            else if (r < lo + 4) w[k][j] &= 0xffffffffu << (8u * (lo + 4 - r));
                            if ((r >> 2) == static_cast<uint32_t>(j)) w[k][j] |= 0x80u << (24 - 8 * (r & 3));
                        }
                        p[k] = end[k];
                        if (r < 56) {
                            const uint64_t bitlen = total[k] * 8ull;
                            w[k][14] = static_cast<uint32_t>(bitlen >> 32);
                            w[k][15] = static_cast<uint32_t>(bitlen);
                            final_block[k] = true;
                        } else {
                            phase[k] = 1;
                            // (the length-only block is built on the next iteration)
                            final_block[k] = false;
                            // mark: this iteration still compresses the padded data block
                        }
                    }
                }
            } else if (phase[k] == 1) {
#pragma unroll
                for (int i = 0; i < 14; ++i) w[k][i] = 0;
                const uint64_t bitlen = total[k] * 8ull;
                w[k][14] = static_cast<uint32_t>(bitlen >> 32);
                w[k][15] = static_cast<uint32_t>(bitlen);
                final_block[k] = true;
                phase[k] = 5; // length block in flight
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) w[k][i] = 0;
            }
        }
        // Slots that are idle (phase 3) or pass-through (phase 4) compress zeros into a state that
        // is re-initialised before its next use; saving the branch keeps the two chains interleaved.
        uint32_t keep[NS][8];
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) keep[k][i] = st[k][i];
        sha256_compress_n<NS>(st, w);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (phase[k] == 4) { // raw pass-through: restore the untouched state
#pragma unroll
                for (int i = 0; i < 8; ++i) st[k][i] = keep[k][i];
            }
            if (final_block[k]) {
                if (out_state) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) out_state[index[k] * 8 + i] = st[k][i];
                }
                if (digests) {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(digests + index[k] * 32);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        dst[i] = __builtin_amdgcn_perm(0u, st[k][i], 0x00010203u); // big-endian bytes
                }
                phase[k] = 2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Long messages (whole-blob digests).  A 4 MiB blob is ONE chain of 65 536 dependent compressions
// on one lane, and a wave issues at most one instruction per ~4.8 cycles however much ILP it has,
// so the per-lane kernel above needs ~197 ms for it regardless of how many blobs there are.  Here
// a workgroup of two waves shares 64 messages (lane j of both waves <-> message j): the PRODUCER
// wave fetches block b, builds padding, runs the 48-step message schedule and pre-adds the round
// constants; it hands the 64 words (w[i] + K[i]) to the CONSUMER wave through LDS, double-buffered
// with one s_barrier per block; the consumer executes only the 64 rounds (~14 instructions each)
// and carries the chaining state.  Critical path per block: ~920 instead of ~1400 instructions.
// ------------------------------------------------------------------------------------------------
constexpr int LONG_LANE_STRIDE = 272;                  // bytes per lane per buffer: 64 words + pad,
                                                       // 68-word stride keeps ds_*_b128 conflict-free
constexpr int LONG_BUF_BYTES = 64 * LONG_LANE_STRIDE;  // 17 408
constexpr int LONG_PAIRS = 2;                          // producer/consumer pairs per workgroup
template <int CONSUMER_PRIO = 3, int PRODUCER_PRIO = 2> // (wave priorities; other values: measurement build)
__global__ __launch_bounds__(128 * LONG_PAIRS) void sha256_long_kernel(const uint8_t* data, const uint64_t* offs,
                                                          const uint64_t* lens,
                                                          const uint32_t* out_slot /*nullable*/,
                                                          uint64_t n_msgs, uint8_t* digests) {
    __shared__ __attribute__((aligned(16))) unsigned char kw_lds_all[LONG_PAIRS * 2 * LONG_BUF_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave & 1; // 0 = consumer, 1 = producer; a pair sits on neighbouring SIMDs
    const int pair = wave >> 1;
    unsigned char* kw_lds = kw_lds_all + pair * (2 * LONG_BUF_BYTES);
    const uint64_t m = (static_cast<uint64_t>(blockIdx.x) * LONG_PAIRS + pair) * 64 + lane;
    const bool have = m < n_msgs;
    const uint64_t total = have ? lens[m] : 0;
    // blocks this lane's message needs, padding included: data + 0x80 + 8-byte length
    const uint64_t nblk = have ? (total + 8) / 64 + 1 : 0;
    uint64_t maxblk = nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t o = __shfl_xor(maxblk, d);
        maxblk = o > maxblk ? o : maxblk;
    }
    {   // the barrier count must agree across the whole workgroup: take the max over its pairs
        __shared__ unsigned long long s_maxblk[2 * LONG_PAIRS];
        if (lane == 0) s_maxblk[wave] = maxblk;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * LONG_PAIRS; ++i) maxblk = s_maxblk[i] > maxblk ? s_maxblk[i] : maxblk;
    }
    unsigned char* mybuf0 = kw_lds + lane * LONG_LANE_STRIDE;

    if (role == 1) {
        // ---------------- producer ----------------
        __builtin_amdgcn_s_setprio(PRODUCER_PRIO);
        const uint8_t* p = have ? data + offs[m] : nullptr;
        const uint8_t* end = p + total;
        ShaWindow win;
#pragma unroll
        for (int i = 0; i < 4; ++i) win.v[i] = u32x4_a4{0, 0, 0, 0};
        win.tail = 0;
        if (have) fetch_window(p, end, win);
        int phase = have ? 0 : 3; // 0 data, 1 length-only block pending, 3 done
        for (uint64_t b = 0; b < maxblk; ++b) {
            uint32_t w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = 0;
            if (phase == 0) {
                const uint64_t rem = static_cast<uint64_t>(end - p);
                window_words_be(win, p, w);
                if (rem >= 64) {
                    p += 64;
                    if (p < end) fetch_window(p, end, win);
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) win.v[i] = u32x4_a4{0, 0, 0, 0};
                        win.tail = 0;
                    }
                } else {
                    const uint32_t r = static_cast<uint32_t>(rem);
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t lo = 4u * j;
                        if (r <= lo) w[j] = 0;
                        else if (r < lo + 4) w[j] &= 0xffffffffu << (8u * (lo + 4 - r));
                        if ((r >> 2) == static_cast<uint32_t>(j)) w[j] |= 0x80u << (24 - 8 * (r & 3));
                    }
                    p = end;
                    if (r < 56) {
                        const uint64_t bitlen = total * 8ull;
                        w[14] = static_cast<uint32_t>(bitlen >> 32);
                        w[15] = static_cast<uint32_t>(bitlen);
                        phase = 3;
                    } else {
                        phase = 1;
                    }
                }
            } else if (phase == 1) {
                const uint64_t bitlen = total * 8ull;
                w[14] = static_cast<uint32_t>(bitlen >> 32);
                w[15] = static_cast<uint32_t>(bitlen);
                phase = 3;
            }
            unsigned char* dst = mybuf0 + (b & 1) * LONG_BUF_BYTES;
            // rounds 0..15 use the block words, 16..63 the schedule; K is folded in here
#pragma unroll
            for (int i = 0; i < 64; i += 4) {
                uint32_t o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = i + e;
                    if (r >= 16) {
                        const uint32_t w15 = w[(r - 15) & 15], w2 = w[(r - 2) & 15];
                        const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
                        const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
                        w[r & 15] = w[r & 15] + s0 + w[(r - 7) & 15] + s1;
                    }
                    o4[e] = w[r & 15] + kSha256K[r];
                }
                *reinterpret_cast<uint4*>(dst + i * 4) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the LDS writes have landed
            __builtin_amdgcn_s_barrier();       // block b is published; buffer (b+1)&1 is free again
        }
        __builtin_amdgcn_s_barrier(); // pairs with the consumer's final barrier
    } else {
        // ---------------- consumer ----------------
        __builtin_amdgcn_s_setprio(CONSUMER_PRIO);
        uint32_t st[8];
        sha256_init(st);
        __builtin_amdgcn_s_barrier(); // block 0 is published
        for (uint64_t b = 0; b < maxblk; ++b) {
            const unsigned char* src = mybuf0 + (b & 1) * LONG_BUF_BYTES;
            if (b < nblk) {
                uint32_t a = st[0], bb = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
                for (int i = 0; i < 64; i += 4) {
                    const uint4 k4 = *reinterpret_cast<const uint4*>(src + i * 4);
                    const uint32_t kw[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
                        const uint32_t ch = __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA);
                        const uint32_t t1 = h + S1 + ch + kw[q];
                        const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
                        const uint32_t mj = __builtin_amdgcn_bitop3_b32(a, bb, c, 0xE8);
                        h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + S0 + mj;
                    }
                }
                st[0] += a; st[1] += bb; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // my LDS reads of this buffer are complete
            __builtin_amdgcn_s_barrier();       // producer may overwrite it; block b+1 is published
        }
        if (have) {
            const uint64_t slot = out_slot ? out_slot[m] : m;
            uint32_t* dstd = reinterpret_cast<uint32_t*>(digests + slot * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) dstd[i] = __builtin_amdgcn_perm(0u, st[i], 0x00010203u);
        }
    }
}

// Batched integrity check: valid[i] = (digest_i == expected_i); counts the mismatches.
// (ChunkValidator::validateChunkInternal compares the hex strings, src/integrity/chunk_validator.cpp:
// 230-262; the 32 raw bytes decide the same thing.)
__global__ __launch_bounds__(256) void digest_compare_kernel(const uint8_t* actual, const uint8_t* expected,
                                                             uint64_t n, uint8_t* valid,
                                                             unsigned long long* n_invalid) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(actual + i * 32);
        const uint8_t* e = expected + i * 32; // any alignment
        uint32_t diff = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const uint32_t ew = static_cast<uint32_t>(e[4 * w]) | (static_cast<uint32_t>(e[4 * w + 1]) << 8) |
                                (static_cast<uint32_t>(e[4 * w + 2]) << 16) | (static_cast<uint32_t>(e[4 * w + 3]) << 24);
            diff |= a[w] ^ ew;
        }
        bad = diff != 0;
        valid[i] = bad ? 0 : 1;
    }
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_invalid, static_cast<unsigned long long>(__popcll(m)));
}

// =================================================================================================
// Launchers
// =================================================================================================
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_cdc_candidates(hipStream_t st, const uint8_t* data, const uint64_t* blob_off,
                                 const uint64_t* blob_len, const uint64_t* piece_prefix,
                                 uint32_t n_blobs, uint64_t n_pieces, const CdcParams& cp,
                                 uint32_t* bitmap) {
    if (n_pieces == 0) return hipSuccess;
    if (cp.window == 48 && cp.mask < (1ull << 31) && !cp.generic)
        hipLaunchKernelGGL(cdc_candidates_w48_kernel, dim3(static_cast<uint32_t>(n_pieces)),
                           dim3(CDC_THREADS), 0, st, data, blob_off, blob_len, piece_prefix, n_blobs,
                           cp, bitmap);
    else
        hipLaunchKernelGGL(cdc_candidates_kernel, dim3(static_cast<uint32_t>(n_pieces)),
                           dim3(CDC_THREADS), 0, st, data, blob_off, blob_len, piece_prefix, n_blobs,
                           cp, bitmap);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_cdc_walk(hipStream_t st, const uint32_t* bitmap, const uint64_t* blob_len,
                           const uint64_t* piece_prefix, const uint64_t* slot_prefix,
                           uint32_t n_blobs, const CdcParams& cp, uint64_t* slot_off,
                           uint64_t* slot_size, uint64_t* blob_count) {
    if (n_blobs == 0) return hipSuccess;
    hipLaunchKernelGGL(cdc_walk_kernel, dim3(n_blobs), dim3(64), 0, st, bitmap, blob_len,
                       piece_prefix, slot_prefix, n_blobs, cp, slot_off, slot_size, blob_count);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_chunk_compact(hipStream_t st, const uint64_t* slot_prefix,
                                const uint64_t* slot_off, const uint64_t* slot_size,
                                const uint64_t* blob_count, uint64_t* blob_first,
                                const uint64_t* blob_off, uint32_t n_blobs, uint64_t* chunk_offset,
                                uint64_t* chunk_size, uint32_t* chunk_blob, uint64_t* msg_off,
                                uint64_t* msg_len) {
    hipLaunchKernelGGL(chunk_prefix_kernel, dim3(1), dim3(1024), 0, st, blob_count, n_blobs,
                       blob_first);
    LAUNCH_CHECK();
    if (n_blobs == 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_compact_kernel, dim3(n_blobs), dim3(256), 0, st, slot_prefix, slot_off,
                       slot_size, blob_count, blob_first, blob_off, n_blobs, chunk_offset,
                       chunk_size, chunk_blob, msg_off, msg_len);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sha256(hipStream_t st, const uint8_t* data, const uint64_t* offs,
                         const uint64_t* lens, uint64_t n_long, uint64_t n_msgs, uint8_t* digests,
                         unsigned long long* queue_heads, const uint32_t* init_state,
                         uint32_t* out_state, int raw_blocks_only, uint32_t max_blocks, int slots) {
    if (n_msgs == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(queue_heads, 0, 2 * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const uint64_t per_block = 256ull * static_cast<uint64_t>(slots);
    uint64_t want = (n_msgs + per_block - 1) / per_block;
    if (want > max_blocks) want = max_blocks;
    if (want == 0) want = 1;
    if (slots == 2)
        hipLaunchKernelGGL(sha256_batch_kernel<2>, dim3(static_cast<uint32_t>(want)), dim3(256), 0, st,
                           data, offs, lens, n_long, n_msgs, digests, queue_heads, init_state, out_state,
                           raw_blocks_only);
    else
        hipLaunchKernelGGL(sha256_batch_kernel<1>, dim3(static_cast<uint32_t>(want)), dim3(256), 0, st,
                           data, offs, lens, n_long, n_msgs, digests, queue_heads, init_state, out_state,
                           raw_blocks_only);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sha256_long(hipStream_t st, const uint8_t* data, const uint64_t* offs,
                              const uint64_t* lens, const uint32_t* out_slot, uint64_t n_msgs,
                              uint8_t* digests) {
    if (n_msgs == 0) return hipSuccess;
    const dim3 grid(static_cast<uint32_t>((n_msgs + 64 * LONG_PAIRS - 1) / (64 * LONG_PAIRS))), block(128 * LONG_PAIRS);
#ifdef YAMS_ACCEL_MEASURE
    if (const char* e = std::getenv("YAMS_ACCEL_LONG_PRIO")) { // consumer / producer wave priorities, e.g. "10"
        const int v = std::atoi(e);
        if (v == 0) hipLaunchKernelGGL((sha256_long_kernel<0, 0>), grid, block, 0, st, data, offs, lens, out_slot, n_msgs, digests);
        else if (v == 10) hipLaunchKernelGGL((sha256_long_kernel<1, 0>), grid, block, 0, st, data, offs, lens, out_slot, n_msgs, digests);
        else if (v == 11) hipLaunchKernelGGL((sha256_long_kernel<1, 1>), grid, block, 0, st, data, offs, lens, out_slot, n_msgs, digests);
        else if (v == 21) hipLaunchKernelGGL((sha256_long_kernel<2, 1>), grid, block, 0, st, data, offs, lens, out_slot, n_msgs, digests);
        else hipLaunchKernelGGL((sha256_long_kernel<3, 2>), grid, block, 0, st, data, offs, lens, out_slot, n_msgs, digests);
        LAUNCH_CHECK();
        return hipSuccess;
    }
#endif
    hipLaunchKernelGGL((sha256_long_kernel<3, 2>), grid, block, 0, st, data, offs, lens, out_slot, n_msgs, digests);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_digest_compare(hipStream_t st, const uint8_t* actual, const uint8_t* expected, uint64_t n,
                                 uint8_t* valid, unsigned long long* n_invalid) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(digest_compare_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, st,
                       actual, expected, n, valid, n_invalid);
    LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace yams_accel
