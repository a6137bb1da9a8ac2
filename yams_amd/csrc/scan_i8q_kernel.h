// scan_i8q_kernel.h — MEASUREMENT BUILD ONLY (included by scan_i8_kernel.hip under YAMS_ACCEL_MEASURE, inside
// namespace yams_accel, after the resident-query kernel whose constants and helpers it shares).
// The 128 x 128 wave-tile form of the resident-query int8 filter: DESIGN.md 3.6.
// -------------------------------------------------------------------------------------------------
// The resident-query filter with 128 x 128 WAVE TILES (round 3; cosine only): ONE wave per SIMD, four waves per
// workgroup, 512 registers per wave.  Why: with 64 x 128 wave tiles (scan_tiles_i8r_kernel) every 32 MFMAs cost
// 12 KiB of LDS fragment reads + 4 KiB of DMA writes; at the matrix pipe's full rate that is exactly the 128 B/clk
// an LDS delivers, and the kernel sits at 70 % MFMA busy with the LDS 70 % busy beside it (profiles/r02_pmc.json;
// no fragment reads at all: the bare-MFMA 4.3 ms).  Here
//   * the ROW fragments do not pass through the LDS at all: a row piece is read by exactly one wave, the blocked
//     shadow already is the fragment image (a lane's 16 bytes of a 1 KiB piece), and a wave with 512 registers can
//     hold several slabs of its 128 rows in flight, loaded straight from global memory;
//   * the QUERY fragments (shared by the four waves and by every strip) stay in the LDS: 8 KiB of reads per
//     64 MFMAs and wave — 32 B/clk per CU instead of 128.
// The price: no second wave on the SIMD to hide a wave's epilogue — whatever the wave does between two strips is
// added to the launch, at one instruction per 4+ clocks.  So
//   * the accumulators are born in the first slab's MFMAs (C operand = the constant 0: no initialisation pass);
//     the thresholds enter in the epilogue (a survivor is I >= T: one maximum per 64-row block and query block
//     against T);
//   * the register files are used the other way round: the FRAGMENTS (MFMA A / B operands, the targets of
//     global_load / ds_read) live in the AGPRs, three quarters of the ACCUMULATORS (query blocks 0-5) in the
//     architectural VGPRs, where v_max3 reads them directly — a v_accvgpr_read per accumulator register made the
//     sign test 420 instructions per strip and 1.1 ms of the launch; it is 130 + 100 now (query blocks 6, 7 still
//     sit in AGPRs: 256 VGPRs do not hold 256 accumulator registers and the addresses);
//   * MFMAs, loads and waits are written out (inline asm) — the register allocator has no slack to play with:
//     through the builtins it kept 18 accumulators in the wrong file and shuttled them through a[0:3].
// Same strip geometry as scan_tiles_i8r_kernel (a unit = 512 rows = two filter tiles; wave w owns rows
// [128 w, +128) of every unit of its stream — a static assignment: the four waves sit on four SIMDs of their own),
// the same blocked shadow, query tile, thresholds, survivor log and pacing counters.  dim % (64 Q_DEPTH) == 0 (the
// register ring is indexed statically: Q_DEPTH slabs per trip of the k loop) — 384 and 768, the dimensions of
// BASELINE.json, with Q_DEPTH = 3.
// -------------------------------------------------------------------------------------------------
constexpr int Q_THREADS = 256;
#ifndef Q_DEPTH_SLABS
#define Q_DEPTH_SLABS 3
#endif
constexpr int Q_DEPTH = Q_DEPTH_SLABS;              // slabs of row fragments in registers (one being multiplied, the others on their way)
constexpr int Q_LDS = R_MAX_SLABS * R_B_SLAB;       // 96 KiB: the query tile
constexpr int Q_BLK_DWORDS = 36;                    // a block entry of the survivor log: 16-byte head + a lane's 32 accumulators of a query block
#define Q_ACC_IN_AGPR(cb) ((cb) >= 6)               // query blocks whose accumulators live in AGPRs

template <int ABL = 0>
__global__ __launch_bounds__(Q_THREADS, 1) void scan_tiles_i8q_kernel(ScanArgs a, uint32_t n_units, uint32_t n_qt, uint32_t n_streams, uint32_t window) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[Q_LDS];

    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, slot = bid >> 3;       // workgroup b runs on XCD b % 8
    const uint32_t qt = slot % n_qt, st = slot / n_qt;    // the query tile it holds, its row stream on this XCD
    if (st >= (n_streams >> 3)) return;
    const uint32_t stream = st * 8u + xcd;
    if (stream >= n_units) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const uint32_t dim = a.dim;
    const int nslab = dim / I8_SLAB; // a multiple of Q_DEPTH, >= 2 Q_DEPTH (checked by the host)
    const uint32_t q0 = qt * R_QUERIES;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) unsigned char*)lds));
    const uint64_t n_blocks = (a.n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
    const uint64_t past_end = n_blocks * I8_BLOCK_ROWS;

    struct Geo {
        uint64_t row0;               // first row of the strip; >= n_rows when the strip does not exist
        const unsigned char* base;   // first piece of the strip (uniform); an absent strip reads the shard's last rows
        uint32_t rbmax;              // last 16-row block of the strip inside the (64-row padded) shadow: the blocks behind it re-read that one
    };
    const uint32_t piece_row_stride = static_cast<uint32_t>(nslab) * 1024u; // bytes between the pieces of consecutive 16-row blocks
    // strip i of this wave: unit stream + i * n_streams, tile (w >> 1) of the unit, rows [128 (w & 1), +128) of the tile
    auto unit_of = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t { return stream + i * n_streams; };
    auto locate = [&](uint32_t i, Geo& g) __attribute__((always_inline)) {
        const uint32_t un_ = unit_of(i);
        const uint32_t sel = un_ < n_units ? 2u * un_ + static_cast<uint32_t>(wid >> 1) : 0xffffffffu;
        uint64_t row0 = past_end;
        if (sel < a.n_sel_tiles) {
            const uint32_t tile = sel + sel / (a.stride - 1u) + 1u;
            row0 = static_cast<uint64_t>(tile) * I8_ROWS + static_cast<uint32_t>((wid & 1) * 128);
        }
        g.row0 = row0;
        // (n_rows >= 4096 on this path.)  The shadow is padded to 64 rows, a strip is 128: the shard's last strip may
        // have only its first 64 rows there — its row blocks 4-7 re-read block 3 (rows >= n_rows: never emitted);
        // a strip that does not exist reads the shard's last 64 rows
        const uint64_t rowb = row0 < past_end ? row0 : past_end - 64;
        g.rbmax = rowb + 128 <= past_end ? 7u : 3u;
        g.base = reinterpret_cast<const unsigned char*>(a.rows_i8) + (rowb / 16) * piece_row_stride;
    };
    typedef float f2_t __attribute__((ext_vector_type(2)));
    // block scales / residue bounds of the strip's two 64-row blocks: {s0, e0, s1, e1}, one scalar load.  The shard's
    // last strip may own the table's last block only: the load is taken one block earlier then (`shifted`) and the
    // strip's first block is the loaded pair's second (its second block has no rows)
    auto meta_ptr = [&](uint64_t row0, bool& shifted) __attribute__((always_inline)) -> const float* {
        uint64_t blk = row0 / I8_BLOCK_ROWS;
        shifted = blk + 1 == n_blocks;
        if (blk + 2 > n_blocks) blk = n_blocks - 2; // (a strip past the end: nothing of it is ever emitted)
        return a.rows_i8_meta + 2ull * blk;
    };

    // ---- prologue: the resident query tile (wave w stages 32 queries of every slab) ----------------------
    {
        const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.q_i8) + static_cast<uint64_t>(q0) * 64;
        const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const int prow = lane >> 2;                        // the query of a piece this lane fetches 16 bytes of
            const int rowB = (wid * 2 + hq) * 16 + prow;
            const uint32_t voffB = static_cast<uint32_t>(rowB) * 64u + ((lane & 3) ^ i8_swz(prow)) * 16u;
            for (int s = 0; s < nslab; ++s)
                lds_dma16_s(baseB + s * qslab_bytes, voffB, __builtin_amdgcn_readfirstlane(lds0 + s * R_B_SLAB + (wid * 2 + hq) * 1024));
        }
    }
    uint32_t* const my_cnt = a.i8_sync + (static_cast<uint64_t>(stream) * 4u + static_cast<uint32_t>(wid)) * 32u;
    const uint32_t* sync_sib = my_cnt + (static_cast<uint32_t>(lane) < n_qt ? static_cast<uint32_t>(lane) : qt);
    Geo cur, nxt;
    locate(0u, cur);
    float sb[2], eb[2];
    {
        bool shifted;
        const float* mp = meta_ptr(cur.row0, shifted);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            sb[b] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[shifted ? 2 : 2 * b])));
            eb[b] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[shifted ? 3 : 2 * b + 1])));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // the only one: the query tile is shared, everything after it is wave-private

    // a lane's 16 bytes of a 1 KiB piece (16 rows x one slab): row l15, 16-byte chunk lq (stored swizzled) — in the
    // query tile's LDS image and in the blocked shadow alike
    const uint32_t offF = static_cast<uint32_t>(l15 * 64 + ((lq ^ i8_swz(l15)) << 4));
    const uint32_t ldsF = lds0 + offF;
    i32x4v acc[8][8];               // [row block][query block]: VGPRs, query blocks 6 and 7 AGPRs
    i32x4v fa[Q_DEPTH][8], fb[2][4]; // AGPRs
    // row block rb of a slab (`sslab` = the strip's piece of that slab, uniform) into a register slot; lands later (see body)
    auto fetch1 = [](i32x4v& dst, uint32_t vo, const unsigned char* sslab) __attribute__((always_inline)) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(dst) : "v"(vo), "s"(sslab) : "memory");
    };
#define YAMS_Q_LDSQ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=a"(DST) : "v"(ADDR) : "memory")
    // {A_lo, B_hi} of this lane's eight queries: loaded once, resident
    f2_t qthr[8];
    {
        const f2_t* qthr_p = reinterpret_cast<const f2_t*>(a.q_thr) + (q0 + l15); // < q_pad: the table is padded
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) qthr[cb] = qthr_p[cb * 16];
    }
    // -T(block b of this strip, query block cb): computed UNDER the strip's MFMAs (sixteen thresholds, four
    // instructions each, one per MFMA slot of the first slab) — after the loop they would be 64 instructions of the
    // serial epilogue
    int nt[2][8];
    float thr_is[2], thr_g[2];
    auto threshold1 = [&](int b, int cb) __attribute__((always_inline)) {
        nt[b][cb] = i8_neg_threshold(qthr[cb][0], thr_is[b], qthr[cb][1], thr_g[b]); // (the same expressions as in i8_log_gather_kernel)
        asm volatile("" : "+v"(nt[b][cb]));
    };
    // 32 MFMAs: the strip's eight row blocks x four query blocks.  FIRST: the strip's first slab — the accumulators
    // are BORN here (C operand = 0)
    auto half = [&](const i32x4v (&A)[8], const i32x4v (&B)[4], auto cb0_tag, auto first_tag, auto&& filler) __attribute__((always_inline)) {
        constexpr bool first = decltype(first_tag)::value;
        constexpr int cb0 = decltype(cb0_tag)::value;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int rb = i >> 2, c = i & 3;
            if (ABL == 2) { // measurement: no MFMAs (the memory pipeline alone)
                if (first) { if (Q_ACC_IN_AGPR(cb0 + c)) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc[rb][cb0 + c][0])); else acc[rb][cb0 + c] = i32x4v{-1, -1, -1, -1}; }
                if (i == 0) asm volatile("" :: "a"(A[0]), "a"(A[1]), "a"(A[2]), "a"(A[3]), "a"(A[4]), "a"(A[5]), "a"(A[6]), "a"(A[7]), "a"(B[0]), "a"(B[1]), "a"(B[2]), "a"(B[3]));
            } else if (Q_ACC_IN_AGPR(cb0 + c)) {
                if (first) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=a"(acc[rb][cb0 + c]) : "a"(A[rb]), "a"(B[c]));
                else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[rb][cb0 + c]) : "a"(A[rb]), "a"(B[c]));
            } else {
                if (first) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=v"(acc[rb][cb0 + c]) : "a"(A[rb]), "a"(B[c]));
                else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[rb][cb0 + c]) : "a"(A[rb]), "a"(B[c]));
            }
            __builtin_amdgcn_sched_barrier(0);
            filler(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C4 = std::integral_constant<int, 4>;

    uint32_t sib = 0;       // pacing: the siblings' strip counters, requested in a strip's last slab
    uint32_t st_prev = 0;   // survivor stores of the previous strip's epilogue (a multiple of 9; anything else: wait them out)
    uint32_t k_cur = 0;
    // One slab (register slot S = slab % Q_DEPTH):
    //   wait:   this slab's row fragments and its query blocks 0-3 have landed (the loads of the slabs behind it are younger);
    //   half 1: fa[S] x query blocks 0-3; requests query blocks 4-7 of this slab and the row fragments of the slab
    //           Q_DEPTH - 1 ahead (into the slot the previous slab has just left);
    //   half 2: fa[S] x query blocks 4-7; requests the next slab's query blocks 0-3.
    // `sslab` names the slab the row fragments are requested from (the last slabs of a strip request the next
    // strip's first ones), `sn` the next slab of the query tile.
    // `stores`: survivor stores of the previous strip's epilogue that may still be in flight in front of this slab's
    // row loads (the first two slabs of a strip; 0 elsewhere): the counter is in order, so the wait allows for them —
    // waiting them out (a store's round trip) cost 0.6 ms of the launch
    struct Slab { const unsigned char* p; uint32_t rbmax; }; // a strip's piece of a slab + the strip's last row block
    auto body = [&](int sn, auto first_tag, Slab sslab, int s, auto slot_tag, bool last, uint32_t stores) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        constexpr int SN = (S + Q_DEPTH - 1) % Q_DEPTH;
        const uint32_t aq = ldsF + static_cast<uint32_t>(s) * R_B_SLAB, aqn = ldsF + static_cast<uint32_t>(sn) * R_B_SLAB;
#define YAMS_Q_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" : "+a"(fa[S][0]), "+a"(fa[S][1]), "+a"(fa[S][2]), "+a"(fa[S][3]), "+a"(fa[S][4]), \
                                    "+a"(fa[S][5]), "+a"(fa[S][6]), "+a"(fa[S][7]), "+a"(fb[0][0]), "+a"(fb[0][1]), "+a"(fb[0][2]), "+a"(fb[0][3]) :: "memory")
        if (ABL == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fb[0][0]), "+a"(fb[0][1]), "+a"(fb[0][2]), "+a"(fb[0][3]) :: "memory");
        else if (Q_DEPTH == 4) YAMS_Q_WAIT(16);
        else if (stores == 9) YAMS_Q_WAIT(17);
        else if (stores == 18) YAMS_Q_WAIT(26);
        else if (stores == 27) YAMS_Q_WAIT(35);
        else YAMS_Q_WAIT(8);
#undef YAMS_Q_WAIT
        half(fa[S], fb[0], C0{}, first_tag, [&](int i) __attribute__((always_inline)) {
            if (i == 0) YAMS_Q_LDSQ(fb[1][0], aq, 4096);
            if (i == 1) YAMS_Q_LDSQ(fb[1][1], aq, 5120);
            if (i == 2) YAMS_Q_LDSQ(fb[1][2], aq, 6144);
            if (i == 3) YAMS_Q_LDSQ(fb[1][3], aq, 7168);
            // the strip's last slab: "this wave has finished strip k_cur" (a slab early; pacing is best effort) and the
            // siblings' counters — OLDER than the eight loads below, so the counted wait at the strip's end covers them
            if (i == 5 && last) {
                unsigned long long keep;
                asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_store_dword %2, %3, off sc0 sc1\n\ts_mov_b64 exec, %1\n\t"
                             "global_load_dword %0, %4, off sc0 sc1"
                             : "=&v"(sib), "=&s"(keep) : "v"(my_cnt + qt), "v"(k_cur + 1u), "v"(sync_sib) : "memory");
            }
            if (ABL != 1 && i >= 6 && i < 14)
                fetch1(fa[SN][(i - 6) & 7], offF, sslab.p + (static_cast<uint32_t>((i - 6) & 7) < sslab.rbmax ? static_cast<uint32_t>((i - 6) & 7) : sslab.rbmax) * piece_row_stride);
            if (decltype(first_tag)::value && i >= 16) threshold1(((i - 16) >> 3) & 1, (i - 16) & 7);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fb[1][0]), "+a"(fb[1][1]), "+a"(fb[1][2]), "+a"(fb[1][3]) :: "memory");
        half(fa[S], fb[1], C4{}, first_tag, [&](int i) __attribute__((always_inline)) {
            if (i == 0) YAMS_Q_LDSQ(fb[0][0], aqn, 0);
            if (i == 1) YAMS_Q_LDSQ(fb[0][1], aqn, 1024);
            if (i == 2) YAMS_Q_LDSQ(fb[0][2], aqn, 2048);
            if (i == 3) YAMS_Q_LDSQ(fb[0][3], aqn, 3072);
        });
    };

    // the first strip's first Q_DEPTH - 1 slabs; the first query fragments
#pragma unroll
    for (int d = 0; d < Q_DEPTH - 1; ++d)
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
            fetch1(fa[d][rb], offF, cur.base + static_cast<uint32_t>(d) * 1024u + (static_cast<uint32_t>(rb) < cur.rbmax ? static_cast<uint32_t>(rb) : cur.rbmax) * piece_row_stride);
    YAMS_Q_LDSQ(fb[0][0], ldsF, 0); YAMS_Q_LDSQ(fb[0][1], ldsF, 1024); YAMS_Q_LDSQ(fb[0][2], ldsF, 2048); YAMS_Q_LDSQ(fb[0][3], ldsF, 3072);

    constexpr uint32_t Q_POLLS = 1024;
    bool pacing = true;
    const uint32_t log_region = (stream * n_qt + qt) * 4u + static_cast<uint32_t>(wid); // (four waves: four regions per workgroup)
    const uint64_t region = static_cast<uint64_t>(log_region) * a.log_cap;
    uint32_t log_pos = 0;

    uint64_t tm_loop = 0, tm_epi = 0, tm_wait = 0, tm_pace = 0; // (100 MHz ticks; measurement only)
    const uint64_t ck_begin = clock64(), wk_begin = wall_clock64(); // shader clocks / 100 MHz ticks over the wave's life: the clock it ran at
    if (unit_of(k_cur) < n_units) for (;;) {
        const uint64_t tm0 = wall_clock64();
        const bool more = unit_of(k_cur + 1) < n_units;
        locate(k_cur + 1, nxt); // (past the end of the stream: the spare loads read the shard's last rows; nobody consumes them)
        float meta_n[4];        // the next strip's block scales, on their way through the scalar cache
        {
            bool shifted;
            const float* mp = meta_ptr(nxt.row0, shifted);
            typedef float f4s __attribute__((ext_vector_type(4)));
            f4s mv;
            asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(mv) : "s"(mp) : "memory");
            asm volatile("" : "+s"(mv));
            meta_n[0] = shifted ? mv[2] : mv[0]; meta_n[1] = shifted ? mv[3] : mv[1]; meta_n[2] = mv[2]; meta_n[3] = mv[3];
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) { thr_is[b] = 1.0f / sb[b]; thr_g[b] = eb[b] * thr_is[b]; }
        // Q_DEPTH slabs per trip (the register slot is a compile-time constant); the strip's first slab stands apart: it
        // gives birth to the accumulators.  Slab s requests the row fragments of slab s + Q_DEPTH - 1.
        using T = std::true_type; using F = std::false_type;
        auto src = [&](int s3) __attribute__((always_inline)) -> Slab { // slab s3 of this strip, or slab s3 - nslab of the next
            return s3 >= nslab ? Slab{nxt.base + static_cast<uint32_t>(s3 - nslab) * 1024u, nxt.rbmax} : Slab{cur.base + static_cast<uint32_t>(s3) * 1024u, cur.rbmax};
        };
        if constexpr (Q_DEPTH == 3) {
            body(1, T{}, src(2), 0, C0{}, false, st_prev);
            body(2, F{}, src(3), 1, C1{}, false, st_prev);
            body(3, F{}, src(4), 2, C2{}, false, 0u);
            int s = 3;
            do { // (nslab >= 6)
                body(s + 1, F{}, src(s + 2), s, C0{}, false, 0u);
                body(s + 2, F{}, src(s + 3), s + 1, C1{}, false, 0u);
                body(s + 3 >= nslab ? 0 : s + 3, F{}, src(s + 4), s + 2, C2{}, s + 3 >= nslab, 0u);
                s += 3;
            } while (s < nslab);
        } else {
            using C3 = std::integral_constant<int, 3>;
            body(1, T{}, src(3), 0, C0{}, false, 0u);
            body(2, F{}, src(4), 1, C1{}, false, st_prev);
            body(3, F{}, src(5), 2, C2{}, false, 0u);
            body(4, F{}, src(6), 3, C3{}, false, 0u);
            int s = 4;
            do { // (nslab >= 8)
                body(s + 1, F{}, src(s + 3), s, C0{}, false, 0u);
                body(s + 2, F{}, src(s + 4), s + 1, C1{}, false, 0u);
                body(s + 3, F{}, src(s + 5), s + 2, C2{}, false, 0u);
                body(s + 4 >= nslab ? 0 : s + 4, F{}, src(s + 6), s + 3, C3{}, s + 4 >= nslab, 0u);
                s += 4;
            } while (s < nslab);
        }
        const uint64_t tm1 = wall_clock64();

        // ---- epilogue of the strip: acc[rb][cb][r] = I of row row0 + 16 rb + 4 lq + r and query q0 + 16 cb + l15; a
        //      survivor is I >= T(64-row block, query), i.e. I + nt >= 0: one maximum per (64-row block, query block)
        //      against its threshold — 8 v_max3 per 16 accumulator registers in VGPRs; 8 v_accvgpr_read + 4 v_max3 per
        //      8 in AGPRs (four independent chains either way) ------------------------------------------------
        const uint64_t strip = cur.row0;
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); // (the last MFMAs' results: they were written by inline asm, no hazard recogniser saw them)
        auto mx3 = [](int x, int y, int z) __attribute__((always_inline)) -> int { const int t = x > y ? x : y; return t > z ? t : z; };
        uint32_t hotw = 0; // bit cb: some lane holds a survivor in query block cb (wave-uniform)
        uint32_t hot = 0;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            int mb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int m;
                if (Q_ACC_IN_AGPR(cb)) {
                    int m0 = static_cast<int>(0x80000000u), m1 = m0, m2 = m0, m3 = m0;
#pragma unroll
                    for (int i = 16 * b; i < 16 * b + 16; i += 8) {
                        int t0, t1, t2, t3, t4, t5, t6, t7;
                        asm volatile("v_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\tv_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15\n\t"
                                     "v_accvgpr_read_b32 %8, %16\n\tv_accvgpr_read_b32 %9, %17\n\tv_accvgpr_read_b32 %10, %18\n\tv_accvgpr_read_b32 %11, %19\n\t"
                                     "v_max3_i32 %0, %4, %5, %0\n\tv_max3_i32 %1, %6, %7, %1\n\tv_max3_i32 %2, %8, %9, %2\n\tv_max3_i32 %3, %10, %11, %3"
                                     : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                                     : "a"(acc[i >> 2][cb][0]), "a"(acc[i >> 2][cb][1]), "a"(acc[i >> 2][cb][2]), "a"(acc[i >> 2][cb][3]),
                                       "a"(acc[(i >> 2) + 1][cb][0]), "a"(acc[(i >> 2) + 1][cb][1]), "a"(acc[(i >> 2) + 1][cb][2]), "a"(acc[(i >> 2) + 1][cb][3]));
                    }
                    m = mx3(m0, m1, m2 > m3 ? m2 : m3);
                } else {
                    const i32x4v &x0 = acc[4 * b][cb], &x1 = acc[4 * b + 1][cb], &x2 = acc[4 * b + 2][cb], &x3 = acc[4 * b + 3][cb];
                    const int m0 = mx3(x0[0], x0[1], x0[2]), m1 = mx3(x0[3], x1[0], x1[1]), m2 = mx3(x1[2], x1[3], x2[0]);
                    const int m3 = mx3(x2[1], x2[2], x2[3]), m4 = mx3(x3[0], x3[1], x3[2]);
                    m = mx3(mx3(m0, m1, m2), mx3(m3, m4, x3[3]), static_cast<int>(0x80000000u));
                }
                mb[b] = m + nt[b][cb];
            }
            const bool h = (mb[0] >= 0 || mb[1] >= 0) && q0 + cb * 16 + l15 < a.n_queries;
            if (h) hot |= 1u << cb;
            if (__builtin_amdgcn_ballot_w64(h) != 0) hotw |= 1u << cb;
        }
        if (strip >= a.n_rows) hotw = 0;
        asm volatile("" : "+s"(hotw));
        const uint64_t tm1b = wall_clock64();
        st_prev = 0;
        if (hotw != 0) {
            // A lane that holds a survivor of query block cb writes ALL its 32 accumulators of that block: a BLOCK entry
            // (Q_BLK_DWORDS dwords: {query, first row, -, -}, then I of element 4 rb + r = row first + 16 rb + r) —
            // a ballot for the slots, nine stores, and the log gather kernel finds the survivors in it (thresholds, the
            // shard's end and the row mask included).  Picking them out here, element by element, was 350 instructions
            // per strip with nothing else running on the SIMD.
            uint32_t base = log_pos;
            int32_t* const blk = reinterpret_cast<int32_t*>(a.log_key) + region * Q_BLK_DWORDS;
            const uint32_t row_first = static_cast<uint32_t>(strip) + 4u * static_cast<uint32_t>(lq);
            do { // one trip per query block that holds a survivor (one or two, typically): ONE copy of the code
                const int cb = __builtin_ctz(hotw);
                hotw &= hotw - 1u;
                const bool p = (hot >> cb) & 1u;
                const uint64_t m = __builtin_amdgcn_ballot_w64(p);
                const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                base += static_cast<uint32_t>(__builtin_popcountll(m));
                const uint32_t qi = q0 + static_cast<uint32_t>(cb) * 16u + l15;
                st_prev += 9;
                if (__builtin_amdgcn_ballot_w64(p && pos >= a.log_cap) != 0) st_prev = 1000; // (an atomic besides the stores: no counting)
                if (p && pos >= a.log_cap) atomicOr(&a.q_over[qi], 1u); // no room: this query's list is incomplete -> exhaustive path
                if (p && pos < a.log_cap) {
                    i32x4v* const ent = reinterpret_cast<i32x4v*>(blk + static_cast<uint64_t>(pos) * Q_BLK_DWORDS);
                    const i32x4v head = {static_cast<int>(qi), static_cast<int>(row_first), 0, 0};
                    ent[0] = head;
#define YAMS_Q_ST8(C) ent[1] = acc[0][C]; ent[2] = acc[1][C]; ent[3] = acc[2][C]; ent[4] = acc[3][C]; \
                      ent[5] = acc[4][C]; ent[6] = acc[5][C]; ent[7] = acc[6][C]; ent[8] = acc[7][C];
                    switch (cb) {
                        case 0: YAMS_Q_ST8(0) break;
                        case 1: YAMS_Q_ST8(1) break;
                        case 2: YAMS_Q_ST8(2) break;
                        case 3: YAMS_Q_ST8(3) break;
                        case 4: YAMS_Q_ST8(4) break;
                        case 5: YAMS_Q_ST8(5) break;
                        case 6: YAMS_Q_ST8(6) break;
                        default: YAMS_Q_ST8(7) break;
                    }
#undef YAMS_Q_ST8
                }
            } while (hotw != 0);
            log_pos = base;
        }
        const uint64_t tm2 = wall_clock64();
        // the next strip's block scales
        sb[0] = meta_n[0]; eb[0] = meta_n[1]; sb[1] = meta_n[2]; eb[1] = meta_n[3];
        // the next strip's first slab and the siblings' counters are older than the eight loads of its last prefetched
        // slab — and than any survivor store of this strip: a counted wait, the stores drain under the next strip's
        // first slab
        if (st_prev == 9) asm volatile("s_waitcnt vmcnt(17)" : "+v"(sib) :: "memory");
        else if (st_prev == 18) asm volatile("s_waitcnt vmcnt(26)" : "+v"(sib) :: "memory");
        else if (st_prev == 27) asm volatile("s_waitcnt vmcnt(35)" : "+v"(sib) :: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(sib) :: "memory");
        const uint64_t tm3 = wall_clock64();
        if (n_qt > 1 && more && pacing) {
            asm volatile("" : "+v"(sib));
            uint32_t polls = 0;
            for (; polls < Q_POLLS; ++polls) {
                if (__builtin_amdgcn_ballot_w64(sib + window < k_cur + 1u) == 0) break;
                __builtin_amdgcn_s_sleep(8);
                asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(sib) : "v"(sync_sib) : "memory");
            }
            if (polls == Q_POLLS) pacing = false;
        }
        tm_loop += tm1 - tm0; tm_epi += tm1b - tm1; tm_wait += tm2 - tm1b; tm_pace += wall_clock64() - tm2; (void)tm3;
        if (!more) break;
        cur = nxt;
        ++k_cur;
    }
#undef YAMS_Q_LDSQ
    if (lane == 0 && n_qt <= 8) { // where the time went, per wave
        uint32_t* const dbg = a.i8_sync + (static_cast<uint64_t>(n_streams) * 4u + static_cast<uint64_t>(stream) * 8u + static_cast<uint32_t>(wid)) * 32u;
        dbg[qt * 4 + 0] = static_cast<uint32_t>(tm_loop); dbg[qt * 4 + 1] = static_cast<uint32_t>(tm_epi);
        dbg[qt * 4 + 2] = static_cast<uint32_t>(tm_wait);
        dbg[qt * 4 + 3] = static_cast<uint32_t>((clock64() - ck_begin) * 100ull / (wall_clock64() - wk_begin + 1)); // MHz (the pace phase's slot)
        (void)tm_pace;
    }
    if (lane == 0) a.log_cnt[log_region] = log_pos < a.log_cap ? log_pos : a.log_cap;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the spare loads of the strip that does not exist)
}


