// scan_i8d_kernel.h — included by scan_i8_kernel.hip (inside namespace yams_accel, after scan_tiles_i8r_kernel).
//
// scan_tiles_i8d_kernel: the resident-query filter with TWO slabs of row fragments in flight per wave.
//
// Round 6 measured where the launch of scan_tiles_i8r_kernel goes (profiles/r06_filter_forms.json): at the bench shape the
// multiply-adds alone take 5.5 ms, the row stream alone 4.3 ms, both together 7.2 — a wave of the direct form has ONE slab
// of row fragments in flight (a register double buffer) and waits for it at the head of every slab: 0.82 us per slab
// against 0.55 us of matrix work for the two waves of a SIMD; at 256 queries x dim 384 (BASELINE config 2) 1.07 us against
// 0.49.  The fix is bytes in flight, and the registers for them come from the QUERY side: the sixteen-register second set
// of query fragments goes (a rotating window of four fragments, refilled three multiply-add groups ahead of their use,
// covers the LDS latency), a third set of row fragments takes its place.  Everything else is the direct form as shipped
// (ZSM 2 | 4 | 64): query tile, threshold halves and survivor buffers resident in LDS, strips drawn per SIMD pair from a
// counter, siblings paced every 2^n-th strip, the boundary's memory-independent work in front of the wait — and the same
// survivor log, so i8_log_gather_wave_kernel and the host code do not change.
//
// The three row-fragment sets rotate with the slab number; their indices must be compile-time constants (a runtime index
// puts the arrays into scratch), so the slab loop is unrolled by three and the form exists for slab counts that are
// multiples of three AND even — dims 384 and 768, the two the product's embedding models use (all-MiniLM 384, the 768-wide
// families) and BASELINE's.  The wait at the head of a slab is COUNTED: `s_waitcnt vmcnt(4)` — the four loads of the slab
// one ahead may stay in flight; a strip's counters (the pair's next strip, the siblings' progress) are requested at the
// head of its last slab, in front of that slab's four loads, so the same count covers them.
template <int DUMMY = 0>
__global__ __launch_bounds__(R_THREADS, 1) void scan_tiles_i8d_kernel(ScanArgs a, uint32_t n_units, uint32_t n_qt, uint32_t n_streams, uint32_t window) {
    const uint32_t pace_mask = (1u << ((window >> 16) & 255u)) - 1u;
    __shared__ __attribute__((aligned(16))) unsigned char lds[R_LDS];

    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, slot = bid >> 3;       // workgroup b runs on XCD b % 8
    const uint32_t qt = slot % n_qt, st = slot / n_qt;    // the query tile it holds, its row stream on this XCD
    if (st >= (n_streams >> 3)) return;
    const uint32_t stream = st * 8u + xcd;
    if (stream >= n_units) return;

#ifdef YAMS_ACCEL_MEASURE
    const uint64_t t_begin = wall_clock64();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const uint32_t dim = a.dim;
    const int nslab = dim / I8_SLAB; // 6 or 12 (checked by the host)
    const uint32_t q0 = qt * R_QUERIES;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds));
    const uint64_t n_blocks = (a.n_rows + I8_BLOCK_ROWS - 1) / I8_BLOCK_ROWS;
    const uint64_t past_end = n_blocks * I8_BLOCK_ROWS;
    const uint32_t piece_row_stride = static_cast<uint32_t>(nslab) * 1024u; // bytes between the pieces of consecutive 16-row blocks

    struct Geo { uint64_t row0; const unsigned char* base; };
    auto unit_of = [&](uint32_t k) __attribute__((always_inline)) -> uint32_t { return stream + (k >> 1) * n_streams; };
    auto locate = [&](uint32_t k, Geo& g) __attribute__((always_inline)) {
        const uint32_t un_ = unit_of(k);
        const uint32_t sel = un_ < n_units ? 2u * un_ + (k & 1u) : 0xffffffffu;
        uint64_t row0 = past_end;
        if (sel < a.n_sel_tiles) {
            const uint32_t tile = sel + sel / (a.stride - 1u) + 1u;
            row0 = static_cast<uint64_t>(tile) * I8_ROWS + static_cast<uint32_t>((wid & 3) * 64);
        }
        g.row0 = row0;
        const uint64_t rowb = row0 < past_end ? row0 : past_end - 64; // (n_rows >= 4096 on this path)
        g.base = reinterpret_cast<const unsigned char*>(a.rows_i8) + (rowb / 16) * piece_row_stride;
    };
    typedef float f2_t __attribute__((ext_vector_type(2)));
    auto meta_ptr = [&](uint64_t row0) __attribute__((always_inline)) -> const float* {
        uint64_t blk = row0 / I8_BLOCK_ROWS;
        if (blk >= n_blocks) blk = n_blocks - 1;
        return a.rows_i8_meta + 2ull * blk;
    };

    // ---- prologue: the resident query tile (wave w stages 16 queries of every slab) and its threshold halves ----------------
    {
        const int prow = lane >> 2;
        const int rowB = wid * 16 + prow;
        const uint32_t voffB = static_cast<uint32_t>(rowB) * 64u + ((lane & 3) ^ i8_swz(prow)) * 16u;
        const unsigned char* baseB = reinterpret_cast<const unsigned char*>(a.q_i8) + static_cast<uint64_t>(q0) * 64;
        const uint64_t qslab_bytes = static_cast<uint64_t>(a.q_pad) * 64;
        for (int s = 0; s < nslab; ++s)
            lds_dma16_s(baseB + s * qslab_bytes, voffB, __builtin_amdgcn_readfirstlane(lds0 + s * R_B_SLAB + wid * 1024));
    }
    if (tid < R_QUERIES) reinterpret_cast<f2_t*>(lds + R_ZS_THR)[tid] = reinterpret_cast<const f2_t*>(a.q_thr)[q0 + static_cast<uint32_t>(tid)];
    // work sharing and pacing: as in scan_tiles_i8r_kernel (one strip counter per SIMD pair and query tile)
    uint32_t* const pair_cnt = a.i8_sync + (static_cast<uint64_t>(stream) * 4u + static_cast<uint32_t>(wid & 3)) * 32u;
    uint32_t take_v;
    auto take_request = [&]() __attribute__((always_inline)) {
        unsigned long long keep;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                     : "=&v"(take_v), "=&s"(keep) : "v"(pair_cnt + qt), "v"(1u) : "memory");
    };
    auto take_result = [&]() __attribute__((always_inline)) -> uint32_t {
        asm volatile("" : "+v"(take_v));
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(take_v)));
    };
    uint32_t k_cur, k_nxt, k_fut;
    take_request(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k_cur = take_result();
    take_request(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k_nxt = take_result();
    take_request(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k_fut = take_result();
    Geo cur, nxt;
    locate(k_cur, cur);
    float sb, eb;
    {
        const float* mp = meta_ptr(cur.row0);
        const float m0 = mp[0], m1 = mp[1];
        sb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m0)));
        eb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m1)));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // the only one: the query tile is shared, everything after it is wave-private

    const int offF = l15 * 64 + ((lq ^ i8_swz(l15)) << 4);
    i32x4v acc[4][8];
    i32x4v fa[3][4], fb[4];
    auto dfetch = [](i32x4v& dst, uint32_t vo, const unsigned char* src) __attribute__((always_inline)) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(vo), "s"(src) : "memory");
    };
    auto ldq = [&](int slab, int cb) __attribute__((always_inline)) -> i32x4v {
        return *reinterpret_cast<const i32x4v*>(lds + slab * R_B_SLAB + offF + cb * 1024);
    };
    // pipeline fill: slabs 0 and 1 of the first strip on their way, query blocks 0-2 of slab 0 in the window
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) dfetch(fa[0][rb], static_cast<uint32_t>(offF), cur.base + static_cast<uint32_t>(rb) * piece_row_stride);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) dfetch(fa[1][rb], static_cast<uint32_t>(offF), cur.base + 1024u + static_cast<uint32_t>(rb) * piece_row_stride);
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) fb[cb] = ldq(0, cb);

    constexpr uint32_t R_POLLS = 1024;
    bool pacing = true;
    const uint32_t R_WINDOW = window & 0xffffu;
    const uint32_t* sync_sib = pair_cnt + (static_cast<uint32_t>(lane) < n_qt ? static_cast<uint32_t>(lane) : qt);
#ifdef YAMS_ACCEL_MEASURE
    uint32_t units_read = 0;
#endif
    uint32_t q_live = 0;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) q_live |= (q0 + static_cast<uint32_t>(cb * 16 + l15) < a.n_queries) ? 1u << cb : 0u;

    const f2_t* const thr_lds = reinterpret_cast<const f2_t*>(lds + R_ZS_THR) + l15;
    auto acc_init = [&](float sb_, float eb_) __attribute__((always_inline)) { // -T(strip, query block) into all of its accumulators
        const float is = 1.0f / sb_, g = eb_ * is; // (the same expressions as in i8_log_gather_kernel)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const f2_t h = thr_lds[cb * 16];
            const int nt = i8_neg_threshold(h[0], is, h[1], g);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rb][cb][r] = nt;
        }
    };
    const uint32_t log_region = (stream * n_qt + qt) * 8u + static_cast<uint32_t>(wid);
    const uint64_t region = static_cast<uint64_t>(log_region) * a.log_cap;
    uint32_t log_pos = 0;
    uint64_t* const zs_key = reinterpret_cast<uint64_t*>(lds + R_ZS_LOG + wid * (R_ZS_ENTRIES * 12));
    uint32_t* const zs_q = reinterpret_cast<uint32_t*>(zs_key + R_ZS_ENTRIES);
    uint32_t zs_n = 0;
    auto zs_flush = [&]() __attribute__((always_inline)) {
        for (uint32_t i = static_cast<uint32_t>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); i < zs_n; i += 64u) {
            const uint64_t key = zs_key[i];
            const uint32_t qi = zs_q[i];
            const uint32_t pos = log_pos + i;
            if (pos < a.log_cap) { a.log_key[region + pos] = key; a.log_q[region + pos] = qi; }
            else atomicOr(&a.q_over[qi], 1u);
        }
        log_pos += zs_n;
        zs_n = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the k loop's waits count loads only)
    };

    acc_init(sb, eb);
    uint32_t sib = 0;
    // One slab with row-fragment set J (= slab % 3): eight groups of four multiply-adds, group c = query block c against the
    // four row blocks.  Behind group c: row block c of the slab TWO ahead is requested into set (J + 2) % 3 (c < 4), and
    // query block c + 3 (of this slab, or of the next one: the window holds four fragments, the one refilled was used by
    // group c - 1).  `a2` = where the slab two ahead lives, `sq` / `sqn` = this and the next slab of the query tile.
    auto slab = [&](auto j_tag, const unsigned char* a2, int sq, int sqn, auto&& head) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
        constexpr int J2 = (J + 2) % 3;
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(fa[J][0]), "+v"(fa[J][1]), "+v"(fa[J][2]), "+v"(fa[J][3]) :: "memory");
        head();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                // (accumulate IN PLACE: left to itself the register allocator lets the 128 accumulators wander — destination
                //  != C — and pays for it with copies and spills of registers whose loads are still in flight)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[rb][c]) : "v"(fa[J][rb]), "v"(fb[c & 3]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c < 4) dfetch(fa[J2][c], static_cast<uint32_t>(offF), a2 + static_cast<uint32_t>(c) * piece_row_stride);
            fb[(c + 3) & 3] = c + 3 < 8 ? ldq(sq, c + 3) : ldq(sqn, c + 3 - 8);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2t = std::integral_constant<int, 2>;

    if (unit_of(k_cur) < n_units) for (;;) {
        const bool more = unit_of(k_nxt) < n_units;
        locate(k_nxt, nxt); // (past the end of the stream: the spare loads read the shard's last rows; nobody consumes them)
        unsigned long long meta_n;  // the next strip's block scale, on its way through the scalar cache
        {
            const float* mp = meta_ptr(nxt.row0);
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(meta_n) : "s"(mp) : "memory");
        }
        const bool pace_now = n_qt > 1 && (k_cur & pace_mask) == 0u;
        // where the slab two ahead of slab s lives: this strip's, or (the last two slabs) the next strip's slabs 0 and 1
        auto two_ahead = [&](int s) __attribute__((always_inline)) -> const unsigned char* {
            return s + 2 < nslab ? cur.base + static_cast<uint32_t>(s + 2) * 1024u : nxt.base + static_cast<uint32_t>(s + 2 - nslab) * 1024u;
        };
        asm volatile("s_nop 4" ::: "memory"); // (the accumulators were just written by moves; the multiply-adds below are inline asm: no automatic wait states)
        for (int s = 0; s < nslab; s += 3) { // (nslab = 6 or 12)
            const bool tail = s + 3 >= nslab;
            slab(J0{}, two_ahead(s), s, s + 1, [] {});
            slab(J1{}, two_ahead(s + 1), s + 1, s + 2, [] {});
            // the strip's counters — this pair's strip after the next two, the siblings' progress — in front of the LAST slab's
            // loads: the boundary's counted wait covers them and leaves only those four loads in flight
            slab(J2t{}, two_ahead(s + 2), s + 2, tail ? 0 : s + 3, [&]() __attribute__((always_inline)) {
                if (tail) {
                    take_request();
                    if (pace_now) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(sib) : "v"(sync_sib) : "memory");
                }
            });
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+s"(meta_n)); // (the last multiply-add's result is read by the sign test below)
#ifdef YAMS_ACCEL_MEASURE
        ++units_read;
#endif
        // ---- the strip's end: acc[rb][cb][r] is row = row0 + 16 rb + 4 lq + r, query = q0 + 16 cb + l15; the accumulators hold
        //      I - T, a survivor is a non-negative one -----------------------------------------------------------------------
        const uint64_t strip = cur.row0;
        uint32_t hot = 0;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            int m = acc[0][cb][0];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = acc[rb][cb][r] > m ? acc[rb][cb][r] : m;
            if (m >= 0) hot |= 1u << cb;
        }
        hot &= q_live;
        if (strip >= a.n_rows) hot = 0;
        if (__builtin_amdgcn_ballot_w64(hot != 0) != 0) { // about half of the strips hold a survivor somewhere
            const uint32_t rows_left = strip < a.n_rows ? static_cast<uint32_t>(a.n_rows - strip < 64 ? a.n_rows - strip : 64) : 0u;
            uint32_t base = zs_n;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const bool hot_cb = (hot >> cb) & 1u;
                if (__builtin_amdgcn_ballot_w64(hot_cb) == 0) continue;
                const uint32_t qi = q0 + cb * 16 + l15;
                uint32_t pm = 0;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const uint32_t off0 = 16 * rb + 4 * lq;
                    uint32_t mw = 0xfu;
                    if (a.row_mask) { const uint64_t rbase = strip + off0; mw = mask_word(a.row_mask, rbase & ~31ull, a.n_rows) >> (rbase & 31u); }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pm |= (acc[rb][cb][r] >= 0 && off0 + r < rows_left && ((mw >> r) & 1u)) ? 1u << (4 * rb + r) : 0u;
                }
                if (!hot_cb) pm = 0;
                for (;;) { // one trip per survivor of the busiest lane (one, typically)
                    const bool p = pm != 0;
                    const uint64_t m = __builtin_amdgcn_ballot_w64(p);
                    if (m == 0) break;
                    if (base + 64u > static_cast<uint32_t>(R_ZS_ENTRIES)) { zs_n = base; zs_flush(); base = 0; }
                    const int e = p ? __builtin_ctz(pm) : 0;
                    int val = acc[0][cb][0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) val = e == i ? acc[i >> 2][cb][i & 3] : val;
                    const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                    base += static_cast<uint32_t>(__builtin_popcountll(m));
                    if (p) {
                        const uint64_t row = strip + static_cast<uint32_t>(16 * (e >> 2) + 4 * lq + (e & 3));
                        zs_key[pos] = (static_cast<uint64_t>(static_cast<uint32_t>(val)) << 32) | static_cast<uint32_t>(row);
                        zs_q[pos] = qi;
                    }
                    pm &= pm - 1u;
                }
            }
            zs_n = base;
        }
        // the next strip's accumulators, then the counted wait: its first slab and the counters have landed, its second slab
        // (four loads) stays in flight
        sb = __uint_as_float(static_cast<uint32_t>(meta_n));
        eb = __uint_as_float(static_cast<uint32_t>(meta_n >> 32));
        acc_init(sb, eb);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t k_new = take_result();
        if (more && pacing && pace_now) {
            asm volatile("" : "+v"(sib));
            uint32_t polls = 0;
            for (; polls < R_POLLS; ++polls) {
                if (__builtin_amdgcn_ballot_w64(sib + R_WINDOW < k_new) == 0) break;
                __builtin_amdgcn_s_sleep(8);
                asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(sib) : "v"(sync_sib) : "memory");
            }
            if (polls == R_POLLS) pacing = false;
        }
        if (!more) break;
        cur = nxt;
        k_cur = k_nxt; k_nxt = k_fut; k_fut = k_new;
    }
    // The spare loads of the last strip (the slabs 0 and 1 of a strip that does not exist) are still on their way into two of
    // the fragment sets: they must land BEFORE anything else is allowed to live in those registers — the flush below computes
    // addresses, and a register the allocator hands it while a load is still due would be overwritten under it (a stray store
    // far outside the log: found as a bus error on shards of 4M rows and more, never on the small ones).  The "+v" operands
    // keep all twelve registers alive up to this wait.
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]),
                 "+v"(fa[1][3]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[2][2]), "+v"(fa[2][3]) :: "memory");
    zs_flush();
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) a.log_cnt[log_region] = log_pos < a.log_cap ? log_pos : a.log_cap;
#ifdef YAMS_ACCEL_MEASURE
    if (lane == 0 && n_qt <= 8) {
        uint32_t* const dbg = a.i8_sync + (static_cast<uint64_t>(n_streams) * 4u + static_cast<uint64_t>(stream) * 8u + static_cast<uint32_t>(wid)) * 32u;
        dbg[8 + qt] = static_cast<uint32_t>(t_begin);
        dbg[16 + qt] = static_cast<uint32_t>(wall_clock64());
        dbg[24 + qt] = units_read;
    }
#endif
}
