// scan_i8q_gather.h — MEASUREMENT BUILD ONLY (included by scan_i8_kernel.hip under YAMS_ACCEL_MEASURE, inside
// namespace yams_accel, after i8_log_gather_wave_kernel): the survivor log of scan_tiles_i8q_kernel.
// The same for the BLOCK entries of scan_tiles_i8q_kernel: an entry is a lane's 32 accumulators I of one query
// block — element i = row first + 16 (i >> 2) + (i & 3) — and the survivors are picked out HERE: I >= T of the
// row's 64-row block (the same instructions as everywhere: i8_neg_threshold), inside the shard, allowed by the row
// mask.  One thread per element; the two passes (count, place) as above.
__global__ __launch_bounds__(256) void i8_log_gather_blocks_kernel(const int32_t* log_blk, const uint32_t* log_cnt, uint32_t log_cap,
                                                                   uint32_t n_qt, const float* rows_meta, const float* q_meta,
                                                                   const float* q_thr, uint64_t n_rows, const uint32_t* row_mask,
                                                                   uint32_t* list_count, uint64_t* list, uint32_t list_cap) {
    __shared__ uint32_t hist[R_QUERIES], slot0[R_QUERIES];
    const uint32_t r = blockIdx.x;
    const uint32_t n = log_cnt[r];
    if (n == 0) return;
    const uint32_t q0 = ((r >> 2) % n_qt) * R_QUERIES;
    const int tid = threadIdx.x;
    if (tid < R_QUERIES) hist[tid] = 0u;
    __syncthreads();
    const int32_t* ents = log_blk + static_cast<uint64_t>(r) * log_cap * Q_BLK_DWORDS;
    auto element = [&](uint32_t j, uint32_t& q, uint32_t& row, float& u) -> bool {
        const int32_t* e = ents + static_cast<uint64_t>(j >> 5) * Q_BLK_DWORDS;
        const uint32_t i = j & 31u;
        q = static_cast<uint32_t>(e[0]);
        row = static_cast<uint32_t>(e[1]) + 16u * (i >> 2) + (i & 3u);
        if (row >= n_rows) return false;
        if (row_mask && !((row_mask[row >> 5] >> (row & 31u)) & 1u)) return false;
        const int I = e[4 + i];
        const float2 m = reinterpret_cast<const float2*>(rows_meta)[row / I8_BLOCK_ROWS];
        const float2 qt = reinterpret_cast<const float2*>(q_thr)[q];
        const float is = 1.0f / m.x;
        if (I + i8_neg_threshold(qt.x, is, qt.y, m.y * is) < 0) return false;
        const float4 qm = reinterpret_cast<const float4*>(q_meta)[q];
        u = fmaf(static_cast<float>(I), m.x * qm.x, fmaf(m.y, qm.y, qm.z));
        return true;
    };
    for (uint32_t j = tid; j < n * 32u; j += 256) {
        uint32_t q, row; float u;
        if (element(j, q, row, u)) atomicAdd(&hist[q - q0], 1u);
    }
    __syncthreads();
    if (tid < R_QUERIES) {
        const uint32_t c = hist[tid];
        slot0[tid] = c ? atomicAdd(&list_count[q0 + tid], c) : 0u;
        hist[tid] = 0u;
    }
    __syncthreads();
    for (uint32_t j = tid; j < n * 32u; j += 256) {
        uint32_t q, row; float u;
        if (!element(j, q, row, u)) continue;
        const uint32_t pos = slot0[q - q0] + atomicAdd(&hist[q - q0], 1u);
        if (pos < list_cap) list[static_cast<uint64_t>(q) * list_cap + pos] = pack_key(u, row);
    }
}

