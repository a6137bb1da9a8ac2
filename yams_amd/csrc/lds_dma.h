// lds_dma.h — LDS-DMA (global_load_lds_dwordx4) issue helpers shared by the filter kernels.
//
// The DMAs are issued from inline asm so hipcc does not drain them (it waits vmcnt(0) before any
// ds_read while a builtin LDS-DMA is in flight); completion is tracked by hand with counted
// s_waitcnt vmcnt(N) + s_barrier.  The LDS image is lane-linear per DMA instruction (lane i writes
// 16 bytes at base + 16 i), so bank swizzles are applied to the SOURCE chunk a lane fetches.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace yams_accel {

__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Same, address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: the base moves
// with scalar adds, so no VALU address arithmetic sits between the MFMAs.
__device__ __forceinline__ void lds_dma16_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}


} // namespace yams_accel
