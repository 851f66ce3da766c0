// Inline-asm LDS fragment reads with waits the KERNEL counts (cdna guide 5.7, form iii) and a compile-time loop helper.
#pragma once
#include <type_traits>
#include "common.h"

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N) - the body sees its index as a constant expression
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// LDS fragment read the compiler does not count (cdna guide 5.7 form iii): with hipcc's own bookkeeping the K loop of this
// one-wave-per-SIMD kernel drained the LDS queue (lgkmcnt(0)) every second step - 44 instead of 32 cycles per MFMA
template <int OFF>
__device__ __forceinline__ void lds_read16_asm(bf16x8_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read16f_asm(f32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
// one LDS-DMA piece (16 bytes per lane, 1 KB per wave) with a wave-uniform SGPR base and a 32-bit per-lane byte offset: no 64-bit per-lane pointer to form or
// keep (the builtin only takes the VGPR-address form); M0 = the LDS destination, written - and declared clobbered - inside the statement.  hipcc does not count
// it: the kernels that use it count their vmcnt by hand, and a hidden VMEM operation only makes the compiler's own waits more conservative.
__device__ __forceinline__ void dma16_sbase(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void lgkm_wait_asm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);                             // no MFMA is scheduled above the wait that covers its operand
}



// 16 consecutive floats from LDS as one accumulator tuple: four 16-byte reads CONCATENATED (shufflevector -> REG_SEQUENCE: the reads land in the
// tuple's sub-registers).  Element-wise assignment from four f32x4 temporaries cost 16 v_mov per tile in the AKGM kernels' fold-constant start.
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
__device__ __forceinline__ f32x16_t lds_read_f32x16(const unsigned char* p) {
    const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(p), c1 = *reinterpret_cast<const f32x4_t*>(p + 16);
    const f32x4_t c2 = *reinterpret_cast<const f32x4_t*>(p + 32), c3 = *reinterpret_cast<const f32x4_t*>(p + 48);
    const f32x8_t lo = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(c2, c3, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}
