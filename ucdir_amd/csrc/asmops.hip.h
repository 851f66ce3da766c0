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
template <int N>
__device__ __forceinline__ void lgkm_wait_asm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);                             // no MFMA is scheduled above the wait that covers its operand
}


