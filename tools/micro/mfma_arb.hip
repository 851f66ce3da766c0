// Micro-benchmark: how do the two waves of a SIMD share the matrix pipe?  (round 5; gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_arb.hip -o /tmp/mfma_arb && /tmp/mfma_arb
// 256 workgroups x 512 threads (two waves per SIMD, 256 VGPRs).  Every wave runs NM MFMAs (two independent accumulator chains, operands in
// registers) and records start / end s_memtime and its HW_ID.  Modes: 0 all priority 0; 1 waves 4-7 s_setprio 3; 2 waves 0-3 s_setprio 3;
// 3 priority by HW wave-slot parity; 4 fine-grained yield (s_setprio 0 / 1 toggled around every MFMA pair)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int nm, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (lane + i)); b[i] = (__bf16)(seed * (lane * 3 + i)); }
    f32x16_t c0 = {0}, c1 = {0};
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 32)" : "=s"(hwid));
    if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    if (MODE == 2 && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (MODE == 3 && (hwid & 1)) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
    for (int i = 0; i < nm; i += 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 4) __builtin_amdgcn_s_setprio(1);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            if (MODE == 4) __builtin_amdgcn_s_setprio(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (lane == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = t0; o[1] = t1; o[2] = hwid; o[3] = (unsigned long long)(c0[0] + c1[3]);
    }
}

template <int MODE> void run(int nm) {
    unsigned long long* d; hipMalloc(&d, 256 * 8 * 4 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, nm, 0.001f);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, nm, 0.001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // per workgroup: duration of waves 0-3 (older) and 4-7 (younger), relative to the workgroup's first start
    double older = 0, younger = 0, span = 0; int n = 0;
    for (int g = 0; g < 256; ++g) {
        unsigned long long s = ~0ull, e = 0;
        for (int w = 0; w < 8; ++w) { s = std::min(s, h[(g * 8 + w) * 4]); e = std::max(e, h[(g * 8 + w) * 4 + 1]); }
        for (int w = 0; w < 8; ++w) { const double dt = (double)(h[(g * 8 + w) * 4 + 1] - s); if (w < 4) older += dt; else younger += dt; }
        span += (double)(e - s); ++n;
    }
    printf("mode %d  nm %d: older waves end at %.0f, younger at %.0f, workgroup span %.0f cycles; ideal 2 x nm x 32 = %d\n", MODE, nm, older / (4 * n), younger / (4 * n), span / n, 2 * nm * 32);
    if (MODE == 0) {
        printf("  HW_ID of workgroup 0 (wave: wave_id simd_id):");
        for (int w = 0; w < 8; ++w) printf("  %d: %llu %llu", w, h[w * 4 + 2] & 15, (h[w * 4 + 2] >> 4) & 3);
        printf("\n");
    }
    hipFree(d);
}
int main() {
    for (int nm : {64, 512}) { run<0>(nm); run<1>(nm); run<2>(nm); run<3>(nm); run<4>(nm); }
    return 0;
}
