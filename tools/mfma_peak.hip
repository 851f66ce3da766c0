// Sustained matrix-core rate of this box: nothing but v_mfma_f32_32x32x16_bf16 on eight independent accumulator tiles per wave,
// two waves per SIMD on every CU (the occupancy of conv_sk_kernel), no memory traffic.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
// Prints TFLOP/s for bursts of different length (the clock the chip holds under matrix load depends on it) - the practical roof the
// conv kernels are compared with in DESIGN.md 4.9 next to the nominal 2.5 PFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_only(int iters, float* out, int random) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a[4], b[2];
    unsigned r = (threadIdx.x + 977u * blockIdx.x) * 2654435761u + 12345u;
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 8; ++e) {
        r = r * 1664525u + 1013904223u;
        // random: values like the conv's operands (weights ~ N(0, 0.02), activations ~ N(0, 1) as sums of uniforms); else small integers
        const float u = ((r >> 8) & 0xffff) / 65536.f + ((r >> 20) & 0xfff) / 4096.f - 1.f;
        a[k][e] = random ? (__bf16)(u * 0.03f) : (__bf16)(float)(threadIdx.x & 3);
        if (k < 2) b[k][e] = random ? (__bf16)(u * 1.7f) : (__bf16)(float)(threadIdx.x & 1);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);   // the conv's 4 x 2 fragment pattern
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int random = 0; random <= 1; ++random)
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
        for (int iters : {500, 2000, 20000, 200000}) {
            const int grid = cus * wg_per_cu;
            hipLaunchKernelGGL(mfma_only<8>, dim3(grid), dim3(256), 0, 0, iters, out, random);   // warm
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mfma_only<8>, dim3(grid), dim3(256), 0, 0, iters, out, random);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double flops = 2.0 * 32 * 32 * 16 * 8.0 * iters * 4 * grid;
            printf("%s operands, CUs %d, %d wave(s)/SIMD, %7d x 8 MFMAs per wave: %9.1f us  %7.1f TFLOP/s  (%.0f MHz if one 32x32x16 = 32 cycles per SIMD)\n", random ? "random" : "small-integer", cus, wg_per_cu, iters, best * 1e3,
                   flops / (best * 1e-3) / 1e12, 8.0 * iters * wg_per_cu * 32 / (best * 1e-3) / 1e6);
        }
    return 0;
}
