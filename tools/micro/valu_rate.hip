// Micro-benchmark: issue rate of VALU instruction classes on one SIMD (gfx950).  hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/vr && /tmp/vr
// One workgroup per CU, W waves per SIMD; every wave runs N x 8 independent instructions of one class; cycles per instruction and SIMD from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* t, int n, float seed) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (threadIdx.x + i + 1);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 3) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&v[i & 6]));
            if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name) {
    float* o; unsigned long long* t; hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&t, 256 * 8);
    for (int waves : {4, 8, 16}) {             // waves per CU: 1, 2, 4 per SIMD
        const int n = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, o, t, n, 0.001f);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, o, t, n, 0.001f);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-22s %d wave(s) per SIMD: %6.2f cycles per instruction and SIMD (s_memtime ticks)\n", name, waves / 4, avg / (n * 8.0 * (waves / 4)));
    }
}
int main() {
    run<0>("v_fma_f32"); run<3>("v_mul_f32"); run<1>("v_exp_f32"); run<2>("v_rcp_f32"); run<4>("v_pk_fma_f32"); run<5>("v_cvt_pk_bf16_f32");
    return 0;
}
