// v_permlane16_swap_b32 / v_permlane32_swap_b32 as a 4-lane maximum over lanes l, l ^ 16, l ^ 32, l ^ 48 (flash_attn2.hip.h softmax row max)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    float mx = (float)((threadIdx.x * 37) % 64);
    { float u = mx, v = mx; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(v)); mx = fmaxf(u, v); }
    { float u = mx, v = mx; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v)); mx = fmaxf(u, v); }
    o[threadIdx.x] = mx;
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float want = 0;
        for (int m = 0; m < 64; m += 16) { const float v = (float)((((l & 15) + m) * 37) % 64); want = v > want ? v : want; }
        if (h[l] != want) ++bad;
    }
    printf("permlane row max: %d lanes wrong\n", bad);
    return bad != 0;
}
