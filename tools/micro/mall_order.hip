// Micro-benchmark: does the order in which a kernel walks a tensor its predecessor just wrote matter?  (round 6; gfx950; round-5 verdict item 6)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_order.hip -o /tmp/mall_order && /tmp/mall_order
// The 288^2 level moves 172 MB tensors between persistent kernels (conv_ws writes h1, akgm_ws<8> reads h1 + the residual x and writes y): each kernel
// gives workgroup g a contiguous range of tiles and walks it in ascending order.  A 172 MB tensor does not fit the 8 x 4 MB of L2 but it fits the
// 256 MB Infinity Cache - if the memory side keeps what was written last, a consumer that walks its range BACKWARDS meets the producer's most recent
// lines first.  Kernel P (producer): every workgroup reads its range of X and writes its range of Y, ascending.  Kernel C (consumer): reads its range
// of Y and of X and writes Z - ascending or descending, the same ranges or ranges shifted by half the grid (another XCD).  Time of C after P.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512, 1) void prod(const uint4* __restrict__ x, uint4* __restrict__ y, long long per) {
    const long long base = (long long)blockIdx.x * per;
    for (long long i = threadIdx.x; i < per; i += 512) { uint4 v = x[base + i]; v.x += 1; y[base + i] = v; }
}
// dir 0: ascending, 1: descending (chunks of 512 x 16 B walked from the range's end); shift: the range of workgroup (g + shift) % G
__global__ __launch_bounds__(512, 1) void cons(const uint4* __restrict__ x, const uint4* __restrict__ y, uint4* __restrict__ z, long long per, int dir, int shift, int use_x) {
    const int g = (blockIdx.x + shift) % gridDim.x;
    const long long base = (long long)g * per, nch = per / 512;
    for (long long c = 0; c < nch; ++c) {
        const long long i = base + (dir ? nch - 1 - c : c) * 512 + threadIdx.x;
        uint4 v = y[i];
        if (use_x) { const uint4 w = x[i]; v.x ^= w.x; v.y += w.y; }
        z[i] = v;
    }
}

int main() {
    const long long bytes = 172LL << 20, n16 = bytes / 16;
    const int G = 256;
    const long long per = n16 / G / 512 * 512;
    uint4 *x, *y, *z; hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&z, bytes);
    hipMemset(x, 1, bytes); hipMemset(y, 0, bytes); hipMemset(z, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("tensors of %lld MB, %d workgroups with contiguous ranges of %.2f MB; consumer time after the producer (min of 5)\n", bytes >> 20, G, per * 16 / 1048576.0);
    for (int use_x = 0; use_x < 2; ++use_x)
        for (int shift : {0, 128})
            for (int dir = 0; dir < 2; ++dir) {
                float best = 1e30f, bestp = 1e30f;
                for (int r = 0; r < 5; ++r) {
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(prod, dim3(G), dim3(512), 0, 0, x, y, per);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < bestp) bestp = ms;
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(cons, dim3(G), dim3(512), 0, 0, x, y, z, per, dir, shift, use_x);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                const double moved = (double)G * per * 16 * (use_x ? 3 : 2);
                printf("  consumer reads y%s, %s ranges, %-10s: %7.1f us = %5.2f TB/s   (producer %7.1f us = %5.2f TB/s)\n", use_x ? " + x" : "    ",
                       shift ? "shifted" : "the same", dir ? "descending" : "ascending", best * 1e3, moved / (best * 1e-3) / 1e12, bestp * 1e3,
                       (double)G * per * 32 / (bestp * 1e-3) / 1e12);
            }
    return 0;
}
