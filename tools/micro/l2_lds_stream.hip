// Micro-benchmark: what bounds flash attention's K / V' stream - the L2 -> LDS path or latency x depth?  (round 6; gfx950; round-5 verdict item 2)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_lds_stream.hip -o /tmp/l2_lds_stream && /tmp/l2_lds_stream
// flash_attn2 at N = 1296, C = 512, B = 16: 176 workgroups (one per CU, 146 KB of LDS), the 11 query blocks of a sample - neighbours on one XCD -
// each pull the sample's whole K and V't (2 x 1.33 MB) through LDS in 64 KB tiles, 467 MB per launch, almost all of it out of L2 (97 MB of HBM
// traffic).  The kernel's DMA-only ablation runs 70.6 us = 6.6 TB/s over the chip = 37 GB/s per CU.
// Here G workgroups x 512 threads (XCD-contiguous logical ids as in the kernel) stream the buffer of their group of `share` workgroups
// (`bytes` per group) into LDS by global_load_lds_dwordx4, `depth` tiles of 64 KB in flight per CU, `reps` passes; nothing else runs.
//   mode 0: tile t + depth is requested when tile t has landed (counted vmcnt), no barrier        - the path's own rate
//   mode 1: one __syncthreads per tile (as the kernel: every wave waits for every wave's pieces)   - + the barrier
//   mode 2: depth = 1 and a full drain per tile: request, wait, barrier, request ...               - the kernel's serial K / V't phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* __restrict__ src, long long bytes, int share, int reps, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const unsigned char* s = src + (long long)(lid / share) * bytes;
    const int ntiles = (int)(bytes / 65536);
    auto issue = [&](int t) {                                          // tile t: 64 KB = 8 waves x 8 pieces of 1 KB
        const unsigned off = (unsigned)t * 65536u + (unsigned)wave * 8192u + lane * 16;
        const unsigned lbase = (unsigned)((t % DEPTH) * 65536 + wave * 8192);
#pragma unroll
        for (int u = 0; u < 8; ++u) dma16(s, off + u * 1024, (unsigned)__builtin_amdgcn_readfirstlane(lbase + u * 1024));
    };
    unsigned acc = 0;
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (MODE == 2) {
#pragma unroll 1
            for (int t = 0; t < ntiles; ++t) { issue(t); vm_wait<0>(); __syncthreads(); acc += t; }
        } else {
            for (int t = 0; t < DEPTH - 1 && t < ntiles; ++t) issue(t);
#pragma unroll 1
            for (int t = 0; t < ntiles; ++t) {
                issue(t + DEPTH - 1 < ntiles ? t + DEPTH - 1 : ntiles - 1);
                vm_wait<8 * (DEPTH - 1)>();                            // tile t landed (this wave's pieces)
                if constexpr (MODE == 1) __syncthreads();
                acc += t;
            }
            vm_wait<0>();
            __syncthreads();
        }
    }
    acc ^= *(const unsigned*)(smem + lane * 4);
    if (acc == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH>
static void run(const unsigned char* d, unsigned* sink, int G, long long bytes, int share, int reps, const char* what) {
    hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(G), dim3(512), 150 * 1024, 0, d, bytes, share, reps, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double tot = (double)G * bytes * reps;
    printf("  %-34s G %3d share %2d depth %d x 64 KB: %8.1f us per pass, %6.2f TB/s over the chip, %6.1f GB/s per CU\n", what, G, share, DEPTH,
           best * 1e3 / reps, tot / (best * 1e-3) / 1e12, tot / G / (best * 1e-3) / 1e9);
}

int main() {
    const long long bytes = 2654208 / 65536 * 65536 + 65536;            // K + V't of one sample at N = 1296, C = 512, rounded up to 64 KB tiles (2.69 MB)
    unsigned char* d; hipMalloc(&d, 64 * bytes); hipMemset(d, 1, 64 * bytes);
    unsigned* sink; hipMalloc(&sink, 4);
    const int reps = 8;
    printf("groups of 11 workgroups share a %.2f MB buffer (L2-resident after the first pass: 2 groups = 5.4 MB per XCD)\n", bytes / 1048576.0);
    run<0, 1>(d, sink, 176, bytes, 11, reps, "no barrier");
    run<0, 2>(d, sink, 176, bytes, 11, reps, "no barrier");
    run<1, 2>(d, sink, 176, bytes, 11, reps, "barrier per tile");
    run<2, 1>(d, sink, 176, bytes, 11, reps, "request / drain / barrier");
    run<0, 2>(d, sink, 256, bytes, 16, reps, "no barrier, every CU");
    run<1, 2>(d, sink, 256, bytes, 16, reps, "barrier per tile, every CU");
    run<2, 1>(d, sink, 256, bytes, 16, reps, "request / drain / barrier, every CU");
    printf("one workgroup per buffer (nothing shared: the stream comes through the fabric / Infinity Cache)\n");
    run<0, 2>(d, sink, 64, bytes, 1, reps, "no barrier");
    run<1, 2>(d, sink, 64, bytes, 1, reps, "barrier per tile");
    printf("one CU alone\n");
    run<0, 2>(d, sink, 1, bytes, 1, reps, "no barrier");
    run<2, 1>(d, sink, 1, bytes, 1, reps, "request / drain / barrier");
    return 0;
}
