// Micro-benchmark: what does a CU's memory pipe deliver to a persistent workgroup?  (round 5; gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
// G workgroups x 512 threads stream a private contiguous slice each.  Per iteration a wave moves U KB (U instructions of 1 KB):
//   mode 0  global_load_dwordx4 into registers (xor-reduced), two iterations in flight
//   mode 1  global_load_lds_dwordx4 into an LDS ring, two iterations in flight (counted vmcnt)
//   mode 2  mode 0 + every loaded KB stored back to a second buffer (copy, R : W = 1 : 1)
//   mode 3  mode 1 + one 1 KB store (from registers) per WR reads of 1 KB (R : W = WR : 1)
//   mode 4  mode 1, but only waves 0-3 issue (twice the pieces each)
// Buffers: 2 GiB (HBM) or 128 MiB (fits the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int MODE, int U, int WR>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long slice, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned char* s = src + (long long)blockIdx.x * slice;
    unsigned char* d = dst + (long long)blockIdx.x * slice;
    constexpr int NWV = MODE == 4 ? 4 : 8;
    constexpr int UU = MODE == 4 ? 2 * U : U;
    const long long step = 8LL * U * 1024;                       // bytes per iteration and workgroup
    const int iters = (int)(slice / step);
    uint4 acc = {0, 0, 0, 0};
    if constexpr (MODE == 0 || MODE == 2) {
        uint4 cur[U], nxt[U];
        const unsigned char* wp = s + (long long)wave * U * 1024 + lane * 16;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = *(const uint4*)(wp + u * 1024);
#pragma unroll 1
        for (int i = 0; i < iters; ++i) {
            const unsigned char* np = wp + (i + 1 < iters ? (long long)(i + 1) * step : 0);
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = *(const uint4*)(np + u * 1024);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (MODE == 2) *(uint4*)(d + (long long)i * step + (long long)wave * U * 1024 + lane * 16 + u * 1024) = cur[u];
                acc.x ^= cur[u].x; acc.y ^= cur[u].y; acc.z ^= cur[u].z; acc.w ^= cur[u].w;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    } else {
        if (wave < NWV) {
            const unsigned lbase = (unsigned)wave * (2 * UU * 1024);
            auto issue = [&](int i) {
                const unsigned off = (unsigned)((long long)i * step) + (unsigned)wave * UU * 1024 + lane * 16;
#pragma unroll
                for (int u = 0; u < UU; ++u) dma16(s, off + u * 1024, (unsigned)__builtin_amdgcn_readfirstlane(lbase + ((i & 1) * UU + u) * 1024));
            };
            issue(0);
            int wcount = 0;
#pragma unroll 1
            for (int i = 0; i < iters; ++i) {
                issue(i + 1 < iters ? i + 1 : 0);
                if constexpr (MODE == 3) {
                    constexpr int NST = (UU + WR - 1) / WR;          // stores per iteration
#pragma unroll
                    for (int t = 0; t < NST; ++t) {
                        *(uint4*)(d + (long long)wcount * 8192 + wave * 1024 + lane * 16) = acc; ++wcount;
                    }
                    vm_wait<UU + NST>();
                } else {
                    vm_wait<UU>();
                }
                acc.x += i;
            }
            vm_wait<0>();
        }
        __syncthreads();
        acc.y ^= *(const unsigned*)(smem + lane * 4);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

static double run_one(void (*kern)(const unsigned char*, unsigned char*, long long, unsigned*), int G, size_t lds, const unsigned char* src, unsigned char* dst,
                      long long total, unsigned* sink, int reps) {
    long long slice = total / G; slice -= slice % (64 * 1024);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, 0, src, dst, slice, sink);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, 0, src, dst, slice, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
    return (double)slice * G * reps / (ms * 1e-3) / 1e12;     // TB/s of READ traffic
}

template <int MODE, int U, int WR = 1> void bench(const char* what, const unsigned char* src, unsigned char* dst, unsigned* sink) {
    const size_t lds = (MODE == 0 || MODE == 2) ? 1024 : (size_t)8 * 2 * U * 1024 + 1024;
    for (long long total : {2LL << 30, 128LL << 20, 16LL << 20}) {
        const int reps = total > (1LL << 30) ? 4 : (total > (64LL << 20) ? 40 : 200);
        printf("%-34s U=%d %4lld MiB:", what, U, total >> 20);
        for (int G : {32, 64, 256, 512, 1024}) {
            if (lds * (G / 256) > 160 * 1024 && G > 256) { printf("      -"); continue; }
            { const double tb = run_one(k<MODE, U, WR>, G, lds, src, dst, total, sink, reps); printf("  G=%4d %5.2f (%4.1f B/ns/CU)", G, tb, tb * 1e3 / (G < 256 ? G : 256)); }
        }
        printf("  TB/s read\n");
    }
}

int main() {
    unsigned char *src, *dst; unsigned* sink;
    hipMalloc(&src, 2LL << 30); hipMalloc(&dst, 2LL << 30); hipMalloc(&sink, 4);
    hipMemset(src, 1, 2LL << 30); hipMemset(dst, 0, 2LL << 30);
    bench<0, 1>("plain loads", src, dst, sink);
    bench<0, 2>("plain loads", src, dst, sink);
    bench<0, 4>("plain loads", src, dst, sink);
    bench<0, 8>("plain loads", src, dst, sink);
    bench<1, 1>("LDS-DMA", src, dst, sink);
    bench<1, 2>("LDS-DMA", src, dst, sink);
    bench<1, 4>("LDS-DMA", src, dst, sink);
    bench<1, 8>("LDS-DMA", src, dst, sink);
    bench<4, 2>("LDS-DMA, 4 waves issue", src, dst, sink);
    bench<4, 4>("LDS-DMA, 4 waves issue", src, dst, sink);
    bench<2, 4>("copy, plain loads + stores", src, dst, sink);
    bench<3, 4, 1>("LDS-DMA + stores 1:1", src, dst, sink);
    bench<3, 4, 2>("LDS-DMA + stores 2:1", src, dst, sink);
    bench<3, 8, 2>("LDS-DMA + stores 2:1", src, dst, sink);
    return 0;
}
