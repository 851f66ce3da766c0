// Micro-benchmark: a persistent workgroup per CU that alternates "load the next tile by LDS-DMA" / "compute" / "store", as akgm_ws / conv_ws do.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tile_stream.hip -o /tmp/tile_stream && /tmp/tile_stream
// Per tile and wave: P pieces of 1 KB in (for tile t + 1, double-buffered LDS), M MFMAs 32x32x16 (register operands), S stores of 1 KB (full lines).
//   mode 0  all eight waves: pieces in one burst at the tile top, then the MFMAs, then the stores, vmcnt(0), barrier
//   mode 1  pieces and stores spread evenly between the MFMAs
//   mode 2  waves 0-3 do all memory work (2 P pieces, 2 S stores each, burst) and M MFMAs; waves 4-7 only M MFMAs
//   mode 3  mode 2, but the memory waves do NO MFMAs and the compute waves 2 M (dedicated loader waves)
//   mode 4  mode 0 without the tile barrier (every wave waits only for its own pieces)
// Reported: cycles per tile (s_memtime ticks scaled by the measured kernel time), and GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int MODE, int P, int M, int S>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int tiles, long long slice, float seed, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned char* s = src + (long long)blockIdx.x * slice;
    unsigned char* d = dst + (long long)blockIdx.x * slice;
    const bool memw = MODE < 2 || MODE == 4 || wave < 4;
    constexpr int PP = (MODE == 2 || MODE == 3) ? 2 * P : P;
    constexpr int SS = (MODE == 2 || MODE == 3) ? 2 * S : S;
    const int slot = (MODE == 2 || MODE == 3) ? wave : wave;       // memory-wave index
    const int nmw = (MODE == 2 || MODE == 3) ? 4 : 8;
    const int mym = MODE == 3 ? (wave < 4 ? 0 : 2 * M) : M;
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (lane + i)); b[i] = (__bf16)(seed * (lane * 3 + i)); }
    f32x16_t c0 = {0}, c1 = {0};
    uint4 val = {1u, 2u, 3u, (unsigned)lane};
    auto piece = [&](int t, int u) {
        const unsigned off = (unsigned)(((long long)t * nmw + slot) * PP + u) * 1024u + lane * 16;
        dma16(s, off, (unsigned)__builtin_amdgcn_readfirstlane((((t & 1) * nmw + slot) * PP + u) * 1024));
    };
    auto store = [&](int t, int u) {
        *(uint4*)(d + ((long long)(t * nmw + slot) * SS + u) * 1024 + lane * 16) = val;
    };
    if (memw) for (int u = 0; u < PP; ++u) piece(0, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < tiles; ++t) {
        if constexpr (MODE == 1) {
            constexpr int EV = M / (P + S) > 0 ? M / (P + S) : 1;
            int pi = 0, si = 0;
#pragma unroll 1
            for (int i = 0; i < M; i += 2) {
                if (i % EV < 2) { if (pi < P) { piece(t + 1, pi); ++pi; } else if (si < S) { store(t, si); ++si; } }
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
            for (; pi < P; ++pi) piece(t + 1, pi);
            for (; si < S; ++si) store(t, si);
        } else {
            if (memw) {
#pragma unroll
                for (int u = 0; u < PP; ++u) piece(t + 1, u);
            }
#pragma unroll 1
            for (int i = 0; i < mym; i += 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
            if (memw) {
#pragma unroll
                for (int u = 0; u < SS; ++u) store(t, u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MODE != 4) __syncthreads();
        val.x += (unsigned)(c0[0] + c1[1]);
    }
    if ((val.x ^ *(const unsigned*)(smem + lane * 4)) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int P, int M, int S> void bench(const unsigned char* src, unsigned char* dst, unsigned* sink, double mhz) {
    const size_t lds = (size_t)2 * 8 * P * 1024 + 1024;
    auto kern = k<MODE, P, M, S>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("mode %d  P=%2d M=%3d S=%d:", MODE, P, M, S);
    for (int G : {256, 64}) {
        const int tiles = 120;
        const long long slice = (long long)(tiles + 2) * 8 * (P > S ? P : S) * 1024;
        if (slice * G > (3LL << 30)) { printf(" (too big)"); continue; }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, 0, src, dst, tiles, slice, 0.001f, sink);
        hipEventRecord(e0);
        const int reps = 3;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, 0, src, dst, tiles, slice, 0.001f, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
        const double us_tile = ms * 1e3 / reps / tiles;
        const double bytes = (double)G * 8 * (P + S) * 1024;
        printf("   G=%3d %6.2f us/tile = %5.0f cyc @%.0f MHz, %5.2f TB/s, %4.1f B/cyc/CU", G, us_tile, us_tile * mhz, mhz, bytes / us_tile / 1e6, 8.0 * (P + S) * 1024 / (us_tile * mhz));
    }
    printf("\n");
}

int main() {
    unsigned char *src, *dst; unsigned* sink;
    hipMalloc(&src, 3LL << 30); hipMalloc(&dst, 3LL << 30); hipMalloc(&sink, 4);
    hipMemset(src, 1, 3LL << 30); hipMemset(dst, 0, 3LL << 30);
    const double mhz = 2400;
    // compute only / memory only
    bench<0, 1, 160, 0>(src, dst, sink, mhz);
    bench<0, 9, 0, 0>(src, dst, sink, mhz);
    bench<0, 9, 0, 4>(src, dst, sink, mhz);
    // akgm_ws<8>-like tile: 80 KB in, 32 KB out, ~10 k cycles of matrix work per SIMD
    bench<0, 9, 160, 4>(src, dst, sink, mhz);
    bench<1, 9, 160, 4>(src, dst, sink, mhz);
    bench<2, 9, 160, 4>(src, dst, sink, mhz);
    bench<3, 9, 160, 4>(src, dst, sink, mhz);
    bench<4, 9, 160, 4>(src, dst, sink, mhz);
    // half the compute
    bench<0, 9, 80, 4>(src, dst, sink, mhz);
    bench<1, 9, 80, 4>(src, dst, sink, mhz);
    bench<2, 9, 80, 4>(src, dst, sink, mhz);
    bench<3, 9, 80, 4>(src, dst, sink, mhz);
    bench<4, 9, 80, 4>(src, dst, sink, mhz);
    return 0;
}
