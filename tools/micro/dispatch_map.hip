// Micro-benchmark: which workgroups of a launch share a CU?  (round 6; gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_map.hip -o /tmp/dispatch_map && /tmp/dispatch_map [blocks] [lds_kb] [threads]
// Every workgroup records HW_REG_HW_ID / HW_REG_XCC_ID of its first wave and its start time, then spins ~30 us so that the whole grid is
// resident at once.  With 80 KB of LDS two workgroups fit a CU: the question is whether the hardware hands the k-th and the (k + 32)-th
// workgroup of an XCD to the same CU (i.e. fills every CU with one workgroup before it gives any CU a second one, in the same CU order) -
// conv_sk's mixed wide / narrow launch at 36^2 orders its units on that assumption.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

__global__ void k(unsigned long long* out, long long spin) {
    extern __shared__ unsigned char smem[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 32)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 32)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) smem[0] = 1;
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hwid; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = t0; out[blockIdx.x * 4 + 3] = smem[0];
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, lds_kb = argc > 2 ? atoi(argv[2]) : 80, threads = argc > 3 ? atoi(argv[3]) : 256;
    unsigned long long* d; hipMalloc(&d, blocks * 4 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_kb * 1024, 0, d, 60000LL);   // s_memtime runs at the shader clock here (~1.7 GHz): ~35 us
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // per XCD (= blockIdx % 8 by observation; checked against XCC_ID): in-XCD index j = blockIdx / 8 -> CU key
    int xcc_mismatch = 0, same = 0, pairs = 0;
    std::map<int, std::vector<unsigned>> per_x;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = (unsigned)h[b * 4], xcc = (unsigned)h[b * 4 + 1] & 0xf;
        if ((int)xcc != b % 8) ++xcc_mismatch;
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_x[b % 8].push_back((se << 8) | (sh << 4) | cu);
        tmin = std::min(tmin, h[b * 4 + 2]); tmax = std::max(tmax, h[b * 4 + 2]);
    }
    for (auto& kv : per_x) {
        auto& v = kv.second;
        std::map<unsigned, int> cnt;
        for (unsigned c : v) cnt[c]++;
        int mx = 0; for (auto& c : cnt) mx = std::max(mx, c.second);
        const int n = (int)v.size(), half = (int)cnt.size();
        int s = 0, p = 0;
        for (int j = 0; j + half < n; ++j) { ++p; if (v[j] == v[j + half]) ++s; }
        same += s; pairs += p;
        printf("XCD %d: %d workgroups on %d distinct CUs (max %d per CU); (j, j + %d) on the same CU: %d of %d\n  order:", kv.first, n, half, mx, half, s, p);
        for (int j = 0; j < n && j < 72; ++j) printf(" %03x", v[j]);
        printf("\n");
    }
    printf("blocks %d, lds %d KB, threads %d: XCC_ID != blockIdx %% 8 for %d blocks; same-CU pairs (j, j + #CUs) %d / %d; start spread %llu ticks\n",
           blocks, lds_kb, threads, xcc_mismatch, same, pairs, tmax - tmin);
    return 0;
}
