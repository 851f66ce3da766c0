// AKGM tail at 8 channels per group (C = 64, the 288^2 level) as a PERSISTENT, weight-stationary kernel (gfx950).
// Reference: model/ucdir.py:129-140.
//
// Why (round 3): akgm_pre_kernel<8> launches 10,368 workgroups per B = 16 layer; each DMAs 40 KB of unit weights, its
// fold tables and its halo before its first MFMA and lives 20-24 k cycles for 2.5 k cycles of matrix-core work.  The
// L2 -> LDS fill volume (0.9 GB per launch, three quarters of it weights) and the exposed first-byte latency put the
// launch at 2.7x its HBM floor.  Here
//   * the grid is ONE workgroup per CU (8 wave64, two per SIMD, 256 VGPRs each); a workgroup walks a contiguous
//     range of pixel tiles (XCD-contiguous, so neighbouring tiles share an L2);
//   * wave g owns GROUP g for the whole launch: its 64 weight rows x K = 72 (+ one zero tap) sit in 40 VGPRs as
//     MFMA A fragments, loaded once - the K loop reads only B fragments (its group's 16 bytes of each halo pixel)
//     from LDS, there is no weight traffic, no ring and no barrier inside a tile;
//   * the halo (all 64 channels: 128-byte LDS rows, 16-byte chunk XOR (pixel >> 1) & 7 on the source side) and the
//     guide weights G of tile t + 1 are requested by LDS-DMA at the top of tile t into the second buffer: ONE barrier
//     per tile, and the wait in front of it is counted (the tile's own stores stay in flight);
//   * GroupNorm-fold table Tc[9][512] of the current sample resident in LDS (reloaded when the range crosses a
//     sample); accumulators start at it, modulation sum / half-wave exchange / swish / residual / 16-byte store as in
//     akgm_pre.hip.h;
//   * GroupNorm statistics of the output: a lane's per-tile fp32 partial is converted to 2^-20 fixed point and
//     accumulated in 64-bit integers, so the totals do not depend on how tiles are dealt to workgroups (bit-identical
//     for any batch size or CU count) and no barrier is needed for them.
#pragma once
#include "akgm_pre.hip.h"

struct AkWs {
    static constexpr int HALO_PX = 336;                           // 324 rounded up to 16: the read swizzle (pixel >> 1) & 7 is the same in both buffers
    static constexpr int HALO = HALO_PX * 128;                    // 43,008: [336 px][64 ch] bf16
    static constexpr int ATT = 256 * 32;                          // [256 px][8] fp32
    static constexpr int OFF_ATT = 2 * HALO;
    static constexpr int OFF_TCS = OFF_ATT + 2 * ATT;             // [9][512] fp32
    static constexpr int OFF_SCAL = OFF_TCS + 9 * 512 * 4;
    static constexpr int LDS = OFF_SCAL + 128;                    // 120,960: one workgroup per CU
};

__device__ __forceinline__ void stat_add_fx(stat_t* stats, int b, stat_t s, stat_t q) {
    stat_t* dst = stats + ((long long)b * UCDIR_STAT_SLOTS + (blockIdx.x % UCDIR_STAT_SLOTS)) * 2;
    (void)__hip_atomic_fetch_add(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_add(dst + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ stat_t wave_sum_ll(stat_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
// 16-byte global load the compiler does not know about: it is absent from hipcc's vmcnt bookkeeping (no vmcnt(0) in front of
// its first use that would drain the LDS-DMA queue) and its result is only touched behind ws_wait_res (cdna guide 5.7 form ii)
__device__ __forceinline__ u32x4_t asm_load16(const void* ptr) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// Wait until at most n VMEM operations of this wave are outstanding, THEN hand the asm-loaded registers to the compiler.  The
// wait statements carry no register operand on purpose: with "+v"(v) on the s_waitcnt itself hipcc allocated the operand
// elsewhere and copied the load's destination registers IN FRONT of the wait (v_mov_b64 v[0:1], v[116:117]; s_waitcnt ...):
// garbage whenever the load had not landed yet (found by test_akgm_persistent[level0]: the last tiles of the longer ranges).
// A copy in front of the empty statement below is behind the wait and harmless.
__device__ __forceinline__ void ws_wait_res(int n, u32x4_t& v) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    asm volatile("" : "+v"(v) :: "memory");
}

__global__ __launch_bounds__(HC_THREADS, 2) void akgm_ws_kernel(const AkgmHP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo = smem;
    unsigned char* const attb = smem + AkWs::OFF_ATT;
    float* const tcs = reinterpret_cast<float*>(smem + AkWs::OFF_TCS);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = group
    const int U = wave >> 1, wm = wave & 1;                       // unit / row half of pack_akgm_pre
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tps = p.tiles_x * p.tiles_y;
    const int T = p.nbatch * tps;
    const int t_beg = (int)((long long)lid * T / (int)gridDim.x), t_end = (int)((long long)(lid + 1) * T / (int)gridDim.x);
    if (t_beg >= t_end) return;
    const int th = p.th, tw = p.tw, hw = tw + 2;
    const int hcount = (th + 2) * hw, nslots = th * tw;
    const float inv_hw = 1.0f / (float)hw, inv_tw = 1.0f / (float)tw;
    const bool even = (p.H % th == 0) && (p.W % tw == 0) && nslots == 256;   // every tile full: no masks, no clamps

    // ---- this wave's weights: A fragments of pack_akgm_pre's image, resident for the whole launch -------------------
    bf16x8_t af[2][5];
    {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + (long long)U * AkPre<8>::A_UNIT + (hh * 128 + wm * 64 + l31) * 16;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int j = 0; j < 5; ++j) af[tm][j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * 4096 + tm * 512);
    }
    // the compiler's wait for these loads goes HERE (an asm statement reading them), not in front of their first use inside the
    // tile loop, where a vmcnt(0) per pixel pair would drain the next tile's DMA and the previous pair's store
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(af[tm][j]));

    // ---- tile-invariant lane constants ---------------------------------------------------------------------------------
    int hrel[6];                                   // halo piece i*8 + wave: this lane's 16 bytes, element offset from the tile's halo origin
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hp = (i * 8 + wave) * 8 + (lane >> 3);
        hrel[i] = -1;
        if (hp < hcount) {
            const int hr = fdiv_small(hp, inv_hw), hc = hp - hr * hw;
            hrel[i] = (hr * p.Wp + hc) * 64 + (((lane & 7) ^ ((hp >> 1) & 7)) << 3);
        }
    }
    int grel;                                      // guide piece `wave`: pixels 32 wave + (lane >> 1), half lane & 1
    {
        const int slot = wave * 32 + (lane >> 1);
        const int r = fdiv_small(slot < nslots ? slot : nslots - 1, inv_tw), c = (slot < nslots ? slot : nslots - 1) - r * tw;
        grel = (r * p.W + c) * 8 + (lane & 1) * 4;
    }
    int shj[5];                                    // k step j: tap 2j (lanes 0-31) | 2j + 1 (lanes 32-63; tap 9 has zero weights: reads tap 8)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int t = (2 * j + hh > 8) ? 8 : 2 * j + hh;
        shj[j] = (t / 3) * hw + (t % 3);
    }
    const int tc_lane = 8 * (16 * U + 8 * wm + 2 * hh);            // + 32 tm: first of this lane's 16 table entries per row tile

    auto decode = [&](int t, int& b, int& y0, int& x0) {
        b = t / tps;
        const int r = t - b * tps;
        const int ty = r / p.tiles_x;
        y0 = ty * th; x0 = (r - ty * p.tiles_x) * tw;
    };
    auto issue_tile = [&](int t, int buf) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const bf16_t* hb = p.h + (long long)b * p.h_bstride + (long long)(y0 * p.Wp + x0) * 64;
        unsigned char* hd = halo + buf * AkWs::HALO;
        const bool clampd = !even && ((y0 + th > p.H) || (x0 + tw > p.W));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if ((i * 8 + wave) * 8 < hcount) {                   // wave-uniform
                if (!clampd) {
                    if (hrel[i] >= 0) stage16(hb + hrel[i], hd + (i * 8 + wave) * 1024, lane);
                } else {
                    const int hp = (i * 8 + wave) * 8 + (lane >> 3);
                    if (hp < hcount) {
                        const int hr = fdiv_small(hp, inv_hw), hc = hp - hr * hw;
                        int gy = y0 + hr, gx = x0 + hc;
                        gy = gy > p.H + 1 ? p.H + 1 : gy;
                        gx = gx > p.W + 1 ? p.W + 1 : gx;
                        stage16(p.h + (long long)b * p.h_bstride + (long long)(gy * p.Wp + gx) * 64 + (((lane & 7) ^ ((hp >> 1) & 7)) << 3),
                                hd + (i * 8 + wave) * 1024, lane);
                    }
                }
            }
        }
        const float* gb = p.G + (long long)b * p.g_bstride;
        if (!clampd) {
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gb + (long long)(y0 * p.W + x0) * 8 + grel),
                                             (LDS_AS void*)(attb + buf * AkWs::ATT + wave * 1024), 16, 0, 0);
        } else {
            int slot = wave * 32 + (lane >> 1);
            slot = slot < nslots ? slot : nslots - 1;
            const int r = fdiv_small(slot, inv_tw), c = slot - r * tw;
            int y = y0 + r, x = x0 + c;
            y = y < p.H ? y : p.H - 1; x = x < p.W ? x : p.W - 1;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gb + (long long)(y * p.W + x) * 8 + (lane & 1) * 4),
                                             (LDS_AS void*)(attb + buf * AkWs::ATT + wave * 1024), 16, 0, 0);
        }
    };

    issue_tile(t_beg, 0);

    int b_cur = -1;
    float rstd = 1.f, aw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) aw[s] = 0.f;
    stat_t S1 = 0, S2 = 0;                                          // fixed-point statistics of sample b_cur, this lane

#pragma unroll 1
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        int b, y0, x0;
        decode(t, b, y0, x0);
        const bool full = even || ((y0 + th <= p.H) && (x0 + tw <= p.W) && nslots == 256);
        // this tile's halo / guide pieces were issued one tile ago and have landed: this wave consumed a younger load (res 3)
        if (t == t_beg || (p.usplit & 1)) { HC_WAIT(0); }
        asm volatile("s_barrier" ::: "memory");
        if (p.usplit & 8) { for (int z = 0; z < 64; ++z) __builtin_amdgcn_s_sleep(127); }
        if ((p.usplit & 32) && p.dbg && t == t_end - 1 && wave == 0) {       // debug: LDS content right behind the barrier
            u32x4_t* d = reinterpret_cast<u32x4_t*>(p.dbg) + (long long)lid * 256;
            d[lane] = *reinterpret_cast<const u32x4_t*>(halo + buf * AkWs::HALO + lane * 16);
            d[128 + lane] = *reinterpret_cast<const u32x4_t*>(attb + buf * AkWs::ATT + lane * 16);
            if (lane == 0) { d[255][0] = t; d[255][1] = buf; d[255][2] = t_beg; d[255][3] = (unsigned)__builtin_amdgcn_s_memtime(); }
        }
        if (b != b_cur) {                                           // range enters a new sample: its fold table, rstd, attw
            if (b_cur >= 0 && p.stats_out) {
                const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
                if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
            }
            S1 = 0; S2 = 0;
            b_cur = b;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pc = i * 8 + wave;
                if (pc < 18)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.Tc + (long long)b * 9 * 512 + pc * 256 + lane * 4),
                                                     (LDS_AS void*)(reinterpret_cast<unsigned char*>(tcs) + pc * 1024), 16, 0, 0);
            }
            rstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ms[2 * b + 1])));
#pragma unroll
            for (int s = 0; s < 8; ++s) aw[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.attw[b * 8 + s])));
            HC_WAIT(0);
            asm volatile("s_barrier" ::: "memory");
        }
        const int hpb = buf * AkWs::HALO_PX;                        // halo pixel index of this tile's buffer origin
        const float* attl = reinterpret_cast<const float*>(attb + buf * AkWs::ATT);
        const bool interior = y0 > 0 && x0 > 0 && y0 + th < p.H && x0 + tw < p.W;
        const bf16_t* resb = p.res + (long long)b * p.res_bstride + wave * 8;
        bf16_t* outb = p.out + (long long)b * p.out_bstride + wave * 8;

        // store item pp of this lane: pixel 64 pp + lane, features 8 g .. 8 g + 7.  Its residual (16 bytes) is requested by inline
        // asm (hipcc neither unpacks it right behind the load nor drains the DMA queue with a vmcnt(0) of its own), in this order:
        //     res 0, res 1, [DMA pieces of tile t + 1], res 2, res 3
        // LOADS retire in order (LDS-DMA included); STORES may retire early, so every count below is the number of younger LOADS
        // only (vmcnt also counts the pair stores still in flight: the waits are conservative, never short).  Pairs 0 / 1 do
        // not wait for the 50 KB behind them; by pair 2 (~3 k cycles in) the DMA has landed, and having consumed res 3 a wave
        // knows its own DMA pieces are in LDS: the barrier at the top of the next tile needs no vmcnt in front of it.
        int off2[4];
        u32x4_t rv[4];
        auto request_res = [&](int pp) {
            const int px2 = pp * 64 + lane;
            const int pc = px2 < nslots ? px2 : nslots - 1;
            const int r = fdiv_small(pc, inv_tw), c = pc - r * tw;
            const int y = y0 + r, x = x0 + c;
            const bool ok = full || (px2 < nslots && y < p.H && x < p.W);
            const int yc = y < p.H ? y : p.H - 1, xc = x < p.W ? x : p.W - 1;       // masked lanes read a valid address and store nothing
            const int o = ((yc + 1) * p.Wp + (xc + 1)) * 64;
            off2[pp] = ok ? o : -1;
            rv[pp] = asm_load16(resb + o);
        };
        request_res(0); request_res(1);
        int ndma = 0;                                                // this wave's DMA instructions for tile t + 1
        if (t + 1 < t_end && !(p.usplit & 4)) {
            issue_tile(t + 1, buf ^ 1);
#pragma unroll
            for (int i = 0; i < 6; ++i) ndma += ((i * 8 + wave) * 8 < hcount) ? 1 : 0;
            ndma += 1;
        }
        request_res(2); request_res(3);

        float s1 = 0.f, s2 = 0.f;
        auto do_pair = [&](const int pp, u32x4_t& rvp, const int offp, const int nyounger) {
            // ---- accumulators start at the fold constants of the pixel's border class --------------------------------
            f32x16_t acc[2][2];
            int hpq[2];
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                int slot = pp * 64 + tp * 32 + l31;
                slot = slot < nslots ? slot : nslots - 1;
                const int r = fdiv_small(slot, inv_tw), c = slot - r * tw;
                hpq[tp] = hpb + r * hw + c;
                int cls = 4;
                if (!interior) {
                    int y = y0 + r, x = x0 + c;
                    y = y < p.H ? y : p.H - 1; x = x < p.W ? x : p.W - 1;
                    cls = (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1));
                }
                const float* tc = tcs + cls * 512 + tc_lane;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(tc + 32 * tm + 4 * g4);   // typed vector load: no vmcnt(0) against the DMA in flight
                        acc[tm][tp][4 * g4 + 0] = c4[0]; acc[tm][tp][4 * g4 + 1] = c4[1]; acc[tm][tp][4 * g4 + 2] = c4[2]; acc[tm][tp][4 * g4 + 3] = c4[3];
                    }
            }
            // ---- K loop: 5 steps of two taps x 8 channels; B fragments from the halo, A fragments from registers ------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                bf16x8_t bfr[2];
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const int hp = hpq[tp] + shj[j];
                    bfr[tp] = *reinterpret_cast<const bf16x8_t*>(halo + ((hp << 7) | ((wave ^ ((hp >> 1) & 7)) << 4)));
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp)
                        acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][j], bfr[tp], acc[tm][tp], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- modulation sum in registers: vq[tm][q][tp] = feature 8 g + 4 tm + 2 hh + q of pixel tp ----------------
            float vq[2][2][2];
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int px = pp * 64 + tp * 32 + l31;
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(attl + px * 8), a1 = *reinterpret_cast<const f32x4_t*>(attl + px * 8 + 4);
                const float att[8] = {a0[0] * aw[0], a0[1] * aw[1], a0[2] * aw[2], a0[3] * aw[3], a1[0] * aw[4], a1[1] * aw[5], a1[2] * aw[6], a1[3] * aw[7]};
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float sa = 0.f;
#pragma unroll
                        for (int s = 0; s < 8; ++s) sa += att[s] * acc[tm][tp][8 * q + s];
                        vq[tm][q][tp] = rstd * sa;
                    }
            }
            // ---- half-wave exchange: lane L ends up with all eight features of pixel 64 pp + L --------------------------
            float o8[8];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float lo = vq[tm][q][0], hi = vq[tm][q][1];
                    permlane32_swap(lo, hi);
                    o8[4 * tm + q] = lo;
                    o8[4 * tm + 2 + q] = hi;
                }
            // ---- swish + residual + statistics + 16-byte store -----------------------------------------------------------
            ws_wait_res((p.usplit & 2) ? 0 : nyounger, rvp);                             // younger loads of this wave (see the request order above)
            float vv[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vv[2 * i] = silu_fast(o8[2 * i]) + __builtin_bit_cast(float, rvp[i] << 16);
                vv[2 * i + 1] = silu_fast(o8[2 * i + 1]) + __builtin_bit_cast(float, rvp[i] & 0xffff0000u);
            }
            if (full || offp >= 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { s1 += vv[i]; s2 += vv[i] * vv[i]; }
                *reinterpret_cast<uint4*>(outb + offp) = pack8_bf16(vv);
            }
            __builtin_amdgcn_sched_barrier(0);                       // pairs are not interleaved by the compiler (register pressure)
        };
        do_pair(0, rv[0], off2[0], ndma + 3);
        do_pair(1, rv[1], off2[1], ndma + 2);
        do_pair(2, rv[2], off2[2], 1);
        do_pair(3, rv[3], off2[3], 0);
        if ((p.usplit & 32) && p.dbg && t == t_end - 1 && wave == 0) {       // debug: the same LDS bytes at the end of the tile
            u32x4_t* d = reinterpret_cast<u32x4_t*>(p.dbg) + (long long)lid * 256;
            d[64 + lane] = *reinterpret_cast<const u32x4_t*>(halo + buf * AkWs::HALO + lane * 16);
            d[192 + lane] = *reinterpret_cast<const u32x4_t*>(attb + buf * AkWs::ATT + lane * 16);
        }
        if ((p.usplit & 4) && t + 1 < t_end) { asm volatile("s_barrier" ::: "memory"); issue_tile(t + 1, buf ^ 1); HC_WAIT(0); }
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
    }
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
