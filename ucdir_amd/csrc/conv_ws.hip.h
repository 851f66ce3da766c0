// 3x3 stride-1 convolution 64 -> 64 channels (the 288^2 level: conv1 of downs.1 / downs.2, predictor layers) as a
// PERSISTENT, weight-stationary kernel (gfx950).  Reference: model/ucdir.py:110,122-127 (GroupNorm -> conv1 -> Swish).
//
// conv3x3_halo_kernel<64> runs these layers at 530 TFLOP/s: 5,184 workgroups per B = 16 launch, each staging 74 KB of
// weights through a 2-deep LDS ring for 256 pixels, two barriers per 16 MFMAs, a table-building prologue and a staged
// epilogue per workgroup.  With K = 576 and 64 rows the whole weight matrix fits in REGISTERS:
//   * one workgroup per CU (8 wave64, 256 VGPRs each) walks a contiguous range of 16 x 16 pixel tiles;
//   * wave (rw, pw) owns rows 32 rw .. +31 for the whole launch - 36 A fragments = 144 VGPRs, loaded once - and the
//     64 pixels (two 32-pixel MFMA tiles = tile rows 4 pw .. 4 pw + 3) of every tile;
//   * the K loop is 36 x { two ds_read_b128 (B fragments of the two pixel tiles), two MFMAs }: no weight traffic, no
//     barrier; the halo ([18][24-pixel pitch][64 ch] bf16, chunk XOR (pixel >> 1) & 7, see akgm_ws.hip.h) of tile
//     t + 1 arrives by LDS-DMA under tile t's MFMAs; ONE barrier per tile;
//   * K order: step j = 4 tap + c16, lanes 0-31 channels 16 c16 .. +7, lanes 32-63 the next 8: a B fragment is one
//     16-byte chunk of one halo pixel, its address one of 12 per-lane registers (3 kx x 4 swizzled chunks) + immediate;
//   * rows are permuted at pack time (pack_conv_ws) so that a lane's 16 accumulators of a pixel tile are 16 CONSECUTIVE
//     channels of ONE pixel: the epilogue (GroupNorm fold, activation, statistics) runs in registers and stores
//     2 x 16 bytes per lane and pixel tile - no LDS stage, no barrier;
//   * fold table Tc[9][64] of the current sample in LDS (rebuilt when the range crosses a sample); output statistics as
//     fixed-point per-tile partials in 64-bit integers (partition-independent, see akgm_ws.hip.h).
#pragma once
#include <type_traits>
#include "akgm_ws.hip.h"

#ifndef CW_DEPTH
#define CW_DEPTH 1                                                // B fragment reads issued this many K steps ahead of their MFMAs
#endif

struct CvWs {
    static constexpr int PITCH = AkWs::PITCH, HALO = AkWs::HALO, QSTEP = AkWs::QSTEP;
    static constexpr int OFF_TCS = 2 * HALO;                      // [9][64] fp32
    static constexpr int OFF_SCAL = OFF_TCS + 9 * 64 * 4;
    static constexpr int LDS = OFF_SCAL + 128;                    // 113,024: one workgroup per CU
    static constexpr int A_ELEMS = 2 * 36 * 2 * 32 * 8;           // packed image [rw][j][hh][32 rows][8] bf16
};

__global__ __launch_bounds__(HC_THREADS, 2) void conv_ws_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const tcs = reinterpret_cast<float*>(smem + CvWs::OFF_TCS);
    float* const scal = reinterpret_cast<float*>(smem + CvWs::OFF_SCAL);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave & 1, pw = wave >> 1;
    // lane -> pixel of a 32-pixel MFMA tile (two tile rows): row l31 / 16, column (l31 % 16) ^ 8 in the second row.  With the 24-pixel
    // pitch the 16 lanes of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) then hit 16 different
    // 16-byte slots (halo pixel index distinct mod 16); the plain mapping was 2-way on every read (PMC: conflicts 47 % of LDS cycles)
    const int prow = l31 >> 4, pcol = (l31 & 15) ^ (prow << 3);
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 1 || wave == 5);
    unsigned long long* const dbgp = p.dbg + (wave == 5 ? 256 : 0);
    int dbg_n = 0;
#define CW_STAMP() do { if (dbg_on && dbg_n < 250) dbgp[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CW_STAMP() do {} while (0)
#endif
    CW_STAMP();
    const int tps = p.tiles_x * p.tiles_y;
    const int T = p.nbatch * tps;
    const int t_beg = (int)((long long)lid * T / (int)gridDim.x), t_end = (int)((long long)(lid + 1) * T / (int)gridDim.x);
    if (t_beg >= t_end) return;

    // ---- this wave's weights: 36 A fragments, resident for the whole launch ----------------------------------------------
    bf16x8_t af[36];
    {
        const bf16_t* Ab = p.A + ((rw * 36 * 2 + hh) * 32 + l31) * 8;
#pragma unroll
        for (int j = 0; j < 36; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * (2 * 32 * 8));
    }
#pragma unroll
    for (int j = 0; j < 36; ++j) asm volatile("" : "+v"(af[j]));    // the wait for these loads goes here, not into the tile loop

    // ---- tile-invariant lane constants (halo staging as in akgm_ws.hip.h) ---------------------------------------------------
    // Halo staging: piece (r, c3) = row r, pixel columns 8 c3 .. 8 c3 + 7 (columns >= 18 do not exist) goes to LDS bytes
    // (3 r + c3) * 1024.  Wave w stages rows w and w + 8 (the per-lane source offset of row r + 8 is that of row r plus a
    // wave-uniform 8 rows: the swizzle (hp >> 1) & 7 is the same), waves 2-7 one piece of rows 16 / 17 each: 4 offset registers.
    int hrel[3], hrel_x = -1;
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
        const int col = c3 * 8 + (lane >> 3), hp = wave * CvWs::PITCH + col;
        hrel[c3] = col < 18 ? (wave * p.Wp + col) * 64 + (((lane & 7) ^ ((hp >> 1) & 7)) << 3) : -1;
    }
    const int xr = 16 + (wave - 2) / 3, xc3 = (wave - 2) % 3;      // waves 2-7: the extra piece
    if (wave >= 2) {
        const int col = xc3 * 8 + (lane >> 3), hp = xr * CvWs::PITCH + col;
        hrel_x = col < 18 ? (xr * p.Wp + col) * 64 + (((lane & 7) ^ ((hp >> 1) & 7)) << 3) : -1;
    }
    // B fragment of tap (ky, kx), chunk pair c16, pixel tile 2 pw (+ QSTEP: 2 pw + 1): halo pixel hp = hp0 + 24 ky + kx,
    // 16-byte chunk (2 c16 + hh) ^ ((hp >> 1) & 7).  (hp + 24) >> 1 adds 12: the swizzle flips bit 2 for ky = 1 and is
    // unchanged for ky = 2, so with K = 2 c16 ^ (ky == 1 ? 4 : 0) in {0, 2, 4, 6} the address is  ba[kx][K / 2] + 3072 ky:
    // twelve registers, no address arithmetic in the K loop (the kernel is bound by instruction ISSUE: ~4.5 cycles per instruction and SIMD).
    unsigned ba[3][4];                             // [kx][K / 2]: LDS byte address of the fragment (buffer 0; flipped in place every tile)
    {
        const int hp0 = (4 * pw + prow) * CvWs::PITCH + pcol;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sxh = (((hp0 + kx) >> 1) & 7) ^ hh;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) ba[kx][k2] = ((hp0 + kx) << 7) + (((2 * k2) ^ sxh) << 4);
        }
    }
    const unsigned tc_lane = CvWs::OFF_TCS + (32 * rw + 16 * hh) * 4;              // + 256 cls: this lane's 16 channels of the fold table
    const unsigned rel_out = (unsigned)(((4 * pw + prow + 1) * p.Wp + pcol + 1) * 64 + 32 * rw + 16 * hh) * 2;   // pixel tile 2 pw; + 2 Wp rows: 2 pw + 1

    int b, ty, tx;
    {
        b = t_beg / tps;
        const int r = t_beg - b * tps;
        ty = r / p.tiles_x; tx = r - ty * p.tiles_x;
    }
    auto issue_tile = [&](int nb, int nty, int ntx, int buf) {
        const bf16_t* hb = p.B0 + (long long)nb * p.b0_bstride + (long long)(nty * 16 * p.Wp + ntx * 16) * 64;
        unsigned char* hd = smem + buf * CvWs::HALO;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
            if (hrel[c3] >= 0) {
                stage16(hb + hrel[c3], hd + (3 * wave + c3) * 1024, lane);
                stage16(hb + (long long)8 * p.Wp * 64 + hrel[c3], hd + (3 * (wave + 8) + c3) * 1024, lane);
            }
        if (wave >= 2) { if (hrel_x >= 0) stage16(hb + hrel_x, hd + (3 * xr + xc3) * 1024, lane); }
    };
    issue_tile(b, ty, tx, 0);

    int b_cur = -1;
    float rstd_a = 1.f;                                             // alpha * rstd of the current sample
    stat_t S1 = 0, S2 = 0;
    const int act = p.act;

    // The two waves of a SIMD (w and w + 4) run STAGGERED by half a tile: waves 4-7 ("late") keep the accumulators of tile t
    // across the barrier and run its epilogue at the top of tile t + 1, i.e. while waves 0-3 are in their K loop - and are in
    // their own K loop while waves 0-3 run their epilogue.  In lockstep (one barrier per tile aligns all eight waves) both
    // waves of a SIMD wanted the matrix pipe at the same time and the VALU at the same time: MFMA busy 39 %.
    f32x16_t acc[2];
    bool pend = false;                                              // late waves: epilogue of tile (pb, pty, ptx) not yet run
    int pb = 0, pty = 0, ptx = 0;
    // ---- epilogue in registers: lane = pixel (prow, pcol) of the pixel tile, channels 32 rw + 16 hh .. + 15 -----------------
    // wait_dma: the early waves pass their vmcnt(0) between the arithmetic and the stores (see the tile loop)
    auto epilogue = [&](int eb, int ety, int etx, const bool wait_dma) {
        const bool interior = ety > 0 && etx > 0 && ety + 1 < p.tiles_y && etx + 1 < p.tiles_x;
        unsigned char* outb = reinterpret_cast<unsigned char*>(reinterpret_cast<bf16_t*>(p.out) + (long long)eb * p.out_bstride + (long long)(ety * 16 * p.Wp + etx * 16) * 64);
        float s1 = 0.f, s2 = 0.f;
        uint4 pk[2][2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            unsigned tca = tc_lane + 4 * 256;                       // class 4
            if (!interior) {
                const int r = 4 * pw + 2 * tp + prow, c = pcol;
                const int cy = (ety == 0 && r == 0) ? 0 : ((ety + 1 == p.tiles_y && r == 15) ? 2 : 1);
                const int cx = (etx == 0 && c == 0) ? 0 : ((etx + 1 == p.tiles_x && c == 15) ? 2 : 1);
                tca = tc_lane + (cy * 3 + cx) * 256;
            }
            // (scalar fp32 on purpose: the same arithmetic on v_pk_fma / v_pk_mul / v_pk_add_f32 - half the issue slots - ran the
            // epilogue at 4.8 k instead of 3.3 k cycles beside the partner wave's MFMAs: 105 -> 130 us per launch)
            float v[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(smem + tca + 16 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g4 + e] = fmaf(acc[tp][4 * g4 + e], rstd_a, c4[e]);
            }
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = silu_fast(v[i]);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaxf(0.2f * v[i], v[i]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
            pk[tp][0] = pack8_bf16(v); pk[tp][1] = pack8_bf16(v + 8);
        }
        if (wait_dma) { HC_WAIT(0); }
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            unsigned char* op = outb + rel_out + (long long)tp * (2 * p.Wp * 64 * 2);
            *reinterpret_cast<uint4*>(op) = pk[tp][0];
            *reinterpret_cast<uint4*>(op + 16) = pk[tp][1];
        }
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
    };

    // two specialised copies of the tile loop (a wave-uniform flag in ONE loop made hipcc carry both liveness patterns: 48 spills)
    auto tile_loop = [&](auto late_tag) {
    constexpr bool late = decltype(late_tag)::value;
#pragma unroll 1
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        const bool last = t + 1 == t_end;
        // every wave passed vmcnt(0) behind the K loop of the tile before (its DMA pieces of this tile included)
        CW_STAMP();                                                 // tile top
        if (t == t_beg) { HC_WAIT(0); }
        asm volatile("s_barrier" ::: "memory");
        CW_STAMP();                                                 // behind the barrier
        if (late && pend) { epilogue(pb, pty, ptx, false); pend = false; }  // late waves: tile t - 1, under the early waves' K loop (and in front of a table rebuild)
        if (b != b_cur) {                                           // range enters a new sample: statistics -> fold table
            if (b_cur >= 0 && p.stats_out) {
                const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
                if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
            }
            S1 = 0; S2 = 0;
            b_cur = b;
            if (wave == 0) {
                float mean = 0.f, rstd = 1.f;
                if (p.fold) {
                    long long v = 0;
                    if (lane < 2 * UCDIR_STAT_SLOTS) v = p.stats0[(long long)b * (2 * UCDIR_STAT_SLOTS) + lane];   // lane l: slot l / 2, sum | sum of squares
#pragma unroll
                    for (int off = 2; off < 2 * UCDIR_STAT_SLOTS; off <<= 1) v += __shfl_xor(v, off);
                    const long long q = __shfl(v, 1);
                    mean_rstd(stat_val(v), stat_val(q), p.inv_count, mean, rstd);
                }
                if (lane == 0) { scal[0] = mean; scal[1] = rstd; }
            }
            __syncthreads();
            const float mean = scal[0], rstd = scal[1];
            for (int i = tid; i < 9 * 64; i += HC_THREADS) {
                const int cls = i >> 6, f = i & 63;
                float v = p.bias ? p.bias[f] : 0.f;
                if (p.fold) v += p.Tb[(long long)cls * p.tab_ld + f] - mean * rstd * p.Tg[(long long)cls * p.tab_ld + f];
                tcs[i] = v;
            }
            rstd_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.alpha * (p.fold ? rstd : 1.0f))));
            __syncthreads();
        }
        int nb = b, nty = ty, ntx = tx + 1;                          // tile t + 1
        if (ntx == p.tiles_x) { ntx = 0; if (++nty == p.tiles_y) { nty = 0; ++nb; } }
        CW_STAMP();                                                 // (late waves: epilogue of tile t - 1 done)
        if (!last) issue_tile(nb, nty, ntx, buf ^ 1);
        CW_STAMP();                                                 // DMA of tile t + 1 issued

        if (t != t_beg) {                                           // the fragment addresses follow the halo buffer
            const int d = buf ? CvWs::HALO : -CvWs::HALO;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) ba[kx][k2] += d;
        }
        // ---- K loop: 9 taps x 4 chunk pairs, two pixel tiles share every A fragment -------------------------------------
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tp][e] = 0.f;
        // software pipeline by hand (round 5): inline-asm fragment reads with counted lgkmcnt, the fragments of steps j + 1 and j + 2 in flight
        // under the two MFMAs of step j.  (Compiler-counted reads, one or two steps ahead, ran the same: hipcc's LDS bookkeeping waits with
        // lgkmcnt(0) - the read it has just issued included - so the depth never mattered: 109.3 vs 109.5 us in round 3.)
        bf16x8_t bq[3][2];
        auto read_b = [&](auto jc, bf16x8_t (&dst)[2]) {
            constexpr int j = decltype(jc)::value;
            constexpr int tap = j >> 2, c16 = j & 3;
            constexpr int ky = tap / 3, kx = tap - 3 * ky;
            constexpr int k2 = c16 ^ (ky == 1 ? 2 : 0);
            lds_read16_asm<ky * (CvWs::PITCH * 128)>(dst[0], ba[kx][k2]);
            lds_read16_asm<ky * (CvWs::PITCH * 128) + CvWs::QSTEP>(dst[1], ba[kx][k2]);
        };
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // nothing of the compiler's own is queued behind this
        read_b(std::integral_constant<int, 0>{}, bq[0]);
        read_b(std::integral_constant<int, 1>{}, bq[1]);
        __builtin_amdgcn_s_setprio(1);
        static_for<0, 36>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 2 < 36) read_b(std::integral_constant<int, j + 2>{}, bq[(j + 2) % 3]);
            constexpr int younger = (j + 2 < 36) ? 4 : ((j + 1 < 36) ? 2 : 0);
            lgkm_wait_asm<younger>();
            const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // step 0: C = 0 as an inline constant
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bq[j % 3][0], j ? acc[0] : zero, 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bq[j % 3][1], j ? acc[1] : zero, 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        CW_STAMP();                                                 // K loop done
        // This wave's DMA pieces of tile t + 1 must be in LDS before it reaches the next barrier, and an LDS-DMA fill runs at
        // ~12 bytes per cycle and CU: the 54 KB of a tile need ~5 k cycles from issue.  Late waves (DMA issued behind their
        // epilogue at the top, stores long retired) wait here, behind the K loop; early waves wait between the arithmetic of
        // their epilogue and its four stores - waiting in front of it cost them 1.5 k cycles per tile, waiting behind the
        // stores would wait for the stores.
        if (late) { HC_WAIT(0); }
        CW_STAMP();                                                 // (late waves: vmcnt(0) passed)
        if (late) { pend = true; pb = b; pty = ty; ptx = tx; } else epilogue(b, ty, tx, true);
        CW_STAMP();                                                 // (early waves: epilogue done)
        b = nb; ty = nty; tx = ntx;
    }
    if (late && pend) epilogue(pb, pty, ptx, false);
#ifdef UCDIR_TIMING
    if (dbg_on) dbgp[255] = dbg_n;
#endif
    };
    if (wave >= 4) tile_loop(std::true_type{}); else tile_loop(std::false_type{});
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
