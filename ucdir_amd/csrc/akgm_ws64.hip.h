// AKGM tail at 64 channels per group (C = 512, the 36^2 / 18^2 levels) as a PERSISTENT, weight-stationary kernel (gfx950).
// Reference: model/ucdir.py:116,129-140.  The frame of akgm_ws32.hip.h (its best AKGM kernel) for the width that lacked it:
//   * a workgroup (8 wave64, one per CU) owns ONE HALF GROUP for the whole launch: 32 output features x 8 kernel sets = 256
//     weight rows x K = 9 taps x 64 channels.  Wave w keeps the rows of features 4 w .. 4 w + 3 - 32 rows x K = 576 = 36 MFMA A
//     fragments = 144 registers (pack_akgm_ws64), loaded once; the K loop reads only B fragments from LDS: no weight ring, no
//     weight traffic and no prologue per unit (the streaming kernels were prologue / epilogue bound at 18 sub-steps per unit);
//   * 16 roles (half groups) x gridDim / 16 tile ranges; workgroups of one group sit on one XCD (XCD-contiguous logical ids, role =
//     lid / ranges), so the two halves of a group share the group's halo lines in that L2;
//   * LINEAR tiles: 36 and 18 are no multiples of a 4 x 8 pixel tile, so a tile is NPX = 64 | 128 consecutive POSITIONS of the
//     sample's zero-bordered [H + 2][W + 2] plane, from (1, 1) to (H, W) - crossing rows; border columns are computed and not
//     stored (95 % useful positions at 36^2, 90 % at 18^2).  Tap (ky, kx) of position s is halo position s + ky (W + 2) + kx: a
//     uniform shift.  Halo in LDS: [NPX + 2 (W + 2) + 2 positions][128 B = the group's 64 channels], 16-byte chunk XOR
//     (position >> 1) & 7 on the DMA's source side: the 16 lanes of a ds_read_b128 group read 16 different slots at every shift;
//     nine per-lane tap addresses, the k step's channel quarter is an XOR of bits 5-6, the pixel tile an immediate;
//   * halo, guide weights G and the residual's 64-byte segments of tile t + 1 arrive by LDS-DMA during tile t (second buffers):
//     ONE barrier per tile; results go into the residual's staging slot in place and leave as 16-byte pieces of whole 64-byte
//     segments behind the next tile's barrier (akgm_ws32.hip.h);
//   * fold table Tc[9][256] of the current sample in LDS, accumulators start at it; statistics in 2^-20 fixed point per lane
//     (partition-independent: a sample restored alone or in a batch gets bit-identical sums).
#pragma once
#include "akgm_ws32.hip.h"

template <int NW> struct AkWs64 {
    static constexpr int HPOS = 272;                              // halo positions per buffer: NPX + 2 (W + 2) + 2 <= 272
    static constexpr int HALO = HPOS * 128;                       // 34,816 = 17 x 2048: the swizzle survives the buffer switch
    static constexpr int HBYTES(int hpos) { return ((hpos + 15) / 16) * 2048; }
    static constexpr int PSTEP = 32 * 128;                        // one pixel tile = 32 positions further
    static constexpr int SEGB = 8 * NW;                           // bytes of a position this workgroup produces (4 NW features)
    static constexpr int CPS = SEGB / 16;                         // 16-byte chunks per segment
    static constexpr int STAGE = 128 * SEGB;                      // [128 positions][SEGB] residual in / result out
    static constexpr int ATT = 128 * 32;                          // [128 positions][8] fp32
    static constexpr int TCS = 9 * 32 * NW * 4;                   // [9][32 NW] fp32
    static constexpr int NK = 36;                                 // k steps: tap j / 4, channels 16 (j & 3) .. + 15
    // dynamic LDS for a halo of hpos positions: 2 halo buffers | 2 staging slots | 2 guide buffers | fold table
    static constexpr int lds(int hpos) { return 2 * HBYTES(hpos) + 2 * STAGE + 2 * ATT + TCS; }
};

// one LDS-DMA piece (1 KB: 64 lanes x 16 B) from a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset to a wave-uniform
// LDS address: no 64-bit VALU address arithmetic per piece, and hipcc does not count it (cdna guide 5.7): the kernel waits for its DMAs at the
// tile top by hand, and no compiler-made LDS access is held back by a vmcnt(0) for a DMA the compiler knows nothing about
__device__ __forceinline__ void w64_dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// NPT = pixel tiles per tile (2 | 4: 64 | 128 positions); NW = waves per workgroup (8: half a group per workgroup, one workgroup per CU;
// 4: a QUARTER group - 16 features x 8 sets = 128 rows - per workgroup, 80 KB of LDS: TWO independent workgroups per CU, so the two waves
// of a SIMD belong to different workgroups and no barrier aligns them).  p.tiles_x = tiles per sample, p.tiles_y unused (1), p.tw = halo bytes
template <int NPT, int NW, bool ASYM = false>
__global__ __launch_bounds__(64 * NW, 2) void akgm_ws64_kernel(const AkgmHP p) {
    using L = AkWs64<NW>;
    constexpr int NK = L::NK;
    constexpr int CPX = 512;                                       // channels per position
    constexpr int NPX = 32 * NPT;
    constexpr int FEAT = 4 * NW;                                   // output features of the workgroup
    constexpr int PP = 1024 / L::SEGB;                             // positions per residual / output piece of 1 KB
    // ASYM (NW = 8): ALL the memory chores of a tile - the LDS-DMAs of the next tile, the stores of the previous one - are done by waves 0 - 3.
    // The matrix pipe of a SIMD goes to its OLDER wave whenever both have an MFMA ready (tools/micro/mfma_arb.hip): waves 0 - 3 run their K
    // loops unimpeded and then wait at the tile barrier, waves 4 - 7 only get the pipe while their partner is outside its K loops.  With the
    // chores on the older half, the younger half starts its first K loop right behind the barrier, under the partner's chores.
    constexpr int NDW = ASYM ? 4 : NW;                             // waves that do the chores
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
#ifndef W64_DBGWAVE
#define W64_DBGWAVE (NW - 3)
#endif
#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == W64_DBGWAVE);
    int dbg_n = 0;
#define W64_STAMP() do { if (dbg_on && dbg_n < 250) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W64_STAMP() do {} while (0)
#endif
    W64_STAMP();
    constexpr int NROLE = 512 / FEAT;                              // 16 half groups | 32 quarter groups
    const int nslots = (int)gridDim.x / NROLE;                     // tile ranges per role (the grid is a multiple of NROLE)
    const int role = lid / nslots, slot = lid - role * nslots;     // roles of one group sit on one XCD (XCD-contiguous lid): its halo lines stay in that L2
    const int Wp = p.Wp, Ptot = (p.H + 2) * Wp, Plast = p.H * Wp + p.W;     // positions of a sample; last valid position
    const float inv_wp = 1.0f / (float)Wp;
    const int tps = p.tiles_x, T = p.nbatch * tps;
    const int t_beg = (int)((long long)slot * T / nslots), t_end = (int)((long long)(slot + 1) * T / nslots);
    if (t_beg >= t_end) return;
    const int chan0 = FEAT * role;                                 // this workgroup's first output channel
    const int gch0 = 64 * (chan0 >> 6);                            // its group's first input channel
    const int HB = p.tw;                                           // bytes per halo buffer (a multiple of 2048)
    const unsigned OFF_STAGE = 2 * HB, OFF_ATT = OFF_STAGE + 2 * L::STAGE, OFF_TCS = OFF_ATT + 2 * L::ATT;

    // ---- this wave's weights, resident ------------------------------------------------------------------------------------
    bf16x8_t af[NK];
    {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + ((long long)(role * NW + wave) * NK) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < NK; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) asm volatile("" : "+v"(af[j]));
    W64_STAMP();                                                   // weights resident

    // ---- tile-invariant lane constants ---------------------------------------------------------------------------------
    // halo piece k (8 positions x 128 B): wave w stages pieces w, w + NW, ...; lane -> (halo position 8 k + lane / 8, physical chunk lane & 7);
    // the logical chunk is physical ^ (position >> 1) & 7 = physical ^ (4 (k & 1) + lane / 16)
    const int npiece = (NPX + 2 * Wp + 2 + 7) >> 3;
    constexpr int NHOP = (L::HPOS / 8 + NDW - 1) / NDW;            // halo DMA operations per (chore) wave at most (5 | 9)
    constexpr int NROP = NW / NDW;                                 // residual pieces per chore wave
    // B fragment of tap t, channel quarter 0, pixel tile 0, buffer 0: LDS byte address (quarter cq: ^ (cq << 5); pixel tile q: + q PSTEP)
    unsigned bt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int hp = l31 + (t / 3) * Wp + t % 3;
        bt[t] = hp * 128 + ((hh ^ ((hp >> 1) & 7)) << 4) + HB;                     // (flipped to buffer 0 at the top of the first tile)
    }
    const unsigned tc_lane = OFF_TCS + 4 * 8 * (4 * wave + 2 * hh);                      // + 128 NW cls: this lane's 16 table entries (2 features x 8 sets)
    const unsigned att_lane = OFF_ATT + l31 * 32;                                        // + 1024 q: this lane's position of pixel tile q
    // this lane's 4 bytes (features 4 w + 2 hh, + 1) of position 32 q + l31 in the staging slot: chunk w / 2 ^ (pos >> 2) & (CPS - 1), + 8 (w & 1) + 4 hh
    const unsigned st_lane = OFF_STAGE + l31 * L::SEGB + (((wave >> 1) ^ ((l31 >> 2) & (L::CPS - 1))) << 4) + 8 * (wave & 1) + 4 * hh;   // + 32 SEGB q

    int b = t_beg / tps, ti = t_beg - b * tps;                      // tile t = (sample b, tile ti of the sample)
    // DMA operation `op` of tile (nb, nti) into buffer buf: 0 .. NHOP - 1 = this wave's halo pieces NDW op + w, NHOP = its guide piece (32
    // positions x 32 B, waves < NPT), NHOP + 1 + rk = its residual pieces NDW rk + w (PP positions x SEGB, pieces < NPX / PP)
    auto dma_op = [&](auto opc, int nb, int nti, int buf) {
        constexpr int op = decltype(opc)::value;
        // the per-lane offsets are formed HERE (three or four VALU per piece), from a lane id that is opaque at this point: hoisted to
        // the tile top they were seven more registers live across the K loops (spills)
        unsigned ln = lane;
        asm volatile("" : "+v"(ln));
        if constexpr (op < NHOP) {
            const int k = op * NDW + wave;
            if (k < npiece) {
                int sp = nti * NPX + 8 * k + (int)(ln >> 3);
                sp = sp < Ptot ? sp : Ptot - 1;                     // (the last tile's halo may run past the sample: clamped, feeds dropped positions only)
                const unsigned off = (unsigned)sp * (CPX * 2) + (unsigned)(gch0 + (((ln & 7) ^ (4 * (k & 1) + (ln >> 4))) << 3)) * 2;
                w64_dma16(p.h + (long long)nb * p.h_bstride, off, (unsigned)__builtin_amdgcn_readfirstlane(buf * HB + k * 1024));
            }
        } else if constexpr (op == NHOP) {
            if (wave < NPT) {
                int P = Wp + 1 + nti * NPX + 32 * wave + (int)(ln >> 1); P = P < Ptot ? P : Ptot - 1;
                int y = fdiv_small(P, inv_wp), x = P - y * Wp;
                y = y < 1 ? 1 : (y > p.H ? p.H : y); x = x < 1 ? 1 : (x > p.W ? p.W : x);
                w64_dma16(p.G + (long long)nb * p.g_bstride, (unsigned)((((y - 1) * p.W + (x - 1)) * 8 + (int)(ln & 1) * 4) * 4),
                          (unsigned)__builtin_amdgcn_readfirstlane(OFF_ATT + buf * L::ATT + wave * 1024));
            }
        } else {
            constexpr int rk = op - NHOP - 1;                       // this wave's residual piece rk: piece NDW rk + w
            const int piece = NDW * rk + wave;
            if (piece < NPX / PP) {
                const unsigned pos = PP * piece + ln / L::CPS;
                int P = Wp + 1 + nti * NPX + (int)pos; P = P < Ptot ? P : Ptot - 1;
                const unsigned off = (unsigned)P * (CPX * 2) + (unsigned)(chan0 + (((ln & (L::CPS - 1)) ^ ((pos >> 2) & (L::CPS - 1))) << 3)) * 2;
                w64_dma16(p.res + (long long)nb * p.res_bstride, off, (unsigned)__builtin_amdgcn_readfirstlane(OFF_STAGE + buf * L::STAGE + piece * 1024));
            }
        }
    };
    constexpr int NOPS = NHOP + 1 + NROP;
    if (wave < NDW) static_for<0, NOPS>([&](auto opc) { dma_op(opc, b, ti, 0); });

    int b_cur = -1;
    float rstd = 1.f, aw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) aw[s] = 0.f;
    stat_t S1 = 0, S2 = 0;
    bool have_prev = false;                                          // results of the previous tile wait in its staging slot
    long long prev_out[NROP];
#pragma unroll
    for (int i = 0; i < NROP; ++i) prev_out[i] = -1;

#pragma unroll 1
    for (int t = t_beg; t <= t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        // every LDS-DMA of this tile (issued during the previous tile) has landed; every wave is done with the previous tile (its
        // results are in the other staging slot, its halo buffer is free)
        W64_STAMP();                                                // tile top
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        W64_STAMP();                                                // behind the barrier
        if (have_prev && wave < NDW) {                              // the previous tile's finished segments out
#pragma unroll
            for (int i = 0; i < NROP; ++i) {
#ifndef W64_ABL_NOSTORE
                if (prev_out[i] >= 0)
#else
                if (prev_out[i] == -12345)
#endif
                {
                    const u32x4_t ln = *reinterpret_cast<const u32x4_t*>(smem + OFF_STAGE + (buf ^ 1) * L::STAGE + (i * 64 * NDW + tid) * 16);
                    *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(p.out) + prev_out[i]) = ln;
                }
            }
        }
        if (t == t_end) break;
        if (b != b_cur) {                                           // range enters a new sample: its fold table, rstd, attw
            if (b_cur >= 0 && p.stats_out) {
                const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
                if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
            }
            S1 = 0; S2 = 0;
            b_cur = b;
            // fold table slice [9][32 NW] of this sample: piece pc = classes (NW == 8: pc, 64 lanes x 16 B; NW == 4: 2 pc and 2 pc + 1, 32 lanes each)
            constexpr int NTP = (NW == 8) ? 9 : 5;
            float mean_b = 0.f;
            if (p.own_tc) stat_mean_rstd_wave(p.stats, b, p.inv_count, lane, mean_b, rstd);
            else rstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ms[2 * b + 1])));
            const float inv_b = 1.0f / rstd;
#pragma unroll
            for (int i = 0; i < (NTP + NW - 1) / NW; ++i) {
                const int pc = i * NW + wave;
                if (pc < NTP) {
                    const int cl = (NW == 8) ? pc : 2 * pc + (lane >> 5), e4 = (NW == 8) ? lane : (lane & 31);
                    if (cl < 9) {
                        const long long rel = (long long)cl * (8 * CPX) + 8 * chan0 + e4 * 4;
                        if (p.own_tc) akgm_tc_piece(p, rel, smem + OFF_TCS + pc * 1024 + lane * 16, inv_b, mean_b);
                        else __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.Tc + (long long)b * 9 * (8 * CPX) + rel),
                                                              (LDS_AS void*)(smem + OFF_TCS + pc * 1024), 16, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) aw[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.attw[b * 8 + s])));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const bool last = t + 1 == t_end;
        int nb = b, nti = ti + 1;
        if (nti == tps) { nti = 0; ++nb; }
        // (the staging slot buf ^ 1 was just read by the store above: the DMA of the next tile's residual goes behind it in program
        // order, the data is in registers by then)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W64_STAMP();                                                // stores issued
#ifndef W64_ABL_NODMA
        if (!last && wave < NDW) static_for<0, NOPS>([&](auto opc) { dma_op(opc, nb, nti, buf ^ 1); });
#endif

        const int P0 = Wp + 1 + ti * NPX;
        const unsigned ab0 = buf * L::ATT, sb0 = buf * L::STAGE;
#pragma unroll
        for (int k = 0; k < 9; ++k) bt[k] = buf ? bt[k] + HB : bt[k] - HB;       // in place: 9 registers, not 18
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
        for (int qp = 0; qp < NPT; qp += 2) {                       // two pixel tiles at a time
            // accumulators start at the fold constants of each position's border class.  (Rotating this block behind the previous pair's
            // epilogue - so that the younger wave of a SIMD enters its first K loop sooner behind the tile barrier - was measured: with the
            // accumulators live across the tile top hipcc spills all 32 of them; with only the class addresses carried 92.4 -> 97.6 us.)
            f32x16_t acc[2];
            bool valid[2];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const int P = P0 + 32 * (qp + tq) + l31;
                const int y = fdiv_small(P, inv_wp), x = P - y * Wp;
                valid[tq] = x >= 1 && x <= p.W && P <= Plast;
                const int cy = (y <= 1) ? 0 : ((y >= p.H) ? 2 : 1);
                const int cx = (x <= 1) ? 0 : ((x >= p.W) ? 2 : 1);
                const unsigned tca = tc_lane + (cy * 3 + cx) * (128 * NW);
#ifdef W64_ABL_NOINIT
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[tq][e] = (float)tca;
#else
                acc[tq] = lds_read_f32x16(smem + tca);            // (concatenated reads: no v_mov, asmops.hip.h)
#endif
            }
            // K loop software-pipelined by hand (inline-asm fragment reads, counted lgkmcnt; see akgm_ws.hip.h): the fragments of
            // steps j + 1 and j + 2 are in flight under the two MFMAs of step j (64 matrix-core cycles < one LDS round trip)
            bf16x8_t bfr[3][2];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fold constants; nothing of the compiler's own is queued behind this
            W64_STAMP();                                            // accumulators initialised
            auto frag = [&](auto jc, bf16x8_t (&dst)[2]) {
                constexpr int j = decltype(jc)::value;
                const unsigned a0 = bt[j >> 2] ^ ((j & 3) << 5);
                lds_read16_asm<0>(dst[0], a0);
                lds_read16_asm<L::PSTEP>(dst[1], a0);
            };
            frag(std::integral_constant<int, 0>{}, bfr[0]);
            frag(std::integral_constant<int, 1>{}, bfr[1]);
            __builtin_amdgcn_s_setprio(1);
            static_for<0, NK>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + 2 < NK) frag(std::integral_constant<int, j + 2>{}, bfr[(j + 2) % 3]);
                constexpr int younger = (j + 2 < NK) ? 4 : ((j + 1 < NK) ? 2 : 0);
                lgkm_wait_asm<younger>();
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) acc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bfr[j % 3][tq], acc[tq], 0, 0, 0);
            });
            __builtin_amdgcn_s_setprio(0);
            W64_STAMP();                                            // K loop done
            // ---- modulation sum, swish, residual, statistics; the lane's two features go back into the staging slot -------------
#ifdef W64_ABL_NOEPI
            s1 += acc[0][0] + acc[1][5];
#else
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const unsigned aq = att_lane + ab0 + (qp + tq) * 1024;
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(smem + aq), a1 = *reinterpret_cast<const f32x4_t*>(smem + aq + 16);
                const float att[8] = {a0[0] * aw[0], a0[1] * aw[1], a0[2] * aw[2], a0[3] * aw[3], a1[0] * aw[4], a1[1] * aw[5], a1[2] * aw[6], a1[3] * aw[7]};
                float o2[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float sa = 0.f;
#pragma unroll
                    for (int s = 0; s < 8; ++s) sa += att[s] * acc[tq][8 * k + s];
                    o2[k] = rstd * sa;
                }
                const unsigned sa = st_lane + sb0 + (qp + tq) * (32 * L::SEGB);
                const unsigned rv = *reinterpret_cast<const unsigned*>(smem + sa);
                float v0 = silu_fast(o2[0]) + __builtin_bit_cast(float, rv << 16);
                float v1 = silu_fast(o2[1]) + __builtin_bit_cast(float, rv & 0xffff0000u);
                *reinterpret_cast<unsigned*>(smem + sa) = pack2_bf16(v0, v1);
                v0 = valid[tq] ? v0 : 0.f; v1 = valid[tq] ? v1 : 0.f;        // border columns / positions past the sample: dropped
                s1 += v0 + v1; s2 += v0 * v0 + v1 * v1;
            }
#endif
#pragma unroll
            for (int k = 0; k < 9; ++k) bt[k] += 2 * L::PSTEP;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) bt[k] -= NPT * L::PSTEP;
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
        have_prev = true;
        if (wave < NDW) {
#pragma unroll
            for (int i = 0; i < NROP; ++i) {                        // line mover items of this thread: the 16 bytes at (i 64 NDW + tid) 16 of the slot
                const int it = i * 64 * NDW + tid, mpos = it / L::CPS, P = P0 + mpos;
                const int y = fdiv_small(P < Ptot ? P : Ptot - 1, inv_wp), x = P - y * Wp;
                const bool ok = mpos < NPX && x >= 1 && x <= p.W && P <= Plast;
                prev_out[i] = ok ? ((long long)b * p.out_bstride + (long long)P * CPX + chan0 + (((it & (L::CPS - 1)) ^ ((mpos >> 2) & (L::CPS - 1))) << 3)) * 2 : -1;
            }
        }
        b = nb; ti = nti;
    }
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
