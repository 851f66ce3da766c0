// AKGM tail at 32 channels per group (C = 256, the 72^2 level) as a PERSISTENT, weight-stationary kernel (gfx950).
// Reference: model/ucdir.py:129-140.  Same idea as akgm_ws.hip.h, different split:
//   * a workgroup (8 wave64, one per CU) owns ONE GROUP for the whole launch: its 256 weight rows (32 output features x 8
//     modulation samples) x K = 9 taps x 32 channels.  Wave w keeps the rows of features 4 w .. 4 w + 3 - 32 rows x K = 288 =
//     72 registers of MFMA A fragments (pack_akgm_ws32); the K loop reads only B fragments from LDS;
//   * tiles of TH x 8 pixels (TH = 32 | 24 | 16 | 8, the largest that divides H; W a multiple of 8), pixel tiles of 4 rows x 8
//     columns; the group's 64 bytes of every halo pixel in LDS: [TH + 2 rows][12-pixel pitch][64 B], 16-byte chunk XOR
//     f(row, col) = [col & 4] / 4 + 2 [row & 1] on the DMA's source side - the 16 lanes of a ds_read_b128 group (two tile rows x 8
//     columns, any tap shift) read 16 different slots; nine per-lane tap addresses + immediates;
//   * halo, guide weights G and the residual's 64-byte segments of tile t + 1 arrive by LDS-DMA during tile t (second buffers):
//     ONE barrier per tile, no vmcnt drain inside a tile;
//   * a lane ends up with 2 features of one pixel (4 bytes): results go into the residual's staging slot in place
//     ([pixels][64 B], chunk XOR (pixel >> 2) & 3), and BEHIND the next tile's barrier every thread stores 2 x 16 bytes of
//     finished 64-byte segments (16-byte pieces of 64 different lines per store instruction cost akgm_ws 50 of 186 us);
//   * fold table Tc[9][256] of the current sample in LDS, accumulators start at it; statistics as in akgm_ws.hip.h
//     (2^-20 fixed point, partition-independent).
#pragma once
#include "akgm_ws.hip.h"

struct AkWs32 {
    static constexpr int PITCH = 12;                              // halo pixels per LDS row (10 used)
    static constexpr int HROW = PITCH * 64;                       // 768 bytes
    static constexpr int HALO = 34 * HROW;                        // 26,112: TH <= 32
    static constexpr int QSTEP = 4 * HROW;                        // one pixel tile = four tile rows further
    static constexpr int STAGE = 256 * 64;                        // [256 px][64 B] residual in / result out
    static constexpr int ATT = 256 * 32;                          // [256 px][8] fp32
    static constexpr int OFF_STAGE = 2 * HALO;
    static constexpr int OFF_ATT = OFF_STAGE + 2 * STAGE;
    static constexpr int OFF_TCS = OFF_ATT + 2 * ATT;             // [9][256] fp32
    static constexpr int LDS = OFF_TCS + 9 * 1024;                // 110,592
};

// CG = 32: as described above.  CG = 16 | 8 (C = 128 | 64): the workgroup owns a BLOCK of 32 features = 2 | 4 groups (the same 64
// bytes of a pixel), wave w the features 4 w .. 4 w + 3 of the block (group w / 4 | w / 2 of it): K = 9 x 16 = 9 steps of one tap |
// 9 x 8 (+ one zero tap) = 5 steps of two taps; everything else - tiles, staging, epilogue - is shared.
template <int CG>
__global__ __launch_bounds__(HC_THREADS, 2) void akgm_ws32_kernel(const AkgmHP p) {
    constexpr int NK = (CG == 32) ? 18 : ((CG == 16) ? 9 : 5);     // k steps (CG 32: tap j / 2, channels 16 (j & 1) .. + 15)
    constexpr int CPX = 8 * CG;                                    // channels per pixel
    constexpr int NB = CPX / 32;                                   // 32-feature blocks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prow = l31 >> 3, pcol = l31 & 7;                    // lane -> pixel of a 4 x 8 pixel tile
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // workgroup -> (block, range of tiles): workgroups lid, lid + NB, ... share block lid % NB
    const int g = lid % NB, slot = lid / NB, nwg = ((int)gridDim.x - g + NB - 1) / NB;
    const int TH = p.th, NPT = TH >> 2;                            // tile rows, pixel tiles per tile
    const int tps = p.tiles_x * p.tiles_y, T = p.nbatch * tps;
    const int t_beg = (int)((long long)slot * T / nwg), t_end = (int)((long long)(slot + 1) * T / nwg);
    if (t_beg >= t_end) return;

    // ---- this wave's weights, resident ------------------------------------------------------------------------------------
    bf16x8_t af[NK];
    {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + ((long long)(g * 8 + wave) * NK) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < NK; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) asm volatile("" : "+v"(af[j]));

    // ---- tile-invariant lane constants ---------------------------------------------------------------------------------
    auto fsw = [](int r, int c) { return ((c >> 2) & 1) | ((r & 1) << 1); };
    // halo piece k (16 halo pixels, linear over the 12-pixel pitch): wave w stages pieces w, w + 8, w + 16, w + 24
    const int npiece = ((TH + 2) * AkWs32::PITCH + 15) >> 4;
    int hrel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = i * 8 + wave, hp = 16 * k + (lane >> 2);
        const int r = hp / AkWs32::PITCH, c = hp - r * AkWs32::PITCH;
        hrel[i] = (k < npiece && c < 10 && r < TH + 2) ? (r * p.Wp + c) * CPX + 32 * g + (((lane & 3) ^ fsw(r, c)) << 3) : -1;
    }
    // guide piece k (4 tile rows = 32 pixels x 32 B): wave w stages piece w (k < NPT)
    const int grel = ((4 * wave + (lane >> 4)) * p.W + ((lane >> 1) & 7)) * 8 + (lane & 1) * 4;
    // residual piece k (16 pixels x 64 B): wave w stages pieces w, w + 8 (k < 2 NPT); lane -> (pixel 16 k + lane / 4, physical chunk lane & 3)
    int rrel[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = i * 8 + wave, px = 16 * k + (lane >> 2);
        rrel[i] = (((px >> 3) + 1) * p.Wp + (px & 7) + 1) * CPX + 32 * g + (((lane & 3) ^ ((px >> 2) & 3)) << 3);
    }
    // line mover: thread -> (pixel 128 i + tid / 4, physical chunk tid & 3), i = 0, 1
    int srel[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = 128 * i + (tid >> 2);
        srel[i] = (((px >> 3) + 1) * p.Wp + (px & 7) + 1) * CPX + 32 * g + (((tid & 3) ^ ((px >> 2) & 3)) << 3);
    }
    // B fragment of k step j, pixel tile 0, buffer 0: LDS byte address (pixel tile q: + q QSTEP).
    // CG 32: bt[t] = tap t, channel half 0; step j = tap j / 2, half j & 1: bt[j / 2] ^ 32 [j & 1]
    // CG 16: bt[j] = tap j, the group's chunk pair 2 (w / 4) + lane half
    // CG 8 : bt[j] = tap 2 j (lanes 0-31) | 2 j + 1 (lanes 32-63; tap 9 has zero weights: reads tap 8), the group's chunk w / 2
    constexpr int NBT = (CG == 8) ? 5 : 9;
    unsigned bt[NBT];
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
        const int tap = (CG == 8) ? ((2 * t + hh > 8) ? 8 : 2 * t + hh) : t;
        const int chunk = (CG == 32) ? hh : ((CG == 16) ? 2 * (wave >> 2) + hh : (wave >> 1));
        const int r = prow + tap / 3, c = pcol + tap % 3;
        bt[t] = (r * AkWs32::PITCH + c) * 64 + ((chunk ^ fsw(r, c)) << 4);
    }
    const unsigned tc_lane = AkWs32::OFF_TCS + 4 * 8 * (4 * wave + 2 * hh);            // + 1024 cls: this lane's 16 table entries (2 features x 8 samples)
    const unsigned att_lane = AkWs32::OFF_ATT + l31 * 32;                                // + 1024 q: this lane's pixel of pixel tile q
    // this lane's 4 bytes (features 4 w + 2 hh, + 1) of pixel 32 q + l31 in the staging slot: chunk w / 2 ^ (px >> 2) & 3, + 8 (w & 1) + 4 hh
    const unsigned st_lane = AkWs32::OFF_STAGE + l31 * 64 + (((wave >> 1) ^ ((l31 >> 2) & 3)) << 4) + 8 * (wave & 1) + 4 * hh;   // + 2048 q (32 px; (px >> 2) & 3 unchanged)

    int b, ty, tx;
    {
        b = t_beg / tps;
        const int r = t_beg - b * tps;
        ty = r / p.tiles_x; tx = r - ty * p.tiles_x;
    }
    auto issue_tile = [&](int nb, int nty, int ntx, int buf) {
        const bf16_t* hb = p.h + (long long)nb * p.h_bstride + (long long)(nty * TH * p.Wp + ntx * 8) * CPX;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (hrel[i] >= 0) stage16(hb + hrel[i], smem + buf * AkWs32::HALO + (i * 8 + wave) * 1024, lane);
        if (wave < NPT)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.G + (long long)nb * p.g_bstride + (long long)(nty * TH * p.W + ntx * 8) * 8 + grel),
                                             (LDS_AS void*)(smem + AkWs32::OFF_ATT + buf * AkWs32::ATT + wave * 1024), 16, 0, 0);
        const bf16_t* rb = p.res + (long long)nb * p.res_bstride + (long long)(nty * TH * p.Wp + ntx * 8) * CPX;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (i * 8 + wave < 2 * NPT) stage16(rb + rrel[i], smem + AkWs32::OFF_STAGE + buf * AkWs32::STAGE + (i * 8 + wave) * 1024, lane);
    };
    issue_tile(b, ty, tx, 0);

    int b_cur = -1;
    float rstd = 1.f, aw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) aw[s] = 0.f;
    stat_t S1 = 0, S2 = 0;
    bool have_prev = false;                                          // results of the previous tile wait in its staging slot
    long long prev_out = 0;

#pragma unroll 1
    for (int t = t_beg; t <= t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        // every LDS-DMA of this tile (issued one tile ago) has landed; every wave is done with the previous tile (its results are
        // in the other staging slot, its halo buffer is free)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (have_prev) {                                            // the previous tile's finished 64-byte segments out
            unsigned char* ob = reinterpret_cast<unsigned char*>(p.out) + prev_out;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (128 * i + (tid >> 2) < 32 * NPT) {
                    const u32x4_t ln = *reinterpret_cast<const u32x4_t*>(smem + AkWs32::OFF_STAGE + (buf ^ 1) * AkWs32::STAGE + i * 8192 + tid * 16);
                    *reinterpret_cast<u32x4_t*>(ob + (long long)srel[i] * 2) = ln;
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
            float mean_b = 0.f;
            if (p.own_tc) stat_mean_rstd_wave(p.stats, b, p.inv_count, lane, mean_b, rstd);
            else rstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ms[2 * b + 1])));
            const float inv_b = 1.0f / rstd;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pc = i * 8 + wave;
                if (pc < 9) {
                    const long long rel = (long long)pc * (8 * CPX) + 256 * g + lane * 4;
                    if (p.own_tc) akgm_tc_piece(p, rel, smem + AkWs32::OFF_TCS + pc * 1024 + lane * 16, inv_b, mean_b);
                    else __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.Tc + (long long)b * 9 * (8 * CPX) + rel),
                                                          (LDS_AS void*)(smem + AkWs32::OFF_TCS + pc * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) aw[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.attw[b * 8 + s])));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const bool last = t + 1 == t_end;
        int nb = b, nty = ty, ntx = tx + 1;
        if (ntx == p.tiles_x) { ntx = 0; if (++nty == p.tiles_y) { nty = 0; ++nb; } }
        // (the staging slot buf ^ 1 was just read by the stores above: the DMA of the next tile's residual goes behind them in
        // program order; LDS reads of a wave complete before its later LDS-DMA writes are issued: the data is in registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!last) issue_tile(nb, nty, ntx, buf ^ 1);

        const bool interior = ty > 0 && tx > 0 && ty + 1 < p.tiles_y && tx + 1 < p.tiles_x;
        const unsigned hb0 = buf * AkWs32::HALO, ab0 = buf * AkWs32::ATT, sb0 = buf * AkWs32::STAGE;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
        for (int qp = 0; qp < NPT; qp += 2) {                       // two pixel tiles at a time (NPT is even)
            f32x16_t acc[2];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                unsigned tca = tc_lane + 4 * 1024;                  // class 4
                if (!interior) {
                    const int r = 4 * (qp + tq) + prow, c = pcol;
                    const int cy = (ty == 0 && r == 0) ? 0 : ((ty + 1 == p.tiles_y && r == TH - 1) ? 2 : 1);
                    const int cx = (tx == 0 && c == 0) ? 0 : ((tx + 1 == p.tiles_x && c == 7) ? 2 : 1);
                    tca = tc_lane + (cy * 3 + cx) * 1024;
                }
                acc[tq] = lds_read_f32x16(smem + tca);              // (concatenated reads: no v_mov, asmops.hip.h)
            }
            const unsigned qoff = hb0 + qp * AkWs32::QSTEP;
            // K loop software-pipelined by hand (inline-asm fragment reads, counted lgkmcnt; see akgm_ws.hip.h): the fragments of
            // steps j + 1 and j + 2 are in flight under the two MFMAs of step j (64 matrix-core cycles < one LDS round trip)
            bf16x8_t bfr[3][2];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fold constants; nothing of the compiler's own is queued behind this
            auto frag = [&](auto jc, bf16x8_t (&dst)[2]) {
                constexpr int j = decltype(jc)::value;
                const unsigned a0 = ((CG == 32) ? (bt[j >> 1] ^ ((j & 1) << 5)) : bt[j < NBT ? j : 0]) + qoff;
                lds_read16_asm<0>(dst[0], a0);
                lds_read16_asm<AkWs32::QSTEP>(dst[1], a0);
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
            // ---- modulation sum, swish, residual, statistics; the lane's two features go back into the staging slot -------------
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
                const unsigned sa = st_lane + sb0 + (qp + tq) * 2048;
                const unsigned rv = *reinterpret_cast<const unsigned*>(smem + sa);
                const float v0 = silu_fast(o2[0]) + __builtin_bit_cast(float, rv << 16);
                const float v1 = silu_fast(o2[1]) + __builtin_bit_cast(float, rv & 0xffff0000u);
                s1 += v0 + v1; s2 += v0 * v0 + v1 * v1;
                *reinterpret_cast<unsigned*>(smem + sa) = pack2_bf16(v0, v1);
            }
        }
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
        have_prev = true;
        prev_out = ((long long)b * p.out_bstride + (long long)(ty * TH * p.Wp + tx * 8) * CPX) * 2;
        b = nb; ty = nty; tx = ntx;
    }
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
