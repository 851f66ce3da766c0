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

struct AkWs64 {
    static constexpr int HPOS = 272;                              // halo positions per buffer: NPX + 2 (W + 2) + 2 <= 272
    static constexpr int HALO = HPOS * 128;                       // 34,816 = 17 x 2048: the swizzle survives the buffer switch
    static constexpr int PSTEP = 32 * 128;                        // one pixel tile = 32 positions further
    static constexpr int STAGE = 128 * 64;                        // [128 positions][64 B] residual in / result out
    static constexpr int ATT = 128 * 32;                          // [128 positions][8] fp32
    static constexpr int OFF_STAGE = 2 * HALO;
    static constexpr int OFF_ATT = OFF_STAGE + 2 * STAGE;
    static constexpr int OFF_TCS = OFF_ATT + 2 * ATT;             // [9][256] fp32
    static constexpr int LDS = OFF_TCS + 9 * 1024;                // 103,424
    static constexpr int NK = 36;                                 // k steps: tap j / 4, channels 16 (j & 3) .. + 15
};

// p.th = pixel tiles per tile (2 | 4), p.tiles_x = tiles per sample, p.tiles_y unused (1)
__global__ __launch_bounds__(HC_THREADS, 2) void akgm_ws64_kernel(const AkgmHP p) {
    constexpr int NK = AkWs64::NK;
    constexpr int CPX = 512;                                       // channels per position
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nslots = (int)gridDim.x >> 4;                        // tile ranges per role (the grid is a multiple of 16)
    const int hg = lid / nslots, slot = lid - hg * nslots;         // half group 0 .. 15, range
    const int NPT = p.th, NPX = 32 * NPT;
    const int Wp = p.Wp, Ptot = (p.H + 2) * Wp, Plast = p.H * Wp + p.W;     // positions of a sample; last valid position
    const float inv_wp = 1.0f / (float)Wp;
    const int tps = p.tiles_x, T = p.nbatch * tps;
    const int t_beg = (int)((long long)slot * T / nslots), t_end = (int)((long long)(slot + 1) * T / nslots);
    if (t_beg >= t_end) return;
    const int chan0 = 32 * hg;                                     // this workgroup's first output channel (64 bytes of a position)

    // ---- this wave's weights, resident ------------------------------------------------------------------------------------
    bf16x8_t af[NK];
    {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + ((long long)(hg * 8 + wave) * NK) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < NK; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < NK; ++j) asm volatile("" : "+v"(af[j]));

    // ---- tile-invariant lane constants ---------------------------------------------------------------------------------
    // halo piece k (8 positions x 128 B): wave w stages pieces w, w + 8, ...; lane -> (position 8 k + lane / 8, physical chunk lane & 7)
    const int npiece = (NPX + 2 * Wp + 2 + 7) >> 3;
    int hsw[5];                                                    // channel offset of the lane's logical chunk (elements), per piece
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int hp = 8 * (i * 8 + wave) + (lane >> 3);
        hsw[i] = 64 * (hg >> 1) + (((lane & 7) ^ ((hp >> 1) & 7)) << 3);
    }
    // B fragment of tap t, channel quarter 0, pixel tile 0, buffer 0: LDS byte address (quarter cq: ^ (cq << 5); pixel tile q: + q PSTEP)
    unsigned bt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int hp = l31 + (t / 3) * Wp + t % 3;
        bt[t] = hp * 128 + ((hh ^ ((hp >> 1) & 7)) << 4) + AkWs64::HALO;          // (flipped to buffer 0 at the top of the first tile)
    }
    const unsigned tc_lane = AkWs64::OFF_TCS + 4 * 8 * (4 * wave + 2 * hh);              // + 1024 cls: this lane's 16 table entries (2 features x 8 sets)
    const unsigned att_lane = AkWs64::OFF_ATT + l31 * 32;                                // + 1024 q: this lane's position of pixel tile q
    // this lane's 4 bytes (features 4 w + 2 hh, + 1) of position 32 q + l31 in the staging slot: chunk w / 2 ^ (pos >> 2) & 3, + 8 (w & 1) + 4 hh
    const unsigned st_lane = AkWs64::OFF_STAGE + l31 * 64 + (((wave >> 1) ^ ((l31 >> 2) & 3)) << 4) + 8 * (wave & 1) + 4 * hh;   // + 2048 q
    // residual piece `wave` (16 positions x 64 B): lane -> (position 16 w + lane / 4, physical chunk lane & 3)
    const int rpos = 16 * wave + (lane >> 2);
    const int rsw = chan0 + (((lane & 3) ^ ((rpos >> 2) & 3)) << 3);
    // guide piece `wave` (32 positions x 32 B): lane -> (position 32 w + lane / 2, half lane & 1)
    const int gpos = 32 * wave + (lane >> 1);
    // line mover: thread -> (position tid / 4, physical chunk tid & 3) = its wave's residual piece
    const int mpos = tid >> 2;
    const int msw = chan0 + (((tid & 3) ^ ((mpos >> 2) & 3)) << 3);

    int b = t_beg / tps, ti = t_beg - b * tps;                      // tile t = (sample b, tile ti of the sample)
    auto issue_tile = [&](int nb, int nti, int buf) {
        const int P0 = Wp + 1 + nti * NPX;                          // first output position; the halo starts Wp + 1 positions earlier
        const bf16_t* hb = p.h + (long long)nb * p.h_bstride;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = i * 8 + wave;
            if (k < npiece) {
                int sp = nti * NPX + 8 * k + (lane >> 3);
                sp = sp < Ptot ? sp : Ptot - 1;                     // (the last tile's halo may run past the sample: clamped, feeds dropped positions only)
                stage16(hb + (long long)sp * CPX + hsw[i], smem + buf * AkWs64::HALO + k * 1024, lane);
            }
        }
        if (wave < NPT) {
            int P = P0 + gpos; P = P < Ptot ? P : Ptot - 1;
            int y = fdiv_small(P, inv_wp), x = P - y * Wp;
            y = y < 1 ? 1 : (y > p.H ? p.H : y); x = x < 1 ? 1 : (x > p.W ? p.W : x);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.G + (long long)nb * p.g_bstride + ((y - 1) * p.W + (x - 1)) * 8 + (lane & 1) * 4),
                                             (LDS_AS void*)(smem + AkWs64::OFF_ATT + buf * AkWs64::ATT + wave * 1024), 16, 0, 0);
        }
        if (wave < 2 * NPT) {
            int P = P0 + rpos; P = P < Ptot ? P : Ptot - 1;
            stage16(p.res + (long long)nb * p.res_bstride + (long long)P * CPX + rsw, smem + AkWs64::OFF_STAGE + buf * AkWs64::STAGE + wave * 1024, lane);
        }
    };
    issue_tile(b, ti, 0);

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
            if (prev_out >= 0) {
                const u32x4_t ln = *reinterpret_cast<const u32x4_t*>(smem + AkWs64::OFF_STAGE + (buf ^ 1) * AkWs64::STAGE + tid * 16);
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(p.out) + prev_out) = ln;
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
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pc = i * 8 + wave;
                if (pc < 9)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.Tc + ((long long)b * 9 + pc) * (8 * CPX) + 256 * hg + lane * 4),
                                                     (LDS_AS void*)(smem + AkWs64::OFF_TCS + pc * 1024), 16, 0, 0);
            }
            rstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ms[2 * b + 1])));
#pragma unroll
            for (int s = 0; s < 8; ++s) aw[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.attw[b * 8 + s])));
            HC_WAIT(0);
            asm volatile("s_barrier" ::: "memory");
        }
        const bool last = t + 1 == t_end;
        int nb = b, nti = ti + 1;
        if (nti == tps) { nti = 0; ++nb; }
        // (the staging slot buf ^ 1 was just read by the stores above: the DMA of the next tile's residual goes behind them in
        // program order; LDS reads of a wave complete before its later LDS-DMA writes are issued: the data is in registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!last) issue_tile(nb, nti, buf ^ 1);

        const int P0 = Wp + 1 + ti * NPX;
        const unsigned ab0 = buf * AkWs64::ATT, sb0 = buf * AkWs64::STAGE;
#pragma unroll
        for (int k = 0; k < 9; ++k) bt[k] = buf ? bt[k] + AkWs64::HALO : bt[k] - AkWs64::HALO;       // in place: 9 registers, not 18
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
        for (int qp = 0; qp < NPT; qp += 2) {                       // two pixel tiles at a time (NPT is even)
            f32x16_t acc[2];
            bool valid[2];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const int P = P0 + 32 * (qp + tq) + l31;
                const int y = fdiv_small(P, inv_wp), x = P - y * Wp;
                valid[tq] = x >= 1 && x <= p.W && P <= Plast;
                const int cy = (y <= 1) ? 0 : ((y >= p.H) ? 2 : 1);
                const int cx = (x <= 1) ? 0 : ((x >= p.W) ? 2 : 1);
                const unsigned tca = tc_lane + (cy * 3 + cx) * 1024;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(smem + tca + 16 * g4);
                    acc[tq][4 * g4 + 0] = c4[0]; acc[tq][4 * g4 + 1] = c4[1]; acc[tq][4 * g4 + 2] = c4[2]; acc[tq][4 * g4 + 3] = c4[3];
                }
            }
            // K loop software-pipelined by hand (inline-asm fragment reads, counted lgkmcnt; see akgm_ws.hip.h): the fragments of
            // steps j + 1 and j + 2 are in flight under the two MFMAs of step j (64 matrix-core cycles < one LDS round trip)
            bf16x8_t bfr[3][2];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fold constants; nothing of the compiler's own is queued behind this
            auto frag = [&](auto jc, bf16x8_t (&dst)[2]) {
                constexpr int j = decltype(jc)::value;
                const unsigned a0 = bt[j >> 2] ^ ((j & 3) << 5);
                lds_read16_asm<0>(dst[0], a0);
                lds_read16_asm<AkWs64::PSTEP>(dst[1], a0);
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
                float v0 = silu_fast(o2[0]) + __builtin_bit_cast(float, rv << 16);
                float v1 = silu_fast(o2[1]) + __builtin_bit_cast(float, rv & 0xffff0000u);
                *reinterpret_cast<unsigned*>(smem + sa) = pack2_bf16(v0, v1);
                v0 = valid[tq] ? v0 : 0.f; v1 = valid[tq] ? v1 : 0.f;        // border columns / positions past the sample: dropped
                s1 += v0 + v1; s2 += v0 * v0 + v1 * v1;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) bt[k] += 2 * AkWs64::PSTEP;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) bt[k] -= NPT * AkWs64::PSTEP;
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
        have_prev = true;
        {
            const int P = P0 + mpos;
            const int y = fdiv_small(P < Ptot ? P : Ptot - 1, inv_wp), x = P - y * Wp;
            const bool ok = mpos < NPX && x >= 1 && x <= p.W && P <= Plast;
            prev_out = ok ? ((long long)b * p.out_bstride + (long long)P * CPX + msw) * 2 : -1;
        }
        b = nb; ti = nti;
    }
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
