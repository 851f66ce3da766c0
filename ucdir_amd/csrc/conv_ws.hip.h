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
#include "akgm_ws.hip.h"

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
    int hrel[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int k = i * 8 + wave; k = k > 53 ? 53 : k;
        const int r = k / 3, col = (k - 3 * r) * 8 + (lane >> 3);
        const int hp = r * CvWs::PITCH + col;
        hrel[i] = col < 18 ? (r * p.Wp + col) * 64 + (((lane & 7) ^ ((hp >> 1) & 7)) << 3) : -1;
    }
    // B fragment of tap (ky, kx), chunk pair c16, pixel tile 2 pw (+ QSTEP: 2 pw + 1): halo pixel hp = hp0 + 24 ky + kx,
    // 16-byte chunk (2 c16 + hh) ^ ((hp >> 1) & 7).  (hp + 24) >> 1 adds 12: the swizzle flips bit 2 for ky = 1 and is
    // unchanged for ky = 2, so with K = 2 c16 ^ (ky == 1 ? 4 : 0) in {0, 2, 4, 6} the address is  ba0[kx][K / 2] + 3072 ky.
    unsigned ba0[3][4];
    {
        const int hp0 = (4 * pw + (l31 >> 4)) * CvWs::PITCH + (l31 & 15);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sxh = (((hp0 + kx) >> 1) & 7) ^ hh;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) ba0[kx][k2] = ((hp0 + kx) << 7) + (((2 * k2) ^ sxh) << 4);
        }
    }
    const unsigned tc_lane = CvWs::OFF_TCS + (32 * rw + 16 * hh) * 4;              // + 256 cls: this lane's 16 channels of the fold table
    const unsigned rel_out = (unsigned)(((4 * pw + (l31 >> 4) + 1) * p.Wp + (l31 & 15) + 1) * 64 + 32 * rw + 16 * hh) * 2;   // pixel tile 2 pw; + 2 Wp rows: 2 pw + 1

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
        for (int i = 0; i < 7; ++i) {
            int k = i * 8 + wave; k = k > 53 ? 53 : k;
            if (hrel[i] >= 0) stage16(hb + hrel[i], hd + k * 1024, lane);
        }
    };
    issue_tile(b, ty, tx, 0);

    int b_cur = -1;
    float rstd_a = 1.f;                                             // alpha * rstd of the current sample
    stat_t S1 = 0, S2 = 0;
    const int act = p.act;

#pragma unroll 1
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        const bool last = t + 1 == t_end;
        // every wave drained its VMEM queue at the end of the tile before (its DMA pieces of this tile included)
        if (t == t_beg) { HC_WAIT(0); }
        asm volatile("s_barrier" ::: "memory");
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
        if (!last) issue_tile(nb, nty, ntx, buf ^ 1);

        const unsigned bufh = buf * CvWs::HALO;
        unsigned ba[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) ba[kx][k2] = ba0[kx][k2] + bufh;

        // ---- K loop: 9 taps x 4 chunk pairs, two pixel tiles share every A fragment -------------------------------------
        f32x16_t acc[2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tp][e] = 0.f;
        __builtin_amdgcn_s_setprio(1);
        // software pipeline, depth one: the two fragment reads of step j + 1 are issued in front of the two MFMAs of step j
        // (order pinned with sched_group_barrier: left alone, hipcc hoists dozens of reads and spills 67 registers)
        bf16x8_t bq[2][2];
        auto read_b = [&](int j, int slot) {
            const int tap = j >> 2, c16 = j & 3;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int k2 = c16 ^ (ky == 1 ? 2 : 0);
            const unsigned a0 = ba[kx][k2] + ky * (CvWs::PITCH * 128);
            bq[slot][0] = *reinterpret_cast<const bf16x8_t*>(smem + a0);
            bq[slot][1] = *reinterpret_cast<const bf16x8_t*>(smem + a0 + CvWs::QSTEP);
        };
        __builtin_amdgcn_sched_barrier(0);
        read_b(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int j = 0; j < 36; ++j) {
            if (j + 1 < 36) read_b(j + 1, (j + 1) & 1);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bq[j & 1][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bq[j & 1][1], acc[1], 0, 0, 0);
            if (j + 1 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);

        // ---- epilogue in registers: lane = pixel l31 of the pixel tile, channels 32 rw + 16 hh .. + 15 ----------------------
        const bool interior = ty > 0 && tx > 0 && ty + 1 < p.tiles_y && tx + 1 < p.tiles_x;
        unsigned char* outb = reinterpret_cast<unsigned char*>(reinterpret_cast<bf16_t*>(p.out) + (long long)b * p.out_bstride + (long long)(ty * 16 * p.Wp + tx * 16) * 64);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            unsigned tca = tc_lane + 4 * 256;                       // class 4
            if (!interior) {
                const int r = 4 * pw + 2 * tp + (l31 >> 4), c = l31 & 15;
                const int cy = (ty == 0 && r == 0) ? 0 : ((ty + 1 == p.tiles_y && r == 15) ? 2 : 1);
                const int cx = (tx == 0 && c == 0) ? 0 : ((tx + 1 == p.tiles_x && c == 15) ? 2 : 1);
                tca = tc_lane + (cy * 3 + cx) * 256;
            }
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
            unsigned char* op = outb + rel_out + (long long)tp * (2 * p.Wp * 64 * 2);
            *reinterpret_cast<uint4*>(op) = pack8_bf16(v);
            *reinterpret_cast<uint4*>(op + 16) = pack8_bf16(v + 8);
        }
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
        // this wave's DMA pieces of tile t + 1 must be in LDS before it reaches the next barrier (its own four stores ride along)
        HC_WAIT(0);
        b = nb; ty = nty; tx = ntx;
    }
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
