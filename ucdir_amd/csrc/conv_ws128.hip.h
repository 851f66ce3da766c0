// 3x3 stride-1 convolution 128 -> 64 channels with the block's 1x1 res_conv riding along (conv1 + res_conv of ups.17 /
// ups.18 at the 288^2 level: GroupNorm -> conv1 -> Swish, and res_conv on the raw input; model/ucdir.py:110,120,122-140)
// as a PERSISTENT, weight-stationary kernel with ONE wave per SIMD (gfx950).
//
// conv3x3_halo_kernel<64, true> runs these layers at 600 TFLOP/s (319 us per B = 16 launch).  K = 1152 x 64 rows do not fit
// the 256 registers a wave has at two waves per SIMD (conv_ws.hip.h), but they fit 512:
//   * one workgroup of FOUR wave64 per CU (one per SIMD, the whole 512-entry register file each) walks a contiguous range
//     of 8 x 16 pixel tiles; wave (rw, pw) owns rows 32 rw .. +31 - 72 A fragments = 288 VGPRs, loaded once - and the 64
//     pixels (two 32-pixel MFMA tiles = tile rows 4 pw .. 4 pw + 3) of every tile;
//   * the input is the channel concatenation of two 64-channel tensors (block input and skip) that is never materialised:
//     the halo in LDS is two PLANES, one per tensor, each in conv_ws.hip.h's format ([10 rows][24-pixel pitch][64 ch] bf16,
//     chunk XOR (pixel >> 1) & 7): every DMA instruction reads one tensor (wave-uniform base + 32-bit lane offset), the
//     plane of a K step is a compile-time constant, 61 KB per buffer, two buffers;
//   * K loop: 72 x { two ds_read_b128, two MFMAs }, fragment reads three steps ahead (a lone wave has nobody to hide its
//     LDS latency); then the res_conv as a second, 8-step pass over the centre tap with its A fragments read from LDS;
//   * epilogues in registers as in conv_ws.hip.h (a lane holds 16 consecutive channels of one pixel); the wave's DMA pieces
//     of tile t + 1 are waited for between the arithmetic and the stores; ONE barrier per tile.
#pragma once
#include <type_traits>
#include "conv_ws.hip.h"

struct CvWs128 {
    static constexpr int PITCH = 24;
    static constexpr int PLANE = 10 * PITCH * 128;                // 30,720: [10][24][64 ch] bf16 of one input tensor
    static constexpr int HALO = 2 * PLANE;                        // 61,440
    static constexpr int QSTEP = 2 * PITCH * 128;                 // 6,144: one 32-pixel MFMA tile = two tile rows further
    static constexpr int OFF_A2 = 2 * HALO;                       // res_conv A fragments [rw][8 steps][hh][32 rows][8] bf16
    static constexpr int A2_BYTES = 2 * 8 * 2 * 32 * 16;          // 16,384
    static constexpr int OFF_TCS = OFF_A2 + A2_BYTES;             // [9][64] fp32
    static constexpr int OFF_B2 = OFF_TCS + 9 * 64 * 4;           // res_conv bias [64] fp32
    static constexpr int OFF_SCAL = OFF_B2 + 64 * 4;
    static constexpr int LDS = OFF_SCAL + 128;                    // 141,952
    static constexpr int THREADS = 256;
    static constexpr int DEPTH = 3;                               // B fragment reads issued this many K steps ahead
};

__global__ __launch_bounds__(CvWs128::THREADS, 1) void conv_ws128_kernel(const GemmP p) {
    using L = CvWs128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const tcs = reinterpret_cast<float*>(smem + L::OFF_TCS);
    float* const b2s = reinterpret_cast<float*>(smem + L::OFF_B2);
    float* const scal = reinterpret_cast<float*>(smem + L::OFF_SCAL);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave & 1, pw = wave >> 1;
    const int prow = l31 >> 4, pcol = (l31 & 15) ^ (prow << 3);    // conflict-free lane -> pixel mapping (conv_ws.hip.h)
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 1);
    int dbg_n = 0;
#define C8_STAMP() do { if (dbg_on && dbg_n < 250) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C8_STAMP() do {} while (0)
#endif
    C8_STAMP();
    const int tps = p.tiles_x * p.tiles_y;
    const int T = p.nbatch * tps;
    const int t_beg = (int)((long long)lid * T / (int)gridDim.x), t_end = (int)((long long)(lid + 1) * T / (int)gridDim.x);
    if (t_beg >= t_end) return;

    // ---- this wave's weights: 72 A fragments (9 taps x 8 chunk pairs), resident for the whole launch -------------------------
    bf16x8_t af[72];
    {
        const bf16_t* Ab = p.A + ((rw * 72 * 2 + hh) * 32 + l31) * 8;
#pragma unroll
        for (int j = 0; j < 72; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * (2 * 32 * 8));
    }
#pragma unroll
    for (int j = 0; j < 72; ++j) asm volatile("" : "+v"(af[j]));
    // res_conv fragments and bias -> LDS (linear copies)
    for (int i = tid; i < L::A2_BYTES / 16; i += L::THREADS)
        reinterpret_cast<uint4*>(smem + L::OFF_A2)[i] = reinterpret_cast<const uint4*>(p.alt_A)[i];
    if (tid < 64) b2s[tid] = p.bias2 ? p.bias2[tid] : 0.f;
    __syncthreads();

    // ---- halo staging: piece (plane, r, c3) = row r, pixel columns 8 c3 .. 8 c3 + 7 of one tensor -> plane bytes (3 r + c3) * 1024.
    // Wave w stages rows w, w + 4 and (w < 2) w + 8 of both planes; the per-lane source offset of row r + 4 is that of row r plus a
    // wave-uniform 4 rows (same swizzle): 3 offset registers.
    unsigned hrel[3];                                              // BYTE offsets (unsigned 32-bit: zero-extended at the use, nothing 64-bit lives across the tile loop)
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
        const int col = c3 * 8 + (lane >> 3), hp = wave * L::PITCH + col;
        hrel[c3] = col < 18 ? (unsigned)((wave * p.Wp + col) * 64 + (((lane & 7) ^ ((hp >> 1) & 7)) << 3)) * 2u : 0xffffffffu;
    }
    // B fragment of tap (ky, kx), chunk pair c16 (plane c16 / 4), pixel tile 2 pw (+ QSTEP: 2 pw + 1): as in conv_ws.hip.h,
    // address = plane * PLANE + ba[kx][(c16 & 3) ^ (ky == 1 ? 2 : 0)] + 3072 ky
    unsigned ba[3][4];
    {
        const int hp0 = (4 * pw + prow) * L::PITCH + pcol;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sxh = (((hp0 + kx) >> 1) & 7) ^ hh;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) ba[kx][k2] = ((hp0 + kx) << 7) + (((2 * k2) ^ sxh) << 4);
        }
    }
    const unsigned tc_lane = L::OFF_TCS + (32 * rw + 16 * hh) * 4;
    const unsigned b2_lane = L::OFF_B2 + (32 * rw + 16 * hh) * 4;
    const unsigned a2_lane = L::OFF_A2 + ((rw * 8 * 2 + hh) * 32 + l31) * 16;        // + 1024 j
    const unsigned rel_out = (unsigned)(((4 * pw + prow + 1) * p.Wp + pcol + 1) * 64 + 32 * rw + 16 * hh) * 2;   // pixel tile 2 pw; + 2 Wp rows: 2 pw + 1

    int b, ty, tx;
    {
        b = t_beg / tps;
        const int r = t_beg - b * tps;
        ty = r / p.tiles_x; tx = r - ty * p.tiles_x;
    }
    auto issue_tile = [&](int nb, int nty, int ntx, int buf) {
        const long long pix0 = (long long)nty * 8 * p.Wp + ntx * 16;
        const unsigned char* h0 = reinterpret_cast<const unsigned char*>(p.B0 + (long long)nb * p.b0_bstride + pix0 * 64);
        const unsigned char* h1 = reinterpret_cast<const unsigned char*>(p.B1 + (long long)nb * p.b1_bstride + pix0 * 64);
        unsigned char* hd = smem + buf * L::HALO;
        const long long r4 = (long long)4 * p.Wp * 64 * 2;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
            if (hrel[c3] != 0xffffffffu) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < 2 || wave < 2) {                       // rows w, w + 4, w + 8 (< 10)
                        stage16(reinterpret_cast<const bf16_t*>(h0 + k * r4 + hrel[c3]), hd + (3 * (wave + 4 * k) + c3) * 1024, lane);
                        stage16(reinterpret_cast<const bf16_t*>(h1 + k * r4 + hrel[c3]), hd + L::PLANE + (3 * (wave + 4 * k) + c3) * 1024, lane);
                    }
            }
    };
    issue_tile(b, ty, tx, 0);

    int b_cur = -1;
    float rstd_a = 1.f;
    stat_t S1 = 0, S2 = 0;
    const int act = p.act;

#pragma unroll 1
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        const bool last = t + 1 == t_end;
        C8_STAMP();
        if (t == t_beg) { HC_WAIT(0); }
        asm volatile("s_barrier" ::: "memory");
        C8_STAMP();                                                 // behind the barrier
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
                    if (lane < 2 * UCDIR_STAT_SLOTS) {
                        v = p.stats0[(long long)b * (2 * UCDIR_STAT_SLOTS) + lane];
                        if (p.stats1) v += p.stats1[(long long)b * (2 * UCDIR_STAT_SLOTS) + lane];
                    }
#pragma unroll
                    for (int off = 2; off < 2 * UCDIR_STAT_SLOTS; off <<= 1) v += __shfl_xor(v, off);
                    const long long q = __shfl(v, 1);
                    mean_rstd(stat_val(v), stat_val(q), p.inv_count, mean, rstd);
                }
                if (lane == 0) { scal[0] = mean; scal[1] = rstd; }
            }
            __syncthreads();
            const float mean = scal[0], rstd = scal[1];
            for (int i = tid; i < 9 * 64; i += L::THREADS) {
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
        C8_STAMP();

        if (t != t_beg) {                                           // the fragment addresses follow the halo buffer
            const int d = buf ? L::HALO : -L::HALO;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) ba[kx][k2] += d;
        }

        // ---- K loop: 9 taps x 8 chunk pairs; two pixel tiles share every A fragment.  Fragment reads DEPTH steps ahead by inline
        // asm with counted lgkmcnt waits; the DMA pieces of tile t + 1 are issued one at a time between the MFMAs ---------------
        f32x16_t acc[2];
        {
            bf16x8_t bq[L::DEPTH + 1][2];
            const long long npix0 = (long long)nty * 8 * p.Wp + ntx * 16;
            const unsigned char* h0 = reinterpret_cast<const unsigned char*>(p.B0 + (long long)nb * p.b0_bstride + npix0 * 64);
            const unsigned char* h1 = reinterpret_cast<const unsigned char*>(p.B1 + (long long)nb * p.b1_bstride + npix0 * 64);
            unsigned char* const hd = smem + (buf ^ 1) * L::HALO;
            const long long r4 = (long long)4 * p.Wp * 64 * 2;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // nothing of the compiler's own is in the LDS / scalar queue from here on
            static_for<0, L::DEPTH>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int tap = j >> 3, c16 = j & 7, ky = tap / 3, kx = tap - 3 * ky;
                constexpr int off = (c16 >> 2) * L::PLANE + ky * (L::PITCH * 128);
                lds_read16_asm<off>(bq[j][0], ba[kx][(c16 & 3) ^ (ky == 1 ? 2 : 0)]);
                lds_read16_asm<off + L::QSTEP>(bq[j][1], ba[kx][(c16 & 3) ^ (ky == 1 ? 2 : 0)]);
            });
            static_for<0, 72>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + L::DEPTH < 72) {
                    constexpr int jn = j + L::DEPTH;
                    constexpr int tap = jn >> 3, c16 = jn & 7, ky = tap / 3, kx = tap - 3 * ky;
                    constexpr int off = (c16 >> 2) * L::PLANE + ky * (L::PITCH * 128);
                    lds_read16_asm<off>(bq[jn % (L::DEPTH + 1)][0], ba[kx][(c16 & 3) ^ (ky == 1 ? 2 : 0)]);
                    lds_read16_asm<off + L::QSTEP>(bq[jn % (L::DEPTH + 1)][1], ba[kx][(c16 & 3) ^ (ky == 1 ? 2 : 0)]);
                }
                if constexpr (j % 4 == 1 && j / 4 < 18) {           // DMA piece e = j / 4: (c3, row block k, plane)
                    constexpr int e = j / 4, c3 = e / 6, k = (e % 6) >> 1, pl = e & 1;
                    if (!last && (k < 2 || wave < 2) && hrel[c3] != 0xffffffffu)
                        stage16(reinterpret_cast<const bf16_t*>((pl ? h1 : h0) + k * r4 + hrel[c3]),
                                hd + pl * L::PLANE + (3 * (wave + 4 * k) + c3) * 1024, lane);
                }
                constexpr int ahead = (71 - j) < L::DEPTH ? (71 - j) : L::DEPTH;      // steps whose reads are younger than step j's
                lgkm_wait_asm<2 * ahead>();
                const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bq[j % (L::DEPTH + 1)][0], j ? acc[0] : zero, 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bq[j % (L::DEPTH + 1)][1], j ? acc[1] : zero, 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
        C8_STAMP();                                                 // K loop done
        // ---- res_conv: centre tap, 8 chunk pairs, A fragments from LDS ------------------------------------------------------
        f32x16_t acc2[2];
        {
            const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8_t a2 = *reinterpret_cast<const bf16x8_t*>(smem + a2_lane + j * 1024);
                const unsigned a0 = ba[1][(j & 3) ^ 2] + (j >> 2) * L::PLANE + L::PITCH * 128;
                const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(smem + a0);
                const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(smem + a0 + L::QSTEP);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, j ? acc2[0] : zero, 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, j ? acc2[1] : zero, 0, 0, 0);
            }
        }

        C8_STAMP();                                                 // res_conv done
        // ---- epilogues in registers: lane = pixel (prow, pcol) of the pixel tile, channels 32 rw + 16 hh .. + 15 ------------------
        const bool interior = ty > 0 && tx > 0 && ty + 1 < p.tiles_y && tx + 1 < p.tiles_x;
        const long long tile_el = (long long)(ty * 8 * p.Wp + tx * 16) * 64;
        unsigned char* outb = reinterpret_cast<unsigned char*>(reinterpret_cast<bf16_t*>(p.out) + (long long)b * p.out_bstride + tile_el);
        unsigned char* out2b = reinterpret_cast<unsigned char*>(p.out2 + (long long)b * p.out2_bstride + tile_el);
        float s1 = 0.f, s2 = 0.f;
        uint4 pk[2][2], pk2[2][2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            unsigned tca = tc_lane + 4 * 256;                       // class 4
            if (!interior) {
                const int r = 4 * pw + 2 * tp + prow, c = pcol;
                const int cy = (ty == 0 && r == 0) ? 0 : ((ty + 1 == p.tiles_y && r == 7) ? 2 : 1);
                const int cx = (tx == 0 && c == 0) ? 0 : ((tx + 1 == p.tiles_x && c == 15) ? 2 : 1);
                tca = tc_lane + (cy * 3 + cx) * 256;
            }
            float v[16], w[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_t c4 = *reinterpret_cast<const f32x4_t*>(smem + tca + 16 * g4);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(smem + b2_lane + 16 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[4 * g4 + e] = fmaf(acc[tp][4 * g4 + e], rstd_a, c4[e]);
                    w[4 * g4 + e] = acc2[tp][4 * g4 + e] + d4[e];
                }
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
            pk2[tp][0] = pack8_bf16(w); pk2[tp][1] = pack8_bf16(w + 8);
        }
        C8_STAMP();                                                 // epilogue arithmetic done
        // this wave's DMA pieces of tile t + 1 must be in LDS before it reaches the next barrier: wait here, in front of the stores
        HC_WAIT(0);
        C8_STAMP();                                                 // vmcnt(0) passed
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const long long o = rel_out + (long long)tp * (2 * p.Wp * 64 * 2);
            *reinterpret_cast<uint4*>(outb + o) = pk[tp][0];
            *reinterpret_cast<uint4*>(outb + o + 16) = pk[tp][1];
            *reinterpret_cast<uint4*>(out2b + o) = pk2[tp][0];
            *reinterpret_cast<uint4*>(out2b + o + 16) = pk2[tp][1];
        }
        S1 += stat_fx((double)s1); S2 += stat_fx((double)s2);
        b = nb; ty = nty; tx = ntx;
    }
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    if (p.stats_out) {
        const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
        if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
    }
}
