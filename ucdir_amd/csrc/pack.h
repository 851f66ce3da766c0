// Host-side weight packing for the MFMA kernels (pure C++, no device code).
//
// * dense conv (3x3 / 1x1): A[o][k], k = tap*Cin + c (tap = ky*3+kx), bf16, optional fold of
//   the preceding GroupNorm's gamma into the weights;  Tb/Tg border-class tables for the fold
//   (see cgemm.hip.h header).
// * spdyconv (grouped, groups = 8, reference model/ucdir.py:116,136-137): per group a
//   [C][Kpad] matrix whose rows are permuted so that, in the 32x32 MFMA accumulator layout, one
//   lane holds all 8 kernel sets of a feature (row rho of a 32-row tile <-> feature
//   4*t + 2*((rho>>4)&1) + ((rho>>2)&1), set (rho&3) + 4*((rho>>3)&1)).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "common.h"

struct PackedConv {
    std::vector<bf16_t> A;        // [rows_pad][Kpad]
    std::vector<float> bias;      // [cout] (zeros if none)
    std::vector<float> Tb, Tg;    // [ncls][cout]
    int rows_pad = 0, Kpad = 0, ntaps = 0, cin = 0, cout = 0, ncls = 0;
};

static inline bool tap_valid(int cls, int tap) {
    const int cy = cls / 3, cx = cls % 3, ky = tap / 3, kx = tap % 3;
    if (cy == 0 && ky == 0) return false;
    if (cy == 2 && ky == 2) return false;
    if (cx == 0 && kx == 0) return false;
    if (cx == 2 && kx == 2) return false;
    return true;
}

// w: [cout][cin][ks][ks] fp32 (reference Conv2d layout)
static inline PackedConv pack_conv(const float* w, const float* bias, const float* gamma, const float* beta,
                                   int cout, int cin, int ks, int TM) {
    PackedConv P;
    P.ntaps = ks * ks; P.cin = cin; P.cout = cout;
    P.Kpad = ((P.ntaps * cin + 63) / 64) * 64;
    P.rows_pad = ((cout + TM - 1) / TM) * TM;
    P.A.assign((size_t)P.rows_pad * P.Kpad, 0);
    P.bias.assign(P.rows_pad, 0.f);        // padded: the epilogue reads 8 features at a time
    if (bias) for (int o = 0; o < cout; ++o) P.bias[o] = bias[o];
    P.ncls = (ks == 3) ? 9 : 1;
    const bool fold = gamma != nullptr;
    std::vector<double> wb((size_t)P.ntaps * cout, 0.0), wg((size_t)P.ntaps * cout, 0.0);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < P.ntaps; ++t) {
                const float wv = w[((size_t)o * cin + c) * P.ntaps + t];
                const float ws = fold ? wv * gamma[c] : wv;
                const bf16_t q = f2bf(ws);
                P.A[(size_t)o * P.Kpad + (size_t)t * cin + c] = q;
                if (fold) {
                    wb[(size_t)t * cout + o] += (double)wv * beta[c];
                    wg[(size_t)t * cout + o] += (double)bf2f(q);
                }
            }
    if (fold) {
        P.Tb.assign((size_t)P.ncls * cout, 0.f);
        P.Tg.assign((size_t)P.ncls * cout, 0.f);
        for (int cls = 0; cls < P.ncls; ++cls)
            for (int o = 0; o < cout; ++o) {
                double sb = 0, sg = 0;
                for (int t = 0; t < P.ntaps; ++t) {
                    const int t9 = (P.ntaps == 9) ? t : 4;
                    if (P.ncls == 9 && !tap_valid(cls, t9)) continue;
                    sb += wb[(size_t)t * cout + o]; sg += wg[(size_t)t * cout + o];
                }
                P.Tb[(size_t)cls * cout + o] = (float)sb;
                P.Tg[(size_t)cls * cout + o] = (float)sg;
            }
    }
    return P;
}

// conv_ws_kernel (conv_ws.hip.h), 3x3 conv 64 -> 64: A fragments as the waves load them, [rw][step j][hh][32 rows][8] bf16 with
// j = 4 tap + c16 (input channels 16 c16 + 8 hh .. + 7) and MFMA row rho of row tile rw <-> output channel
// 32 rw + 16 ((rho >> 2) & 1) + (rho & 3) + 4 (rho >> 3): in the 32x32 accumulator layout (row = (reg & 3) + 8 (reg >> 2) + 4 hh)
// register r of lane half hh is then channel 32 rw + 16 hh + r - 16 consecutive channels per lane.
// P: pack_conv(...) of the same layer with Kpad == 9 * 64 (GroupNorm gamma already folded in).
static inline std::vector<bf16_t> pack_conv_ws(const PackedConv& P) {
    std::vector<bf16_t> img((size_t)2 * 36 * 2 * 32 * 8, 0);
    for (int rw = 0; rw < 2; ++rw)
        for (int j = 0; j < 36; ++j)
            for (int hk = 0; hk < 2; ++hk)
                for (int rho = 0; rho < 32; ++rho) {
                    const int o = 32 * rw + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
                    const int tap = j >> 2, c0 = 16 * (j & 3) + 8 * hk;
                    for (int e = 0; e < 8; ++e)
                        img[((((size_t)rw * 36 + j) * 2 + hk) * 32 + rho) * 8 + e] = P.A[(size_t)o * P.Kpad + (size_t)tap * 64 + c0 + e];
                }
    return img;
}

// conv_ws128_kernel (conv_ws128.hip.h), 3x3 conv 128 -> 64 + the block's 1x1 res_conv: from the 10-tap image of pack_conv_res10
// ([rows][10 * 128], k = tap * 128 + c, tap 9 = res_conv) the A fragments [rw][step j][hh][32 rows][8] bf16 with j = 8 tap + c16
// (input channels 16 c16 + 8 hh .. + 7), rows permuted as in pack_conv_ws; the res_conv's 8 steps follow the 72 of the 3x3.
static inline std::vector<bf16_t> pack_conv_ws128(const std::vector<bf16_t>& A10) {
    const int K10 = 10 * 128, nj = 80;
    std::vector<bf16_t> img((size_t)2 * nj * 2 * 32 * 8, 0);
    for (int rw = 0; rw < 2; ++rw)
        for (int j = 0; j < nj; ++j)
            for (int hk = 0; hk < 2; ++hk)
                for (int rho = 0; rho < 32; ++rho) {
                    const int o = 32 * rw + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
                    const int tap = j >> 3, c0 = 16 * (j & 7) + 8 * hk;
                    // main image: [rw][72][hh][32][8] first, then the res image [rw][8][hh][32][8] (two separate blocks, see below)
                    const size_t dst = j < 72 ? ((((size_t)rw * 72 + j) * 2 + hk) * 32 + rho) * 8
                                              : (size_t)2 * 72 * 2 * 32 * 8 + ((((size_t)rw * 8 + (j - 72)) * 2 + hk) * 32 + rho) * 8;
                    for (int e = 0; e < 8; ++e) img[dst + e] = A10[(size_t)o * K10 + (size_t)tap * 128 + c0 + e];
                }
    return img;
}

// conv1 of a residual block whose res_conv is fused into it (conv3x3_halo_kernel<64>): the packed rows get a 10th
// tap block holding the 1x1 res_conv weights (plain bf16, no GroupNorm fold: res_conv sees the raw input).
// P: pack_conv(conv1) with Kpad == 9*cin; wres: [cout][cin].  Returns [rows_pad][10*cin].
static inline std::vector<bf16_t> pack_conv_res10(const PackedConv& P, const float* wres) {
    const int cin = P.cin, K10 = 10 * cin;
    std::vector<bf16_t> A((size_t)P.rows_pad * K10, 0);
    for (int o = 0; o < P.rows_pad; ++o) {
        for (int k = 0; k < 9 * cin; ++k) A[(size_t)o * K10 + k] = P.A[(size_t)o * P.Kpad + k];
        if (o < P.cout)
            for (int c = 0; c < cin; ++c) A[(size_t)o * K10 + 9 * cin + c] = f2bf(wres[(size_t)o * cin + c]);
    }
    return A;
}

struct PackedAkgm {
    std::vector<bf16_t> A;        // [8 groups][C rows][Kpad]
    std::vector<float> bias;      // [8C] original order
    std::vector<float> Tb, Tg;    // [9][8C] original order
    int C = 0, cg = 0, Kpad = 0;
};

// wsp: [8C][C/8][3][3], bsp: [8C], gamma/beta: [C] (norm2)
static inline PackedAkgm pack_akgm(const float* wsp, const float* bsp, const float* gamma, const float* beta, int C,
                                   int kpad_override = 0) {
    PackedAkgm P;
    P.C = C; P.cg = C / 8;
    const int cg = P.cg;
    // cg == 64 (akgm_halo): K is ordered [32-channel chunk][tap][32] with each chunk padded to 320, so
    // that every 64-k A stage belongs to one halo chunk; otherwise k = tap*cg + ci.
    const bool chunked = (cg == 64) && !kpad_override;
    P.Kpad = kpad_override ? kpad_override : (chunked ? 640 : ((9 * cg + 63) / 64) * 64);
    P.A.assign((size_t)8 * C * P.Kpad, 0);
    P.bias.assign((size_t)8 * C, 0.f);
    P.Tb.assign((size_t)9 * 8 * C, 0.f);
    P.Tg.assign((size_t)9 * 8 * C, 0.f);
    for (int o = 0; o < 8 * C; ++o) P.bias[o] = bsp[o];
    for (int g = 0; g < 8; ++g)
        for (int pr = 0; pr < C; ++pr) {
            const int t32 = pr / 32, rho = pr % 32;
            const int floc = 4 * t32 + ((rho >> 4) & 1) * 2 + ((rho >> 2) & 1);
            const int s = (rho & 3) + 4 * ((rho >> 3) & 1);
            const int o = g * C + 8 * floc + s;
            bf16_t* row = &P.A[((size_t)g * C + pr) * P.Kpad];
            double wb[9] = {0}, wg[9] = {0};
            for (int ci = 0; ci < cg; ++ci) {
                const int cglob = g * cg + ci;
                for (int t = 0; t < 9; ++t) {
                    const float wv = wsp[((size_t)o * cg + ci) * 9 + t];
                    const bf16_t q = f2bf(wv * gamma[cglob]);
                    row[chunked ? (size_t)(ci / 32) * 320 + (size_t)t * 32 + (ci % 32) : (size_t)t * cg + ci] = q;
                    wb[t] += (double)wv * beta[cglob];
                    wg[t] += (double)bf2f(q);
                }
            }
            for (int cls = 0; cls < 9; ++cls) {
                double sb = 0, sg = 0;
                for (int t = 0; t < 9; ++t) if (tap_valid(cls, t)) { sb += wb[t]; sg += wg[t]; }
                P.Tb[(size_t)cls * 8 * C + o] = (float)sb;
                P.Tg[(size_t)cls * 8 * C + o] = (float)sg;
            }
        }
    return P;
}

// LDS image of the spdyconv weights for akgm_pre.hip.h (cg = 8 / 16): [unit][k16 step][k half][128 rows][8],
// a unit being 128 MFMA rows = 16 output features x 8 kernel sets (cg 8: two adjacent groups; cg 16: one group).
// Row R of a unit: 32-row tile t32 = R/32, rho = R%32 -> lane half hr = (rho>>2)&1, q = (rho>>4)&1, set s; feature
// 4*t32 + 2*hr + q, so a lane's two features are adjacent.  k: cg 8 -> (tap 2j + half, channel e), tap 9 = 0;
// cg 16 -> (tap j, channel 8*half + e).  Values are W*gamma rounded exactly as in pack_akgm (same fold tables).
static inline std::vector<bf16_t> pack_akgm_pre(const float* wsp, const float* gamma, int C) {
    const int cg = C / 8;
    const int nk16 = (cg == 8) ? 5 : 9;
    const int nunits = C / 16;
    std::vector<bf16_t> img((size_t)nunits * nk16 * 2 * 128 * 8, 0);
    for (int U = 0; U < nunits; ++U)
        for (int R = 0; R < 128; ++R) {
            const int t32 = R >> 5, rho = R & 31;
            const int hr = (rho >> 2) & 1, q = (rho >> 4) & 1, s = (rho & 3) + 4 * ((rho >> 3) & 1);
            const int c = U * 16 + 4 * t32 + 2 * hr + q;          // output feature
            const int g = c / cg;                                  // its group
            const int o = 8 * c + s;
            for (int j = 0; j < nk16; ++j)
                for (int hk = 0; hk < 2; ++hk)
                    for (int e = 0; e < 8; ++e) {
                        const int tap = (cg == 8) ? 2 * j + hk : j;
                        const int ci = (cg == 8) ? e : 8 * hk + e;
                        if (tap > 8) continue;
                        img[(((size_t)U * nk16 + j) * 2 + hk) * 1024 + (size_t)R * 8 + e] =
                            f2bf(wsp[((size_t)o * cg + ci) * 9 + tap] * gamma[g * cg + ci]);
                    }
        }
    return img;
}

// A / B fragments of qkv_ws_kernel (qkv_ws.hip.h): [row tile rt of 256][wave 8][k step j = C/16][lane half hh][32 rows][8] bf16;
// lane (hh, rho) of wave w holds W[R][16 j + 8 hh .. + 7].  q / k row tiles (rows < 2C): R = 256 rt + 32 w + 16 ((rho >> 2) & 1) +
// 4 (rho >> 3) + (rho & 3) - the MFMA's output rows 8 a + 4 hh + i of a lane are then 16 consecutive features; v' row tiles: R = 256 rt
// + 32 w + rho (the registers are the B operand there, a lane = one feature).  P: pack_conv image of the 1x1 conv, [rows_pad][Kpad].
static inline std::vector<bf16_t> pack_qkv_ws(const PackedConv& P, int C) {
    const int RT = 3 * C / 256, NKS = C / 16;
    std::vector<bf16_t> img((size_t)RT * 8 * NKS * 2 * 32 * 8, 0);
    for (int rt = 0; rt < RT; ++rt)
        for (int w = 0; w < 8; ++w)
            for (int rho = 0; rho < 32; ++rho) {
                const bool v = rt * 256 >= 2 * C;
                const int R = 256 * rt + 32 * w + (v ? rho : 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3));
                for (int j = 0; j < NKS; ++j)
                    for (int hh = 0; hh < 2; ++hh)
                        for (int e = 0; e < 8; ++e)
                            img[(((((size_t)rt * 8 + w) * NKS + j) * 2 + hh) * 32 + rho) * 8 + e] = P.A[(size_t)R * P.Kpad + 16 * j + 8 * hh + e];
            }
    return img;
}

// conv3x3_halo_kernel<TM>'s weight stages as they sit in LDS, [row tile][32-channel chunk c][K step u][piece k 16][lane 64][8] bf16:
// piece k = rows (k % (TM/16)) * 16 .. + 15 of tap TPS u + k / (TM/16) (tail taps repeat tap 8), lane -> (row lane / 4, physical
// 16-byte chunk lane & 3 = logical ^ ((row >> 2) & 3)).  A stage is then 16 KB of contiguous memory.  P: pack_conv image [rows_pad][Kpad].
static inline std::vector<bf16_t> pack_conv_tiled(const PackedConv& P, int TM) {
    const int TPS = 256 / TM, nsteps = (9 + TPS - 1) / TPS, nch = P.cin / 32, nrt = P.rows_pad / TM, ipt = TM / 16;
    std::vector<bf16_t> img((size_t)nrt * nch * nsteps * 8192, 0);
    for (int rt = 0; rt < nrt; ++rt)
        for (int c = 0; c < nch; ++c)
            for (int u = 0; u < nsteps; ++u)
                for (int k = 0; k < 16; ++k)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = (k % ipt) * 16 + (lane >> 2);
                        int tap = TPS * u + k / ipt; tap = tap < 9 ? tap : 8;
                        const int lc = (lane & 3) ^ ((row >> 2) & 3);
                        const bf16_t* src = &P.A[(size_t)(rt * TM + row) * P.Kpad + (size_t)tap * P.cin + c * 32 + lc * 8];
                        bf16_t* dst = &img[((((size_t)rt * nch + c) * nsteps + u) * 16 + k) * 512 + lane * 8];
                        for (int e = 0; e < 8; ++e) dst[e] = src[e];
                    }
    return img;
}

// conv_sk_kernel<MW, NTAPS> (conv_sk.hip.h): every K sub-step's weights as they sit in LDS,
// [parity class][row tile of 128 MW][32-channel chunk c][tap t][fragment f (4 MW)][k16 j (2)][lane 64][8] bf16: lane (hh = lane >> 5,
// rho = lane & 31) of fragment f holds W[row][t * cin + 32 c + 16 j + 8 hh .. + 7] with row = 32 f + 16 ((rho >> 2) & 1) + (rho & 3) +
// 4 (rho >> 3): in the 32 x 32 accumulator layout register r of lane half hh is then channel 32 f + 16 hh + r (16 consecutive channels
// per lane).  A stage (one tap x 32 channels x all rows of the tile) is 8 MW KB of contiguous memory, fragment reads are lane-linear.
// A: [npar][rows_pad][Kld] with k = tap * cin + c (pack_conv / pack_upconv images); rows beyond rows_pad are zero.
static inline std::vector<bf16_t> pack_conv_sk(const std::vector<bf16_t>& A, int npar, int rows_pad, int Kld, int cin, int ntaps, int MW) {
    const int ROWS = 128 * MW, nrt = (rows_pad + ROWS - 1) / ROWS, nch = cin / 32, nf = 4 * MW;
    // (+ four stages of zeros: the kernel's weight ring requests up to four stages past the end of a row tile's sequence)
    std::vector<bf16_t> img((size_t)npar * nrt * nch * ntaps * nf * 2 * 512 + (size_t)4 * nf * 2 * 512, 0);
    for (int par = 0; par < npar; ++par)
        for (int rt = 0; rt < nrt; ++rt)
            for (int c = 0; c < nch; ++c)
                for (int t = 0; t < ntaps; ++t)
                    for (int f = 0; f < nf; ++f)
                        for (int j = 0; j < 2; ++j)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int rho = lane & 31, hh = lane >> 5;
                                const int row = rt * ROWS + 32 * f + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
                                if (row >= rows_pad) continue;
                                const bf16_t* src = &A[((size_t)par * rows_pad + row) * Kld + (size_t)t * cin + 32 * c + 16 * j + 8 * hh];
                                bf16_t* dst = &img[(((((((size_t)par * nrt + rt) * nch + c) * ntaps + t) * nf + f) * 2 + j) * 64 + lane) * 8];
                                for (int e = 0; e < 8; ++e) dst[e] = src[e];
                            }
    return img;
}

// A fragments of akgm_ws32_kernel<CG> (akgm_ws32.hip.h): [32-feature block C/32][wave 8][k step j NK][lane half hh][32 rows][8] bf16.
// Wave w of block blk owns features c = 32 blk + 4 w + c4; MFMA row rho <-> (c4 = 2 ((rho >> 2) & 1) + (rho >> 4), sample s = 4 ((rho >> 3) & 1)
// + (rho & 3)): a lane's 16 accumulators (rows 8 a + 4 hh + i) are then the 8 samples of features 2 hh and 2 hh + 1.  k step j, lane half hk:
// cg 32: tap j / 2, channels 16 (j & 1) + 8 hk ..;  cg 16: tap j, channels 8 hk ..;  cg 8: tap 2 j + hk (tap 9 = zero), channels 0 .. 7.
// wsp: [8C][cg][3][3] (row o = 8 c + s), gamma: [C].
static inline std::vector<bf16_t> pack_akgm_ws32(const float* wsp, const float* gamma, int C) {
    const int cg = C / 8, nk = (cg == 32) ? 18 : ((cg == 16) ? 9 : 5), nb = C / 32;
    std::vector<bf16_t> img((size_t)nb * 8 * nk * 2 * 32 * 8, 0);
    for (int blk = 0; blk < nb; ++blk)
        for (int w = 0; w < 8; ++w)
            for (int rho = 0; rho < 32; ++rho) {
                const int c4 = 2 * ((rho >> 2) & 1) + (rho >> 4), sm = 4 * ((rho >> 3) & 1) + (rho & 3);
                const int c = 32 * blk + 4 * w + c4, o = 8 * c + sm, g = c / cg;
                for (int j = 0; j < nk; ++j)
                    for (int hk = 0; hk < 2; ++hk)
                        for (int e = 0; e < 8; ++e) {
                            const int tap = (cg == 32) ? (j >> 1) : ((cg == 16) ? j : 2 * j + hk);
                            const int ci = (cg == 32) ? 16 * (j & 1) + 8 * hk + e : ((cg == 16) ? 8 * hk + e : e);
                            if (tap > 8) continue;
                            img[(((((size_t)blk * 8 + w) * nk + j) * 2 + hk) * 32 + rho) * 8 + e] = f2bf(wsp[((size_t)o * cg + ci) * 9 + tap] * gamma[g * cg + ci]);
                        }
            }
    return img;
}

// A fragments of akgm_ws64_kernel (akgm_ws64.hip.h; 64 channels per group, C = 512): [half group hg 16][wave 8][k step j 36][lane half hk][32 rows][8]
// bf16.  Row rho of wave w: feature 32 hg + 4 w + 2 ((rho >> 2) & 1) + (rho >> 4), kernel set 4 ((rho >> 3) & 1) + (rho & 3) - a lane's 16
// accumulators are then 2 features x 8 sets (as pack_akgm_ws32); k step j: tap j / 4, channels 16 (j & 3) + 8 hk + e of the group.
// Values are W * gamma rounded exactly as in pack_akgm (same fold tables).
static inline std::vector<bf16_t> pack_akgm_ws64(const float* wsp, const float* gamma, int C) {
    const int cg = C / 8, nk = 36, nh = C / 32;
    std::vector<bf16_t> img((size_t)nh * 8 * nk * 2 * 32 * 8, 0);
    for (int hg = 0; hg < nh; ++hg)
        for (int w = 0; w < 8; ++w)
            for (int rho = 0; rho < 32; ++rho) {
                const int c4 = 2 * ((rho >> 2) & 1) + (rho >> 4), sm = 4 * ((rho >> 3) & 1) + (rho & 3);
                const int c = 32 * hg + 4 * w + c4, o = 8 * c + sm, g = c / cg;
                for (int j = 0; j < nk; ++j)
                    for (int hk = 0; hk < 2; ++hk)
                        for (int e = 0; e < 8; ++e) {
                            const int tap = j >> 2, ci = 16 * (j & 3) + 8 * hk + e;
                            img[(((((size_t)hg * 8 + w) * nk + j) * 2 + hk) * 32 + rho) * 8 + e] = f2bf(wsp[((size_t)o * cg + ci) * 9 + tap] * gamma[g * cg + ci]);
                        }
            }
    return img;
}

// A fragments of stem_mfma_kernel (misc.hip.h): [C0/64 blocks][tm 2][k16 step j 5][lane 64][8] bf16; lane = (k half
// hh = lane >> 5, row = lane & 31): tap 2j + hh, channel slot e (e >= cin zero).  The tenth tap slot (j = 4, hh = 1)
// carries the BIAS as bf16 hi + lo parts in slots 0 and 1 (the kernel feeds it the constant (1, 1, 0, ...)), so the
// epilogue has no bias to load or add and the sum is exact to 2^-17.
// t: [9 * cin][C0] fp32 with k = tap * cin + ci; bias: [C0].
static inline std::vector<bf16_t> pack_stem_frags(const std::vector<float>& t, const std::vector<float>& bias, int cin, int C0) {
    const int nb = C0 / 64;
    std::vector<bf16_t> f((size_t)nb * 2 * 5 * 64 * 8, 0);
    for (int cb = 0; cb < nb; ++cb)
        for (int tm = 0; tm < 2; ++tm)
            for (int j = 0; j < 5; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = cb * 64 + tm * 32 + (lane & 31), tap = 2 * j + (lane >> 5);
                    bf16_t* d = &f[((((size_t)cb * 2 + tm) * 5 + j) * 64 + lane) * 8];
                    if (tap > 8) {
                        const bf16_t hi = f2bf(bias[row]);
                        d[0] = hi; d[1] = f2bf(bias[row] - bf2f(hi));
                        continue;
                    }
                    for (int e = 0; e < cin; ++e) d[e] = f2bf(t[(size_t)(tap * cin + e) * C0 + row]);
                }
    return f;
}

// A fragments of final_conv_kernel (misc.hip.h): [step = tap * (C/32) + c32][lane][8] bf16 for v_mfma_f32_16x16x32_bf16,
// lane = (k group g = lane >> 4, row = lane & 15): W[row][c32*32 + g*8 + e][tap], rows >= cout zero.  w: [cout][C][3][3]
static inline std::vector<bf16_t> pack_final_frags(const float* w, int cout, int C) {
    const int nc = C / 32;
    std::vector<bf16_t> f((size_t)9 * nc * 64 * 8, 0);
    for (int tap = 0; tap < 9; ++tap)
        for (int c32 = 0; c32 < nc; ++c32)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = lane & 15, g = lane >> 4;
                if (row >= cout) continue;
                for (int e = 0; e < 8; ++e)
                    f[(((size_t)tap * nc + c32) * 64 + lane) * 8 + e] = f2bf(w[((size_t)row * C + c32 * 32 + g * 8 + e) * 9 + tap]);
            }
    return f;
}

// Upsample(nearest x2) + conv3x3 as four parity classes of 2x2 convolutions on the low-res grid:
//   out[2y+py][2x+px] = sum_{dy,dx in {0,1}} Wp[py][px][dy][dx] . in[y+py+dy-1][x+px+dx-1]
// with Wp = sums of the original taps that land on the same source pixel:
//   parity 0: dy=0 <- {k=0}, dy=1 <- {k=1,2};   parity 1: dy=0 <- {k=0,1}, dy=1 <- {k=2}.
// Layout [4 parities][rows_pad][4*Cin], k = (dy*2+dx)*Cin + c.  (reference: model/ucdir.py:53-60)
static inline PackedConv pack_upconv(const float* w, const float* bias, int cout, int cin, int TM) {
    PackedConv P;
    P.ntaps = 4; P.cin = cin; P.cout = cout; P.ncls = 1;
    P.Kpad = 4 * cin;
    P.rows_pad = ((cout + TM - 1) / TM) * TM;
    P.A.assign((size_t)4 * P.rows_pad * P.Kpad, 0);
    P.bias.assign(P.rows_pad, 0.f);
    if (bias) for (int o = 0; o < cout; ++o) P.bias[o] = bias[o];
    auto in_set = [](int par, int d, int k) { return par == 0 ? (d == 0 ? k == 0 : k >= 1) : (d == 0 ? k <= 1 : k == 2); };
    for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px)
        for (int o = 0; o < cout; ++o) for (int c = 0; c < cin; ++c)
            for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
                double sum = 0;
                for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx)
                    if (in_set(py, dy, ky) && in_set(px, dx, kx)) sum += w[((size_t)o * cin + c) * 9 + ky * 3 + kx];
                P.A[((size_t)(py * 2 + px) * P.rows_pad + o) * P.Kpad + (size_t)(dy * 2 + dx) * cin + c] = f2bf((float)sum);
            }
    return P;
}
