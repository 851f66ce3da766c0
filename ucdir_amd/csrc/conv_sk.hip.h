// 3x3 stride-1 convolution (and the parity classes of Upsample + conv3x3) as a PERSISTENT stream-K kernel on 256-row tiles (gfx950).
//
// conv3x3_halo_kernel<128> (128 rows x 256 pixels per workgroup, 64 x 64 wave tiles, two workgroups per CU) sits at the ceiling of
// its structure: one LDS fragment read per MFMA, 16 MFMAs between two barriers, a prologue / epilogue bubble per workgroup and whole
// rounds of workgroups (round-3 verdict: 0.32 of the MFMA peak for 38 % of the forward).  This kernel is the structural step:
//   * ONE workgroup (8 wave64) per CU walks work units; a unit = MW x 128 output rows x NPX = 512 / MW pixel positions
//     (MW = 2: 256 x 256, C_out multiples of 256; MW = 1: 128 x 512, C_out = 128); a wave owns 128 rows x 64 pixels = 4 x 2 MFMA
//     tiles of 32 x 32 x 16 - 128 accumulator registers, SIX fragment reads per EIGHT MFMAs (0.75 instead of 1.0);
//   * pixel tiles are LINEAR in the zero-bordered [B][H + 2][W + 2] position space, batch-flattened: a tile is NPX consecutive
//     positions (crossing rows and samples), its halo the NPX + 2 Wp + 2 positions around them - one contiguous range of the
//     tensor.  Tap (ky, kx) of slot s is halo position s + ky Wp + kx: a uniform shift, 16 consecutive lanes always read 16
//     consecutive halo positions (conflict-free ds_read_b128 at any width), border positions are computed and not stored
//     (utilisation H W / ((H + 2)(W + 2)): 95 % at 72^2, 90 % at 36^2, 81 % at 18^2 - the 2-D tiles of conv3x3_halo reach 84 / 84 /
//     63 %); per-sample GroupNorm statistics enter in the EPILOGUE (rstd, mean * rstd per lane), so a tile may span samples;
//   * K advances one tap x 32 channels per sub-step (16 MFMAs per wave, two k16 units); the weights are packed on the host as the
//     LDS image of every sub-step (16 KB, fragment-major: a fragment read is 1 KB lane-linear, conflict-free, and every LDS-DMA
//     piece is 1 KB of contiguous memory) and stream through a FOUR-deep ring: the stage of sub-step h + 4 is requested at sub-step
//     h and waited for (counted vmcnt) at h + 2, one sub-step before its first read - fragment reads run two units ahead ACROSS the
//     one barrier per sub-step, which only protects the ring slot being refilled;
//   * the halo of the next 32-channel chunk arrives under the current chunk (double buffer, 64-byte pixels, 16-byte chunk
//     XOR (position >> 2) & 3 on the DMA's source side);
//   * stream-K: the units that do not fill a whole round of workgroups are cut into chunk ranges dealt evenly to ALL workgroups;
//     a partial unit leaves raw fp32 accumulators (accumulator layout, coalesced 16-byte stores) in a scratch slot and
//     conv_sk_finish_kernel sums the parts in ascending workgroup order (fixed: bit-reproducible) and runs the same epilogue;
//   * epilogue in registers: rows are permuted at pack time so that a lane holds 16 consecutive channels of one pixel per MFMA tile
//     (32-byte pieces); GroupNorm fold from two sample-independent tables in LDS (bias + Tb, Tg; 9 border classes) and the
//     per-sample (rstd, mean * rstd) list built once per workgroup; output statistics as 2^-20 fixed-point integers.
// What ships as the default (measured): the SAME wave tile and loop in 4-wave workgroups of 128 rows x 256 positions with
// 80 KB of LDS, two per CU, one unit each (template <1, 4, NTAPS>): one's prologue / epilogue runs under the other's K loop, the fold tables
// arrive by LDS-DMA under the last chunk, the block's 1x1 res_conv rides as the grid's last workgroups (sk_alt_unit), wide images are cut
// into vertical strips.  Launches with few long units (the 18^2 level at B = 16) cut every unit's K range in two; the partial tiles are
// summed in part order by conv_sk_finish_kernel (one wave per 32-row fragment).  The 8-wave persistent stream-K kind above is the tested
// alternative (ucdir_debug_flag("convsk", 1)).
// Reference: model/ucdir.py:110 (block conv1), :57 (Upsample conv), SURVEY.md Appendix A.
#pragma once
#include "conv_halo.hip.h"
#include "akgm_ws.hip.h"

struct ConvSkP {
    const bf16_t* A; long long a_par_stride;          // stage images [parity][row tile][chunk][tap][MW * 8 KB]; elements per parity class
    const bf16_t* B0; const bf16_t* B1; int c0, ld0, ld1;   // input(s), zero-bordered NHWC; channels [0, c0) from B0, the rest from B1
    int nchunks;                                      // 32-channel chunks = C_in / 32
    int nb, H, W, Wp, Hp;                             // input grid (= output grid; low-res grid for the Upsample parity classes)
    // position space: [sample][strip][HR][Wpe] - the image is cut into ns vertical strips of Ws columns, each carried with its two neighbour
    // columns; npos = nb * ns * HpWpe positions (HpWpe = HR * Wpe), tiles are NPX consecutive ones.  Classic: HR = H + 2, Wpe = Ws + 2 (ns = 1:
    // the zero-bordered tensor itself).  Round 6, SHARED BORDERS (default): HR = H + 1 - a sample's bottom border row IS the next (sample,
    // strip)'s top border row (both zeros: row yp = 0 of the space, any tensor's padded row 0) - and with a single strip Wpe = W + 1: a row's
    // right border pixel IS the next row's left border pixel (padded column 0).  Taps stay uniform shifts of ky Wpe + kx; the decode below is
    // unchanged (yp = 0 .. H, xs = 0 .. Wpe - 1 are padded coordinates).  Computed-and-dropped positions: 36^2 10.3 -> 5.3 %, 18^2 19 -> 10.2 %.
    int ns, Ws, Wpe, HpWpe, npos;
    float inv_per_b, inv_HpWpe, inv_Wpe;               // reciprocals for sk_udiv (1 / (ns * HpWpe), 1 / HpWpe, 1 / Wpe)
    int ntiles, rowtiles, npar;                       // pixel tiles, row tiles, parity classes (1 | 4); units = npar * rowtiles * ntiles
    int nhp;                                          // halo DMA pieces (16 positions each) per chunk
    int nfeat;
    float alpha; int fold; int act;
    const stat_t* stats0; const stat_t* stats1; double inv_count;
    const float* bias; const float* Tb; const float* Tg; int tab_ld;
    const bf16_t* res; int res_ld;
    bf16_t* out; int out_ld;
    stat_t* stats_out;
    // schedule: units [0, ndp) whole, round-robin over the grid; units [ndp, units) cut into chunk ranges over all workgroups
    int units, ndp;
    float* partial;                                   // [2 * grid][MW * 32768] fp32: slot 2 g = workgroup g's first partial segment, 2 g + 1 its last
    // the block's 1x1 res_conv on the same input (4-wave kind, one workgroup per unit): the LAST alt_units workgroups of the grid compute
    // out2 = res_conv(x) + bias2, unit (row tile, pixel tile) - dispatched last, they fill the workgroup slots the 3x3 conv's last,
    // partly empty round leaves idle instead of re-reading x in a launch of their own
    int alt_units; const bf16_t* alt_A; const float* alt_bias; bf16_t* alt_out; int alt_out_ld;
    // mixed launch (conv_sk_kernel<1, 4, 9>, mix_nunits > 0): units (= ndp) wide units over the first mix_wt pixel tiles, then mix_nunits (= mix_nndp) SHORT units (64 rows: two per
    // 128-row tile) over the mix_nt pixel tiles from position mix_q0 = 256 mix_wt on
    // (mix_srt: short units per pixel tile = 2 x rowtiles)
    int mix_wt, mix_nt, mix_nhp, mix_nunits, mix_nndp, mix_q0, mix_srt;
    unsigned long long* dbg;
};

// MW row blocks of 128 x NW wave64 per workgroup: <2, 8> 256 rows x 256 positions and <1, 8> 128 x 512, one persistent workgroup per CU
// (160 KB of LDS); <1, 4> 128 x 256 with 80 KB, TWO workgroups per CU: one's prologue / epilogue runs under the other's K loop
template <int MW, int NW = 8>
struct CvSk {
    static constexpr int NWN = NW / MW;               // waves along the positions
    static constexpr int NPX = 64 * NWN;              // pixel positions per unit
    static constexpr int ROWS = 128 * MW;
    static constexpr int STAGE = 8192 * MW;           // bytes of one sub-step's weights
    static constexpr int PW = 8 * MW / NW;            // DMA pieces per wave and stage
    static constexpr bool LDS_TAB = NW == 8;          // fold tables in LDS (persistent workgroups) or read from global memory in the epilogue
    static constexpr int OFF_W = 0;
    static constexpr int OFF_TB = 4 * STAGE;          // [9][ROWS] fp32: bias + Tb[cls]
    static constexpr int OFF_TG = OFF_TB + (LDS_TAB ? 9 * ROWS * 4 : 0);
    static constexpr int OFF_MS = OFF_TG + (LDS_TAB ? 9 * ROWS * 4 : 0);     // [64][2] fp32: (alpha * rstd, mean * rstd) per sample
    static constexpr int MAXB = 64;
    static constexpr int OFF_H = OFF_MS + MAXB * 8;   // two halo buffers of nhp KB each
    static constexpr int THREADS = 64 * NW;
    static constexpr int NHW = NW == 4 ? 6 : (MW == 2 ? 4 : 7);   // halo DMA pieces per wave and chunk (a fixed number: the counted waits are literals);
    static constexpr int NHP_MAX = NW * NHW;          // pieces past the halo's end repeat its last piece
    static constexpr int LDS_MAX = NW == 8 ? 160 * 1024 : 80 * 1024;
    __host__ __device__ static constexpr int lds_bytes(int nhp) { return OFF_H + 2 * nhp * 1024; }
    __host__ __device__ static constexpr int part_floats() { return ROWS * NPX; }
};

#ifdef UCDIR_TIMING
#define SK_STAMP() do { if (dbg_on && dbg_n < 120) p.dbg[dbg_off + dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SK_STAMP() do {} while (0)
#endif

// one LDS-DMA piece with a wave-uniform (SGPR) base and a 32-bit per-lane byte offset: no 64-bit VALU address arithmetic per piece (the
// builtin always takes a per-lane 64-bit pointer).  hipcc does not count it (cdna guide 5.7): the kernel's waits are counted by hand anyway;
// M0 (the LDS destination) is written inside the statement.
__device__ __forceinline__ void sk_dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void sk_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// segment list of workgroup g: its whole units (round-robin) and its chunk range of the stream-K part
struct SkSched {
    int units, ndp, nch, G;
    long long tc;                                      // chunks of the stream-K part
    __host__ __device__ SkSched(int units_, int ndp_, int nch_, int G_) : units(units_), ndp(ndp_), nch(nch_), G(G_), tc((long long)(units_ - ndp_) * nch_) {}
    __host__ __device__ long long start(int g) const { return tc * g / G; }
};

// position q of the strip space -> sample, padded row, column inside the strip (0 and Ws + 1: the neighbour columns) and padded column
// of the tensor
// Exact q / d for 0 <= q < 2^31 and quotients below 2^22 (samples, strips, rows): the float product is within one of the quotient, two
// integer fix-ups make it exact - ~10 instructions instead of the ~35 of a 32-bit integer division (eight decodes of three divisions per unit
// were ~3 k cycles of a unit's prologue and epilogue)
__device__ __forceinline__ int sk_udiv(int q, int d, float inv_d) {
    int t = (int)((float)q * inv_d);
    t = t * d > q ? t - 1 : t;
    t = (t + 1) * d <= q ? t + 1 : t;
    return t;
}
__device__ __forceinline__ void sk_decode(const ConvSkP& p, int q, int& b, int& yp, int& xs, int& xpm) {
    const int per_b = p.ns * p.HpWpe;
    b = sk_udiv(q, per_b, p.inv_per_b);
    int r = q - b * per_b;
    int strip = 0;
    if (p.ns > 1) { strip = sk_udiv(r, p.HpWpe, p.inv_HpWpe); r -= strip * p.HpWpe; }
    yp = sk_udiv(r, p.Wpe, p.inv_Wpe); xs = r - yp * p.Wpe;
    xpm = strip * p.Ws + xs;
}

// LDS reads the compiler does not count (an epilogue that runs while the next segment's LDS-DMAs are in flight must not be made to
// wait for them: hipcc puts vmcnt(0) in front of every LDS read it knows about while an LDS-DMA is outstanding)
__device__ __forceinline__ void lds_read8f_asm(f32x2_t& v, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
}

// 64-bit wave sum on the DPP network instead of six dependent ds_bpermute pairs (~700 cycles each in the epilogue): prefix sums inside
// the rows of 16 (row_shr 1 / 2 / 4 / 8), row totals into the odd rows (row_bcast:15, rows 1 and 3), the lower half's total into the
// upper half (row_bcast:31, rows 2 and 3); lane 63 holds the wave's sum.  Integer adds: same bits as any other order.
__device__ __forceinline__ stat_t wave_sum_ll_dpp(stat_t v) {
    auto step = [&](auto ctrl, auto rmask) {
        constexpr int C = decltype(ctrl)::value, RM = decltype(rmask)::value;
        const unsigned lo = (unsigned)(v & 0xffffffffll), hi = (unsigned)((unsigned long long)v >> 32);
        const unsigned tlo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, C, RM, 0xf, true);
        const unsigned thi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, C, RM, 0xf, true);
        v += (stat_t)(((unsigned long long)thi << 32) | tlo);
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 -> rows 1, 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 -> rows 2, 3
    const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffll), 63);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), 63);
    return (stat_t)(((unsigned long long)rhi << 32) | rlo);
}

// epilogue of one wave's 128 x 64 tile from its accumulators: GroupNorm fold, activation, residual, statistics, bf16 NHWC store.
// Shared by the conv kernel (LDS_TAB: fold tables and per-sample scalars in LDS, read by uncounted inline asm) and the finish kernel
// (tables straight from bias / Tb / Tg in global memory, scalars in its own LDS): identical arithmetic - a unit finished in-kernel
// or from partial tiles gives the same bits for the same accumulator values.
template <int MW, int NW, int TAB, bool MS_ASM, int NF = 4>
__device__ __forceinline__ void sk_epilogue(const ConvSkP& p, const unsigned char* smem, const float* ms_plain, f32x16_t (&acc)[NF][2],
                                            int par, int rt, int q0, int wm, int wn, int lane, unsigned tab_base = 0, int fsel = -1, int roff = 0) {
    using L = CvSk<MW, NW>;
    const int hh = lane >> 5, l31 = lane & 31;
    const int act = p.act;
    const int py = par >> 1, pxp = par & 1;
    int sb_n[2] = {0, 0};
    stat_t sf1[2] = {0, 0}, sf2[2] = {0, 0};
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = q0 + wn * 64 + n * 32 + l31;                     // position in the strip space (q0: the unit's first position)
        int b, yp, xs, xp;
        sk_decode(p, q, b, yp, xs, xp);
        const bool valid = q < p.npos && yp >= 1 && yp <= p.H && xs >= 1 && xs <= p.Ws && xp <= p.W;
        const int bb = b < p.nb ? b : p.nb - 1;
        float ra, mr;
        if (MS_ASM) {
            f32x2_t m2;
            lds_read8f_asm(m2, (unsigned)(L::OFF_MS + 8 * bb));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(m2));                             // (the value is only touched behind the wait: cdna guide 5.7 form iii)
            __builtin_amdgcn_sched_barrier(0);
            ra = m2[0]; mr = m2[1];
        } else { ra = ms_plain[2 * bb]; mr = ms_plain[2 * bb + 1]; }
        const int y = yp - 1, x = xp - 1;
        const int cls = p.npar > 1 ? 4 : (y <= 0 ? 0 : (y >= p.H - 1 ? 2 : 1)) * 3 + (x <= 0 ? 0 : (x >= p.W - 1 ? 2 : 1));
        long long opos = ((long long)b * p.Hp + yp) * p.Wp + xp;
        if (p.npar > 1) opos = ((long long)b * (2 * p.H + 2) + (2 * y + py + 1)) * (2 * p.W + 2) + (2 * x + pxp + 1);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (fsel >= 0 && f != fsel) continue;                      // (finish kernel: one 32-row fragment per workgroup)
            const int ch = wm * 128 + roff + f * 32 + 16 * hh;         // channel within the 128-row tile (roff: a SHORT unit's half, 0 | 64)
            const int fo = rt * L::ROWS + ch;
            f32x4_t b4[4], g4v[4];
            if (TAB == 2) {
                // segments of 128 floats at tab_base: bias | Tb[0..8] | Tg[0..8] of this row tile (DMA'd into the idle halo buffer under the last chunk)
                const unsigned ba = tab_base + (unsigned)(ch * 4), ta = ba + (unsigned)((1 + cls) * 512), ga = ba + (unsigned)((10 + cls) * 512);
                f32x4_t bb4[4];
                lds_read16f_asm<0>(bb4[0], ba); lds_read16f_asm<16>(bb4[1], ba); lds_read16f_asm<32>(bb4[2], ba); lds_read16f_asm<48>(bb4[3], ba);
                if (p.fold) {
                    lds_read16f_asm<0>(b4[0], ta); lds_read16f_asm<16>(b4[1], ta); lds_read16f_asm<32>(b4[2], ta); lds_read16f_asm<48>(b4[3], ta);
                    lds_read16f_asm<0>(g4v[0], ga); lds_read16f_asm<16>(g4v[1], ga); lds_read16f_asm<32>(g4v[2], ga); lds_read16f_asm<48>(g4v[3], ga);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(bb4[g4]));
                if (p.fold) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(b4[g4]), "+v"(g4v[g4]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bv = p.bias ? bb4[g4][e] : 0.f;
                        b4[g4][e] = p.fold ? bv + b4[g4][e] : bv;
                        g4v[g4][e] = p.fold ? g4v[g4][e] : 0.f;
                    }
            } else if (TAB == 1) {
                const unsigned ta = (unsigned)(L::OFF_TB + (cls * L::ROWS + ch) * 4), ga = ta + (unsigned)(L::OFF_TG - L::OFF_TB);
                lds_read16f_asm<0>(b4[0], ta); lds_read16f_asm<16>(b4[1], ta); lds_read16f_asm<32>(b4[2], ta); lds_read16f_asm<48>(b4[3], ta);
                lds_read16f_asm<0>(g4v[0], ga); lds_read16f_asm<16>(g4v[1], ga); lds_read16f_asm<32>(g4v[2], ga); lds_read16f_asm<48>(g4v[3], ga);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) asm volatile("" : "+v"(b4[g4]), "+v"(g4v[g4]));
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int fg = fo + 4 * g4;
                    f32x4_t vb = {0.f, 0.f, 0.f, 0.f}, vg = {0.f, 0.f, 0.f, 0.f};
                    if (fg < p.nfeat) {
                        if (p.bias) vb = *reinterpret_cast<const f32x4_t*>(p.bias + fg);
                        if (p.fold) {
                            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p.Tb + (long long)cls * p.tab_ld + fg);
                            vb[0] += t[0]; vb[1] += t[1]; vb[2] += t[2]; vb[3] += t[3];
                            vg = *reinterpret_cast<const f32x4_t*>(p.Tg + (long long)cls * p.tab_ld + fg);
                        }
                    }
                    b4[g4] = vb; g4v[g4] = vg;
                }
            }
            float v[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g4 + e] = fmaf(acc[f][n][4 * g4 + e], ra, fmaf(-mr, g4v[g4][e], b4[g4][e]));
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = silu_fast(v[i]);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaxf(0.2f * v[i], v[i]);
            }
            if (valid && fo < p.nfeat) {
                if (p.res) {
                    const uint4 r0 = *reinterpret_cast<const uint4*>(p.res + opos * p.res_ld + fo);
                    const uint4 r1 = *reinterpret_cast<const uint4*>(p.res + opos * p.res_ld + fo + 8);
                    const bf16_t* h0 = reinterpret_cast<const bf16_t*>(&r0);
                    const bf16_t* h1 = reinterpret_cast<const bf16_t*>(&r1);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { v[i] += bf2f(h0[i]); v[8 + i] += bf2f(h1[i]); }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
                bf16_t* op = p.out + opos * p.out_ld + fo;
                *reinterpret_cast<uint4*>(op) = pack8_bf16(v);
                *reinterpret_cast<uint4*>(op + 8) = pack8_bf16(v + 8);
            }
        }
        if (p.stats_out) { sb_n[n] = b; sf1[n] = valid ? stat_fx((double)s1) : 0; sf2[n] = valid ? stat_fx((double)s2) : 0; }
    }
    if (p.stats_out) {
        // per-sample sums of the wave's 64 positions (both MFMA tiles in ONE pair of wave reductions: the 64-bit shuffles are a dependent
        // chain of ~700 cycles each): they usually belong to one sample; loop over the few they can span.  Integer sums: any grouping
        // gives the same bits.
        const int b_lo = __builtin_amdgcn_readlane(sb_n[0], 0), b_hi0 = __builtin_amdgcn_readlane(sb_n[1], 31);
        const int b_hi = b_hi0 < p.nb ? b_hi0 : p.nb - 1;
        for (int sb = b_lo; sb <= b_hi; ++sb) {
            const stat_t a = wave_sum_ll_dpp((sb_n[0] == sb ? sf1[0] : 0) + (sb_n[1] == sb ? sf1[1] : 0));
            const stat_t q2 = wave_sum_ll_dpp((sb_n[0] == sb ? sf2[0] : 0) + (sb_n[1] == sb ? sf2[1] : 0));
            if (lane == 0 && (a != 0 || q2 != 0)) stat_add_fx(p.stats_out, sb, a, q2);
        }
    }
}

// (alpha * rstd, mean * rstd) of samples [b0, b1) into ms[2 (b - b0) ..]; wave `wave` of `nwaves` takes every nwaves-th sample
__device__ __forceinline__ void sk_sample_table(const ConvSkP& p, float* ms, int b0, int b1, int wave, int nwaves, int lane) {
    for (int b = b0 + wave; b < b1; b += nwaves) {
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
        if (lane == 0) { ms[2 * (b - b0)] = p.alpha * rstd; ms[2 * (b - b0) + 1] = p.fold ? mean * rstd : 0.f; }
    }
}
// (bias + Tb, Tg) of row tile rt, all nine border classes
template <int MW, int NW>
__device__ __forceinline__ void sk_row_table(const ConvSkP& p, unsigned char* smem, int rt, int tid) {
    using L = CvSk<MW, NW>;
    float* tb = reinterpret_cast<float*>(smem + L::OFF_TB);
    float* tg = reinterpret_cast<float*>(smem + L::OFF_TG);
    for (int i = tid; i < 9 * L::ROWS; i += L::THREADS) {
        const int cls = i / L::ROWS, fl = i - cls * L::ROWS, f = rt * L::ROWS + fl;
        float vb = 0.f, vg = 0.f;
        if (f < p.nfeat) {
            vb = p.bias ? p.bias[f] : 0.f;
            if (p.fold) { vb += p.Tb[(long long)cls * p.tab_ld + f]; vg = p.Tg[(long long)cls * p.tab_ld + f]; }
        }
        tb[i] = vb; tg[i] = vg;
    }
}


// One unit of the block's 1x1 res_conv (K = C_in, no halo): a two-deep pipeline per 32-channel chunk - weights (8 KB) into the ring's
// slots, the tile's 256 positions (16 KB) into the halo area - with a full wait per chunk: the pass is bound by the LDS fill (24 KB per
// 16 MFMAs of a wave), not by the matrix cores, which the CU's other workgroup is using meanwhile.
template <int NW>
__device__ __forceinline__ void sk_alt_unit(const ConvSkP& p, unsigned char* smem, int unit, int wave, int lane) {
    using L = CvSk<1, NW>;
    constexpr int NPC = L::NPX / 16 / NW;                           // position pieces per wave and chunk (4)
    const int l31 = lane & 31, hh = lane >> 5, wn = wave;
    const int nch = p.nchunks;
    const int tile = unit / p.rowtiles, rt = unit - tile * p.rowtiles;     // row tile fastest, as the main units
    const int q0 = tile * L::NPX;
    unsigned hq[NPC], hsw[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int px = 16 * (j * NW + wave) + (lane >> 2);
        int q = q0 + px; q = q < p.npos ? q : p.npos - 1;
        int b, yp, xs, xp;
        sk_decode(p, q, b, yp, xs, xp);
        xp = xp < p.Wp ? xp : p.Wp - 1;
        hq[j] = (unsigned)((b * p.Hp + yp) * p.Wp + xp);
        hsw[j] = (unsigned)(((lane & 3) ^ ((px >> 2) & 3)) << 3);
    }
    const bf16_t* const a_src = p.alt_A + ((long long)rt * nch) * (L::STAGE / 2) + (wave * L::PW) * 512 + lane * 8;
    auto issue = [&](int c) {
        int ch = c * 32;
        const bf16_t* src; int ld;
        if (ch < p.c0) { src = p.B0; ld = p.ld0; } else { src = p.B1; ld = p.ld1; ch -= p.c0; }
#pragma unroll
        for (int j = 0; j < NPC; ++j)
            stage16(src + (hq[j] * (unsigned)ld + hsw[j] + (unsigned)ch), smem + L::OFF_H + (c & 1) * (L::NPX * 64) + (j * NW + wave) * 1024, lane);
#pragma unroll
        for (int j = 0; j < L::PW; ++j)
            stage16(a_src + (long long)c * (L::STAGE / 2) + j * 512, smem + L::OFF_W + (c & 1) * L::STAGE + (wave * L::PW + j) * 1024, lane);
    };
    unsigned bx[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int px = wn * 64 + n * 32 + l31;
        bx[n] = L::OFF_H + (px << 6) + (((hh ^ (px >> 2)) & 3) << 4);
    }
    const unsigned a_lane = L::OFF_W + lane * 16;
    f32x16_t acc[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][n][e] = 0.f;
    issue(0);
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // chunk c is in LDS; nobody reads chunk c - 1 any more
        if (c + 1 < nch) issue(c + 1);
        bf16x8_t fa[2][4], fb[2][2];
        const unsigned aa = a_lane + (unsigned)((c & 1) * L::STAGE), bo = (unsigned)((c & 1) * (L::NPX * 64));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (jj == 0) {
                lds_read16_asm<0>(fa[0][0], aa); lds_read16_asm<2048>(fa[0][1], aa); lds_read16_asm<4096>(fa[0][2], aa); lds_read16_asm<6144>(fa[0][3], aa);
            } else {
                lds_read16_asm<1024>(fa[1][0], aa); lds_read16_asm<3072>(fa[1][1], aa); lds_read16_asm<5120>(fa[1][2], aa); lds_read16_asm<7168>(fa[1][3], aa);
            }
            lds_read16_asm<0>(fb[jj][0], (bx[0] + bo) ^ (jj << 5));
            lds_read16_asm<0>(fb[jj][1], (bx[1] + bo) ^ (jj << 5));
        }
        lgkm_wait_asm<6>();
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][f], fb[0][n], acc[f][n], 0, 0, 0);
        lgkm_wait_asm<0>();
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][f], fb[1][n], acc[f][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // out2 = acc + bias2: bf16 NHWC, valid positions only (no GroupNorm fold, no activation, no statistics)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = q0 + wn * 64 + n * 32 + l31;
        int b, yp, xs, xp;
        sk_decode(p, q, b, yp, xs, xp);
        const bool valid = q < p.npos && yp >= 1 && yp <= p.H && xs >= 1 && xs <= p.Ws && xp <= p.W;
        const long long opos = ((long long)b * p.Hp + yp) * p.Wp + xp;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int fo = rt * L::ROWS + f * 32 + 16 * hh;
            if (!valid || fo >= p.nfeat) continue;
            float v[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
                if (p.alt_bias) bv = *reinterpret_cast<const f32x4_t*>(p.alt_bias + fo + 4 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g4 + e] = acc[f][n][4 * g4 + e] + bv[e];
            }
            bf16_t* op = p.alt_out + opos * p.alt_out_ld + fo;
            *reinterpret_cast<uint4*>(op) = pack8_bf16(v);
            *reinterpret_cast<uint4*>(op + 8) = pack8_bf16(v + 8);
        }
    }
}

// The schedule of one workgroup: `lid` of `G` workgroups over `units` units (the first `ndp` whole, round-robin; the rest as stream-K chunk
// ranges), units laid over the position space from position `qbase` on in tiles of NPX, `nhp` halo pieces per chunk.  conv_sk_kernel passes the
// launch's own figures; a mixed launch (below) runs wide units and SHORT units (NF = 2) side by side.
// NF = 4: the wave tile is 128 rows x 64 positions; NF = 2: SHORT units - 64 rows (one half of a 128-row tile: the upper or lower 4 KB of every 8 KB
// stage of the packed image, the parent tile's fold tables) x the same positions, wave tile 64 x 64: half the matrix work per wave at the same halo,
// the quantum a mixed launch balances the SIMDs with.
template <int MW, int NW, int NTAPS, int NF = 4>
__device__ __forceinline__ void conv_sk_body(const ConvSkP& p, unsigned char* smem, const int lid, const int G, const int units_arg, const int ndp_arg,
                                             const int qbase, const int nhp) {
    using L = CvSk<MW, NW>;
    static_assert(NF == 4 || (NF == 2 && MW == 1 && NW == 4), "short units: four waves of 64 x 64");
    constexpr int PWN = L::PW * NF / 4;                              // DMA pieces per wave and stage
    constexpr int RSUB = 4 / NF;                                     // units per 128-row tile
    constexpr int NWN = L::NWN;                                      // waves along the pixel dimension
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
#ifdef UCDIR_TIMING
    // two stamped workgroups: one of the first round (entries 0 ..), one of the last (entries 128 ..)
    const bool dbg_on = p.dbg && (lid == G / 2 + 3 || lid == G - 3) && (lane == 0) && (wave == NW - 1);
    const int dbg_off = lid == G - 3 ? 128 : 0;
    int dbg_n = 0;
#endif
    SK_STAMP();
    const int Wp = p.Wpe, nch = p.nchunks;                           // (Wp: row pitch of the strip space)
    const int HB = nhp * 1024;                                       // bytes of one halo buffer
    constexpr int NHW = L::NHW;                                      // halo pieces per wave and chunk
    constexpr int TA = NTAPS - 3 > 1 ? NTAPS - 3 : 1;                // sub-steps of a chunk whose S point may carry halo pieces: the
                                                                     // next chunk's first fragment reads are issued behind S(NTAPS - 2), whose
                                                                     // wait covers everything requested up to S(NTAPS - 4)
    constexpr int HPER = (NHW + TA - 1) / TA;                        // pieces per such S point

    // ---- per-lane constants of the fragment reads ------------------------------------------------------------------------
    // A: stage slot s, fragment f (32 rows), k16 j: 1 KB lane-linear at s * STAGE + ((4 wm + f) * 2 + j) * 1024 + lane * 16
    // (short units: their 4 KB - fragments 0, 1 of the slot - sit at the slot's start)
    const unsigned a_lane = L::OFF_W + (NF == 4 ? (4 * wm) * 2048 : 0) + lane * 16;
    // B: tap t, MFMA tile n: halo position hp = 64 wn + 32 n + l31 + ky Wp + kx; 64 bytes per position, 16-byte chunk
    // (2 jj + hh) ^ ((hp >> 2) & 3): address(jj = 1) = address(jj = 0) ^ 32.  (Upsample parity classes: per segment.)
    // (the second MFMA tile of a wave is 32 positions = 2048 bytes further with the same swizzle: an immediate offset, not a register)
    unsigned bx[NTAPS];
    auto set_bx = [&](int py, int pxp) {
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            int ky, kx;
            if (NTAPS == 9) { ky = t / 3; kx = t - 3 * ky; } else { ky = py + (t >> 1); kx = pxp + (t & 1); }
            const int hp = wn * 64 + l31 + ky * Wp + kx;
            bx[t] = L::OFF_H + (hp << 6) + (((hh ^ (hp >> 2)) & 3) << 4);
        }
    };
    if (NTAPS == 9) set_bx(0, 0);
    // halo staging: piece i = 16 positions; lane -> (position 16 i + lane / 4, physical 16-byte chunk lane & 3 = logical ^ ((pos >> 2) & 3));
    // this wave's j-th piece is piece j * NW + wave (past the halo's end: a repeat of its last piece)
    // (per lane: one base position and one source swizzle - the chunk XOR ((16 i + lane / 4) >> 2) & 3 = (lane >> 4) & 3 does not depend on the
    // piece; the piece index i of (j, wave) is wave-uniform)
    const int hpos0 = (lane >> 2) - Wp - 1;                         // + 16 i + q0: position in the strip space
    const unsigned hsw = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 3);
    auto piece_of = [&](int j) { const int i = j * NW + wave; return i < nhp ? i : nhp - 1; };

    const SkSched sch(units_arg, ndp_arg, nch, G);
    const long long c_beg = sch.start(lid), c_end = sch.start(lid + 1);      // stream-K chunk range of this workgroup
    const int ndp_mine = lid < ndp_arg ? (ndp_arg - lid + G - 1) / G : 0;    // whole units lid, lid + G, ...
    long long cpos = c_beg;
    int dpi = 0;
    struct Seg { int unit, cb, ce, par, rt, tile; bool first_sk, ok; };
    auto next_segment = [&]() {
        Seg s; s.ok = true; s.first_sk = false;
        if (dpi < ndp_mine) { s.unit = lid + dpi * G; s.cb = 0; s.ce = nch; ++dpi; }
        else if (cpos < c_end) {
            const long long u = cpos / nch;
            s.unit = ndp_arg + (int)u; s.cb = (int)(cpos - u * nch);
            const long long e = (u + 1) * nch < c_end ? (u + 1) * nch : c_end;
            s.ce = (int)(e - u * nch);
            s.first_sk = cpos == c_beg;
            cpos = e;
        } else { s.ok = false; s.unit = 0; s.cb = 0; s.ce = 0; }
        int u = s.unit;
        const int RT = NF == 4 ? p.rowtiles : p.mix_srt;              // units per pixel tile (short: two per 128-row tile)
        const int per_par = RT * p.ntiles;
        s.par = 0;
        if (p.npar > 1) { s.par = u / per_par; u -= s.par * per_par; }
        // row tile fastest: the row tiles of one pixel tile are neighbours in the (XCD-contiguous) unit order, so they run at the same time on
        // one XCD and share its halo in that L2 (row tile slowest re-read every halo C_out / 128 times from the fabric: 2.2 x the algorithmic bytes)
        s.tile = u / RT; s.rt = u - s.tile * RT;                       // (s.rt counts units of 32 NF rows)
        return s;
    };

    // per-segment state of the DMA streams
    unsigned hq[NHW];                                                // this wave's halo pieces: per-lane BYTE offsets in the tensor whose row pitch is ld_cur
    int ld_cur = 0;                                                  // (position * ld_cur + swizzle) * 2; re-scaled when the chunk sequence changes tensor
    const bf16_t* hsrc = nullptr;                                    // wave-uniform source of the chunk whose halo is being requested (tensor + channel)
    auto halo_chunk = [&](int c) {                                   // once per chunk, not per piece
        int ch = c * 32;
        const bool first = ch < p.c0;
        const int ld = first ? p.ld0 : p.ld1;
        hsrc = first ? p.B0 + ch : p.B1 + (ch - p.c0);
        if (ld != ld_cur) {
#pragma unroll
            for (int j = 0; j < NHW; ++j) hq[j] = ((((hq[j] >> 1) - hsw) / (unsigned)ld_cur) * (unsigned)ld + hsw) * 2u;
            ld_cur = ld;
        }
    };
    const unsigned char* a_seg = nullptr;                            // wave-uniform source of this wave's pieces of stage 0 of the segment (+ k * STAGE bytes)
    const unsigned lane16 = (unsigned)lane * 16u;
    int seg_cb = 0;
    auto issue_halo = [&](int j, int buf) {                          // piece j of the chunk of halo_chunk() into halo buffer buf
        sk_dma16(hsrc, hq[j], (unsigned)(L::OFF_H + piece_of(j) * 1024 + buf * HB));
    };
    auto issue_stage = [&](int k, int slot) {                        // stage k of the segment's (parity, row tile); past its end: whatever follows
#pragma unroll                                                       // in the image (the image is padded by four stages) into a slot nobody reads any more
        for (int j = 0; j < PWN; ++j)
            sk_dma16(a_seg + (long long)k * L::STAGE + j * 1024, lane16, (unsigned)(L::OFF_W + slot * L::STAGE + (wave * PWN + j) * 1024));
    };
    // one-shot kind: the fold tables of the segment's row tile (bias | Tb[9] | Tg[9], 128 floats each: 19 half pieces) take the place of the
    // halo pieces that the last chunk would request for a chunk that does not exist - same instruction count, idle halo buffer
    constexpr int NTP = 10;
    auto issue_table = [&](int rt, int i, int buf) {                 // piece i = segments 2 i (lanes 0 - 31) and 2 i + 1 (lanes 32 - 63)
        const int seg = 2 * i + (lane >> 5);
        const float* src = reinterpret_cast<const float*>(p.A);      // (a missing array: anything readable; the epilogue does not look at it)
        if (seg == 0 || seg > 18) { if (p.bias) src = p.bias + rt * L::ROWS; }
        else if (seg < 10) { if (p.fold) src = p.Tb + (long long)(seg - 1) * p.tab_ld + rt * L::ROWS; }
        else if (p.fold) src = p.Tg + (long long)(seg - 10) * p.tab_ld + rt * L::ROWS;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + (lane & 31) * 4), (LDS_AS void*)(smem + L::OFF_H + buf * HB + i * 1024), 16, 0, 0);
    };
    int hb0 = 0;                                                     // halo buffer of the segment's first chunk
    auto prefetch = [&](const Seg& s) {                              // halo of the first chunk and stages 0 .. 3 of segment s
        if (NTAPS != 9) set_bx(s.par >> 1, s.par & 1);
        const int q0 = qbase + s.tile * L::NPX;
#pragma unroll
        for (int j = 0; j < NHW; ++j) {
            int q = q0 + hpos0 + 16 * piece_of(j);
            // in front of the space: position 0 (a border pixel); behind it: the positions of a virtual further sample, read from sample 0 -
            // its row 0 and the first pixel of its row 1 are border zeros, which is all a VALID position's taps can reach (shared-border
            // space: the last position is an interior pixel, a clamp to it would hand its value to the bottom taps of the last row)
            q = q < 0 ? 0 : (q >= p.npos ? (q - p.npos < p.npos ? q - p.npos : 0) : q);
            int b, yp, xs, xp;
            sk_decode(p, q, b, yp, xs, xp);
            xp = xp < p.Wp ? xp : p.Wp - 1;                          // (a ragged last strip)
            hq[j] = (unsigned)((b * p.Hp + yp) * p.Wp + xp);
        }
        ld_cur = s.cb * 32 < p.c0 ? p.ld0 : p.ld1;
#pragma unroll
        for (int j = 0; j < NHW; ++j) hq[j] = (hq[j] * (unsigned)ld_cur + hsw) * 2u;
        halo_chunk(s.cb);
        a_seg = reinterpret_cast<const unsigned char*>(p.A + (long long)s.par * p.a_par_stride + ((long long)(s.rt / RSUB) * nch * NTAPS) * (L::STAGE / 2) +
                                                       (s.rt % RSUB) * (L::STAGE / 2 / RSUB) + (wave * PWN) * 512);
        seg_cb = s.cb;
#pragma unroll
        for (int j = 0; j < NHW; ++j) issue_halo(j, hb0);
#pragma unroll
        for (int k = 0; k < 4; ++k) issue_stage(s.cb * NTAPS + k, k);
    };

    Seg cur = next_segment();
    if (!cur.ok) return;
    int rt_cur = cur.rt;
    prefetch(cur);
    // (alpha * rstd, mean * rstd): persistent workgroups build the whole list once, one-shot ones the few samples their tile spans
    auto sample_range = [&](const Seg& s) {
        const int per_b = p.ns * p.HpWpe;
        const int b0 = sk_udiv(qbase + s.tile * L::NPX, per_b, p.inv_per_b);
        int b1 = sk_udiv(qbase + s.tile * L::NPX + L::NPX - 1, per_b, p.inv_per_b) + 1; b1 = b1 < p.nb ? b1 : p.nb;
        if (b0 < b1) sk_sample_table(p, reinterpret_cast<float*>(smem + L::OFF_MS) + 2 * b0, b0, b1, wave, NW, lane);
    };
    if (L::LDS_TAB) {
        sk_sample_table(p, reinterpret_cast<float*>(smem + L::OFF_MS), 0, p.nb, wave, NW, lane);
        sk_row_table<MW, NW>(p, smem, cur.rt, tid);
    } else sample_range(cur);

#pragma unroll 1
    while (true) {
        const Seg nxt = next_segment();
        const int cb = cur.cb, ce = cur.ce;
        f32x16_t acc[NF][2];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[f][n][e] = 0.f;

        HC_WAIT(0);                                                  // this segment's first halo and stages 0 .. 3 (requested under the previous epilogue)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SK_STAMP();

        bf16x8_t fa[2][NF], fb[2][2];
        // R(t, jj) of sub-step h: A from slot h & 3, B from the halo buffer of its chunk
        auto reads = [&](auto tc, auto jc, int h, unsigned boff) {
            constexpr int t = decltype(tc)::value, jj = decltype(jc)::value;
            const unsigned aa = a_lane + (unsigned)((h & 3) * L::STAGE);
            lds_read16_asm<0 * 2048 + jj * 1024>(fa[jj][0], aa);
            lds_read16_asm<1 * 2048 + jj * 1024>(fa[jj][1], aa);
            if constexpr (NF == 4) {
                lds_read16_asm<2 * 2048 + jj * 1024>(fa[jj][2], aa);
                lds_read16_asm<3 * 2048 + jj * 1024>(fa[jj][3], aa);
            }
            const unsigned b0 = (bx[t] + boff) ^ (jj << 5);
            lds_read16_asm<0>(fb[jj][0], b0);
            lds_read16_asm<2048>(fb[jj][1], b0);
        };
        __builtin_amdgcn_sched_barrier(0);
        unsigned bcur = hb0 ? (unsigned)HB : 0u;                     // halo buffer offset of the current chunk (0 | HB)
        reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, bcur);
        reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 0, bcur);

        int h = 0;                                                   // sub-step counter of this segment
        const bool want_tab = !L::LDS_TAB && cb == 0 && ce == nch;   // (a partial segment has no epilogue)
#pragma unroll 1
        for (int c = cb; c < ce; ++c) {
            const unsigned bnext = bcur ? 0u : (unsigned)HB;
            const int cn = c + 1 < ce ? c + 1 : c;                   // (behind the last chunk: its own halo again, into the OTHER buffer, which nobody reads any more)
            halo_chunk(cn);
            static_for<0, NTAPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int tn = (t + 1) % NTAPS;
                constexpr int tm1 = (t + NTAPS - 1) % NTAPS;
                // A unit's eight MFMAs in program order with the NEXT-BUT-ONE unit's fragment reads (and, in a sub-step's second unit, the S
                // point's DMA pieces) placed in their shadows: a wave issues in order, so an instruction behind the cluster would wait for
                // all eight (a lone wave per SIMD ran the cluster-then-reads order at 54 % of the matrix pipe).  A fragment register is
                // re-requested right behind the last MFMA that reads it; sched_barrier pins the order.
                const unsigned boffn = tn == 0 ? bnext : bcur;
                const unsigned aan = a_lane + (unsigned)(((h + 1) & 3) * L::STAGE);        // sub-step h + 1: slot, halo buffer
                auto unit = [&](auto jc, auto&& shadow) {
                    constexpr int jj = decltype(jc)::value;
                    const unsigned b0 = (bx[tn] + boffn) ^ (jj << 5);
#define SK_MF(f, n) acc[f][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[jj][f], fb[jj][n], acc[f][n], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
                    if constexpr (NF == 2) {
                        // four MFMAs; every fragment register re-requested right behind the last MFMA that reads it
                        SK_MF(0, 0); shadow(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
                        SK_MF(1, 0); lds_read16_asm<0>(fb[jj][0], b0); shadow(std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
                        SK_MF(0, 1); lds_read16_asm<0 * 2048 + jj * 1024>(fa[jj][0], aan); shadow(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
                        SK_MF(1, 1); lds_read16_asm<1 * 2048 + jj * 1024>(fa[jj][1], aan); lds_read16_asm<2048>(fb[jj][1], b0); __builtin_amdgcn_sched_barrier(0);
                    } else {
                    SK_MF(0, 0); shadow(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
                    SK_MF(1, 0); shadow(std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
                    SK_MF(2, 0); shadow(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
#ifdef SK_ABL_NOREAD
                    SK_MF(3, 0); SK_MF(0, 1); SK_MF(1, 1); SK_MF(2, 1); SK_MF(3, 1); (void)b0; (void)aan;
#else
                    SK_MF(3, 0); lds_read16_asm<0>(fb[jj][0], b0); __builtin_amdgcn_sched_barrier(0);
                    SK_MF(0, 1); lds_read16_asm<0 * 2048 + jj * 1024>(fa[jj][0], aan); __builtin_amdgcn_sched_barrier(0);
                    SK_MF(1, 1); lds_read16_asm<1 * 2048 + jj * 1024>(fa[jj][1], aan); __builtin_amdgcn_sched_barrier(0);
                    SK_MF(2, 1); lds_read16_asm<2 * 2048 + jj * 1024>(fa[jj][2], aan); __builtin_amdgcn_sched_barrier(0);
                    SK_MF(3, 1); lds_read16_asm<3 * 2048 + jj * 1024>(fa[jj][3], aan); lds_read16_asm<2048>(fb[jj][1], b0); __builtin_amdgcn_sched_barrier(0);
#endif
                    }
#undef SK_MF
                };
                // ---- unit (t, 0) ----
#ifndef SK_ABL_NOREAD
                lgkm_wait_asm<NF + 2>();                             // R(t, 0) landed (R(t, 1) may be in flight)
#endif
                unit(std::integral_constant<int, 0>{}, [&](auto) {});
                // ---- unit (t, 1) ----
#ifndef SK_ABL_NOREAD
                lgkm_wait_asm<NF + 2>();                             // R(t, 1) landed: every read of stage h is complete
#endif
                {
                    // S point: stage h + 2 must be in LDS (requested two sub-steps ago); allowed in flight: the halo pieces of S(t - 1) and stage h + 3
                    constexpr int j0p = tm1 * HPER < NHW ? tm1 * HPER : NHW, j1p = (tm1 + 1) * HPER < NHW ? (tm1 + 1) * HPER : NHW;
                    constexpr int c1 = tm1 < TA ? j1p - j0p : 0;      // halo pieces S(t - 1) issued
#ifndef SK_ABL_NODMA
                    sk_wait_vm<PWN + c1>();
#endif
#ifndef SK_ABL_NOBAR
                    asm volatile("s_barrier" ::: "memory");
#endif
                }
                // the S point's DMA pieces: halo pieces first, then the stage (a later wait for the stage then covers the halo pieces in front of
                // it), spread over the shadows of the first three MFMAs
                unit(std::integral_constant<int, 1>{}, [&](auto sc) {
                    constexpr int sidx = decltype(sc)::value;
                    constexpr int j0 = t < TA ? (t * HPER < NHW ? t * HPER : NHW) : 0, j1 = t < TA ? ((t + 1) * HPER < NHW ? (t + 1) * HPER : NHW) : 0;
                    constexpr int nh = j1 - j0;                       // halo pieces of this S point; then PW stage pieces
                    // piece list: [halo j0 .. j1) [stage 0 .. PW): piece index q goes to shadow min(q * 3 / total, 2)
                    constexpr int total = nh + PWN;
                    static_for<0, total>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
#ifdef SK_ABL_NODMA
                        constexpr int sh = 99;
#else
                        constexpr int sh = q * 3 / total;
#endif
                        if constexpr (sh == sidx) {
                            if constexpr (q < nh) {
                                constexpr int j = j0 + q;
                                if (want_tab && c + 1 == ce && j * NW + wave < NTP) issue_table(cur.rt / RSUB, j * NW + wave, bnext ? 1 : 0);
                                else issue_halo(j, bnext ? 1 : 0);
                            } else {
                                constexpr int j = q - nh;
                                sk_dma16(a_seg + (long long)(cb * NTAPS + h + 4) * L::STAGE + j * 1024, lane16, (unsigned)(L::OFF_W + (h & 3) * L::STAGE + (wave * PWN + j) * 1024));
                            }
                        }
                    });
                });
                ++h;
            });
            bcur = bnext;
        }
        // drain: the reads past the segment's end (garbage, never used) and the DMAs past its end must be gone before the buffers are
        // refilled; the barrier makes that true for every wave's reads
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(fa[jj][f]));
            asm volatile("" : "+v"(fb[jj][0]), "+v"(fb[jj][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        SK_STAMP();
        // the next segment's first halo and stages fly under this segment's epilogue (same row tile: its LDS reads are uncounted inline asm)
        const bool same_rt = nxt.ok && (nxt.rt == rt_cur || !L::LDS_TAB);
        const unsigned tab_base = (unsigned)L::OFF_H + bcur;         // (bcur: the buffer behind the last chunk's = where its S points put the tables)
        hb0 = bcur ? 0 : 1;                                          // the next segment starts in the other one
        if (same_rt) prefetch(nxt);
        __builtin_amdgcn_sched_barrier(0);

        if (NF != 4 || (cb == 0 && ce == nch)) {                      // (short units are never cut)
            sk_epilogue<MW, NW, L::LDS_TAB ? 1 : 2, true, NF>(p, smem, nullptr, acc, cur.par, cur.rt / RSUB, qbase + cur.tile * L::NPX, wm, wn, lane, tab_base, -1,
                                                             (cur.rt % RSUB) * (32 * NF));
        } else if constexpr (NF == 4) {
            // raw accumulators, accumulator layout: [wave][f][n][reg / 4][lane][4] fp32 - coalesced 16-byte stores
            float* pw = p.partial + ((long long)(2 * lid + (cur.first_sk ? 0 : 1))) * L::part_floats() + wave * (128 * 64) + lane * 4;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<float4*>(pw + ((f * 2 + n) * 4 + g4) * 256) =
                            make_float4(acc[f][n][4 * g4], acc[f][n][4 * g4 + 1], acc[f][n][4 * g4 + 2], acc[f][n][4 * g4 + 3]);
        }
        SK_STAMP();
        if (!nxt.ok) break;
        if (!same_rt) {                                              // another row tile: its fold tables replace the current ones behind the epilogue
            __syncthreads();
            if (L::LDS_TAB) sk_row_table<MW, NW>(p, smem, nxt.rt, tid);
            rt_cur = nxt.rt;
            prefetch(nxt);
        }
        if (!L::LDS_TAB) { __syncthreads(); sample_range(nxt); }
        cur = nxt;
    }
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[dbg_off ? 254 : 255] = dbg_n;
#endif
}

// WIDE and SHORT units in one launch (round 6: the 36^2 level at B = 16; p.mix_nunits > 0, the <1, 4, 9> instantiation only).  344 units of 128 rows x
// 256 positions on 512 resident slots leave 168 CUs with ONE workgroup and give 88 two; a workgroup puts one wave on every SIMD and the matrix pipe
// belongs to the SIMD, so the launch lasts as long as two 128 x 64 wave tiles on one SIMD (a unit takes ~96 us alone, ~180 us beside a second one).
// In a mixed launch the first mix_wt pixel tiles are computed as wide units, the remaining mix_nt as SHORT units (conv_sk_body<.., NF = 2>: 64 rows,
// wave tile 64 x 64 - half the work per wave, same halo) such that wide + short units = 2 x CUs: every CU gets exactly two workgroups and no SIMD
// more than a wide and a short wave tile.  (A first version halved the POSITIONS instead - 128 x 128 units of two waves - and changed nothing: that
// halves the waves, not a wave's work; profiles/EXPERIMENTS.md.)  Order: XCD x gets the x-th eighth of the wide units, then the x-th eighth of the
// short ones (both counts multiples of 8); within an XCD workgroups are handed out in blockIdx order and every CU takes one before any takes a
// second, in the same CU order (tools/micro/dispatch_map.hip: (j, j + 32) on one CU for 512 / 512 pairs), so the wide ones, dispatched first, land
// on different CUs and each gets a short partner.  The block's res_conv units stay wide and stay the grid's tail.
template <int MW, int NW, int NTAPS>
__global__ __launch_bounds__(64 * NW, 2) void conv_sk_kernel(const ConvSkP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int lid, G;
    {
        int nblk = gridDim.x, bid = blockIdx.x;
        bool alt = false;
        if (MW == 1 && NW == 4 && p.alt_units) {                     // the grid's last workgroups: the block's 1x1 res_conv
            const int nmain = nblk - p.alt_units;
            if (bid >= nmain) { alt = true; bid -= nmain; nblk = p.alt_units; } else nblk = nmain;
        }
        bool mixed = false;
        if constexpr (MW == 1 && NW == 4 && NTAPS == 9) {
            if (p.mix_nunits && !alt) {
                // (G, the unit count and the whole-unit count reach the body as separate run-time values although a mixed launch has them equal:
                // with `ndp == units == G` visible at compile time hipcc folds the segment loop away and the code that is left spills 56 - 125 registers)
                mixed = true;
                const int wu = p.units, nu = p.mix_nunits;           // wide / short units (multiples of 8)
                const int xcd = bid & 7, j = bid >> 3, wpx = wu >> 3, npx = nu >> 3;
                if (j < wpx) { lid = xcd * wpx + j; G = wu; }
                else { conv_sk_body<1, 4, 9, 2>(p, smem, xcd * npx + (j - wpx), nu, nu, p.mix_nndp, p.mix_q0, p.nhp); return; }
            }
        }
        if (!mixed) {
            const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
            lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
            G = nblk;
        }
        if (MW == 1 && NW == 4) {
            if (alt) {
                const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                sk_alt_unit<NW>(p, smem, lid, wave, lane);
                return;
            }
        }
    }
    conv_sk_body<MW, NW, NTAPS>(p, smem, lid, G, p.units, p.ndp, 0, p.nhp);
}

// Second half of a stream-K launch: sums the partial tiles of every cut unit in ascending workgroup order (fixed: bit-reproducible)
// and runs the epilogue.  grid = (4 x waves, stream-K units): one WAVE per workgroup, one 32-row fragment of the slice of the unit that wave `blockIdx.x / 4` of the conv
// kernel owns (many small workgroups: the pass is bound by reading the partial tiles, ~80 MB per launch); units that one
// workgroup computed whole exit at once.
template <int MW, int NW>
__global__ __launch_bounds__(64) void conv_sk_finish_kernel(const ConvSkP p, int G) {
    using L = CvSk<MW, NW>;
    __shared__ float ms[2 * L::MAXB];
    constexpr int NWN = L::NWN;
    const int lane = threadIdx.x;
    const int wave = blockIdx.x >> 2, fsel = blockIdx.x & 3;         // one 32-row fragment (two MFMA tiles) of one wave's slice per workgroup
    const int wm = wave / NWN, wn = wave % NWN;
    const int nch = p.nchunks;
    const SkSched sch(p.units, p.ndp, nch, G);
    const int u = blockIdx.y;                                        // stream-K unit index
    const long long a = (long long)u * nch, b = a + nch;             // its chunk range
    // workgroups whose range meets [a, b): the first is the one that holds chunk a
    int g = (int)(a * G / sch.tc);
    if (g >= G) g = G - 1;
    while (g > 0 && sch.start(g) > a) --g;
    while (sch.start(g + 1) <= a) ++g;
    if (sch.start(g) <= a && sch.start(g + 1) >= b) return;          // computed whole by one workgroup
    int par = 0, rt, tile;
    {
        int uu = p.ndp + u;
        const int per_par = p.rowtiles * p.ntiles;
        if (p.npar > 1) { par = uu / per_par; uu -= par * per_par; }
        tile = uu / p.rowtiles; rt = uu - tile * p.rowtiles;
    }
    // the first two parts (every cut unit has at least two) are requested BEFORE the per-sample scalars are built: the statistics loads,
    // the fp64 mean / variance and the partial tiles are independent chains, the pass is latency-bound
    const int fo = fsel * 2 * 4 * 256;                               // this workgroup's fragment inside a wave slice [f][n][reg / 4][lane][4]
    auto part_ptr = [&](int gg) {
        const bool first = sch.start(gg) >= a;                       // this unit holds the workgroup's first chunk: its first segment
        return p.partial + ((long long)(2 * gg + (first ? 0 : 1))) * L::part_floats() + wave * (128 * 64) + lane * 4 + fo;
    };
    float4 v0[2][4], v1[2][4];
    // (workgroups with an empty chunk range - start(g) == start(g + 1) - wrote nothing in this launch: never a part)
    const bool two = g + 1 < G && sch.start(g + 1) < b && sch.start(g + 2) > sch.start(g + 1);
    {
        const float* p0 = part_ptr(g);
        const float* p1 = part_ptr(two ? g + 1 : g);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { v0[n][g4] = *reinterpret_cast<const float4*>(p0 + (n * 4 + g4) * 256); v1[n][g4] = *reinterpret_cast<const float4*>(p1 + (n * 4 + g4) * 256); }
    }
    {
        const int per_b = p.ns * p.HpWpe;                            // the few samples this wave's 64 positions span
        const int q0 = tile * L::NPX + wn * 64;
        const int b0 = sk_udiv(q0, per_b, p.inv_per_b);
        int b1 = sk_udiv(q0 + 63, per_b, p.inv_per_b) + 1; b1 = b1 < p.nb ? b1 : p.nb;
        if (b0 < b1) sk_sample_table(p, ms + 2 * b0, b0, b1, 0, 1, lane);
    }
    f32x16_t acc[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        if (f != fsel) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                acc[f][n][4 * g4] = v0[n][g4].x; acc[f][n][4 * g4 + 1] = v0[n][g4].y; acc[f][n][4 * g4 + 2] = v0[n][g4].z; acc[f][n][4 * g4 + 3] = v0[n][g4].w;
                if (two) { acc[f][n][4 * g4] += v1[n][g4].x; acc[f][n][4 * g4 + 1] += v1[n][g4].y; acc[f][n][4 * g4 + 2] += v1[n][g4].z; acc[f][n][4 * g4 + 3] += v1[n][g4].w; }
            }
    }
    for (g += two ? 2 : 1; g < G && sch.start(g) < b; ++g) {          // further parts, in part order
        if (sch.start(g + 1) == sch.start(g)) continue;
        const float* pr = part_ptr(g);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f != fsel) continue;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 v = *reinterpret_cast<const float4*>(pr + (n * 4 + g4) * 256);
                    acc[f][n][4 * g4] += v.x; acc[f][n][4 * g4 + 1] += v.y; acc[f][n][4 * g4 + 2] += v.z; acc[f][n][4 * g4 + 3] += v.w;
                }
        }
    }
    __syncthreads();
    sk_epilogue<MW, NW, 0, false>(p, nullptr, ms, acc, par, rt, tile * L::NPX, wm, wn, lane, 0, fsel);
}
