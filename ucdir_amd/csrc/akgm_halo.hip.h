// AKGM tail (GroupNorm2 -> grouped spdyconv -> per-pixel modulation sum -> swish -> + residual) for
// C >= 128 (16 / 32 / 64 channels per group) on a 2-D pixel tile with an LDS halo (gfx950).
// Reference: model/ucdir.py:129-140.
//
// Same structure as conv_halo.hip.h (512 threads, 16x16-style tile, 64-byte swizzled LDS rows, A
// tiles of 128 rows x 64 k through a 2-deep ring, raw s_barrier), plus:
//   * the K order of the packed weights makes every 16-wide MFMA k step one tap x 16 contiguous
//     channels, so K is exact (144 / 288 / 576 per group: no padded taps on the matrix cores);
//   * one workgroup stages the halo of its 32-channel chunk(s) ONCE and runs several "units" over
//     it: the 2 groups of a chunk (cg = 16), or the 2 / 4 row tiles of a group (cg = 32 / 64);
//   * per unit the 8 kernel sets of a feature sit in one lane's accumulators (row permutation of
//     pack_akgm), the modulation sum happens in registers against a [9][128] fold table in LDS, and the
//     epilogue needs neither an LDS stage nor a barrier (half-wave exchange, see akgm_pre.hip.h): the weight
//     ring keeps streaming across unit boundaries, so the next unit's first stage lands under this unit's epilogue.
#pragma once
#include "conv_halo.hip.h"

struct AkgmHP {
    const bf16_t* A; int Kpad;                 // [8 groups][C rows][Kpad], K order see pack_akgm_halo
    const bf16_t* h; long long h_bstride;      // swish(conv1), zero-bordered NHWC, C channels
    int C, cg, H, W, Wp, th, tw, tiles_x, tiles_y, nbatch;
    const stat_t* stats; double inv_count;
    const float* Tc;                           // [B][9][8C]: (bias + Tb)/rstd - mean*Tg (akgm_tc_kernel), original row order
    const float* ms;                           // [B][2]: (mean, rstd) of the input, written by akgm_tc_kernel
    const float* G; long long g_bstride; const float* attw;
    const bf16_t* res; long long res_bstride;
    bf16_t* out; long long out_bstride;
    stat_t* stats_out;                         // (sum, sum of squares) accumulators of the output (stat_add)
    int usplit;                                // akgm_halo_stage_kernel, 64 per group: the 4 units of a group go to 1 | 2 | 4 workgroups
    // own_tc (the persistent kernels): no akgm_tc_kernel launch in front - a workgroup entering sample b sums the statistics slots itself and
    // forms its slice of Tc from the sample-independent tables Tbb = bias + Tb and Tg ([9][8C] each) on the way into LDS (akgm_tc_piece)
    int own_tc; const float* Tbb; const float* Tgt;
    int reverse;                               // persistent kernels: walk the tile range from its end (meets the lines the producer wrote LAST - still in L2 / the Infinity Cache - first)
    unsigned long long* dbg;
};

#define AH_TM 128
#define AH_ASTAGE (2 * AH_TM * 64)               // 16384: [2 halves of 32 k][128 rows][64 B]
#define AH_SL 20                                 // floats per pixel in the output stage (16 features + pad)
#define AH_LDS (2 * HC_HALO_BYTES + 2 * AH_ASTAGE + 128 + 9 * AH_TM * 4)

// Tc[b][cls][o] = (bias[o] + Tb[cls][o]) / rstd_b - mean_b * Tg[cls][o]   (once per launch; grid (9, B)): the GroupNorm
// fold constant of row o DIVIDED by rstd, so rstd is applied once after the modulation sum,
//   rstd * sum_s att_s (conv_s + Tc_s) = sum_s att_s (rstd conv_s + fold_s),
// and akgm_pre.hip.h can start its accumulators at Tc instead of zero
__global__ void akgm_tc_kernel(const stat_t* __restrict__ stats, double inv_count, const float* __restrict__ bias,
                               const float* __restrict__ Tb, const float* __restrict__ Tg, int n, float* __restrict__ Tc,
                               float* __restrict__ ms) {
    // grid (9 classes x ceil(n / 1024), B), 256 threads, four values per thread (n = 8C is a multiple of 512)
    const int b = blockIdx.y, cls = blockIdx.x % 9, o = ((blockIdx.x / 9) * 256 + threadIdx.x) * 4;
    float mean, rstd;
    double S, Q;
    stat_read(stats, nullptr, b, S, Q);
    mean_rstd(S, Q, inv_count, mean, rstd);
    // the AKGM workgroups (10,368 per launch at the 288^2 level) read these two floats instead of each of their 512
    // threads summing 32 fixed-point slots and redoing the fp64 mean / variance arithmetic
    if (blockIdx.x == 0 && threadIdx.x == 0) { ms[2 * b] = mean; ms[2 * b + 1] = rstd; }
    if (o >= n) return;
    const float inv = 1.0f / rstd;
    const float4 bi = *reinterpret_cast<const float4*>(bias + o);
    const float4 tb = *reinterpret_cast<const float4*>(Tb + (long long)cls * n + o);
    const float4 tg = *reinterpret_cast<const float4*>(Tg + (long long)cls * n + o);
    *reinterpret_cast<float4*>(Tc + ((long long)b * 9 + cls) * n + o) =
        make_float4((bi.x + tb.x) * inv - mean * tg.x, (bi.y + tb.y) * inv - mean * tg.y,
                    (bi.z + tb.z) * inv - mean * tg.z, (bi.w + tb.w) * inv - mean * tg.w);
}

// 16 bytes of a persistent kernel's Tc slice, the arithmetic of akgm_tc_kernel: rel = float index in the [9][8C] tables, inv = 1 / rstd_b
__device__ __forceinline__ void akgm_tc_piece(const AkgmHP& p, long long rel, unsigned char* dst, float inv, float mean) {
    const float4 tb = *reinterpret_cast<const float4*>(p.Tbb + rel), tg = *reinterpret_cast<const float4*>(p.Tgt + rel);
    *reinterpret_cast<float4*>(dst) = make_float4(tb.x * inv - mean * tg.x, tb.y * inv - mean * tg.y, tb.z * inv - mean * tg.z, tb.w * inv - mean * tg.w);
}

// the slice Tc[b][cls][8 fbase .. + 128) of a one-shot AKGM workgroup, formed where the LDS-DMA from akgm_tc_kernel's table was issued: waves 0 - 4,
// lane -> (class 2 w + lane / 32, 16 bytes), the DMA's LDS addresses.  The caller makes the writes visible (lgkmcnt(0) in front of its next barrier).
__device__ __forceinline__ void akgm_tc_slice(const AkgmHP& p, int fbase, float* dst, int wave, int lane, float inv, float mean) {
    if (wave < 5) {
        const int cl = 2 * wave + (lane >> 5);
        if (cl < 9) akgm_tc_piece(p, (long long)cl * 8 * p.C + 8 * fbase + (lane & 31) * 4, reinterpret_cast<unsigned char*>(dst) + wave * 1024 + lane * 16, inv, mean);
    }
}

// ATT_LDS (16 / 32 channels per group: one halo chunk per workgroup, the second halo buffer is free): the per-pixel
// modulation weights G*attw live in that buffer instead of 16 VGPRs, and the accumulators start at the fold constants
// (as in akgm_pre.hip.h) so the epilogue needs 8 instead of 16 FMAs per output.  With 64 per group both halo buffers
// are in use and the register version stays.
template <bool ATT_LDS>
__global__ __launch_bounds__(HC_THREADS, 4) void akgm_halo_kernel(const AkgmHP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* aring = smem + 2 * HC_HALO_BYTES;
    float* scal = reinterpret_cast<float*>(smem + 2 * HC_HALO_BYTES + 2 * AH_ASTAGE);
    float* tcs = scal + 32;                                          // [9][128]
    float* attl = reinterpret_cast<float*>(halo + HC_HALO_BYTES);    // ATT_LDS: [256 px][8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: no waterfall loops around global_load_lds
    const int wm = wave >> 2, wq = wave & 3, hh = lane >> 5;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int cg = p.cg;
    const int nsec = (cg == 8) ? p.C / 32 : ((cg == 16) ? 4 : 8);   // work sections per pixel tile: 32-channel chunks or groups
    const int sec = lid % nsec;
    int tq = lid / nsec;
    const int tx = tq % p.tiles_x; tq /= p.tiles_x;
    const int ty = tq % p.tiles_y;
    const int b = tq / p.tiles_y;
    const int th = p.th, tw = p.tw, hw = tw + 2;
    const int y0 = ty * th, x0 = tx * tw;
    const int hcount = (th + 2) * hw;
    const int nslots = th * tw;
    const float inv_hw = 1.0f / (float)hw, inv_tw = 1.0f / (float)tw;
    const int nunits = (cg == 64) ? 4 : 2;
    const int nchunks = (cg == 64) ? 2 : 1;                 // halo chunks (32 channels each) this workgroup needs
    const int chunk0 = (cg == 64) ? 2 * sec : sec;
    const int tshift = (cg == 16) ? 0 : 1;                  // k16 step -> tap: tap = k16 >> tshift   (cg >= 16)
    const int spc = (cg == 8) ? 2 : ((cg == 16) ? 3 : 5);   // A stages (64 k) per 32-channel period

    float rstd, mean_b = 0.f;
    if (p.own_tc) stat_mean_rstd_wave(p.stats, b, p.inv_count, lane, mean_b, rstd);   // (no akgm_tc_kernel launch in front: AkgmHP::own_tc)
    else rstd = p.ms[2 * b + 1];
    const float inv_b = 1.0f / rstd;

    // ---- halo: stage chunk(s) once -----------------------------------------------------------------
    {
        const bf16_t* hb = p.h + (long long)b * p.h_bstride;
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int hp = (i * 8 + wave) * 16 + (lane >> 2);
                if ((i * 8 + wave) * 16 < hcount) {
                    if (hp < hcount) {
                        const int hr = fdiv_small(hp, inv_hw), hc = hp - hr * hw;
                        int gy = y0 + hr, gx = x0 + hc;
                        gy = gy > p.H + 1 ? p.H + 1 : gy;
                        gx = gx > p.W + 1 ? p.W + 1 : gx;
                        const int j = (lane & 3) ^ ((hp >> 2) & 3);
                        stage16(hb + (long long)(gy * p.Wp + gx) * p.C + (chunk0 + c) * 32 + j * 8,
                                halo + c * HC_HALO_BYTES + (i * 8 + wave) * 1024, lane);
                    }
                }
            }
        }
    }

    // ---- per-lane pixel constants --------------------------------------------------------------------
    int hp0[2], cls[2], goff[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
        int slot = wq * 64 + tp * 32 + (lane & 31);
        const bool inb = slot < nslots;
        slot = inb ? slot : nslots - 1;
        const int r = fdiv_small(slot, inv_tw), c = slot - r * tw;
        hp0[tp] = r * hw + c;
        int y = y0 + r, x = x0 + c;
        y = y < p.H ? y : p.H - 1; x = x < p.W ? x : p.W - 1;
        cls[tp] = (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1));
        goff[tp] = (y * p.W + x) * 8;
        if (ATT_LDS) {
            const float* gp = p.G + (long long)b * p.g_bstride + goff[tp];
            const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
            const float* aw = p.attw + b * 8;
            if (wm == 0) {                   // the two row halves see the same pixels: one of them publishes G * attw
                float* ap = attl + (wq * 64 + tp * 32 + (lane & 31)) * 8;
                *reinterpret_cast<float4*>(ap) = make_float4(g0.x * aw[0], g0.y * aw[1], g0.z * aw[2], g0.w * aw[3]);
                *reinterpret_cast<float4*>(ap + 4) = make_float4(g1.x * aw[4], g1.y * aw[5], g1.z * aw[6], g1.w * aw[7]);
            }
        }
    }
    int a_off[2], a_sw[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int row = wm * 64 + tm * 32 + (lane & 31);
        a_off[tm] = row * 64; a_sw[tm] = (row >> 2) & 3;
    }
    const int arow = wave * 16 + (lane >> 2);
    const int ajsw = (lane & 3) ^ ((arow >> 2) & 3);

#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 5);
    int dbg_n = 0;
#define AH_STAMP() do { if (dbg_on) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AH_STAMP() do {} while (0)
#endif
    AH_STAMP();
    // store item of this lane: pixel 64 wq + lane, features 8 wm .. 8 wm + 7 of the unit (see the epilogue); offset or -1
    int off2;
    {
        const int px = wq * 64 + lane;
        const int r = fdiv_small(px, inv_tw), c = px - r * tw;
        const int y = y0 + r, x = x0 + c;
        off2 = (px < nslots && y < p.H && x < p.W) ? ((y + 1) * p.Wp + (x + 1)) * p.C + wm * 8 : -1;
    }
    // unit -> (weights, first feature, first 16-byte chunk of the halo row)
    auto unit_geom = [&](int unit, const bf16_t*& Au, int& fbase, int& base16) {
        if (cg == 8) { const int group = 4 * sec + 2 * unit; fbase = group * 8; base16 = 2 * unit + wm; Au = p.A + (long long)group * 64 * p.Kpad; }
        else if (cg == 16) { const int group = 2 * sec + unit; fbase = group * 16; base16 = unit * 2; Au = p.A + (long long)group * p.C * p.Kpad; }
        else { fbase = sec * cg + unit * 16; base16 = 0; Au = p.A + ((long long)sec * p.C + unit * AH_TM) * p.Kpad; }
    };
    // fold table slice of a unit, Tc[b][cls][8*fbase .. +128), DMA'd into LDS: instruction w (waves 0-4) carries classes
    // 2w and 2w+1 (32 lanes x 16 B each).  ATT_LDS keeps two slices (the accumulators of unit u+1 start from its slice
    // while nothing of unit u reads its own any more); the second one sits behind the modulation weights in the free halo
    // buffer.  Without ATT_LDS the slice is read in the epilogue: one buffer, refilled after the next unit's first barrier.
    float* tcs1 = ATT_LDS ? attl + 256 * 8 : tcs;
    // (own_tc: the slice is formed from the sample-independent tables by plain loads - issued and waited for IN FRONT of the weight stage's DMA
    // that follows every call, so that only an L2 round trip of two 16-byte loads is exposed - and written to the DMA's LDS addresses)
    auto issue_Tc = [&](int fbase, float* dst) {
        if (p.own_tc) {
            akgm_tc_slice(p, fbase, dst, wave, lane, inv_b, mean_b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (wave < 5) {
            const int cl = 2 * wave + (lane >> 5);
            if (cl < 9)
                __builtin_amdgcn_global_load_lds(
                    (const GLOBAL_AS void*)(p.Tc + ((long long)b * 9 + cl) * 8 * p.C + 8 * fbase + (lane & 31) * 4),
                    (LDS_AS void*)(reinterpret_cast<unsigned char*>(dst) + wave * 1024), 16, 0, 0);
        }
    };
    auto issue_A = [&](const bf16_t* Au, int st, int slot) {
        unsigned char* ab = aring + slot * AH_ASTAGE + wave * 1024;
        const bf16_t* src = Au + (long long)arow * p.Kpad + st * 64 + ajsw * 8;
        stage16(src, ab, lane);
        stage16(src + 32, ab + AH_TM * 64, lane);
    };
    const int nk = spc * nchunks;
    float s1 = 0.f, s2 = 0.f;
    int gs = 0;                                  // weight stages consumed so far: stage k of the whole workgroup lives in ring slot k & 1
    {
        const bf16_t* Au0; int fb0, b16;
        unit_geom(0, Au0, fb0, b16);
        issue_Tc(fb0, tcs);
        issue_A(Au0, 0, 0);
    }
    for (int unit = 0; unit < nunits; ++unit) {
        int fbase, base16;
        const bf16_t* Au;
        unit_geom(unit, Au, fbase, base16);
        const float* tcu = (ATT_LDS && (unit & 1)) ? tcs1 : tcs;
        AH_STAMP();
        f32x16_t acc[2][2];
        if (!ATT_LDS) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tm][tp][e] = 0.f;
        }

        int cch = 0, sp = 0;                     // halo chunk and stage index within the chunk period
        auto kstep = [&](int s) {
            HC_WAIT(0);
            asm volatile("s_barrier" ::: "memory");          // stage gs landed; every wave is past stage gs - 1 (and past the previous unit's epilogue)
            if (ATT_LDS && s == 0) {
                // the fold table slice has landed: start at Tc[cls(pixel)][row] (registers 8q..8q+7 of a tile = the 8 sets
                // of feature 4*t32 + 2q + hh, original row order 8*feature + set)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const float* tc = tcu + cls[tp] * AH_TM;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float* t8 = tc + 8 * (4 * (wm * 2 + tm) + 2 * q + hh);
                            const float4 c0 = *reinterpret_cast<const float4*>(t8), c1 = *reinterpret_cast<const float4*>(t8 + 4);
                            acc[tm][tp][8 * q + 0] = c0.x; acc[tm][tp][8 * q + 1] = c0.y; acc[tm][tp][8 * q + 2] = c0.z; acc[tm][tp][8 * q + 3] = c0.w;
                            acc[tm][tp][8 * q + 4] = c1.x; acc[tm][tp][8 * q + 5] = c1.y; acc[tm][tp][8 * q + 6] = c1.z; acc[tm][tp][8 * q + 7] = c1.w;
                        }
                }
            }
            if (!ATT_LDS && s == 0 && unit > 0) issue_Tc(fbase, tcs);      // single slice: the previous unit's epilogue is behind the barrier
            // the next stage goes into the slot stage gs - 1 has just left: this unit's, or - across the unit boundary, so that
            // it lands under this unit's epilogue - the next unit's first one (with its fold table slice)
            if (s + 1 < nk) issue_A(Au, s + 1, (gs + 1) & 1);
            else if (unit + 1 < nunits) {
                const bf16_t* Aun; int fbn, b16n;
                unit_geom(unit + 1, Aun, fbn, b16n);
                if (ATT_LDS) issue_Tc(fbn, ((unit + 1) & 1) ? tcs1 : tcs);
                issue_A(Aun, 0, (gs + 1) & 1);
            }
            const unsigned char* Hb = halo + cch * HC_HALO_BYTES;
            const unsigned char* Ab = aring + (gs & 1) * AH_ASTAGE;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k16 = sp * 4 + j;
                // cg == 8: one k16 step = taps 2*k16 (lanes 0-31) and 2*k16+1 (lanes 32-63) x 8 channels
                int tap = (cg == 8) ? 2 * k16 : (k16 >> tshift);
                if (tap < 9) {
                    if (cg == 8) { tap += hh; tap = tap > 8 ? 8 : tap; }       // tap 9 has zero weights
                    const int ky = tap_ky(tap), kx = tap - 3 * ky;
                    const int sh = ky * hw + kx;
                    const int ch16 = (cg == 8) ? base16 : base16 + ((k16 & tshift) << 1) + hh;   // 16-byte chunk inside the 64-byte halo row
                    const int kch = (j & 1) * 2 + hh;                           // chunk inside the A half-stage row
                    const unsigned char* Ah = Ab + (j >> 1) * (AH_TM * 64);
                    bf16x8_t af[2], bfr[2];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        af[tm] = *reinterpret_cast<const bf16x8_t*>(Ah + a_off[tm] + ((kch ^ a_sw[tm]) << 4));
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) {
                        const int hp = hp0[tp] + sh;
                        bfr[tp] = *reinterpret_cast<const bf16x8_t*>(Hb + hp * 64 + ((ch16 ^ ((hp >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tp = 0; tp < 2; ++tp)
                            acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tp], acc[tm][tp], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (++sp == spc) { sp = 0; ++cch; }
            ++gs;
        };
        for (int s = 0; s < nk - 1; ++s) kstep(s);
        // what the epilogue needs from HBM / L2 is requested one K step ahead of it (the last step is peeled: values
        // defined inside the loop would occupy their registers for ALL its iterations and spill), and ahead of the next
        // unit's DMA in the in-order VMEM queue
        uint4 rv = make_uint4(0, 0, 0, 0);       // residual of this lane's store item
        float4 gq[ATT_LDS ? 1 : 2][2];           // without ATT_LDS: G of the lane's two pixels
        if (off2 >= 0) rv = *reinterpret_cast<const uint4*>(p.res + (long long)b * p.res_bstride + off2 + fbase);
        kstep(nk - 1);
        if (!ATT_LDS) {                          // 64 channels per group: no room for these 16 registers during a K step
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const float* gp = p.G + (long long)b * p.g_bstride + goff[tp];
                gq[ATT_LDS ? 0 : tp][0] = *reinterpret_cast<const float4*>(gp);
                gq[ATT_LDS ? 0 : tp][1] = *reinterpret_cast<const float4*>(gp + 4);
            }
        }

        AH_STAMP();
        // ---- epilogue WITHOUT LDS stage and WITHOUT barriers (see akgm_pre.hip.h): modulation sum in registers,
        // vq[tm][q][tp] = feature 8 wm + 4 tm + 2 q + hh of pixel tp; one v_permlane32_swap per feature pair gives lane L
        // all eight features 8 wm .. + 7 of pixel 64 wq + L ---------------------------------------------------------------
        float vq[2][2][2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int px = wq * 64 + tp * 32 + (lane & 31);
            const float* tc = tcu + cls[tp] * AH_TM;
            float av[8];
            if (ATT_LDS) {
                const float4 a0 = *reinterpret_cast<const float4*>(attl + px * 8), a1 = *reinterpret_cast<const float4*>(attl + px * 8 + 4);
                av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            } else {
                const float* aw = p.attw + b * 8;
                const float4 g0 = gq[ATT_LDS ? 0 : tp][0], g1 = gq[ATT_LDS ? 0 : tp][1];
                av[0] = g0.x * aw[0]; av[1] = g0.y * aw[1]; av[2] = g0.z * aw[2]; av[3] = g0.w * aw[3];
                av[4] = g1.x * aw[4]; av[5] = g1.y * aw[5]; av[6] = g1.z * aw[6]; av[7] = g1.w * aw[7];
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int t32 = wm * 2 + tm;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int floc = 4 * t32 + 2 * q + hh;
                    float sa = 0.f, sb = 0.f;
                    if (!ATT_LDS) {
                        const float4 c0 = *reinterpret_cast<const float4*>(tc + 8 * floc);
                        const float4 c1 = *reinterpret_cast<const float4*>(tc + 8 * floc + 4);
                        const float tcv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                        for (int s = 0; s < 8; ++s) sb += av[s] * tcv[s];
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) sa += av[s] * acc[tm][tp][8 * q + s];
                    vq[tm][q][tp] = rstd * (sa + sb);
                }
            }
        }
        float o8[8];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float lo = vq[tm][q][0], hi = vq[tm][q][1];
                permlane32_swap(lo, hi);
                o8[4 * tm + 2 * q + 0] = lo;
                o8[4 * tm + 2 * q + 1] = hi;
            }
        AH_STAMP();
        if (off2 >= 0) {
            const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv);
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vv[i] = silu_fast(o8[i]) + bf2f(rh[i]);
                s1 += vv[i]; s2 += vv[i] * vv[i];
            }
            *reinterpret_cast<uint4*>(p.out + (long long)b * p.out_bstride + off2 + fbase) = pack8_bf16(vv);
        }
    }
    AH_STAMP();
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    if (p.stats_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        __syncthreads();
        if (lane == 0) { scal[2 + wave * 2] = s1; scal[3 + wave * 2] = s2; }
        __syncthreads();
        if (tid == 0) {
            float t1 = 0.f, t2 = 0.f;
            for (int w = 0; w < 8; ++w) { t1 += scal[2 + w * 2]; t2 += scal[3 + w * 2]; }
            stat_add(p.stats_out, b, t1, t2);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// 64 channels per group (the 36^2 / 18^2 levels; also the fallback for 8 per group): both halo buffers hold channels, there
// is no LDS left for the modulation weights or a second fold-table slice, and the in-register epilogue above costs 18
// registers more than the 128 a wave has here (it spilled, with scratch reloads inside the K loop).  This variant keeps the
// fp32 LDS stage (aliasing the weight ring) between two barriers per unit; its K loops are 10 steps long, so the
// epilogue is a fifth of a unit, not a third.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HC_THREADS, 4) void akgm_halo_stage_kernel(const AkgmHP p) {
    constexpr bool ATT_LDS = false;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* aring = smem + 2 * HC_HALO_BYTES;
    float* stage = reinterpret_cast<float*>(aring);                  // aliases the A ring between units
    float* scal = reinterpret_cast<float*>(smem + 2 * HC_HALO_BYTES + 2 * AH_ASTAGE);
    float* tcs = scal + 32;                                          // [9][128]
    float* attl = reinterpret_cast<float*>(halo + HC_HALO_BYTES);    // ATT_LDS: [256 px][8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: no waterfall loops around global_load_lds
    const int wm = wave >> 2, wq = wave & 3, hh = lane >> 5;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int cg = p.cg;
    const int nsec = (cg == 8) ? p.C / 32 : ((cg == 16) ? 4 : 8);   // work sections per pixel tile: 32-channel chunks or groups
    // under-filled grids (B = 1, the 18^2 level): the units of a section are spread over usplit workgroups
    const int usplit = (cg == 64 && p.usplit > 1) ? p.usplit : 1;
    const int usub = lid % usplit;
    lid /= usplit;
    const int sec = lid % nsec;
    int tq = lid / nsec;
    const int tx = tq % p.tiles_x; tq /= p.tiles_x;
    const int ty = tq % p.tiles_y;
    const int b = tq / p.tiles_y;
    const int th = p.th, tw = p.tw, hw = tw + 2;
    const int y0 = ty * th, x0 = tx * tw;
    const int hcount = (th + 2) * hw;
    const int nslots = th * tw;
    const float inv_hw = 1.0f / (float)hw, inv_tw = 1.0f / (float)tw;
    const int nunits = ((cg == 64) ? 4 : 2) / usplit;
    const int unit0 = usub * nunits;
    const int nchunks = (cg == 64) ? 2 : 1;                 // halo chunks (32 channels each) this workgroup needs
    const int chunk0 = (cg == 64) ? 2 * sec : sec;
    const int tshift = (cg == 16) ? 0 : 1;                  // k16 step -> tap: tap = k16 >> tshift   (cg >= 16)
    const int spc = (cg == 8) ? 2 : ((cg == 16) ? 3 : 5);   // A stages (64 k) per 32-channel period

    float rstd, mean_b = 0.f;
    if (p.own_tc) stat_mean_rstd_wave(p.stats, b, p.inv_count, lane, mean_b, rstd);   // (no akgm_tc_kernel launch in front: AkgmHP::own_tc)
    else rstd = p.ms[2 * b + 1];
    const float inv_b = 1.0f / rstd;

    // ---- halo: stage chunk(s) once -----------------------------------------------------------------
    {
        const bf16_t* hb = p.h + (long long)b * p.h_bstride;
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int hp = (i * 8 + wave) * 16 + (lane >> 2);
                if ((i * 8 + wave) * 16 < hcount) {
                    if (hp < hcount) {
                        const int hr = fdiv_small(hp, inv_hw), hc = hp - hr * hw;
                        int gy = y0 + hr, gx = x0 + hc;
                        gy = gy > p.H + 1 ? p.H + 1 : gy;
                        gx = gx > p.W + 1 ? p.W + 1 : gx;
                        const int j = (lane & 3) ^ ((hp >> 2) & 3);
                        stage16(hb + (long long)(gy * p.Wp + gx) * p.C + (chunk0 + c) * 32 + j * 8,
                                halo + c * HC_HALO_BYTES + (i * 8 + wave) * 1024, lane);
                    }
                }
            }
        }
    }

    // ---- per-lane pixel constants --------------------------------------------------------------------
    int hp0[2], cls[2]; bool valid[2];
    float att[ATT_LDS ? 1 : 2][8];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
        int slot = wq * 64 + tp * 32 + (lane & 31);
        const bool inb = slot < nslots;
        slot = inb ? slot : nslots - 1;
        const int r = fdiv_small(slot, inv_tw), c = slot - r * tw;
        hp0[tp] = r * hw + c;
        int y = y0 + r, x = x0 + c;
        valid[tp] = inb && y < p.H && x < p.W;
        y = y < p.H ? y : p.H - 1; x = x < p.W ? x : p.W - 1;
        cls[tp] = (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1));
        const float* gp = p.G + (long long)b * p.g_bstride + ((long long)y * p.W + x) * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        const float* aw = p.attw + b * 8;
        if (ATT_LDS) {
            if (wm == 0) {                   // the two row halves see the same pixels: one of them publishes G * attw
                float* ap = attl + (wq * 64 + tp * 32 + (lane & 31)) * 8;
                *reinterpret_cast<float4*>(ap) = make_float4(g0.x * aw[0], g0.y * aw[1], g0.z * aw[2], g0.w * aw[3]);
                *reinterpret_cast<float4*>(ap + 4) = make_float4(g1.x * aw[4], g1.y * aw[5], g1.z * aw[6], g1.w * aw[7]);
            }
        } else {
            att[tp][0] = g0.x * aw[0]; att[tp][1] = g0.y * aw[1]; att[tp][2] = g0.z * aw[2]; att[tp][3] = g0.w * aw[3];
            att[tp][4] = g1.x * aw[4]; att[tp][5] = g1.y * aw[5]; att[tp][6] = g1.z * aw[6]; att[tp][7] = g1.w * aw[7];
        }
    }
    int a_off[2], a_sw[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int row = wm * 64 + tm * 32 + (lane & 31);
        a_off[tm] = row * 64; a_sw[tm] = (row >> 2) & 3;
    }
    const int arow = wave * 16 + (lane >> 2);
    const int ajsw = (lane & 3) ^ ((arow >> 2) & 3);

#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 5);
    int dbg_n = 0;
#endif
    AH_STAMP();
    float s1 = 0.f, s2 = 0.f;
    for (int unit = unit0; unit < unit0 + nunits; ++unit) {
        int group, fbase, base16;
        const bf16_t* Au;
        if (cg == 8) {       // unit = two adjacent groups (64 rows each): row half wm belongs to group 4*sec + 2*unit + wm
            group = 4 * sec + 2 * unit; fbase = group * 8; base16 = 2 * unit + wm; Au = p.A + (long long)group * 64 * p.Kpad;
        } else if (cg == 16) { group = 2 * sec + unit; fbase = group * 16; base16 = unit * 2; Au = p.A + (long long)group * p.C * p.Kpad; }
        else { group = sec; fbase = group * cg + unit * 16; base16 = 0; Au = p.A + ((long long)group * p.C + unit * AH_TM) * p.Kpad; }
        const int nk = spc * nchunks;

        __syncthreads();                        // previous unit's phase 2 done with stage / tcs
        // fold table slice of this unit, Tc[b][cls][8*fbase .. +128), DMA'd into LDS next to the first A stage:
        // instruction w (waves 0-4) carries classes 2w and 2w+1 (32 lanes x 16 B each)
        if (p.own_tc) {
            akgm_tc_slice(p, fbase, tcs, wave, lane, inv_b, mean_b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (visible behind the first K step's barrier)
        } else if (wave < 5) {
            const int cl = 2 * wave + (lane >> 5);
            if (cl < 9)
                __builtin_amdgcn_global_load_lds(
                    (const GLOBAL_AS void*)(p.Tc + ((long long)b * 9 + cl) * 8 * p.C + 8 * fbase + (lane & 31) * 4),
                    (LDS_AS void*)(reinterpret_cast<unsigned char*>(tcs) + wave * 1024), 16, 0, 0);
        }
        auto issue_A = [&](int st, int slot) {
            unsigned char* ab = aring + slot * AH_ASTAGE + wave * 1024;
            const bf16_t* src = Au + (long long)arow * p.Kpad + st * 64 + ajsw * 8;
            stage16(src, ab, lane);
            stage16(src + 32, ab + AH_TM * 64, lane);
        };
        AH_STAMP();
        issue_A(0, 0);
        f32x16_t acc[2][2];
        if (!ATT_LDS) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tm][tp][e] = 0.f;
        }

        int cch = 0, sp = 0;                     // halo chunk and stage index within the chunk period
        for (int s = 0; s < nk; ++s) {
            HC_WAIT(0);
            asm volatile("s_barrier" ::: "memory");
            if (ATT_LDS && s == 0) {
                // the fold table slice has landed: start at Tc[cls(pixel)][row] (registers 8q..8q+7 of a tile = the 8 sets
                // of feature 4*t32 + 2q + hh, original row order 8*feature + set)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const float* tc = tcs + cls[tp] * AH_TM;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float* t8 = tc + 8 * (4 * (wm * 2 + tm) + 2 * q + hh);
                            const float4 c0 = *reinterpret_cast<const float4*>(t8), c1 = *reinterpret_cast<const float4*>(t8 + 4);
                            acc[tm][tp][8 * q + 0] = c0.x; acc[tm][tp][8 * q + 1] = c0.y; acc[tm][tp][8 * q + 2] = c0.z; acc[tm][tp][8 * q + 3] = c0.w;
                            acc[tm][tp][8 * q + 4] = c1.x; acc[tm][tp][8 * q + 5] = c1.y; acc[tm][tp][8 * q + 6] = c1.z; acc[tm][tp][8 * q + 7] = c1.w;
                        }
                }
            }
            if (s + 1 < nk) issue_A(s + 1, (s + 1) & 1);
            const unsigned char* Hb = halo + cch * HC_HALO_BYTES;
            const unsigned char* Ab = aring + (s & 1) * AH_ASTAGE;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k16 = sp * 4 + j;
                // cg == 8: one k16 step = taps 2*k16 (lanes 0-31) and 2*k16+1 (lanes 32-63) x 8 channels
                int tap = (cg == 8) ? 2 * k16 : (k16 >> tshift);
                if (tap < 9) {
                    if (cg == 8) { tap += hh; tap = tap > 8 ? 8 : tap; }       // tap 9 has zero weights
                    const int ky = tap_ky(tap), kx = tap - 3 * ky;
                    const int sh = ky * hw + kx;
                    const int ch16 = (cg == 8) ? base16 : base16 + ((k16 & tshift) << 1) + hh;   // 16-byte chunk inside the 64-byte halo row
                    const int kch = (j & 1) * 2 + hh;                           // chunk inside the A half-stage row
                    const unsigned char* Ah = Ab + (j >> 1) * (AH_TM * 64);
                    bf16x8_t af[2], bfr[2];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        af[tm] = *reinterpret_cast<const bf16x8_t*>(Ah + a_off[tm] + ((kch ^ a_sw[tm]) << 4));
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) {
                        const int hp = hp0[tp] + sh;
                        bfr[tp] = *reinterpret_cast<const bf16x8_t*>(Hb + hp * 64 + ((ch16 ^ ((hp >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tp = 0; tp < 2; ++tp)
                            acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tp], acc[tm][tp], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (++sp == spc) { sp = 0; ++cch; }
        }

        AH_STAMP();
        // the unit's residual (HBM) is requested a phase ahead of its use
        int off2 = -1;
        {
            const int px = tid >> 1;
            if (px < nslots) {
                const int r = fdiv_small(px, inv_tw), c = px - r * tw;
                const int y = y0 + r, x = x0 + c;
                if (y < p.H && x < p.W) off2 = ((y + 1) * p.Wp + (x + 1)) * p.C + (tid & 1) * 8;
            }
        }
        uint4 rv = make_uint4(0, 0, 0, 0);
        if (off2 >= 0) rv = *reinterpret_cast<const uint4*>(p.res + (long long)b * p.res_bstride + off2 + fbase);
        // ---- phase 1: modulation sum in registers -> stage[px][16 features] (aliases the A ring) ---
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int px = wq * 64 + tp * 32 + (lane & 31);
            const float* tc = tcs + cls[tp] * AH_TM;
            float av[8];
            if (ATT_LDS) {
                const float4 a0 = *reinterpret_cast<const float4*>(attl + px * 8), a1 = *reinterpret_cast<const float4*>(attl + px * 8 + 4);
                av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) av[s] = att[ATT_LDS ? 0 : tp][s];
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int t32 = wm * 2 + tm;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int floc = 4 * t32 + 2 * q + hh;
                    float sa = 0.f, sb = 0.f;
                    if (!ATT_LDS) {
                        const float4 c0 = *reinterpret_cast<const float4*>(tc + 8 * floc);
                        const float4 c1 = *reinterpret_cast<const float4*>(tc + 8 * floc + 4);
                        const float tcv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                        for (int s = 0; s < 8; ++s) sb += av[s] * tcv[s];
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) sa += av[s] * acc[tm][tp][8 * q + s];
                    stage[px * AH_SL + floc] = valid[tp] ? rstd * (sa + sb) : 0.f;
                }
            }
        }
        __syncthreads();
        AH_STAMP();
        // ---- phase 2: one (pixel, 8 features) item per thread ----------------------------------------
        if (off2 >= 0) {
            const int px = tid >> 1, f8 = (tid & 1) * 8;
            const float4 a = *reinterpret_cast<const float4*>(&stage[px * AH_SL + f8]);
            const float4 d = *reinterpret_cast<const float4*>(&stage[px * AH_SL + f8 + 4]);
            const float v0[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
            const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv);
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vv[i] = silu_fast(v0[i]) + bf2f(rh[i]);
                s1 += vv[i]; s2 += vv[i] * vv[i];
            }
            *reinterpret_cast<uint4*>(p.out + (long long)b * p.out_bstride + off2 + fbase) = pack8_bf16(vv);
        }
    }
    AH_STAMP();
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    if (p.stats_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        __syncthreads();
        if (lane == 0) { scal[2 + wave * 2] = s1; scal[3 + wave * 2] = s2; }
        __syncthreads();
        if (tid == 0) {
            float t1 = 0.f, t2 = 0.f;
            for (int w = 0; w < 8; ++w) { t1 += scal[2 + w * 2]; t2 += scal[3 + w * 2]; }
            stat_add(p.stats_out, b, t1, t2);
        }
    }
}
