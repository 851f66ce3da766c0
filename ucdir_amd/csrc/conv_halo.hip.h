// 3x3 stride-1 convolution on the matrix cores with a 2-D pixel tile and an LDS halo (gfx950).
//
// Why: in the shifted-GEMM kernel (cgemm.hip.h) every K step (one tap x 64 channels) re-stages a
// fresh 16 KB activation tile and drains its loads before the next step; on MI355X that loop is
// load-LATENCY bound (~3000 cycles per step against ~1000 cycles of MFMA).  Here
//   * a workgroup (512 threads, 8 wave64) owns TM output features x a th x tw pixel tile
//     (th*tw <= 256, e.g. 16x16) of one sample;
//   * the (th+2) x (tw+2) x 64-channel input halo is staged into LDS ONCE per 64-channel chunk
//     (double buffered, prefetched a whole chunk = 9 K steps ahead) and all 9 taps read their B
//     fragments from it at a uniform shift: 6x less activation traffic than per-tap staging;
//   * K advances two taps x 32 channels per step (16 MFMAs per wave between barriers); the weight
//     tiles of the next step stream into a 2-deep ring while the current step is on the matrix
//     cores, with raw s_barrier and counted s_waitcnt so the next chunk's halo stays in flight;
//   * nearest-x2 Upsample + 3x3 conv (model/ucdir.py:53-60) runs here as four parity classes of
//     2x2 convolutions on the LOW-resolution grid (taps are just other shifts into the same halo,
//     weights pre-summed at pack time): 2.25x fewer FLOPs than convolving the upsampled image;
//   * LDS rows are 128 B with the 16-byte chunk XOR-swizzled by (row>>1)&7 on the source side.
// Epilogue = the same GroupNorm-fold / swish / residual / partial-sum / coalesced-store phase as
// cgemm.hip.h, run in passes of 128 pixels through an fp32 LDS stage.
#pragma once
#include "cgemm.hip.h"
#include "asmops.hip.h"

#define HC_THREADS 512
#define HC_HALO_PX 324
#define HC_BK 32                                   // channels per chunk (64-byte LDS rows)
#define HC_HALO_BYTES (HC_HALO_PX * HC_BK * 2)     // 20736

// LDS: [halo x2][A ring x3] (K loop) aliased by the fp32 epilogue stage; then scalars and the
// per-workgroup GroupNorm-fold table Tc[9][TM].  <= 80 KB so two workgroups share a CU: one
// workgroup's prologue / epilogue overlaps the other's matrix-core loop.
template <int TM>
__host__ __device__ constexpr int hc_kloop_bytes() { return 2 * HC_HALO_BYTES + 2 * (256 / TM) * TM * HC_BK * 2; }   // taps/step = 256/TM: 16 KB stage
template <int TM>
__host__ __device__ constexpr int hc_stage_bytes() { return ((TM == 128) ? 128 : 256) * (TM + 4) * 4; }
template <int TM>
__host__ __device__ constexpr int hc_scal_off() { return hc_kloop_bytes<TM>() > hc_stage_bytes<TM>() ? hc_kloop_bytes<TM>() : hc_stage_bytes<TM>(); }
template <int TM>
__host__ __device__ constexpr int hc_lds_bytes() { return hc_scal_off<TM>() + 128 + 9 * TM * 4; }

#ifdef UCDIR_TIMING
#define HC_STAMP(i) do { if (dbg_on) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HC_STAMP(i) do {} while (0)
#endif
#define HC_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// s_waitcnt needs an immediate: dispatch a wave-uniform count to the literal forms
__device__ __forceinline__ void hc_wait_vm(int n) {
    switch (n) {
        case 0: HC_WAIT(0); break;
        case 1: HC_WAIT(1); break;
        case 2: HC_WAIT(2); break;
        case 3: HC_WAIT(3); break;
        case 4: HC_WAIT(4); break;
        case 5: HC_WAIT(5); break;
        case 6: HC_WAIT(6); break;
        default: HC_WAIT(7); break;
    }
}

// MFMA column (lane & 31) -> pixel slot of a 32-pixel tile.  ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+ 32): with this permutation every group reads 16 CONSECUTIVE pixel slots (consecutive halo positions
// except where a tile row wraps) instead of pieces of two tile rows.  -1..2 % per launch.  (Making the wrap conflict-free too - halo
// pitch tw + 4 and a swizzle keyed on row * tw + column for widths that are no multiple of 16 - took the conflicts from 25 % to 2-4 %
// of the LDS cycles at 72^2 / 36^2 and the launches nowhere: 230.6 vs 228.4 us, 264.5 vs 265.1, 18^2 64.5 vs 62.0 - LDS conflicts are not
// what this kernel waits for; not kept, profiles/EXPERIMENTS.md.)
__device__ __forceinline__ int hc_px_perm(int l) {
    return l < 4 ? l : (l < 12 ? l + 12 : (l < 16 ? l - 8 : (l < 20 ? l + 8 : (l < 28 ? l - 12 : l))));
}

template <int TM, bool DUAL = false>
__global__ __launch_bounds__(HC_THREADS, 4) void conv3x3_halo_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WM = TM / 64;               // waves along rows
    constexpr int NPG = 8 / WM;               // pixel groups
    constexpr int TPW = 256 / NPG;            // pixels per wave (64 | 32)
    constexpr int NTP = TPW / 32;             // 32-pixel MFMA tiles per wave (2 | 1)
    constexpr int PXH = (TM == 128) ? 128 : 256;   // pixels per epilogue pass
    constexpr int TPS = 256 / TM;             // taps per K step (2 | 4): 16 MFMAs per wave between barriers
    constexpr int ASTAGE = TPS * TM * HC_BK * 2;  // 16384
    unsigned char* halo = smem;
    unsigned char* aring = smem + 2 * HC_HALO_BYTES;
    float* scal = reinterpret_cast<float*>(smem + hc_scal_off<TM>());
    float* tcs = scal + 32;                    // Tc[9][TM] = bias + Tb - mean*rstd*Tg
    float* stage = reinterpret_cast<float*>(smem);
    constexpr int SL = TM + 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: no waterfall loops around global_load_lds
    const int wm = wave / NPG, wq = wave % NPG;
    int lid;
    bool alt = false;                                            // tail filler: this workgroup computes the 1x1 res_conv
    {
        int nblk = gridDim.x, bid = blockIdx.x;
        if (!DUAL && p.alt_blocks) {
            const int nmain = nblk - p.alt_blocks;
            if (bid >= nmain) { alt = true; bid -= nmain; nblk = p.alt_blocks; } else nblk = nmain;
        }
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const bool fold = p.fold && !alt;
    const float* const bias = alt ? p.bias2 : p.bias;
    const int ksplit = (DUAL || alt) ? 1 : p.ksplit;                      // (the fused-res_conv variant has no register to spare;
                                                                          //  tail workgroups run their short K loop whole)
    int split = 0;
    if (ksplit > 1) { split = lid % ksplit; lid /= ksplit; }       // the splits of a tile are neighbours (same XCD)
    const int wgid = lid;
    int par = 0;
    if (p.up_phase) { par = lid & 3; lid >>= 2; }
    const int py = par >> 1, pxp = par & 1;
    const int rowtile = lid % p.rowtiles;
    int tq = lid / p.rowtiles;
    const int tx = tq % p.tiles_x; tq /= p.tiles_x;
    const int ty = tq % p.tiles_y;
    const int b = tq / p.tiles_y;
    const int th = p.th, tw = p.tw, hw = tw + 2;
    const int y0 = ty * th, x0 = tx * tw;
    const int hcount = (th + 2) * hw;
    const int nslots = th * tw;
    const float inv_hw = 1.0f / (float)hw, inv_tw = 1.0f / (float)tw;
#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 5);
    int dbg_n = 0;
#endif
    HC_STAMP(-1);                                               // kernel entry
    // Prologue order (round 2): (1) the fold table's operands (bias, Tb, Tg of this thread's 9 TM / 512 entries) and the input's
    // statistics slots (wave 0, one slot value per lane) are REQUESTED here, into registers; (2) the geometry below is computed
    // and the first halo / weight DMAs are issued while they fly; (3) a counted wait - VMEM returns in order, the DMAs behind
    // them stay in flight - then mean / rstd, one barrier, and the table goes to LDS.  One memory round trip before the first
    // MFMA instead of three (statistics -> table operands -> DMA): ~12 k -> ~5 k cycles.  (Issuing the DMA FIRST was measured in
    // round 2 and did not help: the ordinary loads then queue behind it in the in-order VMEM path.)
    constexpr int NE = (9 * TM + HC_THREADS - 1) / HC_THREADS;
    float te_b[NE], te_tb[NE], te_tg[NE];
    long long st_v = 0;
    const bool want_table = ksplit <= 1;                        // split-K workgroups have no epilogue: no table
    if (want_table) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = tid + e * HC_THREADS;
            te_b[e] = 0.f; te_tb[e] = 0.f; te_tg[e] = 0.f;
            if (i < 9 * TM) {
                const int cls = i / TM, fl = i - cls * TM;
                const int f = rowtile * TM + fl;
                if (bias) te_b[e] = bias[f];
                if (fold && f < p.nfeat) { te_tb[e] = p.Tb[(long long)cls * p.tab_ld + f]; te_tg[e] = p.Tg[(long long)cls * p.tab_ld + f]; }
            }
        }
        if (fold && wave == 0 && lane < 2 * UCDIR_STAT_SLOTS) {  // lane l: slot l / 2, (sum | sum of squares) = l & 1
            st_v = p.stats0[(long long)b * (2 * UCDIR_STAT_SLOTS) + lane];
            if (p.stats1) st_v += p.stats1[(long long)b * (2 * UCDIR_STAT_SLOTS) + lane];
        }
    }
    // second half of the prologue: called right behind the first DMA issue with this wave's DMA instruction count
    auto build_table = [&](int ndma) {
        hc_wait_vm(ndma);
        if (wave == 0) {
            float mean = 0.f, rstd = 1.f;
            if (fold) {
                long long v = st_v;                              // lanes of equal parity hold the 16 slots of one quantity
#pragma unroll
                for (int off = 2; off < 2 * UCDIR_STAT_SLOTS; off <<= 1) v += __shfl_xor(v, off);
                const long long q = __shfl(v, 1);
                mean_rstd(stat_val(v), stat_val(q), p.inv_count, mean, rstd);   // lane 0: v = sum, q = sum of squares
            }
            if (lane == 0) { scal[0] = mean; scal[1] = rstd; }
        }
        __syncthreads();
        // Tc[cls][f] = bias + Tb[cls] - mean*rstd*Tg[cls] for this workgroup's TM features (read in the epilogue)
        const float mr = scal[0] * scal[1];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = tid + e * HC_THREADS;
            if (i < 9 * TM) tcs[i] = te_b[e] + te_tb[e] - mr * te_tg[e];
        }
    };

    // ---- halo loader geometry: instruction k = i*8 + wave stages halo pixels 16k .. 16k+15 ------
    int hsrc[3], hjsw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int hp = (i * 8 + wave) * 16 + (lane >> 2);
        int hr = fdiv_small(hp, inv_hw), hc = hp - hr * hw;
        int gy = y0 + hr, gx = x0 + hc;
        gy = gy > p.H + 1 ? p.H + 1 : gy;
        gx = gx > p.W + 1 ? p.W + 1 : gx;
        hsrc[i] = (hp < hcount) ? (gy * p.Wp + gx) : -1;
        hjsw[i] = (lane & 3) ^ ((hp >> 2) & 3);
    }
    const int nchunks_all = p.cg / HC_BK;
    const int cbeg = (ksplit > 1) ? split * nchunks_all / ksplit : 0;            // this workgroup's channel chunks
    const int nchunks = (ksplit > 1) ? (split + 1) * nchunks_all / ksplit : nchunks_all;   // (end of the range)
    const int nh_min = ((hcount + 15) / 16) / 8;    // halo staging instructions every wave issues (some issue one more)
    // DUAL (64-row tiles only): a second accumulator set carries the block's res_conv as a 10th tap
    static_assert(!DUAL || TM == 64, "fused res_conv needs the 64-row tile");
    constexpr bool res_fused = DUAL;
    const int ntap = alt ? 1 : (p.up_phase ? 4 : (res_fused ? 10 : 9));   // tap 9 = the block's 1x1 res_conv on the centre pixel
    const int nsteps_c = (ntap + TPS - 1) / TPS;   // steps per chunk
    const int nk = (nchunks - cbeg) * nsteps_c;
    auto issue_halo = [&](int c, int buf) {
        int ch = c * HC_BK;
        const bf16_t* src; int ld;
        if (ch < p.c0) { src = p.B0 + (long long)b * p.b0_bstride; ld = p.ld0; }
        else { src = p.B1 + (long long)b * p.b1_bstride; ld = p.ld1; ch -= p.c0; }
        src += ch;
        unsigned char* hb = halo + buf * HC_HALO_BYTES;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if ((i * 8 + wave) * 16 < hcount) {            // wave-uniform
                if (hsrc[i] >= 0) stage16(src + hsrc[i] * ld + hjsw[i] * 8, hb + (i * 8 + wave) * 1024, lane);
            }
        }
    };
    // ---- weight tile loader: a stage holds TPS taps x TM rows x 64 B = 16 KB = 16 wave instructions, two per wave:
    // instruction k = 2*wave + j stages rows (k % (TM/16))*16 .. +15 of tap k / (TM/16)
    int arow_off[2], atap[2], adst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 2 * wave + j;
        const int row = (k % (TM / 16)) * 16 + (lane >> 2);
        const int ajsw = (lane & 3) ^ ((row >> 2) & 3);
        atap[j] = k / (TM / 16);
        arow_off[j] = (rowtile * TM + row) * (alt ? p.alt_a_ld : p.a_ld) + ajsw * 8;
        adst[j] = atap[j] * (TM * 64) + (k % (TM / 16)) * 1024;
    }
    const bf16_t* Abase = alt ? p.alt_A : p.A + (long long)par * p.a_gstride;
    const bool a_tiled = p.A_tiled != nullptr && !alt && !p.up_phase && !res_fused;
    const bf16_t* Atl = p.A_tiled + ((long long)rowtile * nchunks_all * nsteps_c) * (ASTAGE / 2) + (2 * wave) * 512 + lane * 8;
    auto issue_A = [&](int c, int u, int slot) {      // taps TPS*u .. TPS*u+TPS-1 (tail taps re-stage the last valid one)
        unsigned char* ab = aring + slot * ASTAGE;
        if (a_tiled) {                                 // pre-tiled image: this wave's two pieces are 2 KB of contiguous memory
            const bf16_t* src = Atl + (long long)(c * nsteps_c + u) * (ASTAGE / 2);
            stage16(src, ab + (2 * wave) * 1024, lane);
            stage16(src + 512, ab + (2 * wave + 1) * 1024, lane);
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int t = TPS * u + atap[j];
            t = t < ntap ? t : ntap - 1;
            stage16(Abase + t * p.cg + c * HC_BK + arow_off[j], ab + adst[j], lane);
        }
    };

    // ---- fragment geometry -----------------------------------------------------------------------
    int a_off[2], a_sw[2], hp0[NTP];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int row = wm * 64 + tm * 32 + (lane & 31);
        a_off[tm] = row * 64; a_sw[tm] = (row >> 2) & 3;
    }
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) {
        int slot = wq * TPW + tp * 32 + hc_px_perm(lane & 31);
        slot = slot < nslots ? slot : nslots - 1;
        const int r = fdiv_small(slot, inv_tw), c = slot - r * tw;
        hp0[tp] = r * hw + c;
    }

    f32x16_t acc[2][NTP];
    f32x16_t acc2[DUAL ? 2 : 1][NTP];           // res_conv(x) of the same rows / pixels (DUAL only)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[tm][tp][e] = 0.f; if (DUAL) acc2[DUAL ? tm : 0][tp][e] = 0.f; }

    HC_STAMP(0);
    // ---- K loop: s = c*9 + t -------------------------------------------------------------------------
    issue_halo(cbeg, cbeg & 1);
    issue_A(cbeg, 0, 0);
    if (want_table) {
        int ndma = 2;                               // this wave's instructions in flight: two weight pieces + its halo pieces
#pragma unroll
        for (int i = 0; i < 3; ++i) ndma += ((i * 8 + wave) * 16 < hcount) ? 1 : 0;
        build_table(ndma);
    }
    int c = cbeg, u = 0;                        // chunk and tap pair of the step being computed
    for (int s = 0; s < nk; ++s) {
        // everything older than the halo prefetch issued one step ago must have landed
        if (u == 1 && c + 1 < nchunks) hc_wait_vm(nh_min); else { HC_WAIT(0); }
        HC_STAMP(1);
        asm volatile("s_barrier" ::: "memory");
        HC_STAMP(2);
        {
            int cn = c, un = u + 1;
            if (un == nsteps_c) { un = 0; ++cn; }
            if (s + 1 < nk) issue_A(cn, un, (s + 1) & 1);
        }
        if (u == 0 && c + 1 < nchunks) issue_halo(c + 1, (c + 1) & 1);
        HC_STAMP(3);
        const unsigned char* Hb = halo + (c & 1) * HC_HALO_BYTES;
        // (A hand-pipelined version of this phase - inline-asm fragment reads one sub-step ahead, counted lgkmcnt, a second register
        // set, 125 VGPRs - was measured: +1..5 % SLOWER at 144^2 / 72^2 / 36^2.  With four waves per SIMD the other waves already
        // cover a wave's LDS round trips, and the scheduling barriers the counted waits need keep address arithmetic out from under
        // the MFMAs.)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int t = TPS * u + tt;
            if (t < ntap) {
                const unsigned char* Ab = aring + (s & 1) * ASTAGE + tt * (TM * HC_BK * 2);
                const bool is_res = DUAL && t == 9;                 // wave-uniform
                int sh;
                if (p.up_phase) sh = (py + (t >> 1)) * hw + (pxp + (t & 1));
                else if (is_res || alt) sh = hw + 1;
                else { const int ky = tap_ky(t); sh = ky * hw + (t - 3 * ky); }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int kch = kk * 2 + (lane >> 5);
                    bf16x8_t af[2], bfr[NTP];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        af[tm] = *reinterpret_cast<const bf16x8_t*>(Ab + a_off[tm] + ((kch ^ a_sw[tm]) << 4));
#pragma unroll
                    for (int tp = 0; tp < NTP; ++tp) {
                        const int hp = hp0[tp] + sh;
                        bfr[tp] = *reinterpret_cast<const bf16x8_t*>(Hb + hp * 64 + ((kch ^ ((hp >> 2) & 3)) << 4));
                    }
                    if (is_res) {
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                            for (int tp = 0; tp < NTP; ++tp)
                                acc2[DUAL ? tm : 0][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tp], acc2[DUAL ? tm : 0][tp], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                            for (int tp = 0; tp < NTP; ++tp)
                                acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tp], acc[tm][tp], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef UCDIR_TIMING
        asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][NTP - 1][15]));
#endif
        HC_STAMP(4);
        if (++u == nsteps_c) { u = 0; ++c; }
    }
    HC_STAMP(5);

    if (!DUAL && ksplit > 1) {
        // split-K: raw fp32 partial sums go to partial [wg][split][px][TM] straight from the accumulator layout (4 consecutive
        // features per lane); conv_splitk_finish_kernel sums them and runs the epilogue.  (Letting the workgroup that draws
        // a tile's last ticket do that here was measured: the device-scope release / acquire fences it needs write back and
        // invalidate the XCD's whole L2 on this chip - ~80 us per launch, B = 1 forward 2.8 -> 4.6 ms.)
        float* pw = p.partial + ((long long)wgid * ksplit + split) * (256 * TM);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) {
                const int px = wq * TPW + tp * 32 + hc_px_perm(lane & 31);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int f = wm * 64 + tm * 32 + 8 * rg + 4 * (lane >> 5);
                    *reinterpret_cast<float4*>(&pw[px * TM + f]) =
                        make_float4(acc[tm][tp][rg * 4 + 0], acc[tm][tp][rg * 4 + 1], acc[tm][tp][rg * 4 + 2], acc[tm][tp][rg * 4 + 3]);
                }
            }
        return;
    }

    // ---- epilogue in passes of PXH pixels ------------------------------------------------------------
    // (The register epilogue of the AKGM kernels - v_permlane32_swap hands a lane 8 consecutive rows of one pixel, no LDS stage,
    // no barriers - was built for this kernel too in round 2: bit-identical results, same-box A/B 29.10 vs 29.21 img/s for
    // the staged version.  Here a wave's 16-byte stores land 2 x C_out bytes apart per lane, the staged pass writes whole
    // pixel rows; kept staged.)
    const float rstd_s = scal[1];
    const float alpha = alt ? 1.0f : p.alpha * (fold ? rstd_s : 1.0f);
    const int act = alt ? 0 : p.act;
    const bf16_t* const resp = alt ? nullptr : p.res;
    bf16_t* const outp = alt ? p.out2 + (long long)b * p.out2_bstride : reinterpret_cast<bf16_t*>(p.out) + (long long)b * p.out_bstride + p.out_coff;
    const int out_ld = alt ? p.out2_ld : p.out_ld;
    const int fbase = rowtile * TM;
    constexpr int nf8 = TM / 8;
    float s1 = 0.f, s2 = 0.f;
    for (int pass = 0; pass < 256 / PXH; ++pass) {
        __syncthreads();
        if ((wq * TPW) / PXH == pass) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tp = 0; tp < NTP; ++tp) {
                    const int px = (wq * TPW) % PXH + tp * 32 + hc_px_perm(lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int f = wm * 64 + tm * 32 + 8 * rg + 4 * (lane >> 5);
                        *reinterpret_cast<float4*>(&stage[px * SL + f]) =
                            make_float4(acc[tm][tp][rg * 4 + 0], acc[tm][tp][rg * 4 + 1], acc[tm][tp][rg * 4 + 2], acc[tm][tp][rg * 4 + 3]);
                    }
                }
        }
        __syncthreads();
        for (int it = tid; it < PXH * nf8; it += HC_THREADS) {
            const int px = it / nf8;
            const int f8 = (it - px * nf8) * 8;
            const int slot = pass * PXH + px;
            if (slot >= nslots) continue;
            const int r = fdiv_small(slot, inv_tw), cc = slot - r * tw;
            const int y = y0 + r, x = x0 + cc;                 // 0-based valid coordinates
            if (y >= p.H || x >= p.W) continue;
            const int f = fbase + f8;
            if (f >= p.nfeat) continue;
            long long cp;
            int cls = 4;
            if (p.up_phase) cp = (long long)(2 * y + py + 1) * (2 * p.W + 2) + (2 * x + pxp + 1);
            else {
                cp = (long long)(y + 1) * p.Wp + (x + 1);
                cls = (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1));
            }
            float v[8];
            {
                const float4 a = *reinterpret_cast<const float4*>(&stage[px * SL + f8]);
                const float4 d = *reinterpret_cast<const float4*>(&stage[px * SL + f8 + 4]);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= alpha;
            {
                const float4 t0 = *reinterpret_cast<const float4*>(&tcs[cls * TM + f8]);
                const float4 t1 = *reinterpret_cast<const float4*>(&tcs[cls * TM + f8 + 4]);
                v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
            }
            if (act == 1) {                        // one uniform branch per item, not one per element
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = silu_fast(v[i]);
            } else if (act == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(0.2f * v[i], v[i]);
            }
            if (resp) {
                const uint4 rv = *reinterpret_cast<const uint4*>(resp + (long long)b * p.res_bstride + cp * p.res_ld + p.res_coff + f);
                const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bf2f(rh[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
            if (p.out_nchw) {
                if (y < p.crop_h && x < p.crop_w) {
                    float* op = reinterpret_cast<float*>(p.out) + (((long long)b * p.nfeat + f) * p.crop_h + y) * p.crop_w + x;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (f + i < p.nfeat) op[(long long)i * p.crop_h * p.crop_w] = v[i];
                }
            } else {
                *reinterpret_cast<uint4*>(outp + cp * out_ld + f) = pack8_bf16(v);
            }
        }
    }
    if (res_fused) {
        // ---- second output: res = res_conv(x) + bias2 (no GroupNorm fold, no activation, no statistics), bf16 NHWC ----
        for (int pass = 0; pass < 256 / PXH; ++pass) {
            __syncthreads();
            if ((wq * TPW) / PXH == pass) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tp = 0; tp < NTP; ++tp) {
                        const int px = (wq * TPW) % PXH + tp * 32 + hc_px_perm(lane & 31);
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int f = wm * 64 + tm * 32 + 8 * rg + 4 * (lane >> 5);
                            const f32x16_t& a2 = acc2[DUAL ? tm : 0][tp];
                            *reinterpret_cast<float4*>(&stage[px * SL + f]) = make_float4(a2[rg * 4 + 0], a2[rg * 4 + 1], a2[rg * 4 + 2], a2[rg * 4 + 3]);
                        }
                    }
            }
            __syncthreads();
            for (int it = tid; it < PXH * nf8; it += HC_THREADS) {
                const int px = it / nf8;
                const int f8 = (it - px * nf8) * 8;
                const int slot = pass * PXH + px;
                if (slot >= nslots) continue;
                const int r = fdiv_small(slot, inv_tw), cc = slot - r * tw;
                const int y = y0 + r, x = x0 + cc;
                if (y >= p.H || x >= p.W) continue;
                const int f = fbase + f8;
                if (f >= p.nfeat) continue;
                const float4 a = *reinterpret_cast<const float4*>(&stage[px * SL + f8]);
                const float4 d = *reinterpret_cast<const float4*>(&stage[px * SL + f8 + 4]);
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias2 + f), b1 = *reinterpret_cast<const float4*>(p.bias2 + f + 4);
                const float v[8] = {a.x + b0.x, a.y + b0.y, a.z + b0.z, a.w + b0.w, d.x + b1.x, d.y + b1.y, d.z + b1.z, d.w + b1.w};
                *reinterpret_cast<uint4*>(p.out2 + (long long)b * p.out2_bstride + ((long long)(y + 1) * p.Wp + (x + 1)) * p.out2_ld + f) = pack8_bf16(v);
            }
        }
    }
    HC_STAMP(6);
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    if (p.stats_out && !alt) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        __syncthreads();
        if (lane == 0) { scal[2 + wave * 2] = s1; scal[3 + wave * 2] = s2; }
        __syncthreads();
        if (tid == 0) {
            float t1 = 0.f, t2s = 0.f;
            for (int w = 0; w < 8; ++w) { t1 += scal[2 + w * 2]; t2s += scal[3 + w * 2]; }
            stat_add(p.stats_out, b, t1, t2s);
        }
    }
}


// Second half of a split-K conv3x3_halo launch: sums the ksplit partial tiles of every output element in split order (fixed:
// bit-reproducible) and applies conv3x3_halo_kernel's epilogue: GroupNorm fold, bias, activation, residual, statistics of
// the output, bf16 NHWC store.  grid = (blocks per sample, B); a thread owns 8 features of one output pixel.
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const GemmP p, int TM) {
    __shared__ float sh[20];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid == 0) {
        float mean = 0.f, rstd = 1.f;
        if (p.fold) {
            double S, Q;
            stat_read(p.stats0, p.stats1, b, S, Q);
            mean_rstd(S, Q, p.inv_count, mean, rstd);
        }
        sh[0] = mean; sh[1] = rstd;
    }
    __syncthreads();
    const float mean = sh[0], rstd = sh[1], mr = mean * rstd;
    const float alpha = p.alpha * (p.fold ? rstd : 1.0f);
    const int Ho = p.up_phase ? 2 * p.H : p.H, Wo = p.up_phase ? 2 * p.W : p.W;
    const int nf8 = p.nfeat / 8;
    const int items = Ho * Wo * nf8;
    const long long tile = 256LL * TM;
    float s1 = 0.f, s2 = 0.f;
    for (int it = blockIdx.x * 256 + tid; it < items; it += gridDim.x * 256) {
        const int pix = it / nf8, f = (it - pix * nf8) * 8;
        const int Y = pix / Wo, X = pix - Y * Wo;
        int y = Y, x = X, par = 0;
        if (p.up_phase) { par = (Y & 1) * 2 + (X & 1); y = Y >> 1; x = X >> 1; }
        const int ty = y / p.th, tx = x / p.tw;
        const int slot = (y - ty * p.th) * p.tw + (x - tx * p.tw);
        const int rowtile = f / TM, fl = f - rowtile * TM;
        long long wg = (((long long)b * p.tiles_y + ty) * p.tiles_x + tx) * p.rowtiles + rowtile;
        if (p.up_phase) wg = wg * 4 + par;
        const float* pb = p.partial + wg * p.ksplit * tile + (long long)slot * TM + fl;
        // four splits per round trip (the loads of a group are independent: issued together, then added in split order)
        float4 a = *reinterpret_cast<const float4*>(pb), d = *reinterpret_cast<const float4*>(pb + 4);
        int s = 1;
        for (; s + 3 < p.ksplit; s += 4) {
            float4 aa[4], dd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                aa[j] = *reinterpret_cast<const float4*>(pb + (s + j) * tile);
                dd[j] = *reinterpret_cast<const float4*>(pb + (s + j) * tile + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a.x += aa[j].x; a.y += aa[j].y; a.z += aa[j].z; a.w += aa[j].w;
                d.x += dd[j].x; d.y += dd[j].y; d.z += dd[j].z; d.w += dd[j].w;
            }
        }
        for (; s < p.ksplit; ++s) {
            const float4 a1 = *reinterpret_cast<const float4*>(pb + s * tile), d1 = *reinterpret_cast<const float4*>(pb + s * tile + 4);
            a.x += a1.x; a.y += a1.y; a.z += a1.z; a.w += a1.w; d.x += d1.x; d.y += d1.y; d.z += d1.z; d.w += d1.w;
        }
        float v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + f), b1 = *reinterpret_cast<const float4*>(p.bias + f + 4);
            t[0] = b0.x; t[1] = b0.y; t[2] = b0.z; t[3] = b0.w; t[4] = b1.x; t[5] = b1.y; t[6] = b1.z; t[7] = b1.w;
        }
        if (p.fold) {
            const int cls = (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1));
            const float* tb = p.Tb + (long long)cls * p.tab_ld + f;
            const float* tg = p.Tg + (long long)cls * p.tab_ld + f;
            const float4 b0 = *reinterpret_cast<const float4*>(tb), b1 = *reinterpret_cast<const float4*>(tb + 4);
            const float4 g0 = *reinterpret_cast<const float4*>(tg), g1 = *reinterpret_cast<const float4*>(tg + 4);
            t[0] += b0.x - mr * g0.x; t[1] += b0.y - mr * g0.y; t[2] += b0.z - mr * g0.z; t[3] += b0.w - mr * g0.w;
            t[4] += b1.x - mr * g1.x; t[5] += b1.y - mr * g1.y; t[6] += b1.z - mr * g1.z; t[7] += b1.w - mr * g1.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + t[i];
        if (p.act == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = silu_fast(v[i]);
        } else if (p.act == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(0.2f * v[i], v[i]);
        }
        const long long cp = (long long)(Y + 1) * (Wo + 2) + (X + 1);
        if (p.res) {
            const uint4 rv = *reinterpret_cast<const uint4*>(p.res + (long long)b * p.res_bstride + cp * p.res_ld + p.res_coff + f);
            const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bf2f(rh[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (long long)b * p.out_bstride + cp * p.out_ld + p.out_coff + f) = pack8_bf16(v);
    }
    if (p.stats_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        if ((tid & 63) == 0) { sh[2 + (tid >> 6) * 2] = s1; sh[3 + (tid >> 6) * 2] = s2; }
        __syncthreads();
        if (tid == 0) stat_add(p.stats_out, b, sh[2] + sh[4] + sh[6] + sh[8], sh[3] + sh[5] + sh[7] + sh[9]);
    }
}
