// AKGM tail for 8 / 16 channels per group (C = 64 / 128: the 288^2 and 144^2 levels) with the unit's
// weights RESIDENT in LDS (gfx950).  Reference: model/ucdir.py:129-140.
//
// At these levels a unit (128 rows x 256 pixels) has only K = 72 / 144, i.e. 20 / 36 MFMAs per wave:
// streaming its weights through a ring (akgm_halo.hip.h) spends the time in barriers and exposed
// load latency.  Here the packed weights are stored in HBM as the LDS image itself,
//     [unit][k16 step j][k half][128 rows][8 bf16]            (20 KB / 36 KB per unit)
// and DMA'd linearly (1 KB pieces) together with the halo before anything else happens:
//   cg = 8 : both units of the workgroup (2 x 20 KB) up front; no barrier inside the K loops;
//   cg = 16: unit 0 up front, unit 1 streams in underneath unit 0's epilogue (instantiated, not dispatched: the ring
//            kernel of akgm_halo.hip.h measured faster there).
// A fragments are conflict-free without a swizzle (32 lanes read 32 consecutive 16-byte slots).
// Epilogue WITHOUT LDS and WITHOUT barriers: after the modulation sum a lane holds the lower (lane < 32) or upper
// (lane >= 32) feature pair of each 4-feature group for its two pixels; one v_permlane32_swap per value hands
// lanes 0-31 all eight features of pixel tp = 0 and lanes 32-63 all eight of pixel tp = 1 (lane L <-> pixel
// 64 wq + L), i.e. exactly one 16-byte residual load and one 16-byte store per lane.  The fp32 stage, its two
// barriers per unit and the LDS round trip are gone: after the single barrier that publishes the DMA'd tiles the
// eight waves of a workgroup never synchronise again (cg = 8).
#pragma once
#include "akgm_halo.hip.h"

template <int CG>
struct AkPre {
    static constexpr int NK16 = (CG == 8) ? 5 : 9;            // 16-wide k steps per unit (cg 8: two taps x 8 ch per step, tap 9 = 0)
    static constexpr int A_UNIT = NK16 * 4096;                // bytes
    static constexpr int NA = (CG == 8) ? 2 : 1;              // resident units
    static constexpr int OFF_A = HC_HALO_BYTES;
    static constexpr int OFF_SCAL = OFF_A + NA * A_UNIT;
    static constexpr int OFF_TCS = OFF_SCAL + 128;
    static constexpr int OFF_ATT = OFF_TCS + NA * 9 * 128 * 4; // [256 px][8] modulation weights G * attw
    static constexpr int LDS = OFF_ATT + 256 * 8 * 4;         // 79,232 / 70,528: two workgroups per CU
};

template <int CG>
__global__ __launch_bounds__(HC_THREADS, 4) void akgm_pre_kernel(const AkgmHP p) {
    using L = AkPre<CG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* abuf = smem + L::OFF_A;
    float* scal = reinterpret_cast<float*>(smem + L::OFF_SCAL);
    float* tcs = reinterpret_cast<float*>(smem + L::OFF_TCS);          // [NA][9][128]
    float* attl = reinterpret_cast<float*>(smem + L::OFF_ATT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wq = wave & 3, hh = lane >> 5;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nsec = p.C / 32;                                  // 32-channel chunks: 2 (C = 64) / 4 (C = 128)
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
    const int unit0 = 2 * sec;                                  // first of this workgroup's two units
#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 5);
    int dbg_n = 0;
#endif
    AH_STAMP();                                                 // kernel entry

    // ---- everything the workgroup needs up front goes into flight now ---------------------------------
    {
        const bf16_t* hb = p.h + (long long)b * p.h_bstride;
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
                    stage16(hb + (long long)(gy * p.Wp + gx) * p.C + sec * 32 + j * 8, halo + (i * 8 + wave) * 1024, lane);
                }
            }
        }
    }
    auto issue_A = [&](int unit, int p0, int p1) {                       // linear copy of pieces [p0, p1) of the LDS image
        int lo = lane * 16;
        asm volatile("" : "+v"(lo));          // addresses are formed here, not hoisted to the kernel top and spilled
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.A) + (long long)unit * L::A_UNIT + lo;
#pragma unroll
        for (int i = 0; i < (L::NA * L::A_UNIT / 1024 + 7) / 8; ++i) {
            const int pi = p0 + i * 8 + wave;
            if (pi < p1) stage16(reinterpret_cast<const bf16_t*>(src + pi * 1024), abuf + pi * 1024, lane);
        }
    };
    auto issue_Tc = [&](int fbase, float* dst) {                         // Tc[b][cls][8*fbase .. +128), two classes per instruction
        if (wave < 5) {
            const int cl = 2 * wave + (lane >> 5);
            if (cl < 9)
                __builtin_amdgcn_global_load_lds(
                    (const GLOBAL_AS void*)(p.Tc + ((long long)b * 9 + cl) * 8 * p.C + 8 * fbase + (lane & 31) * 4),
                    (LDS_AS void*)(reinterpret_cast<unsigned char*>(dst) + wave * 1024), 16, 0, 0);
        }
    };
    issue_A(unit0, 0, L::NA * (L::A_UNIT / 1024));
    float rstd, mean_b = 0.f;
    if (p.own_tc) stat_mean_rstd_wave(p.stats, b, p.inv_count, lane, mean_b, rstd);   // (no akgm_tc_kernel launch in front: AkgmHP::own_tc)
    else rstd = p.ms[2 * b + 1];
    const float inv_b = 1.0f / rstd;
    auto tc_slice = [&](int fbase, float* dst) {
        if (p.own_tc) akgm_tc_slice(p, fbase, dst, wave, lane, inv_b, mean_b); else issue_Tc(fbase, dst);
    };
    tc_slice(unit0 * 16, tcs);
    if (CG == 8) tc_slice(unit0 * 16 + 16, tcs + 9 * 128);

    // ---- per-lane pixel constants (K loop / phase 1) ---------------------------------------------------
    int hp0[2], cls[2];                                                // cls: border class, or -1 for a pixel outside the image / tile
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
        int slot = wq * 64 + tp * 32 + (lane & 31);
        const bool inb = slot < nslots;
        slot = inb ? slot : nslots - 1;
        const int r = fdiv_small(slot, inv_tw), c = slot - r * tw;
        hp0[tp] = r * hw + c;
        int y = y0 + r, x = x0 + c;
        const bool vld = inb && y < p.H && x < p.W;
        y = y < p.H ? y : p.H - 1; x = x < p.W ? x : p.W - 1;
        cls[tp] = vld ? (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1)) : -1;
        const float* gp = p.G + (long long)b * p.g_bstride + ((long long)y * p.W + x) * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        const float* aw = p.attw + b * 8;
        if (wm == 0) {                       // the two row halves see the same pixels: one of them publishes G * attw
            float* ap = attl + (wq * 64 + tp * 32 + (lane & 31)) * 8;
            *reinterpret_cast<float4*>(ap) = make_float4(g0.x * aw[0], g0.y * aw[1], g0.z * aw[2], g0.w * aw[3]);
            *reinterpret_cast<float4*>(ap + 4) = make_float4(g1.x * aw[4], g1.y * aw[5], g1.z * aw[6], g1.w * aw[7]);
        }
    }
    // ---- store item of this lane: pixel 64 wq + lane, features 8 wm .. 8 wm + 7 of the unit ----------------
    int off2;                                                          // element offset inside the sample, -1 = nothing to store
    {
        const int px2 = wq * 64 + lane;
        const int r = fdiv_small(px2, inv_tw), c = px2 - r * tw;
        const int y = y0 + r, x = x0 + c;
        off2 = (px2 < nslots && y < p.H && x < p.W) ? ((y + 1) * p.Wp + (x + 1)) * p.C + wm * 8 : -1;
    }
    const int a_lane = (hh * 128 + wm * 64 + (lane & 31)) * 16;        // + j*4096 + tm*512
    // the residual (HBM, 16 bytes per lane and unit) is requested a phase EARLY: unit 0's under the weight / halo DMA,
    // unit 1's under unit 0's epilogue.  Loaded where it is used, its full HBM latency sat on every unit's critical path.
    const bf16_t* resp = p.res + (long long)b * p.res_bstride + (off2 >= 0 ? off2 : 0) + unit0 * 16;
    uint4 rv_cur = off2 >= 0 ? *reinterpret_cast<const uint4*>(resp) : make_uint4(0, 0, 0, 0);

    AH_STAMP();
    HC_WAIT(0);
    AH_STAMP();
    __syncthreads();
    AH_STAMP();

    float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
    for (int u = 0; u < 2; ++u) {
        const int fbase = (unit0 + u) * 16;
        const unsigned char* Au = abuf + ((CG == 8) ? u * L::A_UNIT : 0);
        const float* tcu = tcs + ((CG == 8) ? u * 9 * 128 : 0);

        // accumulators start at the fold constants Tc[cls(pixel)][row] (akgm_tc_kernel: already divided by rstd) instead
        // of zero: registers 0..15 of a tile = features floc0, floc0 + 1 x 8 sets = 16 consecutive floats of the table
        f32x16_t acc[2][2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const float* tc = tcu + (cls[tp] < 0 ? 0 : cls[tp]) * 128;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const float* t16 = tc + 8 * (4 * (wm * 2 + tm) + 2 * hh);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 c4 = *reinterpret_cast<const float4*>(t16 + 4 * g);
                    acc[tm][tp][4 * g + 0] = c4.x; acc[tm][tp][4 * g + 1] = c4.y; acc[tm][tp][4 * g + 2] = c4.z; acc[tm][tp][4 * g + 3] = c4.w;
                }
            }
        }

        asm volatile("" : "+v"(hp0[0]), "+v"(hp0[1]));                  // no hoisting of per-step addresses out of the unit loop (spills)
        const int ch16 = 2 * u + ((CG == 8) ? wm : hh);                // 16-byte chunk of the halo row this lane's k half reads
        __builtin_amdgcn_s_setprio(1);
        bf16x8_t af[2][2], bfr[2][2];                                   // fragments of step j live in buffer j & 1
        auto load_frags = [&](int j, int buf) {
            int sh;
            if (CG == 8) {
                const int t0 = 2 * j, t1 = (2 * j + 1 > 8) ? 8 : 2 * j + 1;     // tap 9 has zero weights: read tap 8's pixels
                const int sh0 = (t0 / 3) * hw + (t0 % 3), sh1 = (t1 / 3) * hw + (t1 % 3);
                sh = hh ? sh1 : sh0;
            } else {
                sh = (j / 3) * hw + (j % 3);
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                af[buf][tm] = *reinterpret_cast<const bf16x8_t*>(Au + j * 4096 + tm * 512 + a_lane);
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int hp = hp0[tp] + sh;
                bfr[buf][tp] = *reinterpret_cast<const bf16x8_t*>(halo + hp * 64 + ((ch16 ^ ((hp >> 2) & 3)) << 4));
            }
        };
        __builtin_amdgcn_sched_barrier(0);          // the table reads above stay out of the pipelined region below
        load_frags(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int j = 0; j < L::NK16; ++j) {
            if (j + 1 < L::NK16) load_frags(j + 1, (j + 1) & 1);       // next step's fragments fly under this step's MFMAs
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
                    acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j & 1][tm], bfr[j & 1][tp], acc[tm][tp], 0, 0, 0);
            if (j + 1 < L::NK16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);    // 4 LDS reads (step j + 1) ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                         // ... then the 4 MFMAs of step j
        }
        __builtin_amdgcn_s_setprio(0);
        AH_STAMP();

        uint4 rv_next = make_uint4(0, 0, 0, 0);
        if (u == 0 && off2 >= 0) rv_next = *reinterpret_cast<const uint4*>(resp + 16);
        if (CG == 16 && u == 0) {            // unit 1's weights / fold table replace unit 0's: every wave is done reading them
            __syncthreads();
            issue_A(unit0 + 1, 0, 36);
            tc_slice(fbase + 16, tcs);
        }
        // ---- modulation sum in registers: vq[tm][q][tp] = feature 8 wm + 4 tm + 2 hh + q of pixel tp (pack_akgm_pre) --
        float vq[2][2][2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int px = wq * 64 + tp * 32 + (lane & 31);
            float att[8];
            {
                const float4 a0 = *reinterpret_cast<const float4*>(attl + px * 8), a1 = *reinterpret_cast<const float4*>(attl + px * 8 + 4);
                att[0] = a0.x; att[1] = a0.y; att[2] = a0.z; att[3] = a0.w; att[4] = a1.x; att[5] = a1.y; att[6] = a1.z; att[7] = a1.w;
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float sa = 0.f;
#pragma unroll
                    for (int s = 0; s < 8; ++s) sa += att[s] * acc[tm][tp][8 * q + s];
                    vq[tm][q][tp] = rstd * sa;
                }
        }
        AH_STAMP();
        // ---- half-wave exchange: lanes 0-31 keep pixel tp = 0 and receive its features 4 tm + 2 + q from lanes 32-63, which
        // keep pixel tp = 1 and receive its features 4 tm + q -----------------------------------------------------------
        float o8[8];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float lo = vq[tm][q][0], hi = vq[tm][q][1];
                permlane32_swap(lo, hi);
                o8[4 * tm + q] = lo;                    // features 4 t32 + {0, 1} (q) come from lanes 0-31, ...
                o8[4 * tm + 2 + q] = hi;                // ... features 4 t32 + 2 + {0, 1} from lanes 32-63
            }
        // ---- swish + residual + statistics + store, 16 bytes per lane -------------------------------------------------
        if (off2 >= 0) {
            const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv_cur);
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vv[i] = silu_fast(o8[i]) + bf2f(rh[i]);
                s1 += vv[i]; s2 += vv[i] * vv[i];
            }
            *reinterpret_cast<uint4*>(p.out + (long long)b * p.out_bstride + off2 + fbase) = pack8_bf16(vv);
        }
        AH_STAMP();
        rv_cur = rv_next;
        if (CG == 16 && u == 0) { HC_WAIT(0); __syncthreads(); }
    }
#ifdef UCDIR_TIMING
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    if (p.stats_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        if (lane == 0) { scal[2 + wave * 2] = s1; scal[3 + wave * 2] = s2; }
        __syncthreads();
        if (tid == 0) {
            float t1 = 0.f, t2 = 0.f;
            for (int w = 0; w < 8; ++w) { t1 += scal[2 + w * 2]; t2 += scal[3 + w * 2]; }
            stat_add(p.stats_out, b, t1, t2);
        }
    }
}
