// Implicit-GEMM convolution / GEMM kernel for gfx950 (MI355X).
//
//   D[row][col] = sum_k A[row][k] * Bm[col][k]          bf16 operands, fp32 accumulate on
//   v_mfma_f32_32x32x16_bf16; rows = output features, cols = pixel positions (or tokens).
//
// One 256-thread workgroup (4 wave64, 2x2) owns a TM x 128 output tile; K advances in steps of
// 64 (8 x 16-byte chunks).  Both operand tiles are staged into LDS with global_load_lds
// (16 B/lane, no VGPR round trip), double buffered: the loads of step k+1 are in flight while
// step k is on the matrix cores.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled
// with (row>>1)&7 *on the source address* (global_load_lds writes lane-linear), which makes
// the 32x32x16 fragment ds_read_b128 pattern bank-conflict free.
//
// A 3x3 convolution is the same GEMM with K = 9 taps x C_in: because activations live in a
// zero-bordered NHWC layout, tap (dy,dx) of column position p is simply row p + dy*Wp + dx of
// the same matrix — no predication, no im2col buffer.  Stride-2 and nearest-x2 convolutions
// only change that row mapping.  GroupNorm(1 group) in front of a convolution is folded into
// the epilogue:   conv(GN(x)) = rstd * conv_{W*gamma}(x) + Tb[cls] - mean*rstd*Tg[cls]
// with 9 border classes (which taps fall on the zero border), so the kernel reads raw x.
#pragma once
#include "common.h"
#include "asmops.hip.h"

#define CG_TP 128
#define CG_BK 64
#define CG_THREADS 256

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at a wave-uniform base
// (written lane-linearly: any swizzle goes on the SOURCE address)
__device__ __forceinline__ void stage16(const bf16_t* gsrc, unsigned char* lds_wave_base, int lane) {
    (void)lane;
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int tap_ky(int tap) { return (tap * 11) >> 5; }   // tap / 3 for 0..8

template <int TM>
__host__ __device__ constexpr int cgemm_kbuf_bytes() { return 2 * TM * 128 + 2 * CG_TP * 128; }

// dynamic LDS size for a launch
static inline size_t cgemm_lds_bytes(int TM, int epi, int groups_per_wg) {
    size_t kb = 2 * (size_t)TM * 128 + 2 * CG_TP * 128;
    if (epi == EPI_STD) {
        size_t st = (size_t)CG_TP * (TM + 4) * 4;
        return (kb > st ? kb : st) + 64;
    }
    size_t nf = (size_t)groups_per_wg * TM / 8;
    return kb + (size_t)CG_TP * (nf + 4) * 4 + 64;
}

// MODE: column-position -> input-row mapping (compile time so the per-K-step staging code is lean)
enum { MODE_S1 = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_PLAIN = 3, MODE_S1C = 4 };

template <int TM, int EPI, int MODE>
__global__ __launch_bounds__(CG_THREADS, 2) void cgemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TMT = TM / 64;            // 32-row MFMA tiles per wave along rows
    constexpr int KBUF = cgemm_kbuf_bytes<TM>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef UCDIR_TIMING
    int cg_n = 0;
#define CG_STAMP() do { if (p.dbg && blockIdx.x == gridDim.x / 2 && tid == 0 && cg_n < 60) p.dbg[cg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CG_STAMP() do {} while (0)
#endif
    CG_STAMP();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: no waterfall loops around global_load_lds
    const int wm = wave >> 1, wp = wave & 1;

    // ---- XCD-aware workgroup -> (batch, column tile, row tile) mapping ----------------------
    // Hardware places block b on XCD b%8; give each XCD a contiguous run of logical tiles so the
    // workgroups that share an activation slab (all row tiles of a column tile, and neighbouring
    // column tiles whose 3x3 halos overlap) hit the same private L2.
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int rowtile = lid % p.rowtiles;
    const int tcol = (lid / p.rowtiles) % p.tiles;
    const int b = lid / (p.rowtiles * p.tiles);

    // LDS carve (single dynamic array; see cdna guide G17)
    float* stage;
    int SL, NF;
    if (EPI == EPI_STD) { stage = reinterpret_cast<float*>(smem); NF = TM; }
    else { stage = reinterpret_cast<float*>(smem + KBUF); NF = p.groups_per_wg * (TM / 8); }
    SL = NF + 4;
    size_t scal_off = (EPI == EPI_STD)
        ? ((size_t)KBUF > (size_t)CG_TP * (TM + 4) * 4 ? (size_t)KBUF : (size_t)CG_TP * (TM + 4) * 4)
        : (size_t)KBUF + (size_t)CG_TP * SL * 4;
    float* scal = reinterpret_cast<float*>(smem + scal_off);     // [0]=mean [1]=rstd [2..9]=reduce scratch

    if (tid == 0) {
        float mean = 0.f, rstd = 1.f;
        if (p.fold) {
            double S, Q;
            stat_read(p.stats0, p.stats1, b, S, Q);
            double m = S * p.inv_count;
            double var = Q * p.inv_count - m * m;
            if (var < 0) var = 0;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + 1e-5));
        }
        scal[0] = mean; scal[1] = rstd;
    }

    // ---- per-lane loader geometry --------------------------------------------------------------
    const int lrow = lane >> 3;
    const int jsw = (lane & 7) ^ (((wave & 1) << 2) | (lrow >> 1));   // logical chunk held by this lane's slot
    const int plast = p.p0 + p.pn - 1;
    int rowoff[4], byo[4], bxo[4];      // rowoff: input row index of the centre / first tap
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rb = (i * 4 + wave) * 8 + lrow;
        int cp = p.p0 + tcol * CG_TP + rb;
        cp = cp < plast ? cp : plast;
        rowoff[i] = cp; byo[i] = 0; bxo[i] = 0;
        if (MODE == MODE_DOWN || MODE == MODE_UP || MODE == MODE_S1C) {
            int y = cp / p.Wp, x = cp - y * p.Wp;
            y = y < 1 ? 1 : (y > p.H ? p.H : y);
            x = x < 1 ? 1 : (x > p.W ? p.W : x);
            byo[i] = y; bxo[i] = x;
            if (MODE == MODE_S1C) rowoff[i] = (y - 1) * p.W + (x - 1);
            if (MODE == MODE_DOWN) rowoff[i] = 2 * (y - 1) * p.Wpi + 2 * (x - 1);
        }
    }

    f32x16_t acc[TMT][2];

    const int ngroups_loop = (EPI == EPI_AKGM) ? p.groups_per_wg : 1;
    for (int gi = 0; gi < ngroups_loop; ++gi) {
        int g = 0, rtg = rowtile;
        if (EPI == EPI_AKGM) {
            const int rt_per_group = (8 * p.cg) / TM;          // rows per group = 8*cg
            if (p.groups_per_wg > 1) { g = rowtile * p.groups_per_wg + gi; rtg = 0; }
            else { g = rowtile / rt_per_group; rtg = rowtile - g * rt_per_group; }
            __syncthreads();                                    // previous group's K buffers fully consumed
        }
        const int gbase = g * ((EPI == EPI_AKGM) ? p.cg : 0);
        const bf16_t* Abase = p.A + (long long)b * p.a_bstride + (long long)g * p.a_gstride;
        int arow_off[TM / 32];
#pragma unroll
        for (int i = 0; i < TM / 32; ++i) {
            int ra = rtg * TM + (i * 4 + wave) * 8 + lrow;
            ra = ra < p.a_rows ? ra : p.a_rows - 1;
            arow_off[i] = ra * p.a_ld + jsw * 8;
        }
#pragma unroll
        for (int tm = 0; tm < TMT; ++tm)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[tm][tp][e] = 0.f;

        // running (tap, channel) of the K step being *issued* (dense case: cpt % 8 == 0)
        int is_tap = 0, is_cc = 0;

        auto issue = [&](int ks, int buf) {
            unsigned char* Ab = smem + buf * (TM * 128);
            unsigned char* Bb = smem + 2 * TM * 128 + buf * (CG_TP * 128);
#pragma unroll
            for (int i = 0; i < TM / 32; ++i)
                stage16(Abase + arow_off[i] + ks * CG_BK, Ab + (i * 4 + wave) * 1024, lane);
            if (EPI == EPI_AKGM && p.cpt < 8) {
                // grouped conv with < 64 channels per group: the 8 chunks of a K step belong to
                // different taps, so tap and channel are per lane
                int q = ks * 8 + jsw;
                int tap = q >> p.cpt_shift, ch = (q & (p.cpt - 1)) * 8;
                if (tap >= 9) { tap = 4; ch = 0; }                    // zero-weight padding chunk
                const int ky = tap_ky(tap), kx = tap - 3 * ky;
                const bf16_t* src = p.B0 + (long long)b * p.b0_bstride + gbase + ch;
                const int sh = (ky - 1) * p.Wp + (kx - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    stage16(src + (rowoff[i] + sh) * p.ld0, Bb + (i * 4 + wave) * 1024, lane);
            } else {
                const int tap = is_tap;
                int ch = is_cc + gbase;                                // wave-uniform
                const bf16_t* src; int ld;
                if (ch < p.c0) { src = p.B0 + (long long)b * p.b0_bstride; ld = p.ld0; }
                else { src = p.B1 + (long long)b * p.b1_bstride; ld = p.ld1; ch -= p.c0; }
                src += ch + jsw * 8;
                int ky = 1, kx = 1;
                if (p.ntaps == 9) { ky = tap_ky(tap); kx = tap - 3 * ky; }
                int sh = 0;
                if (MODE == MODE_S1) sh = (ky - 1) * p.Wp + (kx - 1);
                if (MODE == MODE_DOWN) sh = ky * p.Wpi + kx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int idx;
                    if (MODE == MODE_UP) idx = (((byo[i] + ky - 2) >> 1) + 1) * p.Wpi + (((bxo[i] + kx - 2) >> 1) + 1);
                    else idx = rowoff[i] + sh;
                    stage16(src + idx * ld, Bb + (i * 4 + wave) * 1024, lane);
                }
                is_cc += CG_BK;
                if (is_cc >= p.cg) { is_cc = 0; ++is_tap; }
            }
        };

        // fragment read addresses (LDS bytes, buffer 0, k16 step 0), fixed per lane: row * 128 + ((chunk ^ swizzle) << 4) with chunk = 2 kk + lane / 32 =
        // (2 kk) ^ (lane / 32): step kk is the same address ^ (kk << 5), buffer 1 is + TM * 128 (A) / + CG_TP * 128 (B)
        unsigned a_ad[TMT], b_ad[2];
#pragma unroll
        for (int tm = 0; tm < TMT; ++tm) {
            const int row = wm * (TM / 2) + tm * 32 + (lane & 31);
            a_ad[tm] = (unsigned)(row * 128 + ((((lane >> 5) ^ (row >> 1)) & 7) << 4));
        }
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int row = wp * 64 + tp * 32 + (lane & 31);
            b_ad[tp] = (unsigned)(2 * TM * 128 + row * 128 + ((((lane >> 5) ^ (row >> 1)) & 7) << 4));
        }

        CG_STAMP();
        issue(0, 0);
        for (int ks = 0; ks < p.nk; ++ks) {
            const int buf = ks & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            CG_STAMP();
            if (ks + 1 < p.nk) issue(ks + 1, buf ^ 1);
            // the four k16 steps of the stage, software-pipelined by hand (round 5): inline-asm fragment reads with counted lgkmcnt, the reads of
            // step kk + 1 in flight under the MFMAs of step kk.  hipcc's own schedule was read -> lgkmcnt(0) -> MFMAs per step: an LDS round trip in
            // front of every four MFMAs (the K step of a stride-2 conv ran at ~1.3 k cycles for 512 cycles of matrix work)
            const unsigned ao = (unsigned)(buf * (TM * 128)), bo = (unsigned)(buf * (CG_TP * 128));
            bf16x8_t af[2][TMT], bfr[2][2];
            auto rd = [&](auto kkc) {
                constexpr int kk = decltype(kkc)::value;
#pragma unroll
                for (int tm = 0; tm < TMT; ++tm) lds_read16_asm<0>(af[kk & 1][tm], (a_ad[tm] + ao) ^ (unsigned)(kk << 5));
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) lds_read16_asm<0>(bfr[kk & 1][tp], (b_ad[tp] + bo) ^ (unsigned)(kk << 5));
            };
            __builtin_amdgcn_sched_barrier(0);
            rd(std::integral_constant<int, 0>{});
            static_for<0, 4>([&](auto kkc) {
                constexpr int kk = decltype(kkc)::value;
                if constexpr (kk + 1 < 4) rd(std::integral_constant<int, kk + 1>{});
                lgkm_wait_asm<(kk + 1 < 4) ? TMT + 2 : 0>();
#pragma unroll
                for (int tm = 0; tm < TMT; ++tm)
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp)
                        acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][tm], bfr[kk & 1][tp], acc[tm][tp], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        }

        CG_STAMP();
        // ---- epilogue phase 1: accumulator layout -> fp32 staging tile [pixel][feature] -------
        if (EPI == EPI_STD) {
            __syncthreads();                       // staging aliases the K buffers
#pragma unroll
            for (int tm = 0; tm < TMT; ++tm)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const int px = wp * 64 + tp * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int f = wm * (TM / 2) + tm * 32 + 8 * rg + 4 * (lane >> 5);
                        float4 v = make_float4(acc[tm][tp][rg * 4 + 0], acc[tm][tp][rg * 4 + 1],
                                               acc[tm][tp][rg * 4 + 2], acc[tm][tp][rg * 4 + 3]);
                        *reinterpret_cast<float4*>(&stage[px * SL + f]) = v;
                    }
                }
        } else {
            // AKGM: rows of a 32-row tile are packed so that one lane holds all 8 kernel sets of
            // two features: regs 0..7 -> feature 4*t + h, regs 8..15 -> feature 4*t + 2 + h (h = lane>>5).
            const float mean = scal[0], rstd = scal[1];
            const float mr = mean * rstd;
            const int fgroup0 = g * p.cg + rtg * (TM / 8);      // first global feature of this (group,row tile)
            const int fcol0 = gi * (TM / 8);
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int px = wp * 64 + tp * 32 + (lane & 31);
                int cp = p.p0 + tcol * CG_TP + px;
                cp = cp < plast ? cp : plast;
                int y = cp / p.Wp, x = cp - y * p.Wp;
                const bool valid = (y >= 1 && y <= p.H && x >= 1 && x <= p.W);
                y = y < 1 ? 1 : (y > p.H ? p.H : y);
                x = x < 1 ? 1 : (x > p.W ? p.W : x);
                const int cls = (y == 1 ? 0 : (y == p.H ? 2 : 1)) * 3 + (x == 1 ? 0 : (x == p.W ? 2 : 1));
                const float* gp = p.G + (long long)b * p.g_bstride + ((long long)(y - 1) * p.W + (x - 1)) * 8;
                float att[8];
                {
                    float4 g0 = *reinterpret_cast<const float4*>(gp);
                    float4 g1 = *reinterpret_cast<const float4*>(gp + 4);
                    const float* aw = p.attw + b * 8;
                    att[0] = g0.x * aw[0]; att[1] = g0.y * aw[1]; att[2] = g0.z * aw[2]; att[3] = g0.w * aw[3];
                    att[4] = g1.x * aw[4]; att[5] = g1.y * aw[5]; att[6] = g1.z * aw[6]; att[7] = g1.w * aw[7];
                }
#pragma unroll
                for (int tm = 0; tm < TMT; ++tm) {
                    const int t32 = wm * TMT + tm;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int floc = 4 * t32 + 2 * q + (lane >> 5);
                        const int o = 8 * (fgroup0 + floc);
                        const float* tb = p.Tb + (long long)cls * p.tab_ld + o;
                        const float* tg = p.Tg + (long long)cls * p.tab_ld + o;
                        const float* bs = p.bias + o;
                        float sum = 0.f;
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            float hv = rstd * acc[tm][tp][8 * q + s] + (bs[s] + tb[s] - mr * tg[s]);
                            sum += att[s] * hv;
                        }
                        if (valid) stage[px * SL + fcol0 + floc] = sum;
                        else stage[px * SL + fcol0 + floc] = 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();

    CG_STAMP();
    // ---- epilogue phase 2: coalesced layout, 8 features per thread-item ---------------------------
    const float rstd_s = scal[1];
    const float mr_s = scal[0] * scal[1];
    const float alpha = p.alpha * (p.fold && EPI == EPI_STD ? rstd_s : 1.0f);
    int fbase;
    if (EPI == EPI_STD) fbase = rowtile * TM;
    else {
        if (p.groups_per_wg > 1) fbase = rowtile * p.groups_per_wg * p.cg;
        else { const int rtpg = (8 * p.cg) / TM; const int g = rowtile / rtpg; fbase = g * p.cg + (rowtile - g * rtpg) * (TM / 8); }
    }
    const int nf8 = NF >> 3;
    float s1 = 0.f, s2 = 0.f;
    // A thread's eight features are the same in every item when nf8 divides the workgroup size (it = tid + 256 k): its bias and the fold-table
    // entries of the interior class are loaded ONCE in front of the loop (round 5: six 16-byte L2 round trips per item, serialised by the
    // not-unrolled loop, were a quarter of a small launch - stamps in profiles/EXPERIMENTS.md); border pixels of a 3x3 conv reload their class
    const bool hoist = EPI == EPI_STD && (CG_THREADS % nf8) == 0;
    float4 hb0 = make_float4(0.f, 0.f, 0.f, 0.f), hb1 = hb0, ht0 = hb0, ht1 = hb0, hg0 = hb0, hg1 = hb0;
    const int hcls = (p.ntaps == 9) ? 4 : 0;
    if (hoist) {
        const int fh = fbase + (tid % nf8) * 8;
        if (fh < p.nfeat) {
            if (p.bias) { hb0 = *reinterpret_cast<const float4*>(p.bias + fh); hb1 = *reinterpret_cast<const float4*>(p.bias + fh + 4); }
            if (p.fold) {
                const float* tb = p.Tb + (long long)hcls * p.tab_ld + fh;
                const float* tg = p.Tg + (long long)hcls * p.tab_ld + fh;
                ht0 = *reinterpret_cast<const float4*>(tb); ht1 = *reinterpret_cast<const float4*>(tb + 4);
                hg0 = *reinterpret_cast<const float4*>(tg); hg1 = *reinterpret_cast<const float4*>(tg + 4);
            }
        }
    }
    for (int it = tid; it < CG_TP * nf8; it += CG_THREADS) {
        const int px = it / nf8;
        const int f8 = (it - px * nf8) * 8;
        const int cp = p.p0 + tcol * CG_TP + px;
        if (cp > plast) continue;
        const int f = fbase + f8;
        if (f >= p.nfeat) continue;
        int cls = 0;
        long long opos = cp, rpos = cp;
        if (MODE == MODE_PLAIN) {
            if (p.plain_w) {
                const int y = cp / p.plain_w, x = cp - y * p.plain_w;
                opos = rpos = (long long)(y + 1) * (p.plain_w + 2) + x + 1;
            }
        } else {
            int y = cp / p.Wp, x = cp - y * p.Wp;
            if (!(y >= 1 && y <= p.H && x >= 1 && x <= p.W)) continue;
            cls = (y == 1 ? 0 : (y == p.H ? 2 : 1)) * 3 + (x == 1 ? 0 : (x == p.W ? 2 : 1));
            if (p.out_compact) opos = (long long)(y - 1) * p.W + (x - 1);
        }
        float v[8];
        {
            float4 a = *reinterpret_cast<const float4*>(&stage[px * SL + f8]);
            float4 c = *reinterpret_cast<const float4*>(&stage[px * SL + f8 + 4]);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        }
        if (EPI == EPI_STD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= alpha;
            // (16-byte table loads: f is a multiple of 8 and the tables are allocation-aligned; eight scalar loads per array and item - 24 dependent
            // L2 round trips per item, eight items per thread - were most of a small launch's epilogue)
            if (p.bias) {
                float4 b0 = hb0, b1 = hb1;
                if (!hoist) { b0 = *reinterpret_cast<const float4*>(p.bias + f); b1 = *reinterpret_cast<const float4*>(p.bias + f + 4); }
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (p.fold) {
                const int ncls_row = (p.ntaps == 9) ? cls : 0;
                float4 t0 = ht0, t1 = ht1, g0 = hg0, g1 = hg1;
                if (!hoist || ncls_row != hcls) {
                    const float* tb = p.Tb + (long long)ncls_row * p.tab_ld + f;
                    const float* tg = p.Tg + (long long)ncls_row * p.tab_ld + f;
                    t0 = *reinterpret_cast<const float4*>(tb); t1 = *reinterpret_cast<const float4*>(tb + 4);
                    g0 = *reinterpret_cast<const float4*>(tg); g1 = *reinterpret_cast<const float4*>(tg + 4);
                }
                v[0] += t0.x - mr_s * g0.x; v[1] += t0.y - mr_s * g0.y; v[2] += t0.z - mr_s * g0.z; v[3] += t0.w - mr_s * g0.w;
                v[4] += t1.x - mr_s * g1.x; v[5] += t1.y - mr_s * g1.y; v[6] += t1.z - mr_s * g1.z; v[7] += t1.w - mr_s * g1.w;
            }
        }
        if (p.act) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (p.act == 1) ? silu_f(v[i]) : fmaxf(0.2f * v[i], v[i]);
        }
        if (p.res) {
            const bf16_t* rp = p.res + (long long)b * p.res_bstride + rpos * p.res_ld + p.res_coff + f;
            uint4 rv = *reinterpret_cast<const uint4*>(rp);
            const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bf2f(rh[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
        if (p.shuffle_c) {
            const int q = f / p.shuffle_c, o = f - q * p.shuffle_c;
            const int y = cp / p.Wp - 1, x = cp - (y + 1) * p.Wp - 1;
            const long long op2 = (long long)(2 * y + (q >> 1) + 1) * (2 * p.W + 2) + (2 * x + (q & 1) + 1);
            bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (long long)b * p.out_bstride + op2 * p.out_ld + p.out_coff + o;
            *reinterpret_cast<uint4*>(op) = pack8_bf16(v);
        } else if (p.out_nchw) {
            const int y = cp / p.Wp - 1, x = cp - (y + 1) * p.Wp - 1;
            if (y < p.crop_h && x < p.crop_w) {
                float* op = reinterpret_cast<float*>(p.out) + (((long long)b * p.nfeat + f) * p.crop_h + y) * p.crop_w + x;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (f + i < p.nfeat) op[(long long)i * p.crop_h * p.crop_w] = v[i];
            }
        } else if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.out) + (long long)b * p.out_bstride + opos * p.out_ld + p.out_coff + f;
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (long long)b * p.out_bstride + opos * p.out_ld + p.out_coff + f;
            *reinterpret_cast<uint4*>(op) = p.out_f16 ? pack8_f16(v) : pack8_bf16(v);
        }
    }
    CG_STAMP();
#ifdef UCDIR_TIMING
    if (p.dbg && blockIdx.x == gridDim.x / 2 && tid == 0) p.dbg[63] = cg_n;
#endif
    if (p.stats_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        if (lane == 0) { scal[2 + wave * 2] = s1; scal[3 + wave * 2] = s2; }
        __syncthreads();
        if (tid == 0) {
            const float t1 = scal[2] + scal[4] + scal[6] + scal[8];
            const float t2 = scal[3] + scal[5] + scal[7] + scal[9];
            stat_add(p.stats_out, b, t1, t2);
        }
    }
}
