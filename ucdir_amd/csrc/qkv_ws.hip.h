// q, k, v' = conv1x1(GroupNorm(x)) of SelfAttention (model/ucdir.py:165-182) as a PERSISTENT, weight-stationary GEMM (gfx950).
//
//   [3C x C] (GroupNorm scale folded, v' = W_o W_v: fold_out_into_v) x [C x N pixels] per sample, N = H W.
//   q, k -> qkv [B][N][ld] (q at 0, k at C), v' -> V't [B][C][Npad] TRANSPOSED, which is what flash_attn.hip.h stages by LDS-DMA:
//   the separate transpose_v_kernel launch (14 us x 6 per forward at B = 16) and its round trip through HBM are gone.
//
// Why (round 3): the shifted-GEMM kernel (cgemm_kernel<128, std, s1>) ran this K = 512 product at 410 TFLOP/s (79 us per B = 16
// launch at 36^2) - eight K steps per workgroup, every one waiting for its own LDS-DMA round trip between two barriers.  Here
//   * one workgroup per CU (8 wave64, two per SIMD) owns ONE 256-row tile of the weight for the whole launch: wave w keeps rows
//     32 w .. 32 w + 31 x K = C as MFMA fragments in C / 4 registers (128 at C = 512) - no weight traffic after the prologue;
//   * it walks a contiguous range of 128-pixel tiles (tiles never cross a sample); the pixels' channels arrive in chunks of 128
//     channels ([128 px][128 ch] bf16 = 32 KB, ring of three) by LDS-DMA, chunks c + 1 and c + 2 (of this or the next tile) in
//     flight under the MFMAs of chunk c: ONE barrier per chunk, counted vmcnt;
//   * LDS rows of 256 bytes, 16-byte chunk XOR f(pixel), f = px & 15, on the DMA's source side: the 16 lanes of a
//     ds_read_b128 group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) read 16 different slots in both lane -> pixel mappings below;
//   * q / k tiles: weights are the A operand, rows permuted at pack time so that a lane ends up with 16 consecutive features of
//     one pixel (32-byte stores);  v' tiles: the SAME registers are the B operand and the pixels the A operand (the fragment
//     layouts of v_mfma_f32_32x32x16 are symmetric), pixel rows permuted in the LDS read address instead: a lane ends up with 16
//     consecutive PIXELS of one feature = 32 contiguous bytes of a V't row;
//   * GroupNorm fold in the epilogue as in cgemm.hip.h: y = rstd acc + Tb[f] - mean rstd Tg[f], the per-sample constant
//     Tb - mean rstd Tg of the tile's 256 rows in LDS.
#pragma once
#include "akgm_ws.hip.h"

struct QkvP {
    const bf16_t* A;                           // pack_qkv_ws image
    const bf16_t* x; long long x_bstride;      // GroupNorm input, zero-bordered NHWC, C channels
    int H, W, Wp, N, nbatch, tps, rowtiles;    // tps = 128-pixel tiles per sample, rowtiles = 3C / 256
    const stat_t* stats; double inv_count;
    const float* Tb; const float* Tg;          // [3C]
    bf16_t* qkv; long long qkv_bstride; int ld;
    bf16_t* vt; long long vt_bstride; int Npad;
};

struct QkvWs {
    static constexpr int CHUNK = 128 * 256;                       // [128 px][128 ch] bf16
    static constexpr int NBUF = 3;
    static constexpr int OFF_TBM = NBUF * CHUNK;                     // [256] fp32: Tb - mean rstd Tg of this workgroup's rows, current sample
    static constexpr int OFF_SCAL = OFF_TBM + 1024;
    static constexpr int LDS = OFF_SCAL + 64;
};

template <int C>
__global__ __launch_bounds__(HC_THREADS, 2) void qkv_ws_kernel(const QkvP p) {
    constexpr int NKS = C / 16;                                    // k steps = A fragments per wave
    constexpr int NCH = C / 128;                                   // 128-channel chunks per pixel tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const tbm = reinterpret_cast<float*>(smem + QkvWs::OFF_TBM);
    float* const scal = reinterpret_cast<float*>(smem + QkvWs::OFF_SCAL);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // workgroup -> (row tile, range of pixel tiles): workgroups lid, lid + RT, lid + 2 RT, ... share row tile lid % RT
    const int RT = p.rowtiles, rt = lid % RT, slot = lid / RT;
    const int nwg = ((int)gridDim.x - rt + RT - 1) / RT;           // workgroups on this row tile
    const int T = p.nbatch * p.tps;
    const int t_beg = (int)((long long)slot * T / nwg), t_end = (int)((long long)(slot + 1) * T / nwg);
    if (t_beg >= t_end) return;
    const bool is_v = rt * 256 >= 2 * C;

    // ---- this wave's weight rows, resident ----------------------------------------------------------------------------------
    bf16x8_t af[NKS];
    {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + ((long long)(rt * 8 + wave) * NKS) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < NKS; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < NKS; ++j) asm volatile("" : "+v"(af[j]));

    // ---- lane constants ----------------------------------------------------------------------------------------------------------
    // DMA piece i of a chunk (i = 0..3): pixels 4 (8 i + wave) .. + 3 of the tile, lane -> (pixel + lane / 16, physical chunk lane & 15)
    // B / A fragment of pixel tile tp, k step j of a chunk: pixel 32 tp + px, logical chunk 2 j + hh
    //   q / k: px = l31;  v': px = pi(l31) = 16 ((l31 >> 2) & 1) + 4 (l31 >> 3) + (l31 & 3)  (a lane's 16 accumulators = 16 consecutive pixels)
    const int pxl = is_v ? (16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3)) : l31;
    // (round 5: was (px & 15) ^ 8 [px & 16], built for lane groups {0-7, 16-23}; ds_read_b128 is serviced in {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
    // (conv_halo.hip.h), where that XOR makes the two tile rows of a group collide - SQ_LDS_BANK_CONFLICT 48 % of the LDS cycles, 51.0 -> 48.8 us)
    auto swz = [](int px) { return px & 15; };
    // LDS byte address of (pixel tile 0, k step 0) in buffer 0; k step j: ^ 32 j (chunk 2 j + hh = (2 j) ^ hh), pixel tile tp: + 8192 tp
    const unsigned fr0 = pxl * 256 + ((hh ^ swz(pxl)) << 4);

    int b_cur = -1;
    float rstd = 1.f;
    // pixel byte offsets of this lane's four DMA pieces, for tile (tb, tn)
    unsigned prel[4];
    auto tile_prel = [&](int tn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = 4 * (8 * i + wave) + (lane >> 4);
            int n = tn * 128 + px; n = n < p.N ? n : p.N - 1;
            const int y = n / p.W, x = n - y * p.W;
            prel[i] = (unsigned)(((y + 1) * p.Wp + x + 1) * C + (((lane & 15) ^ swz(px)) << 3)) * 2;
        }
    };
    auto issue_chunk = [&](int tb, int c, int buf) {
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x + (long long)tb * p.x_bstride) + c * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(xb + prel[i]), (LDS_AS void*)(smem + buf * QkvWs::CHUNK + (8 * i + wave) * 1024), 16, 0, 0);
    };

    // issue stream: chunk ic of tile (ib, in), running up to two chunks ahead of the consume stream
    int tb = t_beg / p.tps, tn = t_beg - tb * p.tps;
    int it = t_beg, ib = tb, in = tn, ic = 0, ibuf = 0, ahead = 0;
    tile_prel(in);
    auto issue_next = [&]() {                                       // the next chunk of the stream, if there is one
        if (it >= t_end) return;
        issue_chunk(ib, ic, ibuf);
        ibuf = ibuf + 1 == QkvWs::NBUF ? 0 : ibuf + 1;
        ++ahead;
        if (++ic == NCH) {
            ic = 0; ++it;
            if (++in == p.tps) { in = 0; ++ib; }
            if (it < t_end) tile_prel(in);
        }
    };
    issue_next();
    issue_next();
    int gbuf = 0;                                                  // buffer of the chunk about to be consumed

    // two specialised copies of the tile loop (operand order of the MFMAs and the epilogue differ; a per-MFMA select on the
    // wave-uniform flag made hipcc branch around every MFMA)
    auto tile_loop = [&](auto vtag) {
    constexpr bool VT = decltype(vtag)::value;
#pragma unroll 1
    for (int t = t_beg; t < t_end; ++t) {
        int nb = tb, nn = tn + 1;
        if (nn == p.tps) { nn = 0; ++nb; }
        if (tb != b_cur) {                                          // new sample: fold constants of this workgroup's 256 rows
            asm volatile("s_barrier" ::: "memory");                  // everybody is done with the previous sample's table
            b_cur = tb;
            if (tid == 0) {
                double S, Q; float mean, r2;
                stat_read(p.stats, nullptr, tb, S, Q);
                mean_rstd(S, Q, p.inv_count, mean, r2);
                scal[0] = mean * r2; scal[1] = r2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            rstd = scal[1];
            if (tid < 256) tbm[tid] = p.Tb[rt * 256 + tid] - scal[0] * p.Tg[rt * 256 + tid];
            // (visible to everybody behind the first chunk barrier below)
        }
        f32x16_t acc[4];
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tp][e] = 0.f;

#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // The chunk about to be consumed has landed: this wave's LDS-DMAs retire in order, `ahead` - 1 younger chunks (four
            // pieces each) may stay in flight; epilogue stores retire at any time - a count of 4 (ahead - 1) holds whether or
            // not they have.  The barrier also says everybody is done with the chunk consumed last: its buffer takes the
            // next chunk of the stream.
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            --ahead;
            issue_next();
            const unsigned bb = gbuf * QkvWs::CHUNK;
            // 16 steps (k step j = s / 2, pixel-tile pair th = s % 2) of {two fragment reads, two MFMAs}.  Inline-asm reads with counted lgkmcnt, the
            // fragments of steps s + 1 and s + 2 in flight under the MFMAs of step s (round 5: hipcc's own bookkeeping put a full lgkmcnt(0) between
            // each read pair and its MFMAs - waves waiting 48 %, MFMA busy 26 % in profiles/r04_pmc_sq.csv)
            bf16x8_t xf[3][2];
            auto frag = [&](auto sc, bf16x8_t (&dst)[2]) {
                constexpr int st = decltype(sc)::value, j = st >> 1, th = st & 1;
                const unsigned fa = bb + (fr0 ^ (32u * j));
                lds_read16_asm<(2 * th) * 8192>(dst[0], fa);
                lds_read16_asm<(2 * th + 1) * 8192>(dst[1], fa);
            };
            __builtin_amdgcn_sched_barrier(0);
            frag(std::integral_constant<int, 0>{}, xf[0]);
            frag(std::integral_constant<int, 1>{}, xf[1]);
            __builtin_amdgcn_s_setprio(1);
            static_for<0, 16>([&](auto sc) {
                constexpr int st = decltype(sc)::value, j = st >> 1, th = st & 1;
                if constexpr (st + 2 < 16) frag(std::integral_constant<int, st + 2>{}, xf[(st + 2) % 3]);
                constexpr int younger = (st + 2 < 16) ? 4 : ((st + 1 < 16) ? 2 : 0);
                lgkm_wait_asm<younger>();
#pragma unroll
                for (int tq = 0; tq < 2; ++tq)
                    acc[2 * th + tq] = VT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[st % 3][tq], af[8 * c + j], acc[2 * th + tq], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[8 * c + j], xf[st % 3][tq], acc[2 * th + tq], 0, 0, 0);
            });
            __builtin_amdgcn_s_setprio(0);
            gbuf = gbuf + 1 == QkvWs::NBUF ? 0 : gbuf + 1;
        }

        // ---- epilogue ----------------------------------------------------------------------------------------------------------
        const int n0 = tn * 128;
        if constexpr (!VT) {
            // lane = pixel 32 tp + l31, features 256 rt + 32 wave + 16 hh .. + 15
            const int f0 = 32 * wave + 16 * hh;
            bf16_t* ob = p.qkv + (long long)tb * p.qkv_bstride + rt * 256 + f0;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const int n = n0 + 32 * tp + l31;
                bf16_t* op = ob + (long long)n * p.ld;
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) {                    // eight features at a time (registers)
                    const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(smem + QkvWs::OFF_TBM + (f0 + 8 * h8) * 4);
                    const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(smem + QkvWs::OFF_TBM + (f0 + 8 * h8 + 4) * 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = fmaf(acc[tp][8 * h8 + e], rstd, c0[e]); v[4 + e] = fmaf(acc[tp][8 * h8 + 4 + e], rstd, c1[e]); }
                    if (n < p.N) *reinterpret_cast<uint4*>(op + 8 * h8) = pack8_bf16(v);
                }
            }
        } else {
            // lane = feature 256 rt - 2C + 32 wave + l31, pixels 32 tp + 16 hh .. + 15
            const int f = 32 * wave + l31;
            const float cst = tbm[f];
            bf16_t* ob = p.vt + (long long)tb * p.vt_bstride + (long long)(rt * 256 - 2 * C + f) * p.Npad;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const int n = n0 + 32 * tp + 16 * hh;
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(acc[tp][8 * h8 + e], rstd, cst);
                    if (n < p.Npad) *reinterpret_cast<uint4*>(ob + n + 8 * h8) = pack8_bf16(v);   // Npad is a multiple of 64: a group of 16 is inside or outside as a whole
                }
            }
        }
        tb = nb; tn = nn;
    }
    };
    if (is_v) tile_loop(std::true_type{}); else tile_loop(std::false_type{});
}
