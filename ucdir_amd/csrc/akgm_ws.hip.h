// AKGM tail at 8 channels per group (C = 64, the 288^2 level) as a PERSISTENT, weight-stationary kernel (gfx950).
// Reference: model/ucdir.py:129-140.
//
// Why (round 3): akgm_pre_kernel<8> launches 10,368 workgroups per B = 16 layer; each DMAs 40 KB of unit weights, its
// fold tables and its halo before its first MFMA and lives 20-24 k cycles for 2.5 k cycles of matrix-core work.  The
// L2 -> LDS fill volume (0.9 GB per launch, three quarters of it weights) and the exposed first-byte latency put the
// launch at 2.7x its HBM floor.  Here (H and W multiples of 16, which every level-0 grid of the network is):
//   * the grid is ONE workgroup per CU (8 wave64, two per SIMD, 256 VGPRs each); a workgroup walks a contiguous
//     range of 16 x 16 pixel tiles (XCD-contiguous, so neighbouring tiles share an L2);
//   * wave g owns GROUP g for the whole launch: its 64 weight rows x K = 72 (+ one zero tap) sit in 40 VGPRs as
//     MFMA A fragments, loaded once - the K loop reads only B fragments (its group's 16 bytes of each halo pixel)
//     from LDS, there is no weight traffic, no ring and no barrier inside a tile;
//   * halo in LDS: [18 rows][24-pixel pitch][64 ch] bf16, 16-byte chunk XOR (pixel >> 1) & 7 (on the DMA's source
//     side).  The pitch makes a 32-pixel MFMA tile (two tile rows) exactly 48 pixels = 6,144 bytes and leaves the
//     swizzle unchanged, so every B fragment address is one of five per-lane registers + an immediate;
//   * the halo and the guide weights G of tile t + 1 are requested by LDS-DMA during tile t into the second buffer, two or
//     three pieces per pixel pair (all at the tile top: every wave stalls at issue for thousands of cycles): ONE barrier
//     per tile and no vmcnt in front of it (see the residual order below);
//   * GroupNorm-fold table Tc[9][512] of the current sample resident in LDS (reloaded when the range crosses a
//     sample); accumulators start at it, modulation sum / half-wave exchange / swish / residual / 16-byte store as in
//     akgm_pre.hip.h;
//   * GroupNorm statistics of the output: a lane's per-tile fp32 partial is converted to 2^-20 fixed point and
//     accumulated in 64-bit integers, so the totals do not depend on how tiles are dealt to workgroups (bit-identical
//     for any batch size or CU count) and no barrier is needed for them.
#pragma once
#include "akgm_pre.hip.h"

struct AkWs {
    static constexpr int PITCH = 24;                              // halo pixels per LDS row (18 used)
    static constexpr int HALO = 18 * PITCH * 128;                 // 55,296: [18][24][64 ch] bf16
    static constexpr int QSTEP = 2 * PITCH * 128;                 // 6,144: one 32-pixel MFMA tile = two tile rows further
    static constexpr int ATT = 256 * 32;                          // [256 px][8] fp32
    static constexpr int OFF_ATT = 2 * HALO;
    static constexpr int OFF_TCS = OFF_ATT + 2 * ATT;             // [9][512] fp32
    static constexpr int OFF_SCAL = OFF_TCS + 9 * 512 * 4;
    static constexpr int OFF_RES = OFF_SCAL + 128;                // residual staging: [8 waves][2 slots][64 lanes x 16 B]
    static constexpr int LDS = OFF_RES + 8 * 2 * 1024;            // 161,920 of 163,840: one workgroup per CU
    static constexpr int NDMA = 8;                                // LDS-DMA instructions per wave and tile (7 halo pieces + 1 guide piece)
};

__device__ __forceinline__ void stat_add_fx(stat_t* stats, int b, stat_t s, stat_t q) {
    stat_t* dst = stats + ((long long)b * UCDIR_STAT_SLOTS + (blockIdx.x % UCDIR_STAT_SLOTS)) * 2;
    (void)__hip_atomic_fetch_add(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_add(dst + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ stat_t wave_sum_ll(stat_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// 16-byte global load the compiler does not know about (address = wave-uniform base + per-lane byte offset): it is absent
// from hipcc's vmcnt bookkeeping (no vmcnt(0) in front of its first use that would drain the LDS-DMA queue) and its result
// is only touched behind WS_WAIT_RES (cdna guide 5.7, form iii)
__device__ __forceinline__ u32x4_t asm_load16(const void* base, unsigned voff) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base) : "memory");
    return v;
}
// Wait until at most N VMEM operations of this wave are outstanding, THEN hand the asm-loaded registers to the compiler.  The
// wait statement carries no register operand on purpose: with "+v"(v) on the s_waitcnt itself hipcc allocated the operand
// elsewhere and copied the load's destination registers IN FRONT of the wait (v_mov_b64 v[0:1], v[116:117]; s_waitcnt ...):
// garbage whenever the load had not landed yet (found by test_akgm_persistent[level0]: the last tiles of the longer ranges).
// A copy in front of the empty statement is behind the wait and harmless.
#define WS_WAIT_RES(N, v) do { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); asm volatile("" : "+v"(v) :: "memory"); } while (0)

#ifndef WS_PK
#define WS_PK 0                                                   // 1: modulation sum on v_pk_fma_f32 - half the issue slots, but measured 2 % SLOWER (same box, 198 vs 194 us): packed fp32 beside the partner wave's MFMAs is an anti-lever (cdna guide, price list)
#endif

// CG = 8 : C = 64, wave = group (above).
// CG = 16: C = 128 (the 144^2 level).  A group's 128 rows x K = 144 are 144 A registers - too many next to 64 accumulators -
//          so a workgroup owns ONE 64-channel plane of the tensor (ch = lid & 1; the two workgroups of a pair walk the same tile
//          range and share G and the pixel lines in their XCD's L2) and wave w owns output features 64 ch + 8 w .. + 7: the
//          64 rows of row half w & 1 of group 4 ch + w / 2, K = 9 taps x 16 channels = 72 A registers.  Everything else - halo
//          plane in LDS, fold-table slice [9][512], epilogue, 16-byte stores - is the CG = 8 kernel with a 256-byte pixel.
template <int CG>
__global__ __launch_bounds__(HC_THREADS, 2) void akgm_ws_kernel(const AkgmHP p) {
    constexpr int NK = (CG == 8) ? 5 : 9;                          // 16-wide k steps
    constexpr int NCH = CG / 8;                                    // 64-channel planes per pixel
    constexpr int CPX = 64 * NCH;                                  // channels per pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = group (CG 8) | half a group (CG 16)
    const int wm = wave & 1;                                      // row half of pack_akgm_pre's unit
    // lane -> pixel of a 32-pixel MFMA tile (two tile rows): row l31 / 16, column (l31 % 16) ^ 8 in the second row: with the
    // 24-pixel pitch every ds_read_b128 lane group then reads 16 different 16-byte slots (the plain mapping was 2-way everywhere)
    const int prow = l31 >> 4, pcol = (l31 & 15) ^ (prow << 3);
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2 + 3) && (lane == 0) && (wave == 5);
    int dbg_n = 0;
#define WS_STAMP() do { if (dbg_on && dbg_n < 250) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP() do {} while (0)
#endif
#ifdef WS_FINE_STAMPS                                              // store-phase anatomy: behind the wait | DMA issued | residual requested
#define WS_STAMP2() WS_STAMP()
#else
#define WS_STAMP2() do {} while (0)
#endif
    WS_STAMP();
    const int tps = p.tiles_x * p.tiles_y;
    const int T = p.nbatch * tps;
    const int ch = (NCH == 2) ? (lid & 1) : 0;                     // this workgroup's channel plane
    const int rid = (NCH == 2) ? (lid >> 1) : lid, nrange = (int)gridDim.x / NCH;
    const int t_beg = (int)((long long)rid * T / nrange), t_end = (int)((long long)(rid + 1) * T / nrange);
    if (t_beg >= t_end) return;
    const int U = (NCH == 2 ? 4 * ch : 0) + (wave >> 1);           // unit of pack_akgm_pre (CG 8: 16 features of two groups, CG 16: one group)

    // ---- this wave's weights: A fragments of pack_akgm_pre's image, resident for the whole launch -------------------
    bf16x8_t af[2][NK];
    {
        const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + (long long)U * AkPre<CG>::A_UNIT + (hh * 128 + wm * 64 + l31) * 16;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int j = 0; j < NK; ++j) af[tm][j] = *reinterpret_cast<const bf16x8_t*>(Ab + j * 4096 + tm * 512);
    }
    // the compiler's wait for these loads goes HERE (an asm statement reading them), not in front of their first use inside the
    // tile loop, where a vmcnt(0) per pixel pair would drain the next tile's DMA and the previous pair's store
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int j = 0; j < NK; ++j) asm volatile("" : "+v"(af[tm][j]));

    // ---- tile-invariant lane constants ---------------------------------------------------------------------------------
    // halo piece k = r * 3 + c3 (row r, pixel columns 8 c3 .. 8 c3 + 7; columns >= 18 do not exist): wave w stages pieces
    // w, w + 8, ..., w + 48 (waves 6 and 7 repeat piece 53: every wave issues the same number of DMA instructions)
    int hrel[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int k = i * 8 + wave; k = k > 53 ? 53 : k;
        const int r = k / 3, col = (k - 3 * r) * 8 + (lane >> 3);
        const int hp = r * AkWs::PITCH + col;
        hrel[i] = col < 18 ? (r * p.Wp + col) * CPX + 64 * ch + (((lane & 7) ^ ((hp >> 1) & 7)) << 3) : -1;
    }
    const int grel = (((wave * 32 + (lane >> 1)) >> 4) * p.W + ((lane >> 1) & 15)) * 8 + (lane & 1) * 4;   // guide piece `wave`: pixel 32 wave + lane / 2
    unsigned bj[NK];                               // B fragment of k step j, px-tile 0, current buffer: LDS byte address
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        // CG 8 : tap 2j (lanes 0-31) | 2j + 1 (lanes 32-63; tap 9 has zero weights: reads tap 8), the group's one 16-byte chunk
        // CG 16: tap j, the group's chunk pair 2 (w / 2) + lane half
        const int t = (CG == 8) ? ((2 * j + hh > 8) ? 8 : 2 * j + hh) : j;
        const int chunk = (CG == 8) ? wave : 2 * (wave >> 1) + hh;
        const int hp = (prow + t / 3) * AkWs::PITCH + pcol + t % 3;
        bj[j] = ((hp << 7) | ((chunk ^ ((hp >> 1) & 7)) << 4)) + AkWs::HALO;      // (flipped to buffer 0 at the top of the first tile)
    }
    const unsigned tc_lane = AkWs::OFF_TCS + 4 * 8 * (8 * wave + 2 * hh);   // + 128 tm + 2048 cls: first of this lane's 16 table entries
    const unsigned att_lane = AkWs::OFF_ATT + (prow * 16 + pcol) * 32;              // + 1024 q: this lane's pixel of px-tile q
    const long long pp_step = (long long)4 * p.Wp * CPX * 2;           // bytes between the store items of consecutive pairs

    int b, ty, tx;                                 // tile t
    {
        const int t_first = p.reverse ? t_end - 1 : t_beg;
        b = t_first / tps;
        const int r = t_first - b * tps;
        ty = r / p.tiles_x; tx = r - ty * p.tiles_x;
    }
    // DMA piece i of a tile (i < 7: this wave's halo piece 8 i + wave; 7: its guide piece) into buffer `buf`
    auto issue_piece = [&](auto ic, const bf16_t* hb, const float* gb, int buf) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 7) {
            int k = i * 8 + wave; k = k > 53 ? 53 : k;
            if (hrel[i] >= 0) stage16(hb + hrel[i], smem + buf * AkWs::HALO + k * 1024, lane);
        } else {
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gb + grel), (LDS_AS void*)(smem + AkWs::OFF_ATT + buf * AkWs::ATT + wave * 1024), 16, 0, 0);
        }
    };
    auto halo_base = [&](int nb, int nty, int ntx) { return p.h + (long long)nb * p.h_bstride + (long long)(nty * 16 * p.Wp + ntx * 16) * CPX; };
    auto guide_base = [&](int nb, int nty, int ntx) { return p.G + (long long)nb * p.g_bstride + (long long)(nty * 16 * p.W + ntx * 16) * 8; };
    auto res_base = [&](int nb, int nty, int ntx) {
        return reinterpret_cast<const unsigned char*>(p.res + (long long)nb * p.res_bstride + (long long)(nty * 16 * p.Wp + ntx * 16) * CPX);
    };
    {
        const bf16_t* hb = halo_base(b, ty, tx);
        const float* gb = guide_base(b, ty, tx);
        static_for<0, 8>([&](auto ic) { issue_piece(ic, hb, gb, 0); });
    }
    // Output / residual staging (AkWs::OFF_RES): [2 slots][64 pixels of a pair][128 B = this plane's 64 channels], 16-byte
    // chunk XOR (pixel & 7).  A wave owns 16 bytes (its 8 features) of EVERY pixel: stored straight from registers that is 64
    // sixteen-byte pieces of 64 different 128-byte lines per instruction - the stores alone cost 50 of 186 us per level-0 launch
    // (measured with the store removed), the residual loads of the same shape 20 more.  Here wave w moves whole lines instead:
    //   residual of pair q + 2: LDS-DMA of pixels 8 w .. 8 w + 7 (1 KB, eight full lines) into the slot, one pair ahead;
    //   epilogue of pair q   : lane = pixel reads its residual chunk from the slot and writes the result back in place;
    //   sync S(q)            : own DMA landed (counted vmcnt), own chunks written (lgkmcnt), s_barrier;
    //   behind S(q)          : lane = (pixel 8 w + j, chunk c) reads the finished lines and stores 1 KB = eight full lines.
    // S(3) is also the tile barrier (all K loops of the tile done, all pieces of the next tile landed).
    const unsigned stg_lane = AkWs::OFF_RES + lane * 128 + ((wave ^ (lane & 7)) << 4);      // epilogue: pixel = lane, this wave's chunk
    const unsigned line_lane = AkWs::OFF_RES + wave * 1024 + lane * 16;                      // line mover: lane-linear
    unsigned rel3;                                                                           // line mover: byte offset of (pixel 8 w + lane / 8, logical chunk) in the tile
    {
        const int j = lane >> 3, c = lane & 7, row = wave >> 1, col = (8 * (wave & 1) + j) ^ ((row & 1) << 3);
        rel3 = (unsigned)(((row + 1) * p.Wp + col + 1) * CPX + 64 * ch) * 2 + ((c ^ j) << 4);
    }
    auto issue_res = [&](const unsigned char* src, int slot) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + rel3), (LDS_AS void*)(smem + AkWs::OFF_RES + slot * 8192 + wave * 1024), 16, 0, 0);
    };
    issue_res(res_base(b, ty, tx), 0);
    issue_res(res_base(b, ty, tx) + pp_step, 1);
    HC_WAIT(0);
    asm volatile("s_barrier" ::: "memory");

    int b_cur = -1;
    float rstd = 1.f, aw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) aw[s] = 0.f;
    stat_t S1 = 0, S2 = 0;                                          // fixed-point statistics of sample b_cur, this lane

#pragma unroll 1
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        const bool last = t + 1 == t_end;
        WS_STAMP();                                                 // tile top
        // (the barrier that opens this tile is S(3) of the previous one - or the one in front of the loop)
        WS_STAMP();                                                 // behind the barrier
        if (b != b_cur) {                                           // range enters a new sample: its fold table, rstd, attw
            if (b_cur >= 0 && p.stats_out) {
                const stat_t a = wave_sum_ll(S1), q2 = wave_sum_ll(S2);
                if (lane == 0) stat_add_fx(p.stats_out, b_cur, a, q2);
            }
            S1 = 0; S2 = 0;
            b_cur = b;
            float mean_b = 0.f;
            if (p.own_tc) stat_mean_rstd_wave(p.stats, b, p.inv_count, lane, mean_b, rstd);
            else rstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ms[2 * b + 1])));
            const float inv_b = 1.0f / rstd;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pc = i * 8 + wave;
                if (pc < 18) {
                    const long long rel = (long long)(pc >> 1) * (512 * NCH) + 512 * ch + (pc & 1) * 256 + lane * 4;
                    if (p.own_tc) akgm_tc_piece(p, rel, smem + AkWs::OFF_TCS + pc * 1024 + lane * 16, inv_b, mean_b);
                    else __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.Tc + (long long)b * 9 * (512 * NCH) + rel),
                                                          (LDS_AS void*)(smem + AkWs::OFF_TCS + pc * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) aw[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.attw[b * 8 + s])));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const bool interior = ty > 0 && tx > 0 && ty + 1 < p.tiles_y && tx + 1 < p.tiles_x;
        const long long tile_el = (long long)(ty * 16 * p.Wp + tx * 16) * CPX;
        const unsigned char* resb = reinterpret_cast<const unsigned char*>(p.res + (long long)b * p.res_bstride + tile_el);
        unsigned char* outb = reinterpret_cast<unsigned char*>(p.out + (long long)b * p.out_bstride + tile_el);

        // Store item pp of this lane: pixel 64 pp + lane, features 8 g .. 8 g + 7.  Its residual (16 bytes) is requested one pair
        // AHEAD by LDS-DMA into this wave's staging slot, in front of that pair's share of the DMA pieces of tile t + 1
        // (do_pair).  History: plain asm loads with counted waits that took younger DMA pieces for "still in flight" broke
        // bit-reproducibility of a B = 32, T = 100 restoration - an LDS-DMA can retire before an older plain load; plain
        // loads with vmcnt(0) drained the queue every pair.
        int nb = b, nty = ty, ntx = tx + 1;                          // tile t + 1 (reverse: the tile in front of this one)
        if (!p.reverse) { if (ntx == p.tiles_x) { ntx = 0; if (++nty == p.tiles_y) { nty = 0; ++nb; } } }
        else { ntx = tx - 1; if (ntx < 0) { ntx = p.tiles_x - 1; if (--nty < 0) { nty = p.tiles_y - 1; --nb; } } }
        const bf16_t* hbn = halo_base(nb, nty, ntx);
        const float* gbn = guide_base(nb, nty, ntx);
        const unsigned char* resn = res_base(nb, nty, ntx);
        WS_STAMP();                                                 // (nothing is issued here any more)

#pragma unroll
        for (int j = 0; j < NK; ++j) bj[j] = buf ? bj[j] + AkWs::HALO : bj[j] - AkWs::HALO;     // in place: NK registers, not 2 NK
        const unsigned attq = att_lane + buf * AkWs::ATT;

        float s1 = 0.f, s2 = 0.f;
        // This pair's share of the DMA pieces of tile t + 1.  (Issuing them at a different point of the pair per wave class -
        // before / inside / behind the K loop / here - so that the eight waves do not ask at the same moment was measured:
        // 188 -> 193 us and 108 -> 115 us.  The wait is not a queue of simultaneous requests.)
        auto issue_dma = [&](auto ppc) {
            constexpr int PP = decltype(ppc)::value;
            if constexpr (PP == 0) { issue_piece(std::integral_constant<int, 0>{}, hbn, gbn, buf ^ 1); issue_piece(std::integral_constant<int, 1>{}, hbn, gbn, buf ^ 1); issue_piece(std::integral_constant<int, 2>{}, hbn, gbn, buf ^ 1); }
            if constexpr (PP == 1) { issue_piece(std::integral_constant<int, 3>{}, hbn, gbn, buf ^ 1); issue_piece(std::integral_constant<int, 4>{}, hbn, gbn, buf ^ 1); issue_piece(std::integral_constant<int, 5>{}, hbn, gbn, buf ^ 1); }
            if constexpr (PP == 2) { issue_piece(std::integral_constant<int, 6>{}, hbn, gbn, buf ^ 1); issue_piece(std::integral_constant<int, 7>{}, hbn, gbn, buf ^ 1); }
        };
        auto do_pair = [&](auto ppc) {
            constexpr int PP = decltype(ppc)::value;
            const int pp = PP;
            // ---- accumulators start at the fold constants of the pixel's border class --------------------------------
            f32x16_t acc[2][2];
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                unsigned tca = tc_lane + 4 * 2048;                  // class 4
                if (!interior) {
                    const int r = 4 * pp + 2 * tp + prow, c = pcol;
                    const int cy = (ty == 0 && r == 0) ? 0 : ((ty + 1 == p.tiles_y && r == 15) ? 2 : 1);
                    const int cx = (tx == 0 && c == 0) ? 0 : ((tx + 1 == p.tiles_x && c == 15) ? 2 : 1);
                    tca = tc_lane + (cy * 3 + cx) * 2048;
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) acc[tm][tp] = lds_read_f32x16(smem + tca + 128 * tm);   // typed vector loads (no vmcnt(0) against the DMA in flight), concatenated: no v_mov
            }
            // ---- K loop: 5 steps of two taps x 8 channels; B fragments from the halo, A fragments from registers ------
            // software pipeline by hand: B fragments by inline-asm ds_read_b128 with counted lgkmcnt (hipcc's own bookkeeping put
            // a full lgkmcnt(0) in front of every step - or, with a second register set, every second step - ~100 idle
            // matrix-core cycles each): the fragments of step j + 1 are requested in front of the MFMAs of step j, the pixel's
            // modulation weights in front of the last two steps instead of behind the loop
            bf16x8_t bfr[2][2];
            f32x4_t a0[2], a1[2];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fold constants; nothing of the compiler's own is queued behind this
            static_for<0, 2>([&](auto tpc) { constexpr int tp = decltype(tpc)::value; lds_read16_asm<(2 * PP + tp) * AkWs::QSTEP>(bfr[0][tp], bj[0]); });
            __builtin_amdgcn_s_setprio(1);
            static_for<0, NK>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + 1 < NK) {
                    static_for<0, 2>([&](auto tpc) { constexpr int tp = decltype(tpc)::value; lds_read16_asm<(2 * PP + tp) * AkWs::QSTEP>(bfr[(j + 1) & 1][tp], bj[j + 1]); });
                }
                if constexpr (j == NK - 2) {
                    static_for<0, 2>([&](auto tpc) {
                        constexpr int tp = decltype(tpc)::value;
                        lds_read16f_asm<(2 * PP + tp) * 1024>(a0[tp], attq);
                        lds_read16f_asm<(2 * PP + tp) * 1024 + 16>(a1[tp], attq);
                    });
                }
                // reads younger than step j's: step j + 1's two, and at j = NK - 2 the four weight reads; at the last step the
                // weight reads are younger than nothing that is needed -> they are waited for behind the loop
                constexpr int younger = (j + 1 < NK ? 2 : 0) + (j == NK - 2 ? 4 : 0) + (j == NK - 1 ? 4 : 0);
                lgkm_wait_asm<(j == NK - 1) ? 4 : younger>();
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp)
                        acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][j], bfr[j & 1][tp], acc[tm][tp], 0, 0, 0);
            });
            lgkm_wait_asm<0>();
            __builtin_amdgcn_s_setprio(0);
            WS_STAMP();                                             // K loop done
            // ---- modulation sum in registers: vq[tm][q][tp] = feature 8 g + 4 tm + 2 hh + q of pixel tp ----------------
            float vq[2][2][2];
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
#if WS_PK
                const f32x2_t at2[4] = {{a0[tp][0] * aw[0], a0[tp][1] * aw[1]}, {a0[tp][2] * aw[2], a0[tp][3] * aw[3]}, {a1[tp][0] * aw[4], a1[tp][1] * aw[5]}, {a1[tp][2] * aw[6], a1[tp][3] * aw[7]}};
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x2_t sa = {0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const f32x2_t a2 = {acc[tm][tp][8 * q + 2 * s], acc[tm][tp][8 * q + 2 * s + 1]};
                            sa = __builtin_elementwise_fma(at2[s], a2, sa);
                        }
                        vq[tm][q][tp] = rstd * (sa[0] + sa[1]);
                    }
#else
                const float att[8] = {a0[tp][0] * aw[0], a0[tp][1] * aw[1], a0[tp][2] * aw[2], a0[tp][3] * aw[3], a1[tp][0] * aw[4], a1[tp][1] * aw[5], a1[tp][2] * aw[6], a1[tp][3] * aw[7]};
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float sa = 0.f;
#pragma unroll
                        for (int s = 0; s < 8; ++s) sa += att[s] * acc[tm][tp][8 * q + s];
                        vq[tm][q][tp] = rstd * sa;
                    }
#endif
            }
            // ---- half-wave exchange: lane L ends up with all eight features of pixel 64 pp + L --------------------------
            float o8[8];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float lo = vq[tm][q][0], hi = vq[tm][q][1];
                    permlane32_swap(lo, hi);
                    o8[4 * tm + q] = lo;
                    o8[4 * tm + 2 + q] = hi;
                }
            WS_STAMP();                                             // modulation sum + exchange done
            // ---- swish + residual + statistics + 16-byte store -----------------------------------------------------------
            // (c) this lane's pixel: residual chunk from the staging slot, swish + residual, statistics, result back in place
            const unsigned stg = stg_lane + (PP & 1) * 8192;
            const u32x4_t rvp = *reinterpret_cast<const u32x4_t*>(smem + stg);
            float vv[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vv[2 * i] = silu_fast(o8[2 * i]) + __builtin_bit_cast(float, rvp[i] << 16);
                vv[2 * i + 1] = silu_fast(o8[2 * i + 1]) + __builtin_bit_cast(float, rvp[i] & 0xffff0000u);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1 += vv[i]; s2 += vv[i] * vv[i]; }
            {
                const uint4 pk = pack8_bf16(vv);
                *reinterpret_cast<u32x4_t*>(smem + stg) = (u32x4_t){pk.x, pk.y, pk.z, pk.w};
            }
            WS_STAMP2();
            // S(q): the residual lines of pair q + 1 (this wave's DMA, issued behind S(q - 1)) have landed.  In flight, oldest
            // first: the store behind S(q - 1), that residual DMA, the n pieces issued behind it (3 | 3, none in the last tile
            // of the range).  LDS-DMAs retire in order, the store at any time: "at most n outstanding" = at least two retired
            // = the residual among them.  Pairs 0 and 3 have nothing (that may stay) in flight: pair 3 is the tile barrier.
            if constexpr (PP == 0 || PP == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (last) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            WS_STAMP2();
            // (e) whole lines out: pixels 8 w .. 8 w + 7 of the pair
            {
                const u32x4_t ln = *reinterpret_cast<const u32x4_t*>(smem + line_lane + (PP & 1) * 8192);
                *reinterpret_cast<u32x4_t*>(outb + pp * pp_step + rel3) = ln;
            }
            asm volatile("" ::: "memory");
            // residual lines of pair q + 2 into the rows just read (nobody else touches them before S(q + 1)), then this
            // pair's share of the DMA pieces of tile t + 1
            if constexpr (PP < 2) issue_res(resb + (PP + 2) * pp_step, PP & 1);
            else if (!last) issue_res(resn + (PP - 2) * pp_step, PP & 1);
            if (!last) issue_dma(ppc);
            WS_STAMP2();
            WS_STAMP();                                             // pair stored
            __builtin_amdgcn_sched_barrier(0);                       // pairs are not interleaved by the compiler (register pressure)
        };
        do_pair(std::integral_constant<int, 0>{});
        do_pair(std::integral_constant<int, 1>{});
        do_pair(std::integral_constant<int, 2>{});
        do_pair(std::integral_constant<int, 3>{});
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
