// Flash attention, second version (round 5; model/ucdir.py:165-182): flash_attn.hip.h's structure, operands, LDS layout and epilogue - 128 queries per
// workgroup, 64-key tiles, S phase / softmax / PV phase with two barriers per tile - with every fragment read of the two MFMA phases as INLINE
// ASM with COUNTED lgkmcnt waits.  hipcc's own LDS bookkeeping drains the queue (lgkmcnt(0)) every few MFMAs, so both phases of
// flash_attn_kernel are LDS-LATENCY bound: ~50 cycles per 16-cycle MFMA in the S phase, ~150 per 32-cycle MFMA in the PV phase (s_memtime
// stamps), 9.4 k cycles per tile for 4.1 k of matrix work.  Here the S phase keeps four fragment reads (one k32 step) in flight under the four
// MFMAs of the step before, the PV phase one whole k16 step (2 P + NC V't fragments); the row maximum of the softmax uses
// v_permlane16_swap / v_permlane32_swap instead of two ds_bpermute round trips.
//
// (Built first, measured, replaced: the two query halves of the workgroup one phase apart on 32-key tiles with K and V't double-buffered
// - the round-4 verdict's schedule.  It removes the matrix-pipe contention between the two waves of a SIMD, but these phases are not
// pipe-bound: each wave's own S + softmax chain was 1.9 k cycles per 32 keys whichever phase its partner was in, and with a single S wave per
// SIMD nothing covers its LDS latency - 120-126 us against the 111 us of flash_attn_kernel at N = 1296, B = 16.  profiles/EXPERIMENTS.md.)
#pragma once
#include "flash_attn.hip.h"
#include "asmops.hip.h"

// LDS fragment read the compiler does not count (cdna guide 5.7 form iii), for either operand type
template <int OFF, typename V>
__device__ __forceinline__ void fa2_read16(V& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// one LDS-DMA piece with a wave-uniform (SGPR) base and a 32-bit per-lane byte offset (conv_sk.hip.h's form: no 64-bit per-lane pointer - the pieces are
// issued between the MFMAs now, where two more live registers per piece spilled at C = 512); M0 = the LDS destination, written inside the statement
__device__ __forceinline__ void fa2_dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// C = 128 * NC channels (NC = 1 .. 4); the head is the whole channel dimension
template <int NC, bool HALF>
__global__ __launch_bounds__(FA_THREADS, 2) void flash_attn2_kernel(const FlashP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C = 128 * NC;
    constexpr int KROW = C * 2;                        // bytes per key row of the K tile
    constexpr int NKS = C / 32;                        // k32 steps of the S phase
    unsigned char* Kl = smem;
    unsigned char* Vl = smem + FA_BK * KROW;
    unsigned char* Pl = Vl + C * (FA_BK * 2);
    float* rowsc = reinterpret_cast<float*>(Pl + FA_BQ * FA_BK * 2);       // [0..127] rescale factor | [128..255] 1 / l
    float* red = rowsc + 256;                                               // statistics scratch

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int b = lid / p.nq, qtile = lid - b * p.nq;
    const int q0 = qtile * FA_BQ;
    const int N = p.N;
    const int ntiles = (N + FA_BK - 1) / FA_BK;
    const bf16_t* qkvb = p.qkv + (long long)b * p.qkv_bstride;
    const bf16_t* vtb = p.vt + (long long)b * p.vt_bstride;

    // ---- tile loaders -----------------------------------------------------------------------------------------
    // Source addresses are a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset that is re-formed at every
    // call from `lane` (the asm barrier): hoisted per-instruction 64-bit pointers cost 32 VGPRs and spilled.
    // K tile: rows of KROW bytes (one wave instruction = 1024 / KROW key rows), physical 16-byte chunk = logical ^ (key & 15)
    const unsigned char* kglob = reinterpret_cast<const unsigned char*>(qkvb + C);
    const unsigned char* vglob = reinterpret_cast<const unsigned char*>(vtb);
    const unsigned krow_bytes = (unsigned)p.ld * 2, vrow_bytes = (unsigned)p.Npad * 2;
    // (round 6: one piece per call - the pieces of K(t + 1) and V't(t) are issued BETWEEN the MFMAs of the PV / S phase instead of as a block of
    // 2 NC instructions per wave at the phase's head.  s_memtime stamps: with every fragment read removed the PV phase still took 4.6 k cycles for 2 k of
    // matrix work - all eight waves issued their eight LDS-DMAs (~100 - 185 cycles each inside such a phase) at the same time and the matrix pipe sat
    // idle meanwhile; spread out, one wave's DMA issue runs under its SIMD partner's MFMAs.)
    constexpr int NPIECE = 2 * NC;                     // DMA instructions per wave for a K tile, and for a V't tile (C / 64)
    auto issue_K_piece = [&](int t, int i) {
        constexpr int LPR = KROW / 16;                 // lanes per key row (64 | 48 | 32 | 16)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int inst = i * 8 + wave;
        if constexpr (LPR == 64) {                     // C = 512: one key row per instruction - the row is wave-uniform and goes into the SGPR base
            int kg = t * FA_BK + inst; kg = kg < N ? kg : N - 1;
            const unsigned char* rowp = kglob + (size_t)__builtin_amdgcn_readfirstlane(kg) * krow_bytes;
            fa2_dma16(rowp, (unsigned)((ln ^ (inst & 15)) << 4), (unsigned)__builtin_amdgcn_readfirstlane(inst * 1024));
        } else {
            const int e = inst * 64 + ln, r = e / LPR, j = e - r * LPR;
            int kg = t * FA_BK + r; kg = kg < N ? kg : N - 1;
            const unsigned off = (unsigned)kg * krow_bytes + (unsigned)((j ^ (r & 15)) << 4);
            fa2_dma16(kglob, off, (unsigned)__builtin_amdgcn_readfirstlane(inst * 1024));                     // (the K tile sits at LDS offset 0)
        }
    };
    auto issue_K = [&](int t) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) issue_K_piece(t, i);
    };
    // V't tile: rows = channels, 64 keys (128 bytes) per row; one instruction = 8 rows; chunk ^= (row >> 1) & 7.
    // (row >> 1) & 7 = 4 (inst & 1) | (lane >> 4) and inst & 1 = wave & 1 for all of a wave's instructions: the per-lane
    // offset is the same for all of them
    auto issue_V_piece = [&](int t, int i) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned lc = (unsigned)((ln & 7) ^ (ln >> 4) ^ ((wave & 1) << 2));
        const unsigned off = (unsigned)(ln >> 3) * vrow_bytes + lc * 16;
        const int inst = i * 8 + wave;
        const unsigned char* rowp = vglob + (size_t)(inst * 8) * vrow_bytes + (size_t)t * (FA_BK * 2);         // wave-uniform
        fa2_dma16(rowp, off, (unsigned)__builtin_amdgcn_readfirstlane(FA_BK * KROW + inst * 1024));
    };

#ifdef UCDIR_TIMING
    const bool dbg_on = p.dbg && (lid == (int)gridDim.x / 2) && (lane == 0) && (wave == 5);
    int dbg_n = 0;
#endif
    FA_STAMP();
    issue_K(0);

    // ---- resident Q fragments of this wave's 16 queries: lane (x = lane & 15, g = lane >> 4) holds
    // Q[q0 + 16 wave + x][32 ks + 8 g .. + 7] -------------------------------------------------------------------
    const int x = lane & 15, g = lane >> 4;
    typedef typename FaVec<HALF>::T vec_t;
    vec_t qf[NKS];
    {
        int qg = q0 + 16 * wave + x; qg = qg < N ? qg : N - 1;
        const bf16_t* qp = qkvb + (long long)qg * p.ld + 8 * g;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const vec_t*>(qp + 32 * ks);
    }

    // ---- per-lane LDS offsets (kept to ONE register each: the swizzle is one v_xor with a literal per access; twelve
    // precomputed offsets spilled at C = 512, and a scratch reload inside the loop would drain the LDS-DMA queue) ----------
    // S phase, K fragment (16x16x32 A operand): key = 16 kt + x, logical chunk 4 ks + g  ->  physical (4 ks) ^ (g ^ x)
    const int kbase = x * KROW, kxor = (g ^ x) << 4;
    // PV phase: rows of 128 bytes, logical chunk 2 k16 + h  ->  physical (2 k16) ^ (h ^ z), z = ((lane & 31) >> 1) & 7
    const int h = lane >> 5, l31 = lane & 31;
    const int qh = wave >> 2, dq = wave & 3;
    const int vbase = l31 * 128, vxor = (h ^ ((l31 >> 1) & 7)) << 4;
    // P write (S phase): query row 16 wave + x, keys 16 kt + 4 g .. + 3 -> chunk 2 kt + (g >> 1), 8-byte half g & 1
    const int pwbase = (16 * wave + x) * 128 + (g & 1) * 8, pwxor = ((g >> 1) ^ ((x >> 1) & 7)) << 4;

    f32x16_t oacc[NC][2];                               // O^T tiles: [channel tile][query tile] (NC x 32 channels per wave)
#pragma unroll
    for (int dt = 0; dt < NC; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[dt][qt][e] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;                // running max (raw scores) and this lane group's partial row sum
    const float c1 = p.scale_log2e;
    constexpr int DPW = 32 * NC;                        // channels per wave in the PV phase

    for (int t = 0; t < ntiles; ++t) {
        FA_STAMP();                                                 // [6k+1] tile start
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // K(t) (and, first time, Q) landed
        FA_STAMP();                                                 // [6k+2] K wait done
        __syncthreads();                                            // ... for every wave; PV(t-1) done: P, V't free
        FA_STAMP();                                                 // [6k+3] barrier A passed
#ifdef FA2_BLOCK_V
        for (int i = 0; i < NPIECE; ++i) issue_V_piece(t, i);
#endif
        // ---- S^T = K Q^T --------------------------------------------------------------------------------------
        f32x4_t sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) sacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // fragment of (ks, kt): byte address ka[ks & 3] + (ks >> 2) 256 + kt 16 KROW - four per-lane registers (the swizzle XOR), the rest immediates.
        // Inline-asm reads with counted lgkmcnt (hipcc's own bookkeeping drains the LDS queue - lgkmcnt(0) - every few MFMAs: flash_attn.hip.h
        // runs this phase at ~50 cycles per 16-cycle MFMA).
        // (kbase has its low eight bits clear and kxor lives in bits 4 - 7: kbase + ((64 c4) ^ kxor) = (kbase | kxor) ^ (64 c4): ONE register and one
        // v_xor per k step instead of four address registers; dynamic LDS starts at byte 0 - no static __shared__ - and the K tile comes first)
        int kx = kxor;
        asm volatile("" : "+v"(kx));
        const unsigned kb2 = (unsigned)kbase | (unsigned)kx;
        auto ka = [&](int c4) { return kb2 ^ (unsigned)(64 * c4); };
        // one fragment set: the register of (ks, kt) is re-requested for (ks + 1, kt) right behind its MFMA - four reads always in flight,
        // each three MFMAs ahead of its use (a second set, 16 more registers, spilled at C = 512)
        vec_t kf[4];
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // nothing of the compiler's own is queued behind this
        static_for<0, 4>([&](auto ktc) { constexpr int kt = decltype(ktc)::value; fa2_read16<kt * 16 * KROW>(kf[kt], kb2); });
        __builtin_amdgcn_s_setprio(1);
        static_for<0, NKS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            static_for<0, 4>([&](auto ktc) {
                constexpr int kt = decltype(ktc)::value;
                lgkm_wait_asm<(ks + 1 < NKS) ? 3 : 3 - kt>();
                sacc[kt] = fa_mfma16(kf[kt], qf[ks], sacc[kt]);
                if constexpr (ks + 1 < NKS) fa2_read16<((ks + 1) >> 2) * 256 + kt * 16 * KROW>(kf[kt], ka((ks + 1) & 3));
            });
#ifndef FA2_BLOCK_V
            // (measured against this placement, same box, N = 1296 / B = 16: every second step over the whole phase 100.5 us, two pieces per step over
            // the first quarter 100.0, this 95.6, the block at the phase's head 103.0)
            if constexpr (ks < NPIECE) issue_V_piece(t, ks);        // V't(t): one piece behind each of the first 2 NC k steps (of 4 NC)
#endif
        });
        __builtin_amdgcn_s_setprio(0);
#ifdef UCDIR_TIMING
        asm volatile("" :: "v"(sacc[0][0]), "v"(sacc[3][3]));
#endif
        FA_STAMP();                                                 // [6k+4] S phase done
        // ---- online softmax of this wave's 16 rows ---------------------------------------------------------------
        if (t == ntiles - 1 && (N & (FA_BK - 1))) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * FA_BK + 16 * kt + 4 * g + r >= N) sacc[kt][r] = -3.0e38f;
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kt][r]);
        // the row's 64 keys sit in lanes x, x + 16, x + 32, x + 48: two VALU row / half swaps instead of two ds_bpermute round trips
        { float u = mx, v = mx; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(v)); mx = fmaxf(u, v); }
        { float u = mx, v = mx; permlane32_swap(u, v); mx = fmaxf(u, v); }
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c1);
        const float mc = m_new * c1;
        m_run = m_new;
        float psum = 0.f;
        int pwx = pwxor;
        asm volatile("" : "+v"(pwx));          // formed here: the four variants are not kept live across the S phase
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(sacc[kt][r] * c1 - mc); psum += pv[r]; }
            { u32x2_t pk; pk[0] = fa_pack2<HALF>(pv[0], pv[1]); pk[1] = fa_pack2<HALF>(pv[2], pv[3]); *reinterpret_cast<u32x2_t*>(Pl + pwbase + ((32 * kt) ^ pwx)) = pk; }
        }
        l_run = l_run * alpha + psum;
        if (g == 0) rowsc[16 * wave + x] = alpha;
        FA_STAMP();                                                 // [6k+5] softmax + P write done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // V't(t) landed
        __syncthreads();                                            // P, alpha visible; every wave done with K(t)
        FA_STAMP();                                                 // [6k+6] V wait + barrier B passed
        const bool more = t + 1 < ntiles;
        const int tnext = more ? t + 1 : t;
#ifdef FA2_BLOCK_K
        if (more) issue_K(t + 1);
#endif
        // ---- O^T = alpha O^T + V't P^T ----------------------------------------------------------------------------
        // the rescale decision first (its two LDS reads are the compiler's own: waited for before the uncounted fragment reads go out)
        const float a0 = rowsc[64 * qh + l31], a1 = rowsc[64 * qh + 32 + l31];
        const bool rescale = !__all(a0 == 1.0f && a1 == 1.0f);      // a row's max moved (exact; uniform)
        // fragments by inline-asm reads with counted lgkmcnt, one k16 step (2 P + NC V't fragments) ahead: the reads of step k16 + 1 go out in
        // front of the MFMAs of step k16 (the compiler's version: reads, lgkmcnt(0), MFMAs - the phase ran at ~150 cycles per 32-cycle MFMA)
        // (both bases are multiples of 128 and vxor lives in bits 4 - 6: base + ((32 k16) ^ vxor) = (base | vxor) ^ (32 k16))
        int vx = vxor;
        asm volatile("" : "+v"(vx));                                // formed here: not two more registers live across the S phase
        const unsigned pa = ((unsigned)(FA_BK * KROW + C * (FA_BK * 2)) + (unsigned)vbase + (unsigned)((64 * qh) * 128)) | (unsigned)vx;
        const unsigned va = ((unsigned)(FA_BK * KROW) + (unsigned)vbase + (unsigned)(DPW * dq * 128)) | (unsigned)vx;
        // P fragments double-buffered (step k16 + 1's two go out at the top of step k16), each V't register re-requested for step k16 + 1 right
        // behind its two MFMAs of step k16 (a second V't set - 16 more registers at C = 512 - spilled)
        vec_t pf[2][2], vf[NC];
        auto pfrag = [&](auto kc, vec_t (&pd)[2]) {
            constexpr int k16 = decltype(kc)::value;
            fa2_read16<0>(pd[0], pa ^ (unsigned)(32 * k16)); fa2_read16<32 * 128>(pd[1], pa ^ (unsigned)(32 * k16));
        };
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pfrag(std::integral_constant<int, 0>{}, pf[0]);
        static_for<0, NC>([&](auto dc) { constexpr int dt = decltype(dc)::value; fa2_read16<32 * dt * 128>(vf[dt], va); });
        if (rescale) {                                              // under the reads' flight
#pragma unroll
            for (int dt = 0; dt < NC; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { oacc[dt][0][e] *= a0; oacc[dt][1][e] *= a1; }
        }
#ifndef FA2_ABL_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        static_for<0, 4>([&](auto kc) {
            constexpr int k16 = decltype(kc)::value;
#ifndef FA2_ABL_PVNOREAD
            if constexpr (k16 + 1 < 4) pfrag(std::integral_constant<int, k16 + 1>{}, pf[(k16 + 1) & 1]);
#endif
            static_for<0, NC>([&](auto dc) {
                constexpr int dt = decltype(dc)::value;
                // younger than V't(k16, dt): the rest of this step's V't, the next step's two P, the next step's V't requested so far
#ifndef FA2_ABL_PVNOREAD
                lgkm_wait_asm<(k16 + 1 < 4) ? NC + 1 : NC - 1 - dt>();
#else
                if constexpr (k16 == 0 && dt == 0) lgkm_wait_asm<0>();
#endif
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = fa_mfma32(vf[dt], pf[k16 & 1][qt], oacc[dt][qt]);
#ifndef FA2_ABL_PVNOREAD
                if constexpr (k16 + 1 < 4) fa2_read16<32 * dt * 128>(vf[dt], va ^ (unsigned)(32 * (k16 + 1)));
#endif
#ifndef FA2_BLOCK_K
                // K(t + 1): behind the first 2 NC of the 4 NC MFMA pairs (behind the last tile: that tile again, into the dead K buffer - no branch
                // around the pieces: with one hipcc kept all eight offsets live across the loop and spilled them)
                if constexpr (k16 * NC + dt < NPIECE) issue_K_piece(tnext, k16 * NC + dt);
#endif
            });
        });
        __builtin_amdgcn_s_setprio(0);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the pieces issued behind the last tile)
#ifdef UCDIR_TIMING
    asm volatile("" :: "v"(oacc[0][0][0]), "v"(oacc[NC - 1][1][15]));
    FA_STAMP();
    if (dbg_on) p.dbg[255] = dbg_n;
#endif
    // ---- epilogue ----------------------------------------------------------------------------------------------------
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    __syncthreads();                                                // last PV done reading rowsc
    if (g == 0) rowsc[128 + 16 * wave + x] = 1.0f / l_run;
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int ql = 64 * qh + 32 * qt + l31, qg = q0 + ql;
        const float inv_l = rowsc[128 + ql];
        if (qg < N) {
            const int yy = qg / p.W, xx = qg - yy * p.W;
            const long long pix = ((long long)(yy + 1) * (p.W + 2) + xx + 1) * C;
            const bf16_t* rp = p.res + (long long)b * p.res_bstride + pix;
            bf16_t* op = p.out + (long long)b * p.out_bstride + pix;
#pragma unroll
            for (int dt = 0; dt < NC; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int c = DPW * dq + 32 * dt + 8 * rq + 4 * h;
                    const float4 bs = *reinterpret_cast<const float4*>(p.bias + c);
                    const uint2 rv = *reinterpret_cast<const uint2*>(rp + c);
                    float v[4];
                    v[0] = oacc[dt][qt][4 * rq + 0] * inv_l + bs.x + bf2f((bf16_t)(rv.x & 0xffffu));
                    v[1] = oacc[dt][qt][4 * rq + 1] * inv_l + bs.y + bf2f((bf16_t)(rv.x >> 16));
                    v[2] = oacc[dt][qt][4 * rq + 2] * inv_l + bs.z + bf2f((bf16_t)(rv.y & 0xffffu));
                    v[3] = oacc[dt][qt][4 * rq + 3] * inv_l + bs.w + bf2f((bf16_t)(rv.y >> 16));
#pragma unroll
                    for (int i = 0; i < 4; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
                    *reinterpret_cast<uint2*>(op + c) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
                }
        }
    }
    if (p.stats_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
        __syncthreads();
        if (tid == 0) {
            float t1 = 0.f, t2 = 0.f;
            for (int w = 0; w < 8; ++w) { t1 += red[w * 2]; t2 += red[w * 2 + 1]; }
            stat_add(p.stats_out, b, t1, t2);
        }
    }
}
