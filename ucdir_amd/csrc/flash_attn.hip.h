// Single-head self-attention with a 512-wide (C = 128 .. 512) head as ONE kernel (gfx950):
//   y[i][:] = sum_j softmax_j(q_i . k_j / sqrt(C)) v'_j + bias + x_i          (model/ucdir.py:165-182)
// q, k come from the compact token tensor qkv [B][N][3C] (q at 0, k at C), v' = W_o W_v GN(x) arrives transposed,
// V't [B][C][Npad] (fold_out_into_v in engine.hip: the out-projection is already inside the value rows).
// The score matrix is never materialised (the reference writes B x N x N fp32: 1 GiB per sample at the 1024^2
// patch windows): online softmax, fp32 running max / sum, exact rescale only when a row's max moved.
//
// Workgroup = 512 threads (8 wave64) = 128 queries of one sample; KV tiles of 64 keys.
//   S phase  : wave w owns queries 16w .. 16w+15 against all 64 keys: S^T = K Q^T on v_mfma_f32_16x16x32
//              (A = K rows from LDS, B = the wave's Q fragments, RESIDENT in 16 x 4 VGPRs for the whole kernel), so a
//              lane holds 16 keys of ONE query: the row max / sum are in-lane + two wave shuffles, nothing crosses
//              waves.  P (bf16 | fp16) goes to LDS together with the row's rescale factor.
//   PV phase : wave (qh, dq) owns queries 64 qh .. +63 x channels 128 dq .. +127 of O^T = V't P^T on
//              v_mfma_f32_32x32x16 (A = V't rows, B = P rows): 8 tiles = 128 accumulator registers.
//   LDS      : K tile [64 keys][C] (64 KB) | V't tile [C][64 keys] (64 KB) | P [128][64] (16 KB) | row scalars.
//              Both tiles arrive by LDS-DMA (global_load_lds, 16 B / lane) with the XOR swizzle on the SOURCE
//              address; K(t+1) flies under PV(t), V't(t) under S(t): two barriers per KV tile.  (Requesting K(t+1) a phase
//              earlier - third barrier after the S loop, counted vmcnt before PV - was measured: 5 % slower at N = 1296 and
//              at N = 16384; K's latency is already covered by the PV phase, the barrier is not free.)
// Epilogue: O / l + bias + residual -> zero-bordered NHWC bf16 + GroupNorm statistics of the output (stat_add).
// Registers: O 128 + Q 64 of the 256 a wave has at two waves per SIMD; the S phase is software-pipelined by hand (the
// fragment of step ks+1 is requested as soon as the MFMA of step ks has consumed its register: four LDS reads in
// flight) and per-tile LDS offsets are re-formed per tile instead of living across phases - one spilled register
// would put a scratch reload (a VMEM wait) into the loop and drain the LDS-DMA queue.
// hipcc (ROCm 7.2) detail this kernel depends on: an LDS access whose memory operand carries no TBAA tag (a uint4 /
// uint2 struct load, a bit_cast-wrapped load) is made to wait vmcnt(0) for every LDS-DMA in flight ("may alias"),
// a typed ext_vector load / store is not - so every LDS fragment access below uses the MFMA operand vector types.
// HALF selects fp16 operands (BASELINE configs[4] "fp16 attention MFMA path"): qkv / V't / P are then IEEE half.
#pragma once
#include "cgemm.hip.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

struct FlashP {
    const bf16_t* qkv; long long qkv_bstride; int ld;      // [B][N][ld], q at 0, k at C
    const bf16_t* vt; long long vt_bstride; int Npad;      // [B][C][Npad]
    int N, C, W;                                            // tokens, channels, image width (token n = pixel (n / W, n % W))
    float scale_log2e;                                      // log2(e) / sqrt(C)
    const float* bias;                                      // [C]
    const bf16_t* res; long long res_bstride;               // x, zero-bordered NHWC
    bf16_t* out; long long out_bstride;                     // y, zero-bordered NHWC
    stat_t* stats_out;
    int nq;                                                 // query tiles per sample
    unsigned long long* dbg;                                // UCDIR_TIMING builds: s_memtime stamps of one wave
};

#ifdef UCDIR_TIMING
#define FA_STAMP() do { if (dbg_on && dbg_n < 250) p.dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FA_STAMP() do {} while (0)
#endif

#define FA_BQ 128
#define FA_BK 64
#define FA_THREADS 512

__host__ __device__ constexpr int fa_lds_bytes(int C) { return FA_BK * C * 2 * 2 + FA_BQ * FA_BK * 2 + 2048; }

template <bool HALF> struct FaVec { typedef bf16x8_t T; };
template <> struct FaVec<true> { typedef f16x8_t T; };
__device__ __forceinline__ f32x4_t fa_mfma16(const bf16x8_t& a, const bf16x8_t& b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4_t fa_mfma16(const f16x8_t& a, const f16x8_t& b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t fa_mfma32(const bf16x8_t& a, const bf16x8_t& b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t fa_mfma32(const f16x8_t& a, const f16x8_t& b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <bool HALF>
__device__ __forceinline__ uint32_t fa_pack2(float lo, float hi) {
    if constexpr (HALF) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        h2 v; v[0] = (_Float16)lo; v[1] = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    } else return pack2_bf16(lo, hi);
}

// C = 128 * NC channels (NC = 1 .. 4); the head is the whole channel dimension
template <int NC, bool HALF>
__global__ __launch_bounds__(FA_THREADS, 2) void flash_attn_kernel(const FlashP p) {
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
    auto issue_K = [&](int t) {
        constexpr int LPR = KROW / 16;                 // lanes per key row (64 | 48 | 32 | 16)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < (FA_BK * LPR / 64) / 8; ++i) {
            const int inst = i * 8 + wave;
            const int e = inst * 64 + ln, r = e / LPR, j = e - r * LPR;      // LPR = 64: r = inst (uniform), j = lane
            int kg = t * FA_BK + r; kg = kg < N ? kg : N - 1;
            const unsigned off = (unsigned)kg * krow_bytes + (unsigned)((j ^ (r & 15)) << 4);
            stage16(reinterpret_cast<const bf16_t*>(kglob + off), Kl + inst * 1024, lane);
        }
    };
    // V't tile: rows = channels, 64 keys (128 bytes) per row; one instruction = 8 rows; chunk ^= (row >> 1) & 7.
    // (row >> 1) & 7 = 4 (inst & 1) | (lane >> 4) and inst & 1 = wave & 1 for all of a wave's instructions: the per-lane
    // offset is the same for all of them
    auto issue_V = [&](int t) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned lc = (unsigned)((ln & 7) ^ (ln >> 4) ^ ((wave & 1) << 2));
        const unsigned off = (unsigned)(ln >> 3) * vrow_bytes + lc * 16;
#pragma unroll
        for (int i = 0; i < C / 8 / 8; ++i) {
            const int inst = i * 8 + wave;
            const unsigned char* rowp = vglob + (size_t)(inst * 8) * vrow_bytes + (size_t)t * (FA_BK * 2);     // wave-uniform
            stage16(reinterpret_cast<const bf16_t*>(rowp + off), Vl + inst * 1024, lane);
        }
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
#ifdef FA_ABL_NODMA
        if (t == 0)
#endif
        issue_V(t);
        // ---- S^T = K Q^T --------------------------------------------------------------------------------------
        f32x4_t sacc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) sacc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // software pipeline: the fragment of (ks + 1, kt) is requested right after the MFMA of (ks, kt) has consumed its
        // register: four LDS reads stay in flight under four MFMAs (the compiler alone kept ONE read in flight)
        auto kaddr = [&](int ks, int kt) { return Kl + kbase + ((64 * (ks & 3)) ^ kxor) + (ks >> 2) * 256 + kt * (16 * KROW); };
        vec_t kf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kf[kt] = *reinterpret_cast<const vec_t*>(kaddr(0, kt));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#ifndef FA_ABL_NOS
                sacc[kt] = fa_mfma16(kf[kt], qf[ks], sacc[kt]);
#else
                if (ks == 0) sacc[kt] = fa_mfma16(kf[kt], qf[ks], sacc[kt]);
#endif
#ifndef FA_ABL_NOSREAD
                if (ks + 1 < NKS) kf[kt] = *reinterpret_cast<const vec_t*>(kaddr(ks + 1, kt));
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks + 1 < NKS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
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
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
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
#ifndef FA_ABL_NODMA
        if (t + 1 < ntiles) issue_K(t + 1);
#endif
        // ---- O^T = alpha O^T + V't P^T ----------------------------------------------------------------------------
        {
            const float a0 = rowsc[64 * qh + l31], a1 = rowsc[64 * qh + 32 + l31];
            if (!__all(a0 == 1.0f && a1 == 1.0f)) {                 // a row's max moved: rescale (exact; uniform branch)
#pragma unroll
                for (int dt = 0; dt < NC; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { oacc[dt][0][e] *= a0; oacc[dt][1][e] *= a1; }
            }
        }
        int vx = vxor;
        asm volatile("" : "+v"(vx));
        // (Re-requesting each fragment register for step k16 + 1 right behind its last MFMA of step k16 - the S phase's trick,
        // no extra registers - shortens this phase in the s_memtime stamps, 4780 -> 4160 cycles, and leaves the kernel's
        // duration where it was at N = 1296 and N = 16384: with two waves per SIMD the phase is bound by the MFMA / LDS pipes
        // the other wave shares, not by this wave's read latency.  Kept in the plain form.)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
#ifdef FA_ABL_NOPV
        for (int k16 = 0; k16 < 1; ++k16) {
#else
        for (int k16 = 0; k16 < 4; ++k16) {
#endif
            vec_t pf[2], vf[NC];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) pf[qt] = *reinterpret_cast<const vec_t*>(Pl + vbase + ((32 * k16) ^ vx) + (64 * qh + 32 * qt) * 128);
#pragma unroll
            for (int dt = 0; dt < NC; ++dt) vf[dt] = *reinterpret_cast<const vec_t*>(Vl + vbase + ((32 * k16) ^ vx) + (DPW * dq + 32 * dt) * 128);
#pragma unroll
            for (int dt = 0; dt < NC; ++dt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = fa_mfma32(vf[dt], pf[qt], oacc[dt][qt]);
        }
        __builtin_amdgcn_s_setprio(0);
    }

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
