// Shared device/host definitions for libucdir_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// round-to-nearest-even float -> bf16 (inputs are finite on this path)
__host__ __device__ inline bf16_t f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ inline float bf2f(bf16_t h) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)h) << 16;
    return v.f;
}

// two fp32 -> packed bf16x2 in one instruction (v_cvt_pk_bf16_f32: round-to-nearest-even, same result as f2bf).
// Written as native casts, NOT inline asm: hipcc selects the same instruction, and it also pads the hazards around it -
// an asm statement that reads the result of a transcendental (v_exp_f32 in the flash-attention softmax) with no
// instruction in between got a stale operand (gfx950 TRANS -> VALU use needs one wait state; nothing inside or in front
// of an asm statement is padded, cdna guide §5.7).  Found by test_attention at C = 128.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    bf16x2_t v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint4 pack8_bf16(const float* v) {
    return make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
}
// two fp32 -> packed IEEE half x2 (round-to-nearest-even)
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    h2_t v; v[0] = (_Float16)lo; v[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint4 pack8_f16(const float* v) {
    return make_uint4(pack2_f16(v[0], v[1]), pack2_f16(v[2], v[3]), pack2_f16(v[4], v[5]), pack2_f16(v[6], v[7]));
}

// v_permlane32_swap_b32: lanes 32-63 of `a` swap with lanes 0-31 of `b` (the other two halves stay).  Inline asm on
// purpose: with hipcc 7.2 the builtin (__builtin_amdgcn_permlane32_swap) loses its SECOND result when several swaps feed
// an array - the generated code reuses result 0 for both (reproduced standalone: four swaps, eight stores, registers
// v1/v3 stored twice, v2/v4 never).  The two wait states a VALU write -> permlane read needs are inside the string.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ inline float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// x / d for small non-negative ints (x < 2^15, d <= 1024) with inv = 1.0f / d: the +0.5 keeps the product at least
// 0.5/d away from an integer, far more than the fp32 rounding error, so the truncation is exact (3 VALU, no division)
__device__ __forceinline__ int fdiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// GroupNorm(1 group) mean / rstd of one sample from its fp64 (sum, sum of squares): the cancellation-prone
// part stays in fp64 (3 operations), the reciprocal square root is v_rsq_f32 + one Newton step (<= 1 ulp)
__device__ __forceinline__ void mean_rstd(double S, double Q, double inv_count, float& mean, float& rstd) {
    const double m = S * inv_count;
    float v = (float)(Q * inv_count - m * m);
    v = (v > 0.f ? v : 0.f) + 1e-5f;
    float r = __builtin_amdgcn_rsqf(v);
    r = r * (1.5f - 0.5f * v * r * r);
    mean = (float)m; rstd = r;
}

// swish with v_rcp_f32 instead of an IEEE divide (1 ulp; the result is rounded to bf16 anyway)
// act codes of the epilogues: 0 none, 1 swish, 2 LeakyReLU(0.2) as max(0.2x, x) (model/ucdir.py:414-416)
__device__ inline float act_apply(float v, int act);
__device__ inline float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ inline float act_apply(float v, int act) { return act == 1 ? silu_fast(v) : (act == 2 ? fmaxf(0.2f * v, v) : v); }

// GroupNorm statistics of an activation: per-sample (sum, sum of squares) kept as 2^-20 FIXED POINT in 64-bit
// integers.  Every producer workgroup adds its fp32 partial with one fire-and-forget integer atomic per value:
// integer addition is associative, so the result is bit-identical whatever order the workgroups retire in (an
// fp64 atomic would not be), nobody waits on anything, and the per-activation reduction launch
// (69 per forward) disappears.  Resolution 1e-6 absolute against sums of 1e3..1e7;
// range +-8.8e12.  The accumulators of all activations sit in one slab that forward() zeroes with one memset.
// Device-scope atomics on one address serialise at ~150 ns each on this chip (10,368 workgroups on 32 addresses
// cost a short kernel 47 us), so every sample has UCDIR_STAT_SLOTS accumulator pairs, a workgroup adds to slot
// blockIdx.x % SLOTS and readers sum the slots (integers: still order-independent).  Layout [B][SLOTS][2].
typedef long long stat_t;
#define UCDIR_STAT_SCALE 1048576.0
#define UCDIR_STAT_SLOTS 16
__device__ __forceinline__ double stat_val(stat_t v) { return (double)v * (1.0 / UCDIR_STAT_SCALE); }
__device__ __forceinline__ stat_t stat_fx(double v) { return __double2ll_rn(v * UCDIR_STAT_SCALE); }
__device__ __forceinline__ void stat_add(stat_t* stats, int b, float t1, float t2) {      // call from ONE thread of the workgroup
    stat_t* dst = stats + ((long long)b * UCDIR_STAT_SLOTS + (blockIdx.x % UCDIR_STAT_SLOTS)) * 2;
    (void)__hip_atomic_fetch_add(dst, stat_fx((double)t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_add(dst + 1, stat_fx((double)t2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (sum, sum of squares) of sample b, optionally of two tensors together (channel concatenation)
__device__ __forceinline__ void stat_read(const stat_t* s0, const stat_t* s1, int b, double& S, double& Q) {
    stat_t a = 0, q = 0;
#pragma unroll
    for (int k = 0; k < UCDIR_STAT_SLOTS; ++k) {
        a += s0[((long long)b * UCDIR_STAT_SLOTS + k) * 2]; q += s0[((long long)b * UCDIR_STAT_SLOTS + k) * 2 + 1];
        if (s1) { a += s1[((long long)b * UCDIR_STAT_SLOTS + k) * 2]; q += s1[((long long)b * UCDIR_STAT_SLOTS + k) * 2 + 1]; }
    }
    S = stat_val(a); Q = stat_val(q);
}
// (mean, rstd) of sample b by ONE WAVE: lane l < 2 SLOTS reads accumulator l (even lanes sums, odd lanes sums of squares); integer sums -
// the same bits as stat_read's serial order - then mean_rstd on the same two doubles
__device__ __forceinline__ void stat_mean_rstd_wave(const stat_t* stats, int b, double inv_count, int lane, float& mean, float& rstd) {
    stat_t v = lane < 2 * UCDIR_STAT_SLOTS ? stats[(long long)b * (2 * UCDIR_STAT_SLOTS) + lane] : 0;
#pragma unroll
    for (int off = 2; off < 2 * UCDIR_STAT_SLOTS; off <<= 1) v += __shfl_xor(v, off);
    const stat_t a = __shfl(v, 0), q = __shfl(v, 1);
    mean_rstd(stat_val(a), stat_val(q), inv_count, mean, rstd);
}
// direct (non-atomic) write of a sample's totals: slot 0 carries them, the other slots are cleared
__device__ __forceinline__ void stat_store(stat_t* stats, int b, double S, double Q) {
    stat_t* d = stats + (long long)b * UCDIR_STAT_SLOTS * 2;
    d[0] = stat_fx(S); d[1] = stat_fx(Q);
    for (int k = 2; k < 2 * UCDIR_STAT_SLOTS; ++k) d[k] = 0;
}

// Activation tensor in HBM: zero-bordered NHWC bf16, [B][H+2][W+2][C].  The one-pixel zero
// border makes every 3x3 tap of an interior pixel an in-bounds read of the right value, so the
// implicit GEMM needs no per-tap predication.  Borders are zeroed once and never written.
struct Act {
    bf16_t* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    stat_t* stats = nullptr;       // [B][SLOTS][2] (sum, sum of squares) over the valid region, fixed point (stat_read)
    __host__ __device__ int Hp() const { return H + 2; }
    __host__ __device__ int Wp() const { return W + 2; }
    long long bstride() const { return (long long)(H + 2) * (W + 2) * C; }
    long long elems() const { return bstride() * B; }
};

enum { COLS_S1 = 0, COLS_DOWN = 1, COLS_UP = 2, COLS_PLAIN = 3 };
enum { EPI_STD = 0, EPI_AKGM = 1 };

// Parameters of one implicit-GEMM launch:  D[row][col] = sum_k A[row][k] * Bm[col][k]
//   rows = output features (weights) or tokens, cols = pixel positions or tokens.
struct GemmP {
    // A operand: row-major [rows][a_ld] bf16, K contiguous
    const bf16_t* A; long long a_bstride; long long a_gstride; int a_ld; int a_rows;
    // B operand: up to two NHWC sources concatenated along channels
    const bf16_t* B0; const bf16_t* B1; long long b0_bstride, b1_bstride; int ld0, ld1; int c0;
    int cols_mode; int in_compact;
    int H, W, Wp;            // column grid (valid H x W, padded width Wp); PLAIN: ncols = W
    int Hi, Wi, Wpi;         // input grid (DOWN / UP / compact-in)
    int p0, pn;              // first column position and number of positions per sample
    int ntaps, cpt, cpt_shift, cg;   // taps (1|9), 16-byte chunks per tap, log2(cpt) if cpt<8, channels/group
    int nk;                  // K steps of 64
    int tiles;               // column tiles per sample
    int rowtiles;            // row tiles (grid.y equivalent)
    int groups_per_wg;       // AKGM: groups looped inside one workgroup
    int nbatch;
    int th, tw, tiles_x, tiles_y;   // conv3x3_halo: 2-D pixel tile and tile grid per sample
    int up_phase;                   // conv3x3_halo: nearest-x2 + 3x3 as four 2x2 parity convolutions
    // epilogue
    float alpha; int fold; int act;
    const stat_t* stats0; const stat_t* stats1; double inv_count;   // GN of the input (fold)
    const float* bias; const float* Tb; const float* Tg; int tab_ld;  // tables [ncls][tab_ld]
    const bf16_t* res; long long res_bstride; int res_ld; int res_coff;
    void* out; long long out_bstride; int out_ld; int out_coff; int out_f32; int out_compact;
    int out_f16;                        // bf16-path stores write IEEE half instead (fp16-operand attention, qkv only)
    int out_nchw; int crop_h, crop_w;   // final conv: fp32 NCHW (B, nfeat, crop_h, crop_w)
    int shuffle_c;                      // > 0: ConvTranspose2d(2,2): feature f = q*shuffle_c + o goes to pixel (2y+q/2, 2x+q%2), channel o
    int nfeat;               // valid output features (rows) in total
    stat_t* stats_out;                      // != nullptr: add this launch's (sum, sum of squares) of the output here (stat_add)
    // fused res_conv (conv3x3_halo_kernel<64>): A carries a 10th tap = the block's 1x1 res_conv; second bf16 NHWC output
    int plain_w;                            // COLS_PLAIN only, > 0: column n is pixel (n / plain_w, n % plain_w); output and residual use the zero-bordered layout
    int res_fused; bf16_t* out2; long long out2_bstride; int out2_ld; const float* bias2;
    // AKGM
    const float* G; long long g_bstride;   // guide branch, compact [B][H*W][8]
    const float* attw;                      // [B][8]
    // conv3x3_halo split-K (grids that leave most of the chip idle): ksplit workgroups share a tile, each reduces a range of
    // channel chunks and writes raw fp32 accumulators to partial [wg][ksplit][256 px][TM]; conv_splitk_finish_kernel sums them
    // in a FIXED order (bit-reproducible) and runs the epilogue
    int ksplit; float* partial;
    // conv3x3_halo tail filler: the last alt_blocks workgroups of the grid run the block's 1x1 res_conv on the same input
    // (weights alt_A [rows][alt_a_ld], bias2, output out2): dispatched last, they fill the CUs the 3x3 conv's last partial
    // round of workgroups leaves idle instead of waiting for it in a launch of their own
    int alt_blocks; const bf16_t* alt_A; int alt_a_ld;
    // conv3x3_halo: the same weights pre-tiled as the kernel's LDS stage images, [row tile][32-channel chunk][K step][16 KB]
    // (pack_conv_tiled): every LDS-DMA piece of a weight stage is then 1 KB of CONTIGUOUS memory (eight full lines) instead of
    // sixteen 64-byte half lines.  nullptr: stage from A.
    const bf16_t* A_tiled;
    unsigned long long* dbg;                // UCDIR_TIMING builds: s_memtime stamps of one workgroup
};
