// AKGM tail for C = 64 (8 channels per group, K = 9 taps x 8 = 72): the full-resolution level,
// where the generic implicit-GEMM path wastes most of its time (K padded 72 -> 128, B tile
// re-staged per group, one workgroup per CU).  Here one 256-thread workgroup owns 128 consecutive
// pixel positions and
//   * stages the 3 x 130-pixel x 64-channel halo of swish(conv1) into LDS ONCE (full 128-byte
//     lines via global_load_lds, chunk index XOR-swizzled with (pixel>>1)&7 -> conflict-free
//     ds_read_b128 for any tap shift),
//   * loops over the 8 groups: A_g (64 rows x 80 k, 10 KB) streams through a double buffer, every
//     tap's B fragment is read straight from the halo tile (lanes 0-31 take tap 2k, lanes 32-63
//     tap 2k+1 of v_mfma_f32_32x32x16_bf16), 10 MFMAs per wave per group,
//   * reduces the 8 kernel sets per feature in registers with the per-pixel modulation, and
//     finishes (swish, residual, GroupNorm partial sums, 16-byte stores) from a small LDS stage.
// Reference: model/ucdir.py:129-140 (norm2, spdyconv, modulation sum, swish, residual).
#pragma once
#include "cgemm.hip.h"

struct Akgm64P {
    const bf16_t* A;                         // [8][64][80] bf16, rows permuted like pack_akgm
    const bf16_t* h; long long h_bstride;    // swish(conv1), zero-bordered NHWC, C = 64
    int H, W, Wp, p0, pn, tiles, nbatch;
    const double* stats; double inv_count;   // GroupNorm (norm2) statistics of h
    const float* Tc;                         // [B][9][512]: bias + Tb - mean*rstd*Tg (akgm_tc_kernel)
    const float* G; long long g_bstride; const float* attw;
    const bf16_t* res; long long res_bstride;
    bf16_t* out; long long out_bstride;
    float* partials; int npart;
};

#define A64_RUN 16640            // 130 pixels x 128 B
#define A64_HALO (3 * A64_RUN)   // 49920
#define A64_AROW 176             // 80 k x 2 B + 16 B pad: conflict-free fragment reads
#define A64_ABUF (64 * A64_AROW) // 11264 = 11 x 1024
#define A64_STG 9                // floats per pixel in the stage (8 features + 1 pad)
#define A64_LDS (A64_HALO + 2 * A64_ABUF + 2 * 128 * A64_STG * 4 + 64)

// Tc[b][cls][o] = bias[o] + Tb[cls][o] - mean_b * rstd_b * Tg[cls][o]
__global__ void akgm_tc_kernel(const double* __restrict__ stats, double inv_count, const float* __restrict__ bias,
                               const float* __restrict__ Tb, const float* __restrict__ Tg, int n, float* __restrict__ Tc) {
    const int b = blockIdx.y, cls = blockIdx.x;
    double m = stats[b * 2] * inv_count;
    double var = stats[b * 2 + 1] * inv_count - m * m;
    if (var < 0) var = 0;
    const float mr = (float)m * (float)(1.0 / sqrt(var + 1e-5));
    for (int o = threadIdx.x; o < n; o += blockDim.x)
        Tc[((long long)b * 9 + cls) * n + o] = bias[o] + Tb[(long long)cls * n + o] - mr * Tg[(long long)cls * n + o];
}

__global__ __launch_bounds__(CG_THREADS, 2) void akgm64_kernel(const Akgm64P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* abuf = smem + A64_HALO;
    float* stage = reinterpret_cast<float*>(smem + A64_HALO + 2 * A64_ABUF);
    float* scal = reinterpret_cast<float*>(smem + A64_HALO + 2 * A64_ABUF + 2 * 128 * A64_STG * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tcol = lid % p.tiles, b = lid / p.tiles;
    const int t0 = p.p0 + tcol * CG_TP;
    const int plast = p.p0 + p.pn - 1;

    // ---- halo: 3 runs of 130 pixels starting at t0 - 1 + (r-1)*Wp -------------------------------
    {
        const int sample_last = (p.H + 2) * p.Wp - 1;
        const bf16_t* hb = p.h + (long long)b * p.h_bstride;
        for (int it = wave; it < 51; it += 4) {
            const int r = it / 17, i8 = it - r * 17;
            const int pxl = i8 * 8 + (lane >> 3);
            if (pxl < 130) {
                int pos = t0 - 1 + (r - 1) * p.Wp + pxl;
                pos = pos > sample_last ? sample_last : pos;
                const int j = (lane & 7) ^ ((pxl >> 1) & 7);
                stage16(hb + (long long)pos * 64 + j * 8, halo + r * A64_RUN + i8 * 1024, lane);
            }
        }
    }
    auto issue_A = [&](int g, int buf) {
        const bf16_t* Ag = p.A + (long long)g * 64 * 80;
        for (int k = wave; k < 11; k += 4) {
            const int n = k * 64 + lane;
            const int row = n / 11;
            int ch = n - row * 11;
            ch = ch == 10 ? 0 : ch;                       // pad slot of the 176-byte row
            stage16(Ag + row * 80 + ch * 8, abuf + buf * A64_ABUF + k * 1024, lane);
        }
    };
    issue_A(0, 0);

    // ---- per-lane pixel constants ---------------------------------------------------------------
    const int px = wave * 32 + (lane & 31);
    const int hh = lane >> 5;
    int cp = t0 + px;
    cp = cp < plast ? cp : plast;
    int y = cp / p.Wp, x = cp - y * p.Wp;
    const bool valid = (y >= 1 && y <= p.H && x >= 1 && x <= p.W) && (t0 + px <= plast);
    y = y < 1 ? 1 : (y > p.H ? p.H : y);
    x = x < 1 ? 1 : (x > p.W ? p.W : x);
    const int cls = (y == 1 ? 0 : (y == p.H ? 2 : 1)) * 3 + (x == 1 ? 0 : (x == p.W ? 2 : 1));
    float rstd;
    {
        double m = p.stats[b * 2] * p.inv_count;
        double var = p.stats[b * 2 + 1] * p.inv_count - m * m;
        if (var < 0) var = 0;
        rstd = (float)(1.0 / sqrt(var + 1e-5));
    }
    float att[8], attr[8];
    {
        const float* gp = p.G + (long long)b * p.g_bstride + ((long long)(y - 1) * p.W + (x - 1)) * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        const float* aw = p.attw + b * 8;
        att[0] = g0.x * aw[0]; att[1] = g0.y * aw[1]; att[2] = g0.z * aw[2]; att[3] = g0.w * aw[3];
        att[4] = g1.x * aw[4]; att[5] = g1.y * aw[5]; att[6] = g1.z * aw[6]; att[7] = g1.w * aw[7];
#pragma unroll
        for (int s = 0; s < 8; ++s) attr[s] = att[s] * rstd;
    }
    const float* tcb = p.Tc + ((long long)b * 9 + cls) * 512;

    float s1 = 0.f, s2 = 0.f;
    auto phase2 = [&](int g) {          // threads 0..127: one pixel, the 8 features of group g
        if (tid < 128) {
            const int c2 = t0 + tid;
            if (c2 <= plast) {
                const int yy = c2 / p.Wp, xx = c2 - yy * p.Wp;
                if (yy >= 1 && yy <= p.H && xx >= 1 && xx <= p.W) {
                    const float* sp = stage + (g & 1) * 128 * A64_STG + tid * A64_STG;
                    const long long off = (long long)c2 * 64 + g * 8;
                    const uint4 rv = *reinterpret_cast<const uint4*>(p.res + (long long)b * p.res_bstride + off);
                    const bf16_t* rh = reinterpret_cast<const bf16_t*>(&rv);
                    uint4 ov; bf16_t* oh = reinterpret_cast<bf16_t*>(&ov);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = silu_f(sp[i]) + bf2f(rh[i]);
                        s1 += v; s2 += v * v;
                        oh[i] = f2bf(v);
                    }
                    *reinterpret_cast<uint4*>(p.out + (long long)b * p.out_bstride + off) = ov;
                }
            }
        }
    };

    for (int g = 0; g < 8; ++g) {
        const int buf = g & 1;
#ifndef UCDIR_REGSTAGE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        if (g + 1 < 8) issue_A(g + 1, buf ^ 1);
        if (g > 0) phase2(g - 1);
        f32x16_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const unsigned char* Ab = abuf + buf * A64_ABUF;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            int tap = 2 * kk + hh;
            tap = tap > 8 ? 8 : tap;                      // k chunk 9 is zero in A
            const int ky = tap_ky(tap), kx = tap - 3 * ky;
            const int pxl = px + kx;
            const bf16x8_t bfr = *reinterpret_cast<const bf16x8_t*>(halo + ky * A64_RUN + pxl * 128 + ((g ^ ((pxl >> 1) & 7)) << 4));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(Ab + (t * 32 + (lane & 31)) * A64_AROW + (2 * kk + hh) * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
            }
        }
        float* sg = stage + buf * 128 * A64_STG + px * A64_STG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int floc = 4 * t + 2 * q + hh;
                const float* tc = tcb + 8 * (g * 8 + floc);
                const float4 c0 = *reinterpret_cast<const float4*>(tc), c1 = *reinterpret_cast<const float4*>(tc + 4);
                const float tcv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) sum += attr[s] * acc[t][8 * q + s] + att[s] * tcv[s];
                sg[floc] = valid ? sum : 0.f;
            }
    }
    __syncthreads();
    phase2(7);
    if (p.partials) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        if (lane == 0) { scal[wave * 2] = s1; scal[wave * 2 + 1] = s2; }
        __syncthreads();
        if (tid == 0) {
            float* pp = p.partials + ((long long)b * p.npart + tcol) * 2;
            pp[0] = scal[0] + scal[2] + scal[4] + scal[6];
            pp[1] = scal[1] + scal[3] + scal[5] + scal[7];
        }
    }
}
