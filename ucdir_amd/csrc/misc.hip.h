// Small HBM-bound kernels around the GEMM core (gfx950).
#pragma once
#include "common.h"

// direct statistics of an activation tensor (tests / inputs produced outside the GEMM core)
__global__ void act_stats_kernel(const bf16_t* __restrict__ x, int H, int W, int C, stat_t* __restrict__ stats) {
    const int b = blockIdx.x;
    const int Wp = W + 2;
    const long long bs = (long long)(H + 2) * Wp * C;
    const int c8 = C / 8;
    double s1 = 0, s2 = 0;
    for (long long i = threadIdx.x; i < (long long)H * W * c8; i += blockDim.x) {
        int c = (int)(i % c8);
        long long pix = i / c8;
        int y = (int)(pix / W), xx = (int)(pix % W);
        const uint4 v = *reinterpret_cast<const uint4*>(x + b * bs + ((long long)(y + 1) * Wp + xx + 1) * C + c * 8);
        const bf16_t* h = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) { float f = bf2f(h[k]); s1 += f; s2 += (double)f * f; }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) stat_store(stats, b, sh[0][0], sh[1][0]);
}

// ------------------------------------------------------------------------------------------------
// layout converters (ABI boundary, tests, debug taps)
// ------------------------------------------------------------------------------------------------
__global__ void nchw_to_act_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B, int C, int H, int W) {
    long long n = (long long)B * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C); long long t = i / C;
        int x = (int)(t % W); t /= W;
        int y = (int)(t % H); int b = (int)(t / H);
        float v = src[(((long long)b * C + c) * H + y) * W + x];
        dst[(((long long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c] = f2bf(v);
    }
}
__global__ void act_to_nchw_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W) {
    long long n = (long long)B * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = (int)(i % W); long long t = i / W;
        int y = (int)(t % H); t /= H;
        int c = (int)(t % C); int b = (int)(t / C);
        dst[i] = bf2f(src[(((long long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c]);
    }
}
// (B,8,H,W) fp32 NCHW -> compact [B][H*W][8] fp32 (AKGM modulation, tests)
__global__ void nchw8_to_compact_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W) {
    long long n = (long long)B * H * W * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int s = (int)(i % 8); long long t = i / 8;
        long long pix = t % ((long long)H * W); int b = (int)(t / ((long long)H * W));
        dst[i] = src[((long long)b * 8 + s) * H * W + pix];
    }
}

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < n ? i : 2 * (n - 1) - i; }

// ------------------------------------------------------------------------------------------------
// Stem on the matrix cores: conv3x3 CIN (6 | 3) -> 64 per block of output channels, straight from the NCHW fp32
// inputs (bottom/right reflect pad of DY3h.forward / UNetSeeInDark by indexing, model/ucdir.py:303-306,354-361).
// K = 9 taps x 8 channel slots (CIN real, rest zero) in the tap-pair layout of akgm_pre.hip.h: k16 step j carries
// tap 2j on lanes 0-31 and tap 2j+1 on lanes 32-63, so a B fragment is the 16 bytes of ONE halo pixel.  The tenth
// tap slot carries the BIAS (pack_stem_frags: bf16 hi + lo parts against the constant (1, 1, 0, ...)).
// A workgroup (4 waves) owns a 16x16 pixel tile: the 18x18 halo is gathered once (CIN fp32 loads per pixel -> 8 bf16
// in LDS, 5 KB), the block's weight fragments (10 KB) are copied to LDS once, wave v takes pixel rows 4v..4v+3 (two
// 32-pixel MFMA tiles) x 64 channels: 5 steps x 4 MFMAs.  Epilogue: optional LeakyReLU, GroupNorm statistics
// (stat_add), bf16 NHWC stores through a per-wave LDS tile (16 B per lane, a tile row's 2 KB contiguous).
// The VALU version of this kernel took 180 us at 288^2 x 16 (v_pk_fma issue bound); this one 60 us + its atomics.
// grid min(units, 4 x CUs) persistent workgroups over (tile, channel block, sample), block 256
// ------------------------------------------------------------------------------------------------
template <int CIN, int ACT>
__global__ __launch_bounds__(256, 4) void stem_mfma_kernel(const float* __restrict__ cond, const float* __restrict__ xt,
                                                           int H, int W, int Hc, int Wc, int C0, int tiles_x,
                                                           const bf16_t* __restrict__ wfrag,
                                                           bf16_t* __restrict__ out, stat_t* __restrict__ stats_out,
                                                           int tiles, int ncb, int nb) {
    __shared__ uint4 halo[324];
    __shared__ uint4 wl[2 * 5 * 64];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) unsigned char otile[4 * 32 * 144];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5;
    // PERSISTENT (round 5): workgroup g walks units g, g + gridDim.x, ... of (sample, channel block, tile), tile fastest; the gather of the
    // NEXT unit's halo (CIN scalar fp32 loads per pixel, NCHW) is in flight under the current unit's MFMAs, epilogue and stores, and the
    // weight fragments are copied when the channel block changes (once per workgroup at C0 = 64)
    const int total = tiles * ncb * nb;
    auto gather = [&](int u, float (&v)[2][8]) {        // both pixels of a thread (324 = 256 + 68) have their loads in flight together
        const int tile = u % tiles, b = u / (tiles * ncb);
        const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int hp = tid + uu * 256;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[uu][k] = 0.f;
            if (hp < 324) {
                const int hr = hp / 18, hc = hp - hr * 18;
                const int yy = y0 + hr - 1, xx = x0 + hc - 1;
                if (yy >= 0 && yy < Hc && xx >= 0 && xx < Wc) {        // zero padding of the (reflect-extended) Hc x Wc image
                    const int ys = reflect_idx(yy, H), xs = reflect_idx(xx, W);
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        const float* src = ci < 3 ? cond : xt;
                        v[uu][ci] = src[(((long long)b * 3 + (ci % 3)) * H + ys) * W + xs];
                    }
                }
            }
        }
    };
    float v[2][8];
    if ((int)blockIdx.x < total) gather(blockIdx.x, v);
    int cb_cur = -1;
#pragma unroll 1
    for (int u = blockIdx.x; u < total; u += gridDim.x) {
        const int tile = u % tiles, cb = (u / tiles) % ncb, b = u / (tiles * ncb);
        const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
        __syncthreads();                                 // the previous unit is done with the halo (and with `red`)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
            if (tid + uu * 256 < 324) halo[tid + uu * 256] = pack8_bf16(v[uu]);
        // this block's weight fragments (10 KB, shared by the four waves) go to LDS; 40 VGPRs of resident fragments would cost half the occupancy
        if (cb != cb_cur) {
            for (int i = tid; i < 2 * 5 * 64; i += 256)
                wl[i] = *reinterpret_cast<const uint4*>(wfrag + ((long long)cb * 2 * 5 * 64 + i) * 8);
            cb_cur = cb;
        }
        if (u + (int)gridDim.x < total) gather(u + gridDim.x, v);
        __syncthreads();
        int hp0[2];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int slot = tp * 32 + (lane & 31);
            hp0[tp] = (wv * 4 + (slot >> 4)) * 18 + (slot & 15);
        }
        f32x16_t acc[2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[tm][tp][e] = 0.f;
#pragma unroll 1
        for (int j = 0; j < 5; ++j) {                       // not unrolled: hipcc would hoist all 20 fragment reads (80 VGPRs)
            const int t0 = 2 * j, t1 = (2 * j + 1 > 8) ? 8 : 2 * j + 1;
            const int sh = hh ? (t1 / 3) * 18 + (t1 % 3) : (t0 / 3) * 18 + (t0 % 3);
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) af[tm] = *reinterpret_cast<const bf16x8_t*>(&wl[(tm * 5 + j) * 64 + lane]);
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                bfr[tp] = *reinterpret_cast<const bf16x8_t*>(&halo[hp0[tp] + sh]);
                if (j == 4 && hh) bfr[tp] = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0u, 0u, 0u));   // bias slot: (1, 1, 0, ...) x (hi, lo)
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
                    acc[tm][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tp], acc[tm][tp], 0, 0, 0);
        }
        float s1 = 0.f, s2 = 0.f;
        unsigned char* ot = otile + wv * (32 * 144);
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {                       // one 32-pixel MFMA tile (two tile rows) per pass
            {
                const int slot = lane & 31;
                const int y = y0 + wv * 4 + tp * 2 + (slot >> 4), x = x0 + (slot & 15);
                const bool inb = y < Hc && x < Wc;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = tm * 32 + 8 * g + 4 * hh;
                        float vv[4] = {acc[tm][tp][4 * g + 0], acc[tm][tp][4 * g + 1], acc[tm][tp][4 * g + 2], acc[tm][tp][4 * g + 3]};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (ACT == 2) vv[i] = fmaxf(0.2f * vv[i], vv[i]);
                            if (inb) { s1 += vv[i]; s2 += vv[i] * vv[i]; }
                        }
                        *reinterpret_cast<uint2*>(ot + slot * 144 + ch * 2) = make_uint2(pack2_bf16(vv[0], vv[1]), pack2_bf16(vv[2], vv[3]));
                    }
            }
            // the tile is private to the wave: program order + the compiler's lgkmcnt waits make its writes visible to it
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane, slot = idx >> 3, c16 = idx & 7;
                const int y = y0 + wv * 4 + tp * 2 + (slot >> 4), x = x0 + (slot & 15);
                if (y < Hc && x < Wc)
                    *reinterpret_cast<uint4*>(out + (((long long)b * (Hc + 2) + y + 1) * (Wc + 2) + x + 1) * C0 + cb * 64 + c16 * 8) =
                        *reinterpret_cast<const uint4*>(ot + slot * 144 + c16 * 16);
            }
        }
        if (stats_out) {                                   // per unit, as the one-shot kernel did: the same fp32 partials -> the same fixed-point sums
            for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
            if (lane == 0) { red[wv * 2] = s1; red[wv * 2 + 1] = s2; }
            __syncthreads();
            if (tid == 0) stat_add(stats_out, b, red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// y = swish(GroupNorm(x)) on the valid region (final_conv.0-1, model/ucdir.py:266-267); the zero
// border of y stays zero so the following 3x3 conv sees the reference's zero padding.
// HBM-bound: 16 B per lane in, 16 B out.  grid (blocks, 1, B)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W, int C,
                                                      const stat_t* __restrict__ stats, double inv_count,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int b = blockIdx.z;
    float mean, rstd;
    {
        double S, Q;
        stat_read(stats, nullptr, b, S, Q);
        double m = S * inv_count;
        double var = Q * inv_count - m * m;
        if (var < 0) var = 0;
        mean = (float)m; rstd = (float)(1.0 / sqrt(var + 1e-5));
    }
    const int c8n = C / 8;
    const long long bs = (long long)(H + 2) * (W + 2) * C;
    const long long n = (long long)H * W * c8n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8;
        const long long pix = i / c8n;
        const int yy = (int)(pix / W), xx = (int)(pix % W);
        const long long off = b * bs + ((long long)(yy + 1) * (W + 2) + xx + 1) * C + c;
        const uint4 v = *reinterpret_cast<const uint4*>(x + off);
        const bf16_t* h = reinterpret_cast<const bf16_t*>(&v);
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float fv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) fv[k] = silu_fast((bf2f(h[k]) - mean) * rstd * gg[k] + bb[k]);
        *reinterpret_cast<uint4*>(y + off) = pack8_bf16(fv);
    }
}

// ------------------------------------------------------------------------------------------------
// final_conv fused (model/ucdir.py:266-268): out = conv3x3(swish(GroupNorm(x))), C -> cout <= 4 channels, written
// as the cropped fp32 NCHW result.  With 3 output channels the problem is HBM-bound (read x once, 170 MB at the
// 288^2 level; 2.3 GFLOP): no activated copy of x is written and re-read, no 64-row matrix-core tile for 3 rows.
// A workgroup (4 waves) owns a 16x16 pixel tile: the 18x18 halo is normalised + activated ONCE per element while
// it is staged into LDS (bf16, 32 channels per pass, rows padded by 16 B), then wave v takes pixel rows 4v..4v+3 as four 16-pixel
// column tiles of v_mfma_f32_16x16x32_bf16 (A = weights, rows >= cout zero; K step = one tap x 32 channels).
// The weight fragments arrive pre-laid-out ([step][lane][8] bf16, pack_final_frags) and are copied to LDS once.
// Pixels outside the image contribute 0 (the conv zero-pads the ACTIVATED tensor).
// grid min(tiles, 3 x CUs) persistent workgroups, block 256, dynamic LDS 324 * 80 + 9 * (C/32) * 1024 + 8C + 8B bytes; C % 32 == 0.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ __launch_bounds__(256) void final_conv_kernel(const bf16_t* __restrict__ x, int H, int W, int C,
                                                         const stat_t* __restrict__ stats, double inv_count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const bf16_t* __restrict__ wfrag, const float* __restrict__ bias, int cout,
                                                         float* __restrict__ out, int crop_h, int crop_w, int nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    constexpr int RS = 80;                             // LDS row: 32 channels of one halo pixel + 16 B pad
    unsigned char* wl = fsm + 324 * RS;                // weight fragments, all steps
    const int nc32 = C / 32, nstep = 9 * nc32;
    // PERSISTENT (round 5): a workgroup walks tiles blockIdx.x, + gridDim.x, ... (three resident workgroups per CU); the weight fragments,
    // the per-sample (mean, rstd) list and - per sample - the scale / shift table are set up once instead of once per 16 x 16 tile
    // (5,184 workgroups per B = 16 launch each paid a serial statistics read, an 18 KB copy and a table build for 2.3 us of work)
    for (int it = threadIdx.x; it < nstep * 64; it += 256)
        *reinterpret_cast<uint4*>(wl + it * 16) = *reinterpret_cast<const uint4*>(wfrag + (long long)it * 8);
    float* gb = reinterpret_cast<float*>(wl + nstep * 1024);          // GN as one FMA: scale = rstd*gamma | shift = beta - mean*rstd*gamma
    float* ms = gb + 2 * C;                                            // [nb][2]
    for (int i = threadIdx.x; i < nb; i += 256) {
        double S, Q;
        float mean, rstd;
        stat_read(stats, nullptr, i, S, Q);
        mean_rstd(S, Q, inv_count, mean, rstd);
        ms[2 * i] = mean; ms[2 * i + 1] = rstd;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, kg = lane >> 4;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4, tps = tiles_x * tiles_y, total = nb * tps;
    int b_cur = -1;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int b = t / tps, r = t - b * tps, ty = r / tiles_x, tx = r - ty * tiles_x;
        const int x0 = tx * 16, y0 = ty * 16;
        const bf16_t* xb = x + (long long)b * (H + 2) * (W + 2) * C;
        // 32 channels per pass keep the halo at 26 KB (three workgroups per CU instead of two at C = 64).
        // halo staging: 324 pixels x 4 sixteen-byte items = 1296 items, 6 per thread; the items of pass c + 1 are requested in front of the
        // MFMAs of pass c (their registers are free once pass c is in LDS)
        uint4 v[6]; int dst[6]; long long src[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = threadIdx.x + u * 256;
            dst[u] = -1; src[u] = 0;
            if (it < 324 * 4) {
                const int hp = it >> 2, c = (it & 3) * 8;
                const int hr = hp / 18, hc = hp - hr * 18;
                const int gy = y0 + hr - 1, gx = x0 + hc - 1;          // image coordinates of this halo pixel
                dst[u] = (hp * RS + c * 2) | ((gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0 : (1 << 30));
                src[u] = ((long long)(gy + 1) * (W + 2) + gx + 1) * C + c;
            }
        }
        auto request = [&](int c32) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                v[u] = make_uint4(0, 0, 0, 0);
                if (dst[u] >= 0 && !(dst[u] >> 30)) v[u] = *reinterpret_cast<const uint4*>(xb + src[u] + c32 * 32);
            }
        };
        request(0);
        if (b != b_cur) {                              // (every thread is behind the previous tile's last activation pass: nobody reads gb now)
            const float mean = ms[2 * b], rstd = ms[2 * b + 1];
            for (int i = threadIdx.x; i < C; i += 256) { const float sc = rstd * gamma[i]; gb[i] = sc; gb[C + i] = beta[i] - mean * sc; }
            b_cur = b;
        }
        f32x4_t acc[4];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) acc[tq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int c32 = 0; c32 < nc32; ++c32) {
            __syncthreads();                           // gb ready / previous pass (or tile) done with the halo
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                if (dst[u] < 0) continue;
                uint4 o = make_uint4(0, 0, 0, 0);
                const int off = dst[u] & ((1 << 30) - 1);
                if (!(dst[u] >> 30)) {
                    const int c = c32 * 32 + ((threadIdx.x + u * 256) & 3) * 8;
                    const bf16_t* h = reinterpret_cast<const bf16_t*>(&v[u]);
                    float fv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) fv[k] = silu_fast(fmaf(bf2f(h[k]), gb[c + k], gb[C + c + k]));
                    o = pack8_bf16(fv);
                }
                *reinterpret_cast<uint4*>(fsm + off) = o;
            }
            if (c32 + 1 < nc32) request(c32 + 1);
            __syncthreads();
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(wl + ((tap * nc32 + c32) * 64 + lane) * 16);
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const int hp = (wv * 4 + tq + ky) * 18 + col + kx;
                    const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(fsm + hp * RS + kg * 16);
                    acc[tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[tq], 0, 0, 0);
                }
            }
        }
        if (kg == 0) {                                 // lanes 0..15 hold output rows (channels) 0..3 of their pixel column
            const int xx = x0 + col;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const int y = y0 + wv * 4 + tq;
                if (y < crop_h && xx < crop_w) {
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < cout) out[(((long long)b * cout + o) * crop_h + y) * crop_w + xx] = acc[tq][o] + bias[o];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// noise-level embedding + every block's time weights (model/ucdir.py:24-29,212-214,106,125)
// one block per sample.  tw layout per block layer: [8][inner] W0, [8] b0, [8][8] W2, [8] b2
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_mlp_kernel(const float* __restrict__ level, int inner,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ tw, int nblocks,
                                                       float* __restrict__ attw /*[nblocks][B][8]*/, int B) {
    extern __shared__ float tsm[];
    float* enc = tsm;                 // [inner]
    float* h1 = tsm + inner;          // [4*inner]
    float* temb = h1 + 4 * inner;     // [inner]
    float* hid = temb + inner;        // [nblocks][8]
    const int b = blockIdx.x;
    const float lv = level[b];
    const int half = inner / 2;
    for (int k = threadIdx.x; k < half; k += 256) {
        const float step = (float)k / (float)half;
        const float e = lv * expf(-9.210340371976184f * step);
        enc[k] = sinf(e); enc[k + half] = cosf(e);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * inner; o += 256) {
        float a = b1[o];
        for (int k = 0; k < inner; ++k) a += w1[o * inner + k] * enc[k];
        h1[o] = a / (1.0f + expf(-a));
    }
    __syncthreads();
    for (int o = threadIdx.x; o < inner; o += 256) {
        float a = b2[o];
        for (int k = 0; k < 4 * inner; ++k) a += w2[o * 4 * inner + k] * h1[k];
        temb[o] = a;
    }
    __syncthreads();
    const int per = 8 * inner + 8 + 64 + 8;
    for (int i = threadIdx.x; i < nblocks * 8; i += 256) {
        const int l = i / 8, o = i % 8;
        const float* W0 = tw + (long long)l * per;
        float a = W0[8 * inner + o];
        for (int k = 0; k < inner; ++k) a += W0[o * inner + k] * temb[k];
        hid[i] = a / (1.0f + expf(-a));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nblocks * 8; i += 256) {
        const int l = i / 8, o = i % 8;
        const float* W2 = tw + (long long)l * per + 8 * inner + 8;
        float a = W2[64 + o];
        for (int k = 0; k < 8; ++k) a += W2[o * 8 + k] * hid[l * 8 + k];
        attw[((long long)l * B + b) * 8 + o] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// guide branch of one block (model/ucdir.py:133-135 without attw): bilinear 1/k resample of the
// (reflect padded) guide = mean of the centre 2x2 of each k x k cell, conv1x1 3->16, SimpleGate,
// conv3x3 8->8.  Independent of t and x_t -> evaluated once per image.  Output compact
// [B][Hl*Wl][8] fp32.  gw: [16][3] W0, [16] b0, [8][8][3][3] W2, [8] b2
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void guide_branch_kernel(const float* __restrict__ guide, int H, int W, int Hc, int Wc,
                                                           int k, const float* __restrict__ gw, float* __restrict__ G) {
    __shared__ float sw[16 * 3 + 16 + 576 + 8];
    for (int i = threadIdx.x; i < 16 * 3 + 16 + 576 + 8; i += 256) sw[i] = gw[i];
    __syncthreads();
    const float* W0 = sw; const float* b0 = sw + 48; const float* W2 = sw + 64; const float* b2 = sw + 64 + 576;
    const int Hl = Hc / k, Wl = Wc / k;
    const int b = blockIdx.z;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hl * Wl) return;
    const int y = pix / Wl, x = pix % Wl;
    float out[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) out[o] = b2[o];
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= Hl) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if (xx < 0 || xx >= Wl) continue;
            float g3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* gp = guide + ((long long)b * 3 + c) * H * W;
                if (k == 1) {
                    g3[c] = gp[(long long)reflect_idx(yy, H) * W + reflect_idx(xx, W)];
                } else {
                    const int y0 = k * yy + k / 2 - 1, x0 = k * xx + k / 2 - 1;
                    const int ya = reflect_idx(y0, H), yb = reflect_idx(y0 + 1, H);
                    const int xa = reflect_idx(x0, W), xb = reflect_idx(x0 + 1, W);
                    // bilinear with lambda = 0.5 in both directions (ATen order: rows then columns)
                    const float top = 0.5f * gp[(long long)ya * W + xa] + 0.5f * gp[(long long)ya * W + xb];
                    const float bot = 0.5f * gp[(long long)yb * W + xa] + 0.5f * gp[(long long)yb * W + xb];
                    g3[c] = 0.5f * top + 0.5f * bot;
                }
            }
            float v[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) v[o] = b0[o] + W0[o * 3] * g3[0] + W0[o * 3 + 1] * g3[1] + W0[o * 3 + 2] * g3[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = v[i] * v[i + 8];
#pragma unroll
                for (int o = 0; o < 8; ++o) out[o] += W2[((o * 8 + i) * 3 + ky) * 3 + kx] * a;
            }
        }
    }
    float* gp = G + ((long long)b * Hl * Wl + pix) * 8;
    *reinterpret_cast<float4*>(gp) = make_float4(out[0], out[1], out[2], out[3]);
    *reinterpret_cast<float4*>(gp + 4) = make_float4(out[4], out[5], out[6], out[7]);
}

// ------------------------------------------------------------------------------------------------
// row softmax: S fp32 [B][N][Npad] -> P bf16 [B][N][Npad] (columns >= N written as 0).
// One wave per row.  Rows up to 64*4*NV floats are read ONCE with 16-byte loads and kept in
// registers (max, exp, sum, normalise, 8-byte bf16 stores); longer rows (1024^2 patch windows,
// N = 16384) stream three times from L2.
// ------------------------------------------------------------------------------------------------
template <int NV>   // float4 vectors per lane held in registers
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ S, bf16_t* __restrict__ P,
                                                      long long rows, int N, int Npad) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* s = S + row * Npad;
    bf16_t* pr = P + row * Npad;
    if (NV > 0) {
        float4 v[NV > 0 ? NV : 1];
        float m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = (i * 64 + lane) * 4;
            if (j < Npad) v[i] = *reinterpret_cast<const float4*>(s + j); else v[i] = make_float4(0, 0, 0, 0);
            if (j + 0 < N) m = fmaxf(m, v[i].x);
            if (j + 1 < N) m = fmaxf(m, v[i].y);
            if (j + 2 < N) m = fmaxf(m, v[i].z);
            if (j + 3 < N) m = fmaxf(m, v[i].w);
        }
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = (i * 64 + lane) * 4;
            v[i].x = j + 0 < N ? expf(v[i].x - m) : 0.f;
            v[i].y = j + 1 < N ? expf(v[i].y - m) : 0.f;
            v[i].z = j + 2 < N ? expf(v[i].z - m) : 0.f;
            v[i].w = j + 3 < N ? expf(v[i].w - m) : 0.f;
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = (i * 64 + lane) * 4;
            if (j < Npad)
                *reinterpret_cast<uint2*>(pr + j) = make_uint2(pack2_bf16(v[i].x * inv, v[i].y * inv), pack2_bf16(v[i].z * inv, v[i].w * inv));
        }
    } else {
        float m = -3.0e38f;
        for (int j = lane; j < N; j += 64) m = fmaxf(m, s[j]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float sum = 0.f;
        for (int j = lane; j < N; j += 64) sum += expf(s[j] - m);
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const float inv = 1.0f / sum;
        for (int j = lane; j < Npad; j += 64) pr[j] = j < N ? f2bf(expf(s[j] - m) * inv) : (bf16_t)0;
    }
}

// V^T: QKV compact [B][N][ld] (v at channel offset voff) -> [B][C][Npad] bf16
__global__ void transpose_v_kernel(const bf16_t* __restrict__ qkv, int N, int ld, int voff, int C, int Npad,
                                   bf16_t* __restrict__ vt) {
    __shared__ bf16_t tile[32][33];
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r;
        tile[r][tx] = n < N ? qkv[((long long)b * N + n) * ld + voff + c0 + tx] : (bf16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + tx;
        if (n < Npad) vt[((long long)b * C + c0 + r) * Npad + n] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// ancestral sampler update (model/diffusion.py:150-158,171-183), in place
// ------------------------------------------------------------------------------------------------
__global__ void sampler_step_kernel(float* __restrict__ xt, const float* __restrict__ eps, const float* __restrict__ noise,
                                    long long n, float c_recip, float c_recipm1, float coef1, float coef2, float sigma) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = xt[i];
        float x0 = c_recip * x - c_recipm1 * eps[i];
        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float r = coef1 * x0 + coef2 * x;
        if (noise) r += noise[i] * sigma;
        xt[i] = r;
    }
}

// Counter-based standard-normal noise for the sampler (SURVEY.md section 7 "RNG"; replaces torch.randn + clone per step on the
// throughput path - parity tests inject recorded noise instead): Philox4x32-10 (Salmon et al. 2011) keyed by the 64-bit seed,
// counter = (group of four elements, step) -> four uniforms -> two Box-Muller pairs.  The value of element i at step k depends on
// (seed, k, i) only: identical on every rank of a sharded restoration, independent of grid size, batch split and launch order.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void normal4(unsigned long long seed, uint32_t step, unsigned long long group, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)group, (uint32_t)(group >> 32), step, 0x55434449u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // (0, 1), never 0 or 1: 23 bits + 0.5 is exactly representable in fp32 (24 bits + 0.5 is not above 2^23: it rounded to
        // even, u could be exactly 1 and the upper half of the grid was coarsened - round-3 advice)
        const float u1 = ((float)(r[2 * h] >> 9) + 0.5f) * (1.0f / 8388608.0f);
        const float u2 = ((float)(r[2 * h + 1] >> 9) + 0.5f) * (1.0f / 8388608.0f);
        const float rad = sqrtf(-2.0f * __logf(u1));
        z[2 * h] = rad * __builtin_amdgcn_cosf(u2);                                      // v_cos_f32 / v_sin_f32 take revolutions
        z[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
}
// x <- N(0, 1) (x_T of a restoration); n elements, four per thread and counter
// seeds != nullptr (round 6, ucdir_*_batched): the buffer is B samples of `per` elements (a multiple of 4), sample b draws the stream of
// seeds[b] with counters that restart at its first element - the noise of an image does not depend on which batch it is restored in
__device__ __forceinline__ void sample_stream(const unsigned long long* __restrict__ seeds, long long per4, long long g, unsigned long long& seed, unsigned long long& grp) {
    if (seeds) { const long long b = g / per4; seed = seeds[b]; grp = (unsigned long long)(g - b * per4); }
    else grp = (unsigned long long)g;
}
__global__ void fill_normal_kernel(float* __restrict__ x, long long n, unsigned long long seed0, uint32_t step, const unsigned long long* __restrict__ seeds, long long per4) {
    const long long ng = (n + 3) >> 2;
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < ng; g += (long long)gridDim.x * blockDim.x) {
        float z[4];
        unsigned long long seed = seed0, grp;
        sample_stream(seeds, per4, g, seed, grp);
        normal4(seed, step, grp, z);
        if (4 * g + 3 < n) *reinterpret_cast<float4*>(x + 4 * g) = make_float4(z[0], z[1], z[2], z[3]);
        else for (int e = 0; e < 4; ++e) if (4 * g + e < n) x[4 * g + e] = z[e];
    }
}
// the sampler update with its noise generated in registers (no noise tensor, no randn / clone launches)
__global__ void sampler_step_rng_kernel(float* __restrict__ xt, const float* __restrict__ eps, long long n, float c_recip, float c_recipm1,
                                        float coef1, float coef2, float sigma, unsigned long long seed0, uint32_t step,
                                        const unsigned long long* __restrict__ seeds, long long per4) {
    const long long ng = (n + 3) >> 2;
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < ng; g += (long long)gridDim.x * blockDim.x) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (sigma != 0.f) {
            unsigned long long seed = seed0, grp;
            sample_stream(seeds, per4, g, seed, grp);
            normal4(seed, step, grp, z);
        }
        if (4 * g + 3 < n) {
            const float4 x4 = *reinterpret_cast<const float4*>(xt + 4 * g), e4 = *reinterpret_cast<const float4*>(eps + 4 * g);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, es[4] = {e4.x, e4.y, e4.z, e4.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = fminf(fmaxf(c_recip * xs[e] - c_recipm1 * es[e], -1.0f), 1.0f);
                o[e] = coef1 * x0 + coef2 * xs[e] + z[e] * sigma;
            }
            *reinterpret_cast<float4*>(xt + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < n) {
                    const float x = xt[4 * g + e];
                    const float x0 = fminf(fmaxf(c_recip * x - c_recipm1 * eps[4 * g + e], -1.0f), 1.0f);
                    xt[4 * g + e] = coef1 * x0 + coef2 * x + z[e] * sigma;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2x2 max pooling (predictor, model/ucdir.py:317,363): zero-bordered NHWC bf16 in and out.
// ------------------------------------------------------------------------------------------------
__global__ void maxpool2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c8n = C / 8;
    const long long n = (long long)B * Ho * Wo * c8n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8; long long t = i / c8n;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho); const int b = (int)(t / Ho);
        const bf16_t* base = x + (((long long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1) * C + c;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -3.0e38f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const uint4 v = *reinterpret_cast<const uint4*>(base + ((long long)dy * (W + 2) + dx) * C);
                const bf16_t* h = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], bf2f(h[k]));
            }
        uint4 ov; bf16_t* oh = reinterpret_cast<bf16_t*>(&ov);
#pragma unroll
        for (int k = 0; k < 8; ++k) oh[k] = f2bf(m[k]);
        *reinterpret_cast<uint4*>(y + (((long long)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * C + c) = ov;
    }
}

// ------------------------------------------------------------------------------------------------
// Matrix-core rate the device sustains (ucdir_matrix_rate; tools/mfma_peak.hip is the stand-alone version): MFMAs only, the
// 4 x 2 fragment pattern of conv_sk_kernel's wave tile, operands resident in registers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void matrix_rate_kernel(int iters, int random, float* out) {
    f32x16_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8_t a[4], b[2];
    unsigned r = (threadIdx.x + 977u * blockIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            r = r * 1664525u + 1013904223u;
            const float u = ((r >> 8) & 0xffff) / 65536.f + ((r >> 20) & 0xfff) / 4096.f - 1.f;
            a[k][e] = (__bf16)(random ? u * 0.03f : (float)(threadIdx.x & 3));
            if (k < 2) b[k][e] = (__bf16)(random ? u * 1.7f : (float)(threadIdx.x & 1));
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;                   // (keeps the loop alive)
}

// ------------------------------------------------------------------------------------------------
// Inter-step patch split (utils/util.py:108-146): the reference reflect-pads the canvas and slices one window after the other; here
// one launch writes the whole window batch out[(w * B + b)][c][y][x] = x[b][c][refl(h0_w + y - pd)][refl(w0_w + x - pd)] straight from
// the un-padded canvas (no padded copy, no per-window cat).  win = [nwin][2] (h0, w0) in padded coordinates; reflect without the
// edge pixel (torch 'reflect').  grid (ceil(skip / 256) * skip, C, nwin * B), 256 threads: one block = 256 columns of one window row.  The host validates the window
// list before it is uploaded (ucdir_amd/patch.py); the reflected index is clamped into the canvas all the same, so a bad entry can never read out of bounds.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_windows_kernel(const float* __restrict__ x, int B, int C, int H, int W, int pd, const int* __restrict__ win,
                                                             int skip, float* __restrict__ out) {
    const int wb = blockIdx.z, w = wb / B, b = wb - w * B, c = blockIdx.y;
    const int xblocks = (skip + 255) / 256;
    const int y = blockIdx.x / xblocks, xx = (blockIdx.x - y * xblocks) * 256 + threadIdx.x;
    if (xx >= skip) return;
    int sy = win[2 * w] + y - pd, sx = win[2 * w + 1] + xx - pd;
    sy = sy < 0 ? -sy : (sy >= H ? 2 * H - 2 - sy : sy);
    sx = sx < 0 ? -sx : (sx >= W ? 2 * W - 2 - sx : sx);
    sy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
    sx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
    out[(((long long)wb * C + c) * skip + y) * skip + xx] = x[(((long long)b * C + c) * H + sy) * W + sx];
}
