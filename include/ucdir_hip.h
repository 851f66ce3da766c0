/*
 * ucdir_hip.h — C ABI of libucdir_hip.so, the MI355X (gfx950) denoiser engine.
 *
 * Drop-in boundary: this library replaces the *inner operator* that the reference's
 * sampler calls once per DDPM step,
 *
 *     denoise_fn(cat[cond, x_t] (B,6,H,W), noise_level (B,1), guide=(B,3,H,W)) -> eps (B,3,H,W)
 *         reference: model/diffusion.py:166 (call site), model/ucdir.py:295-307 (DY3h.forward),
 *                    model/ucdir.py:270-293 (DY3h.naiveforward)
 *
 * plus the point-wise ancestral update around it (model/diffusion.py:150-158,171-183).
 * The reference has no FFI of its own (it is pure Python/PyTorch), so the entry points below
 * are what a ctypes binding added to the reference's `model/networks.py:define_G` would
 * bind; INTEGRATION.md shows that stub.
 *
 * Conventions
 *   - plain C: opaque handle, raw device pointers, sizes; no C++/torch types cross the ABI;
 *   - every function returns 0 on success, non-zero on error; ucdir_last_error() describes
 *     the last failure on the calling thread;
 *   - all tensor arguments are DEVICE pointers unless the name ends in `_host`;
 *   - tensors crossing the ABI use the reference's layout: NCHW, fp32, contiguous;
 *   - all work is enqueued on the hipStream_t passed in (as void*) and is asynchronous;
 *   - a handle is not thread-safe (one handle per device / per process, like the reference's
 *     one-process-per-GPU use); the library owns packed weights + workspace, the caller owns
 *     every tensor it passes in;
 *   - every entry point that takes a handle runs on the handle's device and restores the calling
 *     thread's current HIP device before it returns; ucdir_sampler_step runs on the device that
 *     owns x_t; the single-operator test entry points (ucdir_op_*) use the current device;
 *   - the handles of one device share a 64 MiB split-K scratch: enqueue their work on ONE stream (as the sampler does:
 *     predictor, then 50 x denoiser), or order the streams yourself.
 */
#ifndef UCDIR_HIP_H
#define UCDIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: ucdir_gather_windows and ucdir_matrix_rate joined the interface (round 4); a binding built against version 3 must not load this library
 * silently (round-4 verdict).  Round 5 added no symbol: new kernels are dispatch changes behind the same entry points (A/B switches: the
 * environment variables of DESIGN.md and ucdir_debug_flag names "flash2", "persist_grid", ...). */
#define UCDIR_ABI_VERSION 5
#define UCDIR_MAX_MULTS 8

typedef struct ucdir_ctx ucdir_ctx;

/* Mirrors the `model.unet` section of config/sid.yaml:41-56 (reference: DY3h.__init__,
 * model/ucdir.py:205-207). */
typedef struct ucdir_config {
    int32_t in_channel;                     /* 6  = cat[cond(3), x_t(3)]            */
    int32_t out_channel;                    /* 3                                     */
    int32_t inner_channel;                  /* 64 (must be a multiple of 64)         */
    int32_t n_mults;
    int32_t channel_mults[UCDIR_MAX_MULTS]; /* 1,2,4,8,8                             */
    int32_t n_attn_res;
    int32_t attn_res[UCDIR_MAX_MULTS];      /* 16                                    */
    int32_t res_blocks;                     /* 2                                     */
    int32_t image_size;                     /* 128 (only used to place attention)    */
    int32_t device;                         /* HIP device ordinal                    */
    int32_t attn_fp16;                      /* 0: bf16 attention operands (default); 1: IEEE-half q, k, v', P on
                                             * v_mfma_*_f16 (the JPEG configuration's "fp16 attention MFMA path",
                                             * BASELINE.json configs[4]); accumulation, softmax and output stay fp32/bf16 */
} ucdir_config;

int32_t     ucdir_abi_version(void);
const char* ucdir_last_error(void);

/* ---- lifetime ------------------------------------------------------------------------- */
int32_t ucdir_create(const ucdir_config* cfg, ucdir_ctx** out);
void    ucdir_destroy(ucdir_ctx* ctx);

/* ---- weights (replaces nn.Module.load_state_dict for denoise_fn.*, model/model.py:224-251) --
 * `name` is the reference state_dict key without the "denoise_fn." prefix
 * (e.g. "downs.4.res_block.conv1.weight"); `data_host` is fp32, reference shape. After all
 * tensors are supplied, ucdir_finalize_weights folds GroupNorm affines into the following
 * convolution, converts to bf16 MFMA layouts and uploads. */
int32_t ucdir_load_weight(ucdir_ctx* ctx, const char* name, const float* data_host,
                          const int64_t* shape, int32_t ndim);
int32_t ucdir_finalize_weights(ucdir_ctx* ctx);
/* number of parameter tensors the config expects / the i-th expected name */
int32_t     ucdir_num_weights(const ucdir_ctx* ctx);
const char* ucdir_weight_name(const ucdir_ctx* ctx, int32_t i);

/* ---- per-image state: the guide branch of every block (model/ucdir.py:133-135) does not
 * depend on t or x_t, so it is evaluated once per image.  guide: (B,3,H,W) fp32.
 * `pad_mode` 1 = DY3h.forward semantics (reflect-pad bottom/right to (d/32+1)*32, crop on
 * output; model/ucdir.py:303-307); 0 = naiveforward (H, W multiples of 32; used for the windows
 * of the inter-step patch split, utils/util.py:108-146). Allocates workspace on first use /
 * on shape change. */
int32_t ucdir_prepare_guide(ucdir_ctx* ctx, const float* guide, int32_t B, int32_t H, int32_t W,
                            int32_t pad_mode, void* stream);

/* ---- the denoiser: eps = DY3h(cat[cond, x_t], noise_level, guide) ----------------------
 * cond, x_t: (B,3,H,W) fp32 (the channel concat of model/diffusion.py:166 is done on the fly);
 * noise_level: (B) fp32; eps: (B,3,H,W) fp32.  (B,H,W) must equal the shape of the last
 * ucdir_prepare_guide: a mismatch is an error, never an out-of-bounds access. */
int32_t ucdir_unet_forward(ucdir_ctx* ctx, const float* cond, const float* x_t,
                           const float* noise_level, float* eps,
                           int32_t B, int32_t H, int32_t W, void* stream);
/* B = 1 / `-p val` latency path: with on != 0 every ucdir_unet_forward is replayed from a HIP graph
 * captured once per (cond, x_t, noise_level, eps) pointer set (keep them in persistent buffers, as
 * ucdir_amd.diffusion.p_sample_loop does); ~150 launches become one hipGraphLaunch.  Graphs are
 * dropped when weights are re-finalised or the planned shape changes. */
int32_t ucdir_set_graph(ucdir_ctx* ctx, int32_t on);

/* ---- ancestral sampler update (model/diffusion.py:150-158,171-183), in place on x_t:
 *   x0   = clamp(c_recip * x_t - c_recipm1 * eps, -1, 1)
 *   x_t <- coef1 * x0 + coef2 * x_t + sigma * noise      (noise may be NULL when sigma == 0)
 * n = number of fp32 elements. */
int32_t ucdir_sampler_step(float* x_t, const float* eps, const float* noise, int64_t n,
                           float c_recip, float c_recipm1, float coef1, float coef2, float sigma,
                           void* stream);

/* The same update with its noise generated in registers (ABI 3): element i at step `step` gets the standard normal
 * Philox4x32-10(key = seed, counter = (i / 4, step)) -> Box-Muller value number i % 4 - a function of (seed, step, i) only, so
 * every rank of a sharded restoration draws the same noise without a generator or a broadcast, and no noise tensor exists.
 * ucdir_fill_normal writes the same stream into a buffer (x_T = step 0).  Pointers 16-byte aligned. */
int32_t ucdir_sampler_step_rng(float* x_t, const float* eps, int64_t n,
                               float c_recip, float c_recipm1, float coef1, float coef2, float sigma,
                               uint64_t seed, uint32_t step, void* stream);
int32_t ucdir_fill_normal(float* x, int64_t n, uint64_t seed, uint32_t step, void* stream);
/* Per-sample streams (ABI 5; sr.py -p val restores same-sized images as one batch, reference sr.py:518-561 runs them one by one through
 * data/__init__.py:47 batch_size 1): the buffer is n / per samples of `per` fp32 elements (per a multiple of 4); sample b draws
 * Philox4x32-10(key = seeds_dev[b], counter = (local element / 4, step)) - exactly what ucdir_sampler_step_rng / ucdir_fill_normal
 * draw for a buffer that holds this sample alone with seed = seeds_dev[b].  An image's noise therefore does not depend on the batch it
 * is grouped into.  seeds_dev: n / per uint64 values ON THE DEVICE of the buffer. */
int32_t ucdir_sampler_step_rng_batched(float* x_t, const float* eps, int64_t n, int64_t per,
                                       float c_recip, float c_recipm1, float coef1, float coef2, float sigma,
                                       const uint64_t* seeds_dev, uint32_t step, void* stream);
int32_t ucdir_fill_normal_batched(float* x, int64_t n, int64_t per, const uint64_t* seeds_dev, uint32_t step, void* stream);
/* Window batch of the inter-step patch split (utils/util.py:113-137: F.pad(..., mode='reflect') then one slice per window) in ONE launch,
 * straight from the un-padded canvas: out[(w * B + b)][c][y][x] = x[b][c][refl(h0_w + y - pad)][refl(w0_w + x - pad)], x (B, C, H, W) fp32,
 * out (nwin * B, C, skip, skip) fp32, win_dev = nwin pairs (h0, w0) of int32 ON THE DEVICE in padded coordinates (the window list of
 * utils/util.py:119-137).  Windows must lie inside the padded canvas (H + 2 pad) x (W + 2 pad); pad < H, W. */
int32_t ucdir_gather_windows(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, int32_t pad, const int32_t* win_dev,
                             int32_t nwin, int32_t skip, float* out, void* stream);

/* ---- introspection (tests / profiling) ---------------------------------------------------
 * Copy the activation a layer produced in the last forward into dst as (B,C,Hc,Wc) fp32 NCHW
 * (Hc, Wc = compute size).  layer = state_dict prefix ("downs.0", "ups.7", "mid.0", ...),
 * what = "out" (layer output), "h1" (swish(conv1) inside a block), or "attw" (a block's eight time weights,
 * noise_func(noise_level_mlp(PosEnc(level))), (B,8) fp32: model/ucdir.py:24-29,106,125,212-214). */
int32_t ucdir_debug_read(ucdir_ctx* ctx, const char* layer, const char* what, float* dst,
                         int64_t dst_elems, void* stream);
int64_t ucdir_workspace_bytes(const ucdir_ctx* ctx);
/* Process-wide test switches, read when a shape is planned / an op entry point runs.
 * "flash": 1 = flash-attention kernel, 0 = materialised-score path (QK^T, softmax, PV as three launches),
 * -1 = environment default (UCDIR_NO_FLASH).  "splitk": 1 / 0 / -1 the same for split-K and unit splits of
 * under-filled grids (UCDIR_SPLITK).  "persist_grid": n > 0 launches the persistent kernels (akgm_ws) with n workgroups
 * instead of one per compute unit, 0 restores the default.  "wsb": 1 routes the AKGM tails of 8 / 16 channels per group
 * through akgm_ws32_kernel<8 | 16> instead of akgm_ws_kernel (A/B and tests), 0 / -1 (UCDIR_WSB) as above.  "convsk": 0 = 3x3 convs
 * on conv3x3_halo (the round-3 dispatch), 1 = conv_sk_kernel's persistent 8-wave stream-K kind forced, 2 = its one-shot 4-wave kind
 * forced (both regardless of the size thresholds: tests), -1 = environment default (UCDIR_NO_CONV_SK, UCDIR_CONV_SK_MODE).  Unknown
 * names are an error. */
int32_t ucdir_debug_flag(const char* name, int32_t value);
/* Host-side launch planning, callable without a device (tests): what = "ksplit" -> the K-split factor conv3x3_halo would use
 * for a grid of `wgs` workgroups over `nchunks` 32-channel chunks of `steps_per_chunk` K steps producing `out_elems` outputs;
 * what = "usplit" -> the unit split (1 | 2 | 4) of the 64-per-group AKGM kernel for `wgs` workgroups.  -1 on a bad name. */
int32_t ucdir_debug_launch_plan(const char* what, int32_t wgs, int32_t nchunks, int32_t steps_per_chunk, double out_elems);
/* Per-launch HIP-event timing of the GEMM-core kernels (bench.py's roofline leg).  While enabled,
 * every launch is bracketed by events on its stream; ucdir_profile_read synchronises the stream and
 * aggregates per kernel instantiation: key = 100*[TM==128] + 10*[AKGM epilogue] + column mode
 * (0 stride-1, 1 down, 2 up, 3 plain GEMM, 4 compact-in), launches, total ms, algorithmic FLOPs
 * (2*MAC of the un-padded problem) and algorithmic bytes (operands + output once). */
int32_t ucdir_profile_enable(int32_t on);
int32_t ucdir_profile_read(int32_t cap, int32_t* keys, int32_t* launches, double* ms, double* flops,
                           double* bytes, int32_t* nrows, void* stream);
/* algorithmic FLOPs of one forward at the prepared shape (2*MAC, reference op count) */
double  ucdir_forward_flops(const ucdir_ctx* ctx);
/* Matrix-core rate this device sustains (bench.py's `roofline.sustained_peak`; nothing on the reference side corresponds to it):
 * a kernel of v_mfma_f32_32x32x16_bf16 only - eight accumulator tiles per wave fed from four A and two B fragments held in
 * registers, two waves per SIMD on every CU, no memory traffic - run for `iters` x 8 MFMAs per wave; best of three launches in
 * *tflops.  random = 1: operands drawn like the conv kernels' (weights ~0.03 sigma, activations ~1 sigma); 0: small integers
 * (the clock, hence the rate, depends on how many operand bits toggle: 1.75-1.8 against 2.45 PFLOP/s on the boxes measured). */
int32_t ucdir_matrix_rate(int32_t iters, int32_t random, double* tflops, void* stream);

/* ---- UNetSeeInDark predictor (model/ucdir.py:310-416): initial restoration = guide = residual base.
 * `name` = reference state_dict key without the "predictor." prefix ("conv1_1.weight", "upv6.bias", ...).
 * forward: x (B,3,H,W) fp32 -> y (B,3,H,W) fp32, same +32 reflect pad / crop as the reference. */
typedef struct ucdir_predictor ucdir_predictor;
int32_t ucdir_predictor_create(int32_t device, ucdir_predictor** out);
void    ucdir_predictor_destroy(ucdir_predictor* p);
int32_t ucdir_predictor_load_weight(ucdir_predictor* p, const char* name, const float* data_host,
                                    const int64_t* shape, int32_t ndim);
int32_t ucdir_predictor_finalize(ucdir_predictor* p);
int32_t ucdir_predictor_forward(ucdir_predictor* p, const float* x, float* y, int32_t B, int32_t H, int32_t W,
                                void* stream);

/* ---- single-operator entry points (unit parity tests; fp32 NCHW in/out, bf16 inside) -----
 * conv: y = act(conv(GN?(cat[x0,x1]))) with 3x3 (mode 0 stride 1, 1 stride-2 down,
 * 2 nearest-x2-up then 3x3) or 1x1 (ksize 1).  gamma/beta NULL = no GroupNorm fold. */
int32_t ucdir_op_conv(const float* x0, int32_t c0, const float* x1, int32_t c1,
                      int32_t B, int32_t H, int32_t W,
                      const float* w_host, const float* bias_host,
                      const float* gamma_host, const float* beta_host,
                      int32_t cout, int32_t ksize, int32_t mode, int32_t silu,
                      const float* residual, float* y, double* stats_out_host, void* stream);
/* conv1 of a residual block + the block's 1x1 res_conv on the same concatenated input in one launch (ABI 3; the launch the
 * UNet's "ups" blocks issue, model/ucdir.py:110,120): y = act(conv3x3(GN(cat[x0,x1]))), yres = conv1x1(cat[x0,x1]) + bres */
int32_t ucdir_op_conv_res(const float* x0, int32_t c0, const float* x1, int32_t c1,
                          int32_t B, int32_t H, int32_t W,
                          const float* w_host, const float* bias_host,
                          const float* gamma_host, const float* beta_host,
                          const float* wres_host, const float* bres_host,
                          int32_t cout, int32_t silu, float* y, float* yres, double* stats_out_host, void* stream);
/* AKGM block tail: y = swish(sum_s spdyconv(GN2(h))[c,s] * att[s]) + res
 * h: (B,C,H,W); att: (B,8,H,W) (= conv2(guide) * attw, already multiplied); res: (B,C,H,W);
 * stats_out_host (ABI 3, may be NULL): (B,2) doubles = the (sum, sum of squares) the launch accumulated for y. */
int32_t ucdir_op_akgm(const float* h, const float* att, const float* res,
                      int32_t B, int32_t C, int32_t H, int32_t W,
                      const float* wsp_host, const float* bsp_host,
                      const float* gamma_host, const float* beta_host,
                      float* y, double* stats_out_host, void* stream);
/* SelfAttention.forward (model/ucdir.py:165-182): y = out(softmax(q^T k / sqrt(C)) v) + x */
int32_t ucdir_op_attention(const float* x, int32_t B, int32_t C, int32_t H, int32_t W,
                           const float* gamma_host, const float* beta_host,
                           const float* wqkv_host, const float* wout_host, const float* bout_host,
                           int32_t fp16, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UCDIR_HIP_H */
