/* wsl_hip.h -- C ABI of libwslhip.so: the MI355X (gfx950) hot path of WSL4MIS' 2-D weakly-supervised
 * training step (dual-branch UNet + scribble losses), as hand-written HIP kernels.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors on the Python side); the library
 *     borrows it for the duration of the enqueue, never allocates, never frees, never synchronises;
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns 0 (WSL_OK) or a negative
 *     WSL_E* code; wsl_last_error() returns a thread-local description of the last failure;
 *   - tensors are fp32 NCHW, contiguous unless a *_bs ("batch stride", in elements) argument says otherwise:
 *     a batch stride lets a channel-slice of a wider tensor be passed without a copy;
 *   - scalar results (losses) are written to device memory; reading them is the caller's (only) sync;
 *   - reductions are two-stage and order-fixed (no float atomics): results are run-to-run reproducible.
 *
 * Citations "ref:" are file:line under the reference checkout's code/ directory.
 */
#ifndef WSL_HIP_H
#define WSL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSL_OK 0
#define WSL_EINVAL (-1)       /* bad shape / null pointer / inconsistent arguments */
#define WSL_EUNSUPPORTED (-2) /* valid in the reference but not built (reported, never ignored) */
#define WSL_EHIP (-3)         /* a HIP runtime error was reported at launch */
#define WSL_EWORKSPACE (-4)   /* workspace smaller than the matching *_ws_bytes() query */

#define WSL_LEAKY_SLOPE 0.01f /* nn.LeakyReLU() default, ref: networks/unet.py:21,25 */

int wsl_version(void);              /* 100*major + minor */
const char* wsl_last_error(void);   /* thread-local, valid until the next failing call on this thread */
const char* wsl_build_info(void);   /* "gfx950 hipcc sha256:<hex>" (product), "gfx950 hipcc EXPERIMENTS ... sha256:<hex>" (tuning build) or
                                     * "HOST-EMULATION (tests only ...) sha256:<hex>"; <hex> = SHA-256 over the library's sources
                                     * (every .hip and .h file of csrc/ sorted by name, then this header) as build.sh saw them */

/* Opt-in measurement: HIP events bracket every launch of the heavy kernel families on the launch stream while
 * enabled; wsl_prof_report() waits for those events (the library's only synchronising call) and fills one row per
 * family with the launch count, summed duration and the ALGORITHMIC flops / bytes those launches covered. */
#define WSL_PROF_FAMILIES 20
typedef struct WslProfRow {
  char name[48];
  int64_t calls;
  double ms, flops, bytes;
  double issued_flops;   /* matrix-core flops actually issued: = flops for the direct kernels, flops / 2.25 for Winograd F(2x2,3x3) */
} WslProfRow;
int wsl_prof_enable(int on);                      /* also clears what was recorded so far */
int wsl_prof_report(WslProfRow* rows, int max_rows);

/* ------------------------------------------------------------------------------------------------ conv stack
 * One channel-range of a convolution's *virtual* input.  The loader applies, per element,
 *     v = x[n, c, y, x]
 *     if (scale)  v = leaky_relu(v * scale[c] + shift[c])      (BatchNorm apply + LeakyReLU of the producer)
 *     if (emask)  v = emask[n, c, y, x] ? v * emask_scale : 0  (nn.Dropout(p): keep mask, scale 1/(1-p))
 *     if (cmask)  v = v * cmask[n, c]                          (F.dropout2d channel multiplier, 0 or 1/(1-p))
 * and then zero-pads.  Two sources concatenated along C reproduce torch.cat([skip, up], 1) without a copy
 * (ref: networks/unet.py:18-26 ConvBlock, :63-68 UpBlock.forward, :254-256 Dropout). */
typedef struct WslSrc {
  const float* x;        /* [N, C, H, W] with batch stride `bs` */
  int64_t bs;            /* elements between consecutive samples (>= C*H*W) */
  int32_t C;             /* 0 = source absent */
  int32_t _pad0;
  const float* scale;    /* [C] or NULL */
  const float* shift;    /* [C] (required iff scale) */
  const uint8_t* emask;  /* [N, C, H, W] dense, or NULL */
  float emask_scale;
  float _pad1;
  const float* cmask;    /* [N, C] or NULL */
} WslSrc;

/* y = conv2d(cat(a, b), w) + bias, stride 1, zero padding ks/2, ks in {1,3} (nn.Conv2d, ref: unet.py:19,23,55,120).
 * wmode 0: w is [Co][Ci][ks][ks] (forward).  wmode 1: data-gradient mode, w is the FORWARD weight
 * [Ci][Co][ks][ks] and the kernel uses w[ci][co][ks-1-ky][ks-1-kx] (what autograd's conv backward computes).
 * If stat_part != NULL the epilogue also emits per-block (sum, M2) of y per channel for BatchNorm
 * (layout [Co][nblk][2] -- channel-major, so the finalize kernel's per-channel scan is contiguous; nblk =
 * wsl_conv2d_stat_blocks(); stat_cnt [nblk] = valid pixels per block). */
int wsl_conv2d_fwd(const WslSrc* a, const WslSrc* b, const float* w, const float* bias, float* y, int64_t y_bs,
                   int N, int H, int W, int Co, int ks, int wmode, float* stat_part, float* stat_cnt, void* stream);
int wsl_conv2d_stat_blocks(int N, int H, int W, int Ci, int Co, int ks);
/* Fast path: wmode 2 (forward) / 3 (data gradient) take `w` = a PACKED weight image [ks*ks][Ci][Co] produced by
 * wsl_conv2d_pack_weights(w_raw, packed, Co, Ci, ks, wmode_raw = 0 or 1) -- for wmode_raw 1 the raw tensor is the
 * forward weight [Ci][Co][ks][ks] and Co/Ci are the data-gradient GEMM's output/input channel counts.  The packed path
 * needs W % 4 == 0 and 16-byte aligned tensors and batch strides (wsl_conv2d_fast_ok() != 0); it stages tiles with
 * aligned float4 loads and prefetches the next channel chunk during the MFMA loop. */
int wsl_conv2d_pack_weights(const float* w, float* packed, int Co, int Ci, int ks, int wmode_raw, void* stream);
int wsl_conv2d_fast_ok(const WslSrc* a, const WslSrc* b, const float* y, int64_t y_bs, int W);
/* Winograd F(2x2, 3x3) path (2.25x fewer matrix instructions than the direct form; fp32 throughout): wmode 4 (forward) /
 * 5 (data gradient) take `w` = the filter image U = G g G^T produced by
 * wsl_conv2d_pack_weights(w_raw, U, Co, Ci, 3, wmode_raw = 2 | 3): 16 * Ci * Co floats in the kernel's operand order
 * (opaque to the caller; needs Ci % 8 == 0 and Co % 16 == 0).  Available for the layers with
 * wsl_conv2d_wino_ok() != 0: ks 3, (Ca + Cb) % 8 == 0 and <= 256, Ca % 8 == 0 when there are two sources, Co % 16 == 0,
 * (H % 8 == 0 and W % 32 == 0) or (H % 16 == 0 and W % 16 == 0), plus the wsl_conv2d_fast_ok() alignment.  Results agree
 * with the direct kernels to fp32 round-off (~1e-6 relative); same BatchNorm partial layout and block count.
 * The product library reads no environment variable and exports no tuning switch; the test hooks that force a tile shape or
 * switch the Winograd path off live in the private header wsl4mis_amd/csrc/wsl_debug.h. */
int wsl_conv2d_wino_ok(int N, int H, int W, int Ca, int Cb, int Co, int ks);
/* Threading option.  unet_cct runs its auxiliary decoder (forward and backward) on a library-owned side stream, forked from /
 * joined to the caller's stream with events, so the two independent decoders fill each other's launch gaps and workgroup
 * tails (+5 % step rate).  One side stream + event pair exists per (device, caller stream).  on = 0 serialises everything on
 * the caller's stream (per-launch timings then do not overlap: what bench.py's roofline segment and profiles/ use); 1 restores
 * the default. */
int wsl_net_concurrent(int on);

/* dw[Co][Ci][ks][ks] = sum_{n,y,x} dy[n,co,y,x] * in[n,ci,y+ky-p,x+kx-p];  db[Co] = sum dy  (db may be NULL).
 * Split over pixels into partials in `ws`, then an order-fixed second stage. */
int wsl_conv2d_wgrad(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, float* dw, float* db,
                     int N, int H, int W, int Co, int ks, void* ws, size_t ws_bytes, void* stream);
size_t wsl_conv2d_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co, int ks);
/* The two stages separately, so that a whole network's second stages (36 tiny launches per step) run as ONE launch:
 * wsl_conv2d_wgrad_partial = stage 1 (partials into ws, which must stay untouched until the batch ran) and a record of the
 * pending stage 2; wsl_wgrad_reduce_batch = stage 2 of n such records, order-fixed sums (bit-reproducible). */
typedef struct WslWgradPending {
  const float* part_dw;
  const float* part_db;
  float* dw;
  float* db;
  int Co, Ci, KK, nsplit;
} WslWgradPending;
int wsl_conv2d_wgrad_partial(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, float* dw, float* db, int N, int H,
                             int W, int Co, int ks, void* ws, size_t ws_bytes, WslWgradPending* pending, void* stream);
int wsl_wgrad_reduce_batch(const WslWgradPending* items, int n, void* stream);

/* BatchNorm2d, training mode (ref: unet.py:20,24; eps 1e-5, momentum 0.1): merge the conv epilogue's partials
 * (Chan's parallel variance), write mean/invstd (saved for backward) and the fused apply coefficients
 * scale = gamma*invstd, shift = beta - mean*scale; update running_mean / running_var (unbiased) /
 * num_batches_tracked in place (any of the three may be NULL). */
int wsl_bn_stats_finalize(const float* stat_part, const float* stat_cnt, int nblk, int C, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                          void* stream);
/* eval mode: scale/shift from the running statistics. */
int wsl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* scale, float* shift, void* stream);
/* out = the loader's transform of one source, materialised (used for returning features / tests). */
int wsl_src_materialize(const WslSrc* s, float* out, int64_t out_bs, int N, int H, int W, void* stream);
/* out[n,c,i,j] = max over the 2x2 window of transform(src) (nn.MaxPool2d(2), ref: unet.py:38). H, W even or odd
 * (floor), window scan row-major, strict '>' (first maximum wins). */
int wsl_pool2_fwd(const WslSrc* s, float* out, int N, int H, int W, void* stream);

/* Gradient that reaches one encoder feature map f = transform(y) from its three consumers:
 *   g[n,c,y,x] = ga[n,c,y,x] + gb[n,c,y,x]*gb_cmask[n,c] + (gp[n,c,y/2,x/2] if (y,x) is the arg-max of its window)
 * ga/gb/gp may each be NULL.  The arg-max is recomputed from `f` (same transform as the forward pool). */
int wsl_feat_grad_combine(const WslSrc* f, const float* ga, int64_t ga_bs, const float* gb, int64_t gb_bs,
                          const float* gb_cmask, const float* gp, float* g, int N, int H, int W, void* stream);

/* Backward of  out = dropout(leaky(bn(y)))  for one conv output y, given g = dL/d(out):
 *   stage 1: per-channel sums  S1 = sum dz, S2 = sum dz*xhat   (dz = g*emask*emask_scale*leaky'(z))
 *   stage 2: dgamma = S2, dbeta = S1, dy = gamma*invstd*(dz - S1/n - xhat*S2/n)
 * ws: wsl_bnact_bwd_ws_bytes().  (autograd of ref: unet.py:18-26) */
int wsl_bnact_bwd(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                  float* dgamma, float* dbeta, int N, int C, int H, int W, void* ws, size_t ws_bytes, void* stream);
size_t wsl_bnact_bwd_ws_bytes(int N, int C, int H, int W);

/* Stage 1 of wsl_bnact_bwd fused into the kernel that PRODUCES g (autograd of ref: unet.py:18-29; saves the reduction pass'
 * 8 bytes per element of HBM traffic).  Producers that can emit the partial sums S1, S2 per workgroup:
 *   wsl_conv2d_dgrad_bn      the data-gradient convolution (wmode 1 | 3 | 5 of wsl_conv2d_fwd, one plain source dy, no bias)
 *                            whose output g [N,Co,H,W] (dense) is dL/d(out) of the BatchNorm that normalised bn_y [N,Co,H,W];
 *                            bn_st = mean | invstd | scale | shift (4 * Co floats), bn_emask the nn.Dropout keep mask or
 *                            NULL.  *fused = 1: bn_part holds [Co][wsl_conv2d_stat_blocks(N,H,W,Ci,Co,ks)][2] partials
 *                            (channel-major); *fused = 0: the kernel this shape dispatches to has no such epilogue, g is
 *                            written as usual and the caller runs wsl_bnact_bwd().
 *   wsl_feat_grad_combine_bn wsl_feat_grad_combine() for a feature f = leaky(bn(y)) (plain scale/shift source): bn_part holds
 *                            [wsl_feat_grad_combine_blocks(N,H,W)][C][2] partials (block-major).
 * wsl_bnact_bwd_finish = stage 2 from such partials: dgamma, dbeta, dy.  ws: 2 * C floats. */
int wsl_conv2d_dgrad_bn(const WslSrc* dy, const float* w, float* g, int64_t g_bs, int N, int H, int W, int Co, int ks,
                        int wmode, const float* bn_y, const float* bn_st, const uint8_t* bn_emask, float bn_emask_scale,
                        float* bn_part, int* fused, void* stream);
int wsl_feat_grad_combine_blocks(int N, int H, int W);
int wsl_feat_grad_combine_bn(const WslSrc* f, const float* ga, int64_t ga_bs, const float* gb, int64_t gb_bs,
                             const float* gb_cmask, const float* gp, float* g, int N, int H, int W, const float* bn_mean,
                             const float* bn_invstd, float* bn_part, void* stream);
int wsl_bnact_bwd_finish(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                         float* dgamma, float* dbeta, int N, int C, int H, int W, const float* part, int nblk,
                         int channel_major, void* ws, size_t ws_bytes, void* stream);
/* Round 6: the same pair with the ACTIVATION + DROPOUT backward moved into the producer as well (autograd of ref: unet.py:18-29).  The
 * data-gradient epilogue forms dz = g * keep * scale * leaky'(z) of every element for its sums; wsl_conv2d_dgrad_bn_d lets it WRITE dz where
 * it would have written g.  *fused = 2: g holds dz, statistics in bn_part -- finish with wsl_bnact_bwd_finish_d_amax, whose pass reads dz
 * and y only (12 instead of 13 bytes per element, no keep-mask unpacking, no select).  *fused = 1 / 0: exactly wsl_conv2d_dgrad_bn's
 * meaning (g is the plain gradient: the kernel this launch dispatched to does not hold the consumer's y / keep bytes when it stores). */
int wsl_conv2d_dgrad_bn_d(const WslSrc* dy, const float* w, float* g, int64_t g_bs, int N, int H, int W, int Co, int ks,
                          int wmode, const float* bn_y, const float* bn_st, const uint8_t* bn_emask, float bn_emask_scale,
                          float* bn_part, int* fused, void* stream);
int wsl_bnact_bwd_finish_d_amax(const float* dz, int64_t dz_bs, const float* y, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta, int N, int C, int H,
                                int W, const float* part, int nblk, int channel_major, void* ws, size_t ws_bytes, uint32_t* dy_amax,
                                void* stream);

/* ------------------------------------------------------------------------------------------------ split-precision conv path
 * (opt-in; ref semantics unchanged: networks/unet.py:13-29.)  The 3x3 convolutions on the f16 matrix cores
 * (16x the f32 MFMA rate) with fp32-class accuracy: while an operand tile is staged it is split into
 *     x * 2^e = hi + lo,   hi = f16(x * 2^e) rounded toward zero,  lo = f16 of the exact remainder   (21-22 significant bits)
 * and the kernels issue three passes  hi*hi + hi*lo + lo*hi  of v_mfma_f32_16x16x32_f16 into ONE fp32 accumulator; the epilogue
 * undoes the power-of-two scales (exact) and is otherwise the f32 kernels' (bias, BatchNorm partial statistics, BatchNorm-backward
 * statistics).  Scales: weights per layer from max |w| (wsl_sp_pack_weights writes the bit pattern into *w_amax); gradient tensors
 * from max |dy| as left by their producer (wsl_bnact_bwd*_amax below); activations (everything a forward convolution or the
 * weight gradient reads through a WslSrc) a fixed 2^WSL_SP_ACT_EXP: f16 then holds |v| < 4094 -- BatchNorm-normalised values
 * cannot reach that, and a larger value saturates instead of overflowing.  Results are independent of the scale chosen as long as
 * nothing under- or overflows.  Eligible layers (wsl_sp_conv2d_ok): ks 3, (Ca + Cb) % 16 == 0 (Ca % 16 == 0 with two sources),
 * Co % 16 == 0, H % 8 == 0 and W % 16 == 0, float4-aligned tensors.
 * A `w_amax` argument is the DEVICE address of one uint32, the bit pattern of a non-negative float; a `dy_amax` / `in_amax`
 * argument is the device address of WSL_SP_AMAX_SLOTS such words whose maximum is the tensor's (the producer's workgroups spread
 * their integer atomic max over the slots -- order-independent, so results stay run-to-run reproducible); the caller zeroes the
 * words before the producer runs. */
#define WSL_SP_ACT_EXP 4
#define WSL_SP_AMAX_SLOTS 64
int wsl_sp_conv2d_ok(const WslSrc* a, const WslSrc* b, const float* y, int64_t y_bs, int N, int H, int W, int Co, int ks);
/* one weight image (hi and lo halves in the kernels' operand order; the 9 taps padded to 10): 40 * Ci * Co bytes */
size_t wsl_sp_weight_image_bytes(int Co, int Ci);
/* w: the layer's raw weight.  dgrad 0: [Co][Ci][3][3] -> forward image.  dgrad 1: w is the FORWARD weight [Ci][Co][3][3] and Co / Ci
 * are the data-gradient GEMM's output / input channel counts (as wsl_conv2d_pack_weights wmode_raw 1).  Two launches. */
int wsl_sp_pack_weights(const float* w, void* image, uint32_t* w_amax, int Co, int Ci, int dgrad, void* stream);
/* y = conv2d(cat(a, b), w) + bias on the split path.  in_amax NULL: the sources are activations (fixed scale); else the scale of
 * the (plain) source comes from *in_amax.  stat_part / stat_cnt as wsl_conv2d_fwd ([Co][nblk][2], nblk = wsl_sp_conv2d_stat_blocks). */
int wsl_sp_conv2d_fwd(const WslSrc* a, const WslSrc* b, const void* image, const uint32_t* w_amax, const uint32_t* in_amax,
                      const float* bias, float* y, int64_t y_bs, int N, int H, int W, int Co, float* stat_part, float* stat_cnt,
                      void* stream);
int wsl_sp_conv2d_stat_blocks(int N, int H, int W, int Ci, int Co);
/* wsl_conv2d_dgrad_bn on the split path (image = the data-gradient image); *fused as there. */
int wsl_sp_conv2d_dgrad_bn(const WslSrc* dy, const uint32_t* dy_amax, const void* image, const uint32_t* w_amax, float* g,
                           int64_t g_bs, int N, int H, int W, int Co, const float* bn_y, const float* bn_st,
                           const uint8_t* bn_emask, float bn_emask_scale, float* bn_part, int* fused, void* stream);
/* wsl_conv2d_wgrad_partial on the split path: dy scaled from *dy_amax, the sources a / b with the activation scale. */
size_t wsl_sp_conv2d_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co);
int wsl_sp_conv2d_wgrad_partial(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, const uint32_t* dy_amax,
                                float* dw, float* db, int N, int H, int W, int Co, void* ws, size_t ws_bytes,
                                WslWgradPending* pending, void* stream);
/* Raw (scale == NULL) FORWARD sources -- the upsampled tensor of a decoder block (ref: networks/unet.py:63-68: x1 = up(conv1x1(x1)),
 * cat([x2, x1])) -- are not BatchNorm-normalised, so the static activation scale could saturate them silently.  Their maximum
 * is tracked: wsl_bilinear_up2_fwd_amax leaves max |u| (>= max of the bilinear output: a convex combination) in amax_slots
 * [WSL_SP_AMAX_SLOTS]; wsl_sp_conv2d_fwd takes it as `in_amax` (with a BatchNorm source present the operand scale is then
 * min(2^WSL_SP_ACT_EXP, the scale of that maximum): unchanged results unless the raw source really exceeds the static range) and
 * wsl_sp_conv2d_wgrad_partial_amax as `in_amax`.  ws: wsl_bilinear_up2_fwd_amax_ws_bytes (partial maxima); amax_slots NULL = the plain call. */
size_t wsl_bilinear_up2_fwd_amax_ws_bytes(int N, int C, int h, int w);
int wsl_bilinear_up2_fwd_amax(const float* u, float* out, int64_t out_bs, int N, int C, int h, int w, void* ws, size_t ws_bytes,
                              uint32_t* amax_slots, void* stream);
int wsl_sp_conv2d_wgrad_partial_amax(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, const uint32_t* dy_amax,
                                     const uint32_t* in_amax, float* dw, float* db, int N, int H, int W, int Co, void* ws,
                                     size_t ws_bytes, WslWgradPending* pending, void* stream);
/* wsl_bnact_bwd / wsl_bnact_bwd_finish that also leave max |dy| in dy_amax[0 .. WSL_SP_AMAX_SLOTS) (NULL = the plain calls; every
 * slot is written, none needs clearing).  Workspace of the finish form: wsl_bnact_bwd_finish_ws_bytes(..., dy_amax != NULL). */
size_t wsl_bnact_bwd_finish_ws_bytes(int N, int C, int H, int W, int with_amax);
int wsl_bnact_bwd_amax(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, const uint8_t* emask, float emask_scale, float* dy, float* dgamma, float* dbeta, int N,
                       int C, int H, int W, void* ws, size_t ws_bytes, uint32_t* dy_amax, void* stream);
int wsl_bnact_bwd_finish_amax(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                              float* dgamma, float* dbeta, int N, int C, int H, int W, const float* part, int nblk,
                              int channel_major, void* ws, size_t ws_bytes, uint32_t* dy_amax, void* stream);

/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (ref: unet.py:56-57) and its transpose. */
int wsl_bilinear_up2_fwd(const float* u, float* out, int64_t out_bs, int N, int C, int h, int w, void* stream);
int wsl_bilinear_up2_bwd(const float* dout, int64_t dout_bs, float* du, int N, int C, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------ losses
 * Labels: uint8 (label_i64 == 0) or int64 (label_i64 == 1), values 0..C-1 or `ignore`. C <= 8. */
int wsl_softmax_fwd(const float* z, float* s, int N, int C, int HW, void* stream);
int wsl_softmax_bwd(const float* s, const float* ds, float* dz, int N, int C, int HW, void* stream);

/* torch.nn.CrossEntropyLoss(ignore_index) (ref: train_weakly_supervised_pCE_2D.py:81,100): loss[0] = mean NLL over
 * non-ignored pixels (NaN if none), dz = gscale*(softmax - onehot)/n_valid on valid pixels, 0 elsewhere. */
int wsl_ce_fwd_bwd(const float* z, const void* label, int label_i64, int ignore, float* loss, float* dz, float gscale,
                   int N, int C, int HW, void* ws, size_t ws_bytes, void* stream);
/* pseudo = argmax_c(beta*s1 + (1-beta)*s2), two fp32 multiplies + one add, no FMA contraction, first index on
 * ties -- bit-exact with torch (ref: ...pCE_ours_proposed.py:117-120).  beta is the python double. */
int wsl_mix_argmax(const float* s1, const float* s2, double beta, int64_t* pseudo, int N, int C, int HW, void* stream);
/* utils.losses.pDLoss(n_classes, ignore_index).forward (ref: utils/losses.py:195-232) including its
 * [N,H,W]x[N,1,H,W] broadcast (sums weighted by the batch-summed ignore mask), and DiceLoss (ignore < 0:
 * ref: utils/losses.py:156-192).  sums[3*C] = {I, Z, Y} per class kept for the backward. */
int wsl_pdice_fwd(const float* s, const void* target, int target_i64, int ignore, float* loss, float* sums, int N,
                  int C, int HW, void* ws, size_t ws_bytes, void* stream);
int wsl_pdice_bwd(const float* s, const void* target, int target_i64, int ignore, const float* sums,
                  const float* gout /* device scalar or NULL (=1) */, float* ds, int N, int C, int HW, void* stream);

/* Fused loss head of `ours_proposed` on logits (ref: ...pCE_ours_proposed.py:110-125):
 *   s_k = softmax(z_k); ce = 0.5*(CE(z1,l)+CE(z2,l)); pseudo = argmax(beta*s1+(1-beta)*s2);
 *   pse = 0.5*(pDice(s1,pseudo)+pDice(s2,pseudo)); loss = ce + w_pse*pse.
 * out[0..3] = {loss, ce, pse, n_valid}.  dz1/dz2 = dloss/dz (times gscale).  pseudo may be NULL.
 * z2 == NULL runs the single-branch pCE form (unet): loss = CE(z1, l). */
int wsl_head_fwd_bwd(const float* z1, const float* z2, const uint8_t* label, int ignore, double beta, float w_pse,
                     float gscale, float* out, int64_t* pseudo, float* dz1, float* dz2, int N, int C, int HW,
                     void* ws, size_t ws_bytes, void* stream);
size_t wsl_loss_ws_bytes(int N, int C, int HW);
/* The headline composition in one call (ref: ..._pCE_GatedCRFLoss_2D.py:108-123; dual branch: train_ACDC_scribblevc.py:171-206):
 * loss = pCE(z1 [, z2]) + crf_weight * GatedCRF(y, img), y = beta*softmax(z1) + (1-beta)*softmax(z2) (z2 == NULL: softmax(z1)).
 * Equals wsl_head_fwd_bwd(w_pse 0) + wsl_mixprob_fwd + wsl_gatedcrf_fwd + wsl_mixprob_bwd (to the last ulp), with y written by the
 * head's reduction pass and the gradient through y added inside the head's backward pass (two launches and the re-reads of
 * the logits / logit gradients fewer).  out[0..3] as wsl_head_fwd_bwd, out[4] = raw GatedCRF loss; y, msg: [N,C,H,W]. */
int wsl_head_gatedcrf_fwd_bwd(const float* z1, const float* z2, const uint8_t* label, int ignore, double beta, const float* img,
                              int radius, float sigma_xy, float sigma_rgb, float weight, float crf_weight, float* out,
                              float* dz1, float* dz2, float* y, float* msg, int N, int C, int H, int W, void* ws,
                              size_t ws_bytes, void* stream);

/* y = beta*softmax(z1) + (1-beta)*softmax(z2) (z2 == NULL: softmax(z1)) -- the prediction the GatedCRF term regularises
 * in the dual-branch composition (ref: train_ACDC_scribblevc.py:171-206; single branch: ...pCE_GatedCRFLoss_2D.py:112).
 * Backward: dz_k (+)= softmax_bwd(s_k, w_k * k * dy), w_1 = beta, w_2 = 1-beta; accumulate != 0 adds into dz. */
int wsl_mixprob_fwd(const float* z1, const float* z2, double beta, float* y, int N, int C, int HW, void* stream);
int wsl_mixprob_bwd(const float* z1, const float* z2, double beta, const float* dy, float k, float* dz1, float* dz2,
                    int accumulate, int N, int C, int HW, void* stream);

/* ModelLossSemsegGatedCRF.forward, one {'weight','xy','rgb'} descriptor, Potts model, no masks, prediction at input
 * resolution (ref: utils/gate_crf_loss.py:20-124,135-188).  loss[0] = (sum K - sum y*msg)/(N*H*W);
 * msg [N,C,H,W] is kept: dL/dy = -2*msg/(N*H*W) (wsl_gatedcrf_bwd scales it by gout). radius 1..8, C <= 8. */
int wsl_gatedcrf_fwd(const float* y, const float* img, float* msg, float* loss, int N, int C, int H, int W, int radius,
                     float sigma_xy, float sigma_rgb, float weight, void* ws, size_t ws_bytes, void* stream);
int wsl_gatedcrf_bwd(const float* msg, const float* gout, float gscale, float* dy, int N, int C, int H, int W,
                     void* stream);
/* The single-branch regulariser compositions in ONE call: partial CE + reg_weight * R(softmax(z))
 * [+ cons_weight * mean((softmax(z) - softmax(zt))^2)], R = tv_loss(softmax[1:]) | MumfordShah_Loss(image, softmax) | entropy_loss(softmax, C)
 * (ref: train_weakly_supervised_pCE_TV_2D.py:108-114, ..._pCE_MumfordShah_Loss_2D.py:97-107, ..._pCE_Entropy_Mini_2D.py:99-102,
 * train_mean_teacher_2D.py:147-171).  The head's first pass keeps softmax(z) in `s`, the regulariser kernels turn it into the weighted
 * gradient `ds`, the head's second pass writes dz = w_ce * dCE/dz + softmax_backward(s, ds) once: instead of the chain
 * head -> softmax -> R -> softmax-backward -> axpy (-> softmax-MSE -> axpy) three to five launches and as many passes over [N,C,H,W] fewer.
 * out[0..3] as wsl_head_fwd_bwd (single branch), out[4] = R (unweighted), out[5] = the consistency term (unweighted; only with zt). */
#define WSL_REG_TV 1
#define WSL_REG_MS 2
#define WSL_REG_ENTROPY 3
int wsl_head_reg_fwd_bwd(const float* z, const uint8_t* label, int ignore, float w_ce, int reg_kind, float reg_weight,
                         const float* img, const float* zt, float cons_weight, float* out, float* dz, float* s, float* ds, int N,
                         int C, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* tv_loss(p) (ref: train_weakly_supervised_pCE_TV_2D.py:58-65) on p[n0:] (n0 = 1 reproduces outputs_soft[1:]). */
int wsl_tv_fwd_bwd(const float* p, int n0, float* loss, float* dp, float gscale, int N, int C, int H, int W, void* ws,
                   size_t ws_bytes, void* stream);
/* MumfordShah_Loss().forward(image, prediction) (ref: utils/losses.py:275-309). */
int wsl_mumford_shah_fwd_bwd(const float* img, const float* p, float* loss, float* dp, float gscale, int N, int C, int H,
                             int W, void* ws, size_t ws_bytes, void* stream);
/* mean((softmax(a)-softmax(b))^2) and its gradient wrt a (ref: utils/losses.py:65-82, train_mean_teacher_2D.py:164). */
int wsl_softmax_mse_fwd_bwd(const float* a, const float* b, float* loss, float* da, float gscale, int N, int C, int HW,
                            void* ws, size_t ws_bytes, void* stream);

/* dst += k * src (sums the logit-gradients of several loss terms). */
/* Uncertainty-aware mean teacher (ref: train_weakly_supervised_ustm_2D.py:121-157).
 * wsl_rot90: torch.rot90(x, k, [2,3]) of `planes` [H,W] planes (output planes [W,H] for odd k; x != y).
 * wsl_softmax_accum: acc = (init ? 0 : acc) + scale * softmax(z) -- the mean of the T stochastic teacher predictions.
 * wsl_ustm_consistency_fwd_bwd: mask = [-sum_c pm log(pm + 1e-6) < threshold] per pixel (pm = pmean [N,C,HW]);
 *   loss[0] = sum(mask * (softmax(a) - softmax(b))^2) / (2 sum(mask) + 1e-16), loss[1] = sum(mask), loss[2] = 1/(2 sum(mask)+1e-16);
 *   da = gscale * dloss/da (no gradient to b or the mask).  `loss` holds 3 floats. */
int wsl_rot90(const float* x, float* y, int planes, int H, int W, int k, void* stream);
int wsl_softmax_accum(const float* z, float* acc, float scale, int init, int N, int C, int HW, void* stream);
int wsl_ustm_consistency_fwd_bwd(const float* a, const float* b, const float* pmean, float threshold, float* loss, float* da,
                                 float gscale, int N, int C, int HW, void* ws, size_t ws_bytes, void* stream);
/* entropy_loss(p, C) = mean over pixels of -sum_c p log(p + 1e-6), divided by log(C) (ref: utils/losses.py:30-36);
 * dp = gscale * dloss/dp.  p is [N,C,HW] (already a softmax). */
int wsl_entropy_fwd_bwd(const float* p, float* loss, float* dp, float gscale, int N, int C, int HW, int norm_classes,
                        void* ws, size_t ws_bytes, void* stream);
int wsl_axpy(float* dst, const float* src, float k, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------ optimiser
 * torch.optim.SGD(momentum, weight_decay) over a flat arena (ref: ...pCE_ours_proposed.py:89-90,126-132):
 *   g = grad*grad_scale + wd*p; buf = first ? g : mu*buf + g; p -= lr*buf;
 *   if (ema) ema = ema_alpha*ema + (1-ema_alpha)*p      (update_ema_variables, ref: ..._ustm_2D.py:61-65) */
int wsl_sgd_step(float* p, const float* grad, float* buf, int64_t n, float lr, float momentum, float wd, int first,
                 float grad_scale, float* ema, float ema_alpha, void* stream);

/* ------------------------------------------------------------------------------------------------ data path
 * The per-slice augmentation of RandomGenerator (ref: code/dataloaders/dataset_semi.py:128-171) for a batch of slices
 * of different native sizes, one gather per output pixel:  op 0 nothing | 1 np.rot90(k) then np.flip(axis) |
 * 2 scipy.ndimage.rotate(order 0, reshape=False, constant cval) with the host-computed 2x2 matrix m and offset
 * (rows first, as scipy builds them from cosdg/sindg and the centres (shape-1)/2);  then scipy.ndimage.zoom(order 0) to
 * [Ho, Wo].  Image out [n,1,Ho,Wo] float32, label out [n,Ho,Wo] uint8.  The descriptor array lives on the HOST
 * (copied into the launch); img / lab are device pointers.  Results equal numpy/scipy bit for bit. */
typedef struct {
  const float* img;    /* [h, w] */
  const uint8_t* lab;  /* [h, w] */
  int h, w;
  int op, k, axis;     /* op 1: k in 0..3, axis in {0,1} */
  int lab_cval;        /* op 2: fill of the label outside the rotated image (4 for scribbles, else 0) */
  float img_cval;      /* op 2: fill of the image (0) */
  double m00, m01, m10, m11, off0, off1;   /* op 2: input coord = m * output coord + off */
} WslAugSample;
int wsl_augment_batch(const WslAugSample* samples, int n, float* out_img, uint8_t* out_lab, int Ho, int Wo, void* stream);

/* Validation metric pieces (medpy.metric.binary.hd95 as called by code/val_2D.py:7-15): the surface of a binary [D,H,W]
 * volume (object minus its erosion by the 6-neighbourhood, background outside the array) and, for every point of one
 * voxel list ([n][3] int64 z,y,x -- torch.nonzero layout), the exact squared distance to the nearest point of another.
 * D == 0 denotes a 2-D [H,W] array: medpy then erodes with the 4-neighbourhood (no z test). */
int wsl_surface_u8(const uint8_t* vol, uint8_t* border, int D, int H, int W, void* stream);
int wsl_nearest_dist2(const int64_t* a_zyx, int na, const int64_t* b_zyx, int nb, int64_t* out, void* stream);

/* Noisy copies of a batch for the mean-teacher / USTM teachers (ref: train_mean_teacher_2D.py:147-149, ..._ustm_2D.py:125-135):
 * out[r*n + i] = x[i] + d,  r < reps, with d = noise[r*n + i] when `noise` is given (parity tests replay the reference's draw)
 * or clamp(N(0,1) * sigma, -clip, clip) drawn by the library (Philox4x32-10 + Box-Muller, reproducible per seed). */
int wsl_noisy_copy(const float* x, const float* noise, float* out, int64_t n, int reps, float sigma, float clip, uint64_t seed,
                   void* stream);

/* Bernoulli masks for nn.Dropout / F.dropout2d in ONE launch (Philox4x32-10, counter-based: reproducible per seed).
 * Mask i: is_f32[i] == 0 -> uint8 keep mask (1 with probability keep_probs[i]); == 1 -> float multiplier
 * (scales[i] with probability keep_probs[i], else 0).  uint8 outputs must be 4-byte aligned.  n_masks <= 12.
 * uint8 masks draw sixteen random bits per element: their keep probability is keep_probs[i] ROUNDED to the nearest 1 / 65536
 * (1.0 keeps every element, 0.0 none); float multipliers use 32 bits (keep_probs[i] to 2^-32; 1.0 drops one draw in 2^32). */
int wsl_draw_masks(int n_masks, void* const* outs, const int64_t* numels, const float* keep_probs, const float* scales,
                   const int* is_f32, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------------ network
 * UNet / UNet_CCT (ref: networks/unet.py:286-303, 327-346; factory: networks/net_factory.py:6-22) over a flat fp32
 * parameter arena whose entry order is the reference module's parameters() order and whose names are its
 * state_dict keys.  Buffers (running_mean/var) live in a second arena, num_batches_tracked in an int64 array. */
typedef struct WslNetDesc {
  int32_t in_chns, n_class;
  int32_t n_dec;       /* 1 = 'unet', 2 = 'unet_cct' (main + aux decoder) */
  int32_t N, H, W;     /* H, W multiples of 16 */
  int32_t precision;   /* 0 = fp32 kernels (default); 1 = the eligible 3x3 layers on the split-precision path (f16 hi / lo
                          operands, three MFMA passes, fp32 accumulate -- "split-precision conv path" above); same arenas, same
                          results to fp32 round-off, a larger workspace (wsl_net_ws_bytes knows) */
  int32_t _pad;
} WslNetDesc;

typedef struct WslNetEntry {
  char name[96];       /* state_dict key */
  int32_t kind;        /* 0 parameter (fp32, param arena), 1 buffer (fp32, buffer arena), 2 num_batches_tracked */
  int32_t ndim;
  int64_t shape[4];
  int64_t offset;      /* element offset in its arena */
} WslNetEntry;

int wsl_net_num_entries(const WslNetDesc* d);
int wsl_net_entry(const WslNetDesc* d, int i, WslNetEntry* out);
int64_t wsl_net_param_count(const WslNetDesc* d);    /* 2,447,064 for unet_cct(1,4); 1,813,764 for unet(1,4) */
int64_t wsl_net_encoder_param_count(const WslNetDesc* d);
int64_t wsl_net_buffer_count(const WslNetDesc* d);
size_t wsl_net_ws_bytes(const WslNetDesc* d);

/* Forward.  emasks[5]: uint8 keep masks of the encoder's nn.Dropout sites ([N,C_l,H_l,W_l]; training only);
 * cmasks[5]: [N,C_l] multipliers of the aux branch's F.dropout2d (required iff n_dec == 2 -- the reference applies
 * it in eval mode too).  Writes logits [N,n_class,H,W] (aux may be NULL iff n_dec == 1); keeps what the backward
 * needs inside `ws`. */
int wsl_net_forward(const WslNetDesc* d, const float* params, float* buffers, int64_t* nbt, const float* x,
                    const uint8_t* const* emasks, const float* const* cmasks, int training, float* logits_main,
                    float* logits_aux, void* ws, size_t ws_bytes, void* stream);
/* Backward of the last training forward held in `ws`.  grads: flat arena, same layout as params (overwritten).
 * phase: 0 = everything, 1 = decoders only (their grads are final when it returns), 2 = encoder only (after 1) --
 * the split lets the caller start the decoder bucket's all-reduce while the encoder backward runs. */
int wsl_net_backward(const WslNetDesc* d, const float* params, const float* x, const uint8_t* const* emasks,
                     const float* const* cmasks, const float* dlogits_main, const float* dlogits_aux, float* grads,
                     void* ws, size_t ws_bytes, int phase, void* stream);

/* ------------------------------------------------------------------------------------------------ transposed-conv UpBlock
 * (opt-in: ref networks/unet.py:47-68 with bilinear=False -- a branch the reference's Decoder never selects.)
 * nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2): out[n][co][2i+a][2j+b] = bias[co] + sum_ci x[n][ci][i][j] w[ci][co][a][b],
 * its data gradient and its weight gradient (ws: wsl_convt2x2_wgrad_ws_bytes; per-sample partials summed in sample order). */
int wsl_convt2x2_fwd(const float* x, const float* w, const float* bias, float* out, int N, int Ci, int Co, int h, int wd,
                     void* stream);
int wsl_convt2x2_dgrad(const float* dy, int64_t dy_bs, const float* w, float* dx, int N, int Ci, int Co, int h, int wd,
                       void* stream);
size_t wsl_convt2x2_wgrad_ws_bytes(int N, int Ci, int Co);
int wsl_convt2x2_wgrad(const float* x, const float* dy, int64_t dy_bs, float* dw, float* db, int N, int Ci, int Co, int h, int wd,
                       void* ws, size_t ws_bytes, void* stream);
/* The whole block: x1 [N,C1,h,w], x2 [N,C2,2h,2w] -> ConvBlock(2 C2, Co, dropout_p)(cat([x2, ConvTranspose2d(x1)], 1)), with
 * the reference module's parameter order (up.weight [C1][C2][2][2], up.bias, conv.conv_conv.{0,1,4,5}.{weight,bias}) in one
 * arena; buffers = running mean / var of the two BatchNorms (4 Co floats), nbt = their two num_batches_tracked.  forward keeps
 * what backward needs in ws (same ws for both calls). */
typedef struct WslUpBlockDesc {
  int32_t C1, C2, Co, N, h, w;
  float dropout_p;
} WslUpBlockDesc;
int64_t wsl_upblock_t_param_count(const WslUpBlockDesc* d);
size_t wsl_upblock_t_ws_bytes(const WslUpBlockDesc* d);
int wsl_upblock_t_forward(const WslUpBlockDesc* d, const float* params, float* buffers, int64_t* nbt, const float* x1,
                          const float* x2, const uint8_t* emask, int training, float* out, void* ws, size_t ws_bytes,
                          void* stream);
int wsl_upblock_t_backward(const WslUpBlockDesc* d, const float* params, const float* x1, const float* x2, const uint8_t* emask,
                           const float* dout, float* grads, float* dx1, float* dx2, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WSL_HIP_H */
