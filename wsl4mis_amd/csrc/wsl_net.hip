// UNet / UNet_CCT forward + backward over flat arenas (ref: networks/unet.py:71-135 Encoder/Decoder, :286-303 UNet,
// :327-346 UNet_CCT; factory networks/net_factory.py:6-22).  Host-side sequencing only: every arithmetic step is one of
// the kernels of this library enqueued on the caller's stream; nothing here allocates or synchronises.
//
// What is kept in HBM between forward and backward: the raw (pre-BatchNorm) output of every convolution, the pooled
// encoder inputs, the 1x1-conv / upsampled decoder tensors and 4*C BatchNorm coefficients per layer.  Normalised /
// activated / dropped-out / concatenated tensors are never materialised: consumers rebuild them while staging tiles.
#include <stdio.h>
#include <string.h>
#include <initializer_list>

#include "wsl_rt.h"

namespace wsl {

int sp_pack_table(const PackTable& t, const int64_t* img_off_bytes, const float* params, void* imgf, void* imgd, uint32_t* amax,
                  int with_dgrad, void* stream);   // wsl_convsp.hip

static const int kFt[5] = {16, 32, 64, 128, 256};            // unet.py:291
static const float kDrop[5] = {0.05f, 0.1f, 0.2f, 0.3f, 0.5f};  // unet.py:292
static const float kEps = 1e-5f, kMom = 0.1f;
constexpr int kSpMaxLayers = 40;   // = PackTable's capacity: conv layers of one network (36 in unet_cct)

struct ConvRef { int64_t w, b; int Ci, Co, ks; int li; };   // li: index in the pack table
struct BnRef { int64_t gamma, beta, rmean, rvar; int nbt, C; };
struct BlockRef { ConvRef c1, c2; BnRef b1, b2; };
struct BlkWs { size_t y1, y2, st1, st2; };  // st*: mean | invstd | scale | shift (4*C floats)

struct Plan {
  WslNetDesc d;
  int H[5], W[5];
  BlockRef enc[5];
  struct Dec { ConvRef c1x1[4]; BlockRef blk[4]; ConvRef out; } dec[2];
  int64_t n_param, n_enc_param, n_buf, n_bn;
  // workspace (float offsets)
  BlkWs wenc[5];
  size_t pooled[5];
  struct DecWs { size_t u[4], up[4], dcat[4], glow4; BlkWs blk[4]; } wdec[2];
  // scratch reused from layer to layer; a second set lets the two decoders of unet_cct run on two streams
  struct Scratch { size_t tmp_g, tmp_g1, tmp_dy, tmp_du, tmp_gpool, stat_part, stat_cnt, wg_ws, bn_ws, bn_coef; } scr[2];
  size_t packf, packd;
  size_t winof, winod;   // Winograd filter images [16][Ci][Co] of the 3x3 layers, at twice the raw weight's offset
  // split-precision path (d.precision == 1): f16 hi / lo weight images (10 floats per 9 raw weights: sp_img_off()), one max |w|
  // slot per conv layer (pack-table order) and one max |dy| slot per BatchNorm layer
  size_t spf, spd, sp_wmax, sp_dymax, sp_upmax;   // sp_upmax: WSL_SP_AMAX_SLOTS max |u| words per decoder stage (the raw upsampled source)
  int sp;
  size_t wg_bytes, bn_bytes, bn_coef_bytes, total_floats;   // wg_bytes: per scratch set, room for the partials of EVERY layer of a phase
  // named regions of the workspace, in allocation order (wsl_debug_net_ws_region: the tools that compare two runs' workspaces)
  // (recorded only when want_regions is set -- by wsl_debug_net_ws_region alone: make_plan runs several times per training step and has no
  //  use for a hundred formatted names; overflow is an error of the debug entry point, never a silent truncation)
  struct Region { char name[48]; size_t off, n; } regs[192];
  int nregs;
  bool want_regions = false, regs_overflow = false;
};

struct Bump {
  size_t off = 0;
  size_t take(size_t n) {
    const size_t o = off;
    off += (n + 63) & ~(size_t)63;
    return o;
  }
};

static void plan_conv(ConvRef& c, int Ci, int Co, int ks, int64_t& po) {
  c.Ci = Ci, c.Co = Co, c.ks = ks;
  c.w = po, po += (int64_t)Co * Ci * ks * ks;
  c.b = po, po += Co;
}
static void plan_bn(BnRef& b, int C, int64_t& po, int64_t& bo, int64_t& nbn) {
  b.C = C;
  b.gamma = po, po += C;
  b.beta = po, po += C;
  b.rmean = bo, bo += C;
  b.rvar = bo, bo += C;
  b.nbt = (int)nbn++;
}
static void plan_block(BlockRef& k, int Ci, int Co, int64_t& po, int64_t& bo, int64_t& nbn) {
  plan_conv(k.c1, Ci, Co, 3, po);
  plan_bn(k.b1, Co, po, bo, nbn);
  plan_conv(k.c2, Co, Co, 3, po);
  plan_bn(k.b2, Co, po, bo, nbn);
}

static int make_plan(const WslNetDesc* d, Plan& P) {
  WSL_REQUIRE(d, "net: null descriptor");
  WSL_REQUIRE(d->n_dec == 1 || d->n_dec == 2, "net: n_dec must be 1 (unet) or 2 (unet_cct)");
  WSL_REQUIRE(d->in_chns > 0 && d->n_class > 0 && d->n_class <= 8, "net: bad channel counts");
  WSL_REQUIRE(d->N > 0 && d->H >= 16 && d->W >= 16 && d->H % 16 == 0 && d->W % 16 == 0,
              "net: N=%d H=%d W=%d (H, W must be multiples of 16)", d->N, d->H, d->W);
  WSL_REQUIRE(d->precision == 0 || d->precision == 1, "net: precision %d (0 = f32, 1 = split f16 x 3)", d->precision);
  P.d = *d;
  P.sp = d->precision;
  for (int l = 0; l < 5; ++l) P.H[l] = d->H >> l, P.W[l] = d->W >> l;
  int64_t po = 0, bo = 0, nbn = 0;
  plan_block(P.enc[0], d->in_chns, kFt[0], po, bo, nbn);
  for (int l = 1; l < 5; ++l) plan_block(P.enc[l], kFt[l - 1], kFt[l], po, bo, nbn);
  P.n_enc_param = po;
  for (int k = 0; k < d->n_dec; ++k) {
    for (int i = 0; i < 4; ++i) {  // stage i+1 ("up{i+1}"): level 3-i, C1 = ft[4-i] -> C2 = ft[3-i]
      const int c1 = kFt[4 - i], c2 = kFt[3 - i];
      plan_conv(P.dec[k].c1x1[i], c1, c2, 1, po);
      plan_block(P.dec[k].blk[i], 2 * c2, c2, po, bo, nbn);
    }
    plan_conv(P.dec[k].out, kFt[0], d->n_class, 3, po);
  }
  P.n_param = po, P.n_buf = bo, P.n_bn = nbn;
  {
    int li = 0;
    for (int l = 0; l < 5; ++l) P.enc[l].c1.li = li++, P.enc[l].c2.li = li++;
    for (int k = 0; k < d->n_dec; ++k) {
      for (int i = 0; i < 4; ++i) P.dec[k].c1x1[i].li = li++, P.dec[k].blk[i].c1.li = li++, P.dec[k].blk[i].c2.li = li++;
      P.dec[k].out.li = li++;
    }
  }

  // ---- workspace
  Bump B0;
  P.nregs = 0;
  struct Named {   // Bump that also records what it hands out
    Bump& b;
    Plan& P;
    const char* tag = "";
    int i0 = 0, i1 = 0;
    size_t& off;
    size_t take(size_t n, const char* what = "") {
      const size_t o = b.take(n);
      if (P.want_regions) {
        if (P.nregs < 192) {
          Plan::Region& r = P.regs[P.nregs++];
          snprintf(r.name, sizeof(r.name), "%s[%d,%d].%s", tag, i0, i1, what);
          r.off = o, r.n = n;
        } else {
          P.regs_overflow = true;
        }
      }
      return o;
    }
  } B{B0, P, "", 0, 0, B0.off};
  const size_t N = d->N;
  size_t max_stat = 0, max_cnt = 0;
  // weight-gradient partials stay alive until the phase's single second-stage launch: one region per layer.  A scratch set
  // holds a decoder's layers or (set 0, after the main decoder's batch ran) the encoder's -> the larger of the two sums
  size_t wg_enc = 0, wg_dec = 0;
  P.bn_bytes = 0;
  auto wg_add = [](size_t& acc, size_t bytes) { acc += (bytes + 255) & ~(size_t)255; };
  size_t* wg_acc = &wg_enc;
  auto plan_blkws = [&](BlkWs& w, const BlockRef& k, int l) {
    const size_t e = N * k.c1.Co * P.H[l] * P.W[l];
    w.y1 = B.take(e, "y1"), w.y2 = B.take(e, "y2");
    w.st1 = B.take(4 * k.c1.Co, "st1"), w.st2 = B.take(4 * k.c1.Co, "st2");
    for (const ConvRef* c : {&k.c1, &k.c2}) {
      size_t nb = wsl_conv2d_stat_blocks(d->N, P.H[l], P.W[l], c->Ci, c->Co, 3);
      size_t wgb = wsl_conv2d_wgrad_ws_bytes(d->N, P.H[l], P.W[l], c->Ci, c->Co, 3);
      // BatchNorm-backward statistics arrive as one partial per tile of whichever kernel produces the gradient: the data gradient
      // of this layer's consumer (tile counts of the f32 and of the split plans, for either channel count) or the fan-in kernel
      size_t nbb = nb;
      for (int co_alt : {c->Ci, c->Co}) {
        const size_t a = wsl_conv2d_stat_blocks(d->N, P.H[l], P.W[l], c->Co, co_alt, 3), a1 = wsl_conv2d_stat_blocks(d->N, P.H[l], P.W[l], c->Co, co_alt, 1);
        nbb = a > nbb ? a : nbb, nbb = a1 > nbb ? a1 : nbb;
      }
      if (P.sp) {
        const size_t nb2 = wsl_sp_conv2d_stat_blocks(d->N, P.H[l], P.W[l], c->Ci, c->Co);
        const size_t wgb2 = wsl_sp_conv2d_wgrad_ws_bytes(d->N, P.H[l], P.W[l], c->Ci, c->Co);
        nb = nb2 > nb ? nb2 : nb, wgb = wgb2 > wgb ? wgb2 : wgb, nbb = nb2 > nbb ? nb2 : nbb;
      }
      if (nb * c->Co * 2 > max_stat) max_stat = nb * c->Co * 2;
      if (nb > max_cnt) max_cnt = nb;
      wg_add(*wg_acc, wgb);
      const size_t fan = wsl_feat_grad_combine_blocks(d->N, P.H[l], P.W[l]);
      nbb = fan > nbb ? fan : nbb;
      const size_t pb = sizeof(float) * (nbb * c->Co * 2 + 2 * (size_t)c->Co);
      if (pb > P.bn_bytes) P.bn_bytes = pb;
    }
    const size_t bb = wsl_bnact_bwd_ws_bytes(d->N, k.c1.Co, P.H[l], P.W[l]);
    if (bb > P.bn_bytes) P.bn_bytes = bb;
  };
  for (int l = 0; l < 5; ++l) {
    B.tag = "enc", B.i0 = l, B.i1 = 0;
    plan_blkws(P.wenc[l], P.enc[l], l);
    P.pooled[l] = l ? B.take(N * kFt[l - 1] * P.H[l] * P.W[l], "pooled") : 0;
  }
  for (int k = 0; k < d->n_dec; ++k) {
    wg_dec = 0, wg_acc = &wg_dec;
    for (int i = 0; i < 4; ++i) {
      const int l = 3 - i, c2 = kFt[l];
      B.tag = "dec", B.i0 = k, B.i1 = i;
      P.wdec[k].u[i] = B.take(N * c2 * P.H[l + 1] * P.W[l + 1], "u");
      P.wdec[k].up[i] = B.take(N * c2 * P.H[l] * P.W[l], "up");
      P.wdec[k].dcat[i] = B.take(N * 2 * c2 * P.H[l] * P.W[l], "dcat");
      plan_blkws(P.wdec[k].blk[i], P.dec[k].blk[i], l);
      wg_add(wg_dec, wsl_conv2d_wgrad_ws_bytes(d->N, P.H[l + 1], P.W[l + 1], kFt[l + 1], c2, 1));
    }
    P.wdec[k].glow4 = B.take(N * kFt[4] * P.H[4] * P.W[4], "glow4");
    wg_add(wg_dec, wsl_conv2d_wgrad_ws_bytes(d->N, P.H[0], P.W[0], kFt[0], d->n_class, 3));
  }
  P.wg_bytes = wg_enc > wg_dec ? wg_enc : wg_dec;
  const size_t big = N * kFt[0] * P.H[0] * P.W[0];  // largest activation (level 0; every deeper level is <= half)
  // stage 2 of the BatchNorm backward from a producer's partial sums: coefficients + (split path) one partial maximum per workgroup
  P.bn_coef_bytes = 0;
  for (int l = 0; l < 5; ++l) {
    const size_t b = wsl_bnact_bwd_finish_ws_bytes(d->N, kFt[l], P.H[l], P.W[l], 1);
    if (b > P.bn_coef_bytes) P.bn_coef_bytes = b;
  }
  if (P.sp)
    for (int l = 0; l < 4; ++l) {   // + the partial maxima of the upsampling passes (split path; forward only)
      const size_t b = wsl_bilinear_up2_fwd_amax_ws_bytes(d->N, kFt[l], P.H[l + 1], P.W[l + 1]);
      if (b > P.bn_coef_bytes) P.bn_coef_bytes = b;
    }
  for (int k = 0; k < d->n_dec; ++k) {
    Plan::Scratch& S = P.scr[k];
    B.tag = "scratch", B.i0 = k, B.i1 = 0;
    S.tmp_g = B.take(big, "tmp_g"), S.tmp_g1 = B.take(big, "tmp_g1"), S.tmp_dy = B.take(big, "tmp_dy");
    S.tmp_du = B.take(big / 4, "tmp_du"), S.tmp_gpool = B.take(big / 4, "tmp_gpool");
    S.stat_part = B.take(max_stat, "stat_part"), S.stat_cnt = B.take(max_cnt, "stat_cnt");
    S.wg_ws = B.take((P.wg_bytes + 3) / 4, "wg_ws"), S.bn_ws = B.take((P.bn_bytes + 3) / 4, "bn_ws");
    S.bn_coef = B.take((P.bn_coef_bytes + 3) / 4, "bn_coef");
  }
  if (d->n_dec == 1) P.scr[1] = P.scr[0];
  B.tag = "images", B.i0 = 0, B.i1 = 0;
  P.packf = B.take(P.n_param, "packf"), P.packd = B.take(P.n_param, "packd");  // packed [tap][ci][co] weight images (fwd / data-gradient)
  P.winof = B.take(2 * P.n_param, "winof"), P.winod = B.take(2 * P.n_param, "winod");
  P.spf = P.spd = P.sp_wmax = P.sp_dymax = P.sp_upmax = 0;
  if (P.sp) {
    // one max |w| slot per pack-table entry, WSL_SP_AMAX_SLOTS max |dy| slots per BatchNorm layer: sized from the plan
    P.spf = B.take(P.n_param * 10 / 9 + 64, "spf"), P.spd = B.take(P.n_param * 10 / 9 + 64, "spd");
    P.sp_wmax = B.take(kSpMaxLayers, "sp_wmax"), P.sp_dymax = B.take((size_t)P.n_bn * WSL_SP_AMAX_SLOTS, "sp_dymax");
    P.sp_upmax = B.take((size_t)d->n_dec * 4 * WSL_SP_AMAX_SLOTS, "sp_upmax");
  }
  P.total_floats = B.off;
  return WSL_OK;
}

// ------------------------------------------------------------------------------------------------ helpers
#define WSL_TRY(expr)           \
  do {                          \
    if (int rc_ = (expr)) return rc_; \
  } while (0)

struct Ctx {
  const Plan& P;
  const float* params;
  float* buffers;
  int64_t* nbt;
  float* grads;
  float* ws;
  void* stream;
  int training;
  int si = 0;   // scratch set
  struct WgBatch* wb = nullptr;   // weight gradients whose second stage is pending (one launch per phase)
  const Plan::Scratch& S() const { return P.scr[si]; }
};

// pending second stages of a phase (a decoder's backward, the encoder's backward) + the bump pointer into the set's wg_ws
struct WgBatch {
  WslWgradPending items[24];
  int n = 0;
  size_t off = 0;   // bytes
};

// weight gradient of one layer: stage 1 now (its partials get their own region of wg_ws), stage 2 with the phase's batch
static int wgrad_layer(const Ctx& c, const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, float* dw, float* db, int H,
                       int W, int Co, int ks, const uint32_t* dy_amax = nullptr, const uint32_t* raw_amax = nullptr) {
  const int N = c.P.d.N, Ci = a->C + (b ? b->C : 0);
  // (the split weight gradient is taken where it wins: 32-channel blocks on both sides.  On the 16-channel full-resolution layers
  // the f32 Winograd kernel is the faster one -- profiles/r3_sweep_layers_sp.md -- and just as much an fp32 result.)
  const bool sp = c.P.sp && dy_amax && ks == 3 && wsl_sp_conv2d_ok(a, b, nullptr, 0, N, H, W, Co, 3) &&
                  !(reinterpret_cast<uintptr_t>(dy) & 15) && !(dy_bs & 3) && Co % 32 == 0 && a->C % 32 == 0 && (!b || b->C % 32 == 0);
  const size_t need = ((sp ? wsl_sp_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co) : wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, ks)) + 255) &
                      ~(size_t)255;
  WgBatch* wb = c.wb;
  if (!wb || wb->n >= 24 || wb->off + need > c.P.wg_bytes) {
    set_error("net: weight-gradient batch overflow (%d pending, %zu + %zu of %zu bytes)", wb ? wb->n : -1, wb ? wb->off : 0, need,
              c.P.wg_bytes);
    return WSL_EWORKSPACE;
  }
  char* ws = reinterpret_cast<char*>(c.ws + c.S().wg_ws) + wb->off;
  if (sp) WSL_TRY(wsl_sp_conv2d_wgrad_partial_amax(a, b, dy, dy_bs, dy_amax, raw_amax, dw, db, N, H, W, Co, ws, need, &wb->items[wb->n], c.stream));
  else WSL_TRY(wsl_conv2d_wgrad_partial(a, b, dy, dy_bs, dw, db, N, H, W, Co, ks, ws, need, &wb->items[wb->n], c.stream));
  wb->n += 1, wb->off += need;
  return WSL_OK;
}
static int wgrad_flush(const Ctx& c) {
  WgBatch* wb = c.wb;
  if (!wb || wb->n == 0) return WSL_OK;
  const int rc = wsl_wgrad_reduce_batch(wb->items, wb->n, c.stream);
  wb->n = 0, wb->off = 0;
  return rc;
}

static WslSrc raw_src(const float* x, int C, int64_t bs) {
  WslSrc s{};
  s.x = x, s.C = C, s.bs = bs, s.emask_scale = 1.f;
  return s;
}
// virtual tensor leaky(bn(y)) [* emask] [* cmask]
static WslSrc act_src(const Ctx& c, size_t y, size_t st, int C, int HW, const uint8_t* emask, float es, const float* cmask) {
  WslSrc s{};
  s.x = c.ws + y, s.C = C, s.bs = (int64_t)C * HW;
  s.scale = c.ws + st + 2 * C, s.shift = c.ws + st + 3 * C;
  s.emask = emask, s.emask_scale = es, s.cmask = cmask;
  return s;
}

// split-precision path: float offset of a layer's weight image inside its arena (16-byte aligned, images never overlap: a 3x3
// layer's 9 Ci Co raw weights make room for its 10 Ci Co image floats); the layer's max |w| slot; a BatchNorm layer's max |dy| slot
static size_t sp_img_off(const ConvRef& cv) { return 4 * (size_t)((10 * cv.w + 35) / 36); }
static const uint32_t* sp_wmax(const Ctx& c, const ConvRef& cv) { return reinterpret_cast<const uint32_t*>(c.ws + c.P.sp_wmax) + cv.li; }
static uint32_t* sp_dymax(const Ctx& c, const BnRef& bn) {
  return c.P.sp ? reinterpret_cast<uint32_t*>(c.ws + c.P.sp_dymax) + (size_t)bn.nbt * WSL_SP_AMAX_SLOTS : nullptr;
}
static uint32_t* sp_upmax(const Ctx& c, int k, int i) {
  return c.P.sp ? reinterpret_cast<uint32_t*>(c.ws + c.P.sp_upmax) + (size_t)(k * 4 + i) * WSL_SP_AMAX_SLOTS : nullptr;
}
static bool sp_takes(const Ctx& c, const ConvRef& cv, const WslSrc* a, const WslSrc* b, const float* y, int64_t y_bs, int H, int W,
                     int Co) {
  return c.P.sp && cv.ks == 3 && wsl_sp_conv2d_ok(a, b, y, y_bs, c.P.d.N, H, W, Co, 3);
}

// forward (dgrad == 0) or data-gradient (dgrad == 1) convolution of layer `cv`: the packed fast path when the shapes
// are float4-aligned (every layer of a net whose H, W are multiples of 16), the generic kernel otherwise.
// (in_amax: the max |x| slot of a plain gradient source -- data-gradient launches of the split path)
static int conv_any(const Ctx& c, const ConvRef& cv, int dgrad, const WslSrc* a, const WslSrc* b, const float* bias,
                    float* y, int64_t y_bs, int H, int W, float* stp, float* stc, const uint32_t* in_amax = nullptr) {
  const int N = c.P.d.N, Co = dgrad ? cv.Ci : cv.Co;
  // (in_amax: a data-gradient launch's max |dy| slots, or a forward launch's max of its RAW source -- the decoder blocks' upsampled tensor)
  if ((!dgrad || in_amax) && sp_takes(c, cv, a, b, y, y_bs, H, W, Co))
    return wsl_sp_conv2d_fwd(a, b, c.ws + (dgrad ? c.P.spd : c.P.spf) + sp_img_off(cv), sp_wmax(c, cv), in_amax, bias,
                             y, y_bs, N, H, W, Co, stp, stc, c.stream);
  if (wsl_conv2d_fast_ok(a, b, y, y_bs, W)) {
    if (wsl_conv2d_wino_ok(N, H, W, a->C, b ? b->C : 0, Co, cv.ks))
      return wsl_conv2d_fwd(a, b, c.ws + (dgrad ? c.P.winod : c.P.winof) + 2 * cv.w, bias, y, y_bs, N, H, W, Co, cv.ks,
                            dgrad ? 5 : 4, stp, stc, c.stream);
    return wsl_conv2d_fwd(a, b, c.ws + (dgrad ? c.P.packd : c.P.packf) + cv.w, bias, y, y_bs, N, H, W, Co, cv.ks,
                          dgrad ? 3 : 2, stp, stc, c.stream);
  }
  return wsl_conv2d_fwd(a, b, c.params + cv.w, bias, y, y_bs, N, H, W, Co, cv.ks, dgrad, stp, stc, c.stream);
}

// Partial sums of a BatchNorm-backward statistics pass that the producer of g already emitted (into the scratch set's bn_ws)
struct GStats {
  int nblk = 0;            // 0: none, run the stand-alone reduction pass
  int channel_major = 0;
  int g_is_d = 0;          // the producer wrote d = g * keep * scale * leaky'(z) in place of g (wsl_conv2d_dgrad_bn_d, *fused == 2)
};

// data-gradient convolution of layer `cv` (dy -> g, dense) that also emits the backward statistics of the BatchNorm whose raw
// input is ws[y] (coefficients ws[st], dropout keep mask emask) when the kernel it dispatches to can (wsl_conv2d_dgrad_bn)
static int conv_dgrad_bn(const Ctx& c, const ConvRef& cv, const float* dy, float* g, int H, int W, size_t y, size_t st,
                         const uint8_t* emask, float es, GStats* gs, const uint32_t* dy_amax = nullptr) {
  const int N = c.P.d.N, Cg = cv.Ci;
  const WslSrc dys = raw_src(dy, cv.Co, (int64_t)cv.Co * H * W);
  const int64_t g_bs = (int64_t)Cg * H * W;
  if (dy_amax && sp_takes(c, cv, &dys, nullptr, g, g_bs, H, W, Cg)) {
    int fused = 0;
    WSL_TRY(wsl_sp_conv2d_dgrad_bn(&dys, dy_amax, c.ws + c.P.spd + sp_img_off(cv), sp_wmax(c, cv), g, g_bs, N, H, W, Cg, c.ws + y,
                                   c.ws + st, emask, es, c.ws + c.S().bn_ws, &fused, c.stream));
    gs->nblk = fused ? wsl_sp_conv2d_stat_blocks(N, H, W, cv.Co, Cg) : 0;
    gs->channel_major = 1;
    gs->g_is_d = 0;          // (the split kernels write the plain gradient)
    return WSL_OK;
  }
  const float* w = c.params + cv.w;
  int wmode = 1;
  if (wsl_conv2d_fast_ok(&dys, nullptr, g, g_bs, W)) {
    if (wsl_conv2d_wino_ok(N, H, W, cv.Co, 0, Cg, cv.ks)) w = c.ws + c.P.winod + 2 * cv.w, wmode = 5;
    else w = c.ws + c.P.packd + cv.w, wmode = 3;
  }
  int fused = 0;
  WSL_TRY(wsl_conv2d_dgrad_bn_d(&dys, w, g, g_bs, N, H, W, Cg, cv.ks, wmode, c.ws + y, c.ws + st, emask, es, c.ws + c.S().bn_ws,
                                &fused, c.stream));
  gs->nblk = fused ? wsl_conv2d_stat_blocks(N, H, W, cv.Co, Cg, cv.ks) : 0;
  gs->channel_major = 1;
  gs->g_is_d = fused == 2;
  return WSL_OK;
}

// BatchNorm + LeakyReLU (+ Dropout) backward of one conv output: from the producer's partial sums when it emitted them
static int bn_bwd(const Ctx& c, const float* g, int64_t g_bs, size_t y, size_t st, const BnRef& bn, const uint8_t* emask, float es,
                  float* dy, int H, int W, const GStats& gs) {
  const Plan& P = c.P;
  const int C = bn.C;
  const float* s = c.ws + st;
  // (split path: the apply pass also leaves max |dy| in the layer's slot -- the operand scale of dy's two consumers)
  if (gs.nblk && gs.g_is_d)   // g already is the gradient in front of the BatchNorm output: the apply pass reads it and y only
    return wsl_bnact_bwd_finish_d_amax(g, g_bs, c.ws + y, s, s + C, c.params + bn.gamma, c.params + bn.beta, dy, c.grads + bn.gamma,
                                       c.grads + bn.beta, P.d.N, C, H, W, c.ws + c.S().bn_ws, gs.nblk, gs.channel_major,
                                       c.ws + c.S().bn_coef, P.bn_coef_bytes, sp_dymax(c, bn), c.stream);
  if (gs.nblk)
    return wsl_bnact_bwd_finish_amax(g, g_bs, c.ws + y, s, s + C, c.params + bn.gamma, c.params + bn.beta, emask, es, dy,
                                     c.grads + bn.gamma, c.grads + bn.beta, P.d.N, C, H, W, c.ws + c.S().bn_ws, gs.nblk,
                                     gs.channel_major, c.ws + c.S().bn_coef, P.bn_coef_bytes, sp_dymax(c, bn), c.stream);
  return wsl_bnact_bwd_amax(g, g_bs, c.ws + y, s, s + C, c.params + bn.gamma, c.params + bn.beta, emask, es, dy, c.grads + bn.gamma,
                            c.grads + bn.beta, P.d.N, C, H, W, c.ws + c.S().bn_ws, P.bn_bytes, sp_dymax(c, bn), c.stream);
}

static int pack_all(const Ctx& c, int with_dgrad) {
  const Plan& P = c.P;
  PackTable t;
  t.n = 0;
  // `skip`: bit 0 / bit 1 = the forward / data-gradient launches of this layer take the Winograd kernels at its level (the very test
  // conv_any and conv_dgrad_bn apply), so nobody reads its direct [tap][ci][co] image: 3x3 layers are 99 % of the parameters, and
  // packing images that are never read was 41 us of every step (round 5).  The prediction and the launch apply the SAME test,
  // wsl_conv2d_wino_ok(), which in the product library is a pure function of the layer's shape (round 6: no routing state left -- ADVICE r5;
  // in the experiments build the tuning tools set their switches before the forward and leave them until the backward has run)
  auto one = [&](const ConvRef& cv, int l, int ca = 0, int cb = 0) {
    int skip = 0;
    if (cv.ks == 3) {
      const int N = P.d.N, H = P.H[l], W = P.W[l];
      if (wsl_conv2d_wino_ok(N, H, W, ca ? ca : cv.Ci, cb, cv.Co, 3)) skip |= 1;
      if (wsl_conv2d_wino_ok(N, H, W, cv.Co, 0, cv.Ci, 3)) skip |= 2;
    }
    t.e[t.n++] = PackEntry{cv.w, cv.Co, cv.Ci, cv.ks * cv.ks, skip};
  };
  for (int l = 0; l < 5; ++l) one(P.enc[l].c1, l), one(P.enc[l].c2, l);
  for (int k = 0; k < P.d.n_dec; ++k) {
    for (int i = 0; i < 4; ++i) {
      const int l = 3 - i;
      one(P.dec[k].c1x1[i], l + 1), one(P.dec[k].blk[i].c1, l, kFt[l], kFt[l]), one(P.dec[k].blk[i].c2, l);
    }
    one(P.dec[k].out, 0);
  }
  WSL_TRY(conv2_pack_table(t, c.params, c.ws + P.packf, c.ws + P.packd, with_dgrad, c.stream));
  WSL_TRY(wino_pack_table(t, c.params, c.ws + P.winof, c.ws + P.winod, with_dgrad, c.stream));
  if (P.sp) {   // f16 hi / lo images + the layers' max |w|; a training forward also clears the max |dy| slots of its backward
    WSL_REQUIRE(t.n <= kSpMaxLayers, "net: %d conv layers exceed the split path's table (%d)", t.n, kSpMaxLayers);
    int64_t off[kSpMaxLayers];
    for (int i = 0; i < t.n; ++i) off[i] = 4 * (int64_t)(4 * ((10 * t.e[i].w + 35) / 36));
    // two regions, two clears (their sizes come from the plan; nothing assumes they are neighbours)
    if (hipMemsetAsync(c.ws + P.sp_wmax, 0, kSpMaxLayers * sizeof(float), (hipStream_t)c.stream) != hipSuccess ||
        (with_dgrad && hipMemsetAsync(c.ws + P.sp_dymax, 0, (size_t)P.n_bn * WSL_SP_AMAX_SLOTS * sizeof(float), (hipStream_t)c.stream) != hipSuccess)) {
      set_error("net: clearing the split path's maxima failed");
      return WSL_EHIP;
    }
    WSL_TRY(sp_pack_table(t, off, c.params, c.ws + P.spf, c.ws + P.spd, reinterpret_cast<uint32_t*>(c.ws + P.sp_wmax), with_dgrad,
                          c.stream));
  }
  return WSL_OK;
}

// conv + (train: batch statistics -> BN coefficients | eval: running statistics)
static int conv_bn_fwd(const Ctx& c, const ConvRef& cv, const BnRef& bn, const WslSrc* a, const WslSrc* b, size_t y,
                       size_t st, int H, int W, const uint32_t* raw_amax = nullptr) {
  const Plan& P = c.P;
  const int N = P.d.N, C = cv.Co;
  float* stp = c.training ? c.ws + c.S().stat_part : nullptr;
  float* stc = c.training ? c.ws + c.S().stat_cnt : nullptr;
  WSL_TRY(conv_any(c, cv, 0, a, b, c.params + cv.b, c.ws + y, (int64_t)C * H * W, H, W, stp, stc, raw_amax));
  float* s = c.ws + st;
  if (c.training) {
    const int nblk = sp_takes(c, cv, a, b, c.ws + y, (int64_t)C * H * W, H, W, C) ? wsl_sp_conv2d_stat_blocks(N, H, W, cv.Ci, C)
                                                                                 : wsl_conv2d_stat_blocks(N, H, W, cv.Ci, C, cv.ks);
    return wsl_bn_stats_finalize(stp, stc, nblk, C, c.params + bn.gamma, c.params + bn.beta, kEps, kMom,
                                 c.buffers + bn.rmean, c.buffers + bn.rvar, c.nbt ? c.nbt + bn.nbt : nullptr, s, s + C,
                                 s + 2 * C, s + 3 * C, c.stream);
  }
  return wsl_bn_eval_affine(c.params + bn.gamma, c.params + bn.beta, c.buffers + bn.rmean, c.buffers + bn.rvar, kEps, C,
                            s + 2 * C, s + 3 * C, c.stream);
}

static int block_fwd(const Ctx& c, const BlockRef& k, const BlkWs& w, const WslSrc* a, const WslSrc* b, int l,
                     const uint8_t* emask, float es, const uint32_t* raw_amax = nullptr) {
  const int H = c.P.H[l], W = c.P.W[l], C = k.c1.Co;
  WSL_TRY(conv_bn_fwd(c, k.c1, k.b1, a, b, w.y1, w.st1, H, W, raw_amax));
  const WslSrc mid = act_src(c, w.y1, w.st1, C, H * W, c.training ? emask : nullptr, es, nullptr);
  return conv_bn_fwd(c, k.c2, k.b2, &mid, nullptr, w.y2, w.st2, H, W);
}

// backward of one ConvBlock given g = dL/d(block output) in `g` (batch stride g_bs; `gs`: the statistics of the second
// BatchNorm's backward when the producer of g emitted them).  Writes parameter grads; if dgrad_out != NULL also d(block input)
// (all Ci channels, dense).
static int block_bwd(const Ctx& c, const BlockRef& k, const BlkWs& w, const WslSrc* a, const WslSrc* b, int l,
                     const uint8_t* emask, float es, const float* g, int64_t g_bs, const GStats& gs, float* dgrad_out,
                     const uint32_t* raw_amax = nullptr) {
  const Plan& P = c.P;
  const int H = P.H[l], W = P.W[l], C = k.c1.Co;
  const int64_t CHW = (int64_t)C * H * W;
  float* dy = c.ws + c.S().tmp_dy;
  float* g1 = c.ws + c.S().tmp_g1;
  // BN2 + LeakyReLU (no dropout after the second activation)
  WSL_TRY(bn_bwd(c, g, g_bs, w.y2, w.st2, k.b2, nullptr, 1.f, dy, H, W, gs));
  const WslSrc mid = act_src(c, w.y1, w.st1, C, H * W, emask, es, nullptr);
  WSL_TRY(wgrad_layer(c, &mid, nullptr, dy, CHW, c.grads + k.c2.w, c.grads + k.c2.b, H, W, C, 3, sp_dymax(c, k.b2)));
  // data gradient of the second convolution; its epilogue carries the statistics of BN1 + LeakyReLU + Dropout(p)
  GStats g1s;
  WSL_TRY(conv_dgrad_bn(c, k.c2, dy, g1, H, W, w.y1, w.st1, emask, es, &g1s, sp_dymax(c, k.b2)));
  WSL_TRY(bn_bwd(c, g1, CHW, w.y1, w.st1, k.b1, emask, es, dy, H, W, g1s));
  WSL_TRY(wgrad_layer(c, a, b, dy, CHW, c.grads + k.c1.w, c.grads + k.c1.b, H, W, C, 3, sp_dymax(c, k.b1), raw_amax));
  if (dgrad_out) {
    const WslSrc dys1 = raw_src(dy, C, CHW);
    WSL_TRY(conv_any(c, k.c1, 1, &dys1, nullptr, nullptr, dgrad_out, (int64_t)k.c1.Ci * H * W, H, W, nullptr, nullptr,
                     sp_dymax(c, k.b1)));
  }
  return WSL_OK;
}

static WslSrc feat_src(const Ctx& c, int l, const float* cmask) {
  return act_src(c, c.P.wenc[l].y2, c.P.wenc[l].st2, kFt[l], c.P.H[l] * c.P.W[l], nullptr, 1.f, cmask);
}

static int decoder_fwd(const Ctx& c, int k, const float* const* cmasks, float* logits) {
  const Plan& P = c.P;
  const int N = P.d.N;
  for (int i = 0; i < 4; ++i) {
    const int l = 3 - i, c1 = kFt[l + 1], c2 = kFt[l];
    const int h = P.H[l + 1], w = P.W[l + 1], H = P.H[l], W = P.W[l];
    const WslSrc low = i == 0 ? feat_src(c, 4, cmasks ? cmasks[4] : nullptr)
                              : act_src(c, P.wdec[k].blk[i - 1].y2, P.wdec[k].blk[i - 1].st2, c1, h * w, nullptr, 1.f, nullptr);
    const ConvRef& cv = P.dec[k].c1x1[i];
    WSL_TRY(conv_any(c, cv, 0, &low, nullptr, c.params + cv.b, c.ws + P.wdec[k].u[i], (int64_t)c2 * h * w, h, w, nullptr,
                     nullptr));
    // (split path: the upsampling pass also leaves max |u| -- the bound of its raw output, the block's second source -- in the stage's slots;
    //  the partial maxima use the scratch set's BatchNorm-backward coefficient area, idle in the forward)
    WSL_TRY(wsl_bilinear_up2_fwd_amax(c.ws + P.wdec[k].u[i], c.ws + P.wdec[k].up[i], (int64_t)c2 * H * W, N, c2, h, w,
                                      c.ws + c.S().bn_coef, P.bn_coef_bytes, sp_upmax(c, k, i), c.stream));
    const WslSrc skip = feat_src(c, l, cmasks ? cmasks[l] : nullptr);
    const WslSrc up = raw_src(c.ws + P.wdec[k].up[i], c2, (int64_t)c2 * H * W);
    WSL_TRY(block_fwd(c, P.dec[k].blk[i], P.wdec[k].blk[i], &skip, &up, l, nullptr, 1.f, sp_upmax(c, k, i)));
  }
  const WslSrc last = act_src(c, P.wdec[k].blk[3].y2, P.wdec[k].blk[3].st2, kFt[0], P.H[0] * P.W[0], nullptr, 1.f, nullptr);
  const ConvRef& oc = P.dec[k].out;
  return conv_any(c, oc, 0, &last, nullptr, c.params + oc.b, logits, (int64_t)oc.Co * P.H[0] * P.W[0], P.H[0], P.W[0], nullptr,
                  nullptr);
}

static int decoder_bwd(const Ctx& c, int k, const float* const* cmasks, const float* dlogits) {
  const Plan& P = c.P;
  const int N = P.d.N, H0 = P.H[0], W0 = P.W[0];
  float* g = c.ws + c.S().tmp_g;
  const ConvRef& oc = P.dec[k].out;
  const WslSrc last = act_src(c, P.wdec[k].blk[3].y2, P.wdec[k].blk[3].st2, kFt[0], H0 * W0, nullptr, 1.f, nullptr);
  WSL_TRY(wgrad_layer(c, &last, nullptr, dlogits, (int64_t)oc.Co * H0 * W0, c.grads + oc.w, c.grads + oc.b, H0, W0, oc.Co, 3));
  // data gradient of the classifier = dL/d(output of up4's block): its epilogue carries that block's BN2 statistics
  GStats gs;
  WSL_TRY(conv_dgrad_bn(c, oc, dlogits, g, H0, W0, P.wdec[k].blk[3].y2, P.wdec[k].blk[3].st2, nullptr, 1.f, &gs));
  for (int i = 3; i >= 0; --i) {
    const int l = 3 - i, c1 = kFt[l + 1], c2 = kFt[l];
    const int h = P.H[l + 1], w = P.W[l + 1], H = P.H[l], W = P.W[l];
    const WslSrc skip = feat_src(c, l, cmasks ? cmasks[l] : nullptr);
    const WslSrc up = raw_src(c.ws + P.wdec[k].up[i], c2, (int64_t)c2 * H * W);
    float* dcat = c.ws + P.wdec[k].dcat[i];
    WSL_TRY(block_bwd(c, P.dec[k].blk[i], P.wdec[k].blk[i], &skip, &up, l, nullptr, 1.f, g, (int64_t)c2 * H * W, gs, dcat, sp_upmax(c, k, i)));
    // d(up) = dcat[:, c2:]  ->  d(u)  ->  conv1x1 backward
    float* du = c.ws + c.S().tmp_du;
    WSL_TRY(wsl_bilinear_up2_bwd(dcat + (int64_t)c2 * H * W, (int64_t)2 * c2 * H * W, du, N, c2, h, w, c.stream));
    const WslSrc low = i == 0 ? feat_src(c, 4, cmasks ? cmasks[4] : nullptr)
                              : act_src(c, P.wdec[k].blk[i - 1].y2, P.wdec[k].blk[i - 1].st2, c1, h * w, nullptr, 1.f, nullptr);
    const ConvRef& cv = P.dec[k].c1x1[i];
    WSL_TRY(wgrad_layer(c, &low, nullptr, du, (int64_t)c2 * h * w, c.grads + cv.w, c.grads + cv.b, h, w, c2, 1));
    if (i == 0) {   // gradient of the bottleneck feature: joins the other decoder's in the encoder's fan-in
      const WslSrc dus = raw_src(du, c2, (int64_t)c2 * h * w);
      WSL_TRY(conv_any(c, cv, 1, &dus, nullptr, nullptr, c.ws + P.wdec[k].glow4, (int64_t)c1 * h * w, h, w, nullptr, nullptr));
    } else {        // = dL/d(output of the previous stage's block): carries that block's BN2 statistics
      WSL_TRY(conv_dgrad_bn(c, cv, du, g, h, w, P.wdec[k].blk[i - 1].y2, P.wdec[k].blk[i - 1].st2, nullptr, 1.f, &gs));
    }
  }
  return wgrad_flush(c);   // second stage of this decoder's 13 weight gradients: one launch
}

static int encoder_bwd(const Ctx& c, const float* x, const uint8_t* const* emasks, const float* const* cmasks) {
  const Plan& P = c.P;
  const int N = P.d.N;
  const bool dual = P.d.n_dec == 2;
  float* g = c.ws + c.S().tmp_g;
  float* gpool = c.ws + c.S().tmp_gpool;
  for (int l = 4; l >= 0; --l) {
    const int C = kFt[l], H = P.H[l], W = P.W[l];
    const WslSrc f = feat_src(c, l, nullptr);
    const float *ga, *gb = nullptr;
    int64_t bs;
    if (l == 4) {
      ga = c.ws + P.wdec[0].glow4, bs = (int64_t)C * H * W;
      if (dual) gb = c.ws + P.wdec[1].glow4;
    } else {
      const int i = 3 - l;
      ga = c.ws + P.wdec[0].dcat[i], bs = (int64_t)2 * C * H * W;
      if (dual) gb = c.ws + P.wdec[1].dcat[i];
    }
    // fan-in of the feature's gradient + the statistics of its BatchNorm's backward in one pass
    const float* st2 = c.ws + P.wenc[l].st2;
    WSL_TRY(wsl_feat_grad_combine_bn(&f, ga, bs, gb, bs, dual ? cmasks[l] : nullptr, l < 4 ? gpool : nullptr, g, N, H, W, st2,
                                     st2 + C, c.ws + c.S().bn_ws, c.stream));
    GStats gs;
    gs.nblk = wsl_feat_grad_combine_blocks(N, H, W), gs.channel_major = 0;
    const WslSrc in = l == 0 ? raw_src(x, P.d.in_chns, (int64_t)P.d.in_chns * H * W)
                             : raw_src(c.ws + P.pooled[l], kFt[l - 1], (int64_t)kFt[l - 1] * H * W);
    WSL_TRY(block_bwd(c, P.enc[l], P.wenc[l], &in, nullptr, l, emasks[l], 1.f / (1.f - kDrop[l]), g, (int64_t)C * H * W, gs,
                      l > 0 ? gpool : nullptr));
  }
  return wgrad_flush(c);   // second stage of the encoder's 10 weight gradients: one launch
}

}  // namespace wsl

using namespace wsl;

static void entry_set(WslNetEntry* e, const char* name, int kind, int ndim, int64_t s0, int64_t s1, int64_t s2, int64_t s3,
                      int64_t off) {
  memset(e, 0, sizeof(*e));
  snprintf(e->name, sizeof(e->name), "%s", name);
  e->kind = kind, e->ndim = ndim, e->offset = off;
  e->shape[0] = s0, e->shape[1] = s1, e->shape[2] = s2, e->shape[3] = s3;
}

// state_dict order of the reference module (7 entries per conv+bn pair, see oracle/torch_ref.py:state_layout)
static int enumerate_entries(const Plan& P, int want, WslNetEntry* out) {
  int idx = 0;
  char nm[128];
  auto conv = [&](const char* pre, const char* sfx, const ConvRef& c) {
    snprintf(nm, sizeof(nm), "%s%s.weight", pre, sfx);
    if (idx++ == want) entry_set(out, nm, 0, 4, c.Co, c.Ci, c.ks, c.ks, c.w);
    snprintf(nm, sizeof(nm), "%s%s.bias", pre, sfx);
    if (idx++ == want) entry_set(out, nm, 0, 1, c.Co, 0, 0, 0, c.b);
  };
  auto bn = [&](const char* pre, const char* sfx, const BnRef& b) {
    const char* f[5] = {"weight", "bias", "running_mean", "running_var", "num_batches_tracked"};
    const int64_t off[5] = {b.gamma, b.beta, b.rmean, b.rvar, b.nbt};
    const int kind[5] = {0, 0, 1, 1, 2};
    for (int k = 0; k < 5; ++k) {
      snprintf(nm, sizeof(nm), "%s%s.%s", pre, sfx, f[k]);
      if (idx++ == want) entry_set(out, nm, kind[k], k == 4 ? 0 : 1, k == 4 ? 0 : b.C, 0, 0, 0, off[k]);
    }
  };
  auto block = [&](const char* pre, const BlockRef& k) {
    conv(pre, ".0", k.c1), bn(pre, ".1", k.b1), conv(pre, ".4", k.c2), bn(pre, ".5", k.b2);
  };
  char pre[96];
  block("encoder.in_conv.conv_conv", P.enc[0]);
  for (int l = 1; l < 5; ++l) {
    snprintf(pre, sizeof(pre), "encoder.down%d.maxpool_conv.1.conv_conv", l);
    block(pre, P.enc[l]);
  }
  for (int k = 0; k < P.d.n_dec; ++k) {
    const char* dn = P.d.n_dec == 1 ? "decoder" : (k == 0 ? "main_decoder" : "aux_decoder1");
    for (int i = 0; i < 4; ++i) {
      snprintf(pre, sizeof(pre), "%s.up%d.conv1x1", dn, i + 1);
      conv(pre, "", P.dec[k].c1x1[i]);
      snprintf(pre, sizeof(pre), "%s.up%d.conv.conv_conv", dn, i + 1);
      block(pre, P.dec[k].blk[i]);
    }
    snprintf(pre, sizeof(pre), "%s.out_conv", dn);
    conv(pre, "", P.dec[k].out);
  }
  return idx;
}

extern "C" int wsl_net_num_entries(const WslNetDesc* d) {
  Plan P;
  if (make_plan(d, P)) return -1;
  return enumerate_entries(P, -1, nullptr);
}
extern "C" int wsl_net_entry(const WslNetDesc* d, int i, WslNetEntry* out) {
  Plan P;
  WSL_TRY(make_plan(d, P));
  WSL_REQUIRE(out && i >= 0, "net_entry: bad args");
  const int n = enumerate_entries(P, i, out);
  WSL_REQUIRE(i < n, "net_entry: index %d out of %d", i, n);
  return WSL_OK;
}
extern "C" int64_t wsl_net_param_count(const WslNetDesc* d) {
  Plan P;
  return make_plan(d, P) ? -1 : P.n_param;
}
extern "C" int64_t wsl_net_encoder_param_count(const WslNetDesc* d) {
  Plan P;
  return make_plan(d, P) ? -1 : P.n_enc_param;
}
extern "C" int64_t wsl_net_buffer_count(const WslNetDesc* d) {
  Plan P;
  return make_plan(d, P) ? -1 : P.n_buf;
}
extern "C" size_t wsl_net_ws_bytes(const WslNetDesc* d) {
  Plan P;
  return make_plan(d, P) ? 0 : P.total_floats * sizeof(float);
}

extern "C" int wsl_net_forward(const WslNetDesc* d, const float* params, float* buffers, int64_t* nbt, const float* x,
                               const uint8_t* const* emasks, const float* const* cmasks, int training, float* logits_main,
                               float* logits_aux, void* ws, size_t ws_bytes, void* stream) {
  Plan P;
  WSL_TRY(make_plan(d, P));
  WSL_REQUIRE(params && buffers && x && logits_main && ws, "net_forward: null argument");
  WSL_REQUIRE(d->n_dec == 1 || (logits_aux && cmasks), "net_forward: unet_cct needs logits_aux and cmasks");
  WSL_REQUIRE(!training || emasks, "net_forward: training needs the 5 dropout keep-masks");
  if (ws_bytes < P.total_floats * sizeof(float)) {
    set_error("net_forward: workspace %zu < %zu", ws_bytes, P.total_floats * sizeof(float));
    return WSL_EWORKSPACE;
  }
  Ctx c{P, params, buffers, nbt, nullptr, static_cast<float*>(ws), stream, training};
  const int N = d->N;
  WSL_TRY(pack_all(c, training));   // packed weight images; training also builds the data-gradient ones for the backward
  for (int l = 0; l < 5; ++l) {
    const int H = P.H[l], W = P.W[l];
    WslSrc in;
    if (l == 0) {
      in = raw_src(x, d->in_chns, (int64_t)d->in_chns * H * W);
    } else {
      const WslSrc prev = feat_src(c, l - 1, nullptr);
      WSL_TRY(wsl_pool2_fwd(&prev, c.ws + P.pooled[l], N, P.H[l - 1], P.W[l - 1], stream));
      in = raw_src(c.ws + P.pooled[l], kFt[l - 1], (int64_t)kFt[l - 1] * H * W);
    }
    WSL_TRY(block_fwd(c, P.enc[l], P.wenc[l], &in, nullptr, l, training ? emasks[l] : nullptr, 1.f / (1.f - kDrop[l])));
  }
  // the two decoders only share read-only inputs: the auxiliary one runs on the library's side stream with its own
  // scratch set and is joined before returning (fills the other's launch gaps and workgroup tails)
  void* side = d->n_dec == 2 ? side_stream(stream) : nullptr;
  if (side) {
    Ctx c2 = c;
    c2.stream = side, c2.si = 1;
    WSL_TRY(stream_fork(stream, side));
    WSL_TRY(decoder_fwd(c, 0, nullptr, logits_main));
    WSL_TRY(decoder_fwd(c2, 1, cmasks, logits_aux));
    return stream_join(stream, side);
  }
  WSL_TRY(decoder_fwd(c, 0, nullptr, logits_main));
  if (d->n_dec == 2) WSL_TRY(decoder_fwd(c, 1, cmasks, logits_aux));
  return WSL_OK;
}

extern "C" int wsl_net_backward(const WslNetDesc* d, const float* params, const float* x, const uint8_t* const* emasks,
                                const float* const* cmasks, const float* dlogits_main, const float* dlogits_aux,
                                float* grads, void* ws, size_t ws_bytes, int phase, void* stream) {
  Plan P;
  WSL_TRY(make_plan(d, P));
  WSL_REQUIRE(params && x && emasks && grads && ws, "net_backward: null argument");
  WSL_REQUIRE(phase >= 0 && phase <= 2, "net_backward: phase %d", phase);
  WSL_REQUIRE(phase == 2 || (dlogits_main && (d->n_dec == 1 || (dlogits_aux && cmasks))), "net_backward: missing dlogits");
  if (ws_bytes < P.total_floats * sizeof(float)) {
    set_error("net_backward: workspace %zu < %zu", ws_bytes, P.total_floats * sizeof(float));
    return WSL_EWORKSPACE;
  }
  Ctx c{P, params, nullptr, nullptr, grads, static_cast<float*>(ws), stream, 1};
  WgBatch wb0, wb1;
  c.wb = &wb0;
  if (phase == 0 || phase == 1) {
    void* side = d->n_dec == 2 ? side_stream(stream) : nullptr;
    if (side) {
      Ctx c2 = c;
      c2.stream = side, c2.si = 1, c2.wb = &wb1;
      WSL_TRY(stream_fork(stream, side));
      WSL_TRY(decoder_bwd(c, 0, nullptr, dlogits_main));
      WSL_TRY(decoder_bwd(c2, 1, cmasks, dlogits_aux));
      WSL_TRY(stream_join(stream, side));
    } else {
      WSL_TRY(decoder_bwd(c, 0, nullptr, dlogits_main));
      if (d->n_dec == 2) WSL_TRY(decoder_bwd(c, 1, cmasks, dlogits_aux));
    }
  }
  if (phase == 0 || phase == 2) WSL_TRY(encoder_bwd(c, x, emasks, cmasks));
  return WSL_OK;
}

// ------------------------------------------------------------------------------------------------ test hook: the forward's decisions
namespace wsl {
__global__ __launch_bounds__(256) void dbg_sign_kernel(const float* y, const float* sc, const float* sh, unsigned char* out, int C, int HW) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float a = sc[c], b = sh[c];
  const int64_t base = ((int64_t)n * C + c) * HW;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) out[base + i] = fmaf(y[base + i], a, b) > 0.f ? 1 : 0;
}
__global__ __launch_bounds__(256) void dbg_argmax_kernel(const float* y, const float* sc, const float* sh, unsigned char* out, int C, int H,
                                                         int W) {
  const int c = blockIdx.y, n = blockIdx.z, Ho = H / 2, Wo = W / 2;
  const float a = sc[c], b = sh[c];
  const float* p = y + ((int64_t)n * C + c) * H * W;
  for (int o = blockIdx.x * kThreads + threadIdx.x; o < Ho * Wo; o += gridDim.x * kThreads) {
    const int oy = o / Wo, ox = o - oy * Wo;
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = leaky(fmaf(p[(int64_t)(2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)], a, b));
      if (k == 0 || v > best) best = v, arg = k;
    }
    out[((int64_t)n * C + c) * Ho * Wo + o] = (unsigned char)arg;
  }
}
}  // namespace wsl

extern "C" int wsl_debug_net_ws_region(const WslNetDesc* d, int index, char* name, size_t name_len, size_t* off_floats, size_t* n_floats) {
  Plan P;
  P.want_regions = true;
  WSL_TRY(make_plan(d, P));
  WSL_REQUIRE(!P.regs_overflow, "debug_net_ws_region: more than 192 workspace regions (raise Plan::regs)");
  if (index < 0 || index >= P.nregs) return 1;   // (past the end: not an error worth a message)
  WSL_REQUIRE(name && name_len > 0 && off_floats && n_floats, "debug_net_ws_region: null argument");
  snprintf(name, name_len, "%s", P.regs[index].name);
  *off_floats = P.regs[index].off, *n_floats = P.regs[index].n;
  return WSL_OK;
}

extern "C" int wsl_debug_net_decisions(const WslNetDesc* d, const void* ws, size_t ws_bytes, int which, int index, unsigned char* out,
                                       void* stream) {
  Plan P;
  WSL_TRY(make_plan(d, P));
  WSL_REQUIRE(ws && out && ws_bytes >= P.total_floats * sizeof(float), "debug_net_decisions: bad workspace");
  const float* w = static_cast<const float*>(ws);
  size_t y = 0, st = 0;
  int C = 0, l = 0;
  if (which == 0) {
    WSL_REQUIRE(index >= 0 && index < (int)P.n_bn, "debug_net_decisions: BatchNorm layer %d of %d", index, (int)P.n_bn);
    bool found = false;
    auto look = [&](const BlockRef& k, const BlkWs& bw, int lev) {
      if (k.b1.nbt == index) y = bw.y1, st = bw.st1, C = k.b1.C, l = lev, found = true;
      if (k.b2.nbt == index) y = bw.y2, st = bw.st2, C = k.b2.C, l = lev, found = true;
    };
    for (int lev = 0; lev < 5; ++lev) look(P.enc[lev], P.wenc[lev], lev);
    for (int k = 0; k < d->n_dec; ++k)
      for (int i = 0; i < 4; ++i) look(P.dec[k].blk[i], P.wdec[k].blk[i], 3 - i);
    WSL_REQUIRE(found, "debug_net_decisions: layer not found");
    const int HW = P.H[l] * P.W[l];
    WSL_LAUNCH(dbg_sign_kernel, dim3(cdiv(HW, 4 * kThreads), C, d->N), dim3(kThreads), 0, stream, w + y, w + st + 2 * C, w + st + 3 * C, out,
               C, HW);
    return check_launch("dbg_sign_kernel");
  }
  WSL_REQUIRE(which == 1 && index >= 1 && index <= 4, "debug_net_decisions: which %d index %d", which, index);
  l = index - 1, C = kFt[l];
  WSL_LAUNCH(dbg_argmax_kernel, dim3(cdiv(P.H[l] * P.W[l] / 4, 4 * kThreads), C, d->N), dim3(kThreads), 0, stream, w + P.wenc[l].y2,
             w + P.wenc[l].st2 + 2 * C, w + P.wenc[l].st2 + 3 * C, out, C, P.H[l], P.W[l]);
  return check_launch("dbg_argmax_kernel");
}

// ================================================================================================ UpBlock, transposed-conv branch
// UpBlock(in_channels1, in_channels2, out_channels, dropout_p, bilinear=False).forward(x1, x2)  (ref: networks/unet.py:47-68):
//     x1 = ConvTranspose2d(C1, C2, kernel_size=2, stride=2)(x1);  x = cat([x2, x1], 1);  return ConvBlock(2 C2, Co, p)(x)
// SURVEY 8f rank 4, opt-in: the reference's Decoder never passes bilinear=False, so this block is not part of UNet / UNet_CCT; it
// is exported as a module of its own with the reference's state_dict layout (up.weight [C1][C2][2][2], up.bias,
// conv.conv_conv.{0,1,4,5}.*).  Same ConvBlock machinery as the networks (block_fwd / block_bwd on a one-level plan).
namespace wsl {

struct UpPlan {
  Plan P;                  // one level: H[0] x W[0] = the output resolution
  ConvRef up;              // w: [C1][C2][2][2], b: [C2]
  BlockRef blk;
  BlkWs wblk;
  size_t upt, dcat, tmp_out;
  int C1, C2, Co, h, w;
};

static int make_up_plan(const WslUpBlockDesc* d, UpPlan& U) {
  WSL_REQUIRE(d && d->N > 0 && d->C1 > 0 && d->C1 <= 256 && d->C2 > 0 && d->C2 <= 256 && d->Co > 0 && d->h > 0 && d->w > 0,
              "upblock_t: bad descriptor (C1, C2 <= 256)");
  WSL_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "upblock_t: dropout_p %f", (double)d->dropout_p);
  Plan& P = U.P;
  memset(&P, 0, sizeof(P));
  P.d.in_chns = d->C1, P.d.n_class = 1, P.d.n_dec = 1, P.d.N = d->N, P.d.H = 2 * d->h, P.d.W = 2 * d->w;
  P.H[0] = 2 * d->h, P.W[0] = 2 * d->w;
  U.C1 = d->C1, U.C2 = d->C2, U.Co = d->Co, U.h = d->h, U.w = d->w;
  int64_t po = 0, bo = 0, nbn = 0;
  U.up.Ci = d->C1, U.up.Co = d->C2, U.up.ks = 2;
  U.up.w = po, po += (int64_t)d->C1 * d->C2 * 4;
  U.up.b = po, po += d->C2;
  plan_block(U.blk, 2 * d->C2, d->Co, po, bo, nbn);
  P.n_param = po, P.n_enc_param = 0, P.n_buf = bo, P.n_bn = nbn;
  Bump B;
  const size_t N = d->N, HW = (size_t)P.H[0] * P.W[0];
  const size_t e = N * d->Co * HW;
  U.wblk.y1 = B.take(e), U.wblk.y2 = B.take(e), U.wblk.st1 = B.take(4 * d->Co), U.wblk.st2 = B.take(4 * d->Co);
  U.upt = B.take(N * d->C2 * HW), U.dcat = B.take(N * 2 * d->C2 * HW), U.tmp_out = B.take(e);
  size_t max_stat = 0, max_cnt = 0, wg = 0;
  for (const ConvRef* cv : {&U.blk.c1, &U.blk.c2}) {
    const size_t nb = wsl_conv2d_stat_blocks(d->N, P.H[0], P.W[0], cv->Ci, cv->Co, 3);
    if (nb * cv->Co * 2 > max_stat) max_stat = nb * cv->Co * 2;
    if (nb > max_cnt) max_cnt = nb;
    wg += (wsl_conv2d_wgrad_ws_bytes(d->N, P.H[0], P.W[0], cv->Ci, cv->Co, 3) + 255) & ~(size_t)255;
  }
  const size_t ct = wsl_convt2x2_wgrad_ws_bytes(d->N, d->C1, d->C2);
  P.wg_bytes = wg > ct ? wg : ct;
  P.bn_bytes = wsl_bnact_bwd_ws_bytes(d->N, d->Co, P.H[0], P.W[0]);
  Plan::Scratch& S = P.scr[0];
  S.tmp_g = B.take(e), S.tmp_g1 = B.take(e), S.tmp_dy = B.take(e);
  S.tmp_du = 0, S.tmp_gpool = 0;
  S.stat_part = B.take(max_stat), S.stat_cnt = B.take(max_cnt);
  S.wg_ws = B.take((P.wg_bytes + 3) / 4), S.bn_ws = B.take((P.bn_bytes + 3) / 4), S.bn_coef = B.take(2 * (size_t)d->Co + 64);
  P.bn_coef_bytes = sizeof(float) * (2 * (size_t)d->Co + 64);
  P.scr[1] = P.scr[0];
  P.packf = B.take(P.n_param), P.packd = B.take(P.n_param);
  P.winof = B.take(2 * P.n_param), P.winod = B.take(2 * P.n_param);
  P.total_floats = B.off;
  return WSL_OK;
}

static int up_pack(const Ctx& c, const UpPlan& U, int with_dgrad) {
  PackTable t;
  t.n = 0;
  for (const ConvRef* cv : {&U.blk.c1, &U.blk.c2}) t.e[t.n++] = PackEntry{cv->w, cv->Co, cv->Ci, 9, 0};
  WSL_TRY(conv2_pack_table(t, c.params, c.ws + U.P.packf, c.ws + U.P.packd, with_dgrad, c.stream));
  return wino_pack_table(t, c.params, c.ws + U.P.winof, c.ws + U.P.winod, with_dgrad, c.stream);
}

}  // namespace wsl

extern "C" int64_t wsl_upblock_t_param_count(const WslUpBlockDesc* d) {
  UpPlan U;
  return make_up_plan(d, U) ? -1 : U.P.n_param;
}
extern "C" size_t wsl_upblock_t_ws_bytes(const WslUpBlockDesc* d) {
  UpPlan U;
  return make_up_plan(d, U) ? 0 : U.P.total_floats * sizeof(float);
}

// buffers: conv.conv_conv.1.running_mean | .running_var | conv.conv_conv.5.running_mean | .running_var (4 * Co floats);
// nbt: the two num_batches_tracked counters.  emask: keep mask of the ConvBlock's nn.Dropout(p) [N][Co][2h][2w] (training, p > 0).
extern "C" int wsl_upblock_t_forward(const WslUpBlockDesc* d, const float* params, float* buffers, int64_t* nbt, const float* x1,
                                     const float* x2, const uint8_t* emask, int training, float* out, void* ws, size_t ws_bytes,
                                     void* stream) {
  UpPlan U;
  WSL_TRY(make_up_plan(d, U));
  WSL_REQUIRE(params && buffers && x1 && x2 && out && ws, "upblock_t_forward: null argument");
  WSL_REQUIRE(!(training && d->dropout_p > 0.f) || emask, "upblock_t_forward: training with dropout_p > 0 needs the keep mask");
  if (ws_bytes < U.P.total_floats * sizeof(float)) {
    set_error("upblock_t_forward: workspace %zu < %zu", ws_bytes, U.P.total_floats * sizeof(float));
    return WSL_EWORKSPACE;
  }
  const Plan& P = U.P;
  Ctx c{P, params, buffers, nbt, nullptr, static_cast<float*>(ws), stream, training};
  const int H = P.H[0], W = P.W[0];
  WSL_TRY(up_pack(c, U, training));
  WSL_TRY(wsl_convt2x2_fwd(x1, params + U.up.w, params + U.up.b, c.ws + U.upt, d->N, U.C1, U.C2, U.h, U.w, stream));
  const WslSrc skip = raw_src(x2, U.C2, (int64_t)U.C2 * H * W);
  const WslSrc up = raw_src(c.ws + U.upt, U.C2, (int64_t)U.C2 * H * W);
  const float es = d->dropout_p > 0.f ? 1.f / (1.f - d->dropout_p) : 1.f;
  WSL_TRY(block_fwd(c, U.blk, U.wblk, &skip, &up, 0, d->dropout_p > 0.f ? emask : nullptr, es));
  const WslSrc act = act_src(c, U.wblk.y2, U.wblk.st2, U.Co, H * W, nullptr, 1.f, nullptr);
  return wsl_src_materialize(&act, out, (int64_t)U.Co * H * W, d->N, H, W, stream);
}

// grads: parameter-arena layout; dx1 [N][C1][h][w], dx2 [N][C2][2h][2w] (either may be NULL)
extern "C" int wsl_upblock_t_backward(const WslUpBlockDesc* d, const float* params, const float* x1, const float* x2,
                                      const uint8_t* emask, const float* dout, float* grads, float* dx1, float* dx2, void* ws,
                                      size_t ws_bytes, void* stream) {
  UpPlan U;
  WSL_TRY(make_up_plan(d, U));
  WSL_REQUIRE(params && x1 && x2 && dout && grads && ws, "upblock_t_backward: null argument");
  if (ws_bytes < U.P.total_floats * sizeof(float)) {
    set_error("upblock_t_backward: workspace %zu < %zu", ws_bytes, U.P.total_floats * sizeof(float));
    return WSL_EWORKSPACE;
  }
  const Plan& P = U.P;
  Ctx c{P, params, nullptr, nullptr, grads, static_cast<float*>(ws), stream, 1};
  WgBatch wb;
  c.wb = &wb;
  const int H = P.H[0], W = P.W[0];
  const int64_t HW = (int64_t)H * W;
  const WslSrc skip = raw_src(x2, U.C2, U.C2 * HW);
  const WslSrc up = raw_src(c.ws + U.upt, U.C2, U.C2 * HW);
  const float es = d->dropout_p > 0.f ? 1.f / (1.f - d->dropout_p) : 1.f;
  float* dcat = c.ws + U.dcat;
  WSL_TRY(block_bwd(c, U.blk, U.wblk, &skip, &up, 0, d->dropout_p > 0.f ? emask : nullptr, es, dout, U.Co * HW, GStats{}, dcat));
  WSL_TRY(wgrad_flush(c));
  const float* dup = dcat + U.C2 * HW;       // d(cat)[:, C2:] = gradient of the transposed convolution's output
  WSL_TRY(wsl_convt2x2_wgrad(x1, dup, 2 * U.C2 * HW, grads + U.up.w, grads + U.up.b, d->N, U.C1, U.C2, U.h, U.w,
                             c.ws + c.S().wg_ws, P.wg_bytes, stream));
  if (dx1) WSL_TRY(wsl_convt2x2_dgrad(dup, 2 * U.C2 * HW, params + U.up.w, dx1, d->N, U.C1, U.C2, U.h, U.w, stream));
  if (dx2) {
    const WslSrc dskip = raw_src(dcat, U.C2, 2 * U.C2 * HW);
    WSL_TRY(wsl_src_materialize(&dskip, dx2, U.C2 * HW, d->N, H, W, stream));
  }
  return WSL_OK;
}
