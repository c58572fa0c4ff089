// v2 of the MFMA convolution kernels: same implicit-GEMM mapping and LDS operand layout as wsl_conv.hip, with the
// HBM -> LDS staging rebuilt for throughput (profiles/r1a: staging, not MFMA issue, bounded v1):
//   * every global access is an aligned 16-byte load: the halo tile is staged with a row pitch of TW+8 (4 columns of
//     padding on each side instead of 1) so rows start on a float4 boundary; keep-masks travel as uchar4;
//   * each thread owns ONE float4 position of the tile and walks the channels of a chunk: all index arithmetic and the
//     bounds test are hoisted out of the loops, the loads of a chunk are independent and issued back to back;
//   * register prefetch: the loads of chunk k+1 (tile t+1 in the weight-gradient) are issued before the MFMA loop of
//     chunk k and written to LDS after it (one LDS buffer, two barriers per chunk);
//   * weights are read from a packed [tap][ci][co] image (wsl_conv2d_pack_weights), so a chunk's B operands are
//     contiguous rows copied with float4 loads instead of a 4-byte gather;
//   * the producer's BN scale/shift live in LDS for the whole workgroup.
// Requires W % 4 == 0 and 16-byte aligned tensors / batch strides; anything else takes the v1 kernels.
#include <stdlib.h>

#include "wsl_rt.h"

namespace wsl {

struct Src2 {            // one source, device view
  const float* x;
  const uint8_t* emask;
  const float* scale;
  const float* shift;
  const float* cmask;
  int64_t bs;
  int C;
  float es;
};

struct Conv2P {
  Src2 a, b;
  const float* wp;       // packed [KK][Ci][Co]
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, H, W, Ci, Co, tiles_x, tiles_y;
  float* stat_part;
  float* stat_cnt;
  int slots;    // BatchNorm partial slots per tile (1; 4 when the wave-specialised variant is active)
  int ablate;   // debug (env WSL_CONV_ABLATE): 1 skip MFMA, 2 skip staging after the first chunk, 4 skip epilogue
  BnBwdEpi bn;  // data-gradient launches: BatchNorm-backward statistics of the consumer of y (wsl_rt.h)
};

template <int KS, int TH, int TW, int CO_T, int KC>
struct Conv2Cfg {
  static constexpr int P = KS / 2, KK = KS * KS, PADL = P ? 4 : 0;
  static constexpr int ROWP = TW + 2 * PADL, ROWS = TH + 2 * P, ROWP4 = ROWP / 4, POS = ROWS * ROWP4;
  static constexpr int G = 256 / POS, NLD = (KC + G - 1) / G;
  static constexpr int PLANE = ((ROWS * ROWP - 16 + 31) / 32) * 32 + 16;  // == 16 (mod 32)
  static constexpr int CSTR = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;
  static constexpr int SEGS = TW / 16, MT_TOTAL = TH * SEGS, MT = MT_TOTAL / 4, NT = CO_T / 16;
  static constexpr int IN_FLOATS = KC * PLANE, W_FLOATS = KK * KC * CSTR;
  static constexpr int WQ = CO_T / 4, WF4 = KK * KC * WQ, NWL = (WF4 + 255) / 256;
  static constexpr int MAXC = 512;  // channels whose BN coefficients fit the LDS table
  static constexpr size_t SMEM = sizeof(float) * (IN_FLOATS + W_FLOATS + 3 * MAXC);
  // workgroups per CU the register allocator must leave room for (2nd __launch_bounds__ argument = waves per SIMD)
  static constexpr int MINW = (MT * NT * 4 <= 32) ? 3 : 2;
  static constexpr int MINWL = (MT * NT * 4 <= 64) ? 3 : 2;   // lean kernel: its staging state is smaller
  static_assert(POS <= 256 && G >= 1, "one float4 position per thread");
  static_assert(MT_TOTAL % 4 == 0 && KC % 4 == 0 && (8 * CO_T) <= IN_FLOATS, "tile shape");
};

template <typename C, int KS, int KC, int NG>
__device__ __forceinline__ void conv2_mfma_stages(const float* in_t, const float* w_t, const int (&abase)[C::MT], int bbase,
                                                   v4f (&acc)[C::MT][C::NT]) {
  constexpr int NS = KS * KS * NG;   // stages
  float av[2][C::MT], bv[2][C::NT];
  auto load = [&](int s, int buf) {
    const int tap = s / NG, cg = s % NG, ky = tap / KS, kx = tap % KS;
#pragma unroll
    for (int j = 0; j < C::NT; ++j) bv[buf][j] = w_t[(tap * KC + cg * 4) * C::CSTR + j * 16 + bbase];
#pragma unroll
    for (int i = 0; i < C::MT; ++i) av[buf][i] = in_t[cg * 4 * C::PLANE + ky * C::ROWP + kx + abase[i]];
  };
  load(0, 0);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + 1 < NS) load(s + 1, (s + 1) & 1);
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
      for (int j = 0; j < C::NT; ++j) acc[i][j] = WSL_MFMA16(av[s & 1][i], bv[s & 1][j], acc[i][j]);
    WSL_SCHED_BARRIER();
  }
}

template <int KS, int TH, int TW, int CO_T, int KC>
__global__ __launch_bounds__(256, (Conv2Cfg<KS, TH, TW, CO_T, KC>::MINW)) void conv_mfma2_kernel(Conv2P p) {
  using C = Conv2Cfg<KS, TH, TW, CO_T, KC>;
  WSL_DYN_SMEM(smem);
  float* in_t = reinterpret_cast<float*>(smem);
  float* w_t = in_t + C::IN_FLOATS;
  float* sc_l = w_t + C::W_FLOATS;   // [Ci] scale (1 when the source is raw)
  float* sh_l = sc_l + C::MAXC;      // [Ci] shift
  float* cm_l = sh_l + C::MAXC;      // [Ci] channel multiplier of this sample (1 if none)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order (speed only): workgroup b is observed to run on XCD b % 8, each XCD has its own L2.  Give
  // every XCD a contiguous run of spatial tiles so vertically adjacent tiles share their halo rows in one L2.
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int tile_id = bid;     // index of the per-block BatchNorm partials
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = blockIdx.y * CO_T;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = p.H, W = p.W, Ci = p.Ci;
  const int64_t HW = (int64_t)H * W;

  for (int c = tid; c < Ci; c += kThreads) {
    const bool ina = c < p.a.C;
    const Src2& s = ina ? p.a : p.b;
    const int ch = ina ? c : c - p.a.C;
    sc_l[c] = s.scale ? s.scale[ch] : 1.f;
    sh_l[c] = s.scale ? s.shift[ch] : 0.f;
    cm_l[c] = s.cmask ? s.cmask[(int64_t)n * s.C + ch] : 1.f;
  }

  // ---- this thread's staging position (fixed for the whole kernel)
  const int grp = tid / C::POS, pos = tid - grp * C::POS;
  const int pty = pos / C::ROWP4, ptx4 = pos - pty * C::ROWP4;
  const int gy = y0 + pty - C::P, gx = x0 + ptx4 * 4 - C::PADL;
  const bool pvalid = grp < C::G && gy >= 0 && gy < H && gx >= 0 && gx < W;
  const int64_t goff = (int64_t)gy * W + gx;
  const int loff = pty * C::ROWP + ptx4 * 4;

  v4f acc[C::MT][C::NT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i)
#pragma unroll
    for (int j = 0; j < C::NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  int abase[C::MT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i) {
    const int mt = wave * C::MT + i;
    abase[i] = (lane >> 4) * C::PLANE + (mt / C::SEGS) * C::ROWP + (mt % C::SEGS) * 16 + (lane & 15) + (C::PADL - C::P);
  }
  const int bbase = (lane >> 4) * C::CSTR + (lane & 15);

  float4 pre[C::NLD];
  uchar4 prm[C::NLD];
  float4 prw[C::NWL];
  const bool co_vec = (p.Co & 3) == 0;

  // Addresses = workgroup-uniform channel base (scalar registers) + ONE per-thread 32-bit element offset: a 64-bit
  // address pair per in-flight load made the allocator spill, and a spill reload's vmcnt(0) drains the prefetch.
  const uint32_t toff = (uint32_t)(grp * (int)HW + (int)goff);   // valid threads only; < 2^31 elements per sample
  auto issue = [&](int c0) {
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
      const int cb = c0 + i * C::G;            // uniform: first channel of this load instruction
      const int cg = cb + grp;
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      prm[i] = make_uchar4(1, 1, 1, 1);
      if (cb < Ci) {                           // uniform branch
        const bool ina = cb < p.a.C;           // uniform (eligibility: a.C % G == 0 when two sources are present)
        const Src2& s = ina ? p.a : p.b;
        const int chb = ina ? cb : cb - p.a.C;
        const float* xb = s.x + n * s.bs + (int64_t)chb * HW;
        const uint8_t* mb = s.emask ? s.emask + ((int64_t)n * s.C + chb) * HW : nullptr;
        if (pvalid && i * C::G + grp < KC && cg < Ci) {
          pre[i] = *reinterpret_cast<const float4*>(xb + toff);
          if (mb) prm[i] = *reinterpret_cast<const uchar4*>(mb + toff);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) {
      const int f = tid + i * kThreads;
      prw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < C::WF4) {
        const int row = f / C::WQ, q = f - row * C::WQ;
        const int tap = row / KC, c = row - tap * KC;
        const int cg = c0 + c, cog = co0 + q * 4;
        if (cg < Ci && cog < p.Co) {
          const float* src = p.wp + ((int64_t)tap * Ci + cg) * p.Co + cog;
          if (co_vec) {
            prw[i] = *reinterpret_cast<const float4*>(src);
          } else {
            prw[i].x = src[0];
            if (cog + 1 < p.Co) prw[i].y = src[1];
            if (cog + 2 < p.Co) prw[i].z = src[2];
            if (cog + 3 < p.Co) prw[i].w = src[3];
          }
        }
      }
    }
  };

  auto commit = [&](int c0) {
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
      const int c = grp + i * C::G, cg = c0 + c;
      if (grp < C::G && c < KC) {
        float4 v = pre[i];
        if (pvalid && cg < Ci) {
          const bool ina = cg < p.a.C;
          const Src2& s = ina ? p.a : p.b;
          if (s.scale) {
            const float sc = sc_l[cg], sh = sh_l[cg];
            v.x = leaky(fmaf(v.x, sc, sh)), v.y = leaky(fmaf(v.y, sc, sh));
            v.z = leaky(fmaf(v.z, sc, sh)), v.w = leaky(fmaf(v.w, sc, sh));
          }
          if (s.emask) {
            const uchar4 m = prm[i];
            v.x = m.x ? v.x * s.es : 0.f, v.y = m.y ? v.y * s.es : 0.f;
            v.z = m.z ? v.z * s.es : 0.f, v.w = m.w ? v.w * s.es : 0.f;
          }
          if (s.cmask) {
            const float cm = cm_l[cg];
            v.x *= cm, v.y *= cm, v.z *= cm, v.w *= cm;
          }
        }
        *reinterpret_cast<float4*>(in_t + c * C::PLANE + loff) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) {
      const int f = tid + i * kThreads;
      if (f < C::WF4) {
        const int row = f / C::WQ, q = f - row * C::WQ;
        *reinterpret_cast<float4*>(w_t + row * C::CSTR + q * 4) = prw[i];
      }
    }
  };

  issue(0);
  __syncthreads();  // BN tables visible
  for (int c0 = 0; c0 < Ci; c0 += KC) {
    if (!WSL_ABLATED(p, 2) || c0 == 0) commit(c0);
    __syncthreads();
    if (c0 + KC < Ci && !WSL_ABLATED(p, 2)) issue(c0 + KC);  // prefetch: in flight during the MFMA loop below
    // MFMA loop.  Stages = (tap, channel group); the A/B operands of stage s+1 are read from LDS before the MFMAs of
    // stage s are issued (explicit one-stage software pipeline: profiles/r1b showed ds_read -> waitcnt -> mfma chains).
    // Channels past Ci were staged as zeros, so a short last chunk needs no branch here; a chunk holding <= 4 channels
    // (Ci = 1 or 4 layers) runs the single-group variant.
    if (!WSL_ABLATED(p, 1)) {
      if (Ci - c0 > 4) conv2_mfma_stages<C, KS, KC, KC / 4>(in_t, w_t, abase, bbase, acc);
      else conv2_mfma_stages<C, KS, KC, 1>(in_t, w_t, abase, bbase, acc);
    }
    __syncthreads();
  }

  // ---- epilogue (identical to v1): bias, float4 stores, BatchNorm partial statistics
  if (WSL_ABLATED(p, 4)) {
    if (acc[0][0][0] == 123.456f) p.y[0] = 1.f;   // keep the accumulators live
    return;
  }
  float bsum[C::NT];
#pragma unroll
  for (int j = 0; j < C::NT; ++j) {
    const int co = co0 + j * 16 + (lane & 15);
    const float bias = (p.bias && co < p.Co) ? p.bias[co] : 0.f;
    bsum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < C::MT; ++i) {
      const int mt = wave * C::MT + i;
      const int oy = y0 + mt / C::SEGS, ox = x0 + (mt % C::SEGS) * 16 + (lane >> 4) * 4;
      v4f v = acc[i][j];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bias;
      acc[i][j] = v;
      if (co < p.Co && oy < H && ox < W) {  // W % 4 == 0: a float4 is inside or outside as a whole
        *reinterpret_cast<float4*>(p.y + n * p.y_bs + co * HW + (int64_t)oy * W + ox) = make_float4(v[0], v[1], v[2], v[3]);
        bsum[j] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
  }
  if (p.bn.part) {   // BatchNorm-backward statistics of the layer that consumes this gradient (dense y, same [N][Co][H][W] shape)
    float s1[C::NT], s2[C::NT];
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int co = co0 + j * 16 + (lane & 15);
      BnBwdAcc ba;
      if (co < p.Co) {
        const float mean = p.bn.st[co], invstd = p.bn.st[p.Co + co], sc = p.bn.st[2 * p.Co + co], sh = p.bn.st[3 * p.Co + co];
#pragma unroll
        for (int i = 0; i < C::MT; ++i) {
          const int mt = wave * C::MT + i;
          const int oy = y0 + mt / C::SEGS, ox = x0 + (mt % C::SEGS) * 16 + (lane >> 4) * 4;
          if (oy < H && ox < W)
            bn_bwd_acc4(p.bn, ((int64_t)n * p.Co + co) * HW + (int64_t)oy * W + ox, acc[i][j][0], acc[i][j][1], acc[i][j][2],
                        acc[i][j][3], mean, invstd, sc, sh, ba);
        }
      }
      bn_bwd_fold(ba, s1[j], s2[j]);
    }
    bn_bwd_store<C::NT, CO_T>(p.bn, s1, s2, in_t, co0, p.Co, tile_id, (int)gridDim.x);
    return;
  }
  if (p.stat_part) {
    float* red1 = in_t;
    float* red2 = in_t + 4 * CO_T;
    const int vh = (H - y0 < TH) ? H - y0 : TH, vw = (W - x0 < TW) ? W - x0 : TW;
    const float cnt = (float)(vh * vw);
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      float s = bsum[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) red1[wave * CO_T + j * 16 + lane] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int col = j * 16 + (lane & 15);
      const int co = co0 + col;
      const float mean_b = (red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col]) / cnt;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < C::MT; ++i) {
        const int mt = wave * C::MT + i;
        const int oy = y0 + mt / C::SEGS, ox = x0 + (mt % C::SEGS) * 16 + (lane >> 4) * 4;
        if (co < p.Co && oy < H && ox < W) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc[i][j][r] - mean_b;
            q = fmaf(d, d, q);
          }
        }
      }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (lane < 16) red2[wave * CO_T + j * 16 + lane] = q;
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        const int col = j * 16 + lane, co = co0 + col;
        if (co < p.Co) {
          const int64_t nsl = (int64_t)gridDim.x * p.slots;                                 // [Co][slots][2]
          float* dst = p.stat_part + ((int64_t)co * nsl + (int64_t)tile_id * p.slots) * 2;   // slot 0 of this tile's slots
          dst[0] = red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col];
          dst[1] = red2[col] + red2[CO_T + col] + red2[2 * CO_T + col] + red2[3 * CO_T + col];
        }
      }
      if (lane < p.slots && blockIdx.y == 0) p.stat_cnt[tile_id * p.slots + lane] = lane == 0 ? cnt : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ lean variant
// Same tiling, LDS layout and MFMA stages as conv_mfma2_kernel, for the shapes every layer of the training step has:
// Ci % KC == 0 (a chunk never straddles the two sources), Co % CO_T == 0, H % TH == 0 and W % TW == 0.  The per-workgroup
// timeline (WSL_CONV_ABLATE=128, profiles/r1g_conv_timeline.md) showed the generic kernel's waves spending more cycles
// ISSUING the staging code (scalar 64-bit address chains, per-element source selects, bounds tests) than in the MFMA
// stream, with memory latency fully hidden.  Here the staging is cut to the minimum instruction count:
//   * one workgroup-uniform source select and base pointer per chunk, per-load addresses = scalar base + i * stride
//     + ONE per-thread 32-bit offset; threads outside the image load a valid dummy address and never write LDS (their
//     LDS slots are zeroed once), so no load sits under a divergent branch;
//   * BN scale/shift as one LDS float2 per channel, LeakyReLU as max(z, slope z), keep-mask bytes (0/1) applied as a
//     float factor; the whole transform under a single exec region;
//   * epilogue without bounds tests, store offsets folded into immediates / scalar adds.
template <int KS, int TH, int TW, int CO_T, int KC>
__global__ __launch_bounds__(256, (Conv2Cfg<KS, TH, TW, CO_T, KC>::MINWL)) void conv_mfma2l_kernel(Conv2P p) {
  using C = Conv2Cfg<KS, TH, TW, CO_T, KC>;
  static_assert(KC % C::G == 0 && C::MT % C::SEGS == 0, "lean staging shape");
  WSL_DYN_SMEM(smem);
  float* in_t = reinterpret_cast<float*>(smem);
  float* w_t = in_t + C::IN_FLOATS;
  float2* tab = reinterpret_cast<float2*>(w_t + C::W_FLOATS);   // [Ci] {scale, shift}  (2 * MAXC floats)
  float* cm_l = w_t + C::W_FLOATS + 2 * C::MAXC;                // [Ci] channel multiplier of this sample
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);   // XCD-aware tile order, as conv_mfma2_kernel
  const int tile_id = bid;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = blockIdx.y * CO_T;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = p.H, W = p.W, Ci = p.Ci, Co = p.Co;
  const int HW = H * W;

  // ---- this thread's staging position (fixed for the whole kernel)
  const int grp = tid / C::POS, pos = tid - grp * C::POS;
  const int pty = pos / C::ROWP4, ptx4 = pos - pty * C::ROWP4;
  const int gy = y0 + pty - C::P, gx = x0 + ptx4 * 4 - C::PADL;
  const bool owner = grp < C::G;                                   // owns LDS slots (pos, grp + i * G)
  const bool pvalid = owner && gy >= 0 && gy < H && gx >= 0 && gx < W;
  const uint32_t toff = pvalid ? (uint32_t)(grp * HW + gy * W + gx) : 0u;   // element offset inside a chunk's channels
  const int loff = grp * C::PLANE + pty * C::ROWP + ptx4 * 4;
  const int64_t gstride = (int64_t)C::G * HW;                      // elements between consecutive loads of a thread
  const float* xa_n = p.a.x + n * p.a.bs;
  const float* xb_n = p.b.C ? p.b.x + n * p.b.bs : nullptr;
  const uint8_t* ma_n = p.a.emask ? p.a.emask + (int64_t)n * p.a.C * HW : nullptr;
  const uint8_t* mb_n = (p.b.C && p.b.emask) ? p.b.emask + (int64_t)n * p.b.C * HW : nullptr;

  // weights of a chunk: row = (tap, c) of the packed image, this thread copies float4 #(tid + i * 256)
  uint32_t woff[C::NWL];
  int wl[C::NWL];
#pragma unroll
  for (int i = 0; i < C::NWL; ++i) {
    const int f = tid + i * kThreads;
    const int row = f / C::WQ, q = f - row * C::WQ;
    const int tap = row / KC, c = row - tap * KC;
    woff[i] = f < C::WF4 ? (uint32_t)((tap * Ci + c) * Co + q * 4) : 0u;   // threads past the image copy nothing
    wl[i] = row * C::CSTR + q * 4;
  }
  const float* w_n = p.wp + co0;

  float4 pre[C::NLD];
  uint32_t prm[C::NLD];
  v4f prw[C::NWL];   // a native vector: a float4 struct copied global -> private -> LDS stays a memcpy pair in scratch

  auto issue = [&](int c0) __attribute__((always_inline)) {
    const bool ina = c0 < p.a.C;                                   // uniform
    const int chb = ina ? c0 : c0 - p.a.C;
    const float* xb = (ina ? xa_n : xb_n) + (int64_t)chb * HW;
    const uint8_t* mb = ina ? ma_n : mb_n;
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) pre[i] = *reinterpret_cast<const float4*>(xb + i * gstride + toff);
    if (mb) {
      mb += (int64_t)chb * HW;
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) prm[i] = *reinterpret_cast<const uint32_t*>(mb + i * gstride + toff);
    }
    const float* wb = w_n + (int64_t)c0 * Co;
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) prw[i] = *reinterpret_cast<const v4f*>(wb + woff[i]);   // unconditional
  };

  auto commit = [&](int c0) __attribute__((always_inline)) {
    if (pvalid) {
      const bool ina = c0 < p.a.C;
      const bool has_scale = (ina ? p.a.scale : p.b.scale) != nullptr;
      const bool has_mask = (ina ? p.a.emask : p.b.emask) != nullptr;
      const bool has_cm = (ina ? p.a.cmask : p.b.cmask) != nullptr;
      const float es = ina ? p.a.es : p.b.es;
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        wsl_v2f lo = {pre[i].x, pre[i].y}, hi = {pre[i].z, pre[i].w};
        const int c = c0 + grp + i * C::G;
        if (has_scale) {
          const float2 t = tab[c];
          xform_bn_leaky(lo, hi, t.x, t.y);
        }
        if (has_mask) xform_mask(lo, hi, prm[i], es);   // keep-mask bytes are 0 or 1
        if (has_cm) {
          const float cm = cm_l[c];
          lo = lo * cm, hi = hi * cm;
        }
        *reinterpret_cast<float4*>(in_t + i * (C::G * C::PLANE) + loff) = make_float4(lo[0], lo[1], hi[0], hi[1]);
      }
    }
#pragma unroll
    for (int i = 0; i < C::NWL; ++i)
      if ((i + 1) * kThreads <= C::WF4 || tid + i * kThreads < C::WF4) *reinterpret_cast<v4f*>(w_t + wl[i]) = prw[i];
  };

#ifndef WSL_HOST_EMUL
  // debug timeline (WSL_CONV_ABLATE & 128): thread 0 stamps s_memtime at the phase boundaries of this workgroup
  uint64_t* tl = WSL_ABLATED(p, 128) ? reinterpret_cast<uint64_t*>(p.stat_part) + (int64_t)tile_id * 32 : nullptr;
#define WSL_MARK(k) do { if (tl && tid == 0) tl[(k)] = __builtin_amdgcn_s_memtime(); } while (0)
  if (tl && tid == 0) tl[29] = __builtin_amdgcn_s_memrealtime(), tl[28] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#else
#define WSL_MARK(k)
#endif
  WSL_MARK(0);
  issue(0);
  // tables and the one-time zero fill of this thread's LDS slots (positions outside the image stay zero for good)
  for (int c = tid; c < Ci; c += kThreads) {
    const bool ina = c < p.a.C;
    const Src2& s = ina ? p.a : p.b;
    const int ch = ina ? c : c - p.a.C;
    tab[c] = s.scale ? make_float2(s.scale[ch], s.shift[ch]) : make_float2(1.f, 0.f);
    cm_l[c] = s.cmask ? s.cmask[(int64_t)n * s.C + ch] : 1.f;
  }
  if (owner && !pvalid) {
#pragma unroll
    for (int i = 0; i < C::NLD; ++i)
      *reinterpret_cast<float4*>(in_t + i * (C::G * C::PLANE) + loff) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  v4f acc[C::MT][C::NT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i)
#pragma unroll
    for (int j = 0; j < C::NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  int abase[C::MT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i) {
    const int mt = wave * C::MT + i;
    abase[i] = (lane >> 4) * C::PLANE + (mt / C::SEGS) * C::ROWP + (mt % C::SEGS) * 16 + (lane & 15) + (C::PADL - C::P);
  }
  const int bbase = (lane >> 4) * C::CSTR + (lane & 15);
  __syncthreads();  // tables visible
  WSL_MARK(1);
  for (int c0 = 0; c0 < Ci; c0 += KC) {
#ifndef WSL_HOST_EMUL
    const int mk = 2 + 6 * (c0 / KC < 4 ? c0 / KC : 3);
    if (tl) __builtin_amdgcn_s_waitcnt(0);   // prefetched data has arrived
#endif
    WSL_MARK(mk);
    if (!WSL_ABLATED(p, 2) || c0 == 0) commit(c0);
    WSL_MARK(mk + 1);
    __syncthreads();
    WSL_MARK(mk + 2);
    if (c0 + KC < Ci && !WSL_ABLATED(p, 2)) issue(c0 + KC);   // in flight during the MFMA loop below
    WSL_MARK(mk + 3);
    if (!WSL_ABLATED(p, 1)) conv2_mfma_stages<C, KS, KC, KC / 4>(in_t, w_t, abase, bbase, acc);
    WSL_MARK(mk + 4);
    __syncthreads();
    WSL_MARK(mk + 5);
  }

  // ---- epilogue: bias, float4 stores, BatchNorm partial statistics (tile and channel block are full by eligibility)
  if (WSL_ABLATED(p, 4)) {
    if (acc[0][0][0] == 123.456f) p.y[0] = 1.f;   // keep the accumulators live
    return;
  }
  float bsum[C::NT];
  {
    constexpr int RPW = C::MT / C::SEGS;   // output rows per wave
    float* yb = p.y + n * p.y_bs + (int64_t)(co0 + (lane & 15)) * HW + (int64_t)(y0 + wave * RPW) * W + x0 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const float bias = p.bias ? p.bias[co0 + j * 16 + (lane & 15)] : 0.f;
      float* yj = yb + (int64_t)j * 16 * HW;
      float bs = 0.f;
#pragma unroll
      for (int i = 0; i < C::MT; ++i) {
        v4f v = acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias;
        acc[i][j] = v;
        *reinterpret_cast<float4*>(yj + (i / C::SEGS) * W + (i % C::SEGS) * 16) = make_float4(v[0], v[1], v[2], v[3]);
        bs += (v[0] + v[1]) + (v[2] + v[3]);
      }
      bsum[j] = bs;
    }
  }
#ifndef WSL_HOST_EMUL
  if (tl) {
    __builtin_amdgcn_s_waitcnt(0);   // stores acknowledged
    WSL_MARK(26);
    if (tid == 0) tl[30] = __builtin_amdgcn_s_memrealtime();
    return;
  }
#endif
  // BatchNorm-backward statistics of the layer that consumes this gradient (tile and channel block are full).  Only in the
  // instantiations with <= 32 accumulator registers: with 64 the extra loads push the allocator into scratch (72-92 spills)
  if constexpr (C::MT * C::NT * 4 <= 32) if (p.bn.part) {
    constexpr int RPW = C::MT / C::SEGS;
    float s1[C::NT], s2[C::NT];
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int co = co0 + j * 16 + (lane & 15);
      const float mean = p.bn.st[co], invstd = p.bn.st[Co + co], sc = p.bn.st[2 * Co + co], sh = p.bn.st[3 * Co + co];
      const int64_t base = ((int64_t)n * Co + co) * HW + (int64_t)(y0 + wave * RPW) * W + x0 + (lane >> 4) * 4;
      BnBwdAcc ba;
#pragma unroll
      for (int i = 0; i < C::MT; ++i) {
        bn_bwd_acc4(p.bn, base + (i / C::SEGS) * W + (i % C::SEGS) * 16, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3],
                    mean, invstd, sc, sh, ba);
      }
      bn_bwd_fold(ba, s1[j], s2[j]);
    }
    bn_bwd_store<C::NT, CO_T>(p.bn, s1, s2, in_t, co0, Co, tile_id, nb);
    return;
  }
  if (p.stat_part) {
    float* red1 = in_t;
    float* red2 = in_t + 4 * CO_T;
    constexpr float cnt = (float)(TH * TW);
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      float s = bsum[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) red1[wave * CO_T + j * 16 + lane] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int col = j * 16 + (lane & 15);
      const float mean_b = (red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col]) / cnt;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[i][j][r] - mean_b;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (lane < 16) red2[wave * CO_T + j * 16 + lane] = q;
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        const int col = j * 16 + lane, co = co0 + col;
        float* dst = p.stat_part + ((int64_t)co * ((int64_t)nb * p.slots) + (int64_t)tile_id * p.slots) * 2;   // [Co][slots][2]
        dst[0] = red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col];
        dst[1] = red2[col] + red2[CO_T + col] + red2[2 * CO_T + col] + red2[3 * CO_T + col];
      }
      if (lane < p.slots && blockIdx.y == 0) p.stat_cnt[tile_id * p.slots + lane] = lane == 0 ? cnt : 0.f;
    }
  }
}
#undef WSL_MARK

// ------------------------------------------------------------------------------------------------ raw-source variant (LDS DMA)
// conv_mfma2l_kernel for launches whose sources carry no loader transform (every data-gradient launch: the source is a
// plain gradient tensor).  The input tile then needs no VGPR round trip at all: global_load_lds_dwordx4 streams it into
// LDS (the tile layout is lane-contiguous: a thread's slot is 16 * tid bytes into its plane group), double-buffered so the
// DMA of chunk k+1 flies during the MFMA stages of chunk k.  Per chunk a wave issues NLD DMA instructions + the weight
// prefetch; no staging VALU, no input ds_write, ~35 fewer VGPRs than the lean kernel.
template <int KS, int TH, int TW, int CO_T, int KC>
struct Conv2RCfg : Conv2Cfg<KS, TH, TW, CO_T, KC> {
  using B = Conv2Cfg<KS, TH, TW, CO_T, KC>;
  static constexpr size_t SMEM = sizeof(float) * (2 * B::IN_FLOATS + B::W_FLOATS);
  static constexpr int MINWR = (B::MT * B::NT * 4 <= 64) ? 3 : 2;
  static_assert(B::PLANE == B::ROWS * B::ROWP && B::PLANE == 4 * B::POS, "lane-contiguous tile layout");
};

template <int KS, int TH, int TW, int CO_T, int KC>
__global__ __launch_bounds__(256, (Conv2RCfg<KS, TH, TW, CO_T, KC>::MINWR)) void conv_mfma2r_kernel(Conv2P p) {
  using C = Conv2RCfg<KS, TH, TW, CO_T, KC>;
  static_assert(KC % C::G == 0 && C::MT % C::SEGS == 0, "lean staging shape");
  WSL_DYN_SMEM(smem);
  float* in_b = reinterpret_cast<float*>(smem);          // two input tiles
  float* w_t = in_b + 2 * C::IN_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);   // XCD-aware tile order, as conv_mfma2_kernel
  const int tile_id = bid;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = blockIdx.y * CO_T;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = p.H, W = p.W, Ci = p.Ci, Co = p.Co;
  const int HW = H * W;

  const int grp = tid / C::POS, pos = tid - grp * C::POS;
  const int pty = pos / C::ROWP4, ptx4 = pos - pty * C::ROWP4;
  const int gy = y0 + pty - C::P, gx = x0 + ptx4 * 4 - C::PADL;
  const bool owner = grp < C::G;
  const bool pvalid = owner && gy >= 0 && gy < H && gx >= 0 && gx < W;
  const uint32_t toff = pvalid ? (uint32_t)(grp * HW + gy * W + gx) : 0u;
  const int64_t gstride = (int64_t)C::G * HW;
  const float* xa_n = p.a.x + n * p.a.bs;
  const float* xb_n = p.b.C ? p.b.x + n * p.b.bs : nullptr;

  uint32_t woff[C::NWL];
  int wl[C::NWL];
#pragma unroll
  for (int i = 0; i < C::NWL; ++i) {
    const int f = tid + i * kThreads;
    const int row = f / C::WQ, q = f - row * C::WQ;
    const int tap = row / KC, c = row - tap * KC;
    woff[i] = f < C::WF4 ? (uint32_t)((tap * Ci + c) * Co + q * 4) : 0u;
    wl[i] = row * C::CSTR + q * 4;
  }
  const float* w_n = p.wp + co0;
  v4f prw[C::NWL];

  // slots of positions outside the image (and of both buffers) stay zero for the whole kernel: the DMA never touches them
  if (owner && !pvalid) {
#pragma unroll
    for (int bsel = 0; bsel < 2; ++bsel)
#pragma unroll
      for (int i = 0; i < C::NLD; ++i)
        *reinterpret_cast<float4*>(in_b + bsel * C::IN_FLOATS + i * (C::G * C::PLANE) + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  auto issue = [&](int c0, int bsel) __attribute__((always_inline)) {
    const bool ina = c0 < p.a.C;                                   // uniform
    const int chb = ina ? c0 : c0 - p.a.C;
    const float* xb = (ina ? xa_n : xb_n) + (int64_t)chb * HW;
    float* dst = in_b + bsel * C::IN_FLOATS + wave * 256;          // wave-uniform: lane l lands at dst + 4 * l floats
    if (pvalid) {
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) WSL_LDS_DMA16(xb + i * gstride + toff, dst + i * (C::G * C::PLANE));
    }
    const float* wb = w_n + (int64_t)c0 * Co;
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) prw[i] = *reinterpret_cast<const v4f*>(wb + woff[i]);
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::NWL; ++i)
      if ((i + 1) * kThreads <= C::WF4 || tid + i * kThreads < C::WF4) *reinterpret_cast<v4f*>(w_t + wl[i]) = prw[i];
  };

  issue(0, 0);
  v4f acc[C::MT][C::NT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i)
#pragma unroll
    for (int j = 0; j < C::NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  int abase[C::MT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i) {
    const int mt = wave * C::MT + i;
    abase[i] = (lane >> 4) * C::PLANE + (mt / C::SEGS) * C::ROWP + (mt % C::SEGS) * 16 + (lane & 15) + (C::PADL - C::P);
  }
  const int bbase = (lane >> 4) * C::CSTR + (lane & 15);
  int bsel = 0;
  for (int c0 = 0; c0 < Ci; c0 += KC, bsel ^= 1) {
    commit();
    WSL_WAIT_ALL();        // this chunk's DMA has landed
    __syncthreads();
    if (c0 + KC < Ci) issue(c0 + KC, bsel ^ 1);   // next chunk streams into the other buffer during the MFMA loop
    conv2_mfma_stages<typename C::B, KS, KC, KC / 4>(in_b + bsel * C::IN_FLOATS, w_t, abase, bbase, acc);
    __syncthreads();
  }

  // ---- epilogue: as conv_mfma2l_kernel
  float bsum[C::NT];
  {
    constexpr int RPW = C::MT / C::SEGS;
    float* yb = p.y + n * p.y_bs + (int64_t)(co0 + (lane & 15)) * HW + (int64_t)(y0 + wave * RPW) * W + x0 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const float bias = p.bias ? p.bias[co0 + j * 16 + (lane & 15)] : 0.f;
      float* yj = yb + (int64_t)j * 16 * HW;
      float bs = 0.f;
#pragma unroll
      for (int i = 0; i < C::MT; ++i) {
        v4f v = acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias;
        acc[i][j] = v;
        *reinterpret_cast<float4*>(yj + (i / C::SEGS) * W + (i % C::SEGS) * 16) = make_float4(v[0], v[1], v[2], v[3]);
        bs += (v[0] + v[1]) + (v[2] + v[3]);
      }
      bsum[j] = bs;
    }
  }
  if (p.stat_part) {
    float* red1 = in_b;
    float* red2 = in_b + 4 * CO_T;
    constexpr float cnt = (float)(TH * TW);
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      float s = bsum[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) red1[wave * CO_T + j * 16 + lane] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int col = j * 16 + (lane & 15);
      const float mean_b = (red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col]) / cnt;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[i][j][r] - mean_b;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (lane < 16) red2[wave * CO_T + j * 16 + lane] = q;
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        const int col = j * 16 + lane, co = co0 + col;
        float* dst = p.stat_part + ((int64_t)co * ((int64_t)nb * p.slots) + (int64_t)tile_id * p.slots) * 2;
        dst[0] = red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col];
        dst[1] = red2[col] + red2[CO_T + col] + red2[2 * CO_T + col] + red2[3 * CO_T + col];
      }
      if (lane < p.slots && blockIdx.y == 0) p.stat_cnt[tile_id * p.slots + lane] = lane == 0 ? cnt : 0.f;
    }
  }
}

template <int KS, int TH, int TW, int CO_T>
static int launch_conv2r(Conv2P& p, int wmode_for_prof, void* stream) {
  using C = Conv2RCfg<KS, TH, TW, CO_T, 8>;
  auto kern = conv_mfma2r_kernel<KS, TH, TW, CO_T, 8>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, p.Co / CO_T);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(wmode_for_prof ? 1 : 0, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci) + bn_epi_bytes(p.bn, px * p.Co), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_mfma2r_kernel");
}

// packed[tap][ci][co] = wmode 0: w[co][ci][tap]   |   wmode 1 (data gradient): w[ci][co][KK-1-tap]
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, float* wp, int Co, int Ci, int KK, int wmode) {
  const int64_t total = (int64_t)KK * Ci * Co;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
    const int co = (int)(e % Co);
    const int64_t r = e / Co;
    const int ci = (int)(r % Ci), tap = (int)(r / Ci);
    wp[e] = wmode == 0 ? w[((int64_t)co * Ci + ci) * KK + tap] : w[((int64_t)ci * Co + co) * KK + (KK - 1 - tap)];
  }
}

// every conv layer of a network in ONE launch: blockIdx.y = layer, blockIdx.z = 0 forward image / 1 data-gradient image
__global__ __launch_bounds__(256) void pack_table_kernel(PackTable t, const float* params, float* packf, float* packd) {
  const PackEntry e = t.e[blockIdx.y];
  const int dgrad = blockIdx.z;
  if ((e._pad >> dgrad) & 1) return;   // (the network's launches of this layer and direction take the Winograd image: wsl_net.hip, pack_all)
  const int Co = dgrad ? e.Ci : e.Co, Ci = dgrad ? e.Co : e.Ci, KK = e.KK;   // GEMM-out / GEMM-in of this image
  const float* w = params + e.w;
  float* wp = (dgrad ? packd : packf) + e.w;
  const int64_t total = (int64_t)KK * Ci * Co;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int co = (int)(i % Co);
    const int64_t r = i / Co;
    const int ci = (int)(r % Ci), tap = (int)(r / Ci);
    wp[i] = dgrad ? w[((int64_t)ci * Co + co) * KK + (KK - 1 - tap)] : w[((int64_t)co * Ci + ci) * KK + tap];
  }
}

static double pack_table_elems(const PackTable& t) {
  double e = 0.0;
  for (int i = 0; i < t.n; ++i) e += (double)t.e[i].Co * t.e[i].Ci * t.e[i].KK;
  return e;
}

int conv2_pack_table(const PackTable& t, const float* params, float* packf, float* packd, int with_dgrad, void* stream) {
  ProfScope ps(PF_PREP, 0.0, 4.0 * pack_table_elems(t) * (with_dgrad ? 3.0 : 2.0), stream);
  WSL_LAUNCH(pack_table_kernel, dim3(32, t.n, with_dgrad ? 2 : 1), dim3(kThreads), 0, stream, t, params, packf, packd);
  return check_launch("pack_table_kernel");
}

template <int KS, int TH, int TW, int CO_T>
static int launch_conv2l(Conv2P& p, int wmode_for_prof, void* stream) {
  using C = Conv2Cfg<KS, TH, TW, CO_T, 8>;
  auto kern = conv_mfma2l_kernel<KS, TH, TW, CO_T, 8>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, p.Co / CO_T);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(wmode_for_prof ? 1 : 0, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci) + bn_epi_bytes(p.bn, px * p.Co), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_mfma2l_kernel");
}

template <int KS, int TH, int TW, int CO_T>
static int launch_conv2(Conv2P& p, int wmode_for_prof, void* stream) {
  using C = Conv2Cfg<KS, TH, TW, CO_T, 8>;
  // the lean kernel takes every shape it is eligible for (WSL_CONV_LEAN=0 forces the generic one, for A/B timing)
  static const bool lean_on = (WSL_TUNE("WSL_CONV_LEAN", 1) != 0);
  const int64_t span = (int64_t)(p.a.C > p.b.C ? p.a.C : p.b.C) * p.H * p.W;
  if (lean_on && p.Ci % 8 == 0 && (p.b.C == 0 || p.a.C % 8 == 0) && p.Co % CO_T == 0 && p.H % TH == 0 && p.W % TW == 0 &&
      span < (int64_t(1) << 31) && (int64_t)KS * KS * p.Ci * p.Co < (int64_t(1) << 31)) {
    // measured neutral (+-3 % per layer, +0.2 % per step: profiles/r1g_conv_timeline.md), so it stays opt-in
    static const bool dma_on = WSL_TUNE("WSL_CONV_DMA", 0) != 0;
    const bool raw = !p.a.scale && !p.a.emask && !p.a.cmask && (p.b.C == 0 || (!p.b.scale && !p.b.emask && !p.b.cmask));
    if constexpr (KS == 3) {   // (1x1 tiles are padded per plane: their slots are not lane-contiguous)
      if (dma_on && raw && !p.bn.part) return launch_conv2r<KS, TH, TW, CO_T>(p, wmode_for_prof, stream);   // plain sources: LDS DMA (... except this opt-in one)
    }
    return launch_conv2l<KS, TH, TW, CO_T>(p, wmode_for_prof, stream);
  }
  auto kern = conv_mfma2_kernel<KS, TH, TW, CO_T, 8>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, cdiv(p.Co, CO_T));
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(wmode_for_prof ? 1 : 0, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci) + bn_epi_bytes(p.bn, px * p.Co), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_mfma2_kernel");
}

static Src2 to_src2(const WslSrc& s) {
  return Src2{s.x, s.emask, s.scale, s.shift, s.cmask, s.bs, s.C, s.emask_scale};
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool conv2_eligible(const WslSrc& a, const WslSrc* b, const float* y, int64_t y_bs, int W, int Ci) {
  if ((W & 3) || Ci > 512) return false;
  if (b && b->C > 0 && (a.C & 3)) return false;   // a channel group of one load never straddles the two sources
  if (!aligned16(a.x) || (a.bs & 3) || (a.emask && (reinterpret_cast<uintptr_t>(a.emask) & 3))) return false;
  if (b && b->C > 0 && (!aligned16(b->x) || (b->bs & 3) || (b->emask && (reinterpret_cast<uintptr_t>(b->emask) & 3))))
    return false;
  return y == nullptr || (aligned16(y) && !(y_bs & 3));
}

int conv2_pack(const float* w, float* wp, int Co, int Ci, int ks, int wmode, void* stream) {
  const int64_t total = (int64_t)ks * ks * Ci * Co;
  int64_t blocks = (total + kThreads - 1) / kThreads;
  if (blocks > 1024) blocks = 1024;
  WSL_LAUNCH(pack_weights_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, w, wp, Co, Ci, ks * ks, wmode);
  return check_launch("pack_weights_kernel");
}

int conv2_fwd(const WslSrc& a, const WslSrc* b, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H,
              int W, int Co, int ks, int is_dgrad, int th, int tw, int co_t, float* stat_part, float* stat_cnt,
              int slots, void* stream, const BnBwdEpi* bn, int* bn_done) {
  Conv2P p;
  p.slots = slots;
  // the generic and the lean kernel carry the statistics epilogue -- the lean one in its instantiations with <= 32
  // accumulator registers (th * tw * co_t <= 8192); the caller falls back to the stand-alone reduction pass otherwise
  if (bn && bn->part && th * tw * co_t <= 8192) p.bn = *bn;
  if (bn_done) *bn_done = p.bn.part ? 1 : 0;
  p.a = to_src2(a);
  p.b = (b && b->C > 0) ? to_src2(*b) : Src2{};
  p.wp = wp, p.bias = bias, p.y = y, p.y_bs = y_bs;
  p.N = N, p.H = H, p.W = W, p.Ci = a.C + p.b.C, p.Co = Co;
  p.tiles_x = cdiv(W, tw), p.tiles_y = cdiv(H, th);
  p.stat_part = stat_part, p.stat_cnt = stat_cnt;
  static const int ablate = WSL_TUNE("WSL_CONV_ABLATE", 0);
  p.ablate = ablate;
#define WSL_CASE(KS_, TH_, TW_, CO_) \
  if (ks == KS_ && th == TH_ && tw == TW_ && co_t == CO_) return launch_conv2<KS_, TH_, TW_, CO_>(p, is_dgrad, stream);
  WSL_CASE(3, 8, 64, 16) WSL_CASE(3, 8, 64, 32) WSL_CASE(3, 8, 32, 16) WSL_CASE(3, 8, 32, 32) WSL_CASE(3, 8, 32, 64)
  WSL_CASE(3, 16, 16, 16) WSL_CASE(3, 16, 16, 32) WSL_CASE(3, 16, 16, 64)
  WSL_CASE(1, 8, 64, 16) WSL_CASE(1, 8, 64, 32) WSL_CASE(1, 8, 32, 16) WSL_CASE(1, 8, 32, 32) WSL_CASE(1, 8, 32, 64)
  WSL_CASE(1, 16, 16, 16) WSL_CASE(1, 16, 16, 32) WSL_CASE(1, 16, 16, 64)
#undef WSL_CASE
  set_error("conv2: no kernel for ks %d tile %dx%d co_t %d", ks, th, tw, co_t);
  return WSL_EUNSUPPORTED;
}

// ================================================================================================ weight gradient v2
struct Wgrad2P {
  Src2 a, b;
  const float* dy;
  int64_t dy_bs;
  float* part_dw;  // [nsplit][KK][Co][Ci]
  float* part_db;  // [nsplit][Co]
  int N, H, W, Ci, Co, tiles_x, tiles_y, items, nsplit, co_blocks;
  int ablate;   // debug (env WSL_WGRAD_ABLATE): 1 skip MFMA, 2 skip staging after the first tile, 8 MFMA without LDS reads
};

template <int KS, int TH, int TW, int CB, int IB, int WK>
struct Wgrad2Cfg {
  static constexpr int P = KS / 2, KK = KS * KS, PADL = P ? 4 : 0;
  static constexpr int ROWP = TW + 2 * PADL, ROWS = TH + 2 * P, ROWP4 = ROWP / 4, S = TH * TW;
  static constexpr int PD = S / 4, PA = ROWS * ROWP4;          // float4 positions per channel (dy / input)
  static constexpr int GD = 256 / PD, GA = 256 / PA;
  static constexpr int ND = (CB + GD - 1) / GD, NA = (IB + GA - 1) / GA;
  static constexpr int PLD = ((S - 2 + 31) / 32) * 32 + 2;                // == 2 (mod 32)
  static constexpr int PLA = ((ROWS * ROWP - 2 + 31) / 32) * 32 + 2;      // == 2 (mod 32)
  static constexpr int CBT = CB / 16, IBT = IB / 16, PAIRS = CBT * IBT, WP = 4 / WK, PP = PAIRS / WP;
  static constexpr int DY_FLOATS = CB * PLD, A_FLOATS = IB * PLA;
  static constexpr int RED_FLOATS = (WK > 1) ? 4 * 64 * (PP * (KK + 1) * 4) : 0;
  static constexpr int MAIN_FLOATS = DY_FLOATS + A_FLOATS > RED_FLOATS ? DY_FLOATS + A_FLOATS : RED_FLOATS;
  static constexpr size_t SMEM = sizeof(float) * (MAIN_FLOATS + 2 * IB);
  static_assert(PD <= 256 && PA <= 256 && GD >= 1 && GA >= 1 && PAIRS % WP == 0 && TH % WK == 0, "wgrad2 tile shape");
};

template <int KS, int TH, int TW, int CB, int IB, int WK>
__global__ __launch_bounds__(256, 2) void wgrad_mfma2_kernel(Wgrad2P p) {
  using C = Wgrad2Cfg<KS, TH, TW, CB, IB, WK>;
  WSL_DYN_SMEM(smem);
  float* dy_t = reinterpret_cast<float*>(smem);
  float* a_t = dy_t + C::DY_FLOATS;
  float* sc_l = dy_t + C::MAIN_FLOATS;  // [IB] scale / shift of this block's input channels
  float* sh_l = sc_l + IB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x % p.co_blocks, ib = blockIdx.x / p.co_blocks, split = blockIdx.y;
  const int co0 = cb * CB, ci0 = ib * IB;
  const int wp_ = wave % C::WP, wk = wave / C::WP;
  const int H = p.H, W = p.W, Ci = p.Ci;
  const int64_t HW = (int64_t)H * W;

  for (int c = tid; c < IB; c += kThreads) {
    const int cg = ci0 + c;
    float sc = 1.f, sh = 0.f;
    if (cg < Ci) {
      const bool ina = cg < p.a.C;
      const Src2& s = ina ? p.a : p.b;
      const int ch = ina ? cg : cg - p.a.C;
      if (s.scale) sc = s.scale[ch], sh = s.shift[ch];
    }
    sc_l[c] = sc, sh_l[c] = sh;
  }

  // fixed staging positions of this thread
  const int gd = tid / C::PD, pd = tid - gd * C::PD;             // dy: float4 index pd inside the TH x TW tile
  const int dty = (pd * 4) / TW, dtx = (pd * 4) - dty * TW;
  const int ga = tid / C::PA, pa = tid - ga * C::PA;             // input: float4 index inside the halo tile
  const int aty = pa / C::ROWP4, atx4 = pa - aty * C::ROWP4;
  const int aloff = aty * C::ROWP + atx4 * 4;

  v4f acc[C::PP][C::KK];
  v4f accb[C::PP];
#pragma unroll
  for (int j = 0; j < C::PP; ++j) {
    accb[j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < C::KK; ++t) acc[j][t] = v4f{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_db = (ib == 0) && (p.part_db != nullptr);
  const int it0 = (int)((int64_t)split * p.items / p.nsplit), it1 = (int)((int64_t)(split + 1) * p.items / p.nsplit);

  float4 prd[C::ND], pra[C::NA];
  uchar4 prm[C::NA];
  float prc[C::NA];
  bool pr_aok = false;   // the prefetched input position lies inside the image (the BN transform applies)

  auto issue = [&](int item) {
    int q = item;
    const int tx_i = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty_i = q % p.tiles_y;
    const int n = q / p.tiles_y;
    const int y0 = ty_i * TH, x0 = tx_i * TW;
    {
      const int gy = y0 + dty, gx = x0 + dtx;
      const bool ok = gd < C::GD && gy < H && gx < W;
      const uint32_t toff = (uint32_t)(gd * (int)HW + gy * W + gx);
#pragma unroll
      for (int i = 0; i < C::ND; ++i) {
        const int cbu = co0 + i * C::GD;        // uniform channel base of this load instruction
        prd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cbu < p.Co) {
          const float* db = p.dy + n * p.dy_bs + (int64_t)cbu * HW;
          if (ok && i * C::GD + gd < CB && cbu + gd < p.Co) prd[i] = *reinterpret_cast<const float4*>(db + toff);
        }
      }
    }
    {
      const int gy = y0 + aty - C::P, gx = x0 + atx4 * 4 - C::PADL;
      pr_aok = ga < C::GA && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const uint32_t toff = (uint32_t)(ga * (int)HW + gy * W + gx);
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        const int cbu = ci0 + i * C::GA, cg = cbu + ga;
        pra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        prm[i] = make_uchar4(1, 1, 1, 1);
        prc[i] = 1.f;
        if (cbu < Ci) {
          const bool ina = cbu < p.a.C;
          const Src2& s = ina ? p.a : p.b;
          const int chb = ina ? cbu : cbu - p.a.C;
          const float* xb = s.x + n * s.bs + (int64_t)chb * HW;
          const uint8_t* mb = s.emask ? s.emask + ((int64_t)n * s.C + chb) * HW : nullptr;
          if (pr_aok && i * C::GA + ga < IB && cg < Ci) {
            pra[i] = *reinterpret_cast<const float4*>(xb + toff);
            if (mb) prm[i] = *reinterpret_cast<const uchar4*>(mb + toff);
            if (s.cmask) prc[i] = s.cmask[(int64_t)n * s.C + chb + ga];
          }
        }
      }
    }
  };

  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < C::ND; ++i) {
      const int c = gd + i * C::GD;
      if (gd < C::GD && c < CB) {
        float* dst = dy_t + c * C::PLD + pd * 4;   // plane stride == 2 (mod 32): 8-byte aligned, not 16
        dst[0] = prd[i].x, dst[1] = prd[i].y, dst[2] = prd[i].z, dst[3] = prd[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      const int c = ga + i * C::GA, cg = ci0 + c;
      if (ga < C::GA && c < IB) {
        float4 v = pra[i];
        if (pr_aok && cg < Ci) {
          const bool ina = cg < p.a.C;
          const Src2& s = ina ? p.a : p.b;
          if (s.scale) {
            const float sc = sc_l[c], sh = sh_l[c];
            v.x = leaky(fmaf(v.x, sc, sh)), v.y = leaky(fmaf(v.y, sc, sh));
            v.z = leaky(fmaf(v.z, sc, sh)), v.w = leaky(fmaf(v.w, sc, sh));
          }
          if (s.emask) {
            const uchar4 m = prm[i];
            v.x = m.x ? v.x * s.es : 0.f, v.y = m.y ? v.y * s.es : 0.f;
            v.z = m.z ? v.z * s.es : 0.f, v.w = m.w ? v.w * s.es : 0.f;
          }
          if (s.cmask) v.x *= prc[i], v.y *= prc[i], v.z *= prc[i], v.w *= prc[i];
        }
        float* dst = a_t + c * C::PLA + aloff;
        dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
      }
    }
  };

  if (it0 < it1) issue(it0);
  __syncthreads();  // BN tables visible
  for (int item = it0; item < it1; ++item) {
    commit();
    __syncthreads();
    if (item + 1 < it1) issue(item + 1);   // prefetch the next tile; in flight during the MFMA loop
    constexpr int RW = TH / WK, NX = TW / 4, NSTEP = RW * NX;
    static_assert(C::PP == 1, "one channel pair per wave");
    const int cot = wp_ / C::IBT, cit = wp_ % C::IBT;
    const float* dyp = dy_t + (cot * 16 + (lane & 15)) * C::PLD + (lane >> 4);
    const float* ap = a_t + (cit * 16 + (lane & 15)) * C::PLA + (lane >> 4) + (C::PADL - C::P);
    float avv[2], bvv[2][C::KK];
    auto load = [&](int st, int buf) {   // step st = (row, group of 4 pixels)
      const int r = wk * RW + st / NX, x4 = st % NX;
      avv[buf] = dyp[r * TW + x4 * 4];
#pragma unroll
      for (int t = 0; t < C::KK; ++t) bvv[buf][t] = ap[(r + t / KS) * C::ROWP + x4 * 4 + (t % KS)];
    };
    load(0, 0);
    const bool dbw = want_db && cit == 0;
#pragma unroll 2
    for (int st = 0; st < NSTEP; ++st) {   // operands of step st+1 are read before the MFMAs of step st issue
      const int cur = st & 1;
      if (st + 1 < NSTEP) load(st + 1, cur ^ 1);
      if (dbw) accb[0] = WSL_MFMA16(avv[cur], 1.0f, accb[0]);
#pragma unroll
      for (int t = 0; t < C::KK; ++t) acc[0][t] = WSL_MFMA16(avv[cur], bvv[cur][t], acc[0][t]);
      WSL_SCHED_BARRIER();
    }
    __syncthreads();
  }
  // ---- merge the WK row-groups (fixed order) and store partials
  if (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);
    constexpr int PER = C::PP * (C::KK + 1) * 4;
    float* mine = red + (wave * 64 + lane) * PER;
#pragma unroll
    for (int j = 0; j < C::PP; ++j) {
#pragma unroll
      for (int t = 0; t < C::KK; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(j * (C::KK + 1) + t) * 4 + r] = acc[j][t][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(j * (C::KK + 1) + C::KK) * 4 + r] = accb[j][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int j = 0; j < C::PP; ++j) {
#pragma unroll
        for (int t = 0; t <= C::KK; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sum = 0.f;
            for (int k = 0; k < WK; ++k) sum += red[((k * C::WP + wp_) * 64 + lane) * PER + (j * (C::KK + 1) + t) * 4 + r];
            if (t < C::KK) acc[j][t][r] = sum; else accb[j][r] = sum;
          }
      }
    }
  }
  if (wk == 0) {
#pragma unroll
    for (int j = 0; j < C::PP; ++j) {
      const int pr = wp_ * C::PP + j, cot = pr / C::IBT, cit = pr % C::IBT;
      const int ci = ci0 + cit * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + cot * 16 + (lane >> 4) * 4 + r;
        if (co < p.Co && ci < Ci) {
#pragma unroll
          for (int t = 0; t < C::KK; ++t)
            p.part_dw[(((int64_t)split * C::KK + t) * p.Co + co) * Ci + ci] = acc[j][t][r];
        }
        if (want_db && cit == 0 && (lane & 15) == 0 && co < p.Co) p.part_db[(int64_t)split * p.Co + co] = accb[j][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ lean weight gradient
// wgrad_mfma2_kernel for the shapes of the training step (Ci % IB == 0, Co % CB == 0, full tiles, a channel block that
// lies inside one source), with the staging rebuilt like conv_mfma2l_kernel: the source of the block is chosen once per
// workgroup, addresses are a scalar base + one per-thread offset, no load sits under a divergent branch (threads
// outside the image read a valid dummy element and stage zeros), LDS writes are 8-byte pairs.
template <int KS, int TH, int TW, int CB, int IB, int WK>
__global__ __launch_bounds__(256, 2) void wgrad_mfma2l_kernel(Wgrad2P p) {
  using C = Wgrad2Cfg<KS, TH, TW, CB, IB, WK>;
  static_assert(CB % C::GD == 0 && IB % C::GA == 0 && C::PP == 1, "lean wgrad staging shape");
  WSL_DYN_SMEM(smem);
  float* dy_t = reinterpret_cast<float*>(smem);
  float* a_t = dy_t + C::DY_FLOATS;
  float2* tab = reinterpret_cast<float2*>(dy_t + C::MAIN_FLOATS);   // [IB] {scale, shift} of this block's channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x % p.co_blocks, ib = blockIdx.x / p.co_blocks, split = blockIdx.y;
  const int co0 = cb * CB, ci0 = ib * IB;
  const int wp_ = wave % C::WP, wk = wave / C::WP;
  const int H = p.H, W = p.W, Ci = p.Ci;
  const int HW = H * W;

  // the source this block of input channels comes from (uniform for the whole workgroup)
  const bool ina = ci0 < p.a.C;
  const Src2& s = ina ? p.a : p.b;
  const int chb0 = ina ? ci0 : ci0 - p.a.C;
  const bool has_scale = s.scale != nullptr, has_mask = s.emask != nullptr, has_cm = s.cmask != nullptr;
  const float es = s.es;
  for (int c = tid; c < IB; c += kThreads) tab[c] = has_scale ? make_float2(s.scale[chb0 + c], s.shift[chb0 + c]) : make_float2(1.f, 0.f);

  // fixed staging positions of this thread
  const int gd = tid / C::PD, pd = tid - gd * C::PD;             // dy: float4 index pd inside the TH x TW tile
  const int dty = (pd * 4) / TW, dtx = (pd * 4) - dty * TW;
  const bool owner_d = gd < C::GD;
  const uint32_t tdoff = owner_d ? (uint32_t)(gd * HW + dty * W + dtx) : 0u;
  const int dloff = gd * C::PLD + pd * 4;
  const int ga = tid / C::PA, pa = tid - ga * C::PA;             // input: float4 index inside the halo tile
  const int aty = pa / C::ROWP4, atx4 = pa - aty * C::ROWP4;
  const bool owner_a = ga < C::GA;
  const int aloff = ga * C::PLA + aty * C::ROWP + atx4 * 4;
  const int64_t dstride = (int64_t)C::GD * HW, astride = (int64_t)C::GA * HW;

  v4f acc[C::KK];
  v4f accb = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < C::KK; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  const bool want_db = (ib == 0) && (p.part_db != nullptr);
  const int it0 = (int)((int64_t)split * p.items / p.nsplit), it1 = (int)((int64_t)(split + 1) * p.items / p.nsplit);

  float4 prd[C::ND], pra[C::NA];
  uint32_t prm[C::NA];
  float prc[C::NA];
  bool pr_aok = false;   // the prefetched input position lies inside the image
  int nx_tx, nx_ty, nx_n;   // tile the next issue() fetches
  {
    int q = it0;
    nx_tx = q % p.tiles_x;
    q /= p.tiles_x;
    nx_ty = q % p.tiles_y;
    nx_n = q / p.tiles_y;
  }

  auto issue = [&]() __attribute__((always_inline)) {
    const int n = nx_n, y0 = nx_ty * TH, x0 = nx_tx * TW;
    if (++nx_tx == p.tiles_x) {
      nx_tx = 0;
      if (++nx_ty == p.tiles_y) nx_ty = 0, ++nx_n;
    }
    const float* dyb = p.dy + n * p.dy_bs + (int64_t)co0 * HW + y0 * W + x0;
#pragma unroll
    for (int i = 0; i < C::ND; ++i) prd[i] = *reinterpret_cast<const float4*>(dyb + i * dstride + tdoff);
    const int gy = y0 + aty - C::P, gx = x0 + atx4 * 4 - C::PADL;
    pr_aok = owner_a && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const uint32_t taoff = pr_aok ? (uint32_t)(ga * HW + gy * W + gx) : 0u;
    const float* xb = s.x + n * s.bs + (int64_t)chb0 * HW;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) pra[i] = *reinterpret_cast<const float4*>(xb + i * astride + taoff);
    if (has_mask) {
      const uint8_t* mb = s.emask + ((int64_t)n * s.C + chb0) * HW;
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prm[i] = *reinterpret_cast<const uint32_t*>(mb + i * astride + taoff);
    }
    if (has_cm) {
      const float* cmb = s.cmask + (int64_t)n * s.C + chb0 + (owner_a ? ga : 0);
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prc[i] = cmb[i * C::GA];
    }
  };

  auto commit = [&]() __attribute__((always_inline)) {
    if (owner_d) {
#pragma unroll
      for (int i = 0; i < C::ND; ++i) {
        float* dst = dy_t + i * (C::GD * C::PLD) + dloff;   // plane stride == 2 (mod 32): 8-byte aligned, not 16
        *reinterpret_cast<float2*>(dst) = make_float2(prd[i].x, prd[i].y);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(prd[i].z, prd[i].w);
      }
    }
    if (owner_a) {
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        wsl_v2f lo = {pra[i].x, pra[i].y}, hi = {pra[i].z, pra[i].w};
        if (has_scale) {
          const float2 t = tab[ga + i * C::GA];
          xform_bn_leaky(lo, hi, t.x, t.y);
        }
        if (has_mask) xform_mask(lo, hi, prm[i], es);   // keep-mask bytes are 0 or 1
        if (has_cm) lo = lo * prc[i], hi = hi * prc[i];
        if (!pr_aok) lo = wsl_v2f{0.f, 0.f}, hi = wsl_v2f{0.f, 0.f};
        float* dst = a_t + i * (C::GA * C::PLA) + aloff;
        *reinterpret_cast<float2*>(dst) = make_float2(lo[0], lo[1]);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(hi[0], hi[1]);
      }
    }
  };

  if (it0 < it1) issue();
  __syncthreads();  // BN table visible
  const int cot = wp_ / C::IBT, cit = wp_ % C::IBT;
  const float* dyp = dy_t + (cot * 16 + (lane & 15)) * C::PLD + (lane >> 4);
  const float* ap = a_t + (cit * 16 + (lane & 15)) * C::PLA + (lane >> 4) + (C::PADL - C::P);
  const bool dbw = want_db && cit == 0;
  for (int item = it0; item < it1; ++item) {
    if (!WSL_ABLATED(p, 2) || item == it0) commit();
    __syncthreads();
    if (item + 1 < it1 && !WSL_ABLATED(p, 2)) issue();   // prefetch the next tile; in flight during the MFMA loop
    constexpr int RW = TH / WK, NX = TW / 4, NSTEP = RW * NX;
    float avv[2], bvv[2][C::KK];
    auto load = [&](int st, int buf) {   // step st = (row, group of 4 pixels)
      const int r = wk * RW + st / NX, x4 = st % NX;
      avv[buf] = dyp[r * TW + x4 * 4];
#pragma unroll
      for (int t = 0; t < C::KK; ++t) bvv[buf][t] = ap[(r + t / KS) * C::ROWP + x4 * 4 + (t % KS)];
    };
    load(0, 0);
    if (WSL_ABLATED(p, 1)) continue;
#pragma unroll 2
    for (int st = 0; st < NSTEP; ++st) {   // operands of step st+1 are read before the MFMAs of step st issue
      const int cur = st & 1;
      if (st + 1 < NSTEP && !WSL_ABLATED(p, 8)) load(st + 1, cur ^ 1);
      if (dbw) accb = WSL_MFMA16(avv[cur], 1.0f, accb);
#pragma unroll
      for (int t = 0; t < C::KK; ++t) acc[t] = WSL_MFMA16(avv[cur], bvv[cur][t], acc[t]);
      WSL_SCHED_BARRIER();
    }
    __syncthreads();
  }
  // ---- merge the WK row-groups (fixed order) and store partials
  if (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);
    constexpr int PER = (C::KK + 1) * 4;
    float* mine = red + (wave * 64 + lane) * PER;
#pragma unroll
    for (int t = 0; t < C::KK; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[t * 4 + r] = acc[t][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[C::KK * 4 + r] = accb[r];
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int t = 0; t <= C::KK; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = 0.f;
          for (int k = 0; k < WK; ++k) sum += red[((k * C::WP + wp_) * 64 + lane) * PER + t * 4 + r];
          if (t < C::KK) acc[t][r] = sum; else accb[r] = sum;
        }
    }
  }
  if (wk == 0) {
    const int ci = ci0 + cit * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + cot * 16 + (lane >> 4) * 4 + r;
#pragma unroll
      for (int t = 0; t < C::KK; ++t) p.part_dw[(((int64_t)split * C::KK + t) * p.Co + co) * Ci + ci] = acc[t][r];
      if (dbw && (lane & 15) == 0) p.part_db[(int64_t)split * p.Co + co] = accb[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ split-halo weight gradient
// Same blocking and staging discipline as wgrad_mfma2l_kernel, with the 3x3 tap offset split between the two operands:
//   dw[co][ci][ky][kx] = sum_p dy[co][p] * a[ci][p + (ky-1, kx-1)] = sum_q dy[co][q - (ky-1) rows] * a[ci][q + (kx-1) cols]
// so a step needs 3 row-shifted A operands (dy, tile with a +-1 row halo) and 3 column-shifted B operands (activations,
// tile with a +-4 column halo): 6 LDS reads feed 9 MFMAs where one-sided shifting needs 10, and the BN/LeakyReLU/mask
// transform runs over TH x (TW+8) activations instead of (TH+2) x (TW+8).  Tiles partition the image in q, out-of-image
// rows of dy and columns of a are staged as zeros (the convolution's zero padding), so every product is counted once.
template <int KS, int TH, int TW, int CB, int IB, int WK>
struct WgradSCfg {
  static constexpr int P = KS / 2, KK = KS * KS, PADL = P ? 4 : 0;
  static constexpr int ROWS_D = TH + 2 * P, SD = ROWS_D * TW, PD = SD / 4, GD = 256 / PD, ND = (CB + GD - 1) / GD;
  static constexpr int ROWP = TW + 2 * PADL, ROWP4 = ROWP / 4, SA = TH * ROWP, PA = SA / 4, GA = 256 / PA, NA = (IB + GA - 1) / GA;
  static constexpr int PLD = ((SD - 2 + 31) / 32) * 32 + 2;   // == 2 (mod 32)
  static constexpr int PLA = ((SA - 2 + 31) / 32) * 32 + 2;   // == 2 (mod 32)
  static constexpr int CBT = CB / 16, IBT = IB / 16, PAIRS = CBT * IBT, WP = 4 / WK, PP = PAIRS / WP;
  static constexpr int DY_FLOATS = CB * PLD, A_FLOATS = IB * PLA;
  static constexpr int RED_FLOATS = (WK > 1) ? 4 * 64 * ((KK + 1) * 4) : 0;
  static constexpr int MAIN_FLOATS = DY_FLOATS + A_FLOATS > RED_FLOATS ? DY_FLOATS + A_FLOATS : RED_FLOATS;
  static constexpr size_t SMEM = sizeof(float) * (MAIN_FLOATS + 2 * IB);
  // PP channel pairs per wave: they share the input-channel tile (one B operand set feeds PP x 9 MFMAs)
  static_assert(PD <= 256 && PA <= 256 && PAIRS % WP == 0 && (PP == 1 || (WK == 1 && CBT % PP == 0)) && TH % WK == 0,
                "split-halo wgrad tile shape");
};

template <int KS, int TH, int TW, int CB, int IB, int WK>
__global__ __launch_bounds__(256, 2) void wgrad_mfma2s_kernel(Wgrad2P p) {
  using C = WgradSCfg<KS, TH, TW, CB, IB, WK>;
  WSL_DYN_SMEM(smem);
  float* dy_t = reinterpret_cast<float*>(smem);
  float* a_t = dy_t + C::DY_FLOATS;
  float2* tab = reinterpret_cast<float2*>(dy_t + C::MAIN_FLOATS);   // [IB] {scale, shift} of this block's channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (nsplit % 8 == 0: all channel blocks of a split on ONE XCD -- dealt round-robin in launch order -- so the dy / input tiles they share
  //  come out of its L2: the 1x1 layers fetched 1.33 x their bytes with the blocks of a split spread over the XCDs)
  int blk = blockIdx.x, split = blockIdx.y;
  if ((p.nsplit & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, j = lin >> 3;
    blk = j % (int)gridDim.x;
    split = (lin & 7) * (p.nsplit >> 3) + j / (int)gridDim.x;
  }
  const int cb = blk % p.co_blocks, ib = blk / p.co_blocks;
  const int co0 = cb * CB, ci0 = ib * IB;
  const int wp_ = wave % C::WP, wk = wave / C::WP;
  const int H = p.H, W = p.W, Ci = p.Ci;
  const int HW = H * W;

  // the source this block of input channels comes from (uniform for the whole workgroup)
  const bool ina = ci0 < p.a.C;
  const Src2& s = ina ? p.a : p.b;
  const int chb0 = ina ? ci0 : ci0 - p.a.C;
  const bool has_scale = s.scale != nullptr, has_mask = s.emask != nullptr, has_cm = s.cmask != nullptr;
  const float es = s.es;
  for (int c = tid; c < IB; c += kThreads) tab[c] = has_scale ? make_float2(s.scale[chb0 + c], s.shift[chb0 + c]) : make_float2(1.f, 0.f);

  // fixed staging positions of this thread
  const int gd = tid / C::PD, pd = tid - gd * C::PD;             // dy: float4 index pd inside the (TH+2P) x TW tile
  const int dty = (pd * 4) / TW, dtx = (pd * 4) - dty * TW;
  const bool owner_d = gd < C::GD;
  const int tdconst = gd * HW + (dty - C::P) * W + dtx;          // + y0 * W + x0 per tile (valid rows only)
  const int dloff = gd * C::PLD + pd * 4;
  const int ga = tid / C::PA, pa = tid - ga * C::PA;             // input: float4 index inside the TH x (TW+8) tile
  const int aty = pa / C::ROWP4, atx4 = pa - aty * C::ROWP4;
  const bool owner_a = ga < C::GA;
  const int aloff = ga * C::PLA + aty * C::ROWP + atx4 * 4;
  const int64_t dstride = (int64_t)C::GD * HW, astride = (int64_t)C::GA * HW;

  v4f acc[C::PP][C::KK];
  v4f accb[C::PP];
#pragma unroll
  for (int j = 0; j < C::PP; ++j) {
    accb[j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < C::KK; ++t) acc[j][t] = v4f{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_db = (ib == 0) && (p.part_db != nullptr);
  const int it0 = (int)((int64_t)split * p.items / p.nsplit), it1 = (int)((int64_t)(split + 1) * p.items / p.nsplit);

  float4 prd[C::ND], pra[C::NA];
  uint32_t prm[C::NA];
  float prc[C::NA];
  bool pr_aok = false, pr_dok = false;   // the prefetched input / dy position lies inside the image
  int nx_tx, nx_ty, nx_n;   // tile the next issue() fetches
  {
    int q = it0;
    nx_tx = q % p.tiles_x;
    q /= p.tiles_x;
    nx_ty = q % p.tiles_y;
    nx_n = q / p.tiles_y;
  }

  // channel ga + i * GA of the block exists (the last group of a ragged split may not)
  auto a_has = [&](int i) { return (i + 1) * C::GA <= IB || ga + i * C::GA < IB; };
  auto issue = [&]() __attribute__((always_inline)) {
    const int n = nx_n, y0 = nx_ty * TH, x0 = nx_tx * TW;
    if (++nx_tx == p.tiles_x) {
      nx_tx = 0;
      if (++nx_ty == p.tiles_y) nx_ty = 0, ++nx_n;
    }
    const float* dyb = p.dy + n * p.dy_bs + (int64_t)co0 * HW;
    const int dgy = y0 + dty - C::P;
    pr_dok = owner_d && dgy >= 0 && dgy < H;
    const uint32_t tdoff = pr_dok ? (uint32_t)(tdconst + y0 * W + x0) : 0u;
#pragma unroll
    for (int i = 0; i < C::ND; ++i)   // a ragged last group re-reads a valid channel; commit() skips it
      prd[i] = *reinterpret_cast<const float4*>(dyb + ((i + 1) * C::GD <= CB || gd + i * C::GD < CB ? i * dstride : 0) + tdoff);
    const int gy = y0 + aty, gx = x0 + atx4 * 4 - C::PADL;
    pr_aok = owner_a && gx >= 0 && gx < W;
    const uint32_t taoff = pr_aok ? (uint32_t)(ga * HW + gy * W + gx) : 0u;
    const float* xb = s.x + n * s.bs + (int64_t)chb0 * HW;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) pra[i] = *reinterpret_cast<const float4*>(xb + (a_has(i) ? i * astride : 0) + taoff);
    if (has_mask) {
      const uint8_t* mb = s.emask + ((int64_t)n * s.C + chb0) * HW;
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prm[i] = *reinterpret_cast<const uint32_t*>(mb + (a_has(i) ? i * astride : 0) + taoff);
    }
    if (has_cm) {
      const float* cmb = s.cmask + (int64_t)n * s.C + chb0 + (owner_a ? ga : 0);
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prc[i] = cmb[a_has(i) ? i * C::GA : 0];
    }
  };

  auto commit = [&]() __attribute__((always_inline)) {
    if (owner_d) {
#pragma unroll
      for (int i = 0; i < C::ND; ++i) {
        if (!((i + 1) * C::GD <= CB || gd + i * C::GD < CB)) continue;
        float* dst = dy_t + i * (C::GD * C::PLD) + dloff;   // plane stride == 2 (mod 32): 8-byte aligned, not 16
        *reinterpret_cast<float2*>(dst) = pr_dok ? make_float2(prd[i].x, prd[i].y) : make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(dst + 2) = pr_dok ? make_float2(prd[i].z, prd[i].w) : make_float2(0.f, 0.f);
      }
    }
    if (owner_a) {
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        if (!a_has(i)) continue;
        wsl_v2f lo = {pra[i].x, pra[i].y}, hi = {pra[i].z, pra[i].w};
        if (has_scale) {
          const float2 t = tab[ga + i * C::GA];
          xform_bn_leaky(lo, hi, t.x, t.y);
        }
        if (has_mask) xform_mask(lo, hi, prm[i], es);   // keep-mask bytes are 0 or 1
        if (has_cm) lo = lo * prc[i], hi = hi * prc[i];
        if (!pr_aok) lo = wsl_v2f{0.f, 0.f}, hi = wsl_v2f{0.f, 0.f};
        float* dst = a_t + i * (C::GA * C::PLA) + aloff;
        *reinterpret_cast<float2*>(dst) = make_float2(lo[0], lo[1]);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(hi[0], hi[1]);
      }
    }
  };

  if (it0 < it1) issue();
  __syncthreads();  // BN table visible
  const int cot = (wp_ / C::IBT) * C::PP, cit = wp_ % C::IBT;   // this wave: output-channel tiles cot .. cot+PP-1, one ci tile
  const float* dyp = dy_t + (cot * 16 + (lane & 15)) * C::PLD + (lane >> 4);
  const float* ap = a_t + (cit * 16 + (lane & 15)) * C::PLA + (lane >> 4) + (C::PADL - C::P);
  const bool dbw = want_db && cit == 0;
  for (int item = it0; item < it1; ++item) {
    if (!WSL_ABLATED(p, 2) || item == it0) commit();
    __syncthreads();
    if (item + 1 < it1 && !WSL_ABLATED(p, 2)) issue();   // prefetch the next tile; in flight during the MFMA loop
    constexpr int RW = TH / WK, NX = TW / 4, NSTEP = RW * NX;
    float avv[2][C::PP][KS], bvv[2][KS];
    auto load = [&](int st, int buf) {   // step st = (row, group of 4 pixels)
      const int r = wk * RW + st / NX, x4 = st % NX;
#pragma unroll
      for (int j = 0; j < C::PP; ++j)
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)   // dy shifted by -(ky-P) rows
          avv[buf][j][ky] = dyp[j * 16 * C::PLD + (r + 2 * C::P - ky) * TW + x4 * 4];
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) bvv[buf][kx] = ap[r * C::ROWP + x4 * 4 + kx];           // a shifted by +(kx-P) cols
    };
    load(0, 0);
    if (WSL_ABLATED(p, 1)) continue;
#pragma unroll 2
    for (int st = 0; st < NSTEP; ++st) {   // operands of step st+1 are read before the MFMAs of step st issue
      const int cur = st & 1;
      if (st + 1 < NSTEP && !WSL_ABLATED(p, 8)) load(st + 1, cur ^ 1);
#pragma unroll
      for (int j = 0; j < C::PP; ++j) {
        if (dbw) accb[j] = WSL_MFMA16(avv[cur][j][C::P], 1.0f, accb[j]);
#pragma unroll
        for (int t = 0; t < C::KK; ++t) acc[j][t] = WSL_MFMA16(avv[cur][j][t / KS], bvv[cur][t % KS], acc[j][t]);
      }
      WSL_SCHED_BARRIER();
    }
    __syncthreads();
  }
  // ---- merge the WK row-groups (fixed order) and store partials
  if (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);
    constexpr int PER = (C::KK + 1) * 4;
    float* mine = red + (wave * 64 + lane) * PER;
#pragma unroll
    for (int t = 0; t < C::KK; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[t * 4 + r] = acc[0][t][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[C::KK * 4 + r] = accb[0][r];
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int t = 0; t <= C::KK; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = 0.f;
          for (int k = 0; k < WK; ++k) sum += red[((k * C::WP + wp_) * 64 + lane) * PER + t * 4 + r];
          if (t < C::KK) acc[0][t][r] = sum; else accb[0][r] = sum;
        }
    }
  }
  if (wk == 0) {
    const int ci = ci0 + cit * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < C::PP; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (cot + j) * 16 + (lane >> 4) * 4 + r;
#pragma unroll
        for (int t = 0; t < C::KK; ++t) p.part_dw[(((int64_t)split * C::KK + t) * p.Co + co) * Ci + ci] = acc[j][t][r];
        if (dbw && (lane & 15) == 0) p.part_db[(int64_t)split * p.Co + co] = accb[j][r];
      }
  }
}

template <int KS, int TH, int TW, int CB, int IB, int WK>
static int launch_wgrad2s(Wgrad2P& p, int ci_blocks, void* stream) {
  using C = WgradSCfg<KS, TH, TW, CB, IB, WK>;
  auto kern = wgrad_mfma2s_kernel<KS, TH, TW, CB, IB, WK>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.co_blocks * ci_blocks, p.nsplit);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(PF_WGRAD_DIRECT, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_mfma2s_kernel");
}

template <int KS, int TH, int TW, int CB, int IB, int WK>
static int launch_wgrad2l(Wgrad2P& p, int ci_blocks, void* stream) {
  using C = Wgrad2Cfg<KS, TH, TW, CB, IB, WK>;
  auto kern = wgrad_mfma2l_kernel<KS, TH, TW, CB, IB, WK>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.co_blocks * ci_blocks, p.nsplit);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(PF_WGRAD_DIRECT, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_mfma2l_kernel");
}

template <int KS, int TH, int TW, int CB, int IB, int WK>
static int launch_wgrad2(Wgrad2P& p, int ci_blocks, void* stream) {
  using C = Wgrad2Cfg<KS, TH, TW, CB, IB, WK>;
  static const bool lean_on = (WSL_TUNE("WSL_CONV_LEAN", 1) != 0);
  const int64_t span = (int64_t)(p.a.C > p.b.C ? p.a.C : p.b.C) * p.H * p.W;
  if (lean_on && p.Ci % IB == 0 && p.Co % CB == 0 && (p.b.C == 0 || p.a.C % IB == 0) && p.H % TH == 0 && p.W % TW == 0 &&
      span < (int64_t(1) << 31) && (int64_t)p.Co * p.H * p.W < (int64_t(1) << 31)) {
    static const bool split_on = (WSL_TUNE("WSL_WGRAD_SPLIT", 1) != 0);
    if (split_on) return launch_wgrad2s<KS, TH, TW, CB, IB, WK>(p, ci_blocks, stream);
    return launch_wgrad2l<KS, TH, TW, CB, IB, WK>(p, ci_blocks, stream);
  }
  auto kern = wgrad_mfma2_kernel<KS, TH, TW, CB, IB, WK>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.co_blocks * ci_blocks, p.nsplit);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(PF_WGRAD_DIRECT, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_mfma2_kernel");
}

bool wgrad2_eligible(const WslSrc& a, const WslSrc* b, const float* dy, int64_t dy_bs, int W) {
  return conv2_eligible(a, b, nullptr, 0, W, 0) && aligned16(dy) && !(dy_bs & 3);
}

// 64 output channels x 32 input channels per workgroup (wgrad_mfma2s_kernel with two co tiles per wave): non-small layers
// with Co % 64 == 0 that satisfy the lean contract.  WSL_WGRAD_CB64=0 keeps the 32 x 32 blocking.
bool wgrad2s_wide_ok(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks) {
  static const bool on = WSL_TUNE("WSL_WGRAD_CB64", 0) != 0;
  static const bool lean_on = (WSL_TUNE("WSL_CONV_LEAN", 1) != 0);
  static const bool split_on = (WSL_TUNE("WSL_WGRAD_SPLIT", 1) != 0);
  if (!on || !lean_on || !split_on) return false;
  const int bC = b ? b->C : 0, Ci = a.C + bC;
  if (Co <= 16 || Ci <= 16 || (Co % 64) || (Ci % 32) || (bC > 0 && (a.C % 32))) return false;
  const int th = W >= 32 ? 4 : 8, tw = W >= 32 ? 32 : 16;
  if ((H % th) || (W % tw) || (ks != 1 && ks != 3)) return false;
  const int64_t span = (int64_t)(a.C > bC ? a.C : bC) * H * W;
  return span < (int64_t(1) << 31) && (int64_t)Co * H * W < (int64_t(1) << 31);
}

bool wgrad_wino_ok(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks, int th, int tw, int cb, int ib);
int wgrad_wino_launch(const WslSrc& a, const WslSrc* b, const float* dy, int64_t dy_bs, float* part_dw, float* part_db, int N,
                      int H, int W, int Co, int th, int tw, int cb, int nsplit, int items, int tiles_x, int tiles_y,
                      int co_blocks, int ci_blocks, void* stream);

int wgrad2_launch(const WslSrc& a, const WslSrc* b, const float* dy, int64_t dy_bs, float* part_dw, float* part_db, int N,
                  int H, int W, int Co, int ks, int th, int tw, int cb, int ib, int nsplit, int items, int tiles_x,
                  int tiles_y, int co_blocks, int ci_blocks, void* stream) {
  Wgrad2P p;
  p.a = to_src2(a);
  p.b = (b && b->C > 0) ? to_src2(*b) : Src2{};
  p.dy = dy, p.dy_bs = dy_bs, p.part_dw = part_dw, p.part_db = part_db;
  p.N = N, p.H = H, p.W = W, p.Ci = a.C + p.b.C, p.Co = Co;
  p.tiles_x = tiles_x, p.tiles_y = tiles_y, p.items = items, p.nsplit = nsplit, p.co_blocks = co_blocks;
  static const int ablate = WSL_TUNE("WSL_WGRAD_ABLATE", 0);
  p.ablate = ablate;
  static const bool lean_on = (WSL_TUNE("WSL_CONV_LEAN", 1) != 0);
  if (lean_on && !ablate && wgrad_wino_ok(a, b, H, W, Co, ks, th, tw, cb, ib))   // Winograd form (wsl_conv5.hip)
    return wgrad_wino_launch(a, b, dy, dy_bs, part_dw, part_db, N, H, W, Co, th, tw, cb, nsplit, items, tiles_x, tiles_y,
                             co_blocks, ci_blocks, stream);
  if (cb == 64 && ib == 32) {   // two output-channel tiles per wave: only the split-halo kernel is built for this blocking
    if (ks == 3 && th == 4 && tw == 32) return launch_wgrad2s<3, 4, 32, 64, 32, 1>(p, ci_blocks, stream);
    if (ks == 3 && th == 8 && tw == 16) return launch_wgrad2s<3, 8, 16, 64, 32, 1>(p, ci_blocks, stream);
    if (ks == 1 && th == 4 && tw == 32) return launch_wgrad2s<1, 4, 32, 64, 32, 1>(p, ci_blocks, stream);
    if (ks == 1 && th == 8 && tw == 16) return launch_wgrad2s<1, 8, 16, 64, 32, 1>(p, ci_blocks, stream);
  }
#define WSL_CASE(KS_, TH_, TW_, CB_, IB_, WK_) \
  if (ks == KS_ && th == TH_ && tw == TW_ && cb == CB_ && ib == IB_)  \
    return launch_wgrad2<KS_, TH_, TW_, CB_, IB_, WK_>(p, ci_blocks, stream);
  WSL_CASE(3, 4, 64, 16, 16, 4) WSL_CASE(3, 4, 32, 16, 16, 4) WSL_CASE(3, 8, 16, 16, 16, 4)
  WSL_CASE(3, 4, 32, 32, 32, 1) WSL_CASE(3, 8, 16, 32, 32, 1)
  WSL_CASE(1, 4, 64, 16, 16, 4) WSL_CASE(1, 4, 32, 16, 16, 4) WSL_CASE(1, 8, 16, 16, 16, 4)
  WSL_CASE(1, 4, 32, 32, 32, 1) WSL_CASE(1, 8, 16, 32, 32, 1)
#undef WSL_CASE
  set_error("wgrad2: no kernel for ks %d tile %dx%d cb %d ib %d", ks, th, tw, cb, ib);
  return WSL_EUNSUPPORTED;
}

}  // namespace wsl
