// Convolution stack of the UNet (ref: networks/unet.py:19,23 Conv2d 3x3 p1; :55 Conv2d 1x1; :120 out_conv) as
// fp32-MFMA implicit GEMMs for gfx950.
//
//   forward / data-gradient:  M = 16 consecutive pixels of a row, N = 16 output channels, K = 4 input channels of
//       one filter tap per v_mfma_f32_16x16x4_f32.  A workgroup (4 waves) owns a TH x TW pixel tile x CO_T output
//       channels; the input tile (with halo) of KC channels and the matching weights are staged through LDS, the
//       producer's BatchNorm-apply + LeakyReLU + dropout being applied while staging (nothing normalised is ever
//       written to HBM).  LDS planes are padded to 16 (mod 32) words so the four k-groups of an A-operand read hit
//       disjoint banks.  The epilogue adds the bias, stores float4 rows and emits per-block (sum, M2) per channel
//       for the following BatchNorm (Chan merge in wsl_bn.hip) -- no atomics.
//   weight-gradient:  M = 16 output channels, N = 16 input channels, K = 4 pixels; nine accumulators (one per
//       tap) per 16x16 channel pair; split over pixels into partials, second stage order-fixed.
//       db comes from one extra MFMA against a ones operand.
#include <stdio.h>
#include <stdlib.h>

#include "wsl_rt.h"

namespace wsl {

// ------------------------------------------------------------------------------------------------ staging
struct TileSrc {
  WslSrc a, b;
  int H, W, Ci;
};

// value of virtual-input channel `c` (over cat(a,b)) at (n, gy, gx); zero outside the image / channel range.
__device__ __forceinline__ float tile_value(const TileSrc& t, int n, int c, int gy, int gx) {
  if (c >= t.Ci || gy < 0 || gy >= t.H || gx < 0 || gx >= t.W) return 0.f;
  const int64_t hw = (int64_t)t.H * t.W;
  const int64_t off = (int64_t)gy * t.W + gx;
  if (c < t.a.C) return src_value(t.a, n, c, n * t.a.bs + c * hw + off, ((int64_t)n * t.a.C + c) * hw + off);
  c -= t.a.C;
  return src_value(t.b, n, c, n * t.b.bs + c * hw + off, ((int64_t)n * t.b.C + c) * hw + off);
}

// ------------------------------------------------------------------------------------------------ forward
struct ConvP {
  TileSrc in;
  const float* w;
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, Co, wmode, tiles_x, tiles_y, vec_ok, slots;
  float* stat_part;
  float* stat_cnt;
};

template <int KS, int TH, int TW, int CO_T, int KC>
struct ConvCfg {
  static constexpr int P = KS / 2, KK = KS * KS;
  static constexpr int ROWP = TW + 2 * P, ROWS = TH + 2 * P, TILE = ROWS * ROWP;
  static constexpr int PLANE = ((TILE - 16 + 31) / 32) * 32 + 16;  // >= TILE, == 16 (mod 32)
  static constexpr int CSTR = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;  // == 16 (mod 32)
  static constexpr int SEGS = TW / 16, MT_TOTAL = TH * SEGS, MT = MT_TOTAL / 4, NT = CO_T / 16;
  static constexpr int IN_FLOATS = KC * PLANE, W_FLOATS = KK * KC * CSTR;
  static constexpr int RED_FLOATS = 8 * CO_T;
  static constexpr size_t SMEM = sizeof(float) * (IN_FLOATS + W_FLOATS > RED_FLOATS ? IN_FLOATS + W_FLOATS : RED_FLOATS);
  static_assert(MT_TOTAL % 4 == 0 && TW % 16 == 0 && CO_T % 16 == 0 && KC % 4 == 0, "tile shape");
};

template <int KS, int TH, int TW, int CO_T, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvP p) {
  using C = ConvCfg<KS, TH, TW, CO_T, KC>;
  WSL_DYN_SMEM(smem);
  float* in_t = reinterpret_cast<float*>(smem);
  float* w_t = in_t + C::IN_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int co0 = blockIdx.y * CO_T;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = p.in.H, W = p.in.W, Ci = p.in.Ci;

  v4f acc[C::MT][C::NT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i)
#pragma unroll
    for (int j = 0; j < C::NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  int abase[C::MT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i) {
    const int mt = wave * C::MT + i;
    abase[i] = (lane >> 4) * C::PLANE + (mt / C::SEGS) * C::ROWP + (mt % C::SEGS) * 16 + (lane & 15);
  }
  const int bbase = (lane >> 4) * C::CSTR + (lane & 15);

  for (int c0 = 0; c0 < Ci; c0 += KC) {
    // ---- stage the input tile (+halo) of channels [c0, c0+KC), transformed, zero padded
    for (int e = tid; e < KC * C::TILE; e += kThreads) {
      const int c = e / C::TILE, rem = e - c * C::TILE;
      const int ty = rem / C::ROWP, tx = rem - ty * C::ROWP;
      in_t[c * C::PLANE + rem] = tile_value(p.in, n, c0 + c, y0 + ty - C::P, x0 + tx - C::P);
    }
    // ---- stage the weights of this channel chunk as w_t[tap][c][co]
    for (int e = tid; e < CO_T * KC * C::KK; e += kThreads) {
      const int co = e / (KC * C::KK), rem = e - co * (KC * C::KK);
      const int c = rem / C::KK, tap = rem - c * C::KK;
      const int cog = co0 + co, cg = c0 + c;
      float v = 0.f;
      if (cog < p.Co && cg < Ci)
        v = p.wmode == 0 ? p.w[((int64_t)cog * Ci + cg) * C::KK + tap]
                         : p.w[((int64_t)cg * p.Co + cog) * C::KK + (C::KK - 1 - tap)];
      w_t[(tap * KC + c) * C::CSTR + co] = v;
    }
    __syncthreads();
    const int ngroups = (Ci - c0 >= KC) ? KC / 4 : (Ci - c0 + 3) / 4;
#pragma unroll
    for (int tap = 0; tap < C::KK; ++tap) {
      const int ky = tap / KS, kx = tap % KS;
#pragma unroll
      for (int cg = 0; cg < KC / 4; ++cg) {
        if (cg < ngroups) {
          float bv[C::NT];
#pragma unroll
          for (int j = 0; j < C::NT; ++j) bv[j] = w_t[(tap * KC + cg * 4) * C::CSTR + j * 16 + bbase];
#pragma unroll
          for (int i = 0; i < C::MT; ++i) {
            const float av = in_t[cg * 4 * C::PLANE + ky * C::ROWP + kx + abase[i]];
#pragma unroll
            for (int j = 0; j < C::NT; ++j) acc[i][j] = WSL_MFMA16(av, bv[j], acc[i][j]);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, store, BatchNorm partial statistics
  const int64_t HW = (int64_t)H * W;
  float bsum[C::NT];
#pragma unroll
  for (int j = 0; j < C::NT; ++j) {
    const int co = co0 + j * 16 + (lane & 15);
    const float bias = (p.bias && co < p.Co) ? p.bias[co] : 0.f;
    bsum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < C::MT; ++i) {
      const int mt = wave * C::MT + i;
      const int gy = y0 + mt / C::SEGS, gx = x0 + (mt % C::SEGS) * 16 + (lane >> 4) * 4;
      v4f v = acc[i][j];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bias;
      acc[i][j] = v;
      if (co < p.Co && gy < H) {
        float* dst = p.y + n * p.y_bs + co * HW + (int64_t)gy * W + gx;
        if (p.vec_ok && gx + 3 < W) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (gx + r < W) dst[r] = v[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gx + r < W) bsum[j] += v[r];
      }
    }
  }
  if (p.stat_part) {  // uniform branch
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
    float m2[C::NT];
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
      const int col = j * 16 + (lane & 15);
      const int co = co0 + col;
      const float mean_b = (red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col]) / cnt;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < C::MT; ++i) {
        const int mt = wave * C::MT + i;
        const int gy = y0 + mt / C::SEGS, gx = x0 + (mt % C::SEGS) * 16 + (lane >> 4) * 4;
        if (co < p.Co && gy < H) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (gx + r < W) {
              const float d = acc[i][j][r] - mean_b;
              q = fmaf(d, d, q);
            }
        }
      }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      m2[j] = q;
      if (lane < 16) red2[wave * CO_T + j * 16 + lane] = q;
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        const int col = j * 16 + lane, co = co0 + col;
        if (co < p.Co) {
          const int64_t nsl = (int64_t)gridDim.x * p.slots;                                   // [Co][slots][2]
          float* dst = p.stat_part + ((int64_t)co * nsl + (int64_t)blockIdx.x * p.slots) * 2;   // slot 0 of this tile's slots
          dst[0] = red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col];
          dst[1] = red2[col] + red2[CO_T + col] + red2[2 * CO_T + col] + red2[3 * CO_T + col];
        }
      }
      if (lane < p.slots && blockIdx.y == 0) p.stat_cnt[blockIdx.x * p.slots + lane] = lane == 0 ? cnt : 0.f;
    }
    (void)m2;
  }
}

struct FwdPlan {
  int th, tw, co_t;
  bool wino;   // the layer has a Winograd shape and the Winograd kernels are enabled
};
// Tile shape of one layer.  Table from tools/sweep_conv_plans.py on MI355X (batch 64; profiles/r1g_conv_timeline.md):
// 64-pixel-wide tiles only pay for <= 16 output channels; wider layers take 8x32 pixels x as many channels as one
// workgroup can hold (halves the staging per MFMA); at 16x16 resolution the channel block shrinks until the launch
// has >= 2 workgroups per CU (a 128-workgroup launch leaves half the chip idle).
bool wino_shape_ok(int H, int W, int Ci, int Co, int ks, bool allow16, int* th, int* tw, int* co_t);   // wsl_conv5.hip
// Winograd F(2x2,3x3) for the 3x3 layers it fits (wsl_conv5.hip): 0 off, 1 only layers with Co % 32 == 0, 2 also Co % 16 == 0 (the
// product's fixed setting).
#define WSL_WINO_DEFAULT 2
#ifdef WSL_EXPERIMENTS
// Routing overrides of the EXPERIMENTS build only (tools/exp/libwslhip_exp.so and the test-only host emulator): the product library has
// no routing state at all -- its plan of a launch is a pure function of the launch's shape.  Atomics: the tuning
// tools set them from the host thread between launches.
static std::atomic<int> g_forced_plan[3] = {{-1}, {0}, {0}};   // th, tw, co_t; th < 0: environment not read yet, 0: none
static std::atomic<int> g_wino{-1};                            // env WSL_CONV_WINO / wsl_debug_conv_wino()
static int wino_mode() {
  int w = g_wino.load();
  if (w < 0) {
    w = WSL_TUNE("WSL_CONV_WINO", WSL_WINO_DEFAULT);
    if (w < 0 || w > 2) w = WSL_WINO_DEFAULT;
    g_wino.store(w);
  }
  return w;
}
#else
static constexpr int wino_mode() { return WSL_WINO_DEFAULT; }
#endif
static bool wino_enabled() { return wino_mode() != 0; }
static FwdPlan fwd_plan(int N, int H, int W, int Co, int Ci, int ks) {
  FwdPlan f;
#ifdef WSL_EXPERIMENTS
  // tuning / test aid (experiments build only): WSL_CONV_PLAN=th,tw,co_t or wsl_debug_conv_plan() force one tile shape wherever it divides the layer
  if (g_forced_plan[0].load() < 0) {
    const char* e = getenv("WSL_CONV_PLAN");
    int th = 0, tw = 0, ct = 0;
    if (e && sscanf(e, "%d,%d,%d", &th, &tw, &ct) == 3 && th > 0) g_forced_plan[1] = tw, g_forced_plan[2] = ct, g_forced_plan[0] = th;
    else g_forced_plan[0] = 0;
  }
  if (g_forced_plan[0].load() > 0) {
    const int th = g_forced_plan[0], tw = g_forced_plan[1], ct = g_forced_plan[2];
    if (tw > 0 && W % tw == 0 && ct > 0 && Co % ct == 0) {
      f.th = th, f.tw = tw, f.co_t = ct, f.wino = false;
      return f;
    }
  }
#endif
  f.wino = false;
  if (Co <= 16) {
    f.co_t = 16;
    if (W >= 64) f.th = 8, f.tw = 64; else if (W >= 32) f.th = 8, f.tw = 32; else f.th = 16, f.tw = 16;
    // a Winograd-shaped layer keeps the Winograd tile in the direct kernels too: one BatchNorm-partial count per layer
    if (wino_enabled() && wino_shape_ok(H, W, Ci, Co, ks, wino_mode() == 2, &f.th, &f.tw, nullptr)) f.wino = true;
    return f;
  }
  if (W >= 32) f.th = 8, f.tw = 32; else f.th = 16, f.tw = 16;
  if (wino_enabled() && wino_shape_ok(H, W, Ci, Co, ks, wino_mode() == 2, &f.th, &f.tw, nullptr)) f.wino = true;
  const int64_t tiles = (int64_t)N * cdiv(H, f.th) * cdiv(W, f.tw);
  const int64_t enough = 2 * (int64_t)device_cu_count();
  f.co_t = Co <= 32 ? 32 : 64;
  while (f.co_t > 16 && tiles * cdiv(Co, f.co_t) < enough) f.co_t >>= 1;
  return f;
}

template <int KS, int TH, int TW, int CO_T>
static int launch_conv(ConvP& p, void* stream) {
  using C = ConvCfg<KS, TH, TW, CO_T, 8>;
  auto kern = conv_mfma_kernel<KS, TH, TW, CO_T, 8>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, cdiv(p.Co, CO_T));
  const double px = (double)p.N * p.in.H * p.in.W;
  void* tok = prof_begin(p.wmode ? 1 : 0, 2.0 * px * p.Co * p.in.Ci * KS * KS, 4.0 * px * (p.Co + p.in.Ci), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_mfma_kernel");
}

template <int KS>
static int dispatch_conv(ConvP& p, const FwdPlan& f, void* stream) {
#define WSL_CASE(TH_, TW_, CO_) \
  if (f.th == TH_ && f.tw == TW_ && f.co_t == CO_) return launch_conv<KS, TH_, TW_, CO_>(p, stream);
  WSL_CASE(8, 64, 16)
  WSL_CASE(8, 64, 32)
  WSL_CASE(8, 32, 16)
  WSL_CASE(8, 32, 32)
  WSL_CASE(8, 32, 64)
  WSL_CASE(16, 16, 16)
  WSL_CASE(16, 16, 32)
  WSL_CASE(16, 16, 64)
#undef WSL_CASE
  set_error("conv: no kernel for tile %dx%d co_t %d", f.th, f.tw, f.co_t);
  return WSL_EUNSUPPORTED;
}

static int check_src(const WslSrc* s, int HW, const char* who) {
  WSL_REQUIRE(s->x != nullptr && s->C > 0, "%s: source has no data", who);
  WSL_REQUIRE(s->bs >= (int64_t)s->C * HW, "%s: batch stride %lld < C*H*W", who, (long long)s->bs);
  WSL_REQUIRE((s->scale == nullptr) == (s->shift == nullptr), "%s: scale and shift must come together", who);
  return WSL_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgradP {
  TileSrc in;
  const float* dy;
  int64_t dy_bs;
  float* part_dw;  // [nsplit][KK][Co][Ci]
  float* part_db;  // [nsplit][Co]
  int N, Co, tiles_x, tiles_y, items, nsplit, co_blocks;
};

template <int KS, int TH, int TW, int CB, int IB, int WK>
struct WgradCfg {
  static constexpr int P = KS / 2, KK = KS * KS;
  static constexpr int ROWP = TW + 2 * P, ROWS = TH + 2 * P, S = TH * TW;
  static constexpr int PLD = ((S - 2 + 31) / 32) * 32 + 2;             // == 2 (mod 32)
  static constexpr int PLA = ((ROWS * ROWP - 2 + 31) / 32) * 32 + 2;   // == 2 (mod 32)
  static constexpr int CBT = CB / 16, IBT = IB / 16, PAIRS = CBT * IBT, WP = 4 / WK, PP = PAIRS / WP;
  static constexpr int DY_FLOATS = CB * PLD, A_FLOATS = IB * PLA;
  static constexpr int RED_FLOATS = (WK > 1) ? 4 * 64 * (PP * (KK + 1) * 4) : 0;
  static constexpr size_t SMEM =
      sizeof(float) * (DY_FLOATS + A_FLOATS > RED_FLOATS ? DY_FLOATS + A_FLOATS : RED_FLOATS);
  static_assert(PAIRS % WP == 0 && TH % WK == 0 && TW % 4 == 0, "wgrad tile shape");
};

template <int KS, int TH, int TW, int CB, int IB, int WK>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradP p) {
  using C = WgradCfg<KS, TH, TW, CB, IB, WK>;
  WSL_DYN_SMEM(smem);
  float* dy_t = reinterpret_cast<float*>(smem);
  float* a_t = dy_t + C::DY_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x % p.co_blocks, ib = blockIdx.x / p.co_blocks, split = blockIdx.y;
  const int co0 = cb * CB, ci0 = ib * IB;
  const int wp = wave % C::WP, wk = wave / C::WP;
  const int H = p.in.H, W = p.in.W, Ci = p.in.Ci;
  const int64_t HW = (int64_t)H * W;

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
  for (int item = it0; item < it1; ++item) {
    int q = item;
    const int tx_i = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty_i = q % p.tiles_y;
    const int n = q / p.tiles_y;
    const int y0 = ty_i * TH, x0 = tx_i * TW;
    // ---- stage dy tile [CB][S] (zero outside image / channel range)
    for (int e = tid; e < CB * C::S; e += kThreads) {
      const int c = e / C::S, rem = e - c * C::S;
      const int ty = rem / TW, tx = rem - ty * TW;
      const int gy = y0 + ty, gx = x0 + tx, co = co0 + c;
      float v = 0.f;
      if (co < p.Co && gy < H && gx < W) v = p.dy[n * p.dy_bs + co * HW + (int64_t)gy * W + gx];
      dy_t[c * C::PLD + rem] = v;
    }
    // ---- stage input tile [IB][ROWS*ROWP] with halo, transformed
    for (int e = tid; e < IB * C::ROWS * C::ROWP; e += kThreads) {
      const int c = e / (C::ROWS * C::ROWP), rem = e - c * (C::ROWS * C::ROWP);
      const int ty = rem / C::ROWP, tx = rem - ty * C::ROWP;
      a_t[c * C::PLA + rem] = tile_value(p.in, n, ci0 + c, y0 + ty - C::P, x0 + tx - C::P);
    }
    __syncthreads();
    constexpr int RW = TH / WK;
#pragma unroll 1
    for (int r = wk * RW; r < wk * RW + RW; ++r) {
#pragma unroll 2
      for (int x4 = 0; x4 < TW / 4; ++x4) {
        const int pix = r * TW + x4 * 4 + (lane >> 4);
        const int apix = r * C::ROWP + x4 * 4 + (lane >> 4);
#pragma unroll
        for (int j = 0; j < C::PP; ++j) {
          const int pr = wp * C::PP + j, cot = pr / C::IBT, cit = pr % C::IBT;
          const float av = dy_t[(cot * 16 + (lane & 15)) * C::PLD + pix];
          if (want_db && cit == 0) accb[j] = WSL_MFMA16(av, 1.0f, accb[j]);
#pragma unroll
          for (int t = 0; t < C::KK; ++t) {
            const float bv = a_t[(cit * 16 + (lane & 15)) * C::PLA + apix + (t / KS) * C::ROWP + (t % KS)];
            acc[j][t] = WSL_MFMA16(av, bv, acc[j][t]);
          }
        }
      }
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
            float s = 0.f;
            for (int k = 0; k < WK; ++k) s += red[((k * C::WP + wp) * 64 + lane) * PER + (j * (C::KK + 1) + t) * 4 + r];
            if (t < C::KK) acc[j][t][r] = s; else accb[j][r] = s;
          }
      }
    }
  }
  if (wk == 0) {
#pragma unroll
    for (int j = 0; j < C::PP; ++j) {
      const int pr = wp * C::PP + j, cot = pr / C::IBT, cit = pr % C::IBT;
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

// second stage: dw[co][ci][tap] = sum_s part[s][tap][co][ci]; db[co] = sum_s part_db[s][co].  One thread per
// (element, s-group); s-groups merged through LDS in fixed order.
template <int SG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part_dw, const float* part_db, float* dw,
                                                           float* db, int Co, int Ci, int KK, int nsplit) {
  __shared__ float red[kThreads];
  constexpr int EPB = kThreads / SG;  // elements per block
  const int64_t E = (int64_t)KK * Co * Ci;
  const int64_t total = E + (db ? Co : 0);
  const int el = threadIdx.x % EPB, sg = threadIdx.x / EPB;
  const int64_t e = (int64_t)blockIdx.x * EPB + el;
  float s = 0.f;
  if (e < total) {
    const float* src = e < E ? part_dw + e : part_db + (e - E);
    const int64_t stride = e < E ? E : Co;
    for (int k = sg; k < nsplit; k += SG) s += src[k * stride];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SG; ++k) t += red[k * EPB + el];
    if (e < E) {
      const int tap = (int)(e / ((int64_t)Co * Ci));
      const int64_t rem = e - (int64_t)tap * Co * Ci;  // co*Ci + ci
      dw[rem * KK + tap] = t;
    } else {
      db[e - E] = t;
    }
  }
}

#ifdef WSL_EXPERIMENTS
static std::atomic<int> g_forced_wgrad_wgs{0};   // wsl_debug_wgrad_workgroups(): small launches walk several tiles per workgroup in the tests
int forced_wgrad_wgs() { return g_forced_wgrad_wgs.load(); }   // (declared in wsl_rt.h; the product's is a constexpr 0)
#endif
struct WgPlan {
  int th, tw, cb, ib, wk, nsplit, items, tiles_x, tiles_y, co_blocks, ci_blocks;
};
// v2 = the prefetching kernels of wsl_conv2.hip: half-height tiles keep the register-held prefetch set small
static WgPlan wgrad_plan(int N, int H, int W, int Ci, int Co, bool v2 = false, bool wide = false) {
  WgPlan g;
  const bool small = (Co <= 16 || Ci <= 16);
  if (small) {
    g.cb = g.ib = 16, g.wk = 4;
    if (W >= 64) g.th = v2 ? 4 : 8, g.tw = 64; else if (W >= 32) g.th = v2 ? 4 : 8, g.tw = 32; else g.th = v2 ? 8 : 16, g.tw = 16;
  } else {
    g.cb = g.ib = 32, g.wk = 1;
    if (wide) g.cb = 64;   // two output-channel tiles per wave (split-halo kernel only): 2 resident workgroups per CU
    if (W >= 32) g.th = v2 ? 4 : 8, g.tw = 32; else g.th = v2 ? 8 : 16, g.tw = 16;
  }
  g.tiles_x = cdiv(W, g.tw), g.tiles_y = cdiv(H, g.th);
  g.items = N * g.tiles_x * g.tiles_y;
  g.co_blocks = cdiv(Co, g.cb), g.ci_blocks = cdiv(Ci, g.ib);
  static const int wgs = WSL_TUNE("WSL_WGRAD_WGS", 768);   // 3 resident workgroups x 256 CUs
  int want = (forced_wgrad_wgs() > 0 ? forced_wgrad_wgs() : wide ? 512 : wgs) / (g.co_blocks * g.ci_blocks);
  if (want < 1) want = 1;
  g.nsplit = g.items < want ? g.items : want;
  return g;
}

template <int KS, int TH, int TW, int CB, int IB, int WK>
static int launch_wgrad(WgradP& p, const WgPlan& g, void* stream) {
  using C = WgradCfg<KS, TH, TW, CB, IB, WK>;
  auto kern = wgrad_mfma_kernel<KS, TH, TW, CB, IB, WK>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(g.co_blocks * g.ci_blocks, g.nsplit);
  const double px = (double)p.N * p.in.H * p.in.W;
  void* tok = prof_begin(PF_WGRAD_DIRECT, 2.0 * px * p.Co * p.in.Ci * KS * KS, 4.0 * px * (p.Co + p.in.Ci), stream);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_mfma_kernel");
}

template <int KS>
static int dispatch_wgrad(WgradP& p, const WgPlan& g, void* stream) {
#define WSL_CASE(TH_, TW_, CB_, WK_) \
  if (g.th == TH_ && g.tw == TW_ && g.cb == CB_) return launch_wgrad<KS, TH_, TW_, CB_, CB_, WK_>(p, g, stream);
  WSL_CASE(8, 64, 16, 4)
  WSL_CASE(8, 32, 16, 4)
  WSL_CASE(16, 16, 16, 4)
  WSL_CASE(8, 32, 32, 1)
  WSL_CASE(16, 16, 32, 1)
#undef WSL_CASE
  set_error("wgrad: no kernel for tile %dx%d cb %d", g.th, g.tw, g.cb);
  return WSL_EUNSUPPORTED;
}

}  // namespace wsl

namespace wsl {
// wsl_conv2.hip
bool conv2_eligible(const WslSrc& a, const WslSrc* b, const float* y, int64_t y_bs, int W, int Ci);
int conv2_pack(const float* w, float* wp, int Co, int Ci, int ks, int wmode, void* stream);
int conv2_fwd(const WslSrc& a, const WslSrc* b, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H,
              int W, int Co, int ks, int is_dgrad, int th, int tw, int co_t, float* stat_part, float* stat_cnt,
              int slots, void* stream, const BnBwdEpi* bn = nullptr, int* bn_done = nullptr);
int wgrad_small_kind(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks);   // wsl_conv4.hip
int wgrad_small_launch(int kind, const WslSrc& a, const float* dy, int64_t dy_bs, float* part_dw, float* part_db, int N,
                       int H, int W, int nsplit, void* stream);
bool conv_cls_eligible(const WslSrc& a, const WslSrc* b, const float* y, int64_t y_bs, int H, int W, int Co, int ks,
                       const float* stat_part);                                              // wsl_conv4.hip
bool conv_nk16_eligible(const WslSrc& a, const WslSrc* b, const float* y, int64_t y_bs, int H, int W, int Co, int ks, int th, int tw);
int conv_nk16_launch(const WslSrc& a, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H, int W, bool dgrad,
                     float* stat_part, float* stat_cnt, int slots, const BnBwdEpi* bn, int* bn_done, void* stream);
int conv_cls_launch(const WslSrc& a, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H, int W,
                    void* stream);
int wino_pack(const float* w, float* u, int Co, int Ci, int dgrad, void* stream);   // wsl_conv5.hip
int wino_fwd(const WslSrc& a, const WslSrc* b, const float* u, const float* bias, float* y, int64_t y_bs, int N, int H,
             int W, int Co, int is_dgrad, float* stat_part, float* stat_cnt, int slots, void* stream,
             const BnBwdEpi* bn = nullptr, int* bn_done = nullptr);
bool wgrad2_eligible(const WslSrc& a, const WslSrc* b, const float* dy, int64_t dy_bs, int W);
bool wgrad2s_wide_ok(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks);
bool wgrad_wino_ok(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks, int th, int tw, int cb, int ib);   // wsl_conv5.hip
int wgrad2_launch(const WslSrc& a, const WslSrc* b, const float* dy, int64_t dy_bs, float* part_dw, float* part_db, int N,
                  int H, int W, int Co, int ks, int th, int tw, int cb, int ib, int nsplit, int items, int tiles_x,
                  int tiles_y, int co_blocks, int ci_blocks, void* stream);
}  // namespace wsl

using namespace wsl;

#ifdef WSL_EXPERIMENTS
// (private header wsl_debug.h; experiments build and host emulator only)
extern "C" int wsl_debug_conv_plan(int th, int tw, int co_t) {
  g_forced_plan[1] = tw, g_forced_plan[2] = co_t, g_forced_plan[0] = th > 0 ? th : 0;
  return WSL_OK;
}

extern "C" int wsl_debug_wgrad_workgroups(int n) {
  g_forced_wgrad_wgs = n > 0 ? n : 0;
  return WSL_OK;
}

extern "C" int wsl_debug_conv_wino(int on) {
  g_wino = on < 0 ? -1 : (on > 2 ? 2 : on);
  return WSL_OK;
}
#endif

extern "C" int wsl_conv2d_wino_ok(int N, int H, int W, int Ca, int Cb, int Co, int ks) {
  if (N <= 0 || Ca <= 0 || Cb < 0) return 0;
  if (Cb > 0 && (Ca % 8)) return 0;   // a channel chunk never straddles the two sources
  return fwd_plan(N, H, W, Co, Ca + Cb, ks).wino ? 1 : 0;
}

extern "C" int wsl_conv2d_pack_weights(const float* w, float* packed, int Co, int Ci, int ks, int wmode_raw, void* stream) {
  WSL_REQUIRE(w && packed && Co > 0 && Ci > 0 && (ks == 1 || ks == 3) && wmode_raw >= 0 && wmode_raw <= 3,
              "conv2d_pack_weights: bad args");
  if (wmode_raw >= 2) {   // Winograd image U = G g G^T, [16][Ci][Co]
    WSL_REQUIRE(ks == 3, "conv2d_pack_weights: the Winograd image exists for 3x3 filters only");
    return wino_pack(w, packed, Co, Ci, wmode_raw == 3, stream);
  }
  return conv2_pack(w, packed, Co, Ci, ks, wmode_raw, stream);
}

extern "C" int wsl_conv2d_fast_ok(const WslSrc* a, const WslSrc* b, const float* y, int64_t y_bs, int W) {
  if (!a) return 0;
  const int Ci = a->C + ((b && b->C > 0) ? b->C : 0);
  return conv2_eligible(*a, b, y, y_bs, W, Ci) ? 1 : 0;
}

extern "C" int wsl_conv2d_stat_blocks(int N, int H, int W, int Ci, int Co, int ks) {
  if (N <= 0 || H <= 0 || W <= 0 || Co <= 0) return 0;
  const FwdPlan f = fwd_plan(N, H, W, Co, Ci, ks);
  // one partial per tile; the (opt-in) wave-specialised kernel emits one per MFMA wave = four slots per tile
  return N * cdiv(H, f.th) * cdiv(W, f.tw);
}

static int conv2d_fwd_impl(const WslSrc* a, const WslSrc* b, const float* w, const float* bias, float* y, int64_t y_bs, int N,
                           int H, int W, int Co, int ks, int wmode, float* stat_part, float* stat_cnt, void* stream,
                           const BnBwdEpi* bn, int* bn_done) {
  if (bn_done) *bn_done = 0;
  WSL_REQUIRE(a && w && y, "conv2d_fwd: null argument");
  WSL_REQUIRE(N > 0 && H > 0 && W > 0 && Co > 0, "conv2d_fwd: bad shape N=%d H=%d W=%d Co=%d", N, H, W, Co);
  WSL_REQUIRE(ks == 1 || ks == 3, "conv2d_fwd: kernel size %d not built (1 and 3 are)", ks);
  WSL_REQUIRE(wmode >= 0 && wmode <= 5, "conv2d_fwd: wmode %d", wmode);
  WSL_REQUIRE((stat_part == nullptr) == (stat_cnt == nullptr), "conv2d_fwd: stat_part and stat_cnt come together");
  if (int rc = check_src(a, H * W, "conv2d_fwd(a)")) return rc;
  ConvP p;
  p.in.a = *a;
  if (b && b->C > 0) {
    if (int rc = check_src(b, H * W, "conv2d_fwd(b)")) return rc;
    p.in.b = *b;
  } else {
    p.in.b = WslSrc{};
  }
  p.in.H = H, p.in.W = W, p.in.Ci = a->C + p.in.b.C;
  WSL_REQUIRE(y_bs >= (int64_t)Co * H * W, "conv2d_fwd: y batch stride too small");
  p.w = w, p.bias = bias, p.y = y, p.y_bs = y_bs, p.N = N, p.Co = Co, p.wmode = wmode;
  p.stat_part = stat_part, p.stat_cnt = stat_cnt;
  p.slots = 1;
  const FwdPlan f = fwd_plan(N, H, W, Co, p.in.Ci, ks);
  if (wmode >= 2) {
    if (!conv2_eligible(p.in.a, &p.in.b, y, y_bs, W, p.in.Ci)) {
      set_error("conv2d_fwd: packed weights (wmode %d) need W %% 4 == 0 and 16-byte aligned tensors", wmode);
      return WSL_EINVAL;
    }
    if (wmode >= 4) {   // `w` is the Winograd image of wsl_conv2d_pack_weights(wmode_raw 2 | 3)
      WSL_REQUIRE(f.wino, "conv2d_fwd: wmode %d needs wsl_conv2d_wino_ok() != 0 for this layer", wmode);
      return wino_fwd(p.in.a, &p.in.b, w, bias, y, y_bs, N, H, W, Co, wmode == 5, stat_part, stat_cnt, p.slots, stream, bn, bn_done);
    }
    static const bool nk_on = (WSL_TUNE("WSL_CONV_NK16", 1) != 0);
    if (nk_on && conv_nk16_eligible(p.in.a, &p.in.b, y, y_bs, H, W, Co, ks, f.th, f.tw))
      return conv_nk16_launch(p.in.a, w, bias, y, y_bs, N, H, W, wmode == 3, stat_part, stat_cnt, p.slots, bn, bn_done,
                              stream);   // first conv forward / classifier data gradient (wsl_conv4.hip)
    static const bool cls_on = (WSL_TUNE("WSL_CONV_CLS", 1) != 0);
    if (cls_on && wmode == 2 && conv_cls_eligible(p.in.a, &p.in.b, y, y_bs, H, W, Co, ks, stat_part))
      return conv_cls_launch(p.in.a, w, bias, y, y_bs, N, H, W, stream);   // 4-class classifier (wsl_conv4.hip)
    return conv2_fwd(p.in.a, &p.in.b, w, bias, y, y_bs, N, H, W, Co, ks, wmode == 3, f.th, f.tw, f.co_t, stat_part,
                     stat_cnt, p.slots, stream, bn, bn_done);
  }
  p.tiles_x = cdiv(W, f.tw), p.tiles_y = cdiv(H, f.th);
  p.vec_ok = (W % 4 == 0) && (y_bs % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  return ks == 3 ? dispatch_conv<3>(p, f, stream) : dispatch_conv<1>(p, f, stream);
}

extern "C" int wsl_conv2d_fwd(const WslSrc* a, const WslSrc* b, const float* w, const float* bias, float* y,
                              int64_t y_bs, int N, int H, int W, int Co, int ks, int wmode, float* stat_part,
                              float* stat_cnt, void* stream) {
  return conv2d_fwd_impl(a, b, w, bias, y, y_bs, N, H, W, Co, ks, wmode, stat_part, stat_cnt, stream, nullptr, nullptr);
}

static int dgrad_bn_impl(const WslSrc* dy, const float* w, float* g, int64_t g_bs, int N, int H, int W, int Co, int ks, int wmode,
                         const float* bn_y, const float* bn_st, const uint8_t* bn_emask, float bn_emask_scale, float* bn_part, int* fused,
                         bool allow_d, void* stream) {
  WSL_REQUIRE(bn_y && bn_st && bn_part && fused, "conv2d_dgrad_bn: null BatchNorm argument");
  WSL_REQUIRE(wmode == 1 || wmode == 3 || wmode == 5, "conv2d_dgrad_bn: wmode %d is not a data-gradient mode", wmode);
  BnBwdEpi e;
  e.y = bn_y, e.st = bn_st, e.emask = bn_emask, e.es = bn_emask_scale, e.part = bn_part, e.store_d = allow_d;
  if (g_bs != (int64_t)Co * H * W || (W & 3) || (reinterpret_cast<uintptr_t>(bn_y) & 15) ||
      (bn_emask && (reinterpret_cast<uintptr_t>(bn_emask) & 3)))
    e.part = nullptr;        // the statistics address y through the dense index of g: same shape, float4-aligned rows
  return conv2d_fwd_impl(dy, nullptr, w, nullptr, g, g_bs, N, H, W, Co, ks, wmode, nullptr, nullptr, stream, &e, fused);
}

extern "C" int wsl_conv2d_dgrad_bn(const WslSrc* dy, const float* w, float* g, int64_t g_bs, int N, int H, int W, int Co,
                                   int ks, int wmode, const float* bn_y, const float* bn_st, const uint8_t* bn_emask,
                                   float bn_emask_scale, float* bn_part, int* fused, void* stream) {
  return dgrad_bn_impl(dy, w, g, g_bs, N, H, W, Co, ks, wmode, bn_y, bn_st, bn_emask, bn_emask_scale, bn_part, fused, false, stream);
}

extern "C" int wsl_conv2d_dgrad_bn_d(const WslSrc* dy, const float* w, float* g, int64_t g_bs, int N, int H, int W, int Co,
                                     int ks, int wmode, const float* bn_y, const float* bn_st, const uint8_t* bn_emask,
                                     float bn_emask_scale, float* bn_part, int* fused, void* stream) {
  return dgrad_bn_impl(dy, w, g, g_bs, N, H, W, Co, ks, wmode, bn_y, bn_st, bn_emask, bn_emask_scale, bn_part, fused, true, stream);
}

extern "C" size_t wsl_conv2d_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co, int ks) {
  if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  const WgPlan g = wgrad_plan(N, H, W, Ci, Co), g2 = wgrad_plan(N, H, W, Ci, Co, true), g3 = wgrad_plan(N, H, W, Ci, Co, true, true);
  int ns = g.nsplit > g2.nsplit ? g.nsplit : g2.nsplit;
  if (g3.nsplit > ns) ns = g3.nsplit;
  return sizeof(float) * (size_t)ns * ((size_t)ks * ks * Co * Ci + Co);
}

// first stage of the weight gradient: per-split partial sums into `ws`; *out describes the pending second stage
static int wgrad_stage1(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, float* dw, float* db, int N, int H, int W,
                        int Co, int ks, void* ws, size_t ws_bytes, void* stream, WslWgradPending* out) {
  WSL_REQUIRE(a && dy && dw && ws, "conv2d_wgrad: null argument");
  WSL_REQUIRE(N > 0 && H > 0 && W > 0 && Co > 0, "conv2d_wgrad: bad shape");
  WSL_REQUIRE(ks == 1 || ks == 3, "conv2d_wgrad: kernel size %d not built (1 and 3 are)", ks);
  if (int rc = check_src(a, H * W, "conv2d_wgrad(a)")) return rc;
  WgradP p;
  p.in.a = *a;
  if (b && b->C > 0) {
    if (int rc = check_src(b, H * W, "conv2d_wgrad(b)")) return rc;
    p.in.b = *b;
  } else {
    p.in.b = WslSrc{};
  }
  p.in.H = H, p.in.W = W, p.in.Ci = a->C + p.in.b.C;
  const int Ci = p.in.Ci;
  WSL_REQUIRE(dy_bs >= (int64_t)Co * H * W, "conv2d_wgrad: dy batch stride too small");
  const bool v2 = wgrad2_eligible(p.in.a, &p.in.b, dy, dy_bs, W);
  const bool wide = v2 && wgrad2s_wide_ok(p.in.a, &p.in.b, H, W, Co, ks);
  WgPlan g = wgrad_plan(N, H, W, Ci, Co, v2, wide);
  if (v2 && g.cb == 32 && wgrad_wino_ok(p.in.a, &p.in.b, H, W, Co, ks, g.th, g.tw, g.cb, g.ib)) {
    // the 32 x 32 Winograd weight gradient holds 128 accumulator registers: 2 resident workgroups per CU -> 512 persistent ones
    // (256 when it runs as one 8-wave double-buffered workgroup per CU)
    static const int wgs_env = WSL_TUNE("WSL_WGRAD_WINO_WGS", 0);
    const int wgs = wgs_env > 0 ? wgs_env : 512;
    int want = wgs / (g.co_blocks * g.ci_blocks);
    if (want < 1) want = 1;
    if (want < g.nsplit) g.nsplit = want;   // never more partials than the workspace was sized for
  }
  const size_t need = wsl_conv2d_wgrad_ws_bytes(N, H, W, Ci, Co, ks);
  if (ws_bytes < need) {
    set_error("conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    return WSL_EWORKSPACE;
  }
  const int KK = ks * ks;
  // (the plan actually launched, not only the sizing query: ADVICE r5)
  WSL_REQUIRE(sizeof(float) * (size_t)g.nsplit * ((size_t)KK * Co * Ci + Co) <= ws_bytes, "conv2d_wgrad: %d splits do not fit the workspace", g.nsplit);
  p.dy = dy, p.dy_bs = dy_bs, p.N = N, p.Co = Co;
  p.part_dw = static_cast<float*>(ws);
  p.part_db = p.part_dw + (size_t)g.nsplit * KK * Co * Ci;
  p.tiles_x = g.tiles_x, p.tiles_y = g.tiles_y, p.items = g.items, p.nsplit = g.nsplit, p.co_blocks = g.co_blocks;
  // a ci-block beyond the first never writes db and a (co,ci) element outside the tensor is never written: no memset
  static const bool small_on = (WSL_TUNE("WSL_WGRAD_SMALL", 1) != 0);
  const int small_kind = (v2 && small_on) ? wgrad_small_kind(p.in.a, &p.in.b, H, W, Co, ks) : 0;
  if (small_kind) {   // first convolution / 4-class classifier: one narrow operand (wsl_conv4.hip)
    if (int rc = wgrad_small_launch(small_kind, p.in.a, dy, dy_bs, p.part_dw, p.part_db, N, H, W, g.nsplit, stream)) return rc;
  } else if (v2) {
    if (int rc = wgrad2_launch(p.in.a, &p.in.b, dy, dy_bs, p.part_dw, p.part_db, N, H, W, Co, ks, g.th, g.tw, g.cb, g.ib,
                               g.nsplit, g.items, g.tiles_x, g.tiles_y, g.co_blocks, g.ci_blocks, stream))
      return rc;
  } else if (int rc = (ks == 3 ? dispatch_wgrad<3>(p, g, stream) : dispatch_wgrad<1>(p, g, stream))) {
    return rc;
  }
  out->part_dw = p.part_dw, out->part_db = p.part_db, out->dw = dw, out->db = db;
  out->Co = Co, out->Ci = Ci, out->KK = KK, out->nsplit = g.nsplit;
  return WSL_OK;
}

extern "C" int wsl_conv2d_wgrad_partial(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, float* dw, float* db,
                                        int N, int H, int W, int Co, int ks, void* ws, size_t ws_bytes, WslWgradPending* pending,
                                        void* stream) {
  WSL_REQUIRE(pending, "conv2d_wgrad_partial: null pending record");
  return wgrad_stage1(a, b, dy, dy_bs, dw, db, N, H, W, Co, ks, ws, ws_bytes, stream, pending);
}

namespace wsl {
// second stage of up to kReduceMax pending weight gradients in ONE launch: a workgroup = 32 consecutive elements of one
// layer x 8 split groups, merged through LDS in a fixed order (same scheme as wgrad_reduce_kernel)
constexpr int kReduceMax = 40;
struct ReduceTable {
  int n;
  int _pad;
  struct E {
    const float* pdw;
    const float* pdb;
    float* dw;
    float* db;
    int Co, Ci, KK, nsplit;
    int64_t blk0;          // first workgroup of this layer
  } e[kReduceMax];
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceTable t) {
  __shared__ float red[kThreads];
  constexpr int SG = 8, EPB = kThreads / SG;
  int li = 0;
  for (int i = 1; i < t.n; ++i) li = (int64_t)blockIdx.x >= t.e[i].blk0 ? i : li;   // uniform
  const ReduceTable::E& L = t.e[li];
  const int64_t E = (int64_t)L.KK * L.Co * L.Ci;
  const int64_t total = E + (L.db ? L.Co : 0);
  const int el = threadIdx.x % EPB, sg = threadIdx.x / EPB;
  const int64_t e = ((int64_t)blockIdx.x - L.blk0) * EPB + el;
  float s = 0.f;
  if (e < total) {
    const float* src = e < E ? L.pdw + e : L.pdb + (e - E);
    const int64_t stride = e < E ? E : L.Co;
    for (int k = sg; k < L.nsplit; k += SG) s += src[k * stride];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < SG; ++k) v += red[k * EPB + el];
    if (e < E) {
      const int tap = (int)(e / ((int64_t)L.Co * L.Ci));
      const int64_t rem = e - (int64_t)tap * L.Co * L.Ci;  // co*Ci + ci
      L.dw[rem * L.KK + tap] = v;
    } else {
      L.db[e - E] = v;
    }
  }
}

// the same second stage with four consecutive elements per thread (512-byte instead of 128-byte runs per split row; round 5: 67 us per
// launch at 2 TB/s before): a workgroup = 128 consecutive elements x 8 split groups; the same sums in the same order, element by element
__global__ __launch_bounds__(256) void wgrad_reduce_batch4_kernel(ReduceTable t) {
  __shared__ float4 red[kThreads];
  constexpr int SG = 8, EPB = kThreads / SG;
  int li = 0;
  for (int i = 1; i < t.n; ++i) li = (int64_t)blockIdx.x >= t.e[i].blk0 ? i : li;   // uniform
  const ReduceTable::E& L = t.e[li];
  const int64_t E = (int64_t)L.KK * L.Co * L.Ci;
  const int64_t total = E + (L.db ? L.Co : 0);
  const int el = threadIdx.x % EPB, sg = threadIdx.x / EPB;
  const int64_t e = (((int64_t)blockIdx.x - L.blk0) * EPB + el) * 4;   // E and Co are multiples of 4: a quad never straddles the two regions
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total) {
    const float* src = e < E ? L.pdw + e : L.pdb + (e - E);
    const int64_t stride = e < E ? E : L.Co;
#pragma unroll 4
    for (int k = sg; k < L.nsplit; k += SG) {
      const float4 v = *reinterpret_cast<const float4*>(src + k * stride);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < SG; ++k) {
      const float4 r = red[k * EPB + el];
      v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
    }
    if (e < E) {
      const int64_t CC = (int64_t)L.Co * L.Ci;
      const int tap = (int)(e / CC);             // (CC is a multiple of 4: the quad stays inside one tap)
      const int64_t rem = e - (int64_t)tap * CC;  // co*Ci + ci
#pragma unroll
      for (int q = 0; q < 4; ++q) L.dw[(rem + q) * L.KK + tap] = v[q];
    } else {
      *reinterpret_cast<float4*>(L.db + (e - E)) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

}  // namespace wsl

extern "C" int wsl_wgrad_reduce_batch(const WslWgradPending* items, int n, void* stream) {
  WSL_REQUIRE(items && n > 0, "wgrad_reduce_batch: nothing to reduce");
  for (int i0 = 0; i0 < n; i0 += kReduceMax) {
    ReduceTable t;
    t.n = n - i0 < kReduceMax ? n - i0 : kReduceMax, t._pad = 0;
    int64_t blk = 0;
    double bytes = 0.0;
    bool quad = true;   // four elements per thread where every layer of the batch allows it
    for (int i = 0; i < t.n; ++i) {
      const WslWgradPending& q = items[i0 + i];
      WSL_REQUIRE(q.part_dw && q.dw && q.Co > 0 && q.Ci > 0 && q.KK > 0 && q.nsplit > 0, "wgrad_reduce_batch: item %d is malformed", i0 + i);
      quad = quad && ((int64_t)q.Co * q.Ci) % 4 == 0 && q.Co % 4 == 0 && (reinterpret_cast<uintptr_t>(q.part_dw) & 15) == 0 &&
             (!q.db || ((reinterpret_cast<uintptr_t>(q.part_db) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.db) & 15) == 0));
    }
    const int epw = quad ? 128 : 32;
    for (int i = 0; i < t.n; ++i) {
      const WslWgradPending& q = items[i0 + i];
      const int64_t total = (int64_t)q.KK * q.Co * q.Ci + (q.db ? q.Co : 0);
      t.e[i] = ReduceTable::E{q.part_dw, q.part_db, q.dw, q.db, q.Co, q.Ci, q.KK, q.nsplit, blk};
      blk += (total + epw - 1) / epw;
      bytes += 4.0 * (double)total * (q.nsplit + 1);
    }
    ProfScope ps(PF_WGRAD_REDUCE, 0.0, bytes, stream);
    if (quad) WSL_LAUNCH(wgrad_reduce_batch4_kernel, dim3((unsigned)blk), dim3(kThreads), 0, stream, t);
    else WSL_LAUNCH(wgrad_reduce_batch_kernel, dim3((unsigned)blk), dim3(kThreads), 0, stream, t);
  }
  return check_launch("wgrad_reduce_batch_kernel");
}

extern "C" int wsl_conv2d_wgrad(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, float* dw, float* db,
                                int N, int H, int W, int Co, int ks, void* ws, size_t ws_bytes, void* stream) {
  WslWgradPending q;
  if (int rc = wgrad_stage1(a, b, dy, dy_bs, dw, db, N, H, W, Co, ks, ws, ws_bytes, stream, &q)) return rc;
  const int KK = q.KK, Ci = q.Ci;
  struct { int nsplit; } g{q.nsplit};
  struct { const float* part_dw; const float* part_db; } p{q.part_dw, q.part_db};
  const int64_t total = (int64_t)KK * Co * Ci + (db ? Co : 0);
  void* tok = prof_begin(PF_WGRAD_REDUCE, 0.0, 4.0 * (double)total * (g.nsplit + 1), stream);
  struct EndProf { void* t; void* s; ~EndProf() { prof_end(t, s); } } endprof{tok, stream};
  if (g.nsplit >= 64) {
    WSL_LAUNCH((wgrad_reduce_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(kThreads), 0, stream, p.part_dw,
               p.part_db, dw, db, Co, Ci, KK, g.nsplit);
  } else {
    WSL_LAUNCH((wgrad_reduce_kernel<4>), dim3((unsigned)((total + 63) / 64)), dim3(kThreads), 0, stream, p.part_dw,
               p.part_db, dw, db, Co, Ci, KK, g.nsplit);
  }
  return check_launch("wgrad_reduce_kernel");
}
