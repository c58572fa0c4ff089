// v3 of the forward / data-gradient convolution: WAVE-SPECIALISED persistent workgroups.
//
// Same implicit-GEMM mapping, LDS operand layout, packed weights and aligned float4 staging as wsl_conv2.hip, but the
// work inside a workgroup is split by ROLE (profiles/r1c ablation: with every wave doing load -> commit -> MFMA in
// lock-step, staging (+25 %), the per-tile prologue latency and the epilogue store burst stayed exposed, because
// co-resident workgroups run phase-aligned and never fill each other's gaps):
//
//   waves 0-3  MFMA waves    v_mfma_f32_16x16x4_f32 on LDS stage b; they never touch global memory except for the
//                            epilogue stores of a finished tile, and never wait on a global load;
//   waves 4-7  loader waves  aligned float4 loads of the NEXT chunk (register prefetch one step ahead), BN-apply +
//                            LeakyReLU + dropout transform, ds_write into LDS stage b^1; they own the BN / channel-mask
//                            tables and never issue an MFMA.
//
// One s_barrier per channel chunk ("step") for all eight waves.  Workgroups are persistent: each walks a contiguous
// run of spatial tiles, the step sequence simply continues across tile boundaries, so while the MFMA waves run a
// tile's epilogue the loaders are already filling the next tile's first chunk.  The BatchNorm partial statistics are
// reduced per WAVE with shuffles only (four slots per tile), so the epilogue needs no workgroup barrier at all.
#include <stdlib.h>

#include "wsl_rt.h"

#ifndef WSL_EXPERIMENTS
// Measured slower than the lock-step kernel (profiles/r1d_conv_variants.md): this translation unit is compiled only into the
// EXPERIMENTS build (build.sh exp) and the test-only host emulator; the product library holds these three stubs instead.
namespace wsl {
bool conv3_enabled() { return false; }
void conv_set_variant(int) {}
int conv3_fwd(const WslSrc&, const WslSrc*, const float*, const float*, float*, int64_t, int, int, int, int, int, int, int, int, int,
              float*, float*, void*) {
  set_error("conv3: the wave-specialised variant is not part of the product build");
  return WSL_EUNSUPPORTED;
}
}  // namespace wsl
#else


namespace wsl {

struct Src3 {
  const float* x;
  const uint8_t* emask;
  const float* scale;
  const float* shift;
  const float* cmask;
  int64_t bs;
  int C;
  float es;
};

struct Conv3P {
  Src3 a, b;
  const float* wp;  // packed [KK][Ci][Co]
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, H, W, Ci, Co, tiles_x, tiles_y;
  float* stat_part;  // [tiles*4][Co][2]
  float* stat_cnt;   // [tiles*4]
  int ablate;        // debug (env WSL_CONV_ABLATE): 1 MFMA waves skip the stages, 2 loaders idle after the first step
};

template <int KS, int TH, int TW, int CO_T, int KC>
struct Conv3Cfg {
  static constexpr int NLT = 256;  // loader threads (waves 4-7); MFMA threads are waves 0-3
  static constexpr int P = KS / 2, KK = KS * KS, PADL = P ? 4 : 0;
  static constexpr int ROWP = TW + 2 * PADL, ROWS = TH + 2 * P, ROWP4 = ROWP / 4, POS = ROWS * ROWP4;
  static constexpr int G = NLT / POS, NLD = (KC + G - 1) / G;
  static constexpr int PLANE = ((ROWS * ROWP - 16 + 31) / 32) * 32 + 16;  // == 16 (mod 32)
  static constexpr int CSTR = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;
  static constexpr int SEGS = TW / 16, MT_TOTAL = TH * SEGS, MT = MT_TOTAL / 4, NT = CO_T / 16;
  static constexpr int IN_FLOATS = KC * PLANE, W_FLOATS = KK * KC * CSTR, BUF = IN_FLOATS + W_FLOATS;
  static constexpr int WQ = CO_T / 4, WF4 = KK * KC * WQ, NWL = (WF4 + NLT - 1) / NLT;
  static constexpr int MAXC = 256;
  static constexpr size_t SMEM = sizeof(float) * (2 * BUF + 4 * MAXC);
  static_assert(POS <= NLT && G >= 1 && MT_TOTAL % 4 == 0 && KC % 4 == 0, "tile shape");
};

template <typename C, int KS, int KC, int NG>
__device__ __forceinline__ void conv3_mfma_stages(const float* in_t, const float* w_t, const int (&abase)[C::MT], int bbase,
                                                   v4f (&acc)[C::MT][C::NT]) {
  constexpr int NS = KS * KS * NG;  // stages = (tap, channel group); operands of stage s+1 are read before stage s issues
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
__global__ __launch_bounds__(512, 2) void conv_mfma3_kernel(Conv3P p) {
  using C = Conv3Cfg<KS, TH, TW, CO_T, KC>;
  WSL_DYN_SMEM(smem);
  float* lds = reinterpret_cast<float*>(smem);  // stage b: input tile at lds + b*BUF, its weights right behind
  float* sc_l = lds + 2 * C::BUF;               // [Ci] BN scale (1 when the source is raw)
  float* sh_l = sc_l + C::MAXC;                 // [Ci] BN shift
  float* cm_l = sh_l + C::MAXC;                 // [2][Ci] channel multipliers of a tile's sample, by tile parity
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.y * CO_T;
  const int H = p.H, W = p.W, Ci = p.Ci;
  const int64_t HW = (int64_t)H * W;
  const int T = p.tiles_x * p.tiles_y * p.N;
  const int t0 = (int)((int64_t)blockIdx.x * T / gridDim.x), t1 = (int)((int64_t)(blockIdx.x + 1) * T / gridDim.x);
  const int nch = (Ci + KC - 1) / KC;
  const int nsteps = (t1 - t0) * nch;  // step g = (tile t0 + g / nch, chunk g % nch), LDS stage g & 1
  if (nsteps <= 0) return;

  if (wave >= 4) {
    // ============================================================ loader waves
    const int lt = tid - 256;
    const int grp = lt / C::POS, pos = lt - grp * C::POS;
    const int pty = pos / C::ROWP4, ptx4 = pos - pty * C::ROWP4;
    const int loff = pty * C::ROWP + ptx4 * 4;
    const bool co_vec = (p.Co & 3) == 0;
    const bool any_cmask = p.a.cmask || p.b.cmask;
    float4 pre[C::NLD];
    uchar4 prm[C::NLD];
    float4 prw[C::NWL];
    bool pre_valid = false;

    auto issue = [&](int g) {  // start the loads of step g into registers
      int t = t0 + g / nch;
      const int c0 = (g % nch) * KC;
      const int tx_i = t % p.tiles_x;
      t /= p.tiles_x;
      const int ty_i = t % p.tiles_y, n = t / p.tiles_y;
      const int gy = ty_i * TH + pty - C::P, gx = tx_i * TW + ptx4 * 4 - C::PADL;
      pre_valid = grp < C::G && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const uint32_t toff = (uint32_t)(grp * (int)HW + gy * W + gx);  // + uniform channel base below
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        const int cb = c0 + i * C::G, cg = cb + grp;  // cb: uniform first channel of this load instruction
        pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        prm[i] = make_uchar4(1, 1, 1, 1);
        if (cb < Ci) {
          const bool ina = cb < p.a.C;  // uniform (eligibility: a.C % 4 == 0 when two sources are present)
          const Src3& s = ina ? p.a : p.b;
          const int chb = ina ? cb : cb - p.a.C;
          const float* xb = s.x + n * s.bs + (int64_t)chb * HW;
          const uint8_t* mb = s.emask ? s.emask + ((int64_t)n * s.C + chb) * HW : nullptr;
          if (pre_valid && i * C::G + grp < KC && cg < Ci) {
            pre[i] = *reinterpret_cast<const float4*>(xb + toff);
            if (mb) prm[i] = *reinterpret_cast<const uchar4*>(mb + toff);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < C::NWL; ++i) {
        const int f = lt + i * C::NLT;
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
      // first chunk of a tile: (re)build that tile's channel-multiplier table in its parity slot; it is read by the
      // commit of this step, one barrier later
      if (any_cmask && c0 == 0) {
        float* tab = cm_l + ((t0 + g / nch) & 1) * C::MAXC;
        for (int c = lt; c < Ci; c += C::NLT) {
          const bool ina = c < p.a.C;
          const Src3& s = ina ? p.a : p.b;
          tab[c] = s.cmask ? s.cmask[(int64_t)n * s.C + (ina ? c : c - p.a.C)] : 1.f;
        }
      }
    };

    auto commit = [&](int g) {  // transform the registers of step g and write them to LDS stage g & 1
      float* in_b = lds + (g & 1) * C::BUF;
      const int c0 = (g % nch) * KC;
      const float* tab = cm_l + ((t0 + g / nch) & 1) * C::MAXC;
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        const int c = grp + i * C::G, cg = c0 + c;
        if (grp < C::G && c < KC) {
          float4 v = pre[i];
          if (pre_valid && cg < Ci) {
            const bool ina = cg < p.a.C;
            const Src3& s = ina ? p.a : p.b;
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
              const float cm = tab[cg];
              v.x *= cm, v.y *= cm, v.z *= cm, v.w *= cm;
            }
          }
          *reinterpret_cast<float4*>(in_b + c * C::PLANE + loff) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < C::NWL; ++i) {
        const int f = lt + i * C::NLT;
        if (f < C::WF4) {
          const int row = f / C::WQ, q = f - row * C::WQ;
          *reinterpret_cast<float4*>(in_b + C::IN_FLOATS + row * C::CSTR + q * 4) = prw[i];
        }
      }
    };

    issue(0);  // tile loads first, then the BN tables: both in flight together
    for (int c = lt; c < Ci; c += C::NLT) {
      const bool ina = c < p.a.C;
      const Src3& s = ina ? p.a : p.b;
      const int ch = ina ? c : c - p.a.C;
      sc_l[c] = s.scale ? s.scale[ch] : 1.f;
      sh_l[c] = s.scale ? s.shift[ch] : 0.f;
    }
    __syncthreads();  // (A) tables visible to every loader thread
    commit(0);
    if (nsteps > 1) issue(1);
    __syncthreads();  // (B) stage 0 ready
    for (int g = 0; g < nsteps; ++g) {  // MFMA waves compute step g meanwhile
      if (!WSL_ABLATED(p, 2)) {
        if (g + 1 < nsteps) commit(g + 1);
        if (g + 2 < nsteps) issue(g + 2);
      }
      __syncthreads();
    }
    return;
  }

  // ============================================================== MFMA waves
  int abase[C::MT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i) {
    const int mt = wave * C::MT + i;
    abase[i] = (lane >> 4) * C::PLANE + (mt / C::SEGS) * C::ROWP + (mt % C::SEGS) * 16 + (lane & 15) + (C::PADL - C::P);
  }
  const int bbase = (lane >> 4) * C::CSTR + (lane & 15);
  v4f acc[C::MT][C::NT];
  __syncthreads();  // (A)
  __syncthreads();  // (B)
  for (int g = 0; g < nsteps; ++g) {
    const int k = g % nch;
    if (k == 0) {
#pragma unroll
      for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    const float* in_b = lds + (g & 1) * C::BUF;
    // channels past Ci were staged as zeros: no branch inside; a chunk holding <= 4 channels (Ci = 1 or 4 layers) runs
    // the single-group variant
    if (WSL_ABLATED(p, 1)) {
    } else if (Ci - k * KC > 4) conv3_mfma_stages<C, KS, KC, KC / 4>(in_b, in_b + C::IN_FLOATS, abase, bbase, acc);
    else conv3_mfma_stages<C, KS, KC, 1>(in_b, in_b + C::IN_FLOATS, abase, bbase, acc);

    if (k == nch - 1) {
      // ---- epilogue of this tile (MFMA waves only; the loaders are already on the next tile): bias, float4 stores,
      //      per-WAVE BatchNorm partials (sum, M2 about the wave's own mean, count) -- shuffles only, no barrier
      int t = t0 + g / nch;
      const int tile = t;
      const int tx_i = t % p.tiles_x;
      t /= p.tiles_x;
      const int ty_i = t % p.tiles_y, n = t / p.tiles_y;
      const int y0 = ty_i * TH, x0 = tx_i * TW;
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        const int co = co0 + j * 16 + (lane & 15);
        const float bias = (p.bias && co < p.Co) ? p.bias[co] : 0.f;
        float sum = 0.f, cnt = 0.f;
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
            sum += (v[0] + v[1]) + (v[2] + v[3]);
            cnt += 4.f;
          }
        }
        if (p.stat_part) {  // uniform
          sum += __shfl_xor(sum, 16), cnt += __shfl_xor(cnt, 16);
          sum += __shfl_xor(sum, 32), cnt += __shfl_xor(cnt, 32);
          const float mean_w = cnt > 0.f ? sum / cnt : 0.f;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < C::MT; ++i) {
            const int mt = wave * C::MT + i;
            const int oy = y0 + mt / C::SEGS, ox = x0 + (mt % C::SEGS) * 16 + (lane >> 4) * 4;
            if (co < p.Co && oy < H && ox < W) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float d = acc[i][j][r] - mean_w;
                q = fmaf(d, d, q);
              }
            }
          }
          q += __shfl_xor(q, 16);
          q += __shfl_xor(q, 32);
          if (lane < 16 && co < p.Co) {
            float* dst = p.stat_part + ((int64_t)co * (4 * T) + tile * 4 + wave) * 2;   // [Co][slots][2]
            dst[0] = sum, dst[1] = q;
          }
          // the count is the same for every channel column of a wave; channel 0 of co-tile 0 always exists
          if (lane == 0 && j == 0 && blockIdx.y == 0) p.stat_cnt[tile * 4 + wave] = cnt;
        }
      }
    }
    __syncthreads();
  }
}

template <int KS, int TH, int TW, int CO_T>
static int launch_conv3(Conv3P& p, int is_dgrad, void* stream) {
  using C = Conv3Cfg<KS, TH, TW, CO_T, 8>;
  auto kern = conv_mfma3_kernel<KS, TH, TW, CO_T, 8>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  const int tiles = p.tiles_x * p.tiles_y * p.N, gy = cdiv(p.Co, CO_T);
  int gx = (device_cu_count() * 1 + gy - 1) / gy;  // one resident 8-wave workgroup per CU (256 registers per thread)
  if (gx > tiles) gx = tiles;
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(is_dgrad ? 1 : 0, 2.0 * px * p.Co * p.Ci * KS * KS, 4.0 * px * (p.Co + p.Ci), stream);
  WSL_LAUNCH(kern, dim3(gx, gy), dim3(512), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_mfma3_kernel");
}

static Src3 to_src3(const WslSrc& s) { return Src3{s.x, s.emask, s.scale, s.shift, s.cmask, s.bs, s.C, s.emask_scale}; }

// Measured on MI355X (profiles/r1d_conv_variants.md): this variant is NOT faster than the lock-step v2 kernel -- run
// alone the MFMA waves need ~196 us and the loader waves ~145 us for a 32->32 128^2 N=64 layer, together 272 us (v2:
// 237 us): two roles on one SIMD barely overlap.  It stays in the tree as an opt-in experiment (wsl_debug_conv_variant(3)
// or env WSL_CONV_V3=1) for the next round; the default is v2.
static int g_variant = 0;
bool conv3_enabled() {
  if (g_variant == 0) g_variant = WSL_TUNE("WSL_CONV_V3", 0) ? 3 : 2;
  return g_variant == 3;
}
void conv_set_variant(int v) { g_variant = (v == 3) ? 3 : 2; }

int conv3_fwd(const WslSrc& a, const WslSrc* b, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H,
              int W, int Co, int ks, int is_dgrad, int th, int tw, int co_t, float* stat_part, float* stat_cnt,
              void* stream) {
  Conv3P p;
  p.a = to_src3(a);
  p.b = (b && b->C > 0) ? to_src3(*b) : Src3{};
  p.wp = wp, p.bias = bias, p.y = y, p.y_bs = y_bs;
  p.N = N, p.H = H, p.W = W, p.Ci = a.C + p.b.C, p.Co = Co;
  p.tiles_x = cdiv(W, tw), p.tiles_y = cdiv(H, th);
  p.stat_part = stat_part, p.stat_cnt = stat_cnt;
  static const int ablate = WSL_TUNE("WSL_CONV_ABLATE", 0);
  p.ablate = ablate;
#define WSL_CASE(KS_, TH_, TW_, CO_) \
  if (ks == KS_ && th == TH_ && tw == TW_ && co_t == CO_) return launch_conv3<KS_, TH_, TW_, CO_>(p, is_dgrad, stream);
  WSL_CASE(3, 8, 64, 16) WSL_CASE(3, 8, 64, 32) WSL_CASE(3, 8, 32, 16) WSL_CASE(3, 8, 32, 32) WSL_CASE(3, 8, 32, 64)
  WSL_CASE(3, 16, 16, 16) WSL_CASE(3, 16, 16, 32) WSL_CASE(3, 16, 16, 64)
  WSL_CASE(1, 8, 64, 16) WSL_CASE(1, 8, 64, 32) WSL_CASE(1, 8, 32, 16) WSL_CASE(1, 8, 32, 32) WSL_CASE(1, 8, 32, 64)
  WSL_CASE(1, 16, 16, 16) WSL_CASE(1, 16, 16, 32) WSL_CASE(1, 16, 16, 64)
#undef WSL_CASE
  set_error("conv3: no kernel for ks %d tile %dx%d co_t %d", ks, th, tw, co_t);
  return WSL_EUNSUPPORTED;
}

}  // namespace wsl

#endif  // WSL_EXPERIMENTS
