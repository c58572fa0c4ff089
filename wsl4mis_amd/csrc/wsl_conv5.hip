// Winograd F(2x2, 3x3) convolution on the f32 matrix cores (forward and data-gradient of every 3x3 layer whose channel
// counts are multiples of 8 / 16).  The direct implicit-GEMM kernels (wsl_conv2.hip) are bound by the f32 MFMA rate:
// 9 * Ci * Co multiply-adds per output pixel.  The minimal-filtering form spends 16 * Ci * Co per 2x2 output tile = 4 per
// pixel, i.e. 2.25x fewer matrix instructions, for one extra pass over the staged tile:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
//   * U = G g G^T is computed once per optimiser step and layer by wino_pack_table_kernel ([16][Ci][Co], fp32; G holds
//     0, 1, +-1/2 only);
//   * conv_wino2_kernel / conv_wino2r_kernel below: the input transform B^T d B is formed in registers straight in the MFMA
//     operand layout, D_xi[tile][co] += V_xi[tile][ci] * U_xi[ci][co] per transform position xi (v_mfma_f32_16x16x4_f32), the
//     output transform A^T M A is register-only, the epilogue is the direct kernels' (bias, float4 stores, BatchNorm partials,
//     BatchNorm-backward statistics).
// (The first form -- V staged through LDS -- and the 8-wave weight gradient lost their A/B measurements in rounds 1-2 and were
//  deleted in round 3: profiles/r1z_winograd.md keeps the numbers.)
#include <stdlib.h>

#include <type_traits>

#include "wsl_rt.h"

namespace wsl {

struct WSrc {            // one source, device view (as Src2 of wsl_conv2.hip)
  const float* x;
  const uint8_t* emask;
  const float* scale;
  const float* shift;
  const float* cmask;
  int64_t bs;
  int C;
  float es;
};

struct WinoP {
  WSrc a, b;
  const float* u;        // 16 * Ci * Co floats, blocked operand order (wino_filter)
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, H, W, Ci, Co, tiles_x, tiles_y;
  float* stat_part;
  float* stat_cnt;
  int slots;
  int ablate;   // debug (env WSL_CONV_ABLATE): 128 = per-workgroup timeline stamps into stat_part (tools/microbench_conv.py)
  BnBwdEpi bn;  // data-gradient launches: BatchNorm-backward statistics of the consumer of y (wsl_rt.h)
};

// ------------------------------------------------------------------------------------------------ conv kernels
// The input transform sits INSIDE the MFMA operand: lane (tile = l & 15, k = l >> 4) of a wave is
// exactly the A-operand slot of (tile, input channel k of the current group of four), so it reads that channel's 4x4 patch
// of its tile from the staged raw halo tile and forms B^T d B in registers (8 packed + 16 scalar adds) -- no V buffer in
// LDS, no transform pass, two barriers per chunk instead of three, 24 instead of 64 LDS reads + 16 writes per chunk.
// A wave owns MTW row-groups of 16 tiles and NT channel tiles (MTW * NT = 2: 64 tiles x 32 channels, or -- for the
// 16-channel layers -- 128 tiles = 8 x 64 pixels x 16 channels).
// one aligned 8-byte LDS read that the load/store optimiser must not fuse with its neighbour into ds_read2_b64 (that form takes twice
// the LDS cycles of two ds_read_b64: profiles/r5_probe_lds5.md) -- hence volatile
// (volatile on a GENERIC pointer would turn the read into flat_load_dwordx2: the address-space inference pass leaves volatile accesses
//  alone -- so the pointer is cast to the LDS address space first)
__device__ __forceinline__ wsl_v2f wino_read_pair(const float* p) {
#ifdef WSL_HOST_EMUL
  return *reinterpret_cast<const wsl_v2f*>(p);
#else
  return *(const volatile __attribute__((address_space(3))) wsl_v2f*)(p);
#endif
}
// a staged float4 goes one float to the right of its 16-byte slot (the shifted image): 4 + 8 + 4 bytes
__device__ __forceinline__ void wino_store_shifted(float* slot, float a, float b, float c, float d) {
  slot[1] = a;
  *reinterpret_cast<float2*>(slot + 2) = make_float2(b, c);
  slot[4] = d;
}

template <int TH, int TW, int NT>
struct Wino2Cfg {
  static constexpr int KC = 8, PADL = 4, CO_T = 16 * NT;
  static constexpr int ROWP = TW + 2 * PADL, ROWS = TH + 2, ROWP4 = ROWP / 4, POS = ROWS * ROWP4;
  static constexpr int G = 256 / POS, NLD = KC / G;   // staging: thread t = float4 position t % POS of plane t / POS of a pass of G planes
  // LDS image of one channel: ROWS x ROWP floats SHIFTED BY ONE FLOAT (column c of the image holds global column x0 - PADL - 1 + c), so
  // that a tile's 4 x 4 patch starts at the EVEN offset PADL + 2 * txx and is read as aligned ds_read_b64; planes 32 (mod 64) floats
  // apart: the two channels of a 32-lane group (lanes = 16 tiles x 2 channels, 8 bytes each) cover the 64 banks exactly once.
  // (Rounds 1-4: unshifted image, ds_read2_b32 at odd offsets, planes 16 (mod 32): all 32 lanes of a group on the 16 even -- then
  // the 16 odd -- banks = 2-way conflicts on every patch read and 4x the LDS cycles; profiles/r5_probe_lds5.md.)
  static constexpr int SHIFT = 1;
  static constexpr int PLANE = ((ROWS * ROWP + SHIFT - 32 + 63) / 64) * 64 + 32;   // == 32 (mod 64), >= ROWS * ROWP + SHIFT
  static constexpr int TTY = TH / 2, TTX = TW / 2, TILES = TTY * TTX, MTW = TILES / 64, NA = MTW * NT;
  static constexpr int IN_FLOATS = KC * PLANE, W_FLOATS = 16 * KC * CO_T;
  static constexpr int WF4 = W_FLOATS / 4, NWL = WF4 / 256;
  static constexpr int MAXC = 256;
  static constexpr size_t SMEM = sizeof(float) * (IN_FLOATS + W_FLOATS + 3 * MAXC);
  static_assert(TILES % 64 == 0 && (NA == 1 || NA == 2) && POS <= 256 && KC % G == 0 && WF4 % 256 == 0 && TTX % 4 == 0, "tile shape");
  // resident workgroups per CU the register allocator leaves room for: 64 accumulator registers (NA == 1) fit three
  static constexpr int MINW = NA == 1 ? 3 : 2;
  static_assert(8 * CO_T <= IN_FLOATS && (TTX % 16 == 0 || MTW == 1), "reduction scratch / row groups");
};

// epilogue of both conv_wino2 kernels: output transform (register-only), bias, float4 stores, BatchNorm partials
// positions of the lane's output float4s (and of the BatchNorm-backward epilogue's y / keep-mask reads): q = (m * NT + j) * 4
// + {0: row 0, 1: row 0 + 4 px, 2: row 1, 3: row 1 + 4 px}, as a 32-bit ELEMENT offset from the workgroup's corner (n, co0, y0, x0):
// the corner is wave-uniform (a scalar base register), so every access is base + 32-bit lane offset -- no 64-bit multiplies per
// lane (the 64-bit form cost ~8 vector instructions per float4 pair, two of them quarter-rate v_mul_lo_u32).  wino_fwd() admits only
// shapes with CO_T * H * W < 2^30.
template <typename C, int NT>
__device__ __forceinline__ uint32_t wino2_out_lane_off(int HW, int W, int q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = q >> 2, m = a / NT, j = a - m * NT, r = q & 3;
  const int tb = (wave * C::MTW + m) * 16 + 4 * (lane >> 4);
  const int tyy = tb / C::TTX, txb = tb - tyy * C::TTX;
  return (uint32_t)((j * 16 + (lane & 15)) * HW + (2 * tyy + (r >> 1)) * W + 2 * txb + 4 * (r & 1));
}
__device__ __forceinline__ int64_t wino2_corner(int n, int64_t bs_or_chw, int co0, int HW, int y0, int W, int x0) {
  return (int64_t)n * bs_or_chw + (int64_t)co0 * HW + (int64_t)y0 * W + x0;
}

// PRE: the caller has the BatchNorm-backward epilogue's y / keep-mask values in registers (loaded before its channel loop)
// (EABL, experiments build only: 32 = the output stores are predicated off (same instructions otherwise), 64 = no statistics)
// (SD: a data-gradient launch that writes d = g * keep * scale * leaky'(z) in place of g -- BnBwdEpi::store_d -- as its own instantiation:
//  as a run-time branch the second store path cost the kernels 14-24 spilled registers)
template <typename C, int TH, int TW, int NT, bool PRE = false, int EABL = 0, bool SD = false>
__device__ __forceinline__ void wino2_epilogue(const WinoP& p, v4f (&acc)[16][C::NA], float* scratch, int n, int co0, int y0, int x0,
                                               int tile_id, int nb, int cby, const float4* ypre = nullptr,
                                               const uint32_t* mpre = nullptr) {
  constexpr int CO_T = C::CO_T, MTW = C::MTW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = p.W, HW = p.H * p.W;
  float* in_t = scratch;
  float o[C::NA][16];
  float bsum[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bsum[j] = 0.f;
  char* const ycorner = reinterpret_cast<char*>(p.y + wino2_corner(n, p.y_bs, co0, HW, y0, W, x0));   // wave-uniform
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int a = m * NT + j;
      const float bias = p.bias ? p.bias[co0 + j * 16 + (lane & 15)] : 0.f;
      float bs = 0.f;
      // first stage (A^T M, down the rows) on register pairs = two of the lane's four tiles per packed add
      wsl_v2f s0p[4][2], s1p[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const wsl_v2f m0 = {acc[c][a][2 * h], acc[c][a][2 * h + 1]}, m1 = {acc[4 + c][a][2 * h], acc[4 + c][a][2 * h + 1]};
          const wsl_v2f m2 = {acc[8 + c][a][2 * h], acc[8 + c][a][2 * h + 1]}, m3 = {acc[12 + c][a][2 * h], acc[12 + c][a][2 * h + 1]};
          s0p[c][h] = (m0 + m1) + m2;   // (plain C++: these read MFMA results -- see the hazard note in wsl_rt.h)
          s1p[c][h] = (m1 - m2) - m3;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) s0[c] = s0p[c][r >> 1][r & 1], s1[c] = s1p[c][r >> 1][r & 1];
        const float y00 = ((s0[0] + s0[1]) + s0[2]) + bias, y01 = ((s0[1] - s0[2]) - s0[3]) + bias;
        const float y10 = ((s1[0] + s1[1]) + s1[2]) + bias, y11 = ((s1[1] - s1[2]) - s1[3]) + bias;
        o[a][2 * r] = y00, o[a][2 * r + 1] = y01, o[a][8 + 2 * r] = y10, o[a][8 + 2 * r + 1] = y11;
        bs += (y00 + y01) + (y10 + y11);
      }
      const uint32_t yo = 4u * wino2_out_lane_off<C, NT>(HW, W, a * 4), wb = 4u * (uint32_t)W;   // byte offsets
      if constexpr ((EABL & 128) != 0) {
        // (ablation: the ADDRESS pattern of a lane-transposed store -- the four lanes of a channel row write 16-byte pieces 0-3, then 4-7 of
        //  its 128 bytes, so every store instruction fills whole 64-byte lines; the data land in the wrong place, the bytes are the same)
        const uint32_t q = (uint32_t)(lane >> 4), ya = yo - 16u * q, yb = ya + 64u;
        *reinterpret_cast<float4*>(ycorner + ya) = make_float4(o[a][0], o[a][1], o[a][2], o[a][3]);
        *reinterpret_cast<float4*>(ycorner + yb) = make_float4(o[a][4], o[a][5], o[a][6], o[a][7]);
        *reinterpret_cast<float4*>(ycorner + (ya + wb)) = make_float4(o[a][8], o[a][9], o[a][10], o[a][11]);
        *reinterpret_cast<float4*>(ycorner + (yb + wb)) = make_float4(o[a][12], o[a][13], o[a][14], o[a][15]);
      } else if constexpr (SD) {
        // (the BatchNorm-backward section below writes d = g * keep * scale * leaky'(z) in place of g: wsl_conv2d_dgrad_bn_d)
      } else if ((EABL & 32) == 0 || bs == 123.456f) {
        *reinterpret_cast<float4*>(ycorner + yo) = make_float4(o[a][0], o[a][1], o[a][2], o[a][3]);
        *reinterpret_cast<float4*>(ycorner + (yo + 16u)) = make_float4(o[a][4], o[a][5], o[a][6], o[a][7]);
        *reinterpret_cast<float4*>(ycorner + (yo + wb)) = make_float4(o[a][8], o[a][9], o[a][10], o[a][11]);
        *reinterpret_cast<float4*>(ycorner + (yo + wb + 16u)) = make_float4(o[a][12], o[a][13], o[a][14], o[a][15]);
      }
      bsum[j] += bs;
    }
  }
  if constexpr ((EABL & 64) != 0) return;
  if (p.bn.part) {   // BatchNorm-backward statistics of the layer that consumes this gradient (o[][] holds 2 rows x 8 pixels)
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = co0 + j * 16 + (lane & 15);
      const float mean = p.bn.st[co], invstd = p.bn.st[p.Co + co], sc = p.bn.st[2 * p.Co + co], sh = p.bn.st[3 * p.Co + co];
      BnBwdAcc ba;
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int tb = (wave * MTW + m) * 16 + 4 * (lane >> 4);
        const int tyy = tb / C::TTX, txb = tb - tyy * C::TTX;
        const int64_t base = ((int64_t)n * p.Co + co) * HW + (int64_t)(y0 + 2 * tyy) * W + x0 + 2 * txb;
        const int a = m * NT + j;
        if constexpr (PRE && SD) {   // d goes where g would have gone; the consumer's apply pass then reads no keep mask
          const uint32_t yo = 4u * wino2_out_lane_off<C, NT>(HW, W, a * 4), wb = 4u * (uint32_t)W;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float d[4];
            bn_bwd_acc4v(p.bn, ypre[a * 4 + r], mpre[a * 4 + r], o[a][4 * r], o[a][4 * r + 1], o[a][4 * r + 2], o[a][4 * r + 3], mean,
                         invstd, sc, sh, ba, d);
            *reinterpret_cast<float4*>(ycorner + (yo + (r & 1) * 16u + (r >> 1) * wb)) = make_float4(d[0], d[1], d[2], d[3]);
          }
        } else if constexpr (PRE) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bn_bwd_acc4v(p.bn, ypre[a * 4 + r], mpre[a * 4 + r], o[a][4 * r], o[a][4 * r + 1], o[a][4 * r + 2], o[a][4 * r + 3], mean,
                         invstd, sc, sh, ba);
        } else {
          bn_bwd_acc4(p.bn, base, o[a][0], o[a][1], o[a][2], o[a][3], mean, invstd, sc, sh, ba);
          bn_bwd_acc4(p.bn, base + 4, o[a][4], o[a][5], o[a][6], o[a][7], mean, invstd, sc, sh, ba);
          bn_bwd_acc4(p.bn, base + W, o[a][8], o[a][9], o[a][10], o[a][11], mean, invstd, sc, sh, ba);
          bn_bwd_acc4(p.bn, base + W + 4, o[a][12], o[a][13], o[a][14], o[a][15], mean, invstd, sc, sh, ba);
        }
      }
      bn_bwd_fold(ba, s1[j], s2[j]);
    }
    bn_bwd_store<NT, CO_T, true>(p.bn, s1, s2, in_t, co0, p.Co, tile_id, nb);   // (LDS-only barrier: the tile's stores keep draining)
    return;
  }
  if (p.stat_part) {
    float* red1 = in_t;
    float* red2 = in_t + 4 * CO_T;
    constexpr float cnt = (float)(TH * TW);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s = bsum[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) red1[wave * CO_T + j * 16 + lane] = s;
    }
    // (LDS-only barriers: __syncthreads() would also wait -- vmcnt(0) -- until the 32 KB of output stores issued above are acknowledged,
    //  two microseconds in which a full-resolution workgroup has nothing else to do; only the LDS scratch is shared here)
    WSL_LDS_BARRIER();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = j * 16 + (lane & 15);
      const float mean_b = (red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col]) / cnt;
      float q = 0.f;
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float d = o[m * NT + j][e] - mean_b;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (lane < 16) red2[wave * CO_T + j * 16 + lane] = q;
    }
    WSL_LDS_BARRIER();
    if (wave == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 16 + lane, co = co0 + col;
        float* dst = p.stat_part + ((int64_t)co * ((int64_t)nb * p.slots) + (int64_t)tile_id * p.slots) * 2;   // [Co][slots][2]
        dst[0] = red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col];
        dst[1] = red2[col] + red2[CO_T + col] + red2[2 * CO_T + col] + red2[3 * CO_T + col];
      }
      if (lane < p.slots && cby == 0) p.stat_cnt[tile_id * p.slots + lane] = lane == 0 ? cnt : 0.f;
    }
  }
}

template <int TH, int TW, int NT>
__global__ __launch_bounds__(256, (Wino2Cfg<TH, TW, NT>::MINW)) void conv_wino2_kernel(WinoP p) {
  using C = Wino2Cfg<TH, TW, NT>;
  constexpr int KC = C::KC, CO_T = C::CO_T, MTW = C::MTW;
  WSL_DYN_SMEM(smem);
  float* in_t = reinterpret_cast<float*>(smem);                 // raw (transformed-on-load) halo tile [KC][PLANE]
  float* w_t = in_t + C::IN_FLOATS;                             // U chunk, operand order: [16 xi][2 kg][4 k][16 col][NT j]
  float2* tab = reinterpret_cast<float2*>(w_t + C::W_FLOATS);   // [Ci] {scale, shift}
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
  const int cby = blockIdx.y;
  const int co0 = cby * CO_T;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = p.H, W = p.W, Ci = p.Ci, Co = p.Co;
  const int HW = H * W;

  // ---- staging position of this thread (as conv_mfma2l_kernel)
  const int grp = tid / C::POS, pos = tid - grp * C::POS;
  const int pty = pos / C::ROWP4, ptx4 = pos - pty * C::ROWP4;
  const int gy = y0 + pty - 1, gx = x0 + ptx4 * 4 - C::PADL;
  const bool owner = grp < C::G;
  const bool pvalid = owner && gy >= 0 && gy < H && gx >= 0 && gx < W;
  const uint32_t toff = pvalid ? (uint32_t)(grp * HW + gy * W + gx) : 0u;
  const int loff = grp * C::PLANE + pty * C::ROWP + ptx4 * 4;
  const int64_t gstride = (int64_t)C::G * HW;
  const float* xa_n = p.a.x + n * p.a.bs;
  const float* xb_n = p.b.C ? p.b.x + n * p.b.bs : nullptr;
  const uint8_t* ma_n = p.a.emask ? p.a.emask + (int64_t)n * p.a.C * HW : nullptr;
  const uint8_t* mb_n = (p.b.C && p.b.emask) ? p.b.emask + (int64_t)n * p.b.C * HW : nullptr;
  const float* w_n = p.u + (int64_t)cby * C::W_FLOATS + 4 * tid;
  const int64_t w_cstride = (int64_t)(Co / CO_T) * C::W_FLOATS;

  float4 pre[C::NLD];
  uint32_t prm[C::NLD];
  v4f prw[C::NWL];

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
    const float* wb = w_n + (c0 / KC) * w_cstride;
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) prw[i] = *reinterpret_cast<const v4f*>(wb + i * (4 * kThreads));
  };

  auto commit = [&](int c0) __attribute__((always_inline)) {
    if (pvalid) {
      const bool ina = c0 < p.a.C;
      const bool has_scale = (ina ? p.a.scale : p.b.scale) != nullptr;
      const bool has_mask = (ina ? p.a.emask : p.b.emask) != nullptr;
      const float es = ina ? p.a.es : p.b.es;
      float2 tb[C::NLD];
      float cmv[C::NLD];
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) tb[i] = tab[c0 + grp + i * C::G], cmv[i] = cm_l[c0 + grp + i * C::G];
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        wsl_v2f lo = {pre[i].x, pre[i].y}, hi = {pre[i].z, pre[i].w};
        if (has_scale) xform_bn_leaky(lo, hi, tb[i].x, tb[i].y);
        if (has_mask) xform_mask(lo, hi, prm[i], es);
        lo = lo * cmv[i], hi = hi * cmv[i];   // (1.0 without a channel mask: exact, and cheaper than selecting)
        // (4 + 8 + 4 bytes, one float right of the 16-byte slot.  These stores keep 2- / 4-way bank conflicts; the conflict-free form -- one
        //  aligned ds_write_b128 of {previous lane's last element (v_mov_b32_dpp wave_shr:1), own first three} -- costs more vector and
        //  branch instructions than the conflicts cost LDS cycles: measured 3-7 % slower per layer here, profiles/r5_wino_lds_layout.md)
        wino_store_shifted(in_t + i * (C::G * C::PLANE) + loff, lo[0], lo[1], hi[0], hi[1]);
      }
    }
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) *reinterpret_cast<v4f*>(w_t + 4 * tid + i * (4 * kThreads)) = prw[i];
  };

  issue(0);
  for (int c = tid; c < Ci; c += kThreads) {
    const bool ina = c < p.a.C;
    const WSrc& s = ina ? p.a : p.b;
    const int ch = ina ? c : c - p.a.C;
    tab[c] = s.scale ? make_float2(s.scale[ch], s.shift[ch]) : make_float2(1.f, 0.f);
    cm_l[c] = s.cmask ? s.cmask[(int64_t)n * s.C + ch] : 1.f;
  }

  v4f acc[16][C::NA];   // [xi][m * NT + j]
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int a = 0; a < C::NA; ++a) acc[i][a] = v4f{0.f, 0.f, 0.f, 0.f};
  // this lane's A-operand slots: tile (wave * MTW + m) * 16 + (lane & 15), channel (lane >> 4) of a group of four
  int poff[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int t = (wave * MTW + m) * 16 + (lane & 15), tyy = t / C::TTX, txx = t - tyy * C::TTX;
    poff[m] = (lane >> 4) * C::PLANE + (2 * tyy) * C::ROWP + (C::PADL - 1 + C::SHIFT) + 2 * txx;   // even
  }
  const int b_off = lane * NT;   // ((k = lane >> 4) * 16 + (col = lane & 15)) * NT j: lanes 2 * NT bytes apart -- a 32-lane group covers
                                 // every LDS bank exactly once (round 4's [k][col][kg][j] order put lanes l and l + 16 on the same banks)
  if (owner && !pvalid) {   // positions outside the image stay zero for good
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) wino_store_shifted(in_t + i * (C::G * C::PLANE) + loff, 0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();   // tables visible

  // (peeling the first chunk to skip the accumulator clear, as the raw-source kernel does, costs 23 VGPRs here: spills)
  for (int c0 = 0; c0 < Ci; c0 += KC) {
    commit(c0);
    if (c0 + KC < Ci) issue(c0 + KC);   // flies across the whole compute phase
    __syncthreads();
    // packed input transform (wsl_rt.h) where the registers allow it: the 128-tile variant sits at 256 VGPRs and the even
    // alignment of the register pairs costs it two spilled dwords, so it keeps the scalar form
    constexpr bool PK = MTW == 1;
    wsl_v2f rlo[MTW][4], rhi[MTW][4];   // the 4 x 4 patches, one row per pair of register pairs
    v4f rd[MTW][4];                     // (scalar form: one row per vector)
    auto fetch = [&](int kg) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* r = in_t + kg * (4 * C::PLANE) + poff[m] + i * C::ROWP;
          const wsl_v2f p0 = wino_read_pair(r), p1 = wino_read_pair(r + 2);   // two ds_read_b64 (even offsets), conflict-free
          if constexpr (PK) {
            rlo[m][i] = p0, rhi[m][i] = p1;
          } else {
            rd[m][i] = v4f{p0[0], p0[1], p1[0], p1[1]};
          }
        }
    };
    fetch(0);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      wsl_v2f va[MTW][4], vb[MTW][4];   // V[4 i + {0, 3}] = va[i], V[4 i + {1, 2}] = vb[i]: 16 packed adds
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        if constexpr (PK) {
          wino_btdb_pk(rlo[m], rhi[m], va[m], vb[m]);
        } else {
          const v4f rt[4] = {rd[m][0] - rd[m][2], rd[m][1] + rd[m][2], rd[m][2] - rd[m][1], rd[m][1] - rd[m][3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            va[m][i] = wsl_v2f{rt[i][0] - rt[i][2], rt[i][1] - rt[i][3]};
            vb[m][i] = wsl_v2f{rt[i][1] + rt[i][2], rt[i][2] - rt[i][1]};
          }
        }
      }
      if (kg == 0) fetch(1);   // the second channel group's patches are read while the first group's MFMAs issue
      // B operands are requested BD transform positions ahead of their MFMAs (a position's two MFMAs take 64 cycles, an
      // LDS read comes back after more than that)
      constexpr int BD = NT == 2 ? 4 : 2, BR = BD + 1;   // (the 128-tile variant has no registers to spare)
      float bv[BR][NT];
      auto loadb = [&](int xi, int buf) __attribute__((always_inline)) {
        const float* bp = w_t + xi * (4 * 16 * 2 * NT) + kg * (4 * 16 * NT) + b_off;
        if constexpr (NT == 2) {
          const wsl_v2f b2 = wino_read_pair(bp);   // (never fused into ds_read2st64_b64)
          bv[buf][0] = b2[0], bv[buf][1] = b2[1];
        } else {
          bv[buf][0] = bp[0];
        }
      };
#pragma unroll
      for (int xi = 0; xi < BD; ++xi) loadb(xi, xi % BR);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (xi + BD < 16) loadb(xi + BD, (xi + BD) % BR);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[xi][m * NT + j] = WSL_MFMA16(wino_pick(va[m], vb[m], xi), bv[xi % BR][j], acc[xi][m * NT + j]);
        WSL_SCHED_BARRIER();
      }
    }
    __syncthreads();
  }

  wino2_epilogue<C, TH, TW, NT>(p, acc, in_t, n, co0, y0, x0, tile_id, nb, cby);
}

// conv_wino2_kernel for launches whose sources carry no loader transform (every data-gradient launch): the raw tile and the
// filter block need no VGPR round trip -- global_load_lds_dwordx4 streams both into LDS (lane-contiguous layouts: a thread's
// tile slot is 16 * tid bytes into its plane group, the filter block is linear), double-buffered, so a chunk costs the
// input transform + MFMAs and ONE barrier: no staging VALU, no LDS writes, no prefetch registers.
// Same shifted image and plane pitch as Wino2Cfg (the DMA lands 4 bytes past a 16-byte boundary: tools/probe_lds5.hip, probe 2).  A DMA
// instruction writes lane l of a wave at the wave's base + 16 l bytes, so a plane must start at a wave boundary to have its own pitch:
// with two planes per pass (G == 2) waves 0-1 stage the first and waves 2-3 the second (positions 64 (wave & 1) + lane < POS).
template <int TH, int TW, int NT>
struct Wino2RCfg : Wino2Cfg<TH, TW, NT> {
  using B = Wino2Cfg<TH, TW, NT>;
  static constexpr size_t SMEM = sizeof(float) * (2 * B::IN_FLOATS + 2 * B::W_FLOATS);
  static_assert((B::G == 1 || (B::G == 2 && B::POS <= 128)) && 8 * B::CO_T <= B::IN_FLOATS, "lane-contiguous tile layout");
};

// (ABL: compile-time phase ablations of the experiments build, env WSL_WINO2R_ABLATE -- 1 no MFMAs, 2 no DMA after the first
//  chunk, 4 no epilogue, 8 no input-patch reads from LDS after the first chunk; wrong results by design; tools/abl_wino2r.sh)
template <int TH, int TW, int NT, bool SD = false WSL_ABL_TPARAM>
__global__ __launch_bounds__(256, (Wino2Cfg<TH, TW, NT>::MINW)) void conv_wino2r_kernel(WinoP p) {
  WSL_ABL_CONST   // (product build: no ablation parameter, the arms below fold away)
  using C = Wino2RCfg<TH, TW, NT>;
  constexpr int KC = C::KC, CO_T = C::CO_T, MTW = C::MTW;
  WSL_DYN_SMEM(smem);
  float* in_b = reinterpret_cast<float*>(smem);      // two raw halo tiles [KC][PLANE]
  float* w_b = in_b + 2 * C::IN_FLOATS;              // two filter blocks (operand order of conv_wino2_kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);   // XCD-aware tile order, as conv_mfma2_kernel
  const int tile_id = bid;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int cby = blockIdx.y;
  const int co0 = cby * CO_T;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = p.H, W = p.W, Ci = p.Ci, Co = p.Co;
  const int HW = H * W;

  const int grp = C::G == 2 ? (wave >> 1) : 0, pos = C::G == 2 ? ((wave & 1) * 64 + lane) : tid;   // (plane of the pass, float4 position)
  const int pty = pos / C::ROWP4, ptx4 = pos - pty * C::ROWP4;
  const int gy = y0 + pty - 1, gx = x0 + ptx4 * 4 - C::PADL;
  const bool owner = pos < C::POS;
  const bool pvalid = owner && gy >= 0 && gy < H && gx >= 0 && gx < W;
  const uint32_t toff = pvalid ? (uint32_t)(grp * HW + gy * W + gx) : 0u;
  const int64_t gstride = (int64_t)C::G * HW;
  const float* xa_n = p.a.x + n * p.a.bs;
  const float* xb_n = p.b.C ? p.b.x + n * p.b.bs : nullptr;
  const int64_t w_cstride = (int64_t)(Co / CO_T) * C::W_FLOATS;
  // this wave's slots of a pass: lane l at wslot + 4 l floats (the image is shifted by one float)
  const int wslot = grp * C::PLANE + (pos - lane) * 4 + C::SHIFT;

  // slots of positions outside the image stay zero in both buffers for the whole kernel: the DMA never touches them
  if (owner && !pvalid) {
#pragma unroll
    for (int bsel = 0; bsel < 2; ++bsel)
#pragma unroll
      for (int i = 0; i < C::NLD; ++i)
        wino_store_shifted(in_b + bsel * C::IN_FLOATS + i * (C::G * C::PLANE) + wslot - C::SHIFT + 4 * lane, 0.f, 0.f, 0.f, 0.f);
  }
  auto issue = [&](int c0, int bsel) __attribute__((always_inline)) {
    const bool ina = c0 < p.a.C;                                   // uniform
    const int chb = ina ? c0 : c0 - p.a.C;
    const float* xb = (ina ? xa_n : xb_n) + (int64_t)chb * HW;
    float* dst = in_b + bsel * C::IN_FLOATS + wslot;               // wave-uniform: lane l lands at dst + 4 * l floats
    // (the DMAs are issued from inline assembly -- wsl_rt.h -- so that hipcc does not wait for them, vmcnt(0): it must assume that any later
    //  LDS read aliases their destination, in front of THIS chunk's first LDS reads: with the builtin the double buffering never overlapped,
    //  +0.7 % on the f32 step.  The wait before the barrier is WSL_WAIT_ALL)
    if (pvalid) {
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) WSL_LDS_DMA16_UNTRACKED_SO(xb + i * gstride, (uint32_t)toff * 4u, dst + i * (C::G * C::PLANE));
    }
    const float* wb = p.u + (int64_t)cby * C::W_FLOATS + (c0 / KC) * w_cstride;
    float* wdst = w_b + bsel * C::W_FLOATS + wave * 256;
#pragma unroll
    for (int i = 0; i < C::NWL; ++i) WSL_LDS_DMA16_UNTRACKED_SO(wb + i * (4 * kThreads), (uint32_t)tid * 16u, wdst + i * (4 * kThreads));
  };

  // (experiments build, -DWSL_WINO_V=bits via `build.sh expvar`: 1 = the next chunk's DMA pieces are spread over the matrix loop instead of
  //  issued as one burst at its head, 2 = no scheduling fences in the matrix loop, 4 = s_setprio 1 around the matrix loop)
#ifndef WSL_WINO_V
#define WSL_WINO_V 0
#endif
  auto issue_piece = [&](int c0, int bsel, int k) __attribute__((always_inline)) {
    if (k < C::NLD) {
      const bool ina = c0 < p.a.C;
      const int chb = ina ? c0 : c0 - p.a.C;
      const float* xb = (ina ? xa_n : xb_n) + (int64_t)chb * HW;
      float* dst = in_b + bsel * C::IN_FLOATS + wslot;
      if (pvalid) WSL_LDS_DMA16_UNTRACKED_SO(xb + k * gstride, (uint32_t)toff * 4u, dst + k * (C::G * C::PLANE));
    } else {
      const int i = k - C::NLD;
      const float* wb = p.u + (int64_t)cby * C::W_FLOATS + (c0 / KC) * w_cstride;
      float* wdst = w_b + bsel * C::W_FLOATS + wave * 256;
      WSL_LDS_DMA16_UNTRACKED_SO(wb + i * (4 * kThreads), (uint32_t)tid * 16u, wdst + i * (4 * kThreads));
    }
  };
  issue(0, 0);
  // data-gradient launches with the BatchNorm-backward statistics epilogue: its y / keep-mask reads are issued HERE, a whole
  // channel loop ahead of their use (this kernel has the 40 registers; the epilogue would otherwise sit out their latency)
  float4 ypre[4 * C::NA];
  uint32_t mpre[4 * C::NA];
  const bool bn_epi = SD || p.bn.part != nullptr;   // (an SD launch always carries the BatchNorm-backward epilogue)
  if (bn_epi) {
    const int64_t corner = wino2_corner(n, (int64_t)Co * HW, co0, HW, y0, W, x0);                      // wave-uniform
    const char* const yc = reinterpret_cast<const char*>(p.bn.y + corner);
    const uint8_t* const mc = p.bn.emask ? p.bn.emask + corner : nullptr;
#pragma unroll
    for (int q = 0; q < 4 * C::NA; ++q) {
      const uint32_t off = wino2_out_lane_off<C, NT>(HW, W, q);
      ypre[q] = *reinterpret_cast<const float4*>(yc + 4u * off);
      mpre[q] = mc ? *reinterpret_cast<const uint32_t*>(mc + off) : 0u;
    }
  }
  v4f acc[16][C::NA];   // [xi][m * NT + j]; first written by the first chunk's MFMAs
  int poff[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int t = (wave * MTW + m) * 16 + (lane & 15), tyy = t / C::TTX, txx = t - tyy * C::TTX;
    poff[m] = (lane >> 4) * C::PLANE + (2 * tyy) * C::ROWP + (C::PADL - 1 + C::SHIFT) + 2 * txx;   // even
  }
  const int b_off = lane * NT;
  WSL_WAIT_ALL();
  __syncthreads();   // first chunk landed, zero slots visible

  // one chunk; FIRST: the accumulators start from the MFMA's zero C operand (no 128-register clear)
  auto chunk = [&](int c0, int bsel, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const bool more = (ABL & 2) == 0 && c0 + KC < Ci;
    if ((WSL_WINO_V & 1) == 0 && more) issue(c0 + KC, bsel ^ 1);   // the next chunk streams into the other buffers during this one's compute
    const float* in_t = in_b + bsel * C::IN_FLOATS;
    const float* w_t = w_b + bsel * C::W_FLOATS;
    wsl_v2f rlo[MTW][4], rhi[MTW][4];   // the 4 x 4 patches, one row per pair of register pairs
    auto fetch = [&](int kg) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // (even float offset: two ds_read_b64, conflict-free -- see Wino2Cfg::PLANE)
          const float* r = in_t + kg * (4 * C::PLANE) + poff[m] + i * C::ROWP;
          rlo[m][i] = wino_read_pair(r), rhi[m][i] = wino_read_pair(r + 2);
        }
    };
    if constexpr ((ABL & 8) == 0 || FIRST) {
      fetch(0);
    } else {   // (ablation: patch values that are not compile-time constants, no LDS read)
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) rlo[m][i] = wsl_v2f{(float)lane, (float)i}, rhi[m][i] = wsl_v2f{(float)wave, (float)lane};
    }
    wsl_v2f va[MTW][4], vb[MTW][4];   // V[4 i + {0, 3}] = va[i], V[4 i + {1, 2}] = vb[i]: 16 packed adds (wsl_rt.h)
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      // (ablation 16: the second channel group re-uses the first one's transformed patches -- HALF the input transforms and patch reads per
      //  MFMA, the upper bound of what a 64-channel output block (NT = 4) could save without its occupancy cost; wrong results by design)
      if ((ABL & 16) == 0 || kg == 0) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) wino_btdb_pk(rlo[m], rhi[m], va[m], vb[m]);
      }
      if constexpr ((ABL & 8) == 0 && (ABL & 16) == 0) {
        if (kg == 0) fetch(1);
      }
      constexpr int BD = NT == 2 ? 4 : 2, BR = BD + 1;
      float bv[BR][NT];
      auto loadb = [&](int xi, int buf) __attribute__((always_inline)) {
        const float* bp = w_t + xi * (4 * 16 * 2 * NT) + kg * (4 * 16 * NT) + b_off;
        if constexpr (NT == 2) {
          const wsl_v2f b2 = wino_read_pair(bp);   // (never fused into ds_read2st64_b64)
          bv[buf][0] = b2[0], bv[buf][1] = b2[1];
        } else {
          bv[buf][0] = bp[0];
        }
      };
#pragma unroll
      for (int xi = 0; xi < BD; ++xi) loadb(xi, xi % BR);
#if (WSL_WINO_V & 4)
      if (kg == 0) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (xi + BD < 16) loadb(xi + BD, (xi + BD) % BR);
        if ((WSL_WINO_V & 1) != 0 && more) {   // one DMA piece every few transform positions (compile-time slots after unrolling)
          constexpr int NP = C::NLD + C::NWL;
#pragma unroll
          for (int k = 0; k < NP; ++k)
            if (kg * 16 + xi == (k * 32) / NP) issue_piece(c0 + KC, bsel ^ 1, k);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
            const v4f cin = (FIRST && kg == 0) ? zero4 : acc[xi][m * NT + j];
            if constexpr ((ABL & 1) != 0) acc[xi][m * NT + j][0] = cin[0] + wino_pick(va[m], vb[m], xi) * bv[xi % BR][j];
            else acc[xi][m * NT + j] = WSL_MFMA16(wino_pick(va[m], vb[m], xi), bv[xi % BR][j], cin);
          }
#if !(WSL_WINO_V & 2)
        WSL_SCHED_BARRIER();
#endif
      }
    }
#if (WSL_WINO_V & 4)
    __builtin_amdgcn_s_setprio(0);
#endif
    WSL_WAIT_ALL();    // the next chunk's DMA has landed
    __syncthreads();   // ... and every wave is done with this chunk's buffers
  };
  chunk(0, 0, std::true_type{});
  for (int c0 = KC, bsel = 1; c0 < Ci; c0 += KC, bsel ^= 1) chunk(c0, bsel, std::false_type{});
  if constexpr ((ABL & 4) != 0) {
    float t = 0.f;   // (keeps every accumulator alive)
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int a = 0; a < C::NA; ++a) t += (acc[xi][a][0] + acc[xi][a][1]) + (acc[xi][a][2] + acc[xi][a][3]);
    if (t == 123.456f) p.y[0] = t;
    return;
  }
  if constexpr (SD) {
    wino2_epilogue<C, TH, TW, NT, true, 0, true>(p, acc, in_b, n, co0, y0, x0, tile_id, nb, cby, ypre, mpre);
  } else {
    if (bn_epi) wino2_epilogue<C, TH, TW, NT, true, (ABL & 224)>(p, acc, in_b, n, co0, y0, x0, tile_id, nb, cby, ypre, mpre);
    else wino2_epilogue<C, TH, TW, NT, false, (ABL & 224)>(p, acc, in_b, n, co0, y0, x0, tile_id, nb, cby);
  }
}

// ------------------------------------------------------------------------------------------------ filter transform
// U(xi = 4 r + c, ci, co) = (G g G^T)[r][c], g = w[co][ci][:, :] (forward) or the flipped, transposed filter of the
// data gradient (w[ci][co][2-ky][2-kx] with the GEMM's in/out roles swapped, as pack_weights_kernel's wmode 1), stored in
// the order the kernel's B-operand reads want (below): a chunk's block is one contiguous 16 * 8 * co_t float copy.
__device__ __forceinline__ void wino_filter(const float* w, float* u, int Co, int Ci, int dgrad, int64_t i) {
  const int co = (int)(i % Co), ci = (int)(i / Co);
  float g[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
    g[tap] = dgrad ? w[((int64_t)ci * Co + co) * 9 + (8 - tap)] : w[((int64_t)co * Ci + ci) * 9 + tap];
  float t[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[0][c] = g[c];
    t[1][c] = 0.5f * ((g[c] + g[3 + c]) + g[6 + c]);
    t[2][c] = 0.5f * ((g[c] - g[3 + c]) + g[6 + c]);
    t[3][c] = g[6 + c];
  }
  // operand order of the conv kernels: [chunk = ci / 8][channel block = co / co_t][xi][kg = (ci / 4) % 2][k = ci % 4]
  //                                     [col = co % 16][j = (co % co_t) / 16], co_t = 32 (16 when Co % 32 != 0)
  const int co_t = (Co % 32 == 0) ? 32 : 16, nt = co_t / 16;
  const int chunk = ci >> 3, kg = (ci >> 2) & 1, k = ci & 3, cob = co / co_t, j = (co % co_t) >> 4, col = co & 15;
  float* dst = u + ((int64_t)chunk * (Co / co_t) + cob) * (16 * 8 * co_t) + kg * (64 * nt) + (k * 16 + col) * nt + j;
  const int xs = 4 * 16 * 2 * nt;   // floats per transform position
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dst[(4 * r + 0) * xs] = t[r][0];
    dst[(4 * r + 1) * xs] = 0.5f * ((t[r][0] + t[r][1]) + t[r][2]);
    dst[(4 * r + 2) * xs] = 0.5f * ((t[r][0] - t[r][1]) + t[r][2]);
    dst[(4 * r + 3) * xs] = t[r][2];
  }
}

__global__ __launch_bounds__(256) void wino_pack_kernel(const float* w, float* u, int Co, int Ci, int dgrad) {
  const int64_t total = (int64_t)Ci * Co;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads)
    wino_filter(w, u, Co, Ci, dgrad, i);
}

// every 3x3 layer of a network in ONE launch: blockIdx.y = layer, blockIdx.z = 0 forward image / 1 data-gradient image;
// the images live at twice the raw weight's arena offset (16/9 of its size).  A workgroup builds one (chunk, channel block) = one
// contiguous 16 x 8 x co_t block of the image at a time: its 8 x co_t (ci, co) pairs read their nine taps in runs along the raw tensor's
// fastest axis, the sixteen transformed values go to their operand slots in LDS, and the block leaves as a linear float4 copy
// (round 5: the one-pair-per-thread form wrote 34 MB per step in scattered 4-byte stores, 50 us; wino_filter above keeps that form for
// the single-layer entry point)
__global__ __launch_bounds__(256) void wino_pack_table_kernel(PackTable t, const float* params, float* uf, float* ud) {
  __shared__ __attribute__((aligned(16))) float blk[16 * 8 * 32];
  const PackEntry e = t.e[blockIdx.y];
  if (e.KK != 9) return;
  const int dgrad = blockIdx.z;
  const int Co = dgrad ? e.Ci : e.Co, Ci = dgrad ? e.Co : e.Ci;
  if (Ci % 8 || Co % 16) return;   // not a Winograd layer in this direction (the blocked image needs whole blocks)
  const int co_t = (Co % 32 == 0) ? 32 : 16, nt = co_t / 16, ncb = Co / co_t, nblocks = (Ci / 8) * ncb;
  const float* w = params + e.w;
  float* u = (dgrad ? ud : uf) + 2 * e.w;
  const int tid = threadIdx.x;
  // forward: w[co][ci][tap] -- ci fastest among the threads; data gradient: w[ci][co][8 - tap] with the roles swapped -- co fastest
  const int ci_l = dgrad ? tid / co_t : tid % 8, co_l = dgrad ? tid % co_t : tid / 8;
  const bool live = tid < 8 * co_t;
  const int xs = 4 * 16 * 2 * nt;   // floats per transform position
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int chunk = b / ncb, cob = b - chunk * ncb;
    if (live) {
      const int ci = chunk * 8 + ci_l, co = cob * co_t + co_l;
      float g[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        g[tap] = dgrad ? w[((int64_t)ci * Co + co) * 9 + (8 - tap)] : w[((int64_t)co * Ci + ci) * 9 + tap];
      float tt[4][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tt[0][c] = g[c];
        tt[1][c] = 0.5f * ((g[c] + g[3 + c]) + g[6 + c]);
        tt[2][c] = 0.5f * ((g[c] - g[3 + c]) + g[6 + c]);
        tt[3][c] = g[6 + c];
      }
      const int kg = (ci_l >> 2) & 1, k = ci_l & 3, j = co_l >> 4, col = co_l & 15;
      float* dst = blk + kg * (64 * nt) + (k * 16 + col) * nt + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dst[(4 * r + 0) * xs] = tt[r][0];
        dst[(4 * r + 1) * xs] = 0.5f * ((tt[r][0] + tt[r][1]) + tt[r][2]);
        dst[(4 * r + 2) * xs] = 0.5f * ((tt[r][0] - tt[r][1]) + tt[r][2]);
        dst[(4 * r + 3) * xs] = tt[r][2];
      }
    }
    __syncthreads();
    float* out = u + (int64_t)b * (16 * 8 * co_t);
    for (int q = tid; q < 16 * 8 * co_t / 4; q += kThreads) reinterpret_cast<float4*>(out)[q] = reinterpret_cast<const float4*>(blk)[q];
    __syncthreads();
  }
}

int wino_pack_table(const PackTable& t, const float* params, float* uf, float* ud, int with_dgrad, void* stream) {
  double e = 0.0;
  for (int i = 0; i < t.n; ++i) e += t.e[i].KK == 9 ? (double)t.e[i].Co * t.e[i].Ci : 0.0;
  ProfScope ps(PF_PREP, 0.0, 4.0 * e * (9.0 + 16.0 * (with_dgrad ? 2.0 : 1.0)), stream);
  WSL_LAUNCH(wino_pack_table_kernel, dim3(32, t.n, with_dgrad ? 2 : 1), dim3(kThreads), 0, stream, t, params, uf, ud);
  return check_launch("wino_pack_table_kernel");
}

int wino_pack(const float* w, float* u, int Co, int Ci, int dgrad, void* stream) {
  if (Ci % 8 || Co % 16) {
    set_error("conv2d_pack_weights: the Winograd image needs Ci %% 8 == 0 and Co %% 16 == 0 (got %d, %d)", Ci, Co);
    return WSL_EINVAL;
  }
  int64_t blocks = ((int64_t)Ci * Co + kThreads - 1) / kThreads;
  if (blocks > 256) blocks = 256;
  WSL_LAUNCH(wino_pack_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, w, u, Co, Ci, dgrad);
  return check_launch("wino_pack_kernel");
}

// ------------------------------------------------------------------------------------------------ launch
static WSrc to_wsrc(const WslSrc& s) { return WSrc{s.x, s.emask, s.scale, s.shift, s.cmask, s.bs, s.C, s.emask_scale}; }

// shapes the Winograd kernels take; the tile (th x tw pixels) is also what the direct kernels use for such a layer, so
// the number of BatchNorm partial blocks does not depend on which of the two runs (wsl_conv2d_stat_blocks)
// (16-channel blocks keep the staging + input transform of a 32-channel block for half the matrix work: measured slower
// than the direct kernel on the 256x256 layers, so they are taken only when asked for -- WSL_CONV_WINO=2)
bool wino_shape_ok(int H, int W, int Ci, int Co, int ks, bool allow16, int* th, int* tw, int* co_t) {
  if (ks != 3 || Ci % 8 || Ci > 256 || Co % (allow16 ? 16 : 32) || H <= 0 || W <= 0) return false;
  int h = 0, w = 0;
  static const int tile16 = WSL_TUNE("WSL_WINO16_TILE", 64);   // (experiments build: 32 = 64 tiles x 16 channels per workgroup)
  if (Co == 16 && W % 64 == 0 && H % 8 == 0 && tile16 == 64) h = 8, w = 64;   // 16 channels: 128 tiles per workgroup (second form)
  else if (W % 32 == 0 && H % 8 == 0) h = 8, w = 32;
  else if (W % 16 == 0 && H % 16 == 0) h = 16, w = 16;
  else return false;
  if ((int64_t)Ci * H * W >= (int64_t(1) << 31) || (int64_t)16 * Ci * Co >= (int64_t(1) << 31)) return false;
  if ((int64_t)32 * H * W >= (int64_t(1) << 30)) return false;   // the epilogue's 32-bit byte offsets inside a channel block (wino2_out_lane_off)
  if (Co % 32 && !(h == 8 && (w == 64 || w == 32))) return false;   // 16-channel blocks: 8 x 64 / 8 x 32 tiles only (else direct kernels)
  if (th) *th = h;
  if (tw) *tw = w;
  if (co_t) *co_t = Co % 32 == 0 ? 32 : 16;
  return true;
}

// the loader's keep-mask reads (one byte per element of a source with an nn.Dropout mask): part of a forward launch's algorithmic bytes
static inline double wsrc_mask_bytes(const WinoP& p, double px) { return px * ((p.a.emask ? p.a.C : 0) + (p.b.C && p.b.emask ? p.b.C : 0)); }

template <int TH, int TW, int NT>
static int launch_wino2r(WinoP& p, int is_dgrad, void* stream) {
  using C = Wino2RCfg<TH, TW, NT>;
  auto kern = (p.bn.part && p.bn.store_d) ? conv_wino2r_kernel<TH, TW, NT, true> : conv_wino2r_kernel<TH, TW, NT>;
#ifdef WSL_EXPERIMENTS
  if (!(p.bn.part && p.bn.store_d)) {
    static const int abl = WSL_TUNE("WSL_WINO2R_ABLATE", 0);
    if (abl == 1) kern = conv_wino2r_kernel<TH, TW, NT, false, 1>;
    if (abl == 2) kern = conv_wino2r_kernel<TH, TW, NT, false, 2>;
    if (abl == 4) kern = conv_wino2r_kernel<TH, TW, NT, false, 4>;
    if (abl == 6) kern = conv_wino2r_kernel<TH, TW, NT, false, 6>;     // channel loop only, no memory traffic
    if (abl == 7) kern = conv_wino2r_kernel<TH, TW, NT, false, 7>;     // ... without MFMAs
    if (abl == 14) kern = conv_wino2r_kernel<TH, TW, NT, false, 14>;   // ... without operand reads (transforms + MFMAs)
    if (abl == 3) kern = conv_wino2r_kernel<TH, TW, NT, false, 3>;
    if (abl == 128) kern = conv_wino2r_kernel<TH, TW, NT, false, 128>; // epilogue stores in the address pattern of a lane-transposed epilogue
    if (abl == 32) kern = conv_wino2r_kernel<TH, TW, NT, false, 32>;   // epilogue without its global stores
    if (abl == 64) kern = conv_wino2r_kernel<TH, TW, NT, false, 64>;   // epilogue without its statistics
    if (abl == 96) kern = conv_wino2r_kernel<TH, TW, NT, false, 96>;   // epilogue = output transform only
    if (abl == 16) kern = conv_wino2r_kernel<TH, TW, NT, false, 16>;   // half the input transforms per MFMA (NT = 4's upper bound)
    if (abl == 20) kern = conv_wino2r_kernel<TH, TW, NT, false, 20>;   // ... and no epilogue
  }
#endif
#ifdef WSL_EXPERIMENTS
  (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);   // (the ablation variants: whichever was picked)
#else
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM((conv_wino2r_kernel<TH, TW, NT>), C::SMEM);
    (void)WSL_SET_MAX_DYN_SMEM((conv_wino2r_kernel<TH, TW, NT, true>), C::SMEM);
    attr_done = true;
  }
#endif
  dim3 grid(p.tiles_x * p.tiles_y * p.N, p.Co / C::CO_T);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(is_dgrad ? PF_WINO_DGRAD : PF_WINO_FWD, 2.0 * px * p.Co * p.Ci * 9, 4.0 * px * (p.Co + p.Ci) + bn_epi_bytes(p.bn, px * p.Co) + wsrc_mask_bytes(p, px), stream,
                         2.0 * px * p.Co * p.Ci * 4);   // issued: 16 instead of 36 multiply-adds per 2x2 tile
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_wino2r_kernel");
}

template <int TH, int TW, int NT>
static int launch_wino2(WinoP& p, int is_dgrad, void* stream) {
  static const bool dma_on = (WSL_TUNE("WSL_WINO_DMA", 1) != 0);
  const bool raw = !p.a.scale && !p.a.emask && !p.a.cmask && (p.b.C == 0 || (!p.b.scale && !p.b.emask && !p.b.cmask));
  if (dma_on && raw) return launch_wino2r<TH, TW, NT>(p, is_dgrad, stream);   // plain sources: LDS DMA
  using C = Wino2Cfg<TH, TW, NT>;
  auto kern = conv_wino2_kernel<TH, TW, NT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, p.Co / C::CO_T);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(is_dgrad ? PF_WINO_DGRAD : PF_WINO_FWD, 2.0 * px * p.Co * p.Ci * 9, 4.0 * px * (p.Co + p.Ci) + bn_epi_bytes(p.bn, px * p.Co) + wsrc_mask_bytes(p, px), stream,
                         2.0 * px * p.Co * p.Ci * 4);   // issued: 16 instead of 36 multiply-adds per 2x2 tile
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_wino2_kernel");
}

int wino_fwd(const WslSrc& a, const WslSrc* b, const float* u, const float* bias, float* y, int64_t y_bs, int N, int H,
             int W, int Co, int is_dgrad, float* stat_part, float* stat_cnt, int slots, void* stream, const BnBwdEpi* bn,
             int* bn_done) {
  WinoP p;
  p.a = to_wsrc(a);
  p.b = (b && b->C > 0) ? to_wsrc(*b) : WSrc{};
  p.u = u, p.bias = bias, p.y = y, p.y_bs = y_bs;
  p.N = N, p.H = H, p.W = W, p.Ci = a.C + p.b.C, p.Co = Co;
  p.stat_part = stat_part, p.stat_cnt = stat_cnt, p.slots = slots;
  static const int ablate = WSL_TUNE("WSL_CONV_ABLATE", 0);
  p.ablate = ablate;
  int th = 0, tw = 0, co_t = 0;
  if (!wino_shape_ok(H, W, p.Ci, Co, 3, true, &th, &tw, &co_t) || (p.b.C && (p.a.C % 8))) {
    set_error("conv2d_fwd: shape N=%d H=%d W=%d Ci=%d(+%d) Co=%d is not a Winograd shape", N, H, W, p.a.C, p.b.C, Co);
    return WSL_EINVAL;
  }
#ifdef WSL_EXPERIMENTS
  // WIDE tiles (round 6, measured and not kept: profiles/r6_wino2r_ablations.md section 5; experiments build only): the same number of
  // pixels per workgroup as 8 x 64 / 8 x 32 -- so the same BatchNorm partial count as the plan the direct kernels and
  // wsl_conv2d_stat_blocks() use -- but rows twice as long: 512- / 256-byte row segments instead of 256 / 128 (tools/probe_tile_loads.hip:
  // the pure halo fetch of a 16-channel 256^2 tensor takes 86 us in 8 x 64 tiles and 72 in 4 x 128; in the kernels it buys nothing)
  static const int wide16 = WSL_TUNE("WSL_WINO16_WIDE", 0), wide32 = WSL_TUNE("WSL_WINO32_WIDE", 0);   // (wide32: minimum image width)
  const bool w16 = wide16 && co_t == 16 && th == 8 && tw == 64 && W % 128 == 0 && H % 4 == 0;
  const bool w32 = wide32 && co_t == 32 && th == 8 && tw == 32 && W % 64 == 0 && H % 4 == 0 && W >= wide32;
  if (w16) th = 4, tw = 128;
  if (w32) th = 4, tw = 64;
#endif
  p.tiles_x = W / tw, p.tiles_y = H / th;
  // every shape wino_shape_ok() admits has a kernel with the BatchNorm-backward statistics in its epilogue
  const bool narrow16 = co_t == 16 && th == 8 && tw == 32;   // 64 tiles x 16 channels: one accumulator set, 3 workgroups per CU
  const bool bn_ok = bn && bn->part;
  if (bn_ok) p.bn = *bn;
  // d instead of g (BnBwdEpi::store_d) is written by the raw-source kernel only: it has the consumer's y / keep bytes in registers when it stores
  static const bool dma_on = (WSL_TUNE("WSL_WINO_DMA", 1) != 0);
  const bool raw_src = !p.a.scale && !p.a.emask && !p.a.cmask && (p.b.C == 0 || (!p.b.scale && !p.b.emask && !p.b.cmask));
  if (!bn_ok || !(raw_src && dma_on)) p.bn.store_d = false;
  if (bn_done) *bn_done = bn_ok ? (p.bn.store_d ? 2 : 1) : 0;
#ifdef WSL_EXPERIMENTS
  if (w16) return launch_wino2<4, 128, 1>(p, is_dgrad, stream);
  if (w32) return launch_wino2<4, 64, 2>(p, is_dgrad, stream);
#endif
  if (tw == 64) return launch_wino2<8, 64, 1>(p, is_dgrad, stream);
  if (narrow16) return launch_wino2<8, 32, 1>(p, is_dgrad, stream);
  if (th == 8) return launch_wino2<8, 32, 2>(p, is_dgrad, stream);
  return launch_wino2<16, 16, 2>(p, is_dgrad, stream);
}

// ================================================================================================ Winograd weight gradient
// The adjoint of the kernel above:  dL/dg = G^T [ sum over tiles (A dY A^T) .* (B^T d B) ] G, i.e. per tile and channel
// pair 16 multiply-adds instead of the direct form's 36.  The transforms never touch LDS: lane (c = l & 15, t = l >> 4) of
// a wave reads the 2x2 output gradients of channels (c, 16 + c) and the 4x4 input patch of ONE input channel for tile t
// of a group of four from the staged raw tiles, forms Z = A dY A^T (12 adds each) and V = B^T d B (32 adds) in registers --
// which IS the operand layout of v_mfma_f32_16x16x4_f32 with K = the four tiles (A operand: Z[xi][co = c][k = t], B operand:
// V[xi][k = t][ci = c]) -- and issues 16 x 2 MFMAs: M_xi[32 co][16 ci] += Z_xi V_xi^T.  56 vector instructions per 32
// MFMAs.  A workgroup = 32 x 32 channels x (4 x 32 | 8 x 16) pixels per step of its persistent tile loop (same plan,
// splits and partial layout as wgrad_mfma2s_kernel: wgrad_reduce_kernel is unchanged); wave = (input-channel tile, half
// of the tile groups).  Epilogue: G^T M G per lane (all 16 positions of a channel pair sit in one lane), the two halves
// merged through LDS in a fixed order, 9 taps stored; db = sum of Z[1][1] (= the 2x2 sum of dY).
struct WgWinoP {
  WSrc a, b;
  const float* dy;
  int64_t dy_bs;
  float* part_dw;   // [nsplit][9][Co][Ci]
  float* part_db;   // [nsplit][Co]
  int N, H, W, Ci, Co, tiles_x, tiles_y, items, nsplit, co_blocks, interleave;
};

template <int TH, int TW, int NCO, int NCI, int NW>   // NCO dY channel tiles per wave, NCI input-channel tiles and NW waves per workgroup
struct WgWinoCfg {
  static constexpr int THREADS = 64 * NW, CB = 16 * NCO, IB = 16 * NCI, NSUB = NW / NCI, PADL = 4;
  static constexpr int ROWP = TW + 2 * PADL, ROWS = TH + 2, ROWP4 = ROWP / 4;
  // input staging map as Wino2Cfg: whole rows per wave (RW), WPP waves per plane, GA planes per pass, NA passes
  static constexpr int RW = 64 / ROWP4, WPP = (ROWS + RW - 1) / RW, GA = NW / WPP, NA = IB / GA;
  static constexpr int SD = TH * TW, PD = SD / 4, GD = THREADS / PD, ND = CB / GD;
  // Plane pitches == 4 (mod 64) floats: a 32-lane group = 16 channels x 2 tiles reads 8 bytes per lane at 16 c + 8 t bytes (mod 256):
  // every bank once (rounds 1-4: == 2 (mod 32) and, for the input patches, ds_read2_b32 at odd offsets: 2-way conflicts on every operand
  // read; profiles/r5_probe_lds5.md).  The input image is shifted by one float like the conv kernels' (Wino2Cfg::SHIFT).
  static constexpr int SHIFT = 1;
  static constexpr int PLD = ((SD - 4 + 63) / 64) * 64 + 4;
  static constexpr int PLA = ((ROWS * ROWP + SHIFT - 4 + 63) / 64) * 64 + 4;
  static constexpr int TTX = TW / 2, TILES = (TH / 2) * TTX, GROUPS = TILES / 4;
  static constexpr int DY_FLOATS = CB * PLD, A_FLOATS = IB * PLA, BUF_FLOATS = DY_FLOATS + A_FLOATS;
  // NW == 8: one 8-wave workgroup per CU with TWO tile buffers -- the next tile's loads are issued before the compute phase
  // of the current one and written to the other buffer after it (the 4-wave form has no registers left for that and relies
  // on a second resident workgroup to cover its load latency)
  static constexpr int NBUF = NW == 8 ? 2 : 1;
  static constexpr int PER = NCO * 37;                                   // per lane: NCO x (4 x 9 taps + 1 db)
  static constexpr int RED_FLOATS = (NSUB - 1) * NCI * 64 * PER;
  static constexpr int MAIN_FLOATS = NBUF * BUF_FLOATS > RED_FLOATS ? NBUF * BUF_FLOATS : RED_FLOATS;
  static constexpr size_t SMEM = sizeof(float) * (MAIN_FLOATS + 2 * IB);
  static_assert(NW % WPP == 0 && PD <= THREADS && IB % GA == 0 && CB % GD == 0 && GROUPS % NSUB == 0 && TTX % 4 == 0 &&
                    (NCI == 1 || NCI == 2) && (NW == 4 || NW == 8), "tile shape");
};

template <int TH, int TW, int NCO, int NCI, int NW WSL_ABL_TPARAM>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 1 : 2)) void wgrad_wino_kernel(WgWinoP p) {
  WSL_ABL_CONST   // (product build: no ablation parameter, the arms below fold away)
  using C = WgWinoCfg<TH, TW, NCO, NCI, NW>;
  WSL_DYN_SMEM(smem);
  float* tiles = reinterpret_cast<float*>(smem);                    // NBUF x {dy tile, input tile}
  float2* tab = reinterpret_cast<float2*>(tiles + C::MAIN_FLOATS);   // [IB] {scale, shift} of this block's channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Which tiles a workgroup walks.  The hardware deals workgroups to the eight XCDs round-robin in launch order (x fastest) and every
  // XCD has an L2 of its own, so (nsplit % 8 == 0) the workgroups of one XCD take ALL channel blocks of nsplit / 8 consecutive splits,
  // and split s walks the tiles s, s + nsplit, s + 2 nsplit, ...: in every round the workgroups of an XCD stage neighbouring tiles
  // together -- halo rows, the partly used 128-byte lines at a tile's left / right edge and the tiles the channel blocks share come out
  // of that L2.  (Rounds 2-4 gave every split one contiguous run of tiles: by the time a workgroup came back to a halo line, 6 MB of
  // other workgroups' tiles had gone through the 4 MB L2 -- 7 % hits, 2.1 x the algorithmic bytes fetched on the full-resolution
  // layers, which made the kernel HBM-bound there: profiles/r5_wgrad_item_order.md.)
  int blk = blockIdx.x, split = blockIdx.y;
  const bool interleaved = (p.nsplit & 7) == 0 && p.interleave;
  if (interleaved) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, j = lin >> 3;
    blk = j % (int)gridDim.x;
    split = (lin & 7) * (p.nsplit >> 3) + j / (int)gridDim.x;
  }
  const int cb = blk % p.co_blocks, ib = blk / p.co_blocks;
  const int co0 = cb * C::CB, ci0 = ib * C::IB;
  const int H = p.H, W = p.W, Ci = p.Ci;
  const int HW = H * W;

  const bool ina = ci0 < p.a.C;                          // the source of this block of input channels (uniform)
  const WSrc& s = ina ? p.a : p.b;
  const int chb0 = ina ? ci0 : ci0 - p.a.C;
  const bool has_scale = s.scale != nullptr, has_mask = s.emask != nullptr, has_cm = s.cmask != nullptr;
  const float es = s.es;
  for (int c = tid; c < C::IB; c += C::THREADS)
    tab[c] = has_scale ? make_float2(s.scale[chb0 + c], s.shift[chb0 + c]) : make_float2(1.f, 0.f);

  // fixed staging positions of this thread
  const int gd = tid / C::PD, pd = tid - gd * C::PD;             // dy: float4 #pd of the TH x TW tile of channel gd (+ i * GD)
  const int dty = (pd * 4) / TW, dtx = (pd * 4) - dty * TW;
  const int tdconst = gd * HW + dty * W + dtx;
  const int dloff = gd * C::PLD + pd * 4;
  const int ga = wave / C::WPP, arl = lane / C::ROWP4;           // input: plane ga of the pass, row aty, float4 atx4 of the halo tile
  const int atx4 = lane - arl * C::ROWP4, aty = (wave % C::WPP) * C::RW + arl;
  const bool owner_a = arl < C::RW && aty < C::ROWS;
  const int aloff = ga * C::PLA + aty * C::ROWP + atx4 * 4;
  const int64_t dstride = (int64_t)C::GD * HW, astride = (int64_t)C::GA * HW;

  v4f acc[16][NCO];
  float accb[NCO];
#pragma unroll
  for (int jc = 0; jc < NCO; ++jc) {
    accb[jc] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i][jc] = v4f{0.f, 0.f, 0.f, 0.f};
  }
  const int cit = wave % NCI, sub = wave / NCI;
  const bool dbw = (ib == 0) && (p.part_db != nullptr) && cit == 0;
  // this workgroup's tiles: it0 + k * step, k < it_count; (tx, ty, n) advance by the step's own decomposition with carries
  const int it0 = interleaved ? split : (int)((int64_t)split * p.items / p.nsplit);
  const int it_count = interleaved ? (p.items - split + p.nsplit - 1) / p.nsplit : (int)((int64_t)(split + 1) * p.items / p.nsplit) - it0;
  const int step = interleaved ? p.nsplit : 1;
  int nx_tx, nx_ty, nx_n;
  {
    int q = it0;
    nx_tx = q % p.tiles_x;
    q /= p.tiles_x;
    nx_ty = q % p.tiles_y;
    nx_n = q / p.tiles_y;
  }
  const int st_tx = step % p.tiles_x, st_ty = (step / p.tiles_x) % p.tiles_y, st_n = step / (p.tiles_x * p.tiles_y);
  const int c16 = lane & 15, t4 = lane >> 4;

  // ---- staging: global -> registers (issue), registers -> transform -> LDS tile buffer (commit)
  float4 prd[C::ND], pra[C::NA];
  uint32_t prm[C::NA];
  float prc[C::NA];
  bool pr_aok = false;
  auto issue = [&]() __attribute__((always_inline)) {
    const int n = nx_n, y0 = nx_ty * TH, x0 = nx_tx * TW;
    nx_tx += st_tx;
    const int cx = nx_tx >= p.tiles_x ? 1 : 0;
    nx_tx -= cx ? p.tiles_x : 0;
    nx_ty += st_ty + cx;
    const int cy = nx_ty >= p.tiles_y ? 1 : 0;
    nx_ty -= cy ? p.tiles_y : 0;
    nx_n += st_n + cy;
    const float* dyb = p.dy + n * p.dy_bs + (int64_t)co0 * HW + (uint32_t)(tdconst + y0 * W + x0);
#pragma unroll
    for (int i = 0; i < C::ND; ++i) prd[i] = *reinterpret_cast<const float4*>(dyb + i * dstride);
    const int gy = y0 + aty - 1, gx = x0 + atx4 * 4 - C::PADL;
    pr_aok = owner_a && gx >= 0 && gx < W && gy >= 0 && gy < H;
    const uint32_t taoff = pr_aok ? (uint32_t)(ga * HW + gy * W + gx) : 0u;
    const float* xb = s.x + n * s.bs + (int64_t)chb0 * HW;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) pra[i] = *reinterpret_cast<const float4*>(xb + i * astride + taoff);
    if (has_mask) {
      const uint8_t* mb = s.emask + ((int64_t)n * s.C + chb0) * HW;
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prm[i] = *reinterpret_cast<const uint32_t*>(mb + i * astride + taoff);
    }
    // per-channel factor of this position: the channel mask (1 without one), 0 outside the image -- one multiply replaces
    // two selects per element (the out-of-image slots read finite data at offset 0, so the product is a zero)
    if (has_cm) {
      const float* cmb = s.cmask + (int64_t)n * s.C + chb0 + (owner_a ? ga : 0);
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prc[i] = pr_aok ? cmb[i * C::GA] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < C::NA; ++i) prc[i] = pr_aok ? 1.f : 0.f;
    }
  };
  auto commit = [&](float* dy_t, float* a_t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::ND; ++i)   // (element-wise: a whole-struct copy of prd[i] keeps the array on the stack -- scratch round trip)
      *reinterpret_cast<v4f*>(dy_t + i * (C::GD * C::PLD) + dloff) = v4f{prd[i].x, prd[i].y, prd[i].z, prd[i].w};
    {   // (every lane: the lane shift below wants whole waves)
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        wsl_v2f lo = {pra[i].x, pra[i].y}, hi = {pra[i].z, pra[i].w};
        if (has_scale) {
          const float2 t = tab[ga + i * C::GA];
          xform_bn_leaky(lo, hi, t.x, t.y);
        }
        if (has_mask) xform_mask(lo, hi, prm[i], es);
        lo = lo * prc[i], hi = hi * prc[i];
        // (the shifted image: previous lane's last element + this thread's first three, one aligned conflict-free 16-byte store; rows
        //  never straddle waves in this kernel's staging map.  Equal within noise to three stores 4 + 8 + 4 bytes: profiles/r5_wino_lds_layout.md)
        const float left = wsl_prev_lane(hi[1]);
        if (owner_a) *reinterpret_cast<float4*>(a_t + i * (C::GA * C::PLA) + aloff) = make_float4(left, lo[0], lo[1], hi[0]);
      }
    }
  };

  // ---- compute on one tile buffer: this wave's subset of the tile groups.  Raw operands of group g+1 are read from LDS
  // before the MFMAs of group g issue.  Z is formed WITHOUT the two negations of A (rows/columns with index 3 carry the
  // opposite sign: Z'[p][q] = s_p s_q Z[p][q], s_3 = -1); the epilogue puts the signs back into M.
  auto compute = [&](const float* dy_t, const float* a_t) __attribute__((always_inline)) {
    wsl_v2f rdy[NCO][2];     // [jc][row]
    wsl_v2f rlo[4], rhi[4];  // the 4 x 4 input patch, one row per pair of register pairs
    auto fetch = [&](int g) __attribute__((always_inline)) {
      const int tau = g * 4 + t4, tyy = tau / C::TTX, txx = tau - tyy * C::TTX;
#pragma unroll
      for (int jc = 0; jc < NCO; ++jc) {
        const float* dp = dy_t + (jc * 16 + c16) * C::PLD + (2 * tyy) * TW + 2 * txx;
        rdy[jc][0] = wino_read_pair(dp), rdy[jc][1] = wino_read_pair(dp + TW);
      }
      const float* ap = a_t + (cit * 16 + c16) * C::PLA + (2 * tyy) * C::ROWP + (C::PADL - 1 + C::SHIFT) + 2 * txx;
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // (even float offset: two ds_read_b64, conflict-free)
        const float* r = ap + i * C::ROWP;
        rlo[i] = wino_read_pair(r), rhi[i] = wino_read_pair(r + 2);
      }
    };
    constexpr int GN = C::GROUPS / C::NSUB;
    fetch(sub * GN);
#pragma unroll   // (fully: a rolled loop pays 24 register moves per group for the prefetched operands)
    for (int gi = 0; gi < GN; ++gi) {
      if constexpr ((ABL & 8) != 0) {   // (ablation: operands of the first group for every group -- no further LDS reads)
        if (gi > 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rlo[i] += 1.f, rhi[i] += 1.f;
        }
      }
      // Z[4 i + {0, 3}] = zq[i], Z[4 i + {1, 2}] = zm[i]; V likewise in (va, vb): packed adds (wsl_rt.h), 28 instead of 56
      wsl_v2f zq[NCO][4], zm[NCO][4], va[4], vb[4];
#pragma unroll
      for (int jc = 0; jc < NCO; ++jc) {
        zq[jc][0] = rdy[jc][0], zq[jc][3] = rdy[jc][1];   // zq[3] = +dY[1], zm[..][1] = q0 - q1: signs folded
        wino_aya_pk(rdy[jc][0], rdy[jc][1], zq[jc][1], zq[jc][2], zm[jc]);
        if (dbw) accb[jc] += zm[jc][1][0];
      }
      wino_btdb_pk(rlo, rhi, va, vb);
      if constexpr ((ABL & 8) == 0) {
        if (gi + 1 < GN) fetch(sub * GN + gi + 1);
      }
      WSL_SCHED_BARRIER();   // the next group's operand reads stay IN FRONT of this group's MFMAs (the pair reads are volatile: the
                             // scheduler would otherwise sink them behind the matrix block and wait for them right there)
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        const int i = xi >> 2, c = xi & 3;
        const float vv = c == 0 ? va[i][0] : c == 3 ? va[i][1] : c == 1 ? vb[i][0] : vb[i][1];
#pragma unroll
        for (int jc = 0; jc < NCO; ++jc) {
          const float zz = c == 0 ? zq[jc][i][0] : c == 3 ? zq[jc][i][1] : c == 1 ? zm[jc][i][0] : zm[jc][i][1];
          if constexpr ((ABL & 4) != 0) acc[xi][jc][0] += zz * vv;   // (ablation: one vector multiply-add instead of the MFMA)
          else acc[xi][jc] = WSL_MFMA16(zz, vv, acc[xi][jc]);
        }
      }
    }
  };

  __syncthreads();   // BN table visible
  if constexpr (C::NBUF == 1 && (ABL & 32) != 0) {
    // (experiment: the next tile's global loads are in flight during the matrix phase, committed after it)
    if (it_count > 0) {
      issue();
      commit(tiles, tiles + C::DY_FLOATS);
    }
    __syncthreads();
    for (int item = 0; item < it_count; ++item) {
      const bool more = item + 1 < it_count;
      if (more) issue();
      compute(tiles, tiles + C::DY_FLOATS);
      __syncthreads();
      if (more) commit(tiles, tiles + C::DY_FLOATS);
      __syncthreads();
    }
  } else if constexpr (C::NBUF == 1) {
    // a step is ~5000 cycles of matrix work per wave; the co-resident workgroup covers this one's load latency
    for (int item = 0; item < it_count; ++item) {
      // (ABL: compile-time phase ablations of the experiments build, env WSL_WGWINO_ABLATE -- 1 no matrix phase, 2 global loads +
      //  staging only for the first tile, 4 / 8 inside the matrix phase; wrong results by design)
      if ((ABL & 2) == 0 || item == 0) {
        issue();
        commit(tiles, tiles + C::DY_FLOATS);
      }
      __syncthreads();
      if constexpr ((ABL & 1) == 0) compute(tiles, tiles + C::DY_FLOATS);
      __syncthreads();
    }
  } else {
    if (it_count > 0) {
      issue();
      commit(tiles, tiles + C::DY_FLOATS);
    }
    __syncthreads();
    int cur = 0;
    for (int item = 0; item < it_count; ++item, cur ^= 1) {
      const bool more = item + 1 < it_count;
      if (more) issue();                                         // in flight during the compute phase
      compute(tiles + cur * C::BUF_FLOATS, tiles + cur * C::BUF_FLOATS + C::DY_FLOATS);
      if (more) commit(tiles + (cur ^ 1) * C::BUF_FLOATS, tiles + (cur ^ 1) * C::BUF_FLOATS + C::DY_FLOATS);
      __syncthreads();   // the other buffer was last read before the previous barrier
    }
  }

  // ---- epilogue: dg = G^T M G per lane, merge the tile subsets (fixed order), store the 9 taps
  float dg[NCO][4][9];
#pragma unroll
  for (int jc = 0; jc < NCO; ++jc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t[3][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (undo the folded signs: M[p][q] = s_p s_q M'[p][q], s_3 = -1)
        const float sq = q == 3 ? -1.f : 1.f;
        const float m0 = sq * acc[q][jc][r], m1 = sq * acc[4 + q][jc][r], m2 = sq * acc[8 + q][jc][r], m3 = -sq * acc[12 + q][jc][r];
        t[0][q] = m0 + 0.5f * (m1 + m2);
        t[1][q] = 0.5f * (m1 - m2);
        t[2][q] = 0.5f * (m1 + m2) + m3;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dg[jc][r][3 * a + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        dg[jc][r][3 * a + 1] = 0.5f * (t[a][1] - t[a][2]);
        dg[jc][r][3 * a + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
      }
    }
  float dbv[NCO];
#pragma unroll
  for (int jc = 0; jc < NCO; ++jc) {
    float b = accb[jc];
    b += __shfl_xor(b, 16);
    b += __shfl_xor(b, 32);
    dbv[jc] = b;   // sum over this wave's tiles for channel jc * 16 + c16
  }
  float* red = reinterpret_cast<float*>(smem);   // the tiles are dead: every wave passed the loop's last barrier
  if (sub > 0) {
    float* mine = red + (((sub - 1) * NCI + cit) * 64 + lane) * C::PER;
#pragma unroll
    for (int jc = 0; jc < NCO; ++jc) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 9; ++t) mine[(jc * 4 + r) * 9 + t] = dg[jc][r][t];
      mine[NCO * 36 + jc] = dbv[jc];
    }
  }
  __syncthreads();
  if (sub == 0) {
    const int ci = ci0 + cit * 16 + c16;
#pragma unroll
    for (int jc = 0; jc < NCO; ++jc) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + jc * 16 + t4 * 4 + r;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          float sum = dg[jc][r][t];
#pragma unroll
          for (int k = 1; k < C::NSUB; ++k) sum += red[(((k - 1) * NCI + cit) * 64 + lane) * C::PER + (jc * 4 + r) * 9 + t];
          p.part_dw[(((int64_t)split * 9 + t) * p.Co + co) * Ci + ci] = sum;
        }
      }
      if (dbw && lane < 16) {
        float sum = dbv[jc];
#pragma unroll
        for (int k = 1; k < C::NSUB; ++k) sum += red[(((k - 1) * NCI + cit) * 64 + lane) * C::PER + NCO * 36 + jc];
        p.part_db[(int64_t)split * p.Co + co0 + jc * 16 + lane] = sum;
      }
    }
  }
}

template <int TH, int TW, int NCO, int NCI, int NW>
static int launch_wgrad_wino(WgWinoP& p, int ci_blocks, void* stream) {
  using C = WgWinoCfg<TH, TW, NCO, NCI, NW>;
  auto kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW>;
#ifdef WSL_EXPERIMENTS
  if constexpr (NW == 4) {
    static const int abl = WSL_TUNE("WSL_WGWINO_ABLATE", 0);   // (WSL_WGRAD_ABLATE belongs to the direct kernel and disables this one)
    if (abl == 1) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 1>;
    if (abl == 2) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 2>;
    if (abl == 3) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 3>;
    if (abl == 32) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 32>;   // in-workgroup prefetch
    if (abl == 6) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 6>;     // LDS reads + transforms, no MFMA, no global traffic
    if (abl == 10) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 10>;   // transforms + MFMA, no LDS reads, no global traffic
    if (abl == 14) kern = wgrad_wino_kernel<TH, TW, NCO, NCI, NW, 14>;   // transforms only
  }
#endif
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(p.co_blocks * ci_blocks, p.nsplit);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(PF_WGRAD_WINO, 2.0 * px * p.Co * p.Ci * 9, 4.0 * px * (p.Co + p.Ci), stream, 2.0 * px * p.Co * p.Ci * 4);
  WSL_LAUNCH(kern, grid, dim3(C::THREADS), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_wino_kernel");
}

// takes the launches wgrad_mfma2s_kernel would get with the same plan: 3x3, 32 x 32 channel blocks (two dY tiles per wave,
// two input-channel tiles per workgroup) or 16 x 16 (the 16-channel layers: one tile each, four tile subsets)
bool wgrad_wino_ok(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks, int th, int tw, int cb, int ib) {
  static const int on = WSL_TUNE("WSL_WGRAD_WINO", 2);   // 0 off, 1 only 32 x 32 blocks, 2 all
  if (!on || ks != 3 || cb != ib) return false;
  if (cb == 32) {
    if (!((th == 4 && tw == 32) || (th == 8 && tw == 16))) return false;
  } else if (cb == 16 && on >= 2) {
    if (!((th == 4 && tw == 64) || (th == 4 && tw == 32) || (th == 8 && tw == 16))) return false;
  } else {
    return false;
  }
  const int bC = b ? b->C : 0, Ci = a.C + bC;
  if ((Co % cb) || (Ci % ib) || (bC > 0 && (a.C % ib)) || (H % th) || (W % tw)) return false;
  const int64_t span = (int64_t)(a.C > bC ? a.C : bC) * H * W;
  return span < (int64_t(1) << 31) && (int64_t)Co * H * W < (int64_t(1) << 31);
}

int wgrad_wino_launch(const WslSrc& a, const WslSrc* b, const float* dy, int64_t dy_bs, float* part_dw, float* part_db, int N,
                      int H, int W, int Co, int th, int tw, int cb, int nsplit, int items, int tiles_x, int tiles_y,
                      int co_blocks, int ci_blocks, void* stream) {
  WgWinoP p;
  p.a = to_wsrc(a);
  p.b = (b && b->C > 0) ? to_wsrc(*b) : WSrc{};
  p.dy = dy, p.dy_bs = dy_bs, p.part_dw = part_dw, p.part_db = part_db;
  p.N = N, p.H = H, p.W = W, p.Ci = a.C + p.b.C, p.Co = Co;
  p.tiles_x = tiles_x, p.tiles_y = tiles_y, p.items = items, p.nsplit = nsplit, p.co_blocks = co_blocks;
  static const int il = WSL_TUNE("WSL_WGWINO_INTERLEAVE", 1);
  p.interleave = il;
  if (cb == 32) {
    return th == 4 ? launch_wgrad_wino<4, 32, 2, 2, 4>(p, ci_blocks, stream) : launch_wgrad_wino<8, 16, 2, 2, 4>(p, ci_blocks, stream);
  }
  if (tw == 64) return launch_wgrad_wino<4, 64, 1, 1, 4>(p, ci_blocks, stream);
  return th == 4 ? launch_wgrad_wino<4, 32, 1, 1, 4>(p, ci_blocks, stream) : launch_wgrad_wino<8, 16, 1, 1, 4>(p, ci_blocks, stream);
}

}  // namespace wsl
