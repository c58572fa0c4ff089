// Weight gradient of the two layer shapes whose channel counts starve the 16x16 blocking of wgrad_mfma2l_kernel:
//   * the first convolution  (Ci = 1,  Co = 16):  dw[co][0][tap]  = sum_p dy[co][p] * x[p + tap]
//   * the 4-class classifier (Ci = 16, Co = 4):   dw[co][ci][tap] = sum_p a[ci][p] * dy[co][p - tap]
// Both are ONE matrix product per 4 pixels if the 16 rows of the MFMA tile are the 16-channel operand ("A": dy resp. the
// activations) and the columns enumerate (tap, narrow channel) pairs of the other operand ("B": x resp. dy), read from
// an LDS halo tile with a per-lane shift: 1 resp. 3 MFMAs per step where the general kernel issues 9 that are 1/16 resp.
// 1/4 full (profiles/r1f: 261 us per launch for ~0.6 GFLOP).  These launches are HBM-bound (A is a 268 MB tensor).
// Partials go to the same [split][tap][Co][Ci] workspace, reduced by wgrad_reduce_kernel.
#include <type_traits>

#include "wsl_rt.h"

namespace wsl {

struct WgSmallP {
  const float* a;        // 16-channel operand [N,16,H,W]
  int64_t a_bs;
  const float* a_scale;  // its BN+LeakyReLU loader transform, or null (raw)
  const float* a_shift;
  const float* b;        // narrow operand [N,CBN,H,W], raw
  int64_t b_bs;
  float* part_dw;
  float* part_db;
  int N, H, W, tiles_x, tiles_y, items, nsplit;
};

template <int CBN>
struct WgSmallCfg {
  static constexpr int TH = 8, TW = 64, ROWP = TW + 8, ROWS = TH + 2, ROWP4 = ROWP / 4, S = TH * TW;
  static constexpr int PD = S / 4, GD = 256 / PD, ND = 16 / GD;           // A: float4 positions per channel, loads per thread
  static constexpr int PB = ROWS * ROWP4, NBF4 = CBN * PB, NB = (NBF4 + 255) / 256;   // B float4 of a tile, loads per thread
  static constexpr int PLA = ((S - 2 + 31) / 32) * 32 + 2;               // == 2 (mod 32): conflict-free A-operand reads
  static constexpr int PLB = ROWS * ROWP + 4;
  static constexpr int J = 9 * CBN, NT = (J + 15) / 16, TC = (4 * CBN) / 16;   // TC: tile holding the centre-tap columns
  static constexpr int A_FLOATS = 16 * PLA, B_FLOATS = CBN * PLB;
  static constexpr int RED_FLOATS = 4 * 64 * (NT + 1) * 4;
  static constexpr int MAIN_FLOATS = A_FLOATS + B_FLOATS > RED_FLOATS ? A_FLOATS + B_FLOATS : RED_FLOATS;
  static constexpr size_t SMEM = sizeof(float) * (MAIN_FLOATS + 32);
  static_assert(16 % GD == 0 && (4 * CBN) / 16 == (5 * CBN - 1) / 16, "staging shape / centre columns in one tile");
};

// SGN = +1: B is read at p + tap (first convolution: A = dy, B = x);  -1: at p - tap (classifier: A = activations, B = dy)
template <int CBN, int SGN>
__global__ __launch_bounds__(256, 2) void wgrad_small_kernel(WgSmallP p) {
  using C = WgSmallCfg<CBN>;
  WSL_DYN_SMEM(smem);
  float* a_t = reinterpret_cast<float*>(smem);
  float* b_t = a_t + C::A_FLOATS;
  float2* tab = reinterpret_cast<float2*>(a_t + C::MAIN_FLOATS);   // [16] {scale, shift}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (nsplit % 8 == 0: the workgroups of one XCD take consecutive splits, split s walks the tiles s, s + nsplit, ... -- the narrow
  //  operand's halo lines come out of the XCD's L2: wgrad_wino_kernel's order, profiles/r5_wgrad_item_order.md)
  const bool interleaved = (p.nsplit & 7) == 0;
  const int split = interleaved ? ((int)blockIdx.x & 7) * (p.nsplit >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int H = p.H, W = p.W, HW = H * W;
  const bool has_scale = p.a_scale != nullptr;
  if (tid < 16) tab[tid] = has_scale ? make_float2(p.a_scale[tid], p.a_shift[tid]) : make_float2(1.f, 0.f);

  // fixed staging positions
  const int gd = tid / C::PD, pd = tid - gd * C::PD;
  const int dty = (pd * 4) / C::TW, dtx = pd * 4 - dty * C::TW;
  const uint32_t taoff = (uint32_t)(gd * HW + dty * W + dtx);
  const int aloff = gd * C::PLA + pd * 4;
  int bch[C::NB], brow[C::NB], bcol[C::NB], bl[C::NB];
  bool bown[C::NB];
#pragma unroll
  for (int i = 0; i < C::NB; ++i) {
    const int e = tid + i * kThreads;
    bown[i] = e < C::NBF4;
    const int ee = bown[i] ? e : 0;
    bch[i] = ee / C::PB;
    const int pos = ee - bch[i] * C::PB;
    brow[i] = pos / C::ROWP4, bcol[i] = (pos - brow[i] * C::ROWP4) * 4;
    bl[i] = bch[i] * C::PLB + brow[i] * C::ROWP + bcol[i];
  }

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
  float4 pra[C::ND], prb[C::NB];
  bool prok[C::NB];
  auto issue = [&]() __attribute__((always_inline)) {
    const int n = nx_n, y0 = nx_ty * C::TH, x0 = nx_tx * C::TW;
    nx_tx += st_tx;
    const int cx = nx_tx >= p.tiles_x ? 1 : 0;
    nx_tx -= cx ? p.tiles_x : 0;
    nx_ty += st_ty + cx;
    const int cy = nx_ty >= p.tiles_y ? 1 : 0;
    nx_ty -= cy ? p.tiles_y : 0;
    nx_n += st_n + cy;
    const float* ab = p.a + n * p.a_bs + y0 * W + x0;
#pragma unroll
    for (int i = 0; i < C::ND; ++i) pra[i] = *reinterpret_cast<const float4*>(ab + (int64_t)i * C::GD * HW + taoff);
    const float* bb = p.b + n * p.b_bs;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
      const int gy = y0 + brow[i] - 1, gx = x0 + bcol[i] - 4;
      prok[i] = bown[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const uint32_t off = prok[i] ? (uint32_t)(bch[i] * HW + gy * W + gx) : 0u;
      prb[i] = *reinterpret_cast<const float4*>(bb + off);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::ND; ++i) {
      wsl_v2f lo = {pra[i].x, pra[i].y}, hi = {pra[i].z, pra[i].w};
      if (has_scale) {
        const float2 t = tab[gd + i * C::GD];
        xform_bn_leaky(lo, hi, t.x, t.y);
      }
      float* dst = a_t + i * (C::GD * C::PLA) + aloff;   // plane stride == 2 (mod 32): 8-byte aligned
      *reinterpret_cast<float2*>(dst) = make_float2(lo[0], lo[1]);
      *reinterpret_cast<float2*>(dst + 2) = make_float2(hi[0], hi[1]);
    }
#pragma unroll
    for (int i = 0; i < C::NB; ++i)
      if (bown[i])
        *reinterpret_cast<float4*>(b_t + bl[i]) = prok[i] ? prb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  // operand addresses of this lane: A row m = lane & 15, pixel k = lane >> 4; B column jj = t * 16 + (lane & 15)
  const float* ap = a_t + (lane & 15) * C::PLA + (lane >> 4);
  int bbase[C::NT];
#pragma unroll
  for (int t = 0; t < C::NT; ++t) {
    const int jj = t * 16 + (lane & 15);
    const int tap = jj < C::J ? jj / CBN : 4, cb = jj < C::J ? jj % CBN : 0;   // columns past J compute unused garbage
    const int ky = tap / 3, kx = tap - ky * 3;
    bbase[t] = cb * C::PLB + (1 + SGN * (ky - 1)) * C::ROWP + 4 + SGN * (kx - 1) + (lane >> 4);
  }
  v4f acc[C::NT];
  v4f accb = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < C::NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};

  if (it_count > 0) issue();
  __syncthreads();   // table visible
  for (int item = 0; item < it_count; ++item) {
    commit();
    __syncthreads();
    if (item + 1 < it_count) issue();   // next tile in flight during the MFMA loop
    constexpr int RW = C::TH / 4, NX = C::TW / 4;
#pragma unroll 4
    for (int st = 0; st < RW * NX; ++st) {
      const int r = wave * RW + st / NX, x4 = st % NX;
      const float av = ap[r * C::TW + x4 * 4];
      float bv[C::NT];
#pragma unroll
      for (int t = 0; t < C::NT; ++t) bv[t] = b_t[bbase[t] + r * C::ROWP + x4 * 4];
#pragma unroll
      for (int t = 0; t < C::NT; ++t) acc[t] = WSL_MFMA16(av, bv[t], acc[t]);
      accb = SGN > 0 ? WSL_MFMA16(av, 1.0f, accb) : WSL_MFMA16(1.0f, bv[C::TC], accb);
    }
    __syncthreads();
  }

  // ---- merge the four row groups in fixed order, store this split's partials
  float* red = reinterpret_cast<float*>(smem);
  constexpr int PER = (C::NT + 1) * 4;
  float* mine = red + (wave * 64 + lane) * PER;
#pragma unroll
  for (int t = 0; t < C::NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[t * 4 + r] = acc[t][r];
#pragma unroll
  for (int r = 0; r < 4; ++r) mine[C::NT * 4 + r] = accb[r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t <= C::NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
        for (int k = 0; k < 4; ++k) sum += red[(k * 64 + lane) * PER + t * 4 + r];
        if (t < C::NT) acc[t][r] = sum; else accb[r] = sum;
      }
    constexpr int Co = SGN > 0 ? 16 : CBN, Ci = SGN > 0 ? CBN : 16;
#pragma unroll
    for (int t = 0; t < C::NT; ++t) {
      const int jj = t * 16 + (lane & 15);
      if (jj < C::J) {
        const int tap = jj / CBN, cb = jj % CBN;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (lane >> 4) * 4 + r;
          const int co = SGN > 0 ? m : cb, ci = SGN > 0 ? cb : m;
          p.part_dw[(((int64_t)split * 9 + tap) * Co + co) * Ci + ci] = acc[t][r];
        }
      }
    }
    if (SGN > 0) {          // db[co = m]: row sums of A, identical in every column
      if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.part_db[(int64_t)split * 16 + (lane >> 4) * 4 + r] = accb[r];
      }
    } else {                // db[co = cb]: column sums of the centre-tap columns, identical in every row
      const int jj = C::TC * 16 + lane;
      if (lane < 16 && jj >= 4 * CBN && jj < 5 * CBN) p.part_db[(int64_t)split * CBN + (jj - 4 * CBN)] = accb[0];
    }
  }
}

template <int CBN, int SGN>
static int launch_wgrad_small(WgSmallP& p, void* stream) {
  using C = WgSmallCfg<CBN>;
  auto kern = wgrad_small_kernel<CBN, SGN>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(PF_WGRAD_DIRECT, 2.0 * px * 16 * CBN * 9, 4.0 * px * (16 + CBN), stream);
  WSL_LAUNCH(kern, dim3(p.nsplit), dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_small_kernel");
}

// Shapes this file takes (3x3, full 8x64 tiles, aligned planes); the caller falls back to the general kernels otherwise.
//   kind 1: Ci = 1,  Co = 16, raw input           kind 2: Ci = 16, Co = 4, input with at most the BN+LeakyReLU transform
int wgrad_small_kind(const WslSrc& a, const WslSrc* b, int H, int W, int Co, int ks) {
  if (ks != 3 || (H % 8) || (W % 64) || (b && b->C > 0) || a.emask || a.cmask) return 0;
  if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.bs & 3)) return 0;
  if (a.C == 1 && Co == 16 && !a.scale) return 1;
  if (a.C == 16 && Co == 4) return 2;
  return 0;
}

int wgrad_small_launch(int kind, const WslSrc& a, const float* dy, int64_t dy_bs, float* part_dw, float* part_db, int N,
                       int H, int W, int nsplit, void* stream) {
  WgSmallP p;
  p.part_dw = part_dw, p.part_db = part_db;
  p.N = N, p.H = H, p.W = W, p.tiles_x = W / 64, p.tiles_y = H / 8;
  p.items = N * p.tiles_x * p.tiles_y, p.nsplit = nsplit;
  if (kind == 1) {
    p.a = dy, p.a_bs = dy_bs, p.a_scale = nullptr, p.a_shift = nullptr;
    p.b = a.x, p.b_bs = a.bs;
    return launch_wgrad_small<1, 1>(p, stream);
  }
  p.a = a.x, p.a_bs = a.bs, p.a_scale = a.scale, p.a_shift = a.shift;
  p.b = dy, p.b_bs = dy_bs;
  return launch_wgrad_small<4, -1>(p, stream);
}


// ================================================================================================ classifier forward
// out_conv of a decoder: 3x3, 16 -> 4 channels at full resolution.  On the 16x16x4 tiling three quarters of every MFMA are
// padding (conv_mfma2_kernel: ~270 us per launch).  v_mfma_f32_4x4x1_16b_f32 fits exactly: 16 independent 4x4 outer
// products per instruction = 64 consecutive pixels of a row (4 per block) x the 4 classes, K = 1 (one (ci, tap) pair).
// Lane l supplies the input of pixel l (A) and the weight of class l & 3 (B) and receives pixels 4*(l>>2)..+3 of class
// l & 3 -- a float4 store.  The 144 weights of a lane's class live in registers for the whole (persistent) workgroup.
struct ClsP {
  const float* x;        // [N,16,H,W] raw conv output of the last decoder block
  int64_t x_bs;
  const float* scale;    // BN+LeakyReLU loader transform (or null)
  const float* shift;
  const float* wp;       // packed [9][16][4]
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, H, W, tiles_x, tiles_y, items;
};

struct ClsCfg {
  static constexpr int TH = 8, TW = 64, ROWP = TW + 8, ROWS = TH + 2, ROWP4 = ROWP / 4, POS = ROWS * ROWP4, KC = 8;
  static constexpr int PLANE = ROWS * ROWP;
  // TWO staged halves (round 6): the 8-channel half being multiplied and the one being committed live in different buffers, so a tile costs
  // two barriers instead of four (46 KB per workgroup, two workgroups per CU)
  static constexpr size_t SMEM = sizeof(float) * (2 * KC * PLANE + 32);
  static_assert(POS <= 256, "one float4 position per thread");
};

__global__ __launch_bounds__(256, 2) void conv_cls_kernel(ClsP p) {
  using C = ClsCfg;
  WSL_DYN_SMEM(smem);
  float* in_b = reinterpret_cast<float*>(smem);                      // two staged halves [2][KC][PLANE]
  float2* tab = reinterpret_cast<float2*>(in_b + 2 * C::KC * C::PLANE);   // [16] {scale, shift}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, W = p.W, HW = H * W;
  const bool has_scale = p.scale != nullptr;
  if (tid < 16) tab[tid] = has_scale ? make_float2(p.scale[tid], p.shift[tid]) : make_float2(1.f, 0.f);

  // weights of this lane's class, all (tap, ci): registers (every index below is a compile-time constant)
  float wreg[9 * 16];
#pragma unroll
  for (int k = 0; k < 9 * 16; ++k) wreg[k] = p.wp[k * 4 + (lane & 3)];
  const float bias = p.bias ? p.bias[lane & 3] : 0.f;

  // staging position of this thread (fixed) -- threads past the halo tile idle while staging
  const int pty = tid / C::ROWP4, ptx4 = tid - pty * C::ROWP4;
  const bool owner = tid < C::POS;
  const int loff = pty * C::ROWP + ptx4 * 4;
  // tiles of this (persistent) workgroup: it0, it0 + step, ... < it1.  gridDim.x % 8 == 0: the workgroups of one XCD take neighbouring
  // tiles in every round, so the halo lines two tiles share come out of that XCD's L2 (one contiguous run per workgroup read the input
  // twice: 535 MB fetched for a 268 MB tensor, L2 hit share 0.19; profiles/r5_wgrad_item_order.md)
  const int nwg = (int)gridDim.x, bx = (int)blockIdx.x;
  const bool interleaved = (nwg & 7) == 0;
  const int it0 = interleaved ? (bx & 7) * (nwg >> 3) + (bx >> 3) : (int)((int64_t)bx * p.items / nwg);
  const int it1 = interleaved ? p.items : (int)((int64_t)(bx + 1) * p.items / nwg);
  const int step = interleaved ? nwg : 1;

  float4 pre[C::KC];   // (BOTH halves of the next tile in flight during a whole tile's matrix phase -- 64 registers, 242 in all -- measured
                       //  113.4 us against 104.5 for this form and 107.7 for the four-barrier single buffer: round 6, tools/gpu_kernel_ab.sh)
  bool pok = false;
  auto tile_of = [&](int item, int& n, int& y0, int& x0) {
    int q = item;
    const int tx = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty = q % p.tiles_y;
    n = q / p.tiles_y, y0 = ty * C::TH, x0 = tx * C::TW;
  };
  auto issue = [&](int item, int ch) __attribute__((always_inline)) {
    int n, y0, x0;
    tile_of(item, n, y0, x0);
    const int gy = y0 + pty - 1, gx = x0 + ptx4 * 4 - 4;
    pok = owner && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const uint32_t off = pok ? (uint32_t)(gy * W + gx) : 0u;
    const float* xb = p.x + n * p.x_bs + (int64_t)ch * C::KC * HW;
#pragma unroll
    for (int i = 0; i < C::KC; ++i) pre[i] = *reinterpret_cast<const float4*>(xb + (int64_t)i * HW + off);
  };
  auto commit = [&](int ch) __attribute__((always_inline)) {
    float* in_t = in_b + ch * (C::KC * C::PLANE);
    if (owner) {
#pragma unroll
      for (int i = 0; i < C::KC; ++i) {
        wsl_v2f lo = {pre[i].x, pre[i].y}, hi = {pre[i].z, pre[i].w};
        if (has_scale) {
          const float2 t = tab[ch * C::KC + i];
          xform_bn_leaky(lo, hi, t.x, t.y);
        }
        if (!pok) lo = wsl_v2f{0.f, 0.f}, hi = wsl_v2f{0.f, 0.f};
        *reinterpret_cast<float4*>(in_t + i * C::PLANE + loff) = make_float4(lo[0], lo[1], hi[0], hi[1]);
      }
    }
  };

  v4f acc[2];
  // one 8-channel half of the reduction: rows 2*wave and 2*wave+1 of the tile, pixel = lane
  auto mfma_half = [&](auto ch_tag) __attribute__((always_inline)) {
    constexpr int CH = decltype(ch_tag)::value;   // std::integral_constant
    const float* in_t = in_b + CH * (C::KC * C::PLANE);
#pragma unroll
    for (int cc = 0; cc < C::KC; ++cc)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float av[2][3];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) av[rr][kx] = in_t[cc * C::PLANE + (wave * 2 + rr + ky) * C::ROWP + lane + kx + 3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) acc[rr] = WSL_MFMA4(av[rr][kx], wreg[(ky * 3 + kx) * 16 + CH * C::KC + cc], acc[rr]);
      }
  };

  if (it0 < it1) issue(it0, 0);
  __syncthreads();   // table visible
  for (int item = it0; item < it1; item += step) {
    acc[0] = v4f{bias, bias, bias, bias}, acc[1] = acc[0];
    commit(0);                     // (buffer 0 is free: every wave passed barrier B of the previous tile after its first half)
    issue(item, 1);
    __syncthreads();               // A: half 0 staged; every wave is done with buffer 1 (second half of the previous tile)
    mfma_half(std::integral_constant<int, 0>{});
    commit(1);
    if (item + step < it1) issue(item + step, 0);
    __syncthreads();               // B: half 1 staged; every wave is done with buffer 0
    mfma_half(std::integral_constant<int, 1>{});
    int n, y0, x0;
    tile_of(item, n, y0, x0);
    float* yb = p.y + n * p.y_bs + (int64_t)(lane & 3) * HW + (int64_t)(y0 + wave * 2) * W + x0 + (lane >> 2) * 4;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
      *reinterpret_cast<float4*>(yb + rr * W) = make_float4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
  }
}

// 3x3, 16 -> 4, one source with at most the BN+LeakyReLU transform, no statistics, full 8x64 tiles, aligned planes
bool conv_cls_eligible(const WslSrc& a, const WslSrc* b, const float* y, int64_t y_bs, int H, int W, int Co, int ks,
                       const float* stat_part) {
  if (ks != 3 || Co != 4 || a.C != 16 || (b && b->C > 0) || a.emask || a.cmask || stat_part) return false;
  if ((H % 8) || (W % 64) || (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.bs & 3)) return false;
  return (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_bs & 3) == 0;
}

int conv_cls_launch(const WslSrc& a, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H, int W,
                    void* stream) {
  ClsP p;
  p.x = a.x, p.x_bs = a.bs, p.scale = a.scale, p.shift = a.shift, p.wp = wp, p.bias = bias, p.y = y, p.y_bs = y_bs;
  p.N = N, p.H = H, p.W = W, p.tiles_x = W / 64, p.tiles_y = H / 8, p.items = N * p.tiles_x * p.tiles_y;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(conv_cls_kernel, ClsCfg::SMEM);
    attr_done = true;
  }
  int wgs = forced_wgrad_wgs() > 0 ? forced_wgrad_wgs() : 2 * device_cu_count();   // (wsl_debug_wgrad_workgroups(): tests force a few)
  if (wgs > p.items) wgs = p.items;
  const double px = (double)N * H * W;
  void* tok = prof_begin(0, 2.0 * px * 4 * 16 * 9, 4.0 * px * (16 + 4), stream);
  WSL_LAUNCH(conv_cls_kernel, dim3(wgs), dim3(kThreads), ClsCfg::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_cls_kernel");
}

// ================================================================================================ narrow-K convolution
// The first convolution (1 -> 16, forward) and the 4-class classifiers' data gradient (4 -> 16): 3x3, 16 output channels, so few
// reduction elements (9 resp. 36) that the generic kernel's 8-channel chunks, bounds tests and staging dominate -- these launches
// are HBM-side (one 16-channel full-resolution tensor written).  Here the whole reduction is ONE pass of v_mfma_f32_16x16x4_f32
// operands: K = the 4 input channels of one tap (9 MFMAs per 16 pixels x 16 channels), or -- single input channel -- K = 4 taps
// (3 MFMAs, the last group padded with zero weights).  A operand = one 4-byte LDS read of the staged halo tile per MFMA, B
// operands = 9 / 3 registers loaded once from the packed [tap][ci][co] image the generic kernel uses.  Persistent workgroups
// over 8 x 64-pixel tiles (the same tiles and statistics slots as the generic plan), next tile's rows prefetched into registers.
// Epilogue as the other conv kernels: bias + BatchNorm partials (forward), or the BatchNorm-backward statistics (data gradient).
struct NkP {
  const float* x;
  int64_t x_bs;
  const float* wp;   // packed [9][CI][16]
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, H, W, tiles_x, tiles_y, items;
  float* stat_part;
  float* stat_cnt;
  int slots;
  BnBwdEpi bn;
};

template <int CI, int TH_ = 8, int TW_ = 64>
struct NkCfg {
  static constexpr int TH = TH_, TW = TW_, SEGW = TW / 16;   // (TH x TW = 512 pixels: 32 segments of 16, eight per wave, row-major)
  static constexpr int PADL = 4, ROWP = TW + 2 * PADL, ROWS = TH + 2, ROWP4 = ROWP / 4, POS = ROWS * ROWP4;
  static constexpr int PLANE = ((ROWS * ROWP - 16 + 31) / 32) * 32 + 16;   // == 16 (mod 32): the four channel planes of a read hit distinct banks
  static constexpr int NG = CI == 4 ? 9 : 3;                               // MFMAs per 16 pixels
  static constexpr size_t SMEM = sizeof(float) * (CI * PLANE + 8 * 16);
  static_assert(POS <= 256 && (CI == 1 || CI == 4) && TH * TW == 512, "one float4 position per thread");
};

// (NHWC: layout probe of the experiments build -- the 16 output channels of a pixel stored contiguously, [N][H][W][16]; the
//  result is not what any consumer reads, only the store pattern is of interest: tools/microbench_nk16.py)
constexpr int kNk16MinW = 2;   // waves per SIMD the register allocator leaves room for.  Round 3 had 3: a 168-register cap that put 56 (CI = 4) /
                               // 12 (CI = 1) values into scratch; measured in round 4 (profiles/r4_minw_ab.log): classifier data gradient
                               // 134 -> 97 us with 2, first convolution 79 -> 78 us
template <int CI, int TH = 8, int TW = 64, bool NHWC = false>
__global__ __launch_bounds__(256, kNk16MinW) void conv_nk16_kernel(NkP p) {
  using C = NkCfg<CI, TH, TW>;
  WSL_DYN_SMEM(smem);
  float* in_t = reinterpret_cast<float*>(smem);
  float* red = in_t + CI * C::PLANE;   // 128 floats of reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, W = p.W, HW = H * W;
  const int c16 = lane & 15, k4 = lane >> 4;

  // B operands: lane (j = c16, k = k4).  CI == 4: group g = tap, k = input channel.  CI == 1: tap = 4 g + k (zero past 8)
  float bw[C::NG];
  int aoff[C::NG];   // LDS offset of this lane's A element relative to (row, pixel) of the output
#pragma unroll
  for (int g = 0; g < C::NG; ++g) {
    if (CI == 4) {
      bw[g] = p.wp[(g * 4 + k4) * 16 + c16];
      aoff[g] = k4 * C::PLANE + (g / 3) * C::ROWP + (g % 3);
    } else {
      const int tap = 4 * g + k4;
      bw[g] = tap < 9 ? p.wp[tap * 16 + c16] : 0.f;
      aoff[g] = tap < 9 ? (tap / 3) * C::ROWP + (tap % 3) : 0;
    }
  }
  const float bias = p.bias ? p.bias[c16] : 0.f;

  const int pty = tid / C::ROWP4, ptx4 = tid - pty * C::ROWP4;
  const bool owner = tid < C::POS;
  const int loff = pty * C::ROWP + ptx4 * 4;
  const int it0 = (int)((int64_t)blockIdx.x * p.items / gridDim.x), it1 = (int)((int64_t)(blockIdx.x + 1) * p.items / gridDim.x);
  float4 pre[CI];
  bool pok = false;
  auto tile_of = [&](int item, int& n, int& y0, int& x0) {
    int q = item;
    const int tx = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty = q % p.tiles_y;
    n = q / p.tiles_y, y0 = ty * C::TH, x0 = tx * C::TW;
  };
  auto issue = [&](int item) __attribute__((always_inline)) {
    int n, y0, x0;
    tile_of(item, n, y0, x0);
    const int gy = y0 + pty - 1, gx = x0 + ptx4 * 4 - C::PADL;
    pok = owner && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const uint32_t off = pok ? (uint32_t)(gy * W + gx) : 0u;
    const float* xb = p.x + n * p.x_bs;
#pragma unroll
    for (int i = 0; i < CI; ++i) pre[i] = *reinterpret_cast<const float4*>(xb + (int64_t)i * HW + off);
  };
  auto commit = [&]() __attribute__((always_inline)) {
    if (owner) {
#pragma unroll
      for (int i = 0; i < CI; ++i)
        *reinterpret_cast<float4*>(in_t + i * C::PLANE + loff) = pok ? pre[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  if (it0 < it1) issue(it0);
  for (int item = it0; item < it1; ++item) {
    commit();
    __syncthreads();
    if (item + 1 < it1) issue(item + 1);
    int n, y0, x0;
    tile_of(item, n, y0, x0);
    // this wave: rows 2 * wave, 2 * wave + 1; 4 segments of 16 pixels each; lane (c16, k4) holds pixels 4 k4 .. 4 k4 + 3 of channel c16
    v4f acc[8];
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) {
      acc[sg] = v4f{bias, bias, bias, bias};
      const float* ab = in_t + ((8 * wave + sg) / C::SEGW) * C::ROWP + (C::PADL - 1) + 16 * ((8 * wave + sg) % C::SEGW) + c16;
#pragma unroll
      for (int g = 0; g < C::NG; ++g) acc[sg] = WSL_MFMA16(ab[aoff[g]], bw[g], acc[sg]);
    }
    // (staging the result through LDS so that every store instruction writes four 256-byte row segments instead of sixteen
    //  64-byte pieces was measured SLOWER: 92.5 vs 75.8 us forward, 132.9 vs 115.7 data gradient)
    if constexpr (NHWC) {
      float* yb = p.y + n * p.y_bs + ((int64_t)y0 * W + x0 + 4 * k4) * 16 + c16;
#pragma unroll
      for (int sg = 0; sg < 8; ++sg)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          yb[((int64_t)((8 * wave + sg) / C::SEGW) * W + 16 * ((8 * wave + sg) % C::SEGW) + r) * 16] = acc[sg][r];
    } else {
    float* yb = p.y + n * p.y_bs + (int64_t)c16 * HW + (int64_t)y0 * W + x0 + 4 * k4;
#pragma unroll
    for (int sg = 0; sg < 8; ++sg)
      *reinterpret_cast<float4*>(yb + ((8 * wave + sg) / C::SEGW) * W + 16 * ((8 * wave + sg) % C::SEGW)) =
          make_float4(acc[sg][0], acc[sg][1], acc[sg][2], acc[sg][3]);
    }

    if (p.bn.part) {   // data gradient: BatchNorm-backward statistics of the layer that consumes it
      const float mean = p.bn.st[c16], invstd = p.bn.st[16 + c16], sc = p.bn.st[32 + c16], sh = p.bn.st[48 + c16];
      const int64_t base = ((int64_t)n * 16 + c16) * HW + (int64_t)y0 * W + x0 + 4 * k4;
      BnBwdAcc ba;
#pragma unroll
      for (int sg = 0; sg < 8; ++sg)
        bn_bwd_acc4(p.bn, base + ((8 * wave + sg) / C::SEGW) * W + 16 * ((8 * wave + sg) % C::SEGW), acc[sg][0], acc[sg][1], acc[sg][2], acc[sg][3], mean, invstd, sc,
                    sh, ba);
      float s1[1], s2[1];
      bn_bwd_fold(ba, s1[0], s2[0]);
      bn_bwd_store<1, 16>(p.bn, s1, s2, red, 0, 16, item, p.items);
    } else if (p.stat_part) {   // forward: per-tile (sum, M2) per channel for the BatchNorm that follows
      constexpr float cnt = (float)(C::TH * C::TW);
      float* red1 = red;
      float* red2 = red + 64;
      float bs = 0.f;
#pragma unroll
      for (int sg = 0; sg < 8; ++sg) bs += (acc[sg][0] + acc[sg][1]) + (acc[sg][2] + acc[sg][3]);
      bs += __shfl_xor(bs, 16);
      bs += __shfl_xor(bs, 32);
      if (lane < 16) red1[wave * 16 + lane] = bs;
      __syncthreads();
      const float mean_b = (red1[c16] + red1[16 + c16] + red1[32 + c16] + red1[48 + c16]) / cnt;
      float q = 0.f;
#pragma unroll
      for (int sg = 0; sg < 8; ++sg)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = acc[sg][e] - mean_b;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (lane < 16) red2[wave * 16 + lane] = q;
      __syncthreads();
      if (wave == 0 && lane < 16) {
        float* dst = p.stat_part + ((int64_t)lane * ((int64_t)p.items * p.slots) + (int64_t)item * p.slots) * 2;   // [Co][slots][2]
        dst[0] = red1[lane] + red1[16 + lane] + red1[32 + lane] + red1[48 + lane];
        dst[1] = red2[lane] + red2[16 + lane] + red2[32 + lane] + red2[48 + lane];
        if (lane < p.slots) p.stat_cnt[item * p.slots + lane] = lane == 0 ? cnt : 0.f;
      }
    }
    __syncthreads();   // the tile and the scratch are free for the next item
  }
}

// 3x3, Ci in {1, 4} -> 16 channels, one plain source, the generic plan's 8 x 64 tiles, aligned planes
bool conv_nk16_eligible(const WslSrc& a, const WslSrc* b, const float* y, int64_t y_bs, int H, int W, int Co, int ks, int th, int tw) {
  if (ks != 3 || Co != 16 || (a.C != 1 && a.C != 4) || (b && b->C > 0) || a.scale || a.emask || a.cmask) return false;
  if (th != 8 || tw != 64 || (H % 8) || (W % 64) || (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.bs & 3)) return false;
  return (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_bs & 3) == 0 && (int64_t)16 * H * W < (int64_t(1) << 31);
}

template <int CI, int TH, int TW>
static int launch_nk16(NkP& p, bool dgrad, void* stream) {
  using C = NkCfg<CI, TH, TW>;
  auto kern = conv_nk16_kernel<CI, TH, TW>;
  p.tiles_x = p.W / TW, p.tiles_y = p.H / TH;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  int wgs = kNk16MinW * device_cu_count();
  if (wgs > p.items) wgs = p.items;
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(dgrad ? PF_CONV_DGRAD : PF_CONV_FWD, 2.0 * px * CI * 16 * 9, 4.0 * px * (16 + CI) + bn_epi_bytes(p.bn, px * 16), stream);
  WSL_LAUNCH(kern, dim3(wgs), dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_nk16_kernel");
}

int conv_nk16_launch(const WslSrc& a, const float* wp, const float* bias, float* y, int64_t y_bs, int N, int H, int W, bool dgrad,
                     float* stat_part, float* stat_cnt, int slots, const BnBwdEpi* bn, int* bn_done, void* stream) {
  NkP p;
  p.x = a.x, p.x_bs = a.bs, p.wp = wp, p.bias = bias, p.y = y, p.y_bs = y_bs;
  p.N = N, p.H = H, p.W = W, p.tiles_x = W / 64, p.tiles_y = H / 8, p.items = N * p.tiles_x * p.tiles_y;
  p.stat_part = stat_part, p.stat_cnt = stat_cnt, p.slots = slots;
  const bool bn_ok = bn && bn->part;
  if (bn_ok) p.bn = *bn;
  if (bn_done) *bn_done = bn_ok ? 1 : 0;
#ifdef WSL_EXPERIMENTS
  // same number of tiles (and statistics slots) either way; 4 x 128 writes 512-byte row segments instead of 256-byte ones:
  // 113 vs 127-136 us for the data gradient alone, no difference in the step (3862 vs 3862 slices/s) -- experiments build only
  static const int nhwc = WSL_TUNE("WSL_NK16_NHWC", 0);
  if (nhwc) {
    auto kern = a.C == 4 ? conv_nk16_kernel<4, 8, 64, true> : conv_nk16_kernel<1, 8, 64, true>;
    p.tiles_x = W / 64, p.tiles_y = H / 8;
    int wgs = kNk16MinW * device_cu_count();
    if (wgs > p.items) wgs = p.items;
    WSL_LAUNCH(kern, dim3(wgs), dim3(kThreads), (a.C == 4 ? NkCfg<4>::SMEM : NkCfg<1>::SMEM), stream, p);
    return check_launch("conv_nk16_kernel(nhwc probe)");
  }
  static const int wide = WSL_TUNE("WSL_NK16_WIDE", 0);
  if (wide && W % 128 == 0) return a.C == 4 ? launch_nk16<4, 4, 128>(p, dgrad, stream) : launch_nk16<1, 4, 128>(p, dgrad, stream);
#endif
  return a.C == 4 ? launch_nk16<4, 8, 64>(p, dgrad, stream) : launch_nk16<1, 8, 64>(p, dgrad, stream);
}

// ---- machine probes (EXPERIMENTS build only; tools/mfma_ceiling.py, tools/probe_lds_dma.py, tools/probe_mfma4.py)
#ifdef WSL_EXPERIMENTS
// Pure MFMA stream (no memory): the practical f32 matrix ceiling of the machine at its sustained clock, per MFMA shape.
// shape 0: 16x16x4 (16 independent accumulators), 1: 32x32x2 (4 accumulators), 2: 4x4x1 (16 accumulators).
#ifndef WSL_HOST_EMUL
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_stream_kernel(float* out, int iters) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef float v16 __attribute__((ext_vector_type(16)));
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  float r = 0.f;
  if (SHAPE == 1) {
    v16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) r += acc[i][0];
  } else {
    v4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = SHAPE == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) r += acc[i][0];
  }
  if (r == 123.456f) out[0] = r;
}

// Does ordinary f32 VALU work overlap with the f32 MFMA stream?  MODE 0: every wave issues K independent v_fma_f32 after
// each MFMA (same wave).  MODE 1: 8-wave workgroups, waves 0-3 run the pure MFMA stream, waves 4-7 (the second wave of
// each SIMD) run K v_fma_f32 per MFMA-time slot -- tools/mfma_ceiling.py prints the MFMA rate each way.
template <int MODE, int K>
__global__ __launch_bounds__(512) void mfma_valu_mix_kernel(float* out, int iters) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  v4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = v4{0.f, 0.f, 0.f, 0.f};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  const bool mfma_wave = MODE == 0 || (threadIdx.x >> 6) < 4;
  const bool valu_wave = MODE == 0 || (threadIdx.x >> 6) >= 4;
  if (mfma_wave && valu_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) x[k & 7] = __builtin_fmaf(x[k & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
      }
    }
  } else if (mfma_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16 * K; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], 1.0001f, 0.5f);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[i][0];
  for (int i = 0; i < 8; ++i) r += x[i];
  if (r == 123.456f) out[0] = r;
}

// The same question for PACKED f32 adds (v_pk_add_f32 with op_sel, as the Winograd transforms use them): K per MFMA in the
// same wave.  Does one packed instruction cost one vector-instruction slot or two?
template <int K>
__global__ __launch_bounds__(256) void mfma_pk_mix_kernel(float* out, int iters) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  v4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = v4{0.f, 0.f, 0.f, 0.f};
  wsl_v2f x[8];
  for (int i = 0; i < 8; ++i) x[i] = wsl_v2f{a + i, a - i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k)
        asm volatile("v_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "+v"(x[k & 7]));
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[i][0];
  for (int i = 0; i < 8; ++i) r += x[i][0] + x[i][1];
  if (r == 123.456f) out[0] = r;
}
#endif

// Destination-layout probe of global_load_lds_dwordx4 (LDS DMA): every lane fetches its own 16 bytes; where do they land?
#ifndef WSL_HOST_EMUL
__global__ void lds_dma_probe_kernel(const float* g, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = -1.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) != 5)    // one lane masked off: its slot must stay untouched
    __builtin_amdgcn_global_load_lds((gptr_t)(g + threadIdx.x * 4), (lptr_t)(lds + wave * 256 + 8), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = lds[i];
}
#endif

// The packed Winograd transforms of wsl_rt.h on given data: in[t] = a 4 x 4 patch + a 2 x 2 tile, out[t] = V[16], Z[16]
// in transform-position order (tools/probe_pk.py compares with the definitions)
#ifndef WSL_HOST_EMUL
__global__ void pk_probe_kernel(const float* in, float* out) {
  const int t = threadIdx.x;
  const float* d = in + 20 * t;
  wsl_v2f lo[4], hi[4], a[4], b[4], q1, q2, m[4];
  for (int i = 0; i < 4; ++i) lo[i] = wsl_v2f{d[4 * i], d[4 * i + 1]}, hi[i] = wsl_v2f{d[4 * i + 2], d[4 * i + 3]};
  const wsl_v2f r0 = {d[16], d[17]}, r1 = {d[18], d[19]};
  wino_btdb_pk(lo, hi, a, b);
  wino_aya_pk(r0, r1, q1, q2, m);
  const wsl_v2f q[4] = {r0, q1, q2, r1};
  float* o = out + 32 * t;
  for (int xi = 0; xi < 16; ++xi) o[xi] = wino_pick(a, b, xi), o[16 + xi] = wino_pick(q, m, xi);
}
#endif

// Operand-layout probe of the 4x4x1 (16 blocks) f32 MFMA used by the classifier forward: d[lane][r] for given a, b.
#ifndef WSL_HOST_EMUL
__global__ void mfma4_probe_kernel(const float* a, const float* b, float* d) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = acc[r];
}
#endif

#endif  // WSL_EXPERIMENTS

}  // namespace wsl

#ifdef WSL_EXPERIMENTS
extern "C" int wsl_debug_lds_dma_probe(const float* g, float* out, void* stream) {
#ifndef WSL_HOST_EMUL
  WSL_LAUNCH(wsl::lds_dma_probe_kernel, dim3(1), dim3(256), 0, stream, g, out);
  return wsl::check_launch("lds_dma_probe_kernel");
#else
  (void)g, (void)out, (void)stream;
  return WSL_EUNSUPPORTED;
#endif
}

extern "C" int wsl_debug_mfma_stream(int shape, int blocks, int iters, float* out, void* stream) {
#ifndef WSL_HOST_EMUL
  if (shape == 0) WSL_LAUNCH(wsl::mfma_stream_kernel<0>, dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 1) WSL_LAUNCH(wsl::mfma_stream_kernel<1>, dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 2) WSL_LAUNCH(wsl::mfma_stream_kernel<2>, dim3(blocks), dim3(256), 0, stream, out, iters);
  // 100 + K: K v_fma_f32 per MFMA in the same wave; 200 + K: in the partner wave of the SIMD (8-wave workgroups)
  else if (shape == 102) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<0, 2>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 104) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<0, 4>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 108) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<0, 8>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 116) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<0, 16>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 202) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<1, 2>), dim3(blocks), dim3(512), 0, stream, out, iters);
  else if (shape == 204) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<1, 4>), dim3(blocks), dim3(512), 0, stream, out, iters);
  else if (shape == 208) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<1, 8>), dim3(blocks), dim3(512), 0, stream, out, iters);
  else if (shape == 216) WSL_LAUNCH((wsl::mfma_valu_mix_kernel<1, 16>), dim3(blocks), dim3(512), 0, stream, out, iters);
  // 300 + K: K v_pk_add_f32 (op_sel form) per MFMA in the same wave
  else if (shape == 301) WSL_LAUNCH((wsl::mfma_pk_mix_kernel<1>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 302) WSL_LAUNCH((wsl::mfma_pk_mix_kernel<2>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 304) WSL_LAUNCH((wsl::mfma_pk_mix_kernel<4>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else if (shape == 308) WSL_LAUNCH((wsl::mfma_pk_mix_kernel<8>), dim3(blocks), dim3(256), 0, stream, out, iters);
  else {
    wsl::set_error("debug_mfma_stream: shape %d", shape);
    return WSL_EINVAL;
  }
  return wsl::check_launch("mfma_stream_kernel");
#else
  (void)shape, (void)blocks, (void)iters, (void)out, (void)stream;
  return WSL_EUNSUPPORTED;
#endif
}

extern "C" int wsl_debug_pk_probe(const float* in, float* out, void* stream) {
#ifndef WSL_HOST_EMUL
  WSL_LAUNCH(wsl::pk_probe_kernel, dim3(1), dim3(64), 0, stream, in, out);
  return wsl::check_launch("pk_probe_kernel");
#else
  (void)in, (void)out, (void)stream;
  return WSL_EUNSUPPORTED;
#endif
}

extern "C" int wsl_debug_mfma4_probe(const float* a, const float* b, float* d, void* stream) {
#ifndef WSL_HOST_EMUL
  WSL_LAUNCH(wsl::mfma4_probe_kernel, dim3(1), dim3(64), 0, stream, a, b, d);
  return wsl::check_launch("mfma4_probe_kernel");
#else
  (void)a, (void)b, (void)d, (void)stream;
  return WSL_EUNSUPPORTED;
#endif
}
#endif  // WSL_EXPERIMENTS
