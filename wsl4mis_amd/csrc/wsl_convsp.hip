// Split-precision convolution kernels (opt-in): the 3x3 convolutions of the UNet -- forward,
// data gradient and weight gradient -- as DIRECT implicit GEMMs on v_mfma_f32_16x16x32_f16 (16x the f32 MFMA rate), with
// fp32-class accuracy from a two-term operand split done while a tile is staged (wsl_rt.h: sp_split2):
//     x * 2^e = hi + lo        acc += hi_a * lo_b;  acc += lo_a * hi_b;  acc += hi_a * hi_b      (fp32 accumulator)
// The dropped lo * lo term is 2^-22 of the product.  Reference semantics are those of the f32 kernels
// (ref: networks/unet.py:13-29 ConvBlock; autograd of the same).
//
// One LDS image serves all three kernels: CHANNEL-INNERMOST octets -- 8 channels x f16 = one 16-byte slot per (octet, pixel) --
// because the f16 MFMA wants 8 consecutive K values per lane:
//   * conv (K = input channels of a tap): lane (pixel m = l & 15, k-group g = l >> 4) reads ONE slot (ds_read_b128) per operand;
//     a 16-channel chunk gives K = 32 as two taps x 16 channels (g >> 1 = which tap, g & 1 = which octet): 5 K-steps per chunk,
//     the tenth tap is a zero block of the weight image;
//   * weight gradient (K = pixels): the same slots read with ds_read_b64_tr_b16, which hands lane (channel, 4 pixels) the
//     transposed 4 x 16 block -- a tap shift is a whole-slot offset, so no operand is ever misaligned.
// Inside a plane the slots of a row are stored pixel-of-quad major (slot(c) = (c & 3) * P + (c >> 2), P = 4 or 12 mod 16): a
// staging thread (4 pixels x 8 channels from eight float4 loads) then writes four slots that are each contiguous across lanes
// (conflict-free ds_write_b128), and the 16 pixels of an MFMA row tile still fall into 16 different 16-byte bank groups.
#include "wsl_rt.h"

#include <type_traits>
#include <utility>

namespace wsl {

struct SpSrc {           // one source, device view
  const float* x;
  const uint8_t* emask;
  const float* scale;
  const float* shift;
  const float* cmask;
  int64_t bs;
  int C;
  float es;
};

static SpSrc to_spsrc(const WslSrc& s) { return SpSrc{s.x, s.emask, s.scale, s.shift, s.cmask, s.bs, s.C, s.emask_scale}; }

constexpr int kSpMaxC = 512;   // channels whose loader coefficients fit the LDS table

// ------------------------------------------------------------------------------------------------ tile image + staging
// A staged tile: ROWS x (4 * NQ) pixels x NOCT octets, hi and lo images.  Byte offset of (hl, octet o, row r, staged column c):
//     hl * HL + o * PLANE + r * ROWB + ((c & 3) * P + (c >> 2)) * 16
// P_ = 0: the pitch the conv kernels' row reads need (>= NQ and = 4 or 12 mod 16); the weight gradient's transpose reads take
// compact images (P_ = NQ).
template <int ROWS_, int NQ_, int NOCT_, int P_ = 0, int PAD_ = 0>
struct SpImg {
  static constexpr int ROWS = ROWS_, NQ = NQ_, NOCT = NOCT_;
  static constexpr int P = P_ ? P_ : (NQ <= 12 ? 12 : (NQ <= 20 ? 20 : (NQ <= 28 ? 28 : 36)));
  static constexpr int RS = 4 * P + 2;                              // slots per row (+2: consecutive rows rotate by two slots)
  static constexpr int ROWB = RS * 16;
  static constexpr int PLANE = ((ROWS * ROWB + 255) / 256) * 256 + PAD_;   // PAD_ = 0: = 0 (mod 256), the octets of one read hit the same bank groups
  static constexpr int HL = NOCT * PLANE;
  static constexpr int BYTES = 2 * HL;
  static constexpr int NTASK = ROWS * NQ * NOCT, NR = (NTASK + 255) / 256;
  static_assert(NQ <= 36, "tile too wide");
};

template <int NR>
struct SpTasks {            // this thread's staging tasks: (row, quad, octet) = NR rounds of 256
  uint32_t toff[NR];        // element offset of the quad inside a channel plane (0 when outside the image)
  int loff[NR];             // byte offset of pixel 0 of the quad inside the hi image
  int oct[NR];
  bool valid[NR], active[NR];
};

template <int NR>
struct SpRegs {             // prefetched raw data of the tasks: 8 channels x 4 pixels (+ keep bytes), and whether the quad was inside
  float4 v[NR][8];          // the image of the tile it was requested for (the task table may already describe the next tile)
  uint32_t m[NR][8];
  bool valid[NR];
};

// (y0, x0) = image coordinates of staged row 0 / staged column 0
template <typename I>
__device__ __forceinline__ void sp_tasks_init(SpTasks<I::NR>& t, int tid, int y0, int x0, int H, int W) {
#pragma unroll
  for (int r = 0; r < I::NR; ++r) {
    const int k = tid + r * kThreads;
    const int q = k % I::NQ, rest = k / I::NQ;
    const int row = rest % I::ROWS, o = rest / I::ROWS;
    const int gy = y0 + row, gx = x0 + 4 * q;
    t.active[r] = k < I::NTASK;
    t.valid[r] = t.active[r] && gy >= 0 && gy < H && gx >= 0 && gx < W;     // W % 4 == 0: a quad is inside or outside as a whole
    t.toff[r] = t.valid[r] ? (uint32_t)(gy * W + gx) : 0u;
    t.oct[r] = t.active[r] ? o : 0;
    t.loff[r] = t.oct[r] * I::PLANE + row * I::ROWB + q * 16;
  }
}

// loads of one channel block (8 * NOCT channels starting at channel `chb` of sample-local source planes xs / ms)
template <typename I>
__device__ __forceinline__ void sp_issue(const SpTasks<I::NR>& t, SpRegs<I::NR>& g, const float* xs, const uint8_t* ms, int chb,
                                         int HW) {
  // address = uniform base of the channel (scalar registers) + the thread's 32-bit element offset: one vector register per task instead of
  // a 64-bit vector add per load (the kernels are short of vector issue slots)
  const float* xc = xs + (int64_t)chb * HW;
  const uint8_t* mc = ms ? ms + (int64_t)chb * HW : nullptr;
#pragma unroll
  for (int r = 0; r < I::NR; ++r) {
    g.valid[r] = t.valid[r];
    const uint32_t off = (uint32_t)(t.oct[r] * 8 * HW) + t.toff[r];
#pragma unroll
    for (int c = 0; c < 8; ++c) g.v[r][c] = *reinterpret_cast<const float4*>(xc + (int64_t)c * HW + off);
    if (ms) {
#pragma unroll
      for (int c = 0; c < 8; ++c) g.m[r][c] = *reinterpret_cast<const uint32_t*>(mc + (int64_t)c * HW + off);
    }
  }
}

// the {scale, shift} pairs of the tasks' octets: channels chb + 8 oct .. + 7 of one source (arrays 16-byte aligned: sp_src_ok),
// fetched as two float4 pairs per task
template <int NR>
struct SpCoef {
  float4 s[NR][2], h[NR][2];
};
template <typename I>
__device__ __forceinline__ void sp_coef_issue(const SpTasks<I::NR>& t, SpCoef<I::NR>& k, const float* scale, const float* shift, int chb) {
#pragma unroll
  for (int r = 0; r < I::NR; ++r) {
    const float4* ps = reinterpret_cast<const float4*>(scale + chb + t.oct[r] * 8);
    const float4* ph = reinterpret_cast<const float4*>(shift + chb + t.oct[r] * 8);
    k.s[r][0] = ps[0], k.s[r][1] = ps[1], k.h[r][0] = ph[0], k.h[r][1] = ph[1];
  }
}
__device__ __forceinline__ float sp_f4(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// transform (BN + LeakyReLU, keep mask, channel multiplier -- the WslSrc loader), scale by `mul`, split, write hi / lo slots.
// cf: the thread's OWN BatchNorm coefficients {scale, shift} of the eight channels of each task, in REGISTERS, times cmul (the
// operand scale of a BatchNorm source is folded into its coefficients: exact, a power of two) -- a
// staging thread works on the same octet in every tile, so they are fetched from global memory (L1) once per kernel (weight
// gradient) or per channel chunk (conv).  (Round 3 read them from an LDS float2 table and got wrong tiles whenever a second workgroup
// shared the CU: not the table reads -- the packed FMA hipcc built from the float2 is unsafe next to f16 MFMAs on gfx950,
// profiles/r4_sp_root_cause.md; xform_bn_leaky now detaches its scalars, so either source of coefficients is safe.)  cml: this sample's channel multipliers (global
// memory), indexed by the channel inside the concatenated input (tc0 = index of channel chb).  `zero_fill`: tasks outside the
// image write zero slots.
// CMSH: the multipliers are not read from `cml` but from `cm_lane` -- lane l of every wave holds the multiplier of channel (l & 15) of the
// 16-channel chunk (requested with the tile data: a load issued HERE would have to wait for every store and prefetch in flight).
// MODE: the four loader switches as compile-time constants (bit 0 has_scale, 1 has_mask, 2 has_cm, 3 has_mul) or -1 = the run-time
// arguments.  With run-time switches hipcc if-converts the per-channel `if (has_...)` blocks: it computes EVERY transform for every value
// and selects (2 v_cndmask per value pair and switch) -- ~530 vector instructions per task where a BatchNorm source needs ~300 and a
// gradient ~120, in a kernel that is bound by its vector + matrix issue slots (profiles/r4_conv_sp_where_the_time_goes.md section 5).
// sp_commit() branches ONCE per task on the (uniform) switches to the four forms the networks use and keeps the generic one for the rest.
template <typename I, bool CMSH, int MODE>
__device__ __forceinline__ void sp_commit_mode(const SpTasks<I::NR>& t, const SpRegs<I::NR>& g, unsigned char* img, const SpCoef<I::NR>& cf,
                                               float cmul, const float* cml, int tc0, bool has_scale_, bool has_mask_, bool has_cm_, float es,
                                               bool has_mul_, float mul, bool zero_fill, float cm_lane) {
  const bool has_scale = MODE < 0 ? has_scale_ : (MODE & 1) != 0, has_mask = MODE < 0 ? has_mask_ : (MODE & 2) != 0;
  const bool has_cm = MODE < 0 ? has_cm_ : (MODE & 4) != 0, has_mul = MODE < 0 ? has_mul_ : (MODE & 8) != 0;
#pragma unroll
  for (int r = 0; r < I::NR; ++r) {
    float cmv[8];
    if constexpr (CMSH) {
      if (has_cm) {   // (uniform; every lane takes part in the shuffles)
#pragma unroll
        for (int c = 0; c < 8; ++c) cmv[c] = __shfl(cm_lane, t.oct[r] * 8 + c);
      }
    }
    if (g.valid[r]) {
      wsl_v2f a[8], b[8];   // a[c] = pixels 0, 1 of channel c; b[c] = pixels 2, 3
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        a[c] = wsl_v2f{g.v[r][c].x, g.v[r][c].y}, b[c] = wsl_v2f{g.v[r][c].z, g.v[r][c].w};
        const int ch = tc0 + t.oct[r] * 8 + c;
        if (has_scale) {
          xform_bn_leaky(a[c], b[c], sp_f4(cf.s[r][c >> 2], c & 3) * cmul, sp_f4(cf.h[r][c >> 2], c & 3) * cmul);
        }
        if (has_mask) xform_mask(a[c], b[c], g.m[r][c], es);
        if (has_cm) {
          float cm;
          if constexpr (CMSH) cm = cmv[c];
          else cm = cml[ch];
          a[c] = a[c] * cm, b[c] = b[c] * cm;
        }
        if (has_mul) a[c] = a[c] * mul, b[c] = b[c] * mul;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wsl_u4 hi, lo;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float x0 = j < 2 ? a[2 * p][j & 1] : b[2 * p][j & 1], x1 = j < 2 ? a[2 * p + 1][j & 1] : b[2 * p + 1][j & 1];
          uint32_t h, l;
          sp_split2(x0, x1, h, l);
          hi[p] = h, lo[p] = l;
        }
        unsigned char* dst = img + t.loff[r] + j * (I::P * 16);
        *reinterpret_cast<wsl_u4*>(dst) = hi;
        *reinterpret_cast<wsl_u4*>(dst + I::HL) = lo;
      }
    } else if (zero_fill && t.active[r]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned char* dst = img + t.loff[r] + j * (I::P * 16);
        *reinterpret_cast<wsl_u4*>(dst) = wsl_u4{0u, 0u, 0u, 0u};
        *reinterpret_cast<wsl_u4*>(dst + I::HL) = wsl_u4{0u, 0u, 0u, 0u};
      }
    }
  }
}

template <typename I, bool CMSH = false>
__device__ __forceinline__ void sp_commit(const SpTasks<I::NR>& t, const SpRegs<I::NR>& g, unsigned char* img, const SpCoef<I::NR>& cf,
                                          float cmul, const float* cml, int tc0, bool has_scale, bool has_mask, bool has_cm, float es, bool has_mul,
                                          float mul, bool zero_fill, float cm_lane = 1.f) {
#define WSL_SP_COMMIT(MODE_) sp_commit_mode<I, CMSH, MODE_>(t, g, img, cf, cmul, cml, tc0, has_scale, has_mask, has_cm, es, has_mul, mul, zero_fill, cm_lane)
  const int mode = (has_scale ? 1 : 0) | (has_mask ? 2 : 0) | (has_cm ? 4 : 0) | (has_mul ? 8 : 0);   // uniform
  if (mode == 1) WSL_SP_COMMIT(1);         // BatchNorm source
  else if (mode == 3) WSL_SP_COMMIT(3);    // BatchNorm source with the keep mask of its Dropout
  else if (mode == 7) WSL_SP_COMMIT(7);    // ... and the channel multipliers of a Dropout2d on top (the auxiliary decoder's skip features)
  else if (mode == 8) WSL_SP_COMMIT(8);    // plain source scaled to the operand range: gradients, the upsampled tensor
  else WSL_SP_COMMIT(-1);
#undef WSL_SP_COMMIT
}

// byte offset of staged column c inside a row
template <typename I>
__device__ __forceinline__ int sp_slot(int c) { return ((c & 3) * I::P + (c >> 2)) * 16; }

// ------------------------------------------------------------------------------------------------ weight images
// image (one per layer and direction), in 16-byte pieces of 8 halves:  [chunk = ci / 16][hl][kstep 0..4][g 0..3][co][8]
//   k-group g of K-step s holds tap 2 s + (g >> 1), input channels 16 chunk + 8 (g & 1) .. + 7; tap 9 does not exist: zeros.
// (amax zeroed by the caller; blockIdx.y = table entry, a few workgroups per layer merged with an atomic max)
__global__ __launch_bounds__(256) void sp_amax_table_kernel(PackTable t, const float* params, uint32_t* amax) {
  __shared__ float red[4];
  const PackEntry e = t.e[blockIdx.y];
  const int64_t n = (int64_t)e.Co * e.Ci * e.KK;
  const float* w = params + e.w;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    uint32_t u;
    memcpy(&u, &m, 4);
    if (u) atomicMax(amax + blockIdx.y, u);
  }
}

// blockIdx.y = table entry, blockIdx.z + dir0 = 0 forward image / 1 data-gradient image.  imgf / imgd: byte bases of the two image
// arenas; the image of entry i starts io.off[i] bytes into its arena.
struct SpPackOffsets {
  int64_t off[40];
};
__global__ __launch_bounds__(256) void sp_pack_table_kernel(PackTable t, SpPackOffsets io, const float* params, unsigned char* imgf,
                                                            unsigned char* imgd, const uint32_t* amax, int dir0) {
  const PackEntry e = t.e[blockIdx.y];
  if (e.KK != 9 || (e.Ci % 16) || (e.Co % 16)) return;   // (1x1 layers, first convolution, classifiers: not on this path)
  const int dgrad = blockIdx.z + dir0;
  const int Co = dgrad ? e.Ci : e.Co, Ci = dgrad ? e.Co : e.Ci;   // GEMM-out / GEMM-in of this image
  const float* w = params + e.w;
  wsl_u4* img = reinterpret_cast<wsl_u4*>((dgrad ? imgd : imgf) + io.off[blockIdx.y]);
  const float mul = sp_pow2(sp_exp_of(amax[blockIdx.y]));
  const int64_t pieces = (int64_t)(Ci / 16) * 5 * 4 * Co;          // per hl
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * kThreads) {
    const int co = (int)(i % Co);
    int64_t r = i / Co;
    const int g = (int)(r % 4);
    r /= 4;
    const int s = (int)(r % 5), chunk = (int)(r / 5);
    const int tap = 2 * s + (g >> 1), ci0 = chunk * 16 + (g & 1) * 8;
    wsl_u4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (tap < 9) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float x[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ci = ci0 + 2 * p + h;
          x[h] = mul * (dgrad ? w[((int64_t)ci * Co + co) * 9 + (8 - tap)] : w[((int64_t)co * Ci + ci) * 9 + tap]);
        }
        uint32_t a, b;
        sp_split2(x[0], x[1], a, b);
        hi[p] = a, lo[p] = b;
      }
    }
    const int64_t dst = ((((int64_t)chunk * 2 + 0) * 5 + s) * 4 + g) * Co + co;
    img[dst] = hi;
    img[dst + (int64_t)5 * 4 * Co] = lo;
  }
}

int sp_pack_table(const PackTable& t, const int64_t* img_off_bytes, const float* params, void* imgf, void* imgd, uint32_t* amax,
                  int with_dgrad, void* stream) {
  SpPackOffsets io;
  double bytes = 0.0;
  for (int i = 0; i < t.n; ++i) io.off[i] = img_off_bytes[i], bytes += t.e[i].KK == 9 ? 40.0 * t.e[i].Co * t.e[i].Ci : 0.0;
  ProfScope ps(PF_PREP, 0.0, bytes * (with_dgrad ? 2.9 : 1.9), stream);
  WSL_LAUNCH(sp_amax_table_kernel, dim3(16, t.n), dim3(kThreads), 0, stream, t, params, amax);   // (amax zeroed by the caller)
  WSL_LAUNCH(sp_pack_table_kernel, dim3(16, t.n, with_dgrad ? 2 : 1), dim3(kThreads), 0, stream, t, io, params,
             static_cast<unsigned char*>(imgf), static_cast<unsigned char*>(imgd), amax, 0);
  return check_launch("sp_pack_table_kernel");
}

// ------------------------------------------------------------------------------------------------ conv forward / data gradient
// Persistent workgroups: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and, inside a tile, the 16-channel
// chunks of the input.  Everything a (tile, chunk) item needs from memory is requested one item ahead and NOTHING inside the loop waits
// for anything younger (round 4; vmcnt is one in-order counter for loads, stores and LDS DMAs of a wave, so one wait for a young
// instruction drains the whole queue -- profiles/r4_conv_sp_where_the_time_goes.md section 5):
//   * the raw tile data of the NEXT item is requested (into registers) before the MFMA loop of the current one;
//   * the weight block of the next chunk goes straight into LDS (global_load_lds, issued from inline assembly so that hipcc places no
//     wait in front of the MFMA loop's LDS reads): into the OTHER of two buffers before the MFMA loop where LDS has room for two (DB:
//     blocks of <= 32 output channels), else into the one buffer right after the MFMA loop, in front of the epilogue;
//   * BatchNorm coefficients come from an LDS table filled once per workgroup, the bias from registers loaded once;
//   * barriers publish LDS only (WSL_LDS_BARRIER): the output stores of a finished tile stay in flight across them, and the one
//     explicit wait before the barrier that publishes a weight block leaves exactly those stores outstanding.
// BRES: the weight image of the block fits LDS for all chunks (Ci * CO_T <= 1024: the two full-resolution levels) and is staged once per
// workgroup.
struct ConvSpP {
  SpSrc a, b;
  const wsl_u4* img;        // this layer's weight image
  const uint32_t* w_amax;
  const uint32_t* in_amax;  // null: activation scale 2^WSL_SP_ACT_EXP
  const float* bias;
  float* y;
  int64_t y_bs;
  int N, H, W, Ci, Co, tiles_x, tiles_y, ntiles;
  float* stat_part;
  float* stat_cnt;
  BnBwdEpi bn;              // data-gradient launches: BatchNorm-backward statistics of the consumer of y
  int stagger;              // EXPERIMENTS build (env WSL_SP_STAGGER = k, WSL_SP_STAGGER_MODE): k x ~1 us of s_sleep before the tile loop for
                            // mode 0 the second half of the grid, mode 1 every second workgroup to ARRIVE on a CU (ticket per CU from HW_ID)
  int stagger_mode;
  int* cu_tickets;
  int ablate;               // EXPERIMENTS build (env WSL_SP_ABLATE; results are WRONG by design): 1 no MFMA, 2 no transform / split /
                            // LDS writes after the first commit, 4 no output stores, 8 no global loads after the first request, 16 no weight DMA after the first
};

template <typename F, int... Is>
__device__ __forceinline__ void sp_for_each_index(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

template <int TH, int TW, int CO_T>
struct ConvSpCfg {
  using Img = SpImg<TH + 2, (TW + 8) / 4, 2>;
  static constexpr int SEGS = TW / 16, MT_TOTAL = TH * SEGS, MT = MT_TOTAL / 4, NT = CO_T / 16;
  static constexpr int B_BYTES = 2 * 5 * 4 * CO_T * 16;            // one chunk: [hl][kstep][g][co][8 halves]
  static constexpr int B_PIECES = B_BYTES / 16, NBW = (B_PIECES + 255) / 256;
  static constexpr int RED_BYTES = 8 * CO_T * 4;
  // waves per SIMD the register allocator must leave room for (a tighter cap spills the staging state into scratch)
  static constexpr int MINW = (NT == 1 && Img::NR == 1) ? 3 : 2;   // (measured against 2 for the 16-channel blocks: profiles/r4_minw_ab.log)
  // streamed weight blocks double-buffered where two workgroups per CU still fit: 32 KB image + 2 x 20 KB + table <= 80 KB
  static constexpr bool DB = CO_T <= 32;
  static constexpr int COEF_BYTES_PER_CH = 8;                      // {scale, shift} x operand scale, [octet][scale x 8 | shift x 8]
  // Resident-weight kernels (<= 64 input channels = 8 octets) keep the table in the tail padding of the image planes (ROWS * ROWB = 8000 of
  // 8192 bytes used: three 64-byte octet entries per plane, four planes): the 16-wide blocks sit exactly at three workgroups per CU
  // (42 LDS allocation units of 1280 bytes each; 256 more bytes are a 43rd unit and the third workgroup is gone: measured +21 %)
  static constexpr int PLANE_USED = Img::ROWS * Img::ROWB, SLACK_OCTETS = (Img::PLANE - PLANE_USED) / 64;
  static_assert(PLANE_USED % 16 == 0 && 4 * SLACK_OCTETS >= 8, "coefficient table of <= 64 channels fits the planes' padding");
  static size_t smem(int Ci, bool bres) {
    return Img::BYTES + (size_t)(bres ? Ci / 16 : (DB ? 2 : 1)) * B_BYTES + RED_BYTES + (bres ? 0 : (size_t)Ci * COEF_BYTES_PER_CH);
  }
  static_assert(MT_TOTAL % 4 == 0 && MT % SEGS == 0 && MT % 2 == 0, "tile shape");
};

// EPI: what the epilogue of a tile emits besides the tile -- 0 nothing, 1 BatchNorm partial statistics of the output (forward launches),
// 2 BatchNorm-backward statistics of the layer that consumes this gradient (data-gradient launches; blocks of <= 32 accumulator
// registers).  One epilogue per instantiation: with all three in one kernel the register allocator spilled inside the tile loop, and a
// spill reload is a vector-memory load that waits for every prefetch in flight.
template <int TH, int TW, int CO_T, bool BRES, int EPI>
__global__ __launch_bounds__(256, (ConvSpCfg<TH, TW, CO_T>::MINW)) void conv_sp_kernel(ConvSpP p) {
  using C = ConvSpCfg<TH, TW, CO_T>;
  using I = typename C::Img;
  constexpr bool DB = !BRES && C::DB;
  WSL_DYN_SMEM(smem);
  const int Ci = p.Ci, Co = p.Co, H = p.H, W = p.W, HW = H * W;
  unsigned char* a_img = smem;
  unsigned char* b_img = smem + I::BYTES;
  float* red = reinterpret_cast<float*>(b_img + (size_t)(BRES ? Ci / 16 : (DB ? 2 : 1)) * C::B_BYTES);
  // coefficient entry of octet q (64 bytes: scale x 8 | shift x 8)
  auto ctab_octet = [&](int q) __attribute__((always_inline)) -> float* {
    if constexpr (BRES) return reinterpret_cast<float*>(a_img + (q / C::SLACK_OCTETS) * I::PLANE + C::PLANE_USED + (q % C::SLACK_OCTETS) * 64);
    else return red + C::RED_BYTES / 4 + q * 16;
  };
  const int tid = threadIdx.x, lane = tid & 63, wave = WSL_WAVE_UNIFORM(tid >> 6);
  const int co0 = blockIdx.y * CO_T;
  const int nb = p.ntiles;

  // operand scale of the input: a plain source (a gradient: data-gradient launches) takes it from the tensor's tracked maximum; BatchNorm
  // sources the static 2^WSL_SP_ACT_EXP (normalised values stay far below 4094).  Both at once -- the decoder blocks' first convolution:
  // BatchNorm-ed skip + the raw upsampled tensor, whose maximum is tracked -- share ONE scale (one accumulator): the smaller
  int e_in = WSL_SP_ACT_EXP;
  if (p.in_amax) {
    const int e_t = sp_exp_of(sp_amax_fold(p.in_amax));
    e_in = (p.a.scale || p.b.scale) ? (e_t < WSL_SP_ACT_EXP ? e_t : WSL_SP_ACT_EXP) : e_t;
  }
  const int e_w = sp_exp_of(*p.w_amax);
  const float in_mul = sp_pow2(e_in);
  const float u1 = sp_pow2(-e_w), u2 = sp_pow2(-e_in);

  // loader coefficients of every input channel (concatenated index), operand scale folded in (exact: a power of two)
  for (int c = tid; c < Ci; c += kThreads) {
    const bool ina = c < p.a.C;
    const SpSrc& s = ina ? p.a : p.b;
    const int cc = ina ? c : c - p.a.C;
    float* q = ctab_octet(c >> 3) + (c & 7);
    q[0] = s.scale ? s.scale[cc] * in_mul : 0.f;
    q[8] = s.scale ? s.shift[cc] * in_mul : 0.f;
  }
  // bias of this lane's output columns
  float biasv[C::NT];
#pragma unroll
  for (int j = 0; j < C::NT; ++j) biasv[j] = p.bias ? p.bias[co0 + j * 16 + (lane & 15)] : 0.f;

  // weight pieces of a chunk: LDS piece u = ((hl * 5 + s) * 4 + g) * CO_T + col  <-  image piece ((hl * 5 + s) * 4 + g) * Co + co0 + col
  const int64_t chunk_pieces = (int64_t)40 * Co;
  if constexpr (BRES) {   // every chunk's weights, once
    uint32_t woff[C::NBW];
#pragma unroll
    for (int i = 0; i < C::NBW; ++i) {
      const int u = tid + i * kThreads;
      const int col = u % CO_T, rest = u / CO_T;
      woff[i] = u < C::B_PIECES ? (uint32_t)(rest * Co + co0 + col) : 0u;
    }
    for (int ch = 0; ch < Ci / 16; ++ch) {
      const wsl_u4* wb = p.img + (int64_t)ch * chunk_pieces;
#pragma unroll
      for (int i = 0; i < C::NBW; ++i)
        if ((i + 1) * kThreads <= C::B_PIECES || tid + i * kThreads < C::B_PIECES)
          reinterpret_cast<wsl_u4*>(b_img + (size_t)ch * C::B_BYTES)[tid + i * kThreads] = wb[woff[i]];
    }
  }
  // A operand of K-step s: pixel m = l & 15 of the row tile, octet (l >> 4) & 1, tap 2 s + (l >> 5); B operand: k-group l >> 4, column l & 15
  int aoff[5], bbase = 0;
  auto operand_offsets = [&](int l) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + (l >> 5) < 9 ? 2 * s + (l >> 5) : 8;   // (the tenth tap meets a zero weight block)
      const int ky = tap / 3, kx = tap - 3 * ky;
      aoff[s] = ((l >> 4) & 1) * I::PLANE + ky * I::ROWB + sp_slot<I>(3 + (l & 15) + kx);
    }
    bbase = ((l >> 4) * CO_T + (l & 15)) * 16;
  };
  // The 64-wide block with the statistics epilogue is the one instantiation whose register file is full: it recomputes the six offsets
  // per item (a dozen instructions in front of 240 MFMAs) from a lane index hipcc cannot trace back to threadIdx -- kept across the tile
  // loop they are live through the staging code -- and fences the scheduler between K-steps (below); without both it spills
  constexpr bool kTightRegs = C::NT == 4 && EPI == 1;
  constexpr bool kOffsetsPerItem = kTightRegs;
  if constexpr (!kOffsetsPerItem) operand_offsets(lane);

  SpTasks<I::NR> tk;
  SpRegs<I::NR> pre;
  float pre_cm = 1.f;
  // Tile order: workgroup b runs on XCD b % 8 (observed dispatch; speed only) -- give every XCD one contiguous eighth of the tiles and
  // let its workgroups walk it side by side, so the halo lines two neighbouring tiles share meet in ONE L2 instead of being fetched
  // over the fabric by three (measured: the loads alone of 16 -> 16 @ 256 x 256 took 118 us with tiles dealt round-robin)
  const bool xcd = (nb & 7) == 0 && (gridDim.x & 7) == 0;
  const int tstep = xcd ? gridDim.x >> 3 : gridDim.x;
  const int tend = xcd ? ((int)(blockIdx.x & 7) + 1) * (nb >> 3) : nb;
  // (tile, chunk) whose raw data sits in `pre`
  int nt = xcd ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x, nc0 = 0, n_n = 0;
  const int t_first = nt;
  auto issue = [&]() __attribute__((always_inline)) {
    if (nc0 == 0) {   // first chunk of a tile: its coordinates
      const int tx = nt % p.tiles_x, r = nt / p.tiles_x;
      const int ty = r % p.tiles_y;
      n_n = r / p.tiles_y;
      sp_tasks_init<I>(tk, tid, ty * TH - 1, tx * TW - 4, H, W);
    }
    const bool ina = nc0 < p.a.C;   // uniform
    const SpSrc& s = ina ? p.a : p.b;
    const int chb = ina ? nc0 : nc0 - p.a.C;
    sp_issue<I>(tk, pre, s.x + n_n * s.bs, s.emask ? s.emask + (int64_t)n_n * s.C * HW : nullptr, chb, HW);
    // channel multipliers of the chunk (Dropout2d of a skip feature): lane l holds channel (l & 15); the commit fetches its eight by shuffle
    if (s.cmask) pre_cm = s.cmask[(int64_t)n_n * s.C + chb + (lane & 15)];
  };

  // streamed weights: one chunk's block goes straight into LDS buffer `buf` (lane-linear pieces, no registers, no ds_write, and
  // invisible to hipcc's wait insertion: the wait is vm_wait_weights() below)
  auto dma_weights = [&](int c0, int buf) __attribute__((always_inline)) {
    const wsl_u4* wb = p.img + (int64_t)(c0 >> 4) * chunk_pieces + co0;
    unsigned char* dst = b_img + (size_t)buf * C::B_BYTES;
    // piece u = tid + 256 i of the block lies (u / CO_T) * Co + u % CO_T pieces into the image chunk; 256 % CO_T == 0, so piece i of a
    // thread is its piece 0 plus i * (256 / CO_T) * Co -- a UNIFORM step that goes into the scalar base.  One vector offset per item (four
    // instructions, from a thread index hipcc cannot trace back to threadIdx: hoisted out of the tile loop it would be one more register
    // live across everything) instead of ten 64-bit vector address computations: the kernel is short of vector issue slots.
    unsigned tv = (unsigned)tid;
    WSL_DETACH32(tv);
    const uint32_t off0 = ((tv / (unsigned)CO_T) * (unsigned)Co + (tv % (unsigned)CO_T)) * 16u;
    static_assert(kThreads % CO_T == 0, "piece step is uniform");
#pragma unroll
    for (int i = 0; i < C::NBW; ++i)
      if ((i + 1) * kThreads <= C::B_PIECES || (i * kThreads + wave * 64) < C::B_PIECES)   // whole waves (B_PIECES % 64 == 0)
        WSL_LDS_DMA16_UNTRACKED_SO(wb + (int64_t)i * (kThreads / CO_T) * Co, off0, dst + (size_t)(i * kThreads + wave * 64) * 16);
  };

#if defined(WSL_EXPERIMENTS) && !defined(WSL_HOST_EMUL)
  if (p.stagger > 0) {
    bool late = blockIdx.x >= gridDim.x / 2;
    if (p.stagger_mode == 1) {
      // HW_ID (hwreg 4): CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID (hwreg 20) [3:0]
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      const unsigned key = ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
      int ticket = 0;
      if (tid == 0) ticket = atomicAdd(p.cu_tickets + key, 1);
      int* flag = reinterpret_cast<int*>(red);   // (free until the first epilogue; no static LDS: the 16-wide blocks sit at an allocation-unit edge)
      if (tid == 0) *flag = ticket & 1;
      __syncthreads();
      late = *flag != 0;
      __syncthreads();
    }
    if (late)
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
  }
#endif
  v4f acc[C::MT][C::NT];
  int buf = 0;            // streamed weights: the buffer of the current chunk
  bool stored = false;    // a tile's output stores were issued AFTER the weight DMA now awaited (uniform)
  if (nt < tend) {
    issue();
    if constexpr (!BRES) dma_weights(0, 0);
  }
  __syncthreads();   // resident weights and the coefficient table visible

  while (nt < tend) {
    const int t = nt, c0 = nc0, n = n_n;
    {   // ---- raw data -> hi / lo images
      const bool ina = c0 < p.a.C;
      const SpSrc& s = ina ? p.a : p.b;
      SpCoef<I::NR> coef;
      if (s.scale) {
#pragma unroll
        for (int r = 0; r < I::NR; ++r) {
          const float4* q = reinterpret_cast<const float4*>(ctab_octet((c0 >> 3) + tk.oct[r]));
          coef.s[r][0] = q[0], coef.s[r][1] = q[1], coef.h[r][0] = q[2], coef.h[r][1] = q[3];
        }
      }
      if (!WSL_ABLATED(p, 2) || (t == t_first && c0 == 0))
        sp_commit<I, true>(tk, pre, a_img, coef, 1.f, nullptr, c0, s.scale != nullptr, s.emask != nullptr, s.cmask != nullptr, s.es,
                     s.scale == nullptr, in_mul, true, pre_cm);
      if constexpr (!BRES) {
        // this chunk's weight block has landed: everything of this wave but the output stores issued after its DMA is complete
        // (C::MT * C::NT float4 stores per wave and tile, all unconditional; stores of single lanes that follow them are younger still)
        if (stored) WSL_VM_WAIT(C::MT * C::NT);
        else WSL_VM_WAIT(0);
        stored = false;
      }
    }
    WSL_LDS_BARRIER();
    // ---- request the next (tile, chunk)
    if (c0 + 16 < Ci) nc0 = c0 + 16;
    else nc0 = 0, nt = t + tstep;
    if (nt < tend && !WSL_ABLATED(p, 8)) issue();   // in flight during the MFMA loop, the epilogue's stores and the statistics
    if constexpr (DB) if (nt < tend && !WSL_ABLATED(p, 16)) dma_weights(nc0, buf ^ 1);
    if (c0 == 0) {
#pragma unroll
      for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* b_c = b_img + (BRES ? (size_t)(c0 >> 4) * C::B_BYTES : (size_t)buf * C::B_BYTES);
    if constexpr (kOffsetsPerItem) {
      int lv = lane;
      WSL_DETACH32(lv);
      operand_offsets(lv);
    }
    // blocks of <= 32 output channels: operands of the next group of MFMAs read while the current group issues, the order pinned with
    // sched_group_barrier (+0.45 % on the split step against hipcc's own placement; profiles/r4_conv_sp_where_the_time_goes.md section 7)
#ifndef WSL_HOST_EMUL
#define WSL_SP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define WSL_SP_SGB(mask, n)
#endif
    constexpr bool kPipe = C::NT <= 2;
    if constexpr (kPipe) {
      // groups g = (K-step s, row-tile pair i0): A operands double-buffered per group, B operands per K-step
      constexpr int GPS = C::MT / 2, G = 5 * GPS;
      wsl_u4 Ah[2][2], Al[2][2], Bh[2][C::NT], Bl[2][C::NT];
      auto load_b = [&](int s, int bb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
          const unsigned char* q = b_c + bbase + (s * 4 * CO_T + j * 16) * 16;
          Bh[bb][j] = *reinterpret_cast<const wsl_u4*>(q);
          Bl[bb][j] = *reinterpret_cast<const wsl_u4*>(q + 5 * 4 * CO_T * 16);
        }
      };
      auto load_a = [&](int g, int ab) __attribute__((always_inline)) {
        const int s = g / GPS, i0 = 2 * (g % GPS);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int mt = wave * C::MT + i0 + d;
          const unsigned char* q = a_img + aoff[s] + (mt / C::SEGS) * I::ROWB + (mt % C::SEGS) * 64;
          Ah[ab][d] = *reinterpret_cast<const wsl_u4*>(q), Al[ab][d] = *reinterpret_cast<const wsl_u4*>(q + I::HL);
        }
      };
      auto group = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, s = g / GPS, i0 = 2 * (g % GPS);
        constexpr bool more = g + 1 < G, new_step = more && (g + 1) % GPS == 0;
        constexpr int nload = (more ? 4 : 0) + (new_step ? 2 * C::NT : 0), NM = 6 * C::NT, il = nload < NM ? nload : NM;
        if constexpr (new_step) load_b(s + 1, (s + 1) & 1);
        if constexpr (more) load_a(g + 1, (g + 1) & 1);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int i = i0 + d;
#pragma unroll
          for (int j = 0; j < C::NT; ++j) {
            acc[i][j] = WSL_MFMA_F16(Ah[g & 1][d], Bl[s & 1][j], acc[i][j]);
            acc[i][j] = WSL_MFMA_F16(Al[g & 1][d], Bh[s & 1][j], acc[i][j]);
            acc[i][j] = WSL_MFMA_F16(Ah[g & 1][d], Bh[s & 1][j], acc[i][j]);
          }
        }
        // issue order of this group: one operand read of the NEXT group behind each of the first MFMAs, the remaining MFMAs after
#pragma unroll
        for (int k = 0; k < il; ++k) {
          WSL_SP_SGB(0x008, 1);
          WSL_SP_SGB(0x100, 1);
        }
        if constexpr (nload > il) WSL_SP_SGB(0x100, nload - il);
        if constexpr (NM > il) WSL_SP_SGB(0x008, NM - il);
      };
      if (!WSL_ABLATED(p, 1)) {
        load_b(0, 0);
        load_a(0, 0);
        WSL_SP_SGB(0x100, 2 * C::NT + 4);
        sp_for_each_index(std::make_integer_sequence<int, G>{}, group);
      }
    } else
    if (!WSL_ABLATED(p, 1))
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      wsl_u4 bh[C::NT], bl[C::NT];
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        const unsigned char* q = b_c + bbase + (s * 4 * CO_T + j * 16) * 16;
        bh[j] = *reinterpret_cast<const wsl_u4*>(q);
        bl[j] = *reinterpret_cast<const wsl_u4*>(q + 5 * 4 * CO_T * 16);
      }
      // pairs of row tiles: four operand reads, 6 NT MFMAs
#pragma unroll
      for (int i0 = 0; i0 < C::MT; i0 += 2) {
        wsl_u4 ah[2], al[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int mt = wave * C::MT + i0 + d;
          const unsigned char* q = a_img + aoff[s] + (mt / C::SEGS) * I::ROWB + (mt % C::SEGS) * 64;
          ah[d] = *reinterpret_cast<const wsl_u4*>(q), al[d] = *reinterpret_cast<const wsl_u4*>(q + I::HL);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int i = i0 + d;
#pragma unroll
          for (int j = 0; j < C::NT; ++j) {
            acc[i][j] = WSL_MFMA_F16(ah[d], bl[j], acc[i][j]);
            acc[i][j] = WSL_MFMA_F16(al[d], bh[j], acc[i][j]);
            acc[i][j] = WSL_MFMA_F16(ah[d], bh[j], acc[i][j]);
          }
        }
      }
      // keep the scheduler from pulling the next K-steps' operand reads (40 registers) up into this one -- with them that kernel spills
      // inside the tile loop, and a spill reload is a vector-memory load that waits for every prefetch in flight
      if constexpr (kTightRegs) WSL_SCHED_BARRIER();
    }
    WSL_LDS_BARRIER();   // the tile image and this chunk's weight block are free again
    if constexpr (!BRES && !DB) if (nt < tend && !WSL_ABLATED(p, 16)) dma_weights(nc0, 0);
    if constexpr (DB) buf ^= 1;
    if (c0 + 16 >= Ci) {
      // ---- epilogue of tile t: undo the operand scales, bias, statistics, float4 stores (tiles and channel blocks are full).  Loads
      // first, stores last: a load waits for every older store of the wave.
      const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y;
      const int y0 = ty * TH, x0 = tx * TW;
      float bsum[C::NT];
      constexpr int RPW = C::MT / C::SEGS;   // output rows per wave
#pragma unroll
      for (int j = 0; j < C::NT; ++j) {
        float bs = 0.f;
#pragma unroll
        for (int i = 0; i < C::MT; ++i) {
          v4f v = acc[i][j];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (v[r] * u1) * u2 + biasv[j];
          acc[i][j] = v;
          bs += (v[0] + v[1]) + (v[2] + v[3]);
        }
        bsum[j] = bs;
      }
      float s1[C::NT], s2[C::NT];
      if constexpr (EPI == 2) {
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
      }
      {
        // (uniform base per store + one 32-bit lane offset: see sp_issue)
        float* yb = p.y + n * p.y_bs + (int64_t)co0 * HW + (int64_t)(y0 + wave * RPW) * W + x0;
        int lv = lane;
        WSL_DETACH32(lv);   // (recomputed per tile: hoisted out of the tile loop it is one more register live across everything)
        const uint32_t lane_off = (uint32_t)((lv & 15) * HW + (lv >> 4) * 4);
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
          float* yj = yb + (int64_t)j * 16 * HW;
#pragma unroll
          for (int i = 0; i < C::MT; ++i) {
            const v4f v = acc[i][j];
            if (!WSL_ABLATED(p, 4) || v[0] == 123.456f)
              *reinterpret_cast<float4*>(yj + (i / C::SEGS) * W + (i % C::SEGS) * 16 + lane_off) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
        // WSL_VM_WAIT(C::MT * C::NT) above counts exactly THESE stores: it may only be taken when every one of them was issued
        // (the ablation build skips them), and the count must fit the 6-bit vmcnt field
        static_assert(C::MT * C::NT <= 63, "vmcnt field");
        stored = !WSL_ABLATED(p, 4);
      }
      if constexpr (EPI == 2) bn_bwd_store<C::NT, CO_T, true>(p.bn, s1, s2, red, co0, Co, t, nb);
      if constexpr (EPI == 1) {
        float* red1 = red;
        float* red2 = red + 4 * CO_T;
        constexpr float cnt = (float)(TH * TW);
#pragma unroll
        for (int j = 0; j < C::NT; ++j) {
          float s = bsum[j];
          s += __shfl_xor(s, 16);
          s += __shfl_xor(s, 32);
          if (lane < 16) red1[wave * CO_T + j * 16 + lane] = s;
        }
        WSL_LDS_BARRIER();
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
        WSL_LDS_BARRIER();
        if (wave == 0 && lane < 16) {
#pragma unroll
          for (int j = 0; j < C::NT; ++j) {
            const int col = j * 16 + lane, co = co0 + col;
            float* dst = p.stat_part + ((int64_t)co * nb + t) * 2;   // [Co][nblk][2]
            dst[0] = red1[col] + red1[CO_T + col] + red1[2 * CO_T + col] + red1[3 * CO_T + col];
            dst[1] = red2[col] + red2[CO_T + col] + red2[2 * CO_T + col] + red2[3 * CO_T + col];
          }
          if (lane == 0 && blockIdx.y == 0) p.stat_cnt[t] = cnt;
        }
      }
    }
  }
}

struct SpPlan {
  int th, tw, co_t;
  bool ok;
};
// tile shape per layer: 8 x 32 pixels (8 x 16 below 32 columns), the widest output-channel block that still leaves >= 2
// workgroups per CU
static SpPlan sp_plan(int N, int H, int W, int Ci, int Co, bool want_bn_epilogue = false) {
  SpPlan f{0, 0, 0, false};
  if (Ci <= 0 || Co <= 0 || (Ci % 16) || (Co % 16) || Ci > kSpMaxC) return f;
  // (4 x 128 tiles for the 16-channel layers -- 544-byte instead of 160-byte runs per halo row, 3.7 against 2.5 TB/s of pure tile fetch in
  //  tools/probe_tile_loads.hip -- were built and measured in round 4: 204 / 291 us against 198 / 273 for 16 -> 16 / 32 -> 16 @ 256 x 256.
  //  The wider tile costs a second round of staging registers and the third workgroup per CU; profiles/r4_conv_sp_where_the_time_goes.md)
  if (H % 8 == 0 && W % 32 == 0) f.th = 8, f.tw = 32;
  else if (H % 8 == 0 && W % 16 == 0) f.th = 8, f.tw = 16;
  else return f;
  const int64_t tiles = (int64_t)N * (H / f.th) * (W / f.tw);
  f.co_t = 16;
  if (Co % 32 == 0) f.co_t = 32;
  // 64-wide blocks wherever they still fill the chip twice: a staged tile (its ~250 vector instructions per thread and chunk are what the
  // kernel is short of) then feeds 240 / 120 MFMAs per wave instead of 120 / 60
  // (16-column tiles: not where the caller wants the BatchNorm-backward statistics from the epilogue -- only blocks of <= 2 column tiles
  //  have the registers for them, and at 16 x 16 the stand-alone reduction pass costs more than the wider block saves)
  if (Co % 64 == 0 && tiles * (Co / 64) >= 2 * device_cu_count() && !(want_bn_epilogue && f.tw == 16)) f.co_t = 64;
  f.ok = true;
  return f;
}

// workgroups of `smem` bytes of LDS that stay resident on one CU, capped by the register budget (`minw` waves per SIMD).  LDS is handed
// out in units of 1280 bytes (inferred: 42 units x 3 workgroups ran three per CU, 43 x 3 did not; tests/test_ops_convsp.py pins the
// layer shapes that sit at such an edge through wsl_debug_sp_conv_residency)
static int sp_resident_per_cu(size_t smem, int minw) {
  const int by_lds = (int)((size_t)160 * 1024 / (((smem + 1279) / 1280) * 1280));
  return by_lds < minw ? by_lds : minw;
}

template <int TH, int TW, int CO_T, bool BRES, int EPI>
static int launch_conv_sp(ConvSpP& p, int is_dgrad, void* stream) {
  using C = ConvSpCfg<TH, TW, CO_T>;
  auto kern = conv_sp_kernel<TH, TW, CO_T, BRES, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::smem(BRES ? 1024 / CO_T : kSpMaxC, BRES) + 4096);
    attr_done = true;
  }
  const size_t smem = C::smem(p.Ci, BRES);
  // persistent: as many workgroups as stay resident (registers: MINW per SIMD; LDS), spread over the output-channel blocks
  int per_cu = sp_resident_per_cu(smem, C::MINW);
  static const int percu_env = WSL_TUNE("WSL_SP_PERCU", 0);   // (experiments build)
  if (percu_env > 0) per_cu = percu_env;
  if (per_cu < 1) per_cu = 1;
  const int co_blocks = p.Co / CO_T;
  int gx = per_cu * device_cu_count() / co_blocks;
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  if (gx >= 8) gx &= ~7;   // the XCD-wise tile walk wants a multiple of eight
  dim3 grid(gx, co_blocks);
  const double px = (double)p.N * p.H * p.W;
  // issued = the three f16 passes over the tap-padded K (10 / 9)
  void* tok = prof_begin(is_dgrad ? PF_SP_DGRAD : PF_SP_FWD, 2.0 * px * p.Co * p.Ci * 9, 4.0 * px * (p.Co + p.Ci) + bn_epi_bytes(p.bn, px * p.Co), stream,
                         2.0 * px * p.Co * p.Ci * 10 * 3);
  WSL_LAUNCH(kern, grid, dim3(kThreads), smem, stream, p);
  prof_end(tok, stream);
  return check_launch("conv_sp_kernel");
}

template <int TH, int TW, int CO_T, bool BRES>
static int launch_conv_sp_epi(ConvSpP& p, int epi, int is_dgrad, void* stream) {
  if constexpr (TH * TW * CO_T <= 8192 && CO_T <= 32)   // instantiations with <= 32 accumulator registers in <= 2 column tiles
    if (epi == 2) return launch_conv_sp<TH, TW, CO_T, BRES, 2>(p, is_dgrad, stream);
  if (epi == 1) return launch_conv_sp<TH, TW, CO_T, BRES, 1>(p, is_dgrad, stream);
  return launch_conv_sp<TH, TW, CO_T, BRES, 0>(p, is_dgrad, stream);
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

static bool sp_src_ok(const WslSrc& s) {
  return s.x && aligned16(s.x) && !(s.bs & 3) && (!s.emask || !(reinterpret_cast<uintptr_t>(s.emask) & 3)) &&
         (!s.scale || (s.shift && aligned16(s.scale) && aligned16(s.shift)));   // coefficient octets are fetched as float4 pairs
}

static int sp_conv_launch(const WslSrc& a, const WslSrc* b, const void* image, const uint32_t* w_amax, const uint32_t* in_amax,
                          const float* bias, float* y, int64_t y_bs, int N, int H, int W, int Co, int is_dgrad, float* stat_part,
                          float* stat_cnt, const BnBwdEpi* bn, int* bn_done, void* stream) {
  ConvSpP p;
  p.a = to_spsrc(a);
  p.b = (b && b->C > 0) ? to_spsrc(*b) : SpSrc{};
  p.img = static_cast<const wsl_u4*>(image), p.w_amax = w_amax, p.in_amax = in_amax;
  p.bias = bias, p.y = y, p.y_bs = y_bs, p.N = N, p.H = H, p.W = W, p.Ci = a.C + p.b.C, p.Co = Co;
  const SpPlan f = sp_plan(N, H, W, p.Ci, Co, bn && bn->part);
  p.tiles_x = W / f.tw, p.tiles_y = H / f.th, p.ntiles = N * p.tiles_x * p.tiles_y;
  static const int ablate = WSL_TUNE("WSL_SP_ABLATE", 0);
  p.ablate = ablate;
  static const int stagger = WSL_TUNE("WSL_SP_STAGGER", 0), stagger_mode = WSL_TUNE("WSL_SP_STAGGER_MODE", 0);
  p.stagger = stagger, p.stagger_mode = stagger_mode, p.cu_tickets = nullptr;
#if defined(WSL_EXPERIMENTS) && !defined(WSL_HOST_EMUL)
  if (stagger > 0 && stagger_mode == 1) {
    static int* tickets = nullptr;
    if (!tickets) {
      (void)hipMalloc(&tickets, 4096 * sizeof(int));
      (void)hipMemset(tickets, 0, 4096 * sizeof(int));
    }
    p.cu_tickets = tickets;
  }
#endif
  p.stat_part = stat_part, p.stat_cnt = stat_cnt;
  if (bn && bn->part && f.th * f.tw * f.co_t <= 8192 && f.co_t <= 32) p.bn = *bn;   // instantiations with <= 32 accumulator registers in <= 2 column tiles
  if (bn_done) *bn_done = p.bn.part ? 1 : 0;
  const bool bres = p.Ci * f.co_t <= 1024;   // the block's weight image of every chunk stays in LDS (<= 40 KB)
  const int epi = p.bn.part ? 2 : (p.stat_part ? 1 : 0);
#define WSL_CASE(TH_, TW_, CO_)                                                    \
  if (f.th == TH_ && f.tw == TW_ && f.co_t == CO_)                                 \
    return bres ? launch_conv_sp_epi<TH_, TW_, CO_, true>(p, epi, is_dgrad, stream) : launch_conv_sp_epi<TH_, TW_, CO_, false>(p, epi, is_dgrad, stream);
  WSL_CASE(8, 32, 16) WSL_CASE(8, 32, 32) WSL_CASE(8, 32, 64) WSL_CASE(8, 16, 16) WSL_CASE(8, 16, 32) WSL_CASE(8, 16, 64)
#undef WSL_CASE
  set_error("sp_conv: no kernel for tile %dx%d co_t %d", f.th, f.tw, f.co_t);
  return WSL_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][ci][tap] = sum_px dy[co][px] * in[ci][px + tap]:  M = 16 output channels, N = 16 input channels, K = 32 pixels of the tile.
// Both operands come out of channel-innermost tile images through ds_read_b64_tr_b16 (lane = channel, 4 pixels per read); a tap
// is a slot offset of the input image.  Persistent workgroups over a run of tiles, next tile prefetched into registers.
//   CB = 32: a 32 x 32 channel block; wave w owns the (co tile w >> 1, ci tile w & 1) pair over every pixel of a tile;
//   CB = 16: one pair; the four waves take every fourth K-step and write four partials of their own (4 * nsplit splits).
struct WgradSpP {
  SpSrc a, b;
  const float* dy;
  int64_t dy_bs;
  const uint32_t* dy_amax;
  const uint32_t* in_amax;   // tracked maximum of a raw input source (null: static activation scale)
  float* part_dw;   // [splits][9][Co][Ci]
  float* part_db;   // [splits][Co]
  int N, H, W, Ci, Co, tiles_x, tiles_y, items, nsplit, ci_blocks;
};

template <int TH, int TW, int CB>
struct WgradSpCfg {
  // Round 6: at 32-pixel rows the images take a pixel pitch of 12 quads and planes 16 bytes off a multiple of 256 -- the two octets of a
  // transpose read and the four pixels of a quad then fall on different bank groups (tools/lds_tr_conflicts.py: 6.0 -> 3.0 LDS clocks per
  // read; measured: bank-conflict share of LDS-active cycles 0.524 -> 0.289, 118.4 -> 115.3 us per launch, the split step unchanged within
  // 0.1 %; 51 -> 66 KB per workgroup, still two per CU).  16-pixel rows keep the compact images (the padded ones would cost a workgroup per CU).
  using In = SpImg<TH + 2, (TW + 8) / 4, CB / 8, (TW == 32 ? 12 : (TW + 8) / 4), (TW == 32 ? 16 : 0)>;
  using Dy = SpImg<TH, TW / 4, CB / 8, (TW == 32 ? 12 : TW / 4), (TW == 32 ? 16 : 0)>;
  static constexpr int KS = TH * TW / 32, KROWS = 32 / TW;      // K-steps per tile; tile rows per K-step
  static constexpr size_t SMEM = In::BYTES + Dy::BYTES;
  static_assert(TW == 16 || TW == 32, "a K-step is one row of 32 pixels or two rows of 16");
  static_assert(CB == 32 || KS % 4 == 0, "the one-pair form splits the K-steps over the four waves");
};

template <int TH, int TW, int CB>
__global__ __launch_bounds__(256, 2) void wgrad_sp_kernel(WgradSpP p) {
  using C = WgradSpCfg<TH, TW, CB>;
  using II = typename C::In;
  using ID = typename C::Dy;
  WSL_DYN_SMEM(smem);
  unsigned char* in_img = smem;
  unsigned char* dy_img = smem + II::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = WSL_WAVE_UNIFORM(tid >> 6);
  // (nsplit % 8 == 0: the workgroups of one XCD -- dealt round-robin in launch order -- take all channel blocks of nsplit / 8 consecutive
  //  splits and split s walks the tiles s, s + nsplit, ...: wgrad_wino_kernel's order, profiles/r5_wgrad_item_order.md)
  int blk = blockIdx.x, split = blockIdx.y;
  const bool interleaved = (p.nsplit & 7) == 0;
  if (interleaved) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, j = lin >> 3;
    blk = j % (int)gridDim.x;
    split = (lin & 7) * (p.nsplit >> 3) + j / (int)gridDim.x;
  }
  const int cob = blk / p.ci_blocks, cib = blk - cob * p.ci_blocks;
  const int H = p.H, W = p.W, Ci = p.Ci, Co = p.Co, HW = H * W;
  const int ci0 = cib * CB, co0 = cob * CB;
  const bool ina = ci0 < p.a.C;                           // the block's input channels live in one source
  const SpSrc& src = ina ? p.a : p.b;
  const int chb = ina ? ci0 : ci0 - p.a.C;
  const int e_dy = sp_exp_of(sp_amax_fold(p.dy_amax));
  int e_act = WSL_SP_ACT_EXP;   // (the conv kernel's rule: one scale for both sources, lowered when a raw source's maximum asks for it)
  if (p.in_amax) {
    const int e_t = sp_exp_of(sp_amax_fold(p.in_amax));
    e_act = e_t < WSL_SP_ACT_EXP ? e_t : WSL_SP_ACT_EXP;
  }
  const float dy_mul = sp_pow2(e_dy), act_mul = sp_pow2(e_act);

  // operand addresses of K-step 0: supplier lane s = lane & 15 of a 16-lane group hands out pixel (s >> 2) (+ 4 for the second
  // read) of the group's eight, channels 4 (s & 3) .. + 3 of the 16-channel tile
  const int cot = CB == 32 ? wave >> 1 : 0, cit = CB == 32 ? wave & 1 : 0;
  const int sup = lane & 15, grp = lane >> 4;
  int dyo[2], ino[2][3];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int pk = 8 * grp + 4 * rd + (sup >> 2);
    const int row = pk / TW, col = pk % TW, o = (sup & 3) >> 1, bo = (sup & 1) * 8;
    dyo[rd] = (2 * cot + o) * ID::PLANE + row * ID::ROWB + sp_slot<ID>(col) + bo;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) ino[rd][kx] = (2 * cit + o) * II::PLANE + row * II::ROWB + sp_slot<II>(col + kx + 3) + bo;
  }

  v4f acc[9], accdb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  const bool want_db = p.part_db != nullptr && cib == 0 && cit == 0;
  const wsl_u4 ones = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // f16 1.0

  SpTasks<II::NR> tki;
  SpTasks<ID::NR> tkd;
  SpRegs<II::NR> pri;
  SpRegs<ID::NR> prd;
  int n = 0;
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int tx = t % p.tiles_x, r = t / p.tiles_x;
    const int ty = r % p.tiles_y;
    n = r / p.tiles_y;
    sp_tasks_init<II>(tki, tid, ty * TH - 1, tx * TW - 4, H, W);
    sp_tasks_init<ID>(tkd, tid, ty * TH, tx * TW, H, W);
    sp_issue<II>(tki, pri, src.x + n * src.bs, src.emask ? src.emask + (int64_t)n * src.C * HW : nullptr, chb, HW);
    sp_issue<ID>(tkd, prd, p.dy + n * p.dy_bs, nullptr, co0, HW);
  };
  // neighbouring tiles share their halo lines: the workgroups of an XCD stage neighbouring tiles in the same round, so those lines come
  // out of its L2 (rounds 3-4: one contiguous run of tiles per split -- by the time a workgroup came back to a line, the XCD's other
  // workgroups had pushed it out); launches whose split count does not divide by eight keep the contiguous runs
  const int run = (p.items + p.nsplit - 1) / p.nsplit;
  const int step = interleaved ? p.nsplit : 1;
  int t = interleaved ? split : split * run;
  const int t_end = interleaved ? p.items : ((t + run < p.items) ? t + run : p.items);
  if (t < t_end) issue(t);
  // a thread stages the same octet of every tile: its eight BatchNorm coefficient pairs stay in registers for the whole kernel
  SpCoef<II::NR> cfr;
  if (src.scale && t < t_end) sp_coef_issue<II>(tki, cfr, src.scale, src.shift, chb);
  while (t < t_end) {
    sp_commit<II>(tki, pri, in_img, cfr, act_mul, src.cmask ? src.cmask + (int64_t)n * src.C + chb : nullptr, 0, src.scale != nullptr,
                  src.emask != nullptr, src.cmask != nullptr, src.es, src.scale == nullptr, act_mul, true);
    sp_commit<ID>(tkd, prd, dy_img, SpCoef<ID::NR>{}, 1.f, nullptr, 0, false, false, false, 1.f, true, dy_mul, true);
    __syncthreads();
    const int tn = t + step;
    if (tn < t_end) issue(tn);   // in flight during the MFMA phase
    auto read_a = [&](int ks, wsl_u4& ah, wsl_u4& al) __attribute__((always_inline)) {
      const int dk = ks * C::KROWS * ID::ROWB;
      const wsl_u2 a0 = WSL_DS_READ_TR16(dy_img + dyo[0] + dk), a1 = WSL_DS_READ_TR16(dy_img + dyo[1] + dk);
      const wsl_u2 a2 = WSL_DS_READ_TR16(dy_img + ID::HL + dyo[0] + dk), a3 = WSL_DS_READ_TR16(dy_img + ID::HL + dyo[1] + dk);
      ah = wsl_u4{a0[0], a0[1], a1[0], a1[1]}, al = wsl_u4{a2[0], a2[1], a3[0], a3[1]};
    };
    auto read_b = [&](int row_off, int kx, wsl_u4& bh, wsl_u4& bl) __attribute__((always_inline)) {
      const unsigned char* q = in_img + row_off;
      const wsl_u2 b0 = WSL_DS_READ_TR16(q + ino[0][kx]), b1 = WSL_DS_READ_TR16(q + ino[1][kx]);
      const wsl_u2 b2 = WSL_DS_READ_TR16(q + II::HL + ino[0][kx]), b3 = WSL_DS_READ_TR16(q + II::HL + ino[1][kx]);
      bh = wsl_u4{b0[0], b0[1], b1[0], b1[1]}, bl = wsl_u4{b2[0], b2[1], b3[0], b3[1]};
    };
    if constexpr (CB == 32 && TW == 32) {
      // a K-step is one tile row: walk the INPUT rows r = 0 .. TH + 1 of the halo image; the three column-shifted operands of row r
      // serve output rows r, r - 1, r - 2 (taps ky = 0, 1, 2), whose dy operands stay in registers for three input rows --
      // 12 + 4 transpose reads per 27 MFMAs instead of 36 + 4
      wsl_u4 ah[3], al[3];   // dy operands of output rows r, r - 1, r - 2 (slot y % 3)
#pragma unroll
      for (int r = 0; r < TH + 2; ++r) {
        if (r < TH) {
          read_a(r, ah[r % 3], al[r % 3]);
          if (want_db) {
            accdb = WSL_MFMA_F16(al[r % 3], ones, accdb);
            accdb = WSL_MFMA_F16(ah[r % 3], ones, accdb);
          }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          wsl_u4 bh, bl;
          read_b(r * II::ROWB, kx, bh, bl);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int y = r - ky;
            if (y >= 0 && y < TH) {
              acc[ky * 3 + kx] = WSL_MFMA_F16(ah[y % 3], bl, acc[ky * 3 + kx]);
              acc[ky * 3 + kx] = WSL_MFMA_F16(al[y % 3], bh, acc[ky * 3 + kx]);
              acc[ky * 3 + kx] = WSL_MFMA_F16(ah[y % 3], bh, acc[ky * 3 + kx]);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        if (CB == 16 && (ks & 3) != wave) continue;
        wsl_u4 ah, al;
        read_a(ks, ah, al);
        if (want_db) {
          accdb = WSL_MFMA_F16(al, ones, accdb);
          accdb = WSL_MFMA_F16(ah, ones, accdb);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          wsl_u4 bh, bl;
          read_b(ks * C::KROWS * II::ROWB + (tap / 3) * II::ROWB, tap % 3, bh, bl);
          acc[tap] = WSL_MFMA_F16(ah, bl, acc[tap]);
          acc[tap] = WSL_MFMA_F16(al, bh, acc[tap]);
          acc[tap] = WSL_MFMA_F16(ah, bh, acc[tap]);
        }
      }
    }
    __syncthreads();
    t = tn;
  }

  // ---- partials: D[row = co][col = ci]; lane holds rows 4 (lane >> 4) + r of column lane & 15
  const float u1 = sp_pow2(-e_dy), u2 = sp_pow2(-e_act);
  const int sidx = CB == 32 ? split : split * 4 + wave;
  const int64_t E = (int64_t)9 * Co * Ci;
  float* pdw = p.part_dw + (int64_t)sidx * E;
  const int co = co0 + cot * 16 + 4 * grp, ci = ci0 + cit * 16 + sup;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 4; ++r) pdw[((int64_t)tap * Co + co + r) * Ci + ci] = (acc[tap][r] * u1) * u2;
  if (want_db && sup == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p.part_db[(int64_t)sidx * Co + co + r] = accdb[r] * u1;
  }
}

struct WgSpPlan {
  int th, tw, cb, nsplit, splits, items, tiles_x, tiles_y, co_blocks, ci_blocks;
  bool ok;
};
static WgSpPlan wgrad_sp_plan(int N, int H, int W, int Ca, int Cb, int Co) {
  WgSpPlan g{};
  const int Ci = Ca + Cb;
  if (Ci <= 0 || Co <= 0 || (Ci % 16) || (Co % 16) || (Cb && (Ca % 16))) return g;
  if (H % 4 == 0 && W % 32 == 0) g.th = 4, g.tw = 32;
  else if (H % 8 == 0 && W % 16 == 0) g.th = 8, g.tw = 16;
  else return g;
  g.cb = (Co % 32 == 0 && Ci % 32 == 0 && (Cb == 0 || Ca % 32 == 0)) ? 32 : 16;
  g.tiles_x = W / g.tw, g.tiles_y = H / g.th, g.items = N * g.tiles_x * g.tiles_y;
  g.co_blocks = Co / g.cb, g.ci_blocks = Ci / g.cb;
  int want = (forced_wgrad_wgs() > 0 ? forced_wgrad_wgs() : 2 * device_cu_count()) / (g.co_blocks * g.ci_blocks);   // two persistent workgroups per CU
  if (want < 1) want = 1;
  g.nsplit = g.items < want ? g.items : want;
  g.splits = g.cb == 32 ? g.nsplit : 4 * g.nsplit;
  g.ok = true;
  return g;
}

template <int TH, int TW, int CB>
static int launch_wgrad_sp(WgradSpP& p, const WgSpPlan& g, void* stream) {
  using C = WgradSpCfg<TH, TW, CB>;
  auto kern = wgrad_sp_kernel<TH, TW, CB>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  dim3 grid(g.co_blocks * g.ci_blocks, g.nsplit);
  const double px = (double)p.N * p.H * p.W;
  void* tok = prof_begin(PF_SP_WGRAD, 2.0 * px * p.Co * p.Ci * 9, 4.0 * px * (p.Co + p.Ci), stream, 2.0 * px * p.Co * p.Ci * 9 * 3);
  WSL_LAUNCH(kern, grid, dim3(kThreads), C::SMEM, stream, p);
  prof_end(tok, stream);
  return check_launch("wgrad_sp_kernel");
}

}  // namespace wsl

using namespace wsl;

extern "C" int wsl_sp_conv2d_ok(const WslSrc* a, const WslSrc* b, const float* y, int64_t y_bs, int N, int H, int W, int Co, int ks) {
  if (!a || ks != 3 || N <= 0 || H <= 0 || W <= 0 || a->C <= 0) return 0;
  const int Cb = (b && b->C > 0) ? b->C : 0;
  if (Cb && (a->C % 16)) return 0;                       // a 16-channel chunk never straddles the two sources
  if (!sp_src_ok(*a) || (Cb && !sp_src_ok(*b))) return 0;
  if (y && (!aligned16(y) || (y_bs & 3))) return 0;
  if ((int64_t)(a->C > Cb ? a->C : Cb) * H * W >= (int64_t(1) << 31) || (int64_t)Co * H * W >= (int64_t(1) << 31)) return 0;
  return sp_plan(N, H, W, a->C + Cb, Co).ok ? 1 : 0;
}

extern "C" size_t wsl_sp_weight_image_bytes(int Co, int Ci) {
  return Co > 0 && Ci > 0 && Ci % 16 == 0 ? (size_t)40 * Ci * Co : 0;
}

extern "C" int wsl_sp_pack_weights(const float* w, void* image, uint32_t* w_amax, int Co, int Ci, int dgrad, void* stream) {
  WSL_REQUIRE(w && image && w_amax && Co > 0 && Ci > 0 && Ci % 16 == 0 && aligned16(image), "sp_pack_weights: bad args (Ci %% 16 == 0)");
  PackTable t;
  t.n = 1;
  // the table entry describes the RAW tensor [e.Co][e.Ci][3][3]; in data-gradient mode the raw tensor is [Ci][Co][3][3]
  t.e[0] = PackEntry{0, dgrad ? Ci : Co, dgrad ? Co : Ci, 9, 0};
  SpPackOffsets io{};
  if (hipMemsetAsync(w_amax, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
    set_error("sp_pack_weights: clearing *w_amax failed");
    return WSL_EHIP;
  }
  WSL_LAUNCH(sp_amax_table_kernel, dim3(16, 1), dim3(kThreads), 0, stream, t, w, w_amax);
  // one image per call: route it through the slot of its direction
  WSL_LAUNCH(sp_pack_table_kernel, dim3(16, 1, 1), dim3(kThreads), 0, stream, t, io, w, static_cast<unsigned char*>(image),
             static_cast<unsigned char*>(image), w_amax, dgrad ? 1 : 0);
  return check_launch("sp_pack_table_kernel");
}

// (wsl_debug.h) what a launch of this layer shape would use: tile, block width, LDS bytes per workgroup, resident workgroups per CU
extern "C" int wsl_debug_sp_conv_residency(int N, int H, int W, int Ci, int Co, int want_bn_epilogue, int* tile_h, int* tile_w, int* co_t,
                                           size_t* lds_bytes, int* per_cu) {
  const SpPlan f = sp_plan(N, H, W, Ci, Co, want_bn_epilogue != 0);
  if (!f.ok) return 1;
  const bool bres = Ci * f.co_t <= 1024;
  size_t smem = 0;
  int minw = 0;
#define WSL_CASE(TH_, TW_, CO_)                                   \
  if (f.th == TH_ && f.tw == TW_ && f.co_t == CO_) smem = ConvSpCfg<TH_, TW_, CO_>::smem(Ci, bres), minw = ConvSpCfg<TH_, TW_, CO_>::MINW;
  WSL_CASE(8, 32, 16) WSL_CASE(8, 32, 32) WSL_CASE(8, 32, 64) WSL_CASE(8, 16, 16) WSL_CASE(8, 16, 32) WSL_CASE(8, 16, 64)
#undef WSL_CASE
  if (!smem) return 1;
  if (tile_h) *tile_h = f.th;
  if (tile_w) *tile_w = f.tw;
  if (co_t) *co_t = f.co_t;
  if (lds_bytes) *lds_bytes = smem;
  if (per_cu) *per_cu = sp_resident_per_cu(smem, minw);
  return 0;
}

extern "C" int wsl_sp_conv2d_stat_blocks(int N, int H, int W, int Ci, int Co) {
  const SpPlan f = sp_plan(N, H, W, Ci, Co);
  return f.ok ? N * (H / f.th) * (W / f.tw) : 0;
}

extern "C" int wsl_sp_conv2d_fwd(const WslSrc* a, const WslSrc* b, const void* image, const uint32_t* w_amax, const uint32_t* in_amax,
                                 const float* bias, float* y, int64_t y_bs, int N, int H, int W, int Co, float* stat_part,
                                 float* stat_cnt, void* stream) {
  WSL_REQUIRE(a && image && w_amax && y, "sp_conv2d_fwd: null argument");
  WSL_REQUIRE((stat_part == nullptr) == (stat_cnt == nullptr), "sp_conv2d_fwd: stat_part and stat_cnt come together");
  WSL_REQUIRE(y_bs >= (int64_t)Co * H * W, "sp_conv2d_fwd: y batch stride too small");
  WSL_REQUIRE(wsl_sp_conv2d_ok(a, b, y, y_bs, N, H, W, Co, 3), "sp_conv2d_fwd: layer not eligible (wsl_sp_conv2d_ok)");
  const int is_dgrad = in_amax != nullptr && !a->scale && !(b && b->C > 0);   // (profiling family only: a plain single source = a gradient)
  return sp_conv_launch(*a, b, image, w_amax, in_amax, bias, y, y_bs, N, H, W, Co, is_dgrad, stat_part, stat_cnt, nullptr,
                        nullptr, stream);
}

extern "C" int wsl_sp_conv2d_dgrad_bn(const WslSrc* dy, const uint32_t* dy_amax, const void* image, const uint32_t* w_amax, float* g,
                                      int64_t g_bs, int N, int H, int W, int Co, const float* bn_y, const float* bn_st,
                                      const uint8_t* bn_emask, float bn_emask_scale, float* bn_part, int* fused, void* stream) {
  WSL_REQUIRE(dy && dy_amax && image && w_amax && g && bn_y && bn_st && bn_part && fused, "sp_conv2d_dgrad_bn: null argument");
  WSL_REQUIRE(wsl_sp_conv2d_ok(dy, nullptr, g, g_bs, N, H, W, Co, 3), "sp_conv2d_dgrad_bn: layer not eligible (wsl_sp_conv2d_ok)");
  BnBwdEpi e;
  e.y = bn_y, e.st = bn_st, e.emask = bn_emask, e.es = bn_emask_scale, e.part = bn_part;
  if (g_bs != (int64_t)Co * H * W || !aligned16(bn_y) || (bn_emask && (reinterpret_cast<uintptr_t>(bn_emask) & 3))) e.part = nullptr;
  return sp_conv_launch(*dy, nullptr, image, w_amax, dy_amax, nullptr, g, g_bs, N, H, W, Co, 1, nullptr, nullptr, &e, fused, stream);
}

extern "C" size_t wsl_sp_conv2d_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const WgSpPlan g = wgrad_sp_plan(N, H, W, Ci, 0, Co);
  if (!g.ok) return 0;
  // upper bound over both channel blockings (how Ci splits over two sources may lower the block to 16): 4 partials per
  // persistent workgroup, at most two workgroups per CU
  const int want = forced_wgrad_wgs() > 0 ? forced_wgrad_wgs() : 2 * device_cu_count();
  const size_t splits = (size_t)4 * (g.items < want ? g.items : want);
  return sizeof(float) * splits * ((size_t)9 * Co * Ci + Co);
}

extern "C" int wsl_sp_conv2d_wgrad_partial(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, const uint32_t* dy_amax,
                                           float* dw, float* db, int N, int H, int W, int Co, void* ws, size_t ws_bytes,
                                           WslWgradPending* pending, void* stream) {
  return wsl_sp_conv2d_wgrad_partial_amax(a, b, dy, dy_bs, dy_amax, nullptr, dw, db, N, H, W, Co, ws, ws_bytes, pending, stream);
}

extern "C" int wsl_sp_conv2d_wgrad_partial_amax(const WslSrc* a, const WslSrc* b, const float* dy, int64_t dy_bs, const uint32_t* dy_amax,
                                                const uint32_t* in_amax, float* dw, float* db, int N, int H, int W, int Co, void* ws,
                                                size_t ws_bytes, WslWgradPending* pending, void* stream) {
  WSL_REQUIRE(a && dy && dy_amax && dw && ws && pending, "sp_conv2d_wgrad_partial: null argument");
  WSL_REQUIRE(wsl_sp_conv2d_ok(a, b, nullptr, 0, N, H, W, Co, 3) && aligned16(dy) && !(dy_bs & 3) && dy_bs >= (int64_t)Co * H * W,
              "sp_conv2d_wgrad_partial: layer not eligible (wsl_sp_conv2d_ok, float4-aligned dy)");
  const int Cb = (b && b->C > 0) ? b->C : 0, Ci = a->C + Cb;
  const WgSpPlan g = wgrad_sp_plan(N, H, W, a->C, Cb, Co);
  WSL_REQUIRE(g.ok, "sp_conv2d_wgrad_partial: no tile shape for %d x %d", H, W);
  const size_t need = sizeof(float) * (size_t)g.splits * ((size_t)9 * Co * Ci + Co);
  if (ws_bytes < need) {
    set_error("sp_conv2d_wgrad_partial: workspace %zu < %zu", ws_bytes, need);
    return WSL_EWORKSPACE;
  }
  WgradSpP p;
  p.a = to_spsrc(*a);
  p.b = Cb ? to_spsrc(*b) : SpSrc{};
  p.dy = dy, p.dy_bs = dy_bs, p.dy_amax = dy_amax, p.in_amax = in_amax;
  p.part_dw = static_cast<float*>(ws);
  p.part_db = db ? p.part_dw + (size_t)g.splits * 9 * Co * Ci : nullptr;
  p.N = N, p.H = H, p.W = W, p.Ci = Ci, p.Co = Co;
  p.tiles_x = g.tiles_x, p.tiles_y = g.tiles_y, p.items = g.items, p.nsplit = g.nsplit, p.ci_blocks = g.ci_blocks;
  int rc = WSL_EUNSUPPORTED;
  if (g.th == 4 && g.tw == 32 && g.cb == 32) rc = launch_wgrad_sp<4, 32, 32>(p, g, stream);
  else if (g.th == 4 && g.tw == 32 && g.cb == 16) rc = launch_wgrad_sp<4, 32, 16>(p, g, stream);
  else if (g.th == 8 && g.tw == 16 && g.cb == 32) rc = launch_wgrad_sp<8, 16, 32>(p, g, stream);
  else if (g.th == 8 && g.tw == 16 && g.cb == 16) rc = launch_wgrad_sp<8, 16, 16>(p, g, stream);
  if (rc) return rc;
  pending->part_dw = p.part_dw, pending->part_db = p.part_db, pending->dw = dw, pending->db = db;
  pending->Co = Co, pending->Ci = Ci, pending->KK = 9, pending->nsplit = g.splits;
  return WSL_OK;
}

