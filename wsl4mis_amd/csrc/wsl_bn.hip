// BatchNorm statistics / backward, 2x2 max-pool, bilinear x2 and the gradient fan-in of an encoder feature map
// (ref: networks/unet.py:20-25 BatchNorm2d+LeakyReLU+Dropout, :38 MaxPool2d(2), :56-57 Upsample, :63-68 cat).
// All of these are HBM-bound scans; reductions are two-stage with a fixed merge order (no atomics).
#include <stdlib.h>

#include "wsl_rt.h"

namespace wsl {

constexpr int kChunk = 4096;  // elements of one (n, c) plane handled by one workgroup

// ------------------------------------------------------------------------------------------------ BN forward stats
// One workgroup per channel: Chan-merge the conv epilogue's per-block (sum, M2, count) partials in fp64.
// The kernel is a chain of memory round trips (16 .. 256 workgroups, each walking up to 16 384 partials of ITS channel): 4 or 16 waves per
// workgroup (kFinalizeWide) -- with 16 the full-resolution layers need two batches of eight loads per thread instead of eight.
constexpr int kFinalizeWideFrom = 2048;   // partials per channel from which the 16-wave form is launched
static inline int finalize_threads(int nblk) { return nblk >= kFinalizeWideFrom ? 1024 : kThreads; }
// fixed-order merge of per-wave values through LDS: adjacent pairs, level by level ((w0 + w1) + (w2 + w3) for four waves); every thread
// gets the total.  red: >= 16 doubles per value.
template <int NV>
__device__ __forceinline__ void finalize_merge_waves(double (&v)[NV], double* red) {
  const int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] += __shfl_xor(v[k], m);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k * 16 + w] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = i < nw ? red[k * 16 + i] : 0.0;
#pragma unroll
    for (int span = 1; span < 16; span <<= 1)
#pragma unroll
      for (int i = 0; i < 16; i += 2 * span) t[i] = t[i] + t[i + span];   // (absent waves add an exact 0.0)
    v[k] = t[0];
  }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* part, const float* cnt, int nblk, int C,
                                                           const float* gamma, const float* beta, float eps,
                                                           float momentum, float* rmean, float* rvar, int64_t* nbt,
                                                           float* mean_o, float* invstd_o, float* scale_o,
                                                           float* shift_o) {
  __shared__ double red[3 * 16];
  const int c = blockIdx.x;
  // ONE pass: n = sum of counts, s = sum of block sums, q = sum of (M2_b + sum_b^2 / n_b); then M2 = q - s^2 / n, the same
  // Chan merge sum of [M2_b + n_b (mean_b - mean)^2] with the square expanded -- in fp64 the cancellation costs nothing here
  // (q / M2 = 1 + mean^2 / var).  (Loads are unconditional and the empty-slot test is a select: a branch on cnt[b]
  // serialised one memory round trip per iteration.)
  double acc[3] = {0, 0, 0};   // n, s, q
  const float* pc = part + (int64_t)c * nblk * 2;
#pragma unroll 8
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    const float cb = cnt[b];
    const float2 v = *reinterpret_cast<const float2*>(pc + 2 * (int64_t)b);
    const bool live = cb > 0.f;   // empty slots (count 0) carry no data
    const double nbk = live ? (double)cb : 1.0;
    acc[0] += live ? (double)cb : 0.0;
    acc[1] += live ? (double)v.x : 0.0;
    acc[2] += live ? (double)v.y + (double)v.x * (double)v.x / nbk : 0.0;
  }
  finalize_merge_waves<3>(acc, red);
  const double n = acc[0], s = acc[1], q = acc[2];
  const double mean = s / n;
  double m2 = q - s * s / n;
  if (m2 < 0.0) m2 = 0.0;
  if (threadIdx.x == 0) {
    const double var = m2 / n;  // biased (normalisation)
    const float meanf = (float)mean;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_o[c] = meanf;
    invstd_o[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale_o[c] = sc;
    shift_o[c] = fmaf(-meanf, sc, beta[c]);
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * meanf;
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(n > 1 ? m2 / (n - 1) : var);
    if (nbt && c == 0) nbt[0] += 1;
  }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float invstd = 1.f / sqrtf(rvar[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = fmaf(-rmean[c], sc, beta[c]);
  }
}

// ------------------------------------------------------------------------------------------------ materialise / pool
__global__ __launch_bounds__(256) void src_materialize_kernel(WslSrc s, float* out, int64_t out_bs, int HW) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int base = blockIdx.x * kChunk;
  for (int i = base + threadIdx.x; i < base + kChunk && i < HW; i += kThreads)
    out[n * out_bs + (int64_t)c * HW + i] = src_value(s, n, c, n * s.bs + (int64_t)c * HW + i, ((int64_t)n * s.C + c) * HW + i);
}

__global__ __launch_bounds__(256) void pool2_fwd_kernel(WslSrc s, float* out, int H, int W) {
  const int c = blockIdx.y, n = blockIdx.z, Ho = H / 2, Wo = W / 2;
  const int64_t HW = (int64_t)H * W;
  const int base = blockIdx.x * kChunk;
  // plain BatchNorm + LeakyReLU source with even, 8-byte aligned rows (the encoder features): a window row is one float2
  const bool vec = s.scale && !s.emask && !s.cmask && !(W & 1) && !(HW & 1) && !(s.bs & 1) && !(reinterpret_cast<uintptr_t>(s.x) & 7);
  const float sc = vec ? s.scale[c] : 0.f, sh = vec ? s.shift[c] : 0.f;
  for (int o = base + threadIdx.x; o < base + kChunk && o < Ho * Wo; o += kThreads) {
    const int oy = o / Wo, ox = o - oy * Wo;
    float best = 0.f;
    if (vec) {
      const float* p0 = s.x + n * s.bs + c * HW + (int64_t)(2 * oy) * W + 2 * ox;
      const float2 t = *reinterpret_cast<const float2*>(p0), b = *reinterpret_cast<const float2*>(p0 + W);
      const float v[4] = {leaky(fmaf(t.x, sc, sh)), leaky(fmaf(t.y, sc, sh)), leaky(fmaf(b.x, sc, sh)), leaky(fmaf(b.y, sc, sh))};
      best = v[0];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k] > best) best = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t off = (int64_t)(2 * oy + (k >> 1)) * W + 2 * ox + (k & 1);
        const float v = src_value(s, n, c, n * s.bs + c * HW + off, ((int64_t)n * s.C + c) * HW + off);
        if (k == 0 || v > best) best = v;
      }
    }
    out[((int64_t)n * s.C + c) * Ho * Wo + o] = best;
  }
}

// One thread per 2x2 cell of the full-resolution map (cells on an odd border are partial and carry no pool term).
__global__ __launch_bounds__(256) void feat_grad_combine_kernel(WslSrc f, const float* ga, int64_t ga_bs,
                                                                const float* gb, int64_t gb_bs, const float* gb_cmask,
                                                                const float* gp, float* g, int H, int W) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
  const int64_t HW = (int64_t)H * W;
  const float cm = gb_cmask ? gb_cmask[(int64_t)n * f.C + c] : 1.f;
  const int base = blockIdx.x * kChunk;
  for (int o = base + threadIdx.x; o < base + kChunk && o < Hc * Wc; o += kThreads) {
    const int cy = o / Wc, cx = o - cy * Wc;
    int arg = -1;
    float gpool = 0.f;
    if (gp && cy < Ho && cx < Wo) {
      float best = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t off = (int64_t)(2 * cy + (k >> 1)) * W + 2 * cx + (k & 1);
        const float v = src_value(f, n, c, n * f.bs + c * HW + off, ((int64_t)n * f.C + c) * HW + off);
        if (k == 0 || v > best) best = v, arg = k;
      }
      gpool = gp[((int64_t)n * f.C + c) * Ho * Wo + (int64_t)cy * Wo + cx];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 2 * cy + (k >> 1), x = 2 * cx + (k & 1);
      if (y < H && x < W) {
        const int64_t off = (int64_t)y * W + x;
        float v = 0.f;
        if (ga) v = ga[n * ga_bs + c * HW + off];
        if (gb) v = fmaf(gb[n * gb_bs + c * HW + off], cm, v);
        if (k == arg) v += gpool;
        g[((int64_t)n * f.C + c) * HW + off] = v;
      }
    }
  }
}

// feat_grad_combine_kernel that ALSO emits the BatchNorm + LeakyReLU backward statistics of the feature it produces the
// gradient of (the second BatchNorm of an encoder block: no dropout after it): sum(dz), sum(dz * xhat) per workgroup =
// (chunk, channel, sample), layout part[(n * chunks + chunk) * C + c][2] -- what bnact_bwd_reduce_kernel would write, without
// its 8 bytes per element of traffic (the raw feature is read here anyway for the max-pool routing).
__global__ __launch_bounds__(256) void feat_grad_combine_bn_kernel(WslSrc f, const float* ga, int64_t ga_bs, const float* gb,
                                                                   int64_t gb_bs, const float* gb_cmask, const float* gp,
                                                                   float* g, int H, int W, const float* bn_mean,
                                                                   const float* bn_invstd, float* bn_part) {
  __shared__ float red[8];
  const int c = blockIdx.y, n = blockIdx.z;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
  const int64_t HW = (int64_t)H * W;
  const float cm = gb_cmask ? gb_cmask[(int64_t)n * f.C + c] : 1.f;
  const float sc = f.scale[c], sh = f.shift[c], mean = bn_mean[c], invstd = bn_invstd[c];
  const float* fy = f.x + n * f.bs + c * HW;
  float s1 = 0.f, s2 = 0.f;
  const int base = blockIdx.x * kChunk;
  // even H, W and 8-byte aligned planes (every level of the networks): a cell's two pixels of a row travel as one float2
  const bool vec = !(H & 1) && !(W & 1) && !(HW & 1) && !(f.bs & 1) && !(ga_bs & 1) && !(gb_bs & 1) &&
                   !(reinterpret_cast<uintptr_t>(f.x) & 7) && !(reinterpret_cast<uintptr_t>(g) & 7) &&
                   !(reinterpret_cast<uintptr_t>(ga) & 7) && !(reinterpret_cast<uintptr_t>(gb) & 7);
  for (int o = base + threadIdx.x; o < base + kChunk && o < Hc * Wc; o += kThreads) {
    const int cy = o / Wc, cx = o - cy * Wc;
    float yv[4], zv[4];
    bool in[4];
    if (vec) {
      const float2 t = *reinterpret_cast<const float2*>(fy + (int64_t)(2 * cy) * W + 2 * cx);
      const float2 b = *reinterpret_cast<const float2*>(fy + (int64_t)(2 * cy + 1) * W + 2 * cx);
      yv[0] = t.x, yv[1] = t.y, yv[2] = b.x, yv[3] = b.y;
#pragma unroll
      for (int k = 0; k < 4; ++k) in[k] = true, zv[k] = fmaf(yv[k], sc, sh);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = 2 * cy + (k >> 1), x = 2 * cx + (k & 1);
        in[k] = y < H && x < W;
        yv[k] = in[k] ? fy[(int64_t)y * W + x] : 0.f;
        zv[k] = fmaf(yv[k], sc, sh);
      }
    }
    int arg = -1;
    float gpool = 0.f;
    if (gp && cy < Ho && cx < Wo) {
      float best = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = leaky(zv[k]);            // the pooled quantity: leaky(bn(y)), as src_value() computes it
        if (k == 0 || v > best) best = v, arg = k;
      }
      gpool = gp[((int64_t)n * f.C + c) * Ho * Wo + (int64_t)cy * Wo + cx];
    }
    float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      const int64_t o0 = c * HW + (int64_t)(2 * cy) * W + 2 * cx;
      if (ga) {
        const float2 t = *reinterpret_cast<const float2*>(ga + n * ga_bs + o0), b = *reinterpret_cast<const float2*>(ga + n * ga_bs + o0 + W);
        av[0] = t.x, av[1] = t.y, av[2] = b.x, av[3] = b.y;
      }
      if (gb) {
        const float2 t = *reinterpret_cast<const float2*>(gb + n * gb_bs + o0), b = *reinterpret_cast<const float2*>(gb + n * gb_bs + o0 + W);
        bv[0] = t.x, bv[1] = t.y, bv[2] = b.x, bv[3] = b.y;
      }
    }
    float vout[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vout[k] = 0.f;
      if (in[k]) {
        const int64_t off = (int64_t)(2 * cy + (k >> 1)) * W + 2 * cx + (k & 1);
        float v = 0.f;
        if (ga) v = vec ? av[k] : ga[n * ga_bs + c * HW + off];
        if (gb) v = fmaf(vec ? bv[k] : gb[n * gb_bs + c * HW + off], cm, v);
        if (k == arg) v += gpool;
        if (!vec) g[((int64_t)n * f.C + c) * HW + off] = v;
        vout[k] = v;
        const float d = zv[k] > 0.f ? v : WSL_LEAKY_SLOPE * v;
        s1 += d;
        s2 = fmaf(d, (yv[k] - mean) * invstd, s2);
      }
    }
    if (vec) {
      float* go = g + ((int64_t)n * f.C + c) * HW + (int64_t)(2 * cy) * W + 2 * cx;
      *reinterpret_cast<float2*>(go) = make_float2(vout[0], vout[1]);
      *reinterpret_cast<float2*>(go + W) = make_float2(vout[2], vout[3]);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red + 4);
  if (threadIdx.x == 0) {
    float* dst = bn_part + (((int64_t)n * gridDim.x + blockIdx.x) * f.C + c) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
}

// ------------------------------------------------------------------------------------------------ BN + act backward
struct BnBwdP {
  const float* g;
  int64_t g_bs;
  const float* y;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const uint8_t* emask;
  float es;
  int C, HW, chunks;
  int g_is_d = 0;   // g already holds d = g * keep * scale * leaky'(z) (written by wsl_conv2d_dgrad_bn_d): no keep-mask read, no select
};

__device__ __forceinline__ float bn_dz(const BnBwdP& p, int n, int c, int i, float sc, float sh, float mean, float invstd,
                                       float* xhat) {
  const int64_t idx = ((int64_t)n * p.C + c) * p.HW + i;
  const float yv = p.y[idx];
  float d = p.g[n * p.g_bs + (int64_t)c * p.HW + i];
  *xhat = (yv - mean) * invstd;
  if (p.g_is_d) return d;
  if (p.emask) d = p.emask[idx] ? d * p.es : 0.f;
  const float z = fmaf(yv, sc, sh);           // same expression as the forward loader: identical sign decisions
  return z > 0.f ? d : WSL_LEAKY_SLOPE * d;
}

__global__ __launch_bounds__(256) void bnact_bwd_reduce_kernel(BnBwdP p, float* part) {
  __shared__ float red[8];
  const int c = blockIdx.y, n = blockIdx.z;
  const float mean = p.mean[c], invstd = p.invstd[c];
  const float sc = p.gamma[c] * invstd, sh = fmaf(-mean, sc, p.beta[c]);
  float s1 = 0.f, s2 = 0.f;
  const int base = blockIdx.x * kChunk;
  for (int i = base + threadIdx.x; i < base + kChunk && i < p.HW; i += kThreads) {
    float xh;
    const float d = bn_dz(p, n, c, i, sc, sh, mean, invstd, &xh);
    s1 += d;
    s2 = fmaf(d, xh, s2);
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red + 4);
  if (threadIdx.x == 0) {
    float* dst = part + (((int64_t)n * p.chunks + blockIdx.x) * p.C + c) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
}

// float4 forms of the two passes (HW % 4 == 0, 16-byte aligned tensors: every layer of the network): four elements per
// load instruction instead of one.  Same per-element arithmetic as bn_dz().
__device__ __forceinline__ void bn_dz4(const BnBwdP& p, int n, int c, int i, float sc, float sh, float mean, float invstd,
                                       float (&d)[4], float (&xh)[4]) {
  const int64_t idx = ((int64_t)n * p.C + c) * p.HW + i;
  const float4 yv = *reinterpret_cast<const float4*>(p.y + idx);
  const float4 gv = *reinterpret_cast<const float4*>(p.g + n * p.g_bs + (int64_t)c * p.HW + i);
  const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
  float g4[4] = {gv.x, gv.y, gv.z, gv.w};
  if (p.g_is_d) {   // (kernel argument: uniform)
#pragma unroll
    for (int k = 0; k < 4; ++k) xh[k] = (y4[k] - mean) * invstd, d[k] = g4[k];
    return;
  }
  if (p.emask) {
    const uint32_t m = *reinterpret_cast<const uint32_t*>(p.emask + idx);
#pragma unroll
    for (int k = 0; k < 4; ++k) g4[k] = ((m >> (8 * k)) & 0xffu) ? g4[k] * p.es : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float z = fmaf(y4[k], sc, sh);
    xh[k] = (y4[k] - mean) * invstd;
    d[k] = z > 0.f ? g4[k] : WSL_LEAKY_SLOPE * g4[k];
  }
}

__global__ __launch_bounds__(256) void bnact_bwd_reduce4_kernel(BnBwdP p, float* part) {
  __shared__ float red[8];
  const int c = blockIdx.y, n = blockIdx.z;
  const float mean = p.mean[c], invstd = p.invstd[c];
  const float sc = p.gamma[c] * invstd, sh = fmaf(-mean, sc, p.beta[c]);
  float s1 = 0.f, s2 = 0.f;
  const int base = blockIdx.x * kChunk;
  for (int i = base + 4 * threadIdx.x; i < base + kChunk && i < p.HW; i += 4 * kThreads) {
    float d[4], xh[4];
    bn_dz4(p, n, c, i, sc, sh, mean, invstd, d, xh);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s1 += d[k];
      s2 = fmaf(d[k], xh[k], s2);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red + 4);
  if (threadIdx.x == 0) {
    float* dst = part + (((int64_t)n * p.chunks + blockIdx.x) * p.C + c) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
}

// max |dy| for the split-precision consumers of dy (wsl_convsp.hip scales its f16 operands from it): every workgroup of the apply
// pass leaves ONE partial maximum (bit pattern of a non-negative float: orders like the float) in a scratch array, and a 64-block
// fold kernel reduces them into the tensor's WSL_SP_AMAX_SLOTS slots.  No atomics: tens of thousands of short-lived workgroups
// raising 64 words with atomicMax cost 70 us per launch (bnact_bwd_apply4: 126 us against 56 without the maximum, 19 % of the
// split step -- profiles/r3_rocprofv3_kernel_stats_split_asm_chains_atomic_amax.csv), with or without a pre-check load of the slot.
__device__ __forceinline__ void amax_block_store(float m, uint32_t* pmax) {
  __shared__ float mred[4];
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(mred[0], mred[1]), fmaxf(mred[2], mred[3]));
    uint32_t u;
    memcpy(&u, &m, 4);
    pmax[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = u;
  }
}
// grid WSL_SP_AMAX_SLOTS: block b folds partials b, b + 64, ... into slots[b] (a plain store: the slots need no clearing)
__global__ __launch_bounds__(256) void amax_fold_kernel(const uint32_t* pmax, int n, uint32_t* slots) {
  __shared__ uint32_t ured[4];
  uint32_t u = 0u;
  for (int i = blockIdx.x + WSL_SP_AMAX_SLOTS * threadIdx.x; i < n; i += WSL_SP_AMAX_SLOTS * kThreads) {
    const uint32_t v = pmax[i];
    u = v > u ? v : u;
  }
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)u, k);
    u = o > u ? o : u;
  }
  if ((threadIdx.x & 63) == 0) ured[threadIdx.x >> 6] = u;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = ured[0] > ured[1] ? ured[0] : ured[1], b = ured[2] > ured[3] ? ured[2] : ured[3];
    slots[blockIdx.x] = a > b ? a : b;
  }
}

__global__ __launch_bounds__(256) void bnact_bwd_apply4_kernel(BnBwdP p, const float* coef, float* dy, uint32_t* pmax) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float mean = p.mean[c], invstd = p.invstd[c];
  const float sc = p.gamma[c] * invstd, sh = fmaf(-mean, sc, p.beta[c]);
  const float c1 = coef[2 * c], c2 = coef[2 * c + 1];
  const int base = blockIdx.x * kChunk;
  float m = 0.f;
  for (int i = base + 4 * threadIdx.x; i < base + kChunk && i < p.HW; i += 4 * kThreads) {
    float d[4], xh[4];
    bn_dz4(p, n, c, i, sc, sh, mean, invstd, d, xh);
    const float4 v = make_float4(sc * (d[0] - c1 - xh[0] * c2), sc * (d[1] - c1 - xh[1] * c2), sc * (d[2] - c1 - xh[2] * c2),
                                 sc * (d[3] - c1 - xh[3] * c2));
    *reinterpret_cast<float4*>(dy + ((int64_t)n * p.C + c) * p.HW + i) = v;
    if (pmax) m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (pmax) amax_block_store(m, pmax);   // (pmax is a kernel argument: uniform)
}

// part: [nblk][C][2] (sb = C, sc = 1: the stand-alone reduction pass and the fan-in kernel) or [C][nblk][2] (sb = 1, sc = nblk:
// the convolution epilogues)
__global__ __launch_bounds__(1024) void bnact_bwd_finalize_kernel(const float* part, int nblk, int C, int64_t sb, int64_t sc,
                                                                  double count, float* dgamma, float* dbeta, float* coef) {
  __shared__ double red[2 * 16];
  const int c = blockIdx.x;
  double acc[2] = {0, 0};
#pragma unroll 8
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((int64_t)b * sb + (int64_t)c * sc) * 2);
    acc[0] += v.x;
    acc[1] += v.y;
  }
  finalize_merge_waves<2>(acc, red);   // (4 or 16 waves: bn_finalize_kernel)
  const double s1 = acc[0], s2 = acc[1];
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    coef[2 * c] = (float)(s1 / count);
    coef[2 * c + 1] = (float)(s2 / count);
  }
}

__global__ __launch_bounds__(256) void bnact_bwd_apply_kernel(BnBwdP p, const float* coef, float* dy, uint32_t* pmax) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float mean = p.mean[c], invstd = p.invstd[c];
  const float sc = p.gamma[c] * invstd, sh = fmaf(-mean, sc, p.beta[c]);
  const float c1 = coef[2 * c], c2 = coef[2 * c + 1];
  const int base = blockIdx.x * kChunk;
  float m = 0.f;
  for (int i = base + threadIdx.x; i < base + kChunk && i < p.HW; i += kThreads) {
    float xh;
    const float d = bn_dz(p, n, c, i, sc, sh, mean, invstd, &xh);
    const float v = sc * (d - c1 - xh * c2);
    dy[((int64_t)n * p.C + c) * p.HW + i] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (pmax) amax_block_store(m, pmax);   // (pmax is a kernel argument: uniform)
}

// ------------------------------------------------------------------------------------------------ bilinear x2
// src = dst * (in-1)/(out-1) (align_corners=True), i0 = floor, i1 = min(i0+1, in-1), lambda = src - i0.
__device__ __forceinline__ void lerp_coord(int dst, float scale, int in, int* i0, int* i1, float* l) {
  const float src = scale * (float)dst;
  int a = (int)src;
  if (a > in - 1) a = in - 1;
  *i0 = a;
  *i1 = a + (a < in - 1 ? 1 : 0);
  *l = src - (float)a;
}

__global__ __launch_bounds__(256) void bilinear_up2_fwd_kernel(const float* u, float* out, int64_t out_bs, int C, int h,
                                                               int w) {
  const int c = blockIdx.y, n = blockIdx.z, Ho = 2 * h, Wo = 2 * w;
  const float sy = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
  const float* src = u + ((int64_t)n * C + c) * h * w;
  const int base = blockIdx.x * kChunk;
  for (int o = base + threadIdx.x; o < base + kChunk && o < Ho * Wo; o += kThreads) {
    const int oy = o / Wo, ox = o - oy * Wo;
    int y0, y1, x0, x1;
    float ly, lx;
    lerp_coord(oy, sy, h, &y0, &y1, &ly);
    lerp_coord(ox, sx, w, &x0, &x1, &lx);
    const float top = (1.f - lx) * src[y0 * w + x0] + lx * src[y0 * w + x1];
    const float bot = (1.f - lx) * src[y1 * w + x0] + lx * src[y1 * w + x1];
    out[n * out_bs + (int64_t)c * Ho * Wo + o] = (1.f - ly) * top + ly * bot;
  }
}

// Transposed gather: input pixel (i, j) collects from the <= 5x5 output pixels whose stencil touches it.
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const float* dout, int64_t dout_bs, float* du, int C,
                                                               int h, int w) {
  const int c = blockIdx.y, n = blockIdx.z, Ho = 2 * h, Wo = 2 * w;
  const float sy = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
  const float* g = dout + n * dout_bs + (int64_t)c * Ho * Wo;
  const int base = blockIdx.x * kChunk;
  for (int e = base + threadIdx.x; e < base + kChunk && e < h * w; e += kThreads) {
    const int i = e / w, j = e - i * w;
    float wy[5], wx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int Y = 2 * i - 2 + k, X = 2 * j - 2 + k;
      wy[k] = 0.f, wx[k] = 0.f;
      if (Y >= 0 && Y < Ho) {
        int a, b;
        float l;
        lerp_coord(Y, sy, h, &a, &b, &l);
        wy[k] = (a == i ? 1.f - l : 0.f) + (b == i ? l : 0.f);
      }
      if (X >= 0 && X < Wo) {
        int a, b;
        float l;
        lerp_coord(X, sx, w, &a, &b, &l);
        wx[k] = (a == j ? 1.f - l : 0.f) + (b == j ? l : 0.f);
      }
    }
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      if (wy[ky] != 0.f) {
        const int Y = 2 * i - 2 + ky;
        float row = 0.f;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
          if (wx[kx] != 0.f) row = fmaf(wx[kx], g[(int64_t)Y * Wo + 2 * j - 2 + kx], row);
        acc = fmaf(wy[ky], row, acc);
      }
    }
    du[((int64_t)n * C + c) * h * w + e] = acc;
  }
}

// ---- fast variants for power-of-two widths 16..128 and 16-byte aligned planes (every level of the network): a fixed
// output column per thread with its coordinates / weights hoisted out of the row loop (forward); an LDS tile of the
// output gradient filled with aligned float4 loads (transpose).  Same expressions, same summation order.
__global__ __launch_bounds__(256) void bilinear_up2_fwdc_kernel(const float* u, float* out, int64_t out_bs, int C, int h,
                                                                int w, int lw, int rows_per_wg) {
  // one output column per thread (column coordinates / weights computed once), rows strided over the workgroup:
  // adjacent lanes read the same or the next source element, a wave writes 256 contiguous bytes
  const int c = blockIdx.y, n = blockIdx.z, Ho = 2 * h, Wo = 2 * w;
  const float sy = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
  const float* src = u + ((int64_t)n * C + c) * h * w;
  float* dst = out + n * out_bs + (int64_t)c * Ho * Wo;
  const int ox = threadIdx.x & (Wo - 1), r0 = threadIdx.x >> lw, rgroups = kThreads >> lw;
  int x0, x1;
  float lx;
  lerp_coord(ox, sx, w, &x0, &x1, &lx);
  // each row group of the workgroup walks a CONTIGUOUS run of output rows: consecutive output rows share their source rows
  // (scale ~ 1/2), so the horizontally interpolated value of a source row is kept in a register and reused -- about one
  // new source row (2 loads) per two outputs instead of 4 loads per output.  Same expressions, same results.
  const int per = (rows_per_wg + rgroups - 1) / rgroups;
  const int base = blockIdx.x * rows_per_wg + r0 * per;
  int end = base + per;
  if (end > blockIdx.x * rows_per_wg + rows_per_wg) end = blockIdx.x * rows_per_wg + rows_per_wg;
  if (end > Ho) end = Ho;
  int ya = -1, yb = -1;      // source rows whose interpolated values ha / hb are cached
  float ha = 0.f, hb = 0.f;
  for (int oy = base; oy < end; ++oy) {
    int y0, y1;
    float ly;
    lerp_coord(oy, sy, h, &y0, &y1, &ly);
    float top, bot;
    if (y0 == ya) top = ha;
    else if (y0 == yb) top = hb;
    else {
      const float* s0 = src + y0 * w;
      top = (1.f - lx) * s0[x0] + lx * s0[x1];
    }
    if (y1 == y0) bot = top;   // (last row: i1 == i0) -- the same expression on the same operands
    else if (y1 == yb) bot = hb;
    else if (y1 == ya) bot = ha;
    else {
      const float* s1 = src + y1 * w;
      bot = (1.f - lx) * s1[x0] + lx * s1[x1];
    }
    ya = y0, ha = top, yb = y1, hb = bot;
    dst[(int64_t)oy * Wo + ox] = (1.f - ly) * top + ly * bot;
  }
}

// forward, third form: the source rows an output row block touches are staged in LDS with coalesced float4 loads, a thread
// owns FOUR consecutive output columns (their coordinates / weights in registers) and walks rows -> one 16-byte store per
// output quad instead of four 4-byte ones, sources read from LDS.  Same expressions as the kernels above.
constexpr int kUpFR = 32;                    // output rows per workgroup
__global__ __launch_bounds__(256) void bilinear_up2_fwd4_kernel(const float* u, float* out, int64_t out_bs, int C, int h, int w,
                                                                int lq, uint32_t* pmax) {
  __shared__ __attribute__((aligned(16))) float st[(kUpFR / 2 + 3) * 128];
  const int c = blockIdx.y, n = blockIdx.z, Ho = 2 * h, Wo = 2 * w;
  const float sy = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
  const float* src = u + ((int64_t)n * C + c) * h * w;
  float* dst = out + n * out_bs + (int64_t)c * Ho * Wo;
  const int oy0 = blockIdx.x * kUpFR;
  int oy1 = oy0 + kUpFR;
  if (oy1 > Ho) oy1 = Ho;
  int ya, yb, dummy;
  float fdummy;
  lerp_coord(oy0, sy, h, &ya, &dummy, &fdummy);          // first source row of the block
  lerp_coord(oy1 - 1, sy, h, &dummy, &yb, &fdummy);      // last one
  const int nrows = yb - ya + 1, w4 = w >> 2;
  float vmax = 0.f;   // max |source| of the rows this workgroup stages: bounds its outputs (bilinear weights are a convex combination)
  for (int e = threadIdx.x; e < nrows * w4; e += kThreads) {
    const int r = e / w4, q = e - r * w4;
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)(ya + r) * w + 4 * q);
    *reinterpret_cast<float4*>(st + r * w + 4 * q) = v;
    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  const int quads = Wo >> 2;                              // 2^lq column quads per output row
  const int cq = threadIdx.x & (quads - 1), rl = threadIdx.x >> lq, rstep = kThreads >> lq;
  int x0[4], x1[4];
  float lx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lerp_coord(4 * cq + j, sx, w, &x0[j], &x1[j], &lx[j]);
  __syncthreads();
  for (int oy = oy0 + rl; oy < oy1; oy += rstep) {
    int y0, y1;
    float ly;
    lerp_coord(oy, sy, h, &y0, &y1, &ly);
    const float* s0 = st + (y0 - ya) * w;
    const float* s1 = st + (y1 - ya) * w;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float top = (1.f - lx[j]) * s0[x0[j]] + lx[j] * s0[x1[j]];
      const float bot = (1.f - lx[j]) * s1[x0[j]] + lx[j] * s1[x1[j]];
      o[j] = (1.f - ly) * top + ly * bot;
    }
    *reinterpret_cast<float4*>(dst + (int64_t)oy * Wo + 4 * cq) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (pmax) amax_block_store(vmax, pmax);   // (kernel argument: uniform)
}

// max |x| of a plain tensor into WSL_SP_AMAX_SLOTS slots (cleared by the caller): the fallback producer of a raw source's maximum
__global__ __launch_bounds__(256) void amax_tensor_kernel(const float* x, int64_t n4, uint32_t* slots) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  if ((threadIdx.x & 63) == 0) {
    uint32_t u;
    memcpy(&u, &m, 4);
    atomicMax(slots + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (WSL_SP_AMAX_SLOTS - 1)), u);   // integer max: order-independent
  }
}

// Backward of the x2 upsampling, separable form (round 6).  Input pixel (i, j) collects from the 5 x 5 output-gradient pixels around
// (2 i, 2 j):  du[i][j] = sum_ky wy[i][ky] * ( sum_kx wx[j][kx] * g[2 i - 2 + ky][2 j - 2 + kx] ).  A thread owns ONE input column and
// a run of RPT consecutive input rows: it forms the inner (horizontal) sums of the 2 RPT + 3 gradient rows its run touches ONCE -- three
// aligned 8-byte LDS reads and five multiply-adds per row -- keeps them in registers and combines five of them per output.  Rounds 2-5
// evaluated the 25-tap stencil per output (25 four-byte LDS reads at a two-word lane stride = 2-way bank conflicts, 30 multiply-adds and
// ten divergent zero-weight branches per element: VALU-active 0.80, LDS-conflict share 0.48 on an HBM-bound pass -- VERDICT r5 weak 7).
// Same products in the same order (a zero weight now adds an exact zero instead of being skipped), so results are equal up to the sign of
// a zero.  TI = input rows per workgroup: 32 where the plane has that many (row halo 67 / 64 instead of 35 / 32).
constexpr int kUpMaxPitch = 2 * 64 + 8;
template <int TI>
__global__ __launch_bounds__(256) void bilinear_up2_bwd4_kernel(const float* dout, int64_t dout_bs, float* du, int C, int h,
                                                                int w, int ltj) {
  constexpr int ROWS = 2 * TI + 3;                      // output-gradient rows the workgroup's stencils touch
  __shared__ __attribute__((aligned(16))) float tile[ROWS * kUpMaxPitch];
  __shared__ float wyt[TI * 5];   // row weights of the workgroup's input rows
  const int c = blockIdx.y, n = blockIdx.z, Ho = 2 * h, Wo = 2 * w;
  const float sy = Ho > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
  const float* g = dout + n * dout_bs + (int64_t)c * Ho * Wo;
  const int TJ = 1 << ltj, tiles_j = w >> ltj;
  const int j0 = (blockIdx.x % tiles_j) << ltj, i0 = (blockIdx.x / tiles_j) * TI;
  const int pitch = 2 * TJ + 8, nf4 = pitch >> 2;
  const int Y0 = 2 * i0 - 2, X0 = 2 * j0 - 4;   // X0 is a multiple of 4: rows are fetched as aligned float4
  for (int e = threadIdx.x; e < ROWS * nf4; e += kThreads) {
    const int row = e / nf4, q = e - row * nf4;
    const int Y = Y0 + row, X = X0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (Y >= 0 && Y < Ho && X >= 0 && X < Wo) v = *reinterpret_cast<const float4*>(g + (int64_t)Y * Wo + X);
    *reinterpret_cast<float4*>(tile + row * pitch + 4 * q) = v;
  }
  if (threadIdx.x < TI * 5) {
    const int ti = threadIdx.x / 5, ky = threadIdx.x - 5 * ti, i = i0 + ti, Y = 2 * i - 2 + ky;
    float wy = 0.f;
    if (Y >= 0 && Y < Ho) {
      int a, b;
      float l;
      lerp_coord(Y, sy, h, &a, &b, &l);
      wy = (a == i ? 1.f - l : 0.f) + (b == i ? l : 0.f);
    }
    wyt[threadIdx.x] = wy;
  }
  const int tj = threadIdx.x & (TJ - 1), ty = threadIdx.x >> ltj, lrpt = ltj + (TI == 32 ? 5 : 4) - 8;   // rows per thread = TI * TJ / 256
  const int rpt = 1 << lrpt, ti0 = ty << lrpt;
  const int j = j0 + tj;
  float wx[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int X = 2 * j - 2 + k;
    wx[k] = 0.f;
    if (X >= 0 && X < Wo) {
      int a, b;
      float l;
      lerp_coord(X, sx, w, &a, &b, &l);
      wx[k] = (a == j ? 1.f - l : 0.f) + (b == j ? l : 0.f);
    }
  }
  __syncthreads();
  // horizontal sum of gradient row r of the tile at this thread's column: taps at floats 2 tj + 2 .. 2 tj + 6 of the row (an even offset:
  // three aligned 8-byte reads, lanes 8 bytes apart -- conflict-free)
  auto hsum = [&](int r) __attribute__((always_inline)) {
    const float* tr = tile + r * pitch + 2 * tj + 2;
    const float2 a = *reinterpret_cast<const float2*>(tr), b = *reinterpret_cast<const float2*>(tr + 2), cc = *reinterpret_cast<const float2*>(tr + 4);
    float row = 0.f;
    row = fmaf(wx[0], a.x, row), row = fmaf(wx[1], a.y, row), row = fmaf(wx[2], b.x, row), row = fmaf(wx[3], b.y, row);
    return fmaf(wx[4], cc.x, row);
  };
  // sliding window over the gradient rows of the run: rows 2 ti - 2 + {0..4} feed input row ti, the next input row re-uses three of them
  float r0 = hsum(2 * ti0), r1 = hsum(2 * ti0 + 1), r2 = hsum(2 * ti0 + 2);
  for (int k = 0; k < rpt; ++k) {
    const int ti = ti0 + k;
    const float r3 = hsum(2 * ti + 3), r4 = hsum(2 * ti + 4);
    const float* wy = wyt + 5 * ti;
    float acc = 0.f;
    acc = fmaf(wy[0], r0, acc), acc = fmaf(wy[1], r1, acc), acc = fmaf(wy[2], r2, acc), acc = fmaf(wy[3], r3, acc), acc = fmaf(wy[4], r4, acc);
    if (i0 + ti < h) du[((int64_t)n * C + c) * h * w + (int64_t)(i0 + ti) * w + j] = acc;
    r0 = r2, r1 = r3, r2 = r4;
  }
}

static inline bool up2_fast_ok(const void* big, int64_t big_bs, int w) {
  return w >= 16 && w <= 128 && (w & (w - 1)) == 0 && (reinterpret_cast<uintptr_t>(big) & 15) == 0 && (big_bs & 3) == 0;
}
static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace wsl

using namespace wsl;

extern "C" int wsl_bn_stats_finalize(const float* stat_part, const float* stat_cnt, int nblk, int C, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, int64_t* nbt, float* mean, float* invstd, float* scale,
                                     float* shift, void* stream) {
  WSL_REQUIRE(stat_part && stat_cnt && gamma && beta && mean && invstd && scale && shift, "bn_stats_finalize: null");
  WSL_REQUIRE(nblk > 0 && C > 0, "bn_stats_finalize: bad sizes");
  ProfScope ps(PF_BN_FINALIZE, 0.0, 4.0 * (3.0 * nblk * C + 8.0 * C), stream);
  WSL_LAUNCH(bn_finalize_kernel, dim3(C), dim3(finalize_threads(nblk)), 0, stream, stat_part, stat_cnt, nblk, C, gamma, beta, eps,
             momentum, running_mean, running_var, nbt, mean, invstd, scale, shift);
  return check_launch("bn_finalize_kernel");
}

extern "C" int wsl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
  WSL_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_eval_affine: bad args");
  WSL_LAUNCH(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, running_mean, running_var,
             eps, C, scale, shift);
  return check_launch("bn_eval_affine_kernel");
}

extern "C" int wsl_src_materialize(const WslSrc* s, float* out, int64_t out_bs, int N, int H, int W, void* stream) {
  WSL_REQUIRE(s && s->x && out && N > 0 && H > 0 && W > 0 && s->C > 0, "src_materialize: bad args");
  WSL_LAUNCH(src_materialize_kernel, dim3(cdiv(H * W, kChunk), s->C, N), dim3(kThreads), 0, stream, *s, out, out_bs,
             H * W);
  return check_launch("src_materialize_kernel");
}

extern "C" int wsl_pool2_fwd(const WslSrc* s, float* out, int N, int H, int W, void* stream) {
  WSL_REQUIRE(s && s->x && out && N > 0 && H > 1 && W > 1 && s->C > 0, "pool2_fwd: bad args");
  ProfScope ps(PF_POOL_FANIN, 0.0, 5.0 * (double)N * s->C * H * W, stream);        // read 4 B, write 1 B per input element
  WSL_LAUNCH(pool2_fwd_kernel, dim3(cdiv((H / 2) * (W / 2), kChunk), s->C, N), dim3(kThreads), 0, stream, *s, out, H, W);
  return check_launch("pool2_fwd_kernel");
}

extern "C" int wsl_feat_grad_combine(const WslSrc* f, const float* ga, int64_t ga_bs, const float* gb, int64_t gb_bs,
                                     const float* gb_cmask, const float* gp, float* g, int N, int H, int W, void* stream) {
  WSL_REQUIRE(f && g && N > 0 && H > 0 && W > 0 && f->C > 0, "feat_grad_combine: bad args");
  WSL_REQUIRE(!gp || f->x, "feat_grad_combine: pool routing needs the feature map");
  // reads: skip gradient(s) 4 B (+4 B), the feature 4 B and the pooled gradient 1 B when routing; writes 4 B per element
  ProfScope ps(PF_POOL_FANIN, 0.0, (double)N * f->C * H * W * (4.0 * (ga ? 1 : 0) + 4.0 * (gb ? 1 : 0) + (gp ? 5.0 : 0.0) + 4.0), stream);
  WSL_LAUNCH(feat_grad_combine_kernel, dim3(cdiv(((H + 1) / 2) * ((W + 1) / 2), kChunk), f->C, N), dim3(kThreads), 0,
             stream, *f, ga, ga_bs, gb, gb_bs, gb_cmask, gp, g, H, W);
  return check_launch("feat_grad_combine_kernel");
}

extern "C" size_t wsl_bnact_bwd_ws_bytes(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  // partials: one per 4096-element chunk (reduction pass, fan-in kernel) or one per conv tile (>= 256 pixels) when the
  // statistics come from a convolution epilogue; + 2 C coefficients
  return sizeof(float) * ((size_t)N * cdiv(H * W, 256) * C * 2 + 2 * (size_t)C);
}

extern "C" int wsl_bnact_bwd_amax(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                                  float* dgamma, float* dbeta, int N, int C, int H, int W, void* ws, size_t ws_bytes,
                                  uint32_t* dy_amax, void* stream) {
  WSL_REQUIRE(g && y && mean && invstd && gamma && beta && dy && ws, "bnact_bwd: null argument");
  WSL_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "bnact_bwd: bad shape");
  if (ws_bytes < wsl_bnact_bwd_ws_bytes(N, C, H, W)) {
    set_error("bnact_bwd: workspace %zu < %zu", ws_bytes, wsl_bnact_bwd_ws_bytes(N, C, H, W));
    return WSL_EWORKSPACE;
  }
  BnBwdP p{g, g_bs, y, mean, invstd, gamma, beta, emask, emask_scale, C, H * W, cdiv(H * W, kChunk)};
  // reduce pass reads g, y (+ keep mask); apply pass reads them again and writes dy: 20 (+2) B per element
  ProfScope ps(PF_BN_BWD, 0.0, (double)N * C * H * W * (20.0 + (emask ? 2.0 : 0.0)), stream);
  float* part = static_cast<float*>(ws);
  float* coef = part + (size_t)N * p.chunks * C * 2;
  dim3 grid(p.chunks, C, N);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  static const bool vec_on = (WSL_TUNE("WSL_BN_VEC", 1) != 0);
  const bool vec = vec_on && (H * W) % 4 == 0 && (g_bs % 4) == 0 && al16(g) && al16(y) && al16(dy) &&
                   (!emask || (reinterpret_cast<uintptr_t>(emask) & 3) == 0);
  if (vec) WSL_LAUNCH(bnact_bwd_reduce4_kernel, grid, dim3(kThreads), 0, stream, p, part);
  else WSL_LAUNCH(bnact_bwd_reduce_kernel, grid, dim3(kThreads), 0, stream, p, part);
  WSL_LAUNCH(bnact_bwd_finalize_kernel, dim3(C), dim3(finalize_threads(N * p.chunks)), 0, stream, part, N * p.chunks, C, (int64_t)C, (int64_t)1,
             (double)N * H * W, dgamma, dbeta, coef);
  // (the partial sums were consumed by the finalize kernel: their area now takes the apply pass' partial maxima)
  uint32_t* pmax = dy_amax ? reinterpret_cast<uint32_t*>(part) : nullptr;
  if (vec) WSL_LAUNCH(bnact_bwd_apply4_kernel, grid, dim3(kThreads), 0, stream, p, coef, dy, pmax);
  else WSL_LAUNCH(bnact_bwd_apply_kernel, grid, dim3(kThreads), 0, stream, p, coef, dy, pmax);
  if (dy_amax) WSL_LAUNCH(amax_fold_kernel, dim3(WSL_SP_AMAX_SLOTS), dim3(kThreads), 0, stream, pmax, N * p.chunks * C, dy_amax);
  return check_launch("bnact_bwd");
}

extern "C" int wsl_bnact_bwd(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                             float* dgamma, float* dbeta, int N, int C, int H, int W, void* ws, size_t ws_bytes,
                             void* stream) {
  return wsl_bnact_bwd_amax(g, g_bs, y, mean, invstd, gamma, beta, emask, emask_scale, dy, dgamma, dbeta, N, C, H, W, ws, ws_bytes,
                            nullptr, stream);
}

extern "C" size_t wsl_bnact_bwd_finish_ws_bytes(int N, int C, int H, int W, int with_amax) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return sizeof(float) * (2 * (size_t)C + (with_amax ? (size_t)N * cdiv(H * W, kChunk) * C : 0));
}

static int bnact_bwd_finish_impl(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                                 float* dgamma, float* dbeta, int N, int C, int H, int W, const float* part, int nblk,
                                 int channel_major, void* ws, size_t ws_bytes, uint32_t* dy_amax, int g_is_d, void* stream) {
  WSL_REQUIRE(g && y && mean && invstd && gamma && beta && dy && ws && part, "bnact_bwd_finish: null argument");
  WSL_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && nblk > 0, "bnact_bwd_finish: bad shape");
  WSL_REQUIRE(ws_bytes >= wsl_bnact_bwd_finish_ws_bytes(N, C, H, W, dy_amax != nullptr),
              "bnact_bwd_finish: workspace too small (wsl_bnact_bwd_finish_ws_bytes)");
  BnBwdP p{g, g_bs, y, mean, invstd, gamma, beta, emask, emask_scale, C, H * W, cdiv(H * W, kChunk)};
  p.g_is_d = g_is_d;
  ProfScope ps(PF_BN_BWD, 0.0, (double)N * C * H * W * (12.0 + (emask && !g_is_d ? 1.0 : 0.0)), stream);     // the apply pass alone
  float* coef = static_cast<float*>(ws);
  dim3 grid(p.chunks, C, N);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = (H * W) % 4 == 0 && (g_bs % 4) == 0 && al16(g) && al16(y) && al16(dy) &&
                   (!emask || (reinterpret_cast<uintptr_t>(emask) & 3) == 0);
  WSL_LAUNCH(bnact_bwd_finalize_kernel, dim3(C), dim3(finalize_threads(nblk)), 0, stream, part, nblk, C,
             channel_major ? (int64_t)1 : (int64_t)C, channel_major ? (int64_t)nblk : (int64_t)1, (double)N * H * W, dgamma, dbeta,
             coef);
  uint32_t* pmax = dy_amax ? reinterpret_cast<uint32_t*>(coef + 2 * (size_t)C) : nullptr;
  if (vec) WSL_LAUNCH(bnact_bwd_apply4_kernel, grid, dim3(kThreads), 0, stream, p, coef, dy, pmax);
  else WSL_LAUNCH(bnact_bwd_apply_kernel, grid, dim3(kThreads), 0, stream, p, coef, dy, pmax);
  if (dy_amax) WSL_LAUNCH(amax_fold_kernel, dim3(WSL_SP_AMAX_SLOTS), dim3(kThreads), 0, stream, pmax, N * p.chunks * C, dy_amax);
  return check_launch("bnact_bwd_finish");
}

extern "C" int wsl_bnact_bwd_finish_amax(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                                         float* dgamma, float* dbeta, int N, int C, int H, int W, const float* part, int nblk,
                                         int channel_major, void* ws, size_t ws_bytes, uint32_t* dy_amax, void* stream) {
  return bnact_bwd_finish_impl(g, g_bs, y, mean, invstd, gamma, beta, emask, emask_scale, dy, dgamma, dbeta, N, C, H, W, part, nblk,
                               channel_major, ws, ws_bytes, dy_amax, 0, stream);
}

extern "C" int wsl_bnact_bwd_finish_d_amax(const float* d, int64_t d_bs, const float* y, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta, int N, int C,
                                           int H, int W, const float* part, int nblk, int channel_major, void* ws, size_t ws_bytes,
                                           uint32_t* dy_amax, void* stream) {
  return bnact_bwd_finish_impl(d, d_bs, y, mean, invstd, gamma, beta, nullptr, 1.f, dy, dgamma, dbeta, N, C, H, W, part, nblk,
                               channel_major, ws, ws_bytes, dy_amax, 1, stream);
}

extern "C" int wsl_bnact_bwd_finish(const float* g, int64_t g_bs, const float* y, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const uint8_t* emask, float emask_scale, float* dy,
                                    float* dgamma, float* dbeta, int N, int C, int H, int W, const float* part, int nblk,
                                    int channel_major, void* ws, size_t ws_bytes, void* stream) {
  return wsl_bnact_bwd_finish_amax(g, g_bs, y, mean, invstd, gamma, beta, emask, emask_scale, dy, dgamma, dbeta, N, C, H, W, part,
                                   nblk, channel_major, ws, ws_bytes, nullptr, stream);
}

extern "C" int wsl_feat_grad_combine_blocks(int N, int H, int W) {
  return N > 0 && H > 0 && W > 0 ? N * cdiv(((H + 1) / 2) * ((W + 1) / 2), kChunk) : 0;
}

extern "C" int wsl_feat_grad_combine_bn(const WslSrc* f, const float* ga, int64_t ga_bs, const float* gb, int64_t gb_bs,
                                        const float* gb_cmask, const float* gp, float* g, int N, int H, int W,
                                        const float* bn_mean, const float* bn_invstd, float* bn_part, void* stream) {
  WSL_REQUIRE(f && f->x && f->scale && f->shift && !f->emask && !f->cmask && g && N > 0 && H > 0 && W > 0 && f->C > 0,
              "feat_grad_combine_bn: the feature must be a plain BatchNorm + LeakyReLU source");
  WSL_REQUIRE(bn_mean && bn_invstd && bn_part, "feat_grad_combine_bn: null BatchNorm argument");
  ProfScope ps(PF_POOL_FANIN, 0.0, (double)N * f->C * H * W * (4.0 * (ga ? 1 : 0) + 4.0 * (gb ? 1 : 0) + (gp ? 1.0 : 0.0) + 8.0), stream);
  WSL_LAUNCH(feat_grad_combine_bn_kernel, dim3(cdiv(((H + 1) / 2) * ((W + 1) / 2), kChunk), f->C, N), dim3(kThreads), 0,
             stream, *f, ga, ga_bs, gb, gb_bs, gb_cmask, gp, g, H, W, bn_mean, bn_invstd, bn_part);
  return check_launch("feat_grad_combine_bn_kernel");
}

extern "C" size_t wsl_bilinear_up2_fwd_amax_ws_bytes(int N, int C, int h, int w) {
  return N > 0 && C > 0 && h > 0 && w > 0 ? sizeof(uint32_t) * (size_t)N * C * cdiv(2 * h, kUpFR) : 0;
}

extern "C" int wsl_bilinear_up2_fwd_amax(const float* u, float* out, int64_t out_bs, int N, int C, int h, int w, void* ws, size_t ws_bytes,
                                         uint32_t* amax_slots, void* stream) {
  WSL_REQUIRE(u && out && N > 0 && C > 0 && h > 0 && w > 0, "bilinear_up2_fwd: bad args");
  WSL_REQUIRE(out_bs >= (int64_t)C * 4 * h * w, "bilinear_up2_fwd: out batch stride too small");
  WSL_REQUIRE(!amax_slots || (ws && ws_bytes >= wsl_bilinear_up2_fwd_amax_ws_bytes(N, C, h, w)), "bilinear_up2_fwd_amax: workspace too small");
  ProfScope ps(PF_BILINEAR, 0.0, 20.0 * (double)N * C * h * w, stream);            // read 4 B per input, write 4 x 4 B
  if (up2_fast_ok(out, out_bs, w) && (reinterpret_cast<uintptr_t>(u) & 15) == 0) {
    const dim3 grid(cdiv(2 * h, kUpFR), C, N);
    uint32_t* pmax = amax_slots ? static_cast<uint32_t*>(ws) : nullptr;
    WSL_LAUNCH(bilinear_up2_fwd4_kernel, grid, dim3(kThreads), 0, stream, u, out, out_bs, C, h, w, ilog2(2 * w) - 2, pmax);
    if (amax_slots) WSL_LAUNCH(amax_fold_kernel, dim3(WSL_SP_AMAX_SLOTS), dim3(kThreads), 0, stream, pmax, (int)(grid.x * grid.y * grid.z), amax_slots);
    return check_launch("bilinear_up2_fwd4_kernel");
  }
  if (amax_slots) {   // the other forms do not carry the maximum: one pass over the (4x smaller) source
    WSL_REQUIRE(((int64_t)N * C * h * w) % 4 == 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0, "bilinear_up2_fwd_amax: source not float4-aligned");
    if (hipMemsetAsync(amax_slots, 0, WSL_SP_AMAX_SLOTS * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
      set_error("bilinear_up2_fwd_amax: clearing the slots failed");
      return WSL_EHIP;
    }
    WSL_LAUNCH(amax_tensor_kernel, dim3(256), dim3(kThreads), 0, stream, u, (int64_t)N * C * h * w / 4, amax_slots);
  }
  if (up2_fast_ok(out, out_bs, w)) {
    const int lw = ilog2(2 * w);                                    // output row width = 2^lw <= 256
    int rows = 8192 / (2 * w);                                      // ~8K outputs per workgroup
    if (rows < (kThreads >> lw)) rows = kThreads >> lw;
    WSL_LAUNCH(bilinear_up2_fwdc_kernel, dim3(cdiv(2 * h, rows), C, N), dim3(kThreads), 0, stream, u, out, out_bs, C, h, w,
               lw, rows);
    return check_launch("bilinear_up2_fwdc_kernel");
  }
  WSL_LAUNCH(bilinear_up2_fwd_kernel, dim3(cdiv(4 * h * w, kChunk), C, N), dim3(kThreads), 0, stream, u, out, out_bs, C,
             h, w);
  return check_launch("bilinear_up2_fwd_kernel");
}

extern "C" int wsl_bilinear_up2_fwd(const float* u, float* out, int64_t out_bs, int N, int C, int h, int w, void* stream) {
  return wsl_bilinear_up2_fwd_amax(u, out, out_bs, N, C, h, w, nullptr, 0, nullptr, stream);
}

extern "C" int wsl_bilinear_up2_bwd(const float* dout, int64_t dout_bs, float* du, int N, int C, int h, int w,
                                    void* stream) {
  WSL_REQUIRE(dout && du && N > 0 && C > 0 && h > 0 && w > 0, "bilinear_up2_bwd: bad args");
  WSL_REQUIRE(dout_bs >= (int64_t)C * 4 * h * w, "bilinear_up2_bwd: dout batch stride too small");
  ProfScope ps(PF_BILINEAR, 0.0, 20.0 * (double)N * C * h * w, stream);
  if (up2_fast_ok(dout, dout_bs, w)) {
    const int ltj = ilog2(w < 64 ? w : 64);
    if (h % 32 == 0) {
      WSL_LAUNCH(bilinear_up2_bwd4_kernel<32>, dim3((w >> ltj) * (h / 32), C, N), dim3(kThreads), 0, stream, dout, dout_bs, du, C, h, w, ltj);
    } else {
      WSL_LAUNCH(bilinear_up2_bwd4_kernel<16>, dim3((w >> ltj) * cdiv(h, 16), C, N), dim3(kThreads), 0, stream, dout, dout_bs, du, C, h, w, ltj);
    }
    return check_launch("bilinear_up2_bwd4_kernel");
  }
  WSL_LAUNCH(bilinear_up2_bwd_kernel, dim3(cdiv(h * w, kChunk), C, N), dim3(kThreads), 0, stream, dout, dout_bs, du, C,
             h, w);
  return check_launch("bilinear_up2_bwd_kernel");
}
