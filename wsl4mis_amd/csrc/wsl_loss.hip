// Weak-supervision losses as wavefront-reduction / stencil kernels (HBM-bound scans over [N,C,H,W] probabilities):
//   CrossEntropyLoss(ignore_index)      ref: train_weakly_supervised_pCE_2D.py:81,100
//   mixed pseudo labels + pDLoss        ref: train_weakly_supervised_segmentation_pCE_ours_proposed.py:110-125,
//                                            utils/losses.py:195-232 (incl. its [N,H,W]x[N,1,H,W] broadcast)
//   ModelLossSemsegGatedCRF             ref: utils/gate_crf_loss.py:20-124,135-188
//   tv_loss                             ref: train_weakly_supervised_pCE_TV_2D.py:58-65
//   MumfordShah_Loss                    ref: utils/losses.py:275-309
//   softmax_mse_loss                    ref: utils/losses.py:65-82
// Every reduction is two-stage: per-workgroup partials, then a single-workgroup finalize that merges them in a fixed
// order in fp64 and does the scalar arithmetic on the device (no host sync, no float atomics).
#include <math.h>

#include "wsl_rt.h"

namespace wsl {

constexpr int kMaxC = 8;
constexpr int kMaxBlocks = 1024;
constexpr int kMaxK = 48;  // partial values per workgroup

__device__ __forceinline__ double block_sum_d2(double v, double* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  return red[0];
}

// Sum column k of part[nblk][K] over the workgroup (single-workgroup finalize kernels).
__device__ __forceinline__ double col_sum(const float* part, int nblk, int K, int k, double* red) {
  double s = 0;
  for (int b = threadIdx.x; b < nblk; b += kThreads) s += part[(int64_t)b * K + k];
  return block_sum_d2(s, red);
}

template <int K>
__device__ __forceinline__ void write_partials(float (&v)[K], float* part, float* red) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float s = block_sum(v[k], red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * K + k] = s;
  }
}

__device__ __forceinline__ int load_label(const void* lab, int i64, int64_t idx) {
  return i64 ? (int)static_cast<const int64_t*>(lab)[idx] : (int)static_cast<const uint8_t*>(lab)[idx];
}

// softmax over C strided values; returns log-sum-exp (max-subtracted).
__device__ __forceinline__ float softmax_c(const float* z, int64_t stride, int C, float* s) {
  float m = z[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, z[c * stride]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) {
    s[c] = expf(z[c * stride] - m);
    sum += s[c];
  }
  const float inv = 1.f / sum;
  for (int c = 0; c < C; ++c) s[c] *= inv;
  return m + logf(sum);
}

__device__ __forceinline__ int mix_argmax_c(const float* s1, const float* s2, int C, float bf, float omb) {
  int best = 0;
  float bv = __fadd_rn(__fmul_rn(bf, s1[0]), __fmul_rn(omb, s2[0]));
  for (int c = 1; c < C; ++c) {
    const float v = __fadd_rn(__fmul_rn(bf, s1[c]), __fmul_rn(omb, s2[c]));
    if (v > bv) bv = v, best = c;
  }
  return best;
}

// ------------------------------------------------------------------------------------------------ softmax
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* z, float* s, int C, int HW, int64_t P) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float v[kMaxC];
    softmax_c(z + base, HW, C, v);
    for (int c = 0; c < C; ++c) s[base + (int64_t)c * HW] = v[c];
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* s, const float* ds, float* dz, int C, int HW,
                                                          int64_t P) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot = fmaf(ds[base + (int64_t)c * HW], s[base + (int64_t)c * HW], dot);
    for (int c = 0; c < C; ++c) dz[base + (int64_t)c * HW] = s[base + (int64_t)c * HW] * (ds[base + (int64_t)c * HW] - dot);
  }
}

// ------------------------------------------------------------------------------------------------ cross entropy
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* z, const void* lab, int i64, int ignore, int C, int HW,
                                                        int64_t P, float* part) {
  __shared__ float red[4];
  float v[2] = {0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int l = load_label(lab, i64, i);
    if (l != ignore && l >= 0 && l < C) {
      const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
      float s[kMaxC];
      const float lse = softmax_c(z + base, HW, C, s);
      v[0] += lse - z[base + (int64_t)l * HW];
      v[1] += 1.f;
    }
  }
  write_partials<2>(v, part, red);
}

__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* part, int nblk, float* loss, float* scal) {
  __shared__ double red[kThreads];
  const double nll = col_sum(part, nblk, 2, 0, red), cnt = col_sum(part, nblk, 2, 1, red);
  if (threadIdx.x == 0) {
    loss[0] = (float)(nll / cnt);  // 0/0 -> NaN like torch when every pixel is ignored
    scal[0] = cnt > 0 ? (float)(1.0 / cnt) : 0.f;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* z, const void* lab, int i64, int ignore, int C, int HW,
                                                     int64_t P, const float* scal, float gscale, float* dz) {
  const float k = scal[0] * gscale;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    const int l = load_label(lab, i64, i);
    if (l != ignore && l >= 0 && l < C) {
      float s[kMaxC];
      softmax_c(z + base, HW, C, s);
      for (int c = 0; c < C; ++c) dz[base + (int64_t)c * HW] = k * (s[c] - (c == l ? 1.f : 0.f));
    } else {
      for (int c = 0; c < C; ++c) dz[base + (int64_t)c * HW] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ mix + argmax
__global__ __launch_bounds__(256) void mix_argmax_kernel(const float* s1, const float* s2, float bf, float omb,
                                                         int64_t* pseudo, int C, int HW, int64_t P) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float a[kMaxC], b[kMaxC];
    for (int c = 0; c < C; ++c) a[c] = s1[base + (int64_t)c * HW], b[c] = s2[base + (int64_t)c * HW];
    pseudo[i] = mix_argmax_c(a, b, C, bf, omb);
  }
}

// ------------------------------------------------------------------------------------------------ pDLoss / DiceLoss
// One thread per pixel position, looping over the batch: the reference's broadcast makes every sum
//   sum_{h,w} (sum_b term[b,h,w]) * (sum_a mask[a,h,w]).
__global__ __launch_bounds__(256) void pdice_reduce_kernel(const float* s, const void* tgt, int i64, int ignore, int N,
                                                           int C, int HW, float* part) {
  __shared__ float red[4];
  float v[3 * kMaxC];
  for (int k = 0; k < 3 * kMaxC; ++k) v[k] = 0.f;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < HW; p += gridDim.x * kThreads) {
    float I[kMaxC], Z[kMaxC], Y[kMaxC], M = 0.f;
    for (int c = 0; c < C; ++c) I[c] = Z[c] = Y[c] = 0.f;
    for (int n = 0; n < N; ++n) {
      const int l = load_label(tgt, i64, (int64_t)n * HW + p);
      M += (ignore >= 0 && l == ignore) ? 0.f : 1.f;
      for (int c = 0; c < C; ++c) {
        const float sv = s[((int64_t)n * C + c) * HW + p];
        Z[c] = fmaf(sv, sv, Z[c]);
        if (l == c) I[c] += sv, Y[c] += 1.f;
      }
    }
    if (ignore < 0) M = 1.f;
    for (int c = 0; c < C; ++c) v[c] = fmaf(I[c], M, v[c]), v[kMaxC + c] = fmaf(Z[c], M, v[kMaxC + c]),
                               v[2 * kMaxC + c] = fmaf(Y[c], M, v[2 * kMaxC + c]);
  }
  write_partials<3 * kMaxC>(v, part, red);
}

__global__ __launch_bounds__(256) void pdice_finalize_kernel(const float* part, int nblk, int C, float* loss, float* sums) {
  __shared__ double red[kThreads];
  double acc = 0;
  for (int c = 0; c < C; ++c) {
    const double I = col_sum(part, nblk, 3 * kMaxC, c, red), Z = col_sum(part, nblk, 3 * kMaxC, kMaxC + c, red),
                 Y = col_sum(part, nblk, 3 * kMaxC, 2 * kMaxC + c, red);
    const float If = (float)I, Zf = (float)Z, Yf = (float)Y;
    acc += 1.0 - (double)((2.f * If + 1e-5f) / (Zf + Yf + 1e-5f));
    if (threadIdx.x == 0) sums[c] = If, sums[C + c] = Zf, sums[2 * C + c] = Yf;
  }
  if (threadIdx.x == 0) loss[0] = (float)(acc / C);
}

__global__ __launch_bounds__(256) void pdice_bwd_kernel(const float* s, const void* tgt, int i64, int ignore,
                                                        const float* sums, const float* gout, float* ds, int N, int C,
                                                        int HW) {
  const float go = (gout ? gout[0] : 1.f) / (float)C;
  float ca[kMaxC], cb[kMaxC];
  for (int c = 0; c < C; ++c) {
    const float I = sums[c], D = sums[C + c] + sums[2 * C + c] + 1e-5f;
    ca[c] = -2.f / D * go;                              // d/ds of -(2I+eps)/D through I  (times t)
    cb[c] = 2.f * (2.f * I + 1e-5f) / (D * D) * go;     // ... through Z                   (times s)
  }
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < HW; p += gridDim.x * kThreads) {
    float M = 1.f;
    if (ignore >= 0) {
      M = 0.f;
      for (int n = 0; n < N; ++n) M += load_label(tgt, i64, (int64_t)n * HW + p) == ignore ? 0.f : 1.f;
    }
    for (int n = 0; n < N; ++n) {
      const int l = load_label(tgt, i64, (int64_t)n * HW + p);
      for (int c = 0; c < C; ++c) {
        const int64_t idx = ((int64_t)n * C + c) * HW + p;
        ds[idx] = M * (ca[c] * (l == c ? 1.f : 0.f) + cb[c] * s[idx]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused head
// partial columns: 0 nll1, 1 nll2, 2 n_valid, 3+c I1, 3+C+c Z1, 3+2C+c I2, 3+3C+c Z2, 3+4C+c Y
struct HeadP {
  const float* z1;
  const float* z2;
  const uint8_t* label;
  int ignore, C, HW, N;
  int64_t P;
  float bf, omb;
};

// CT > 0: class count fixed at compile time (loops unroll, per-class arrays live in registers); CT == 0: generic
template <int CT>
__global__ __launch_bounds__(256) void head_reduce_kernel(HeadP h, int64_t* pseudo, float* part, float* y_mix) {
  __shared__ float red[4];
  float v[3 + 5 * kMaxC];
  for (int k = 0; k < 3 + 5 * kMaxC; ++k) v[k] = 0.f;
  const int C = CT > 0 ? CT : h.C;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < h.P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / h.HW, p = i - n * h.HW, base = n * C * h.HW + p;
    const int l = h.label[i];
    const bool valid = (l != h.ignore) && l < C;
    float s1[kMaxC], s2[kMaxC];
    const float lse1 = softmax_c(h.z1 + base, h.HW, C, s1);
    if (valid) v[0] += lse1 - h.z1[base + (int64_t)l * h.HW], v[2] += 1.f;
    if (h.z2) {
      const float lse2 = softmax_c(h.z2 + base, h.HW, C, s2);
      if (valid) v[1] += lse2 - h.z2[base + (int64_t)l * h.HW];
      const int t = mix_argmax_c(s1, s2, C, h.bf, h.omb);
      if (pseudo) pseudo[i] = t;
      if (y_mix)   // the mixed prediction the GatedCRF term regularises: same expression (and bits) as mixprob_fwd_kernel
        for (int c = 0; c < C; ++c) y_mix[base + (int64_t)c * h.HW] = __fadd_rn(__fmul_rn(h.bf, s1[c]), __fmul_rn(h.omb, s2[c]));
      for (int c = 0; c < C; ++c) {
        v[3 + kMaxC + c] = fmaf(s1[c], s1[c], v[3 + kMaxC + c]);
        v[3 + 3 * kMaxC + c] = fmaf(s2[c], s2[c], v[3 + 3 * kMaxC + c]);
        if (c == t) v[3 + c] += s1[c], v[3 + 2 * kMaxC + c] += s2[c], v[3 + 4 * kMaxC + c] += 1.f;
      }
    } else if (y_mix) {
      for (int c = 0; c < C; ++c) y_mix[base + (int64_t)c * h.HW] = s1[c];
    }
  }
  write_partials<3 + 5 * kMaxC>(v, part, red);
}

// scal: 0 inv_nvalid; 1+c ca1, 1+C+c cb1, 1+2C+c ca2, 1+3C+c cb2 (already times w_pse*0.5/C)
__global__ __launch_bounds__(256) void head_finalize_kernel(const float* part, int nblk, int C, int N, int dual,
                                                            float w_pse, float* out, float* scal) {
  // all K <= 64 column sums of part[nblk][K] in ONE sweep: thread (quarter w, column k) adds every fourth row in fp64 (consecutive
  // lanes read consecutive floats), the four quarters are merged in a fixed order.  (Round 3: 23 separate block reductions of 17
  // barriers each -- 34 us for this single-workgroup kernel.)
  constexpr int K = 3 + 5 * kMaxC;
  static_assert(K <= 64, "one column per lane");
  __shared__ double red4[4][64];
  __shared__ double tot[64];
  {
    const int w = threadIdx.x >> 6, k = threadIdx.x & 63;
    // (eight rows in flight per thread: one dependent load per iteration made this single-workgroup kernel 64 us of L2 round trips --
    //  round 5; the eight partial sums are merged in a fixed order)
    double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k < K) {
      int b = w;
      for (; b + 28 < nblk; b += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 4 * u) * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] += (double)v[u];
      }
      for (; b < nblk; b += 4) a8[0] += part[(int64_t)b * K + k];
    }
    red4[w][k] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    __syncthreads();
    if (threadIdx.x < 64) tot[threadIdx.x] = (red4[0][threadIdx.x] + red4[1][threadIdx.x]) + (red4[2][threadIdx.x] + red4[3][threadIdx.x]);
    __syncthreads();
  }
  const double nll1 = tot[0], nll2 = tot[1], cnt = tot[2];
  const float ce1 = (float)(nll1 / cnt), ce2 = (float)(nll2 / cnt);
  float pse = 0.f;
  if (dual) {
    float d1 = 0.f, d2 = 0.f;
    const float Nf = (float)N;  // pseudo labels are never `ignore`: the broadcast multiplies every sum by N
    for (int c = 0; c < C; ++c) {
      const float I1 = Nf * (float)tot[3 + c], Z1 = Nf * (float)tot[3 + kMaxC + c], I2 = Nf * (float)tot[3 + 2 * kMaxC + c],
                  Z2 = Nf * (float)tot[3 + 3 * kMaxC + c], Y = Nf * (float)tot[3 + 4 * kMaxC + c];
      const float D1 = Z1 + Y + 1e-5f, D2 = Z2 + Y + 1e-5f;
      d1 += 1.f - (2.f * I1 + 1e-5f) / D1;
      d2 += 1.f - (2.f * I2 + 1e-5f) / D2;
      if (threadIdx.x == 0) {
        const float k = w_pse * 0.5f / (float)C * Nf;
        scal[1 + c] = -2.f / D1 * k;
        scal[1 + C + c] = 2.f * (2.f * I1 + 1e-5f) / (D1 * D1) * k;
        scal[1 + 2 * C + c] = -2.f / D2 * k;
        scal[1 + 3 * C + c] = 2.f * (2.f * I2 + 1e-5f) / (D2 * D2) * k;
      }
    }
    pse = 0.5f * (d1 / (float)C + d2 / (float)C);
  }
  if (threadIdx.x == 0) {
    const float ce = dual ? 0.5f * (ce1 + ce2) : ce1;
    out[0] = dual ? ce + w_pse * pse : ce;
    out[1] = ce;
    out[2] = pse;
    out[3] = (float)cnt;
    scal[0] = cnt > 0 ? (float)(1.0 / cnt) : 0.f;
  }
}

// gy != NULL: + wgt * softmax_bwd(s, gy) -- the gradient through the mixed prediction y (GatedCRF term), what
// mixprob_bwd_kernel would accumulate in a pass of its own (same expressions; the contraction of the final sum may differ)
// (has_dice / has_gy as flags next to array REFERENCES: a pointer that may be null -- `dy_mix ? gy : nullptr` -- makes the per-class
//  arrays address-taken and puts them into scratch memory: 48 bytes per lane in round 3's head_bwd_kernel<4>)
__device__ __forceinline__ void head_branch_bwd(const float (&s)[kMaxC], int C, int t, int l, bool valid, bool has_dice,
                                                const float (&ca)[kMaxC], const float (&cb)[kMaxC], float kce, float gscale, float* dz,
                                                int64_t stride, bool has_gy, const float (&gy)[kMaxC], float wgt) {
  float dsv[kMaxC], dot = 0.f, doty = 0.f;
  for (int c = 0; c < C; ++c) {
    dsv[c] = has_dice ? ca[c] * (c == t ? 1.f : 0.f) + cb[c] * s[c] : 0.f;
    dot = fmaf(dsv[c], s[c], dot);
  }
  if (has_gy)
    for (int c = 0; c < C; ++c) doty = fmaf(gy[c], s[c], doty);
  for (int c = 0; c < C; ++c) {
    float g = s[c] * (dsv[c] - dot);
    if (valid) g += kce * (s[c] - (c == l ? 1.f : 0.f));
    g = g * gscale;
    if (has_gy) g = g + wgt * s[c] * (gy[c] - doty);
    dz[c * stride] = g;
  }
}

template <int CT>
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadP h, const float* scal, float gscale, float* dz1, float* dz2,
                                                       const float* dy_mix, float ky) {
  const int C = CT > 0 ? CT : h.C;
  const bool dual = h.z2 != nullptr;
  const float kce = scal[0] * (dual ? 0.5f : 1.f);
  float ca1[kMaxC], cb1[kMaxC], ca2[kMaxC], cb2[kMaxC];
  for (int c = 0; c < C; ++c) ca1[c] = scal[1 + c], cb1[c] = scal[1 + C + c], ca2[c] = scal[1 + 2 * C + c], cb2[c] = scal[1 + 3 * C + c];
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < h.P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / h.HW, p = i - n * h.HW, base = n * C * h.HW + p;
    const int l = h.label[i];
    const bool valid = (l != h.ignore) && l < C;
    float s1[kMaxC], s2[kMaxC], gy[kMaxC];
    const bool has_gy = dy_mix != nullptr;
    for (int c = 0; c < C; ++c) gy[c] = has_gy ? ky * dy_mix[base + (int64_t)c * h.HW] : 0.f;
    softmax_c(h.z1 + base, h.HW, C, s1);
    if (dual) {
      softmax_c(h.z2 + base, h.HW, C, s2);
      const int t = mix_argmax_c(s1, s2, C, h.bf, h.omb);
      head_branch_bwd(s1, C, t, l, valid, true, ca1, cb1, kce, gscale, dz1 + base, h.HW, has_gy, gy, h.bf);
      head_branch_bwd(s2, C, t, l, valid, true, ca2, cb2, kce, gscale, dz2 + base, h.HW, has_gy, gy, h.omb);
    } else {
      head_branch_bwd(s1, C, 0, l, valid, false, ca1, cb1, kce, gscale, dz1 + base, h.HW, has_gy, gy, 1.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ GatedCRF
// 16x16 pixels per workgroup, y (C planes) and the image staged in LDS with a `r` halo.  A tap outside the image sees
// feature vector 0 and y = 0 (zero-padded unfold): it adds to sum(K) only.
struct CrfP {
  const float* y;
  const float* img;
  float* msg;
  int N, C, H, W, r;
  float sxy, srgb, weight;
  int tiles_x, tiles_y;
};

// 32 x 8 pixels per workgroup: a wave covers two full 32-pixel rows, so its ds_read_b32 of a tap (two groups of 32
// consecutive floats) is bank-conflict free for any row pitch (the 16 x 16 layout measured a 0.48 conflict ratio).
constexpr int kCrfTW = 32, kCrfTH = 8;
// CT > 0: class count known at compile time (per-class accumulators stay in registers, loops unroll); CT == 0: generic
template <int CT>
__global__ __launch_bounds__(256) void gatedcrf_fwd_kernel(CrfP q, float* part) {
  WSL_DYN_SMEM(smem);
  __shared__ float red[4];
  const int r = q.r, TSX = kCrfTW + 2 * r, TSY = kCrfTH + 2 * r, TSS = TSX * TSY;
  const int C = CT > 0 ? CT : q.C;
  constexpr int MC = CT > 0 ? CT : kMaxC;
  float* yt = reinterpret_cast<float*>(smem);   // [C][TSY*TSX]
  float* it = yt + C * TSS;                     // [TSY*TSX] image / sigma_rgb, 0 outside
  float* fxt = it + TSS;                        // [TSX] column feature x / sigma_xy (0 outside the image)
  float* fyt = fxt + TSX;                       // [TSY] row feature
  int bid = blockIdx.x;
  const int tx_i = bid % q.tiles_x;
  bid /= q.tiles_x;
  const int ty_i = bid % q.tiles_y, n = bid / q.tiles_y;
  const int y0 = ty_i * kCrfTH, x0 = tx_i * kCrfTW;
  const int64_t HW = (int64_t)q.H * q.W;
  for (int e = threadIdx.x; e < TSS; e += kThreads) {
    const int ty = e / TSX, tx = e - ty * TSX, gy = y0 + ty - r, gx = x0 + tx - r;
    const bool in = gy >= 0 && gy < q.H && gx >= 0 && gx < q.W;
    it[e] = in ? q.img[n * HW + (int64_t)gy * q.W + gx] / q.srgb : 0.f;
    for (int c = 0; c < C; ++c) yt[c * TSS + e] = in ? q.y[((int64_t)n * C + c) * HW + (int64_t)gy * q.W + gx] : 0.f;
  }
  if ((int)threadIdx.x < TSX) {
    const int gx = x0 + (int)threadIdx.x - r;
    fxt[threadIdx.x] = (gx >= 0 && gx < q.W) ? (float)gx / q.sxy : 0.f;
  }
  if ((int)threadIdx.x < TSY) {
    const int gy = y0 + (int)threadIdx.x - r;
    fyt[threadIdx.x] = (gy >= 0 && gy < q.H) ? (float)gy / q.sxy : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31, gy = y0 + ly, gx = x0 + lx;
  // every tap of every pixel of this tile inside the image?  (uniform; true for ~70 % of the tiles of a 256^2 slice)
  const bool interior = y0 - r >= 0 && y0 + kCrfTH + r <= q.H && x0 - r >= 0 && x0 + kCrfTW + r <= q.W;
  float v[2] = {0.f, 0.f};
  if (gy < q.H && gx < q.W) {
    const int ce = (ly + r) * TSX + lx + r;
    const float fpx = fxt[lx + r], fpy = fyt[ly + r], fpi = it[ce];
    float m[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) m[c] = 0.f;
    float ks = 0.f;
    for (int dy = -r; dy <= r; ++dy) {
      const int qy = gy + dy;
      const bool iny = qy >= 0 && qy < q.H;
      const float rowf = fyt[ly + r + dy];
      for (int dx = -r; dx <= r; ++dx) {
        if (dy == 0 && dx == 0) continue;
        const int te = ce + dy * TSX + dx;
        float fqx = fxt[lx + r + dx], fqy = rowf;
        if (!interior) {
          // a tap outside the image sees the all-zero feature vector of the zero-padded unfold (gate_crf_loss.py:184-188)
          const int qx = gx + dx;
          const bool in = iny && qx >= 0 && qx < q.W;
          fqx = in ? fqx : 0.f, fqy = in ? fqy : 0.f;
        }
        const float ddx = fqx - fpx, ddy = fqy - fpy, ddi = it[te] - fpi;
        const float e = (-0.5f * (ddx * ddx)) + (-0.5f * (ddy * ddy)) + (-0.5f * (ddi * ddi));
        const float k = q.weight * expf(e);
        ks += k;
#pragma unroll
        for (int c = 0; c < MC; ++c)
          if (c < C) m[c] = fmaf(k, yt[c * TSS + te], m[c]);
      }
    }
    float ym = 0.f;
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      if (c < C) {
        q.msg[((int64_t)n * C + c) * HW + (int64_t)gy * q.W + gx] = m[c];
        ym = fmaf(m[c], yt[c * TSS + ce], ym);
      }
    }
    v[0] = ks, v[1] = ym;
  }
  write_partials<2>(v, part, red);
}

// ---- fast form: C == 4, radius R known at compile time, W % 4 == 0.
// A workgroup owns 32 x 32 pixels, a thread 4 consecutive pixels of one row.  K(p, q) = w exp(-(dx^2 + dy^2) / (2 sxy^2)) *
// exp(-(I_q - I_p)^2 / (2 srgb^2)): the position factor does not depend on the pixel, so it enters as a per-tap constant
// lxy = log2(w) - log2(e) (dx^2 + dy^2) / (2 sxy^2) (kernel argument, scalar registers) and the image is staged pre-scaled by
// sqrt(log2(e) / 2) / srgb -- one subtract, one fma and one v_exp_f32 per tap and pixel:  k = exp2(lxy - (I'_q - I'_p)^2).
// Per tap row a thread reads the 4 + 2R image values and class vectors its four pixels need ONCE (aligned 16-byte LDS
// reads; y is staged class-interleaved, one float4 per pixel, row pitch odd in pixels -> conflict-free) and reuses them
// across the 2R + 1 column offsets: 3.1x fewer LDS reads than a read per tap.
// Taps outside the image (zero-padded unfold, gate_crf_loss.py:184-188) see the all-zero feature vector and y = 0: they add
// w exp(-(fx^2 + fy^2 + fi^2) / 2) of the PIXEL's own features to sum(K) and nothing to the message.  Only workgroups
// within R of the image border (BORDER, workgroup-uniform) pay for that: a staged validity plane turns sum(K) into
// sum(k * valid) + (#outside taps) * k_outside.
template <int R>
struct Crf4P {
  const float* y;
  const float* img;
  float* msg;
  int N, H, W, tiles_x, tiles_y;
  float sxy, srgb, weight, iscale;
  float lxy[R + 1][R + 1];   // [|dy|][|dx|]
};

template <int R>
struct Crf4Cfg {
  static constexpr int TW = 32, TH = 32;   // tile column c is image column x0 - R + c: the window of segment s starts at column 4 s
  static constexpr int WIN = 4 + 2 * R, WIN4 = (WIN + 3) / 4;                        // values / float4s per window
  static constexpr int COLS = 4 * (TW / 4 - 1) + 4 * WIN4;                           // columns a window read may touch
  static constexpr int YP = (COLS % 4 == 1) ? COLS : COLS + ((5 - COLS % 4) % 4);    // y pitch in pixels, == 1 (mod 4)
  static constexpr int IP = (COLS + 3) / 4 * 4;                                      // image / validity pitch in floats
  static constexpr int ROWS = TH + 2 * R;
  static constexpr size_t SMEM = sizeof(float) * ((size_t)ROWS * YP * 4 + 2 * (size_t)ROWS * IP);
};

__device__ __forceinline__ float crf_exp2(float x) {
#ifdef WSL_HOST_EMUL
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);   // v_exp_f32 (results below 2^-126 flush to zero: 40 orders below any term that counts)
#endif
}

template <int R, bool BORDER>
__device__ __forceinline__ void gatedcrf4_body(const Crf4P<R>& q, float* part, float* red, unsigned char* smem, int bid) {
  using C = Crf4Cfg<R>;
  constexpr int TW = C::TW, TH = C::TH, YP = C::YP, IP = C::IP, ROWS = C::ROWS, WIN4 = C::WIN4;
  float4* yt = reinterpret_cast<float4*>(smem);                 // [ROWS][YP] class vectors, 0 outside the image
  float* it = reinterpret_cast<float*>(yt + ROWS * YP);         // [ROWS][IP] image * iscale, 0 outside
  float* vt = it + ROWS * IP;                                   // [ROWS][IP] 1 inside the image, 0 outside (BORDER only)
  const int tx_i = bid % q.tiles_x;
  bid /= q.tiles_x;
  const int ty_i = bid % q.tiles_y, n = bid / q.tiles_y;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int H = q.H, W = q.W;
  const int64_t HW = (int64_t)H * W;
  const float* yn = q.y + (int64_t)n * 4 * HW;
  const float* in = q.img + (int64_t)n * HW;
  // ---- stage: column c of the tile arrays is image column x0 - R + c
  for (int e = threadIdx.x; e < ROWS * C::COLS; e += kThreads) {
    const int ty = e / C::COLS, tx = e - ty * C::COLS, gy = y0 + ty - R, gx = x0 + tx - R;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int64_t g = (int64_t)gy * W + gx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float iv = 0.f;
    if (ok) v = make_float4(yn[g], yn[HW + g], yn[2 * HW + g], yn[3 * HW + g]), iv = in[g] * q.iscale;
    yt[ty * YP + tx] = v;
    it[ty * IP + tx] = iv;
    if (BORDER) vt[ty * IP + tx] = ok ? 1.f : 0.f;
  }
  __syncthreads();
  const int seg = threadIdx.x & 7, ly = threadIdx.x >> 3;          // 8 segments of 4 pixels x 32 rows
  const int gy = y0 + ly, gx = x0 + 4 * seg;
  float v[2] = {0.f, 0.f};
  if (gy < H && gx < W) {                                           // (W % 4 == 0: a segment is inside or outside as a whole)
    float ip[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ip[p] = it[(ly + R) * IP + 4 * seg + R + p];
    wsl_v2f m01[4], m23[4];
    float ks[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) m01[p] = wsl_v2f{0.f, 0.f}, m23[p] = wsl_v2f{0.f, 0.f}, ks[p] = 0.f;
#pragma unroll 1
    for (int dy = -R; dy <= R; ++dy) {
      const int ady = dy < 0 ? -dy : dy;
      const float* irow = it + (ly + R + dy) * IP + 4 * seg;
      const float4* yrow = yt + (ly + R + dy) * YP + 4 * seg;
      float iw[4 * WIN4], vw[4 * WIN4];
#pragma unroll
      for (int j = 0; j < WIN4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(irow + 4 * j);
        iw[4 * j] = t.x, iw[4 * j + 1] = t.y, iw[4 * j + 2] = t.z, iw[4 * j + 3] = t.w;
        if (BORDER) {
          const float4 u = *reinterpret_cast<const float4*>(vt + (ly + R + dy) * IP + 4 * seg + 4 * j);
          vw[4 * j] = u.x, vw[4 * j + 1] = u.y, vw[4 * j + 2] = u.z, vw[4 * j + 3] = u.w;
        }
      }
      // per-tap constants of this row: a uniform (scalar) select over |dy|
      float lrow[R + 1];
#pragma unroll
      for (int a = 0; a <= R; ++a) {
        float l = q.lxy[0][a];
#pragma unroll
        for (int b = 1; b <= R; ++b) l = ady == b ? q.lxy[b][a] : l;
        lrow[a] = l;
      }
#pragma unroll
      for (int j = 0; j < 4 + 2 * R; ++j) {        // window column j = pixel offset (p + dx + R)
        const float4 yq = yrow[j];
        const wsl_v2f y01 = {yq.x, yq.y}, y23 = {yq.z, yq.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int dx = j - p - R;
          if (dx < -R || dx > R) continue;
          const int adx = dx < 0 ? -dx : dx;
          const float d = iw[j] - ip[p];
          float k = crf_exp2(fmaf(-d, d, lrow[adx]));
          if (dx == 0) k = dy == 0 ? 0.f : k;                      // centre tap := 0 (gate_crf_loss.py:171)
          const wsl_v2f kk = {k, k};
          m01[p] = __builtin_elementwise_fma(kk, y01, m01[p]);
          m23[p] = __builtin_elementwise_fma(kk, y23, m23[p]);
          if (BORDER) ks[p] = fmaf(k, vw[j], ks[p]);
          else ks[p] += k;
        }
      }
    }
    float ym = 0.f, ksum = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float4 yc = yt[(ly + R) * YP + 4 * seg + R + p];
      ym = fmaf(m01[p][0], yc.x, ym), ym = fmaf(m01[p][1], yc.y, ym), ym = fmaf(m23[p][0], yc.z, ym), ym = fmaf(m23[p][1], yc.w, ym);
      float kp = ks[p];
      if (BORDER) {
        // taps outside the image: all-zero feature vector against the pixel's own (x / sxy, y / sxy, I / srgb)
        const int px = gx + p;
        const int ny = (gy + R < H ? gy + R : H - 1) - (gy - R > 0 ? gy - R : 0) + 1;
        const int nx = (px + R < W ? px + R : W - 1) - (px - R > 0 ? px - R : 0) + 1;
        const int n_out = (2 * R + 1) * (2 * R + 1) - ny * nx;
        if (n_out > 0) {
          const float fx = (float)px / q.sxy, fy = (float)gy / q.sxy, fi = in[(int64_t)gy * W + px] / q.srgb;
          const float e = (-0.5f * (fx * fx)) + (-0.5f * (fy * fy)) + (-0.5f * (fi * fi));
          kp = fmaf((float)n_out, q.weight * expf(e), kp);
        }
      }
      ksum += kp;
    }
    float* mo = q.msg + (int64_t)n * 4 * HW + (int64_t)gy * W + gx;
    *reinterpret_cast<float4*>(mo) = make_float4(m01[0][0], m01[1][0], m01[2][0], m01[3][0]);
    *reinterpret_cast<float4*>(mo + HW) = make_float4(m01[0][1], m01[1][1], m01[2][1], m01[3][1]);
    *reinterpret_cast<float4*>(mo + 2 * HW) = make_float4(m23[0][0], m23[1][0], m23[2][0], m23[3][0]);
    *reinterpret_cast<float4*>(mo + 3 * HW) = make_float4(m23[0][1], m23[1][1], m23[2][1], m23[3][1]);
    v[0] = ksum, v[1] = ym;
  }
  write_partials<2>(v, part, red);
}

template <int R>
__global__ __launch_bounds__(256) void gatedcrf_fwd4_kernel(Crf4P<R> q, float* part) {
  WSL_DYN_SMEM(smem);
  __shared__ float red[4];
  using C = Crf4Cfg<R>;
  // XCD-aware tile order (the hardware deals workgroups to the eight XCDs round-robin): every XCD takes a contiguous range of tiles, so
  // the (2 R + 1)-wide halos of neighbouring tiles meet in one L2 (round-robin tiles fetched 2.9 x the bytes of y and the image)
  const int nb = (int)gridDim.x;
  const int tile = (nb & 7) == 0 ? ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  int bid = tile;
  const int tx_i = bid % q.tiles_x;
  bid /= q.tiles_x;
  const int ty_i = bid % q.tiles_y;
  const int y0 = ty_i * C::TH, x0 = tx_i * C::TW;
  const bool interior = y0 - R >= 0 && y0 + C::TH + R <= q.H && x0 - R >= 0 && x0 + C::TW + R <= q.W;   // uniform
  if (interior) gatedcrf4_body<R, false>(q, part, red, smem, tile);
  else gatedcrf4_body<R, true>(q, part, red, smem, tile);
}

__global__ __launch_bounds__(256) void gatedcrf_finalize_kernel(const float* part, int nblk, double denom, float* loss) {
  __shared__ double red[kThreads];
  const double ks = col_sum(part, nblk, 2, 0, red), ym = col_sum(part, nblk, 2, 1, red);
  if (threadIdx.x == 0) loss[0] = (float)((ks - ym) / denom);
}

__global__ __launch_bounds__(256) void scale_kernel(const float* x, const float* gout, float k, float* out, int64_t n) {
  const float s = k * (gout ? gout[0] : 1.f);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) out[i] = s * x[i];
}

__global__ __launch_bounds__(256) void axpy_kernel(float* dst, const float* src, float k, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
    dst[i] = fmaf(k, src[i], dst[i]);
}

// ------------------------------------------------------------------------------------------------ tv_loss
// 16x16 outputs of one (n, c) plane per workgroup.  Regions (halo): p +4, erosion e +3 (with arg-min), dilation +2
// (contour flag + arg-max), ge +1, dp +0.  Windows ignore out-of-image taps (max_pool2d pads with -inf); extrema take
// the first position in row-major window order.
struct TvP {
  const float* p;
  float* dp;
  int n0, N, C, H, W, tiles_x, tiles_y;
  float gk;  // gscale / numel
};

__global__ __launch_bounds__(256) void tv_fwd_bwd_kernel(TvP q, float* part) {
  __shared__ float pt[24 * 24];
  __shared__ float et[22 * 22];
  __shared__ int eargt[22 * 22];
  __shared__ int dargt[20 * 20];   // arg-max of the dilation (index into et) or -1 when contour <= 0 / outside
  __shared__ float get[18 * 18];
  __shared__ float red[4];
  int bid = blockIdx.x;
  const int tx_i = bid % q.tiles_x;
  bid /= q.tiles_x;
  const int ty_i = bid % q.tiles_y;
  bid /= q.tiles_y;
  const int c = bid % q.C, n = bid / q.C;
  const int y0 = ty_i * 16, x0 = tx_i * 16;
  const int64_t plane = ((int64_t)n * q.C + c) * q.H * q.W;
  const bool active = n >= q.n0;
  float v[1] = {0.f};
  for (int e = threadIdx.x; e < 24 * 24; e += kThreads) {
    const int gy = y0 + e / 24 - 4, gx = x0 + e % 24 - 4;
    pt[e] = (gy >= 0 && gy < q.H && gx >= 0 && gx < q.W) ? q.p[plane + (int64_t)gy * q.W + gx] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 22 * 22; e += kThreads) {       // erosion on halo 3
    const int ty = e / 22, tx = e % 22, gy = y0 + ty - 3, gx = x0 + tx - 3;
    float best = 0.f;
    int arg = -1;
    if (gy >= 0 && gy < q.H && gx >= 0 && gx < q.W) {
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = gy + dy, xx = gx + dx;
          if (yy < 0 || yy >= q.H || xx < 0 || xx >= q.W) continue;
          const int pe = (ty + 1 + dy) * 24 + tx + 1 + dx;
          if (arg < 0 || pt[pe] < best) best = pt[pe], arg = pe;
        }
    }
    et[e] = best;
    eargt[e] = arg;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 20 * 20; e += kThreads) {       // dilation + contour on halo 2
    const int ty = e / 20, tx = e % 20, gy = y0 + ty - 2, gx = x0 + tx - 2;
    int res = -1;
    if (gy >= 0 && gy < q.H && gx >= 0 && gx < q.W) {
      float best = 0.f;
      int arg = -1;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = gy + dy, xx = gx + dx;
          if (yy < 0 || yy >= q.H || xx < 0 || xx >= q.W) continue;
          const int ee = (ty + 1 + dy) * 22 + tx + 1 + dx;
          if (arg < 0 || et[ee] > best) best = et[ee], arg = ee;
        }
      const float contour = best - et[(ty + 1) * 22 + tx + 1];
      if (contour > 0.f) {
        res = arg;
        if (active && ty >= 2 && ty < 18 && tx >= 2 && tx < 18) v[0] += contour;
      }
    }
    dargt[e] = res;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 18 * 18; e += kThreads) {       // ge (gradient wrt the erosion output) on halo 1
    const int ty = e / 18, tx = e % 18, gy = y0 + ty - 1, gx = x0 + tx - 1;
    float g = 0.f;
    if (gy >= 0 && gy < q.H && gx >= 0 && gx < q.W) {
      const int me = (ty + 2) * 22 + tx + 2;  // this position in et coordinates
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = gy + dy, xx = gx + dx;
          if (yy < 0 || yy >= q.H || xx < 0 || xx >= q.W) continue;
          if (dargt[(ty + 1 + dy) * 20 + tx + 1 + dx] == me) g += 1.f;
        }
      if (dargt[(ty + 1) * 20 + tx + 1] >= 0) g -= 1.f;
    }
    get[e] = g;
  }
  __syncthreads();
  {
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15, gy = y0 + ty, gx = x0 + tx;
    if (gy < q.H && gx < q.W) {
      float g = 0.f;
      if (active) {
        const int pe = (ty + 4) * 24 + tx + 4;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int yy = gy + dy, xx = gx + dx;
            if (yy < 0 || yy >= q.H || xx < 0 || xx >= q.W) continue;
            if (eargt[(ty + 3 + dy) * 22 + tx + 3 + dx] == pe) g += get[(ty + 1 + dy) * 18 + tx + 1 + dx];
          }
      }
      q.dp[plane + (int64_t)gy * q.W + gx] = g * q.gk;
    }
  }
  write_partials<1>(v, part, red);
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* part, int nblk, int K, double k, float* loss) {
  __shared__ double red[kThreads];
  double s = 0;
  for (int j = 0; j < K; ++j) s += col_sum(part, nblk, K, j, red);
  if (threadIdx.x == 0) loss[0] = (float)(s * k);
}

// ------------------------------------------------------------------------------------------------ Mumford-Shah
__global__ __launch_bounds__(256) void ms_moment_kernel(const float* img, const float* p, int C, int HW, int chunks,
                                                        float* mom) {  // mom[n][chunk][C+1]
  __shared__ float red[4];
  const int n = blockIdx.y, base = blockIdx.x * 4096;
  float v[kMaxC + 1];
  for (int k = 0; k <= kMaxC; ++k) v[k] = 0.f;
  for (int i = base + threadIdx.x; i < base + 4096 && i < HW; i += kThreads) {
    const float I = img[(int64_t)n * HW + i];
    v[kMaxC] += I;
    for (int c = 0; c < C; ++c) v[c] = fmaf(p[((int64_t)n * C + c) * HW + i], I, v[c]);
  }
  for (int k = 0; k <= C; ++k) {
    const float s = block_sum(v[k == C ? kMaxC : k], red);
    if (threadIdx.x == 0) mom[((int64_t)n * chunks + blockIdx.x) * (C + 1) + k] = s;
  }
}

__global__ __launch_bounds__(256) void ms_main_kernel(const float* img, const float* p, const float* mom, int C, int H,
                                                      int W, int chunks, float gscale, float* dp, float* part) {
  __shared__ float red[4];
  __shared__ float cen[kMaxC + 1];
  const int n = blockIdx.y, HW = H * W, base = blockIdx.x * 4096;
  if (threadIdx.x <= (unsigned)C) {  // per-sample sums: classes 0..C-1 = sum(p_c*I), entry C = sum(I)
    double s = 0;
    for (int k = 0; k < chunks; ++k) s += mom[((int64_t)n * chunks + k) * (C + 1) + threadIdx.x];
    cen[threadIdx.x] = (float)s;
  }
  __syncthreads();
  float cv[kMaxC];
  for (int c = 0; c < C; ++c) cv[c] = cen[c] / cen[C];   // centroid divides by sum(I) (losses.py:286-287)
  float v[2] = {0.f, 0.f};
  for (int i = base + threadIdx.x; i < base + 4096 && i < HW; i += kThreads) {
    const int y = i / W, x = i - y * W;
    const float I = img[(int64_t)n * HW + i];
    for (int c = 0; c < C; ++c) {
      const float* pc = p + ((int64_t)n * C + c) * HW;
      const float pv = pc[i], d = pv - cv[c];
      v[0] = fmaf(d * d, I, v[0]);
      float g = 2.f * d * I;
      if (y + 1 < H) { const float t = pc[i + W] - pv; v[1] += fabsf(t); g -= (t > 0.f) - (t < 0.f); }
      if (x + 1 < W) { const float t = pc[i + 1] - pv; v[1] += fabsf(t); g -= (t > 0.f) - (t < 0.f); }
      if (y > 0) { const float t = pv - pc[i - W]; g += (t > 0.f) - (t < 0.f); }
      if (x > 0) { const float t = pv - pc[i - 1]; g += (t > 0.f) - (t < 0.f); }
      dp[((int64_t)n * C + c) * HW + i] = g * gscale;
    }
  }
  for (int k = 0; k < 2; ++k) {
    const float s = block_sum(v[k], red);
    if (threadIdx.x == 0) part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + k] = s;
  }
}

// ------------------------------------------------------------------------------------------------ softmax MSE
__global__ __launch_bounds__(256) void softmax_mse_kernel(const float* a, const float* b, int C, int HW, int64_t P,
                                                          float k, float* da, float* part) {
  __shared__ float red[4];
  float v[1] = {0.f};
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float sa[kMaxC], sb[kMaxC], ds[kMaxC], dot = 0.f;
    softmax_c(a + base, HW, C, sa);
    softmax_c(b + base, HW, C, sb);
    for (int c = 0; c < C; ++c) {
      const float d = sa[c] - sb[c];
      v[0] = fmaf(d, d, v[0]);
      ds[c] = 2.f * d * k;
      dot = fmaf(ds[c], sa[c], dot);
    }
    for (int c = 0; c < C; ++c) da[base + (int64_t)c * HW] = sa[c] * (ds[c] - dot);
  }
  write_partials<1>(v, part, red);
}

// the same term for the FUSED regulariser head (wsl_head_reg_fwd_bwd): the student's softmax `sa` is already in memory and the
// gradient is wanted with respect to it (the head's backward pass does the ONE softmax backward of all terms): ds += 2 (sa - sb) k
__global__ __launch_bounds__(256) void softmax_mse_ds_kernel(const float* sa_, const float* b, int C, int HW, int64_t P, float k,
                                                             float* ds, float* part) {
  __shared__ float red[4];
  float v[1] = {0.f};
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float sb[kMaxC];
    softmax_c(b + base, HW, C, sb);
    for (int c = 0; c < C; ++c) {
      const int64_t idx = base + (int64_t)c * HW;
      const float d = sa_[idx] - sb[c];
      v[0] = fmaf(d, d, v[0]);
      ds[idx] += 2.f * d * k;
    }
  }
  write_partials<1>(v, part, red);
}

// ------------------------------------------------------------------------------------------------ entropy minimisation
// entropy_loss(p, C) = mean_px( -sum_c p log(p + 1e-6) ) / log(C)   (ref: utils/losses.py:30-36; used on softmax(outputs)
// with weight 0.1 by train_weakly_supervised_pCE_Entropy_Mini_2D.py:99-102).  d/dp_c = -(log(p_c + 1e-6) + p_c/(p_c + 1e-6)) * k.
__global__ __launch_bounds__(256) void entropy_kernel(const float* p, int C, int HW, int64_t P, float k, float* dp,
                                                      float* part) {
  __shared__ float red[4];
  float v[1] = {0.f};
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, q = i - n * HW, base = n * C * HW + q;
    for (int c = 0; c < C; ++c) {
      const float x = p[base + (int64_t)c * HW], xe = x + 1e-6f, lg = logf(xe);
      v[0] = fmaf(-x, lg, v[0]);
      dp[base + (int64_t)c * HW] = -(lg + x / xe) * k;
    }
  }
  write_partials<1>(v, part, red);
}

// ------------------------------------------------------------------------------------------------ uncertainty-aware mean teacher
// Pieces of train_weakly_supervised_ustm_2D.py:121-157.
// torch.rot90(x, k, [2, 3]) of every [H, W] plane (output planes are [W, H] for odd k)
__global__ __launch_bounds__(256) void rot90_kernel(const float* x, float* y, int H, int W, int k) {
  const int Ho = (k & 1) ? W : H, Wo = (k & 1) ? H : W;
  const float* xp = x + (int64_t)blockIdx.y * H * W;
  float* yp = y + (int64_t)blockIdx.y * H * W;
  for (int o = blockIdx.x * kThreads + threadIdx.x; o < Ho * Wo; o += gridDim.x * kThreads) {
    const int i = o / Wo, j = o - i * Wo;
    int sy, sx;
    switch (k & 3) {
      case 0: sy = i, sx = j; break;
      case 1: sy = j, sx = W - 1 - i; break;
      case 2: sy = H - 1 - i, sx = W - 1 - j; break;
      default: sy = H - 1 - j, sx = i; break;
    }
    yp[o] = xp[sy * W + sx];
  }
}

// acc = (init ? 0 : acc) + scale * softmax(z): the Monte-Carlo mean of the teacher's predictions, one pass per forward
__global__ __launch_bounds__(256) void softmax_accum_kernel(const float* z, float* acc, float scale, int init, int C, int HW,
                                                            int64_t P) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, q = i - n * HW, base = n * C * HW + q;
    float sm[kMaxC];
    softmax_c(z + base, HW, C, sm);
    for (int c = 0; c < C; ++c) {
      const int64_t e = base + (int64_t)c * HW;
      acc[e] = init ? scale * sm[c] : fmaf(scale, sm[c], acc[e]);
    }
  }
}

// mask(px) = [ -sum_c pm log(pm + 1e-6) < threshold ]  (uncertainty of the mean prediction pm)
__device__ __forceinline__ bool ustm_certain(const float* pm, int64_t base, int HW, int C, float threshold) {
  float u = 0.f;
  for (int c = 0; c < C; ++c) {
    const float x = pm[base + (int64_t)c * HW];
    u = fmaf(-x, logf(x + 1e-6f), u);
  }
  return u < threshold;
}

// partials: { sum_px mask * sum_c (softmax(a) - softmax(b))^2 , sum_px mask }
__global__ __launch_bounds__(256) void ustm_reduce_kernel(const float* a, const float* b, const float* pm, float threshold,
                                                          int C, int HW, int64_t P, float* part) {
  __shared__ float red[4];
  float v[2] = {0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, q = i - n * HW, base = n * C * HW + q;
    if (!ustm_certain(pm, base, HW, C, threshold)) continue;
    float sa[kMaxC], sb[kMaxC];
    softmax_c(a + base, HW, C, sa);
    softmax_c(b + base, HW, C, sb);
    for (int c = 0; c < C; ++c) {
      const float d = sa[c] - sb[c];
      v[0] = fmaf(d, d, v[0]);
    }
    v[1] += 1.f;
  }
  write_partials<2>(v, part, red);
}

// loss[0] = S / (2 M + 1e-16), loss[1] = M, scal[0] = gradient factor 1 / (2 M + 1e-16)
__global__ __launch_bounds__(256) void ustm_finalize_kernel(const float* part, int nblk, float* loss) {
  __shared__ double red[kThreads];
  const double S = col_sum(part, nblk, 2, 0, red), M = col_sum(part, nblk, 2, 1, red);
  if (threadIdx.x == 0) {
    const double den = 2.0 * M + 1e-16;
    loss[0] = (float)(S / den);
    loss[1] = (float)M;
    loss[2] = (float)(1.0 / den);
  }
}

__global__ __launch_bounds__(256) void ustm_bwd_kernel(const float* a, const float* b, const float* pm, float threshold,
                                                       const float* loss, float gscale, int C, int HW, int64_t P, float* da) {
  const float k = 2.f * gscale * loss[2];
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, q = i - n * HW, base = n * C * HW + q;
    if (!ustm_certain(pm, base, HW, C, threshold)) {
      for (int c = 0; c < C; ++c) da[base + (int64_t)c * HW] = 0.f;
      continue;
    }
    float sa[kMaxC], sb[kMaxC], ds[kMaxC], dot = 0.f;
    softmax_c(a + base, HW, C, sa);
    softmax_c(b + base, HW, C, sb);
    for (int c = 0; c < C; ++c) {
      ds[c] = k * (sa[c] - sb[c]);
      dot = fmaf(ds[c], sa[c], dot);
    }
    for (int c = 0; c < C; ++c) da[base + (int64_t)c * HW] = sa[c] * (ds[c] - dot);
  }
}

// ------------------------------------------------------------------------------------------------ mixed probabilities
// y = beta*softmax(z1) + (1-beta)*softmax(z2): the prediction the dual-branch + GatedCRF composition regularises
// (ref: train_ACDC_scribblevc.py:171-206).  z2 == NULL: y = softmax(z1).
template <int CT>
__global__ __launch_bounds__(256) void mixprob_fwd_kernel(const float* z1, const float* z2, float bf, float omb, float* y,
                                                          int C_, int HW, int64_t P) {
  const int C = CT > 0 ? CT : C_;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float s1[kMaxC], s2[kMaxC];
    softmax_c(z1 + base, HW, C, s1);
    if (z2) {
      softmax_c(z2 + base, HW, C, s2);
      for (int c = 0; c < C; ++c) y[base + (int64_t)c * HW] = __fadd_rn(__fmul_rn(bf, s1[c]), __fmul_rn(omb, s2[c]));
    } else {
      for (int c = 0; c < C; ++c) y[base + (int64_t)c * HW] = s1[c];
    }
  }
}

// dz_k (+)= softmax_bwd(s_k, w_k * k * dy)   with w_1 = beta, w_2 = 1-beta
template <int CT>
__global__ __launch_bounds__(256) void mixprob_bwd_kernel(const float* z1, const float* z2, float bf, float omb,
                                                          const float* dy, float k, float* dz1, float* dz2, int accumulate,
                                                          int C_, int HW, int64_t P) {
  const int C = CT > 0 ? CT : C_;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / HW, p = i - n * HW, base = n * C * HW + p;
    float g[kMaxC], s[kMaxC];
    for (int c = 0; c < C; ++c) g[c] = k * dy[base + (int64_t)c * HW];
    for (int br = 0; br < (z2 ? 2 : 1); ++br) {
      const float wgt = z2 ? (br == 0 ? bf : omb) : 1.f;
      float* dz = br == 0 ? dz1 : dz2;
      softmax_c((br == 0 ? z1 : z2) + base, HW, C, s);
      float dot = 0.f;
      for (int c = 0; c < C; ++c) dot = fmaf(g[c], s[c], dot);
      for (int c = 0; c < C; ++c) {
        const float v = wgt * s[c] * (g[c] - dot);
        dz[base + (int64_t)c * HW] = accumulate ? dz[base + (int64_t)c * HW] + v : v;
      }
    }
  }
}

static int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace wsl

using namespace wsl;

extern "C" size_t wsl_loss_ws_bytes(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  const size_t tiles = (size_t)N * C * ((HW + 255) / 256 + 64);  // generous bound on 16x16 tile counts (any aspect)
  const size_t part = (tiles > (size_t)kMaxBlocks ? tiles : (size_t)kMaxBlocks) * kMaxK;
  // + a region of its own for the GatedCRF partials (2 per tile of >= 256 pixels) when the CRF runs between the two passes of
  //   the fused head (wsl_head_gatedcrf_fwd_bwd): the head's coefficients must survive it
  return sizeof(float) * (part + 64 + (size_t)N * ((HW + 4095) / 4096) * (kMaxC + 1) + 2 * (size_t)N * ((HW + 255) / 256 + 64));
}

#define WSL_WS_OK(fn)                                                              \
  do {                                                                             \
    if (!ws || ws_bytes < wsl_loss_ws_bytes(N, C, HW_)) {                           \
      set_error(fn ": workspace %zu < %zu", ws_bytes, wsl_loss_ws_bytes(N, C, HW_)); \
      return WSL_EWORKSPACE;                                                       \
    }                                                                              \
  } while (0)

extern "C" int wsl_softmax_fwd(const float* z, float* s, int N, int C, int HW, void* stream) {
  WSL_REQUIRE(z && s && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "softmax_fwd: bad args (C <= %d)", kMaxC);
  const int64_t P = (int64_t)N * HW;
  WSL_LAUNCH(softmax_fwd_kernel, dim3(grid_for(P)), dim3(kThreads), 0, stream, z, s, C, HW, P);
  return check_launch("softmax_fwd_kernel");
}

extern "C" int wsl_softmax_bwd(const float* s, const float* ds, float* dz, int N, int C, int HW, void* stream) {
  WSL_REQUIRE(s && ds && dz && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "softmax_bwd: bad args");
  const int64_t P = (int64_t)N * HW;
  WSL_LAUNCH(softmax_bwd_kernel, dim3(grid_for(P)), dim3(kThreads), 0, stream, s, ds, dz, C, HW, P);
  return check_launch("softmax_bwd_kernel");
}

extern "C" int wsl_ce_fwd_bwd(const float* z, const void* label, int label_i64, int ignore, float* loss, float* dz,
                              float gscale, int N, int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(z && label && loss && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "ce_fwd_bwd: bad args");
  const int HW_ = HW;
  WSL_WS_OK("ce_fwd_bwd");
  const int64_t P = (int64_t)N * HW;
  const int nb = grid_for(P);
  float* part = static_cast<float*>(ws);
  float* scal = part + (size_t)kMaxBlocks * kMaxK;
  WSL_LAUNCH(ce_reduce_kernel, dim3(nb), dim3(kThreads), 0, stream, z, label, label_i64, ignore, C, HW, P, part);
  WSL_LAUNCH(ce_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, loss, scal);
  if (dz) WSL_LAUNCH(ce_bwd_kernel, dim3(nb), dim3(kThreads), 0, stream, z, label, label_i64, ignore, C, HW, P, scal, gscale, dz);
  return check_launch("ce_fwd_bwd");
}

extern "C" int wsl_mix_argmax(const float* s1, const float* s2, double beta, int64_t* pseudo, int N, int C, int HW,
                              void* stream) {
  WSL_REQUIRE(s1 && s2 && pseudo && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "mix_argmax: bad args");
  const int64_t P = (int64_t)N * HW;
  WSL_LAUNCH(mix_argmax_kernel, dim3(grid_for(P)), dim3(kThreads), 0, stream, s1, s2, (float)beta, (float)(1.0 - beta),
             pseudo, C, HW, P);
  return check_launch("mix_argmax_kernel");
}

extern "C" int wsl_pdice_fwd(const float* s, const void* target, int target_i64, int ignore, float* loss, float* sums,
                             int N, int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(s && target && loss && sums && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "pdice_fwd: bad args");
  const int HW_ = HW;
  WSL_WS_OK("pdice_fwd");
  const int nb = grid_for(HW);
  float* part = static_cast<float*>(ws);
  WSL_LAUNCH(pdice_reduce_kernel, dim3(nb), dim3(kThreads), 0, stream, s, target, target_i64, ignore, N, C, HW, part);
  WSL_LAUNCH(pdice_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, C, loss, sums);
  return check_launch("pdice_fwd");
}

extern "C" int wsl_pdice_bwd(const float* s, const void* target, int target_i64, int ignore, const float* sums,
                             const float* gout, float* ds, int N, int C, int HW, void* stream) {
  WSL_REQUIRE(s && target && sums && ds && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "pdice_bwd: bad args");
  WSL_LAUNCH(pdice_bwd_kernel, dim3(grid_for(HW)), dim3(kThreads), 0, stream, s, target, target_i64, ignore, sums, gout,
             ds, N, C, HW);
  return check_launch("pdice_bwd_kernel");
}

static int head_stage1(const HeadP& h, int64_t* pseudo, float* y_mix, float w_pse, float* out, int C, int N, int dual, float* part,
                       float* scal, int nb, void* stream) {
  if (C == 4) WSL_LAUNCH((head_reduce_kernel<4>), dim3(nb), dim3(kThreads), 0, stream, h, pseudo, part, y_mix);
  else WSL_LAUNCH((head_reduce_kernel<0>), dim3(nb), dim3(kThreads), 0, stream, h, pseudo, part, y_mix);
  WSL_LAUNCH(head_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, C, N, dual, w_pse, out, scal);
  return WSL_OK;
}
static void head_stage2(const HeadP& h, const float* scal, float gscale, float* dz1, float* dz2, const float* dy_mix, float ky, int C,
                        int nb, void* stream) {
  if (C == 4) WSL_LAUNCH((head_bwd_kernel<4>), dim3(nb), dim3(kThreads), 0, stream, h, scal, gscale, dz1, dz2, dy_mix, ky);
  else WSL_LAUNCH((head_bwd_kernel<0>), dim3(nb), dim3(kThreads), 0, stream, h, scal, gscale, dz1, dz2, dy_mix, ky);
}

extern "C" int wsl_head_fwd_bwd(const float* z1, const float* z2, const uint8_t* label, int ignore, double beta,
                                float w_pse, float gscale, float* out, int64_t* pseudo, float* dz1, float* dz2, int N,
                                int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(z1 && label && out && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "head_fwd_bwd: bad args");
  WSL_REQUIRE((dz1 == nullptr) || (z2 == nullptr) || (dz2 != nullptr), "head_fwd_bwd: dz2 missing");
  const int HW_ = HW;
  WSL_WS_OK("head_fwd_bwd");
  HeadP h{z1, z2, label, ignore, C, HW, N, (int64_t)N * HW, (float)beta, (float)(1.0 - beta)};
  // forward reads the logits + 1 B label; backward reads them again and writes the logit gradients
  const double nbr = z2 ? 2.0 : 1.0;
  ProfScope ps(PF_LOSS_HEAD, 0.0, (double)N * HW * ((dz1 ? 2.0 : 1.0) * (4.0 * C * nbr + 1.0) + (dz1 ? 4.0 * C * nbr : 0.0) + (pseudo ? 8.0 : 0.0)), stream);
  const int nb = grid_for(h.P);
  float* part = static_cast<float*>(ws);
  float* scal = part + (size_t)kMaxBlocks * kMaxK;
  head_stage1(h, pseudo, nullptr, w_pse, out, C, N, z2 ? 1 : 0, part, scal, nb, stream);
  if (dz1) head_stage2(h, scal, gscale, dz1, dz2, nullptr, 0.f, C, nb, stream);
  return check_launch("head_fwd_bwd");
}

template <int R>
static int crf4_launch(const float* y, const float* img, float* msg, int N, int H, int W, float sxy, float srgb, float weight,
                       float* part, int nb, void* stream) {
  using C = Crf4Cfg<R>;
  Crf4P<R> q;
  q.y = y, q.img = img, q.msg = msg, q.N = N, q.H = H, q.W = W, q.tiles_x = cdiv(W, C::TW), q.tiles_y = cdiv(H, C::TH);
  q.sxy = sxy, q.srgb = srgb, q.weight = weight;
  const double l2e = 1.4426950408889634;
  q.iscale = (float)(sqrt(0.5 * l2e) / (double)srgb);
  for (int b = 0; b <= R; ++b)
    for (int a = 0; a <= R; ++a)
      q.lxy[b][a] = (float)(log2((double)weight) - l2e * 0.5 * (double)(a * a + b * b) / ((double)sxy * (double)sxy));
  auto kern = gatedcrf_fwd4_kernel<R>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)WSL_SET_MAX_DYN_SMEM(kern, C::SMEM);
    attr_done = true;
  }
  WSL_LAUNCH(kern, dim3(nb), dim3(kThreads), C::SMEM, stream, q, part);
  return check_launch("gatedcrf_fwd4_kernel");
}

static int gatedcrf_fwd_impl(const float* y, const float* img, float* msg, float* loss, int N, int C, int H, int W, int radius,
                             float sigma_xy, float sigma_rgb, float weight, float* part, void* stream) {
  WSL_REQUIRE(y && img && msg && loss && N > 0 && H > 0 && W > 0 && C > 0 && C <= kMaxC, "gatedcrf_fwd: bad args");
  if (radius < 1 || radius > 8) {
    set_error("gatedcrf_fwd: radius %d not built (1..8 are)", radius);
    return WSL_EUNSUPPORTED;
  }
  CrfP q{y, img, msg, N, C, H, W, radius, sigma_xy, sigma_rgb, weight, cdiv(W, kCrfTW), cdiv(H, kCrfTH)};
  const int nb = N * q.tiles_x * q.tiles_y;
  if (C == 4 && (radius == 5 || radius == 2) && (W & 3) == 0 && weight > 0.f &&
      (reinterpret_cast<uintptr_t>(msg) & 15) == 0) {
    const int nb4 = N * cdiv(W, 32) * cdiv(H, 32);
    ProfScope ps(PF_GATEDCRF, 0.0, 4.0 * (double)N * H * W * (2 * C + 1), stream);
    const int rc = radius == 5 ? crf4_launch<5>(y, img, msg, N, H, W, sigma_xy, sigma_rgb, weight, part, nb4, stream)
                               : crf4_launch<2>(y, img, msg, N, H, W, sigma_xy, sigma_rgb, weight, part, nb4, stream);
    if (rc) return rc;
    WSL_LAUNCH(gatedcrf_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb4, (double)N * H * W, loss);
    return check_launch("gatedcrf_fwd");
  }
  const int TSX = kCrfTW + 2 * radius, TSY = kCrfTH + 2 * radius;
  const size_t smem = sizeof(float) * ((size_t)(C + 1) * TSX * TSY + TSX + TSY);
  void* tok = prof_begin(PF_GATEDCRF, 0.0, 4.0 * (double)N * H * W * (2 * C + 1), stream);   // 36 B/px at C = 4
  if (C == 4) WSL_LAUNCH((gatedcrf_fwd_kernel<4>), dim3(nb), dim3(kThreads), smem, stream, q, part);
  else if (C == 2) WSL_LAUNCH((gatedcrf_fwd_kernel<2>), dim3(nb), dim3(kThreads), smem, stream, q, part);
  else WSL_LAUNCH((gatedcrf_fwd_kernel<0>), dim3(nb), dim3(kThreads), smem, stream, q, part);
  prof_end(tok, stream);
  WSL_LAUNCH(gatedcrf_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, (double)N * H * W, loss);
  return check_launch("gatedcrf_fwd");
}

extern "C" int wsl_gatedcrf_fwd(const float* y, const float* img, float* msg, float* loss, int N, int C, int H, int W,
                                int radius, float sigma_xy, float sigma_rgb, float weight, void* ws, size_t ws_bytes,
                                void* stream) {
  WSL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "gatedcrf_fwd: bad args");
  const int HW_ = H * W;
  WSL_WS_OK("gatedcrf_fwd");
  return gatedcrf_fwd_impl(y, img, msg, loss, N, C, H, W, radius, sigma_xy, sigma_rgb, weight, static_cast<float*>(ws), stream);
}

extern "C" int wsl_gatedcrf_bwd(const float* msg, const float* gout, float gscale, float* dy, int N, int C, int H, int W,
                                void* stream) {
  WSL_REQUIRE(msg && dy && N > 0 && C > 0 && H > 0 && W > 0, "gatedcrf_bwd: bad args");
  const int64_t n = (int64_t)N * C * H * W;
  WSL_LAUNCH(scale_kernel, dim3(grid_for(n / 4 + 1)), dim3(kThreads), 0, stream, msg, gout,
             (float)(-2.0 * gscale / ((double)N * H * W)), dy, n);
  return check_launch("scale_kernel");
}

// The headline composition in one entry point (ref: train_weakly_supervised_pCE_GatedCRFLoss_2D.py:108-123; dual branch as
// train_ACDC_scribblevc.py:171-206):  loss = pCE(z1 [, z2]) + crf_weight * GatedCRF(y, image),  y = beta softmax(z1) + (1 - beta)
// softmax(z2) (single branch: softmax(z1)).  Same kernels as wsl_head_fwd_bwd + wsl_mixprob_fwd + wsl_gatedcrf_fwd + wsl_mixprob_bwd,
// but y is written by the head's reduction pass and the gradient through y is added inside the head's backward pass: two launches
// and ~400 MB of logit / gradient re-reads fewer at 64 x 256 x 256; same results to the last ulp.
// out[0..3] as wsl_head_fwd_bwd (w_pse = 0), out[4] = the raw GatedCRF loss.  y, msg: [N,C,H,W] buffers (msg is kept: it is the
// CRF term's gradient).
extern "C" int wsl_head_gatedcrf_fwd_bwd(const float* z1, const float* z2, const uint8_t* label, int ignore, double beta,
                                         const float* img, int radius, float sigma_xy, float sigma_rgb, float weight,
                                         float crf_weight, float* out, float* dz1, float* dz2, float* y, float* msg, int N, int C,
                                         int H, int W, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(z1 && label && img && out && dz1 && y && msg && N > 0 && H > 0 && W > 0 && C > 0 && C <= kMaxC,
              "head_gatedcrf_fwd_bwd: bad args");
  WSL_REQUIRE(z2 == nullptr || dz2 != nullptr, "head_gatedcrf_fwd_bwd: dz2 missing");
  const int HW = H * W, HW_ = HW;
  WSL_WS_OK("head_gatedcrf_fwd_bwd");
  HeadP h{z1, z2, label, ignore, C, HW, N, (int64_t)N * HW, (float)beta, (float)(1.0 - beta)};
  const int nb = grid_for(h.P);
  float* part = static_cast<float*>(ws);
  float* scal = part + (size_t)kMaxBlocks * kMaxK;
  const double nbr = z2 ? 2.0 : 1.0;
  {
    ProfScope ps(PF_LOSS_HEAD, 0.0, (double)N * HW * (4.0 * C * nbr + 1.0 + 4.0 * C), stream);
    head_stage1(h, nullptr, y, 0.f, out, C, N, z2 ? 1 : 0, part, scal, nb, stream);
  }
  // the CRF's partials go behind the head's coefficients (the last region of wsl_loss_ws_bytes), which the backward pass needs
  float* crf_part = static_cast<float*>(ws) + (ws_bytes / sizeof(float) - 2 * (size_t)N * ((HW + 255) / 256 + 64));
  if (int rc = gatedcrf_fwd_impl(y, img, msg, out + 4, N, C, H, W, radius, sigma_xy, sigma_rgb, weight, crf_part, stream)) return rc;
  {
    ProfScope ps(PF_LOSS_HEAD, 0.0, (double)N * HW * (4.0 * C * nbr + 1.0 + 4.0 * C + 4.0 * C * nbr), stream);
    head_stage2(h, scal, 1.f, dz1, dz2, msg, (float)(-2.0 * (double)crf_weight / ((double)N * HW)), C, nb, stream);
  }
  return check_launch("head_gatedcrf_fwd_bwd");
}

extern "C" int wsl_head_reg_fwd_bwd(const float* z, const uint8_t* label, int ignore, float w_ce, int reg_kind, float reg_weight,
                                    const float* img, const float* zt, float cons_weight, float* out, float* dz, float* s, float* ds,
                                    int N, int C, int H, int W, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(z && label && out && dz && s && ds && N > 0 && H > 0 && W > 0 && C > 0 && C <= kMaxC, "head_reg_fwd_bwd: bad args");
  WSL_REQUIRE(reg_kind >= WSL_REG_TV && reg_kind <= WSL_REG_ENTROPY, "head_reg_fwd_bwd: reg_kind %d", reg_kind);
  WSL_REQUIRE(reg_kind != WSL_REG_MS || img, "head_reg_fwd_bwd: Mumford-Shah needs the image");
  const int HW = H * W, HW_ = HW;
  WSL_WS_OK("head_reg_fwd_bwd");
  HeadP h{z, nullptr, label, ignore, C, HW, N, (int64_t)N * HW, 0.f, 1.f};
  const int nb = grid_for(h.P);
  float* part = static_cast<float*>(ws);
  float* scal = part + (size_t)kMaxBlocks * kMaxK;
  // the regulariser's partial sums: the head's own partials (ws[0 .. kMaxBlocks * kMaxK)) are dead once its finalize kernel ran, but its
  // coefficients `scal` right behind them (and the Mumford-Shah moments behind those) must survive until the last pass -- small
  // problems re-use the head's region, large ones (more TV tiles than it holds) the rest of the workspace behind the moments.
  // Workspace layout (wsl_loss_ws_bytes, floats): [part: max(tiles, kMaxBlocks) * kMaxK][scal: 64][mom: N * chunks * (kMaxC + 1)][CRF /
  // large-layout regulariser partials: 2 * N * (HW / 256 + 64)] -- the bound is checked below, not assumed
  const int chunks = cdiv(HW, 4096);
  float* mom = scal + 64;
  const size_t rneed = (size_t)N * C * cdiv(W, 16) * cdiv(H, 16) + (size_t)N * chunks * 2 + kMaxBlocks;
  float* rpart = rneed <= (size_t)kMaxBlocks * kMaxK ? part : mom + (size_t)N * chunks * (kMaxC + 1);
  WSL_REQUIRE((size_t)(rpart - part) + rneed <= ws_bytes / sizeof(float), "head_reg_fwd_bwd: workspace too small for the regulariser's partials");
  {   // pass 1: softmax (kept in s), partial-CE sums -> the head's coefficients
    ProfScope ps(PF_LOSS_HEAD, 0.0, (double)N * HW * (4.0 * C + 1.0 + 4.0 * C), stream);
    head_stage1(h, nullptr, s, 0.f, out, C, N, 0, part, scal, nb, stream);
  }
  {   // regulariser on s: value -> out[4], weighted gradient -> ds
    ProfScope ps(PF_LOSS_HEAD, 0.0, (double)N * HW * 8.0 * C, stream);
    if (reg_kind == WSL_REG_TV) {
      WSL_REQUIRE(N > 1, "head_reg_fwd_bwd: tv_loss(outputs_soft[1:]) needs a batch of at least two");
      const double numel = (double)(N - 1) * C * H * W;
      TvP q{s, ds, 1, N, C, H, W, cdiv(W, 16), cdiv(H, 16), (float)(reg_weight / numel)};
      const int nt = N * C * q.tiles_x * q.tiles_y;
      WSL_LAUNCH(tv_fwd_bwd_kernel, dim3(nt), dim3(kThreads), 0, stream, q, rpart);
      WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, rpart, nt, 1, 1.0 / numel, out + 4);
    } else if (reg_kind == WSL_REG_MS) {
      WSL_LAUNCH(ms_moment_kernel, dim3(chunks, N), dim3(kThreads), 0, stream, img, s, C, HW, chunks, mom);
      WSL_LAUNCH(ms_main_kernel, dim3(chunks, N), dim3(kThreads), 0, stream, img, s, mom, C, H, W, chunks, reg_weight, ds, rpart);
      WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, rpart, N * chunks, 2, 1.0, out + 4);
    } else {
      const double norm = 1.0 / ((double)h.P * log((double)C));
      WSL_LAUNCH(entropy_kernel, dim3(nb), dim3(kThreads), 0, stream, s, C, HW, h.P, (float)(reg_weight * norm), ds, rpart);
      WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, rpart, nb, 1, norm, out + 4);
    }
    if (zt) {   // consistency with a teacher's logits: mean((softmax(z) - softmax(zt))^2), gradient added to ds
      const double numel = (double)N * C * HW;
      WSL_LAUNCH(softmax_mse_ds_kernel, dim3(nb), dim3(kThreads), 0, stream, s, zt, C, HW, h.P, (float)(cons_weight / numel), ds, rpart);
      WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, rpart, nb, 1, 1.0 / numel, out + 5);
    } else {
      WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, rpart, 0, 1, 0.0, out + 5);   // no teacher: out[5] = 0, never stale
    }
  }
  {   // pass 2: dz = w_ce * dCE/dz + softmax_backward(s, ds) -- written once
    ProfScope ps(PF_LOSS_HEAD, 0.0, (double)N * HW * (4.0 * C + 1.0 + 4.0 * C + 4.0 * C), stream);
    head_stage2(h, scal, w_ce, dz, nullptr, ds, 1.f, C, nb, stream);
  }
  return check_launch("head_reg_fwd_bwd");
}

extern "C" int wsl_tv_fwd_bwd(const float* p, int n0, float* loss, float* dp, float gscale, int N, int C, int H, int W,
                              void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(p && loss && dp && N > 0 && C > 0 && H > 0 && W > 0 && n0 >= 0 && n0 < N, "tv_fwd_bwd: bad args");
  const int HW_ = H * W;
  WSL_WS_OK("tv_fwd_bwd");
  const double numel = (double)(N - n0) * C * H * W;
  TvP q{p, dp, n0, N, C, H, W, cdiv(W, 16), cdiv(H, 16), (float)(gscale / numel)};
  const int nb = N * C * q.tiles_x * q.tiles_y;
  float* part = static_cast<float*>(ws);
  WSL_LAUNCH(tv_fwd_bwd_kernel, dim3(nb), dim3(kThreads), 0, stream, q, part);
  WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, 1, 1.0 / numel, loss);
  return check_launch("tv_fwd_bwd");
}

extern "C" int wsl_mumford_shah_fwd_bwd(const float* img, const float* p, float* loss, float* dp, float gscale, int N,
                                        int C, int H, int W, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(img && p && loss && dp && N > 0 && C > 0 && C <= kMaxC && H > 0 && W > 0, "mumford_shah: bad args");
  const int HW_ = H * W;
  WSL_WS_OK("mumford_shah_fwd_bwd");
  const int chunks = cdiv(H * W, 4096);
  float* part = static_cast<float*>(ws);
  float* mom = part + (size_t)kMaxBlocks * kMaxK + 64;
  if ((size_t)N * chunks * 2 > (size_t)kMaxBlocks * kMaxK) {
    set_error("mumford_shah: N*chunks too large for the partial buffer");
    return WSL_EUNSUPPORTED;
  }
  WSL_LAUNCH(ms_moment_kernel, dim3(chunks, N), dim3(kThreads), 0, stream, img, p, C, H * W, chunks, mom);
  WSL_LAUNCH(ms_main_kernel, dim3(chunks, N), dim3(kThreads), 0, stream, img, p, mom, C, H, W, chunks, gscale, dp, part);
  WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, N * chunks, 2, 1.0, loss);
  return check_launch("mumford_shah_fwd_bwd");
}

extern "C" int wsl_softmax_mse_fwd_bwd(const float* a, const float* b, float* loss, float* da, float gscale, int N, int C,
                                       int HW, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(a && b && loss && da && N > 0 && C > 0 && C <= kMaxC && HW > 0, "softmax_mse: bad args");
  const int HW_ = HW;
  WSL_WS_OK("softmax_mse_fwd_bwd");
  const int64_t P = (int64_t)N * HW;
  const double numel = (double)P * C;
  const int nb = grid_for(P);
  float* part = static_cast<float*>(ws);
  WSL_LAUNCH(softmax_mse_kernel, dim3(nb), dim3(kThreads), 0, stream, a, b, C, HW, P, (float)(gscale / numel), da, part);
  WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, 1, 1.0 / numel, loss);
  return check_launch("softmax_mse_fwd_bwd");
}

extern "C" int wsl_rot90(const float* x, float* y, int planes, int H, int W, int k, void* stream) {
  WSL_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && x != y, "rot90: bad args");
  WSL_LAUNCH(rot90_kernel, dim3(cdiv(H * W, kThreads * 4), planes), dim3(kThreads), 0, stream, x, y, H, W, ((k % 4) + 4) % 4);
  return check_launch("rot90_kernel");
}

extern "C" int wsl_softmax_accum(const float* z, float* acc, float scale, int init, int N, int C, int HW, void* stream) {
  WSL_REQUIRE(z && acc && N > 0 && C > 0 && C <= kMaxC && HW > 0, "softmax_accum: bad args");
  const int64_t P = (int64_t)N * HW;
  WSL_LAUNCH(softmax_accum_kernel, dim3(grid_for(P)), dim3(kThreads), 0, stream, z, acc, scale, init, C, HW, P);
  return check_launch("softmax_accum_kernel");
}

extern "C" int wsl_ustm_consistency_fwd_bwd(const float* a, const float* b, const float* pmean, float threshold, float* loss,
                                            float* da, float gscale, int N, int C, int HW, void* ws, size_t ws_bytes,
                                            void* stream) {
  WSL_REQUIRE(a && b && pmean && loss && da && N > 0 && C > 0 && C <= kMaxC && HW > 0, "ustm_consistency: bad args");
  const int HW_ = HW;
  WSL_WS_OK("ustm_consistency_fwd_bwd");
  const int64_t P = (int64_t)N * HW;
  const int nb = grid_for(P);
  float* part = static_cast<float*>(ws);
  WSL_LAUNCH(ustm_reduce_kernel, dim3(nb), dim3(kThreads), 0, stream, a, b, pmean, threshold, C, HW, P, part);
  WSL_LAUNCH(ustm_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, loss);
  WSL_LAUNCH(ustm_bwd_kernel, dim3(nb), dim3(kThreads), 0, stream, a, b, pmean, threshold, loss, gscale, C, HW, P, da);
  return check_launch("ustm_consistency_fwd_bwd");
}

extern "C" int wsl_entropy_fwd_bwd(const float* p, float* loss, float* dp, float gscale, int N, int C, int HW,
                                   int norm_classes, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(p && loss && dp && N > 0 && C > 0 && C <= kMaxC && HW > 0 && norm_classes > 1, "entropy: bad args");
  const int HW_ = HW;
  WSL_WS_OK("entropy_fwd_bwd");
  const int64_t P = (int64_t)N * HW;
  const double norm = 1.0 / ((double)P * log((double)norm_classes));   // losses.py:32-33: C is only the log(C) normaliser
  const int nb = grid_for(P);
  float* part = static_cast<float*>(ws);
  WSL_LAUNCH(entropy_kernel, dim3(nb), dim3(kThreads), 0, stream, p, C, HW, P, (float)(gscale * norm), dp, part);
  WSL_LAUNCH(sum_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, part, nb, 1, norm, loss);
  return check_launch("entropy_fwd_bwd");
}

extern "C" int wsl_mixprob_fwd(const float* z1, const float* z2, double beta, float* y, int N, int C, int HW, void* stream) {
  WSL_REQUIRE(z1 && y && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "mixprob_fwd: bad args");
  const int64_t P = (int64_t)N * HW;
  ProfScope ps(PF_LOSS_HEAD, 0.0, (double)P * 4.0 * C * (z2 ? 3.0 : 2.0), stream);
  if (C == 4) WSL_LAUNCH((mixprob_fwd_kernel<4>), dim3(grid_for(P)), dim3(kThreads), 0, stream, z1, z2, (float)beta, (float)(1.0 - beta), y, C, HW, P);
  else WSL_LAUNCH((mixprob_fwd_kernel<0>), dim3(grid_for(P)), dim3(kThreads), 0, stream, z1, z2, (float)beta, (float)(1.0 - beta), y, C, HW, P);
  return check_launch("mixprob_fwd_kernel");
}

extern "C" int wsl_mixprob_bwd(const float* z1, const float* z2, double beta, const float* dy, float k, float* dz1,
                               float* dz2, int accumulate, int N, int C, int HW, void* stream) {
  WSL_REQUIRE(z1 && dy && dz1 && (!z2 || dz2) && N > 0 && HW > 0 && C > 0 && C <= kMaxC, "mixprob_bwd: bad args");
  const int64_t P = (int64_t)N * HW;
  const double nbr_ = z2 ? 2.0 : 1.0;     // reads the logits and dy, reads (accumulate) and writes the logit gradients
  ProfScope ps(PF_LOSS_HEAD, 0.0, (double)P * 4.0 * C * (nbr_ + 1.0 + nbr_ * (accumulate ? 2.0 : 1.0)), stream);
  if (C == 4) WSL_LAUNCH((mixprob_bwd_kernel<4>), dim3(grid_for(P)), dim3(kThreads), 0, stream, z1, z2, (float)beta, (float)(1.0 - beta), dy, k, dz1, dz2, accumulate, C, HW, P);
  else WSL_LAUNCH((mixprob_bwd_kernel<0>), dim3(grid_for(P)), dim3(kThreads), 0, stream, z1, z2, (float)beta, (float)(1.0 - beta), dy, k, dz1, dz2, accumulate, C, HW, P);
  return check_launch("mixprob_bwd_kernel");
}

extern "C" int wsl_axpy(float* dst, const float* src, float k, int64_t n, void* stream) {
  WSL_REQUIRE(dst && src && n > 0, "axpy: bad args");
  WSL_LAUNCH(axpy_kernel, dim3(grid_for(n / 4 + 1)), dim3(kThreads), 0, stream, dst, src, k, n);
  return check_launch("axpy_kernel");
}
