// SGD(momentum, weight_decay) + optional EMA over a flat fp32 arena -- one multi-tensor launch per step
// (ref: train_weakly_supervised_segmentation_pCE_ours_proposed.py:89-90,126-132; EMA: ..._ustm_2D.py:61-65).
#include "wsl_rt.h"

namespace wsl {

template <bool VEC>
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* buf, int64_t n, float lr, float mu,
                                                  float wd, int first, float gs, float* ema, float ea) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  if (VEC) {
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* b4 = reinterpret_cast<float4*>(buf);
    float4* e4 = reinterpret_cast<float4*>(ema);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
      float4 pv = p4[i], gv = g4[i], bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : b4[i];
      float* pp = &pv.x;
      const float* gp = &gv.x;
      float* bp = &bv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = fmaf(wd, pp[k], gp[k] * gs);
        bp[k] = first ? gk : fmaf(mu, bp[k], gk);
        pp[k] = fmaf(-lr, bp[k], pp[k]);
      }
      p4[i] = pv;
      b4[i] = bv;
      if (ema) {
        float4 ev = e4[i];
        float* ep = &ev.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) ep[k] = fmaf(ea, ep[k], (1.f - ea) * pp[k]);
        e4[i] = ev;
      }
    }
  }
  const int64_t start = VEC ? (n & ~(int64_t)3) : 0;
  for (int64_t i = start + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const float gk = fmaf(wd, p[i], g[i] * gs);
    const float b = first ? gk : fmaf(mu, buf[i], gk);
    buf[i] = b;
    const float pn = fmaf(-lr, b, p[i]);
    p[i] = pn;
    if (ema) ema[i] = fmaf(ea, ema[i], (1.f - ea) * pn);
  }
}

// ------------------------------------------------------------------------------------------------ dropout masks
// Philox4x32-10 counter-based generator (Salmon et al. 2011): 4 x 32 random bits per (key, counter); element e of mask
// m uses counter (e / 4, m) -- reproducible for a given seed, independent of grid shape.
__device__ __forceinline__ void philox4(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1, c3 = (uint32_t)p0, c0 = n0, c2 = n2;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

struct MaskTable {
  int n;
  struct {
    void* out;        // uint8 keep mask (is_f32 == 0) or float multiplier (is_f32 == 1)
    int64_t numel;
    uint32_t thresh;  // keep iff u32 < thresh  (thresh = keep_prob * 2^32)
    float scale;      // multiplier written for kept entries of a float mask
    int is_f32, _pad;
  } e[12];
};

__global__ __launch_bounds__(256) void masks_kernel(MaskTable t, uint32_t k0, uint32_t k1) {
  const int m = blockIdx.y;
  const uint32_t th = t.e[m].thresh;
  if (!t.e[m].is_f32) {
    // uint8 keep masks (one byte per activation element, 130 MB per step at N = 64): SIXTEEN random bits per element, eight elements per
    // Philox call -- the generator's integer multiplies are what this kernel is made of (68 -> ~36 us per step); the keep probability is
    // quantised to 1 / 65536 (0.95 -> 0.949997), far inside the sampling noise of any mask
    const int64_t n8 = (t.e[m].numel + 7) >> 3;
    // rounded to the nearest 1 / 65536 and able to reach 65536 (keep_prob 1.0 keeps everything, 0.0 nothing: ADVICE r5 -- flooring
    // dropped one element in 65536 at keep_prob 1.0 and biased every keep rate down by up to 1.5e-5)
    const uint32_t th16 = th == 0xFFFFFFFFu ? 0x10000u : (uint32_t)(((uint64_t)th + 0x8000u) >> 16);
    uint8_t* o = static_cast<uint8_t*>(t.e[m].out);
    for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n8; q += (int64_t)gridDim.x * kThreads) {
      uint32_t r[4];
      philox4(k0, k1, (uint32_t)q, (uint32_t)(q >> 32), (uint32_t)m, 0x57534c38u, r);
      const int64_t e = q << 3;
      uint32_t lo = 0, hi = 0;   // bytes 0-3 and 4-7
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        lo |= ((r[k] & 0xffffu) < th16 ? 1u : 0u) << (16 * k) | ((r[k] >> 16) < th16 ? 1u : 0u) << (16 * k + 8);
        hi |= ((r[2 + k] & 0xffffu) < th16 ? 1u : 0u) << (16 * k) | ((r[2 + k] >> 16) < th16 ? 1u : 0u) << (16 * k + 8);
      }
      if (e + 7 < t.e[m].numel && (reinterpret_cast<uintptr_t>(o) & 7) == 0) {
        *reinterpret_cast<uint64_t*>(o + e) = (uint64_t)lo | ((uint64_t)hi << 32);
      } else {
        for (int k = 0; k < 8; ++k)
          if (e + k < t.e[m].numel) o[e + k] = (uint8_t)(((k < 4 ? lo : hi) >> (8 * (k & 3))) & 1u);
      }
    }
    return;
  }
  const int64_t n4 = (t.e[m].numel + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n4; q += (int64_t)gridDim.x * kThreads) {
    uint32_t r[4];
    philox4(k0, k1, (uint32_t)q, (uint32_t)(q >> 32), (uint32_t)m, 0x57534c34u, r);
    const int64_t e = q << 2;
    if (t.e[m].is_f32) {
      float* o = static_cast<float*>(t.e[m].out);
      for (int k = 0; k < 4; ++k)
        if (e + k < t.e[m].numel) o[e + k] = r[k] < th ? t.e[m].scale : 0.f;
    } else {
      uint8_t* o = static_cast<uint8_t*>(t.e[m].out);
      if (e + 3 < t.e[m].numel) {
        *reinterpret_cast<uchar4*>(o + e) = make_uchar4(r[0] < th, r[1] < th, r[2] < th, r[3] < th);
      } else {
        for (int k = 0; k < 4; ++k)
          if (e + k < t.e[m].numel) o[e + k] = r[k] < th;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ teacher input noise
// out[r * n + i] = x[i] + (noise ? noise[r * n + i] : clamp(N(0, 1) * sigma, -clip, clip)),  r < reps
// (ref: train_mean_teacher_2D.py:147-149 / ..._ustm_2D.py:125-127,133: `torch.clamp(torch.randn_like(x) * 0.1, -0.2, 0.2)` added
// to the batch, and `volume_batch_r.repeat(2, 1, 1, 1)` + noise for the uncertainty passes).  Normals by Box-Muller on
// Philox4x32-10 words; like the dropout masks only the distribution is part of the contract, the stream is this library's.
__global__ __launch_bounds__(256) void noisy_copy_kernel(const float* x, const float* noise, float* out, int64_t n, int reps,
                                                         float sigma, float clip, uint32_t k0, uint32_t k1) {
  const int64_t total = n * reps, n4 = (total + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < n4; q += (int64_t)gridDim.x * kThreads) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (!noise) {
      uint32_t r[4];
      philox4(k0, k1, (uint32_t)q, (uint32_t)(q >> 32), 0x6e6f6973u, 0x57534c35u, r);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)r[2 * h] + 1.f) * 2.3283064365386963e-10f;      // (0, 1]
        const float u2 = (float)r[2 * h + 1] * 2.3283064365386963e-10f;
        const float rad = sqrtf(-2.f * logf(u1));
        z[2 * h] = rad * cosf(6.283185307179586f * u2), z[2 * h + 1] = rad * sinf(6.283185307179586f * u2);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t e = (q << 2) + k;
      if (e < total) {
        float d = noise ? noise[e] : z[k] * sigma;
        if (!noise) d = d < -clip ? -clip : (d > clip ? clip : d);
        out[e] = x[e % n] + d;
      }
    }
  }
}

}  // namespace wsl

using namespace wsl;

extern "C" int wsl_noisy_copy(const float* x, const float* noise, float* out, int64_t n, int reps, float sigma, float clip,
                              uint64_t seed, void* stream) {
  WSL_REQUIRE(x && out && n > 0 && reps > 0 && sigma >= 0.f && clip >= 0.f, "noisy_copy: bad arguments");
  int64_t blocks = ((n * reps + 3) / 4 + kThreads - 1) / kThreads;
  if (blocks > 4096) blocks = 4096;
  WSL_LAUNCH(noisy_copy_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, x, noise, out, n, reps, sigma, clip, (uint32_t)seed,
             (uint32_t)(seed >> 32));
  return check_launch("noisy_copy_kernel");
}

extern "C" int wsl_draw_masks(int n_masks, void* const* outs, const int64_t* numels, const float* keep_probs,
                              const float* scales, const int* is_f32, uint64_t seed, void* stream) {
  WSL_REQUIRE(n_masks > 0 && n_masks <= 12 && outs && numels && keep_probs && scales && is_f32, "draw_masks: bad args");
  MaskTable t;
  t.n = n_masks;
  int64_t biggest = 1;
  for (int i = 0; i < n_masks; ++i) {
    WSL_REQUIRE(outs[i] && numels[i] > 0 && keep_probs[i] >= 0.f && keep_probs[i] <= 1.f, "draw_masks: bad mask %d", i);
    const double th = (double)keep_probs[i] * 4294967296.0;
    t.e[i].out = outs[i], t.e[i].numel = numels[i], t.e[i].scale = scales[i], t.e[i].is_f32 = is_f32[i], t.e[i]._pad = 0;
    t.e[i].thresh = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)th;
    if (numels[i] > biggest) biggest = numels[i];
  }
  int64_t blocks = (biggest / 8 + kThreads - 1) / kThreads;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  double mbytes = 0.0;
  for (int i = 0; i < n_masks; ++i) mbytes += (double)numels[i] * (is_f32[i] ? 4.0 : 1.0);
  ProfScope ps(PF_PREP, 0.0, mbytes, stream);
  WSL_LAUNCH(masks_kernel, dim3((unsigned)blocks, n_masks), dim3(kThreads), 0, stream, t, (uint32_t)seed,
             (uint32_t)(seed >> 32));
  return check_launch("masks_kernel");
}

extern "C" int wsl_sgd_step(float* p, const float* grad, float* buf, int64_t n, float lr, float momentum, float wd,
                            int first, float grad_scale, float* ema, float ema_alpha, void* stream) {
  WSL_REQUIRE(p && grad && buf && n > 0, "sgd_step: bad args");
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = al(p) && al(grad) && al(buf) && (!ema || al(ema));
  ProfScope ps(PF_SGD, 0.0, (double)n * (20.0 + (ema ? 8.0 : 0.0)), stream);      // read p, g, buf; write p, buf (+ ema r/w)
  int64_t blocks = (n / 4 + kThreads - 1) / kThreads;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  if (vec)
    WSL_LAUNCH((sgd_kernel<true>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, grad, buf, n, lr, momentum, wd,
               first, grad_scale, ema, ema_alpha);
  else
    WSL_LAUNCH((sgd_kernel<false>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, grad, buf, n, lr, momentum, wd,
               first, grad_scale, ema, ema_alpha);
  return check_launch("sgd_kernel");
}
