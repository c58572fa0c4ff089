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

}  // namespace wsl

using namespace wsl;

extern "C" int wsl_sgd_step(float* p, const float* grad, float* buf, int64_t n, float lr, float momentum, float wd,
                            int first, float grad_scale, float* ema, float ema_alpha, void* stream) {
  WSL_REQUIRE(p && grad && buf && n > 0, "sgd_step: bad args");
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = al(p) && al(grad) && al(buf) && (!ema || al(ema));
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
