// nn.ConvTranspose2d(C1, C2, kernel_size=2, stride=2): the non-bilinear branch of the reference's UpBlock
// (ref: networks/unet.py:58-60, 63-68; SURVEY 8f rank 4 -- opt-in: the reference's Decoder never selects it).
//   out[n][co][2i+a][2j+b] = bias[co] + sum_ci x[n][ci][i][j] * w[ci][co][a][b]
// Stride == kernel size, so every output pixel has exactly one input pixel: four independent 1x1 convolutions scattered to
// the four sub-positions.  HBM-bound, FLOP-light (8 * C1 * C2 per input pixel): plain fp32 vector kernels, a thread owns one
// input pixel and four channels (16 accumulators), weights broadcast from LDS.  Deterministic (order-fixed sums, no atomics).
#include "wsl_rt.h"

namespace wsl {

constexpr int kCtMaxC = 256;

// forward: grid (pixel chunks, cdiv(Co, 4), N)
__global__ __launch_bounds__(256) void convt2x2_fwd_kernel(const float* x, const float* w, const float* bias, float* out, int Ci,
                                                           int Co, int h, int wd) {
  __shared__ float wl[kCtMaxC * 16];   // [ci][co4][ab]
  const int co0 = blockIdx.y * 4, n = blockIdx.z, hw = h * wd;
  for (int e = threadIdx.x; e < Ci * 16; e += kThreads) {
    const int ci = e >> 4, c = (e >> 2) & 3, ab = e & 3;
    wl[e] = co0 + c < Co ? w[((int64_t)ci * Co + co0 + c) * 4 + ab] : 0.f;
  }
  __syncthreads();
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  float acc[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float b = (bias && co0 + c < Co) ? bias[co0 + c] : 0.f;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) acc[c][ab] = b;
  }
  const float* xp = x + (int64_t)n * Ci * hw + p;
  for (int ci = 0; ci < Ci; ++ci) {
    const float xv = xp[(int64_t)ci * hw];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[c][ab] = fmaf(xv, wl[ci * 16 + c * 4 + ab], acc[c][ab]);
  }
  const int i = p / wd, j = p - i * wd, Wo = 2 * wd;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (co0 + c < Co) {
      float* o = out + (((int64_t)n * Co + co0 + c) * 2 * h + 2 * i) * Wo + 2 * j;
      *reinterpret_cast<float2*>(o) = make_float2(acc[c][0], acc[c][1]);
      *reinterpret_cast<float2*>(o + Wo) = make_float2(acc[c][2], acc[c][3]);
    }
  }
}

// data gradient: dx[n][ci][i][j] = sum_co sum_ab dy[n][co][2i+a][2j+b] * w[ci][co][a][b];  grid (pixel chunks, cdiv(Ci, 4), N)
__global__ __launch_bounds__(256) void convt2x2_dgrad_kernel(const float* dy, int64_t dy_bs, const float* w, float* dx, int Ci,
                                                             int Co, int h, int wd) {
  __shared__ float wl[kCtMaxC * 16];   // [co][ci4][ab]
  const int ci0 = blockIdx.y * 4, n = blockIdx.z, hw = h * wd, Wo = 2 * wd;
  for (int e = threadIdx.x; e < Co * 16; e += kThreads) {
    const int co = e >> 4, c = (e >> 2) & 3, ab = e & 3;
    wl[e] = ci0 + c < Ci ? w[((int64_t)(ci0 + c) * Co + co) * 4 + ab] : 0.f;
  }
  __syncthreads();
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const int i = p / wd, j = p - i * wd;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* g = dy + n * dy_bs + (int64_t)(2 * i) * Wo + 2 * j;
  for (int co = 0; co < Co; ++co) {
    const float* gc = g + (int64_t)co * 4 * hw;
    const float2 t = *reinterpret_cast<const float2*>(gc), b = *reinterpret_cast<const float2*>(gc + Wo);
    const float v[4] = {t.x, t.y, b.x, b.y};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) acc[c] = fmaf(v[ab], wl[co * 16 + c * 4 + ab], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (ci0 + c < Ci) dx[((int64_t)n * Ci + ci0 + c) * hw + p] = acc[c];
}

// weight gradient, stage 1: per sample partial sums.  grid (cdiv(Ci, 4), cdiv(Co, 4), N); part_w[n][Ci][Co][4], part_b[n][Co]
__global__ __launch_bounds__(256) void convt2x2_wgrad_kernel(const float* x, const float* dy, int64_t dy_bs, float* part_w,
                                                             float* part_b, int Ci, int Co, int h, int wd) {
  __shared__ float red[4];
  const int ci0 = blockIdx.x * 4, co0 = blockIdx.y * 4, n = blockIdx.z, hw = h * wd, Wo = 2 * wd;
  float acc[4][4][4], bs[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bs[a] = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.f;
  }
  for (int p = threadIdx.x; p < hw; p += kThreads) {
    const int i = p / wd, j = p - i * wd;
    float xv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) xv[a] = ci0 + a < Ci ? x[((int64_t)n * Ci + ci0 + a) * hw + p] : 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (co0 + b < Co) {
        const float* gc = dy + n * dy_bs + (int64_t)(co0 + b) * 4 * hw + (int64_t)(2 * i) * Wo + 2 * j;
        const float2 t = *reinterpret_cast<const float2*>(gc), u = *reinterpret_cast<const float2*>(gc + Wo);
        const float v[4] = {t.x, t.y, u.x, u.y};
        bs[b] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[a][b][c] = fmaf(xv[a], v[c], acc[a][b][c]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float s = block_sum(acc[a][b][c], red);
        if (threadIdx.x == 0 && ci0 + a < Ci && co0 + b < Co) part_w[(((int64_t)n * Ci + ci0 + a) * Co + co0 + b) * 4 + c] = s;
      }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float s = block_sum(bs[b], red);
      if (threadIdx.x == 0 && co0 + b < Co) part_b[(int64_t)n * Co + co0 + b] = s;
    }
  }
}

// stage 2: sum the per-sample partials in sample order
__global__ __launch_bounds__(256) void convt2x2_wreduce_kernel(const float* part_w, const float* part_b, float* dw, float* db, int N,
                                                               int64_t E, int Co) {
  const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (e < E) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += part_w[n * E + e];
    dw[e] = s;
  } else if (db && e < E + Co) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += part_b[(int64_t)n * Co + (e - E)];
    db[e - E] = s;
  }
}

}  // namespace wsl

using namespace wsl;

extern "C" int wsl_convt2x2_fwd(const float* x, const float* w, const float* bias, float* out, int N, int Ci, int Co, int h, int wd,
                                void* stream) {
  WSL_REQUIRE(x && w && out && N > 0 && Ci > 0 && Ci <= kCtMaxC && Co > 0 && h > 0 && wd > 0, "convt2x2_fwd: bad arguments (Ci <= 256)");
  WSL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7) == 0, "convt2x2_fwd: out must be 8-byte aligned");
  WSL_LAUNCH(convt2x2_fwd_kernel, dim3(cdiv(h * wd, kThreads), cdiv(Co, 4), N), dim3(kThreads), 0, stream, x, w, bias, out, Ci, Co, h,
             wd);
  return check_launch("convt2x2_fwd_kernel");
}

extern "C" int wsl_convt2x2_dgrad(const float* dy, int64_t dy_bs, const float* w, float* dx, int N, int Ci, int Co, int h, int wd,
                                  void* stream) {
  WSL_REQUIRE(dy && w && dx && N > 0 && Ci > 0 && Co > 0 && Co <= kCtMaxC && h > 0 && wd > 0, "convt2x2_dgrad: bad arguments (Co <= 256)");
  WSL_REQUIRE(dy_bs >= (int64_t)Co * 4 * h * wd && (dy_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0,
              "convt2x2_dgrad: dy batch stride too small / not 8-byte aligned");
  WSL_LAUNCH(convt2x2_dgrad_kernel, dim3(cdiv(h * wd, kThreads), cdiv(Ci, 4), N), dim3(kThreads), 0, stream, dy, dy_bs, w, dx, Ci, Co,
             h, wd);
  return check_launch("convt2x2_dgrad_kernel");
}

extern "C" size_t wsl_convt2x2_wgrad_ws_bytes(int N, int Ci, int Co) {
  return N > 0 && Ci > 0 && Co > 0 ? sizeof(float) * (size_t)N * ((size_t)Ci * Co * 4 + Co) : 0;
}

extern "C" int wsl_convt2x2_wgrad(const float* x, const float* dy, int64_t dy_bs, float* dw, float* db, int N, int Ci, int Co, int h,
                                  int wd, void* ws, size_t ws_bytes, void* stream) {
  WSL_REQUIRE(x && dy && dw && ws && N > 0 && Ci > 0 && Co > 0 && h > 0 && wd > 0, "convt2x2_wgrad: bad arguments");
  WSL_REQUIRE(dy_bs >= (int64_t)Co * 4 * h * wd && (dy_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0,
              "convt2x2_wgrad: dy batch stride too small / not 8-byte aligned");
  if (ws_bytes < wsl_convt2x2_wgrad_ws_bytes(N, Ci, Co)) {
    set_error("convt2x2_wgrad: workspace %zu < %zu", ws_bytes, wsl_convt2x2_wgrad_ws_bytes(N, Ci, Co));
    return WSL_EWORKSPACE;
  }
  float* part_w = static_cast<float*>(ws);
  float* part_b = part_w + (size_t)N * Ci * Co * 4;
  WSL_LAUNCH(convt2x2_wgrad_kernel, dim3(cdiv(Ci, 4), cdiv(Co, 4), N), dim3(kThreads), 0, stream, x, dy, dy_bs, part_w, part_b, Ci, Co,
             h, wd);
  const int64_t E = (int64_t)Ci * Co * 4;
  WSL_LAUNCH(convt2x2_wreduce_kernel, dim3((unsigned)((E + Co + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, part_w, part_b,
             dw, db, N, E, Co);
  return check_launch("convt2x2_wgrad");
}
