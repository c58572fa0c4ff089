// TEST INFRASTRUCTURE ONLY -- never part of the product.
//
// A single-threaded lock-step emulator for the subset of HIP the kernels in
// wsl4mis_amd/csrc use, so the *same kernel sources* can be compiled for the host
// (clang++ -x c++ -DWSL_HOST_EMUL) and their indexing / tiling / reduction logic
// checked against the oracle on a machine with no GPU.  One workgroup runs at a time;
// each thread of the workgroup is a fiber (hand-rolled x86-64 context switch, no
// syscalls); __syncthreads(), wave shuffles and MFMA are rendezvous points.  MFMA is
// modelled with the gfx950 operand/accumulator lane maps documented in
// /opt/skills/guides/cdna_hip_programming.md section 3 (k-ordered fmaf chain).
//
// The package loader (wsl4mis_amd/_lib.py) only ever loads the hipcc-built
// libwslhip.so; nothing here is reachable from the product path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local   /* one workgroup at a time per host thread */
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

namespace wsl_emu {

struct Fiber {
  void* sp = nullptr;
  bool done = false;
  unsigned long seq = 0;  // count of wave collectives this lane has executed
};

struct WaveState {
  int arrived = 0;
  unsigned long gen = 0;
  uint32_t xa[2][64];
  uint32_t xb[2][64];
};

struct State {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  std::vector<unsigned char> stacks;
  void* sched_sp = nullptr;
  int cur = -1;
  int nthreads = 0, ndone = 0;
  int bar_arrived = 0;
  unsigned long bar_gen = 0;
  unsigned long progress = 0;
  const std::function<void()>* body = nullptr;
  unsigned char* dyn_smem = nullptr;
};
State& st();
extern "C" void wsl_emu_switch(void** from_sp, void* to_sp);
void yield();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
int lane();
WaveState& wave();
void wave_sync();
unsigned char* dyn_smem();
}  // namespace wsl_emu

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline void __syncthreads() {
  auto& s = wsl_emu::st();
  unsigned long g = s.bar_gen;
  s.bar_arrived++;
  s.progress++;
  if (s.bar_arrived >= s.nthreads - s.ndone) {
    s.bar_arrived = 0;
    s.bar_gen++;
    return;
  }
  while (s.bar_gen == g) wsl_emu::yield();
}

template <typename T>
static inline uint32_t wsl_emu_bits(T v) {
  static_assert(sizeof(T) == 4, "32-bit shuffles only");
  uint32_t u;
  memcpy(&u, &v, 4);
  return u;
}
template <typename T>
static inline T wsl_emu_unbits(uint32_t u) {
  T v;
  memcpy(&v, &u, 4);
  return v;
}

template <typename T>
static inline T wsl_emu_shfl(T v, int src_lane) {
  auto& w = wsl_emu::wave();
  auto& f = wsl_emu::st().fibers[wsl_emu::st().cur];
  int buf = f.seq & 1;
  f.seq++;
  w.xa[buf][wsl_emu::lane()] = wsl_emu_bits(v);
  wsl_emu::wave_sync();
  return wsl_emu_unbits<T>(w.xa[buf][src_lane & 63]);
}
// 64-bit values travel as two 32-bit shuffles (what the device runtime does)
static inline double wsl_emu_shfl(double v, int src_lane) {
  uint64_t u;
  memcpy(&u, &v, 8);
  const uint32_t lo = wsl_emu_shfl<uint32_t>((uint32_t)u, src_lane), hi = wsl_emu_shfl<uint32_t>((uint32_t)(u >> 32), src_lane);
  u = ((uint64_t)hi << 32) | lo;
  memcpy(&v, &u, 8);
  return v;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int = 64) { return wsl_emu_shfl(v, wsl_emu::lane() ^ mask); }
template <typename T>
static inline T __shfl_down(T v, int d, int = 64) {
  int l = wsl_emu::lane();
  return wsl_emu_shfl(v, l + d < 64 ? l + d : l);
}
template <typename T>
static inline T __shfl(T v, int src, int = 64) { return wsl_emu_shfl(v, src); }

struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct alignas(4) uchar4 { unsigned char x, y, z, w; };
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }
typedef float wsl_v4f __attribute__((ext_vector_type(4)));
typedef float wsl_v16f __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D col=l&15,row=(l>>4)*4+r.
static inline wsl_v4f wsl_emu_mfma16(float a, float b, wsl_v4f c) {
  auto& w = wsl_emu::wave();
  auto& f = wsl_emu::st().fibers[wsl_emu::st().cur];
  int buf = f.seq & 1;
  f.seq++;
  int l = wsl_emu::lane();
  w.xa[buf][l] = wsl_emu_bits(a);
  w.xb[buf][l] = wsl_emu_bits(b);
  wsl_emu::wave_sync();
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k)
      acc = fmaf(wsl_emu_unbits<float>(w.xa[buf][k * 16 + row]), wsl_emu_unbits<float>(w.xb[buf][k * 16 + col]), acc);
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products; layout probed on the device by tools/probe_mfma4.py):
// block = l >> 2;  D[l][r] = A[4 * block + r] * B[l] + C[l][r]   (row r of the block comes from lane 4*block + r).
static inline wsl_v4f wsl_emu_mfma4(float a, float b, wsl_v4f c) {
  auto& w = wsl_emu::wave();
  auto& f = wsl_emu::st().fibers[wsl_emu::st().cur];
  int buf = f.seq & 1;
  f.seq++;
  int l = wsl_emu::lane();
  w.xa[buf][l] = wsl_emu_bits(a);
  wsl_emu::wave_sync();
  for (int r = 0; r < 4; ++r) c[r] = fmaf(wsl_emu_unbits<float>(w.xa[buf][(l & ~3) + r]), b, c[r]);
  return c;
}
// global_load_lds_dwordx4: lane l's 16 bytes land at the wave's LDS base + 16 * l
static inline void wsl_emu_lds_dma16(const void* gsrc, void* lds_wave_base) {
  memcpy((char*)lds_wave_base + 16 * wsl_emu::lane(), gsrc, 16);
}
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5).
static inline wsl_v16f wsl_emu_mfma32(float a, float b, wsl_v16f c) {
  auto& w = wsl_emu::wave();
  auto& f = wsl_emu::st().fibers[wsl_emu::st().cur];
  int buf = f.seq & 1;
  f.seq++;
  int l = wsl_emu::lane();
  w.xa[buf][l] = wsl_emu_bits(a);
  w.xb[buf][l] = wsl_emu_bits(b);
  wsl_emu::wave_sync();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k)
      acc = fmaf(wsl_emu_unbits<float>(w.xa[buf][k * 32 + row]), wsl_emu_unbits<float>(w.xb[buf][k * 32 + col]), acc);
    c[r] = acc;
  }
  return c;
}

template <typename T>
static inline T atomicAdd(T* p, T v) {
  T o = *p;
  *p = o + v;
  return o;
}
static inline float __fmul_rn(float a, float b) { return a * b; }  // emulator TU is built with -ffp-contract=off
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fdividef(float a, float b) { return a / b; }

static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emul"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return 0;
}
#define hipMemcpyDeviceToDevice 3
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
  memmove(d, s, n);
  return 0;
}
