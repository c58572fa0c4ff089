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
  uint32_t xw[2][2][64][4];   // wide operands (f16 MFMA: 8 halves per lane and operand; transpose read: one pointer per lane)
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

// ---- f16 pieces of the split-precision path ------------------------------------------------------------------------------
typedef uint32_t wsl_emu_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t wsl_emu_u2 __attribute__((ext_vector_type(2)));
static inline float wsl_emu_h2f(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
static inline uint16_t wsl_emu_f2h_rne(float f) {        // v_cvt_pk_f16_f32 (round to nearest even; overflow -> inf)
  const _Float16 h = (_Float16)f;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static inline uint16_t wsl_emu_f2h_rtz(float f) {        // v_cvt_pkrtz_f16_f32 (toward zero; a finite overflow saturates at 65504)
  const uint32_t u = wsl_emu_bits(f), sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (a == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);
  if (a >= 0x477fe000u) return (uint16_t)(sign | 0x7bffu);
  const int e = (int)(a >> 23) - 127;
  if (e >= -14) return (uint16_t)(sign | (uint32_t)((e + 15) << 10) | ((a >> 13) & 0x3ffu));
  const int sh = (-14 - e) + 13;
  const uint32_t m = (a & 0x7fffffu) | 0x800000u;
  return (uint16_t)(sign | (sh < 32 ? (m >> sh) : 0u));
}
// v_mfma_f32_16x16x32_f16: A[i = l & 15][k = 8 (l >> 4) + e], B[k = 8 (l >> 4) + e][j = l & 15], D col = l & 15, row = 4 (l >> 4) + r
// (operand layout probed on the device: tools/probe_sp.hip).  Products of two halves are exact in fp32; they are summed in k order.
static inline wsl_v4f wsl_emu_mfma16x32_f16(wsl_emu_u4 a, wsl_emu_u4 b, wsl_v4f c) {
  auto& w = wsl_emu::wave();
  auto& f = wsl_emu::st().fibers[wsl_emu::st().cur];
  const int buf = f.seq & 1;
  f.seq++;
  const int l = wsl_emu::lane();
  for (int q = 0; q < 4; ++q) w.xw[buf][0][l][q] = a[q], w.xw[buf][1][l][q] = b[q];
  wsl_emu::wave_sync();
  const int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      const uint32_t ua = w.xw[buf][0][(k >> 3) * 16 + row][(k & 7) >> 1], ub = w.xw[buf][1][(k >> 3) * 16 + col][(k & 7) >> 1];
      const float av = wsl_emu_h2f((uint16_t)((k & 1) ? ua >> 16 : ua)), bv = wsl_emu_h2f((uint16_t)((k & 1) ? ub >> 16 : ub));
      acc += av * bv;
    }
    c[r] = acc;
  }
  return c;
}
// ds_read_b64_tr_b16 (probed on the device: tools/probe_sp.hip): inside a group of 16 lanes, lane i receives as element e the
// (i & 3)-th half of the four contiguous halves at the address supplied by lane 4 e + (i >> 2) of the group
static inline wsl_emu_u2 wsl_emu_ds_read_tr16(const void* p) {
  auto& w = wsl_emu::wave();
  auto& f = wsl_emu::st().fibers[wsl_emu::st().cur];
  const int buf = f.seq & 1;
  f.seq++;
  const int l = wsl_emu::lane();
  const uint64_t u = (uint64_t)(uintptr_t)p;
  w.xw[buf][0][l][0] = (uint32_t)u, w.xw[buf][0][l][1] = (uint32_t)(u >> 32);
  wsl_emu::wave_sync();
  const int g = l & ~15, i = l & 15;
  uint16_t h[4];
  for (int e = 0; e < 4; ++e) {
    const int s = g + 4 * e + (i >> 2);
    const uint16_t* q = (const uint16_t*)(uintptr_t)(((uint64_t)w.xw[buf][0][s][1] << 32) | w.xw[buf][0][s][0]);
    h[e] = q[i & 3];
  }
  return wsl_emu_u2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
}
// order-independent maximum of non-negative floats through their bit patterns (workgroups of a launch run on several host threads)
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) {
  uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
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
