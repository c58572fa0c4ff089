// Machine probe (tools only; not part of the product library): does a STREAM of matrix instructions of one type give the same result
// when waves of the same workgroup issue matrix instructions of ANOTHER type on the same SIMDs?  (profiles/r4_sp_root_cause.md:
// an f32-MFMA kernel is wrong whenever another kernel's f16 MFMAs share the CU; this separates "two instruction types on one
// SIMD" from "two kernels on one CU".)
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_mix.hip -o tools/exp/probe_mfma_mix && tools/exp/probe_mfma_mix
// Waves 0-3 of a workgroup are victims: 64 matrix instructions on 4 accumulators (each accumulator revisited every 4th instruction,
// as in the conv kernels), inputs from the lane id, results stored.  Waves 4.. are aggressors issuing their type back to back until
// the victims are done.  The victims' stored results of a loaded launch are compared bit for bit with those of a launch without aggressors.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

enum { T_F32_16 = 0, T_F16_16 = 1, T_F32_32 = 2, T_F16_32 = 3, T_F16_16K16 = 4 };
static const char* kNames[] = {"f32_16x16x4", "f16_16x16x32", "f32_32x32x2", "f16_32x32x16", "f16_16x16x16"};

template <int T>
__device__ __forceinline__ void stream64(float seed, float* out) {
  if constexpr (T == T_F32_16) {
    v4f acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = v4f{seed + k, seed * 0.5f, 1.f, -seed};
    float a = 0.5f + seed * 0.01f, b = 1.25f - seed * 0.02f;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + 0.125f * (i >> 2), b, acc[i & 3], 0, 0, 0);
    for (int k = 0; k < 4; ++k)
      for (int r = 0; r < 4; ++r) out[k * 4 + r] = acc[k][r];
  } else if constexpr (T == T_F16_16 || T == T_F16_16K16) {
    v4f acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = v4f{seed + k, seed * 0.5f, 1.f, -seed};
    h8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.5f + seed * 0.01f + e * 0.0625f), b[e] = (_Float16)(1.25f - e * 0.03125f);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if constexpr (T == T_F16_16) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 3], 0, 0, 0);
      else acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4{a[0], a[1], a[2], a[3]}, h4{b[0], b[1], b[2], b[3]}, acc[i & 3], 0, 0, 0);
    }
    for (int k = 0; k < 4; ++k)
      for (int r = 0; r < 4; ++r) out[k * 4 + r] = acc[k][r];
  } else if constexpr (T == T_F32_32) {
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed + r;
    float a = 0.5f + seed * 0.01f, b = 1.25f - seed * 0.02f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a + 0.125f * i, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[r] = acc[r];
  } else {
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = seed + r;
    h8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.5f + seed * 0.01f + e * 0.0625f), b[e] = (_Float16)(1.25f - e * 0.03125f);
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[r] = acc[r];
  }
}

template <int A>
__device__ __forceinline__ void aggress(volatile int* done) {
  float sink[16];
  for (int guard = 0; guard < 20000 && *done < 4; ++guard) stream64<A>(threadIdx.x * 0.001f + guard, sink);
  if (sink[0] == 123.456f) *done = 7;
}

template <int V, int A>
__global__ void mix_kernel(float* out, int iters) {
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (w >= 4) { aggress<A>(&done); return; }
  float r[16], s[16];
  for (int k = 0; k < 16; ++k) s[k] = 0.f;
  for (int it = 0; it < iters; ++it) {
    stream64<V>(l * 0.25f + it, r);
    for (int k = 0; k < 16; ++k) s[k] += r[k] * (1.f / 1024.f);   // (plain vector math after the compiler's own wait states)
  }
  float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  for (int k = 0; k < 16; ++k) o[k] = s[k];
  if (l == 0) atomicAdd(&done, 1);
}

typedef void (*Kern)(float*, int);
template <int V, int A> static Kern get() { return mix_kernel<V, A>; }
template <int V> static Kern getA(int a) {
  switch (a) { case 0: return get<V, 0>(); case 1: return get<V, 1>(); case 2: return get<V, 2>(); case 3: return get<V, 3>(); default: return get<V, 4>(); }
}
static Kern getVA(int v, int a) {
  switch (v) { case 0: return getA<0>(a); case 1: return getA<1>(a); case 2: return getA<2>(a); case 3: return getA<3>(a); default: return getA<4>(a); }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const size_t n = (size_t)cus * 256 * 16;
  float* d;
  (void)hipMalloc(&d, n * 4);
  std::vector<float> ref(n), got(n);
  printf("# %s, %d CUs; victim type (waves 0-3, %d x 64 instructions) x aggressor type (1 / 3 wave quads): words wrong of %zu\n", prop.gcnArchName, cus, iters, n);
  for (int v = 0; v < 5; ++v) {
    (void)hipMemset(d, 0, n * 4);
    hipLaunchKernelGGL(getVA(v, 0), dim3(cus), dim3(256), 0, 0, d, iters);   // no aggressor waves
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    (void)hipMemcpy(ref.data(), d, n * 4, hipMemcpyDeviceToHost);
    for (int a = 0; a < 5; ++a) {
      printf("victim %-13s aggressor %-13s", kNames[v], kNames[a]);
      for (int q : {1, 3}) {
        (void)hipMemset(d, 0, n * 4);
        hipLaunchKernelGGL(getVA(v, a), dim3(cus), dim3(256 * (1 + q)), 0, 0, d, iters);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
        (void)hipMemcpy(got.data(), d, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
        printf("  x%d: %8zu", q, bad);
      }
      printf("\n");
      fflush(stdout);
    }
  }
  return 0;
}
