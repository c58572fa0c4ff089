// Machine probe (tools only), round 5: what does ONE non-matrix instruction cost next to a stream of v_mfma_f32_16x16x4_f32?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_fill.hip -o tools/exp/probe_mfma_fill && tools/exp/probe_mfma_fill
// 512 workgroups x 4 waves, 2 workgroups per CU = 2 waves per SIMD (the conv kernels' residency).  Every wave runs the same loop:
// 16 independent accumulators, K fillers of one kind behind every MFMA.  Reported: matrix TFLOP/s and cycles per MFMA per SIMD at
// 2.4 GHz (32 = the pipe's own rate), i.e. (cycles - 32) / K = what one filler costs in matrix-pipe time.
//   kinds: 0 none | 1 v_pk_add_f32 | 2 v_add_f32 | 3 ds_read_b64 (conflict-free, waited for once per 16 MFMAs) | 4 s_add_u32 (SALU)
//          5 s_nop 0 | 6 s_waitcnt lgkmcnt(0) with nothing outstanding | 7 v_mov_b32 | 8 ds_read_b64 + its own s_waitcnt lgkmcnt(0)
//          9 s_barrier (whole workgroup, once per 16 MFMAs, K ignored)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int KIND, int K>
__global__ __launch_bounds__(256, 2) void k_fill(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds[i] = (float)(i & 15);
  __syncthreads();
  v4f acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  float a = 1.f + lane * 0.001f, b = 0.5f - lane * 0.002f;
  v2f pa = {a, b}, pb = {b, a}, pc = {0.f, 0.f};
  float fa = a, fb = b;
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 8u * lane;
  unsigned sacc = 0;
  v2f r = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pc) : "v"(pa), "v"(pb));
        if (KIND == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(fa) : "v"(fa), "v"(fb));
        if (KIND == 3) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(la), "n"(((0) & 7) * 512) : "memory");
        if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (KIND == 5) asm volatile("s_nop 0");
        if (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(fa) : "v"(fb));
        if (KIND == 8) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(la) : "memory");
      }
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 9) __builtin_amdgcn_s_barrier();
  }
  float t = pc[0] + pc[1] + fa + r[0] + r[1] + (float)sacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 123.456f) sink[0] = t;
}

template <int KIND, int K>
static void run(const char* name, float* sink) {
  const int iters = 2000, blocks = 512;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_fill<KIND, K>), dim3(blocks), dim3(256), 0, 0, sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double mfma_per_simd = (double)iters * 16 * 2;   // 2 waves per SIMD
  const double cyc = best * 1e-3 * 2.4e9 / mfma_per_simd;
  const double tf = (double)blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 4 / (best * 1e-3) / 1e12;
  printf("  %-46s K=%2d: %7.3f ms  %6.1f TFLOP/s  %5.1f cycles per MFMA per SIMD", name, K, best, tf, cyc);
  if (K > 0 && KIND != 9) printf("  -> %.1f cycles per filler", (cyc - 32.0) / K);
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs; 2 waves per SIMD, every wave: MFMA + K fillers, 16 independent accumulators\n", prop.gcnArchName, prop.multiProcessorCount);
  float* sink;
  CK(hipMalloc(&sink, 64));
  run<0, 0>("MFMA only", sink);
  run<1, 1>("v_pk_add_f32", sink);
  run<1, 2>("v_pk_add_f32", sink);
  run<1, 4>("v_pk_add_f32", sink);
  run<2, 1>("v_add_f32", sink);
  run<2, 2>("v_add_f32", sink);
  run<2, 4>("v_add_f32", sink);
  run<7, 2>("v_mov_b32", sink);
  run<3, 1>("ds_read_b64 (waited for per 16 MFMAs)", sink);
  run<3, 2>("ds_read_b64 (waited for per 16 MFMAs)", sink);
  run<3, 4>("ds_read_b64 (waited for per 16 MFMAs)", sink);
  run<8, 1>("ds_read_b64 + s_waitcnt lgkmcnt(0)", sink);
  run<4, 1>("s_add_u32", sink);
  run<4, 2>("s_add_u32", sink);
  run<4, 4>("s_add_u32", sink);
  run<4, 8>("s_add_u32", sink);
  run<5, 2>("s_nop 0", sink);
  run<5, 4>("s_nop 0", sink);
  run<6, 1>("s_waitcnt lgkmcnt(0), nothing outstanding", sink);
  run<6, 2>("s_waitcnt lgkmcnt(0), nothing outstanding", sink);
  run<9, 0>("s_barrier per 16 MFMAs", sink);
  return 0;
}
