// Machine probe (tools only; not part of the product library) for profiles/r4_sp_root_cause.md:
//   packed f32 vector instructions (VOP3P: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) under a PARTIAL execution mask give wrong
//   results while a wave of another kernel issues f16 matrix instructions on the same CU.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_pk_exec.hip -o tools/exp/probe_pk_exec && tools/exp/probe_pk_exec
// Victim waves run the instruction under test in a loop on lane-dependent values, with the destination preset to a sentinel, and
// check it against two scalar v_fma_f32 (same IEEE operation): an ACTIVE lane must hold the fma, an INACTIVE lane the sentinel.
// Load: none | f16-MFMA waves in the SAME workgroup | ANOTHER kernel of f16 MFMAs on a second stream | another kernel of f32 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct Rec { unsigned long long active_wrong, inactive_written, checked; unsigned first[8]; };

template <int T>   // 0 f16 16x16x32, 1 f32 16x16x4
__global__ __launch_bounds__(256, 2) void other_mfma(int iters, float* sink) {
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 x, y;
  for (int e = 0; e < 8; ++e) x[e] = (_Float16)(0.25f * e + threadIdx.x * 0.001f), y[e] = (_Float16)(0.5f - e * 0.01f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (T == 0) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k & 3], 0, 0, 0);
      else acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)x[1], (float)y[1], acc[k & 3], 0, 0, 0);
    }
  }
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) sink[0] = 1.f;
}

// OP 0: v_pk_fma_f32 (VGPR sources)  1: v_pk_mul_f32  2: v_pk_add_f32  3: v_fma_f32 x 2 (control)  4: v_pk_fma_f32 with op_sel broadcast
// MASK 0: all lanes  1: lanes with (lane * 7 + 3) % 5 != 0  2: lower half of the wave  3: one lane in four
template <int OP, int MASK, bool INKERNEL>
__global__ __launch_bounds__(512, 1) void victim(Rec* rec, int iters) {
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (w >= 4) {
    if (INKERNEL) {
      v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      h8 x, y;
      for (int e = 0; e < 8; ++e) x[e] = (_Float16)(0.25f * e + tid * 0.001f), y[e] = (_Float16)(0.5f - e * 0.01f);
      for (int g = 0; g < 100000 && *(volatile int*)&done < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k & 3], 0, 0, 0);
      }
      if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) done = 9;
    }
    return;
  }
  const bool act = MASK == 0 ? true : MASK == 1 ? ((lane * 7 + 3) % 5 != 0) : MASK == 2 ? lane < 32 : (lane & 3) == 1;
  unsigned long long aw = 0, iw = 0, n = 0;
  unsigned first[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float SENT = 12345.678f;
  for (int it = 0; it < iters; ++it) {
    v2f a = {0.5f + 0.001f * lane + it * 0.01f, -0.25f + 0.002f * lane - it * 0.02f};
    v2f b = {0.75f + 0.0001f * it, 1.1f - 0.0003f * lane};
    v2f c = {0.1f * (it & 7), -0.3f + 0.004f * lane};
    v2f d = {SENT, SENT};
    float e0, e1;
    if (OP == 0 || OP == 3) e0 = __builtin_fmaf(a[0], b[0], c[0]), e1 = __builtin_fmaf(a[1], b[1], c[1]);
    else if (OP == 1) e0 = a[0] * b[0], e1 = a[1] * b[1];
    else if (OP == 2) e0 = a[0] + b[0], e1 = a[1] + b[1];
    else e0 = __builtin_fmaf(a[0], b[0], b[1]), e1 = __builtin_fmaf(a[1], b[0], b[1]);
    asm volatile("" : "+v"(e0), "+v"(e1), "+v"(d));
    if (act) {
      if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(d) : "v"(a), "v"(b), "v"(c));
      else if (OP == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
      else if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
      else if (OP == 3) {
        float d0 = d[0], d1 = d[1];
        const float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1], c0 = c[0], c1 = c[1];
        asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "+v"(d0), "+v"(d1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1));
        d = v2f{d0, d1};
      }
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(d) : "v"(a), "v"(b));
    }
    asm volatile("" : "+v"(d));
    const bool w0 = act ? (d[0] != e0) : (d[0] != SENT), w1 = act ? (d[1] != e1) : (d[1] != SENT);
    n += 2;
    if (w0 || w1) {
      if (act) aw += w0 + w1; else iw += w0 + w1;
      if (first[0] == 0) {
        first[0] = 1u + it, first[1] = lane | (w << 8) | ((act ? 1u : 0u) << 16) | ((w0 ? 1u : 0u) << 20) | ((w1 ? 1u : 0u) << 21);
        first[2] = __float_as_uint(d[0]), first[3] = __float_as_uint(e0), first[4] = __float_as_uint(d[1]), first[5] = __float_as_uint(e1);
        first[6] = __float_as_uint(a[0]), first[7] = __float_as_uint(a[1]);
      }
    }
  }
  if (aw) atomicAdd(&rec->active_wrong, aw);
  if (iw) atomicAdd(&rec->inactive_written, iw);
  atomicAdd(&rec->checked, n);
  if (first[0] && atomicCAS(&rec->first[0], 0u, first[0]) == 0u)
    for (int k = 1; k < 8; ++k) rec->first[k] = first[k];
  if (lane == 0) atomicAdd(&done, 1);
}

typedef void (*VK)(Rec*, int);
struct V { const char* name; VK quiet; VK inker; };
#define ROW(NAME, OP, MASK) {NAME, victim<OP, MASK, false>, victim<OP, MASK, true>}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  V vs[] = {
      ROW("v_pk_fma_f32, all lanes", 0, 0), ROW("v_pk_fma_f32, 4 of 5 lanes", 0, 1), ROW("v_pk_fma_f32, lower half", 0, 2), ROW("v_pk_fma_f32, 1 lane in 4", 0, 3),
      ROW("v_pk_mul_f32, 4 of 5 lanes", 1, 1), ROW("v_pk_add_f32, 4 of 5 lanes", 2, 1), ROW("v_pk_fma op_sel, all lanes", 4, 0), ROW("v_pk_fma op_sel, 4 of 5", 4, 1),
      ROW("2 x v_fma_f32, 4 of 5 lanes", 3, 1),
  };
  hipStream_t s1, s2;
  (void)hipStreamCreate(&s1);
  (void)hipStreamCreate(&s2);
  Rec* d;
  float* sink;
  (void)hipMalloc(&d, sizeof(Rec));
  (void)hipMalloc(&sink, 16);
  printf("# %s, %d CUs, %d iterations per lane.  cells: wrong results in ACTIVE lanes / writes to INACTIVE lanes (of results checked)\n", prop.gcnArchName, cus, iters);
  const char* cfgn[] = {"quiet", "same-kernel f16 MFMA waves", "other kernel: f16 MFMA", "other kernel: f32 MFMA"};
  for (auto& v : vs) {
    printf("%-30s", v.name);
    for (int cfg = 0; cfg < 4; ++cfg) {
      (void)hipMemsetAsync(d, 0, sizeof(Rec), s1);
      (void)hipDeviceSynchronize();
      if (cfg == 2) hipLaunchKernelGGL(other_mfma<0>, dim3(2 * cus), dim3(256), 0, s2, 60000, sink);
      if (cfg == 3) hipLaunchKernelGGL(other_mfma<1>, dim3(2 * cus), dim3(256), 0, s2, 30000, sink);
      hipLaunchKernelGGL(cfg == 1 ? v.inker : v.quiet, dim3(cus), dim3(cfg == 1 ? 512 : 256), 0, s1, d, iters);
      if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
      Rec h;
      (void)hipMemcpy(&h, d, sizeof(Rec), hipMemcpyDeviceToHost);
      printf(" | %s: %llu / %llu", cfgn[cfg], h.active_wrong, h.inactive_written);
      if (h.first[0]) {
        float got0, exp0, got1, exp1;
        memcpy(&got0, &h.first[2], 4), memcpy(&exp0, &h.first[3], 4), memcpy(&got1, &h.first[4], 4), memcpy(&exp1, &h.first[5], 4);
        printf(" [it %u lane %u wave %u %s lo %s%.6g/%.6g hi %s%.6g/%.6g]", h.first[0] - 1, h.first[1] & 63, (h.first[1] >> 8) & 7, (h.first[1] >> 16) & 1 ? "active" : "INACTIVE",
               (h.first[1] >> 20) & 1 ? "!" : "", got0, exp0, (h.first[1] >> 21) & 1 ? "!" : "", got1, exp1);
      }
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
