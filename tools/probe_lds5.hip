// Machine probe (tools only; not part of the product library), round 5: what the conflict-free LDS layouts of the Winograd kernels may rely on.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_lds5.hip -o tools/exp/probe_lds5 && tools/exp/probe_lds5
//  1. ds_read_b64 / ds_write_b128 at a 4-byte-aligned (not 8 / 16) LDS address: right values?  what rate?
//  2. global_load_lds_dwordx4 with a 4-byte-aligned LDS destination (M0) and with a 4-byte-aligned global source: right bytes where?
//  3. v_mov_b32_dpp wave_shr:1 (whole-wave shift by one lane).
//  4. LDS array time of the operand-read patterns of conv_wino2 / wgrad_wino: today's (ds_read2_b32 at a 2-word lane stride, planes
//     16 mod 32; B operands at a 4-word lane stride) against the candidates (ds_read_b64 at even offsets with planes 32 mod 64; B operands
//     at a 2-word stride; ds_read_b128 for 64-channel blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------- 1. unaligned LDS accesses
__global__ __launch_bounds__(64) void k_unaligned(float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = (float)i;
  __syncthreads();
  // ds_read_b64 at float offset 2 * lane + 3 (odd)
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 4u * (2 * lane + 3);
  v2f r;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[2 * lane] = r[0], out[2 * lane + 1] = r[1];
  // ds_write_b128 at float offset 1024 + 4 * lane + 1
  __syncthreads();
  const unsigned b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 4u * (1024 + 4 * lane + 1);
  v4f w = {1000.f + lane, 2000.f + lane, 3000.f + lane, 4000.f + lane};
  asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(b), "v"(w) : "memory");
  __syncthreads();
  for (int i = lane; i < 264; i += 64) out[128 + i] = lds[1024 + i];
  // ds_read_b128 at float offset 4 * lane + 1
  const unsigned c = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 4u * (4 * lane + 1);
  v4f q;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(c) : "memory");
  for (int e = 0; e < 4; ++e) out[512 + 4 * lane + e] = q[e];
}

// ---------------------------------------------------------------------------------------------- 2. LDS DMA alignment
// MODE 0: aligned reference; 1: LDS destination + 4 bytes; 2: global source + 4 bytes; 3: both
template <int MODE>
__global__ __launch_bounds__(256) void k_dma(const float* g, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2048; i += 256) lds[i] = -1.f;
  __syncthreads();
  float* dst = lds + wave * 272 + ((MODE & 1) ? 1 : 0);   // wave-uniform base
  const float* src = g + tid * 4 + ((MODE & 2) ? 1 : 0);
  __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = tid; i < 2048; i += 256) out[i] = lds[i];
}

// ---------------------------------------------------------------------------------------------- 3. DPP wave_shr:1
__global__ __launch_bounds__(64) void k_dpp(float* out) {
  const int lane = threadIdx.x;
  const float v = 100.f + lane;
  const int sh = __builtin_amdgcn_update_dpp(__float_as_int(-7.f), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  out[lane] = __int_as_float(sh);
  const int sh2 = __builtin_amdgcn_update_dpp(__float_as_int(-7.f), __float_as_int(v), 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  out[64 + lane] = __int_as_float(sh2);
  const int sh3 = __builtin_amdgcn_update_dpp(__float_as_int(-7.f), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  out[128 + lane] = __int_as_float(sh3);
}

// ---------------------------------------------------------------------------------------------- 4. LDS array time per pattern
// PAT 0: patch row today      : ds_read2_b32 {off, off+4 B} x2 at (k * 400 + 2 t + 3) floats             [4 dwords / lane]
//     1: patch row candidate  : ds_read_b64 x2 at (k * 416 + 2 t + 4) floats (even)                      [4 dwords / lane]
//     2: patch row unaligned  : ds_read_b64 x2 at (k * 416 + 2 t + 3) floats (odd)                       [4 dwords / lane]
//     3: B operand today      : ds_read_b64 at lane * 4 floats                                           [2 dwords / lane]
//     4: B operand candidate  : ds_read_b64 at lane * 2 floats                                           [2 dwords / lane]
//     5: B operand, 64 ch     : ds_read_b128 at lane * 4 floats                                          [4 dwords / lane]
//     6: patch row, odd plane : ds_read2_b32 x2 at (k * 401 + 2 t + 3) floats (planes odd: b32 conflict-free) [4 dwords / lane]
//     7: wgrad dy today       : ds_read_b64 at (c * 130 + 2 t) floats, c = l & 15, t = l >> 4
//     8: wgrad dy candidate   : ds_read_b64 at (c * 132 + 2 t)
//     9: wgrad patch today    : ds_read2_b32 x2 at (c * 258 + 2 t + 3)   (PLA == 2 mod 32)
//    10: wgrad patch candidate: ds_read_b64 x2 at (c * 260 + 2 t + 4)   (PLA == 4 mod 64)
template <int PAT>
__global__ __launch_bounds__(256, 2) void k_time(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 255);
  __syncthreads();
  const int t = lane & 15, k = lane >> 4;
  int off;   // floats
  if (PAT == 0) off = k * 400 + 2 * t + 3;
  else if (PAT == 1) off = k * 416 + 2 * t + 4;
  else if (PAT == 2) off = k * 416 + 2 * t + 3;
  else if (PAT == 3) off = lane * 4;
  else if (PAT == 4) off = lane * 2;
  else if (PAT == 5) off = lane * 4;
  else if (PAT == 6) off = k * 401 + 2 * t + 3;
  else if (PAT == 7) off = t * 130 + 2 * k;
  else if (PAT == 8) off = t * 132 + 2 * k;
  else if (PAT == 9) off = t * 258 + 2 * k + 3;
  else if (PAT == 10) off = t * 260 + 2 * k + 4;
  else if (PAT == 11) off = k * 416 + 2 * t + 4;
  else if (PAT == 12) off = lane * 2;
  else off = lane;
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 4u * off;
  v4f s = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (PAT == 11) {
        v4f r0;
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r0) : "v"(a), "n"(u * 20), "n"(u * 20 + 1) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s += r0;
        continue;
      }
      if (PAT == 12 || PAT == 13) {
        float r0;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(a), "n"(u * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s[0] += r0;
        continue;
      }   // 8 "rows" at 40-float pitch (immediate offsets)
      if (PAT == 0 || PAT == 6 || PAT == 9) {
        v2f r0, r1;
        asm volatile("ds_read2_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2_b32 %1, %2 offset0:%5 offset1:%6"
                     : "=v"(r0), "=v"(r1) : "v"(a), "n"(u * 30), "n"(u * 30 + 1), "n"(u * 30 + 2), "n"(u * 30 + 3) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s[0] += r0[0], s[1] += r0[1], s[2] += r1[0], s[3] += r1[1];
      } else if (PAT == 1 || PAT == 2 || PAT == 10) {
        v2f r0, r1;
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=v"(r0), "=v"(r1) : "v"(a), "n"(u * 160), "n"(u * 160 + 8) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s[0] += r0[0], s[1] += r0[1], s[2] += r1[0], s[3] += r1[1];
      } else if (PAT == 3 || PAT == 4 || PAT == 7 || PAT == 8) {
        v2f r0;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r0) : "v"(a), "n"(u * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s[0] += r0[0], s[1] += r0[1];
      } else {
        v4f r0;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(a), "n"(u * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s += r0;
      }
    }
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

// throughput form: 8 independent reads in flight per wait
template <int PAT>
__global__ __launch_bounds__(256, 2) void k_tput(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 255);
  __syncthreads();
  const int t = lane & 15, k = lane >> 4;
  int off;
  if (PAT == 0) off = k * 400 + 2 * t + 3;
  else if (PAT == 1) off = k * 416 + 2 * t + 4;
  else if (PAT == 2) off = k * 416 + 2 * t + 3;
  else if (PAT == 3) off = lane * 4;
  else if (PAT == 4) off = lane * 2;
  else if (PAT == 5) off = lane * 4;
  else if (PAT == 6) off = k * 401 + 2 * t + 3;
  else if (PAT == 7) off = t * 130 + 2 * k;
  else if (PAT == 8) off = t * 132 + 2 * k;
  else if (PAT == 9) off = t * 258 + 2 * k + 3;
  else if (PAT == 10) off = t * 260 + 2 * k + 4;
  else if (PAT == 11) off = k * 416 + 2 * t + 4;
  else if (PAT == 12) off = lane * 2;
  else off = lane;
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 4u * off;
  v4f s = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (PAT == 11) {
      v4f r[4];
      asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:1\n\tds_read2_b64 %1, %4 offset0:20 offset1:21\n\t"
                   "ds_read2_b64 %2, %4 offset0:40 offset1:41\n\tds_read2_b64 %3, %4 offset0:60 offset1:61\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "v"(a) : "memory");
      s += (r[0] + r[1]) + (r[2] + r[3]);
      continue;
    }
    if (PAT == 12 || PAT == 13) {
      float r[8];
      asm volatile(
          "ds_read_b32 %0, %8 offset:0\n\tds_read_b32 %1, %8 offset:1024\n\tds_read_b32 %2, %8 offset:2048\n\tds_read_b32 %3, %8 offset:3072\n\t"
          "ds_read_b32 %4, %8 offset:4096\n\tds_read_b32 %5, %8 offset:5120\n\tds_read_b32 %6, %8 offset:6144\n\tds_read_b32 %7, %8 offset:7168\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(a) : "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e & 3] += r[e];
      continue;
    }
    if (PAT == 0 || PAT == 6 || PAT == 9) {
      v2f r[8];
      asm volatile(
          "ds_read2_b32 %0, %8 offset0:0 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\t"
          "ds_read2_b32 %2, %8 offset0:40 offset1:41\n\tds_read2_b32 %3, %8 offset0:42 offset1:43\n\t"
          "ds_read2_b32 %4, %8 offset0:80 offset1:81\n\tds_read2_b32 %5, %8 offset0:82 offset1:83\n\t"
          "ds_read2_b32 %6, %8 offset0:120 offset1:121\n\tds_read2_b32 %7, %8 offset0:122 offset1:123\n\ts_waitcnt lgkmcnt(0)"
          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(a) : "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e & 3] += r[e][0] + r[e][1];
    } else if (PAT == 1 || PAT == 2 || PAT == 10) {
      v2f r[8];
      asm volatile(
          "ds_read_b64 %0, %8 offset:0\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:160\n\tds_read_b64 %3, %8 offset:168\n\t"
          "ds_read_b64 %4, %8 offset:320\n\tds_read_b64 %5, %8 offset:328\n\tds_read_b64 %6, %8 offset:480\n\tds_read_b64 %7, %8 offset:488\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(a) : "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e & 3] += r[e][0] + r[e][1];
    } else if (PAT == 3 || PAT == 4 || PAT == 7 || PAT == 8) {
      v2f r[8];
      asm volatile(
          "ds_read_b64 %0, %8 offset:0\n\tds_read_b64 %1, %8 offset:1024\n\tds_read_b64 %2, %8 offset:2048\n\tds_read_b64 %3, %8 offset:3072\n\t"
          "ds_read_b64 %4, %8 offset:4096\n\tds_read_b64 %5, %8 offset:5120\n\tds_read_b64 %6, %8 offset:6144\n\tds_read_b64 %7, %8 offset:7168\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(a) : "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e & 3] += r[e][0] + r[e][1];
    } else {
      v4f r[8];
      asm volatile(
          "ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
          "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(a) : "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) s += r[e];
    }
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

template <int PAT>
static void time_pat(const char* name, float* sink, int dwords_per_lane_8) {
  const int iters = 4000, blocks = 512;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int form = 0; form < 2; ++form) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, 0));
      if (form == 0) hipLaunchKernelGGL(k_time<PAT>, dim3(blocks), dim3(256), 0, 0, sink, iters);
      else hipLaunchKernelGGL(k_tput<PAT>, dim3(blocks), dim3(256), 0, 0, sink, iters);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    // per CU: 2 workgroups x 4 waves, each iters x 8 accesses
    const double bytes = (double)blocks * 256 * iters * dwords_per_lane_8 * 4.0;
    printf("  %-44s %s: %8.3f ms  %7.1f TB/s chip-wide  (%.1f B/clk/CU at 2.4 GHz)\n", name, form == 0 ? "one at a time " : "8 in flight    ", best,
           bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256.0 / 2.4e9);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  float *out, *g;
  CK(hipMalloc(&out, 4096 * sizeof(float)));
  CK(hipMalloc(&g, 4096 * sizeof(float)));
  std::vector<float> h(4096), hg(4096);
  for (int i = 0; i < 4096; ++i) hg[i] = (float)i;
  CK(hipMemcpy(g, hg.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));

  // 1
  CK(hipMemset(out, 0, 4096 * sizeof(float)));
  hipLaunchKernelGGL(k_unaligned, dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), out, 4096 * sizeof(float), hipMemcpyDeviceToHost));
  {
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += (h[2 * l] != (float)(2 * l + 3)) + (h[2 * l + 1] != (float)(2 * l + 4));
    printf("1a ds_read_b64 at an odd dword address: %s (lane 0 got %g %g, lane 5 got %g %g)\n", bad ? "WRONG" : "right", h[0], h[1], h[10], h[11]);
    bad = 0;
    for (int i = 0; i < 264; ++i) {
      float exp = (float)(1024 + i);
      if (i >= 1 && i < 257) { const int l = (i - 1) / 4, e = (i - 1) % 4; exp = 1000.f * (e + 1) + l; }
      bad += h[128 + i] != exp;
    }
    printf("1b ds_write_b128 at a 4-byte-aligned address: %s (first words %g %g %g %g %g %g)\n", bad ? "WRONG" : "right", h[128], h[129], h[130], h[131], h[132],
           h[133]);
    bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) bad += h[512 + 4 * l + e] != (float)(4 * l + 1 + e);
    printf("1c ds_read_b128 at a 4-byte-aligned address: %s (lane 0 got %g %g %g %g)\n", bad ? "WRONG" : "right", h[512], h[513], h[514], h[515]);
  }
  // 2
  for (int mode = 0; mode < 4; ++mode) {
    CK(hipMemset(out, 0, 4096 * sizeof(float)));
    if (mode == 0) hipLaunchKernelGGL(k_dma<0>, dim3(1), dim3(256), 0, 0, g, out);
    if (mode == 1) hipLaunchKernelGGL(k_dma<1>, dim3(1), dim3(256), 0, 0, g, out);
    if (mode == 2) hipLaunchKernelGGL(k_dma<2>, dim3(1), dim3(256), 0, 0, g, out);
    if (mode == 3) hipLaunchKernelGGL(k_dma<3>, dim3(1), dim3(256), 0, 0, g, out);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("2.%d LDS DMA: kernel failed: %s\n", mode, hipGetErrorString(e)); break; }
    CK(hipMemcpy(h.data(), out, 4096 * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<float> exp(2048, -1.f);
    for (int t = 0; t < 256; ++t)
      for (int e2 = 0; e2 < 4; ++e2) exp[(t >> 6) * 272 + ((mode & 1) ? 1 : 0) + (t & 63) * 4 + e2] = (float)(t * 4 + ((mode & 2) ? 1 : 0) + e2);
    int bad = 0, first = -1;
    for (int i = 0; i < 2048; ++i)
      if (h[i] != exp[i]) { if (first < 0) first = i; ++bad; }
    printf("2.%d LDS DMA dwordx4, LDS destination %s, global source %s: %s", mode, (mode & 1) ? "+4 B" : "aligned", (mode & 2) ? "+4 B" : "aligned",
           bad ? "DIFFERENT" : "as expected (base + lane * 16 B)");
    if (bad) {
      printf(" -- %d words differ, first at %d: got", bad, first);
      for (int i = first; i < first + 8 && i < 2048; ++i) printf(" %g", h[i]);
      printf(" | expected");
      for (int i = first; i < first + 8 && i < 2048; ++i) printf(" %g", exp[i]);
    }
    printf("\n");
  }
  // 3
  CK(hipMemset(out, 0, 4096 * sizeof(float)));
  hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), out, 4096 * sizeof(float), hipMemcpyDeviceToHost));
  {
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[l] != (l == 0 ? -7.f : 100.f + l - 1);
    printf("3a v_mov_b32_dpp wave_shr:1: %s (lanes 0 1 16 32 63 got %g %g %g %g %g)\n", bad ? "NOT lane - 1 across the wave" : "lane - 1 across the whole wave, lane 0 keeps old",
           h[0], h[1], h[16], h[32], h[63]);
    bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[64 + l] != ((l & 15) == 0 ? -7.f : 100.f + l - 1);
    printf("3b row_shr:1: %s (lanes 0 1 16 17 got %g %g %g %g)\n", bad ? "unexpected" : "lane - 1 inside rows of 16, first lane of a row keeps old", h[64], h[65], h[80], h[81]);
    bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[128 + l] != (l == 63 ? -7.f : 100.f + l + 1);
    printf("3c wave_shl:1: %s (lanes 0 15 62 63 got %g %g %g %g)\n", bad ? "NOT lane + 1 across the wave" : "lane + 1 across the whole wave", h[128], h[143], h[190], h[191]);
  }
  // 4
  printf("4 LDS array time per pattern (512 workgroups x 4 waves, 2 workgroups per CU):\n");
  time_pat<0>("patch today (read2_b32, planes 16 mod 32)", out, 4);
  time_pat<1>("patch even (read_b64, planes 32 mod 64)", out, 4);
  time_pat<2>("patch ODD address read_b64, planes 32 mod 64", out, 4);
  time_pat<6>("patch read2_b32, odd planes", out, 4);
  time_pat<3>("B today (read_b64, 4-word lane stride)", out, 2);
  time_pat<4>("B candidate (read_b64, 2-word lane stride)", out, 2);
  time_pat<5>("B 64 channels (read_b128, 4-word stride)", out, 4);
  time_pat<7>("wgrad dy today (read_b64, planes 2 mod 32)", out, 2);
  time_pat<8>("wgrad dy candidate (planes 4 mod 64)", out, 2);
  time_pat<9>("wgrad patch today (read2_b32, planes 2 mod 32)", out, 4);
  time_pat<10>("wgrad patch candidate (read_b64, planes 4 mod 64)", out, 4);
  time_pat<11>("patch even as ds_read2_b64", out, 4);
  time_pat<12>("B 16 ch today (read_b32, 2-word lane stride)", out, 1);
  time_pat<13>("B 16 ch candidate (read_b32, 1-word stride)", out, 1);
  return 0;
}
