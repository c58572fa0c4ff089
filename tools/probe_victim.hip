// Machine probe (tools only): a SYNTHETIC f32-MFMA kernel shaped like the conv kernels' main loop, run while ANOTHER kernel issues f16
// MFMAs on a second stream -- which ingredient of the victim makes it fail?  (profiles/r4_sp_root_cause.md)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_victim.hip -o tools/exp/probe_victim && tools/exp/probe_victim
// Victim variants (template flags): LDSOP operands come from LDS (ds_read -> counted waits -> MFMA, registers re-used, as hipcc schedules
// it) instead of registers; BAR a workgroup barrier per chunk; RESTAGE the LDS tiles are rewritten every chunk (ds_write before the
// barrier); NACC accumulators per wave (4 or 16).  Each victim thread stores its accumulators; a loaded launch is compared bit for bit
// with a quiet one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void other_f16(int iters, float* sink) {
  h8 x, y;
  for (int e = 0; e < 8; ++e) x[e] = (_Float16)(0.25f * e + threadIdx.x * 0.001f), y[e] = (_Float16)(0.5f - e * 0.01f);
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k & 3], 0, 0, 0);
  }
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256, 2) void other_f32(int iters, float* sink) {
  float x = 0.25f + threadIdx.x * 0.001f, y = 0.5f;
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[k & 3], 0, 0, 0);
  }
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) sink[0] = 1.f;
}

template <bool LDSOP, bool BAR, bool RESTAGE, int MT, int NT>
__global__ __launch_bounds__(256, 2) void victim(float* out, int chunks) {
  __shared__ float in_t[8 * 272];    // 8 channels x (16 rows x 17): A[m][k] at k * 272 + row * 17 ... (plain layout, conflicts irrelevant)
  __shared__ float w_t[8 * 80];      // 8 channels x 64 output channels (+ pad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8 * 272; i += 256) in_t[i] = 0.01f * (i % 97) - 0.3f;
  for (int i = tid; i < 8 * 80; i += 256) w_t[i] = 0.02f * (i % 53) - 0.4f;
  __syncthreads();
  v4f acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  float ra[MT] = {}, rb[NT] = {};
  for (int i = 0; i < MT; ++i) ra[i] = 0.5f + 0.01f * lane + i;
  for (int j = 0; j < NT; ++j) rb[j] = 0.25f - 0.02f * lane + j;
  for (int c = 0; c < chunks; ++c) {
    if (RESTAGE) {
      for (int i = tid; i < 8 * 272; i += 256) in_t[i] = 0.01f * ((i + c) % 97) - 0.3f;
      for (int i = tid; i < 8 * 80; i += 256) w_t[i] = 0.02f * ((i + 3 * c) % 53) - 0.4f;
    }
    if (BAR) __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {   // two K groups of four channels
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = LDSOP ? in_t[(4 * g + (lane >> 4)) * 272 + ((wave * MT + i) & 15) * 17 + (lane & 15)] : ra[i] + g;
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = LDSOP ? w_t[(4 * g + (lane >> 4)) * 80 + j * 16 + (lane & 15)] : rb[j] - g;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float* o = out + ((size_t)blockIdx.x * 256 + tid) * (MT * NT * 4);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(i * NT + j) * 4 + r] = acc[i][j][r];
}

// no matrix instruction at all: LDS TABLE reads as in the conv kernels' staging code -- every active lane reads the SAME float2 (a
// broadcast read), optionally under a partial execution mask (the `valid quad` mask), then plain vector math
template <bool MASKED, bool PAIR>
__global__ __launch_bounds__(256, 2) void victim_table(float* out, int chunks) {
  __shared__ float2 tab[512];
  __shared__ float tile[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 512; i += 256) tab[i] = make_float2(0.5f + 0.001f * i, 0.25f - 0.002f * i);
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
  const bool act = !MASKED || ((tid * 7 + 3) % 5 != 0);
  for (int c = 0; c < chunks * 8; ++c) {
    if (act) {
      if (PAIR) {
        const float2 t = tab[c & 511];
        s0 = fmaf(s0, 0.999f, t.x), s1 = fmaf(s1, 0.998f, t.y);
      } else {
        const float t = tab[c & 511].x;
        s0 = fmaf(s0, 0.999f, t), s1 += 1.f;
      }
      tile[(tid * 4 + c) & 2047] = s0;    // (an LDS store burst next to the reads, as the staging code has)
    }
  }
  out[((size_t)blockIdx.x * 256 + tid) * 2] = s0;
  out[((size_t)blockIdx.x * 256 + tid) * 2 + 1] = s1 + tile[tid];
}

// no matrix instruction, no LDS: packed f32 math with operand selection, the instruction of the conv kernels' loader transform
//     v_pk_fma_f32 d, x, t, t op_sel:[0,0,1] op_sel_hi:[1,0,1]      d.lo = x.lo * t.lo + t.hi,  d.hi = x.hi * t.lo + t.hi
// MODE 0: that form; 1: plain v_pk_fma_f32 without op_sel (three full pairs); 2: two scalar v_fma_f32 (control)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void victim_pk(float* out, int chunks) {
  const int tid = threadIdx.x;
  v2f x = {0.5f + 0.001f * tid, -0.25f + 0.002f * tid}, t = {0.75f + 0.0001f * tid, 0.1f + 0.0003f * tid}, d = {0.f, 0.f};
  v2f tt = {t[0], t[0]}, uu = {t[1], t[1]};
  for (int c = 0; c < chunks * 64; ++c) {
    if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(t));
    else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(tt), "v"(uu));
    else asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5" : "=&v"(d[0]), "=&v"(d[1]) : "v"(x[0]), "v"(x[1]), "v"(t[0]), "v"(t[1]));
    x[0] = d[0] * 0.5f + 0.125f, x[1] = d[1] * 0.5f - 0.125f;     // (keeps the values bounded and every iteration dependent)
  }
  out[((size_t)blockIdx.x * 256 + tid) * 2] = x[0];
  out[((size_t)blockIdx.x * 256 + tid) * 2 + 1] = x[1];
}

typedef void (*VK)(float*, int);
struct V { const char* name; VK k; int words; };

int main(int argc, char** argv) {
  const int chunks = argc > 1 ? atoi(argv[1]) : 64;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = 2 * cus;
  V vs[] = {
      {"regs,    no barrier, 2x2", victim<false, false, false, 2, 2>, 16},
      {"regs,    no barrier, 4x4", victim<false, false, false, 4, 4>, 64},
      {"regs,    barrier,    4x4", victim<false, true, false, 4, 4>, 64},
      {"LDS ops, no barrier, 2x2", victim<true, false, false, 2, 2>, 16},
      {"LDS ops, no barrier, 4x4", victim<true, false, false, 4, 4>, 64},
      {"LDS ops, barrier,    4x4", victim<true, true, false, 4, 4>, 64},
      {"LDS ops, restaged,   4x4", victim<true, true, true, 4, 4>, 64},
      {"LDS ops, restaged,   2x4", victim<true, true, true, 2, 4>, 32},
      {"v_pk_fma_f32 op_sel", victim_pk<0>, 2},
      {"v_pk_fma_f32 plain", victim_pk<1>, 2},
      {"2 x v_fma_f32 (control)", victim_pk<2>, 2},
      {"table b64, all lanes", victim_table<false, true>, 2},
      {"table b64, masked lanes", victim_table<true, true>, 2},
      {"table b32, all lanes", victim_table<false, false>, 2},
      {"table b32, masked lanes", victim_table<true, false>, 2},
  };
  hipStream_t s1, s2;
  (void)hipStreamCreate(&s1);
  (void)hipStreamCreate(&s2);
  float *d, *sink;
  const size_t nmax = (size_t)blocks * 256 * 64;
  (void)hipMalloc(&d, nmax * 4);
  (void)hipMalloc(&sink, 16);
  std::vector<float> ref(nmax), got(nmax);
  printf("# %s, %d CUs; synthetic f32-MFMA victim (%d blocks, %d chunks) | quiet again | other kernel f16 MFMA | other kernel f32 MFMA : launches wrong of 10 (words wrong in the worst)\n",
         prop.gcnArchName, cus, blocks, chunks);
  for (auto& v : vs) {
    const size_t n = (size_t)blocks * 256 * v.words;
    hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, s1, d, chunks);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(ref.data(), d, n * 4, hipMemcpyDeviceToHost);
    printf("%-28s", v.name);
    for (int cfg = 0; cfg < 3; ++cfg) {
      int badl = 0;
      size_t worst = 0;
      for (int rep = 0; rep < 10; ++rep) {
        (void)hipMemsetAsync(d, 0, n * 4, s1);
        if (cfg == 1) hipLaunchKernelGGL(other_f16, dim3(2 * cus), dim3(256), 0, s2, 20000, sink);
        if (cfg == 2) hipLaunchKernelGGL(other_f32, dim3(2 * cus), dim3(256), 0, s2, 10000, sink);
        hipLaunchKernelGGL(v.k, dim3(blocks), dim3(256), 0, s1, d, chunks);
        if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
        (void)hipMemcpy(got.data(), d, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
        badl += bad != 0;
        worst = bad > worst ? bad : worst;
      }
      printf("  | %2d/10 (%zu)", badl, worst);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
