// Machine probe for the split-precision conv path (tools only; not part of the product library).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_sp.hip -o tools/exp/probe_sp && tools/exp/probe_sp
// 1. ds_read_b64_tr_b16: which 16-bit element each lane receives, as a function of the per-lane addresses
// 2. v_mfma_f32_16x16x32_f16 operand layout (A[i = l & 15][k = 8 (l >> 4) + e], B[k][j = l & 15]) against a host product
// 3. f16 subnormal operands of the MFMA (kept or flushed), v_cvt_pkrtz_f16_f32 on overflow, v_cvt_pk_f16_f32 on overflow
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void tr_probe(const int* addr_of_lane, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  const int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr_of_lane[l]));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}

__global__ void mfma_probe(const _Float16* A, const _Float16* B, float* D) {   // A [16][32], B [32][16] row-major
  const int l = threadIdx.x;
  h8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = A[(l & 15) * 32 + 8 * (l >> 4) + e], b[e] = B[(8 * (l >> 4) + e) * 16 + (l & 15)];
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

__global__ void misc_probe(float* out) {
  const int l = threadIdx.x;
  // subnormal f16 operand: 2^-20 * 2^10
  h8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = (_Float16)0.f, b[e] = (_Float16)0.f;
  a[0] = (_Float16)9.5367431640625e-07f;   // 2^-20 (subnormal in f16)
  b[0] = (_Float16)1024.f;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (l == 0) out[0] = c[0];                // lanes 0..15 x k-group 0: 2^-10 if subnormals are kept
  volatile float big = 1.0e6f, neg = -7.0e4f, small = 3.0e-8f;
  fp16x2 z = __builtin_amdgcn_cvt_pkrtz(big, neg);
  if (l == 0) out[1] = (float)z[0], out[2] = (float)z[1];
  h2 rne = {(_Float16)big, (_Float16)small};
  if (l == 0) out[3] = (float)rne[0], out[4] = (float)rne[1];
  fp16x2 z2 = __builtin_amdgcn_cvt_pkrtz(small, 65519.f);
  if (l == 0) out[5] = (float)z2[0], out[6] = (float)z2[1];
}

// MFMA f16 issue rate with K f32 vector instructions per MFMA in the same wave (do they overlap?)
template <int K>
__global__ __launch_bounds__(256) void mix_probe(float* out, int iters) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = (_Float16)(threadIdx.x * 0.001f + e), b[e] = (_Float16)(e * 0.01f);
  v4f c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[q], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[(q * K + k) & 7] = fmaf(v[(q * K + k) & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K>
static void run_mix(float* d_out, int blocks) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(mix_probe<K>, dim3(blocks), dim3(256), 0, 0, d_out, 16);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mix_probe<K>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)blocks * 4 * iters * 4;
  printf("mix K=%d blocks=%d: %.3f ms, %.1f TFLOP/s f16 MFMA, %.2f ns per MFMA per wave\n", K, blocks, ms,
         mf * 2 * 16 * 16 * 32 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 4));
}

int main() {
  // ---- 1
  int h_addr[64];
  short h_out[256];
  int* d_addr;
  short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)), hipMalloc(&d_out, sizeof(h_out));
  const char* names[3] = {"addr = 4*l (lane-linear 8 bytes)", "addr = 4*(l ^ 5)", "addr = 64*(l&15) + 4*(l>>4)  (row pitch 128 B)"};
  for (int v = 0; v < 3; ++v) {
    for (int l = 0; l < 64; ++l) h_addr[l] = v == 0 ? 4 * l : v == 1 ? 4 * (l ^ 5) : 64 * (l & 15) + 4 * (l >> 4);
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("tr_b16 %s\n", names[v]);
    // hypothesis: within a 16-lane group, lane i's element e = element (i & 3) of the 4 halves at lane (4 e + (i >> 2))'s address
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const int g = l & ~15, i = l & 15;
      for (int e = 0; e < 4; ++e) {
        const int want = h_addr[g + 4 * e + (i >> 2)] + (i & 3);
        if (h_out[l * 4 + e] != want) ++bad;
      }
    }
    printf("  hypothesis out[l][e] = lds[addr[g + 4 e + (i >> 2)] + (i & 3)]: %s (%d mismatches)\n", bad ? "FAILS" : "holds", bad);
    if (bad || v == 0)
      for (int l = 0; l < 64; l += (bad ? 1 : 17)) printf("  lane %2d: %4d %4d %4d %4d\n", l, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  // ---- 2
  _Float16 hA[512], hB[512];
  float hD[256], ref[256];
  srand(3);
  for (int i = 0; i < 512; ++i) hA[i] = (_Float16)(float)(rand() % 15 - 7), hB[i] = (_Float16)(float)(rand() % 13 - 6);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0;
      for (int k = 0; k < 32; ++k) s += (float)hA[i * 32 + k] * (float)hB[k * 16 + j];
      ref[i * 16 + j] = s;
    }
  _Float16 *dA, *dB;
  float* dD;
  hipMalloc(&dA, sizeof(hA)), hipMalloc(&dB, sizeof(hB)), hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice), hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("mfma_f32_16x16x32_f16 layout A[l&15][8(l>>4)+e], B[8(l>>4)+e][l&15], D[4(l>>4)+r][l&15]: %s (%d mismatches)\n", bad ? "FAILS" : "holds", bad);
  // ---- 3
  float hm[8], *dm;
  hipMalloc(&dm, sizeof(hm));
  hipLaunchKernelGGL(misc_probe, dim3(1), dim3(64), 0, 0, dm);
  hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
  printf("mfma f16 subnormal operand 2^-20 * 2^10 = %g (2^-10 = %g if kept, 0 if flushed)\n", hm[0], ldexp(1.0, -10));
  printf("cvt_pkrtz(1e6, -7e4) = %g, %g ; RNE cvt(1e6) = %g, RNE cvt(3e-8) = %g ; cvt_pkrtz(3e-8, 65519) = %g, %g\n", hm[1], hm[2], hm[3], hm[4], hm[5], hm[6]);
  // ---- 4
  float* dmix;
  hipMalloc(&dmix, 4096 * 256 * sizeof(float));
  for (int blocks : {256, 512, 1024}) {
    run_mix<0>(dmix, blocks), run_mix<1>(dmix, blocks), run_mix<2>(dmix, blocks), run_mix<4>(dmix, blocks), run_mix<8>(dmix, blocks);
  }
  return 0;
}
