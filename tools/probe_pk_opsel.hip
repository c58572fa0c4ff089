// Machine probe (tools only; not part of the product library) -- the characterisation behind profiles/r4_sp_root_cause.md:
//   WHICH packed-f32 (VOP3P) instruction forms return a wrong result while another wave of the SIMD issues matrix instructions, and
//   for WHICH matrix instruction types.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_pk_opsel.hip -o tools/exp/probe_pk_opsel && tools/exp/probe_pk_opsel
// Waves 0-3 of a workgroup execute one packed instruction form in a loop on lane-dependent values and compare both halves with the
// same arithmetic done by scalar instructions (IEEE fma / mul / add: bit-identical by definition); waves 4-7 issue one matrix
// instruction type back to back until the victims are done.  One line per form: wrong low halves / wrong high halves per aggressor type.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

struct Rec { unsigned long long lo_wrong, hi_wrong, checked; };

enum { A_NONE = 0, A_F16_16x16x32, A_BF16_16x16x32, A_F16_32x32x16, A_F16_16x16x16, A_F32_16x16x4, A_F32_32x32x2, A_I8_16x16x64, A_FP8_16x16x32, A_COUNT };
static const char* kAgg[] = {"none", "f16_16x16x32", "bf16_16x16x32", "f16_32x32x16", "f16_16x16x16", "f32_16x16x4", "f32_32x32x2", "i8_16x16x64", "fp8_16x16x32"};

template <int A>
__device__ __forceinline__ void aggress(volatile int* done) {
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v16f big;
  for (int r = 0; r < 16; ++r) big[r] = 0.f;
  i4 iacc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 x, y;
  b8 bx, by;
  for (int e = 0; e < 8; ++e) x[e] = (_Float16)(0.25f * e + threadIdx.x * 0.001f), y[e] = (_Float16)(0.5f - e * 0.01f), bx[e] = (__bf16)(0.25f * e), by[e] = (__bf16)(0.5f);
  const i4 ix = {0x01020304, 0x05060708, 0x01010101, 0x02020202}, iy = {0x01010101, 0x01010101, 0x02020202, 0x01010101};
  const long fx = 0x3838383838383838L, fy = 0x3030303030303030L;
  for (int g = 0; g < 200000 && *done < 4; ++g) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (A == A_F16_16x16x32) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k & 3], 0, 0, 0);
      if (A == A_BF16_16x16x32) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, acc[k & 3], 0, 0, 0);
      if (A == A_F16_32x32x16) big = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, big, 0, 0, 0);
      if (A == A_F16_16x16x16) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4{x[0], x[1], x[2], x[3]}, h4{y[0], y[1], y[2], y[3]}, acc[k & 3], 0, 0, 0);
      if (A == A_F32_16x16x4) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)x[1], (float)y[1], acc[k & 3], 0, 0, 0);
      if (A == A_F32_32x32x2) big = __builtin_amdgcn_mfma_f32_32x32x2f32((float)x[1], (float)y[1], big, 0, 0, 0);
      if (A == A_I8_16x16x64) iacc[k & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ix, iy, iacc[k & 1], 0, 0, 0);
      if (A == A_FP8_16x16x32) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fx, fy, acc[k & 3], 0, 0, 0);
    }
  }
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + big[0] + (float)iacc[0][0] + (float)iacc[1][0] == 123.456f) *done = 9;
}

// the forms: d = OP(a, b, c) with modifiers; E0 / E1 = what the low / high half must be
#define FORMS(X)                                                                                                                         \
  X(0, "v_pk_fma_f32 d,a,b,c", "v_pk_fma_f32 %0, %1, %2, %3", fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1]))                              \
  X(1, "v_pk_fma_f32 d,a,b,b op_sel:[0,0,1] op_sel_hi:[1,0,1]", "v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]", fmaf(a[0], b[0], b[1]), fmaf(a[1], b[0], b[1])) \
  X(2, "v_pk_fma_f32 d,a,b,c op_sel:[0,0,1] (src2.lo <- c.hi)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]", fmaf(a[0], b[0], c[1]), fmaf(a[1], b[1], c[1])) \
  X(3, "v_pk_fma_f32 d,a,b,c op_sel_hi:[1,1,0] (src2.hi <- c.lo)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]", fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[0])) \
  X(4, "v_pk_fma_f32 d,a,b,c op_sel:[0,1,0] (src1.lo <- b.hi)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]", fmaf(a[0], b[1], c[0]), fmaf(a[1], b[1], c[1])) \
  X(5, "v_pk_fma_f32 d,a,b,c op_sel_hi:[1,0,1] (src1.hi <- b.lo)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]", fmaf(a[0], b[0], c[0]), fmaf(a[1], b[0], c[1])) \
  X(6, "v_pk_fma_f32 d,a,b,c op_sel:[1,0,0] (src0.lo <- a.hi)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]", fmaf(a[1], b[0], c[0]), fmaf(a[1], b[1], c[1])) \
  X(7, "v_pk_mul_f32 d,a,b", "v_pk_mul_f32 %0, %1, %2", a[0] * b[0], a[1] * b[1])                                                        \
  X(8, "v_pk_mul_f32 d,a,b op_sel_hi:[1,0] (src1.hi <- b.lo)", "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", a[0] * b[0], a[1] * b[0])      \
  X(9, "v_pk_mul_f32 d,a,b op_sel:[0,1] (src1.lo <- b.hi)", "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]", a[0] * b[1], a[1] * b[1]) \
  X(10, "v_pk_add_f32 d,a,b", "v_pk_add_f32 %0, %1, %2", a[0] + b[0], a[1] + b[1])                                                      \
  X(11, "v_pk_add_f32 d,a,b op_sel_hi:[0,1] (src0.hi <- a.lo)", "v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]", a[0] + b[0], a[0] + b[1])     \
  X(12, "v_pk_add_f32 d,a,b op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0] (Winograd mid)", "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]", a[1] + b[0], b[0] - a[1]) \
  X(13, "v_pk_add_f32 d,a,a op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1] (Winograd sd)", "v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]", a[0] + a[1], a[0] - a[1]) \
  X(14, "v_pk_add_f32 d,a,b neg_lo:[0,1] neg_hi:[0,1] (a - b)", "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]", a[0] - b[0], a[1] - b[1]) \
  X(15, "v_pk_fma_f32 d,a,b,c op_sel_hi:[1,0,0] (the library's broadcast form)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]", fmaf(a[0], b[0], c[0]), fmaf(a[1], b[0], c[0])) \
  X(16, "v_pk_mul_f32 d,a,b op_sel_hi:[0,1] (src0.hi <- a.lo)", "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", a[0] * b[0], a[0] * b[1]) \
  X(17, "v_pk_fma_f32 d,a,b,c op_sel:[0,1,1] (src1.lo, src2.lo <- hi)", "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]", fmaf(a[0], b[1], c[1]), fmaf(a[1], b[1], c[1]))

template <int F>
__device__ __forceinline__ void form(v2f& d, v2f a, v2f b, v2f c, float& e0, float& e1) {
#define X(ID, NAME, ASM, E0, E1)                                          \
  if (F == ID) {                                                          \
    e0 = (E0), e1 = (E1);                                                 \
    asm volatile("" : "+v"(e0), "+v"(e1));                                \
    asm volatile(ASM : "=v"(d) : "v"(a), "v"(b), "v"(c));                 \
  }
  FORMS(X)
#undef X
}
static const char* kForm[] = {
#define X(ID, NAME, ASM, E0, E1) NAME,
    FORMS(X)
#undef X
};
constexpr int kNForms = 18;

template <int F, int A>
__global__ __launch_bounds__(512, 1) void probe(Rec* rec, int iters) {
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (w >= 4) {
    if (A != A_NONE) aggress<A>(&done);
    return;
  }
  unsigned long long lw = 0, hw = 0;
  for (int it = 0; it < iters; ++it) {
    v2f a = {0.5f + 0.001f * lane + it * 0.01f, -0.25f + 0.002f * lane - it * 0.02f};
    v2f b = {0.75f + 0.0001f * it, 1.1f - 0.0003f * lane};
    v2f c = {0.1f * (it & 7) + 0.3f, -0.3f + 0.004f * lane};
    v2f d;
    float e0, e1;
    form<F>(d, a, b, c, e0, e1);
    asm volatile("" : "+v"(d));
    lw += d[0] != e0, hw += d[1] != e1;
  }
  if (lw) atomicAdd(&rec->lo_wrong, lw);
  if (hw) atomicAdd(&rec->hi_wrong, hw);
  if (lane == 0) atomicAdd(&done, 1);
}

typedef void (*PK)(Rec*, int);
template <int F, int A> struct Tab {
  static void fill(PK (*t)[A_COUNT]) {
    t[F][A] = probe<F, A>;
    if constexpr (A + 1 < A_COUNT) Tab<F, A + 1>::fill(t);
    else if constexpr (F + 1 < kNForms) Tab<F + 1, 0>::fill(t);
  }
};

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  static PK tab[kNForms][A_COUNT];
  Tab<0, 0>::fill(tab);
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  Rec* d;
  (void)hipMalloc(&d, sizeof(Rec));
  printf("# %s, %d CUs; %llu results checked per cell; cells = wrong LOW halves / wrong HIGH halves\n", prop.gcnArchName, cus,
         (unsigned long long)cus * 256 * iters);
  printf("%-70s", "# packed form \\ matrix instructions issued by the other waves:");
  for (int a = 0; a < A_COUNT; ++a) printf(" %15s", kAgg[a]);
  printf("\n");
  for (int f = 0; f < kNForms; ++f) {
    printf("%-70s", kForm[f]);
    for (int a = 0; a < A_COUNT; ++a) {
      (void)hipMemset(d, 0, sizeof(Rec));
      hipLaunchKernelGGL(tab[f][a], dim3(cus), dim3(512), 0, 0, d, iters);
      if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
      Rec h;
      (void)hipMemcpy(&h, d, sizeof(Rec), hipMemcpyDeviceToHost);
      char buf[40];
      snprintf(buf, sizeof(buf), "%llu/%llu", h.lo_wrong, h.hi_wrong);
      printf(" %15s", buf);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
