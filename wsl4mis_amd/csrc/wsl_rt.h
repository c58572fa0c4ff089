// Internal runtime shim shared by all kernel translation units.
// Product build: hipcc --offload-arch=gfx950.  Test-only build: tests/emul (lock-step host emulator,
// -DWSL_HOST_EMUL) compiles the very same sources to check kernel logic on a machine without a GPU.
#pragma once
#include <stdint.h>
#ifdef __cplusplus
#include <atomic>
#endif
#include <string.h>

#include "../../include/wsl_hip.h"
#include "wsl_debug.h"

#ifdef WSL_HOST_EMUL
#include "hip_emul.h"
#define WSL_LAUNCH(kern, grid, block, smem, stream, ...) \
  wsl_emu::launch(grid, block, smem, [&]() { kern(__VA_ARGS__); })
#define WSL_DYN_SMEM(name) unsigned char* name = wsl_emu::dyn_smem()
#define WSL_SET_MAX_DYN_SMEM(kern, bytes) 0
typedef wsl_v4f v4f;
#define WSL_MFMA16(a, b, c) wsl_emu_mfma16(a, b, c)
#define WSL_MFMA4(a, b, c) wsl_emu_mfma4(a, b, c)
#define WSL_LDS_DMA16(gsrc, lds_wave_base) wsl_emu_lds_dma16(gsrc, lds_wave_base)
#define WSL_WAIT_ALL()
#define WSL_LDS_BARRIER() __syncthreads()
#define WSL_LDS_DMA16_UNTRACKED_SO(base_uniform, byte_off, lds_wave_base) \
  wsl_emu_lds_dma16(reinterpret_cast<const unsigned char*>(base_uniform) + (byte_off), lds_wave_base)
#define WSL_VM_WAIT(n)
#define WSL_SCHED_BARRIER()
typedef wsl_emu_u4 wsl_u4;
typedef wsl_emu_u2 wsl_u2;
#define WSL_MFMA_F16(a, b, c) wsl_emu_mfma16x32_f16(a, b, c)
#define WSL_WAVE_UNIFORM(x) (x)
#define WSL_DS_READ_TR16(p) wsl_emu_ds_read_tr16(p)
#else
#include <hip/hip_runtime.h>
#define WSL_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#define WSL_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define WSL_SET_MAX_DYN_SMEM(kern, bytes) \
  hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
typedef float v4f __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4] (1 VGPR), B[k=l>>4][j=l&15] (1 VGPR), D col=l&15,row=(l>>4)*4+r.
#define WSL_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products; block = l >> 2, D[l][r] = A[4*block + r] * B[l] + C[l][r]
#define WSL_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0)
// global_load_lds_dwordx4: every active lane fetches 16 bytes at its own global address; they land at the wave-uniform LDS
// base + lane * 16 (probed: tools/probe_lds_dma.py), bypassing the VGPRs.  Counted in vmcnt; WSL_WAIT_ALL() before the
// barrier that publishes the tile.
#define WSL_LDS_DMA16(gsrc, lds_wave_base)                                                             \
  __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(gsrc),              \
                                   (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#define WSL_WAIT_ALL() __builtin_amdgcn_s_waitcnt(0)
// Workgroup barrier that publishes LDS contents ONLY: every LDS access of the wave is complete, then s_barrier.  __syncthreads() also
// waits for every global store and load of the wave (s_waitcnt vmcnt(0)): in a persistent kernel that drains the output stores of the
// tile just finished and the prefetch of the next one at every barrier.  Nothing written to GLOBAL memory before this barrier may be
// read by another wave of the workgroup after it.
#define WSL_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The same LDS DMA issued from inline assembly (M0 = LDS base of the wave, restored afterwards): hipcc does not know the instruction
// is in flight, so it places no wait in front of later LDS reads or barriers -- with the builtin it must assume that ANY later LDS read
// aliases the destination and waits for the DMA (and, vmcnt being an in-order counter, for everything older) right there.  The caller
// owns the wait: WSL_VM_WAIT(n) = at most n vector-memory instructions of this wave still in flight (issue order; loads, stores and
// DMAs count alike) before the barrier that publishes the block.  hipcc's own vmcnt waits stay correct: it only under-counts the
// instructions in flight, so it waits longer than needed, never shorter.
// The source is a wave-uniform base (scalar registers) + a 32-bit byte offset per lane: no 64-bit vector address arithmetic.
__device__ __forceinline__ void wsl_lds_dma16_untracked_so(const void* base_uniform, uint32_t byte_off, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  const uint64_t b = (uint64_t)(size_t)base_uniform;
  const uint64_t bs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(byte_off), "s"(bs), "s"(dst)
               : "memory");
}
#define WSL_LDS_DMA16_UNTRACKED_SO(base_uniform, byte_off, lds_wave_base) wsl_lds_dma16_untracked_so(base_uniform, byte_off, lds_wave_base)
#define WSL_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// keeps the instruction scheduler from moving LDS reads / MFMAs across a software-pipeline stage boundary
#define WSL_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// v_mfma_f32_16x16x32_f16: A[i = l & 15][k = 8 (l >> 4) + e], B[k = 8 (l >> 4) + e][j = l & 15] (8 halves = 4 VGPRs each, passed as
// four 32-bit words), D as WSL_MFMA16 (probed: tools/probe_sp.hip).  Products exact in fp32, fp32 accumulation, f16 subnormals kept.
// The split-precision kernels use the compiler's builtin and its scheduling: round 3 had wrapped these chains in inline assembly with
// fences and wait states because the full-size network gradient was wrong with the builtin -- the chains were never the problem
// (profiles/r4_sp_root_cause.md: every MFMA source / destination hazard is interlocked on gfx950; the wrong results came from a packed-f32
// op_sel form in the OTHER kernels on the CU, see WSL_DETACH32 below).
typedef uint32_t wsl_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t wsl_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 wsl_h8 __attribute__((ext_vector_type(8)));
typedef short wsl_s4 __attribute__((ext_vector_type(4)));
#define WSL_MFMA_F16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wsl_h8, (a)), __builtin_bit_cast(wsl_h8, (b)), (c), 0, 0, 0)
// a value that is the same in every lane of the wave, held in a scalar register (branches on it are real branches)
#define WSL_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// ds_read_b64_tr_b16: inside a group of 16 lanes, lane i receives as element e the (i & 3)-th half of the four contiguous halves
// at the (8-byte aligned) LDS address supplied by lane 4 e + (i >> 2) of the group -- a 4 x 16 block of halves read row-wise,
// delivered column-wise (probed: tools/probe_sp.hip)
#define WSL_DS_READ_TR16(p) \
  __builtin_bit_cast(wsl_u2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wsl_s4*)(p)))
#endif

// Tuning / ablation / probe switches exist only in the EXPERIMENTS build (build.sh exp -> tools/exp/libwslhip_exp.so, and
// the test-only host emulator).  The product library reads no environment variable: every switch is its default there, and
// the ablation paths (which produce WRONG results by design) are compiled out.
#ifdef WSL_EXPERIMENTS
#include <stdlib.h>
static inline int wsl_tune_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#define WSL_TUNE(name, dflt) wsl_tune_int(name, dflt)
#define WSL_ABLATED(p, bits) (((p).ablate & (bits)) != 0)
namespace wsl { int forced_wgrad_wgs(); }   // wsl_conv.hip: wsl_debug_wgrad_workgroups()
// compile-time phase ablations of a kernel (wrong results by design): an extra template parameter in the experiments build only
#define WSL_ABL_TPARAM , int ABL = 0
#define WSL_ABL_CONST
#else
#define WSL_TUNE(name, dflt) (dflt)
#define WSL_ABLATED(p, bits) (false)
namespace wsl { constexpr int forced_wgrad_wgs() { return 0; } }
#define WSL_ABL_TPARAM
#define WSL_ABL_CONST constexpr int ABL = 0;
#endif

namespace wsl {

void set_error(const char* fmt, ...);
// opt-in HIP-event bracketing of one launch (or one entry point's launches) on the stream they are enqueued on (wsl_api.hip).
// flops / bytes are the ALGORITHMIC ones; issued < 0 means "= flops".
enum ProfFam {
  PF_CONV_FWD = 0, PF_CONV_DGRAD = 1, PF_WGRAD_WINO = 2, PF_WGRAD_REDUCE = 3, PF_GATEDCRF = 4, PF_OTHER = 5, PF_WINO_FWD = 6,
  PF_WINO_DGRAD = 7, PF_WGRAD_DIRECT = 8, PF_BN_BWD = 9, PF_BN_FINALIZE = 10, PF_BILINEAR = 11, PF_POOL_FANIN = 12,
  PF_LOSS_HEAD = 13, PF_SGD = 14, PF_PREP = 15, PF_SP_FWD = 16, PF_SP_DGRAD = 17, PF_SP_WGRAD = 18, PF_SPARE = 19
};
void* prof_begin(int fam, double flops, double bytes, void* stream, double issued = -1.0);
void prof_end(void* tok, void* stream);
struct ProfScope {
  void *tok, *stream;
  ProfScope(int fam, double flops, double bytes, void* s, double issued = -1.0) : tok(prof_begin(fam, flops, bytes, s, issued)), stream(s) {}
  ~ProfScope() { prof_end(tok, stream); }
  ProfScope(const ProfScope&) = delete;
};
int check_launch(const char* what);
int device_cu_count();   // compute units of the current device (cached)
// Library-owned side stream for work that is independent of the caller's stream for a while (the second decoder of
// unet_cct).  fork: side waits for everything enqueued on `main` so far; join: `main` waits for the side stream.
// Both are event waits enqueued on the streams -- the host never blocks.  Returns null when disabled / emulated.
void* side_stream(void* main);
void set_concurrent(int on);
int stream_fork(void* main, void* side);
int stream_join(void* main, void* side);

#define WSL_REQUIRE(cond, ...)     \
  do {                             \
    if (!(cond)) {                 \
      wsl::set_error(__VA_ARGS__); \
      return WSL_EINVAL;           \
    }                              \
  } while (0)

constexpr int kWave = 64;
constexpr int kThreads = 256;  // every kernel in this library uses 4-wave workgroups

__device__ __forceinline__ float leaky(float z) { return z > 0.f ? z : WSL_LEAKY_SLOPE * z; }

// The loader transform on one float4, written on 2-wide native vectors so the compiler emits packed f32 math
// (v_pk_fma_f32 / v_pk_mul_f32): BN affine, LeakyReLU as max(z, slope*z) (bit-identical to leaky()), keep mask bytes
// (0 or 1) and scale as float factors.  `m` is the uchar4 of keep bytes read as one 32-bit word.
//
// HARDWARE RULE (gfx950, measured: profiles/r4_sp_root_cause.md, tools/probe_pk_opsel.hip): v_pk_fma_f32 / v_pk_mul_f32 whose op_sel
// takes the LOW half of src1 (or src2) from the HIGH register of the source pair return a wrong low half while another wave of the SIMD
// issues v_mfma_f32_16x16x32_f16 -- and with the split-precision conv path one decoder's f16-MFMA kernels run beside every other
// kernel of the step.  hipcc emits exactly that form when a broadcast operand is the .y of a float2 that was loaded as one 64-bit
// value (`x * {t.x, t.x} + {t.y, t.y}` -> v_pk_fma_f32 d, x, t, t op_sel:[0,0,1] op_sel_hi:[1,0,1]).  WSL_DETACH32 cuts a scalar
// loose from the register pair it was loaded in (a 32-bit copy), after which a broadcast is formed from the LOW register of a fresh
// pair (op_sel_hi only: measured clean).  tools/scan_vop3p.py (tests/test_abi.py) fails the build if an unsafe form is left anywhere.
typedef float wsl_v2f __attribute__((ext_vector_type(2)));
#ifdef WSL_HOST_EMUL
#define WSL_DETACH32(x)
#else
#define WSL_DETACH32(x) asm volatile("" : "+v"(x))
#endif
__device__ __forceinline__ void xform_bn_leaky(wsl_v2f& lo, wsl_v2f& hi, float sc, float sh) {
  WSL_DETACH32(sc);
  WSL_DETACH32(sh);
  const wsl_v2f sc2 = {sc, sc}, sh2 = {sh, sh};
  lo = __builtin_elementwise_fma(lo, sc2, sh2), hi = __builtin_elementwise_fma(hi, sc2, sh2);
  const wsl_v2f l2 = lo * WSL_LEAKY_SLOPE, h2 = hi * WSL_LEAKY_SLOPE;
  lo = __builtin_elementwise_max(lo, l2), hi = __builtin_elementwise_max(hi, h2);
}
__device__ __forceinline__ void xform_mask(wsl_v2f& lo, wsl_v2f& hi, uint32_t m, float es) {
  const wsl_v2f m01 = {(float)(m & 0xffu), (float)((m >> 8) & 0xffu)}, m23 = {(float)((m >> 16) & 0xffu), (float)(m >> 24)};
  lo = (lo * es) * m01, hi = (hi * es) * m23;
}

// the value the PREVIOUS lane of the wave holds (lane 0: unspecified) -- v_mov_b32_dpp wave_shr:1, one vector instruction, no LDS
// (tools/probe_lds5.hip, probe 3).  Every lane that reads must have an active left neighbour.
__device__ __forceinline__ float wsl_prev_lane(float v) {
#ifdef WSL_HOST_EMUL
  const int l = wsl_emu::lane();
  return __shfl(v, l > 0 ? l - 1 : 0);
#else
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
#endif
}

// Packed f32 adds whose two results take their operands from DIFFERENT halves of the source register pairs (VOP3P op_sel /
// neg_hi), so the Winograd transforms run two outputs per vector instruction with every result already in the register an
// MFMA operand wants -- the plain vector form needs v_mov_b32 to un-interleave and loses what it saved.  Same IEEE adds as
// the scalar form: bit-identical results.
//   mid(P, Q) = (P.y + Q.x, Q.x - P.y)   [op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]]
//   sd(q)     = (q.x + q.y, q.x - q.y)   [op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]]       (tools/probe_pk.py checks both)
// The compiler has no pattern for these forms, so they are inline assembly -- and the hazard recogniser does not look
// inside inline assembly: a vector write needs two wait states before an MFMA reads it as A / B operand, and an MFMA result
// needs far more before a vector instruction may read it (a first per-instruction version left ONE wait state between a
// packed add and the MFMA reading it and accumulated stale operands).  Hence (1) every transform is ONE asm block that ends
// in `s_nop 3` (four wait states: twice what was measured sufficient), so whatever follows is safe, and (2) the blocks only ever read registers loaded from LDS -- transforms of MFMA
// results (the output transforms) stay in C++, where the compiler sees the instructions and pads them itself.
#define WSL_PK_SUB " neg_lo:[0,1] neg_hi:[0,1]\n"
#define WSL_PK_MID " op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n"
#define WSL_PK_SD " op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n"

// B^T d B of one 4 x 4 patch given as rows of two register pairs each ((d0, d1), (d2, d3)); results as the pairs
// a[i] = (V[i][0], V[i][3]), b[i] = (V[i][1], V[i][2]): 16 packed adds instead of 32 scalar ones
__device__ __forceinline__ void wino_btdb_pk(const wsl_v2f (&lo)[4], const wsl_v2f (&hi)[4], wsl_v2f (&a)[4], wsl_v2f (&b)[4]) {
#ifdef WSL_HOST_EMUL
  const wsl_v2f tl[4] = {lo[0] - lo[2], lo[1] + lo[2], lo[2] - lo[1], lo[1] - lo[3]};
  const wsl_v2f th[4] = {hi[0] - hi[2], hi[1] + hi[2], hi[2] - hi[1], hi[1] - hi[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = tl[i] - th[i], b[i] = wsl_v2f{tl[i][1] + th[i][0], th[i][0] - tl[i][1]};
#else
  wsl_v2f t0, t1, t2, t3;   // the high halves' row combinations (dead after the block)
  asm("v_pk_add_f32 %0, %12, %14" WSL_PK_SUB      // a[i] <- tl[i] = row combinations of the low halves
      "v_pk_add_f32 %1, %13, %14\n"
      "v_pk_add_f32 %2, %14, %13" WSL_PK_SUB
      "v_pk_add_f32 %3, %13, %15" WSL_PK_SUB
      "v_pk_add_f32 %8, %16, %18" WSL_PK_SUB      // t[i] <- th[i]
      "v_pk_add_f32 %9, %17, %18\n"
      "v_pk_add_f32 %10, %18, %17" WSL_PK_SUB
      "v_pk_add_f32 %11, %17, %19" WSL_PK_SUB
      "v_pk_add_f32 %4, %0, %8" WSL_PK_MID        // b[i] = mid(tl[i], th[i])
      "v_pk_add_f32 %5, %1, %9" WSL_PK_MID
      "v_pk_add_f32 %6, %2, %10" WSL_PK_MID
      "v_pk_add_f32 %7, %3, %11" WSL_PK_MID
      "v_pk_add_f32 %0, %0, %8" WSL_PK_SUB        // a[i] = tl[i] - th[i] (in place)
      "v_pk_add_f32 %1, %1, %9" WSL_PK_SUB
      "v_pk_add_f32 %2, %2, %10" WSL_PK_SUB
      "v_pk_add_f32 %3, %3, %11" WSL_PK_SUB
      "s_nop 3"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(t0),
        "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
#endif
}

// A dY A^T of one 2 x 2 output-gradient tile given as its rows r0, r1, WITHOUT the two negations of A (folded into the
// caller's epilogue): rows q1 = r0 + r1, q2 = r0 - r1 (q0 = r0, q3 = r1 need no instruction) and m[i] = sd(q[i]);
// Z[4 i + {0, 3}] = q[i], Z[4 i + {1, 2}] = m[i].  6 packed adds instead of 12 scalar ones
__device__ __forceinline__ void wino_aya_pk(wsl_v2f r0, wsl_v2f r1, wsl_v2f& q1, wsl_v2f& q2, wsl_v2f (&m)[4]) {
#ifdef WSL_HOST_EMUL
  q1 = r0 + r1, q2 = r0 - r1;
  const wsl_v2f q[4] = {r0, q1, q2, r1};
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = wsl_v2f{q[i][0] + q[i][1], q[i][0] - q[i][1]};
#else
  asm("v_pk_add_f32 %0, %6, %7\n"
      "v_pk_add_f32 %1, %6, %7" WSL_PK_SUB
      "v_pk_add_f32 %2, %6, %6" WSL_PK_SD
      "v_pk_add_f32 %5, %7, %7" WSL_PK_SD
      "v_pk_add_f32 %3, %0, %0" WSL_PK_SD
      "v_pk_add_f32 %4, %1, %1" WSL_PK_SD
      "s_nop 3"
      : "=&v"(q1), "=&v"(q2), "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3])
      : "v"(r0), "v"(r1));
#endif
}

// element xi = 4 i + c of a transform held as the pairs a[i] = (X[i][0], X[i][3]), b[i] = (X[i][1], X[i][2]) (xi constant after unrolling)
__device__ __forceinline__ float wino_pick(const wsl_v2f (&a)[4], const wsl_v2f (&b)[4], int xi) {
  const int i = xi >> 2, c = xi & 3;
  return c == 0 ? a[i][0] : c == 3 ? a[i][1] : c == 1 ? b[i][0] : b[i][1];
}

// ---- split-precision operands (wsl_convsp.hip): x * 2^e = hi + lo, hi = the scaled value as f16 rounded TOWARD ZERO (a finite
// overflow saturates at 65504 instead of turning into inf), lo = the exact remainder as f16 (round to nearest): 21-22 significant
// bits, f16 subnormals included.  Two values per 32-bit word (element 0 in the low half).
#ifdef WSL_HOST_EMUL
__device__ __forceinline__ uint32_t sp_pkrtz(float x0, float x1) { return (uint32_t)wsl_emu_f2h_rtz(x0) | ((uint32_t)wsl_emu_f2h_rtz(x1) << 16); }
__device__ __forceinline__ uint32_t sp_pkrne(float x0, float x1) { return (uint32_t)wsl_emu_f2h_rne(x0) | ((uint32_t)wsl_emu_f2h_rne(x1) << 16); }
__device__ __forceinline__ float sp_h0(uint32_t u) { return wsl_emu_h2f((uint16_t)u); }
__device__ __forceinline__ float sp_h1(uint32_t u) { return wsl_emu_h2f((uint16_t)(u >> 16)); }
#else
typedef _Float16 wsl_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sp_pkrtz(float x0, float x1) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x0, x1)); }
__device__ __forceinline__ uint32_t sp_pkrne(float x0, float x1) {
  const wsl_h2 v = {(_Float16)x0, (_Float16)x1};      // v_cvt_pk_f16_f32
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float sp_h0(uint32_t u) { return (float)__builtin_bit_cast(wsl_h2, u)[0]; }
__device__ __forceinline__ float sp_h1(uint32_t u) { return (float)__builtin_bit_cast(wsl_h2, u)[1]; }
#endif
__device__ __forceinline__ void sp_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = sp_pkrtz(x0, x1);
  lo = sp_pkrne(x0 - sp_h0(hi), x1 - sp_h1(hi));
}
// power-of-two operand scale from the bit pattern of max |x| (a non-negative float): max * 2^e lands in [2^14, 2^15).  The same
// function runs where an operand is split and where the product is unscaled, so the pair always cancels exactly.
__host__ __device__ __forceinline__ int sp_exp_of(uint32_t amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xffu);
  if (be == 0) return 0;                         // all-zero (or subnormal) tensor
  const int e = 14 - (be - 127);
  return e > 100 ? 100 : (e < -100 ? -100 : e);  // 2^e and 2^-e stay normal floats
}
// A tensor's maximum is kept as WSL_SP_AMAX_SLOTS partial maxima (a producer's workgroups spread their atomic max over the
// slots: tens of thousands of atomics on ONE address serialise at ~12 ns each); a consumer wave folds them with one load per lane
__device__ __forceinline__ uint32_t sp_amax_fold(const uint32_t* slots) {
  uint32_t u = slots[threadIdx.x & (WSL_SP_AMAX_SLOTS - 1)];
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)u, k);
    u = o > u ? o : u;
  }
  return u;
}
__host__ __device__ __forceinline__ float sp_pow2(int e) {
  const uint32_t u = (uint32_t)(e + 127) << 23;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Sum over the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Sum over the workgroup (256 threads); result valid in every thread. `red` is >= 4 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// The loader transform of WslSrc (see wsl_hip.h).  `idx` = element index inside the dense [N,C,H,W] view
// (used for emask), `xi` = element index in the (possibly batch-strided) x.
__device__ __forceinline__ float src_value(const WslSrc& s, int n, int c, int64_t xi, int64_t idx) {
  float v = s.x[xi];
  if (s.scale) v = leaky(fmaf(v, s.scale[c], s.shift[c]));
  if (s.emask) v = s.emask[idx] ? v * s.emask_scale : 0.f;
  if (s.cmask) v *= s.cmask[(int64_t)n * s.C + c];
  return v;
}

// BatchNorm + LeakyReLU (+ Dropout) BACKWARD statistics fused into the epilogue of the kernel that PRODUCES g = dL/d(block
// activation): per tile and channel  sum(dz)  and  sum(dz * xhat),  dz = g * leaky'(bn(y)) [* keep * scale], xhat = (y - mean) *
// invstd -- the reduction pass of wsl_bnact_bwd (8 bytes per element of HBM traffic) without its own launch.  Same per-element
// arithmetic as bn_dz4() in wsl_bn.hip (identical sign decisions at the LeakyReLU kink).
struct BnBwdEpi {
  const float* y = nullptr;        // raw conv output the BatchNorm normalised, dense [N][C][H][W]
  const float* st = nullptr;       // mean | invstd | scale | shift (4 * C floats)
  const uint8_t* emask = nullptr;  // keep mask of the nn.Dropout after the activation (or null)
  float es = 1.f;
  float* part = nullptr;           // [C][tiles][2]; null = no statistics
  bool store_d = false;            // the launch may write d = g * (keep byte * scale) * leaky'(z) -- the gradient in front of the BatchNorm
                                   // output, which the epilogue forms for its sums anyway -- instead of g (wsl_conv2d_dgrad_bn_d: round 6)
};
// algorithmic HBM bytes the BatchNorm-backward statistics epilogue adds to a data-gradient launch over `elems` output elements: one read
// of the consumer layer's raw output y (4 B) and of its keep mask (1 B) per element -- counted in the launch's algorithmic bytes (the
// profiling records bench.py's roofline.traffic is compared with: VERDICT r5 weak 6)
static inline double bn_epi_bytes(const BnBwdEpi& e, double elems) { return e.part ? elems * (4.0 + (e.emask ? 1.0 : 0.0)) : 0.0; }

// four consecutive gradient values g of channel statistics (mean, invstd, sc, sh) at dense element index idx, accumulated
// into PAIRS of partial sums (even / odd elements) so the arithmetic runs on packed f32 instructions -- 7 instead of 12 vector
// instructions per element; bn_bwd_fold() gives the lane's two sums
struct BnBwdAcc {
  wsl_v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
};
// (yv, m: the four conv outputs and keep bytes at the gradient's position -- a caller with registers to spare loads them
//  early, bn_bwd_acc4 loads them here)
// (dout: where the caller wants d itself, the four values in element order)
__device__ __forceinline__ void bn_bwd_acc4v(const BnBwdEpi& e, float4 yv, uint32_t m, float g0, float g1, float g2, float g3,
                                             float mean, float invstd, float sc, float sh, BnBwdAcc& a, float* dout = nullptr) {
  const wsl_v2f y[2] = {{yv.x, yv.y}, {yv.z, yv.w}};
  wsl_v2f g[2] = {{g0, g1}, {g2, g3}};
  if (e.emask) {   // keep bytes are 0 or 1: (g * scale) * byte == the selected value
    const wsl_v2f m01 = {(float)(m & 0xffu), (float)((m >> 8) & 0xffu)}, m23 = {(float)((m >> 16) & 0xffu), (float)(m >> 24)};
    g[0] = (g[0] * e.es) * m01, g[1] = (g[1] * e.es) * m23;
  }
  const wsl_v2f sc2 = {sc, sc}, sh2 = {sh, sh}, mean2 = {mean, mean};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const wsl_v2f z = __builtin_elementwise_fma(y[h], sc2, sh2);
    const wsl_v2f xh = (y[h] - mean2) * invstd;
    const wsl_v2f gl = g[h] * WSL_LEAKY_SLOPE;
    const wsl_v2f d = {z[0] > 0.f ? g[h][0] : gl[0], z[1] > 0.f ? g[h][1] : gl[1]};
    a.s1 += d;
    a.s2 = __builtin_elementwise_fma(d, xh, a.s2);
    if (dout) dout[2 * h] = d[0], dout[2 * h + 1] = d[1];
  }
}
__device__ __forceinline__ void bn_bwd_acc4(const BnBwdEpi& e, int64_t idx, float g0, float g1, float g2, float g3, float mean,
                                            float invstd, float sc, float sh, BnBwdAcc& a) {
  const float4 yv = *reinterpret_cast<const float4*>(e.y + idx);
  const uint32_t m = e.emask ? *reinterpret_cast<const uint32_t*>(e.emask + idx) : 0u;
  bn_bwd_acc4v(e, yv, m, g0, g1, g2, g3, mean, invstd, sc, sh, a);
}
__device__ __forceinline__ void bn_bwd_fold(const BnBwdAcc& a, float& s1, float& s2) { s1 = a.s1[0] + a.s1[1], s2 = a.s2[0] + a.s2[1]; }

// merge the per-lane sums of a 4-wave workgroup whose lanes (l & 15) own channel column col = j * 16 + (l & 15): lane groups
// (l >> 4), then waves (through `red`, >= 8 * CO_T floats), then one store per channel into part[C][nb][2]
// (LDS_ONLY: the barrier publishes `red` without waiting for the caller's global stores and prefetches -- WSL_LDS_BARRIER)
template <int NT, int CO_T, bool LDS_ONLY = false>
__device__ __forceinline__ void bn_bwd_store(const BnBwdEpi& e, float (&s1)[NT], float (&s2)[NT], float* red, int co0, int Co,
                                             int tile_id, int nb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* r1 = red;
  float* r2 = red + 4 * CO_T;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float a = s1[j], b = s2[j];
    a += __shfl_xor(a, 16), b += __shfl_xor(b, 16);
    a += __shfl_xor(a, 32), b += __shfl_xor(b, 32);
    if (lane < 16) r1[wave * CO_T + j * 16 + lane] = a, r2[wave * CO_T + j * 16 + lane] = b;
  }
  if constexpr (LDS_ONLY) WSL_LDS_BARRIER();
  else __syncthreads();
  if (wave == 0 && lane < 16) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = j * 16 + lane, co = co0 + col;
      if (co < Co) {
        float* dst = e.part + ((int64_t)co * nb + tile_id) * 2;
        dst[0] = (r1[col] + r1[CO_T + col]) + (r1[2 * CO_T + col] + r1[3 * CO_T + col]);
        dst[1] = (r2[col] + r2[CO_T + col]) + (r2[2 * CO_T + col] + r2[3 * CO_T + col]);
      }
    }
  }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// table of the conv layers of one network (kernel argument of pack_table_kernel, wsl_conv2.hip)
struct PackEntry {
  int64_t w;       // offset of the raw [Co][Ci][ks][ks] weight in the parameter arena (= offset of its packed images)
  int Co, Ci, KK;
  int _pad;        // pack_table_kernel: bit 0 / bit 1 = skip the direct forward / data-gradient image (nobody reads it)
};
struct PackTable {
  int n;
  PackEntry e[40];
};
int conv2_pack_table(const PackTable& t, const float* params, float* packf, float* packd, int with_dgrad, void* stream);
// Winograd filter images of the 3x3 entries (wsl_conv5.hip); image of entry e at uf/ud + 2 * e.w
int wino_pack_table(const PackTable& t, const float* params, float* uf, float* ud, int with_dgrad, void* stream);

}  // namespace wsl
