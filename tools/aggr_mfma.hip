// Aggressor kernels for tools/pair_race2.py (tools only; not part of the product library): streams of v_mfma_f32_16x16x32_f16 in
// the forms hipcc emits for the split-precision chains, one form per variant, so that the form which disturbs a co-resident f32
// MFMA kernel (profiles/r4_sp_root_cause.md) can be named.  The aggressors' own results are irrelevant.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/aggr_mfma.hip -o tools/exp/libaggr_mfma.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// fixed registers v[64:127]; A = v[64:67], B = v[68:71], accumulators / temporaries above
#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"

#define BODY_INPLACE                                                     \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[72:75]\n"     \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[76:79]\n"     \
  "v_mfma_f32_16x16x32_f16 v[80:83], v[64:67], v[68:71], v[80:83]\n"     \
  "v_mfma_f32_16x16x32_f16 v[84:87], v[64:67], v[68:71], v[84:87]\n"
// the same accumulator three times back to back (the hi*lo + lo*hi + hi*hi chain, in place)
#define BODY_INPLACE_CHAIN                                               \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[72:75]\n"     \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[72:75]\n"     \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[72:75]\n"     \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[76:79]\n"     \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[76:79]\n"     \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[76:79]\n"
// renamed chain: every link reads the previous link's vDst as SrcC and writes another tuple (hipcc's form), with the wait
// states hipcc places between the links (s_nop 0 / none: copied from its listing)
#define BODY_RENAMED                                                     \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[72:75]\n"     \
  "s_nop 0\n"                                                            \
  "v_mfma_f32_16x16x32_f16 v[80:83], v[64:67], v[68:71], v[76:79]\n"     \
  "s_nop 0\n"                                                            \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[80:83]\n"     \
  "s_nop 0\n"
#define BODY_RENAMED_NOPS                                                \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[72:75]\n"     \
  "s_nop 7\n"                                                            \
  "v_mfma_f32_16x16x32_f16 v[80:83], v[64:67], v[68:71], v[76:79]\n"     \
  "s_nop 7\n"                                                            \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[80:83]\n"     \
  "s_nop 7\n"
// destination = the A operand's tuple / the B operand's tuple (SrcC elsewhere)
#define BODY_DST_ON_A                                                    \
  "v_mfma_f32_16x16x32_f16 v[88:91], v[88:91], v[68:71], v[72:75]\n"     \
  "v_mfma_f32_16x16x32_f16 v[92:95], v[92:95], v[68:71], v[76:79]\n"
#define BODY_DST_ON_B                                                    \
  "v_mfma_f32_16x16x32_f16 v[88:91], v[64:67], v[88:91], v[72:75]\n"     \
  "v_mfma_f32_16x16x32_f16 v[92:95], v[64:67], v[92:95], v[76:79]\n"
// in-place chain with a DIFFERENT wave-visible density: LDS reads between the links
#define BODY_INPLACE_LDS                                                 \
  "ds_read_b128 v[88:91], %0\n"                                          \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[72:75]\n"     \
  "ds_read_b128 v[92:95], %0 offset:1024\n"                              \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[76:79]\n"     \
  "s_waitcnt lgkmcnt(0)\n"
// renamed link whose old accumulator is immediately re-used as an LDS read destination (the pattern of hipcc's listing)
#define BODY_RENAMED_LDS                                                 \
  "v_mfma_f32_16x16x32_f16 v[76:79], v[64:67], v[68:71], v[72:75]\n"     \
  "ds_read_b128 v[72:75], %0\n"                                          \
  "s_nop 0\n"                                                            \
  "v_mfma_f32_16x16x32_f16 v[80:83], v[64:67], v[68:71], v[76:79]\n"     \
  "ds_read_b128 v[76:79], %0 offset:1024\n"                              \
  "s_waitcnt lgkmcnt(0)\n"                                               \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[80:83]\n"

// controls: the f32 matrix instruction, plain vector FMAs, packed vector FMAs -- same loop, no f16 MFMA
#define BODY_F32MFMA                                                     \
  "v_mfma_f32_16x16x4_f32 v[72:75], v64, v68, v[72:75]\n"                \
  "v_mfma_f32_16x16x4_f32 v[76:79], v64, v68, v[76:79]\n"                \
  "v_mfma_f32_16x16x4_f32 v[80:83], v64, v68, v[80:83]\n"                \
  "v_mfma_f32_16x16x4_f32 v[84:87], v64, v68, v[84:87]\n"
#define BODY_VALU                                                        \
  "v_fma_f32 v72, v64, v68, v72\nv_fma_f32 v73, v64, v68, v73\nv_fma_f32 v74, v64, v68, v74\nv_fma_f32 v75, v64, v68, v75\n" \
  "v_fma_f32 v76, v64, v68, v76\nv_fma_f32 v77, v64, v68, v77\nv_fma_f32 v78, v64, v68, v78\nv_fma_f32 v79, v64, v68, v79\n" \
  "v_fma_f32 v80, v64, v68, v80\nv_fma_f32 v81, v64, v68, v81\nv_fma_f32 v82, v64, v68, v82\nv_fma_f32 v83, v64, v68, v83\n" \
  "v_fma_f32 v84, v64, v68, v84\nv_fma_f32 v85, v64, v68, v85\nv_fma_f32 v86, v64, v68, v86\nv_fma_f32 v87, v64, v68, v87\n"
// one f16 MFMA per 16 wait states: a sparse stream (a quarter of the matrix pipe per wave)
#define BODY_SPARSE                                                      \
  "v_mfma_f32_16x16x32_f16 v[72:75], v[64:67], v[68:71], v[72:75]\n"     \
  "s_nop 7\ns_nop 7\n"

#define KERNEL(NAME, BODY)                                                                           \
  __global__ __launch_bounds__(256, 2) void aggr_##NAME(int iters, float* sink) {                    \
    __shared__ __attribute__((aligned(16))) float lds[1024];                                         \
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = 0.001f * i;                               \
    __syncthreads();                                                                                 \
    const uint32_t addr = (uint32_t)(size_t)(&lds[(threadIdx.x & 63) * 4]);                          \
    asm volatile("v_mov_b32 v64, 0x3c003c00\nv_mov_b32 v65, 0x3c003c00\nv_mov_b32 v66, 0x3c003c00\nv_mov_b32 v67, 0x3c003c00\n" \
                 "v_mov_b32 v68, 0x38003800\nv_mov_b32 v69, 0x38003800\nv_mov_b32 v70, 0x38003800\nv_mov_b32 v71, 0x38003800\n" \
                 "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v74, 0\nv_mov_b32 v75, 0\nv_mov_b32 v76, 0\nv_mov_b32 v77, 0\n" \
                 "v_mov_b32 v78, 0\nv_mov_b32 v79, 0\nv_mov_b32 v80, 0\nv_mov_b32 v81, 0\nv_mov_b32 v82, 0\nv_mov_b32 v83, 0\n" \
                 "v_mov_b32 v84, 0\nv_mov_b32 v85, 0\nv_mov_b32 v86, 0\nv_mov_b32 v87, 0\nv_mov_b32 v88, 0\nv_mov_b32 v89, 0\n" \
                 "v_mov_b32 v90, 0\nv_mov_b32 v91, 0\nv_mov_b32 v92, 0\nv_mov_b32 v93, 0\nv_mov_b32 v94, 0\nv_mov_b32 v95, 0\ns_nop 7\n" \
                 ::: CLOB);                                                                          \
    for (int it = 0; it < iters; ++it) asm volatile(BODY BODY BODY BODY :: "v"(addr) : "memory", CLOB); \
    float r;                                                                                         \
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7\nv_mov_b32 %0, v72" : "=v"(r) :: CLOB);                  \
    if (r == 123.456f) sink[0] = r;                                                                  \
  }                                                                                                  \
  extern "C" int launch_aggr_##NAME(int blocks, int iters, float* sink, void* stream) {              \
    hipLaunchKernelGGL(aggr_##NAME, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);   \
    return (int)hipGetLastError();                                                                   \
  }

KERNEL(inplace, BODY_INPLACE)
KERNEL(inplace_chain, BODY_INPLACE_CHAIN)
KERNEL(renamed, BODY_RENAMED)
KERNEL(renamed_nops, BODY_RENAMED_NOPS)
KERNEL(dst_on_a, BODY_DST_ON_A)
KERNEL(dst_on_b, BODY_DST_ON_B)
KERNEL(inplace_lds, BODY_INPLACE_LDS)
KERNEL(renamed_lds, BODY_RENAMED_LDS)
KERNEL(f32mfma, BODY_F32MFMA)
KERNEL(valu, BODY_VALU)
KERNEL(sparse, BODY_SPARSE)
