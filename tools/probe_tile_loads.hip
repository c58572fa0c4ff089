// Machine probe (tools only): how fast can conv-shaped halo tiles be fetched at the full-resolution level, as a function of the
// tensor LAYOUT and of how many tiles a workgroup keeps in flight?  (round 4: the split conv kernels run at 3 TB/s on the 256 x 256
// layers; ablations say the time is the loads, not the arithmetic.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_tile_loads.hip -o tools/exp/probe_tile_loads && tools/exp/probe_tile_loads
// Every variant fetches, for each 8 x 32 output tile of an [N, C, 256, 256] activation, the 10 x 40 (NCHW f32) or 10 x 34 (packed) input
// halo tile of all C channels and stores a 16-byte checksum per thread (so the loads are not dead) -- nothing else.
//   nchw      f32 planes: a thread loads float4s (4 pixels of one channel): 160-byte runs, one per (channel, row)   [today's loader]
//   packed    f16 hi / lo images, channel-innermost octets: 16 bytes per (pixel, octet): 544-byte runs per (octet, row, hi | lo)
//   *_dma     the same bytes through global_load_lds_dwordx4 into an LDS ring of DEPTH tile buffers (no registers held)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;

// LDS DMA the compiler does not see (it would drain every outstanding DMA -- vmcnt(0) -- at each __syncthreads(), which defeats a ring
// of buffers): M0 = wave-uniform LDS byte address, saved / restored around the instruction
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// workgroup barrier without the fence of __syncthreads() (no wait for the DMAs still in flight)
__device__ __forceinline__ u4 lds_read16(const void* p) {   // (inline assembly: a plain LDS read, not counted against the DMAs in flight)
  u4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}
__device__ __forceinline__ void bare_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void tile_of(int t, int tiles_x, int tiles_y, int& n, int& y0, int& x0) {
  const int tx = t % tiles_x, r = t / tiles_x;
  n = r / tiles_y, y0 = (r % tiles_y) * TH, x0 = tx * TW;
}

// today's pattern: tasks (row, quad, octet) -> eight float4 loads (one per channel of the octet)
template <int C>
__global__ __launch_bounds__(256, 2) void nchw_regs(const float* x, u4* out, int N, int H, int W, int ntiles) {
  const int tid = threadIdx.x, tiles_x = W / TW, tiles_y = H / TH;
  u4 acc = {0, 0, 0, 0};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int n, y0, x0;
    tile_of(t, tiles_x, tiles_y, n, y0, x0);
    for (int c0 = 0; c0 < C; c0 += 16) {
      const int q = tid % 10, rest = tid / 10, row = rest % 10, o = rest / 10;
      const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * q;
      if (tid < 200 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const float* p = x + ((size_t)(n * C + c0 + 8 * o) * H + gy) * W + gx;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const u4 v = *reinterpret_cast<const u4*>(p + (size_t)c * H * W);
          acc ^= v;
        }
      }
    }
  }
  out[blockIdx.x * 256 + tid] = acc;
}

// the same loader with other tile shapes (rows x columns of OUTPUT pixels; the fetched tile is (TH_ + 2) x (TW_ + 8)) and, EXACT, with the
// column halo fetched as single floats instead of whole quads ((TW_ + 2) columns: what the convolution actually needs)
template <int C, int TH_, int TW_>
__global__ __launch_bounds__(256, 2) void nchw_shape(const float* x, u4* out, int N, int H, int W, int ntiles) {
  const int tid = threadIdx.x, tiles_x = W / TW_, tiles_y = H / TH_;
  constexpr int NQ = (TW_ + 8) / 4, ROWS = TH_ + 2, NTASK = ROWS * NQ * 2;
  u4 acc = {0, 0, 0, 0};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tx = t % tiles_x, r0 = t / tiles_x, n = r0 / tiles_y, y0 = (r0 % tiles_y) * TH_, x0 = tx * TW_;
    for (int c0 = 0; c0 < C; c0 += 16) {
      for (int k = tid; k < NTASK; k += 256) {
        const int q = k % NQ, rest = k / NQ, row = rest % ROWS, o = rest / ROWS;
        const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * q;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          const float* p = x + ((size_t)(n * C + c0 + 8 * o) * H + gy) * W + gx;
#pragma unroll
          for (int c = 0; c < 8; ++c) acc ^= *reinterpret_cast<const u4*>(p + (size_t)c * H * W);
        }
      }
    }
  }
  out[blockIdx.x * 256 + tid] = acc;
}

// packed: slot (hl, n, octet, y, x) = 16 bytes; a tile chunk of 16 channels = 4 planes x 10 rows x 34 slots = 1360 slots
template <int C>
__global__ __launch_bounds__(256, 2) void packed_regs(const u4* x, u4* out, int N, int H, int W, int ntiles) {
  const int tid = threadIdx.x, tiles_x = W / TW, tiles_y = H / TH;
  const size_t HLS = (size_t)N * (C / 8) * H * W;   // slots per hi / lo image
  u4 acc = {0, 0, 0, 0};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int n, y0, x0;
    tile_of(t, tiles_x, tiles_y, n, y0, x0);
    for (int c0 = 0; c0 < C; c0 += 16) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int s = tid + i * 256;
        if (s < 1360) {
          const int col = s % 34, r2 = s / 34, row = r2 % 10, pl = r2 / 10;   // pl = hl * 2 + octet
          const int gy = y0 - 1 + row, gx = x0 - 1 + col;
          if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const u4 v = x[(pl >> 1) * HLS + ((size_t)(n * (C / 8) + c0 / 8 + (pl & 1)) * H + gy) * W + gx];
            acc ^= v;
          }
        }
      }
    }
  }
  out[blockIdx.x * 256 + tid] = acc;
}

// packed through LDS DMA, DEPTH tile-chunks in flight per workgroup
template <int C, int DEPTH>
__global__ __launch_bounds__(256, 2) void packed_dma(const u4* x, const u4* zero, u4* out, int N, int H, int W, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // DEPTH x 24 KB
  constexpr int STAGE = 1536 * 16;                                        // 24 wave-instructions of 64 slots (6 per wave)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tiles_x = W / TW, tiles_y = H / TH;
  const size_t HLS = (size_t)N * (C / 8) * H * W;
  constexpr int NCH = C / 16;
  const int my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
  const int items = my * NCH;
  u4 acc = {0, 0, 0, 0};
  auto issue = [&](int it) {
    const int t = blockIdx.x + (it / NCH) * gridDim.x, c0 = (it % NCH) * 16;
    int n, y0, x0;
    tile_of(t, tiles_x, tiles_y, n, y0, x0);
    unsigned char* st = smem + (it % DEPTH) * STAGE;
    for (int i = wave; i < 24; i += 4) {
      const int s = i * 64 + lane;
      const int col = s % 34, r2 = s / 34, row = r2 % 10, pl = r2 / 10;
      const int gy = y0 - 1 + row, gx = x0 - 1 + col;
      const u4* g = (s < 1360 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                        ? x + (pl >> 1) * HLS + ((size_t)(n * (C / 8) + c0 / 8 + (pl & 1)) * H + gy) * W + gx
                        : zero;
      dma16(g, st + i * 1024);
    }
  };
  for (int it = 0; it < DEPTH - 1 && it < items; ++it) issue(it);
  for (int it = 0; it < items; ++it) {
    if (it + DEPTH - 1 < items) issue(it + DEPTH - 1);
    // wait for item `it`: everything but the DMA instructions of the (up to DEPTH - 1) later items (6 or 5 per wave and item)
    if (DEPTH == 1 || it + DEPTH - 1 >= items) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail: no later item was issued)
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    bare_barrier();
    acc ^= lds_read16(smem + (it % DEPTH) * STAGE + tid * 16);   // (touch the data)
    bare_barrier();
  }
  out[blockIdx.x * 256 + tid] = acc;
}

// NCHW f32 through LDS DMA (a thread's own eight float4s land in its own slots), DEPTH chunks in flight
template <int C, int DEPTH>
__global__ __launch_bounds__(256, 2) void nchw_dma(const float* x, const u4* zero, u4* out, int N, int H, int W, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // DEPTH x 32 KB
  constexpr int STAGE = 8 * 256 * 16;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), tiles_x = W / TW, tiles_y = H / TH;
  constexpr int NCH = C / 16;
  const int my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int items = my * NCH;
  u4 acc = {0, 0, 0, 0};
  auto issue = [&](int it) {
    const int t = blockIdx.x + (it / NCH) * gridDim.x, c0 = (it % NCH) * 16;
    int n, y0, x0;
    tile_of(t, tiles_x, tiles_y, n, y0, x0);
    unsigned char* st = smem + (it % DEPTH) * STAGE;
    const int q = tid % 10, rest = tid / 10, row = rest % 10, o = rest / 10;
    const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * q;
    const bool ok = tid < 200 && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* p = x + ((size_t)(n * C + c0 + 8 * (o & 1)) * H + (ok ? gy : 0)) * W + (ok ? gx : 0);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const void* g = ok ? (const void*)(p + (size_t)c * H * W) : (const void*)zero;
      dma16(g, st + (c * 256 + wave * 64) * 16);
    }
  };
  for (int it = 0; it < DEPTH - 1 && it < items; ++it) issue(it);
  for (int it = 0; it < items; ++it) {
    if (it + DEPTH - 1 < items) issue(it + DEPTH - 1);
    if (DEPTH == 1 || it + DEPTH - 1 >= items) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    bare_barrier();
    acc ^= lds_read16(smem + (it % DEPTH) * STAGE + tid * 16);
    bare_barrier();
  }
  out[blockIdx.x * 256 + tid] = acc;
}

template <typename F>
static float timeit(F f) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 100.f;   // us per launch
}

template <int C>
static void run(int N, int H, int W, int cus) {
  const size_t elems = (size_t)N * C * H * W;
  float* x;
  u4 *out, *zero;
  (void)hipMalloc(&x, elems * 4);
  (void)hipMemset(x, 0x11, elems * 4);
  (void)hipMalloc(&out, (size_t)4096 * 256 * 16);
  (void)hipMalloc(&zero, 64);
  (void)hipMemset(zero, 0, 64);
  const int ntiles = N * (H / TH) * (W / TW);
  const double mb = elems * 4 / 1e6;
  printf("[N=%d C=%d %dx%d: %.0f MB tensor, %d tiles]\n", N, C, H, W, mb, ntiles);
  for (int per_cu : {2, 3, 4}) {
    const int g = per_cu * cus;
    float us = timeit([&] { hipLaunchKernelGGL(nchw_regs<C>, dim3(g), dim3(256), 0, 0, x, out, N, H, W, ntiles); });
    printf("  nchw   registers        %d WG/CU: %7.1f us  %5.2f TB/s of the tensor\n", per_cu, us, mb / us);
    us = timeit([&] { hipLaunchKernelGGL(packed_regs<C>, dim3(g), dim3(256), 0, 0, (const u4*)x, out, N, H, W, ntiles); });
    printf("  packed registers        %d WG/CU: %7.1f us  %5.2f TB/s\n", per_cu, us, mb / us);
  }
#define SHAPE(TH_, TW_)                                                                                                         \
  {                                                                                                                            \
    const int nt = N * (H / TH_) * (W / TW_);                                                                                  \
    for (int per_cu : {2, 3}) {                                                                                                \
      float us = timeit([&] { hipLaunchKernelGGL((nchw_shape<C, TH_, TW_>), dim3(per_cu * cus), dim3(256), 0, 0, x, out, N, H, W, nt); }); \
      printf("  nchw   registers tile %2d x %-3d (fetch %.2fx) %d WG/CU: %7.1f us  %5.2f TB/s\n", TH_, TW_,                    \
             (double)(TH_ + 2) * (TW_ + 8) / (TH_ * TW_), per_cu, us, mb / us);                                                \
    }                                                                                                                          \
  }
  if (W >= 64) { SHAPE(8, 64) SHAPE(16, 32) SHAPE(16, 64) SHAPE(4, 64) }
  if (W >= 128) { SHAPE(8, 128) SHAPE(4, 128) }
#define DMA(K, D, PER)                                                                                                          \
  {                                                                                                                            \
    const size_t sm = (size_t)D * (K == 0 ? 1536 * 16 : 8 * 256 * 16);                                                          \
    if (K == 0) (void)hipFuncSetAttribute((const void*)packed_dma<C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);  \
    else (void)hipFuncSetAttribute((const void*)nchw_dma<C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);           \
    float us = timeit([&] {                                                                                                    \
      if (K == 0) hipLaunchKernelGGL((packed_dma<C, D>), dim3(PER * cus), dim3(256), sm, 0, (const u4*)x, zero, out, N, H, W, ntiles); \
      else hipLaunchKernelGGL((nchw_dma<C, D>), dim3(PER * cus), dim3(256), sm, 0, x, zero, out, N, H, W, ntiles);             \
    });                                                                                                                        \
    printf("  %s LDS DMA depth %d   %d WG/CU: %7.1f us  %5.2f TB/s\n", K == 0 ? "packed" : "nchw  ", D, PER, us, mb / us);     \
  }
  DMA(0, 1, 2) DMA(0, 2, 2) DMA(0, 3, 2) DMA(0, 2, 1) DMA(0, 4, 1) DMA(0, 6, 1) DMA(0, 3, 3)
  DMA(1, 1, 2) DMA(1, 2, 2) DMA(1, 2, 1) DMA(1, 3, 1) DMA(1, 4, 1)
#undef DMA
  (void)hipFree(x), (void)hipFree(out), (void)hipFree(zero);
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs; halo-tile fetch only (8 x 32 output tiles)\n", prop.gcnArchName, cus);
  run<16>(64, 256, 256, cus);
  run<32>(64, 128, 128, cus);
  run<64>(64, 64, 64, cus);
  return 0;
}
