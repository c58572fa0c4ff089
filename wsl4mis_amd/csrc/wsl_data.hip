// Data path, SURVEY 8f rank 2: the per-slice augmentation of the reference's RandomGenerator
// (code/dataloaders/dataset_semi.py:128-171: rot90+flip | rotate(order 0, constant) then zoom(order 0) to the network size)
// as ONE gather per output pixel on the device, for a whole batch of slices of different native sizes.
// Every step of the reference is a nearest-neighbour index map, so their composition is exact as a composed index map:
//   out[oy, ox] = step1[ Z(oy), Z(ox) ],  Z(o) = floor(o * (in-1)/(out-1) + 0.5)          (scipy.ndimage.zoom, order 0)
//   step1 = flip(rot90(img, k), axis)                                                      (numpy index permutations)
//         | img[floor(M (i,j) + off + 0.5)] if 0 <= M (i,j) + off <= n-1 else cval          (scipy.ndimage.rotate, order 0)
// Coordinates are computed in double without contraction, in the order that reproduces scipy bit for bit on the tests.
#include <stdint.h>

#include "wsl_rt.h"

namespace wsl {

constexpr int kAugMax = 32;   // samples per launch: the descriptor table travels as a kernel argument (32 * 96 + 8 B < 4 KB)
struct AugTable {
  int n;
  WslAugSample s[kAugMax];
};

#ifdef WSL_HOST_EMUL
static inline double dmul(double a, double b) { return a * b; }   // emulator TU is built with -ffp-contract=off
static inline double dadd(double a, double b) { return a + b; }
#else
__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
#endif

__global__ __launch_bounds__(256) void augment_kernel(AugTable t, float* out_img, uint8_t* out_lab, int Ho, int Wo) {
  const WslAugSample& s = t.s[blockIdx.y];
  const int h = s.h, w = s.w;
  // shape after step 1 (rot90 by an odd k swaps the axes)
  const bool swap = s.op == 1 && (s.k & 1);
  const int R = swap ? w : h, Cc = swap ? h : w;
  const double sy = Ho > 1 ? (double)(R - 1) / (double)(Ho - 1) : 0.0, sx = Wo > 1 ? (double)(Cc - 1) / (double)(Wo - 1) : 0.0;
  float* oi = out_img + (int64_t)blockIdx.y * Ho * Wo;
  uint8_t* ol = out_lab + (int64_t)blockIdx.y * Ho * Wo;
  for (int o = blockIdx.x * kThreads + threadIdx.x; o < Ho * Wo; o += gridDim.x * kThreads) {
    const int oy = o / Wo, ox = o - oy * Wo;
    int i = (int)floor(dadd(dmul((double)oy, sy), 0.5)), j = (int)floor(dadd(dmul((double)ox, sx), 0.5));
    i = i < 0 ? 0 : (i > R - 1 ? R - 1 : i), j = j < 0 ? 0 : (j > Cc - 1 ? Cc - 1 : j);
    int y = i, x = j;
    bool inside = true;
    if (s.op == 1) {
      if (s.axis == 0) i = R - 1 - i; else j = Cc - 1 - j;          // undo np.flip
      switch (s.k & 3) {                                            // undo np.rot90(m, k): r[i][j] = m[y][x]
        case 0: y = i, x = j; break;
        case 1: y = j, x = w - 1 - i; break;
        case 2: y = h - 1 - i, x = w - 1 - j; break;
        default: y = h - 1 - j, x = i; break;
      }
    } else if (s.op == 2) {
      const double cy = dadd(dadd(dmul(s.m00, (double)i), dmul(s.m01, (double)j)), s.off0);
      const double cx = dadd(dadd(dmul(s.m10, (double)i), dmul(s.m11, (double)j)), s.off1);
      inside = cy >= 0.0 && cy <= (double)(h - 1) && cx >= 0.0 && cx <= (double)(w - 1);
      y = (int)floor(dadd(cy, 0.5)), x = (int)floor(dadd(cx, 0.5));
    }
    oi[o] = inside ? s.img[(int64_t)y * w + x] : s.img_cval;
    ol[o] = inside ? s.lab[(int64_t)y * w + x] : (uint8_t)s.lab_cval;
  }
}

// ---- validation metric: the two device pieces of medpy.metric.binary.hd95
// surface of a binary volume = the object minus its erosion by the 6-neighbourhood with a background border
// (scipy.ndimage.binary_erosion(structure=generate_binary_structure(3, 1), border_value=0))
// D == 0: a 2-D [H,W] array, for which medpy's structure is the 4-neighbourhood (generate_binary_structure(2, 1)) -- not the
// same as a [1,H,W] volume, every voxel of which touches the background border along z.
__global__ __launch_bounds__(256) void surface_kernel(const uint8_t* vol, uint8_t* border, int D, int H, int W) {
  const bool flat = D == 0;
  if (flat) D = 1;
  const int64_t n = (int64_t)D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((int64_t)W * H));
    uint8_t b = 0;
    if (vol[i]) {
      const bool inner = y > 0 && y < H - 1 && x > 0 && x < W - 1 && vol[i - 1] && vol[i + 1] && vol[i - W] && vol[i + W] &&
                         (flat || (z > 0 && z < D - 1 && vol[i - (int64_t)W * H] && vol[i + (int64_t)W * H]));
      b = inner ? 0 : 1;
    }
    border[i] = b;
  }
}

// out[i] = min_j |a_i - b_j|^2 over integer voxel coordinates (exact in int64); b is streamed through LDS
__global__ __launch_bounds__(256) void nearest_d2_kernel(const int64_t* a, int na, const int64_t* b, int nb, int64_t* out) {
  __shared__ int bs[3 * 1024];
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const bool live = i < na;
  const int az = live ? (int)a[3 * (int64_t)i] : 0, ay = live ? (int)a[3 * (int64_t)i + 1] : 0, ax = live ? (int)a[3 * (int64_t)i + 2] : 0;
  int64_t best = INT64_MAX;
  for (int j0 = 0; j0 < nb; j0 += 1024) {
    const int m = nb - j0 < 1024 ? nb - j0 : 1024;
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * m; k += kThreads) bs[k] = (int)b[3 * (int64_t)j0 + k];
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      const int64_t dz = az - bs[3 * j], dy = ay - bs[3 * j + 1], dx = ax - bs[3 * j + 2];
      const int64_t d = dz * dz + dy * dy + dx * dx;
      best = d < best ? d : best;
    }
  }
  if (live) out[i] = best;
}

}  // namespace wsl

using namespace wsl;

extern "C" int wsl_surface_u8(const uint8_t* vol, uint8_t* border, int D, int H, int W, void* stream) {
  WSL_REQUIRE(vol && border && D >= 0 && H > 0 && W > 0, "surface_u8: bad arguments");
  const int64_t n = (int64_t)(D ? D : 1) * H * W;
  int64_t blocks = (n + kThreads - 1) / kThreads;
  if (blocks > 4096) blocks = 4096;
  WSL_LAUNCH(surface_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, vol, border, D, H, W);
  return check_launch("surface_kernel");
}

extern "C" int wsl_nearest_dist2(const int64_t* a_zyx, int na, const int64_t* b_zyx, int nb, int64_t* out, void* stream) {
  WSL_REQUIRE(a_zyx && b_zyx && out && na > 0 && nb > 0, "nearest_dist2: both point sets must be non-empty");
  WSL_LAUNCH(nearest_d2_kernel, dim3(cdiv(na, kThreads)), dim3(kThreads), 0, stream, a_zyx, na, b_zyx, nb, out);
  return check_launch("nearest_d2_kernel");
}

extern "C" int wsl_augment_batch(const WslAugSample* samples, int n, float* out_img, uint8_t* out_lab, int Ho, int Wo,
                                 void* stream) {
  WSL_REQUIRE(samples && out_img && out_lab && n > 0 && Ho > 0 && Wo > 0, "augment_batch: bad arguments");
  for (int base = 0; base < n; base += kAugMax) {
    AugTable t;
    t.n = n - base < kAugMax ? n - base : kAugMax;
    for (int k = 0; k < t.n; ++k) {
      const WslAugSample& s = samples[base + k];
      WSL_REQUIRE(s.img && s.lab && s.h > 0 && s.w > 0 && s.op >= 0 && s.op <= 2, "augment_batch: sample %d is malformed", base + k);
      t.s[k] = s;
    }
    const int bx = cdiv(Ho * Wo, kThreads * 4);
    WSL_LAUNCH(augment_kernel, dim3(bx, t.n), dim3(kThreads), 0, stream, t, out_img + (int64_t)base * Ho * Wo,
               out_lab + (int64_t)base * Ho * Wo, Ho, Wo);
  }
  return check_launch("augment_kernel");
}
