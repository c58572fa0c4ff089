// Error reporting / version entry points of libwslhip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "wsl_rt.h"

namespace wsl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return WSL_EHIP;
  }
  return WSL_OK;
}

int device_cu_count() {
#ifdef WSL_HOST_EMUL
  return 1;   // few workgroups: the emulator tests then exercise the multi-tile (persistent) path
#else
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
#endif
}

#ifndef WSL_HOST_EMUL
// One side stream + fork/join event pair per (device, caller stream): two networks driven from different streams (or on
// different devices) never share them; networks enqueued on the same caller stream are ordered by it anyway.  The table is the
// library's only process-global mutable state besides the profiling records; it is guarded by a mutex.
struct SideCtx { int dev; hipStream_t main, side; hipEvent_t fork, join; };
static std::vector<SideCtx> g_sides;
static std::mutex g_side_mu;
static int g_conc = 1;   // wsl_net_concurrent(): 1 on (default), 0 off
void set_concurrent(int on) { g_conc = on ? 1 : 0; }
static SideCtx* side_ctx(void* main) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  for (auto& c : g_sides)
    if (c.dev == dev && c.main == (hipStream_t)main) return &c;
  SideCtx c{dev, (hipStream_t)main, nullptr, nullptr, nullptr};
  // (experiments build: WSL_SIDE_PRIO = 1 / -1 creates the side stream at the highest / lowest stream priority -- an asymmetry between the
  //  two decoder streams, whose kernel sequences are identical and otherwise run in lock-step; measured: profiles/r6_side_stream_priority.md)
  static const int side_prio = WSL_TUNE("WSL_SIDE_PRIO", 0);
  int plo = 0, phi = 0;
  hipError_t made = hipErrorUnknown;
  if (side_prio != 0 && hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess)
    made = hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, side_prio > 0 ? phi : plo);
  if ((made != hipSuccess && hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) ||
      hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess)
    return nullptr;
  g_sides.reserve(64);     // pointers into the table stay valid: it never grows past its reservation
  if (g_sides.size() >= 64) return nullptr;
  g_sides.push_back(c);
  return &g_sides.back();
}
void* side_stream(void* main) {
  if (!g_conc) return nullptr;
  SideCtx* c = side_ctx(main);
  return c ? (void*)c->side : nullptr;
}
int stream_fork(void* main, void* side) {
  SideCtx* c = side_ctx(main);
  if (!c || (void*)c->side != side || hipEventRecord(c->fork, (hipStream_t)main) != hipSuccess ||
      hipStreamWaitEvent((hipStream_t)side, c->fork, 0) != hipSuccess) {
    set_error("stream_fork: HIP error");
    return WSL_EHIP;
  }
  return WSL_OK;
}
int stream_join(void* main, void* side) {
  SideCtx* c = side_ctx(main);
  if (!c || (void*)c->side != side || hipEventRecord(c->join, (hipStream_t)side) != hipSuccess ||
      hipStreamWaitEvent((hipStream_t)main, c->join, 0) != hipSuccess) {
    set_error("stream_join: HIP error");
    return WSL_EHIP;
  }
  return WSL_OK;
}
#else
void set_concurrent(int) {}
void* side_stream(void*) { return nullptr; }
int stream_fork(void*, void*) { return WSL_OK; }
int stream_join(void*, void*) { return WSL_OK; }
#endif

}  // namespace wsl

// ---------------------------------------------------------------------------------------------- opt-in profiling
// HIP events around the launches of the heavy kernel families, recorded on the stream the kernel is launched on
// (what bench.py's roofline object is computed from).  Off by default; wsl_prof_report() is the only call of the
// library that synchronises (on its own events).
namespace wsl {
#ifndef WSL_HOST_EMUL
struct ProfRec { int fam; double flops, bytes, issued; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static const char* kFamNames[WSL_PROF_FAMILIES] = {
    "conv_mfma2l_kernel(fwd)", "conv_mfma2l_kernel(dgrad)", "wgrad_wino_kernel", "wgrad_reduce_kernel", "gatedcrf_fwd_kernel",
    "other", "conv_wino2_kernel(fwd)", "conv_wino2_kernel(dgrad)", "wgrad_direct_kernels", "bnact_bwd(reduce+finalize+apply)",
    "bn_finalize_kernel", "bilinear_up2(fwd+bwd)", "pool2_fwd+feat_grad_combine", "loss_head(reduce+finalize+bwd+mix)", "sgd_kernel",
    "masks+filter_images", "conv_sp_kernel(fwd)", "conv_sp_kernel(dgrad)", "wgrad_sp_kernel", "spare"};
void* prof_begin(int fam, double flops, double bytes, void* stream, double issued) {
  if (!g_prof_on) return nullptr;
  ProfRec r{fam, flops, bytes, issued < 0.0 ? flops : issued, nullptr, nullptr};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  g_prof.push_back(r);
  return (void*)(uintptr_t)g_prof.size();
}
void prof_end(void* tok, void* stream) {
  if (!tok) return;
  (void)hipEventRecord(g_prof[(size_t)(uintptr_t)tok - 1].b, (hipStream_t)stream);
}
#else
void* prof_begin(int, double, double, void*, double) { return nullptr; }
void prof_end(void*, void*) {}
#endif
}  // namespace wsl

extern "C" int wsl_prof_enable(int on) {
#ifndef WSL_HOST_EMUL
  for (auto& r : wsl::g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  wsl::g_prof.clear();
  wsl::g_prof_on = on != 0;
#else
  (void)on;
#endif
  return WSL_OK;
}

extern "C" int wsl_prof_report(WslProfRow* rows, int max_rows) {
  WSL_REQUIRE(rows && max_rows >= WSL_PROF_FAMILIES, "prof_report: need %d rows", WSL_PROF_FAMILIES);
  for (int f = 0; f < WSL_PROF_FAMILIES; ++f) {
    memset(&rows[f], 0, sizeof(WslProfRow));
#ifndef WSL_HOST_EMUL
    snprintf(rows[f].name, sizeof(rows[f].name), "%s", wsl::kFamNames[f]);
#endif
  }
#ifndef WSL_HOST_EMUL
  for (auto& r : wsl::g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    rows[r.fam].calls += 1;
    rows[r.fam].ms += ms;
    rows[r.fam].flops += r.flops;
    rows[r.fam].bytes += r.bytes;
    rows[r.fam].issued_flops += r.issued;
  }
#endif
  return WSL_PROF_FAMILIES;
}

extern "C" int wsl_net_concurrent(int on) {
  wsl::set_concurrent(on);
  return WSL_OK;
}

extern "C" int wsl_version(void) { return 100; }
extern "C" const char* wsl_last_error(void) { return wsl::g_err; }
// WSL_SRC_SHA256: SHA-256 of the library's sources (csrc/*.hip, csrc/*.h sorted, include/wsl_hip.h), put on this file's command line by
// build.sh -- the binary names the tree it was built from (bench.py and tests/test_abi.py compare it with the tree: VERDICT r5 item 4a)
#ifndef WSL_SRC_SHA256
#define WSL_SRC_SHA256 "unknown"
#endif
extern "C" const char* wsl_build_info(void) {
#if defined(WSL_HOST_EMUL)
  return "HOST-EMULATION (tests only; not a product build) sha256:" WSL_SRC_SHA256;
#elif defined(WSL_EXPERIMENTS)
  return "gfx950 hipcc EXPERIMENTS (tuning tools / route-forcing tests only; not a product build) sha256:" WSL_SRC_SHA256;
#else
  return "gfx950 hipcc sha256:" WSL_SRC_SHA256;
#endif
}
