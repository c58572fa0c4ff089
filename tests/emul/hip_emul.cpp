// TEST INFRASTRUCTURE ONLY -- see hip_emul.h.
#include "hip_emul.h"

#include <stdlib.h>

#include <thread>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace wsl_emu {

static thread_local State g_state;   // workgroups of one launch are spread over host threads
State& st() { return g_state; }

// Minimal x86-64 SysV context switch: callee-saved GPRs + stack pointer.
asm(R"(
.text
.globl wsl_emu_switch
.type wsl_emu_switch,@function
wsl_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size wsl_emu_switch, .-wsl_emu_switch
)");

static const size_t kStack = 96 * 1024;
static thread_local std::vector<unsigned char> g_dyn;

unsigned char* dyn_smem() {
  if (g_dyn.empty()) g_dyn.resize(192 * 1024 + 64);
  uintptr_t p = (uintptr_t)g_dyn.data();
  return (unsigned char*)((p + 63) & ~(uintptr_t)63);
}

int lane() { return g_state.cur & 63; }
WaveState& wave() { return g_state.waves[g_state.cur >> 6]; }

void yield() {
  State& s = g_state;
  wsl_emu_switch(&s.fibers[s.cur].sp, s.sched_sp);
}

static int wave_live(int w) {
  State& s = g_state;
  int lo = w * 64, hi = lo + 64 < s.nthreads ? lo + 64 : s.nthreads, n = 0;
  for (int t = lo; t < hi; ++t) n += !s.fibers[t].done;
  return n;
}

void wave_sync() {
  State& s = g_state;
  WaveState& w = wave();
  unsigned long g = w.gen;
  w.arrived++;
  s.progress++;
  if (w.arrived >= wave_live(s.cur >> 6)) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == g) yield();
}

static void fiber_entry() {
  State& s = g_state;
  (*s.body)();
  s.fibers[s.cur].done = true;
  s.ndone++;
  s.progress++;
  // a thread that exits releases any barrier the rest is already waiting at
  if (s.bar_arrived > 0 && s.bar_arrived >= s.nthreads - s.ndone) {
    s.bar_arrived = 0;
    s.bar_gen++;
  }
  WaveState& w = s.waves[s.cur >> 6];
  if (w.arrived > 0 && w.arrived >= wave_live(s.cur >> 6)) {
    w.arrived = 0;
    w.gen++;
  }
  void* dummy;
  wsl_emu_switch(&dummy, s.sched_sp);
  abort();
}

static void run_block(const std::function<void()>& body) {
  State& s = g_state;
  int n = s.nthreads;
  s.fibers.assign(n, Fiber());
  s.waves.assign((n + 63) / 64, WaveState());
  if (s.stacks.size() < (size_t)n * kStack) s.stacks.resize((size_t)n * kStack);
  s.ndone = 0;
  s.bar_arrived = 0;
  s.body = &body;
  for (int t = 0; t < n; ++t) {
    uintptr_t top = ((uintptr_t)s.stacks.data() + (size_t)(t + 1) * kStack) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                // fake return address of fiber_entry (keeps rsp = 8 mod 16 at entry)
    *--sp = (void*)&fiber_entry;    // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    s.fibers[t].sp = sp;
  }
  unsigned bx = blockDim.x, by = blockDim.y;
  while (s.ndone < n) {
    unsigned long before = s.progress;
    for (int t = 0; t < n; ++t) {
      if (s.fibers[t].done) continue;
      s.cur = t;
      threadIdx = dim3(t % bx, (t / bx) % by, t / (bx * by));
      wsl_emu_switch(&s.sched_sp, s.fibers[t].sp);
    }
    if (s.progress == before) {
      fprintf(stderr, "wsl_emu: deadlock in block (%u,%u,%u): %d/%d threads done, barrier %d, no progress\n",
              blockIdx.x, blockIdx.y, blockIdx.z, s.ndone, n, s.bar_arrived);
      abort();
    }
  }
}

static int host_threads() {
  static int n = 0;
  if (n == 0) {
    const char* e = getenv("WSL_EMU_THREADS");
    n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (n > 16) n = 16;
  }
  return n;
}

// Workgroups are independent (the kernels use no inter-workgroup atomics or ordering), so a launch's blocks are dealt
// round-robin to a few host threads; inside a block the lock-step fiber schedule is unchanged.
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  if (smem > 160 * 1024) {
    fprintf(stderr, "wsl_emu: %zu bytes of dynamic LDS requested (> 160 KiB)\n", smem);
    abort();
  }
  const unsigned long total = (unsigned long)grid.x * grid.y * grid.z;
  int nt = host_threads();
  if ((unsigned long)nt > total) nt = (int)total;
  auto worker = [&](int w, int stride) {
    gridDim = grid;
    blockDim = block;
    g_state.nthreads = block.x * block.y * block.z;
    for (unsigned long b = w; b < total; b += stride) {
      blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long)grid.x * grid.y)));
      run_block(body);
    }
  };
  if (nt <= 1) {
    worker(0, 1);
    return;
  }
  std::vector<std::thread> pool;
  for (int w = 1; w < nt; ++w) pool.emplace_back(worker, w, nt);
  worker(0, nt);
  for (auto& t : pool) t.join();
}

}  // namespace wsl_emu
