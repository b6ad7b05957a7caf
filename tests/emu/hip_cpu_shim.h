// hip_cpu_shim.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal single-source CPU emulation of the HIP subset the csnet kernels use, so that the very
// same kernel and plan sources (sod100k_amd/csrc/*) can be compiled with g++ (-DCSN_CPU_EMU) and the
// index arithmetic of every kernel can be checked against the oracle in the build container, which
// has no GPU.  Threads of a block run as ucontext fibers scheduled round-robin between
// __syncthreads() points; blocks run in parallel under OpenMP.  Nothing here is ever loaded by the
// product package (sod100k_amd/_native.py only loads libcsnet_hip.so and raises if it is missing).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__

struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

using std::min;
using std::max;

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };

static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

namespace csn_emu {
struct Ctx {
  uint3 tIdx, bIdx;
  dim3 bDim, gDim;
  unsigned char* smem = nullptr;
  float xch[1024];            // cross-lane exchange table of the block being run (DPP emulation, k_ilb.hip)
  void* sched_sp = nullptr;   // saved stack pointer of the block scheduler (x86-64 fast path)
  ucontext_t sched;           // portable path
};
extern thread_local Ctx g;
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
// CSN_EMU_PROFILE=1: wall time and block count per kernel name, printed at exit (finds launches whose grid does not
// shrink with the problem -- they dominate the emulated tests and the small-batch GPU steps alike)
void launch_named(const char* name, dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void yield_barrier();
}  // namespace csn_emu

#define threadIdx (csn_emu::g.tIdx)
#define blockIdx (csn_emu::g.bIdx)
#define blockDim (csn_emu::g.bDim)
#define gridDim (csn_emu::g.gDim)
static inline void __syncthreads() { csn_emu::yield_barrier(); }
// atomics: fibers of a block are sequential, blocks run on OpenMP threads
template <class T>
static inline T atomicAdd(T* p, T v) {
  T old;
#pragma omp atomic capture
  { old = *p; *p += v; }
  return old;
}

#ifdef CSN_EMU_IMPL
#include <omp.h>
#include <chrono>
#include <map>
#include <string>
namespace csn_emu {
thread_local Ctx g;

// Fiber switch.  glibc's swapcontext saves and restores the signal mask with a system call per switch; a barrier of a
// 256-thread block is 256 switches, so the emulated suites spent most of their time there.  On x86-64 the switch is the
// classic callee-saved-register swap (no signal mask, no FP environment: kernels change neither).
#if defined(__x86_64__) && !defined(CSN_EMU_UCONTEXT)
#define CSN_EMU_FAST_SWITCH 1
extern "C" void csn_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl csn_emu_switch
.type csn_emu_switch,@function
csn_emu_switch:
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
.size csn_emu_switch,.-csn_emu_switch
)");
#endif

struct Fiber {
#ifdef CSN_EMU_FAST_SWITCH
  void* sp = nullptr;
#else
  ucontext_t ctx;
#endif
  bool done = false;
};
struct Pool {
  std::vector<Fiber> fibers;
  std::vector<unsigned char> stacks;
  std::vector<unsigned char> smem;
  const std::function<void()>* body = nullptr;
  Fiber* running = nullptr;
};
static thread_local Pool pool;
static const size_t kStack = 96 * 1024;

#ifdef CSN_EMU_FAST_SWITCH
static void to_sched() { csn_emu_switch(&pool.running->sp, g.sched_sp); }
static void to_fiber(Fiber& f) { csn_emu_switch(&g.sched_sp, f.sp); }
#else
static void to_sched() { swapcontext(&pool.running->ctx, &g.sched); }
static void to_fiber(Fiber& f) { swapcontext(&g.sched, &f.ctx); }
#endif

static void trampoline() {
  (*pool.body)();
  pool.running->done = true;
  to_sched();
  std::abort();   // a finished fiber is never resumed
}

void yield_barrier() { to_sched(); }

static void run_block(dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz, size_t smem_bytes,
                      const std::function<void()>& body) {
  const unsigned n = block.x * block.y * block.z;
  if (pool.fibers.size() < n) {
    pool.fibers.resize(n);
    pool.stacks.resize((size_t)n * kStack);
  }
  // CSN_EMU_STRICT_LDS=1: a fresh, exactly sized allocation per block, so that an address sanitizer build
  // (make ASAN=1) flags any access past the launch's dynamic LDS size instead of reading the pool's stale bytes
  static const bool strict_lds = std::getenv("CSN_EMU_STRICT_LDS") != nullptr;
  if (strict_lds) {
    std::vector<unsigned char>(smem_bytes + 16).swap(pool.smem);
  } else if (pool.smem.size() < smem_bytes + 64) {
    pool.smem.resize(smem_bytes + 64);
  }
  // 16-byte aligned dynamic LDS base; poison so that reads of unwritten LDS show up
  unsigned char* sm = pool.smem.data();
  sm += (16 - ((uintptr_t)sm & 15)) & 15;
  std::memset(sm, 0xFF, smem_bytes);
  g.smem = sm;
  g.bIdx = uint3{bx, by, bz};
  g.bDim = block;
  g.gDim = grid;
  pool.body = &body;
  // CSN_EMU_POISON_STACK=1: fill the part of every fiber stack a kernel frame lives in with 0xFF so that reads of
  // uninitialised locals (undefined on the GPU as well) show up as NaNs / huge integers instead of stale finite data
  static const bool poison_stack = std::getenv("CSN_EMU_POISON_STACK") != nullptr;
  for (unsigned t = 0; t < n; ++t) {
    Fiber& f = pool.fibers[t];
    f.done = false;
    if (poison_stack) std::memset(pool.stacks.data() + (size_t)(t + 1) * kStack - 48 * 1024, 0xFF, 48 * 1024 - 256);
#ifdef CSN_EMU_FAST_SWITCH
    // initial frame: six callee-saved registers, the entry point as return address, and a slot that keeps the stack
    // pointer at 8 (mod 16) on entry as the ABI requires after a call
    uintptr_t top = (uintptr_t)(pool.stacks.data() + (size_t)(t + 1) * kStack);
    top &= ~(uintptr_t)15;
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;
    *--sp = reinterpret_cast<void*>(&trampoline);
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    f.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = pool.stacks.data() + (size_t)t * kStack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
  }
  unsigned alive = n;
  while (alive) {
    for (unsigned t = 0; t < n; ++t) {
      Fiber& f = pool.fibers[t];
      if (f.done) continue;
      g.tIdx = uint3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
      pool.running = &f;
      to_fiber(f);
      if (f.done) --alive;
    }
  }
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
  const long nb = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic, 1)
  for (long b = 0; b < nb; ++b) {
    unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y), bz = (unsigned)(b / ((long)grid.x * grid.y));
    run_block(grid, block, bx, by, bz, smem_bytes, body);
  }
}

struct ProfEntry { double s = 0; long blocks = 0, launches = 0; };
static std::map<std::string, ProfEntry>* g_prof = nullptr;
static void prof_dump() {
  if (!g_prof) return;
  std::vector<std::pair<std::string, ProfEntry>> v(g_prof->begin(), g_prof->end());
  std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.s > b.second.s; });
  for (const auto& e : v)
    std::fprintf(stderr, "[emu-profile] %-48s %9.3f s %8ld launches %10ld blocks\n", e.first.c_str(), e.second.s,
                 e.second.launches, e.second.blocks);
}
void launch_named(const char* name, dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
  static const bool on = std::getenv("CSN_EMU_PROFILE") != nullptr;
  if (!on) { launch(grid, block, smem_bytes, body); return; }
  if (!g_prof) { g_prof = new std::map<std::string, ProfEntry>(); std::atexit(prof_dump); }
  const auto t0 = std::chrono::steady_clock::now();
  launch(grid, block, smem_bytes, body);
  ProfEntry& e = (*g_prof)[name];
  e.s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  e.blocks += (long)grid.x * grid.y * grid.z;
  e.launches += 1;
}
}  // namespace csn_emu
#endif  // CSN_EMU_IMPL
