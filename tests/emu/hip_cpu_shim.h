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
  ucontext_t sched;
  ucontext_t* cur = nullptr;
};
extern thread_local Ctx g;
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
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
namespace csn_emu {
thread_local Ctx g;

struct Fiber {
  ucontext_t ctx;
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

static void trampoline() {
  (*pool.body)();
  pool.running->done = true;
  swapcontext(&pool.running->ctx, &g.sched);
}

void yield_barrier() { swapcontext(&pool.running->ctx, &g.sched); }

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
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = pool.stacks.data() + (size_t)t * kStack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  unsigned alive = n;
  while (alive) {
    for (unsigned t = 0; t < n; ++t) {
      Fiber& f = pool.fibers[t];
      if (f.done) continue;
      g.tIdx = uint3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
      pool.running = &f;
      swapcontext(&g.sched, &f.ctx);
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
}  // namespace csn_emu
#endif  // CSN_EMU_IMPL
