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
static const char* g_kernel_name = "?";   // of the launch in flight (diagnostics)
#ifdef CSN_EMU_LANES
static unsigned long long g_lane_ops[16];   // completed cross-lane instructions per kind (wave_xchg's op), all launches
#endif
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
// CSN_EMU_PROFILE=1: wall time and block count per kernel name, printed at exit (finds launches whose grid does not
// shrink with the problem -- they dominate the emulated tests and the small-batch GPU steps alike)
void launch_named(const char* name, dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void yield_barrier();
#ifdef CSN_EMU_LANES
// ---- lane-exact mode (make LANES=1 -> libcsnet_emu_lanes.so) ------------------------------------------------------------------
// The kernels' DEVICE paths are compiled: one lane's row of an MFMA A operand, halo columns by DPP, readfirstlane.  A cross-lane
// instruction is a rendezvous of the wave's 64 fibers: every lane deposits its source registers, waits until the wave is
// complete -- each other lane has deposited for the same instruction, has finished, or waits at a block barrier (= lanes the
// hardware would have masked off in EXEC) -- and then forms ITS result registers from the table by the instruction's lane map
// (the maps are pinned to the hardware by tests/test_gpu_lane_ops.py, which runs the real instructions next to these).
// __syncthreads() is a real barrier here (arrival counts), and a scheduler pass in which no fiber made progress is reported
// as a deadlock (e.g. a cross-lane instruction under lane-divergent control flow) instead of hanging.
struct Xchg { const unsigned char* tab; unsigned long long mask; };   // tab: [64 lanes][64 bytes]; mask: lanes that deposited
Xchg wave_xchg(int op, const void* site, const void* src, unsigned nbytes, bool need_all);
void lanes_mfma_f32_4x4x1(float a, float b, float* d4);
void lanes_mfma_f32_4x4x4_bf16(const void* a4, const void* b4, float* d4);
void lanes_mfma_f32_16x16x4_f32(float a, float b, float* d4);
void lanes_mfma_f32_32x32x16_bf16(const void* a8, const void* b8, float* d16);
void lanes_mfma_f32_16x16x32_bf16(const void* a8, const void* b8, float* d4);
unsigned lanes_dpp_wave_shr1(unsigned v);   // lane i <- lane i - 1 (row_shr across the wave, bound_ctrl: 0 where there is no source)
unsigned lanes_dpp_wave_shl1(unsigned v);   // lane i <- lane i + 1
unsigned lanes_readfirstlane(unsigned v);
unsigned long long lanes_shfl_xor64(unsigned long long v, int x);   // the 64-bit value of lane (lane ^ x): __shfl_xor(v, x, 64)
void lanes_wave_sync();
#endif
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
#include <dlfcn.h>
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
#ifdef CSN_EMU_LANES
struct OpenOp {              // a cross-lane instruction some lanes of a wave have arrived at
  const void* site = nullptr;  // its call instruction (lanes at different sites = lanes in different branches)
  int op = 0;
  bool need_all = false, complete = false;
  unsigned ndep = 0, nread = 0;
  unsigned long long mask = 0;
  alignas(16) unsigned char tab[64][64];
};
struct WaveSt {
  unsigned size = 64;        // lanes of this wave (the block's last wave may be short)
  unsigned completed = 0;    // cross-lane instructions completed so far
  unsigned nx = 0, nbar = 0, ndone = 0;   // lanes waiting in an open instruction / at the block barrier / finished
  std::vector<OpenOp*> open;
};
#endif
struct Pool {
  std::vector<Fiber> fibers;
  std::vector<unsigned char> stacks;
  std::vector<unsigned char> smem;
  const std::function<void()>* body = nullptr;
  Fiber* running = nullptr;
#ifdef CSN_EMU_LANES
  std::vector<WaveSt> waves;
  unsigned nthreads = 0, n_bar = 0, n_done = 0, bar_gen = 0;
  unsigned long progress = 0;
#endif
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

#ifndef CSN_EMU_LANES
void yield_barrier() { to_sched(); }
#else
static inline unsigned cur_thread() { return (unsigned)(pool.running - pool.fibers.data()); }
void yield_barrier() {
  WaveSt& w = pool.waves[cur_thread() >> 6];
  ++w.nbar; ++pool.n_bar; ++pool.progress;
  const unsigned gen = pool.bar_gen;
  for (;;) {
    if (pool.bar_gen != gen) break;
    if (pool.n_bar + pool.n_done == pool.nthreads) {   // the last arrival (or a poller after the last other fiber finished) opens it
      ++pool.bar_gen; pool.n_bar = 0; ++pool.progress;
      for (WaveSt& x : pool.waves) x.nbar = 0;
      break;
    }
    to_sched();
  }
}
[[noreturn]] static void lanes_fail(const char* what) {
  std::fprintf(stderr, "[emu-lanes] %s: %s (block %u,%u,%u thread %u)\n", g_kernel_name, what, g.bIdx.x, g.bIdx.y, g.bIdx.z, cur_thread());
  for (size_t i = 0; i < pool.waves.size(); ++i) {
    const WaveSt& w = pool.waves[i];
    std::fprintf(stderr, "[emu-lanes]   wave %zu: %u lanes in open instructions, %u at the barrier, %u finished of %u; %u instructions done\n",
                 i, w.nx, w.nbar, w.ndone, w.size, w.completed);
    for (const OpenOp* o : w.open)
      if (!o->complete) {
        Dl_info di;   // (addr2line -e libcsnet_emu_lanes.so -f -C -i <offset> names the source line)
        const unsigned long off = dladdr(o->site, &di) ? (unsigned long)((const char*)o->site - (const char*)di.dli_fbase) : 0ul;
        std::fprintf(stderr, "[emu-lanes]     open: op %d at +0x%lx, lanes %016llx\n", o->op, off, o->mask);
      }
  }
  std::abort();
}
// Lanes of a wave may sit at DIFFERENT instructions (an if without else that only some rows of lanes take, loops with lane-dependent
// trip counts): the hardware runs the branches one after the other under partial EXEC masks.  A CPU compiler knows nothing of
// convergence (g++ duplicates loop bodies, so one source instruction has several call addresses), hence the rule here: lanes
// waiting at the same KIND of instruction are in the same instruction, and an instruction completes when every lane of the wave is
// blocked (in an open instruction, at the block barrier) or finished.  Of several open kinds, those that tolerate missing lanes
// (DPP moves, readfirstlane, wave sync: the lanes of a divergent branch) go before a matrix instruction, which needs all 64.
Xchg wave_xchg(int op, const void* site, const void* src, unsigned nbytes, bool need_all) {
  const unsigned t = cur_thread(), lane = t & 63;
  WaveSt& w = pool.waves[t >> 6];
  OpenOp* o = nullptr;
  for (OpenOp* c : w.open)
    if (!c->complete && c->op == op) o = c;
  if (!o) {
    for (OpenOp* c : w.open)
      if (c->complete && c->nread == c->ndep) o = c;   // every participant has taken its result: recycle
    if (!o) { o = new OpenOp(); w.open.push_back(o); }
    o->site = site; o->op = op; o->need_all = need_all; o->complete = false; o->ndep = o->nread = 0; o->mask = 0;
  }
  if (nbytes) std::memcpy(o->tab[lane], src, nbytes);
  o->mask |= 1ull << lane;
  ++o->ndep; ++w.nx; ++pool.progress;
  for (;;) {
    if (o->complete) break;
    if (w.nx + w.nbar + w.ndone == w.size) {
      OpenOp* m = nullptr;
      for (OpenOp* c : w.open)
        if (!c->complete && (!m || (m->need_all && !c->need_all) || (m->need_all == c->need_all && c->op < m->op))) m = c;
      if (m->need_all && m->ndep != w.size) lanes_fail("a matrix instruction with lanes missing (in another branch, finished or at a barrier)");
      static const bool trace = std::getenv("CSN_EMU_LANES_TRACE") != nullptr;
      if (trace && g.bIdx.x == 0) std::fprintf(stderr, "[trace] %s wave %u op %d lanes %016llx\n", g_kernel_name, t >> 6, m->op, m->mask);
      m->complete = true; w.nx -= m->ndep; ++w.completed; ++pool.progress;
#pragma omp atomic
      ++g_lane_ops[m->op & 15];
      if (m == o) break;
      continue;   // (another instruction went first: look again -- this one may be complete now as well)
    }
    to_sched();
  }
  ++o->nread;
  return Xchg{&o->tab[0][0], o->mask};
}
static inline float bf16_to_f(unsigned short h) { const unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
// v_mfma_f32_4x4x1_16b_f32: 16 blocks of four lanes; lane l = (block l / 4, column l % 4) supplies A[row l % 4] and B[column l % 4]
// of its block and holds D[0 .. 3][l % 4]
__attribute__((noinline)) void lanes_mfma_f32_4x4x1(float a, float b, float* d) {
  const unsigned lane = cur_thread() & 63;
  const Xchg x = wave_xchg(1, __builtin_return_address(0), &a, 4, true);
  for (int i = 0; i < 4; ++i) {
    float ai; std::memcpy(&ai, x.tab + 64 * ((lane & ~3u) + i), 4);
    d[i] = fmaf(ai, b, d[i]);
  }
}
// v_mfma_f32_4x4x4_16b_bf16 (_1k): as above with four k values per lane (A[row][0 .. 3], B[0 .. 3][column])
__attribute__((noinline)) void lanes_mfma_f32_4x4x4_bf16(const void* a4, const void* b4, float* d) {
  const unsigned lane = cur_thread() & 63;
  const Xchg x = wave_xchg(2, __builtin_return_address(0), a4, 8, true);
  unsigned short bs[4]; std::memcpy(bs, b4, 8);
  for (int i = 0; i < 4; ++i) {
    unsigned short as[4]; std::memcpy(as, x.tab + 64 * ((lane & ~3u) + i), 8);
    float sum = d[i];
    for (int k = 0; k < 4; ++k) sum = fmaf(bf16_to_f(as[k]), bf16_to_f(bs[k]), sum);
    d[i] = sum;
  }
}
// v_mfma_f32_16x16x4_f32: lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16], holds D[4 (l / 16) + r][l % 16], r = 0 .. 3
__attribute__((noinline)) void lanes_mfma_f32_16x16x4_f32(float a, float b, float* d) {
  const unsigned lane = cur_thread() & 63;
  const float ab[2] = {a, b};
  const Xchg x = wave_xchg(3, __builtin_return_address(0), ab, 8, true);
  const unsigned j = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const unsigned i = 4 * (lane >> 4) + r;
    float sum = d[r];
    for (unsigned k = 0; k < 4; ++k) {
      float av, bv;
      std::memcpy(&av, x.tab + 64 * (16 * k + i), 4);
      std::memcpy(&bv, x.tab + 64 * (16 * k + j) + 4, 4);
      sum = fmaf(av, bv, sum);
    }
    d[r] = sum;
  }
}
// v_mfma_f32_32x32x16_bf16: lane l supplies A[l % 32][8 (l / 32) .. + 7] and B[8 (l / 32) .. + 7][l % 32]; register r holds
// D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32]
__attribute__((noinline)) void lanes_mfma_f32_32x32x16_bf16(const void* a8, const void* b8, float* d) {
  const unsigned lane = cur_thread() & 63;
  unsigned char ab[32]; std::memcpy(ab, a8, 16); std::memcpy(ab + 16, b8, 16);
  const Xchg x = wave_xchg(4, __builtin_return_address(0), ab, 32, true);
  const unsigned j = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const unsigned i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
    float sum = d[r];
    for (unsigned k = 0; k < 16; ++k) {
      unsigned short av, bv;
      std::memcpy(&av, x.tab + 64 * (32 * (k >> 3) + i) + 2 * (k & 7), 2);
      std::memcpy(&bv, x.tab + 64 * (32 * (k >> 3) + j) + 16 + 2 * (k & 7), 2);
      sum = fmaf(bf16_to_f(av), bf16_to_f(bv), sum);
    }
    d[r] = sum;
  }
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[l % 16][8 (l / 16) .. + 7] and B[8 (l / 16) .. + 7][l % 16]; register r holds
// D[4 (l / 16) + r][l % 16]
__attribute__((noinline)) void lanes_mfma_f32_16x16x32_bf16(const void* a8, const void* b8, float* d) {
  const unsigned lane = cur_thread() & 63;
  unsigned char ab[32]; std::memcpy(ab, a8, 16); std::memcpy(ab + 16, b8, 16);
  const Xchg x = wave_xchg(5, __builtin_return_address(0), ab, 32, true);
  const unsigned j = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const unsigned i = 4 * (lane >> 4) + r;
    float sum = d[r];
    for (unsigned k = 0; k < 32; ++k) {
      unsigned short av, bv;
      std::memcpy(&av, x.tab + 64 * (16 * (k >> 3) + i) + 2 * (k & 7), 2);
      std::memcpy(&bv, x.tab + 64 * (16 * (k >> 3) + j) + 16 + 2 * (k & 7), 2);
      sum = fmaf(bf16_to_f(av), bf16_to_f(bv), sum);
    }
    d[r] = sum;
  }
}
static inline unsigned lanes_from(int op, const void* site, unsigned v, int delta) {
  const int lane = (int)(cur_thread() & 63), srcl = lane + delta;
  const Xchg x = wave_xchg(op, site, &v, 4, false);
  if (srcl < 0 || srcl > 63 || !((x.mask >> srcl) & 1ull)) return 0u;   // bound_ctrl: no source lane (or one masked off) reads as 0
  unsigned r; std::memcpy(&r, x.tab + 64 * srcl, 4);
  return r;
}
__attribute__((noinline)) unsigned lanes_dpp_wave_shr1(unsigned v) { return lanes_from(6, __builtin_return_address(0), v, -1); }
__attribute__((noinline)) unsigned lanes_dpp_wave_shl1(unsigned v) { return lanes_from(7, __builtin_return_address(0), v, +1); }
__attribute__((noinline)) unsigned lanes_readfirstlane(unsigned v) {
  const Xchg x = wave_xchg(8, __builtin_return_address(0), &v, 4, false);
  unsigned r; std::memcpy(&r, x.tab + 64 * __builtin_ctzll(x.mask), 4);
  return r;
}
__attribute__((noinline)) unsigned long long lanes_shfl_xor64(unsigned long long v, int xr) {
  const unsigned lane = cur_thread() & 63, srcl = (lane ^ (unsigned)xr) & 63;
  const Xchg x = wave_xchg(10, __builtin_return_address(0), &v, 8, false);
  if (!((x.mask >> srcl) & 1ull)) lanes_fail("__shfl_xor from a lane that does not execute it");
  unsigned long long r; std::memcpy(&r, x.tab + 64 * srcl, 8);
  return r;
}
__attribute__((noinline)) void lanes_wave_sync() { (void)wave_xchg(9, __builtin_return_address(0), nullptr, 0, false); }
#endif

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
#ifdef CSN_EMU_LANES
  pool.nthreads = n; pool.n_bar = pool.n_done = 0; pool.bar_gen = 0;
  pool.waves.resize((n + 63) / 64);
  for (size_t w = 0; w < pool.waves.size(); ++w) {
    WaveSt& x = pool.waves[w];
    x.size = std::min(64u, n - 64u * (unsigned)w);
    x.completed = x.nx = x.nbar = x.ndone = 0;
    for (OpenOp* o : x.open) { o->complete = true; o->ndep = o->nread = 0; }
  }
#endif
  unsigned alive = n;
  while (alive) {
#ifdef CSN_EMU_LANES
    const unsigned long before = pool.progress;
#endif
    for (unsigned t = 0; t < n; ++t) {
      Fiber& f = pool.fibers[t];
      if (f.done) continue;
      g.tIdx = uint3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
      pool.running = &f;
      to_fiber(f);
      if (f.done) {
        --alive;
#ifdef CSN_EMU_LANES
        ++pool.waves[t >> 6].ndone; ++pool.n_done; ++pool.progress;
#endif
      }
    }
#ifdef CSN_EMU_LANES
    if (alive && pool.progress == before) lanes_fail("deadlock: every live fiber waits (barrier / cross-lane instruction never completed)");
#endif
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
  g_kernel_name = name;
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
#ifdef CSN_EMU_LANES
// tests: the number of cross-lane instructions executed so far (kind: 1 mfma 4x4x1, 2 4x4x4 bf16, 3 16x16x4, 4 32x32x16 bf16,
// 5 16x16x32 bf16, 6 / 7 DPP wave_shr / wave_shl, 8 readfirstlane, 9 wave sync, 10 shuffle-xor); proof that the lane-exact paths ran
extern "C" unsigned long long csn_emu_lane_ops(int kind) { return csn_emu::g_lane_ops[kind & 15]; }
#endif
#endif  // CSN_EMU_IMPL
