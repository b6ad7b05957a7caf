// csn_device.h -- shared device-side helpers for the csnet HIP kernels (gfx950 / CDNA4).
//
// All activations are planar [B][C][H][W] fp32.  Within a kernel the channel index is wave-uniform,
// so conv weights and the folded BN/PReLU parameters are read through the scalar path (s_load) and
// feed v_fmac as SGPR operands; only activations travel through VGPRs / LDS.
#pragma once

// CSN_CPU_EMU (tests/emu: the same sources compiled by g++, TEST INFRASTRUCTURE) comes in two modes.  Default: the lanes of a wave
// are independent sequential fibers and every cross-lane operand -- another lane's row of an MFMA A operand, a halo column by DPP --
// is replaced by a functional stand-in (CSN_EMU_SEQ: checks index arithmetic).  `make LANES=1` (CSN_EMU_LANES): the DEVICE paths
// are compiled and the cross-lane instructions execute lane-exactly through hip_cpu_shim.h's wave rendezvous (checks lane maps).
#if defined(CSN_CPU_EMU) && !defined(CSN_EMU_LANES)
#define CSN_EMU_SEQ 1
#endif

#ifdef CSN_CPU_EMU
#include "hip_cpu_shim.h"
#define CSN_LAUNCH(kern, grid, block, smem, stream, ...) \
  csn_emu::launch_named(#kern, (grid), (block), (smem), [&]() { kern(__VA_ARGS__); })
#define CSN_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(csn_emu::g.smem)
#else
#include <hip/hip_runtime.h>
#define CSN_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define CSN_DYN_SMEM(type, name)                                                  \
  extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_lds[];  \
  type* name = reinterpret_cast<type*>(name##_raw_lds)
#endif

#include <stdint.h>

// Constant address space (4) views.  Packed weights / folded BN tables are written by csn_prep_kernel in
// an EARLIER launch and never by the kernels that consume them, so they may be read through the scalar
// cache (s_load) and used as SGPR operands of v_fmac; the by-value argument block is read in place
// from the kernarg segment (dynamic indexing of a by-value struct would otherwise be copied to scratch).
#ifdef CSN_CPU_EMU
#define CSN_CONST_AS
#define CSN_KERNARG(T, a) (&(a))
#else
#define CSN_CONST_AS __attribute__((address_space(4)))
#define CSN_KERNARG(T, a) ((const CSN_CONST_AS T*)__builtin_amdgcn_kernarg_segment_ptr())
#endif
typedef const CSN_CONST_AS float* csn_cfp;
__device__ __forceinline__ csn_cfp csn_const(const float* p) {
#ifdef CSN_CPU_EMU
  return p;
#else
  return (csn_cfp)(p);
#endif
}

#define CSN_BLOCK 256

#ifndef CSN_CPU_EMU
#include <mutex>
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: a process-wide "done" flag leaves the kernels of a second GPU
// without it.  One bit per device ordinal, set under a mutex once the attributes of that device are in place.
struct CsnPerDeviceOnce {
  std::mutex mu;
  unsigned long long mask = 0;
  template <class F> int run(F&& set_attributes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> g(mu);
    if (mask & bit) return 0;
    const int st = set_attributes();
    if (st == 0) mask |= bit;
    return st;
  }
};
#endif

// Buffer-resource loads (CDNA "SRSRC" addressing): the wave-uniform base lives in 4 SGPRs, the per-lane
// part is ONE 32-bit VGPR byte offset and a per-channel uniform byte offset rides in an SGPR, so a lane
// needs a single address register for all channels of a gather.
#ifdef CSN_CPU_EMU
struct csn_buf { const char* p; unsigned n; };
static inline csn_buf csn_make_buf(const void* p) { return csn_buf{reinterpret_cast<const char*>(p), 0xffffffffu}; }
// bounded resource: loads whose byte offset is >= nbytes return 0 (the hardware's out-of-range rule)
static inline csn_buf csn_make_buf_n(const void* p, unsigned nbytes) { return csn_buf{reinterpret_cast<const char*>(p), nbytes}; }
static inline float csn_ld1(csn_buf b, unsigned voff, unsigned soff) {
  const unsigned o = voff + soff;
  return o + 4u <= b.n && o + 4u > o ? *reinterpret_cast<const float*>(b.p + o) : 0.f;
}
static inline float2 csn_ld2(csn_buf b, unsigned voff, unsigned soff) {
  const unsigned o = voff + soff;
  if (!(o + 8u <= b.n && o + 8u > o)) return make_float2(0.f, 0.f);
  return *reinterpret_cast<const float2*>(b.p + o);
}
// store through a bounded resource: dropped when voffset + soffset is out of range (the hardware's rule, see below)
static inline void csn_st1(csn_buf b, unsigned voff, unsigned soff, float v) {
  const unsigned o = voff + soff;
  if (o + 4u <= b.n && o + 4u > o && o >= voff) *reinterpret_cast<float*>(const_cast<char*>(b.p) + o) = v;
}
static inline void csn_st2(csn_buf b, unsigned voff, unsigned soff, float2 v) {
  const unsigned o = voff + soff;
  if (o + 8u <= b.n && o + 8u > o && o >= voff) {
    float* q = reinterpret_cast<float*>(const_cast<char*>(b.p) + o);
    q[0] = v.x; q[1] = v.y;
  }
}
static inline float4 csn_ld4(csn_buf b, unsigned voff, unsigned soff) {
  const unsigned o = voff + soff;
  if (!(o + 16u <= b.n && o + 16u > o)) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float* q = reinterpret_cast<const float*>(b.p + o);
  return make_float4(q[0], q[1], q[2], q[3]);
}
static inline void csn_st4(csn_buf b, unsigned voff, unsigned soff, float4 v) {
  const unsigned o = voff + soff;
  if (o + 16u <= b.n && o + 16u > o && o >= voff) {
    float* q = reinterpret_cast<float*>(const_cast<char*>(b.p) + o);
    q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
  }
}
static inline unsigned short csn_ld_u16(csn_buf b, unsigned voff, unsigned soff) {
  const unsigned o = voff + soff;
  return o + 2u <= b.n && o + 2u > o ? *reinterpret_cast<const unsigned short*>(b.p + o) : (unsigned short)0;
}
static inline unsigned csn_ld_u32(csn_buf b, unsigned voff, unsigned soff) {
  const unsigned o = voff + soff;
  return o + 4u <= b.n && o + 4u > o ? *reinterpret_cast<const unsigned*>(b.p + o) : 0u;
}
static inline uint2 csn_ld_u64(csn_buf b, unsigned voff, unsigned soff) {
  const unsigned o = voff + soff;
  if (!(o + 8u <= b.n && o + 8u > o)) return make_uint2(0u, 0u);
  const unsigned* q = reinterpret_cast<const unsigned*>(b.p + o);
  return make_uint2(q[0], q[1]);
}
static inline void csn_st_u16(csn_buf b, unsigned voff, unsigned soff, unsigned short v) {
  const unsigned o = voff + soff;
  if (o + 2u <= b.n && o + 2u > o && o >= voff) *reinterpret_cast<unsigned short*>(const_cast<char*>(b.p) + o) = v;
}
static inline void csn_st_u32(csn_buf b, unsigned voff, unsigned soff, unsigned v) {
  const unsigned o = voff + soff;
  if (o + 4u <= b.n && o + 4u > o && o >= voff) *reinterpret_cast<unsigned*>(const_cast<char*>(b.p) + o) = v;
}
static inline void csn_st_u64(csn_buf b, unsigned voff, unsigned soff, uint2 v) {
  const unsigned o = voff + soff;
  if (o + 8u <= b.n && o + 8u > o && o >= voff) {
    unsigned* q = reinterpret_cast<unsigned*>(const_cast<char*>(b.p) + o);
    q[0] = v.x; q[1] = v.y;
  }
}
typedef uint2 csn_u2;
typedef uint4 csn_u4;
#else
typedef __amdgpu_buffer_rsrc_t csn_buf;
typedef unsigned csn_u2 __attribute__((ext_vector_type(2)));
typedef unsigned csn_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ csn_buf csn_make_buf(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ csn_buf csn_make_buf_n(const void* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nbytes, 0x00020000);
}
__device__ __forceinline__ float csn_ld1(csn_buf b, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, voff, soff, 0));
}
// NOTE (measured on gfx950): the range check of a raw buffer access is applied to voffset + soffset, so a
// bounded resource must span everything the soffset can reach
__device__ __forceinline__ void csn_st1(csn_buf b, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b, voff, soff, 0);
}
__device__ __forceinline__ void csn_st2(csn_buf b, unsigned voff, unsigned soff, float2 v) {
  csn_u2 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b64(u, b, voff, soff, 0);
}
__device__ __forceinline__ float2 csn_ld2(csn_buf b, unsigned voff, unsigned soff) {
  const csn_u2 v = __builtin_amdgcn_raw_buffer_load_b64(b, voff, soff, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ float4 csn_ld4(csn_buf b, unsigned voff, unsigned soff) {
  const csn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(b, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void csn_st4(csn_buf b, unsigned voff, unsigned soff, float4 v) {
  csn_u4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, b, voff, soff, 0);
}
__device__ __forceinline__ unsigned short csn_ld_u16(csn_buf b, unsigned voff, unsigned soff) {
  return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(b, voff, soff, 0);
}
__device__ __forceinline__ unsigned csn_ld_u32(csn_buf b, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b32(b, voff, soff, 0);
}
__device__ __forceinline__ uint2 csn_ld_u64(csn_buf b, unsigned voff, unsigned soff) {
  const csn_u2 v = __builtin_amdgcn_raw_buffer_load_b64(b, voff, soff, 0);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void csn_st_u16(csn_buf b, unsigned voff, unsigned soff, unsigned short v) {
  __builtin_amdgcn_raw_buffer_store_b16(v, b, voff, soff, 0);
}
__device__ __forceinline__ void csn_st_u32(csn_buf b, unsigned voff, unsigned soff, unsigned v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, b, voff, soff, 0);
}
__device__ __forceinline__ void csn_st_u64(csn_buf b, unsigned voff, unsigned soff, uint2 v) {
  csn_u2 u;
  u.x = v.x; u.y = v.y;
  __builtin_amdgcn_raw_buffer_store_b64(u, b, voff, soff, 0);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// Activation element types.  Eval mode and the fp32 train mode keep activations as float; the bf16 train mode
// (CSN_OPT_TRAIN_BF16, BASELINE config 3) stores every activation / activation gradient in HBM as bfloat16 and does
// all arithmetic in fp32 registers: kernels are templates over the element type AT and touch memory only through
// the accessors below (round-to-nearest-even on store, exact widening on load).
// ---------------------------------------------------------------------------------------------------------------
struct csn_bf16 { unsigned short u; };

__device__ __forceinline__ float csn_bits_f(unsigned u) {
#ifdef CSN_CPU_EMU
  float f; __builtin_memcpy(&f, &u, 4); return f;
#else
  return __uint_as_float(u);
#endif
}
__device__ __forceinline__ unsigned short csn_f2bf(float f) {
#ifdef CSN_CPU_EMU
  unsigned u; __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                  // round to nearest even
  return (unsigned short)(u >> 16);
#else
  return __builtin_bit_cast(unsigned short, (__bf16)f);                              // v_cvt_pk_bf16_f32
#endif
}
__device__ __forceinline__ float csn_bf2f(unsigned short h) { return csn_bits_f((unsigned)h << 16); }
__device__ __forceinline__ unsigned csn_f_bits(float f) {
#ifdef CSN_CPU_EMU
  unsigned u; __builtin_memcpy(&u, &f, 4); return u;
#else
  return __float_as_uint(f);
#endif
}

// ---- cross-lane instructions (device: the builtins; CSN_EMU_LANES: the shim's lane-exact forms; CSN_EMU_SEQ: not available --
// the call sites carry their stand-ins) ----
#ifdef CSN_CPU_EMU
struct csn_f4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};
struct csn_f16v {
  float v[16];
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};
#else
typedef float csn_f4 __attribute__((ext_vector_type(4)));
typedef float csn_f16v __attribute__((ext_vector_type(16)));
#endif
#ifndef CSN_EMU_SEQ
// the value of the lane below / above (v_mov_b32_dpp wave_shr:1 / wave_shl:1, bound_ctrl:0 -> 0 where there is no such lane)
__device__ __forceinline__ unsigned csn_from_lane_below(unsigned v) {
#ifdef CSN_EMU_LANES
  return csn_emu::lanes_dpp_wave_shr1(v);
#else
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
#endif
}
__device__ __forceinline__ unsigned csn_from_lane_above(unsigned v) {
#ifdef CSN_EMU_LANES
  return csn_emu::lanes_dpp_wave_shl1(v);
#else
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
#endif
}
__device__ __forceinline__ int csn_readfirstlane(int v) {
#ifdef CSN_EMU_LANES
  return (int)csn_emu::lanes_readfirstlane((unsigned)v);
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}
// the value of lane (lane ^ x)
__device__ __forceinline__ double csn_shfl_xor(double v, int x) {
#ifdef CSN_EMU_LANES
  unsigned long long u; __builtin_memcpy(&u, &v, 8);
  u = csn_emu::lanes_shfl_xor64(u, x);
  __builtin_memcpy(&v, &u, 8);
  return v;
#else
  return __shfl_xor(v, x, 64);
#endif
}
// the float of lane (lane ^ X), X a power of two: inside a row of 16 lanes by DPP (quad_perm; row_shl:4 / row_shr:4 under bank masks;
// row_ror:8), across rows by ds_bpermute
template <int X>
__device__ __forceinline__ float csn_lane_xor_f32(float v) {
#ifdef CSN_EMU_LANES
  return (float)csn_shfl_xor((double)v, X);
#else
  const int i = __float_as_int(v);
  if (X == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, true));    // quad_perm:[1,0,3,2]
  if (X == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, true));    // quad_perm:[2,3,0,1]
  if (X == 4) {   // banks 0 / 2 of a row (lanes 0-3, 8-11) read lane + 4, banks 1 / 3 read lane - 4
    int r = __builtin_amdgcn_update_dpp(0, i, 0x104, 0xf, 0x5, false);   // row_shl:4
    r = __builtin_amdgcn_update_dpp(r, i, 0x114, 0xf, 0xa, false);       // row_shr:4
    return __int_as_float(r);
  }
  if (X == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x128, 0xf, 0xf, true));   // row_ror:8
  return __shfl_xor(v, X, 64);
#endif
}
// Sums over the wave of EIGHT per-lane values at once: in the stages X = 32, 16, 8 a lane keeps the values whose index parity equals
// its bit X and adds the partner's copies (8 -> 4 -> 2 -> 1 values: 7 exchanges + additions), then a butterfly over the lanes that
// share bits 5 .. 3 (3 more) -- 10 instead of 6 x 8.  Every lane ends with the wave's total of value 4 * bit3 + 2 * bit4 + bit5 of
// its lane number.  One fixed summation tree per value: deterministic.
template <int N, int X>
__device__ __forceinline__ void csn_rs_stage(float* v, bool bit) {
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const float keep = bit ? v[2 * j + 1] : v[2 * j];
    const float send = bit ? v[2 * j] : v[2 * j + 1];
    v[j] = keep + csn_lane_xor_f32<X>(send);
  }
}
__device__ __forceinline__ float csn_wave_reduce_scatter8(float (&v)[8], int lane) {
  csn_rs_stage<8, 32>(v, (lane & 32) != 0);
  csn_rs_stage<4, 16>(v, (lane & 16) != 0);
  csn_rs_stage<2, 8>(v, (lane & 8) != 0);
  float t = v[0];
  t += csn_lane_xor_f32<4>(t);
  t += csn_lane_xor_f32<2>(t);
  t += csn_lane_xor_f32<1>(t);
  return t;
}
__device__ __forceinline__ int csn_rs8_index(int lane) { return ((lane & 8) >> 1) | ((lane & 16) >> 3) | ((lane & 32) >> 5); }
// acc[i] += A[i] * b: A[i] = the `a` of lane (lane & ~3) + i, b the lane's own (v_mfma_f32_4x4x1_16b_f32)
__device__ __forceinline__ csn_f4 csn_mfma_4x4x1(float a, float b, csn_f4 acc) {
#ifdef CSN_EMU_LANES
  csn_emu::lanes_mfma_f32_4x4x1(a, b, acc.v);
  return acc;
#else
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
#endif
}
// ... with four bfloat16 k values per lane (a, b: four bfloat16 in two dwords; v_mfma_f32_4x4x4_16b_bf16)
__device__ __forceinline__ csn_f4 csn_mfma_4x4x4_bf16(uint2 a, uint2 b, csn_f4 acc) {
#ifdef CSN_EMU_LANES
  csn_emu::lanes_mfma_f32_4x4x4_bf16(&a, &b, acc.v);
  return acc;
#else
  typedef short s4 __attribute__((ext_vector_type(4)));
  union { uint2 u; s4 s; } ca, cb;
  ca.u = a; cb.u = b;
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ca.s, cb.s, acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ csn_f4 csn_mfma_16x16x4(float a, float b, csn_f4 acc) {
#ifdef CSN_EMU_LANES
  csn_emu::lanes_mfma_f32_16x16x4_f32(a, b, acc.v);
  return acc;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#endif
}
// the 8-pixel bfloat16 forms (a, b: eight bfloat16 in four dwords)
__device__ __forceinline__ csn_f16v csn_mfma_32x32x16_bf16(csn_u4 a, csn_u4 b, csn_f16v acc) {
#ifdef CSN_EMU_LANES
  csn_emu::lanes_mfma_f32_32x32x16_bf16(&a, &b, acc.v);
  return acc;
#else
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ csn_f4 csn_mfma_16x16x32_bf16(csn_u4 a, csn_u4 b, csn_f4 acc) {
#ifdef CSN_EMU_LANES
  csn_emu::lanes_mfma_f32_16x16x32_bf16(&a, &b, acc.v);
  return acc;
#else
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
#endif
}
#endif  // !CSN_EMU_SEQ
// v_alignbit_b32: the low dword of ({hi, lo} >> sh)
__device__ __forceinline__ unsigned csn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
#ifdef CSN_CPU_EMU
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31));
#else
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#endif
}
// loads above stay above, arithmetic below stays below
#ifdef CSN_CPU_EMU
#define CSN_SCHED_FENCE()
#else
#define CSN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// two consecutive elements packed in a dword (element 0 in the low half)
__device__ __forceinline__ unsigned csn_pack_bf2(float a, float b) { return (unsigned)csn_f2bf(a) | ((unsigned)csn_f2bf(b) << 16); }

// plain-pointer accessors (p is aligned to the access: 1, 2 or 4 elements)
__device__ __forceinline__ float act_ld(const float* p) { return *p; }
__device__ __forceinline__ float act_ld(const csn_bf16* p) { return csn_bf2f(p->u); }
__device__ __forceinline__ void act_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void act_st(csn_bf16* p, float v) { p->u = csn_f2bf(v); }
__device__ __forceinline__ float2 act_ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 act_ld2(const csn_bf16* p) {
  const unsigned u = *reinterpret_cast<const unsigned*>(p);
  return make_float2(csn_bits_f(u << 16), csn_bits_f(u & 0xffff0000u));
}
__device__ __forceinline__ void act_st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void act_st2(csn_bf16* p, float2 v) { *reinterpret_cast<unsigned*>(p) = csn_pack_bf2(v.x, v.y); }
__device__ __forceinline__ float4 act_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 act_ld4(const csn_bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(csn_bits_f(u.x << 16), csn_bits_f(u.x & 0xffff0000u), csn_bits_f(u.y << 16), csn_bits_f(u.y & 0xffff0000u));
}
__device__ __forceinline__ void act_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void act_st4(csn_bf16* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(csn_pack_bf2(v.x, v.y), csn_pack_bf2(v.z, v.w));
}
// eight consecutive elements (p 16-byte aligned for bfloat16, 32 for float): ONE 128-bit access per lane in the bf16 mode, two
// in flight for float -- the streaming BatchNorm kernels of the train step (round 4)
struct csn_f8 { float v[8]; };
__device__ __forceinline__ csn_f8 act_ld8(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  csn_f8 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ csn_f8 act_ld8(const csn_bf16* p) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  csn_f8 r;
  r.v[0] = csn_bits_f(u.x << 16); r.v[1] = csn_bits_f(u.x & 0xffff0000u); r.v[2] = csn_bits_f(u.y << 16); r.v[3] = csn_bits_f(u.y & 0xffff0000u);
  r.v[4] = csn_bits_f(u.z << 16); r.v[5] = csn_bits_f(u.z & 0xffff0000u); r.v[6] = csn_bits_f(u.w << 16); r.v[7] = csn_bits_f(u.w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void act_st8(float* p, const csn_f8& r) {
  *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ void act_st8(csn_bf16* p, const csn_f8& r) {
  *reinterpret_cast<uint4*>(p) = make_uint4(csn_pack_bf2(r.v[0], r.v[1]), csn_pack_bf2(r.v[2], r.v[3]), csn_pack_bf2(r.v[4], r.v[5]),
                                             csn_pack_bf2(r.v[6], r.v[7]));
}
// typed view of an argument-block pointer (the blocks carry `float*` whatever the element type)
template <typename AT> __device__ __forceinline__ const AT* act_cast(const float* p) { return reinterpret_cast<const AT*>(p); }
template <typename AT> __device__ __forceinline__ AT* act_cast(float* p) { return reinterpret_cast<AT*>(p); }

// buffer-resource accessors over the element type: offsets are BYTES (callers scale by sizeof(AT)); a 2- / 4-element
// access starts on an even / multiple-of-4 element
template <typename AT> struct csn_bufacc;
template <> struct csn_bufacc<float> {
  static __device__ __forceinline__ float ld1(csn_buf b, unsigned voff, unsigned soff) { return csn_ld1(b, voff, soff); }
  static __device__ __forceinline__ float2 ld2(csn_buf b, unsigned voff, unsigned soff) { return csn_ld2(b, voff, soff); }
  static __device__ __forceinline__ float4 ld4(csn_buf b, unsigned voff, unsigned soff) { return csn_ld4(b, voff, soff); }
  // raw forms: what a load leaves in the registers / its float value.  Kernels that keep a batch of loads in flight across a
  // scheduling fence hold the RAW registers and convert where the values are used (round 6: with ld1 / ld2 the bfloat16
  // conversion sat in front of the fence and every batch waited for its own loads -- pw4_kernel<bf16> at 0.21 of the HBM peak)
  typedef float r1; typedef float2 r2; typedef float4 r4;
  static __device__ __forceinline__ r1 ldr1(csn_buf b, unsigned voff, unsigned soff) { return csn_ld1(b, voff, soff); }
  static __device__ __forceinline__ r2 ldr2(csn_buf b, unsigned voff, unsigned soff) { return csn_ld2(b, voff, soff); }
  static __device__ __forceinline__ r4 ldr4(csn_buf b, unsigned voff, unsigned soff) { return csn_ld4(b, voff, soff); }
  static __device__ __forceinline__ float cv1(r1 v) { return v; }
  static __device__ __forceinline__ float2 cv2(r2 v) { return v; }
  static __device__ __forceinline__ float4 cv4(r4 v) { return v; }
  static __device__ __forceinline__ void st1(csn_buf b, unsigned voff, unsigned soff, float v) { csn_st1(b, voff, soff, v); }
  static __device__ __forceinline__ void st2(csn_buf b, unsigned voff, unsigned soff, float2 v) { csn_st2(b, voff, soff, v); }
  static __device__ __forceinline__ void st4(csn_buf b, unsigned voff, unsigned soff, float4 v) { csn_st4(b, voff, soff, v); }
};
template <> struct csn_bufacc<csn_bf16> {
  static __device__ __forceinline__ float ld1(csn_buf b, unsigned voff, unsigned soff) { return csn_bf2f(csn_ld_u16(b, voff, soff)); }
  static __device__ __forceinline__ float2 ld2(csn_buf b, unsigned voff, unsigned soff) {
    const unsigned u = csn_ld_u32(b, voff, soff);
    return make_float2(csn_bits_f(u << 16), csn_bits_f(u & 0xffff0000u));
  }
  static __device__ __forceinline__ float4 ld4(csn_buf b, unsigned voff, unsigned soff) {
    const uint2 u = csn_ld_u64(b, voff, soff);
    return make_float4(csn_bits_f(u.x << 16), csn_bits_f(u.x & 0xffff0000u), csn_bits_f(u.y << 16), csn_bits_f(u.y & 0xffff0000u));
  }
  typedef unsigned r1; typedef unsigned r2; typedef uint2 r4;
  static __device__ __forceinline__ r1 ldr1(csn_buf b, unsigned voff, unsigned soff) { return (unsigned)csn_ld_u16(b, voff, soff); }
  static __device__ __forceinline__ r2 ldr2(csn_buf b, unsigned voff, unsigned soff) { return csn_ld_u32(b, voff, soff); }
  static __device__ __forceinline__ r4 ldr4(csn_buf b, unsigned voff, unsigned soff) { return csn_ld_u64(b, voff, soff); }
  static __device__ __forceinline__ float cv1(r1 u) { return csn_bits_f(u << 16); }
  static __device__ __forceinline__ float2 cv2(r2 u) { return make_float2(csn_bits_f(u << 16), csn_bits_f(u & 0xffff0000u)); }
  static __device__ __forceinline__ float4 cv4(r4 u) {
    return make_float4(csn_bits_f(u.x << 16), csn_bits_f(u.x & 0xffff0000u), csn_bits_f(u.y << 16), csn_bits_f(u.y & 0xffff0000u));
  }
  static __device__ __forceinline__ void st1(csn_buf b, unsigned voff, unsigned soff, float v) { csn_st_u16(b, voff, soff, csn_f2bf(v)); }
  static __device__ __forceinline__ void st2(csn_buf b, unsigned voff, unsigned soff, float2 v) { csn_st_u32(b, voff, soff, csn_pack_bf2(v.x, v.y)); }
  static __device__ __forceinline__ void st4(csn_buf b, unsigned voff, unsigned soff, float4 v) {   // one 64-bit store
    csn_st_u64(b, voff, soff, make_uint2(csn_pack_bf2(v.x, v.y), csn_pack_bf2(v.z, v.w)));
  }
};

#ifndef CSN_FILL_U
#define CSN_FILL_U 8   // steps of loads in flight (1: the plain loop, A/B builds)
#endif
// Block-wide copy of a weight image into LDS, 16 bytes per thread and step, EIGHT steps of loads in flight before the first
// LDS write: a plain `dst[i] = src[i]` loop compiles to load / s_waitcnt vmcnt(0) / ds_write per 4 KB -- one L2 round trip
// each, 7-37 of them in front of every block's first item (round 3: 5-25 us of the small launches).
__device__ __forceinline__ void csn_fill_lds16(float* lds, const float* __restrict__ img, int n4, int tid) {
  const float4* __restrict__ src = reinterpret_cast<const float4*>(img);
  float4* dst = reinterpret_cast<float4*>(lds);
  constexpr int U = CSN_FILL_U;
  for (int i0 = tid; i0 < n4; i0 += U * CSN_BLOCK) {
    float4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = src[min(i0 + j * CSN_BLOCK, n4 - 1)];
#ifndef CSN_CPU_EMU
#pragma unroll
    for (int j = 0; j < U; ++j)   // a use right here: keeps the loads from being sunk into the guarded stores below
      asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j].z), "+v"(v[j].w));
#endif
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i0 + j * CSN_BLOCK < n4) dst[i0 + j * CSN_BLOCK] = v[j];
  }
}

// Folded epilogue of one output channel: y = z*scale + shift; y = y >= 0 ? y : alpha*y
// (nn.BatchNorm2d in eval mode followed by nn.PReLU; for cls_layer scale=1, shift=bias, alpha=1).
struct CsnEpi {
  const float* scale;
  const float* shift;
  const float* alpha;
};

__device__ __forceinline__ float csn_epi(float z, float sc, float sh, float al) {
  float y = fmaf(z, sc, sh);
  return y >= 0.f ? y : al * y;
}

// PyTorch's area_pixel_compute_source_index for align_corners=False (upsample_bilinear2d):
// src = (dst + 0.5) * (1/f) - 0.5, clamped at 0; i1 = min(i0 + 1, n - 1).
__device__ __forceinline__ void csn_bilin(int dst, float inv_f, int n, int& i0, int& i1, float& l1) {
  float src = (static_cast<float>(dst) + 0.5f) * inv_f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = static_cast<int>(src);
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = src - static_cast<float>(i0);
}
