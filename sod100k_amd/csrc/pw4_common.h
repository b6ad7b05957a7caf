// pw4_common.h -- helpers shared by the "lane = pixel quad, v_mfma_f32_4x4x1 from the load registers" kernels
// (k_pw4.hip: two-branch 1x1 units, k_c3q.hip: 3x3 passes).
#pragma once
#include "pw_gather.h"

namespace {

template <int NT4> struct Pw4A {
#ifdef CSN_EMU_SEQ
  csn_f4 r[4][NT4 / 4];   // all four rows of every tile (lanes are sequential fibers: no cross-lane operand)
#else
  csn_f4 r[1][NT4 / 4];   // this lane's row (lane & 3) of every tile
#endif
};

// A operands of one input channel: wk = image row of the channel ([4][P] floats), already offset by (lane & 3) * P on the device
template <int NT4, int P>
__device__ __forceinline__ void pw4_load_a(const float* wk, Pw4A<NT4>& a) {
#ifdef CSN_EMU_SEQ
  for (int i = 0; i < 4; ++i)
    for (int u = 0; u < NT4 / 4; ++u)
      for (int e = 0; e < 4; ++e) a.r[i][u][e] = wk[i * P + 4 * u + e];
#else
#pragma unroll
  for (int u = 0; u < NT4 / 4; ++u) a.r[0][u] = *reinterpret_cast<const csn_f4*>(wk + 4 * u);
#endif
}

// acc[i] += W[4 t + i][k] * x   for the lane's own pixel
template <int NT4>
__device__ __forceinline__ void pw4_mfma(const Pw4A<NT4>& a, int t, float x, csn_f4& acc) {
#ifdef CSN_EMU_SEQ
  for (int i = 0; i < 4; ++i) acc[i] = fmaf(a.r[i][t >> 2][t & 3], x, acc[i]);
#else
  acc = csn_mfma_4x4x1(a.r[0][t >> 2][t & 3], x, acc);
#endif
}

// (the register sets hold the loads RAW -- csn_bufacc<AT>::r1 / r2 -- and the batches convert where they use them: a conversion in
// front of PW4_FENCE would make every batch wait for its own loads, csn_device.h)
template <int HB, typename AT = float>
__device__ __forceinline__ void pw4_load_hi(csn_buf rb, unsigned o0, unsigned o1, unsigned cs, int k0, int C,
                                            typename csn_bufacc<AT>::r2 (&v)[HB][2]) {
#pragma unroll
  for (int j = 0; j < HB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, C - 1) * cs;
    v[j][0] = csn_bufacc<AT>::ldr2(rb, o0, so);
    v[j][1] = csn_bufacc<AT>::ldr2(rb, o1, so);
  }
}

template <int LB, typename AT = float>
__device__ __forceinline__ void pw4_load_lo(csn_buf rb, const unsigned (&o)[9], unsigned cs, int k0, int C,
                                            typename csn_bufacc<AT>::r1 (&v)[LB][9]) {
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, C - 1) * cs;
#pragma unroll
    for (int t = 0; t < 9; ++t) v[j][t] = csn_bufacc<AT>::ldr1(rb, o[t], so);
  }
}

// bilinear x2 (align_corners=False) of a 3x3 neighbourhood v[3 a + b] = t[row a - 1][column b - 1] (clamped addresses) to the
// 2x2 quad above its centre: q[2 dy + dx].  Weights 0.75 on the centre row / column, 0.25 on the row above (dy = 0) or
// below (dy = 1) and the column left (dx = 0) or right (dx = 1); at the borders the clamped neighbour IS the centre, which
// reproduces PyTorch's clamped source index (area_pixel_compute_source_index) to within one rounding.
__device__ __forceinline__ void pw4_up2_quad(const float (&v)[9], float (&q)[4]) {
  float h0[3], h1[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float c = 0.75f * v[3 * r + 1];
    h0[r] = fmaf(0.25f, v[3 * r], c);
    h1[r] = fmaf(0.25f, v[3 * r + 2], c);
  }
  const float m0 = 0.75f * h0[1], m1 = 0.75f * h1[1];
  q[0] = fmaf(0.25f, h0[0], m0); q[1] = fmaf(0.25f, h1[0], m1);
  q[2] = fmaf(0.25f, h0[2], m0); q[3] = fmaf(0.25f, h1[2], m1);
}

// contract `n` (<= HB, exact when !GUARD) high-branch channels: quad pixels -> high rows, their 2x2 maximum -> low rows
template <int NTH, int NTL, int HB, int P, bool GUARD, typename AT = float>
__device__ __forceinline__ void pw4_hi_batch(const typename csn_bufacc<AT>::r2 (&v)[HB][2], const float* wk, int n,
                                             csn_f4 (&acch)[4][NTH > 0 ? NTH : 1], csn_f4 (&accl)[NTL > 0 ? NTL : 1]) {
  constexpr int NT4 = (NTH + NTL + 3) & ~3;
#pragma unroll
  for (int j = 0; j < HB; ++j) {
    if (GUARD && j >= n) break;
    Pw4A<NT4> a;
    pw4_load_a<NT4, P>(wk + j * 4 * P, a);
    const float2 r0 = csn_bufacc<AT>::cv2(v[j][0]), r1 = csn_bufacc<AT>::cv2(v[j][1]);
    const float q[4] = {r0.x, r0.y, r1.x, r1.y};
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) pw4_mfma<NT4>(a, t, q[s], acch[s][t]);
    if (NTL > 0) {
      const float m = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
#pragma unroll
      for (int t = 0; t < NTL; ++t) pw4_mfma<NT4>(a, NTH + t, m, accl[t]);
    }
  }
}

// contract `n` low-branch channels: bilinear x2 of the 3x3 neighbourhood -> high rows, the centre -> low rows.
// v[j][3 a + b] = x_l[row a - 1][column b - 1] (clamped).  Quad pixel (dy, dx) interpolates with 0.75 on the centre row /
// column and 0.25 on the row above (dy = 0) or below (dy = 1), the column left (dx = 0) or right (dx = 1):
// upsample_bilinear2d, align_corners=False, scale 2 -- at the borders the clamped neighbour IS the centre, which gives
// PyTorch's clamped source index (area_pixel_compute_source_index) to within one rounding.
template <int NTH, int NTL, int LB, int P, bool GUARD, typename AT = float>
__device__ __forceinline__ void pw4_lo_batch(const typename csn_bufacc<AT>::r1 (&v)[LB][9], const float* wk, int n,
                                             csn_f4 (&acch)[4][NTH > 0 ? NTH : 1], csn_f4 (&accl)[NTL > 0 ? NTL : 1]) {
  constexpr int NT4 = (NTH + NTL + 3) & ~3;
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    if (GUARD && j >= n) break;
    Pw4A<NT4> a;
    pw4_load_a<NT4, P>(wk + j * 4 * P, a);
    float f[9], q[4];
#pragma unroll
    for (int t = 0; t < 9; ++t) f[t] = csn_bufacc<AT>::cv1(v[j][t]);
    if (NTH > 0) pw4_up2_quad(f, q);
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) pw4_mfma<NT4>(a, t, q[s], acch[s][t]);
#pragma unroll
    for (int t = 0; t < NTL; ++t) pw4_mfma<NT4>(a, NTH + t, f[4], accl[t]);
  }
}

// ---- v_mfma_f32_4x4x4_16B_bf16 from packed bfloat16 operands (bf16 train mode: k_pwq.hip pwq16_kernel, k_c3q.hip c3q16_kernel) ----

__device__ __forceinline__ unsigned pw16_perm(unsigned hi, unsigned lo, unsigned sel) {
#ifdef CSN_CPU_EMU
  const unsigned long long v = ((unsigned long long)hi << 32) | lo;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xffu) << (8 * i);
  return r;
#else
  return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}

// acc[i] += sum_k W[4 t + i][k0 + k] * x[k] for the lane's own element; a = the lane's row of the tile (device) / all four rows (emu)
struct Pw16A {
#ifdef CSN_EMU_SEQ
  uint2 r[4];
#else
  uint2 r[1];
#endif
};
__device__ __forceinline__ void pw16_mfma(const Pw16A& a, uint2 x, csn_f4& acc) {
#ifdef CSN_EMU_SEQ
  const unsigned xs[4] = {x.x & 0xffffu, x.x >> 16, x.y & 0xffffu, x.y >> 16};
  for (int i = 0; i < 4; ++i) {
    const unsigned ws[4] = {a.r[i].x & 0xffffu, a.r[i].x >> 16, a.r[i].y & 0xffffu, a.r[i].y >> 16};
    float sum = acc[i];
    for (int k = 0; k < 4; ++k) sum = fmaf(csn_bf2f((unsigned short)ws[k]), csn_bf2f((unsigned short)xs[k]), sum);
    acc[i] = sum;
  }
#else
  acc = csn_mfma_4x4x4_bf16(a.r[0], x, acc);
#endif
}


// y = z * scale + shift;  PReLU as max(y, 0) + alpha * min(y, 0): bit-identical to the select form of csn_epi (one of the
// two terms is an exact zero) without the v_cmp -> v_cndmask SGPR round trip
__device__ __forceinline__ float pw4_epi(float z, float sc, float sh, float al) {
  const float y = fmaf(z, sc, sh);
  return fmaf(al, fminf(y, 0.f), fmaxf(y, 0.f));
}

// keep a batch of loads together in front of the contraction it overlaps with (the scheduler would otherwise sink the
// loads into the MFMA stream and shorten the distance between issue and first use)
#ifdef CSN_CPU_EMU
#define PW4_FENCE()
#else
#define PW4_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

__device__ __forceinline__ int pw4_uniform(int v) {
#ifdef CSN_EMU_SEQ
  return v;
#else
  return csn_readfirstlane(v);
#endif
}

}  // namespace
