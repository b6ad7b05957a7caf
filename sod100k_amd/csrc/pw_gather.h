// pw_gather.h -- per-lane gather of a channel slice into a wave's LDS panel x[k][64 px] (shared by the
// forward contraction k_goct_pw.hip and the weight-gradient kernel k_wgrad.hip).  XP = panel row pitch.
#pragma once
#include "csn_kernels.h"

#if defined(CSN_EMU_SEQ)
// lanes of a wave run as sequential fibers: make LDS hand-offs inside a wave visible
#define CSN_WAVE_SYNC() __syncthreads()
#elif defined(CSN_EMU_LANES)
#define CSN_WAVE_SYNC() csn_emu::lanes_wave_sync()
#else
// a wave executes in lockstep and its LDS operations retire in order: only stop the compiler from
// moving LDS accesses across the hand-off
#define CSN_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// paired gathers (pw_batch_own_pair): bf16 always; float by default too (measured: train step 95.6 -> 91.7 ms), -DCSN_PAIR_F32=0 for A/B
#ifndef CSN_PAIR_F32
#define CSN_PAIR_F32 1
#endif
#define CSN_PAIR_GATHER(AT) (sizeof(AT) == 2 || CSN_PAIR_F32)

#define PW_EP 68   // pitch of the epilogue transpose: rows 4 apart land on the other half of the banks

typedef const CSN_CONST_AS PwPass* PwPassP;

// LP: type of the panel pointer (float*, or an LDS address-space pointer when the caller is not inlined into the kernel).
// Gather channels [c_lo, c_hi) of one slice for this lane's pixel into the panel rows starting at xrow
// (`rmax` rows are left in the panel).  Loads go through a buffer resource whose base is channel c_lo of
// image b (wave-uniform, SGPRs); the lane contributes one 32-bit byte offset, the channel a uniform SGPR
// offset.  Every batch issues ALL its loads before the first use (fixed trip count, channel index clamped
// instead of predicated: a predicated load would be waited for at the join), so a lane has 16-32 loads
// in flight; rows written past the slice are overwritten by the next slice / the zero padding.
template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_own(csn_buf rb, unsigned lo, unsigned cs4, int k0, int n, int rmax,
                                             LP xrow) {
  float v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = csn_bufacc<AT>::ld1(rb, lo, (unsigned)min(k0 + j, n - 1) * cs4);
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * XP] = v[j];
}

// Groups of 64 CONSECUTIVE pixels (weight-gradient kernels): lanes 0-31 fetch the group's 32 pixel pairs of one channel (one
// dword of two bf16 / one dwordx2 of two floats per lane) and lanes 32-63 those of the next channel -- one load instruction
// per TWO channels.  16-bit loads retire at the 32-bit instruction rate (tools/probes/ld16_probe.hip), and these kernels are
// bound by the number of instructions, so this halves the gather's cost in the bf16 mode.  `pair_off` = byte offset of
// this lane's pixel pair inside the plane (+ one plane for the upper half, 0 there when the slice has a single channel),
// xpair = panel base + 2 * (lane & 31), half = lane >> 5.  Rows past the slice repeat its last channel pair (same data, same
// row: harmless); rows >= rmax are not written.
template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_own_pair(csn_buf rb, unsigned pair_off, int half, unsigned cs2, int k0, int n, int rmax,
                                                  LP xpair) {
  float2 v[NB / 2];
  int row[NB / 2];
  const int last = n >= 2 ? n - 2 : 0;
#pragma unroll
  for (int j = 0; j < NB / 2; ++j) {
    const int cb = min(k0 + 2 * j, last);
    row[j] = cb + (n >= 2 ? half : 0);
    v[j] = csn_bufacc<AT>::ld2(rb, pair_off, (unsigned)cb * cs2);
  }
#pragma unroll
  for (int j = 0; j < NB / 2; ++j)
    if (row[j] < rmax) {
      xpair[row[j] * XP] = v[j].x;
      xpair[row[j] * XP + 1] = v[j].y;
    }
}

template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_pool2(csn_buf rb, unsigned lo, unsigned cs4, unsigned ws4, int k0, int n,
                                               int rmax, LP xrow) {
  float2 a0[NB], a1[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, n - 1) * cs4;
    a0[j] = csn_bufacc<AT>::ld2(rb, lo, so);
    a1[j] = csn_bufacc<AT>::ld2(rb, lo, so + ws4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * XP] = fmaxf(fmaxf(a0[j].x, a0[j].y), fmaxf(a1[j].x, a1[j].y));
}

template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_pool4(csn_buf rb, unsigned lo, unsigned cs4, unsigned ws4, int k0, int n,
                                               int rmax, LP xrow) {
  float4 q[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, n - 1) * cs4;
#pragma unroll
    for (int r = 0; r < 4; ++r) q[j][r] = csn_bufacc<AT>::ld4(rb, lo, so + r * ws4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    float m = -3.402823466e+38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(fmaxf(q[j][r].x, q[j][r].y), fmaxf(q[j][r].z, q[j][r].w)));
    if (k0 + j < rmax) xrow[(k0 + j) * XP] = m;
  }
}

template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_up(csn_buf rb, unsigned o00, unsigned o01, unsigned o10, unsigned o11,
                                            float w00, float w01, float w10, float w11, unsigned cs4, int k0, int n,
                                            int rmax, LP xrow) {
  float t0[NB], t1[NB], t2[NB], t3[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, n - 1) * cs4;
    t0[j] = csn_bufacc<AT>::ld1(rb, o00, so);
    t1[j] = csn_bufacc<AT>::ld1(rb, o01, so);
    t2[j] = csn_bufacc<AT>::ld1(rb, o10, so);
    t3[j] = csn_bufacc<AT>::ld1(rb, o11, so);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * XP] = w00 * t0[j] + w01 * t1[j] + w10 * t2[j] + w11 * t3[j];
}

// 3x3 taps of an own-resolution slice: gathered entry kk = 9*ch + t, t = 3*(dy+1) + (dx+1).  `vm` has
// bit t set when tap t of this lane's pixel lies inside the image (zero padding otherwise).
template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_taps(csn_buf rb, unsigned lo, unsigned cs4, int Wr, int dil, unsigned vm,
                                              int k_lo, int k0, int n, int rmax, LP xrow) {
  float v[NB];
  unsigned m[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int kk = k_lo + min(k0 + j, n - 1);
    const int ch = kk / 9, t = kk - 9 * ch;
    const int dy = t / 3 - 1, dx = t - 3 * (t / 3) - 1;
    v[j] = csn_bufacc<AT>::ld1(rb, lo + (unsigned)((dy * Wr + dx) * dil * (int)sizeof(AT)), (unsigned)ch * cs4);
    m[j] = (vm >> t) & 1u;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * XP] = m[j] ? v[j] : 0.f;
}

// 3x3 taps of a 2x2-max-pooled slice (source at twice the resolution).
template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_pool2_taps(csn_buf rb, unsigned lo, unsigned cs4, unsigned ws4, unsigned vm,
                                                    int k_lo, int k0, int n, int rmax, LP xrow) {
  float2 a0[NB], a1[NB];
  unsigned m[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int kk = k_lo + min(k0 + j, n - 1);
    const int ch = kk / 9, t = kk - 9 * ch;
    const int dy = t / 3 - 1, dx = t - 3 * (t / 3) - 1;
    const unsigned vo = lo + (unsigned)(2 * dy) * ws4 + (unsigned)(2 * (int)sizeof(AT) * dx);
    a0[j] = csn_bufacc<AT>::ld2(rb, vo, (unsigned)ch * cs4);
    a1[j] = csn_bufacc<AT>::ld2(rb, vo + ws4, (unsigned)ch * cs4);
    m[j] = (vm >> t) & 1u;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax)
      xrow[(k0 + j) * XP] = m[j] ? fmaxf(fmaxf(a0[j].x, a0[j].y), fmaxf(a1[j].x, a1[j].y)) : 0.f;
}

// 3x3 taps of a zero-stuffed slice (source at half the resolution): the adjoint of a stride-2 3x3 convolution.
// Tap t of output pixel (y, x) reads source ((y + dy) / 2, (x + dx) / 2) when both coordinates are even and inside.
template <typename AT, int NB, int XP, typename LP>
__device__ __forceinline__ void pw_batch_taps_ups2(csn_buf rb, int y, int x, int Hr, int Wr, int Ws, unsigned cs4,
                                                   int k_lo, int k0, int n, int rmax, LP xrow) {
  constexpr unsigned E = (unsigned)sizeof(AT);
  float v[NB];
  bool m[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int kk = k_lo + min(k0 + j, n - 1);
    const int ch = kk / 9, t = kk - 9 * ch;
    const int h = y + t / 3 - 1, w = x + (t - 3 * (t / 3)) - 1;
    m[j] = h >= 0 && w >= 0 && h < Hr && w < Wr && ((h | w) & 1) == 0;
    v[j] = csn_bufacc<AT>::ld1(rb, m[j] ? (unsigned)((h >> 1) * Ws + (w >> 1)) * E : 0u, (unsigned)ch * cs4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * XP] = m[j] ? v[j] : 0.f;
}

__device__ __forceinline__ unsigned pw_tap_mask(int y, int x, int Hr, int Wr, int dil) {
  unsigned vm = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + (t / 3 - 1) * dil, xx = x + (t % 3 - 1) * dil;
    vm |= (yy >= 0 && yy < Hr && xx >= 0 && xx < Wr) ? (1u << t) : 0u;
  }
  return vm;
}

// MODES 1: the caller guarantees a 1x1 slice (own / max-pool / bilinear) and the 3x3 tap modes are compiled out;
// MODES 2: the reverse (tap slices only); 0: everything
// pair_elem: element index inside the plane of the pixel PAIR this lane fetches in the paired own-resolution gather -- pixels
// 2q, 2q + 1 of the wave's group, q = lane & 31, which must be neighbours in memory (clamped into the plane by the caller);
// CSN_NO_PAIR: not available (odd plane / row width).
#define CSN_NO_PAIR 0xffffffffu
template <typename AT, int XP, int MODES = 0, typename LP = float*>
__device__ __forceinline__ void pw_gather_slice(PwPassP ps, int s, int c_lo, int c_hi, LP xrow, int rmax, int b,
                                                int y, int x, int Hr, int Wr, unsigned pair_elem = CSN_NO_PAIR) {
  const int mode = ps->src[s].mode;
  const int n = c_hi - c_lo;   // 1..16
  constexpr unsigned E = (unsigned)sizeof(AT);   // bytes per element
  if (MODES != 2 && mode == PW_OWN) {
    const unsigned cs = (unsigned)(Hr * Wr);
    const csn_buf rb = csn_make_buf(act_cast<AT>(ps->src[s].ptr) + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    if (CSN_PAIR_GATHER(AT) && pair_elem != CSN_NO_PAIR) {
      // a lane writes OTHER lanes' panel columns here: everything the previous slice wrote to those rows (its clamped
      // duplicates past the slice end) must be in the panel first.  A wave runs in lockstep, so on the device this is only a
      // compiler fence; the CPU emulation (lanes = sequential fibers) needs the hand-off.
      CSN_WAVE_SYNC();
      const int lane = (int)(threadIdx.x & 63), q = lane & 31, half = lane >> 5;
      const unsigned pair_off = pair_elem * E + ((n >= 2 && half) ? cs * E : 0u);
      LP xpair = xrow - lane + 2 * q;
      if (n <= 8) pw_batch_own_pair<AT, 8, XP>(rb, pair_off, half, cs * E, 0, n, rmax, xpair);
      else pw_batch_own_pair<AT, 16, XP>(rb, pair_off, half, cs * E, 0, n, rmax, xpair);
      return;
    }
    const unsigned lo = (unsigned)(y * Wr + x) * E;
    if (n <= 8) pw_batch_own<AT, 8, XP>(rb, lo, cs * E, 0, n, rmax, xrow);
    else for (int k0 = 0; k0 < n; k0 += 16) pw_batch_own<AT, 16, XP>(rb, lo, cs * E, k0, n, rmax, xrow);
  } else if (MODES != 2 && mode == PW_POOL2) {
    const unsigned Ws = (unsigned)Wr * 2u;
    const unsigned cs = (unsigned)(Hr * 2) * Ws;
    const csn_buf rb = csn_make_buf(act_cast<AT>(ps->src[s].ptr) + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned lo = ((unsigned)(2 * y) * Ws + 2u * x) * E;
    for (int k0 = 0; k0 < n; k0 += 8) pw_batch_pool2<AT, 8, XP>(rb, lo, cs * E, Ws * E, k0, n, rmax, xrow);
  } else if (MODES != 2 && mode == PW_POOL4) {
    const unsigned Ws = (unsigned)Wr * 4u;
    const unsigned cs = (unsigned)(Hr * 4) * Ws;
    const csn_buf rb = csn_make_buf(act_cast<AT>(ps->src[s].ptr) + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned lo = ((unsigned)(4 * y) * Ws + 4u * x) * E;
    for (int k0 = 0; k0 < n; k0 += 2) pw_batch_pool4<AT, 2, XP>(rb, lo, cs * E, Ws * E, k0, n, rmax, xrow);
  } else if (MODES != 1 && mode == PW_TAPS) {
    const unsigned cs = (unsigned)(Hr * Wr);
    const int dil = ps->src[s].dil;
    const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[s].ptr) + (int64_t)b * ps->src[s].Ctot * cs,
                                      (unsigned)ps->src[s].Ctot * cs * E);
    const unsigned lo = (unsigned)(y * Wr + x) * E;
    const unsigned vm = pw_tap_mask(y, x, Hr, Wr, dil);
    for (int k0 = 0; k0 < n; k0 += 16) pw_batch_taps<AT, 16, XP>(rb, lo, cs * E, Wr, dil, vm, c_lo, k0, n, rmax, xrow);
  } else if (MODES != 1 && mode == PW_POOL2_TAPS) {
    const unsigned Ws = (unsigned)Wr * 2u;
    const unsigned cs = (unsigned)(Hr * 2) * Ws;
    const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[s].ptr) + (int64_t)b * ps->src[s].Ctot * cs,
                                      (unsigned)ps->src[s].Ctot * cs * E);
    const unsigned lo = ((unsigned)(2 * y) * Ws + 2u * x) * E;
    const unsigned vm = pw_tap_mask(y, x, Hr, Wr, 1);
    for (int k0 = 0; k0 < n; k0 += 8) pw_batch_pool2_taps<AT, 8, XP>(rb, lo, cs * E, Ws * E, vm, c_lo, k0, n, rmax, xrow);
  } else if (MODES != 1 && mode == PW_TAPS_S2) {   // Conv2dX100 with stride 2 (csnet.py:751-754): taps (2y + dy, 2x + dx), zero padding
    const unsigned Ws = (unsigned)Wr * 2u;
    const unsigned cs = (unsigned)(Hr * 2) * Ws;
    const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[s].ptr) + (int64_t)b * ps->src[s].Ctot * cs,
                                      (unsigned)ps->src[s].Ctot * cs * E);
    const unsigned lo = ((unsigned)(2 * y) * Ws + 2u * x) * E;
    unsigned vm = 0x1ffu;            // only the top row / left column can fall outside
    if (y == 0) vm &= ~0x007u;
    if (x == 0) vm &= ~0x049u;
    for (int k0 = 0; k0 < n; k0 += 16) pw_batch_taps<AT, 16, XP>(rb, lo, cs * E, (int)Ws, 1, vm, c_lo, k0, n, rmax, xrow);
  } else if (MODES != 1 && mode == PW_TAPS_UPS2) {
    const int Hs = Hr >> 1, Ws = Wr >> 1;
    const unsigned cs = (unsigned)(Hs * Ws);
    const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[s].ptr) + (int64_t)b * ps->src[s].Ctot * cs,
                                      (unsigned)ps->src[s].Ctot * cs * E);
    for (int k0 = 0; k0 < n; k0 += 16) pw_batch_taps_ups2<AT, 16, XP>(rb, y, x, Hr, Wr, Ws, cs * E, c_lo, k0, n, rmax, xrow);
  } else if (MODES != 2) {  // bilinear from a 2x / 4x coarser branch, align_corners=False
    const int sh = mode == PW_UP2 ? 1 : 2;
    const int Hs = Hr >> sh, Ws = Wr >> sh;
    int y0, y1, x0, x1;
    float ly, lx;
    csn_bilin(y, mode == PW_UP2 ? 0.5f : 0.25f, Hs, y0, y1, ly);
    csn_bilin(x, mode == PW_UP2 ? 0.5f : 0.25f, Ws, x0, x1, lx);
    const unsigned cs = (unsigned)(Hs * Ws);
    const csn_buf rb = csn_make_buf(act_cast<AT>(ps->src[s].ptr) + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned o00 = (unsigned)(y0 * Ws + x0) * E, o01 = (unsigned)(y0 * Ws + x1) * E,
                   o10 = (unsigned)(y1 * Ws + x0) * E, o11 = (unsigned)(y1 * Ws + x1) * E;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    for (int k0 = 0; k0 < n; k0 += 8)
      pw_batch_up<AT, 8, XP>(rb, o00, o01, o10, o11, w00, w01, w10, w11, cs * E, k0, n, rmax, xrow);
  }
}

