// k_ms.hip -- MSBlock: five dilated 3x3 convolutions (weight x100) -> channel concat -> BN -> PReLU.
//
// Reference: MSBlock.forward CSNet/model/csnet.py:141-149 (Conv2dX100 with dilation d in {1,2,4,8,16},
// padding d, conv2d.py:104); dilations with 0 output channels are absent (csnet.py:128-137).
//
// Every dilation produces only 1..7 output channels from 17..38 input channels: a contraction with a tiny
// M, so the matrix pipe (16-row tiles) would idle; the cost is the 45 taps per input channel.  One lane
// owns one output pixel; per dilation it keeps <= 8 accumulators and walks the input channels two at a
// time: the 18 tap loads of a pair are buffer loads on a resource bounded to the image (no branches, zero
// padding by a per-lane 9-bit mask) and are all in flight before the first FMA; weights are wave-uniform
// (s_load).  The taps of neighbouring lanes are contiguous, and the three inputs (17x112^2, 38x56^2,
// 32x28^2 per image) are re-read from L1/L2 only.
#include <cstdlib>

#include "csn_kernels.h"

__device__ __forceinline__ unsigned ms_tap_mask(int y, int x, int H, int W, int dil) {
  unsigned vm = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + (t / 3 - 1) * dil, xx = x + (t % 3 - 1) * dil;
    vm |= (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (1u << t) : 0u;
  }
  return vm;
}

template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void msblock_kernel(MsArgs a) {
  constexpr unsigned E = (unsigned)sizeof(AT);
  const int H = a.H, W = a.W;
  const int hw = H * W;
  // 1-D grid, XCD-aware: workgroups go round-robin to the 8 XCDs (blockIdx.x & 7).  The five dilation blocks of a
  // pixel tile are dealt to the SAME XCD, back to back, so the tile's input (read by all five) stays in that L2
  // (with a (tiles, B, 5) grid they ran far apart on different XCDs: 382 MB of HBM reads per launch for 47 MB).
  const int ntx = (hw + CSN_BLOCK - 1) / CSN_BLOCK;
  const int slot = blockIdx.x >> 3;
  const int tile = (slot / 5) * 8 + (blockIdx.x & 7);
  if (tile >= ntx * a.B) return;
  const int b = tile / ntx;
  const int p0 = (tile - b * ntx) * CSN_BLOCK + threadIdx.x;
  const bool valid = p0 < hw;
  const int p = valid ? p0 : hw - 1;
  const int y = p / W, x = p - y * W;
  const csn_buf rb = csn_make_buf_n(act_cast<AT>(a.in) + (int64_t)b * a.cin * hw, (unsigned)(a.cin * hw) * E);
  const unsigned lo = (unsigned)p * E;
  const unsigned cs4 = (unsigned)hw * E;
  AT* __restrict__ op = act_cast<AT>(a.out) + (int64_t)b * a.cout * hw + p;
  csn_cfp scale = csn_const(a.scale), shift = csn_const(a.shift), alpha = csn_const(a.alpha);
  const int cinp = (a.cin + 1) & ~1;
  {
    const int d = slot % 5;   // one dilation per block: 5x more blocks for the small low-resolution maps
    const int nco = a.dch[d];
    if (nco == 0) return;
    const int dil = 1 << d;
    const unsigned vm = ms_tap_mask(y, x, H, W, dil);
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * W + (t % 3 - 1)) * dil * (int)E;
    const int ngrp = (nco + 7) >> 3;
    for (int g = 0; g < ngrp; ++g) {
      float acc[8];
#pragma unroll
      for (int co = 0; co < 8; ++co) acc[co] = 0.f;
      csn_cfp wg = csn_const(a.w[d]) + (int64_t)g * cinp * 72;
      for (int ci = 0; ci < cinp; ci += 2) {
        float v[2][9];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned so = (unsigned)min(ci + u, a.cin - 1) * cs4;   // pad channel: its weights are zero
#pragma unroll
          for (int t = 0; t < 9; ++t) v[u][t] = csn_bufacc<AT>::ld1(rb, lo + (unsigned)toff[t], so);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          csn_cfp wc = wg + (ci + u) * 72;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float val = ((vm >> t) & 1u) ? v[u][t] : 0.f;
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = fmaf(wc[t * 8 + co], val, acc[co]);
          }
        }
      }
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        const int lc = g * 8 + co;
        if (lc < nco && valid) {
          const int oc = a.cobase[d] + lc;
          act_st(op + (int64_t)oc * hw, csn_epi(acc[co], scale[oc], shift[oc], alpha[oc]));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ LDS-tiled version
// The per-pixel kernel above re-reads the input five times (once per dilation block: 4.4x the algorithmic bytes in the
// round-1 counter profile) and issues 45 tap loads per input channel and pixel.  Here a block owns a band of rows over the
// full width and stages the band + halo of a few input channels in LDS (zero rows / columns outside the image = the
// convolutions' zero padding, no masks); a thread owns FOUR consecutive pixels of a row, reads its taps as aligned 128-bit
// LDS accesses (dilations 1 and 2 pick theirs out of the three quads around the pixel, 4 / 8 / 16 are aligned by
// construction) and keeps 8 accumulators per dilation and pixel; weights are wave-uniform (s_load), each serving four
// pixels.  Two passes so that the accumulators fit: dilations {1, 2, 4} (halo 4) and {8, 16} (halo 16).
template <int PASS>
__global__ __launch_bounds__(CSN_BLOCK, PASS == 0 ? 2 : 3) void msblock2_kernel(MsArgs a, int RB, int CC) {
  CSN_DYN_SMEM(float, lds);
  constexpr int ND = PASS == 0 ? 3 : 2;          // dilations of this pass
  constexpr int D0 = PASS == 0 ? 0 : 3;          // index of the first one (dilation 2^index)
  constexpr int HALO = PASS == 0 ? 4 : 16;
  const int H = a.H, W = a.W, hw = H * W;
  const int QW = W >> 2;
  const int WP = W + 2 * HALO, RT = RB + 2 * HALO;
  const int bands = (H + RB - 1) / RB;
  const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
  const int y0 = band * RB;
  const int tid = threadIdx.x;
  const int qy = tid / QW, qx = tid - qy * QW;
  const bool active = qy < RB && y0 + qy < H;
  // staging roles: thread (tr, tq) copies quad tq of tile rows tr, tr + rps, ...
  const int QWP = WP >> 2;
  const int tq = tid % QWP, tr = tid / QWP, rps = CSN_BLOCK / QWP;
  const int scol = 4 * tq - HALO;                 // source column of the quad (outside [0, W): zero padding)
  const float* __restrict__ src = a.in + (int64_t)b * a.cin * hw;
  float acc[ND][8][4];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int co = 0; co < 8; ++co)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[d][co][p] = 0.f;
  const int plane = RT * WP;
  for (int c0 = 0; c0 < a.cin; c0 += CC) {
    const int nc = min(CC, a.cin - c0);
    __syncthreads();
    if (tr < rps) {
      for (int cc = 0; cc < nc; ++cc) {
        const float* __restrict__ sp = src + (int64_t)(c0 + cc) * hw;
        for (int r = tr; r < RT; r += rps) {
          const int yy = y0 - HALO + r;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (yy >= 0 && yy < H && scol >= 0 && scol < W) v = *reinterpret_cast<const float4*>(sp + (int64_t)yy * W + scol);
          *reinterpret_cast<float4*>(lds + cc * plane + r * WP + 4 * tq) = v;
        }
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int cc = 0; cc < nc; ++cc) {
      const float* tp = lds + cc * plane + (qy + HALO) * WP + HALO + 4 * qx;   // this thread's quad, dilation-free position
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const int dil = 1 << (D0 + d);
        const int nco = a.dch[D0 + d];
        if (nco == 0) continue;
        csn_cfp wc = csn_const(a.w[D0 + d]) + (int64_t)(c0 + cc) * 72;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float* rp = tp + (dy - 1) * dil * WP;
          float tl[4], tc[4], trr[4];          // the four pixels' taps dx = -1, 0, +1
          const float4 C = *reinterpret_cast<const float4*>(rp);
          tc[0] = C.x; tc[1] = C.y; tc[2] = C.z; tc[3] = C.w;
          if (dil >= 4) {
            const float4 L = *reinterpret_cast<const float4*>(rp - dil), R = *reinterpret_cast<const float4*>(rp + dil);
            tl[0] = L.x; tl[1] = L.y; tl[2] = L.z; tl[3] = L.w;
            trr[0] = R.x; trr[1] = R.y; trr[2] = R.z; trr[3] = R.w;
          } else {
            const float4 L = *reinterpret_cast<const float4*>(rp - 4), R = *reinterpret_cast<const float4*>(rp + 4);
            if (dil == 1) {
              tl[0] = L.w; tl[1] = C.x; tl[2] = C.y; tl[3] = C.z;
              trr[0] = C.y; trr[1] = C.z; trr[2] = C.w; trr[3] = R.x;
            } else {
              tl[0] = L.z; tl[1] = L.w; tl[2] = C.x; tl[3] = C.y;
              trr[0] = C.z; trr[1] = C.w; trr[2] = R.x; trr[3] = R.y;
            }
          }
          csn_cfp w0 = wc + (3 * dy) * 8, w1 = w0 + 8, w2 = w0 + 16;
#pragma unroll
          for (int co = 0; co < 4; ++co)
#pragma unroll
            for (int p = 0; p < 4; ++p)
              acc[d][co][p] = fmaf(w2[co], trr[p], fmaf(w1[co], tc[p], fmaf(w0[co], tl[p], acc[d][co][p])));
          if (nco > 4) {
#pragma unroll
            for (int co = 4; co < 8; ++co)
#pragma unroll
              for (int p = 0; p < 4; ++p)
                acc[d][co][p] = fmaf(w2[co], trr[p], fmaf(w1[co], tc[p], fmaf(w0[co], tl[p], acc[d][co][p])));
          }
        }
      }
    }
  }
  if (!active) return;
  csn_cfp scale = csn_const(a.scale), shift = csn_const(a.shift), alpha = csn_const(a.alpha);
  float* __restrict__ op = a.out + (int64_t)b * a.cout * hw + (int64_t)(y0 + qy) * W + 4 * qx;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int nco = a.dch[D0 + d];
#pragma unroll
    for (int co = 0; co < 8; ++co) {
      if (co < nco) {
        const int oc = a.cobase[D0 + d] + co;
        const float sc = scale[oc], sh = shift[oc], al = alpha[oc];
        *reinterpret_cast<float4*>(op + (int64_t)oc * hw) =
            make_float4(csn_epi(acc[d][co][0], sc, sh, al), csn_epi(acc[d][co][1], sc, sh, al),
                        csn_epi(acc[d][co][2], sc, sh, al), csn_epi(acc[d][co][3], sc, sh, al));
      }
    }
  }
}

// geometry of the tiled version: rows per band / channels per staged chunk (0: the shape does not qualify)
static bool ms2_geometry(const MsArgs& a, int halo, int& RB, int& CC, size_t& lds) {
  if ((a.W & 3) || a.W < 4 || (a.W >> 2) > CSN_BLOCK) return false;
  for (int d = 0; d < 5; ++d)
    if (a.dch[d] > 8) return false;
  const int QW = a.W >> 2;
  RB = CSN_BLOCK / QW;
  if (RB > a.H) RB = a.H;
  if (RB >= 8) RB &= ~7;                      // whole bands of 8 / 16 / 24 rows
  const int WP = a.W + 2 * halo, RT = RB + 2 * halo;
  if ((WP >> 2) > CSN_BLOCK) return false;
  const size_t plane = (size_t)RT * WP * sizeof(float);
  CC = (int)((48 * 1024) / plane);
  if (CC < 1) CC = 1;
  if (CC > 8) CC = 8;
  if (CC > a.cin) CC = a.cin;
  lds = plane * CC;
  return lds <= 96 * 1024;
}

int csn_launch_ms(const MsArgs& a, void* stream) {
  const int hw = a.H * a.W;
#ifndef CSN_CPU_EMU
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&msblock2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&msblock2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
#endif
  // the LDS-tiled version measured 2.6x SLOWER than the per-pixel kernel on MI355X (0.69 vs 0.26 ms for the three MSBlocks at
  // batch 64, profiles/r2_notes.md): opt-in only (CSN_MS_TILED=1, tests)
  static const bool force_old = !(std::getenv("CSN_MS_TILED") && std::getenv("CSN_MS_TILED")[0] == '1');
  int RB0, CC0, RB1, CC1;
  size_t l0, l1;
  if (!force_old && !a.a16 && ms2_geometry(a, 4, RB0, CC0, l0) && ms2_geometry(a, 16, RB1, CC1, l1)) {
    if (a.dch[0] + a.dch[1] + a.dch[2] > 0)
      CSN_LAUNCH(msblock2_kernel<0>, dim3((unsigned)(a.B * ((a.H + RB0 - 1) / RB0))), dim3(CSN_BLOCK), l0, stream, a, RB0, CC0);
    if (a.dch[3] + a.dch[4] > 0)
      CSN_LAUNCH(msblock2_kernel<1>, dim3((unsigned)(a.B * ((a.H + RB1 - 1) / RB1))), dim3(CSN_BLOCK), l1, stream, a, RB1, CC1);
    return (int)hipGetLastError();
  }
  const int tiles = ((hw + CSN_BLOCK - 1) / CSN_BLOCK) * a.B;
  const dim3 grid((unsigned)(((tiles + 7) / 8) * 5 * 8));
  if (a.a16) CSN_LAUNCH((msblock_kernel<csn_bf16>), grid, dim3(CSN_BLOCK), 0, stream, a);
  else CSN_LAUNCH((msblock_kernel<float>), grid, dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
