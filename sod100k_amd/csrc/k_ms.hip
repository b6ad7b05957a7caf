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

// ------------------------------------------------------------------------------------------------ four pixels per lane
// Round 3 (profiles/r3_notes.md): the per-pixel kernel above is bound by L1 tag traffic, not arithmetic -- its tap loads are
// 64 x 4 B rows shifted by the dilation, i.e. 2-3 cache lines per wave-load of which a third of the bytes is used (161 loads
// per wave and dilation at 8.6 cycles each).  Here a lane owns FOUR consecutive pixels of a row (W % 4 == 0): for the
// dilations 4, 8, 16 every tap of the quad is ONE aligned 128-bit load (whole cache lines, a quarter of the load
// instructions), dilations 1 and 2 take the quad plus two edge pieces per row; row / column padding = out-of-range offsets
// of a bounded buffer resource (a shifted quad is entirely inside or outside the row because x0, W and d are multiples of
// 4, the edge pieces of d = 1, 2 likewise).  NCO x 4 accumulators, weights wave-uniform (s_load), one dilation per block.
// DC: column step of the taps inside a lane's row window -- 1, 2 (those dilations) or 4 (dilations 4, 8, 16: whole quads)
template <int NCO, int DC>
__device__ __forceinline__ void msq_group(const MsArgs& a, csn_buf rb, const unsigned (&ro)[3], unsigned dl, unsigned dr,
                                          unsigned cs4, int cinp, csn_cfp wg, float* __restrict__ op, int hw, int g, int d, bool valid) {
  float acc[NCO][4];
#pragma unroll
  for (int co = 0; co < NCO; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;
  for (int ci = 0; ci < a.cin; ++ci) {
    const unsigned so = (unsigned)ci * cs4;
    float v[3][12];   // row r: columns x0 - 4 .. x0 + 7 as far as the dilation needs them
    if (DC == 4) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float4 l = csn_ld4(rb, ro[r] + dl, so), c = csn_ld4(rb, ro[r], so), rr = csn_ld4(rb, ro[r] + dr, so);
        v[r][0] = l.x; v[r][1] = l.y; v[r][2] = l.z; v[r][3] = l.w;
        v[r][4] = c.x; v[r][5] = c.y; v[r][6] = c.z; v[r][7] = c.w;
        v[r][8] = rr.x; v[r][9] = rr.y; v[r][10] = rr.z; v[r][11] = rr.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float2 l = csn_ld2(rb, ro[r] + dl, so), rr = csn_ld2(rb, ro[r] + dr, so);   // columns x0 - 2, x0 - 1 / x0 + 4, x0 + 5
        const float4 c = csn_ld4(rb, ro[r], so);
        v[r][0] = 0.f; v[r][1] = 0.f; v[r][2] = l.x; v[r][3] = l.y;
        v[r][4] = c.x; v[r][5] = c.y; v[r][6] = c.z; v[r][7] = c.w;
        v[r][8] = rr.x; v[r][9] = rr.y; v[r][10] = 0.f; v[r][11] = 0.f;
      }
    }
    csn_cfp wc = wg + ci * 72;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, dx = t % 3 - 1;
#pragma unroll
      for (int co = 0; co < NCO; ++co) {
        const float w = wc[t * 8 + co];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(w, v[r][4 + DC * dx + p], acc[co][p]);
      }
    }
  }
  csn_cfp scale = csn_const(a.scale), shift = csn_const(a.shift), alpha = csn_const(a.alpha);
#pragma unroll
  for (int co = 0; co < NCO; ++co) {
    if (valid) {
      const int oc = a.cobase[d] + g * 8 + co;
      const float sc = scale[oc], shf = shift[oc], al = alpha[oc];
      *reinterpret_cast<float4*>(op + (int64_t)oc * hw) =
          make_float4(csn_epi(acc[co][0], sc, shf, al), csn_epi(acc[co][1], sc, shf, al), csn_epi(acc[co][2], sc, shf, al),
                      csn_epi(acc[co][3], sc, shf, al));
    }
  }
}

__global__ __launch_bounds__(CSN_BLOCK) void msq_kernel(MsArgs a) {
  const int H = a.H, W = a.W, QW = W >> 2;
  const int hw = H * W, nq = H * QW;
  const int ntx = (nq + CSN_BLOCK - 1) / CSN_BLOCK;
  // same XCD-aware order as msblock_kernel: the five dilation blocks of a tile go to one XCD, back to back
  const int slot = blockIdx.x >> 3;
  const int tile = (slot / 5) * 8 + (blockIdx.x & 7);
  if (tile >= ntx * a.B) return;
  const int d = slot % 5;
  const int nco = a.dch[d];
  if (nco == 0) return;
  const int b = tile / ntx;
  const int q0 = (tile - b * ntx) * CSN_BLOCK + threadIdx.x;
  const bool valid = q0 < nq;
  const int q = valid ? q0 : nq - 1;
  const int y = q / QW, x0 = 4 * (q - y * QW);
  const int dil = 1 << d;
  const csn_buf rb = csn_make_buf_n(a.in + (int64_t)b * a.cin * hw, (unsigned)(a.cin * hw) * 4u);
  unsigned ro[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int yy = y + (r - 1) * dil;
    ro[r] = (yy >= 0 && yy < H) ? (unsigned)(yy * W + x0) * 4u : 0x80000000u;
  }
  // left / right piece: a quad at x0 -+ dil (dil >= 4) or the two columns next to the quad (dil = 1, 2)
  const int step = dil >= 4 ? dil : 2, rstep = dil >= 4 ? dil : 4;
  const unsigned dl = x0 - step >= 0 ? (unsigned)(-step * 4) : 0x40000000u;
  const unsigned dr = x0 + rstep + (dil >= 4 ? 3 : 1) < W ? (unsigned)(rstep * 4) : 0x40000000u;
  float* __restrict__ op = a.out + (int64_t)b * a.cout * hw + (int64_t)y * W + x0;
  const int cinp = (a.cin + 1) & ~1;
  const int ngrp = (nco + 7) >> 3;
  for (int g = 0; g < ngrp; ++g) {
    csn_cfp wg = csn_const(a.w[d]) + (int64_t)g * cinp * 72;
    const int live = min(8, nco - 8 * g);
#define MSQ_CASE(N)                                                                                                        \
  case N:                                                                                                                   \
    if (d == 0) msq_group<N, 1>(a, rb, ro, dl, dr, (unsigned)hw * 4u, cinp, wg, op, hw, g, d, valid);                       \
    else if (d == 1) msq_group<N, 2>(a, rb, ro, dl, dr, (unsigned)hw * 4u, cinp, wg, op, hw, g, d, valid);                  \
    else msq_group<N, 4>(a, rb, ro, dl, dr, (unsigned)hw * 4u, cinp, wg, op, hw, g, d, valid);                              \
    break;
    switch (live) {
      MSQ_CASE(1) MSQ_CASE(2) MSQ_CASE(3) MSQ_CASE(4) MSQ_CASE(5) MSQ_CASE(6) MSQ_CASE(7)
      default:
        if (d == 0) msq_group<8, 1>(a, rb, ro, dl, dr, (unsigned)hw * 4u, cinp, wg, op, hw, g, d, valid);
        else if (d == 1) msq_group<8, 2>(a, rb, ro, dl, dr, (unsigned)hw * 4u, cinp, wg, op, hw, g, d, valid);
        else msq_group<8, 4>(a, rb, ro, dl, dr, (unsigned)hw * 4u, cinp, wg, op, hw, g, d, valid);
        break;
    }
#undef MSQ_CASE
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
  static const bool quad = !(std::getenv("CSN_MS_QUAD") && std::getenv("CSN_MS_QUAD")[0] == '0');
  // four pixels per lane (float, rows of whole quads); small maps keep one pixel per lane (28^2 x 64 images is 320 blocks of
  // quads: 42 us against 29 us, profiles/r3_notes.md)
  if (quad && !a.a16 && (a.W & 3) == 0 && a.H * (a.W >> 2) >= 512) {
    const int tq = ((a.H * (a.W >> 2) + CSN_BLOCK - 1) / CSN_BLOCK) * a.B;
    CSN_LAUNCH(msq_kernel, dim3((unsigned)(((tq + 7) / 8) * 5 * 8)), dim3(CSN_BLOCK), 0, stream, a);
    return (int)hipGetLastError();
  }
  const int tiles = ((hw + CSN_BLOCK - 1) / CSN_BLOCK) * a.B;
  const dim3 grid((unsigned)(((tiles + 7) / 8) * 5 * 8));
  if (a.a16) CSN_LAUNCH((msblock_kernel<csn_bf16>), grid, dim3(CSN_BLOCK), 0, stream, a);
  else CSN_LAUNCH((msblock_kernel<float>), grid, dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
