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

__global__ __launch_bounds__(CSN_BLOCK) void msblock_kernel(MsArgs a) {
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
  const csn_buf rb = csn_make_buf_n(a.in + (int64_t)b * a.cin * hw, (unsigned)(a.cin * hw) * 4u);
  const unsigned lo = (unsigned)p * 4u;
  const unsigned cs4 = (unsigned)hw * 4u;
  float* __restrict__ op = a.out + (int64_t)b * a.cout * hw + p;
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
    for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * W + (t % 3 - 1)) * dil * 4;
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
          for (int t = 0; t < 9; ++t) v[u][t] = csn_ld1(rb, lo + (unsigned)toff[t], so);
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
          op[(int64_t)oc * hw] = csn_epi(acc[co], scale[oc], shift[oc], alpha[oc]);
        }
      }
    }
  }
}

int csn_launch_ms(const MsArgs& a, void* stream) {
  const int hw = a.H * a.W;
  const int tiles = ((hw + CSN_BLOCK - 1) / CSN_BLOCK) * a.B;
  const dim3 grid((unsigned)(((tiles + 7) / 8) * 5 * 8));
  CSN_LAUNCH(msblock_kernel, grid, dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
