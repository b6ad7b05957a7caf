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

// ---------------------------------------------------------------------------------------------------------------------
// msr_kernel: four pixels per lane with vertical reuse.  A lane owns QUADS of four consecutive pixels: for dilations 4 / 8 / 16 every
// tap of a quad is one aligned 128-bit load (a shifted quad is entirely inside or outside the row because x0, W and d are multiples of
// 4; row / column padding = out-of-range offsets of a bounded buffer resource), dilations 1 and 2 take the quad plus two edge pairs per
// row; NCO x 4 accumulators per quad (v_pk_fma_f32 over pixel pairs), weights wave-uniform (s_load), one dilation per wave.  A lane
// that owned ONE quad (msq_kernel, earlier in round 3) loaded 9 pieces per channel and used none twice (the taps of a dilated window do
// not overlap): bound by the number of tap loads going through L1 (profiles/r3_notes.md).  Here a lane owns R quads in rows y0, y0 + d, ..., y0 + (R - 1) d of one column block:
// the rows y0 - d .. y0 + R d serve all of them -- 3 (R + 2) piece loads per channel for R quads instead of 9 R (R = 4: half,
// for launches with <= 3 output channels per dilation; R = 2: two thirds, <= 8 output channels, there with the next channel's
// rows in flight during the FMAs -- few waves per SIMD on the 56^2 map).  Measured alternatives (R = 6, 8; two pixels per
// lane; double buffering with R = 4): profiles/r3_notes.md.  Lanes of one (image, dilation): x quads
// fastest, then the d row residues, then the groups of R d rows; items are WAVES (64 lanes of one image and dilation), all
// waves of an image on one XCD.
template <int DC, int R, typename AT>
__device__ __forceinline__ void msr_load(csn_buf rb, const unsigned (&ro)[R + 2], unsigned dl, unsigned dr, unsigned so,
                                         float (&v)[R + 2][12]) {
#pragma unroll
  for (int r = 0; r < R + 2; ++r) {
    if (DC == 4) {
      const float4 l = csn_bufacc<AT>::ld4(rb, ro[r] + dl, so), c = csn_bufacc<AT>::ld4(rb, ro[r], so), rr = csn_bufacc<AT>::ld4(rb, ro[r] + dr, so);
      v[r][0] = l.x; v[r][1] = l.y; v[r][2] = l.z; v[r][3] = l.w;
      v[r][4] = c.x; v[r][5] = c.y; v[r][6] = c.z; v[r][7] = c.w;
      v[r][8] = rr.x; v[r][9] = rr.y; v[r][10] = rr.z; v[r][11] = rr.w;
    } else {
      const float2 l = csn_bufacc<AT>::ld2(rb, ro[r] + dl, so), rr = csn_bufacc<AT>::ld2(rb, ro[r] + dr, so);
      const float4 c = csn_bufacc<AT>::ld4(rb, ro[r], so);
      v[r][0] = 0.f; v[r][1] = 0.f; v[r][2] = l.x; v[r][3] = l.y;
      v[r][4] = c.x; v[r][5] = c.y; v[r][6] = c.z; v[r][7] = c.w;
      v[r][8] = rr.x; v[r][9] = rr.y; v[r][10] = 0.f; v[r][11] = 0.f;
    }
  }
}

template <int NCO, int DC, int R>
__device__ __forceinline__ void msr_fma(const float (&v)[R + 2][12], csn_cfp wc, float (&acc)[R][NCO][4]) {
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int tr = t / 3, dx = t % 3 - 1;
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      const float w = wc[t * 8 + co];
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[i][co][p] = fmaf(w, v[i + tr][4 + DC * dx + p], acc[i][co][p]);
    }
  }
}

template <int NCO, int DC, int R, bool DB, typename AT>
__device__ __forceinline__ void msr_group(const MsArgs& a, csn_buf rb, const unsigned (&ro)[R + 2], unsigned dl, unsigned dr,
                                          unsigned cs4, int cinp, csn_cfp wg, AT* __restrict__ op, int hw, int rstride, int g,
                                          int d, const bool (&st)[R]) {
  float acc[R][NCO][4];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[i][co][p] = 0.f;
  const int cin = a.cin;
  if (DB) {   // channel c + 1 in flight while channel c is contracted (two register sets)
    float vA[R + 2][12], vB[R + 2][12];
    msr_load<DC, R, AT>(rb, ro, dl, dr, 0u, vA);
    CSN_SCHED_FENCE();
    for (int ci = 0; ci < cin; ci += 2) {
      msr_load<DC, R, AT>(rb, ro, dl, dr, (unsigned)min(ci + 1, cin - 1) * cs4, vB);
      CSN_SCHED_FENCE();
      msr_fma<NCO, DC, R>(vA, wg + ci * 72, acc);
      msr_load<DC, R, AT>(rb, ro, dl, dr, (unsigned)min(ci + 2, cin - 1) * cs4, vA);
      CSN_SCHED_FENCE();
      if (ci + 1 < cin) msr_fma<NCO, DC, R>(vB, wg + (ci + 1) * 72, acc);
    }
  } else {
    for (int ci = 0; ci < cin; ++ci) {
      float v[R + 2][12];
      msr_load<DC, R, AT>(rb, ro, dl, dr, (unsigned)ci * cs4, v);
      msr_fma<NCO, DC, R>(v, wg + ci * 72, acc);
    }
  }
  csn_cfp scale = csn_const(a.scale), shift = csn_const(a.shift), alpha = csn_const(a.alpha);
#pragma unroll
  for (int co = 0; co < NCO; ++co) {
    const int oc = a.cobase[d] + g * 8 + co;
    const float sc = scale[oc], shf = shift[oc], al = alpha[oc];
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (st[i]) {
        AT* q = op + (int64_t)oc * hw + (int64_t)i * rstride;
        act_st4(q, make_float4(csn_epi(acc[i][co][0], sc, shf, al), csn_epi(acc[i][co][1], sc, shf, al),
                               csn_epi(acc[i][co][2], sc, shf, al), csn_epi(acc[i][co][3], sc, shf, al)));
      }
  }
}

// waves per (image, dilation): QW column quads x 2^d row residues x groups of R 2^d rows
__host__ __device__ inline int msr_waves(int H, int QW, int d, int R) {
  const int span = R << d;
  return (QW * (((H + span - 1) / span) << d) + 63) >> 6;
}

template <int R, bool DB, typename AT = float>
__global__ __launch_bounds__(CSN_BLOCK) void msr_kernel(MsArgs a) {
  constexpr unsigned E = (unsigned)sizeof(AT);   // (bfloat16: the train-mode forward of the bf16 step, identity epilogue)
  const int H = a.H, W = a.W, QW = W >> 2;
  const int hw = H * W;
  int wmax = 0;
#pragma unroll
  for (int dd = 0; dd < 5; ++dd)
    if (a.dch[dd] > 0) wmax = max(wmax, msr_waves(H, QW, dd, R));
  const int bpi = (5 * wmax + 3) >> 2;                 // blocks per image
  const int slot = blockIdx.x >> 3;
  const int b = (slot / bpi) * 8 + (int)(blockIdx.x & 7);
  if (b >= a.B) return;
#ifdef CSN_EMU_SEQ
  const int wave = (int)threadIdx.x >> 6;
#else
  const int wave = csn_readfirstlane((int)threadIdx.x >> 6);
#endif
  const int w = (slot % bpi) * 4 + wave;
  const int d = w / wmax, wv = w - d * wmax;
  if (d >= 5) return;
  const int nco = a.dch[d];
  if (nco == 0 || wv >= msr_waves(H, QW, d, R)) return;
  const int dil = 1 << d, span = R << d;
  const int idx = wv * 64 + ((int)threadIdx.x & 63);
  const int xq = idx % QW, rest = idx / QW;
  const int y0 = (rest >> d) * span + (rest & (dil - 1));
  const int x0 = 4 * xq;
  const bool valid = y0 < H;   // (rows past the last group: lanes idle)
  const csn_buf rb = csn_make_buf_n(act_cast<AT>(a.in) + (int64_t)b * a.cin * hw, (unsigned)(a.cin * hw) * E);
  unsigned ro[R + 2];
#pragma unroll
  for (int r = 0; r < R + 2; ++r) {
    const int yy = y0 + (r - 1) * dil;
    ro[r] = (valid && yy >= 0 && yy < H) ? (unsigned)(yy * W + x0) * E : 0x80000000u;
  }
  bool st[R];
#pragma unroll
  for (int i = 0; i < R; ++i) st[i] = valid && y0 + i * dil < H;
  // left / right piece: a quad at x0 -+ dil (dil >= 4), else the two columns next to the quad
  const int step = dil >= 4 ? dil : 2, rstep = dil >= 4 ? dil : 4;
  const int rlast = dil >= 4 ? 3 : 1;   // last column of the right piece
  const unsigned dl = x0 - step >= 0 ? (unsigned)(-step * (int)E) : 0x40000000u;
  const unsigned dr = x0 + rstep + rlast < W ? (unsigned)(rstep * (int)E) : 0x40000000u;
  AT* __restrict__ op = act_cast<AT>(a.out) + (int64_t)b * a.cout * hw + (int64_t)(valid ? y0 : 0) * W + x0;
  const int cinp = (a.cin + 1) & ~1;
  const int ngrp = (nco + 7) >> 3;
  for (int g = 0; g < ngrp; ++g) {
    csn_cfp wg = csn_const(a.w[d]) + (int64_t)g * cinp * 72;
    const int live = min(8, nco - 8 * g);
#define MSR_ARGS a, rb, ro, dl, dr, (unsigned)hw * E, cinp, wg, op, hw, dil * W, g, d, st
#define MSR_CASE(N)                                                                \
  case N:                                                                           \
    if (d == 0) msr_group<N, 1, R, DB, AT>(MSR_ARGS);                           \
    else if (d == 1) msr_group<N, 2, R, DB, AT>(MSR_ARGS);                      \
    else msr_group<N, 4, R, DB, AT>(MSR_ARGS);                                  \
    break;
    if (R > 2) {
      switch (live) { MSR_CASE(1) MSR_CASE(2) default: MSR_CASE(3) }
    } else {
      switch (live) { MSR_CASE(1) MSR_CASE(2) MSR_CASE(3) MSR_CASE(4) MSR_CASE(5) MSR_CASE(6) MSR_CASE(7) default: MSR_CASE(8) }
    }
#undef MSR_CASE
#undef MSR_ARGS
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ms_dx_kernel: the block's backward data pass (round 4; before: two launches of the generic tap kernel, 0.7 ms for the 112^2 block
// of a bf16 step).  The adjoint of five dilated convolutions that WRITE disjoint channel slices is one convolution that READS them:
//   dx[ci][p] = sum_d sum_co sum_t dz[cobase_d + co][p + off(t) 2^d] * 100 w_d[co][ci][8 - t]
// One lane owns one pixel and ALL its cin <= 56 sums (NG groups of 8 accumulators): every tap of dz is loaded exactly once per pixel
// -- 9 cout loads against the forward's 45 cin -- and meets 8 NG wave-uniform weights (s_load, CSN_PREP_MSDX image).  Channels in
// pairs with the 18 tap loads in flight before the first FMA, bounded resource + 9-bit tap mask as in msblock_kernel.
template <int NG, typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void ms_dx_kernel(MsDxArgs a) {
  constexpr unsigned E = (unsigned)sizeof(AT);
  constexpr int NC = NG * 8;
  const int H = a.H, W = a.W, hw = H * W;
  const int ntx = (hw + CSN_BLOCK - 1) / CSN_BLOCK;
  const int tile = blockIdx.x;
  if (tile >= ntx * a.B) return;
  const int b = tile / ntx;
  const int p0 = (tile - b * ntx) * CSN_BLOCK + threadIdx.x;
  const bool valid = p0 < hw;
  const int p = valid ? p0 : hw - 1;
  const int y = p / W, x = p - y * W;
  const csn_buf rb = csn_make_buf_n(act_cast<AT>(a.dz) + (int64_t)b * a.cout * hw, (unsigned)(a.cout * hw) * E);
  const unsigned lo = (unsigned)p * E;
  const unsigned cs4 = (unsigned)hw * E;
  float acc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) acc[i] = 0.f;
  for (int d = 0; d < 5; ++d) {
    const int nco = a.dch[d];
    if (nco == 0) continue;
    const int dil = 1 << d;
    const unsigned vm = ms_tap_mask(y, x, H, W, dil);
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * W + (t % 3 - 1)) * dil * (int)E;
    csn_cfp wd = csn_const(a.w[d]);
    const int ncop = (nco + 1) & ~1;
    for (int co = 0; co < ncop; co += 2) {
      float v[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned so = (unsigned)(a.cobase[d] + min(co + u, nco - 1)) * cs4;   // pad channel: its weights are zero
#pragma unroll
        for (int t = 0; t < 9; ++t) v[u][t] = csn_bufacc<AT>::ld1(rb, lo + (unsigned)toff[t], so);
      }
      CSN_SCHED_FENCE();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        csn_cfp wc = wd + (int64_t)(co + u) * 9 * NC;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float val = ((vm >> t) & 1u) ? v[u][t] : 0.f;
#pragma unroll
          for (int i = 0; i < NC; ++i) acc[i] = fmaf(wc[t * NC + i], val, acc[i]);
        }
      }
    }
  }
  if (valid) {
    AT* __restrict__ op = act_cast<AT>(a.dx) + (int64_t)b * a.cin * hw + p;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (i < a.cin) act_st(op + (int64_t)i * hw, acc[i]);
  }
}

int csn_launch_ms_dx(const MsDxArgs& a, void* stream) {
  if (a.ng < 1 || a.ng > 7 || a.cin > a.ng * 8) return 1;
  const int hw = a.H * a.W;
  const dim3 grid((unsigned)(((hw + CSN_BLOCK - 1) / CSN_BLOCK) * a.B));
#define MSDX_LAUNCH(G)                                                                                     \
  do {                                                                                                    \
    if (a.a16) CSN_LAUNCH((ms_dx_kernel<G, csn_bf16>), grid, dim3(CSN_BLOCK), 0, stream, a);               \
    else CSN_LAUNCH((ms_dx_kernel<G, float>), grid, dim3(CSN_BLOCK), 0, stream, a);                        \
  } while (0)
  switch (a.ng) {
    case 1: MSDX_LAUNCH(1); break;
    case 2: MSDX_LAUNCH(2); break;
    case 3: MSDX_LAUNCH(3); break;
    case 4: MSDX_LAUNCH(4); break;
    case 5: MSDX_LAUNCH(5); break;
    case 6: MSDX_LAUNCH(6); break;
    default: MSDX_LAUNCH(7); break;
  }
#undef MSDX_LAUNCH
  return (int)hipGetLastError();
}

int csn_launch_ms(const MsArgs& a, void* stream) {
  const int hw = a.H * a.W;
  // four pixels per lane (rows of whole quads, float and bfloat16 tensors); small maps keep one pixel per lane (28^2 x 64 images is
  // 320 blocks of quads: 42 us against 29 us, profiles/r3_notes.md).
  // Round 6, measured and NOT kept (profiles/r6_notes.md): the block from an LDS band (band + 16 halo rows of one channel staged per
  // step, double buffered; a lane owns pixel pairs and all dilations; packed FMAs; weights through LDS): 102 / 152 / 104 us for the
  // three blocks against 73 / 62 / 27 here -- 45 tap reads per pixel pair and channel make it LDS-bandwidth bound (1.8 us per channel
  // and band on the 112^2 map) on top of a barrier per channel with one block per CU.
  if ((a.W & 3) == 0 && a.H * (a.W >> 2) >= 512) {
    int mx = 0;
    for (int d = 0; d < 5; ++d) mx = a.dch[d] > mx ? a.dch[d] : mx;
    // R quads per lane in dilation-strided rows (profiles/r3_notes.md: 94 -> 68 us, 62 -> 57 us)
    const int R = mx <= 3 ? 4 : 2;
    int wmax = 0;
    for (int d = 0; d < 5; ++d)
      if (a.dch[d] > 0) { const int wv = msr_waves(a.H, a.W >> 2, d, R); wmax = wv > wmax ? wv : wmax; }
    const int bpi = (5 * wmax + 3) >> 2;
    const dim3 grid((unsigned)(((a.B + 7) / 8) * bpi * 8));
    if (a.a16) {   // (round 4: the train-mode forward of the bf16 step ran the one-pixel-per-lane kernel: 0.92 ms for three launches)
      if (R == 4) CSN_LAUNCH((msr_kernel<4, false, csn_bf16>), grid, dim3(CSN_BLOCK), 0, stream, a);
      else CSN_LAUNCH((msr_kernel<2, true, csn_bf16>), grid, dim3(CSN_BLOCK), 0, stream, a);
    } else if (R == 4) CSN_LAUNCH((msr_kernel<4, false>), grid, dim3(CSN_BLOCK), 0, stream, a);
    else CSN_LAUNCH((msr_kernel<2, true>), grid, dim3(CSN_BLOCK), 0, stream, a);
    return (int)hipGetLastError();
  }
  const int tiles = ((hw + CSN_BLOCK - 1) / CSN_BLOCK) * a.B;
  const dim3 grid((unsigned)(((tiles + 7) / 8) * 5 * 8));
  if (a.a16) CSN_LAUNCH((msblock_kernel<csn_bf16>), grid, dim3(CSN_BLOCK), 0, stream, a);
  else CSN_LAUNCH((msblock_kernel<float>), grid, dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
