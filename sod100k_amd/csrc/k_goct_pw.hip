// k_goct_pw.hip -- gOctConv with 1x1 kernels + BN + PReLU, ALL output branches of a unit per block.
//
// Reference semantics (CSNet/model/csnet.py):
//   gOctaveConv.forward 664-726:  y_j = sum_i T_ij(x_i), weight block W[co_j, ci_i];
//       i == j : conv(x_i)                                     (716-717)
//       i >  j : bilinear_up_{2^(i-j)}( conv(x_i) )            (702-707)  "low -> high"
//       i <  j : conv( max_pool_{2^(j-i)}(x_i) )               (708-714)  "high -> low"
//   gOctaveCBR.forward 778-792:   PReLU_j(BN_j(y_j)) per output branch (eval BN folded to scale/shift)
//   also used for cls_layer (1x1 + bias, csnet.py:306-308,381) with scale=1, shift=bias, alpha=1.
// A 1x1 convolution commutes with bilinear interpolation (both linear, the conv acts on channels only), so
// the low->high term is evaluated as conv(bilinear_up(x_i)): every output branch becomes ONE contraction
// over the gathered vector [x_j ; maxpool(x_i<j) ; bilinear(x_i>j)] of its pixel -- no partial sums to
// exchange between resolutions and no barrier in the main loop.
//
// MI355X mapping.  A block owns a 16x32 tile of branch 0 plus the matching 8x16 / 4x8 tiles of branches
// 1 / 2 and walks tiles grid-stride.  Work items are 64-pixel groups; the 4 waves of a block take groups
// round-robin and never synchronise with each other after the weight image has been staged in LDS:
//   gather   16 input channels at a time: every lane fetches ITS pixel (own resolution: one coalesced dword
//            per channel; max-pool / bilinear taps hit L1/L2 -- their HBM bytes are paid by the pass that
//            owns that resolution) in straight-line, unrolled loops so that 8-16 loads are in flight per
//            lane, and drops the values into the wave's private LDS panel x[k][64 px];
//   contract v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain): A = W[row0 + (lane&15)][k0 + (lane>>4)] and
//            B = x[k0 + (lane>>4)][16 s + (lane&15)] are single conflict-free ds_read_b32 each, one A
//            register feeds the 4 pixel sub-groups, accumulators (<= 64 rows x 64 px) stay in VGPRs across
//            the channel chunks; the matrix pipe does the FMAs, VALU only addressing + epilogue;
//   store    folded BN + PReLU, 16 consecutive pixels of a row per 16-lane group.
// All loops are run-time: one kernel for every channel plan.  HBM traffic = unit inputs once + outputs once.
#include "csn_kernels.h"

#ifdef CSN_CPU_EMU
struct csn_f4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};
// lanes of a wave run as sequential fibers: make LDS hand-offs inside a wave visible
#define CSN_WAVE_SYNC() __syncthreads()
#else
typedef float csn_f4 __attribute__((ext_vector_type(4)));
// a wave executes in lockstep and its LDS operations retire in order: only stop the compiler from
// moving LDS accesses across the hand-off
#define CSN_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

#define PW_KC 16   // channels per LDS panel
#define PW_XP 80   // panel row pitch (floats): == 16 (mod 32), so the two k rows a 32-lane half reads hit disjoint banks
#define PW_EP 68   // pitch of the epilogue transpose: rows 4 apart land on the other half of the banks

// acc[i] += sum_{u<4} W[row0 + (lane>>4)*4 + i][k0 + u] * x[k0 + u][16 s + (lane&15)]
// wt = &W[row0][k0] in the LDS weight image (row pitch `stride`), xs = &x[k0][16 s] in the wave's panel.
__device__ __forceinline__ void pw_mfma16(const float* wt, int stride, const float* xs, int lane, csn_f4& acc) {
#ifdef CSN_CPU_EMU
  const int col = lane & 15;
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 4) * 4 + i;
    float a = acc[i];
    for (int u = 0; u < 4; ++u) a = fmaf(wt[row * stride + u], xs[u * PW_XP + col], a);
    acc[i] = a;
  }
#else
  const float av = wt[(lane & 15) * stride + (lane >> 4)];
  const float bv = xs[(lane >> 4) * PW_XP + (lane & 15)];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
#endif
}

typedef const CSN_CONST_AS PwPass* PwPassP;

// Gather channels [c_lo, c_hi) of one slice for this lane's pixel into the panel rows starting at xrow
// (`rmax` rows are left in the panel).  Loads go through a buffer resource whose base is channel c_lo of
// image b (wave-uniform, SGPRs); the lane contributes one 32-bit byte offset, the channel a uniform SGPR
// offset.  Every batch issues ALL its loads before the first use (fixed trip count, channel index clamped
// instead of predicated: a predicated load would be waited for at the join), so a lane has 16-32 loads
// in flight; rows written past the slice are overwritten by the next slice / the zero padding.
template <int NB>
__device__ __forceinline__ void pw_batch_own(csn_buf rb, unsigned lo, unsigned cs4, int k0, int n, int rmax,
                                             float* xrow) {
  float v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = csn_ld1(rb, lo, (unsigned)min(k0 + j, n - 1) * cs4);
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * PW_XP] = v[j];
}

template <int NB>
__device__ __forceinline__ void pw_batch_pool2(csn_buf rb, unsigned lo, unsigned cs4, unsigned ws4, int k0, int n,
                                               int rmax, float* xrow) {
  float2 a0[NB], a1[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, n - 1) * cs4;
    a0[j] = csn_ld2(rb, lo, so);
    a1[j] = csn_ld2(rb, lo, so + ws4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * PW_XP] = fmaxf(fmaxf(a0[j].x, a0[j].y), fmaxf(a1[j].x, a1[j].y));
}

template <int NB>
__device__ __forceinline__ void pw_batch_pool4(csn_buf rb, unsigned lo, unsigned cs4, unsigned ws4, int k0, int n,
                                               int rmax, float* xrow) {
  float4 q[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, n - 1) * cs4;
#pragma unroll
    for (int r = 0; r < 4; ++r) q[j][r] = csn_ld4(rb, lo, so + r * ws4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    float m = -3.402823466e+38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(fmaxf(q[j][r].x, q[j][r].y), fmaxf(q[j][r].z, q[j][r].w)));
    if (k0 + j < rmax) xrow[(k0 + j) * PW_XP] = m;
  }
}

template <int NB>
__device__ __forceinline__ void pw_batch_up(csn_buf rb, unsigned o00, unsigned o01, unsigned o10, unsigned o11,
                                            float w00, float w01, float w10, float w11, unsigned cs4, int k0, int n,
                                            int rmax, float* xrow) {
  float t0[NB], t1[NB], t2[NB], t3[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const unsigned so = (unsigned)min(k0 + j, n - 1) * cs4;
    t0[j] = csn_ld1(rb, o00, so);
    t1[j] = csn_ld1(rb, o01, so);
    t2[j] = csn_ld1(rb, o10, so);
    t3[j] = csn_ld1(rb, o11, so);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * PW_XP] = w00 * t0[j] + w01 * t1[j] + w10 * t2[j] + w11 * t3[j];
}

// 3x3 taps of an own-resolution slice: gathered entry kk = 9*ch + t, t = 3*(dy+1) + (dx+1).  `vm` has
// bit t set when tap t of this lane's pixel lies inside the image (zero padding otherwise).
template <int NB>
__device__ __forceinline__ void pw_batch_taps(csn_buf rb, unsigned lo, unsigned cs4, int Wr, int dil, unsigned vm,
                                              int k_lo, int k0, int n, int rmax, float* xrow) {
  float v[NB];
  unsigned m[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int kk = k_lo + min(k0 + j, n - 1);
    const int ch = kk / 9, t = kk - 9 * ch;
    const int dy = t / 3 - 1, dx = t - 3 * (t / 3) - 1;
    v[j] = csn_ld1(rb, lo + (unsigned)((dy * Wr + dx) * dil * 4), (unsigned)ch * cs4);
    m[j] = (vm >> t) & 1u;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax) xrow[(k0 + j) * PW_XP] = m[j] ? v[j] : 0.f;
}

// 3x3 taps of a 2x2-max-pooled slice (source at twice the resolution).
template <int NB>
__device__ __forceinline__ void pw_batch_pool2_taps(csn_buf rb, unsigned lo, unsigned cs4, unsigned ws4, unsigned vm,
                                                    int k_lo, int k0, int n, int rmax, float* xrow) {
  float2 a0[NB], a1[NB];
  unsigned m[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int kk = k_lo + min(k0 + j, n - 1);
    const int ch = kk / 9, t = kk - 9 * ch;
    const int dy = t / 3 - 1, dx = t - 3 * (t / 3) - 1;
    const unsigned vo = lo + (unsigned)(2 * dy) * ws4 + (unsigned)(8 * dx);
    a0[j] = csn_ld2(rb, vo, (unsigned)ch * cs4);
    a1[j] = csn_ld2(rb, vo + ws4, (unsigned)ch * cs4);
    m[j] = (vm >> t) & 1u;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (k0 + j < rmax)
      xrow[(k0 + j) * PW_XP] = m[j] ? fmaxf(fmaxf(a0[j].x, a0[j].y), fmaxf(a1[j].x, a1[j].y)) : 0.f;
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

__device__ __forceinline__ void pw_gather_slice(PwPassP ps, int s, int c_lo, int c_hi, float* xrow, int rmax, int b,
                                                int y, int x, int Hr, int Wr) {
  const int mode = ps->src[s].mode;
  const int n = c_hi - c_lo;   // 1..16
  if (mode == PW_OWN) {
    const unsigned cs = (unsigned)(Hr * Wr);
    const csn_buf rb = csn_make_buf(ps->src[s].ptr + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned lo = (unsigned)(y * Wr + x) * 4u;
    if (n <= 8) pw_batch_own<8>(rb, lo, cs * 4u, 0, n, rmax, xrow);
    else pw_batch_own<16>(rb, lo, cs * 4u, 0, n, rmax, xrow);
  } else if (mode == PW_POOL2) {
    const unsigned Ws = (unsigned)Wr * 2u;
    const unsigned cs = (unsigned)(Hr * 2) * Ws;
    const csn_buf rb = csn_make_buf(ps->src[s].ptr + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned lo = ((unsigned)(2 * y) * Ws + 2u * x) * 4u;
    for (int k0 = 0; k0 < n; k0 += 8) pw_batch_pool2<8>(rb, lo, cs * 4u, Ws * 4u, k0, n, rmax, xrow);
  } else if (mode == PW_POOL4) {
    const unsigned Ws = (unsigned)Wr * 4u;
    const unsigned cs = (unsigned)(Hr * 4) * Ws;
    const csn_buf rb = csn_make_buf(ps->src[s].ptr + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned lo = ((unsigned)(4 * y) * Ws + 4u * x) * 4u;
    for (int k0 = 0; k0 < n; k0 += 2) pw_batch_pool4<2>(rb, lo, cs * 4u, Ws * 4u, k0, n, rmax, xrow);
  } else if (mode == PW_TAPS) {
    const unsigned cs = (unsigned)(Hr * Wr);
    const int dil = ps->src[s].dil;
    const csn_buf rb = csn_make_buf_n(ps->src[s].ptr + (int64_t)b * ps->src[s].Ctot * cs,
                                      (unsigned)ps->src[s].Ctot * cs * 4u);
    const unsigned lo = (unsigned)(y * Wr + x) * 4u;
    const unsigned vm = pw_tap_mask(y, x, Hr, Wr, dil);
    pw_batch_taps<16>(rb, lo, cs * 4u, Wr, dil, vm, c_lo, 0, n, rmax, xrow);
  } else if (mode == PW_POOL2_TAPS) {
    const unsigned Ws = (unsigned)Wr * 2u;
    const unsigned cs = (unsigned)(Hr * 2) * Ws;
    const csn_buf rb = csn_make_buf_n(ps->src[s].ptr + (int64_t)b * ps->src[s].Ctot * cs,
                                      (unsigned)ps->src[s].Ctot * cs * 4u);
    const unsigned lo = ((unsigned)(2 * y) * Ws + 2u * x) * 4u;
    const unsigned vm = pw_tap_mask(y, x, Hr, Wr, 1);
    for (int k0 = 0; k0 < n; k0 += 8) pw_batch_pool2_taps<8>(rb, lo, cs * 4u, Ws * 4u, vm, c_lo, k0, n, rmax, xrow);
  } else {  // bilinear from a 2x / 4x coarser branch, align_corners=False
    const int sh = mode == PW_UP2 ? 1 : 2;
    const int Hs = Hr >> sh, Ws = Wr >> sh;
    int y0, y1, x0, x1;
    float ly, lx;
    csn_bilin(y, mode == PW_UP2 ? 0.5f : 0.25f, Hs, y0, y1, ly);
    csn_bilin(x, mode == PW_UP2 ? 0.5f : 0.25f, Ws, x0, x1, lx);
    const unsigned cs = (unsigned)(Hs * Ws);
    const csn_buf rb = csn_make_buf(ps->src[s].ptr + ((int64_t)b * ps->src[s].Ctot + c_lo) * cs);
    const unsigned o00 = (unsigned)(y0 * Ws + x0) * 4u, o01 = (unsigned)(y0 * Ws + x1) * 4u,
                   o10 = (unsigned)(y1 * Ws + x0) * 4u, o11 = (unsigned)(y1 * Ws + x1) * 4u;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    for (int k0 = 0; k0 < n; k0 += 8)
      pw_batch_up<8>(rb, o00, o01, o10, o11, w00, w01, w10, w11, cs * 4u, k0, n, rmax, xrow);
  }
}

// NT row tiles (16 rows each) x 4 pixel sub-groups against the `kn` (multiple of 4) channels of the panel.
template <int NT>
__device__ __forceinline__ void pw_contract(const float* wl, int stride, const float* xb, int kn, int lane,
                                            csn_f4 (&acc)[NT][4]) {
  for (int k0 = 0; k0 < kn; k0 += 4) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s = 0; s < 4; ++s) pw_mfma16(wl + (16 * t) * stride + k0, stride, xb + k0 * PW_XP + 16 * s, lane, acc[t][s]);
    }
  }
}

// One sweep of NT row tiles (16 output channels each) for the wave's current 64-pixel group: channel chunks
// are gathered into the panel and contracted, then every tile is transposed through the panel so that each
// lane gets ITS pixel back and the wave stores 256 contiguous bytes per output channel (buffer stores: one
// VGPR offset per lane, the channel offset rides in an SGPR; lanes outside the image are exec-masked).
template <int NT>
__device__ __forceinline__ void pw_sweep(PwPassP ps, const float* wl0, float* xb, int row0, int b, int gy, int gx,
                                         int Hr, int Wr, unsigned ovoff, bool valid, int lane) {
  const int cin = ps->cin, cin4 = ps->cin4, nrows = ps->nrows, stride = ps->w_stride;
  const int c1 = ps->src[0].K, c2 = c1 + ps->src[1].K;
  csn_f4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][s][i] = 0.f;
  for (int kc = 0; kc < cin4; kc += PW_KC) {
    const int kend = min(kc + PW_KC, cin4);
    CSN_WAVE_SYNC();  // previous panel fully consumed
    if (kc < c1) pw_gather_slice(ps, 0, kc, min(kend, c1), xb + lane, PW_KC, b, gy, gx, Hr, Wr);
    if (max(kc, c1) < min(kend, c2)) {
      const int r0 = max(kc, c1) - kc;
      pw_gather_slice(ps, 1, max(kc, c1) - c1, min(kend, c2) - c1, xb + r0 * PW_XP + lane, PW_KC - r0, b, gy, gx, Hr, Wr);
    }
    if (max(kc, c2) < min(kend, cin)) {
      const int r0 = max(kc, c2) - kc;
      pw_gather_slice(ps, 2, max(kc, c2) - c2, min(kend, cin) - c2, xb + r0 * PW_XP + lane, PW_KC - r0, b, gy, gx, Hr, Wr);
    }
    for (int k = max(cin, kc); k < kend; ++k) xb[(k - kc) * PW_XP + lane] = 0.f;  // pad to a multiple of 4
    CSN_WAVE_SYNC();  // panel complete
    pw_contract<NT>(wl0 + row0 * stride + kc, stride, xb, kend - kc, lane, acc);
  }
  const unsigned cs4 = (unsigned)(Hr * Wr) * 4u;
  const csn_buf ob = csn_make_buf(ps->out + (int64_t)b * ps->out_ctot * (Hr * Wr));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    CSN_WAVE_SYNC();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) xb[((lane >> 4) * 4 + i) * PW_EP + 16 * s + (lane & 15)] = acc[t][s][i];
    CSN_WAVE_SYNC();
    const int rbase = row0 + 16 * t;
    const int rn = min(16, nrows - rbase);
    csn_cfp scale = csn_const(ps->scale) + rbase, shift = csn_const(ps->shift) + rbase, alpha = csn_const(ps->alpha) + rbase;
#pragma unroll 4
    for (int rr = 0; rr < rn; ++rr) {
      const float val = csn_epi(xb[rr * PW_EP + lane], scale[rr], shift[rr], alpha[rr]);
      if (valid) csn_st1(ob, ovoff, (unsigned)(rbase + rr) * cs4, val);
    }
  }
}

__global__ __launch_bounds__(CSN_BLOCK, 4) void goct_pw_kernel(PwArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS PwArgs* a = CSN_KERNARG(PwArgs, a_byval);
  const int tid = threadIdx.x;
  // ---- stage the unit's weight image (all passes, rows padded to 16, zero filled) into LDS once ----
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a->wimg);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int n4 = a->wimg_floats >> 2;
    for (int i = tid; i < n4; i += CSN_BLOCK) dst[i] = src[i];
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  float* xb = lds + a->wimg_floats + wave * (PW_KC * PW_XP);   // this wave's x[k][64 px] panel
  const int H0 = a->H0, W0 = a->W0, npass = a->npass;
  const int tiles_xy = a->tiles_x * a->tiles_y;
  const int ntiles = tiles_xy * a->B;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / tiles_xy;
    const int txy = tile - b * tiles_xy;
    const int tyl = a->ty_log2;   // tile height 16 / 8 / 4 rows of branch 0 (small maps get more, smaller tiles)
    const int ty0 = (txy / a->tiles_x) << tyl, tx0 = (txy % a->tiles_x) * PW_TX0;
    int gbase = 0;  // running group index over the outputs of the unit -> wave assignment
    for (int pi = 0; pi < npass; ++pi) {
      PwPassP ps = &a->pass[pi];
      const int r = ps->r;
      const int Hr = H0 >> r, Wr = W0 >> r;
      const int txl = 5 - r;  // log2(PW_TX0 >> r)
      const int npx = ((1 << tyl) >> r) << txl;
      const int ng = (npx + 63) >> 6;
      const int nrows = ps->nrows;
      const float* wl0 = lds + ps->w_off;
      for (int c = 0; c < ng; ++c) {
        if (((gbase + c) & 3) != wave) continue;
        const int p = (c << 6) + lane;
        const int py_ = (ty0 >> r) + (p >> txl), px_ = (tx0 >> r) + (p & ((1 << txl) - 1));
        const bool valid = p < npx && py_ < Hr && px_ < Wr;
        const int gy = min(py_, Hr - 1), gx = min(px_, Wr - 1);     // lanes off the image gather a valid pixel
        const unsigned ovoff = (unsigned)(gy * Wr + gx) * 4u;        // ... and store nothing (exec-masked)
        for (int row0 = 0; row0 < nrows; row0 += 32) {
          if (nrows - row0 <= 16) pw_sweep<1>(ps, wl0, xb, row0, b, gy, gx, Hr, Wr, ovoff, valid, lane);
          else pw_sweep<2>(ps, wl0, xb, row0, b, gy, gx, Hr, Wr, ovoff, valid, lane);
        }
      }
      gbase += ng;
    }
  }
}

int csn_launch_pw(const PwArgs& a, int maxnt, void* stream) {
  (void)maxnt;
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  const dim3 grid(ntiles < PW_MAX_GRID ? ntiles : PW_MAX_GRID);
  const size_t lds = ((size_t)a.wimg_floats + 4 * PW_KC * PW_XP) * sizeof(float);
#ifndef CSN_CPU_EMU
  static bool attr_done = false;
  if (!attr_done) {
    // units with a large weight image may use the full 160 KiB of LDS of a CDNA4 CU
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&goct_pw_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
#endif
  CSN_LAUNCH(goct_pw_kernel, grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
