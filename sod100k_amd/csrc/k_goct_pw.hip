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
#include "pw_gather.h"

// acc[i] += sum_{u<4} W[row0 + (lane>>4)*4 + i][k0 + u] * x[k0 + u][16 s + (lane&15)]
// wt = &W[row0][k0] in the LDS weight image (row pitch `stride`), xs = &x[k0][16 s] in the wave's panel.
__device__ __forceinline__ void pw_mfma16(const float* wt, int stride, const float* xs, int lane, csn_f4& acc) {
#ifdef CSN_EMU_SEQ
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
  acc = csn_mfma_16x16x4(av, bv, acc);
#endif
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
template <int NT, bool RAW, typename AT>
__device__ __forceinline__ void pw_sweep(PwPassP ps, const float* wl0, float* xb, int row0, int b, int gy, int gx,
                                         int Hr, int Wr, unsigned ovoff, bool valid, int lane, float& red, unsigned pair_el) {
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
    if (kc < c1) pw_gather_slice<AT, PW_XP>(ps, 0, kc, min(kend, c1), xb + lane, PW_KC, b, gy, gx, Hr, Wr, pair_el);
    if (max(kc, c1) < min(kend, c2)) {
      const int r0 = max(kc, c1) - kc;
      pw_gather_slice<AT, PW_XP>(ps, 1, max(kc, c1) - c1, min(kend, c2) - c1, xb + r0 * PW_XP + lane, PW_KC - r0, b, gy, gx, Hr, Wr, pair_el);
    }
    if (max(kc, c2) < min(kend, cin)) {
      const int r0 = max(kc, c2) - kc;
      pw_gather_slice<AT, PW_XP>(ps, 2, max(kc, c2) - c2, min(kend, cin) - c2, xb + r0 * PW_XP + lane, PW_KC - r0, b, gy, gx, Hr, Wr, pair_el);
    }
    for (int k = max(cin, kc); k < kend; ++k) xb[(k - kc) * PW_XP + lane] = 0.f;  // pad to a multiple of 4
    CSN_WAVE_SYNC();  // panel complete
    pw_contract<NT>(wl0 + row0 * stride + kc, stride, xb, kend - kc, lane, acc);
  }
  const unsigned cs4 = (unsigned)(Hr * Wr) * (unsigned)sizeof(AT);
  const csn_buf ob = csn_make_buf(act_cast<AT>(ps->out) + (int64_t)b * ps->out_ctot * (Hr * Wr));
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
    if (!RAW && ps->red_w) {   // fused 1x1 consumer (cls_layer): accumulate w[r] * y_r per pixel, nothing is stored here
      csn_cfp rw = csn_const(ps->red_w) + rbase;
#pragma unroll 4
      for (int rr = 0; rr < rn; ++rr) red = fmaf(rw[rr], csn_epi(xb[rr * PW_EP + lane], scale[rr], shift[rr], alpha[rr]), red);
      continue;
    }
#pragma unroll 4
    for (int rr = 0; rr < rn; ++rr) {
      const float val = RAW ? xb[rr * PW_EP + lane] : csn_epi(xb[rr * PW_EP + lane], scale[rr], shift[rr], alpha[rr]);
      if (valid) csn_bufacc<AT>::st1(ob, ovoff, (unsigned)(rbase + rr) * cs4, val);
    }
  }
}

// RAW = true: plain store (train-mode raw convolution outputs, every backward-data launch) -- its own symbol, so
// kernel traces keep the eval-mode launches (BN + PReLU epilogue) apart from the training ones.
template <bool RAW, typename AT>
__global__ __launch_bounds__(CSN_BLOCK, 4) void goct_pw_kernel(PwArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS PwArgs* a = CSN_KERNARG(PwArgs, a_byval);
  const int tid = threadIdx.x;
  // ---- stage the unit's weight image (all passes, rows padded to 16, zero filled) into LDS once ----
  csn_fill_lds16(lds, a->wimg, a->wimg_floats >> 2, tid);
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  float* xb = lds + a->wimg_floats + wave * (PW_KC * PW_XP);   // this wave's x[k][64 px] panel
  const int H0 = a->H0, W0 = a->W0, npass = a->npass;
  const int tiles_xy = a->tiles_x * a->tiles_y;
  const int ntiles = tiles_xy * a->B;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blockIdx.x & 7), each with its own L2.
  // XCD x walks the contiguous tile range [x*chunk, (x+1)*chunk) -- neighbouring tiles (shared halo rows, the
  // low-resolution sources of the bilinear / pooled slices) then hit the same L2 instead of 8 different ones.
  const int nslot = gridDim.x >> 3;                 // gridDim.x is a multiple of 8
  const int chunk = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min((xcd + 1) * chunk, ntiles);
  for (int tile = xcd * chunk + (blockIdx.x >> 3); tile < tend; tile += nslot) {
    const int b = tile / tiles_xy;
    const int txy = tile - b * tiles_xy;
    const int tyl = a->ty_log2;   // tile height 16 / 8 / 4 rows of branch 0 (small maps get more, smaller tiles)
    const int ty0 = (txy / a->tiles_x) << tyl, tx0 = (txy % a->tiles_x) * PW_TX0;
    int gbase = 0;  // running group index over the outputs of the unit -> wave assignment
    for (int pi = 0; pi < npass; ++pi) {
      PwPassP ps = &a->pass[pi];
      const int r = ps->r;
      const int Hr = H0 >> r, Wr = W0 >> r;
      const int txl = PW_TXL - r;  // log2(PW_TX0 >> r)
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
        const unsigned ovoff = (unsigned)(gy * Wr + gx) * (unsigned)sizeof(AT);        // ... and store nothing (exec-masked)
        // pixel pair (2q, 2q + 1) of the group, q = lane & 31: neighbours in a row (tile widths are even), clamped into the
        // plane like the lane's own pixel -- own-resolution slices are fetched two pixels per lane (pw_batch_own_pair)
        unsigned pair_el = CSN_NO_PAIR;
        if ((Wr & 1) == 0) {
          const int p2 = (c << 6) + 2 * (lane & 31);
          const int qy = min((ty0 >> r) + (p2 >> txl), Hr - 1), qx = min((tx0 >> r) + (p2 & ((1 << txl) - 1)), Wr - 2);
          pair_el = (unsigned)(qy * Wr + qx);
        }
        float red = 0.f;
        for (int row0 = 0; row0 < nrows; row0 += 32) {
          if (nrows - row0 <= 16) pw_sweep<1, RAW, AT>(ps, wl0, xb, row0, b, gy, gx, Hr, Wr, ovoff, valid, lane, red, pair_el);
          else pw_sweep<2, RAW, AT>(ps, wl0, xb, row0, b, gy, gx, Hr, Wr, ovoff, valid, lane, red, pair_el);
        }
        if (!RAW && ps->red_w && valid) {   // out: [B][1][Hr][Wr]
          const csn_buf ob = csn_make_buf(act_cast<AT>(ps->out) + (int64_t)b * (Hr * Wr));
          csn_bufacc<AT>::st1(ob, ovoff, 0, red + csn_const(ps->red_b)[0]);
        }
      }
      gbase += ng;
    }
  }
}

int csn_launch_pw(const PwArgs& a, int raw, void* stream) {
  const int ntiles = a.tiles_x * a.tiles_y * a.B;
  const int nblk = ntiles < PW_MAX_GRID ? ntiles : PW_MAX_GRID;
  const dim3 grid((nblk + 7) & ~7);   // multiple of 8: see the XCD-aware tile order in the kernel
  const size_t lds = ((size_t)a.wimg_floats + 4 * PW_KC * PW_XP) * sizeof(float);
#ifndef CSN_CPU_EMU
  static CsnPerDeviceOnce attr_once;
  const int ast = attr_once.run([&]() {
    // units with a large weight image may use the full 160 KiB of LDS of a CDNA4 CU
    const void* fns[4] = {reinterpret_cast<const void*>(&goct_pw_kernel<false, float>),
                          reinterpret_cast<const void*>(&goct_pw_kernel<true, float>),
                          reinterpret_cast<const void*>(&goct_pw_kernel<false, csn_bf16>),
                          reinterpret_cast<const void*>(&goct_pw_kernel<true, csn_bf16>)};
    for (const void* f : fns) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  });
  if (ast != 0) return ast;
#endif
  if (a.a16) {
    if (raw) CSN_LAUNCH((goct_pw_kernel<true, csn_bf16>), grid, dim3(CSN_BLOCK), lds, stream, a);
    else CSN_LAUNCH((goct_pw_kernel<false, csn_bf16>), grid, dim3(CSN_BLOCK), lds, stream, a);
  } else {
    if (raw) CSN_LAUNCH((goct_pw_kernel<true, float>), grid, dim3(CSN_BLOCK), lds, stream, a);
    else CSN_LAUNCH((goct_pw_kernel<false, float>), grid, dim3(CSN_BLOCK), lds, stream, a);
  }
  return (int)hipGetLastError();
}
