// k_ilb.hip -- a whole 1x1 ILBlock in ONE pass over HBM (eval mode).
//
// Reference semantics (CSNet/model/csnet.py):
//   ILBlock.forward 72-76:            conv1x1 (gOctaveCBR, 1x1) -> conv3x3_1 -> conv3x3_2 (SimplifiedGOctConvBR)
//   gOctaveConv.forward 664-726:      y_j = sum_i T_ij(x_i);  i == j conv;  i > j bilinear_up(conv(x_i));  i < j conv(max_pool(x_i))
//   gOctaveCBR.forward 778-792:       PReLU_j(BN_j(y_j))              (eval BN folded to scale / shift)
//   SimplifiedGOctConvBR.forward 838-851: depthwise 3x3 (padding 1, weight x100, conv2d.py:104) -> BN -> PReLU, per branch
//
// Unit-level kernels move (in + out) of the 1x1 unit and (in + out) of the depthwise pair through HBM (78 floats per
// full-resolution pixel for stage1.1); fused, the block's inputs are read once and its outputs written once (36).
//
// MI355X mapping: ONE WAVE = ONE COLUMN STRIP of one output branch, marching down the rows.  Lane l owns column
// c0 - 4 + l (56 output columns + a 4-column halo each side: the two 3x3 stages need y two columns out), and ALL
// channels of its pixel live in registers -- there is no LDS, no block barrier and no transposition anywhere:
//   stage A  1x1 contraction as a VALU FMA stream: the weights are wave-uniform, so they arrive through the scalar
//            cache (s_load_dwordx16 of a transposed, zero-padded row per input channel) and feed v_fmac as SGPR
//            operands; x is one coalesced dword per lane and channel.  fp32 MFMA runs at the same rate as the vector
//            ALU on CDNA4 and would pad 13..23 input channels to multiples of 4 and 12..18 rows to 16/32.
//            low -> high term in the reference's own order: conv at the LOW resolution (each lane contracts the low
//            pixel under its column), then the 2x bilinear interpolation -- horizontally with two DPP wave shifts
//            (the neighbour low pixel sits in the neighbour lane), vertically between the two low rows held in
//            registers (a new low row enters every second output row);
//            high -> low term: 2x2 max-pool of four lane-local values (two 64-bit loads);
//   stage B/C the depthwise 3x3 convolutions on a rolling three-row register window; the horizontal taps are two DPP
//            shifts of column sums:  out = sum_dy w[dy][0] y[dy]  (shifted right)  +  sum_dy w[dy][1] y[dy]
//            +  sum_dy w[dy][2] y[dy]  (shifted left), 11 VALU per output and channel, weights again SGPRs.
// Rows / columns outside the image are exact zeros in y and in the first depthwise output (the next convolution pads
// its INPUT, csnet.py:815-824 padding=1).  Output channels are processed in groups of at most NC (compile time) per
// wave; a group re-reads the block's input strip (L2).  Work items (branch, image, row segment, strip, channel group) are
// dealt to waves so that the waves sharing an input region run on the same XCD.
#include "csn_kernels.h"

#ifdef CSN_CPU_EMU
// lanes of a wave are fibers: exchange through a per-block table between two wave barriers
static inline float ilb_shr1(float v) {
  float* x = csn_emu::g.xch;
  x[threadIdx.x] = v;
  __syncthreads();
  const float r = (threadIdx.x & 63) ? x[threadIdx.x - 1] : 0.f;
  __syncthreads();
  return r;
}
static inline float ilb_shl1(float v) {
  float* x = csn_emu::g.xch;
  x[threadIdx.x] = v;
  __syncthreads();
  const float r = (threadIdx.x & 63) != 63 ? x[threadIdx.x + 1] : 0.f;
  __syncthreads();
  return r;
}
#else
// DPP wave_shr:1 / wave_shl:1: lane l receives lane l-1 / l+1 of the whole 64-lane wave, zero at the ends
__device__ __forceinline__ float ilb_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float ilb_shl1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
#endif

typedef const CSN_CONST_AS IlbRole* IlbRoleP;

// acc[c] += sum_k wt[k][c] * g_k for the K gathered channels of one source.  POOL: g_k = max of the 2x2 window whose
// top-left element is at byte offset voff of a tensor with row pitch pitch4 (two 64-bit loads), else one dword at voff.
// Channel indices are clamped (a predicated load would be waited for at the join); the weight rows past K are zero.
template <int NC, bool POOL>
__device__ __forceinline__ void ilb_contract(float (&acc)[NC], csn_buf rb, unsigned voff, unsigned pitch4, unsigned cs4,
                                             int K, csn_cfp wt) {
  const unsigned last = (unsigned)(K - 1) * cs4;
  for (int k0 = 0; k0 < K; k0 += 8) {
    float x[8];
    unsigned so = (unsigned)k0 * cs4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (POOL) {
        const float2 a = csn_ld2(rb, voff, so), b = csn_ld2(rb, voff + pitch4, so);
        x[j] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
      } else {
        x[j] = csn_ld1(rb, voff, so);
      }
      so = min(so + cs4, last);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      csn_cfp w = wt + (k0 + j) * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = fmaf(w[c], x[j], acc[c]);
    }
  }
}

// One depthwise 3x3 + BN + PReLU row from the three input rows (top, mid, bot), all channels of the group.
// rec + 32 c + wofs: 9 taps (x100 folded), scale, shift, alpha.  maskf zeroes the result outside the image columns.
template <int NC, bool MASK>
__device__ __forceinline__ void ilb_dw_row(const float (&top)[NC], const float (&mid)[NC], const float (&bot)[NC],
                                           float (&out)[NC], csn_cfp rec, int wofs, float maskf) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    csn_cfp w = rec + 32 * c + wofs;
    float sl = w[0] * top[c];
    sl = fmaf(w[3], mid[c], sl);
    sl = fmaf(w[6], bot[c], sl);
    float sm = w[1] * top[c];
    sm = fmaf(w[4], mid[c], sm);
    sm = fmaf(w[7], bot[c], sm);
    float sr = w[2] * top[c];
    sr = fmaf(w[5], mid[c], sr);
    sr = fmaf(w[8], bot[c], sr);
    const float t = (sm + ilb_shr1(sl)) + ilb_shl1(sr);   // tap dx = -1 reads column x - 1: the sum held by lane l - 1
    const float y = csn_epi(t, w[9], MASK ? w[10] * maskf : w[10], w[11]);
    out[c] = MASK ? y * maskf : y;
  }
}

template <int NC>
__device__ __forceinline__ void ilb_zero(float (&a)[NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) a[c] = 0.f;
}

// Per-wave state of the low -> high path (ROLE 0): the low-resolution contraction, horizontally interpolated, at the two
// low rows the current output row lies between.
template <int NC>
struct IlbUp {
  float tp[NC], tc[NC];
  int m_cur;
};

template <int NC, int ROLE, bool POOL>
__device__ __forceinline__ void ilb_wave(IlbRoleP R, int item, int lane) {
  const int ng = R->ngroups, ns = R->strips, nt = R->segs;
  const int g = item % ng;
  int q = item / ng;
  const int s = q % ns;
  q /= ns;
  const int t = q % nt;
  const int b = q / nt;
  const int H = R->H, W = R->W;
  const int col = s * ILB_SW - ILB_HALO + lane;
  const bool cin = col >= 0 && col < W;
  const float maskf = cin ? 1.f : 0.f;
  const int r0 = t * R->seg_rows, r1 = min(H, r0 + R->seg_rows);
  const int n = min(R->gsize, R->n_out - g * R->gsize);
  const int K_own = R->C_own, K_oth = R->C_oth;
  csn_cfp wt_own = csn_const(R->wt) + (int64_t)g * R->group_stride;
  csn_cfp wt_oth = wt_own + R->K8own * NC;
  csn_cfp rec = wt_oth + R->K8oth * NC;
  const unsigned HW4 = (unsigned)(H * W) * 4u;
  const csn_buf xo = csn_make_buf_n(R->x_own + (int64_t)b * K_own * (H * W), (unsigned)K_own * HW4);
  const unsigned OOB = 0x80000000u;                        // voffset of lanes outside the image: reads return 0
  const unsigned colv = cin ? (unsigned)col * 4u : OOB;

  // ---- the other input branch ----
  csn_buf xt = xo;
  unsigned oth_cs4 = 0, oth_pitch4 = 0, oth_col = 0;
  int Hl = 1;
  float ha = 1.f, hl = 0.f, hr = 0.f;
  if (ROLE == 0) {   // low tensor [K_oth][H/2][W/2]; this lane's low column and the horizontal lerp weights
    Hl = H >> 1;
    const int Wl = W >> 1;
    oth_cs4 = (unsigned)(Hl * Wl) * 4u;
    oth_pitch4 = (unsigned)Wl * 4u;
    xt = csn_make_buf_n(R->x_oth + (int64_t)b * K_oth * (Hl * Wl), (unsigned)K_oth * oth_cs4);
    const int nl = min(max(col >> 1, 0), Wl - 1);
    oth_col = (unsigned)nl * 4u;
    // x2 bilinear, align_corners=False: out[2k] = .25 in[k-1] + .75 in[k], out[2k+1] = .75 in[k] + .25 in[k+1], clamped
    if (cin && col > 0 && col < W - 1) {
      ha = 0.75f;
      if (col & 1) hr = 0.25f; else hl = 0.25f;
    }
  } else {           // high tensor [K_oth][2H][2W]: 2x2 max-pool windows
    oth_cs4 = HW4 * 4u;
    oth_pitch4 = (unsigned)W * 8u;
    xt = csn_make_buf_n(R->x_oth + (int64_t)b * K_oth * (4 * H * W), (unsigned)K_oth * oth_cs4);
    oth_col = cin ? (unsigned)col * 8u : OOB;
  }

  float ya[NC], yb[NC], yc[NC], ea[NC], eb[NC], ec[NC];
  ilb_zero<NC>(ya); ilb_zero<NC>(yb); ilb_zero<NC>(yc);
  ilb_zero<NC>(ea); ilb_zero<NC>(eb); ilb_zero<NC>(ec);
  IlbUp<NC> up;
  ilb_zero<NC>(up.tp); ilb_zero<NC>(up.tc);
  up.m_cur = 0;

  const bool has_up = ROLE == 0 && K_oth > 0;
  auto advance = [&]() {   // next low row: contract at the low resolution, then interpolate along the row
    ++up.m_cur;
    const int mm = min(max(up.m_cur, 0), Hl - 1);
    float tt[NC];
    ilb_zero<NC>(tt);
    ilb_contract<NC, false>(tt, xt, (unsigned)mm * oth_pitch4 + oth_col, 0u, oth_cs4, K_oth, wt_oth);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      up.tp[c] = up.tc[c];
      float h = ha * tt[c];
      h = fmaf(hl, ilb_shr1(tt[c]), h);
      h = fmaf(hr, ilb_shl1(tt[c]), h);
      up.tc[c] = h;
    }
  };
  const int r_first = r0 - 2;
  if (has_up) {     // rows (m - 1, m) of output row r: m = (r + 1) >> 1
    up.m_cur = ((max(r_first, 0) + 1) >> 1) - 2;
    advance();
    advance();
  }

  // bounded to the group's n channels: lanes outside the strip / image (offset OOB) and channels past n are dropped
  const csn_buf ob = csn_make_buf_n(R->out + ((int64_t)b * R->n_out + g * R->gsize) * (H * W), (unsigned)n * HW4);
  const bool st_lane = lane >= ILB_HALO && lane < ILB_HALO + ILB_SW && cin;
  // 2x2 average of the block's output for the stride-2 unit that follows (csnet.py:679-680): even rows park the sum of
  // their column pair, odd rows complete it -- ((a00 + a01) + a10) + a11, the order of avgpool2_kernel
  float op[POOL ? NC : 1];
  const bool do_pool = POOL && R->pool != nullptr;
  const csn_buf pb = csn_make_buf_n(do_pool ? R->pool + ((int64_t)b * R->n_out + g * R->gsize) * ((H >> 1) * (W >> 1)) : R->out,
                                    do_pool ? (unsigned)n * (HW4 >> 2) : 0u);
  const bool skip_out = R->skip_out != 0;

  auto step = [&](int r, float (&Yn)[NC], const float (&Ym)[NC], const float (&Yo)[NC], float (&En)[NC],
                  const float (&Em)[NC], const float (&Eo)[NC]) {
    // ---- A: y row r ----
    if (r >= 0 && r < H) {
      if (has_up) {
        if (up.m_cur < ((r + 1) >> 1)) advance();
        const float wa = (r & 1) ? 0.75f : 0.25f, wb = 1.f - wa;
#pragma unroll
        for (int c = 0; c < NC; ++c) Yn[c] = fmaf(wb, up.tc[c], wa * up.tp[c]);
      } else {
        ilb_zero<NC>(Yn);
      }
      ilb_contract<NC, false>(Yn, xo, colv + (unsigned)r * (unsigned)W * 4u, 0u, HW4, K_own, wt_own);
      if (ROLE == 1 && K_oth > 0)
        ilb_contract<NC, true>(Yn, xt, oth_col + (unsigned)(2 * r) * oth_pitch4, oth_pitch4, oth_cs4, K_oth, wt_oth);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        csn_cfp e = rec + 32 * c;
        Yn[c] = csn_epi(Yn[c], e[0], e[1] * maskf, e[2]) * maskf;
      }
    } else {
      ilb_zero<NC>(Yn);
    }
    // ---- B: first depthwise unit, row r - 1 ----
    if (r - 1 >= 0 && r - 1 < H) ilb_dw_row<NC, true>(Yo, Ym, Yn, En, rec, 4, maskf);
    else ilb_zero<NC>(En);
    // ---- C: second depthwise unit, row r - 2 -> HBM ----
    const int ro = r - 2;
    if (ro >= r0 && ro < r1) {
      float o[NC];
      ilb_dw_row<NC, false>(Eo, Em, En, o, rec, 16, 1.f);
      const unsigned vo = st_lane ? (unsigned)(ro * W + col) * 4u : OOB;
      if (!(POOL && skip_out)) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (c < n) csn_st1(ob, vo, (unsigned)c * HW4, o[c]);
      }
      if (POOL && do_pool) {
        if ((ro & 1) == 0) {
#pragma unroll
          for (int c = 0; c < NC; ++c) op[c] = o[c] + ilb_shl1(o[c]);
        } else {
          const unsigned vp = (st_lane && (col & 1) == 0) ? (unsigned)((ro >> 1) * (W >> 1) + (col >> 1)) * 4u : OOB;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float pv = ((op[c] + o[c]) + ilb_shl1(o[c])) * 0.25f;
            if (c < n) csn_st1(pb, vp, (unsigned)c * (HW4 >> 2), pv);
          }
        }
      }
    }
  };

  for (int r = r_first; r < r1 + 2; r += 3) {
    step(r, ya, yc, yb, ea, ec, eb);
    step(r + 1, yb, ya, yc, eb, ea, ec);
    step(r + 2, yc, yb, ya, ec, eb, ea);
  }
}

template <int NC, bool POOL>
__global__ __launch_bounds__(CSN_BLOCK, (NC <= 12 ? 4 : NC <= 16 ? 3 : 2)) void ilb_kernel(IlbArgs a_byval) {
  const CSN_CONST_AS IlbArgs* a = CSN_KERNARG(IlbArgs, a_byval);
  const int lane = threadIdx.x & 63;
#ifdef CSN_CPU_EMU
  const int wave = threadIdx.x >> 6;
#else
  // the wave index is uniform by construction; say so, or every table pointer derived from the work item is treated as
  // divergent (vector loads + readfirstlane waterfall loops instead of s_load)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; XCD x walks a contiguous range of items, so the
  // channel groups / strips / branches that read the same input region share one L2
  const int nitem = a->items;
  const int nslot = gridDim.x >> 3;
  const int chunk = ((nitem + 4 * 8 - 1) / (4 * 8)) * 4;       // items per XCD, whole blocks
  const int xcd = blockIdx.x & 7;
  const int item = xcd * chunk + (blockIdx.x >> 3) * 4 + wave;
  (void)nslot;
  if (item >= min(nitem, (xcd + 1) * chunk)) return;
  // items are ordered image-major; inside an image the role-0 items come first
  const int per_img = a->role[0].items_img + a->role[1].items_img;
  const int b = item / per_img;
  const int w = item - b * per_img;
  if (w < a->role[0].items_img) ilb_wave<NC, 0, POOL>(&a->role[0], b * a->role[0].items_img + w, lane);
  else ilb_wave<NC, 1, POOL>(&a->role[1], b * a->role[1].items_img + (w - a->role[0].items_img), lane);
}

int csn_launch_ilb(const IlbArgs& a, void* stream) {
  const int chunk = ((a.items + 31) / 32) * 4;
  const dim3 grid((unsigned)(chunk / 4) * 8u);
#define ILB_CASE(NCV)                                                                              \
  case NCV:                                                                                        \
    if (a.pool) CSN_LAUNCH((ilb_kernel<NCV, true>), grid, dim3(CSN_BLOCK), 0, stream, a);          \
    else CSN_LAUNCH((ilb_kernel<NCV, false>), grid, dim3(CSN_BLOCK), 0, stream, a);                \
    break;
  switch (a.nc) {
    ILB_CASE(8)
    ILB_CASE(12)
    ILB_CASE(16)
    ILB_CASE(20)
    default: return -1;
  }
#undef ILB_CASE
  return (int)hipGetLastError();
}
