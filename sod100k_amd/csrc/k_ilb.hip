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

// ---- where the (wave-uniform) weights come from ---------------------------------------------------------------------
// ILB_LANEW = 1 (default): the group's tables live in 3 NC VGPRs of the wave, one value per LANE -- register c of the
//   1x1 weights holds W[c][k] in lane k, register c of the records holds {scale, shift, alpha, 9 taps ...} of channel c in
//   lanes 0..31 -- and every use broadcasts its value with v_readlane_b32 into an SGPR operand.  One more instruction per
//   weight, but no memory access in the row loop: a per-row stream of ~100 scalar loads, each waited for with
//   lgkmcnt(0), ran at 5-9 % of the VALU rate (profiles/r2_notes.md).
// ILB_LANEW = 0: weights through the scalar cache (s_load), tables [k][NC] / [NC][32].
template <int NC>
struct IlbW {
#if ILB_LANEW
  float wo[NC], wt[NC], rc[NC];
#endif
  csn_cfp base;   // LANEW: [NC][64] own, [NC][64] other, [NC][64] records;  else: [K8own][NC], [K8oth][NC], [NC][32]
  int oth_off, rec_off;
};
#if ILB_LANEW
#ifdef CSN_CPU_EMU
#define ILB_BCAST(reg, ptr, ln) ((ptr)[ln])      // lanes are fibers: read the table the registers were loaded from
#else
#define ILB_BCAST(reg, ptr, ln) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(reg), (ln)))
#endif
#define ILB_WOWN(w, c, k) ILB_BCAST((w).wo[c], (w).base + (c) * 64, (k))
#define ILB_WOTH(w, c, k) ILB_BCAST((w).wt[c], (w).base + (w).oth_off + (c) * 64, (k))
#define ILB_REC(w, c, v) ILB_BCAST((w).rc[c], (w).base + (w).rec_off + (c) * 64, (v))
#elif defined(ILB_KNOCK_SMEM)   // measurement build: every table access hits the same few cache lines (results are wrong)
#define ILB_WOWN(w, c, k) ((w).base[(c)])
#define ILB_WOTH(w, c, k) ((w).base[(c)])
#define ILB_REC(w, c, v) ((w).base[(v)])
#else
#define ILB_WOWN(w, c, k) ((w).base[(k) * NC + (c)])
#define ILB_WOTH(w, c, k) ((w).base[(w).oth_off + (k) * NC + (c)])
#define ILB_REC(w, c, v) ((w).base[(w).rec_off + 32 * (c) + (v)])
#endif

// acc[c] += sum_k W[c][k] * g_k for the K gathered channels of one source (OTH: the other input branch's weight block).
// POOL: g_k = max of the 2x2 window whose top-left element is at byte offset voff of a tensor with row pitch pitch4 (two
// 64-bit loads), else one dword at voff.  Channel indices are clamped (a predicated load would be waited for at the join);
// the weights of the channels past K are zero.
template <int NC, bool POOL, bool OTH>
__device__ __forceinline__ void ilb_contract(float (&acc)[NC], csn_buf rb, unsigned voff, unsigned pitch4, unsigned cs4,
                                             int K, const IlbW<NC>& W) {
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
#ifndef ILB_KNOCK_VMEM
      so = min(so + cs4, last);
#endif
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        acc[c] = fmaf(OTH ? ILB_WOTH(W, c, k0 + j) : ILB_WOWN(W, c, k0 + j), x[j], acc[c]);
    }
  }
}

// One depthwise 3x3 + BN + PReLU row from the three input rows (top, mid, bot), all channels of the group.
// Record values wofs .. wofs + 11 of a channel: 9 taps (x100 folded), scale, shift, alpha.  maskf zeroes the result
// outside the image columns.
template <int NC, bool MASK>
__device__ __forceinline__ void ilb_dw_row(const float (&top)[NC], const float (&mid)[NC], const float (&bot)[NC],
                                           float (&out)[NC], const IlbW<NC>& W, int wofs, float maskf) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float sl = ILB_REC(W, c, wofs + 0) * top[c];
    sl = fmaf(ILB_REC(W, c, wofs + 3), mid[c], sl);
    sl = fmaf(ILB_REC(W, c, wofs + 6), bot[c], sl);
    float sm = ILB_REC(W, c, wofs + 1) * top[c];
    sm = fmaf(ILB_REC(W, c, wofs + 4), mid[c], sm);
    sm = fmaf(ILB_REC(W, c, wofs + 7), bot[c], sm);
    float sr = ILB_REC(W, c, wofs + 2) * top[c];
    sr = fmaf(ILB_REC(W, c, wofs + 5), mid[c], sr);
    sr = fmaf(ILB_REC(W, c, wofs + 8), bot[c], sr);
    const float t = (sm + ilb_shr1(sl)) + ilb_shl1(sr);   // tap dx = -1 reads column x - 1: the sum held by lane l - 1
    const float sh = ILB_REC(W, c, wofs + 10);
    const float y = csn_epi(t, ILB_REC(W, c, wofs + 9), MASK ? sh * maskf : sh, ILB_REC(W, c, wofs + 11));
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
  IlbW<NC> WG;
  WG.base = csn_const(R->wt) + (int64_t)g * R->group_stride;
#if ILB_LANEW
  WG.oth_off = NC * 64;
  WG.rec_off = 2 * NC * 64;
  {
    const float* tb = R->wt + (int64_t)g * R->group_stride + lane;   // one coalesced 256-byte row per register
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      WG.wo[c] = tb[c * 64];
      WG.wt[c] = tb[(NC + c) * 64];
      WG.rc[c] = tb[(2 * NC + c) * 64];
    }
  }
#else
  WG.oth_off = R->K8own * NC;
  WG.rec_off = (R->K8own + R->K8oth) * NC;
#endif
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
    ilb_contract<NC, false, true>(tt, xt, (unsigned)mm * oth_pitch4 + oth_col, 0u, oth_cs4, K_oth, WG);
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
      ilb_contract<NC, false, false>(Yn, xo, colv + (unsigned)r * (unsigned)W * 4u, 0u, HW4, K_own, WG);
      if (ROLE == 1 && K_oth > 0)
        ilb_contract<NC, true, true>(Yn, xt, oth_col + (unsigned)(2 * r) * oth_pitch4, oth_pitch4, oth_cs4, K_oth, WG);
#pragma unroll
      for (int c = 0; c < NC; ++c)
        Yn[c] = csn_epi(Yn[c], ILB_REC(WG, c, 0), ILB_REC(WG, c, 1) * maskf, ILB_REC(WG, c, 2)) * maskf;
    } else {
      ilb_zero<NC>(Yn);
    }
    // ---- B: first depthwise unit, row r - 1 ----
#ifdef ILB_KNOCK_DW
#pragma unroll
    for (int c = 0; c < NC; ++c) En[c] = Ym[c];
#else
    if (r - 1 >= 0 && r - 1 < H) ilb_dw_row<NC, true>(Yo, Ym, Yn, En, WG, 4, maskf);
    else ilb_zero<NC>(En);
#endif
    // ---- C: second depthwise unit, row r - 2 -> HBM ----
    const int ro = r - 2;
    if (ro >= r0 && ro < r1) {
      float o[NC];
#ifdef ILB_KNOCK_DW
#pragma unroll
      for (int c = 0; c < NC; ++c) o[c] = Em[c];
#else
      ilb_dw_row<NC, false>(Eo, Em, En, o, WG, 16, 1.f);
#endif
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

#ifdef ILB_UNROLL3
  // three copies of the row body with the window registers renamed: no moves, three times the code
  for (int r = r_first; r < r1 + 2; r += 3) {
    step(r, ya, yc, yb, ea, ec, eb);
    step(r + 1, yb, ya, yc, eb, ea, ec);
    step(r + 2, yc, yb, ya, ec, eb, ea);
  }
#else
  // one copy of the row body (it has to stay resident in the instruction cache next to the other waves' copies); the
  // three-row windows are rotated with 4 NC register moves per row
  for (int r = r_first; r < r1 + 2; ++r) {
    step(r, ya, yb, yc, ea, eb, ec);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      yc[c] = yb[c]; yb[c] = ya[c];
      ec[c] = eb[c]; eb[c] = ea[c];
    }
  }
#endif
}

template <int NC, int ROLE, bool POOL>
__global__ __launch_bounds__(CSN_BLOCK, (NC <= 8 ? 4 : NC <= 12 ? 3 : 2)) void ilb_kernel(IlbArgs a_byval) {
  const CSN_CONST_AS IlbArgs* a = CSN_KERNARG(IlbArgs, a_byval);
  const int lane = threadIdx.x & 63;
#ifdef CSN_CPU_EMU
  const int wave = threadIdx.x >> 6;
#else
  // the wave index is uniform by construction; say so, or every table pointer derived from the work item is treated as
  // divergent (vector loads + readfirstlane waterfall loops instead of s_load)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; XCD x walks a contiguous range of items (image-major),
  // so the channel groups / strips / row segments that read the same input region share one L2
  const int nitem = a->items;
  const int chunk = ((nitem + 4 * 8 - 1) / (4 * 8)) * 4;       // items per XCD, whole blocks
  const int xcd = blockIdx.x & 7;
  const int item = xcd * chunk + (blockIdx.x >> 3) * 4 + wave;
  if (item >= min(nitem, (xcd + 1) * chunk)) return;
  ilb_wave<NC, ROLE, POOL>(&a->role[ROLE], item, lane);
}

// One launch per output branch (the two branches have different geometry and often different group widths; separate
// kernels also keep each wave's instruction footprint to its own role).
int csn_launch_ilb(const IlbArgs& a0, void* stream) {
  for (int role = 0; role < 2; ++role) {
    if (a0.role[role].items_img <= 0) continue;
    IlbArgs a = a0;
    a.items = a0.B * a0.role[role].items_img;
    a.nc = a0.role[role].nc;
    const int chunk = ((a.items + 31) / 32) * 4;
    const dim3 grid((unsigned)(chunk / 4) * 8u);
#define ILB_LAUNCH(NCV, ROLEV)                                                                           \
  do {                                                                                                   \
    if (a.pool) CSN_LAUNCH((ilb_kernel<NCV, ROLEV, true>), grid, dim3(CSN_BLOCK), 0, stream, a);         \
    else CSN_LAUNCH((ilb_kernel<NCV, ROLEV, false>), grid, dim3(CSN_BLOCK), 0, stream, a);               \
  } while (0)
#define ILB_CASE(NCV)                                       \
  case NCV:                                                 \
    if (role == 0) ILB_LAUNCH(NCV, 0); else ILB_LAUNCH(NCV, 1); \
    break;
    switch (a.nc) {
      ILB_CASE(8)
      ILB_CASE(12)
      ILB_CASE(16)
      ILB_CASE(20)
      default: return -1;
    }
#undef ILB_CASE
#undef ILB_LAUNCH
  }
  return (int)hipGetLastError();
}
