// k_ilb.hip -- one whole ILBlock (conv1x1 -> conv3x3_1 -> conv3x3_2, CSNet/model/csnet.py:17-76) in ONE launch, for the small maps
// (stages 3-4 of csnet-L-x2 at 224 x 224: 56^2 / 28^2 / 14^2 planes).  Round 5.
//
// Reference semantics: ILBlock.forward (csnet.py:72-76) = gOctaveCBR with 1x1 kernels (664-726, 778-792: y_h = W_hh x_h +
// bilinear_up2(W_hl x_l), y_l = W_ll x_l + W_lh max_pool2(x_h), then per-branch BN + PReLU) followed by two
// SimplifiedGOctConvBR (795-851: depthwise 3x3, padding 1, x100 weights (conv2d.py:104), BN, PReLU per branch).
//
// Why: on these maps the three units were two launches of 13-23 us each whatever their size (profiles/r4_unit_table.md: 20 units =
// 40 launches = 0.46 ms for 11 % of the forward's bytes) -- latency chains, not bandwidth.  Round 2's whole-ILBlock kernel (lane =
// pixel with every channel in registers) lost because the depthwise stages need wave-uniform weights per CHANNEL; its conclusion
// was "fusion has to keep the per-channel-block structure of the depthwise pair, i.e. transposition through LDS".  This is that:
//   * work item = (image, group of 4 NTH high + 4 NTL low OUTPUT channels).  The rows of a 1x1 convolution are independent, and
//     the depthwise units are per channel, so an item needs ALL input channels of its image (small, L2 / Infinity Cache resident)
//     but no halo and no recomputation: the group's whole planes live in LDS from the contraction to the last store.
//   * phase 1 (contraction, k_pw4.hip's scheme): lane = one low pixel + its 2x2 high quad, v_mfma_f32_4x4x1 with the loaded value
//     as the B operand.  The low -> high term is evaluated in the REFERENCE's order, conv at low resolution then interpolate
//     (csnet.py:702-707): z = W_hl x_l costs NTH matrix instructions per low channel instead of nine loads and a dozen vector
//     instructions per low INPUT channel and item; z goes through LDS and its bilinear x2 is added per OUTPUT channel.
//   * BN + PReLU on the accumulators, rows of four channels to their LDS planes [channel][H + 2][W + 4] (zero frame = the
//     depthwise convolutions' padding; the four zero columns in front of a row double as the right frame of the row above).
//   * phases 2, 3 (depthwise pair, k_misc.hip dw3x3x2's arithmetic): a lane owns a strip of four columns x R rows of ONE channel,
//     rolling three-row window read from LDS (one 128-bit + two 32-bit reads per row), conv3x3_1 -> second LDS plane, conv3x3_2
//     -> HBM (+ the 2x2 averages / their 2x2 maxima for a stride-2 unit that follows, as dw3x3x2 delivers them).
// HBM traffic = block inputs (once per group, from L2) + block outputs: the two intermediate tensors never leave the CU.
#include "pw4_common.h"
#include "dw_core.h"

// tools/probes/ilb_bench.hip (-DILB_TIMING): wall-clock stamps (100 MHz) of block phases, thread 0 of every block
#ifdef ILB_TIMING
__device__ unsigned long long* g_ilb_stamps = nullptr;
#define ILB_STAMP(i) do { if (threadIdx.x == 0 && g_ilb_stamps) g_ilb_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ILB_STAMP(i)
#endif

namespace {

// contract `n` (<= LB) low-branch channels: the centre value -> z rows (the high rows of the weight image, at LOW resolution)
// and -> low rows
template <int NTH, int NTL, int LB, int P, bool GUARD>
__device__ __forceinline__ void ilb_lo_batch(const float (&v)[LB], const float* wk, int n, csn_f4 (&accz)[NTH], csn_f4 (&accl)[NTL > 0 ? NTL : 1]) {
  constexpr int NT4 = (NTH + NTL + 3) & ~3;
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    if (GUARD && j >= n) break;
    Pw4A<NT4> a;
    pw4_load_a<NT4, P>(wk + j * 4 * P, a);
#pragma unroll
    for (int t = 0; t < NTH; ++t) pw4_mfma<NT4>(a, t, v[j], accz[t]);
#pragma unroll
    for (int t = 0; t < NTL; ++t) pw4_mfma<NT4>(a, NTH + t, v[j], accl[t]);
  }
}

template <int LB>
__device__ __forceinline__ void ilb_load_lo(csn_buf rb, unsigned o, unsigned cs, int k0, int C, float (&v)[LB]) {
#pragma unroll
  for (int j = 0; j < LB; ++j) v[j] = csn_ld1(rb, o, (unsigned)min(k0 + j, C - 1) * cs);
}

__device__ __forceinline__ unsigned ilb_div(unsigned n, unsigned m) {   // n / d for n * d < 2^32, m = ceil(2^32 / d) or 0 (d = 1)
#ifdef CSN_CPU_EMU
  return m ? (unsigned)(((unsigned long long)n * m) >> 32) : n;
#else
  return m ? __umulhi(n, m) : n;
#endif
}

// one depthwise unit over the group's planes of one branch: task = (channel, row chunk, strip of four columns) of `nch` channels.
// LAST = false: LDS plane `src` -> LDS plane `dst`;  LAST = true: -> HBM (+ pooled copies).  dw_core.h's row core.
template <bool LAST>
__device__ __forceinline__ void ilb_dw_rows(const DwPar& par, const float* base, float* drow, int pitch, int y0, int H, int W, int R, int x0,
                                            float* __restrict__ g, float* __restrict__ pool, float* __restrict__ pmax, int skip_out) {
  DwRow2 top = dw_row2_lds4(base + y0 * pitch), mid = dw_row2_lds4(base + (y0 + 1) * pitch);
  const bool full = x0 + 4 <= W;
  float mx = 0.f, e0 = 0.f, e1 = 0.f;
  for (int q = 0; q < R; ++q) {
    const int y = y0 + q;
    if (y >= H) break;
    const DwRow2 bot = dw_row2_lds4(base + (y + 2) * pitch);
    csn_v2 o01, o23;
    dw_conv4(par, top, mid, bot, o01, o23);
    o01 = dw_prelu2(o01, par.al, par.lim);
    o23 = dw_prelu2(o23, par.al, par.lim);
    const float o[4] = {o01[0], o01[1], o23[0], o23[1]};
    if (!LAST) {
      float* d = drow + (y + 1) * pitch;
      if (full) {
        *reinterpret_cast<float4*>(__builtin_assume_aligned(d, 16)) = make_float4(o[0], o[1], o[2], o[3]);
      } else {   // the columns past W stay zero: they are the second unit's padding
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x0 + j < W) d[j] = o[j];
      }
    } else {
      if (pool) {   // avg_pool2d(2, 2) of the block's output (csnet.py:679-680 of the stride-2 unit that follows) and the 2x2
                    // maximum of those averages, in dw3x3x2_bn_prelu_kernel's summation order (W % 4 == 0, R % 2 / % 4 == 0)
        if ((q & 1) == 0) {
          e0 = o[0] + o[1];
          e1 = o[2] + o[3];
        } else {
          float2 pv;
          pv.x = (e0 + o[0] + o[1]) * 0.25f;
          pv.y = (e1 + o[2] + o[3]) * 0.25f;
          *reinterpret_cast<float2*>(pool + (y >> 1) * (W >> 1) + (x0 >> 1)) = pv;
          if (pmax) {
            if ((q & 3) == 1) mx = fmaxf(pv.x, pv.y);
            else pmax[(y >> 2) * (W >> 2) + (x0 >> 2)] = fmaxf(mx, fmaxf(pv.x, pv.y));
          }
        }
      }
      if (!skip_out) {
        float* gp = g + y * W + x0;
        if (full && (W & 3) == 0) {
          *reinterpret_cast<float4*>(gp) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (x0 + j < W) gp[j] = o[j];
        }
      }
    }
    top = mid;
    mid = bot;
  }
}

template <bool LAST>
__device__ __forceinline__ void ilb_dw_branch(int task, const float* src, float* dst, int plane, int pitch, int H, int W, int R, int nstrip,
                                              int nrc, unsigned m_sr, unsigned m_s, const float* rec, float* __restrict__ out,
                                              float* __restrict__ pool, float* __restrict__ pmax, int skip_out) {
  // (the divisors are launch constants: the host passes ceil(2^32 / d), 0 for d = 1 -- one multiply-high per division)
  const int c = (int)ilb_div((unsigned)task, m_sr), rem = task - c * (nstrip * nrc);
  const int rc = (int)ilb_div((unsigned)rem, m_s), s = rem - rc * nstrip;
  const int x0 = 4 * s, y0 = rc * R;
  // the channel's record in LDS: conv3x3_1's {w'[9], shift, alpha, .}, then conv3x3_2's (IlbArgs::dwrec_*, CSN_PREP_DWREC)
  const DwPar par = dw_par_load(rec + c * (2 * DWREC_FLOATS) + (LAST ? DWREC_FLOATS : 0));
  const float* base = src + c * plane + 3 + x0;   // column x0 - 1 of plane row 0 (= image row -1)
  float* drow = LAST ? nullptr : dst + c * plane + 4 + x0;
  float* g = LAST ? out + c * H * W : nullptr;
  float* pl = (LAST && pool) ? pool + c * (H >> 1) * (W >> 1) : nullptr;
  float* pm = (LAST && pmax) ? pmax + c * (H >> 2) * (W >> 2) : nullptr;
  ilb_dw_rows<LAST>(par, base, drow, pitch, y0, H, W, R, x0, g, pl, pm, skip_out);
}

}  // namespace

// K3 (round 5): the stride-2 ENTRY block of a stage -- gOctaveCBR with 3x3 kernels on ONE input branch (csnet.py:33-40, 679-680,
// 708-717): y_h = conv3x3(x), y_l = conv3x3(max_pool2(x)), x = the 2x2 average of the block's input, delivered with its 2x2 maximum
// by the depthwise pair in front (xh / xl of the argument block, the SAME CH channels at two resolutions).  The contraction is
// k_c3q.hip's: per channel the 4x4 window around the lane's quad (zero padding = out-of-range offsets of a bounded resource) and the
// 3x3 window around its low pixel, nine taps x (4 NTH + NTL) matrix instructions; no low -> high term.  Everything behind the
// contraction (planes in LDS, depthwise pair) is the 1x1 form's.
template <int NTH, int NTL, bool K3 = false>
__global__ __launch_bounds__(1024) void ilb_kernel(IlbArgs a_byval) {
  constexpr int NT4 = (NTH + NTL + 3) & ~3, P = PW4_PITCH(NT4);
  constexpr int HB = NTH == 1 ? 8 : 4, LB = NTH == 1 ? 16 : 8;   // channels per load batch (two batches in flight)
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS IlbArgs* a = CSN_KERNARG(IlbArgs, a_byval);
  const int tid = threadIdx.x, nthr = blockDim.x;
  // items in XCD-aware order: XCD x = blockIdx.x & 7 takes the images [x * ipx, (x + 1) * ipx) with all their groups, so that the
  // groups of an image (which read the same inputs) share one L2
  const int ng = a->ng;
  const int ipx = (a->B + 7) >> 3;
  const int xcd = blockIdx.x & 7, idx = (int)(blockIdx.x >> 3);
  const int b = xcd * ipx + idx / ng, g = idx % ng;
  if (idx >= ipx * ng || b >= a->B) return;
  const int CH = a->CH, CL = a->CL, OH = a->OH, OL = a->OL, Hl = a->Hl, Wl = a->Wl, Hh = 2 * Hl, Wh = 2 * Wl;
  const int HWl = Hl * Wl;
  const int ph = a->ph, pl = a->pl, plane_h = a->plane_h, plane_l = a->plane_l;
  float* H1 = lds + a->off_h1;
  float* H2 = lds + a->off_h2;
  float* L1 = lds + a->off_l1;
  float* L2 = lds + a->off_l2;
  float* Z = lds + a->off_z;
  const int r0h = 4 * NTH * g, r0l = 4 * NTL * g;                       // first high / low output channel of the group
  const int nch_h = max(0, min(4 * NTH, OH - r0h)), nch_l = NTL > 0 ? max(0, min(4 * NTL, OL - r0l)) : 0;

  // ---- lane geometry of the contraction.  Wave w owns the 64 consecutive low pixels [64 w, 64 w + 64) of the plane (flat tiles,
  // k_pw4.hip); the launcher provides a wave per tile, so the accumulators survive the barrier the z exchange needs ----
  ILB_STAMP(0);
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  // K3: halo-lane tiles (k_c3q.hip C3qWin): 62 pixels per wave in lanes 1 .. 62, lanes 0 / 63 hold the pixel in front of / behind them
  // and only load -- every window's edge columns are the neighbouring lanes' centre values, no gathers for them
  constexpr int TP = K3 ? 62 : 64;
  const int pix = K3 ? wave * 62 + lane - 1 : wave * 64 + lane;
  const bool tile_on = wave * TP < HWl;
  const bool valid = pix < HWl && (!K3 || (lane >= 1 && lane <= 62));
  const int pc = min(max(pix, 0), HWl - 1);
  const int y = pc / Wl, x = pc - y * Wl;
  const unsigned csl = (unsigned)HWl * 4u, csh = csl * 4u;
  const unsigned oh0 = (unsigned)((2 * y) * Wh + 2 * x) * 4u, oh1 = oh0 + (unsigned)Wh * 4u, olc = (unsigned)pc * 4u;
  const csn_buf rbh = csn_make_buf_n(reinterpret_cast<const char*>(a->xh) + (int64_t)b * CH * (int64_t)csh, (unsigned)CH * csh);
  const csn_buf rbl = csn_make_buf_n(reinterpret_cast<const char*>(a->xl) + (int64_t)b * CL * (int64_t)csl, (unsigned)CL * csl);
  // the first two batches of activation loads are issued BEFORE the weights are staged: an item is one latency chain
  // (weights -> loads -> contraction -> planes -> depthwise pair -> stores), every round trip taken off it counts
  float2 hA[HB][2], hB[HB][2];
  float lA[LB], lB[LB];
  const int nfh = (CH - 1) / HB;   // full batches in front of the last one
  if (tile_on && !K3) {
    pw4_load_hi<HB>(rbh, oh0, oh1, csh, 0, CH, hA);
    if (nfh >= 1) pw4_load_hi<HB>(rbh, oh0, oh1, csh, HB, CH, hB);
  }
  PW4_FENCE();

  // ---- phase 0: the group's weight image, epilogue records and depthwise records -> LDS; zero frame (whole planes: the interiors
  // are overwritten below) ----
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a->wimg + (int64_t)g * a->gimg_floats);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < (a->gimg_floats >> 2); i += nthr) dst[i] = src[i];
    // records of the group's channels: [4 NTH] x {scale, shift, alpha, 0} (conv1x1, high), [4 NTL] x the same (low),
    // [4 NTH] x 24 depthwise (high), [4 NTL] x 24 (low).  The tables are padded to whole groups.
    float4* par = reinterpret_cast<float4*>(lds + a->off_par);
    const int n_eh = 4 * NTH, n_el = 4 * NTL, n_dh = 24 * NTH, n_dl = 24 * NTL;
    for (int i = tid; i < n_eh + n_el + n_dh + n_dl; i += nthr) {
      const float4* q;
      if (i < n_eh) q = reinterpret_cast<const float4*>(a->ep_h + 4 * r0h) + i;
      else if (i < n_eh + n_el) q = reinterpret_cast<const float4*>(a->ep_l + 4 * r0l) + (i - n_eh);
      else if (i < n_eh + n_el + n_dh) q = reinterpret_cast<const float4*>(a->dwrec_h + 24 * r0h) + (i - n_eh - n_el);
      else q = reinterpret_cast<const float4*>(a->dwrec_l + 24 * r0l) + (i - n_eh - n_el - n_dh);
      par[i] = *q;
    }
    float4* zp = reinterpret_cast<float4*>(lds + a->off_h1);
    const int nz4 = (a->off_z - a->off_h1) >> 2;
    for (int i = tid; i < nz4; i += nthr) zp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  ILB_STAMP(1);
  const float* par_eh = lds + a->off_par;
  const float* par_el = par_eh + 16 * NTH;
  const float* par_dh = par_el + 16 * NTL;
  const float* par_dl = par_dh + 96 * NTH;

  // ---- phase 1: contraction ----
  csn_f4 acch[4][NTH], accz[NTH], accl[NTL > 0 ? NTL : 1];
#pragma unroll
  for (int t = 0; t < NTH; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      accz[t][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) acch[s][t][i] = 0.f;
    }
  }
#pragma unroll
  for (int t = 0; t < (NTL > 0 ? NTL : 1); ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) accl[t][i] = 0.f;
  if (tile_on && K3) {
#ifdef CSN_EMU_SEQ
    const float* wg = lds;
#else
    const float* wg = lds + (lane & 3) * P;
#endif
    // window geometry: rows 2y - 1 .. 2y + 2 of the high plane at column 2x (64-bit centre pairs), left / right edge columns; the 3x3
    // neighbourhood of (y, x) in the low plane; anything outside a plane gets an out-of-range offset (= the zero padding)
    unsigned rowh[4], ol9[9];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yy = 2 * y - 1 + r;
      rowh[r] = (yy >= 0 && yy < Hh) ? (unsigned)(yy * Wh + 2 * x) * 4u : 0x80000000u;
    }
    const unsigned dl = x > 0 ? 0u - 4u : 0x40000000u, dr = 2 * x + 2 < Wh ? 8u : 0x40000000u;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int yy = y - 1 + r, xx = x - 1 + c;
        ol9[3 * r + c] = (yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) ? (unsigned)(yy * Wl + xx) * 4u : 0x80000000u;
      }
    // the centre pairs of the four high rows and the centre column of the three low rows: what is loaded (7 loads per channel; the
    // gather form took 21) and what stays in flight (11 registers instead of 25)
    struct Win { float2 c[4]; float m[3];
#ifdef CSN_EMU_SEQ
      float l[4], r[4], ml[3], mr[3];   // (the emulator's lanes run one after the other: it loads the edges)
#endif
    };
    const bool has_l = x > 0, has_r = x < Wl - 1;
    auto issue_win = [&](int c, Win& w) {
      const unsigned sh_ = (unsigned)min(c, CH - 1) * csh, sl_ = (unsigned)min(c, CH - 1) * csl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w.c[r] = csn_ld2(rbh, rowh[r], sh_);
#ifdef CSN_EMU_SEQ
        w.l[r] = csn_ld1(rbh, rowh[r] + dl, sh_);
        w.r[r] = csn_ld1(rbh, rowh[r] + dr, sh_);
#endif
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        w.m[r] = csn_ld1(rbl, ol9[3 * r + 1], sl_);
#ifdef CSN_EMU_SEQ
        w.ml[r] = csn_ld1(rbl, ol9[3 * r], sl_);
        w.mr[r] = csn_ld1(rbl, ol9[3 * r + 2], sl_);
#endif
      }
    };
    auto finish_win = [&](const Win& w, float (&v)[16], float (&u)[9]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[4 * r + 1] = w.c[r].x; v[4 * r + 2] = w.c[r].y;
#ifdef CSN_EMU_SEQ
        v[4 * r] = w.l[r]; v[4 * r + 3] = w.r[r];
#else
        const float l = csn_bits_f(csn_from_lane_below(csn_f_bits(w.c[r].y)));
        const float rr = csn_bits_f(csn_from_lane_above(csn_f_bits(w.c[r].x)));
        v[4 * r] = has_l ? l : 0.f; v[4 * r + 3] = has_r ? rr : 0.f;
#endif
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        u[3 * r + 1] = w.m[r];
#ifdef CSN_EMU_SEQ
        u[3 * r] = w.ml[r]; u[3 * r + 2] = w.mr[r];
#else
        const float l = csn_bits_f(csn_from_lane_below(csn_f_bits(w.m[r])));
        const float rr = csn_bits_f(csn_from_lane_above(csn_f_bits(w.m[r])));
        u[3 * r] = has_l ? l : 0.f; u[3 * r + 2] = has_r ? rr : 0.f;
#endif
      }
    };
    auto contract = [&](const float (&v)[16], const float (&u)[9], const float* wk) {
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int ty = t9 / 3, tx = t9 - 3 * ty;
        Pw4A<NT4> aw;
        pw4_load_a<NT4, P>(wk + t9 * 4 * P, aw);
#pragma unroll
        for (int t = 0; t < NTH; ++t)
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) pw4_mfma<NT4>(aw, t, v[4 * ((sq >> 1) + ty) + (sq & 1) + tx], acch[sq][t]);
#pragma unroll
        for (int t = 0; t < NTL; ++t) pw4_mfma<NT4>(aw, NTH + t, u[t9], accl[t]);
      }
    };
    // channel c + 1 is in flight while channel c is completed (lane exchange) and contracted.  (The gather form -- 21 loads per
    // channel and wave against 9 (4 NTH + NTL) matrix instructions -- was bound by the NUMBER of its loads, not their latency: a
    // third register set in flight made it slower, 51 -> 56 us per launch at stage 4.0; profiles/r5_notes.md.)
    Win wA, wB;
    float v[16], u[9];
    issue_win(0, wA);
    PW4_FENCE();
    for (int c = 0; c < CH; c += 2) {
      issue_win(c + 1, wB);
      PW4_FENCE();
      finish_win(wA, v, u);
      contract(v, u, wg + 9 * c * 4 * P);
      if (c + 1 >= CH) break;
      issue_win(c + 2, wA);
      PW4_FENCE();
      finish_win(wB, v, u);
      contract(v, u, wg + 9 * (c + 1) * 4 * P);
    }
  }
  if (tile_on && !K3) {
#ifdef CSN_EMU_SEQ
    const float* wg = lds;
#else
    const float* wg = lds + (lane & 3) * P;
#endif
    // high channels: quad -> high rows, 2x2 maximum -> low rows (pw4_hi_batch); two batches in flight: A and B hold batches
    // k0 and k0 + HB on entry of every trip
    int k0 = 0;
    for (int p = 0; p < (nfh >> 1); ++p) {
      pw4_hi_batch<NTH, NTL, HB, P, false>(hA, wg + k0 * 4 * P, HB, acch, accl);
      pw4_load_hi<HB>(rbh, oh0, oh1, csh, k0 + 2 * HB, CH, hA);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false>(hB, wg + (k0 + HB) * 4 * P, HB, acch, accl);
      if (k0 + 3 * HB < CH) pw4_load_hi<HB>(rbh, oh0, oh1, csh, k0 + 3 * HB, CH, hB);
      PW4_FENCE();
      k0 += 2 * HB;
    }
    if (nfh & 1) {   // A = full batch k0, B = the last (partial) batch
      pw4_hi_batch<NTH, NTL, HB, P, false>(hA, wg + k0 * 4 * P, HB, acch, accl);
      k0 += HB;
#pragma unroll
      for (int j = 0; j < HB; ++j) { hA[j][0] = hB[j][0]; hA[j][1] = hB[j][1]; }
    }
    const int nfl = (CL - 1) / LB;
    ilb_load_lo<LB>(rbl, olc, csl, 0, CL, lA);   // the low channels' first two batches fly during the last high batch
    if (nfl >= 1) ilb_load_lo<LB>(rbl, olc, csl, LB, CL, lB);
    PW4_FENCE();
    pw4_hi_batch<NTH, NTL, HB, P, true>(hA, wg + k0 * 4 * P, CH - k0, acch, accl);
    // low channels: the centre value -> z rows and low rows, same scheme
    const float* wgl = wg + CH * 4 * P;
    int c0 = 0;
    for (int p = 0; p < (nfl >> 1); ++p) {
      ilb_lo_batch<NTH, NTL, LB, P, false>(lA, wgl + c0 * 4 * P, LB, accz, accl);
      ilb_load_lo<LB>(rbl, olc, csl, c0 + 2 * LB, CL, lA);
      PW4_FENCE();
      ilb_lo_batch<NTH, NTL, LB, P, false>(lB, wgl + (c0 + LB) * 4 * P, LB, accz, accl);
      if (c0 + 3 * LB < CL) ilb_load_lo<LB>(rbl, olc, csl, c0 + 3 * LB, CL, lB);
      PW4_FENCE();
      c0 += 2 * LB;
    }
    if (nfl & 1) {
      ilb_lo_batch<NTH, NTL, LB, P, false>(lA, wgl + c0 * 4 * P, LB, accz, accl);
      c0 += LB;
#pragma unroll
      for (int j = 0; j < LB; ++j) lA[j] = lB[j];
    }
    ilb_lo_batch<NTH, NTL, LB, P, true>(lA, wgl + c0 * 4 * P, CL - c0, accz, accl);
    if (valid) {   // z = W_hl x_l at the low resolution, one plane per high row of the group
#pragma unroll
      for (int t = 0; t < NTH; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) Z[(4 * t + i) * HWl + pix] = accz[t][i];
    }
  }
  __syncthreads();
  ILB_STAMP(2);
  if (tile_on && valid) {
    // y_h = W_hh x_h + bilinear_up2(z) (csnet.py:702-707,720-722: the branches are summed in input order), then BN + PReLU
    int zo[9];
    {
      const int yy[3] = {max(y - 1, 0), y, min(y + 1, Hl - 1)};
      const int xx[3] = {max(x - 1, 0), x, min(x + 1, Wl - 1)};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) zo[3 * r + c] = yy[r] * Wl + xx[c];
    }
    const float* eph = par_eh;
    float* h1 = H1 + (2 * y + 1) * ph + 4 + 2 * x;
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * t + i;
        if (r < nch_h) {
          float v[9], q[4] = {0.f, 0.f, 0.f, 0.f};
          if (!K3) {
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = Z[r * HWl + zo[k]];
            pw4_up2_quad(v, q);
          }
          float o[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) o[s] = pw4_epi(acch[s][t][i] + q[s], eph[4 * r], eph[4 * r + 1], eph[4 * r + 2]);
          *reinterpret_cast<float2*>(h1 + r * plane_h) = make_float2(o[0], o[1]);
          *reinterpret_cast<float2*>(h1 + r * plane_h + ph) = make_float2(o[2], o[3]);
        }
      }
    if (NTL > 0) {
      const float* epl = par_el;
      float* l1 = L1 + (y + 1) * pl + 4 + x;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * t + i;
          if (r < nch_l) l1[r * plane_l] = pw4_epi(accl[t][i], epl[4 * r], epl[4 * r + 1], epl[4 * r + 2]);
        }
    }
  }
  __syncthreads();
  ILB_STAMP(3);

  // ---- phases 2, 3: the depthwise pair, tasks = (channel, row chunk, strip of four columns), high planes then low planes ----
  const int nsh = a->nsh, nrh = a->nrh, nsl = a->nsl, nrl = a->nrl;
  const int th = nch_h * nsh * nrh, tl = nch_l * nsl * nrl;
  for (int task = tid; task < th + tl; task += nthr) {
    const bool hi = task < th;
    ilb_dw_branch<false>(hi ? task : task - th, hi ? H1 : L1, hi ? H2 : L2, hi ? plane_h : plane_l, hi ? ph : pl, hi ? Hh : Hl, hi ? Wh : Wl,
                         hi ? a->Rh : a->Rl, hi ? nsh : nsl, hi ? nrh : nrl, hi ? a->m_hsr : a->m_lsr, hi ? a->m_hs : a->m_ls,
                         hi ? par_dh : par_dl, nullptr, nullptr, nullptr, 0);
  }
  __syncthreads();
  ILB_STAMP(4);
  for (int task = tid; task < th + tl; task += nthr) {
    const bool hi = task < th;
    const int Hb = hi ? Hh : Hl, Wb = hi ? Wh : Wl;
    const int64_t o = hi ? ((int64_t)b * OH + r0h) : ((int64_t)b * OL + r0l);
    float* pool = hi ? a->pool_h : a->pool_l;
    float* pmx = hi ? a->mp_h : a->mp_l;
    ilb_dw_branch<true>(hi ? task : task - th, hi ? H2 : L2, nullptr, hi ? plane_h : plane_l, hi ? ph : pl, Hb, Wb, hi ? a->Rh : a->Rl,
                        hi ? nsh : nsl, hi ? nrh : nrl, hi ? a->m_hsr : a->m_lsr, hi ? a->m_hs : a->m_ls, hi ? par_dh : par_dl,
                        (hi ? a->yh : a->yl) + o * Hb * Wb, pool ? pool + o * (Hb >> 1) * (Wb >> 1) : nullptr,
                        pmx ? pmx + o * (Hb >> 2) * (Wb >> 2) : nullptr, hi ? a->skip_h : a->skip_l);
  }
#ifdef ILB_TIMING
  __syncthreads();
#endif
  ILB_STAMP(5);
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
typedef void (*IlbFn)(IlbArgs);
struct IlbEntry { int nth, ntl; IlbFn fn; IlbFn fn3; };   // 1x1 block / 3x3 stride-2 entry block
#define ILB_ENTRY(H, L) {H, L, ilb_kernel<H, L, false>, ilb_kernel<H, L, true>}
static const IlbEntry g_ilb_table[] = {ILB_ENTRY(1, 1), ILB_ENTRY(1, 0), ILB_ENTRY(2, 2), ILB_ENTRY(2, 0), ILB_ENTRY(1, 2), ILB_ENTRY(2, 1)};

// LDS layout of an item for (nth, ntl) row tiles per group; returns the bytes, 0 when the geometry is not supported
size_t csn_ilb_layout(IlbArgs& a) {
  const int Hh = 2 * a.Hl, Wh = 2 * a.Wl;
  const int NT4 = (a.nth + a.ntl + 3) & ~3, P = PW4_PITCH(NT4);
  a.gimg_floats = (a.k3 ? 9 * a.CH : a.CH + a.CL) * 4 * P;
  a.ph = ((Wh + 3) & ~3) + 4; a.pl = ((a.Wl + 3) & ~3) + 4;
  a.plane_h = (Hh + 2) * a.ph + 4; a.plane_l = (a.Hl + 2) * a.pl + 4;   // + 4: the right frame of the last row
  int off = a.gimg_floats;
  a.off_h1 = off; off += 4 * a.nth * a.plane_h;
  a.off_h2 = off; off += 4 * a.nth * a.plane_h;
  a.off_l1 = off; off += 4 * a.ntl * a.plane_l;
  a.off_l2 = off; off += 4 * a.ntl * a.plane_l;
  a.off_z = off; off += 4 * a.nth * a.Hl * a.Wl;
  a.nsh = (Wh + 3) / 4; a.nrh = (Hh + a.Rh - 1) / a.Rh; a.nsl = (a.Wl + 3) / 4; a.nrl = (a.Hl + a.Rl - 1) / a.Rl;
  auto magic = [](unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); };
  a.m_hsr = magic((unsigned)(a.nsh * a.nrh)); a.m_hs = magic((unsigned)a.nsh);
  a.m_lsr = magic((unsigned)(a.nsl * a.nrl)); a.m_ls = magic((unsigned)a.nsl);
  a.off_par = off; off += (a.nth + a.ntl) * (16 + 96);   // epilogue records + depthwise records of the group's channels
  a.lds_floats = off;
  // a wave per tile of 64 low pixels (the accumulators live across the z barrier); one task per lane where the block allows
  const int tiles = a.k3 ? (a.Hl * a.Wl + 61) / 62 : (a.Hl * a.Wl + 63) / 64;   // (K3: halo-lane tiles of 62 pixels)
  if (tiles > 16) return 0;
  const int tasks = 4 * a.nth * ((Wh + 3) / 4) * ((Hh + a.Rh - 1) / a.Rh) + 4 * a.ntl * ((a.Wl + 3) / 4) * ((a.Hl + a.Rl - 1) / a.Rl);
  int nthr = std::max(64 * tiles, std::min(1024, (tasks + 63) & ~63));
  a.nthreads = nthr;
  return (size_t)off * sizeof(float);
}

bool csn_ilb_supported(int nth, int ntl) {
  for (const IlbEntry& e : g_ilb_table)
    if (e.nth == nth && e.ntl == ntl) return true;
  return false;
}

int csn_launch_ilb(const IlbArgs& a, void* stream) {
  const IlbEntry* e = nullptr;
  for (const IlbEntry& t : g_ilb_table)
    if (t.nth == a.nth && t.ntl == a.ntl) e = &t;
  if (!e) return -1;
  const size_t lds = (size_t)a.lds_floats * sizeof(float);
  if (lds > 160 * 1024) return -1;
#ifndef CSN_CPU_EMU
  if (lds > 64 * 1024) {   // once per device and function (it was issued on every launch -- host time on a latency chain)
    static CsnPerDeviceOnce once[sizeof(g_ilb_table) / sizeof(g_ilb_table[0])][2];
    const int st = once[e - g_ilb_table][a.k3 ? 1 : 0].run([&]() {
      return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(a.k3 ? e->fn3 : e->fn),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (st != 0) return st;
  }
#endif
  const int ipx = (a.B + 7) >> 3;
  const dim3 grid(8 * ipx * a.ng);
  if (a.k3) CSN_LAUNCH(e->fn3, grid, dim3(a.nthreads), lds, stream, a);
  else CSN_LAUNCH(e->fn, grid, dim3(a.nthreads), lds, stream, a);
  return (int)hipGetLastError();
}
