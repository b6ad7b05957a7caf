// k_pw4.hip -- gOctConv with 1x1 kernels (+ BN + PReLU) of a two-branch unit, operands straight from the load registers.
//
// Reference semantics (CSNet/model/csnet.py, gOctaveConv.forward 664-726 + gOctaveCBR.forward 778-792), branches
// h (resolution 2H x 2W) and l (H x W):
//   y_h = W_hh x_h + bilinear_up2(W_hl x_l)          (702-707, 716-717)
//   y_l = W_ll x_l + W_lh max_pool2(x_h)             (708-714, 716-717)
//   out_j = PReLU_j(BN_j(y_j))                       (789-791; eval BN folded to scale / shift)
// As in k_goct_pw.hip the low -> high term is evaluated as W_hl bilinear_up2(x_l) (a 1x1 convolution commutes with the
// interpolation), so every output pixel is ONE contraction over its gathered vector.
//
// MI355X mapping (round 3; replaces goct_pw_kernel for these units, profiles/r3_notes.md):
//   * a LANE owns one pixel (y, x) of branch l and the 2x2 quad (2y + dy, 2x + dx) of branch h above it; a wave owns a
//     tile of 64 low pixels (TW x 64/TW).  Everything a lane needs is in ITS registers: the quad (two 64-bit loads per
//     channel, a wave reads whole 128-byte row segments), the 3x3 neighbourhood of its low pixel (the bilinear taps of all
//     four quad pixels: constant weights 9/16, 3/16, 3/16, 1/16, borders by clamped addresses), the 2x2 maximum of the
//     quad.  No LDS panel, no cross-lane traffic, no barrier after the weights are staged.
//   * v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4x1 blocks per wave; block b = lanes 4b .. 4b+3, B operand = the lane's
//     own gathered value (one input channel of its pixel), A operand = W[4t + (lane & 3)][k] (the same four values in
//     every block, one ds_read_b128 from the LDS weight image delivers four row tiles), D = four consecutive output
//     channels 4t .. 4t+3 of the lane's OWN pixel.  Rows are padded to 4 (not 16), K is not padded at all, the
//     accumulators ARE the store registers: no transpose epilogue.  Exact fp32 FMA chain (same rounding as fmaf).
//   * output channels are cut into M groups of (NTH, NTL) row tiles so that the accumulators (16 NTH + 4 NTL registers)
//     leave room for two batches of loads in flight; the groups of a tile run on neighbouring waves of one block
//     (shared inputs hit L1).  Channel counts are run-time (loops over batches of HB / LB channels, next batch's loads
//     issued before the current batch is contracted); the tile counts are the template parameters.
//   * HBM traffic = unit inputs once + outputs once; tiles are walked in the XCD-aware order of k_goct_pw.hip.
#include "pw4_common.h"

#ifndef PW4_HB
#define PW4_HB 2    // high-branch channels per load batch (2 x 64-bit loads each; measured: 2 beats 4 and 8, profiles/r3_notes.md)
#endif
#ifndef PW4_LB
#define PW4_LB 2    // low-branch channels per load batch (9 dword loads each)
#endif
#ifndef PW4_OCC
#define PW4_OCC 2   // blocks per CU the register allocation aims at
#endif



template <int NTH, int NTL, bool RAW>
__global__ __launch_bounds__(CSN_BLOCK, PW4_OCC) void pw4_kernel(Pw4Args a_byval) {
  constexpr int HB = PW4_HB, LB = PW4_LB;
  constexpr int NT4 = (NTH + NTL + 3) & ~3, P = PW4_PITCH(NT4);
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS Pw4Args* a = CSN_KERNARG(Pw4Args, a_byval);
  const int tid = threadIdx.x;
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a->wimg);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int n4 = (a->ngroups * a->gimg_floats) >> 2;
    for (int i = tid; i < n4; i += CSN_BLOCK) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  const int CH = a->CH, CL = a->CL, Hl = a->Hl, Wl = a->Wl, Wh = 2 * Wl;
  const unsigned csl = (unsigned)(Hl * Wl) * 4u, csh = csl * 4u;   // channel strides in bytes
  const int twl = a->twl;
  const int lx = lane & ((1 << twl) - 1), ly = lane >> twl;
  const int ng = a->ngroups;
  const int tiles_xy = a->tiles_x * a->tiles_y;
  const int nitems = tiles_xy * a->B * ng;
  // XCD-aware order (see k_goct_pw.hip): XCD x = blockIdx.x & 7 walks the contiguous item range [x * chunk, (x + 1) * chunk)
  const int nslot = (int)(gridDim.x >> 3) * 4;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;   // whole tiles per XCD
  const int xcd = blockIdx.x & 7;
  const int iend = min((xcd + 1) * chunk, nitems);
#ifdef CSN_CPU_EMU
  const float* wl_lane = lds;
#else
  const float* wl_lane = lds + (lane & 3) * P;
#endif
  for (int item = xcd * chunk + (int)(blockIdx.x >> 3) * 4 + wave; item < iend; item += nslot) {
    const int tile = item / ng, g = item - tile * ng;
    const int b = tile / tiles_xy, txy = tile - b * tiles_xy;
    const int ty = txy / a->tiles_x, tx = txy - ty * a->tiles_x;
    const int y = (ty << (6 - twl)) + ly, x = (tx << twl) + lx;
    const bool valid = y < Hl && x < Wl;
    const int yc = min(y, Hl - 1), xc = min(x, Wl - 1);
    unsigned ol[9];
    {
      const int yy[3] = {max(yc - 1, 0), yc, min(yc + 1, Hl - 1)};
      const int xx[3] = {max(xc - 1, 0), xc, min(xc + 1, Wl - 1)};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) ol[3 * r + c] = (unsigned)(yy[r] * Wl + xx[c]) * 4u;
    }
    const unsigned oh0 = (unsigned)((2 * yc) * Wh + 2 * xc) * 4u, oh1 = oh0 + (unsigned)Wh * 4u;
    const csn_buf rbh = csn_make_buf_n(a->xh + (int64_t)b * CH * (int64_t)(csh >> 2), (unsigned)CH * csh);
    const csn_buf rbl = csn_make_buf_n(a->xl + (int64_t)b * CL * (int64_t)(csl >> 2), (unsigned)CL * csl);
    const float* wg = wl_lane + g * a->gimg_floats;

    csn_f4 acch[4][NTH > 0 ? NTH : 1], accl[NTL > 0 ? NTL : 1];
#pragma unroll
    for (int t = 0; t < (NTH > 0 ? NTH : 1); ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acch[s][t][i] = 0.f;
#pragma unroll
    for (int t = 0; t < (NTL > 0 ? NTL : 1); ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) accl[t][i] = 0.f;

    // ---- high-branch channels: batch k0 is contracted while batch k0 + HB is in flight.  Two register sets; the full
    // batches are walked in pairs so that no set is copied inside the loop, and the last (possibly partial) batch always
    // ends up in set A ----
    float2 hA[HB][2], hB[HB][2];
    float lA[LB][9], lB[LB][9];
    pw4_load_hi<HB>(rbh, oh0, oh1, csh, 0, CH, hA);
      PW4_FENCE();
    const int nfh = (CH - 1) / HB;   // full batches in front of the last one
    int k0 = 0;
    for (int p = 0; p < (nfh >> 1); ++p) {
      pw4_load_hi<HB>(rbh, oh0, oh1, csh, k0 + HB, CH, hB);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false>(hA, wg + k0 * 4 * P, HB, acch, accl);
      pw4_load_hi<HB>(rbh, oh0, oh1, csh, k0 + 2 * HB, CH, hA);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false>(hB, wg + (k0 + HB) * 4 * P, HB, acch, accl);
      k0 += 2 * HB;
    }
    if (nfh & 1) {
      pw4_load_hi<HB>(rbh, oh0, oh1, csh, k0 + HB, CH, hB);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false>(hA, wg + k0 * 4 * P, HB, acch, accl);
      k0 += HB;
#pragma unroll
      for (int j = 0; j < HB; ++j) { hA[j][0] = hB[j][0]; hA[j][1] = hB[j][1]; }
    }
    pw4_load_lo<LB>(rbl, ol, csl, 0, CL, lA);
      PW4_FENCE();
    pw4_hi_batch<NTH, NTL, HB, P, true>(hA, wg + k0 * 4 * P, CH - k0, acch, accl);
    // ---- low-branch channels, same scheme ----
    const float* wgl = wg + CH * 4 * P;
    const int nfl = (CL - 1) / LB;
    int c0 = 0;
    for (int p = 0; p < (nfl >> 1); ++p) {
      pw4_load_lo<LB>(rbl, ol, csl, c0 + LB, CL, lB);
      PW4_FENCE();
      pw4_lo_batch<NTH, NTL, LB, P, false>(lA, wgl + c0 * 4 * P, LB, acch, accl);
      pw4_load_lo<LB>(rbl, ol, csl, c0 + 2 * LB, CL, lA);
      PW4_FENCE();
      pw4_lo_batch<NTH, NTL, LB, P, false>(lB, wgl + (c0 + LB) * 4 * P, LB, acch, accl);
      c0 += 2 * LB;
    }
    if (nfl & 1) {
      pw4_load_lo<LB>(rbl, ol, csl, c0 + LB, CL, lB);
      PW4_FENCE();
      pw4_lo_batch<NTH, NTL, LB, P, false>(lA, wgl + c0 * 4 * P, LB, acch, accl);
      c0 += LB;
#pragma unroll
      for (int j = 0; j < LB; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) lA[j][t] = lB[j][t];
    }
    pw4_lo_batch<NTH, NTL, LB, P, true>(lA, wgl + c0 * 4 * P, CL - c0, acch, accl);

    // ---- epilogue: folded BN + PReLU, the accumulators are the store registers.  Row tiles past the group's list (an
    // instantiation wider than the group) are skipped; rows past the tensor's last channel inside the last tile fall out
    // of the bounded resource and are dropped by the hardware ----
    const unsigned sv0 = valid ? oh0 : 0x80000000u, sv1 = valid ? oh1 : 0x80000000u;
    if (NTH > 0) {
      const int r0 = a->grp[g].r0h, nt = a->grp[g].nth;
      const csn_buf ob = csn_make_buf_n(a->yh + (int64_t)b * a->OH * (int64_t)(csh >> 2), (unsigned)a->OH * csh);
      csn_cfp ep = csn_const(a->ep_h) + 4 * r0;
#pragma unroll
      for (int t = 0; t < NTH; ++t) {
        if (t < nt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * t + i;
            float o[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) o[s] = RAW ? acch[s][t][i] : pw4_epi(acch[s][t][i], ep[4 * r], ep[4 * r + 1], ep[4 * r + 2]);
            const unsigned so = (unsigned)(r0 + r) * csh;
            csn_st2(ob, sv0, so, make_float2(o[0], o[1]));
            csn_st2(ob, sv1, so, make_float2(o[2], o[3]));
          }
        }
      }
    }
    if (NTL > 0) {
      const int r0 = a->grp[g].r0l, nt = a->grp[g].ntl;
      const csn_buf ob = csn_make_buf_n(a->yl + (int64_t)b * a->OL * (int64_t)(csl >> 2), (unsigned)a->OL * csl);
      csn_cfp ep = csn_const(a->ep_l) + 4 * r0;
      const unsigned sv = valid ? ol[4] : 0x80000000u;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        if (t < nt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * t + i;
            const float o = RAW ? accl[t][i] : pw4_epi(accl[t][i], ep[4 * r], ep[4 * r + 1], ep[4 * r + 2]);
            csn_st1(ob, sv, (unsigned)(r0 + r) * csl, o);
          }
        }
      }
    }
  }
}

// ---- instantiation table -----------------------------------------------------------------------------------------
#define PW4_INST_LIST(X) \
  X(3, 3) X(3, 4) X(4, 3) X(4, 4) X(5, 3) X(4, 6) X(4, 0) X(5, 0) X(6, 0) X(2, 2) X(3, 0) X(2, 0) X(1, 1) X(5, 2) X(2, 5)

typedef void (*Pw4Fn)(Pw4Args);
struct Pw4Entry { int nth, ntl; Pw4Fn fn[2]; };
#define PW4_ENTRY(H, L) {H, L, {pw4_kernel<H, L, false>, pw4_kernel<H, L, true>}},
static const Pw4Entry g_pw4_table[] = {PW4_INST_LIST(PW4_ENTRY)};

// smallest instantiation that covers (nth, ntl) row tiles per group (ntl = 0 must stay 0: no low output), or {0, 0}
bool csn_pw4_pick(int nth, int ntl, int* pnth, int* pntl) {
  int best = -1, best_cost = 1 << 30;
  for (size_t i = 0; i < sizeof(g_pw4_table) / sizeof(g_pw4_table[0]); ++i) {
    const Pw4Entry& e = g_pw4_table[i];
    if (e.nth < nth || e.ntl < ntl || (ntl == 0) != (e.ntl == 0)) continue;
    const int cost = 16 * e.nth + 4 * e.ntl;
    if (cost < best_cost) { best_cost = cost; best = (int)i; }
  }
  if (best < 0) return false;
  *pnth = g_pw4_table[best].nth; *pntl = g_pw4_table[best].ntl;
  return true;
}

int csn_launch_pw4(const Pw4Args& a, int raw, void* stream) {
  const Pw4Entry* e = nullptr;
  for (size_t i = 0; i < sizeof(g_pw4_table) / sizeof(g_pw4_table[0]); ++i)
    if (g_pw4_table[i].nth == a.nth && g_pw4_table[i].ntl == a.ntl) e = &g_pw4_table[i];
  if (!e) return 1;   // hipErrorInvalidValue
  const int nitems = a.tiles_x * a.tiles_y * a.B * a.ngroups;
  int nblk = (nitems + 3) / 4;
  if (nblk > a.max_grid) nblk = a.max_grid;
  const dim3 grid((nblk + 7) & ~7);
  const size_t lds = (size_t)a.ngroups * a.gimg_floats * sizeof(float);
  CSN_LAUNCH(e->fn[raw ? 1 : 0], grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
