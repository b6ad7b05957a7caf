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
#include <cstring>

#include "pw4_common.h"

// Three-branch units (CSFHead.fuse / fuse1x1, csnet.py:152-206) add a third input x2 at half the resolution of branch l; it
// is supported by the single-output forms of the kernel: NTL = 0 (only y_h: x2 enters through bilinear x4 of the 3x3
// neighbourhood of the lane's parent pixel, separable three-tap weights that depend on the lane's parity) and NTH = 0 (only
// y_l: the usual four-tap bilinear x2; this form also takes a fourth input xq at FOUR times the resolution of branch l, through a
// 4x4 max-pool -- CSFHead.fuse's lowest output branch, F.max_pool2d(x, 4, 4) of csnet.py:708-714).  With `red_w` the rows are not stored but reduced to ONE channel
// red_b + sum_r red_w[r] * PReLU(BN(y_r)) -- cls_layer (csnet.py:306-308,381) riding in fuse1x1's epilogue; the M groups of
// a tile are then walked by the same wave so that the sum stays in its registers.
#ifndef PW4_HB
#define PW4_HB 2    // high-branch channels per load batch (2 x 64-bit loads each; measured: 2 beats 4 and 8, profiles/r3_notes.md)
#endif
#ifndef PW4_HB_BF
#define PW4_HB_BF 4  // ... with bfloat16 tensors: a channel is two dword loads, so four channels put the float form's bytes in flight
                     // (round 6, same lease: bf16 train step 35.10 -> 34.54 ms)
#endif
#ifndef PW4_LB
#define PW4_LB 2    // low-branch channels per load batch (9 dword loads each)
#endif
#ifndef PW4_OCC
#define PW4_OCC 2   // blocks per CU the register allocation aims at
#endif
// accumulator registers up to which a form is compiled for three waves per SIMD (168 registers).  bfloat16 tensors: the raw
// register sets AND their converted values are live in a batch -- the (5, 3) / (4, 6) forms spilled 136-192 bytes per lane at 168
#ifndef PW4_ACC3_BF
#define PW4_ACC3_BF 84
#endif
#define PW4_ACC3(AT) (sizeof(AT) == 2 ? PW4_ACC3_BF : 92)




namespace {

// bilinear taps of the third input (tensor at half the resolution of branch l: H2 x W2) for the lane's low pixel (y, x)
struct Pw4X2 {
  unsigned o[9];        // NTL = 0: 3x3 neighbourhood of the parent pixel (y >> 1, x >> 1), clamped; NTH = 0: o[0..3] = the four taps
  float wy[2][3];       // NTL = 0: row weights of quad row dy over (parent row - 1, parent row, parent row + 1)
  float wx[2][3];       //          column weights of quad column dx
  float w4[4];          // NTH = 0: weights of the four taps
};

// upsample_bilinear2d, align_corners=False, scale 4: quad row Y = 2y + dy has source (Y + 0.5) / 4 - 0.5 = parent + {-0.375,
// -0.125, +0.125, +0.375} for Y & 3 = 0..3; clamped neighbours reproduce PyTorch's clamped source index at the borders
__device__ __forceinline__ void pw4_up4_weights(int ylow, float (&w)[2][3]) {
  const bool odd = ylow & 1;
  w[0][0] = odd ? 0.f : 0.375f; w[0][1] = odd ? 0.875f : 0.625f; w[0][2] = odd ? 0.125f : 0.f;
  w[1][0] = odd ? 0.f : 0.125f; w[1][1] = odd ? 0.625f : 0.875f; w[1][2] = odd ? 0.375f : 0.f;
}

template <bool HI>
__device__ __forceinline__ void pw4_x2_geo(int y, int x, int H2, int W2, unsigned E, Pw4X2& g) {
  const int py = y >> 1, px = x >> 1;
  if (HI) {
    const int yy[3] = {max(py - 1, 0), py, min(py + 1, H2 - 1)};
    const int xx[3] = {max(px - 1, 0), px, min(px + 1, W2 - 1)};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) g.o[3 * r + c] = (unsigned)(yy[r] * W2 + xx[c]) * E;
    pw4_up4_weights(y, g.wy);
    pw4_up4_weights(x, g.wx);
  } else {   // scale 2: source y / 2 - 0.25 -> (py - 1: 0.25, py: 0.75) for even y, (py: 0.75, py + 1: 0.25) for odd y
    const int ya = max((y & 1) ? py : py - 1, 0), yb = min((y & 1) ? py + 1 : py, H2 - 1);
    const int xa = max((x & 1) ? px : px - 1, 0), xb = min((x & 1) ? px + 1 : px, W2 - 1);
    const float wyb = (y & 1) ? 0.25f : 0.75f, wxb = (x & 1) ? 0.25f : 0.75f;
    g.o[0] = (unsigned)(ya * W2 + xa) * E; g.o[1] = (unsigned)(ya * W2 + xb) * E;
    g.o[2] = (unsigned)(yb * W2 + xa) * E; g.o[3] = (unsigned)(yb * W2 + xb) * E;
    g.w4[0] = (1.f - wyb) * (1.f - wxb); g.w4[1] = (1.f - wyb) * wxb; g.w4[2] = wyb * (1.f - wxb); g.w4[3] = wyb * wxb;
  }
}

template <int N, int XB, typename AT = float>
__device__ __forceinline__ void pw4_load_x2(csn_buf rb, const Pw4X2& g, unsigned cs, int c0, int C,
                                            typename csn_bufacc<AT>::r1 (&v)[XB][N]) {
#pragma unroll
  for (int j = 0; j < XB; ++j) {
    const unsigned so = (unsigned)min(c0 + j, C - 1) * cs;
#pragma unroll
    for (int t = 0; t < N; ++t) v[j][t] = csn_bufacc<AT>::ldr1(rb, g.o[t], so);
  }
}

// one channel of the third input: NTL = 0 -> quad values by separable three-tap interpolation -> high rows;
// NTH = 0 -> one value -> low rows
template <int NTH, int NTL, int P, int N, typename AT = float>
__device__ __forceinline__ void pw4_x2_channel(const typename csn_bufacc<AT>::r1 (&raw)[N], const Pw4X2& g, const float* wk,
                                               csn_f4 (&acch)[4][NTH > 0 ? NTH : 1], csn_f4 (&accl)[NTL > 0 ? NTL : 1]) {
  constexpr int NT4 = (NTH + NTL + 3) & ~3;
  Pw4A<NT4> a;
  pw4_load_a<NT4, P>(wk, a);
  float v[N];
#pragma unroll
  for (int t = 0; t < N; ++t) v[t] = csn_bufacc<AT>::cv1(raw[t]);
  if (NTH > 0) {
    float h[2][3];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
      for (int r = 0; r < 3; ++r)
        h[dx][r] = fmaf(g.wx[dx][2], v[(3 * r + 2) % N], fmaf(g.wx[dx][1], v[(3 * r + 1) % N], g.wx[dx][0] * v[(3 * r) % N]));
    float q[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
        q[2 * dy + dx] = fmaf(g.wy[dy][2], h[dx][2], fmaf(g.wy[dy][1], h[dx][1], g.wy[dy][0] * h[dx][0]));
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) pw4_mfma<NT4>(a, t, q[s], acch[s][t]);
  } else {
    const float m = fmaf(g.w4[3], v[3 % N], fmaf(g.w4[2], v[2 % N], fmaf(g.w4[1], v[1 % N], g.w4[0] * v[0])));
#pragma unroll
    for (int t = 0; t < NTL; ++t) pw4_mfma<NT4>(a, NTH + t, m, accl[t]);
  }
}

// fourth input (low-only form): 4x4 max-pool of a tensor at four times the resolution of branch l; o = byte offset of
// (4 y, 4 x) inside a channel plane (16-byte aligned: W is a multiple of 4 there), ws = its row pitch in bytes
template <int QB, typename AT>
__device__ __forceinline__ void pw4_load_xq(csn_buf rb, unsigned o, unsigned ws, unsigned cs, int c0, int C,
                                            typename csn_bufacc<AT>::r4 (&v)[QB][4]) {
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const unsigned so = (unsigned)min(c0 + j, C - 1) * cs;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[j][r] = csn_bufacc<AT>::ldr4(rb, o + (unsigned)r * ws, so);
  }
}
template <int NTL, int P, typename AT = float>
__device__ __forceinline__ void pw4_xq_channel(const typename csn_bufacc<AT>::r4 (&raw)[4], const float* wk, csn_f4 (&accl)[NTL > 0 ? NTL : 1]) {
  constexpr int NT4 = (NTL + 3) & ~3;
  Pw4A<NT4> a;
  pw4_load_a<NT4, P>(wk, a);
  float4 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = csn_bufacc<AT>::cv4(raw[r]);
  float m = fmaxf(fmaxf(v[0].x, v[0].y), fmaxf(v[0].z, v[0].w));
#pragma unroll
  for (int r = 1; r < 4; ++r) m = fmaxf(m, fmaxf(fmaxf(v[r].x, v[r].y), fmaxf(v[r].z, v[r].w)));
#pragma unroll
  for (int t = 0; t < NTL; ++t) pw4_mfma<NT4>(a, t, m, accl[t]);
}

// Statistics of one row tile of an item (Pw4Args::stats_h / stats_l): v = the lane's {sum, sum of squares} of its stored values
// of the tile's four rows (2 i + {0, 1}; zeros for lanes outside the map).  The eight values are summed over the wave
// (csn_wave_reduce_scatter8) and leave with one store by eight lanes: slab `tile` of channels c0 .. c0 + 3 of `tab` (C channels).
__device__ __forceinline__ void pw4_stats_tile(double* tab, int c0, int C, int64_t stride, int tile, float (&v)[8], int lane) {
  if (!tab) return;   // (uniform)
#ifdef CSN_EMU_SEQ
  // stand-in (the lanes are sequential fibers; the launcher has zeroed the tables): every lane adds its own values
  for (int k = 0; k < 8; ++k)
    if (c0 + (k >> 1) < C) tab[((int64_t)(c0 + (k >> 1)) * stride + tile) * 2 + (k & 1)] += (double)v[k];
#else
  const float tot = csn_wave_reduce_scatter8(v, lane);
  const int k = csn_rs8_index(lane), c = c0 + (k >> 1);
  if ((lane & 7) == 0 && c < C) tab[((int64_t)c * stride + tile) * 2 + (k & 1)] = (double)tot;
#endif
}

}  // namespace

// MODE 0: BN + PReLU epilogue, rows stored; 1 (RAW): plain sums stored; 2 (RED): BN + PReLU, rows reduced with red_w
// AT: element type of the activation tensors (float; csn_bf16 = the bf16 train mode's storage, RAW only)
template <int NTH, int NTL, int MODE, typename AT = float>
__global__ __launch_bounds__(CSN_BLOCK, (16 * NTH + 4 * NTL <= PW4_ACC3(AT) && !(NTL == 0 && NTH >= 5) && PW4_OCC < 3) ? 3 : PW4_OCC)
void pw4_kernel(Pw4Args a_byval) {   // three waves per SIMD where the accumulators leave room (the high-only forms carry the
                                     // third input's staging registers: two waves from five row tiles on)
  constexpr bool RAW = MODE == 1, RED = MODE == 2;
  constexpr unsigned E = (unsigned)sizeof(AT);
  // Pw4Args::stats_h / stats_l: raw launches over bfloat16 tensors, forms with a low output.  The sums are formed whenever the form
  // has them (a run-time switch around the epilogue's statistics cost 30-80 registers: spills at the 168-register cap); only the
  // stores depend on the pointers.  The high-only forms also stage the third input and have no registers to spare: their units'
  // statistics stay with bn_stats_kernel (csn_pw4_has_stats)
  constexpr bool STATS = RAW && E == 2 && NTL > 0;
  // load batches: the low-only form contracts ~2 MFMAs per loaded value and has few accumulators -- its batches are deep
  // (every batch is one exposed memory round trip per item)
  // (bfloat16: the deeper high batches for the two-output forms only -- the high-only forms, which also stage the third input, spill with them)
  constexpr int HB = NTH == 0 ? 8 : ((E == 2 && NTL > 0) ? PW4_HB_BF : PW4_HB), LB = NTH == 0 ? 8 : PW4_LB;
  constexpr int NT4 = (NTH + NTL + 3) & ~3, P = PW4_PITCH(NT4);
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS Pw4Args* a = CSN_KERNARG(Pw4Args, a_byval);
  const int tid = threadIdx.x;
  csn_fill_lds16(lds, a->wimg, (a->ngroups * a->gimg_floats) >> 2, tid);
  __syncthreads();
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  constexpr bool do_stats = STATS;
  const int CH = a->CH, CL = a->CL, Hl = a->Hl, Wl = a->Wl, Wh = 2 * Wl;
  const unsigned csl = (unsigned)(Hl * Wl) * E, csh = csl * 4u;   // channel strides in bytes
  const int twl = a->twl;
  const int lx = lane & ((1 << twl) - 1), ly = lane >> twl;
  constexpr bool gloop = RED;                 // row reduction: the groups of a tile are walked by one wave
  const int ng = gloop ? 1 : a->ngroups;      // items per tile
  const int tiles_xy = a->tiles_x * a->tiles_y;
  const int nitems = tiles_xy * a->B * ng;
  constexpr int NX2 = NTL == 0 ? 9 : 4;      // taps per channel of the third input
  constexpr int XB = NTL == 0 ? 1 : 4;       // ... and channels per step (high-only forms: 2 / 4 per step measured, no gain)
  // XCD-aware order (see k_goct_pw.hip): XCD x = blockIdx.x & 7 walks the contiguous item range [x * chunk, (x + 1) * chunk)
  const int nslot = (int)(gridDim.x >> 3) * 4;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;   // whole tiles per XCD
  const int xcd = blockIdx.x & 7;
  const int iend = min((xcd + 1) * chunk, nitems);
#ifdef CSN_EMU_SEQ
  const float* wl_lane = lds;
#else
  const float* wl_lane = lds + (lane & 3) * P;
#endif
  for (int item = xcd * chunk + (int)(blockIdx.x >> 3) * 4 + wave; item < iend; item += nslot) {
    const int tile = item / ng, g_first = item - tile * ng;
    const int b = tile / tiles_xy, txy = tile - b * tiles_xy;
    int y, x;
    if (twl >= PW4_FLAT_TWL) {
      // flat tiles (round 4): the wave's 64 low pixels are consecutive in the plane's row-major order -- no idle lanes where the
      // row width is not a multiple of the tile width (112, 56, 28, 14: one lane in eight was idle), 12.5 % fewer items; a wave
      // then spans a row boundary, which only splits its row segments (every address is per lane anyway)
      const int p = txy * 64 + lane;
      int q = (int)((float)p * (1.0f / (float)Wl));
      q -= (q * Wl > p) ? 1 : 0;
      q += ((q + 1) * Wl <= p) ? 1 : 0;
      y = q; x = p - q * Wl;   // p >= Hl * Wl: y >= Hl, invalid
    } else {
      const int ty = txy / a->tiles_x, tx = txy - ty * a->tiles_x;
      y = (ty << (6 - twl)) + ly; x = (tx << twl) + lx;
    }
    const bool valid = y < Hl && x < Wl;
    const int yc = min(y, Hl - 1), xc = min(x, Wl - 1);
    unsigned ol[9];
    {
      const int yy[3] = {max(yc - 1, 0), yc, min(yc + 1, Hl - 1)};
      const int xx[3] = {max(xc - 1, 0), xc, min(xc + 1, Wl - 1)};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) ol[3 * r + c] = (unsigned)(yy[r] * Wl + xx[c]) * E;
    }
    const unsigned oh0 = (unsigned)((2 * yc) * Wh + 2 * xc) * E, oh1 = oh0 + (unsigned)Wh * E;
    const csn_buf rbh = csn_make_buf_n(reinterpret_cast<const char*>(a->xh) + (int64_t)b * CH * (int64_t)csh, (unsigned)CH * csh);
    const csn_buf rbl = csn_make_buf_n(reinterpret_cast<const char*>(a->xl) + (int64_t)b * CL * (int64_t)csl, (unsigned)CL * csl);
    const int C2 = (NTH == 0 || NTL == 0) ? a->C2 : 0;
    const unsigned cs2 = csl >> 2;
    const csn_buf rb2 = csn_make_buf_n(C2 > 0 ? reinterpret_cast<const char*>(a->x2) + (int64_t)b * C2 * (int64_t)cs2
                                              : reinterpret_cast<const char*>(a->xl), C2 > 0 ? (unsigned)C2 * cs2 : 4u);
    Pw4X2 g2;
    if (C2 > 0) pw4_x2_geo<NTL == 0>(yc, xc, Hl >> 1, Wl >> 1, E, g2);
    float red[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef CSN_KO_PW4_NOSTORE   // knock-out build: every store out of range (the instructions stay, their bytes go; wrong results)
    const unsigned sv0 = 0x80000000u, sv1 = 0x80000000u;
#else
    const unsigned sv0 = valid ? oh0 : 0x80000000u, sv1 = valid ? oh1 : 0x80000000u;
#endif
    const int g_last = gloop ? a->ngroups : g_first + 1;
    for (int g = g_first; g < g_last; ++g) {
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
    typedef csn_bufacc<AT> ACC;   // register sets hold the loads raw (converted where they are contracted)
    typename ACC::r2 hA[HB][2], hB[HB][2];
    typename ACC::r1 lA[LB][9], lB[LB][9];
    pw4_load_hi<HB, AT>(rbh, oh0, oh1, csh, 0, CH, hA);
      PW4_FENCE();
    const int nfh = (CH - 1) / HB;   // full batches in front of the last one
    int k0 = 0;
    for (int p = 0; p < (nfh >> 1); ++p) {
      pw4_load_hi<HB, AT>(rbh, oh0, oh1, csh, k0 + HB, CH, hB);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false, AT>(hA, wg + k0 * 4 * P, HB, acch, accl);
      pw4_load_hi<HB, AT>(rbh, oh0, oh1, csh, k0 + 2 * HB, CH, hA);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false, AT>(hB, wg + (k0 + HB) * 4 * P, HB, acch, accl);
      k0 += 2 * HB;
    }
    if (nfh & 1) {
      pw4_load_hi<HB, AT>(rbh, oh0, oh1, csh, k0 + HB, CH, hB);
      PW4_FENCE();
      pw4_hi_batch<NTH, NTL, HB, P, false, AT>(hA, wg + k0 * 4 * P, HB, acch, accl);
      k0 += HB;
#pragma unroll
      for (int j = 0; j < HB; ++j) { hA[j][0] = hB[j][0]; hA[j][1] = hB[j][1]; }
    }
    pw4_load_lo<LB, AT>(rbl, ol, csl, 0, CL, lA);
      PW4_FENCE();
    pw4_hi_batch<NTH, NTL, HB, P, true, AT>(hA, wg + k0 * 4 * P, CH - k0, acch, accl);
    // ---- low-branch channels, same scheme ----
    const float* wgl = wg + CH * 4 * P;
    const int nfl = (CL - 1) / LB;
    int c0 = 0;
    for (int p = 0; p < (nfl >> 1); ++p) {
      pw4_load_lo<LB, AT>(rbl, ol, csl, c0 + LB, CL, lB);
      PW4_FENCE();
      pw4_lo_batch<NTH, NTL, LB, P, false, AT>(lA, wgl + c0 * 4 * P, LB, acch, accl);
      pw4_load_lo<LB, AT>(rbl, ol, csl, c0 + 2 * LB, CL, lA);
      PW4_FENCE();
      pw4_lo_batch<NTH, NTL, LB, P, false, AT>(lB, wgl + (c0 + LB) * 4 * P, LB, acch, accl);
      c0 += 2 * LB;
    }
    if (nfl & 1) {
      pw4_load_lo<LB, AT>(rbl, ol, csl, c0 + LB, CL, lB);
      PW4_FENCE();
      pw4_lo_batch<NTH, NTL, LB, P, false, AT>(lA, wgl + c0 * 4 * P, LB, acch, accl);
      c0 += LB;
#pragma unroll
      for (int j = 0; j < LB; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) lA[j][t] = lB[j][t];
    }
    pw4_lo_batch<NTH, NTL, LB, P, true, AT>(lA, wgl + c0 * 4 * P, CL - c0, acch, accl);
    // ---- third input (single-output forms only), XB channels per step, the next step in flight ----
    if ((NTH == 0 || NTL == 0) && C2 > 0) {
      const float* wg2 = wgl + CL * 4 * P;
      typename ACC::r1 xA[XB][NX2], xB[XB][NX2];
      pw4_load_x2<NX2, XB, AT>(rb2, g2, cs2, 0, C2, xA);
      PW4_FENCE();
      const int nf2 = (C2 - 1) / XB;
      int c = 0;
      for (int p = 0; p < (nf2 >> 1); ++p) {
        pw4_load_x2<NX2, XB, AT>(rb2, g2, cs2, c + XB, C2, xB);
        PW4_FENCE();
#pragma unroll
        for (int j = 0; j < XB; ++j) pw4_x2_channel<NTH, NTL, P, NX2, AT>(xA[j], g2, wg2 + (c + j) * 4 * P, acch, accl);
        pw4_load_x2<NX2, XB, AT>(rb2, g2, cs2, c + 2 * XB, C2, xA);
        PW4_FENCE();
#pragma unroll
        for (int j = 0; j < XB; ++j) pw4_x2_channel<NTH, NTL, P, NX2, AT>(xB[j], g2, wg2 + (c + XB + j) * 4 * P, acch, accl);
        c += 2 * XB;
      }
      if (nf2 & 1) {
        pw4_load_x2<NX2, XB, AT>(rb2, g2, cs2, c + XB, C2, xB);
        PW4_FENCE();
#pragma unroll
        for (int j = 0; j < XB; ++j) pw4_x2_channel<NTH, NTL, P, NX2, AT>(xA[j], g2, wg2 + (c + j) * 4 * P, acch, accl);
        c += XB;
#pragma unroll
        for (int j = 0; j < XB; ++j)
#pragma unroll
          for (int t = 0; t < NX2; ++t) xA[j][t] = xB[j][t];
      }
#pragma unroll
      for (int j = 0; j < XB; ++j)
        if (c + j < C2) pw4_x2_channel<NTH, NTL, P, NX2, AT>(xA[j], g2, wg2 + (c + j) * 4 * P, acch, accl);
    }

    // ---- fourth input (low-only form): 4x4 max-pool of xq, two channels per step ----
    if (NTH == 0 && a->CQ > 0) {
      constexpr int QB = 2;
      const int CQ = a->CQ;
      const unsigned csq = csl * 16u, wsq = (unsigned)(4 * Wl) * E;
      const csn_buf rbq = csn_make_buf_n(reinterpret_cast<const char*>(a->xq) + (int64_t)b * CQ * (int64_t)csq, (unsigned)CQ * csq);
      const unsigned oq = (unsigned)((4 * yc) * (4 * Wl) + 4 * xc) * E;
      const float* wgq = wgl + (CL + C2) * 4 * P;
      typename ACC::r4 qA[QB][4], qB[QB][4];
      pw4_load_xq<QB, AT>(rbq, oq, wsq, csq, 0, CQ, qA);
      PW4_FENCE();
      const int nfq = (CQ - 1) / QB;
      int c = 0;
      for (int p = 0; p < (nfq >> 1); ++p) {
        pw4_load_xq<QB, AT>(rbq, oq, wsq, csq, c + QB, CQ, qB);
        PW4_FENCE();
#pragma unroll
        for (int j = 0; j < QB; ++j) pw4_xq_channel<NTL, P, AT>(qA[j], wgq + (c + j) * 4 * P, accl);
        pw4_load_xq<QB, AT>(rbq, oq, wsq, csq, c + 2 * QB, CQ, qA);
        PW4_FENCE();
#pragma unroll
        for (int j = 0; j < QB; ++j) pw4_xq_channel<NTL, P, AT>(qB[j], wgq + (c + QB + j) * 4 * P, accl);
        c += 2 * QB;
      }
      if (nfq & 1) {
        pw4_load_xq<QB, AT>(rbq, oq, wsq, csq, c + QB, CQ, qB);
        PW4_FENCE();
#pragma unroll
        for (int j = 0; j < QB; ++j) pw4_xq_channel<NTL, P, AT>(qA[j], wgq + (c + j) * 4 * P, accl);
        c += QB;
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) qA[j][r] = qB[j][r];
      }
#pragma unroll
      for (int j = 0; j < QB; ++j)
        if (c + j < CQ) pw4_xq_channel<NTL, P, AT>(qA[j], wgq + (c + j) * 4 * P, accl);
    }

    // ---- epilogue: folded BN + PReLU, the accumulators are the store registers.  Row tiles past the group's list (an
    // instantiation wider than the group) are skipped; rows past the tensor's last channel inside the last tile fall out
    // of the bounded resource and are dropped by the hardware ----
    if (NTH > 0) {
      const int r0 = a->grp[g].r0h, nt = a->grp[g].nth;
      const csn_buf ob = csn_make_buf_n(reinterpret_cast<char*>(a->yh) + (int64_t)b * a->OH * (int64_t)csh, (unsigned)a->OH * csh);
      csn_cfp ep = csn_const(a->ep_h) + 4 * r0;
#pragma unroll
      for (int t = 0; t < NTH; ++t) {
        if (t < nt) {
          float stv[8];   // statistics: the lane's {sum, sum of squares} per row of the tile
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * t + i;
            float o[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) o[s] = RAW ? acch[s][t][i] : pw4_epi(acch[s][t][i], ep[4 * r], ep[4 * r + 1], ep[4 * r + 2]);
            if constexpr (STATS) {
              if (do_stats) {   // sums of the values as they are stored (rounded to bfloat16)
                const float2 u0 = csn_bufacc<csn_bf16>::cv2(csn_pack_bf2(o[0], o[1])), u1 = csn_bufacc<csn_bf16>::cv2(csn_pack_bf2(o[2], o[3]));
                const float s1 = (u0.x + u0.y) + (u1.x + u1.y);
                const float s2 = fmaf(u0.x, u0.x, u0.y * u0.y) + fmaf(u1.x, u1.x, u1.y * u1.y);
                stv[2 * i] = valid ? s1 : 0.f; stv[2 * i + 1] = valid ? s2 : 0.f;
              }
            }
            if (RED) {   // fused 1x1 consumer: accumulate red_w[row] * y per quad pixel, nothing is stored here
              const float rw = csn_const(a->red_w)[r0 + r];
#pragma unroll
              for (int s = 0; s < 4; ++s) red[s] = fmaf(rw, o[s], red[s]);
              continue;
            }
            const unsigned so = (unsigned)(r0 + r) * csh;
            csn_bufacc<AT>::st2(ob, sv0, so, make_float2(o[0], o[1]));
            csn_bufacc<AT>::st2(ob, sv1, so, make_float2(o[2], o[3]));
          }
          if constexpr (STATS) {
            if (do_stats) pw4_stats_tile(a->stats_h, r0 + 4 * t, a->OH, a->stats_stride, tile, stv, lane);
          }
        }
      }
    }
    if (NTL > 0) {
      const int r0 = a->grp[g].r0l, nt = a->grp[g].ntl;
      const csn_buf ob = csn_make_buf_n(reinterpret_cast<char*>(a->yl) + (int64_t)b * a->OL * (int64_t)csl, (unsigned)a->OL * csl);
      csn_cfp ep = csn_const(a->ep_l) + 4 * r0;
#ifdef CSN_KO_PW4_NOSTORE
      const unsigned sv = 0x80000000u;
#else
      const unsigned sv = valid ? ol[4] : 0x80000000u;
#endif
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        if (t < nt) {
          float stv[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * t + i;
            const float o = RAW ? accl[t][i] : pw4_epi(accl[t][i], ep[4 * r], ep[4 * r + 1], ep[4 * r + 2]);
            if constexpr (STATS) {
              if (do_stats) {
                const float u = csn_bufacc<csn_bf16>::cv1((unsigned)csn_f2bf(o));
                stv[2 * i] = valid ? u : 0.f; stv[2 * i + 1] = valid ? u * u : 0.f;
              }
            }
            csn_bufacc<AT>::st1(ob, sv, (unsigned)(r0 + r) * csl, o);
          }
          if constexpr (STATS) {
            if (do_stats) pw4_stats_tile(a->stats_l, r0 + 4 * t, a->OL, a->stats_stride, tile, stv, lane);
          }
        }
      }
    }
    }   // groups of the tile
    if (RED && NTH > 0) {   // logits: [B][1][2 Hl][2 Wl]
      const csn_buf lb = csn_make_buf_n(a->logits + (int64_t)b * (int64_t)(csh >> 2), csh);
      const float rb0 = csn_const(a->red_b)[0];
      csn_st2(lb, sv0, 0u, make_float2(red[0] + rb0, red[1] + rb0));
      csn_st2(lb, sv1, 0u, make_float2(red[2] + rb0, red[3] + rb0));
    }
  }
}

// ---- instantiation table -----------------------------------------------------------------------------------------
// (5, 5), round 6: the un-pruned expand-2 net's 80 + 80 -> 80 + 80 units (four M groups; without it they fell back to goct_pw_kernel)
#ifdef PW4_NO55
#define PW4_X55(X)
#else
#define PW4_X55(X) X(5, 5)
#endif
#define PW4_INST_LIST(X) \
  X(3, 3) X(3, 4) X(4, 3) X(4, 4) X(5, 3) X(4, 6) X(4, 0) X(5, 0) X(6, 0) X(2, 2) X(3, 0) X(2, 0) X(1, 1) X(5, 2) X(2, 5) \
  PW4_X55(X) X(1, 0) X(0, 1) X(0, 2) X(0, 3) X(0, 4) X(0, 5) X(0, 6) X(0, 8) X(0, 10) X(8, 0) X(10, 0) X(7, 0) X(2, 3) X(3, 2) X(2, 1) X(1, 2) X(1, 3) X(2, 4)

typedef void (*Pw4Fn)(Pw4Args);
struct Pw4Entry { int nth, ntl; Pw4Fn fn[4]; };   // BN + PReLU / raw / row reduction / raw with bfloat16 tensors
template <int H, int L> struct Pw4RedFn { static Pw4Fn get() { return nullptr; } };
template <int H> struct Pw4RedFn<H, 0> { static Pw4Fn get() { return pw4_kernel<H, 0, 2>; } };   // row reduction: high-only forms
#define PW4_ENTRY(H, L) {H, L, {pw4_kernel<H, L, 0>, pw4_kernel<H, L, 1>, Pw4RedFn<H, L>::get(), pw4_kernel<H, L, 1, csn_bf16>}},
static const Pw4Entry g_pw4_table[] = {PW4_INST_LIST(PW4_ENTRY)};

// smallest instantiation that covers (nth, ntl) row tiles per group (ntl = 0 must stay 0: no low output), or {0, 0}
bool csn_pw4_has_stats(int nth, int ntl) { (void)nth; return ntl > 0; }   // the forms whose bf16 raw launches carry Pw4Args::stats_*

bool csn_pw4_pick(int nth, int ntl, int* pnth, int* pntl) {
  int best = -1, best_cost = 1 << 30;
  for (size_t i = 0; i < sizeof(g_pw4_table) / sizeof(g_pw4_table[0]); ++i) {
    const Pw4Entry& e = g_pw4_table[i];
    if (e.nth < nth || e.ntl < ntl || (ntl == 0) != (e.ntl == 0) || (nth == 0) != (e.nth == 0)) continue;
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
  const int mode = a.red_w ? 2 : (raw ? (a.a16 ? 3 : 1) : 0);
  if (a.a16 && !raw) return 1;   // bfloat16 tensors: train-mode (raw) launches only
  if (!e->fn[mode]) return 1;
  const int nitems = a.tiles_x * a.tiles_y * a.B * (a.red_w ? 1 : a.ngroups);
  int nblk = (nitems + 3) / 4;
  if (nblk > a.max_grid) nblk = a.max_grid;
  const dim3 grid((nblk + 7) & ~7);
  const size_t lds = (size_t)a.ngroups * a.gimg_floats * sizeof(float);
#ifndef CSN_CPU_EMU
  if (lds > 64 * 1024) {   // once per device and function
    static CsnPerDeviceOnce once[sizeof(g_pw4_table) / sizeof(g_pw4_table[0])][4];
    const int st = once[e - g_pw4_table][mode].run([&]() {
      return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(e->fn[mode]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (st != 0) return st;
  }
#endif
#ifdef CSN_EMU_SEQ   // pw4_stats_out's stand-in accumulates
  if (a.stats_h) std::memset(a.stats_h, 0, (size_t)a.OH * a.stats_stride * 2 * sizeof(double));
  if (a.stats_l) std::memset(a.stats_l, 0, (size_t)a.OL * a.stats_stride * 2 * sizeof(double));
#endif
  if ((a.stats_h || a.stats_l) && (mode != 3 || !csn_pw4_has_stats(a.nth, a.ntl) || a.stats_stride != a.tiles_x * a.tiles_y * a.B)) return 1;
  CSN_LAUNCH(e->fn[mode], grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
