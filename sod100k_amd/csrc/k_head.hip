// k_head.hip -- the HIGH output of a three-branch 1x1 gOctaveCBR (CSFHead.fuse / fuse1x1, CSNet/model/csnet.py:152-206) with the
// low -> high terms in the REFERENCE's order: convolve at the low resolution, then interpolate (csnet.py:702-707).  Round 6.
//
//   y_h = W_h0 x_0 + bilinear_up2(W_h1 x_1) + bilinear_up4(W_h2 x_2)        x_0: H0 x W0, x_1: H0/2 x W0/2, x_2: H0/4 x W0/4
//   out = PReLU(BN(y_h))                                                     (gOctaveCBR 778-792; eval BN folded to scale / shift)
//
// pw4_kernel's high-only form (k_pw4.hip) contracts INTERPOLATED inputs at the output resolution (a 1x1 convolution commutes with the
// interpolation): for `fuse` that is 153 gathered channels x 4 pixels x NTH matrix instructions per lane, nine gathers and a dozen
// vector instructions per low / third-input channel, every batch of two channels one exposed memory round trip: 115 us for 262 MB
// (profiles/r5_unit_table.md: 0.175 of the HBM peak, 2.4 x the reference's FLOPs).  Here (VERDICT r5 #1a, the scheme of k_ilb.hip's
// phase 1 carried to the large maps):
//   * work item = (image, band of RB rows of x_2 = 4 RB output rows, M group of NTH row tiles); block = 4 waves.
//   * phase A: z_1 = W_h1 x_1 over the band's 2 RB + 2 rows of x_1 and z_2 = W_h2 x_2 over its RB + 2 rows of x_2 (one halo row each
//     side: the bilinear taps), lane = one pixel, one dword load per channel in batches of eight, NTH matrix instructions per channel
//     (v_mfma_f32_4x4x1, the loaded value is the B operand) -> LDS planes [row of the group][band row][column], with the clamped
//     border taps of the interpolation REPLICATED into a frame of one column / row, so that phase B reads fixed offsets.
//   * phase B: lane = one pixel of x_1's grid + its 2x2 output quad (k_pw4.hip): contraction over x_0 only (two 64-bit loads per
//     channel); then per OUTPUT channel: the 3x3 neighbourhood of z_1 (x2: weights 0.75 / 0.25) and the 2x2 neighbourhood of z_2 the
//     lane's parity selects (x4: 0.375 / 0.625 / 0.125 / 0.875) from LDS, packed interpolation arithmetic, BN + PReLU
//     (multiply + v_med3, dw_core.h), 64-bit stores -- or, for fuse1x1, the group's share of the cls_layer sum
//     red_w[r] * out[r] (csnet.py:306-308,381) into a per-group partial plane that the final upsample adds up.
// Matrix work per output quad: CH x 4 NTH + C1 x NTH + C2 x NTH / 4 instead of (CH + C1 + C2) x 4 NTH; interpolation per OUTPUT
// channel instead of per gathered INPUT channel; x_1 / x_2 are read (1 + 1/RB) / (1 + 2/RB) times (halo rows, L2 hits).
#include "pw4_common.h"
#include "dw_core.h"

#ifndef HZ_ZB
#define HZ_ZB 8    // x_1 / x_2 channels per load batch of phase A (one dword each); two batches in flight
#endif

// knock-out builds (tools/README.md; results wrong, timings valid): HZ_KO_NOA = no phase A, HZ_KO_NOMAIN = no contraction over x_0,
// HZ_KO_NOEPI = no interpolation in the epilogue

namespace {

// block-wide copy of the group's weight image into LDS (csn_fill_lds16 for any block size)
template <int NT>
__device__ __forceinline__ void hz_fill(float* lds, const float* __restrict__ img, int n4, int tid) {
  const float4* __restrict__ src = reinterpret_cast<const float4*>(img);
  float4* dst = reinterpret_cast<float4*>(lds);
  constexpr int U = NT >= 512 ? 2 : 4;
  for (int i0 = tid; i0 < n4; i0 += U * NT) {
    float4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = src[min(i0 + j * NT, n4 - 1)];
#ifndef CSN_CPU_EMU
#pragma unroll
    for (int j = 0; j < U; ++j) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j].z), "+v"(v[j].w));
#endif
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i0 + j * NT < n4) dst[i0 + j * NT] = v[j];
  }
}

template <int ZB>
__device__ __forceinline__ void hz_load(csn_buf rb, unsigned o, unsigned cs, int k0, int C, float (&v)[ZB]) {
#pragma unroll
  for (int j = 0; j < ZB; ++j) v[j] = csn_ld1(rb, o, (unsigned)min(k0 + j, C - 1) * cs);
}

template <int NTH, int P, int ZB, bool GUARD>
__device__ __forceinline__ void hz_zbatch(const float (&v)[ZB], const float* wk, int n, csn_f4 (&acc)[NTH]) {
  constexpr int NT4 = (NTH + 3) & ~3;
#pragma unroll
  for (int j = 0; j < ZB; ++j) {
    if (GUARD && j >= n) break;
    Pw4A<NT4> a;
    pw4_load_a<NT4, P>(wk + j * 4 * P, a);
#pragma unroll
    for (int t = 0; t < NTH; ++t) pw4_mfma<NT4>(a, t, v[j], acc[t]);
  }
}

// one phase-A task: z = W x over a tile of 64 consecutive pixels of a band region (rows [r_lo, r_hi] of a [C][Hs][Ws] tensor).
// Everything but the lane's pixel is block-uniform.
struct HzZTask {
  csn_buf rb;
  unsigned cs;
  int C, woff;              // channels, image row (gathered channel index) of the input's first channel
  int Hs, Ws, r_lo, r_hi;
  int base, rows;           // image row of band row 0; band rows of the plane
  float* dst;               // LDS plane of the group's row 0: [rows][pitch], column 0 = image column -1
  int pitch, plane;
};
struct HzZLane { bool on; int row, col; unsigned off; };

__device__ __forceinline__ HzZLane hz_zlane(const HzZTask& z, int p) {
  const int npix = (z.r_hi - z.r_lo + 1) * z.Ws;
  HzZLane l;
  l.on = p < npix;
  const int pc = min(p, npix - 1);
  int q = (int)((float)pc * (1.0f / (float)z.Ws));
  q -= (q * z.Ws > pc) ? 1 : 0;
  q += ((q + 1) * z.Ws <= pc) ? 1 : 0;
  l.row = z.r_lo + q; l.col = pc - q * z.Ws;
  l.off = (unsigned)(l.row * z.Ws + l.col) * 4u;
  return l;
}
// the task's first two load batches (issued ahead: before the weight image is staged for a wave's first task)
__device__ __forceinline__ void hz_zissue(const HzZTask& z, const HzZLane& l, float (&vA)[HZ_ZB], float (&vB)[HZ_ZB]) {
  hz_load<HZ_ZB>(z.rb, l.off, z.cs, 0, z.C, vA);
  if (z.C > HZ_ZB) hz_load<HZ_ZB>(z.rb, l.off, z.cs, HZ_ZB, z.C, vB);
}
// contraction (A = batch 0, B = batch 1 on entry; two batches in flight throughout) and the planes' stores; `zr` = rows of the group
template <int NTH, int P>
__device__ __forceinline__ void hz_zrun(const HzZTask& z, const HzZLane& l, const float* wl, int zr, float (&vA)[HZ_ZB], float (&vB)[HZ_ZB]) {
  constexpr int ZB = HZ_ZB;
  csn_f4 acc[NTH];
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
  const float* wk = wl + z.woff * 4 * P;
  const int C = z.C, nf = (C - 1) / ZB;   // full batches in front of the last one
  int c0 = 0;
  PW4_FENCE();
  for (int it = 0; it < (nf >> 1); ++it) {
    hz_zbatch<NTH, P, ZB, false>(vA, wk + c0 * 4 * P, ZB, acc);
    hz_load<ZB>(z.rb, l.off, z.cs, c0 + 2 * ZB, C, vA);
    PW4_FENCE();
    hz_zbatch<NTH, P, ZB, false>(vB, wk + (c0 + ZB) * 4 * P, ZB, acc);
    if (c0 + 3 * ZB < C) hz_load<ZB>(z.rb, l.off, z.cs, c0 + 3 * ZB, C, vB);
    PW4_FENCE();
    c0 += 2 * ZB;
  }
  if (nf & 1) {
    hz_zbatch<NTH, P, ZB, false>(vA, wk + c0 * 4 * P, ZB, acc);
    c0 += ZB;
#pragma unroll
    for (int j = 0; j < ZB; ++j) vA[j] = vB[j];
  }
  hz_zbatch<NTH, P, ZB, true>(vA, wk + c0 * 4 * P, C - c0, acc);
  if (!l.on) return;
  float* d = z.dst + (l.row - z.base) * z.pitch + l.col + 1;
  const int dc = l.col == 0 ? -1 : (l.col == z.Ws - 1 ? 1 : 0);            // replicated column of the frame
  // ... row (the image's last row can also be the bottom halo row of the band in front of the last: no row behind it in the plane)
  // (likewise the image's first row can be the TOP halo row of the second band when a band is one row: nothing in front of it)
  const int dr = (l.row == 0 && z.base < 0) ? -z.pitch : ((l.row == z.Hs - 1 && l.row - z.base < z.rows - 1) ? z.pitch : 0);
  // (every value goes to its own column and to the frame column -- the same address when the lane has none: no branch; the frame
  // rows are rare and block-uniform per tile row)
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * t + i < zr) {
        float* e = d + (4 * t + i) * z.plane;
        const float v = acc[t][i];
        e[0] = v;
        e[dc] = v;
        if (dr) {
          e[dr] = v;
          e[dr + dc] = v;
        }
      }
    }
}

}  // namespace

#define HZ_MINBLK(NTH, NW) ((NW) == 4 ? ((NTH) <= 3 ? 4 : 2) : ((NW) == 8 ? ((NTH) <= 3 ? 2 : 1) : 1))

// MODE 0: BN + PReLU, rows stored;  MODE 2 (RED): BN + PReLU, the group's rows reduced with red_w into part[g]
// HB: x_0 channels per load batch of phase B (two batches in flight);  NW: waves per block
template <int NTH, int MODE, int HB, int NW>
__global__ __launch_bounds__(64 * NW, HZ_MINBLK(NTH, NW)) void hz_kernel(HzArgs a_byval) {
  constexpr bool RED = MODE == 2;
  constexpr int NT4 = (NTH + 3) & ~3, P = PW4_PITCH(NT4);
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS HzArgs* a = CSN_KERNARG(HzArgs, a_byval);
  const int tid = threadIdx.x;
  // items in XCD-aware order (k_pw4.hip): XCD x = blockIdx.x & 7 walks the item range [x chunk, (x + 1) chunk); the M groups of a band
  // and the neighbouring bands of an image (shared inputs / halo rows) sit in one L2
  const int ng = a->ngroups, nbands = a->nbands;
  const int nitems = a->B * nbands * ng;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;
  const int xcd = blockIdx.x & 7;
  const int item = xcd * chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunk || item >= nitems) return;
  const int g = item % ng, bb = item / ng;
  const int band = bb % nbands, b = bb / nbands;
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  const int CH = a->CH, C1 = a->C1, C2 = a->C2, H1 = a->H1, W1 = a->W1, H2 = H1 >> 1, W2 = W1 >> 1, W0 = 2 * W1;
  const int RB = a->RB;
  const unsigned cs1 = (unsigned)(H1 * W1) * 4u, cs0 = cs1 * 4u, cs2 = cs1 >> 2;
  float* Z1 = lds + a->off_z1;
  float* Z2 = lds + a->off_z2;
  const int pitch1 = a->pitch1, plane1 = a->plane1, pitch2 = a->pitch2, plane2 = a->plane2;
  const int base1 = 2 * RB * band - 1, base2 = RB * band - 1;   // image row of band row 0 of the z planes
  const int r0 = 4 * NTH * g;                        // first output channel of the group
  const int nr = max(0, min(4 * NTH, a->OH - r0));   // ... and its channel count
#ifdef CSN_EMU_SEQ
  const float* wl = lds;
#else
  const float* wl = lds + (lane & 3) * P;
#endif
  // phase-A tasks: z_1 over rows [base1, base1 + 2 RB + 1], z_2 over rows [base2, base2 + RB + 1] (clipped to the image)
  HzZTask z1, z2;
  z1.rb = csn_make_buf_n(reinterpret_cast<const char*>(a->x1) + (int64_t)b * C1 * (int64_t)cs1, (unsigned)C1 * cs1);
  z1.cs = cs1; z1.C = C1; z1.woff = CH; z1.Hs = H1; z1.Ws = W1; z1.r_lo = max(base1, 0); z1.r_hi = min(base1 + 2 * RB + 1, H1 - 1);
  z1.base = base1; z1.rows = 2 * RB + 2; z1.dst = Z1; z1.pitch = pitch1; z1.plane = plane1;
  z2.rb = csn_make_buf_n(reinterpret_cast<const char*>(a->x2) + (int64_t)b * C2 * (int64_t)cs2, (unsigned)C2 * cs2);
  z2.cs = cs2; z2.C = C2; z2.woff = CH + C1; z2.Hs = H2; z2.Ws = W2; z2.r_lo = max(base2, 0); z2.r_hi = min(base2 + RB + 1, H2 - 1);
  z2.base = base2; z2.rows = RB + 2; z2.dst = Z2; z2.pitch = pitch2; z2.plane = plane2;
  const int nt1 = ((z1.r_hi - z1.r_lo + 1) * W1 + 63) >> 6, nt2 = ((z2.r_hi - z2.r_lo + 1) * W2 + 63) >> 6;
  // phase-B tasks: tiles of 64 consecutive pixels of x_1's grid inside the band's own rows
  const int y_first = 2 * RB * band, nrow = min(2 * RB, H1 - y_first);
  const int npix = nrow * W1;
  const csn_buf rbh = csn_make_buf_n(reinterpret_cast<const char*>(a->xh) + (int64_t)b * CH * (int64_t)cs0, (unsigned)CH * cs0);
  const int nfh = (CH - 1) / HB;   // full batches in front of the last one
  struct MLane { bool valid; int y, x; unsigned oh0, oh1; };
  auto mlane = [&](int t) {
    MLane m;
    const int p = t * 64 + lane;
    m.valid = p < npix;
    const int pc = min(p, npix - 1);
    int q = (int)((float)pc * (1.0f / (float)W1));
    q -= (q * W1 > pc) ? 1 : 0;
    q += ((q + 1) * W1 <= pc) ? 1 : 0;
    m.y = y_first + q; m.x = pc - q * W1;
    m.oh0 = (unsigned)((2 * m.y) * W0 + 2 * m.x) * 4u; m.oh1 = m.oh0 + (unsigned)W0 * 4u;
    return m;
  };

  // ---- the first loads of the wave's first tasks of both phases are issued BEFORE the weight image is staged: an item is a chain
  // of memory round trips (weights -> z inputs -> barrier -> x_0 -> stores), every one taken off it counts (k_ilb.hip) ----
  float vA[HZ_ZB], vB[HZ_ZB];
  float2 hA[HB][2], hB[HB][2];
  HzZLane zl;
#ifndef HZ_KO_NOA
  if (wave < nt1 + nt2) {
    zl = hz_zlane(wave < nt1 ? z1 : z2, (wave < nt1 ? wave : wave - nt1) * 64 + lane);
    hz_zissue(wave < nt1 ? z1 : z2, zl, vA, vB);
  }
#endif
  MLane ml = mlane(wave);
#ifndef HZ_KO_NOMAIN
  if (wave * 64 < npix) {
    pw4_load_hi<HB>(rbh, ml.oh0, ml.oh1, cs0, 0, CH, hA);
    if (nfh >= 1) pw4_load_hi<HB>(rbh, ml.oh0, ml.oh1, cs0, HB, CH, hB);
  }
#endif
  PW4_FENCE();
  hz_fill<64 * NW>(lds, a->wimg + (int64_t)g * a->gimg_floats, a->gimg_floats >> 2, tid);
  __syncthreads();

  // ---- phase A: z_1, z_2 of the band (+ one halo row each side) -> LDS ----
#ifndef HZ_KO_NOA
  for (int t = wave; t < nt1 + nt2; t += NW) {
    const bool one = t < nt1;
    if (t != wave) {
      zl = hz_zlane(one ? z1 : z2, (one ? t : t - nt1) * 64 + lane);
      hz_zissue(one ? z1 : z2, zl, vA, vB);
    }
    hz_zrun<NTH, P>(one ? z1 : z2, zl, wl, nr, vA, vB);
  }
#endif
  __syncthreads();

  // ---- phase B: contraction over x_0, interpolated z added per output channel, epilogue ----
  for (int t = wave; t * 64 < npix; t += NW) {
    if (t != wave) {
      ml = mlane(t);
#ifndef HZ_KO_NOMAIN
      pw4_load_hi<HB>(rbh, ml.oh0, ml.oh1, cs0, 0, CH, hA);
      if (nfh >= 1) pw4_load_hi<HB>(rbh, ml.oh0, ml.oh1, cs0, HB, CH, hB);
#endif
    }
    const int y = ml.y, x = ml.x;
    const unsigned oh0 = ml.oh0, oh1 = ml.oh1;
    csn_f4 acc[4][NTH];
#pragma unroll
    for (int tt = 0; tt < NTH; ++tt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][tt][i] = 0.f;
#ifndef HZ_KO_NOMAIN
    {   // A = batch 0, B = batch 1 on entry; two batches in flight throughout
      csn_f4 dummy[1];
      PW4_FENCE();
      int k0 = 0;
      for (int it = 0; it < (nfh >> 1); ++it) {
        pw4_hi_batch<NTH, 0, HB, P, false>(hA, wl + k0 * 4 * P, HB, acc, dummy);
        pw4_load_hi<HB>(rbh, oh0, oh1, cs0, k0 + 2 * HB, CH, hA);
        PW4_FENCE();
        pw4_hi_batch<NTH, 0, HB, P, false>(hB, wl + (k0 + HB) * 4 * P, HB, acc, dummy);
        if (k0 + 3 * HB < CH) pw4_load_hi<HB>(rbh, oh0, oh1, cs0, k0 + 3 * HB, CH, hB);
        PW4_FENCE();
        k0 += 2 * HB;
      }
      if (nfh & 1) {
        pw4_hi_batch<NTH, 0, HB, P, false>(hA, wl + k0 * 4 * P, HB, acc, dummy);
        k0 += HB;
#pragma unroll
        for (int j = 0; j < HB; ++j) { hA[j][0] = hB[j][0]; hA[j][1] = hB[j][1]; }
      }
      pw4_hi_batch<NTH, 0, HB, P, true>(hA, wl + k0 * 4 * P, CH - k0, acc, dummy);
    }
#endif
    const bool valid = ml.valid;
    // interpolation geometry of the lane.  x2 (upsample_bilinear2d, align_corners=False): quad pixel (dy, dx) takes 0.75 of the centre
    // row / column of z_1 and 0.25 of the row above (dy = 0) / below (dy = 1), the column left (dx = 0) / right (dx = 1).
    // x4: output row Y = 2y + dy has source (Y + 0.5) / 4 - 0.5 = parent + {-0.375, -0.125, +0.125, +0.375} for Y & 3 = 0 .. 3: rows
    // (parent - 1, parent) with weights (0.375, 0.625) / (0.125, 0.875) for even y, (parent, parent + 1) with (0.875, 0.125) /
    // (0.625, 0.375) for odd y; the same along x.  Border taps: the replicated frame = PyTorch's clamped source index.
    const float* z1c = Z1 + (y - base1) * pitch1 + x + 1;
    const int oy = y & 1, ox = x & 1;
    const float* z2a = Z2 + ((y >> 1) - 1 + oy - base2) * pitch2 + ((x >> 1) - 1 + ox) + 1;
    const float wya0 = oy ? 0.875f : 0.375f, wyb0 = oy ? 0.125f : 0.625f, wya1 = oy ? 0.625f : 0.125f, wyb1 = oy ? 0.375f : 0.875f;
    const csn_v2 wxa = csn_mk2(ox ? 0.875f : 0.375f, ox ? 0.625f : 0.125f), wxb = csn_mk2(ox ? 0.125f : 0.625f, ox ? 0.375f : 0.875f);
    const unsigned sv0 = valid ? oh0 : 0x80000000u, sv1 = valid ? oh1 : 0x80000000u;
    const csn_buf ob = RED ? csn_make_buf_n(a->part + ((int64_t)g * a->B + b) * (int64_t)(cs0 >> 2), cs0)
                           : csn_make_buf_n(reinterpret_cast<char*>(a->yh) + (int64_t)b * a->OH * (int64_t)cs0, (unsigned)a->OH * cs0);
    csn_cfp ep = csn_const(a->ep_h) + 4 * r0;
    csn_v2 red01 = csn_mk2(0.f, 0.f), red23 = csn_mk2(0.f, 0.f);
#pragma unroll
    for (int tt = 0; tt < NTH; ++tt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * tt + i;
        if (r < nr) {
          const float* u1 = z1c + r * plane1;
          const float* u2 = z2a + r * plane2;
          csn_v2 hp[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float* row = u1 + (k - 1) * pitch1;
            const float c = 0.75f * row[0];
            hp[k] = csn_fma2(0.25f, csn_mk2(row[-1], row[1]), csn_mk2(c, c));
          }
          const csn_v2 m = csn_mul2(0.75f, hp[1]);
          const csn_v2 q01 = csn_fma2(0.25f, hp[0], m), q23 = csn_fma2(0.25f, hp[2], m);
          const float w00 = u2[0], w01 = u2[1], w10 = u2[pitch2], w11 = u2[pitch2 + 1];
          const csn_v2 ha = csn_fmav2(wxb, csn_mk2(w01, w01), csn_mul2(w00, wxa));
          const csn_v2 hb = csn_fmav2(wxb, csn_mk2(w11, w11), csn_mul2(w10, wxa));
          const csn_v2 p01 = csn_fma2(wyb0, hb, csn_mul2(wya0, ha)), p23 = csn_fma2(wyb1, hb, csn_mul2(wya1, ha));
          // (the branches are summed in input order, csnet.py:720-722)
          csn_v2 t01 = csn_mk2(acc[0][tt][i] + q01[0], acc[1][tt][i] + q01[1]);
          csn_v2 t23 = csn_mk2(acc[2][tt][i] + q23[0], acc[3][tt][i] + q23[1]);
          t01 = csn_mk2(t01[0] + p01[0], t01[1] + p01[1]);
          t23 = csn_mk2(t23[0] + p23[0], t23[1] + p23[1]);
#ifdef HZ_KO_NOEPI
          t01 = csn_mk2(acc[0][tt][i], acc[1][tt][i]); t23 = csn_mk2(acc[2][tt][i], acc[3][tt][i]);
#endif
          const float sc = ep[4 * r], sh = ep[4 * r + 1], al = ep[4 * r + 2];
          const float lim = al <= 1.f ? __builtin_inff() : -__builtin_inff();
          const csn_v2 shv = csn_mk2(sh, sh);
          const csn_v2 o01 = dw_prelu2(csn_fma2(sc, t01, shv), al, lim), o23 = dw_prelu2(csn_fma2(sc, t23, shv), al, lim);
          if (RED) {
            const float rw = csn_const(a->red_w)[r0 + r];
            red01 = csn_fma2(rw, o01, red01);
            red23 = csn_fma2(rw, o23, red23);
          } else {
            const unsigned so = (unsigned)(r0 + r) * cs0;
            csn_st2(ob, sv0, so, make_float2(o01[0], o01[1]));
            csn_st2(ob, sv1, so, make_float2(o23[0], o23[1]));
          }
        }
      }
    if (RED) {
      csn_st2(ob, sv0, 0u, make_float2(red01[0], red01[1]));
      csn_st2(ob, sv1, 0u, make_float2(red23[0], red23[1]));
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
typedef void (*HzFn)(HzArgs);
struct HzEntry { int nth, hb, nw; HzFn fn[2]; };   // rows stored / row reduction
#define HZ_ENTRY(N, HB, NW) {N, HB, NW, {hz_kernel<N, 0, HB, NW>, hz_kernel<N, 2, HB, NW>}}
#define HZ_ENTRIES(N) HZ_ENTRY(N, 2, 4), HZ_ENTRY(N, 4, 4), HZ_ENTRY(N, 2, 8), HZ_ENTRY(N, 4, 8), HZ_ENTRY(N, 2, 16), HZ_ENTRY(N, 4, 16)
static const HzEntry g_hz_table[] = {HZ_ENTRIES(1), HZ_ENTRIES(2), HZ_ENTRIES(3), HZ_ENTRIES(4), HZ_ENTRIES(5)};

// LDS layout of an item; returns its bytes (0: the geometry is not supported).  Fills pitch / plane / offsets / nbands of `a`.
size_t csn_hz_layout(HzArgs& a) {
  if (a.nth < 1 || a.nth > 5 || (a.H1 & 1) || (a.W1 & 1) || a.H1 < 2 || a.W1 < 2 || a.RB < 1) return 0;
  const int NT4 = (a.nth + 3) & ~3, P = PW4_PITCH(NT4);
  a.gimg_floats = (a.CH + a.C1 + a.C2) * 4 * P;
  const int rows1 = 2 * a.RB + 2, rows2 = a.RB + 2;
  a.pitch1 = a.W1 + 2; a.plane1 = rows1 * a.pitch1;
  a.pitch2 = (a.W1 >> 1) + 2; a.plane2 = rows2 * a.pitch2;
  const int zr = std::min(4 * a.nth, a.OH);   // planes per item: the rows of the largest group
  int off = (a.gimg_floats + 3) & ~3;
  a.off_z1 = off; off += zr * a.plane1;
  a.off_z2 = off; off += zr * a.plane2;
  a.nbands = ((a.H1 >> 1) + a.RB - 1) / a.RB;
  return (size_t)off * sizeof(float);
}

bool csn_hz_supported(int nth) { return nth >= 1 && nth <= 5; }

int csn_launch_hz(const HzArgs& a, void* stream) {
  const HzEntry* e = nullptr;
  for (const HzEntry& t : g_hz_table)
    if (t.nth == a.nth && t.hb == a.hb && t.nw == a.nw) e = &t;
  if (!e) return 1;
  HzArgs chk = a;
  const size_t lds = csn_hz_layout(chk);
  if (lds == 0 || lds > 160 * 1024 || chk.off_z1 != a.off_z1 || chk.off_z2 != a.off_z2) return 1;
  HzFn fn = e->fn[a.red_w ? 1 : 0];
#ifndef CSN_CPU_EMU
  if (lds > 64 * 1024) {
    static CsnPerDeviceOnce once[sizeof(g_hz_table) / sizeof(g_hz_table[0])][2];
    const int st = once[e - g_hz_table][a.red_w ? 1 : 0].run([&]() {
      return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (st != 0) return st;
  }
#endif
  const int ng = a.ngroups;
  const int nitems = a.B * a.nbands * ng;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;
  CSN_LAUNCH(fn, dim3(8 * chunk), dim3(64 * a.nw), lds, stream, a);
  return (int)hipGetLastError();
}
