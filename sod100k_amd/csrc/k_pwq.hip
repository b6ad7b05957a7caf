// k_pwq.hip -- plain 1x1 contraction over own-resolution slices, raw output: the input-gradient launches of the 1x1 units.
//
// Backward of gOctaveConv with 1x1 kernels (CSNet/model/csnet.py:664-726, adjoint of 716-717 / 702-707 / 708-714):
//     dx_i = sum_{j = i} W_ij^T dz_j  +  sum_{j < i} W_ij^T adjoint_up(dz_j)  +  sum_{j > i} maxpool_bwd(W_ij^T dz_j)
// csn_backward.inl turns every term into a contraction whose sources are ALL at the pass resolution (dz_j itself, the
// adjoint-upsampled dz_j written by adjup*_kernel, the low-resolution temporary that maxpool*_bwd_add routes afterwards):
//     out[r][p] = sum_s sum_c Wt[r][k_s + c] * src_s[c][p],   Wt = transposed weight blocks (CSN_PREP_PW4_T)
// -- no resampling, no epilogue.  A 1x1 convolution does not see the image geometry, so the planes are walked as flat
// arrays: a LANE owns four consecutive elements of the plane (one aligned 128-bit load per channel, a wave reads 1 KB
// rows), v_mfma_f32_4x4x1_16B_f32 straight from the load registers exactly as in k_pw4.hip (B = the lane's value, A = four
// rows of Wt from the LDS image, D = four consecutive output channels of the lane's own element), 128-bit stores.
// Replaces goct_pw_kernel<true> for these launches (round 3): ~3,000 instructions per 64 elements there, ~25 per channel
// and 256 elements here; float and bfloat16 storage.  Needs H * W % 4 == 0 (else the launch stays on goct_pw_kernel).
#include <cstdlib>

#include "pw4_common.h"

#ifndef PWQ_CB
#define PWQ_CB 4     // channels per load batch
#endif

namespace {

template <typename AT, int CB>
__device__ __forceinline__ void pwq_load(csn_buf rb, unsigned off, unsigned cs, int c0, int C, typename csn_bufacc<AT>::r4 (&v)[CB]) {
#pragma unroll
  for (int j = 0; j < CB; ++j) v[j] = csn_bufacc<AT>::ldr4(rb, off, (unsigned)min(c0 + j, C - 1) * cs);   // raw: converted in pwq_batch
}

template <int NT, int P, int CB, bool GUARD, typename AT = float>
__device__ __forceinline__ void pwq_batch(const typename csn_bufacc<AT>::r4 (&raw)[CB], const float* wk, int n, csn_f4 (&acc)[4][NT]) {
  constexpr int NT4 = (NT + 3) & ~3;
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    if (GUARD && j >= n) break;
    Pw4A<NT4> a;
    pw4_load_a<NT4, P>(wk + j * 4 * P, a);
    const float4 v = csn_bufacc<AT>::cv4(raw[j]);
    const float q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) pw4_mfma<NT4>(a, t, q[s], acc[s][t]);
  }
}

// ---- the adjoint of a 2x2 max-pool routed in the epilogue (PwqArgs::route_x / route_t, round 6) ----
// The lane's four elements are columns x .. x + 3 of row y of the plane (x % 4 == 0): the own halves of the windows (y >> 1, x >> 1)
// and (y >> 1, (x >> 1) + 1).  Per output channel it reads its four values of the pooled tensor, the four of the window's other
// row and the two low-resolution gradients, and adds a gradient where the window's FIRST maximum (row-major scan, `v > best ||
// v != v` -- max_pool2d's rule, as maxpool2_bwd_add_pair_kernel) is one of its own elements.
struct PwqRouteGeo {
  unsigned off_n, off_t;   // byte offsets inside a channel plane: the other row's four elements / the gradient pair
  bool odd;                // the own row is the window's second row
};
template <typename AT>
__device__ __forceinline__ PwqRouteGeo pwq_route_geo(int e0, int W) {
  constexpr unsigned E = (unsigned)sizeof(AT);
  PwqRouteGeo g;
  const int y = e0 / W, x = e0 - y * W;
  g.odd = (y & 1) != 0;
  g.off_n = (unsigned)(e0 + (g.odd ? -W : W)) * E;
  g.off_t = (unsigned)((y >> 1) * (W >> 1) + (x >> 1)) * E;
  return g;
}
__device__ __forceinline__ void pwq_route_window(float& d0, float& d1, float a0, float a1, float b0, float b1, float t, bool odd) {
  const float w0 = odd ? b0 : a0, w1 = odd ? b1 : a1, w2 = odd ? a0 : b0, w3 = odd ? a1 : b1;   // scan order: top row first
  float best = w0; int bi = 0;
  if (w1 > best || w1 != w1) { best = w1; bi = 1; }
  if (w2 > best || w2 != w2) { best = w2; bi = 2; }
  if (w3 > best || w3 != w3) { best = w3; bi = 3; }
  const int own = bi - (odd ? 2 : 0);
  if (own == 0) d0 += t;
  if (own == 1) d1 += t;
}
// one row tile (four output channels) of a lane: loads first, then the four routings, then the stores
template <typename AT>
__device__ __forceinline__ void pwq_store_tile(csn_buf ob, csn_buf xb, csn_buf tb, unsigned sv, unsigned off, const PwqRouteGeo& rg,
                                               bool route, int row0, unsigned cs, const csn_f4 (&acc)[4]) {
  typename csn_bufacc<AT>::r4 xo[4], xn[4];
  typename csn_bufacc<AT>::r2 tv[4];
  if (route) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned so = (unsigned)(row0 + i) * cs;
      xo[i] = csn_bufacc<AT>::ldr4(xb, off, so);
      xn[i] = csn_bufacc<AT>::ldr4(xb, rg.off_n, so);
      tv[i] = csn_bufacc<AT>::ldr2(tb, rg.off_t, so >> 2);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 o = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
    if (route) {
      const float4 a = csn_bufacc<AT>::cv4(xo[i]), n = csn_bufacc<AT>::cv4(xn[i]);
      const float2 t = csn_bufacc<AT>::cv2(tv[i]);
      pwq_route_window(o.x, o.y, a.x, a.y, n.x, n.y, t.x, rg.odd);
      pwq_route_window(o.z, o.w, a.z, a.w, n.z, n.w, t.y, rg.odd);
    }
    csn_bufacc<AT>::st4(ob, sv, (unsigned)(row0 + i) * cs, o);   // (bfloat16: one 64-bit store)
  }
}

}  // namespace

template <int NT, typename AT>
__global__ __launch_bounds__(CSN_BLOCK, 16 * NT <= 96 ? 3 : 2) void pwq_kernel(PwqArgs a_byval) {
  constexpr int NT4 = (NT + 3) & ~3, P = PW4_PITCH(NT4), CB = PWQ_CB;
  constexpr unsigned E = (unsigned)sizeof(AT);
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS PwqArgs* a = CSN_KERNARG(PwqArgs, a_byval);
  const int tid = threadIdx.x;
  csn_fill_lds16(lds, a->wimg, (a->ngroups * a->gimg_floats) >> 2, tid);
  __syncthreads();
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  const int hw = a->HW, nq = hw >> 2;                 // elements / lane quads per plane
  const unsigned cs = (unsigned)hw * E;
  const int ng = a->ngroups;
  const int tiles_img = (nq + 63) >> 6;
  const int nitems = tiles_img * a->B * ng;
  const int nslot = (int)(gridDim.x >> 3) * 4;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;
  const int xcd = blockIdx.x & 7;
  const int iend = min((xcd + 1) * chunk, nitems);
#ifdef CSN_EMU_SEQ
  const float* wl_lane = lds;
#else
  const float* wl_lane = lds + (lane & 3) * P;
#endif
  for (int item = xcd * chunk + (int)(blockIdx.x >> 3) * 4 + wave; item < iend; item += nslot) {
    const int tile = item / ng, g = item - tile * ng;
    const int b = tile / tiles_img, t = tile - b * tiles_img;
    const int q0 = (t << 6) + lane;
    const bool valid = q0 < nq;
    const unsigned off = (unsigned)min(q0, nq - 1) * 4u * E;
    const float* wg = wl_lane + g * a->gimg_floats;
    csn_f4 acc[4][NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][tt][i] = 0.f;
    int krow = 0;
    for (int s = 0; s < a->nsrc; ++s) {
      const int C = a->src[s].C;
      const csn_buf rb = csn_make_buf_n(reinterpret_cast<const char*>(a->src[s].ptr) + (int64_t)b * a->src[s].Ctot * (int64_t)cs,
                                        (unsigned)C * cs);
      // batch c0 is contracted while batch c0 + CB is in flight (two register sets, batches walked in pairs, the last --
      // possibly partial -- batch ends up in set A)
      typename csn_bufacc<AT>::r4 vA[CB], vB[CB];
      pwq_load<AT, CB>(rb, off, cs, 0, C, vA);
      PW4_FENCE();
      const int nf = (C - 1) / CB;
      int c0 = 0;
      for (int p = 0; p < (nf >> 1); ++p) {
        pwq_load<AT, CB>(rb, off, cs, c0 + CB, C, vB);
        PW4_FENCE();
        pwq_batch<NT, P, CB, false, AT>(vA, wg + (krow + c0) * 4 * P, CB, acc);
        pwq_load<AT, CB>(rb, off, cs, c0 + 2 * CB, C, vA);
        PW4_FENCE();
        pwq_batch<NT, P, CB, false, AT>(vB, wg + (krow + c0 + CB) * 4 * P, CB, acc);
        c0 += 2 * CB;
      }
      if (nf & 1) {
        pwq_load<AT, CB>(rb, off, cs, c0 + CB, C, vB);
        PW4_FENCE();
        pwq_batch<NT, P, CB, false, AT>(vA, wg + (krow + c0) * 4 * P, CB, acc);
        c0 += CB;
#pragma unroll
        for (int j = 0; j < CB; ++j) vA[j] = vB[j];
      }
      pwq_batch<NT, P, CB, true, AT>(vA, wg + (krow + c0) * 4 * P, C - c0, acc);
      krow += C;
    }
    // ---- raw stores: rows past the group's tile list are skipped, rows past the last channel fall out of the resource ----
    const int r0 = a->grp_r0[g], nt = a->grp_nt[g];
    const csn_buf ob = csn_make_buf_n(reinterpret_cast<char*>(a->out) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->nrows * cs);
    const unsigned sv = valid ? off : 0x80000000u;
    const bool route = a->route_x != nullptr;   // (uniform)
    PwqRouteGeo rg{0u, 0u, false};
    csn_buf xb = ob, tb = ob;
    if (route) {
      rg = pwq_route_geo<AT>(min(q0, nq - 1) * 4, a->route_W);
      xb = csn_make_buf_n(reinterpret_cast<const char*>(a->route_x) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->nrows * cs);
      tb = csn_make_buf_n(reinterpret_cast<const char*>(a->route_t) + (int64_t)b * a->out_ctot * (int64_t)(cs >> 2), (unsigned)a->nrows * (cs >> 2));
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      if (tt < nt) {
        const csn_f4 at[4] = {acc[0][tt], acc[1][tt], acc[2][tt], acc[3][tt]};
        pwq_store_tile<AT>(ob, xb, tb, sv, off, rg, route, r0 + 4 * tt, cs, at);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// pwq16_kernel (round 4): the same launches for bfloat16 tensors on v_mfma_f32_4x4x4_16B_bf16.  At batch 256 pwq_kernel<bf16> is
// bound by the fp32 matrix pipe, not by HBM: one v_mfma_f32_4x4x1 (256 MACs in 2 passes) per loaded value and row tile, 20 x 29
// rows x channels on the stage-1 passes = the fp32 MFMA peak's worth of time (profiles/r4_notes.md).  The bf16 instruction takes
// FOUR channels of the lane's element per issue (1024 MACs in the same 2 passes):
//   B = {dz[c0 .. c0 + 3][p]} -- the lane's element p of four channels, packed from the four 64-bit loads by v_perm_b32
//       (two per element and channel pair: no conversion to float at all);
//   A = {Wt[4 t + (lane & 3)][c0 .. c0 + 3]} -- 64 bits of a bfloat16 image of the transposed weights in LDS, built by the block
//       from the fp32 image (round to nearest even) while it is filled; D as before.
// The weights of this pass are rounded to bfloat16 (8 bits of mantissa, like its other operand dz; fp32 masters, products exact,
// fp32 accumulation) -- what torch.autocast(bfloat16) does to a convolution's weight; covered by the bf16 unit-local bound (3e-2 of
// the oracle's storage emulation).  CSN_PWQ16=0 (read at plan creation): pwq_kernel<bf16> (fp32 weights, fp32 MFMA).
#ifndef PWQ16_GB
#define PWQ16_GB 1     // channel groups of four per load batch
#endif
template <int NT>
__global__ __launch_bounds__(CSN_BLOCK, 16 * NT <= 96 ? 3 : 2) void pwq16_kernel(PwqArgs a_byval) {
  constexpr int NT4 = (NT + 3) & ~3, P = PW4_PITCH(NT4), GB = PWQ16_GB;
  constexpr unsigned E = 2u;
  CSN_DYN_SMEM(float, lds);
  uint2* wl = reinterpret_cast<uint2*>(lds);            // [group][channel group of 4][tile][row in tile]: four bfloat16 each
  const CSN_CONST_AS PwqArgs* a = CSN_KERNARG(PwqArgs, a_byval);
  const int tid = threadIdx.x;
  const int ng = a->ngroups;
  int kgs[3] = {0, 0, 0}, kg_tot = 0;                    // channel groups per slice (slices are padded to four channels)
  for (int s = 0; s < a->nsrc; ++s) { kgs[s] = (a->src[s].C + 3) >> 2; kg_tot += kgs[s]; }
  {   // the bfloat16 image from the fp32 one: W[4 t + i][k] = wimg[g][k][i][t]
    const int per_g = kg_tot * NT * 4;
    for (int idx = tid; idx < ng * per_g; idx += CSN_BLOCK) {
      const int g = idx / per_g, r = idx - g * per_g;
      const int kg = r / (NT * 4), ti = r - kg * (NT * 4);
      const int t = ti >> 2, i = ti & 3;
      int s = 0, lk = kg, krow = 0;
      while (s < a->nsrc - 1 && lk >= kgs[s]) { lk -= kgs[s]; krow += a->src[s].C; ++s; }
      const int C = a->src[s].C;
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * lk + e;
        w[e] = c < C ? a->wimg[(int64_t)g * a->gimg_floats + ((int64_t)(krow + c) * 4 + i) * P + t] : 0.f;
      }
      wl[idx] = make_uint2(csn_pack_bf2(w[0], w[1]), csn_pack_bf2(w[2], w[3]));
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  const int hw = a->HW, nq = hw >> 2;
  const unsigned cs = (unsigned)hw * E;
  const int tiles_img = (nq + 63) >> 6;
  const int nitems = tiles_img * a->B * ng;
  const int nslot = (int)(gridDim.x >> 3) * 4;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;
  const int xcd = blockIdx.x & 7;
  const int iend = min((xcd + 1) * chunk, nitems);
#ifdef CSN_EMU_SEQ
  const uint2* wl_lane = wl;
#else
  const uint2* wl_lane = wl + (lane & 3);
#endif
  auto load4 = [&](csn_buf rb, unsigned off, int c0, int C, uint2 (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = csn_ld_u64(rb, off, (unsigned)min(c0 + j, C - 1) * cs);   // pad channels: zero weights
  };
  auto contract = [&](const uint2 (&v)[4], const uint2* wk, csn_f4 (&acc)[4][NT]) {
    // element s of channels c0 .. c0 + 3: the low / high halves of .x (s = 0, 1) and of .y (s = 2, 3)
    uint2 x[4];
    x[0] = make_uint2(pw16_perm(v[1].x, v[0].x, 0x05040100u), pw16_perm(v[3].x, v[2].x, 0x05040100u));
    x[1] = make_uint2(pw16_perm(v[1].x, v[0].x, 0x07060302u), pw16_perm(v[3].x, v[2].x, 0x07060302u));
    x[2] = make_uint2(pw16_perm(v[1].y, v[0].y, 0x05040100u), pw16_perm(v[3].y, v[2].y, 0x05040100u));
    x[3] = make_uint2(pw16_perm(v[1].y, v[0].y, 0x07060302u), pw16_perm(v[3].y, v[2].y, 0x07060302u));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      Pw16A wa;
#ifdef CSN_EMU_SEQ
      for (int i = 0; i < 4; ++i) wa.r[i] = wk[t * 4 + i];
#else
      wa.r[0] = wk[t * 4];
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) pw16_mfma(wa, x[s], acc[s][t]);
    }
  };
  for (int item = xcd * chunk + (int)(blockIdx.x >> 3) * 4 + wave; item < iend; item += nslot) {
    const int tile = item / ng, g = item - tile * ng;
    const int b = tile / tiles_img, t = tile - b * tiles_img;
    const int q0 = (t << 6) + lane;
    const bool valid = q0 < nq;
    const unsigned off = (unsigned)min(q0, nq - 1) * 4u * E;
    const uint2* wg = wl_lane + (int64_t)g * kg_tot * NT * 4;
    csn_f4 acc[4][NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][tt][i] = 0.f;
    int kg0 = 0;
    for (int s = 0; s < a->nsrc; ++s) {
      const int C = a->src[s].C;
      const csn_buf rb = csn_make_buf_n(reinterpret_cast<const char*>(a->src[s].ptr) + (int64_t)b * a->src[s].Ctot * (int64_t)cs,
                                        (unsigned)C * cs);
      // channel groups kg .. kg + GB - 1 are contracted while the next GB groups are in flight
      uint2 vC[GB][4], vN[GB][4];
#pragma unroll
      for (int u = 0; u < GB; ++u) load4(rb, off, 4 * u, C, vC[u]);
      const int n = kgs[s];
      for (int kg = 0; kg < n; kg += GB) {
#pragma unroll
        for (int u = 0; u < GB; ++u) load4(rb, off, 4 * (kg + GB + u), C, vN[u]);   // (past the last group: the last channel again, never used)
        PW4_FENCE();
#pragma unroll
        for (int u = 0; u < GB; ++u)
          if (kg + u < n) contract(vC[u], wg + (int64_t)(kg0 + kg + u) * NT * 4, acc);
#pragma unroll
        for (int u = 0; u < GB; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) vC[u][j] = vN[u][j];
      }
      kg0 += n;
    }
    const int r0 = a->grp_r0[g], nt = a->grp_nt[g];
    const csn_buf ob = csn_make_buf_n(reinterpret_cast<char*>(a->out) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->nrows * cs);
    const unsigned sv = valid ? off : 0x80000000u;
    const bool route = a->route_x != nullptr;   // (uniform)
    PwqRouteGeo rg{0u, 0u, false};
    csn_buf xb = ob, tb = ob;
    if (route) {
      rg = pwq_route_geo<csn_bf16>(min(q0, nq - 1) * 4, a->route_W);
      xb = csn_make_buf_n(reinterpret_cast<const char*>(a->route_x) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->nrows * cs);
      tb = csn_make_buf_n(reinterpret_cast<const char*>(a->route_t) + (int64_t)b * a->out_ctot * (int64_t)(cs >> 2), (unsigned)a->nrows * (cs >> 2));
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      if (tt < nt) {
        const csn_f4 at[4] = {acc[0][tt], acc[1][tt], acc[2][tt], acc[3][tt]};
        pwq_store_tile<csn_bf16>(ob, xb, tb, sv, off, rg, route, r0 + 4 * tt, cs, at);
      }
    }
  }
}

#define PWQ_INST_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6)
typedef void (*PwqFn)(PwqArgs);
struct PwqEntry { int nt; PwqFn fn[3]; };
#define PWQ_ENTRY(N) {N, {pwq_kernel<N, float>, pwq_kernel<N, csn_bf16>, pwq16_kernel<N>}},
static const PwqEntry g_pwq_table[] = {PWQ_INST_LIST(PWQ_ENTRY)};

int csn_pwq_max_tiles(void) { return 6; }

int csn_launch_pwq(const PwqArgs& a, void* stream) {
  const PwqEntry* e = nullptr;
  for (size_t i = 0; i < sizeof(g_pwq_table) / sizeof(g_pwq_table[0]); ++i)
    if (g_pwq_table[i].nt == a.nt) e = &g_pwq_table[i];
  if (!e) return 1;
  const int nq = a.HW >> 2;
  const int nitems = ((nq + 63) >> 6) * a.B * a.ngroups;
  int nblk = (nitems + 3) / 4;
  if (nblk > a.max_grid) nblk = a.max_grid;
  const dim3 grid((nblk + 7) & ~7);
  const int fi = a.a16 ? (a.mfma16 ? 2 : 1) : 0;
  size_t lds = (size_t)a.ngroups * a.gimg_floats * sizeof(float);
  if (fi == 2) {   // the bfloat16 image: eight bytes per (channel group of four, tile, row)
    int kg = 0;
    for (int s = 0; s < a.nsrc; ++s) kg += (a.src[s].C + 3) >> 2;
    lds = (size_t)a.ngroups * kg * a.nt * 4 * sizeof(uint2);
  }
#ifndef CSN_CPU_EMU
  if (lds > 64 * 1024) {
    const hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(e->fn[fi]),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (er != hipSuccess) return (int)er;
  }
#endif
  CSN_LAUNCH(e->fn[fi], grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
