// k_c3q.hip -- gOctConv 3x3 passes, lane = 2x2 output quad, v_mfma_f32_4x4x1_16B_f32 straight from the load registers.
//
// Reference semantics (CSNet/model/csnet.py:664-726 with kernel_size 3, padding 1): for output branch j
//     y_j = conv3x3(x_j) [+ conv3x3(max_pool2(x_{j-1}))] [+ bilinear_up2(z)],   z = conv3x3(x_{j+1}) (its own launch)
// followed by BN + PReLU (gOctaveCBR 778-792); stage0.0 and the stride-2 entry units of stages 2-4 (33-40, 703-716).
//
// MI355X mapping (round 3; the eval-mode replacement of goct_c3_kernel, profiles/r3_notes.md).  A 3x3 convolution has
// nine matrix-pipe products per loaded value, so it is bound by the fp32 matrix rate, not by bytes; goct_c3_kernel's LDS
// tile + 16x16x4 MFMA reached ~35 % of that rate (two block barriers per 16-channel chunk, 2 blocks per CU, 16-row / 16-
// channel padding).  Here
//   * a LANE owns the output quad (2y + qy, 2x + qx) and, per input channel, the 4x4 window around it in 16 registers
//     (four 64-bit centre loads + eight edge dwords through a bounded buffer resource whose out-of-range offsets return 0 =
//     the zero padding); a high -> low slice (conv of the 2x2 max-pooled finer branch) reads a max-pooled copy written by
//     maxpool2_kernel in front of the launch (k_misc.hip): pooling inside the gather would hold 64 raw registers per channel;
//   * v_mfma_f32_4x4x1_16B_f32 per (channel, tap, row tile, quad pixel): B = the window register of that pixel and tap,
//     A = W[4t + (lane & 3)][channel, tap] from the LDS image (one ds_read_b128 = four row tiles, shared by the four quad
//     pixels), D = four consecutive output channels of the lane's own pixel.  Rows padded to 4, K not padded, 36 NT
//     MFMAs per channel against 12 loads: the next channel's window is in flight while this one is contracted, no barrier
//     after the weight image is staged, one or two waves per SIMD keep the matrix pipe busy;
//   * the low -> high term bilinear_up2(z) is added to the accumulators in the epilogue from the 3x3 neighbourhood of the
//     quad's parent pixel (the same lane-local interpolation as k_pw4.hip), then folded BN + PReLU, 64-bit stores.
// Output channels are cut into M groups of NT row tiles (16 NT accumulator registers); the groups of a tile run on
// neighbouring waves.  Needs even H and W (every shipped geometry; goct_c3_kernel stays as the fallback and serves
// the train-mode / backward launches).
#include "pw4_common.h"

#ifndef C3Q_OCC
#define C3Q_OCC 2
#endif

namespace {

struct C3qGeo {        // per item, per lane
  unsigned row[4];     // byte offset of (window row r, column 2x) inside a channel plane; 0x80000000 when the row is outside
  unsigned dl, dr;     // add to a row offset for the left (column 2x - 1) / right (column 2x + 2) edge; out of range when outside
};

// 4x4 window of one channel at the pass resolution, as loaded (raw registers: the conversion of a bfloat16 window happens where the
// window is contracted, behind the scheduling fence -- csn_device.h)
template <typename AT> struct C3qRaw {
  typename csn_bufacc<AT>::r2 c[4];
  typename csn_bufacc<AT>::r1 l[4], r[4];
};
template <typename AT>
__device__ __forceinline__ void c3q_load_own(csn_buf rb, const C3qGeo& g, unsigned so, C3qRaw<AT>& w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w.c[r] = csn_bufacc<AT>::ldr2(rb, g.row[r], so);
    w.l[r] = csn_bufacc<AT>::ldr1(rb, g.row[r] + g.dl, so);
    w.r[r] = csn_bufacc<AT>::ldr1(rb, g.row[r] + g.dr, so);
  }
}
template <typename AT>
__device__ __forceinline__ void c3q_finish_own(const C3qRaw<AT>& w, float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float2 c = csn_bufacc<AT>::cv2(w.c[r]);
    v[4 * r] = csn_bufacc<AT>::cv1(w.l[r]); v[4 * r + 1] = c.x; v[4 * r + 2] = c.y; v[4 * r + 3] = csn_bufacc<AT>::cv1(w.r[r]);
  }
}

// HL, halo lanes (round 5): a knock-out build without the eight edge loads of a window ran 11 % faster (the load side is bound by the
// number of vector-memory instructions, profiles/r5_notes.md).  A flat tile is then 62 consecutive quads in lanes 1 .. 62; lane 0 /
// lane 63 hold the quad in front of / behind them and only LOAD (nothing of theirs is stored).  Every working lane's left / right
// edge column is its neighbour lane's centre pair (one DPP move on the loaded register, k_misc.hip csn_from_lane_below / _above); no
// lane needs a fallback load.  Quads in the plane's first / last column take zeros (the padding) instead of the neighbour's value.
// The four centre pairs are what stays in flight (8 registers per channel instead of 16); the window is completed when it is used.
struct C3qWin {
  float2 c[4];
#ifdef CSN_EMU_SEQ
  float l[4], r[4];   // (the emulator's lanes run one after the other: it loads the edges)
#endif
};
__device__ __forceinline__ void c3q_issue_hl(csn_buf rb, const C3qGeo& g, unsigned so, C3qWin& w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#ifdef C3Q_KO_NOLOAD   // knock-out build (tools/README.md): no window loads (wrong results, valid timing)
    w.c[r] = make_float2(csn_bits_f(g.row[r]), csn_bits_f(so));
    continue;
#endif
    w.c[r] = csn_ld2(rb, g.row[r], so);
#ifdef CSN_EMU_SEQ
    w.l[r] = csn_ld1(rb, g.row[r] + g.dl, so);
    w.r[r] = csn_ld1(rb, g.row[r] + g.dr, so);
#endif
  }
}
__device__ __forceinline__ void c3q_finish_hl(const C3qWin& w, bool has_l, bool has_r, float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[4 * r + 1] = w.c[r].x; v[4 * r + 2] = w.c[r].y;
#ifdef CSN_EMU_SEQ
    v[4 * r] = w.l[r]; v[4 * r + 3] = w.r[r];
#else
#ifdef C3Q_KO_NODPP    // knock-out build: the window's edge columns without the lane exchange and the masks
    v[4 * r] = w.c[r].y; v[4 * r + 3] = w.c[r].x;
#else
    const float l = csn_bits_f(csn_from_lane_below(csn_f_bits(w.c[r].y)));
    const float rr = csn_bits_f(csn_from_lane_above(csn_f_bits(w.c[r].x)));
    v[4 * r] = has_l ? l : 0.f; v[4 * r + 3] = has_r ? rr : 0.f;
#endif
#endif
  }
}

// nine taps of one channel: wk = image rows of (channel, tap 0) (each [4][P] floats)
template <int NT, int P>
__device__ __forceinline__ void c3q_channel(const float (&v)[16], const float* wk, csn_f4 (&acc)[4][NT]) {
  constexpr int NT4 = (NT + 3) & ~3;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int ty = t9 / 3, tx = t9 - 3 * ty;
    Pw4A<NT4> a;
    pw4_load_a<NT4, P>(wk + t9 * 4 * P, a);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) pw4_mfma<NT4>(a, t, v[4 * ((s >> 1) + ty) + (s & 1) + tx], acc[s][t]);
  }
}

}  // namespace

// The adjoint of the 2x2 max-pool routed in the epilogue of a raw (gradient) launch (C3qArgs::route_x, round 6): the lane's output
// quad o = {(2y, 2x), (2y, 2x + 1), (2y + 1, 2x), (2y + 1, 2x + 1)} is exactly the window of low pixel (y, x); the gradient goes to
// the window's FIRST maximum in that (row-major) order -- `v > best || v != v`, max_pool2d's rule as maxpool2_bwd_add_pair_kernel.
template <typename AT>
__device__ __forceinline__ void c3q_route(float (&o)[4], csn_buf xb, csn_buf tb, unsigned o0, unsigned o1, unsigned ot, unsigned sx, unsigned st) {
  const float2 r0 = csn_bufacc<AT>::ld2(xb, o0, sx), r1 = csn_bufacc<AT>::ld2(xb, o1, sx);
  const float t = csn_bufacc<AT>::ld1(tb, ot, st);
  float best = r0.x; int bi = 0;
  if (r0.y > best || r0.y != r0.y) { best = r0.y; bi = 1; }
  if (r1.x > best || r1.x != r1.x) { best = r1.x; bi = 2; }
  if (r1.y > best || r1.y != r1.y) { best = r1.y; bi = 3; }
  if (bi == 0) o[0] += t;
  if (bi == 1) o[1] += t;
  if (bi == 2) o[2] += t;
  if (bi == 3) o[3] += t;
}

// AT: element type of the activation tensors (float; csn_bf16 = the bf16 train mode's storage, RAW launches only)
template <int NT, bool RAW, typename AT = float, bool HL = false>
__global__ __launch_bounds__(CSN_BLOCK, C3Q_OCC) void c3q_kernel(C3qArgs a_byval) {
  constexpr int NT4 = (NT + 3) & ~3, P = PW4_PITCH(NT4);
  constexpr unsigned E = (unsigned)sizeof(AT);
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS C3qArgs* a = CSN_KERNARG(C3qArgs, a_byval);
  const int tid = threadIdx.x;
  csn_fill_lds16(lds, a->wimg, (a->ngroups * a->gimg_floats) >> 2, tid);
  __syncthreads();
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  const int H = a->H, W = a->W, Hq = H >> 1, Wq = W >> 1;
  const unsigned cs = (unsigned)(H * W) * E;
  const int twl = a->twl;
  const int lx = lane & ((1 << twl) - 1), ly = lane >> twl;
  const int ng = a->ngroups;
  const int tiles_xy = a->tiles_x * a->tiles_y;
  const int nitems = tiles_xy * a->B * ng;
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
    const int b = tile / tiles_xy, txy = tile - b * tiles_xy;
    int yq, xq;
    bool work = true;   // HL: lanes 0 and 63 only load
    if (HL) {             // flat tiles of 62 quads in lanes 1 .. 62 (the launcher passes flat tiles only)
      const int p = max(txy * 62 + lane - 1, 0);
      int q = (int)((float)p * (1.0f / (float)Wq));
      q -= (q * Wq > p) ? 1 : 0;
      q += ((q + 1) * Wq <= p) ? 1 : 0;
      yq = q; xq = p - q * Wq;
      work = lane >= 1 && lane <= 62;
    } else if (twl >= PW4_FLAT_TWL) {   // flat tiles: 64 consecutive quads of the plane (see k_pw4.hip)
      const int p = txy * 64 + lane;
      int q = (int)((float)p * (1.0f / (float)Wq));
      q -= (q * Wq > p) ? 1 : 0;
      q += ((q + 1) * Wq <= p) ? 1 : 0;
      yq = q; xq = p - q * Wq;
    } else {
      const int ty = txy / a->tiles_x, tx = txy - ty * a->tiles_x;
      yq = (ty << (6 - twl)) + ly; xq = (tx << twl) + lx;
    }
    const bool valid = work && yq < Hq && xq < Wq;
    const int y = min(yq, Hq - 1), x = min(xq, Wq - 1);
    const float* wg = wl_lane + g * a->gimg_floats;

    csn_f4 acc[4][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][t][i] = 0.f;

    const bool has_l = x > 0, has_r = 2 * x + 2 < W;
    int krow = 0;   // image row (= gathered entry) of the slice's first channel, tap 0
    for (int s = 0; s < a->nsrc; ++s) {
      const int C = a->src[s].C;
      C3qGeo geo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = 2 * y - 1 + r;
        geo.row[r] = (yy >= 0 && yy < H) ? (unsigned)(yy * W + 2 * x) * E : 0x80000000u;
      }
      geo.dl = has_l ? 0u - E : 0x40000000u;
      geo.dr = has_r ? 2u * E : 0x40000000u;
      const csn_buf rb = csn_make_buf_n(reinterpret_cast<const char*>(a->src[s].ptr) + (int64_t)b * a->src[s].Ctot * (int64_t)cs,
                                        (unsigned)a->src[s].Ctot * cs);
      if (HL) {   // the centre pairs of channel c + 1 in flight while channel c is completed (lane exchange) and contracted
        C3qWin wA, wB;
        float v[16];
        c3q_issue_hl(rb, geo, 0u, wA);
        PW4_FENCE();
        int c = 0;
        for (; c + 1 < C; c += 2) {
          c3q_issue_hl(rb, geo, (unsigned)(c + 1) * cs, wB);
          PW4_FENCE();
          c3q_finish_hl(wA, has_l, has_r, v);
          c3q_channel<NT, P>(v, wg + (krow + 9 * c) * 4 * P, acc);
          c3q_issue_hl(rb, geo, (unsigned)min(c + 2, C - 1) * cs, wA);
          PW4_FENCE();
          c3q_finish_hl(wB, has_l, has_r, v);
          c3q_channel<NT, P>(v, wg + (krow + 9 * (c + 1)) * 4 * P, acc);
        }
        if (c < C) {
          c3q_finish_hl(wA, has_l, has_r, v);
          c3q_channel<NT, P>(v, wg + (krow + 9 * c) * 4 * P, acc);
        }
        krow += 9 * C;
        continue;
      }
      // channel c is contracted while channel c + 1 is in flight: two register sets, channels walked in pairs
      C3qRaw<AT> wA, wB;
      float v[16];
      c3q_load_own<AT>(rb, geo, 0u, wA);
      PW4_FENCE();
      const int nf = C - 1;
      int c = 0;
      for (int p = 0; p < (nf >> 1); ++p) {
        c3q_load_own<AT>(rb, geo, (unsigned)(c + 1) * cs, wB);
        PW4_FENCE();
        c3q_finish_own<AT>(wA, v);
        c3q_channel<NT, P>(v, wg + (krow + 9 * c) * 4 * P, acc);
        c3q_load_own<AT>(rb, geo, (unsigned)(c + 2) * cs, wA);
        PW4_FENCE();
        c3q_finish_own<AT>(wB, v);
        c3q_channel<NT, P>(v, wg + (krow + 9 * (c + 1)) * 4 * P, acc);
        c += 2;
      }
      if (nf & 1) {
        c3q_load_own<AT>(rb, geo, (unsigned)(c + 1) * cs, wB);
        PW4_FENCE();
        c3q_finish_own<AT>(wA, v);
        c3q_channel<NT, P>(v, wg + (krow + 9 * c) * 4 * P, acc);
        ++c;
        wA = wB;
      }
      c3q_finish_own<AT>(wA, v);
      c3q_channel<NT, P>(v, wg + (krow + 9 * c) * 4 * P, acc);
      krow += 9 * C;
    }

    // ---- epilogue: + bilinear_up2(z), folded BN + PReLU, 64-bit stores of the quad rows ----
    const int r0 = a->grp_r0[g], nt = a->grp_nt[g];
    const unsigned o0 = (unsigned)((2 * y) * W + 2 * x) * E, o1 = o0 + (unsigned)W * E;
    const unsigned sv0 = valid ? o0 : 0x80000000u, sv1 = valid ? o1 : 0x80000000u;
    const csn_buf ob = csn_make_buf_n(reinterpret_cast<char*>(a->out) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->out_ctot * cs);
    const bool route = a->route_x != nullptr;   // (uniform; raw launches only)
    const unsigned ort = (unsigned)(y * Wq + x) * E;
    const csn_buf rxb = route ? csn_make_buf_n(reinterpret_cast<const char*>(a->route_x) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->out_ctot * cs) : ob;
    const csn_buf rtb = route ? csn_make_buf_n(reinterpret_cast<const char*>(a->route_t) + (int64_t)b * a->out_ctot * (int64_t)(cs >> 2), (unsigned)a->out_ctot * (cs >> 2)) : ob;
    csn_cfp ep = csn_const(a->ep) + 4 * r0;
    unsigned oz[9];
    csn_buf zb = ob;
    const unsigned csz = cs >> 2;
    if (a->z) {
      const int yy[3] = {max(y - 1, 0), y, min(y + 1, Hq - 1)};
      const int xx[3] = {max(x - 1, 0), x, min(x + 1, Wq - 1)};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) oz[3 * r + c] = (unsigned)(yy[r] * Wq + xx[c]) * E;
      zb = csn_make_buf_n(reinterpret_cast<const char*>(a->z) + (int64_t)b * a->z_ctot * (int64_t)csz, (unsigned)a->z_ctot * csz);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * t + i;
          float zq[4] = {0.f, 0.f, 0.f, 0.f};
          if (a->z) {
            float zv[9];
            const unsigned zo = (unsigned)min(a->z_c0 + r0 + r, a->z_ctot - 1) * csz;
#pragma unroll
            for (int k = 0; k < 9; ++k) zv[k] = csn_bufacc<AT>::ld1(zb, oz[k], zo);
            pw4_up2_quad(zv, zq);
          }
          float o[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float zsum = acc[s][t][i] + zq[s];
            o[s] = RAW ? zsum : pw4_epi(zsum, ep[4 * r], ep[4 * r + 1], ep[4 * r + 2]);
          }
          const unsigned so = (unsigned)(a->out_c0 + r0 + r) * cs;
          if (RAW && route) c3q_route<AT>(o, rxb, rtb, o0, o1, ort, so, so >> 2);
          csn_bufacc<AT>::st2(ob, sv0, so, make_float2(o[0], o[1]));
          csn_bufacc<AT>::st2(ob, sv1, so, make_float2(o[2], o[3]));
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// c3q16_kernel (round 4): raw launches with bfloat16 tensors on v_mfma_f32_4x4x4_16B_bf16.  The SQ counters of a bf16 train step put
// c3q_kernel<bf16> at 64 % of the fp32 matrix pipe (profiles/r4_sq_train_summary.txt): 36 NT v_mfma_f32_4x4x1 per channel.  Here one
// issue contracts FOUR channels of a window position:
//   per channel the 4x4 window is 12 dwords AS LOADED (rows x {pixels 2x - 2 | 2x - 1, 2x | 2x + 1, 2x + 2 | 2x + 3}: aligned dwords
//   instead of a dword + two shorts -- the outer halves are not used); per group of four channels and window position (row, column)
//   two v_perm_b32 pick that position's half of the four channels' dwords -> the B operand (no conversion to float);
//   A = {W[4 t + (lane & 3)][channels c0 .. c0 + 3, tap]} from a bfloat16 image the block builds in LDS from the fp32 one.
// 36 NT matrix instructions and 32 v_perm per FOUR channels (before: 144 NT and 64 conversions).  Weights rounded to bfloat16 for
// this pass (fp32 masters; see pwq16_kernel).  Epilogue as c3q_kernel<RAW>.
struct C3q16Raw { unsigned c[4][4], l[4][4], r[4][4]; };   // [channel of the group][window row]

__device__ __forceinline__ void c3q16_load(csn_buf rb, const C3qGeo& g, unsigned cs, int c0, int C, C3q16Raw& w) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned so = (unsigned)min(c0 + j, C - 1) * cs;   // pad channels: zero weights
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w.c[j][r] = csn_ld_u32(rb, g.row[r], so);
      w.l[j][r] = csn_ld_u32(rb, g.row[r] + g.dl, so);
      w.r[j][r] = csn_ld_u32(rb, g.row[r] + g.dr, so);
    }
  }
}

// nine taps of four channels: wk = the group's image [tap][tile][row in tile]
// window position (row, column) of the four channels of a group: the B operands
__device__ __forceinline__ void c3q16_pack(const C3q16Raw& w, uint2 (&x)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    x[r][0] = make_uint2(pw16_perm(w.l[1][r], w.l[0][r], 0x07060302u), pw16_perm(w.l[3][r], w.l[2][r], 0x07060302u));
    x[r][1] = make_uint2(pw16_perm(w.c[1][r], w.c[0][r], 0x05040100u), pw16_perm(w.c[3][r], w.c[2][r], 0x05040100u));
    x[r][2] = make_uint2(pw16_perm(w.c[1][r], w.c[0][r], 0x07060302u), pw16_perm(w.c[3][r], w.c[2][r], 0x07060302u));
    x[r][3] = make_uint2(pw16_perm(w.r[1][r], w.r[0][r], 0x05040100u), pw16_perm(w.r[3][r], w.r[2][r], 0x05040100u));
  }
}
template <int NT>
__device__ __forceinline__ void c3q16_group(const uint2 (&x)[4][4], const uint2* wk, csn_f4 (&acc)[4][NT]) {
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int ty = t9 / 3, tx = t9 - 3 * ty;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      Pw16A wa;
#ifdef CSN_EMU_SEQ
      for (int i = 0; i < 4; ++i) wa.r[i] = wk[(t9 * NT + t) * 4 + i];
#else
      wa.r[0] = wk[(t9 * NT + t) * 4];
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) pw16_mfma(wa, x[(s >> 1) + ty][(s & 1) + tx], acc[s][t]);
    }
  }
}

template <int NT>
__global__ __launch_bounds__(CSN_BLOCK, C3Q_OCC) void c3q16_kernel(C3qArgs a_byval) {
  typedef csn_bf16 AT;
  constexpr int NT4 = (NT + 3) & ~3, P = PW4_PITCH(NT4);
  constexpr unsigned E = 2u;
  CSN_DYN_SMEM(float, lds);
  uint2* wl = reinterpret_cast<uint2*>(lds);   // [group][channel group of 4][tap][tile][row in tile]: four bfloat16 each
  const CSN_CONST_AS C3qArgs* a = CSN_KERNARG(C3qArgs, a_byval);
  const int tid = threadIdx.x;
  const int ng = a->ngroups;
  int kgs[3] = {0, 0, 0}, kg_tot = 0;
  for (int s = 0; s < a->nsrc; ++s) { kgs[s] = (a->src[s].C + 3) >> 2; kg_tot += kgs[s]; }
  {   // the bfloat16 image from the fp32 one: W[4 t + i][entry] = wimg[g][entry][i][t], entry = 9 * channel + tap inside the slice
    const int per_g = kg_tot * 9 * NT * 4;
    for (int idx = tid; idx < ng * per_g; idx += CSN_BLOCK) {
      const int g = idx / per_g, r = idx - g * per_g;
      const int kt = r / (NT * 4), ti = r - kt * (NT * 4);
      const int kg = kt / 9, t9 = kt - 9 * kg;
      const int t = ti >> 2, i = ti & 3;
      int s = 0, lk = kg, krow = 0;
      while (s < a->nsrc - 1 && lk >= kgs[s]) { lk -= kgs[s]; krow += 9 * a->src[s].C; ++s; }
      const int C = a->src[s].C;
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * lk + e;
        w[e] = c < C ? a->wimg[(int64_t)g * a->gimg_floats + ((int64_t)(krow + 9 * c + t9) * 4 + i) * P + t] : 0.f;
      }
      wl[idx] = make_uint2(csn_pack_bf2(w[0], w[1]), csn_pack_bf2(w[2], w[3]));
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = pw4_uniform(tid >> 6);
  const int H = a->H, W = a->W, Hq = H >> 1, Wq = W >> 1;
  const unsigned cs = (unsigned)(H * W) * E;
  const int twl = a->twl;
  const int lx = lane & ((1 << twl) - 1), ly = lane >> twl;
  const int tiles_xy = a->tiles_x * a->tiles_y;
  const int nitems = tiles_xy * a->B * ng;
  const int nslot = (int)(gridDim.x >> 3) * 4;
  const int chunk = (((nitems + 7) >> 3) + ng - 1) / ng * ng;
  const int xcd = blockIdx.x & 7;
  const int iend = min((xcd + 1) * chunk, nitems);
#ifdef CSN_EMU_SEQ
  const uint2* wl_lane = wl;
#else
  const uint2* wl_lane = wl + (lane & 3);
#endif
  for (int item = xcd * chunk + (int)(blockIdx.x >> 3) * 4 + wave; item < iend; item += nslot) {
    const int tile = item / ng, g = item - tile * ng;
    const int b = tile / tiles_xy, txy = tile - b * tiles_xy;
    int yq, xq;
    if (twl >= PW4_FLAT_TWL) {
      const int p = txy * 64 + lane;
      int q = (int)((float)p * (1.0f / (float)Wq));
      q -= (q * Wq > p) ? 1 : 0;
      q += ((q + 1) * Wq <= p) ? 1 : 0;
      yq = q; xq = p - q * Wq;
    } else {
      const int ty = txy / a->tiles_x, tx = txy - ty * a->tiles_x;
      yq = (ty << (6 - twl)) + ly; xq = (tx << twl) + lx;
    }
    const bool valid = yq < Hq && xq < Wq;
    const int y = min(yq, Hq - 1), x = min(xq, Wq - 1);
    const uint2* wg = wl_lane + (int64_t)g * kg_tot * 9 * NT * 4;

    csn_f4 acc[4][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][t][i] = 0.f;

    const bool has_l = x > 0, has_r = 2 * x + 2 < W;
    int kg0 = 0;
    for (int s = 0; s < a->nsrc; ++s) {
      const int C = a->src[s].C;
      C3qGeo geo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = 2 * y - 1 + r;
        geo.row[r] = (yy >= 0 && yy < H) ? (unsigned)(yy * W + 2 * x) * E : 0x80000000u;
      }
      geo.dl = has_l ? 0u - 2u * E : 0x40000000u;   // the dword of pixels 2x - 2 | 2x - 1
      geo.dr = has_r ? 2u * E : 0x40000000u;        // ... of pixels 2x + 2 | 2x + 3
      const csn_buf rb = csn_make_buf_n(reinterpret_cast<const char*>(a->src[s].ptr) + (int64_t)b * a->src[s].Ctot * (int64_t)cs,
                                        (unsigned)a->src[s].Ctot * cs);
      // group kg is contracted while the loads of group kg + 1 are in flight: the window as loaded (48 registers) is packed into the
      // operands (32) first, the next group's loads then reuse its registers
      C3q16Raw w;
      c3q16_load(rb, geo, cs, 0, C, w);
      const int n = kgs[s];
      for (int kg = 0; kg < n; ++kg) {
        uint2 x[4][4];
        c3q16_pack(w, x);
        PW4_FENCE();
        c3q16_load(rb, geo, cs, 4 * (kg + 1), C, w);   // (past the last group: the last channel again, never used)
        PW4_FENCE();
        c3q16_group<NT>(x, wg + (int64_t)(kg0 + kg) * 9 * NT * 4, acc);
      }
      kg0 += n;
    }

    // ---- epilogue: + bilinear_up2(z), raw 32-bit stores of the quad rows (as c3q_kernel<RAW>) ----
    const int r0 = a->grp_r0[g], nt = a->grp_nt[g];
    const unsigned o0 = (unsigned)((2 * y) * W + 2 * x) * E, o1 = o0 + (unsigned)W * E;
    const unsigned sv0 = valid ? o0 : 0x80000000u, sv1 = valid ? o1 : 0x80000000u;
    const csn_buf ob = csn_make_buf_n(reinterpret_cast<char*>(a->out) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->out_ctot * cs);
    const bool route = a->route_x != nullptr;   // (uniform; raw launches only)
    const unsigned ort = (unsigned)(y * Wq + x) * E;
    const csn_buf rxb = route ? csn_make_buf_n(reinterpret_cast<const char*>(a->route_x) + (int64_t)b * a->out_ctot * (int64_t)cs, (unsigned)a->out_ctot * cs) : ob;
    const csn_buf rtb = route ? csn_make_buf_n(reinterpret_cast<const char*>(a->route_t) + (int64_t)b * a->out_ctot * (int64_t)(cs >> 2), (unsigned)a->out_ctot * (cs >> 2)) : ob;
    unsigned oz[9];
    csn_buf zb = ob;
    const unsigned csz = cs >> 2;
    if (a->z) {
      const int yy[3] = {max(y - 1, 0), y, min(y + 1, Hq - 1)};
      const int xx[3] = {max(x - 1, 0), x, min(x + 1, Wq - 1)};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) oz[3 * r + c] = (unsigned)(yy[r] * Wq + xx[c]) * E;
      zb = csn_make_buf_n(reinterpret_cast<const char*>(a->z) + (int64_t)b * a->z_ctot * (int64_t)csz, (unsigned)a->z_ctot * csz);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * t + i;
          float zq[4] = {0.f, 0.f, 0.f, 0.f};
          if (a->z) {
            float zv[9];
            const unsigned zo = (unsigned)min(a->z_c0 + r0 + r, a->z_ctot - 1) * csz;
#pragma unroll
            for (int k = 0; k < 9; ++k) zv[k] = csn_bufacc<AT>::ld1(zb, oz[k], zo);
            pw4_up2_quad(zv, zq);
          }
          const unsigned so = (unsigned)(a->out_c0 + r0 + r) * cs;
          float o[4] = {acc[0][t][i] + zq[0], acc[1][t][i] + zq[1], acc[2][t][i] + zq[2], acc[3][t][i] + zq[3]};
          if (route) c3q_route<AT>(o, rxb, rtb, o0, o1, ort, so, so >> 2);
          csn_bufacc<AT>::st2(ob, sv0, so, make_float2(o[0], o[1]));
          csn_bufacc<AT>::st2(ob, sv1, so, make_float2(o[2], o[3]));
        }
      }
    }
  }
}

// ---- instantiation table -----------------------------------------------------------------------------------------
#define C3Q_INST_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

typedef void (*C3qFn)(C3qArgs);
struct C3qEntry { int nt; C3qFn fn[6]; };   // BN + PReLU / raw / raw with bfloat16 tensors / ... on the bf16 matrix instruction /
                                            // the first two on halo-lane tiles (C3qArgs::hl)
#define C3Q_ENTRY(N) {N, {c3q_kernel<N, false>, c3q_kernel<N, true>, c3q_kernel<N, true, csn_bf16>, c3q16_kernel<N>, \
                          c3q_kernel<N, false, float, true>, c3q_kernel<N, true, float, true>}},
static const C3qEntry g_c3q_table[] = {C3Q_INST_LIST(C3Q_ENTRY)};

int csn_c3q_max_tiles(void) { return 7; }

int csn_launch_c3q(const C3qArgs& a, int raw, void* stream) {
  const C3qEntry* e = nullptr;
  for (size_t i = 0; i < sizeof(g_c3q_table) / sizeof(g_c3q_table[0]); ++i)
    if (g_c3q_table[i].nt == a.nt) e = &g_c3q_table[i];
  if (!e) return 1;
  const int nitems = a.tiles_x * a.tiles_y * a.B * a.ngroups;
  int nblk = (nitems + 3) / 4;
  if (nblk > a.max_grid) nblk = a.max_grid;
  const dim3 grid((nblk + 7) & ~7);
  const size_t lds = (size_t)a.ngroups * a.gimg_floats * sizeof(float);
#ifndef CSN_CPU_EMU
  static CsnPerDeviceOnce attr_once;
  const int ast = attr_once.run([&]() {
    for (size_t i = 0; i < sizeof(g_c3q_table) / sizeof(g_c3q_table[0]); ++i)
      for (int r = 0; r < 6; ++r) {
        const hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(g_c3q_table[i].fn[r]),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (er != hipSuccess) return (int)er;
      }
    return 0;
  });
  if (ast != 0) return ast;
#endif
  if (a.a16 && !raw) return 1;   // bfloat16 tensors: train-mode (raw) launches only
  if (a.route_x && (!raw || !a.route_t || a.z)) return 1;   // routing: gradient launches only
  if (a.a16 && a.mfma16) {
    int kg = 0;
    for (int s = 0; s < a.nsrc; ++s) kg += (a.src[s].C + 3) >> 2;
    const size_t lds16 = (size_t)a.ngroups * kg * 9 * a.nt * 4 * sizeof(uint2);
    CSN_LAUNCH(e->fn[3], grid, dim3(CSN_BLOCK), lds16, stream, a);
    return (int)hipGetLastError();
  }
  if (a.hl) {
    if (a.a16 || a.twl < PW4_FLAT_TWL) return 1;   // halo-lane tiles: float tensors, flat tiles of 62 quads (the planner's choice)
    CSN_LAUNCH(e->fn[raw ? 5 : 4], grid, dim3(CSN_BLOCK), lds, stream, a);
    return (int)hipGetLastError();
  }
  CSN_LAUNCH(e->fn[raw ? (a.a16 ? 2 : 1) : 0], grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
