// k_goct_pw.hip -- gOctConv with 1x1 kernels + BN + PReLU, ALL output branches of a unit per block.
//
// Reference semantics (CSNet/model/csnet.py):
//   gOctaveConv.forward 664-726:  y_j = sum_i T_ij(x_i), weight block W[co_j, ci_i];
//       i == j : conv(x_i)                                     (716-717)
//       i >  j : bilinear_up_{2^(i-j)}( conv(x_i) )            (702-707)  "low -> high"
//       i <  j : conv( max_pool_{2^(j-i)}(x_i) )               (708-714)  "high -> low"
//   gOctaveCBR.forward 778-792:   PReLU_j(BN_j(y_j)) per output branch (eval BN folded to scale/shift)
//   also used for cls_layer (1x1 + bias, csnet.py:306-308,381) with scale=1, shift=bias, alpha=1.
//
// MI355X mapping.  A block owns a 16x32 tile of branch 0 and the matching 8x16 / 4x8 tiles of
// branches 1 / 2.  Work is a short list of *passes*; a pass walks the pixels of one branch, gathers
// the input channels of that pixel into VGPRs exactly once (own-resolution channels plus channels
// max-pooled on the fly from the higher-resolution inputs), and then produces output rows one pair at
// a time: every row is a dot product of the register-resident inputs with a weight row that the whole
// wave shares, so weights stream through the scalar cache (s_load) and feed v_fmac as SGPR operands.
//   z pass   (branch i > 0): rows of the low->high blocks W[co_j, ci_i] evaluated at low resolution
//            on the tile plus a 1-pixel ring, written to LDS (never to HBM);
//   main pass (branch j)   : rows of [W_jj | W_ij (i<j, pooled)] + bilinear taps of the LDS z regions
//            (i>j) -> folded BN -> PReLU -> one coalesced store per output channel.
// Every input element is fetched from HBM once per unit (the high-res input is read by the branch-0
// pass and again, max-pooled, by the lower pass of the same block -> L2 hit), every output element is
// written once: algorithmic bytes == HBM bytes.
#include "csn_kernels.h"

template <int R>
struct PwTile {
  static constexpr int TY = PW_TY0 >> R;
  static constexpr int TX = PW_TX0 >> R;
  static constexpr int RY = TY + 2;
  static constexpr int RX = TX + 2;
  static constexpr int RING = RY * RX;
};

__device__ __forceinline__ int pw_ring_px(int r) {
  return ((PW_TY0 >> r) + 2) * ((PW_TX0 >> r) + 2);
}
__device__ __forceinline__ int pw_ring_rx(int r) { return (PW_TX0 >> r) + 2; }

typedef const CSN_CONST_AS PwPass* PwPassP;

// Gather the (<= MAXC) channel vector of pixel (y,x) (branch resolution Hr x Wr) of image b from up to three
// channel slices; a slice with shift s lives at 2^s times the resolution and is max-pooled on the fly.
template <int MAXC>
__device__ __forceinline__ void pw_gather1(PwPassP ps, int b, int y, int x, int Hr, int Wr, float (&v)[1][MAXC]) {
  const int c1 = ps->src[0].C;
  const int c2 = c1 + ps->src[1].C;
  const int c3 = c2 + ps->src[2].C;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    float val = 0.f;
    if (k < c3) {
      const int s = (k < c1) ? 0 : (k < c2) ? 1 : 2;
      const int ch = k - (s == 0 ? 0 : s == 1 ? c1 : c2);
      const int sh = ps->src[s].shift;
      const int Ws = Wr << sh;
      const int64_t hw = (int64_t)(Hr << sh) * Ws;
      const float* __restrict__ p =
          ps->src[s].ptr + ((int64_t)b * ps->src[s].Ctot + ch) * hw + ((int64_t)y << sh) * Ws + (x << sh);
      if (sh == 0) {
        val = p[0];
      } else if (sh == 1) {
        const float2 a0 = *reinterpret_cast<const float2*>(p);
        const float2 a1 = *reinterpret_cast<const float2*>(p + Ws);
        val = fmaxf(fmaxf(a0.x, a0.y), fmaxf(a1.x, a1.y));
      } else {
        float m = -3.402823466e+38f;
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {
          const float4 q = *reinterpret_cast<const float4*>(p + (int64_t)yy * Ws);
          m = fmaxf(m, fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
        }
        val = m;
      }
    }
    v[0][k] = val;
  }
}

// Two horizontally adjacent pixels (x even) of the own-resolution source only (branch-0 pass).
template <int MAXC>
__device__ __forceinline__ void pw_gather2(PwPassP ps, int b, int y, int x, int Hr, int Wr, float (&v)[2][MAXC]) {
  const int c1 = ps->src[0].C;
  const int64_t cs = (int64_t)Hr * Wr;
  const float* __restrict__ base = ps->src[0].ptr + (int64_t)b * ps->src[0].Ctot * cs + (int64_t)y * Wr + x;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    float2 q = make_float2(0.f, 0.f);
    if (k < c1) q = *reinterpret_cast<const float2*>(base + k * cs);
    v[0][k] = q.x;
    v[1][k] = q.y;
  }
}

// Row engine.  Output rows are produced two at a time from the register-resident channel vectors of
// NPX pixels (2*NPX independent accumulation chains); the weight rows are wave-uniform and stream
// through the scalar cache.  Channel groups of 4 beyond cin4 are skipped by a uniform branch.
template <int MAXC, int NPX, class Sink>
__device__ __forceinline__ void pw_rows(csn_cfp w, int cin4, int nrows, const float (&v)[NPX][MAXC], Sink& sink) {
  int row = 0;
  for (; row + 2 <= nrows; row += 2) {
    float a0[NPX], a1[NPX];
#pragma unroll
    for (int p = 0; p < NPX; ++p) a0[p] = a1[p] = 0.f;
    csn_cfp w0 = w + row * cin4;
    csn_cfp w1 = w0 + cin4;
#pragma unroll
    for (int k0 = 0; k0 < MAXC; k0 += 4) {
      if (k0 < cin4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float wa = w0[k0 + u], wb = w1[k0 + u];
#pragma unroll
          for (int p = 0; p < NPX; ++p) {
            a0[p] = fmaf(wa, v[p][k0 + u], a0[p]);
            a1[p] = fmaf(wb, v[p][k0 + u], a1[p]);
          }
        }
      }
    }
    sink(row, a0);
    sink(row + 1, a1);
  }
  if (row < nrows) {
    float a0[NPX];
#pragma unroll
    for (int p = 0; p < NPX; ++p) a0[p] = 0.f;
    csn_cfp w0 = w + row * cin4;
#pragma unroll
    for (int k0 = 0; k0 < MAXC; k0 += 4) {
      if (k0 < cin4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float wa = w0[k0 + u];
#pragma unroll
          for (int p = 0; p < NPX; ++p) a0[p] = fmaf(wa, v[p][k0 + u], a0[p]);
        }
      }
    }
    sink(row, a0);
  }
}

// Bilinear tap set of one destination pixel into an LDS z region of source branch rs.
struct PwTap {
  int i00, i01, i10, i11;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ PwTap pw_tap(int y, int x, int r, int rs, int H0, int W0, int ty0, int tx0) {
  const int d = rs - r;
  const float inv_f = d == 1 ? 0.5f : 0.25f;
  int y0, y1, x0, x1;
  float ly, lx;
  csn_bilin(y, inv_f, H0 >> rs, y0, y1, ly);
  csn_bilin(x, inv_f, W0 >> rs, x0, x1, lx);
  const int oy = (ty0 >> rs) - 1, ox = (tx0 >> rs) - 1;  // image coords of ring cell (0,0)
  const int rx = pw_ring_rx(rs);
  PwTap t;
  t.i00 = (y0 - oy) * rx + (x0 - ox);
  t.i01 = (y0 - oy) * rx + (x1 - ox);
  t.i10 = (y1 - oy) * rx + (x0 - ox);
  t.i11 = (y1 - oy) * rx + (x1 - ox);
  t.w00 = (1.f - ly) * (1.f - lx);
  t.w01 = (1.f - ly) * lx;
  t.w10 = ly * (1.f - lx);
  t.w11 = ly * lx;
  return t;
}

__device__ __forceinline__ float pw_tap_eval(const float* __restrict__ z, const PwTap& t) {
  return t.w00 * z[t.i00] + t.w01 * z[t.i01] + t.w10 * z[t.i10] + t.w11 * z[t.i11];
}

template <int MAXC_TOP, int MAXC_LOW>
__global__ __launch_bounds__(CSN_BLOCK) void goct_pw_kernel(PwArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS PwArgs* a = CSN_KERNARG(PwArgs, a_byval);
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * PW_TY0, tx0 = blockIdx.x * PW_TX0;  // tile origin at branch 0
  const int H0 = a->H0, W0 = a->W0;
  const int nz_pass = a->nz_pass, npass = a->npass, n_top = a->n_top;

  // ---- z passes: low->high partial sums at low resolution, tile + ring, into LDS ----
  for (int pi = 0; pi < nz_pass; ++pi) {
    PwPassP ps = &a->pass[pi];
    const int r = ps->r;
    const int Hr = H0 >> r, Wr = W0 >> r;
    const int rx = pw_ring_rx(r), npx = pw_ring_px(r);
    const int oy = (ty0 >> r) - 1, ox = (tx0 >> r) - 1;
    const int cin4 = ps->cin4, nrows = ps->nrows, acc_in = ps->acc_in;
    csn_cfp w = csn_const(ps->w);
    for (int p = tid; p < npx; p += CSN_BLOCK) {
      const int py = p / rx, px = p - py * rx;
      const int y = oy + py, x = ox + px;
      if (y < 0 || y >= Hr || x < 0 || x >= Wr) continue;  // never sampled (indices are clamped)
      float v[1][MAXC_LOW];
      pw_gather1<MAXC_LOW>(ps, b, y, x, Hr, Wr, v);
      float* __restrict__ zp = lds + ps->z_off + p;
      auto sink = [&](int row, const float (&acc)[1]) {
        zp[row * npx] = acc_in ? zp[row * npx] + acc[0] : acc[0];
      };
      pw_rows<MAXC_LOW, 1>(w, cin4, nrows, v, sink);
    }
  }
  if (nz_pass > 0) __syncthreads();

  // ---- main passes of the branches below 0: one pixel per lane ----
  for (int pi = nz_pass; pi < npass - n_top; ++pi) {
    PwPassP ps = &a->pass[pi];
    const int r = ps->r;
    const int Hr = H0 >> r, Wr = W0 >> r;
    const int tx = PW_TX0 >> r, npx = (PW_TY0 >> r) * tx;
    const int cin4 = ps->cin4, nrows = ps->nrows;
    const int acc_in = ps->acc_in, fin = ps->final_seg;
    const int nz = fin ? ps->nz : 0;
    csn_cfp w = csn_const(ps->w);
    csn_cfp scale = csn_const(ps->scale), shift = csn_const(ps->shift), alpha = csn_const(ps->alpha);
    const int zrs0 = ps->zadd[0].rs, zrs1 = ps->zadd[1].rs;
    const float* __restrict__ zb0 = lds + ps->zadd[0].z_off;
    const float* __restrict__ zb1 = lds + ps->zadd[1].z_off;
    const int zs0 = pw_ring_px(zrs0), zs1 = pw_ring_px(zrs1);
    for (int p = tid; p < npx; p += CSN_BLOCK) {
      const int py = p / tx, px = p - py * tx;
      const int y = (ty0 >> r) + py, x = (tx0 >> r) + px;
      if (y >= Hr || x >= Wr) continue;
      float v[1][MAXC_LOW];
      pw_gather1<MAXC_LOW>(ps, b, y, x, Hr, Wr, v);
      PwTap tap0, tap1;
      if (nz > 0) tap0 = pw_tap(y, x, r, zrs0, H0, W0, ty0, tx0);
      if (nz > 1) tap1 = pw_tap(y, x, r, zrs1, H0, W0, ty0, tx0);
      float* __restrict__ op = ps->out + ((int64_t)b * nrows * Hr + y) * Wr + x;
      const int64_t cs = (int64_t)Hr * Wr;
      auto sink = [&](int row, const float (&acc_)[1]) {
        float acc = acc_[0];
        if (acc_in) acc += op[row * cs];
        if (nz > 0) acc += pw_tap_eval(zb0 + row * zs0, tap0);
        if (nz > 1) acc += pw_tap_eval(zb1 + row * zs1, tap1);
        op[row * cs] = fin ? csn_epi(acc, scale[row], shift[row], alpha[row]) : acc;
      };
      pw_rows<MAXC_LOW, 1>(w, cin4, nrows, v, sink);
    }
  }

  // ---- passes of branch 0: two pixels per lane (float2 loads / stores) ----
  for (int pi = npass - n_top; pi < npass; ++pi) {
    PwPassP ps = &a->pass[pi];
    constexpr int LXN = PW_TX0 / 2;
    const int py = tid / LXN, px = (tid - py * LXN) * 2;
    const int y = ty0 + py, x = tx0 + px;
    const int cin4 = ps->cin4, nrows = ps->nrows;
    const int acc_in = ps->acc_in, fin = ps->final_seg;
    const int nz = fin ? ps->nz : 0;
    csn_cfp w = csn_const(ps->w);
    csn_cfp scale = csn_const(ps->scale), shift = csn_const(ps->shift), alpha = csn_const(ps->alpha);
    const int zrs0 = ps->zadd[0].rs, zrs1 = ps->zadd[1].rs;
    const float* __restrict__ zb0 = lds + ps->zadd[0].z_off;
    const float* __restrict__ zb1 = lds + ps->zadd[1].z_off;
    const int zs0 = pw_ring_px(zrs0), zs1 = pw_ring_px(zrs1);
    if (y < H0 && x < W0) {
      float v[2][MAXC_TOP];
      pw_gather2<MAXC_TOP>(ps, b, y, x, H0, W0, v);
      PwTap ta0, tb0, ta1, tb1;
      if (nz > 0) {
        ta0 = pw_tap(y, x, 0, zrs0, H0, W0, ty0, tx0);
        tb0 = pw_tap(y, x + 1, 0, zrs0, H0, W0, ty0, tx0);
      }
      if (nz > 1) {
        ta1 = pw_tap(y, x, 0, zrs1, H0, W0, ty0, tx0);
        tb1 = pw_tap(y, x + 1, 0, zrs1, H0, W0, ty0, tx0);
      }
      float* __restrict__ op = ps->out + ((int64_t)b * nrows * H0 + y) * W0 + x;
      const int64_t cs = (int64_t)H0 * W0;
      auto sink = [&](int row, const float (&acc_)[2]) {
        float a0 = acc_[0], a1 = acc_[1];
        if (acc_in) {
          const float2 prev = *reinterpret_cast<const float2*>(op + row * cs);
          a0 += prev.x;
          a1 += prev.y;
        }
        if (nz > 0) {
          a0 += pw_tap_eval(zb0 + row * zs0, ta0);
          a1 += pw_tap_eval(zb0 + row * zs0, tb0);
        }
        if (nz > 1) {
          a0 += pw_tap_eval(zb1 + row * zs1, ta1);
          a1 += pw_tap_eval(zb1 + row * zs1, tb1);
        }
        if (fin) {
          const float sc = scale[row], sh = shift[row], al = alpha[row];
          a0 = csn_epi(a0, sc, sh, al);
          a1 = csn_epi(a1, sc, sh, al);
        }
        *reinterpret_cast<float2*>(op + row * cs) = make_float2(a0, a1);
      };
      pw_rows<MAXC_TOP, 2>(w, cin4, nrows, v, sink);
    }
  }
}

size_t csn_pw_lds_bytes(const PwArgs& a) {
  size_t fl = 0;
  for (int pi = 0; pi < a.nz_pass; ++pi) {
    const PwPass& ps = a.pass[pi];
    const size_t ring = (size_t)((PW_TY0 >> ps.r) + 2) * ((PW_TX0 >> ps.r) + 2);
    const size_t end = (size_t)ps.z_off + ring * ps.nrows;
    if (end > fl) fl = end;
  }
  return fl * sizeof(float);
}

#ifdef CSN_CPU_EMU
#define PW_ATTR(T, L)
#else
// instantiations that need more than the default 64 KiB of dynamic LDS (fuse1x1: 79 rows x (180+60) ring
// cells) are allowed the full 160 KiB of a CDNA4 CU; set once per instantiation
#define PW_ATTR(T, L)                                                                             \
  {                                                                                               \
    static bool done = false;                                                                     \
    if (!done && lds > 64 * 1024) {                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&goct_pw_kernel<T, L>),    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return (int)e;                                                         \
      done = true;                                                                                \
    }                                                                                             \
  }
#endif
#define PW_CASE(T, L)                                                                             \
  if (maxc_top == T && maxc_low == L) {                                                           \
    PW_ATTR(T, L)                                                                                 \
    CSN_LAUNCH((goct_pw_kernel<T, L>), grid, dim3(CSN_BLOCK), lds, stream, a);                    \
    return (int)hipGetLastError();                                                                \
  }

// maxc_top in {16, 32, 48, 64} (4: no branch-0 output); maxc_low in {32, 64} (4: unit without lower passes)
int csn_launch_pw(const PwArgs& a, int maxc_top, int maxc_low, void* stream) {
  const dim3 grid((a.W0 + PW_TX0 - 1) / PW_TX0, (a.H0 + PW_TY0 - 1) / PW_TY0, a.B);
  const size_t lds = csn_pw_lds_bytes(a);
  PW_CASE(16, 4) PW_CASE(32, 4) PW_CASE(48, 4) PW_CASE(64, 4)
  PW_CASE(4, 32) PW_CASE(16, 32) PW_CASE(32, 32) PW_CASE(48, 32) PW_CASE(64, 32)
  PW_CASE(4, 64) PW_CASE(16, 64) PW_CASE(32, 64) PW_CASE(48, 64) PW_CASE(64, 64)
  return -1;
}
