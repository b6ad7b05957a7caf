// k_misc.hip -- parameter packing, depthwise 3x3+BN+PReLU, 2x2 avg-pool, final bilinear x2.
//
// Reference semantics:
//   depthwise unit   SimplifiedGOctConvBR.forward  CSNet/model/csnet.py:838-851
//                    (Conv2dX100: conv2d(x, 100.0*w), CSNet/model/conv2d.py:104; BN eval; PReLU)
//   avg-pool 2x2/2   gOctaveConv stride==2 prologue  csnet.py:679-680
//   bilinear         F.interpolate(size=x.size()[2:], 'bilinear', align_corners=False)  csnet.py:382-385
#include "csn_kernels.h"
#include "csn_reduce.h"
#include "dw_core.h"

// ------------------------------------------------------------------------------------------ prep
__global__ __launch_bounds__(CSN_BLOCK) void csn_prep_kernel(const CsnPrepJob* __restrict__ jobs,
                                                              const float* __restrict__ arena,
                                                              float* __restrict__ packed) {
  const CsnPrepJob j = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  float* dst = packed + j.dst;
  const float eps = 1e-5f;  // nn.BatchNorm2d default
  switch (j.kind) {
    case CSN_PREP_COPY:      // p2 > 0: strided destination dst[i * p2 + p3] (interleaved per-channel records)
      for (int i = tid; i < j.n; i += CSN_BLOCK) dst[j.p2 > 0 ? (int64_t)i * j.p2 + j.p3 : i] = j.p0f * arena[j.src0 + i];
      break;
    case CSN_PREP_FILL:
      for (int i = tid; i < j.n; i += CSN_BLOCK) dst[i] = j.p0f;
      break;
    case CSN_PREP_BN_SCALE:
      for (int i = tid; i < j.n; i += CSN_BLOCK)
        dst[j.p2 > 0 ? (int64_t)i * j.p2 + j.p3 : i] = arena[j.src0 + i] / sqrtf(arena[j.src1 + i] + eps);
      break;
    case CSN_PREP_BN_SHIFT:
      for (int i = tid; i < j.n; i += CSN_BLOCK) {
        const float sc = arena[j.src0 + i] / sqrtf(arena[j.src1 + i] + eps);
        dst[j.p2 > 0 ? (int64_t)i * j.p2 + j.p3 : i] = arena[j.src2 + i] - arena[j.src3 + i] * sc;
      }
      break;
    case CSN_PREP_ROWS: {
      const int ncol = j.p1;
      const int tot = j.n * ncol;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int r = i / ncol, c = i - r * ncol;
        dst[(int64_t)r * j.p2 + j.p3 + c] = j.p0f * arena[j.src0 + (int64_t)r * j.p0 + c];
      }
    } break;
    case CSN_PREP_C3: {
      const int ncol = j.p1;
      const int tot = j.n * ncol * 9;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int t = i % 9;
        const int rc = i / 9;
        const int ci = rc % ncol, co = rc / ncol;
        dst[((int64_t)(co >> 3) * j.p2 + j.p3 + ci) * 72 + t * 8 + (co & 7)] =
            j.p0f * arena[j.src0 + ((int64_t)co * j.p0 + ci) * 9 + t];
      }
    } break;
    case CSN_PREP_EYE:
      for (int i = tid; i < j.n; i += CSN_BLOCK) dst[(int64_t)i * j.p2 + j.p3 + i] = j.p0f;
      break;
    case CSN_PREP_ROWS_T: {
      const int kk = j.p3 >> 24, col = j.p3 & 0xffffff, ncol = j.p1;
      const int tot = j.n * ncol;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int ci = i / ncol, c = i - ci * ncol;
        const int co = c / kk, t = c - co * kk;
        dst[(int64_t)ci * j.p2 + col + c] = j.p0f * arena[j.src0 + (int64_t)co * j.p0 + ci * kk + (kk - 1 - t)];
      }
    } break;
    case CSN_PREP_C3T:
    case CSN_PREP_C3T_T: {
      const int C = j.p1;
      const int tot = j.n * C * 9;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int t = i % 9;
        const int rc = i / 9;
        const int c = rc % C, r = rc / C;
        const float v = j.kind == CSN_PREP_C3T ? arena[j.src0 + (int64_t)r * j.p0 + c * 9 + t]
                                               : arena[j.src0 + (int64_t)c * j.p0 + r * 9 + (8 - t)];
        dst[(int64_t)r * j.p2 + j.p3 + (c >> 4) * 144 + t * 16 + (c & 15)] = j.p0f * v;
      }
    } break;
    case CSN_PREP_FLIP9:
      for (int i = tid; i < j.n; i += CSN_BLOCK) dst[i] = j.p0f * arena[j.src0 + (i / 9) * 9 + 8 - (i % 9)];
      break;
    case CSN_PREP_C3Q: {
      const int ncol = j.p1 * 9, t0 = j.p3 & 0xff, k0 = j.p3 >> 8;
      const int tot = j.n * ncol;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int r = i / ncol, c = i - r * ncol;
        dst[((int64_t)(k0 + c) * 4 + (r & 3)) * j.p2 + t0 + (r >> 2)] = j.p0f * arena[j.src0 + (int64_t)r * j.p0 + c];
      }
    } break;
    case CSN_PREP_MSDX: {
      const int tot = j.n * j.p0 * 9;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int t = i % 9, rc = i / 9;
        const int ci = rc % j.p0, co = rc / j.p0;
        dst[((int64_t)co * 9 + t) * j.p2 + ci] = j.p0f * arena[j.src0 + ((int64_t)co * j.p0 + ci) * 9 + (8 - t)];
      }
    } break;
    case CSN_PREP_C3Q_T: {
      const int ncol = j.p1 * 9, t0 = j.p3 & 0xff, k0 = j.p3 >> 8;
      const int tot = j.n * ncol;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int r = i / ncol, ct = i - r * ncol;
        const int c = ct / 9, t = ct - 9 * c;
        dst[((int64_t)(k0 + ct) * 4 + (r & 3)) * j.p2 + t0 + (r >> 2)] = j.p0f * arena[j.src0 + (int64_t)c * j.p0 + r * 9 + (8 - t)];
      }
    } break;
    case CSN_PREP_PW4_T: {
      const int ncol = j.p1, t0 = j.p3 & 0xff, k0 = j.p3 >> 8;
      const int tot = j.n * ncol;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int r = i / ncol, c = i - r * ncol;
        dst[((int64_t)(k0 + c) * 4 + (r & 3)) * j.p2 + t0 + (r >> 2)] = j.p0f * arena[j.src0 + (int64_t)c * j.p0 + r];
      }
    } break;
    case CSN_PREP_DWREC:
      for (int i = tid; i < j.n * 9; i += CSN_BLOCK) {
        const int c = i / 9, t = i - 9 * c;
        const float sc = arena[j.src1 + c] / sqrtf(arena[j.src2 + c] + eps);
        dst[(int64_t)c * j.p2 + j.p3 + t] = (j.p0f * arena[j.src0 + i]) * sc;
      }
      break;
    case CSN_PREP_PW4: {
      const int ncol = j.p1, t0 = j.p3 & 0xff, k0 = j.p3 >> 8;
      const int tot = j.n * ncol;
      for (int i = tid; i < tot; i += CSN_BLOCK) {
        const int r = i / ncol, c = i - r * ncol;
        dst[((int64_t)(k0 + c) * 4 + (r & 3)) * j.p2 + t0 + (r >> 2)] = j.p0f * arena[j.src0 + (int64_t)r * j.p0 + c];
      }
    } break;
    default:
      break;
  }
}

int csn_launch_prep(const CsnPrepJob* jobs_dev, int njobs, const float* arena, float* packed, void* stream) {
  if (njobs <= 0) return 0;
  CSN_LAUNCH(csn_prep_kernel, dim3(njobs), dim3(CSN_BLOCK), 0, stream, jobs_dev, arena, packed);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------- depthwise
// One block = one (image, channel) plane tile of LX*4 columns x NY*R rows.  A lane owns a 4-wide
// column strip and walks R rows with a rolling 3-row register window, so every input row is loaded
// once per lane (one b128 + the two edge dwords); the vertical halo between tiles hits L2.
// All loads are buffer loads on a resource bounded to the plane: rows above / below the image fall
// out of range and return 0 (= the conv's zero padding) without any branch, so the 12 loads of a
// 4-row chunk are issued back to back (a predicated load would be waited for at the join).
struct DwRow {
  float v[6];  // [0]=x0-1, [1..4]=x0..x0+3, [5]=x0+4
};

template <bool VEC, typename AT = float>
__device__ __forceinline__ DwRow dw_load_row(csn_buf rb, int y, int x0, int W, bool has_l, bool has_r) {
  DwRow r;
  constexpr unsigned E = (unsigned)sizeof(AT);
  const unsigned o = (unsigned)(y * W + x0) * E;   // y = -1 wraps to a huge offset -> out of range -> 0
  if (VEC) {
    const float4 c = csn_bufacc<AT>::ld4(rb, o, 0);
    r.v[1] = c.x; r.v[2] = c.y; r.v[3] = c.z; r.v[4] = c.w;
    const float l = csn_bufacc<AT>::ld1(rb, o - E, 0), rr = csn_bufacc<AT>::ld1(rb, o + 4u * E, 0);
    r.v[0] = has_l ? l : 0.f;
    r.v[5] = has_r ? rr : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float t = csn_bufacc<AT>::ld1(rb, o + (unsigned)(i - 1) * E, 0);
      const int xx = x0 + i - 1;
      r.v[i] = (xx >= 0 && xx < W) ? t : 0.f;
    }
  }
  return r;
}

// (the halo columns unmasked: for callers that mask after a transform anyway)
template <bool VEC, typename AT = float>
__device__ __forceinline__ DwRow dw_load_row_raw(csn_buf rb, int y, int x0, int W) {
  if (!VEC) return dw_load_row<false, AT>(rb, y, x0, W, true, true);
  DwRow r;
  constexpr unsigned E = (unsigned)sizeof(AT);
  const unsigned o = (unsigned)(y * W + x0) * E;
  const float4 c = csn_bufacc<AT>::ld4(rb, o, 0);
  r.v[1] = c.x; r.v[2] = c.y; r.v[3] = c.z; r.v[4] = c.w;
  r.v[0] = csn_bufacc<AT>::ld1(rb, o - E, 0);
  r.v[5] = csn_bufacc<AT>::ld1(rb, o + 4u * E, 0);
  return r;
}

// The same row as it comes out of the loads, NOT yet converted: a conversion (bfloat16 -> float is two shifts per dword) is the
// first use of the loaded registers and makes the wave wait for them, so a row that is to stay IN FLIGHT while the previous one is
// worked on (dw3x3_bwd_kernel, round 4) is held in this form and converted one loop trip later.
template <typename AT> struct DwRowRaw;
template <> struct DwRowRaw<float> { float4 c; float l, r; };
template <> struct DwRowRaw<csn_bf16> { uint2 c; unsigned short l, r; };
__device__ __forceinline__ DwRowRaw<float> dw_issue_row(csn_buf rb, int y, int x0, int W, float) {
  DwRowRaw<float> q;
  const unsigned o = (unsigned)(y * W + x0) * 4u;   // rows outside the plane: out of the bounded range -> 0
  q.c = csn_ld4(rb, o, 0); q.l = csn_ld1(rb, o - 4u, 0); q.r = csn_ld1(rb, o + 16u, 0);
  return q;
}
__device__ __forceinline__ DwRowRaw<csn_bf16> dw_issue_row(csn_buf rb, int y, int x0, int W, csn_bf16) {
  DwRowRaw<csn_bf16> q;
  const unsigned o = (unsigned)(y * W + x0) * 2u;
  q.c = csn_ld_u64(rb, o, 0); q.l = csn_ld_u16(rb, o - 2u, 0); q.r = csn_ld_u16(rb, o + 8u, 0);
  return q;
}
__device__ __forceinline__ DwRow dw_row_of(const DwRowRaw<float>& q) {
  DwRow r;
  r.v[0] = q.l; r.v[1] = q.c.x; r.v[2] = q.c.y; r.v[3] = q.c.z; r.v[4] = q.c.w; r.v[5] = q.r;
  return r;
}
__device__ __forceinline__ DwRow dw_row_of(const DwRowRaw<csn_bf16>& q) {
  DwRow r;
  r.v[0] = csn_bf2f(q.l);
  r.v[1] = csn_bits_f(q.c.x << 16); r.v[2] = csn_bits_f(q.c.x & 0xffff0000u);
  r.v[3] = csn_bits_f(q.c.y << 16); r.v[4] = csn_bits_f(q.c.y & 0xffff0000u);
  r.v[5] = csn_bf2f(q.r);
  return r;
}

// Halo columns without loads (round 5).  Measured on the bf16 step (lease r5z): this kernel with all of its arithmetic removed is
// 10 % faster, with its eight 16-bit halo loads per trip removed (of twelve loads) 20 % -- it is bound by the NUMBER of vector-memory
// instructions, not by bytes or arithmetic.  When a row of lanes sits inside one wave (launch geometry: lanes per row = a power of two
// <= 64, one tile per row) the value left / right of a lane's four columns is the neighbouring lane's last / first own value: one DPP
// move (wave_shr:1 / wave_shl:1) on the loaded register instead of a load.  Lanes at a row's ends read a lane of another row or an
// idle lane: those are the positions has_l / has_r mask to zero anyway.  (The CPU emulator runs lanes one after the other: it loads.)
// (csn_from_lane_below / _above: csn_device.h)
// a row as loaded and converted, the halo columns from the neighbouring lanes (no masks: the callers mask or transform anyway)
template <typename AT>
__device__ __forceinline__ DwRow dw_load_row_xl(csn_buf rb, int y, int x0, int W) {
#ifndef CSN_EMU_SEQ
  DwRow r;
  const float4 c = csn_bufacc<AT>::ld4(rb, (unsigned)(y * W + x0) * (unsigned)sizeof(AT), 0);
  r.v[1] = c.x; r.v[2] = c.y; r.v[3] = c.z; r.v[4] = c.w;
  r.v[0] = csn_bits_f(csn_from_lane_below(csn_f_bits(c.w)));
  r.v[5] = csn_bits_f(csn_from_lane_above(csn_f_bits(c.x)));
  return r;
#else
  return dw_load_row_raw<true, AT>(rb, y, x0, W);
#endif
}

// x = PReLU(z * sc + sh) of a loaded row of six values, zero outside the plane -- the lean form (round 5, from dw_core.h's findings:
// the bf16 train step's depthwise kernels ARE bound by their vector instructions).  Bit for bit csn_epi's values: PReLU as
// v_med3(y, alpha y, +-inf) (= max for alpha <= 1, min above) instead of compare + select; a row outside the plane loads zeros, and
// with scale = shift = 0 for that row y = 0 without a mask per element; only the two halo columns are masked.  ~16 instead of ~32
// vector instructions per row.
__device__ __forceinline__ DwRow dw_bn_row(DwRow r, bool rowin, bool has_l, bool has_r, float sc, float sh, float al) {
  const float scr = rowin ? sc : 0.f, shr = rowin ? sh : 0.f;
  const float lim = al <= 1.f ? __builtin_inff() : -__builtin_inff();
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float y = fmaf(r.v[i], scr, shr);
    r.v[i] = dw_med3(y, al * y, lim);
  }
  if (!has_l) r.v[0] = 0.f;
  if (!has_r) r.v[5] = 0.f;
  return r;
}

// row of x = PReLU(z * sc + sh) from a row of the producer's raw output z (see DwBranch::in_scale); outside the plane: 0
template <bool VEC, typename AT>
__device__ __forceinline__ DwRow dw_load_row_bn(csn_buf rb, int y, int H, int x0, int W, bool has_l, bool has_r, float sc,
                                                float sh, float al, float* zc = nullptr) {
  DwRow r = dw_load_row_raw<VEC, AT>(rb, y, x0, W);
  if (zc) { zc[0] = r.v[1]; zc[1] = r.v[2]; zc[2] = r.v[3]; zc[3] = r.v[4]; }   // the raw centre values z
  const bool rowin = y >= 0 && y < H;
  if (VEC) return dw_bn_row(r, rowin, has_l, has_r, sc, sh, al);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const bool colin = i == 0 ? has_l : (i == 5 ? has_r : (VEC || x0 + i - 1 < W));
    const float v = csn_epi(r.v[i], sc, sh, al);
    r.v[i] = (rowin && colin) ? v : 0.f;
  }
  return r;
}

// STATS: also accumulate sum / sum of squares of the values as STORED (train mode: the raw conv output z whose batch
// statistics the BatchNorm that follows needs -- saves bn_stats_kernel's pass over z)
// IDENT: the epilogue is the identity (train mode: the raw sums are stored, BatchNorm runs on the batch statistics afterwards; the
// backward kernel's dx) -- skipped instead of evaluated with scale 1 / shift 0 / slope 1 (four instructions per value of kernels
// whose instruction and memory times add up, round 4)
template <bool VEC, typename AT = float, bool STATS = false, bool IDENT = false>
__device__ __forceinline__ void dw_emit(AT* __restrict__ op, int y, int yend, int x0, int W,
                                        const float (&w)[9], float sc, float sh, float al, const DwRow& top,
                                        const DwRow& mid, const DwRow& bot, double* st = nullptr, float* oret = nullptr) {
  if (y >= yend) return;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float acc = w[0] * top.v[j];
    acc = fmaf(w[1], top.v[j + 1], acc);
    acc = fmaf(w[2], top.v[j + 2], acc);
    acc = fmaf(w[3], mid.v[j], acc);
    acc = fmaf(w[4], mid.v[j + 1], acc);
    acc = fmaf(w[5], mid.v[j + 2], acc);
    acc = fmaf(w[6], bot.v[j], acc);
    acc = fmaf(w[7], bot.v[j + 1], acc);
    acc = fmaf(w[8], bot.v[j + 2], acc);
    o[j] = IDENT ? acc : csn_epi(acc, sc, sh, al);
  }
  AT* q = op + (int64_t)y * W + x0;
  if (STATS) {
    if (sizeof(AT) == 2) {
      // bfloat16 storage (round 4): the stored values have 8 significant bits, so v * v is exact in fp32 and the sums of the row's
      // four values / squares are exact or off by one fp32 rounding -- fp32 over the row, fp64 across rows (the per-element fp64
      // form cost 12 double-rate operations per row of a kernel that is instruction bound)
      float s = 0.f, q2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (VEC || x0 + j < W) {
          const float v = csn_bf2f(csn_f2bf(o[j]));
          s += v;
          q2 = fmaf(v, v, q2);
        }
      st[0] += (double)s;
      st[1] += (double)q2;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (VEC || x0 + j < W) {
          const double v = (double)o[j];
          st[0] += v;
          st[1] += v * v;
        }
    }
  }
  if (oret) {   // the values as STORED
#pragma unroll
    for (int j = 0; j < 4; ++j) oret[j] = sizeof(AT) == 2 ? csn_bf2f(csn_f2bf(o[j])) : o[j];
  }
  if (VEC) {
    act_st4(q, make_float4(o[0], o[1], o[2], o[3]));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x0 + j < W) act_st(q + j, o[j]);
  }
}

#ifndef DW_FWD_PREFETCH
#define DW_FWD_PREFETCH 0   // measured (round 4, same lease): 4.06 -> 4.30 ms per bf16 step with the next chunk in flight -- this kernel
                            // already issues the 12 loads of a four-row chunk back to back; the backward kernel gains (9.70 -> 8.94 ms)
#endif
// INBN (train mode, with STATS): the input is formed on load from the producer's raw output and its plane sums are taken
// XL: halo columns from the neighbouring lanes (see csn_from_lane_below; launches whose rows of lanes sit inside one wave)
template <bool VEC, typename AT, bool STATS, bool INBN = false, bool XL = false>
__global__ __launch_bounds__(CSN_BLOCK) void dw3x3_bn_prelu_kernel(DwArgs a) {
  CSN_DYN_SMEM(double, sm);   // STATS only
  int bid = blockIdx.x;
  int k = 0;
  if (a.nbr > 1 && bid >= a.br[0].blk_end) k = 1;
  if (a.nbr > 2 && bid >= a.br[1].blk_end) k = 2;
  const DwBranch br = a.br[k];
  if (k > 0) bid -= a.br[k - 1].blk_end;
  const int tiles = br.tiles_x * br.tiles_y;
  const int tile = bid % tiles;
  const int pc = bid / tiles;  // b*C + c
  const int c = pc % br.C;
  const int tx = tile % br.tiles_x, ty = tile / br.tiles_x;
  const int tid = threadIdx.x;
  const int lx = tid % br.LX, ly = tid / br.LX;
  const int H = br.H, W = br.W;
  const int x0 = (tx * br.LX + lx) * 4;
  const int y0 = (ty * br.NY + ly) * br.R;
  const bool active = ly < br.NY && x0 < W && y0 < H;
  double st[3] = {0.0, 0.0, 0.0};   // sum / sum of squares of the stored output; INBN: sum of the input over the lane's own rows
  if (active) {
    const csn_buf rb = csn_make_buf_n(act_cast<AT>(br.in) + (int64_t)pc * H * W, (unsigned)(H * W) * (unsigned)sizeof(AT));
    AT* __restrict__ op = act_cast<AT>(br.out) + (int64_t)pc * H * W;
    float w[9];
    csn_cfp w9 = csn_const(br.w9);
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w9[c * 9 + i];
    const float sc = csn_const(br.scale)[c], sh = csn_const(br.shift)[c], al = csn_const(br.alpha)[c];
    const bool has_l = x0 > 0, has_r = x0 + 4 < W;
    const int yend = min(y0 + br.R, H);
    float isc = 1.f, ish = 0.f, ial = 1.f;
    if (INBN) { isc = csn_const(br.in_scale)[c]; ish = csn_const(br.in_shift)[c]; ial = csn_const(br.in_alpha)[c]; }
    float gsum = 0.f;   // <= 16 rows x 4 values per lane: fp32, then fp64 across the block
    auto load_in = [&](int y) {
      if (XL) {
        DwRow r = dw_load_row_xl<AT>(rb, y, x0, W);
        if (INBN) {
          r = dw_bn_row(r, y >= 0 && y < H, has_l, has_r, isc, ish, ial);
          if (y >= y0 && y < yend) gsum += (r.v[1] + r.v[2]) + (r.v[3] + r.v[4]);
        } else {
          if (!has_l) r.v[0] = 0.f;
          if (!has_r) r.v[5] = 0.f;
        }
        return r;
      }
      if (!INBN) return dw_load_row<VEC, AT>(rb, y, x0, W, has_l, has_r);
      const DwRow r = dw_load_row_bn<VEC, AT>(rb, y, H, x0, W, has_l, has_r, isc, ish, ial);
      if (y >= y0 && y < yend) gsum += (r.v[1] + r.v[2]) + (r.v[3] + r.v[4]);   // (columns past W are zero)
      return r;
    };

    // the row as loaded -> the row the convolution sees (train mode, INBN: PReLU(BN(z)) of the producer's z, zero outside the plane)
    auto finish_in = [&](DwRow r, int y) {
      if (INBN) {
        r = dw_bn_row(r, y >= 0 && y < H, has_l, has_r, isc, ish, ial);
        if (y >= y0 && y < yend) gsum += (r.v[1] + r.v[2]) + (r.v[3] + r.v[4]);
        return r;
      }
      if (!has_l) r.v[0] = 0.f;
      if (!has_r) r.v[5] = 0.f;
      return r;
    };
    DwRow r0 = load_in(y0 - 1);
    DwRow r1 = load_in(y0);
    // Round 4 (train-mode kernels, vector path): the four rows of the NEXT chunk are issued before the current chunk is converted and
    // used -- a whole loop trip in flight instead of being waited for where they are issued
    constexpr bool PF = VEC && STATS && DW_FWD_PREFETCH;
    DwRowRaw<AT> q0, q1, q2, q3;
    if (PF) {
      q0 = dw_issue_row(rb, y0 + 1, x0, W, AT()); q1 = dw_issue_row(rb, y0 + 2, x0, W, AT());
      q2 = dw_issue_row(rb, y0 + 3, x0, W, AT()); q3 = dw_issue_row(rb, y0 + 4, x0, W, AT());
    }
    for (int y = y0; y < yend; y += 4) {
      DwRow n0, n1, n2, n3;
      if (PF) {
        const DwRowRaw<AT> p0 = dw_issue_row(rb, y + 5, x0, W, AT()), p1 = dw_issue_row(rb, y + 6, x0, W, AT());
        const DwRowRaw<AT> p2 = dw_issue_row(rb, y + 7, x0, W, AT()), p3 = dw_issue_row(rb, y + 8, x0, W, AT());
#ifndef CSN_CPU_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        n0 = finish_in(dw_row_of(q0), y + 1); n1 = finish_in(dw_row_of(q1), y + 2);
        n2 = finish_in(dw_row_of(q2), y + 3); n3 = finish_in(dw_row_of(q3), y + 4);
        q0 = p0; q1 = p1; q2 = p2; q3 = p3;
      } else {
        n0 = load_in(y + 1); n1 = load_in(y + 2); n2 = load_in(y + 3); n3 = load_in(y + 4);
      }
      dw_emit<VEC, AT, STATS, STATS>(op, y, yend, x0, W, w, sc, sh, al, r0, r1, n0, st);
      dw_emit<VEC, AT, STATS, STATS>(op, y + 1, yend, x0, W, w, sc, sh, al, r1, n0, n1, st);
      dw_emit<VEC, AT, STATS, STATS>(op, y + 2, yend, x0, W, w, sc, sh, al, n0, n1, n2, st);
      dw_emit<VEC, AT, STATS, STATS>(op, y + 3, yend, x0, W, w, sc, sh, al, n1, n2, n3, st);
      r0 = n2;
      r1 = n3;
    }
    st[2] = (double)gsum;
  }
  if (STATS) {   // one (sum, sum of squares) partial per (image, tile) of the channel: slab = b * tiles + tile
    if (INBN) {
      bn_block_sum_n<3>(st, sm);
    } else {
      double s2[2] = {st[0], st[1]};
      bn_block_sum_n<2>(s2, sm);
      st[0] = s2[0]; st[1] = s2[1];
    }
    if (tid == 0) {
      const int b = pc / br.C;
      double* o = br.stats + ((int64_t)c * CSN_BN_NSLAB + (int64_t)b * tiles + tile) * 2;
      o[0] = st[0];
      o[1] = st[1];
      if (INBN) br.gapin[(int64_t)c * CSN_BN_NSLAB + (int64_t)b * tiles + tile] = st[2];
    }
  }
}

// -------------------------------------------------------------- depthwise backward: input AND weight gradient
// One pass over dz and x for both gradients of a depthwise unit (conv2d.py:104: y = conv(x, 100 w)):
//     dx[p]  = sum_t 100 w[8 - t] dz[p + off(t)]              (the forward kernel with flipped taps)
//     dW[t] += dz[p] * x[p + off(t)]                          (x100 in the finaliser)
// Same tile / lane mapping as dw3x3_bn_prelu_kernel; a lane keeps rolling three-row windows of dz AND of x, so every row of
// either tensor is loaded once per lane (the stand-alone weight-gradient kernel re-read dz and gathered ten values per quad).
// The nine per-lane sums (<= 64 terms, fp32) are reduced per block in fp64: one partial per (channel, image, tile).
// BNF: the BatchNorm backward's apply pass is fused in -- dz is formed per loaded element from dy (+ dy2) and the saved z
// with the channel's (block-uniform) tables, bit for bit bn_bwd_apply_kernel's arithmetic; rows / columns outside the plane
// are masked to zero AFTER the formula (z = dy = 0 does not give dz = 0).  Saves the write and the re-read of dz (round 3).
// XBN: the unit's forward input x was never stored: it is formed on load from the producer's raw output (DwBranch::in_scale)
#ifndef DW_BWD_PREFETCH
#define DW_BWD_PREFETCH 1   // 0: A/B builds (loads waited for where they are issued, round 3)
#endif
template <bool VEC, typename AT, bool BNF, bool XBN = false>
__global__ __launch_bounds__(CSN_BLOCK) void dw3x3_bwd_kernel(DwArgs a) {
  CSN_DYN_SMEM(double, sm);
  int bid = blockIdx.x;
  int k = 0;
  if (a.nbr > 1 && bid >= a.br[0].blk_end) k = 1;
  if (a.nbr > 2 && bid >= a.br[1].blk_end) k = 2;
  const DwBranch br = a.br[k];
  if (k > 0) bid -= a.br[k - 1].blk_end;
  const int tiles = br.tiles_x * br.tiles_y;
  const int tile = bid % tiles;
  const int pc = bid / tiles;  // b*C + c
  const int c = pc % br.C;
  const int tx = tile % br.tiles_x, ty = tile / br.tiles_x;
  const int tid = threadIdx.x;
  const int lx = tid % br.LX, ly = tid / br.LX;
  const int H = br.H, W = br.W;
  const int x0 = (tx * br.LX + lx) * 4;
  const int y0 = (ty * br.NY + ly) * br.R;
  const bool active = ly < br.NY && x0 < W && y0 < H;
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  double rs[3] = {0.0, 0.0, 0.0};
  float rf[3] = {0.f, 0.f, 0.f};   // per lane: <= 64 terms in fp32 (like the nine weight-gradient sums), fp64 across the block
  if (active) {
    const unsigned nb = (unsigned)(H * W) * (unsigned)sizeof(AT);
    const csn_buf gb = csn_make_buf_n(act_cast<AT>(br.in) + (int64_t)pc * H * W, nb);     // dz (BNF: dy of the first consumer)
    const csn_buf xb = csn_make_buf_n(act_cast<AT>(br.xin) + (int64_t)pc * H * W, nb);    // x
    AT* __restrict__ op = act_cast<AT>(br.out) + (int64_t)pc * H * W;
    float w[9];
    csn_cfp w9 = csn_const(br.w9);
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w9[c * 9 + i];
    const bool has_l = x0 > 0, has_r = x0 + 4 < W;
    // BNF: the row of dz from dy (+ dy2) and z
    const bool has2 = BNF && br.dy2 != nullptr;
    const csn_buf zb = csn_make_buf_n(act_cast<AT>(BNF ? br.zraw : br.in) + (int64_t)pc * H * W, nb);
    const csn_buf g2b = csn_make_buf_n(act_cast<AT>(has2 ? br.dy2 : br.in) + (int64_t)pc * H * W, nb);
    // dz = gi (dbn - m1 - (z - mu) invstd m2),  dbn = dy (bn > 0 ? 1 : alpha)   [bn_bwd_apply_kernel]
    //    = dy * (bn > 0 ? gi : gi alpha) - (B z + A),   B = gi invstd m2,  A = gi (m1 - mu invstd m2):  five operations per element
    float bsc = 0.f, bsh = 0.f, bgi = 0.f, bga = 0.f, bA = 0.f, bB = 0.f;
    if (BNF) {
      bsc = csn_const(br.bn_scale)[c]; bsh = csn_const(br.bn_shift)[c];
      const float bal = csn_const(br.bn_alpha)[c], bmu = csn_const(br.bn_mean)[c], bis = csn_const(br.bn_invstd)[c];
      const float bm1 = csn_const(br.bn_m1m2)[2 * c], bm2 = csn_const(br.bn_m1m2)[2 * c + 1];
      bgi = csn_const(br.bn_gamma)[c] * bis;
      bga = bgi * bal;
      bB = bgi * (bis * bm2);
      bA = bgi * (bm1 - bmu * (bis * bm2));
    }
    // dz row from the rows of dy, z (and dy2) as loaded
    auto finish_g = [&](DwRow g, const DwRow& z, const DwRow& e, int y) {
      if (!BNF) {
        if (!has_l) g.v[0] = 0.f;
        if (!has_r) g.v[5] = 0.f;
        return g;
      }
      if (has2) {
#pragma unroll
        for (int i = 0; i < 6; ++i) g.v[i] += e.v[i];
      }
      const bool rowin = y >= 0 && y < H;
      if (VEC) {   // a row outside the plane loads dy = z = 0: with A = 0 for that row dz = 0 without a mask per element (round 5)
        const float bAr = rowin ? bA : 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float bn = fmaf(z.v[i], bsc, bsh);
          const float t = fmaf(bB, z.v[i], bAr);
          g.v[i] = fmaf(g.v[i], bn > 0.f ? bgi : bga, -t);
        }
        if (!has_l) g.v[0] = 0.f;
        if (!has_r) g.v[5] = 0.f;
        return g;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float bn = fmaf(z.v[i], bsc, bsh);
        const float t = fmaf(bB, z.v[i], bA);
        const float dzv = fmaf(g.v[i], bn > 0.f ? bgi : bga, -t);
        const bool colin = i == 0 ? has_l : (i == 5 ? has_r : (VEC || x0 + i - 1 < W));
        g.v[i] = (rowin && colin) ? dzv : 0.f;
      }
      return g;
    };
    auto load_g = [&](int y) {
      if (!BNF) return dw_load_row<VEC, AT>(gb, y, x0, W, has_l, has_r);
      DwRow g = dw_load_row_raw<VEC, AT>(gb, y, x0, W);
      const DwRow z = dw_load_row_raw<VEC, AT>(zb, y, x0, W);
      if (has2) {
        const DwRow e = dw_load_row_raw<VEC, AT>(g2b, y, x0, W);
#pragma unroll
        for (int i = 0; i < 6; ++i) g.v[i] += e.v[i];
      }
      const bool rowin = y >= 0 && y < H;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float bn = fmaf(z.v[i], bsc, bsh);
        const float t = fmaf(bB, z.v[i], bA);
        const float dzv = fmaf(g.v[i], bn > 0.f ? bgi : bga, -t);
        const bool colin = i == 0 ? has_l : (i == 5 ? has_r : (VEC || x0 + i - 1 < W));
        g.v[i] = (rowin && colin) ? dzv : 0.f;
      }
      return g;
    };
    float isc = 1.f, ish = 0.f, ial = 1.f;
    if (XBN) { isc = csn_const(br.in_scale)[c]; ish = csn_const(br.in_shift)[c]; ial = csn_const(br.in_alpha)[c]; }
    float imu = 0.f, iis = 0.f;
    if (XBN) { imu = csn_const(br.in_mean)[c]; iis = csn_const(br.in_invstd)[c]; }
    auto load_x = [&](int y, float* zc) {
      if (XBN) return dw_load_row_bn<VEC, AT>(xb, y, H, x0, W, has_l, has_r, isc, ish, ial, zc);
      return dw_load_row<VEC, AT>(xb, y, x0, W, has_l, has_r);
    };
    // the row of x from the row as loaded (XBN: the producer's z -> PReLU(BN(z)); zc keeps the raw centre values)
    auto finish_x = [&](DwRow r, int y, float* zc) {
      if (XBN) {
        if (zc) { zc[0] = r.v[1]; zc[1] = r.v[2]; zc[2] = r.v[3]; zc[3] = r.v[4]; }
        return dw_bn_row(r, y >= 0 && y < H, has_l, has_r, isc, ish, ial);
      }
      if (!has_l) r.v[0] = 0.f;
      if (!has_r) r.v[5] = 0.f;
      return r;
    };
    float zc1[4] = {0.f, 0.f, 0.f, 0.f}, zc2[4];   // XBN: raw z of the producer at the centre columns of rows y, y + 1
    DwRow g0 = load_g(y0 - 1);
    DwRow g1 = load_g(y0);
    DwRow u0 = load_x(y0 - 1, nullptr);
    DwRow u1 = load_x(y0, zc1);
    const int yend = min(y0 + br.R, H);
    // (rotating three register sets through an unrolled-by-three loop instead of copying rows: 17 % fewer VALU instructions per row
    // (220 instead of 264), but 182 instead of 102 VGPRs = two waves per SIMD instead of four: 42.1 instead of 40.2 ms per bf16 step,
    // and 42.8 ms bounded to 168 VGPRs; bounding the loop as it is to 96 / 80 VGPRs for five / six waves spills: 41.8 / 63 ms --
    // round 4, gpurun_out/r4k, r4l.  Two rows of loads in flight instead of one (126 VGPRs, still four waves): 40.1-40.8 instead of
    // 39.3-39.4 ms, r4m.  Neither fewer instructions at lower occupancy nor more loads in flight at the same occupancy helps;
    // five waves (fp32 row sums + a 96-register bound, one reload per row) measure the same as four, r4s.  By the ISA listing the
    // loop is 264 VALU instructions per row of four pixels = ~6.5 ms of pure issue time per bf16 step against 9.0 ms measured.)
    // Round 4: on the vector path the loads of row y + 2 are issued BEFORE row y + 1 is converted and used, i.e. they are in flight
    // during a whole loop trip (they used to be waited for right where they were issued: every trip paid a full memory round trip
    // with only the other waves of the SIMD to cover it).  Rows past the tile / the plane are fetched like any other (zeros past the
    // plane: bounded resources) and never used.
    constexpr bool PF = VEC && DW_BWD_PREFETCH;
    DwRowRaw<AT> rg, rz, re, rx;
    if (PF) {
      rg = dw_issue_row(gb, y0 + 1, x0, W, AT());
      rx = dw_issue_row(xb, y0 + 1, x0, W, AT());
      if (BNF) rz = dw_issue_row(zb, y0 + 1, x0, W, AT());
      if (BNF && has2) re = dw_issue_row(g2b, y0 + 1, x0, W, AT());
    }
    for (int y = y0; y < yend; ++y) {
      DwRow g2, u2;
      if (PF) {
        const DwRowRaw<AT> ng = dw_issue_row(gb, y + 2, x0, W, AT());
        const DwRowRaw<AT> nx = dw_issue_row(xb, y + 2, x0, W, AT());
        DwRowRaw<AT> nz = rz, ne = re;
        if (BNF) nz = dw_issue_row(zb, y + 2, x0, W, AT());
        if (BNF && has2) ne = dw_issue_row(g2b, y + 2, x0, W, AT());
#ifndef CSN_CPU_EMU
        __builtin_amdgcn_sched_barrier(0);   // the loads above stay above the conversions below
#endif
        g2 = finish_g(dw_row_of(rg), BNF ? dw_row_of(rz) : DwRow(), (BNF && has2) ? dw_row_of(re) : DwRow(), y + 1);
        u2 = finish_x(dw_row_of(rx), y + 1, zc2);
        rg = ng; rx = nx; rz = nz; re = ne;
      } else {
        g2 = load_g(y + 1);
        u2 = load_x(y + 1, zc2);
      }
      float dxv[4];
      dw_emit<VEC, AT, false, true>(op, y, yend, x0, W, w, 1.f, 0.f, 1.f, g0, g1, g2, nullptr, XBN ? dxv : nullptr);
      if (XBN) {   // the producer's BatchNorm-backward sums (bn_bwd_reduce_kernel's arithmetic): dy = the dx just stored
        // fp32 over the row's four values, fp64 across the rows of the lane (round 4: the sums -- the PReLU slope's gradient above
        // all -- cancel heavily on the shipped checkpoint; 64 fp32 terms per lane put one tensor at 2.1e-4 of the unit-local bound
        // 2e-4 on the device)
        rf[0] = rf[1] = rf[2] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (VEC || x0 + j < W) {
            const float z = zc1[j], dy = dxv[j];
            const float bn = fmaf(z, isc, ish);
            const float dbn = bn > 0.f ? dy : ial * dy;
            rf[0] += dbn;
            rf[1] = fmaf(dbn, (z - imu) * iis, rf[1]);
            rf[2] = fmaf(bn > 0.f ? 0.f : dy, bn, rf[2]);
          }
        rs[0] += (double)rf[0]; rs[1] += (double)rf[1]; rs[2] += (double)rf[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) zc1[j] = zc2[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = (VEC || x0 + j < W) ? g1.v[j + 1] : 0.f;
        s[0] = fmaf(g, u0.v[j], s[0]); s[1] = fmaf(g, u0.v[j + 1], s[1]); s[2] = fmaf(g, u0.v[j + 2], s[2]);
        s[3] = fmaf(g, u1.v[j], s[3]); s[4] = fmaf(g, u1.v[j + 1], s[4]); s[5] = fmaf(g, u1.v[j + 2], s[5]);
        s[6] = fmaf(g, u2.v[j], s[6]); s[7] = fmaf(g, u2.v[j + 1], s[7]); s[8] = fmaf(g, u2.v[j + 2], s[8]);
      }
      g0 = g1; g1 = g2;
      u0 = u1; u1 = u2;
    }
  }
  double sv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) sv[t] = (double)s[t];
  bn_block_sum_n<9>(sv, sm);
  if (XBN) bn_block_sum_n<3>(rs, sm);
  if (tid == 0) {
    const int b = pc / br.C;
    double* o = br.stats + ((int64_t)c * CSN_BN_NSLAB + (int64_t)b * tiles + tile) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) o[t] = sv[t];
    if (XBN) {
      double* q = br.bnred + ((int64_t)c * CSN_BN_NSLAB + (int64_t)b * tiles + tile) * 3;
      q[0] = rs[0]; q[1] = rs[1]; q[2] = rs[2];
    }
  }
}

// ---- the fully fused form of the kernel above (BNF + XBN, whole four-pixel lanes) on packed pairs (round 5) ----
// The bf16 train step spends a quarter of its time here (60 launches) and the kernel is bound by its vector instructions (85 %
// issue utilisation at 3.3 TB/s, profiles/r4_notes.md): of the 264 per row of four pixels above, 74 are register moves (two
// three-row windows of six values shifted down every row, pairs re-formed for the packed FMAs).  Here
//   * dx is accumulated in SCATTER form: a row of dz, when it arrives, adds its top / middle / bottom tap rows to the outputs
//     of the rows below / at / above it (two pair accumulators per open output row: the roll is the FMA's destination), in the
//     same order of operations per output as the gather form: dx is bit-identical;
//   * the weight gradient pairs the arriving row of x with the CENTRE pairs of three rows of dz (2 pairs per row kept instead of
//     a three-row window of x), summed as even / odd columns per tap (18 packed FMAs) and added at the end;
//   * dx of row r is final when dz of row r + 1 has arrived -- the same trip x of row r is formed in, whose pre-activation values
//     the producer's BatchNorm-backward sums need: nothing is carried for them.
template <bool XL>
__device__ __forceinline__ DwRowRaw<float> dwx_issue(csn_buf rb, int y, int x0, int W, float tag) {
#ifndef CSN_EMU_SEQ
  if (XL) {
    DwRowRaw<float> q;
    q.c = csn_ld4(rb, (unsigned)(y * W + x0) * 4u, 0); q.l = q.r = 0.f;
    return q;
  }
#endif
  return dw_issue_row(rb, y, x0, W, tag);
}
template <bool XL>
__device__ __forceinline__ DwRowRaw<csn_bf16> dwx_issue(csn_buf rb, int y, int x0, int W, csn_bf16 tag) {
#ifndef CSN_EMU_SEQ
  if (XL) {
    DwRowRaw<csn_bf16> q;
    q.c = csn_ld_u64(rb, (unsigned)(y * W + x0) * 2u, 0); q.l = q.r = 0;
    return q;
  }
#endif
  return dw_issue_row(rb, y, x0, W, tag);
}
template <bool XL>
__device__ __forceinline__ DwRow dwx_row_of(const DwRowRaw<float>& q) {
  DwRow r = dw_row_of(q);
#ifndef CSN_EMU_SEQ
  if (XL) {
    r.v[0] = csn_bits_f(csn_from_lane_below(csn_f_bits(q.c.w)));
    r.v[5] = csn_bits_f(csn_from_lane_above(csn_f_bits(q.c.x)));
  }
#endif
  return r;
}
template <bool XL>
__device__ __forceinline__ DwRow dwx_row_of(const DwRowRaw<csn_bf16>& q) {
#ifndef CSN_EMU_SEQ
  if (XL) {
    DwRow r;
    r.v[0] = csn_bits_f(csn_from_lane_below(q.c.y) & 0xffff0000u);   // the neighbour's fourth value = high half of its second dword
    r.v[1] = csn_bits_f(q.c.x << 16); r.v[2] = csn_bits_f(q.c.x & 0xffff0000u);
    r.v[3] = csn_bits_f(q.c.y << 16); r.v[4] = csn_bits_f(q.c.y & 0xffff0000u);
    r.v[5] = csn_bits_f(csn_from_lane_above(q.c.x) << 16);           // ... first value = low half of its first dword
    return r;
  }
#endif
  return dw_row_of(q);
}
// dy + dy2 of one row as loaded
template <bool XL, typename AT>
__device__ __forceinline__ DwRow dwx_dy_row(const DwRowRaw<AT>& qg, const DwRowRaw<AT>& qe, bool has2) {
  DwRow g = dwx_row_of<XL>(qg);
  if (has2) {
    const DwRow e = dwx_row_of<XL>(qe);
#pragma unroll
    for (int i = 0; i < 6; ++i) g.v[i] += e.v[i];
  }
  return g;
}
// dz of one row from dy (+ dy2) and z; rows / columns outside the plane -> 0
__device__ __forceinline__ DwRow2 dwx_dz_row(DwRow g, const DwRow& z, bool has_l, bool has_r, float bsc, float bsh, float bgi, float bga,
                                             float bB, float bAr) {   // g: dy (+ dy2); bAr: A of the row, 0 for a row outside the
                                                                      // plane (it loads dy = z = 0: dz = 0)
#pragma unroll
  for (int i = 0; i < 6; i += 2) {
    const csn_v2 z2 = csn_mk2(z.v[i], z.v[i + 1]);
    const csn_v2 bn = csn_fma2(bsc, z2, csn_mk2(bsh, bsh));
    const csn_v2 t = csn_fma2(bB, z2, csn_mk2(bAr, bAr));
    const csn_v2 sl = csn_mk2(bn[0] > 0.f ? bgi : bga, bn[1] > 0.f ? bgi : bga);
    const csn_v2 d = csn_fmav2(csn_mk2(g.v[i], g.v[i + 1]), sl, csn_mk2(-t[0], -t[1]));
    g.v[i] = d[0]; g.v[i + 1] = d[1];
  }
  if (!has_l) g.v[0] = 0.f;
  if (!has_r) g.v[5] = 0.f;
  return dw_row2_regs(g.v[0], g.v[1], g.v[2], g.v[3], g.v[4], g.v[5]);
}

// x = PReLU(z isc + ish) of one row of the producer's raw output; pre[4]: z isc + ish of the lane's four own columns
__device__ __forceinline__ DwRow2 dwx_x_row(const DwRow& zraw, bool has_l, bool has_r, float sc, float sh, float ial, float ilim,
                                            float (&pre)[4]) {   // sc = sh = 0 for a row outside the plane
  float v[6];
#pragma unroll
  for (int i = 0; i < 6; i += 2) {
    const csn_v2 y = csn_fma2(sc, csn_mk2(zraw.v[i], zraw.v[i + 1]), csn_mk2(sh, sh));
    const csn_v2 x = dw_prelu2(y, ial, ilim);
    v[i] = x[0]; v[i + 1] = x[1];
    if (i == 0) pre[0] = y[1];
    if (i == 2) { pre[1] = y[0]; pre[2] = y[1]; }
    if (i == 4) pre[3] = y[0];
  }
  if (!has_l) v[0] = 0.f;
  if (!has_r) v[5] = 0.f;
  return dw_row2_regs(v[0], v[1], v[2], v[3], v[4], v[5]);
}

// the nine weight-gradient taps' even / odd column sums: centre pairs (g1,g2) (g3,g4) of one row of dz x one row of x, tap row ty
__device__ __forceinline__ void dwx_wgrad(csn_v2 gb, csn_v2 gd, const DwRow2& u, csn_v2 (&s2)[9], int ty) {
  s2[3 * ty] = csn_fmav2(gb, u.a, s2[3 * ty]);         s2[3 * ty] = csn_fmav2(gd, u.c, s2[3 * ty]);
  s2[3 * ty + 1] = csn_fmav2(gb, u.b, s2[3 * ty + 1]); s2[3 * ty + 1] = csn_fmav2(gd, u.d, s2[3 * ty + 1]);
  s2[3 * ty + 2] = csn_fmav2(gb, u.c, s2[3 * ty + 2]); s2[3 * ty + 2] = csn_fmav2(gd, u.e, s2[3 * ty + 2]);
}

// four outputs -> memory; d01 / d23: the values as stored (bfloat16: rounded once, packed, and unpacked again)
__device__ __forceinline__ void dwx_store4(float* q, csn_v2 o01, csn_v2 o23, csn_v2& d01, csn_v2& d23) {
  act_st4(q, make_float4(o01[0], o01[1], o23[0], o23[1]));
  d01 = o01; d23 = o23;
}
__device__ __forceinline__ void dwx_store4(csn_bf16* q, csn_v2 o01, csn_v2 o23, csn_v2& d01, csn_v2& d23) {
  const unsigned lo = csn_pack_bf2(o01[0], o01[1]), hi = csn_pack_bf2(o23[0], o23[1]);
  *reinterpret_cast<uint2*>(q) = make_uint2(lo, hi);
  d01 = csn_mk2(csn_bits_f(lo << 16), csn_bits_f(lo & 0xffff0000u));
  d23 = csn_mk2(csn_bits_f(hi << 16), csn_bits_f(hi & 0xffff0000u));
}

// XL: the halo columns come from the neighbouring LANES (see dwx_row_x) -- launches whose rows sit inside one wave (csn_launch_dw_bwd)
template <typename AT, bool XL>
__global__ __launch_bounds__(CSN_BLOCK, sizeof(AT) == 2 ? 4 : 3) void dw3x3_bwd_x_kernel(DwArgs a) {
  CSN_DYN_SMEM(double, sm);
  int bid = blockIdx.x;
  int kb = 0;
  if (a.nbr > 1 && bid >= a.br[0].blk_end) kb = 1;
  if (a.nbr > 2 && bid >= a.br[1].blk_end) kb = 2;
  const DwBranch br = a.br[kb];
  if (kb > 0) bid -= a.br[kb - 1].blk_end;
  const int tiles = br.tiles_x * br.tiles_y;
  const int tile = bid % tiles;
  const int pc = bid / tiles;  // b*C + c
  const int c = pc % br.C;
  const int tx = tile % br.tiles_x, ty = tile / br.tiles_x;
  const int tid = threadIdx.x;
  const int lx = tid % br.LX, ly = tid / br.LX;
  const int H = br.H, W = br.W;
  const int x0 = (tx * br.LX + lx) * 4;
  const int y0 = (ty * br.NY + ly) * br.R;
  const bool active = ly < br.NY && x0 < W && y0 < H;
  csn_v2 s2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s2[t] = csn_mk2(0.f, 0.f);
  double rs[3] = {0.0, 0.0, 0.0};
  if (active) {
    const unsigned nb = (unsigned)(H * W) * (unsigned)sizeof(AT);
    const csn_buf gb = csn_make_buf_n(act_cast<AT>(br.in) + (int64_t)pc * H * W, nb);      // dy of the first consumer
    const csn_buf xb = csn_make_buf_n(act_cast<AT>(br.xin) + (int64_t)pc * H * W, nb);     // the producer's raw output
    const csn_buf zb = csn_make_buf_n(act_cast<AT>(br.zraw) + (int64_t)pc * H * W, nb);    // the unit's own raw output
    const bool has2 = br.dy2 != nullptr;
    const csn_buf eb = csn_make_buf_n(act_cast<AT>(has2 ? br.dy2 : br.in) + (int64_t)pc * H * W, nb);
    AT* __restrict__ op = act_cast<AT>(br.out) + (int64_t)pc * H * W;
    float w[9];
    csn_cfp w9 = csn_const(br.w9);
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w9[c * 9 + i];
    // the unit's own BatchNorm backward (see finish_g above) ...
    const float bsc = csn_const(br.bn_scale)[c], bsh = csn_const(br.bn_shift)[c];
    const float bal = csn_const(br.bn_alpha)[c], bmu = csn_const(br.bn_mean)[c], bis = csn_const(br.bn_invstd)[c];
    const float bm1 = csn_const(br.bn_m1m2)[2 * c], bm2 = csn_const(br.bn_m1m2)[2 * c + 1];
    const float bgi = csn_const(br.bn_gamma)[c] * bis, bga = bgi * bal, bB = bgi * (bis * bm2), bA = bgi * (bm1 - bmu * (bis * bm2));
    // ... and the producer's BatchNorm + PReLU (x = PReLU(z isc + ish)); (z - mean) invstd = z iis + imi
    const float isc = csn_const(br.in_scale)[c], ish = csn_const(br.in_shift)[c], ial = csn_const(br.in_alpha)[c];
    const float ilim = ial <= 1.f ? __builtin_inff() : -__builtin_inff();
    const float iis = csn_const(br.in_invstd)[c], imi = -(csn_const(br.in_mean)[c] * iis);
    const bool has_l = x0 > 0, has_r = x0 + 4 < W;
    const int yend = min(y0 + br.R, H);
    const AT tag = AT();
    // rows y0 - 1, y0 of dz and row y0 - 1 of x; the first trip's rows (dz y0 + 1, x y0) are issued before those are used
    DwRowRaw<AT> pg0 = dwx_issue<XL>(gb, y0 - 1, x0, W, tag), pz0 = dwx_issue<XL>(zb, y0 - 1, x0, W, tag), pe0 = pg0;
    DwRowRaw<AT> pg1 = dwx_issue<XL>(gb, y0, x0, W, tag), pz1 = dwx_issue<XL>(zb, y0, x0, W, tag), pe1 = pg1;
    if (has2) { pe0 = dwx_issue<XL>(eb, y0 - 1, x0, W, tag); pe1 = dwx_issue<XL>(eb, y0, x0, W, tag); }
    const DwRowRaw<AT> px0 = dwx_issue<XL>(xb, y0 - 1, x0, W, tag);
    DwRowRaw<AT> rg = dwx_issue<XL>(gb, y0 + 1, x0, W, tag), rz = dwx_issue<XL>(zb, y0 + 1, x0, W, tag), re = rg;
    if (has2) re = dwx_issue<XL>(eb, y0 + 1, x0, W, tag);
    DwRowRaw<AT> rx = dwx_issue<XL>(xb, y0, x0, W, tag);
#ifndef CSN_CPU_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
    csn_v2 a1_01, a1_23, a2_01, a2_23;    // open outputs: row r (top + middle taps so far), row r + 1 (top taps)
    csn_v2 gpb, gpd, gcb, gcd;            // centre pairs of dz rows r - 1 (zero when not the lane's own) and r
    {
      const DwRow2 g = dwx_dz_row(dwx_dy_row<XL>(pg0, pe0, has2), dwx_row_of<XL>(pz0), has_l, has_r, bsc, bsh, bgi, bga, bB, y0 > 0 ? bA : 0.f);
      a1_01 = csn_mul2(w[0], g.a);        a1_23 = csn_mul2(w[0], g.c);
      a1_01 = csn_fma2(w[1], g.b, a1_01); a1_23 = csn_fma2(w[1], g.d, a1_23);
      a1_01 = csn_fma2(w[2], g.c, a1_01); a1_23 = csn_fma2(w[2], g.e, a1_23);
    }
    {
      const DwRow2 g = dwx_dz_row(dwx_dy_row<XL>(pg1, pe1, has2), dwx_row_of<XL>(pz1), has_l, has_r, bsc, bsh, bgi, bga, bB, bA);
      a1_01 = csn_fma2(w[3], g.a, a1_01); a1_23 = csn_fma2(w[3], g.c, a1_23);
      a1_01 = csn_fma2(w[4], g.b, a1_01); a1_23 = csn_fma2(w[4], g.d, a1_23);
      a1_01 = csn_fma2(w[5], g.c, a1_01); a1_23 = csn_fma2(w[5], g.e, a1_23);
      a2_01 = csn_mul2(w[0], g.a);        a2_23 = csn_mul2(w[0], g.c);
      a2_01 = csn_fma2(w[1], g.b, a2_01); a2_23 = csn_fma2(w[1], g.d, a2_23);
      a2_01 = csn_fma2(w[2], g.c, a2_01); a2_23 = csn_fma2(w[2], g.e, a2_23);
      gcb = g.b; gcd = g.d;
      gpb = csn_mk2(0.f, 0.f); gpd = gpb;
    }
    {
      float pre[4];
      const DwRow2 u = dwx_x_row(dwx_row_of<XL>(px0), has_l, has_r, y0 > 0 ? isc : 0.f, y0 > 0 ? ish : 0.f, ial, ilim, pre);
      dwx_wgrad(gcb, gcd, u, s2, 0);   // x row y0 - 1 is the top tap row of output row y0
    }
    // One trip = one output row.  The rows loaded during a trip are consumed by the next one: a trip first converts the load
    // registers (its only use of them), then issues the next trip's loads INTO THE SAME registers, then does its arithmetic with
    // those loads in flight -- no second register set and no copies (16 of ~200 vector instructions per trip with copies; the loop
    // unrolled by two with two sets swapping roles needs 137 registers).
#ifndef CSN_CPU_EMU
#define DWX_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define DWX_FENCE()
#endif
#define DWX_TRIP(r, G, Z, E, X, AHEAD)                                                                                          \
  {                                                                                                                             \
    const DwRow fg = dwx_dy_row<XL>(G, E, has2), fz = dwx_row_of<XL>(Z), zr = dwx_row_of<XL>(X);                                              \
    DWX_FENCE();   /* conversions above, a later trip's loads between the fences, arithmetic below */                           \
    G = dwx_issue<XL>(gb, (r) + 1 + AHEAD, x0, W, tag); Z = dwx_issue<XL>(zb, (r) + 1 + AHEAD, x0, W, tag);                       \
    if (has2) E = dwx_issue<XL>(eb, (r) + 1 + AHEAD, x0, W, tag);                                                                \
    X = dwx_issue<XL>(xb, (r) + AHEAD, x0, W, tag);                                                                              \
    DWX_FENCE();                                                                                                                \
    const DwRow2 g = dwx_dz_row(fg, fz, has_l, has_r, bsc, bsh, bgi, bga, bB, (r) + 1 < H ? bA : 0.f);                                                     \
    /* dx of row r is complete with the bottom taps; rows r + 1, r + 2 stay open */                                             \
    csn_v2 o01 = csn_fma2(w[6], g.a, a1_01), o23 = csn_fma2(w[6], g.c, a1_23);                                              \
    o01 = csn_fma2(w[7], g.b, o01); o23 = csn_fma2(w[7], g.d, o23);                                                         \
    o01 = csn_fma2(w[8], g.c, o01); o23 = csn_fma2(w[8], g.e, o23);                                                         \
    a1_01 = csn_fma2(w[3], g.a, a2_01); a1_23 = csn_fma2(w[3], g.c, a2_23);                                                 \
    a1_01 = csn_fma2(w[4], g.b, a1_01); a1_23 = csn_fma2(w[4], g.d, a1_23);                                                 \
    a1_01 = csn_fma2(w[5], g.c, a1_01); a1_23 = csn_fma2(w[5], g.e, a1_23);                                                 \
    a2_01 = csn_mul2(w[0], g.a);        a2_23 = csn_mul2(w[0], g.c);                                                        \
    a2_01 = csn_fma2(w[1], g.b, a2_01); a2_23 = csn_fma2(w[1], g.d, a2_23);                                                 \
    a2_01 = csn_fma2(w[2], g.c, a2_01); a2_23 = csn_fma2(w[2], g.e, a2_23);                                                 \
    csn_v2 d01, d23;   /* the values as stored */                                                                               \
    dwx_store4(op + (int64_t)(r) * W + x0, o01, o23, d01, d23);                                                                 \
    const bool own_n = (r) + 1 < yend;   /* dz row r + 1 is one of the lane's own output rows */                                 \
    const csn_v2 gnb = csn_mk2(own_n ? g.b[0] : 0.f, own_n ? g.b[1] : 0.f), gnd = csn_mk2(own_n ? g.d[0] : 0.f, own_n ? g.d[1] : 0.f); \
    float pre[4];                                                                                                               \
    const DwRow2 u = dwx_x_row(zr, has_l, has_r, isc, ish, ial, ilim, pre);                                                               \
    dwx_wgrad(gnb, gnd, u, s2, 0);                                                                                              \
    dwx_wgrad(gcb, gcd, u, s2, 1);                                                                                              \
    dwx_wgrad(gpb, gpd, u, s2, 2);                                                                                              \
    /* the producer's BatchNorm-backward sums (bn_bwd_reduce_kernel): dy = the dx just stored; fp32 over the row's four values  \
       (as even / odd pairs), fp64 across the rows of the lane (see the kernel above) */                                        \
    const csn_v2 p01 = csn_mk2(pre[0], pre[1]), p23 = csn_mk2(pre[2], pre[3]);                                                  \
    const csn_v2 e01 = csn_mul2(ial, d01), e23 = csn_mul2(ial, d23);                                                        \
    const csn_v2 b01 = csn_mk2(p01[0] > 0.f ? d01[0] : e01[0], p01[1] > 0.f ? d01[1] : e01[1]);                                 \
    const csn_v2 b23 = csn_mk2(p23[0] > 0.f ? d23[0] : e23[0], p23[1] > 0.f ? d23[1] : e23[1]);                                 \
    const csn_v2 h01 = csn_fma2(iis, csn_mk2(zr.v[1], zr.v[2]), csn_mk2(imi, imi));                                       \
    const csn_v2 h23 = csn_fma2(iis, csn_mk2(zr.v[3], zr.v[4]), csn_mk2(imi, imi));                                       \
    const csn_v2 n01 = csn_mk2(fminf(p01[0], 0.f), fminf(p01[1], 0.f)), n23 = csn_mk2(fminf(p23[0], 0.f), fminf(p23[1], 0.f));   \
    const csn_v2 q0 = csn_mk2(b01[0] + b23[0], b01[1] + b23[1]);                                                                \
    const csn_v2 q1 = csn_fmav2(b23, h23, csn_fmav2(b01, h01, csn_mk2(0.f, 0.f)));                                              \
    const csn_v2 q2 = csn_fmav2(d23, n23, csn_fmav2(d01, n01, csn_mk2(0.f, 0.f)));                                              \
    rs[0] += (double)(q0[0] + q0[1]); rs[1] += (double)(q1[0] + q1[1]); rs[2] += (double)(q2[0] + q2[1]);                       \
    gpb = gcb; gpd = gcd; gcb = gnb; gcd = gnd;                                                                                 \
  }
    for (int r = y0; r < yend; ++r) DWX_TRIP(r, rg, rz, re, rx, 1)
#undef DWX_TRIP
#undef DWX_FENCE
    {   // x row yend: the bottom tap row of output row yend - 1 (whose centre pairs the roll left in gpb / gpd)
      float pre[4];
      const DwRow2 u = dwx_x_row(dwx_row_of<XL>(rx), has_l, has_r, yend < H ? isc : 0.f, yend < H ? ish : 0.f, ial, ilim, pre);
      dwx_wgrad(gpb, gpd, u, s2, 2);
    }
  }
  double sv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) sv[t] = (double)s2[t][0] + (double)s2[t][1];
  bn_block_sum_n<9>(sv, sm);
  bn_block_sum_n<3>(rs, sm);
  if (tid == 0) {
    const int b = pc / br.C;
    double* o = br.stats + ((int64_t)c * CSN_BN_NSLAB + (int64_t)b * tiles + tile) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) o[t] = sv[t];
    double* q = br.bnred + ((int64_t)c * CSN_BN_NSLAB + (int64_t)b * tiles + tile) * 3;
    q[0] = rs[0]; q[1] = rs[1]; q[2] = rs[2];
  }
}

int csn_launch_dw_bwd(const DwArgs& a, void* stream) {
  const int nblk = a.br[a.nbr - 1].blk_end;
  if (nblk <= 0) return 0;
  bool vec = true;
  for (int k = 0; k < a.nbr; ++k) vec = vec && (a.br[k].W % 4 == 0);
  const size_t sml = CSN_BLOCK * sizeof(double);
  const bool bnf = a.br[0].zraw != nullptr, xbn = a.br[0].in_scale != nullptr;   // (all branches of a launch alike)
  for (int k = 1; k < a.nbr; ++k)
    if ((a.br[k].zraw != nullptr) != bnf || (a.br[k].in_scale != nullptr) != xbn) return 1;
  if (xbn && !bnf) return 1;   // (the planner only skips an activation when its consumer runs the fully fused backward)
  if (xbn && vec && a.variant >= 1) {
    // halo columns from the neighbouring lanes: every row of lanes inside one wave, the whole plane width in one tile
    bool xl = a.variant == 2;
    for (int k = 0; k < a.nbr; ++k) xl = xl && a.br[k].tiles_x == 1 && a.br[k].LX <= 64 && (a.br[k].LX & (a.br[k].LX - 1)) == 0;
    if (a.a16 && xl) CSN_LAUNCH((dw3x3_bwd_x_kernel<csn_bf16, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    else if (a.a16) CSN_LAUNCH((dw3x3_bwd_x_kernel<csn_bf16, false>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    else if (xl) CSN_LAUNCH((dw3x3_bwd_x_kernel<float, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    else CSN_LAUNCH((dw3x3_bwd_x_kernel<float, false>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    return (int)hipGetLastError();
  }
#define DWB_LAUNCH(V, T)                                                                                         \
  do {                                                                                                          \
    if (xbn) CSN_LAUNCH((dw3x3_bwd_kernel<V, T, true, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);     \
    else if (bnf) CSN_LAUNCH((dw3x3_bwd_kernel<V, T, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);      \
    else CSN_LAUNCH((dw3x3_bwd_kernel<V, T, false>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);              \
  } while (0)
  if (a.a16) {
    if (vec) DWB_LAUNCH(true, csn_bf16);
    else DWB_LAUNCH(false, csn_bf16);
  } else if (vec) {
    DWB_LAUNCH(true, float);
  } else {
    DWB_LAUNCH(false, float);
  }
#undef DWB_LAUNCH
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- two depthwise units back to back
// conv3x3_1 -> conv3x3_2 of an ILBlock (csnet.py:74-75) in one pass over HBM: the block computes the first
// unit's output for its rows plus one halo row above and below into LDS (never written to HBM), then the
// second unit reads its 3x3 windows from LDS.  Rows of the intermediate that lie outside the image are
// ZERO (the second conv pads its input, it does not see the first conv's response to padding).
// Used when the plane is at most 256 pixels wide (one tile in x), i.e. for every CSNet resolution.
template <bool VEC>
__global__ __launch_bounds__(CSN_BLOCK) void dw3x3x2_bn_prelu_kernel(DwArgs a) {
  CSN_DYN_SMEM(float, lds);
  int bid = blockIdx.x;
  int k = 0;
  if (a.nbr > 1 && bid >= a.br[0].blk_end) k = 1;
  if (a.nbr > 2 && bid >= a.br[1].blk_end) k = 2;
  const DwBranch br = a.br[k];
  if (k > 0) bid -= a.br[k - 1].blk_end;
  const int ty = bid % br.tiles_y;
  const int pc = bid / br.tiles_y;  // b*C + c
  const int c = pc % br.C;
  const int tid = threadIdx.x;
  const int lx = tid % br.LX, ly = tid / br.LX;
  const int H = br.H, W = br.W, R = br.R, NY = br.NY;
  const int pitch = br.LX * 4 + 8;           // [4 pad | LX*4 pixels | 4 pad], rows 16-byte aligned
  const int x0 = lx * 4;
  const int yb = ty * NY * R;                // first output row of the block
  const bool active = ly < NY && x0 < W;
  const csn_buf rb = csn_make_buf_n(br.in + (int64_t)pc * H * W, (unsigned)(H * W) * 4u);
  float* __restrict__ op = br.out + (int64_t)pc * H * W;
  csn_cfp w9a = csn_const(br.w9), w9b = csn_const(br.w9b);
  const bool has_l = x0 > 0, has_r = x0 + 4 < W;
  // ---- phase 1: intermediate rows [yb - 1, yb + NY*R] -> LDS row index (y - yb + 1) ----
  if (active) {
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w9a[c * 9 + i];
    const float sc = csn_const(br.scale)[c], sh = csn_const(br.shift)[c], al = csn_const(br.alpha)[c];
    const int r_first = ly * R - (ly == 0 ? 1 : 0);
    const int r_last = ly * R + R + (ly == NY - 1 ? 1 : 0);   // exclusive
    if (x0 == 0 || !has_r) {                                   // zero the pad columns once per row
      for (int r = r_first; r < r_last; ++r) {
        float* row = lds + (r + 1) * pitch;
        if (x0 == 0) { row[0] = row[1] = row[2] = row[3] = 0.f; }
        if (!has_r) {
          const int xe = x0 + 4 + 4;  // first pad column after this lane's strip
#pragma unroll
          for (int j = 0; j < 4; ++j) row[xe + j] = 0.f;
        }
      }
    }
    DwRow r0 = dw_load_row<VEC>(rb, yb + r_first - 1, x0, W, has_l, has_r);
    DwRow r1 = dw_load_row<VEC>(rb, yb + r_first, x0, W, has_l, has_r);
    for (int r = r_first; r < r_last; r += 4) {
      const DwRow n0 = dw_load_row<VEC>(rb, yb + r + 1, x0, W, has_l, has_r);
      const DwRow n1 = dw_load_row<VEC>(rb, yb + r + 2, x0, W, has_l, has_r);
      const DwRow n2 = dw_load_row<VEC>(rb, yb + r + 3, x0, W, has_l, has_r);
      const DwRow n3 = dw_load_row<VEC>(rb, yb + r + 4, x0, W, has_l, has_r);
      const DwRow tp[6] = {r0, r1, n0, n1, n2, n3};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = r + q;
        if (rr < r_last) {
          const int yy = yb + rr;
          const bool inside = yy >= 0 && yy < H;
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float acc = w[0] * tp[q].v[j];
            acc = fmaf(w[1], tp[q].v[j + 1], acc);
            acc = fmaf(w[2], tp[q].v[j + 2], acc);
            acc = fmaf(w[3], tp[q + 1].v[j], acc);
            acc = fmaf(w[4], tp[q + 1].v[j + 1], acc);
            acc = fmaf(w[5], tp[q + 1].v[j + 2], acc);
            acc = fmaf(w[6], tp[q + 2].v[j], acc);
            acc = fmaf(w[7], tp[q + 2].v[j + 1], acc);
            acc = fmaf(w[8], tp[q + 2].v[j + 2], acc);
            const float t = csn_epi(acc, sc, sh, al);
            o[j] = (inside && x0 + j < W) ? t : 0.f;
          }
          *reinterpret_cast<float4*>(lds + (rr + 1) * pitch + 4 + x0) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
      r0 = n2;
      r1 = n3;
    }
  }
  __syncthreads();
  // ---- phase 2: second depthwise unit from LDS ----
  if (active) {
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w9b[c * 9 + i];
    const float sc = csn_const(br.scale_b)[c], sh = csn_const(br.shift_b)[c], al = csn_const(br.alpha_b)[c];
    const float* base = lds + 4 + x0;   // column x0 of LDS row 0 (= image row yb - 1)
    auto ld_row = [&](int lr) {
      DwRow r;
      const float* p = base + lr * pitch;
      const float4 cv = *reinterpret_cast<const float4*>(p);
      r.v[0] = p[-1]; r.v[1] = cv.x; r.v[2] = cv.y; r.v[3] = cv.z; r.v[4] = cv.w; r.v[5] = p[4];
      return r;
    };
    const int rs = ly * R;
    DwRow top = ld_row(rs), mid = ld_row(rs + 1);
    float* __restrict__ pp = br.pool ? br.pool + (int64_t)pc * (H >> 1) * (W >> 1) : nullptr;   // R is even then
    float* __restrict__ pm = br.pool_mp ? br.pool_mp + (int64_t)pc * (H >> 2) * (W >> 2) : nullptr;   // R % 4 == 0 then
    float mx = 0.f;             // maximum of the first average row of a group of four rows
    float e0 = 0.f, e1 = 0.f;   // left-to-right sums of the even row's two column pairs
    for (int q = 0; q < R; ++q) {
      const int y = yb + rs + q;
      if (y >= H) break;
      const DwRow bot = ld_row(rs + q + 2);
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float acc = w[0] * top.v[j];
        acc = fmaf(w[1], top.v[j + 1], acc);
        acc = fmaf(w[2], top.v[j + 2], acc);
        acc = fmaf(w[3], mid.v[j], acc);
        acc = fmaf(w[4], mid.v[j + 1], acc);
        acc = fmaf(w[5], mid.v[j + 2], acc);
        acc = fmaf(w[6], bot.v[j], acc);
        acc = fmaf(w[7], bot.v[j + 1], acc);
        acc = fmaf(w[8], bot.v[j + 2], acc);
        o[j] = csn_epi(acc, sc, sh, al);
      }
      float* q4 = op + (int64_t)y * W + x0;
      if (VEC && pp) {   // avg_pool2d(2, 2) of the output, same summation order as avgpool2_kernel
        if ((q & 1) == 0) {
          e0 = o[0] + o[1];
          e1 = o[2] + o[3];
        } else {
          float2 pv;
          pv.x = (e0 + o[0] + o[1]) * 0.25f;
          pv.y = (e1 + o[2] + o[3]) * 0.25f;
          *reinterpret_cast<float2*>(pp + (int64_t)(y >> 1) * (W >> 1) + (x0 >> 1)) = pv;
          if (pm) {   // the lane's 4 x 4 block holds one 2x2 window of the averages (same values, max is order-free)
            if ((q & 3) == 1) mx = fmaxf(pv.x, pv.y);
            else pm[(int64_t)(y >> 2) * (W >> 2) + (x0 >> 2)] = fmaxf(mx, fmaxf(pv.x, pv.y));
          }
        }
      }
      if (VEC) {
        if (!br.skip_out) *reinterpret_cast<float4*>(q4) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x0 + j < W) q4[j] = o[j];
      }
      top = mid;
      mid = bot;
    }
  }
}

// Round 5: the same pair on dw_core.h's row core.  The kernel above spends ~13 vector instructions per output pixel and unit (ISA:
// 18 packed FMAs per row of four pixels, but also a BN multiply-add, multiply + compare + select for PReLU, per-element masks and
// pair-building moves) and is bound by exactly that (profiles/r4_notes.md: 83 % VALU utilisation; tools/probes/issue_probe2: every
// instruction pays its own issue slot on this part).  Here: BN scale folded into the nine weights and the shift as the accumulator's
// initial value (the per-channel records DwBranch::rec, CSN_PREP_DWREC), PReLU as one packed multiply + one v_med3 per value, rows held
// as the five overlapping pairs a packed FMA takes, masks as 0 / 1 factors per ROW.  Same block / lane geometry and LDS layout.
// Needs W % 4 == 0 (every CSNet resolution: H, W multiples of 16).
// n / d for n * d < 2^32 with the host's m = ceil(2^32 / d) (csn_div_magic): one multiply-high instead of ~25 instructions
__device__ __forceinline__ unsigned dw_div(unsigned n, unsigned m) {   // m = 0: d = 1
#ifdef CSN_CPU_EMU
  return m ? (unsigned)(((unsigned long long)n * m) >> 32) : n;
#else
  return m ? __umulhi(n, m) : n;
#endif
}

// a row of the intermediate + the zero frame next to it
__device__ __forceinline__ void dw2f_put(float* d, bool has_l, bool has_r, csn_v2 o01, csn_v2 o23) {
  dw_st4_lds(d, o01, o23);
  if (!has_l) d[-1] = 0.f;
  if (!has_r) d[4] = 0.f;
}
__device__ __forceinline__ void dw2f_row(const DwPar& pa, float* d, int yy, int H, bool has_l, bool has_r, const DwRow2& t, const DwRow2& m,
                                         const DwRow2& b) {
  if (yy >= 0 && yy < H) {
    csn_v2 o01, o23;
    dw_conv4(pa, t, m, b, o01, o23);
    dw2f_put(d, has_l, has_r, dw_prelu2(o01, pa.al, pa.lim), dw_prelu2(o23, pa.al, pa.lim));
  } else {
    dw2f_put(d, has_l, has_r, csn_mk2(0.f, 0.f), csn_mk2(0.f, 0.f));
  }
}

// XL: LX is a power of two >= the plane's strips (idle lanes at the end of every row of lanes), so that a row of lanes sits inside one
// wave and the two halo columns of a loaded row are the neighbouring lanes' registers (csn_from_lane_below / _above: one DPP move each
// instead of a load -- the knock-out build without the halo loads measured -12 % on this kernel, it is bound by the number of
// vector-memory instructions like the train-mode kernels).
template <bool XL>
__global__ __launch_bounds__(CSN_BLOCK) void dw3x3x2_fast_kernel(DwArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS DwArgs* a = CSN_KERNARG(DwArgs, a_byval);
  int bid = blockIdx.x;
  int k = 0;
  if (a->nbr > 1 && bid >= a->br[0].blk_end) k = 1;
  if (a->nbr > 2 && bid >= a->br[1].blk_end) k = 2;
  const CSN_CONST_AS DwBranch* br = &a->br[k];
  if (k > 0) bid -= a->br[k - 1].blk_end;
  // block -> (plane, row tile), thread -> (lane row, strip): the divisors are launch constants, the host passes their reciprocals
  const int pc = (int)dw_div((unsigned)bid, br->m_tiles_y);   // b*C + c
  const int ty = bid - pc * br->tiles_y;
  const int c = pc - (int)dw_div((unsigned)pc, br->m_C) * br->C;
  const int tid = threadIdx.x;
  const int ly = (int)dw_div((unsigned)tid, br->m_LX), lx = tid - ly * br->LX;
  const int H = br->H, W = br->W, R = br->R, NY = br->NY;   // R % 4 == 0 (launcher)
  const int pitch = W + 8;                   // [4 pad | W pixels | 4 pad], rows 16-byte aligned (W % 4 == 0)
  const int x0 = lx * 4;
  const int yb = ty * NY * R;                // first output row of the block
  const bool active = ly < NY && x0 < W;
  const csn_buf rb = csn_make_buf_n(br->in + (int64_t)pc * H * W, (unsigned)(H * W) * 4u);
  csn_cfp rec = csn_const(br->rec) + c * (2 * DWREC_FLOATS);
  const bool has_l = x0 > 0, has_r = x0 + 4 < W;
  const float ml = has_l ? 1.f : 0.f, mr = has_r ? 1.f : 0.f;
  const unsigned W4 = (unsigned)W * 4u;
  // a row from HBM as loaded (one 128-bit load + the two halo columns through ONE byte offset, of column x0 - 1, and immediates);
  // rows above / below the plane fall out of the bounded resource (zeros)
  struct Raw { float4 c; float l, r; };
  auto issue = [&](unsigned ro) {
    Raw q;
#ifdef CSN_KO_DW2_NOLOAD   // knock-out build (tools/README.md): the pair without its HBM reads (= a fused block's traffic; wrong results)
    q.c = make_float4(1.f, 2.f, 3.f, csn_bits_f(ro)); q.l = q.r = 0.f; return q;
#endif
    q.c = csn_ld4(rb, ro + 4u, 0);
#ifndef CSN_EMU_SEQ
    if (XL) { q.l = q.r = 0.f; return q; }
#endif
    q.l = csn_ld1(rb, ro, 0);
    q.r = csn_ld1(rb, ro + 20u, 0);
    return q;
  };
  auto fin = [&](const Raw& q) {
#ifndef CSN_EMU_SEQ
    if (XL) {
      const float l = csn_bits_f(csn_from_lane_below(csn_f_bits(q.c.w))), r = csn_bits_f(csn_from_lane_above(csn_f_bits(q.c.x)));
      return dw_row2_regs(l * ml, q.c.x, q.c.y, q.c.z, q.c.w, r * mr);
    }
#endif
    return dw_row2_regs(q.l * ml, q.c.x, q.c.y, q.c.z, q.c.w, q.r * mr);
  };
  // ---- phase 1: intermediate rows [yb - 1, yb + NY*R] -> LDS row index (y - yb + 1); rows outside the image are ZERO (the second
  // conv pads its input, it does not see the first conv's response to padding).  Lane row ly owns rows [ly R, ly R + R) in trips of
  // four; the block's two halo rows (-1 and NY R) are one extra row for the first / last lane row -- round 4's loop gave those
  // lanes R + 1 rows, i.e. a whole extra trip of four for every wave that holds one of them ----
  if (active) {
    const DwPar pa = dw_par_load(rec);
    const int y0 = yb + ly * R;                                  // first own image row
    float* lrow = lds + (ly * R + 1) * pitch + 4 + x0;           // its LDS position
    unsigned ro = (unsigned)((y0 - 1) * W + x0) * 4u - 4u;       // image row y0 - 1, column x0 - 1
    const int Rm = y0 > H ? 0 : R;                               // (lane rows past the image's zero row H: nothing of theirs is read)
    // the halo row of the first / last lane row: its own three-row window (two of the rows are loaded again below: L1 hits)
    const bool first = ly == 0, last = ly == NY - 1;
    if (first || last) {
      const int yh = first ? yb - 1 : yb + NY * R;
      const unsigned rh = (unsigned)((yh - 1) * W + x0) * 4u - 4u;
      const Raw h0 = issue(rh), h1 = issue(rh + W4), h2 = issue(rh + 2u * W4);
      dw2f_row(pa, lds + (yh - yb + 1) * pitch + 4 + x0, yh, H, has_l, has_r, fin(h0), fin(h1), fin(h2));
      if (first && last) {   // (one lane row per block: both halo rows)
        const unsigned rg = (unsigned)((yb + R - 1) * W + x0) * 4u - 4u;
        const Raw g0 = issue(rg), g1 = issue(rg + W4), g2 = issue(rg + 2u * W4);
        dw2f_row(pa, lds + (R + 1) * pitch + 4 + x0, yb + R, H, has_l, has_r, fin(g0), fin(g1), fin(g2));
      }
    }
    const Raw a0 = issue(ro), a1 = issue(ro + W4);
    ro += 2u * W4;
    DwRow2 r0 = fin(a0), r1 = fin(a1);
    for (int r = 0; r < Rm; r += 4) {
      const Raw p0 = issue(ro), p1 = issue(ro + W4), p2 = issue(ro + 2u * W4), p3 = issue(ro + 3u * W4);
      ro += 4u * W4;
      const DwRow2 tp[6] = {r0, r1, fin(p0), fin(p1), fin(p2), fin(p3)};
      if (y0 + r + 4 <= H) {   // (y0 >= 0 always: the trip is inside the image -- no per-row tests)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          csn_v2 o01, o23;
          dw_conv4(pa, tp[q], tp[q + 1], tp[q + 2], o01, o23);
          dw2f_put(lrow + q * pitch, has_l, has_r, dw_prelu2(o01, pa.al, pa.lim), dw_prelu2(o23, pa.al, pa.lim));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) dw2f_row(pa, lrow + q * pitch, y0 + r + q, H, has_l, has_r, tp[q], tp[q + 1], tp[q + 2]);
      }
      lrow += 4 * pitch;
      r0 = tp[4];
      r1 = tp[5];
    }
  }
  __syncthreads();
  // ---- phase 2: second depthwise unit from LDS, four rows per trip (no register rotation inside a trip) ----
  if (active) {
    const DwPar pb = dw_par_load(rec + DWREC_FLOATS);
    const int rs = ly * R;
    const float* lp = lds + 3 + x0 + rs * pitch;   // column x0 - 1 of the LDS row of image row yb + rs - 1
    DwRow2 r0 = dw_row2_lds4(lp), r1 = dw_row2_lds4(lp + pitch);
    lp += 2 * pitch;
    float* __restrict__ pp = br->pool ? br->pool + (int64_t)pc * (H >> 1) * (W >> 1) + (x0 >> 1) : nullptr;
    float* __restrict__ pm = br->pool_mp ? br->pool_mp + (int64_t)pc * (H >> 2) * (W >> 2) + (x0 >> 2) : nullptr;
    const int y0 = yb + rs;
    const int nrow = min(R, H - y0);   // rows of this lane inside the image (<= 0: none)
    float* __restrict__ q4 = br->out + (int64_t)pc * H * W + (int64_t)y0 * W + x0;
    const bool skip = br->skip_out != 0;
    for (int q = 0; q < nrow; q += 4) {
      DwRow2 tp[6];
      tp[0] = r0; tp[1] = r1;
#pragma unroll
      for (int i = 0; i < 4; ++i) tp[2 + i] = dw_row2_lds4(lp + i * pitch);   // (rows past the tile's last: unused)
      lp += 4 * pitch;
      csn_v2 o[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dw_conv4(pb, tp[i], tp[i + 1], tp[i + 2], o[i][0], o[i][1]);
        o[i][0] = dw_prelu2(o[i][0], pb.al, pb.lim);
        o[i][1] = dw_prelu2(o[i][1], pb.al, pb.lim);
      }
      if (!skip) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (q + i < nrow) *reinterpret_cast<float4*>(q4 + (q + i) * W) = make_float4(o[i][0][0], o[i][0][1], o[i][1][0], o[i][1][1]);
      }
      if (pp) {   // avg_pool2d(2, 2) of the output, same summation order as avgpool2_kernel; R even: rows q, q + 1 are a pair
        float2 pv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          pv[h].x = ((o[2 * h][0][0] + o[2 * h][0][1]) + o[2 * h + 1][0][0] + o[2 * h + 1][0][1]) * 0.25f;
          pv[h].y = ((o[2 * h][1][0] + o[2 * h][1][1]) + o[2 * h + 1][1][0] + o[2 * h + 1][1][1]) * 0.25f;
          if (q + 2 * h + 1 < nrow) *reinterpret_cast<float2*>(pp + (int64_t)((y0 + q) / 2 + h) * (W >> 1)) = pv[h];
        }
        // the lane's 4 x 4 block holds one 2x2 window of the averages (same values, max is order-free); H % 4 == 0
        if (pm && q + 3 < nrow) pm[(int64_t)((y0 + q) >> 2) * (W >> 2)] = fmaxf(fmaxf(pv[0].x, pv[0].y), fmaxf(pv[1].x, pv[1].y));
      }
      r0 = tp[4];
      r1 = tp[5];
    }
  }
}

static unsigned csn_div_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }   // dw_div

size_t csn_dw2_lds_bytes(const DwArgs& a) {
  size_t m = 0;
  for (int k = 0; k < a.nbr; ++k) {
    // (+ 3 rows: dw3x3x2_fast_kernel's second phase reads its window four rows at a time, past the last row it uses)
    // (row pitch: LX strips of four in dw3x3x2_bn_prelu_kernel, the plane's width in the fast kernel -- LX may be rounded up there)
    const size_t n = (size_t)(a.br[k].NY * a.br[k].R + 2 + 3) * (std::max(a.br[k].LX * 4, a.br[k].W) + 8) * sizeof(float);
    if (n > m) m = n;
  }
  return m;
}

int csn_launch_dw2(const DwArgs& a, void* stream) {
  const int nblk = a.br[a.nbr - 1].blk_end;
  if (nblk <= 0) return 0;
  bool vec = true;
  for (int k = 0; k < a.nbr; ++k) vec = vec && (a.br[k].W % 4 == 0);
  const size_t lds = csn_dw2_lds_bytes(a);
  bool fast = vec;   // every branch carries the folded per-channel records (dw_core.h)
  for (int k = 0; k < a.nbr; ++k) fast = fast && a.br[k].rec != nullptr;
  for (int k = 0; k < a.nbr; ++k) fast = fast && (a.br[k].R % 4) == 0 && a.br[k].tiles_x == 1;
  if (fast) {
    DwArgs f = a;
    for (int k = 0; k < f.nbr; ++k) {
      f.br[k].m_tiles_y = csn_div_magic((unsigned)f.br[k].tiles_y);
      f.br[k].m_C = csn_div_magic((unsigned)f.br[k].C);
      f.br[k].m_LX = csn_div_magic((unsigned)f.br[k].LX);
    }
    bool xl = true;   // rows of lanes inside one wave
    for (int k = 0; k < f.nbr; ++k) xl = xl && f.br[k].LX <= 64 && (f.br[k].LX & (f.br[k].LX - 1)) == 0;
    if (xl) CSN_LAUNCH(dw3x3x2_fast_kernel<true>, dim3(nblk), dim3(CSN_BLOCK), lds, stream, f);
    else CSN_LAUNCH(dw3x3x2_fast_kernel<false>, dim3(nblk), dim3(CSN_BLOCK), lds, stream, f);
  } else if (vec) {
    CSN_LAUNCH((dw3x3x2_bn_prelu_kernel<true>), dim3(nblk), dim3(CSN_BLOCK), lds, stream, a);
  } else {
    CSN_LAUNCH((dw3x3x2_bn_prelu_kernel<false>), dim3(nblk), dim3(CSN_BLOCK), lds, stream, a);
  }
  return (int)hipGetLastError();
}

int csn_launch_dw(const DwArgs& a, void* stream) {
  const int nblk = a.br[a.nbr - 1].blk_end;
  if (nblk <= 0) return 0;
  bool vec = true;  // float4 path needs every branch width to be a multiple of 4
  for (int k = 0; k < a.nbr; ++k) vec = vec && (a.br[k].W % 4 == 0);
  bool stats = true;   // every branch of the launch carries a statistics table (train-mode forward), or none does
  for (int k = 0; k < a.nbr; ++k) stats = stats && a.br[k].stats != nullptr;
  const size_t sml = stats ? CSN_BLOCK * sizeof(double) : 0;
  bool inbn = true;    // ... and the same for the on-load input transform (needs the statistics partials' indexing)
  for (int k = 0; k < a.nbr; ++k) inbn = inbn && a.br[k].in_scale != nullptr && a.br[k].gapin != nullptr;
  for (int k = 0; k < a.nbr; ++k)
    if (!inbn && a.br[k].in_scale != nullptr) return 1;
  if (inbn && !stats) return 1;
  // train-mode launches in the power-of-two geometry (csn_plan::dw_xl -> DwArgs::variant 2): halo columns from the neighbouring lanes
  bool xl = a.variant == 2 && vec && stats;
  for (int k = 0; k < a.nbr; ++k) xl = xl && a.br[k].tiles_x == 1 && a.br[k].LX <= 64 && (a.br[k].LX & (a.br[k].LX - 1)) == 0;
  if (xl) {
    if (a.a16 && inbn) CSN_LAUNCH((dw3x3_bn_prelu_kernel<true, csn_bf16, true, true, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    else if (a.a16) CSN_LAUNCH((dw3x3_bn_prelu_kernel<true, csn_bf16, true, false, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    else if (inbn) CSN_LAUNCH((dw3x3_bn_prelu_kernel<true, float, true, true, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    else CSN_LAUNCH((dw3x3_bn_prelu_kernel<true, float, true, false, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);
    return (int)hipGetLastError();
  }
#define DW_LAUNCH(V, T)                                                                                              \
  do {                                                                                                               \
    if (inbn) CSN_LAUNCH((dw3x3_bn_prelu_kernel<V, T, true, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);    \
    else if (stats) CSN_LAUNCH((dw3x3_bn_prelu_kernel<V, T, true>), dim3(nblk), dim3(CSN_BLOCK), sml, stream, a);    \
    else CSN_LAUNCH((dw3x3_bn_prelu_kernel<V, T, false>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);                \
  } while (0)
  if (a.a16) {
    if (vec) DW_LAUNCH(true, csn_bf16);
    else DW_LAUNCH(false, csn_bf16);
  } else if (vec) {
    DW_LAUNCH(true, float);
  } else {
    DW_LAUNCH(false, float);
  }
#undef DW_LAUNCH
  return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------- avg-pool
// out[y][x] = mean (MAX: maximum, F.max_pool2d(x, 2, 2) of csnet.py:708-714) of the 2x2 input window.  A lane produces 2
// output pixels from two float4 loads.
template <typename AT, bool MAX>
__global__ __launch_bounds__(CSN_BLOCK) void pool2_kernel(PoolArgs a) {
  int bid = blockIdx.x;
  int k = 0;
  if (a.n > 1 && bid >= a.blk_end[0]) k = 1;
  if (a.n > 2 && bid >= a.blk_end[1]) k = 2;
  if (k > 0) bid -= a.blk_end[k - 1];
  const int Ho = a.Ho[k], Wo = a.Wo[k];
  const int Wp = (Wo + 1) >> 1;  // lane columns (2 outputs each)
  const int64_t per_plane = (int64_t)Ho * Wp;
  const int64_t total = per_plane * a.planes[k];
  const int64_t idx = (int64_t)bid * CSN_BLOCK + threadIdx.x;
  if (idx >= total) return;
  const int plane = (int)(idx / per_plane);
  const int rem = (int)(idx - (int64_t)plane * per_plane);
  const int y = rem / Wp, xp = rem - y * Wp;
  const int Wi = Wo * 2;
  const AT* __restrict__ ip = act_cast<AT>(a.in[k]) + ((int64_t)plane * Ho * 2 + 2 * y) * Wi + 4 * xp;
  AT* __restrict__ op = act_cast<AT>(a.out[k]) + ((int64_t)plane * Ho + y) * Wo + 2 * xp;
  if ((Wo & 1) == 0) {
    const float4 r0 = act_ld4(ip);
    const float4 r1 = act_ld4(ip + Wi);
    float2 o;
    if (MAX) {
      o.x = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
      o.y = fmaxf(fmaxf(r0.z, r0.w), fmaxf(r1.z, r1.w));
    } else {
      o.x = (r0.x + r0.y + r1.x + r1.y) * 0.25f;
      o.y = (r0.z + r0.w + r1.z + r1.w) * 0.25f;
    }
    act_st2(op, o);
  } else {
    const float a0 = act_ld(ip), a1 = act_ld(ip + 1), a2 = act_ld(ip + Wi), a3 = act_ld(ip + Wi + 1);
    act_st(op, MAX ? fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) : (a0 + a1 + a2 + a3) * 0.25f);
    if (2 * xp + 1 < Wo) {
      const float b0 = act_ld(ip + 2), b1 = act_ld(ip + 3), b2 = act_ld(ip + Wi + 2), b3 = act_ld(ip + Wi + 3);
      act_st(op + 1, MAX ? fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)) : (b0 + b1 + b2 + b3) * 0.25f);
    }
  }
}

int csn_launch_pool(const PoolArgs& a, void* stream) {
  const int nblk = a.blk_end[a.n - 1];
  if (nblk <= 0) return 0;
  if (a.a16) CSN_LAUNCH((pool2_kernel<csn_bf16, false>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  else CSN_LAUNCH((pool2_kernel<float, false>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

int csn_launch_maxpool(const PoolArgs& a, void* stream) {   // max-pooled copies for the 3x3 passes of k_c3q.hip
  const int nblk = a.blk_end[a.n - 1];
  if (nblk <= 0) return 0;
  if (a.a16) CSN_LAUNCH((pool2_kernel<csn_bf16, true>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  else CSN_LAUNCH((pool2_kernel<float, true>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------ final bilinear x2
template <typename TI>
__global__ __launch_bounds__(CSN_BLOCK) void bilinear_up2_kernel(Up2Args a) {
  const int H = a.H, W = a.W, Hi = H >> 1, Wi = W >> 1;
  const int64_t idx = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x;
  const int64_t total = (int64_t)a.planes * H * W;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const int y = (int)((idx / W) % H);
  const int plane = (int)(idx / ((int64_t)W * H));
  int y0, y1, x0, x1;
  float ly, lx;
  csn_bilin(y, 0.5f, Hi, y0, y1, ly);
  csn_bilin(x, 0.5f, Wi, x0, x1, lx);
  const TI* __restrict__ p = act_cast<TI>(a.in) + (int64_t)plane * Hi * Wi;
  float t00 = act_ld(p + y0 * Wi + x0), t01 = act_ld(p + y0 * Wi + x1), t10 = act_ld(p + y1 * Wi + x0), t11 = act_ld(p + y1 * Wi + x1);
  if (a.nparts > 1 || a.bias) {   // the taps are sums of partial planes (+ the cls_layer bias), added up in plane order
    for (int k = 1; k < a.nparts; ++k) {
      const TI* __restrict__ pk = p + (int64_t)k * a.part_stride;
      t00 += act_ld(pk + y0 * Wi + x0); t01 += act_ld(pk + y0 * Wi + x1); t10 += act_ld(pk + y1 * Wi + x0); t11 += act_ld(pk + y1 * Wi + x1);
    }
    const float bv = a.bias ? a.bias[0] : 0.f;
    t00 += bv; t01 += bv; t10 += bv; t11 += bv;
  }
  const float v0 = (1.f - lx) * t00 + lx * t01;
  const float v1 = (1.f - lx) * t10 + lx * t11;
  a.out[idx] = (1.f - ly) * v0 + ly * v1;
}

int csn_launch_up2(const Up2Args& a, void* stream) {
  const int64_t total = (int64_t)a.planes * a.H * a.W;
  const int nblk = (int)((total + CSN_BLOCK - 1) / CSN_BLOCK);
  if (nblk <= 0) return 0;
  if (a.in16) CSN_LAUNCH((bilinear_up2_kernel<csn_bf16>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);   // out stays float (the caller's y)
  else CSN_LAUNCH((bilinear_up2_kernel<float>), dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------- caller-side pre/post
// test.py:92-96 from the logits onward (no resize): sigmoid -> x255 -> uint8 (numpy astype truncation).
__global__ __launch_bounds__(CSN_BLOCK) void saliency_u8_kernel(const float* __restrict__ y, unsigned char* __restrict__ o,
                                                                 int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const float v = y[i];
    const float e = expf(-fabsf(v));
    const float p = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    o[i] = (unsigned char)(p * 255.f);
  }
}

// test.py:68-69,86: (img - mean) / std per channel, H x W x 3 (float, [0,1]) -> 3 x H x W, batched
__global__ __launch_bounds__(CSN_BLOCK) void normalize_nchw_kernel(const float* __restrict__ hwc, float* __restrict__ chw,
                                                                    int64_t B, int64_t HW) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < B * HW; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t b = i / HW, p = i - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) chw[(b * 3 + c) * HW + p] = (hwc[i * 3 + c] - mean[c]) / stdv[c];
  }
}

// ---------------------------------------------------------------------------------------------- resize (f-2)
// The resizes either side of the forward in the inference caller (test.py:76-85,94-96): skimage.transform.resize(order 1,
// mode='reflect', anti_aliasing=False) = scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True) (skimage maps its
// 'reflect' to ndimage's 'mirror'): bilinear interpolation with half-pixel centres, src = (dst + 0.5) * in / out - 0.5, and
// a source coordinate outside [0, n - 1] is MIRRORED about the centre of the edge pixel -- src = -0.25 samples
// 0.75 a + 0.25 b, not a (clamping, which is what F.interpolate does, differs in the border rows / columns of
// every upsampled picture).  Pinned against scipy.ndimage.zoom in tests/resize_cases.py.
__device__ __forceinline__ int csn_mirror(int i, int n) {
  if (n <= 1) return 0;
  if (i < 0) i = -i;
  if (i > n - 1) i = 2 * (n - 1) - i;
  return min(max(i, 0), n - 1);
}
__device__ __forceinline__ void csn_resize_coord(int dst, float scale, int n, int& i0, int& i1, float& l1) {
  const float src = (static_cast<float>(dst) + 0.5f) * scale - 0.5f;
  const float f = floorf(src);
  l1 = src - f;
  i0 = csn_mirror(static_cast<int>(f), n);
  i1 = csn_mirror(static_cast<int>(f) + 1, n);
}

// pre: B images H_i x W_i x 3 (float, [0,1]) -> bilinear resize to H x W -> (v - mean) / std -> [B][3][H][W]
__global__ __launch_bounds__(CSN_BLOCK) void resize_normalize_kernel(const float* __restrict__ hwc, float* __restrict__ chw,
                                                                      int B, int Hi, int Wi, int H, int W) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const float sy = (float)Hi / (float)H, sx = (float)Wi / (float)W;
  const int64_t n = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
    int y0, y1, x0, x1;
    float ly, lx;
    csn_resize_coord(y, sy, Hi, y0, y1, ly);
    csn_resize_coord(x, sx, Wi, x0, x1, lx);
    const float* im = hwc + (int64_t)b * Hi * Wi * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v00 = im[((int64_t)y0 * Wi + x0) * 3 + c], v01 = im[((int64_t)y0 * Wi + x1) * 3 + c];
      const float v10 = im[((int64_t)y1 * Wi + x0) * 3 + c], v11 = im[((int64_t)y1 * Wi + x1) * 3 + c];
      const float top = v00 + lx * (v01 - v00), bot = v10 + lx * (v11 - v10);
      chw[(((int64_t)b * 3 + c) * H + y) * W + x] = ((top + ly * (bot - top)) - mean[c]) / stdv[c];
    }
  }
}

// post: logits [H][W] -> sigmoid -> bilinear resize to h x w -> (p * 255) truncated to uint8 (test.py:92-96)
__global__ __launch_bounds__(CSN_BLOCK) void saliency_resize_u8_kernel(const float* __restrict__ logits,
                                                                        unsigned char* __restrict__ o, int H, int W, int h,
                                                                        int w) {
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  const int64_t n = (int64_t)h * w;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const int x = (int)(i % w), y = (int)(i / w);
    int y0, y1, x0, x1;
    float ly, lx;
    csn_resize_coord(y, sy, H, y0, y1, ly);
    csn_resize_coord(x, sx, W, x0, x1, lx);
    auto sig = [](float v) {
      const float e = expf(-fabsf(v));
      return v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    };
    const float v00 = sig(logits[(int64_t)y0 * W + x0]), v01 = sig(logits[(int64_t)y0 * W + x1]);
    const float v10 = sig(logits[(int64_t)y1 * W + x0]), v11 = sig(logits[(int64_t)y1 * W + x1]);
    const float top = v00 + lx * (v01 - v00), bot = v10 + lx * (v11 - v10);
    o[i] = (unsigned char)((top + ly * (bot - top)) * 255.f);
  }
}

// planar float resize [planes][Hi][Wi] -> [planes][Ho][Wo]
__global__ __launch_bounds__(CSN_BLOCK) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                     int planes, int Hi, int Wi, int Ho, int Wo) {
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  const int64_t n = (int64_t)planes * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
    const int64_t pl = i / ((int64_t)Wo * Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    csn_resize_coord(y, sy, Hi, y0, y1, ly);
    csn_resize_coord(x, sx, Wi, x0, x1, lx);
    const float* im = in + pl * Hi * Wi;
    const float v00 = im[(int64_t)y0 * Wi + x0], v01 = im[(int64_t)y0 * Wi + x1];
    const float v10 = im[(int64_t)y1 * Wi + x0], v11 = im[(int64_t)y1 * Wi + x1];
    const float top = v00 + lx * (v01 - v00), bot = v10 + lx * (v11 - v10);
    out[i] = top + ly * (bot - top);
  }
}

// Saliency metrics (SalMetric/src/sal_metric.cpp:87-120): the reference makes 256 passes over every image (one per
// threshold).  All of them follow from ONE joint histogram h[v][g], v = predicted value 0..255, g = (gt > 128):
//     a_sum(th) = sum_{v > th} (h[v][0] + h[v][1]),  ab(th) = sum_{v > th} h[v][1],  b_sum = sum_v h[v][1]
// plus sum |sal - gt| for the MAE.  grid (blocks per image, images); LDS histogram, integer atomics (deterministic).
__global__ __launch_bounds__(CSN_BLOCK) void sal_hist_kernel(const unsigned char* __restrict__ sal,
                                                              const unsigned char* __restrict__ gt, int64_t npix,
                                                              unsigned long long* __restrict__ hist,
                                                              unsigned long long* __restrict__ abs_sum) {
  CSN_DYN_SMEM(unsigned int, lh);   // [512] + [1]
  const int img = blockIdx.y;
  for (int i = threadIdx.x; i < 513; i += CSN_BLOCK) lh[i] = 0u;
  __syncthreads();
  const unsigned char* sp = sal + (int64_t)img * npix;
  const unsigned char* gp = gt + (int64_t)img * npix;
  unsigned int ad = 0u;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < npix; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const int v = sp[i], g = gp[i];
    atomicAdd(&lh[2 * v + (g > 128 ? 1 : 0)], 1u);
    ad += (unsigned int)(v > g ? v - g : g - v);
  }
  atomicAdd(&lh[512], ad);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += CSN_BLOCK)
    if (lh[i]) atomicAdd(&hist[(int64_t)img * 512 + i], (unsigned long long)lh[i]);
  if (threadIdx.x == 0) atomicAdd(&abs_sum[img], (unsigned long long)lh[512]);
}

// A plain streaming copy (128-bit loads / stores, 16 elements in flight per lane): bench.py's on-box bandwidth figure next to the
// 8 TB/s spec (`roofline.peak_measured`) -- what a pure HBM-bound kernel of this library's kind reaches on the board.
// (measured, round 6, GB/s of read + written bytes for 1 GiB: ONE 128-bit access per lane on an uncapped grid 6,259; grid-stride
// forms with 2 / 4 / 8 accesses in flight per lane on 1K ... 64K blocks 5,110 ... 5,205, non-temporal accesses 5,150 ... 5,180;
// torch's copy_ 4,760)
#ifndef CSN_COPY_U
#define CSN_COPY_U 1        // 128-bit accesses in flight per lane
#endif
#ifndef CSN_COPY_BLOCKS
#define CSN_COPY_BLOCKS (1 << 22)
#endif
__global__ __launch_bounds__(CSN_BLOCK) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * CSN_BLOCK;
  int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x;
  for (; i + (CSN_COPY_U - 1) * stride < n4; i += CSN_COPY_U * stride) {
    float4 v[CSN_COPY_U];
#pragma unroll
    for (int k = 0; k < CSN_COPY_U; ++k) {
#ifdef CSN_COPY_NT
      v[k].x = __builtin_nontemporal_load(&src[i + k * stride].x); v[k].y = __builtin_nontemporal_load(&src[i + k * stride].y);
      v[k].z = __builtin_nontemporal_load(&src[i + k * stride].z); v[k].w = __builtin_nontemporal_load(&src[i + k * stride].w);
#else
      v[k] = src[i + k * stride];
#endif
    }
#pragma unroll
    for (int k = 0; k < CSN_COPY_U; ++k) {
#ifdef CSN_COPY_NT
      __builtin_nontemporal_store(v[k].x, &dst[i + k * stride].x); __builtin_nontemporal_store(v[k].y, &dst[i + k * stride].y);
      __builtin_nontemporal_store(v[k].z, &dst[i + k * stride].z); __builtin_nontemporal_store(v[k].w, &dst[i + k * stride].w);
#else
      dst[i + k * stride] = v[k];
#endif
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

int csn_launch_stream_copy(const float* src, float* dst, int64_t n, void* stream) {
  const int64_t n4 = n >> 2;
  if (n4 <= 0) return 0;
  int64_t nb = (n4 + CSN_BLOCK * CSN_COPY_U - 1) / (CSN_BLOCK * CSN_COPY_U);
  if (nb > CSN_COPY_BLOCKS) nb = CSN_COPY_BLOCKS;
  CSN_LAUNCH(stream_copy_kernel, dim3((unsigned)nb), dim3(CSN_BLOCK), 0, stream, reinterpret_cast<const float4*>(src),
             reinterpret_cast<float4*>(dst), n4);
  return (int)hipGetLastError();
}

int csn_launch_sal_hist(const unsigned char* sal, const unsigned char* gt, int64_t npix, int n_images,
                        unsigned long long* hist, unsigned long long* abs_sum, void* stream) {
  int64_t nb = (npix + CSN_BLOCK * 16 - 1) / (CSN_BLOCK * 16);
  if (nb < 1) nb = 1;
  if (nb > 64) nb = 64;
  CSN_LAUNCH(sal_hist_kernel, dim3((unsigned)nb, (unsigned)n_images), dim3(CSN_BLOCK), 513 * sizeof(unsigned int), stream,
             sal, gt, npix, hist, abs_sum);
  return (int)hipGetLastError();
}

// val() of the training caller (CSNet_training/train.py:262-276), one picture: sigmoid -> bilinear resize to the
// picture's own size -> (x * 255).int().float() / 255 -> L1 mean against the target.
__device__ __forceinline__ float csn_sigmoid(float v) {
  const float e = expf(-fabsf(v));
  return v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

__global__ __launch_bounds__(CSN_BLOCK) void val_mae_kernel(const float* __restrict__ logits, int Hi, int Wi,
                                                             const float* __restrict__ target, int H, int W, float ry,
                                                             float rx, double* mae) {
  CSN_DYN_SMEM(double, sm);
  const int64_t n = (int64_t)H * W;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    float v;
    if (H == Hi && W == Wi) {     // F.interpolate copies when the size does not change
      v = csn_sigmoid(logits[i]);
    } else {
      int y0, y1, x0, x1;
      float ly, lx;
      csn_bilin(y, ry, Hi, y0, y1, ly);
      csn_bilin(x, rx, Wi, x0, x1, lx);
      const float v0 = (1.f - lx) * csn_sigmoid(logits[y0 * Wi + x0]) + lx * csn_sigmoid(logits[y0 * Wi + x1]);
      const float v1 = (1.f - lx) * csn_sigmoid(logits[y1 * Wi + x0]) + lx * csn_sigmoid(logits[y1 * Wi + x1]);
      v = (1.f - ly) * v0 + ly * v1;
    }
    const float q = (float)(int)(v * 255.0f) / 255.0f;
    s += (double)fabsf(q - target[i]);
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = CSN_BLOCK / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
    __syncthreads();
  }
  // ONE block per picture (launcher): fixed summation order inside the block, and the running sum over pictures is a plain
  // read-modify-write ordered by the stream -- no floating-point atomics, the validation MAE is reproducible bit for bit
  if (threadIdx.x == 0) *mae += sm[0] / (double)n;
}

int csn_launch_val_mae(const float* logits, int hi, int wi, const float* target, int h, int w, double* mae, void* stream) {
  CSN_LAUNCH(val_mae_kernel, dim3(1), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, logits, hi, wi, target, h, w,
             (float)hi / (float)h, (float)wi / (float)w, mae);
  return (int)hipGetLastError();
}

int csn_launch_saliency_u8(const float* y, unsigned char* o, int64_t n, void* stream) {
  const int64_t nb = (n + CSN_BLOCK - 1) / CSN_BLOCK;
  CSN_LAUNCH(saliency_u8_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(CSN_BLOCK), 0, stream, y, o, n);
  return (int)hipGetLastError();
}

int csn_launch_normalize_nchw(const float* hwc, float* chw, int64_t B, int64_t HW, void* stream) {
  const int64_t nb = (B * HW + CSN_BLOCK - 1) / CSN_BLOCK;
  CSN_LAUNCH(normalize_nchw_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(CSN_BLOCK), 0, stream, hwc, chw, B, HW);
  return (int)hipGetLastError();
}

static inline unsigned csn_grid1(int64_t n) {
  const int64_t nb = (n + CSN_BLOCK - 1) / CSN_BLOCK;
  return (unsigned)(nb < 4096 ? (nb > 0 ? nb : 1) : 4096);
}
int csn_launch_resize_normalize(const float* hwc, float* chw, int B, int Hi, int Wi, int H, int W, void* stream) {
  CSN_LAUNCH(resize_normalize_kernel, dim3(csn_grid1((int64_t)B * H * W)), dim3(CSN_BLOCK), 0, stream, hwc, chw, B, Hi, Wi, H, W);
  return (int)hipGetLastError();
}
int csn_launch_saliency_resize_u8(const float* logits, unsigned char* o, int H, int W, int h, int w, void* stream) {
  CSN_LAUNCH(saliency_resize_u8_kernel, dim3(csn_grid1((int64_t)h * w)), dim3(CSN_BLOCK), 0, stream, logits, o, H, W, h, w);
  return (int)hipGetLastError();
}
int csn_launch_resize_bilinear(const float* in, float* out, int planes, int Hi, int Wi, int Ho, int Wo, void* stream) {
  CSN_LAUNCH(resize_bilinear_kernel, dim3(csn_grid1((int64_t)planes * Ho * Wo)), dim3(CSN_BLOCK), 0, stream, in, out, planes, Hi,
             Wi, Ho, Wo);
  return (int)hipGetLastError();
}

// an empty launch: csn_forward_profile brackets it with events to calibrate what an event pair adds to a launch
__global__ void csn_nop_kernel() {}
int csn_launch_nop(void* stream) {
  CSN_LAUNCH(csn_nop_kernel, dim3(1), dim3(64), 0, stream);
  return (int)hipGetLastError();
}

int csn_kernels_init(void) { return 0; }
