// k_train.hip -- train-mode BatchNorm (batch statistics) + PReLU + the dynamic-weight-decay GAP term.
//
// Reference semantics:
//   nn.BatchNorm2d in training mode (csnet.py:764,825,138): normalise with the BIASED batch variance over
//   (N,H,W), eps 1e-5; running_mean/var <- 0.9*running + 0.1*batch (UNBIASED variance), momentum 0.1.
//   Oct_bn_hook (csnet.py:391-410): per hooked module and output branch k
//       0.5 * w_k * sum_{n,c} | mean_hw y[n,c] | * gamma_c^2       (y = post-PReLU output, detached)
//   accumulated over the 54 ILBlock sub-modules; get_flops() divides by the batch size (csnet.py:324-330).
//
// Train-mode units run their convolution kernel with an identity epilogue (raw z), then:
//   bn_stats_kernel     per channel: sum z, sum z^2 in fp64 (the reference accumulates float BN statistics in
//                       double on CPU), one partial per (channel, slab), fixed summation order -> deterministic;
//   bn_finalize_kernel  per channel: mean / biased var -> folded scale/shift for the apply pass, running-stat
//                       update in the caller's parameter arena;
//   bn_apply_gap_kernel y = PReLU(z*scale + shift) in place (float4 stream), per-(n,c) plane sum -> penalty.
#include <algorithm>
#include <cstdlib>

#include "csn_kernels.h"
#include "csn_reduce.h"

#define BN_NSLAB CSN_BN_NSLAB   // max slabs per channel; a launch uses gridDim.x <= BN_NSLAB of them

// A channel's S*HW elements are S contiguous planes.  A slab is either one chunk of one plane (cpp > 0 chunks per plane,
// about 8K elements each: slab = n*cpp + chunk) or, for small planes, -cpp whole planes of consecutive images (cpp < 0), so
// that a block streams >= 16K elements before it pays the reduction tail while the launch still has >= 1024 blocks.
static inline int bn_cpp(int S, int C, int64_t hw) {
  int64_t cpp = (hw + 8191) / 8192;
  if (cpp < 1) cpp = 1;
  while (cpp > 1 && cpp * S > BN_NSLAB) --cpp;
  if (cpp == 1) {
    int64_t ipp = 16384 / (hw > 0 ? hw : 1);
    const int64_t cap = (int64_t)S * C / 1024;
    if (ipp > cap) ipp = cap;
    const char* fe = std::getenv("CSN_BN_IPP");   // tests: exercise the multi-image slabs at small batches
    const int forced = fe ? std::atoi(fe) : 0;
    if (forced > 0) ipp = forced < S ? forced : S;
    if (ipp > 1) return (int)-ipp;
  }
  return (int)cpp;
}
static inline int bn_nslab(int S, int cpp) { return cpp > 0 ? S * cpp : (S + (-cpp) - 1) / (-cpp); }

struct BnRange { int64_t base; int beg, end, nimg; };   // element range [beg, end) of `nimg` planes, the first at `base`

__device__ __forceinline__ BnRange bn_range(int slab, int cpp, int S, int C, int c, int64_t hw) {
  BnRange r;
  if (cpp < 0) {
    const int ipp = -cpp, n0 = slab * ipp;
    r.base = ((int64_t)n0 * C + c) * hw;
    r.beg = 0; r.end = (int)hw; r.nimg = min(ipp, S - n0);
    return r;
  }
  const int n = slab / cpp, ch = slab - n * cpp;
  int per = (int)((hw + cpp - 1) / cpp);
  per = (per + 7) & ~7;   // chunks start on a multiple of eight elements (128-bit accesses in the bf16 mode)
  r.base = ((int64_t)n * C + c) * hw;
  r.beg = min(ch * per, (int)hw);
  r.end = min(r.beg + per, (int)hw);
  r.nimg = 1;
  return r;
}

// grid (nslab, C): block (slab, c) reduces one chunk of one plane of channel c.
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void bn_stats_kernel(BnStatsArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int c = blockIdx.y, slab = blockIdx.x;
  const BnRange r = bn_range(slab, a.cpp, a.S, a.C, c, a.HW);
  double s1 = 0.0, s2 = 0.0;
  for (int img = 0; img < r.nimg; ++img) {
    const AT* __restrict__ p = act_cast<AT>(a.z) + r.base + (int64_t)img * a.C * a.HW;
    if ((a.HW & 7) == 0) {   // eight elements per lane and trip
      for (int i = (r.beg >> 3) + threadIdx.x; i < (r.end >> 3); i += CSN_BLOCK) {
        const csn_f8 v = act_ld8(p + 8 * i);
        if (sizeof(AT) == 2) {
          s1 += (double)(((v.v[0] + v.v[1]) + (v.v[2] + v.v[3])) + ((v.v[4] + v.v[5]) + (v.v[6] + v.v[7])));
          s2 += (double)((fmaf(v.v[0], v.v[0], v.v[1] * v.v[1]) + fmaf(v.v[2], v.v[2], v.v[3] * v.v[3])) +
                         (fmaf(v.v[4], v.v[4], v.v[5] * v.v[5]) + fmaf(v.v[6], v.v[6], v.v[7] * v.v[7])));
        } else {
#pragma unroll
          for (int k = 0; k < 8; k += 4) {
            s1 += ((double)v.v[k] + (double)v.v[k + 1]) + ((double)v.v[k + 2] + (double)v.v[k + 3]);
            s2 += ((double)v.v[k] * v.v[k] + (double)v.v[k + 1] * v.v[k + 1]) + ((double)v.v[k + 2] * v.v[k + 2] + (double)v.v[k + 3] * v.v[k + 3]);
          }
        }
      }
    } else if ((a.HW & 3) == 0) {
      for (int i = (r.beg >> 2) + threadIdx.x; i < (r.end >> 2); i += CSN_BLOCK) {
        const float4 v = act_ld4(p + 4 * i);
        if (sizeof(AT) == 2) {   // bfloat16 values: squares exact in fp32, four-term sums in fp32, fp64 across the quads (round 4)
          s1 += (double)((v.x + v.y) + (v.z + v.w));
          s2 += (double)(fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w));
        } else {
          s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
          s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
      }
    } else {
      for (int i = r.beg + threadIdx.x; i < r.end; i += CSN_BLOCK) {
        const double v = (double)act_ld(p + i);
        s1 += v;
        s2 += v * v;
      }
    }
  }
  double sv[2] = {s1, s2};
  bn_block_sum_n<2>(sv, sm);
  if (threadIdx.x == 0) {
    a.partial[((int64_t)c * BN_NSLAB + slab) * 2 + 0] = sv[0];
    a.partial[((int64_t)c * BN_NSLAB + slab) * 2 + 1] = sv[1];
  }
}

struct __attribute__((aligned(16))) BnD2 { double x, y; };   // one slab's {sum, sum of squares}: a 128-bit load
// one block per channel: the slab partials are summed by the block (fixed order), thread 0 finishes
__device__ __forceinline__ void bn_finalize_body(const BnFinalizeArgs& a, double* sm) {
  const int c = blockIdx.x;
  if (c >= a.C) return;   // (block-uniform: the multi-job launch is as wide as its widest job)
  double s1 = 0.0, s2 = 0.0;
  const int64_t ps = a.pstride > 0 ? a.pstride : BN_NSLAB;
  // (eight slabs per trip in flight, added in slab order: pw4_kernel's own statistics are ~12,000 slabs per channel at batch 256)
  const BnD2* pp = reinterpret_cast<const BnD2*>(a.partial) + (int64_t)c * ps;
  int k = threadIdx.x;
  for (; k + 7 * CSN_BLOCK < a.nslab; k += 8 * CSN_BLOCK) {
    BnD2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pp[k + u * CSN_BLOCK];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s1 += v[u].x; s2 += v[u].y; }
  }
  for (; k < a.nslab; k += CSN_BLOCK) { s1 += pp[k].x; s2 += pp[k].y; }
  s1 = bn_block_sum(s1, sm);
  s2 = bn_block_sum(s2, sm);
  if (threadIdx.x != 0) return;
  const double n = (double)a.count;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float gamma = a.arena[a.off_weight + c], beta = a.arena[a.off_bias + c];
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  const float sc = gamma * invstd;
  a.scale[c] = sc;
  a.shift[c] = beta - (float)mean * sc;
  a.mean[c] = (float)mean;       // kept for the backward pass
  a.invstd[c] = invstd;
  const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
  a.arena[a.off_rmean + c] = 0.9f * a.arena[a.off_rmean + c] + 0.1f * (float)mean;
  a.arena[a.off_rvar + c] = 0.9f * a.arena[a.off_rvar + c] + 0.1f * (float)unbiased;
}
__global__ __launch_bounds__(CSN_BLOCK) void bn_finalize_kernel(BnFinalizeArgs a) {
  CSN_DYN_SMEM(double, sm);
  bn_finalize_body(a, sm);
}
// the output branches of ONE unit in one launch (grid (max C, jobs): ~100 launches of ~5 us less per train-mode forward)
__global__ __launch_bounds__(CSN_BLOCK) void bn_finalize3_kernel(BnFinalizeArgs a0, BnFinalizeArgs a1, BnFinalizeArgs a2) {
  CSN_DYN_SMEM(double, sm);
  if (blockIdx.y == 0) bn_finalize_body(a0, sm);
  else if (blockIdx.y == 1) bn_finalize_body(a1, sm);
  else bn_finalize_body(a2, sm);
}

// grid (C, S): block (c, n) streams one plane: y = PReLU(z*scale + shift) (z is kept for the backward pass), and
// the plane sum feeds the penalty 0.5 * w * |mean_hw y| * gamma^2 (fp64 atomic; w == 0: unit is not hooked).
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void bn_apply_gap_kernel(BnApplyArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int c = blockIdx.x, n = blockIdx.y;
  const int64_t hw = a.HW;
  const AT* __restrict__ p = act_cast<AT>(a.z) + ((int64_t)n * a.C + c) * hw;
  AT* __restrict__ q = act_cast<AT>(a.y) + ((int64_t)n * a.C + c) * hw;
  const float sc = a.scale[c], sh = a.shift[c], al = a.alpha[c];
  double s = 0.0;
  if ((hw & 7) == 0) {
    for (int64_t i = threadIdx.x; i < (hw >> 3); i += CSN_BLOCK) {
      csn_f8 v = act_ld8(p + 8 * i);
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] = csn_epi(v.v[k], sc, sh, al);
      act_st8(q + 8 * i, v);
      if (sizeof(AT) == 2) s += (double)(((v.v[0] + v.v[1]) + (v.v[2] + v.v[3])) + ((v.v[4] + v.v[5]) + (v.v[6] + v.v[7])));
      else
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)v.v[k];
    }
  } else if ((hw & 3) == 0) {
    for (int64_t i = threadIdx.x; i < (hw >> 2); i += CSN_BLOCK) {
      float4 v = act_ld4(p + 4 * i);
      v.x = csn_epi(v.x, sc, sh, al); v.y = csn_epi(v.y, sc, sh, al);
      v.z = csn_epi(v.z, sc, sh, al); v.w = csn_epi(v.w, sc, sh, al);
      act_st4(q + 4 * i, v);
      if (sizeof(AT) == 2) s += (double)((v.x + v.y) + (v.z + v.w));
      else s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    }
  } else {
    for (int64_t i = threadIdx.x; i < hw; i += CSN_BLOCK) {
      const float v = csn_epi(act_ld(p + i), sc, sh, al);
      act_st(q + i, v);
      s += (double)v;
    }
  }
  if (a.flop_w != 0.f) {   // |mean_hw y| per (channel, image); bn_penalty_kernel turns the table into the penalty
    s = bn_block_sum(s, sm);
    if (threadIdx.x == 0) a.gapabs[(int64_t)c * a.S + n] = (float)fabs(s / (double)hw);
  }
}

// |mean_hw y| per (channel, image) from the per-tile plane sums the consuming depthwise kernel left (its input y was formed on
// load and never stored, so bn_apply_gap_kernel did not run for it): fixed order over the tiles
__global__ __launch_bounds__(CSN_BLOCK) void gap_tiles_kernel(GapTilesArgs a) {
  const int i = blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (i >= a.C * a.S) return;
  const int c = i / a.S, n = i - c * a.S;
  const double* p = a.gapin + (int64_t)c * BN_NSLAB + (int64_t)n * a.tiles;
  double s = 0.0;
  for (int t = 0; t < a.tiles; ++t) s += p[t];
  a.gapabs[i] = (float)fabs(s / (double)a.HW);
}

__global__ __launch_bounds__(CSN_BLOCK) void gap_tiles_jobs_kernel(GapTilesBatch b) {
  const CSN_CONST_AS GapTilesArgs* a = &CSN_KERNARG(GapTilesBatch, b)->job[blockIdx.y];   // (read in place: no scratch copy)
  const int C = a->C, S = a->S, tiles = a->tiles;
  const int i = blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (i >= C * S) return;
  const int c = i / S, n = i - c * S;
  const double* p = a->gapin + (int64_t)c * BN_NSLAB + (int64_t)n * tiles;
  double s = 0.0;
  for (int t = 0; t < tiles; ++t) s += p[t];
  a->gapabs[i] = (float)fabs(s / (double)a->HW);
}

// penalty += sum_j 0.5 * w_j * sum_c gamma_c^2 * sum_n |gap_j[c][n]|   (Oct_bn_hook, csnet.py:391-410) over all hooked
// (unit, branch) pairs j of the forward.  One block per job (fixed summation order), then ONE thread adds the job terms to the
// device scalar in job order -- the same sequence of fp64 additions as one launch per job, without ~100 tiny launches per
// step; no floating-point atomics.
__global__ __launch_bounds__(CSN_BLOCK) void bn_penalty_jobs_kernel(BnPenaltyArgs a) {
  CSN_DYN_SMEM(double, sm);
  const BnPenaltyJob j = a.job[blockIdx.x];
  double t = 0.0;
  const int total = j.C * j.S;        // lanes walk the [C][S] table linearly (coalesced), fp64 accumulation
  for (int i = threadIdx.x; i < total; i += CSN_BLOCK) {
    const double g = (double)a.arena[j.off_weight + i / j.S];
    t += (double)j.gapabs[i] * (g * g);
  }
  t = bn_block_sum(t, sm);
  if (threadIdx.x == 0) a.partial[a.first + blockIdx.x] = 0.5 * (double)j.flop_w * t;
}
__global__ void bn_penalty_sum_kernel(const double* partial, int n, double* penalty) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double p = *penalty;
  for (int i = 0; i < n; ++i) p += partial[i];
  *penalty = p;
}

// ------------------------------------------------------------------------------------------------ backward
// BN(train) + PReLU backward of one output branch.  With bn = z*scale + shift, xhat = (z - mean)*invstd:
//   y = bn > 0 ? bn : alpha*bn      dbn = dy * (bn > 0 ? 1 : alpha)      dalpha = sum dy*bn over bn <= 0
//   dbeta = sum dbn    dgamma = sum dbn*xhat    dz = gamma*invstd * (dbn - mean(dbn) - xhat*mean(dbn*xhat))
// (ATen batch_norm_backward / prelu_backward semantics).  dy may arrive from two consumers (the stage outputs feed
// both the next stage and the CSF head); the sums are fp64, one partial per (channel, slab) -> deterministic.
template <typename AT>
__device__ __forceinline__ float bnb_dy(const BnBwdArgs& a, int64_t i) {
  float v = act_ld(act_cast<AT>(a.dyA) + i);
  if (a.dyB) v += act_ld(act_cast<AT>(a.dyB) + i);
  return v;
}

template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void bn_bwd_reduce_kernel(BnBwdArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int c = blockIdx.y, slab = blockIdx.x;
  const BnRange r0 = bn_range(slab, a.cpp, a.S, a.C, c, a.HW);
  const float sc = a.scale[c], sh = a.shift[c], al = a.alpha[c], mu = a.mean[c], is = a.invstd[c];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  auto acc = [&](float z, float dy) {
    const float bn = z * sc + sh;
    const float dbn = bn > 0.f ? dy : al * dy;
    s0 += (double)dbn;
    s1 += (double)dbn * (double)((z - mu) * is);
    if (!(bn > 0.f)) s2 += (double)dy * (double)bn;
  };
  for (int img = 0; img < r0.nimg; ++img) {
  BnRange r = r0;
  r.base += (int64_t)img * a.C * a.HW;
  if ((a.HW & 7) == 0) {   // eight elements per lane and trip: fp32 over the eight, fp64 across (float: per element as below)
    const AT* z8 = act_cast<AT>(a.z) + r.base;
    const AT* a8 = act_cast<AT>(a.dyA) + r.base;
    const AT* b8 = a.dyB ? act_cast<AT>(a.dyB) + r.base : nullptr;
    for (int i = (r.beg >> 3) + threadIdx.x; i < (r.end >> 3); i += CSN_BLOCK) {
      const csn_f8 z = act_ld8(z8 + 8 * i);
      csn_f8 d = act_ld8(a8 + 8 * i);
      if (b8) {
        const csn_f8 e = act_ld8(b8 + 8 * i);
#pragma unroll
        for (int k = 0; k < 8; ++k) d.v[k] += e.v[k];
      }
      if (sizeof(AT) == 2) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float bn = z.v[k] * sc + sh;
          const float dbn = bn > 0.f ? d.v[k] : al * d.v[k];
          t0 += dbn;
          t1 = fmaf(dbn, (z.v[k] - mu) * is, t1);
          t2 = fmaf(bn > 0.f ? 0.f : d.v[k], bn, t2);
        }
        s0 += (double)t0; s1 += (double)t1; s2 += (double)t2;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc(z.v[k], d.v[k]);
      }
    }
  } else if ((a.HW & 3) == 0) {   // chunks start on float4 boundaries (bn_range)
    const AT* z4 = act_cast<AT>(a.z) + r.base;
    const AT* a4 = act_cast<AT>(a.dyA) + r.base;
    const AT* b4 = a.dyB ? act_cast<AT>(a.dyB) + r.base : nullptr;
    for (int i = (r.beg >> 2) + threadIdx.x; i < (r.end >> 2); i += CSN_BLOCK) {
      const float4 z = act_ld4(z4 + 4 * i);
      float4 d = act_ld4(a4 + 4 * i);
      if (b4) { const float4 e = act_ld4(b4 + 4 * i); d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
      if (sizeof(AT) == 2) {   // bf16 storage (round 4): fp32 over the quad, fp64 across quads -- 6 instead of ~28 double-rate operations
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        auto acc4 = [&](float zz, float dy) {
          const float bn = zz * sc + sh;
          const float dbn = bn > 0.f ? dy : al * dy;
          t0 += dbn;
          t1 = fmaf(dbn, (zz - mu) * is, t1);
          t2 = fmaf(bn > 0.f ? 0.f : dy, bn, t2);
        };
        acc4(z.x, d.x); acc4(z.y, d.y); acc4(z.z, d.z); acc4(z.w, d.w);
        s0 += (double)t0; s1 += (double)t1; s2 += (double)t2;
      } else {
        acc(z.x, d.x); acc(z.y, d.y); acc(z.z, d.z); acc(z.w, d.w);
      }
    }
  } else {
    for (int i = r.beg + threadIdx.x; i < r.end; i += CSN_BLOCK)
      acc(act_ld(act_cast<AT>(a.z) + r.base + i), bnb_dy<AT>(a, r.base + i));
  }
  }
  double sv[3] = {s0, s1, s2};
  bn_block_sum_n<3>(sv, sm);
  if (threadIdx.x == 0) {
    double* o = a.partial + ((int64_t)c * BN_NSLAB + slab) * 3;
    o[0] = sv[0]; o[1] = sv[1]; o[2] = sv[2];
  }
}

__device__ __forceinline__ void bn_bwd_finalize_body(const BnBwdArgs& a, double* sm) {
  const int c = blockIdx.x;
  if (c >= a.C) return;   // (block-uniform)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < a.nslab; k += CSN_BLOCK) {
    const double* o = a.partial + ((int64_t)c * BN_NSLAB + k) * 3;
    s0 += o[0]; s1 += o[1]; s2 += o[2];
  }
  s0 = bn_block_sum(s0, sm);
  s1 = bn_block_sum(s1, sm);
  s2 = bn_block_sum(s2, sm);
  double sg = 0.0;
  const bool pen = a.flop_w != 0.f && a.gapabs;
  if (pen) {
    for (int k = threadIdx.x; k < a.S; k += CSN_BLOCK) sg += (double)a.gapabs[(int64_t)c * a.S + k];
    sg = bn_block_sum(sg, sm);
  }
  if (threadIdx.x != 0) return;
  const double n = (double)a.S * (double)a.HW;
  a.m1m2[2 * c + 0] = (float)(s0 / n);
  a.m1m2[2 * c + 1] = (float)(s1 / n);
  const double gamma = (double)a.arena[a.off_weight + c];
  double dgamma = s1;
  if (pen) dgamma += (double)a.pen_scale * (double)a.flop_w * sg * gamma;   // d/dgamma of 0.5 w sum_n|gap| gamma^2
  a.grad[a.off_weight + c] = (float)dgamma;
  a.grad[a.off_bias + c] = (float)s0;
  a.grad[a.off_prelu + c] = (float)s2;
}
__global__ __launch_bounds__(CSN_BLOCK) void bn_bwd_finalize_kernel(BnBwdArgs a) {
  CSN_DYN_SMEM(double, sm);
  bn_bwd_finalize_body(a, sm);
}
__global__ __launch_bounds__(CSN_BLOCK) void bn_bwd_finalize3_kernel(BnBwdArgs a0, BnBwdArgs a1, BnBwdArgs a2) {
  CSN_DYN_SMEM(double, sm);
  if (blockIdx.y == 0) bn_bwd_finalize_body(a0, sm);
  else if (blockIdx.y == 1) bn_bwd_finalize_body(a1, sm);
  else bn_bwd_finalize_body(a2, sm);
}

// grid (C, S): dz written over z (same index, same thread)
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void bn_bwd_apply_kernel(BnBwdArgs a) {
  const int c = blockIdx.x, n = blockIdx.y;
  const int64_t hw = a.HW;
  const int64_t base = ((int64_t)n * a.C + c) * hw;
  const float sc = a.scale[c], sh = a.shift[c], al = a.alpha[c], mu = a.mean[c], is = a.invstd[c];
  const float m1 = a.m1m2[2 * c], m2 = a.m1m2[2 * c + 1];
  const float gi = a.arena[a.off_weight + c] * is;
  auto f = [&](float z, float dy) {
    const float bn = z * sc + sh;
    const float dbn = bn > 0.f ? dy : al * dy;
    return gi * (dbn - m1 - (z - mu) * is * m2);
  };
  if ((hw & 7) == 0) {
    AT* z8 = act_cast<AT>(a.z) + base;
    const AT* a8 = act_cast<AT>(a.dyA) + base;
    const AT* b8 = a.dyB ? act_cast<AT>(a.dyB) + base : nullptr;
    for (int64_t i = threadIdx.x; i < (hw >> 3); i += CSN_BLOCK) {
      csn_f8 z = act_ld8(z8 + 8 * i), d = act_ld8(a8 + 8 * i);
      if (b8) {
        const csn_f8 e = act_ld8(b8 + 8 * i);
#pragma unroll
        for (int k = 0; k < 8; ++k) d.v[k] += e.v[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) z.v[k] = f(z.v[k], d.v[k]);
      act_st8(z8 + 8 * i, z);
    }
  } else if ((hw & 3) == 0) {
    AT* z4 = act_cast<AT>(a.z) + base;
    const AT* a4 = act_cast<AT>(a.dyA) + base;
    const AT* b4 = a.dyB ? act_cast<AT>(a.dyB) + base : nullptr;
    for (int64_t i = threadIdx.x; i < (hw >> 2); i += CSN_BLOCK) {
      float4 z = act_ld4(z4 + 4 * i), d = act_ld4(a4 + 4 * i);
      if (b4) { const float4 e = act_ld4(b4 + 4 * i); d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
      z.x = f(z.x, d.x); z.y = f(z.y, d.y); z.z = f(z.z, d.z); z.w = f(z.w, d.w);
      act_st4(z4 + 4 * i, z);
    }
  } else {
    AT* zp = act_cast<AT>(a.z) + base;
    for (int64_t i = threadIdx.x; i < hw; i += CSN_BLOCK) act_st(zp + i, f(act_ld(zp + i), bnb_dy<AT>(a, base + i)));
  }
}

// The apply pass of a HIGH output branch of a gOctConv unit, and in the same pass the adjoint of the bilinear x2 upsampling
// (csnet.py:702-707: y_hi += up(conv(x_lo))) of the dz it writes:  adj[yl][xl] = sum_{yh, xh} w(yh -> yl) w(xh -> xl) dz[yh][xh],
// w = the forward's own weights (PyTorch align_corners=False: 0.25 / 0.75 towards the two nearest sources, both onto the border
// source where the index clamps), i.e. per axis  adj[s] = .25 dz[2s - 1] + .75 dz[2s] + .75 dz[2s + 1] + .25 dz[2s + 2]  with the
// .75 next to a border replaced by 1.  adjup2_pair_kernel computed the same sums from the stored dz (one more pass over the
// largest gradient tensor of every unit: 2.1 ms of the 49 ms bf16 step); here a lane owns 8 columns of a strip of rows, walks
// down two rows per step with the two previous row sums in registers, and forms dz for its columns plus one on either side
// (the neighbours' values are recomputed from dy and z -- the same arithmetic, hence the same value -- not exchanged).
// grid (C, S), one plane per block; W % 8 == 0, H even.  dz is bit for bit bn_bwd_apply_kernel's; the sums use the values AS
// STORED (rounded through bfloat16 in that mode), like the two-pass scheme.  dz is NOT written over z (lanes read their neighbours'
// z and dy): it goes to a.dz_out.
// eight consecutive elements of a row plus the one on either side, AS LOADED (no conversion: a row that is to stay in flight while
// the previous one is worked on must not be touched)
template <typename AT> struct BnRow8;
template <> struct BnRow8<float> { float4 a, b; float l, r; };
template <> struct BnRow8<csn_bf16> { uint2 a, b; unsigned short l, r; };
__device__ __forceinline__ BnRow8<float> bn_row8_ld(const float* p, bool has_l, bool has_r) {
  BnRow8<float> q;
  q.a = *reinterpret_cast<const float4*>(p); q.b = *reinterpret_cast<const float4*>(p + 4);
  q.l = p[has_l ? -1 : 0]; q.r = p[has_r ? 8 : 7];
  return q;
}
__device__ __forceinline__ BnRow8<csn_bf16> bn_row8_ld(const csn_bf16* p, bool has_l, bool has_r) {
  BnRow8<csn_bf16> q;
  q.a = *reinterpret_cast<const uint2*>(p); q.b = *reinterpret_cast<const uint2*>(p + 4);
  q.l = p[has_l ? -1 : 0].u; q.r = p[has_r ? 8 : 7].u;
  return q;
}
__device__ __forceinline__ void bn_row8_f(const BnRow8<float>& q, float (&v)[10]) {
  v[0] = q.l; v[1] = q.a.x; v[2] = q.a.y; v[3] = q.a.z; v[4] = q.a.w; v[5] = q.b.x; v[6] = q.b.y; v[7] = q.b.z; v[8] = q.b.w; v[9] = q.r;
}
__device__ __forceinline__ void bn_row8_f(const BnRow8<csn_bf16>& q, float (&v)[10]) {
  v[0] = csn_bf2f(q.l);
  v[1] = csn_bits_f(q.a.x << 16); v[2] = csn_bits_f(q.a.x & 0xffff0000u); v[3] = csn_bits_f(q.a.y << 16); v[4] = csn_bits_f(q.a.y & 0xffff0000u);
  v[5] = csn_bits_f(q.b.x << 16); v[6] = csn_bits_f(q.b.x & 0xffff0000u); v[7] = csn_bits_f(q.b.y << 16); v[8] = csn_bits_f(q.b.y & 0xffff0000u);
  v[9] = csn_bf2f(q.r);
}

template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void bn_bwd_apply_adj2_kernel(BnBwdArgs a) {
  const int c = blockIdx.x, n = blockIdx.y;
  const int W = a.W, H = (int)(a.HW / W), Wl = W >> 1, Hl = H >> 1;
  const int64_t base = ((int64_t)n * a.C + c) * a.HW;
  const float sc = a.scale[c], sh = a.shift[c], al = a.alpha[c], mu = a.mean[c], is = a.invstd[c];
  const float m1 = a.m1m2[2 * c], m2 = a.m1m2[2 * c + 1];
  const float gi = a.arena[a.off_weight + c] * is;
  auto f = [&](float z, float dy) {
    const float bn = z * sc + sh;
    const float dbn = bn > 0.f ? dy : al * dy;
    const float v = gi * (dbn - m1 - (z - mu) * is * m2);
    return sizeof(AT) == 2 ? csn_bf2f(csn_f2bf(v)) : v;   // the value the consumers of dz will read back
  };
  const int LX = W >> 3;                       // lanes per row
  const int NS = CSN_BLOCK / LX;               // row strips per block
  const int tid = threadIdx.x;
  const int lx = tid % LX, st = tid / LX;
  const int lrows = (Hl + NS - 1) / NS;        // low rows per strip
  const int yl0 = st * lrows, yl1 = min(Hl, yl0 + lrows);
  if (st >= NS || yl0 >= yl1) return;
  const int x0 = lx * 8;
  const AT* zp = act_cast<AT>(a.z) + base + x0;
  const AT* ap = act_cast<AT>(a.dyA) + base + x0;
  const AT* bp = a.dyB ? act_cast<AT>(a.dyB) + base + x0 : nullptr;
  AT* dzp = act_cast<AT>(a.dz_out) + base + x0;
  AT* op = act_cast<AT>(a.adj2) + ((int64_t)n * a.C + c) * (int64_t)Hl * Wl + (x0 >> 1);
  const bool has_l = x0 > 0, has_r = x0 + 8 < W;
  const float wfirst = has_l ? 0.75f : 1.f, wlast = has_r ? 0.75f : 1.f;
  // The strip's output rows r0 .. r1 in one loop (2 yl0 - 1: the strip above owns and stores it, only its sums are needed; 2 yl1: the
  // strip below's): the loads of row r + 1 are issued BEFORE row r is converted and used (a whole trip in flight), rows outside
  // the plane are fetched from a clamped row and ignored.
  const int r0 = 2 * yl0 - 1, r1 = 2 * yl1;
  auto rowoff = [&](int r) { return (int64_t)min(max(r, 0), H - 1) * W; };
  BnRow8<AT> qz = bn_row8_ld(zp + rowoff(r0), has_l, has_r), qa = bn_row8_ld(ap + rowoff(r0), has_l, has_r), qb = qz;
  if (bp) qb = bn_row8_ld(bp + rowoff(r0), has_l, has_r);
  float h3[4] = {0.f, 0.f, 0.f, 0.f}, h2[4] = {0.f, 0.f, 0.f, 0.f}, h1[4] = {0.f, 0.f, 0.f, 0.f};   // row sums of r - 3, r - 2, r - 1
  for (int r = r0; r <= r1; ++r) {
    const int64_t on = rowoff(r + 1);
    const BnRow8<AT> nz = bn_row8_ld(zp + on, has_l, has_r), na = bn_row8_ld(ap + on, has_l, has_r);
    BnRow8<AT> nb = nz;
    if (bp) nb = bn_row8_ld(bp + on, has_l, has_r);
#ifndef CSN_CPU_EMU
    __builtin_amdgcn_sched_barrier(0);   // the loads above stay above the conversions below
#endif
    float h[4] = {0.f, 0.f, 0.f, 0.f};
    if (r >= 0 && r < H) {
      float z[10], d[10], v[10];
      bn_row8_f(qz, z); bn_row8_f(qa, d);
      if (bp) {
        float e[10];
        bn_row8_f(qb, e);
#pragma unroll
        for (int i = 0; i < 10; ++i) d[i] += e[i];
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) v[i] = f(z[i], d[i]);
      if (!has_l) v[0] = 0.f;
      if (!has_r) v[9] = 0.f;
      if (r >= 2 * yl0 && r < 2 * yl1) {
        act_st4(dzp + (int64_t)r * W, make_float4(v[1], v[2], v[3], v[4]));
        act_st4(dzp + (int64_t)r * W + 4, make_float4(v[5], v[6], v[7], v[8]));
      }
      h[0] = 0.25f * v[0] + wfirst * v[1] + 0.75f * v[2] + 0.25f * v[3];
      h[1] = 0.25f * v[2] + 0.75f * v[3] + 0.75f * v[4] + 0.25f * v[5];
      h[2] = 0.25f * v[4] + 0.75f * v[5] + 0.75f * v[6] + 0.25f * v[7];
      h[3] = 0.25f * v[6] + 0.75f * v[7] + wlast * v[8] + 0.25f * v[9];
    }
    if (((r & 1) == 0) && r >= 2 * yl0 + 2) {   // row 2 yl + 2 closes low row yl: rows 2 yl - 1 .. 2 yl + 2 = h3, h2, h1, h
      const int yl = (r >> 1) - 1;
      const float wa = yl > 0 ? 0.75f : 1.f, wb = yl + 1 < Hl ? 0.75f : 1.f;
      act_st4(op + (int64_t)yl * Wl, make_float4(0.25f * h3[0] + wa * h2[0] + wb * h1[0] + 0.25f * h[0],
                                                  0.25f * h3[1] + wa * h2[1] + wb * h1[1] + 0.25f * h[1],
                                                  0.25f * h3[2] + wa * h2[2] + wb * h1[2] + 0.25f * h[2],
                                                  0.25f * h3[3] + wa * h2[3] + wb * h1[3] + 0.25f * h[3]));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { h3[k] = h2[k]; h2[k] = h1[k]; h1[k] = h[k]; }
    qz = nz; qa = na; qb = nb;
  }
}

bool csn_bn_bwd_adj2_ok(int64_t HW, int W) {
  const bool off = std::getenv("CSN_ADJ_FUSE") && std::getenv("CSN_ADJ_FUSE")[0] == '0';
  if (off || W <= 0 || (W & 7) != 0 || HW % W != 0 || W / 8 > CSN_BLOCK) return false;
  return ((HW / W) & 1) == 0;
}

// depthwise 3x3 weight gradient: dW[c][t] = 100 * sum_{n,p} dz[n,c,p] * x[n,c,p + off(t)]   (conv2d.py:104)
// A lane owns QUADS of four consecutive pixels of a row (W % 4 == 0, chunks start on multiples of four): one vector load
// of dz, three of x (rows y - 1, y, y + 1; rows outside the plane fall out of the bounded buffer range and read as 0) plus
// the two edge columns per row -- 2.5 load instructions per pixel instead of 10.  <= 8 quads per lane: fp32 partials.
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void dw_wgrad_kernel(DwWgradArgs a) {
  CSN_DYN_SMEM(double, sm);
  constexpr unsigned E = (unsigned)sizeof(AT);
  const int c = blockIdx.y, slab = blockIdx.x;
  const int H = a.H, W = a.W;
  const BnRange r = bn_range(slab, a.cpp, a.S, a.C, c, (int64_t)H * W);
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  for (int img = 0; img < r.nimg; ++img) {
  const AT* __restrict__ gp = act_cast<AT>(a.dz) + r.base + (int64_t)img * a.C * H * W;
  const AT* __restrict__ xp = act_cast<AT>(a.x) + r.base + (int64_t)img * a.C * H * W;
  if ((W & 3) == 0) {
    const csn_buf xb = csn_make_buf_n(xp, (unsigned)(H * W) * E);
    for (int q = (r.beg >> 2) + threadIdx.x; q < (r.end >> 2); q += CSN_BLOCK) {
      const int p = q << 2;
      const int y = p / W, x0 = p - y * W;
      const float4 g = act_ld4(gp + p);
      const bool has_l = x0 > 0, has_r = x0 + 4 < W;
      float v[3][6];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const unsigned o = (unsigned)((y + dy - 1) * W + x0) * E;   // row -1 wraps to a huge offset, row H is past the plane: 0
        const float4 cv = csn_bufacc<AT>::ld4(xb, o, 0);
        const float l = csn_bufacc<AT>::ld1(xb, o - E, 0), rr = csn_bufacc<AT>::ld1(xb, o + 4u * E, 0);
        v[dy][0] = has_l ? l : 0.f; v[dy][1] = cv.x; v[dy][2] = cv.y; v[dy][3] = cv.z; v[dy][4] = cv.w; v[dy][5] = has_r ? rr : 0.f;
      }
      const float gq[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[t] = fmaf(gq[j], v[t / 3][t % 3 + j], s[t]);
    }
  } else {
    for (int p = r.beg + threadIdx.x; p < r.end; p += CSN_BLOCK) {   // <= 32 terms per lane: fp32 partials
      const int y = p / W, x = p - y * W;
      const float g = act_ld(gp + p);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float v = act_ld(xp + (in ? yy * W + xx : p));
        s[t] = fmaf(g, in ? v : 0.f, s[t]);
      }
    }
  }
  }
  double sv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) sv[t] = (double)s[t];
  bn_block_sum_n<9>(sv, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) a.partial[((int64_t)c * BN_NSLAB + slab) * 9 + t] = sv[t];
  }
}

// one block per channel
__global__ __launch_bounds__(CSN_BLOCK) void dw_wgrad_finalize_kernel(DwWgradArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int c = blockIdx.x;
  double s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.0;
  for (int k = threadIdx.x; k < a.nslab; k += CSN_BLOCK)
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] += a.partial[((int64_t)c * BN_NSLAB + k) * 9 + t];
  bn_block_sum_n<9>(s, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) a.grad[a.off_w + c * 9 + t] = (float)(100.0 * s[t]);
  }
}

// grid (max C, jobs): the finalise pass of many depthwise units at once (same arithmetic, same order per channel)
__global__ __launch_bounds__(CSN_BLOCK) void dw_wgrad_finalize_jobs_kernel(DwFinBatch b) {
  CSN_DYN_SMEM(double, sm);
  const CSN_CONST_AS DwFinBatch* bp = CSN_KERNARG(DwFinBatch, b);
  const CSN_CONST_AS DwFinJob* j = &bp->job[blockIdx.y];
  const int c = blockIdx.x, nslab = j->nslab;
  if (c >= j->C) return;
  const double* part = j->partial;
  double s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.0;
  for (int k = threadIdx.x; k < nslab; k += CSN_BLOCK)
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] += part[((int64_t)c * BN_NSLAB + k) * 9 + t];
  bn_block_sum_n<9>(s, sm);
  if (threadIdx.x == 0) {
    float* g = bp->grad + j->off_w + c * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = (float)(100.0 * s[t]);
  }
}

// adjoint of F.interpolate(scale_factor=f, mode='bilinear', align_corners=False): out[lo] = sum_hi w(hi->lo) in[hi].
// Source pixel s receives from the outputs o in [f*s - f/2, f*s + f + f/2 - 1] (2f candidates per axis, f = 2 or 4);
// the weight of each candidate is read off the forward's own index computation (csn_bilin), so the border clamping
// is the adjoint of exactly what the forward did.
template <typename TI, typename TO>
__global__ __launch_bounds__(CSN_BLOCK) void adjup_kernel(AdjUpArgs a) {
  const int Hl = a.Hl, Wl = a.Wl, f = a.f;
  const int Hh = Hl * f, Wh = Wl * f;
  const int64_t tot = (int64_t)a.planes * Hl * Wl;
  const float inv = 1.f / (float)f;
  for (int64_t e = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; e < tot; e += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t pl = e / (Hl * Wl);
    const int r = (int)(e - pl * Hl * Wl);
    const int ys = r / Wl, xs = r - ys * Wl;
    const TI* ip = act_cast<TI>(a.in) + pl * (int64_t)Hh * Wh;
    float wx[8];
    const int ox0 = xs * f - (f >> 1), oy0 = ys * f - (f >> 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ox = ox0 + j;
      int x0, x1; float lx;
      csn_bilin(min(max(ox, 0), Wh - 1), inv, Wl, x0, x1, lx);
      const float w = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
      wx[j] = (j < 2 * f && ox >= 0 && ox < Wh) ? w : 0.f;
    }
    float acc = 0.f;
    for (int i = 0; i < 2 * f; ++i) {
      const int oy = oy0 + i;
      if (oy < 0 || oy >= Hh) continue;
      int y0, y1; float ly;
      csn_bilin(oy, inv, Hl, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      const TI* row = ip + (int64_t)oy * Wh;
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < 2 * f) t = fmaf(wx[j], act_ld(row + min(max(ox0 + j, 0), Wh - 1)), t);
      acc = fmaf(wy, t, acc);
    }
    act_st(act_cast<TO>(a.out) + e, acc);
  }
}

// f = 2, even source width: one thread produces TWO neighbouring source pixels (xs even) from the 4 x 6 window of
// outputs they receive from -- per row one aligned float4 (columns 2 xs .. 2 xs + 3) and the two edge columns.
template <typename TI, typename TO>
__global__ __launch_bounds__(CSN_BLOCK) void adjup2_pair_kernel(AdjUpArgs a) {
  const int Hl = a.Hl, Wl = a.Wl, Wp = Wl >> 1;
  const int Hh = Hl * 2, Wh = Wl * 2;
  const int64_t tot = (int64_t)a.planes * Hl * Wp;
  for (int64_t e = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; e < tot; e += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t pl = e / (Hl * Wp);
    const int r = (int)(e - pl * Hl * Wp);
    const int ys = r / Wp, xs = (r - ys * Wp) * 2;
    const TI* ip = act_cast<TI>(a.in) + pl * (int64_t)Hh * Wh;
    // column weights of the six outputs 2 xs - 1 .. 2 xs + 4 towards source columns xs and xs + 1
    float w0[6], w1[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ox = 2 * xs - 1 + j;
      int x0, x1; float lx;
      csn_bilin(min(max(ox, 0), Wh - 1), 0.5f, Wl, x0, x1, lx);
      const bool in = ox >= 0 && ox < Wh;
      w0[j] = in ? (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f) : 0.f;
      w1[j] = in ? (x0 == xs + 1 ? 1.f - lx : 0.f) + (x1 == xs + 1 ? lx : 0.f) : 0.f;
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = 2 * ys - 1 + i;
      if (oy < 0 || oy >= Hh) continue;
      int y0, y1; float ly;
      csn_bilin(oy, 0.5f, Hl, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      const TI* row = ip + (int64_t)oy * Wh + 2 * xs;
      const float4 c = act_ld4(row);
      const float l = act_ld(row + (xs > 0 ? -1 : 0)), rr = act_ld(row + (2 * xs + 4 < Wh ? 4 : 3));
      const float v[6] = {l, c.x, c.y, c.z, c.w, rr};
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) { t0 = fmaf(w0[j], v[j], t0); t1 = fmaf(w1[j], v[j], t1); }
      a0 = fmaf(wy, t0, a0);
      a1 = fmaf(wy, t1, a1);
    }
    TO* op = act_cast<TO>(a.out) + pl * (int64_t)Hl * Wl + (int64_t)ys * Wl + xs;
    act_st2(op, make_float2(a0, a1));
  }
}

// f = 4, W % 8 == 0, H % 4 == 0 (CSFHead.fuse / fuse1x1: dz of the 112^2 branch towards the 28^2 one): the generic kernel above gathers
// an 8 x 8 window per source pixel with scalar loads and recomputes the forward's index arithmetic per tap (0.45 ms per launch for
// 0.5 GB).  Here a lane owns 8 columns (two source columns) of a strip of rows and walks down them once: per row one 128-bit load
// plus the pair of columns on either side, the row reduced along x with the 8 constant weights of the x4 adjoint
// (.125 .375 .625 .875 .875 .625 .375 .125; next to a border the two clamped outputs carry weight 1), and added into the two source
// rows it belongs to.  grid = planes.
template <typename TI, typename TO>
__global__ __launch_bounds__(CSN_BLOCK) void adjup4_rows_kernel(AdjUpArgs a) {
  const int Hl = a.Hl, Wl = a.Wl, H = Hl * 4, W = Wl * 4;
  const int64_t pl = blockIdx.x;
  const TI* ip = act_cast<TI>(a.in) + pl * (int64_t)H * W;
  TO* op = act_cast<TO>(a.out) + pl * (int64_t)Hl * Wl;
  const int LX = W >> 3, NS = CSN_BLOCK / LX;
  const int tid = threadIdx.x, lx = tid % LX, st = tid / LX;
  const int lrows = (Hl + NS - 1) / NS;
  const int yl0 = st * lrows, yl1 = min(Hl, yl0 + lrows);
  if (st >= NS || yl0 >= yl1) return;
  const int x0 = lx * 8;
  const bool has_l = x0 > 0, has_r = x0 + 8 < W;
  const float w8[8] = {0.125f, 0.375f, 0.625f, 0.875f, 0.875f, 0.625f, 0.375f, 0.125f};
  float wa[8], wb[8];   // weights of columns x0 - 2 .. x0 + 5 towards source column x0 / 4, of x0 + 2 .. x0 + 9 towards x0 / 4 + 1
#pragma unroll
  for (int i = 0; i < 8; ++i) { wa[i] = w8[i]; wb[i] = w8[i]; }
  if (!has_l) { wa[2] = 1.f; wa[3] = 1.f; }    // outputs 0 and 1 clamp onto source column 0
  if (!has_r) { wb[4] = 1.f; wb[5] = 1.f; }    // outputs W - 2, W - 1 onto the last one
  auto row = [&](int yh, float& h0, float& h1) {
    const TI* rp = ip + (int64_t)yh * W + x0;
    const float4 c0 = act_ld4(rp), c1 = act_ld4(rp + 4);
    float2 l = make_float2(0.f, 0.f), r = make_float2(0.f, 0.f);
    if (has_l) l = act_ld2(rp - 2);
    if (has_r) r = act_ld2(rp + 8);
    const float v[12] = {l.x, l.y, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, r.x, r.y};
    h0 = 0.f; h1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { h0 = fmaf(wa[i], v[i], h0); h1 = fmaf(wb[i], v[4 + i], h1); }
  };
  // source row yl receives from the output rows 4 yl - 2 .. 4 yl + 5 with the same weights (clamped rows: weight 1)
  for (int yl = yl0; yl < yl1; ++yl) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int yh = 4 * yl - 2 + i;
      if (yh < 0 || yh >= H) continue;
      float wv = w8[i];
      if (yl == 0 && (i == 2 || i == 3)) wv = 1.f;
      if (yl == Hl - 1 && (i == 4 || i == 5)) wv = 1.f;
      float h0, h1;
      row(yh, h0, h1);
      a0 = fmaf(wv, h0, a0);
      a1 = fmaf(wv, h1, a1);
    }
    act_st2(op + (int64_t)yl * Wl + (x0 >> 2), make_float2(a0, a1));
  }
}

// adjoint of avg_pool2d(2, 2): dx[p] = 0.25 * dxp[p >> 1]
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void avgpool2_bwd_kernel(PoolBwdArgs a) {
  const int Hh = a.Hl * 2, Wh = a.Wl * 2;
  const int64_t tot = (int64_t)a.planes * Hh * Wh;
  for (int64_t e = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; e < tot; e += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t pl = e / ((int64_t)Hh * Wh);
    const int r = (int)(e - pl * Hh * Wh);
    const int y = r / Wh, x = r - y * Wh;
    act_st(act_cast<AT>(a.dx) + e, 0.25f * act_ld(act_cast<AT>(a.t) + pl * (int64_t)a.Hl * a.Wl + (y >> 1) * a.Wl + (x >> 1)));
  }
}

// backward of max_pool2d(f, f): the window's gradient goes to its FIRST maximum in row-major order (ATen's
// `val > maxval` scan), added to dx (the own-resolution term was written before).  One thread per window.
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void maxpool_bwd_add_kernel(PoolBwdArgs a) {
  const int f = a.f, Hl = a.Hl, Wl = a.Wl;
  const int Wh = Wl * f;
  const int64_t hwh = (int64_t)Hl * f * Wh;
  const int64_t tot = (int64_t)a.planes * Hl * Wl;
  for (int64_t e = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; e < tot; e += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t pl = e / (Hl * Wl);
    const int r = (int)(e - pl * Hl * Wl);
    const int yl = r / Wl, xl = r - yl * Wl;
    const AT* xp = act_cast<AT>(a.x) + pl * hwh + (int64_t)(yl * f) * Wh + xl * f;
    float best = act_ld(xp);
    int bi = 0;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) {
        const float v = act_ld(xp + dy * Wh + dx);
        if (v > best || v != v) { best = v; bi = dy * Wh + dx; }
      }
    AT* dp = act_cast<AT>(a.dx) + pl * hwh + (int64_t)(yl * f) * Wh + xl * f + bi;
    act_st(dp, act_ld(dp) + act_ld(act_cast<AT>(a.t) + e));
  }
}

// dst[0] = sum in[0..n)   (cls bias gradient): per-block fp64 partials, then one block in a fixed order
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void sum_partial_kernel(const float* in, int64_t n, double* partial) {
  CSN_DYN_SMEM(double, sm);
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK)
    s += (double)act_ld(act_cast<AT>(in) + i);
  s = bn_block_sum(s, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(CSN_BLOCK) void sum_final_kernel(const double* partial, int nblk, float* dst) {
  CSN_DYN_SMEM(double, sm);
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += CSN_BLOCK) s += partial[i];
  s = bn_block_sum(s, sm);
  if (threadIdx.x == 0) dst[0] = (float)s;
}

// mean BCE-with-logits (train.py:209) and its gradient: loss = mean(max(y,0) - y*t + log1p(exp(-|y|))),
// dy = (sigmoid(y) - t) / n.  One fp64 partial per block into a library-owned table, summed in block order by
// bce_final_kernel: the loss is bit-reproducible run to run (no floating-point atomics anywhere in the step).
#define BCE_MAX_BLOCKS 1024
__device__ double g_bce_partial[BCE_MAX_BLOCKS];
__global__ __launch_bounds__(CSN_BLOCK) void bce_logits_kernel(const float* y, const float* t, float* dy, int64_t n) {
  CSN_DYN_SMEM(double, sm);
  double s = 0.0;
  const float inv = 1.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const float v = y[i], tg = t[i];
    const float e = expf(-fabsf(v));
    s += (double)(fmaxf(v, 0.f) - v * tg + log1pf(e));
    const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dy[i] = (sig - tg) * inv;
  }
  s = bn_block_sum(s, sm);
  if (threadIdx.x == 0) g_bce_partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(CSN_BLOCK) void bce_final_kernel(int nblk, int64_t n, double* loss) {
  CSN_DYN_SMEM(double, sm);
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += CSN_BLOCK) s += g_bce_partial[i];
  s = bn_block_sum(s, sm);
  if (threadIdx.x == 0) *loss += s / (double)n;
}

// torch.optim.Adam step (L2 weight decay folded into the gradient, train.py:108-123) over the flat parameter arena;
// wd[i] carries the per-parameter-group weight decay.
__global__ __launch_bounds__(CSN_BLOCK) void adam_kernel(AdamArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * CSN_BLOCK) {
    const float w = a.p[i];
    const float g = a.g[i] + a.wd[i] * w;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    const float denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
    a.p[i] = w - a.step_size * (m / denom);
  }
}

// launch the float or the bfloat16 instantiation of a kernel template
#define CSN_LAUNCH_AT(a16, kern, grid, block, smem, stream, ...)                                   \
  do {                                                                                              \
    if (a16) CSN_LAUNCH((kern<csn_bf16>), grid, block, smem, stream, __VA_ARGS__);                  \
    else CSN_LAUNCH((kern<float>), grid, block, smem, stream, __VA_ARGS__);                         \
  } while (0)

int csn_launch_bn_stats(const BnStatsArgs& a0, void* stream) {
  BnStatsArgs a = a0;
  a.cpp = bn_cpp(a.S, a.C, a.HW);
  CSN_LAUNCH_AT(a.a16, bn_stats_kernel, dim3(bn_nslab(a.S, a.cpp), a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
int csn_launch_bn_finalize(const BnFinalizeArgs& a0, void* stream) {
  BnFinalizeArgs a = a0;
  if (a.nslab <= 0) a.nslab = bn_nslab(a.S, bn_cpp(a.S, a.C, a.count / a.S));
  CSN_LAUNCH(bn_finalize_kernel, dim3(a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
int csn_launch_bn_finalize_n(const BnFinalizeArgs* jobs, int n, void* stream) {
  if (n <= 0) return 0;
  BnFinalizeArgs a[3];
  int mx = 1;
  for (int i = 0; i < 3; ++i) {
    a[i] = jobs[i < n ? i : 0];
    if (a[i].nslab <= 0) a[i].nslab = bn_nslab(a[i].S, bn_cpp(a[i].S, a[i].C, a[i].count / a[i].S));
    if (i < n) mx = std::max(mx, (int)a[i].C);
  }
  if (n == 1) CSN_LAUNCH(bn_finalize_kernel, dim3(a[0].C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a[0]);
  else CSN_LAUNCH(bn_finalize3_kernel, dim3(mx, n), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a[0], a[1], a[2]);
  return (int)hipGetLastError();
}
int csn_launch_bn_apply(const BnApplyArgs& a, void* stream) {
  CSN_LAUNCH_AT(a.a16, bn_apply_gap_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
// all penalty terms of one forward: jobs in forward order, `partial` holds >= njobs doubles
int csn_launch_bn_penalty(const BnPenaltyJob* jobs, int njobs, const float* arena, double* partial, double* penalty, void* stream) {
  if (njobs <= 0) return 0;
  for (int first = 0; first < njobs; first += CSN_PEN_JOBS) {   // the job table rides in the kernel arguments (< 4 KB)
    BnPenaltyArgs a;
    const int n = njobs - first < CSN_PEN_JOBS ? njobs - first : CSN_PEN_JOBS;
    for (int i = 0; i < n; ++i) a.job[i] = jobs[first + i];
    a.arena = arena; a.partial = partial; a.first = first; a.n = n;
    CSN_LAUNCH(bn_penalty_jobs_kernel, dim3(n), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  }
  CSN_LAUNCH(bn_penalty_sum_kernel, dim3(1), dim3(64), 0, stream, (const double*)partial, njobs, penalty);
  return (int)hipGetLastError();
}

static inline int grid_for(int64_t n) { return (int)((n + CSN_BLOCK - 1) / CSN_BLOCK < 4096 ? (n + CSN_BLOCK - 1) / CSN_BLOCK : 4096); }

// the three steps of csn_launch_bn_bwd one by one, so that the finalise steps of a unit's branches share a launch:
// reduce (fills a.cpp / a.nslab) -> csn_launch_bn_bwd_finalize_n -> csn_launch_bn_bwd_apply_step
int csn_launch_bn_bwd_reduce(BnBwdArgs& a, void* stream) {
  a.cpp = bn_cpp(a.S, a.C, a.HW);
  a.nslab = bn_nslab(a.S, a.cpp);
  if (a.nslab_in > 0) a.nslab = a.nslab_in;
  else CSN_LAUNCH_AT(a.a16, bn_bwd_reduce_kernel, dim3(a.nslab, a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
int csn_launch_bn_bwd_finalize_n(const BnBwdArgs* jobs, int n, void* stream) {
  if (n <= 0) return 0;
  int mx = 1;
  for (int i = 0; i < n; ++i) mx = std::max(mx, (int)jobs[i].C);
  if (n == 1) CSN_LAUNCH(bn_bwd_finalize_kernel, dim3(jobs[0].C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, jobs[0]);
  else CSN_LAUNCH(bn_bwd_finalize3_kernel, dim3(mx, n), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, jobs[0], jobs[1], jobs[n > 2 ? 2 : 0]);
  return (int)hipGetLastError();
}
int csn_launch_bn_bwd_apply_step(const BnBwdArgs& a, void* stream) {
  if (!a.skip_apply) {
    if (a.adj2 && a.dz_out && a.dz_out != a.z && csn_bn_bwd_adj2_ok(a.HW, a.W)) CSN_LAUNCH_AT(a.a16, bn_bwd_apply_adj2_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), 0, stream, a);
    else if (a.adj2) return 1;   // (the caller asks csn_bn_bwd_adj2_ok first)
    else CSN_LAUNCH_AT(a.a16, bn_bwd_apply_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), 0, stream, a);
  }
  return (int)hipGetLastError();
}
int csn_launch_bn_bwd(const BnBwdArgs& a0, void* stream) {
  BnBwdArgs a = a0;
  a.cpp = bn_cpp(a.S, a.C, a.HW);
  a.nslab = bn_nslab(a.S, a.cpp);
  if (a.nslab_in > 0) a.nslab = a.nslab_in;
  else CSN_LAUNCH_AT(a.a16, bn_bwd_reduce_kernel, dim3(a.nslab, a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  CSN_LAUNCH(bn_bwd_finalize_kernel, dim3(a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  if (!a.skip_apply) {
    if (a.adj2 && a.dz_out && a.dz_out != a.z && csn_bn_bwd_adj2_ok(a.HW, a.W)) CSN_LAUNCH_AT(a.a16, bn_bwd_apply_adj2_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), 0, stream, a);
    else if (a.adj2) return 1;   // (the caller asks csn_bn_bwd_adj2_ok first)
    else CSN_LAUNCH_AT(a.a16, bn_bwd_apply_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), 0, stream, a);
  }
  return (int)hipGetLastError();
}
int csn_launch_gap_tiles(const GapTilesArgs& a, void* stream) {
  CSN_LAUNCH(gap_tiles_kernel, dim3((a.C * a.S + CSN_BLOCK - 1) / CSN_BLOCK), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
int csn_launch_gap_tiles_batch(const GapTilesArgs* jobs, int njobs, void* stream) {
  for (int first = 0; first < njobs; first += CSN_GAP_JOBS) {
    GapTilesBatch b;
    b.n = njobs - first < CSN_GAP_JOBS ? njobs - first : CSN_GAP_JOBS; b.pad = 0;
    int mx = 1;
    for (int i = 0; i < b.n; ++i) { b.job[i] = jobs[first + i]; mx = std::max(mx, (jobs[first + i].C * jobs[first + i].S + CSN_BLOCK - 1) / CSN_BLOCK); }
    CSN_LAUNCH(gap_tiles_jobs_kernel, dim3(mx, b.n), dim3(CSN_BLOCK), 0, stream, b);
  }
  return (int)hipGetLastError();
}
int csn_launch_dw_wgrad_finalize_batch(const DwFinJob* jobs, int njobs, float* grad, void* stream) {
  for (int first = 0; first < njobs; first += CSN_DWFIN_JOBS) {
    DwFinBatch b;
    b.n = njobs - first < CSN_DWFIN_JOBS ? njobs - first : CSN_DWFIN_JOBS; b.pad = 0; b.grad = grad;
    int mx = 1;
    for (int i = 0; i < b.n; ++i) { b.job[i] = jobs[first + i]; mx = std::max(mx, (int)jobs[first + i].C); }
    CSN_LAUNCH(dw_wgrad_finalize_jobs_kernel, dim3(mx, b.n), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, b);
  }
  return (int)hipGetLastError();
}
int csn_launch_bn_bwd_apply(const BnBwdArgs& a, void* stream) {
  CSN_LAUNCH_AT(a.a16, bn_bwd_apply_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
// a0.nslab > 0 on entry: the partials were already written (by dw3x3_bwd_kernel, that many per channel): finalise only
int csn_launch_dw_wgrad(const DwWgradArgs& a0, void* stream) {
  DwWgradArgs a = a0;
  if (a.nslab <= 0) {
    a.cpp = bn_cpp(a.S, a.C, (int64_t)a.H * a.W);
    a.nslab = bn_nslab(a.S, a.cpp);
    CSN_LAUNCH_AT(a.a16, dw_wgrad_kernel, dim3(a.nslab, a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  }
  CSN_LAUNCH(dw_wgrad_finalize_kernel, dim3(a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
// in16 / out16: element type of the source / destination (the logits gradient arrives as float whatever the mode)
int csn_launch_adjup(const AdjUpArgs& a, void* stream) {
  const bool rows4_off = std::getenv("CSN_ADJ4_ROWS") && std::getenv("CSN_ADJ4_ROWS")[0] == '0';
  if (a.f == 4 && !rows4_off && ((a.Wl * 4) & 7) == 0 && (a.Wl * 4) / 8 <= CSN_BLOCK) {
    const dim3 grid(a.planes), block(CSN_BLOCK);
    if (a.in16 && a.out16) CSN_LAUNCH((adjup4_rows_kernel<csn_bf16, csn_bf16>), grid, block, 0, stream, a);
    else if (a.out16) CSN_LAUNCH((adjup4_rows_kernel<float, csn_bf16>), grid, block, 0, stream, a);
    else if (!a.in16) CSN_LAUNCH((adjup4_rows_kernel<float, float>), grid, block, 0, stream, a);
    if (!(a.in16 && !a.out16)) return (int)hipGetLastError();
  }
  const bool pair = a.f == 2 && (a.Wl & 1) == 0;
  const dim3 grid(grid_for((int64_t)a.planes * a.Hl * (pair ? (a.Wl >> 1) : a.Wl))), block(CSN_BLOCK);
  if (pair) {
    if (a.in16 && a.out16) CSN_LAUNCH((adjup2_pair_kernel<csn_bf16, csn_bf16>), grid, block, 0, stream, a);
    else if (a.out16) CSN_LAUNCH((adjup2_pair_kernel<float, csn_bf16>), grid, block, 0, stream, a);
    else CSN_LAUNCH((adjup2_pair_kernel<float, float>), grid, block, 0, stream, a);
  } else {
    if (a.in16 && a.out16) CSN_LAUNCH((adjup_kernel<csn_bf16, csn_bf16>), grid, block, 0, stream, a);
    else if (a.out16) CSN_LAUNCH((adjup_kernel<float, csn_bf16>), grid, block, 0, stream, a);
    else CSN_LAUNCH((adjup_kernel<float, float>), grid, block, 0, stream, a);
  }
  return (int)hipGetLastError();
}
// Wl % 2 == 0: a thread owns two neighbouring low-resolution pixels of a row = a 2 x 4 block of dx -- one pair load, two
// four-element stores (the one-element-per-thread form above: 2-byte stores in the bf16 mode, 0.8 ms per step for four launches)
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void avgpool2_bwd_pair_kernel(PoolBwdArgs a) {
  const int Hl = a.Hl, Wl = a.Wl, Wp = Wl >> 1, Wh = Wl * 2;
  const int64_t tot = (int64_t)a.planes * Hl * Wp;
  for (int64_t e = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; e < tot; e += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t pl = e / ((int64_t)Hl * Wp);
    const int r = (int)(e - pl * (int64_t)Hl * Wp);
    const int yl = r / Wp, xl = (r - yl * Wp) * 2;
    const float2 t = act_ld2(act_cast<AT>(a.t) + pl * (int64_t)Hl * Wl + (int64_t)yl * Wl + xl);
    const float4 v = make_float4(0.25f * t.x, 0.25f * t.x, 0.25f * t.y, 0.25f * t.y);
    AT* o = act_cast<AT>(a.dx) + pl * (int64_t)Hl * 2 * Wh + (int64_t)(2 * yl) * Wh + 2 * xl;
    act_st4(o, v);
    act_st4(o + Wh, v);
  }
}

int csn_launch_avgpool2_bwd(const PoolBwdArgs& a, void* stream) {
  if ((a.Wl & 1) == 0)
    CSN_LAUNCH_AT(a.a16, avgpool2_bwd_pair_kernel, dim3(grid_for((int64_t)a.planes * a.Hl * (a.Wl >> 1))), dim3(CSN_BLOCK), 0, stream, a);
  else
    CSN_LAUNCH_AT(a.a16, avgpool2_bwd_kernel, dim3(grid_for((int64_t)a.planes * a.Hl * a.Wl * 4)), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
// f = 2, even low-resolution width: one thread routes TWO neighbouring windows -- the 2 x 4 block of x and of dx as aligned
// four-element vectors (read-modify-write of dx), the two gradients as a pair.  Same first-maximum rule.
template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK) void maxpool2_bwd_add_pair_kernel(PoolBwdArgs a) {
  const int Hl = a.Hl, Wl = a.Wl, Wp = Wl >> 1;
  const int Wh = Wl * 2;
  const int64_t hwh = (int64_t)Hl * 2 * Wh;
  const int64_t tot = (int64_t)a.planes * Hl * Wp;
  for (int64_t e = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; e < tot; e += (int64_t)gridDim.x * CSN_BLOCK) {
    const int64_t pl = e / (Hl * Wp);
    const int r = (int)(e - pl * Hl * Wp);
    const int yl = r / Wp, xl = (r - yl * Wp) * 2;
    const int64_t o = pl * hwh + (int64_t)(yl * 2) * Wh + xl * 2;
    const float4 x0 = act_ld4(act_cast<AT>(a.x) + o), x1 = act_ld4(act_cast<AT>(a.x) + o + Wh);
    const float2 t = act_ld2(act_cast<AT>(a.t) + pl * (int64_t)Hl * Wl + (int64_t)yl * Wl + xl);
    float4 d0 = act_ld4(act_cast<AT>(a.dx) + o), d1 = act_ld4(act_cast<AT>(a.dx) + o + Wh);
    {   // window 0: (x0.x, x0.y; x1.x, x1.y), scan order row-major, `v > best || v != v`
      float best = x0.x; int bi = 0;
      if (x0.y > best || x0.y != x0.y) { best = x0.y; bi = 1; }
      if (x1.x > best || x1.x != x1.x) { best = x1.x; bi = 2; }
      if (x1.y > best || x1.y != x1.y) { best = x1.y; bi = 3; }
      if (bi == 0) d0.x += t.x; else if (bi == 1) d0.y += t.x; else if (bi == 2) d1.x += t.x; else d1.y += t.x;
    }
    {   // window 1: (x0.z, x0.w; x1.z, x1.w)
      float best = x0.z; int bi = 0;
      if (x0.w > best || x0.w != x0.w) { best = x0.w; bi = 1; }
      if (x1.z > best || x1.z != x1.z) { best = x1.z; bi = 2; }
      if (x1.w > best || x1.w != x1.w) { best = x1.w; bi = 3; }
      if (bi == 0) d0.z += t.y; else if (bi == 1) d0.w += t.y; else if (bi == 2) d1.z += t.y; else d1.w += t.y;
    }
    act_st4(act_cast<AT>(a.dx) + o, d0);
    act_st4(act_cast<AT>(a.dx) + o + Wh, d1);
  }
}

int csn_launch_maxpool_bwd_add(const PoolBwdArgs& a, void* stream) {
  if (a.f == 2 && (a.Wl & 1) == 0) {
    CSN_LAUNCH_AT(a.a16, maxpool2_bwd_add_pair_kernel, dim3(grid_for((int64_t)a.planes * a.Hl * (a.Wl >> 1))), dim3(CSN_BLOCK), 0,
                  stream, a);
    return (int)hipGetLastError();
  }
  CSN_LAUNCH_AT(a.a16, maxpool_bwd_add_kernel, dim3(grid_for((int64_t)a.planes * a.Hl * a.Wl)), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
int csn_launch_sum_to_grad(const float* in, int64_t n, float* dst, double* partial, int a16, void* stream) {
  const int nblk = grid_for(n) < 512 ? grid_for(n) : 512;
  CSN_LAUNCH_AT(a16, sum_partial_kernel, dim3(nblk), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, in, n, partial);
  CSN_LAUNCH(sum_final_kernel, dim3(1), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, partial, nblk, dst);
  return (int)hipGetLastError();
}
// x (float, the caller's batch) -> the bf16 copy the train-mode kernels read
__global__ __launch_bounds__(CSN_BLOCK) void to_bf16_kernel(const float* __restrict__ in, csn_bf16* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * CSN_BLOCK)
    act_st4(out + 4 * i, act_ld4(in + 4 * i));
}
int csn_launch_to_bf16(const float* in, void* out, int64_t n, void* stream) {   // n: multiple of 4
  CSN_LAUNCH(to_bf16_kernel, dim3(grid_for(n >> 2)), dim3(CSN_BLOCK), 0, stream, in, reinterpret_cast<csn_bf16*>(out), n >> 2);
  return (int)hipGetLastError();
}
int csn_launch_bce(const float* y, const float* t, float* dy, int64_t n, double* loss, void* stream) {
  const int nblk = grid_for(n) < BCE_MAX_BLOCKS ? grid_for(n) : BCE_MAX_BLOCKS;
  CSN_LAUNCH(bce_logits_kernel, dim3(nblk), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, y, t, dy, n);
  CSN_LAUNCH(bce_final_kernel, dim3(1), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, nblk, n, loss);
  return (int)hipGetLastError();
}
int csn_launch_adam(const AdamArgs& a, void* stream) {
  CSN_LAUNCH(adam_kernel, dim3(grid_for(a.n)), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
