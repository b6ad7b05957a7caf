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
#include "csn_kernels.h"

#define BN_NSLAB 32   // slabs per channel for the statistics partials

__device__ __forceinline__ double bn_block_sum(double v, double* sm) {
  const int tid = threadIdx.x;
  sm[tid] = v;
  __syncthreads();
  for (int s = CSN_BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) sm[tid] += sm[tid + s];
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}

// grid (BN_NSLAB, C): block (slab, c) reduces its share of the S*HW elements of channel c.
__global__ __launch_bounds__(CSN_BLOCK) void bn_stats_kernel(BnStatsArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int c = blockIdx.y, slab = blockIdx.x;
  const int64_t hw = a.HW;
  const int64_t per = ((int64_t)a.S * hw + BN_NSLAB - 1) / BN_NSLAB;
  const int64_t beg = (int64_t)slab * per;
  const int64_t end = min(beg + per, (int64_t)a.S * hw);
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = beg + threadIdx.x; i < end; i += CSN_BLOCK) {
    const int64_t n = i / hw, p = i - n * hw;
    const double v = (double)a.z[(n * a.C + c) * hw + p];
    s1 += v;
    s2 += v * v;
  }
  s1 = bn_block_sum(s1, sm);
  s2 = bn_block_sum(s2, sm);
  if (threadIdx.x == 0) {
    a.partial[((int64_t)c * BN_NSLAB + slab) * 2 + 0] = s1;
    a.partial[((int64_t)c * BN_NSLAB + slab) * 2 + 1] = s2;
  }
}

// one thread per channel
__global__ __launch_bounds__(CSN_BLOCK) void bn_finalize_kernel(BnFinalizeArgs a) {
  const int c = blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (c >= a.C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < BN_NSLAB; ++k) {
    s1 += a.partial[((int64_t)c * BN_NSLAB + k) * 2 + 0];
    s2 += a.partial[((int64_t)c * BN_NSLAB + k) * 2 + 1];
  }
  const double n = (double)a.count;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float gamma = a.arena[a.off_weight + c], beta = a.arena[a.off_bias + c];
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  const float sc = gamma * invstd;
  a.scale[c] = sc;
  a.shift[c] = beta - (float)mean * sc;
  const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
  a.arena[a.off_rmean + c] = 0.9f * a.arena[a.off_rmean + c] + 0.1f * (float)mean;
  a.arena[a.off_rvar + c] = 0.9f * a.arena[a.off_rvar + c] + 0.1f * (float)unbiased;
}

// grid (C, S): block (c, n) streams one plane: y = PReLU(z*scale + shift) in place, and the plane sum
// feeds the penalty 0.5 * w * |mean_hw y| * gamma^2 (fp64 atomic; w == 0: unit is not hooked).
__global__ __launch_bounds__(CSN_BLOCK) void bn_apply_gap_kernel(BnApplyArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int c = blockIdx.x, n = blockIdx.y;
  const int64_t hw = a.HW;
  float* __restrict__ p = a.z + ((int64_t)n * a.C + c) * hw;
  const float sc = a.scale[c], sh = a.shift[c], al = a.alpha[c];
  double s = 0.0;
  if ((hw & 3) == 0) {
    float4* p4 = reinterpret_cast<float4*>(p);
    for (int64_t i = threadIdx.x; i < (hw >> 2); i += CSN_BLOCK) {
      float4 v = p4[i];
      v.x = csn_epi(v.x, sc, sh, al); v.y = csn_epi(v.y, sc, sh, al);
      v.z = csn_epi(v.z, sc, sh, al); v.w = csn_epi(v.w, sc, sh, al);
      p4[i] = v;
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    }
  } else {
    for (int64_t i = threadIdx.x; i < hw; i += CSN_BLOCK) {
      const float v = csn_epi(p[i], sc, sh, al);
      p[i] = v;
      s += (double)v;
    }
  }
  if (a.flop_w != 0.f) {
    s = bn_block_sum(s, sm);
    if (threadIdx.x == 0) {
      const double g = (double)a.arena[a.off_weight + c];
      const double term = 0.5 * (double)a.flop_w * fabs(s / (double)hw) * g * g;
#ifdef CSN_CPU_EMU
#pragma omp atomic
      *a.penalty += term;
#else
      atomicAdd(a.penalty, term);
#endif
    }
  }
}

int csn_launch_bn_stats(const BnStatsArgs& a, void* stream) {
  CSN_LAUNCH(bn_stats_kernel, dim3(BN_NSLAB, a.C), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
int csn_launch_bn_finalize(const BnFinalizeArgs& a, void* stream) {
  CSN_LAUNCH(bn_finalize_kernel, dim3((a.C + CSN_BLOCK - 1) / CSN_BLOCK), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
int csn_launch_bn_apply(const BnApplyArgs& a, void* stream) {
  CSN_LAUNCH(bn_apply_gap_kernel, dim3(a.C, a.S), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
