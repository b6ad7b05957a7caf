// csn_reduce.h -- fixed-order block reductions in fp64 (256 threads = four waves), shared by the train-step kernels.
// `sm` is the block's dynamic LDS, >= CSN_BLOCK doubles.
#pragma once
#include "csn_device.h"

__device__ __forceinline__ double bn_block_sum(double v, double* sm) {
  const int tid = threadIdx.x;
#ifdef CSN_EMU_SEQ
  // same summation tree as the device version (butterfly per 64 lanes, then the four wave sums), but evaluated by
  // ONE fiber between two barriers: a barrier costs the emulator 256 context switches
  sm[tid] = v;
  __syncthreads();
  if (tid == 0) {
    double w[4];
    for (int q = 0; q < 4; ++q) {
      double t[64];
      for (int l = 0; l < 64; ++l) t[l] = sm[64 * q + l];
      for (int o = 32; o > 0; o >>= 1)
        for (int l = 0; l < o; ++l) t[l] = t[l] + t[l + o];   // lane l of the butterfly ends with the same pairing tree
      w[q] = t[0];
    }
    sm[0] = (w[0] + w[1]) + (w[2] + w[3]);
  }
  __syncthreads();
  const double r = sm[0];
  __syncthreads();
  return r;
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += csn_shfl_xor(v, o);   // butterfly inside the wave, fixed order
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  __syncthreads();
  const double r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
#endif
}

// N sums at once: the same summation tree per value as bn_block_sum, one pair of barriers for all of them
template <int N>
__device__ __forceinline__ void bn_block_sum_n(double (&v)[N], double* sm) {
#ifdef CSN_EMU_SEQ
  for (int t = 0; t < N; ++t) v[t] = bn_block_sum(v[t], sm);
#else
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < N; ++t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[t] += csn_shfl_xor(v[t], o);
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int t = 0; t < N; ++t) sm[(tid >> 6) * N + t] = v[t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < N; ++t) v[t] = (sm[t] + sm[N + t]) + (sm[2 * N + t] + sm[3 * N + t]);
  __syncthreads();
#endif
}

