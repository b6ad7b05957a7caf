// dpp_probe.hip -- which lane does a wave_shr:1 / wave_shl:1 DPP move read on gfx950?  (k_misc.hip csn_from_lane_below / _above take the
// depthwise kernels' halo columns from the neighbouring lanes with them.)  Expected: below[i] = i - 1 (lane 0: 0 by bound_ctrl),
// above[i] = i + 1 (lane 63: 0); with the upper half of the wave switched off, lane 31's "above" must not be lane 32's stale value
// used for anything (it is masked by the kernels) -- printed for the record.
// build: hipcc --offload-arch=gfx950 -O3 -o dpp_probe dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o, int half) {
  const unsigned v = 100 + threadIdx.x;
  unsigned b = 7777, a = 7777;
  if (!half || threadIdx.x < 32) {
    b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
    a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
  }
  o[threadIdx.x] = b; o[64 + threadIdx.x] = a;
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  for (int half = 0; half < 2; ++half) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, half);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int okb = 0, oka = 0;
    const int n = half ? 32 : 64;
    for (int i = 1; i < n; ++i) okb += h[i] == 100u + i - 1;
    for (int i = 0; i < n - 1; ++i) oka += h[64 + i] == 100u + i + 1;
    printf("%s: from_lane_below correct on %d of %d lanes (lane 0 reads %u); from_lane_above correct on %d of %d (lane %d reads %u)\n",
           half ? "lower half of the wave active" : "whole wave active", okb, n - 1, h[0], oka, n - 1, n - 1, h[64 + n - 1]);
  }
  return 0;
}
