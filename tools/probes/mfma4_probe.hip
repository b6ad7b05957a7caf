// mfma4_probe.hip -- v_mfma_f32_4x4x1_16B_f32 on gfx950: issue interval, dependent-accumulator latency, and what a
// stream of VALU / ds_read_b128 instructions between the MFMAs costs.  One block of WPS*4 waves per CU (WPS waves per
// SIMD), every wave runs ITER iterations of an unrolled body; cycles per MFMA = clocks / (ITER * MFMAs per iteration).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma4_probe mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITER 2000
// MODE 0: NACC independent accumulators, back to back.  MODE 1: + one v_fma per MFMA.  MODE 2: + one ds_read_b128 per 4 MFMAs.
// MODE 3: + two v_fma per MFMA.
template <int NACC, int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* clk, float a0, float b0) {
  __shared__ f4 lds[256];
  lds[threadIdx.x & 255] = f4{a0, a0, a0, a0};
  __syncthreads();
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0, v0 = a0, v1 = b0, v2 = a0 + 1.f, v3 = b0 + 2.f;
  f4 w = lds[threadIdx.x & 3];
  const long long t0 = clock64();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 2 && (i & 3) == 0) w = lds[(threadIdx.x + i + it) & 3];
      acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(MODE == 2 ? w[i & 3] : a, b, acc[i], 0, 0, 0);
      if (MODE == 1 || MODE == 3) v0 = __builtin_fmaf(v0, v1, v2);
      if (MODE == 3) v3 = __builtin_fmaf(v3, v1, v2);
    }
  }
  const long long t1 = clock64();
  float s = v0 + v3;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.6789f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NACC, int MODE>
void run(int wps, const char* what) {
  float* o; long long* c;
  hipMalloc(&o, 4); hipMalloc(&c, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(256 * wps), 0, 0, o, c, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long cl; hipMemcpy(&cl, c, 8, hipMemcpyDeviceToHost);
  const double n = (double)ITER * NACC;
  // clock64 = s_memtime at 100 MHz constant; use wall time with a nominal 2.4 GHz to convert
  printf("%-44s acc %2d waves/SIMD %d: %.3f ms  -> %.2f ns per MFMA per wave, %.2f ns per MFMA per SIMD (memtime ticks %lld)\n", what, NACC, wps, ms,
         ms * 1e6 / n, ms * 1e6 / n / wps, cl);
  hipFree(o); hipFree(c);
}
int main() {
  run<1, 0>(1, "dependent chain");
  run<2, 0>(1, "2 accumulators");
  run<4, 0>(1, "4 accumulators");
  run<16, 0>(1, "16 accumulators");
  run<16, 0>(2, "16 accumulators");
  run<16, 0>(4, "16 accumulators");
  run<16, 1>(1, "16 acc + 1 v_fma per MFMA");
  run<16, 1>(2, "16 acc + 1 v_fma per MFMA");
  run<16, 3>(1, "16 acc + 2 v_fma per MFMA");
  run<16, 3>(2, "16 acc + 2 v_fma per MFMA");
  run<16, 2>(1, "16 acc + ds_read_b128 per 4 MFMA (A operand)");
  run<16, 2>(2, "16 acc + ds_read_b128 per 4 MFMA (A operand)");
  run<16, 2>(3, "16 acc + ds_read_b128 per 4 MFMA (A operand)");
  return 0;
}
