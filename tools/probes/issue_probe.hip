// issue_probe.hip -- does vector work hide in the shadow of a multi-pass MFMA on gfx950?  mfma4_probe.hip (round 3) found that a
// v_mfma_f32_4x4x1 (2 passes) costs 8 issue cycles and every v_fma its own 4 ON TOP -- the budget pw4_kernel runs at.  Here the same
// question for v_mfma_f32_16x16x4_f32 (8 passes, 4x the MACs) and v_mfma_f32_32x32x2_f32 (16 passes), and for the bf16 forms
// v_mfma_f32_4x4x4_16B_bf16 (2 passes) / v_mfma_f32_16x16x16_bf16: NV v_fma per MFMA, WPS waves per SIMD, one block per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
#define ITER 1500
#define NACC 8
// KIND 0: 4x4x1 f32, 1: 16x16x4 f32, 2: 32x32x2 f32, 3: 4x4x4 bf16, 4: 16x16x16 bf16
template <int KIND, int NV>
__global__ __launch_bounds__(1024) void k(float* out, float a0, float b0) {
  f4 acc[NACC];
  f16v big[KIND == 2 ? 4 : 1];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (KIND == 2 ? 4 : 1); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  float v[6] = {a0, b0, a0 + 1.f, b0 + 2.f, a0 + 3.f, b0 + 4.f};
  s4 sa = {(short)threadIdx.x, 1, 2, 3}, sb = {4, 5, 6, (short)threadIdx.x};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      if (KIND == 2) big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i & 3], 0, 0, 0);
      if (KIND == 3) acc[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, acc[i], 0, 0, 0);
      if (KIND == 4) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 6]) : "v"(v[(j + 1) % 6]), "v"(b));   // (not packable)
    }
  }
  float s = v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (KIND == 2 ? 4 : 1); ++i) s += big[i][0] + big[i][7];
  if (s == 12345.6789f) out[0] = s;
}
template <int KIND, int NV>
void run(int wps, const char* what) {
  float* o;
  hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(256 * wps), 0, 0, o, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = (double)ITER * NACC;
  printf("%-26s + %2d v_fma per MFMA, %d waves/SIMD: %.3f ms -> %.2f ns per MFMA(+VALU) per SIMD\n", what, NV, wps, ms, ms * 1e6 / n / wps);
  hipFree(o);
}
#define SWEEP(KIND, WHAT)                                                                                         \
  run<KIND, 0>(1, WHAT); run<KIND, 0>(2, WHAT); run<KIND, 2>(1, WHAT); run<KIND, 2>(2, WHAT); run<KIND, 4>(2, WHAT); \
  run<KIND, 8>(2, WHAT); run<KIND, 16>(2, WHAT);
int main() {
  SWEEP(0, "v_mfma_f32_4x4x1_16B_f32")
  SWEEP(1, "v_mfma_f32_16x16x4_f32")
  SWEEP(2, "v_mfma_f32_32x32x2_f32")
  SWEEP(3, "v_mfma_f32_4x4x4_16B_bf16")
  SWEEP(4, "v_mfma_f32_16x16x16_bf16")
  return 0;
}
