// mfma4_exact.hip -- is v_mfma_f32_4x4x1_16B_f32 the exact fp32 FMA chain (round to nearest even, subnormals kept) that
// v_mfma_f32_16x16x4_f32 is?  Lane l (block b = l / 4, column j = l % 4) accumulates acc[i] += A[i][k] * B[k][l] over K steps
// with the MFMA and with fmaf; the two must agree bit for bit.  build: hipcc --offload-arch=gfx950 -O3 -o mfma4_exact mfma4_exact.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define K 459
__global__ void k(const float* A /* [K][4] */, const float* B /* [K][64] */, float* out_mfma /* [64][4] */, float* out_fma) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float ref[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < K; ++kk) {
    const float a = A[kk * 4 + (l & 3)], b = B[kk * 64 + l];
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) ref[i] = __builtin_fmaf(A[kk * 4 + i], b, ref[i]);
  }
  for (int i = 0; i < 4; ++i) { out_mfma[l * 4 + i] = acc[i]; out_fma[l * 4 + i] = ref[i]; }
}
int main() {
  std::vector<float> A(K * 4), B(K * 64);
  srand(1);
  for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * ((rand() & 7) == 0 ? 1e-3f : 1.f);
  for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * ((rand() & 15) == 0 ? 1e4f : 1.f);
  float *dA, *dB, *o1, *o2;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, o1, o2);
  float h1[256], h2[256];
  hipMemcpy(h1, o1, 1024, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 1024, hipMemcpyDeviceToHost);
  int diff = 0; double maxrel = 0;
  for (int i = 0; i < 256; ++i) {
    if (memcmp(&h1[i], &h2[i], 4)) { ++diff; double r = fabs((double)h1[i] - h2[i]) / (fabs((double)h2[i]) + 1e-30); if (r > maxrel) maxrel = r; }
  }
  printf("mfma4_exact: K = %d, %d of 256 results differ from the fmaf chain, max relative difference %.3g\n", K, diff, maxrel);
  return 0;
}
