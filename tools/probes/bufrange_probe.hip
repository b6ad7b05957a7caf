// bufrange_probe.hip -- how is a multi-dword raw buffer load range-checked on gfx950?  (k_c3q.hip loads a window row of four columns
// starting ONE column left of the lane's pair with one 128-bit load: at the plane's first pixel that is byte offset -4.)
//   case A: voffset = -4 (0xfffffffc), soffset 0        -> per-dword check with 32-bit wrap would give (0, d0, d1, d2)
//   case B: voffset = 4 N - 8 (two dwords before the end) -> per-dword check gives (d[N-2], d[N-1], 0, 0)
//   case C: voffset = -4, soffset = 64                   -> the sum is in range: (d15, d16, d17, d18)
// build: hipcc --offload-arch=gfx950 -O3 -o bufrange_probe bufrange_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, unsigned nbytes, float* out) {
  __amdgpu_buffer_rsrc_t b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, nbytes, 0x00020000);
  const unsigned off[3] = {0xfffffffcu, nbytes - 8u, 0xfffffffcu};
  const unsigned so[3] = {0u, 0u, 64u};
  for (int c = 0; c < 3; ++c) {
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(b, off[c], so[c], 0);
    out[4 * c] = __uint_as_float(v.x); out[4 * c + 1] = __uint_as_float(v.y); out[4 * c + 2] = __uint_as_float(v.z); out[4 * c + 3] = __uint_as_float(v.w);
  }
  typedef unsigned u3 __attribute__((ext_vector_type(3)));
  const u3 w = __builtin_amdgcn_raw_buffer_load_b96(b, 0xfffffffcu, 0u, 0);
  out[12] = __uint_as_float(w.x); out[13] = __uint_as_float(w.y); out[14] = __uint_as_float(w.z);
}
int main() {
  const int N = 64;
  float h[N], *d, *o, r[16];
  for (int i = 0; i < N; ++i) h[i] = 100.f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, (unsigned)sizeof(h), o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("data d[i] = 100 + i, %d dwords\n", N);
  printf("A: voffset -4           : %g %g %g %g   (per dword + wrap: 0 100 101 102)\n", r[0], r[1], r[2], r[3]);
  printf("B: voffset 4N - 8       : %g %g %g %g   (per dword: 162 163 0 0)\n", r[4], r[5], r[6], r[7]);
  printf("C: voffset -4, soffset 64: %g %g %g %g   (sum in range: 115 116 117 118)\n", r[8], r[9], r[10], r[11]);
  printf("D: 96-bit load, voffset -4: %g %g %g   (per dword + wrap: 0 100 101)\n", r[12], r[13], r[14]);
  return 0;
}
