// ld16_probe.hip -- how fast are 16-bit buffer loads next to 32-bit ones on gfx950?  Each wave streams rows of 64 pixels
// of NCH channel planes (the gather pattern of goct_pw / goct_wgrad): (a) one dword per lane, (b) one ushort per lane,
// (c) bf16 pairs: half the lanes load a dword = two pixels, the other half the next channel.
// build: hipcc --offload-arch=gfx950 -O3 -o ld16_probe ld16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t buf;
#define NCH 32
template <int MODE>
__global__ __launch_bounds__(256) void k(const void* p, float* out, int hw, int ngroups) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  float acc = 0.f;
  for (int g = wave; g < ngroups; g += nw) {
    const int img = g / (hw / 64), pg = g % (hw / 64);
    if (MODE == 0) {
      buf b = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p + (size_t)img * NCH * hw), 0, 0xffffffff, 0x00020000);
      float v[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (pg * 64 + lane) * 4, c * hw * 4, 0));
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc += v[c];
    } else if (MODE == 1) {
      buf b = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p + (size_t)img * NCH * hw), 0, 0xffffffff, 0x00020000);
      float v[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[c] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(b, (pg * 64 + lane) * 2, c * hw * 2, 0) << 16);
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc += v[c];
    } else {
      buf b = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p + (size_t)img * NCH * hw), 0, 0xffffffff, 0x00020000);
      unsigned v[NCH / 2];
      const unsigned lo = (pg * 64 + (lane & 31) * 2) * 2 + (lane >> 5) * hw * 2;
#pragma unroll
      for (int c = 0; c < NCH / 2; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b32(b, lo, 2 * c * hw * 2, 0);
#pragma unroll
      for (int c = 0; c < NCH / 2; ++c) acc += __uint_as_float(v[c] << 16) + __uint_as_float(v[c] & 0xffff0000u);
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const int hw = 112 * 112, B = 256;
  const size_t n = (size_t)B * NCH * hw;
  void* p; float* o;
  hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ngroups = B * (hw / 64);
  for (int mode = 0; mode < 3; ++mode) {
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, p, o, hw, ngroups);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, p, o, hw, ngroups);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, p, o, hw, ngroups);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)n * (mode == 0 ? 4 : 2);
      if (it == 2) printf("mode %d (%s): %.3f ms, %.1f GB/s, %.2f G elements/s\n", mode, mode == 0 ? "dword/lane f32" : mode == 1 ? "ushort/lane bf16" : "dword pairs bf16", ms, bytes / ms / 1e6, n / ms / 1e6);
    }
  }
  return 0;
}
