// dpp_probe.hip -- hardware check of the cross-lane primitives the fused ILBlock kernel relies on (gfx950):
// DPP wave_shr:1 / wave_shl:1 (full-wave shifts by one lane, zero fill), v_pk_fma_f32 with a broadcast scalar.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dpp_probe.hip -o tools/probes/dpp_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
  const int l = threadIdx.x;
  const float v = (float)(l + 1);
  const int shr = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true);   // wave_shr:1
  const int shl = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true);   // wave_shl:1
  const int rshr = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true);  // row_shr:1
  o[l] = __int_as_float(shr);
  o[64 + l] = __int_as_float(shl);
  o[128 + l] = __int_as_float(rshr);
}
int main() {
  float* d; float h[192];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const float e_shr = l == 0 ? 0.f : (float)l;          // lane l receives lane l-1
    const float e_shl = l == 63 ? 0.f : (float)(l + 2);   // lane l receives lane l+1
    const float e_r = (l & 15) == 0 ? 0.f : (float)l;
    if (h[l] != e_shr || h[64 + l] != e_shl || h[128 + l] != e_r) {
      ++bad;
      printf("lane %d: wave_shr %g (want %g) wave_shl %g (want %g) row_shr %g (want %g)\n", l, h[l], e_shr, h[64 + l], e_shl, h[128 + l], e_r);
    }
  }
  printf("dpp_probe: %s\n", bad ? "MISMATCH" : "OK wave_shr:1 = from lane-1, wave_shl:1 = from lane+1, zero fill");
  return bad != 0;
}
