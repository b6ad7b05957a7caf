// lane_probe.hip -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_lane_ops.py).  One wave executes ONE cross-lane instruction of the
// set the csnet kernels use, operands and results per lane, so that the lane maps the CPU emulator's lane-exact mode implements
// (tests/emu/hip_cpu_shim.h: lanes_mfma_*, lanes_dpp_*, lanes_readfirstlane) are pinned to the hardware, bit for bit.
// Same interface as csn_emu_lane_probe (emu_impl.cpp).  Build: hipcc --offload-arch=gfx950 -shared -fPIC -o liblane_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../sod100k_amd/csrc/csn_device.h"   // kinds 11 / 12 run the kernels' own helpers (csn_lane_xor_f32, csn_wave_reduce_scatter8)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void lane_probe_kernel(int kind, const u4* a, const u4* b, const float* acc_in, float* acc_out) {
  const unsigned lane = threadIdx.x;
  const u4 la = a[lane], lb = b[lane];
  float d[16];
  for (int i = 0; i < 16; ++i) d[i] = acc_in[16 * lane + i];
  const bool masked = kind >= 16;
  unsigned r = la.x;
  const int k = kind & 15;
  if (k == 1) {
    f4 c = {d[0], d[1], d[2], d[3]};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(la.x), __uint_as_float(lb.x), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i] = c[i];
  } else if (k == 2) {
    f4 c = {d[0], d[1], d[2], d[3]};
    union { unsigned u[2]; s4 s; } ca, cb;
    ca.u[0] = la.x; ca.u[1] = la.y; cb.u[0] = lb.x; cb.u[1] = lb.y;
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ca.s, cb.s, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i] = c[i];
  } else if (k == 3) {
    f4 c = {d[0], d[1], d[2], d[3]};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(la.x), __uint_as_float(lb.x), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i] = c[i];
  } else if (k == 4) {
    f16 c;
    for (int i = 0; i < 16; ++i) c[i] = d[i];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, la), __builtin_bit_cast(bf8, lb), c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) d[i] = c[i];
  } else if (k == 5) {
    f4 c = {d[0], d[1], d[2], d[3]};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, la), __builtin_bit_cast(bf8, lb), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i] = c[i];
  } else if (k == 6) {
    if (!masked || lb.x) r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)la.x, 0x138, 0xf, 0xf, true);
  } else if (k == 7) {
    if (!masked || lb.x) r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)la.x, 0x130, 0xf, 0xf, true);
  } else if (k == 8) {
    if (!masked || lb.x) r = (unsigned)__builtin_amdgcn_readfirstlane((int)la.x);
  } else if (k == 10) {
    const double v = __shfl_xor(__hiloint2double((int)la.y, (int)la.x), (int)(lb.x & 63), 64);
    d[0] = __uint_as_float((unsigned)__double2loint(v)); d[1] = __uint_as_float((unsigned)__double2hiint(v));
  } else if (k == 11) {   // the float of lane (lane ^ X), X = b.x: DPP inside a row of 16, ds_bpermute across rows (round 6)
    const float v = __uint_as_float(la.x);
    const unsigned X = lb.x;   // (uniform)
    float o = v;
    if (X == 1) o = csn_lane_xor_f32<1>(v);
    else if (X == 2) o = csn_lane_xor_f32<2>(v);
    else if (X == 4) o = csn_lane_xor_f32<4>(v);
    else if (X == 8) o = csn_lane_xor_f32<8>(v);
    else if (X == 16) o = csn_lane_xor_f32<16>(v);
    else if (X == 32) o = csn_lane_xor_f32<32>(v);
    d[0] = o;
  } else if (k == 12) {   // wave sums of eight per-lane values (pw4_kernel's statistics): d[0] = total of value csn_rs8_index(lane)
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = d[i];
    d[0] = csn_wave_reduce_scatter8(v, (int)lane);
    d[1] = (float)csn_rs8_index((int)lane);
  }
  if (k >= 6 && k <= 8) d[0] = __uint_as_float(r);
  for (int i = 0; i < 16; ++i) acc_out[16 * lane + i] = d[i];
}

extern "C" int lane_probe_run(int kind, const unsigned char* a, const unsigned char* b, const float* acc_in, float* acc_out) {
  void *da = nullptr, *db = nullptr, *dc = nullptr, *dd = nullptr;
  if (hipMalloc(&da, 1024) || hipMalloc(&db, 1024) || hipMalloc(&dc, 4096) || hipMalloc(&dd, 4096)) return 1;
  int st = 0;
  st |= hipMemcpy(da, a, 1024, hipMemcpyHostToDevice);
  st |= hipMemcpy(db, b, 1024, hipMemcpyHostToDevice);
  st |= hipMemcpy(dc, acc_in, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(64), 0, 0, kind, (const u4*)da, (const u4*)db, (const float*)dc, (float*)dd);
  st |= hipDeviceSynchronize();
  st |= hipMemcpy(acc_out, dd, 4096, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
  return st;
}
