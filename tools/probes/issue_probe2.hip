// issue_probe2.hip -- do vector instructions hide in the shadow of an MFMA on gfx950?  Second version (round 5).
//
// issue_probe.hip (round 4) concluded "matrix and vector work of a SIMD are additive".  Two weaknesses: the
// compiler regrouped its loop body (five MFMAs back to back in the shipped binary), and both waves of a SIMD ran the same,
// phase-locked program.  Here
//   * the loop body is ONE asm statement: the schedule in the binary is the schedule written below
//     (per group: one MFMA followed by NV independent v_fma_f32; eight groups per trip, eight accumulators);
//   * a two-program mode: the waves of a block are split by their position on the SIMD (wave index / 4; the SIMD id of every
//     wave is read back from HW_ID and printed) into an MFMA-only program and a VALU-only program with the same instruction
//     counts as the interleaved loop; each program is timed alone and both together.
// additive  <=> t(both) ~ t(mfma only) + t(valu only);  overlapped <=> t(both) ~ max of the two.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_probe2 issue_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITER 2000

#define F(i) "v_fma_f32 %[v" #i "], %[v" #i "], %[b], %[a]\n"
// the fma's of consecutive groups rotate through eight registers (no fma depends on one of the seven before it)
#define G0_0 ""
#define G0_1 ""
#define G0_2 ""
#define G0_3 ""
#define G1_0 F(0)
#define G1_1 F(1)
#define G1_2 F(2)
#define G1_3 F(3)
#define G1_4 F(4)
#define G1_5 F(5)
#define G1_6 F(6)
#define G1_7 F(7)
#define G2_0 F(0) F(1)
#define G2_1 F(2) F(3)
#define G2_2 F(4) F(5)
#define G2_3 F(6) F(7)
#define G4_0 F(0) F(1) F(2) F(3)
#define G4_1 F(4) F(5) F(6) F(7)
#define G8_0 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define MF4(acc) "v_mfma_f32_4x4x1_16b_f32 %[" #acc "], %[a], %[b], %[" #acc "]\n"
#define MF16(acc) "v_mfma_f32_16x16x4_f32 %[" #acc "], %[a], %[b], %[" #acc "]\n"
#define NOMF(acc) ""
#define BODY0(MF) MF(c0) MF(c1) MF(c2) MF(c3) MF(c4) MF(c5) MF(c6) MF(c7)
#define BODY1(MF) MF(c0) G1_0 MF(c1) G1_1 MF(c2) G1_2 MF(c3) G1_3 MF(c4) G1_4 MF(c5) G1_5 MF(c6) G1_6 MF(c7) G1_7
#define BODY2(MF) MF(c0) G2_0 MF(c1) G2_1 MF(c2) G2_2 MF(c3) G2_3 MF(c4) G2_0 MF(c5) G2_1 MF(c6) G2_2 MF(c7) G2_3
#define BODY4(MF) MF(c0) G4_0 MF(c1) G4_1 MF(c2) G4_0 MF(c3) G4_1 MF(c4) G4_0 MF(c5) G4_1 MF(c6) G4_0 MF(c7) G4_1
#define BODY8(MF) MF(c0) G8_0 MF(c1) G8_0 MF(c2) G8_0 MF(c3) G8_0 MF(c4) G8_0 MF(c5) G8_0 MF(c6) G8_0 MF(c7) G8_0
#define OPERANDS                                                                                                          \
  : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]),  \
    [c7] "+v"(c[7]), [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), \
    [v6] "+v"(v[6]), [v7] "+v"(v[7])                                                                                       \
  : [a] "v"(a), [b] "v"(b)

// ROLE: 0 = interleaved (every wave: MFMA + NV fma per group), 1 = MFMA groups only, 2 = the fma groups only
template <int KIND, int NV, int ROLE>
__device__ __forceinline__ void loop(f4 (&c)[8], float (&v)[8], float a, float b) {
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {
      if (ROLE == 0 && NV == 0) asm volatile(BODY0(MF4) OPERANDS);
      if (ROLE == 0 && NV == 1) asm volatile(BODY1(MF4) OPERANDS);
      if (ROLE == 0 && NV == 2) asm volatile(BODY2(MF4) OPERANDS);
      if (ROLE == 0 && NV == 4) asm volatile(BODY4(MF4) OPERANDS);
      if (ROLE == 0 && NV == 8) asm volatile(BODY8(MF4) OPERANDS);
      if (ROLE == 1) asm volatile(BODY0(MF4) OPERANDS);
    } else {
      if (ROLE == 0 && NV == 0) asm volatile(BODY0(MF16) OPERANDS);
      if (ROLE == 0 && NV == 1) asm volatile(BODY1(MF16) OPERANDS);
      if (ROLE == 0 && NV == 2) asm volatile(BODY2(MF16) OPERANDS);
      if (ROLE == 0 && NV == 4) asm volatile(BODY4(MF16) OPERANDS);
      if (ROLE == 0 && NV == 8) asm volatile(BODY8(MF16) OPERANDS);
      if (ROLE == 1) asm volatile(BODY0(MF16) OPERANDS);
    }
    if (ROLE == 2 && NV == 1) asm volatile(BODY1(NOMF) OPERANDS);
    if (ROLE == 2 && NV == 2) asm volatile(BODY2(NOMF) OPERANDS);
    if (ROLE == 2 && NV == 4) asm volatile(BODY4(NOMF) OPERANDS);
    if (ROLE == 2 && NV == 8) asm volatile(BODY8(NOMF) OPERANDS);
  }
}

// mode 0: every wave runs the interleaved loop;  mode 1: waves of the first half (wave < nw / 2) run the MFMA program, the
// second half the VALU program;  mode 2: only the MFMA half works (the other exits);  mode 3: only the VALU half works
template <int KIND, int NV>
__global__ __launch_bounds__(1024) void k(float* out, unsigned* hwid, float a0, float b0, int mode) {
  f4 c[8];
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { c[i] = f4{0.f, 0.f, 0.f, 0.f}; v[i] = a0 + i; }
  const float a = a0 + threadIdx.x, b = b0;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    hwid[wave] = id;
  }
  if (mode == 0) loop<KIND, NV, 0>(c, v, a, b);
  else if (wave < nw / 2) { if (mode == 1 || mode == 2) loop<KIND, NV, 1>(c, v, a, b); }
  else { if (mode == 1 || mode == 3) loop<KIND, NV, 2>(c, v, a, b); }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (s == 12345.6789f) out[0] = s;
}

template <int KIND, int NV>
float run(int wps, int mode, unsigned* hw_host) {
  float* o; unsigned* hw;
  hipMalloc(&o, 4); hipMalloc(&hw, 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(256 * wps), 0, 0, o, hw, 1.0f, 0.5f, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  if (hw_host) hipMemcpy(hw_host, hw, 64 * 4, hipMemcpyDeviceToHost);
  hipFree(o); hipFree(hw);
  return best;
}

template <int KIND, int NV>
void sweep(const char* what) {
  const double n = (double)ITER * 8;
  for (int wps = 1; wps <= 2; ++wps) {
    const float t = run<KIND, NV>(wps, 0, nullptr);
    printf("%-26s interleaved, %d v_fma per MFMA, %d waves/SIMD: %.3f ms = %.2f ns per group per SIMD\n", what, NV, wps, t, t * 1e6 / n / wps);
  }
  if (NV > 0) {
    unsigned hw[64];
    const float tb = run<KIND, NV>(2, 1, hw), tm = run<KIND, NV>(2, 2, nullptr), tv = run<KIND, NV>(2, 3, nullptr);
    printf("%-26s two programs on one SIMD (one MFMA wave + one wave of %d v_fma per group): both %.3f ms, MFMA alone %.3f, VALU alone %.3f"
           " -> both / (sum) = %.2f, both / max = %.2f\n", what, NV, tb, tm, tv, tb / (tm + tv), tb / (tm > tv ? tm : tv));
    printf("    SIMD id of waves 0..7 of block 0 (HW_ID[5:4]): ");
    for (int w = 0; w < 8; ++w) printf("%u ", (hw[w] >> 4) & 3);
    printf(" (waves 0-3: MFMA program, 4-7: VALU program)\n");
  }
}

int main() {
  sweep<0, 0>("v_mfma_f32_4x4x1_16B_f32"); sweep<0, 1>("v_mfma_f32_4x4x1_16B_f32"); sweep<0, 2>("v_mfma_f32_4x4x1_16B_f32");
  sweep<0, 4>("v_mfma_f32_4x4x1_16B_f32"); sweep<0, 8>("v_mfma_f32_4x4x1_16B_f32");
  sweep<1, 0>("v_mfma_f32_16x16x4_f32"); sweep<1, 2>("v_mfma_f32_16x16x4_f32"); sweep<1, 4>("v_mfma_f32_16x16x4_f32");
  sweep<1, 8>("v_mfma_f32_16x16x4_f32");
  return 0;
}
