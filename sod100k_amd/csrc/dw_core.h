// dw_core.h -- the depthwise 3x3 (+ BN + PReLU) row core shared by the eval-mode kernels (round 5: k_ilb.hip, k_misc.hip
// dw3x3x2_bn_prelu_kernel).  Reference: SimplifiedGOctConvBR (CSNet/model/csnet.py:795-851) = F.conv2d(x, 100 W, padding 1, groups C)
// (conv2d.py:104) -> BatchNorm (eval: running statistics) -> PReLU, per channel.
//
// The depthwise kernels are bound by vector-instruction issue, not by bytes (profiles/r4_notes.md: 83-85 % VALU utilisation; round 5's
// issue_probe2 confirms every instruction pays its own slot).  Round 4's loop spent ~13 vector instructions per output pixel and unit:
// nine taps as packed FMAs (4.5), BN as its own packed FMA (0.5), PReLU as multiply + compare + select (2.5), pair-building moves,
// per-element masks.  This core spends 4.5 + 1.5:
//   * the BatchNorm scale is folded into the nine weights when the parameters are packed (w' = 100 w gamma / sqrt(var + eps),
//     CSN_PREP_DWREC) and the shift is the accumulator's initial value: the nine packed FMAs ARE conv + BN;
//   * PReLU(y) = max(y, alpha y) for alpha <= 1 and min(y, alpha y) for alpha > 1 = v_med3_f32(y, alpha y, +-inf): one packed
//     multiply per pair and one median per value, no compare / select through VCC, any alpha;
//   * a row of the 3-row window is kept as the FIVE overlapping pairs (v0,v1) .. (v4,v5) a packed FMA can take as an operand;
//     from LDS they are loaded as pairs (ds_read2_b32 delivers any two dwords into a register pair: no moves at all);
//   * the two pair chains of a row's four outputs are interleaved, so no packed FMA waits for its predecessor.
#pragma once
#include "csn_device.h"

#ifdef CSN_CPU_EMU
struct csn_v2 {
  float v[2];
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};
__device__ __forceinline__ csn_v2 csn_mk2(float a, float b) { csn_v2 r; r.v[0] = a; r.v[1] = b; return r; }
__device__ __forceinline__ csn_v2 csn_fma2(float w, csn_v2 x, csn_v2 c) { return csn_mk2(fmaf(w, x[0], c[0]), fmaf(w, x[1], c[1])); }
__device__ __forceinline__ csn_v2 csn_mul2(float w, csn_v2 x) { return csn_mk2(w * x[0], w * x[1]); }
__device__ __forceinline__ csn_v2 csn_fmav2(csn_v2 a, csn_v2 b, csn_v2 c) { return csn_mk2(fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])); }
#else
typedef float csn_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ csn_v2 csn_mk2(float a, float b) { csn_v2 r = {a, b}; return r; }
__device__ __forceinline__ csn_v2 csn_fma2(float w, csn_v2 x, csn_v2 c) {   // v_pk_fma_f32
  const csn_v2 ww = {w, w};
  return __builtin_elementwise_fma(ww, x, c);
}
__device__ __forceinline__ csn_v2 csn_mul2(float w, csn_v2 x) { const csn_v2 ww = {w, w}; return ww * x; }
__device__ __forceinline__ csn_v2 csn_fmav2(csn_v2 a, csn_v2 b, csn_v2 c) { return __builtin_elementwise_fma(a, b, c); }   // pair x pair + pair
#endif

// per-channel record of one depthwise unit (CSN_PREP_DWREC + strided BN_SHIFT / COPY jobs): {w'[9], shift, alpha, 0}
#define DWREC_FLOATS 12
struct DwPar { float w[9], sh, al, lim; };   // lim = +inf when alpha <= 1, else -inf (dw_prelu2)
template <typename PTR>   // const float* (LDS, per lane) or csn_cfp (packed buffer through the scalar cache, block-uniform channel)
__device__ __forceinline__ DwPar dw_par_load(PTR rec) {
  DwPar p;
#pragma unroll
  for (int i = 0; i < 9; ++i) p.w[i] = rec[i];
  p.sh = rec[9]; p.al = rec[10];
  p.lim = p.al <= 1.f ? __builtin_inff() : -__builtin_inff();
  return p;
}

struct DwRow2 { csn_v2 a, b, c, d, e; };   // (v0,v1) (v1,v2) (v2,v3) (v3,v4) (v4,v5); v1..v4 = the lane's four columns

// a row out of an LDS plane; p = column x0 - 1 (v0) of the row
__device__ __forceinline__ DwRow2 dw_row2_lds(const float* p) {
  DwRow2 r;
  r.a = csn_mk2(p[0], p[1]); r.b = csn_mk2(p[1], p[2]); r.c = csn_mk2(p[2], p[3]); r.d = csn_mk2(p[3], p[4]); r.e = csn_mk2(p[4], p[5]);
  return r;
}
// ... when p + 1 (the lane's first own column) is 16-byte aligned: ONE 128-bit read for the four own columns (64 lanes x 16
// contiguous bytes: conflict-free) + the two halo columns.  Dword reads at a 16-byte lane stride only reach a quarter of the banks
// -- the five-pair form above made the LDS pipe, not the vector pipe, the limit of ilb_kernel's depthwise phases (round 5).
__device__ __forceinline__ DwRow2 dw_row2_lds4(const float* p) {
#ifdef CSN_CPU_EMU
  const float c[4] = {p[1], p[2], p[3], p[4]};
#else
  typedef float v4 __attribute__((ext_vector_type(4)));   // (HIP's float4 is a struct: its load is scalarised and re-paired)
  const v4 c = *reinterpret_cast<const v4*>(__builtin_assume_aligned(p + 1, 16));
#endif
  const float l = p[0], r = p[5];
  DwRow2 q;
  q.a = csn_mk2(l, c[0]); q.b = csn_mk2(c[0], c[1]); q.c = csn_mk2(c[1], c[2]); q.d = csn_mk2(c[2], c[3]); q.e = csn_mk2(c[3], r);
  return q;
}
// ... out of six registers (rows that come from HBM: one 128-bit load + the two halo columns)
__device__ __forceinline__ DwRow2 dw_row2_regs(float v0, float v1, float v2, float v3, float v4, float v5) {
  DwRow2 r;
  r.a = csn_mk2(v0, v1); r.b = csn_mk2(v1, v2); r.c = csn_mk2(v2, v3); r.d = csn_mk2(v3, v4); r.e = csn_mk2(v4, v5);
  return r;
}

// conv3x3 + BN of the lane's four columns of one output row: o01 = outputs 0, 1; o23 = outputs 2, 3
__device__ __forceinline__ void dw_conv4(const DwPar& p, const DwRow2& t, const DwRow2& m, const DwRow2& b, csn_v2& o01, csn_v2& o23) {
  const csn_v2 s = csn_mk2(p.sh, p.sh);
  o01 = csn_fma2(p.w[0], t.a, s);   o23 = csn_fma2(p.w[0], t.c, s);
  o01 = csn_fma2(p.w[1], t.b, o01); o23 = csn_fma2(p.w[1], t.d, o23);
  o01 = csn_fma2(p.w[2], t.c, o01); o23 = csn_fma2(p.w[2], t.e, o23);
  o01 = csn_fma2(p.w[3], m.a, o01); o23 = csn_fma2(p.w[3], m.c, o23);
  o01 = csn_fma2(p.w[4], m.b, o01); o23 = csn_fma2(p.w[4], m.d, o23);
  o01 = csn_fma2(p.w[5], m.c, o01); o23 = csn_fma2(p.w[5], m.e, o23);
  o01 = csn_fma2(p.w[6], b.a, o01); o23 = csn_fma2(p.w[6], b.c, o23);
  o01 = csn_fma2(p.w[7], b.b, o01); o23 = csn_fma2(p.w[7], b.d, o23);
  o01 = csn_fma2(p.w[8], b.c, o01); o23 = csn_fma2(p.w[8], b.e, o23);
}

// PReLU of a pair, any alpha, no compare / select: y >= 0 ? y : alpha y  ==  max(y, alpha y) for alpha <= 1 (any sign: for y < 0,
// alpha y >= y; for y >= 0, alpha y <= y) and min(y, alpha y) for alpha > 1; both are the median of (y, alpha y, +-inf):
// one packed multiply per pair + one v_med3_f32 per value
__device__ __forceinline__ float dw_med3(float a, float b, float c) {
#ifdef CSN_CPU_EMU
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#else
  return __builtin_amdgcn_fmed3f(a, b, c);
#endif
}
__device__ __forceinline__ csn_v2 dw_prelu2(csn_v2 y, float al, float lim) {
  const csn_v2 t = csn_mul2(al, y);
  return csn_mk2(dw_med3(y[0], t[0], lim), dw_med3(y[1], t[1], lim));
}

// four outputs -> an LDS row position that is 16-byte aligned
__device__ __forceinline__ void dw_st4_lds(float* p, csn_v2 o01, csn_v2 o23) {
#ifdef CSN_CPU_EMU
  p[0] = o01[0]; p[1] = o01[1]; p[2] = o23[0]; p[3] = o23[1];
#else
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = {o01[0], o01[1], o23[0], o23[1]};
  *reinterpret_cast<v4*>(__builtin_assume_aligned(p, 16)) = v;
#endif
}
