// k_csf.hip -- kernels of the CSF+Res2Net decoder head (CSF+Res2Net/networks/csf_res2net.py:227-256).
//
//   csf_gemm_kernel<MT>   out[n][m][p] = sum_k A[m][k] * G(k; n, p): every gOctConv 1x1 block (gOctConv.py:60-114) and
//                         every dense dilated 3x3 of the MSBlocks (csf_res2net.py:192-214) as an implicit GEMM on
//                         v_mfma_f32_16x16x4_f32.  Block tile (16 MT) rows x 256 pixels, K in chunks of 16:
//                           A chunk   one 128-bit load per thread from the zero-padded row-major weight image
//                                     (k contiguous), stored k-major in LDS;
//                           B chunk   lane = pixel; 16 channels of its pixel through a bounded buffer resource
//                                     (per-lane byte offset in ONE VGPR, the channel rides in the scalar offset):
//                                     own resolution / 3x3 tap (zero padding = out-of-range offset -> 0), one dword
//                                     per channel (inputs of another resolution are resized ONCE by csf_resize_kernel
//                                     into a channel slice of the level's gather tensor, gOctConv.py:99-101);
//                           both land in registers while the PREVIOUS chunk is being contracted (register prefetch,
//                           two LDS buffers, one barrier per chunk);
//                           contract  every wave owns (16 MT) x 64 outputs: per 4 k, MT A reads + 4 B reads feed
//                                     4 MT MFMAs; accumulators stay in VGPRs for the whole K loop.
//                         Blocks that share a pixel tile (different row tiles) are adjacent in the launch order and
//                         the order is cut into 8 contiguous chunks, one per XCD, so the re-gathered B operand is
//                         served by that XCD's L2.
//   csf_combine_kernel    adds the coarser levels' partial outputs through bilinear up-sampling (gOctConv.py:96-98,
//                         109-110) and takes the GroupNorm statistics (fp64 partials, fixed order) in the same pass
//   csf_gn_finalize / csf_apply_kernel   nn.GroupNorm(32, C) + nn.PReLU(C) (gOctConv.py:127-130,148-151)
//   csf_cls_kernel        fuse1x1's GroupNorm + PReLU applied on the fly + cls_layer (1x1 + bias) (csf_res2net.py:253)
//   csf_resize_kernel     F.interpolate(size, bilinear, align_corners=False) (csf_res2net.py:254)
//   csf_prep_kernel       weight images
#include <cstdlib>

#include "csf_kernels.h"
#include "pw_gather.h"   // csn_f4

// ---------------------------------------------------------------------------------------------------- GEMM
struct CsfGather {       // per-lane addressing of the current (pseudo-)segment
  unsigned o00;          // byte offset of the lane's pixel (or tap) in channel 0 of the segment
};

__device__ __forceinline__ void csf_seg_setup(const CsfGemmArgs& a, int ps, int dil, int n, int oy, int ox,
                                              CsfGather& g, int& si) {
  if (a.taps) {   // pseudo-segment = tap of the dilated 3x3, zero padding (csf_res2net.py:203)
    si = 0;
    const CsfSeg& s = a.seg[0];
    const int ty = ps / 3, tx = ps - 3 * ty;
    const int y = oy + (ty - 1) * dil, x = ox + (tx - 1) * dil;
    const bool ok = y >= 0 && y < s.Hs && x >= 0 && x < s.Ws;
    g.o00 = ok ? (unsigned)(n * s.nstride + y * s.Ws + x) * 4u : 0x80000000u;
    return;
  }
  si = ps;
  const CsfSeg& s = a.seg[ps];
  g.o00 = (unsigned)(n * s.nstride + oy * s.Ws + ox) * 4u;
}

template <int MT>
__global__ __launch_bounds__(256) void csf_gemm_kernel(CsfGemmArgs a) {
  constexpr int BM = 16 * MT, AP = BM + 16;
  CSN_DYN_SMEM(float, lds);
  float* As = lds;                       // [2][16][AP]
  float* Bs = lds + 2 * CSF_KC * AP;     // [2][16][CSF_BP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int per = gridDim.x >> 3;        // grid is a multiple of 8: contiguous chunk of the tile order per XCD
  const int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lb >= a.total_tiles) return;
  // sub-problem, then (row tile fastest, K slice, pixel tile): blocks that share a B tile are neighbours
  int sp = 0;
#pragma unroll
  for (int q = 1; q < CSF_MAX_SUB; ++q)
    if (q < a.nsub && lb >= a.sub[q].tile0) sp = q;
  const float* Aimg = a.sub[sp].A;
  float* outp = a.sub[sp].out;
  const int M = a.sub[sp].M, dil = a.sub[sp].dil, nmt = a.sub[sp].n_mtiles;
  const int local = lb - a.sub[sp].tile0;
  const int mt = local % nmt, rest = local / nmt;
  const int ks = rest % a.ksplit, nt = rest / a.ksplit;
  const int m0 = mt * BM;

  const int p = nt * CSF_BN + tid;
  const int pc = p < a.Ntot ? p : a.Ntot - 1;
  const int n = pc / a.HWo, pix = pc - n * a.HWo;
  const int oy = pix / a.Wo, ox = pix - oy * a.Wo;

  csn_f4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = csn_f4{0.f, 0.f, 0.f, 0.f};

  const int nps = a.taps ? 9 : a.nseg;
  const int nchunks = a.Kp / CSF_KC;
  const int c0 = ks * a.chunks_per_split;
  const int c1 = min(nchunks, c0 + a.chunks_per_split);
  // A: thread -> (row, 4 consecutive k)
  const bool a_ld = tid < BM * 4;
  const float* ap = Aimg + (size_t)(m0 + (tid >> 2)) * a.Kp + (tid & 3) * 4;

  // (pseudo-)segment and chunk inside it of the slice's first chunk
  int ps = 0, cc = c0, si = 0;
  if (a.taps) {
    ps = c0 / a.seg[0].chunks;
    cc = c0 - ps * a.seg[0].chunks;
  } else {
    while (ps + 1 < nps && cc >= a.seg[ps].chunks) cc -= a.seg[ps++].chunks;
  }
  CsfGather g;
  csf_seg_setup(a, ps, dil, n, oy, ox, g, si);
  csn_buf buf = csn_make_buf_n(a.seg[si].src, a.seg[si].bytes);

  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f);
  float rb[CSF_KC];

  auto fetch = [&](int kc) {
    if (a_ld) ra = *reinterpret_cast<const float4*>(ap + (size_t)kc * CSF_KC);
    const CsfSeg& s = a.seg[si];
    const unsigned cb = (unsigned)(cc * CSF_KC) * (unsigned)s.cstride * 4u;
    const unsigned cs = (unsigned)s.cstride * 4u;
#pragma unroll
    for (int r = 0; r < CSF_KC; ++r) rb[r] = csn_ld1(buf, g.o00, cb + r * cs);
    // advance to the next chunk's (pseudo-)segment
    if (++cc == a.seg[si].chunks) {
      cc = 0;
      if (++ps < nps) {
        csf_seg_setup(a, ps, dil, n, oy, ox, g, si);
        buf = csn_make_buf_n(a.seg[si].src, a.seg[si].bytes);
      }
    }
  };

  if (c0 < c1) fetch(c0);
  for (int kc = c0; kc < c1; ++kc) {
    float* Ab = As + ((kc - c0) & 1) * CSF_KC * AP;
    float* Bb = Bs + ((kc - c0) & 1) * CSF_KC * CSF_BP;
    if (a_ld) {
      const int row = tid >> 2, k4 = (tid & 3) * 4;
      Ab[(k4 + 0) * AP + row] = ra.x;
      Ab[(k4 + 1) * AP + row] = ra.y;
      Ab[(k4 + 2) * AP + row] = ra.z;
      Ab[(k4 + 3) * AP + row] = ra.w;
    }
#pragma unroll
    for (int r = 0; r < CSF_KC; ++r) Bb[r * CSF_BP + tid] = rb[r];
    __syncthreads();
    if (kc + 1 < c1) fetch(kc + 1);
    const float* bw = Bb + wave * 64;
#pragma unroll
    for (int kk = 0; kk < CSF_KC / 4; ++kk) {
#ifdef CSN_EMU_SEQ
      const int col = lane & 15;
      for (int i = 0; i < MT; ++i)
        for (int j = 0; j < 4; ++j)
          for (int e = 0; e < 4; ++e) {
            const int row = (lane >> 4) * 4 + e;
            float v = acc[i][j][e];
            for (int u = 0; u < 4; ++u)
              v = fmaf(Ab[(kk * 4 + u) * AP + i * 16 + row], bw[(kk * 4 + u) * CSF_BP + j * 16 + col], v);
            acc[i][j][e] = v;
          }
#else
      float av[MT], bv[4];
      const int kr = kk * 4 + (lane >> 4), c = lane & 15;
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = Ab[kr * AP + i * 16 + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = bw[kr * CSF_BP + j * 16 + c];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = csn_mfma_16x16x4(av[i], bv[j], acc[i][j]);
#endif
    }
  }

  // D layout: register e of a lane = row (lane>>4)*4 + e, column lane&15
  outp += (size_t)ks * a.split_stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = nt * CSF_BN + wave * 64 + j * 16 + (lane & 15);
    if (q >= a.Ntot) continue;
    const int qn = q / a.HWo, qp = q - qn * a.HWo;
    float* o = outp + (size_t)qn * a.out_nstride + qp;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + i * 16 + (lane >> 4) * 4 + e;
        if (m < M) o[(size_t)m * a.HWo] = acc[i][j][e];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// csf_gemm3_kernel (round 6): the same implicit GEMM on the bf16 matrix instruction, fp32 operands split THREE ways.
// The head is matrix-bound (K = 128 .. 3840) and v_mfma_f32_16x16x4_f32 tops out at 157 TFLOP/s, one sixteenth of the bf16 rate.
// An fp32 number is exactly the sum of three bfloat16 numbers (its 24 mantissa bits cut into 8 + 8 + 8 by truncation:
// x = x1 + x2 + x3), products of bfloat16 pairs are exact in fp32, and
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2) + O(2^-24 |a b|)
// so SIX v_mfma_f32_32x32x16_bf16 issues (fp32 accumulation) reproduce the fp32 product to within the rounding of the fp32
// instruction itself -- 6 / 16 of its matrix time.  The split costs vector work (4 operations per element, paid once per gathered
// B element and reused by all the block's rows); it and the matrix work of a SIMD add (profiles/r5_notes.md), hence 128-row blocks.
// Same block tile, gather, K order, split-K and launch order as csf_gemm_kernel; A is split on the way into LDS from the same
// fp32 weight image.  LDS per buffer: A [3][2][BM][8] + B [3][2][256][8] bfloat16 (k-group-major: a lane's 16-byte operand reads
// of a 32-row / 32-pixel tile are contiguous: no bank conflicts).
typedef csn_f16v csf_f16;
#ifndef CSN_EMU_SEQ
// three truncation parts of x as the upper halves of three floats (x == f(h) + f(m) + f(l) exactly)
__device__ __forceinline__ void csf_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned xb = csn_f_bits(x);
  h = xb & 0xffff0000u;
  const float r1 = x - csn_bits_f(h);
  m = csn_f_bits(r1) & 0xffff0000u;
  const float r2 = r1 - csn_bits_f(m);
  l = csn_f_bits(r2);
}
// bfloat16 pair (element 0 in the low half) from the upper halves of two floats
__device__ __forceinline__ unsigned csf_pack2(unsigned lo_elem, unsigned hi_elem) {
#ifdef CSN_CPU_EMU
  return (lo_elem >> 16) | (hi_elem & 0xffff0000u);
#else
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u);   // bytes {lo[2], lo[3], hi[2], hi[3]}: one v_perm_b32
#endif
}
#endif

template <int MT>   // BM = 16 MT rows per block: 64 (MT = 4) or 128 (MT = 8)
__global__ __launch_bounds__(256) void csf_gemm3_kernel(CsfGemmArgs a) {
  constexpr int BM = 16 * MT, TR = BM / 32;
  constexpr int ASZ = 3 * 2 * BM * 4, BSZ = 3 * 2 * CSF_BN * 4;   // dwords per buffer (8 bfloat16 = 4 dwords per (part, k group, row))
  CSN_DYN_SMEM(unsigned, lds);
  unsigned* As = lds;                 // [2][3][2][BM][4]
  unsigned* Bs = lds + 2 * ASZ;       // [2][3][2][256][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int per = gridDim.x >> 3;
  const int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lb >= a.total_tiles) return;
  int sp = 0;
#pragma unroll
  for (int q = 1; q < CSF_MAX_SUB; ++q)
    if (q < a.nsub && lb >= a.sub[q].tile0) sp = q;
  const float* Aimg = a.sub[sp].A;
  float* outp = a.sub[sp].out;
  const int M = a.sub[sp].M, dil = a.sub[sp].dil, nmt = a.sub[sp].n_mtiles;
  const int local = lb - a.sub[sp].tile0;
  const int mt = local % nmt, rest = local / nmt;
  const int ks = rest % a.ksplit, nt = rest / a.ksplit;
  const int m0 = mt * BM;

  const int p = nt * CSF_BN + tid;
  const int pc = p < a.Ntot ? p : a.Ntot - 1;
  const int n = pc / a.HWo, pix = pc - n * a.HWo;
  const int oy = pix / a.Wo, ox = pix - oy * a.Wo;

  csf_f16 acc[TR][2];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nps = a.taps ? 9 : a.nseg;
  const int nchunks = a.Kp / CSF_KC;
  const int c0 = ks * a.chunks_per_split;
  const int c1 = min(nchunks, c0 + a.chunks_per_split);
  // A: thread -> (row, 8 consecutive k): BM * 2 threads carry the chunk (the weight image is padded to whole block tiles)
  const bool a_ld = tid < BM * 2;
  const float* ap = Aimg + (size_t)(m0 + (tid >> 1)) * a.Kp + (tid & 1) * 8;

  int ps = 0, cc = c0, si = 0;
  if (a.taps) {
    ps = c0 / a.seg[0].chunks;
    cc = c0 - ps * a.seg[0].chunks;
  } else {
    while (ps + 1 < nps && cc >= a.seg[ps].chunks) cc -= a.seg[ps++].chunks;
  }
  CsfGather g;
  csf_seg_setup(a, ps, dil, n, oy, ox, g, si);
  csn_buf buf = csn_make_buf_n(a.seg[si].src, a.seg[si].bytes);

  csn_f4 ra0 = csn_f4{0.f, 0.f, 0.f, 0.f}, ra1 = csn_f4{0.f, 0.f, 0.f, 0.f};
  float rb[CSF_KC];

  auto fetch = [&](int kc) {
    if (a_ld) {
      ra0 = *reinterpret_cast<const csn_f4*>(ap + (size_t)kc * CSF_KC);
      ra1 = *reinterpret_cast<const csn_f4*>(ap + (size_t)kc * CSF_KC + 4);
    }
    const CsfSeg& s = a.seg[si];
    const unsigned cb = (unsigned)(cc * CSF_KC) * (unsigned)s.cstride * 4u;
    const unsigned cs = (unsigned)s.cstride * 4u;
#pragma unroll
    for (int r = 0; r < CSF_KC; ++r) rb[r] = csn_ld1(buf, g.o00, cb + r * cs);
    if (++cc == a.seg[si].chunks) {
      cc = 0;
      if (++ps < nps) {
        csf_seg_setup(a, ps, dil, n, oy, ox, g, si);
        buf = csn_make_buf_n(a.seg[si].src, a.seg[si].bytes);
      }
    }
  };

  if (c0 < c1) fetch(c0);
  for (int kc = c0; kc < c1; ++kc) {
    unsigned* Ab = As + ((kc - c0) & 1) * ASZ;
    unsigned* Bb = Bs + ((kc - c0) & 1) * BSZ;
#ifdef CSN_EMU_SEQ
    // functional stand-in (the lane map itself runs under `make LANES=1`, csn_device.h): the chunk as plain floats, row-major
    float* Af = reinterpret_cast<float*>(Ab);   // [BM][16] needs BM * 16 <= ASZ: 16 BM <= 24 BM
    float* Bf = reinterpret_cast<float*>(Bb);   // [256][16]
    if (a_ld) {
      const int row = tid >> 1, k8 = (tid & 1) * 8;
      for (int e = 0; e < 4; ++e) { Af[row * 16 + k8 + e] = ra0[e]; Af[row * 16 + k8 + 4 + e] = ra1[e]; }
    }
    for (int r = 0; r < CSF_KC; ++r) Bf[tid * 16 + r] = rb[r];
    __syncthreads();
    if (kc + 1 < c1) fetch(kc + 1);
    {
      const int j32 = lane & 31, ih = (lane >> 5) * 4;
      for (int i = 0; i < TR; ++i)
        for (int j = 0; j < 2; ++j)
          for (int e = 0; e < 16; ++e) {
            const int row = i * 32 + (e & 3) + 8 * (e >> 2) + ih, col = wave * 64 + j * 32 + j32;
            float v = acc[i][j][e];
            for (int u = 0; u < 16; ++u) v = fmaf(Af[row * 16 + u], Bf[col * 16 + u], v);
            acc[i][j][e] = v;
          }
    }
    __syncthreads();
#else
    // ---- split and stage: part q, k group h (k = 8 h .. 8 h + 7), row / pixel r: four dwords at ((q * 2 + h) * R + r) * 4
    if (a_ld) {
      const int row = tid >> 1, hh = tid & 1;
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { csf_split3(ra0[e], h[e], m[e], l[e]); csf_split3(ra1[e], h[4 + e], m[4 + e], l[4 + e]); }
      csn_u4 q0, q1, q2;
      q0.x = csf_pack2(h[0], h[1]); q0.y = csf_pack2(h[2], h[3]); q0.z = csf_pack2(h[4], h[5]); q0.w = csf_pack2(h[6], h[7]);
      q1.x = csf_pack2(m[0], m[1]); q1.y = csf_pack2(m[2], m[3]); q1.z = csf_pack2(m[4], m[5]); q1.w = csf_pack2(m[6], m[7]);
      q2.x = csf_pack2(l[0], l[1]); q2.y = csf_pack2(l[2], l[3]); q2.z = csf_pack2(l[4], l[5]); q2.w = csf_pack2(l[6], l[7]);
      *reinterpret_cast<csn_u4*>(Ab + ((0 * 2 + hh) * BM + row) * 4) = q0;
      *reinterpret_cast<csn_u4*>(Ab + ((1 * 2 + hh) * BM + row) * 4) = q1;
      *reinterpret_cast<csn_u4*>(Ab + ((2 * 2 + hh) * BM + row) * 4) = q2;
    }
    {
      unsigned h[16], m[16], l[16];
#pragma unroll
      for (int r = 0; r < CSF_KC; ++r) csf_split3(rb[r], h[r], m[r], l[r]);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        csn_u4 q0, q1, q2;
        q0.x = csf_pack2(h[8 * hh + 0], h[8 * hh + 1]); q0.y = csf_pack2(h[8 * hh + 2], h[8 * hh + 3]);
        q0.z = csf_pack2(h[8 * hh + 4], h[8 * hh + 5]); q0.w = csf_pack2(h[8 * hh + 6], h[8 * hh + 7]);
        q1.x = csf_pack2(m[8 * hh + 0], m[8 * hh + 1]); q1.y = csf_pack2(m[8 * hh + 2], m[8 * hh + 3]);
        q1.z = csf_pack2(m[8 * hh + 4], m[8 * hh + 5]); q1.w = csf_pack2(m[8 * hh + 6], m[8 * hh + 7]);
        q2.x = csf_pack2(l[8 * hh + 0], l[8 * hh + 1]); q2.y = csf_pack2(l[8 * hh + 2], l[8 * hh + 3]);
        q2.z = csf_pack2(l[8 * hh + 4], l[8 * hh + 5]); q2.w = csf_pack2(l[8 * hh + 6], l[8 * hh + 7]);
        *reinterpret_cast<csn_u4*>(Bb + ((0 * 2 + hh) * CSF_BN + tid) * 4) = q0;
        *reinterpret_cast<csn_u4*>(Bb + ((1 * 2 + hh) * CSF_BN + tid) * 4) = q1;
        *reinterpret_cast<csn_u4*>(Bb + ((2 * 2 + hh) * CSF_BN + tid) * 4) = q2;
      }
    }
    __syncthreads();
    if (kc + 1 < c1) fetch(kc + 1);
    {
      // lane l: A row (l & 31) of a 32-row tile, B pixel (l & 31) of a 32-pixel tile, k group l >> 5
      const int r32 = lane & 31, hh = lane >> 5;
      csn_u4 av[3][TR], bv[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TR; ++i) av[q][i] = *reinterpret_cast<const csn_u4*>(Ab + ((q * 2 + hh) * BM + i * 32 + r32) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[q][j] = *reinterpret_cast<const csn_u4*>(Bb + ((q * 2 + hh) * CSF_BN + wave * 64 + j * 32 + r32) * 4);
      }
#define CSF_MMA(QA, QB)                                                                                                        \
  acc[i][j] = csn_mfma_32x32x16_bf16(av[QA][i], bv[QB][j], acc[i][j])
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // smallest terms first
          CSF_MMA(1, 1); CSF_MMA(2, 0); CSF_MMA(0, 2); CSF_MMA(1, 0); CSF_MMA(0, 1); CSF_MMA(0, 0);
        }
#undef CSF_MMA
    }
#endif
  }

  // D layout of the 32 x 32 tile: register e of a lane = row (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column lane & 31
  outp += (size_t)ks * a.split_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = nt * CSF_BN + wave * 64 + j * 32 + (lane & 31);
    if (q >= a.Ntot) continue;
    const int qn = q / a.HWo, qp = q - qn * a.HWo;
    float* o = outp + (size_t)qn * a.out_nstride + qp;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < M) o[(size_t)m * a.HWo] = acc[i][j][e];
      }
  }
}

bool csf_gemm_f32() {   // CSF_GEMM_F32=1: the GEMMs on the fp32 matrix instruction (csf_gemm_kernel; A/B against csf_gemm3_kernel)
  static const bool f32 = getenv("CSF_GEMM_F32") && getenv("CSF_GEMM_F32")[0] == '1';
  return f32;
}

int csf_launch_gemm(const CsfGemmArgs& a, int mt, void* stream) {
  const int tiles = a.total_tiles;
  if (tiles <= 0) return 0;
  const int grid = (tiles + 7) / 8 * 8;
  // fp32 operands as three bfloat16 parts on the bf16 matrix instruction (csf_gemm3_kernel); CSF_GEMM_F32=1: the fp32 instruction (A/B)
  if (!csf_gemm_f32() && (mt == 2 || mt == 4 || mt == 8)) {
    const size_t lds = (size_t)2 * (3 * 2 * 16 * mt * 4 + 3 * 2 * CSF_BN * 4) * sizeof(unsigned);
    if (mt == 8) {
#ifndef CSN_CPU_EMU
      static CsnPerDeviceOnce once;
      const int st = once.run([&]() {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(csf_gemm3_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      });
      if (st != 0) return st;
#endif
      CSN_LAUNCH(csf_gemm3_kernel<8>, dim3(grid), dim3(256), lds, stream, a);
    } else if (mt == 4) {
      CSN_LAUNCH(csf_gemm3_kernel<4>, dim3(grid), dim3(256), lds, stream, a);
    } else {
      CSN_LAUNCH(csf_gemm3_kernel<2>, dim3(grid), dim3(256), lds, stream, a);
    }
    return (int)hipGetLastError();
  }
  if (mt == 4) {
    const size_t lds = (size_t)2 * CSF_KC * (64 + 16 + CSF_BP) * sizeof(float);
    CSN_LAUNCH(csf_gemm_kernel<4>, dim3(grid), dim3(256), lds, stream, a);
  } else if (mt == 2) {
    const size_t lds = (size_t)2 * CSF_KC * (32 + 16 + CSF_BP) * sizeof(float);
    CSN_LAUNCH(csf_gemm_kernel<2>, dim3(grid), dim3(256), lds, stream, a);
  } else {
    return -1;
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------ combine + GroupNorm statistics
// One block per (image, CSF_CG consecutive channels).  The coarser tensors' planes of those channels (a few KB each:
// 44x44 + 22x22 + 11x11 floats at 352x352) are staged in LDS once -- split-K slices summed on the way in -- so the
// 4 taps x nz levels per output value are LDS reads, and a lane computes the interpolation indices / weights of its
// pixel once for the CSF_CG channels; the planes stream through once (coalesced read + write).
#define CSF_CG 4
// fixed-order block sum: butterfly inside the wave, then the four wave results (one barrier pair instead of a
// log2(256)-step LDS tree: the tree's barriers were a quarter of a block's life)
__device__ __forceinline__ double csf_block_sum(double v, double* sm) {
  const int tid = threadIdx.x;
#ifdef CSN_EMU_SEQ
  sm[tid] = v;
  __syncthreads();
  for (int s = CSN_BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) sm[tid] += sm[tid + s];
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += csn_shfl_xor(v, o);
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  __syncthreads();
  const double r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
#endif
}

__global__ __launch_bounds__(CSN_BLOCK) void csf_combine_kernel(CsfCombArgs a) {
  CSN_DYN_SMEM(double, red);   // [2][CSN_BLOCK] doubles, then the staged planes (floats)
  float* zl = reinterpret_cast<float*>(red + 2 * CSN_BLOCK);
  const int ch0 = blockIdx.x * CSF_CG, n = blockIdx.y;
  const int tid = threadIdx.x;
  int zoff[3] = {0, 0, 0};
  if (a.z_in_lds) {
    int off = 0;
    for (int i = 0; i < a.nz; ++i) {
      const CsfZ& z = a.z[i];
      const int len = z.Hz * z.Wz;
      const float* q = z.z + (long long)n * z.nstride + (long long)ch0 * len;   // CSF_CG planes are contiguous
      for (int e = tid; e < CSF_CG * len; e += CSN_BLOCK) {   // [channel][pixel] -> [pixel][channel]
        float v = q[e];
        for (int k = 1; k < z.ns; ++k) v += q[(long long)k * z.split_stride + e];
        const int c = e / len;
        zl[off + (e - c * len) * CSF_CG + c] = v;
      }
      zoff[i] = off;
      off += CSF_CG * len;
    }
    __syncthreads();
  }
  float* sp = a.s + ((long long)n * a.C + ch0) * a.HW;
  double s1[CSF_CG], s2[CSF_CG];
#pragma unroll
  for (int c = 0; c < CSF_CG; ++c) s1[c] = s2[c] = 0.0;
  int y = tid / a.W, x = tid - y * a.W;
  for (int pix = tid; pix < a.HW; pix += CSN_BLOCK) {
    float v[CSF_CG];
#pragma unroll
    for (int c = 0; c < CSF_CG; ++c) {
      const float* q = sp + (long long)c * a.HW + pix;
      v[c] = q[0];
      for (int k = 1; k < a.ns; ++k) v[c] += q[(long long)k * a.split_stride];
    }
    for (int i = 0; i < a.nz; ++i) {
      const CsfZ& z = a.z[i];
      int y0, y1, x0, x1;
      float ly, lx;
      csn_bilin(y, z.ry, z.Hz, y0, y1, ly);
      csn_bilin(x, z.rx, z.Wz, x0, x1, lx);
      const int o00 = y0 * z.Wz + x0, o01 = y0 * z.Wz + x1, o10 = y1 * z.Wz + x0, o11 = y1 * z.Wz + x1;
      const float wx0 = 1.f - lx, wy0 = 1.f - ly;
      if (a.z_in_lds) {   // one 128-bit LDS read per tap serves the CSF_CG channels
        const float4* q = reinterpret_cast<const float4*>(zl + zoff[i]);
        const float4 t00 = q[o00], t01 = q[o01], t10 = q[o10], t11 = q[o11];
        v[0] += wy0 * (wx0 * t00.x + lx * t01.x) + ly * (wx0 * t10.x + lx * t11.x);
        v[1] += wy0 * (wx0 * t00.y + lx * t01.y) + ly * (wx0 * t10.y + lx * t11.y);
        v[2] += wy0 * (wx0 * t00.z + lx * t01.z) + ly * (wx0 * t10.z + lx * t11.z);
        v[3] += wy0 * (wx0 * t00.w + lx * t01.w) + ly * (wx0 * t10.w + lx * t11.w);
      } else {
        const long long len = (long long)z.Hz * z.Wz;
        const float* q = z.z + (long long)n * z.nstride + (long long)ch0 * len;
#pragma unroll
        for (int c = 0; c < CSF_CG; ++c, q += len) {
          float v00 = q[o00], v01 = q[o01], v10 = q[o10], v11 = q[o11];
          for (int k = 1; k < z.ns; ++k) {
            const float* qk = q + (long long)k * z.split_stride;
            v00 += qk[o00]; v01 += qk[o01]; v10 += qk[o10]; v11 += qk[o11];
          }
          v[c] += wy0 * (wx0 * v00 + lx * v01) + ly * (wx0 * v10 + lx * v11);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CSF_CG; ++c) {
      if (a.nz || a.ns > 1) sp[(long long)c * a.HW + pix] = v[c];
      s1[c] += (double)v[c];
      s2[c] += (double)v[c] * (double)v[c];
    }
    x += a.step_x;                       // pix += CSN_BLOCK without a division
    y += a.step_y;
    if (x >= a.W) { x -= a.W; ++y; }
  }
#pragma unroll
  for (int c = 0; c < CSF_CG; ++c) {   // [image][channel][2]: csf_gn_finalize sums the cpg channels of a group in order
    const double r1 = csf_block_sum(s1[c], red), r2 = csf_block_sum(s2[c], red);
    if (tid == 0) {
      a.part[((long long)n * a.C + ch0 + c) * 2 + 0] = r1;
      a.part[((long long)n * a.C + ch0 + c) * 2 + 1] = r2;
    }
  }
}

int csf_launch_combine(const CsfCombArgs& a, void* stream) {
  if (a.C % CSF_CG) return -1;
  const size_t lds = 2 * CSN_BLOCK * sizeof(double) + (a.z_in_lds ? (size_t)CSF_CG * a.z_floats * sizeof(float) : 0);
  CSN_LAUNCH(csf_combine_kernel, dim3(a.C / CSF_CG, a.B, 1), dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}

// one thread per (image, channel): group moments (fixed-order sum of the slabs) -> folded scale / shift
__global__ __launch_bounds__(CSN_BLOCK) void csf_gn_finalize_kernel(CsfGnFinArgs a) {
  const int i = blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (i >= a.B * a.C) return;
  const int n = i / a.C, c = i - n * a.C;
  const int ng = n * a.groups + c / a.cpg;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < a.nslab; ++s) {
    s1 += a.part[((long long)ng * a.nslab + s) * 2 + 0];
    s2 += a.part[((long long)ng * a.nslab + s) * 2 + 1];
  }
  const double cnt = (double)a.cpg * (double)a.HW;
  const double mean = s1 / cnt;
  double var = s2 / cnt - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float sc = rstd * a.gamma[c];
  a.scale[i] = sc;
  a.shift[i] = a.beta[c] - sc * (float)mean;
}

int csf_launch_gn_finalize(const CsfGnFinArgs& a, void* stream) {
  const int nblk = (a.B * a.C + CSN_BLOCK - 1) / CSN_BLOCK;
  CSN_LAUNCH(csf_gn_finalize_kernel, dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(CSN_BLOCK) void csf_apply_kernel(CsfApplyArgs a) {
  for (long long i = (long long)blockIdx.x * CSN_BLOCK + threadIdx.x; i < a.total; i += (long long)gridDim.x * CSN_BLOCK) {
    const long long plane = i / a.HW;          // n*C + c
    const int c = (int)(plane % a.C);
    const float y = fmaf(a.s[i], a.scale[plane], a.shift[plane]);
    a.s[i] = y >= 0.f ? y : a.alpha[c] * y;
  }
}

int csf_launch_apply(const CsfApplyArgs& a, void* stream) {
  long long nblk = (a.total + CSN_BLOCK - 1) / CSN_BLOCK;
  if (nblk > 16384) nblk = 16384;
  if (nblk <= 0) return 0;
  CSN_LAUNCH(csf_apply_kernel, dim3((unsigned)nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

// eval BatchNorm (+ residual) (+ ReLU) in place: one block per chunk of an (image, channel) plane, 128-bit accesses when
// the plane size allows; every lane folds the channel's four BatchNorm scalars itself (cached, wave-uniform addresses)
__global__ __launch_bounds__(CSN_BLOCK) void csf_bn_act_kernel(CsfBnActArgs a) {
  const int plane = blockIdx.x / a.chunks, chunk = blockIdx.x - plane * a.chunks;
  const int c = plane % a.C;
  const float sc = a.gamma[c] / sqrtf(a.var[c] + a.eps);
  const float sh = a.beta[c] - a.mean[c] * sc;
  float* x = a.x + (long long)plane * a.HW;
  const float* r = a.res ? a.res + (long long)plane * a.HW : nullptr;
  if ((a.HW & 3) == 0) {
    const int n4 = a.HW >> 2;
    float4* x4 = reinterpret_cast<float4*>(x);
    const float4* r4 = reinterpret_cast<const float4*>(r);
    for (int i = chunk * CSN_BLOCK + threadIdx.x; i < n4; i += a.chunks * CSN_BLOCK) {
      float4 v = x4[i];
      v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
      if (r) { const float4 q = r4[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      x4[i] = v;
    }
  } else {
    for (int i = chunk * CSN_BLOCK + threadIdx.x; i < a.HW; i += a.chunks * CSN_BLOCK) {
      float v = fmaf(x[i], sc, sh);
      if (r) v += r[i];
      x[i] = a.relu ? fmaxf(v, 0.f) : v;
    }
  }
}

int csf_launch_bn_act(const CsfBnActArgs& a0, int planes, void* stream) {
  CsfBnActArgs a = a0;
  const int per = (a.HW & 3) == 0 ? a.HW >> 2 : a.HW;             // work items of a plane
  a.chunks = (per + 4 * CSN_BLOCK - 1) / (4 * CSN_BLOCK);          // ~4 items per lane
  if (a.chunks < 1) a.chunks = 1;
  const long long nblk = (long long)planes * a.chunks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return -1;
  CSN_LAUNCH(csf_bn_act_kernel, dim3((unsigned)nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

// lane = pixel (coalesced channel planes), four independent accumulators over the channels
__global__ __launch_bounds__(CSN_BLOCK) void csf_cls_kernel(CsfClsArgs a) {
  const long long i = (long long)blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (i >= (long long)a.B * a.HW) return;
  const int n = (int)(i / a.HW), pix = (int)(i - (long long)n * a.HW);
  const float* s = a.s + (long long)n * a.C * a.HW + pix;
  const float* sc = a.scale + (long long)n * a.C;
  const float* sh = a.shift + (long long)n * a.C;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= a.C; c += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s[(long long)(c + u) * a.HW];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float y = fmaf(v[u], sc[c + u], sh[c + u]);
      y = y >= 0.f ? y : a.alpha[c + u] * y;
      acc[u & 3] = fmaf(a.w[c + u], y, acc[u & 3]);
    }
  }
  for (; c < a.C; ++c) {
    float y = fmaf(s[(long long)c * a.HW], sc[c], sh[c]);
    y = y >= 0.f ? y : a.alpha[c] * y;
    acc[0] = fmaf(a.w[c], y, acc[0]);
  }
  a.out[i] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + a.bias[0];
}

int csf_launch_cls(const CsfClsArgs& a, void* stream) {
  const long long total = (long long)a.B * a.HW;
  const int nblk = (int)((total + CSN_BLOCK - 1) / CSN_BLOCK);
  if (nblk <= 0) return 0;
  CSN_LAUNCH(csf_cls_kernel, dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(CSN_BLOCK) void csf_resize_kernel(CsfResizeArgs a) {
  const long long total = (long long)a.planes * a.Ho * a.Wo;
  const long long i = (long long)blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % a.Wo);
  const int y = (int)((i / a.Wo) % a.Ho);
  const long long plane = i / ((long long)a.Wo * a.Ho);
  const float* p = a.in + plane * a.Hi * a.Wi;
  const long long n = plane / a.cpi, c = plane - n * a.cpi;
  float* o = a.out + n * a.out_nstride + (c * a.Ho + y) * a.Wo + x;
  if (a.Hi == a.Ho && a.Wi == a.Wo) {   // F.interpolate copies when the size does not change
    *o = p[y * a.Wi + x];
    return;
  }
  int y0, y1, x0, x1;
  float ly, lx;
  csn_bilin(y, a.ry, a.Hi, y0, y1, ly);
  csn_bilin(x, a.rx, a.Wi, x0, x1, lx);
  const float v0 = (1.f - lx) * p[y0 * a.Wi + x0] + lx * p[y0 * a.Wi + x1];
  const float v1 = (1.f - lx) * p[y1 * a.Wi + x0] + lx * p[y1 * a.Wi + x1];
  *o = (1.f - ly) * v0 + ly * v1;
}

int csf_launch_resize(const CsfResizeArgs& a, void* stream) {
  const long long total = (long long)a.planes * a.Ho * a.Wo;
  const int nblk = (int)((total + CSN_BLOCK - 1) / CSN_BLOCK);
  if (nblk <= 0) return 0;
  CSN_LAUNCH(csf_resize_kernel, dim3(nblk), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- weight images
__global__ __launch_bounds__(CSN_BLOCK) void csf_prep_kernel(const CsfPrepJobDev* __restrict__ jobs, int njobs,
                                                              const float* __restrict__ arena, float* __restrict__ packed) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
  const CsfPrepJobDev* __restrict__ a = jobs + j;   // read in place: a by-value copy with dynamic seg[] indexing goes to scratch
  const float* src = arena + a->src_off;
  float* dst = packed + a->dst_off;
  const long long total = (long long)a->Mp * a->Kp;
  for (long long i = (long long)(blockIdx.x - a->blk0) * CSN_BLOCK + threadIdx.x; i < total; i += (long long)a->nblk * CSN_BLOCK) {
    const int m = (int)(i / a->Kp), k = (int)(i - (long long)m * a->Kp);
    float v = 0.f;
    if (m < a->M) {
      if (a->taps) {
        const int Cp = a->seg[0].k0;            // taps: k = tap * Cp + c (Cp = channels padded to the K chunk)
        const int tap = k / Cp, c = k - tap * Cp;
        if (tap < 9 && c < a->seg[0].C) v = src[(long long)m * a->ld + (long long)c * 9 + tap];
      } else {
        for (int s = 0; s < a->nseg; ++s) {
          const int kk = k - a->seg[s].k0;
          if (kk >= 0 && kk < a->seg[s].C) v = src[(long long)m * a->ld + a->seg[s].col0 + kk];
        }
      }
    }
    dst[i] = v;
  }
}

int csf_launch_prep_all(const CsfPrepJobDev* jobs_dev, int njobs, int total_blocks, const float* arena, float* packed,
                        void* stream) {
  if (njobs <= 0 || total_blocks <= 0) return 0;
  CSN_LAUNCH(csf_prep_kernel, dim3((unsigned)total_blocks), dim3(CSN_BLOCK), 0, stream, jobs_dev, njobs, arena, packed);
  return (int)hipGetLastError();
}
