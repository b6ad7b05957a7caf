// csf_kernels.h -- argument blocks and launchers of the CSF+Res2Net head kernels (k_csf.hip).
//
// Everything is planar fp32 [B][C][H*W].  The head is dense contraction work (K = 128 .. 3840), so unlike the
// CSNet-100K kernels it is priced against the fp32 matrix-core roofline: csf_gemm_kernel is a block-tiled
// implicit GEMM  out[n][m][p] = sum_k A[m][k] * G(k; n, p)  on v_mfma_f32_16x16x4_f32 whose B operand G is gathered
// on the fly from up to four source tensors (own resolution, bilinear resample of another level, dilated 3x3 taps).
#pragma once
#include "csn_device.h"

#define CSF_KC 16            // K chunk staged per barrier
#define CSF_BN 256           // pixels per block: 4 waves x 64
#define CSF_BP (CSF_BN + 16) // LDS pitch of a B row (pitch mod 32 == 16: the 4 k rows of one MFMA read hit distinct banks)
#define CSF_MAX_SEG 4

enum CsfSegMode { CSF_OWN = 0 };   // every segment is read at the output resolution (resized inputs are materialised once)

struct CsfSeg {
  const float* src;      // [B][ctot][Hs*Ws], already offset to the first channel of the slice
  unsigned bytes;        // extent of the tensor from src on (bounded buffer resource)
  int nstride;           // floats between images
  int cstride;           // floats between channels (= Hs*Ws)
  int Hs, Ws;
  int chunks;            // channels / 16
  int mode;
};

#define CSF_MAX_SUB 5
struct CsfSub {          // one GEMM of a launch: the sub-problems share the B operand's sources and the geometry
  const float* A;        // [Mp][Kp] row-major, zero padded (rows to the block tile, K to 16)
  float* out;            // [B][out_ctot][HWo], already offset to the first output row
  int M, dil;
  int n_mtiles;
  int tile0;             // first logical block of the sub-problem
};

struct CsfGemmArgs {
  int Kp;
  int nseg;              // 1x1: concatenated segments; taps: ONE source, 9 pseudo-segments (tap-major K)
  int taps;              // 0 or 9
  CsfSeg seg[CSF_MAX_SEG];
  int nsub;              // e.g. the five dilations of an MSBlock (csf_res2net.py:207-211) in ONE launch
  CsfSub sub[CSF_MAX_SUB];
  int total_tiles;
  int ksplit, chunks_per_split;   // split-K: slice ks covers chunks [ks*cps, (ks+1)*cps) and writes plane ks
  long long split_stride;         // floats between the planes (summed by csf_combine_kernel in a fixed order)
  long long out_nstride;
  int Ho, Wo, HWo, Ntot; // Ntot = B * HWo
  int n_ntiles;
};

struct CsfZ {            // a coarser tensor added through bilinear up-sampling (gOctConv.py:96-98)
  const float* z;        // [B][ctot][Hz*Wz] offset to the first channel
  long long nstride;
  int ns;                // split-K planes of z (summed before the interpolation)
  long long split_stride;
  int Hz, Wz;
  float ry, rx;
};

struct CsfCombArgs {     // s = sum_k s_k + sum_i up(z_i); partial sums of s and s^2 per (image, group, slab) in fp64
  float* s;              // [ns][B][C][HW]; the result replaces plane 0
  int ns;
  long long split_stride;
  int B, C, H, W, HW, cpg, groups;
  int nz;
  CsfZ z[3];
  double* part;          // [B][C][2]: one block (and one partial) per channel plane
  int z_in_lds, z_floats;// the nz coarse planes of a block's channels fit in LDS (z_floats per channel)
  int step_x, step_y;    // CSN_BLOCK % W, CSN_BLOCK / W
};

struct CsfGnFinArgs {    // per (image, group): mean / rstd -> per (image, channel) scale and shift
  const double* part;
  int nslab, cpg, groups, C, HW, B;
  const float* gamma;
  const float* beta;
  float eps;
  float* scale;          // [B][C]
  float* shift;
};

struct CsfApplyArgs {    // y = prelu(s * scale[n][c] + shift[n][c]) in place
  float* s;
  const float* scale;
  const float* shift;
  const float* alpha;
  int C, HW;
  long long total;       // B*C*HW
};

struct CsfClsArgs {      // logits[n][p] = bias + sum_c w[c] * prelu(s[n][c][p] * scale + shift)
  const float* s;
  const float* scale;
  const float* shift;
  const float* alpha;
  const float* w;
  const float* bias;
  float* out;            // [B][HW]
  int C, HW, B;
};

struct CsfResizeArgs {   // F.interpolate(size, bilinear, align_corners=False) of [planes][Hi][Wi]
  const float* in;
  float* out;
  int planes, Hi, Wi, Ho, Wo;
  float ry, rx;
  int cpi;                  // channels per image: plane = n * cpi + c ...
  long long out_nstride;    // ... lands at out + n * out_nstride + c * Ho * Wo (channel slice of a wider tensor)
};

struct CsfPrepSeg { int k0, C, col0; };
struct CsfPrepArgs {     // weight image: dst[m][k] (zero padded) from arena rows
  const float* src;      // arena + offset of W[row 0][0]
  float* dst;
  int M, Mp, Kp, ld;     // ld: floats per source row
  int taps;              // 9: k = tap*Cp + c reads src[m*ld + c*9 + tap] (seg[0] = {Cp, C, 0}); 0: k in segment s reads
                         // src[m*ld + col0 + (k-k0)], columns past C are zero
  int nseg;
  CsfPrepSeg seg[CSF_MAX_SEG];
};

struct CsfBnActArgs {    // x = act(x * scale[c] + shift[c] (+ res)), scale / shift folded from the BatchNorm tensors in the kernel
  float* x;
  const float* res;
  const float* gamma; const float* beta; const float* mean; const float* var;
  float eps;
  int C, HW, relu;
  int chunks;            // blocks per (image, channel) plane
};
int csf_launch_bn_act(const CsfBnActArgs& a, int planes, void* stream);
bool csf_gemm_f32();                                                  // CSF_GEMM_F32=1 (A/B switch)
int csf_launch_gemm(const CsfGemmArgs& a, int mt, void* stream);     // mt: 16-row tiles per block (2, 4 or 8)
int csf_launch_combine(const CsfCombArgs& a, void* stream);
int csf_launch_gn_finalize(const CsfGnFinArgs& a, void* stream);
int csf_launch_apply(const CsfApplyArgs& a, void* stream);
int csf_launch_cls(const CsfClsArgs& a, void* stream);
int csf_launch_resize(const CsfResizeArgs& a, void* stream);
struct CsfPrepJobDev {   // device-resident job table: every weight image and parameter vector of a head in ONE launch
  long long src_off, dst_off;   // floats into the caller's arena / the library's packed buffer
  int M, Mp, Kp, ld, taps, nseg;
  CsfPrepSeg seg[CSF_MAX_SEG];
  int blk0, nblk;               // blocks [blk0, blk0 + nblk) of the launch belong to this job
};
int csf_launch_prep_all(const CsfPrepJobDev* jobs_dev, int njobs, int total_blocks, const float* arena, float* packed,
                        void* stream);
