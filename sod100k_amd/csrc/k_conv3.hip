// k_conv3.hip -- dense 3x3 convolutions of the gOctConv units that use them, and the MSBlock.
//
// Reference semantics (CSNet/model/csnet.py):
//   ILBlock: first block and the first block of every stride-2 stage use a 3x3 gOctaveCBR (33-40);
//   gOctaveConv.forward 664-726 (see k_goct_pw.hip for the branch mesh); padding = 1, no bias.
//   MSBlock.forward 141-149: Conv2dX100 3x3 with dilation d in {1,2,4,8,16}, padding d, -> cat -> BN
//   -> PReLU; dilations with 0 channels are absent.
//
// conv3x3_kernel computes ONE output branch: out = epi( sum_src conv3x3(stage(src)) + up2(zadd) ).
// stage() copies the (optionally 2x2-max-pooled) source tile plus a 1-pixel zero-padded halo for all
// gathered input channels into LDS once; every lane then owns PPL adjacent output pixels and walks the
// output channels in groups of 8 accumulators, reading its 3x(PPL+2) window from LDS per input channel
// (72 scalar weights per (channel, group) come through s_load).  The low->high path of a 2-branch 3x3
// unit is a first launch of the same kernel at low resolution without epilogue (raw partial sums).
#include "csn_kernels.h"

template <int TY, int TX, int PPL>
__global__ __launch_bounds__(TY* TX / PPL) void conv3x3_kernel(C3Args a) {
  CSN_DYN_SMEM(float, lds);
  constexpr int NT = TY * TX / PPL;
  constexpr int SY = TY + 2;
  constexpr int SX = TX + 2;
  constexpr int SXP = SX + 1;  // row pitch (odd -> rows start on different banks)
  constexpr int PLANE = SY * SXP;
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * TY, tx0 = blockIdx.x * TX;
  const int H = a.H, W = a.W;

  // ---- stage all gathered input channels: [cin][SY][SXP] ----
  {
    int cbase = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      const C3Src sr = a.src[s];
      const int tot = sr.C * SY * SX;
      for (int i = tid; i < tot; i += NT) {
        const int sx = i % SX;
        const int t = i / SX;
        const int sy = t % SY, ch = t / SY;
        const int y = ty0 - 1 + sy, x = tx0 - 1 + sx;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
          if (sr.shift == 0) {
            v = sr.ptr[(((int64_t)b * sr.C + ch) * H + y) * W + x];
          } else {
            const int Ws = W * 2;
            const float* __restrict__ p = sr.ptr + (((int64_t)b * sr.C + ch) * (H * 2) + 2 * y) * Ws + 2 * x;
            const float2 r0 = *reinterpret_cast<const float2*>(p);
            const float2 r1 = *reinterpret_cast<const float2*>(p + Ws);
            v = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
          }
        }
        lds[(cbase + ch) * PLANE + sy * SXP + sx] = v;
      }
      cbase += sr.C;
    }
  }
  __syncthreads();

  constexpr int LXN = TX / PPL;
  const int py = tid / LXN, px = (tid - py * LXN) * PPL;
  const int y = ty0 + py, x = tx0 + px;
  if (y >= H || x >= W) return;

  // bilinear x2 taps of the half-resolution addend (low->high partial sums)
  int zi[PPL][4];
  float zw[PPL][4];
  const int Hz = H >> 1, Wz = W >> 1;
  if (a.zadd != nullptr) {
    int y0, y1;
    float ly;
    csn_bilin(y, 0.5f, Hz, y0, y1, ly);
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      int x0, x1;
      float lx;
      csn_bilin(x + p, 0.5f, Wz, x0, x1, lx);
      zi[p][0] = y0 * Wz + x0; zi[p][1] = y0 * Wz + x1; zi[p][2] = y1 * Wz + x0; zi[p][3] = y1 * Wz + x1;
      zw[p][0] = (1.f - ly) * (1.f - lx); zw[p][1] = (1.f - ly) * lx; zw[p][2] = ly * (1.f - lx); zw[p][3] = ly * lx;
    }
  }

  const float* __restrict__ win0 = lds + py * SXP + px;  // top-left of this lane's window
  const int ngrp = (a.cout + 7) >> 3;
  for (int g = 0; g < ngrp; ++g) {
    float acc[8][PPL];
#pragma unroll
    for (int co = 0; co < 8; ++co)
#pragma unroll
      for (int p = 0; p < PPL; ++p) acc[co][p] = 0.f;
    csn_cfp wg = csn_const(a.w) + (int64_t)g * a.cin * 72;
    for (int ci = 0; ci < a.cin; ++ci) {
      float win[3][PPL + 2];
      const float* __restrict__ wp = win0 + ci * PLANE;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < PPL + 2; ++c) win[r][c] = wp[r * SXP + c];
      csn_cfp wc = wg + ci * 72;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int co = 0; co < 8; ++co) {
          const float wv = wc[t * 8 + co];
#pragma unroll
          for (int p = 0; p < PPL; ++p) acc[co][p] = fmaf(wv, win[t / 3][t % 3 + p], acc[co][p]);
        }
      }
    }
    // epilogue for the 8 channels of this group
#pragma unroll
    for (int co = 0; co < 8; ++co) {
      const int oc = g * 8 + co;
      if (oc < a.cout) {
        float o[PPL];
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          float v = acc[co][p];
          if (a.zadd != nullptr) {
            const float* __restrict__ z = a.zadd + ((int64_t)b * a.cout + oc) * Hz * Wz;
            v += zw[p][0] * z[zi[p][0]] + zw[p][1] * z[zi[p][1]] + zw[p][2] * z[zi[p][2]] + zw[p][3] * z[zi[p][3]];
          }
          if (a.scale != nullptr) v = csn_epi(v, csn_const(a.scale)[oc], csn_const(a.shift)[oc], csn_const(a.alpha)[oc]);
          o[p] = v;
        }
        float* __restrict__ q = a.out + (((int64_t)b * a.cout + oc) * H + y) * W + x;
        if (PPL == 2 && x + 1 < W) {
          *reinterpret_cast<float2*>(q) = make_float2(o[0], o[PPL - 1]);
        } else {
#pragma unroll
          for (int p = 0; p < PPL; ++p)
            if (x + p < W) q[p] = o[p];
        }
      }
    }
  }
}

template <int TY, int TX>
static size_t c3_lds_bytes(int cin) {
  return (size_t)cin * (TY + 2) * (TX + 3) * sizeof(float);
}

int csn_launch_c3(const C3Args& a, void* stream) {
  // prefer the largest tile whose staged input fits in 64 KiB of LDS (>= 2 blocks per CU)
  const size_t budget = 64 * 1024;
  const size_t need_a = c3_lds_bytes<16, 32>(a.cin);
  const size_t need_b = c3_lds_bytes<8, 32>(a.cin);
  if (need_a <= budget) {
    const dim3 grid((a.W + 31) / 32, (a.H + 15) / 16, a.B);
    CSN_LAUNCH((conv3x3_kernel<16, 32, 2>), grid, dim3(256), need_a, stream, a);
  } else if (need_b <= budget) {
    const dim3 grid((a.W + 31) / 32, (a.H + 7) / 8, a.B);
    CSN_LAUNCH((conv3x3_kernel<8, 32, 1>), grid, dim3(256), need_b, stream, a);
  } else {
    const size_t need = c3_lds_bytes<8, 16>(a.cin);
    if (need > 160 * 1024) return -2;
    const dim3 grid((a.W + 15) / 16, (a.H + 7) / 8, a.B);
    CSN_LAUNCH((conv3x3_kernel<8, 16, 1>), grid, dim3(128), need, stream, a);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ MSBlock
// Direct form: one output pixel per lane, taps read straight from global memory (L1/L2 resident:
// the three MSBlock inputs are 17x112^2, 38x56^2 and 32x28^2 per image) with zero padding by
// predication.  At most 8 accumulators live per lane; 9 % of the network's FLOPs, 1.3 % of its bytes.
__global__ __launch_bounds__(CSN_BLOCK) void msblock_kernel(MsArgs a) {
  const int H = a.H, W = a.W;
  const int64_t hw = (int64_t)H * W;
  const int64_t p = (int64_t)blockIdx.x * CSN_BLOCK + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= hw) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  const float* __restrict__ ip = a.in + (int64_t)b * a.cin * hw;
  float* __restrict__ op = a.out + (int64_t)b * a.cout * hw + p;
  for (int d = 0; d < 5; ++d) {
    const int nco = a.dch[d];
    if (nco == 0) continue;
    const int dil = 1 << d;
    int off[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + (t / 3 - 1) * dil, xx = x + (t % 3 - 1) * dil;
      ok[t] = yy >= 0 && yy < H && xx >= 0 && xx < W;
      off[t] = ok[t] ? yy * W + xx : 0;
    }
    const int ngrp = (nco + 7) >> 3;
    for (int g = 0; g < ngrp; ++g) {
      float acc[8];
#pragma unroll
      for (int co = 0; co < 8; ++co) acc[co] = 0.f;
      csn_cfp wg = csn_const(a.w[d]) + (int64_t)g * a.cin * 72;
      for (int ci = 0; ci < a.cin; ++ci) {
        const float* __restrict__ pl = ip + ci * hw;
        csn_cfp wc = wg + ci * 72;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float v = ok[t] ? pl[off[t]] : 0.f;
#pragma unroll
          for (int co = 0; co < 8; ++co) acc[co] = fmaf(wc[t * 8 + co], v, acc[co]);
        }
      }
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        const int lc = g * 8 + co;
        if (lc < nco) {
          const int oc = a.cobase[d] + lc;
          op[oc * hw] = csn_epi(acc[co], csn_const(a.scale)[oc], csn_const(a.shift)[oc], csn_const(a.alpha)[oc]);
        }
      }
    }
  }
}

int csn_launch_ms(const MsArgs& a, void* stream) {
  const int64_t hw = (int64_t)a.H * a.W;
  const dim3 grid((unsigned)((hw + CSN_BLOCK - 1) / CSN_BLOCK), a.B);
  CSN_LAUNCH(msblock_kernel, grid, dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}

int csn_kernels_init(void) {
#ifndef CSN_CPU_EMU
  // allow the small-tile 3x3 instantiation to use the whole 160 KiB LDS of a CDNA4 CU
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<8, 16, 1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
#endif
  return 0;
}
