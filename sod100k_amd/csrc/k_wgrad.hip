// k_wgrad.hip -- weight gradient of a gOctConv / MSBlock / cls contraction pass.
//
// Forward (k_goct_pw.hip) evaluates every output branch as ONE contraction  y[row][px] = sum_k W[row][k] g[k][px]
// over the gathered vector g = [own ; max-pool(higher res) ; bilinear(lower res)] (x9 taps for 3x3), which
// restates gOctaveConv.forward (csnet.py:664-726).  Autograd's weight gradient of that expression is
//     dW[row][k] = sum_{n, px} dz[row][px] * g[k][px]
// (max-pool and bilinear are functions of the INPUT only, so they stay inside g), i.e. a GEMM whose reduction
// dimension is the pixel index (up to 3.2 M at batch 64) and whose output is tiny (<= 80 x 464).
//
// MI355X mapping.  A work item is a group of 64 consecutive pixels of one image plane.  The four waves of a block
// share the group's dz panel in LDS (dzp[row][64 px]) and each wave owns one 16-wide chunk of k per sweep: it
// re-uses the forward's per-lane gather (pw_gather.h) to build its private panel xp[16 k][64 px] and issues
// v_mfma_f32_16x16x4_f32 with the PIXEL index as the MFMA reduction dimension:
//     A[i = row][kk = px] = dzp[16 t + (lane & 15)][4 s + (lane >> 4)]
//     B[kk = px][j = k]   = xp [      (lane & 15)][4 s + (lane >> 4)]      (pitch 68 = 4 mod 32: conflict-free)
// so the accumulators D[row][k] (4 VGPRs per 16x16 tile) stay in registers across ALL the groups of the block.
// Blocks write their partial dW to a scratch buffer; wgrad_reduce_kernel sums the partials in a fixed order
// (fp64, deterministic) and scatters the columns back to the reference's [Cout][Cin][k][k] weight layout.
#include "pw_gather.h"

#define WG_P 68          // panel pitch (floats)
#define WG_MAX_NT 5      // up to 80 rows per pass

typedef const CSN_CONST_AS WgArgs* WgArgsP;

template <int NT>
__device__ __forceinline__ void wg_mma(const float* dzp, const float* xp, int lane, csn_f4 (&acc)[WG_MAX_NT]) {
#ifdef CSN_CPU_EMU
  const int j = lane & 15;
  for (int s = 0; s < 16; ++s)
    for (int t = 0; t < NT; ++t)
      for (int reg = 0; reg < 4; ++reg) {
        const int i = (lane >> 4) * 4 + reg;
        float v = acc[t][reg];
        for (int kk = 0; kk < 4; ++kk) v = fmaf(dzp[(16 * t + i) * WG_P + 4 * s + kk], xp[j * WG_P + 4 * s + kk], v);
        acc[t][reg] = v;
      }
#else
  const int o = (lane & 15) * WG_P + (lane >> 4);
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    const float bv = xp[o + 4 * s];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float av = dzp[16 * t * WG_P + o + 4 * s];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
    }
  }
#endif
}

__global__ __launch_bounds__(CSN_BLOCK) void goct_wgrad_kernel(WgArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  WgArgsP a = CSN_KERNARG(WgArgs, a_byval);
  PwPassP ps = &a->ps;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rows16 = a->rows16, k16 = a->k16, nrows = ps->nrows;
  const int Hr = a->Hr, Wr = a->Wr, HW = Hr * Wr;
  float* dzp = lds;                                       // [rows16][WG_P], shared by the block
  float* xp = lds + rows16 * WG_P + wave * (16 * WG_P);   // this wave's x[16 k][64 px] panel
  for (int i = tid; i < rows16 * WG_P; i += CSN_BLOCK) dzp[i] = 0.f;   // rows nrows..rows16 stay zero
  const int cin = ps->cin;
  const int c1 = ps->src[0].K, c2 = c1 + ps->src[1].K;
  const int nt = rows16 >> 4;
  const int nsweep = (k16 / 16 + 3) >> 2;
  for (int sw = 0; sw < nsweep; ++sw) {
    const int kc = (4 * sw + wave) * 16;
    const bool active = kc < k16;
    csn_f4 acc[WG_MAX_NT];
#pragma unroll
    for (int t = 0; t < WG_MAX_NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    for (int g = blockIdx.x; g < a->ngroups; g += a->nblk) {
      const int b = g / a->gpp;
      const int p = (g - b * a->gpp) * 64 + lane;
      const bool valid = p < HW;
      const int pc = valid ? p : HW - 1;
      const int y = pc / Wr, x = pc - y * Wr;
      __syncthreads();   // previous group's dz panel fully consumed
      {
        const float* ap = a->a + (int64_t)b * a->a_ctot * HW + pc;
        for (int r = wave; r < nrows; r += 4) {
          const float v = ap[(int64_t)r * HW];
          dzp[r * WG_P + lane] = valid ? v : 0.f;   // pixels past the plane contribute nothing
        }
      }
      __syncthreads();
      if (active) {
        const int kend = min(kc + 16, cin);
        CSN_WAVE_SYNC();
        if (kc < c1) pw_gather_slice<WG_P>(ps, 0, kc, min(kend, c1), xp + lane, 16, b, y, x, Hr, Wr);
        if (max(kc, c1) < min(kend, c2)) {
          const int r0 = max(kc, c1) - kc;
          pw_gather_slice<WG_P>(ps, 1, max(kc, c1) - c1, min(kend, c2) - c1, xp + r0 * WG_P + lane, 16 - r0, b, y, x, Hr, Wr);
        }
        if (max(kc, c2) < min(kend, cin)) {
          const int r0 = max(kc, c2) - kc;
          pw_gather_slice<WG_P>(ps, 2, max(kc, c2) - c2, min(kend, cin) - c2, xp + r0 * WG_P + lane, 16 - r0, b, y, x, Hr, Wr);
        }
        for (int k = max(cin, kc); k < kc + 16; ++k) xp[(k - kc) * WG_P + lane] = 0.f;
        CSN_WAVE_SYNC();
        switch (nt) {
          case 1: wg_mma<1>(dzp, xp, lane, acc); break;
          case 2: wg_mma<2>(dzp, xp, lane, acc); break;
          case 3: wg_mma<3>(dzp, xp, lane, acc); break;
          case 4: wg_mma<4>(dzp, xp, lane, acc); break;
          default: wg_mma<5>(dzp, xp, lane, acc); break;
        }
      }
    }
    if (active) {
      float* out = a->partial + (int64_t)blockIdx.x * rows16 * k16 + kc + (lane & 15);
#pragma unroll
      for (int t = 0; t < WG_MAX_NT; ++t)
        if (t < nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) out[(int64_t)(16 * t + (lane >> 4) * 4 + i) * k16] = acc[t][i];
    }
  }
}

// grad[dst + r*ld + c] = scale * sum_blk partial[blk][r][col + c]  for every block of weight columns.
__global__ __launch_bounds__(CSN_BLOCK) void wgrad_reduce_kernel(WgReduceArgs a) {
  const int e = blockIdx.x * CSN_BLOCK + threadIdx.x;
  if (e >= a.nrows * a.K) return;
  const int r = e / a.K, k = e - r * a.K;
  int bi = -1;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q < a.nblocks && k >= a.blk[q].col && k < a.blk[q].col + a.blk[q].ncol) bi = q;
  if (bi < 0) return;   // identity columns (already convolved partial sums) carry no parameter
  double s = 0.0;
  const float* p = a.partial + (int64_t)r * a.k16 + k;
  const int64_t stride = (int64_t)a.rows16 * a.k16;
  for (int b = 0; b < a.nblk; ++b) s += (double)p[b * stride];
  a.grad[a.blk[bi].dst + (int64_t)r * a.blk[bi].ld + (k - a.blk[bi].col)] = (float)((double)a.blk[bi].scale * s);
}

int csn_launch_wgrad(const WgArgs& a, void* stream) {
  if (a.rows16 > 16 * WG_MAX_NT) return -1;
  const size_t lds = ((size_t)a.rows16 * WG_P + 4 * 16 * WG_P) * sizeof(float);
  CSN_LAUNCH(goct_wgrad_kernel, dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}

int csn_launch_wgrad_reduce(const WgReduceArgs& a, void* stream) {
  const int n = a.nrows * a.K;
  CSN_LAUNCH(wgrad_reduce_kernel, dim3((n + CSN_BLOCK - 1) / CSN_BLOCK), dim3(CSN_BLOCK), 0, stream, a);
  return (int)hipGetLastError();
}
