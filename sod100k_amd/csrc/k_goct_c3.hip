// k_goct_c3.hip -- gOctConv 3x3 passes as an LDS-tiled implicit GEMM.
//
// Reference semantics (CSNet/model/csnet.py:664-726 with kernel_size 3, padding 1): for output branch j
//     y_j = conv3x3(x_j) [+ conv3x3(max_pool2(x_{j-1}))] [+ bilinear_up2(z)],   z = conv3x3(x_{j+1}) (own launch)
// followed by BN + PReLU (gOctaveCBR 778-792).  goct_pw_kernel evaluates such a pass by gathering the 9 taps of
// every input channel per pixel from global memory (162 .. 459 loads per pixel; 1.1 of the 3.8 ms forward).
//
// Here a block owns an 8 x 32 output tile and stages the (8 + 2) x (32 + 2) input tile of 16 channels at a time
// in LDS (max-pooling on the way in for the high -> low slice, zero padding by the bounds check), so a value is
// fetched from global memory 1.33 times instead of 9.  The weight columns of the same 16 channels (<= 32 rows x 144)
// are staged next to it, so a block needs 41 KB of LDS whatever the size of the unit's weight image.  The contraction keeps goct_pw_kernel's MFMA layout
// (v_mfma_f32_16x16x4_f32, A = W[row][k] from the LDS weight image, k = 9 * channel + tap) but reads the B operand
// x[k][pixel] straight from the tile: entry k of lane (k sub-index, pixel) is tile[ch][row + dy][col + dx] -- no
// gathered panel is materialised.  The optional bilinear z slice (identity weight block) goes through a
// per-wave panel exactly as in goct_pw_kernel, and so does the epilogue.
#include "pw_gather.h"

#define C3_TX 32
#define C3_TY 8
#define C3_CC 16                       // channels per staged chunk
#define C3_TP 36                       // tile row pitch (34 columns used)
#define C3_PLANE ((C3_TY + 2) * C3_TP) // floats per channel of the tile
#define C3_TILE (C3_CC * C3_PLANE)     // 5760 floats = 22.5 KB
#define C3_WP (9 * C3_CC + 2)           // pitch of the staged weight chunk: 146 = 2 (mod 4), conflict-free A reads
#define C3_WCH (32 * C3_WP)            // 32 rows x 144 columns = 18.3 KB

__device__ __forceinline__ int c3_entry_off(int kk) {   // kk = 9 * ch + tap inside the chunk -> tile offset
  const int ch = kk / 9, t = kk - 9 * ch;
  const int dy = (t * 11) >> 5, dx = t - 3 * dy;
  return ch * C3_PLANE + dy * C3_TP + dx;
}

// WCH = false: the pass's whole weight image is staged once per block (small images: no per-chunk copy);
// WCH = true: only the current chunk's columns live in LDS (large images: 41 KB per block, 3 blocks per CU).
template <bool RAW, bool WCH>
__global__ __launch_bounds__(CSN_BLOCK, 3) void goct_c3_kernel(PwArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  const CSN_CONST_AS PwArgs* a = CSN_KERNARG(PwArgs, a_byval);
  PwPassP ps = &a->pass[0];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (!WCH) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a->wimg);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int n4 = a->wimg_floats >> 2;
    for (int i = tid; i < n4; i += CSN_BLOCK) dst[i] = src[i];
  }
  float* wch = lds;                            // WCH: [32][C3_WP] weight columns of the current chunk
  float* tile = lds + (WCH ? C3_WCH : a->wimg_floats);
  const int wp = WCH ? C3_WP : ps->w_stride;   // row pitch of the A operand's source          // [C3_CC][C3_TY + 2][C3_TP]; also the waves' panels / epilogue scratch
  const int Hr = a->H0, Wr = a->W0;
  const int tiles_x = (Wr + C3_TX - 1) / C3_TX, tiles_y = (Hr + C3_TY - 1) / C3_TY;
  const int tiles_xy = tiles_x * tiles_y;
  const int ntiles = tiles_xy * a->B;
  const int nrows = ps->nrows, stride = ps->w_stride;
  const float* __restrict__ wg0 = a->wimg + ps->w_off;   // the pass's rows in the (global) weight image
  // tap slices first (TAPS / POOL2_TAPS), an optional non-tap slice (bilinear z) last
  int ntap = 0;
  for (int s = 0; s < ps->nsrc; ++s)
    if (ps->src[s].mode == PW_TAPS || ps->src[s].mode == PW_POOL2_TAPS) ntap = s + 1;
  // XCD-aware tile order (see goct_pw_kernel)
  const int nslot = gridDim.x >> 3;
  const int chunk = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min((xcd + 1) * chunk, ntiles);
  // this lane's pixel inside the tile: wave w owns tile rows 2w, 2w+1; sub-group s = 16 pixels of one row
  const int kq = lane >> 4, pxi = lane & 15;
  for (int tl = xcd * chunk + (blockIdx.x >> 3); tl < tend; tl += nslot) {
    const int b = tl / tiles_xy;
    const int txy = tl - b * tiles_xy;
    const int y0 = (txy / tiles_x) * C3_TY, x0 = (txy % tiles_x) * C3_TX;
    // lane-as-pixel coordinates (epilogue, z gather): pixel p = lane of the wave's 2 x 32 strip
    const int py_ = y0 + 2 * wave + (lane >> 5), px_ = x0 + (lane & 31);
    const bool valid = py_ < Hr && px_ < Wr;
    const int gy = min(py_, Hr - 1), gx = min(px_, Wr - 1);
    const unsigned ovoff = (unsigned)(gy * Wr + gx) * 4u;
    for (int row0 = 0; row0 < nrows; row0 += 32) {
      const bool two = nrows - row0 > 16;
      csn_f4 acc[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][s][i] = 0.f;
      int kcol = 0;
      for (int s = 0; s < ntap; ++s) {
        const int C = ps->src[s].C, Ctot = ps->src[s].Ctot;
        const bool pooled = ps->src[s].mode == PW_POOL2_TAPS;
        const int Hs = pooled ? Hr * 2 : Hr, Ws = pooled ? Wr * 2 : Wr;
        const float* __restrict__ sp = ps->src[s].ptr + (int64_t)b * Ctot * Hs * Ws;
        for (int c_lo = 0; c_lo < C; c_lo += C3_CC) {
          const int nc = min(C3_CC, C - c_lo);
          __syncthreads();   // previous chunk / panels fully consumed
          // ---- stage nc channels x 10 rows x 34 columns (zero padded; 2x2 max on the way in when pooled)
          const int nel = nc * (C3_TY + 2) * 34;
          for (int e = tid; e < nel; e += CSN_BLOCK) {
            const int ch = e / ((C3_TY + 2) * 34);
            const int r = e - ch * ((C3_TY + 2) * 34);
            const int ty = r / 34, tx = r - ty * 34;
            const int yy = y0 - 1 + ty, xx = x0 - 1 + tx;
            const bool in = yy >= 0 && yy < Hr && xx >= 0 && xx < Wr;
            const int yc = min(max(yy, 0), Hr - 1), xc = min(max(xx, 0), Wr - 1);
            float v;
            if (pooled) {
              const float* q = sp + ((int64_t)(c_lo + ch) * Hs + 2 * yc) * Ws + 2 * xc;
              const float2 t0 = *reinterpret_cast<const float2*>(q), t1 = *reinterpret_cast<const float2*>(q + Ws);
              v = fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
            } else {
              v = sp[((int64_t)(c_lo + ch) * Hs + yc) * Ws + xc];
            }
            tile[ch * C3_PLANE + ty * C3_TP + tx] = in ? v : 0.f;
          }
          const int kn = 9 * nc;
          if (WCH) {   // weight columns [kcol + 9 c_lo, + kn) of rows row0 .. row0 + 31 (rows past nrows are zero in the image)
            const int nr = two ? 32 : 16;
            const float* __restrict__ wg = wg0 + (int64_t)row0 * stride + kcol + 9 * c_lo;
            for (int e = tid; e < nr * kn; e += CSN_BLOCK) {
              const int r = e / kn, k = e - r * kn;
              wch[r * C3_WP + k] = wg[(int64_t)r * stride + k];
            }
          }
          __syncthreads();
          // ---- contract the chunk's 9 * nc entries straight from the tile
          const float* wt = WCH ? wch : lds + ps->w_off + row0 * stride + kcol + 9 * c_lo;
          for (int k0 = 0; k0 < kn; k0 += 4) {
            const int kk = k0 + kq;
            const bool kin = kk < kn;
            const int eo = c3_entry_off(kin ? kk : 0);
#ifdef CSN_CPU_EMU
            // D[row][px] += sum_u W[row][k0 + u] * x[k0 + u][px]; this lane holds rows (lane>>4)*4 + i, px = lane & 15
            for (int t = 0; t < (two ? 2 : 1); ++t)
              for (int sg = 0; sg < 4; ++sg) {
                const int trow = 2 * wave + (sg >> 1), tcol = 16 * (sg & 1) + pxi;
                for (int i = 0; i < 4; ++i) {
                  const int row = kq * 4 + i;
                  float acc_ = acc[t][sg][i];
                  for (int u = 0; u < 4 && k0 + u < kn; ++u)
                    acc_ = fmaf(wt[(16 * t + row) * wp + k0 + u], tile[c3_entry_off(k0 + u) + trow * C3_TP + tcol], acc_);
                  acc[t][sg][i] = acc_;
                }
              }
            (void)eo;
#else
            // A = W[16 t + (lane & 15)][k0 + kq] (0 past the chunk: the image columns beyond belong to other slices)
            const float a0 = kin ? wt[pxi * wp + kk] : 0.f;
            const float a1 = (two && kin) ? wt[(16 + pxi) * wp + kk] : 0.f;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
              const float bv = tile[eo + (2 * wave + (sg >> 1)) * C3_TP + 16 * (sg & 1) + pxi];
              acc[0][sg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][sg], 0, 0, 0);
              if (two) acc[1][sg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][sg], 0, 0, 0);
            }
#endif
          }
        }
        kcol += 9 * C;
      }
      __syncthreads();   // tile consumed: the waves' private panels live in the same LDS from here on
      float* xb = tile + wave * (PW_KC * PW_XP);
      // ---- non-tap slice (bilinear z through an identity block): goct_pw_kernel's panel path
      for (int s = ntap; s < ps->nsrc; ++s) {
        const int K = ps->src[s].K;
        for (int kc = 0; kc < K; kc += PW_KC) {
          const int n = min(PW_KC, K - kc), n4 = (n + 3) & ~3;
          __syncthreads();   // previous step (and, WCH, its weight chunk) consumed by every wave
          if (WCH) {
            const int nr = two ? 32 : 16;
            const float* __restrict__ wg = wg0 + (int64_t)row0 * stride + kcol + kc;
            for (int e = tid; e < nr * n; e += CSN_BLOCK) {
              const int r = e / n, k = e - r * n;
              wch[r * C3_WP + k] = wg[(int64_t)r * stride + k];
            }
          }
          __syncthreads();
          pw_gather_slice<PW_XP>(ps, s, kc, kc + n, xb + lane, PW_KC, b, gy, gx, Hr, Wr);
          for (int k = n; k < n4; ++k) xb[k * PW_XP + lane] = 0.f;
          CSN_WAVE_SYNC();
          // lane-as-pixel panel order = the wave's 2 x 32 strip: sub-group sg = pixels 16 sg .. 16 sg + 15
          const float* wt = WCH ? wch : lds + ps->w_off + row0 * stride + kcol + kc;
          for (int k0 = 0; k0 < n4; k0 += 4) {
#ifdef CSN_CPU_EMU
            for (int t = 0; t < (two ? 2 : 1); ++t)
              for (int sg = 0; sg < 4; ++sg)
                for (int i = 0; i < 4; ++i) {
                  const int row = kq * 4 + i;
                  float acc_ = acc[t][sg][i];
                  for (int u = 0; u < 4; ++u)
                    acc_ = fmaf((k0 + u < n) ? wt[(16 * t + row) * wp + k0 + u] : 0.f, xb[(k0 + u) * PW_XP + 16 * sg + pxi], acc_);
                  acc[t][sg][i] = acc_;
                }
#else
            const bool kin = k0 + kq < n;
            const float a0 = kin ? wt[pxi * wp + k0 + kq] : 0.f;
            const float a1 = (two && kin) ? wt[(16 + pxi) * wp + k0 + kq] : 0.f;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
              const float bv = xb[(k0 + kq) * PW_XP + 16 * sg + pxi];
              acc[0][sg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][sg], 0, 0, 0);
              if (two) acc[1][sg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][sg], 0, 0, 0);
            }
#endif
          }
        }
        kcol += K;
      }
      // ---- epilogue: transpose through the wave's scratch, folded BN + PReLU, 128-byte row segments
      const unsigned cs4 = (unsigned)(Hr * Wr) * 4u;
      const csn_buf ob = csn_make_buf(ps->out + (int64_t)b * ps->out_ctot * (Hr * Wr));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
        CSN_WAVE_SYNC();
#pragma unroll
        for (int sg = 0; sg < 4; ++sg)
#pragma unroll
          for (int i = 0; i < 4; ++i) xb[(kq * 4 + i) * PW_EP + 16 * sg + pxi] = acc[t][sg][i];
        CSN_WAVE_SYNC();
        const int rbase = row0 + 16 * t;
        const int rn = min(16, nrows - rbase);
        csn_cfp scale = csn_const(ps->scale) + rbase, shift = csn_const(ps->shift) + rbase, alpha = csn_const(ps->alpha) + rbase;
#pragma unroll 4
        for (int rr = 0; rr < rn; ++rr) {
          const float val = RAW ? xb[rr * PW_EP + lane] : csn_epi(xb[rr * PW_EP + lane], scale[rr], shift[rr], alpha[rr]);
          if (valid) csn_st1(ob, ovoff, (unsigned)(rbase + rr) * cs4, val);
        }
      }
    }
  }
}

// true when the launch is one pass made of 3x3 tap slices (dilation 1) plus at most one trailing non-tap slice
bool csn_c3_eligible(const PwArgs& a) {
  if (a.npass != 1 || a.pass[0].red_w || a.pass[0].r != 0) return false;   // the kernel walks the launch resolution
  const PwPass& ps = a.pass[0];
  int ntap = 0;
  bool tail = false;
  for (int s = 0; s < ps.nsrc; ++s) {
    const int m = ps.src[s].mode;
    if (m == PW_TAPS || m == PW_POOL2_TAPS) {
      if (tail || ps.src[s].dil != 1) return false;
      ++ntap;
    } else {
      if (m != PW_UP2) return false;
      tail = true;
    }
  }
  return ntap > 0;
}

int csn_launch_c3(const PwArgs& a, int raw, void* stream) {
  const int tiles = ((a.W0 + C3_TX - 1) / C3_TX) * ((a.H0 + C3_TY - 1) / C3_TY) * a.B;
  const int nblk = tiles < PW_MAX_GRID ? tiles : PW_MAX_GRID;
  const dim3 grid((nblk + 7) & ~7);
  // small weight images are staged whole (stage0.0 / 2.0 / 4.0: measured faster), large ones chunk by chunk
  const bool wch = (size_t)a.wimg_floats * sizeof(float) > 48 * 1024;
  const size_t lds = (size_t)((wch ? C3_WCH : a.wimg_floats) + C3_TILE) * sizeof(float);
#ifndef CSN_CPU_EMU
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[4] = {reinterpret_cast<const void*>(&goct_c3_kernel<false, false>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<false, true>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<true, false>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<true, true>)};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    attr_done = true;
  }
#endif
  if (raw && wch) CSN_LAUNCH((goct_c3_kernel<true, true>), grid, dim3(CSN_BLOCK), lds, stream, a);
  else if (raw) CSN_LAUNCH((goct_c3_kernel<true, false>), grid, dim3(CSN_BLOCK), lds, stream, a);
  else if (wch) CSN_LAUNCH((goct_c3_kernel<false, true>), grid, dim3(CSN_BLOCK), lds, stream, a);
  else CSN_LAUNCH((goct_c3_kernel<false, false>), grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
