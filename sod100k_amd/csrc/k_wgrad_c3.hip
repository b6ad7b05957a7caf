// k_wgrad_c3.hip -- weight gradient of a 3x3 gOctConv pass from LDS tiles (the counterpart of k_goct_c3.hip).
//
//     dW[row][9 ch + t] = sum_{n, p} dz[row][p] * x_ch[p + off(t)]          (autograd of csnet.py:664-726 with k = 3,
//                                                                            x = own slice or its 2x2 max-pool)
// The per-pixel kernel (k_wgrad.hip) gathers the nine taps of every channel for every pixel -- nine loads per input value --
// and synchronises the block per 64-pixel group.  Here a block owns 8 x 32 pixel tiles and walks the input channels 16 at a
// time: the (8 + 2) x (32 + 2) tile of the chunk (max-pooled on the way in for the high -> low slice) and the tile's dz rows go
// to LDS once, and v_mfma_f32_16x16x4_f32 contracts over the PIXELS:
//     A[i = row][kk = pixel]      = dz[16 t + (lane & 15)][64 wave + 4 s + (lane >> 4)]            (pitch 260 = 4 mod 32)
//     B[kk = pixel][j = channel]  = tile[lane & 15][y + dy][x + dx + (lane >> 4)]                   (plane 388 = 4 mod 32)
// both conflict-free single ds_read_b32 per lane; the 2 x 9 accumulator tiles D[row][channel] of the nine taps stay in VGPRs
// over ALL tiles of the block.  Per chunk the four waves' accumulators are added through LDS and written to the block's slice
// of the partial buffer in the per-pixel kernel's column order (k = 9 ch + t), so wgrad_reduce_kernel is shared.
#include "pw_gather.h"

#define W3_TX 32
#define W3_TY 8
#define W3_CC 16
#define W3_TP 36                        // tile row pitch (34 columns used)
#define W3_PLANE 388                    // floats per channel: 10 x 36 + 28, == 4 (mod 32)
#define W3_TILE (W3_CC * W3_PLANE)
#define W3_DZP 260                      // dz row pitch: 256 pixels + 4, == 4 (mod 32)
#define W3_MAX_ROWS 32

typedef const CSN_CONST_AS WgArgs* W3ArgsP;

template <typename AT>
__global__ __launch_bounds__(CSN_BLOCK, 2) void goct_wgrad_c3_kernel(WgArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  constexpr unsigned E = (unsigned)sizeof(AT);
  W3ArgsP a = CSN_KERNARG(WgArgs, a_byval);
  PwPassP ps = &a->ps;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CSN_CPU_EMU
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int Hr = a->Hr, Wr = a->Wr, HW = Hr * Wr;
  const int nrows = ps->nrows, rows16 = a->rows16, k16 = a->k16;
  const bool two = rows16 > 16;
  float* tile = lds;                         // [W3_CC][W3_PLANE]
  float* dzt = lds + W3_TILE;                // [rows16][W3_DZP]; after the tile loop: the waves' combine scratch
  const int tiles_x = (Wr + W3_TX - 1) / W3_TX, tiles_y = (Hr + W3_TY - 1) / W3_TY;
  const int tiles_xy = tiles_x * tiles_y;
  const int ntiles = tiles_xy * a->B;
  const int kq = lane >> 4, pxi = lane & 15;
  const int r8 = tid >> 5, lx = tid & 31;
  const int hch = tid / (W3_TY + 2), hrow = tid - hch * (W3_TY + 2);
  const unsigned OOB = 0x80000000u;
  float* out = a->partial + (int64_t)blockIdx.x * rows16 * k16;
  int col0 = 0;                              // first weight column of the current slice
  for (int s = 0; s < ps->nsrc; ++s) {
    const int C = ps->src[s].C;
    const bool pooled = ps->src[s].mode == PW_POOL2_TAPS;
    for (int c_lo = 0; c_lo < C; c_lo += W3_CC) {
      const int nc = min(W3_CC, C - c_lo);
      csn_f4 acc[2][9];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][q][i] = 0.f;
      for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int b = tl / tiles_xy;
        const int txy = tl - b * tiles_xy;
        const int y0 = (txy / tiles_x) * W3_TY, x0 = (txy % tiles_x) * W3_TX;
        __syncthreads();                     // the previous tile is consumed
        // ---- stage the input tile of channels c_lo .. c_lo + 15 (zero padding = out-of-range buffer offsets)
        if (!pooled) {
          const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[s].ptr) + (int64_t)b * ps->src[s].Ctot * HW,
                                            (unsigned)(ps->src[s].Ctot * HW) * E);
          const unsigned HW4 = (unsigned)HW * E;
          const int xx = x0 + lx;
          const unsigned colo = xx < Wr ? (unsigned)xx * E : OOB;
          float pf[22];
#pragma unroll
          for (int i = 0; i < 20; ++i) {
            const int ch = c_lo + 2 * r8 + i / 10, yy = y0 - 1 + (i % 10);
            const bool ok = ch < C && yy >= 0 && yy < Hr;
            pf[i] = csn_bufacc<AT>::ld1(rb, ok ? (unsigned)ch * HW4 + (unsigned)(yy * Wr) * E + colo : OOB, 0u);
          }
          {
            const int ch = c_lo + hch, yy = y0 - 1 + hrow;
            const bool ok = tid < W3_CC * (W3_TY + 2) && ch < C && yy >= 0 && yy < Hr;
            const unsigned ro = (unsigned)ch * HW4 + (unsigned)(yy * Wr) * E;
            pf[20] = csn_bufacc<AT>::ld1(rb, ok && x0 > 0 ? ro + (unsigned)(x0 - 1) * E : OOB, 0u);
            pf[21] = csn_bufacc<AT>::ld1(rb, ok && x0 + W3_TX < Wr ? ro + (unsigned)(x0 + W3_TX) * E : OOB, 0u);
          }
#pragma unroll
          for (int i = 0; i < 20; ++i) tile[(2 * r8 + i / 10) * W3_PLANE + (i % 10) * W3_TP + 1 + lx] = pf[i];
          if (tid < W3_CC * (W3_TY + 2)) {
            tile[hch * W3_PLANE + hrow * W3_TP] = pf[20];
            tile[hch * W3_PLANE + hrow * W3_TP + W3_TX + 1] = pf[21];
          }
        } else {
          const int Hs = 2 * Hr, Ws = 2 * Wr;
          const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[s].ptr) + (int64_t)b * ps->src[s].Ctot * (Hs * Ws),
                                            (unsigned)(ps->src[s].Ctot * Hs * Ws) * E);
          const unsigned HW4 = (unsigned)(Hs * Ws) * E, P4 = (unsigned)Ws * E;
          auto pool = [&](int ch, int yy, int xx) {
            const bool ok = ch < C && yy >= 0 && yy < Hr && xx >= 0 && xx < Wr;
            const unsigned o = ok ? (unsigned)ch * HW4 + (unsigned)(2 * yy) * P4 + (unsigned)xx * (2u * E) : OOB;
            const float2 t0 = csn_bufacc<AT>::ld2(rb, o, 0u), t1 = csn_bufacc<AT>::ld2(rb, ok ? o + P4 : OOB, 0u);
            return fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
          };
#pragma unroll 5
          for (int i = 0; i < 20; ++i)
            tile[(2 * r8 + i / 10) * W3_PLANE + (i % 10) * W3_TP + 1 + lx] = pool(c_lo + 2 * r8 + i / 10, y0 - 1 + (i % 10), x0 + lx);
          if (tid < W3_CC * (W3_TY + 2)) {
            tile[hch * W3_PLANE + hrow * W3_TP] = pool(c_lo + hch, y0 - 1 + hrow, x0 - 1);
            tile[hch * W3_PLANE + hrow * W3_TP + W3_TX + 1] = pool(c_lo + hch, y0 - 1 + hrow, x0 + W3_TX);
          }
        }
        // ---- stage the tile's dz rows: thread = pixel (tid >> 5, tid & 31), pixels off the image contribute nothing
        {
          const int yy = y0 + (tid >> 5), xx = x0 + (tid & 31);
          const bool in = yy < Hr && xx < Wr;
          const unsigned po = in ? (unsigned)(yy * Wr + xx) * E : OOB;
          int rb0 = 0;
          for (int q = 0; q < a->nrs; ++q) {
            const int nq = a->rs[q].n;
            const csn_buf db = csn_make_buf_n(act_cast<AT>(a->rs[q].ptr) + (int64_t)b * a->rs[q].ctot * HW, (unsigned)(nq * HW) * E);
#pragma unroll 4
            for (int r = 0; r < nq; ++r) dzt[(rb0 + r) * W3_DZP + tid] = csn_bufacc<AT>::ld1(db, po, (unsigned)r * (unsigned)HW * E);
            rb0 += nq;
          }
          for (int r = nrows; r < rows16; ++r) dzt[r * W3_DZP + tid] = 0.f;
        }
        __syncthreads();
        // ---- contract over the wave's 64 pixels (tile rows 2 wave, 2 wave + 1), four at a time
#ifdef CSN_EMU_SEQ
        for (int sx = 0; sx < 16; ++sx)
          for (int t = 0; t < (two ? 2 : 1); ++t)
            for (int tp = 0; tp < 9; ++tp) {
              const int dy = tp / 3, dx = tp - 3 * dy;
              for (int i = 0; i < 4; ++i) {
                float v = acc[t][tp][i];
                for (int kk = 0; kk < 4; ++kk) {
                  const int qpx = 4 * sx + kk;
                  v = fmaf(dzt[(16 * t + kq * 4 + i) * W3_DZP + 64 * wave + qpx],
                           tile[pxi * W3_PLANE + (2 * wave + (qpx >> 5) + dy) * W3_TP + (qpx & 31) + dx], v);
                }
                acc[t][tp][i] = v;
              }
            }
#else
        {
          const float* ap = dzt + pxi * W3_DZP + 64 * wave + kq;
          const float* bp = tile + pxi * W3_PLANE + 2 * wave * W3_TP + kq;
#pragma unroll 2
          for (int sx = 0; sx < 16; ++sx) {
            const float a0 = ap[4 * sx];
            const float a1 = two ? ap[16 * W3_DZP + 4 * sx] : 0.f;
            const float* bs = bp + (sx >> 3) * W3_TP + ((4 * sx) & 31);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
              const float bv = bs[(tp / 3) * W3_TP + (tp % 3)];
              acc[0][tp] = csn_mfma_16x16x4(a0, bv, acc[0][tp]);
              if (two) acc[1][tp] = csn_mfma_16x16x4(a1, bv, acc[1][tp]);
            }
          }
        }
#endif
      }
      // ---- add the four waves' tiles (fixed order) and write this chunk's columns of the block's partial
      float* comb = dzt;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 4; ++i) comb[wave * 256 + (kq * 4 + i) * 16 + pxi] = acc[t][tp][i];
          __syncthreads();
          const int row = 16 * t + (tid >> 4), chl = tid & 15;
          if (chl < nc) {
            const float v = (comb[tid] + comb[256 + tid]) + (comb[512 + tid] + comb[768 + tid]);
            out[(int64_t)row * k16 + col0 + 9 * (c_lo + chl) + tp] = v;
          }
        }
      }
    }
    col0 += 9 * C;
  }
}

// one pass of plain 3x3 tap slices (dilation 1; own resolution or 2x2 max-pooled), at most 32 rows
bool csn_wgrad_c3_eligible(const WgArgs& a) {
  const char* env = std::getenv("CSN_WGRAD_TILED3");
  const bool off = env && env[0] == '0';
  if (off || a.ps.nsrc < 1 || a.rows16 > W3_MAX_ROWS) return false;
  for (int s = 0; s < a.ps.nsrc; ++s) {
    const int m = a.ps.src[s].mode;
    if (!(m == PW_TAPS || m == PW_POOL2_TAPS) || a.ps.src[s].dil != 1) return false;
  }
  return true;
}

int csn_wgrad_c3_blocks(const WgArgs& a) {
  const int tiles = ((a.Wr + W3_TX - 1) / W3_TX) * ((a.Hr + W3_TY - 1) / W3_TY) * a.B;
  return tiles < WG_MAX_BLOCKS ? tiles : WG_MAX_BLOCKS;
}

int csn_launch_wgrad_c3(const WgArgs& a, void* stream) {
  const size_t lds = (size_t)(W3_TILE + std::max(a.rows16 * W3_DZP, 1024)) * sizeof(float);
#ifndef CSN_CPU_EMU
  static CsnPerDeviceOnce attr_once;
  const int ast = attr_once.run([&]() {
    const void* fns[2] = {reinterpret_cast<const void*>(&goct_wgrad_c3_kernel<float>),
                          reinterpret_cast<const void*>(&goct_wgrad_c3_kernel<csn_bf16>)};
    for (const void* f : fns) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  });
  if (ast != 0) return ast;
#endif
  if (a.a16) CSN_LAUNCH((goct_wgrad_c3_kernel<csn_bf16>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a);
  else CSN_LAUNCH((goct_wgrad_c3_kernel<float>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
