// k_goct_c3.hip -- gOctConv 3x3 passes as an LDS-tiled implicit GEMM.
//
// Reference semantics (CSNet/model/csnet.py:664-726 with kernel_size 3, padding 1): for output branch j
//     y_j = conv3x3(x_j) [+ conv3x3(max_pool2(x_{j-1}))] [+ bilinear_up2(z)],   z = conv3x3(x_{j+1}) (own launch)
// followed by BN + PReLU (gOctaveCBR 778-792).  Also every 3x3 input-gradient launch of the backward pass (transposed,
// tap-flipped weight blocks) and the train-mode forward (RAW).
//
// A block owns an 8 x 32 output tile and walks the input channels 16 at a time: the (8 + 2) x (32 + 2) input tile of the
// chunk goes to LDS (2x2 max-pool on the way in for the high -> low slice; zero padding = buffer loads whose offset is
// out of range), so a value is fetched from global memory 1.33 times instead of 9.  v_mfma_f32_16x16x4_f32 contracts
// straight from the tile: A = W[row][k], B = x[k][pixel] with the k index TAP-MAJOR inside a chunk (k = 16 tap + channel):
//   * the four k rows of an MFMA step are four CHANNELS of one tap, i.e. four planes of the tile at the same (dy, dx): with a
//     plane pitch of 368 floats (= 16 mod 32) the two k rows a 32-lane half reads sit on disjoint banks;
//   * every address of the unrolled 9 taps x 4 channel groups is lane base + compile-time constant: the ds_read offsets are
//     immediates, no index arithmetic in the loop (round 1 spent 10 VALU instructions per MFMA on k -> (channel, dy, dx)
//     divisions and on a per-element staging loop; SQ_INSTS_VALU 1.1e8 for 1.1e7 MFMAs per forward).
// The weight image of the pass is laid out to match by csn_prep_kernel (CSN_PREP_C3T: [row][chunk][tap][16], zero padded).
// Staging is software-pipelined: the global loads of chunk i + 1 (22 per thread: a thread owns two channels x ten tile
// rows of one column, plus one halo pair) are in flight while chunk i is contracted and land in LDS after the barrier.
// The optional low -> high term bilinear_up2(z) is added in the epilogue, where a lane owns a pixel (round 1 pushed it
// through the matrix pipe with an identity weight block).
#include "pw_gather.h"

#define C3_TX 32
#define C3_TY 8
#define C3_CC 16                        // channels per staged chunk
#define C3_TP 36                        // tile row pitch (34 columns used)
#define C3_PLANE 368                    // floats per channel: 10 rows x 36 + 8, == 16 (mod 32)
#define C3_TILE (C3_CC * C3_PLANE)      // 5888 floats = 23 KB
#define C3_KC (9 * C3_CC)               // weight columns per chunk
#define C3_WP (C3_KC + 2)               // pitch of a staged weight chunk: 146 = 2 (mod 4), conflict-free A reads
#define C3_WCH (32 * C3_WP)             // 32 rows x 144 columns = 18.3 KB

struct C3Chunk {   // chunk id -> slice, first channel
  int s, c_lo;
};

// acc += W[rows][chunk columns] x tile, NG groups of four channels.  wt = &W[lane & 15][lane >> 4] of the chunk's first
// column (pitch wp), tb = &tile[lane >> 4][2 wave][lane & 15].
template <bool TWO>
__device__ __forceinline__ void c3_contract(csn_f4 (&acc)[2][4], const float* wt, int wp, const float* tb, int ng) {
#ifndef CSN_EMU_SEQ
  for (int g = 0; g < ng; ++g) {
    const float* wg = wt + 4 * g;
    const float* tg = tb + 4 * g * C3_PLANE;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - 3 * dy;
      const float a0 = wg[16 * t];
      const float a1 = TWO ? wg[16 * wp + 16 * t] : 0.f;
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const float bv = tg[((sg >> 1) + dy) * C3_TP + 16 * (sg & 1) + dx];
        acc[0][sg] = csn_mfma_16x16x4(a0, bv, acc[0][sg]);
        if (TWO) acc[1][sg] = csn_mfma_16x16x4(a1, bv, acc[1][sg]);
      }
    }
  }
#endif
}

#ifdef CSN_EMU_SEQ
// D[row][px] += sum_k W[row][k] x[k][px] in the kernel's k order; this lane holds rows (lane >> 4) * 4 + i of pixel lane & 15
static inline void c3_contract_emu(csn_f4 (&acc)[2][4], const float* w0, int wp, const float* tile, int wave, int lane,
                                   int ng, bool two) {
  const int kq = lane >> 4, pxi = lane & 15;
  for (int g = 0; g < ng; ++g)
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - 3 * dy;
      for (int tt = 0; tt < (two ? 2 : 1); ++tt)
        for (int sg = 0; sg < 4; ++sg)
          for (int i = 0; i < 4; ++i) {
            const int row = 16 * tt + kq * 4 + i;
            float a_ = acc[tt][sg][i];
            for (int u = 0; u < 4; ++u)
              a_ = fmaf(w0[row * wp + 16 * t + 4 * g + u],
                        tile[(4 * g + u) * C3_PLANE + (2 * wave + (sg >> 1) + dy) * C3_TP + 16 * (sg & 1) + pxi + dx], a_);
            acc[tt][sg][i] = a_;
          }
    }
}
#endif

// WCH = false: the pass's whole weight image is staged once per block (small images);
// WCH = true: only the current chunk's columns live in LDS (large images: 42 KB per block, 3 blocks per CU).
template <bool RAW, bool WCH, typename AT>
__global__ __launch_bounds__(CSN_BLOCK, 3) void goct_c3_kernel(PwArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  constexpr unsigned E = (unsigned)sizeof(AT);   // bytes per activation element
  const CSN_CONST_AS PwArgs* a = CSN_KERNARG(PwArgs, a_byval);
  PwPassP ps = &a->pass[0];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w3s = a->w3_stride;
  if (!WCH) {
    const float2* __restrict__ src = reinterpret_cast<const float2*>(a->wimg3);
    float2* dst = reinterpret_cast<float2*>(lds);
    const int n2 = a->w3_floats >> 1;
    for (int i = tid; i < n2; i += CSN_BLOCK) dst[i] = src[i];
  }
  float* wch = lds;                            // WCH: [32][C3_WP] weight columns of the current chunk
  float* tile = lds + (WCH ? C3_WCH : a->w3_floats);   // [C3_CC][C3_PLANE]; later the waves' epilogue scratch
  const int wp = WCH ? C3_WP : w3s;
  const int Hr = a->H0, Wr = a->W0;
  const int tiles_x = (Wr + C3_TX - 1) / C3_TX, tiles_y = (Hr + C3_TY - 1) / C3_TY;
  const int tiles_xy = tiles_x * tiles_y;
  const int ntiles = tiles_xy * a->B;
  const int nrows = ps->nrows;
  // tap slices first (TAPS / POOL2_TAPS), an optional bilinear z slice last (added in the epilogue)
  int ntap = 0;
  for (int s = 0; s < ps->nsrc; ++s)
    if (ps->src[s].mode == PW_TAPS || ps->src[s].mode == PW_POOL2_TAPS) ntap = s + 1;
  const bool has_z = ps->nsrc > ntap;
  // chunks of 16 channels, slice after slice (scalars, not an array: a dynamically indexed local array lives in scratch)
  const int nch0 = ntap > 0 ? (ps->src[0].C + C3_CC - 1) / C3_CC : 0;
  const int nch1 = ntap > 1 ? (ps->src[1].C + C3_CC - 1) / C3_CC : 0;
  const int nch2 = ntap > 2 ? (ps->src[2].C + C3_CC - 1) / C3_CC : 0;
  const int nchunk = nch0 + nch1 + nch2;
  auto chunk_of = [&](int id) {
    C3Chunk c;
    if (id < nch0) { c.s = 0; c.c_lo = id * C3_CC; }
    else if (id < nch0 + nch1) { c.s = 1; c.c_lo = (id - nch0) * C3_CC; }
    else { c.s = 2; c.c_lo = (id - nch0 - nch1) * C3_CC; }
    return c;
  };
  // XCD-aware tile order (see goct_pw_kernel)
  const int nslot = gridDim.x >> 3;
  const int chunkx = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min((xcd + 1) * chunkx, ntiles);
  const int kq = lane >> 4, pxi = lane & 15;
  // staging roles: thread (r8, lx) owns column lx of channels 2 r8, 2 r8 + 1 (ten tile rows each); threads 0..159 also own
  // the two halo columns of tile row `hrow` of channel `hch`
  const int r8 = tid >> 5, lx = tid & 31;
  const int hch = tid / (C3_TY + 2), hrow = tid - hch * (C3_TY + 2);
  const unsigned OOB = 0x80000000u;
  for (int tl = xcd * chunkx + (blockIdx.x >> 3); tl < tend; tl += nslot) {
    const int b = tl / tiles_xy;
    const int txy = tl - b * tiles_xy;
    const int y0 = (txy / tiles_x) * C3_TY, x0 = (txy % tiles_x) * C3_TX;
    // lane-as-pixel coordinates (epilogue): pixel p = lane of the wave's 2 x 32 strip
    const int py_ = y0 + 2 * wave + (lane >> 5), px_ = x0 + (lane & 31);
    const bool valid = py_ < Hr && px_ < Wr;
    const int gy = min(py_, Hr - 1), gx = min(px_, Wr - 1);
    const unsigned ovoff = (unsigned)(gy * Wr + gx) * E;
    for (int row0 = 0; row0 < nrows; row0 += 32) {
      const bool two = nrows - row0 > 16;
      csn_f4 acc[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][s][i] = 0.f;
      float pf[22];          // prefetched tile elements of the next chunk (plain slices)
      bool pf_valid = false;
      // issue the loads of a plain (not pooled) chunk: out-of-image / past-the-slice elements get an out-of-range offset -> 0
      auto prefetch = [&](const C3Chunk& c) {
        const int C = ps->src[c.s].C;
        const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[c.s].ptr) + (int64_t)b * ps->src[c.s].Ctot * (Hr * Wr),
                                          (unsigned)(ps->src[c.s].Ctot * Hr * Wr) * E);
        const unsigned HW4 = (unsigned)(Hr * Wr) * E;
        const int xx = x0 + lx;
        const unsigned colo = xx < Wr ? (unsigned)xx * E : OOB;
#pragma unroll
        for (int i = 0; i < 20; ++i) {
          const int ch = c.c_lo + 2 * r8 + i / 10, yy = y0 - 1 + (i % 10);
          const bool ok = ch < C && yy >= 0 && yy < Hr;
          pf[i] = csn_bufacc<AT>::ld1(rb, ok ? (unsigned)ch * HW4 + (unsigned)(yy * Wr) * E + colo : OOB, 0u);
        }
        {
          const int ch = c.c_lo + hch, yy = y0 - 1 + hrow;
          const bool ok = tid < C3_CC * (C3_TY + 2) && ch < C && yy >= 0 && yy < Hr;
          const unsigned ro = (unsigned)ch * HW4 + (unsigned)(yy * Wr) * E;
          pf[20] = csn_bufacc<AT>::ld1(rb, ok && x0 > 0 ? ro + (unsigned)(x0 - 1) * E : OOB, 0u);
          pf[21] = csn_bufacc<AT>::ld1(rb, ok && x0 + C3_TX < Wr ? ro + (unsigned)(x0 + C3_TX) * E : OOB, 0u);
        }
      };
      auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 20; ++i) tile[(2 * r8 + i / 10) * C3_PLANE + (i % 10) * C3_TP + 1 + lx] = pf[i];
        if (tid < C3_CC * (C3_TY + 2)) {
          tile[hch * C3_PLANE + hrow * C3_TP] = pf[20];
          tile[hch * C3_PLANE + hrow * C3_TP + C3_TX + 1] = pf[21];
        }
      };
      // pooled chunk: 2x2 max of the source at twice the resolution, staged synchronously
      auto stage_pooled = [&](const C3Chunk& c) {
        const int C = ps->src[c.s].C;
        const int Hs = 2 * Hr, Ws = 2 * Wr;
        const csn_buf rb = csn_make_buf_n(act_cast<AT>(ps->src[c.s].ptr) + (int64_t)b * ps->src[c.s].Ctot * (Hs * Ws),
                                          (unsigned)(ps->src[c.s].Ctot * Hs * Ws) * E);
        const unsigned HW4 = (unsigned)(Hs * Ws) * E, P4 = (unsigned)Ws * E;
        auto pooled = [&](int ch, int yy, int xx) {
          const bool ok = ch < C && yy >= 0 && yy < Hr && xx >= 0 && xx < Wr;
          const unsigned o = ok ? (unsigned)ch * HW4 + (unsigned)(2 * yy) * P4 + (unsigned)xx * (2u * E) : OOB;
          const float2 t0 = csn_bufacc<AT>::ld2(rb, o, 0u), t1 = csn_bufacc<AT>::ld2(rb, ok ? o + P4 : OOB, 0u);
          return fmaxf(fmaxf(t0.x, t0.y), fmaxf(t1.x, t1.y));
        };
#pragma unroll 5
        for (int i = 0; i < 20; ++i)
          tile[(2 * r8 + i / 10) * C3_PLANE + (i % 10) * C3_TP + 1 + lx] = pooled(c.c_lo + 2 * r8 + i / 10, y0 - 1 + (i % 10), x0 + lx);
        if (tid < C3_CC * (C3_TY + 2)) {
          tile[hch * C3_PLANE + hrow * C3_TP] = pooled(c.c_lo + hch, y0 - 1 + hrow, x0 - 1);
          tile[hch * C3_PLANE + hrow * C3_TP + C3_TX + 1] = pooled(c.c_lo + hch, y0 - 1 + hrow, x0 + C3_TX);
        }
      };
      {
        const C3Chunk c0 = chunk_of(0);
        if (ps->src[c0.s].mode == PW_TAPS) { prefetch(c0); pf_valid = true; }
      }
      for (int ci = 0; ci < nchunk; ++ci) {
        const C3Chunk c = chunk_of(ci);
        const int nc = min(C3_CC, ps->src[c.s].C - c.c_lo);
        __syncthreads();   // the previous chunk (tile, weight chunk) / the previous sweep's epilogue scratch is consumed
        if (pf_valid) commit();
        else stage_pooled(c);
        if (WCH) {         // weight columns of this chunk, rows row0 .. row0 + 31: thread (r = tid >> 3) x 9 float2
          if (two || wave < 2) {
            const float* __restrict__ wg = a->wimg3 + (int64_t)(row0 + (tid >> 3)) * w3s + ci * C3_KC;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
              const int k2 = (tid & 7) + 8 * j;
              *reinterpret_cast<float2*>(wch + (tid >> 3) * C3_WP + 2 * k2) = *reinterpret_cast<const float2*>(wg + 2 * k2);
            }
          }
        }
        __syncthreads();
        pf_valid = false;
        if (ci + 1 < nchunk) {   // next chunk's loads fly while this one is contracted
          const C3Chunk cn = chunk_of(ci + 1);
          if (ps->src[cn.s].mode == PW_TAPS) { prefetch(cn); pf_valid = true; }
        }
        const int ng = (nc + 3) >> 2;
        const float* w0 = WCH ? wch : lds + (int64_t)row0 * w3s + ci * C3_KC;
#ifdef CSN_EMU_SEQ
        c3_contract_emu(acc, w0, wp, tile, wave, lane, ng, two);
#else
        const float* wt = w0 + pxi * wp + kq;
        const float* tb = tile + kq * C3_PLANE + 2 * wave * C3_TP + pxi;
        if (two) c3_contract<true>(acc, wt, wp, tb, ng);
        else c3_contract<false>(acc, wt, wp, tb, ng);
#endif
      }
      __syncthreads();   // tile consumed: the waves' private epilogue scratch lives in the same LDS from here on
      float* xb = tile + wave * (PW_KC * PW_XP);
      // ---- low -> high term: this lane's bilinear taps of z (x2, align_corners=False), added per output row below
      csn_buf zb = csn_make_buf(ps->out);
      unsigned zo00 = 0, zo01 = 0, zo10 = 0, zo11 = 0, zcs4 = 0;
      float zw00 = 0.f, zw01 = 0.f, zw10 = 0.f, zw11 = 0.f;
      if (has_z) {
        const int Hs = Hr >> 1, Ws = Wr >> 1;
        int yy0, yy1, xx0, xx1;
        float ly, lxw;
        csn_bilin(gy, 0.5f, Hs, yy0, yy1, ly);
        csn_bilin(gx, 0.5f, Ws, xx0, xx1, lxw);
        zcs4 = (unsigned)(Hs * Ws) * E;
        zb = csn_make_buf(act_cast<AT>(ps->src[ntap].ptr) + (int64_t)b * ps->src[ntap].Ctot * (Hs * Ws));
        zo00 = (unsigned)(yy0 * Ws + xx0) * E; zo01 = (unsigned)(yy0 * Ws + xx1) * E;
        zo10 = (unsigned)(yy1 * Ws + xx0) * E; zo11 = (unsigned)(yy1 * Ws + xx1) * E;
        zw00 = (1.f - ly) * (1.f - lxw); zw01 = (1.f - ly) * lxw; zw10 = ly * (1.f - lxw); zw11 = ly * lxw;
      }
      // ---- epilogue: transpose through the wave's scratch, (+ z), folded BN + PReLU, 128-byte row segments
      const unsigned cs4 = (unsigned)(Hr * Wr) * E;
      const csn_buf ob = csn_make_buf(act_cast<AT>(ps->out) + (int64_t)b * ps->out_ctot * (Hr * Wr));
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
          float v = xb[rr * PW_EP + lane];
          if (has_z) {
            const unsigned so = (unsigned)(a->z_c0 + rbase + rr) * zcs4;
            v += zw00 * csn_bufacc<AT>::ld1(zb, zo00, so) + zw01 * csn_bufacc<AT>::ld1(zb, zo01, so) +
                 zw10 * csn_bufacc<AT>::ld1(zb, zo10, so) + zw11 * csn_bufacc<AT>::ld1(zb, zo11, so);
          }
          const float val = RAW ? v : csn_epi(v, scale[rr], shift[rr], alpha[rr]);
          if (valid) csn_bufacc<AT>::st1(ob, ovoff, (unsigned)(rbase + rr) * cs4, val);
        }
      }
    }
  }
}

// true when the launch is one pass made of 3x3 tap slices (dilation 1) plus at most one trailing bilinear slice, and the
// planner laid out the tap-major weight image for it
bool csn_c3_eligible(const PwArgs& a) {
  if (a.npass != 1 || a.pass[0].red_w || a.pass[0].r != 0 || !a.wimg3) return false;   // the kernel walks the launch resolution
  const PwPass& ps = a.pass[0];
  int ntap = 0;
  bool tail = false;
  for (int s = 0; s < ps.nsrc; ++s) {
    const int m = ps.src[s].mode;
    if (m == PW_TAPS || m == PW_POOL2_TAPS) {
      if (tail || ps.src[s].dil != 1) return false;
      ++ntap;
    } else {
      if (m != PW_UP2 || tail) return false;
      tail = true;
    }
  }
  return ntap > 0;
}

int csn_launch_c3(const PwArgs& a, int raw, void* stream) {
  const int tiles = ((a.W0 + C3_TX - 1) / C3_TX) * ((a.H0 + C3_TY - 1) / C3_TY) * a.B;
  const int nblk = tiles < PW_MAX_GRID ? tiles : PW_MAX_GRID;
  const dim3 grid((nblk + 7) & ~7);
  // small weight images are staged whole, large ones chunk by chunk
  const bool wch = (size_t)a.w3_floats * sizeof(float) > 40 * 1024;
  const size_t lds = (size_t)((wch ? C3_WCH : a.w3_floats) + C3_TILE) * sizeof(float);
#ifndef CSN_CPU_EMU
  static CsnPerDeviceOnce attr_once;
  const int ast = attr_once.run([&]() {
    const void* fns[6] = {reinterpret_cast<const void*>(&goct_c3_kernel<false, false, float>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<false, true, float>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<true, false, float>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<true, true, float>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<true, false, csn_bf16>),
                          reinterpret_cast<const void*>(&goct_c3_kernel<true, true, csn_bf16>)};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  });
  if (ast != 0) return ast;
#endif
  if (a.a16) {   // bf16 activations: train mode only (raw sums; BN runs as its own passes)
    if (!raw) return -1;
    if (wch) CSN_LAUNCH((goct_c3_kernel<true, true, csn_bf16>), grid, dim3(CSN_BLOCK), lds, stream, a);
    else CSN_LAUNCH((goct_c3_kernel<true, false, csn_bf16>), grid, dim3(CSN_BLOCK), lds, stream, a);
    return (int)hipGetLastError();
  }
  if (raw && wch) CSN_LAUNCH((goct_c3_kernel<true, true, float>), grid, dim3(CSN_BLOCK), lds, stream, a);
  else if (raw) CSN_LAUNCH((goct_c3_kernel<true, false, float>), grid, dim3(CSN_BLOCK), lds, stream, a);
  else if (wch) CSN_LAUNCH((goct_c3_kernel<false, true, float>), grid, dim3(CSN_BLOCK), lds, stream, a);
  else CSN_LAUNCH((goct_c3_kernel<false, false, float>), grid, dim3(CSN_BLOCK), lds, stream, a);
  return (int)hipGetLastError();
}
