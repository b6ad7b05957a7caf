// k_wgrad_bf.hip -- weight gradient of the 1x1 gOctConv passes of the bfloat16 train step on the bf16 matrix cores.
//
//     dW[row][k] = sum_{n, p} dz[row][p] * g[k][p]         (autograd of gOctaveConv.forward, csnet.py:664-726, k = 1;
//                                                           g = the unit's own-resolution input or its 2x2 max-pool)
// In the bf16 storage mode (CSN_OPT_TRAIN_BF16) dz, x and the adjoint-upsampled dz are bfloat16 tensors in the planar layout
// [B][C][H*W]: the reduction index -- the pixel -- is the contiguous one for BOTH operands, so eight consecutive pixels of one
// plane (one 128-bit load) ARE an A / B operand of v_mfma_f32_32x32x16_bf16.  bf16 x bf16 products are exact in fp32 and the
// accumulation is fp32: no precision is given up against the fp32-MFMA kernel (k_wgrad.hip), which converted the same
// bfloat16 values to float, staged them in LDS panels and spent a 32-cycle matrix instruction on 16 x 16 x 4 products.
//
// Mapping.  The instruction computes D[i][j] = sum_{k < 16} A[i][k] B[k][j]; lane l holds A[i = l & 31][8 (l >> 5) .. + 7] and
// B[8 (l >> 5) .. + 7][j = l & 31].  The k slot of (lane half, element) is the same for A and B, so ANY assignment of pixels to
// slots is a valid reduction as long as both operands use it.  With S in {1, 2, 4} "sub-blocks" the 32 A rows are (plane r = i / S,
// sub-block s = i % S) and lane l loads the 8 pixels
//     p0 + (l >> 5) * 8 S + s * 8 .. + 7            of plane r,
// likewise for B with channel c = j / S: one load instruction covers 32 / S planes x 16 S pixels, and per plane the lanes of a
// sub-block group read 16 S contiguous bytes per half (S = 4: a whole 128-byte line per plane -- what the memory system wants on
// the large maps; S = 1: 32 planes x 32 bytes for the wide units of the small maps).  D[(r, s)][(c, s')] is the wanted partial
// sum for s == s' (the other entries pair different pixels and are dropped): the matrix pipe runs at 1 / S efficiency, which is
// irrelevant -- the launch is bound by the HBM stream (18 + 13 planes of 12.8 M pixels at stage 1: 0.8 GB per pass).
// Rows / channels past the pass are clamped duplicates (every D entry is an independent dot product: they only fill entries
// nobody reads).  The high -> low passes (g = 2x2 max-pool of the finer input, csnet.py:708-714) form the pooled operand in
// registers from four 128-bit loads (the max of bfloat16 values is a bfloat16: exact).
//
// A wave owns items = runs of L consecutive pixel sets of one image, loads of set s + 1 in flight while set s is contracted, and
// keeps its NTR x NTC accumulator tiles over ALL its items; at the end the four waves' tiles go through LDS, the S diagonal
// blocks are summed and the block writes its partial dW in the layout wgrad_reduce_kernel (k_wgrad.hip) reduces in fixed order.
#include "csn_kernels.h"

struct WgBfSrc {
  const void* ptr;      // first plane of the slice inside [B][ctot][planes' H*W]
  int32_t ctot, n;      // planes per image of the tensor, planes of the slice
};
struct WgBfArgs {
  WgBfSrc rs[3];        // dz rows, source after source
  WgBfSrc cs[3];        // gathered channels, source after source (own resolution, or twice the resolution when pool)
  int32_t nrs, ncs;
  int32_t R, K;         // rows / gathered channels of the pass
  int32_t HW, W;        // pixels per plane and row width at the pass resolution
  int32_t B;
  int32_t slog;         // log2 S
  int32_t L;            // pixel sets (16 S pixels) per item
  int32_t runs;         // items per image
  int32_t nitems;       // B * runs
  int32_t nblk;
  int32_t rows16, k16;  // layout of a block's partial: [rows16][k16]
  float* partial;
};

typedef csn_f16v wgb_f16;

typedef const CSN_CONST_AS WgBfArgs* WgBfArgsP;

// plane `p` of a source list: base pointer of image 0's plane and the image stride in elements
__device__ __forceinline__ void wgbf_plane(const CSN_CONST_AS WgBfSrc* src, int nsrc, int p, int64_t hw, const char*& base,
                                           unsigned& istride) {
  int q = 0;
  while (q + 1 < nsrc && p >= src[q].n) { p -= src[q].n; ++q; }
  if (p >= src[q].n) p = src[q].n - 1;   // past the pass: a clamped duplicate
  base = reinterpret_cast<const char*>(src[q].ptr) + (int64_t)p * hw * 2;
  istride = (unsigned)((int64_t)src[q].ctot * hw);
}

#ifdef CSN_EMU_SEQ
// Functional stand-in (the lane maps themselves run under `make LANES=1`, csn_device.h): the same items per wave, the same partial layout.
template <int NTR, int NTC, bool POOL>
__global__ void wgrad_bf16_kernel(WgBfArgs a_byval) {
  const WgBfArgs* a = &a_byval;
  if (threadIdx.x != 0) return;
  const int S = 1 << a->slog, PXS = 16 * S, HW = a->HW, W = a->W, R = a->R, K = a->K;
  std::vector<float> acc((size_t)R * K, 0.f);
  std::vector<float> dz(R), g(K);
  for (int wave = 0; wave < 4; ++wave)
    for (int it = blockIdx.x * 4 + wave; it < a->nitems; it += a->nblk * 4) {
      const int b = it / a->runs, run = it - b * a->runs;
      const int q0 = run * a->L * PXS, q1 = min(HW, q0 + a->L * PXS);
      for (int p = q0; p < q1; ++p) {
        for (int r = 0; r < R; ++r) {
          const char* base; unsigned is;
          wgbf_plane(a->rs, a->nrs, r, HW, base, is);
          dz[r] = csn_bf2f(reinterpret_cast<const unsigned short*>(base)[(int64_t)b * is + p]);
        }
        for (int k = 0; k < K; ++k) {
          const char* base; unsigned is;
          wgbf_plane(a->cs, a->ncs, k, POOL ? 4 * (int64_t)HW : HW, base, is);
          const unsigned short* pl = reinterpret_cast<const unsigned short*>(base) + (int64_t)b * is;
          if (POOL) {
            const int y = p / W, x = p - y * W;
            const unsigned short* q = pl + (int64_t)(2 * y) * (2 * W) + 2 * x;
            g[k] = fmaxf(fmaxf(csn_bf2f(q[0]), csn_bf2f(q[1])), fmaxf(csn_bf2f(q[2 * W]), csn_bf2f(q[2 * W + 1])));
          } else {
            g[k] = csn_bf2f(pl[p]);
          }
        }
        for (int r = 0; r < R; ++r)
          for (int k = 0; k < K; ++k) acc[(size_t)r * K + k] = fmaf(dz[r], g[k], acc[(size_t)r * K + k]);
      }
    }
  float* out = a->partial + (int64_t)blockIdx.x * a->rows16 * a->k16;
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < K; ++k) out[(int64_t)r * a->k16 + k] = acc[(size_t)r * K + k];
}
#else
// (a pointer rebuilt from integers has no address space: say "global", or the loads are flat_load and count on lgkmcnt too)
#ifdef CSN_CPU_EMU
__device__ __forceinline__ csn_u4 wgbf_ld(const char* p) { return *reinterpret_cast<const csn_u4*>(p); }
#else
typedef const __attribute__((address_space(1))) csn_u4* wgbf_gp;
__device__ __forceinline__ csn_u4 wgbf_ld(const char* p) { return *(wgbf_gp)(unsigned long long)p; }
#endif

// 2x2 max-pool of two rows of 16 bfloat16 values -> 8 bfloat16 values (dword d of a row = the horizontal pair of output d)
__device__ __forceinline__ unsigned wgbf_pool_pair(unsigned u0a, unsigned u1a, unsigned u0b, unsigned u1b) {
  const float a = fmaxf(fmaxf(csn_bits_f(u0a << 16), csn_bits_f(u0a & 0xffff0000u)),
                        fmaxf(csn_bits_f(u1a << 16), csn_bits_f(u1a & 0xffff0000u)));
  const float b = fmaxf(fmaxf(csn_bits_f(u0b << 16), csn_bits_f(u0b & 0xffff0000u)),
                        fmaxf(csn_bits_f(u1b << 16), csn_bits_f(u1b & 0xffff0000u)));
  return (csn_f_bits(a) >> 16) | (csn_f_bits(b) & 0xffff0000u);
}
__device__ __forceinline__ csn_u4 wgbf_pool(const csn_u4 (&r)[4]) {   // r[0], r[1]: row 2y (16 px); r[2], r[3]: row 2y + 1
  csn_u4 o;
  o.x = wgbf_pool_pair(r[0].x, r[2].x, r[0].y, r[2].y);
  o.y = wgbf_pool_pair(r[0].z, r[2].z, r[0].w, r[2].w);
  o.z = wgbf_pool_pair(r[1].x, r[3].x, r[1].y, r[3].y);
  o.w = wgbf_pool_pair(r[1].z, r[3].z, r[1].w, r[3].w);
  return o;
}

template <int NTR, int NTC>
__device__ __forceinline__ void wgbf_mma(const csn_u4 (&A)[NTR], const csn_u4 (&Bv)[NTC], wgb_f16 (&acc)[NTR][NTC]) {
#pragma unroll
  for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
    for (int tc = 0; tc < NTC; ++tc)
      acc[tr][tc] = csn_mfma_32x32x16_bf16(A[tr], Bv[tc], acc[tr][tc]);
}

template <int NTR, int NTC, bool POOL>
__global__ __launch_bounds__(CSN_BLOCK, 2) void wgrad_bf16_kernel(WgBfArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  WgBfArgsP a = CSN_KERNARG(WgBfArgs, a_byval);
  constexpr int NT = NTR + NTC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = csn_readfirstlane(tid >> 6);
  const int sl = a->slog, S = 1 << sl, PXS = 16 << sl, TP = 32 >> sl;
  const int HW = a->HW, W = a->W;
  // ---- per-(tile, lane) plane table: base pointer of image 0 and image stride
  csn_u2* tabp = reinterpret_cast<csn_u2*>(lds);                 // [NT][64]
  unsigned* tabs = reinterpret_cast<unsigned*>(lds) + NT * 128;  // [NT][64]
  for (int e = tid; e < NT * 64; e += CSN_BLOCK) {
    const int t = e >> 6, l = e & 63;
    const int pl = (l & 31) >> sl;
    const char* base;
    unsigned is;
    if (t < NTR) wgbf_plane(a->rs, a->nrs, min(t * TP + pl, a->R - 1), HW, base, is);
    else wgbf_plane(a->cs, a->ncs, min((t - NTR) * TP + pl, a->K - 1), POOL ? 4 * (int64_t)HW : (int64_t)HW, base, is);
    const unsigned long long bv = (unsigned long long)base;
    csn_u2 v;
    v.x = (unsigned)bv; v.y = (unsigned)(bv >> 32);
    tabp[e] = v;
    tabs[e] = is;
  }
  __syncthreads();
  const int lpix = ((lane >> 5) << (3 + sl)) + ((lane & (S - 1)) << 3);   // this lane's first pixel inside a set
  const float rcpW = 1.0f / (float)W;
  wgb_f16 acc[NTR][NTC];
#pragma unroll
  for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
    for (int tc = 0; tc < NTC; ++tc)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tr][tc][i] = 0.f;

  for (int it = blockIdx.x * 4 + wave; it < a->nitems; it += a->nblk * 4) {
    const int b = it / a->runs, run = it - b * a->runs;
    const int q0 = run * a->L * PXS;
    const int nset = min(a->L, (HW - q0) / PXS);
    const char* pr[NTR];
    const char* pc[NTC];
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      const csn_u2 bp = tabp[t * 64 + lane];
      const unsigned long long base = ((unsigned long long)bp.y << 32) | bp.x;
      pr[t] = reinterpret_cast<const char*>(base + 2ull * ((unsigned long long)b * tabs[t * 64 + lane] + (unsigned)(q0 + lpix)));
    }
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      const csn_u2 bp = tabp[(NTR + t) * 64 + lane];
      const unsigned long long base = ((unsigned long long)bp.y << 32) | bp.x;
      pc[t] = reinterpret_cast<const char*>(base + 2ull * ((unsigned long long)b * tabs[(NTR + t) * 64 + lane] + (POOL ? 0u : (unsigned)(q0 + lpix))));
    }
    const int step = PXS * 2;   // bytes per set in an own-resolution plane
    // U sets per trip: ALL their loads are issued before the first contraction (a hand-written prefetch across trips is undone by
    // the compiler: a phi of loads becomes a load of the phi'd pointer); the tail sets of a run are fetched again (clamped)
    // and contracted with a zeroed A operand
    if constexpr (!POOL) {
      constexpr int U = (NTR * NTC * 16 + NT * 16 <= 176) ? 4 : 2;
      for (int s = 0; s < nset; s += U) {
        csn_u4 A[U][NTR], Bv[U][NTC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int o = min(s + u, nset - 1) * step;
#pragma unroll
          for (int t = 0; t < NTR; ++t) A[u][t] = wgbf_ld(pr[t] + o);
#pragma unroll
          for (int t = 0; t < NTC; ++t) Bv[u][t] = wgbf_ld(pc[t] + o);
        }
        CSN_SCHED_FENCE();   // every load of the trip is in flight before the first contraction
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (u > 0) {
            const unsigned m = (s + u < nset) ? 0xffffffffu : 0u;
#pragma unroll
            for (int t = 0; t < NTR; ++t) { A[u][t].x &= m; A[u][t].y &= m; A[u][t].z &= m; A[u][t].w &= m; }
          }
          wgbf_mma<NTR, NTC>(A[u], Bv[u], acc);
        }
      }
    } else {
      // pooled channels: the lane's 8 low pixels (one row: W % 8 == 0) <- 16 px of rows 2y, 2y + 1 of the finer plane
      constexpr int U = NTC <= 2 ? 2 : 1;
      const unsigned rowb = (unsigned)(2 * W) * 2u;   // bytes per row of the finer plane
      auto hoff = [&](int s) {                        // byte offset of (2y, 2x) for this lane's first pixel of set s
        const int p = q0 + s * PXS + lpix;
        int y = (int)((float)p * rcpW);
        y -= (y * W > p) ? 1 : 0;
        y += ((y + 1) * W <= p) ? 1 : 0;
        const int x = p - y * W;
        return (unsigned)(2 * y) * rowb + (unsigned)(4 * x);
      };
      for (int s = 0; s < nset; s += U) {
        csn_u4 A[U][NTR], raw[U][NTC][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int sc = min(s + u, nset - 1);
          const unsigned o = hoff(sc);
#pragma unroll
          for (int t = 0; t < NTR; ++t) A[u][t] = wgbf_ld(pr[t] + sc * step);
#pragma unroll
          for (int t = 0; t < NTC; ++t) {
            raw[u][t][0] = wgbf_ld(pc[t] + o); raw[u][t][1] = wgbf_ld(pc[t] + o + 16);
            raw[u][t][2] = wgbf_ld(pc[t] + o + rowb); raw[u][t][3] = wgbf_ld(pc[t] + o + rowb + 16);
          }
        }
        CSN_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < U; ++u) {
          csn_u4 Bc[NTC];
#pragma unroll
          for (int t = 0; t < NTC; ++t) Bc[t] = wgbf_pool(raw[u][t]);
          if (u > 0) {
            const unsigned m = (s + u < nset) ? 0xffffffffu : 0u;
#pragma unroll
            for (int t = 0; t < NTR; ++t) { A[u][t].x &= m; A[u][t].y &= m; A[u][t].z &= m; A[u][t].w &= m; }
          }
          wgbf_mma<NTR, NTC>(A[u], Bc, acc);
        }
      }
    }
  }

  // ---- the four waves' tiles through LDS: sum the S diagonal blocks, write the block's partial
  float* out = a->partial + (int64_t)blockIdx.x * a->rows16 * a->k16;
  const int j = lane & 31, ih = (lane >> 5) * 4;
  const int rows16 = a->rows16, k16 = a->k16;
#pragma unroll
  for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
    for (int tc = 0; tc < NTC; ++tc) {
      __syncthreads();   // (first round: the plane table is no longer read)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) lds[wave * 1024 + ((reg & 3) + 8 * (reg >> 2) + ih) * 32 + j] = acc[tr][tc][reg];
      __syncthreads();
      for (int o = tid; o < TP * TP; o += CSN_BLOCK) {
        const int r = o / TP, c = o - r * TP;
        float v = 0.f;
        for (int w = 0; w < 4; ++w)
          for (int s = 0; s < S; ++s) v += lds[w * 1024 + (r * S + s) * 32 + c * S + s];
        const int row = tr * TP + r, col = tc * TP + c;
        if (row < rows16 && col < k16) out[(int64_t)row * k16 + col] = v;
      }
    }
}
#endif

// ================================================================================================ 3x3 tap passes
// dW[row][9 ch + t] = sum_{n, p} dz[row][p] * x[ch][p + off(t)],  t = 3 (dy + 1) + (dx + 1)        (csnet.py:664-726 with k = 3;
// x = the pass-resolution input: the avg-pooled copy of a stride-2 unit, or the 2x2 max-pooled copy c3q_kernel reads)
// on v_mfma_f32_16x16x32_bf16 with the same "a load is an operand" mapping: lane l holds A[i = l & 15][8 (l >> 4) .. + 7], four k
// groups of 8 pixels; with S sub-blocks the 16 rows are (plane i / S, sub-block i % S) and a load instruction covers 16 / S planes
// x 32 S pixels.  The B operand of tap (dy, dx) is the lane's 8 pixels shifted by dy rows and dx columns: per (channel tile, dy) ONE
// 128-bit load of the row y + dy plus the dword on either side; dx = -1 / +1 are v_alignbit_b32 of neighbouring dwords
// (8 bfloat16 shifted by one element), rows / columns outside the plane are masked to zero (the conv's padding).  W % 8 == 0: a
// lane's 8 pixels lie in one image row.  9 x NTR x NTC accumulator tiles (4 registers each) live over all items of a wave.
struct WgBf3Args {
  WgBfSrc rs[3];        // dz rows
  WgBfSrc cs[3];        // tap channels (pass resolution), source after source
  int32_t nrs, ncs;
  int32_t R, C;         // rows; channels of THIS launch = [c_first, c_first + C) of the concatenated sources
  int32_t c_first;
  int32_t dil;          // dilation of the taps (1: gOctConv 3x3; 2, 4, 8, 16: MSBlock slices, csnet.py:116-149) = the kernel instantiation
  int32_t HW, W, H, B;
  int32_t slog, L, runs, nitems, nblk;
  int32_t rows16, k16;
  float* partial;       // [nblk][rows16][k16], column 9 (c_first + ch) + t
};
typedef const CSN_CONST_AS WgBf3Args* WgBf3ArgsP;

#ifdef CSN_EMU_SEQ
template <int NTR, int NTC, int DIL>
__global__ void wgrad_bf16_c3_kernel(WgBf3Args a_byval) {
  const WgBf3Args* a = &a_byval;
  if (threadIdx.x != 0) return;
  const int S = 1 << a->slog, PXS = 32 * S, HW = a->HW, W = a->W, H = a->H, R = a->R, C = a->C;
  std::vector<float> acc((size_t)R * C * 9, 0.f), dz(R);
  for (int wave = 0; wave < 4; ++wave)
    for (int it = blockIdx.x * 4 + wave; it < a->nitems; it += a->nblk * 4) {
      const int b = it / a->runs, run = it - b * a->runs;
      const int q0 = run * a->L * PXS, q1 = min(HW, q0 + a->L * PXS);
      for (int p = q0; p < q1; ++p) {
        const int y = p / W, x = p - y * W;
        for (int r = 0; r < R; ++r) {
          const char* base; unsigned is;
          wgbf_plane(a->rs, a->nrs, r, HW, base, is);
          dz[r] = csn_bf2f(reinterpret_cast<const unsigned short*>(base)[(int64_t)b * is + p]);
        }
        for (int ch = 0; ch < C; ++ch) {
          const char* base; unsigned is;
          wgbf_plane(a->cs, a->ncs, a->c_first + ch, HW, base, is);
          const unsigned short* pl = reinterpret_cast<const unsigned short*>(base) + (int64_t)b * is;
          for (int t = 0; t < 9; ++t) {
            const int yy = y + (t / 3 - 1) * DIL, xx = x + (t % 3 - 1) * DIL;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float v = csn_bf2f(pl[yy * W + xx]);
            for (int r = 0; r < R; ++r) acc[((size_t)r * C + ch) * 9 + t] = fmaf(dz[r], v, acc[((size_t)r * C + ch) * 9 + t]);
          }
        }
      }
    }
  float* out = a->partial + (int64_t)blockIdx.x * a->rows16 * a->k16;
  for (int r = 0; r < R; ++r)
    for (int ch = 0; ch < C; ++ch)
      for (int t = 0; t < 9; ++t) out[(int64_t)r * a->k16 + 9 * (a->c_first + ch) + t] = acc[((size_t)r * C + ch) * 9 + t];
}
#else
typedef csn_f4 wgb_f4;
#ifdef CSN_CPU_EMU
__device__ __forceinline__ unsigned wgbf_ld1(const char* p) { return *reinterpret_cast<const unsigned*>(p); }
__device__ __forceinline__ csn_u2 wgbf_ld2(const char* p) { return *reinterpret_cast<const csn_u2*>(p); }
#else
typedef const __attribute__((address_space(1))) unsigned* wgbf_gp1;
__device__ __forceinline__ unsigned wgbf_ld1(const char* p) { return *(wgbf_gp1)(unsigned long long)p; }

typedef const __attribute__((address_space(1))) csn_u2* wgbf_gp2;
__device__ __forceinline__ csn_u2 wgbf_ld2(const char* p) { return *(wgbf_gp2)(unsigned long long)p; }
#endif

// the side pieces of a row for dilation DIL: the DIL columns left of x and right of x + 8 (DIL <= 8), or the whole shifted vectors
// (DIL = 16); element counts 2 (one dword: also DIL = 1), 2, 4, 8, 8
template <int DIL> struct WgbSide { static constexpr int NDW = DIL <= 2 ? 1 : (DIL == 4 ? 2 : 4); };

template <int NTR, int NTC, int DIL>
__global__ __launch_bounds__(CSN_BLOCK, 2) void wgrad_bf16_c3_kernel(WgBf3Args a_byval) {
  CSN_DYN_SMEM(float, lds);
  WgBf3ArgsP a = CSN_KERNARG(WgBf3Args, a_byval);
  constexpr int NT = NTR + NTC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = csn_readfirstlane(tid >> 6);
  const int sl = a->slog, S = 1 << sl, PXS = 32 << sl, TP = 16 >> sl;
  const int HW = a->HW, W = a->W, H = a->H;
  csn_u2* tabp = reinterpret_cast<csn_u2*>(lds);                 // [NT][64]
  unsigned* tabs = reinterpret_cast<unsigned*>(lds) + NT * 128;  // [NT][64]
  for (int e = tid; e < NT * 64; e += CSN_BLOCK) {
    const int t = e >> 6, l = e & 63;
    const int pl = (l & 15) >> sl;
    const char* base;
    unsigned is;
    if (t < NTR) wgbf_plane(a->rs, a->nrs, min(t * TP + pl, a->R - 1), HW, base, is);
    else wgbf_plane(a->cs, a->ncs, a->c_first + min((t - NTR) * TP + pl, a->C - 1), HW, base, is);
    const unsigned long long bv = (unsigned long long)base;
    csn_u2 v;
    v.x = (unsigned)bv; v.y = (unsigned)(bv >> 32);
    tabp[e] = v;
    tabs[e] = is;
  }
  __syncthreads();
  const int lpix = ((lane >> 4) << (3 + sl)) + ((lane & (S - 1)) << 3);   // this lane's first pixel inside a set
  const float rcpW = 1.0f / (float)W;
  wgb_f4 acc[9][NTR][NTC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
      for (int tc = 0; tc < NTC; ++tc)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][tr][tc][i] = 0.f;

  for (int it = blockIdx.x * 4 + wave; it < a->nitems; it += a->nblk * 4) {
    const int b = it / a->runs, run = it - b * a->runs;
    const int q0 = run * a->L * PXS;
    const int nset = min(a->L, (HW - q0) / PXS);
    const char* pr[NTR];
    const char* pc[NTC];
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      const csn_u2 bp = tabp[t * 64 + lane];
      const unsigned long long base = ((unsigned long long)bp.y << 32) | bp.x;
      pr[t] = reinterpret_cast<const char*>(base + 2ull * ((unsigned long long)b * tabs[t * 64 + lane] + (unsigned)(q0 + lpix)));
    }
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      const csn_u2 bp = tabp[(NTR + t) * 64 + lane];
      const unsigned long long base = ((unsigned long long)bp.y << 32) | bp.x;
      pc[t] = reinterpret_cast<const char*>(base + 2ull * ((unsigned long long)b * tabs[(NTR + t) * 64 + lane]));
    }
    for (int s = 0; s < nset; ++s) {
      const int p = q0 + s * PXS + lpix;
      int y = (int)((float)p * rcpW);
      y -= (y * W > p) ? 1 : 0;
      y += ((y + 1) * W <= p) ? 1 : 0;
      const int x = p - y * W;
      // the pieces left / right of the lane's 8 columns are inside the row as a whole or not at all (x, W and DIL >= 8 are
      // multiples of 8; DIL < 8: the piece ends at x - 1 / starts at x + 8)
      constexpr int SD = DIL < 8 ? 8 : DIL;                      // column distance of the side vectors' far end
      const bool has_l = x - SD >= 0, has_r = x + 8 + SD <= W;
      constexpr int NDW = WgbSide<DIL>::NDW;
      csn_u4 A[NTR], V[NTC][3], Lv[NTC][3], Rv[NTC][3];         // (Lv / Rv: only the first NDW dwords are loaded and used)
#pragma unroll
      for (int t = 0; t < NTR; ++t) A[t] = wgbf_ld(pr[t] + s * (PXS * 2));
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yc = min(max(y + (dy - 1) * DIL, 0), H - 1);   // a row outside the plane: fetched from a valid one, masked below
        const unsigned o = (unsigned)(yc * W + x) * 2u;
        // left piece: the DIL columns ending at x - 1 (DIL <= 8) or the vector at x - 16; right piece: starting at x + 8 / x + 16
        const unsigned lo = has_l ? o - (unsigned)(DIL == 1 ? 4 : 2 * DIL) : o;
        const unsigned ro = has_r ? o + (unsigned)(DIL == 16 ? 32 : 16) : o;
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
          V[t][dy] = wgbf_ld(pc[t] + o);
          if (NDW == 1) { Lv[t][dy].x = wgbf_ld1(pc[t] + lo); Rv[t][dy].x = wgbf_ld1(pc[t] + ro); }
          else if (NDW == 2) {
            const csn_u2 l2 = wgbf_ld2(pc[t] + lo), r2 = wgbf_ld2(pc[t] + ro);
            Lv[t][dy].x = l2.x; Lv[t][dy].y = l2.y; Rv[t][dy].x = r2.x; Rv[t][dy].y = r2.y;
          } else { Lv[t][dy] = wgbf_ld(pc[t] + lo); Rv[t][dy] = wgbf_ld(pc[t] + ro); }
        }
      }
      CSN_SCHED_FENCE();   // every load of the set is in flight before the first contraction
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + (dy - 1) * DIL;
        const unsigned mrow = (yy >= 0 && yy < H) ? 0xffffffffu : 0u;
        const unsigned ml = has_l ? mrow : 0u, mr = has_r ? mrow : 0u;
#pragma unroll
        for (int tc = 0; tc < NTC; ++tc) {
          csn_u4 c0 = V[tc][dy];
          c0.x &= mrow; c0.y &= mrow; c0.z &= mrow; c0.w &= mrow;
          csn_u4 cm, cp;   // columns x - DIL .. x - DIL + 7 and x + DIL .. x + DIL + 7
          if (DIL == 1) {
            const unsigned l = Lv[tc][dy].x & ml, r = Rv[tc][dy].x & mr;
            cm.x = csn_alignbit(c0.x, l, 16); cm.y = csn_alignbit(c0.y, c0.x, 16);
            cm.z = csn_alignbit(c0.z, c0.y, 16); cm.w = csn_alignbit(c0.w, c0.z, 16);
            cp.x = csn_alignbit(c0.y, c0.x, 16); cp.y = csn_alignbit(c0.z, c0.y, 16);
            cp.z = csn_alignbit(c0.w, c0.z, 16); cp.w = csn_alignbit(r, c0.w, 16);
          } else if (DIL == 2) {   // one dword = two columns
            cm.x = Lv[tc][dy].x & ml; cm.y = c0.x; cm.z = c0.y; cm.w = c0.z;
            cp.x = c0.y; cp.y = c0.z; cp.z = c0.w; cp.w = Rv[tc][dy].x & mr;
          } else if (DIL == 4) {
            cm.x = Lv[tc][dy].x & ml; cm.y = Lv[tc][dy].y & ml; cm.z = c0.x; cm.w = c0.y;
            cp.x = c0.z; cp.y = c0.w; cp.z = Rv[tc][dy].x & mr; cp.w = Rv[tc][dy].y & mr;
          } else {                 // 8, 16: whole vectors
            cm = Lv[tc][dy]; cm.x &= ml; cm.y &= ml; cm.z &= ml; cm.w &= ml;
            cp = Rv[tc][dy]; cp.x &= mr; cp.y &= mr; cp.z &= mr; cp.w &= mr;
          }
#pragma unroll
          for (int tr = 0; tr < NTR; ++tr) {
            acc[3 * dy + 0][tr][tc] = csn_mfma_16x16x32_bf16(A[tr], cm, acc[3 * dy + 0][tr][tc]);
            acc[3 * dy + 1][tr][tc] = csn_mfma_16x16x32_bf16(A[tr], c0, acc[3 * dy + 1][tr][tc]);
            acc[3 * dy + 2][tr][tc] = csn_mfma_16x16x32_bf16(A[tr], cp, acc[3 * dy + 2][tr][tc]);
          }
        }
      }
    }
  }

  // ---- the four waves' 16 x 16 tiles through LDS, four tiles per round: sum the S diagonal blocks, write the block's partial
  float* out = a->partial + (int64_t)blockIdx.x * a->rows16 * a->k16;
  const int j = lane & 15, i0 = (lane >> 4) * 4;
  const int k16 = a->k16, Rr = a->R, Cc = a->C, cfirst = a->c_first;
  constexpr int NTILE = 9 * NTR * NTC;
#pragma unroll
  for (int r0 = 0; r0 < NTILE; r0 += 4) {
    __syncthreads();   // (first round: the plane table is no longer read)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tl = r0 + q;
      if (tl < NTILE) {
        const int t9 = tl / (NTR * NTC), tr = (tl / NTC) % NTR, tc = tl % NTC;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) lds[(q * 4 + wave) * 256 + (i0 + reg) * 16 + j] = acc[t9][tr][tc][reg];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tl = r0 + q;
      if (tl < NTILE && tid < TP * TP) {
        const int t9 = tl / (NTR * NTC), tr = (tl / NTC) % NTR, tc = tl % NTC;
        const int r = tid / TP, c = tid - r * TP;
        float v = 0.f;
        for (int w = 0; w < 4; ++w)
          for (int s = 0; s < S; ++s) v += lds[(q * 4 + w) * 256 + (r * S + s) * 16 + c * S + s];
        const int row = tr * TP + r, ch = tc * TP + c;
        if (row < Rr && ch < Cc) out[(int64_t)row * k16 + 9 * (cfirst + ch) + t9] = v;
      }
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------ host side
namespace {
struct WgBfCfg { int slog, ntr, ntc, L, runs, nitems, nblk; bool pool; };

bool wgbf_config(const WgArgs& a, WgBfCfg* c) {
  const bool off = std::getenv("CSN_WGRAD_BF") && std::getenv("CSN_WGRAD_BF")[0] == '0';
  if (off || !a.a16) return false;
  const PwPass& ps = a.ps;
  if (ps.nsrc < 1 || ps.nsrc > 3 || a.nrs < 1 || a.nrs > 3) return false;
  bool own = true, pool = true;
  for (int s = 0; s < ps.nsrc; ++s) {
    own = own && ps.src[s].mode == PW_OWN;
    pool = pool && ps.src[s].mode == PW_POOL2;
  }
  if (!own && !pool) return false;
  const int64_t HW = (int64_t)a.Hr * a.Wr;
  if (HW % 16 != 0 || HW > (1 << 24)) return false;
  if (pool && (a.Wr % 8) != 0) return false;
  const int R = ps.nrows, K = ps.cin;
  if (R < 1 || K < 1) return false;
  for (int sl = 2; sl >= 0; --sl) {
    const int S = 1 << sl, TP = 32 >> sl;
    if (HW % (16 * S) != 0) continue;
    const int ntr = (R + TP - 1) / TP, ntc = (K + TP - 1) / TP;
    if (ntr > 4 || ntc > 4 || ntr * ntc > 8) continue;
    c->slog = sl; c->ntr = ntr; c->ntc = ntc; c->pool = pool;
    const int64_t sets_img = HW / (16 * S), total = sets_img * a.B;
    int64_t L = total / (12 * 4 * WG_MAX_BLOCKS);
    L = L < 1 ? 1 : (L > 16 ? 16 : L);
    c->L = (int)L;
    c->runs = (int)((sets_img + L - 1) / L);
    c->nitems = c->runs * a.B;
    const int nb = (c->nitems + 3) / 4;
    c->nblk = nb < WG_MAX_BLOCKS ? nb : WG_MAX_BLOCKS;
    return true;
  }
  return false;
}

template <int NTR, int NTC>
int wgbf_launch_t(const WgBfArgs& q, bool pool, void* stream) {
  const size_t lds = 4 * 1024 * sizeof(float);   // >= the plane table ((NTR + NTC) * 64 * 12 B <= 6 KB)
  if (pool) CSN_LAUNCH((wgrad_bf16_kernel<NTR, NTC, true>), dim3(q.nblk), dim3(CSN_BLOCK), lds, stream, q);
  else CSN_LAUNCH((wgrad_bf16_kernel<NTR, NTC, false>), dim3(q.nblk), dim3(CSN_BLOCK), lds, stream, q);
  return (int)hipGetLastError();
}
}  // namespace

bool csn_wgrad_bf_eligible(const WgArgs& a) {
  WgBfCfg c;
  return wgbf_config(a, &c);
}

int csn_wgrad_bf_blocks(const WgArgs& a) {
  WgBfCfg c;
  return wgbf_config(a, &c) ? c.nblk : 0;
}

int csn_launch_wgrad_bf(const WgArgs& a, void* stream) {
  WgBfCfg c;
  if (!wgbf_config(a, &c) || c.nblk != a.nblk) return -1;
  WgBfArgs q;
  for (int s = 0; s < 3; ++s) {
    q.rs[s].ptr = s < a.nrs ? a.rs[s].ptr : nullptr; q.rs[s].ctot = s < a.nrs ? a.rs[s].ctot : 0; q.rs[s].n = s < a.nrs ? a.rs[s].n : 0;
    q.cs[s].ptr = s < a.ps.nsrc ? a.ps.src[s].ptr : nullptr; q.cs[s].ctot = s < a.ps.nsrc ? a.ps.src[s].Ctot : 0;
    q.cs[s].n = s < a.ps.nsrc ? a.ps.src[s].C : 0;
  }
  q.nrs = a.nrs; q.ncs = a.ps.nsrc;
  q.R = a.ps.nrows; q.K = a.ps.cin;
  q.HW = a.Hr * a.Wr; q.W = a.Wr; q.B = a.B;
  q.slog = c.slog; q.L = c.L; q.runs = c.runs; q.nitems = c.nitems; q.nblk = c.nblk;
  q.rows16 = a.rows16; q.k16 = a.k16; q.partial = a.partial;
  switch (c.ntr * 10 + c.ntc) {
    case 11: return wgbf_launch_t<1, 1>(q, c.pool, stream);
    case 12: return wgbf_launch_t<1, 2>(q, c.pool, stream);
    case 13: return wgbf_launch_t<1, 3>(q, c.pool, stream);
    case 14: return wgbf_launch_t<1, 4>(q, c.pool, stream);
    case 21: return wgbf_launch_t<2, 1>(q, c.pool, stream);
    case 22: return wgbf_launch_t<2, 2>(q, c.pool, stream);
    case 23: return wgbf_launch_t<2, 3>(q, c.pool, stream);
    case 24: return wgbf_launch_t<2, 4>(q, c.pool, stream);
    case 31: return wgbf_launch_t<3, 1>(q, c.pool, stream);
    case 32: return wgbf_launch_t<3, 2>(q, c.pool, stream);
    case 41: return wgbf_launch_t<4, 1>(q, c.pool, stream);
    case 42: return wgbf_launch_t<4, 2>(q, c.pool, stream);
    default: return -1;
  }
}

// ---- 3x3 tap passes
namespace {
struct WgBf3Cfg { int slog, ntr, ntc, maxch, L, runs, nitems, nblk; };

bool wgbf3_config(const WgArgs& a, WgBf3Cfg* c) {
  const bool off = std::getenv("CSN_WGRAD_BF3") && std::getenv("CSN_WGRAD_BF3")[0] == '0';
  if (off || !a.a16) return false;
  const PwPass& ps = a.ps;
  if (ps.nsrc < 1 || ps.nsrc > 3 || a.nrs < 1 || a.nrs > 3) return false;
  int C = 0;
  bool dilated = false;
  for (int s = 0; s < ps.nsrc; ++s) {
    const int d = ps.src[s].dil;
    if (ps.src[s].mode != PW_TAPS || !(d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) return false;
    dilated = dilated || d != 1;
    C += ps.src[s].C;
  }
  const int64_t HW = (int64_t)a.Hr * a.Wr;
  if ((a.Wr % 8) != 0 || HW % 32 != 0 || HW > (1 << 24) || ps.cin != 9 * C) return false;
  const int R = ps.nrows;
  if (R < 1 || C < 1) return false;
  int best = -1, best_chunks = 1 << 30;
  for (int sl = 2; sl >= 0; --sl) {
    const int S = 1 << sl, TP = 16 >> sl;
    if (HW % (32 * S) != 0) continue;
    const int ntr = (R + TP - 1) / TP;
    if (ntr > 4 || (dilated && ntr > 3)) continue;                // (the dilated instantiations: one channel tile, <= 3 row tiles)
    const int maxch = TP * ((ntr <= 2 && !dilated) ? 2 : 1);
    int chunks = 0;
    if (dilated) for (int q = 0; q < ps.nsrc; ++q) chunks += (ps.src[q].C + maxch - 1) / maxch;   // a launch per slice (one dilation)
    else chunks = (C + maxch - 1) / maxch;
    if (chunks < best_chunks) { best = sl; best_chunks = chunks; }   // (ties: the larger S, i.e. the longer contiguous runs)
  }
  if (best < 0) return false;
  const int S = 1 << best, TP = 16 >> best;
  c->slog = best; c->ntr = (R + TP - 1) / TP;
  c->maxch = TP * ((c->ntr <= 2 && !dilated) ? 2 : 1);
  c->ntc = ((C < c->maxch ? C : c->maxch) + TP - 1) / TP;
  const int64_t sets_img = HW / (32 * S), total = sets_img * a.B;
  int64_t L = total / (12 * 4 * WG_MAX_BLOCKS);
  L = L < 1 ? 1 : (L > 16 ? 16 : L);
  c->L = (int)L;
  c->runs = (int)((sets_img + L - 1) / L);
  c->nitems = c->runs * a.B;
  const int nb = (c->nitems + 3) / 4;
  c->nblk = nb < WG_MAX_BLOCKS ? nb : WG_MAX_BLOCKS;
  return true;
}

template <int NTR, int NTC, int DIL = 1>
int wgbf3_launch_t(const WgBf3Args& q, void* stream) {
  CSN_LAUNCH((wgrad_bf16_c3_kernel<NTR, NTC, DIL>), dim3(q.nblk), dim3(CSN_BLOCK), 4 * 1024 * sizeof(float), stream, q);
  return (int)hipGetLastError();
}
template <int DIL>
int wgbf3_launch_dil(int ntr, const WgBf3Args& q, void* stream) {
  switch (ntr) {
    case 1: return wgbf3_launch_t<1, 1, DIL>(q, stream);
    case 2: return wgbf3_launch_t<2, 1, DIL>(q, stream);
    case 3: return wgbf3_launch_t<3, 1, DIL>(q, stream);
    default: return -1;
  }
}
}  // namespace

bool csn_wgrad_bf3_eligible(const WgArgs& a) {
  WgBf3Cfg c;
  return wgbf3_config(a, &c);
}

int csn_wgrad_bf3_blocks(const WgArgs& a) {
  WgBf3Cfg c;
  return wgbf3_config(a, &c) ? c.nblk : 0;
}

int csn_launch_wgrad_bf3(const WgArgs& a, void* stream) {
  WgBf3Cfg c;
  if (!wgbf3_config(a, &c) || c.nblk != a.nblk) return -1;
  WgBf3Args q;
  int Ctot = 0;
  for (int s = 0; s < 3; ++s) {
    q.rs[s].ptr = s < a.nrs ? a.rs[s].ptr : nullptr; q.rs[s].ctot = s < a.nrs ? a.rs[s].ctot : 0; q.rs[s].n = s < a.nrs ? a.rs[s].n : 0;
    q.cs[s].ptr = s < a.ps.nsrc ? a.ps.src[s].ptr : nullptr; q.cs[s].ctot = s < a.ps.nsrc ? a.ps.src[s].Ctot : 0;
    q.cs[s].n = s < a.ps.nsrc ? a.ps.src[s].C : 0;
    Ctot += q.cs[s].n;
  }
  q.nrs = a.nrs; q.ncs = a.ps.nsrc;
  q.R = a.ps.nrows;
  q.HW = a.Hr * a.Wr; q.W = a.Wr; q.H = a.Hr; q.B = a.B;
  q.slog = c.slog; q.L = c.L; q.runs = c.runs; q.nitems = c.nitems; q.nblk = c.nblk;
  q.rows16 = a.rows16; q.k16 = a.k16; q.partial = a.partial;
  const int TP = 16 >> c.slog;
  bool dilated = false;
  for (int s = 0; s < a.ps.nsrc; ++s) dilated = dilated || a.ps.src[s].dil != 1;
  if (dilated) {   // MSBlock: a launch per slice (its dilation = the instantiation) and channel chunk
    int cb = 0;
    for (int s = 0; s < a.ps.nsrc; ++s) {
      for (int c0 = 0; c0 < a.ps.src[s].C; c0 += c.maxch) {
        q.c_first = cb + c0;
        q.C = a.ps.src[s].C - c0 < c.maxch ? a.ps.src[s].C - c0 : c.maxch;
        q.dil = a.ps.src[s].dil;
        int st = -1;
        switch (q.dil) {
          case 1: st = wgbf3_launch_dil<1>(c.ntr, q, stream); break;
          case 2: st = wgbf3_launch_dil<2>(c.ntr, q, stream); break;
          case 4: st = wgbf3_launch_dil<4>(c.ntr, q, stream); break;
          case 8: st = wgbf3_launch_dil<8>(c.ntr, q, stream); break;
          case 16: st = wgbf3_launch_dil<16>(c.ntr, q, stream); break;
          default: return -1;
        }
        if (st != 0) return st;
      }
      cb += a.ps.src[s].C;
    }
    return 0;
  }
  q.dil = 1;
  for (int c0 = 0; c0 < Ctot; c0 += c.maxch) {   // channel chunks: disjoint columns of the same partial slices
    q.c_first = c0;
    q.C = Ctot - c0 < c.maxch ? Ctot - c0 : c.maxch;
    const int ntc = (q.C + TP - 1) / TP;
    int st = -1;
    switch (c.ntr * 10 + ntc) {
      case 11: st = wgbf3_launch_t<1, 1>(q, stream); break;
      case 12: st = wgbf3_launch_t<1, 2>(q, stream); break;
      case 21: st = wgbf3_launch_t<2, 1>(q, stream); break;
      case 22: st = wgbf3_launch_t<2, 2>(q, stream); break;
      case 31: st = wgbf3_launch_t<3, 1>(q, stream); break;
      case 41: st = wgbf3_launch_t<4, 1>(q, stream); break;
      default: return -1;
    }
    if (st != 0) return st;
  }
  return 0;
}
