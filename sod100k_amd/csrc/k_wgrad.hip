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
#include <algorithm>
#include "pw_gather.h"

#define WG_P 68          // panel pitch (floats)
#define WG_MAX_NT 5      // up to 80 rows per pass

typedef const CSN_CONST_AS WgArgs* WgArgsP;

template <int NT>
__device__ __forceinline__ void wg_mma(const float* dzp, const float* xp, int lane, csn_f4 (&acc)[NT]) {
#ifdef CSN_EMU_SEQ
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
      acc[t] = csn_mfma_16x16x4(av, bv, acc[t]);
    }
  }
#endif
}

// Not inlined on purpose: hipcc otherwise keeps the address arithmetic of every gather mode of all three call
// sites live at once (390 VGPRs, one wave per SIMD); as a call the kernel needs 125.
template <typename AT, int MODES>
__device__ __attribute__((noinline)) void wg_gather(PwPassP ps, int s, int c_lo, int c_hi, float* xrow, int rmax, int b,
                                                   int y, int x, int Hr, int Wr, unsigned pair_elem) {
#ifdef CSN_CPU_EMU
  pw_gather_slice<AT, WG_P, MODES>(ps, s, c_lo, c_hi, xrow, rmax, b, y, x, Hr, Wr, pair_elem);
#else
  // arguments of a real call travel in VGPRs: without these the image index (and with it every buffer resource) counts as
  // divergent and each gather load turns into a waterfall loop
  {
    const unsigned long long pv = (unsigned long long)ps;
    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)pv), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    ps = (PwPassP)(((unsigned long long)hi32 << 32) | lo32);
  }
  s = __builtin_amdgcn_readfirstlane(s); c_lo = __builtin_amdgcn_readfirstlane(c_lo); c_hi = __builtin_amdgcn_readfirstlane(c_hi);
  rmax = __builtin_amdgcn_readfirstlane(rmax); b = __builtin_amdgcn_readfirstlane(b);
  Hr = __builtin_amdgcn_readfirstlane(Hr); Wr = __builtin_amdgcn_readfirstlane(Wr);
  // xrow points into the block's LDS panel, but a generic pointer parameter of a real call compiles to flat_store (counted in
  // vmcnt AND lgkmcnt: every panel write then waits for the gather's global loads): go through the LDS address space
  typedef __attribute__((address_space(3))) float* lds_fp;
  pw_gather_slice<AT, WG_P, MODES, lds_fp>(ps, s, c_lo, c_hi, (lds_fp)xrow, rmax, b, y, x, Hr, Wr, pair_elem);
#endif
}

// two smaller functions instead of one with every mode: the all-mode body ran out of SGPRs in the bf16 instantiation (the
// return address then goes through a scratch-saved VGPR on every call)
template <typename AT>
__device__ __forceinline__ void wg_gather_any(PwPassP ps, int s, int c_lo, int c_hi, float* xrow, int rmax, int b, int y,
                                              int x, int Hr, int Wr, unsigned pair_elem) {
  if (pw_mode_taps(ps->src[s].mode)) wg_gather<AT, 2>(ps, s, c_lo, c_hi, xrow, rmax, b, y, x, Hr, Wr, pair_elem);
  else wg_gather<AT, 1>(ps, s, c_lo, c_hi, xrow, rmax, b, y, x, Hr, Wr, pair_elem);
}

template <int NT, typename AT>
__global__ __launch_bounds__(CSN_BLOCK, 3) void goct_wgrad_kernel(WgArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  WgArgsP a = CSN_KERNARG(WgArgs, a_byval);
  PwPassP ps = &a->ps;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CSN_CPU_EMU
  const int wave = tid >> 6;
#else
  // wave-uniform by construction, but the compiler cannot know: without this every buffer resource derived from the wave's
  // group index lives in VGPRs and each load becomes a waterfall loop with a full vmcnt(0) wait
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int rows16 = a->rows16, k16 = a->k16, nrows = ps->nrows;
  const int Hr = a->Hr, Wr = a->Wr, HW = Hr * Wr;
  float* dzp = lds;                                       // [rows16][WG_P], shared by the block
  float* xp = lds + rows16 * WG_P + wave * (16 * WG_P);   // this wave's x[16 k][64 px] panel
  for (int i = tid; i < rows16 * WG_P; i += CSN_BLOCK) dzp[i] = 0.f;   // rows nrows..rows16 stay zero
  const int cin = ps->cin;
  const int c1 = ps->src[0].K, c2 = c1 + ps->src[1].K;
  const int nsweep = (k16 / 16 + 3) >> 2;
  for (int sw = 0; sw < nsweep; ++sw) {
    const int kc = (4 * sw + wave) * 16;
    const bool active = kc < k16;
    csn_f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    for (int g = blockIdx.x; g < a->ngroups; g += a->nblk) {
      const int b = g / a->gpp;
      const int gp0 = (g - b * a->gpp) * 64;    // first pixel of the group (uniform)
      const int p = gp0 + lane;
      // pixel pair of this lane in the paired gathers (clamped into the plane; planes with an odd pixel count: none)
      const unsigned pair_el = (HW & 1) ? CSN_NO_PAIR : min((unsigned)gp0 + 2u * (lane & 31), (unsigned)HW - 2u);
      const bool valid = p < HW;
      const int pc = valid ? p : HW - 1;
      const int y = pc / Wr, x = pc - y * Wr;
      __syncthreads();   // previous group's dz panel fully consumed
      {
        int rb = 0;
        for (int q = 0; q < a->nrs; ++q) {
          const int nq = a->rs[q].n;
          const AT* ap = act_cast<AT>(a->rs[q].ptr) + (int64_t)b * a->rs[q].ctot * HW + pc;
          for (int r = wave; r < nq; r += 4) {
            const float v = act_ld(ap + (int64_t)r * HW);
            dzp[(rb + r) * WG_P + lane] = valid ? v : 0.f;   // pixels past the plane contribute nothing
          }
          rb += nq;
        }
      }
      __syncthreads();
      if (active) {
        const int kend = min(kc + 16, cin);
        CSN_WAVE_SYNC();
        if (kc < c1) wg_gather_any<AT>(ps, 0, kc, min(kend, c1), xp + lane, 16, b, y, x, Hr, Wr, pair_el);
        if (max(kc, c1) < min(kend, c2)) {
          const int r0 = max(kc, c1) - kc;
          wg_gather_any<AT>(ps, 1, max(kc, c1) - c1, min(kend, c2) - c1, xp + r0 * WG_P + lane, 16 - r0, b, y, x, Hr, Wr, pair_el);
        }
        if (max(kc, c2) < min(kend, cin)) {
          const int r0 = max(kc, c2) - kc;
          wg_gather_any<AT>(ps, 2, max(kc, c2) - c2, min(kend, cin) - c2, xp + r0 * WG_P + lane, 16 - r0, b, y, x, Hr, Wr, pair_el);
        }
        for (int k = max(cin, kc); k < kc + 16; ++k) xp[(k - kc) * WG_P + lane] = 0.f;
        CSN_WAVE_SYNC();
        wg_mma<NT>(dzp, xp, lane, acc);
      }
    }
    if (active) {
      float* out = a->partial + (int64_t)blockIdx.x * rows16 * k16 + kc + (lane & 15);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(int64_t)(16 * t + (lane >> 4) * 4 + i) * k16] = acc[t][i];
    }
  }
}

// Variant for few, small passes (K <= 64 gathered channels, <= 48 rows: every 1x1 unit of stages 0-3): each wave
// owns its pixel groups AND all k chunks, with a private dz panel -- no block barrier in the loop, no idle wave
// when K < 64.  The four waves' accumulators are added through LDS once, at the end.
template <int NT, int NCH, typename AT>
__global__ __launch_bounds__(CSN_BLOCK, 3) void goct_wgrad_wave_kernel(WgArgs a_byval) {
  CSN_DYN_SMEM(float, lds);
  WgArgsP a = CSN_KERNARG(WgArgs, a_byval);
  PwPassP ps = &a->ps;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef CSN_CPU_EMU
  const int wave = tid >> 6;
#else
  // wave-uniform by construction, but the compiler cannot know: without this every buffer resource derived from the wave's
  // group index lives in VGPRs and each load becomes a waterfall loop with a full vmcnt(0) wait
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int rows16 = a->rows16, k16 = a->k16, nrows = ps->nrows;
  const int Hr = a->Hr, Wr = a->Wr, HW = Hr * Wr;
  const int wfloats = (rows16 + 16) * WG_P;
  float* dzp = lds + wave * wfloats;          // this wave's dz[rows16][64 px]
  float* xp = dzp + rows16 * WG_P;            // ... and x[16 k][64 px]
  for (int i = lane; i < rows16 * WG_P; i += 64) dzp[i] = 0.f;   // rows nrows..rows16 stay zero
  const int cin = ps->cin;
  const int c1 = ps->src[0].K, c2 = c1 + ps->src[1].K;
  csn_f4 acc[NCH][NT];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[c][t][i] = 0.f;
  const unsigned cs4 = (unsigned)HW * (unsigned)sizeof(AT);
  // block-uniform trip count (the CPU emulation maps the wave hand-offs to block barriers): a wave without a group
  // left walks the last group with an all-zero dz panel
  for (int g0 = blockIdx.x * 4; g0 < a->ngroups; g0 += a->nblk * 4) {
    const int g = min(g0 + wave, a->ngroups - 1);
    const int b = g / a->gpp;
    const int gp0 = (g - b * a->gpp) * 64;      // first pixel of the group (wave-uniform)
    const int p = gp0 + lane;
    const unsigned pair_el = (HW & 1) ? CSN_NO_PAIR : min((unsigned)gp0 + 2u * (lane & 31), (unsigned)HW - 2u);
    const bool valid = p < HW && g0 + wave < a->ngroups;
    const int pc = valid ? p : HW - 1;
    const int y = pc / Wr, x = pc - y * Wr;
    CSN_WAVE_SYNC();   // previous group's panels fully consumed
    {
      int rbase = 0;
      for (int q = 0; q < a->nrs; ++q) {
        const int nq = a->rs[q].n;
        for (int r0 = 0; r0 < nq; r0 += 16) {
          const csn_buf rb = csn_make_buf(act_cast<AT>(a->rs[q].ptr) + ((int64_t)b * a->rs[q].ctot + r0) * HW);
          if (CSN_PAIR_GATHER(AT) && (HW & 1) == 0) {   // two pixels per lane, two rows per load (pw_batch_own_pair)
            constexpr unsigned E = (unsigned)sizeof(AT);
            const int pq = lane & 31, half = lane >> 5, nn = min(16, nq - r0);
            const unsigned pb = min((unsigned)gp0 + 2u * pq, (unsigned)HW - 2u);
            pw_batch_own_pair<AT, 16, WG_P>(rb, pb * E + ((nn >= 2 && half) ? (unsigned)HW * E : 0u), half, (unsigned)HW * E, 0, nn,
                                            16, dzp + (rbase + r0) * WG_P + 2 * pq);
          } else {
            pw_batch_own<AT, 16, WG_P>(rb, (unsigned)pc * (unsigned)sizeof(AT), cs4, 0, min(16, nq - r0), 16,
                                       dzp + (rbase + r0) * WG_P + lane);
          }
        }
        rbase += nq;
      }
    }
    CSN_WAVE_SYNC();   // the paired loads wrote other lanes' pixels
    if (!valid)   // pixels past the plane contribute nothing
      for (int r = 0; r < nrows; ++r) dzp[r * WG_P + lane] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int kc = 16 * c;
      if (kc < k16) {
        const int kend = min(kc + 16, cin);
        CSN_WAVE_SYNC();
        if (kc < c1) wg_gather_any<AT>(ps, 0, kc, min(kend, c1), xp + lane, 16, b, y, x, Hr, Wr, pair_el);
        if (max(kc, c1) < min(kend, c2)) {
          const int r0 = max(kc, c1) - kc;
          wg_gather_any<AT>(ps, 1, max(kc, c1) - c1, min(kend, c2) - c1, xp + r0 * WG_P + lane, 16 - r0, b, y, x, Hr, Wr, pair_el);
        }
        if (max(kc, c2) < min(kend, cin)) {
          const int r0 = max(kc, c2) - kc;
          wg_gather_any<AT>(ps, 2, max(kc, c2) - c2, min(kend, cin) - c2, xp + r0 * WG_P + lane, 16 - r0, b, y, x, Hr, Wr, pair_el);
        }
        for (int k = max(cin, kc); k < kc + 16; ++k) xp[(k - kc) * WG_P + lane] = 0.f;
        CSN_WAVE_SYNC();
        wg_mma<NT>(dzp, xp, lane, acc[c]);
      }
    }
  }
  // ---- add the four waves' tiles and write the block's partial
  __syncthreads();
  float* comb = lds + wave * wfloats;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) comb[(c * NT + t) * 256 + ((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc[c][t][i];
  __syncthreads();
  float* out = a->partial + (int64_t)blockIdx.x * rows16 * k16;
  const int row = tid >> 4, col = tid & 15;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (16 * c >= k16) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int o = (c * NT + t) * 256 + tid;
      const float v = (lds[o] + lds[wfloats + o]) + (lds[2 * wfloats + o] + lds[3 * wfloats + o]);
      out[(int64_t)(16 * t + row) * k16 + 16 * c + col] = v;
    }
  }
}

// grad[dst + r*ld + c] = scale * sum_blk partial[blk][r][col + c]  for every block of weight columns.
// Block = 64 consecutive k (coalesced) x 4 interleaved slices of the partial index; grid (ceil(K/64), nrows).
__global__ __launch_bounds__(CSN_BLOCK) void wgrad_reduce_kernel(WgReduceArgs a) {
  CSN_DYN_SMEM(double, sm);
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane, r = blockIdx.y;
  double s = 0.0;
  if (k < a.K) {
    const float* p = a.partial + (int64_t)r * a.k16 + k;
    const int64_t stride = (int64_t)a.rows16 * a.k16;
#pragma unroll 8
    for (int b = grp; b < a.nblk; b += 4) s += (double)p[b * stride];   // independent loads: keep eight in flight
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (grp != 0 || k >= a.K) return;
  s = (sm[lane] + sm[64 + lane]) + (sm[128 + lane] + sm[192 + lane]);
  int bi = -1;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q < a.nblocks && k >= a.blk[q].col && k < a.blk[q].col + a.blk[q].ncol) bi = q;
  if (bi < 0) return;   // identity columns (already convolved partial sums) carry no parameter
  const int c = k - a.blk[bi].col, tk = a.blk[bi].tk;
  const int64_t idx = tk > 0 ? (int64_t)(c / tk) * a.blk[bi].ld + (int64_t)r * tk + (tk - 1 - c % tk)
                             : (int64_t)r * a.blk[bi].ld + c;
  a.grad[a.blk[bi].dst + idx] = (float)((double)a.blk[bi].scale * s);
}

// several passes per launch: grid (max ceil(K / 64), max nrows, jobs) -- the same per-element arithmetic and order as above
__global__ __launch_bounds__(CSN_BLOCK) void wgrad_reduce_jobs_kernel(WgReduceBatch bt) {
  CSN_DYN_SMEM(double, sm);
  const CSN_CONST_AS WgReduceArgs* a = &CSN_KERNARG(WgReduceBatch, bt)->job[blockIdx.z];   // (read in place: no scratch copy)
  const int K = a->K, k16 = a->k16, nblk = a->nblk;
  if ((int)blockIdx.y >= a->nrows || (int)blockIdx.x * 64 >= K) return;   // (block-uniform)
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane, r = blockIdx.y;
  double s = 0.0;
  if (k < K) {
    const float* p = a->partial + (int64_t)r * k16 + k;
    const int64_t stride = (int64_t)a->rows16 * k16;
#pragma unroll 8
    for (int b = grp; b < nblk; b += 4) s += (double)p[b * stride];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (grp != 0 || k >= K) return;
  s = (sm[lane] + sm[64 + lane]) + (sm[128 + lane] + sm[192 + lane]);
  int bi = -1;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q < a->nblocks && k >= a->blk[q].col && k < a->blk[q].col + a->blk[q].ncol) bi = q;
  if (bi < 0) return;
  const int c = k - a->blk[bi].col, tk = a->blk[bi].tk, ld = a->blk[bi].ld;
  const int64_t idx = tk > 0 ? (int64_t)(c / tk) * ld + (int64_t)r * tk + (tk - 1 - c % tk) : (int64_t)r * ld + c;
  a->grad[a->blk[bi].dst + idx] = (float)((double)a->blk[bi].scale * s);
}

// wave-private variant: 4 groups per block step; returns false when the pass does not fit it
static bool wgrad_wave_fits(const WgArgs& a) { return a.k16 <= 64 && a.rows16 <= 48; }

int csn_wgrad_blocks(const WgArgs& a) {
  if (csn_wgrad_bf3_eligible(a)) return csn_wgrad_bf3_blocks(a);
  if (csn_wgrad_c3_eligible(a)) return csn_wgrad_c3_blocks(a);
  if (csn_wgrad_bf_eligible(a)) return csn_wgrad_bf_blocks(a);
  const int units = wgrad_wave_fits(a) ? (a.ngroups + 3) / 4 : a.ngroups;
  return units < WG_MAX_BLOCKS ? units : WG_MAX_BLOCKS;
}

template <typename AT>
static int launch_wgrad_t(const WgArgs& a, void* stream) {
  if (wgrad_wave_fits(a)) {
    const int nt = a.rows16 >> 4, nch = a.k16 <= 32 ? 2 : 4;
    const size_t wl = (size_t)4 * (a.rows16 + 16) * WG_P * sizeof(float);
    const dim3 grid(a.nblk), block(CSN_BLOCK);
    if (nt == 1 && nch == 2) CSN_LAUNCH((goct_wgrad_wave_kernel<1, 2, AT>), grid, block, wl, stream, a);
    else if (nt == 1) CSN_LAUNCH((goct_wgrad_wave_kernel<1, 4, AT>), grid, block, wl, stream, a);
    else if (nt == 2 && nch == 2) CSN_LAUNCH((goct_wgrad_wave_kernel<2, 2, AT>), grid, block, wl, stream, a);
    else if (nt == 2) CSN_LAUNCH((goct_wgrad_wave_kernel<2, 4, AT>), grid, block, wl, stream, a);
    else CSN_LAUNCH((goct_wgrad_wave_kernel<3, 4, AT>), grid, block, wl, stream, a);
    return (int)hipGetLastError();
  }
  const size_t lds = ((size_t)a.rows16 * WG_P + 4 * 16 * WG_P) * sizeof(float);
  switch (a.rows16 >> 4) {
    case 1: CSN_LAUNCH((goct_wgrad_kernel<1, AT>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a); break;
    case 2: CSN_LAUNCH((goct_wgrad_kernel<2, AT>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a); break;
    case 3: CSN_LAUNCH((goct_wgrad_kernel<3, AT>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a); break;
    case 4: CSN_LAUNCH((goct_wgrad_kernel<4, AT>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a); break;
    default: CSN_LAUNCH((goct_wgrad_kernel<5, AT>), dim3(a.nblk), dim3(CSN_BLOCK), lds, stream, a); break;
  }
  return (int)hipGetLastError();
}

int csn_launch_wgrad(const WgArgs& a, void* stream) {
  if (csn_wgrad_bf3_eligible(a)) return csn_launch_wgrad_bf3(a, stream);   // bf16 tensors, 3x3 taps: shifted operands (k_wgrad_bf.hip)
  if (csn_wgrad_c3_eligible(a)) return csn_launch_wgrad_c3(a, stream);   // 3x3 tap slices: LDS-tiled (k_wgrad_c3.hip)
  if (csn_wgrad_bf_eligible(a)) return csn_launch_wgrad_bf(a, stream);   // bf16 tensors, 1x1: operands straight from the loads (k_wgrad_bf.hip)
  if (a.rows16 > 16 * WG_MAX_NT) return -1;   // (the generic kernels: five row tiles; the bf16 forms above have their own limits)
  return a.a16 ? launch_wgrad_t<csn_bf16>(a, stream) : launch_wgrad_t<float>(a, stream);
}

int csn_launch_wgrad_reduce_batch(const WgReduceArgs* jobs, int njobs, void* stream) {
  for (int first = 0; first < njobs; first += CSN_WGRED_JOBS) {
    WgReduceBatch b;
    b.n = njobs - first < CSN_WGRED_JOBS ? njobs - first : CSN_WGRED_JOBS; b.pad = 0;
    int gx = 1, gy = 1;
    for (int i = 0; i < b.n; ++i) {
      b.job[i] = jobs[first + i];
      gx = std::max(gx, (jobs[first + i].K + 63) / 64);
      gy = std::max(gy, (int)jobs[first + i].nrows);
    }
    CSN_LAUNCH(wgrad_reduce_jobs_kernel, dim3(gx, gy, b.n), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, b);
  }
  return (int)hipGetLastError();
}

int csn_launch_wgrad_reduce(const WgReduceArgs& a, void* stream) {
  CSN_LAUNCH(wgrad_reduce_kernel, dim3((a.K + 63) / 64, a.nrows), dim3(CSN_BLOCK), CSN_BLOCK * sizeof(double), stream, a);
  return (int)hipGetLastError();
}
