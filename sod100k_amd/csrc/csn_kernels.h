// csn_kernels.h -- argument blocks of the csnet kernels + host-side launch prototypes.
#pragma once
#include "csn_device.h"

// ---------------------------------------------------------------------------------------------
// prep: parameter packing jobs (one block per job)
// ---------------------------------------------------------------------------------------------
enum CsnPrepKind {
  CSN_PREP_COPY = 0,      // dst[i] = p0f * src0[i]                                   (n elements)
  CSN_PREP_BN_SCALE = 1,  // dst[i] = gamma[i] / sqrt(var[i] + eps)                   src0=gamma src1=var
  CSN_PREP_BN_SHIFT = 2,  // dst[i] = beta[i] - mean[i] * gamma[i]/sqrt(var[i]+eps)   src0=gamma src1=var src2=beta src3=mean
  CSN_PREP_FILL = 3,      // dst[i] = p0f
  // gOctConv 1x1 block -> rows [nrow][cin4]: dst[r*p2 + p3 + c] = src0[(row0+r)*ld + col0 + c]
  // (p0=ld in floats, p1 = ncol, p2 = dst row stride, p3 = dst col offset; src0 pre-offset to row0/col0)
  CSN_PREP_ROWS = 4,
  // 3x3 block -> [co/8][ci][9][8]: dst[((co/8)*ncol + ci)*72 + t*8 + co%8] = p0f * src0[(co*ld + ci)*9 + t]
  // (n = nrow, p1 = ncol(ci), p0 = ld (total cin of the source weight), p3 = ci offset inside dst, p2 = dst cin total)
  CSN_PREP_C3 = 5,
  CSN_PREP_EYE = 6,       // dst[r*p2 + p3 + r] = p0f  for r < n (identity block inside packed rows)
  // backward-data weights: transposed block with flipped taps
  //   dst[ci*p2 + (p3 & 0xffffff) + co*kk + t] = p0f * src0[co*p0 + ci*kk + (kk-1-t)],  kk = p3 >> 24,
  //   ci < n (rows), co*kk + t < p1 (columns)
  CSN_PREP_ROWS_T = 7,
  CSN_PREP_FLIP9 = 8,     // dst[c*9 + t] = p0f * src0[c*9 + 8 - t]   (n = 9*C)
  // 3x3 block -> tap-major chunks of 16 channels (k_goct_c3.hip): n = rows, p1 = channels C, p2 = dst row pitch, p3 = dst column
  //   dst[r*p2 + p3 + (c/16)*144 + t*16 + c%16] = p0f * src0[r*p0 + c*9 + t]                     (p0 = source row pitch)
  CSN_PREP_C3T = 9,
  // ... of a transposed block with flipped taps (backward data): element (r, c, t) = src0[c*p0 + r*9 + (8 - t)]
  CSN_PREP_C3T_T = 10,
  // 1x1 block -> pw4_kernel's image [k][4][P] (k_pw4.hip): n = rows, p1 = channels, p0 = source row pitch, p2 = P,
  // p3 = t0 | (k0 << 8):  dst[((k0 + c)*4 + (r & 3))*p2 + t0 + (r >> 2)] = p0f * src0[r*p0 + c]
  CSN_PREP_PW4 = 11,
  // 3x3 block -> c3q_kernel's image (k_c3q.hip): n = rows, p1 = channels, p0 = source row pitch, p2 = P, p3 = t0 | (k0 << 8):
  //   dst[((k0 + 9*c + t)*4 + (r & 3))*p2 + t0 + (r >> 2)] = p0f * src0[r*p0 + 9*c + t]
  CSN_PREP_C3Q = 12,
  // transposed 1x1 block (backward data) -> the image of CSN_PREP_PW4: n = rows (= input channels ci), p1 = columns (= output
  // channels co), p0 = source row pitch, p2 = P, p3 = t0 | (k0 << 8):
  //   dst[((k0 + c)*4 + (r & 3))*p2 + t0 + (r >> 2)] = p0f * src0[c*p0 + r]
  CSN_PREP_PW4_T = 13,
  // transposed, tap-flipped 3x3 block (backward data) -> the image of CSN_PREP_C3Q: n = rows (= the forward's input channels ci),
  // p1 = gathered channels (= its output channels co), p0 = source row pitch (cin_tot * 9), p2 = P, p3 = t0 | (k0 << 8):
  //   dst[((k0 + 9*c + t)*4 + (r & 3))*p2 + t0 + (r >> 2)] = p0f * src0[c*p0 + r*9 + (8 - t)]
  CSN_PREP_C3Q_T = 14,
  // MSBlock backward data (ms_dx_kernel): one dilation's [co][ci][3][3] block, tap-flipped, the ci of one (co, tap) contiguous:
  // n = co count, p0 = cin, p2 = padded ci count:   dst[(co*9 + t)*p2 + ci] = p0f * src0[(co*p0 + ci)*9 + (8 - t)]
  CSN_PREP_MSDX = 15,
  // depthwise weights with the eval-mode BatchNorm scale folded in (dw_core.h): n = channels, src0 = w [C][9], src1 = gamma, src2 = running_var,
  // p2 = record pitch, p3 = column offset:   dst[c*p2 + p3 + t] = p0f * src0[c*9 + t] * (gamma[c] / sqrt(var[c] + eps))
  CSN_PREP_DWREC = 16,
};
struct CsnPrepJob {
  int32_t kind, n, p0, p1, p2, p3;
  float p0f;
  int64_t src0, src1, src2, src3;  // float offsets into the caller's arena (-1 unused)
  int64_t dst;                     // float offset into the plan's packed buffer
};

// ---------------------------------------------------------------------------------------------
// depthwise 3x3 (x100 already folded into w9) + BN + PReLU, up to 3 resolution branches per launch
// ---------------------------------------------------------------------------------------------
struct DwBranch {
  const float* in;
  float* out;
  const float* w9;  // [C][9]
  const float* scale;
  const float* shift;
  const float* alpha;
  const float* w9b;      // second unit of a fused pair (dw3x3x2 kernel) -- null otherwise
  const float* scale_b;
  const float* shift_b;
  const float* alpha_b;
  const float* rec = nullptr;   // dw3x3x2_fast_kernel (round 5, dw_core.h): per channel 2 x 12 floats {w'[9] (x100 and the BN scale
                                // folded), shift, alpha, 0} of the pair's first unit, then of its second; null: the round-1 kernel
  const float* xin;      // dw3x3_bwd_kernel: the unit's forward input x (stats then holds the weight-gradient partials [C][NSLAB][9])
  double* stats;         // single-unit kernel, train mode: BN statistics partials [C][CSN_BN_NSLAB][2] of the stored output,
                         // one per (image, tile): slab = b * tiles_x * tiles_y + tile (null: none)
  float* pool;           // dw3x3x2 only: 2x2 average of the pair's output [B*C][H/2][W/2] for the stride-2 unit that
                         // follows (csnet.py:679-680), null = none;  skip_out: that unit is the only reader
  float* pool_mp = nullptr;   // dw3x3x2 only, with `pool`: 2x2 MAXIMUM of those averages [B*C][H/4][W/4] -- the copy c3q_kernel's
                              // high -> low slice of that stride-2 unit reads (F.max_pool2d, csnet.py:708-714); R % 4 == 0 then
  int32_t skip_out;
  int32_t C, H, W;
  int32_t LX, NY, R;          // lanes per row (4 px each), lane rows per block, rows per lane
  int32_t tiles_x, tiles_y;   // tiles per plane
  int32_t blk_end;            // exclusive prefix sum of blocks over branches
  uint32_t m_tiles_y = 0, m_C = 0, m_LX = 0;   // dw3x3x2_fast_kernel: ceil(2^32 / d) of the three divisors (0: d = 1), set by its launcher
  // dw3x3_bwd_kernel with the BatchNorm backward's apply pass fused in (zraw non-null): `in` is then the gradient w.r.t. the
  // unit's OUTPUT y from its first consumer, dy2 the second consumer's (null: none), zraw the saved raw conv output, and
  //   dz = gamma * invstd * (dbn - m1 - (z - mean) * invstd * m2),  dbn = (z * scale + shift > 0 ? 1 : alpha) * (dy + dy2)
  // is formed per loaded element (exactly bn_bwd_apply_kernel's arithmetic) instead of being written and read back
  const float* zraw = nullptr;
  const float* dy2 = nullptr;
  const float* bn_scale = nullptr;   // train-mode folded tables of the unit's own BatchNorm, [C] each
  const float* bn_shift = nullptr;
  const float* bn_alpha = nullptr;
  const float* bn_mean = nullptr;
  const float* bn_invstd = nullptr;
  const float* bn_m1m2 = nullptr;    // [C][2] from bn_bwd_finalize_kernel
  const float* bn_gamma = nullptr;   // [C] in the parameter arena
  // Train mode, input never stored (round 3): `in` (forward kernel) / `xin` (backward kernel) is the PRODUCER's raw conv
  // output z and the unit's input x = PReLU(z * in_scale + in_shift) is formed per loaded element (csn_epi, exactly
  // bn_apply_gap_kernel's arithmetic; positions outside the plane are zero AFTER the transform).  The forward kernel also
  // leaves the plane sums of x that bn_apply_gap_kernel would have taken: one partial per (channel, image, tile) in gapin.
  const float* in_scale = nullptr;   // [C] train-mode folded tables of the producer's BatchNorm
  const float* in_shift = nullptr;
  const float* in_alpha = nullptr;
  double* gapin = nullptr;           // [C][CSN_BN_NSLAB], slab = b * tiles + tile (forward kernel only)
  // ... and the backward kernel, which produces the gradient w.r.t. that never-stored x (its only consumer), also takes the
  // producer's BatchNorm-backward sums over its tile -- bn_bwd_reduce_kernel's three sums, from the z it loads anyway and the dx
  // it has just computed: one partial per (channel, image, tile)
  const float* in_mean = nullptr;    // [C] batch mean / 1 / sqrt(var + eps) of the producer's BatchNorm
  const float* in_invstd = nullptr;
  double* bnred = nullptr;           // [C][CSN_BN_NSLAB][3]
};
struct GapTilesArgs {                // gapabs[c][n] = | sum_tiles gapin[c][n * tiles + t] / HW |
  const double* gapin;
  float* gapabs;                     // [C][S]
  int32_t C, S, tiles, pad;
  int64_t HW;
};
int csn_launch_gap_tiles(const GapTilesArgs& a, void* stream);
// ... of many (unit, branch) pairs in ONE launch (round 4: 66 launches of ~5 us per train-mode forward): the tables are read only by
// the penalty kernel at the end of the forward and by the backward pass
#define CSN_GAP_JOBS 80
struct GapTilesBatch { GapTilesArgs job[CSN_GAP_JOBS]; int32_t n, pad; };
int csn_launch_gap_tiles_batch(const GapTilesArgs* jobs, int njobs, void* stream);
struct DwArgs {
  DwBranch br[3];
  int32_t nbr, B;
  int32_t a16;        // bfloat16 activations (single-unit kernel only; the fused pair is an eval-mode kernel)
  int32_t variant;    // csn_launch_dw_bwd: 1 = the packed-pair form of the fully fused backward (dw3x3_bwd_x_kernel); else 0
};

// ---------------------------------------------------------------------------------------------
// gOctConv 1x1 (+BN+PReLU), all output branches of a unit in one block (see k_goct_pw.hip)
// ---------------------------------------------------------------------------------------------
#define PW_TY0 16
#ifndef PW_TXL
#define PW_TXL 5   // log2 of the tile width of branch 0 in goct_pw_kernel (A/B builds: 6, 7)
#endif
#define PW_TX0 (1 << PW_TXL)
#define PW_KC 16   // gathered channels per LDS panel (32: one load phase less per group but 3 instead of 4 waves/SIMD -- slower)
#define PW_XP 80   // panel row pitch (floats): == 16 (mod 32), so the two k rows a 32-lane half reads hit disjoint banks
#define PW_MAX_PASS 6
#define PW_MAX_GRID 2048
// how a channel slice is brought to the pass resolution; the *_TAPS modes contribute 9 gathered
// entries per channel (k = 9*ch + 3*(dy+1) + (dx+1), zero padded) = a 3x3 convolution as a contraction
enum PwMode { PW_OWN = 0, PW_POOL2 = 1, PW_POOL4 = 2, PW_UP2 = 3, PW_UP4 = 4, PW_TAPS = 5, PW_POOL2_TAPS = 6,
              PW_TAPS_S2 = 7,     // 3x3 taps with stride 2 of a source at twice the resolution (std_conv, csnet.py:751-754)
              PW_TAPS_UPS2 = 8 }; // its adjoint: 3x3 taps of a zero-stuffed source at half the resolution
__host__ __device__ static inline bool pw_mode_taps(int m) { return m == PW_TAPS || m == PW_POOL2_TAPS || m == PW_TAPS_S2 || m == PW_TAPS_UPS2; }
struct PwSrc {
  const float* ptr;  // first channel of the slice inside [B][Ctot][H_s][W_s]
  int32_t C;         // channels of the slice
  int32_t Ctot;      // channels of the whole tensor (image stride)
  int32_t mode;      // PwMode
  int32_t K;         // gathered entries of the slice: C, or 9*C for the *_TAPS modes
  int32_t dil;       // PW_TAPS: dilation (1, 2, 4, 8, 16); padding = dilation
  int32_t pad;
};
struct PwPass {
  int32_t r;          // branch whose pixels this pass walks (resolution H0>>r)
  int32_t nsrc;
  PwSrc src[3];       // channel slices gathered per pixel, in weight-column order
  int32_t cin;        // gathered channels
  int32_t cin4;       // ... rounded up to 4
  int32_t nrows;      // output channels
  int32_t w_off;      // float offset of this pass's rows inside the unit's weight image
  int32_t w_stride;   // row pitch of the weight image (floats): cin4 + 2, conflict-free A-operand reads
  float* out;         // first of the nrows channels written, inside [B][out_ctot][H0>>r][W0>>r]
  int32_t out_ctot;
  int32_t pad2;
  const float* scale;
  const float* shift;
  const float* alpha;
  // row reduction (cls_layer fused into the pass that feeds it): instead of storing its nrows channels the pass
  // stores ONE channel  red_b[0] + sum_r red_w[r] * y_r  (null: normal pass)
  const float* red_w;
  const float* red_b;
};
struct PwArgs {
  PwPass pass[PW_MAX_PASS];
  int32_t npass;
  int32_t H0, W0;      // resolution of branch 0
  int32_t B;
  int32_t tiles_x, tiles_y;
  int32_t ty_log2;     // tile of branch 0 is (1 << ty_log2) rows x 32 columns; 4, 3 or 2
  int32_t wimg_floats; // multiple of 4
  const float* wimg;   // weight image of all passes: rows padded to 16, pitch w_stride, zero filled
  // goct_c3_kernel (single 3x3 pass): the tap-major weight image [rows16][chunk][tap][16 channels] (null: not eligible),
  // its row pitch / size in floats, and the z channel that belongs to output row 0 of the launch (row-chunked launches)
  const float* wimg3;
  int32_t w3_stride, w3_floats, z_c0;
  int32_t a16;         // activations are bfloat16 (bf16 train mode), else float
};

// ---------------------------------------------------------------------------------------------
// gOctConv 1x1 (+BN+PReLU) of a two-branch unit, lane = low pixel + its 2x2 high quad, v_mfma_f32_4x4x1 from the
// load registers (see k_pw4.hip)
// ---------------------------------------------------------------------------------------------
#define PW4_MAX_GROUPS 4
#define PWQ_MAX_GROUPS 8   // pwq_kernel / pwq16_kernel: input gradients of the un-pruned net's 80 .. 160-channel branches (round 6)
#define PW4_FLAT_TWL 7     // Pw4Args / C3qArgs::twl >= 7: flat tiles of 64 consecutive (low) pixels / quads, tiles_y = 1
#define PW4_PITCH(NT4) (((NT4) % 16) == 0 ? (NT4) + 4 : (NT4))   // floats per (channel, row-in-tile) of the weight image
struct Pw4Group { int32_t r0h, nth, r0l, ntl; };   // M group: first high / low output channel (multiples of 4) and row tiles (of 4 channels)
struct Pw4Args {
  const float* xh;     // [B][CH][2 Hl][2 Wl]
  const float* xl;     // [B][CL][Hl][Wl]
  const float* x2;     // [B][C2][Hl / 2][Wl / 2]: third input of a three-branch unit (single-output forms only; null / C2 = 0: none)
  const float* xq;     // [B][CQ][4 Hl][4 Wl]: input two levels finer, 4x4 max-pooled on the way in (low-only form; null / CQ = 0: none)
  float* yh;           // [B][OH][2 Hl][2 Wl]
  float* yl;           // [B][OL][Hl][Wl]   (unused when the unit has no low output)
  const float* red_w;  // row reduction (cls_layer in fuse1x1's epilogue): zero-padded weights per high row; null: rows are stored
  const float* red_b;
  float* logits;       // [B][1][2 Hl][2 Wl]
  const float* wimg;   // [ngroups][CH + CL + C2 + CQ][4][P]: element (g, k, i, t) = W[row 4 t + i of group g's tile list][gathered channel k],
                       // tiles 0 .. nth-1 = high rows, nth .. nth+ntl-1 = low rows, zero padded
  const float* ep_h;   // folded BN / PReLU records {scale, shift, alpha, 0} per high / low output channel (padded to whole tiles)
  const float* ep_l;
  int32_t CH, CL, OH, OL;
  int32_t C2, CQ;
  int32_t Hl, Wl, B;
  int32_t twl;              // log2 of the tile width in low pixels (tile = 2^twl x 64 / 2^twl)
  int32_t tiles_x, tiles_y;
  int32_t ngroups, gimg_floats;
  int32_t nth, ntl;         // instantiation: row tiles per group
  int32_t max_grid;
  int32_t a16;              // activation tensors are bfloat16 (raw launches of the bf16 train mode), else float
  Pw4Group grp[PW4_MAX_GROUPS];
  // round 6 (raw launches over bfloat16 tensors): the BatchNorm statistics of the stored outputs from the epilogue -- per (channel,
  // item tile) the wave's sums of z and z^2 (of the ROUNDED values, fp32 tree over the wave: exact for bfloat16 operands of one
  // magnitude), slab = image * tiles + tile; bn_finalize sums the slabs in a fixed order.  bn_stats_kernel's pass over z is gone.
  double* stats_h = nullptr;   // [OH][stats_stride][2] (null: none)
  double* stats_l = nullptr;   // [OL][stats_stride][2]
  int32_t stats_stride = 0;    // slabs per channel = B * tiles_x * tiles_y
};
bool csn_pw4_pick(int nth, int ntl, int* pnth, int* pntl);
bool csn_pw4_has_stats(int nth, int ntl);
int csn_launch_pw4(const Pw4Args& a, int raw, void* stream);

// ---------------------------------------------------------------------------------------------
// high output of a three-branch 1x1 unit, the low -> high terms as low-resolution products through LDS (k_head.hip, round 6)
// ---------------------------------------------------------------------------------------------
#define HZ_MAX_GROUPS 8
struct HzArgs {
  const float* xh;     // [B][CH][2 H1][2 W1]
  const float* x1;     // [B][C1][H1][W1]
  const float* x2;     // [B][C2][H1 / 2][W1 / 2]
  float* yh;           // [B][OH][2 H1][2 W1]   (rows stored)
  float* part;         // row reduction (red_w): [ngroups][B][2 H1][2 W1], group g's sum of red_w[r] * PReLU(BN(y_r)) over its rows
  const float* red_w;  // cls_layer weights per output row (null: rows are stored)
  const float* wimg;   // [ngroups][CH + C1 + C2][4][P]: pw4_kernel's image (CSN_PREP_PW4), group g = rows [4 nth g, 4 nth (g + 1))
  const float* ep_h;   // {scale, shift, alpha, 0} per output channel, padded to whole groups
  int32_t CH, C1, C2, OH;
  int32_t H1, W1, B;   // H1, W1 even
  int32_t RB;          // rows of x2 per band (4 RB output rows per item)
  int32_t ngroups, nth;
  int32_t hb, nw;      // instantiation: x_0 channels per load batch (2 | 4), waves per block (4 | 8 | 16)
  // set by csn_hz_layout:
  int32_t gimg_floats, nbands;
  int32_t pitch1, plane1, pitch2, plane2;   // LDS planes of z_1 / z_2 (floats): [4 nth][2 RB + 2][W1 + 2], [4 nth][RB + 2][W1 / 2 + 2]
  int32_t off_z1, off_z2;
};
size_t csn_hz_layout(HzArgs& a);   // LDS bytes of an item, 0 = unsupported geometry
bool csn_hz_supported(int nth);
int csn_launch_hz(const HzArgs& a, void* stream);

// ---------------------------------------------------------------------------------------------
// one whole ILBlock (1x1 gOctaveCBR -> depthwise pair) per launch, the block's planes in LDS (small maps; see k_ilb.hip)
// ---------------------------------------------------------------------------------------------
struct IlbArgs {
  const float* xh;     // [B][CH][2 Hl][2 Wl]   block inputs
  const float* xl;     // [B][CL][Hl][Wl]
  float* yh;           // [B][OH][2 Hl][2 Wl]   outputs of conv3x3_2
  float* yl;           // [B][OL][Hl][Wl]       (unused when OL = 0)
  const float* wimg;   // [ng][CH + CL][4][P]: pw4_kernel's image (CSN_PREP_PW4) of every group: tiles 0 .. nth-1 high rows, then low rows
  const float* ep_h;   // conv1x1's {scale, shift, alpha, 0} per high / low output channel, padded to whole groups
  const float* ep_l;
  const float* dwrec_h;  // depthwise records per high / low channel, 2 x 12 floats: {w'[9] (x100 and the BN scale folded), shift, alpha, 0} of
  const float* dwrec_l;  // conv3x3_1, then of conv3x3_2 (dw_core.h); padded to whole groups
  float* pool_h; float* pool_l;   // 2x2 averages of the outputs for a stride-2 unit that follows (null: none) ...
  float* mp_h; float* mp_l;       // ... and the 2x2 maxima of those averages (c3q_kernel's high -> low slice)
  int32_t skip_h, skip_l;         // the full-resolution output has no other reader
  int32_t CH, CL, OH, OL, Hl, Wl, B;
  int32_t ng, gimg_floats, nth, ntl;
  int32_t k3;                     // 3x3 stride-2 entry block: xh = the block input's 2x2 average, xl = its 2x2 maximum (CL = CH), image
                                  // [ng][9 CH][4][P] (CSN_PREP_C3Q)
  int32_t Rh, Rl;                 // rows per depthwise task (even with pool_*, multiple of 4 with mp_*)
  int32_t nthreads;
  int32_t ph, pl, plane_h, plane_l;                               // LDS row pitches / plane sizes (floats), set by csn_ilb_layout
  int32_t off_h1, off_h2, off_l1, off_l2, off_z, off_par, lds_floats;
  int32_t nsh, nrh, nsl, nrl;                                      // strips per row / row chunks per plane of the depthwise tasks
  uint32_t m_hsr, m_hs, m_lsr, m_ls;                               // ... and ceil(2^32 / d) of nsh nrh, nsh, nsl nrl, nsl (0: d = 1)
};
size_t csn_ilb_layout(IlbArgs& a);          // fills the layout fields from (CH, CL, Hl, Wl, nth, ntl, Rh, Rl); LDS bytes, 0 = unsupported
bool csn_ilb_supported(int nth, int ntl);
int csn_launch_ilb(const IlbArgs& a, void* stream);

// ---------------------------------------------------------------------------------------------
// plain 1x1 contraction over own-resolution slices, raw output (input-gradient launches; see k_pwq.hip)
// ---------------------------------------------------------------------------------------------
struct PwqSrc {
  const float* ptr;    // first channel of the slice inside [B][Ctot][HW]
  int32_t C, Ctot;
};
struct PwqArgs {
  PwqSrc src[3];       // gathered entries = the channels of the slices, one after the other
  int32_t nsrc, nrows;
  float* out;          // first of the nrows channels written inside [B][out_ctot][HW]
  const float* wimg;   // [ngroups][K][4][P] (CSN_PREP_PW4 / CSN_PREP_PW4_T), zero padded
  int32_t out_ctot, HW, B;
  int32_t ngroups, gimg_floats, nt, max_grid;
  int32_t a16;         // tensors are bfloat16
  int32_t mfma16;      // ... on pwq16_kernel (v_mfma_f32_4x4x4_16B_bf16, weights of the pass rounded to bfloat16)
  int32_t grp_r0[PWQ_MAX_GROUPS], grp_nt[PWQ_MAX_GROUPS];
  // round 6: the adjoint of a 2x2 max-pool routed in the epilogue (x_i was max-pooled into a lower output branch: the gradient
  // W_ji^T dz_j at the LOW resolution is added to the window's first maximum while the rows are still in the accumulators --
  // maxpool2_bwd_add_pair_kernel's read-modify-write pass over dx is gone).  Needs route_W % 4 == 0 and an even height.
  const float* route_x = nullptr;   // the pooled tensor [B][out_ctot][HW], from the launch's first row on (null: no routing)
  const float* route_t = nullptr;   // the low-resolution gradient [B][out_ctot][HW / 4], same rows
  int32_t route_W = 0;              // row width at the pass resolution
};
int csn_pwq_max_tiles(void);
int csn_launch_pwq(const PwqArgs& a, void* stream);

// ---------------------------------------------------------------------------------------------
// gOctConv 3x3 pass, lane = 2x2 output quad, v_mfma_f32_4x4x1 from the load registers (see k_c3q.hip)
// ---------------------------------------------------------------------------------------------
struct C3qSrc {
  const float* ptr;    // [B][Ctot][H][W] at the pass resolution
  int32_t C, Ctot;
};
struct C3qArgs {
  C3qSrc src[3];       // 3x3 tap slices in weight-column order (gathered entry = 9 * channel + 3 * (dy + 1) + (dx + 1))
  int32_t nsrc;
  int32_t z_ctot;
  const float* z;      // raw tensor [B][z_ctot][H/2][W/2] whose bilinear x2 is added to rows z_c0 + r (null: none)
  int32_t z_c0;
  int32_t out_c0;      // first output channel of the launch inside `out`
  float* out;          // [B][out_ctot][H][W]
  const float* ep;     // {scale, shift, alpha, 0} per output row of the launch, padded to whole tiles of every group
  const float* wimg;   // [ngroups][K][4][P]: element (g, k, i, t) = W[row r0_g + 4 t + i][gathered entry k], zero padded
  int32_t out_ctot, nrows;
  int32_t H, W, B;
  int32_t twl, tiles_x, tiles_y;   // tile = 2^twl x 64 / 2^twl quads
  int32_t ngroups, gimg_floats, nt;
  int32_t max_grid;
  int32_t a16;          // activation tensors (sources, z, out) are bfloat16: raw launches of the bf16 train mode
  int32_t mfma16;       // ... on c3q16_kernel (v_mfma_f32_4x4x4_16B_bf16, weights of the pass rounded to bfloat16)
  int32_t hl;           // float tensors, flat tiles: 62 quads per tile in lanes 1 .. 62, lanes 0 / 63 load their neighbours' halo
                        // (tiles_x = ceil(quads / 62)); edge columns by lane exchange (k_c3q.hip C3qWin)
  int32_t grp_r0[PW4_MAX_GROUPS], grp_nt[PW4_MAX_GROUPS];   // first row / row tiles of every M group
  // round 6, raw (gradient) launches: the adjoint of the 2x2 max-pool routed in the epilogue -- the lane's output quad IS the window:
  // two loads of the pooled tensor's quad rows + one of the low-resolution gradient per output row (see PwqArgs::route_x)
  const float* route_x = nullptr;   // the pooled tensor [B][out_ctot][H][W] (null: no routing)
  const float* route_t = nullptr;   // the low-resolution gradient [B][out_ctot][H / 2][W / 2]
};
int csn_c3q_max_tiles(void);
int csn_launch_c3q(const C3qArgs& a, int raw, void* stream);

// ---------------------------------------------------------------------------------------------
// MSBlock: five dilated 3x3 convs (x100 folded) -> channel concat -> BN -> PReLU (see k_ms.hip)
// ---------------------------------------------------------------------------------------------
struct MsArgs {
  const float* in;     // [B][cin][H][W]
  float* out;          // [B][cout][H][W]
  const float* w[5];   // packed [ceil(dch/8)][cin rounded up to 2][9][8] per dilation (null if absent)
  int32_t dch[5];
  int32_t cobase[5];   // first output channel of each dilation inside the concat
  int32_t cin, cout, H, W, B;
  const float* scale;
  const float* shift;
  const float* alpha;
  int32_t a16, pad;    // bfloat16 activations (per-pixel kernel only)
};

// MSBlock backward data: dx[ci] = sum over the dilations d, their output channels co and the taps t of dz[co][p + off(t) 2^d] w_d[co][ci][8 - t]
struct MsDxArgs {
  const float* dz;     // [B][cout][H][W]: the gradient at the block's pre-activation output (all dilations, concatenated)
  float* dx;           // [B][cin][H][W]
  const float* w[5];   // CSN_PREP_MSDX images [dch rounded up to 2][9][ng * 8] (null if the dilation is absent)
  int32_t dch[5];
  int32_t cobase[5];
  int32_t cin, cout, H, W, B;
  int32_t ng;          // accumulator groups of 8 input channels per lane: ceil(cin / 8) <= 5
  int32_t a16, pad;
};

// ---------------------------------------------------------------------------------------------
// 2x2 average pool of up to 3 tensors; bilinear x2 of one tensor
// ---------------------------------------------------------------------------------------------
struct PoolArgs {
  const float* in[3];
  float* out[3];
  int32_t planes[3];    // B*C
  int32_t Ho[3], Wo[3]; // output size
  int32_t blk_end[3];
  int32_t n;
  int32_t a16, pad;
};
struct Up2Args {
  const float* in;  // [planes][H/2][W/2]
  float* out;       // [planes][H][W]  (always float: the caller's logits)
  int32_t planes, H, W;
  int32_t in16;     // `in` is bfloat16
  // `in` is a sum of partial planes (hz_kernel's row reduction, k_head.hip): value = bias[0] + sum_k in[k * part_stride + .]
  int32_t nparts = 1;
  int64_t part_stride = 0;
  const float* bias = nullptr;
};

// ---------------------------------------------------------------------------------------------
// weight gradient of one contraction pass (see k_wgrad.hip)
// ---------------------------------------------------------------------------------------------
#define WG_MAX_BLOCKS 512
#define WG_MAX_ROWS 80     // output channels per launch (5 MFMA row tiles)
struct WgRows {         // one source of dz rows: `n` consecutive channels starting at `ptr` inside [B][ctot][Hr][Wr]
  const float* ptr;
  int32_t ctot, n;
};
struct WgArgs {
  PwPass ps;            // gather descriptor of the forward pass (src, cin, nrows); out/epilogue fields unused
  WgRows rs[3];         // the ps.nrows rows of dz, source after source (dz of a branch, adjoint-upsampled dz of finer ones)
  int32_t nrs;
  int32_t Hr, Wr, B;
  int32_t gpp;          // 64-pixel groups per image plane
  int32_t ngroups;      // B * gpp
  int32_t nblk;         // blocks (= partial slices)
  int32_t rows16, k16;  // nrows / cin rounded up to 16
  float* partial;       // [nblk][rows16][k16]
  int32_t a16, pad;     // activations (x and dz) are bfloat16
};
struct WgBlock {        // one rectangular block of weight columns inside the reference's weight tensor
  int64_t dst;          // float offset into the gradient arena
  int32_t ld, ncol, col;
  float scale;          // 100 for Conv2dX100 weights (conv2d.py:104), else 1
  int32_t tk;           // > 0: the pass computed dW^T with flipped taps: element (r, col + co*tk + t) belongs to
                        //      grad[dst + co*ld + r*tk + (tk-1-t)]   (MSBlock: gather dz taps, rows = input channels)
};
struct WgReduceArgs {
  const float* partial;
  float* grad;
  WgBlock blk[3];
  int32_t nblocks, nblk, nrows, K, rows16, k16;
};
int csn_launch_wgrad(const WgArgs& a, void* stream);
int csn_wgrad_blocks(const WgArgs& a);   // partial slices (= blocks) the launch will use (ps, rows16, k16, ngroups set)
int csn_launch_wgrad_reduce(const WgReduceArgs& a, void* stream);
#define CSN_WGRED_JOBS 16
struct WgReduceBatch { WgReduceArgs job[CSN_WGRED_JOBS]; int32_t n, pad; };
int csn_launch_wgrad_reduce_batch(const WgReduceArgs* jobs, int njobs, void* stream);   // several passes (own partial regions) per launch
bool csn_wgrad_c3_eligible(const WgArgs& a);                 // k_wgrad_c3.hip: LDS-tiled weight gradient of 3x3 tap passes
int csn_wgrad_c3_blocks(const WgArgs& a);
int csn_launch_wgrad_c3(const WgArgs& a, void* stream);
bool csn_wgrad_bf_eligible(const WgArgs& a);                 // k_wgrad_bf.hip: 1x1 passes of the bf16 train step on the bf16 matrix cores
int csn_wgrad_bf_blocks(const WgArgs& a);
int csn_launch_wgrad_bf(const WgArgs& a, void* stream);
bool csn_wgrad_bf3_eligible(const WgArgs& a);                // ... and the 3x3 tap passes (v_mfma_f32_16x16x32_bf16, shifted operands)
int csn_wgrad_bf3_blocks(const WgArgs& a);
int csn_launch_wgrad_bf3(const WgArgs& a, void* stream);

// ---------------------------------------------------------------------------------------------
// train-mode BatchNorm + PReLU + GAP penalty (see k_train.hip)
// ---------------------------------------------------------------------------------------------
#define CSN_BN_NSLAB 1024
struct BnStatsArgs {
  const float* z;     // [S][C][HW] raw conv output
  double* partial;    // [C][CSN_BN_NSLAB][2]
  int32_t S, C;
  int64_t HW;
  int32_t cpp;        // chunks per plane, set by the launcher
  int32_t a16;        // z is bfloat16
};
struct BnFinalizeArgs {
  const double* partial;
  float* arena;       // parameter arena: gamma/beta read, running stats updated in place
  float* scale;       // folded table of this BN for the apply pass (plan-owned)
  float* shift;
  float* mean;        // batch mean / 1/sqrt(var+eps), kept for the backward pass
  float* invstd;
  int64_t off_weight, off_bias, off_rmean, off_rvar;
  int64_t count;      // S * HW
  int32_t C;
  int32_t S;
  int32_t nslab;      // set by the launcher; > 0 on entry: partials were written by the convolution kernel, that many per channel
  int32_t pstride = 0;// slabs between two channels of `partial` (0: CSN_BN_NSLAB; pw4_kernel's own statistics: nslab)
};
struct BnApplyArgs {
  const float* z;     // raw conv output (kept for the backward pass)
  float* y;           // PReLU(BN(z))
  float* gapabs;      // [C][S] |mean_hw y| per image (required when flop_w != 0)
  const float* scale;
  const float* shift;
  const float* alpha;
  const float* arena;
  double* penalty;    // accumulates 0.5 * flop_w * |mean_hw y| * gamma^2 over (n, c)
  int64_t off_weight;
  int64_t HW;
  int32_t S, C;
  float flop_w;       // 0: not a hooked ILBlock sub-module
  int32_t a16;        // z / y are bfloat16
};
int csn_launch_bn_stats(const BnStatsArgs& a, void* stream);
int csn_launch_bn_finalize(const BnFinalizeArgs& a, void* stream);
int csn_launch_bn_apply(const BnApplyArgs& a, void* stream);
int csn_launch_bn_finalize_n(const BnFinalizeArgs* jobs, int n, void* stream);   // <= 3 branches of one unit in one launch
#define CSN_PEN_JOBS 112
struct BnPenaltyJob {     // one hooked (unit, output branch): its |GAP| table [C][S], gamma offset, branch weight
  const float* gapabs;
  int64_t off_weight;
  int32_t C, S;
  float flop_w;
  int32_t pad;
};
struct BnPenaltyArgs {
  BnPenaltyJob job[CSN_PEN_JOBS];
  const float* arena;
  double* partial;
  int32_t first, n;
};
int csn_launch_bn_penalty(const BnPenaltyJob* jobs, int njobs, const float* arena, double* partial, double* penalty, void* stream);

struct BnBwdArgs {
  const float* dyA;    // gradient w.r.t. the branch output y, from its first consumer
  const float* dyB;    // ... second consumer (null if none)
  float* z;            // in: saved raw conv output, out: dz
  const float* scale;  // train-mode folded tables of the forward pass
  const float* shift;
  const float* alpha;
  const float* mean;
  const float* invstd;
  double* partial;     // [C][CSN_BN_NSLAB][3]
  float* m1m2;         // [C][2] mean(dbn), mean(dbn*xhat)
  const float* arena;  // gamma
  float* grad;         // gradient arena (same offsets as the parameter arena)
  const float* gapabs; // [C][S] from the forward pass (null: no penalty)
  int64_t off_weight, off_bias, off_prelu;
  int64_t HW;
  int32_t S, C;
  float flop_w;        // Oct_bn_hook branch weight (0: not hooked)
  float pen_scale;     // d loss / d (penalty sum) = FLOPS.WEIGHT / batchsize  (train.py:91,210)
  int32_t nslab, cpp;  // set by the launcher
  int32_t a16;         // dy / z are bfloat16
  int32_t skip_apply;  // reduce + finalise only: the consumer of dz forms it on load (dw3x3_bwd_kernel, DwBranch::zraw)
  int32_t nslab_in;    // > 0: `partial` already holds that many partials per channel (written by the depthwise backward of the
                       // activation's only consumer, DwBranch::bnred): no reduce pass
  // round 4: the apply pass also leaves the adjoint of F.interpolate(scale_factor=2, bilinear) of the dz it writes -- the
  // [S][C][H/2][W/2] tensor the low -> high terms of the unit's weight / input gradients read (csn_backward.inl AdjPlan, f = 2) --
  // so that adjup2_pair_kernel does not read dz back (null: none; needs W % 8 == 0 and even H, see csn_bn_bwd_adj2_ok)
  float* adj2 = nullptr;
  float* dz_out = nullptr;   // adj2 only: dz goes HERE, not over z (a lane reads its neighbours' z / dy: in place would race) -- the
                             // planner hands over the buffer of the branch's own activation, which is dead at this point of csn_backward
  int32_t W = 0, pad_ = 0;   // row width of the plane (adj2 only)
};
bool csn_bn_bwd_adj2_ok(int64_t HW, int W);
int csn_launch_bn_bwd(const BnBwdArgs& a, void* stream);
int csn_launch_bn_bwd_reduce(BnBwdArgs& a, void* stream);                       // ... or step by step (<= 3 branches share the finalise launch)
int csn_launch_bn_bwd_finalize_n(const BnBwdArgs* jobs, int n, void* stream);
int csn_launch_bn_bwd_apply_step(const BnBwdArgs& a, void* stream);
int csn_launch_bn_bwd_apply(const BnBwdArgs& a, void* stream);   // the apply pass alone (debug: materialise dz for the probes)

struct DwWgradArgs {
  const float* dz;
  const float* x;
  double* partial;     // [C][CSN_BN_NSLAB][9]
  float* grad;
  int64_t off_w;
  int32_t C, S, H, W;
  int32_t nslab, cpp;  // set by the launcher
  int32_t a16;         // dz / x are bfloat16
};
int csn_launch_dw_wgrad(const DwWgradArgs& a, void* stream);
// finalise pass of many depthwise units in one launch (their partials live in per-unit regions: csn_plan dwwg_off)
#define CSN_DWFIN_JOBS 128
struct DwFinJob { const double* partial; int64_t off_w; int32_t C, nslab; };
struct DwFinBatch { DwFinJob job[CSN_DWFIN_JOBS]; float* grad; int32_t n, pad; };
int csn_launch_dw_wgrad_finalize_batch(const DwFinJob* jobs, int njobs, float* grad, void* stream);

struct AdjUpArgs {
  const float* in;     // [planes][Hl*f][Wl*f]
  float* out;          // [planes][Hl][Wl]
  int32_t planes, Hl, Wl, f;
  int32_t in16, out16;  // element type of in / out: bfloat16 (1) or float (0)
};
int csn_launch_adjup(const AdjUpArgs& a, void* stream);

struct PoolBwdArgs {
  const float* x;      // max-pool: the pooled INPUT [planes][Hl*f][Wl*f]; avg-pool: unused
  const float* t;      // gradient at the low resolution [planes][Hl][Wl]
  float* dx;           // [planes][Hl*f][Wl*f]  (max-pool: accumulated, avg-pool: written)
  int32_t planes, Hl, Wl, f;
  int32_t a16;         // bfloat16 tensors
};
int csn_launch_avgpool2_bwd(const PoolBwdArgs& a, void* stream);
int csn_launch_maxpool_bwd_add(const PoolBwdArgs& a, void* stream);
int csn_launch_sum_to_grad(const float* in, int64_t n, float* dst, double* partial /* >= 512 */, int a16, void* stream);
int csn_launch_to_bf16(const float* in, void* out, int64_t n, void* stream);   // float -> bfloat16 copy, n multiple of 4
int csn_launch_bce(const float* y, const float* t, float* dy, int64_t n, double* loss, void* stream);

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* wd;     // per-element weight decay
  int64_t n;
  float beta1, beta2, eps, step_size, sqrt_bc2;
};
int csn_launch_adam(const AdamArgs& a, void* stream);
int csn_launch_saliency_u8(const float* y, unsigned char* o, int64_t n, void* stream);
int csn_launch_stream_copy(const float* src, float* dst, int64_t n, void* stream);
int csn_launch_sal_hist(const unsigned char* sal, const unsigned char* gt, int64_t npix, int n_images,
                        unsigned long long* hist, unsigned long long* abs_sum, void* stream);
int csn_launch_normalize_nchw(const float* hwc, float* chw, int64_t B, int64_t HW, void* stream);
int csn_launch_resize_normalize(const float* hwc, float* chw, int B, int Hi, int Wi, int H, int W, void* stream);
int csn_launch_saliency_resize_u8(const float* logits, unsigned char* o, int H, int W, int h, int w, void* stream);
int csn_launch_resize_bilinear(const float* in, float* out, int planes, int Hi, int Wi, int Ho, int Wo, void* stream);
int csn_launch_val_mae(const float* logits, int hi, int wi, const float* target, int h, int w, double* mae, void* stream);

// launchers (implemented next to the kernels)
int csn_launch_prep(const CsnPrepJob* jobs_dev, int njobs, const float* arena, float* packed, void* stream);
int csn_launch_dw(const DwArgs& a, void* stream);
int csn_launch_dw2(const DwArgs& a, void* stream);
int csn_launch_dw_bwd(const DwArgs& a, void* stream);   // input + weight gradient of a depthwise unit in one pass
size_t csn_dw2_lds_bytes(const DwArgs& a);
int csn_launch_pw(const PwArgs& a, int raw, void* stream);
bool csn_c3_eligible(const PwArgs& a);                        // one pass of 3x3 tap slices (k_goct_c3.hip)
int csn_launch_c3(const PwArgs& a, int raw, void* stream);   // raw: plain-store instantiation (no BN/PReLU epilogue)
int csn_launch_ms(const MsArgs& a, void* stream);
int csn_launch_ms_dx(const MsDxArgs& a, void* stream);
int csn_launch_pool(const PoolArgs& a, void* stream);
int csn_launch_maxpool(const PoolArgs& a, void* stream);   // 2x2 max instead of the mean (float)
int csn_launch_up2(const Up2Args& a, void* stream);
int csn_launch_nop(void* stream);
int csn_kernels_init(void);  // function attributes (max dynamic LDS)
