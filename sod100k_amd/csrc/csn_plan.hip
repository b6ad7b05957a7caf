// csn_plan.hip -- host side of libcsnet_hip.so: plan construction, parameter packing jobs, workspace
// layout and the launch sequence of CSNet.forward (CSNet/model/csnet.py:365-387) behind the C ABI of
// include/csnet_hip.h.  No torch types; the caller owns every tensor.
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "../../include/csnet_hip.h"
#include "csn_kernels.h"

namespace {

thread_local std::string g_hip_err;
thread_local std::string g_why;
#define FAIL(code, why) do { g_why = (why); return (code); } while (0)

int hip_fail(hipError_t e, const char* what) {
  g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
  return CSN_E_HIP;
}
#define HIP_TRY(expr)                                     \
  do {                                                    \
    hipError_t _e = (expr);                               \
    if (_e != hipSuccess) return hip_fail(_e, #expr);     \
  } while (0)
#define LAUNCH_TRY(expr)                                                  \
  do {                                                                    \
    int _e = (expr);                                                      \
    if (_e < 0) { g_hip_err = #expr ": no kernel instantiation"; return CSN_E_UNSUPPORTED; } \
    if (_e != 0) return hip_fail((hipError_t)_e, #expr);                  \
  } while (0)

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int round4(int v) { return (v + 3) & ~3; }

struct Act {
  int channels = 0, lvl = 0;
  int64_t ws_off = -1;  // bytes; -1 = external input
};

struct Epi {  // float offsets into the packed buffer
  int64_t scale = -1, shift = -1, alpha = -1;
  int64_t dummy = -1;   // identity tables only: a scratch row for outputs nobody reads
};

// ---- plan of the panel + MFMA contraction kernel (k_goct_pw.hip) --------------------------------------
// sources: unit input branch (pooled copy when stride 2) / unit-private scratch / backward: dz of an output branch,
// adjoint-upsampled dz (unit scratch), the gradient buffer being accumulated
enum SrcKind { SRC_IN = 0, SRC_Z = 1, SRC_DZ = 2, SRC_ADJ = 3, SRC_DX = 4 };
enum OutKind { OUT_ACT = 0, OUT_Z = 1, OUT_LOGITS = 2, OUT_DX = 3, OUT_TMP = 4 };

struct WBlock {           // one rectangular block of a pass's weight rows
  int eye = 0;            // 1: identity block (adds an already convolved tensor through the contraction)
  int64_t src = -1;       // arena offset of W[row 0][col 0] of the block
  int ld = 0, ncol = 0, col = 0;
  float scale = 1.f;
  int tk = 0;             // > 0: transposed block with flipped taps (backward data), tk = k*k of the forward weight
  int eye_rows = 0, eye_col0 = 0;   // identity block of a row chunk: rows in the chunk, first row of the chunk
};

struct PwPassPlan {
  int r = 0;                           // resolution of the pass relative to the launch (H0 >> r)
  int nsrc = 0;
  int src_kind[3] = {0, 0, 0};
  int src_branch[3] = {-1, -1, -1};
  int src_C[3] = {0, 0, 0};
  int src_mode[3] = {0, 0, 0};         // PwMode
  int src_dil[3] = {1, 1, 1};
  int src_c0[3] = {0, 0, 0};           // first channel of the slice inside its tensor
  int src_ctot[3] = {0, 0, 0};         // channels of the whole tensor (0: == src_C)
  int K = 0, K4 = 0, nrows = 0;
  int w_off = 0, w_stride = 0;         // inside the launch's weight image
  int out_kind = OUT_ACT, out_branch = 0, out_c0 = 0, out_ctot = 0;
  Epi epi;                             // tables already offset to the first row of the pass
  csn_bn_off bn = {-1, -1, -1, -1, -1};  // arena offsets of the BN / PReLU tensors of row 0 (OUT_ACT passes; c3q records)
  std::vector<WBlock> wb;
};

struct PwLaunchPlan {
  std::vector<PwPassPlan> passes;
  int lvl = 0;                         // activation level (H >> lvl) of r = 0
  int64_t wimg = -1;                   // packed-buffer offset of the weight image
  int wimg_floats = 0;
  // goct_c3_kernel's tap-major image of a single 3x3 pass (k_goct_c3.hip); -1: the launch does not qualify
  int64_t wimg3 = -1;
  int w3_stride = 0, w3_floats = 0, z_c0 = 0;
  // c3q_kernel's plan of the same kind of launch (k_c3q.hip, forward only); c3q = 0: the launch does not qualify
  int c3q = 0, c3q_nt = 0, c3q_ng = 0, c3q_gimg = 0, c3q_ntap = 0, c3q_z = 0;
  int c3q_r0[PW4_MAX_GROUPS] = {0, 0, 0, 0}, c3q_gnt[PW4_MAX_GROUPS] = {0, 0, 0, 0};
  int64_t c3q_wimg = -1, c3q_ep = -1;
  // pwq_kernel's plan (k_pwq.hip): one pass of own-resolution slices, raw output (input-gradient launches); 0: not eligible
  int pwq = 0, pwq_nt = 0, pwq_ng = 0, pwq_gimg = 0;
  int pwq_r0[PWQ_MAX_GROUPS] = {}, pwq_gnt[PWQ_MAX_GROUPS] = {};
  int64_t pwq_wimg = -1;
};

struct UnitPlan {
  csn_unit_desc d;
  int base_lvl = 0;                      // lvl of branch 0 of the unit's compute resolution
  int64_t pooled_off[3] = {-1, -1, -1};  // workspace byte offsets of the 2x2 avg-pooled inputs (stride 2)
  int64_t mp_off[3] = {-1, -1, -1};      // ... of the 2x2 max-pooled inputs (3x3 high -> low slices of c3q_kernel)
  int64_t z_off = -1;                    // workspace byte offset of the 3x3 low->high partial sums
  int z_C = 0;
  int64_t logits_off = -1;               // CLS: logits at H/2
  std::vector<PwLaunchPlan> pwl;         // GOCT / MS / CLS
  // DW
  int64_t dw_w[3] = {-1, -1, -1};
  Epi dw_epi[3];
  int fuse_next = 0;   // DW: the next unit is the second depthwise unit of the same ILBlock
  int64_t dw2rec[3] = {-1, -1, -1};   // ... and the pair's folded per-channel records (dw3x3x2_fast_kernel, dw_core.h), per branch
  int fuse_cls = 0;    // GOCT: the next unit is the cls_layer and nobody else reads this unit's output
  int std_conv = 0;    // GOCT 1 -> 1: Conv2dX100 (x100 weights, real stride 2)
  int c3 = 0;          // GOCT 3x3: every launch of the unit qualifies for the LDS-tiled kernel (profile attribution)
  int pool_unit = -1;  // DW (first of a fused pair): index of the stride-2 unit whose pooled inputs the pair writes
  int pool_skip[3] = {0, 0, 0};   // ... and that unit is the only reader of branch i (full-resolution store skipped)
  int pooled_by_producer = 0;     // GOCT stride 2: the preceding fused depthwise pair delivers the pooled inputs
  int mp_producer = -1;           // ... index of that pair's first unit
  Epi out_epi[3];                        // folded BN/PReLU tables of every output branch (train mode rewrites them)
  int64_t stats_off[3] = {-1, -1, -1};   // workspace byte offsets of the BN statistics partials
  // training (csn_plan_enable_training): per output branch batch mean / invstd / backward means (packed offsets),
  // per-image |GAP| table (workspace bytes), consumer slot of every input branch in its activation's gradient list
  int64_t tr_mean[3] = {-1, -1, -1}, tr_invstd[3] = {-1, -1, -1}, tr_m1m2[3] = {-1, -1, -1};
  int64_t gap_off[3] = {-1, -1, -1};
  int64_t gapin_off[3] = {-1, -1, -1};   // depthwise unit, training: per-tile plane sums of an input that is never stored (virt_cons)
  int64_t bnred_off[3] = {-1, -1, -1};   // ... and the BatchNorm-backward sums of its producer, taken by this unit's backward kernel
  int64_t pstats_off[3] = {-1, -1, -1};  // 1x1 unit on pw4_kernel, bf16 training: statistics partials its epilogue writes, [C][pstats_n][2]
  int pstats_n[3] = {0, 0, 0};           // ... slabs per channel (image x item tile of the launch that stores the branch)
  int64_t dwwg_off[3] = {-1, -1, -1};    // depthwise unit, training: its own weight-gradient partials [C][NSLAB][9] (finalised by ONE
                                         // launch for all units at the end of csn_backward instead of one per unit and branch)
  int in_slot[3] = {-1, -1, -1};
  // GOCT 1x1 with two or three input branches: launches of pw4_kernel (k_pw4.hip); pw4 = 0: the unit does not qualify
  struct Pw4Launch {
    int hi_out = -1, lo_out = -1;    // output branch written from the high / low rows (-1: none)
    int use_x2 = 0;                  // third input branch (single-output forms)
    int bh = 0, bl = 1, bq = -1;     // input branches bound to xh / xl (x2 = bl + 1) / xq
    int nth = 0, ntl = 0, ng = 0, gimg = 0;
    Pw4Group grp[PW4_MAX_GROUPS] = {};
    int64_t wimg = -1, ep[2] = {-1, -1};
  };
  std::vector<Pw4Launch> pw4l;
  int pw4 = 0;
  // GOCT 1x1 at the head of an ILBlock whose planes fit the LDS: this unit + the fused depthwise pair behind it as ONE launch of
  // ilb_kernel (k_ilb.hip); on = 0: not eligible
  struct Ilb {
    int on = 0, nth = 0, ntl = 0, ng = 0, gimg = 0, Rh = 4, Rl = 4;
    int k3 = 0;   // the 3x3 stride-2 entry block of a stage (one input branch; its 2x2 averages / maxima come from the pair in front)
    int64_t wimg = -1, ep[2] = {-1, -1};
    int64_t dwrec[2] = {-1, -1};   // per channel {w9[9] x100, scale, shift, alpha} of conv3x3_1 and of conv3x3_2 (24 floats)
  } ilb;
  int pw4_old_mask = 0;              // output branches that stay on goct_pw_kernel (CSFHead.fuse's lowest branch)
  // GOCT 1x1 with three input branches: the HIGH output on hz_kernel (k_head.hip; eval mode) instead of pw4l[0]; on = 0: not eligible
  struct Hz { int on = 0, nth = 0, ng = 0, gimg = 0, RB = 2, hb = 2, nw = 4; int64_t wimg = -1, ep = -1; } hz;
  // MS
  int64_t ms_w[5] = {-1, -1, -1, -1, -1};
  Epi ms_epi;
  const char* kname = "";
  int64_t alg_bytes = 0;
};

struct UnitBwd;

}  // namespace

struct csn_plan {
  int B = 0, H = 0, W = 0, S = 0;  // S = images per slice
  std::vector<Act> acts;
  std::vector<UnitPlan> units;
  int64_t ws_bytes = 0;
  int64_t packed_floats = 0;
  float* packed = nullptr;  // device
  std::vector<CsnPrepJob> jobs;
  CsnPrepJob* jobs_dev = nullptr;
  bool params_ready = false;
  bool bn_tables_train = false;   // csn_forward_train overwrote the folded BN tables: refresh before eval
  bool fuse_dw = true;
  bool fuse_cls = true;   // CSN_OPT_FUSE_CLS
  bool tiled3 = true;     // CSN_OPT_TILED3
  mutable int ws_regions_reported = 0;   // workspace regions the last csn_plan_workspace_bytes() call sized (0: never queried)
  int input_grad = 0;     // CSN_OPT_INPUT_GRAD: csn_backward also leaves the gradient w.r.t. the image batch (set before csn_plan_enable_training)
  int slice_lanes = 0;    // CSN_OPT_SLICE_LANES: batch slices (sub_batch < B) run concurrently on the plan's stream lanes, each
                          // in its own workspace region
  bool c3q = true;        // CSN_OPT_C3Q: eval-mode 3x3 passes on c3q_kernel (k_c3q.hip)
  int c3q_cap = 4;        // its row tiles per M group
  bool dw_fast = true;    // the fused depthwise pair on dw3x3x2_fast_kernel where every branch qualifies (dw_pair_fast)
  int dwb_fast = 1;       // CSN_DWB_FAST=0: the fully fused depthwise backward on the round-4 loop instead of dw3x3_bwd_x_kernel (A/B)
  bool dw_xl = true;      // CSN_DW_XL=0: train-mode depthwise launches in the round-4 geometry, halo columns loaded (see dw_lanes_x)
  bool ilb = true;        // CSN_OPT_FUSE_ILB / CSN_ILB=0: whole ILBlocks of the small maps on ilb_kernel (k_ilb.hip, round 5)
  int ilb_nt = 0;         // its row tiles per group and branch: 0 = chosen per block (plan_ilb), CSN_ILB_NT=1|2 forces (experiments)
  int ilb_maxpix = 784;   // ... only where the low plane has at most this many pixels (CSN_ILB_MAXPIX).  Round 6: 256 -> 784 (the 56^2 / 28^2
                          // blocks too): slower launch by launch (below), but with the batch in two slices on two stream lanes -- the
                          // default for 32 images and more -- these latency chains run next to the other slice's bandwidth-bound
                          // launches: 30,899 -> 31,008 img/s (three A/B pairs on one box), batch 1: 0.614 -> 0.599 ms
  bool pw4 = true;        // CSN_OPT_PW4: two-branch 1x1 units on pw4_kernel (k_pw4.hip)
  bool hz = true;         // CSN_HZ=0: the high output of the three-branch 1x1 units stays on pw4_kernel (A/B; k_head.hip)
  // its geometry, [0] for a unit with several outputs (CSFHead.fuse), [1] for a single-output unit (fuse1x1); the environment
  // variables take "a" or "a/b" (experiments): row tiles per M group (0: chosen by plan_hz; CSN_HZ_NT), rows of the lowest input per
  // band (CSN_HZ_RB), x_0 channels per load batch (CSN_HZ_HB), waves per block (CSN_HZ_NW)
  // (defaults from the sweep of profiles/r6_notes.md: fuse 4-row bands on 4 waves; fuse1x1 groups of 4 tiles, 4-row bands, 8 waves)
  int hz_nt[2] = {0, 4}, hz_rb[2] = {4, 4}, hz_hb[2] = {2, 2}, hz_nw[2] = {4, 8};
#ifndef CSN_PW4_GRID_CAP
#define CSN_PW4_GRID_CAP 2048
#endif
  int pw4_grid = CSN_PW4_GRID_CAP;    // its block cap (persistent blocks walk the items beyond it)
  int pw4_twl = 6;        // log2 of its widest tile in low pixels: whole rows of up to 64 (CSN_PW4_TWL; 4 = 16 x 4 tiles: 1 % slower)
  bool pw4_flat = true;       // CSN_PW4_FLAT=0: row-segment tiles everywhere (round 3; A/B)
  bool pwq16 = true;          // CSN_PWQ16=0: 1x1 input-gradient launches of the bf16 step on pwq_kernel<bf16> (fp32 matrix instruction)
  bool pool_route = true;     // CSN_POOL_ROUTE=0: the 2x2 max-pool adjoint of the 1x1 units as its own read-modify-write pass over dx
                              // (maxpool2_bwd_add_pair_kernel) instead of pwq_kernel's epilogue (round 6; bit-identical in fp32, tests flip it)
  bool pw4_stats = true;      // CSN_PW4_STATS=0: bn_stats_kernel's pass over the bf16 train forward's 1x1 outputs instead of pw4_kernel's
                              // epilogue sums (round 6; tests flip it)
  bool ms_dx = true;          // CSN_MS_DX=0: MSBlock input gradients on the generic tap kernel (two launches) instead of ms_dx_kernel
  int c3q16 = 1;              // CSN_C3Q16=0: the bf16 step's 3x3 INPUT-GRADIENT launches on c3q_kernel<bf16> instead of c3q16_kernel (the
                              // forward launches always keep fp32 weights: bf16 weights there put z of the stride-2 units past the unit-local bound)
  bool pw4_no_q = false;      // CSN_PW4_NOQ: CSFHead.fuse's lowest output branch stays on goct_pw_kernel (experiments)
  bool bn_bwd_fuse = true;    // depthwise backward forms dz on load, the BatchNorm backward's apply pass is skipped (CSN_BN_BWD_FUSE=0: off)
  bool bn_fwd_fuse = true;    // activations consumed only by a depthwise unit are formed on load (virt_cons)
  bool debug_dz = false;      // CSN_DEBUG_DZ (tests): the skipped passes (y of virt_cons activations, dz below) still run for the probes
  // CSN_DEBUG_DZ: ... and the apply pass still runs AFTER that kernel, so that the probes see dz (tests)
  bool no_mp_fuse = false;    // CSN_NO_MP_FUSE: the max-pooled copies of c3q_kernel always come from pool2_kernel (experiments)
  bool c3q_hl = true;     // CSN_C3Q_HL=0: c3q_kernel's float launches on 64-quad tiles with loaded edge columns (round 3) instead of halo lanes
  int c3q_twl = 6;        // ... of c3q_kernel's tile in output quads (CSN_C3Q_TWL)
  bool pw4_nosplit = false;   // (no extra M groups on small maps: off)
  bool overlap_bwd = false;   // CSN_OPT_OVERLAP value 2: also the weight-gradient side lane of csn_backward
  bool last_bwd_lanes = false;   // ... and whether the last csn_backward really ran with it (csn_plan_train_act_info reports from this)
  int device = 0;             // ordinal of the device the plan's streams / events / packed buffer live on (csn_plan_destroy)
  bool overlap = true;    // CSN_OPT_OVERLAP: independent launches of a unit (and the MSBlocks) on parallel stream lanes
  hipStream_t lane[2] = {nullptr, nullptr};      // auxiliary lanes (lane 0 = the caller's stream)
  hipEvent_t lane_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  Epi ident;   // identity epilogue (scale 1, shift 0, alpha 1): train mode runs the conv kernels raw
  bool use_graph = true;
  // hipGraph of one whole csn_forward (all batch slices), captured on a plan-owned stream on the second
  // call with the same (x, y, workspace) and replayed on the caller's stream afterwards
  hipStream_t cap_stream = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  const void* g_x = nullptr; const void* g_y = nullptr; const void* g_ws = nullptr;
  int eager_calls = 0;
  // same for the train-mode forward and the backward pass (keyed by every pointer / scalar baked into the launches)
  struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    std::vector<uint64_t> key;
    int eager_calls = 0;
  };
  GraphSlot g_train, g_bwd;
  std::vector<hipEvent_t> ev;
  // per-launch profiling (csn_forward_profile): one event after every kernel launch, tagged with its name
  struct KStat { const char* name; double ms; int launches; };
  std::vector<const char*> tags;     // tag of the launch that ended at event i+1 (filled while profiling)
  std::vector<KStat> kstats;         // aggregated over the last csn_forward_profile call
  bool profiling = false;
  size_t ev_used = 0;
  float bracket_ms = 0.f;            // what an event pair adds to one (empty) launch: subtracted from every profiled interval
  // ---- training (csn_plan_enable_training) ----
  int64_t pen_off = -1;                         // penalty job terms (workspace bytes), pen_slots doubles
  int pen_slots = 0;
  bool train = false;
  bool act16 = false;                           // CSN_OPT_TRAIN_BF16: train-mode activations / gradients are bfloat16
  // ... and, when the option was set BEFORE csn_plan_enable_training, every activation-typed region of the workspace is laid
  // out for 2-byte elements (round 3: 61 -> 31 GiB at batch 256); such a plan runs the bf16 train step only
  bool act_half = false;
  struct WsAlloc { int64_t off, bytes; bool act; };
  std::vector<WsAlloc> ws_allocs;               // every region of the workspace, in allocation order
  int64_t x16_off = -1;                         // bf16 copy of the input batch (workspace bytes)
  std::vector<int64_t> tz_off;                  // per act: raw conv output z, later dz (workspace bytes)
  std::vector<std::array<int64_t, 2>> tg_off;   // per act: gradient buffer per consumer
  std::vector<int> n_cons;
  // Training, activations that are never stored (round 3): the output of a unit whose ONLY consumer is a depthwise unit on the
  // one-pass kernels -- that consumer forms y = PReLU(BN(z)) on load from the producer's raw output z (forward and backward) and
  // leaves the plane sums the penalty needs, so the producer's bn_apply_gap pass (read z, write y) does not run.
  std::vector<int> virt_cons;            // per act: index of that consumer unit, -1 = the activation is stored
  std::vector<int> act_prod_unit, act_prod_branch;   // per act: producing unit / its output branch
  std::vector<int> orphan_acts;          // outputs without consumer: zero gradient (training)
  int64_t scratch_off = 0, scratch_bytes = 0;   // per-unit backward temporaries (shared by all units)
  int64_t red_off = 0;                          // fp64 partials of the BN / depthwise reductions: [red_maxc][NSLAB][9] doubles
  int red_maxc = 0;
  int64_t wg_off = 0;                           // partial dW slices (k_wgrad.hip): CSN_WG_REGIONS regions of wg_region_floats
  int64_t wg_region_floats = 0;
  int64_t red2_off = 0, wg2_off = 0;            // ... of the weight-gradient side lane (csn_backward)
  std::vector<UnitBwd> bwd;
  ~csn_plan();
};

namespace {

struct Builder {
  csn_plan& P;
  explicit Builder(csn_plan& p) : P(p) {}

  int64_t alloc_packed(int64_t n) {
    const int64_t off = P.packed_floats;
    P.packed_floats += align_up(n > 0 ? n : 1, 4);
    return off;
  }
  // act: the region holds activation-typed elements (float, or bfloat16 in the bf16 train mode -- see relayout_half)
  int64_t alloc_ws(int64_t bytes, bool act = false) {
    const int64_t off = P.ws_bytes;
    P.ws_bytes += align_up(bytes > 0 ? bytes : 1, 256);   // (never two regions at one offset: relayout_half maps by offset)
    P.ws_allocs.push_back({off, bytes, act});
    return off;
  }
  int64_t alloc_act(int C, int lvl) { return alloc_ws(act_bytes(C, lvl), true); }
  void job(int kind, int n, int64_t dst, int64_t s0 = -1, int64_t s1 = -1, int64_t s2 = -1, int64_t s3 = -1,
           float p0f = 1.f, int p0 = 0, int p1 = 0, int p2 = 0, int p3 = 0) {
    CsnPrepJob j;
    j.kind = kind; j.n = n; j.p0 = p0; j.p1 = p1; j.p2 = p2; j.p3 = p3; j.p0f = p0f;
    j.src0 = s0; j.src1 = s1; j.src2 = s2; j.src3 = s3; j.dst = dst;
    P.jobs.push_back(j);
  }
  Epi bn_epi(const csn_bn_off& bn, int C) {
    Epi e;
    e.scale = alloc_packed(C); e.shift = alloc_packed(C); e.alpha = alloc_packed(C);
    job(CSN_PREP_BN_SCALE, C, e.scale, bn.weight, bn.running_var);
    job(CSN_PREP_BN_SHIFT, C, e.shift, bn.weight, bn.running_var, bn.bias, bn.running_mean);
    job(CSN_PREP_COPY, C, e.alpha, bn.prelu);
    return e;
  }
  int64_t act_bytes(int C, int lvl) const {
    return (int64_t)P.S * C * (P.H >> lvl) * (P.W >> lvl) * (int64_t)(P.act_half ? 2 : 4);
  }
};

bool bn_ok(const csn_bn_off& b) {
  return b.weight >= 0 && b.bias >= 0 && b.running_mean >= 0 && b.running_var >= 0 && b.prelu >= 0;
}

// M groups a pwq launch may have (round 5: 4; the un-pruned net's 80 .. 160-row input gradients need 5 .. 7)
static int pwq_max_groups() { return PWQ_MAX_GROUPS; }

// Lay out the weight image of a launch (rows padded to 16, pitch K4 + 2 floats) and emit the packing jobs.
int finish_launch(Builder& bl, PwLaunchPlan& L) {
  int img = 0;
  for (PwPassPlan& ps : L.passes) {
    ps.K4 = round4(ps.K);
    ps.w_stride = ps.K4 + 2;   // == 2 (mod 4): the 64 A-operand addresses of an MFMA hit 32 distinct banks twice
    ps.w_off = img;
    img += ((ps.nrows + 15) & ~15) * ps.w_stride;
    img = (img + 3) & ~3;
  }
  if ((int)L.passes.size() > PW_MAX_PASS) FAIL(CSN_E_UNSUPPORTED, "too many passes in one launch");
  if (((int64_t)img + 4 * PW_KC * PW_XP) * 4 > 160 * 1024) FAIL(CSN_E_UNSUPPORTED, "weight image exceeds the LDS of a CU");
  L.wimg_floats = img;
  L.wimg = bl.alloc_packed(img);
  for (const PwPassPlan& ps : L.passes)
    for (const WBlock& w : ps.wb) {
      if (w.eye)
        bl.job(CSN_PREP_EYE, w.eye_rows > 0 ? w.eye_rows : w.ncol, L.wimg + ps.w_off, -1, -1, -1, -1, 1.f, 0, 0,
               ps.w_stride, w.col + w.eye_col0);
      else if (w.tk > 0)
        bl.job(CSN_PREP_ROWS_T, ps.nrows, L.wimg + ps.w_off, w.src, -1, -1, -1, w.scale, w.ld, w.ncol, ps.w_stride,
               w.col | (w.tk << 24));
      else
        bl.job(CSN_PREP_ROWS, ps.nrows, L.wimg + ps.w_off, w.src, -1, -1, -1, w.scale, w.ld, w.ncol, ps.w_stride, w.col);
    }
  // ---- pwq_kernel: one pass of plain own-resolution slices on whole planes of a multiple of four elements ----
  if (L.passes.size() == 1 && L.passes[0].r == 0 && L.passes[0].nsrc > 0) {
    const PwPassPlan& ps = L.passes[0];
    const int64_t hw = (int64_t)(bl.P.H >> L.lvl) * (bl.P.W >> L.lvl);
    bool q = (hw & 3) == 0 && (ps.out_kind == OUT_DX || ps.out_kind == OUT_TMP);
    for (int s = 0; s < ps.nsrc; ++s) q = q && ps.src_mode[s] == PW_OWN && !ps.wb[s].eye && ps.wb[s].tk <= 1;
    const int nt_tot = (ps.nrows + 3) / 4;
    const int ng = (nt_tot + csn_pwq_max_tiles() - 1) / csn_pwq_max_tiles();
    if (q && ng >= 1 && ng <= pwq_max_groups()) {
      const int nt = (nt_tot + ng - 1) / ng;
      const int Pp = PW4_PITCH((nt + 3) & ~3);
      const int64_t gimg = (int64_t)ps.K * 4 * Pp;
      if (gimg * ng * 4 <= 150 * 1024) {
        L.pwq = 1; L.pwq_nt = nt; L.pwq_ng = ng; L.pwq_gimg = (int)gimg;
        L.pwq_wimg = bl.alloc_packed(gimg * ng);
        for (int g = 0; g < ng; ++g) {
          L.pwq_r0[g] = 4 * g * nt;
          L.pwq_gnt[g] = std::max(0, std::min(nt, nt_tot - g * nt));
          const int nr = std::min(4 * L.pwq_gnt[g], ps.nrows - L.pwq_r0[g]);
          int kb = 0;
          for (int s = 0; s < ps.nsrc && nr > 0; ++s) {
            const WBlock& w = ps.wb[s];
            if (w.tk > 0)
              bl.job(CSN_PREP_PW4_T, nr, L.pwq_wimg + g * gimg, w.src + L.pwq_r0[g], -1, -1, -1, w.scale, w.ld, ps.src_C[s], Pp,
                     0 | (kb << 8));
            else
              bl.job(CSN_PREP_PW4, nr, L.pwq_wimg + g * gimg, w.src + (int64_t)L.pwq_r0[g] * w.ld, -1, -1, -1, w.scale, w.ld,
                     ps.src_C[s], Pp, 0 | (kb << 8));
            kb += ps.src_C[s];
          }
        }
      }
    }
  }
  // ---- tap-major image for goct_c3_kernel: one pass at the launch resolution, 3x3 tap slices (dilation 1) first, at
  // most one bilinear slice (identity block) last
  if (L.passes.size() == 1 && L.passes[0].r == 0) {
    const PwPassPlan& ps = L.passes[0];
    int ntap = 0;
    bool ok = ps.nsrc > 0, tail = false;
    for (int s = 0; s < ps.nsrc && ok; ++s) {
      const int m = ps.src_mode[s];
      if (m == PW_TAPS || m == PW_POOL2_TAPS) {
        if (tail || ps.src_dil[s] != 1 || ps.wb[s].eye) ok = false;
        ++ntap;
      } else if (m == PW_UP2 && !tail && ps.wb[s].eye) {
        tail = true;
      } else {
        ok = false;
      }
    }
    if (ok && ntap > 0 && ntap <= 3) {
      int cols = 0;
      for (int s = 0; s < ntap; ++s) cols += ((ps.src_C[s] + 15) / 16) * 144;
      L.w3_stride = cols + 2;                                  // == 2 (mod 4): conflict-free A-operand reads
      L.w3_floats = ((ps.nrows + 15) & ~15) * L.w3_stride;
      L.wimg3 = bl.alloc_packed(L.w3_floats);
      int col = 0;
      for (int s = 0; s < ntap; ++s) {
        const WBlock& w = ps.wb[s];
        const int C = ps.src_C[s];
        if (w.tk > 0)      // backward data: element (row ci, channel co, tap t) = W[co][ci][8 - t]
          bl.job(CSN_PREP_C3T_T, ps.nrows, L.wimg3, w.src, -1, -1, -1, w.scale, w.ld, C, L.w3_stride, col);
        else
          bl.job(CSN_PREP_C3T, ps.nrows, L.wimg3, w.src, -1, -1, -1, w.scale, w.ld, C, L.w3_stride, col);
        col += ((C + 15) / 16) * 144;
      }
      if (tail) L.z_c0 = ps.wb[ntap].eye_col0;
      // ---- c3q_kernel: forward passes on whole tensors at an even resolution; round 4: also the input-gradient passes (a 3x3
      // convolution of dz with transposed, tap-flipped weight blocks, CSN_PREP_C3Q_T: plain own-resolution tap slices only) ----
      const bool grad_pass = ps.out_kind == OUT_DX || ps.out_kind == OUT_TMP;
      const bool c3q_bwd_off = std::getenv("CSN_C3Q_BWD") && std::getenv("CSN_C3Q_BWD")[0] == '0';
      bool q = (((bl.P.H >> L.lvl) | (bl.P.W >> L.lvl)) & 1) == 0 && !(grad_pass && (c3q_bwd_off || tail));
      for (int s = 0; s < ntap; ++s) {
        q = q && ps.src_c0[s] == 0 && (ps.src_ctot[s] == 0 || ps.src_ctot[s] == ps.src_C[s]);
        if (grad_pass) q = q && ps.wb[s].tk == 9 && ps.src_mode[s] == PW_TAPS && (ps.src_kind[s] == SRC_DZ || ps.src_kind[s] == SRC_ADJ);
        else q = q && ps.wb[s].tk == 0 && ps.src_kind[s] == SRC_IN;
      }
      if (q) {
        int cap = std::max(1, std::min(bl.P.c3q_cap, csn_c3q_max_tiles()));
        const int nt_tot = (ps.nrows + 3) / 4;
        {   // small maps: more, shorter items (measured per unit with CSN_C3Q_NT: 56^2 quads want 3 tiles, 28^2 and below 2)
          const int Hq = bl.P.H >> (L.lvl + 1), Wq = bl.P.W >> (L.lvl + 1);
          const int64_t tiles = (int64_t)bl.P.S * ((Hq * Wq + 63) / 64);
          while (cap > 2 && !bl.P.pw4_nosplit && tiles * ((nt_tot + cap - 1) / cap) < 1536) --cap;
        }
        int ng = (nt_tot + cap - 1) / cap;
        if (ng > PW4_MAX_GROUPS) ng = PW4_MAX_GROUPS;
        const int nt = (nt_tot + ng - 1) / ng;
        int K = 0;
        for (int s = 0; s < ntap; ++s) K += 9 * ps.src_C[s];
        const int Pp = PW4_PITCH((nt + 3) & ~3);
        const int64_t gimg = (int64_t)K * 4 * Pp;
        if (nt <= csn_c3q_max_tiles() && gimg * ng * 4 <= 150 * 1024) {
          L.c3q = 1; L.c3q_nt = nt; L.c3q_ng = ng; L.c3q_gimg = (int)gimg; L.c3q_ntap = ntap; L.c3q_z = tail ? 1 : 0;
          L.c3q_wimg = bl.alloc_packed(gimg * ng);
          for (int g = 0; g < ng; ++g) {
            L.c3q_r0[g] = 4 * g * nt;
            L.c3q_gnt[g] = std::max(0, std::min(nt, nt_tot - g * nt));
            const int nr = std::min(4 * L.c3q_gnt[g], ps.nrows - L.c3q_r0[g]);
            int kb = 0;
            for (int s = 0; s < ntap && nr > 0; ++s) {
              const WBlock& w = ps.wb[s];
              if (w.tk > 0)   // backward data: row r = input channel, gathered channel = output channel, taps flipped
                bl.job(CSN_PREP_C3Q_T, nr, L.c3q_wimg + g * gimg, w.src + (int64_t)L.c3q_r0[g] * 9, -1, -1, -1, w.scale, w.ld,
                       ps.src_C[s], Pp, 0 | (kb << 8));
              else
                bl.job(CSN_PREP_C3Q, nr, L.c3q_wimg + g * gimg, w.src + (int64_t)L.c3q_r0[g] * w.ld, -1, -1, -1, w.scale, w.ld,
                       ps.src_C[s], Pp, 0 | (kb << 8));
              kb += 9 * ps.src_C[s];
            }
          }
          if (ps.out_kind == OUT_ACT && bn_ok(ps.bn)) {
            L.c3q_ep = bl.alloc_packed((int64_t)(4 * ng * nt + 4) * 4);
            bl.job(CSN_PREP_BN_SCALE, ps.nrows, L.c3q_ep, ps.bn.weight, ps.bn.running_var, -1, -1, 1.f, 0, 0, 4, 0);
            bl.job(CSN_PREP_BN_SHIFT, ps.nrows, L.c3q_ep, ps.bn.weight, ps.bn.running_var, ps.bn.bias, ps.bn.running_mean, 1.f, 0, 0, 4, 1);
            bl.job(CSN_PREP_COPY, ps.nrows, L.c3q_ep, ps.bn.prelu, -1, -1, -1, 1.f, 0, 0, 4, 2);
          } else if (ps.out_kind == OUT_ACT) {
            L.c3q = 0;
          }
        }
      }
    }
  }
  return CSN_OK;
}

// A single-pass launch whose weight image would not fit the LDS comfortably is cut into row chunks (each chunk
// re-gathers the inputs; only the wide un-pruned networks get here).
void push_row_chunks(std::vector<PwLaunchPlan>& dst, const PwLaunchPlan& L) {
  const PwPassPlan& ps = L.passes[0];
  const int64_t stride = round4(ps.K) + 2;
  const int64_t limit = 96 * 1024;
  if (L.passes.size() != 1 || (int64_t)((ps.nrows + 15) & ~15) * stride * 4 <= limit) {
    dst.push_back(L);
    return;
  }
  int rows = (int)((limit / (stride * 4)) & ~15);
  if (rows < 16) rows = 16;
  for (int r0 = 0; r0 < ps.nrows; r0 += rows) {
    PwLaunchPlan one = L;
    PwPassPlan& q = one.passes[0];
    q.nrows = std::min(rows, ps.nrows - r0);
    q.out_c0 = ps.out_c0 + r0;
    q.epi.scale += r0; q.epi.shift += r0; q.epi.alpha += r0;
    if (bn_ok(q.bn)) { q.bn.weight += r0; q.bn.bias += r0; q.bn.running_mean += r0; q.bn.running_var += r0; q.bn.prelu += r0; }
    for (WBlock& w : q.wb) {
      if (w.eye) { w.eye_rows = q.nrows; w.eye_col0 = r0; }
      else if (w.tk > 0) w.src += (int64_t)r0 * w.tk;
      else w.src += (int64_t)r0 * w.ld;
    }
    dst.push_back(one);
  }
}

// Split a launch whose weight image would cap the occupancy through LDS into one launch per pass, each
// re-based on its own resolution (the shared inputs are then re-read through L2 / Infinity Cache).
void add_launch(std::vector<PwLaunchPlan>& dst, PwLaunchPlan L) {
  int64_t img = 0;
  for (const PwPassPlan& ps : L.passes) img += (int64_t)((ps.nrows + 15) & ~15) * (round4(ps.K) + 2);
  bool taps = false;   // 3x3 passes run one per launch (k_goct_c3.hip)
  for (const PwPassPlan& ps : L.passes)
    for (int s = 0; s < ps.nsrc; ++s) taps = taps || pw_mode_taps(ps.src_mode[s]);
  if (L.passes.size() > 1 && (img * 4 > 24 * 1024 || taps)) {
    for (const PwPassPlan& ps : L.passes) {
      PwLaunchPlan one;
      one.lvl = L.lvl + ps.r;
      one.passes.push_back(ps);
      one.passes[0].r = 0;
      push_row_chunks(dst, one);
    }
  } else {
    // a launch left with ONE pass of a lower branch (the finer output branches were pruned to zero channels): re-base it
    // on its own resolution like the split launches above (goct_c3_kernel walks the launch resolution)
    if (L.passes.size() == 1 && L.passes[0].r > 0) {
      L.lvl += L.passes[0].r;
      L.passes[0].r = 0;
    }
    push_row_chunks(dst, L);
  }
}

int plan_pw4(Builder& bl, UnitPlan& u, const int* ci_off, const int* co_off, int cin_tot);

int plan_goct(Builder& bl, UnitPlan& u) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  if (d.n_in < 1 || d.n_in > 3 || d.n_out < 1 || d.n_out > 3) FAIL(CSN_E_INVALID, "branch count");
  if (!(d.ksize == 1 || d.ksize == 3) || !(d.stride == 1 || d.stride == 2)) FAIL(CSN_E_INVALID, "ksize/stride");
  // single branch in, single branch out: the reference builds a Conv2dX100 ("std_conv", csnet.py:751-754): weights x100
  // (conv2d.py:104) and a REAL stride (no 2x2 avg-pool in front, csnet.py:779-786)
  u.std_conv = d.n_in == 1 && d.n_out == 1;
  if (u.std_conv && d.stride == 2 && d.ksize != 3) FAIL(CSN_E_UNSUPPORTED, "std_conv 1x1 with stride 2");
  int cin_tot = 0, cout_tot = 0, ci_off[4] = {0}, co_off[4] = {0};
  for (int i = 0; i < d.n_in; ++i) { ci_off[i] = cin_tot; cin_tot += d.cin[i]; }
  for (int j = 0; j < d.n_out; ++j) { co_off[j] = cout_tot; cout_tot += d.cout[j]; }
  int base = -1;
  for (int j = 0; j < d.n_out; ++j)
    if (d.cout[j] > 0) {
      if (d.out_act[j] < 0 || d.out_act[j] >= (int)P.acts.size()) FAIL(CSN_E_INVALID, "out_act");
      const Act& a = P.acts[d.out_act[j]];
      if (a.channels != d.cout[j]) FAIL(CSN_E_INVALID, "out channels");
      const int b0 = a.lvl - j;
      if (base >= 0 && b0 != base) FAIL(CSN_E_INVALID, "output resolutions are not octave spaced");
      base = b0;
      if (!bn_ok(d.bn[j])) FAIL(CSN_E_INVALID, "missing BN/PReLU offsets");
    }
  if (base < 0) FAIL(CSN_E_INVALID, "no outputs");
  u.base_lvl = base;
  const int ds = d.stride == 2 ? 1 : 0;
  const bool std_s2 = u.std_conv && ds;
  for (int i = 0; i < d.n_in; ++i)
    if (d.cin[i] > 0) {
      if (d.in_act[i] < 0 || d.in_act[i] >= (int)P.acts.size()) FAIL(CSN_E_INVALID, "in_act");
      const Act& a = P.acts[d.in_act[i]];
      if (a.channels != d.cin[i] || a.lvl + ds != base + i) FAIL(CSN_E_INVALID, "input resolution/channels");
      if (ds && !std_s2) u.pooled_off[i] = bl.alloc_act(d.cin[i], base + i);
    }
  const int nb = d.n_in > d.n_out ? d.n_in : d.n_out;
  if (((P.H >> (base + nb - 1)) << (base + nb - 1)) != P.H || ((P.W >> (base + nb - 1)) << (base + nb - 1)) != P.W)
    FAIL(CSN_E_INVALID, "H/W not divisible for the lowest branch");
  Epi epi[3];
  for (int j = 0; j < d.n_out; ++j)
    if (d.cout[j] > 0) { epi[j] = bl.bn_epi(d.bn[j], d.cout[j]); u.out_epi[j] = epi[j]; }
  const int kk = d.ksize * d.ksize;
  const int ld = cin_tot * kk;
  u.kname = "goct_pw_kernel";

  // 3x3 low->high term: conv at the low resolution into a scratch (its own launch: the high-resolution pass
  // samples it bilinearly across tile borders), then added through an identity block of the contraction
  const bool z_path = d.ksize == 3 && d.n_in >= 2 && d.cout[0] > 0 && d.cin[1] > 0;
  if (d.ksize == 3 && (d.n_in > 2 || d.n_out > 2)) FAIL(CSN_E_UNSUPPORTED, "3x3 gOctConv with three branches");
  if (z_path) {
    u.z_C = d.cout[0];
    u.z_off = bl.alloc_act(u.z_C, base + 1);
    PwLaunchPlan L;
    L.lvl = base + 1;
    PwPassPlan ps;
    ps.r = 0; ps.nsrc = 1; ps.src_kind[0] = SRC_IN; ps.src_branch[0] = 1; ps.src_C[0] = d.cin[1]; ps.src_mode[0] = PW_TAPS;
    ps.K = d.cin[1] * 9; ps.nrows = u.z_C; ps.out_kind = OUT_Z; ps.out_ctot = u.z_C;
    ps.epi.scale = bl.alloc_packed(u.z_C); ps.epi.shift = bl.alloc_packed(u.z_C); ps.epi.alpha = bl.alloc_packed(u.z_C);
    bl.job(CSN_PREP_FILL, u.z_C, ps.epi.scale, -1, -1, -1, -1, 1.f);
    bl.job(CSN_PREP_FILL, u.z_C, ps.epi.shift, -1, -1, -1, -1, 0.f);
    bl.job(CSN_PREP_FILL, u.z_C, ps.epi.alpha, -1, -1, -1, -1, 1.f);
    WBlock w; w.src = d.w_off[0] + ((int64_t)co_off[0] * cin_tot + ci_off[1]) * 9; w.ld = ld; w.ncol = d.cin[1] * 9; w.col = 0;
    ps.wb.push_back(w);
    L.passes.push_back(ps);
    add_launch(u.pwl, L);
  }
  PwLaunchPlan L;
  L.lvl = base;
  for (int j = d.n_out - 1; j >= 0; --j) {
    if (d.cout[j] == 0) continue;
    PwPassPlan ps;
    ps.r = j; ps.nrows = d.cout[j]; ps.out_kind = OUT_ACT; ps.out_branch = j; ps.out_ctot = d.cout[j]; ps.epi = epi[j];
    ps.bn = d.bn[j];
    auto add = [&](int kind, int branch, int C, int mode, const WBlock& wproto) {
      if (ps.nsrc >= 3) return false;
      const int s = ps.nsrc++;
      ps.src_kind[s] = kind; ps.src_branch[s] = branch; ps.src_C[s] = C; ps.src_mode[s] = mode;
      WBlock w = wproto;
      w.col = ps.K;
      ps.wb.push_back(w);
      ps.K += pw_mode_taps(mode) ? 9 * C : C;
      return true;
    };
    auto wblk = [&](int i) {
      WBlock w;
      w.src = d.w_off[0] + ((int64_t)co_off[j] * cin_tot + ci_off[i]) * kk;
      w.ld = ld; w.ncol = d.cin[i] * kk;
      if (u.std_conv) w.scale = 100.f;
      return w;
    };
    bool ok = true;
    if (j < d.n_in && d.cin[j] > 0)
      ok = ok && add(SRC_IN, j, d.cin[j], d.ksize == 3 ? (std_s2 ? PW_TAPS_S2 : PW_TAPS) : PW_OWN, wblk(j));
    for (int i = 0; i < j && i < d.n_in; ++i)
      if (d.cin[i] > 0) {
        if (j - i > 2 || (d.ksize == 3 && j - i > 1)) FAIL(CSN_E_UNSUPPORTED, "max-pool factor");
        ok = ok && add(SRC_IN, i, d.cin[i], d.ksize == 3 ? PW_POOL2_TAPS : (j - i == 1 ? PW_POOL2 : PW_POOL4), wblk(i));
      }
    for (int i = j + 1; i < d.n_in; ++i)
      if (d.cin[i] > 0) {
        if (d.ksize == 1) {
          if (i - j > 2) FAIL(CSN_E_UNSUPPORTED, "bilinear factor > 4");
          ok = ok && add(SRC_IN, i, d.cin[i], i - j == 1 ? PW_UP2 : PW_UP4, wblk(i));
        } else {   // the scratch written by the z launch, upsampled, through an identity block
          WBlock w; w.eye = 1; w.ncol = u.z_C;
          ok = ok && add(SRC_Z, 0, u.z_C, PW_UP2, w);
        }
      }
    if (!ok || ps.nsrc == 0) FAIL(CSN_E_INVALID, "output branch without inputs / too many inputs");
    L.passes.push_back(ps);
  }
  add_launch(u.pwl, L);
  if (d.ksize == 3 && !u.std_conv)   // 2x2 max-pooled copies of the inputs that feed a high -> low 3x3 slice (c3q_kernel)
    for (int i = 0; i + 1 < d.n_out && i < d.n_in; ++i)
      if (d.cin[i] > 0 && d.cout[i + 1] > 0) u.mp_off[i] = bl.alloc_act(d.cin[i], base + i + 1);
  u.c3 = d.ksize == 3 && !std_s2;
  for (PwLaunchPlan& l : u.pwl) {
    const int st = finish_launch(bl, l);
    if (st != CSN_OK) return st;
    if (l.passes.size() != 1) u.c3 = 0;   // csn_c3_eligible
  }
  return plan_pw4(bl, u, ci_off, co_off, cin_tot);
}

// 1x1 unit with two or three input branches on pw4_kernel (k_pw4.hip): per launch M groups, weight image [group][K][4][P]
// (gathered channels: branch 0, 1, 2 in order), interleaved epilogue records.
// Input branches: bh -> xh (twice the resolution of bl), bl -> xl, bl + 1 -> x2 (use_x2), bq -> xq (four times the resolution
// of bl, low-only form); the defaults are the unit's branches 0, 1, 2.
int plan_pw4_launch(Builder& bl, UnitPlan& u, const int* ci_off, const int* co_off, int cin_tot, int hi_out, int lo_out, int use_x2,
                    int bh = 0, int bl_ = 1, int bq = -1) {
  const csn_unit_desc& d = u.d;
  const int OH = hi_out >= 0 ? d.cout[hi_out] : 0, OL = lo_out >= 0 ? d.cout[lo_out] : 0;
  const int nth_tot = (OH + 3) / 4, ntl_tot = (OL + 3) / 4;
  // M groups: as few as the accumulator budget allows (a group re-reads the inputs and repeats the interpolation
  // arithmetic) -- but a small map needs more items than that to occupy the 1024 SIMDs, and its items are short
  const int Hl_ = bl.P.H >> (u.base_lvl + bl_), Wl_ = bl.P.W >> (u.base_lvl + bl_);
  const int64_t tiles = (int64_t)bl.P.S * ((Hl_ * Wl_ + 63) / 64);
  int gmin = 1;
  if (!bl.P.pw4_nosplit)
    while (gmin < PW4_MAX_GROUPS && tiles * gmin < 1536 && gmin < std::max(nth_tot, ntl_tot)) ++gmin;
  const int budget = (lo_out < 0 && use_x2) ? 164 : (lo_out < 0 ? 116 : 100);
  int ng = 0, pn = 0, pl = 0;
  for (int g = gmin; g <= PW4_MAX_GROUPS; ++g) {
    int a = 0, b = 0;
    if (!csn_pw4_pick((nth_tot + g - 1) / g, (ntl_tot + g - 1) / g, &a, &b)) continue;
    if (16 * a + 4 * b > budget && g < PW4_MAX_GROUPS) continue;
    ng = g; pn = a; pl = b;
    break;
  }
  if (ng == 0) return -1;
  const int src[4] = {bh, bl_, use_x2 ? bl_ + 1 : -1, bq};   // gathered channel order of the weight image
  int K = 0;
  for (int i = 0; i < 4; ++i) K += src[i] >= 0 ? d.cin[src[i]] : 0;
  const int NT4 = (pn + pl + 3) & ~3, Pp = PW4_PITCH(NT4);
  UnitPlan::Pw4Launch L;
  L.hi_out = hi_out; L.lo_out = lo_out; L.use_x2 = use_x2; L.bh = bh; L.bl = bl_; L.bq = bq;
  L.nth = pn; L.ntl = pl; L.ng = ng; L.gimg = K * 4 * Pp;
  if ((int64_t)ng * L.gimg * 4 > 150 * 1024) return -1;
  L.wimg = bl.alloc_packed((int64_t)ng * L.gimg);
  const int gh = (nth_tot + ng - 1) / ng, gl = (ntl_tot + ng - 1) / ng;
  for (int g = 0; g < ng; ++g) {
    Pw4Group& G = L.grp[g];
    G.r0h = 4 * g * gh; G.nth = std::max(0, std::min(gh, nth_tot - g * gh));
    G.r0l = 4 * g * gl; G.ntl = std::max(0, std::min(gl, ntl_tot - g * gl));
    const int64_t img = L.wimg + (int64_t)g * L.gimg;
    const int nrh = std::min(4 * G.nth, OH - G.r0h), nrl = std::min(4 * G.ntl, OL - G.r0l);
    int k0 = 0;
    for (int s = 0; s < 4; ++s) {   // gathered channels: xh, xl, x2, xq
      const int i = src[s];
      if (i < 0 || d.cin[i] <= 0) continue;
      if (nrh > 0)
        bl.job(CSN_PREP_PW4, nrh, img, d.w_off[0] + (int64_t)(co_off[hi_out] + G.r0h) * cin_tot + ci_off[i], -1, -1, -1, 1.f,
               cin_tot, d.cin[i], Pp, 0 | (k0 << 8));
      if (nrl > 0)
        bl.job(CSN_PREP_PW4, nrl, img, d.w_off[0] + (int64_t)(co_off[lo_out] + G.r0l) * cin_tot + ci_off[i], -1, -1, -1, 1.f,
               cin_tot, d.cin[i], Pp, pn | (k0 << 8));
      k0 += d.cin[i];
    }
  }
  for (int q = 0; q < 2; ++q) {
    const int j = q == 0 ? hi_out : lo_out;
    if (j < 0) continue;
    const int C = d.cout[j];
    const int rows = 4 * ng * (q == 0 ? pn : pl) + 4;   // whole tiles of every group are readable
    L.ep[q] = bl.alloc_packed((int64_t)rows * 4);
    bl.job(CSN_PREP_BN_SCALE, C, L.ep[q], d.bn[j].weight, d.bn[j].running_var, -1, -1, 1.f, 0, 0, 4, 0);
    bl.job(CSN_PREP_BN_SHIFT, C, L.ep[q], d.bn[j].weight, d.bn[j].running_var, d.bn[j].bias, d.bn[j].running_mean, 1.f, 0, 0, 4, 1);
    bl.job(CSN_PREP_COPY, C, L.ep[q], d.bn[j].prelu, -1, -1, -1, 1.f, 0, 0, 4, 2);
  }
  u.pw4l.push_back(L);
  return 0;
}

// High output of a three-branch 1x1 unit on hz_kernel (k_head.hip): uniform M groups of `nth` row tiles, weight image
// [group][K][4][P] in pw4_kernel's form (gathered channels: branch 0, 1, 2), epilogue records padded to whole groups.
void plan_hz(Builder& bl, UnitPlan& u, const int* ci_off, const int* co_off, int cin_tot) {
  const csn_unit_desc& d = u.d;
  if (d.n_in != 3 || d.cin[0] <= 0 || d.cin[1] <= 0 || d.cin[2] <= 0 || d.cout[0] <= 0) return;
  const int H0 = bl.P.H >> u.base_lvl, W0 = bl.P.W >> u.base_lvl;
  if ((H0 % 4) != 0 || (W0 % 4) != 0) return;
  const int tiles = (d.cout[0] + 3) / 4;
  // row tiles per group: as few groups as the register budget allows (<= 5 tiles = 80 accumulators); a group re-reads x_0
  const int ki = d.n_out == 1 ? 1 : 0;
  int nth = std::min(bl.P.hz_nt[ki], tiles);
  if (nth <= 0 || (tiles + nth - 1) / nth > HZ_MAX_GROUPS)   // chosen here: <= 5 tiles per group, groups of equal size
    nth = tiles <= 5 ? tiles : (tiles + ((tiles + 4) / 5) - 1) / ((tiles + 4) / 5);
  const int ng = (tiles + nth - 1) / nth;
  if (ng > HZ_MAX_GROUPS || !csn_hz_supported(nth)) return;
  HzArgs a;
  a.CH = d.cin[0]; a.C1 = d.cin[1]; a.C2 = d.cin[2]; a.OH = d.cout[0]; a.H1 = H0 >> 1; a.W1 = W0 >> 1; a.B = 1;
  a.nth = nth; a.ngroups = ng; a.RB = bl.P.hz_rb[ki]; a.hb = bl.P.hz_hb[ki]; a.nw = bl.P.hz_nw[ki];
  const size_t lds = csn_hz_layout(a);
  if (lds == 0 || lds > 160 * 1024) return;
  const int NT4 = (nth + 3) & ~3, Pp = PW4_PITCH(NT4);
  UnitPlan::Hz& h = u.hz;
  h.nth = nth; h.ng = ng; h.gimg = a.gimg_floats; h.RB = a.RB; h.hb = a.hb; h.nw = a.nw;
  h.wimg = bl.alloc_packed((int64_t)ng * h.gimg);
  for (int g = 0; g < ng; ++g) {
    const int r0 = 4 * nth * g, nr = std::min(4 * nth, d.cout[0] - r0);
    int k0 = 0;
    for (int i = 0; i < 3; ++i) {
      bl.job(CSN_PREP_PW4, nr, h.wimg + (int64_t)g * h.gimg, d.w_off[0] + (int64_t)(co_off[0] + r0) * cin_tot + ci_off[i], -1, -1, -1, 1.f,
             cin_tot, d.cin[i], Pp, 0 | (k0 << 8));
      k0 += d.cin[i];
    }
  }
  const int rows = 4 * ng * nth + 4;
  h.ep = bl.alloc_packed((int64_t)rows * 4);
  bl.job(CSN_PREP_BN_SCALE, d.cout[0], h.ep, d.bn[0].weight, d.bn[0].running_var, -1, -1, 1.f, 0, 0, 4, 0);
  bl.job(CSN_PREP_BN_SHIFT, d.cout[0], h.ep, d.bn[0].weight, d.bn[0].running_var, d.bn[0].bias, d.bn[0].running_mean, 1.f, 0, 0, 4, 1);
  bl.job(CSN_PREP_COPY, d.cout[0], h.ep, d.bn[0].prelu, -1, -1, -1, 1.f, 0, 0, 4, 2);
  h.on = 1;
}

int plan_pw4(Builder& bl, UnitPlan& u, const int* ci_off, const int* co_off, int cin_tot) {
  const csn_unit_desc& d = u.d;
  if (d.ksize != 1 || d.stride != 1 || u.std_conv || d.n_in < 2 || d.cin[0] <= 0 || d.cin[1] <= 0 || d.cout[0] <= 0) return CSN_OK;
  bool ok = true;
  if (d.n_in == 2) {
    if (d.n_out > 2) return CSN_OK;
    // (a launch per output branch, so that each resampling is done once instead of once per M group, was measured and is
    // slower on every map: profiles/r3_notes.md)
    ok = plan_pw4_launch(bl, u, ci_off, co_off, cin_tot, 0, (d.n_out >= 2 && d.cout[1] > 0) ? 1 : -1, 0) == 0;
  } else {   // CSFHead.fuse / fuse1x1: one single-output launch per output branch 0 / 1, the lowest branch stays where it was
    const int x2 = d.cin[2] > 0 ? 1 : 0;
    ok = plan_pw4_launch(bl, u, ci_off, co_off, cin_tot, 0, -1, x2) == 0;
    if (ok && d.n_out >= 2 && d.cout[1] > 0) ok = plan_pw4_launch(bl, u, ci_off, co_off, cin_tot, -1, 1, x2) == 0;
    if (ok && d.n_out >= 3 && d.cout[2] > 0) {
      // the lowest output branch: W_22 x2 + W_21 maxpool2(x1) + W_20 maxpool4(x0) -- the low-only form one level down
      const int W0 = bl.P.W >> u.base_lvl, H0 = bl.P.H >> u.base_lvl;
      const bool fits = x2 && !bl.P.pw4_no_q && (W0 % 4) == 0 && (H0 % 4) == 0;
      if (!fits || plan_pw4_launch(bl, u, ci_off, co_off, cin_tot, -1, 2, 0, 1, 2, 0) != 0) {
        u.pw4_old_mask = 1 << 2;   // stays on goct_pw_kernel
        for (const PwLaunchPlan& L : u.pwl) {   // ... which needs that branch in a launch of its own
          bool has2 = false, other = false;
          for (const PwPassPlan& pp : L.passes) { has2 = has2 || pp.out_branch == 2; other = other || pp.out_branch != 2; }
          if (has2 && other) ok = false;
        }
      }
    }
  }
  if (!ok) { u.pw4l.clear(); u.pw4_old_mask = 0; return CSN_OK; }
  u.pw4 = 1;
  if (d.n_in == 3) plan_hz(bl, u, ci_off, co_off, cin_tot);
  return CSN_OK;
}

int plan_dw(Builder& bl, UnitPlan& u) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  if (d.n_in != d.n_out || d.n_in < 1 || d.n_in > 3) FAIL(CSN_E_INVALID, "csn_plan.hip:308");
  u.kname = "dw3x3_bn_prelu_kernel";
  for (int k = 0; k < d.n_in; ++k) {
    if (d.cout[k] == 0) continue;
    if (d.cin[k] != d.cout[k] || d.in_act[k] < 0 || d.out_act[k] < 0) FAIL(CSN_E_INVALID, "csn_plan.hip:312");
    const Act& ai = P.acts[d.in_act[k]];
    const Act& ao = P.acts[d.out_act[k]];
    if (ai.channels != d.cin[k] || ao.channels != d.cout[k] || ai.lvl != ao.lvl) FAIL(CSN_E_INVALID, "csn_plan.hip:315");
    if (!bn_ok(d.bn[k]) || d.w_off[k] < 0) FAIL(CSN_E_INVALID, "csn_plan.hip:316");
    u.dw_w[k] = bl.alloc_packed((int64_t)d.cout[k] * 9);
    bl.job(CSN_PREP_COPY, d.cout[k] * 9, u.dw_w[k], d.w_off[k], -1, -1, -1, 100.0f);  // conv2d.py:104
    u.dw_epi[k] = bl.bn_epi(d.bn[k], d.cout[k]);
    u.out_epi[k] = u.dw_epi[k];
  }
  return CSN_OK;
}

int plan_ms(Builder& bl, UnitPlan& u) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  u.kname = "msblock_kernel";
  if (d.in_act[0] < 0 || d.out_act[0] < 0) FAIL(CSN_E_INVALID, "ms acts");
  const Act& ai = P.acts[d.in_act[0]];
  const Act& ao = P.acts[d.out_act[0]];
  int tot = 0;
  for (int k = 0; k < CSN_NDIL; ++k) tot += d.dil_ch[k];
  if (ai.channels != d.cin[0] || ao.channels != d.cout[0] || tot != d.cout[0] || ai.lvl != ao.lvl) FAIL(CSN_E_INVALID, "ms shapes");
  if (!bn_ok(d.bn[0])) FAIL(CSN_E_INVALID, "ms BN");
  u.base_lvl = ai.lvl;
  const int cinp = (d.cin[0] + 1) & ~1;
  for (int k = 0; k < CSN_NDIL; ++k) {     // dilation 2^k, padding 2^k (csnet.py:127-135), weight x100 (conv2d.py:104)
    if (d.dil_ch[k] == 0) continue;
    if (d.w_off[k] < 0) FAIL(CSN_E_INVALID, "ms weight");
    u.ms_w[k] = bl.alloc_packed((int64_t)((d.dil_ch[k] + 7) / 8) * cinp * 72);
    bl.job(CSN_PREP_C3, d.dil_ch[k], u.ms_w[k], d.w_off[k], -1, -1, -1, 100.0f, d.cin[0], d.cin[0], cinp, 0);
  }
  u.ms_epi = bl.bn_epi(d.bn[0], d.cout[0]);
  u.out_epi[0] = u.ms_epi;
  return CSN_OK;
}

int plan_cls(Builder& bl, UnitPlan& u) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  u.kname = "goct_pw_kernel";
  if (d.in_act[0] < 0 || d.w_off[0] < 0 || d.bias_off < 0) FAIL(CSN_E_INVALID, "cls: missing tensors");
  const Act& ai = P.acts[d.in_act[0]];
  if (ai.channels != d.cin[0] || ai.lvl != 1) FAIL(CSN_E_INVALID, "cls: input must be at H/2");  // csnet.py:380-385
  u.base_lvl = 1;
  u.logits_off = bl.alloc_act(HZ_MAX_GROUPS, 1);   // (hz_kernel's row reduction leaves one partial plane per M group, k_head.hip)
  PwLaunchPlan L;
  L.lvl = 1;
  PwPassPlan ps;
  ps.r = 0; ps.nsrc = 1; ps.src_kind[0] = SRC_IN; ps.src_branch[0] = 0; ps.src_C[0] = d.cin[0]; ps.src_mode[0] = PW_OWN;
  ps.K = d.cin[0]; ps.nrows = 1; ps.out_kind = OUT_LOGITS; ps.out_ctot = 1;
  ps.epi.scale = bl.alloc_packed(1); ps.epi.shift = bl.alloc_packed(1); ps.epi.alpha = bl.alloc_packed(1);
  bl.job(CSN_PREP_FILL, 1, ps.epi.scale, -1, -1, -1, -1, 1.f);
  bl.job(CSN_PREP_COPY, 1, ps.epi.shift, d.bias_off);
  bl.job(CSN_PREP_FILL, 1, ps.epi.alpha, -1, -1, -1, -1, 1.f);
  WBlock w; w.src = d.w_off[0]; w.ld = d.cin[0]; w.ncol = d.cin[0]; w.col = 0;
  ps.wb.push_back(w);
  L.passes.push_back(ps);
  u.pwl.push_back(L);
  return finish_launch(bl, u.pwl.back());
}

// Whole ILBlock on ilb_kernel (k_ilb.hip): unit k is a two-input 1x1 gOctaveCBR whose outputs are read only by the fused depthwise
// pair (k + 1, k + 2), and the planes of a group of output channels fit the LDS of a CU.  Lays out the per-group weight images
// (pw4_kernel's [K][4][P] form, uniform groups of nth high + ntl low row tiles) and the epilogue records.
bool dw_pair_writes_mp(const csn_plan& P, const UnitPlan& pair, int k);

int plan_ilb(Builder& bl, int k) {
  csn_plan& P = bl.P;
  if (k + 2 >= (int)P.units.size()) return CSN_OK;
  UnitPlan& u = P.units[k];
  const csn_unit_desc& d = u.d;
  const UnitPlan& d1 = P.units[k + 1];
  const UnitPlan& d2 = P.units[k + 2];
  if (d.kind != CSN_UNIT_GOCT || u.std_conv || d.n_out < 1 || d.n_out > 2 || d.cout[0] <= 0) return CSN_OK;
  const bool one = d.ksize == 1 && d.stride == 1 && d.n_in == 2 && d.cin[0] > 0 && d.cin[1] > 0;
  // the stride-2 entry block: 3x3, ONE input branch whose 2x2 averages AND their 2x2 maxima the depthwise pair in front delivers
  const bool k3 = d.ksize == 3 && d.stride == 2 && d.n_in == 1 && d.cin[0] > 0 && d.n_out == 2 && d.cout[1] > 0 && u.pooled_by_producer &&
                  u.mp_producer >= 0 && u.mp_off[0] >= 0 && dw_pair_writes_mp(P, P.units[u.mp_producer], 0);
  if (!one && !k3) return CSN_OK;
  if (d1.d.kind != CSN_UNIT_DW || d2.d.kind != CSN_UNIT_DW || !d1.fuse_next || d1.d.n_in != d.n_out) return CSN_OK;
  for (int j = 0; j < d.n_out; ++j) {
    if (d1.d.cin[j] != d.cout[j]) return CSN_OK;
    if (d.cout[j] == 0) continue;
    if (d1.d.in_act[j] != d.out_act[j]) return CSN_OK;
    for (int q = 0; q < (int)P.units.size(); ++q) {   // nobody but the pair reads the 1x1 unit's outputs
      if (q == k + 1) continue;
      for (int s = 0; s < CSN_MAX_BRANCH; ++s)
        if (s < P.units[q].d.n_in && P.units[q].d.cin[s] > 0 && P.units[q].d.in_act[s] == d.out_act[j]) return CSN_OK;
    }
  }
  const int OH = d.cout[0], OL = d.n_out >= 2 ? d.cout[1] : 0;
  int cin_tot = k3 ? d.cin[0] : d.cin[0] + d.cin[1];
  const int co_off[2] = {0, OH}, ci_off[2] = {0, d.cin[0]};
  IlbArgs a = {};
  a.k3 = k3 ? 1 : 0;
  a.CH = d.cin[0]; a.CL = k3 ? d.cin[0] : d.cin[1]; a.OH = OH; a.OL = OL;
  a.Hl = P.H >> (u.base_lvl + 1); a.Wl = P.W >> (u.base_lvl + 1); a.B = P.S;
  a.Rh = 4; a.Rl = 4;
  a.nth = P.ilb_nt; a.ntl = OL > 0 ? P.ilb_nt : 0;
  if (P.ilb_nt == 0) {
    // Row tiles per group and branch: the fewest rounds of blocks over the 256 CUs (a block is one latency chain: loads -> contraction
    // -> planes -> depthwise pair -> stores), then the fewest groups (every group re-reads the image's inputs).  Measured on the
    // harness (profiles/r5_notes.md): stage 4.2 with 11 groups of (1, 1) starts its blocks in two waves (28.9 us), stage 4.3 with
    // (2, 0) 22.4 instead of 26.6 us.
    const int th_ = (OH + 3) / 4, tl_ = (OL + 3) / 4;
    int64_t best = -1;
    for (int nh = 1; nh <= (k3 ? 1 : 2); ++nh)   // (the 3x3 form with eight high channels per group spills: 112-160 B of scratch per lane)
      for (int nl = (OL > 0 ? 1 : 0); nl <= (OL > 0 ? 2 : 0); ++nl) {
        if (!csn_ilb_supported(nh, nl)) continue;
        IlbArgs t = a;
        t.nth = nh; t.ntl = nl;
        const size_t l = csn_ilb_layout(t);
        if (l == 0 || l > 160 * 1024) continue;
        const int g = std::max((th_ + nh - 1) / nh, nl > 0 ? (tl_ + nl - 1) / nl : 0);
        // blocks per CU: by LDS and by registers (launch bound 1024 = at most 128 VGPRs: 16 waves per CU)
        const size_t wblk = (size_t)(t.nthreads + 63) / 64;
        const int64_t slots = 256 * (int64_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / l, 16 / wblk));
        const int64_t rounds = ((int64_t)P.S * g + slots - 1) / slots;
        const int64_t cost = rounds * 1000 + g;
        if (best < 0 || cost < best) { best = cost; a.nth = nh; a.ntl = nl; }
      }
    if (best < 0) return CSN_OK;
  }
  if (!csn_ilb_supported(a.nth, a.ntl)) return CSN_OK;
  // Measured (round 5, tools/probes/ilb_bench.hip): every group of an image re-reads ALL of the image's input channels, and on the
  // 56^2 / 28^2 maps of stage 3 (7-10 groups, 370 KB of input per group, one 155 KB block per CU) the launch is bound by those
  // re-reads: 45-50 us against 35-40 us for pw4_kernel + the depthwise pair.  Stage 4 (28^2 / 14^2: 18-24 us against 29-37 us) and
  // smaller planes win on their own; under the two-lane slice schedule stage 3 wins as well (ilb_maxpix above).  CSN_ILB_MAXPIX
  // overrides the low-plane pixel limit.
  if (a.Hl * a.Wl > P.ilb_maxpix) return CSN_OK;
  size_t lds = csn_ilb_layout(a);
  if ((lds == 0 || lds > 160 * 1024) && a.nth > 1) {   // the planes of a narrower group
    a.nth = 1; a.ntl = OL > 0 ? 1 : 0;
    lds = csn_ilb_layout(a);
  }
  if (lds == 0 || lds > 160 * 1024) return CSN_OK;
  const int th = (OH + 3) / 4, tl = (OL + 3) / 4;
  const int ng = std::max((th + a.nth - 1) / a.nth, a.ntl > 0 ? (tl + a.ntl - 1) / a.ntl : 0);
  const int NT4 = (a.nth + a.ntl + 3) & ~3, Pp = PW4_PITCH(NT4);
  UnitPlan::Ilb& I = u.ilb;
  I.nth = a.nth; I.ntl = a.ntl; I.ng = ng; I.gimg = a.gimg_floats; I.Rh = a.Rh; I.Rl = a.Rl; I.k3 = a.k3;
  I.wimg = bl.alloc_packed((int64_t)ng * I.gimg);
  for (int g = 0; g < ng; ++g) {
    const int64_t img = I.wimg + (int64_t)g * I.gimg;
    const int r0h = 4 * a.nth * g, r0l = 4 * a.ntl * g;
    const int nrh = std::max(0, std::min(4 * a.nth, OH - r0h)), nrl = a.ntl > 0 ? std::max(0, std::min(4 * a.ntl, OL - r0l)) : 0;
    if (k3) {   // [9 C][4][P]: gathered entry 9 c + tap (CSN_PREP_C3Q), high rows in tiles 0 .., low rows behind them
      const int ld9 = cin_tot * 9;
      if (nrh > 0)
        bl.job(CSN_PREP_C3Q, nrh, img, d.w_off[0] + (int64_t)(co_off[0] + r0h) * ld9, -1, -1, -1, 1.f, ld9, d.cin[0], Pp, 0 | (0 << 8));
      if (nrl > 0)
        bl.job(CSN_PREP_C3Q, nrl, img, d.w_off[0] + (int64_t)(co_off[1] + r0l) * ld9, -1, -1, -1, 1.f, ld9, d.cin[0], Pp, a.nth | (0 << 8));
      continue;
    }
    int k0 = 0;
    for (int i = 0; i < 2; ++i) {   // gathered channels: branch 0 (high), branch 1 (low)
      if (nrh > 0)
        bl.job(CSN_PREP_PW4, nrh, img, d.w_off[0] + (int64_t)(co_off[0] + r0h) * cin_tot + ci_off[i], -1, -1, -1, 1.f, cin_tot, d.cin[i],
               Pp, 0 | (k0 << 8));
      if (nrl > 0)
        bl.job(CSN_PREP_PW4, nrl, img, d.w_off[0] + (int64_t)(co_off[1] + r0l) * cin_tot + ci_off[i], -1, -1, -1, 1.f, cin_tot, d.cin[i],
               Pp, a.nth | (k0 << 8));
      k0 += d.cin[i];
    }
  }
  for (int q = 0; q < 2; ++q) {
    const int C = q == 0 ? OH : OL;
    if (C <= 0) continue;
    const int rows = 4 * ng * (q == 0 ? a.nth : a.ntl) + 4;
    I.ep[q] = bl.alloc_packed((int64_t)rows * 4);
    bl.job(CSN_PREP_BN_SCALE, C, I.ep[q], d.bn[q].weight, d.bn[q].running_var, -1, -1, 1.f, 0, 0, 4, 0);
    bl.job(CSN_PREP_BN_SHIFT, C, I.ep[q], d.bn[q].weight, d.bn[q].running_var, d.bn[q].bias, d.bn[q].running_mean, 1.f, 0, 0, 4, 1);
    bl.job(CSN_PREP_COPY, C, I.ep[q], d.bn[q].prelu, -1, -1, -1, 1.f, 0, 0, 4, 2);
    // the depthwise pair's parameters of the branch as one record per channel (staged in LDS with the weight image)
    I.dwrec[q] = bl.alloc_packed((int64_t)rows * 24);
    for (int h = 0; h < 2; ++h) {
      const csn_unit_desc& w = h == 0 ? d1.d : d2.d;
      const int o = 12 * h;
      bl.job(CSN_PREP_DWREC, C, I.dwrec[q], w.w_off[q], w.bn[q].weight, w.bn[q].running_var, -1, 100.0f, 0, 0, 24, o);   // conv2d.py:104
      bl.job(CSN_PREP_BN_SHIFT, C, I.dwrec[q], w.bn[q].weight, w.bn[q].running_var, w.bn[q].bias, w.bn[q].running_mean, 1.f, 0, 0, 24, o + 9);
      bl.job(CSN_PREP_COPY, C, I.dwrec[q], w.bn[q].prelu, -1, -1, -1, 1.f, 0, 0, 24, o + 10);
    }
  }
  I.on = 1;
  return CSN_OK;
}

// ------------------------------------------------------------------------------------ execution
struct Ctx {
  csn_plan& P;
  const float* x;  // slice pointers
  float* y;
  char* ws;
  void* stream;
  const float* act_in(int id) const {
    const Act& a = P.acts[id];
    if (a.ws_off < 0) return a16 ? reinterpret_cast<const float*>(ws + P.x16_off) : x;
    return reinterpret_cast<const float*>(ws + a.ws_off);
  }
  // pointer `elems` activation elements past p (argument blocks carry float* whatever the element type)
  const float* eo(const float* p, int64_t elems) const {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + elems * (a16 ? 2 : 4));
  }
  float* eo(float* p, int64_t elems) const {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + elems * (a16 ? 2 : 4));
  }
  // train mode with backward buffers: the convolutions write z next to (not over) the activation y
  float* act_out(int id) const {
    return reinterpret_cast<float*>(ws + ((raw && P.train) ? P.tz_off[id] : P.acts[id].ws_off));
  }
  float* act_y(int id) const { return reinterpret_cast<float*>(ws + P.acts[id].ws_off); }
  const float* pk(int64_t off) const { return P.packed + off; }
  bool raw = false;   // train mode: convolutions write the un-normalised z (identity epilogue)
  bool lanes = false; // eval forward outside profiling: independent launches may go to the plan's auxiliary streams
  bool side = false;  // backward: this context enqueues on the weight-gradient side lane (own partial buffers)
  bool a16 = false;   // bf16 train mode: every activation tensor in the workspace is bfloat16
  bool dw_stats = false;   // train-mode forward: depthwise units reduce their own BN statistics
  bool pw4_stats = false;  // ... and this 1x1 unit's pw4_kernel launches theirs (bf16 mode, pw4_unit_stats)
  std::vector<GapTilesArgs>* gap_defer = nullptr;   // train-mode forward: |GAP| table jobs collected for one launch at the end
  const float* sc(const Epi& e) const { return P.packed + (raw ? P.ident.scale : e.scale); }
  const float* sh(const Epi& e) const { return P.packed + (raw ? P.ident.shift : e.shift); }
  const float* al(const Epi& e) const { return P.packed + (raw ? P.ident.alpha : e.alpha); }
  // profiling: close the interval of the launch that was just enqueued
  int mark(const char* kernel) const {
    if (!P.profiling) return CSN_OK;
    if (P.ev_used >= P.ev.size()) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      P.ev.push_back(e);
    }
    HIP_TRY(hipEventRecord(P.ev[P.ev_used++], (hipStream_t)stream));
    P.tags.push_back(kernel);
    return CSN_OK;
  }
};

// Lanes per row of the depthwise kernels' blocks (a lane owns 4 columns; the block's 256 lanes are LX x NY, NY = 256 / LX): whole rows
// up to 64 lanes (256 columns).  Wider rows: the LX in [14, 64] with the largest (covered columns) x (lanes used) product instead
// of 64 and a mostly empty last tile (40 x 6 at 320 columns).  Measured at 224 columns (round 4, bf16 step): 28 x 9 lanes in two
// tiles per row (252 of 256 lanes busy instead of 224) is 0.3-0.4 ms SLOWER than 56 x 4 -- the half-empty fourth wave costs less
// than the shorter row segments and the extra tile row (the search stays off for rows of up to 64 lanes).
// pow2 (csn_plan::dw_xl, train-mode launches): a plane that fits one tile in x gets a power-of-two group of lanes per row, so that
// every row of lanes sits inside ONE wave and the kernels take their halo columns from the neighbouring lanes (dwx_row_of, k_misc.hip)
int dw_lanes_x(int cols, bool pow2 = false) {
  if (pow2 && cols <= 64) {
    int g = 1;
    while (g < cols) g <<= 1;
    return g;
  }
  if (cols <= 64) return cols;
  int best = 14;
  double best_s = -1;
  for (int LX = 14; LX <= 64 && LX <= cols; ++LX) {
    const int tx = (cols + LX - 1) / LX;
    const double s = (double)cols / ((double)tx * LX) * (double)((CSN_BLOCK / LX) * LX) / CSN_BLOCK;
    if (s >= best_s - 1e-12) { best_s = s > best_s ? s : best_s; best = LX; }   // ties: the widest
  }
  return best;
}

// rows per lane of the fused depthwise pair: the intermediate tile (NY*R + 2 rows) must stay small in LDS
int choose_dw2_rows(int H, int NY, int LX, bool even = false, bool quad = false) {
  const int step = quad ? 4 : (even ? 2 : 1);   // even: the pair also writes 2x2 averages; quad: ... and their 2x2 maxima
  int best = step;
  double best_s = -1;
  for (int R = step; R <= 16; R += step) {
    const int rows = NY * R;
    const size_t lds = (size_t)(rows + 2) * (LX * 4 + 8) * 4;
    if (lds > 36 * 1024 && R > step) break;   // (56 KB = 56 rows at 224^2, 7 % instead of 12.5 % halo re-reads: 0.93 vs 0.82 ms)
    const int tiles = (H + rows - 1) / rows;
    const double eff = (double)H / ((double)tiles * rows);
    const double s = eff * rows / (rows + 4.0);   // halo rows are fetched and computed twice
    if (s > best_s + 1e-9) { best_s = s; best = R; }
  }
  return best;
}

int choose_dw_rows(int H, int NY) {
  int best = 4;
  double best_s = -1;
  for (int R = 4; R <= 16; ++R) {
    const int rows = NY * R;
    const int tiles = (H + rows - 1) / rows;
    const double eff = (double)H / ((double)tiles * rows);
    const double s = eff * R / (R + 2.0);
    if (s > best_s + 1e-9) { best_s = s; best = R; }
  }
  return best;
}

// statistics partials per channel when the train-mode depthwise kernel reduces its own output (one per image and tile), 0 when
// they would not fit the [C][CSN_BN_NSLAB] table
int dw_stats_slabs(const csn_plan& P, int lvl) {
  if (std::getenv("CSN_DW_STATS") && std::getenv("CSN_DW_STATS")[0] == '0') return 0;
  const int H = P.H >> lvl, W = P.W >> lvl;
  const int cols = (W + 3) / 4;
  const int LX = dw_lanes_x(cols, P.dw_xl), NY = CSN_BLOCK / LX;
  const int tiles_x = (cols + LX - 1) / LX;
  const int R = choose_dw_rows(H, NY);
  const int tiles_y = (H + NY * R - 1) / (NY * R);
  const int64_t n = (int64_t)P.S * tiles_x * tiles_y;
  return n <= CSN_BN_NSLAB ? (int)n : 0;
}
// The fused depthwise pair in front of a stride-2 3x3 unit writes that unit's avg-pooled inputs; when the unit runs on c3q_kernel
// its high -> low slice of branch k reads a 2x2 max-pooled copy of them: the pair writes that too (no pool2_kernel launch)
bool dw_pair_writes_mp(const csn_plan& P, const UnitPlan& pair, int k) {
  if (pair.pool_unit < 0 || !P.fuse_dw || !P.c3q || !P.tiled3 || P.no_mp_fuse) return false;
  const UnitPlan& su = P.units[pair.pool_unit];
  if (su.d.ksize != 3 || su.mp_off[k] < 0) return false;
  bool uses = false;
  for (const PwLaunchPlan& L : su.pwl) uses = uses || L.c3q;
  const Act& act = P.acts[pair.d.in_act[k]];
  const int H = P.H >> act.lvl, W = P.W >> act.lvl;
  return uses && (H % 4) == 0 && (W % 4) == 0;
}

// ... for every branch of a depthwise unit, or for none
bool dw_unit_stats(const csn_plan& P, const UnitPlan& u) {
  if (u.d.kind != CSN_UNIT_DW) return false;
  for (int k = 0; k < u.d.n_in; ++k)
    if (u.d.cout[k] > 0 && dw_stats_slabs(P, P.acts[u.d.in_act[k]].lvl) == 0) return false;
  return true;
}

// pointers a launch's sources / outputs resolve to, by SrcKind / OutKind and branch
struct PwBind {
  const float* in[3] = {nullptr, nullptr, nullptr};    // SRC_IN: unit input branch i (pooled copy for stride 2)
  const float* mp[3] = {nullptr, nullptr, nullptr};    // 2x2 max-pooled copy of input branch i (c3q_kernel's high -> low slices)
  const float* z = nullptr;                            // SRC_Z
  const float* dz[3] = {nullptr, nullptr, nullptr};    // SRC_DZ
  const float* adj[4] = {nullptr, nullptr, nullptr, nullptr};   // SRC_ADJ
  const float* dxsrc[3] = {nullptr, nullptr, nullptr}; // SRC_DX
  float* act[3] = {nullptr, nullptr, nullptr};         // OUT_ACT: output act of branch j
  float* zout = nullptr;                               // OUT_Z
  float* logits = nullptr;                             // OUT_LOGITS
  float* dx[3] = {nullptr, nullptr, nullptr};          // OUT_DX
  float* tmp = nullptr;                                // OUT_TMP
  const float* red_w = nullptr;                        // fused cls_layer: weights / bias, result -> `logits`
  const float* red_b = nullptr;
  const float* route_x = nullptr;                      // OUT_DX launch on pwq_kernel: the max-pool adjoint routed in its epilogue --
  const float* route_t = nullptr;                      // the pooled input branch / the low-resolution gradient (PwqArgs::route_x)
};

// kernel-side descriptor of one pass: source slices (resolved pointers), weight rows, output, epilogue
void fill_pass(const Ctx& c, const PwLaunchPlan& L, const PwPassPlan& pp, const PwBind& bd, PwPass& ps) {
  const csn_plan& P = c.P;
  ps.r = pp.r; ps.nsrc = pp.nsrc;
  const int Hr = P.H >> (L.lvl + pp.r), Wr = P.W >> (L.lvl + pp.r);
  for (int s = 0; s < 3; ++s) {
    ps.src[s].ptr = nullptr; ps.src[s].C = 0; ps.src[s].Ctot = 0; ps.src[s].mode = PW_OWN; ps.src[s].K = 0;
    ps.src[s].dil = 1; ps.src[s].pad = 0;
    if (s < pp.nsrc) {
      const int br = pp.src_branch[s], mode = pp.src_mode[s];
      const float* base = nullptr;
      switch (pp.src_kind[s]) {
        case SRC_Z: base = bd.z; break;
        case SRC_DZ: base = bd.dz[br]; break;
        case SRC_ADJ: base = bd.adj[br]; break;
        case SRC_DX: base = bd.dxsrc[br]; break;
        default: base = bd.in[br]; break;
      }
      // resolution of the source tensor relative to the pass (channel offset -> plane offset)
      int sh = 0;
      if (mode == PW_POOL2 || mode == PW_POOL2_TAPS || mode == PW_TAPS_S2) sh = -1;
      else if (mode == PW_POOL4) sh = -2;
      else if (mode == PW_UP2 || mode == PW_TAPS_UPS2) sh = 1;
      else if (mode == PW_UP4) sh = 2;
      const int64_t hs = sh >= 0 ? (int64_t)(Hr >> sh) * (Wr >> sh) : (int64_t)(Hr << -sh) * (Wr << -sh);
      ps.src[s].ptr = c.eo(base, (int64_t)pp.src_c0[s] * hs);
      ps.src[s].C = pp.src_C[s];
      ps.src[s].Ctot = pp.src_ctot[s] > 0 ? pp.src_ctot[s] : pp.src_C[s];
      ps.src[s].mode = mode;
      ps.src[s].K = pw_mode_taps(mode) ? 9 * pp.src_C[s] : pp.src_C[s];
      ps.src[s].dil = pp.src_dil[s];
    }
  }
  ps.cin = pp.K; ps.cin4 = pp.K4; ps.nrows = pp.nrows; ps.w_off = pp.w_off; ps.w_stride = pp.w_stride;
  float* ob = nullptr;
  switch (pp.out_kind) {
    case OUT_Z: ob = bd.zout; break;
    case OUT_LOGITS: ob = bd.logits; break;
    case OUT_DX: ob = bd.dx[pp.out_branch]; break;
    case OUT_TMP: ob = bd.tmp; break;
    default: ob = bd.act[pp.out_branch]; break;
  }
  ps.out = ob ? c.eo(ob, (int64_t)pp.out_c0 * Hr * Wr) : nullptr;
  ps.red_w = ps.red_b = nullptr;
  if (bd.red_w && pp.out_kind == OUT_ACT) {   // rows are reduced into the half-resolution logits
    ps.red_w = bd.red_w + pp.out_c0; ps.red_b = bd.red_b; ps.out = bd.logits;
  }
  ps.out_ctot = pp.out_ctot; ps.pad2 = 0;
  const bool bn_out = pp.out_kind == OUT_ACT;   // scratch, logits and gradients have no BN
  ps.scale = bn_out ? c.sc(pp.epi) : c.pk(pp.epi.scale);
  ps.shift = bn_out ? c.sh(pp.epi) : c.pk(pp.epi.shift);
  ps.alpha = bn_out ? c.al(pp.epi) : c.pk(pp.epi.alpha);
}

// item tiles of a pw4_kernel launch over an Hl x Wl low map: row segments, or 64 consecutive pixels where rows do not fill their tiles
void pw4_tile_geo(const csn_plan& P, int Hl, int Wl, int* ptwl, int* ptx, int* pty) {
  int twl = 0;
  while (twl < P.pw4_twl && (1 << twl) < Wl) ++twl;
  int tx = (Wl + (1 << twl) - 1) >> twl, ty = (Hl + (64 >> twl) - 1) / (64 >> twl);
  if (P.pw4_flat && tx * ty > (Hl * Wl + 63) / 64) { twl = PW4_FLAT_TWL; tx = (Hl * Wl + 63) / 64; ty = 1; }
  *ptwl = twl; *ptx = tx; *pty = ty;
}
// bf16 train forward: every output branch of the unit gets its statistics from pw4_kernel's epilogue (UnitPlan::pstats_off, laid
// out by csn_plan_enable_training)
bool pw4_unit_stats(const csn_plan& P, const UnitPlan& u) {
  if (!P.pw4_stats || !P.pw4 || !u.pw4 || u.d.kind != CSN_UNIT_GOCT) return false;
  for (int j = 0; j < u.d.n_out; ++j)
    if (u.d.cout[j] > 0 && u.pstats_off[j] < 0) return false;
  return true;
}

int launch_pw(const Ctx& c, const PwLaunchPlan& L, const PwBind& bd) {
  const csn_plan& P = c.P;
  PwArgs a;
  a.npass = (int)L.passes.size();
  a.H0 = P.H >> L.lvl; a.W0 = P.W >> L.lvl; a.B = P.S;
  // tile height: 16 rows, or 8 / 4 for small maps so that the launch still fills the 256 CUs several times over
  int rmax = 0;
  for (const PwPassPlan& pp : L.passes) rmax = std::max(rmax, pp.r);
  a.ty_log2 = std::max(4 - (PW_TXL - 5), rmax);   // 512 pixels of branch 0 per tile whatever its width
  a.tiles_x = (a.W0 + PW_TX0 - 1) / PW_TX0;
  while (a.ty_log2 > 2 && a.ty_log2 > rmax &&
         (int64_t)a.tiles_x * ((a.H0 + (1 << a.ty_log2) - 1) >> a.ty_log2) * a.B < 512) --a.ty_log2;
  a.tiles_y = (a.H0 + (1 << a.ty_log2) - 1) >> a.ty_log2;
  a.wimg = c.pk(L.wimg); a.wimg_floats = L.wimg_floats;
  a.wimg3 = L.wimg3 >= 0 ? c.pk(L.wimg3) : nullptr; a.w3_stride = L.w3_stride; a.w3_floats = L.w3_floats; a.z_c0 = L.z_c0; a.a16 = c.a16 ? 1 : 0;
  for (int q = 0; q < a.npass; ++q) {
    const PwPassPlan& pp = L.passes[q];
    PwPass& ps = a.pass[q];
    fill_pass(c, L, pp, bd, ps);
  }
  // raw: every pass of the launch stores plain sums (train-mode conv outputs, gradients, scratch)
  bool all_raw = true;
  for (const PwPassPlan& pp : L.passes) all_raw = all_raw && (pp.out_kind == OUT_DX || pp.out_kind == OUT_TMP || (c.raw && pp.out_kind != OUT_LOGITS));
  if (P.pw4 && L.pwq && all_raw) {   // input-gradient launch of a 1x1 unit: plain contraction over flat planes
    const PwPass& ps = a.pass[0];
    PwqArgs q;
    q.nsrc = ps.nsrc; q.nrows = ps.nrows;
    for (int s = 0; s < 3; ++s) { q.src[s].ptr = ps.src[s].ptr; q.src[s].C = ps.src[s].C; q.src[s].Ctot = ps.src[s].Ctot; }
    q.out = ps.out; q.out_ctot = ps.out_ctot;
    q.wimg = c.pk(L.pwq_wimg);
    q.HW = a.H0 * a.W0; q.B = a.B;
    q.ngroups = L.pwq_ng; q.gimg_floats = L.pwq_gimg; q.nt = L.pwq_nt; q.max_grid = P.pw4_grid; q.a16 = c.a16 ? 1 : 0;
    q.mfma16 = (c.a16 && P.pwq16) ? 1 : 0;
    for (int g = 0; g < PWQ_MAX_GROUPS; ++g) { q.grp_r0[g] = L.pwq_r0[g]; q.grp_nt[g] = L.pwq_gnt[g]; }
    bool ok = q.out != nullptr;
    for (int s = 0; s < q.nsrc; ++s) ok = ok && q.src[s].ptr != nullptr;
    if (bd.route_x) {   // (the caller has checked pwq_route_ok for this launch)
      const PwPassPlan& pp = L.passes[0];
      q.route_x = c.eo(bd.route_x, (int64_t)pp.out_c0 * q.HW);
      q.route_t = c.eo(bd.route_t, (int64_t)pp.out_c0 * (q.HW >> 2));
      q.route_W = a.W0;
      if (!ok) { g_hip_err = "launch_pw: routed max-pool adjoint without its pwq launch"; return CSN_E_STATE; }
    }
    if (ok) {
      LAUNCH_TRY(csn_launch_pwq(q, c.stream));
      return c.mark("pwq_kernel");
    }
  }
  if (P.c3q && P.tiled3 && L.c3q && (!c.a16 || c.raw)) {   // 3x3 forward pass: lane = output quad, operands from the load registers
    const PwPassPlan& pp = L.passes[0];
    bool ok = true;
    C3qArgs q;
    q.nsrc = L.c3q_ntap;
    for (int s = 0; s < 3; ++s) { q.src[s].ptr = nullptr; q.src[s].C = 0; q.src[s].Ctot = 0; }
    for (int s = 0; s < L.c3q_ntap; ++s) {
      const int br = pp.src_branch[s];
      q.src[s].ptr = pp.src_kind[s] == SRC_DZ ? bd.dz[br] : pp.src_kind[s] == SRC_ADJ ? bd.adj[br]
                     : pp.src_mode[s] == PW_POOL2_TAPS ? bd.mp[br] : bd.in[br];
      q.src[s].C = pp.src_C[s]; q.src[s].Ctot = pp.src_C[s];
      ok = ok && q.src[s].ptr != nullptr;
    }
    q.z = L.c3q_z ? bd.z : nullptr; q.z_ctot = L.c3q_z ? pp.src_C[L.c3q_ntap] : 0; q.z_c0 = L.z_c0;
    ok = ok && (!L.c3q_z || q.z != nullptr);
    const bool gradq = pp.out_kind == OUT_DX || pp.out_kind == OUT_TMP;
    float* ob = pp.out_kind == OUT_Z ? bd.zout : pp.out_kind == OUT_DX ? bd.dx[pp.out_branch] : pp.out_kind == OUT_TMP ? bd.tmp
                                                                                                : bd.act[pp.out_branch];
    ok = ok && ob != nullptr && (pp.out_kind == OUT_Z || pp.out_kind == OUT_ACT || gradq) && !bd.red_w;
    if (bd.route_x) {   // the max-pool adjoint in this launch's epilogue (the caller has checked c3q_route_ok)
      ok = ok && pp.out_kind == OUT_DX && !L.c3q_z;
      q.route_x = bd.route_x; q.route_t = bd.route_t;
    }
    if (ok) {
      q.out = ob; q.out_c0 = pp.out_c0; q.out_ctot = pp.out_ctot; q.nrows = pp.nrows;
      q.ep = L.c3q_ep >= 0 ? c.pk(L.c3q_ep) : nullptr;
      q.wimg = c.pk(L.c3q_wimg);
      q.H = a.H0; q.W = a.W0; q.B = a.B;
      const int Wq = q.W >> 1, Hq = q.H >> 1;
      int twl = 0;
      while (twl < P.c3q_twl && (1 << twl) < Wq) ++twl;
      q.twl = twl;
      q.tiles_x = (Wq + (1 << twl) - 1) >> twl;
      q.tiles_y = (Hq + (64 >> twl) - 1) / (64 >> twl);
      if (P.pw4_flat && q.tiles_x * q.tiles_y > (Hq * Wq + 63) / 64) {
        q.twl = PW4_FLAT_TWL; q.tiles_x = (Hq * Wq + 63) / 64; q.tiles_y = 1;
      }
      q.ngroups = L.c3q_ng; q.gimg_floats = L.c3q_gimg; q.nt = L.c3q_nt; q.max_grid = P.pw4_grid;
      q.a16 = c.a16 ? 1 : 0;
      q.mfma16 = (c.a16 && gradq && P.c3q16 >= 1) ? 1 : 0;   // (forward launches keep fp32 weights: c3q_kernel<bf16>)
      q.hl = (P.c3q_hl && !c.a16) ? 1 : 0;
      if (q.hl) { q.twl = PW4_FLAT_TWL; q.tiles_x = (Hq * Wq + 61) / 62; q.tiles_y = 1; }
      for (int g = 0; g < PW4_MAX_GROUPS; ++g) { q.grp_r0[g] = L.c3q_r0[g]; q.grp_nt[g] = L.c3q_gnt[g]; }
      const bool rawq = c.raw || pp.out_kind == OUT_Z || gradq;
      LAUNCH_TRY(csn_launch_c3q(q, rawq ? 1 : 0, c.stream));
      return c.mark("c3q_kernel");
    }
  }
  if (bd.route_x) { g_hip_err = "launch_pw: routed max-pool adjoint on a launch that is neither pwq_kernel's nor c3q_kernel's"; return CSN_E_STATE; }
  if (P.tiled3 && csn_c3_eligible(a) && (!c.a16 || all_raw)) {   // 3x3 pass: LDS-tiled implicit GEMM
    LAUNCH_TRY(csn_launch_c3(a, all_raw ? 1 : 0, c.stream));
    return c.mark("goct_c3_kernel");
  }
  LAUNCH_TRY(csn_launch_pw(a, all_raw ? 1 : 0, c.stream));
  return c.mark("goct_pw_kernel");
}

// ---- stream lanes: fork n auxiliary lanes off the caller's stream / join them back (events; under stream capture these
// become parallel branches of the hipGraph) ----
int lanes_fork(const Ctx& c, int n) {
#ifndef CSN_CPU_EMU
  csn_plan& P = c.P;
  HIP_TRY(hipEventRecord(P.lane_ev[0], (hipStream_t)c.stream));
  for (int k = 0; k < n; ++k) HIP_TRY(hipStreamWaitEvent(P.lane[k], P.lane_ev[0], 0));
#else
  (void)c; (void)n;
#endif
  return CSN_OK;
}
int lanes_join(const Ctx& c, int n) {
#ifndef CSN_CPU_EMU
  csn_plan& P = c.P;
  for (int k = 0; k < n; ++k) {
    HIP_TRY(hipEventRecord(P.lane_ev[1 + k], P.lane[k]));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)c.stream, P.lane_ev[1 + k], 0));
  }
#else
  (void)c; (void)n;
#endif
  return CSN_OK;
}

// The fused depthwise pair runs on dw3x3x2_fast_kernel when EVERY branch qualifies (folded records, width a multiple of four, one
// tile of lanes per row: csn_launch_dw2's own predicate) -- decided once per pair, for the launch geometry, the profile tag and
// csn_unit_kernel_name alike (a mixed pair ran the round-1 kernel under the fast kernel's name and lane geometry)
bool dw_pair_fast(const csn_plan& P, const UnitPlan& first) {
  if (!P.dw_fast) return false;
  for (int k = 0; k < first.d.n_in; ++k) {
    if (first.d.cout[k] == 0) continue;
    const int W = P.W >> P.acts[first.d.in_act[k]].lvl;
    if (first.dw2rec[k] < 0 || (W % 4) != 0 || W > 256) return false;
  }
  return true;
}

int run_unit(const Ctx& c, const UnitPlan& u, const UnitPlan* next = nullptr) {
  const csn_plan& P = c.P;
  const csn_unit_desc& d = u.d;
  const int S = P.S;
  switch (d.kind) {
    case CSN_UNIT_DW: {
      DwArgs a;
      const bool fused = next != nullptr;
      a.nbr = 0; a.B = S; a.a16 = c.a16 ? 1 : 0; a.variant = (P.dw_xl && P.train) ? 2 : 0;
      if (fused && c.a16) return CSN_E_UNSUPPORTED;   // the fused pair is an eval-mode (float) kernel
      int blk = 0;
      const bool pair_fast = fused && !c.raw && dw_pair_fast(P, u);
      for (int k = 0; k < d.n_in; ++k) {
        if (d.cout[k] == 0) continue;
        DwBranch& br = a.br[a.nbr++];
        const Act& act = P.acts[d.in_act[k]];
        br.in = c.act_in(d.in_act[k]);
        const bool virt = c.raw && P.train && !P.virt_cons.empty() && d.in_act[k] > 0 &&
                          P.virt_cons[d.in_act[k]] == (int)(&u - P.units.data());
        if (virt) {   // the producer's raw output + its BatchNorm tables of this step (the activation itself is not stored)
          const int ia = d.in_act[k];
          const UnitPlan& pu = P.units[P.act_prod_unit[ia]];
          const int pj = P.act_prod_branch[ia];
          br.in = reinterpret_cast<const float*>(c.ws + P.tz_off[ia]);
          br.in_scale = c.pk(pu.out_epi[pj].scale); br.in_shift = c.pk(pu.out_epi[pj].shift); br.in_alpha = c.pk(pu.out_epi[pj].alpha);
          br.gapin = reinterpret_cast<double*>(c.ws + u.gapin_off[k]);
        }
        br.out = c.act_out(fused ? next->d.out_act[k] : d.out_act[k]);
        br.w9 = c.pk(u.dw_w[k]);
        br.scale = c.sc(u.dw_epi[k]); br.shift = c.sh(u.dw_epi[k]); br.alpha = c.al(u.dw_epi[k]);
        br.w9b = br.scale_b = br.shift_b = br.alpha_b = nullptr;
        br.pool = nullptr; br.skip_out = 0; br.stats = nullptr; br.xin = nullptr;
        br.C = d.cout[k]; br.H = P.H >> act.lvl; br.W = P.W >> act.lvl;
        const int cols = (br.W + 3) / 4;
        // (the fused pair keeps whole rows of its intermediate in LDS; on the fast kernel a row of lanes is a power-of-two group)
        const bool fastk = pair_fast;
        br.LX = fused ? (cols < 64 ? ((fastk && P.dw_xl) ? dw_lanes_x(cols, true) : cols) : 64) : dw_lanes_x(cols, P.dw_xl && P.train);
        br.NY = CSN_BLOCK / br.LX;
        br.tiles_x = (cols + br.LX - 1) / br.LX;
        if (fused) {
          if (fastk) br.rec = c.pk(u.dw2rec[k]);
          br.w9b = c.pk(next->dw_w[k]);
          br.scale_b = c.pk(next->dw_epi[k].scale); br.shift_b = c.pk(next->dw_epi[k].shift);
          br.alpha_b = c.pk(next->dw_epi[k].alpha);
          const bool pool = u.pool_unit >= 0 && (br.W % 4) == 0;
          const bool mp = pool && dw_pair_writes_mp(P, u, k);
          if (pool) {   // the stride-2 unit that follows reads only the 2x2 averages
            br.pool = reinterpret_cast<float*>(c.ws + P.units[u.pool_unit].pooled_off[k]);
            br.skip_out = u.pool_skip[k];
            if (mp) br.pool_mp = reinterpret_cast<float*>(c.ws + P.units[u.pool_unit].mp_off[k]);
          }
          // (the fast kernel walks its rows in trips of four: R % 4 == 0, which the pooled outputs' 4 x 4 blocks ask for anyway)
          br.R = choose_dw2_rows(br.H, br.NY, fastk ? cols : br.LX, pool, mp || br.rec != nullptr);
        } else {
          br.R = choose_dw_rows(br.H, br.NY);
        }
        br.tiles_y = (br.H + br.NY * br.R - 1) / (br.NY * br.R);
        blk += br.tiles_x * br.tiles_y * br.C * S;
        br.blk_end = blk;
        // train-mode forward: the kernel also leaves the BN statistics partials of its (stored) output
        if (c.dw_stats && !fused && dw_stats_slabs(P, act.lvl) > 0)
          br.stats = reinterpret_cast<double*>(c.ws + u.stats_off[k]);
      }
      if (a.nbr == 0) return CSN_OK;
      if (c.dw_stats) {   // all branches or none (one kernel instantiation per launch)
        bool all = true;
        for (int q = 0; q < a.nbr; ++q) all = all && a.br[q].stats != nullptr;
        if (!all) for (int q = 0; q < a.nbr; ++q) a.br[q].stats = nullptr;
      }
      if (fused) LAUNCH_TRY(csn_launch_dw2(a, c.stream));
      else LAUNCH_TRY(csn_launch_dw(a, c.stream));
      { const int ms_ = c.mark(fused ? (a.br[0].rec ? "dw3x3x2_fast_kernel" : "dw3x3x2_bn_prelu_kernel") : "dw3x3_bn_prelu_kernel");
        if (ms_ != CSN_OK) return ms_; }
      for (int q = 0, k = 0; k < d.n_in; ++k) {   // |mean_hw y| tables of the producers whose y was formed on load here
        if (d.cout[k] == 0) continue;
        const DwBranch& br = a.br[q++];
        if (br.gapin == nullptr) continue;
        const int ia = d.in_act[k];
        const UnitPlan& pu = P.units[P.act_prod_unit[ia]];
        GapTilesArgs ga;
        ga.gapin = br.gapin; ga.gapabs = reinterpret_cast<float*>(c.ws + pu.gap_off[P.act_prod_branch[ia]]);
        ga.C = br.C; ga.S = S; ga.tiles = br.tiles_x * br.tiles_y; ga.pad = 0; ga.HW = (int64_t)br.H * br.W;
        if (c.gap_defer) c.gap_defer->push_back(ga);
        else LAUNCH_TRY(csn_launch_gap_tiles(ga, c.stream));
      }
    } break;
    case CSN_UNIT_GOCT: {
      // optional 2x2 avg-pool prologue of every input branch (csnet.py:679-680)
      const float* xin[3] = {nullptr, nullptr, nullptr};
      if (d.stride == 2 && u.pooled_by_producer && P.fuse_dw && !c.raw) {
        for (int i = 0; i < d.n_in; ++i)
          if (d.cin[i] > 0) xin[i] = reinterpret_cast<const float*>(c.ws + u.pooled_off[i]);
      } else if (d.stride == 2 && !u.std_conv) {
        PoolArgs pa;
        pa.n = 0; pa.a16 = c.a16 ? 1 : 0; pa.pad = 0;
        int blk = 0;
        for (int i = 0; i < d.n_in; ++i) {
          if (d.cin[i] == 0) continue;
          const int k = pa.n++;
          pa.in[k] = c.act_in(d.in_act[i]);
          pa.out[k] = reinterpret_cast<float*>(c.ws + u.pooled_off[i]);
          xin[i] = pa.out[k];
          pa.planes[k] = S * d.cin[i];
          pa.Ho[k] = P.H >> (u.base_lvl + i); pa.Wo[k] = P.W >> (u.base_lvl + i);
          const int64_t lanes = (int64_t)pa.planes[k] * pa.Ho[k] * ((pa.Wo[k] + 1) / 2);
          blk += (int)((lanes + CSN_BLOCK - 1) / CSN_BLOCK);
          pa.blk_end[k] = blk;
        }
        LAUNCH_TRY(csn_launch_pool(pa, c.stream));
        { const int ms_ = c.mark("pool2_kernel"); if (ms_ != CSN_OK) return ms_; }
      } else {
        for (int i = 0; i < d.n_in; ++i)
          if (d.cin[i] > 0) xin[i] = c.act_in(d.in_act[i]);
      }
      const bool cls_next = next && next->d.kind == CSN_UNIT_CLS;
      // (eval: BN + PReLU epilogue, float; train-mode forward: raw sums, float or bfloat16 storage)
      if (P.pw4 && u.pw4 && (c.raw || !c.a16) && (!cls_next || (!c.raw && u.pw4l.size() == 1 && u.pw4l[0].lo_out < 0))) {
        // launches of the unit are independent of each other: parallel stream lanes (see below)
        std::vector<const PwLaunchPlan*> old;
        for (const PwLaunchPlan& L : u.pwl) {
          bool keep = false;
          for (const PwPassPlan& pp : L.passes) keep = keep || ((u.pw4_old_mask >> pp.out_branch) & 1);
          if (keep) old.push_back(&L);
        }
        const int nl = (int)u.pw4l.size() + (int)old.size();
        const bool fork = c.lanes && nl >= 2 && nl <= 3;
        const bool use_hz = P.hz && u.hz.on && !c.raw && !c.a16 && d.stride == 1;
        if (fork) { const int st0 = lanes_fork(c, nl - 1); if (st0 != CSN_OK) return st0; }
        int lane = 0;
        for (const UnitPlan::Pw4Launch& L : u.pw4l) {
          Ctx cl = c;
          if (fork && lane > 0) cl.stream = c.P.lane[lane - 1];
          ++lane;
          if (use_hz && L.hi_out == 0 && L.lo_out < 0) {   // the high output of a three-branch unit: hz_kernel (k_head.hip)
            HzArgs h;
            h.xh = xin[0]; h.x1 = xin[1]; h.x2 = xin[2];
            h.yh = c.act_out(d.out_act[0]); h.part = nullptr; h.red_w = nullptr;
            if (cls_next) {   // cls_layer rides in the epilogue: per-group partial sums, added up (+ bias) by the final upsample
              const PwLaunchPlan& CL = next->pwl[0];
              h.red_w = c.pk(CL.wimg + CL.passes[0].w_off);
              h.part = reinterpret_cast<float*>(c.ws + next->logits_off);
            }
            h.wimg = c.pk(u.hz.wimg); h.ep_h = c.pk(u.hz.ep);
            h.CH = d.cin[0]; h.C1 = d.cin[1]; h.C2 = d.cin[2]; h.OH = d.cout[0];
            h.H1 = P.H >> (u.base_lvl + 1); h.W1 = P.W >> (u.base_lvl + 1); h.B = S;
            h.RB = u.hz.RB; h.ngroups = u.hz.ng; h.nth = u.hz.nth; h.hb = u.hz.hb; h.nw = u.hz.nw;
            if (csn_hz_layout(h) == 0) return CSN_E_INVALID;
            LAUNCH_TRY(csn_launch_hz(h, cl.stream));
            { const int ms_ = cl.mark("hz_kernel"); if (ms_ != CSN_OK) return ms_; }
            continue;
          }
          Pw4Args a;
          a.xh = xin[L.bh]; a.xl = xin[L.bl]; a.x2 = L.use_x2 ? xin[L.bl + 1] : nullptr; a.xq = L.bq >= 0 ? xin[L.bq] : nullptr;
          a.yh = L.hi_out >= 0 ? c.act_out(d.out_act[L.hi_out]) : nullptr;
          a.yl = L.lo_out >= 0 ? c.act_out(d.out_act[L.lo_out]) : nullptr;
          a.red_w = a.red_b = nullptr; a.logits = nullptr;
          if (cls_next) {   // cls_layer (csnet.py:306-308,381) rides in this unit's epilogue
            const PwLaunchPlan& CL = next->pwl[0];
            a.red_w = c.pk(CL.wimg + CL.passes[0].w_off);   // row 0 of its (zero padded) weight image
            a.red_b = c.pk(CL.passes[0].epi.shift);
            a.logits = reinterpret_cast<float*>(c.ws + next->logits_off);
          }
          a.wimg = c.pk(L.wimg);
          a.ep_h = L.ep[0] >= 0 ? c.pk(L.ep[0]) : nullptr;
          a.ep_l = L.ep[1] >= 0 ? c.pk(L.ep[1]) : nullptr;
          a.CH = d.cin[L.bh]; a.CL = d.cin[L.bl]; a.C2 = L.use_x2 ? d.cin[L.bl + 1] : 0; a.CQ = L.bq >= 0 ? d.cin[L.bq] : 0;
          a.OH = L.hi_out >= 0 ? d.cout[L.hi_out] : 0; a.OL = L.lo_out >= 0 ? d.cout[L.lo_out] : 0;
#ifdef CSN_KO_HEAD_NOLOW   // knock-out build (tools/README.md): the high-only three-branch launches without their low / third inputs
          if (L.use_x2 && L.lo_out < 0) { a.CL = 1; a.C2 = 1; }
#endif
          a.Hl = P.H >> (u.base_lvl + L.bl); a.Wl = P.W >> (u.base_lvl + L.bl); a.B = S;
          pw4_tile_geo(P, a.Hl, a.Wl, &a.twl, &a.tiles_x, &a.tiles_y);
          if (c.raw && c.a16 && c.pw4_stats) {   // statistics of the stored outputs from the epilogue (forward_train_body skips bn_stats)
            if (L.hi_out >= 0 && u.pstats_off[L.hi_out] >= 0) a.stats_h = reinterpret_cast<double*>(c.ws + u.pstats_off[L.hi_out]);
            if (L.lo_out >= 0 && u.pstats_off[L.lo_out] >= 0) a.stats_l = reinterpret_cast<double*>(c.ws + u.pstats_off[L.lo_out]);
            a.stats_stride = a.tiles_x * a.tiles_y * a.B;
          }
          a.ngroups = L.ng; a.gimg_floats = L.gimg; a.nth = L.nth; a.ntl = L.ntl;
          a.max_grid = P.pw4_grid; a.a16 = c.a16 ? 1 : 0;
          for (int g = 0; g < PW4_MAX_GROUPS; ++g) a.grp[g] = L.grp[g];
          LAUNCH_TRY(csn_launch_pw4(a, c.raw ? 1 : 0, cl.stream));
          { const int ms_ = cl.mark("pw4_kernel"); if (ms_ != CSN_OK) return ms_; }
        }
        if (!old.empty()) {
          PwBind bo;
          for (int i = 0; i < 3; ++i) bo.in[i] = xin[i];
          for (int j = 0; j < d.n_out; ++j)
            if (d.cout[j] > 0) bo.act[j] = c.act_out(d.out_act[j]);
          for (const PwLaunchPlan* L : old) {
            Ctx cl = c;
            if (fork && lane > 0) cl.stream = c.P.lane[lane - 1];
            ++lane;
            const int st = launch_pw(cl, *L, bo);
            if (st != CSN_OK) return st;
          }
        }
        if (fork) { const int st1 = lanes_join(c, nl - 1); if (st1 != CSN_OK) return st1; }
        if (cls_next) {
          Up2Args ua;
          ua.in = reinterpret_cast<const float*>(c.ws + next->logits_off); ua.out = c.y; ua.planes = S; ua.H = P.H; ua.W = P.W; ua.in16 = 0;
          if (use_hz) {   // hz_kernel left one partial plane per M group: the upsample adds them up and the cls_layer bias
            ua.nparts = u.hz.ng; ua.part_stride = (int64_t)S * (P.H >> 1) * (P.W >> 1);
            ua.bias = c.pk(next->pwl[0].passes[0].epi.shift);
          }
          LAUNCH_TRY(csn_launch_up2(ua, c.stream));
          { const int ms_ = c.mark("bilinear_up2_kernel"); if (ms_ != CSN_OK) return ms_; }
        }
        break;
      }
      PwBind bd;
      for (int i = 0; i < 3; ++i) bd.in[i] = xin[i];
      for (int j = 0; j < d.n_out; ++j)
        if (d.cout[j] > 0) bd.act[j] = c.act_out(d.out_act[j]);
      if (P.c3q && P.tiled3 && (!c.a16 || c.raw) && d.ksize == 3) {   // 2x2 max-pooled copies for the high -> low slices of c3q_kernel
        bool uses = false;
        for (const PwLaunchPlan& L : u.pwl) uses = uses || L.c3q;
        PoolArgs pa;
        pa.n = 0; pa.a16 = c.a16 ? 1 : 0; pa.pad = 0;
        int blk = 0;
        for (int i = 0; i < d.n_in && uses; ++i) {
          if (u.mp_off[i] < 0 || !xin[i]) continue;
          if (d.stride == 2 && u.pooled_by_producer && P.fuse_dw && !c.raw && u.mp_producer >= 0 &&
              dw_pair_writes_mp(P, P.units[u.mp_producer], i)) {   // the depthwise pair in front has written it
            bd.mp[i] = reinterpret_cast<float*>(c.ws + u.mp_off[i]);
            continue;
          }
          const int k = pa.n++;
          pa.in[k] = xin[i];
          pa.out[k] = reinterpret_cast<float*>(c.ws + u.mp_off[i]);
          bd.mp[i] = pa.out[k];
          pa.planes[k] = S * d.cin[i];
          pa.Ho[k] = P.H >> (u.base_lvl + i + 1); pa.Wo[k] = P.W >> (u.base_lvl + i + 1);
          const int64_t lanes = (int64_t)pa.planes[k] * pa.Ho[k] * ((pa.Wo[k] + 1) / 2);
          blk += (int)((lanes + CSN_BLOCK - 1) / CSN_BLOCK);
          pa.blk_end[k] = blk;
        }
        if (pa.n > 0) {
          LAUNCH_TRY(csn_launch_maxpool(pa, c.stream));
          { const int ms_ = c.mark("pool2_kernel"); if (ms_ != CSN_OK) return ms_; }
        }
      }
      if (u.z_off >= 0) bd.z = bd.zout = reinterpret_cast<float*>(c.ws + u.z_off);
      if (next && next->d.kind == CSN_UNIT_CLS) {   // cls_layer (csnet.py:306-308,381) rides in this unit's epilogue
        const PwLaunchPlan& CL = next->pwl[0];
        bd.red_w = c.pk(CL.wimg + CL.passes[0].w_off);
        bd.red_b = c.pk(CL.passes[0].epi.shift);
        bd.logits = reinterpret_cast<float*>(c.ws + next->logits_off);
      }
      // Launches of one unit that do not depend on each other run on parallel stream lanes (fork after everything
      // enqueued so far, join before the next unit): a 3x3 unit is {z -> high pass} || {low pass}, a unit split per output
      // branch (CSFHead.fuse) is one launch per lane.  Each of them alone leaves CUs idle in its ramp-up / tail.
      const int nl = (int)u.pwl.size();
      if (c.lanes && nl >= 2 && nl <= 3) {
        int lane_of[3] = {0, 0, 0};
        int next_lane = 1;
        for (int i = 0; i < nl; ++i) {
          const PwPassPlan& pp = u.pwl[i].passes[0];
          bool needs_z = false;
          for (int s2 = 0; s2 < pp.nsrc; ++s2) needs_z = needs_z || pp.src_kind[s2] == SRC_Z;
          if (pp.out_kind == OUT_Z || needs_z || u.pwl[i].passes.size() != 1) lane_of[i] = 0;   // the z chain stays in order
          else lane_of[i] = next_lane++;
        }
        // the last independent launch may as well use lane 0 when nothing else is there
        bool lane0_used = false;
        for (int i = 0; i < nl; ++i) lane0_used = lane0_used || lane_of[i] == 0;
        if (!lane0_used) { for (int i = 0; i < nl; ++i) --lane_of[i]; --next_lane; }
        if (next_lane > 1) {
          const int st0 = lanes_fork(c, next_lane - 1);
          if (st0 != CSN_OK) return st0;
          for (int i = 0; i < nl; ++i) {
            Ctx cl = c;
            if (lane_of[i] > 0) cl.stream = c.P.lane[lane_of[i] - 1];
            const int st = launch_pw(cl, u.pwl[i], bd);
            if (st != CSN_OK) return st;
          }
          const int st1 = lanes_join(c, next_lane - 1);
          if (st1 != CSN_OK) return st1;
        } else {
          for (const PwLaunchPlan& L : u.pwl) {
            const int st = launch_pw(c, L, bd);
            if (st != CSN_OK) return st;
          }
        }
      } else {
        for (const PwLaunchPlan& L : u.pwl) {
          const int st = launch_pw(c, L, bd);
          if (st != CSN_OK) return st;
        }
      }
      if (next && next->d.kind == CSN_UNIT_CLS) {
        Up2Args ua;
        ua.in = bd.logits; ua.out = c.y; ua.planes = S; ua.H = P.H; ua.W = P.W; ua.in16 = c.a16 ? 1 : 0;
        LAUNCH_TRY(csn_launch_up2(ua, c.stream));
        { const int ms_ = c.mark("bilinear_up2_kernel"); if (ms_ != CSN_OK) return ms_; }
      }
    } break;
    case CSN_UNIT_MS: {
      MsArgs a;
      const Act& act = P.acts[d.in_act[0]];
      a.in = c.act_in(d.in_act[0]); a.out = c.act_out(d.out_act[0]);
      int base = 0;
      for (int k = 0; k < 5; ++k) {
        a.dch[k] = d.dil_ch[k]; a.cobase[k] = base; base += d.dil_ch[k];
        a.w[k] = d.dil_ch[k] ? c.pk(u.ms_w[k]) : nullptr;
      }
      a.cin = d.cin[0]; a.cout = d.cout[0]; a.H = P.H >> act.lvl; a.W = P.W >> act.lvl; a.B = S;
      a.scale = c.sc(u.ms_epi); a.shift = c.sh(u.ms_epi); a.alpha = c.al(u.ms_epi); a.a16 = c.a16 ? 1 : 0; a.pad = 0;
      LAUNCH_TRY(csn_launch_ms(a, c.stream));
      { const int ms_ = c.mark("msblock_kernel"); if (ms_ != CSN_OK) return ms_; }
    } break;
    case CSN_UNIT_CLS: {
      PwBind bd;
      bd.in[0] = c.act_in(d.in_act[0]);
      bd.logits = reinterpret_cast<float*>(c.ws + u.logits_off);
      for (const PwLaunchPlan& L : u.pwl) {
        const int st = launch_pw(c, L, bd);
        if (st != CSN_OK) return st;
      }
      Up2Args ua;
      ua.in = reinterpret_cast<const float*>(c.ws + u.logits_off); ua.out = c.y; ua.planes = S; ua.H = P.H; ua.W = P.W;
      ua.in16 = c.a16 ? 1 : 0;
      LAUNCH_TRY(csn_launch_up2(ua, c.stream));
      { const int ms_ = c.mark("bilinear_up2_kernel"); if (ms_ != CSN_OK) return ms_; }
    } break;
    default:
      return CSN_E_INVALID;
  }
  return CSN_OK;
}

// units k, k + 1, k + 2 (1x1 gOctaveCBR + depthwise pair) as one launch of ilb_kernel (eval forward)
int run_ilb(const Ctx& c, int k) {
  const csn_plan& P = c.P;
  const UnitPlan& u = P.units[k];
  const UnitPlan& d1 = P.units[k + 1];
  const UnitPlan& d2 = P.units[k + 2];
  const csn_unit_desc& d = u.d;
  const UnitPlan::Ilb& I = u.ilb;
  IlbArgs a = {};
  a.k3 = I.k3;
  a.CH = d.cin[0]; a.CL = I.k3 ? d.cin[0] : d.cin[1]; a.OH = d.cout[0]; a.OL = d.n_out >= 2 ? d.cout[1] : 0;
  a.Hl = P.H >> (u.base_lvl + 1); a.Wl = P.W >> (u.base_lvl + 1); a.B = P.S;
  a.nth = I.nth; a.ntl = I.ntl; a.ng = I.ng; a.Rh = I.Rh; a.Rl = I.Rl;
  if (csn_ilb_layout(a) == 0 || a.gimg_floats != I.gimg) return CSN_E_STATE;
  if (I.k3) {   // the 2x2 averages of the block's input and their 2x2 maxima, written by the depthwise pair (or ilb launch) in front
    a.xh = reinterpret_cast<const float*>(c.ws + u.pooled_off[0]);
    a.xl = reinterpret_cast<const float*>(c.ws + u.mp_off[0]);
  } else {
    a.xh = c.act_in(d.in_act[0]); a.xl = c.act_in(d.in_act[1]);
  }
  a.yh = c.act_out(d2.d.out_act[0]);
  a.yl = a.OL > 0 ? c.act_out(d2.d.out_act[1]) : nullptr;
  a.wimg = c.pk(I.wimg);
  a.ep_h = c.pk(I.ep[0]);
  a.ep_l = I.ep[1] >= 0 ? c.pk(I.ep[1]) : a.ep_h;
  a.dwrec_h = c.pk(I.dwrec[0]);
  a.dwrec_l = I.dwrec[1] >= 0 ? c.pk(I.dwrec[1]) : a.dwrec_h;
  for (int j = 0; j < 2; ++j) {
    if (j == 1 && a.OL == 0) break;
    const int Wj = P.W >> (u.base_lvl + j);
    if (d1.pool_unit >= 0 && (Wj % 4) == 0) {   // the stride-2 unit that follows reads the 2x2 averages (+ their 2x2 maxima)
      float* pool = reinterpret_cast<float*>(c.ws + P.units[d1.pool_unit].pooled_off[j]);
      float* mp = dw_pair_writes_mp(P, d1, j) ? reinterpret_cast<float*>(c.ws + P.units[d1.pool_unit].mp_off[j]) : nullptr;
      if (j == 0) { a.pool_h = pool; a.mp_h = mp; a.skip_h = d1.pool_skip[j]; }
      else { a.pool_l = pool; a.mp_l = mp; a.skip_l = d1.pool_skip[j]; }
    }
  }
  LAUNCH_TRY(csn_launch_ilb(a, c.stream));
  return c.mark("ilb_kernel");
}

// ... which needs what the stride-2 consumer's planning expects of the pair in front of it
bool ilb_active(const csn_plan& P, int k) {
  if (!(P.ilb && P.fuse_dw && k >= 0 && k + 2 < (int)P.units.size() && P.units[k].ilb.on)) return false;
  const UnitPlan& u = P.units[k];   // the 3x3 entry form reads the max-pooled copy the pair in front writes (options may have changed)
  return !u.ilb.k3 || (u.mp_producer >= 0 && dw_pair_writes_mp(P, P.units[u.mp_producer], 0));
}

}  // namespace

static void drop_slot(csn_plan::GraphSlot& s) {
#ifndef CSN_CPU_EMU
  if (s.exec) { (void)hipGraphExecDestroy(s.exec); s.exec = nullptr; }
#endif
  s.key.clear();
  s.eager_calls = 0;
}

// Run `body(stream)` eagerly the first two times it is seen with `key`, capture it on the third and replay afterwards.
template <class F>
static int run_graphed(csn_plan* P, csn_plan::GraphSlot& slot, const std::vector<uint64_t>& key, void* stream, F body) {
#ifdef CSN_CPU_EMU
  (void)P; (void)slot; (void)key;
  return body(stream);
#else
  if (!P->use_graph) return body(stream);
  if (key != slot.key) {
    if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
    slot.key = key;
    slot.eager_calls = 0;
  }
  if (slot.exec) {
    HIP_TRY(hipGraphLaunch(slot.exec, (hipStream_t)stream));
    return CSN_OK;
  }
  if (slot.eager_calls++ < 2) return body(stream);   // warm-up: lazy initialisation (function attributes) done
  if (!P->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&P->cap_stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamBeginCapture(P->cap_stream, hipStreamCaptureModeThreadLocal));
  const int st = body(P->cap_stream);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(P->cap_stream, &g);
  if (st != CSN_OK || e != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    slot.eager_calls = -1000000;   // stay eager for this slot
    return body(stream);
  }
  const hipError_t e2 = hipGraphInstantiate(&slot.exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e2 != hipSuccess) { slot.exec = nullptr; slot.eager_calls = -1000000; return body(stream); }
  HIP_TRY(hipGraphLaunch(slot.exec, (hipStream_t)stream));
  return CSN_OK;
#endif
}

static uint64_t hash_floats(const float* p, int n) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < n; ++i) {
    uint32_t b;
    std::memcpy(&b, &p[i], 4);
    h = (h ^ b) * 1099511628211ull;
  }
  return h;
}

static void drop_graph(csn_plan* P) {
#ifndef CSN_CPU_EMU
  if (P->graph_exec) { (void)hipGraphExecDestroy(P->graph_exec); P->graph_exec = nullptr; }
#endif
  drop_slot(P->g_train);
  drop_slot(P->g_bwd);
  P->g_x = P->g_y = P->g_ws = nullptr;
  P->eager_calls = 0;
}

// auxiliary lanes exist (created lazily; never under the CPU emulation)
static bool lanes_ready(csn_plan* P) {
#ifdef CSN_CPU_EMU
  (void)P;
  return false;
#else
  if (!P->overlap) return false;
  if (!P->lane[0]) {
    for (int k = 0; k < 2; ++k)
      if (hipStreamCreateWithFlags(&P->lane[k], hipStreamNonBlocking) != hipSuccess) { P->overlap = false; return false; }
    for (int k = 0; k < 6; ++k)
      if (hipEventCreateWithFlags(&P->lane_ev[k], hipEventDisableTiming) != hipSuccess) { P->overlap = false; return false; }
  }
  return true;
#endif
}

#include "csn_backward.inl"   // backward planning + sequencing (shares the plan's private types)

#ifndef CSN_CPU_EMU
// device buffers of plans destroyed while a stream capture was in progress (csn_plan_destroy)
static std::mutex g_pending_mu;
static std::vector<void*> g_pending_free;
static void drain_pending_frees_locked() {
  for (void* p : g_pending_free) (void)hipFree(p);
  g_pending_free.clear();
}
#endif

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int csn_abi_version(void) { return CSN_ABI_VERSION; }

const char* csn_strerror(int s) {
  switch (s) {
    case CSN_OK: return "ok";
    case CSN_E_INVALID: return "invalid argument or inconsistent unit descriptor";
    case CSN_E_UNSUPPORTED: return "configuration not supported by this build";
    case CSN_E_HIP: return "HIP runtime error";
    case CSN_E_NOMEM: return "out of memory";
    case CSN_E_STATE: return "call order violated (refresh parameters before forward)";
    default: return "unknown status";
  }
}

const char* csn_last_hip_error(void) { return g_hip_err.c_str(); }

int csn_plan_create(const csn_unit_desc* units, int32_t n_units, const csn_act_desc* acts, int32_t n_acts,
                    int32_t B, int32_t H, int32_t W, int32_t sub_batch, csn_plan** out_plan) {
  if (!units || !acts || !out_plan || n_units <= 0 || n_acts <= 0 || B <= 0) return CSN_E_INVALID;
  if (H <= 0 || W <= 0 || (H % 16) != 0 || (W % 16) != 0) return CSN_E_INVALID;
  csn_plan* P = new (std::nothrow) csn_plan();
  if (!P) return CSN_E_NOMEM;
#ifndef CSN_CPU_EMU
  {   // buffers of plans that were destroyed inside a stream capture
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(nullptr, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
    if (capturing) (void)hipGetLastError();
    std::lock_guard<std::mutex> g(g_pending_mu);
    if (!capturing) drain_pending_frees_locked();
  }
#endif
  P->B = B; P->H = H; P->W = W;
  P->S = (sub_batch <= 0 || sub_batch > B) ? B : sub_batch;

  if (const char* e = std::getenv("CSN_HZ")) P->hz = std::atoi(e) != 0;
  {
    auto pair = [](const char* name, int (&dst)[2], int lo, int hi) {
      const char* e = std::getenv(name);
      if (!e) return;
      const int a0 = std::atoi(e);
      const char* c = std::strchr(e, '/');
      const int a1 = c ? std::atoi(c + 1) : a0;
      if (a0 >= lo && a0 <= hi) dst[0] = a0;
      if (a1 >= lo && a1 <= hi) dst[1] = a1;
    };
    pair("CSN_HZ_NT", P->hz_nt, 1, 5); pair("CSN_HZ_RB", P->hz_rb, 1, 14); pair("CSN_HZ_HB", P->hz_hb, 2, 4); pair("CSN_HZ_NW", P->hz_nw, 4, 16);
  }
  if (std::getenv("CSN_PW4_NOQ")) P->pw4_no_q = true;
  if (const char* e = std::getenv("CSN_PW4_FLAT")) P->pw4_flat = std::atoi(e) != 0;
  if (const char* e = std::getenv("CSN_POOL_ROUTE")) P->pool_route = std::atoi(e) != 0;
  if (const char* e = std::getenv("CSN_PW4_STATS")) P->pw4_stats = std::atoi(e) != 0;
  if (const char* e = std::getenv("CSN_ILB")) P->ilb = std::atoi(e) != 0;
  if (const char* e = std::getenv("CSN_DWB_FAST")) P->dwb_fast = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("CSN_DW_XL")) P->dw_xl = std::atoi(e) != 0;
  if (const char* e = std::getenv("CSN_ILB_NT")) P->ilb_nt = std::atoi(e) == 2 ? 2 : (std::atoi(e) == 1 ? 1 : 0);
  if (const char* e = std::getenv("CSN_ILB_MAXPIX")) { if (std::atoi(e) > 0) P->ilb_maxpix = std::atoi(e); }
  // (0 / 1.  Round 4's value 2 -- bf16 weights in the FORWARD 3x3 launches -- sat beyond the certified unit-local bound
  // (profiles/r4_notes.md) and went with its CSN_EXPERIMENTS gate in round 6)
  if (const char* e = std::getenv("CSN_C3Q16")) P->c3q16 = std::atoi(e) <= 0 ? 0 : 1;
  if (const char* e = std::getenv("CSN_PWQ16")) P->pwq16 = std::atoi(e) != 0;
  if (const char* e = std::getenv("CSN_MS_DX")) P->ms_dx = std::atoi(e) != 0;
  if (std::getenv("CSN_NO_MP_FUSE")) P->no_mp_fuse = true;
  if (const char* v = std::getenv("CSN_C3Q_HL")) P->c3q_hl = std::atoi(v) != 0;
  if (const char* v = std::getenv("CSN_C3Q_TWL")) { if (std::atoi(v) >= 2 && std::atoi(v) <= 6) P->c3q_twl = std::atoi(v); }
  if (const char* v = std::getenv("CSN_BN_BWD_FUSE")) P->bn_bwd_fuse = v[0] != '0';
  if (const char* v = std::getenv("CSN_DEBUG_DZ")) P->debug_dz = v[0] != '0';

  if (const char* v = std::getenv("CSN_PW4_TWL")) { if (std::atoi(v) >= 2 && std::atoi(v) <= 6) P->pw4_twl = std::atoi(v); }
  Builder bl(*P);
  P->acts.resize(n_acts);
  for (int i = 0; i < n_acts; ++i) {
    P->acts[i].channels = acts[i].channels;
    P->acts[i].lvl = acts[i].lvl;
    if (acts[i].channels <= 0 || acts[i].lvl < 0 || acts[i].lvl > 4) { delete P; return CSN_E_INVALID; }
    P->acts[i].ws_off = (i == 0) ? -1 : bl.alloc_act(acts[i].channels, acts[i].lvl);
  }
  P->units.resize(n_units);
  for (int k = 0; k < n_units; ++k) {
    UnitPlan& u = P->units[k];
    u.d = units[k];
    int st = CSN_E_INVALID;
    for (int i = 0; i < CSN_MAX_BRANCH; ++i) {
      if (u.d.in_act[i] >= n_acts || u.d.out_act[i] >= n_acts) { delete P; return CSN_E_INVALID; }
    }
    switch (u.d.kind) {
      case CSN_UNIT_GOCT: st = plan_goct(bl, u); break;
      case CSN_UNIT_DW: st = plan_dw(bl, u); break;
      case CSN_UNIT_MS: st = plan_ms(bl, u); break;
      case CSN_UNIT_CLS: st = plan_cls(bl, u); break;
      default: st = CSN_E_INVALID;
    }
    if (st != CSN_OK) {
      g_hip_err = "unit " + std::to_string(k) + " (kind " + std::to_string(u.d.kind) + "): " + g_why;
      delete P;
      return st;
    }
    // algorithmic bytes: unit inputs read once + outputs written once, whole batch
    int64_t bytes = 0;
    for (int i = 0; i < u.d.n_in; ++i)
      if (u.d.cin[i] > 0 && u.d.in_act[i] >= 0) {
        const Act& a = P->acts[u.d.in_act[i]];
        bytes += (int64_t)B * a.channels * (H >> a.lvl) * (W >> a.lvl) * 4;
      }
    if (u.d.kind == CSN_UNIT_CLS) {
      bytes += (int64_t)B * H * W * 4;
    } else {
      for (int j = 0; j < u.d.n_out; ++j)
        if (u.d.cout[j] > 0 && u.d.out_act[j] >= 0) {
          const Act& a = P->acts[u.d.out_act[j]];
          bytes += (int64_t)B * a.channels * (H >> a.lvl) * (W >> a.lvl) * 4;
        }
    }
    u.alg_bytes = bytes;
  }
  // fusable depthwise pairs: unit k+1 consumes exactly unit k's outputs, nobody else does, one tile in x
  for (int k = 0; k + 1 < n_units; ++k) {
    const csn_unit_desc& a = P->units[k].d;
    const csn_unit_desc& b = P->units[k + 1].d;
    if (a.kind != CSN_UNIT_DW || b.kind != CSN_UNIT_DW || a.n_in != b.n_in) continue;
    bool ok = true;
    for (int i = 0; i < a.n_in && ok; ++i) {
      if (a.cout[i] != b.cin[i]) ok = false;
      if (a.cout[i] == 0) continue;
      if (a.out_act[i] != b.in_act[i] || (W >> P->acts[a.out_act[i]].lvl) > 256) ok = false;
      for (int q = 0; q < n_units && ok; ++q) {
        if (q == k + 1) continue;
        for (int s = 0; s < CSN_MAX_BRANCH; ++s)
          if (P->units[q].d.in_act[s] == a.out_act[i] && s < P->units[q].d.n_in && P->units[q].d.cin[s] > 0) ok = false;
      }
    }
    if (ok) {
      P->units[k].fuse_next = 1;
      for (int i = 0; i < a.n_in; ++i) {   // {w' = 100 w gamma / sqrt(var + eps), shift, alpha} records of both units (dw_core.h)
        if (a.cout[i] == 0) continue;
        const int C = a.cout[i];
        const int64_t rec = bl.alloc_packed((int64_t)C * 24 + 8);
        P->units[k].dw2rec[i] = rec;
        for (int h = 0; h < 2; ++h) {
          const csn_unit_desc& w = h == 0 ? a : b;
          const int o = 12 * h;
          bl.job(CSN_PREP_DWREC, C, rec, w.w_off[i], w.bn[i].weight, w.bn[i].running_var, -1, 100.0f, 0, 0, 24, o);   // conv2d.py:104
          bl.job(CSN_PREP_BN_SHIFT, C, rec, w.bn[i].weight, w.bn[i].running_var, w.bn[i].bias, w.bn[i].running_mean, 1.f, 0, 0, 24, o + 9);
          bl.job(CSN_PREP_COPY, C, rec, w.bn[i].prelu, -1, -1, -1, 1.f, 0, 0, 24, o + 10);
        }
      }
      ++k;
    }
  }
  // pooled outputs: a fused depthwise pair directly followed by the stride-2 unit that consumes all its branches
  for (int k = 0; k + 2 < n_units; ++k) {
    if (!P->units[k].fuse_next) continue;
    const csn_unit_desc& b = P->units[k + 1].d;
    const csn_unit_desc& g = P->units[k + 2].d;
    if (g.kind != CSN_UNIT_GOCT || g.stride != 2 || g.n_in != b.n_out || (g.n_in == 1 && g.n_out == 1)) continue;
    bool ok = true;
    for (int i = 0; i < g.n_in && ok; ++i) {
      if (g.cin[i] != b.cout[i] || (g.cin[i] > 0 && g.in_act[i] != b.out_act[i])) ok = false;
      if (g.cin[i] > 0 && ((W >> P->acts[b.out_act[i]].lvl) % 4) != 0) ok = false;
    }
    if (!ok) continue;
    P->units[k].pool_unit = k + 2;
    P->units[k + 2].pooled_by_producer = 1;
    P->units[k + 2].mp_producer = k;
    for (int i = 0; i < g.n_in; ++i) {
      if (g.cin[i] == 0) continue;
      bool only = true;
      for (int q = 0; q < n_units && only; ++q) {
        if (q == k + 2) continue;
        for (int s = 0; s < CSN_MAX_BRANCH; ++s)
          if (s < P->units[q].d.n_in && P->units[q].d.cin[s] > 0 && P->units[q].d.in_act[s] == b.out_act[i]) only = false;
      }
      P->units[k].pool_skip[i] = only ? 1 : 0;
    }
  }
  for (int k = 0; k + 2 < n_units; ++k) {   // whole ILBlocks of the small maps on ilb_kernel
    const int st = plan_ilb(bl, k);
    if (st != CSN_OK) { delete P; return st; }
  }
  // cls fusion: a single-output, single-launch 1x1 unit whose only reader is the cls_layer that follows it
  for (int k = 0; k + 1 < n_units; ++k) {
    const csn_unit_desc& a = P->units[k].d;
    const csn_unit_desc& b = P->units[k + 1].d;
    if (a.kind != CSN_UNIT_GOCT || b.kind != CSN_UNIT_CLS || a.n_out != 1 || a.ksize != 1) continue;
    if (b.in_act[0] != a.out_act[0] || P->units[k].pwl.size() != 1 || P->units[k].pwl[0].passes.size() != 1) continue;
    bool ok = true;
    for (int q = 0; q < n_units && ok; ++q) {
      if (q == k + 1) continue;
      for (int s = 0; s < CSN_MAX_BRANCH; ++s)
        if (s < P->units[q].d.n_in && P->units[q].d.cin[s] > 0 && P->units[q].d.in_act[s] == a.out_act[0]) ok = false;
    }
    if (ok) P->units[k].fuse_cls = 1;
  }
  {  // identity epilogue + per-BN statistics partials for the train-mode forward
    int maxc = 1;
    for (int i = 0; i < n_acts; ++i) maxc = std::max(maxc, (int)acts[i].channels);
    P->ident.scale = bl.alloc_packed(maxc); P->ident.shift = bl.alloc_packed(maxc); P->ident.alpha = bl.alloc_packed(maxc);
    P->ident.dummy = bl.alloc_packed(maxc);
    bl.job(CSN_PREP_FILL, maxc, P->ident.scale, -1, -1, -1, -1, 1.f);
    bl.job(CSN_PREP_FILL, maxc, P->ident.shift, -1, -1, -1, -1, 0.f);
    bl.job(CSN_PREP_FILL, maxc, P->ident.alpha, -1, -1, -1, -1, 1.f);
    P->pen_slots = (int)P->units.size() * CSN_MAX_BRANCH;
    P->pen_off = bl.alloc_ws((int64_t)P->pen_slots * sizeof(double));   // per-job penalty terms of a train-mode forward
    for (auto& u : P->units)
      if (u.d.kind != CSN_UNIT_CLS)
        for (int j = 0; j < u.d.n_out; ++j)
          if (u.d.cout[j] > 0) {
            u.stats_off[j] = bl.alloc_ws((int64_t)u.d.cout[j] * CSN_BN_NSLAB * 2 * sizeof(double));
            u.gap_off[j] = bl.alloc_ws((int64_t)u.d.cout[j] * P->S * sizeof(float));   // per-image |GAP| (penalty)
          }
  }
  if (csn_kernels_init() != 0) { delete P; return CSN_E_HIP; }
#ifndef CSN_CPU_EMU
  (void)hipGetDevice(&P->device);
#endif
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&P->packed), (size_t)(P->packed_floats + 4) * sizeof(float));
  if (e != hipSuccess) { delete P; hip_fail(e, "hipMalloc(packed)"); return CSN_E_NOMEM; }
  e = hipMemsetAsync(P->packed, 0, (size_t)(P->packed_floats + 4) * sizeof(float), nullptr);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&P->jobs_dev), P->jobs.size() * sizeof(CsnPrepJob));
  if (e == hipSuccess)
    e = hipMemcpy(P->jobs_dev, P->jobs.data(), P->jobs.size() * sizeof(CsnPrepJob), hipMemcpyHostToDevice);
  if (e != hipSuccess) { hip_fail(e, "plan upload"); csn_plan_destroy(P); return CSN_E_HIP; }
  *out_plan = P;
  return CSN_OK;
}

void csn_plan_destroy(csn_plan* P) {
  if (!P) return;
#ifndef CSN_CPU_EMU
  // callers enqueue asynchronously: the plan's last forward may still be running on the caller's stream and on the lanes when an
  // LRU eviction (sod100k_amd/model/csnet.py engine_for) drops it -- graphs, lane streams, events and the packed weights must
  // outlive that work.  Not legal (and not needed: nothing of this plan can be in flight) while a stream is capturing.
  // (1) wait on the plan's OWN device, not on whatever device is current (a multi-GPU process evicting a plan of
  // another GPU); (2) never from inside a stream capture of this thread: hipDeviceSynchronize() would invalidate the user's
  // capture, and nothing of this plan can be in flight on a capturing stream that has not been launched yet.
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != P->device) (void)hipSetDevice(P->device);
  hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(nullptr, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
  if (capturing) (void)hipGetLastError();   // (the legacy stream cannot be queried during a global-mode capture: that IS the answer)
  if (!capturing) {
    (void)hipDeviceSynchronize();
  } else {
    // the plan's own streams are not the capturing one -- work enqueued on them BEFORE the capture began may still run
    for (int k = 0; k < 2; ++k) if (P->lane[k]) (void)hipStreamSynchronize(P->lane[k]);
    if (P->cap_stream) (void)hipStreamSynchronize(P->cap_stream);
    (void)hipGetLastError();
  }
  if (cur != P->device) (void)hipSetDevice(cur);
#endif
  for (hipEvent_t ev : P->ev) (void)hipEventDestroy(ev);
#ifndef CSN_CPU_EMU
  if (P->graph_exec) (void)hipGraphExecDestroy(P->graph_exec);
  if (P->cap_stream) (void)hipStreamDestroy(P->cap_stream);
  for (int k = 0; k < 2; ++k) if (P->lane[k]) (void)hipStreamDestroy(P->lane[k]);
  for (int k = 0; k < 6; ++k) if (P->lane_ev[k]) (void)hipEventDestroy(P->lane_ev[k]);
  // hipFree is not legal during a global-mode capture (it would fail or invalidate the capture): the buffers wait on a list that
  // the next call outside a capture drains (csn_plan_create / csn_plan_destroy)
  {
    std::lock_guard<std::mutex> g(g_pending_mu);
    if (P->packed) g_pending_free.push_back(P->packed);
    if (P->jobs_dev) g_pending_free.push_back(P->jobs_dev);
    if (!capturing) drain_pending_frees_locked();
  }
#else
  if (P->packed) (void)hipFree(P->packed);
  if (P->jobs_dev) (void)hipFree(P->jobs_dev);
#endif
  delete P;
}

int csn_plan_set_option(csn_plan* P, int32_t option, int32_t value) {
  if (!P) return CSN_E_INVALID;
  switch (option) {
    case CSN_OPT_FUSE_DW: P->fuse_dw = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_GRAPH: P->use_graph = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_FUSE_CLS: P->fuse_cls = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_TILED3: P->tiled3 = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_PW4: P->pw4 = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_C3Q: P->c3q = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_SLICE_LANES: P->slice_lanes = value != 0; drop_graph(P); return CSN_OK;
    case CSN_OPT_INPUT_GRAD:
      if (P->train) { g_hip_err = "CSN_OPT_INPUT_GRAD must be set before csn_plan_enable_training (it adds a gradient buffer)"; return CSN_E_STATE; }
      P->input_grad = value != 0; return CSN_OK;
    case CSN_OPT_FUSE_ILB: P->ilb = value != 0; drop_graph(P); return CSN_OK;   // (round 5: ilb_kernel, k_ilb.hip)
    case CSN_OPT_OVERLAP: P->overlap = value != 0; P->overlap_bwd = value == 2; drop_graph(P); return CSN_OK;
    case CSN_OPT_TRAIN_BF16:
      if (P->act_half && value == 0) { g_hip_err = "the workspace of this plan is laid out for bfloat16 tensors"; return CSN_E_STATE; }
      P->act16 = value != 0; drop_graph(P); return CSN_OK;
    default: return CSN_E_INVALID;
  }
}

size_t csn_plan_workspace_bytes(const csn_plan* P) {
  if (!P) return 0;
  const int nslices = (P->B + P->S - 1) / P->S;
  P->ws_regions_reported = (P->slice_lanes && nslices > 1) ? std::min(nslices, 3) : 1;
  return (size_t)P->ws_bytes * (size_t)P->ws_regions_reported;
}
int32_t csn_plan_num_units(const csn_plan* P) { return P ? (int32_t)P->units.size() : 0; }

int csn_plan_act_info(const csn_plan* P, int32_t id, csn_act_info* out) {
  if (!P || !out || id < 0 || id >= (int)P->acts.size()) return CSN_E_INVALID;
  const Act& a = P->acts[id];
  out->ws_offset_bytes = a.ws_off;
  out->channels = a.channels;
  out->height = P->H >> a.lvl;
  out->width = P->W >> a.lvl;
  out->batch = P->S;
  return CSN_OK;
}

int csn_plan_refresh_params(csn_plan* P, const float* arena, int64_t arena_floats, void* stream) {
  if (!P || !arena) return CSN_E_INVALID;
  for (const CsnPrepJob& j : P->jobs) {
    const int64_t srcs[4] = {j.src0, j.src1, j.src2, j.src3};
    for (int64_t s : srcs)
      if (s >= arena_floats) return CSN_E_INVALID;
  }
  LAUNCH_TRY(csn_launch_prep(P->jobs_dev, (int)P->jobs.size(), arena, P->packed, stream));
  P->params_ready = true;
  P->bn_tables_train = false;
  return CSN_OK;
}

static int forward_body(csn_plan* P, const float* x, float* y, void* workspace, void* stream, int32_t iters,
                        float* unit_ms) {
  if (!P || !x || !y || !workspace) return CSN_E_INVALID;
  if (!P->params_ready || P->bn_tables_train) return CSN_E_STATE;   // train forward rewrote the BN tables
  if (P->act_half) { g_hip_err = "plan laid out for the bfloat16 train step: no fp32 eval forward"; return CSN_E_STATE; }
  const int nu = (int)P->units.size();
  const bool prof = unit_ms != nullptr;
  if (prof) {
    for (int u = 0; u < nu; ++u) unit_ms[u] = 0.f;
    P->kstats.clear();
  }
  const int64_t in_stride = (int64_t)P->acts[0].channels * P->H * P->W, out_stride = (int64_t)P->H * P->W;
  const int reps = prof ? (iters > 0 ? iters : 1) : 1;
#ifndef CSN_CPU_EMU
  if (prof) {
    // An interval between two events around a launch = the kernel + the dispatch / event latency of the pair.  The pair's own
    // cost is measured on empty launches (lower quartile of 48) and subtracted, so that the per-kernel times agree with the
    // durations a kernel trace (rocprofv3) reports for the same launches.  A trace gives the empty launch itself 3.6 us
    // (profiles/r3_kernel_stats_eval*.md, csn_nop_kernel row: begin-to-end of a dispatch that does nothing), and every real
    // launch carries the same begin-to-end overhead inside its traced duration -- so that part stays in.
    const int NB = 49;
    while (P->ev.size() < (size_t)NB) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); P->ev.push_back(e); }
    HIP_TRY(hipEventRecord(P->ev[0], (hipStream_t)stream));
    for (int i = 1; i < NB; ++i) {
      LAUNCH_TRY(csn_launch_nop(stream));
      HIP_TRY(hipEventRecord(P->ev[i], (hipStream_t)stream));
    }
    HIP_TRY(hipEventSynchronize(P->ev[NB - 1]));
    std::vector<float> br;
    for (int i = 1; i < NB; ++i) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, P->ev[i - 1], P->ev[i])); br.push_back(ms); }
    std::sort(br.begin(), br.end());
    // the lower quartile of 48: scheduling noise (and a profiler that intercepts the launches) only ever adds to an interval
    P->bracket_ms = std::max(br[br.size() / 4] - 3.6e-3f, 0.f);
  }
#endif
  // concurrent slices: slice i runs on stream lane i % nconc in workspace region i % nconc (the small maps of the deep
  // stages are latency-bound per launch: two half-batches side by side fill the gaps a single chain of launches leaves)
  const int nslices = (P->B + P->S - 1) / P->S;
  int nconc = 1;
  if (P->slice_lanes && nslices > 1 && !prof && lanes_ready(P)) nconc = std::min(nslices, 3);
  // the concurrent slices live in workspace + lane * ws_bytes: the caller's buffer was sized by csn_plan_workspace_bytes(), which
  // must have seen the option (set afterwards, the extra regions would lie beyond the buffer -- ADVICE r3)
  if (nconc > 1 && P->ws_regions_reported > 0 && nconc > P->ws_regions_reported) {
    g_hip_err = "CSN_OPT_SLICE_LANES was set after csn_plan_workspace_bytes(): query the workspace size again";
    return CSN_E_STATE;
  }
  if (nconc > 1) {
    Ctx c0{*P, x, y, static_cast<char*>(workspace), stream};
    const int st = lanes_fork(c0, nconc - 1);
    if (st != CSN_OK) return st;
  }
  for (int it = 0; it < reps; ++it) {
    for (int b0 = 0, si = 0; b0 < P->B; b0 += P->S, ++si) {
      // the last slice may overlap the previous one when S does not divide B (same results, written twice)
      const int start = (b0 + P->S <= P->B) ? b0 : P->B - P->S;
      const int ln = si % nconc;
      Ctx c{*P, x + start * in_stride, y + start * out_stride, static_cast<char*>(workspace) + (int64_t)ln * P->ws_bytes,
            ln == 0 ? stream : (void*)P->lane[ln - 1]};
      std::vector<int> unit_of_tag;
      if (prof) {
        P->profiling = true; P->ev_used = 0; P->tags.clear();
        const int st0 = c.mark("start");
        if (st0 != CSN_OK) { P->profiling = false; return st0; }
      }
      c.lanes = lanes_ready(P) && !prof && nconc == 1;
      for (int u = 0; u < nu; ++u) {
        // consecutive MSBlocks (one per CSFHead branch, csnet.py:92-113) read different tensors: one lane each
        if (c.lanes && P->units[u].d.kind == CSN_UNIT_MS) {
          int m = 1;
          while (u + m < nu && m < 3 && P->units[u + m].d.kind == CSN_UNIT_MS) ++m;
          if (m > 1) {
            int st = lanes_fork(c, m - 1);
            for (int k = 0; k < m && st == CSN_OK; ++k) {
              Ctx cl = c;
              if (k > 0) cl.stream = P->lane[k - 1];
              st = run_unit(cl, P->units[u + k], nullptr);
            }
            if (st == CSN_OK) st = lanes_join(c, m - 1);
            if (st != CSN_OK) return st;
            u += m - 1;
            continue;
          }
        }
        if (ilb_active(*P, u)) {   // the whole ILBlock (1x1 unit + depthwise pair) as one launch
          const int st = run_ilb(c, u);
          if (st != CSN_OK) { P->profiling = false; return st; }
          if (prof) unit_of_tag.resize(P->tags.size(), u);
          u += 2;
          continue;
        }
        const bool fuse = u + 1 < nu && ((P->fuse_dw && P->units[u].fuse_next) || (P->fuse_cls && P->units[u].fuse_cls));
        const size_t t0 = P->tags.size();
        const int st = run_unit(c, P->units[u], fuse ? &P->units[u + 1] : nullptr);
        if (st != CSN_OK) { P->profiling = false; return st; }
        if (prof) unit_of_tag.resize(P->tags.size(), u), (void)t0;
        if (fuse) ++u;   // the pair ran as one kernel: the second unit takes no time of its own
      }
      if (prof) {
        P->profiling = false;
        HIP_TRY(hipEventSynchronize(P->ev[P->ev_used - 1]));
        for (size_t i = 1; i < P->ev_used; ++i) {
          float ms = 0.f;
          HIP_TRY(hipEventElapsedTime(&ms, P->ev[i - 1], P->ev[i]));
          ms = std::max(ms - P->bracket_ms, 0.25f * ms);      // the event pair's own cost (see above)
          unit_ms[unit_of_tag[i]] += ms;
          bool found = false;
          for (auto& k : P->kstats)
            if (k.name == P->tags[i]) { k.ms += ms; k.launches += 1; found = true; break; }
          if (!found) P->kstats.push_back(csn_plan::KStat{P->tags[i], ms, 1});
        }
      }
    }
  }
  if (nconc > 1) {
    Ctx c0{*P, x, y, static_cast<char*>(workspace), stream};
    const int st = lanes_join(c0, nconc - 1);
    if (st != CSN_OK) return st;
  }
  if (prof) {
    for (int u = 0; u < nu; ++u) unit_ms[u] /= (float)reps;
    for (auto& k : P->kstats) { k.ms /= reps; k.launches /= reps; }
  }
  return CSN_OK;
}

int csn_forward(csn_plan* P, const float* x, float* y, void* workspace, void* stream) {
#ifdef CSN_CPU_EMU
  return forward_body(P, x, y, workspace, stream, 1, nullptr);
#else
  if (!P || !P->use_graph) return forward_body(P, x, y, workspace, stream, 1, nullptr);
  if (x != P->g_x || y != P->g_y || workspace != P->g_ws) {
    drop_graph(P);
    P->g_x = x; P->g_y = y; P->g_ws = workspace;
  }
  if (P->graph_exec) {
    HIP_TRY(hipGraphLaunch(P->graph_exec, (hipStream_t)stream));
    return CSN_OK;
  }
  if (P->eager_calls++ < 1) return forward_body(P, x, y, workspace, stream, 1, nullptr);  // warm-up: lazy init done
  if (!P->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&P->cap_stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamBeginCapture(P->cap_stream, hipStreamCaptureModeThreadLocal));
  const int st = forward_body(P, x, y, workspace, P->cap_stream, 1, nullptr);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(P->cap_stream, &g);
  if (st != CSN_OK || e != hipSuccess || !g) {     // capture failed: stay eager
    if (g) (void)hipGraphDestroy(g);
    P->use_graph = false;
    return forward_body(P, x, y, workspace, stream, 1, nullptr);
  }
  const hipError_t e2 = hipGraphInstantiate(&P->graph_exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e2 != hipSuccess) { P->graph_exec = nullptr; P->use_graph = false; return forward_body(P, x, y, workspace, stream, 1, nullptr); }
  HIP_TRY(hipGraphLaunch(P->graph_exec, (hipStream_t)stream));
  return CSN_OK;
#endif
}

int csn_forward_profile(csn_plan* P, const float* x, float* y, void* workspace, void* stream, int32_t iters,
                        float* unit_ms) {
  if (!unit_ms) return CSN_E_INVALID;
  return forward_body(P, x, y, workspace, stream, iters, unit_ms);
}

static int forward_train_body(csn_plan* P, const float* x, float* y, void* workspace, float* arena,
                              const float* flop_w, double* penalty, void* stream);

int csn_forward_train(csn_plan* P, const float* x, float* y, void* workspace, float* arena, int64_t arena_floats,
                      const float* flop_w, double* penalty, void* stream) {
  if (!P || !x || !y || !workspace || !arena || !flop_w || !penalty) return CSN_E_INVALID;
  if (!P->params_ready) return CSN_E_STATE;
  (void)arena_floats;
  P->bn_tables_train = true;   // host-side state: set here, a graph replay does not run the body
  const std::vector<uint64_t> key = {(uint64_t)(uintptr_t)x, (uint64_t)(uintptr_t)y, (uint64_t)(uintptr_t)workspace,
                                     (uint64_t)(uintptr_t)arena, (uint64_t)(uintptr_t)penalty,
                                     hash_floats(flop_w, (int)P->units.size() * CSN_MAX_BRANCH)};
  return run_graphed(P, P->g_train, key, stream, [&](void* s) {
    return forward_train_body(P, x, y, workspace, arena, flop_w, penalty, s);
  });
}

static int forward_train_body(csn_plan* P, const float* x, float* y, void* workspace, float* arena,
                              const float* flop_w, double* penalty, void* stream) {
  P->bn_tables_train = true;
  if (P->S != P->B) { g_hip_err = "train mode needs the whole batch in one slice (sub_batch = 0)"; return CSN_E_UNSUPPORTED; }
  const int nu = (int)P->units.size();
  Ctx c{*P, x, y, static_cast<char*>(workspace), stream};
  std::vector<BnPenaltyJob> pen_jobs;
  if (P->act16) {   // bf16 activations: needs the training buffers (the bf16 copy of x lives there)
    if (!P->train) { g_hip_err = "CSN_OPT_TRAIN_BF16 needs csn_plan_enable_training"; return CSN_E_STATE; }
    c.a16 = true;
    const int64_t nx = (int64_t)P->S * P->acts[0].channels * P->H * P->W;   // H, W are multiples of 16
    LAUNCH_TRY(csn_launch_to_bf16(x, c.ws + P->x16_off, nx, stream));
  }
  c.dw_stats = true;
  std::vector<GapTilesArgs> gap_jobs;    // (read by the penalty kernel below and by csn_backward only)
  c.gap_defer = &gap_jobs;
  for (int u = 0; u < nu; ++u) {
    const UnitPlan& up = P->units[u];
    const csn_unit_desc& d = up.d;
    c.raw = d.kind != CSN_UNIT_CLS;
    const bool next_cls = u + 1 < nu && P->units[u + 1].d.kind == CSN_UNIT_CLS;   // (run_unit keeps that unit off pw4_kernel in train mode)
    const bool pw4_st = c.a16 && !next_cls && pw4_unit_stats(*P, up);
    c.pw4_stats = pw4_st;
    const bool stats_done = pw4_st || dw_unit_stats(*P, up);   // the convolution kernel writes the statistics partials itself
    const int st = run_unit(c, up, nullptr);           // raw z of every output branch (no depthwise fusion)
    if (st != CSN_OK) return st;
    if (d.kind == CSN_UNIT_CLS) continue;
    // statistics of every output branch, then ONE finalise launch for the unit, then the apply passes
    BnFinalizeArgs fas[CSN_MAX_BRANCH];
    int fa_of[CSN_MAX_BRANCH] = {-1, -1, -1}, nfa = 0;
    for (int j = 0; j < d.n_out; ++j) {
      if (d.cout[j] == 0) continue;
      const Act& act = P->acts[d.out_act[j]];
      const int64_t hw = (int64_t)(P->H >> act.lvl) * (P->W >> act.lvl);
      float* z = c.act_out(d.out_act[j]);      // raw conv output (its own buffer when training is enabled)
      double* part = reinterpret_cast<double*>(c.ws + up.stats_off[j]);
      BnStatsArgs sa; sa.z = z; sa.partial = part; sa.S = P->S; sa.C = d.cout[j]; sa.HW = hw; sa.a16 = c.a16 ? 1 : 0;
      if (!stats_done) LAUNCH_TRY(csn_launch_bn_stats(sa, stream));
      BnFinalizeArgs& fa = fas[nfa]; fa.partial = part; fa.arena = arena;
      fa.nslab = pw4_st ? up.pstats_n[j] : stats_done ? dw_stats_slabs(*P, act.lvl) : 0;
      if (pw4_st) { fa.partial = reinterpret_cast<double*>(c.ws + up.pstats_off[j]); fa.pstride = up.pstats_n[j]; }
      fa.scale = P->packed + up.out_epi[j].scale; fa.shift = P->packed + up.out_epi[j].shift;
      fa.off_weight = d.bn[j].weight; fa.off_bias = d.bn[j].bias; fa.off_rmean = d.bn[j].running_mean;
      fa.off_rvar = d.bn[j].running_var; fa.count = (int64_t)P->S * hw; fa.C = d.cout[j]; fa.S = P->S;
      // backward needs the batch mean / invstd; without training buffers they land in a dummy slot of the tables
      fa.mean = P->packed + (P->train ? up.tr_mean[j] : P->ident.dummy);
      fa.invstd = P->packed + (P->train ? up.tr_invstd[j] : P->ident.dummy);
      fa_of[j] = nfa++;
    }
    LAUNCH_TRY(csn_launch_bn_finalize_n(fas, nfa, stream));
    for (int j = 0; j < d.n_out; ++j) {
      if (d.cout[j] == 0) continue;
      const Act& act = P->acts[d.out_act[j]];
      const int64_t hw = (int64_t)(P->H >> act.lvl) * (P->W >> act.lvl);
      float* z = c.act_out(d.out_act[j]);
      const BnFinalizeArgs& fa = fas[fa_of[j]];
      BnApplyArgs aa; aa.z = z; aa.y = c.act_y(d.out_act[j]);
      aa.gapabs = reinterpret_cast<float*>(c.ws + up.gap_off[j]);
      aa.scale = fa.scale; aa.shift = fa.shift; aa.alpha = P->packed + up.out_epi[j].alpha;
      aa.arena = arena; aa.penalty = penalty; aa.off_weight = d.bn[j].weight; aa.HW = hw; aa.S = P->S; aa.C = d.cout[j];
      aa.flop_w = flop_w[u * CSN_MAX_BRANCH + j]; aa.a16 = c.a16 ? 1 : 0;
      const bool virt = P->train && !P->virt_cons.empty() && P->virt_cons[d.out_act[j]] >= 0;
      if (!virt) {
        LAUNCH_TRY(csn_launch_bn_apply(aa, stream));
      } else if (P->debug_dz) {   // probes only: y is materialised, the |GAP| table stays the consumer's
        BnApplyArgs dbg = aa;
        dbg.flop_w = 0.f;
        LAUNCH_TRY(csn_launch_bn_apply(dbg, stream));
      }
      if (aa.flop_w != 0.f) {
        BnPenaltyJob pj;
        pj.gapabs = aa.gapabs; pj.off_weight = aa.off_weight; pj.C = aa.C; pj.S = aa.S; pj.flop_w = aa.flop_w; pj.pad = 0;
        pen_jobs.push_back(pj);
      }
    }
  }
  if (!gap_jobs.empty()) LAUNCH_TRY(csn_launch_gap_tiles_batch(gap_jobs.data(), (int)gap_jobs.size(), stream));
  // the Oct_bn_hook penalty of all hooked sub-modules: two launches at the end of the forward (partials live in the first
  // unit's statistics slab, which the backward pass does not read)
  if (!pen_jobs.empty()) {
    if ((int64_t)pen_jobs.size() > (int64_t)P->pen_slots) return CSN_E_UNSUPPORTED;
    LAUNCH_TRY(csn_launch_bn_penalty(pen_jobs.data(), (int)pen_jobs.size(), arena,
                                     reinterpret_cast<double*>(c.ws + P->pen_off), penalty, stream));
  }
  return CSN_OK;
}

const char* csn_unit_kernel_name(const csn_plan* P, int32_t u) {
  if (!P || u < 0 || u >= (int)P->units.size()) return "";
  if (ilb_active(*P, u) || (u >= 1 && ilb_active(*P, u - 1)) || (u >= 2 && ilb_active(*P, u - 2))) return "ilb_kernel";
  if (P->fuse_dw && P->units[u].d.kind == CSN_UNIT_DW) {
    if (P->units[u].fuse_next || (u > 0 && P->units[u - 1].fuse_next)) {
      const UnitPlan& first = P->units[u].fuse_next ? P->units[u] : P->units[u - 1];
      return dw_pair_fast(*P, first) ? "dw3x3x2_fast_kernel" : "dw3x3x2_bn_prelu_kernel";
    }
  }
  if (P->tiled3 && P->units[u].c3) {
    bool q = P->c3q;
    for (const PwLaunchPlan& L : P->units[u].pwl) q = q && L.c3q;
    return q ? "c3q_kernel" : "goct_c3_kernel";
  }
  if (P->pw4 && P->hz && P->units[u].pw4 && P->units[u].hz.on &&
      !(P->fuse_cls && P->units[u].fuse_cls && !(P->units[u].pw4l.size() == 1 && P->units[u].pw4l[0].lo_out < 0)))
    return "hz_kernel";
  if (P->pw4 && P->units[u].pw4 &&
      !(P->fuse_cls && P->units[u].fuse_cls && !(P->units[u].pw4l.size() == 1 && P->units[u].pw4l[0].lo_out < 0)))
    return "pw4_kernel";
  return P->units[u].kname;
}

int32_t csn_profile_num_kernels(const csn_plan* P) { return P ? (int32_t)P->kstats.size() : 0; }
double csn_profile_bracket_us(const csn_plan* P) { return P ? 1e3 * (double)P->bracket_ms : 0.0; }

int csn_profile_kernel(const csn_plan* P, int32_t i, const char** name, double* ms_per_forward, int32_t* launches) {
  if (!P || i < 0 || i >= (int)P->kstats.size() || !name || !ms_per_forward || !launches) return CSN_E_INVALID;
  *name = P->kstats[i].name; *ms_per_forward = P->kstats[i].ms; *launches = P->kstats[i].launches;
  return CSN_OK;
}

int64_t csn_unit_algorithmic_bytes(const csn_plan* P, int32_t u) {
  if (!P || u < 0 || u >= (int)P->units.size()) return 0;
  return P->units[u].alg_bytes;
}

}  // extern "C"

#include "csf_head.inl"      // CSF+Res2Net decoder head (include/csf_hip.h)
