/*
 * csnet_hip.h -- C ABI of the MI355X-native CSNet engine (libcsnet_hip.so).
 *
 * This is the drop-in boundary for the reference's model seam
 *     model_lib = importlib.import_module("model." + cfg.MODEL.ARCH)      (CSNet/test.py:37-39,
 *     model = model_lib.build_model(...); predict = model(input_var)       CSNet_training/train.py:70-80,203)
 * i.e. everything `CSNet.forward` (CSNet/model/csnet.py:365-387) and its autograd backward do on the
 * device.  The reference itself has no FFI (pure Python over torch/ATen); the host mirror
 * `sod100k_amd/model/csnet.py` keeps the reference's nn.Module parameter tree and binds these entry
 * points through ctypes.
 *
 * Conventions
 *  - plain C types only; every function returns an int status (0 = CSN_OK) and never throws;
 *  - the library owns the plan (and the packed-parameter buffer inside it); the CALLER owns the input,
 *    output, parameter arena and workspace device buffers (e.g. torch tensors -> data_ptr());
 *  - all launches go to the caller's stream (`void* stream` is a hipStream_t); no hidden
 *    synchronisation and no allocation after csn_plan_create();
 *  - a plan is not re-entrant: one plan per (device, stream); data parallel = one process per GPU;
 *  - external tensor layout is the reference's: contiguous NCHW float32.  Internally every activation
 *    is also planar [B][C][H][W] per resolution branch (channel is wave-uniform -> weights, BN and
 *    PReLU parameters live in SGPRs; no channel padding traffic).
 *
 * A network is described as a list of *units*; a unit is one of the reference's sub-modules that owns
 * an activation boundary in HBM (SURVEY.md section 8(d) "unit"):
 *   CSN_UNIT_GOCT  gOctaveCBR            csnet.py:729-792 (gOctaveConv 604-726 + BN + PReLU per branch)
 *   CSN_UNIT_DW    SimplifiedGOctConvBR  csnet.py:795-851 (depthwise 3x3, weight x100, + BN + PReLU)
 *   CSN_UNIT_MS    MSBlock               csnet.py:116-149 (dilated 3x3 group, weight x100, cat, BN, PReLU)
 *   CSN_UNIT_CLS   cls_layer + F.interpolate(size=input)   csnet.py:306-308,381-385
 */
#ifndef CSNET_HIP_H
#define CSNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSN_ABI_VERSION 3   /* 2 (round 4): csn_profile_bracket_us, options 8-10; 3 (round 5): csn_build_sources_sha16, CSN_OPT_FUSE_ILB live again */
#define CSN_MAX_BRANCH 3
#define CSN_NDIL 5

enum csn_status {
  CSN_OK = 0,
  CSN_E_INVALID = 1,      /* bad argument / inconsistent descriptor            */
  CSN_E_UNSUPPORTED = 2,  /* valid reference configuration not implemented yet */
  CSN_E_HIP = 3,          /* a HIP runtime call failed (see csn_last_hip_error)*/
  CSN_E_NOMEM = 4,
  CSN_E_STATE = 5         /* call order violated (e.g. forward before refresh) */
};

enum csn_unit_kind { CSN_UNIT_GOCT = 1, CSN_UNIT_DW = 2, CSN_UNIT_MS = 3, CSN_UNIT_CLS = 4 };

/* One activation tensor [B][channels][H >> lvl][W >> lvl].  id 0 is always the network input. */
typedef struct csn_act_desc {
  int32_t channels;
  int32_t lvl;
} csn_act_desc;

/* Offsets are in floats into the caller's parameter arena (csn_plan_refresh_params); -1 = absent. */
typedef struct csn_bn_off {
  int64_t weight, bias, running_mean, running_var; /* nn.BatchNorm2d, eps 1e-5        */
  int64_t prelu;                                   /* nn.PReLU(C) weight              */
} csn_bn_off;

typedef struct csn_unit_desc {
  int32_t kind;                    /* csn_unit_kind                                                     */
  int32_t n_in, n_out;             /* number of branches (hi-res first); DW/MS: n_in == n_out            */
  int32_t cin[CSN_MAX_BRANCH];     /* channels per input branch; 0 = branch absent (None)               */
  int32_t cout[CSN_MAX_BRANCH];    /* channels per output branch; 0 = absent                            */
  int32_t in_act[CSN_MAX_BRANCH];  /* activation ids, -1 = absent                                       */
  int32_t out_act[CSN_MAX_BRANCH]; /* CLS: out_act[0] is ignored (result goes to the caller's y)        */
  int32_t ksize;                   /* GOCT: 1 or 3 (padding = ksize/2), csnet.py:33-48                   */
  int32_t stride;                  /* GOCT: 1, or 2 = 2x2 avg-pool of every input first, csnet.py:679-680 */
  int32_t dil_ch[CSN_NDIL];        /* MS: output channels of dilation 1,2,4,8,16 (0 = absent)            */
  int64_t w_off[CSN_NDIL];         /* GOCT/CLS: [0] = weight [sum cout][sum cin][k][k]; DW: per branch
                                      [C][1][3][3]; MS: per dilation [dil_ch][cin][3][3]                 */
  int64_t bias_off;                /* CLS bias, else -1                                                  */
  csn_bn_off bn[CSN_MAX_BRANCH];   /* per output branch (MS: [0])                                        */
} csn_unit_desc;

typedef struct csn_act_info {
  int64_t ws_offset_bytes; /* offset inside the workspace, -1 for the external input */
  int32_t channels, height, width, batch;
} csn_act_info;

typedef struct csn_plan csn_plan;

int csn_abi_version(void);
/* sha256[:16] over the kernel sources (sod100k_amd/csrc: *.hip, *.h, *.inl) this library was BUILT from.  The host side
 * (sod100k_amd/_native.py load()) compares it with the sources in the tree and refuses a stale library; bench lines and
 * counter files carry the same stamp. */
const char* csn_build_sources_sha16(void);
const char* csn_strerror(int status);
/* Text of the last failing HIP call of this thread (empty string if none). */
const char* csn_last_hip_error(void);

/* Build a plan for a fixed (B, H, W).  H and W must be multiples of 16 (CSNet/test.py:80-85).
 * `sub_batch` (0 = B) makes csn_forward walk the batch in slices of that many images so that a
 * unit's output is still resident in the 256 MiB Infinity Cache when the next unit reads it. */
int csn_plan_create(const csn_unit_desc* units, int32_t n_units, const csn_act_desc* acts, int32_t n_acts,
                    int32_t B, int32_t H, int32_t W, int32_t sub_batch, csn_plan** out_plan);
void csn_plan_destroy(csn_plan* plan);

/* Options (default in brackets).  CSN_OPT_FUSE_DW [1]: run the two depthwise units of an ILBlock
 * (conv3x3_1 -> conv3x3_2, csnet.py:74-75) as one kernel that keeps the intermediate in LDS; with 0 every
 * unit's output is materialised in the workspace (per-unit parity probes).
 * CSN_OPT_GRAPH [1]: csn_forward captures its launch sequence into a hipGraph on the second call with the same
 * (x, y, workspace) and replays it afterwards (launch-gap removal); 0 = always launch eagerly.
 * CSN_OPT_FUSE_CLS [1]: the cls_layer (1x1 + bias, csnet.py:306-308) is evaluated in the epilogue of the unit that
 * feeds it (CSFHead.fuse1x1), whose 79-channel output is then never written; 0 = separate launches (probes).
 * CSN_OPT_TILED3 [1]: 3x3 gOctConv passes run as an LDS-tiled implicit GEMM (goct_c3_kernel); 0 = per-pixel tap
 * gathers in goct_pw_kernel (same arithmetic up to the summation order inside a k step).
 * CSN_OPT_FUSE_ILB [1]: (round 5) an ILBlock whose first unit is a 1x1 gOctaveCBR with two input branches and whose low plane
 * has at most 256 pixels (stage 4 at 224 x 224) runs as ONE launch of ilb_kernel (k_ilb.hip: contraction -> the group's planes in
 * LDS -> depthwise pair): its two intermediate tensors are then never written.  0 = the unit kernels (probes).  Needs
 * CSN_OPT_FUSE_DW.  (Rounds 3-4 accepted and ignored the option: round 2's register-resident kernel had been retired.)
 * CSN_OPT_OVERLAP [1]: launches that do not depend on each other -- {z -> high pass} || {low pass} of a 3x3 unit, the
 * per-branch launches of CSFHead.fuse, the three MSBlocks -- are enqueued on parallel stream lanes (fork / join by events on
 * the caller's stream; parallel branches of the hipGraph); 0 = one stream, strictly in order; 2 = additionally all
 * weight-gradient launches of csn_backward on a side lane (own partial buffers; measured: no gain, hence not the default).
 * CSN_OPT_TRAIN_BF16 [0]: BASELINE config 3's dtype -- csn_forward_train / csn_backward keep every activation and activation
 * gradient in the workspace as bfloat16 (round-to-nearest-even on store; all arithmetic, the BN statistics, the weight
 * gradients, parameters and optimizer state stay fp32 / fp64).  x, y, dy at the boundary stay float.  Needs
 * csn_plan_enable_training; csn_forward (eval) is unaffected and stays fp32 (the 1e-4 parity configuration).  Set BEFORE
 * csn_plan_enable_training (and the workspace query), every activation-typed region of the workspace is laid out with 2-byte
 * elements (batch 256: 61 -> 31 GiB); such a plan runs the bf16 train step only -- csn_forward and a switch back to 0 return
 * CSN_E_STATE.  Set afterwards, the tensors keep fp32-sized regions (first half used) and the option can be toggled.
 * CSN_OPT_PW4 [1]: 1x1 gOctaveCBR units with two input branches run on pw4_kernel (k_pw4.hip: lane = low pixel + its 2x2
 * high quad, v_mfma_f32_4x4x1 straight from the load registers, no LDS panel / transpose); 0 = goct_pw_kernel for every
 * 1x1 unit (the round-1/2 path; three-branch input gradients and odd geometries stay on it).
 * CSN_OPT_C3Q [1]: 3x3 gOctConv forward passes at an even resolution run on c3q_kernel (k_c3q.hip: lane = 2x2 output quad,
 * v_mfma_f32_4x4x1 from the load registers, max-pooled copies of the finer branch written by pool2_kernel); 0 =
 * goct_c3_kernel (which also serves bf16 storage, odd sizes and the backward-data launches).
 * CSN_OPT_SLICE_LANES [0]: with sub_batch < B, up to three batch slices run concurrently on the plan's stream lanes, each in
 * its own workspace region (set BEFORE csn_plan_workspace_bytes, which then reports that many regions); the per-launch
 * latency of the small maps of one slice is covered by the other slices' launches.  Results are identical to the
 * sequential slices (same kernels on the same data).
 * CSN_OPT_INPUT_GRAD [0]: (round 6) csn_backward also forms the gradient w.r.t. the image batch x -- autograd's x.grad through
 * csnet.py:365-387, which SURVEY 8(b) lists as csn_backward's `dx`.  The reference's callers never ask for it (train.py:203-216
 * feeds images that do not require grad), so it is an option and not an argument: set BEFORE csn_plan_enable_training (it adds one
 * gradient buffer of x's shape to the workspace); after csn_backward the gradient lies in the workspace at
 * csn_plan_train_act_info(plan, 0).grad_offset_bytes[0], [B][3][H][W], float -- or bfloat16 under CSN_OPT_TRAIN_BF16 (`bf16`). */
enum csn_option { CSN_OPT_FUSE_DW = 1, CSN_OPT_GRAPH = 2, CSN_OPT_FUSE_CLS = 3, CSN_OPT_TILED3 = 4, CSN_OPT_FUSE_ILB = 5,
                  CSN_OPT_OVERLAP = 6, CSN_OPT_TRAIN_BF16 = 7, CSN_OPT_PW4 = 8, CSN_OPT_C3Q = 9, CSN_OPT_SLICE_LANES = 10,
                  CSN_OPT_INPUT_GRAD = 11 };
int csn_plan_set_option(csn_plan* plan, int32_t option, int32_t value);

size_t csn_plan_workspace_bytes(const csn_plan* plan);
int csn_plan_act_info(const csn_plan* plan, int32_t act_id, csn_act_info* out);
int32_t csn_plan_num_units(const csn_plan* plan);

/* Debug / parity probes of the train step (valid after csn_plan_enable_training): where an activation's companions live
 * in the workspace.  z: the raw convolution output of the producing unit after csn_forward_train, OVERWRITTEN by the
 * gradient w.r.t. z (dz) in csn_backward (or dz goes over the activation: dz_offset_bytes); grad[s]: the gradient w.r.t. the activation contributed by its s-th consumer
 * (-1: none).  Elements are bfloat16 when CSN_OPT_TRAIN_BF16 is set (`bf16` = 1), else float. */
typedef struct csn_train_act_info {
  int64_t act_offset_bytes;     /* the activation itself (-1: the external input; bf16 mode keeps a copy at x16)  */
  int64_t z_offset_bytes;       /* -1 for the external input                                                       */
  int64_t grad_offset_bytes[2];
  int64_t x16_offset_bytes;     /* bf16 copy of the external input (act 0), -1 otherwise / in fp32 mode            */
  int32_t n_consumers;
  int32_t bf16;
  int64_t dz_offset_bytes;      /* where csn_backward leaves dz: z_offset_bytes (over z), or act_offset_bytes when the BatchNorm
                                   backward of that output runs fused with the adjoint upsampling (the activation is dead by then) */
} csn_train_act_info;
int csn_plan_train_act_info(const csn_plan* plan, int32_t act_id, csn_train_act_info* out);
/* Which of the two gradient buffers of its input activation `branch` unit `unit` writes (0 / 1; < 0: none). */
int32_t csn_plan_unit_in_slot(const csn_plan* plan, int32_t unit, int32_t branch);

/* (Re)pack parameters: folds BN running stats + affine into per-channel scale/shift, applies the x100
 * of Conv2dX100 (CSNet/model/conv2d.py:104) and re-lays the gOctConv weight blocks for scalar loads.
 * Must be called after every parameter change and before csn_forward.  `arena` is a device pointer. */
int csn_plan_refresh_params(csn_plan* plan, const float* arena, int64_t arena_floats, void* stream);

/* Eval-mode forward: x [B][3][H][W] -> y [B][1][H][W] logits (no sigmoid), csnet.py:365-387. */
int csn_forward(csn_plan* plan, const float* x, float* y, void* workspace, void* stream);

/* Train-mode forward (nn.BatchNorm2d batch statistics, csnet.py:764,825,138; Oct_bn_hook csnet.py:391-410):
 * every BatchNorm normalises with the biased variance of the current batch, its running_mean/running_var
 * inside `arena` are updated in place (momentum 0.1, unbiased variance) and `*penalty` (device, fp64, NOT
 * cleared) accumulates  sum 0.5 * flop_w[u][j] * sum_{n,c} |mean_hw y[n,c]| * gamma_c^2  over the output branches
 * j of the units u (flop_w is a HOST array [n_units][CSN_MAX_BRANCH]; 0 = not hooked).  The caller increments
 * num_batches_tracked and divides the penalty by its batch size (csnet.py:324-330).  Requires sub_batch == 0.
 * With csn_plan_enable_training the raw outputs are kept for csn_backward. */
int csn_forward_train(csn_plan* plan, const float* x, float* y, void* workspace, float* arena, int64_t arena_floats,
                      const float* flop_w, double* penalty, void* stream);

/* Training buffers.  Call once after csn_plan_create and BEFORE csn_plan_workspace_bytes / csn_plan_refresh_params:
 * adds, per activation, a buffer for the raw convolution output z (saved for backward, later overwritten by dz) and
 * one gradient buffer per consumer, the backward scratch, and the transposed / tap-flipped weight images of the
 * input-gradient launches (re-packed by every csn_plan_refresh_params).  Requires sub_batch == 0. */
int csn_plan_enable_training(csn_plan* plan);

/* Backward of the last csn_forward_train call (autograd through csnet.py:365-387 as train.py:211-214 drives it).
 * dy [B][1][H][W] is d loss / d logits; `grad` is an arena with the parameter arena's offsets: the gradient of every
 * convolution weight, BN weight/bias, PReLU weight and the cls bias is WRITTEN (not accumulated) there, other
 * positions are left untouched.  `pen_scale` = d loss / d (penalty sum) = FLOPS.WEIGHT / batchsize (train.py:91,210)
 * adds the dynamic-weight-decay term's gradient w.r.t. the BN weights of the hooked units (same flop_w table as the
 * forward call; y.detach() in Oct_bn_hook: no gradient through the activations).  The gradient w.r.t. x (SURVEY 8(b)'s `dx`) is
 * formed on plans with CSN_OPT_INPUT_GRAD and left in the workspace (csn_plan_train_act_info(plan, 0).grad_offset_bytes[0]). */
int csn_backward(csn_plan* plan, const float* x, const float* dy, void* workspace, const float* arena, float* grad,
                 int64_t arena_floats, const float* flop_w, float pen_scale, void* stream);

/* Caller-side training maths on flat device buffers.
 * csn_bce_with_logits: *loss (device, fp64, NOT cleared) += mean BCE-with-logits (train.py:209), dy = d mean / d y.
 * csn_adam_step: torch.optim.Adam with L2 weight decay folded into the gradient (train.py:108-123); wd[i] is the
 * per-element weight decay of the parameter group element i belongs to; `step` counts from 1. */
int csn_bce_with_logits(const float* y, const float* t, float* dy, int64_t n, double* loss, void* stream);
int csn_adam_step(float* p, const float* g, float* m, float* v, const float* wd, int64_t n, float lr, float beta1,
                  float beta2, float eps, int32_t step, void* stream);

/* The steps either side of csn_forward in the inference caller (CSNet/test.py), on the device:
 * csn_normalize_nchw: float H x W x 3 images in [0,1] (already at the network size) -> ImageNet-normalised NCHW
 *   ((img - mean) / std, test.py:68-69,86);
 * csn_saliency_u8: logits -> (sigmoid(y) * 255).astype(uint8), truncation as numpy does (test.py:92-96).
 * The resizes of test.py:76-85 / 94-96 (skimage.transform.resize, order 1, mode='reflect', anti_aliasing=False = bilinear
 * with half-pixel centres, source coordinates outside the image MIRRORED about the edge pixel centre -- what skimage's
 * scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True) call computes) on the device as well:
 * csn_resize_normalize_nchw: B float images Hi x Wi x 3 in [0,1] -> resize to H x W -> normalise -> NCHW;
 * csn_saliency_resize_u8: logits H x W -> sigmoid -> resize to h x w -> (p * 255) truncated to uint8;
 * csn_resize_bilinear: planar float tensors [planes][Hi][Wi] -> [planes][Ho][Wo].
 * skimage is not available to the build; the rule is pinned against scipy.ndimage.zoom (the routine skimage calls) and
 * through properties (identity, constants, affine ramps) in tests/resize_cases.py. */
int csn_normalize_nchw(const float* hwc, float* nchw, int64_t B, int64_t H, int64_t W, void* stream);
int csn_saliency_u8(const float* logits, uint8_t* out, int64_t n, void* stream);
int csn_resize_normalize_nchw(const float* hwc, float* nchw, int32_t B, int32_t Hi, int32_t Wi, int32_t H, int32_t W, void* stream);
int csn_saliency_resize_u8(const float* logits, uint8_t* out, int32_t H, int32_t W, int32_t h, int32_t w, void* stream);
int csn_resize_bilinear(const float* in, float* out, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, void* stream);

/* Evaluation metrics (SalMetric/src/sal_metric.cpp:87-120, the reference's only native component): for n_images
 * equally sized uint8 maps, ACCUMULATES (caller zeroes) per image the joint histogram hist[img][2*v + (gt > 128)]
 * of the predicted value v and the binarised ground truth, and abs_sum[img] = sum |sal - gt|.  Every threshold's
 * precision / recall and the MAE follow from these (sod100k_amd/metric.py) -- one pass instead of 256. */
int csn_sal_hist(const uint8_t* sal, const uint8_t* gt, int64_t npix, int32_t n_images, uint64_t* hist, uint64_t* abs_sum,
                 void* stream);

/* Measurement helper (no counterpart in the reference): dst[0 .. n) = src[0 .. n) floats, n a multiple of 4, as a plain streaming
 * kernel.  bench.py times it for `roofline.peak_measured`, the on-box HBM figure next to the 8 TB/s spec. */
int csn_stream_copy(const float* src, float* dst, int64_t n, void* stream);

/* The validation loop of the training caller (CSNet_training/train.py:262-276) for ONE picture: sigmoid of the
 * hi x wi logits -> F.interpolate(size=(h, w), bilinear, align_corners=False) -> (x * 255).int() / 255 ->
 * mean |. - target| over the h x w target (float, the picture's own size).  ADDS the mean to *mae (caller zeroes). */
int csn_val_mae(const float* logits, int32_t hi, int32_t wi, const float* target, int32_t h, int32_t w, double* mae,
                void* stream);

/* Same as csn_forward (eager launches) but records a HIP event on `stream` after every kernel launch and
 * returns the mean duration per unit over `iters` passes (unit_ms[n_units], milliseconds).  Synchronises. */
int csn_forward_profile(csn_plan* plan, const float* x, float* y, void* workspace, void* stream,
                        int32_t iters, float* unit_ms);

/* Per-kernel aggregation of the last csn_forward_profile call: every launch was bracketed by HIP events on
 * the launch stream; entry i gives the kernel name, its total milliseconds per forward and its launch count
 * per forward (so ms / launches is the mean launch duration rocprofv3 --stats reports for that name). */
int32_t csn_profile_num_kernels(const csn_plan* plan);
/* microseconds an event pair adds to one launch (measured on empty launches by the last csn_forward_profile call and already
 * subtracted from every per-unit / per-kernel time it reports, so that they agree with a kernel trace's durations) */
double csn_profile_bracket_us(const csn_plan* plan);
int csn_profile_kernel(const csn_plan* plan, int32_t i, const char** name, double* ms_per_forward, int32_t* launches);

/* Name of the dominant kernel of unit `u` (static string) and the algorithmic bytes it moves per
 * csn_forward (sum of unit input + output activation bytes). */
const char* csn_unit_kernel_name(const csn_plan* plan, int32_t u);
int64_t csn_unit_algorithmic_bytes(const csn_plan* plan, int32_t u);

#ifdef __cplusplus
}
#endif
#endif /* CSNET_HIP_H */
