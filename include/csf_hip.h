/* csf_hip.h -- C ABI of the CSF+Res2Net decoder head (BASELINE config 5) in libcsnet_hip.so.
 *
 * The reference network is CSF+Res2Net/networks/csf_res2net.py:227-256 (CSFNet): a Res2Net-50 v1b backbone whose
 * four stage outputs feed a cross-stage-fusion head
 *
 *     fuse     gOctaveCBR 4 -> 4 branches, 1x1            gOctConv.py:60-114, 116-152   (csf_res2net.py:240-241)
 *     ms       PallMSBlock, dense dilated 3x3 (d=1..16)   csf_res2net.py:174-223        (csf_res2net.py:242)
 *     fuse1x1  gOctaveCBR 4 -> 1 branch, 1x1              gOctConv.py                    (csf_res2net.py:243-244)
 *     cls_layer (1x1 + bias) and bilinear resize to the input size                      (csf_res2net.py:245,253-254)
 *
 * with GroupNorm(32) + PReLU after every gOctConv / MSBlock (gOctConv.py:127-130, csf_res2net.py:204-205).
 * This library computes the HEAD (16.2 of the 38.4 GFLOP per 352x352 image, all of it dense contractions on the fp32
 * matrix cores); the backbone's plain convolutions stay with the caller (PyTorch / MIOpen), which hands over the four
 * feature maps as device pointers.
 *
 * Conventions are those of csnet_hip.h: plain C types, int status (0 = CSN_OK, codes of csn_status, text through
 * csn_strerror / csn_last_hip_error), launches on the caller's stream, no allocation or sync after creation, the
 * caller owns features, output, parameter arena and workspace; the library owns the plan and its weight images.
 * All tensors are planar fp32: feature i is [batch][cin[i]][h[i]][w[i]], the result [batch][1][out_h][out_w] (logits).
 */
#ifndef CSF_HIP_H_
#define CSF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#include "csnet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CSF_MAX_BRANCH 4
#define CSF_NDIL 5

/* Offsets are in floats into the caller's flat parameter arena (the tensors of CSFNet.state_dict() minus `base.*`). */
typedef struct csf_gn_off {
  int64_t weight, bias;    /* nn.GroupNorm(32, C) affine, gOctConv.py:128 / csf_res2net.py:204 */
  int64_t prelu;           /* nn.PReLU(C).weight, gOctConv.py:129 / csf_res2net.py:205 */
} csf_gn_off;

typedef struct csf_head_desc {
  int32_t n_branch;                          /* 4 (csf_res2net.py:235-238) */
  int32_t gn_groups;                         /* 32 */
  int32_t cin[CSF_MAX_BRANCH];               /* backbone channels per level: 256, 512, 1024, 2048 */
  int32_t cmid[CSF_MAX_BRANCH];              /* head channels per level: 128, 256, 512, 512 (gOctConv.py:78-83 bounds) */
  int32_t ms_split[CSF_MAX_BRANCH][CSF_NDIL];/* MSBlock output channels per dilation 1,2,4,8,16 (csf_res2net.py:196-202) */
  int64_t fuse_w;                            /* fuse.conv.weights    [sum cmid][sum cin][1][1] */
  csf_gn_off fuse_gn[CSF_MAX_BRANCH];        /* fuse.bns.j / fuse.prelus.j */
  int64_t ms_w[CSF_MAX_BRANCH][CSF_NDIL];    /* ms.convs.j.msconv.d.weight [split][cmid[j]][3][3] */
  csf_gn_off ms_gn[CSF_MAX_BRANCH];          /* ms.convs.j.bn / .prelu */
  int64_t fuse1_w;                           /* fuse1x1.conv.weights [sum cmid][sum cmid][1][1] */
  csf_gn_off fuse1_gn;                       /* fuse1x1.bns.0 / fuse1x1.prelus.0 */
  int64_t cls_w, cls_b;                      /* cls_layer.weight [1][sum cmid][1][1], cls_layer.bias [1] */
} csf_head_desc;

typedef struct csf_head csf_head;

/* Compiles the launch sequence for one geometry: batch images, feature i of h[i] x w[i] pixels (any sizes: the
 * cross-branch resizes follow F.interpolate(size=..., mode='bilinear', align_corners=False), gOctConv.py:96-101),
 * logits resized to out_h x out_w (csf_res2net.py:254). */
int csf_head_create(const csf_head_desc* desc, int32_t batch, const int32_t* h, const int32_t* w, int32_t out_h,
                    int32_t out_w, csf_head** out);
void csf_head_destroy(csf_head* head);
size_t csf_head_workspace_bytes(const csf_head* head);

/* (Re)builds the zero-padded row-major weight images from the arena; call after creation and whenever parameters change. */
int csf_head_refresh_params(csf_head* head, const float* arena, int64_t arena_floats, void* stream);

/* CSFNet.forward from `features = self.base(x)` onward (csf_res2net.py:250-255). */
int csf_head_forward(csf_head* head, const float* const* features, float* logits, void* workspace, void* stream);

/* Introspection for tests: stage 0 = fuse, 1 = ms (outputs of branch `branch` after GroupNorm + PReLU),
 * 2 = fuse1x1 (branch 0; kept BEFORE its GroupNorm, which is fused into the cls kernel). */
int csf_head_stage_info(const csf_head* head, int32_t stage, int32_t branch, int64_t* ws_offset_bytes, int32_t* channels,
                        int32_t* height, int32_t* width);

/* Multiply-accumulate work of one forward (2 flops each), for the matrix-core roofline of the GEMM kernel. */
int64_t csf_head_macs(const csf_head* head);

/* ---- eval BatchNorm (+ residual) (+ ReLU) of the backbone in ONE elementwise pass ------------------------------------
 * The Res2Net bottlenecks (csf_res2net.py:69-103) follow every convolution with  relu(bn(.))  or
 * relu(bn3(.) + residual); PyTorch issues BatchNorm, add and ReLU as separate kernels (2 + 3 + 2 trips over the tensor
 * for the residual sites, 30 % of the backbone's time at batch 32).  In place on x = [batch][channels][hw]:
 *     x = act((x - mean) / sqrt(var + eps) * gamma + beta (+ residual)),   act = max(., 0) when relu != 0
 * gamma / beta / mean / var are nn.BatchNorm2d's parameters and running statistics (device pointers, [channels]). */
int csf_bn_act(float* x, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
               const float* residual, int32_t batch, int32_t channels, int32_t hw, int32_t relu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CSF_HIP_H_ */
