"""CPU oracle for the CSNet hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``sod100k_amd``) never does; it fails
loudly when its HIP library is missing instead of falling back to this file.

What it is: a *functional* restatement (plain ``torch`` CPU ops over a flat
``state_dict``; no ``nn.Module``) of the reference path
``CSNet.forward`` and of the training-step maths around it.  The primitive
arithmetic of the reference (conv2d, batch_norm, prelu, pooling, bilinear, BCE,
Adam) lives in PyTorch ATen, which is not vendored under the reference tree
("PyTorch 1.0+", CSNet/README.md:14); the restatement therefore calls the same
ATen CPU kernels through ``torch.nn.functional`` and restates only the
reference's own composition logic.  Citations are ``file:line`` into the
reference checkout (never present at run time on the GPU box).

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, imported in the build container by ``oracle/make_goldens.py`` and frozen
under ``tests/golden/`` (logits, per-unit probes, op micro-goldens, one train
step, DP emulation).  ``tests/test_oracle_vs_golden.py`` checks them on CPU.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, csnet.py:764,825,138
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default
DILATIONS = (1, 2, 4, 8, 16)   # csnet.py:121


# --------------------------------------------------------------------------
# layer_config handling (csnet.py:414-523, Appendix A of SURVEY.md)
# --------------------------------------------------------------------------
def load_layer_config_json(path: str) -> list:
    """Load the JSON re-encoding of a pickled ``layer_config`` (G1 fixture)."""
    with open(path) as f:
        raw = json.load(f)
    return layer_config_from_lists(raw["layer_config"])


def layer_config_from_lists(raw: list) -> list:
    out = []
    for e in raw[:-1]:
        out.append([np.asarray(v, dtype=np.float64) for v in e])
    out.append([int(v) for v in raw[-1]])
    return out


def init_layers(basewidth: int, basic_split: Sequence[float] = (1,)) -> list:
    """Un-pruned config generator, restating csnet.py:414-518."""
    sp = np.array([float(v) for v in basic_split])
    one = np.array([1.0])
    stages = [3, 4, 6, 4]
    w = basewidth
    cfg = [[np.array([3.0]), w * sp], [w * sp, w * sp]]
    cfg += [[w * sp, w * sp] for _ in range(1, stages[0])]
    cfg += [[w * sp, 2 * w * sp]]
    cfg += [[2 * w * sp, 2 * w * sp] for _ in range(1, stages[1] - 1)]
    cfg += [[2 * w * sp, 2 * w * one]]
    cfg += [[2 * w * one, 4 * w * sp]]
    cfg += [[4 * w * sp, 4 * w * sp] for _ in range(1, stages[2] - 1)]
    cfg += [[4 * w * sp, 4 * w * one]]
    cfg += [[4 * w * one, 4 * w * sp]]
    cfg += [[4 * w * sp, 4 * w * sp] for _ in range(1, stages[3] - 1)]
    cfg += [[4 * w * sp, 4 * w * one]]
    s2, s3, s4 = 2 * w, 4 * w, 4 * w
    mid = np.array([s2 // 3, s3 // 3, s4 // 3])
    cfg.append([np.array([s2, s3, s4]), mid.copy()])
    dil = []
    for br in mid:
        each = br // 5
        dil.append([each] * 4 + [br - each * 4])
    cfg.append([mid.copy(), mid.copy(), np.array(dil)])
    cfg.append([mid.copy(), np.array([mid.sum()])])
    for e in cfg:
        e[0] = np.round(e[0]).astype(np.int32)
        e[1] = np.round(e[1]).astype(np.int32)
    cfg.append(stages)
    return cfg


def _alphas(split) -> List[float]:
    """ILBlock / CSFHead: ``alpha = split / int(round(sum(split)))`` (csnet.py:26-31,157-165)."""
    split = np.asarray(split, dtype=np.float64)
    total = int(round(float(split.sum())))
    return (split * 1.0 / total).tolist(), total


def _bounds(total: int, alphas: Sequence[float]) -> List[int]:
    """Cumulative-alpha channel boundaries of gOctaveConv (csnet.py:641-650,683-691)."""
    cum = [0]
    t = 0
    for a in alphas:
        t += a
        cum.append(t)
    return [int(round(total * c)) for c in cum]


# --------------------------------------------------------------------------
# bf16 activation storage (BASELINE config 3) -- emulation of what the HIP train path keeps in HBM
# --------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    """A tensor that lives in memory as bfloat16: the forward value and the gradient that flows back through it are both
    rounded to nearest-even bf16 (torch's float -> bfloat16 cast), everything else stays float32."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundBF16Fwd(torch.autograd.Function):
    """Stored as bfloat16 on the way forward only: the gradient w.r.t. this tensor is consumed where it is formed and never
    goes to memory (HIP path: dz of a depthwise unit, formed on load inside dw3x3_bwd_kernel since round 3)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBF16Bwd(torch.autograd.Function):
    """The mirror image: the VALUE never goes to memory (HIP path: an activation whose only consumer is a depthwise unit is
    formed on load from the stored z, round 3), its gradient does (the consumer writes dx)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


_ACT_BF16 = False
# bf16 storage emulation of the one-pass depthwise kernels (HIP default; the tests that force the two-pass fallbacks -- CSN_DW_BWD_SPLIT,
# CSN_BN_BWD_FUSE=0, CSN_BN_FWD_FUSE=0 -- flip these): DW_IN_STORED = is the activation a depthwise unit consumes a stored tensor
# (ILBlock: the outputs of conv1x1 and conv3x3_1)?  DW_DZ_STORED =  False = the HIP default (one-pass depthwise backward with
# the BatchNorm backward's apply step fused in); tests that force the two-pass fallback (CSN_DW_BWD_SPLIT, CSN_BN_BWD_FUSE=0)
# set it to True.
DW_DZ_STORED = False
DW_IN_STORED = False
Z_CAPTURE = None     # tests: set to a list to collect every BatchNorm input (the raw conv outputs z) in call order
# tests (unit-local train checks): PReLU has a kink at 0, so a pre-activation that is zero to within fp32 rounding may
# legitimately take either branch on the device.  PRELU_Y: set to a list to collect every pre-activation (BatchNorm output) in
# call order; PRELU_FLIP: {call index: bool mask} of elements that take the OTHER branch (value unchanged to ~1e-9, derivative
# swapped between 1 and alpha).
PRELU_Y = None
PRELU_FLIP = None


class bf16_activations:
    """``with bf16_activations():`` every tensor the HIP bf16 train mode stores (input copy, pooled copies, raw conv
    outputs z, activations, half-resolution logits) is rounded through bf16 at that point, forward and backward."""

    def __enter__(self):
        global _ACT_BF16
        self._old = _ACT_BF16
        _ACT_BF16 = True

    def __exit__(self, *exc):
        global _ACT_BF16
        _ACT_BF16 = self._old


def _st(x):
    return _RoundBF16.apply(x) if (_ACT_BF16 and x is not None) else x


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def goct_conv(xs, weight, alpha_in, alpha_out, stride, padding):
    """gOctaveConv.forward, csnet.py:664-726.  ``xs``: list of tensors/None, hi-res first."""
    cout, cin = weight.shape[0], weight.shape[1]
    bi = _bounds(cin, alpha_in)
    bo = _bounds(cout, alpha_out)
    ysets = [[] for _ in alpha_out]
    for i in range(len(alpha_in)):
        if xs[i] is None:
            continue
        x = _st(F.avg_pool2d(xs[i], (2, 2), stride=2)) if stride == 2 else xs[i]   # :679-682
        if bi[i] == bi[i + 1]:
            continue
        for j in range(len(alpha_out)):
            if bo[j] == bo[j + 1]:
                continue
            w = weight[bo[j]:bo[j + 1], bi[i]:bi[i + 1]]
            if i > j:        # low -> high: conv, then bilinear up  (:702-707)
                y = F.conv2d(x, w, None, 1, padding)
                if padding:      # 3x3: the HIP path materialises this partial sum (1x1 units interpolate x instead)
                    y = _st(y)
                y = F.interpolate(y, scale_factor=2 ** (i - j), mode="bilinear")
            elif i < j:      # high -> low: max-pool, then conv     (:708-714)
                k = 2 ** (j - i)
                y = F.conv2d(F.max_pool2d(x, k, stride=k), w, None, 1, padding)
            else:
                y = F.conv2d(x, w, None, 1, padding)
            ysets[j].append(y)
    return [sum(v) if len(v) else None for v in ysets]     # python sum: 0 + y0 + y1 (:720-722)


def bn_prelu(x, sd, bn_prefix, prelu_key, training, dz_stored=True, y_stored=True):
    """nn.BatchNorm2d (eps 1e-5, momentum 0.1) followed by per-channel nn.PReLU."""
    # the raw conv output z is a stored tensor (statistics are taken from what was stored); so is its gradient dz, except
    # where the consumer forms it on load (dz_stored False)
    if dz_stored or not _ACT_BF16:
        x = _st(x)
    else:
        x = _RoundBF16Fwd.apply(x)
    if Z_CAPTURE is not None:
        Z_CAPTURE.append(x)
    y = F.batch_norm(x, sd[bn_prefix + ".running_mean"], sd[bn_prefix + ".running_var"],
                     sd[bn_prefix + ".weight"], sd[bn_prefix + ".bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[bn_prefix + ".num_batches_tracked"] += 1
    out = F.prelu(y, sd[prelu_key])
    if PRELU_Y is not None:
        idx = len(PRELU_Y)
        PRELU_Y.append(y.detach())
        if PRELU_FLIP is not None and idx in PRELU_FLIP:
            al = sd[prelu_key].view(1, -1, 1, 1)
            out = torch.where(PRELU_FLIP[idx], torch.where(y >= 0, al * y, y), out)
    if not y_stored and _ACT_BF16 and training:
        return _RoundBF16Bwd.apply(out)
    return _st(out)


def goct_cbr(xs, sd, prefix, alpha_in, alpha_out, k, stride, training, y_stored=True):
    """gOctaveCBR.forward, csnet.py:778-792 (std_conv branch :779-786 uses Conv2dX100)."""
    w = sd[prefix + ".conv.weight"]
    pad = 1 if k == 3 else 0
    if len(alpha_in) == 1 and len(alpha_out) == 1:       # :751-754  Conv2dX100, real stride
        x = xs[0] if isinstance(xs, (list, tuple)) else xs
        y = F.conv2d(x, 100.0 * w, None, stride, pad)     # conv2d.py:104
        return bn_prelu(y, sd, prefix + ".bns.0", prefix + ".prelus.0.weight", training, y_stored=y_stored)
    ys = goct_conv(xs, w, alpha_in, alpha_out, stride, pad)
    for j in range(len(ys)):
        if ys[j] is not None:
            ys[j] = bn_prelu(ys[j], sd, f"{prefix}.bns.{j}", f"{prefix}.prelus.{j}.weight", training, y_stored=y_stored)
    return ys


def simplified_cbr(xs, sd, prefix, training, y_stored=True):
    """SimplifiedGOctConvBR.forward, csnet.py:838-851: depthwise 3x3 (x100) + BN + PReLU per branch."""
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    ys = []
    for i, x in enumerate(xs):
        if x is None:
            ys.append(None)
            continue
        w = sd[f"{prefix}.convs.{i}.weight"]
        y = F.conv2d(x, 100.0 * w, None, 1, 1, 1, w.shape[0])          # conv2d.py:104
        ys.append(bn_prelu(y, sd, f"{prefix}.bns.{i}", f"{prefix}.prelus.{i}.weight", training, dz_stored=DW_DZ_STORED,
                           y_stored=y_stored))
    return ys


def ms_block(x, sd, prefix, dil_channels, training):
    """MSBlock.forward, csnet.py:141-149: dilated 3x3 (x100) group -> cat -> BN -> PReLU."""
    outs = []
    for d, dil in enumerate(DILATIONS):
        if int(dil_channels[d]) != 0:
            w = sd[f"{prefix}.msconv.{d}.weight"]
            outs.append(F.conv2d(x, 100.0 * w, None, 1, dil, dil))
    out = torch.cat(outs, dim=1)
    return bn_prelu(out, sd, prefix + ".bn", prefix + ".prelu.weight", training)


# --------------------------------------------------------------------------
# network description derived from layer_config (csnet.py:209-308)
# --------------------------------------------------------------------------
def block_table(layer_config) -> List[dict]:
    """One entry per ILBlock in forward order: name, in/out splits, stride, first."""
    stages = layer_config[-1]
    blocks = []
    idx = 0
    blocks.append(dict(name="stage0.0", inlist=np.array([3.0]), outlist=layer_config[0][1],
                       stride=1, first=True))
    idx = 1
    for s, n in enumerate(stages):
        for b in range(n):
            blocks.append(dict(name=f"stage{s + 1}.{b}", inlist=layer_config[idx][0],
                               outlist=layer_config[idx][1],
                               stride=2 if (s >= 1 and b == 0) else 1, first=False))
            idx += 1
    return blocks


def il_block(xs, sd, blk, training, taps=None):
    """ILBlock.forward, csnet.py:72-76."""
    a_in, _ = _alphas(blk["inlist"])
    a_out, _ = _alphas(blk["outlist"])
    k = 3 if (blk["first"] or blk["stride"] == 2) else 1           # :33-48
    p = blk["name"]
    y = goct_cbr(xs, sd, p + ".conv1x1", a_in, a_out, k, blk["stride"], training, y_stored=DW_IN_STORED)
    if isinstance(y, torch.Tensor):
        y = [y]
    if taps is not None:
        taps[p + ".conv1x1"] = y
    y = simplified_cbr(y, sd, p + ".conv3x3_1", training, y_stored=DW_IN_STORED)
    if taps is not None:
        taps[p + ".conv3x3_1"] = y
    y = simplified_cbr(y, sd, p + ".conv3x3_2", training)
    if taps is not None:
        taps[p + ".conv3x3_2"] = y
    return y


def csf_head(xs, sd, cfg3, training, taps=None):
    """CSFHead.forward, csnet.py:152-206."""
    a_in, _ = _alphas(cfg3[0][0])
    a_mid_in, mid_in = _alphas(cfg3[1][0])
    a_mid_out, mid_out = _alphas(cfg3[1][1])
    dils = cfg3[1][2]
    y = goct_cbr(xs, sd, "oct_fuse.fuse", a_in, a_mid_in, 1, 1, training)
    if taps is not None:
        taps["oct_fuse.fuse"] = y
    z = []
    for i in range(len(a_mid_in)):                                  # PallMSBlock, :92-113
        if max(dils[i]) != 0:
            z.append(ms_block(y[i], sd, f"oct_fuse.ms.convs.{i}", dils[i], training))
        else:
            z.append(None)
    if taps is not None:
        taps["oct_fuse.ms"] = z
        for i, t in enumerate(z):
            if t is not None:
                taps[f"oct_fuse.ms.convs.{i}"] = [t]
    out = goct_cbr(z, sd, "oct_fuse.fuse1x1", a_mid_out, [1], 1, 1, training)
    if isinstance(out, torch.Tensor):
        out = [out]
    if taps is not None:
        taps["oct_fuse.fuse1x1"] = out
    return out


def csnet_forward(layer_config, sd: Dict[str, torch.Tensor], x: torch.Tensor,
                  training: bool = False, taps: Optional[dict] = None) -> torch.Tensor:
    """CSNet.forward, csnet.py:365-387.  ``sd`` uses the reference's state_dict keys.

    With ``training=True`` BN uses batch statistics and updates the running buffers in
    ``sd`` in place (like nn.BatchNorm2d).  ``taps`` (optional dict) receives every unit's
    output branch list, keyed by the reference module path.
    """
    stages = layer_config[-1]
    blocks = block_table(layer_config)
    cur = [_st(x)]
    heads = []
    bi = 0
    cur = il_block(cur, sd, blocks[bi], training, taps); bi += 1
    for s, n in enumerate(stages):
        for _ in range(n):
            cur = il_block(cur, sd, blocks[bi], training, taps); bi += 1
        if s >= 1:
            heads.append(cur[0])                                     # x2[0], x3[0], x4[0]  (:380)
    fuse = csf_head(heads, sd, layer_config[bi:bi + 3], training, taps)
    out = _st(F.conv2d(fuse[0], sd["cls_layer.weight"], sd["cls_layer.bias"]))        # :381
    if taps is not None:
        taps["cls_layer"] = [out]
    return F.interpolate(out, x.shape[2:], mode="bilinear", align_corners=False)  # :382-385


# --------------------------------------------------------------------------
# training-step maths (csnet.py:313-355,391-410; train.py:97-123,203-216)
# --------------------------------------------------------------------------
def flop_weights(layer_config, expandflop: float = 2) -> Dict[str, List[float]]:
    """Per hooked sub-module branch weights of Oct_bn_hook, restating csnet.py:332-355,393-398."""
    stages = list(layer_config[-1])
    real = stages.copy()
    real[0] += 1
    base = expandflop ** (len(stages) - 1)
    out = {}
    stage = 0
    in_stage = 0
    for blk in block_table(layer_config):
        nb = len(np.atleast_1d(blk["outlist"]))
        for sub in ("conv1x1", "conv3x3_1", "conv3x3_2"):
            f = base * (expandflop ** (nb - 1))
            ws = []
            for _ in range(nb):
                ws.append(f)
                f /= expandflop
            out[f"{blk['name']}.{sub}"] = ws
        in_stage += 1
        if in_stage == real[stage]:
            base /= expandflop
            stage += 1
            in_stage = 0
    return out


def gap_penalty(sd, taps, fweights, batchsize: int):
    """``model.get_flops()`` after one hooked forward: csnet.py:324-330 and 391-410.

    0.5 * sum_modules sum_branches w_k * sum_{n,c} |mean_hw y[n,c]| * gamma_c^2, / batchsize.
    """
    total = 0
    for mod, ws in fweights.items():
        ys = taps[mod]
        terms = []
        for k, y in enumerate(ys):
            if y is None:
                continue
            gap = F.adaptive_avg_pool2d(y.detach(), 1).squeeze().abs()
            gamma = sd[f"{mod}.bns.{k}.weight"]
            terms.append((ws[k] * gap * torch.pow(gamma, 2)).sum())
        total = total + 0.5 * sum(terms)
    return total / batchsize


def param_groups(names: Sequence[str]):
    """train.py:101-107 (the typo'd condition is reproduced: conv3x3_2.bns is NOT picked)."""
    picked, normal = [], []
    for n in names:
        if "stage" in n and ("conv1x1.bns" in n or "conv3x3_1.bns" in n or "conv3x3_1.bns" in n) \
                and "weight" in n:
            picked.append(n)
        else:
            normal.append(n)
    return normal, picked


def is_param(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


def train_step(layer_config, sd, x, target, *, expandflop=1.0, flops_weight=3.0, batchsize=None,
               lr=1e-4, wd=5e-3, betas=(0.9, 0.99), eps=1e-8, adam_state=None, use_penalty=True, act_dtype=None):
    """One iteration of train.py:203-216 on a flat state_dict (updated in place).  ``act_dtype="bf16"`` emulates the bf16
    activation storage of BASELINE config 3 (see bf16_activations).

    Returns dict(loss_bce, penalty, grads).  Adam is torch.optim.Adam semantics with L2-in-gradient
    weight decay, two groups (train.py:108-123).
    """
    batchsize = batchsize or x.shape[0]
    pnames = [k for k in sd if is_param(k)]
    for k in pnames:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    taps = {}
    import contextlib
    with (bf16_activations() if act_dtype == "bf16" else contextlib.nullcontext()):
        out = csnet_forward(layer_config, sd, x, training=True, taps=taps)
        bce = F.binary_cross_entropy_with_logits(out, target)
        loss = bce
        pen = None
        if use_penalty:
            pen = gap_penalty(sd, taps, flop_weights(layer_config, expandflop), batchsize)
            loss = loss + flops_weight * pen
        loss.backward()
    # a parameter without a path to the loss (an output branch nobody consumes in a pruned net) has grad None in autograd
    grads = {k: (sd[k].grad.detach().clone() if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in pnames}
    normal, picked = param_groups(pnames)
    if adam_state is None:
        adam_state = {}
    with torch.no_grad():
        for k in pnames:
            g = grads[k].clone()
            w = sd[k]
            this_wd = 0.0 if k in picked else wd
            if this_wd != 0:
                g = g.add(w, alpha=this_wd)
            st = adam_state.setdefault(k, dict(step=0, m=torch.zeros_like(w), v=torch.zeros_like(w)))
            st["step"] += 1
            st["m"].mul_(betas[0]).add_(g, alpha=1 - betas[0])
            st["v"].mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1 = 1 - betas[0] ** st["step"]
            bc2 = 1 - betas[1] ** st["step"]
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            w.addcdiv_(st["m"], denom, value=-lr / bc1)
    for k in pnames:
        sd[k].requires_grad_(False)
        sd[k].grad = None
    return dict(loss_bce=float(bce.detach()), penalty=None if pen is None else float(pen.detach()), grads=grads,
                out=out.detach(), adam_state=adam_state)


# --------------------------------------------------------------------------
# weights fixture (G1): raw little-endian blob + JSON manifest
# --------------------------------------------------------------------------
def load_weights(manifest_path: str) -> Dict[str, torch.Tensor]:
    with open(manifest_path) as f:
        man = json.load(f)
    blob = np.fromfile(os.path.join(os.path.dirname(manifest_path), man["blob"]), dtype=np.uint8)
    sd = {}
    for e in man["tensors"]:
        dt = np.dtype(e["dtype"])
        n = int(np.prod(e["shape"])) if len(e["shape"]) else 1
        arr = np.frombuffer(blob, dtype=dt, count=n, offset=e["offset"]).reshape(e["shape"]).copy()
        sd[e["name"]] = torch.from_numpy(arr)
    return sd


def caller_postprocess(logits: torch.Tensor) -> np.ndarray:
    """test.py:91-96 from the logits onward (no resize): sigmoid -> *255 -> uint8 truncation."""
    p = torch.sigmoid(logits[0].squeeze(0).squeeze(0)).cpu().numpy()
    return (p * 255).astype(np.uint8)


def val_mae(logits: torch.Tensor, targets) -> float:
    """train.py:262-276 for MLOSS == 1: per picture ``(F.interpolate(sigmoid(out[idx])[None], size=(h, w),
    mode='bilinear') * 255.0).int().float() / 255.0`` against its own-size target, ``F.l1_loss(mean)``, averaged over
    the pictures (AverageMeter with n = 1)."""
    sig = torch.sigmoid(logits)
    maes = []
    for idx, t in enumerate(targets):
        h, w = t.shape[-2:]
        r = (F.interpolate(sig[idx].unsqueeze(0), size=(h, w), mode="bilinear") * 255.0).int().float() / 255.0
        maes.append(F.l1_loss(r, t.float().reshape(1, 1, h, w), reduction="mean").item())
    return sum(maes) / len(maes)
