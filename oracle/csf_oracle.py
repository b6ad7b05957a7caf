"""CPU oracle for the CSF+Res2Net path (BASELINE config 5)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline leg may import this module (same rule as
oracle/csnet_oracle.py).  Functional restatement over a flat ``state_dict`` (plain torch CPU ops, no nn.Module) of

    CSFNet.forward            CSF+Res2Net/networks/csf_res2net.py:248-256
    Res2Net.forward           csf_res2net.py:153-169   (v1b stem 111-119, Bottle2neck 26-103, downsample 136-144)
    gOctaveConv.forward       CSF+Res2Net/networks/gOctConv.py:60-114  (parameter is called ``weights``; hi->lo is a
                              bilinear resize BEFORE the conv (99-101), lo->hi a bilinear resize AFTER it (96-98);
                              sizes are taken from input branch j (84))
    gOctaveCBR.forward        gOctConv.py:141-152      (GroupNorm(32, C) + PReLU(C) per output branch, 127-130)
    PallMSBlock / MSBlock     csf_res2net.py:174-223   (five dense dilated 3x3 convs, C//5 channels each, the rest on
                              d=16; GroupNorm(32) + PReLU)

The primitive arithmetic is PyTorch ATen (not vendored under the reference); the composition logic is what is restated.

Pinning: the reference ships neither weights nor tests for this network.  ``oracle/make_golden_csf.py`` loads the
deterministic synthetic state of ``synthetic_state()`` below into the REFERENCE modules (imported in the build
container) and freezes its outputs under tests/golden/g8_*; tests/test_oracle_vs_golden.py checks this file against
them on CPU.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)                 # csf_res2net.py:232
PLANES = (64, 128, 256, 512)
BASE_WIDTH, SCALE, EXPANSION = 26, 4, 4
DILATIONS = (1, 2, 4, 8, 16)          # csf_res2net.py:194
GN_GROUPS, GN_EPS, BN_EPS = 32, 1e-5, 1e-5
FUSE_IN = (256, 512, 1024, 2048)      # csf_res2net.py:235-236
FUSE_OUT = (128, 256, 512, 512)       # csf_res2net.py:237-238


def bounds(total: int, alpha: Sequence[float]) -> List[int]:
    """gOctConv.py:33-42,78-83: running python-float sums of alpha, int(round(C * a))."""
    acc, out = 0, [0]
    for a in alpha:
        acc += a
        out.append(int(round(total * acc)))
    return out


def head_channels(cin=FUSE_IN, cmid=FUSE_OUT):
    """Channel boundaries exactly as the constructors derive them (csf_res2net.py:235-245)."""
    tin, tmid = sum(cin), sum(cmid)
    a_in = [c / tin for c in cin]
    a_mid = [c / tmid for c in cmid]
    return dict(bi=bounds(tin, a_in), bo=bounds(tmid, a_mid),
                cmid=[int(round(tmid * a)) for a in a_mid])


def ms_split(c: int) -> List[int]:
    each = c // 5                                          # csf_res2net.py:196-202
    return [each] * 4 + [c - 4 * each]


# --------------------------------------------------------------------------------------------------------------
# deterministic synthetic parameters (no shipped checkpoint exists for this network)
# --------------------------------------------------------------------------------------------------------------
def _rng(key: str) -> np.random.Generator:
    return np.random.default_rng(zlib.crc32(key.encode()))


def _conv(key, co, ci, k, gain=1.0):
    std = gain * math.sqrt(2.0 / (ci * k * k))
    return torch.from_numpy(_rng(key).standard_normal((co, ci, k, k), dtype=np.float32) * np.float32(std))


def _bn(sd, key, c, wscale=1.0):
    r = _rng(key)
    sd[key + ".weight"] = torch.from_numpy((wscale * r.uniform(0.6, 1.2, c)).astype(np.float32))
    sd[key + ".bias"] = torch.from_numpy((0.1 * r.standard_normal(c)).astype(np.float32))
    sd[key + ".running_mean"] = torch.from_numpy((0.1 * r.standard_normal(c)).astype(np.float32))
    sd[key + ".running_var"] = torch.from_numpy(r.uniform(0.6, 1.4, c).astype(np.float32))
    sd[key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def _gn_prelu(sd, gn_key, prelu_key, c):
    r = _rng(gn_key)
    sd[gn_key + ".weight"] = torch.from_numpy(r.uniform(0.6, 1.4, c).astype(np.float32))
    sd[gn_key + ".bias"] = torch.from_numpy((0.2 * r.standard_normal(c)).astype(np.float32))
    sd[prelu_key + ".weight"] = torch.from_numpy(r.uniform(0.05, 0.45, c).astype(np.float32))


def backbone_state(sd: Dict[str, torch.Tensor]) -> None:
    sd["base.conv1.0.weight"] = _conv("base.conv1.0.weight", 32, 3, 3)
    _bn(sd, "base.conv1.1", 32)
    sd["base.conv1.3.weight"] = _conv("base.conv1.3.weight", 32, 32, 3)
    _bn(sd, "base.conv1.4", 32)
    sd["base.conv1.6.weight"] = _conv("base.conv1.6.weight", 64, 32, 3)
    _bn(sd, "base.bn1", 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip(PLANES, LAYERS)):
        width = int(math.floor(planes * (BASE_WIDTH / 64.0)))
        for b in range(nblk):
            p = f"base.layer{li + 1}.{b}"
            sd[p + ".conv1.weight"] = _conv(p + ".conv1.weight", width * SCALE, inplanes, 1)
            _bn(sd, p + ".bn1", width * SCALE)
            for i in range(SCALE - 1):
                sd[f"{p}.convs.{i}.weight"] = _conv(f"{p}.convs.{i}.weight", width, width, 3)
                _bn(sd, f"{p}.bns.{i}", width)
            sd[p + ".conv3.weight"] = _conv(p + ".conv3.weight", planes * EXPANSION, width * SCALE, 1)
            _bn(sd, p + ".bn3", planes * EXPANSION, wscale=0.5)       # keeps the residual sum from growing
            if b == 0:
                sd[p + ".downsample.1.weight"] = _conv(p + ".downsample.1.weight", planes * EXPANSION, inplanes, 1)
                _bn(sd, p + ".downsample.2", planes * EXPANSION, wscale=0.7)
            inplanes = planes * EXPANSION


def head_state(sd: Dict[str, torch.Tensor], cin=FUSE_IN, cmid=FUSE_OUT) -> None:
    ch = head_channels(cin, cmid)
    tin, tmid = sum(cin), sum(cmid)
    sd["fuse.conv.weights"] = _conv("fuse.conv.weights", tmid, tin, 1, gain=0.7)
    for j, c in enumerate(ch["cmid"]):
        _gn_prelu(sd, f"fuse.bns.{j}", f"fuse.prelus.{j}", c)
    for j, c in enumerate(ch["cmid"]):
        for d, cd in enumerate(ms_split(c)):
            sd[f"ms.convs.{j}.msconv.{d}.weight"] = _conv(f"ms.convs.{j}.msconv.{d}.weight", cd, c, 3)
        _gn_prelu(sd, f"ms.convs.{j}.bn", f"ms.convs.{j}.prelu", c)
    sd["fuse1x1.conv.weights"] = _conv("fuse1x1.conv.weights", tmid, tmid, 1, gain=0.7)
    _gn_prelu(sd, "fuse1x1.bns.0", "fuse1x1.prelus.0", tmid)
    r = _rng("cls_layer")
    sd["cls_layer.weight"] = torch.from_numpy((0.05 * r.standard_normal((1, tmid, 1, 1))).astype(np.float32))
    sd["cls_layer.bias"] = torch.from_numpy(np.array([0.1], dtype=np.float32))


def synthetic_state(backbone: bool = True, cin=FUSE_IN, cmid=FUSE_OUT) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    if backbone:
        backbone_state(sd)
    head_state(sd, cin, cmid)
    return sd


def synthetic_features(seed: int, b: int, sizes: Sequence[Sequence[int]], cin=FUSE_IN) -> List[torch.Tensor]:
    """ReLU-like (non-negative, sparse) feature maps for head-only checks."""
    out = []
    for i, (c, (h, w)) in enumerate(zip(cin, sizes)):
        a = np.random.default_rng(seed * 16 + i).standard_normal((b, c, h, w), dtype=np.float32)
        out.append(torch.from_numpy(np.maximum(a, 0.0)))
    return out


# --------------------------------------------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------------------------------------------
def _bn_eval(sd, key, x):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        False, 0.1, BN_EPS)


def bottle2neck(sd, p, x, stride, first):
    """csf_res2net.py:69-103 (stype 'stage' for the first block of a layer, 'normal' otherwise)."""
    out = F.relu(_bn_eval(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    width = out.shape[1] // SCALE
    spx = torch.split(out, width, 1)
    outs, sp = [], None
    for i in range(SCALE - 1):
        sp = spx[i] if (i == 0 or first) else sp + spx[i]
        sp = F.relu(_bn_eval(sd, f"{p}.bns.{i}", F.conv2d(sp, sd[f"{p}.convs.{i}.weight"], None, stride, 1)))
        outs.append(sp)
    outs.append(F.avg_pool2d(spx[SCALE - 1], 3, stride, 1) if first else spx[SCALE - 1])
    out = _bn_eval(sd, p + ".bn3", F.conv2d(torch.cat(outs, 1), sd[p + ".conv3.weight"]))
    if first:
        r = F.avg_pool2d(x, stride, stride, ceil_mode=True, count_include_pad=False)
        r = _bn_eval(sd, p + ".downsample.2", F.conv2d(r, sd[p + ".downsample.1.weight"]))
    else:
        r = x
    return F.relu(out + r)


def res2net_forward(sd, x) -> List[torch.Tensor]:
    x = F.relu(_bn_eval(sd, "base.conv1.1", F.conv2d(x, sd["base.conv1.0.weight"], None, 2, 1)))
    x = F.relu(_bn_eval(sd, "base.conv1.4", F.conv2d(x, sd["base.conv1.3.weight"], None, 1, 1)))
    x = F.relu(_bn_eval(sd, "base.bn1", F.conv2d(x, sd["base.conv1.6.weight"], None, 1, 1)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nblk in enumerate(LAYERS):
        for b in range(nblk):
            x = bottle2neck(sd, f"base.layer{li + 1}.{b}", x, 2 if (b == 0 and li > 0) else 1, b == 0)
        feats.append(x)
    return feats


# --------------------------------------------------------------------------------------------------------------
# head
# --------------------------------------------------------------------------------------------------------------
def goct_1x1(w, xs, bi, bo):
    """gOctConv.py:60-114 for kernel 1x1 / stride 1 / groups 1 / no bias."""
    nin, nout = len(bi) - 1, len(bo) - 1
    ys: List[list] = [[] for _ in range(nout)]
    for i in range(nin):
        x = xs[i]
        for j in range(nout):
            wij = w[bo[j]:bo[j + 1], bi[i]:bi[i + 1]]
            size = xs[j].shape[2:4]
            if i > j:
                y = F.interpolate(F.conv2d(x, wij), size=size, mode="bilinear", align_corners=False)
            elif i < j:
                y = F.conv2d(F.interpolate(x, size=size, mode="bilinear", align_corners=False), wij)
            else:
                y = F.conv2d(x, wij)
            ys[j].append(y)
    return [sum(v) for v in ys]


def gn_prelu(sd, gn_key, prelu_key, x):
    return F.prelu(F.group_norm(x, GN_GROUPS, sd[gn_key + ".weight"], sd[gn_key + ".bias"], GN_EPS),
                   sd[prelu_key + ".weight"])


def head_forward(sd, feats, out_size, cin=FUSE_IN, cmid=FUSE_OUT, probes=None):
    ch = head_channels(cin, cmid)
    ys = goct_1x1(sd["fuse.conv.weights"], feats, ch["bi"], ch["bo"])
    ys = [gn_prelu(sd, f"fuse.bns.{j}", f"fuse.prelus.{j}", y) for j, y in enumerate(ys)]
    if probes is not None:
        probes["fuse"] = ys
    zs = []
    for j, y in enumerate(ys):
        parts = [F.conv2d(y, sd[f"ms.convs.{j}.msconv.{d}.weight"], None, 1, dil, dil)
                 for d, dil in enumerate(DILATIONS)]
        zs.append(gn_prelu(sd, f"ms.convs.{j}.bn", f"ms.convs.{j}.prelu", torch.cat(parts, 1)))
    if probes is not None:
        probes["ms"] = zs
    f = goct_1x1(sd["fuse1x1.conv.weights"], zs, ch["bo"], [0, sum(cmid)])[0]
    f = gn_prelu(sd, "fuse1x1.bns.0", "fuse1x1.prelus.0", f)
    if probes is not None:
        probes["fuse1x1"] = [f]
    o = F.conv2d(f, sd["cls_layer.weight"], sd["cls_layer.bias"])
    return F.interpolate(o, size=tuple(out_size), mode="bilinear", align_corners=False)


def csfnet_forward(sd, x, probes=None):
    feats = res2net_forward(sd, x)
    if probes is not None:
        probes["features"] = feats
    return head_forward(sd, feats, x.shape[2:], probes=probes)
