"""Shared checks of the CSF+Res2Net head (HIP path or its CPU emulation) against oracle/csf_oracle.py."""
import numpy as np
import torch

from oracle import csf_oracle as CO


def build_csfnet(device="cpu", lib=None, backbone_state=True):
    from sod100k_amd.networks import csf_res2net as R
    net = R.build_model()
    sd = CO.synthetic_state(backbone=True)
    net.load_state_dict(sd, strict=True)
    net = net.to(device).eval()
    if lib is not None:
        object.__setattr__(net, "_lib", lib)
    return net, sd


def head_errors(net, sd, feats, out_size):
    """max |HIP - oracle| of the logits and of every stage, plus the fp64 yardstick of the logits."""
    dev = next(net.parameters()).device
    y = net.head_forward([f.to(dev) for f in feats], out_size).cpu()
    probes = {}
    with torch.no_grad():
        ref = CO.head_forward(sd, feats, out_size, probes=probes)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ref64 = CO.head_forward(sd64, [f.double() for f in feats], out_size)
    eng = list(net._engines.values())[-1]
    errs = {"logits": (y - ref).abs().max().item(), "oracle_vs_fp64": (ref.double() - ref64).abs().max().item(),
            "hip_vs_fp64": (y.double() - ref64).abs().max().item()}
    for st, name in ((0, "fuse"), (1, "ms")):
        for j, t in enumerate(probes[name]):
            errs[f"{name}.{j}"] = (eng.stage(st, j).cpu() - t).abs().max().item()
    return y, ref, errs


def custom_head_error(lib, nb, cin, cmid, sizes, out_size, batch, zero_dil=None):
    """max |HIP - reference composition| of the logits for a descriptor-driven head: any branch count (1..4), any channel
    counts (multiples of the 32 GroupNorm groups for cmid), optionally one empty dilation group (rows moved to d=16)."""
    import torch.nn.functional as F
    from sod100k_amd import _native as N
    from sod100k_amd.networks.csf_res2net import _HeadEngine
    sd = CO.synthetic_state(backbone=False, cin=cin, cmid=cmid)
    # the oracle's head_forward is written for 4 branches with the reference's split rule; emulate fewer branches / empty
    # dilation groups by restating here with the same primitives
    splits = [CO.ms_split(c) for c in cmid]
    if zero_dil is not None:
        j, k = zero_dil
        splits[j][4] += splits[j][k]; splits[j][k] = 0
        w4 = sd[f"ms.convs.{j}.msconv.4.weight"]; wk = sd[f"ms.convs.{j}.msconv.{k}.weight"]
        sd[f"ms.convs.{j}.msconv.4.weight"] = torch.cat([w4, wk], 0)
        del sd[f"ms.convs.{j}.msconv.{k}.weight"]
    offs, chunks, top = {}, [], 0
    for key, v in sd.items():
        offs[key] = top; chunks.append(v.reshape(-1).float()); top += v.numel()
        pad = (-top) % 4
        if pad: chunks.append(torch.zeros(pad)); top += pad
    flat = torch.cat(chunks)
    d = N.CsfHeadDesc(); d.n_branch, d.gn_groups = nb, 32
    d.fuse_w, d.fuse1_w = offs["fuse.conv.weights"], offs["fuse1x1.conv.weights"]
    for j in range(nb):
        d.cin[j], d.cmid[j] = cin[j], cmid[j]
        d.fuse_gn[j] = N.CsfGnOff(offs[f"fuse.bns.{j}.weight"], offs[f"fuse.bns.{j}.bias"], offs[f"fuse.prelus.{j}.weight"])
        d.ms_gn[j] = N.CsfGnOff(offs[f"ms.convs.{j}.bn.weight"], offs[f"ms.convs.{j}.bn.bias"], offs[f"ms.convs.{j}.prelu.weight"])
        for k, co in enumerate(splits[j]):
            d.ms_split[j][k] = co
            d.ms_w[j][k] = offs.get(f"ms.convs.{j}.msconv.{k}.weight", -1)
    d.fuse1_gn = N.CsfGnOff(offs["fuse1x1.bns.0.weight"], offs["fuse1x1.bns.0.bias"], offs["fuse1x1.prelus.0.weight"])
    d.cls_w, d.cls_b = offs["cls_layer.weight"], offs["cls_layer.bias"]
    eng = _HeadEngine(lib, d, batch, tuple(sizes), out_size, torch.device("cpu"))
    eng.refresh(flat)
    feats = CO.synthetic_features(13, batch, sizes, cin=cin)
    y = eng.forward(feats)
    # reference composition
    bi = [0] + list(np.cumsum(cin)); bo = [0] + list(np.cumsum(cmid))
    with torch.no_grad():
        ys = CO.goct_1x1(sd["fuse.conv.weights"], feats, bi, bo)
        ys = [CO.gn_prelu(sd, f"fuse.bns.{j}", f"fuse.prelus.{j}", v) for j, v in enumerate(ys)]
        zs = []
        for j, v in enumerate(ys):
            parts = [F.conv2d(v, sd[f"ms.convs.{j}.msconv.{k}.weight"], None, 1, dl, dl) for k, dl in enumerate(CO.DILATIONS) if splits[j][k] > 0]
            zs.append(CO.gn_prelu(sd, f"ms.convs.{j}.bn", f"ms.convs.{j}.prelu", torch.cat(parts, 1)))
        f = CO.goct_1x1(sd["fuse1x1.conv.weights"], zs, bo, [0, sum(cmid)])[0]
        f = CO.gn_prelu(sd, "fuse1x1.bns.0", "fuse1x1.prelus.0", f)
        ref = F.interpolate(F.conv2d(f, sd["cls_layer.weight"], sd["cls_layer.bias"]), size=out_size, mode="bilinear", align_corners=False)
    return (y - ref).abs().max().item()
