#!/usr/bin/env python3
"""Generate the golden fixtures by importing the REFERENCE implementation (build container only).

TEST INFRASTRUCTURE.  Run from the repo root:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py [--ref /root/reference]

The reference checkout never travels to the GPU box; only the small data fixtures written
here (tests/golden/*, sod100k_amd/data/*) do.  Nothing of the reference's source text is
stored: fixtures are tensors, scalars, key/shape manifests.  Weights are the shipped
checkpoints (CC BY-NC-SA 4.0, (c) the SOD100K authors) re-encoded as a raw little-endian blob
plus a JSON manifest; source md5s are recorded.

Fixture ids follow SURVEY.md section 8(c): G1 weights, G2 logits, G3 per-unit probes,
G4 op micro-goldens, G5 one train step, G6 simplesum + key manifest, G7 DP emulation,
G9 uint8 saliency map.
"""
import argparse
import collections
import collections.abc
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile

import numpy as np

collections.Iterable = collections.abc.Iterable      # shim for reference conv2d.py:15 on py>=3.10

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import torch                                          # noqa: E402
import torch.nn.functional as F                       # noqa: E402

from oracle import inputs as I                        # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "sod100k_amd", "data")


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def encode_checkpoint(ref, name, csnet):
    """G1: checkpoint + layer_config -> <name>.json (manifest) + <name>.bin (raw blob)."""
    base = os.path.join(ref, "CSNet", "checkpoints", name, name)
    lc = csnet.load_layer_config(base + ".bin")
    ck = torch.load(base + ".pth.tar", map_location="cpu")
    sd = ck["state_dict"]
    blob = bytearray()
    tensors = []
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        shape = list(a.shape)                 # taken BEFORE ascontiguousarray, which promotes 0-d arrays to shape [1]
        a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<")))
        while len(blob) % 16:
            blob.append(0)
        tensors.append(dict(name=k, dtype=str(a.dtype), shape=shape, offset=len(blob)))
        blob += a.tobytes()
    lc_json = []
    for e in lc[:-1]:
        lc_json.append([np.asarray(v).astype(np.float64).tolist() if not np.isscalar(v) else [float(v)]
                        for v in e])
    lc_json.append([int(v) for v in lc[-1]])
    man = dict(format="csnet-weights-v1", name=name, blob=name + ".bin", epoch=int(ck["epoch"]),
               arch=str(ck["arch"]), source_md5=dict(checkpoint=md5(base + ".pth.tar"),
                                                     layer_config=md5(base + ".bin")),
               license="CC BY-NC-SA 4.0, weights (c) SOD100K authors (ShangHua-Gao/SOD100K)",
               layer_config=lc_json, tensors=tensors)
    os.makedirs(DATA, exist_ok=True)
    with open(os.path.join(DATA, name + ".bin"), "wb") as f:
        f.write(bytes(blob))
    with open(os.path.join(DATA, name + ".json"), "w") as f:
        json.dump(man, f)
    return lc, sd


def build_ref(csnet, ref, name):
    base = os.path.join(ref, "CSNet", "checkpoints", name, name)
    with quiet():
        m = csnet.build_model(predefine=base + ".bin")
    ck = torch.load(base + ".pth.tar", map_location="cpu")
    m.load_state_dict(ck["state_dict"])
    return m


def unit_probes(csnet, m, x):
    """G3: forward hooks on the 60 units."""
    recs = {}
    hooks = []
    for name, mod in m.named_modules():
        if isinstance(mod, (csnet.gOctaveCBR, csnet.SimplifiedGOctConvBR, csnet.MSBlock)) or name == "cls_layer":
            def hk(_m, _i, out, name=name):
                outs = out if isinstance(out, (list, tuple)) else [out]
                recs[name] = [None if o is None else I.probe(o.detach().numpy()) for o in outs]
            hooks.append(mod.register_forward_hook(hk))
    with torch.no_grad():
        m(x)
    for h in hooks:
        h.remove()
    return recs


def rand_state(mod, rng):
    """Randomise a module's parameters and BN buffers deterministically."""
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            new[k] = v.clone()
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy(rng.uniform(0.3, 2.0, size=tuple(v.shape)).astype(np.float32))
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy((rng.standard_normal(tuple(v.shape)) * 0.5).astype(np.float32))
        elif ".bns." in k or ".bn." in k:
            new[k] = torch.from_numpy(rng.uniform(-1.5, 1.5, size=tuple(v.shape)).astype(np.float32))
        elif "prelu" in k:
            new[k] = torch.from_numpy(rng.uniform(-0.3, 0.6, size=tuple(v.shape)).astype(np.float32))
        else:   # conv weights
            scale = 0.01 if ("convs." in k or "msconv" in k) else 0.4
            new[k] = torch.from_numpy((rng.standard_normal(tuple(v.shape)) * scale).astype(np.float32))
    mod.load_state_dict(new)
    return new


def op_goldens(csnet):
    """G4: op-level micro-goldens on tiny shapes, random weights, eval mode."""
    rng = np.random.default_rng(1234)
    out = {}
    meta = {}
    H = 32
    B = 2

    def mk_inputs(chs, base_h, stride):
        xs = []
        for i, c in enumerate(chs):
            h = (base_h * stride) >> i
            xs.append(torch.from_numpy(rng.standard_normal((B, c, h, h)).astype(np.float32)))
        return xs

    cases = [
        # tag, in split, out split, k, stride
        ("cbr_1to2_k3", [3], [5, 6], 3, 1),
        ("cbr_2to2_k1", [5, 7], [6, 3], 1, 1),
        ("cbr_2to1_k1", [5, 7], [9], 1, 1),
        ("cbr_2to2_k3s2", [5, 4], [7, 6], 3, 2),
        ("cbr_1to2_k3s2", [9], [4, 7], 3, 2),
        ("cbr_3to3_k1", [6, 5, 9], [3, 7, 4], 1, 1),
        ("cbr_3to1_k1", [4, 6, 5], [11], 1, 1),
        ("cbr_2to2_k3", [4, 5], [5, 3], 3, 1),
        ("cbr_1to1_std_k3s2", [6], [8], 3, 2),
        ("cbr_1to1_std_k1", [6], [8], 1, 1),
    ]
    for tag, cin, cout, k, stride in cases:
        ain = (np.array(cin, dtype=np.float64) / sum(cin)).tolist()
        aout = (np.array(cout, dtype=np.float64) / sum(cout)).tolist()
        with quiet():
            mod = csnet.gOctaveCBR(sum(cin), sum(cout), kernel_size=(k, k), padding=1 if k == 3 else 0,
                                   alpha_in=ain, alpha_out=aout, stride=stride)
        sd = rand_state(mod, rng)
        mod.eval()
        xs = mk_inputs(cin, H, stride)
        with torch.no_grad():
            ys = mod([t.clone() for t in xs] if len(xs) > 1 or not mod.std_conv else xs[0].clone())
        ys = ys if isinstance(ys, (list, tuple)) else [ys]
        meta[tag] = dict(kind="cbr", cin=cin, cout=cout, k=k, stride=stride, n_in=len(xs), n_out=len(ys))
        for kk, v in sd.items():
            out[f"{tag}/sd/{kk}"] = v.numpy()
        for i, t in enumerate(xs):
            out[f"{tag}/x{i}"] = t.numpy()
        for j, t in enumerate(ys):
            out[f"{tag}/y{j}"] = t.numpy()

    # depthwise x100 + BN + PReLU, two branches (non-square spatial size)
    tag = "dw_2br"
    with quiet():
        mod = csnet.SimplifiedGOctConvBR(11, 11, alpha=[6 / 11, 5 / 11], groups=11)
    sd = rand_state(mod, rng)
    mod.eval()
    xs = [torch.from_numpy(rng.standard_normal((B, 6, 16, 48)).astype(np.float32)),
          torch.from_numpy(rng.standard_normal((B, 5, 8, 24)).astype(np.float32))]
    with torch.no_grad():
        ys = mod([t.clone() for t in xs])
    meta[tag] = dict(kind="dw", ch=[6, 5])
    for kk, v in sd.items():
        out[f"{tag}/sd/{kk}"] = v.numpy()
    for i, t in enumerate(xs):
        out[f"{tag}/x{i}"] = t.numpy()
    for j, t in enumerate(ys):
        out[f"{tag}/y{j}"] = t.numpy()

    # MSBlock with a zero-channel dilation
    for tag, cin, dil, hw in (("ms_a", 6, [2, 0, 1, 3, 2], (32, 32)), ("ms_b", 5, [0, 2, 2, 0, 3], (16, 48))):
        with quiet():
            mod = csnet.MSBlock(cin, sum(dil), dil)
        sd = rand_state(mod, rng)
        mod.eval()
        x = torch.from_numpy(rng.standard_normal((B, cin, *hw)).astype(np.float32))
        with torch.no_grad():
            y = mod(x.clone())
        meta[tag] = dict(kind="ms", cin=cin, dil=dil)
        for kk, v in sd.items():
            out[f"{tag}/sd/{kk}"] = v.numpy()
        out[f"{tag}/x0"] = x.numpy()
        out[f"{tag}/y0"] = y.numpy()

    # classifier + final bilinear x2 (csnet.py:381-385)
    tag = "cls_up"
    w = torch.from_numpy((rng.standard_normal((1, 9, 1, 1)) * 0.3).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal((1,)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((B, 9, 16, 24)).astype(np.float32))
    y = F.interpolate(F.conv2d(x, w, b), (32, 48), mode="bilinear", align_corners=False)
    meta[tag] = dict(kind="cls")
    out[f"{tag}/w"] = w.numpy(); out[f"{tag}/b"] = b.numpy(); out[f"{tag}/x0"] = x.numpy(); out[f"{tag}/y0"] = y.numpy()
    return out, meta


def train_goldens(csnet, ref, name, shards, expandflop, tag):
    """G5 (shards=1) / G7 (shards=2): one reference training step from the shipped weights."""
    B = 4
    x = torch.from_numpy(I.randn_batch(10, B))
    t = torch.from_numpy(I.binary_target(11, B))
    m = build_ref(csnet, ref, name)
    m.train()
    with quiet():
        if expandflop is None:
            m.flops_hook()
        else:
            m.flops_hook(expandflop=expandflop)
    per = B // shards
    m.set_batchsize(per)                                   # train.py:91 (local batch)
    normal, picked = [], []
    for pname, p in m.named_parameters():                  # train.py:101-107
        if 'stage' in pname and ('conv1x1.bns' in pname or 'conv3x3_1.bns' in pname) and 'weight' in pname:
            picked.append(p)
        else:
            normal.append(p)
    opt = torch.optim.Adam([{'params': normal, 'lr': 1e-4, 'weight_decay': 5e-3},
                            {'params': picked, 'lr': 1e-4, 'weight_decay': 0.}],
                           lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-3)
    names = [n for n, _ in m.named_parameters()]
    acc = None
    bces, pens = [], []
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    bn_after = []
    for s in range(shards):
        if s > 0:   # every DP rank starts from the same weights and buffers
            m.load_state_dict(sd0)
        out = m(x[s * per:(s + 1) * per])
        bce = F.binary_cross_entropy_with_logits(out, t[s * per:(s + 1) * per])
        pen = m.get_flops()
        loss = bce + 3.0 * pen                             # train.py:213, FLOPS.WEIGHT 3.0
        opt.zero_grad()
        loss.backward()
        m.clear_flops()
        g = [p.grad.detach().clone() for p in m.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        bces.append(float(bce)); pens.append(float(pen))
        bn_after.append({k: v.clone() for k, v in m.state_dict().items()
                         if k.endswith(("running_mean", "running_var", "num_batches_tracked"))})
    grads = [a / shards for a in acc]
    for p, g in zip(m.parameters(), grads):
        p.grad = g
    opt.step()
    after = m.state_dict()
    res = dict(tag=tag, B=B, shards=shards, expandflop=expandflop, bce=bces, penalty=pens,
               n_picked=len(picked), n_normal=len(normal))
    res["grad_l2"] = {n: float(g.double().norm()) for n, g in zip(names, grads)}
    res["grad_samples"] = {n: [float(v) for v in g.reshape(-1)[I.probe_indices(g.numel(), 4)]]
                           for n, g in zip(names, grads)}
    res["param_after"] = {n: dict(sum=float(after[n].double().sum()), l2=float(after[n].double().norm()),
                                  samples=[float(v) for v in after[n].reshape(-1)[I.probe_indices(after[n].numel(), 4)]])
                          for n in names}
    # BN buffers after the step of shard 0 (rank-0 view)
    res["bn_after_rank0"] = {k: dict(sum=float(v.double().sum()), l2=float(v.double().norm()))
                             for k, v in bn_after[0].items()}
    res["out_probe"] = None
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    ref = args.ref
    sys.path.insert(0, os.path.join(ref, "CSNet"))
    sys.dont_write_bytecode = True
    from model import csnet                                 # the REFERENCE module (read-only)
    from model.utils.simplesum_octconv import simplesum
    torch.manual_seed(0)
    os.makedirs(GOLD, exist_ok=True)

    # ---- G1 ------------------------------------------------------------
    for name in ("csnet-L-x2", "csnet-L-x1"):
        encode_checkpoint(ref, name, csnet)

    # ---- G6: simplesum pairs + key manifest ------------------------------
    g6 = {}
    for name in ("csnet-L-x2", "csnet-L-x1"):
        base = os.path.join(ref, "CSNet", "checkpoints", name, name)
        with quiet():
            m = csnet.build_model(predefine=base + ".bin")
            params, flops = simplesum(m, inputsize=(3, 224, 224), device=-1)
        sd = m.state_dict()
        g6[name] = dict(params=int(params), flops=int(flops),
                        keys=[[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()])
    for expand, split in ((1.0, [0.5, 0.5]), (2.0, [0.5, 0.5]), (1.0, [1])):
        with quiet():
            m = csnet.build_model(basic_split=split, expand=expand)
            params, flops = simplesum(m, inputsize=(3, 224, 224), device=-1)
        sd = m.state_dict()
        g6[f"init_e{expand}_s{len(split)}"] = dict(
            params=int(params), flops=int(flops), expand=expand, basic_split=split,
            keys=[[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()])
    json.dump(g6, open(os.path.join(GOLD, "g6_simplesum_keys.json"), "w"))

    # ---- G2 / G3 / G9 ----------------------------------------------------
    m = build_ref(csnet, ref, "csnet-L-x2").eval()
    x = torch.from_numpy(I.randn_batch(0, 2))
    with torch.no_grad():
        y = m(x)
    np.save(os.path.join(GOLD, "g2_logits_x2_randn_b2.npy"), y.numpy())
    xi = torch.from_numpy(I.image_like())
    with torch.no_grad():
        yi = m(xi)
    np.save(os.path.join(GOLD, "g2_logits_x2_image.npy"), yi.numpy())
    pred = torch.sigmoid(yi[0].squeeze(0).squeeze(0)).data.cpu().numpy()      # test.py:91-93
    np.save(os.path.join(GOLD, "g9_uint8_x2_image.npy"), (pred * 255).astype(np.uint8))  # test.py:94-96
    xr = torch.from_numpy(I.randn_batch(3, 2, 96, 160))      # non-square, multiples of 16
    with torch.no_grad():
        yr = m(xr)
    np.save(os.path.join(GOLD, "g2_logits_x2_randn_b2_96x160.npy"), yr.numpy())
    json.dump(unit_probes(csnet, m, x), open(os.path.join(GOLD, "g3_unit_probes_x2.json"), "w"))

    m1 = build_ref(csnet, ref, "csnet-L-x1").eval()
    with torch.no_grad():
        y1 = m1(x[:1])
    np.save(os.path.join(GOLD, "g2_logits_x1_randn_b1.npy"), y1.numpy())

    # ---- G4 --------------------------------------------------------------
    arrs, meta = op_goldens(csnet)
    np.savez_compressed(os.path.join(GOLD, "g4_ops.npz"), **arrs)
    json.dump(meta, open(os.path.join(GOLD, "g4_ops_meta.json"), "w"))

    # ---- G5 / G7 ----------------------------------------------------------
    g5 = [train_goldens(csnet, ref, "csnet-L-x2", 1, 1.0, "g5_expand1"),
          train_goldens(csnet, ref, "csnet-L-x2", 1, None, "g5_expand_default2"),
          train_goldens(csnet, ref, "csnet-L-x2", 2, 1.0, "g7_dp2_expand1")]
    json.dump(g5, open(os.path.join(GOLD, "g5_g7_train_step.json"), "w"))
    print("goldens written to", GOLD, "and", DATA)


if __name__ == "__main__":
    main()
