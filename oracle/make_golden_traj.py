#!/usr/bin/env python3
"""G11: the BCE trajectory of the REFERENCE's training loop on fixed synthetic data (build container only).

TEST INFRASTRUCTURE, run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_traj.py [--ref /root/reference]

The reference model (CSNet/model/csnet.py, shipped csnet-L-x2 weights) is driven exactly as CSNet_training/train.py:101-123,203-216
drives it -- Adam with the two parameter groups, loss = BCE + FLOPS.WEIGHT * get_flops() -- for STEPS iterations on ONE batch of 8
synthetic 64x64 pictures; the fixture holds the BCE / penalty of every step and the hyper-parameters (no source text).
The GPU test (tests/test_gpu_paths.py::test_gpu_loss_goes_down) replays the same loop on the HIP kernels."""
import argparse
import collections
import collections.abc
import json
import os
import sys

collections.Iterable = collections.abc.Iterable      # shim for reference conv2d.py:15 on py>=3.10
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import torch                                          # noqa: E402
import torch.nn.functional as F                       # noqa: E402

from oracle import inputs as I                        # noqa: E402
import make_goldens as G                              # noqa: E402

B, S, STEPS, LR, WD, EPS, FLOPS_W = 8, 64, 40, 1e-3, 5e-3, 1e-3, 3.0


def data():
    x = torch.from_numpy(I.randn_batch(90, B, S, S))
    t = (F.avg_pool2d(x[:, :1], 9, 1, 4) > 0).float()      # a learnable target: a blurred threshold of the first channel
    return x, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(args.ref, "CSNet"))
    sys.dont_write_bytecode = True
    from model import csnet                                 # the REFERENCE module (read-only)
    torch.manual_seed(0)
    m = G.build_ref(csnet, args.ref, "csnet-L-x2")
    m.train()
    with G.quiet():
        m.flops_hook(expandflop=1.0)
    m.set_batchsize(B)
    normal, picked = [], []
    for pname, p in m.named_parameters():                  # train.py:101-107
        if 'stage' in pname and ('conv1x1.bns' in pname or 'conv3x3_1.bns' in pname) and 'weight' in pname:
            picked.append(p)
        else:
            normal.append(p)
    opt = torch.optim.Adam([{'params': normal, 'lr': LR, 'weight_decay': WD}, {'params': picked, 'lr': LR, 'weight_decay': 0.}],
                           lr=LR, betas=(0.9, 0.99), eps=EPS, weight_decay=WD)
    x, t = data()
    bce, pen = [], []
    for _ in range(STEPS):
        out = m(x)
        lb = F.binary_cross_entropy_with_logits(out, t)
        lp = m.get_flops()
        opt.zero_grad()
        (lb + FLOPS_W * lp).backward()
        opt.step()
        m.clear_flops()
        bce.append(float(lb)); pen.append(float(lp))
    json.dump(dict(B=B, S=S, steps=STEPS, lr=LR, weight_decay=WD, eps=EPS, betas=[0.9, 0.99], flops_weight=FLOPS_W, expandflop=1.0,
                   input="oracle.inputs.randn_batch(90, 8, 64, 64)", target="avg_pool2d(x[:, :1], 9, 1, 4) > 0",
                   checkpoint="csnet-L-x2", bce=bce, penalty=pen),
              open(os.path.join(G.GOLD, "g11_train_trajectory_x2.json"), "w"), indent=1)
    print("bce", " ".join(f"{v:.4f}" for v in bce))
    print("pen", " ".join(f"{v:.4f}" for v in pen))


if __name__ == "__main__":
    main()
