#!/usr/bin/env python3
"""Evaluation caller, counterpart of the reference's CSNet/eval.py (47-79) with its ``salmetric`` binary replaced.

    python -m sod100k_amd.tools.eval --save_dir results/csnet-L-x2 --gt_dir datasets/sal --datasets ECSSD --epoch 0

For every ``<save_dir>/<dataset>_<epoch>`` directory of predicted maps and ``<gt_dir>/<dataset>/GT`` it writes the
same ``FmeasureResult_<dataset>_<epoch>.txt`` report the reference gets from ``SalMetric/build/salmetric <list> 8``
(sal_metric.cpp:122-189) and tracks the best Max-F over the epochs (eval.py:71-79).  PNG decoding is host IO (PIL);
histograms / MAE run on the device (csn_sal_hist), images of equal size are batched.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sod100k_amd import _native as N, metric as MT        # noqa: E402


def evaluate_pairs(pairs, device="cuda", lib=None, batch=64):
    """pairs: iterable of (prediction uint8 HxW ndarray, ground truth uint8 HxW ndarray) -> SalMetric"""
    lib = lib if lib is not None else N.load()
    acc = MT.SalMetric()
    pending = {}

    def flush(shape):
        items = pending.pop(shape, [])
        if not items:
            return
        sal = torch.from_numpy(np.stack([a for a, _ in items])).to(device)
        gt = torch.from_numpy(np.stack([b for _, b in items])).to(device)
        hist, abs_sum = MT.sal_hist(lib, sal, gt)
        hist, abs_sum = hist.cpu().numpy(), abs_sum.cpu().numpy()
        for i in range(len(items)):
            acc.add_hist(hist[i], int(abs_sum[i]), shape[0] * shape[1])

    for sal, gt in pairs:
        if sal.shape != gt.shape:
            print("Saliency map should share the same size as ground truth")      # sal_metric.cpp:32-34
            continue
        pending.setdefault(sal.shape, []).append((np.ascontiguousarray(sal), np.ascontiguousarray(gt)))
        if len(pending[sal.shape]) >= batch:
            flush(sal.shape)
    for shape in list(pending):
        flush(shape)
    return acc


def main(argv=None):
    from PIL import Image
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--save_dir", required=True)
    ap.add_argument("--gt_dir", required=True)
    ap.add_argument("--datasets", nargs="+", default=["ECSSD"])
    ap.add_argument("--epoch", type=int, default=0)
    ap.add_argument("--testrange", default="", help="start,end epochs (eval.py:40-45)")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    start, end = (map(int, args.testrange.split(","))) if args.testrange else (args.epoch, args.epoch + 1)
    best_f, best_epoch = 0.0, 0
    for epoch in range(start, end):
        for ds in args.datasets:
            pred_dir = os.path.join(args.save_dir, f"{ds}_{epoch}")
            if not os.path.isdir(pred_dir):
                continue
            gt_dir = os.path.join(args.gt_dir, ds, "GT")
            names = sorted(os.listdir(pred_dir))
            pairs = ((np.asarray(Image.open(os.path.join(pred_dir, n)).convert("L")),
                      np.asarray(Image.open(os.path.join(gt_dir, n)).convert("L"))) for n in names)
            acc = evaluate_pairs(pairs, device=args.device)
            content = acc.report()
            out = os.path.join(args.save_dir, f"FmeasureResult_{ds}_{epoch}.txt")
            with open(out, "w") as f:
                f.write(content)
            results = content.split("\n")[-8:]                                       # eval.py:71
            print(results)
            this_max_f = float(results[0].split()[1])
            if best_f < this_max_f:
                best_f, best_epoch = this_max_f, epoch
            print(out + " eval done.")
    print("BestF: " + str(best_f) + " in Epoch: " + str(best_epoch))


if __name__ == "__main__":
    main()
