"""Saliency evaluation metrics from the device histograms of ``csn_sal_hist``.

Counterpart of the reference's only native component, ``SalMetric`` (CSNet_training/SalMetric/src/sal_metric.cpp):
per image, precision / recall at the 256 thresholds and the MAE; over a dataset, their means, the F-measure curve
(beta^2 = 0.3), Max-F / Mean-F and the report text that ``eval.py`` parses (eval.py:71-73: the last 8 lines, first of
them ``Max_F-measre:   <value>``).  The reference scans every image 256 times; here one joint histogram per image
(value x binarised ground truth) gives all thresholds at once:

    a_sum(th) = #{sal > th},  ab(th) = #{sal > th and gt > 128},  b_sum = #{gt > 128}
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
import torch

from . import _native as N

THRESHOLDS = 256
EPSILON = np.float32(1e-4)     # sal_metric.hpp:51
BETA = np.float32(0.3)         # sal_metric.hpp:52


def sal_hist(lib: C.CDLL, sal: torch.Tensor, gt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """uint8 maps [N, H, W] (same size) -> (hist [N, 256, 2] int64, abs_sum [N] int64) on the device."""
    assert sal.dtype == torch.uint8 and gt.dtype == torch.uint8 and sal.shape == gt.shape and sal.dim() == 3
    sal, gt = sal.contiguous(), gt.contiguous()
    n, h, w = sal.shape
    hist = torch.zeros((n, 256, 2), dtype=torch.int64, device=sal.device)
    abs_sum = torch.zeros((n,), dtype=torch.int64, device=sal.device)
    stream = torch.cuda.current_stream(sal.device).cuda_stream if sal.is_cuda else 0
    N.check(lib, lib.csn_sal_hist(sal.data_ptr(), gt.data_ptr(), h * w, n, hist.data_ptr(), abs_sum.data_ptr(), stream),
            "csn_sal_hist")
    return hist, abs_sum


def image_metrics(hist: np.ndarray, abs_sum: int, npix: int) -> Tuple[np.float32, np.ndarray, np.ndarray]:
    """(mae, precision[256], recall[256]) of one image, float32 like the reference (sal_metric.cpp:87-120)."""
    hist = np.asarray(hist, dtype=np.int64).reshape(256, 2)
    tot = hist.sum(axis=1)
    # counts of values strictly above th: reversed cumulative sums
    above_all = np.concatenate([np.cumsum(tot[::-1])[::-1][1:], [0]]).astype(np.float32)
    above_fg = np.concatenate([np.cumsum(hist[::-1, 1])[::-1][1:], [0]]).astype(np.float32)
    b_sum = np.float32(hist[:, 1].sum())
    precision = (above_fg + EPSILON) / (above_all + EPSILON)
    recall = (above_fg + EPSILON) / (b_sum + EPSILON)
    mae = np.float32(np.float64(abs_sum) / 255.0 / npix)
    return mae, precision.astype(np.float32), recall.astype(np.float32)


class SalMetric:
    """Dataset accumulator with the reference's report (do_evaluation, sal_metric.cpp:122-189)."""

    def __init__(self):
        self.n = 0
        self.mae = np.float64(0)
        self.precision = np.zeros(THRESHOLDS, np.float64)
        self.recall = np.zeros(THRESHOLDS, np.float64)

    def add(self, mae, precision, recall) -> None:
        self.n += 1
        self.mae += np.float64(mae)
        self.precision += precision.astype(np.float64)
        self.recall += recall.astype(np.float64)

    def add_hist(self, hist: np.ndarray, abs_sum: int, npix: int) -> None:
        self.add(*image_metrics(hist, abs_sum, npix))

    def summary(self):
        n = max(self.n, 1)
        p = (self.precision / n).astype(np.float32)
        r = (self.recall / n).astype(np.float32)
        mae = np.float32(self.mae / n)
        f = ((np.float32(1) + BETA) * p * r) / (BETA * p + r)
        best = int(np.argmax(f))            # first maximum, like the reference's strict '>' scan
        return dict(mae=mae, precision=p, recall=r, fmeasure=f, argmax=best, max_f=f[best], mean_f=np.float32(f.mean()),
                    mean_precision=np.float32(p.mean()), mean_recall=np.float32(r.mean()))

    def report(self, num_threads: int = 8) -> str:
        """The text ``salmetric`` prints (cout's default 6 significant digits = ``%g``)."""
        s = self.summary()
        g = lambda v: "%g" % float(v)
        lines: List[str] = [f"{num_threads} threads are being used for accelerating."]
        for th in range(THRESHOLDS):
            lines.append(f"Threshold {th}:\tMAE: {g(s['mae'])}\tPrecision: {g(s['precision'][th])}"
                         f"\tRecall: {g(s['recall'][th])}\tFmeasure: {g(s['fmeasure'][th])}")
        lines += [f"Max_F-measre:   {g(s['max_f'])}", f"Mean_F-measre:  {g(s['mean_f'])}",
                  f"Precision:      {g(s['precision'][s['argmax']])}", f"Recall:         {g(s['recall'][s['argmax']])}",
                  f"Mean_Precision: {g(s['mean_precision'])}", f"Mean_Recall:    {g(s['mean_recall'])}",
                  f"MAE:            {g(s['mae'])}"]
        return "\n".join(lines) + "\n"
