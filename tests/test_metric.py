"""Saliency metrics: csn_sal_hist + sod100k_amd/metric.py against the C restatement of SalMetric."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from sod100k_amd import metric as MT
from sod100k_amd.tools.eval import evaluate_pairs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def salm_oracle():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsalmetric_oracle.so"))
    lib.salm_mae.restype = ctypes.c_float
    return lib


def _images(seed, n, h, w):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        yy, xx = np.mgrid[0:h, 0:w]
        blob = ((yy - h * rng.random()) ** 2 + (xx - w * rng.random()) ** 2) < (min(h, w) * (0.2 + 0.3 * rng.random())) ** 2
        gt = (blob * 255).astype(np.uint8)
        if i % 3 == 2:
            gt[::7] = 100                       # some grey (not exactly 0 / 255) ground-truth pixels
        sal = np.clip(blob * 200 + rng.normal(0, 60, (h, w)), 0, 255).astype(np.uint8)
        out.append((sal, gt))
    out.append((np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)))          # empty prediction and ground truth
    out.append((np.full((h, w), 255, np.uint8), out[0][1]))                       # saturated prediction
    return out


def _oracle_metrics(lib, pairs):
    p = np.zeros(256, np.float32); r = np.zeros(256, np.float32)
    maes = []
    per_image = []
    for sal, gt in pairs:
        h, w = sal.shape
        pi = np.zeros(256, np.float32); ri = np.zeros(256, np.float32)
        args = (sal.ctypes.data_as(ctypes.c_void_p), gt.ctypes.data_as(ctypes.c_void_p), h, w)
        lib.salm_precision_recall(*args, pi.ctypes.data_as(ctypes.c_void_p), ri.ctypes.data_as(ctypes.c_void_p))
        maes.append(lib.salm_mae(*args))
        per_image.append((pi, ri))
    return maes, per_image


def check_metrics(lib, device, salm):
    pairs = _images(0, 5, 48, 80) + _images(1, 3, 32, 32)
    maes, per_image = _oracle_metrics(salm, pairs)
    for (sal, gt), mae_ref, (p_ref, r_ref) in zip(pairs, maes, per_image):
        hist, abs_sum = MT.sal_hist(lib, torch.from_numpy(sal[None]).to(device), torch.from_numpy(gt[None]).to(device))
        ref_hist = np.zeros((256, 2), np.int64)
        np.add.at(ref_hist, (sal.reshape(-1), (gt.reshape(-1) > 128).astype(np.int64)), 1)
        assert np.array_equal(hist[0].cpu().numpy(), ref_hist)
        assert int(abs_sum[0]) == int(np.abs(sal.astype(np.int64) - gt.astype(np.int64)).sum())
        mae, p, r = MT.image_metrics(ref_hist, int(abs_sum[0]), sal.size)
        assert np.array_equal(p, p_ref) and np.array_equal(r, r_ref)       # integer counts -> identical float32 ratios
        assert abs(float(mae) - mae_ref) <= 1e-6 * max(1.0, mae_ref)       # the reference accumulates the MAE in float
    acc = evaluate_pairs(pairs, device=device, lib=lib, batch=3)
    s = acc.summary()
    p_mean = np.mean([p for p, _ in per_image], axis=0)
    assert np.abs(s["precision"] - p_mean).max() <= 1e-6
    content = acc.report()
    results = content.split("\n")[-8:]                                     # what eval.py:71-73 does with the output
    assert results[0].startswith("Max_F-measre:") and abs(float(results[0].split()[1]) - float(s["max_f"])) <= 1e-5
    assert content.count("Threshold ") == 256 and results[6].startswith("MAE:")


def test_emu_sal_metric(emu_lib, salm_oracle):
    check_metrics(emu_lib, torch.device("cpu"), salm_oracle)


@pytest.mark.gpu
def test_gpu_sal_metric(salm_oracle):
    from sod100k_amd import _native as N
    check_metrics(N.load(), torch.device("cuda", 0), salm_oracle)


def test_val_mae_matches_training_caller(emu_lib):
    """csn_val_mae (train.py:262-276: sigmoid -> bilinear to the picture's own size -> .int()/255 -> L1) against the
    torch restatement, for up-, down- and same-size targets."""
    import torch
    from oracle import csnet_oracle as O
    from sod100k_amd.engine import val_mae
    g = torch.Generator().manual_seed(4)
    logits = 3.0 * torch.randn(4, 1, 32, 48, generator=g)
    targets = [(torch.rand(50, 70, generator=g) > 0.5).float(), (torch.rand(32, 48, generator=g) > 0.5).float(),
               torch.rand(17, 23, generator=g), (torch.rand(96, 31, generator=g) > 0.3).float()]
    total = torch.zeros(1, dtype=torch.float64)
    for i, t in enumerate(targets):
        val_mae(emu_lib, logits[i], t, out=total)
    ref = O.val_mae(logits, targets)
    assert abs(float(total) / 4 - ref) <= 2e-6, (float(total) / 4, ref)


@pytest.mark.gpu
def test_gpu_val_mae():
    import torch
    from oracle import csnet_oracle as O
    from sod100k_amd import _native as N
    from sod100k_amd.engine import val_mae
    lib, dev = N.load(), torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    logits = 3.0 * torch.randn(3, 1, 224, 224, generator=g)
    targets = [(torch.rand(300, 400, generator=g) > 0.5).float(), (torch.rand(224, 224, generator=g) > 0.5).float(),
               torch.rand(120, 90, generator=g)]
    total = torch.zeros(1, dtype=torch.float64, device=dev)
    for i, t in enumerate(targets):
        val_mae(lib, logits[i].to(dev), t.to(dev), out=total)
    assert abs(float(total) / 3 - O.val_mae(logits, targets)) <= 2e-6


def test_val_loop_of_the_training_caller(emu_lib, x2_manifest):
    """sod100k_amd/tools/train.py:val (train.py:250-293) on the emulated kernels against the oracle's forward + val_mae."""
    import torch
    from oracle import csnet_oracle as O, inputs as I
    from sod100k_amd.model import csnet as M
    from sod100k_amd.tools import train as T
    sd = O.load_weights(x2_manifest)
    m = M.build_model(predefine=x2_manifest)
    m.load_state_dict(sd)
    m._lib = emu_lib
    x = torch.from_numpy(I.randn_batch(12, 2, 64, 64))
    g = torch.Generator().manual_seed(6)
    targets = [(torch.rand(70, 50, generator=g) > 0.5).float(), (torch.rand(64, 64, generator=g) > 0.5).float()]
    got = T.val(m, [(x, targets)], lib=emu_lib)
    with torch.no_grad():
        ref = O.val_mae(O.csnet_forward(O.load_layer_config_json(x2_manifest), sd, x), targets)
    assert abs(got - ref) <= 1e-5, (got, ref)
