"""CSF+Res2Net (BASELINE config 5) on the MI355X: the HIP decoder head through the C ABI against the
oracle and the G8 goldens of the reference; the backbone runs on PyTorch-ROCm / MIOpen.

Tolerances: head alone (fed the oracle's CPU features) 1e-4 max-abs on the logits; whole network 5e-4 (the backbone's
convolutions are MIOpen's, summation order differs from the CPU reference through 50 layers)."""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import csf_oracle as CO, inputs as I

import csf_cases as K
from conftest import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net_sd():
    from sod100k_amd import _native as N
    assert torch.cuda.is_available(), "no ROCm device"
    N.load()
    return K.build_csfnet(torch.device("cuda", 0))


@pytest.mark.parametrize("sizes,out_size,batch", [
    ([(12, 16), (6, 8), (3, 4), (2, 2)], (48, 64), 2),
    ([(13, 10), (7, 5), (4, 3), (2, 2)], (50, 38), 1),
    ([(44, 44), (22, 22), (11, 11), (6, 6)], (176, 176), 3),
])
def test_gpu_csf_head_vs_oracle(net_sd, sizes, out_size, batch):
    net, sd = net_sd
    y, ref, errs = K.head_errors(net, sd, CO.synthetic_features(5, batch, sizes), out_size)
    print(errs)
    for k, v in errs.items():
        if k.startswith(("fuse.", "ms.")):
            assert v <= 2e-4, (k, v)
    assert errs["logits"] <= 1e-4, errs
    assert errs["hip_vs_fp64"] <= 3 * errs["oracle_vs_fp64"] + 2e-5, errs


@pytest.mark.parametrize("name", ["96x128", "100x76", "352"])
def test_gpu_csf_goldens(net_sd, name):
    net, sd = net_sd
    meta = json.load(open(os.path.join(GOLD, "g8_csf_probes.json")))["cases"][name]
    b, _, h, w = meta["shape"]
    x = torch.from_numpy(I.randn_batch(meta["seed"], b, h, w))
    g = torch.from_numpy(np.load(os.path.join(GOLD, f"g8_csf_logits_{name}.npy")))
    with torch.no_grad():
        feats = CO.res2net_forward(sd, x)                       # oracle backbone on the host
        y_head = net.head_forward([f.cuda() for f in feats], x.shape[2:]).cpu()
        y_full = net(x.cuda()).cpu()
    e_head, e_full = (y_head - g).abs().max().item(), (y_full - g).abs().max().item()
    print(f"{name}: |head(HIP) on oracle features - golden| = {e_head:.3e}   |MIOpen backbone + HIP head - golden| = {e_full:.3e}")
    assert e_head <= 1e-4 and e_full <= 5e-4


def test_gpu_csf_config5_batch32():
    """BASELINE config 5 shape (batch 32, 3 x 352 x 352): batch invariance at full size + timing printout."""
    net, sd = K.build_csfnet(torch.device("cuda", 0))
    x1 = torch.from_numpy(I.randn_batch(80, 1, 352, 352)).cuda()
    x = x1.repeat(32, 1, 1, 1).contiguous()
    g = torch.from_numpy(np.load(os.path.join(GOLD, "g8_csf_logits_352.npy")))
    with torch.no_grad():
        feats = [f.contiguous() for f in net.base(x)]
        y = net.head_forward(feats, x.shape[2:])
        torch.cuda.synchronize()
        assert (y[0:1].cpu() - g).abs().max().item() <= 5e-4
        # every image of the batch is the same picture: the head must give the same answer in every slot
        f1 = [f[:1].contiguous() for f in feats]
        y1 = net.head_forward(f1, x.shape[2:])
        assert torch.equal(y[7], y[0]) and torch.equal(y[31], y[0])
        # batch 1 picks other split-K factors (fewer tiles to fill the chip with): same numbers up to summation order
        assert (y[7] - y1[0]).abs().max().item() <= 2e-5
        eng = [e for e in net._engines.values() if e.batch == 32][0]
        for fn, label in ((lambda: net.head_forward(feats, x.shape[2:]), "head (HIP)"), (lambda: net.base(x), "backbone (MIOpen)")):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            extra = f"  {2 * eng.macs / ms / 1e9:.1f} TFLOP/s fp32" if label.startswith("head") else ""
            print(f"config 5, batch 32: {label} {ms:.2f} ms{extra}")
