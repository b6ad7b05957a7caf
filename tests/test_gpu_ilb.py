"""The whole-ILBlock kernel (k_ilb.hip, CSN_OPT_FUSE_ILB; off by default because it measured slower) on the MI355X: DPP wave
shifts, scalar-cache weight streams and the pooled outputs against the reference goldens and the unit-level kernels."""
import os

import numpy as np
import pytest
import torch

from oracle import inputs as I
from sod100k_amd import _native as N

import parity_cases as P

pytestmark = pytest.mark.gpu


def _run(x, manifest, dev, ilb):
    m, sd = P.make_model(N.load(), manifest, dev)
    eng = m.engine_for(x)
    eng.set_option(N.OPT_FUSE_ILB, ilb)
    names = [eng.lib.csn_unit_kernel_name(eng.plan, u).decode() for u in range(eng.n_units)]
    return m(x).cpu(), sd, names


@pytest.mark.parametrize("shape,golden", [((2, 224, 224), "g2_logits_x2_randn_b2.npy"), ((2, 96, 160), "g2_logits_x2_randn_b2_96x160.npy")])
def test_gpu_ilb_goldens(x2_manifest, shape, golden):
    dev = torch.device("cuda", 0)
    seed = 0 if shape[1] == 224 else 3
    x = torch.from_numpy(I.randn_batch(seed, *shape)).to(dev)
    y, _, names = _run(x, x2_manifest, dev, 1)
    assert names.count("ilb_kernel") == 42
    g = torch.from_numpy(np.load(os.path.join(P.GOLD, golden)))
    assert (y - g).abs().max().item() <= P.TOL
    y0, _, names0 = _run(x, x2_manifest, dev, 0)
    assert "ilb_kernel" not in names0 and (y - y0).abs().max().item() <= 2e-5


def test_gpu_ilb_batch64_properties(x2_manifest):
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(I.randn_batch(21, 64)).to(dev)
    y, sd, _ = _run(x, x2_manifest, dev, 56)
    y2, _, _ = _run(x, x2_manifest, dev, 56)
    assert torch.equal(y, y2)
    ref = P.oracle_forward(x2_manifest, sd, x[[0, 33, 63]].cpu())
    assert (y[[0, 33, 63]] - ref).abs().max().item() <= P.TOL
