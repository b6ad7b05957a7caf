"""Parity tests proper: the hand-written HIP kernels on an MI355X, driven through the C ABI, against the
committed goldens of the reference and against the CPU oracle.  Tolerance 1e-4 max-abs on the logits
(BASELINE.json north_star), 2e-5 relative per unit."""
import numpy as np
import pytest
import torch

from oracle import csnet_oracle as O, inputs as I

import parity_cases as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from sod100k_amd import _native as N
    assert torch.cuda.is_available(), "no ROCm device"
    return N.load(), torch.device("cuda", 0)


def test_gpu_golden_224(hip, x2_manifest):
    lib, dev = hip
    err = P.check_golden_logits(lib, dev, x2_manifest, "g2_logits_x2_randn_b2.npy", torch.from_numpy(I.randn_batch(0, 2)))
    print("max-abs vs reference golden:", err)


def test_gpu_golden_image_and_uint8(hip, x2_manifest):
    lib, dev = hip
    m, _ = P.make_model(lib, x2_manifest, dev)
    y = m(torch.from_numpy(I.image_like()).to(dev)).cpu()
    g = torch.from_numpy(np.load(P.os.path.join(P.GOLD, "g2_logits_x2_image.npy")))
    assert (y - g).abs().max().item() <= P.TOL
    u8 = O.caller_postprocess(y)
    g9 = np.load(P.os.path.join(P.GOLD, "g9_uint8_x2_image.npy"))
    assert np.abs(u8.astype(int) - g9.astype(int)).max() <= 1 and (u8 != g9).mean() < 5e-3


def test_gpu_golden_nonsquare(hip, x2_manifest):
    lib, dev = hip
    P.check_golden_logits(lib, dev, x2_manifest, "g2_logits_x2_randn_b2_96x160.npy",
                          torch.from_numpy(I.randn_batch(3, 2, 96, 160)))


def test_gpu_x1_config(hip, x1_manifest):
    lib, dev = hip
    P.check_golden_logits(lib, dev, x1_manifest, "g2_logits_x1_randn_b1.npy", torch.from_numpy(I.randn_batch(0, 2))[:1])


def test_gpu_unit_probes(hip, x2_manifest):
    lib, dev = hip
    print("worst unit rel err:", P.check_unit_probes(lib, dev, x2_manifest))


def test_gpu_op_goldens(hip):
    lib, dev = hip
    print(P.check_g4(lib, dev))


@pytest.mark.parametrize("shape", [(3, 16, 16), (2, 32, 48), (1, 224, 224), (5, 64, 64)])
def test_gpu_vs_oracle_shapes(hip, x2_manifest, shape):
    lib, dev = hip
    b, h, w = shape
    P.check_vs_oracle(lib, dev, x2_manifest, torch.from_numpy(I.randn_batch(11, b, h, w)))


def test_gpu_full_size_properties(hip, x2_manifest):
    """BASELINE config 2 size (batch 64): size-independent properties + an oracle check on a sample."""
    lib, dev = hip
    B = 64
    x = torch.from_numpy(I.randn_batch(21, B))
    m, sd = P.make_model(lib, x2_manifest, dev)
    xd = x.to(dev)
    y = m(xd)
    assert y.shape == (B, 1, 224, 224) and torch.isfinite(y).all()
    # determinism
    assert torch.equal(y, m(xd))
    # per-image independence / batch permutation equivariance (eval mode has no cross-image op)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    assert torch.equal(m(xd[perm.to(dev)]), y[perm.to(dev)])
    # Infinity-Cache sized slicing of the batch must not change a single bit
    for sb in (16, 24):
        m2, _ = P.make_model(lib, x2_manifest, dev, sub_batch=sb)
        assert torch.equal(m2(xd), y)
    # oracle on 4 images of the batch
    idx = [0, 17, 40, 63]
    ref = P.oracle_forward(x2_manifest, sd, x[idx])
    assert (y[idx].cpu() - ref).abs().max().item() <= P.TOL


def test_gpu_requires_device_tensor(hip, x2_manifest):
    lib, dev = hip
    m, _ = P.make_model(lib, x2_manifest, dev)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))


def test_gpu_train_forward_vs_oracle(hip, x2_manifest):
    """Train-mode forward (batch-stat BN + running-stat update + dynamic-weight-decay penalty) vs the oracle."""
    lib, dev = hip
    err, pen, pen_ref, worst = P.check_train_forward(lib, dev, x2_manifest, B=4, size=96)
    print(f"train-mode logits max-abs {err:.2e}; penalty {pen:.7f} vs {pen_ref:.7f}; running stats rel {worst:.1e}")


@pytest.mark.parametrize("idx", [0, 1])
def test_gpu_train_forward_golden(hip, x2_manifest, idx):
    """G5: penalty + BN buffers after the reference's own train-mode forward (expandflop 1 and default 2)."""
    lib, dev = hip
    P.check_train_golden(lib, dev, x2_manifest, idx)


def test_gpu_train_step_gradients(hip, x2_manifest):
    """All 419 parameter gradients of one train step vs autograd through the oracle."""
    lib, dev = hip
    worst, loss, pen = P.check_train_step(lib, dev, x2_manifest, B=4, size=96)
    print(f"worst relative gradient error {worst:.2e}; bce {loss:.6f}; penalty {pen:.6f}")


@pytest.mark.parametrize("idx", [0, 1])
def test_gpu_train_step_golden(hip, x2_manifest, idx):
    """G5: the reference's own train step (grad norms, parameters after Adam, BN buffers)."""
    lib, dev = hip
    print("worst grad-norm deviation", P.check_train_golden_step(lib, dev, x2_manifest, idx))


def test_gpu_autograd_seam(hip, x2_manifest):
    lib, dev = hip
    P.check_autograd_seam(lib, dev, x2_manifest, B=2, size=64)
