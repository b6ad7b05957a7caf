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
    # ... nor does the default schedule of this batch size (round 6: two half-batches side by side on the plan's stream lanes)
    # against the whole batch on one stream
    assert m.engine_for(xd).slice_lanes and m.engine_for(xd).sub_batch == 32
    e1 = m.engine_for(xd, slice_lanes=False)
    assert not e1.slice_lanes
    e1.refresh(m._arena.flat)
    assert torch.equal(e1.forward(xd), y)
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


@pytest.mark.parametrize("B,size", [(1, 48), (3, 16), (2, 112)])
def test_gpu_train_step_shapes(hip, x2_manifest, B, size):
    """Edge shapes of the train step: single image, the 16-pixel minimum (1x1 maps on the lowest branch), mid size."""
    lib, dev = hip
    rel = 2e-3 if size >= 48 else 2e-2      # 16x16: BN over 1..4 samples per channel at the deep layers
    worst, loss, pen = P.check_train_step(lib, dev, x2_manifest, B=B, size=size, rel=rel)
    print(f"B={B} {size}x{size}: worst relative gradient error {worst:.2e}")


def test_gpu_train_nonsquare_and_repeat(hip, x2_manifest):
    """Non-square input; two consecutive steps (state carried in the arena: weights, BN buffers, Adam moments)."""
    from sod100k_amd.tools.train import FusedTrainer
    lib, dev = hip
    m, sd = P.make_model(lib, x2_manifest, dev)
    m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
    # eps 1e-3: with the default 1e-8 Adam's first steps move every element by ~lr * sign(g), so elements whose
    # gradient is rounding noise walk in implementation-dependent directions (see check_train_golden_step)
    tr = FusedTrainer(m, lr=1e-4, weight_decay=5e-3, eps=1e-3, flops_weight=3.0, batchsize=2, lib=lib)
    cfg = O.load_layer_config_json(x2_manifest)
    sd_ref = {k: v.clone() for k, v in sd.items()}
    state = None
    for step in range(2):
        x = torch.from_numpy(I.randn_batch(20 + step, 2, 64, 96))
        t = torch.from_numpy(I.binary_target(30 + step, 2, 64, 96))
        loss, pen = tr.step(x.to(dev), t.to(dev))
        m.clear_flops()
        r = O.train_step(cfg, sd_ref, x, t, expandflop=1.0, flops_weight=3.0, batchsize=2, lr=1e-4, wd=5e-3, eps=1e-3,
                         adam_state=state)
        state = r["adam_state"]
        # step 1 starts from parameters that already carry the first step's ~1e-3 relative gradient differences
        tol = 2e-5 if step == 0 else 1e-3
        assert abs(float(loss) - r["loss_bce"]) <= tol * max(1.0, abs(r["loss_bce"])), (step, float(loss), r["loss_bce"])
    got = m.state_dict()
    for k, v in sd_ref.items():
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v), k       # both advanced by 2
        elif k.endswith("running_mean") or k.endswith("running_var"):
            assert ((got[k].cpu() - v).abs() / (1 + v.abs())).max().item() <= 1e-4, k


def test_gpu_pre_post_processing(hip, x2_manifest):
    lib, dev = hip
    P.check_pre_post(lib, dev, x2_manifest)


def test_gpu_std_conv_network(hip):
    """build_model() defaults (basic_split=[1]): Conv2dX100 std_conv units, real stride-2 3x3 -- forward and all gradients."""
    from test_unpruned_emu import _random_state
    lib, dev = hip
    P.check_std_conv_network(lib, dev, _random_state)


def test_gpu_train_step_well_conditioned(hip, x2_manifest):
    """well-conditioned state (gamma in [0.5, 1.5]); judged against fp64 with the fp32 oracle's own
    distance as the yardstick (its noise floor is 3e-3, so an absolute 1e-4 is unreachable for any fp32 implementation)."""
    lib, dev = hip
    # (size 64, the unit-local gate on all eight seeds: round 6 had cut this to size 48 / two seeds for the suite's time -- the time
    # was the oracle on 128 host threads, tests/conftest.py; with the cap the whole test takes seconds)
    print("rel-L2 vs fp64: kernels %.2e, fp32 oracle %.2e" % P.check_train_step_well_conditioned(lib, dev, x2_manifest, B=2, size=64))


def test_gpu_resizes(hip):
    import resize_cases as RC
    lib, dev = hip
    RC.check_resize_bilinear(lib, dev)
    RC.check_pre_post(lib, dev)


@pytest.mark.parametrize("act_dtype,B,size,state", [("fp32", 4, 96, "shipped"), ("fp32", 2, 224, "well"), ("bf16", 4, 96, "shipped"),
                                                    ("bf16", 2, 224, "shipped"), ("bf16", 3, 48, "well"), ("bf16", 2, 16, "shipped"),
                                                    # badly centred depthwise channels (|mean| ~ 10 sigma): parity_cases.off_centre_state
                                                    ("fp32", 2, 96, "offcentre"), ("bf16", 2, 96, "offcentre")])
def test_gpu_train_units_local(hip, x2_manifest, act_dtype, B, size, state):
    """Every unit's train-mode forward and backward (z, activation, dz, dx per consumer slot, every parameter gradient)
    against the oracle applied to the tensors the kernels themselves produced around that unit: no amplification through
    the 60 batch-normalised layers, so the bounds are tight (fp32: 2e-5 forward / 2e-4 backward relative L2; bf16 storage,
    BASELINE config 3: 2e-3 / 3e-2)."""
    lib, dev = hip
    print(P.check_train_units_local(lib, dev, x2_manifest, B=B, size=size, act_dtype=act_dtype, state=state))


@pytest.mark.parametrize("act_dtype,B,size,state", [("fp32", 2, 224, "well"), ("bf16", 4, 96, "shipped"), ("fp32", 3, 48, "shipped")])
def test_gpu_input_gradient_unit_local(hip, x2_manifest, act_dtype, B, size, state):
    """CSN_OPT_INPUT_GRAD (csn_backward's `dx`, autograd's x.grad): the first unit's input gradient inside the
    unit-local bounds, every other unit unchanged."""
    lib, dev = hip
    print(P.check_train_units_local(lib, dev, x2_manifest, B=B, size=size, act_dtype=act_dtype, state=state, input_grad=True))


def test_gpu_train_step_bf16(hip, x2_manifest):
    """bf16 activation storage, whole step: logits / loss / penalty / BN statistics against the oracle with the same storage
    points rounded through bf16 and against the plain fp32 oracle (~1e-2 relative)."""
    lib, dev = hip
    print(P.check_train_step_bf16(lib, dev, x2_manifest, B=4, size=96, state="shipped"))
    print(P.check_train_step_bf16(lib, dev, x2_manifest, B=2, size=64, state="well"))
