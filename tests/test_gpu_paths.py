"""Paths the shipped csnet-L-x2 configuration does not take on its own, on the MI355X:
the un-pruned training network of csnet-L-x2_train.yml, the pruned slim network (zero-channel branch), the opt-in stream-lane
modes (real concurrency: a race shows up as run-to-run differences), and an optimisation run whose loss must go down."""
import os

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


def test_gpu_unpruned_expand2_eval_and_train_step(hip):
    """expand 2.0 / basic_split [0.5, 0.5] (CSNet_training/configs/csnet-L-x2_train.yml:9-18, 788,631 parameters): weight images
    beyond the LDS budget take the M-group / row-chunk paths the shipped x2 net never touches."""
    lib, dev = hip
    err, mine, yard = P.check_unpruned(lib, dev, expand=2.0, width=40, B=2, size=64, seed=4)
    print(f"unpruned x2: eval max-abs {err:.2e}; gradient median rel err {mine:.2e} (fp32 oracle vs fp64: {yard:.2e})")


def test_gpu_unpruned_expand2_train_step_bf16(hip):
    lib, dev = hip
    err, mine, yard = P.check_unpruned(lib, dev, expand=2.0, width=40, B=2, size=64, seed=4, act_dtype="bf16")
    print(f"unpruned x2 bf16 storage: gradient median rel err {mine:.2e}")


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_gpu_unpruned_expand2_train_units_local(hip, x2_manifest, act_dtype):
    """Every unit of the un-pruned expand-2 net (40-160 channels per branch: several M groups / row chunks per launch) judged
    LOCALLY on the device's own tensors, fp32 and bf16 storage -- the sharp gate check_unpruned's whole-step yardstick is not."""
    lib, dev = hip
    net = P.unpruned_network(2.0, 40, seed=4)
    worst = P.check_train_units_local(lib, dev, x2_manifest, B=2, size=64, act_dtype=act_dtype, net=net, input_grad=(act_dtype == "bf16"))
    print(f"unpruned x2 unit-local [{act_dtype}] worst relative L2 per kind: {worst}")


def test_gpu_unpruned_expand1(hip):
    lib, dev = hip
    P.check_unpruned(lib, dev, expand=1.0, width=20, B=2, size=64, seed=0)


def test_gpu_slim_network_forward(hip, x2_manifest, tmp_path):
    """G10's network (prune threshold 0.01: one output branch has zero channels) on the device."""
    lib, dev = hip
    err = P.check_slim_network(lib, dev, x2_manifest, tmp_path)
    print(f"slim network: max-abs {err:.2e}")
    P.check_slim_network(lib, dev, x2_manifest, tmp_path, B=3, H=224, W=224)


@pytest.mark.gpu
@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_gpu_slim_network_train_units_local(hip, x2_manifest, tmp_path, act_dtype):
    """... and its train step (the reference finetunes exactly this network): every unit's forward and backward on the device inside
    the unit-local bounds (zero-channel branches / dilations, odd channel counts)."""
    lib, dev = hip
    net = P.slim_network(x2_manifest, tmp_path)
    print(P.check_train_units_local(lib, dev, x2_manifest, B=2, size=96, act_dtype=act_dtype, net=net))


def _train_grad(lib, dev, manifest, x, t, B):
    from sod100k_amd.tools.train import FusedTrainer
    m, _ = P.make_model(lib, manifest, dev)
    m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(1.0)
    tr = FusedTrainer(m, lr=0.0, weight_decay=0.0, flops_weight=3.0, batchsize=B, lib=lib)
    outs = []
    for _ in range(20):                       # eager, capture, replays: the parameters do not move (lr 0)
        loss, pen = tr.step(x, t)
        outs.append((tr.grad.clone(), float(loss), float(pen)))
    torch.cuda.synchronize()
    return outs


def test_gpu_overlap2_weight_gradient_lane_is_deterministic(hip, x2_manifest, monkeypatch):
    """CSN_OPT_OVERLAP = 2 moves every weight-gradient launch of csn_backward to a side stream (the mode ADVICE r2 found a race
    in): 20 repeated steps must be bit-identical to each other, and equal to the default mode's gradient up to summation order."""
    lib, dev = hip
    B, S = 4, 96
    x = torch.from_numpy(I.randn_batch(70, B, S, S)).to(dev)
    t = torch.from_numpy(I.binary_target(71, B, S, S)).to(dev)
    base = _train_grad(lib, dev, x2_manifest, x, t, B)
    monkeypatch.setenv("CSN_OVERLAP", "2")
    lane = _train_grad(lib, dev, x2_manifest, x, t, B)
    g0, l0 = base[0][0], lane[0][0]
    for k, (g, loss, pen) in enumerate(base):
        assert torch.equal(g, g0), f"default mode: step {k} differs from step 0"
    # the side-lane mode takes other kernels for some passes (the depthwise units' weight gradients on their own kernel, the
    # adjoint upsampling as its own pass): the same gradient up to fp32 summation order -- and bit-identical from step to step
    # (a race between the lanes shows up as a run-to-run difference, as it did on the first graph replay in round 4)
    for k, (g, loss, pen) in enumerate(lane):
        assert torch.equal(g, l0), f"overlap 2: step {k} differs from step 0 ({(g - l0).abs().max().item():.3e})"
        assert loss == base[0][1] and abs(pen - base[0][2]) <= 1e-6 * max(1.0, abs(base[0][2]))
    rel = float((l0 - g0).norm() / g0.norm())
    assert rel <= 1e-4, f"overlap 2 vs default mode: relative L2 {rel:.3e}"


def test_gpu_slice_lanes_is_deterministic(hip, x2_manifest, monkeypatch):
    """CSN_SLICE_LANES = 1: two half-batches side by side on the plan's stream lanes; 20 forwards bit-identical to the default."""
    lib, dev = hip
    x = torch.randn(16, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(dev)
    m, _ = P.make_model(lib, x2_manifest, dev)
    y0 = m(x).clone()
    monkeypatch.setenv("CSN_SLICE_LANES", "1")
    m2, _ = P.make_model(lib, x2_manifest, dev)
    y = torch.empty_like(y0)
    eng = m2.engine_for(x)
    eng.refresh(m2._arena.flat)
    for k in range(20):
        y.fill_(float("nan"))
        eng.forward(x, out=y)
        assert torch.equal(y, y0), f"slice lanes: forward {k} differs ({(y - y0).abs().max().item():.3e})"


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_gpu_loss_goes_down(hip, x2_manifest, act_dtype):
    """The reference's training loop on fixed synthetic data, replayed on the device: 40 optimiser steps at lr 1e-3 on 8 pictures
    (train.py:101-123,203-216: Adam with the two parameter groups, loss = BCE + 3.0 * get_flops(), shipped x2 weights as the start)
    against golden G11 -- the BCE of every step of the REFERENCE itself (oracle/make_golden_traj.py)."""
    import json
    from sod100k_amd.tools.train import FusedTrainer
    lib, dev = hip
    g = json.load(open(os.path.join(P.GOLD, "g11_train_trajectory_x2.json")))
    B, S, steps = g["B"], g["S"], g["steps"]
    x = torch.from_numpy(I.randn_batch(90, B, S, S))
    t = (torch.nn.functional.avg_pool2d(x[:, :1], 9, 1, 4) > 0).float()
    m, sd = P.make_model(lib, x2_manifest, dev)
    m.train(); m.set_batchsize(B); m.clear_flops(); m.flops_hook(g["expandflop"])
    tr = FusedTrainer(m, lr=g["lr"], weight_decay=g["weight_decay"], eps=g["eps"], betas=tuple(g["betas"]),
                      flops_weight=g["flops_weight"], batchsize=B, lib=lib, act_dtype=act_dtype)
    xd, td = x.to(dev), t.to(dev)
    mine, pens = [], []
    for _ in range(steps):
        loss, pen = tr.step(xd, td)
        mine.append(float(loss)); pens.append(float(pen))
    ref = g["bce"]
    end_dev = abs(np.mean(mine[-10:]) - np.mean(ref[-10:])) / np.mean(ref[-10:])
    print(f"{act_dtype}: bce {mine[0]:.4f} -> {mine[-1]:.4f}; reference {ref[0]:.4f} -> {ref[-1]:.4f}; "
          f"max |diff| {max(abs(a - b) for a, b in zip(mine, ref)):.3e} (step {int(np.argmax([abs(a - b) for a, b in zip(mine, ref)]))}); "
          f"first step {abs(mine[0] - ref[0]):.2e}, second {abs(mine[1] - ref[1]):.2e}, mean of the last ten: relative {end_dev:.2e}")
    assert mine[-1] < 0.6 * mine[0], mine
    assert np.mean(mine[-10:]) < np.mean(mine[10:20]) < np.mean(mine[:10])
    # Two free-running trajectories are NOT 1e-2 apart step by step: at lr 1e-3 the first updates move the ~1e-6-gamma channels of
    # the shipped checkpoint across PReLU kinks (the CPU emulation of these very kernels, fp32, is 4.5e-2 from the oracle at step 7
    # and back within 1.1 % at the end; with bf16 storage 0.13 / 1.5 %; MI355X bf16: 0.10 at step 8).  Hence: the first two steps
    # (same parameters up to one update) tight, the whole curve loose, the end of the run in between.
    tol0, tol_curve, tol_end = (2e-3, 0.1, 0.05) if act_dtype == "fp32" else (3e-2, 0.2, 0.1)
    assert abs(mine[0] - ref[0]) <= tol0 * max(1.0, abs(ref[0])) and abs(pens[0] - g["penalty"][0]) <= 5 * tol0 * max(1.0, g["penalty"][0])
    assert abs(mine[1] - ref[1]) <= 5 * tol0 * max(1.0, abs(ref[1])), (mine[:2], ref[:2])
    for k, (a, b) in enumerate(zip(mine, ref)):
        assert abs(a - b) <= tol_curve * max(1.0, abs(b)), (k, a, b)
    assert abs(np.mean(mine[-10:]) - np.mean(ref[-10:])) <= tol_end * np.mean(ref[-10:]), (mine[-10:], ref[-10:])


@pytest.mark.parametrize("shape,min_blocks,maxpix", [((2, 224, 224), 8, 1024), ((3, 64, 64), 8, 1024), ((2, 96, 160), 8, 1024),
                                                      ((64, 224, 224), 3, None)])
def test_gpu_ilb_matches_unit_kernels(hip, x2_manifest, shape, min_blocks, maxpix):
    """Whole-ILBlock launches (k_ilb.hip, round 5) on the device: every block output against the unit kernels they replace
    (CSN_ILB=0), the logits of both against the oracle; batch 64 = the bench's own plan."""
    lib, dev = hip
    n, worst, err = P.check_ilb_vs_unit_kernels(lib, dev, x2_manifest, *shape, min_blocks=min_blocks,
                                                env={"CSN_ILB_MAXPIX": str(maxpix)} if maxpix else None)
    print(f"{shape}: {n} units on ilb_kernel, worst block deviation {worst:.2e}, logits vs oracle {err:.2e}")


def test_gpu_ilb_x1_and_two_tile_groups(hip, x1_manifest, x2_manifest):
    lib, dev = hip
    print("x1:", P.check_ilb_vs_unit_kernels(lib, dev, x1_manifest, 2, 224, 224, min_blocks=4, env={"CSN_ILB_MAXPIX": "1024"}))
    print("nt 2:", P.check_ilb_vs_unit_kernels(lib, dev, x2_manifest, 2, 64, 64, env={"CSN_ILB_NT": "2", "CSN_ILB_MAXPIX": "1024"}))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 224, 224), (3, 64, 64), (2, 96, 160), (1, 256, 256)])
def test_gpu_lane_exchange_equals_loaded_halos_eval(hip, x2_manifest, shape):
    """Round 5: the depthwise pair's halo columns from the neighbouring lanes (one DPP move each, rows of lanes as power-of-two groups
    inside a wave) and c3q_kernel's window edge columns on halo-lane tiles (62 quads per wave, lanes 0 / 63 only load) -- bit-identical
    logits to the kernels that load them (CSN_DW_XL=0, CSN_C3Q_HL=0), and inside the oracle bound.  256 x 256: 64 strips per row,
    no idle lane in a group (the last lane's right neighbour is the first lane of the next ROW: masked)."""
    lib, dev = hip
    print(shape, "logits vs oracle %.2e, bit-identical to CSN_DW_XL=0" % P.check_lane_exchange_vs_loaded_halos(lib, dev, x2_manifest, *shape))


@pytest.mark.gpu
@pytest.mark.parametrize("act_dtype,size", [("fp32", 96), ("bf16", 224)])
def test_gpu_lane_exchange_equals_loaded_halos_train(hip, x2_manifest, act_dtype, size):
    """... and in the train step: the lane geometry changes the tiling of the partial sums (statistics, weight gradients), not any
    pixel's arithmetic: gradients agree to rounding (fp32) / to bf16 rounding flips."""
    lib, dev = hip
    f1, g1 = P.train_backward_probes(lib, dev, x2_manifest, 2, size, act_dtype, {"CSN_DW_XL": "1"})
    f0, g0 = P.train_backward_probes(lib, dev, x2_manifest, 2, size, act_dtype, {"CSN_DW_XL": "0"})
    worst = max(float((g1[k].double() - g0[k].double()).norm() / (g0[k].double().norm() + 1e-30)) for k in g1)
    rel = float((f1.double() - f0.double()).norm() / f0.double().norm())
    print(f"{act_dtype} {size}: worst stored input gradient {worst:.2e}, flat gradient {rel:.2e}")
    assert worst < (2e-2 if act_dtype == "bf16" else 5e-6) and rel < (2e-2 if act_dtype == "bf16" else 5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("act_dtype,size", [("fp32", (48, 80)), ("bf16", 224)])
def test_gpu_epilogue_routing_and_statistics_equal_their_own_passes(hip, x2_manifest, act_dtype, size):
    """Round 6: pwq_kernel's epilogue routes the 2x2 max-pool adjoint (CSN_POOL_ROUTE) and, in the bf16 mode, pw4_kernel's epilogue
    leaves the BatchNorm statistics of its outputs (CSN_PW4_STATS: wave sums by DPP / ds_bpermute) -- against the passes they replace
    (maxpool2_bwd_add_pair_kernel, bn_stats_kernel).  fp32: the routing adds the same numbers in the same order: bit-identical;
    bf16: one rounding less per routed gradient and another summation order of the statistics: bf16 rounding flips."""
    lib, dev = hip
    f1, g1 = P.train_backward_probes(lib, dev, x2_manifest, 2, size, act_dtype, {"CSN_POOL_ROUTE": "1", "CSN_PW4_STATS": "1"})
    f0, g0 = P.train_backward_probes(lib, dev, x2_manifest, 2, size, act_dtype, {"CSN_POOL_ROUTE": "0", "CSN_PW4_STATS": "0"})
    worst = max(float((g1[k].double() - g0[k].double()).norm() / (g0[k].double().norm() + 1e-30)) for k in g1)
    rel = float((f1.double() - f0.double()).norm() / f0.double().norm())
    print(f"{act_dtype} {size}: worst stored input gradient {worst:.2e}, flat gradient {rel:.2e}")
    if act_dtype == "fp32":
        assert torch.equal(f1, f0) and worst == 0.0
    else:
        assert worst < 2e-2 and rel < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("shape,env,fuse_cls", [((2, 224, 224), None, True), ((3, 224, 224), None, False),
                                                ((2, 96, 160), {"CSN_HZ_RB": "3", "CSN_HZ_NT": "3/5", "CSN_HZ_HB": "4"}, False),
                                                ((3, 16, 16), {"CSN_HZ_RB": "1", "CSN_HZ_NW": "16"}, True)])
def test_gpu_hz_matches_pw4_and_oracle(hip, x2_manifest, shape, env, fuse_cls):
    """Round 6: hz_kernel (k_head.hip; the high output of CSFHead.fuse / fuse1x1 with the low -> high terms as low-resolution products
    through LDS, csnet.py:702-707) on the device against pw4_kernel's high-only form (CSN_HZ=0) and the oracle; the last two cases
    walk other band heights / M groups / block sizes than the product's."""
    lib, dev = hip
    n, worst, err = P.check_hz_vs_pw4(lib, dev, x2_manifest, *shape, env=env, fuse_cls=fuse_cls)
    print(f"{shape} {env}: {n} launches on hz_kernel, worst unit deviation {worst:.2e}, logits vs oracle {err:.2e}")
