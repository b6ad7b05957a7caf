"""CPU-emulated kernels (same sources as libcsnet_hip.so, g++ + fibers) against goldens and oracle.

No GPU is needed: this checks the index arithmetic / data flow of every kernel and of the plan before the
MI355X run; the ``-m gpu`` twins in test_gpu_parity.py are the parity tests proper."""
import numpy as np
import torch

from oracle import inputs as I

import parity_cases as P

CPU = torch.device("cpu")


def test_emu_small_vs_oracle(emu_lib, x2_manifest):
    P.check_vs_oracle(emu_lib, CPU, x2_manifest, torch.from_numpy(I.randn_batch(5, 2, 32, 48)))


def test_emu_minimum_size_16(emu_lib, x2_manifest):
    """16x16 input: the lowest branch is a single pixel (test.py:80-85 only demands multiples of 16)."""
    P.check_vs_oracle(emu_lib, CPU, x2_manifest, torch.from_numpy(I.randn_batch(6, 3, 16, 16)))


def test_emu_golden_224(emu_lib, x2_manifest):
    x = torch.from_numpy(I.randn_batch(0, 2))
    P.check_golden_logits(emu_lib, CPU, x2_manifest, "g2_logits_x2_randn_b2.npy", x)


def test_emu_golden_nonsquare(emu_lib, x2_manifest):
    x = torch.from_numpy(I.randn_batch(3, 2, 96, 160))
    P.check_golden_logits(emu_lib, CPU, x2_manifest, "g2_logits_x2_randn_b2_96x160.npy", x)


def test_emu_x1_config(emu_lib, x1_manifest):
    x = torch.from_numpy(I.randn_batch(0, 2))[:1]
    P.check_golden_logits(emu_lib, CPU, x1_manifest, "g2_logits_x1_randn_b1.npy", x)


def test_emu_unit_probes(emu_lib, x2_manifest):
    P.check_unit_probes(emu_lib, CPU, x2_manifest)


def test_emu_op_goldens(emu_lib):
    rep = P.check_g4(emu_lib, CPU)
    assert len(rep) >= 11


def test_emu_sub_batch_slicing(emu_lib, x2_manifest):
    x = torch.from_numpy(I.randn_batch(7, 5, 32, 32))
    y_full, _ = P.check_vs_oracle(emu_lib, CPU, x2_manifest, x, sub_batch=0)
    y_sl, _ = P.check_vs_oracle(emu_lib, CPU, x2_manifest, x, sub_batch=2)   # 2+2+(overlapping last) slices
    assert torch.equal(y_full, y_sl)


def test_emu_train_forward_vs_oracle(emu_lib, x2_manifest):
    err, pen, pen_ref, worst = P.check_train_forward(emu_lib, CPU, x2_manifest, B=3, size=48)
    assert pen_ref > 0


def test_emu_train_forward_expandflop1_x1(emu_lib, x1_manifest):
    P.check_train_forward(emu_lib, CPU, x1_manifest, B=2, size=32, expandflop=1.0, seed=3)


def test_emu_train_step_gradients(emu_lib, x2_manifest):
    """Every parameter gradient of one train step (BCE + dynamic weight decay) vs autograd through the oracle."""
    worst, loss, pen = P.check_train_step(emu_lib, CPU, x2_manifest, B=2, size=32)
    print("worst relative gradient error", worst)


def test_emu_autograd_seam(emu_lib, x2_manifest):
    P.check_autograd_seam(emu_lib, CPU, x2_manifest, B=2, size=16)


def test_emu_pre_post_processing(emu_lib, x2_manifest):
    P.check_pre_post(emu_lib, CPU, x2_manifest)


def test_emu_train_step_well_conditioned(emu_lib, x2_manifest):
    """Gradients on a well-conditioned state of the shipped architecture: no further from fp64 than the fp32 reference."""
    print("rel-L2 vs fp64: kernels %.2e, fp32 oracle %.2e" % P.check_train_step_well_conditioned(emu_lib, torch.device("cpu"), x2_manifest, B=2, size=32, local_seeds=(31, 101)))


def test_emu_train_step_bf16(emu_lib, x2_manifest):
    """BASELINE config 3's dtype (bfloat16 activation storage) on the CPU emulation of the kernels."""
    print(P.check_train_step_bf16(emu_lib, CPU, x2_manifest, B=2, size=32))


import pytest


@pytest.mark.parametrize("act_dtype,B,size,state", [("fp32", 2, 64, "shipped"), ("fp32", 3, 48, "well"), ("bf16", 2, 64, "shipped"),
                                                    ("bf16", 3, 48, "well"), ("bf16", 2, 16, "shipped"),
                                                    # 224 / 320 wide: flat pw4 / c3q tiles, two depthwise tiles per row at 320
                                                    ("fp32", 1, (32, 224), "well"), ("bf16", 1, (32, 320), "shipped")])
def test_emu_train_units_local(emu_lib, x2_manifest, act_dtype, B, size, state):
    """Every unit's train-mode forward and backward (dz, dx per consumer slot, every parameter gradient) against the oracle
    applied to the tensors the kernels themselves produced around that unit -- no error amplification through depth."""
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=B, size=size, act_dtype=act_dtype, state=state))


@pytest.mark.parametrize("act_dtype,env", [("fp32", {}), ("bf16", {}),
                                           # the first unit's input gradient on the generic kernels (the round-4 switches off / unfused)
                                           ("bf16", {"CSN_PWQ16": "0", "CSN_C3Q16": "0", "CSN_C3Q_BWD": "0", "CSN_POOL_ROUTE": "0"}),
                                           ("fp32", {"CSN_C3Q_BWD": "0", "CSN_ADJ_FUSE": "0"})])
def test_emu_input_gradient_unit_local(emu_lib, x2_manifest, monkeypatch, act_dtype, env):
    """CSN_OPT_INPUT_GRAD (csn_backward's `dx`): the plan also forms the gradient w.r.t. the image batch; checked as the
    first unit's dx inside the unit-local bounds, next to every other unit (whose results must not move)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=2, size=48, act_dtype=act_dtype, state="shipped", input_grad=True))


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_emu_train_units_local_badly_centred_channels(emu_lib, x2_manifest, act_dtype):
    """the fused depthwise backward forms dz = g sel - (B z + A) with the batch mean folded into A; on channels whose raw
    output has |mean| ~ 10 standard deviations (as far as zero padding lets a depthwise output go) B z and A cancel to ~10 % of their size.
    Every unit stays inside the unit-local bounds (measured: dz 3e-7 in fp32)."""
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=2, size=48, act_dtype=act_dtype, state="offcentre"))


@pytest.mark.parametrize("act_dtype,ipp", [("fp32", 2), ("bf16", 3)])
def test_emu_train_units_local_multi_image_slabs(emu_lib, x2_manifest, monkeypatch, act_dtype, ipp):
    """The BN / depthwise reductions hand several whole planes to one block when the batch is large (batch 256: 5-20 images
    per block); CSN_BN_IPP forces that path at a batch the emulator can run (5 images: slabs of ipp, ..., remainder)."""
    monkeypatch.setenv("CSN_BN_IPP", str(ipp))
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=5, size=32, act_dtype=act_dtype, state="well"))


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_emu_train_units_local_fallback_paths(emu_lib, x2_manifest, monkeypatch, act_dtype):
    """The round-1 schemes stay in the library as fallbacks (batches whose per-(image, tile) partials do not fit the reduction
    table, rows beyond the wave kernel, ...): one weight-gradient launch per forward pass with bilinear gathers, 3x3 weight
    gradients on the per-pixel kernel, stand-alone BN statistics / depthwise weight-gradient passes."""
    for k, v in (("CSN_WGRAD_REGROUP", "1"), ("CSN_WGRAD_TILED3", "0"), ("CSN_DW_STATS", "0"), ("CSN_DW_BWD_SPLIT", "1")):
        monkeypatch.setenv(k, v)
    import oracle.csnet_oracle as O
    monkeypatch.setattr(O, "DW_DZ_STORED", True)   # two-pass depthwise backward: dz goes through memory (bf16 rounding point)
    monkeypatch.setattr(O, "DW_IN_STORED", True)   # ... and every activation is stored
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=2, size=48, act_dtype=act_dtype, state="shipped"))


@pytest.mark.parametrize("env", [{"CSN_PWQ16": "0", "CSN_C3Q16": "0", "CSN_MS_DX": "0", "CSN_WGRAD_BF": "0", "CSN_WGRAD_BF3": "0"},
                                 {"CSN_ADJ_FUSE": "0", "CSN_ADJ4_ROWS": "0", "CSN_C3Q_BWD": "0", "CSN_BWD_NO_DEFER": "1", "CSN_DWB_FAST": "0"}])
def test_emu_bf16_units_local_with_the_round4_kernels_switched(emu_lib, x2_manifest, monkeypatch, env):
    """The round-4 kernels of the bf16 step behind their switches: (a) all off -- the fp32 matrix instruction for the 1x1 / 3x3 input
    gradients and the weight gradients, the generic tap kernel for the MSBlock input gradient; (b) apply / adjoint unfused, 3x3 input
    gradients on the LDS-tiled kernel, no launch batching.  Every unit stays inside the same unit-local bounds.  (CSN_C3Q16=2, the 3x3
    FORWARD launches with bf16 weights, is NOT certified: z of the stride-2 units lands at 2.9e-3 against the 2e-3 bound.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=2, size=64, act_dtype="bf16", state="shipped"))


@pytest.mark.parametrize("act_dtype,size", [("fp32", 48), ("bf16", (32, 320))])
def test_emu_packed_pair_depthwise_backward_equals_the_round4_loop(emu_lib, x2_manifest, act_dtype, size):
    """dw3x3_bwd_x_kernel (round 5: dx in scatter form on packed pairs) against dw3x3_bwd_kernel (CSN_DWB_FAST=0): the order of
    operations per dx value is the same, so a depthwise unit given the same dy produces the same dx bit for bit; the weight-gradient
    partial sums are taken in another order (even / odd columns per tap) and (z - mean) invstd is one FMA, so the producer's
    BatchNorm-backward sums -- and with them every gradient further upstream -- differ in the last bits (bf16 storage: a last-bit
    difference flips roundings, measured 4.7e-3 at worst against 2.9e-7 in fp32)."""
    f1, g1 = P.train_backward_probes(emu_lib, CPU, x2_manifest, 2, size, act_dtype, {"CSN_DWB_FAST": "1"})
    f0, g0 = P.train_backward_probes(emu_lib, CPU, x2_manifest, 2, size, act_dtype, {"CSN_DWB_FAST": "0"})
    assert g1.keys() == g0.keys() and len(g1) > 40
    same = sum(bool(torch.equal(g1[k], g0[k])) for k in g1)
    worst = max(float((g1[k].double() - g0[k].double()).norm() / (g0[k].double().norm() + 1e-30)) for k in g1)
    rel = float((f1.double() - f0.double()).norm() / f0.double().norm())
    print(f"{act_dtype}: {same} of {len(g1)} stored input gradients bit-identical, worst relative L2 {worst:.2e}; flat gradient {rel:.2e}")
    assert worst < (1e-2 if act_dtype == "bf16" else 2e-6) and rel < (1e-2 if act_dtype == "bf16" else 2e-6)
    assert same >= 2   # the depthwise units nearest the loss see identical inputs


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_emu_train_units_local_on_the_pruned_network(emu_lib, x2_manifest, tmp_path, act_dtype):
    """The network the reference FINETUNES (finetune_model / build_model_with_weight: branches and dilations with zero channels, odd
    channel counts): every unit's train-mode forward and backward inside the same unit-local bounds."""
    net = P.slim_network(x2_manifest, tmp_path)
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=2, size=64, act_dtype=act_dtype, net=net))


def test_emu_bf16_trainer_steps_and_eval_afterwards(emu_lib, x2_manifest):
    """FusedTrainer in bf16 storage mode: a few optimizer steps move the parameters, the loss stays finite and close to the
    fp32 trainer's, and the eval-mode forward afterwards is the fp32 path (equal to the oracle on the updated state)."""
    from sod100k_amd.tools.train import FusedTrainer
    from oracle import csnet_oracle as O
    x = torch.from_numpy(I.randn_batch(90, 2, 32, 32))
    t = torch.from_numpy(I.binary_target(91, 2, 32, 32))
    losses = {}
    for dt in ("fp32", "bf16"):
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        tr = FusedTrainer(m, lr=1e-4, weight_decay=5e-3, eps=1e-3, flops_weight=3.0, batchsize=2, lib=emu_lib, act_dtype=dt)
        ls = []
        for _ in range(3):
            loss, pen = tr.step(x, t)
            m.clear_flops()
            ls.append(float(loss))
        assert all(np.isfinite(ls)), (dt, ls)
        losses[dt] = ls
        if dt == "bf16":
            m.eval()
            with torch.no_grad():
                y = m(x)
                cur = {k: v.detach().clone() for k, v in m.state_dict().items()}
                ref = O.csnet_forward(O.load_layer_config_json(x2_manifest), cur, x)
            assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    # (batch 2 at 32x32 through 57 batch-normalised layers: ANY change of a bf16 rounding point moves the first loss by a few
    # per cent -- 1.708 with every activation stored, 1.675 with the never-stored ones, 1.721 in fp32; the per-unit bounds of
    # check_train_units_local are the sharp test)
    assert abs(losses["bf16"][0] - losses["fp32"][0]) <= 5e-2 * max(1.0, abs(losses["fp32"][0])), losses


def test_bf16_option_needs_training_buffers(emu_lib, x2_manifest):
    """CSN_OPT_TRAIN_BF16 on a plan without csn_plan_enable_training: csn_forward_train refuses (CSN_E_STATE), never a silent
    fp32 run."""
    from sod100k_amd import _native as N
    m, _ = P.make_model(emu_lib, x2_manifest, CPU)
    x = torch.from_numpy(I.randn_batch(92, 1, 32, 32))
    m.train()
    eng = m.engine_for(x, train=False)
    eng.set_option(N.OPT_TRAIN_BF16, 1)
    eng.refresh(m._arena.flat)
    with pytest.raises(RuntimeError):
        eng.forward_train(x, m._arena.flat, [0.0] * (len(m.describe(m._arena.offsets)[0]) * N.MAX_BRANCH),
                          torch.zeros(1, dtype=torch.float64))


def test_emu_fuse_lowest_branch_both_routes(emu_lib, x2_manifest, monkeypatch):
    """CSFHead.fuse's lowest output branch (W_22 x2 + W_21 maxpool2(x1) + W_20 maxpool4(x0), csnet.py:708-714) runs on
    pw4_kernel's low-only form with the 4x4-pooled fourth input; CSN_PW4_NOQ keeps it on goct_pw_kernel.  Both against
    the oracle, and the launch census says which one ran."""
    x = torch.from_numpy(I.randn_batch(11, 2, 32, 64))
    census = {}
    for noq in (False, True):
        if noq:
            monkeypatch.setenv("CSN_PW4_NOQ", "1")
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        y = m(x)
        ref = P.oracle_forward(x2_manifest, sd, x)
        assert (y - ref).abs().max().item() <= P.TOL
        eng = m.engine_for(x)
        eng.profile(x, iters=1)
        census[noq] = {k: v[1] for k, v in eng.kernel_stats().items()}
    assert census[False]["pw4_kernel"] == census[True]["pw4_kernel"] + 1, census
    assert census[False].get("goct_pw_kernel", 0) == census[True].get("goct_pw_kernel", 0) - 1, census


def test_emu_one_pass_depthwise_kernels_against_the_two_pass_scheme(emu_lib, x2_manifest):
    """fp32: activations formed on load + BatchNorm backward apply inside the depthwise backward = the stored scheme, gradient
    for gradient (bf16 storage has fewer rounding points that way: judged per unit by check_train_units_local)."""
    print(P.check_bn_bwd_fusion_bit_identical(emu_lib, CPU, x2_manifest, B=2, size=32, act_dtype="fp32"))


def test_bf16_train_plan_has_bf16_sized_workspace(emu_lib, x2_manifest):
    """With bfloat16 storage chosen before the training buffers are laid out, every activation-typed region of the workspace has
    2-byte elements (batch 256: 61 -> 31 GiB); such a plan refuses the fp32 eval forward and a switch back to fp32 storage."""
    from sod100k_amd import _native as N
    x = torch.zeros(8, 3, 224, 224)   # (plans only: at this size the activations outweigh the fixed-size reduction tables)
    sizes = {}
    for dt in ("fp32", "bf16"):
        m, _ = P.make_model(emu_lib, x2_manifest, CPU)
        m.set_train_act_dtype(dt)
        eng = m.engine_for(x, train=True)
        sizes[dt] = eng.workspace.numel()
        if dt == "bf16":
            with pytest.raises(RuntimeError):
                eng.forward(x)
            with pytest.raises(RuntimeError):
                eng.set_option(N.OPT_TRAIN_BF16, 0)
    # the float / double tables (statistics partials, |GAP| tables, the weight-gradient partial regions: ~0.5 GB whatever the
    # batch since round 4's per-pass / per-unit regions) do not shrink: 0.57 here, 0.50 at batch 256
    assert sizes["bf16"] < 0.65 * sizes["fp32"], sizes
    print(sizes)


def test_emu_max_pooled_copies_from_the_depthwise_pair(emu_lib, x2_manifest, monkeypatch):
    """The fused depthwise pair in front of a stride-2 3x3 unit writes the 2x2 averages AND their 2x2 maxima (the copy c3q_kernel's
    high -> low slice reads, F.max_pool2d of csnet.py:708-714): three of the four pool2_kernel launches of the forward are gone
    (the fourth pools the input image); CSN_NO_MP_FUSE keeps them.  Same logits, bit for bit."""
    x = torch.from_numpy(I.randn_batch(13, 2, 64, 96))
    out, census = {}, {}
    # (round 5: with the copies in place the stride-2 entry blocks run on ilb_kernel, without them on c3q_kernel -- different
    # summation order; the bit-for-bit statement is about the pooled copies, so both runs take the unit kernels)
    monkeypatch.setenv("CSN_ILB", "0")
    for nofuse in (False, True):
        if nofuse:
            monkeypatch.setenv("CSN_NO_MP_FUSE", "1")
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        out[nofuse] = m(x).clone()
        eng = m.engine_for(x)
        eng.profile(x, iters=1)
        census[nofuse] = eng.kernel_stats().get("pool2_kernel", (0, 0))[1]
    ref = P.oracle_forward(x2_manifest, sd, x)
    assert (out[False] - ref).abs().max().item() <= P.TOL
    assert torch.equal(out[False], out[True])
    assert (census[False], census[True]) == (1, 4), census


def test_emu_backward_with_weight_gradient_side_lane(emu_lib, x2_manifest, monkeypatch):
    """CSN_OPT_OVERLAP = 2 moves the weight-gradient launches of csn_backward to a side lane (own partial buffers); the depthwise
    units whose input was never stored stay on the one-pass kernel on the caller's stream (they also write the input gradient).
    Same gradients as the default schedule, bit for bit (the emulator runs the lanes in order: this checks the bookkeeping --
    buffers, partial tables, which kernel runs where -- not the concurrency)."""
    flats = {}
    for ov in ("1", "2"):
        monkeypatch.setenv("CSN_OVERLAP", ov)
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        x = torch.from_numpy(I.randn_batch(23, 2, 32, 32))
        t = torch.from_numpy(I.binary_target(24, 2, 32, 32))
        y, pen = m._train_forward_raw(x)
        loss, dy = P.bce_and_grad(emu_lib, y, t)
        flats[ov] = m._train_backward_raw(x, dy, 1.5).clone()
    assert torch.equal(flats["1"], flats["2"]), float((flats["1"] - flats["2"]).abs().max())


def test_emu_max_pool_adjoint_routed_in_the_input_gradient_launch(emu_lib, x2_manifest, monkeypatch):
    """Round 6: pwq_kernel's epilogue routes the 2x2 max-pool adjoint of the 1x1 units (PwqArgs::route_x) instead of
    maxpool2_bwd_add_pair_kernel's read-modify-write pass over dx (CSN_POOL_ROUTE=0).  fp32: the same additions in the same order --
    every gradient bit for bit, and the routing kernel's launches are gone; 48 x 80 pictures put odd row counts into the
    item tiles (lanes whose window partner row lies in another wave's tile)."""
    flats = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("CSN_POOL_ROUTE", sw)
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        x = torch.from_numpy(I.randn_batch(41, 2, 48, 80))
        t = torch.from_numpy(I.binary_target(42, 2, 48, 80))
        y, pen = m._train_forward_raw(x)
        loss, dy = P.bce_and_grad(emu_lib, y, t)
        flats[sw] = m._train_backward_raw(x, dy, 1.5).clone()
    assert torch.equal(flats["1"], flats["0"]), float((flats["1"] - flats["0"]).abs().max())


def test_emu_bf16_statistics_from_the_contraction_epilogue(emu_lib, x2_manifest, monkeypatch):
    """Round 6, bf16 train forward: pw4_kernel's epilogue leaves the BatchNorm statistics of its stored (rounded) outputs per
    (channel, item tile) (Pw4Args::stats_h) instead of bn_stats_kernel's pass over z (CSN_PW4_STATS=0).  The same sums in another
    order: logits, penalty and every gradient agree to summation-order accuracy (the running statistics ride in the parameter
    arena and are compared as well)."""
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("CSN_PW4_STATS", sw)
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        m.set_train_act_dtype("bf16")
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        x = torch.from_numpy(I.randn_batch(51, 2, 48, 80))
        t = torch.from_numpy(I.binary_target(52, 2, 48, 80))
        y, pen = m._train_forward_raw(x)
        loss, dy = P.bce_and_grad(emu_lib, y, t)
        out[sw] = (y.clone(), float(pen), m._train_backward_raw(x, dy, 1.5).clone())
    ya, pa, ga = out["1"]; yb, pb, gb = out["0"]
    assert (ya - yb).abs().max().item() <= 2e-3 * max(1.0, yb.abs().max().item())
    assert abs(pa - pb) <= 1e-4 * max(1e-6, abs(pb))
    assert (ga - gb).norm().item() <= 2e-2 * gb.norm().item()


def test_emu_results_do_not_depend_on_the_tile_geometry(emu_lib, x2_manifest, monkeypatch):
    """pw4_kernel / c3q_kernel tiles as 16 x 4 blocks (CSN_PW4_TWL = CSN_C3Q_TWL = 4), as row segments (6 with CSN_PW4_FLAT=0, round 3)
    or as 64 consecutive pixels of the plane (flat tiles, the default where rows do not fill their tiles): the same per-pixel
    arithmetic in the same order -- eval logits, train-mode logits and all gradients are equal bit for bit."""
    out = {}
    for tw, flat in (("4", "0"), ("6", "0"), ("6", "1")):
        monkeypatch.setenv("CSN_PW4_TWL", tw)
        monkeypatch.setenv("CSN_C3Q_TWL", tw)
        monkeypatch.setenv("CSN_PW4_FLAT", flat)
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        x = torch.from_numpy(I.randn_batch(5, 2, 96, 160))
        t = torch.from_numpy(I.binary_target(6, 2, 96, 160))
        ye = m(x).clone()
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        y, pen = m._train_forward_raw(x)
        loss, dy = P.bce_and_grad(emu_lib, y, t)
        out[tw + flat] = (ye, y.clone(), m._train_backward_raw(x, dy, 1.5).clone())
    for k in ("60", "61"):
        for a, b in zip(out["40"], out[k]):
            assert torch.equal(a, b), k


def test_emu_msblock_input_gradient_kernel_equals_the_generic_tap_kernel(emu_lib, x2_manifest, monkeypatch):
    """ms_dx_kernel (one launch, every dz tap loaded once) against the two generic tap launches it replaced (CSN_MS_DX=0): the same sums
    in another order -- all gradients equal to summation-order accuracy in fp32."""
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("CSN_MS_DX", sw)
        m, sd = P.make_model(emu_lib, x2_manifest, CPU)
        x = torch.from_numpy(I.randn_batch(5, 2, 64, 96))
        t = torch.from_numpy(I.binary_target(6, 2, 64, 96))
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        y, pen = m._train_forward_raw(x)
        loss, dy = P.bce_and_grad(emu_lib, y, t)
        out[sw] = m._train_backward_raw(x, dy, 1.5).clone().double()
    rel = float((out["1"] - out["0"]).norm() / out["0"].norm())
    assert 0.0 < rel <= 1e-5, rel   # (> 0: the two runs really took different kernels)

