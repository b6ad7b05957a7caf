"""Prune-and-finetune surgery against G10, the reference's own result on the shipped x2 weights."""
import json
import os

import numpy as np
import torch

from oracle import csnet_oracle as O
from sod100k_amd.model import csnet as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(e):
    if isinstance(e, (list, tuple)):
        return [_norm(v) for v in e]
    if isinstance(e, np.ndarray):
        return _norm(e.tolist())
    return float(e)


def test_prune_matches_reference(tmp_path):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "g10_prune_x2.json")))
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    m = M.build_model(predefine=man)
    m.load_state_dict(O.load_weights(man))
    new_cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=M.load_layer_config(man), thres=g["thres"])
    assert _norm(new_cfg) == _norm(g["layer_config"])
    assert [[int(np.count_nonzero(b)) for b in layer] for layer in mask if layer is not None] == g["mask_counts"]
    slim = M.build_model_with_weight(new_cfg, m, mask)
    sd = slim.state_dict()
    assert list(sd.keys()) == list(g["keys"].keys())
    assert sum(p.numel() for p in slim.parameters()) == g["n_params"]
    for k, e in g["keys"].items():
        v = sd[k]
        assert list(v.shape) == e["shape"], k
        assert abs(float(v.double().sum()) - e["sum"]) <= 1e-9 * max(1.0, abs(e["sum"])), k
        assert np.allclose(v.double().reshape(-1)[:3].numpy(), e["head"], rtol=0, atol=0), k


def test_build_model_finetune_path(tmp_path):
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    m = M.build_model(predefine=man)
    m.load_state_dict(O.load_weights(man))
    slim = M.build_model(epoch=3, predefine=man, save_path=str(tmp_path), model=m, load_weight="FINETUNE",
                         finetune_thres=0.01, finetune=True)
    assert sum(p.numel() for p in slim.parameters()) == 55740
    assert os.path.isfile(os.path.join(str(tmp_path), "layer_config_finetune_3.bin"))
    again = M.CSNet(M.load_layer_config(os.path.join(str(tmp_path), "layer_config_finetune_3.bin")))
    assert [tuple(v.shape) for v in again.state_dict().values()] == [tuple(v.shape) for v in slim.state_dict().values()]


def test_slim_model_runs_through_the_kernels(emu_lib, tmp_path):
    """The pruned network (it has an output branch with zero channels) through the same kernels, against the oracle."""
    from oracle import inputs as I
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    m = M.build_model(predefine=man)
    m.load_state_dict(O.load_weights(man))
    new_cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=M.load_layer_config(man), thres=0.01)
    slim = M.build_model_with_weight(new_cfg, m, mask).eval()
    slim._lib = emu_lib
    x = torch.from_numpy(I.randn_batch(2, 2, 32, 48))
    sd = {k: v.clone() for k, v in slim.state_dict().items()}
    with torch.no_grad():
        ref = O.csnet_forward(new_cfg, sd, x)
    y = slim(x)
    assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_finetune_caller_end_to_end(emu_lib, tmp_path, monkeypatch):
    """sod100k_amd/tools/finetune.py (finetune.py:84-207): training checkpoint -> prune -> a finetune step on the slim
    network (emulated kernels) -> val() -> checkpoint whose keys are the slim network's."""
    import sys
    from sod100k_amd.configs import defaults
    from sod100k_amd.tools import finetune as F
    from oracle import inputs as I
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    cfg = defaults()
    cfg.DATA.SAVEDIR, cfg.TASK = str(tmp_path), "toy"
    cfg.DATA.BATCH_SIZE, cfg.DATA.IMAGE_H, cfg.DATA.IMAGE_W = 2, 32, 32
    cfg.FINETUNE.THRES = 0.01
    cfg.FINETUNE.SOLVER.MAX_EPOCHS, cfg.FINETUNE.SOLVER.LR = 1, 1e-5
    cfg.FINETUNE.SOLVER.ADJUST_STEP, cfg.FINETUNE.SOLVER.LR_SCHEDULER = True, 'cosine'
    lc_dir = os.path.join(str(tmp_path), "toy", "layer_configs")
    M.save_layer_config(M.load_layer_config(man), lc_dir, 0)
    trained = M.build_model(predefine=man)
    trained.load_state_dict(O.load_weights(man))
    os.makedirs(os.path.join(str(tmp_path), "toy", "checkpoint"))
    torch.save({'epoch': 7, 'arch': 'csnet', 'state_dict': trained.state_dict()},
               os.path.join(str(tmp_path), "toy", "checkpoint", "checkpoint_epoch7.pth.tar"))
    sys.path.insert(0, os.path.join(ROOT, "sod100k_amd"))
    x = torch.from_numpy(I.randn_batch(3, 2, 32, 32))
    targets = [(torch.rand(40, 30) > 0.5).float(), (torch.rand(32, 32) > 0.5).float()]
    slim = F.run(cfg, 7, device="cpu", synthetic=1, max_steps=1, val_batches=[(x, targets)], lib=emu_lib)
    assert sum(p.numel() for p in slim.parameters()) == 55740           # G10: the pruned x2 network
    assert abs(F.finetune_lr(cfg.FINETUNE.SOLVER, 0) - 0.0) < 1e-12      # cosine, T_max = 1: annealed to eta_min
    out = torch.load(os.path.join(str(tmp_path), "toy", "finetune_checkpoint", "checkpoint_epoch1.pth.tar"))
    assert out['epoch'] == 1 and list(out['state_dict'].keys()) == list(slim.state_dict().keys())
    assert os.path.isfile(os.path.join(lc_dir, "layer_config_finetune_7.bin"))


import pytest


@pytest.mark.parametrize("seed,kill", [(0, 0.35), (21, 0.7), (33, 0.9)])
def test_randomly_pruned_networks_through_the_kernels(emu_lib, tmp_path, seed, kill):
    """Planner robustness: prune a random 35 / 70 / 90 % of every BatchNorm's channels of the un-pruned expand-1 network with
    the real surgery (odd channel counts, single channels, empty branches), then eval forward on the emulated kernels
    against the oracle."""
    import contextlib
    import io
    from oracle import inputs as I
    from test_unpruned_emu import _random_state
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.build_model(basic_split=[0.5, 0.5], expand=1.0, save_path=str(tmp_path))
    sd = _random_state(m, seed)
    g = torch.Generator().manual_seed(1000 + seed)
    for k in sd:
        if ('.bns.' in k or '.bn.' in k) and k.endswith('weight'):
            dead = torch.rand(sd[k].shape, generator=g) < kill
            sd[k] = torch.where(dead, torch.full_like(sd[k], 1e-6), sd[k])
    m.load_state_dict(sd)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=O.init_layers(20, [0.5, 0.5]), thres=1e-3)
        slim = M.build_model_with_weight(cfg, m, mask).eval()
    slim._lib = emu_lib
    x = torch.from_numpy(I.randn_batch(seed, 2, 32, 32))
    ssd = {k: v.clone() for k, v in slim.state_dict().items()}
    with torch.no_grad():
        ref = O.csnet_forward(cfg, ssd, x)
    y = slim(x)
    assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def check_top_branch_pruned(lib, device, tmp_path):
    """Regression (found by fuzzing): 90 % pruning leaves stage0.0 with NO channels in its full-resolution output branch,
    i.e. a launch whose only pass walks the half-resolution branch.  It used to reach the LDS-tiled 3x3 kernel with the
    launch resolution of branch 0 (out-of-bounds reads and writes); such launches are now re-based on their own
    resolution.  Train-mode forward: logits, penalty and the BN running statistics against the oracle."""
    import contextlib
    import io
    from oracle import inputs as I
    from test_unpruned_emu import _random_state
    seed, kill = 33, 0.9
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.build_model(basic_split=[0.5, 0.5], expand=1.0, save_path=str(tmp_path))
    sd = _random_state(m, seed)
    g = torch.Generator().manual_seed(1000 + seed)
    for k in sd:
        if ('.bns.' in k or '.bn.' in k) and k.endswith('weight'):
            dead = torch.rand(sd[k].shape, generator=g) < kill
            sd[k] = torch.where(dead, torch.full_like(sd[k], 1e-6), sd[k])
    m.load_state_dict(sd)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=O.init_layers(20, [0.5, 0.5]), thres=1e-3)
        slim = M.build_model_with_weight(cfg, m, mask)
    assert int(np.asarray(cfg[0][1])[0]) == 0            # the full-resolution output branch of stage0.0 is gone
    slim = slim.to(device)
    if device.type == "cpu":
        slim._lib = lib
    slim.train(); slim.set_batchsize(2); slim.clear_flops(); slim.flops_hook(1.0)
    x = torch.from_numpy(I.randn_batch(seed, 2, 32, 32))
    ssd = {k: v.detach().cpu().clone() for k, v in slim.state_dict().items()}
    yt, pen = slim._train_forward_raw(x.to(device))
    yt, pen = yt.cpu(), pen.cpu()
    taps = {}
    with torch.no_grad():
        ref = O.csnet_forward(cfg, ssd, x, training=True, taps=taps)
    pen_ref = float(O.gap_penalty(ssd, taps, O.flop_weights(cfg, 1.0), 2))
    assert (yt - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    assert abs(float(pen) / 2 - pen_ref) <= 1e-5 * max(1.0, abs(pen_ref))
    got = {k: v.cpu() for k, v in slim.state_dict().items()}
    for k, v in ssd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert ((got[k] - v).abs() / (1.0 + v.abs())).max().item() <= 1e-5, k



def test_train_forward_when_the_top_branch_is_pruned_away(emu_lib, tmp_path):
    check_top_branch_pruned(emu_lib, torch.device("cpu"), tmp_path)


@pytest.mark.gpu
def test_gpu_train_forward_when_the_top_branch_is_pruned_away(tmp_path):
    from sod100k_amd import _native as N
    check_top_branch_pruned(N.load(), torch.device("cuda", 0), tmp_path)


def test_train_step_with_an_output_branch_nobody_consumes(emu_lib, tmp_path):
    """85 % pruning (seed 41) removes a whole MSBlock, so one output branch of CSFHead.fuse has no consumer: autograd gives its
    parameters no gradient; the plan gives the branch a zero gradient (it used to refuse the configuration).  All
    gradients against the fp64 oracle, in units of the largest gradient norm."""
    import contextlib
    import io
    import parity_cases as P
    from oracle import inputs as I
    from test_unpruned_emu import _random_state
    seed, kill = 41, 0.85
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.build_model(basic_split=[0.5, 0.5], expand=1.0, save_path=str(tmp_path))
    sd = _random_state(m, seed)
    g = torch.Generator().manual_seed(1000 + seed)
    for k in sd:
        if ('.bns.' in k or '.bn.' in k) and k.endswith('weight'):
            dead = torch.rand(sd[k].shape, generator=g) < kill
            sd[k] = torch.where(dead, torch.full_like(sd[k], 1e-6), sd[k])
    m.load_state_dict(sd)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=O.init_layers(20, [0.5, 0.5]), thres=1e-3)
        slim = M.build_model_with_weight(cfg, m, mask)
    assert any(c is None for c in slim.oct_fuse.ms.convs)          # a pruned-away MSBlock
    slim._lib = emu_lib
    x = torch.from_numpy(I.randn_batch(seed, 2, 32, 32))
    t = torch.from_numpy(I.binary_target(seed + 1, 2, 32, 32))
    ssd = {k: v.clone() for k, v in slim.state_dict().items()}
    slim.train(); slim.set_batchsize(2); slim.clear_flops(); slim.flops_hook(1.0)
    yt, pen = slim._train_forward_raw(x)
    loss, dy = P.bce_and_grad(emu_lib, yt, t)
    flat = slim._train_backward_raw(x, dy, 3.0 / 2)
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in ssd.items()}, x.double(),
                       t.double(), expandflop=1.0, flops_weight=3.0, batchsize=2, lr=0.0, wd=0.0)
    assert abs(float(loss) - r64["loss_bce"]) <= 1e-5
    errs = P.grad_errors(slim, flat, r64["grads"])
    gmax = max(n for _, n in errs.values())
    # BatchNorms the surgery kept with gamma = 1e-6 put EVERY element of their output within rounding distance of the PReLU
    # kink (bn = beta + 1e-6 * xhat), where the derivative is discontinuous: their own gamma / alpha gradients are
    # decided by last-bit differences between any two implementations (ATen CPU vs GPU included) -- not compared
    dead = {k.rsplit(".", 1)[0] for k, v in ssd.items() if ".bns." in k and k.endswith(".weight") and v.abs().max() <= 1e-5}
    live = {k: e for k, (e, _) in errs.items() if k.rsplit(".", 1)[0] not in dead and k.replace(".prelus.", ".bns.").rsplit(".", 1)[0] not in dead}
    assert len(live) > 0.5 * len(errs)
    assert max(live.values()) <= 1e-4 * gmax, max(live.items(), key=lambda kv: kv[1])
