"""The un-pruned `expand` networks of the training recipe (init_layers, csnet.py:414-518) through the same kernels."""
import numpy as np
import torch

from oracle import csnet_oracle as O, inputs as I
from sod100k_amd.model import csnet as M

import parity_cases as P

CPU = torch.device("cpu")


def _random_state(m, seed):
    g = torch.Generator().manual_seed(seed)
    sd = m.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("running_mean") or k.endswith("bias"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif "bn" in k and k.endswith("weight"):
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif "prelu" in k:
            out[k] = 0.25 + 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan = max(1, int(np.prod(v.shape[1:])))
            out[k] = torch.randn(v.shape, generator=g) * (0.5 / fan ** 0.5) * (0.01 if "convs" in k or "msconv" in k else 1.0)
    return out


def test_emu_unpruned_expand1_train_step(emu_lib):
    """basic_split [0.5, 0.5], expand 1.0 (width 20): eval forward and one train step against the oracle."""
    m = M.build_model(basic_split=[0.5, 0.5], expand=1.0, save_path="/tmp")
    sd = _random_state(m, 0)
    m.load_state_dict(sd)
    m._lib = emu_lib
    cfg = O.init_layers(20, [0.5, 0.5])
    x = torch.from_numpy(I.randn_batch(5, 2, 64, 64))
    t = torch.from_numpy(I.binary_target(6, 2, 64, 64))
    m.eval()
    with torch.no_grad():
        ref = O.csnet_forward(cfg, {k: v.clone() for k, v in sd.items()}, x)
    y = m(x)
    assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
    yt, pen = m._train_forward_raw(x)
    loss, dy = P.bce_and_grad(emu_lib, yt, t)
    flat = m._train_backward_raw(x, dy, 3.0 / 2)
    # Random weights with unit BN gains and 8..2048 samples per BN make this step ill-conditioned: the fp32 oracle
    # itself is ~1e-3 away from an fp64 run of the same step.  Judge the kernels against the fp64 run, with the fp32
    # oracle's own deviation as the yardstick.
    kw = dict(expandflop=1.0, flops_weight=3.0, batchsize=2, lr=0.0, wd=0.0)
    r32 = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                       x.double(), t.double(), **kw)
    assert abs(float(pen) / 2 - r64["penalty"]) <= 1e-5 * max(1.0, abs(r64["penalty"]))
    mine = np.array([e / (n + 1e-12) for e, n in P.grad_errors(m, flat, r64["grads"]).values()])
    ref = np.array([float((r32["grads"][k].double() - g).norm() / (g.norm() + 1e-12)) for k, g in r64["grads"].items()])
    assert np.median(mine) <= 2 * np.median(ref) and mine.max() <= 3 * ref.max(), (np.median(mine), np.median(ref),
                                                                                  mine.max(), ref.max())


def test_emu_unpruned_expand2_forward(emu_lib):
    """expand 2.0 (width 40, the training recipe's starting point): weight images beyond the LDS budget are cut into
    row chunks; eval forward and train-mode forward against the oracle."""
    m = M.build_model(basic_split=[0.5, 0.5], expand=2.0, save_path="/tmp")
    sd = _random_state(m, 1)
    m.load_state_dict(sd)
    m._lib = emu_lib
    cfg = O.init_layers(40, [0.5, 0.5])
    x = torch.from_numpy(I.randn_batch(7, 2, 32, 32))
    m.eval()
    with torch.no_grad():
        ref = O.csnet_forward(cfg, {k: v.clone() for k, v in sd.items()}, x)
    y = m(x)
    assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_emu_single_branch_std_conv_network(emu_lib):
    P.check_std_conv_network(emu_lib, CPU, _random_state)


def test_emu_unpruned_expand2_train_step(emu_lib):
    """The training recipe's starting point (expand 2.0, 788,631 parameters): weight images beyond the LDS budget are cut
    into row chunks in the backward launches and the weight-gradient kernels too -- one full step against the oracle
    (fp64 run as the truth, the fp32 oracle's own deviation as the yardstick; random state, hence ill-conditioned)."""
    m = M.build_model(basic_split=[0.5, 0.5], expand=2.0, save_path="/tmp")
    sd = _random_state(m, 4)
    m.load_state_dict(sd)
    m._lib = emu_lib
    cfg = O.init_layers(40, [0.5, 0.5])
    x = torch.from_numpy(I.randn_batch(5, 2, 32, 32))
    t = torch.from_numpy(I.binary_target(6, 2, 32, 32))
    m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
    yt, pen = m._train_forward_raw(x)
    loss, dy = P.bce_and_grad(emu_lib, yt, t)
    flat = m._train_backward_raw(x, dy, 3.0 / 2)
    kw = dict(expandflop=1.0, flops_weight=3.0, batchsize=2, lr=0.0, wd=0.0)
    r32 = O.train_step(cfg, {k: v.clone() for k, v in sd.items()}, x, t, **kw)
    r64 = O.train_step(cfg, {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                       x.double(), t.double(), **kw)
    assert abs(float(loss) - r64["loss_bce"]) <= 1e-5
    assert abs(float(pen) / 2 - r64["penalty"]) <= 1e-5 * max(1.0, abs(r64["penalty"]))
    mine = np.array([e / (n + 1e-12) for e, n in P.grad_errors(m, flat, r64["grads"]).values()])
    ref = np.array([float((r32["grads"][k].double() - g).norm() / (g.norm() + 1e-12)) for k, g in r64["grads"].items()])
    assert np.median(mine) <= 3 * np.median(ref) and mine.max() <= 3 * ref.max(), (np.median(mine), np.median(ref),
                                                                                  mine.max(), ref.max())


import pytest


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_emu_unpruned_expand2_train_units_local(emu_lib, x2_manifest, act_dtype):
    """Every unit of the un-pruned expand-2 net judged locally (the GPU twin: tests/test_gpu_paths.py) -- the wide-channel launches
    (several M groups / row chunks) of the train step, forward and backward."""
    net = P.unpruned_network(2.0, 40, seed=4)
    print(P.check_train_units_local(emu_lib, CPU, x2_manifest, B=2, size=32, act_dtype=act_dtype, net=net, input_grad=(act_dtype == "fp32")))
