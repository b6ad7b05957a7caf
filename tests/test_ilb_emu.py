"""Whole-ILBlock kernel (k_ilb.hip: conv1x1 -> conv3x3_1 -> conv3x3_2 per wave strip, csnet.py:72-76) on the CPU emulation:
strip / segment / channel-group decompositions, image borders, the pooled output for the stride-2 unit that follows,
pruned channel plans -- always against the oracle, and against the unit-level kernels of the same library."""
import os

import numpy as np
import pytest
import torch

from oracle import csnet_oracle as O, inputs as I
from sod100k_amd import _native as N

import parity_cases as P


def _forward(lib, manifest, x, ilb, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = str(v)
    try:
        m, sd = P.make_model(lib, manifest, torch.device("cpu"))
        eng = m.engine_for(x)
        eng.set_option(N.OPT_FUSE_ILB, ilb)
        y = m(x)
        names = [eng.lib.csn_unit_kernel_name(eng.plan, u).decode() for u in range(eng.n_units)]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return y, sd, names


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 96, 160), (1, 224, 224), (3, 48, 80)])
def test_ilb_vs_oracle_and_unit_kernels(emu_lib, x2_manifest, shape):
    b, h, w = shape
    x = torch.from_numpy(I.randn_batch(7, b, h, w))
    y, sd, names = _forward(emu_lib, x2_manifest, x, 1)          # every eligible block fused, whatever its width
    assert names.count("ilb_kernel") == 42                        # 14 one-by-one ILBlocks x 3 units
    ref = P.oracle_forward(x2_manifest, sd, x)
    assert (y - ref).abs().max().item() <= P.TOL
    y0, _, names0 = _forward(emu_lib, x2_manifest, x, 0)
    assert "ilb_kernel" not in names0
    assert (y - y0).abs().max().item() <= 2e-5                     # same arithmetic up to summation order


@pytest.mark.parametrize("env", [{"CSN_ILB_NC": 8}, {"CSN_ILB_NC": 12}, {"CSN_ILB_NC": 20}, {"CSN_ILB_SEG": 6},
                                 {"CSN_ILB_SEG": 10, "CSN_ILB_NC": 16}])
def test_ilb_decompositions(emu_lib, x2_manifest, env):
    """Forced channel-group widths (groups that do not divide the channel count, one padded group) and short row
    segments (every segment start re-derives the two-row history of both depthwise stages and the low-row pair)."""
    x = torch.from_numpy(I.randn_batch(9, 1, 112, 128))
    y, sd, _ = _forward(emu_lib, x2_manifest, x, 1, env)
    ref = P.oracle_forward(x2_manifest, sd, x)
    assert (y - ref).abs().max().item() <= P.TOL


def test_ilb_unit_probes_when_materialised(emu_lib, x2_manifest):
    """The block OUTPUT of every fused ILBlock against the reference's G3 probes of its conv3x3_2 unit."""
    import json
    m, _ = P.make_model(emu_lib, x2_manifest, torch.device("cpu"))
    x = torch.from_numpy(I.randn_batch(0, 2))
    eng = m.engine_for(x)
    eng.set_option(N.OPT_FUSE_ILB, 1)
    eng.set_option(N.OPT_FUSE_CLS, 0)
    m(x)
    units, acts, names = m.describe(m._arena.offsets)
    probes = json.load(open(os.path.join(P.GOLD, "g3_unit_probes_x2.json")))
    checked = 0
    for ui, (u, name) in enumerate(zip(units, names)):
        if not name.endswith("conv3x3_2") or eng.lib.csn_unit_kernel_name(eng.plan, ui).decode() != "ilb_kernel":
            continue
        nxt = units[ui + 1] if ui + 1 < len(units) else None
        for j in range(N.MAX_BRANCH):
            a = u.out_act[j]
            if a < 0:
                continue
            if nxt is not None and nxt.kind == N.UNIT_GOCT and nxt.stride == 2:
                continue           # only the pooled copy of this output is written (its stride-2 reader is the only one)
            pr = probes[name][j]
            got = eng.activation(a).numpy()
            flat = got.reshape(-1)
            s = flat[I.probe_indices(flat.size)]
            err = float(np.abs(s - np.array(pr["samples"], dtype=np.float32)).max()) / max(1.0, pr["absmax"])
            assert err <= P.UNIT_TOL, (name, j, err)
            checked += 1
    assert checked >= 20


@pytest.mark.parametrize("seed,kill", [(5, 0.4), (17, 0.75)])
def test_ilb_on_pruned_networks(emu_lib, tmp_path, seed, kill):
    """Randomly pruned channel plans (odd counts, single channels, empty branches) through the fused kernel."""
    import contextlib
    import io
    from sod100k_amd.model import csnet as M
    from test_unpruned_emu import _random_state
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.build_model(basic_split=[0.5, 0.5], expand=1.0, save_path=str(tmp_path))
    sd = _random_state(m, seed)
    g = torch.Generator().manual_seed(2000 + seed)
    for k in sd:
        if ('.bns.' in k or '.bn.' in k) and k.endswith('weight'):
            dead = torch.rand(sd[k].shape, generator=g) < kill
            sd[k] = torch.where(dead, torch.full_like(sd[k], 1e-6), sd[k])
    m.load_state_dict(sd)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, mask = M.finetune_model(m, save_path=str(tmp_path), base_layer_config=O.init_layers(20, [0.5, 0.5]), thres=1e-3)
        slim = M.build_model_with_weight(cfg, m, mask).eval()
    slim._lib = emu_lib
    x = torch.from_numpy(I.randn_batch(seed, 2, 64, 80))
    eng = slim.engine_for(x)
    eng.set_option(N.OPT_FUSE_ILB, 1)
    names = [eng.lib.csn_unit_kernel_name(eng.plan, u).decode() for u in range(eng.n_units)]
    assert "ilb_kernel" in names
    ssd = {k: v.clone() for k, v in slim.state_dict().items()}
    with torch.no_grad():
        ref = O.csnet_forward(cfg, ssd, x)
    y = slim(x)
    assert (y - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
