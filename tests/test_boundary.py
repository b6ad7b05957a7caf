"""Host-side boundary: module tree / state_dict keys, counters, C-ABI exports, error behaviour (no GPU)."""
import contextlib
import ctypes
import io
import json
import os
import re

import pytest
import torch

from sod100k_amd import _native as N
from sod100k_amd.checkpoint import load_manifest_state_dict
from sod100k_amd.model import csnet as M
from sod100k_amd.model.utils.simplesum_octconv import simplesum

from conftest import GOLD, ROOT


def _keys(m):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]


@pytest.mark.parametrize("name", ["csnet-L-x2", "csnet-L-x1"])
def test_state_dict_manifest_and_counters(name):
    g6 = json.load(open(os.path.join(GOLD, "g6_simplesum_keys.json")))[name]
    m = M.build_model(predefine=os.path.join(ROOT, "sod100k_amd", "data", name + ".json"))
    assert _keys(m) == g6["keys"]                      # same names, order, shapes, dtypes (737 for x2)
    with contextlib.redirect_stdout(io.StringIO()):
        params, flops = simplesum(m, inputsize=(3, 224, 224), device=-1)
    assert (params, int(flops)) == (g6["params"], g6["flops"])     # 140,894 / 716,713,200 for x2
    sd = load_manifest_state_dict(os.path.join(ROOT, "sod100k_amd", "data", name + ".json"))
    assert m.load_state_dict(sd, strict=True).missing_keys == []


@pytest.mark.parametrize("tag,kw", [("init_e1.0_s2", dict(basic_split=[0.5, 0.5], expand=1.0)),
                                    ("init_e2.0_s2", dict(basic_split=[0.5, 0.5], expand=2.0))])
def test_init_configs(tag, kw):
    g6 = json.load(open(os.path.join(GOLD, "g6_simplesum_keys.json")))[tag]
    m = M.build_model(**kw)
    assert _keys(m) == g6["keys"]
    with contextlib.redirect_stdout(io.StringIO()):
        params, flops = simplesum(m, inputsize=(3, 224, 224), device=-1)
    assert (params, int(flops)) == (g6["params"], g6["flops"])


def test_param_group_name_matching():
    """train.py:101-107 picks 66 BN weights by NAME; the names must survive."""
    m = M.build_model(predefine=os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json"))
    picked = [n for n, _ in m.named_parameters()
              if 'stage' in n and ('conv1x1.bns' in n or 'conv3x3_1.bns' in n) and 'weight' in n]
    assert len(picked) == 66 and len(list(m.parameters())) == 419
    assert sum(isinstance(x, M.gOctaveCBR) for x in m.modules()) == 20
    assert sum(isinstance(x, torch.nn.BatchNorm2d) for x in m.modules()) == 106


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "csnet_hip.h")).read()
    hdr += open(os.path.join(ROOT, "include", "csf_hip.h")).read()
    declared = set(re.findall(r"\b(cs[nf]_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.EXPORTS), declared ^ set(N.EXPORTS)
    if not os.path.exists(N.LIB_PATH):
        pytest.skip("libcsnet_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(N.LIB_PATH)
    for s in N.EXPORTS:
        assert hasattr(lib, s), s
    assert lib.csn_abi_version() == N.ABI_VERSION


def test_no_cpu_fallback_and_clear_errors():
    m = M.build_model(predefine=os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")).eval()
    with pytest.raises(RuntimeError, match="ROCm"):
        m(torch.zeros(1, 3, 32, 32))                  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 32, 32))
    with pytest.raises(RuntimeError, match="parameter container"):
        m.stage0[0]([torch.zeros(1, 3, 32, 32)])


def test_bad_plans_are_rejected(emu_lib):
    u = N.new_unit(N.UNIT_DW)
    u.n_in = u.n_out = 1
    u.cin[0] = u.cout[0] = 4
    u.in_act[0], u.out_act[0] = 0, 1
    ua = (N.UnitDesc * 1)(u)
    aa = (N.ActDesc * 2)(N.ActDesc(4, 0), N.ActDesc(4, 0))
    plan = ctypes.c_void_p()
    assert emu_lib.csn_plan_create(ua, 1, aa, 2, 1, 32, 32, 0, ctypes.byref(plan)) == 1   # missing BN offsets
    assert emu_lib.csn_plan_create(ua, 1, aa, 2, 1, 30, 32, 0, ctypes.byref(plan)) == 1   # H not multiple of 16
    assert emu_lib.csn_strerror(1).decode().startswith("invalid")


def test_csfnet_state_dict_matches_reference_manifest():
    """CSF+Res2Net drop-in: same state_dict names / shapes / dtypes as the reference's CSFNet (G8 manifest)."""
    from sod100k_amd.networks import csf_res2net as R
    from oracle import csf_oracle as CO
    keys = json.load(open(os.path.join(GOLD, "g8_csf_probes.json")))["keys"]
    net = R.build_model()
    mine = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    assert list(mine) == list(keys) and mine == keys
    assert net.load_state_dict(CO.synthetic_state(), strict=True).missing_keys == []
    d = net.describe_head(net._ensure_arena().offsets)
    assert list(d.cin) == [256, 512, 1024, 2048] and list(d.cmid) == [128, 256, 512, 512]
    assert [list(r) for r in d.ms_split] == [[25, 25, 25, 25, 28], [51, 51, 51, 51, 52], [102, 102, 102, 102, 104]] * 1 + [[102, 102, 102, 102, 104]]


def test_flat_tile_index_arithmetic_is_exact_for_every_supported_plane():
    """pw4 / c3q flat tiles (k_pw4.hip, k_c3q.hip): row = p / W through a float reciprocal and a +-1 correction.  Exact for every
    plane width up to 2048 and every pixel index up to 2048 x 2048 + 63 (float32 model of the device arithmetic)."""
    import numpy as np
    rng = np.random.default_rng(0)
    for W in list(range(1, 130)) + [224, 448, 1000, 1023, 1024, 1025, 2047, 2048]:
        pmax = min(W * 2048 + 63, (1 << 22) - 1)
        p = np.unique(np.concatenate([np.arange(0, min(pmax, 70000)), rng.integers(0, pmax + 1, 20000),
                                      np.arange(max(pmax - 5000, 0), pmax + 1)])).astype(np.int64)
        inv = np.float32(1.0) / np.float32(W)
        q = (p.astype(np.float32) * inv).astype(np.int64)
        q = q - (q * W > p)
        q = q + ((q + 1) * W <= p)
        assert np.array_equal(q, p // W), W
