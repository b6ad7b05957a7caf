"""The CPU oracle against the goldens frozen from the REFERENCE implementation (oracle/make_goldens.py).

These pin the oracle: logits (G2), per-unit probes (G3), op micro-goldens (G4), one
training step incl. the dynamic-weight-decay penalty and Adam (G5), the 2-shard DP emulation (G7) and the
uint8 saliency map of the caller (G9).  The oracle calls the same ATen CPU kernels as the reference, so
agreement is expected at (or very near) the bit level; tolerances are 1e-5 (fp32 floor).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import csnet_oracle as O, inputs as I

from conftest import GOLD

TOL = 1e-5


def _fwd(manifest, x, taps=None):
    sd = O.load_weights(manifest)
    with torch.no_grad():
        return O.csnet_forward(O.load_layer_config_json(manifest), sd, x, taps=taps), sd


def test_g2_logits_randn(x2_manifest):
    y, _ = _fwd(x2_manifest, torch.from_numpy(I.randn_batch(0, 2)))
    g = np.load(os.path.join(GOLD, "g2_logits_x2_randn_b2.npy"))
    assert np.abs(y.numpy() - g).max() <= TOL


def test_g2_logits_nonsquare(x2_manifest):
    y, _ = _fwd(x2_manifest, torch.from_numpy(I.randn_batch(3, 2, 96, 160)))
    g = np.load(os.path.join(GOLD, "g2_logits_x2_randn_b2_96x160.npy"))
    assert np.abs(y.numpy() - g).max() <= TOL


def test_g2_logits_x1(x1_manifest):
    y, _ = _fwd(x1_manifest, torch.from_numpy(I.randn_batch(0, 2))[:1])
    g = np.load(os.path.join(GOLD, "g2_logits_x1_randn_b1.npy"))
    assert np.abs(y.numpy() - g).max() <= TOL


def test_g2_g9_image_and_uint8_map(x2_manifest):
    y, _ = _fwd(x2_manifest, torch.from_numpy(I.image_like()))
    g = np.load(os.path.join(GOLD, "g2_logits_x2_image.npy"))
    assert np.abs(y.numpy() - g).max() <= TOL
    u8 = O.caller_postprocess(y)
    g9 = np.load(os.path.join(GOLD, "g9_uint8_x2_image.npy"))
    assert u8.dtype == np.uint8 and u8.shape == g9.shape
    assert np.abs(u8.astype(int) - g9.astype(int)).max() <= 1 and (u8 != g9).mean() < 1e-3


def test_g3_unit_probes(x2_manifest):
    taps = {}
    _fwd(x2_manifest, torch.from_numpy(I.randn_batch(0, 2)), taps)
    probes = json.load(open(os.path.join(GOLD, "g3_unit_probes_x2.json")))
    assert len(probes) == 60
    for name, plist in probes.items():
        for j, p in enumerate(plist):
            if p is None:
                assert taps[name][j] is None
                continue
            t = taps[name][j].numpy().reshape(-1)
            assert list(taps[name][j].shape) == p["shape"]
            assert np.abs(t[I.probe_indices(t.size)] - np.array(p["samples"], np.float32)).max() <= TOL


def test_g4_ops():
    meta = json.load(open(os.path.join(GOLD, "g4_ops_meta.json")))
    arrs = np.load(os.path.join(GOLD, "g4_ops.npz"))
    for tag, mt in meta.items():
        sd = {k[len(tag) + 4:]: torch.from_numpy(arrs[k]) for k in arrs.files if k.startswith(tag + "/sd/")}
        with torch.no_grad():
            if mt["kind"] == "cbr":
                xs = [torch.from_numpy(arrs[f"{tag}/x{i}"]) for i in range(mt["n_in"])]
                ain = (np.array(mt["cin"], dtype=np.float64) / sum(mt["cin"])).tolist()
                aout = (np.array(mt["cout"], dtype=np.float64) / sum(mt["cout"])).tolist()
                sdp = {"m." + k: v for k, v in sd.items()}
                ys = O.goct_cbr(xs, sdp, "m", ain, aout, mt["k"], mt["stride"], False)
                ys = ys if isinstance(ys, list) else [ys]
            elif mt["kind"] == "dw":
                xs = [torch.from_numpy(arrs[f"{tag}/x{i}"]) for i in range(len(mt["ch"]))]
                ys = O.simplified_cbr(xs, {"m." + k: v for k, v in sd.items()}, "m", False)
            elif mt["kind"] == "ms":
                ys = [O.ms_block(torch.from_numpy(arrs[f"{tag}/x0"]), {"m." + k: v for k, v in sd.items()}, "m",
                                 mt["dil"], False)]
            else:
                x = torch.from_numpy(arrs[f"{tag}/x0"])
                w, b = torch.from_numpy(arrs[f"{tag}/w"]), torch.from_numpy(arrs[f"{tag}/b"])
                y = torch.nn.functional.conv2d(x, w, b)
                ys = [torch.nn.functional.interpolate(y, (2 * x.shape[2], 2 * x.shape[3]), mode="bilinear",
                                                      align_corners=False)]
        for j, y in enumerate(ys):
            assert np.abs(y.numpy() - arrs[f"{tag}/y{j}"]).max() <= TOL, tag


@pytest.mark.parametrize("idx", [0, 1])
def test_g5_train_step(x2_manifest, idx):
    rec = json.load(open(os.path.join(GOLD, "g5_g7_train_step.json")))[idx]
    sd = O.load_weights(x2_manifest)
    lc = O.load_layer_config_json(x2_manifest)
    x = torch.from_numpy(I.randn_batch(10, 4))
    t = torch.from_numpy(I.binary_target(11, 4))
    ef = 2 if rec["expandflop"] is None else rec["expandflop"]
    r = O.train_step(lc, sd, x, t, expandflop=ef, batchsize=4)
    assert abs(r["loss_bce"] - rec["bce"][0]) <= 1e-6
    assert abs(r["penalty"] - rec["penalty"][0]) <= 1e-6 * max(1.0, abs(rec["penalty"][0]))
    normal, picked = O.param_groups(list(rec["grad_l2"].keys()))
    assert (len(picked), len(normal)) == (rec["n_picked"], rec["n_normal"]) == (66, 353)
    for n, v in rec["grad_l2"].items():
        assert abs(float(r["grads"][n].double().norm()) - v) <= 1e-5 * max(v, 1e-3), n
    for n, v in rec["param_after"].items():
        assert abs(float(sd[n].double().sum()) - v["sum"]) <= 1e-5 * max(abs(v["sum"]), 1.0), n
    for n, v in rec["bn_after_rank0"].items():
        assert abs(float(sd[n].double().sum()) - v["sum"]) <= 1e-5 * max(abs(v["sum"]), 1.0), n


def test_g7_dp_two_shards(x2_manifest):
    """Gradient averaging over two independently normalised shards == the reference DP emulation."""
    rec = json.load(open(os.path.join(GOLD, "g5_g7_train_step.json")))[2]
    lc = O.load_layer_config_json(x2_manifest)
    x = torch.from_numpy(I.randn_batch(10, 4))
    t = torch.from_numpy(I.binary_target(11, 4))
    grads = None
    for s in range(2):
        sd = O.load_weights(x2_manifest)
        r = O.train_step(lc, sd, x[2 * s:2 * s + 2], t[2 * s:2 * s + 2], expandflop=1.0, batchsize=2, lr=0.0, wd=0.0)
        assert abs(r["loss_bce"] - rec["bce"][s]) <= 1e-6
        assert abs(r["penalty"] - rec["penalty"][s]) <= 1e-6 * max(1.0, abs(rec["penalty"][s]))
        grads = r["grads"] if grads is None else {k: grads[k] + v for k, v in r["grads"].items()}
    for n, v in rec["grad_l2"].items():
        assert abs(float((grads[n] / 2).double().norm()) - v) <= 1e-5 * max(v, 1e-3), n


def test_init_layers_matches_reference_manifest():
    g6 = json.load(open(os.path.join(GOLD, "g6_simplesum_keys.json")))
    lc = O.init_layers(40, [0.5, 0.5])
    assert lc[-1] == [3, 4, 6, 4] and len(lc) == 22
    assert g6["init_e2.0_s2"]["params"] == 788631


# ---- G8: CSF+Res2Net (oracle/csf_oracle.py against the reference's CSFNet, oracle/make_golden_csf.py) -------------
@pytest.mark.parametrize("name", ["96x128", "100x76", "352"])
def test_g8_csf_logits_and_probes(name):
    from oracle import csf_oracle as CO
    meta = json.load(open(os.path.join(GOLD, "g8_csf_probes.json")))["cases"][name]
    b, _, h, w = meta["shape"]
    x = torch.from_numpy(I.randn_batch(meta["seed"], b, h, w))
    probes = {}
    with torch.no_grad():
        y = CO.csfnet_forward(CO.synthetic_state(), x, probes=probes)
    g = np.load(os.path.join(GOLD, f"g8_csf_logits_{name}.npy"))
    assert np.abs(y.numpy() - g).max() <= TOL
    for key in ("features", "fuse", "ms", "fuse1x1"):
        for t, p in zip(probes[key], meta[key]):
            q = I.probe(t.numpy())
            assert q["shape"] == p["shape"]
            assert abs(q["l2"] - p["l2"]) <= 1e-5 * max(1.0, p["l2"]) and np.abs(np.array(q["samples"]) - np.array(p["samples"])).max() <= 1e-5 * max(1.0, p["absmax"])
