"""BASELINE config 0 ("test.py forward, batch 1, shipped checkpoint -- plumbing, no GPU"): the inference caller
sod100k_amd/tools/test.py end to end on this CPU-only box.  The product has no CPU path, so the kernels are the
emulated build of the same sources (tests/emu), injected where the caller would load libcsnet_hip.so; everything else
-- config merge, model.<ARCH> import, build_model, strict checkpoint load, normalise, forward, sigmoid -> uint8, PNG --
is the shipped code.  The written maps must equal the oracle's."""
import os

import numpy as np
import pytest
import torch

from oracle import csnet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inference_caller_writes_the_oracles_maps(emu_lib, tmp_path, monkeypatch):
    from PIL import Image
    from sod100k_amd import _native as N
    from sod100k_amd.configs import defaults
    from sod100k_amd.tools import test as T
    from sod100k_amd.model import csnet as M

    rng = np.random.default_rng(0)
    img_dir = tmp_path / "sal" / "TOY" / "images"
    img_dir.mkdir(parents=True)
    imgs = {}
    for name in ("a.png", "b.png"):
        im = (rng.random((224, 224, 3)) * 255).astype(np.uint8)
        Image.fromarray(im).save(img_dir / name)
        imgs[name] = im
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    cfg = defaults()
    cfg.merge_from_file(os.path.join(ROOT, "sod100k_amd", "configs", "csnet-L-x2.yml"))
    cfg.TEST.DATASET_PATH = str(tmp_path / "sal")
    cfg.TEST.DATASETS = ["TOY"]
    cfg.TEST.CHECKPOINT = man
    cfg.TEST.MODEL_CONFIG = man
    cfg.DATA.SAVEDIR = str(tmp_path / "results")
    # no GPU here: hand the caller the emulated build of the kernels instead of libcsnet_hip.so
    monkeypatch.setattr(N, "load", lambda: emu_lib)
    orig_init = M.CSNet.__init__

    def init_with_emu(self, *a, **kw):
        orig_init(self, *a, **kw)
        self._lib = emu_lib
    monkeypatch.setattr(M.CSNet, "__init__", init_with_emu)
    import importlib
    monkeypatch.setattr(importlib, "import_module", lambda name: M if name == "model.csnet" else __import__(name))
    T.run(cfg, batch=1, device="cpu")
    (out_dir,) = [d for d in (tmp_path / "results" / cfg.TASK).iterdir() if d.name.startswith("TOY_")]   # TOY_<epoch>
    sd = O.load_weights(man)
    lc = O.load_layer_config_json(man)
    mean = np.array([0.485, 0.456, 0.406], np.float32); std = np.array([0.229, 0.224, 0.225], np.float32)
    for name, im in imgs.items():
        got = np.asarray(Image.open(out_dir / name))
        x = torch.from_numpy(np.transpose((im / 255.0 - mean) / std, (2, 0, 1))[None].astype(np.float32))
        with torch.no_grad():
            ref = O.caller_postprocess(O.csnet_forward(lc, sd, x))
        assert got.shape == ref.shape and got.dtype == np.uint8
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1 and (got != ref).mean() < 5e-3


def test_csf_res2net_caller_writes_the_oracles_maps(emu_lib, tmp_path):
    """Solver.test counterpart (sod100k_amd/tools/csf_test.py) end to end on CPU: pictures of their own (non-octave)
    size, backbone through torch, decoder head through the emulated kernels; the PNGs equal the oracle's maps."""
    from PIL import Image
    from oracle import csf_oracle as CO
    from sod100k_amd.networks import csf_res2net as R
    from sod100k_amd.tools import csf_test as T

    rng = np.random.default_rng(1)
    root = tmp_path / "imgs"
    root.mkdir()
    sizes = {"p.jpg.png": (44, 60), "q.png": (37, 52)}
    for name, (h, w) in sizes.items():
        Image.fromarray((rng.random((h, w, 3)) * 255).astype(np.uint8)).save(root / name)
    sd = CO.synthetic_state()
    net = R.build_model()
    net.load_state_dict(sd, strict=True)
    net.eval()
    object.__setattr__(net, "_lib", emu_lib)
    T.test(net, list(sizes), str(root), str(tmp_path / "out"), device="cpu", lib=emu_lib)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    for name in sizes:
        im = np.asarray(Image.open(root / name).convert("RGB"), dtype=np.float32) / 255.0
        x = torch.from_numpy(np.transpose((im - mean) / std, (2, 0, 1))[None])
        with torch.no_grad():
            ref = torch.sigmoid(CO.csfnet_forward(sd, x)).squeeze().numpy()
        got = np.asarray(Image.open(tmp_path / "out" / (name[:-4] + "_sal_fuse.png")))
        exp = np.clip(np.rint(255.0 * ref), 0, 255).astype(np.uint8)
        assert got.shape == exp.shape and np.abs(got.astype(int) - exp.astype(int)).max() <= 1
        assert (got != exp).mean() <= 0.01
