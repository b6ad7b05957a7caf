"""Device resizes on the CPU emulation + the inference caller at the pictures' own resolution (TEST.IMAGE_H/W = 0)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import csnet_oracle as O

import resize_cases as RC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resize_bilinear(emu_lib):
    RC.check_resize_bilinear(emu_lib, torch.device("cpu"))


def test_resize_pre_post(emu_lib):
    RC.check_pre_post(emu_lib, torch.device("cpu"))


def test_caller_native_resolution_and_fixed_size(emu_lib, x2_manifest):
    """run_pictures: pictures of different sizes, (a) at the configured 224 x 224, (b) with IMAGE_H/W = 0 at their own size
    rounded up to multiples of 16 (test.py:76-85); maps against the oracle with skimage's resize rule (scipy.ndimage.zoom,
    resize_cases.sk_resize) either side."""
    import parity_cases as P
    from sod100k_amd.tools import test as T
    m, sd = P.make_model(emu_lib, x2_manifest, torch.device("cpu"))
    rng = np.random.default_rng(3)
    imgs = [rng.random((100, 140, 3)), rng.random((64, 64, 3)), rng.random((100, 140, 3))]
    lc = O.load_layer_config_json(x2_manifest)
    for cfg_hw in ((224, 224), (0, 0)):
        maps = T.run_pictures(m, imgs, cfg_hw[0], cfg_hw[1], batch=2, device="cpu", lib=emu_lib)
        for im, got in zip(imgs, maps):
            h, w = im.shape[:2]
            H, W = T.network_size(h, w, *cfg_hw)
            if cfg_hw == (0, 0):
                assert (H, W) == (-(-h // 16) * 16, -(-w // 16) * 16)
            t = torch.from_numpy(im.astype(np.float32)).permute(2, 0, 1)[None]
            x = (RC.sk_resize(t[0], H, W)[None] - RC.MEAN) / RC.STD
            with torch.no_grad():
                y = O.csnet_forward(lc, sd, x)
            p = RC.sk_resize(torch.sigmoid(y)[0], h, w)[0]
            want = (p.numpy() * 255).astype(np.uint8)
            assert got.shape == (h, w) and got.dtype == np.uint8
            d = np.abs(got.astype(int) - want.astype(int))
            assert d.max() <= 1 and (d != 0).mean() < 1e-2
