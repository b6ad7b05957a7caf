#!/usr/bin/env python3
"""Inference caller of the CSF+Res2Net network, counterpart of ``Solver.test`` (CSF+Res2Net/solver.py:61-77) with the
test loader of dataset/dataset.py:45-62,93-106:

    python -m sod100k_amd.tools.csf_test --model final.pth --test_root DIR --test_list LIST --test_fold OUT

Same flow: ``build_model()`` -> ``load_state_dict(torch.load(model), strict=False)`` -> eval -> per picture (batch 1, its
own size, no resize): RGB / 255, ImageNet mean / std, CHW -> ``net(images)`` -> ``255 * sigmoid`` -> ``<name>_sal_fuse.png``.
Differences, all host side: pictures are read / written with PIL (OpenCV is not a dependency; ``cv2.imwrite`` of a float
array rounds to nearest and saturates, reproduced with ``rint`` + ``clip``), the normalisation runs on the device
(``csn_normalize_nchw``).  The network itself is ``sod100k_amd/networks/csf_res2net.py`` (HIP decoder head).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sod100k_amd")):       # ``networks.csf_res2net`` resolves to sod100k_amd/networks
    if p not in sys.path:
        sys.path.insert(0, p)

from sod100k_amd import _native as N, engine as E         # noqa: E402


def load_image_test(path):
    """dataset.py:93-106: RGB float32 H x W x 3 in [0,1] (normalised on the device) and the picture's size."""
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return im, im.shape[:2]


def to_png(pred):
    """``cv2.imwrite(path, 255 * pred)`` with a float array: saturate_cast<uchar> = round to nearest even, clip."""
    return np.clip(np.rint(255.0 * pred), 0, 255).astype(np.uint8)


def test(net, names, test_root, test_fold, device="cuda", lib=None):
    from PIL import Image
    os.makedirs(test_fold, exist_ok=True)
    lib = lib if lib is not None else (getattr(net, "_lib", None) or N.load())
    t0 = time.time()
    for name in names:
        im, _ = load_image_test(os.path.join(test_root, name))
        x = E.normalize_nchw(lib, torch.from_numpy(im).unsqueeze(0).to(device))
        with torch.no_grad():
            preds = net(x)
        pred = np.squeeze(torch.sigmoid(preds).cpu().numpy())
        Image.fromarray(to_png(pred)).save(os.path.join(test_fold, name[:-4] + "_sal_fuse.png"))
    print("Speed: %f FPS" % (len(names) / max(time.time() - t0, 1e-9)))
    print("Test Done!")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--model", required=True, help="state_dict saved by the reference's solver (torch.save)")
    ap.add_argument("--test_root", required=True)
    ap.add_argument("--test_list", required=True)
    ap.add_argument("--test_fold", required=True)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    from networks.csf_res2net import build_model
    net = build_model()
    net.load_state_dict(torch.load(args.model, map_location="cpu"), strict=False)       # solver.py:28-31
    net = net.to(args.device).eval()
    names = [l.strip() for l in open(args.test_list) if l.strip()]
    test(net, names, args.test_root, args.test_fold, device=args.device)


if __name__ == "__main__":
    main()
