#!/usr/bin/env python3
"""Inference caller, counterpart of the reference's CSNet/test.py (main 35-55, test 58-100).

    python -m sod100k_amd.tools.test --config sod100k_amd/configs/csnet-L-x2.yml [--batch 16]

Same flow: cfg merge -> ``import_module("model." + cfg.MODEL.ARCH).build_model(predefine=...)`` -> device ->
``simplesum`` -> strict ``load_state_dict`` -> per image: normalise (mean/std of test.py:68-69), ``model(x)``,
``predict[0]``, sigmoid, resize back, ``(p * 255).astype(uint8)``, PNG.  Both resizes run on the device as float,
non-antialiased bilinear interpolation with half-pixel centres (csn_resize_normalize_nchw / csn_saliency_resize_u8 = the
reference's skimage ``resize(mode='reflect', anti_aliasing=False)``, test.py:76-85,94-96); with ``TEST.IMAGE_H/W == 0`` the
pictures run at their own size rounded up to multiples of 16 (test.py:80-85).  Differences, all host side: files are
decoded with PIL (skimage is not a dependency) and pictures of the same network size are pushed through the plan in
batches (the HIP plan is compiled per (B, H, W)).
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sod100k_amd")):       # ``model.csnet`` resolves to sod100k_amd/model/csnet.py
    if p not in sys.path:
        sys.path.insert(0, p)

from sod100k_amd import _native as N, engine as E         # noqa: E402
from sod100k_amd.checkpoint import load_checkpoint       # noqa: E402
from sod100k_amd.configs import defaults                  # noqa: E402

MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])


def network_size(h: int, w: int, cfg_h: int, cfg_w: int):
    """test.py:76-85: the configured size, or the picture's own size rounded up to multiples of 16."""
    if cfg_h != 0 and cfg_w != 0:
        return cfg_h, cfg_w
    return -(-h // 16) * 16, -(-w // 16) * 16


def run_pictures(model, imgs, cfg_h, cfg_w, batch, device, lib=None):
    """Saliency maps (uint8, each at its picture's own size) of a list of H x W x 3 float pictures in [0,1]: pictures
    are grouped by network size; resize + normalise, forward, sigmoid + resize back + quantise all run on the device."""
    lib = lib if lib is not None else (getattr(model, "_lib", None) or N.load())
    groups = {}
    for idx, im in enumerate(imgs):
        key = im.shape[:2] + network_size(im.shape[0], im.shape[1], cfg_h, cfg_w)
        groups.setdefault(key, []).append(idx)
    out = [None] * len(imgs)
    for (h, w, H, W), members in groups.items():
        for i in range(0, len(members), batch):
            chunk = members[i:i + batch]      # a short last chunk runs at its own batch size (no padded pictures)
            hwc = np.stack([imgs[k] for k in chunk])
            with torch.no_grad():
                x = E.resize_normalize_nchw(lib, torch.from_numpy(hwc.astype(np.float32)).to(device), H, W)
                pred = model(x)
                for j, k in enumerate(chunk):
                    out[k] = E.saliency_resize_u8(lib, pred[j, 0], h, w).cpu().numpy()
    return out


def run(cfg, batch: int = 16, device: str = "cuda"):
    from PIL import Image
    model_lib = importlib.import_module("model." + cfg.MODEL.ARCH)                    # test.py:37
    from model.utils.simplesum_octconv import simplesum
    model = model_lib.build_model(predefine=cfg.TEST.MODEL_CONFIG)
    model = model.to(device)
    prams, flops = simplesum(model, inputsize=(3, 224, 224), device=0)
    print('  + Number of params: %.4fM' % (prams / 1e6))
    print('  + Number of FLOPs: %.4fG' % (flops / 1e9))
    if not os.path.isfile(cfg.TEST.CHECKPOINT):
        print(cfg.TEST.CHECKPOINT, "Not found.")
        return
    ck = load_checkpoint(cfg.TEST.CHECKPOINT)
    model.load_state_dict(ck['state_dict'])
    model.eval()
    for dataset in cfg.TEST.DATASETS:
        img_dir = os.path.join(cfg.TEST.DATASET_PATH, dataset, 'images')
        if not os.path.isdir(img_dir):
            print("dataset directory", img_dir, "not found -- nothing to do")
            continue
        out_dir = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK or cfg.MODEL.ARCH, dataset + '_' + str(ck['epoch']))
        os.makedirs(out_dir, exist_ok=True)
        names = sorted(os.listdir(img_dir))
        for i in range(0, len(names), 4 * batch):
            chunk = names[i:i + 4 * batch]
            imgs = [np.asarray(Image.open(os.path.join(img_dir, n)).convert("RGB")) / 255.0 for n in chunk]
            maps = run_pictures(model, imgs, cfg.TEST.IMAGE_H, cfg.TEST.IMAGE_W, batch, device)
            for n, u8 in zip(chunk, maps):
                Image.fromarray(u8).save(os.path.join(out_dir, n[:-4] + '.png'))
        print('Dataset: {}, {} images'.format(dataset, len(names)))


def main():
    ap = argparse.ArgumentParser(description='CSNet SOD inference on MI355X')
    ap.add_argument("--config", required=True, metavar="FILE")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    assert os.path.isfile(args.config)
    cfg = defaults()
    cfg.merge_from_file(args.config)
    if cfg.TASK == '':
        cfg.TASK = cfg.MODEL.ARCH
    run(cfg, args.batch, args.device)


if __name__ == '__main__':
    main()
