#!/usr/bin/env python3
"""Inference caller, counterpart of the reference's CSNet/test.py (main 35-55, test 58-100).

    python -m sod100k_amd.tools.test --config sod100k_amd/configs/csnet-L-x2.yml [--batch 16]

Same flow: cfg merge -> ``import_module("model." + cfg.MODEL.ARCH).build_model(predefine=...)`` -> device ->
``simplesum`` -> strict ``load_state_dict`` -> per image: normalise (mean/std of test.py:68-69), ``model(x)``,
``predict[0]``, sigmoid, resize back, ``(p * 255).astype(uint8)``, PNG.  Differences, all host side: images are
read/resized with PIL (skimage is not a dependency; host IO is outside the accelerated path)
and images are pushed through the plan in batches (the HIP plan is compiled per (B, H, W)).
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


def preprocess(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """H x W x 3 float image in [0,1] -> 3 x h x w float32, ImageNet-normalised (test.py:72-86)."""
    from PIL import Image
    if img.shape[:2] != (h, w):
        img = np.asarray(Image.fromarray((img * 255).astype(np.uint8)).resize((w, h), Image.BILINEAR)) / 255.0
    return np.transpose((img - MEAN) / STD, (2, 0, 1)).astype(np.float32)


def resize_hw(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """H x W x 3 float image in [0,1] at the network size (bilinear, host side)."""
    from PIL import Image
    if img.shape[:2] == (h, w):
        return img
    return np.asarray(Image.fromarray((img * 255).astype(np.uint8)).resize((w, h), Image.BILINEAR)) / 255.0


def postprocess(logits: torch.Tensor, h: int, w: int) -> np.ndarray:
    """1 x H x W logits -> uint8 saliency map of the original size (test.py:91-96)."""
    from PIL import Image
    p = torch.sigmoid(logits.squeeze(0)).cpu().numpy()
    if p.shape != (h, w):
        p = np.asarray(Image.fromarray(p).resize((w, h), Image.BILINEAR))
    return (p * 255).astype(np.uint8)


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
    H, W = cfg.TEST.IMAGE_H or 224, cfg.TEST.IMAGE_W or 224
    for dataset in cfg.TEST.DATASETS:
        img_dir = os.path.join(cfg.TEST.DATASET_PATH, dataset, 'images')
        if not os.path.isdir(img_dir):
            print("dataset directory", img_dir, "not found -- nothing to do")
            continue
        out_dir = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK or cfg.MODEL.ARCH, dataset + '_' + str(ck['epoch']))
        os.makedirs(out_dir, exist_ok=True)
        names = sorted(os.listdir(img_dir))
        for i in range(0, len(names), batch):
            chunk = names[i:i + batch]
            imgs = [np.asarray(Image.open(os.path.join(img_dir, n)).convert("RGB")) / 255.0 for n in chunk]
            # host: resize to the network size (IO side); device: normalise + NCHW pack, forward, sigmoid -> uint8
            hwc = np.stack([resize_hw(im, H, W) for im in imgs] + [np.zeros((H, W, 3), np.float32)] * (batch - len(chunk)))
            with torch.no_grad():
                x = E.normalize_nchw(N.load(), torch.from_numpy(hwc.astype(np.float32)).to(device))
                pred = model(x)
                maps = E.saliency_u8(N.load(), pred).cpu().numpy()
            for k, (n, im, lg) in enumerate(zip(chunk, imgs, pred)):
                u8 = maps[k, 0] if im.shape[:2] == (H, W) else postprocess(lg, *im.shape[:2])
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
