#!/usr/bin/env python3
"""Prune-and-finetune caller, counterpart of the reference's CSNet_training/finetune.py (main 84-207, train 210-254).

    python -m sod100k_amd.tools.finetune --config CFG.yml --epoch N [--synthetic 8]

Flow of finetune.py:84-207: build the network of ``layer_configs/layer_config_0.bin``, load the training checkpoint
``checkpoint/checkpoint_epoch<N>.pth.tar`` (strict), prune it with
``build_model(epoch, model=trained, predefine=..., finetune=True, finetune_thres=cfg.FINETUNE.THRES,
load_weight='FINETUNE')`` (channels whose |BN gamma| fell under the threshold go; csnet.py:821-945), then train the slim
network with the FINETUNE.SOLVER settings (Adam betas (0.9, 0.99) / eps 1e-8 with one weight decay, MultiStepLR or
cosine schedule stepped at the START of every epoch), ``val()`` after every epoch, one checkpoint per epoch.

Difference: the reference creates its optimizer on the UN-pruned model (finetune.py:110-124) and then rebuilds the
model (159-167), so its ``optimizer.step()`` no longer touches the network being trained; here the optimizer state is
created for the slim network, which is what the recipe means.  Steps run on the flat-buffer ``FusedTrainer``
(csn_forward_train / csn_backward / csn_adam_step); the penalty is off (finetune.py:218-226 has no FLOPs term).
"""
import argparse
import importlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sod100k_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from sod100k_amd.configs import defaults                  # noqa: E402
from sod100k_amd.tools import train as T                  # noqa: E402


def finetune_lr(solver, epoch):
    """lr_scheduler.MultiStepLR(gamma=0.1) / CosineAnnealingLR(T_max=MAX_EPOCHS, eta_min=0) stepped once before the
    epoch's first iteration (finetune.py:142-154,174-177)."""
    if not solver.ADJUST_STEP:
        return solver.LR
    if solver.LR_SCHEDULER == 'step':
        return T.multistep_lr(solver.LR, solver.STEPS, epoch)
    if solver.LR_SCHEDULER == 'cosine':
        return solver.LR * 0.5 * (1.0 + math.cos(math.pi * (epoch + 1) / solver.MAX_EPOCHS))
    raise ValueError("Unsupported scheduler.")


def run(cfg, epoch, device="cuda", synthetic=0, max_steps=0, val_batches=None, lib=None):
    model_lib = importlib.import_module("model." + cfg.MODEL.ARCH)
    layer_config_dir = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK, 'layer_configs')
    predefine_file = os.path.join(layer_config_dir, "layer_config_0.bin")
    model = model_lib.build_model(predefine=predefine_file)                              # finetune.py:96-99
    ckpt = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK, 'checkpoint', 'checkpoint_epoch{}.pth.tar'.format(epoch))
    if not os.path.isfile(ckpt):
        print("=> no checkpoint found at '{}'".format(ckpt))
        return None
    checkpoint = torch.load(ckpt, map_location="cpu")
    from_epoch = checkpoint['epoch']
    model.load_state_dict(checkpoint['state_dict'])
    model = model_lib.build_model(epoch=from_epoch, basic_split=cfg.MODEL.BASIC_SPLIT, model=model,
                                  save_path=layer_config_dir, predefine=predefine_file, finetune=True,
                                  finetune_thres=cfg.FINETUNE.THRES, load_weight='FINETUNE')   # finetune.py:159-167
    if lib is not None:
        model._lib = lib
    model = model.to(device).train()
    solver = cfg.FINETUNE.SOLVER
    if solver.METHOD != 'Adam':
        print("WARNING: Method not implmented.")
        return None
    # finetune.py:118-124: ONE parameter group, weight decay on every parameter
    trainer = T.FusedTrainer(model, lr=solver.LR, weight_decay=solver.WEIGHT_DECAY, flops_weight=0.0,
                             batchsize=cfg.DATA.BATCH_SIZE, lib=lib)
    trainer.wd.fill_(float(solver.WEIGHT_DECAY))
    out_dir = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK, 'finetune_checkpoint')
    os.makedirs(out_dir, exist_ok=True)
    best_mae, best_epoch, done = 1000000, -1, 0
    for ep in range(solver.MAX_EPOCHS):
        trainer.lr = finetune_lr(solver, ep)
        if synthetic <= 0:
            print("dataset loading (prepare_data.py) is host IO outside this build; use --synthetic N")
            return model
        for i, (x, t) in enumerate(T.synthetic_batches(synthetic, cfg.DATA.BATCH_SIZE, cfg.DATA.IMAGE_H, cfg.DATA.IMAGE_W,
                                                       device, seed=ep)):
            loss, _ = trainer.step(x, t)
            if i % cfg.PRINT_FREQ == 0:
                print(f"Epoch: [{ep}][{i}/{synthetic}] Loss {float(loss):.4f} lr {trainer.lr:g}")
            done += 1
            if max_steps and done >= max_steps:
                break
        mae = T.val(model, val_batches, lib=lib) if val_batches is not None else float("nan")
        if mae < best_mae:
            best_mae, best_epoch = mae, ep + 1
        print(" epoch: " + str(ep + 1) + " mae: " + str(mae) + " best_epoch: " + str(best_epoch) + " best_mae: " + str(best_mae))
        torch.save({'epoch': ep + 1, 'arch': cfg.MODEL.ARCH, 'state_dict': model.state_dict()},
                   os.path.join(out_dir, 'checkpoint_epoch{}.pth.tar'.format(ep + 1)))
        if max_steps and done >= max_steps:
            break
    return model


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default=os.path.join(ROOT, "sod100k_amd", "configs", "csnet-L-x2_train.yml"))
    ap.add_argument("--epoch", type=int, required=True, help="training epoch whose checkpoint is pruned (finetune.py --epoch)")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--max-steps", type=int, default=0)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    cfg = defaults()
    cfg.merge_from_file(args.config)
    if args.opts:
        cfg.merge_from_list(args.opts)
    run(cfg, args.epoch, device=args.device, synthetic=args.synthetic, max_steps=args.max_steps)


if __name__ == "__main__":
    main()
