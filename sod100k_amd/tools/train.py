#!/usr/bin/env python3
"""Training caller, counterpart of the reference's CSNet_training/train.py (main 67-181, train 184-247).

    python -m sod100k_amd.tools.train --config sod100k_amd/configs/csnet-L-x2_train.yml [--synthetic 8]

Same step as train.py:203-216:

    output = model(input)                                   # train-mode forward (csn_forward_train)
    loss   = BCEWithLogits(output, target)                  # csn_bce_with_logits
    loss  += FLOPS.WEIGHT * model.get_flops()               # penalty fused into the BN pass of the forward
    optimizer.zero_grad(); loss.backward(); optimizer.step()   # csn_backward + csn_adam_step
    model.clear_flops()

with the reference's two Adam parameter groups (train.py:97-123; the typo'd name test is reproduced, so
``conv3x3_2.bns`` keeps its weight decay), betas (0.9, 0.99), eps 1e-8 and MultiStepLR(gamma 0.1) stepped at the
start of every epoch (train.py:146-157).  ``FusedTrainer`` drives the kernels on flat buffers (one gradient arena,
one Adam state, no per-parameter Python loop); ``reference_style_step`` is the unchanged
``loss.backward(); optimizer.step()`` sequence over the autograd seam for callers that keep their own loop.
Multi-GPU: one process per GPU, per-GPU BN statistics (the reference has no SyncBN) and ONE collective per
step -- all-reduce(sum)/world of the flat gradient arena (sod100k_amd/dist.py) -- before the Adam kernel.
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sod100k_amd")):       # ``model.csnet`` resolves to sod100k_amd/model/csnet.py
    if p not in sys.path:
        sys.path.insert(0, p)

from sod100k_amd import _native as N                      # noqa: E402
from sod100k_amd.configs import defaults                  # noqa: E402


def is_picked(pname: str) -> bool:
    """Parameter-group rule of train.py:101-107 (weight decay 0 for the BN weights that the penalty shrinks)."""
    return 'stage' in pname and ('conv1x1.bns' in pname or 'conv3x3_1.bns' in pname or 'conv3x3_1.bns' in pname) \
        and 'weight' in pname


class FusedTrainer:
    """One train step = forward(train) + BCE + backward + [gradient all-reduce] + Adam, all on flat device buffers."""

    def __init__(self, model, lr=1e-4, weight_decay=5e-3, betas=(0.9, 0.99), eps=1e-8, flops_weight=0.0,
                 batchsize=None, lib=None):
        self.model = model
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.flops_weight = float(flops_weight)
        self.batchsize = batchsize
        arena = model._ensure_arena()
        self.n = arena.n_param_floats
        dev = arena.flat.device
        self.lib = lib if lib is not None else (model._lib or N.load())
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.grad)
        self.v = torch.zeros_like(self.grad)
        wd = torch.zeros(self.n, dtype=torch.float32)
        for name, p in model.named_parameters():
            o = arena.offsets[name]
            wd[o:o + p.numel()] = 0.0 if is_picked(name) else float(weight_decay)
        self.wd = wd.to(dev)
        self.steps = 0
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.pen = torch.zeros(1, dtype=torch.float64, device=dev)
        self.y = self.dy = None        # allocated on the first step, then re-used (stable pointers -> graph replay)

    def _stream(self, t):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def step(self, x, target, world_size=1):
        """Returns (mean BCE loss, penalty / batchsize) as fp64 device scalars (no host sync)."""
        model = self.model
        assert model.training
        if model._arena is None or not model._arena.is_current():
            raise RuntimeError("the parameter arena was re-allocated (model.to()/cuda() after creating the trainer)")
        B = x.shape[0]
        if self.y is None or self.y.shape[0] != B or tuple(self.y.shape[2:]) != tuple(x.shape[2:]):
            self.y = torch.empty((B, 1) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
            self.dy = torch.empty_like(self.y)
        y, pen = model._train_forward_raw(x, y=self.y, penalty=self.pen)
        self.loss.zero_()
        dy = self.dy
        N.check(self.lib, self.lib.csn_bce_with_logits(y.data_ptr(), target.data_ptr(), dy.data_ptr(), y.numel(),
                                                       self.loss.data_ptr(), self._stream(y)), "csn_bce_with_logits")
        bs = self.batchsize or B
        model._train_backward_raw(x, dy, self.flops_weight / bs, grad=self.grad)
        if world_size > 1:          # the step's only collective: average the flat gradient over the data-parallel ranks
            from sod100k_amd.dist import allreduce_mean_
            allreduce_mean_(self.grad, world_size)
        self.steps += 1
        flat = model._arena.flat
        N.check(self.lib, self.lib.csn_adam_step(flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(),
                                                 self.v.data_ptr(), self.wd.data_ptr(), self.n, self.lr,
                                                 self.betas[0], self.betas[1], self.eps, self.steps,
                                                 self._stream(y)), "csn_adam_step")
        return self.loss.clone(), pen.clone() / bs


def reference_style_step(model, optimizer, x, target, flops_weight):
    """train.py:203-216 verbatim over the autograd seam (torch BCE + any torch optimizer)."""
    output = model(x)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(output, target)
    if flops_weight:
        loss = loss + flops_weight * model.get_flops()
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    model.clear_flops()
    return loss.detach()


def multistep_lr(base_lr, steps, epoch, gamma=0.1):
    """lr_scheduler.MultiStepLR as train.py:146-157 uses it (stepped at the START of every epoch)."""
    return base_lr * gamma ** sum(1 for s in steps if epoch + 1 >= s)


def val(model, batches, lib=None):
    """Validation loop of train.py:250-293: eval-mode forward, then per picture sigmoid -> bilinear resize to the
    picture's own (h, w) -> (x * 255).int() / 255 -> L1 mean against its target; returns the average over pictures
    (``maes.avg``).  ``batches`` yields (images B x 3 x H x W, [target_i of shape h_i x w_i]); resize + quantise +
    L1 run in one kernel per picture (csn_val_mae), one host read at the end."""
    from sod100k_amd.engine import val_mae
    was_training = model.training
    model.eval()
    lib = lib if lib is not None else (model._lib or N.load())
    total, count = None, 0
    with torch.no_grad():
        for img, targets in batches:
            out = model(img.float())
            if total is None:
                total = torch.zeros(1, dtype=torch.float64, device=out.device)
            for idx, t in enumerate(targets):
                val_mae(lib, out[idx], t.to(out.device).float(), out=total)
                count += 1
    model.train(was_training)
    return float(total) / max(count, 1) if total is not None else 0.0


def synthetic_batches(n, batch, h, w, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        yield (torch.randn(batch, 3, h, w, generator=g).to(device),
               (torch.rand(batch, 1, h, w, generator=g) > 0.5).float().to(device))


def run(cfg, device="cuda", synthetic=0, max_steps=0):
    model_lib = importlib.import_module("model." + cfg.MODEL.ARCH)                     # train.py:70
    if not cfg.AUTO.ENABLE:
        print("Enable AUTO to train CSNet!")
        return None
    layer_config_dir = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK, 'layer_configs')
    os.makedirs(layer_config_dir, exist_ok=True)
    model = model_lib.build_model(basic_split=cfg.MODEL.BASIC_SPLIT, predefine=cfg.AUTO.PREDEFINE,
                                  save_path=layer_config_dir, expand=cfg.AUTO.EXPAND)
    if cfg.AUTO.FLOPS.ENABLE:
        if cfg.AUTO.FLOPS.EXPAND != -1.0:
            model.flops_hook(expandflop=cfg.AUTO.FLOPS.EXPAND)
        else:
            model.flops_hook()
        model.set_batchsize(cfg.DATA.BATCH_SIZE)
    model = model.to(device).train()
    if cfg.SOLVER.METHOD != 'Adam_dynamic_weight_decay':
        print("WARNING: Method not implmented.")
        return None
    trainer = FusedTrainer(model, lr=cfg.SOLVER.LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY,
                           flops_weight=cfg.AUTO.FLOPS.WEIGHT if cfg.AUTO.FLOPS.ENABLE else 0.0,
                           batchsize=cfg.DATA.BATCH_SIZE)
    done = 0
    for epoch in range(cfg.SOLVER.MAX_EPOCHS):
        if cfg.SOLVER.ADJUST_STEP:
            trainer.lr = multistep_lr(cfg.SOLVER.LR, cfg.SOLVER.STEPS, epoch)
        if synthetic <= 0:
            print("dataset loading (prepare_data.py) is host IO outside this build; use --synthetic N")
            return trainer
        for i, (x, t) in enumerate(synthetic_batches(synthetic, cfg.DATA.BATCH_SIZE, cfg.DATA.IMAGE_H, cfg.DATA.IMAGE_W,
                                                     device, seed=epoch)):
            loss, pen = trainer.step(x, t)
            model.clear_flops()
            if i % cfg.PRINT_FREQ == 0:
                print(f"Epoch: [{epoch}][{i}/{synthetic}] Loss {float(loss):.4f} flops-penalty {float(pen):.6f} "
                      f"lr {trainer.lr:g}")
            done += 1
            if max_steps and done >= max_steps:
                return trainer
    return trainer


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default=os.path.join(ROOT, "sod100k_amd", "configs", "csnet-L-x2_train.yml"))
    ap.add_argument("--synthetic", type=int, default=0, help="batches of synthetic data per epoch")
    ap.add_argument("--max-steps", type=int, default=0)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    cfg = defaults()
    cfg.merge_from_file(args.config)
    if args.opts:
        cfg.merge_from_list(args.opts)
    run(cfg, device=args.device, synthetic=args.synthetic, max_steps=args.max_steps)


if __name__ == "__main__":
    main()
