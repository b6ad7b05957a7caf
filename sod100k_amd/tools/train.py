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
                 batchsize=None, lib=None, act_dtype=None):
        self.model = model
        if act_dtype is not None:          # "bf16": BASELINE config 3 (bfloat16 activation storage, fp32 everything else)
            model.set_train_act_dtype(act_dtype)
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
        dev = model._arena.flat.device
        # the kernels read raw fp32 memory: the reference calls .float() on both tensors (train.py:199-202)
        if x.dtype != torch.float32 or x.device != dev or not x.is_contiguous():
            x = x.to(dev, torch.float32).contiguous()
        if target.dtype != torch.float32 or target.device != dev or not target.is_contiguous():
            target = target.to(dev, torch.float32).contiguous()
        if target.numel() != B * x.shape[2] * x.shape[3]:
            raise ValueError(f"target of {tuple(target.shape)} does not match logits of {(B, 1) + tuple(x.shape[2:])}")
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
        from sod100k_amd.dist import allreduce_mean_, active
        if world_size > 1 or active():   # the step's only collective: average the flat gradient over the data-parallel ranks
            allreduce_mean_(self.grad, world_size)
        self.steps += 1
        flat = model._arena.flat
        N.check(self.lib, self.lib.csn_adam_step(flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(),
                                                 self.v.data_ptr(), self.wd.data_ptr(), self.n, self.lr,
                                                 self.betas[0], self.betas[1], self.eps, self.steps,
                                                 self._stream(y)), "csn_adam_step")
        return self.loss.clone(), pen.clone() / bs

    # ---- checkpointing in torch.optim.Adam's own format (train.py:172-181 saves optimizer.state_dict()) ----
    def _param_order(self):
        named = list(self.model.named_parameters())
        normal = [(n, p) for n, p in named if not is_picked(n)]
        picked = [(n, p) for n, p in named if is_picked(n)]
        return normal, picked                     # group 0, group 1 of train.py:97-123

    def state_dict(self):
        """``torch.optim.Adam(...).state_dict()`` of the reference's two-group optimizer, filled from the flat buffers, so a
        checkpoint written here resumes under the reference's train.py:130-141 and vice versa."""
        normal, picked = self._param_order()
        offs = self.model._arena.offsets
        state = {}
        idx = 0
        for grp in (normal, picked):
            for name, p in grp:
                if self.steps > 0:
                    o = offs[name]
                    state[idx] = {"step": torch.tensor(float(self.steps)),
                                  "exp_avg": self.m[o:o + p.numel()].view(p.shape).detach().cpu().clone(),
                                  "exp_avg_sq": self.v[o:o + p.numel()].view(p.shape).detach().cpu().clone()}
                idx += 1
        wd0 = float(self.wd[offs[normal[0][0]]]) if normal else 0.0

        def group(lo, n, wd):
            return {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": wd, "amsgrad": False,
                    "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                    "params": list(range(lo, lo + n))}
        return {"state": state, "param_groups": [group(0, len(normal), wd0), group(len(normal), len(picked), 0.0)]}

    def load_state_dict(self, sd):
        normal, picked = self._param_order()
        order = normal + picked
        offs = self.model._arena.offsets
        self.m.zero_(); self.v.zero_()
        steps = 0
        for idx, (name, p) in enumerate(order):
            st = sd.get("state", {}).get(idx)
            if st is None:
                continue
            o = offs[name]
            self.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps = max(steps, int(float(st["step"])))
        self.steps = steps                         # one step counter for all parameters, as one optimizer.step() gives
        groups = sd.get("param_groups") or []
        if groups:
            self.lr = float(groups[0].get("lr", self.lr))


def load_pretrained(model, pretrained_path):
    """utils/utils.py:6-24: copy every tensor of a raw state_dict file whose key exists in the model (train.py:124-125)."""
    if os.path.isfile(pretrained_path):
        print("=> loading checkpoint '{}'".format(pretrained_path))
        pretrain = torch.load(pretrained_path, map_location="cpu")
        state = model.state_dict()
        state.update({k: v for k, v in pretrain.items() if k in state})
        model.load_state_dict(state)
        return model
    print("=> no checkpoint found at '{}'".format(pretrained_path))
    return model


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


def run(cfg, device="cuda", synthetic=0, max_steps=0, val_batches=None, lib=None):
    """train.py:67-181.  One process per GPU when launched by ``torch.distributed.run`` (RANK / WORLD_SIZE in the
    environment): every rank trains on its own image shard with per-GPU BN statistics and the step's single collective
    (gradient all-reduce) inside ``FusedTrainer.step``; rank 0 validates and writes the checkpoints."""
    from sod100k_amd import dist as D
    rank = int(os.environ.get("RANK", "0"))
    if str(device).startswith("cuda") and "LOCAL_RANK" in os.environ:
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        device = f"cuda:{int(os.environ['LOCAL_RANK'])}"
    world = D.init(device=torch.device(device))
    model_lib = importlib.import_module("model." + cfg.MODEL.ARCH)                     # train.py:70
    if not cfg.AUTO.ENABLE:
        print("Enable AUTO to train CSNet!")
        return None
    task_dir = os.path.join(cfg.DATA.SAVEDIR, cfg.TASK)
    layer_config_dir = os.path.join(task_dir, 'layer_configs')
    check_point_dir = os.path.join(task_dir, 'checkpoint')
    os.makedirs(layer_config_dir, exist_ok=True)
    os.makedirs(check_point_dir, exist_ok=True)
    # epoch-0 build: writes layer_config_0.bin / layer_config_latest.bin (init path) and checkpoint/checkpoint_init.pth.tar,
    # which finetune.py:96-99 reads back (CSNet_training/model/csnet.py:903-945)
    # Only rank 0 writes files; the other ranks wait and build from the layer_config it left (or from AUTO.PREDEFINE).
    if rank == 0:
        model = model_lib.build_model(basic_split=cfg.MODEL.BASIC_SPLIT, predefine=cfg.AUTO.PREDEFINE,
                                      save_path=layer_config_dir, expand=cfg.AUTO.EXPAND)
    if D.active():
        torch.distributed.barrier()
    if rank != 0:
        pre = cfg.AUTO.PREDEFINE if os.path.isfile(cfg.AUTO.PREDEFINE) else os.path.join(layer_config_dir, 'layer_config_0.bin')
        model = model_lib.build_model(basic_split=cfg.MODEL.BASIC_SPLIT, predefine=pre, save_path='tmp', expand=cfg.AUTO.EXPAND)
    if cfg.AUTO.FLOPS.ENABLE:
        if cfg.AUTO.FLOPS.EXPAND != -1.0:
            model.flops_hook(expandflop=cfg.AUTO.FLOPS.EXPAND)
        else:
            model.flops_hook()
        model.set_batchsize(cfg.DATA.BATCH_SIZE)
    if lib is not None:
        model._lib = lib
    model = model.to(device).train()
    if cfg.SOLVER.METHOD != 'Adam_dynamic_weight_decay':
        print("WARNING: Method not implmented.")
        return None
    trainer = FusedTrainer(model, lr=cfg.SOLVER.LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY,
                           flops_weight=0.0, batchsize=cfg.DATA.BATCH_SIZE, lib=lib)
    if cfg.DATA.PRETRAIN != '':
        load_pretrained(model, cfg.DATA.PRETRAIN)                                      # train.py:124-125
    start_epoch = 0
    if cfg.DATA.RESUME != '':                                                          # train.py:127-141
        if os.path.isfile(cfg.DATA.RESUME):
            print("=> loading checkpoint '{}'".format(cfg.DATA.RESUME))
            checkpoint = torch.load(cfg.DATA.RESUME, map_location="cpu", weights_only=False)
            start_epoch = checkpoint['epoch']
            model.load_state_dict(checkpoint['state_dict'])
            if checkpoint.get('optimizer'):
                trainer.load_state_dict(checkpoint['optimizer'])
            print("=> loaded checkpoint '{}' (epoch {})".format(cfg.DATA.RESUME, checkpoint['epoch']))
        else:
            print("=> no checkpoint found at '{}'".format(cfg.DATA.RESUME))
    # every rank initialised its own random weights (and PRETRAIN may match only some keys): one model for all replicas
    D.broadcast_model_(model, src=0)
    if cfg.SOLVER.ADJUST_STEP and cfg.SOLVER.LR_SCHEDULER != 'step':
        raise ValueError("Unsupported scheduler.")
    best_mae, best_epoch, done = 1000000, -1, 0
    sched_steps = 0                  # the reference's scheduler restarts its own count on resume (train.py:143-157)
    for epoch in range(start_epoch, cfg.SOLVER.MAX_EPOCHS):
        if (cfg.SOLVER.FINETUNE.ADJUST_STEP and epoch > cfg.AUTO.FINETUNE) or cfg.SOLVER.ADJUST_STEP:   # train.py:152-156
            sched_steps += 1
            trainer.lr = cfg.SOLVER.LR * 0.1 ** sum(1 for s_ in cfg.SOLVER.STEPS if sched_steps >= s_)
        # the FLOPs penalty is applied only while epoch < AUTO.FINETUNE (train.py:212-213)
        penal = cfg.AUTO.FLOPS.ENABLE and epoch < cfg.AUTO.FINETUNE
        trainer.flops_weight = float(cfg.AUTO.FLOPS.WEIGHT) if penal else 0.0
        if synthetic <= 0:
            print("dataset loading (prepare_data.py) is host IO outside this build; use --synthetic N")
            return trainer
        model.train()
        for i, (x, t) in enumerate(synthetic_batches(synthetic, cfg.DATA.BATCH_SIZE, cfg.DATA.IMAGE_H, cfg.DATA.IMAGE_W,
                                                     device, seed=epoch * world + rank)):
            loss, pen = trainer.step(x, t, world_size=world)
            model.clear_flops()
            if i % cfg.PRINT_FREQ == 0 and rank == 0:
                print(f"Epoch: [{epoch + 1}][{i}/{synthetic}] Loss {float(loss):.4f} FakeFLOPs {float(pen):.3f} "
                      f"lr {trainer.lr:g}")
            done += 1
            if max_steps and done >= max_steps:
                break
        if rank == 0:
            mae = val(model, val_batches, lib=lib) if val_batches is not None else float("nan")       # train.py:160
            if mae < best_mae:
                best_mae, best_epoch = mae, epoch + 1
            print(" epoch: " + str(epoch + 1) + " mae: " + str(mae) + " best_epoch: " + str(best_epoch) + " best_mae: "
                  + str(best_mae))
            from sod100k_amd.checkpoint import save_checkpoint
            save_checkpoint({'epoch': epoch + 1, 'arch': cfg.MODEL.ARCH, 'state_dict': model.state_dict(),
                             'optimizer': trainer.state_dict()},
                            os.path.join(check_point_dir, 'checkpoint_epoch{}.pth.tar'.format(epoch + 1)))   # train.py:172-181
        if max_steps and done >= max_steps:
            break
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
