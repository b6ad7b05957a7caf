"""Drop-in ``model.csnet`` for the MI355X-native CSNet engine.

Mirrors the public surface of the reference module (CSNet/model/csnet.py; training copy
CSNet_training/model/csnet.py): ``build_model``, ``CSNet``, ``ILBlock``, ``gOctaveCBR``, ``gOctaveConv``,
``SimplifiedGOctConvBR``, ``CSFHead``, ``PallMSBlock``, ``MSBlock``, ``init_layers``,
``load_layer_config``, ``save_layer_config``, ``Oct_bn_hook``.  The module tree, parameter names,
shapes, dtypes and default initialisers are the reference's (checked key-for-key against the 737-key
manifest in tests/golden/g6_simplesum_keys.json), so ``load_state_dict(strict=True)``, optimizers that
pattern-match parameter names (train.py:101-107) and ``isinstance`` walks (train.py:320-330) keep working.

What differs is WHO computes: the sub-modules are parameter containers and ``CSNet.forward`` hands the
whole network to one fused plan of hand-written HIP kernels (libcsnet_hip.so, include/csnet_hip.h).
There is no PyTorch / CPU fallback: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import json
import math
import os
import pickle
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from .conv2d import Conv2dX100
from sod100k_amd import _native as N
from sod100k_amd.engine import Engine, ParamArena

DILATIONS = (1, 2, 4, 8, 16)      # csnet.py:121


def _split_fractions(split):
    """``alpha = split / int(round(sum(split)))`` as python floats (csnet.py:26-31, 157-165)."""
    split = np.asarray(split)
    total = int(round(float(np.sum(split))))
    return (split * 1.0 / total).tolist(), total


def _cumulative(alphas: Sequence[float]) -> List[float]:
    cum, run = [0], 0
    for a in alphas:
        run += a
        cum.append(run)
    return cum


def _container_forward(self, *args, **kwargs):
    raise RuntimeError(f"{type(self).__name__} is a parameter container in sod100k_amd; the fused HIP plan behind "
                       "CSNet.forward executes it (there is no per-layer PyTorch path).")


class gOctaveConv(nn.Module):
    """Weight holder of the generalized OctConv (csnet.py:604-726): one ``[Cout, Cin/groups, k, k]`` tensor
    block-partitioned ``[out-branch j][in-branch i]`` at ``int(round(C * cumsum(alpha)))`` (csnet.py:683-691)."""

    def __init__(self, in_channels, out_channels, kernel_size, alpha_in=[0.5, 0.5], alpha_out=[0.5, 0.5],
                 stride=1, padding=1, dilation=1, groups=1, bias=False, up_kwargs=None):
        super().__init__()
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, round(in_channels / groups),
                                               kernel_size[0], kernel_size[1]))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.alpha_in = _cumulative(alpha_in)
        self.alpha_out = _cumulative(alpha_out)
        self.inbranch, self.outbranch = len(alpha_in), len(alpha_out)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def in_bounds(self):
        return [int(round(self.in_channels * a / self.groups)) for a in self.alpha_in]

    def out_bounds(self):
        return [int(round(self.out_channels * a)) for a in self.alpha_out]

    forward = _container_forward


class gOctaveCBR(nn.Module):
    """gOctConv + BatchNorm2d + PReLU per output branch (csnet.py:729-792)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), alpha_in=[0.5, 0.5], alpha_out=[0.5, 0.5],
                 stride=1, padding=1, dilation=1, groups=1, bias=False, up_kwargs=None, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = tuple(kernel_size), stride
        self.std_conv = len(alpha_in) == 1 and len(alpha_out) == 1            # csnet.py:751-754
        if self.std_conv:
            self.conv = Conv2dX100(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        else:
            self.conv = gOctaveConv(in_channels, out_channels, kernel_size, alpha_in, alpha_out, stride, padding,
                                    dilation, groups, bias, up_kwargs)
        self.bns = nn.ModuleList()
        self.prelus = nn.ModuleList()
        for a in alpha_out:
            c = int(round(out_channels * a))
            self.bns.append(norm_layer(c) if c != 0 else None)
            self.prelus.append(nn.PReLU(c) if c != 0 else None)
        self.outbranch = len(alpha_out)
        self.alpha_in, self.alpha_out = alpha_in, alpha_out
        self.all_flops = 0
        self.baseflop = None
        self.expandflop = None

    forward = _container_forward


class SimplifiedGOctConvBR(nn.Module):
    """Per-branch depthwise 3x3 (Conv2dX100) + BatchNorm2d + PReLU (csnet.py:795-851)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), alpha=[0.5, 0.5], stride=1, padding=1,
                 dilation=1, groups=1, bias=False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.std_conv = False
        self.convs, self.bns, self.prelus = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for a in alpha:
            cin, cout = int(round(in_channels * a)), int(round(out_channels * a))
            if cin >= 1:
                self.convs.append(Conv2dX100(cin, cout, kernel_size=(3, 3), groups=cout, padding=padding,
                                             dilation=dilation, bias=bias))
                self.bns.append(norm_layer(cout))
                self.prelus.append(nn.PReLU(cout))
            else:
                self.convs.append(None)
                self.bns.append(None)
                self.prelus.append(None)
        self.outbranch = len(alpha)
        self.all_flops = 0
        self.baseflop = None
        self.expandflop = None

    forward = _container_forward


class ILBlock(nn.Module):
    """gOctaveCBR (3x3 when first / stride 2, else 1x1) followed by two depthwise units (csnet.py:17-76)."""

    def __init__(self, inlist, outlist, stride=1, nextstride=1, nextoutlist=None, first=False):
        super().__init__()
        alpha_in, ninput = _split_fractions(inlist)
        alpha_out, noutput = _split_fractions(outlist)
        self.first = first
        k = 3 if (first or stride == 2) else 1                                  # csnet.py:33-48
        self.conv1x1 = gOctaveCBR(ninput, noutput, kernel_size=(k, k), padding=k // 2, alpha_in=alpha_in,
                                  alpha_out=alpha_out, stride=stride if k == 3 else 1)
        self.conv3x3_1 = SimplifiedGOctConvBR(noutput, noutput, stride=1, kernel_size=(3, 3), padding=1,
                                              alpha=alpha_out, groups=noutput)
        self.conv3x3_2 = SimplifiedGOctConvBR(noutput, noutput, stride=1, kernel_size=(3, 3), padding=1,
                                              alpha=alpha_out, groups=noutput)
        self.all_flops = 0
        self.stride, self.nextstride, self.nextoutlist = stride, nextstride, nextoutlist
        self.baseflop = None
        self.expandflop = None

    forward = _container_forward


class MSBlock(nn.Module):
    """Five dilated 3x3 Conv2dX100 (absent when 0 channels) -> cat -> BN -> PReLU (csnet.py:116-149)."""

    def __init__(self, in_channels, out_channels, dil_channels, dilations=[1, 2, 4, 8, 16]):
        super().__init__()
        self.dilations = dilations
        self.in_channels, self.out_channels = in_channels, out_channels
        self.dil_channels = [int(c) for c in dil_channels]
        self.msconv = nn.ModuleList()
        self.real_dil_branch = len(dilations)
        for d, c in zip(dilations, self.dil_channels):
            self.msconv.append(Conv2dX100(in_channels, c, 3, padding=d, dilation=d, bias=False) if c != 0 else None)
        self.bn = nn.BatchNorm2d(out_channels)
        self.prelu = nn.PReLU(out_channels)

    forward = _container_forward


class PallMSBlock(nn.Module):
    """One MSBlock per resolution branch; ``None`` where every dilation has 0 channels (csnet.py:79-113)."""

    def __init__(self, in_channels, out_channels, dil_channels, alpha_in=[0.5, 0.5], alpha_out=[0.5, 0.5],
                 bias=False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.std_conv = False
        self.convs = nn.ModuleList()
        for i in range(len(alpha_in)):
            if max(dil_channels[i]) != 0:
                self.convs.append(MSBlock(int(round(in_channels * alpha_in[i])),
                                          int(round(out_channels * alpha_out[i])), dil_channels[i]))
            else:
                self.convs.append(None)
        self.outbranch = len(alpha_in)

    forward = _container_forward


class CSFHead(nn.Module):
    """Cross-stage fusion decoder: gOctaveCBR 3->3 (1x1) -> PallMSBlock -> gOctaveCBR 3->1 (csnet.py:152-206)."""

    def __init__(self, fuse_layer_config):
        super().__init__()
        self.layer_config = fuse_layer_config
        in_split, in_ch = _split_fractions(fuse_layer_config[0][0])
        mid_in_split, mid_in_ch = _split_fractions(fuse_layer_config[1][0])
        mid_out_split, mid_out_ch = _split_fractions(fuse_layer_config[1][1])
        dils = fuse_layer_config[1][2]
        out_ch = int(round(float(np.sum(fuse_layer_config[2][1]))))
        self.fuse = gOctaveCBR(in_ch, mid_in_ch, kernel_size=(1, 1), padding=0, alpha_in=in_split,
                               alpha_out=mid_in_split, stride=1)
        self.ms = PallMSBlock(mid_in_ch, mid_out_ch, alpha_in=mid_in_split, alpha_out=mid_out_split,
                              dil_channels=dils)
        self.fuse1x1 = gOctaveCBR(mid_out_ch, out_ch, kernel_size=(1, 1), padding=0, alpha_in=mid_out_split,
                                  alpha_out=[1], stride=1)

    forward = _container_forward


class CSNet(nn.Module):
    """CSNet (csnet.py:209-387): stage0 (1 ILBlock) + 4 stages + CSF head + 1x1 classifier + bilinear up.

    ``forward(x)``: ``x`` float32 ``B x 3 x H x W`` on a ROCm device (H, W multiples of 16) ->
    float32 ``B x 1 x H x W`` logits, exactly the reference's tensor contract (csnet.py:365-387).
    """

    def __init__(self, layer_config, num_classes=1):
        super().__init__()
        if num_classes != 1:
            raise ValueError("sod100k_amd implements the shipped single-class saliency head (num_classes=1)")
        self.stages = list(layer_config[-1])
        self.layer_config = layer_config
        idx = 0
        self.stage0 = nn.ModuleList([ILBlock(np.array([3]), layer_config[0][1],
                                             nextoutlist=layer_config[1][1], stride=1, first=True)])
        idx = 1
        stage_lists = []
        for s, n in enumerate(self.stages):
            blocks = nn.ModuleList()
            for b in range(n):
                last = b == n - 1
                blocks.append(ILBlock(layer_config[idx][0], layer_config[idx][1],
                                      nextoutlist=None if (s == 3 and b > 0) else layer_config[idx + 1][1],
                                      stride=2 if (s >= 1 and b == 0) else 1,
                                      nextstride=1 if (b == 0 or s == 3) else (2 if last else 1)))
                idx += 1
            stage_lists.append(blocks)
        self.stage1, self.stage2, self.stage3, self.stage4 = stage_lists
        self.oct_fuse = CSFHead(layer_config[idx:idx + 3])
        fuse_out = int(round(float(np.sum(layer_config[-2][1]))))
        self.cls_layer = nn.Conv2d(fuse_out, num_classes, kernel_size=1)
        self.all_flops = 0
        self.batchsize = 0
        # engine state (not part of state_dict)
        self._arena: Optional[ParamArena] = None
        self._engines = {}
        self._lib = None            # tests may inject another build of the same C ABI; None -> libcsnet_hip.so
        self._train_generation = 0  # counts train-mode forwards (autograd seam: backward must belong to the last one)
        self._sub_batch = int(os.environ.get("CSN_SUB_BATCH", "0"))
        self._penalty_cfg = None    # set by flops_hook()

    # ---- dynamic-weight-decay API (csnet.py:313-363) --------------------------------------------------
    def set_batchsize(self, batchsize):
        self.batchsize = batchsize

    def clear_flops(self):
        self.all_flops = 0
        for m in self.modules():
            if isinstance(m, ILBlock):
                m.conv1x1.all_flops = 0
                m.conv3x3_1.all_flops = 0
                m.conv3x3_2.all_flops = 0

    def get_flops(self):
        for m in self.modules():
            if isinstance(m, ILBlock):
                self.all_flops = m.conv1x1.all_flops + m.conv3x3_1.all_flops + m.conv3x3_2.all_flops + self.all_flops
        return self.all_flops / self.batchsize

    def flops_hook(self, expandflop=2):
        """Assign the per-stage penalty weights of the reference hook (csnet.py:332-355).

        The penalty itself (Oct_bn_hook, csnet.py:391-410) is fused into the train-mode BN/PReLU pass
        (k_train.hip: bn_apply_gap_kernel); this only records the weights."""
        baseflop = expandflop ** (len(self.stages) - 1)
        real_stages = list(self.stages)
        real_stages[0] += 1
        stage = in_stage = 0
        for m in self.modules():
            if isinstance(m, ILBlock):
                for sub in (m.conv1x1, m.conv3x3_1, m.conv3x3_2):
                    sub.baseflop, sub.expandflop = baseflop, expandflop
                in_stage += 1
                if in_stage == real_stages[stage]:
                    baseflop /= expandflop
                    stage += 1
                    in_stage = 0
        self._penalty_cfg = dict(expandflop=expandflop)

    def updateWeight(self, s=0.001):
        for m in self.modules():
            if isinstance(m, gOctaveCBR):
                for n in m.modules():
                    if isinstance(n, nn.BatchNorm2d):
                        n.weight.grad.data.add_(s * torch.sign(n.weight.data))

    # ---- plan description -------------------------------------------------------------------------------
    def _blocks(self) -> List[ILBlock]:
        out = [self.stage0[0]]
        for st in (self.stage1, self.stage2, self.stage3, self.stage4):
            out.extend(st)
        return out

    def describe(self, offsets):
        """Translate the module tree into the unit / activation descriptors of include/csnet_hip.h.

        ``offsets``: parameter / buffer name -> float offset in the arena.  Returns (units, acts, names).
        """
        acts = [(3, 0)]                       # id 0 = network input
        units, names = [], []

        def new_act(c, lvl):
            acts.append((int(c), int(lvl)))
            return len(acts) - 1

        def bn_fill(u, j, prefix_bn, prefix_prelu):
            u.bn[j].weight = offsets[prefix_bn + ".weight"]
            u.bn[j].bias = offsets[prefix_bn + ".bias"]
            u.bn[j].running_mean = offsets[prefix_bn + ".running_mean"]
            u.bn[j].running_var = offsets[prefix_bn + ".running_var"]
            u.bn[j].prelu = offsets[prefix_prelu + ".weight"]

        def add_goct(name, cbr: gOctaveCBR, cur):
            """cur: list of (act_id | None, channels, lvl) per input branch."""
            conv = cbr.conv
            u = N.new_unit(N.UNIT_GOCT)
            if cbr.std_conv:      # Conv2dX100 (csnet.py:751-754): the plan infers x100 / real stride from 1 -> 1 branches
                bi, bo = [0, cbr.in_channels], [0, cbr.out_channels]
                u.n_in = u.n_out = 1
                stride = cbr.stride
                if isinstance(cur, tuple):
                    cur = [cur]
            else:
                bi, bo = conv.in_bounds(), conv.out_bounds()
                u.n_in, u.n_out = conv.inbranch, conv.outbranch
                stride = conv.stride
            nin, nout = int(u.n_in), int(u.n_out)
            u.ksize, u.stride = cbr.kernel_size[0], stride
            u.w_off[0] = offsets[name + ".conv.weight"]
            lvl0 = None
            for i in range(nin):
                c = bi[i + 1] - bi[i]
                present = cur[i] is not None and cur[i][0] is not None and c > 0
                u.cin[i] = c if present else 0
                u.in_act[i] = cur[i][0] if present else -1
                if present:
                    assert cur[i][1] == c, (name, i, cur[i], c)
                    if lvl0 is None:
                        lvl0 = cur[i][2] - i
            base = lvl0 + (1 if stride == 2 else 0)
            outs = []
            for j in range(nout):
                c = bo[j + 1] - bo[j]
                if c > 0 and cbr.bns[j] is not None:
                    assert cbr.bns[j].num_features == c
                    a = new_act(c, base + j)
                    u.cout[j], u.out_act[j] = c, a
                    bn_fill(u, j, f"{name}.bns.{j}", f"{name}.prelus.{j}")
                    outs.append((a, c, base + j))
                else:
                    outs.append(None)
            units.append(u)
            names.append(name)
            return outs

        def add_dw(name, dw: SimplifiedGOctConvBR, cur):
            u = N.new_unit(N.UNIT_DW)
            u.n_in = u.n_out = dw.outbranch
            outs = []
            for k in range(dw.outbranch):
                if cur[k] is None or dw.convs[k] is None:
                    outs.append(None)
                    continue
                a_in, c, lvl = cur[k]
                assert dw.convs[k].out_channels == c
                a = new_act(c, lvl)
                u.cin[k] = u.cout[k] = c
                u.in_act[k], u.out_act[k] = a_in, a
                u.w_off[k] = offsets[f"{name}.convs.{k}.weight"]
                bn_fill(u, k, f"{name}.bns.{k}", f"{name}.prelus.{k}")
                outs.append((a, c, lvl))
            units.append(u)
            names.append(name)
            return outs

        cur = [(0, 3, 0)]
        heads = []
        blocks = self._blocks()
        block_names = ["stage0.0"] + [f"stage{s + 1}.{b}" for s, n in enumerate(self.stages) for b in range(n)]
        ends = np.cumsum([1] + self.stages)          # block index after which a stage ends
        for bi_, (blk, bname) in enumerate(zip(blocks, block_names)):
            cur = add_goct(bname + ".conv1x1", blk.conv1x1, cur)
            cur = add_dw(bname + ".conv3x3_1", blk.conv3x3_1, cur)
            cur = add_dw(bname + ".conv3x3_2", blk.conv3x3_2, cur)
            if (bi_ + 1) in ends[2:]:                 # end of stage2 / stage3 / stage4  (csnet.py:380)
                heads.append(cur[0])
        fuse = add_goct("oct_fuse.fuse", self.oct_fuse.fuse, heads)
        mids = []
        for i, ms in enumerate(self.oct_fuse.ms.convs):
            if ms is None or fuse[i] is None:
                mids.append(None)
                continue
            a_in, c, lvl = fuse[i]
            assert ms.in_channels == c
            u = N.new_unit(N.UNIT_MS)
            u.n_in = u.n_out = 1
            u.cin[0], u.cout[0] = c, ms.out_channels
            a = new_act(ms.out_channels, lvl)
            u.in_act[0], u.out_act[0] = a_in, a
            for d in range(N.NDIL):
                u.dil_ch[d] = ms.dil_channels[d]
                if ms.dil_channels[d]:
                    u.w_off[d] = offsets[f"oct_fuse.ms.convs.{i}.msconv.{d}.weight"]
            bn_fill(u, 0, f"oct_fuse.ms.convs.{i}.bn", f"oct_fuse.ms.convs.{i}.prelu")
            units.append(u)
            names.append(f"oct_fuse.ms.convs.{i}")
            mids.append((a, ms.out_channels, lvl))
        last = add_goct("oct_fuse.fuse1x1", self.oct_fuse.fuse1x1, mids)
        a_in, c, lvl = last[0]
        u = N.new_unit(N.UNIT_CLS)
        u.n_in = u.n_out = 1
        u.cin[0], u.cout[0] = c, 1
        u.in_act[0] = a_in
        u.w_off[0] = offsets["cls_layer.weight"]
        u.bias_off = offsets["cls_layer.bias"]
        units.append(u)
        names.append("cls_layer")
        return units, acts, names

    # ---- execution -----------------------------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._arena = None        # .cuda()/.to() re-allocate parameters: rebuild the arena lazily
        self._engines = {}
        return out

    def _ensure_arena(self) -> ParamArena:
        if self._arena is None or not self._arena.is_current():
            self._arena = ParamArena(self)
            self._engines = {}
        return self._arena

    def engine_for(self, x: torch.Tensor, train: bool = False, slice_lanes: Optional[bool] = None) -> Engine:
        arena = self._ensure_arena()
        if x.device != arena.flat.device:
            raise RuntimeError(f"input on {x.device} but model parameters on {arena.flat.device}")
        lib = self._lib
        if lib is None:
            if not x.is_cuda:
                raise RuntimeError("sod100k_amd.CSNet runs on ROCm devices only (hand-written HIP kernels); "
                                   "move the model and the input to the GPU (`model.cuda()`, `x.cuda()`).")
            lib = N.load()
        bf16 = bool(train) and getattr(self, "_train_act_dtype", "fp32") == "bf16"
        # Eval batches of 32 images and more run as two half-batches side by side on the plan's stream lanes (CSN_OPT_SLICE_LANES):
        # one half's launch tails and small-map launches overlap the other half's work.  Bit-identical results; round 6, same
        # lease: 30.4K -> 31.3K img/s at batch 64 (three slices 31.0K, four 25.3K).  CSN_SLICE_LANES=0 / slice_lanes=False: whole
        # batch on one stream (what the per-kernel profiles of bench.py are taken on); CSN_SLICE_LANES=1: from batch 2 on.
        env = os.environ.get("CSN_SLICE_LANES")
        if slice_lanes is None:
            slice_lanes = env == "1" or (env != "0" and x.shape[0] >= 32 and not self._sub_batch)
        lanes = bool(slice_lanes) and not train and x.is_cuda and x.shape[0] >= 2
        # the gradient w.r.t. the image batch (x.requires_grad under autograd): a plan with one more gradient buffer (CSN_OPT_INPUT_GRAD)
        input_grad = bool(train) and bool(getattr(self, "_want_input_grad", False))
        key = (tuple(x.shape), x.device, bool(train), bf16, lanes, input_grad)
        eng = self._engines.get(key)
        if eng is not None:
            self._engines[key] = self._engines.pop(key)      # most recently used last
        if eng is None:
            # a plan holds its workspace and graphs: native-resolution inference (test.py:80-85) meets a new (B, H, W) per
            # picture size, so the least recently used plans are dropped beyond CSN_MAX_ENGINES
            cap = max(1, int(os.environ.get("CSN_MAX_ENGINES", "8")))
            while len(self._engines) >= cap:
                self._engines.pop(next(iter(self._engines)))
            B, _, H, W = x.shape
            if H % 16 or W % 16:
                raise ValueError("CSNet needs H and W to be multiples of 16 (cf. test.py:80-85)")
            units, acts, names = self.describe(arena.offsets)
            sub_batch = 0 if train else self._sub_batch
            if lanes and sub_batch == 0:
                sub_batch = (B + 1) // 2
            # (bf16: the option goes in before the training buffers are laid out, so that every activation-typed region of the
            # workspace has 2-byte elements -- 61 -> 31 GiB at batch 256)
            eng = Engine(lib, units, acts, B, H, W, x.device, sub_batch=sub_batch, unit_names=names, train=train,
                         slice_lanes=lanes, train_bf16=bf16, input_grad=input_grad)
            self._engines[key] = eng
        return eng

    def set_train_act_dtype(self, dtype: str = "fp32"):
        """Storage type of the train step's activations and activation gradients in HBM: "fp32" (default; the reference's
        arithmetic) or "bf16" (BASELINE config 3: bfloat16 storage, fp32 arithmetic / statistics / parameters / optimizer).
        Eval-mode forwards are always fp32."""
        if dtype not in ("fp32", "bf16"):
            raise ValueError("train activation dtype must be 'fp32' or 'bf16'")
        self._train_act_dtype = dtype
        return self

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
            raise ValueError("expected a float32 tensor of shape B x 3 x H x W")
        if self.training:
            return self._forward_train(x)
        eng = self.engine_for(x)
        eng.refresh(self._arena.flat)
        return eng.forward(x)


    def _flop_weight_table(self, names, units):
        """Host table [n_units][3] of the Oct_bn_hook branch weights (csnet.py:393-398); zeros when
        flops_hook() was not called (no hooks registered -> no penalty)."""
        tab = [0.0] * (len(units) * N.MAX_BRANCH)
        if self._penalty_cfg is None:
            return tab
        mods = dict(self.named_modules())
        for ui, (name, u) in enumerate(zip(names, units)):
            sub = mods.get(name)
            if sub is None or getattr(sub, "baseflop", None) is None:   # only ILBlock sub-modules are hooked
                continue
            branches = int(u.n_out)
            w = sub.baseflop * (sub.expandflop ** (branches - 1))
            for k in range(branches):
                tab[ui * N.MAX_BRANCH + k] = w
                w /= sub.expandflop
        return tab

    def _forward_train(self, x):
        """Train-mode forward/backward on the device: batch-statistics BN with running-stat update
        (csnet.py:764,825,138), the dynamic-weight-decay penalty (csnet.py:391-410, read back through
        get_flops()) and, under autograd, the hand-written backward kernels (csn_backward)."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            params = [p for p in self.parameters()]
            y, pen = _CSNetTrainFn.apply(self, x, *params)
        else:
            y, pen = self._train_forward_raw(x, with_backward=False)
            pen = pen.to(torch.float32)[0]
        if self._penalty_cfg is not None:
            # the reference spreads the sum over the ILBlock sub-modules' all_flops; get_flops() only ever reads
            # the total, which is kept on the first hooked module
            first = self.stage0[0].conv1x1
            first.all_flops = first.all_flops + pen
        return y

    def _train_forward_raw(self, x, with_backward=True, y=None, penalty=None):
        """(logits, fp64 device scalar penalty SUM) of one train-mode forward; keeps what csn_backward needs."""
        eng = self.engine_for(x, train=with_backward)
        arena = self._arena
        eng.refresh(arena.flat)                       # weight blocks + PReLU tables; BN tables are rewritten per batch
        units, _, names = self._desc_cache(arena)
        self._flop_tab = self._flop_weight_table(names, units)
        # persistent output / penalty buffers (FusedTrainer) keep every pointer of the step stable, which lets the
        # library replay the step's launch sequence as a hipGraph
        if penalty is None:
            penalty = torch.zeros(1, dtype=torch.float64, device=x.device)
        else:
            penalty.zero_()
        y = eng.forward_train(x, arena.flat, self._flop_tab, penalty, out=y)
        self._train_generation = getattr(self, "_train_generation", 0) + 1
        with torch.no_grad():
            nbt = [m.num_batches_tracked for m in self.modules()
                   if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
            if nbt:
                torch._foreach_add_(nbt, 1)
        return y, penalty

    def _train_backward_raw(self, x, dy, pen_scale, grad=None):
        """Parameter gradients of the last _train_forward_raw call into a flat tensor (arena offsets)."""
        arena = self._arena
        eng = self.engine_for(x, train=True)
        if grad is None:
            grad = torch.zeros(arena.n_param_floats, dtype=torch.float32, device=x.device)
        eng.backward(x.contiguous(), dy.contiguous(), arena.flat, grad, self._flop_tab, pen_scale)
        return grad

    def _desc_cache(self, arena):
        key = id(arena)
        if getattr(self, "_desc", None) is None or self._desc[0] != key:
            self._desc = (key, self.describe(arena.offsets))
        return self._desc[1]


class _CSNetTrainFn(torch.autograd.Function):
    """autograd seam of the train-mode forward: outputs (logits, penalty sum); backward runs csn_backward and hands
    every parameter its slice of the flat gradient (callers keep `loss.backward(); optimizer.step()`, train.py:211-214)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        # x.requires_grad (autograd's x.grad; none of the reference's callers asks for it, train.py:203-216): the step runs on a plan
        # with CSN_OPT_INPUT_GRAD, whose backward also forms the first unit's input gradient
        ctx.input_grad = bool(x.requires_grad)
        model._want_input_grad = ctx.input_grad
        y, pen = model._train_forward_raw(x, with_backward=True)
        ctx.model, ctx.x = model, x
        # backward reads z / activations / BN statistics of THIS forward from the plan's workspace: remember which
        # train-mode forward filled it, so that a second forward before this graph's backward is caught
        ctx.generation = model._train_generation
        return y, pen.to(torch.float32)[0]

    @staticmethod
    def backward(ctx, dy, dpen):
        model = ctx.model
        if ctx.generation != model._train_generation:
            raise RuntimeError("CSNet backward after a newer train-mode forward: the plan keeps the activations of the LAST "
                               "forward only (no gradient accumulation over several forwards / combined losses of two "
                               "forwards); call backward() before the next model(x)")
        pen_scale = float(dpen) if dpen is not None else 0.0
        if dy is None:
            dy = torch.zeros((ctx.x.shape[0], 1) + tuple(ctx.x.shape[2:]), dtype=torch.float32, device=ctx.x.device)
        model._want_input_grad = ctx.input_grad
        flat = model._train_backward_raw(ctx.x, dy, pen_scale)
        dx = model.engine_for(ctx.x, train=True).train_probe(0, "grad0") if ctx.input_grad else None
        model._want_input_grad = False      # (FusedTrainer and plain steps keep the plan without the extra buffer)
        offs = model._arena.offsets
        grads = []
        for name, p in model.named_parameters():
            o = offs[name]
            grads.append(flat[o:o + p.numel()].view(p.shape) if p.requires_grad else None)
        return (None, dx) + tuple(grads)


# ---- dynamic weight decay hook (csnet.py:391-410) -----------------------------------------------------------
def Oct_bn_hook(module, input, output):
    """Same signature and arithmetic as the reference hook, for callers that register it themselves on a
    module whose forward they drive; CSNet.flops_hook() does not need it (the plan fuses the GAP)."""
    branches = len(output)
    w = module.baseflop * (module.expandflop ** (branches - 1))
    weights = []
    for _ in range(branches):
        weights.append(w)
        w /= module.expandflop
    terms = []
    gap_id = 0
    for name, m in module.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            gap = torch.nn.functional.adaptive_avg_pool2d(output[gap_id].detach(), 1).squeeze().abs()
            gap_id += 1
            terms.append((weights[int(name.split('.')[-1])] * gap * torch.pow(m.weight, 2)).sum())
    module.all_flops += 0.5 * sum(terms)


# ---- layer_config helpers (csnet.py:414-568) ----------------------------------------------------------------
def init_layers(basewidth, basic_split=[1, ]):
    """Un-pruned channel plan for stages [3,4,6,4] + CSF head (csnet.py:414-518)."""
    sp = np.array([float(v) for v in basic_split])
    one = np.array([1.0])
    stages = [3, 4, 6, 4]
    w = basewidth
    cfg = [[np.array([3, ]), w * sp], [w * sp, w * sp]]
    cfg += [[w * sp, w * sp] for _ in range(1, stages[0])]
    cfg += [[w * sp, w * 2 * sp]] + [[w * 2 * sp, w * 2 * sp] for _ in range(1, stages[1] - 1)] + [[w * 2 * sp, w * 2 * one]]
    cfg += [[w * 2 * one, w * 4 * sp]] + [[w * 4 * sp, w * 4 * sp] for _ in range(1, stages[2] - 1)] + [[w * 4 * sp, w * 4 * one]]
    cfg += [[w * 4 * one, w * 4 * sp]] + [[w * 4 * sp, w * 4 * sp] for _ in range(1, stages[3] - 1)] + [[w * 4 * sp, w * 4 * one]]
    sides = np.array([w * 2, w * 4, w * 4])
    mid = sides // 3
    cfg.append([sides.copy(), mid.copy()])
    dil = []
    for br in mid:
        each = br // len(DILATIONS)
        dil.append([each] * (len(DILATIONS) - 1) + [br - each * (len(DILATIONS) - 1)])
    cfg.append([mid.copy(), mid.copy(), np.array(dil)])
    cfg.append([mid.copy(), np.array([mid.sum(), ])])
    for e in cfg:
        e[0] = np.round(e[0]).astype(np.int32)
        e[1] = np.round(e[1]).astype(np.int32)
    cfg.append(stages)
    return cfg


def load_layer_config(predefine):
    """Reference ``.bin`` pickle (csnet.py:521-523) or this repo's JSON manifest (``*.json``)."""
    if str(predefine).endswith(".json"):
        with open(predefine) as f:
            raw = json.load(f)["layer_config"]
        return [[np.asarray(v, dtype=np.float64) for v in e] for e in raw[:-1]] + [[int(v) for v in raw[-1]]]
    with open(predefine, "rb") as data:
        return pickle.load(data)


def save_layer_config(layer_config, save_path, epoch, latest=False, finetune=False):
    os.makedirs(save_path, exist_ok=True)
    stem = ("layer_config_finetune_" if finetune else "layer_config_") + str(epoch) + ".bin"
    targets = [os.path.join(save_path, stem)]
    if latest and not finetune:
        targets.append(os.path.join(save_path, "layer_config_latest.bin"))
    for t in targets:
        with open(t, "wb") as f:
            pickle.dump(layer_config, f)
    print("Saved in:", targets[0])


# ---- prune-and-finetune surgery (CSNet_training/model/csnet.py:526-538, 779-879) ---------------------------------
def get_CSFHead_dliconf(mask, old_split):
    """Surviving channels per dilation of every MSBlock: ``mask[i]`` covers branch i's concat of dilation groups."""
    old_split = np.asarray(old_split)
    new_split = np.zeros(old_split.shape)
    for i, branch_mask in enumerate(mask):
        bounds = np.concatenate([[0], np.cumsum(old_split[i]).astype(np.int64)])
        for j in range(old_split.shape[1]):
            new_split[i][j] = int(np.count_nonzero(branch_mask[bounds[j]:bounds[j + 1]]))
    return new_split


def finetune_model(model, save_path, base_layer_config, thres, maxpram=None):
    """Channel masks and the slim ``layer_config`` from a trained model: an output channel of a gOctaveCBR / MSBlock
    survives when ``|BN gamma| >= thres`` (the dynamic weight decay drives unneeded gammas to zero).  The two
    depthwise units of an ILBlock follow their block's conv1x1 mask.  Returns (new_layer_config, out_mask)."""
    thres = float(thres)
    n_layers = len(base_layer_config)
    new_cfg = [None] * n_layers
    out_mask = [None] * n_layers
    stages = base_layer_config[-1]
    layer = 0
    for m in model.modules():
        if not isinstance(m, (gOctaveCBR, PallMSBlock)):
            continue
        this_out = base_layer_config[layer][1]
        gammas = np.concatenate([n.weight.data.detach().cpu().numpy() for n in m.modules() if isinstance(n, nn.BatchNorm2d)])
        keep = np.ones(len(gammas))
        keep[np.abs(gammas) < thres] = 0
        cuts = np.cumsum([int(c) for c in this_out])[:-1]
        mask = np.split(keep, cuts)
        newsplit = np.array([float(np.count_nonzero(b)) for b in mask])
        out_mask[layer] = mask
        if layer == 0:
            new_cfg[layer] = [3, newsplit]
        elif layer == n_layers - 4:          # CSF fuse: its inputs are the high branches of the last block of stages 2-4
            side4 = sum(new_cfg[layer - 1][1])
            side3 = sum(new_cfg[layer - stages[3] - 1][1])
            side2 = sum(new_cfg[layer - stages[3] - stages[2] - 1][1])
            new_cfg[layer] = [np.array([side2, side3, side4]), newsplit]
        elif layer == n_layers - 3:          # PallMSBlock: also the per-dilation split
            new_cfg[layer] = [new_cfg[layer - 1][1], newsplit,
                              get_CSFHead_dliconf(mask, np.asarray(base_layer_config[layer][2]).astype(np.int32))]
        else:
            new_cfg[layer] = [new_cfg[layer - 1][1], newsplit]
        layer += 1
    new_cfg[-1] = stages
    return new_cfg, out_mask


def _keep(mask_list):
    return np.nonzero(np.concatenate([np.asarray(b) for b in mask_list]))[0]


def _copy_bn_prelu(old, new, idx):
    """BN weight / bias and PReLU slope of the surviving channels (running statistics restart, like the reference)."""
    if old is None or new is None:
        return
    sel = torch.as_tensor(np.nonzero(np.asarray(idx))[0], dtype=torch.long)
    with torch.no_grad():
        new.weight.copy_(old.weight.detach().cpu()[sel])
        if getattr(new, "bias", None) is not None and getattr(old, "bias", None) is not None:
            new.bias.copy_(old.bias.detach().cpu()[sel])


def _copy_goct(old, new, this_mask, last_mask):
    rows = torch.as_tensor(_keep(this_mask), dtype=torch.long)
    cols = torch.as_tensor(_keep(last_mask), dtype=torch.long)
    with torch.no_grad():
        new.conv.weight.copy_(old.conv.weight.detach().cpu()[rows][:, cols])
    for j in range(len(this_mask)):
        if j < len(old.bns) and j < len(new.bns):
            _copy_bn_prelu(old.bns[j], new.bns[j], this_mask[j])
            _copy_bn_prelu(old.prelus[j], new.prelus[j], this_mask[j])


def _copy_dw(old, new, this_mask):
    for j in range(len(this_mask)):
        if j >= len(old.convs) or old.convs[j] is None or new.convs[j] is None:
            continue
        sel = torch.as_tensor(np.nonzero(np.asarray(this_mask[j]))[0], dtype=torch.long)
        with torch.no_grad():
            new.convs[j].weight.copy_(old.convs[j].weight.detach().cpu()[sel])
        _copy_bn_prelu(old.bns[j], new.bns[j], this_mask[j])
        _copy_bn_prelu(old.prelus[j], new.prelus[j], this_mask[j])


def _copy_ms(old, new, this_mask, last_mask):
    for i in range(len(this_mask)):
        mo, mn = old.convs[i], new.convs[i]
        if mo is None or mn is None:
            continue
        cols = torch.as_tensor(np.nonzero(np.asarray(last_mask[i]))[0], dtype=torch.long)
        off = 0
        for d in range(len(mo.msconv)):
            if mo.msconv[d] is None:
                continue
            w = mo.msconv[d].weight.detach().cpu()
            seg = np.asarray(this_mask[i])[off:off + w.shape[0]]
            off += w.shape[0]
            if mn.msconv[d] is None:
                continue
            rows = torch.as_tensor(np.nonzero(seg)[0], dtype=torch.long)
            with torch.no_grad():
                mn.msconv[d].weight.copy_(w[rows][:, cols])
        _copy_bn_prelu(mo.bn, mn.bn, this_mask[i])
        _copy_bn_prelu(mo.prelu, mn.prelu, this_mask[i])


def build_model_with_weight(layer_config, old_model, mask, verbose=False):
    """Slim CSNet for ``layer_config`` carrying the surviving channels of ``old_model`` (``mask`` from finetune_model)."""
    model = CSNet(layer_config=layer_config)
    stages = layer_config[-1]
    blocks_old, blocks_new = old_model._blocks(), model._blocks()
    for k, (bo, bn_) in enumerate(zip(blocks_old, blocks_new)):
        last = [np.array([1, 1, 1])] if k == 0 else mask[k - 1]
        _copy_goct(bo.conv1x1, bn_.conv1x1, mask[k], last)
        _copy_dw(bo.conv3x3_1, bn_.conv3x3_1, mask[k])
        _copy_dw(bo.conv3x3_2, bn_.conv3x3_2, mask[k])
    k = len(blocks_old)                       # CSF head: fuse, ms, fuse1x1
    sides = [mask[k - stages[3] - stages[2] - 1][0], mask[k - stages[3] - 1][0], mask[k - 1][0]]
    _copy_goct(old_model.oct_fuse.fuse, model.oct_fuse.fuse, mask[k], sides)
    _copy_ms(old_model.oct_fuse.ms, model.oct_fuse.ms, mask[k + 1], mask[k])
    _copy_goct(old_model.oct_fuse.fuse1x1, model.oct_fuse.fuse1x1, mask[k + 2], mask[k + 1])
    cols = torch.as_tensor(_keep(mask[k + 2]), dtype=torch.long)
    with torch.no_grad():
        model.cls_layer.weight.copy_(old_model.cls_layer.weight.detach().cpu()[:, cols])
        model.cls_layer.bias.copy_(old_model.cls_layer.bias.detach().cpu())
    return model


def build_model(epoch=0, predefine='', basic_split=[1, ], save_path='tmp', expand=1.0, model=None,
                load_weight="NO", finetune_thres='1e-20', finetune=False):
    """Inference signature csnet.py:571-597 plus the training copy's keyword arguments
    (CSNet_training/model/csnet.py:882-945): ``finetune=True`` prunes ``model`` against the layer_config in ``predefine``
    (channels with |BN gamma| < finetune_thres go), writes ``layer_config_finetune_<epoch>.bin`` and, with
    ``load_weight='FINETUNE'`` and ``epoch != 0``, returns the slim model carrying the surviving weights.
    ``redefine_model`` (re-widening during training, csnet.py:940-944) is not provided."""
    basewidth = 20
    real_width = int(round(basewidth * expand)) if expand > 1 else basewidth
    out_mask = None
    if finetune:
        layer_config, out_mask = finetune_model(model, base_layer_config=load_layer_config(predefine), save_path=save_path,
                                                thres=finetune_thres)
        save_layer_config(layer_config, save_path, epoch, finetune=True)
    elif os.path.isfile(predefine):
        layer_config = load_layer_config(predefine)
    else:
        if epoch != 0:
            raise NotImplementedError("redefine_model (re-widening at epoch > 0) is undefined in the reference as well "
                                      "(CSNet_training/model/csnet.py:918)")
        layer_config = init_layers(real_width, basic_split)
        save_layer_config(layer_config, save_path, epoch, latest=True)     # layer_config_0.bin + layer_config_latest.bin
    if out_mask is not None and load_weight == 'FINETUNE' and epoch != 0:
        newmodel = build_model_with_weight(layer_config=layer_config, old_model=model, mask=out_mask)
    else:
        newmodel = CSNet(layer_config=layer_config)
    # CSNet_training/model/csnet.py:933-945 saves the freshly built network at epoch 0.  The inference copy of build_model
    # (CSNet/model/csnet.py:571-597) has no such side effect; its callers never pass save_path, so the file is written
    # whenever a save_path was given or a new layer_config was initialised.
    if epoch == 0 and not finetune and (save_path != 'tmp' or not os.path.isfile(predefine)):
        ckdir = os.path.join(save_path, 'checkpoint')
        os.makedirs(ckdir, exist_ok=True)
        init_save_file = os.path.join(ckdir, 'checkpoint_init.pth.tar')
        torch.save({'epoch': -1, 'arch': "CSNet", 'state_dict': newmodel.state_dict()}, init_save_file)
        print("Save init model:", init_save_file)
    return newmodel
