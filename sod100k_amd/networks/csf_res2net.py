"""Drop-in for ``CSF+Res2Net/networks/csf_res2net.py`` (``build_model() -> CSFNet``; BASELINE config 5).

Same module tree and ``state_dict`` keys as the reference (``base.*`` Res2Net-50 v1b 26w4s, ``fuse.*``, ``ms.*``,
``fuse1x1.*``, ``cls_layer.*``), same tensor contract (float32 B x 3 x H x W -> float32 B x 1 x H x W logits).

Split of the work:
  * the decoder head -- gOctConv cross-stage fusion, GroupNorm + PReLU, the dense dilated MSBlocks, fuse1x1, cls_layer
    and the final resize (csf_res2net.py:240-255), 16.2 of 38.4 GFLOP per 352 x 352 image -- runs in libcsnet_hip.so
    (include/csf_hip.h: implicit-GEMM MFMA kernel + fused combine/GroupNorm passes);
  * the backbone's plain convolutions (csf_res2net.py:26-169) are issued through PyTorch-ROCm (MIOpen), exactly the
    "backbone can stay on MIOpen" boundary of SURVEY 8 f-1.  ``Res2Net`` below is an ordinary ``nn.Module``; in eval mode
    on the device every BatchNorm (+ residual) + ReLU that follows a convolution is one in-place launch of ``csf_bn_act``.
There is no CPU path for the head: without the HIP library ``CSFNet.forward`` raises.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from sod100k_amd import _native as N
from sod100k_amd.engine import ParamArena
from sod100k_amd.networks.gOctConv import gOctaveCBR, gOctaveConv  # noqa: F401  (re-exported like the reference)

affine_par = True


def _frozen_bn(c):
    bn = nn.BatchNorm2d(c, affine=affine_par)
    for p in bn.parameters():                 # csf_res2net.py:48-49,63-68: backbone BN affine parameters are frozen
        p.requires_grad = False
    return bn


def _bn_act(owner, bn, y, residual=None, relu=True):
    """Eval BatchNorm (+ residual) (+ ReLU) after a backbone convolution.  On a ROCm device (or with the emulated library
    injected through ``owner._lib`` by a test) this is ONE in-place launch of ``csf_bn_act`` (include/csf_hip.h) instead
    of PyTorch's separate BatchNorm / add / ReLU kernels; training / autograd / CPU tensors take the plain modules."""
    lib = getattr(owner, "_lib", None)
    fused = (y.is_cuda or lib is not None) and not bn.training and not torch.is_grad_enabled() and y.dtype == torch.float32
    if not fused:
        y = bn(y)
        if residual is not None:
            y = y + residual
        return torch.relu_(y) if relu else y
    if lib is None:
        lib = N.load()
    y = y.contiguous()
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == y.shape
    b, c, h, w = y.shape
    stream = torch.cuda.current_stream(y.device).cuda_stream if y.is_cuda else 0
    N.check(lib, lib.csf_bn_act(y.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                bn.running_var.data_ptr(), bn.eps, residual.data_ptr() if residual is not None else None,
                                b, c, h * w, 1 if relu else 0, stream), "csf_bn_act")
    return y


class Bottle2neck(nn.Module):
    """Res2Net bottleneck, scale 4 / width 26 (csf_res2net.py:26-103)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation_=1, downsample=None, baseWidth=26, scale=4, stype='normal'):
        super().__init__()
        width = int(math.floor(planes * (baseWidth / 64.0)))
        self.conv1 = nn.Conv2d(inplanes, width * scale, kernel_size=1, bias=False)
        self.bn1 = _frozen_bn(width * scale)
        self.nums = 1 if scale == 1 else scale - 1
        if stype == 'stage':
            self.pool = nn.AvgPool2d(kernel_size=3, stride=stride, padding=1)
        self.convs = nn.ModuleList(nn.Conv2d(width, width, kernel_size=3, stride=stride, dilation=dilation_,
                                             padding=dilation_, bias=False) for _ in range(self.nums))
        self.bns = nn.ModuleList(_frozen_bn(width) for _ in range(self.nums))
        self.conv3 = nn.Conv2d(width * scale, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = _frozen_bn(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stype, self.scale, self.width = stype, scale, width

    def forward(self, x):
        chunks = torch.split(_bn_act(self, self.bn1, self.conv1(x)), self.width, 1)
        pieces, carry = [], None
        for i in range(self.nums):
            carry = chunks[i] if (i == 0 or self.stype == 'stage') else carry + chunks[i]
            carry = _bn_act(self, self.bns[i], self.convs[i](carry))
            pieces.append(carry)
        if self.scale != 1:
            pieces.append(self.pool(chunks[self.nums]) if self.stype == 'stage' else chunks[self.nums])
        if self.downsample is None:
            shortcut = x
        else:       # AvgPool2d -> conv1x1 -> BatchNorm (csf_res2net.py:136-144)
            shortcut = _bn_act(self, self.downsample[2], self.downsample[1](self.downsample[0](x)), relu=False)
        return _bn_act(self, self.bn3, self.conv3(torch.cat(pieces, 1)), residual=shortcut)


class Res2Net(nn.Module):
    """v1b backbone (3x3 deep stem, avg-pool shortcuts); returns the four stage outputs (csf_res2net.py:105-169)."""

    def __init__(self, block, layers, baseWidth=26, scale=4):
        super().__init__()
        self.inplanes = 64
        self.baseWidth, self.scale = baseWidth, scale
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 64, 3, 1, 1, bias=False))
        self.bn1 = _frozen_bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def load_pretrained_model(self, model):
        self.load_state_dict(model, strict=False)

    def _make_layer(self, block, planes, blocks, stride=1, dilation__=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion or dilation__ in (2, 4):
            downsample = nn.Sequential(
                nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True, count_include_pad=False),
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=1, bias=False),
                nn.BatchNorm2d(planes * block.expansion, affine=affine_par))
            for p in downsample[1].parameters():          # csf_res2net.py:145-146 (sic: the 1x1 conv is frozen)
                p.requires_grad = False
        seq = [block(self.inplanes, planes, stride, dilation_=dilation__, downsample=downsample, stype='stage',
                     baseWidth=self.baseWidth, scale=self.scale)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, dilation_=dilation__, baseWidth=self.baseWidth, scale=self.scale)
                for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        stem = self.conv1
        x = _bn_act(self, stem[1], stem[0](x))
        x = _bn_act(self, stem[4], stem[3](x))
        x = self.maxpool(_bn_act(self, self.bn1, stem[6](x)))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(x)
        return feats


class MSBlock(nn.Module):
    """Five dense dilated 3x3 convolutions + GroupNorm + PReLU (csf_res2net.py:192-223); parameters only."""

    def __init__(self, in_channels, out_channels, dilations=[1, 2, 4, 8, 16]):
        super().__init__()
        self.dilations = dilations
        each = out_channels // 5
        self.msconv = nn.ModuleList()
        for i, d in enumerate(dilations):
            co = each if i != len(dilations) - 1 else out_channels - each * (len(dilations) - 1)
            self.msconv.append(nn.Conv2d(in_channels, co, 3, padding=d, dilation=d, bias=False))
        self.bn = nn.GroupNorm(32, out_channels)
        self.prelu = nn.PReLU(out_channels)

    def forward(self, x):
        raise RuntimeError("MSBlock is computed inside the fused HIP head; call CSFNet.forward")


class PallMSBlock(nn.Module):
    def __init__(self, in_channels, out_channels, alpha=[0.5, 0.5], bias=False):
        super().__init__()
        self.std_conv = False
        self.convs = nn.ModuleList(MSBlock(int(round(in_channels * a)), int(round(out_channels * a))) for a in alpha)
        self.outbranch = len(alpha)

    def forward(self, xset):
        raise RuntimeError("PallMSBlock is computed inside the fused HIP head; call CSFNet.forward")


class _HeadParams(nn.Module):
    """The head's sub-modules under their reference names, for ONE flat parameter arena (not a child of CSFNet)."""

    def __init__(self, net):
        super().__init__()
        self.fuse, self.ms, self.fuse1x1, self.cls_layer = net.fuse, net.ms, net.fuse1x1, net.cls_layer


class _HeadEngine:
    """One csf_head plan (fixed batch / feature sizes / output size) + its workspace."""

    def __init__(self, lib, desc, batch, sizes, out_size, device):
        self.lib = lib
        hs = (C.c_int32 * len(sizes))(*[s[0] for s in sizes])
        ws = (C.c_int32 * len(sizes))(*[s[1] for s in sizes])
        head = C.c_void_p()
        N.check(lib, lib.csf_head_create(C.byref(desc), batch, hs, ws, out_size[0], out_size[1], C.byref(head)),
                "csf_head_create")
        self.head = head
        self.batch, self.sizes, self.out_size = batch, sizes, out_size
        self.workspace = torch.empty(max(int(lib.csf_head_workspace_bytes(head)), 16), dtype=torch.uint8, device=device)
        self.macs = int(lib.csf_head_macs(head))

    def __del__(self):
        head = getattr(self, "head", None)
        if head is not None and head.value:
            self.lib.csf_head_destroy(head)
            self.head = None

    @staticmethod
    def _stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def refresh(self, flat):
        N.check(self.lib, self.lib.csf_head_refresh_params(self.head, flat.data_ptr(), flat.numel(), self._stream(flat)),
                "csf_head_refresh_params")

    def forward(self, feats):
        feats = [f.contiguous() for f in feats]
        ptrs = (C.c_void_p * len(feats))(*[f.data_ptr() for f in feats])
        y = torch.empty((self.batch, 1) + tuple(self.out_size), dtype=torch.float32, device=feats[0].device)
        N.check(self.lib, self.lib.csf_head_forward(self.head, ptrs, y.data_ptr(), self.workspace.data_ptr(),
                                                    self._stream(y)), "csf_head_forward")
        return y

    def stage(self, stage, branch):
        """Workspace view of a head stage (tests): 0 fuse, 1 ms (after GroupNorm + PReLU), 2 fuse1x1 (before its GroupNorm)."""
        off, c, h, w = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        N.check(self.lib, self.lib.csf_head_stage_info(self.head, stage, branch, C.byref(off), C.byref(c), C.byref(h),
                                                       C.byref(w)), "csf_head_stage_info")
        n = self.batch * c.value * h.value * w.value
        return self.workspace[off.value:off.value + 4 * n].view(torch.float32).view(self.batch, c.value, h.value, w.value)


class CSFNet(nn.Module):
    def __init__(self, num_classes=1):
        super().__init__()
        if num_classes != 1:
            raise NotImplementedError("the head kernels produce one saliency channel (csf_res2net.py:228)")
        self.base = Res2Net(Bottle2neck, [3, 4, 6, 3], baseWidth=26, scale=4)
        fuse_in_channel = 256 + 512 + 1024 + 2048                    # csf_res2net.py:235-238
        fuse_in_split = [1 / 15, 2 / 15, 4 / 15, 8 / 15]
        fuse_out_channel = 128 + 256 + 512 + 512
        fuse_out_split = [1 / 11, 2 / 11, 4 / 11, 4 / 11]
        self.fuse = gOctaveCBR(fuse_in_channel, fuse_out_channel, kernel_size=(1, 1), padding=0,
                               alpha_in=fuse_in_split, alpha_out=fuse_out_split, stride=1)
        self.ms = PallMSBlock(fuse_out_channel, fuse_out_channel, alpha=fuse_out_split)
        self.fuse1x1 = gOctaveCBR(fuse_out_channel, fuse_out_channel, kernel_size=(1, 1), padding=0,
                                  alpha_in=fuse_out_split, alpha_out=[1, ], stride=1)
        self.cls_layer = nn.Conv2d(fuse_out_channel, num_classes, kernel_size=1)
        object.__setattr__(self, "_head_params", None)
        object.__setattr__(self, "_arena", None)
        object.__setattr__(self, "_engines", {})
        object.__setattr__(self, "_lib", None)                        # tests may inject the emulated library

    # ---- head plumbing --------------------------------------------------------------------------------------
    def _ensure_arena(self):
        if self._arena is None or not self._arena.is_current():
            object.__setattr__(self, "_head_params", _HeadParams(self))
            object.__setattr__(self, "_arena", ParamArena(self._head_params))
            object.__setattr__(self, "_engines", {})
        return self._arena

    def describe_head(self, offsets):
        """csf_head_desc from the module tree: channel boundaries as the reference constructors derive them."""
        d = N.CsfHeadDesc()
        bi, bo = self.fuse.conv.in_bounds(), self.fuse.conv.out_bounds()
        nb = len(bo) - 1
        assert len(bi) - 1 == nb and self.fuse1x1.conv.in_bounds() == bo
        d.n_branch, d.gn_groups = nb, self.fuse.bns[0].num_groups
        d.fuse_w = offsets["fuse.conv.weights"]
        d.fuse1_w = offsets["fuse1x1.conv.weights"]
        for j in range(nb):
            d.cin[j], d.cmid[j] = bi[j + 1] - bi[j], bo[j + 1] - bo[j]
            assert self.fuse.bns[j].num_channels == d.cmid[j] and self.ms.convs[j].bn.num_channels == d.cmid[j]
            d.fuse_gn[j] = N.CsfGnOff(offsets[f"fuse.bns.{j}.weight"], offsets[f"fuse.bns.{j}.bias"],
                                      offsets[f"fuse.prelus.{j}.weight"])
            d.ms_gn[j] = N.CsfGnOff(offsets[f"ms.convs.{j}.bn.weight"], offsets[f"ms.convs.{j}.bn.bias"],
                                    offsets[f"ms.convs.{j}.prelu.weight"])
            for k, conv in enumerate(self.ms.convs[j].msconv):
                assert conv.in_channels == d.cmid[j] and conv.dilation[0] == (1, 2, 4, 8, 16)[k]
                d.ms_split[j][k] = conv.out_channels
                d.ms_w[j][k] = offsets[f"ms.convs.{j}.msconv.{k}.weight"] if conv.out_channels else -1
        d.fuse1_gn = N.CsfGnOff(offsets["fuse1x1.bns.0.weight"], offsets["fuse1x1.bns.0.bias"],
                                offsets["fuse1x1.prelus.0.weight"])
        d.cls_w, d.cls_b = offsets["cls_layer.weight"], offsets["cls_layer.bias"]
        return d

    def head_engine(self, feats, out_size):
        arena = self._ensure_arena()
        dev = feats[0].device
        if dev != arena.flat.device:
            raise RuntimeError(f"features on {dev} but head parameters on {arena.flat.device}")
        lib = self._lib
        if lib is None:
            if not feats[0].is_cuda:
                raise RuntimeError("the CSF head runs on ROCm devices only (hand-written HIP kernels); "
                                   "move the model and the input to the GPU")
            lib = N.load()
        sizes = tuple((int(f.shape[2]), int(f.shape[3])) for f in feats)
        key = (int(feats[0].shape[0]), sizes, tuple(int(v) for v in out_size), dev)
        eng = self._engines.get(key)
        if eng is None:
            eng = _HeadEngine(lib, self.describe_head(arena.offsets), key[0], sizes, key[2], dev)
            self._engines[key] = eng
        return eng

    def head_forward(self, feats, out_size):
        """csf_res2net.py:250-255 from the backbone features onward."""
        for f in feats:
            if f.dtype != torch.float32:
                raise ValueError("the head kernels take float32 features")
        eng = self.head_engine(feats, out_size)
        eng.refresh(self._arena.flat)
        return eng.forward(feats)

    def forward(self, x):
        if self.training:
            raise NotImplementedError("CSFNet: only the inference forward (BASELINE config 5) is implemented")
        with torch.no_grad():
            features = self.base(x)
            return self.head_forward(features, x.shape[2:])


def build_model():
    return CSFNet()


def weights_init(m):
    if isinstance(m, nn.Conv2d):
        m.weight.data.normal_(0, 0.01)
        if m.bias is not None:
            m.bias.data.zero_()
