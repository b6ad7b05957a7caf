"""Parameter containers with the reference's names for ``CSF+Res2Net/networks/gOctConv.py``.

The arithmetic of ``gOctaveConv.forward`` (gOctConv.py:60-114) and of the GroupNorm + PReLU that follows it
(gOctConv.py:141-152) runs in libcsnet_hip.so (``csf_gemm_kernel`` & co., include/csf_hip.h); these modules only
own the parameters under the reference's ``state_dict`` keys (``conv.weights``, ``bns.j.*``, ``prelus.j.weight``) and
the channel-boundary rule.  Calling them directly raises: there is no eager path.
"""
import math

import torch
import torch.nn as nn
from torch.nn import init

up_kwargs = {'mode': 'bilinear'}


def branch_bounds(channels, alpha):
    """gOctConv.py:33-42 (running float sums of alpha) and 78-83 (``int(round(C * a))``)."""
    acc, out = 0, [0]
    for a in alpha:
        acc += a
        out.append(int(round(channels * acc)))
    return out


class gOctaveConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, alpha_in=[0.5, 0.5], alpha_out=[0.5, 0.5], stride=1,
                 padding=1, dilation=1, groups=1, bias=False, up_kwargs=up_kwargs):
        super().__init__()
        if groups != 1 or bias or stride != 1 or dilation != 1:
            raise NotImplementedError("the CSF head uses groups=1, stride=1, dilation=1, bias=False (csf_res2net.py:240-244)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        self.weights = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size[0], kernel_size[1]))
        self.register_parameter('bias', None)
        self.alpha_in, self.alpha_out = list(alpha_in), list(alpha_out)
        self.inbranch, self.outbranch = len(alpha_in), len(alpha_out)
        init.kaiming_uniform_(self.weights, a=math.sqrt(5))          # gOctConv.py:53-55

    def in_bounds(self):
        return branch_bounds(self.in_channels, self.alpha_in)

    def out_bounds(self):
        return branch_bounds(self.out_channels, self.alpha_out)

    def forward(self, xset):
        raise RuntimeError("gOctaveConv is computed inside the fused HIP head; call CSFNet.forward")


class gOctaveCBR(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), alpha_in=[0.5, 0.5], alpha_out=[0.5, 0.5], stride=1,
                 padding=1, dilation=1, groups=1, bias=False, up_kwargs=up_kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.std_conv = False
        if len(alpha_in) == 1 and len(alpha_out) == 1:
            raise NotImplementedError("single-branch gOctaveCBR (std_conv, gOctConv.py:121-123) is not used by CSFNet")
        self.conv = gOctaveConv(in_channels, out_channels, kernel_size, alpha_in, alpha_out, stride, padding, dilation,
                                groups, bias, up_kwargs)
        self.bns = nn.ModuleList()
        self.prelus = nn.ModuleList()
        for a in alpha_out:                                           # gOctConv.py:126-133
            c = int(round(out_channels * a))
            self.bns.append(nn.GroupNorm(32, c) if c != 0 else None)
            self.prelus.append(nn.PReLU(c) if c != 0 else None)
        self.outbranch = len(alpha_out)
        self.alpha_in, self.alpha_out = alpha_in, alpha_out

    def forward(self, xset):
        raise RuntimeError("gOctaveCBR is computed inside the fused HIP head; call CSFNet.forward")
