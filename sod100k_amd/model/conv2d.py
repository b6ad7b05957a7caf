"""``Conv2dX100`` parameter container (mirror of CSNet/model/conv2d.py:28-105).

The reference layer is an ``nn.Conv2d`` clone whose forward convolves with ``100.0 * weight``
("for faster convergence", conv2d.py:102-104).  Here the module only owns the parameters (same
names, shapes and default initialisation, so checkpoints load key-for-key); the arithmetic -- including
the x100 -- is done by the HIP kernels of the enclosing unit (csrc/k_misc.hip depthwise units, csrc/k_ms.hip MSBlock, csrc/k_goct_pw.hip / k_goct_c3.hip std_conv).
"""
import math

import torch
import torch.nn as nn
from torch.nn import init


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Conv2dX100(nn.Module):
    WEIGHT_SCALE = 100.0

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=False, padding_mode='zeros', transposed=False, output_padding=0):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')      # conv2d.py:45-46
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')     # conv2d.py:47-48
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.transposed, self.output_padding = transposed, _pair(output_padding)
        self.groups, self.padding_mode = groups, padding_mode
        shape = (in_channels, out_channels // groups) if transposed else (out_channels, in_channels // groups)
        self.weight = nn.Parameter(torch.empty(*shape, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def extra_repr(self):
        return (f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, '
                f'padding={self.padding}, dilation={self.dilation}, groups={self.groups}, x100')

    def forward(self, input):
        raise RuntimeError("Conv2dX100 is a parameter container in sod100k_amd: it is executed as part of its "
                           "enclosing unit by the fused HIP plan (call the CSNet module, not the layer).")
