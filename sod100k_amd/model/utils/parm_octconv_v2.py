"""Parameter / "FLOPs" counters with the reference's accounting (CSNet/model/utils/parm_octconv_v2.py).

The reference registers forward hooks by class name and pushes one random image through the network
(parm_octconv_v2.py:188-223; the hooks are never removed).  The fused HIP plan never calls the
sub-modules, so the same MAC-style totals are evaluated analytically from the plan description:
  conv (Conv2d / Conv2dX100)  Cout * (k*k*Cin/groups + bias) * Hout * Wout          (:19-33)
  gOctaveConv                 per (i, j) block with int()-truncated channel bounds, avg-pool (k*k+1),
                              max-pool counted with the CONV kernel size, bilinear 9/elem     (:72-132)
  BatchNorm2d 4/elem, PReLU 3/elem                                                            (:159-170)
The trailing F.interpolate of CSNet.forward is a functional call and is not counted by the reference.
"""


def print_model_parm_nums(model):
    total = sum(p.numel() for p in model.parameters())
    print('  + Number of params: %.4fM' % (total / 1e6))
    return total


def _octconv_flops(conv, in_shapes, k):
    flops = 0
    for i in range(conv.inbranch):
        if in_shapes[i] is None:
            continue
        c, h, w = in_shapes[i]
        shape = (1, c, h, w)
        if conv.stride == 2:
            flops += shape[0] * shape[1] * shape[2] * shape[3] * (2 * 2 + 1)
            shape = (shape[0], shape[1], shape[2] / 2, shape[3] / 2)
        for j in range(conv.outbranch):
            bx = int(conv.in_channels * conv.alpha_in[i] / conv.groups)
            ex = int(conv.in_channels * conv.alpha_in[i + 1] / conv.groups)
            by = int(conv.out_channels * conv.alpha_out[j])
            ey = int(conv.out_channels * conv.alpha_out[j + 1])
            sf = 2 ** (i - j)
            kops = k * k * ((ex - bx) / conv.groups)
            if sf > 1:
                flops += kops * shape[0] * (ey - by) * shape[2] * shape[3]
                flops += shape[0] * (ey - by) * (shape[2] * sf) * (shape[3] * sf) * 9
            elif sf < 1:
                flops += shape[0] * (ex - bx) * (shape[2] * sf) * (shape[3] * sf) * (k * k)
                flops += kops * shape[0] * (ey - by) * (shape[2] * sf) * (shape[3] * sf)
            else:
                flops += kops * shape[0] * (ey - by) * shape[2] * shape[3]
    return flops


def print_model_parm_flops(model, inputsize, device=-1):
    from .. import csnet as M
    from sod100k_amd import _native as N

    class _Zero(dict):
        def __missing__(self, key):
            return 0

    units, acts, names = model.describe(_Zero())
    _, H, W = inputsize
    shape_of = lambda a: (acts[a][0], H >> acts[a][1], W >> acts[a][1])
    mods = dict(model.named_modules())
    total = 0
    for u, name in zip(units, names):
        if u.kind == N.UNIT_GOCT:
            cbr = mods[name]
            ins = [shape_of(u.in_act[i]) if u.in_act[i] >= 0 else None for i in range(u.n_in)]
            total += _octconv_flops(cbr.conv, ins, u.ksize)
            for j in range(u.n_out):
                if u.out_act[j] >= 0:
                    c, h, w = shape_of(u.out_act[j])
                    total += c * h * w * (4 + 3)
        elif u.kind == N.UNIT_DW:
            for k in range(u.n_in):
                if u.out_act[k] >= 0:
                    c, h, w = shape_of(u.out_act[k])
                    total += c * 9 * h * w + c * h * w * (4 + 3)
        elif u.kind == N.UNIT_MS:
            c, h, w = shape_of(u.out_act[0])
            total += c * 9 * u.cin[0] * h * w + c * h * w * (4 + 3)
        elif u.kind == N.UNIT_CLS:
            c, h, w = shape_of(u.in_act[0])
            total += 1 * (c + 1) * h * w
    print('  + Number of FLOPs: %.4fG' % (total / 1e9))
    return total
