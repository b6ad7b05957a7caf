"""hz_kernel (sod100k_amd/csrc/k_head.hip, round 6): the high output of the three-branch 1x1 units of the decoder (CSFHead.fuse /
fuse1x1, CSNet/model/csnet.py:152-206) with the low -> high terms convolved at the low resolution and interpolated per output channel
from LDS (csnet.py:702-707), on the CPU emulation of the kernels: against pw4_kernel's high-only form and against the oracle."""
import pytest
import torch

import parity_cases as P

CPU = torch.device("cpu")


@pytest.mark.parametrize("shape,env,fuse_cls", [
    ((2, 224, 224), None, True),                                        # BASELINE geometry, product settings (cls_layer in the epilogue)
    ((1, 112, 112), None, False),                                       # ... fuse1x1's rows stored
    ((1, 80, 112), {"CSN_HZ_RB": "2", "CSN_HZ_NW": "4"}, False),        # non-square, bands of two rows (the last band is short)
    ((3, 16, 16), {"CSN_HZ_RB": "1"}, True),                            # smallest input: 4 x 4 / 2 x 2 planes, one-row bands
    ((1, 48, 32), {"CSN_HZ_RB": "3", "CSN_HZ_NT": "2/5", "CSN_HZ_HB": "4", "CSN_HZ_NW": "8"}, False),   # three M groups, ragged last band
    ((1, 64, 96), {"CSN_HZ_RB": "14", "CSN_HZ_NW": "16", "CSN_HZ_NT": "3/3"}, True)])   # one band per image, 16 waves
def test_emu_hz_matches_pw4_and_oracle(emu_lib, x2_manifest, shape, env, fuse_cls):
    n, worst, err = P.check_hz_vs_pw4(emu_lib, CPU, x2_manifest, *shape, env=env, fuse_cls=fuse_cls)
    print(f"{shape} {env}: {n} launches on hz_kernel, worst unit deviation {worst:.2e}, logits vs oracle {err:.2e}")


def test_emu_hz_x1_network(emu_lib, x1_manifest):
    n, worst, err = P.check_hz_vs_pw4(emu_lib, CPU, x1_manifest, 2, 96, 96, fuse_cls=False)
    print(f"x1: worst unit deviation {worst:.2e}, logits vs oracle {err:.2e}")
