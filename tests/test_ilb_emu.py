"""Whole-ILBlock launches (sod100k_amd/csrc/k_ilb.hip: conv1x1 -> conv3x3_1 -> conv3x3_2 of CSNet/model/csnet.py:17-76 in one kernel,
the block's planes in LDS) on the CPU emulation of the kernels: against the unit kernels they replace and against the oracle."""
import pytest
import torch

import parity_cases as P

CPU = torch.device("cpu")


@pytest.mark.parametrize("shape,min_blocks,maxpix", [
    ((2, 224, 224), 3, None),     # BASELINE geometry, product setting: stage 4 (28^2 / 14^2 planes)
    ((2, 224, 224), 8, 1024),     # ... with the plane limit lifted: stages 3-4 (the 56^2 / 28^2 planes fill the LDS of a CU)
    ((3, 64, 64), 8, 1024),       # small input: more blocks fit, pooled outputs in front of stride-2 units
    ((2, 96, 160), 8, 1024),      # non-square planes
    ((1, 32, 48), 8, None)])      # 2-wide lowest maps (strips with fewer than four columns)
def test_emu_ilb_matches_unit_kernels_and_oracle(emu_lib, x2_manifest, shape, min_blocks, maxpix):
    env = {"CSN_ILB_MAXPIX": str(maxpix)} if maxpix else None
    n, worst, err = P.check_ilb_vs_unit_kernels(emu_lib, CPU, x2_manifest, *shape, min_blocks=min_blocks, env=env)
    print(f"{shape}: {n} units on ilb_kernel, worst block deviation {worst:.2e}, logits vs oracle {err:.2e}")


def test_emu_ilb_two_tiles_per_group(emu_lib, x2_manifest):
    """CSN_ILB_NT=2: eight output channels per branch and group (the (2, 2) / (2, 0) instantiations)."""
    n, worst, err = P.check_ilb_vs_unit_kernels(emu_lib, CPU, x2_manifest, 2, 64, 64, env={"CSN_ILB_NT": "2", "CSN_ILB_MAXPIX": "1024"})
    print(f"nt 2: {n} units on ilb_kernel, worst block deviation {worst:.2e}, logits vs oracle {err:.2e}")


def test_emu_ilb_x1_network(emu_lib, x1_manifest):
    n, worst, err = P.check_ilb_vs_unit_kernels(emu_lib, CPU, x1_manifest, 2, 224, 224, min_blocks=4, env={"CSN_ILB_MAXPIX": "1024"})
    print(f"x1: {n} units on ilb_kernel, worst block deviation {worst:.2e}, logits vs oracle {err:.2e}")
