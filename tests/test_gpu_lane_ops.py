"""Pins the lane maps of the CPU emulator's lane-exact mode (tests/emu/hip_cpu_shim.h, `make LANES=1`) to the hardware: one wave
executes ONE real instruction (tests/emu/lane_probe.hip, compiled here by hipcc) next to the shim's form of it
(csn_emu_lane_probe) on the same per-lane operands.  Integer operands -> every summation order is exact -> bit-for-bit; a second
round with random values bounds the rounding difference of the bf16 forms' internal accumulation order (fp32: one fused
multiply-add per k, bit-for-bit as well).  With this green, `tests/test_emu_lanes.py` in the container certifies the kernels' lane
maps against what the MI355X actually does."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import lane_ops_cases as L

pytestmark = pytest.mark.gpu
EMU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def gpu_probe():
    import emu_build
    so = emu_build.lane_probe_make()
    return L.bind(ctypes.CDLL(so), "lane_probe_run")


@pytest.fixture(scope="module")
def emu_probe(emu_lanes_lib):
    return L.bind(emu_lanes_lib[1], "csn_emu_lane_probe")


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5])
def test_gpu_mfma_lane_maps(gpu_probe, emu_probe, kind):
    rng = np.random.default_rng(100 + kind)
    nreg = 16 if kind == 4 else 4
    for _ in range(8):
        a, b, acc = L.operands(kind, rng)
        ra, rb = L.pack(kind, a, b)
        hw, emu = L.probe(gpu_probe, kind, ra, rb, acc), L.probe(emu_probe, kind, ra, rb, acc)
        assert np.array_equal(hw[:, :nreg].view(np.uint32), emu[:, :nreg].view(np.uint32)), L.KINDS[kind]
        assert np.array_equal(hw[:, :nreg].astype(np.float64), L.define(kind, a, b, acc)[:, :nreg]), L.KINDS[kind]
    worst = 0.0
    for _ in range(8):
        a, b, acc = L.operands(kind, rng, integers=False)
        ra, rb = L.pack(kind, a, b)
        hw, emu = L.probe(gpu_probe, kind, ra, rb, acc), L.probe(emu_probe, kind, ra, rb, acc)
        want = L.define(kind, a, b, acc)[:, :nreg]
        scale = np.abs(want).max()
        worst = max(worst, np.abs(hw[:, :nreg] - emu[:, :nreg]).max() / scale)
        assert np.abs(hw[:, :nreg] - want).max() <= 4e-6 * scale
        assert np.abs(emu[:, :nreg] - want).max() <= 4e-6 * scale
    print(f"{L.KINDS[kind]}: hardware vs emulator on random operands, worst {worst:.2e} of the largest element")


def test_gpu_lane_moves(gpu_probe, emu_probe):
    rng = np.random.default_rng(7)
    v = rng.integers(1, 2 ** 31, size=64).astype(np.uint32)
    ra = np.zeros((64, 16), np.uint8); ra[:, :4] = v.view(np.uint8).reshape(64, 4)
    zero = np.zeros((64, 16), np.float32)
    masks = [np.zeros(64, bool)] + [rng.random(64) < p for p in (0.5, 0.8, 0.2)]
    for kind in (6, 7, 8):
        for i, on in enumerate(masks):
            rb = np.zeros((64, 16), np.uint8); rb[on, 0] = 1
            k = kind + (16 if i else 0)
            if i and kind == 8 and not on.any():
                continue
            hw = L.probe(gpu_probe, k, ra, rb, zero)[:, 0].copy().view(np.uint32)
            emu = L.probe(emu_probe, k, ra, rb, zero)[:, 0].copy().view(np.uint32)
            assert np.array_equal(hw, emu), (L.KINDS[kind], "masked" if i else "all lanes")


def test_gpu_shuffle_xor(gpu_probe, emu_probe):
    """__shfl_xor of a 64-bit value (the fp64 butterflies of csn_reduce.h)"""
    rng = np.random.default_rng(9)
    v = rng.integers(1, 2 ** 62, size=64).astype(np.uint64)
    ra = np.zeros((64, 16), np.uint8); ra[:, :8] = v.view(np.uint8).reshape(64, 8)
    zero = np.zeros((64, 16), np.float32)
    for x in (1, 2, 4, 8, 16, 32, 5, 63):
        rb = np.zeros((64, 16), np.uint8); rb[:, 0] = x
        hw = np.ascontiguousarray(L.probe(gpu_probe, 10, ra, rb, zero)[:, :2]).view(np.uint64)[:, 0]
        emu = np.ascontiguousarray(L.probe(emu_probe, 10, ra, rb, zero)[:, :2]).view(np.uint64)[:, 0]
        assert np.array_equal(hw, emu) and np.array_equal(emu, v[np.arange(64) ^ x]), x


def test_gpu_lane_xor_and_wave_reduce_scatter(gpu_probe, emu_probe):
    """Round 6 (pw4_kernel's statistics epilogue): csn_lane_xor_f32 -- quad_perm for 1 / 2, row_shl:4 + row_shr:4 under bank masks
    for 4, row_ror:8 for 8, ds_bpermute for 16 / 32 -- and csn_wave_reduce_scatter8 built on it, the kernels' own code
    (csn_device.h) on one wave against their definitions.  Integer operands: every summation order is exact."""
    rng = np.random.default_rng(11)
    lanes = np.arange(64)
    zero = np.zeros((64, 16), np.float32)
    v = rng.integers(-2 ** 20, 2 ** 20, size=64).astype(np.float32)
    ra = np.zeros((64, 16), np.uint8); ra[:, :4] = v.view(np.uint8).reshape(64, 4)
    for x in (1, 2, 4, 8, 16, 32):
        rb = np.zeros((64, 16), np.uint8); rb[:, 0] = x
        hw, emu = L.probe(gpu_probe, 11, ra, rb, zero)[:, 0], L.probe(emu_probe, 11, ra, rb, zero)[:, 0]
        assert np.array_equal(hw, v[lanes ^ x]) and np.array_equal(emu, hw), x
    for _ in range(8):
        vals = rng.integers(-1000, 1001, size=(64, 8)).astype(np.float32)
        acc = np.zeros((64, 16), np.float32); acc[:, :8] = vals
        z8 = np.zeros((64, 16), np.uint8)
        out = L.probe(gpu_probe, 12, z8, z8, acc)
        assert np.array_equal(L.probe(emu_probe, 12, z8, z8, acc)[:, :2], out[:, :2])
        idx = ((lanes & 8) >> 1) | ((lanes & 16) >> 3) | ((lanes & 32) >> 5)
        assert np.array_equal(out[:, 1].astype(np.int64), idx)
        assert np.array_equal(out[:, 0].astype(np.float64), vals.astype(np.float64).sum(axis=0)[idx])
