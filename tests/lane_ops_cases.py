"""Operands and matrix definitions for the cross-lane instructions the csnet kernels use (TEST INFRASTRUCTURE).

`probe(fn, kind, ...)` drives one wave through ONE instruction: `fn` is csn_emu_lane_probe of the emulator's lane-exact library
(tests/emu/emu_impl.cpp) or lane_probe_run of the GPU twin (tests/emu/lane_probe.hip).  `define(kind, a, b, acc)` states what the
instruction computes as a MATRIX product (CDNA3/4 ISA, "matrix arithmetic instructions": which lane / register holds which element),
written with numpy indexing only -- no shared code with hip_cpu_shim.h."""
import ctypes

import numpy as np

KINDS = {1: "v_mfma_f32_4x4x1_16b_f32", 2: "v_mfma_f32_4x4x4_16b_bf16", 3: "v_mfma_f32_16x16x4_f32", 4: "v_mfma_f32_32x32x16_bf16",
         5: "v_mfma_f32_16x16x32_bf16", 6: "v_mov_b32_dpp wave_shr:1 bound_ctrl:0", 7: "v_mov_b32_dpp wave_shl:1 bound_ctrl:0",
         8: "v_readfirstlane_b32"}


def bind(cdll, name):
    fn = getattr(cdll, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
    return fn


def bf16_bits(x):
    """float32 array (values exactly representable in bfloat16) -> uint16 bit patterns"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    assert not (u & 0xffff).any(), "operand is not a bfloat16 value"
    return (u >> 16).astype(np.uint16)


def operands(kind, rng, integers=True):
    """per-lane operand values: a, b as float arrays [64][n] (n = k values per lane), acc [64][16]"""
    n = {1: 1, 2: 4, 3: 1, 4: 8, 5: 8}[kind]
    if integers:   # every product and partial sum exact in fp32 whatever the order
        a = rng.integers(-8, 9, size=(64, n)).astype(np.float32)
        b = rng.integers(-8, 9, size=(64, n)).astype(np.float32)
        acc = rng.integers(-64, 65, size=(64, 16)).astype(np.float32)
    else:
        a = rng.standard_normal((64, n)).astype(np.float32)
        b = rng.standard_normal((64, n)).astype(np.float32)
        acc = rng.standard_normal((64, 16)).astype(np.float32)
        if kind in (2, 4, 5):   # bfloat16 operands: drop the low mantissa half
            a = (a.view(np.uint32) & 0xffff0000).view(np.float32)
            b = (b.view(np.uint32) & 0xffff0000).view(np.float32)
    return a, b, acc


def pack(kind, a, b):
    """-> the [64][16]-byte register images of the probes"""
    ra = np.zeros((64, 16), np.uint8)
    rb = np.zeros((64, 16), np.uint8)
    if kind in (1, 3):
        ra[:, :4] = a.astype(np.float32).view(np.uint8).reshape(64, 4)
        rb[:, :4] = b.astype(np.float32).view(np.uint8).reshape(64, 4)
    else:
        n = a.shape[1]
        ra[:, :2 * n] = bf16_bits(a).view(np.uint8).reshape(64, 2 * n)
        rb[:, :2 * n] = bf16_bits(b).view(np.uint8).reshape(64, 2 * n)
    return ra, rb


def probe(fn, kind, ra, rb, acc):
    ra = np.ascontiguousarray(ra, np.uint8); rb = np.ascontiguousarray(rb, np.uint8)
    acc = np.ascontiguousarray(acc, np.float32)
    out = np.zeros((64, 16), np.float32)
    st = fn(kind, ra.ctypes.data, rb.ctypes.data, acc.ctypes.data, out.ctypes.data)
    assert st == 0, f"probe failed: {st}"
    return out


def define(kind, a, b, acc):
    """D = A B + C of the instruction, returned in the per-lane register layout [64][16] (float64 arithmetic)"""
    a = a.astype(np.float64); b = b.astype(np.float64)
    out = acc.astype(np.float64).copy()
    lanes = np.arange(64)
    if kind in (1, 2):          # 16 blocks of 4 lanes: A[i][k] from lane 4 blk + i, B[k][j] from lane 4 blk + j, D[i][j] -> lane 4 blk + j, reg i
        for blk in range(16):
            A = a[4 * blk:4 * blk + 4, :]            # [i][k]
            B = b[4 * blk:4 * blk + 4, :].T          # [k][j]
            D = A @ B                                # [i][j]
            for j in range(4):
                out[4 * blk + j, :4] += D[:, j]
    elif kind == 3:             # one block: A[i][k] from lane 16 k + i, B[k][j] from lane 16 k + j, D[4 (l / 16) + r][l % 16] -> lane l, reg r
        A = a[:, 0].reshape(4, 16).T                 # [i][k]
        B = b[:, 0].reshape(4, 16)                   # [k][j]
        D = A @ B
        for l in lanes:
            out[l, :4] += D[4 * (l // 16):4 * (l // 16) + 4, l % 16]
    elif kind == 4:             # A[i][8 g + e] from lane 32 g + i element e; D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32] -> lane l, reg r
        A = np.zeros((32, 16)); B = np.zeros((16, 32))
        for l in lanes:
            A[l % 32, 8 * (l // 32):8 * (l // 32) + 8] = a[l]
            B[8 * (l // 32):8 * (l // 32) + 8, l % 32] = b[l]
        D = A @ B
        for l in lanes:
            for r in range(16):
                out[l, r] += D[8 * (r // 4) + 4 * (l // 32) + r % 4, l % 32]
    elif kind == 5:             # A[i][8 g + e] from lane 16 g + i element e; D[4 (l / 16) + r][l % 16] -> lane l, reg r
        A = np.zeros((16, 32)); B = np.zeros((32, 16))
        for l in lanes:
            A[l % 16, 8 * (l // 16):8 * (l // 16) + 8] = a[l]
            B[8 * (l // 16):8 * (l // 16) + 8, l % 16] = b[l]
        D = A @ B
        for l in lanes:
            out[l, :4] += D[4 * (l // 16):4 * (l // 16) + 4, l % 16]
    return out
