"""The emulator's lane-exact mode: the same parity cases as the sequential-fiber emulator, through
``tests/emu/libcsnet_emu_lanes.so`` -- the kernels' DEVICE code paths (one lane's row of every MFMA A operand, halo columns by DPP,
readfirstlane) with the cross-lane instructions executed lane-exactly by hip_cpu_shim.h's wave rendezvous.  The default emulator
replaces those operands by functional stand-ins and certifies index arithmetic only; this mode is what catches a wrong lane map
(`lds + (lane & 3) * P`, a DPP shift in the wrong direction, an accumulator register mapped to the wrong row) in the container,
before a GPU lease does.  The lane maps the shim implements are themselves pinned to the hardware by
tests/test_gpu_lane_ops.py (the real instructions next to the shim's, bit for bit).

Instruction kinds counted by csn_emu_lane_ops: 1 v_mfma_f32_4x4x1 (pw4 / c3q / pwq / hz / ilb), 2 v_mfma_f32_4x4x4_bf16 (pwq16 /
c3q16), 3 v_mfma_f32_16x16x4_f32 (generic contraction and weight-gradient kernels), 4 v_mfma_f32_32x32x16_bf16
(wgrad_bf16_kernel), 5 v_mfma_f32_16x16x32_bf16 (wgrad_bf16_c3_kernel), 6 / 7 DPP wave_shr:1 / wave_shl:1 (c3q, ilb, the depthwise
kernels), 8 readfirstlane, 10 __shfl_xor (the fp64 block sums of csn_reduce.h)."""
import torch

from oracle import inputs as I

import parity_cases as P

CPU = torch.device("cpu")


def _ops(raw):
    return [raw.csn_emu_lane_ops(k) for k in range(11)]


def _ran(raw, before, kinds):
    after = _ops(raw)
    d = [a - b for a, b in zip(after, before)]
    for k in kinds:
        assert d[k] > 0, f"no cross-lane instruction of kind {k} was executed: {d}"
    return d


def test_lanes_forward_vs_oracle(emu_lanes_lib, x2_manifest):
    lib, raw = emu_lanes_lib
    b = _ops(raw)
    P.check_vs_oracle(lib, CPU, x2_manifest, torch.from_numpy(I.randn_batch(5, 2, 32, 48)))
    print(_ran(raw, b, (1, 6, 7, 8)))


def test_lanes_golden_nonsquare(emu_lanes_lib, x2_manifest):
    lib, raw = emu_lanes_lib
    x = torch.from_numpy(I.randn_batch(3, 2, 96, 160))
    P.check_golden_logits(lib, CPU, x2_manifest, "g2_logits_x2_randn_b2_96x160.npy", x)


def test_lanes_unit_probes(emu_lanes_lib, x2_manifest):
    lib, raw = emu_lanes_lib
    b = _ops(raw)
    P.check_unit_probes(lib, CPU, x2_manifest)
    print(_ran(raw, b, (1, 3, 6, 7)))


def test_lanes_op_goldens(emu_lanes_lib):
    lib, raw = emu_lanes_lib
    assert len(P.check_g4(lib, CPU)) >= 11


def test_lanes_ilb_hz_and_lane_exchange(emu_lanes_lib, x2_manifest):
    """ilb_kernel (DPP windows + MFMA + depthwise from LDS), hz_kernel, and the depthwise kernels' halo columns by DPP against
    their loaded-halo forms -- here the DPP forms really move registers between lanes."""
    lib, raw = emu_lanes_lib
    b = _ops(raw)
    P.check_ilb_vs_unit_kernels(lib, CPU, x2_manifest, 2, 64, 64)
    P.check_hz_vs_pw4(lib, CPU, x2_manifest, 1, 80, 112, env={"CSN_HZ_RB": "2", "CSN_HZ_NW": "4"}, fuse_cls=False)
    P.check_lane_exchange_vs_loaded_halos(lib, CPU, x2_manifest, 2, 64, 64)
    print(_ran(raw, b, (1, 6, 7)))


def test_lanes_train_step_fp32(emu_lanes_lib, x2_manifest):
    lib, raw = emu_lanes_lib
    b = _ops(raw)
    P.check_train_forward(lib, CPU, x2_manifest, B=3, size=48)
    P.check_train_step(lib, CPU, x2_manifest)
    print(_ran(raw, b, (1, 3, 6, 7, 10)))


def test_lanes_train_step_bf16(emu_lanes_lib, x2_manifest):
    """bf16 storage: pwq16 / c3q16 on v_mfma_f32_4x4x4_bf16 and the weight gradients on the 32x32x16 / 16x16x32 bf16 forms -- the
    kernels the default emulator replaces wholesale by functional stand-ins."""
    lib, raw = emu_lanes_lib
    b = _ops(raw)
    P.check_train_step_bf16(lib, CPU, x2_manifest)
    P.check_train_units_local(lib, CPU, x2_manifest, B=2, size=64, act_dtype="bf16")
    print(_ran(raw, b, (1, 2, 3, 4, 5, 6, 7)))


def test_lanes_instruction_definitions(emu_lanes_lib):
    """The shim's lane maps against the instructions' MATRIX definitions (tests/lane_ops_cases.py: numpy, no shared code), one wave
    and one instruction at a time; integer operands, so every summation order gives the same bits."""
    import numpy as np
    import lane_ops_cases as L
    _, raw = emu_lanes_lib
    fn = L.bind(raw, "csn_emu_lane_probe")
    rng = np.random.default_rng(11)
    for kind in (1, 2, 3, 4, 5):
        for _ in range(4):
            a, b, acc = L.operands(kind, rng)
            got = L.probe(fn, kind, *L.pack(kind, a, b), acc)
            want = L.define(kind, a, b, acc)
            nreg = 16 if kind == 4 else 4
            assert np.array_equal(got[:, :nreg].astype(np.float64), want[:, :nreg]), L.KINDS[kind]
    # the lane moves: lane i <- lane i - 1 / i + 1 (0 where there is none), the first lane's value; then under a lane mask
    v = rng.integers(1, 2 ** 31, size=64).astype(np.uint32)
    ra = np.zeros((64, 16), np.uint8); ra[:, :4] = v.view(np.uint8).reshape(64, 4)
    zero = np.zeros((64, 16), np.float32)
    res = lambda kind, rb: L.probe(fn, kind, ra, rb, zero)[:, 0].copy().view(np.uint32)
    none = np.zeros((64, 16), np.uint8)
    assert np.array_equal(res(6, none), np.concatenate([[0], v[:-1]]).astype(np.uint32))
    assert np.array_equal(res(7, none), np.concatenate([v[1:], [0]]).astype(np.uint32))
    assert np.array_equal(res(8, none), np.full(64, v[0], np.uint32))
    on = (np.arange(64) % 3 != 0) & (np.arange(64) != 1)
    rb = np.zeros((64, 16), np.uint8); rb[on, 0] = 1
    shr = np.where(on, np.where(np.roll(on, 1) & (np.arange(64) > 0), np.roll(v, 1), 0), v).astype(np.uint32)
    shl = np.where(on, np.where(np.roll(on, -1) & (np.arange(64) < 63), np.roll(v, -1), 0), v).astype(np.uint32)
    assert np.array_equal(res(16 + 6, rb), shr)       # a masked-off source lane reads as 0 (bound_ctrl); masked-off lanes keep their value
    assert np.array_equal(res(16 + 7, rb), shl)
    assert np.array_equal(res(16 + 8, rb), np.where(on, v[np.argmax(on)], v).astype(np.uint32))
    v64 = rng.integers(1, 2 ** 62, size=64).astype(np.uint64)
    ra = np.zeros((64, 16), np.uint8); ra[:, :8] = v64.view(np.uint8).reshape(64, 8)
    for x in (1, 32, 21):
        rb = np.zeros((64, 16), np.uint8); rb[:, 0] = x
        got = np.ascontiguousarray(L.probe(fn, 10, ra, rb, zero)[:, :2]).view(np.uint64)[:, 0]
        assert np.array_equal(got, v64[np.arange(64) ^ x])
    # round 6: csn_lane_xor_f32 and the eight-value wave reduce-scatter of pw4_kernel's statistics (csn_device.h on the shim)
    lanes = np.arange(64)
    vf = rng.integers(-2 ** 20, 2 ** 20, size=64).astype(np.float32)
    ra = np.zeros((64, 16), np.uint8); ra[:, :4] = vf.view(np.uint8).reshape(64, 4)
    for x in (1, 2, 4, 8, 16, 32):
        rb = np.zeros((64, 16), np.uint8); rb[:, 0] = x
        assert np.array_equal(L.probe(fn, 11, ra, rb, zero)[:, 0], vf[lanes ^ x]), x
    vals = rng.integers(-1000, 1001, size=(64, 8)).astype(np.float32)
    acc = np.zeros((64, 16), np.float32); acc[:, :8] = vals
    out = L.probe(fn, 12, none, none, acc)
    idx = ((lanes & 8) >> 1) | ((lanes & 16) >> 3) | ((lanes & 32) >> 5)
    assert np.array_equal(out[:, 1].astype(np.int64), idx)
    assert np.array_equal(out[:, 0].astype(np.float64), vals.astype(np.float64).sum(axis=0)[idx])


def test_lanes_csf_head(emu_lanes_lib, monkeypatch):
    """CSF+Res2Net head: csf_gemm3_kernel's real path -- the three-way bfloat16 split of the fp32 operands and six
    v_mfma_f32_32x32x16_bf16 per product -- against the oracle with the GPU test's bounds."""
    from oracle import csf_oracle as CO
    import csf_cases as K
    lib, raw = emu_lanes_lib
    b = _ops(raw)
    net, sd = K.build_csfnet("cpu", lib)
    feats = CO.synthetic_features(3, 2, [(12, 16), (6, 8), (3, 4), (2, 2)])
    y, ref, errs = K.head_errors(net, sd, feats, (48, 64))
    assert errs["logits"] <= 1e-4, errs
    assert max(v for k, v in errs.items() if k.startswith(("fuse.", "ms."))) <= 2e-4, errs
    assert errs["hip_vs_fp64"] <= 3 * errs["oracle_vs_fp64"] + 2e-5, errs
    print(errs, _ran(raw, b, (4,)))
