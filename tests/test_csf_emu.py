"""CSF+Res2Net head on the CPU emulation of the kernels vs the oracle: index logic of the implicit-GEMM
gather (own / bilinear-resampled / dilated-tap segments), the combine + GroupNorm passes, cls + resize."""
import pytest
import torch

from oracle import csf_oracle as CO
import csf_cases as K


@pytest.fixture(scope="module")
def net_sd(emu_lib):
    return K.build_csfnet("cpu", emu_lib)


@pytest.mark.parametrize("sizes,out_size,batch", [
    ([(12, 16), (6, 8), (3, 4), (2, 2)], (48, 64), 2),        # octave-spaced levels, batch straddling pixel tiles
    ([(13, 10), (7, 5), (4, 3), (2, 2)], (50, 38), 1),        # the sizes a 50 x 38 input produces (non-integer ratios)
])
def test_head_matches_oracle(net_sd, sizes, out_size, batch):
    net, sd = net_sd
    feats = CO.synthetic_features(3, batch, sizes)
    y, ref, errs = K.head_errors(net, sd, feats, out_size)
    print(errs)
    assert y.shape == ref.shape
    for k, v in errs.items():
        if k.startswith(("fuse.", "ms.")):
            assert v <= 2e-4, (k, v)
    # logits: within 1e-4 of the fp32 oracle, and not further from the fp64 truth than a few times the oracle itself
    assert errs["logits"] <= 1e-4, errs
    assert errs["hip_vs_fp64"] <= 3 * errs["oracle_vs_fp64"] + 2e-5, errs


def test_head_global_tap_path(emu_lib, monkeypatch):
    """Coarse planes too large for LDS: the combine pass takes its taps from global memory (forced here)."""
    monkeypatch.setenv("CSF_Z_LDS_MAX", "0")
    net, sd = K.build_csfnet("cpu", emu_lib)
    feats = CO.synthetic_features(4, 1, [(8, 8), (4, 4), (2, 2), (1, 1)])
    y, ref, errs = K.head_errors(net, sd, feats, (32, 32))
    assert errs["logits"] <= 1e-4 and max(v for k, v in errs.items() if k.startswith(("fuse.", "ms."))) <= 2e-4, errs


def test_head_requires_device_without_library():
    from sod100k_amd.networks import csf_res2net as R
    net = R.build_model().eval()
    with pytest.raises(RuntimeError):
        net.head_forward(CO.synthetic_features(1, 1, [(4, 4), (2, 2), (1, 1), (1, 1)]), (16, 16))
