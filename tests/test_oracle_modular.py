"""Pins the oracle's Modular inverse transforms with exact round trips: forward transforms written
independently in numpy (jxl_oxide_amd/synth_modular.py, from the definition of the inverse) must
come back bit-for-bit."""
import numpy as np
import pytest

from jxl_oxide_amd.synth_modular import ModularWorkload


@pytest.mark.parametrize("i16", [True, False])
@pytest.mark.parametrize("size", [(256, 256), (70, 45), (9, 200), (1, 17), (33, 1)])
def test_lossless_squeeze_roundtrip(oracle, size, i16):
    w, h = size
    wl = ModularWorkload(w, h, kind="squeeze", lossy=False, xyb=False, i16=i16, seed=w + h)
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c]), f"channel {c}"


@pytest.mark.parametrize("rct_type", [0, 6, 7 + 3, 14 + 5, 21 + 2, 28 + 6, 35 + 1, 41])
def test_rct_squeeze_roundtrip(oracle, rct_type):
    wl = ModularWorkload(120, 90, kind="squeeze", lossy=False, xyb=False, rct_type=rct_type, seed=rct_type)
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c]), f"rct_type {rct_type} channel {c}"


@pytest.mark.parametrize("i16", [True, False])
def test_config1_lossless_rgb8(oracle, i16):
    """BASELINE config 1: 256x256 lossless Modular RGB8 (Gradient residuals + YCoCg RCT)."""
    wl = ModularWorkload(256, 256, kind="lossless_rgb8", i16=i16)
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


def test_gradient_multi_group(oracle):
    wl = ModularWorkload(300, 270, kind="lossless_rgb8", seed=3)
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


def test_palette_simple(oracle):
    wl = ModularWorkload(64, 48, kind="palette")
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


def test_render_xyb_runs(oracle):
    from jxl_oxide_amd import abi
    wl = ModularWorkload(96, 64, kind="squeeze", lossy=True, epf_iters=1)
    out = oracle.modular_render(wl.desc(), abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT, 96, 64)
    assert np.isfinite(out).all()
