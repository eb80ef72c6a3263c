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


@pytest.mark.parametrize("predictor", [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("i16", [True, False])
def test_single_leaf_predictors(oracle, predictor, i16):
    """M4: every stateless predictor of a single-leaf tree, against residuals computed by a
    vectorised numpy forward pass over the finished image (tile borders, 1-wide tiles included)."""
    for (w, h) in [(300, 270), (257, 3), (2, 40)]:
        wl = ModularWorkload(w, h, kind="predictor", predictor=predictor, i16=i16, seed=predictor,
                             pred_offset=(0 if predictor % 2 else 3))
        got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
        for c in range(3):
            assert np.array_equal(got[c], wl.expected[c]), f"predictor {predictor} {w}x{h} channel {c}"


@pytest.mark.parametrize("size", [(40, 24), (1, 9), (9, 1), (2, 2), (70, 33)])
def test_self_correcting_predictor(oracle, size):
    """Predictor 6 against a sequential Python transcription of the format's weighted predictor."""
    w, h = size
    wl = ModularWorkload(w, h, kind="predictor", predictor=6, i16=False, seed=w)
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c]), f"channel {c}"


def test_predictor_multiplier_and_wrapping(oracle):
    """multiplier != 1 and i16 wrap-around: value = residual * multiplier + offset + prediction in
    Wrapping<S>; checked against a direct Python loop (West predictor, where it is a running sum)."""
    from jxl_oxide_amd import abi
    wl = ModularWorkload(50, 7, kind="predictor", predictor=1, i16=True, seed=1)
    rng = np.random.default_rng(5)
    res = rng.integers(-3000, 3000, size=(7, 50)).astype(np.int16)
    wl.buffers = [res.copy() for _ in range(3)]
    wl.residual_multiplier, wl.residual_offset = 37, -11
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    exp = np.zeros((7, 50), dtype=np.int64)
    wrap = lambda v: ((int(v) + 32768) % 65536) - 32768
    for y in range(7):
        for x in range(50):
            west = exp[y, x - 1] if x > 0 else (exp[y - 1, 0] if y > 0 else 0)
            exp[y, x] = wrap(wrap(wrap(int(res[y, x]) * 37) - 11) + west)
    assert np.array_equal(got[0].astype(np.int64), exp)


@pytest.mark.parametrize("d_pred", [0, 1, 2, 5])
@pytest.mark.parametrize("i16", [True, False])
def test_palette_with_delta_entries(oracle, d_pred, i16):
    """M3 slow path: implicit colours, DELTA_PALETTE entries, delta palette rows + predictor pass,
    against a direct Python evaluation."""
    from jxl_oxide_amd.synth_modular import palette_delta_reference
    wl = ModularWorkload(37, 21, kind="palette_delta", predictor=d_pred, i16=i16, seed=d_pred)
    got = oracle.modular_inverse(wl.desc(), wl.shapes(), wl.dtype)
    exp = palette_delta_reference(wl.index_plane, wl.palette, 29, 4, d_pred, 8, 16 if i16 else 32)
    for c in range(3):
        assert np.array_equal(got[c].astype(np.int64), exp[c]), f"channel {c}"
