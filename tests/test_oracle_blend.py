"""The oracle's blend_single (jxl-render/src/blend.rs:550-728) against the compositing formulas."""
import ctypes as C

import numpy as np
import pytest

from jxl_oxide_amd import abi


def _rect(mode, w, h, clamp=0, swapped=0, premultiplied=0, base_alpha=None, new_alpha=None, bx=0, by=0, nx=0, ny=0):
    r = abi.BlendRect()
    r.mode, r.clamp, r.swapped, r.premultiplied = mode, clamp, swapped, premultiplied
    if base_alpha is not None:
        r.base_alpha, r.base_alpha_stride = base_alpha.ctypes.data, base_alpha.shape[1]
    if new_alpha is not None:
        r.new_alpha, r.new_alpha_stride = new_alpha.ctypes.data, new_alpha.shape[1]
    r.base_x, r.base_y, r.new_x, r.new_y, r.width, r.height = bx, by, nx, ny, w, h
    return r


def blend(oracle, base, new, rect):
    out = base.copy()
    f = oracle.lib().orc_blend_rect
    f.argtypes = [oracle.f32p, C.c_size_t, oracle.f32p, C.c_size_t, C.c_void_p]
    f.restype = None
    f(out.ctypes.data_as(oracle.f32p), out.shape[1], new.ctypes.data_as(oracle.f32p), new.shape[1], C.byref(rect))
    return out


@pytest.fixture
def planes():
    rng = np.random.default_rng(0)
    mk = lambda lo, hi: rng.uniform(lo, hi, size=(20, 31)).astype(np.float32)
    return dict(base=mk(-0.2, 1.2), new=mk(-0.2, 1.2), ba=mk(-0.1, 1.1), na=mk(-0.1, 1.1))


def test_simple_modes(oracle, planes):
    b, n = planes["base"], planes["new"]
    assert np.array_equal(blend(oracle, b, n, _rect(abi.BLEND_REPLACE, 31, 20)), n)
    assert np.array_equal(blend(oracle, b, n, _rect(abi.BLEND_ADD, 31, 20)), b + n)
    assert np.array_equal(blend(oracle, b, n, _rect(abi.BLEND_MUL, 31, 20, clamp=1)), b * np.clip(n, 0, 1))
    assert np.array_equal(blend(oracle, b, n, _rect(abi.BLEND_SKIP, 31, 20)), b)
    # without a new-alpha plane Blend is Replace and MulAdd is Add (blend.rs:565, 578)
    assert np.array_equal(blend(oracle, b, n, _rect(abi.BLEND_BLEND, 31, 20)), n)
    assert np.array_equal(blend(oracle, b, n, _rect(abi.BLEND_MULADD, 31, 20)), b + n)


@pytest.mark.parametrize("swapped", [0, 1])
@pytest.mark.parametrize("premultiplied", [0, 1])
@pytest.mark.parametrize("clamp", [0, 1])
def test_alpha_blend_is_porter_duff_over(oracle, planes, swapped, premultiplied, clamp):
    b, n, ba, na = [planes[k].astype(np.float64) for k in ("base", "new", "ba", "na")]
    got = blend(oracle, planes["base"], planes["new"],
                _rect(abi.BLEND_BLEND, 31, 20, clamp, swapped, premultiplied, planes["ba"], planes["na"]))
    bot, top, a_bot, a_top = (n, b, na, ba) if swapped else (b, n, ba, na)
    if clamp:
        a_top = np.clip(a_top, 0, 1)
    if premultiplied:
        exp = top + bot * (1 - a_top)
    else:
        mixed = 1 - (1 - a_top) * (1 - a_bot)
        exp = np.where(mixed > 0, (a_top * top + a_bot * bot * (1 - a_top)) / np.where(mixed > 0, mixed, 1), 0)
        # 1 - (1-a)(1-b) cancels in f32 when both alphas are tiny: compare where it is well conditioned
        ok = np.abs(mixed) > 0.05
        assert ok.mean() > 0.8
        got, exp = got[ok], exp[ok]
    assert np.allclose(got, exp, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("swapped", [0, 1])
def test_muladd_and_mixalpha(oracle, planes, swapped):
    b, n, ba, na = [planes[k].astype(np.float64) for k in ("base", "new", "ba", "na")]
    got = blend(oracle, planes["base"], planes["new"], _rect(abi.BLEND_MULADD, 31, 20, 1, swapped, 0, planes["ba"], planes["na"]))
    exp = n + np.clip(ba, 0, 1) * b if swapped else b + np.clip(na, 0, 1) * n
    assert np.allclose(got, exp, atol=1e-6)
    got = blend(oracle, planes["base"], planes["new"], _rect(abi.BLEND_MIXALPHA, 31, 20, 1, swapped))
    bb, nn = (n, b) if swapped else (b, n)
    assert np.allclose(got, bb + np.clip(nn, 0, 1) * (1 - bb), atol=1e-6)


def test_rectangle_offsets(oracle, planes):
    b, n = planes["base"], planes["new"]
    got = blend(oracle, b, n, _rect(abi.BLEND_ADD, 7, 5, bx=3, by=2, nx=20, ny=11))
    exp = b.copy()
    exp[2:7, 3:10] += n[11:16, 20:27]
    assert np.array_equal(got, exp)
