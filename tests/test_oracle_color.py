"""The oracle's colour chain against exact f64 formulas: XYB cube + opsin inverse, IEC 61966-2-1
sRGB, SMPTE ST 2084 (PQ)."""
import ctypes as C

import numpy as np

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import OPSIN_BIAS, OPSIN_INV


def _run(oracle, xyb, cp):
    a = np.ascontiguousarray(xyb, dtype=np.float32).copy()
    arr = (oracle.f32p * 3)(*[a[c].ctypes.data_as(oracle.f32p) for c in range(3)])
    oracle.lib().orc_color_transform(arr, a[0].size, C.byref(cp))
    return a


def _params(tf, intensity=255.0):
    cp = abi.ColorParams()
    cp.enabled = 1
    cp.opsin_bias[:] = [OPSIN_BIAS] * 3
    cp.intensity_target = intensity
    cp.matrix[:] = list(OPSIN_INV)
    cp.transfer_function = tf
    return cp


def _linear_rgb_f64(xyb, intensity):
    ob = float(OPSIN_BIAS)
    c = np.cbrt(ob)
    x, y, b = [v.astype(np.float64) for v in xyb]
    lms = np.stack([(y + x - c) ** 3 + ob, (y - x - c) ** 3 + ob, (b - c) ** 3 + ob]) * (255.0 / intensity)
    return np.tensordot(OPSIN_INV.astype(np.float64).reshape(3, 3), lms, axes=1)


def _xyb_samples(n=20000, seed=0):
    rng = np.random.default_rng(seed)
    y = rng.uniform(0.0, 0.85, n)
    x = rng.uniform(-0.02, 0.02, n)
    b = y + rng.uniform(-0.1, 0.1, n)
    return np.stack([x, y, b]).astype(np.float32)


def test_xyb_to_linear(oracle):
    xyb = _xyb_samples()
    got = _run(oracle, xyb, _params(abi.TF_LINEAR))
    exp = _linear_rgb_f64(xyb, 255.0)
    assert np.allclose(got, exp, atol=2e-5)


def test_srgb_transfer(oracle):
    xyb = _xyb_samples(seed=1)
    got = _run(oracle, xyb, _params(abi.TF_SRGB))
    lin = _linear_rgb_f64(xyb, 255.0)
    a = np.abs(lin)
    exp = np.sign(lin) * np.where(a <= 0.0031308, 12.92 * a, 1.055 * a ** (1 / 2.4) - 0.055)
    assert np.allclose(got, exp, atol=3e-4)  # libjxl's fast polynomial, documented ~1e-4 accuracy


def test_pq_transfer(oracle):
    xyb = _xyb_samples(seed=2)
    it = 4000.0
    got = _run(oracle, xyb, _params(abi.TF_PQ, it))
    lin = _linear_rgb_f64(xyb, it)
    a = np.abs(lin) * it / 10000.0
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    exp = np.sign(lin) * ((c1 + c2 * a ** m1) / (1 + c3 * a ** m1)) ** m2
    ok = np.abs(lin) > 1e-4
    assert np.allclose(got[ok], exp[ok], atol=2e-4)
