"""The oracle's colour chain against exact f64 formulas: XYB cube + opsin inverse, IEC 61966-2-1
sRGB, SMPTE ST 2084 (PQ), BT.709 OETF, pure gamma, and the Rec. ITU-R BT.2408 EETF tone map."""
import ctypes as C

import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import OPSIN_BIAS, OPSIN_INV, SRGB_LUMINANCES, SRGB_TO_P3, configure_color


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


def _oracle_linear(oracle, xyb, cp):
    """The f32 linear values the transfer function receives (same chain, TF switched off)."""
    lin_cp = abi.ColorParams.from_buffer_copy(cp)
    lin_cp.transfer_function = abi.TF_LINEAR
    return _run(oracle, xyb, lin_cp).astype(np.float64)


def test_bt709_transfer(oracle):
    xyb = _xyb_samples(seed=3)
    cp = _params(abi.TF_LINEAR)
    configure_color(cp, "bt709")
    got = _run(oracle, xyb, cp)
    lin = _oracle_linear(oracle, xyb, cp)
    with np.errstate(invalid="ignore"):
        exp = np.where(lin <= 0.018, 4.5 * lin, 1.099 * np.abs(lin) ** 0.45 - 0.099)
    assert np.allclose(got, exp, rtol=2e-6, atol=2e-6)  # fast_powf: rational 2^x / log2 x, ~3e-7 relative


def test_gamma_transfer(oracle):
    xyb = _xyb_samples(seed=4)
    for mode, g in (("gamma22", 1 / 2.2), ("clip_p3_dci", 1 / 2.6)):
        cp = _params(abi.TF_LINEAR)
        configure_color(cp, mode)
        got = _run(oracle, xyb, cp)
        lin = _oracle_linear(oracle, xyb, cp)
        if mode == "clip_p3_dci":  # Clip -> Matrix against the f64 chain
            ref = np.tensordot(np.array(SRGB_TO_P3).reshape(3, 3), np.clip(_linear_rgb_f64(xyb, 255.0), 0.0, 1.0), axes=1)
            assert np.allclose(lin, ref, atol=2e-5)
        exp = np.where(lin <= 1e-7, 0.0, np.abs(lin) ** g)
        assert np.allclose(got, exp, rtol=2e-6, atol=1e-7), mode
        assert (got[lin <= 0] == 0).all()


def _pq_oetf(y):  # y in units of 10000 nits
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    return ((c1 + c2 * y ** m1) / (1 + c3 * y ** m1)) ** m2


def _pq_eotf(e):
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    with np.errstate(invalid="ignore"):
        p = e ** (1 / m2)
    return (np.maximum(p - c1, 0) / (c2 - c3 * p)) ** (1 / m1)


def _bt2408_f64(lin, it, min_nits, target, lum):
    """Rec. ITU-R BT.2408 annex 5 EETF on luminance, RGB scaled by the luminance ratio."""
    y = np.tensordot(np.array(lum), lin, axes=1)  # 1.0 = intensity_target nits
    lb, lw, lmin, lmax = [_pq_oetf(v / 10000.0) for v in (min_nits, it, 0.0, target)]
    e1 = (_pq_oetf(np.abs(y) * it / 10000.0) - lb) / (lw - lb)
    mn, mx = (lmin - lb) / (lw - lb), (lmax - lb) / (lw - lb)
    ks = 1.5 * mx - 0.5
    t = (e1 - ks) / (1 - ks)
    spline = (2 * t ** 3 - 3 * t ** 2 + 1) * ks + (t ** 3 - 2 * t ** 2 + t) * (1 - ks) + (-2 * t ** 3 + 3 * t ** 2) * mx
    e2 = np.where(e1 < ks, e1, spline)
    e3 = e2 + mn * (1 - e2) ** 4
    y_mapped = _pq_eotf(e3 * (lw - lb) + lb) * 10000.0 / it
    ratio = y_mapped / y * (it / target)
    return lin * ratio, y


def test_tone_map_rec2408(oracle):
    xyb = _xyb_samples(seed=5)
    it = 4000.0
    cp = _params(abi.TF_LINEAR, it)
    configure_color(cp, "tone_map_min_nits")
    got = _run(oracle, xyb, cp)
    lin = _linear_rgb_f64(xyb, it)
    exp, y = _bt2408_f64(lin, it, 0.05, 255.0, SRGB_LUMINANCES)
    ok = y > 1e-3
    assert ok.sum() > 10000
    assert np.allclose(got[:, ok], exp[:, ok], rtol=3e-3, atol=3e-4)
    # mapped luminance never exceeds the 255-nit display (1.0 after the rescale), monotone in y
    ym = np.tensordot(np.array(SRGB_LUMINANCES), got.astype(np.float64), axes=1)[ok]
    assert ym.max() <= 1.0 + 1e-3
    order = np.argsort(y[ok])
    assert (np.diff(ym[order]) > -1e-4).all()


def test_tone_map_then_gamut_map_in_gamut(oracle):
    xyb = _xyb_samples(seed=6)
    cp = _params(abi.TF_LINEAR, 1000.0)
    configure_color(cp, "tone_map_srgb")
    got = _run(oracle, xyb, cp)
    assert np.isfinite(got).all()
    # GamutMap divides by max(1, r, g, b): nothing above 1 is left before the sRGB curve
    assert got.max() <= 1.0 + 1e-3
    # and with saturation_factor 0.3 it pulls negative components towards grey, never away from it
    cp2 = _params(abi.TF_LINEAR, 1000.0)
    configure_color(cp2, "tone_map_srgb")
    cp2.tm_gamut_map = 0
    raw = _run(oracle, xyb, cp2)
    assert got.min() >= min(raw.min(), 0.0) - 1e-6


# ---- HLG (Rec. ITU-R BT.2100 table 5): tf.rs:101-160, convert.rs:501-536 / :1021-1032 ----
HLG_A, HLG_B, HLG_C = 0.17883277, 0.28466892, 0.55991073


def _hlg_oetf_f64(e):
    a = np.abs(e)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sign(e) * np.where(a <= 1 / 12, np.sqrt(3 * a), HLG_A * np.log(12 * a - HLG_B) + HLG_C)


def _hlg_gamma(it):
    return 1.2 * 1.111 ** np.log2(it / 1000.0)


def test_hlg_oetf(oracle):
    lib = oracle.lib()
    lib.orc_test_linear_to_hlg.restype = C.c_float
    lib.orc_test_linear_to_hlg.argtypes = [C.c_float]
    xs = np.concatenate([np.linspace(0, 1 / 12, 200), np.linspace(1 / 12, 1.0, 400), np.geomspace(1.0, 50.0, 50)]).astype(np.float32)
    for s in (1.0, -1.0):
        got = np.array([lib.orc_test_linear_to_hlg(float(s * x)) for x in xs], dtype=np.float64)
        assert np.allclose(got, _hlg_oetf_f64(s * xs.astype(np.float64)), rtol=3e-7, atol=1e-7)
    # the two branches meet at 1/12 (0.5) and the curve reaches 1.0 at 1.0, as BT.2100 defines a, b, c
    assert abs(lib.orc_test_linear_to_hlg(1.0 / 12.0) - 0.5) < 1e-6
    assert abs(lib.orc_test_linear_to_hlg(1.0) - 1.0) < 1e-6


def test_hlg_inverse_ootf(oracle):
    """Scene light from display light: rgb_s = rgb_d * Y_d^((1 - gamma) / gamma); unity system gamma around 300 nits is
    skipped altogether (tf.rs:126-128)."""
    lib = oracle.lib()
    rng = np.random.default_rng(9)
    lum = (C.c_float * 3)(*SRGB_LUMINANCES)
    for it in (1000.0, 4000.0, 400.0, 300.0, 295.0, 305.0, 306.0):
        g = _hlg_gamma(it)
        for _ in range(200):
            rgb = rng.uniform(0.01, 1.0, 3).astype(np.float32)
            buf = (C.c_float * 3)(*rgb)
            lib.orc_test_hlg_inverse_oo(buf, lum, C.c_float(it))
            got = np.array(buf[:], dtype=np.float64)
            if 295.0 <= it <= 305.0:
                assert (got == rgb).all()
                continue
            y = float(np.dot(np.array(SRGB_LUMINANCES), rgb.astype(np.float64)))
            assert np.allclose(got, rgb * y ** ((1 - g) / g), rtol=2e-6), it
    # a negative luminance mix has no real power: NaN, as powf returns it
    buf = (C.c_float * 3)(-0.5, -0.5, -0.5)
    lib.orc_test_hlg_inverse_oo(buf, lum, C.c_float(1000.0))
    assert all(np.isnan(v) for v in buf[:])


@pytest.mark.parametrize("mode,it", [("hlg", 1000.0), ("hlg", 300.0), ("hlg", 4000.0), ("pq_to_hlg", 4000.0), ("pq_to_hlg_1000", 1000.0)])
def test_hlg_op_lists(oracle, mode, it):
    """Order of the HLG op lists: [tone map to 1000 nits] -> inverse OOTF -> [GamutMap 0.1] -> OETF, every op against f64."""
    xyb = _xyb_samples(seed=11)
    cp = _params(abi.TF_LINEAR, it)
    configure_color(cp, mode)
    got = _run(oracle, xyb, cp).astype(np.float64)
    # the f32 linear values the HLG ops receive: the oracle's own chain with those ops switched off
    base = abi.ColorParams.from_buffer_copy(cp)
    base.transfer_function, base.tone_map, base.tm_gamut_map, base.hlg_ootf_intensity_target = abi.TF_LINEAR, 0, 0, 0.0
    lin = _run(oracle, xyb, base).astype(np.float64)
    assert np.allclose(lin, _linear_rgb_f64(xyb, it), atol=2e-5)
    y_ok = np.tensordot(np.array(SRGB_LUMINANCES), lin, axes=1) > 1e-3
    assert y_ok.sum() > 10000
    if mode == "hlg":
        g = _hlg_gamma(it)
        if not 295.0 <= it <= 305.0:
            y = np.tensordot(np.array(SRGB_LUMINANCES), lin, axes=1)
            with np.errstate(invalid="ignore"):
                lin = lin * y ** ((1 - g) / g)
        exp = _hlg_oetf_f64(lin)
        assert np.allclose(got[:, y_ok], exp[:, y_ok], rtol=3e-6, atol=3e-7)
        return
    # the GamutMap is checked through what it guarantees (nothing above 1 afterwards) and against the oracle's own
    # chain with it switched off; the ops in front of it against f64
    cp_nog = abi.ColorParams.from_buffer_copy(cp)
    cp_nog.tm_gamut_map = 0
    nog = _run(oracle, xyb, cp_nog).astype(np.float64)
    if mode == "pq_to_hlg":
        mapped, _ = _bt2408_f64(lin, it, 0.0, 1000.0, SRGB_LUMINANCES)
        y = np.tensordot(np.array(SRGB_LUMINANCES), mapped, axes=1)
        g = _hlg_gamma(1000.0)
        with np.errstate(invalid="ignore"):
            mapped = mapped * y ** ((1 - g) / g)
    else:
        mapped = lin
    exp = _hlg_oetf_f64(mapped)
    assert np.allclose(nog[:, y_ok], exp[:, y_ok], rtol=4e-3, atol=4e-4)
    fin = np.isfinite(got).all(axis=0)
    assert fin.sum() > 0.9 * got.shape[1]
    assert got[:, fin].max() <= 1.0 + 1e-6          # OETF(1) = 1 and the GamutMap left nothing above 1
    assert np.abs(got[:, fin] - nog[:, fin]).max() > 1e-3   # and it did something
