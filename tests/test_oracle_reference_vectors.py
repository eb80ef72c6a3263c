"""The reference's own unit tests of the colour functions on the path, replayed against the oracle
(VERDICT r2 item 4).  Inputs and tolerances are the reference's:

  * jxl-color/src/convert/tone_map.rs:764-791 `tone_map_range`: ten grey samples (idx / 5) * 0.1, HDR
    parameters {sRGB luminances, intensity_target 10000, min_nits 0}, target display 255 nits, no peak
    detection -> (idx / 5) * 0.8714331 within 2e-5;
  * jxl-color/src/tf/pq.rs:460-478 `pq_inverse_eotf_100k_generic`: linear_to_pq(idx * 1e-5, 10000)
    against the ST 2084 formula evaluated in f32, within 1e-6;
  * jxl-color/src/tf/pq.rs:498-515 `pq_roundtrip_10k_generic`: linear -> PQ at 10000 nits -> linear at
    1000 nits = 10 x the input, within 1e-5.
(The reference fills its inputs through `Vec::with_capacity`, so its loops run over empty vectors;
the values the code intends are used here.)  The 1-D DCT tests of vardct/generic/dct.rs are in
tests/test_oracle_dct.py."""
import ctypes as C

import numpy as np


def _lib(oracle):
    lib = oracle.lib()
    lib.orc_test_linear_to_pq.restype = C.c_float
    lib.orc_test_linear_to_pq.argtypes = [C.c_float, C.c_float]
    lib.orc_test_pq_to_linear.restype = C.c_float
    lib.orc_test_pq_to_linear.argtypes = [C.c_float, C.c_float]
    lib.orc_test_tone_map.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float]
    return lib


def test_tone_map_range(oracle):
    lib = _lib(oracle)
    lum = (C.c_float * 3)(0.2126, 0.7152, 0.0722)
    for idx in range(10):
        v = np.float32(idx // 5) * np.float32(0.1)
        rgb = (C.c_float * 3)(v, v, v)
        lib.orc_test_tone_map(rgb, lum, 10000.0, 0.0, 255.0)
        expected = np.float32(idx // 5) * np.float32(0.8714331)
        for c in range(3):
            assert abs(rgb[c] - expected) < 2e-5, (idx, c, rgb[c], expected)


def test_pq_inverse_eotf_100k(oracle):
    """The ST 2084 inverse EOTF is evaluated in f64 here: the reference's f32 `powf` form of the expected
    value is itself only good to ~1e-5 near 1.0 (exponent 78.84), while its tolerance is 1e-6 — which the
    rational polynomial does meet against the exact formula."""
    lib = _lib(oracle)
    M1, M2, C1, C2, C3 = 1305.0 / 8192.0, 2523.0 / 32.0, 107.0 / 128.0, 2413.0 / 128.0, 2392.0 / 128.0
    idx = np.arange(0, 100000, 7)
    linear = (idx.astype(np.float32) * np.float32(1e-5))
    y_m1 = linear.astype(np.float64) ** M1
    expected = ((y_m1 * C2 + C1) / (y_m1 * C3 + 1.0)) ** M2
    got = np.array([lib.orc_test_linear_to_pq(float(v), 10000.0) for v in linear], dtype=np.float64)
    assert np.abs(got - expected).max() < 1e-6


def test_pq_roundtrip_10k(oracle):
    lib = _lib(oracle)
    for idx in range(0, 10000, 3):
        v = np.float32(idx) * np.float32(1e-5)
        t = lib.orc_test_linear_to_pq(float(v), 10000.0)
        back = lib.orc_test_pq_to_linear(t, 1000.0)
        assert abs(back - np.float32(idx) * np.float32(1e-4)) < 1e-5, idx
