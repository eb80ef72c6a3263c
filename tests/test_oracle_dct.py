"""Pins the oracle's 1-D DCT against the reference's own unit tests
(jxl-render/src/vardct/generic/dct.rs:299-435): same inputs, expected value = f64 direct cosine
sum, comparison after `(x * 65536) as i32` — reproduced verbatim, then extended to N = 16..256
and to 2-D with a tolerance (the reference has no vectors there)."""
import math

import numpy as np
import pytest


def _expected_forward(original):
    s = len(original)
    out = []
    for k in range(s):
        v = 0.0
        for n, x in enumerate(original):
            v += float(np.float32(x)) * math.cos((k * (2 * n + 1)) / s * (math.pi / 2))
        v /= s
        if k != 0:
            v *= math.sqrt(2.0)
        out.append(v)
    return out


def _expected_inverse(original):
    s = len(original)
    out = []
    for k in range(s):
        v = float(np.float32(original[0]))
        for n in range(1, s):
            v += float(np.float32(original[n])) * math.cos((n * (2 * k + 1)) / s * (math.pi / 2)) * math.sqrt(2.0)
        out.append(v)
    return out


def _q(v):
    return int(v * 65536.0)  # Rust `as i32` truncates toward zero, like int()


REF_FORWARD = [[-1.0, 3.0], [-1.0, 2.0, 3.0, -4.0], [1.0, 0.3, 1.0, 2.0, -2.0, -0.1, 1.0, 0.1]]
REF_INVERSE = [[3.0, 0.2], [3.0, 0.2, 0.3, -1.0], [3.0, 0.0, 0.0, -1.0, 0.0, 0.3, 0.2, 0.0]]


@pytest.mark.parametrize("original", REF_FORWARD)
def test_reference_forward_vectors(oracle, original):
    got = oracle.dct_1d(original, True)
    exp = _expected_forward(original)
    assert [_q(float(g)) for g in got] == [_q(e) for e in exp]


@pytest.mark.parametrize("original", REF_INVERSE)
def test_reference_inverse_vectors(oracle, original):
    got = oracle.dct_1d(original, False)
    exp = _expected_inverse(original)
    assert [_q(float(g)) for g in got] == [_q(e) for e in exp]


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256])
def test_large_n_vs_f64(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, size=n).astype(np.float32)
    fwd = oracle.dct_1d(x, True)
    inv = oracle.dct_1d(x, False)
    assert np.allclose(fwd, _expected_forward(x), atol=2e-6 * math.log2(n))
    assert np.allclose(inv, _expected_inverse(x), atol=4e-5 * math.sqrt(n))
    # inverse(forward(x)) == x
    assert np.allclose(oracle.dct_1d(fwd, False), x, atol=1e-5)


@pytest.mark.parametrize("shape", [(1, 2), (2, 1), (2, 2), (1, 4), (4, 1), (4, 2), (2, 4), (4, 4),
                                   (4, 8), (8, 4), (8, 8), (16, 8), (8, 16), (32, 32), (16, 32), (64, 64)])
def test_2d_separable_vs_f64(oracle, shape):
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    x = rng.uniform(-1, 1, size=(h, w)).astype(np.float32)

    def mat(n, forward):
        m = np.zeros((n, n))
        for k in range(n):
            for i in range(n):
                if forward:
                    m[k, i] = math.cos(k * (2 * i + 1) / n * math.pi / 2) / n * (math.sqrt(2) if k else 1)
                else:
                    m[k, i] = math.cos(i * (2 * k + 1) / n * math.pi / 2) * (math.sqrt(2) if i else 1)
        return m
    for forward in (True, False):
        exp = mat(h, forward) @ x.astype(np.float64) @ mat(w, forward).T
        got = oracle.dct_2d(x, forward)
        assert np.allclose(got, exp, atol=3e-5 * math.sqrt(h * w)), (shape, forward)
