import numpy as np


def ulp_diff(a, b):
    """Max distance in units-in-the-last-place between two f32 arrays (0 == bit-identical,
    treating +0/-0 as equal)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return int(np.max(np.abs(ai - bi))) if a.size else 0


def assert_same_bits_or_nan(got, exp, what="", min_finite=0.5):
    """Bit-identical wherever the oracle is not NaN, NaN exactly where it is NaN (payloads are not compared: x86 and gfx950
    differ in the default NaN's sign).  For the HLG op lists, where a negative luminance mix has no real power
    (tf.rs:118-143) and the reference itself writes NaN.  At least `min_finite` of the samples must be finite."""
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    en, gn = np.isnan(exp), np.isnan(got)
    assert (~en).mean() >= min_finite, f"{what}: only {(~en).mean():.3f} of the oracle's samples are numbers"
    if not np.array_equal(en, gn):
        bad = np.argwhere(en != gn)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: NaN positions differ at {len(bad)} samples, first at {i}: got {got[i]!r} exp {exp[i]!r}")
    g = np.where(en, np.float32(0), got).view(np.uint32)
    e = np.where(en, np.float32(0), exp).view(np.uint32)
    if not np.array_equal(g, e):
        bad = np.argwhere(g != e)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} samples differ (max ULP {ulp_diff(np.where(en, 0, got), np.where(en, 0, exp))}), "
                             f"first at {i}: got {got[i]!r} exp {exp[i]!r}")
    return float(en.mean())


def assert_ulp(got, exp, max_ulp, what=""):
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    assert np.isfinite(exp).all(), f"{what}: oracle produced non-finite values"
    d = ulp_diff(got, exp)
    if d > max_ulp:
        bad = np.argwhere(got != exp)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: max ULP diff {d} > {max_ulp}; {len(bad)} mismatches, "
                             f"first at {i}: got {got[i]!r} exp {exp[i]!r}")
    return d
