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
