"""Output formatting oracle (ImageStream semantics) against numpy flips / transposes."""
import numpy as np
import pytest

from jxl_oxide_amd import abi


def _expected_orientation(img, o):
    # img: (h, w, 3).  EXIF orientation -> displayed image
    if o == 1: return img
    if o == 2: return img[:, ::-1]
    if o == 3: return img[::-1, ::-1]
    if o == 4: return img[::-1, :]
    if o == 5: return img.transpose(1, 0, 2)
    if o == 6: return img[::-1, :].transpose(1, 0, 2)
    if o == 7: return img[::-1, ::-1].transpose(1, 0, 2)
    return img[:, ::-1].transpose(1, 0, 2)


@pytest.mark.parametrize("o", range(1, 9))
def test_orientation_and_interleave(oracle, o):
    rng = np.random.default_rng(o)
    planes = rng.uniform(-0.2, 1.2, size=(3, 5, 7)).astype(np.float32)
    got = oracle.format_output(planes, abi.FMT_F32, o)
    exp = _expected_orientation(planes.transpose(1, 2, 0), o)
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_integer_conversion(oracle):
    v = np.array([-1.0, 0.0, 0.001, 0.5, 1.0, 1.7, 0.49999 / 255, 0.5 / 255, np.nan], dtype=np.float32)
    planes = np.stack([v.reshape(1, -1)] * 3)
    u8 = oracle.format_output(planes, abi.FMT_U8, 1)[0, :, 0]
    u16 = oracle.format_output(planes, abi.FMT_U16, 1)[0, :, 0]
    exp8 = [0, 0, 0, 128, 255, 255, 0, 1, 0]
    assert list(u8) == exp8
    t = np.clip(np.nan_to_num(v.astype(np.float64), nan=-1.0) * 65535.0 + 0.5, 0, 65535)
    assert list(u16) == [int(x) for x in t]
