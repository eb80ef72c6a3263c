"""Oracle pins for the extra-channel path (oracle/extra.c): BitDepth::parse_integer_sample's bit layout against numpy's
own IEEE decoders, the integer case against exact rational arithmetic, the upsampling chain against the (separately
tested) single-pass oracle upsampler, and the n-channel formatter against the 3-channel one."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import make_extra_channel


def test_integer_samples_are_value_over_max(oracle):
    for bits, i16 in ((8, True), (12, True), (16, False), (1, True), (24, False)):
        ec, keep = make_extra_channel(37, 19, seed=bits, i16=i16, bit_depth=bits)
        got = oracle.extra_channel(ec)
        div = np.float32((1 << bits) - 1)
        assert np.array_equal(got, keep[0].astype(np.float32) / div)   # one correctly rounded f32 division, as lib.rs:461-462


def test_float_samples_decode_like_ieee(oracle):
    # binary16 (bits 16, exp 5) and binary32 (bits 32, exp 8): the formats numpy can decode itself; normal numbers only
    ec, keep = make_extra_channel(41, 23, seed=1, i16=True, bit_depth=16, float_sample=True, exp_bits=5)
    got = oracle.extra_channel(ec)
    want = keep[0].view(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ec, keep = make_extra_channel(41, 23, seed=2, i16=False, bit_depth=32, float_sample=True, exp_bits=8)
    got = oracle.extra_channel(ec)
    assert np.array_equal(got.view(np.uint32), keep[0].view(np.uint32))
    # a 24-bit float (1 + 7 + 16): exponent bias 63, mantissa widened to 23 bits
    ec, keep = make_extra_channel(16, 9, seed=3, i16=False, bit_depth=24, float_sample=True, exp_bits=7)
    got = oracle.extra_channel(ec)
    p = keep[0].astype(np.int64)
    val = (1.0 + (p & 0xffff) / 65536.0) * np.exp2(((p >> 16) & 0x7f) - 63.0) * np.where(p >> 23 & 1, -1.0, 1.0)
    assert np.array_equal(got, val.astype(np.float32))


@pytest.mark.parametrize("log2", [1, 2, 3, 4, 5])
def test_upsampling_chain_is_8x_passes_then_the_remainder(oracle, log2):
    import ctypes as C
    ec, keep = make_extra_channel(13, 7, seed=log2, bit_depth=10, upsampling_log2=log2)
    got = oracle.extra_channel(ec)
    cur = keep[0].astype(np.float32) / np.float32(1023)
    up = keep[1]
    f = oracle.lib().orc_upsample_inner
    f.argtypes = [abi.f32p, C.c_size_t, C.c_size_t, C.c_size_t, abi.f32p, C.c_size_t, C.c_int, abi.f32p]
    f.restype = None
    for k in [8] * (log2 // 3) + ([] if log2 % 3 == 0 else [2 if log2 % 3 == 1 else 4]):
        h, w = cur.shape
        nxt = np.zeros((h * k, w * k), dtype=np.float32)
        wts = np.ascontiguousarray(up[{2: 0, 4: 1, 8: 2}[k]], dtype=np.float32)
        f(cur.ctypes.data_as(abi.f32p), w, w, h, nxt.ctypes.data_as(abi.f32p), w * k, k, wts.ctypes.data_as(abi.f32p))
        cur = nxt
    assert got.shape == (7 << log2, 13 << log2)
    assert np.array_equal(got.view(np.uint32), cur.view(np.uint32))


@pytest.mark.parametrize("orientation", [1, 3, 6, 8])
@pytest.mark.parametrize("fmt", [abi.FMT_U8, abi.FMT_U16, abi.FMT_F32])
def test_n_channel_formatter_extends_the_three_channel_one(oracle, fmt, orientation):
    rng = np.random.default_rng(5)
    planes = (rng.random(size=(4, 21, 34)) * 1.2 - 0.1).astype(np.float32)
    rgb = oracle.format_output(planes[:3], fmt, orientation)
    rgba = oracle.format_output_n(list(planes), fmt, orientation)
    assert np.array_equal(rgba[..., :3], rgb)
    a = oracle.format_output(np.stack([planes[3]] * 3), fmt, orientation)
    assert np.array_equal(rgba[..., 3], a[..., 0])
