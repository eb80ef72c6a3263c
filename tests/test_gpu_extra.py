"""GPU parity of the extra-channel path (jxlgpu_frame_render_extra, RGBA formatting) against the oracle: bit-exact
(the conversion is one division or a bit shuffle; the upsampling kernels are the colour path's, 0 ULP vs the oracle)."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload, make_extra_channel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame(gpu_ctx):
    wl = VardctWorkload(96, 64, seed=31)
    f = gpu_ctx.vardct_upload(wl.desc())
    gpu_ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
    yield f
    f.free()


@pytest.mark.parametrize("case", [
    dict(w=96, h=64, i16=True, bit_depth=8),
    dict(w=96, h=64, i16=False, bit_depth=16),
    dict(w=96, h=64, i16=True, bit_depth=16, float_sample=True, exp_bits=5),
    dict(w=96, h=64, i16=False, bit_depth=32, float_sample=True, exp_bits=8),
    dict(w=48, h=32, i16=True, bit_depth=8, upsampling_log2=1),
    dict(w=24, h=16, i16=True, bit_depth=12, upsampling_log2=2),
    dict(w=12, h=8, i16=False, bit_depth=10, upsampling_log2=3),
    dict(w=7, h=5, i16=True, bit_depth=8, upsampling_log2=4),
    dict(w=3, h=2, i16=True, bit_depth=8, upsampling_log2=5),
    dict(w=33, h=17, i16=True, bit_depth=1, upsampling_log2=1),
])
def test_extra_channel_matches_the_oracle(gpu_ctx, oracle, frame, case):
    c = dict(case)
    ec, keep = make_extra_channel(c.pop("w"), c.pop("h"), seed=7, **c)
    exp = oracle.extra_channel(ec)
    got = gpu_ctx.render_extra(frame, 1, ec)
    assert got.shape == exp.shape
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("fmt", [abi.FMT_U8, abi.FMT_U16, abi.FMT_F32])
@pytest.mark.parametrize("orientation", [1, 2, 5, 7])
def test_rgba_output_matches_the_oracle(gpu_ctx, oracle, frame, fmt, orientation):
    wl = VardctWorkload(96, 64, seed=31)
    rgb, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, 96, 64)
    ec_a, keep_a = make_extra_channel(48, 32, seed=11, bit_depth=8, upsampling_log2=1)   # alpha at half resolution
    ec_d, keep_d = make_extra_channel(96, 64, seed=12, i16=False, bit_depth=16)          # a 16-bit depth channel
    gpu_ctx.render_extra(frame, 0, ec_a, to_host=False)
    gpu_ctx.render_extra(frame, 3, ec_d, to_host=False)
    got = gpu_ctx.format_output(frame, fmt, orientation, extra=(0, 3))
    exp = oracle.format_output_n([rgb[0], rgb[1], rgb[2], oracle.extra_channel(ec_a), oracle.extra_channel(ec_d)], fmt, orientation)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    # colour only is unchanged by the presence of extra planes
    assert np.array_equal(gpu_ctx.format_output(frame, fmt, orientation), oracle.format_output(rgb, fmt, orientation))


def test_bad_extra_requests_are_refused(gpu_ctx, frame):
    from jxl_oxide_amd.runtime import JxlGpuError
    ec, keep = make_extra_channel(48, 32, seed=1, upsampling_log2=0)
    gpu_ctx.render_extra(frame, 2, ec, to_host=False)   # 48 x 32: not the size of the 96 x 64 colour result
    with pytest.raises(JxlGpuError):
        gpu_ctx.format_output(frame, abi.FMT_U8, 1, extra=(2,))
    with pytest.raises(JxlGpuError):
        gpu_ctx.format_output(frame, abi.FMT_U8, 1, extra=(6,))   # never rendered
    ec.upsampling_log2 = 7
    with pytest.raises(JxlGpuError):
        gpu_ctx.render_extra(frame, 2, ec, to_host=False)
