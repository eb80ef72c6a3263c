"""GPU parity for the JPEG-transcode flavour of the VarDCT path: YCbCr frames,
chroma subsampling in every arrangement, chroma upsampling, optional restoration filters."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import JpegWorkload
from util import assert_ulp

pytestmark = pytest.mark.gpu
S_ALL = abi.STAGE_ALL


def _both(gpu_ctx, oracle, wl, stages):
    d = wl.desc()
    exp, _ = oracle.vardct_render(d, stages, wl.width, wl.height)
    frame = gpu_ctx.vardct_upload(d)
    try:
        got = gpu_ctx.vardct_render(frame, stages)
        again = gpu_ctx.vardct_render(frame, stages)
    finally:
        frame.free()
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32))
    return got, exp


@pytest.mark.parametrize("mode", ["444", "420", "422", "440", "mixed"])
@pytest.mark.parametrize("size", [(72, 40), (83, 45), (300, 270), (2100, 24)])
def test_jpeg_frames(gpu_ctx, oracle, mode, size):
    w, h = size
    wl = JpegWorkload(w, h, mode=mode, seed=w)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, 1, f"jpeg {mode} {w}x{h}")
    got, exp = _both(gpu_ctx, oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM)
    assert_ulp(got, exp, 1, f"jpeg {mode} {w}x{h} (YCbCr planes after chroma upsampling)")


@pytest.mark.parametrize("mode", ["420", "422"])
def test_jpeg_frames_with_restoration_filters(gpu_ctx, oracle, mode):
    wl = JpegWorkload(264, 200, mode=mode, seed=9, epf_iters=2, gabor=True, lf_i16=False)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, 1, f"jpeg {mode} + gabor + epf")


def test_jpeg_multi_lf_group(gpu_ctx, oracle):
    """2100 px wide: two LF groups, the second one narrow; extra_precision differs per LF group."""
    wl = JpegWorkload(2100, 300, mode="420", seed=2)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, 1, "jpeg 420 2100x300")


def test_subsampled_frame_with_other_varblocks_is_refused(gpu_ctx):
    wl = JpegWorkload(64, 64, mode="420", seed=1)
    wl.kind[0, 0] = 4  # DCT16
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(wl.desc())
    assert e.value.code == abi.ERR_UNSUPPORTED
