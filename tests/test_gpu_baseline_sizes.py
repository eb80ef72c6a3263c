"""Parity at BASELINE.json's full sizes (configs 2, 3 and 5): the frames bench.py times are
compared with the oracle here, sample for sample.  Geometry that only shows up at these sizes:
2 x 2 LF groups (2048 px) with per-group LF scales, 15 x 9 groups of 256 px with a 112-px last
group row, the EPF sigma lookup across LF groups (filter/epf.rs:174-201), 22 Squeeze steps."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload
from jxl_oxide_amd.synth_modular import ModularWorkload
from util import assert_ulp

pytestmark = pytest.mark.gpu
MAX_ULP = 1  # north_star tolerance for float VarDCT; integer Modular results must be bit-exact


def _render_both(gpu_ctx, oracle, wl, stages):
    d = wl.desc()
    ow, oh = wl.out_size(stages)
    exp, _ = oracle.vardct_render(d, stages, ow, oh)
    frame = gpu_ctx.vardct_upload(d)
    try:
        got = gpu_ctx.vardct_render(frame, stages)
    finally:
        frame.free()
    return got, exp


def test_config2_4k_vardct_full_pipeline(gpu_ctx, oracle):
    """3840x2160 VarDCT d1, mixed transform types, Gabor + EPF iters 2, XYB -> sRGB (the headline)."""
    wl = VardctWorkload(3840, 2160, seed=2000)  # bench.py's rank-0 frame
    for t in (0, 4, 5, 18):  # Dct8, Dct16, Dct32, Dct64 all present
        assert (wl.kind == t).any()
    got, exp = _render_both(gpu_ctx, oracle, wl, abi.STAGE_ALL)
    assert_ulp(got, exp, MAX_ULP, "config 2: 4K VarDCT, all stages")
    got, exp = _render_both(gpu_ctx, oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM)
    assert_ulp(got, exp, MAX_ULP, "config 2: 4K VarDCT, V1-V8 only")


def test_config2_sparse_i16_transport_matches_dense(gpu_ctx):
    """The compact coefficient transport bench.py's end-to-end figure uses renders the same frame."""
    wl = VardctWorkload(3840, 2160, seed=2001)
    outs = []
    for tr in ("dense_i32", "sparse_i16"):
        f = gpu_ctx.vardct_upload(wl.desc(coeff_transport=tr))
        try:
            outs.append(gpu_ctx.vardct_render(f, abi.STAGE_ALL))
        finally:
            f.free()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_config5_hdr_upsampled(gpu_ctx, oracle):
    """Coded 3840x2160 VarDCT, EPF iters 3, 2x upsampling to 7680x4320, intensity target 4000,
    Rec.2100 PQ (gamut map + second matrix + PQ transfer)."""
    wl = VardctWorkload(3840, 2160, seed=5000, epf_iters=3, upsampling=2, intensity_target=4000.0, hdr_pq=True)
    got, exp = _render_both(gpu_ctx, oracle, wl, abi.STAGE_ALL)
    assert got.shape == (3, 4320, 7680)
    assert_ulp(got, exp, MAX_ULP, "config 5: 4K coded -> 8K PQ")


@pytest.fixture(scope="module")
def modular_8k():
    # BASELINE config 3 as worded: lossy Squeeze AND self-correcting-predictor residuals (single-leaf tree) on
    # every carved sub-channel
    return ModularWorkload(7680, 4320, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=3, residual=6)


def test_config3_8k_squeeze_inverse_bit_exact(gpu_ctx, oracle, modular_8k):
    """7680x4320 Modular, lossy Squeeze (default 22-step schedule) + self-correcting predictor residuals on
    the 67 carved sub-channels, 16-bit buffers: the integer reconstruction must be bit-identical."""
    wl = modular_8k
    d = wl.desc()
    exp = oracle.modular_inverse(d, wl.shapes(), wl.dtype)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_inverse(f, wl.shapes(), wl.dtype)
    finally:
        f.free()
    for c in range(3):
        assert np.array_equal(got[c], exp[c]), f"channel {c}"


def test_config3_8k_modular_render(gpu_ctx, oracle, modular_8k):
    """... and through XYB dequantisation (M5), EPF (sigma_for_modular) and XYB -> sRGB."""
    wl = modular_8k
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, wl.width, wl.height)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert_ulp(got, exp, MAX_ULP, "config 3: 8K Modular render")


def test_config3_8k_weighted_predictor_i32(gpu_ctx, oracle):
    """The self-correcting (weighted) predictor of config 3 at 8K, single-leaf tree, 32-bit buffers:
    30 x 17 group tiles of 256 px, bit-exact against the oracle's PredictorState restatement."""
    wl = ModularWorkload(7680, 4320, kind="predictor_random", predictor=6, i16=False, seed=4)
    d = wl.desc()
    exp = oracle.modular_inverse(d, wl.shapes(), wl.dtype)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_inverse(f, wl.shapes(), wl.dtype)
    finally:
        f.free()
    for c in range(3):
        assert np.array_equal(got[c], exp[c]), f"channel {c}"
