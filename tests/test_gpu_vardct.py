"""GPU parity: libjxlgpu.so (through the C ABI) vs the CPU oracle on the same seeded synthetic
frames.  North-star tolerance for float VarDCT is 1 ULP; the kernels are built to be bit-exact
against the oracle, so the asserted bound is MAX_ULP = 1 and the observed value is normally 0."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload
from util import assert_ulp

pytestmark = pytest.mark.gpu
MAX_ULP = 1  # tolerance stated by BASELINE.json's north_star for float VarDCT

S_LF = abi.STAGE_LF
S_TR = S_LF | abi.STAGE_TRANSFORM
S_GAB = S_TR | abi.STAGE_GABOR
S_EPF = S_GAB | abi.STAGE_EPF
S_ALL = abi.STAGE_ALL


def _both(gpu_ctx, oracle, wl, stages):
    d = wl.desc()
    ow, oh = wl.out_size(stages)
    exp, _ = oracle.vardct_render(d, stages, ow, oh)
    frame = gpu_ctx.vardct_upload(d)
    try:
        got = gpu_ctx.vardct_render(frame, stages)
    finally:
        frame.free()
    return got, exp


def test_lf_stage(gpu_ctx, oracle):
    wl = VardctWorkload(520, 264, seed=1)
    d = wl.desc()
    _, lf_exp = oracle.vardct_render(d, S_LF, wl.width, wl.height, want_lf=True, w8=wl.w8, h8=wl.h8)
    frame = gpu_ctx.vardct_upload(d)
    try:
        gpu_ctx.vardct_render(frame, S_LF, to_host=False)
        lf = gpu_ctx.download_lf(frame, wl.w8, wl.h8)
    finally:
        frame.free()
    assert_ulp(lf, lf_exp, 0, "LF image (V1-V3)")


@pytest.mark.parametrize("t", list(range(27)))
def test_each_transform_type(gpu_ctx, oracle, t):
    bw, bh = abi.DCT_SELECT_SIZE[t]
    w = max(64, bw * 8 * 2 + 8)
    h = max(64, bh * 8 * 2 + 8)
    w, h = min(w, 256 + 64), min(h, 256 + 64)
    wl = VardctWorkload(w, h, seed=100 + t, types=[t], zero_fraction=0.5)
    assert (wl.kind == t).any(), "generator placed no block of this type"
    got, exp = _both(gpu_ctx, oracle, wl, S_TR)
    assert_ulp(got, exp, MAX_ULP, f"transform {abi.TRANSFORM_NAMES[t]}")


@pytest.mark.parametrize("size", [(8, 8), (9, 7), (24, 40), (255, 257), (300, 520)])
def test_ragged_sizes_all_stages(gpu_ctx, oracle, size):
    w, h = size
    wl = VardctWorkload(w, h, seed=7 + w)
    for stages, name in ((S_TR, "transform"), (S_GAB, "gabor"), (S_EPF, "epf"), (S_ALL, "colour")):
        got, exp = _both(gpu_ctx, oracle, wl, stages)
        assert_ulp(got, exp, MAX_ULP, f"{name} {w}x{h}")


@pytest.mark.parametrize("iters", [0, 1, 2, 3])
@pytest.mark.parametrize("gab", [False, True])
def test_filter_configs(gpu_ctx, oracle, iters, gab):
    wl = VardctWorkload(264, 200, seed=40 + iters, epf_iters=iters, gabor=gab)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, MAX_ULP, f"iters={iters} gab={gab}")


def test_staged_post_path_matches(gpu_ctx, oracle, monkeypatch):
    """The stage-at-a-time kernels (fallback path) must agree with the oracle too.  The switches
    are read once per context, at jxlgpu_create, so the test opens its own context."""
    from jxl_oxide_amd import runtime
    monkeypatch.setenv("JXLGPU_NO_FUSED", "1")
    ctx = runtime.Context(0)
    try:
        for iters in (1, 2, 3):
            wl = VardctWorkload(200, 136, seed=70 + iters, epf_iters=iters)
            got, exp = _both(ctx, oracle, wl, S_ALL)
            assert_ulp(got, exp, MAX_ULP, f"staged iters={iters}")
    finally:
        ctx.close()


def test_tile_kernel_whole_frame_matches(gpu_ctx, oracle, monkeypatch):
    """JXLGPU_NO_STREAM: the LDS tile kernel (normally only the border ring) over the whole frame."""
    from jxl_oxide_amd import runtime
    monkeypatch.setenv("JXLGPU_NO_STREAM", "1")
    ctx = runtime.Context(0)
    try:
        wl = VardctWorkload(264, 200, seed=75)
        got, exp = _both(ctx, oracle, wl, S_ALL)
        assert_ulp(got, exp, MAX_ULP, "tile kernel, whole frame")
    finally:
        ctx.close()


@pytest.mark.parametrize("size", [(520, 300), (1000, 264), (301, 299)])
def test_packed_and_scalar_streaming_kernels_match(gpu_ctx, oracle, monkeypatch, size):
    """The default post stage is the packed kernel (two columns per lane, post_pk.inc); JXLGPU_NO_PK
    selects the scalar one.  Both must give the oracle's bits (several strips, a partial last strip,
    an odd width)."""
    from jxl_oxide_amd import runtime
    wl = VardctWorkload(size[0], size[1], seed=76)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, MAX_ULP, f"packed kernel {size}")
    monkeypatch.setenv("JXLGPU_NO_PK", "1")
    ctx = runtime.Context(0)
    try:
        got2, _ = _both(ctx, oracle, wl, S_ALL)
        assert_ulp(got2, exp, MAX_ULP, f"scalar kernel {size}")
    finally:
        ctx.close()


@pytest.mark.parametrize("case", ["zero_x", "zero_all", "huge", "tiny"])
@pytest.mark.parametrize("stages", [S_EPF, S_ALL])
def test_packed_kernel_division_guard(gpu_ctx, oracle, case, stages):
    """The packed kernel replaces sum_c / sum_w by a shared-reciprocal fma chain where every numerator
    lies in [2^-100, 2^20]; anything else (zeros of either sign, huge or tiny samples) must take the
    ordinary divisions.  S_EPF (no colour) keeps the sign of a zero visible in the output."""
    wl = VardctWorkload(328, 232, seed=77, lf_i16=False)
    if case in ("zero_x", "zero_all"):
        chans = (0,) if case == "zero_x" else (0, 1, 2)
        for c in chans:
            wl.coeff[c, 64:200, 96:300] = 0
        wl.xfy[:] = 0
        wl.bfy[:] = 0
        # lfq order is Y, X, B
        for i in ((1,) if case == "zero_x" else (0, 1, 2)):
            wl.lfq[i][8:25, 12:38] = 0
    elif case == "huge":
        wl.lfq[0][10:20, 14:30] = 2 ** 30
        wl.lfq[2][12:22, 10:36] = -(2 ** 30)
    else:
        # samples far below 2^-100 after the dequantisation: all-zero LF and HF except single +-1 coefficients
        # scaled down through the smallest multipliers the descriptor allows is not reachable; the
        # closest the format gets is exact zeros next to ordinary samples
        wl.coeff[:, 100:164, 100:228] = 0
        for i in range(3):
            wl.lfq[i][12:21, 12:29] = 0
        wl.coeff[1, 120, 130] = 1
    d = wl.desc()
    ow, oh = wl.out_size(stages)
    exp, _ = oracle.vardct_render(d, stages, ow, oh)
    frame = gpu_ctx.vardct_upload(d)
    try:
        got = gpu_ctx.vardct_render(frame, stages)
    finally:
        frame.free()
    assert got.shape == exp.shape
    if case == "huge":
        assert np.isfinite(exp).all()
    # bit patterns, so that -0.0 vs +0.0 counts
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (
        case, int((got.view(np.uint32) != exp.view(np.uint32)).sum()))


@pytest.mark.parametrize("size", [(1, 1), (2, 3), (5, 4), (3, 40), (33, 2)])
def test_tiny_images(gpu_ctx, oracle, size):
    w, h = size
    for iters in (0, 2, 3):
        wl = VardctWorkload(w, h, seed=80 + w + iters, epf_iters=iters)
        got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
        assert_ulp(got, exp, MAX_ULP, f"tiny {w}x{h} iters={iters}")


def test_mixed_frame_multi_lf_group(gpu_ctx, oracle):
    # > 2048 px wide: two LF groups with different extra_precision
    wl = VardctWorkload(2100, 300, seed=9, lf_i16=False)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, MAX_ULP, "2100x300 full pipeline")


def test_hdr_pq_chain(gpu_ctx, oracle):
    wl = VardctWorkload(200, 136, seed=11, epf_iters=3, intensity_target=4000.0, hdr_pq=True)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, MAX_ULP, "PQ chain")


@pytest.mark.parametrize("mode,it", [("tone_map_srgb", 4000.0), ("tone_map_min_nits", 1000.0), ("bt709", 255.0),
                                     ("clip_p3_dci", 255.0), ("gamma22", 255.0)])
def test_colour_op_lists(gpu_ctx, oracle, mode, it):
    """C4: Rec.2408 tone map (+ GamutMap), Clip, BT.709 and gamma transfer functions, through both
    the fused post path and the staged colour kernel."""
    wl = VardctWorkload(200, 136, seed=31, epf_iters=1, intensity_target=it, color_mode=mode)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, MAX_ULP, f"colour {mode} (fused)")
    d = wl.desc()
    exp2, _ = oracle.vardct_render(d, abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_COLOR, wl.width, wl.height)
    frame = gpu_ctx.vardct_upload(d)
    try:
        got2 = gpu_ctx.vardct_render(frame, abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_COLOR)
    finally:
        frame.free()
    assert_ulp(got2, exp2, MAX_ULP, f"colour {mode} (colour-only stage)")
    # default filter configuration at a size where the streaming kernel covers the interior
    wl = VardctWorkload(264, 200, seed=32, intensity_target=it, color_mode=mode)
    got3, exp3 = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got3, exp3, MAX_ULP, f"colour {mode} (streaming kernel)")


@pytest.mark.parametrize("w,h,up,epf", [(520, 300, 1, 2), (264, 200, 1, 0), (72, 56, 2, 1), (40, 24, 8, 1), (257, 3, 1, 1)])
def test_noise_synthesis(gpu_ctx, oracle, w, h, up, epf):
    """xorshift128+ noise with GF(2) jump-ahead, ring-ordered 5x5 kernel, LUT
    modulation; between upsampling and the colour transform."""
    wl = VardctWorkload(w, h, seed=50 + w, epf_iters=epf, gabor=epf > 0, upsampling=up, noise=True)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_ulp(got, exp, MAX_ULP, f"noise {w}x{h} up{up}")
    # and it is not a no-op
    off, _ = oracle.vardct_render(wl.desc(), S_ALL & ~abi.STAGE_NOISE, *wl.out_size(S_ALL))
    assert np.abs(off - exp).max() > 1e-3
    # stopping before the colour transform exposes the noisy XYB planes
    got2, exp2 = _both(gpu_ctx, oracle, wl, S_ALL & ~abi.STAGE_COLOR)
    assert_ulp(got2, exp2, MAX_ULP, f"noise {w}x{h} up{up} (XYB)")


def test_noise_where_the_reference_panics_is_refused(gpu_ctx):
    wl = VardctWorkload(40, 257, seed=5, noise=True)
    frame = gpu_ctx.vardct_upload(wl.desc())
    try:
        with pytest.raises(Exception) as e:
            gpu_ctx.vardct_render(frame, S_ALL)
        assert e.value.code == abi.ERR_UNSUPPORTED
    finally:
        frame.free()


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_upsampling(gpu_ctx, oracle, factor):
    wl = VardctWorkload(72, 56, seed=20 + factor, upsampling=factor, epf_iters=1)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert got.shape == (3, 56 * factor, 72 * factor)
    assert_ulp(got, exp, MAX_ULP, f"upsampling x{factor}")


@pytest.mark.parametrize("transport,split", [("dense_i16", False), ("sparse_i32", False), ("sparse_i16", False),
                                             ("sparse_i32", True)])
def test_compact_coefficient_transport(gpu_ctx, oracle, transport, split):
    """16-bit planes / (position, value) lists rebuild the reference's dense i32
    framebuffer on the device; the oracle always sees the dense form."""
    wl = VardctWorkload(520, 264, seed=41)
    exp, _ = oracle.vardct_render(wl.desc(), S_ALL, wl.width, wl.height)
    frame = gpu_ctx.vardct_upload(wl.desc(coeff_transport=transport, sparse_split=split))
    try:
        got = gpu_ctx.vardct_render(frame, S_ALL)
    finally:
        frame.free()
    assert_ulp(got, exp, MAX_ULP, f"{transport} split={split}")


def test_sparse_position_outside_frame_is_rejected(gpu_ctx):
    import ctypes as C
    wl = VardctWorkload(64, 64, seed=5)
    d = wl.desc(coeff_transport="sparse_i32")
    pos = np.array([64 * 64 + 3], dtype=np.uint32)
    val = np.array([7], dtype=np.int32)
    d.coeff[1] = val.ctypes.data
    d.sparse_pos[1] = pos.ctypes.data_as(C.POINTER(C.c_uint32))
    d.sparse_count[1] = 1
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG


def test_render_host_one_shot(gpu_ctx, oracle):
    wl = VardctWorkload(136, 120, seed=3)
    d = wl.desc()
    exp, _ = oracle.vardct_render(d, S_ALL, wl.width, wl.height)
    got = gpu_ctx.vardct_render_host(d, S_ALL, wl.width, wl.height)
    assert_ulp(got, exp, MAX_ULP, "render_host")


def test_error_codes(gpu_ctx):
    wl = VardctWorkload(64, 64, seed=5)
    d = wl.desc()
    d.abi = 999
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_ABI
    d = wl.desc()
    d.jpeg_upsampling[1] = 1
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_UNSUPPORTED
    d = wl.desc()
    d.coeff[0] = None
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG
    # an LF-only render has no pixel output to copy
    frame = gpu_ctx.vardct_upload(wl.desc())
    try:
        with pytest.raises(Exception) as e:
            gpu_ctx.vardct_render(frame, abi.STAGE_LF)
        assert e.value.code == abi.ERR_INVALID_ARG
    finally:
        frame.free()
    # per-row kernels use one grid row per image row: outputs taller than 65535 rows are refused at upload
    d = wl.desc()
    d.height = 70000
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(d)
    assert e.value.code == abi.ERR_UNSUPPORTED


@pytest.mark.parametrize("fmt", [abi.FMT_F32, abi.FMT_U16, abi.FMT_U8])
def test_device_output_formatting(gpu_ctx, oracle, fmt):
    """interleave + orientation + u8/u16 conversion on the device."""
    for size in ((136, 72), (134, 70)):  # a multiple of 4 (four-pixel u8 kernel) and not
        wl = VardctWorkload(size[0], size[1], seed=21)
        d = wl.desc()
        planes, _ = oracle.vardct_render(d, S_ALL, wl.width, wl.height)
        frame = gpu_ctx.vardct_upload(d)
        try:
            gpu_ctx.vardct_render(frame, S_ALL, to_host=False)
            for o in range(1, 9):
                got = gpu_ctx.format_output(frame, fmt, o)
                exp = oracle.format_output(planes, fmt, o)
                assert got.shape == exp.shape
                assert np.array_equal(got.view(np.uint8), exp.view(np.uint8)), f"{size} fmt {fmt} orientation {o}"
        finally:
            frame.free()


def test_all_zero_channel_keeps_the_packed_division_path_exact(gpu_ctx, oracle):
    """A gray image in XYB has an X plane of exact zeros of either sign (zero coefficients through the IDCT's
    negative constants give -0).  The packed EPF kernel's shared-reciprocal quotient must return n / d bit for bit
    for n = +-0 as well (csrc/post_pk.inc div3_shared: the residual is formed as -(d q - n)): EPF output planes
    (no colour transform, so the sign of a zero is visible) against the oracle."""
    wl = VardctWorkload(520, 300, seed=77)
    wl.coeff[0] = 0                      # no X coefficients
    wl.lfq[1][...] = 0                   # X of the LF image
    wl.xfy[...] = 0                      # no chroma-from-luma into X (base_correlation_x is 0)
    stages = abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_GABOR | abi.STAGE_EPF
    d = wl.desc()
    d.x_factor_lf = 128                  # CfL-LF factor 0 for X: ((128 - 128) / 128)
    exp, _ = oracle.vardct_render(d, stages, wl.width, wl.height)
    assert not exp[0].any(), "the X plane is expected to be all zeros"
    d2 = wl.desc(coeff_transport="grouped")
    d2.x_factor_lf = 128
    f = gpu_ctx.vardct_upload(d2)
    try:
        got = gpu_ctx.vardct_render(f, stages)
        gpu_ctx.vardct_render_batch([f], stages)
        gpu_ctx.synchronize()
        got_b = gpu_ctx.download_result(f, stages)
    finally:
        f.free()
    for g, what in ((got, "single frame"), (got_b, "batch")):
        assert np.array_equal(g.view(np.uint32), exp.view(np.uint32)), what


@pytest.mark.parametrize("dark", [False, True])
def test_upsampled_hdr_pq_packed_colour_chain(gpu_ctx, oracle, dark):
    """2x upsampling with the HDR op list (XybToMixedLms -> Matrix -> GamutMap -> Matrix -> PQ) takes the packed colour
    chain of upsample2_lds_kernel<2>: pixel pairs as packed f32, constants as pairs in LDS.  `dark`: an image around
    black (zero LF, small coefficients) so that samples fall below linear_to_pq's 1e-4 threshold (the second pair of
    rational polynomials, chosen per element under a wave-uniform test) and on both sides of zero (copysign).  The
    general colour code (JXLGPU_UP2_VARIANT=2) must give the same bits — checked through the oracle."""
    wl = VardctWorkload(264, 200, seed=61, epf_iters=1, upsampling=2, intensity_target=4000.0, hdr_pq=True)
    if dark:
        for p in wl.lfq:
            p[...] = 0
        wl.coeff //= 8
    exp, _ = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, 528, 400)
    if dark:
        lin = np.abs(exp)
        assert (lin < 0.2).mean() > 0.5, "the dark case is expected to be dark"
    f = gpu_ctx.vardct_upload(wl.desc(coeff_transport="grouped"))
    try:
        got = gpu_ctx.vardct_render(f, abi.STAGE_ALL)
    finally:
        f.free()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
