"""GPU parity for the Modular stage: integer results must be bit-exact against the oracle (and,
for lossless chains, against the original integers); the float tail within 1 ULP."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth_modular import ModularWorkload
from util import assert_ulp

pytestmark = pytest.mark.gpu


def _inverse_both(gpu_ctx, oracle, wl):
    d = wl.desc()
    exp = oracle.modular_inverse(d, wl.shapes(), wl.dtype)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_inverse(f, wl.shapes(), wl.dtype)
        again = gpu_ctx.modular_inverse(f, wl.shapes(), wl.dtype)  # re-runnable from the uploaded state
    finally:
        f.free()
    for c in range(len(exp)):
        assert np.array_equal(got[c], exp[c]), f"channel {c} differs from the oracle"
        assert np.array_equal(again[c], exp[c]), f"channel {c} differs on the second run"
    return got


@pytest.mark.parametrize("i16", [True, False])
@pytest.mark.parametrize("size", [(256, 256), (70, 45), (9, 200), (1, 17), (33, 1), (600, 333)])
def test_squeeze_lossless(gpu_ctx, oracle, size, i16):
    w, h = size
    wl = ModularWorkload(w, h, kind="squeeze", lossy=False, xyb=False, i16=i16, seed=w + h)
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


@pytest.mark.parametrize("i16", [True, False])
def test_squeeze_lossy_and_wrapping(gpu_ctx, oracle, i16):
    _inverse_both(gpu_ctx, oracle, ModularWorkload(333, 200, kind="squeeze", lossy=True, i16=i16))
    _inverse_both(gpu_ctx, oracle, ModularWorkload(130, 97, kind="raw", i16=i16, seed=5))


@pytest.mark.parametrize("i16", [True, False])
def test_segment_parallel_squeeze(gpu_ctx, oracle, i16):
    """Rectangles long enough for the segment-parallel kernels (>= 2 segments of 128 pairs),
    including random data whose guessed carries fail and take the serial fix-up path."""
    wl = ModularWorkload(1100, 720, kind="squeeze", lossy=False, xyb=False, i16=i16, seed=11)
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])
    _inverse_both(gpu_ctx, oracle, ModularWorkload(1100, 720, kind="squeeze", lossy=True, i16=i16, seed=12))
    _inverse_both(gpu_ctx, oracle, ModularWorkload(1040, 530, kind="raw", i16=i16, seed=13))


@pytest.mark.parametrize("rct_type", [0, 6, 10, 19, 23, 34, 36, 41])
def test_rct_types(gpu_ctx, oracle, rct_type):
    wl = ModularWorkload(120, 90, kind="squeeze", lossy=False, xyb=False, rct_type=rct_type, seed=rct_type)
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


@pytest.mark.parametrize("i16", [True, False])
def test_config1_lossless_rgb8(gpu_ctx, oracle, i16):
    wl = ModularWorkload(256, 256, kind="lossless_rgb8", i16=i16)
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])
    wl = ModularWorkload(300, 270, kind="lossless_rgb8", seed=3, i16=i16)
    _inverse_both(gpu_ctx, oracle, wl)


@pytest.mark.parametrize("predictor", [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13])
def test_single_leaf_predictors(gpu_ctx, oracle, predictor):
    """M4: every stateless predictor, wavefront kernel vs oracle vs the original image."""
    for (w, h, i16) in [(300, 270, True), (257, 3, False), (2, 40, True)]:
        wl = ModularWorkload(w, h, kind="predictor", predictor=predictor, i16=i16, seed=predictor,
                             pred_offset=(0 if predictor % 2 else 3))
        got = _inverse_both(gpu_ctx, oracle, wl)
        for c in range(3):
            assert np.array_equal(got[c], wl.expected[c])


@pytest.mark.parametrize("size,i16", [((40, 24), False), ((1, 9), True), ((9, 1), False), ((2, 2), True), ((70, 33), True)])
def test_self_correcting_predictor(gpu_ctx, oracle, size, i16):
    w, h = size
    wl = ModularWorkload(w, h, kind="predictor", predictor=6, i16=i16, seed=w)
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


def test_self_correcting_predictor_full_tiles(gpu_ctx, oracle):
    """300x270 = full 256-row tiles plus edge tiles; random residuals (the Python forward pass is
    too slow at this size), device against oracle only, with a multiplier and an offset."""
    wl = ModularWorkload(300, 270, kind="predictor", predictor=1, i16=False, seed=2)
    rng = np.random.default_rng(9)
    wl.buffers = [rng.integers(-40, 40, size=(270, 300)).astype(np.int32) for _ in range(3)]
    wl.residual_predictor, wl.residual_multiplier, wl.residual_offset = 6, 3, -2
    wl.expected = None
    _inverse_both(gpu_ctx, oracle, wl)
    wl.sample_type, wl.dtype = abi.SAMPLE_I16, np.int16
    wl.buffers = [b.astype(np.int16) for b in wl.buffers]
    _inverse_both(gpu_ctx, oracle, wl)


@pytest.mark.parametrize("predictor", [6, 13, 5])
@pytest.mark.parametrize("size", [(3, 40), (5, 300), (13, 7), (100, 90), (257, 130), (600, 700)])
def test_lane_packed_subgrid_shapes(gpu_ctx, oracle, predictor, size):
    """The lane-packed predictor kernel gives a subgrid P = pow2ceil(gw) / 4 lanes and lets a lane walk rows
    k, k + P, ...: widths of 1-4 columns (P = 1: one lane does every row), non-power-of-two widths (idle columns
    in a round), several subgrids per wave (edge tiles of 257 x 130), and with group_dim 512 subgrids of up to
    512 x 512 (D = 8; more rows than the workgroup-per-subgrid kernel takes).  Random residuals, device against
    oracle, i32 and i16 buffers."""
    w, h = size
    gd = 512 if w > 512 else 256
    wl = ModularWorkload(w, h, kind="predictor", predictor=1, i16=False, seed=2, group_dim=gd)
    rng = np.random.default_rng(w * 31 + h + predictor)
    wl.buffers = [rng.integers(-40, 40, size=(h, w)).astype(np.int32) for _ in range(3)]
    wl.residual_predictor, wl.residual_multiplier, wl.residual_offset = predictor, 2, 1
    wl.expected = None
    _inverse_both(gpu_ctx, oracle, wl)
    wl.sample_type, wl.dtype = abi.SAMPLE_I16, np.int16
    wl.buffers = [b.astype(np.int16) for b in wl.buffers]
    _inverse_both(gpu_ctx, oracle, wl)


@pytest.mark.parametrize("predictor", [6, 13, 5])
@pytest.mark.parametrize("size", [(1100, 1060), (1024, 300), (777, 1500)])
def test_group_dim_1024_subgrids(gpu_ctx, oracle, predictor, size):
    """`group_dim` 1024 (jxl-frame/src/header.rs:299-301, group_size_shift 3): predictor subgrids of up to 1024 x 1024.  A
    subgrid wider than 512 columns takes all 64 lanes of a wave with rows trailing by D = 16 columns, so row r - 2 is
    32 ring columns ahead: the lane kernels run with the 64-column sample ring (and 1024-column error rows).  Sizes: a
    full 1024 x 1024 subgrid with 76- and 36-wide edge subgrids beside it (mixed D in one launch), one 1024-wide
    subgrid of 300 rows, a 777-wide one (idle columns in every round) over two rows of groups.  Random residuals,
    device against oracle, i32 and i16 buffers, narrow (32-bit) and reference (64-bit) forms of predictor 6."""
    w, h = size
    wl = ModularWorkload(w, h, kind="predictor", predictor=1, i16=False, seed=3, group_dim=1024)
    rng = np.random.default_rng(w * 17 + h + predictor)
    wl.buffers = [rng.integers(-40, 40, size=(h, w)).astype(np.int32) for _ in range(3)]
    wl.residual_predictor, wl.residual_multiplier, wl.residual_offset = predictor, 2, 1
    wl.expected = None
    _inverse_both(gpu_ctx, oracle, wl)
    wl.sample_type, wl.dtype = abi.SAMPLE_I16, np.int16
    wl.buffers = [b.astype(np.int16) for b in wl.buffers]
    _inverse_both(gpu_ctx, oracle, wl)


@pytest.mark.parametrize("amp", [40, 1 << 15, 1 << 21])
def test_self_correcting_predictor_leaves_the_32_bit_range(oracle, amp):
    """The self-correcting predictor runs in 32-bit arithmetic while |sample| < 2^17 and |true_err| < 2^19 and is
    redone in the reference's 64-bit arithmetic, from the untouched residuals, for every wave that leaves that range
    (predict_lanes_narrow_kernel -> predict_lanes_kernel).  Residuals of +-40 never do, +-2^15 do in places, +-2^21
    everywhere; i32 buffers, device against oracle.  Runs in a child process with JXLGPU_DEBUG_SYNC (which reports
    the number of redone waves)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from jxl_oxide_amd import runtime
from jxl_oxide_amd.synth_modular import ModularWorkload
from oracle import pyoracle
amp = int(sys.argv[1])
ctx = runtime.Context(0)
wl = ModularWorkload(300, 270, kind="predictor", predictor=1, i16=False, seed=2)
rng = np.random.default_rng(amp)
wl.buffers = [rng.integers(-amp, amp + 1, size=(270, 300)).astype(np.int32) for _ in range(3)]
wl.residual_predictor, wl.residual_multiplier, wl.residual_offset = 6, 1, 0
d = wl.desc()
exp = pyoracle.modular_inverse(d, wl.shapes(), wl.dtype)
f = ctx.modular_upload(d)
got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
again = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
f.free()
assert all(np.array_equal(g, e) for g, e in zip(got, exp))
assert all(np.array_equal(g, e) for g, e in zip(again, exp))
print("NARROW_OK")
"""
    env = dict(os.environ, JXLGPU_DEBUG_SYNC="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, str(amp)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert "NARROW_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    redone = [int(l.split(":")[1].split("of")[0]) for l in r.stderr.splitlines() if "redone in 64-bit arithmetic" in l]
    assert redone, r.stderr[-1000:]
    if amp <= 40:
        assert sum(redone) == 0
    if amp >= 1 << 21:
        assert sum(redone) > 0, "the test did not exercise the 64-bit redo"


@pytest.mark.parametrize("first", [1 << 18, 1 << 30, -(1 << 30), (1 << 31) - 1])
def test_self_correcting_predictor_large_flat_samples(gpu_ctx, oracle, first):
    """Samples far outside the 32-bit form's range with SMALL prediction errors (one large first residual, then a flat image): the
    range guard has to look at the sample itself — 8 * sample does not fit — and must not lose a large value to its own
    arithmetic (2^30 + 2^17, shifted left by two, is back in range)."""
    wl = ModularWorkload(300, 270, kind="predictor", predictor=1, i16=False, seed=2)
    rng = np.random.default_rng(7)
    wl.buffers = [rng.integers(-3, 4, size=(270, 300)).astype(np.int32) for _ in range(3)]
    for b in wl.buffers:
        b[0, 0] = first
    wl.residual_predictor, wl.residual_multiplier, wl.residual_offset = 6, 1, 0
    wl.expected = None
    _inverse_both(gpu_ctx, oracle, wl)


def test_workgroup_per_subgrid_kernel_still_matches(oracle, monkeypatch):
    """JXLGPU_PRED_WG routes every subgrid through the workgroup-per-subgrid kernel (the form that serves subgrids
    wider than 512 columns)."""
    from jxl_oxide_amd import runtime
    monkeypatch.setenv("JXLGPU_PRED_WG", "1")
    monkeypatch.setenv("JXLGPU_PRED_WIDE", "1")
    ctx = runtime.Context(0)
    try:
        for kw in (dict(kind="squeeze", lossy=True, residual=6), dict(kind="palette", residual=6)):
            wl = ModularWorkload(300, 270, seed=4, **kw)
            _inverse_both(ctx, oracle, wl)
    finally:
        ctx.close()


@pytest.mark.parametrize("d_pred", [0, 1, 5, 6, 13])
@pytest.mark.parametrize("i16", [True, False])
def test_palette_with_delta_entries(gpu_ctx, oracle, d_pred, i16):
    """M3 slow path: implicit colours, delta palette, and the whole-channel predictor pass as a
    wavefront across workgroups (600 rows = three 256-row bands)."""
    from jxl_oxide_amd.synth_modular import palette_delta_reference
    wl = ModularWorkload(37, 21, kind="palette_delta", predictor=d_pred, i16=i16, seed=d_pred)
    got = _inverse_both(gpu_ctx, oracle, wl)
    if d_pred in (0, 1, 5):
        exp = palette_delta_reference(wl.index_plane, wl.palette, 29, 4, d_pred, 8, 16 if i16 else 32)
        for c in range(3):
            assert np.array_equal(got[c].astype(np.int64), exp[c])
    wl = ModularWorkload(300, 600, kind="palette_delta", predictor=d_pred, i16=i16, seed=10 + d_pred)
    _inverse_both(gpu_ctx, oracle, wl)


def test_palette(gpu_ctx, oracle):
    wl = ModularWorkload(64, 48, kind="palette")
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


@pytest.mark.parametrize("cfg", [dict(epf_iters=0), dict(epf_iters=1), dict(epf_iters=2, gabor=True)])
def test_render_xyb_tail(gpu_ctx, oracle, cfg):
    wl = ModularWorkload(200, 136, kind="squeeze", lossy=True, **cfg)
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, 200, 136)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert_ulp(got, exp, 1, f"modular render {cfg}")


@pytest.mark.parametrize("i16", [True, False])
@pytest.mark.parametrize("gabor", [False, True])
@pytest.mark.parametrize("size", [(520, 300), (200, 136), (331, 260)])
def test_post_stage_reads_the_integer_planes(oracle, monkeypatch, size, gabor, i16):
    """JXLGPU_INT_POST=1 (a measured option, off by default: 4 % slower on config 3): XYB Modular frames with EPF iters 2 — the
    packed streaming post kernel and the border-ring kernel read the integer planes of the inverse transforms and convert on
    the fly (convert_to_float_modular_xyb, jxl-render/src/image.rs:148-189: B + Y saturating in the sample type, times
    m_lf_unscaled) instead of reading a float copy made by to_float_kernel.  The same bits as the oracle and as the default
    float-copy path; an odd plane stride (331) takes the float copy either way."""
    from jxl_oxide_amd import runtime
    w, h = size
    wl = ModularWorkload(w, h, kind="squeeze", lossy=True, i16=i16, epf_iters=2, gabor=gabor, seed=w + h)
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, w, h)
    got = {}
    for mode in ("float", "int"):
        if mode == "int":
            monkeypatch.setenv("JXLGPU_INT_POST", "1")
        ctx = runtime.Context(0)
        try:
            f = ctx.modular_upload(d)
            try:
                got[mode] = ctx.modular_render(f, stages)
                again = ctx.modular_render(f, stages)
            finally:
                f.free()
        finally:
            ctx.close()
        assert np.array_equal(again.view(np.uint32), got[mode].view(np.uint32)), f"{mode}: second render differs"
    assert np.array_equal(got["int"].view(np.uint32), got["float"].view(np.uint32)), "integer-input post stage differs from the float-copy path"
    assert np.array_equal(got["int"].view(np.uint32), exp.view(np.uint32)), "post stage differs from the oracle"


@pytest.mark.parametrize("case", [
    dict(width=300, height=270, kind="predictor", i16=False, seed=1),
    dict(width=300, height=270, kind="predictor", i16=True, seed=6),
    dict(width=600, height=333, kind="squeeze", lossy=False, xyb=False, seed=2),
    dict(width=300, height=200, kind="palette", seed=3, i16=False),
    dict(width=1100, height=600, kind="squeeze", lossy=False, xyb=False, seed=4, group_dim=128, i16=False),
    dict(width=520, height=300, kind="squeeze", lossy=False, xyb=False, seed=5, leaves=[6]),
    dict(width=1100, height=700, kind="squeeze", lossy=False, xyb=False, seed=7, group_dim=1024, leaves=[6, 5, 13]),
])
def test_per_unit_leaves(gpu_ctx, oracle, case):
    """JxlGpuModularDesc::unit_leaves: every decode unit with the single node make_flat_tree leaves it (a tree that splits on
    the static properties channel / stream index: ma.rs:38-41, image.rs:477-490, 553-562) — its own predictor, offset and
    multiplier.  The host forms the predictor waves by predictor, multiplier and offset travel with the subgrid.  Device
    against oracle, and against the original image (the residuals come from the independent numpy forward, unit by unit);
    then the same buffers with multipliers other than 1 (no ground truth: device against oracle only); a wrong count is
    refused at the first inverse."""
    from jxl_oxide_amd.runtime import JxlGpuError
    kw = dict(case)
    kw.setdefault("leaves", "mixed")
    wl = ModularWorkload(**kw)
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c]), f"channel {c} differs from the original image"
    rng = np.random.default_rng(kw["seed"])
    wl.unit_leaves = [(p, int(rng.integers(-3, 4)), o) for p, _, o in wl.unit_leaves]
    wl.expected = None
    _inverse_both(gpu_ctx, oracle, wl)
    d = wl.desc()
    d.num_unit_leaves -= 1
    f = gpu_ctx.modular_upload(d)
    try:
        with pytest.raises(JxlGpuError):
            gpu_ctx.modular_inverse(f, wl.shapes(), wl.dtype)
    finally:
        f.free()


@pytest.mark.parametrize("kind", ["squeeze", "lossless_rgb8"])
def test_render_with_noise(gpu_ctx, oracle, kind):
    """Noise on Modular frames: base correlations (0, 1) (noise.rs:35), on XYB and on plain RGB."""
    from jxl_oxide_amd.synth import make_noise_params
    wl = ModularWorkload(300, 264, kind=kind, lossy=True, epf_iters=1 if kind == "squeeze" else 0)
    wl.noise = make_noise_params(3)
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, 300, 264)
    off = oracle.modular_render(d, stages & ~abi.STAGE_NOISE, 300, 264)
    assert np.abs(exp - off).max() > 1e-3
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert_ulp(got, exp, 1, f"modular noise {kind}")


def test_render_rgb8_no_colour_transform(gpu_ctx, oracle):
    wl = ModularWorkload(256, 256, kind="lossless_rgb8")
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, 256, 256)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert_ulp(got, exp, 0, "rgb8 -> float")
    assert np.array_equal(got[0], wl.expected[0].astype(np.float32) / np.float32(255))


@pytest.mark.parametrize("i16", [True, False])
@pytest.mark.parametrize("case", [dict(), dict(gabor=True), dict(epf_iters=2), dict(gabor=True, epf_iters=3),
                                  dict(epf_iters=1, residual=6), dict(gabor=True, epf_iters=2, up=2)])
def test_grayscale_frames(gpu_ctx, oracle, case, i16):
    """encoded_color_channels == 1 (jxl-render/src/render.rs:74-134): one colour channel through the inverse
    transforms (Squeeze without the chroma pre-steps, optional predictor residuals), cloned into the three filter
    inputs; plane 0 of the result is the image.  Integer result against the original, render against the oracle."""
    case = dict(case)
    up = case.pop("up", 1)
    w, h = 300, 270
    wl = ModularWorkload(w, h, kind="gray", i16=i16, seed=11, **case)
    got_int = _inverse_both(gpu_ctx, oracle, wl)
    assert np.array_equal(got_int[0], wl.expected[0])
    d = wl.desc()
    if up > 1:
        from jxl_oxide_amd.synth import _load_up_weights
        upw = _load_up_weights()
        d.upsampling.factor = up
        d.upsampling.up2_weight = upw[0].ctypes.data_as(abi.f32p)
        d.upsampling.up4_weight = upw[1].ctypes.data_as(abi.f32p)
        d.upsampling.up8_weight = upw[2].ctypes.data_as(abi.f32p)
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, w * up, h * up)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
        reg = (16, 40, 200, 96)
        got_reg = gpu_ctx.modular_render_region(f, stages, reg)
    finally:
        f.free()
    assert np.array_equal(got[0].view(np.uint32), exp[0].view(np.uint32))
    assert np.array_equal(got_reg[0].view(np.uint32), exp[0][reg[1]:reg[1] + reg[3], reg[0]:reg[0] + reg[2]].view(np.uint32))


@pytest.mark.parametrize("kind", ["ycbcr420", "ycbcr422", "ycbcr440", "ycbcr444"])
@pytest.mark.parametrize("case", [dict(), dict(gabor=True, epf_iters=2)])
@pytest.mark.parametrize("size", [(300, 270), (301, 271)])
def test_chroma_subsampled_ycbcr_frames(gpu_ctx, oracle, kind, case, size):
    """do_ycbcr Modular frames with jpeg_upsampling (jxl-render/src/render.rs:70-72, image.rs:448-485): colour channels
    Cb, Y, Cr of different sizes — int -> float per channel at its own size, upsample_jpeg, filters, ycbcr_to_rgb."""
    w, h = size
    wl = ModularWorkload(w, h, kind=kind, i16=True, seed=5, xyb=False, **case)
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, w, h)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_squeeze_fixup_path(oracle):
    """Without the run-in the guessed carries are usually wrong: the check kernel must catch every
    broken link and redo those lines serially.  Runs in a child process (the switches are read once
    per process)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from jxl_oxide_amd import runtime
from jxl_oxide_amd.synth_modular import ModularWorkload
from oracle import pyoracle
ctx = runtime.Context(0)
# (the two wide cases: horizontal steps of more than one wave of lane-pieces per row, so that links BETWEEN waves break too)
for (w, h, kind, i16) in [(700, 520, "squeeze", True), (520, 700, "squeeze", False), (640, 400, "raw", True),
                          (4400, 40, "squeeze", True), (2200, 72, "squeeze", False)]:
    wl = ModularWorkload(w, h, kind=kind, lossy=True, i16=i16, seed=w)
    d = wl.desc()
    exp = pyoracle.modular_inverse(d, wl.shapes(), wl.dtype)
    f = ctx.modular_upload(d)
    got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
    f.free()
    assert all(np.array_equal(g, e) for g, e in zip(got, exp)), (w, h, kind)
print("FIXUP_OK")
'''
    env = dict(os.environ, JXLGPU_SQZ_RUNIN="0", JXLGPU_SQZ_SEG="32", JXLGPU_DEBUG_SYNC="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert "FIXUP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    redone = [int(l.rsplit(":", 1)[1]) for l in r.stderr.splitlines() if "redone serially" in l]
    assert sum(redone) > 0, "the test did not exercise the serial fix-up path"


def test_upload_rejects_what_the_device_path_would_get_wrong(gpu_ctx):
    """Descriptor validation of jxlgpu_modular_upload: malformed upsampling / EPF / meta-channel
    parameters are INVALID_ARG instead of out-of-bounds device accesses.  (Predictor + Squeeze / Palette,
    refused until round 2, is implemented: test_predictor_on_transformed_channels.)"""
    from jxl_oxide_amd.runtime import JxlGpuError
    wl = ModularWorkload(64, 48, kind="squeeze", lossy=False, xyb=False, seed=1)
    d = wl.desc()
    d.upsampling.factor = 3
    with pytest.raises(JxlGpuError) as e:
        gpu_ctx.modular_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG
    d = wl.desc()
    d.filter.epf_iters = 4
    with pytest.raises(JxlGpuError) as e:
        gpu_ctx.modular_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG
    d = wl.desc()
    d.upsampling.factor = 2  # weights missing
    with pytest.raises(JxlGpuError) as e:
        gpu_ctx.modular_upload(d)
    assert e.value.code == abi.ERR_INVALID_ARG


def test_all_rct_codes(gpu_ctx, oracle):
    """M2: all 42 RCT codes (7 types x 6 permutations), lossless round trip through the device."""
    for rct_type in range(42):
        wl = ModularWorkload(72, 40, kind="squeeze", lossy=False, xyb=False, rct_type=rct_type, seed=rct_type)
        got = _inverse_both(gpu_ctx, oracle, wl)
        for c in range(3):
            assert np.array_equal(got[c], wl.expected[c]), rct_type


@pytest.mark.parametrize("case", [
    dict(kind="squeeze", lossy=False, xyb=False, residual=5),
    dict(kind="squeeze", lossy=False, xyb=False, residual=6),
    dict(kind="squeeze", lossy=False, xyb=False, residual=13, pred_offset=3),
    dict(kind="squeeze", lossy=False, xyb=False, residual=6, i16=False, rct_type=6),
    dict(kind="squeeze", lossy=True, residual=6),
    dict(kind="palette", residual=6),
    dict(kind="palette", residual=4, i16=False),
])
@pytest.mark.parametrize("size", [(300, 200), (1100, 720), (257, 600), (40, 9)])
def test_predictor_on_transformed_channels(gpu_ctx, oracle, case, size):
    """M4 per carved sub-channel (prepare_groups, image.rs:209-340): predictor residuals of every Squeeze
    sub-channel / palette table on its own tile grid, then the inverse transforms — HIP vs oracle, and vs
    the original image where the chain is lossless (residuals from the independent numpy forward)."""
    w, h = size
    wl = ModularWorkload(w, h, seed=3, **case)
    got = _inverse_both(gpu_ctx, oracle, wl)
    if wl.expected is not None:
        for c in range(3):
            assert np.array_equal(got[c], wl.expected[c]), f"channel {c} differs from the original image"


@pytest.mark.parametrize("case", [
    dict(kind="squeeze", lossy=False, xyb=False, residual=6),
    dict(kind="squeeze", lossy=True, residual=6, i16=False),
    dict(kind="squeeze", lossy=False, xyb=False, residual=5, i16=False, rct_type=6),
])
def test_group_dim_1024_on_squeeze_subchannels(gpu_ctx, oracle, case):
    """`group_dim` 1024 with a Squeeze chain: the shift-0 residual sub-channels are cut into 1024-wide subgrids, the deeper
    ones into (1024 >> shift) subgrids and 8192-sample LF groups (image.rs:258-306), all in one set of launches."""
    wl = ModularWorkload(2100, 1100, seed=4, group_dim=1024, **case)
    got = _inverse_both(gpu_ctx, oracle, wl)
    if wl.expected is not None:
        for c in range(3):
            assert np.array_equal(got[c], wl.expected[c]), f"channel {c} differs from the original image"


def test_predictor_squeeze_render_tail(gpu_ctx, oracle):
    """BASELINE config 3 in small: lossy Squeeze + self-correcting predictor residuals, XYB dequantisation,
    EPF, XYB -> sRGB."""
    wl = ModularWorkload(520, 300, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=9, residual=6)
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, wl.width, wl.height)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert_ulp(got, exp, 1, "lossy Squeeze + WP residuals render")


@pytest.mark.parametrize("plan", ["explicit_steps_over_residuals", "two_squeeze_transforms", "appended_then_squeezed"])
@pytest.mark.parametrize("residual", [None, 6, 5])
def test_resqueezed_residuals(gpu_ctx, oracle, plan, residual):
    """A residual rectangle of an early Squeeze step that a later step (or a second Squeeze transform) squeezes again: the
    inverse of the LATER step reads it as its average input, first.  With the self-correcting predictor the waves of the first
    steps' residuals run on a side stream (ModularState::ev_late): such a rectangle must not be among them."""
    from test_oracle_modular import RESQUEEZE_PLANS
    for size, seed in (((1100, 720), 31), ((600, 333), 32)):
        wl = ModularWorkload(size[0], size[1], kind="squeeze", lossy=False, xyb=False, seed=seed, residual=residual,
                             squeeze_plan=RESQUEEZE_PLANS[plan])
        for _ in range(2):
            got = _inverse_both(gpu_ctx, oracle, wl)
            for c in range(3):
                assert np.array_equal(got[c], wl.expected[c])


@pytest.mark.parametrize("plan", ["explicit_steps_over_residuals", "two_squeeze_transforms", "appended_then_squeezed", "default_then_residuals"])
@pytest.mark.parametrize("i16", [True, False])
def test_resqueezed_residuals_small_levels(gpu_ctx, oracle, plan, i16):
    """The same chains at sizes whose steps all go through the ONE launch for the small levels (squeeze_chain_kernel: three
    workgroups side by side, a workgroup barrier between a workgroup's steps).  A step that squeezes a residual channel again, or
    a channel that moved to another index, reads what ANOTHER workgroup's step wrote: the queue is flushed first (found by
    tests/tools/fuzz_parity.py, seed 6001: 73 x 50 `appended_then_squeezed`, a race between workgroups)."""
    from test_oracle_modular import RESQUEEZE_PLANS
    plans = dict(RESQUEEZE_PLANS, default_then_residuals=[None, [(0, 1, 3, 2), (1, 1, 3, 2)]])
    for size, seed, residual in (((73, 50), 953, 0), ((73, 50), 1, None), ((40, 41), 2, 5), ((130, 97), 3, 6), ((200, 160), 4, None)):
        wl = ModularWorkload(size[0], size[1], kind="squeeze", lossy=False, xyb=False, seed=seed, residual=residual, i16=i16,
                             squeeze_plan=plans[plan])
        for _ in range(3):
            got = _inverse_both(gpu_ctx, oracle, wl)
            for c in range(3):
                assert np.array_equal(got[c], wl.expected[c]), (size, plan, residual, c)


def _axis_cases():
    from test_oracle_modular import AXIS_TREE_CASES
    return AXIS_TREE_CASES + [dict(width=1100, height=720, kind="squeeze", lossy=False, xyb=False, seed=46, i16=False),
                              dict(width=1100, height=1060, kind="predictor", seed=47, group_dim=1024)]


@pytest.mark.parametrize("case", range(7))
def test_row_column_property_trees(gpu_ctx, oracle, case):
    """Trees that still split on property 2 (y) / 3 (x) inside a decode unit: JxlGpuModularDesc::axis_leaves (ABI 24), one leaf
    per row or per column of the unit, one PredictorState per unit (decode_slow, image.rs:1169-1228).  Device = oracle = the
    original image (the residuals come from the independent forward)."""
    from test_oracle_modular import _axis_workload
    wl = _axis_workload(_axis_cases()[case])
    kinds = {p for p, _, _ in wl.unit_leaves}
    assert abi.LEAF_BY_ROW in kinds and abi.LEAF_BY_COLUMN in kinds
    got = _inverse_both(gpu_ctx, oracle, wl)
    for c in range(3):
        assert np.array_equal(got[c], wl.expected[c])


def test_row_column_property_trees_workgroup_kernel(oracle, monkeypatch):
    """The same through the workgroup-per-subgrid kernel (JXLGPU_PRED_WG), and what the upload refuses."""
    from jxl_oxide_amd import runtime
    from test_oracle_modular import _axis_workload
    monkeypatch.setenv("JXLGPU_PRED_WG", "1")
    ctx = runtime.Context(0)
    try:
        wl = _axis_workload(dict(width=300, height=270, kind="squeeze", lossy=False, xyb=False, seed=48, leaves_preds=[6, 5, 13, 2]))
        got = _inverse_both(ctx, oracle, wl)
        for c in range(3):
            assert np.array_equal(got[c], wl.expected[c])
        # a unit entry that points past the axis table / carries an offset / an axis leaf that is itself a table: refused at upload
        d = wl.desc()
        d.num_axis_leaves = 1
        with pytest.raises(runtime.JxlGpuError):
            f = ctx.modular_upload(d)
            try:
                ctx.modular_inverse(f, wl.shapes(), wl.dtype)
            finally:
                f.free()
        bad = list(wl.axis_leaves)
        bad[0] = (abi.LEAF_BY_ROW, 1, 0)
        wl.axis_leaves = bad
        with pytest.raises(runtime.JxlGpuError):
            ctx.modular_upload(wl.desc())
    finally:
        ctx.close()
