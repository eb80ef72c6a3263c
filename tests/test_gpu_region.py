"""Region (cropped) renders: jxlgpu_vardct_render_region / jxlgpu_modular_render_region must return
exactly the rectangle a whole-frame render holds there — the reference's own contract for
`render_frame_cropped` (jxl-oxide-tests/tests/crop/mod.rs:8-107: four random crops per image with
sides in [128, size / 2], plus fixed crops; tolerance 1e-6 there, bit-exact here).  Every whole-frame
render a test crops from is compared with the oracle first (`_pin`), so a region equals the ORACLE's
rectangle, not only the device's own."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload
from jxl_oxide_amd.synth_modular import ModularWorkload

pytestmark = pytest.mark.gpu
S_ALL = abi.STAGE_ALL


def _pin(full, exp, what):
    """the whole-frame render the regions are compared with is the oracle's, bit for bit"""
    assert full.shape == exp.shape, (what, full.shape, exp.shape)
    assert np.array_equal(full.view(np.uint32), exp.view(np.uint32)), f"{what}: the whole-frame render differs from the oracle"


def _check(ctx, frame, full, regions, stages, render):
    for (left, top, w, h) in regions:
        got = render(frame, stages, (left, top, w, h))
        exp = full[:, top:top + h, left:left + w]
        assert got.shape == exp.shape, (left, top, w, h)
        if not np.array_equal(got.view(np.uint32), exp.view(np.uint32)):
            bad = np.argwhere(got.view(np.uint32) != exp.view(np.uint32))
            raise AssertionError(f"region {(left, top, w, h)}: {bad.shape[0]} samples differ, first at {bad[0]}")


def _random_regions(rng, width, height, n=4):
    """crop/mod.rs:14-33: sides uniform in [128, max(size / 2, 128)] (clamped to the frame here)."""
    out = []
    for _ in range(n):
        w = int(rng.integers(min(128, width), max(width // 2, min(128, width)) + 1))
        h = int(rng.integers(min(128, height), max(height // 2, min(128, height)) + 1))
        out.append((int(rng.integers(0, width - w + 1)), int(rng.integers(0, height - h + 1)), w, h))
    return out


CASES = [
    dict(),                                                  # Gabor + EPF 2, sRGB: streaming + ring kernels
    dict(epf_iters=3),                                       # tile kernel (step 0) + streaming
    dict(epf_iters=1, gabor=False),                          # tile kernel only
    dict(epf_iters=0, gabor=False),                          # no filters: untile + colour
    dict(epf_iters=0, gabor=True),
    dict(intensity_target=4000.0, hdr_pq=True, epf_iters=3, upsampling=2),   # BASELINE config 5's pipeline
    dict(upsampling=4, epf_iters=2),
    dict(upsampling=8, epf_iters=1),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("transport", ["grouped", "dense_i32"])
def test_vardct_region_equals_full_frame_rectangle(gpu_ctx, oracle, case, transport):
    w, h = (1040, 800) if case.get("upsampling", 1) == 1 else (328, 264)
    wl = VardctWorkload(w, h, seed=17, **case)
    frame = gpu_ctx.vardct_upload(wl.desc(coeff_transport=transport))
    try:
        full = gpu_ctx.vardct_render(frame, S_ALL)
        W, H = full.shape[2], full.shape[1]
        _pin(full, oracle.vardct_render(wl.desc(), S_ALL, W, H)[0], f"{case} {transport}")
        rng = np.random.default_rng(11)
        regions = _random_regions(rng, W, H) + [
            (0, 0, W, H), (0, 0, 1, 1), (W - 1, H - 1, 1, 1), (0, H // 3, W, 9), (W // 2 - 3, 0, 7, H),
            (5, 7, 130, 41), (W - 150, H - 140, 150, 140), (0, 0, 33, 33), (W - 20, 3, 20, 300 if H > 310 else H - 3)]
        _check(gpu_ctx, frame, full, regions, S_ALL, gpu_ctx.vardct_render_region)
        # a whole-frame render after region renders is unchanged (the transform output is rebuilt every time)
        again = gpu_ctx.vardct_render(frame, S_ALL)
        assert np.array_equal(again.view(np.uint32), full.view(np.uint32))
        # stage masks: a region of the bare transform output, and of the filters without colour
        for stages in (abi.STAGE_LF | abi.STAGE_TRANSFORM, S_ALL & ~abi.STAGE_COLOR):
            f2 = gpu_ctx.vardct_render(frame, stages)
            _check(gpu_ctx, frame, f2, _random_regions(rng, f2.shape[2], f2.shape[1], 2), stages, gpu_ctx.vardct_render_region)
    finally:
        frame.free()


def test_fixed_crops_of_the_reference_suite(gpu_ctx, oracle):
    """The fixed regions of crop/mod.rs:196-222, scaled into a 2600 x 2500 frame where they fit as they are."""
    wl = VardctWorkload(2600, 2500, seed=23)
    frame = gpu_ctx.vardct_upload(wl.desc(coeff_transport="grouped"))
    try:
        full = gpu_ctx.vardct_render(frame, S_ALL)
        _pin(full, oracle.vardct_render(wl.desc(), S_ALL, 2600, 2500)[0], "2600 x 2500")
        regions = [(527, 298, 179, 258), (1711, 800, 315, 571), (776, 1745, 1159, 359), (169, 194, 195, 162),
                   (81, 302, 242, 163), (468, 356, 460, 325), (524, 475, 361, 147), (1893, 35, 707, 659),
                   (850, 929, 1750, 1220), (1568, 1460, 1032, 814), (877, 2353, 936, 137), (90, 460, 368, 128)]
        _check(gpu_ctx, frame, full, regions, S_ALL, gpu_ctx.vardct_render_region)
    finally:
        frame.free()


def test_region_clipping_and_errors(gpu_ctx, oracle):
    wl = VardctWorkload(300, 200, seed=3)
    frame = gpu_ctx.vardct_upload(wl.desc())
    try:
        full = gpu_ctx.vardct_render(frame, S_ALL)
        _pin(full, oracle.vardct_render(wl.desc(), S_ALL, 300, 200)[0], "300 x 200")
        # a region reaching outside the frame is intersected with it (Region::intersection, render.rs:39-44)
        import ctypes as C
        out = np.zeros((3, 60, 70), dtype=np.float32)
        o = abi.Out()
        for c in range(3):
            o.planes[c] = out[c].ctypes.data_as(abi.f32p)
        o.stride, o.mem = 70, abi.MEM_HOST
        r = abi.Region(-30, 160, 100, 100)   # -> (0, 160, 70, 40)
        gpu_ctx._check(gpu_ctx.lib.jxlgpu_vardct_render_region(gpu_ctx.handle, frame.handle, S_ALL, C.byref(r), C.byref(o)))
        assert np.array_equal(out[:, :40, :70].view(np.uint32), full[:, 160:200, 0:70].view(np.uint32))
        with pytest.raises(Exception) as e:
            gpu_ctx.vardct_render_region(frame, S_ALL, (300, 0, 10, 10))
        assert e.value.code == abi.ERR_INVALID_ARG
        with pytest.raises(Exception) as e:
            gpu_ctx.vardct_render_region(frame, abi.STAGE_LF, (0, 0, 10, 10))
        assert e.value.code == abi.ERR_INVALID_ARG
    finally:
        frame.free()
    # noise is seeded per absolute group: the frame renders whole and the region is cropped from it
    wl = VardctWorkload(264, 200, seed=4, noise=True)
    frame = gpu_ctx.vardct_upload(wl.desc())
    try:
        full = gpu_ctx.vardct_render(frame, S_ALL)
        _pin(full, oracle.vardct_render(wl.desc(), S_ALL, 264, 200)[0], "264 x 200 with noise")
        got = gpu_ctx.vardct_render_region(frame, S_ALL, (10, 10, 50, 50))
        assert np.array_equal(got.view(np.uint32), full[:, 10:60, 10:60].view(np.uint32))
        # a region that overhangs the frame comes back at the intersection's size
        got = gpu_ctx.vardct_render_region(frame, S_ALL, (230, 180, 100, 100))
        assert got.shape == (3, 20, 34)
        assert np.array_equal(got.view(np.uint32), full[:, 180:200, 230:264].view(np.uint32))
    finally:
        frame.free()


def test_modular_region(gpu_ctx, oracle):
    stages = S_ALL | abi.STAGE_MODULAR_TO_FLOAT
    for kw in (dict(epf_iters=2), dict(epf_iters=0), dict(epf_iters=3, gabor=True)):
        wl = ModularWorkload(700, 520, kind="squeeze", lossy=True, i16=True, seed=8, residual=6, **kw)
        frame = gpu_ctx.modular_upload(wl.desc())
        try:
            full = gpu_ctx.modular_render(frame, stages)
            _pin(full, oracle.modular_render(wl.desc(), stages, 700, 520), f"modular {kw}")
            rng = np.random.default_rng(2)
            _check(gpu_ctx, frame, full, _random_regions(rng, 700, 520) + [(0, 0, 700, 520), (690, 510, 10, 10)], stages,
                   gpu_ctx.modular_render_region)
        finally:
            frame.free()


@pytest.mark.parametrize("kind", ["ycbcr420", "ycbcr422", "ycbcr440"])
def test_subsampled_modular_region_is_cropped_from_the_whole_frame(gpu_ctx, oracle, kind):
    """do_ycbcr Modular frames with jpeg_upsampling: the planes are upsampled whole, the region is cropped from the result."""
    stages = S_ALL | abi.STAGE_MODULAR_TO_FLOAT
    for kw, size in ((dict(), (301, 271)), (dict(gabor=True, epf_iters=2), (520, 300))):
        wl = ModularWorkload(*size, kind=kind, i16=True, seed=6, xyb=False, **kw)
        frame = gpu_ctx.modular_upload(wl.desc())
        try:
            full = gpu_ctx.modular_render(frame, stages)
            rng = np.random.default_rng(3)
            w, h = size
            _pin(full, oracle.modular_render(wl.desc(), stages, w, h), f"{kind} {kw}")
            _check(gpu_ctx, frame, full, _random_regions(rng, w, h, 3) + [(0, 0, w, h), (w - 9, h - 7, 9, 7), (1, 1, 2, 2)], stages,
                   gpu_ctx.modular_render_region)
        finally:
            frame.free()


def test_jpeg_transcode_region_is_cropped_from_the_whole_frame(gpu_ctx, oracle):
    from jxl_oxide_amd.synth import JpegWorkload
    wl = JpegWorkload(328, 264, mode="420", seed=2)
    frame = gpu_ctx.vardct_upload(wl.desc())
    try:
        full = gpu_ctx.vardct_render(frame, S_ALL)
        _pin(full, oracle.vardct_render(wl.desc(), S_ALL, 328, 264)[0], "JPEG transcode 4:2:0")
        _check(gpu_ctx, frame, full, [(17, 9, 200, 131), (0, 0, 328, 264)], S_ALL, gpu_ctx.vardct_render_region)
    finally:
        frame.free()
