"""HLG targets on the device (ABI 23): hlg_inverse_oo + linear_to_hlg (jxl-color/src/tf.rs:118-160) through every route
the colour stage has — behind the fused filter kernels, behind upsampling and noise, on a region, in a batch, on a Modular
frame — against the oracle, whose HLG ops call the platform libm as the reference does.  The device evaluates glibc's
logf / powf restated (csrc/libm_f32.h): the first test compares exactly those device functions with this box's libm."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import SRGB_LUMINANCES, VardctWorkload, configure_color
from util import assert_same_bits_or_nan

pytestmark = pytest.mark.gpu
S_ALL = abi.STAGE_ALL

MODES = [("hlg", 1000.0), ("hlg", 4000.0), ("hlg", 300.0), ("hlg", 255.0), ("pq_to_hlg", 4000.0), ("pq_to_hlg", 10000.0),
         ("pq_to_hlg_1000", 1000.0)]


def _libm():
    lib = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    lib.powf.restype = C.c_float
    lib.powf.argtypes = [C.c_float, C.c_float]
    lib.logf.restype = C.c_float
    lib.logf.argtypes = [C.c_float]
    lib.log2f.restype = C.c_float
    lib.log2f.argtypes = [C.c_float]
    return lib


def _same(a, b):
    an, bn = np.isnan(a), np.isnan(b)
    return np.array_equal(an, bn) and np.array_equal(np.where(an, np.float32(0), a).view(np.uint32),
                                                     np.where(bn, np.float32(0), b).view(np.uint32))


def test_device_libm_equals_the_hosts(gpu_ctx):
    """csrc/libm_f32.h as the gfx950 build evaluates it == this box's libm: 2^22 random bit patterns (every class of
    float), a dense sweep of the range the HLG ops feed it, and the special values."""
    lib = _libm()
    rng = np.random.default_rng(23)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754942e-38, 1.17549435e-38,
                        3.4028235e38, 0.5, 2.0, 0.715, 11.7, 1.0 / 12.0], dtype=np.float32)
    x = np.concatenate([rng.integers(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        rng.uniform(1e-6, 16.0, 1 << 20).astype(np.float32), special])
    got = gpu_ctx.selftest_libm(0, x)
    vlog = np.vectorize(lambda v: lib.logf(float(v)), otypes=[np.float32])
    sample = np.concatenate([np.arange(0, x.size, 37), np.arange(x.size - special.size, x.size)])   # ctypes calls are slow
    assert _same(got[sample], vlog(x[sample])), "device logf differs from the host's libm"
    f = np.float32
    for it in (1000.0, 4000.0, 400.0, 10000.0):
        gamma = f(1.2) * f(lib.powf(f(1.111), lib.log2f(f(it) / f(1e3))))
        y = float((f(1.0) - gamma) / gamma)
        got = gpu_ctx.selftest_libm(1, x, y)
        vpow = np.vectorize(lambda v: lib.powf(float(v), y), otypes=[np.float32])
        assert _same(got[sample], vpow(x[sample])), f"device powf(x, {y}) differs from the host's libm"
    # integer exponents: the sign of a negative base survives an odd one
    for y in (3.0, -3.0, 2.0):
        got = gpu_ctx.selftest_libm(1, x, y)
        vpow = np.vectorize(lambda v: lib.powf(float(v), y), otypes=[np.float32])
        assert _same(got[sample], vpow(x[sample])), f"device powf(x, {y}) differs from the host's libm"


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


@pytest.mark.parametrize("mode,it", MODES)
def test_hlg_op_lists(gpu_ctx, oracle, mode, it):
    """Behind the fused tile kernel, as a colour-only stage, and behind the streaming + ring kernels of the default filters."""
    wl = VardctWorkload(200, 136, seed=61, epf_iters=1, intensity_target=it, color_mode=mode)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_same_bits_or_nan(got, exp, f"{mode} {it} (fused filters)")
    cs = abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_COLOR
    got, exp = _both(gpu_ctx, oracle, wl, cs)
    assert_same_bits_or_nan(got, exp, f"{mode} {it} (colour-only stage)")
    wl = VardctWorkload(264, 200, seed=62, intensity_target=it, color_mode=mode)
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_same_bits_or_nan(got, exp, f"{mode} {it} (streaming kernel)")
    # the op list is not a no-op dressed as one: against the same frame with a linear target
    wl2 = VardctWorkload(264, 200, seed=62, intensity_target=it)
    wl2.color.transfer_function = abi.TF_LINEAR
    lin, _ = oracle.vardct_render(wl2.desc(), S_ALL, 264, 200)
    assert np.nanmax(np.abs(lin - exp)) > 1e-2


@pytest.mark.parametrize("up,noise,epf", [(2, False, 2), (2, True, 1), (4, False, 0), (8, False, 1), (1, True, 2)])
def test_hlg_behind_upsampling_and_noise(gpu_ctx, oracle, up, noise, epf):
    """The upsampling kernels run without their fused colour epilogue for HLG op lists; the staged colour kernel follows."""
    wl = VardctWorkload(136 if up > 1 else 264, 72 if up > 1 else 200, seed=70 + up, epf_iters=epf, gabor=epf > 0, upsampling=up,
                        noise=noise, intensity_target=4000.0, color_mode="pq_to_hlg")
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert got.shape[1:] == (wl.height * up, wl.width * up)
    assert_same_bits_or_nan(got, exp, f"pq_to_hlg up{up} noise={noise}")


def test_hlg_region_and_batch(gpu_ctx, oracle):
    wls = [VardctWorkload(520, 300, seed=80 + i, intensity_target=it, color_mode=mode)
           for i, (mode, it) in enumerate([("hlg", 1000.0), ("pq_to_hlg", 4000.0), ("pq_to_hlg_1000", 1000.0)])]
    wls.append(VardctWorkload(520, 300, seed=84))     # a plain sRGB frame in the same batch keeps its batched post launch
    exps = [oracle.vardct_render(wl.desc(), S_ALL, 520, 300)[0] for wl in wls]
    frames = [gpu_ctx.vardct_upload(wl.desc()) for wl in wls]
    try:
        gpu_ctx.vardct_render_batch(frames, S_ALL)
        gpu_ctx.synchronize()
        for i, (f, exp) in enumerate(zip(frames, exps)):
            assert_same_bits_or_nan(gpu_ctx.download_result(f), exp, f"batched frame {i}")
        for (left, top, w, h) in [(0, 0, 520, 300), (17, 9, 200, 131), (300, 150, 220, 150), (511, 291, 9, 9)]:
            got = gpu_ctx.vardct_render_region(frames[1], S_ALL, (left, top, w, h))
            assert_same_bits_or_nan(got, exps[1][:, top:top + h, left:left + w], f"region {(left, top, w, h)}", min_finite=0.0)
    finally:
        for f in frames:
            f.free()


def test_hlg_modular_frame(gpu_ctx, oracle):
    from jxl_oxide_amd.synth_modular import ModularWorkload
    wl = ModularWorkload(200, 136, kind="squeeze", lossy=True, epf_iters=2, gabor=True)
    wl.color.intensity_target = 1000.0
    configure_color(wl.color, "hlg")
    d = wl.desc()
    stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
    exp = oracle.modular_render(d, stages, 200, 136)
    f = gpu_ctx.modular_upload(d)
    try:
        got = gpu_ctx.modular_render(f, stages)
    finally:
        f.free()
    assert_same_bits_or_nan(got, exp, "modular XYB frame, HLG target")


def test_gamut_map_without_a_tone_map(gpu_ctx, oracle):
    """tm_gamut_map without tone_map is ONE op list of the reference: PQ -> HLG of a 1000-nit image (convert.rs:521-528; served,
    MODES "pq_to_hlg_1000" above).  With any other transfer function it is a stale field of the caller: refused since round 6
     instead of applying an extra GamutMap."""
    wl = VardctWorkload(264, 200, seed=90, intensity_target=1000.0)
    configure_color(wl.color, "pq_to_hlg_1000")
    assert wl.color.tm_gamut_map == 1 and wl.color.tone_map == 0 and wl.color.transfer_function == abi.TF_HLG
    got, exp = _both(gpu_ctx, oracle, wl, S_ALL)
    assert_same_bits_or_nan(got, exp, "GamutMap{0.1} -> HLG", min_finite=1.0)
    wl.color.tm_gamut_map = 0
    _, off = _both(gpu_ctx, oracle, wl, S_ALL)
    assert np.nanmax(np.abs(off - exp)) > 1e-3
    stale = VardctWorkload(264, 200, seed=90, intensity_target=1000.0)   # plain sRGB target with the field left set
    stale.color.tm_luminances[:] = SRGB_LUMINANCES
    stale.color.tm_gamut_map = 1
    stale.color.tm_gamut_saturation_factor = 0.1
    with pytest.raises(Exception) as e:
        gpu_ctx.vardct_upload(stale.desc())
    assert e.value.code == abi.ERR_UNSUPPORTED


def test_bad_hlg_parameters_are_refused(gpu_ctx):
    for bad in (-1.0, float("nan"), float("inf")):
        wl = VardctWorkload(64, 64, seed=5)
        wl.color.hlg_ootf_intensity_target = bad
        with pytest.raises(Exception) as e:
            gpu_ctx.vardct_upload(wl.desc())
        assert e.value.code == abi.ERR_UNSUPPORTED
