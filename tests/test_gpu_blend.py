"""GPU parity for the compositing primitive (blend_single): random rectangle
lists, every mode, overlapping patches applied in list order."""
import numpy as np
import pytest

from jxl_oxide_amd import abi
from test_oracle_blend import _rect, blend

pytestmark = pytest.mark.gpu


class _Hip:
    """Device buffers through the HIP runtime the library itself is linked against (no torch: a
    second HIP runtime in the process would not see the GPU the first one holds)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        try:
            self.lib = C.CDLL("libamdhip64.so")
        except OSError:
            self.lib = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def upload(self, a):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), a.nbytes) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, like):
        out = np.empty_like(like)
        assert self.lib.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.ptrs:
            self.lib.hipFree(p)


def test_blend_rect_lists(gpu_ctx, oracle):
    hip = _Hip()
    rng = np.random.default_rng(4)
    BW, BH, NW, NH = 300, 200, 128, 96
    mk = lambda h, w: rng.uniform(-0.2, 1.2, size=(h, w)).astype(np.float32)
    base, new, ba, na = mk(BH, BW), mk(NH, NW), mk(BH, BW), mk(NH, NW)
    t_base, t_new, t_ba, t_na = [hip.upload(a) for a in (base, new, ba, na)]
    rects_dev, rects_host = [], []
    for i in range(120):  # many overlap: patches of one reference sprinkled over the canvas
        w, h = int(rng.integers(1, 60)), int(rng.integers(1, 40))
        bx, by = int(rng.integers(0, BW - w + 1)), int(rng.integers(0, BH - h + 1))
        nx, ny = int(rng.integers(0, NW - w + 1)), int(rng.integers(0, NH - h + 1))
        mode = int(rng.integers(0, 7))
        flags = [int(v) for v in rng.integers(0, 2, 3)]
        use_ba, use_na = bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
        kw = dict(clamp=flags[0], swapped=flags[1], premultiplied=flags[2], bx=bx, by=by, nx=nx, ny=ny)
        rects_host.append(_rect(mode, w, h, base_alpha=ba if use_ba else None, new_alpha=na if use_na else None, **kw))
        rd = _rect(mode, w, h, **kw)
        if use_ba:
            rd.base_alpha, rd.base_alpha_stride = t_ba, BW
        if use_na:
            rd.new_alpha, rd.new_alpha_stride = t_na, NW
        rects_dev.append(rd)
    exp = base
    for r in rects_host:
        exp = blend(oracle, exp, new, r)
    gpu_ctx.blend_rects(t_base, BW, BW, BH, t_new, NW, NW, NH, rects_dev)
    got = hip.download(t_base, base)
    hip.free()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_blend_rejects_rectangles_outside_the_planes(gpu_ctx):
    hip = _Hip()
    t = hip.upload(np.zeros((8, 8), dtype=np.float32))
    with pytest.raises(Exception) as e:
        gpu_ctx.blend_rects(t, 8, 8, 8, t, 8, 8, 8, [_rect(abi.BLEND_ADD, 5, 5, bx=4, by=4)])
    assert e.value.code == abi.ERR_INVALID_ARG
    hip.free()
