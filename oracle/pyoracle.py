"""ORACLE — test infrastructure only.  ctypes access to oracle/_build/liboracle.so (the plain-C
restatement of the reference's generic CPU path).  Import this only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
f32p = C.POINTER(C.c_float)
_LIB_NATIVE = os.path.join(_HERE, "_build", "liboracle_native.so")
_lib = None


def use_native(on=True):
    """Switch to the -O3 -march=native build (the cpu_baseline timing copy; `make -C oracle native`
    on this box first).  Bit-identical results: tests/test_oracle_native.py."""
    global _LIB, _lib
    if on:
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
        _LIB = _LIB_NATIVE
    else:
        _LIB = os.path.join(_HERE, "_build", "liboracle.so")
    _lib = None


def build(force=False):
    deps = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    deps.append(os.path.join(_HERE, "..", "include", "jxlgpu.h"))  # the shared POD descriptors
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(f) > os.path.getmtime(_LIB) for f in deps):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_dct_1d.argtypes = [f32p, f32p, C.c_size_t, C.c_int]
        _lib.orc_dct_2d.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        _lib.orc_sec_half.restype = f32p
        _lib.orc_sec_half.argtypes = [C.c_size_t]
        _lib.orc_set_sec_half_large.argtypes = [C.c_size_t, f32p]
        _lib.orc_transform_block.argtypes = [f32p, C.c_size_t, C.c_int]
        _lib.orc_inject_llf.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_int]
        _lib.jxl_oracle_vardct_render.restype = C.c_int
        _lib.jxl_oracle_vardct_render.argtypes = [C.c_void_p, C.c_uint32, f32p * 3, C.c_uint32, f32p * 3]
        _lib.orc_gabor_plane.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
        _lib.orc_upsample_inner.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_int, f32p]
        _lib.orc_color_transform.argtypes = [f32p * 3, C.c_size_t, C.c_void_p]
        _lib.orc_format_output.restype = C.c_int
        _lib.orc_format_output.argtypes = [f32p * 3, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.jxl_oracle_modular_inverse.restype = C.c_int
        _lib.jxl_oracle_modular_inverse.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _lib.jxl_oracle_modular_render.restype = C.c_int
        _lib.jxl_oracle_modular_render.argtypes = [C.c_void_p, C.c_uint32, f32p * 3, C.c_uint32]
    return _lib


def _p(a):
    return a.ctypes.data_as(f32p)


def dct_1d(x, forward):
    io = np.ascontiguousarray(x, dtype=np.float32).copy()
    scratch = np.zeros_like(io)
    lib().orc_dct_1d(_p(io), _p(scratch), io.size, 1 if forward else 0)
    return io


def dct_2d(x, forward):
    io = np.ascontiguousarray(x, dtype=np.float32).copy()
    h, w = io.shape
    lib().orc_dct_2d(_p(io), w, w, h, 1 if forward else 0)
    return io


def transform_block(coeff, dct_select):
    io = np.ascontiguousarray(coeff, dtype=np.float32).copy()
    lib().orc_transform_block(_p(io), io.shape[1], dct_select)
    return io


def set_threads(n):
    """omp_set_num_threads for the oracle's loops; returns the thread count in effect."""
    f = lib().jxl_oracle_set_threads
    f.argtypes, f.restype = [C.c_int], C.c_int
    return int(f(int(n)))


def vardct_render(desc, stages, out_w, out_h, want_lf=False, w8=0, h8=0, out=None):
    """Returns (planes[3][h,w] or None, lf[3][h8,w8] or None).  `out` may be a preallocated
    (3, out_h, out_w) float32 array (the CPU-baseline timing loop reuses one)."""
    if out is None:
        out = np.zeros((3, out_h, out_w), dtype=np.float32)
    outp = (f32p * 3)(*[_p(out[c]) for c in range(3)])
    lf = None
    lfp = (f32p * 3)()
    if want_lf:
        lf = np.zeros((3, h8, w8), dtype=np.float32)
        lfp = (f32p * 3)(*[_p(lf[c]) for c in range(3)])
    rc = lib().jxl_oracle_vardct_render(C.byref(desc), stages, outp, out_w, lfp)
    if rc != 0:
        raise RuntimeError(f"oracle vardct_render failed: {rc}")
    return out, lf


def modular_inverse(desc, shapes, dtype):
    """shapes: [(h, w)] per channel.  Returns the list of reconstructed integer planes."""
    outs = [np.zeros(s, dtype=dtype) for s in shapes]
    arr = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
    rc = lib().jxl_oracle_modular_inverse(C.byref(desc), arr)
    if rc != 0:
        raise RuntimeError(f"oracle modular_inverse failed: {rc}")
    return outs


def modular_render(desc, stages, out_w, out_h):
    out = np.zeros((3, out_h, out_w), dtype=np.float32)
    outp = (f32p * 3)(*[_p(out[c]) for c in range(3)])
    rc = lib().jxl_oracle_modular_render(C.byref(desc), stages, outp, out_w)
    if rc != 0:
        raise RuntimeError(f"oracle modular_render failed: {rc}")
    return out


def format_output(planes, sample_format, orientation):
    """planes: (3, h, w) float32 -> (oh, ow, 3) interleaved array of f32 / u16 / u8."""
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    _, h, w = planes.shape
    ow, oh = (w, h) if orientation <= 4 else (h, w)
    dt = {0: np.float32, 1: np.uint16, 2: np.uint8}[sample_format]
    out = np.zeros((oh, ow, 3), dtype=dt)
    arr = (f32p * 3)(*[_p(planes[c]) for c in range(3)])
    rc = lib().orc_format_output(arr, w, w, h, sample_format, orientation, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle format_output failed: {rc}")
    return out


def extra_channel(ec):
    """jxlgpu_frame_render_extra's contract on the CPU: (h << L, w << L) f32."""
    w, h = ec.width << ec.upsampling_log2, ec.height << ec.upsampling_log2
    out = np.zeros((h, w), dtype=np.float32)
    f = lib().orc_extra_channel
    f.argtypes, f.restype = [C.c_void_p, f32p, C.c_size_t], C.c_int
    rc = f(C.addressof(ec), _p(out), w)
    if rc != 0:
        raise RuntimeError(f"oracle extra_channel failed: {rc}")
    return out


def format_output_n(planes, sample_format, orientation):
    """planes: list of (h, w) float32 arrays (3 colour + extra channels) -> (oh, ow, n) interleaved."""
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    h, w = planes[0].shape
    n = len(planes)
    ow, oh = (w, h) if orientation <= 4 else (h, w)
    dt = {0: np.float32, 1: np.uint16, 2: np.uint8}[sample_format]
    out = np.zeros((oh, ow, n), dtype=dt)
    arr = (f32p * n)(*[_p(p) for p in planes])
    strides = (C.c_size_t * n)(*[p.shape[1] for p in planes])
    f = lib().orc_format_output_n
    f.argtypes, f.restype = [C.POINTER(f32p), C.POINTER(C.c_size_t), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p], C.c_int
    rc = f(arr, strides, n, w, h, sample_format, orientation, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle format_output_n failed: {rc}")
    return out


def noise_group(width, height, seed0, seed1):
    """Raw noise of one group (NoiseGroup::new): (3, height, stride) floats in [1, 2)."""
    stride = -(-width // 16) * 16
    out = np.zeros((3, height, stride), dtype=np.float32)
    st = C.c_uint32(0)
    f = lib().orc_noise_group
    f.argtypes, f.restype = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, f32p, C.POINTER(C.c_uint32)], None
    f(width, height, seed0, seed1, _p(out), C.byref(st))
    assert st.value == stride
    return out


def render_noise(planes, group_dim, noise_params, corr_x, corr_b):
    """features/noise.rs on (3, h, w) float planes, in place on a copy; returns the copy."""
    a = np.ascontiguousarray(planes, dtype=np.float32).copy()
    _, h, w = a.shape
    f = lib().orc_render_noise
    f.restype = C.c_int
    f.argtypes = [f32p * 3, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_float, C.c_float]
    rc = f((f32p * 3)(*[_p(a[c]) for c in range(3)]), w, w, h, group_dim, C.byref(noise_params), corr_x, corr_b)
    if rc != 0:
        raise RuntimeError(f"orc_render_noise -> {rc}")
    return a
