"""csrc/libm_f32.h — the restatement of glibc's logf / powf that the device's HLG path evaluates — against the installed
libm, on the CPU: the header is host + device code, arithmetic in IEEE double with every fused multiply-add written out, so
what g++ computes from it here is what the gfx950 build computes.  The reference (Rust f32::ln / f32::powf) calls exactly
these libm functions (jxl-color/src/tf.rs:118-160)."""
import ctypes as C
import ctypes.util
import importlib.util
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("libm") / "libm_check")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-builtin", "-fopenmp"]
    if "fma" in open("/proc/cpuinfo").read().split():
        flags.append("-mfma")    # __builtin_fma as the instruction; without it the (exact) library fma is called
    subprocess.check_call([gxx, *flags, os.path.join(HERE, "c", "libm_check.cc"), "-o", out, "-lm"])
    return out


def _libm():
    lib = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    lib.powf.restype = C.c_float
    lib.powf.argtypes = [C.c_float, C.c_float]
    lib.log2f.restype = C.c_float
    lib.log2f.argtypes = [C.c_float]
    return lib


def hlg_exponent(intensity_target):
    """(1 - gamma) / gamma with gamma = 1.2 * 1.111^log2(it / 1000) in f32, as tf.rs:130-132 evaluates it."""
    import numpy as np
    lib = _libm()
    f = np.float32
    gamma = f(1.2) * f(lib.powf(f(1.111), lib.log2f(f(intensity_target) / f(1e3))))
    return float((f(1.0) - gamma) / gamma)


def test_tables_are_the_installed_libms():
    spec = importlib.util.spec_from_file_location("libm_tables", os.path.join(ROOT, "tools", "libm_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ok, why = mod.check()
    if ok is None:
        pytest.skip(why)
    assert ok, why


def test_logf_every_float(checker):
    r = subprocess.run([checker, "logf"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "logf mismatches: 0" in r.stdout, (r.stdout, r.stderr)


def test_powf_every_float_for_the_hlg_exponents(checker):
    """Every one of the 2^32 bit patterns of x (negative, zero, subnormal, inf, NaN included) for the exponents the inverse
    OOTF takes at 1000, 4000 and 400 nits."""
    ys = [repr(hlg_exponent(it)) for it in (1000.0, 4000.0, 400.0)]
    r = subprocess.run([checker, "powf", *ys], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("mismatches: 0") == len(ys), (r.stdout, r.stderr)


@pytest.mark.skipif(not os.environ.get("JXL_LIBM_CAMPAIGN"), reason="opt-in: JXL_LIBM_CAMPAIGN=<number of random exponents> (8 s each on 8 cores)")
def test_powf_campaign(checker):
    """The long form of the test above: random exponents in [-1, 2.5] (the HLG exponent (1 - gamma) / gamma lies in (-1, inf)),
    every float as the base.  188 exponents were run once in round 5: no mismatch."""
    import numpy as np
    rng = np.random.default_rng(int(os.environ.get("JXL_LIBM_SEED", "1")))
    ys = [repr(float(np.float32(y))) for y in rng.uniform(-1.0, 2.5, int(os.environ["JXL_LIBM_CAMPAIGN"]))]
    r = subprocess.run([checker, "powf", *ys], capture_output=True, text=True, timeout=36000)
    assert r.returncode == 0 and r.stdout.count("mismatches: 0") == len(ys), (r.stdout[-2000:], r.stderr[-2000:])

