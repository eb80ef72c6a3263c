#!/usr/bin/env python3
"""Which device buffer is read before it is written?  Runs a scenario under JXLGPU_GUARD=1 (every buffer filled
with 0xff) and, if it fails, once more per allocation index k with JXLGPU_GUARD_ZERO=k (that buffer zero-filled):
the k that makes the scenario pass again is the buffer whose initial contents matter.
usage: tools/guard_bisect.py [scenario ...]"""
import io
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["JXLGPU_GUARD"] = os.environ.get("JXLGPU_GUARD", "1")

from jxl_oxide_amd import abi, runtime  # noqa: E402
from jxl_oxide_amd.synth import VardctWorkload  # noqa: E402
from jxl_oxide_amd.synth_modular import ModularWorkload  # noqa: E402
from oracle import pyoracle  # noqa: E402


def sc_squeeze(ctx):
    wl = ModularWorkload(256, 256, kind="squeeze", lossy=False, xyb=False, i16=False, seed=512)
    d = wl.desc()
    exp = pyoracle.modular_inverse(d, wl.shapes(), wl.dtype)
    f = ctx.modular_upload(d)
    try:
        got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
    finally:
        f.free()
    return all(np.array_equal(g, e) for g, e in zip(got, exp))


def sc_squeeze_small(ctx):
    wl = ModularWorkload(70, 45, kind="squeeze", lossy=False, xyb=False, i16=True, seed=115)
    d = wl.desc()
    exp = pyoracle.modular_inverse(d, wl.shapes(), wl.dtype)
    f = ctx.modular_upload(d)
    try:
        got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
    finally:
        f.free()
    return all(np.array_equal(g, e) for g, e in zip(got, exp))


def sc_sparse(ctx):
    wl = VardctWorkload(520, 264, seed=2001)
    exp, _ = pyoracle.vardct_render(wl.desc(), abi.STAGE_ALL, wl.width, wl.height)
    f = ctx.vardct_upload(wl.desc(coeff_transport="sparse_i16"))
    try:
        got = ctx.vardct_render(f, abi.STAGE_ALL)
    finally:
        f.free()
    return np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def sc_types(ctx):
    """test_each_transform_type[0..10] back to back in one context (the suite fails some of them under guard)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_vardct as T
    ok = True
    for t in range(0, 12):
        try:
            T.test_each_transform_type(ctx, pyoracle, t)
        except AssertionError:
            print(f"    transform type {t}: differs")
            ok = False
    return ok


SCENARIOS = {"squeeze": sc_squeeze, "squeeze_small": sc_squeeze_small, "sparse": sc_sparse, "types": sc_types}


def run(fn, zero=None, log=False):
    os.environ.pop("JXLGPU_GUARD_ZERO", None)
    os.environ.pop("JXLGPU_GUARD_LOG", None)
    if zero is not None:
        os.environ["JXLGPU_GUARD_ZERO"] = str(zero)
    if log:
        os.environ["JXLGPU_GUARD_LOG"] = "1"
    ctx = runtime.Context(0)
    try:
        return bool(fn(ctx))
    except runtime.JxlGpuError as e:
        print("    error:", e)
        return False
    finally:
        ctx.close()


def main():
    names = sys.argv[1:] or list(SCENARIOS)
    for name in names:
        fn = SCENARIOS[name]
        ok = run(fn)
        print(f"{name}: poisoned buffers -> {'pass' if ok else 'FAIL'}", flush=True)
        if ok:
            continue
        ok_all = run(fn, "all")
        print(f"{name}: all buffers zero-filled -> {'pass' if ok_all else 'FAIL (not an initialisation problem)'}", flush=True)
        if not ok_all:
            continue
        sys.stderr.flush()
        run(fn, None, log=True)   # the allocation list (sizes) on stderr
        culprits = [k for k in range(0, 120) if run(fn, k)]
        print(f"{name}: passes again when allocation # {culprits} alone is zero-filled", flush=True)


if __name__ == "__main__":
    main()
