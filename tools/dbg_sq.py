import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from jxl_oxide_amd import runtime
from jxl_oxide_amd.synth_modular import ModularWorkload
from oracle import pyoracle
ctx = runtime.Context(0)
w, h = int(sys.argv[1]), int(sys.argv[2])
wl = ModularWorkload(w, h, kind="squeeze", lossy=False, xyb=False, i16=False, seed=w + h)
d = wl.desc()
exp = pyoracle.modular_inverse(d, wl.shapes(), wl.dtype)
f = ctx.modular_upload(d)
got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
print("OK", all(np.array_equal(g, e) for g, e in zip(got, exp)), flush=True)
