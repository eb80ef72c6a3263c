import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth_modular import ModularWorkload
from oracle import pyoracle
ctx = runtime.Context(0)
wl = ModularWorkload(256, 256, kind="squeeze", lossy=False, xyb=False, i16=True, seed=512)
d = wl.desc()
f = ctx.modular_upload(d)
got = ctx.modular_inverse(f, wl.shapes(), wl.dtype)
print('ok', [np.array_equal(got[c], wl.expected[c]) for c in range(3)])
