import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth import VardctWorkload
from oracle import pyoracle
ctx = runtime.Context(0)
wls = [VardctWorkload(w, h, seed=7 + w) for (w, h) in [(24, 40), (255, 257), (300, 520), (520, 300)]]
exps = {}
nbad = 0
for rep in range(12):
    for wi, wl in enumerate(wls):
        for st in (3, 7, 15, 63):
            d = wl.desc()
            if (wi, st) not in exps:
                exps[(wi, st)] = pyoracle.vardct_render(d, st, wl.width, wl.height)[0]
                e2 = pyoracle.vardct_render(d, st, wl.width, wl.height)[0]
                assert np.array_equal(e2, exps[(wi, st)]), "oracle nondeterministic"
            exp = exps[(wi, st)]
            f = ctx.vardct_upload(d)
            got = ctx.vardct_render(f, st)
            f.free()
            bad = np.argwhere(got.view(np.uint32) != exp.view(np.uint32))
            if len(bad):
                nbad += 1
                print(rep, wi, st, len(bad), 'c', np.unique(bad[:, 0]), 'y', bad[:, 1].min(), bad[:, 1].max(), 'x', bad[:, 2].min(), bad[:, 2].max(), flush=True)
print('done, bad runs:', nbad)
