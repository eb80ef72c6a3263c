import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth import VardctWorkload
from oracle import pyoracle
ctx = runtime.Context(0)
def run(name, **kw):
    wl = VardctWorkload(520, 300, seed=7 + 520, **kw)
    d = wl.desc()
    exp = pyoracle.vardct_render(d, 63, 520, 300)[0]
    res = []
    for rep in range(8):
        f = ctx.vardct_upload(d)
        got = ctx.vardct_render(f, 63)
        f.free()
        bad = np.argwhere(got.view(np.uint32) != exp.view(np.uint32))
        res.append((len(bad), tuple(np.unique(bad[:, 0])) if len(bad) else ()))
    print(name, res, flush=True)
run('default')
run('tf linear', tf=abi.TF_LINEAR)
run('no filters', epf_iters=0, gabor=False)
run('no filters linear', epf_iters=0, gabor=False, tf=abi.TF_LINEAR)
run('pq', hdr_pq=True, intensity_target=4000.0)
