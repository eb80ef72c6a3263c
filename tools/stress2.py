import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from jxl_oxide_amd import abi, runtime
which = sys.argv[1]
t_end = time.time() + float(sys.argv[2])
if which == "modular":
    from jxl_oxide_amd.synth_modular import ModularWorkload
    ctx = runtime.Context(0)
    wls = [ModularWorkload(w, h, kind=k, lossy=True, i16=i16, seed=w) for (w, h, k, i16) in
           [(600, 333, "squeeze", True), (333, 200, "squeeze", False), (130, 97, "raw", True), (256, 256, "squeeze", True),
            (70, 45, "squeeze", False), (9, 200, "squeeze", True), (300, 270, "lossless_rgb8", True)]]
    n = 0
    while time.time() < t_end:
        for wl in wls:
            f = ctx.modular_upload(wl.desc())
            ctx.modular_inverse(f, wl.shapes(), wl.dtype)
            f.free()
            n += 1
    print("modular iterations", n, flush=True)
else:
    from jxl_oxide_amd.synth import VardctWorkload
    ctxs = [runtime.Context(0) for _ in range(4)]
    wls = [VardctWorkload(3840, 2160, seed=2000 + i) for i in range(2)]
    frames = [ctxs[i % 4].vardct_upload(wls[i % 2].desc()) for i in range(8)]
    n = 0
    while time.time() < t_end:
        for f in frames:
            f.ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
        n += 1
        if n % 20 == 0:
            for c in ctxs:
                c.synchronize()
    for c in ctxs:
        c.synchronize()
    print("vardct steps", n, flush=True)
