"""Times the Modular stage on BASELINE config 3 (8K Squeeze lossy i16, XYB, EPF iters 2) through the
C ABI: inverse transforms (profile group 3) and the whole render.  `python tools/bench_modular.py
[width height]`; run under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import json
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth_modular import ModularWorkload

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
runtime.gpu_canary()
ctx = runtime.Context(0)
stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
wl = ModularWorkload(W, H, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=3)
frames = [ctx.modular_upload(wl.desc()) for _ in range(3)]  # 3 device copies: nothing survives in the Infinity Cache
for f in frames:
    ctx.modular_render(f, stages, to_host=False)
ctx.synchronize()
ctx.profile_select(3)
n = 4
t0 = time.perf_counter()
for _ in range(n):
    for f in frames:
        ctx.modular_render(f, stages, to_host=False)
ctx.synchronize()
dt = (time.perf_counter() - t0) / (n * len(frames))
inv_ms, k = ctx.profile_read()
print(json.dumps({"workload": f"{W}x{H} Modular Squeeze lossy i16 + XYB dequant + EPF2 + sRGB", "ms_per_frame": round(dt * 1e3, 3),
                  "MP/s": round(W * H / 1e6 / dt, 1), "inverse_transforms_ms": round(inv_ms / max(k, 1), 3)}), flush=True)
