"""Times the Modular stage on BASELINE config 3 (8K Squeeze lossy, XYB, EPF iters 1) and config 1
(256x256 lossless RGB8) through the C ABI; prints one JSON line each.  Parity-test cases, not the
headline bench (bench.py)."""
import json
import sys
import time

sys.path.insert(0, '.')
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth_modular import ModularWorkload

ctx = runtime.Context(0)
for name, wl, stages in [
    ("cfg3 8K Modular Squeeze lossy i16 + XYB dequant + EPF1 + sRGB",
     ModularWorkload(7680, 4320, kind="squeeze", lossy=True, i16=True, epf_iters=1),
     abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT),
    ("cfg3 (i32 buffers)", ModularWorkload(7680, 4320, kind="squeeze", lossy=True, i16=False, epf_iters=1),
     abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT),
    ("cfg1 256x256 lossless RGB8", ModularWorkload(256, 256, kind="lossless_rgb8"),
     abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT),
]:
    f = ctx.modular_upload(wl.desc())
    for _ in range(2):
        ctx.modular_render(f, stages, to_host=False)
    ctx.synchronize()
    ctx.profile_select(3)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.modular_render(f, stages, to_host=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    inv_ms, k = ctx.profile_read()
    ctx.profile_select(-1)
    mp = wl.width * wl.height / 1e6
    print(json.dumps({"workload": name, "ms_per_frame": round(dt * 1e3, 3), "MP/s": round(mp / dt, 1),
                      "inverse_transforms_ms": round(inv_ms / max(k, 1), 3)}), flush=True)
    f.free()
