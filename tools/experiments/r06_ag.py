"""Round 6: how long does the HOST take to enqueue one 8K Modular frame (config 3)?  If that is the frame time with several contexts,
the job is launch-bound on the calling thread."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth_modular import ModularWorkload

wl = ModularWorkload(7680, 4320, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=3, residual=6)
stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
ctxs = [runtime.Context(0) for _ in range(3)]
frames = [c.modular_upload(wl.desc()) for c in ctxs]
for c, f in zip(ctxs, frames):
    c.modular_render(f, stages, to_host=False)
for c in ctxs:
    c.synchronize()
# one context: enqueue time of a frame with an idle device, then the device time
c, f = ctxs[0], frames[0]
for rep in range(3):
    t0 = time.perf_counter(); c.modular_render(f, stages, to_host=False); t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
    print(f"one frame: enqueue {1e3 * (t1 - t0):.3f} ms, until done {1e3 * (t2 - t0):.3f} ms", flush=True)
# three contexts, 12 frames: enqueue-only time against the total
t0 = time.perf_counter()
for k in range(12):
    ctxs[k % 3].modular_render(frames[k % 3], stages, to_host=False)
t1 = time.perf_counter()
for c in ctxs:
    c.synchronize()
t2 = time.perf_counter()
print(f"12 frames on 3 contexts: enqueue {1e3 * (t1 - t0) / 12:.3f} ms per frame, total {1e3 * (t2 - t0) / 12:.3f} ms per frame")
