#!/usr/bin/env python3
"""Where does the end-to-end time of one 4K frame go, and what does a depth-3 pipeline reach?"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from jxl_oxide_amd import abi, runtime
from jxl_oxide_amd.synth import VardctWorkload
ctx = runtime.Context(0)
wl = VardctWorkload(3840, 2160, seed=2000)
d = wl.desc(coeff_transport="grouped")
ref = None
for it in range(5):
    t0 = time.perf_counter(); f = ctx.vardct_upload(d)
    t1 = time.perf_counter(); ctx.vardct_render(f, abi.STAGE_ALL, to_host=False); ctx.synchronize()
    t2 = time.perf_counter(); out = ctx.format_output(f, abi.FMT_U8, 1)
    t3 = time.perf_counter(); sp = ctx.upload_split(); f.free()
    t4 = time.perf_counter()
    ref = out
    print(f"serial {it}: upload {1e3*(t1-t0):.3f} (build {sp[0]:.3f} alloc+enqueue {sp[2]:.3f} call {sp[3]:.3f} h2d {sp[4]:.3f})  render+sync {1e3*(t2-t1):.3f}  format+D2H(pageable) {1e3*(t3-t2):.3f}  free {1e3*(t4-t3):.3f} ms", flush=True)
# pipelined: depth D frames in flight, pinned output ring
D, N = 3, 40
outs = [ctx.host_alloc((2160, 3840, 3), np.uint8) for _ in range(D)]
for rep in range(2):
    inflight = []
    t0 = time.perf_counter()
    for k in range(N):
        f = ctx.vardct_upload(d)
        ctx.vardct_render(f, abi.STAGE_ALL, to_host=False)
        ctx.format_output_async(f, abi.FMT_U8, outs[k % D])
        inflight.append(f)
        if len(inflight) == D:
            g = inflight.pop(0); ctx.frame_wait(g); g.free()
    for g in inflight:
        ctx.frame_wait(g); g.free()
    dt = time.perf_counter() - t0
    print(f"pipelined depth {D}: {1e3*dt/N:.3f} ms/frame = {3840*2160/1e6/(dt/N):.0f} MP/s; last output identical to the serial one: {np.array_equal(outs[(N-1)%D], ref)}", flush=True)
for o in outs: ctx.host_free(o)
ctx.close()
