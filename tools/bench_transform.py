#!/usr/bin/env python3
"""Transform-group (V1-V8) timing of one 4K frame, isolated on one stream, for a list of tuning
environments (read at jxlgpu_create, so one context per variant, all on the same box):

    python tools/bench_transform.py "" "JXLGPU_TR_WGS_PER_CU=0,0,0" "JXLGPU_TR_WGS_PER_CU=3,2,1"

Prints the event-bracketed transform group time per frame.  Run it under
`rocprofv3 --kernel-trace --stats` with ONE variant for per-kernel durations."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    variants = sys.argv[1:] or [""]
    from jxl_oxide_amd import abi, runtime
    from jxl_oxide_amd.synth import VardctWorkload
    runtime.gpu_canary()
    nz = os.environ.get("NZ")
    wl = VardctWorkload(3840, 2160, seed=2000, nz_fraction=float(nz) if nz else None)
    d = wl.desc(coeff_transport=os.environ.get("TRANSPORT", "grouped"))
    stages = abi.STAGE_LF | abi.STAGE_TRANSFORM
    reps = int(os.environ.get("REPS", "10"))
    for v in variants:
        saved = dict(os.environ)
        for kv in v.split():
            k, _, val = kv.partition("=")
            os.environ[k] = val
        ctx = runtime.Context(0)
        os.environ.clear()
        os.environ.update(saved)
        # 8 resident frames (4 GB of state) rendered round-robin: nothing survives in the 256 MB
        # Infinity Cache from one render of a frame to the next, as in bench.py
        frames = [ctx.vardct_upload(d) for _ in range(int(os.environ.get("FRAMES", "8")))]
        for f in frames:
            ctx.vardct_render(f, stages, to_host=False)
        ctx.synchronize()
        ctx.profile_select(1)
        for _ in range(reps):
            for f in frames:
                ctx.vardct_render(f, stages, to_host=False)
        ms, n = ctx.profile_read()
        print(f"{v or '(default)':50s} transform group {ms / n * 1e3:8.1f} us / frame ({n} frames)", flush=True)
        # the same frames through the batched launches (all stages: that is what a batch runs)
        for g, name in ((1, "transform"), (2, "post")):
            ctx.vardct_render_batch(frames, abi.STAGE_ALL)
            ctx.synchronize()
            ctx.profile_select(g)
            for _ in range(reps):
                ctx.vardct_render_batch(frames, abi.STAGE_ALL)
            ms, n = ctx.profile_read()
            print(f"{'':50s} batched {name:9s} {ms / (n * len(frames)) * 1e3:8.1f} us / frame ({n} batches of {len(frames)})", flush=True)
        ctx.profile_select(-1)
        import time
        dt = 1e9
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.vardct_render_batch(frames, abi.STAGE_ALL)
            ctx.synchronize()
            dt = min(dt, time.perf_counter() - t0)
        print(f"{'':50s} batched all stages: {dt / (reps * len(frames)) * 1e6:8.1f} us / frame wall = {3840 * 2160 * reps * len(frames) / dt / 1e9:.1f} GP/s", flush=True)
        for f in frames:
            f.free()
        ctx.close()


if __name__ == "__main__":
    main()
