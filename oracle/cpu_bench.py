"""ORACLE — test infrastructure only.  Times the CPU restatement of the reference's generic path on
this box's host cores for bench.py's `cpu_baseline` leg:

    python -m oracle.cpu_bench --config 2 --procs 16 --threads 16 --seconds 6

`procs` worker processes render whole frames of the bench workload concurrently, each with
`threads` OpenMP threads (the reference's own decomposition inside a frame: per-group jobs, 8-row
stripes, 65536-sample chunks; frames in parallel are what jxl-oxide-cli's keyframe loop does,
decode.rs:293-304).  Uses oracle/_build/liboracle_native.so (-O3 -march=native, results identical
to the -O2 build: tests/test_oracle_native.py) when it is there.  Prints one JSON line."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _worker(cfg, threads, seconds, seed, start_evt, q):
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import pyoracle
    pyoracle.use_native(True)
    pyoracle.set_threads(threads)
    import numpy as np
    from jxl_oxide_amd import abi
    if cfg == 3:
        from jxl_oxide_amd.synth_modular import ModularWorkload
        wl = ModularWorkload(1920, 1080, kind="squeeze", lossy=True, i16=True, epf_iters=2, seed=seed)  # 1/16 of an 8K frame
        d = wl.desc()
        stages = abi.STAGE_ALL | abi.STAGE_MODULAR_TO_FLOAT
        run = lambda: pyoracle.modular_render(d, stages, wl.width, wl.height)
        mp_frame = wl.width * wl.height / 1e6
    else:
        from jxl_oxide_amd.synth import VardctWorkload
        if cfg == 5:
            wl = VardctWorkload(3840, 2160, seed=5000 + seed, epf_iters=3, upsampling=2, intensity_target=4000.0, hdr_pq=True)
            ow, oh = 7680, 4320
        else:
            wl = VardctWorkload(3840, 2160, seed=2000 + seed)
            ow, oh = 3840, 2160
        d = wl.desc()
        buf = np.zeros((3, oh, ow), dtype=np.float32)
        run = lambda: pyoracle.vardct_render(d, abi.STAGE_ALL, ow, oh, out=buf)
        mp_frame = ow * oh / 1e6
    run()  # warm: page faults, table init
    q.put(("ready", 0, 0.0))
    start_evt.wait()
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    q.put(("done", n * mp_frame, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=6.0)
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    q, start = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_worker, args=(a.config, a.threads, a.seconds, i % 2, start, q)) for i in range(a.procs)]
    for p in ps:
        p.start()
    for _ in ps:
        assert q.get(timeout=600)[0] == "ready"
    start.set()
    mpix, tmax = 0.0, 0.0
    for _ in ps:
        tag, m, dt = q.get(timeout=600)
        mpix += m
        tmax = max(tmax, dt)
    for p in ps:
        p.join()
    print(json.dumps({"MP_per_s": round(mpix / tmax, 2), "procs": a.procs, "threads": a.threads,
                      "cores": a.procs * a.threads, "seconds": round(tmax, 2)}), flush=True)


if __name__ == "__main__":
    main()
