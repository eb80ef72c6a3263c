#!/usr/bin/env python3
"""bench.py — decode throughput of the MI355X hot path on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W

A "step" = every rank renders its batch of 4K (3840x2160) VarDCT d1 frames (XYB, Gabor + EPF
iters 2, XYB->sRGB) from decoded state already resident in HBM to f32 RGB planes in HBM: the full
hot path V1-V8 + F1 + F2 + C1-C3 of SURVEY.md §8(a).  Frames shard by frame across ranks
(BASELINE config 4: 64 frames over 8 GPUs = 8 per GPU) with no data-path collective, so scaling is
weak.  `value` = frames x 8.2944 MP / wall time, whole job.

One JSON line on rank 0, with `roofline` for the dominant kernel group (HIP events on the library's
own stream, recorded inside the timed region) and `cpu_baseline` (the CPU oracle = C restatement
of the reference's generic path, OpenMP over the reference's own rayon work units, timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W4K, H4K = 3840, 2160
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic frames generated per rank")
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--prof-group", type=int, default=None, help="force the event-bracketed kernel group")
    ap.add_argument("--streams", type=int, default=4, help="contexts (HIP streams) per rank; frames round-robin over them")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")

    import numpy as np
    import torch
    import torch.distributed as dist

    from jxl_oxide_amd import abi, runtime
    from jxl_oxide_amd.synth import VardctWorkload

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library is the thing measured (no CPU fallback)")
    if rank == 0:
        runtime.prime_gpu()  # disposable first GPU process (see runtime.prime_gpu)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctxs = [runtime.Context(local_rank) for _ in range(max(1, args.streams))]
    ctx = ctxs[0]
    stages = abi.STAGE_ALL

    # ---- synthetic frames (decoded state), uploaded once; untimed
    wls = [VardctWorkload(args.width, args.height, seed=2 * 1000 + rank * 64 + i) for i in range(args.distinct)]
    frames = []
    for i in range(args.frames_per_gpu):
        frames.append(ctxs[i % len(ctxs)].vardct_upload(wls[i % args.distinct].desc()))  # own device copy each
    mp_per_frame = args.width * args.height / 1e6

    def step():
        for f in frames:
            f.ctx.vardct_render(f, stages, to_host=False)

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- warmup; also find the dominant kernel group with event brackets
    group_ms = {}
    for w in range(max(args.warmup, 1)):
        step()
    barrier()
    mine = [f for f in frames if f.ctx is ctx]
    for g in (1, 2):  # isolated: only ctx 0's frames, one stream busy
        ctx.profile_select(g)
        for f in mine:
            ctx.vardct_render(f, stages, to_host=False)
        ms, n = ctx.profile_read()
        group_ms[g] = ms / max(n, 1)
    dominant = args.prof_group if args.prof_group is not None else max(group_ms, key=group_ms.get)
    ctx.profile_select(dominant)

    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    prof_ms, prof_n = ctx.profile_read()
    ctx.profile_select(-1)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_frames = args.frames_per_gpu * world * args.steps
    value = total_frames * mp_per_frame / elapsed

    out = None
    if rank == 0:
        f0 = frames[0]
        npx = args.width * args.height
        # algorithmic bytes of ONE launch of the bracketed group (DESIGN.md "Roofline"):
        #  transform group: 3 x i32 coeff in + 3 x f32 out + side data;  post group: 3 x f32 in +
        #  3 x f32 out + sigma.
        ncell = ((args.width + 7) // 8) * ((args.height + 7) // 8)
        if dominant == 1:
            alg_bytes = f0.algorithmic_bytes(abi.STAGE_LF | abi.STAGE_TRANSFORM)
            kname = "transform_kernel<W,H> x varblock shapes (V4-V8)"
        else:
            alg_bytes = npx * 24 + ncell * 4
            kname = "post_stream_kernel<sRGB> (+ fused_post_kernel<true,2> border ring): Gabor + EPF steps 1,2 + XYB->sRGB"
        traffic = None
        try:  # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/README.md)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_4k.json")))["kernels"]
            if dominant == 2 and args.width == W4K and args.height == H4K:
                kb = 0.0
                for name, v in pmc.items():
                    if "post_stream_kernel" in name or "fused_post_kernel" in name:
                        # gfx950: FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md)
                        kb += 2.0 * v["FETCH_SIZE_KB_mean_per_dispatch"] + v["WRITE_SIZE_KB_mean_per_dispatch"]
                traffic = int(kb * 1024)
        except Exception:
            traffic = None
        avg_ms = prof_ms / max(prof_n, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": kname, "avg_launch_ms": round(avg_ms, 4), "launches": int(prof_n),
            "isolated_launch_ms": round(group_ms.get(dominant, 0.0), 4),
            "frac_isolated": round(alg_bytes / (group_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if group_ms.get(dominant) else None,
            "note": "kernel is VALU-issue bound (scalar f32, bit-exact op order), not HBM bound; with streams_per_gpu > 1 "
                    "launches of different frames overlap, so avg_launch_ms (timed region) exceeds isolated_launch_ms (warm-up, one stream busy)",
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "other_group_ms": {("transform" if g == 1 else "post"): round(v, 4) for g, v in group_ms.items()},
            "pipeline_algorithmic_frac": round(
                (f0.algorithmic_bytes(stages) * total_frames / elapsed / 1e9) / HBM_PEAK_GBS, 4),
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (bounded CPU sample)
            cpu = cpu_baseline(wls[0], stages, args.cpu_seconds, mp_per_frame)
        out = {
            "metric": "Megapixels/sec decoded (4K VarDCT d1)",
            "value": round(value, 1),
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.width}x{args.height} VarDCT d1 XYB, Gabor + EPF iters 2, XYB->sRGB f32 planar",
                "frames_per_gpu_per_step": args.frames_per_gpu,
                "streams_per_gpu": len(ctxs),
                "distinct_frames_per_gpu": args.distinct,
                "sharding": "frames across ranks, no data-path collective",
                "input": "decoded state resident in HBM (i32 coefficients, LF quant, block map)",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    for f in frames:
        f.free()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def cpu_baseline(wl, stages, seconds, mp_per_frame):
    """Times the oracle (the C restatement of the reference's generic CPU path, OpenMP with the
    reference's rayon decomposition) on this box's host cores, on whole frames of the same
    workload until ~`seconds` of wall time have been spent."""
    from oracle import pyoracle
    # The box exposes 256 hardware threads but the oracle scales only to ~16 of them (measured on
    # the GPU box: 16 thr 57.8, 32 thr 41.5, 64 thr 27.7, 256 thr 2.3 MP/s): it opens one small OpenMP
    # region per stage per frame, mirroring the reference's rayon work units.  Use what helps.
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("JXL_CPU_BASELINE_THREADS", "16")))
    cores = pyoracle.set_threads(cores)  # the env var is too late: torch already loaded an OpenMP runtime
    import numpy as np
    d = wl.desc()
    buf = np.zeros((3, wl.height, wl.width), dtype=np.float32)
    pyoracle.vardct_render(d, stages, wl.width, wl.height, out=buf)  # warm (page faults, table init)
    n, t0 = 0, time.perf_counter()
    while True:
        pyoracle.vardct_render(d, stages, wl.width, wl.height, out=buf)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 64:
            break
    return {
        "value": round(n * mp_per_frame / dt, 2), "unit": "MP/s", "cores": cores, "kind": "port",
        "sample": f"{n} full {wl.width}x{wl.height} frames of the same workload in {dt:.1f} s "
                  "(oracle/: scalar C restatement of jxl-oxide's generic path, OpenMP over groups / 8-row stripes / 65536-sample chunks)",
    }


if __name__ == "__main__":
    main()
